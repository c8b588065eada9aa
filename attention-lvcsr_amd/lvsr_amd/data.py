"""Batch pipeline with the reference's semantics (lvsr/datasets/__init__.py:122-310) on in-memory examples: append
`<eol>` / prepend `<bol>`, length filter on the recordings, sort-k-batches length bucketing, batching, zero padding with
masks (fuel `Padding`, libs/fuel/fuel/transformers/__init__.py:691-720), transposition to time-major and C-contiguity.
Output = the dictionaries `SpeechRecognizer.cost` / `Trainer.train_step` consume.

The reference reads Fuel HDF5 files; h5py is not part of this image, so datasets come in as arrays (`ArrayDataset`,
`.npz` via `ArrayDataset.from_npz`).  An HDF5 reader is SURVEY.md §8f N3.
"""
import numpy


class ArrayDataset(object):
    """recordings: list of (T_i, F) float arrays; labels: list of int sequences (without <eol>)."""
    def __init__(self, recordings, labels, num_characters, eos_label=None, bos_label=None, uttids=None):
        assert len(recordings) == len(labels)
        self.recordings = [numpy.asarray(r, dtype=numpy.float32) for r in recordings]
        self.labels = [numpy.asarray(l, dtype=numpy.int64) for l in labels]
        self.num_characters = int(num_characters)
        self.eos_label = self.num_characters - 1 if eos_label is None else int(eos_label)
        self.bos_label = bos_label
        self.uttids = list(uttids) if uttids is not None else list(range(len(recordings)))

    @property
    def num_examples(self):
        return len(self.recordings)

    def dim(self):
        return int(self.recordings[0].shape[1])

    @classmethod
    def from_npz(cls, path, **kw):
        z = numpy.load(path, allow_pickle=False)
        shapes, flat = z["recordings_shapes"], z["recordings"]
        recs, off = [], 0
        for t, f in shapes:
            recs.append(flat[off: off + t * f].reshape(t, f))
            off += t * f
        ll, labs, off = z["labels_lengths"], [], 0
        for n in ll:
            labs.append(z["labels"][off: off + n])
            off += n
        return cls(recs, labs, int(z["num_characters"]), **kw)


class Data(object):
    def __init__(self, datasets, batch_size, validation_batch_size=None, sort_k_batches=None, max_length=None,
                 normalization=None, add_eos=True, eos_label=None, add_bos=0, prepend_eos=False, pad_frames_to=None,
                 pad_labels_to=None):
        """datasets: {'train': ArrayDataset, 'valid': ..., ...}; normalization: (mean (F,), std (F,)) or None.
        pad_frames_to / pad_labels_to (not in the reference): round the padded batch lengths up to a multiple, so that the
        time-loop hipGraphs (one per distinct (T, L)) are re-used across minibatches; masked positions are exact no-ops."""
        assert not prepend_eos                                              # lvsr/datasets/__init__.py:165
        self.datasets = datasets
        self.batch_size = batch_size
        self.validation_batch_size = batch_size if validation_batch_size is None else validation_batch_size
        self.sort_k_batches = sort_k_batches
        self.max_length = max_length
        self.normalization = normalization
        self.add_eos = add_eos
        self._eos_label = eos_label
        self.add_bos = add_bos
        self.pad_frames_to = pad_frames_to
        self.pad_labels_to = pad_labels_to

    @property
    def info_dataset(self):
        return self.datasets["train"]

    @property
    def num_labels(self):
        return self.info_dataset.num_characters

    @property
    def eos_label(self):
        return self._eos_label if self._eos_label else self.info_dataset.eos_label

    @property
    def bos_label(self):
        return self.info_dataset.bos_label

    def num_features(self):
        return self.info_dataset.dim()

    def _examples(self, ds, order):
        for i in order:
            rec, lab = ds.recordings[i], ds.labels[i]
            if self.add_eos:                                                # :267-270
                lab = numpy.hstack([lab, [self.eos_label]])
            if self.add_bos:                                                # :271-276
                if self.bos_label is None:
                    raise Exception("No bos label given")
                lab = numpy.hstack([self.add_bos * [self.bos_label], lab])
            if self.max_length and len(rec) > self.max_length:              # :278-279 (filter on source 0)
                continue
            yield rec, lab.astype(numpy.int64)

    def get_stream(self, part, batches=True, shuffle=True, num_examples=None, rng=None, seed=None):
        """Generator over one epoch: examples (recordings (T,F), labels (L,)) or, with batches=True, padded dictionaries."""
        ds = self.datasets[part]
        n = ds.num_examples if num_examples is None else num_examples
        order = numpy.arange(n)
        if shuffle:
            rng = rng if rng is not None else numpy.random.RandomState(seed)
            order = rng.permutation(n)
        stream = self._examples(ds, order)
        if self.sort_k_batches and batches:                                 # :281-293
            stream = self._sort_k(stream, self.batch_size * self.sort_k_batches)
        if self.normalization is not None:
            mean, std = self.normalization
            stream = (((r - mean) / std, l) for r, l in stream)
        stream = ((numpy.asarray(r, numpy.float32), l) for r, l in stream)  # ForceFloatX, :297
        if not batches:
            return stream
        bs = self.batch_size if part == "train" else self.validation_batch_size
        return self._batches(stream, bs)

    @staticmethod
    def _sort_k(stream, chunk):
        buf = []
        for ex in stream:
            buf.append(ex)
            if len(buf) == chunk:
                for e in sorted(buf, key=lambda e: len(e[0])):
                    yield e
                buf = []
        for e in sorted(buf, key=lambda e: len(e[0])):
            yield e

    @staticmethod
    def pad_batch(examples, pad_frames_to=None, pad_labels_to=None):
        """fuel Padding + switch_first_two_axes + ForceCContiguous (:303-309)."""
        B = len(examples)
        T = max(len(r) for r, _ in examples)
        L = max(len(l) for _, l in examples)
        if pad_frames_to:
            T = -(-T // pad_frames_to) * pad_frames_to
        if pad_labels_to:
            L = -(-L // pad_labels_to) * pad_labels_to
        F = examples[0][0].shape[1]
        rec = numpy.zeros((T, B, F), numpy.float32)
        rmask = numpy.zeros((T, B), numpy.float32)
        lab = numpy.zeros((L, B), numpy.int64)
        lmask = numpy.zeros((L, B), numpy.float32)
        for b, (r, l) in enumerate(examples):
            rec[: len(r), b] = r
            rmask[: len(r), b] = 1.0
            lab[: len(l), b] = l
            lmask[: len(l), b] = 1.0
        return dict(recordings=rec, recordings_mask=rmask, labels=lab, labels_mask=lmask)

    def _batches(self, stream, bs):
        buf = []
        for ex in stream:
            buf.append(ex)
            if len(buf) == bs:
                yield self.pad_batch(buf, self.pad_frames_to, self.pad_labels_to)
                buf = []
        if buf:
            yield self.pad_batch(buf, self.pad_frames_to, self.pad_labels_to)
