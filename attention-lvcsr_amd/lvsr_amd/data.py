"""Batch pipeline with the reference's semantics (lvsr/datasets/__init__.py:122-310) on in-memory examples: append
`<eol>` / prepend `<bol>`, length filter on the recordings, sort-k-batches length bucketing, batching, zero padding with
masks (fuel `Padding`, libs/fuel/fuel/transformers/__init__.py:691-720), transposition to time-major and C-contiguity.
Output = the dictionaries `SpeechRecognizer.cost` / `Trainer.train_step` consume.

Sources: arrays / `.npz` (`ArrayDataset.from_npz`), Kaldi feature tables + transcripts
(`ArrayDataset.from_kaldi`, what `bin/kaldi2fuel.py` converts), and the reference's Fuel HDF5 files
(`ArrayDataset.from_fuel_hdf5`; needs h5py, which this image does not have — the reader raises ImportError without it).
"""
import numpy


class ArrayDataset(object):
    """recordings: list of (T_i, F) float arrays; labels: list of int sequences (without <eol>)."""
    def __init__(self, recordings, labels, num_characters, eos_label=None, bos_label=None, uttids=None, char2num=None):
        assert len(recordings) == len(labels)
        self.char2num = dict(char2num) if char2num is not None else None
        self.num2char = {v: k for k, v in self.char2num.items()} if char2num is not None else None
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

    # ---- H5PYAudioDataset helpers (lvsr/datasets/h5py.py:24-46) ----
    def decode(self, labels, keep_eos=False):
        return [self.num2char[int(l)] for l in labels
                if (l != self.eos_label or keep_eos) and l != self.bos_label]

    def pretty_print(self, labels, example=None):
        return "".join(" " if c == "<spc>" else c for c in self.decode(labels))

    def monospace_print(self, labels):
        sub = {"<spc>": "_", "<noise>": "~", "<eol>": "$", "<bol>": "^"}
        return "".join(sub.get(c, c) for c in self.decode(labels, keep_eos=True))

    @classmethod
    def from_kaldi(cls, feats, text, char2num, tokens="chars", keep=None):
        """feats: `.scp` or `.ark` of (T,F) matrices; text: Kaldi `text` file 'uttid word word ...'; char2num: the
        `value_map` of the reference's HDF5 label source (symbol -> id, with '<eol>' and optionally '<spc>', '<noise>',
        '<bol>').  tokens='chars' spells the transcript out character by character with '<spc>' between words
        (the WSJ recipe); 'words' maps whitespace tokens (phone strings, TIMIT).  Utterances are joined on the key."""
        from . import kaldi_io
        reader = kaldi_io.read_mat_scp if feats.endswith(".scp") else kaldi_io.read_mat_ark
        trans = dict(kaldi_io.read_text(text))
        recs, labs, ids = [], [], []
        for key, mat in reader(feats):
            if key not in trans or (keep is not None and key not in keep):
                continue
            words = trans[key]
            if tokens == "chars":
                syms = []
                for wi, w in enumerate(words):
                    if wi:
                        syms.append("<spc>")
                    syms.extend([w] if w in char2num and len(w) > 1 else list(w))
            else:
                syms = list(words)
            labs.append([char2num[s] for s in syms])
            recs.append(mat)
            ids.append(key)
        return cls(recs, labs, len(char2num), eos_label=char2num["<eol>"], bos_label=char2num.get("<bol>"), uttids=ids,
                   char2num=char2num)

    @classmethod
    def from_fuel_hdf5(cls, path, split, sources_map=None):
        """The reference's dataset file (Fuel `H5PYDataset` layout written by bin/kaldi2fuel.py:103-360): per source a
        vlen dataset `<source>` of flattened examples with `<source>_shapes` (n, ndim) int32, the label source carrying
        a `value_map` attribute (symbol, id) records, and the `split` attribute table (split, source, start, stop,
        indices, available, comment) (lvsr/datasets/h5py.py:5-23, fuel H5PYDataset.create_split_array)."""
        try:
            import h5py
        except ImportError:
            raise ImportError("reading Fuel HDF5 datasets needs h5py, which is not installed; convert the Kaldi tables "
                              "with ArrayDataset.from_kaldi or ship an .npz (ArrayDataset.from_npz)")
        smap = dict(recordings="recordings", labels="labels")
        smap.update(sources_map or {})
        with h5py.File(path, "r") as f:
            start, stop, indices = 0, None, None
            for row in f.attrs["split"]:
                rs = row["split"].decode() if isinstance(row["split"], bytes) else row["split"]
                src = row["source"].decode() if isinstance(row["source"], bytes) else row["source"]
                if rs == split and src == smap["recordings"] and row["available"]:
                    start, stop = int(row["start"]), int(row["stop"])
                    if row["indices"]:
                        indices = numpy.asarray(f[row["indices"]])
            if stop is None:
                raise KeyError("split %r not found in %s" % (split, path))
            idx = indices if indices is not None else numpy.arange(start, stop)
            rec_ds, lab_ds = f[smap["recordings"]], f[smap["labels"]]
            shapes = f[smap["recordings"] + "_shapes"]
            recs = [numpy.asarray(rec_ds[i], dtype=numpy.float32).reshape(tuple(shapes[i])) for i in idx]
            labs = [numpy.asarray(lab_ds[i], dtype=numpy.int64) for i in idx]
            vm = {}
            for k, v in lab_ds.attrs["value_map"]:
                vm[k.decode() if isinstance(k, bytes) else str(k)] = int(v)
            ids = [u.decode() if isinstance(u, bytes) else u for u in (f["uttids"][i] for i in idx)] if "uttids" in f else None
        return cls(recs, labs, len(vm), eos_label=vm["<eol>"], bos_label=vm.get("<bol>"), uttids=ids, char2num=vm)

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

    def get_stream(self, part, batches=True, shuffle=True, num_examples=None, rng=None, seed=None, rank=0, world=1):
        """Generator over one epoch: examples (recordings (T,F), labels (L,)) or, with batches=True, padded dictionaries.
        Data parallelism (rank, world; not in the reference, whose stream lvsr/datasets/__init__.py:253-310 feeds one device):
        every rank walks the SAME seeded stream of global minibatches (`batch_size` utterances, same shuffle, same sort-k
        chunks) and keeps utterances rank::world of each; the padded lengths are those of the global minibatch, so all ranks
        run the same shapes, and every dictionary carries `global_batch_size` (the divisor of the summed cost,
        lvsr/main.py:340-345).  A trailing minibatch with fewer utterances than ranks is dropped."""
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
        return self._batches(stream, bs, rank, world)

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
    def pad_batch(examples, pad_frames_to=None, pad_labels_to=None, min_frames=0, min_labels=0):
        """fuel Padding + switch_first_two_axes + ForceCContiguous (:303-309)."""
        B = len(examples)
        T = max(min_frames, max(len(r) for r, _ in examples))
        L = max(min_labels, max(len(l) for _, l in examples))
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

    def _batches(self, stream, bs, rank=0, world=1):
        def emit(buf):
            if world == 1:
                return self.pad_batch(buf, self.pad_frames_to, self.pad_labels_to)
            if len(buf) < world:
                return None
            # pad to the lengths of the GLOBAL minibatch, then keep this rank's columns
            T = max(len(r) for r, _ in buf)
            L = max(len(l) for _, l in buf)
            full = self.pad_batch(buf[rank::world], self.pad_frames_to, self.pad_labels_to, T, L)
            full["global_batch_size"] = len(buf)
            return full
        buf = []
        for ex in stream:
            buf.append(ex)
            if len(buf) == bs:
                yield emit(buf)
                buf = []
        if buf:
            b = emit(buf)
            if b is not None:
                yield b
