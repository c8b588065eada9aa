"""Batch pipeline semantics (lvsr/datasets/__init__.py:253-310) and a few training steps driven by it on the emulated
recognizer (loss goes down)."""
import os

import numpy
import pytest
import torch

from lvsr_amd.data import ArrayDataset, Data


def _dataset(n=11, F=5, V=6, seed=0):
    rng = numpy.random.RandomState(seed)
    recs = [rng.normal(size=(rng.randint(6, 14), F)) for _ in range(n)]
    labs = [rng.randint(0, V - 1, size=rng.randint(2, 5)) for _ in range(n)]
    return ArrayDataset(recs, labs, num_characters=V, bos_label=V - 2)


def test_layout_eos_bos_filter_and_sorting():
    ds = _dataset()
    data = Data({"train": ds, "valid": ds}, batch_size=4, validation_batch_size=3, sort_k_batches=2, max_length=12, add_bos=2)
    batches = list(data.get_stream("train", shuffle=False))
    kept = [i for i in range(ds.num_examples) if len(ds.recordings[i]) <= 12]
    assert sum(b["labels"].shape[1] for b in batches) == len(kept)
    for b in batches:
        T, B, F = b["recordings"].shape
        assert b["recordings"].dtype == numpy.float32 and b["labels"].dtype == numpy.int64
        assert b["recordings"].flags["C_CONTIGUOUS"] and b["recordings_mask"].shape == (T, B)
        assert B <= 4 and (b["recordings_mask"].sum(0) >= 1).all()
        for j in range(B):
            t = int(b["recordings_mask"][:, j].sum()); l = int(b["labels_mask"][:, j].sum())
            assert (b["recordings"][t:, j] == 0).all() and (b["labels"][l:, j] == 0).all()      # zero padding
            assert (b["labels"][:2, j] == ds.bos_label).all() and b["labels"][l - 1, j] == ds.eos_label
    # sort-k: within every group of batch_size*k consecutive examples the input lengths are non-decreasing
    lens = [int(b["recordings_mask"][:, j].sum()) for b in batches for j in range(b["labels"].shape[1])]
    for g in range(0, len(lens), 8):
        grp = lens[g: g + 8]
        assert grp == sorted(grp)
    valid = list(data.get_stream("valid", shuffle=False))
    assert valid[0]["labels"].shape[1] == 3                                     # validation batch size
    bucketed = Data({"train": ds}, batch_size=4, pad_frames_to=8, pad_labels_to=4)
    for b in bucketed.get_stream("train", shuffle=False):
        assert b["recordings"].shape[0] % 8 == 0 and b["labels"].shape[0] % 4 == 0
        assert (b["recordings_mask"].sum(0) >= 6).all()
    ex = list(data.get_stream("train", batches=False, shuffle=True, seed=1))
    assert len(ex) == len(kept) and ex[0][0].ndim == 2


def test_rank_sharded_streams_partition_the_global_minibatches():
    """Data.get_stream(rank=, world=): the ranks' shards of every global minibatch are its utterances r::world, padded to the
    GLOBAL lengths (identical shapes on all ranks), and carry the global batch size; a trailing minibatch smaller than the
    number of ranks is dropped on every rank alike."""
    ds = _dataset(n=11)
    for sort_k in (None, 2):
        data = Data({"train": ds}, batch_size=4, sort_k_batches=sort_k)
        whole = list(data.get_stream("train", shuffle=True, seed=3))
        world = 3
        shards = [list(data.get_stream("train", shuffle=True, seed=3, rank=r, world=world)) for r in range(world)]
        assert len(whole) == 3 and whole[-1]["labels"].shape[1] == 3
        assert all(len(s) == 3 for s in shards)
        for i, gb in enumerate(whole):
            n = gb["labels"].shape[1]
            for r in range(world):
                sh = shards[r][i]
                assert sh["global_batch_size"] == n
                for k in ("recordings", "recordings_mask", "labels", "labels_mask"):
                    assert numpy.array_equal(sh[k], gb[k][:, r::world]), (k, i, r)
    # 9 examples in minibatches of 4: the last one holds a single utterance, fewer than the 2 ranks -> dropped everywhere
    data = Data({"train": _dataset(n=9)}, batch_size=4)
    assert [len(list(data.get_stream("train", shuffle=False, rank=r, world=2))) for r in range(2)] == [2, 2]


def test_training_on_the_pipeline_reduces_the_cost():
    from emu import emu_lib
    from lvsr_amd import synthetic
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    from lvsr_amd.training import Trainer
    ds = _dataset(n=6)
    data = Data({"train": ds}, batch_size=3)
    cfg = dict(input_dim=5, num_phonemes=6, dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content",
               post_merge_dims=None, embed_outputs=True, data_prepend_eos=False)
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(cfg, seed=3, scale=0.5), lib=emu_lib(), net_config=cfg)
    tr = Trainer(rec, gradient_threshold=10.0, rules=("momentum",), scale=0.05, momentum=0.0, distributed=False)
    costs = []
    for epoch in range(6):
        tot = 0.0
        for batch in data.get_stream("train", shuffle=False):
            tot += float(tr.train_step(batch).sum())
        costs.append(tot)
    assert costs[-1] < costs[0] * 0.95, costs


def test_kaldi_tables_round_trip_and_dataset(tmp_path):
    from lvsr_amd import kaldi_io
    rng = numpy.random.RandomState(0)
    mats = [("utt%d" % i, rng.normal(size=(5 + 3 * i, 4)).astype(numpy.float32)) for i in range(4)]
    mats.append(("vec", rng.normal(size=7)))                                       # float64 vector
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    kaldi_io.write_mat_ark(ark, mats, scp=scp)
    for reader, src in ((kaldi_io.read_mat_ark, ark), (kaldi_io.read_mat_scp, scp)):
        got = list(reader(src))
        assert [k for k, _ in got] == [k for k, _ in mats]
        for (_, a), (_, b) in zip(got, mats):
            assert a.dtype == b.dtype and numpy.array_equal(a, b)
    # text-format archive (copy-feats ark,t:)
    txt = str(tmp_path / "feats.txt")
    with open(txt, "w") as fh:
        for k, m in mats[:2]:
            fh.write("%s  [\n" % k)
            for r, row in enumerate(m):
                fh.write("  " + " ".join(repr(float(v)) for v in row) + (" ]\n" if r == len(m) - 1 else "\n"))
    got = list(kaldi_io.read_mat_ark(txt))
    assert [k for k, _ in got] == ["utt0", "utt1"] and numpy.allclose(got[1][1], mats[1][1])
    with open(str(tmp_path / "c.ark"), "wb") as fh:
        fh.write(b"k \x00BCM \x04")
    with pytest.raises(ValueError):
        list(kaldi_io.read_mat_ark(str(tmp_path / "c.ark")))
    # transcripts -> character labels as in the WSJ recipe
    text = str(tmp_path / "text")
    with open(text, "w") as fh:
        fh.write("utt0 AB A\nutt1 <noise> B\nutt3 BA\nother X\n")
    c2n = {"A": 0, "B": 1, "<spc>": 2, "<noise>": 3, "<eol>": 4}
    ds = ArrayDataset.from_kaldi(ark, text, c2n)
    assert ds.uttids == ["utt0", "utt1", "utt3"] and ds.eos_label == 4 and ds.num_characters == 5
    assert ds.labels[0].tolist() == [0, 1, 2, 0] and ds.labels[1].tolist() == [3, 2, 1]
    assert ds.pretty_print(ds.labels[0]) == "AB A" and ds.monospace_print([3, 2, 1, 4]) == "~_B$"
    mean, std = kaldi_io.compute_cmvn_stats(ds.recordings)
    allf = numpy.concatenate(ds.recordings)
    assert numpy.allclose(mean, allf.mean(0), atol=1e-6) and numpy.allclose(std, allf.std(0), atol=1e-5)
    data = Data({"train": ds}, batch_size=2, normalization=(mean, std))
    b = next(iter(data.get_stream("train", shuffle=False)))
    assert b["recordings"].shape[1:] == (2, 4) and b["labels"][-1].max() == 4
    with pytest.raises(ImportError):
        ArrayDataset.from_fuel_hdf5(str(tmp_path / "x.h5"), "train")


class _FakeH5:
    """A stand-in for the FEW h5py calls ArrayDataset.from_fuel_hdf5 makes (File as a context manager, root / dataset `attrs`, item access by
    name and by object reference, per-example reads of vlen datasets, `in`).  h5py is not in the image: this executes the reader's LOGIC —
    split table, subset indices, example shapes, value map — on the layout bin/kaldi2fuel.py:103-360 writes; the HDF5 container itself stays
    untested (N3 remains partial for exactly that reason)."""

    class Dataset(list):
        def __init__(self, items, attrs=None):
            list.__init__(self, items)
            self.attrs = attrs or {}
            self.ref = self                      # an "object reference" dereferences to the dataset itself

    class File(object):
        registry = {}

        def __init__(self, path, mode="r"):
            self.root = self.registry[path]
            self.attrs = self.root["__attrs__"]

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def __getitem__(self, key):
            return key if isinstance(key, _FakeH5.Dataset) else self.root[key]

        def __contains__(self, key):
            return key in self.root


def _fuel_layout(recs, labels, uttids, char2num, splits):
    """What bin/kaldi2fuel.py writes: vlen `recordings` (flattened) + `recordings_shapes`, vlen `labels` with the `value_map` records attribute,
    `uttids`, `<split>_indices` datasets, and the root `split` table of fuel's H5PYDataset.create_split_array (libs/fuel/fuel/datasets/hdf5.py:
    233-300) — (start, stop) = (-1, -1) with an indices reference, as kaldi2fuel's `add_sets` does, or a plain range without one."""
    vm = numpy.array([(k.encode(), v) for k, v in char2num.items()], dtype=[("key", "S8"), ("val", "<i4")])
    root = {"recordings": _FakeH5.Dataset([r.ravel() for r in recs]),
            "recordings_shapes": _FakeH5.Dataset([numpy.array(r.shape, numpy.int32) for r in recs]),
            "labels": _FakeH5.Dataset([numpy.asarray(l, numpy.int32) for l in labels], attrs={"value_map": vm}),
            "uttids": _FakeH5.Dataset([u.encode() for u in uttids])}
    rows = []
    for name, spec_ in splits.items():
        ref = None
        if isinstance(spec_, list):
            root[name + "_indices"] = _FakeH5.Dataset(spec_)
            ref, start, stop = root[name + "_indices"].ref, -1, -1
        else:
            start, stop = spec_
        for source in ("labels", "recordings", "uttids"):
            rows.append((name.encode(), source.encode(), start, stop, ref, True, b"."))
    rows.append((b"test", b"recordings", 0, 0, None, False, b"."))          # a split / source pair that is not available
    table = numpy.empty(len(rows), dtype=[("split", "S8"), ("source", "S12"), ("start", "<i8"), ("stop", "<i8"), ("indices", "O"),
                                          ("available", "?"), ("comment", "S1")])
    for i, r in enumerate(rows):
        table[i] = r
    root["__attrs__"] = {"split": table}
    return root


def test_fuel_hdf5_reader_logic_on_a_stand_in_for_h5py(monkeypatch):
    import sys
    import types
    rng = numpy.random.RandomState(3)
    char2num = {"<eol>": 0, "<bol>": 1, "<spc>": 2, "a": 3, "b": 4, "c": 5}
    recs = [rng.normal(size=(t, 7)).astype(numpy.float32) for t in (5, 9, 4, 11, 6)]
    labels = [[3, 4, 0], [1, 5, 2, 3, 0], [4, 0], [3, 3, 5, 0], [5, 0]]
    uttids = ["u%02d" % i for i in range(5)]
    fake = types.ModuleType("h5py")
    fake.File = _FakeH5.File
    _FakeH5.File.registry = {"wsj.h5": _fuel_layout(recs, labels, uttids, char2num, {"train": [3, 0, 4], "valid": (1, 3)})}
    monkeypatch.setitem(sys.modules, "h5py", fake)
    # a split kept as subset indices, in Kaldi's (unsorted) order, as kaldi2fuel writes them
    tr = ArrayDataset.from_fuel_hdf5("wsj.h5", "train")
    assert tr.num_examples == 3 and tr.uttids == ["u03", "u00", "u04"]
    for got, want in zip(tr.recordings, (recs[3], recs[0], recs[4])):
        assert got.dtype == numpy.float32 and got.shape == want.shape and (got == want).all()
    assert [list(l) for l in tr.labels] == [labels[3], labels[0], labels[4]]
    assert tr.num_characters == 6 and tr.eos_label == 0 and tr.bos_label == 1 and tr.char2num == char2num
    # a split kept as a contiguous range
    va = ArrayDataset.from_fuel_hdf5("wsj.h5", "valid")
    assert va.uttids == ["u01", "u02"] and (va.recordings[1] == recs[2]).all()
    # renamed sources (`sources_map` of the data section), an unavailable split, a missing one
    _FakeH5.File.registry["wsj.h5"]["fbank_dd"] = _FakeH5.File.registry["wsj.h5"]["recordings"]
    _FakeH5.File.registry["wsj.h5"]["fbank_dd_shapes"] = _FakeH5.File.registry["wsj.h5"]["recordings_shapes"]
    with pytest.raises(KeyError):
        ArrayDataset.from_fuel_hdf5("wsj.h5", "train", sources_map=dict(recordings="fbank_dd"))          # no split row names that source
    with pytest.raises(KeyError):
        ArrayDataset.from_fuel_hdf5("wsj.h5", "test")
    with pytest.raises(KeyError):
        ArrayDataset.from_fuel_hdf5("wsj.h5", "eval92")
    # the dataset feeds the pipeline like any other
    data = Data(dict(train=tr), batch_size=2, add_eos=False)
    batches = list(data.get_stream("train"))
    assert sum(int(b["recordings"].shape[1]) for b in batches) == 3 and batches[0]["recordings"].shape[2] == 7


def test_multistage_driver_on_the_emulated_recognizer(tmp_path):
    """train_multistage (lvsr/main.py:896-922): stages in `number` order, stage 2 restarts from `<stage1><restart_from>.zip`,
    per-epoch validation, `_best_ll` checkpoints, FinishAfter(num_batches / num_epochs), AdaptiveClipping always on."""
    from emu import emu_lib
    from lvsr_amd import config, main
    y = tmp_path / "two_stage.yaml"
    y.write_text("""
net:
  dims_bidir: [4]
  dim_dec: 5
  dim_matcher: 6
  attention_type: content
  embed_outputs: true
  enc_transition: !!python/name:blocks.bricks.recurrent.GatedRecurrent
  dec_transition: !!python/name:blocks.bricks.recurrent.GatedRecurrent
initialization:
  /recognizer:
    weights_init: !!python/object/apply:blocks.initialization.IsotropicGaussian [0.3]
    biases_init: !!python/object/apply:blocks.initialization.Constant [0.0]
    rec_weights_init: !!python/object/apply:blocks.initialization.Orthogonal []
    initial_states_init: !!python/object/apply:blocks.initialization.IsotropicGaussian [0.001]
training:
  gradient_threshold: 10.0
  scale: 0.05
  momentum: 0.0
  rules: [momentum]
regularization:
  max_norm: 3.0
stages:
  pretraining:
    number: 0
    training:
      num_epochs: 2
  main:
    number: 1
    training:
      num_batches: 3
      restart_from: _best_ll
      scale: 0.02
""")
    cfg = config.Configuration(str(y))
    ds = _dataset(n=6)
    data = Data({"train": ds, "valid": ds}, batch_size=3)
    save = str(tmp_path / "run")
    rec, log = main.train_multistage(cfg, data, save, device="cpu", lib=emu_lib(), distributed=False)
    assert sorted(os.listdir(save)) == ["main.zip", "main_best_ll.zip", "pretraining.zip", "pretraining_best_ll.zip"]
    stages = [r["stage"] for r in log if "stage" in r]
    assert stages == ["pretraining", "main"]
    batch_rows = [r for r in log if "train_cost" in r]
    assert len(batch_rows) == 2 * 2 + 3 and batch_rows[-1]["iterations_done"] == 3          # 2 epochs x 2 batches, then 3 batches
    assert all(r["gradient_norm_threshold"] > 0 for r in batch_rows)
    valid = [r["valid_cost"] for r in log if "valid_cost" in r]
    assert valid[1] < valid[0]                                                             # it learns
    with pytest.raises(NotImplementedError):
        main.train(dict(cfg.ordered_stages["main"], regularization={"noise": 0.075}), data, save + "/x.zip", device="cpu",
                   lib=emu_lib(), distributed=False)
    # stage 2 started from stage 1's best-likelihood checkpoint, not from a fresh initialisation
    from lvsr_amd.checkpoint import load_parameters
    a, b = load_parameters(save + "/pretraining_best_ll.zip"), load_parameters(save + "/main.zip")
    name = "/recognizer/generator/readout/bias.b" if "/recognizer/generator/readout/bias.b" in a else sorted(a)[0]
    assert numpy.abs(a[name] - b[name]).max() < 0.2 and set(a) == set(b)


def test_training_state_survives_a_restart(tmp_path):
    """A stage interrupted after one epoch and resumed from its checkpoint (parameters + `_training_state`: AdaDelta
    accumulators, adaptive-clipping statistics, counters) ends exactly where the uninterrupted run ends; a Blocks-style
    checkpoint without that member still loads as plain parameters."""
    from emu import emu_lib
    from lvsr_amd import main
    from lvsr_amd.checkpoint import load_member, load_parameters
    from lvsr_amd import blocks_compat as BC
    cfg = dict(net=dict(dims_bidir=[4], dim_dec=5, dim_matcher=6, attention_type="content", embed_outputs=True),
               initialization={"/recognizer": dict(weights_init=BC.IsotropicGaussian(0.3), biases_init=BC.Constant(0.0),
                                                   rec_weights_init=BC.Orthogonal(), initial_states_init=BC.IsotropicGaussian(0.001))},
               training=dict(gradient_threshold=10.0, scale=0.05, momentum=0.0, rules=["momentum", "adadelta"], num_epochs=2))
    data = Data({"train": _dataset(n=6), "valid": _dataset(n=6)}, batch_size=3)
    full, _ = main.train(cfg, data, str(tmp_path / "full.zip"), device="cpu", lib=emu_lib(), distributed=False)
    one = dict(cfg, training=dict(cfg["training"], num_epochs=1))
    first, _ = main.train(one, data, str(tmp_path / "part.zip"), device="cpu", lib=emu_lib(), distributed=False)
    state = load_member(str(tmp_path / "part.zip"), "_training_state")
    assert int(state["epochs_done"]) == 1 and int(state["iterations_done"]) == 2 and "ms_step" in state and "clip_state" in state
    resumed, log = main.train(cfg, data, str(tmp_path / "part.zip"), params=str(tmp_path / "part.zip"), device="cpu", lib=emu_lib(),
                              distributed=False, resume=True)
    a, b = full.store.get_values(), resumed.store.get_values()
    for k in a:
        assert numpy.array_equal(a[k], b[k]), k
    assert [r["iterations_done"] for r in log if "train_cost" in r] == [3, 4]
    assert set(load_parameters(str(tmp_path / "full.zip"))) == set(a)
    with pytest.raises(ValueError):
        bare = str(tmp_path / "bare.zip")
        full.save_params(bare)
        main.train(cfg, data, str(tmp_path / "x.zip"), params=bare, device="cpu", lib=emu_lib(), distributed=False, resume=True)
