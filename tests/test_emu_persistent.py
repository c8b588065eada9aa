"""The persistent cluster BiGRU kernels (csrc/encoder_persist.hip: one launch per layer and pass, work-groups exchanging the
phase vectors through {epoch,value} granules) on the emulator with concurrent work-groups (one OS thread per work-group),
against the float64 oracle and against the step kernels.  On the GPU they are covered by tests/test_gpu_kernels.py."""
import os

import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from emu import emu_lib
from oracle import lvsr_oracle as O
from lvsr_amd import spec, synthetic
from lvsr_amd.bricks import Encoder
from lvsr_amd.params import ParameterStore, Workspace


@pytest.fixture
def concurrent_lib(request):
    lib = emu_lib()
    lib._dll.hipemu_set_concurrent(1)
    # utterances per cluster (1, 2, 4 or 8): decides how many work-groups (OS threads here) a launch has
    lib.set_knob("persist_rows", request.node.callspec.params.get("rows", 8))
    try:
        yield lib
    finally:
        lib._dll.hipemu_set_concurrent(0)
        lib.set_knob("persist_rows", 0)


# H <= 128: one work-group per cluster (no exchange); 128 < H <= 256: 4 work-groups exchange the phase vectors;
# H > 256: 16 work-groups (launches of more than 64 work-groups are not run concurrently by the emulator: B and rows chosen so)
@pytest.mark.parametrize("Hs,sub,B,T,use_mask,rows", [([20], [1], 3, 7, True, 1), ([32, 16], [2, 1], 5, 8, True, 2),
                                                        ([40], [1], 17, 5, False, 8), ([140], [1], 3, 6, True, 1),
                                                        ([130, 24], [1, 2], 6, 5, True, 4), ([260], [1], 3, 4, True, 2),
                                                        ([260], [1], 2, 3, False, 1)])          # rows = 1 at H > 256: the unit-blocked forward kernel
def test_persistent_encoder_matches_oracle_and_step_kernels(concurrent_lib, Hs, sub, B, T, use_mask, rows):
    run_against_oracle_and_step_kernels(concurrent_lib, Hs, sub, B, T, use_mask)


# The publish form is chosen per cluster and launch from the work-groups' XCC_ID (persist.h cluster_shares_xcd): plain stores
# when the members share an XCD, write-through stores otherwise.  HIPEMU_XCDS=8 places block b on "XCD" b % 8 as the MI355X
# does.  With the default numbering every cluster then sits on one XCD whatever the batch size (padded grid, persist.h
# cluster_of_block: B = 3 and 5 below leave part of the grid idle) — the plain path; with knob persist_flags = 2 consecutive
# blocks form a cluster and straddle XCDs — the write-through path.  Both must give the same numbers.
@pytest.mark.parametrize("Hs,sub,B,T,use_mask,rows,xcds,flags", [([140], [1], 3, 6, True, 1, 8, 0), ([140], [1], 3, 6, True, 1, 8, 2),
                                                                   pytest.param([140, 130], [1, 2], 5, 5, True, 1, 8, 0, marks=pytest.mark.slow),
                                                                   ([260], [1], 2, 4, False, 2, 4, 2)])
def test_persistent_encoder_under_xcd_placement(concurrent_lib, monkeypatch, Hs, sub, B, T, use_mask, rows, xcds, flags):
    monkeypatch.setenv("HIPEMU_XCDS", str(xcds))
    concurrent_lib.set_knob("persist_flags", flags)
    try:
        run_against_oracle_and_step_kernels(concurrent_lib, Hs, sub, B, T, use_mask)
    finally:
        concurrent_lib.set_knob("persist_flags", 0)


def run_against_oracle_and_step_kernels(lib, Hs, sub, B, T, use_mask):
    cfg = dict(input_dim=6, num_phonemes=6, dims_bidir=Hs, subsample=sub, dim_dec=4, dim_matcher=7,
               attention_type="content", post_merge_dims=None, embed_outputs=True)
    params = synthetic.make_params(cfg, seed=3)
    batch = synthetic.make_batch(cfg, B, T, 4, seed=5, ragged=True)
    x = torch.from_numpy(batch["recordings"])
    m = torch.from_numpy(batch["recordings_mask"]) if use_mask else None
    orc = O.OracleRecognizer(cfg, params, dtype=torch.float64)
    results = {}
    for persistent in (True, False):
        store = ParameterStore(cfg, torch.device("cpu"), params)
        enc = Encoder(spec.Dims(cfg), store, lib, Workspace(torch.device("cpu")), use_graph=False, use_persistent=persistent)
        assert enc.use_persistent == persistent
        y, ym = enc.apply(x, m)
        dy = torch.from_numpy(numpy.random.RandomState(9).normal(size=tuple(y.shape)).astype(numpy.float32))
        enc.backward(dy)
        if persistent:
            enc.check_persistent()
            assert any(k[0].endswith(".sync") for k in enc.ws._bufs), "persistent mode did not engage"
        results[persistent] = (y.clone(), store.get_grads(), dy)
    y_p, g_p, dy = results[True]
    y_s, g_s, _ = results[False]
    assert_allclose(y_p.numpy(), y_s.numpy(), rtol=1e-5, atol=1e-6)
    enc_ref, _ = orc.encode(torch.from_numpy(batch["recordings"]).double(),
                            torch.from_numpy(batch["recordings_mask"]).double() if use_mask else None)
    assert_allclose(y_p.numpy(), enc_ref.detach().numpy(), rtol=2e-4, atol=2e-5)
    for name in g_s:
        if "/encoder/" in name:
            scale = max(1e-3, numpy.abs(g_s[name]).max())
            assert numpy.abs(g_p[name] - g_s[name]).max() / scale < 2e-5, name


def test_step_of_an_aborted_cluster_is_skipped_on_the_device_and_recovered():
    """A persistent cluster kernel that gives up waiting raises the (sticky) abort word of its workspace; lvsr_guard_collect carries
    it into the guard word in front of the gradient bucket and lvsr_opt_step skips the whole step on the device: parameters, rule
    state and clipping statistics unchanged.  Trainer.recover() (round-5 policy) first keeps the cluster kernels and leaves CUs
    free; the batch run again gives the update of an undisturbed step.  A second abort right away puts the run on the step kernels."""
    from lvsr_amd.bricks.recognizer import SpeechRecognizer
    from lvsr_amd.training import Trainer
    lib = emu_lib()
    lib._dll.hipemu_set_concurrent(1)
    lib.set_knob("persist_rows", 1)
    try:
        cfg = dict(input_dim=6, num_phonemes=6, dims_bidir=[140], subsample=[1], dim_dec=4, dim_matcher=7,
                   attention_type="content", post_merge_dims=None, embed_outputs=True)
        params = synthetic.make_params(cfg, seed=3)
        batch = synthetic.make_batch(cfg, 2, 5, 3, seed=5, ragged=True)
        rules = dict(gradient_threshold=5.0, rules=("momentum", "adadelta"), scale=0.5, decay_rate=0.9, epsilon=1e-6, max_norm=1.0,
                     adaptive_clipping=True)
        # reference: the same two steps without any disturbance, on the step kernels
        ref = SpeechRecognizer(device="cpu", params=params, lib=lib, net_config=cfg, use_persistent=False, use_persistent_decoder=False)
        tr_ref = Trainer(ref, distributed=False, **rules)
        tr_ref.train_step(batch)
        tr_ref.train_step(batch)
        rec = SpeechRecognizer(device="cpu", params=params, lib=lib, net_config=cfg, use_persistent=True, use_persistent_decoder=False)
        tr = Trainer(rec, distributed=False, **rules)
        tr.train_step(batch)
        assert not tr.step_was_skipped() and any(k[0] == "enc0.sync" for k in rec.ws._bufs)
        before = rec.store.get_values()
        state = {k: v.copy() for k, v in tr.state_dict().items() if k != "layout"}
        [t for k, t in rec.ws._bufs.items() if k[0] == "enc0.sync"][0][0] = 1          # "a work-group of the cluster was never scheduled"
        tr.train_step(batch)
        assert tr.step_was_skipped()
        for k, v in rec.store.get_values().items():
            assert (v == before[k]).all(), "a skipped step changed %s" % k
        for k, v in tr.state_dict().items():
            if k != "layout":
                assert (numpy.asarray(v) == state[k]).all(), "a skipped step changed the optimiser's %s" % k
        action = tr.recover()
        assert action["action"] == "cluster_reserve" and lib.get_knob("cluster_reserve") == Trainer.RECOVER_RESERVE
        tr.train_step(batch)                     # the batch again, still on the cluster kernels
        assert not tr.step_was_skipped() and rec.encoder.use_persistent
        got, want = rec.store.get_values(), ref.store.get_values()
        for k in want:
            assert_allclose(got[k], want[k], rtol=2e-4, atol=2e-6, err_msg=k)
        [t for k, t in rec.ws._bufs.items() if k[0] == "enc0.sync"][0][0] = 1          # again, right away
        tr.train_step(batch)
        assert tr.step_was_skipped()
        action = tr.recover()
        assert action["action"] == "step_kernels" and not rec.encoder.use_persistent
        tr.train_step(batch)
        assert not tr.step_was_skipped()
    finally:
        lib._dll.hipemu_set_concurrent(0)
        lib.set_knob("persist_rows", 0)
        lib.set_knob("cluster_reserve", 0)
