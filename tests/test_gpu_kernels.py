"""GPU parity tests (run on the MI355X box with -m gpu): HIP library through the C-ABI vs the CPU oracle."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from oracle import lvsr_oracle as O
from lvsr_amd import spec, synthetic, native
from lvsr_amd.params import ParameterStore, Workspace
from lvsr_amd.bricks import Encoder

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("transA,transB", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(70, 37, 29), (16, 16, 4), (257, 130, 1033), (512, 768, 2048), (12800, 768, 512), (768, 512, 12800)])
def test_sgemm(gpu_device, transA, transB, M, N, K):
    lib = native.get()
    rng = numpy.random.RandomState(0)
    A = torch.tensor(rng.normal(size=(K, M) if transA else (M, K)), dtype=torch.float32, device=gpu_device)
    B = torch.tensor(rng.normal(size=(N, K) if transB else (K, N)), dtype=torch.float32, device=gpu_device)
    C0 = torch.tensor(rng.normal(size=(M, N)), dtype=torch.float32, device=gpu_device)
    bias = torch.tensor(rng.normal(size=(N,)), dtype=torch.float32, device=gpu_device)
    ws = torch.empty(1 << 22, device=gpu_device)
    C = C0.clone()
    lib.sgemm(A, B, C, transA=transA, transB=transB, alpha=0.5, beta=2.0, bias=bias, ws=ws)
    ref = 0.5 * ((A.T if transA else A).double() @ (B.T if transB else B).double()) + 2.0 * C0.double() + bias.double()
    assert_allclose(C.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4 * numpy.sqrt(K))


def _enc_cfg(Hs, sub, F=40):
    return dict(input_dim=F, num_phonemes=6, dims_bidir=Hs, subsample=sub, dim_dec=4, dim_matcher=7,
                attention_type="content", post_merge_dims=None, embed_outputs=True)


@pytest.mark.parametrize("use_graph,persistent", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("Hs,sub,B,T,use_mask", [([3, 3], [1, 2], 3, 13, True), ([20], [3], 17, 7, True),
                                                   ([64, 48], [2, 1], 16, 60, True), ([32], [1], 5, 40, False),
                                                   ([130], [1], 33, 21, True), ([512], [2], 8, 12, True),
                                                   ([256], [1], 16, 10, True), ([192], [1], 20, 9, False)])
def test_encoder_forward_backward(gpu_device, Hs, sub, B, T, use_mask, use_graph, persistent):
    lib = native.get()
    cfg = _enc_cfg(Hs, sub)
    params = synthetic.make_params(cfg, seed=3)
    batch = synthetic.make_batch(cfg, B, T, 4, seed=5, ragged=True)
    x = torch.tensor(batch["recordings"])
    m = torch.tensor(batch["recordings_mask"]) if use_mask else None
    orc = O.OracleRecognizer(cfg, params, dtype=torch.float64)
    enc_ref, mask_ref = orc.encode(x.double(), None if m is None else m.double())
    rng = numpy.random.RandomState(0)
    dy = torch.tensor(rng.normal(size=tuple(enc_ref.shape)), dtype=torch.float64)
    (enc_ref * dy).sum().backward()
    store = ParameterStore(cfg, gpu_device, params)
    enc = Encoder(spec.Dims(cfg), store, lib, Workspace(gpu_device), use_graph=use_graph, use_persistent=persistent)
    side = torch.cuda.Stream(gpu_device)     # hipGraph capture needs a non-null stream
    xd, md, dyd = x.to(gpu_device), None if m is None else m.to(gpu_device), dy.float().to(gpu_device)
    torch.cuda.synchronize()
    for rep in range(2):      # second pass replays the cached graphs
        with torch.cuda.stream(side):
            out, out_mask = enc.apply(xd, md)
            enc.backward(dyd)
        torch.cuda.synchronize()
        enc.check_persistent()
        assert_allclose(out.cpu().numpy(), enc_ref.detach().numpy(), rtol=1e-4, atol=1e-5)
        assert_allclose(out_mask.cpu().numpy(), mask_ref.numpy())
        for name, g in store.g.items():
            if "/encoder/" not in name:
                continue
            ref = orc.p[name].grad.numpy()
            scale = max(1e-3, numpy.abs(ref).max())
            assert_allclose(g.cpu().numpy() / scale, ref / scale, atol=1e-4, rtol=0, err_msg=name)
    if use_graph:
        assert lib._lvsr_graph_count() > 0, "hipGraph capture did not engage"
    if persistent:
        assert any(k[0].endswith(".sync") for k in enc.ws._bufs), "persistent mode did not engage"


# ---- whole recognizer on the GPU: cost matrix, alignments, all parameter gradients -------------------
from conftest import load_golden                                   # noqa: E402
from lvsr_amd.bricks.recognizer import SpeechRecognizer            # noqa: E402

SMALL = ["tiny_conv_expanding", "tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_mean", "tiny_conv_logistic",
         "tiny_conv_relu", "tiny_conv_bottom", "tiny_conv_postmerge2", "tiny_content_embed",
         "tiny_content_relu", "small_conv", "small_conv_median", "small_conv_expanding", "mid_conv_median",
         "tiny_conv_stack2", "tiny_content_stack3", "small_conv_stack2"]


def _setup(meta):
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"],
                                 ragged=meta["ragged"])
    return params, batch


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("case", SMALL)
def test_recognizer_vs_reference_golden(gpu_device, case, use_graph):
    """HIP path vs the fixtures produced by the reference's own Theano graph (and the oracle for full tensors)."""
    z, meta = load_golden(case)
    params, batch = _setup(meta)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=meta["cfg"], use_graph=use_graph)
    # element-wise tolerances: the long case (300 frames, 150 attended positions, 20 labels) accumulates more float32 rounding
    # per element than the tiny ones (the reference's own arithmetic in float64 differs from its float32 run by as much,
    # tests/test_oracle_golden.py TOL); the north-star bar below (cost sum 1e-4, identical argmax) is the same for all
    long_case = case.startswith("mid_")
    for rep in range(2):
        cm = rec.cost_and_gradients(batch)
        torch.cuda.synchronize()
        cmn = cm.cpu().numpy()
        assert_allclose(cmn, z["cost_matrix"], rtol=5e-4 if long_case else 2e-4, atol=2e-5)
        assert abs(cmn.sum() - z["cost_sum"]) / abs(z["cost_sum"]) < 1e-4              # north_star: 1e-4 relative
        w = rec.generator.last["weights"].cpu().numpy()
        assert_allclose(w, z["weights"], rtol=1e-2 if long_case else 2e-4, atol=2e-5 if long_case else 2e-6)
        assert (w.argmax(axis=2) == z["weights_argmax"]).all()                          # bit-exact alignment indices
        assert_allclose(rec.encoded.cpu().numpy(), z["encoded"], rtol=1e-4, atol=2e-6)
        got = rec.store.get_grads()
        for name in z["grad_names"]:
            ref = z["grad:" + str(name)]
            scale = max(1e-3, numpy.abs(ref).max())
            assert_allclose(got[str(name)] / scale, ref / scale, rtol=0, atol=1e-3 if long_case else 2e-4, err_msg=str(name))


# Fingerprint = (norm, sum, dot with a fixed random direction) of a gradient tensor; sum and dot are cancellation residues far below
# the norm, so their absolute tolerance is a fraction of the NORM.  wsj_base_median: the backward chain through 100 labels under a
# window prior has a conditioning of its own — the float32 and float64 oracles differ by 5e-4 of a tensor's maximum there, the
# reference and the float32 oracle by up to 9e-4 on the norms (gen_golden.py WSJ_COND_TRAIN); the HIP path is within 1.4e-3 with its
# default kernels and 2.2e-3 under the kernel-variant knobs (`pytest --knob persist_flags=64`: another summation order in the
# encoder) — hence 3e-3 of the norm for it.
FP_ATOL = {"wsj_base_median": 3e-3, "wsj_base_ragged": 6e-3, "wsj_base_mean": 6e-3}
# norms: 2e-3 relative, 5e-3 on the two round-5 fixtures (measured: the step kernels' bidir0 state_to_gates norm 3.2e-3 above the
# reference's on the ragged batch — the float32 oracle is as far from it, 3.3e-3 of a tensor's maximum, tests/test_oracle_golden.py; the
# element-wise comparison below is the sharper statement for these fixtures)
FP_RTOL = {"wsj_base_ragged": 5e-3, "wsj_base_mean": 5e-3}
# Round 5: the full-size fixtures carry the reference's gradient ELEMENTS at fixed sample positions (2048 per tensor, small tensors
# whole: `gsub:<name>`, synthetic.grad_sample_index) — the element-wise pin the fingerprints could not give.  Bars: cosine >= 0.9999
# per tensor and max |difference| <= SAMPLE_RTOL of the tensor's maximum.  SURVEY's 1e-3 is what float64 arithmetic
# reaches against the reference's float32 run on the ragged fixture (8.5e-4, tests/test_oracle_golden.py); two float32 runs that add in
# different orders over 800 time steps x 100 labels differ by more — the float32 torch restatement is 3.3e-3 from the reference there.
# Measured (profiles/r05_full_size_parity.md; cluster kernels / step kernels, worst tensor, against the float64 oracle's full tensors):
# wsj_base 4.7e-6 / 4.9e-6, median 4.9e-4 / 2.7e-4, mean 1.7e-4 / 1.1e-3, ragged 3.0e-3 / 5.6e-3 (cosine >= 0.999996 everywhere); the
# REFERENCE's float32 run itself is 3.5e-3 (mean) / 8.5e-4 (ragged) from the float64 oracle, the float32 oracle 3.3e-3 (ragged).
SAMPLE_RTOL = {"wsj_base": 1e-4, "wsj_base_median": 3e-3, "wsj_base_mean": 6e-3, "wsj_base_ragged": 8e-3}
# Round 6 (profiles/r06_ragged_deviation.md): on the ragged fixture the deviation sits on a few (label, utterance) rows of ONE
# ill-conditioned region (utterance 2, labels 7-12) in every float32 computation — the float64 oracle on inputs moved by one float32
# rounding moves the same tensors by 1.0-1.5e-3, the torch float32 oracle is 3.7e-3 on one host and 9.2e-3 on another; masked and last
# labels contribute exactly zero in all paths.  The default path (cluster kernels, 2.95e-3 from the oracle, 3.8e-3 from the
# reference's elements) gets the tighter bar.
SAMPLE_RTOL_CLUSTER = {"wsj_base_ragged": 5e-3}


def sample_rtol(case, persistent_decoder, default=5e-3):
    if persistent_decoder and case in SAMPLE_RTOL_CLUSTER:
        return SAMPLE_RTOL_CLUSTER[case]
    return SAMPLE_RTOL.get(case, default)
FULL_SIZE = [("timit_tiny", None), ("wsj_base", None), ("wsj_deep", None), ("wsj_stack2", None), ("wsj_paper", None),
             ("wsj_base_median", True), ("wsj_base_median", False), ("wsj_base_ragged", True), ("wsj_base_ragged", False),
             ("wsj_base_mean", True), ("wsj_base_mean", False)]


def gradient_errors(got, ref, scale=None):
    """(max |got - ref| / max |ref| [or `scale`], cosine) of two tensors, float64."""
    a, b = numpy.asarray(got, numpy.float64).ravel(), numpy.asarray(ref, numpy.float64).ravel()
    scale = float(numpy.abs(b).max()) if scale is None else float(scale)
    den = numpy.sqrt((a * a).sum() * (b * b).sum())
    return float(numpy.abs(a - b).max() / max(scale, 1e-30)), (float((a * b).sum() / den) if den > 0 else 1.0)


@pytest.mark.parametrize("case,persistent_decoder", FULL_SIZE)
def test_full_size_configs_vs_reference_golden(gpu_device, case, persistent_decoder):
    """BASELINE.json configs[0], configs[1] and configs[3] at full size against the reference's outputs (fingerprints);
    wsj_stack2 = configs[1] with the two-layer stacked decoder of the wsj_jan_* configs; wsj_paper = the README-recommended model
    (250-unit layers: a decoder width that is not a multiple of 4); wsj_base_median (round 4) = configs[1] under
    window_around_median(10, 100), the prior the shipped models train with (wsj_paper.yaml:7-10, lvsr/bricks/attention.py:138-157),
    on well-conditioned parameter scales (gen_golden.WSJ_COND_TRAIN): every one of the 100 x 16 alignment argmax, the cost matrix
    and the gradient fingerprints of the reference, through the persistent decoder kernels AND through the step kernels."""
    z, meta = load_golden(case)
    params, batch = _setup(meta)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=meta["cfg"], use_persistent_decoder=persistent_decoder)
    cm = rec.cost_and_gradients(batch)
    torch.cuda.synchronize()
    rec.generator.check_persistent()
    if persistent_decoder is not None:
        assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == persistent_decoder, "persistent decoder engaged / did not engage"
    cmn = cm.cpu().numpy()
    assert abs(cmn.sum() - z["cost_sum"]) / abs(z["cost_sum"]) < 1e-4
    assert_allclose(cmn, z["cost_matrix"], rtol=1e-3, atol=1e-4)
    w = rec.generator.last["weights"].cpu().numpy()
    nb = z["weights_sub"].shape[1]
    # (wsj_base_median: sharp energies — energy_comp x 2 — turn float32 rounding of an energy into a relative error of the small
    # weights next to the peak: 1.5e-3 observed on one of 80 000 elements through the step kernels)
    # ragged fixtures: rows past an utterance's last label carry no cost; the reference's scan runs on there on its own numbers
    real = (batch["labels_mask"] > 0)
    conditioned = case.startswith("wsj_base_")
    if conditioned:
        # sharp energies (energy_comp x 2) turn float32 rounding of an energy into a relative error of the weights that compete
        # with the peak: all but one in 10 000 elements tightly, every element within 2e-3 absolute (measured: 1-2 of 80 000
        # elements 1.4-4.8 % off, 1.7e-3 absolute at most, next to a window edge of the mean prior)
        got_w, ref_w = w[:, :nb][real[:, :nb]], z["weights_sub"][real[:, :nb]]
        assert numpy.isclose(got_w, ref_w, rtol=5e-3, atol=1e-6).mean() > 1.0 - 1e-4
        assert numpy.abs(got_w - ref_w).max() < 3e-3
    else:
        assert_allclose(w[:, :nb][real[:, :nb]], z["weights_sub"][real[:, :nb]], rtol=1e-3, atol=1e-6)
    assert (w.argmax(axis=2) == z["weights_argmax"])[real].all()
    got = rec.store.get_grads()
    for name, fp in zip(z["grad_names"], z["grad_fp"]):
        mine = synthetic.fingerprint(str(name), got[str(name)])
        assert_allclose(mine, fp, rtol=FP_RTOL.get(case, 2e-3), atol=FP_ATOL.get(case, 2e-4) * max(1.0, fp[0]), err_msg=str(name))
    if ("gsub:" + str(z["grad_names"][0])) in z.files:          # the reference's gradient elements themselves
        for name in z["grad_names"]:
            name = str(name)
            idx = synthetic.grad_sample_index(name, got[name].shape)
            rel, cos = gradient_errors(got[name].ravel()[idx], z["gsub:" + name], z["gmax:" + name])
            assert rel <= sample_rtol(case, persistent_decoder) and cos >= 0.99999, (name, rel, cos)


# the float64 oracle's FULL gradient tensors as the yardstick at full size (round-4 verdict, weak 1): the restatement is pinned to the
# same fixtures (tests/test_oracle_golden.py: 8.5e-4 of a tensor's maximum from the reference's elements, cost 5e-9) and costs
# ~40-90 s of host time per case on the GPU box.  One oracle run per case serves both decoder paths.
@pytest.fixture(scope="module")
def float64_oracle_gradients():
    cache = {}

    def get(case):
        if case not in cache:
            z, meta = load_golden(case)
            params, batch = _setup(meta)
            import os
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            out, grads = O.OracleRecognizer(meta["cfg"], params, dtype=torch.float64).cost_and_grads(batch)
            cache[case] = (out["cost_matrix"].detach().numpy(), out["weights"].detach().numpy().argmax(axis=2), grads)
        return cache[case]
    return get


@pytest.mark.parametrize("persistent_decoder", [True, False])
@pytest.mark.parametrize("case", ["wsj_base_ragged", "wsj_base_mean", "wsj_base_median", "wsj_base"])
def test_full_size_full_gradient_tensors_vs_float64_oracle(gpu_device, float64_oracle_gradients, case, persistent_decoder):
    """Every element of every parameter gradient at BASELINE configs[1] size — all-ones and RAGGED masks, no prior / median / mean
    window priors, cluster kernels and step kernels — against the float64 oracle: cosine >= 0.99999 and max |difference| <=
    SAMPLE_RTOL[case] of the tensor's maximum, per tensor; summed cost to 1e-5; alignment argmax of every real label identical."""
    z, meta = load_golden(case)
    params, batch = _setup(meta)
    ref_cm, ref_arg, ref_grads = float64_oracle_gradients(case)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=meta["cfg"], use_persistent_decoder=persistent_decoder)
    cm = rec.cost_and_gradients(batch).cpu().numpy()
    torch.cuda.synchronize()
    rec.generator.check_persistent()
    rec.encoder.check_persistent()
    assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == persistent_decoder
    assert abs(cm.sum() - ref_cm.sum()) / abs(ref_cm.sum()) < 1e-5
    assert_allclose(cm, ref_cm, rtol=1e-3, atol=2e-4)
    real = batch["labels_mask"] > 0
    if case != "wsj_base":          # (wsj_base: scale-1 random weights, flat alignments — the argmax of the reference fixture is checked above)
        assert (rec.generator.last["weights"].cpu().numpy().argmax(axis=2) == ref_arg)[real].all()
    got = rec.store.get_grads()
    for name, ref in ref_grads.items():
        rel, cos = gradient_errors(got[name], ref)
        assert rel <= sample_rtol(case, persistent_decoder) and cos >= 0.99999, (name, rel, cos)


def test_colsum_many_gpu(gpu_device):
    from test_emu_gemm import run_colsum_many
    from lvsr_amd import native
    run_colsum_many(gpu_device, native.get())


def test_gemm_tile_shape_independence_gpu(gpu_device):
    from test_emu_gemm import run_tile_shape_independence
    from lvsr_amd import native
    run_tile_shape_independence(gpu_device, native.get())


# ---- beam search on the GPU vs the hypotheses the reference produced --------------------------------
@pytest.mark.parametrize("case", ["tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_logistic", "tiny_conv_relu",
                                  "tiny_conv_bottom", "tiny_conv_postmerge2", "tiny_content_embed", "tiny_content_relu", "small_conv_median",
                                  "tiny_conv_stack2", "tiny_content_stack3", "small_conv_stack2"])
def test_beam_search_vs_reference_golden(gpu_device, case):
    """Twice: the first search of a shape runs its positions eagerly / captures the step graph, the second replays it."""
    from test_emu_beam import run_beam_case
    run_beam_case(case, gpu_device, None)
    run_beam_case(case, gpu_device, None)


@pytest.mark.parametrize("case", ["tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_logistic", "tiny_conv_relu", "tiny_conv_bottom",
                                  "tiny_conv_postmerge2", "tiny_content_embed", "tiny_content_relu", "small_conv_median",
                                  "tiny_conv_stack2", "tiny_content_stack3", "small_conv_stack2"])
def test_batched_beam_search_vs_single_searches_and_golden(gpu_device, case):
    """BeamSearch.search_batch: the fixture's whole ragged batch decoded side by side == every utterance decoded alone == the
    reference's hypotheses for the fixture's utterance.  Twice (eager / captured, then replayed step graphs)."""
    from test_emu_beam import run_batched_case
    run_batched_case(case, gpu_device, None)
    run_batched_case(case, gpu_device, None)


def test_topk_smallest_on_the_gpu(gpu_device):
    from test_emu_beam import check_smallest
    from lvsr_amd.search import BeamSearch
    z, meta = load_golden("tiny_conv_median")
    rec = SpeechRecognizer(device=gpu_device, params=synthetic.make_params(meta["cfg"], seed=5), net_config=meta["cfg"])
    check_smallest(BeamSearch(4, rec))


def test_beam_step_graph_replay_equals_eager_steps(gpu_device):
    """The same search three times on one recognizer (eager positions, capture, pure replay) and once on a recognizer without
    graphs: identical hypotheses and costs; the device loop leaves no host synchronisation but the polls."""
    z, meta = load_golden("small_conv_median")
    params, batch = _setup(meta)
    x = batch["recordings"][: int(batch["recordings_mask"][:, 1].sum()), 1]
    kw = dict(char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")
    results = []
    for use_graph in (True, True, True, False):
        if not results or not use_graph:
            rec = SpeechRecognizer(device=gpu_device, params=params, net_config=meta["cfg"], use_graph=use_graph)
            rec.init_beam_search(6)
        results.append(rec.beam_search({"recordings": x}, **kw))
    for outs, costs in results[1:]:
        assert outs == results[0][0] and costs == results[0][1]


def test_wsj_deep_shapes_vs_oracle(gpu_device):
    """The 512-unit kernels paths (fixed NQ = 8 contraction, K = 1536 decoder contraction) at WSJ-deep layer shapes but short
    sequences and a ragged batch, against the float64 oracle (the full-size WSJ-deep step is checked against the reference's own
    output in test_full_size_configs_vs_reference_golden)."""
    cfg = spec.wsj_deep()
    cfg["dims_bidir"] = [512, 512]
    cfg["subsample"] = [1, 2]
    params = synthetic.make_params(cfg, seed=12)
    batch = synthetic.make_batch(cfg, 8, 30, 7, seed=21, ragged=True)
    orc = O.OracleRecognizer(cfg, params, dtype=torch.float64)
    out, grads = orc.cost_and_grads(batch)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    cm = rec.cost_and_gradients(batch)
    torch.cuda.synchronize()
    ref = out["cost_matrix"].detach().numpy()
    assert abs(float(cm.sum()) - ref.sum()) / abs(ref.sum()) < 1e-4
    assert_allclose(cm.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)
    w = rec.generator.last["weights"].cpu().numpy()
    assert (w.argmax(axis=2) == out["weights"].detach().numpy().argmax(axis=2)).all()
    got = rec.store.get_grads()
    for name, g in grads.items():
        scale = max(1e-3, numpy.abs(g).max())
        assert numpy.abs(got[name] - g).max() / scale < 5e-4, name


def test_one_recognizer_many_shapes_gpu(gpu_device):
    """Same recognizer, changing minibatch shapes (workspace views, graph cache keyed by shape), incl. bucket-padded batches
    whose extra masked frames / labels must be exact no-ops."""
    z, meta = load_golden("small_conv_median")
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"])
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    orc = O.OracleRecognizer(cfg, params, dtype=torch.float64)
    for k, (B, T, L) in enumerate([(5, 50, 12), (3, 31, 7), (7, 64, 14), (5, 50, 12)]):
        batch = synthetic.make_batch(cfg, B, T, L, seed=60 + k, ragged=True)
        out, grads = orc.cost_and_grads(batch)
        cm = rec.cost_and_gradients(batch)
        torch.cuda.synchronize()
        ref = out["cost_matrix"].detach().numpy()
        assert_allclose(cm.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
        got = rec.store.get_grads()
        for name, g_ in grads.items():
            scale = max(1e-3, numpy.abs(g_).max())
            assert numpy.abs(got[name] - g_).max() / scale < 3e-4, (name, k)
        if k == 0:      # the same batch padded to a bucket: identical costs on the real positions
            pad = {kk: numpy.concatenate([v, numpy.zeros((6,) + v.shape[1:], v.dtype)], 0) for kk, v in batch.items()}
            cm2 = rec.cost_and_gradients(pad).cpu().numpy()
            assert_allclose(cm2[:L], cm.cpu().numpy(), rtol=1e-5, atol=1e-6)
            assert (cm2[L:] == 0).all()


def test_whole_step_graph_region_gpu(gpu_device):
    """Trainer.train_step on the GPU = one graph region per minibatch shape (eager pass, capture, then replays), with the
    optimiser inside.  Eight steps on changing batches of two shapes must track the ORACLES step by step: the float64 torch
    restatement of the recognizer (oracle/lvsr_oracle.py, pinned to the reference's goldens) for cost and gradients, fed into the
    restatement of the reference's step rules (oracle/optimizer_oracle.py, pinned to the reference's own known-answer tests and
    Theano-evaluated fixtures) for the parameter update — and end at the same parameters."""
    from collections import OrderedDict
    from oracle import optimizer_oracle as OO
    from lvsr_amd.training import Trainer
    z, meta = load_golden("tiny_conv_median")
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"])
    conf = dict(gradient_threshold=100.0, rules=("momentum", "adadelta"), scale=0.1, momentum=0.0, decay_rate=0.95, epsilon=1e-8,
                max_norm=1.0)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    trainer = Trainer(rec, **conf)
    rules = OO.TrainingRules(**conf)
    cur = OrderedDict((k, v.copy()) for k, v in params.items())
    shapes = [(4, 40, 9), (4, 40, 9), (3, 28, 6), (4, 40, 9), (4, 40, 9), (3, 28, 6), (4, 40, 9), (3, 28, 6)]
    for k, (B, T, L) in enumerate(shapes):
        batch = synthetic.make_batch(cfg, B, T, L, seed=90 + k, ragged=True)
        cost = float(trainer.train_step(batch).sum())
        out, grads = O.OracleRecognizer(cfg, cur, dtype=torch.float64).cost_and_grads(batch)
        ref = float(out["cost_matrix"].sum())
        assert abs(cost - ref) <= 2e-4 * abs(ref), (k, cost, ref)
        # the reference divides the summed cost by the batch size before differentiating (lvsr/main.py:340-345)
        cur = rules.step(cur, OrderedDict((n, (grads[n] / B).astype(numpy.float32)) for n in cur))
    assert rec._regions, "no graph region was recorded"
    states = [s for s in rec._regions.values()]
    assert any(s["seen"] >= 3 for s in states) and not any(s.get("bad") for s in states)
    got = rec.get_parameter_values()
    for name in got:
        scale = max(1e-3, numpy.abs(cur[name]).max())
        assert numpy.abs(got[name] - cur[name]).max() / scale < 2e-3, name
