"""Size-independent properties at BASELINE.json's full WSJ-base size on the MI355X (the oracle is too slow there):
batch independence of utterances, hipGraph == eager, persistent == step kernels, shard gradients add up (the
data-parallel invariant), cost at near-zero weights = N_labels * ln V."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from lvsr_amd import spec, synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer

pytestmark = pytest.mark.gpu

B, T, L = 16, 800, 100


@pytest.fixture(scope="module")
def setup(gpu_device):
    cfg = spec.wsj_base()
    params = synthetic.make_params(cfg, seed=10)
    batch = synthetic.make_batch(cfg, B, T, L, seed=77, ragged=True)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    cm = rec.cost_and_gradients(batch)
    torch.cuda.synchronize()
    return dict(cfg=cfg, params=params, batch=batch, rec=rec, cm=cm.cpu().numpy().copy(),
                w=rec.generator.last["weights"].cpu().numpy().copy(), grads=rec.store.get_grads())


def test_utterances_are_independent_at_full_size(gpu_device, setup):
    """Every utterance decoded alone (B=1, its own lengths) gives the same costs / alignments as inside the ragged batch."""
    s = setup
    solo = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"])
    for b in (0, 5, 15):
        one = {k: numpy.ascontiguousarray(v[:, b:b + 1]) for k, v in s["batch"].items()}
        cm = solo.cost(recordings=one["recordings"], inputs_mask=one["recordings_mask"], labels=one["labels"],
                       labels_mask=one["labels_mask"], save_for_backward=False).cpu().numpy()
        assert_allclose(cm[:, 0], s["cm"][:, b], rtol=2e-5, atol=2e-5)
        w = solo.generator.last["weights"].cpu().numpy()[:, 0]
        assert (w.argmax(axis=1) == s["w"][:, b].argmax(axis=1)).all()


def test_graph_replay_equals_eager_bitwise(gpu_device, setup):
    s = setup
    eager = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"], use_graph=False)
    cm = eager.cost_and_gradients(s["batch"]).cpu().numpy()
    assert (cm == s["cm"]).all()
    g = eager.store.get_grads()
    for k in g:
        assert (g[k] == s["grads"][k]).all(), k
    again = s["rec"].cost_and_gradients(s["batch"]).cpu().numpy()          # replay of the cached graphs: deterministic
    assert (again == s["cm"]).all()


def test_persistent_cluster_kernels_agree_with_step_kernels(gpu_device, setup):
    s = setup
    # the fixture's recognizer picks the encoder kernels itself; compare forced step kernels with forced persistent ones
    per = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"], use_persistent=True)
    cm = per.cost_and_gradients(s["batch"]).cpu().numpy()
    torch.cuda.synchronize()
    per.encoder.check_persistent()
    assert any(k[0].startswith("enc") and k[0].endswith(".sync") for k in per.ws._bufs), "persistent mode did not engage"
    stp = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"], use_persistent=False)
    cm_s = stp.cost_and_gradients(s["batch"]).cpu().numpy()
    assert not any(k[0].startswith("enc") and k[0].endswith(".sync") for k in stp.ws._bufs)
    assert_allclose(cm, cm_s, rtol=1e-3, atol=1e-3)
    assert abs(cm.sum() - s["cm"].sum()) / abs(s["cm"].sum()) < 1e-5
    assert_allclose(cm, s["cm"], rtol=1e-3, atol=1e-3)
    g = per.store.get_grads()
    for k in g:
        scale = max(1e-3, numpy.abs(s["grads"][k]).max())
        assert numpy.abs(g[k] - s["grads"][k]).max() / scale < 2e-3, k


# Well-conditioned parameter scales for the window priors (oracle/theano_harness/gen_golden.py WSJ_COND_TRAIN, found with
# tools/probes/wsj_conditioning_search.py): contractive recurrences and sharp energies, on which the float32 and float64 oracles agree on
# every alignment argmax and to 7e-8 on the summed cost — so two float32 implementations can be compared on ALL labels.  (With
# plain scale-1 random weights the label loop amplifies rounding ~1.7x per label and a window centre, a step function of the
# alignment, flips somewhere after ~30 labels: round 3 could only compare the first 8 labels there.)
WSJ_COND_TRAIN = {"transition.state_to": 0.3, "gatedrecurrent.state_to": 0.5, "energy_comp": 2.0, "handler": 2.0, "transform_states": 0.3}


@pytest.mark.parametrize("prior", [None, dict(type="window_around_median", before=10, after=100),
                                   dict(type="window_around_mean", before=30, after=40)])
def test_persistent_decoder_agrees_with_step_kernels(gpu_device, setup, prior):
    """The persistent label loop (csrc/decoder_persist.hip) and reverse walk (csrc/decoder_persist_bwd.hip) against the step kernels
    (five + four launches per label), full WSJ-base size, expanding and window_around_* priors: costs, alignments of ALL labels and
    every gradient.  The two differ by float32 rounding only (order of additions, the reassociated glimpse)."""
    s = setup
    cfg = dict(s["cfg"])
    params = s["params"]
    if prior is not None:
        cfg["prior"] = prior
        params = synthetic.make_params(cfg, seed=13, scale=1.0, scales=WSJ_COND_TRAIN)
    out = {}
    for persistent in (True, False):
        rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg, use_persistent_decoder=persistent)
        cm = rec.cost_and_gradients(s["batch"]).cpu().numpy()
        torch.cuda.synchronize()
        rec.generator.check_persistent()
        assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == persistent, "persistent decoder engaged / did not engage"
        assert any(k[0] == "gen.sync_bwd" for k in rec.ws._bufs) == persistent, "persistent decoder backward engaged / did not engage"
        out[persistent] = (cm, rec.generator.last["weights"].cpu().numpy(), rec.generator.last["weighted_averages"].cpu().numpy(),
                           rec.store.get_grads())
    (cm_p, w_p, wa_p, g_p), (cm_s, w_s, wa_s, g_s) = out[True], out[False]
    assert abs(cm_p.sum() - cm_s.sum()) / abs(cm_s.sum()) < 1e-5
    assert_allclose(cm_p, cm_s, rtol=1e-3, atol=1e-3)
    assert (w_p.argmax(axis=2) == w_s.argmax(axis=2)).all()            # every label of every utterance
    # element-wise: sharp energies (the conditioned scales) turn float32 rounding of an energy into a relative error of the weights
    # that compete with the peak — a handful of the 320 000 elements differ by up to 3e-3 absolute between the two float32 paths
    # (the summed cost above still agrees to 1e-5): all elements within 1e-2 absolute, all but 1 in 10 000 tightly
    assert numpy.abs(w_p - w_s).max() < 1e-2
    close = numpy.isclose(w_p, w_s, rtol=2e-3, atol=2e-6)
    assert close.mean() > 1.0 - 1e-4, "%d of %d alignment weights differ" % ((~close).sum(), close.size)
    assert_allclose(wa_p, wa_s, rtol=2e-2, atol=2e-3)
    # gradients: under the window priors the backward chain through 100 labels has a conditioning of its own even on these scales
    # (float32 vs float64 oracle: 5e-4 of a tensor's maximum, reference vs float32 oracle 9e-4 of the norms, gen_golden.py
    # WSJ_COND_TRAIN — tuned under the median prior); the two GPU paths (hardware exp / rcp, reassociated sums) are within 6e-3
    # (median) / 1.3e-2 (mean) of a tensor's maximum there
    # (window_around_mean: the window's edges are floor / ceil of a float32 mean position — a rounding difference in the location
    # convolution moves an edge by one position for some label of some utterance; 3.2e-2 measured after the convolution's summation
    # order changed in round 4.  Round 5 put both paths of THIS batch against the float64 oracle's full tensors
    # (profiles/r05_prop_parity.md): median — persistent 8.6e-3, step 6.0e-3; mean — persistent 2.7e-2 (one edge moved), step 7.2e-3;
    # on the reference-generated all-ones fixture wsj_base_mean it is the persistent path that is closer (1.7e-4 vs 1.1e-3) and the
    # reference's own float32 run is 3.5e-3 from the oracle: an edge flips where the mean is within float32 rounding of an integer,
    # whoever computes it.  The reference-pinned bars for both priors and both paths are in
    # test_gpu_kernels.py::test_full_size_full_gradient_tensors_vs_float64_oracle)
    gtol = 2e-3 if prior is None else (5e-2 if prior["type"] == "window_around_mean" else 3e-2)
    for k in g_s:
        scale = max(1e-3, numpy.abs(g_s[k]).max())
        assert numpy.abs(g_p[k] - g_s[k]).max() / scale < gtol, k


def test_context_gradient_in_lds_equals_the_atomic_form_at_full_size(gpu_device, setup):
    """The reverse walk's opt-in LDS-resident sum of the gradient wrt the preprocessed contexts (round 6, PbGeom.DPAL, persist_flags 16384)
    against the default L2 atomics on the ragged full-size batch: the same adds in the same order — every gradient bit for bit."""
    from lvsr_amd import native
    s = setup
    lib = native.get()
    lib.set_knob("persist_flags", 16384)
    try:
        rec = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"])
        cm = rec.cost_and_gradients(s["batch"]).cpu().numpy()
        torch.cuda.synchronize()
        rec.generator.check_persistent()
        assert any(k[0] == "gen.sync_bwd" for k in rec.generator.ws._bufs), "persistent decoder backward did not engage"
        g = rec.store.get_grads()
    finally:
        lib.set_knob("persist_flags", 0)
    assert (cm == s["cm"]).all()
    for k in g:
        assert (g[k] == s["grads"][k]).all(), k


def test_persistent_decoder_in_passes_at_batch_64(gpu_device, setup):
    """Per-GPU batch 64: 64 clusters of 8 work-groups do not fit the 256 CUs at once; under the expanding prior the utterances are
    independent and the persistent kernels run two passes of 32 utterances (decoder_persist.h pd_pick_passes).  Costs and alignments
    of the real labels against the step kernels; gradients against the sum over the two half-batches run on their own (one pass
    each, the same cluster shape: the data-parallel invariant — free of the conditioning of a 64-utterance gradient on random
    weights, where the step kernels' float32 rounding alone moves every recurrent gradient by a few per cent; the decoder's passes
    meet the float64 oracle on a conditioned 128-utterance batch in test_large_per_gpu_batches_vs_float64_oracle)."""
    s = setup
    batch = synthetic.make_batch(s["cfg"], 64, 480, 60, seed=79, ragged=True)
    out = {}
    for persistent in (True, False):
        rec = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"], use_persistent_decoder=persistent)
        cm = rec.cost_and_gradients(batch).cpu().numpy()
        torch.cuda.synchronize()
        rec.generator.check_persistent()
        assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == persistent, "persistent decoder engaged / did not engage"
        assert any(k[0] == "gen.sync_bwd" for k in rec.ws._bufs) == persistent, "persistent decoder backward engaged / did not engage"
        out[persistent] = (cm, rec.generator.last["weights"].cpu().numpy(), rec.store.grad.clone())
    (cm_p, w_p, g_p), (cm_s, w_s, _) = out[True], out[False]
    assert abs(cm_p.sum() - cm_s.sum()) / abs(cm_s.sum()) < 1e-5
    assert_allclose(cm_p, cm_s, rtol=1e-3, atol=1e-3)
    # on the real labels (past an utterance's last label the recurrence runs on in both paths, on random weights chaotically: those
    # rows carry no cost and no gradient)
    real = batch["labels_mask"] > 0
    # alignment peaks: the same position in all but a handful of the 2 853 rows — flat alignments of random weights, where the two
    # largest weights of a row are closer than the float32 differences between the paths (bounded below: with |w_p - w_s| <= d
    # everywhere the step kernels' weight at the persistent path's peak is within 2 d of their own)
    mism = (w_p.argmax(axis=2) != w_s.argmax(axis=2))[real]
    assert mism.mean() < 2e-3, "%d of %d alignment peaks differ" % (mism.sum(), mism.size)
    assert numpy.abs(w_p[real] - w_s[real]).max() < 2e-2
    assert numpy.isclose(w_p[real], w_s[real], rtol=2e-3, atol=2e-6).mean() > 0.99
    rec = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"], use_persistent_decoder=True)
    total, cost = None, 0.0
    for r in range(2):
        cost += float(rec.cost_and_gradients(synthetic.shard_batch(batch, r, 2)).sum())
        total = rec.store.grad.clone() if total is None else total + rec.store.grad
    assert abs(cost - cm_p.sum()) / abs(cm_p.sum()) < 1e-5
    assert float((total - g_p).abs().max()) / float(g_p.abs().max()) < 2e-4


def test_encoder_in_passes_above_64_utterances_per_gpu(gpu_device, setup):
    """Per-GPU batches above 64: the encoder runs in passes over utterance columns on the cluster kernels (bricks.Encoder
    _apply_in_passes, round 5) while the decoder sees the whole batch.  80 ragged utterances (two passes of 40): costs, alignments and
    every gradient against the same batch with the encoder in ONE pass on the step kernels (round 4's form), and the gradient
    against the sum over the two halves run on their own (the data-parallel invariant)."""
    s = setup
    batch = synthetic.make_batch(s["cfg"], 80, 240, 30, seed=83, ragged=True)
    rec = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"])
    cm = rec.cost_and_gradients(batch).cpu().numpy()
    torch.cuda.synchronize()
    rec.encoder.check_persistent()
    assert rec.encoder._pass_cols == [(0, 40), (40, 80)], "the encoder did not run in passes"
    assert any(k[0].startswith("enc.p0_") and k[0].endswith(".sync") for k in rec.ws._bufs), "the passes did not run on the cluster kernels"
    w, g = rec.generator.last["weights"].cpu().numpy(), rec.store.grad.clone()
    one = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"])
    one.encoder.PASS_ROWS = 1 << 30
    cm1 = one.cost_and_gradients(batch).cpu().numpy()
    torch.cuda.synchronize()
    assert one.encoder._pass_cols is None
    assert abs(cm.sum() - cm1.sum()) / abs(cm1.sum()) < 1e-5
    assert_allclose(cm, cm1, rtol=1e-3, atol=1e-3)
    real = batch["labels_mask"] > 0
    mism = (w.argmax(axis=2) != one.generator.last["weights"].cpu().numpy().argmax(axis=2))[real]
    assert mism.mean() < 2e-3, "%d of %d alignment peaks differ" % (mism.sum(), mism.size)
    # gradients: the sum over the two halves run on their own (the data-parallel invariant: the accumulation between the passes); against
    # exact arithmetic the passes are checked on a conditioned batch in test_large_per_gpu_batches_vs_float64_oracle below (on THIS
    # batch — scale-1 random weights — the float32 and float64 oracles themselves differ by 5.4 % of the gradient maximum)
    total, cost = None, 0.0
    for r in range(2):
        cost += float(rec.cost_and_gradients(synthetic.shard_batch(batch, r, 2)).sum())
        total = rec.store.grad.clone() if total is None else total + rec.store.grad
    assert abs(cost - cm.sum()) / abs(cm.sum()) < 1e-5
    assert float((total - g).abs().max()) / float(g.abs().max()) < 2e-4


LARGE_BATCHES = {"80 ragged": (80, 240, 30, 83), "128 ragged": (128, 160, 20, 84)}


@pytest.fixture(scope="module")
def large_batch_oracle():
    """float64 oracle (cost matrix, alignment argmax, FULL gradient tensors) of the large-batch cases: WSJ-base layers, the
    well-conditioned scales WSJ_COND_TRAIN (the float32 oracle is within 1e-4 of the float64 one on these batches: measured in the
    build container, profiles/r06_large_batch_parity.md), ragged lengths.  One oracle run per case serves all kernel paths."""
    import os
    from oracle import lvsr_oracle as O
    cache = {}

    def get(case):
        if case not in cache:
            Bn, Tn, Ln, seed = LARGE_BATCHES[case]
            cfg = spec.wsj_base()
            params = synthetic.make_params(cfg, seed=13, scale=1.0, scales=WSJ_COND_TRAIN)
            batch = synthetic.make_batch(cfg, Bn, Tn, Ln, seed=seed, ragged=True)
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            out, grads = O.OracleRecognizer(cfg, params, dtype=torch.float64).cost_and_grads(batch)
            cache[case] = (cfg, params, batch, out["cost_matrix"].detach().numpy(), out["weights"].detach().numpy().argmax(axis=2), grads)
        return cache[case]
    return get


@pytest.mark.parametrize("path", ["encoder in passes", "encoder in one pass on the step kernels", "encoder in passes, decoder step kernels"])
@pytest.mark.parametrize("case", list(LARGE_BATCHES))
def test_large_per_gpu_batches_vs_float64_oracle(gpu_device, large_batch_oracle, case, path):
    """Per-GPU batches above 64 (round-5 verdict, weak 1): 80 ragged utterances (two encoder passes of 40) and 128 (the `strong` leg's
    shape: two passes of 64; the persistent decoder in passes of its own) — the default path (Encoder._apply_in_passes /
    _backward_in_passes on the cluster kernels), round 4's one pass on the step kernels, and the passes under the decoder's step
    kernels, EACH against the float64 oracle: cost sum 1e-5, every real-label alignment argmax, and every element of every gradient
    tensor (cosine >= 0.99999, max |difference| <= 3e-3 of the tensor's maximum).  Masks and independence of utterances as in
    libs/blocks/blocks/bricks/recurrent.py:608-624,655-663 and lvsr/bricks/__init__.py:71-78."""
    from test_gpu_kernels import gradient_errors
    cfg, params, batch, ref_cm, ref_arg, ref_grads = large_batch_oracle(case)
    Bn = batch["labels"].shape[1]
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg,
                           use_persistent_decoder=False if "decoder step" in path else None)
    if "one pass" in path:
        rec.encoder.PASS_ROWS = 1 << 30
    cm = rec.cost_and_gradients(batch).cpu().numpy()
    torch.cuda.synchronize()
    rec.encoder.check_persistent()
    rec.generator.check_persistent()
    if "one pass" in path:
        assert rec.encoder._pass_cols is None
        assert not any(k[0].startswith("enc") and k[0].endswith(".sync") for k in rec.ws._bufs), "the one-pass encoder was to run on the step kernels"
    else:
        assert rec.encoder._pass_cols == [(0, Bn // 2), (Bn // 2, Bn)], "the encoder did not run in passes"
        assert any(k[0].startswith("enc.p0_") and k[0].endswith(".sync") for k in rec.ws._bufs), "the passes did not run on the cluster kernels"
    assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == ("decoder step" not in path)
    assert abs(cm.sum() - ref_cm.sum()) / abs(ref_cm.sum()) < 1e-5
    assert_allclose(cm, ref_cm, rtol=1e-3, atol=2e-4)
    real = batch["labels_mask"] > 0
    assert (rec.generator.last["weights"].cpu().numpy().argmax(axis=2) == ref_arg)[real].all()
    got = rec.store.get_grads()
    worst = max((gradient_errors(got[name], ref)[0], name) for name, ref in ref_grads.items())
    for name, ref in ref_grads.items():
        rel, cos = gradient_errors(got[name], ref)
        assert rel <= 3e-3 and cos >= 0.99999, (name, rel, cos, worst)


def test_persistent_decoder_at_the_paper_width(gpu_device):
    """The README-recommended model (wsj_paper7: 250-unit BiGRUs, decoder and matcher, one location filter) has a decoder width that
    is not a multiple of 4: the persistent reverse walk reads its AW rows 16 bytes at a time and runs there with padded rows
    (`lvsr_attdec_plain.AW_ld`).  Persistent kernels against the step kernels: costs, alignments, every gradient."""
    from lvsr_amd import spec
    cfg = dict(spec.wsj_paper(), prior=None)              # expanding prior: no window centres, the two paths differ by rounding only
    params = synthetic.make_params(cfg, seed=21, scale=0.7)
    batch = synthetic.make_batch(cfg, 6, 400, 40, seed=22, ragged=True)
    out = {}
    for persistent in (True, False):
        rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg, use_persistent_decoder=persistent)
        cm = rec.cost_and_gradients(batch).cpu().numpy()
        torch.cuda.synchronize()
        rec.generator.check_persistent()
        assert any(k[0] == "gen.sync" for k in rec.ws._bufs) == persistent, "persistent decoder engaged / did not engage"
        assert any(k[0] == "gen.sync_bwd" for k in rec.ws._bufs) == persistent, "persistent decoder backward engaged / did not engage"
        out[persistent] = (cm, rec.generator.last["weights"].cpu().numpy(), rec.store.get_grads())
    (cm_p, w_p, g_p), (cm_s, w_s, g_s) = out[True], out[False]
    assert abs(cm_p.sum() - cm_s.sum()) / abs(cm_s.sum()) < 1e-5
    assert_allclose(cm_p, cm_s, rtol=1e-3, atol=1e-3)
    assert (w_p.argmax(axis=2) == w_s.argmax(axis=2)).all()
    assert_allclose(w_p, w_s, rtol=1e-3, atol=1e-6)
    for k in g_s:
        scale = max(1e-3, numpy.abs(g_s[k]).max())
        assert numpy.abs(g_p[k] - g_s[k]).max() / scale < 2e-3, k


def test_shard_gradients_add_up_to_the_batch_gradient(gpu_device, setup):
    """The data-parallel invariant behind the single all-reduce: grad(sum over all utterances) = sum over shards r::N."""
    s = setup
    rec = SpeechRecognizer(device=gpu_device, params=s["params"], net_config=s["cfg"])
    total, cost = None, 0.0
    for r in range(4):
        shard = synthetic.shard_batch(s["batch"], r, 4)
        cost += float(rec.cost_and_gradients(shard).sum())
        g = rec.store.grad.clone()
        total = g if total is None else total + g
    assert abs(cost - s["cm"].sum()) / abs(s["cm"].sum()) < 1e-5
    ref = setup["rec"].store.grad
    setup["rec"].cost_and_gradients(s["batch"])
    torch.cuda.synchronize()
    denom = float(ref.abs().max())
    assert float((total - ref).abs().max()) / denom < 1e-4


def test_cost_at_near_zero_weights_is_n_labels_ln_v(gpu_device):
    """Free sanity check the survey recorded for the reference (5594.415 = 1600*ln 33 at near-zero init)."""
    cfg = spec.wsj_base()
    params = synthetic.make_params(cfg, seed=1, scale=1e-4)
    params = {k: (v if not k.endswith(".b") and not k.endswith("initial_state") else numpy.zeros_like(v)) for k, v in params.items()}
    batch = synthetic.make_batch(cfg, B, T, L, seed=1234)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    cm = rec.cost(recordings=batch["recordings"], inputs_mask=batch["recordings_mask"], labels=batch["labels"],
                  labels_mask=batch["labels_mask"], save_for_backward=False)
    assert abs(float(cm.sum()) - B * L * numpy.log(33.0)) / (B * L * numpy.log(33.0)) < 1e-4


def test_concurrent_searches_equal_sequential_ones(gpu_device):
    """Several beam searches kept in flight from one host thread (one recognizer + stream each, begin / advance / finish never
    blocking) return exactly what the same searches return one after the other."""
    from conftest import load_golden
    from lvsr_amd.search import CandidateNotFoundError
    z, meta = load_golden("small_conv_median")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    rng = numpy.random.RandomState(3)
    utts = [rng.normal(size=(int(rng.randint(30, 60)), meta["cfg"]["input_dim"])).astype(numpy.float32) for _ in range(7)]
    kw = dict(char_discount=0.2, round_to_inf=1e9, stop_on="optimistic_future_cost")
    recs = [SpeechRecognizer(device=gpu_device, params=params, net_config=meta["cfg"]) for _ in range(3)]
    for r in recs:
        r.init_beam_search(6)

    def one(rec, x):
        try:
            return rec.beam_search({"recordings": x}, **kw)
        except CandidateNotFoundError:
            return None
    want = [one(recs[0], x) for x in utts]
    got, pending, slots = {}, list(range(len(utts))), [None] * len(recs)
    while pending or any(s is not None for s in slots):
        for k, rec in enumerate(recs):
            bs = rec._beam_search
            with torch.cuda.stream(rec.stream):
                if slots[k] is None:
                    if not pending:
                        continue
                    i = pending.pop(0)
                    x = utts[i]
                    slots[k] = (i, bs.begin({"recordings": x[:, None, :]}, rec.eos_label, int(x.shape[0] / rec.max_decoded_length_scale),
                                            ignore_first_eol=rec.data_prepend_eos, **kw))
                i, run = slots[k]
                if bs.advance(run, 3, wait=False):
                    try:
                        got[i] = bs.finish(run)
                    except CandidateNotFoundError:
                        got[i] = None
                    slots[k] = None
    for i in range(len(utts)):
        assert got[i] == want[i], i


def test_step_of_an_aborted_cluster_is_skipped_inside_the_graph_and_recovered(gpu_device):
    """The guard at full size, inside the replayed whole-step graph: a raised (sticky) abort word of the decoder's cluster workspace
    makes the captured lvsr_opt_step skip the step on the device — parameters and rule state bit-identical — and train_step refuses
    to go on over it.  Trainer.recover() (round 5) first KEEPS the cluster kernels and leaves CUs free (`cluster_reserve`), the batch
    run again gives the step an undisturbed run takes; a second abort right away moves the run onto the step kernels, and after
    REARM_STEPS clean steps the cluster kernels are armed again."""
    from lvsr_amd.training import Trainer
    cfg = spec.wsj_base()
    params = synthetic.make_params(cfg, seed=13, scales=WSJ_COND_TRAIN)
    batch = synthetic.make_batch(cfg, B, T, L, seed=78, ragged=True)
    rules = dict(gradient_threshold=100.0, rules=("momentum", "adadelta"), scale=0.1, momentum=0.0, decay_rate=0.95, epsilon=1e-8,
                 max_norm=1.0)
    ref = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg, use_persistent=False, use_persistent_decoder=False)
    tr_ref = Trainer(ref, distributed=False, **rules)
    for _ in range(4):
        tr_ref.train_step(batch)
    rec = SpeechRecognizer(device=gpu_device, params=params, net_config=cfg)
    tr = Trainer(rec, distributed=False, **rules)
    lib = rec.lib
    assert lib.get_knob("cluster_reserve") == 0
    try:
        for _ in range(3):                       # eager, capture, replay
            tr.train_step(batch)
        torch.cuda.synchronize()
        assert not tr.step_was_skipped()
        assert any(s["seen"] >= 3 for s in rec._regions.values())
        before = rec.store.get_values()
        state = {k: numpy.asarray(v).copy() for k, v in tr.state_dict().items() if k != "layout"}
        words = [t for k, t in rec.ws._bufs.items() if k[0] in ("gen.sync", "gen.sync_bwd")]
        assert len(words) == 2, "the decoder did not run on its cluster kernels"
        words[1][0] = 1        # "a work-group of the cluster was never scheduled"
        tr.train_step(batch)                     # a replay of the captured step
        torch.cuda.synchronize()
        assert tr.step_was_skipped()
        for k, v in rec.store.get_values().items():
            assert (v == before[k]).all(), "a skipped step changed %s" % k
        for k, v in tr.state_dict().items():
            if k != "layout":
                assert (numpy.asarray(v) == state[k]).all(), "a skipped step changed the optimiser's %s" % k
        with pytest.raises(RuntimeError):        # nobody recovered: the next step refuses
            tr.train_step(batch)
        action = tr.recover()
        assert action["action"] == "cluster_reserve" and lib.get_knob("cluster_reserve") == Trainer.RECOVER_RESERVE
        tr.train_step(batch)
        torch.cuda.synchronize()
        assert not tr.step_was_skipped() and rec.generator.use_persistent is not False and rec.encoder.use_persistent
        got, want = rec.store.get_values(), ref.store.get_values()
        for k in want:
            scale = max(1e-3, numpy.abs(want[k]).max())
            # (four AdaDelta steps from zero statistics: the first steps are sign-like and amplify the float32 differences between the
            # cluster kernels and the step kernels of the reference run)
            assert numpy.abs(got[k] - want[k]).max() / scale < 2e-2, k
        # a second abort within REARM_STEPS: step kernels for a while, then the cluster kernels again
        tr.REARM_STEPS = 2
        words = [t for k, t in rec.ws._bufs.items() if k[0] in ("gen.sync", "gen.sync_bwd")]
        words[0][0] = 1
        tr.train_step(batch)
        torch.cuda.synchronize()
        assert tr.step_was_skipped()
        action = tr.recover()
        assert action["action"] == "step_kernels" and rec.generator.use_persistent is False and not rec.encoder.use_persistent
        for _ in range(2):
            tr.train_step(batch)
            torch.cuda.synchronize()
            assert not tr.step_was_skipped() and tr._fallback is not None
        tr.train_step(batch)                     # armed again
        torch.cuda.synchronize()
        assert tr._fallback is None and rec.encoder.use_persistent and rec.generator.use_persistent is not False
        assert not tr.step_was_skipped()
        tr.train_step(batch)
        torch.cuda.synchronize()
        rec.generator.check_persistent()
        rec.encoder.check_persistent()
        assert not tr.step_was_skipped()
    finally:
        lib.set_knob("cluster_reserve", 0)
