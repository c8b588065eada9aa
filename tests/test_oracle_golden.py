"""Pin the CPU oracle (oracle/lvsr_oracle.py) against (a) the reference's own known-answer tests and
(b) golden fixtures produced by running the reference itself (oracle/theano_harness/gen_golden.py)."""
import itertools

import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from conftest import load_golden
from oracle import lvsr_oracle as O
from lvsr_amd import synthetic

SMALL_CASES = ["tiny_conv_expanding", "tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_mean", "tiny_conv_logistic",
               "tiny_conv_relu", "tiny_conv_bottom", "tiny_conv_postmerge2",
               "tiny_content_embed", "tiny_content_relu", "small_conv", "small_conv_median",
               "small_conv_expanding", "mid_conv_median",
               "tiny_conv_stack2", "tiny_content_stack3", "small_conv_stack2"]


def test_conv1d_reference_golden():
    # /root/reference/tests/test_conv1d.py:6-13
    a = torch.tensor([[1.0, 2, 3], [1, 0, 1]])
    b = torch.tensor([[2, 1], [1, 3.0]])
    assert_allclose(O.conv1d_valid(a, b).numpy(), [[[5, 8], [5, 9]], [[1, 2], [3, 1]]])
    assert_allclose(O.conv1d_full(a, b).numpy(), [[[2, 5, 8, 3], [1, 5, 9, 9]], [[2, 1, 2, 1], [1, 3, 1, 3]]])


def test_smallest_reference_golden():
    # /root/reference/libs/blocks/tests/test_search.py:65-69
    a = numpy.array([[3, 6, 4], [1, 2, 7]])
    ind, mins = O.smallest(a, 2)
    assert numpy.all(numpy.array(ind) == numpy.array([[1, 1], [0, 1]]))
    assert numpy.all(mins == [1, 2])


def test_gru_one_step_reference_golden():
    # /root/reference/libs/blocks/tests/bricks/test_recurrent.py:432-455 (gate activation Tanh there)
    h0 = 0.1 * numpy.array([[1, 1, 0], [0, 1, 1]], dtype=numpy.float32)
    x = 0.1 * numpy.array([[1, 2, 3], [4, 5, 6]], dtype=numpy.float32)
    zi, ri = (h0 + x) / 2, -x
    W = 2 * numpy.ones((3, 3), dtype=numpy.float32)
    z = numpy.tanh(h0.dot(W) + zi)
    r = numpy.tanh(h0.dot(W) + ri)
    h1 = z * numpy.tanh((r * h0).dot(W) + x) + (1 - z) * h0
    t = torch.tensor
    got = O.gru_step(t(h0), t(x), t(numpy.hstack([zi, ri])), t(W), t(numpy.hstack([W, W])), gate_act=torch.tanh)
    assert_allclose(h1, got.numpy(), rtol=1e-6)


def test_gru_masked_sequence_and_bidirectional_reference_semantics():
    # test_recurrent.py:457-495 (masked 24-step sequence) and :498-535 (backward = forward on reversed input/mask)
    rng = numpy.random.RandomState(1)
    W = rng.normal(0, 1, (3, 3)).astype(numpy.float32)
    Wg = rng.normal(0, 1, (3, 6)).astype(numpy.float32)
    x = 0.1 * numpy.asarray(list(itertools.permutations(range(4))), dtype=numpy.float32)
    x = numpy.ones((24, 4, 3), dtype=numpy.float32) * x[..., None]
    ri = 0.3 - x
    zi = 2 * ri
    mask = numpy.ones((24, 4), dtype=numpy.float32)
    mask[12:24, 3] = 0
    h = numpy.zeros((25, 4, 3), dtype=numpy.float32)
    for i in range(1, 25):
        z = numpy.tanh(h[i - 1].dot(Wg[:, :3]) + zi[i - 1])
        r = numpy.tanh(h[i - 1].dot(Wg[:, 3:]) + ri[i - 1])
        h[i] = numpy.tanh((r * h[i - 1]).dot(W) + x[i - 1])
        h[i] = z * h[i] + (1 - z) * h[i - 1]
        h[i] = mask[i - 1, :, None] * h[i] + (1 - mask[i - 1, :, None]) * h[i - 1]
    t = torch.tensor
    gi = numpy.concatenate([zi, ri], axis=2)
    got = O.gru_sequence(t(x), t(gi), t(mask), t(W), t(Wg), torch.zeros(3), gate_act=torch.tanh)
    assert_allclose(h[1:], got.numpy(), rtol=1e-4, atol=1e-6)
    fwd_on_rev = O.gru_sequence(t(x[::-1].copy()), t(gi[::-1].copy()), t(mask[::-1].copy()), t(W), t(Wg),
                                torch.zeros(3), gate_act=torch.tanh)
    bwd = O.gru_sequence(t(x), t(gi), t(mask), t(W), t(Wg), torch.zeros(3), reverse=True, gate_act=torch.tanh)
    assert_allclose(fwd_on_rev.numpy()[::-1], bwd.numpy(), rtol=1e-6)


def _oracle_for(meta, dtype):
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"], scales=meta.get("scales"))
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"],
                                 ragged=meta["ragged"])
    return O.OracleRecognizer(meta["cfg"], params, dtype=dtype), batch


# element-wise tolerances; the long case (300 frames, 150 attended positions, 20 labels, float32 reference) accumulates more
# rounding per element than the tiny ones — the north-star bar (cost sum within 1e-4, identical argmax) is the same for all
TOL = {"mid_conv_median": dict(cost=3e-4, weights=1e-2, watol=2e-5, grad=1e-3, energies=2e-2)}      # grad: SURVEY 8(d) bar


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("case", SMALL_CASES)
def test_cost_alignment_gradients_vs_reference(case, dtype):
    z, meta = load_golden(case)
    tol = dict(dict(cost=2e-5, weights=1e-4, watol=2e-6, grad=5e-5, energies=5e-5), **TOL.get(case, {}))
    orc, batch = _oracle_for(meta, dtype)
    out, grads = orc.cost_and_grads(batch)
    cm = out["cost_matrix"].detach().numpy()
    assert_allclose(cm, z["cost_matrix"], rtol=tol["cost"], atol=2e-6)
    assert abs(cm.sum() - z["cost_sum"]) / abs(z["cost_sum"]) < 1e-5          # north_star: 1e-4 relative
    w = out["weights"].detach().numpy()
    assert_allclose(w, z["weights"], rtol=tol["weights"], atol=tol["watol"])
    assert (w.argmax(axis=2) == z["weights_argmax"]).all()                      # bit-exact alignment indices
    assert_allclose(out["encoded"].detach().numpy(), z["encoded"], rtol=1e-4, atol=2e-6)
    if "energies" in z.files:
        assert_allclose(out["energies"].detach().numpy(), z["energies"], rtol=1e-4, atol=tol["energies"])   # float32 sums of M tanh terms
    for name in z["grad_names"]:
        name = str(name)
        ref = z["grad:" + name]
        scale = max(1e-3, numpy.abs(ref).max())
        assert_allclose(grads[name] / scale, ref / scale, rtol=0, atol=tol["grad"], err_msg=name)


@pytest.mark.parametrize("case", SMALL_CASES)
def test_beam_search_vs_reference(case):
    z, meta = load_golden(case)
    if not meta.get("beam"):
        pytest.skip("no beam fixture in this case")
    orc, batch = _oracle_for(meta, torch.float32)
    for bi, b in enumerate(meta["beam"]):
        s = dict(b["settings"])
        utt = s.pop("utt", 0)
        bs = s.pop("beam_size")
        tl = int(batch["recordings_mask"][:, utt].sum())
        if b.get("error"):
            # the reference raised blocks.search.CandidateNotFoundError (search.py:379-380): part of the contract
            assert b["error"] == "CandidateNotFoundError"
            with pytest.raises(LookupError):
                orc.beam_search(batch["recordings"][:tl, utt], bs, **s)
            continue
        outs, costs = orc.beam_search(batch["recordings"][:tl, utt], bs, **s)
        assert outs == b["outputs"], (case, bi)
        assert_allclose(costs, b["costs"], rtol=1e-5, atol=1e-5)
        key = "analyze%d_cost" % bi
        if key in z.files and outs:
            hyp = numpy.array(outs[0], dtype=numpy.int64)
            c, w = orc.analyze(batch["recordings"][:tl, utt], hyp)
            assert_allclose(c, z[key], rtol=1e-4, atol=1e-5)
            assert_allclose(w, z["analyze%d_weights" % bi], rtol=1e-4, atol=2e-6)


# wsj_base_median (round 4): WSJ-base under window_around_median(10, 100) — the prior the shipped models train with — on the
# well-conditioned parameter scales of gen_golden.WSJ_COND_TRAIN: ALL 100 x 16 alignment argmax of the reference, its cost matrix
# and gradient fingerprints (the float32 and float64 restatements agree with each other and with the reference there)
# round 5: wsj_base_ragged (configs[1] with T_i in [400, 800] and L_i in [50, 100]: the reference's mask semantics at full size)
# and wsj_base_mean (window_around_mean(30, 40)); these fixtures (and the regenerated wsj_base / wsj_base_median) also carry the
# reference's gradient ELEMENTS at fixed sample positions (`gsub:<name>`, synthetic.grad_sample_index): the float32 restatement is
# within 3.3e-3 of a tensor's maximum of them (another order of float32 additions over 800 steps x 100 labels), the float64 one
# within 8.5e-4 — i.e. the reference's own float32 run sits 8.5e-4 from exact arithmetic
def sampled_gradient_errors(z, grads):
    """-> (worst max |got - ref| / max |ref tensor|, worst cosine) over the tensors, on the fixture's sampled elements."""
    worst, wcos = 0.0, 1.0
    for name in z["grad_names"]:
        name = str(name)
        idx = synthetic.grad_sample_index(name, grads[name].shape)
        a, b = numpy.asarray(grads[name], numpy.float64).ravel()[idx], z["gsub:" + name].astype(numpy.float64)
        worst = max(worst, float(numpy.abs(a - b).max() / float(z["gmax:" + name])))
        den = numpy.sqrt((a * a).sum() * (b * b).sum())
        wcos = min(wcos, float((a * b).sum() / den) if den > 0 else 1.0)
    return worst, wcos


@pytest.mark.parametrize("case", ["timit_tiny", "wsj_stack2", "wsj_paper", "wsj_base_median", "wsj_base_ragged", "wsj_base_mean"])
def test_full_size_config_vs_reference(case):
    z, meta = load_golden(case)
    orc, batch = _oracle_for(meta, torch.float32)
    out, grads = orc.cost_and_grads(batch)
    cm = out["cost_matrix"].detach().numpy()
    assert abs(cm.sum() - z["cost_sum"]) / abs(z["cost_sum"]) < 1e-5
    # (wsj_base_mean: one of the 1 600 label costs is 2.1e-4 off — a window edge of the mean prior one position over for that label)
    assert_allclose(cm, z["cost_matrix"], rtol=1e-3 if case == "wsj_base_mean" else 1e-4, atol=1e-5)
    w = out["weights"].detach().numpy()
    nb = z["weights_sub"].shape[1]
    real = batch["labels_mask"] > 0          # (ragged fixtures: rows past an utterance's last label carry no cost)
    got_w, ref_w = w[:, :nb][real[:, :nb]], z["weights_sub"][real[:, :nb]]
    if case.startswith("wsj_base_"):
        # sharp energies (energy_comp x 2) turn float32 rounding of an energy into a relative error of the weights that compete with
        # the peak: all but a few in 10 000 elements to 5e-3, every element within 1e-2 absolute (measured: 3.2e-3 on the mean fixture, where a window edge sits one position over for one label)
        assert numpy.isclose(got_w, ref_w, rtol=5e-3, atol=1e-6).mean() > 1.0 - 5e-4
        assert numpy.abs(got_w - ref_w).max() < 1e-2
    else:
        assert_allclose(got_w, ref_w, rtol=1e-3, atol=1e-6)
    assert (w.argmax(axis=2) == z["weights_argmax"])[real].all()
    for name, fp in zip(z["grad_names"], z["grad_fp"]):
        got = synthetic.fingerprint(str(name), grads[str(name)])
        # (wsj_base_median: the gradients of this fixture are conditioned to ~1e-3 of a tensor's norm, tests/test_gpu_kernels.py FP_ATOL)
        # (the two round-5 fixtures: the float32 restatement is 3.3e-3 of a tensor's maximum from the reference's elements; the
        # element-wise comparison below is the sharper statement there)
        loose = case in ("wsj_base_ragged", "wsj_base_mean")
        assert_allclose(got, fp, rtol=5e-3 if loose else 2e-3,
                        atol=(6e-3 if loose else 2e-3 if case.startswith("wsj_base_") else 2e-4) * max(1.0, fp[0]), err_msg=str(name))
    if ("gsub:" + str(z["grad_names"][0])) in z.files:
        worst, wcos = sampled_gradient_errors(z, grads)
        # (measured, float32 oracle vs the reference's elements: ragged 3.3e-3, mean 7.1e-3 — the float64 oracle: 8.5e-4 / 3.5e-3 —
        # median and the unconditioned wsj_base far below)
        assert worst < 1e-2 and wcos > 0.99999, (worst, wcos)


@pytest.mark.slow
@pytest.mark.parametrize("case", ["wsj_base_ragged", "wsj_base_mean", "wsj_base_median"])
def test_float64_oracle_vs_the_reference_gradient_elements_at_full_size(case):
    """The float64 restatement against the reference's own gradient elements (~90 s per case, hence `--runslow`): within ~1e-3 of
    every tensor's maximum on the ragged and median fixtures (SURVEY 8(d)'s bar), 3.5e-3 on the mean fixture — this is what lets the
    GPU tests use its FULL tensors as the yardstick."""
    z, meta = load_golden(case)
    if ("gsub:" + str(z["grad_names"][0])) not in z.files:
        pytest.skip("fixture without sampled gradient elements")
    orc, batch = _oracle_for(meta, torch.float64)
    out, grads = orc.cost_and_grads(batch)
    cm = out["cost_matrix"].detach().numpy()
    assert abs(cm.sum() - z["cost_sum"]) / abs(z["cost_sum"]) < 1e-6
    assert (out["weights"].detach().numpy().argmax(axis=2) == z["weights_argmax"]).all()
    worst, wcos = sampled_gradient_errors(z, grads)
    # measured on the MI355X box's host (profiles/r05_full_size_parity.md, "(float64 oracle) vs reference"): ragged 8.5e-4, median
    # 9.6e-4, mean 3.5e-3 (there the REFERENCE's float32 run has a window edge one position off exact arithmetic for one label)
    bound = {"wsj_base_ragged": 1.5e-3, "wsj_base_median": 1.5e-3, "wsj_base_mean": 5e-3}[case]
    assert worst < bound and wcos > 0.99999, (worst, wcos)


@pytest.mark.slow
def test_oracle_reproduces_the_full_size_wsj_deep_reference_step():
    """BASELINE.json configs[3] at full size (6x512 BiGRU, 8 x 1500 frames): the torch restatement against the reference's own
    output (tests/golden/wsj_deep.npz).  ~2 minutes per pass on 8 cores, hence `--runslow`; observed: cost 6e-8 relative,
    identical alignment argmax, gradient fingerprints within 6e-6."""
    z, meta = load_golden("wsj_deep")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    orc = O.OracleRecognizer(meta["cfg"], params, dtype=torch.float32)
    out, grads = orc.cost_and_grads(batch)
    cm = out["cost_matrix"].detach().numpy()
    assert abs(cm.astype(numpy.float64).sum() - float(z["cost_sum"])) / float(z["cost_sum"]) < 1e-5
    assert (out["weights"].detach().numpy().argmax(axis=2) == z["weights_argmax"]).all()
    for i, name in enumerate(z["grad_names"]):
        fp = z["grad_fp"][i]
        mine = synthetic.fingerprint(str(name), grads[str(name)])
        assert_allclose(mine, fp, rtol=2e-3, atol=2e-4 * max(1.0, fp[0]), err_msg=str(name))
