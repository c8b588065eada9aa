"""Beam search through the emulated HIP generation step vs the hypotheses the reference produced (golden fixtures)."""
import numpy
import torch
import pytest
from numpy.testing import assert_allclose

from conftest import load_golden
from emu import emu_lib
from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.search import CandidateNotFoundError

CASES = ["tiny_conv_nowindow", "tiny_conv_median", "tiny_conv_logistic", "tiny_conv_relu", "tiny_conv_bottom", "tiny_content_embed",
         "tiny_content_relu"]


def run_beam_case(case, device, lib):
    z, meta = load_golden(case)
    if not meta.get("beam"):
        pytest.skip("no beam fixture")
    params = synthetic.make_params(meta["cfg"], seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(meta["cfg"], meta["B"], meta["T"], meta["L"], seed=meta["batch_seed"], ragged=meta["ragged"])
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=meta["cfg"])
    for bi, b in enumerate(meta["beam"]):
        s = dict(b["settings"])
        utt = s.pop("utt", 0)
        bs = s.pop("beam_size")
        tl = int(batch["recordings_mask"][:, utt].sum())
        x = batch["recordings"][:tl, utt]
        rec.init_beam_search(bs)
        if b.get("error"):
            with pytest.raises(CandidateNotFoundError):
                rec.beam_search({"recordings": x}, **s)
            continue
        outs, costs = rec.beam_search({"recordings": x}, **s)
        assert outs == b["outputs"], (case, bi)                                  # bit-exact hypotheses
        assert_allclose(costs, b["costs"], rtol=2e-5, atol=2e-5)
        key = "analyze%d_cost" % bi
        if key in z.files and outs:
            c, w, _ = rec.analyze({"recordings": x}, numpy.array(outs[0]))
            assert_allclose(c, z[key], rtol=2e-4, atol=2e-5)
            assert_allclose(w, z["analyze%d_weights" % bi], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_beam_search_emulated(case):
    run_beam_case(case, "cpu", emu_lib())


def test_generation_states_are_not_overwritten_by_the_next_step():
    """The beam search keeps the tensors `generation_next_states` returned across the next `generation_logprobs` call (it
    only re-gathers them when hypotheses are dropped): they must not alias the step buffers."""
    z, meta = load_golden("tiny_conv_median")
    cfg = meta["cfg"]
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(cfg, seed=5, scale=meta["scale"]), lib=emu_lib(), net_config=cfg)
    gen = rec.generator
    x = numpy.random.RandomState(0).normal(size=(24, cfg["input_dim"])).astype(numpy.float32)
    rec.compute_contexts(x)
    st = gen.generation_initial_states(3)
    S, W = st["states"], st["weights"]
    for step in range(3):
        gen.generation_logprobs(S, W, step)
        nxt = gen.generation_next_states(S, W, step, numpy.array([1, 2, 0]))
        S, W = nxt["states"], nxt["weights"]
        keepS, keepW = S.clone(), W.clone()
        gen.generation_logprobs(S, W, step + 1)
        assert torch.equal(S, keepS) and torch.equal(W, keepW)
