"""SpeechRecognizer.generate / sample and SoftmaxEmitter.emit against what the reference's own `generate` graph produced
(tests/golden/*_generate.npz, oracle/theano_harness/gen_golden.py `run_generate_case`): with the recorded uniforms the
emitted label sequences are identical, states / alignments / glimpses / costs within float32 tolerance."""
import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from conftest import load_golden
from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer


def run_generate_case(case, device, lib):
    z, meta = load_golden(case)
    cfg = meta["cfg"]
    params = synthetic.make_params(cfg, seed=meta["param_seed"], scale=meta["scale"])
    batch = synthetic.make_batch(cfg, meta["B"], meta["T"], 4, seed=meta["batch_seed"], ragged=True)
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=cfg)
    out = rec.generate(n_steps=meta["n_steps"], recordings=batch["recordings"], inputs_mask=batch["recordings_mask"],
                       uniforms=z["uniforms"])
    got = {k: v.cpu().numpy() for k, v in out.items()}
    assert numpy.array_equal(got["outputs"], z["outputs"])                      # the same labels, step by step
    assert_allclose(got["costs"], z["costs"], rtol=2e-4, atol=2e-5)
    assert_allclose(got["states"], z["states"], rtol=1e-3, atol=1e-4)
    assert_allclose(got["weights"], z["weights"], rtol=2e-4, atol=2e-6)
    assert_allclose(got["weighted_averages"], z["weighted_averages"], rtol=1e-3, atol=1e-4)
    assert (got["weights"].argmax(axis=2) == z["weights"].argmax(axis=2)).all()
    # emit alone: the reference's probabilities + the recorded uniforms give the recorded classes
    logp = numpy.log(numpy.maximum(z["probs"], 1e-30)).reshape(-1, z["probs"].shape[2]).astype(numpy.float32)
    cls, cost = rec.generator.emit(torch.from_numpy(logp).to(rec.device), uniforms=z["uniforms"].reshape(-1))
    assert numpy.array_equal(cls.cpu().numpy(), z["outputs"].reshape(-1))
    assert_allclose(cost.cpu().numpy(), z["costs"].reshape(-1), rtol=2e-4, atol=2e-5)
    # seeded draws: reproducible, and a different seed gives another sequence (the stream is torch's Philox, not MRG31k3p)
    a = rec.generate(n_steps=5, recordings=batch["recordings"], inputs_mask=batch["recordings_mask"], seed=3)["outputs"].cpu().numpy()
    b = rec.generate(n_steps=5, recordings=batch["recordings"], inputs_mask=batch["recordings_mask"], seed=3)["outputs"].cpu().numpy()
    assert numpy.array_equal(a, b) and a.shape == (5, meta["B"]) and a.min() >= 0 and a.max() < cfg["num_phonemes"]
    # sample(): one utterance without mask, default length frames / max_decoded_length_scale
    tl = int(batch["recordings_mask"][:, 0].sum())
    s = rec.sample({"recordings": batch["recordings"][:tl, 0]}, seed=5)
    assert s.shape == (int(tl / rec.max_decoded_length_scale), 1)
    init = rec.generator.initial_states(2, attended=rec.generator.preprocess(torch.zeros(6, 2, rec.d.E, device=rec.device)))
    assert int(init["outputs"][0]) == cfg["num_phonemes"] and tuple(init["states"].shape) == (2, rec.d.D_tot)


@pytest.mark.parametrize("case", ["tiny_conv_generate", "small_conv_generate", "tiny_conv_stack2_generate"])
def test_generate_emulated(case):
    from emu import emu_lib
    run_generate_case(case, "cpu", emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["tiny_conv_generate", "small_conv_generate", "tiny_conv_stack2_generate"])
def test_generate_gpu(gpu_device, case):
    run_generate_case(case, gpu_device, None)
