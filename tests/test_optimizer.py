"""Optimiser: (a) pin oracle/optimizer_oracle.py to the reference's own known-answer tests
(libs/blocks/tests/algorithms/test_algorithms.py), (b) the fused HIP step (emulated here, real on the GPU box)
against that oracle on a recognizer-shaped parameter set."""
from collections import OrderedDict

import numpy
import pytest
import torch
from numpy.testing import assert_allclose

from oracle import optimizer_oracle as OO
from lvsr_amd import synthetic
from lvsr_amd.bricks.recognizer import SpeechRecognizer
from lvsr_amd.training import Trainer


def test_momentum_known_answers():          # test_algorithms.py:96-104
    a = numpy.array([3, 4], numpy.float32)
    rule = OO.Momentum(0.1, 0.5)
    for want in ([0.6, 0.8], [0.9, 1.2], [1.05, 1.4]):
        assert_allclose(rule.compute_steps({"a": 2 * a})["a"], want, rtol=1e-6)


def test_adadelta_known_answers():          # test_algorithms.py:111-119
    a = numpy.array([3, 4], numpy.float32)
    rule = OO.AdaDelta(decay_rate=0.5, epsilon=1e-7)
    for want in ([0.00044721, 0.00044721], [0.0005164, 0.0005164], [0.00056904, 0.00056904]):
        assert_allclose(rule.compute_steps({"a": 2 * a})["a"], want, rtol=1e-5)
    with pytest.raises(ValueError):
        OO.AdaDelta(-1.0)
    with pytest.raises(ValueError):
        OO.AdaDelta(2.0)


def test_step_clipping_known_answers():     # test_algorithms.py:182-192, :255-260
    g = OrderedDict([(0, numpy.float32(3.0)), (1, numpy.float32(4.0))])
    c1 = OO.step_clipping(g, 4)
    assert_allclose([c1[0], c1[1]], [12 / 5.0, 16 / 5.0], rtol=1e-6)
    c2 = OO.step_clipping(g, 5)
    assert_allclose([c2[0], c2[1]], [3.0, 4.0])
    comp = OO.Momentum(0.1, 0.0).compute_steps(OO.step_clipping(g, 4))
    assert_allclose([comp[0], comp[1]], [12 / 50.0, 16 / 50.0], rtol=1e-6)


def test_variable_clipping_known_answers():     # test_algorithms.py:199-225
    assert_allclose(OO.variable_clipping([1, 1], [3, 2], 5), [3, 2])
    assert_allclose(OO.variable_clipping([-1, -1, -1], [[3, 9, 2]], 5), [[0.78885438, 3.47213595, 0.34164079]], rtol=1e-5)
    got = OO.variable_clipping([[1, -1, 1, -1], [-1, 1, -1, 1]], [[1, 2, 3, 4], [5, 6, 7, 8]], 10, axis=1)
    assert_allclose(got, [[1, 2, 3, 4], [3.54858826, 4.79049022, 5.06478435, 6.30668631]], rtol=1e-5)


def test_remove_not_finite_known_answers():     # test_algorithms.py:312-324
    assert_allclose(OO.remove_not_finite(1.0, numpy.nan, 0.1), 0.9)
    assert_allclose(OO.remove_not_finite(2.0, numpy.inf, 0.1), 1.8)
    assert_allclose(OO.remove_not_finite(3.0, 0.123, 0.1), 0.123)
    assert_allclose(OO.remove_not_finite(1.0, numpy.nan), 0.0)


CFG = dict(input_dim=5, num_phonemes=6, dims_bidir=[3, 3], subsample=[1, 2], dim_dec=4, dim_matcher=7,
           attention_type="content_and_conv", conv_n=2, conv_num_filters=3, post_merge_dims=[8],
           post_merge_activation="maxout2", embed_outputs=True, data_prepend_eos=False)
RULES = dict(gradient_threshold=2.0, rules=("momentum", "adadelta"), scale=0.5, momentum=0.3, decay_rate=0.9,
             epsilon=1e-6, max_norm=0.9)


def run_fused_vs_oracle(device, lib, poison=False, rules=RULES, steps=3):
    params = synthetic.make_params(CFG, seed=21)
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=CFG)
    tr = Trainer(rec, distributed=False, **rules)
    orc = OO.TrainingRules(**rules)
    cur = OrderedDict((k, v.copy()) for k, v in params.items())
    rng = numpy.random.RandomState(3)
    for it in range(steps):
        grads = OrderedDict((k, rng.normal(0, 1.0, v.shape).astype(numpy.float32)) for k, v in cur.items())
        if poison and it == 1:
            grads["/recognizer/generator/readout/post_merge/bias.b"][2] = numpy.nan
        for k, g in grads.items():
            rec.store.g[k].copy_(torch.from_numpy(g))
        tr.apply_gradients(global_batch_size=4)
        if poison and it == 1:
            # a NaN anywhere makes the global norm NaN -> every step is NaN -> RemoveNotFinite(0.0) zeroes every parameter
            cur = OrderedDict((k, numpy.zeros_like(v)) for k, v in cur.items())
            got = rec.store.get_values()
            for k in cur:
                assert_allclose(got[k], cur[k], err_msg=k)
            return
        cur = orc.step(cur, OrderedDict((k, g / numpy.float32(4)) for k, g in grads.items()))
        got = rec.store.get_values()
        for k in cur:
            assert_allclose(got[k], cur[k], rtol=2e-5, atol=2e-6, err_msg="%s it %d" % (k, it))
        if orc.adaptive:
            assert_allclose(tr.gradient_threshold(), orc.adaptive.threshold, rtol=1e-6)


def test_fused_step_emulated():
    from emu import emu_lib
    run_fused_vs_oracle("cpu", emu_lib())


ADAPTIVE = dict(RULES, gradient_threshold=30.0, burn_in_steps=2, adaptive_clipping=dict(decay_rate=0.9, burnin_period=4))


def test_adaptive_clipping_known_answers():
    """AdaptiveClipping.after_batch by hand (lvsr/extensions.py:76-91)."""
    import math
    ac = OO.AdaptiveClipping(10.0, burnin_period=2, decay_rate=0.5)
    ac.after_batch(math.e)                       # log norm = 1: mean .5, mean2 .5, std .5, exp(1) at confidence 1/2
    assert_allclose(ac.threshold, 0.5 * math.exp(1.0) + 0.5 * 10.0, rtol=1e-6)
    ac.after_batch(math.e)                       # mean .75, mean2 .75, std sqrt(.1875); confidence 1
    assert_allclose(ac.threshold, math.exp(0.75 + math.sqrt(0.1875)), rtol=1e-6)
    big = OO.AdaptiveClipping(1.0, burnin_period=1, decay_rate=0.0)
    big.after_batch(1e6)
    assert big.threshold == 5.0                  # capped at 5 x the initial threshold


def test_fused_step_adaptive_clipping_and_burn_in_emulated():
    from emu import emu_lib
    run_fused_vs_oracle("cpu", emu_lib(), rules=ADAPTIVE, steps=6)


def test_trainer_from_reference_config_sections():
    from emu import emu_lib
    rec = SpeechRecognizer(device="cpu", params=synthetic.make_params(CFG, seed=21), lib=emu_lib(), net_config=CFG)
    tr = Trainer.from_config(rec, dict(gradient_threshold=100.0, scale=0.1, momentum=0.0, rules=["momentum", "adadelta"],
                                       decay_rate=0.95, epsilon=1e-8, burn_in_steps=3), dict(max_norm=1.0), distributed=False)
    assert tr.conf["use_adadelta"] and tr.conf["adaptive_clipping"] and tr.conf["adaptive_burnin"] == 500
    assert tr.gradient_threshold() == 100.0 and float(tr.clip_state[4]) == 3.0
    plain = Trainer.from_config(rec, dict(scale=0.1), distributed=False)
    assert plain.clip_state is None and plain.conf["clip_threshold"] == 0.0


def test_fused_step_nonfinite_emulated():
    from emu import emu_lib
    run_fused_vs_oracle("cpu", emu_lib(), poison=True)


@pytest.mark.gpu
def test_fused_step_gpu(gpu_device):
    run_fused_vs_oracle(gpu_device, None)
    run_fused_vs_oracle(gpu_device, None, poison=True)
    run_fused_vs_oracle(gpu_device, None, rules=ADAPTIVE, steps=6)


# ---- pinned to the reference's own AdaptiveClipping / BurnIn classes (tests/golden/adaptive_clipping.npz) -------------
def _adaptive_golden():
    import json
    from conftest import golden_path
    z = numpy.load(golden_path("adaptive_clipping"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def test_adaptive_clipping_oracle_matches_the_reference_class():
    z, meta = _adaptive_golden()
    for key in ("wired", "short"):
        m = meta[key]
        ac = OO.AdaptiveClipping(m["initial"], burnin_period=m["burnin_period"], decay_rate=m["decay_rate"])
        got = []
        for g in z["norms"]:
            ac.after_batch(g)
            got.append(ac.threshold)
        assert_allclose(got, z[key], rtol=1e-6)
    assert z["burn_in"][:, 0].tolist() == [0, 0, 0, 1, 1, 1]             # BurnIn(3): three zeroed steps, then untouched


def run_adaptive_kernel_vs_reference(device, lib):
    """Feed the fused optimiser gradients whose norms are the golden's sequence: the device-resident threshold must follow
    the thresholds the reference's AdaptiveClipping set, and BurnIn must freeze the parameters for its first steps."""
    z, meta = _adaptive_golden()
    m = meta["short"]
    params = synthetic.make_params(CFG, seed=21)
    rec = SpeechRecognizer(device=device, params=params, lib=lib, net_config=CFG)
    tr = Trainer(rec, distributed=False, gradient_threshold=m["initial"], rules=("momentum",), scale=0.01, burn_in_steps=3,
                 adaptive_clipping=dict(decay_rate=m["decay_rate"], burnin_period=m["burnin_period"]))
    n = rec.store.grad.numel()
    before = rec.store.flat.clone()
    direction = torch.randn(n, generator=torch.Generator().manual_seed(1)).to(rec.store.grad.device)
    direction /= direction.norm()
    for it, g in enumerate(z["norms"][:12]):
        rec.store.grad.copy_(direction * float(g))
        tr.apply_gradients(global_batch_size=1)
        assert_allclose(tr.gradient_norm(), float(g), rtol=1e-5)
        assert_allclose(tr.gradient_threshold(), z["short"][it], rtol=2e-5)
        if it < 3:
            assert torch.equal(rec.store.flat, before), "BurnIn step %d moved the parameters" % it
    assert not torch.equal(rec.store.flat, before)


def test_adaptive_clipping_kernel_matches_the_reference_class_emulated():
    from emu import emu_lib
    run_adaptive_kernel_vs_reference("cpu", emu_lib())


@pytest.mark.gpu
def test_adaptive_clipping_kernel_matches_the_reference_class_gpu(gpu_device):
    run_adaptive_kernel_vs_reference(gpu_device, None)


def test_recover_state_machine_emulated():
    """Trainer.recover(): the first abort keeps the cluster kernels and leaves CUs free (`cluster_reserve`); another abort within
    REARM_STEPS steps moves encoder and decoder to the step kernels; after REARM_STEPS clean steps the cluster kernels are armed
    again; every change forgets the captured graph regions of recognizer, trainer, encoder and generator."""
    from emu import emu_lib
    lib = emu_lib()
    params = synthetic.make_params(CFG, seed=21)
    rec = SpeechRecognizer(device="cpu", params=params, lib=lib, net_config=CFG)
    tr = Trainer(rec, distributed=False, **RULES)
    tr.REARM_STEPS = 2
    batch = synthetic.make_batch(CFG, 3, 12, 4, seed=5, ragged=True)
    rec.encoder.use_persistent, rec.encoder.persist_auto, rec.generator.use_persistent = True, True, True      # as on the GPU
    old = lib.get_knob("cluster_reserve")
    try:
        lib.set_knob("cluster_reserve", 0)
        for owner in (rec, rec.encoder, rec.generator):
            owner._regions = {"k": dict(seen=5)}
        a = tr.recover()
        assert a["action"] == "cluster_reserve" and lib.get_knob("cluster_reserve") == Trainer.RECOVER_RESERVE
        assert rec.encoder.use_persistent and rec.generator.use_persistent and tr._fallback is None
        assert not rec._regions and not rec.encoder._regions and not rec.generator._regions
        a = tr.recover()                                                            # second abort right away
        assert a["action"] == "step_kernels" and tr.aborts == 2
        assert rec.encoder.use_persistent is False and rec.encoder.persist_auto is False and rec.generator.use_persistent is False
        saved = tr._fallback["saved"]
        assert saved[0] is True and saved[1] is True and saved[2] is True
        rec.encoder.use_persistent = rec.generator.use_persistent = False
        tr._fallback["saved"] = (False, True, False, saved[3])                       # what the emulator can run when re-armed
        for k in range(2):
            tr.train_step(batch)
            assert tr._fallback is not None
        tr.train_step(batch)                                                        # the third step arms the cluster kernels again
        assert tr._fallback is None and rec.encoder.persist_auto is True
        assert lib.get_knob("cluster_reserve") == Trainer.RECOVER_RESERVE and tr._reserve_before == 0
        for k in range(3):                                                          # 2 x REARM_STEPS clean steps behind the last abort:
            tr.train_step(batch)                                                    # the reserved CUs are given back
        assert lib.get_knob("cluster_reserve") == 0 and tr._reserve_before is None
    finally:
        lib.set_knob("cluster_reserve", old)
