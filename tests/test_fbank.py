"""Filterbank front end: HIP kernels (emulated on CPU, real on the GPU box) vs oracle/fbank_oracle.py.
Parity with Kaldi itself is UNPINNED (no Kaldi in the reference tree) — see the oracle header."""
import numpy
import pytest
from numpy.testing import assert_allclose

from oracle import fbank_oracle as FO
from lvsr_amd.features import Fbank


def _wav(n, seed=0):
    rng = numpy.random.RandomState(seed)
    t = numpy.arange(n) / 16000.0
    x = 3000 * numpy.sin(2 * numpy.pi * 440 * t) + 1500 * numpy.sin(2 * numpy.pi * 2300 * t + 1.0) + rng.normal(0, 300, n) + 120
    return numpy.clip(x, -32768, 32767).astype(numpy.int16)


def run_fbank(device, lib, n):
    wav = _wav(n)
    fb = Fbank(device=device, lib=lib)
    got = fb(wav).cpu().numpy()
    ref = FO.fbank(wav)
    assert got.shape == ref.shape == (1 + (n - 400) // 160, 41)
    assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
    full = fb.add_deltas_cmvn(fb(wav)).cpu().numpy()
    assert_allclose(full, FO.add_deltas(ref), rtol=2e-4, atol=5e-4)
    mean, std = full.mean(0), full.std(0) + 1e-3
    normed = fb.add_deltas_cmvn(fb(wav), mean, std).cpu().numpy()
    assert_allclose(normed, (FO.add_deltas(ref) - mean) / std, rtol=1e-3, atol=2e-3)


def test_oracle_self_consistency():
    w = FO.mel_weights()
    assert w.shape == (40, 256) and (w >= 0).all() and (w.sum(1) > 0).all()
    assert FO.fbank(numpy.zeros(100, numpy.int16)).shape == (0, 41)          # shorter than one frame: no frames
    const = FO.add_deltas(numpy.ones((7, 3)))
    assert_allclose(const[:, 3:], 0, atol=1e-12)                               # deltas of a constant are zero


def test_fbank_emulated():
    from emu import emu_lib
    run_fbank("cpu", emu_lib(), 400 + 160 * 5)


@pytest.mark.gpu
def test_fbank_gpu(gpu_device):
    run_fbank(gpu_device, None, 16000 * 3)
