"""Filterbank front end: HIP kernels (emulated on CPU, real on the GPU box) vs oracle/fbank_oracle.py and vs an independent
Kaldi-compatible implementation (HuggingFace transformers.audio_utils, tests/golden/fbank_hf_kaldi.npz).  Kaldi's own binary is not
in the image: parity with IT stays unproven — see the oracle header."""
import numpy
import pytest
from numpy.testing import assert_allclose

import torch

from conftest import golden_path
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


def run_fbank_batch(device, lib, lengths):
    """The batched front end (FFT per frame, one launch for the set) against the oracle and against the per-utterance kernels:
    utterances of different lengths, one of them shorter than a frame (no frames)."""
    wavs = [_wav(n, seed=i) for i, n in enumerate(lengths)]
    fb = Fbank(device=device, lib=lib)
    feats, off = fb.batch(wavs)
    off_h = off.cpu().numpy()
    full = fb.add_deltas_cmvn_batch(feats, off).cpu().numpy()
    feats = feats.cpu().numpy()
    for u, w in enumerate(wavs):
        ref = FO.fbank(w)
        got = feats[off_h[u]: off_h[u + 1]]
        assert got.shape == ref.shape
        if not len(ref):
            continue
        assert_allclose(got, ref, rtol=2e-4, atol=2e-4)
        assert_allclose(got, fb(w).cpu().numpy(), rtol=2e-4, atol=2e-4)              # the direct-DFT kernel
        assert_allclose(full[off_h[u]: off_h[u + 1]], FO.add_deltas(ref), rtol=2e-4, atol=5e-4)


def test_oracle_self_consistency():
    w = FO.mel_weights()
    assert w.shape == (40, 256) and (w >= 0).all() and (w.sum(1) > 0).all()
    assert FO.fbank(numpy.zeros(100, numpy.int16)).shape == (0, 41)          # shorter than one frame: no frames
    const = FO.add_deltas(numpy.ones((7, 3)))
    assert_allclose(const[:, 3:], 0, atol=1e-12)                               # deltas of a constant are zero


def _direct_fbank_frame(x, num_mel=40):
    """A third, deliberately naive formulation of ONE frame (O(N^2) DFT by its definition, mel filters from their defining
    triangle in mel space sampled per bin) sharing no code with the oracle or the kernel."""
    import math
    x = [float(v) for v in x]
    mean = sum(x) / len(x)
    x = [v - mean for v in x]
    energy = math.log(max(sum(v * v for v in x), 1.1920929e-07))
    y = [x[0] - 0.97 * x[0]] + [x[i] - 0.97 * x[i - 1] for i in range(1, len(x))]
    y = [v * (0.5 - 0.5 * math.cos(2 * math.pi * i / (len(x) - 1))) ** 0.85 for i, v in enumerate(y)] + [0.0] * (512 - len(x))
    power = []
    for k in range(256):
        re = sum(v * math.cos(2 * math.pi * k * i / 512) for i, v in enumerate(y))
        im = sum(v * math.sin(2 * math.pi * k * i / 512) for i, v in enumerate(y))
        power.append(re * re + im * im)
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    lo, hi = mel(20.0), mel(8000.0)
    step = (hi - lo) / (num_mel + 1)
    out = [energy]
    for b in range(num_mel):
        left, center, right = lo + b * step, lo + (b + 1) * step, lo + (b + 2) * step
        acc = 0.0
        for k in range(256):
            m = mel(16000.0 / 512 * k)
            if left < m < right:
                acc += power[k] * ((m - left) / (center - left) if m <= center else (right - m) / (right - center))
        out.append(math.log(max(acc, 1.1920929e-07)))
    return out


def test_oracle_against_a_frozen_vector_a_naive_formulation_and_closed_forms():
    """What can be pinned without Kaldi: (a) the oracle has not drifted (frozen vector, oracle/gen_fbank_frozen.py);
    (b) a naive O(N^2) restatement of the documented algorithm agrees frame by frame; (c) closed forms: a pure tone peaks in
    the mel bin whose triangle contains it, the raw log-energy of a constant-amplitude tone is log(N A^2 / 2), the delta of
    a linear ramp is its slope away from the edges and the delta-delta is zero."""
    from conftest import golden_path
    z = numpy.load(golden_path("fbank_frozen"))
    f = FO.fbank(z["wav"])
    assert_allclose(f, z["fbank"], rtol=1e-10, atol=1e-10)
    assert_allclose(FO.add_deltas(f), z["full"], rtol=1e-10, atol=1e-10)
    for frame in (0, 5):
        naive = _direct_fbank_frame(z["wav"][frame * 160: frame * 160 + 400])
        assert_allclose(f[frame], naive, rtol=1e-8, atol=1e-8)
    t = numpy.arange(400 + 160 * 3) / 16000.0
    tone = (8000 * numpy.sin(2 * numpy.pi * 1000.0 * t)).astype(numpy.int16)
    ft = FO.fbank(tone)
    mel = FO.mel
    edges = mel(20.0) + (mel(8000.0) - mel(20.0)) / 41 * numpy.arange(42)
    expect = int(numpy.searchsorted(edges, mel(1000.0))) - 1          # triangle b spans edges[b]..edges[b+2], centre edges[b+1]
    assert ft[1, 1:].argmax() in (expect - 1, expect)
    assert_allclose(ft[1, 0], numpy.log(400 * 8000.0 ** 2 / 2), rtol=2e-3)
    ramp = numpy.arange(12, dtype=numpy.float64)[:, None] * numpy.array([[1.0, -2.5]])
    d = FO.add_deltas(ramp)
    assert_allclose(d[2:-2, 2:4], numpy.tile([[1.0, -2.5]], (8, 1)), atol=1e-12)
    assert_allclose(d[4:-4, 4:6], 0.0, atol=1e-12)


# Round 5: an INDEPENDENT Kaldi-compatible implementation as the yardstick for the 40 log-mel columns — HuggingFace
# transformers.audio_utils (mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True), the Povey window, spectrogram with
# preemphasis / remove_dc_offset / snip-edges framing / FLT_EPSILON floor): the numpy path HF's feature extractors fall back to without
# torchaudio and that HF holds to torchaudio.compliance.kaldi.fbank — the PyTorch port of Kaldi's feature-fbank.cc
# (tests/golden/fbank_hf_kaldi.npz, oracle/gen_fbank_hf_golden.py; Kaldi, torchaudio and librosa are not in the image).
def test_oracle_matches_the_independent_kaldi_compatible_implementation():
    z = numpy.load(golden_path("fbank_hf_kaldi"))
    f = FO.fbank(z["wav"])
    assert f.shape == (z["fbank"].shape[0], 41)
    assert_allclose(f[:, 1:], z["fbank"], rtol=0, atol=1e-6)                  # observed: 9.4e-8 in the log-mel domain
    assert_allclose(FO.povey_window(400), z["window"], rtol=0, atol=1e-12)
    assert_allclose(FO.mel_weights().T, z["mel_filters"][:256], rtol=0, atol=1e-12)
    assert numpy.abs(z["mel_filters"][256]).max() == 0.0                    # the Nyquist bin Kaldi drops carries no weight anyway
    # the energy column (--use-energy=true --raw-energy=true) has no counterpart there: log sum of squares of the DC-removed frame
    frame = z["wav"][160 * 5: 160 * 5 + 400].astype(numpy.float64)
    assert_allclose(f[5, 0], numpy.log(((frame - frame.mean()) ** 2).sum()), rtol=1e-12)


def run_fbank_vs_hf(device, lib, nsamples=None):
    """The HIP kernels (per-utterance direct DFT and batched FFT) against the independent implementation's features."""
    z = numpy.load(golden_path("fbank_hf_kaldi"))
    wav = z["wav"] if nsamples is None else z["wav"][:nsamples]
    ref = z["fbank"][: 1 + (len(wav) - 400) // 160]
    fb = Fbank(device=device, lib=lib)
    one = fb(torch.from_numpy(wav).to(device)).cpu().numpy()
    assert_allclose(one[:, 1:], ref, rtol=2e-4, atol=2e-4)
    feats, off = fb.batch([wav, wav[: 400 + 160 * 3]])
    feats = feats.cpu().numpy()
    assert_allclose(feats[: len(ref), 1:], ref, rtol=2e-4, atol=2e-4)
    assert_allclose(feats[len(ref): len(ref) + 4, 1:], ref[:4], rtol=2e-4, atol=2e-4)


def test_fbank_vs_independent_implementation_emulated():
    from emu import emu_lib
    run_fbank_vs_hf("cpu", emu_lib(), nsamples=400 + 160 * 14)


@pytest.mark.gpu
def test_fbank_vs_independent_implementation_gpu(gpu_device):
    run_fbank_vs_hf(gpu_device, None)


def run_fbank_other_frame_lengths(device, lib):
    """Frame lengths other than the recipe's 400 samples: the batched kernel's generic instantiation (`KFULL = -1`: the 64-sample group
    the frame ends in is found at run time) — 20-ms frames (320 samples: the frame ends exactly on a group boundary), 30-ms frames
    (480) and a full 512-sample frame, batched == per-utterance kernel == oracle."""
    for ms in (20.0, 30.0, 32.0):
        fb = Fbank(device=device, lib=lib, frame_length_ms=ms)
        n = fb.frame_length
        wavs = [_wav(n + 160 * 4 + 13, seed=3), _wav(n + 160 * 2, seed=4)]
        feats, off = fb.batch(wavs)
        feats, off = feats.cpu().numpy(), off.cpu().numpy()
        for u, w in enumerate(wavs):
            ref = FO.fbank(w, frame_length=n)
            assert_allclose(feats[off[u]: off[u + 1]], ref, rtol=2e-4, atol=2e-4)
            assert_allclose(fb(w).cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


def run_fbank_few_wide_filters(device, lib):
    """Filter banks of few, wide filters: at 16 kHz a filter of an 8-, 12- or 16-filter bank spans 8 / 6 / 5 of the batched kernel's
    16-bin chunks (the recipe's 40 filters: at most 2; 23: 3) — the chunk sums of a filter beyond the fourth (round-5 advisor
    finding: they were dropped).  Batched == per-utterance kernel == oracle; 64 filters are more than 64 chunk items: refused."""
    wavs = [_wav(400 + 160 * 4 + 13, seed=5), _wav(400 + 160 * 3, seed=6)]
    for num_mel in (8, 12, 16, 23):
        fb = Fbank(device=device, lib=lib, num_mel=num_mel)
        assert fb.batchable
        feats, off = fb.batch(wavs)
        feats, off = feats.cpu().numpy(), off.cpu().numpy()
        for u, w in enumerate(wavs):
            ref = FO.fbank(w, num_mel=num_mel)
            assert_allclose(feats[off[u]: off[u + 1]], ref, rtol=2e-4, atol=2e-4, err_msg="num_mel %d" % num_mel)
            assert_allclose(fb(w).cpu().numpy(), ref, rtol=2e-4, atol=2e-4)
    fb = Fbank(device=device, lib=lib, num_mel=64)
    assert not fb.batchable
    with pytest.raises(ValueError):
        fb.batch(wavs)
    assert_allclose(fb(wavs[0]).cpu().numpy(), FO.fbank(wavs[0], num_mel=64), rtol=2e-4, atol=2e-4)


def test_fbank_few_wide_filters_emulated():
    from emu import emu_lib
    run_fbank_few_wide_filters("cpu", emu_lib())


@pytest.mark.gpu
def test_fbank_few_wide_filters_gpu(gpu_device):
    run_fbank_few_wide_filters(gpu_device, None)


def test_fbank_other_frame_lengths_emulated():
    from emu import emu_lib
    run_fbank_other_frame_lengths("cpu", emu_lib())


@pytest.mark.gpu
def test_fbank_other_frame_lengths_gpu(gpu_device):
    run_fbank_other_frame_lengths(gpu_device, None)


def test_fbank_emulated():
    from emu import emu_lib
    run_fbank("cpu", emu_lib(), 400 + 160 * 5)


def test_fbank_batch_emulated():
    from emu import emu_lib
    run_fbank_batch("cpu", emu_lib(), [400 + 160 * 3, 100, 400, 400 + 160 * 6 + 77])
    run_fbank_batch("cpu", emu_lib(), [400 + 160 * 3, 100, 400, 400 + 160 * 6 + 77, 400 + 160 * 2])     # an odd number of frames: the last pair is half empty


@pytest.mark.gpu
def test_fbank_gpu(gpu_device):
    run_fbank(gpu_device, None, 16000 * 3)


@pytest.mark.gpu
def test_fbank_batch_gpu(gpu_device):
    run_fbank_batch(gpu_device, None, [16000 * 3, 100, 16000 * 2 + 123, 400, 16000 * 5 + 7] + [16000 + 37 * i for i in range(27)])
