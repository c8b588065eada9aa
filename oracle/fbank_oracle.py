"""CPU ORACLE of the filterbank front end.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

PARITY: PINNED TO AN INDEPENDENT KALDI-COMPATIBLE IMPLEMENTATION, NOT TO KALDI ITSELF (round 5).  The 40 log-mel columns agree to 9.4e-8
with HuggingFace `transformers.audio_utils` — mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True), the Povey window and
spectrogram(preemphasis, remove_dc_offset, snip-edges framing, FLT_EPSILON floor): the numpy path HF's feature extractors use without
torchaudio and that HF holds to torchaudio.compliance.kaldi.fbank, the PyTorch port of Kaldi's feature-fbank.cc
(tests/golden/fbank_hf_kaldi.npz, oracle/gen_fbank_hf_golden.py, tests/test_fbank.py; window 3e-16, mel weights 6e-15).  Kaldi's
binary, torchaudio and librosa are not in the image, so no Kaldi-PRODUCED vector exists; the energy column and add-deltas have no
counterpart in that library and rest on Kaldi's documented formulas.  The reference computes features with Kaldi (`compute-fbank-feats --use-energy=true
--num-mel-bins=40 | add-deltas` + global CMVN, exp/wsj/write_hdf_dataset.sh:94-104, exp/timit/write_hdf_dataset.sh:45-55);
Kaldi's source is not under /root/reference and no version is pinned, and no reference test touches features.  This is
a float64 numpy restatement of Kaldi's published algorithm and defaults (feature-window / mel-computations / feature-fbank /
feature-functions descriptions): snip-edges framing, DC removal, raw log energy, pre-emphasis 0.97, Povey window,
zero-padding to 512, power spectrum, triangular filters equally spaced on mel(f) = 1127 ln(1 + f/700) between 20 Hz and
Nyquist, log with floor FLT_EPSILON; deltas = regression over +-2 frames applied once / twice with edge replication.
It checks the HIP kernel against an independent formulation (numpy rfft vs the kernel's table DFT); it does not prove
Kaldi parity.

Kaldi options ASSUMED (the recipe passes only --use-energy=true --num-mel-bins=40, so everything else is Kaldi's documented
default — except dither): --sample-frequency=16000 --frame-length=25 --frame-shift=10 --snip-edges=true
--remove-dc-offset=true --preemphasis-coefficient=0.97 --window-type=povey --round-to-power-of-two=true (512-point FFT)
--raw-energy=true (log energy BEFORE pre-emphasis and windowing) --energy-floor=0 (floored at FLT_EPSILON instead of -inf)
--low-freq=20 --high-freq=0 (Nyquist) --use-log-fbank=true --use-power=true --htk-compat=false (energy is column 0) --vtln-warp=1;
**--dither=0** (Kaldi's default 1.0 adds Gaussian noise per sample: not reproducible, switched off here, SURVEY.md A0');
add-deltas: --delta-order=2 --delta-window=2 with edge replication.  Pinned without Kaldi: a frozen vector of this oracle
(tests/golden/fbank_frozen.npz, oracle/gen_fbank_frozen.py), a naive O(N^2) third formulation and closed forms
(tests/test_fbank.py).
"""
import numpy

EPS = float(numpy.finfo(numpy.float32).eps)


def povey_window(n):
    a = 2 * numpy.pi / (n - 1)
    return numpy.power(0.5 - 0.5 * numpy.cos(a * numpy.arange(n)), 0.85)


def mel(f):
    return 1127.0 * numpy.log(1.0 + f / 700.0)


def mel_weights(num_mel=40, nfft=512, sample_rate=16000.0, low=20.0, high=0.0):
    nyq = 0.5 * sample_rate
    high = nyq + high if high <= 0 else high
    nbins = nfft // 2
    width = sample_rate / nfft
    lo, hi = mel(low), mel(high)
    delta = (hi - lo) / (num_mel + 1)
    w = numpy.zeros((num_mel, nbins))
    for b in range(num_mel):
        left, center, right = lo + b * delta, lo + (b + 1) * delta, lo + (b + 2) * delta
        for i in range(nbins):
            m = mel(width * i)
            if left < m < right:
                w[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return w


def fbank(wav, frame_length=400, frame_shift=160, num_mel=40, use_energy=True, preemph=0.97, remove_dc=True,
          sample_rate=16000.0):
    wav = numpy.asarray(wav, dtype=numpy.float64)
    if len(wav) < frame_length:
        return numpy.zeros((0, num_mel + int(use_energy)))
    nf = 1 + (len(wav) - frame_length) // frame_shift
    win = povey_window(frame_length)
    W = mel_weights(num_mel, 512, sample_rate)
    out = numpy.zeros((nf, num_mel + int(use_energy)))
    for f in range(nf):
        x = wav[f * frame_shift: f * frame_shift + frame_length].copy()
        if remove_dc:
            x -= x.mean()
        energy = numpy.log(max((x * x).sum(), EPS))
        y = x.copy()
        y[1:] -= preemph * x[:-1]
        y[0] -= preemph * x[0]
        y *= win
        spec = numpy.fft.rfft(y, 512)
        power = (spec.real ** 2 + spec.imag ** 2)[:256]
        m = numpy.log(numpy.maximum(W @ power, EPS))
        if use_energy:
            out[f, 0] = energy
            out[f, 1:] = m
        else:
            out[f] = m
    return out


def add_deltas(feats, window=2):
    feats = numpy.asarray(feats, dtype=numpy.float64)
    T = feats.shape[0]
    base = numpy.arange(-window, window + 1, dtype=numpy.float64)
    base /= (base ** 2).sum()
    scales = [numpy.array([1.0]), base, numpy.convolve(base, base)]
    outs = []
    for sc in scales:
        half = (len(sc) - 1) // 2
        acc = numpy.zeros_like(feats)
        for j in range(-half, half + 1):
            idx = numpy.clip(numpy.arange(T) + j, 0, T - 1)
            acc += sc[j + half] * feats[idx]
        outs.append(acc)
    return numpy.concatenate(outs, axis=1)
