#!/usr/bin/env python3
"""Generate tests/golden/fbank_hf_kaldi.npz: log-mel filterbank features of a synthetic utterance computed by an INDEPENDENT
Kaldi-compatible implementation — HuggingFace `transformers.audio_utils` (installed in the build image, version recorded in the
fixture).  TEST INFRASTRUCTURE.

Kaldi itself, torchaudio and librosa are not in the image and there is no network; `transformers.audio_utils` carries a numpy
restatement of `compute-fbank-feats` that HuggingFace's feature extractors (SeamlessM4T, AST, ...) use when torchaudio is absent and
that its own test-suite holds to `torchaudio.compliance.kaldi.fbank` (itself the PyTorch port of Kaldi's feature-fbank.cc):
`mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True)`, `window_function("povey", periodic=False)` and
`spectrogram(center=False, preemphasis=0.97, remove_dc_offset=True, power=2.0, log_mel="log", mel_floor=FLT_EPSILON)` — the call
below is SeamlessM4TFeatureExtractor._extract_fbank_features's, with 40 mel bins and the recipe's 20 Hz .. Nyquist range
(exp/wsj/write_hdf_dataset.sh:94-104: compute-fbank-feats --num-mel-bins=40, every other option Kaldi's default, dither off).
Waveform in Kaldi's scale (16-bit integer values as floats).  The energy column (--use-energy=true) has no counterpart there and is
not part of this fixture; add-deltas neither.

    python oracle/gen_fbank_hf_golden.py
"""
import os

import numpy
import transformers
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kaldi_fbank_hf(wav_int16, num_mel=40, sample_rate=16000):
    mel_filters = mel_filter_bank(num_frequency_bins=257, num_mel_filters=num_mel, min_frequency=20, max_frequency=sample_rate // 2,
                                  sampling_rate=sample_rate, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    window = window_function(400, "povey", periodic=False)
    feats = spectrogram(numpy.asarray(wav_int16, numpy.float64), window, frame_length=400, hop_length=160, fft_length=512, power=2.0,
                        center=False, preemphasis=0.97, mel_filters=mel_filters, log_mel="log", mel_floor=1.192092955078125e-07,
                        remove_dc_offset=True, dtype=numpy.float64).T
    return feats, mel_filters, window


def main():
    rng = numpy.random.RandomState(11)
    n = 16000 * 2 + 77
    t = numpy.arange(n) / 16000.0
    # a chirp, two tones, noise, a DC offset and a silent stretch (the log floor)
    wav = (2200 * numpy.sin(2 * numpy.pi * (300 + 1500 * t) * t) + 1300 * numpy.sin(2 * numpy.pi * 2750 * t + 0.3)
           + 600 * numpy.sin(2 * numpy.pi * 6100 * t) + rng.normal(0, 250, n) + 40)
    wav[9000:9800] = 0.0
    wav = numpy.clip(numpy.round(wav), -32768, 32767).astype(numpy.int16)
    feats, mel_filters, window = kaldi_fbank_hf(wav)
    out = os.path.join(REPO, "tests", "golden", "fbank_hf_kaldi.npz")
    numpy.savez_compressed(out, wav=wav, fbank=feats.astype(numpy.float64), mel_filters=mel_filters.astype(numpy.float64),
                           window=numpy.asarray(window, numpy.float64), transformers_version=numpy.array(transformers.__version__))
    print(out, feats.shape, feats[0, :4], feats[57, :4])


if __name__ == "__main__":
    main()
