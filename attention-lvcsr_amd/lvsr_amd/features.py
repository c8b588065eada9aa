"""Feature front end on the device: log mel-filterbank (+energy), deltas, global CMVN — the offline Kaldi step of the
reference's recipe (exp/wsj/write_hdf_dataset.sh:94-104) as HIP kernels (csrc/fbank.hip).  Parity with Kaldi is
unpinned."""
import ctypes

import numpy
import torch

from .native import ptr


class Fbank(object):
    def __init__(self, device="cuda:0", lib=None, num_mel=40, use_energy=True, sample_rate=16000.0, frame_length_ms=25.0,
                 frame_shift_ms=10.0, preemph=0.97, remove_dc=True, low_freq=20.0, high_freq=0.0):
        from . import native
        self.lib = lib if lib is not None else native.get()
        self.device = torch.device(device)
        self.frame_length = int(round(sample_rate * 1e-3 * frame_length_ms))
        self.frame_shift = int(round(sample_rate * 1e-3 * frame_shift_ms))
        self.num_mel, self.use_energy = int(num_mel), bool(use_energy)
        self.cfg = self.lib.make("lvsr_fbank_cfg", frame_length=self.frame_length, frame_shift=self.frame_shift,
                                 num_mel=self.num_mel, use_energy=int(use_energy), remove_dc=int(remove_dc), preemph=preemph)
        n = numpy.arange(self.frame_length)
        win = numpy.power(0.5 - 0.5 * numpy.cos(2 * numpy.pi / (self.frame_length - 1) * n), 0.85)
        mel = lambda f: 1127.0 * numpy.log(1.0 + f / 700.0)
        nyq = 0.5 * sample_rate
        hi = nyq + high_freq if high_freq <= 0 else high_freq
        lo_m, hi_m = mel(low_freq), mel(hi)
        delta = (hi_m - lo_m) / (self.num_mel + 1)
        m = mel(sample_rate / 512.0 * numpy.arange(256))
        W = numpy.zeros((self.num_mel, 256))
        for b in range(self.num_mel):
            left, center, right = lo_m + b * delta, lo_m + (b + 1) * delta, lo_m + (b + 2) * delta
            up, down = (m - left) / (center - left), (right - m) / (right - center)
            W[b] = numpy.where((m > left) & (m < right), numpy.where(m <= center, up, down), 0.0)
        ang = 2 * numpy.pi * numpy.arange(512) / 512.0
        t = lambda a: torch.tensor(numpy.asarray(a, numpy.float32), device=self.device)
        self.window, self.melw, self.twiddle = t(win), t(W), t(numpy.stack([numpy.cos(ang), numpy.sin(ang)]))

    def num_frames(self, nsamp):
        return int(self.lib._lvsr_fbank_num_frames(int(nsamp), ctypes.byref(self.cfg)))

    def __call__(self, wav_i16):
        """wav_i16: 1-D int16 tensor/ndarray of PCM samples -> (T, num_mel + use_energy) fp32 device tensor."""
        wav = torch.as_tensor(numpy.asarray(wav_i16, dtype=numpy.int16) if not torch.is_tensor(wav_i16) else wav_i16)
        wav = wav.to(self.device).contiguous()
        T = self.num_frames(wav.numel())
        out = torch.empty(T, self.num_mel + int(self.use_energy), dtype=torch.float32, device=self.device)
        self.lib.call("lvsr_fbank", self.lib.stream_for(out), ptr(wav), wav.numel(), ctypes.byref(self.cfg), ptr(self.window),
                      ptr(self.melw), ptr(self.twiddle), ptr(out))
        return out

    def add_deltas_cmvn(self, feats, mean=None, std=None):
        """(T,dim) -> (T,3*dim) [static|delta|delta-delta], then (x-mean)/std when global statistics are given."""
        T, dim = int(feats.shape[0]), int(feats.shape[1])
        out = torch.empty(T, 3 * dim, dtype=torch.float32, device=self.device)
        m = None if mean is None else torch.as_tensor(mean, dtype=torch.float32, device=self.device).contiguous()
        i = None if std is None else (1.0 / torch.as_tensor(std, dtype=torch.float32, device=self.device)).contiguous()
        self.lib.call("lvsr_add_deltas_cmvn", self.lib.stream_for(out), ptr(feats.contiguous()), T, dim, ptr(m), ptr(i), ptr(out))
        return out
