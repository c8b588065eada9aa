"""Feature front end on the device: log mel-filterbank (+energy), deltas, global CMVN — the offline Kaldi step of the
reference's recipe (exp/wsj/write_hdf_dataset.sh:94-104) as HIP kernels (csrc/fbank.hip).  The log-mel columns are pinned to an
independent Kaldi-compatible implementation, not to Kaldi's own binary."""
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
        # the filters as work items of the batched kernel (lvsr_fbank_batch): (filter, chunk of 16 bins starting at a multiple of 4)
        CH = 16
        bins, first, weights = [], [0], []
        for b in range(self.num_mel):
            nz = numpy.nonzero(W[b])[0]
            if len(nz):
                s4 = int(nz[0]) // 4 * 4
                for b0 in range(s4, int(nz[-1]) + 1, CH):
                    w = numpy.zeros(CH, numpy.float32)
                    hi = min(256, b0 + CH)
                    w[: hi - b0] = W[b, b0: hi]
                    bins.append(b0)
                    weights.append(w)
            first.append(len(bins))
        self.batchable = self.num_mel <= 64 and 0 < len(bins) <= 64
        self.mel_items = len(bins)
        self.mel_item_bin = torch.tensor(numpy.asarray(bins or [0], numpy.int32), device=self.device)
        self.mel_item_first = torch.tensor(numpy.asarray(first, numpy.int32), device=self.device)
        self.mel_item_w = t(numpy.stack(weights) if weights else numpy.zeros((1, CH), numpy.float32))

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

    def batch(self, wavs):
        """A set of utterances in ONE launch (lvsr_fbank_batch: a 512-point FFT per frame, one wave per frame).  wavs: list of 1-D
        int16 arrays / tensors -> (feats (total_frames, num_mel + use_energy) device tensor, frame_off (n + 1) int32 device tensor);
        utterance u = rows frame_off[u]:frame_off[u+1]."""
        if not self.batchable:
            raise ValueError("the batched front end holds the filters as at most 64 (filter, 16-bin chunk) items: use __call__ per utterance")
        host = [numpy.asarray(w.cpu() if torch.is_tensor(w) else w, dtype=numpy.int16).ravel() for w in wavs]
        lens = numpy.array([len(w) for w in host], dtype=numpy.int64)
        nfr = numpy.array([self.num_frames(int(n)) for n in lens], dtype=numpy.int64)
        wav_off = numpy.concatenate([[0], numpy.cumsum(lens)]).astype(numpy.int64)
        frame_off = numpy.concatenate([[0], numpy.cumsum(nfr)]).astype(numpy.int32)
        wav = torch.from_numpy(numpy.concatenate(host) if host else numpy.zeros(0, numpy.int16)).to(self.device)
        return self.batch_resident(wav, torch.from_numpy(wav_off).to(self.device), torch.from_numpy(frame_off).to(self.device), len(host),
                                   int(frame_off[-1]))

    def batch_resident(self, wav, wav_off, frame_off, n, total_frames):
        """The same with everything already on the device (wav int16 back to back, wav_off int64 / frame_off int32 of n + 1 entries)."""
        out = torch.empty(total_frames, self.num_mel + int(self.use_energy), dtype=torch.float32, device=self.device)
        if total_frames:
            self.lib.call("lvsr_fbank_batch", self.lib.stream_for(out), ptr(wav), ptr(wav_off), ptr(frame_off), int(n), int(total_frames),
                          ctypes.byref(self.cfg), ptr(self.window), ptr(self.mel_item_bin), ptr(self.mel_item_first), ptr(self.mel_item_w), int(self.mel_items),
                          ptr(self.twiddle), ptr(out))
        return out, frame_off

    def add_deltas_cmvn_batch(self, feats, frame_off, mean=None, std=None):
        """(total_frames, dim) of a set of utterances -> (total_frames, 3*dim); the deltas' edge frames are replicated per utterance."""
        T, dim = int(feats.shape[0]), int(feats.shape[1])
        out = torch.empty(T, 3 * dim, dtype=torch.float32, device=self.device)
        m = None if mean is None else torch.as_tensor(mean, dtype=torch.float32, device=self.device).contiguous()
        i = None if std is None else (1.0 / torch.as_tensor(std, dtype=torch.float32, device=self.device)).contiguous()
        if T:
            self.lib.call("lvsr_add_deltas_cmvn_batch", self.lib.stream_for(out), ptr(feats.contiguous()), ptr(frame_off), int(frame_off.numel()) - 1,
                          T, dim, ptr(m), ptr(i), ptr(out))
        return out

    def add_deltas_cmvn(self, feats, mean=None, std=None):
        """(T,dim) -> (T,3*dim) [static|delta|delta-delta], then (x-mean)/std when global statistics are given."""
        T, dim = int(feats.shape[0]), int(feats.shape[1])
        out = torch.empty(T, 3 * dim, dtype=torch.float32, device=self.device)
        m = None if mean is None else torch.as_tensor(mean, dtype=torch.float32, device=self.device).contiguous()
        i = None if std is None else (1.0 / torch.as_tensor(std, dtype=torch.float32, device=self.device)).contiguous()
        self.lib.call("lvsr_add_deltas_cmvn", self.lib.stream_for(out), ptr(feats.contiguous()), T, dim, ptr(m), ptr(i), ptr(out))
        return out
