// Log mel-filterbank front end (A0' of SURVEY.md §8a): the arithmetic of Kaldi's `compute-fbank-feats
// --use-energy=true --num-mel-bins=40` followed by `add-deltas` and global CMVN, which the reference runs offline
// (exp/wsj/write_hdf_dataset.sh:94-104).  Kaldi is not part of the reference tree: the algorithm below restates
// Kaldi's documented defaults (25 ms / 10 ms frames, snip-edges, DC removal, raw log-energy, pre-emphasis 0.97,
// Povey window, 512-point power spectrum, triangular mel filters 20 Hz..Nyquist on the mel scale 1127 ln(1+f/700),
// dither OFF for determinism).  PARITY UNPINNED — validated against oracle/fbank_oracle.py only.
//
// One work-group per frame: samples -> LDS, wave-shuffle reductions for mean/energy, power spectrum by direct DFT
// over a 512-entry twiddle table (257 bins x 400 samples per frame: the kernel stays HBM/launch bound, an FFT would
// not change its cost class), mel filters as a dense (nmel x 256) product, log.
#include "common.h"
#include "lvsr_hip.h"
#include <string.h>

#define FB_MAX_FRAME 512
#define FB_NFFT 512

__device__ __forceinline__ float fb_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void fbank_kernel(const short* wav, long long nsamp, lvsr_fbank_cfg c, const float* window,
                                                    const float* melw, const float* twid, float* out, int nframes) {
    __shared__ float x[FB_NFFT];
    __shared__ float cs[FB_NFFT], sn[FB_NFFT];
    __shared__ float pw[FB_NFFT / 2 + 1];
    __shared__ float red[4];
    const int f = blockIdx.x;
    const long long s0 = (long long)f * c.frame_shift;
    for (int n = threadIdx.x; n < FB_NFFT; n += 256) {
        x[n] = (n < c.frame_length && s0 + n < nsamp) ? (float)wav[s0 + n] : 0.f;
        cs[n] = twid[n];
        sn[n] = twid[FB_NFFT + n];
    }
    __syncthreads();
    float s = 0.f;
    for (int n = threadIdx.x; n < c.frame_length; n += 256) s += x[n];
    const float mean = fb_block_sum(s, red) / (float)c.frame_length;
    float e = 0.f;
    for (int n = threadIdx.x; n < c.frame_length; n += 256) {
        const float v = c.remove_dc ? x[n] - mean : x[n];
        e += v * v;
    }
    e = fb_block_sum(e, red);                       // raw energy: after DC removal, before pre-emphasis / windowing
    // pre-emphasis reads the neighbour: compute into registers, then write back
    float y0 = 0.f, y1 = 0.f;
    {
        const int n = threadIdx.x;
        if (n < c.frame_length) {
            const float v = c.remove_dc ? x[n] - mean : x[n];
            const float p = n > 0 ? (c.remove_dc ? x[n - 1] - mean : x[n - 1]) : v;
            y0 = (v - c.preemph * p) * window[n];
        }
        const int n2 = n + 256;
        if (n2 < c.frame_length) {
            const float v = c.remove_dc ? x[n2] - mean : x[n2];
            const float p = c.remove_dc ? x[n2 - 1] - mean : x[n2 - 1];
            y1 = (v - c.preemph * p) * window[n2];
        }
    }
    __syncthreads();
    x[threadIdx.x] = y0;
    x[threadIdx.x + 256] = y1;
    __syncthreads();
    for (int k = threadIdx.x; k <= FB_NFFT / 2; k += 256) {
        float re = 0.f, im = 0.f;
        for (int n = 0; n < c.frame_length; ++n) {
            const int a = (k * n) & (FB_NFFT - 1);
            re += x[n] * cs[a];
            im -= x[n] * sn[a];
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    float* o = out + (size_t)f * (c.num_mel + (c.use_energy ? 1 : 0));
    if (threadIdx.x < c.num_mel) {
        const float* wr = melw + (size_t)threadIdx.x * (FB_NFFT / 2);
        float m = 0.f;
        for (int i = 0; i < FB_NFFT / 2; ++i) m += wr[i] * pw[i];
        o[threadIdx.x + (c.use_energy ? 1 : 0)] = logf(fmaxf(m, 1.1920929e-07f));
    }
    if (c.use_energy && threadIdx.x == 0) o[0] = logf(fmaxf(e, 1.1920929e-07f));
}

// add-deltas (order 2, window 2: 5-tap delta, 9-tap delta-delta on the static features, edge frames replicated)
// then global mean/variance normalisation:  out (T, 3*dim)
__global__ __launch_bounds__(256) void deltas_cmvn_kernel(const float* feats, int T, int dim, const float* mean,
                                                         const float* istd, float* out) {
    const int t = blockIdx.x;
    for (int j = threadIdx.x; j < dim; j += 256) {
        float st = feats[(size_t)t * dim + j], d1 = 0.f, d2 = 0.f;
        // scales: delta = [-2,-1,0,1,2]/10 ; delta-delta = delta (*) delta = [4,4,1,-4,-10,-4,1,4,4]/100
        const float s1[5] = {-0.2f, -0.1f, 0.f, 0.1f, 0.2f};
        const float s2[9] = {0.04f, 0.04f, 0.01f, -0.04f, -0.1f, -0.04f, 0.01f, 0.04f, 0.04f};
#pragma unroll
        for (int k = -2; k <= 2; ++k) d1 += s1[k + 2] * feats[(size_t)min(T - 1, max(0, t + k)) * dim + j];
#pragma unroll
        for (int k = -4; k <= 4; ++k) d2 += s2[k + 4] * feats[(size_t)min(T - 1, max(0, t + k)) * dim + j];
        float* o = out + (size_t)t * 3 * dim;
        const float v[3] = {st, d1, d2};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float r = v[q];
            if (mean) r = (r - mean[q * dim + j]) * istd[q * dim + j];
            o[q * dim + j] = r;
        }
    }
}

// ---- batched front end: 512-point FFT per frame, one wave per frame, many utterances per launch -----------------------------------
// The per-utterance kernel above spends 257 x 400 multiply-adds per frame on a direct DFT and is launched once per utterance
// (0.4 MB of PCM): 53 us per 8-second utterance, 0.09 % of the HBM roofline (round-4 bench).  The front end is integer-in /
// float-out streaming work — 320 bytes of new PCM and 164 bytes of features per frame — so this version (a) takes a whole set
// of utterances per launch (PCM back to back, per-utterance sample / frame offsets on the device), (b) gives each frame to ONE
// wave: samples -> LDS (coalesced 2-byte loads), mean / raw energy by DPP sums, pre-emphasis + window, an in-place radix-2 FFT in
// the wave's own LDS slice (9 stages x 4 butterflies per lane, no work-group barrier), power spectrum, and the mel filters from a
// packed table of their non-zero spans (a triangle covers <= 31 of the 256 bins at 40 filters; the table lives in LDS).
#define FBF_WAVES 4
#define FBF_SPAN 64            // longest mel-filter span the packed table holds

__device__ __forceinline__ int fb_bitrev9(int n) {
    int r = 0;
#pragma unroll
    for (int bit = 0; bit < 9; ++bit) r |= ((n >> bit) & 1) << (8 - bit);
    return r;
}

#define FBF_OFFS 2048          // frame offsets staged in LDS (a binary search through global memory costs ~10 dependent loads per frame)
// utterance of global frame f: frame_off[u] <= f < frame_off[u+1]
__device__ __forceinline__ int fb_find_utt(const int* frame_off, int n, int f) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (frame_off[mid] <= f) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(64 * FBF_WAVES) void fbank_fft_kernel(const short* wav, const long long* wav_off, const int* frame_off, int n_utts,
                                                                   int total_frames, lvsr_fbank_cfg c, const float* window, const int* mel_start,
                                                                   const float* mel_w, const float* twid, float* out) {
    // twiddles per stage, contiguous: entry [half + j] = exp(-2 pi i j / (2 half)) for the stage with butterflies `half` apart (the
    // strided reads of one 512-entry table hit 2 banks in the middle stages)
    __shared__ float cs[FB_NFFT], sn[FB_NFFT], win[FB_NFFT];
    __shared__ float mw[64][FBF_SPAN + 1];
    __shared__ int ms[64];
    __shared__ int offs[FBF_OFFS];
    __shared__ float re_all[FBF_WAVES][FB_NFFT], im_all[FBF_WAVES][FB_NFFT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int x = tid; x < FB_NFFT; x += 64 * FBF_WAVES) {
        win[x] = x < c.frame_length ? window[x] : 0.f;
        if (x >= 1) {
            int half = 1;
            while (2 * half <= x) half *= 2;
            const int j = x - half;                                        // x = half + j
            const int idx = j * (FB_NFFT / (2 * half));
            cs[x] = twid[idx];
            sn[x] = twid[FB_NFFT + idx];
        } else { cs[0] = 1.f; sn[0] = 0.f; }
    }
    const bool offs_lds = n_utts + 1 <= FBF_OFFS;
    if (offs_lds)
        for (int x = tid; x <= n_utts; x += 64 * FBF_WAVES) offs[x] = frame_off[x];
    const int* const foff = offs_lds ? offs : frame_off;
    for (int x = tid; x < 64 * FBF_SPAN; x += 64 * FBF_WAVES) {
        const int m = x / FBF_SPAN, i = x % FBF_SPAN;
        mw[m][i] = m < c.num_mel ? mel_w[(size_t)m * FBF_SPAN + i] : 0.f;
    }
    if (tid < 64) ms[tid] = tid < c.num_mel ? mel_start[tid] : 0;
    __syncthreads();
    float* const re = re_all[wave];
    float* const im = im_all[wave];
    const int width = c.num_mel + (c.use_energy ? 1 : 0);
    for (int f = blockIdx.x * FBF_WAVES + wave; f < total_frames; f += gridDim.x * FBF_WAVES) {
        const int u = fb_find_utt(foff, n_utts, f);
        const long long s0 = wav_off[u] + (long long)(f - foff[u]) * c.frame_shift;
        // ---- samples, mean, raw energy (after DC removal, before pre-emphasis / windowing)
        float xv[FB_NFFT / 64];
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < FB_NFFT / 64; ++r) {
            const int n = lane + 64 * r;
            xv[r] = n < c.frame_length ? (float)wav[s0 + n] : 0.f;
            sum += xv[r];
        }
        const float mean = c.remove_dc ? wave_sum(sum) / (float)c.frame_length : 0.f;
        float e = 0.f;
#pragma unroll
        for (int r = 0; r < FB_NFFT / 64; ++r) {
            const int n = lane + 64 * r;
            xv[r] = n < c.frame_length ? xv[r] - mean : 0.f;
            e += xv[r] * xv[r];
            re[n] = xv[r];
        }
        e = wave_sum(e);
        __builtin_amdgcn_wave_barrier();
        // ---- pre-emphasis (needs the neighbour: through LDS), window; into bit-reversed order for the in-place transform
        float yv[FB_NFFT / 64];
#pragma unroll
        for (int r = 0; r < FB_NFFT / 64; ++r) {
            const int n = lane + 64 * r;
            const float prev = n > 0 ? re[n - 1] : xv[r];
            yv[r] = n < c.frame_length ? (xv[r] - c.preemph * prev) * win[n] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < FB_NFFT / 64; ++r) {
            const int n = lane + 64 * r, br = fb_bitrev9(n);
            re[br] = yv[r];
            im[br] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- radix-2 decimation in time, 9 stages, 256 butterflies each (4 per lane)
#pragma unroll
        for (int st = 1; st <= 9; ++st) {
            const int half = 1 << (st - 1);
            float a0[4], a1[4], b0[4], b1[4];
            int i0[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bf = lane + 64 * r, j = bf & (half - 1);
                i0[r] = ((bf >> (st - 1)) << st) + j;
                const float wr = cs[half + j], wi = -sn[half + j];
                const float xr = re[i0[r] + half], xi = im[i0[r] + half];
                b0[r] = wr * xr - wi * xi;
                b1[r] = wr * xi + wi * xr;
                a0[r] = re[i0[r]];
                a1[r] = im[i0[r]];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                re[i0[r]] = a0[r] + b0[r]; im[i0[r]] = a1[r] + b1[r];
                re[i0[r] + half] = a0[r] - b0[r]; im[i0[r] + half] = a1[r] - b1[r];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- power spectrum of bins 0..255 (in place: a lane only touches its own bins), mel filters, log
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = lane + 64 * r;
            re[k] = re[k] * re[k] + im[k] * im[k];
        }
        __builtin_amdgcn_wave_barrier();
        float* o = out + (size_t)f * width;
        {
            const int m = lane < c.num_mel ? lane : 0;
            const int st = ms[m];
            float acc = 0.f;
#pragma unroll 8
            for (int i = 0; i < FBF_SPAN; ++i) acc += mw[m][i] * re[min(st + i, FB_NFFT / 2 - 1)];
            if (lane < c.num_mel) o[lane + (c.use_energy ? 1 : 0)] = logf(fmaxf(acc, 1.1920929e-07f));
        }
        if (c.use_energy && lane == 0) o[0] = logf(fmaxf(e, 1.1920929e-07f));
        __builtin_amdgcn_wave_barrier();
    }
}

// add-deltas + CMVN over a set of utterances: edge frames are replicated per UTTERANCE; 4 frames per work-group
__global__ __launch_bounds__(256) void deltas_cmvn_batch_kernel(const float* feats, const int* frame_off, int n_utts, int total_frames, int dim,
                                                               const float* mean, const float* istd, float* out) {
    __shared__ int offs[FBF_OFFS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool offs_lds = n_utts + 1 <= FBF_OFFS;
    if (offs_lds)
        for (int x = threadIdx.x; x <= n_utts; x += 256) offs[x] = frame_off[x];
    __syncthreads();
    const int* const foff = offs_lds ? offs : frame_off;
    for (int t = blockIdx.x * 4 + wave; t < total_frames; t += gridDim.x * 4) {
        const int u = fb_find_utt(foff, n_utts, t);
        const int t0 = foff[u], t1 = foff[u + 1] - 1;
        for (int j = lane; j < dim; j += 64) {
            const float s1[5] = {-0.2f, -0.1f, 0.f, 0.1f, 0.2f};
            const float s2[9] = {0.04f, 0.04f, 0.01f, -0.04f, -0.1f, -0.04f, 0.01f, 0.04f, 0.04f};
            float v9[9];
#pragma unroll
            for (int k = -4; k <= 4; ++k) v9[k + 4] = feats[(size_t)min(t1, max(t0, t + k)) * dim + j];
            float d1 = 0.f, d2 = 0.f;
#pragma unroll
            for (int k = -2; k <= 2; ++k) d1 += s1[k + 2] * v9[k + 4];
#pragma unroll
            for (int k = -4; k <= 4; ++k) d2 += s2[k + 4] * v9[k + 4];
            float* o = out + (size_t)t * 3 * dim;
            const float v[3] = {v9[4], d1, d2};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float r = v[q];
                if (mean) r = (r - mean[q * dim + j]) * istd[q * dim + j];
                o[q * dim + j] = r;
            }
        }
    }
}

extern "C" {

int lvsr_fbank_num_frames(long long nsamp, const lvsr_fbank_cfg* c) {
    if (!c || nsamp < c->frame_length) return 0;
    return 1 + (int)((nsamp - c->frame_length) / c->frame_shift);      // snip-edges
}

int lvsr_fbank(void* stream, const short* wav, long long nsamp, const lvsr_fbank_cfg* cfg, const float* window,
               const float* melw, const float* twiddle, float* out) {
    LVSR_REQUIRE(cfg && wav && window && melw && twiddle && out, "lvsr_fbank: null argument");
    lvsr_fbank_cfg c;
    memcpy(&c, cfg, sizeof(c));
    LVSR_REQUIRE(c.frame_length > 1 && c.frame_length <= FB_MAX_FRAME && c.frame_shift > 0 && c.num_mel > 0 && c.num_mel <= 256,
                 "lvsr_fbank: unsupported framing (frame_length<=512, num_mel<=256)");
    const int nf = lvsr_fbank_num_frames(nsamp, &c);
    if (nf <= 0) return LVSR_OK;
    hipLaunchKernelGGL(fbank_kernel, dim3(nf), dim3(256), 0, (hipStream_t)stream, wav, nsamp, c, window, melw, twiddle, out, nf);
    return lvsr_check_launch("lvsr_fbank");
}

int lvsr_add_deltas_cmvn(void* stream, const float* feats, int T, int dim, const float* mean, const float* istd, float* out) {
    if (T <= 0 || dim <= 0) return LVSR_OK;
    hipLaunchKernelGGL(deltas_cmvn_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, feats, T, dim, mean, istd, out);
    return lvsr_check_launch("lvsr_add_deltas_cmvn");
}

int lvsr_fbank_batch(void* stream, const short* wav, const long long* wav_off, const int* frame_off, int n, int total_frames,
                     const lvsr_fbank_cfg* cfg, const float* window, const int* mel_start, const float* mel_w, const float* twiddle,
                     float* out) {
    LVSR_REQUIRE(cfg && wav && wav_off && frame_off && window && mel_start && mel_w && twiddle && out && n > 0, "lvsr_fbank_batch: null argument");
    lvsr_fbank_cfg c;
    memcpy(&c, cfg, sizeof(c));
    LVSR_REQUIRE(c.frame_length > 1 && c.frame_length <= FB_MAX_FRAME && c.frame_shift > 0 && c.num_mel > 0 && c.num_mel <= 64,
                 "lvsr_fbank_batch: unsupported framing (frame_length <= 512, num_mel <= 64; filter spans <= 64 bins: use lvsr_fbank otherwise)");
    if (total_frames <= 0) return LVSR_OK;
    int nb = (total_frames + FBF_WAVES - 1) / FBF_WAVES;
    if (nb > 2048) nb = 2048;          // grid-stride over the frames: the tables are staged once per work-group
    hipLaunchKernelGGL(fbank_fft_kernel, dim3(nb), dim3(64 * FBF_WAVES), 0, (hipStream_t)stream, wav, wav_off, frame_off, n, total_frames, c,
                       window, mel_start, mel_w, twiddle, out);
    return lvsr_check_launch("lvsr_fbank_batch");
}

int lvsr_add_deltas_cmvn_batch(void* stream, const float* feats, const int* frame_off, int n, int total_frames, int dim, const float* mean,
                               const float* istd, float* out) {
    LVSR_REQUIRE(feats && frame_off && out && n > 0 && dim > 0, "lvsr_add_deltas_cmvn_batch: bad arguments");
    if (total_frames <= 0) return LVSR_OK;
    int nb = (total_frames + 3) / 4;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(deltas_cmvn_batch_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, feats, frame_off, n, total_frames, dim, mean, istd, out);
    return lvsr_check_launch("lvsr_add_deltas_cmvn_batch");
}

}  // extern "C"
