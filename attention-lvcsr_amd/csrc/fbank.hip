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

}  // extern "C"
