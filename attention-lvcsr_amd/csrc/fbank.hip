// Log mel-filterbank front end (A0' of SURVEY.md §8a): the arithmetic of Kaldi's `compute-fbank-feats
// --use-energy=true --num-mel-bins=40` followed by `add-deltas` and global CMVN, which the reference runs offline
// (exp/wsj/write_hdf_dataset.sh:94-104).  Kaldi is not part of the reference tree: the algorithm below restates
// Kaldi's documented defaults (25 ms / 10 ms frames, snip-edges, DC removal, raw log-energy, pre-emphasis 0.97,
// Povey window, 512-point power spectrum, triangular mel filters 20 Hz..Nyquist on the mel scale 1127 ln(1+f/700),
// dither OFF for determinism).  Validated against oracle/fbank_oracle.py and an independent Kaldi-compatible implementation
// (tests/golden/fbank_hf_kaldi.npz); no Kaldi-produced vector exists in the build image.
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
// wave: lane L loads samples n = L + 64 k (coalesced 2-byte loads), mean / raw energy by wave sums, pre-emphasis (neighbour through
// a lane shift) and window in registers, then a 512-point FFT as THREE radix-8 passes in registers with two exchanges through the
// wave's own LDS slice (the first version ran nine radix-2 stages through LDS: LDS-bandwidth bound, 130 KB of LDS traffic per frame):
//   positions idx = 64 h + 8 m + lo of the bit-reversed order; lane L's samples sit at idx = 8 brev6(L) + brev3(k): stages 1-3 are
//   local to the loading lane (constant twiddles); pass 2 (stages 4-6) regroups by (lo, h), pass 3 (stages 7-9) by (m, lo) and
//   leaves bin l + 64 m in lane l; the LDS address of a position is swizzled so that all three access patterns are 2-way conflicts
//   at most (fbf_addr);
// then the power spectrum and the mel filters from their non-zero spans, the weights of a lane's filter in REGISTERS.
#define FBF_WAVES 8            // waves per work-group: they share the offset / address / twiddle / weight tables in LDS (10, five waves per SIMD: 344 vs 302 us)
#define FBF_OFFS 1024          // frame / sample offsets staged in LDS (a binary search through global memory costs ~10 dependent loads per frame)

__device__ __forceinline__ int fb_bitrev(int n, int bits) {
    int r = 0;
    for (int bit = 0; bit < bits; ++bit) r |= ((n >> bit) & 1) << (bits - 1 - bit);
    return r;
}
// LDS word of position (h, m, lo): 32 * row + bank with bank bits chosen so that each of the three access patterns — fixed lo over
// (h, m), fixed m over (lo, h), fixed h over (m, lo) — maps its 64 lanes onto all 32 banks (two lanes each); bijective on 0..511
__device__ __forceinline__ int fbf_addr(int h, int m, int lo) {
    const int bank = ((lo ^ m ^ h) & 7) | ((((m & 3) ^ (h >> 1)) & 3) << 3);
    return 32 * (lo + 8 * (m >> 2)) + bank;
}
// utterance of global frame f: frame_off[u] <= f < frame_off[u+1]
__device__ __forceinline__ int fb_find_utt(const int* frame_off, int n, int f) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (frame_off[mid] <= f) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// Complex values as (re, im) pairs on the packed-float32 pipe (v_pk_mul / v_pk_fma / v_pk_add_f32: two float32 operations per
// instruction).  A radix-2 butterfly with twiddle w = (wr, wi):  t = w v = wr (vr, vi) + (-wi, wi) (vi, vr);  v' = u - t;  u' = u + t
// — four instructions: the broadcast of wr / wi, the swap of v's halves and the negation are operand modifiers (op_sel, neg_lo).
typedef float fb_c32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fbf_bfly(fb_c32& u, fb_c32& v, float wr, float wi) {
    const fb_c32 sv = __builtin_shufflevector(v, v, 1, 0);
    const fb_c32 wi2 = {-wi, wi};
    const fb_c32 t = wr * v + wi2 * sv;
    v = u - t;
    u = u + t;
}
__device__ __forceinline__ void fbf_bfly_one(fb_c32& u, fb_c32& v) {          // twiddle 1
    const fb_c32 t = v;
    v = u - t;
    u = u + t;
}
__device__ __forceinline__ void fbf_bfly_mi(fb_c32& u, fb_c32& v) {           // twiddle -i: t = (vi, -vr)
    const fb_c32 t = {v.y, -v.x};
    v = u - t;
    u = u + t;
}
// three radix-2 stages on the 8 values a lane holds (local index = the three index bits the pass works on): stage a pairs (x, x+1)
// with twiddle w1, stage b pairs (x, x+2) with w2[x & 1], stage c pairs (x, x+4) with w4[x & 3]
__device__ __forceinline__ void fbf_radix8(fb_c32 (&z)[8], float w1r, float w1i, const float (&w2r)[2], const float (&w2i)[2],
                                           const float (&w4r)[4], const float (&w4i)[4]) {
#pragma unroll
    for (int x = 0; x < 8; x += 2) fbf_bfly(z[x], z[x + 1], w1r, w1i);
#pragma unroll
    for (int blk = 0; blk < 8; blk += 4)
#pragma unroll
        for (int j = 0; j < 2; ++j) fbf_bfly(z[blk + j], z[blk + j + 2], w2r[j], w2i[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fbf_bfly(z[j], z[j + 4], w4r[j], w4i[j]);
}
// pass 1 of the FFT: the three stages inside a lane's group of 8 positions have the constant twiddles 1 | 1, -i | 1, W8, -i, W8^3
__device__ __forceinline__ void fbf_radix8_first(fb_c32 (&z)[8]) {
    const float R = 0.70710678118654752f;
#pragma unroll
    for (int x = 0; x < 8; x += 2) fbf_bfly_one(z[x], z[x + 1]);
#pragma unroll
    for (int blk = 0; blk < 8; blk += 4) {
        fbf_bfly_one(z[blk], z[blk + 2]);
        fbf_bfly_mi(z[blk + 1], z[blk + 3]);
    }
    fbf_bfly_one(z[0], z[4]);
    fbf_bfly(z[1], z[5], R, -R);
    fbf_bfly_mi(z[2], z[6]);
    fbf_bfly(z[3], z[7], -R, -R);
}

// Round 5: TWO frames per transform, on the packed-float32 pipe.  The samples are real, so one 512-point complex FFT serves a pair of
// frames: z = x_a + i x_b,
//   X_a[k] = (Z[k] + conj Z[N-k]) / 2,   X_b[k] = (Z[k] - conj Z[N-k]) / 2i
// — half the butterflies and half the LDS exchanges per frame; a (re, im) pair IS (frame a, frame b), so windowing, butterflies, power
// spectra and mel sums all run two-wide.  The split needs bin N-k, which sits in another lane: Z goes through the wave's LDS slice in
// plain order once (where the power spectrum went before) and comes back reversed.
// Mel filters as at most 64 work items (filter, 16-bin chunk starting at a multiple of 4 bins), one per lane: 8 x 16-byte LDS reads and
// 16 packed multiply-adds per PAIR and lane (the recipe's 40 filters, spans 3..31 bins, are 53 items; a lane per FILTER had to carry the
// longest span: 32 reads / multiply-adds and 32 weight registers per frame), the chunks of a filter summed by its lane through LDS.
#define FBF_CHUNK 16
// OFFS_LDS: the utterances' frame / sample offsets fit the LDS tables (n_utts < FBF_OFFS): searched there, else in global memory.
// KFULL: frame_length / 64 when known at compile time (6 = the recipe's 400-sample frames), -1 = any: sample groups k < KFULL lie
// wholly inside the frame (no masking), k > KFULL wholly outside (no loads, no arithmetic: they are zeros).
template <bool OFFS_LDS, int KFULL>
__global__ __launch_bounds__(64 * FBF_WAVES) void fbank_fft_kernel(const short* wav, const long long* wav_off, const int* frame_off, int n_utts,
                                                                   int total_frames, lvsr_fbank_cfg c, const float* window, const int* item_bin,
                                                                   const int* item_first, const float* item_w, int n_items, const float* twid,
                                                                   float* out) {
    __shared__ int offs[FBF_OFFS];
    // (the utterances' sample offsets too: from global memory a frame's fetch was a chain of TWO dependent memory latencies — offset,
    // then samples)
    __shared__ long long woffs[FBF_OFFS];
    __shared__ __attribute__((aligned(16))) fb_c32 z_all[FBF_WAVES][FB_NFFT];
    // per-lane tables every wave of the work-group shares (the same for all frames; in registers they cost the third wave per SIMD):
    // the swizzled LDS words of the three exchange patterns (6 x int4 per lane) and the lane's 16 mel weights (4 x float4)
    __shared__ int4 tab_at[6 * 64];
    __shared__ float4 tab_w[4 * 64];
    __shared__ float4 tab_tw[8 * 64];      // twiddles of passes 2 and 3: (w1 | w2[0]), (w2[1] | w4[0]), (w4[1] | w4[2]), (w4[3] | -) as (re, im) pairs
    __shared__ float win_lds[FB_NFFT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (OFFS_LDS)
        for (int x = tid; x <= n_utts; x += 64 * FBF_WAVES) { offs[x] = frame_off[x]; woffs[x] = wav_off[x]; }
    __syncthreads();
    // (two code paths with static address spaces: a pointer selected at run time between LDS and global memory compiles to FLAT
    // loads in the binary search)
    auto frame_start = [&](int f, int& u) -> long long {
        if (OFFS_LDS) { u = fb_find_utt(offs, n_utts, f); return woffs[u] + (long long)(f - offs[u]) * c.frame_shift; }
        u = fb_find_utt(frame_off, n_utts, f);
        return wav_off[u] + (long long)(f - frame_off[u]) * c.frame_shift;
    };
    fb_c32* const zb = z_all[wave];
    auto zat = [&](int byte_off) -> fb_c32& { return *(fb_c32*)((char*)zb + byte_off); };
    // ---- per-lane constants, in registers for every frame of the wave
    // twiddle of stage s (butterflies `half` = 2^(s-1) apart) at offset j: exp(-2 pi i j / (2 half)) = (cos, -sin)(2 pi j (256 / half) / 512)
    auto tw = [&](int half, int j, float& wr, float& wi) {
        const int idx = j * (FB_NFFT / 2 / half);
        wr = twid[idx]; wi = -twid[FB_NFFT + idx];
    };
    const int lo2 = lane & 7, h2 = lane >> 3;          // pass 2: this lane's (lo, h); its 8 values run over m
    const int lo3 = lane & 7, m3 = lane >> 3;          // pass 3: this lane's (m, lo) = bin l + 64 x; its 8 values run over h
    if (wave == 0) {
        float w[16][2];
        tw(8, lo2, w[0][0], w[0][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) tw(16, lo2 + 8 * j, w[1 + j][0], w[1 + j][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) tw(32, lo2 + 8 * j, w[3 + j][0], w[3 + j][1]);
        w[7][0] = w[7][1] = 0.f;
        tw(64, lane, w[8][0], w[8][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) tw(128, lane + 64 * j, w[9 + j][0], w[9 + j][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) tw(256, lane + 64 * j, w[11 + j][0], w[11 + j][1]);
        w[15][0] = w[15][1] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) tab_tw[j * 64 + lane] = make_float4(w[2 * j][0], w[2 * j][1], w[2 * j + 1][0], w[2 * j + 1][1]);
    }
    for (int x = tid; x < FB_NFFT; x += 64 * FBF_WAVES) win_lds[x] = x < c.frame_length ? window[x] : 0.f;
    const int g1 = fb_bitrev(lane, 6), h1 = g1 >> 3, m1 = g1 & 7;      // pass 1: this lane's samples sit at positions (h1, m1, lo = brev3(k))
    if (wave == 0) {
        int at[24];
#pragma unroll
        for (int x = 0; x < 8; ++x) {          // BYTE offsets into the wave's slice (8-byte complex words): no shift per access
            at[x] = 8 * fbf_addr(h1, m1, x); at[8 + x] = 8 * fbf_addr(h2, x, lo2); at[16 + x] = 8 * fbf_addr(x, m3, lo3);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) tab_at[j * 64 + lane] = make_int4(at[4 * j], at[4 * j + 1], at[4 * j + 2], at[4 * j + 3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = lane < n_items ? item_w[(size_t)lane * FBF_CHUNK + 4 * j + e] : 0.f;
            tab_w[j * 64 + lane] = make_float4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    __syncthreads();
    // this lane's mel work item: first bin (a multiple of 4); as a FILTER lane: the range of items to add up
    const int ibin = lane < n_items ? item_bin[lane] : 0;
    const int it0 = lane < c.num_mel ? item_first[lane] : 0, it1 = lane < c.num_mel ? item_first[lane + 1] : 0;
    const int width = c.num_mel + (c.use_energy ? 1 : 0);
    // the samples of a pair are fetched one pair ahead (a wave works on its pairs one after the other).  A frame's first sample is
    // wave-uniform (scalar registers); the loads are unconditional on clamped offsets — lanes beyond the frame, or beyond the end of
    // the PCM buffer, re-read its last sample and are zeroed by `keep` below — so they issue back to back instead of one exec-masked
    // branch each
    short nxa[8], nxb[8];
    const long long wav_last = (OFFS_LDS ? woffs[n_utts] : wav_off[n_utts]) - 1;
    auto fetch = [&](int f, short (&nx)[8]) {
        const int fc = min(f, total_frames - 1);
        int u;
        const long long s0v = frame_start(fc, u);
        const long long s0 = ((long long)__builtin_amdgcn_readfirstlane((int)(s0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)s0v);
        const long long room = wav_last - s0;
        const unsigned lim2 = 2u * (unsigned)(room < (long long)(c.frame_length - 1) ? room : (long long)(c.frame_length - 1));
        const char* base = (const char*)(wav + s0);          // wave-uniform base + one 32-bit byte offset per load (one v_min each)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned want = 2u * (unsigned)(lane + 64 * k);
            nx[k] = (KFULL >= 0 && k > KFULL) ? (short)0 : *(const short*)(base + (want < lim2 ? want : lim2));
        }
    };
    // keep(k): is sample lane + 64 k inside the frame?  Uniform but for the one k the frame ends in
    const int kfull = KFULL >= 0 ? KFULL : c.frame_length >> 6;
    const float keep_edge = lane < (c.frame_length & 63) ? 1.f : 0.f;
    auto keep = [&](int k) -> float { return k < kfull ? 1.f : (k == kfull ? keep_edge : 0.f); };
    auto inside = [&](int k) -> bool { return KFULL >= 0 && k < KFULL; };       // compile-time: the whole group lies inside the frame
    auto outside = [&](int k) -> bool { return KFULL >= 0 && k > KFULL; };      // ... or beyond it
    const float inv_len = 1.f / (float)c.frame_length;
    const int n_pairs = (total_frames + 1) / 2;
    const int p_first = blockIdx.x * FBF_WAVES + wave, p_step = gridDim.x * FBF_WAVES;
    if (p_first < n_pairs && total_frames > 0) { fetch(2 * p_first, nxa); fetch(2 * p_first + 1, nxb); }
    for (int pr = p_first; pr < n_pairs; pr += p_step) {
        const int fa = 2 * pr, fb = fa + 1;
        // ---- samples n = lane + 64 k of both frames as (a, b) pairs: DC removal, raw energy (after DC removal, before pre-emphasis /
        // windowing), pre-emphasis (x[n-1] sits one lane down; lane 0: lane 63 of the previous k), window
        fb_c32 xv[8], z[8];
        fb_c32 sum = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (outside(k)) { xv[k] = (fb_c32){0.f, 0.f}; continue; }
            xv[k] = (fb_c32){(float)nxa[k], (float)nxb[k]};
            if (!inside(k)) xv[k] *= keep(k);
            sum += xv[k];
        }
        // (unconditional: past the last pair the clamped frame index re-reads the last frame — a conditional prefetch costs the
        // loop-carried sample registers a copy each way)
        fetch(2 * (pr + p_step), nxa);
        fetch(2 * (pr + p_step) + 1, nxb);
        fb_c32 mean = {0.f, 0.f};
        if (c.remove_dc) mean = (fb_c32){wave_sum_dpp(sum.x), wave_sum_dpp(sum.y)} * inv_len;
        fb_c32 e2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (outside(k)) continue;
            xv[k] = xv[k] - mean;
            if (!inside(k)) xv[k] *= keep(k);
            e2 += xv[k] * xv[k];
        }
        const float ea = wave_sum_dpp(e2.x), eb = wave_sum_dpp(e2.y);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (outside(k)) { z[((k & 1) << 2) | (k & 2) | ((k >> 2) & 1)] = (fb_c32){0.f, 0.f}; continue; }
            // x[n-1] in two DPP moves per component: wave_ror:1 of the previous k brings ITS lane 63 to lane 0 (the value lane 0
            // needs), wave_shr:1 of this k then fills lanes 1..63 and leaves lane 0 alone (`old` operand) — no v_readlane, no select
            // (fetching x[n-1] with a second set of loads instead was measured slower: 377 vs 354 us)
            auto shifted = [&](float cur, float before) -> float {
                const int wrapv = k > 0 ? __builtin_amdgcn_mov_dpp((int)__float_as_uint(before), 0x13C, 0xf, 0xf, false) : (int)__float_as_uint(cur);
                return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(wrapv, (int)__float_as_uint(cur), 0x138, 0xf, 0xf, false));
            };
            const fb_c32 prev = {shifted(xv[k].x, xv[k > 0 ? k - 1 : 0].x), shifted(xv[k].y, xv[k > 0 ? k - 1 : 0].y)};
            const int lo = ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1);        // brev3(k): position of sample k inside the lane's group
            z[lo] = (xv[k] - c.preemph * prev) * win_lds[lane + 64 * k];      // (the window is zero beyond the frame)
        }
        // ---- pass 1: stages 1-3 inside the lane's group of 8 positions (twiddles 1 | 1, -i | W8^j)
        fbf_radix8_first(z);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int4 a = tab_at[j * 64 + lane];
            zat(a.x) = z[4 * j]; zat(a.y) = z[4 * j + 1]; zat(a.z) = z[4 * j + 2]; zat(a.w) = z[4 * j + 3];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- pass 2: stages 4-6 over m (positions 64 h + 8 m + lo of this lane's (lo, h))
        const int4 a2lo = tab_at[2 * 64 + lane], a2hi = tab_at[3 * 64 + lane];
        z[0] = zat(a2lo.x); z[1] = zat(a2lo.y); z[2] = zat(a2lo.z); z[3] = zat(a2lo.w);
        z[4] = zat(a2hi.x); z[5] = zat(a2hi.y); z[6] = zat(a2hi.z); z[7] = zat(a2hi.w);
        {
            const float4 t0 = tab_tw[0 * 64 + lane], t1 = tab_tw[1 * 64 + lane], t2 = tab_tw[2 * 64 + lane], t3 = tab_tw[3 * 64 + lane];
            const float w2r[2] = {t0.z, t1.x}, w2i[2] = {t0.w, t1.y}, w4r[4] = {t1.z, t2.x, t2.z, t3.x}, w4i[4] = {t1.w, t2.y, t2.w, t3.y};
            fbf_radix8(z, t0.x, t0.y, w2r, w2i, w4r, w4i);
        }
        __builtin_amdgcn_wave_barrier();
        zat(a2lo.x) = z[0]; zat(a2lo.y) = z[1]; zat(a2lo.z) = z[2]; zat(a2lo.w) = z[3];
        zat(a2hi.x) = z[4]; zat(a2hi.y) = z[5]; zat(a2hi.z) = z[6]; zat(a2hi.w) = z[7];
        __builtin_amdgcn_wave_barrier();
        // ---- pass 3: stages 7-9 over h (positions 64 h + lane): bin lane + 64 h ends up in z[h]
        {
            const int4 a3lo = tab_at[4 * 64 + lane], a3hi = tab_at[5 * 64 + lane];
            z[0] = zat(a3lo.x); z[1] = zat(a3lo.y); z[2] = zat(a3lo.z); z[3] = zat(a3lo.w);
            z[4] = zat(a3hi.x); z[5] = zat(a3hi.y); z[6] = zat(a3hi.z); z[7] = zat(a3hi.w);
        }
        {
            const float4 t0 = tab_tw[4 * 64 + lane], t1 = tab_tw[5 * 64 + lane], t2 = tab_tw[6 * 64 + lane], t3 = tab_tw[7 * 64 + lane];
            const float w2r[2] = {t0.z, t1.x}, w2i[2] = {t0.w, t1.y}, w4r[4] = {t1.z, t2.x, t2.z, t3.x}, w4i[4] = {t1.w, t2.y, t2.w, t3.y};
            fbf_radix8(z, t0.x, t0.y, w2r, w2i, w4r, w4i);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- split the pair: Z in plain order through the slice, bin N - k read back
#pragma unroll
        for (int h = 0; h < 8; ++h) zb[lane + 64 * h] = z[h];
        __builtin_amdgcn_wave_barrier();
        fb_c32 pw[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const fb_c32 y = zb[(FB_NFFT - (lane + 64 * h)) & (FB_NFFT - 1)];
            const fb_c32 sgn = {1.f, -1.f};                       // conj(y) = y * (1, -1): one packed multiply-add each for sum and difference
            const fb_c32 s = y * sgn + z[h], d = z[h] - y * sgn;  // 2 X_a[k] = (sr, si);  2 i X_b[k] = (dr, di)
            // (|X_a[k]|^2, |X_b[k]|^2): scalar multiply-adds — the packed form needs the two squares of a value in different pairs
            pw[h] = (fb_c32){0.25f * (s.x * s.x + s.y * s.y), 0.25f * (d.x * d.x + d.y * d.y)};
        }
        __builtin_amdgcn_wave_barrier();
        // ---- power spectra of bins 0..255 in plain order, (frame a, frame b) side by side; mel filters, log
#pragma unroll
        for (int h = 0; h < 4; ++h) zb[lane + 64 * h] = pw[h];
        __builtin_amdgcn_wave_barrier();
        {
            fb_c32 acc = {0.f, 0.f};
            const float4* p4 = (const float4*)(zb + ibin);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 w4 = tab_w[j * 64 + lane];
                const float4 v0 = p4[2 * j], v1 = p4[2 * j + 1];           // bins ibin + 4 j .. + 3 of both frames
                acc += w4.x * (fb_c32){v0.x, v0.y};
                acc += w4.y * (fb_c32){v0.z, v0.w};
                acc += w4.z * (fb_c32){v1.x, v1.y};
                acc += w4.w * (fb_c32){v1.z, v1.w};
            }
            // (words 384..447 of the slice: beyond every chunk — bins < 256 + 16 — and free once the split has read Z)
            zb[384 + lane] = acc;
            __builtin_amdgcn_wave_barrier();
            acc = (fb_c32){0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                   // the recipe's filters span 2 chunks; up to 4 without a branch
                const fb_c32 part = zb[384 + min(it0 + j, 63)];
                acc += it0 + j < it1 ? part : (fb_c32){0.f, 0.f};
            }
            // filter banks of few, wide filters (num_mel 16 at 16 kHz: 5 chunks, 8: 8 chunks): the remaining chunks, in order
            for (int it = it0 + 4; it < it1; ++it) acc += zb[384 + it];
            float* oa = out + (size_t)fa * width;
            if (lane < c.num_mel) oa[lane + (c.use_energy ? 1 : 0)] = logf(fmaxf(acc.x, 1.1920929e-07f));
            if (c.use_energy && lane == 0) oa[0] = logf(fmaxf(ea, 1.1920929e-07f));
            if (fb < total_frames) {
                float* ob = out + (size_t)fb * width;
                if (lane < c.num_mel) ob[lane + (c.use_energy ? 1 : 0)] = logf(fmaxf(acc.y, 1.1920929e-07f));
                if (c.use_energy && lane == 0) ob[0] = logf(fmaxf(eb, 1.1920929e-07f));
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// add-deltas + CMVN over a set of utterances: edge frames are replicated per UTTERANCE.  One thread per (frame, coefficient) in
// flat order — consecutive lanes read consecutive floats of a row and write three coalesced runs (the first version gave a wave to a
// frame: 41 of 64 lanes live, 1.1 TB/s).  Round 5: a work-group takes a TILE of DC_TILE consecutive frames: the rows [tile - 4, tile + DC_TILE + 4) are ONE contiguous span of
// `feats` — staged in LDS with coalesced loads, once (the first version read every row nine times out of L2: 1.65 TB/s) — the tile's
// outputs one contiguous span of `out`.  Edge frames are replicated per utterance: neighbour t + k of frame t is row
// clamp(t + k, first, last frame of t's utterance), which the clamp keeps inside the staged span.
#define DC_TILE 56
#define DC_MAXDIM 64
__global__ __launch_bounds__(256) void deltas_cmvn_batch_kernel(const float* feats, const int* frame_off, int n_utts, int total_frames, int dim,
                                                               const float* mean, const float* istd, float* out) {
    __shared__ float rows[(DC_TILE + 8) * DC_MAXDIM];
    __shared__ int lo_of[DC_TILE], hi_of[DC_TILE];
    __shared__ int u_first;
    // (the utterance bounds come from global memory through L2; staging the offsets in LDS per work-group first was measured slower:
    // 155 vs 141 us)
    const int* const foff = frame_off;
    const int ntiles = (total_frames + DC_TILE - 1) / DC_TILE;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ta = tile * DC_TILE, tb = min(total_frames, ta + DC_TILE);          // frames [ta, tb)
        const int ra = max(0, ta - 4), rb = min(total_frames, tb + 4);                // staged rows [ra, rb)
        __syncthreads();                                                              // (the previous tile's readers are done)
        {
            const float* src = feats + (size_t)ra * dim;
            const int n = (rb - ra) * dim;
            for (int x = threadIdx.x; x < n; x += 256) rows[x] = src[x];
        }
        // the utterance of the tile's first frame: wave 0 probes 64 boundaries per round (two dependent loads for up to 4 096
        // utterances instead of a ten-step binary search per frame), the frames of the tile then walk on from it
        if (threadIdx.x < 64) {
            int lo = 0, hi = n_utts;                 // the answer u — the last utterance with foff[u] <= ta — lies in [lo, hi)
            while (hi - lo > 1) {
                const int step = (hi - lo + 63) / 64, idx = lo + (int)threadIdx.x * step;
                const bool le = idx < hi && foff[idx] <= ta;
                const int c = __popcll(__ballot(le));             // monotone: the first c probes are <= ta (c >= 1: foff[lo] <= ta)
                lo += (c - 1) * step;
                hi = min(hi, lo + step);
            }
            if (threadIdx.x == 0) u_first = lo;
        }
        __syncthreads();
        if (threadIdx.x < tb - ta) {                 // the utterance bounds of the tile's frames
            const int t = ta + threadIdx.x;
            int u = u_first;
            while (u + 1 < n_utts && foff[u + 1] <= t) ++u;
            lo_of[threadIdx.x] = foff[u];
            hi_of[threadIdx.x] = foff[u + 1] - 1;
        }
        __syncthreads();
        const float s1[5] = {-0.2f, -0.1f, 0.f, 0.1f, 0.2f};
        const float s2[9] = {0.04f, 0.04f, 0.01f, -0.04f, -0.1f, -0.04f, 0.01f, 0.04f, 0.04f};
        // thread = (frame slot, coefficient): the divisions by the run-time `dim` happen once per tile, not once per element (the
        // kernel is bound by vector-ALU issue: 152 vector instructions per wave and element before, profiles/r05_pmc_fbank.md)
        const int fpi = 256 / dim;                         // frames per iteration of the work-group (6 at dim = 41)
        const int tl0 = threadIdx.x / dim, j = threadIdx.x - tl0 * dim;
        float mj[3] = {0.f, 0.f, 0.f}, ij[3] = {1.f, 1.f, 1.f};
        if (mean && tl0 < fpi)
#pragma unroll
            for (int q = 0; q < 3; ++q) { mj[q] = mean[q * dim + j]; ij[q] = istd[q * dim + j]; }
        if (tl0 < fpi)
            for (int tl = tl0; tl < tb - ta; tl += fpi) {
                const int t = ta + tl;
                const int t0 = lo_of[tl], t1 = hi_of[tl];
                float v9[9];
#pragma unroll
                for (int k = -4; k <= 4; ++k) v9[k + 4] = rows[(min(t1, max(t0, t + k)) - ra) * dim + j];
                float d1 = 0.f, d2 = 0.f;
#pragma unroll
                for (int k = -2; k <= 2; ++k) d1 += s1[k + 2] * v9[k + 4];
#pragma unroll
                for (int k = -4; k <= 4; ++k) d2 += s2[k + 4] * v9[k + 4];
                float* o = out + (size_t)t * 3 * dim + j;
                o[0] = (v9[4] - mj[0]) * ij[0];
                o[dim] = (d1 - mj[1]) * ij[1];
                o[2 * dim] = (d2 - mj[2]) * ij[2];
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
                     const lvsr_fbank_cfg* cfg, const float* window, const int* item_bin, const int* item_first, const float* item_w,
                     int n_items, const float* twiddle, float* out) {
    LVSR_REQUIRE(cfg && wav && wav_off && frame_off && window && item_bin && item_first && item_w && twiddle && out && n > 0, "lvsr_fbank_batch: null argument");
    LVSR_REQUIRE(n_items > 0 && n_items <= 64, "lvsr_fbank_batch: the mel filters must be at most 64 (filter, 16-bin chunk) items (use lvsr_fbank otherwise)");
    lvsr_fbank_cfg c;
    memcpy(&c, cfg, sizeof(c));
    LVSR_REQUIRE(c.frame_length > 1 && c.frame_length <= FB_MAX_FRAME && c.frame_shift > 0 && c.num_mel > 0 && c.num_mel <= 64,
                 "lvsr_fbank_batch: unsupported framing (frame_length <= 512, num_mel <= 64: use lvsr_fbank otherwise)");
    if (total_frames <= 0) return LVSR_OK;
    int nb = ((total_frames + 1) / 2 + FBF_WAVES - 1) / FBF_WAVES;        // a wave takes a PAIR of frames per transform
    if (nb > 2048) nb = 2048;          // grid-stride over the pairs (capping the grid at the 512 resident work-groups was measured slower: 388 vs 377 us)
    const bool lds = n + 1 <= FBF_OFFS, k6 = (c.frame_length >> 6) == 6;
#define FBF_LAUNCH(L, K) hipLaunchKernelGGL((fbank_fft_kernel<L, K>), dim3(nb), dim3(64 * FBF_WAVES), 0, (hipStream_t)stream, wav, wav_off, \
                                            frame_off, n, total_frames, c, window, item_bin, item_first, item_w, n_items, twiddle, out)
    if (lds && k6) FBF_LAUNCH(true, 6);
    else if (lds) FBF_LAUNCH(true, -1);
    else if (k6) FBF_LAUNCH(false, 6);
    else FBF_LAUNCH(false, -1);
#undef FBF_LAUNCH
    return lvsr_check_launch("lvsr_fbank_batch");
}

int lvsr_add_deltas_cmvn_batch(void* stream, const float* feats, const int* frame_off, int n, int total_frames, int dim, const float* mean,
                               const float* istd, float* out) {
    LVSR_REQUIRE(feats && frame_off && out && n > 0 && dim > 0, "lvsr_add_deltas_cmvn_batch: bad arguments");
    if (total_frames <= 0) return LVSR_OK;
    LVSR_REQUIRE(dim <= DC_MAXDIM, "lvsr_add_deltas_cmvn_batch: at most %d coefficients per frame (use lvsr_add_deltas_cmvn otherwise)", DC_MAXDIM);
    const int ntiles = (total_frames + DC_TILE - 1) / DC_TILE;
    const int nb = ntiles > 8192 ? 8192 : ntiles;
    hipLaunchKernelGGL(deltas_cmvn_batch_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, feats, frame_off, n, total_frames, dim, mean, istd, out);
    return lvsr_check_launch("lvsr_add_deltas_cmvn_batch");
}

}  // extern "C"
