// Shared definitions of the attention-decoder kernels (decoder_fwd.hip / decoder_bwd.hip).
#pragma once
#include "common.h"
#include "graph_cache.h"
#include "lvsr_hip.h"

typedef lvsr_attdec_args AttDec;

#define ATT_MAX_T 4096       // attended length held in LDS (alignment / energy rows)
#define ATT_MAX_M 1024       // match dim held in LDS
#define ATT_MAX_KM 8192      // conv_num_filters * match dim (handler matrix in LDS)
#define ATT_MAX_FW 1024      // conv filter width 2c+1
#define ATT_TB 16            // attended positions per work-group in the q kernel (4 per wave)
#define ATT_MS 32            // match-dim slice per work-group in the energy kernels
#define ATT_TT 64            // attended positions per tile in the energy kernels (8 t-groups x 8)
#define ATT_KMAX 16          // conv filters held in registers
#define ATT_MAX_KT 16384     // conv_num_filters * attended length held in LDS (energy kernels)
#define ATT_MAX_KF 8192      // conv_num_filters * filter width held in LDS

// compile-time filter-loop bound used for K conv filters: the smallest instantiated value >= K
inline int att_kc(int K) {
    const int inst[7] = {0, 1, 2, 4, 8, 10, 16};
    for (int i = 0; i < 7; ++i)
        if (K <= inst[i]) return inst[i];
    return 16;
}

struct Win { int begin, end; };
int attdec_check(const AttDec& a, const char* what);

// Row groups (lvsr_attdec_args.group_rows: batched beam search): group of row b, the context column a row reads, its attended length
__device__ __forceinline__ int attdec_group(const AttDec& a, int b) { return a.group_rows > 0 ? b / a.group_rows : 0; }
__device__ __forceinline__ int attdec_ctx(const AttDec& a, int b) { return a.group_rows > 0 ? b / a.group_rows : b; }

// lvsr_attdec_args.skip: the attention part of row b's group has been filled in by the caller
__device__ __forceinline__ bool attdec_skip(const AttDec& a, int b) {
    return a.skip != nullptr && a.skip[(size_t)attdec_group(a, b) * a.skip_stride] != 0;
}

// Window of take_glimpses (lvsr/bricks/attention.py:123-161) for the rows [b0, b0 + nb) that form one batch of the reference (all
// rows; with row groups the rows of a group), Tp = the attended length of that batch, stepw = its position counter.
// Content-only attention: whole sequence.
__device__ __forceinline__ Win attdec_window_of(const AttDec& a, int i, int b0, int nb, int Tp, const int* stepw) {
    Win w;
    w.begin = 0; w.end = Tp;
    if (a.K == 0) return w;
    if (a.prior_type == 0) {
        // int64 step * floatX constant -> float64 arithmetic on the f32-rounded speeds (:127-132,160-161)
        const double step = (double)(a.step0 + i + (stepw ? *stepw : 0));
        double bg = a.p0 + step * a.p2, en = a.p1 + step * a.p3;
        // clamp(., 0, Tp - 1) / clamp(., 0, Tp) commute with floor / ceil (integer bounds): the upper clamps are done on the integers,
        // so no (double)Tp lives in a register pair across the label loop of the persistent kernels (it was spilled there)
        bg = fmax(0.0, fmin(2.0e9, bg));
        en = fmax(0.0, fmin(2.0e9, en));
        w.begin = min((int)floor(bg), Tp - 1); w.end = min((int)ceil(en), Tp);
        return w;
    }
    const float before = (float)a.p0, after = (float)a.p1;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int b = b0; b < b0 + nb; ++b) {
        const float p = a.pos[(size_t)i * a.B + b];
        mn = fminf(mn, floorf(p - before));
        mx = fmaxf(mx, ceilf(p + after));
    }
    w.begin = (int)fmaxf(0.f, mn);
    w.end = (int)fminf((float)Tp, mx);
    if (w.end < w.begin) w.end = w.begin;
    return w;
}
__device__ __forceinline__ Win attdec_window(const AttDec& a, int i) { return attdec_window_of(a, i, 0, a.B, a.Tp, a.step_dev); }
// ... as row b sees it
__device__ __forceinline__ Win attdec_window_row(const AttDec& a, int i, int b) {
    if (a.group_rows <= 0) return attdec_window(a, i);
    const int g = b / a.group_rows;
    return attdec_window_of(a, i, g * a.group_rows, a.group_rows, a.group_Tp ? a.group_Tp[g] : a.Tp,
                            a.step_dev ? a.step_dev + (size_t)g * a.step_stride : nullptr);
}

// attended_mask_cut * additional_mask for position t of utterance b (:148-168)
__device__ __forceinline__ float attdec_mask(const AttDec& a, int i, int b, int t) {
    float m = a.Am[(size_t)t * a.Am_ts + (size_t)attdec_ctx(a, b) * a.Am_bs];
    if (a.K > 0 && a.prior_type != 0) {
        const float p = a.pos[(size_t)i * a.B + b];
        const float lo = floorf(p - (float)a.p0), hi = ceilf(p + (float)a.p1);
        m *= ((float)t > lo && (float)t < hi) ? 1.f : 0.f;
    }
    return m;
}

// Window centre of one alignment row (lvsr/bricks/attention.py:133-144).  The order of the float32 additions decides where
// the median crossing lands (cumsum), so the sum stays sequential — but it is run by a whole wave: the row is fetched 64
// positions at a time with one coalesced load, every lane then walks the same chain of adds with v_readlane broadcasts
// (~10 cycles per position instead of one dependent memory access per position: 30 us -> 1 us at T' = 200).
// Call with all 64 lanes of a wave; `w` may be global or LDS; the result is wave-uniform.
__device__ __forceinline__ float attdec_pos_of_row_wave(const AttDec& a, const float* w) {
    const int lane = threadIdx.x & 63, Tp = a.Tp;
    float v = lane < Tp ? w[lane] : 0.f;
    if (a.prior_type == 1) {               // window_around_mean: sum_t alpha[t] * t
        float p = 0.f;
        for (int t0 = 0; t0 < Tp; t0 += 64) {
            const float cur = v;
            if (t0 + 64 < Tp) v = (t0 + 64 + lane < Tp) ? w[t0 + 64 + lane] : 0.f;       // next chunk in flight
#pragma unroll
            for (int l = 0; l < 64; ++l)          // positions beyond T' add 0.0
                p += __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cur), l)) * (float)(t0 + l);
        }
        return p;
    }
    // window_around_median: first crossing of cumsum(alpha) - 0.5 >= 0.  Only the additions are a chain; every lane keeps the
    // running sum of ITS position and the crossing is found with one ballot per 64 positions.
    float c = 0.f, res = 0.f;
    bool found = false, carry = false;     // carry: (c - 0.5 >= 0) at the last position of the previous chunk
    for (int t0 = 0; t0 < Tp; t0 += 64) {
        const float cur = v;
        if (t0 + 64 < Tp) v = (t0 + 64 + lane < Tp) ? w[t0 + 64 + lane] : 0.f;
        float mine = 0.f;
#pragma unroll
        for (int l0 = 0; l0 < 64; l0 += 8) {       // 8 broadcasts ahead of the adds that consume them (64 would need 64 SGPRs)
            float xs[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) xs[l] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(cur), l0 + l));
#pragma unroll
            for (int l = 0; l < 8; ++l) {          // positions beyond T' add 0.0: neither sum nor crossing changes
                c += xs[l];
                mine = lane == l0 + l ? c : mine;
            }
        }
        const bool ge = (mine - 0.5f) >= 0.f;
        const int below = __shfl_up((int)ge, 1, 64);
        const bool prev = lane == 0 ? carry : below != 0;
        const bool cross = ge && !prev && (t0 + lane) > 0;
        const unsigned long long m = __ballot(cross);
        if (m != 0ull) { res = (float)(t0 + (__ffsll((long long)m) - 1) - 1); found = true; }
        if (found) break;                  // (wave-uniform: the rest of the chain cannot move the first crossing)
        carry = (c - 0.5f) >= 0.f;
    }
    return res;
}

__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max(float v, float* red /*[4]*/) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
