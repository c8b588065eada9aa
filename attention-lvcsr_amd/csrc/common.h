// Shared device helpers for the gfx950 kernels of the attention-LVCSR hot path.
// wave = 64 lanes everywhere; MFMA = v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LVSR_OK 0
#define LVSR_ERR_ARG (-1)
#define LVSR_ERR_HIP (-2)
#define LVSR_ERR_WS (-3)

typedef float f32x4 __attribute__((ext_vector_type(4)));

void lvsr_set_error(const char* fmt, ...);
int lvsr_check_launch(const char* what);
// tuning knobs (include/lvsr_hip.h LVSR_KNOB_*; process-wide, part of every cached graph's key through lvsr_knob_bytes)
int lvsr_knob(int knob);
// Largest grid of a persistent cluster launch with ONE work-group per CU (its work-groups wait for each other, so all of them must
// be resident at once): the device's CU count (256 on MI355X) — what a cooperative launch would check —, minus
// LVSR_KNOB_CLUSTER_RESERVE CUs left to other work when the caller shares the device; LVSR_KNOB_MAX_CLUSTER_WGS overrides it.
// Kernels of which several work-groups fit a CU multiply it by their occupancy (encoder_persist.hip wide_cluster_capacity).
// Safety does not rest on this number: every wait is bounded and raises the abort word.
int lvsr_max_cluster_wgs();

#define LVSR_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            lvsr_set_error(__VA_ARGS__);        \
            return LVSR_ERR_ARG;                \
        }                                       \
    } while (0)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// The same functions on the hardware transcendental units (v_exp_f32, v_rcp_f32: 1 ulp each; ~6 instructions instead of ~25 /
// ~40), for the chains where the activation sits between two cross-CU hand-offs.  Absolute error <= ~2e-7 (float32 rounding
// of the result is 6e-8); saturates correctly (exp -> inf / 0).
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// value of lane (l ^ 1) / (l ^ 2) inside each quad of lanes: one DPP move (quad_perm), no LDS crossbar
__device__ __forceinline__ float lvsr_dpp_quad_xor1(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float lvsr_dpp_quad_xor2(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
}

// lane i <- lane 7-i of its half row (8 lanes) / lane 15-i of its row (16 lanes): after the two quad exchanges every lane of
// a quad holds the quad's sum, so one mirror folds two quads (8 lanes), a second one two half rows (16 lanes) — no LDS crossbar
// x, through an identity lane permutation the compiler does not see through and — being a cross-lane operation — does not
// move out of loops: index arithmetic derived from the result is redone where it is used instead of being hoisted out of a
// long loop and kept (or spilled) across it.  One v_mov_b32_dpp.
__device__ __forceinline__ int lvsr_unhoisted(int x) {
    return __builtin_amdgcn_mov_dpp(x, 0xE4, 0xf, 0xf, true);                                                      // quad_perm:[0,1,2,3]
}
__device__ __forceinline__ float lvsr_dpp_half_mirror(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
}
__device__ __forceinline__ float lvsr_dpp_mirror(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x140, 0xf, 0xf, true));   // row_mirror
}
// v + the value of the lane 16 away inside each half of the wave (lanes 0-31, 32-63): v_permlane16_swap_b32 (gfx950) exchanges the
// odd rows of its first operand with the even rows of its second; with both = v the two results are [r0 r0 r2 r2] and [r1 r1 r3 r3]
// (r = rows of 16 lanes), whose sum is the pair total in every lane — the fifth step of a 32-lane fold without the LDS crossbar
__device__ __forceinline__ float lvsr_swap16_sum(float v) {
    const unsigned b = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float lvsr_dpp_row_ror4(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x124, 0xf, 0xf, true));   // row_ror:4
}
__device__ __forceinline__ float lvsr_dpp_row_ror8(float v) {
    return __uint_as_float((unsigned)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x128, 0xf, 0xf, true));   // row_ror:8
}
// Wave-wide sum / max without the LDS crossbar: four DPP moves fold each row of 16 lanes (every lane of the row gets the row's
// total), four v_readlane fetch the rows' totals as scalars.  ~12 instructions instead of six ds_bpermute round trips; the
// result is wave-uniform.  (Order of the float additions differs from wave_sum's xor tree.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += lvsr_dpp_quad_xor1(v);
    v += lvsr_dpp_quad_xor2(v);
    v += lvsr_dpp_row_ror4(v);
    v += lvsr_dpp_row_ror8(v);
    const int b = (int)__float_as_uint(v);
    const float r0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 0)), r1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 16));
    const float r2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 32)), r3 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, lvsr_dpp_quad_xor1(v));
    v = fmaxf(v, lvsr_dpp_quad_xor2(v));
    v = fmaxf(v, lvsr_dpp_row_ror4(v));
    v = fmaxf(v, lvsr_dpp_row_ror8(v));
    const int b = (int)__float_as_uint(v);
    const float r0 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 0)), r1 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 16));
    const float r2 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 32)), r3 = __uint_as_float((unsigned)__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// "row-block matmul" used by every recurrent step kernel.
//
// A 256-thread work-group (4 waves) produces one 16x16 tile  out[r][c] = sum_k A(r,k) * W[k][c0+c]
// for its 16-row (utterance) tile; batch rows are the MFMA M dimension, so one recurrence step for
// 16 utterances is exactly one 16-row tile.  A recurrent step is latency bound (a few thousand
// dependent steps per layer), so the body is organised to have every operand load in flight at once:
//   * K is split in 4 contiguous slices of Kw (one per wave); inside a wave MFMA lane (i, kk) owns
//     Kw/4 CONSECUTIVE k values, so its A operand is NQ = Kw/16 float4 loads from one row;
//   * the weight is pre-packed (lvsr_pack_b) in exactly that order: P[tile][wave][q][lane][4], one
//     coalesced 1 KiB wave-load per q; the packed copy stays L2 resident across steps;
//   * all 2*NQ float4 loads are issued before the first MFMA; two accumulators alternate so the
//     40-cycle dependent MFMA latency never exceeds the 32-cycle issue interval.
// The order in which k values meet inside the fmaf chain is a property of the packing only, so it is
// identical for every launch (deterministic results).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int lvsr_pack_kw(int K) { return ((K + 3) / 4 + 15) / 16 * 16; }

struct RowSrc {            // A operand rows: row i at base + i*ld (ld = 0 broadcasts one row), valid rows < nrows
    const float* base; long long ld; int nrows; int K; bool vec; bool fast;
    // FAST (uniform, decided once per kernel): 16-B aligned rows and no K padding -> one unguarded float4 load; rows
    // beyond nrows re-read the last valid row (their results are discarded by the epilogue).
    template <bool FAST>
    __device__ __forceinline__ float4 get(int i, int k) const {
        if (FAST) return *(const float4*)(base + (size_t)min(i, nrows - 1) * ld + k);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= nrows || k >= K) return v;
        const float* p = base + (size_t)i * ld + k;
        if (vec && k + 3 < K) return *(const float4*)p;
        v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
        return v;
    }
};
__device__ __forceinline__ bool rb_no_kpad(int K) { return K == 4 * lvsr_pack_kw(K); }
__device__ __forceinline__ RowSrc row_src(const float* base, long long ld, int nrows, int K) {
    RowSrc s;
    s.base = base; s.ld = ld; s.nrows = nrows; s.K = K;
    s.vec = ((ld & 3) == 0) && ((K & 3) == 0) && ((((size_t)base) & 15) == 0);
    s.fast = s.vec && nrows > 0 && rb_no_kpad(K);
    return s;
}

// guarded 4-element load: p[0..3] where only the first `nvalid` exist; vec = 16-B aligned fast path allowed
__device__ __forceinline__ float4 ld4g(const float* p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid <= 0) return v;
    if (vec && nvalid >= 4) return *(const float4*)p;
    v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

template <int N, bool FAST, class A4>
__device__ __forceinline__ void rb_chunk(f32x4& acc0, f32x4& acc1, const A4& a4, const float4* __restrict__ p, int i,
                                         int k0) {
    float4 a[N], b[N];
#pragma unroll
    for (int q = 0; q < N; ++q) {
        b[q] = p[q * 64];
        a[q] = a4.template get<FAST>(i, k0 + 4 * q);
    }
    // keep every load above the first MFMA: the scheduler otherwise interleaves load / wait / 4 MFMAs and exposes the
    // memory latency once per q instead of once per chunk
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < N; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc1, 0, 0, 0);
    }
}

// Accumulate this wave's K-slice of tile `tile` of packed weight P (logical K x N) into acc0/acc1.
// a4.get<FAST>(i, k) returns A[i][k..k+3] (zero beyond K / beyond the valid rows on the guarded path).
template <bool FAST, class A4>
__device__ __forceinline__ void rb_mm_impl(f32x4& acc0, f32x4& acc1, const A4& a4, const float* __restrict__ P, int K,
                                           int tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int Kw = lvsr_pack_kw(K), NQ = Kw >> 4;
    int k0 = wave * Kw + kk * (Kw >> 2);
    const float4* p = (const float4*)P + ((size_t)(tile * 4 + wave) * NQ) * 64 + lane;
    int q = 0;
    for (; q + 8 <= NQ; q += 8, p += 8 * 64, k0 += 32) rb_chunk<8, FAST>(acc0, acc1, a4, p, i, k0);
    if (q + 4 <= NQ) { rb_chunk<4, FAST>(acc0, acc1, a4, p, i, k0); q += 4; p += 4 * 64; k0 += 16; }
    if (q + 2 <= NQ) { rb_chunk<2, FAST>(acc0, acc1, a4, p, i, k0); q += 2; p += 2 * 64; k0 += 8; }
    if (q < NQ) rb_chunk<1, FAST>(acc0, acc1, a4, p, i, k0);
}
// Compile-time K (= 64*NQ, no padding): one straight-line chunk, no dispatch on the chunk count.
template <int NQ, class A4>
__device__ __forceinline__ void rb_mm_fixed(f32x4& acc0, f32x4& acc1, const A4& a4, const float* __restrict__ P, int tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int k0 = wave * (16 * NQ) + kk * (4 * NQ);
    const float4* p = (const float4*)P + ((size_t)(tile * 4 + wave) * NQ) * 64 + lane;
    rb_chunk<NQ, true>(acc0, acc1, a4, p, i, k0);
}
// FAST kernels call this: H = 256 / 512 take the fixed-size path, other multiples of 64 the generic unguarded one.
template <class A4>
__device__ __forceinline__ void rb_mm_fast(f32x4& acc0, f32x4& acc1, const A4& a4, const float* __restrict__ P, int K, int tile) {
    if (K == 256) rb_mm_fixed<4>(acc0, acc1, a4, P, tile);
    else if (K == 512) rb_mm_fixed<8>(acc0, acc1, a4, P, tile);
    else rb_mm_impl<true>(acc0, acc1, a4, P, K, tile);
}
template <bool FAST, class A4>
__device__ __forceinline__ void rb_mm_sel(f32x4& acc0, f32x4& acc1, const A4& a4, const float* __restrict__ P, int K, int tile) {
    if (FAST) rb_mm_fast(acc0, acc1, a4, P, K, tile);
    else rb_mm_impl<false>(acc0, acc1, a4, P, K, tile);
}

template <class A4>
__device__ __forceinline__ void rb_mm(f32x4& acc0, f32x4& acc1, const A4& a4, const float* __restrict__ P, int K,
                                      int tile) {
    if (a4.fast) rb_mm_impl<true>(acc0, acc1, a4, P, K, tile);
    else rb_mm_impl<false>(acc0, acc1, a4, P, K, tile);
}

// Sum the per-wave partial tiles; afterwards thread tid owns element (row = tid>>4, col = tid&15).
__device__ __forceinline__ float rb_reduce(f32x4 acc0, f32x4 acc1) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // protect `red` against a previous use in the same kernel
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc0[r] + acc1[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    return ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
}
// same, for kernels that fold exactly one tile (no earlier use of `red` to protect): one barrier instead of two
__device__ __forceinline__ float rb_reduce_once(f32x4 acc0, f32x4 acc1) {
    __shared__ float red1[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) red1[wave][(lane >> 4) * 4 + r][lane & 15] = acc0[r] + acc1[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    return ((red1[0][row][col] + red1[1][row][col]) + red1[2][row][col]) + red1[3][row][col];
}
#define F32X4_ZERO ((f32x4){0.f, 0.f, 0.f, 0.f})
