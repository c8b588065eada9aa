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

#define LVSR_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            lvsr_set_error(__VA_ARGS__);        \
            return LVSR_ERR_ARG;                \
        }                                       \
    } while (0)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

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
// for r in the work-group's 16-row (utterance) tile.  K is split over the 4 waves in MFMA-sized
// chunks of 4 (wave w takes chunks w, w+4, ...), partial tiles are summed through LDS in a fixed
// order (deterministic).  A(r,k) is a functor so elementwise pre-processing of the operand
// (e.g. dh*u*(1-c^2) in BPTT) is fused into the load.  W is row-major (K, ldw).
// Batch rows map to the MFMA M dimension: one step of the recurrence for 16 utterances is exactly
// one 16-row MFMA tile, so nothing is padded at the reference batch size.
// ---------------------------------------------------------------------------------------------
template <class AFn>
__device__ __forceinline__ f32x4 rb_partial(f32x4 acc, AFn afn, const float* __restrict__ W, int ldw, int K,
                                            int c0, int ncols) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const bool colok = (c0 + i) < ncols;
    for (int k0 = wave * 4; k0 < K; k0 += 16) {
        const int k = k0 + kk;
        float a = 0.f, b = 0.f;
        if (k < K) {
            a = afn(i, k);
            if (colok) b = W[(size_t)k * ldw + c0 + i];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// Sum the 4 per-wave partial tiles; afterwards thread tid owns element (row = tid>>4, col = tid&15).
__device__ __forceinline__ float rb_reduce(f32x4 acc) {
    __shared__ float red[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();   // protect `red` against a previous use in the same kernel
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    return ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
}
