// Standalone probe (not part of the product): what keeps an fp32-MFMA GEMM inner loop below the instruction's rate on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip && ./mfma_rate_probe
// The loop body of csrc/gemm.hip's 128 x 128 x 32 tile (per wave: 64 v_mfma_f32_32x32x2_f32 on 4 accumulator blocks per k-tile) is
// rebuilt with its ingredients switched on one at a time (MODE bits):
//   1  MFMA operands come from LDS (2 ds_read2_b32 per 4 MFMAs, the GEMM's addressing) instead of registers
//   2  per k-tile: the 20 ds_write of the operand staging + the two work-group barriers
//   4  per k-tile: 8 global_load_dwordx4 per thread (operand panels of a 12 800 x 512 / 512 x 1 536 product, L2 resident), consumed by the LDS writes
// grid = 2 work-groups of 256 threads per CU (67 KB of LDS each, as the product kernel), or 1 work-group of 512 threads (WAVES8).
// Prints TFLOP/s per mode: the differences say whether the gap to 157 TFLOP/s is the LDS traffic, the barriers, the loads or the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GK 32
#define LD 132

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void probe(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int lda, int ldb) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = threadIdx.x; k < 2 * GK * LD; k += NT) { (&As[0][0][0])[k] = 1e-3f * (k & 15); (&Bs[0][0][0])[k] = 1e-3f * (k & 7); }
    __syncthreads();
    const int kr0 = lane >> 5, li = lane & 31;
    float ra0 = 1e-3f * lane, ra1 = 2e-3f * lane, rb0 = 1e-3f, rb1 = 3e-3f;
    float4 ga[4], gb[4];
    const int m0 = (blockIdx.x % 100) * 128, n0 = (blockIdx.x % 12) * 128;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 4) {
            const int k0 = (it & 15) * GK;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int u = (threadIdx.x & 255) + h * 256;
                ga[h] = *(const float4*)(A + (size_t)(m0 + (u % 128)) * lda + k0 + (u / 128) * 4);
                gb[h] = *(const float4*)(B + (size_t)(k0 + u / 32) * ldb + n0 + (u % 32) * 4);
            }
        }
        float pa[2][2], pb[2][2];
        if (MODE & 1) {
            pa[0][0] = As[cur][kr0][wm + li]; pa[0][1] = As[cur][kr0][wm + 32 + li];
            pb[0][0] = Bs[cur][kr0][wn + li]; pb[0][1] = Bs[cur][kr0][wn + 32 + li];
        } else { pa[0][0] = ra0; pa[0][1] = ra1; pb[0][0] = rb0; pb[0][1] = rb1; }
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < GK / 2) {
                if (MODE & 1) {
                    const int kr = 2 * (s + 1) + kr0;
                    pa[n][0] = As[cur][kr][wm + li]; pa[n][1] = As[cur][kr][wm + 32 + li];
                    pb[n][0] = Bs[cur][kr][wn + li]; pb[n][1] = Bs[cur][kr][wn + 32 + li];
                } else { pa[n][0] = pa[c][1]; pa[n][1] = pa[c][0]; pb[n][0] = pb[c][1]; pb[n][1] = pb[c][0]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c][i], pb[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int u = (threadIdx.x & 255) + h * 256;
                const float4 va = (MODE & 4) ? ga[h] : make_float4(ra0, ra1, rb0, rb1);
                const float4 vb = (MODE & 4) ? gb[h] : make_float4(rb0, rb1, ra0, ra1);
                const int x = u % 128, k = (u / 128) * 4;
                if (!(MODE & 8) && (NT == 256 || wave < 4)) {
                    As[cur ^ 1][k + 0][x] = va.x; As[cur ^ 1][k + 1][x] = va.y; As[cur ^ 1][k + 2][x] = va.z; As[cur ^ 1][k + 3][x] = va.w;
                    *(float4*)&Bs[cur ^ 1][u / 32][(u % 32) * 4] = vb;
                }
                if (MODE & 8) ra0 += va.x;
            }
            if (!(MODE & 16)) __syncthreads();
            if (!(MODE & 16)) cur ^= 1;
        } else if (MODE & 4) {
            ra0 += ga[0].x + ga[1].y + ga[2].z + ga[3].w + gb[0].x + gb[1].y + gb[2].z + gb[3].w;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NT + threadIdx.x] = s + ra0;
}

// ---- the pipelined loop: ONE barrier per k-tile, nothing between the MFMAs of a tile and those of the next but the barrier itself.
// Per tile (buffer b): the operands of tile t+1 (registers, loaded a whole tile ago) are written to buffer b^1 behind MFMA step WR_AT
// and the loads of tile t+2 are issued into the same registers; behind step BAR_AT the operands of the remaining steps of THIS tile
// are preloaded, lgkmcnt(0), s_barrier (no vmcnt wait: __syncthreads() would drain the prefetch), the first operands of tile t+1 are
// fetched from b^1 and the remaining MFMA steps of tile t run while they arrive.
template <int WR_AT, int BAR_AT, int NT, bool GLOBAL>
__global__ __launch_bounds__(NT) void probe2(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int lda, int ldb) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = threadIdx.x; k < 2 * GK * LD; k += NT) { (&As[0][0][0])[k] = 1e-3f * (k & 15); (&Bs[0][0][0])[k] = 1e-3f * (k & 7); }
    __syncthreads();
    const int kr0 = lane >> 5, li = lane & 31;
    float4 ga[4], gb[4];
    const int m0 = (blockIdx.x % 100) * 128, n0 = (blockIdx.x % 12) * 128;
    auto gload = [&](int t) {
        const int k0 = (t & 15) * GK;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int u = (threadIdx.x & 255) + h * 256;
            if (GLOBAL) {
                ga[h] = *(const float4*)(A + (size_t)(m0 + (u % 128)) * lda + k0 + (u / 128) * 4);
                gb[h] = *(const float4*)(B + (size_t)(k0 + u / 32) * ldb + n0 + (u % 32) * 4);
            } else { ga[h] = make_float4(1e-3f * lane, 2e-3f, 3e-3f, 1e-3f * t); gb[h] = make_float4(1e-3f, 2e-3f * lane, 3e-3f, 1e-3f * t); }
        }
    };
    gload(1);
    int cur = 0;
    constexpr int NS = GK / 2, REST = NS - 1 - BAR_AT;            // MFMA steps behind the barrier
    float pa[2][2], pb[2][2], qa[REST > 0 ? REST : 1][2], qb[REST > 0 ? REST : 1][2];
    pa[0][0] = As[cur][kr0][wm + li]; pa[0][1] = As[cur][kr0][wm + 32 + li];
    pb[0][0] = Bs[cur][kr0][wn + li]; pb[0][1] = Bs[cur][kr0][wn + 32 + li];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s < BAR_AT) {
                const int kr = 2 * (s + 1) + kr0;
                pa[n][0] = As[cur][kr][wm + li]; pa[n][1] = As[cur][kr][wm + 32 + li];
                pb[n][0] = Bs[cur][kr][wn + li]; pb[n][1] = Bs[cur][kr][wn + 32 + li];
            } else if (s == BAR_AT) {
#pragma unroll
                for (int q = 0; q < REST; ++q) {
                    const int kr = 2 * (s + 1 + q) + kr0;
                    qa[q][0] = As[cur][kr][wm + li]; qa[q][1] = As[cur][kr][wm + 32 + li];
                    qb[q][0] = Bs[cur][kr][wn + li]; qb[q][1] = Bs[cur][kr][wn + 32 + li];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s <= BAR_AT) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c][i], pb[c][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s - BAR_AT - 1][i], qb[s - BAR_AT - 1][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == WR_AT) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int u = (threadIdx.x & 255) + h * 256;
                    const int x = u % 128, k = (u / 128) * 4;
                    if (NT == 256 || wave < 4) {
                        As[cur ^ 1][k + 0][x] = ga[h].x; As[cur ^ 1][k + 1][x] = ga[h].y; As[cur ^ 1][k + 2][x] = ga[h].z; As[cur ^ 1][k + 3][x] = ga[h].w;
                        *(float4*)&Bs[cur ^ 1][u / 32][(u % 32) * 4] = gb[h];
                    }
                }
                gload(it + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s == BAR_AT) {
                __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0) only: the staging writes and the preloads, not the prefetch
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                cur ^= 1;
                pa[(NS & 1)][0] = As[cur][kr0][wm + li]; pa[(NS & 1)][1] = As[cur][kr0][wm + 32 + li];
                pb[(NS & 1)][0] = Bs[cur][kr0][wn + li]; pb[(NS & 1)][1] = Bs[cur][kr0][wn + 32 + li];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * NT + threadIdx.x] = s + ga[0].x + gb[0].x;
}

template <int WR_AT, int BAR_AT, int NT, bool GLOBAL>
static void run2(const char* name, const float* A, const float* B, float* out, int grid) {
    const int iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe2<WR_AT, BAR_AT, NT, GLOBAL>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe2<WR_AT, BAR_AT, NT, GLOBAL>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * (NT / 64) * iters * 64.0 * (32.0 * 32.0 * 2.0 * 2.0) * reps;
    printf("pipelined WR_AT %2d BAR_AT %2d %-38s grid %4d x %3d threads  %8.1f us per launch  %6.1f TFLOP/s  (%.3f of 157.3)\n", WR_AT, BAR_AT, name, grid, NT,
           ms * 1e3 / reps, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

// ---- no LDS, no barrier: every wave loads its own MFMA fragments from global memory, one k-tile (32) ahead, double-buffered in
// registers.  32x32x2: lane l holds row/column l % 32 and ONE k per instruction; lanes 0-31 take k0 + j, lanes 32-63 k0 + 16 + j
// (every k once; the order of the k summation differs from the LDS kernel's).  A (m, k) with k contiguous: 4 float4 per 32-row block;
// B (k, n) with n contiguous: 16 dwords per 32-column block (each a 128-byte row segment per half-wave).
template <int NT, bool B_FROM_LDS>
__global__ __launch_bounds__(NT) void probe3(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int lda, int ldb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    const int m0 = (blockIdx.x % 100) * 128 + wm, n0 = (blockIdx.x % 12) * 128 + wn;
    const int li = lane & 31, kh = (lane >> 5) * 16;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 fa[2][2][4];
    float fb[2][2][16];
    auto gload = [&](int t, int buf) {
        const int k0 = (t & 15) * GK + kh;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) fa[buf][i][q] = *(const float4*)(A + (size_t)(m0 + 32 * i + li) * lda + k0 + 4 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) fb[buf][j][q] = B[(size_t)(k0 + q) * ldb + n0 + 32 * j + li];
    };
    gload(0, 0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            gload(it + half + 1, half ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 v = fa[half][i][s >> 2];
                        const float a = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, fb[half][j][s], acc[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int NT>
static void run3(const char* name, const float* A, const float* B, float* out, int grid) {
    const int iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe3<NT, false>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe3<NT, false>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * (NT / 64) * iters * 64.0 * (32.0 * 32.0 * 2.0 * 2.0) * reps;
    printf("fragments straight from global memory, no LDS: %-30s grid %4d x %3d threads  %8.1f us per launch  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, grid, NT,
           ms * 1e3 / reps, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

// ---- what ONE work-group barrier costs a wave that does nothing but MFMAs from LDS operands: NBAR barriers spread over the 16 steps
template <int NBAR, int NT, bool RAW>
__global__ __launch_bounds__(NT) void probe4(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][GK][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = threadIdx.x; k < 2 * GK * LD; k += NT) { (&As[0][0][0])[k] = 1e-3f * (k & 15); (&Bs[0][0][0])[k] = 1e-3f * (k & 7); }
    __syncthreads();
    const int kr0 = lane >> 5, li = lane & 31;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        float pa[2][2], pb[2][2];
        pa[0][0] = As[cur][kr0][wm + li]; pa[0][1] = As[cur][kr0][wm + 32 + li];
        pb[0][0] = Bs[cur][kr0][wn + li]; pb[0][1] = Bs[cur][kr0][wn + 32 + li];
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < GK / 2) {
                const int kr = 2 * (s + 1) + kr0;
                pa[n][0] = As[cur][kr][wm + li]; pa[n][1] = As[cur][kr][wm + 32 + li];
                pb[n][0] = Bs[cur][kr][wn + li]; pb[n][1] = Bs[cur][kr][wn + 32 + li];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c][i], pb[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (NBAR > 0 && (s % (16 / (NBAR > 0 ? NBAR : 1))) == 7 % (16 / (NBAR > 0 ? NBAR : 1))) {
                if (RAW) __builtin_amdgcn_s_barrier(); else __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        cur ^= 1;
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int NBAR, int NT, bool RAW>
static void run4(float* out, int grid) {
    const int iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe4<NBAR, NT, RAW>), dim3(grid), dim3(NT), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe4<NBAR, NT, RAW>), dim3(grid), dim3(NT), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_iter = ms * 1e3 / reps / iters;
    printf("barrier cost: %d %s barriers per 64 MFMAs, grid %4d x %3d threads: %7.3f us per k-tile = %6.0f cycles at 2.4 GHz (64 MFMAs = 4096)\n", NBAR,
           RAW ? "raw s_barrier" : "__syncthreads", grid, NT, us_iter, us_iter * 2400.0);
}

// ---- operands as 16-byte LDS reads: LDS image [row][k] (k contiguous, row stride 36 floats: conflict-free b128 reads), lane l reads
// k = (l / 32) * 16 + 4 q .. + 3 of row l % 32 — four MFMA steps per read; lanes 0-31 / 32-63 split the k-tile in halves (every k once).
// VARIANT 0: the 4 reads of quarter q + 1 are issued before the 16 MFMAs of quarter q;  1: all 16 reads of the tile up front;
// 2: as 0 without scheduling fences
template <int VARIANT, int NT, bool TOGGLE>
__global__ __launch_bounds__(NT) void probe5(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][128][36];
    __shared__ __attribute__((aligned(16))) float Bs[2][128][36];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = threadIdx.x; k < 2 * 128 * 36; k += NT) { (&As[0][0][0])[k] = 1e-3f * (k & 15); (&Bs[0][0][0])[k] = 1e-3f * (k & 7); }
    __syncthreads();
    const int li = lane & 31, kh = (lane >> 5) * 16;
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        float4 a[2][2], b[2][2];          // [buffer][block]
        float4 aa[4][2], bb[4][2];
        if (VARIANT == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    aa[q][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh + 4 * q];
                    bb[q][i] = *(const float4*)&Bs[cur][wn + 32 * i + li][kh + 4 * q];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[0][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh]; b[0][i] = *(const float4*)&Bs[cur][wn + 32 * i + li][kh]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q & 1, n = c ^ 1;
            if (VARIANT != 1 && q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[n][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh + 4 * (q + 1)];
                    b[n][i] = *(const float4*)&Bs[cur][wn + 32 * i + li][kh + 4 * (q + 1)];
                }
            }
            if (VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 va = VARIANT == 1 ? aa[q][i] : a[c][i], vb = VARIANT == 1 ? bb[q][j] : b[c][j];
                        const float x = e == 0 ? va.x : e == 1 ? va.y : e == 2 ? va.z : va.w;
                        const float y = e == 0 ? vb.x : e == 1 ? vb.y : e == 2 ? vb.z : vb.w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
            if (VARIANT != 2) __builtin_amdgcn_sched_barrier(0);
        }
        if (TOGGLE) cur ^= 1;
        else As[cur][threadIdx.x & 127][35] = acc[0][0][0];       // (a write the compiler cannot see through: the reads are not loop invariant)
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * NT + threadIdx.x] = sum;
}

template <int VARIANT, int NT, bool TOGGLE>
static void run5(float* out, int grid) {
    const int iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe5<VARIANT, NT, TOGGLE>), dim3(grid), dim3(NT), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe5<VARIANT, NT, TOGGLE>), dim3(grid), dim3(NT), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_iter = ms * 1e3 / reps / iters;
    printf("b128 operand reads, variant %d, toggle %d, grid %4d x %3d threads: %7.3f us per k-tile = %6.0f cycles at 2.4 GHz per %d wave(s) per SIMD (64 MFMAs = 4096)\n",
           VARIANT, (int)TOGGLE, grid, NT, us_iter, us_iter * 2400.0, grid * (NT / 64) / 1024);
}

// ---- the whole loop on the [row][k] image (b128 operand reads): global loads (GL), staging writes (WR), one raw barrier per k-tile.
// SPREAD = false: loads in a block at the top, writes in a block at the bottom (the product kernel's order).
// SPREAD = true: the staging writes of tile t+1 (loaded during tile t-1) go behind the first quarter's MFMAs, the loads of tile t+2
// are spread over the other three quarters — no instruction block without MFMAs in flight except the barrier itself.
template <bool GL, bool WR, bool SPREAD, int TRANSPOSED_B>
__global__ __launch_bounds__(256) void probe6(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters, int lda, int ldb) {
    __shared__ __attribute__((aligned(16))) float As[2][128][36];
    __shared__ __attribute__((aligned(16))) float Bs[2][128][36];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const int wm = ((wave >> 1) & 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int k = tid; k < 2 * 128 * 36; k += 256) { (&As[0][0][0])[k] = 1e-3f * (k & 15); (&Bs[0][0][0])[k] = 1e-3f * (k & 7); }
    __syncthreads();
    const int li = lane & 31, kh = (lane >> 5) * 16;
    const int kslabs = lda / GK;
    const int m0 = (ldb < 0 ? blockIdx.x : blockIdx.x % 100) * 128, n0 = (blockIdx.x % 12) * 128;
    if (ldb < 0) ldb = -ldb;          // (ldb < 0: every work-group its own A panel — operands streamed from HBM)
    float4 ra[4], rb[4];
    for (int h = 0; h < 4; ++h) { ra[h] = make_float4(1e-3f * lane, 0.f, 0.f, 0.f); rb[h] = ra[h]; }
    auto load_a = [&](int t, int h) {          // k contiguous: 8 lanes = one 128-byte row
        const int u = tid + h * 256;
        if (GL) ra[h] = *(const float4*)(A + (size_t)(m0 + (u >> 3)) * lda + (t % kslabs) * GK + (u & 7) * 4);
    };
    auto load_b = [&](int t, int q) {          // rows contiguous: 4 consecutive k of one column per float4
        if (!GL) return;
        if (TRANSPOSED_B == 1) {                // B as (n, k), k contiguous (the NT product): same pattern as A, on the A buffer's rows
            const int u = tid + q * 256;
            rb[q] = *(const float4*)(A + (size_t)(n0 + (u >> 3)) * lda + (t % kslabs) * GK + (u & 7) * 4);
        } else if (TRANSPOSED_B == 2) {         // rows contiguous, float4 along the rows: lane = (8 row groups) x (8 k), a wave = 8 lines of 128 bytes
            const int k = ((tid >> 6) & 3) * 8 + (tid & 7), r = q * 8 + ((tid >> 3) & 7);
            rb[q] = *(const float4*)(B + (size_t)((t % kslabs) * GK + k) * ldb + n0 + 4 * r);
        } else {
            const float* col = B + n0 + (tid & 127);
            const int k = (t % kslabs) * GK + (tid >> 7) * 16 + 4 * q;
            rb[q] = make_float4(col[(size_t)k * ldb], col[(size_t)(k + 1) * ldb], col[(size_t)(k + 2) * ldb], col[(size_t)(k + 3) * ldb]);
        }
    };
    auto store_all = [&](int buf) {
        if (!WR) return;
#pragma unroll
        for (int h = 0; h < 4; ++h) { const int u = tid + h * 256; *(float4*)&As[buf][u >> 3][(u & 7) * 4] = ra[h]; }
        if (TRANSPOSED_B == 1) {
#pragma unroll
            for (int h = 0; h < 4; ++h) { const int u = tid + h * 256; *(float4*)&Bs[buf][u >> 3][(u & 7) * 4] = rb[h]; }
        } else if (TRANSPOSED_B == 2) {         // four scalar writes per float4: banks 16 r + k (+ 36 e): two lanes per bank
            const int k = ((tid >> 6) & 3) * 8 + (tid & 7);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int r = h * 8 + ((tid >> 3) & 7);
                Bs[buf][4 * r + 0][k] = rb[h].x; Bs[buf][4 * r + 1][k] = rb[h].y; Bs[buf][4 * r + 2][k] = rb[h].z; Bs[buf][4 * r + 3][k] = rb[h].w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(float4*)&Bs[buf][tid & 127][(tid >> 7) * 16 + 4 * q] = rb[q];
        }
    };
    int cur = 0;
    for (int it = 0; it < iters; ++it) {
        if (!SPREAD) {
#pragma unroll
            for (int h = 0; h < 4; ++h) { load_a(it + 1, h); load_b(it + 1, h); }
        }
        float4 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[0][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh]; b[0][i] = *(const float4*)&Bs[cur][wn + 32 * i + li][kh]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q & 1, n = c ^ 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[n][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh + 4 * (q + 1)];
                    b[n][i] = *(const float4*)&Bs[cur][wn + 32 * i + li][kh + 4 * (q + 1)];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float x = e == 0 ? a[c][i].x : e == 1 ? a[c][i].y : e == 2 ? a[c][i].z : a[c][i].w;
                        const float y = e == 0 ? b[c][j].x : e == 1 ? b[c][j].y : e == 2 ? b[c][j].z : b[c][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
                if (SPREAD) {
                    // behind every group of 4 MFMAs one piece of the staging work
                    if (q == 0) {
                        if (e == 0) store_all(cur ^ 1);
                    } else {
                        const int piece = (q - 1) * 4 + e;           // 0 .. 11: the 8 loads of tile it + 2 (A 0-3, B 0-3) in the first 8
                        if (piece < 4) load_a(it + 2, piece);
                        else if (piece < 8) load_b(it + 2, piece - 4);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!SPREAD) store_all(cur ^ 1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum + ra[0].x + rb[0].x;
}

template <bool GL, bool WR, bool SPREAD, int TB>
static void run6(const float* A, const float* B, float* out, int grid, int lda = 512, int ldb = 1536) {
    const int iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe6<GL, WR, SPREAD, TB>), dim3(grid), dim3(256), 0, 0, A, B, out, iters, lda, ldb);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe6<GL, WR, SPREAD, TB>), dim3(grid), dim3(256), 0, 0, A, B, out, iters, lda, ldb);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 * iters * 64.0 * (32.0 * 32.0 * 2.0 * 2.0) * reps;
    printf("[row][k] image (lda %d%s): global loads %d staging writes %d spread %d B k-contiguous %d  grid %4d  %8.1f us  %6.1f TFLOP/s  (%.3f of 157.3)\n", lda, ldb < 0 ? ", own A panel per work-group" : "", (int)GL, (int)WR,
           (int)SPREAD, (int)TB, grid, ms * 1e3 / reps, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

template <int MODE, int NT>
static void run(const char* name, const float* A, const float* B, float* out, int grid) {
    const int iters = 256;                     // 256 k-tiles of 64 MFMAs per wave
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, NT>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, NT>), dim3(grid), dim3(NT), 0, 0, A, B, out, iters, 512, 1536);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * (NT / 64) * iters * 64.0 * (32.0 * 32.0 * 2.0 * 2.0) * reps;
    printf("%-64s grid %4d x %3d threads  %8.1f us per launch  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, grid, NT, ms * 1e3 / reps,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    float *A, *B, *out;
    CK(hipMalloc(&A, (size_t)12800 * 512 * 4)); CK(hipMalloc(&B, (size_t)512 * 1536 * 4)); CK(hipMalloc(&out, (size_t)2048 * 512 * 4));
    CK(hipMemset(A, 0, (size_t)12800 * 512 * 4)); CK(hipMemset(B, 0, (size_t)512 * 1536 * 4));
    for (int round = 0; round < 1; ++round) {
        run<0, 256>("registers only (MFMA rate of this run's clocks)", A, B, out, 512);
        run<1, 256>("+ operands from LDS", A, B, out, 512);
        run<3, 256>("+ operands from LDS + staging writes + 2 barriers per k-tile", A, B, out, 512);
        run<7, 256>("+ operands from LDS + staging + barriers + global loads (= the GEMM loop)", A, B, out, 512);
        run<5, 256>("+ operands from LDS + global loads, no staging / barriers", A, B, out, 512);
        run<0, 256>("registers only, ONE work-group per CU (1 wave per SIMD)", A, B, out, 256);
        run<7, 256>("the GEMM loop, ONE work-group per CU", A, B, out, 256);
        run<7, 256>("the GEMM loop, 1024 work-groups (two rounds)", A, B, out, 1024);
        run<0, 512>("registers only, 512-thread work-groups (2 waves per SIMD, one WG per CU)", A, B, out, 256);
        run<7, 512>("the GEMM loop in 512-thread work-groups, one per CU", A, B, out, 256);
    }
    { float* A2; CK(hipMalloc(&A2, (size_t)12800 * 4096 * 4)); CK(hipMemset(A2, 0, (size_t)12800 * 4096 * 4));
      float* B2; CK(hipMalloc(&B2, (size_t)4096 * 1536 * 4)); CK(hipMemset(B2, 0, (size_t)4096 * 1536 * 4));
      run6<true, true, false, 1>(A2, B2, out, 512, 512); run6<true, true, false, 1>(A2, B2, out, 512, 1024); run6<true, true, false, 1>(A2, B2, out, 512, 2048); run6<true, true, false, 1>(A2, B2, out, 512, 4096);
      run6<true, true, false, 2>(A2, B2, out, 512, 512); run6<true, true, false, 2>(A2, B2, out, 512, 2048); run6<true, true, false, 2>(A2, B2, out, 512, 4096);
      run6<true, true, false, 2>(A2, B2, out, 1024, 2048);
      float* A3; CK(hipMalloc(&A3, (size_t)1024 * 128 * 2048 * 4)); CK(hipMemset(A3, 0, (size_t)1024 * 128 * 2048 * 4));
      run6<true, true, false, 2>(A3, B2, out, 512, 2048, -1536); run6<true, true, false, 2>(A3, B2, out, 1024, 2048, -1536); run6<true, true, false, 1>(A3, B2, out, 1024, 2048, -1536);
      run6<true, true, false, 2>(A3, B2, out, 256, 2048, -1536); }
    run6<false, false, false, 0>(A, B, out, 512); run6<false, true, false, 0>(A, B, out, 512); run6<false, true, false, 1>(A, B, out, 512); run6<false, true, false, 2>(A, B, out, 512);
    run6<true, true, false, 0>(A, B, out, 512); run6<true, true, false, 1>(A, B, out, 512); run6<true, true, false, 2>(A, B, out, 512);
    run6<true, true, false, 2>(A, B, out, 1024); run6<true, true, false, 1>(A, B, out, 1024);
    run5<0, 256, true>(out, 256); run5<1, 256, true>(out, 256); run5<2, 256, true>(out, 256); run5<0, 256, false>(out, 256);
    run5<0, 256, true>(out, 512); run5<1, 256, true>(out, 512); run5<2, 256, true>(out, 512);
    run4<0, 256, false>(out, 256); run4<1, 256, false>(out, 256); run4<2, 256, false>(out, 256); run4<4, 256, false>(out, 256); run4<8, 256, false>(out, 256);
    run4<1, 256, true>(out, 256); run4<4, 256, true>(out, 256);
    run4<0, 256, false>(out, 512); run4<1, 256, false>(out, 512); run4<4, 256, false>(out, 512);
    run4<0, 128, false>(out, 512); run4<1, 128, false>(out, 512); run4<4, 128, false>(out, 512);
    run4<0, 64, false>(out, 1024); run4<1, 64, false>(out, 1024); run4<4, 64, false>(out, 1024);
    run3<256>("2 work-groups per CU", A, B, out, 512);
    run3<256>("1 work-group per CU", A, B, out, 256);
    run3<256>("4 rounds", A, B, out, 2048);
    run3<512>("512 threads, 1 per CU", A, B, out, 256);
    run3<512>("512 threads, 2 per CU", A, B, out, 512);
    run3<128>("128 threads, 4 per CU", A, B, out, 1024);
    run<3 + 8, 256>("+ LDS operands + ONE barrier per k-tile, no staging writes", A, B, out, 512);
    run<3 + 16, 256>("+ LDS operands + staging writes, NO barrier", A, B, out, 512);
    run<3, 256>("+ LDS operands + staging writes + barrier, ONE work-group per CU", A, B, out, 256);
    run<3 + 8, 256>("+ LDS operands + barrier only, ONE work-group per CU", A, B, out, 256);
    run<3 + 16, 256>("+ LDS operands + staging writes only, ONE work-group per CU", A, B, out, 256);
    run<1, 256>("+ LDS operands, ONE work-group per CU", A, B, out, 256);
    run<5, 256>("+ LDS operands + global loads (no staging), ONE work-group per CU", A, B, out, 256);
    for (int round = 0; round < 1; ++round) {
        run2<10, 13, 256, true>("global loads", A, B, out, 512);
        run2<10, 13, 256, false>("no global loads", A, B, out, 512);
        run2<6, 13, 256, true>("global loads", A, B, out, 512);
        run2<10, 11, 256, true>("global loads", A, B, out, 512);
        run2<12, 14, 256, true>("global loads", A, B, out, 512);
        run2<10, 15, 256, true>("global loads (barrier behind the last step)", A, B, out, 512);
        run2<10, 13, 256, true>("global loads, one work-group per CU", A, B, out, 256);
        run2<10, 13, 256, true>("global loads, 1024 work-groups", A, B, out, 1024);
        run2<10, 13, 512, true>("global loads, 512 threads, 1 per CU", A, B, out, 256);
    }
    return 0;
}
