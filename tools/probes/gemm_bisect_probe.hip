// Standalone probe (not part of the product): the product's 128 x 128 tile kernel (csrc/gemm.hip, included as is) next to the loop of
// mfma_rate_probe.hip on the SAME operands in the same process — where do 0.6 us per k-tile go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I attention-lvcsr_amd/csrc -o tools/probes/gemm_bisect_probe tools/probes/gemm_bisect_probe.hip attention-lvcsr_amd/csrc/runtime.hip
#include "../../attention-lvcsr_amd/csrc/gemm.hip"
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// the product tile with pieces switched off: VAR bit 1 = no epilogue stores, 2 = identity block -> tile map (no XCD order),
// 4 = raw barrier + lgkmcnt(0) instead of __syncthreads, 8 = every work-group its own rows of A (by = linear block index)
template <int VAR, int ORDER = 0>
__global__ __launch_bounds__(256) void variant_kernel(GemmArgs g) {
    constexpr int TM = 128, TN = 128, MI = 2, NI = 2;
    __shared__ __attribute__((aligned(16))) float As[2][TM][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][TN][GLD];
    const int total = gridDim.x * gridDim.y;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int t = (VAR & 2) ? L : gemm_xcd_order(L, total);
    const int bx = t % gridDim.x, by = t / gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = 0, kend = g.K;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    f32x16 acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[TM / 32], rb[TN / 32];
    gemm2_tile_load<true, TM, false>(g.A, g.lda, m0, g.M, kbeg, kend, true, ra);
    gemm2_tile_load<false, TN, false>(g.B, g.ldb, n0, g.N, kbeg, kend, true, rb);
    gemm2_tile_store<true, TM>(As[0], ra);
    gemm2_tile_store<false, TN>(Bs[0], rb);
    __syncthreads();
    int cur = 0;
    const int li = lane & 31, kh = (lane >> 5) * 16;
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        // ORDER: where the loads of the next k-tile are issued.  0 all in front of the MFMAs (the product), 1 all behind them (latency
        // exposed in front of the staging writes), 2 A in front / B behind, 3 B in front / A behind, 4 two behind each quarter's MFMAs
        // (the k-tile index is clamped instead of branching on `more`: the last tile re-reads itself, harmlessly)
        const int kn = more ? k0 + GK : k0;
        if (ORDER == 0 || ORDER == 2) gemm2_tile_load<true, TM, false>(g.A, g.lda, m0, g.M, kn, kend, true, ra);
        if (ORDER == 0 || ORDER == 3) gemm2_tile_load<false, TN, false>(g.B, g.ldb, n0, g.N, kn, kend, true, rb);
        float4 pa[2][MI], pb[2][NI];
        for (int i = 0; i < MI; ++i) pa[0][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh];
        for (int j = 0; j < NI; ++j) pb[0][j] = *(const float4*)&Bs[cur][wn + 32 * j + li][kh];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q & 1, n = c ^ 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) pa[n][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh + 4 * (q + 1)];
#pragma unroll
                for (int j = 0; j < NI; ++j) pb[n][j] = *(const float4*)&Bs[cur][wn + 32 * j + li][kh + 4 * (q + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const float x = e == 0 ? pa[c][i].x : e == 1 ? pa[c][i].y : e == 2 ? pa[c][i].z : pa[c][i].w;
                        const float y = e == 0 ? pb[c][j].x : e == 1 ? pb[c][j].y : e == 2 ? pb[c][j].z : pb[c][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
            if (ORDER == 4) {
                // two of the eight loads behind each quarter
                const int u = threadIdx.x + q * 256;
                ra[q] = *(const float4*)(g.A + (size_t)(m0 + (u >> 3)) * g.lda + kn + (u & 7) * 4);
                const int k = kn + (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
                rb[q] = *(const float4*)(g.B + (size_t)k * g.ldb + n0 + 4 * (q * 8 + ((threadIdx.x >> 3) & 7)));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ORDER == 1 || ORDER == 3) gemm2_tile_load<true, TM, false>(g.A, g.lda, m0, g.M, kn, kend, true, ra);
        if (ORDER == 1 || ORDER == 2) gemm2_tile_load<false, TN, false>(g.B, g.ldb, n0, g.N, kn, kend, true, rb);
        {
            gemm2_tile_store<true, TM>(As[cur ^ 1], ra);
            gemm2_tile_store<false, TN>(Bs[cur ^ 1], rb);
        }
        if (VAR & 4) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        else __syncthreads();
        cur ^= 1;
    }
    if (VAR & 1) {
        float s = 0.f;
        for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        g.C[(size_t)L * 256 + threadIdx.x] = s;
        return;
    }
    for (int i = 0; i < MI; ++i)
        for (int j = 0; j < NI; ++j)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn + j * 32 + (lane & 31);
                g.C[(size_t)m * g.ldc + n] = acc[i][j][r];
            }
}

// single LDS buffer, loads behind the MFMAs, two barriers per k-tile: 36.9 KB of LDS per work-group -> 3 (registers) or 4 per CU
__global__ __launch_bounds__(256) void single_buffer_kernel(GemmArgs g) {
    constexpr int TM = 128, TN = 128, MI = 2, NI = 2;
    __shared__ __attribute__((aligned(16))) float As[TM][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[TN][GLD];
    const int total = gridDim.x * gridDim.y;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    const int t = gemm_xcd_order(L, total);
    const int bx = t % gridDim.x, by = t / gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = by * TM, n0 = bx * TN;
    const int kend = g.K;
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    f32x16 acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[TM / 32], rb[TN / 32];
    const int li = lane & 31, kh = (lane >> 5) * 16;
    for (int k0 = 0; k0 < kend; k0 += GK) {
        gemm2_tile_load<true, TM, false>(g.A, g.lda, m0, g.M, k0, kend, true, ra);
        gemm2_tile_load<false, TN, false>(g.B, g.ldb, n0, g.N, k0, kend, true, rb);
        if (k0) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }      // everybody is through with the previous tile's image
        gemm2_tile_store<true, TM>(As, ra);
        gemm2_tile_store<false, TN>(Bs, rb);
        __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        float4 pa[2][MI], pb[2][NI];
        for (int i = 0; i < MI; ++i) pa[0][i] = *(const float4*)&As[wm + 32 * i + li][kh];
        for (int j = 0; j < NI; ++j) pb[0][j] = *(const float4*)&Bs[wn + 32 * j + li][kh];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q & 1, n = c ^ 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) pa[n][i] = *(const float4*)&As[wm + 32 * i + li][kh + 4 * (q + 1)];
#pragma unroll
                for (int j = 0; j < NI; ++j) pb[n][j] = *(const float4*)&Bs[wn + 32 * j + li][kh + 4 * (q + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const float x = e == 0 ? pa[c][i].x : e == 1 ? pa[c][i].y : e == 2 ? pa[c][i].z : pa[c][i].w;
                        const float y = e == 0 ? pb[c][j].x : e == 1 ? pb[c][j].y : e == 2 ? pb[c][j].z : pb[c][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < MI; ++i)
        for (int j = 0; j < NI; ++j)
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn + j * 32 + (lane & 31);
                g.C[(size_t)m * g.ldc + n] = acc[i][j][r];
            }
}

template <class F>
static double time_us(F&& launch, int reps = 10) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const int M = 16384, N = 512;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)M * 8192 * 4)); CK(hipMalloc(&B, (size_t)8192 * N * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemset(A, 0, (size_t)M * 8192 * 4)); CK(hipMemset(B, 0, (size_t)8192 * N * 4));
    lvsr_set_knob(LVSR_KNOB_GEMM_MID_TILES, 1);
    double prev[16] = {0};
    for (int K : {2048, 8192}) {
        GemmArgs g;
        g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = N; g.ldc = N; g.transA = 0; g.transB = 0;
        g.alpha = 1.f; g.beta = 0.f; g.ksplit = 1; g.kchunk = K; g.part = nullptr; g.batch = 1; g.sA = g.sB = g.sC = 0;
        dim3 grid(N / 128, M / 128, 1);
        double us[16];
        us[0] = time_us([&] { lvsr_sgemm(nullptr, 0, 0, M, N, K, 1.f, A, K, B, N, 0.f, C, N, nullptr, nullptr, 0); });
        us[1] = time_us([&] { hipLaunchKernelGGL((lvsr_sgemm128_kernel<false, false, true>), grid, dim3(256), 0, 0, g); });
        us[2] = time_us([&] { hipLaunchKernelGGL((variant_kernel<0>), grid, dim3(256), 0, 0, g); });
        us[3] = time_us([&] { hipLaunchKernelGGL((variant_kernel<1>), grid, dim3(256), 0, 0, g); });
        us[4] = time_us([&] { hipLaunchKernelGGL((variant_kernel<2>), grid, dim3(256), 0, 0, g); });
        us[5] = time_us([&] { hipLaunchKernelGGL((variant_kernel<4>), grid, dim3(256), 0, 0, g); });
        us[6] = time_us([&] { hipLaunchKernelGGL((variant_kernel<7>), grid, dim3(256), 0, 0, g); });
        us[7] = time_us([&] { hipLaunchKernelGGL((variant_kernel<4, 1>), grid, dim3(256), 0, 0, g); });
        us[8] = time_us([&] { hipLaunchKernelGGL((variant_kernel<4, 2>), grid, dim3(256), 0, 0, g); });
        us[9] = time_us([&] { hipLaunchKernelGGL((variant_kernel<4, 3>), grid, dim3(256), 0, 0, g); });
        us[10] = time_us([&] { hipLaunchKernelGGL((variant_kernel<4, 4>), grid, dim3(256), 0, 0, g); });
        us[11] = time_us([&] { hipLaunchKernelGGL((variant_kernel<0, 1>), grid, dim3(256), 0, 0, g); });
        us[12] = time_us([&] { hipLaunchKernelGGL((variant_kernel<0, 3>), grid, dim3(256), 0, 0, g); });
        const char* names[13] = {"lvsr_sgemm (entry point)", "lvsr_sgemm128_kernel<NN, fast> launched directly", "variant 0: the same tile, fast path only",
                                "variant 1: no epilogue stores", "variant 2: identity block -> tile map", "variant 4: raw s_barrier + lgkmcnt(0)", "variant 7: all three",
                                "raw barrier, loads all BEHIND the MFMAs", "raw barrier, A in front / B behind", "raw barrier, B in front / A behind",
                                "raw barrier, two loads behind each quarter", "__syncthreads, loads all behind", "__syncthreads, B in front / A behind"};
        for (int v = 0; v < 13; ++v) {
            printf("K = %5d  %-52s %8.1f us  %6.1f TFLOP/s", K, names[v], us[v], 2.0 * M * N * K / us[v] / 1e6);
            if (K == 8192) printf("   slope %.3f us per k-tile", (us[v] - prev[v]) / ((8192 - 2048) / 32));
            printf("\n");
            prev[v] = us[v];
        }
    }
    // one round at 3 work-groups per CU: 768 tiles = 24 576 rows (the A buffer holds 16 384 x 8 192 = 24 576 x 5 461: K = 4 096 fits)
    for (int K : {1024, 4096}) {
        const int M3 = 24576;
        GemmArgs g;
        g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.M = M3; g.N = N; g.K = K; g.lda = K; g.ldb = N; g.ldc = N; g.transA = 0; g.transB = 0;
        g.alpha = 1.f; g.beta = 0.f; g.ksplit = 1; g.kchunk = K; g.part = nullptr; g.batch = 1; g.sA = g.sB = g.sC = 0;
        float* C3; CK(hipMalloc(&C3, (size_t)M3 * N * 4)); g.C = C3;
        dim3 grid(N / 128, M3 / 128, 1);
        const double a = time_us([&] { hipLaunchKernelGGL(single_buffer_kernel, grid, dim3(256), 0, 0, g); });
        const double b = time_us([&] { hipLaunchKernelGGL((variant_kernel<0, 1>), grid, dim3(256), 0, 0, g); });
        const double c = time_us([&] { hipLaunchKernelGGL((lvsr_sgemm128_kernel<false, false, true>), grid, dim3(256), 0, 0, g); });
        printf("24576 x 512 x %4d (768 tiles): single LDS buffer %8.1f us %6.1f TFLOP/s | double buffer, loads behind %8.1f us %6.1f TFLOP/s | product %8.1f us %6.1f TFLOP/s\n", K,
               a, 2.0 * M3 * N * K / a / 1e6, b, 2.0 * M3 * N * K / b / 1e6, c, 2.0 * M3 * N * K / c / 1e6);
        CK(hipFree(C3));
    }
    return 0;
}
