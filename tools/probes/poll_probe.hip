// Standalone probe (not part of the product): does it pay to keep SEVERAL polls of a granule in flight?
//   hipcc --offload-arch=gfx950 -O3 -o poll_probe poll_probe.hip && ./poll_probe
// A member of a cluster publishes its granules and polls its partners'.  All members publish at nearly the same time, so the
// first poll of a member usually reaches the L2 just BEFORE a partner's store does and a whole second round trip is spent.
// With NP loads in flight, issued SLEEP * 64 cycles apart, the successful poll is at most one issue interval late.
// Loop shape of csrc/encoder_persist.hip (gather_plane): plain stores inside an XCD, sc1 loads, a wave agrees with a ballot,
// one barrier per exchange; 512 threads, 256 granules swept by waves 0-3, two exchanges per step, 32 clusters of 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SPIN_LIMIT (1u << 20)

__device__ __forceinline__ void gstore(u64* p, u64 w) { __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }   // plain
__device__ __forceinline__ u64 gload(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }          // sc1

template <int NP, int SLEEP>
__global__ __launch_bounds__(512) void chain(u64* planes, int* abort_word, float* out, int P, int T) {
    __shared__ float vec[256];
    const int tid = threadIdx.x, ncl = gridDim.x / P;
    const int cl = (blockIdx.x % 8) + 8 * (blockIdx.x / (8 * P)), p = (blockIdx.x / 8) % P;      // block b on XCD b % 8
    const int per = 256 / P;
    float own = 0.001f * (float)(tid % per);
    for (int n = 0; n < T; ++n) {
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            u64* g = planes + ((size_t)ph * ncl + cl) * 256;
            const unsigned epoch = (unsigned)(n + 1);
            if (tid < per) gstore(g + p * per + tid, ((u64)epoch << 32) | (u64)__float_as_uint(own));
            if (tid < 256) {
                const u64* src = g + tid;
                u64 w[NP];
                u64 got = 0;
                w[0] = gload(src);
#pragma unroll
                for (int k = 1; k < NP; ++k) { __builtin_amdgcn_s_sleep(SLEEP); w[k] = gload(src); }
                unsigned spins = 0;
                bool done = false;
                while (!done) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if (!done) {
                            if (__all((unsigned)(w[k] >> 32) == epoch)) { got = w[k]; done = true; }
                            else w[k] = gload(src);
                        }
                    }
                    if (++spins > SPIN_LIMIT) { if ((tid & 63) == 0) atomicExch(abort_word, 1); return; }
                }
                vec[tid] = __uint_as_float((unsigned)got);
            }
            __syncthreads();
            const int j = tid % per;
            own = 0.25f * (vec[j] + vec[(j + 64) & 255] + vec[(j + 128) & 255] + vec[(j + 192) & 255]) + 1.0f;
            __syncthreads();
        }
    }
    if (tid < per) out[(size_t)cl * 256 + p * per + tid] = own;
}

template <int NP, int SLEEP>
static void run(int P, int grid, int T) {
    u64* planes; int* ab; float* out;
    const int ncl = grid / P;
    CK(hipMalloc(&planes, (size_t)2 * ncl * 256 * 8));
    CK(hipMalloc(&ab, 4));
    CK(hipMalloc(&out, (size_t)ncl * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    int aborted = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(planes, 0, (size_t)2 * ncl * 256 * 8));
        CK(hipMemset(ab, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((chain<NP, SLEEP>), dim3(grid), dim3(512), 0, 0, planes, ab, out, P, T);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&aborted, ab, 4, hipMemcpyDeviceToHost));
        if (aborted) break;
        if (ms < best) best = ms;
    }
    int bad = 0;
    if (!aborted) {
        const int per = 256 / P;
        float* h = (float*)malloc((size_t)ncl * 256 * 4);
        CK(hipMemcpy(h, out, (size_t)ncl * 256 * 4, hipMemcpyDeviceToHost));
        float v[256], nv[256];
        for (int i = 0; i < 256; ++i) v[i] = 0.001f * (float)(i % per);
        for (int n = 0; n < T * 2; ++n) {
            for (int i = 0; i < 256; ++i) { const int j = i % per; nv[i] = 0.25f * (v[j] + v[(j + 64) & 255] + v[(j + 128) & 255] + v[(j + 192) & 255]) + 1.0f; }
            memcpy(v, nv, sizeof(v));
        }
        for (int c = 0; c < ncl; ++c) for (int i = 0; i < 256; ++i) if (h[c * 256 + i] != v[i]) ++bad;
        free(h);
    }
    if (aborted) printf("polls in flight %d, %4d cycles apart   P=%d grid=%3d  ABORTED\n", NP, SLEEP * 64, P, grid);
    else printf("polls in flight %d, %4d cycles apart   P=%d grid=%3d  %.3f us per exchange  %s\n", NP, SLEEP * 64, P, grid,
                1e3f * best / (float)(T * 2), bad ? "WRONG VALUES" : "values ok");
    fflush(stdout);
    CK(hipFree(planes)); CK(hipFree(ab)); CK(hipFree(out));
}

int main() {
    const int T = 2000, P = 8, grid = 256;
    run<1, 0>(P, grid, T);
    run<2, 1>(P, grid, T);
    run<2, 2>(P, grid, T);
    run<2, 4>(P, grid, T);
    run<2, 8>(P, grid, T);
    run<3, 1>(P, grid, T);
    run<3, 2>(P, grid, T);
    run<3, 4>(P, grid, T);
    run<4, 1>(P, grid, T);
    run<4, 2>(P, grid, T);
    run<4, 4>(P, grid, T);
    run<8, 1>(P, grid, T);
    return 0;
}
