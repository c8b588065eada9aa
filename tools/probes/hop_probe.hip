// Standalone probe (not part of the product): what does ONE cluster-wide exchange of a 256-value phase vector cost on MI355X
// as a function of the instructions that publish and poll it, when the cluster's work-groups share an XCD (one L2)?
//   hipcc --offload-arch=gfx950 -O3 -o hop_probe hop_probe.hip && ./hop_probe
// Geometry of csrc/encoder_persist.hip: a cluster = P work-groups of 256 threads; per exchange every work-group publishes
// H/P {epoch,value} granules (8 bytes) and every thread gathers one of the H = 256 granules, re-polling until the epoch is
// there; values go to LDS, barrier, a short dependent computation produces the next values (so the chain is a true dependency).
// Two exchanges per "step" on two planes, as in the GRU step.  32 clusters run concurrently (the 2 x 16 chains of WSJ-base).
//   store kinds: 0 agent-scope relaxed atomic store (sc1 write-through; the product today)   1 plain volatile store
//                2 agent-scope atomic exchange, result unused (executes at the L2)            3 workgroup-scope atomic exchange
//   load kinds:  0 agent-scope relaxed atomic load (sc1; the product today)   1 returning atomic OR 0 (sc0 = return; agent and workgroup scope are the same encoding)
//                2 returning atomic OR 0, sc1 (system scope)
//                3 global_load_dwordx2 sc0 (asm)        4 global_load_dwordx2 sc0 sc1 (asm)     5 global_load_dwordx2 nt (asm)
// Every run checks the final values (a stale read gives a wrong sum) and reports aborts at the spin limit.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SPIN_LIMIT (1u << 18)

template <int SK>
__device__ __forceinline__ void gstore(u64* p, u64 w) {
    if (SK == 0) __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SK == 1) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(w) : "memory");      // really plain (a volatile store is emitted sc0 sc1)
    else if (SK == 2) (void)__hip_atomic_exchange(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else (void)__hip_atomic_exchange(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int LK>
__device__ __forceinline__ u64 gload(u64* p) {
    if (LK == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
        u64 v;
        const u64 zero = 0;      // (the compiler turns an idempotent fetch_or into a plain atomic LOAD: the RMW is spelled out)
        if (LK == 1) { asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory"); return v; }
        if (LK == 2) { asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory"); return v; }
        if (LK == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else if (LK == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}

template <int SK, int LK>
__global__ __launch_bounds__(256) void chain(u64* planes, int* abort_word, float* out, int P, int T, int spread, int nphase) {
    __shared__ float vec[256];
    __shared__ int lds_abort;
    const int tid = threadIdx.x, ncl = gridDim.x / P;
    int cl, p;
    if (!spread && ncl % 8 == 0) { cl = (blockIdx.x % 8) + 8 * (blockIdx.x / (8 * P)); p = (blockIdx.x / 8) % P; }   // block b on XCD b % 8
    else { cl = blockIdx.x / P; p = blockIdx.x % P; }
    if (tid == 0) lds_abort = 0;
    __syncthreads();
    const int per = 256 / P;
    float own = 0.001f * (float)(tid % per);
    for (int n = 0; n < T; ++n) {
#pragma unroll 1
        for (int ph = 0; ph < nphase; ++ph) {
            u64* g = planes + ((size_t)ph * ncl + cl) * 256;
            const unsigned epoch = (unsigned)(n + 1);
            if (tid < per) gstore<SK>(g + p * per + tid, ((u64)epoch << 32) | (u64)__float_as_uint(own));
            unsigned spins = 0;
            u64 w;
            for (;;) {
                w = gload<LK>(g + tid);
                const bool ok = (unsigned)(w >> 32) == epoch;
                if (__syncthreads_and(ok ? 1 : 0)) break;
                if (++spins > SPIN_LIMIT) { if (tid == 0) { atomicExch(abort_word, 1); } return; }
                if ((spins & 1023) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
            vec[tid] = __uint_as_float((unsigned)w);
            __syncthreads();
            // dependent work: the next value of (p, j) = mean of 4 gathered values + 1
            const int j = tid % per;
            own = 0.25f * (vec[j] + vec[(j + 64) & 255] + vec[(j + 128) & 255] + vec[(j + 192) & 255]) + 1.0f;
            __syncthreads();
        }
    }
    if (tid < per) out[(size_t)cl * 256 + p * per + tid] = own;
}

template <int SK, int LK>
static void run(const char* name, int P, int grid, int T, int spread, int nphase) {
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
        hipLaunchKernelGGL((chain<SK, LK>), dim3(grid), dim3(256), 0, 0, planes, ab, out, P, T, spread, nphase);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&aborted, ab, 4, hipMemcpyDeviceToHost));
        if (aborted) break;
        if (ms < best) best = ms;
    }
    // reference chain on the host
    int bad = 0;
    if (!aborted) {
        const int per = 256 / P;
        float* h = (float*)malloc((size_t)ncl * 256 * 4);
        CK(hipMemcpy(h, out, (size_t)ncl * 256 * 4, hipMemcpyDeviceToHost));
        float v[256], nv[256];
        for (int i = 0; i < 256; ++i) v[i] = 0.001f * (float)(i % per);
        for (int n = 0; n < T * nphase; ++n) {
            for (int i = 0; i < 256; ++i) { const int j = i % per; nv[i] = 0.25f * (v[j] + v[(j + 64) & 255] + v[(j + 128) & 255] + v[(j + 192) & 255]) + 1.0f; }
            memcpy(v, nv, sizeof(v));
        }
        for (int c = 0; c < ncl; ++c) for (int i = 0; i < 256; ++i) if (h[c * 256 + i] != v[i]) ++bad;
        free(h);
    }
    if (aborted) printf("%-58s P=%d grid=%3d %s  ABORTED at the spin limit (stale reads)\n", name, P, grid, spread ? "spread " : "xcd    ");
    else printf("%-58s P=%d grid=%3d %s  %.3f us per exchange  %s\n", name, P, grid, spread ? "spread " : "xcd    ",
                1e3f * best / (float)(T * nphase), bad ? "WRONG VALUES" : "values ok");
    fflush(stdout);
    CK(hipFree(planes)); CK(hipFree(ab)); CK(hipFree(out));
}

int main() {
    const int T = 2000;
    for (int P = 4; P <= 8; P += 4) {
        const int grid = 32 * P > 256 ? 256 : 32 * P;
        printf("--- P = %d work-groups per cluster, %d clusters, 2 exchanges per step\n", P, grid / P);
        run<0, 0>("sc1 store, sc1 load (product)", P, grid, T, 0, 2);
        run<0, 0>("sc1 store, sc1 load (product)", P, grid, T, 1, 2);
        run<0, 1>("sc1 store, agent atomic-or poll", P, grid, T, 0, 2);
        run<0, 1>("sc1 store, agent atomic-or poll", P, grid, T, 1, 2);
        run<0, 2>("sc1 store, system atomic-or poll", P, grid, T, 0, 2);
        run<1, 1>("plain store, agent atomic-or poll", P, grid, T, 0, 2);
        run<1, 2>("plain store, system atomic-or poll", P, grid, T, 0, 2);
        run<1, 2>("plain store, system atomic-or poll", P, grid, T, 1, 2);
        run<2, 1>("agent atomic-exchange store, agent atomic-or poll", P, grid, T, 0, 2);
        run<2, 1>("agent atomic-exchange store, agent atomic-or poll", P, grid, T, 1, 2);
        run<3, 2>("wg atomic-exchange store, system atomic-or poll", P, grid, T, 0, 2);
        run<3, 2>("wg atomic-exchange store, system atomic-or poll", P, grid, T, 1, 2);
        run<2, 0>("agent atomic-exchange store, sc1 load", P, grid, T, 0, 2);
        run<1, 0>("plain store, sc1 load", P, grid, T, 0, 2);
        run<1, 3>("plain store, sc0 load", P, grid, T, 0, 2);
        run<0, 3>("sc1 store, sc0 load", P, grid, T, 0, 2);
        run<0, 4>("sc1 store, sc0 sc1 load", P, grid, T, 0, 2);
        run<1, 5>("plain store, nt load", P, grid, T, 0, 2);
    }
    printf("--- one cluster alone (P = 4)\n");
    run<0, 0>("sc1 store, sc1 load (product)", 4, 4, T, 1, 2);
    run<2, 1>("agent atomic-exchange store, agent atomic-or poll", 4, 4, T, 1, 2);
    run<1, 2>("plain store, system atomic-or poll", 4, 4, T, 1, 2);
    return 0;
}
