// Standalone probe (not part of the product): which XCD does block b of a launch run on, and how fast is a hand-off through that XCD's L2?
//   hipcc --offload-arch=gfx950 -O3 -o xcd_map_probe xcd_map_probe.hip && ./xcd_map_probe
// The cluster kernels number their work-groups so that block b lands on XCD b % 8 (csrc/persist.h) — observed, not promised; a
// cluster whose members do not share an XCD falls back to write-through stores (correct, slower).  The boxes of the pool fall into two
// classes (13.3 vs 14.0 ms per WSJ-base step with identical GEMM times): this prints, for grids of 256 x 512 threads launched eagerly,
// behind a launch of an odd number of blocks, and from a hipGraph, how many blocks sit on XCD (b + s) % 8 for the best shift s, the
// per-XCD block counts, the device's clocks, and the time of a ping-pong between two blocks of one XCD / of two XCDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(512) void where(int* xcc_of_block) {
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u);
    // stay a little so that the whole grid is resident at once, as the cluster kernels' grids are
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
__global__ void odd(int* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1; }

// two blocks bounce an epoch T times through a granule each; `plain`: stores without cache-policy bits (visible in the XCD's L2)
__global__ __launch_bounds__(64) void pingpong(u64* g, int* xcc, int a, int b, int T, int plain, long long* cycles) {
    const int me = blockIdx.x == a ? 0 : blockIdx.x == b ? 1 : -1;
    if (threadIdx.x == 0 && me >= 0) xcc[me] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u);
    if (me < 0 || threadIdx.x != 0) return;
    u64* mine = g + 16 * me;
    u64* theirs = g + 16 * (1 - me);
    const long long t0 = wall_clock64();
    for (int n = 1; n <= T; ++n) {
        if (me == 0) {
            if (plain) __hip_atomic_store(mine, (u64)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else __hip_atomic_store(mine, (u64)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64)n && ++spins < (1u << 22)) {}
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64)n && ++spins < (1u << 22)) {}
            if (plain) __hip_atomic_store(mine, (u64)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else __hip_atomic_store(mine, (u64)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) cycles[0] = wall_clock64() - t0;
}

static void report(const char* what, const int* x, int n) {
    int best = 0, bests = 0, cnt[16] = {0};
    for (int s = 0; s < 8; ++s) {
        int ok = 0;
        for (int b = 0; b < n; ++b) ok += x[b] == (b + s) % 8;
        if (ok > best) { best = ok; bests = s; }
    }
    for (int b = 0; b < n; ++b) cnt[x[b] & 15]++;
    // clusters as the product forms them: blocks with the same b % 8 in runs of 8 * P
    int whole8 = 0, whole16 = 0;
    for (int P = 8; P <= 16; P += 8)
        for (int c0 = 0; c0 + 8 * P <= n; c0 += 8 * P)
            for (int col = 0; col < 8; ++col) {
                bool same = true;
                for (int p = 1; p < P; ++p) same = same && x[c0 + col + 8 * p] == x[c0 + col];
                (P == 8 ? whole8 : whole16) += same;
            }
    printf("%-44s %3d of %3d blocks on XCD (b + %d) %% 8; per XCD", what, best, n, bests);
    for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
    printf("; clusters on one XCD: %d of %d (P = 8), %d of %d (P = 16)\n", whole8, n / 8, whole16, n / 16);
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    int wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("%s: %d CUs, clock %d MHz, memory clock %d MHz, wall clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate / 1000, pr.memoryClockRate / 1000, wall_khz);
    const int n = 256;
    int *d, *sink, h[512];
    CK(hipMalloc(&d, 512 * sizeof(int)));
    CK(hipMalloc(&sink, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(where, dim3(n), dim3(512), 0, s, d);
        CK(hipMemcpyAsync(h, d, n * sizeof(int), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        report("eager launch", h, n);
    }
    for (int k = 1; k <= 5; k += 2) {
        hipLaunchKernelGGL(odd, dim3(k), dim3(64), 0, s, sink);
        hipLaunchKernelGGL(where, dim3(n), dim3(512), 0, s, d);
        CK(hipMemcpyAsync(h, d, n * sizeof(int), hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        char what[64];
        snprintf(what, sizeof what, "behind a launch of %d blocks", k);
        report(what, h, n);
    }
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(odd, dim3(3), dim3(64), 0, s, sink);
        CK(hipMemsetAsync(sink, 0, 64, s));
        hipLaunchKernelGGL(where, dim3(n), dim3(512), 0, s, d);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipGraphLaunch(exec, s));
            CK(hipMemcpyAsync(h, d, n * sizeof(int), hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            report("from a hipGraph (behind 3 blocks + memset)", h, n);
        }
    }
    // ---- ping-pong
    u64* g;
    long long* cyc;
    int* xc;
    CK(hipMalloc(&g, 4096));
    CK(hipMalloc(&cyc, 8));
    CK(hipMalloc(&xc, 8));
    const int T = 20000;
    const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
    for (int plain = 1; plain >= 0; --plain)
        for (int q = 0; q < 3; ++q) {
            if (plain && q > 0) continue;          // plain stores only between blocks of one XCD
            CK(hipMemsetAsync(g, 0, 4096, s));
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, s, g, xc, pairs[q][0], pairs[q][1], T, plain, cyc);
            long long c;
            int x2[2];
            CK(hipMemcpyAsync(&c, cyc, 8, hipMemcpyDeviceToHost, s));
            CK(hipMemcpyAsync(x2, xc, 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            printf("ping-pong blocks %d <-> %d (XCD %d, %d), %s stores: %.3f us per round trip (%.3f per hand-off)\n", pairs[q][0], pairs[q][1], x2[0], x2[1],
                   plain ? "plain" : "write-through", (double)c / (wall_khz ? wall_khz * 1e-3 : 100.0) / T, (double)c / (wall_khz ? wall_khz * 1e-3 : 100.0) / T / 2);
        }
    return 0;
}
