// Standalone probe (not part of the product): what does the all-to-all exchange of a recurrent phase vector cost INSIDE one
// persistent launch on MI355X, as a function of how it is published and swept?  The product question: can a persistent
// BiGRU layer (2 exchanges per time step) beat 2 dependent kernel launches per time step (measured 6.2-6.8 us per step)?
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip && ./handoff_probe
// Geometry as in csrc/encoder_persist.hip: a cluster = C = H/16 work-groups (256 threads) that exchange an (RB x H) fp32
// vector as 8-byte {epoch, value} granules (sc1 stores, sc1 loads, no fences); clusters are independent (direction x row
// group).  Per "step" every work-group publishes its 16 columns and gathers all H columns, twice (two planes).
//   style 0: every lane polls the 16 granules it will feed to its MFMAs (4 waves poll; encoder_persist.hip today)
//   style 1: wave 0 sweeps the whole vector, coalesced (lane-contiguous granules), stages values in LDS; barrier
//   style 2: as 1 but every wave sweeps a quarter
//   style 3: as 1 with 16-byte sc1 loads (two granules per lane per load)
// work = 0: exchange only; 1: + MFMA/LDS-fold/sigmoid-like body between gather and publish (16 MFMA 16x16x4 per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SPIN_LIMIT (1u << 20)

__device__ __forceinline__ void gstore(u64* p, unsigned epoch, float v, int plain) {
    const u64 w = ((u64)epoch << 32) | (u64)__float_as_uint(v);
    if (plain) *(volatile u64*)p = w;          // stays in this XCD's L2: only valid when every reader shares the XCD
    else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gload(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int STYLE, int WORK, int SLEEP>
__global__ __launch_bounds__(256) void persist(u64* planes, int* abort_word, float* out, int H, int RB, int T, int remap, int plain) {
    __shared__ float vec[16][516];
    __shared__ float red[4][16][17];
    __shared__ int lds_abort;
    if (threadIdx.x == 0) lds_abort = 0;
    __syncthreads();
    const int C = H / 16, ncl = gridDim.x / C;
    // remap: block b runs on XCD b % 8 -> with 8 clusters every cluster sits on one XCD
    const int cl = remap ? blockIdx.x % ncl : blockIdx.x / C, p = remap ? blockIdx.x / ncl : blockIdx.x % C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)gridDim.x / C * 16 * H;
    u64* g0 = planes + (size_t)cl * 16 * H;
    u64* g1 = g0 + plane;
    const int eb = threadIdx.x >> 4, ej = threadIdx.x & 15;
    const bool valid = eb < RB;
    const int i = lane & 15, kk = lane >> 4, Kw = H / 4, kbase = wave * Kw + kk * (Kw / 4);
    const int NQ4 = Kw / 4;                     // granules per lane in style 0 (16 at H = 256)
    float own = 1.0f + 0.001f * threadIdx.x;
    float av[32];
    for (int n = 0; n < T; ++n) {
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            u64* g = ph ? g1 : g0;
            const unsigned epoch = (unsigned)(n + 1);
            if (valid) gstore(g + (size_t)eb * H + p * 16 + ej, epoch, own, plain);
            // ---- gather
            unsigned spins = 0;
            if (STYLE == 0) {
                for (;;) {
                    bool ok = true;
                    if (i < RB) {
#pragma unroll
                        for (int x = 0; x < 16; ++x) {
                            if (x < NQ4) {
                                const u64 w = gload(g + (size_t)i * H + kbase + x);
                                av[x] = __uint_as_float((unsigned)w);
                                ok = ok && ((unsigned)(w >> 32) == epoch);
                            }
                        }
                    }
                    if (__all(ok)) break;
                    if (++spins > SPIN_LIMIT) { *abort_word = 1; return; }
                    if (SLEEP) __builtin_amdgcn_s_sleep(1);
                }
            } else {
                const int total = RB * H;                               // granules of the vector
                const int nw = STYLE == 2 ? 4 : 1;                      // sweeping waves
                if (wave < nw) {
                    const int per = total / nw, base = wave * per;
                    if (STYLE == 3) {
                        for (;;) {
                            bool ok = true;
                            for (int j = lane * 2; j < per; j += 128) {
                                const u64* q = g + base + j;
                                __uint128_t t128;
                                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(t128) : "v"(q) : "memory");
                                const u64 w0 = (u64)t128, w1 = (u64)(t128 >> 64);
                                ok = ok && ((unsigned)(w0 >> 32) == epoch) && ((unsigned)(w1 >> 32) == epoch);
                                const int r0 = (base + j) / H, c0 = (base + j) % H;
                                vec[r0][c0] = __uint_as_float((unsigned)w0);
                                vec[r0][c0 + 1] = __uint_as_float((unsigned)w1);
                            }
                            if (__all(ok)) break;
                            if (++spins > SPIN_LIMIT) { *abort_word = 1; lds_abort = 1; break; }
                            if (SLEEP) __builtin_amdgcn_s_sleep(1);
                        }
                    } else {
                        for (;;) {
                            bool ok = true;
                            u64 w[16];
                            const int cnt = per / 64;                    // loads per lane (<= 16 per pass chunk)
                            for (int c0 = 0; c0 < cnt; c0 += 16) {
#pragma unroll
                                for (int x = 0; x < 16; ++x)
                                    if (c0 + x < cnt) w[x] = gload(g + base + (c0 + x) * 64 + lane);
#pragma unroll
                                for (int x = 0; x < 16; ++x)
                                    if (c0 + x < cnt) {
                                        ok = ok && ((unsigned)(w[x] >> 32) == epoch);
                                        const int idx = base + (c0 + x) * 64 + lane;
                                        vec[idx / H][idx % H] = __uint_as_float((unsigned)w[x]);
                                    }
                            }
                            if (__all(ok)) break;
                            if (++spins > SPIN_LIMIT) { *abort_word = 1; lds_abort = 1; break; }
                            if (SLEEP) __builtin_amdgcn_s_sleep(1);
                        }
                    }
                }
                __syncthreads();
                if (lds_abort) return;
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    if (x < NQ4) av[x] = i < RB ? vec[i][kbase + x] : 0.f;
            }
            // ---- body
            if (WORK) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
                for (int x = 0; x < 16; x += 2) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x], 0.01f * (x + 1), a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[x + 1], 0.02f, a1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = a0[r] + a1[r];
                __syncthreads();
                const float s = red[0][eb][ej] + red[1][eb][ej] + red[2][eb][ej] + red[3][eb][ej];
                own = 1.0f / (1.0f + __expf(-1e-3f * s));
                __syncthreads();
            } else {
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < 16; ++x)
                    if (x < NQ4) s += av[x];
                own = own * 0.5f + 1e-6f * s;
                if (STYLE != 0) __syncthreads();                        // vec[] is reused by the next gather
            }
        }
    }
    if (valid) out[(size_t)blockIdx.x * 256 + threadIdx.x] = own;
}

template <int STYLE, int WORK, int SLEEP>
static void run(const char* name, int H, int RB, int B, int T, u64* planes, int* ab, float* out, size_t bytes, int remap = 0, int plain = 0) {
    const int C = H / 16, clusters = 2 * ((B + RB - 1) / RB), grid = clusters * C;
    if (grid > 240) { printf("%-44s skipped (grid %d)\n", name, grid); return; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(planes, 0, bytes, 0));
        CK(hipMemsetAsync(ab, 0, 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((persist<STYLE, WORK, SLEEP>), dim3(grid), dim3(256), 0, 0, planes, ab, out, H, RB, T, remap, plain);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    int abw = 0;
    CK(hipMemcpy(&abw, ab, 4, hipMemcpyDeviceToHost));
    printf("%-44s H=%d RB=%2d grid=%3d  %.2f us/step (%.2f per exchange)%s\n", name, H, RB, grid, best * 1e3f / T, best * 1e3f / T / 2,
           abw ? "  ABORTED (spin limit)" : "");
    fflush(stdout);
}

int main() {
    const int T = 2000, B = 16;
    const size_t bytes = (size_t)2 * 64 * 16 * 512 * 8;
    u64* planes; int* ab; float* out;
    CK(hipMalloc(&planes, bytes)); CK(hipMalloc(&ab, 256)); CK(hipMalloc(&out, 240 * 256 * 4));
    for (int H = 256; H <= 256; H *= 2) {
        for (int RB = 16; RB >= 4; RB /= 2) {
            run<0, 0, 1>("style0 lane-own granules, sleep", H, RB, B, T, planes, ab, out, bytes);
            run<0, 0, 0>("style0 lane-own granules, no sleep", H, RB, B, T, planes, ab, out, bytes);
            run<1, 0, 0>("style1 wave0 coalesced sweep -> LDS", H, RB, B, T, planes, ab, out, bytes);
            run<2, 0, 0>("style2 4 waves sweep a quarter -> LDS", H, RB, B, T, planes, ab, out, bytes);
            run<3, 0, 0>("style3 wave0 16-B loads -> LDS", H, RB, B, T, planes, ab, out, bytes);
            run<0, 1, 0>("style0 + MFMA body", H, RB, B, T, planes, ab, out, bytes);
            run<2, 1, 0>("style2 + MFMA body", H, RB, B, T, planes, ab, out, bytes);
            if (RB == 4) {          // 8 clusters: one per XCD with the remap
                run<0, 0, 0>("style0 XCD-local clusters, sc1 stores", H, RB, B, T, planes, ab, out, bytes, 1, 0);
                run<0, 0, 0>("style0 XCD-local clusters, PLAIN stores", H, RB, B, T, planes, ab, out, bytes, 1, 1);
                run<2, 0, 0>("style2 XCD-local clusters, PLAIN stores", H, RB, B, T, planes, ab, out, bytes, 1, 1);
                run<0, 1, 0>("style0 XCD-local, PLAIN stores + MFMA body", H, RB, B, T, planes, ab, out, bytes, 1, 1);
            }
        }
    }
    return 0;
}
