// Persistent BiGRU layer kernels (forward and BPTT) for gfx950: ONE launch per layer and pass instead of two launches per
// time step.
//
// The recurrence is latency bound (2 800 dependent steps per pass on WSJ-base); with step kernels every phase of a step
// pays a kernel boundary plus a cold start of the body (~3.4 us measured).  Here a cluster of C = ceil(H/16) work-groups
// per (direction, 16-utterance row tile) stays resident, one per CU: work-group p owns hidden units [16p, 16p+16), keeps
// the three 16-column weight tiles it needs (H x 48 floats: 48 KiB at H = 256, 96 KiB at H = 512) in LDS in MFMA operand
// order for the whole sequence, and the only per-step global traffic on the critical path is the exchange of the phase
// vector (h_t, then r*h_t; BPTT: dh_t, then d(r*h)_t: B x H floats) between the work-groups of the cluster:
//   * hand-off = 8-byte {epoch, value} granules written with relaxed agent-scope atomic stores (sc1, write-through) and
//     polled with relaxed agent-scope atomic loads (MI355X_MICROARCH.md "handoff-1to1": ~0.8-1.0 us per hop; no fences,
//     no flags; placement independent: nothing assumes which XCD a work-group runs on);
//   * epoch = step index + 1, buffers zeroed by a memset node before every launch (graph-replay safe);
//   * a phase vector is only overwritten after every consumer has read the previous one (the producer needs all of the
//     consumers' next-phase granules first), so single buffering is enough;
//   * every spin is bounded: a work-group that waits too long raises the abort word and all work-groups leave.
// Saved tensors (u, r, c, rh, y) go out with plain stores off the critical path.
#include "common.h"
#include "graph_cache.h"
#include "lvsr_hip.h"
#include <stdlib.h>

typedef unsigned long long u64;
typedef lvsr_bigru_fwd_args EncFwd;
typedef lvsr_bigru_bwd_args EncBwd0;

#define PERSIST_SPIN_LIMIT (1u << 21)
#define PERSIST_MAX_WG 240

__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Each lane reads its N consecutive granules until every valid one carries `epoch`.  Wave-uniform result.
template <int N>
__device__ __forceinline__ bool granule_poll(const u64* g, int nvalid, unsigned epoch, float (&v)[N], int* abort_word) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int x = 0; x < N; ++x) {
            if (x < nvalid) {
                const u64 w = __hip_atomic_load(g + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[x] = __uint_as_float((unsigned)w);
                ok = ok && ((unsigned)(w >> 32) == epoch);
            } else {
                v[x] = 0.f;
            }
        }
        if (__all(ok)) return true;
        ++spins;
        if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > PERSIST_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

struct PersistGeom { int C, rt, Kw, Kpad, RB; };
// RB = utterances per cluster (<= 16, the MFMA row tile).  Smaller clusters exchange proportionally smaller phase vectors
// (the hand-off cost grows with the polled volume), at the price of more work-groups holding a copy of the weight shard.
__host__ __device__ __forceinline__ PersistGeom persist_geom(int B, int H, int NQ, int RB) {
    PersistGeom g;
    g.RB = RB;
    g.C = (H + 15) / 16; g.rt = (B + RB - 1) / RB; g.Kw = 16 * NQ; g.Kpad = 4 * g.Kw;
    return g;
}
static int persist_nq(int H) {           // K slice per wave = 16*NQ >= ceil(H/4)
    const int need = (((H + 3) / 4) + 15) / 16;
    int nq = 1;
    while (nq < need) nq *= 2;
    return nq;
}

// sum the 4 per-wave partial 16x16 tiles staged in `red`; thread tid -> (row tid>>4, col tid&15)
__device__ __forceinline__ void stage_tile(float (*red)[16][17], f32x4 a0, f32x4 a1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = a0[r] + a1[r];
}
__device__ __forceinline__ float fold_tile(float (*red)[16][17]) {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    return ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
}

template <int NQ>
__device__ __forceinline__ void mfma_tile(f32x4& acc0, f32x4& acc1, const float (&av)[4 * NQ], const float4* __restrict__ bs) {
    // bs: this wave's packed operand of one tile: [q][lane] float4
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 b = bs[q * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 0], b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 1], b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 2], b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * q + 3], b.w, acc1, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int NQ>
__global__ __launch_bounds__(256) void enc_pfwd_kernel(EncFwd a, u64* gh_all, u64* grh_all, int* abort_word, int RB) {
    __shared__ float4 Bs[3 * 4 * NQ * 64];          // tiles: 0 gates-update, 1 gates-reset, 2 candidate
    __shared__ float red[2][4][16][17];
    const PersistGeom geo = persist_geom(a.B, a.H, NQ, RB);
    const int H = a.H, B = a.B, T = a.T, Kw = geo.Kw, Kpad = geo.Kpad;
    const int cl = blockIdx.x / geo.C, p = blockIdx.x % geo.C;
    const int dir = cl / geo.rt, b0 = (cl % geo.rt) * RB, nrows = min(RB, B - b0), j0 = p * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Whg = a.Whg_p[dir];      // in persistent mode these are the PLAIN (H,2H) / (H,H) weights
    const float* Whh = a.Whh_p[dir];
    for (int idx = threadIdx.x; idx < 3 * 4 * NQ * 64; idx += 256) {
        const int ln = idx & 63, q = (idx >> 6) % NQ, wv = ((idx >> 6) / NQ) & 3, tile = (idx >> 6) / NQ / 4;
        const int col = j0 + (ln & 15);
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = wv * Kw + (ln >> 4) * (Kw >> 2) + 4 * q + r;
            float x = 0.f;
            if (k < H && col < H) x = tile == 2 ? Whh[(size_t)k * H + col] : Whg[(size_t)k * 2 * H + tile * H + col];
            w[r] = x;
        }
        Bs[idx] = make_float4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const int eb = threadIdx.x >> 4, ej = threadIdx.x & 15, b = b0 + eb, j = j0 + ej;
    const bool valid = eb < nrows && j < H;
    float hown = valid ? a.h0[dir][j] : 0.f;
    u64* gh = gh_all + (size_t)cl * 16 * Kpad;
    u64* grh = grh_all + (size_t)cl * 16 * Kpad;
    const int i = lane & 15, kk = lane >> 4, kbase = wave * Kw + kk * (Kw >> 2);
    const int nvalid = i < nrows ? max(0, min(4 * NQ, H - kbase)) : 0;
    const float4* bsw = Bs + (size_t)wave * NQ * 64;
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? n : T - 1 - n;
        const size_t row = (size_t)t * B + b;
        float gin_u = 0.f, gin_r = 0.f, xin = 0.f, m = 1.f;
        if (valid) {
            const float* xr = a.xg + row * 6 * H + dir * 3 * H;
            xin = xr[j]; gin_u = xr[H + j]; gin_r = xr[2 * H + j];
            if (a.mask) m = a.mask[row];
        }
        float av[4 * NQ];
        if (n == 0) {
#pragma unroll
            for (int x = 0; x < 4 * NQ; ++x) av[x] = x < nvalid ? a.h0[dir][kbase + x] : 0.f;
        } else if (!granule_poll<4 * NQ>(gh + (size_t)i * Kpad + kbase, nvalid, (unsigned)n, av, abort_word)) {
            return;
        }
        f32x4 u0 = F32X4_ZERO, u1 = F32X4_ZERO, r0 = F32X4_ZERO, r1 = F32X4_ZERO;
        mfma_tile<NQ>(u0, u1, av, bsw + 0 * 4 * NQ * 64);
        mfma_tile<NQ>(r0, r1, av, bsw + 1 * 4 * NQ * 64);
        stage_tile(red[0], u0, u1);
        stage_tile(red[1], r0, r1);
        __syncthreads();
        const float uu = sigmoidf_(fold_tile(red[0]) + gin_u);
        const float rr = sigmoidf_(fold_tile(red[1]) + gin_r);
        const float rh = rr * hown;
        if (valid) {
            granule_store(grh + (size_t)eb * Kpad + j, (unsigned)(n + 1), rh);
            const size_t o = row * 2 * H + dir * H + j;
            a.u[o] = uu; a.r[o] = rr; a.rh[o] = rh;
        }
        if (!granule_poll<4 * NQ>(grh + (size_t)i * Kpad + kbase, nvalid, (unsigned)(n + 1), av, abort_word)) return;
        f32x4 c0 = F32X4_ZERO, c1 = F32X4_ZERO;
        mfma_tile<NQ>(c0, c1, av, bsw + 2 * 4 * NQ * 64);
        __syncthreads();                       // everyone is done reading red[0] of the gates phase
        stage_tile(red[0], c0, c1);
        __syncthreads();
        const float cand = tanhf(fold_tile(red[0]) + xin);
        float hn = cand * uu + hown * (1.f - uu);
        hn = m * hn + (1.f - m) * hown;
        if (valid) {
            granule_store(gh + (size_t)eb * Kpad + j, (unsigned)(n + 1), hn);
            const size_t o = row * 2 * H + dir * H + j;
            a.c[o] = cand; a.y[o] = hn;
            if (a.ysub && (t % a.sub) == 0) a.ysub[((size_t)(t / a.sub) * B + b) * 2 * H + dir * H + j] = hn;
        }
        hown = hn;
        __syncthreads();                       // red[] is reused by the next step's gates phase
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT); same math as enc_bwd_a/b in encoder.hip
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pb_dy_at(const EncBwd0& a, int t, int b, int dir, int j) {
    if (t < 0 || t >= a.T || (t % a.sub) != 0) return 0.f;
    return a.dy[((size_t)(t / a.sub) * a.B + b) * 2 * a.H + dir * a.H + j];
}

template <int NQ>
__global__ __launch_bounds__(256) void enc_pbwd_kernel(EncBwd0 a, u64* gdh_all, u64* gdrh_all, int* abort_word, float* dh_out,
                                                       int Bp, int RB) {
    __shared__ float4 Bs[3 * 4 * NQ * 64];          // tiles: 0 Whh^T, 1 Whg^T (update rows), 2 Whg^T (reset rows)
    __shared__ float red[4][16][17];
    const PersistGeom geo = persist_geom(a.B, a.H, NQ, RB);
    const int H = a.H, B = a.B, T = a.T, Kw = geo.Kw, Kpad = geo.Kpad;
    const int cl = blockIdx.x / geo.C, p = blockIdx.x % geo.C;
    const int dir = cl / geo.rt, b0 = (cl % geo.rt) * RB, nrows = min(RB, B - b0), j0 = p * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Whg = a.WhgT_p[dir];     // persistent mode: PLAIN (H,2H) / (H,H) weights, transposed on the fly
    const float* Whh = a.WhhT_p[dir];
    for (int idx = threadIdx.x; idx < 3 * 4 * NQ * 64; idx += 256) {
        const int ln = idx & 63, q = (idx >> 6) % NQ, wv = ((idx >> 6) / NQ) & 3, tile = (idx >> 6) / NQ / 4;
        const int col = j0 + (ln & 15);          // output unit
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = wv * Kw + (ln >> 4) * (Kw >> 2) + 4 * q + r;     // contracted unit
            float x = 0.f;
            if (k < H && col < H) x = tile == 0 ? Whh[(size_t)col * H + k] : Whg[(size_t)col * 2 * H + (tile - 1) * H + k];
            w[r] = x;
        }
        Bs[idx] = make_float4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const int eb = threadIdx.x >> 4, ej = threadIdx.x & 15, b = b0 + eb, j = j0 + ej;
    const bool valid = eb < nrows && j < H;
    u64* gdh = gdh_all + (size_t)cl * 16 * Kpad;
    u64* gdrh = gdrh_all + (size_t)cl * 16 * Kpad;
    const int i = lane & 15, kk = lane >> 4, kbase = wave * Kw + kk * (Kw >> 2);
    const int nvalid = i < nrows ? max(0, min(4 * NQ, H - kbase)) : 0;
    const float4* bsw = Bs + (size_t)wave * NQ * 64;
    const int t_first = dir == 0 ? T - 1 : 0;
    float dhown = valid ? pb_dy_at(a, t_first, b, dir, j) : 0.f;
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? T - 1 - n : n, tp = dir == 0 ? t - 1 : t + 1;
        const bool first = tp < 0 || tp >= T;                 // previous state in scan order is the initial state
        // operands that do not depend on the recurrence: this lane's K range of row i ...
        float pu[4 * NQ], pc[4 * NQ], pr[4 * NQ], ph[4 * NQ];
        const float mi = (i < nrows && a.mask) ? a.mask[(size_t)t * B + b0 + i] : 1.f;
        {
            const size_t ro = ((size_t)t * B + b0 + i) * 2 * H + dir * H + kbase;
            const float* hp = first ? a.h0[dir] + kbase : a.y + ((size_t)tp * B + b0 + i) * 2 * H + dir * H + kbase;
#pragma unroll
            for (int x = 0; x < 4 * NQ; ++x) {
                const bool okx = x < nvalid;
                pu[x] = okx ? a.u[ro + x] : 0.f;
                pc[x] = okx ? a.c[ro + x] : 0.f;
                pr[x] = okx ? a.r[ro + x] : 0.f;
                ph[x] = okx ? hp[x] : 0.f;
            }
        }
        // ... and this thread's own element
        float uu = 0.f, rr = 0.f, cc = 0.f, hprev = 0.f, m = 1.f, dyp = 0.f;
        const size_t o = ((size_t)t * B + b) * 2 * H + dir * H + j;
        if (valid) {
            uu = a.u[o]; rr = a.r[o]; cc = a.c[o];
            hprev = first ? a.h0[dir][j] : a.y[((size_t)tp * B + b) * 2 * H + dir * H + j];
            if (a.mask) m = a.mask[(size_t)t * B + b];
            dyp = pb_dy_at(a, tp, b, dir, j);
        }
        float av[4 * NQ];
        if (n == 0) {
#pragma unroll
            for (int x = 0; x < 4 * NQ; ++x) av[x] = x < nvalid ? pb_dy_at(a, t_first, b0 + i, dir, kbase + x) : 0.f;
        } else if (!granule_poll<4 * NQ>(gdh + (size_t)i * Kpad + kbase, nvalid, (unsigned)n, av, abort_word)) {
            return;
        }
        float dpu[4 * NQ];
#pragma unroll
        for (int x = 0; x < 4 * NQ; ++x) {
            const float dhn = av[x] * mi;
            dpu[x] = dhn * (pc[x] - ph[x]) * pu[x] * (1.f - pu[x]);
            av[x] = dhn * pu[x] * (1.f - pc[x] * pc[x]);                      // dpc
        }
        f32x4 a0 = F32X4_ZERO, a1 = F32X4_ZERO;
        mfma_tile<NQ>(a0, a1, av, bsw + 0 * 4 * NQ * 64);
        stage_tile(red, a0, a1);
        __syncthreads();
        const float drh = fold_tile(red);
        const float dhn = m * dhown;
        float part = 0.f;
        if (valid) {
            granule_store(gdrh + (size_t)eb * Kpad + j, (unsigned)(n + 1), drh);
            float* dx = a.dxg + ((size_t)t * B + b) * 6 * H + dir * 3 * H;
            dx[j] = dhn * uu * (1.f - cc * cc);
            dx[H + j] = dhn * (cc - hprev) * uu * (1.f - uu);
            dx[2 * H + j] = drh * hprev * rr * (1.f - rr);
            part = dhn * (1.f - uu) + (1.f - m) * dhown + drh * rr + dyp;
        }
        if (!granule_poll<4 * NQ>(gdrh + (size_t)i * Kpad + kbase, nvalid, (unsigned)(n + 1), av, abort_word)) return;
#pragma unroll
        for (int x = 0; x < 4 * NQ; ++x) av[x] = av[x] * ph[x] * pr[x] * (1.f - pr[x]);      // dpr
        f32x4 d0 = F32X4_ZERO, d1 = F32X4_ZERO;
        mfma_tile<NQ>(d0, d1, dpu, bsw + 1 * 4 * NQ * 64);
        mfma_tile<NQ>(d0, d1, av, bsw + 2 * 4 * NQ * 64);
        __syncthreads();
        stage_tile(red, d0, d1);
        __syncthreads();
        const float dhp = part + fold_tile(red);
        if (valid) granule_store(gdh + (size_t)eb * Kpad + j, (unsigned)(n + 1), dhp);
        dhown = dhp;
        __syncthreads();
    }
    if (valid) dh_out[((size_t)dir * Bp + b) * H + j] = dhown;       // gradient wrt the initial state, per utterance
}

// d initial_state[dir][j] = sum_b dh[dir][b][j]
__global__ __launch_bounds__(256) void enc_pbwd_h0_kernel(const float* dh, int Bp, int B, int H, float* out_f, float* out_b) {
    const int dir = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dh[((size_t)dir * Bp + b) * H + j];
    (dir == 0 ? out_f : out_b)[j] = s;
}

// utterances per cluster: the smallest of {16,8,4,2,1}-row clusters whose work-groups all fit the chip (one per CU)
static int persist_rows(int B, int H) {
    const int C = (H + 15) / 16;
    if (H > 512) return 0;
    const char* env = getenv("LVSR_PERSIST_ROWS");
    int best = 0;
    for (int rb = 16; rb >= 1; rb /= 2) {
        const int rt = (B + rb - 1) / rb;
        if (2 * rt * C <= PERSIST_MAX_WG) best = rb;
        else break;
    }
    if (env && best) {
        const int want = atoi(env);
        if (want >= best && want <= 16 && (want & (want - 1)) == 0) best = want;
    }
    return best;
}
static bool persist_fits(int B, int H) { return persist_rows(B, H) > 0; }

extern "C" long long lvsr_bigru_persist_ws_bytes(int B, int H) {
    if (B <= 0 || H <= 0 || !persist_fits(B, H)) return 0;
    const int NQ = persist_nq(H);
    const PersistGeom g = persist_geom(B, H, NQ, 1);                   // worst case: one utterance per cluster
    return 256 + (long long)2 * (2 * g.rt) * 16 * g.Kpad * 8;          // abort word + two granule planes
}

template <int NQ>
static void launch_fwd(hipStream_t s, const EncFwd& a, u64* g0, u64* g1, int* ab, int RB) {
    const PersistGeom g = persist_geom(a.B, a.H, NQ, RB);
    hipLaunchKernelGGL(enc_pfwd_kernel<NQ>, dim3(2 * g.rt * g.C), dim3(256), 0, s, a, g0, g1, ab, RB);
}
template <int NQ>
static void launch_bwd(hipStream_t s, const EncBwd0& a, u64* g0, u64* g1, int* ab, float* dh, int Bp, int RB) {
    const PersistGeom g = persist_geom(a.B, a.H, NQ, RB);
    hipLaunchKernelGGL(enc_pbwd_kernel<NQ>, dim3(2 * g.rt * g.C), dim3(256), 0, s, a, g0, g1, ab, dh, Bp, RB);
}

int lvsr_bigru_fwd_persistent(hipStream_t s, const EncFwd& a0, int use_graph) {
    EncFwd a = a0;
    LVSR_REQUIRE(persist_fits(a.B, a.H) && a.sync_ws, "lvsr_bigru_fwd: persistent mode not available for B=%d H=%d", a.B, a.H);
    const int NQ = persist_nq(a.H), RB = persist_rows(a.B, a.H);
    const PersistGeom g = persist_geom(a.B, a.H, NQ, RB);
    const size_t plane = (size_t)(2 * g.rt) * 16 * g.Kpad;
    int* ab = (int*)a.sync_ws;
    u64* g0 = (u64*)((char*)a.sync_ws + 256);
    u64* g1 = g0 + plane;
    if (a.sub == 1) a.ysub = nullptr;
    auto enqueue = [&]() {
        (void)hipMemsetAsync(a.sync_ws, 0, 256 + 2 * plane * 8, s);
        switch (NQ) {
            case 1: launch_fwd<1>(s, a, g0, g1, ab, RB); break;
            case 2: launch_fwd<2>(s, a, g0, g1, ab, RB); break;
            case 4: launch_fwd<4>(s, a, g0, g1, ab, RB); break;
            default: launch_fwd<8>(s, a, g0, g1, ab, RB); break;
        }
    };
    GraphKey key("bigru_pfwd");
    key.add(&a, sizeof(a));
    key.add(&RB, sizeof(RB));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_fwd(persistent)");
}

int lvsr_bigru_bwd_persistent(hipStream_t s, const EncBwd0& a, int use_graph) {
    LVSR_REQUIRE(persist_fits(a.B, a.H) && a.sync_ws, "lvsr_bigru_bwd: persistent mode not available for B=%d H=%d", a.B, a.H);
    const int NQ = persist_nq(a.H), RB = persist_rows(a.B, a.H);
    const PersistGeom g = persist_geom(a.B, a.H, NQ, RB);
    const size_t plane = (size_t)(2 * g.rt) * 16 * g.Kpad;
    int* ab = (int*)a.sync_ws;
    u64* g0 = (u64*)((char*)a.sync_ws + 256);
    u64* g1 = g0 + plane;
    const int Bp = ((a.B + 15) / 16) * 16;
    float* dh = a.dh_ws;
    auto enqueue = [&]() {
        (void)hipMemsetAsync(a.sync_ws, 0, 256 + 2 * plane * 8, s);
        switch (NQ) {
            case 1: launch_bwd<1>(s, a, g0, g1, ab, dh, Bp, RB); break;
            case 2: launch_bwd<2>(s, a, g0, g1, ab, dh, Bp, RB); break;
            case 4: launch_bwd<4>(s, a, g0, g1, ab, dh, Bp, RB); break;
            default: launch_bwd<8>(s, a, g0, g1, ab, dh, Bp, RB); break;
        }
        hipLaunchKernelGGL(enc_pbwd_h0_kernel, dim3((a.H + 255) / 256, 1, 2), dim3(256), 0, s, dh, Bp, a.B, a.H, a.dh0[0], a.dh0[1]);
    };
    GraphKey key("bigru_pbwd");
    key.add(&a, sizeof(a));
    key.add(&RB, sizeof(RB));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_bwd(persistent)");
}
