// Persistent BiGRU layer kernels with ONE exchange per time step (128 < H <= 256, one utterance per cluster) — opt-in
// (LVSR_PERSIST_ONEHOP=1: clusters of 4 work-groups, =2: clusters of 8), built and parity-checked on the emulator, not yet timed
// on the MI355X.
//
// encoder_persist.hip splits the units of a (direction, utterance) over the work-groups of a cluster for ALL three gates, so a
// step has two dependent exchanges: h -> gates, then r*h -> candidate (BPTT: [dpre_c|dpre_u] -> d(r*h), then dpre_r -> dh).  Each
// costs ~0.85 us of the 2.1 us step, whatever the arithmetic.  Here the quantity in the middle is computed REDUNDANTLY by every
// work-group of the cluster:
//   forward:   every work-group holds the whole reset-gate block (H x H: 128 registers per thread at 512 threads) and forms
//              r and r*h for ALL units from the gathered h; the update gate and the candidate of its OWN units follow from LDS;
//   backward:  every work-group holds the whole state_to_state block and forms d(r*h) and dpre_r for ALL units from the gathered
//              [dpre_c | dpre_u]; the state gradient of its OWN units follows from LDS.
// One exchange per step is left (h forward, [dpre_c | dpre_u] backward); the price is 2.7 times the multiply-adds (80 instead of
// 48 v_pk_fma_f32 per thread and step with clusters of 8) and the whole register file for weights (160 / 192 of 256 registers per
// thread).  With a single exchange per step a producer may run one step ahead of a consumer, so the planes are double-buffered
// by step parity (a plane is rewritten two steps later, which needs every consumer's granules of the step in between).
//
// Thread roles (512 threads, both roles per thread):
//   (a) full block:  k-slice ksl = tid % 8 (32 inputs), unit group ug = tid / 8 (4 outputs): 128 weights; the eight slices of a
//       group are adjacent lanes and fold with DPP; lane ksl = e < 4 finishes unit 4 ug + e;
//   (b) own units:   k-slice q = tid % 16 (16 inputs), UB = UNITS / 32 own units jb * UB + e, two blocks: 32 UB weights; the
//       sixteen slices fold with DPP, every lane of the group keeps the result.
// LDS holds the gathered vector in both slice layouts ((a): 8 rows of 32 + 4, (b): 16 rows of 16 + 4 floats).
// LVSR_PERSIST_FLAGS & 64 (PF_STAGE): the operands of the next step that do not depend on the recurrence are fetched by waves
// 4..7 — which take no part in the sweeps — and handed over through LDS, instead of by every thread for itself.
#include "common.h"
#include "graph_cache.h"
#include "lvsr_hip.h"
#include "persist.h"
#include <stdlib.h>

typedef lvsr_bigru_fwd_args EncFwd;
typedef lvsr_bigru_bwd_args EncBwd0;

#define P1_HP 256
#define P1_NTH 512
#define P1_LA 36
#define P1_LB 20

__device__ __forceinline__ int p1_slot_a(int k) { return (k >> 5) * P1_LA + (k & 31); }
__device__ __forceinline__ int p1_slot_b(int k) { return (k >> 4) * P1_LB + (k & 15); }

// Wait for the 256 (NPL = 1) or 512 (NPL = 2: two planes back to back) granules of a step: thread tid takes granule tid.
// Returns false when the cluster gave up.
template <int NPL>
__device__ __forceinline__ bool p1_gather(const u64* g, unsigned epoch, int tid, int* abort_word, float& out, int flags) {
    const bool mine = tid < NPL * P1_HP;
    u64 w = (u64)epoch << 32;
    unsigned spins = 0;
    for (;;) {
        if (mine) w = __hip_atomic_load(g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all((unsigned)(w >> 32) == epoch) || (flags & PF_NOWAIT)) break;      // (PF_NOWAIT: timing ablation, wrong results)
        if (((++spins) & 127u) == 0u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > PERSIST_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    out = __uint_as_float((unsigned)w);
    return true;
}

// sum_x w[e][x] * v[x] over a 32-float slice for four outputs (role (a)), folded over the eight slices of the unit group
__device__ __forceinline__ void p1_dot_a(const f32x2 (&w)[4][16], const float* slice, float (&out)[4], int flags) {
    if (flags & PF_NODOT) {            // timing ablation (wrong results): what the step costs without the contractions
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = slice[e];
        return;
    }
    f32x2 acc[4][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e][0] = acc[e][1] = (f32x2){0.f, 0.f};
    const float4* hv = (const float4*)slice;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const float4 h4 = hv[x];
        const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e][0] = w[e][2 * x] * lo + acc[e][0];
            acc[e][1] = w[e][2 * x + 1] * hi + acc[e][1];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = group_sum<8>((acc[e][0].x + acc[e][1].x) + (acc[e][0].y + acc[e][1].y));
}

// sum_x w[x] * v[x] over a 16-float slice (role (b)), NOT folded
__device__ __forceinline__ float p1_dot_b(const f32x2 (&w)[8], const float* slice, int flags) {
    if (flags & PF_NODOT) return slice[0];
    const float4* hv = (const float4*)slice;
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const float4 h4 = hv[x];
        const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
        a0 = w[2 * x] * lo + a0;
        a1 = w[2 * x + 1] * hi + a1;
    }
    return (a0.x + a1.x) + (a0.y + a1.y);
}

// ---------------------------------------------------------------------------------------------------------------
// forward (math as enc_pfwd_kernel / lvsr/bricks GatedRecurrent: blocks/bricks/recurrent.py:608-620)
// ---------------------------------------------------------------------------------------------------------------
template <int UNITS>
__global__ __launch_bounds__(P1_NTH) void enc_p1fwd_kernel(EncFwd a, u64* planes, int* abort_word, int flags) {
    constexpr int P = P1_HP / UNITS, UB = UNITS / 32;
    __shared__ __attribute__((aligned(16))) float ha[8 * P1_LA], hb[16 * P1_LB], rhb[16 * P1_LB];
    __shared__ float nx_gr[P1_HP], nx_own[2 * UNITS], nx_m[4];          // PF_STAGE: the next step's operands, staged by waves 4..7
    const bool stage = (flags & PF_STAGE) != 0;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int H = a.H, B = a.B, T = a.T;
    int cl, p;
    cluster_of_block(P, flags, cl, p);
    const bool save = !(flags & PF_NOSAVE);
    const int dir = cl / B, b = cl % B;
    // ---- weights, in registers for the whole sequence (clamped addresses, zeroed afterwards: straight-line loads)
    f32x2 wrf[4][16], wu[UB][8], wc[UB][8];
    {
        const int tid = threadIdx.x, ksl = tid & 7, ug = tid >> 3, q = tid & 15, jb = tid >> 4;
        const float* Whg = a.Whg_p[dir];      // PLAIN (H,2H) / (H,H) weights
        const float* Whh = a.Whh_p[dir];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = 4 * ug + e;
            const size_t uc = (size_t)min(u, H - 1);
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = 32 * ksl + 2 * x + h;
                    v[h] = Whg[(size_t)min(k, H - 1) * 2 * H + H + uc] * ((u < H && k < H) ? 1.f : 0.f);
                }
                wrf[e][x] = (f32x2){v[0], v[1]};
            }
        }
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            const size_t jc = (size_t)min(j, H - 1);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float v[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = 16 * q + 2 * x + h;
                    const float keep = (j < H && k < H) ? 1.f : 0.f;
                    const size_t kc = (size_t)min(k, H - 1);
                    v[h] = Whg[kc * 2 * H + jc] * keep;
                    v[2 + h] = Whh[kc * H + jc] * keep;
                }
                wu[e][x] = (f32x2){v[0], v[1]}; wc[e][x] = (f32x2){v[2], v[3]};
            }
        }
    }
    u64* const gpl = planes + (size_t)cl * 2 * P1_HP;      // two planes, by step parity
    float hown[UB], n_xin[UB], n_gu[UB], n_m = 1.f, n_gr = 0.f;
    {
        const int tid = threadIdx.x, jb = tid >> 4, t0 = dir == 0 ? 0 : T - 1;
        const size_t row = (size_t)t0 * B + b;
        const float* xr = a.xg + row * 6 * H + dir * 3 * H;
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            hown[e] = j < H ? a.h0[dir][j] : 0.f;
            n_xin[e] = j < H ? xr[j] : 0.f;
            n_gu[e] = j < H ? xr[H + j] : 0.f;
        }
        if (a.mask) n_m = a.mask[row];
        const int ua = 4 * (tid >> 3) + (tid & 3);
        n_gr = ua < H ? xr[2 * H + ua] : 0.f;
        for (int k = tid; k < P1_HP; k += P1_NTH) {
            const float v = k < H ? a.h0[dir][k] : 0.f;
            ha[p1_slot_a(k)] = v; hb[p1_slot_b(k)] = v;
        }
    }
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? n : T - 1 - n;
        // per-step re-derivation of the thread indices (see lvsr_unhoisted): nothing index-shaped is kept across the steps
        const int tid = lvsr_unhoisted((int)threadIdx.x), ksl = tid & 7, ug = tid >> 3, q = tid & 15, jb = tid >> 4;
        const size_t orow = ((size_t)t * B + b) * 2 * H + (size_t)dir * H;
        // PF_STAGE: this step's operands come from LDS (written by waves 4..7 during the previous step); waves 4..7 — which never
        // poll — fetch the next step's now, so that no global load sits in front of a sweep in a polling wave's memory queue
        float st_gr = 0.f, st_x = 0.f, st_u = 0.f, st_m = 1.f;
        if (stage) {
            if (n > 0) {
                n_gr = nx_gr[4 * ug + (ksl & 3)];
#pragma unroll
                for (int e = 0; e < UB; ++e) { n_xin[e] = nx_own[jb * UB + e]; n_gu[e] = nx_own[UNITS + jb * UB + e]; }
                n_m = nx_m[0];
            }
            if (wave >= 4 && n + 1 < T) {
                const int x = tid - 256, tn = dir == 0 ? n + 1 : T - 2 - n, jx = p * UNITS + x;
                const size_t row = (size_t)tn * B + b;
                const float* xr = a.xg + row * 6 * H + dir * 3 * H;
                st_gr = x < H ? xr[2 * H + x] : 0.f;
                if (x < UNITS && jx < H) { st_x = xr[jx]; st_u = xr[H + jx]; }
                if (a.mask) st_m = a.mask[row];
            }
        }
        if (n > 0) {
            float v;
            if (!p1_gather<1>(gpl + (n & 1) * P1_HP, (unsigned)n, tid, abort_word, v, flags)) return;
            if (tid < P1_HP) { ha[p1_slot_a(tid)] = v; hb[p1_slot_b(tid)] = v; }
        }
        __syncthreads();
        // ---- (a) reset gate and r*h of ALL units
        {
            float s[4];
            p1_dot_a(wrf, ha + ksl * P1_LA, s, flags);
            const int e = ksl & 3, u = 4 * ug + e;
            const float mine = e == 0 ? s[0] : e == 1 ? s[1] : e == 2 ? s[2] : s[3];
            const float r = sigmoid_fast(mine + n_gr);
            const float rh = u < H ? r * ha[p1_slot_a(u)] : 0.f;
            if (ksl < 4) {
                rhb[p1_slot_b(u)] = rh;
                if (save && u < H && u / UNITS == p) { a.r[orow + u] = r; a.rh[orow + u] = rh; }
            }
        }
        // ---- (b) update gate of the own units
        float uu[UB];
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            uu[e] = sigmoid_fast(group_sum<16>(p1_dot_b(wu[e], hb + q * P1_LB, flags)) + n_gu[e]);
            if (save && q == 0 && j < H) a.u[orow + j] = uu[e];
        }
        if (stage && wave >= 4 && n + 1 < T) {          // every read of the staged operands of THIS step lies before the first barrier
            const int x = tid - 256;
            nx_gr[x] = st_gr;
            if (x < UNITS) { nx_own[x] = st_x; nx_own[UNITS + x] = st_u; }
            if (x == 0) nx_m[0] = st_m;
        }
        __syncthreads();
        // ---- (b) candidate, state update, mask blend; publish
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            const float cand = tanh_fast(group_sum<16>(p1_dot_b(wc[e], rhb + q * P1_LB, flags)) + n_xin[e]);
            float hn = cand * uu[e] + hown[e] * (1.f - uu[e]);
            hn = n_m * hn + (1.f - n_m) * hown[e];
            if (j >= H) hn = 0.f;
            // (the padded units publish their zeros too: the gather waits for all 256 granules)
            if (q == 0 && n + 1 < T) granule_store(gpl + ((n + 1) & 1) * P1_HP + j, (unsigned)(n + 1), hn, flags);
            if (q == 0 && j < H) {
                if (save) a.c[orow + j] = cand;
                a.y[orow + j] = hn;
                if (a.ysub && (t % a.sub) == 0) a.ysub[((size_t)(t / a.sub) * B + b) * 2 * H + (size_t)dir * H + j] = hn;
            }
            hown[e] = hn;
        }
        // ---- operands of the next step (independent of the recurrence): in flight during the hand-off
        if (!stage && n + 1 < T) {
            const int tn = dir == 0 ? n + 1 : T - 2 - n;
            const size_t row = (size_t)tn * B + b;
            const float* xr = a.xg + row * 6 * H + dir * 3 * H;
#pragma unroll
            for (int e = 0; e < UB; ++e) {
                const int j = p * UNITS + jb * UB + e;
                if (j < H) { n_xin[e] = xr[j]; n_gu[e] = xr[H + j]; }
            }
            if (a.mask) n_m = a.mask[row];
            const int ua = 4 * ug + (ksl & 3);
            if (ua < H) n_gr = xr[2 * H + ua];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT; math as enc_pbwd_kernel).  Per step (forward direction walks t = T-1..0, backward direction t = 0..T-1):
//   dhn = m*dh; dpre_c = dhn*u*(1-c^2); dpre_u = dhn*(c-h_prev)*u*(1-u)                      own units, published
//   drh = dpre_c @ Whh^T;  dpre_r = drh*h_prev*r*(1-r)                                        ALL units, every work-group
//   dh_prev = dhn*(1-u) + (1-m)*dh + drh*r + dpre_u @ Whg[:, :H]^T + dpre_r @ Whg[:, H:]^T + dy[t_prev]      own units
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float p1_dy_at(const EncBwd0& a, int t, int b, int dir, int j) {
    if (t < 0 || t >= a.T || (t % a.sub) != 0) return 0.f;
    return a.dy[((size_t)(t / a.sub) * a.B + b) * 2 * a.H + dir * a.H + j];
}

template <int UNITS>
__global__ __launch_bounds__(P1_NTH) void enc_p1bwd_kernel(EncBwd0 a, u64* planes, int* abort_word, float* dh_out, int Bp, int flags) {
    constexpr int P = P1_HP / UNITS, UB = UNITS / 32;
    __shared__ __attribute__((aligned(16))) float dca[8 * P1_LA], dub[16 * P1_LB], drb[16 * P1_LB], drr[P1_HP];
    __shared__ float nx_r[P1_HP], nx_hp[P1_HP], nx_own[3 * UNITS], nx_m[4];      // PF_STAGE: staged by waves 4..7 (u | c | dy of the own units)
    const bool stage = (flags & PF_STAGE) != 0;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int H = a.H, B = a.B, T = a.T;
    int cl, p;
    cluster_of_block(P, flags, cl, p);
    const bool save = !(flags & PF_NOSAVE);
    const int dir = cl / B, b = cl % B;
    // ---- weights: (a) ROWS 4 ug + e of state_to_state, column slice ksl; (b) rows of the own units of both gate blocks, slice q
    f32x2 waf[4][16], wbu[UB][8], wbr[UB][8];
    {
        const int tid = threadIdx.x, ksl = tid & 7, ug = tid >> 3, q = tid & 15, jb = tid >> 4;
        const float* Whg = a.WhgT_p[dir];     // PLAIN (H,2H) / (H,H) weights, rows read in place
        const float* Whh = a.WhhT_p[dir];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * ug + e;
            const size_t kc = (size_t)min(k, H - 1);
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 32 * ksl + 2 * x + h;
                    v[h] = Whh[kc * H + (size_t)min(j, H - 1)] * ((k < H && j < H) ? 1.f : 0.f);
                }
                waf[e][x] = (f32x2){v[0], v[1]};
            }
        }
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int k = p * UNITS + jb * UB + e;
            const size_t kc = (size_t)min(k, H - 1);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float v[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 16 * q + 2 * x + h;
                    const float keep = (k < H && j < H) ? 1.f : 0.f;
                    const size_t jc = (size_t)min(j, H - 1);
                    v[h] = Whg[kc * 2 * H + jc] * keep;
                    v[2 + h] = Whg[kc * 2 * H + H + jc] * keep;
                }
                wbu[e][x] = (f32x2){v[0], v[1]}; wbr[e][x] = (f32x2){v[2], v[3]};
            }
        }
    }
    u64* const gpl = planes + (size_t)cl * 4 * P1_HP;      // [parity][dpre_c | dpre_u]
    const int t_first = dir == 0 ? T - 1 : 0;
    float dh[UB], n_u[UB], n_c[UB], n_hp[UB], n_dy[UB], n_m = 1.f, n_ra = 0.f, n_hpa = 0.f;
    auto prefetch = [&](int t, int tid) {
        const int jb = tid >> 4, tp = dir == 0 ? t - 1 : t + 1;
        const size_t o = ((size_t)t * B + b) * 2 * H + (size_t)dir * H;
        const size_t op = ((size_t)tp * B + b) * 2 * H + (size_t)dir * H;
        const bool first = tp < 0 || tp >= T;
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            if (j < H) {
                n_u[e] = a.u[o + j]; n_c[e] = a.c[o + j];
                n_hp[e] = first ? a.h0[dir][j] : a.y[op + j];
                n_dy[e] = p1_dy_at(a, tp, b, dir, j);
            }
        }
        n_m = a.mask ? a.mask[(size_t)t * B + b] : 1.f;
        const int ka = 4 * (tid >> 3) + (tid & 3);
        if (ka < H) {
            n_ra = a.r[o + ka];
            n_hpa = first ? a.h0[dir][ka] : a.y[op + ka];
        }
    };
    {
        const int tid = threadIdx.x, jb = tid >> 4;
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            dh[e] = j < H ? p1_dy_at(a, t_first, b, dir, j) : 0.f;
            n_u[e] = n_c[e] = n_hp[e] = n_dy[e] = 0.f;
        }
        prefetch(t_first, tid);
    }
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? T - 1 - n : n;
        const int tid = lvsr_unhoisted((int)threadIdx.x), ksl = tid & 7, ug = tid >> 3, q = tid & 15, jb = tid >> 4;
        u64* const gc = gpl + (size_t)(n & 1) * 2 * P1_HP;
        float* const dx = a.dxg + ((size_t)t * B + b) * 6 * H + (size_t)dir * 3 * H;
        // PF_STAGE (see the forward kernel): this step's saved values from LDS, the next step's fetched by waves 4..7 now
        float st_r = 0.f, st_hp = 0.f, st_u = 0.f, st_c = 0.f, st_dy = 0.f, st_m = 1.f;
        if (stage) {
            if (n > 0) {
                const int ka = min(4 * ug + (ksl & 3), P1_HP - 1);
                n_ra = nx_r[ka]; n_hpa = nx_hp[ka];
#pragma unroll
                for (int e = 0; e < UB; ++e) {
                    const int jl = jb * UB + e;
                    n_u[e] = nx_own[jl]; n_c[e] = nx_own[UNITS + jl]; n_dy[e] = nx_own[2 * UNITS + jl];
                    n_hp[e] = nx_hp[p * UNITS + jl];
                }
                n_m = nx_m[0];
            }
            if (wave >= 4 && n + 1 < T) {
                const int x = tid - 256, jx = p * UNITS + x, tn = dir == 0 ? t - 1 : t + 1, tp = dir == 0 ? tn - 1 : tn + 1;
                const size_t o = ((size_t)tn * B + b) * 2 * H + (size_t)dir * H;
                const size_t op = ((size_t)tp * B + b) * 2 * H + (size_t)dir * H;
                const bool first = tp < 0 || tp >= T;
                if (x < H) {
                    st_r = a.r[o + x];
                    st_hp = first ? a.h0[dir][x] : a.y[op + x];
                }
                if (x < UNITS && jx < H) {
                    st_u = a.u[o + jx]; st_c = a.c[o + jx];
                    st_dy = p1_dy_at(a, tp, b, dir, jx);
                }
                st_m = a.mask ? a.mask[(size_t)tn * B + b] : 1.f;
            }
        }
        // ---- everything of this step that depends on dh elementwise only; publish dpre_c and dpre_u of the own units
        float part[UB], rr_a = n_ra, hp_a = n_hpa;
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            const float uu = n_u[e], cc = n_c[e], hp = n_hp[e];
            const float dhn = n_m * dh[e];
            const float dpc = j < H ? dhn * uu * (1.f - cc * cc) : 0.f;
            const float dpu = j < H ? dhn * (cc - hp) * uu * (1.f - uu) : 0.f;
            part[e] = dhn * (1.f - uu) + (1.f - n_m) * dh[e] + n_dy[e];
            if (q == 0 && j < P1_HP) {
                granule_store(gc + j, (unsigned)(n + 1), dpc, flags);
                granule_store(gc + P1_HP + j, (unsigned)(n + 1), dpu, flags);
                if (save && j < H) { dx[j] = dpc; dx[H + j] = dpu; }
            }
        }
        // operands of the next step: in flight during the hand-off
        if (!stage && n + 1 < T) prefetch(dir == 0 ? t - 1 : t + 1, tid);
        {
            float v;
            if (!p1_gather<2>(gc, (unsigned)(n + 1), tid, abort_word, v, flags)) return;
            if (tid < P1_HP) dca[p1_slot_a(tid)] = v;
            else dub[p1_slot_b(tid - P1_HP)] = v;
        }
        __syncthreads();
        // ---- (a) d(r*h) and dpre_r of ALL units
        {
            float s[4];
            p1_dot_a(waf, dca + ksl * P1_LA, s, flags);
            const int e = ksl & 3, k = 4 * ug + e;
            const float drh = e == 0 ? s[0] : e == 1 ? s[1] : e == 2 ? s[2] : s[3];
            const float dpr = k < H ? drh * hp_a * rr_a * (1.f - rr_a) : 0.f;
            if (ksl < 4) {
                drb[p1_slot_b(k)] = dpr;
                drr[k] = k < H ? drh * rr_a : 0.f;
                if (save && k < H && k / UNITS == p) dx[2 * H + k] = dpr;
            }
        }
        // ---- (b) dpre_u @ Whg[:, :H]^T of the own units (needs nothing of (a))
        float su[UB];
#pragma unroll
        for (int e = 0; e < UB; ++e) su[e] = p1_dot_b(wbu[e], dub + q * P1_LB, flags);
        if (stage && wave >= 4 && n + 1 < T) {          // every read of the staged values of THIS step lies before the first barrier
            const int x = tid - 256;
            nx_r[x] = st_r; nx_hp[x] = st_hp;
            if (x < UNITS) { nx_own[x] = st_u; nx_own[UNITS + x] = st_c; nx_own[2 * UNITS + x] = st_dy; }
            if (x == 0) nx_m[0] = st_m;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            const float sum = group_sum<16>(su[e] + p1_dot_b(wbr[e], drb + q * P1_LB, flags));
            dh[e] = j < H ? part[e] + drr[min(j, P1_HP - 1)] + sum : 0.f;
        }
    }
    {
        const int tid = threadIdx.x, q = tid & 15, jb = tid >> 4;
#pragma unroll
        for (int e = 0; e < UB; ++e) {
            const int j = p * UNITS + jb * UB + e;
            if (q == 0 && j < H) dh_out[((size_t)dir * Bp + b) * H + j] = dh[e];       // d initial state, per utterance
        }
    }
}

// 0 = not requested / not applicable; else the units per work-group (64: clusters of 4, 32: clusters of 8)
int lvsr_bigru_onehop_units(int B, int H) {
    const char* env = getenv("LVSR_PERSIST_ONEHOP");
    const int want = env ? atoi(env) : 0;
    if (want <= 0 || H <= 128 || H > P1_HP || B <= 0) return 0;
    if (want >= 2) return 2 * B * 8 <= 256 ? 32 : 0;       // the whole chip at B = 16: one work-group per CU
    return 2 * B * 4 <= PERSIST_MAX_WG ? 64 : 0;
}

void lvsr_bigru_onehop_fwd(hipStream_t s, const EncFwd& a, int units, u64* planes, int* ab, int flags) {
    if (units == 64) hipLaunchKernelGGL(enc_p1fwd_kernel<64>, dim3(2 * a.B * 4), dim3(P1_NTH), 0, s, a, planes, ab, flags);
    else hipLaunchKernelGGL(enc_p1fwd_kernel<32>, dim3(2 * a.B * 8), dim3(P1_NTH), 0, s, a, planes, ab, flags);
}

void lvsr_bigru_onehop_bwd(hipStream_t s, const EncBwd0& a, int units, u64* planes, int* ab, float* dh, int Bp, int flags) {
    if (units == 64) hipLaunchKernelGGL(enc_p1bwd_kernel<64>, dim3(2 * a.B * 4), dim3(P1_NTH), 0, s, a, planes, ab, dh, Bp, flags);
    else hipLaunchKernelGGL(enc_p1bwd_kernel<32>, dim3(2 * a.B * 8), dim3(P1_NTH), 0, s, a, planes, ab, dh, Bp, flags);
}
