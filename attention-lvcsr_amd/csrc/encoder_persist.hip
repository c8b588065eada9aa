// Persistent BiGRU layer kernels (forward and BPTT) for gfx950: ONE launch per layer and pass instead of two launches per
// time step; recurrent weights live in REGISTERS for the whole sequence, the hidden state in LDS.
//
// The recurrence is latency bound (2 800 dependent steps per pass on WSJ-base) and needs every hidden unit of the previous
// phase twice per step (h -> gates, r*h -> candidate; BPTT: dpre_c -> d(r*h), dpre_r -> dh).  What one such exchange costs
// inside a launch grows with the number of work-groups that take part and with the size of the vector
// (profiles/r01_handoff_probe.txt), so the decomposition keeps both minimal:
//   * utterances are independent, so a cluster serves only RB of them (RB = 1 at WSJ-base): the contraction is a GEMV and
//     runs on the VALU (v_pk_fma_f32), not on MFMA tiles that would be 15/16 empty;
//   * a thread keeps 3*KS weights (KS = 64: 192 VGPRs) for the whole sequence: thread (unit, q) owns rows
//     [q*KS, (q+1)*KS) of the unit's three weight columns, the KSPLIT threads of a unit are adjacent lanes and fold their
//     partial sums with DPP.  A 256-thread work-group therefore holds 192 KiB of weights = 256/KSPLIT units, and a
//     (direction, row group) cluster is only P = H*KSPLIT/256 work-groups (4 at H = 256, 16 at H = 512, ONE at H <= 128:
//     no exchange at all), one per CU;
//   * the phase vector (RB x H floats) travels as 8-byte {epoch, value} granules polled with relaxed agent-scope atomic
//     loads (sc1; MI355X_MICROARCH.md "handoff-1to1", granule form R2: no fences, no flags).  A cluster whose work-groups
//     were verified to share an XCD (persist.h cluster_shares_xcd, once per launch) publishes with plain stores — the
//     granules stay in the XCD's L2 —, any other cluster with write-through (sc1) stores: correct under every placement,
//     fast under the usual one.  All threads sweep the plane coalesced (RB*H/256 loads per lane) into LDS; every thread
//     then reads its K slice from LDS as broadcast ds_read_b128;
//   * only what the NEXT exchange waits for is computed before publishing (reset gate forward, d(r*h) backward); the update
//     gate / the dpre_u contraction run in the shadow of the hand-off;
//   * epoch = step + 1, planes zeroed by a memset node before every launch (graph-replay safe); a plane is overwritten only
//     after every consumer has read it (the producer needs all consumers' next-phase granules first), so single buffering
//     is enough;
//   * every spin is bounded: a work-group that waits too long raises the abort word and all work-groups leave.
// Saved tensors (u, r, c, rh, y / dxg) go out with plain stores off the critical path; the per-step operands that do not
// depend on the recurrence are prefetched one step ahead.
#include "common.h"
#include "graph_cache.h"
#include "lvsr_hip.h"
#include "persist.h"

typedef lvsr_bigru_fwd_args EncFwd;
typedef lvsr_bigru_bwd_args EncBwd0;

// Gather one phase vector (NG granules) into an LDS operand buffer dst[row][q][k] (row stride KSPLIT*LDH, slice stride LDH).
// PRIV = false: the 256 threads share the sweep (NG/256 loads per lane) and ONE buffer; the caller's __syncthreads publishes it.
// PRIV = true (experiment, LVSR_PERSIST_FLAGS & 32): every wave sweeps the WHOLE vector (NG/64 loads per lane) into a buffer of
// its own, so no work-group barrier sits between the hand-off and the contraction — measured slower (see PF_PRIVATE).
// Returns false when the cluster gave up.
template <int NG, bool PRIV, int NTH = 256>
struct Sweep {                       // one sweep of granule loads of a lane: NG/64 (wave-private) or NG/256 (shared) of them
    static constexpr int NT = PRIV ? 64 : NTH;
    static constexpr int N = (NG + NT - 1) / NT;
    u64 w[N];
    __device__ __forceinline__ void issue(const u64* g, unsigned epoch) {
        const int tid = PRIV ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = tid + NT * i;
            if (NG % NT == 0 || idx < NG) w[i] = __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else w[i] = (u64)epoch << 32;
        }
    }
    __device__ __forceinline__ bool complete(unsigned epoch) const {       // wave-uniform
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && ((unsigned)(w[i] >> 32) == epoch);
        return __all(ok);
    }
};

// One sweep in flight at a time.  Measured: pipelining sweeps (a second sweep issued before the first is examined, or a first
// sweep issued right after the own publish) makes the step SLOWER, 3.8 instead of 2.5 us — every outstanding sc1 load sits in
// the consumer CU's own memory queue in front of the one that will carry the new epoch (MI355X_MICROARCH.md: "the price sits in
// the CONSUMER CU's own memory queue").
// NPL > 1: NPL planes that were published together and lie back to back are swept as one vector; plane p lands in the LDS
// buffer dst + p * DSTRIDE.
// NTH = number of SWEEPING threads (the first NTH of the work-group): the 512-thread kernels sweep with waves 0..3 only, so that
// waves 4..7 can fetch the next step's operands without a global load ever sitting in front of a sweep in a polling wave's
// in-order memory queue (measured: those loads cost 0.18 / 0.33 us per forward / BPTT step there, tools/probe_persist.py).
template <int NG1, int HP, int KS, int LDH, int KSPLIT, bool PRIV, int NPL = 1, int DSTRIDE = 0, int NTH = 256>
__device__ __forceinline__ bool gather_plane(const u64* g, unsigned epoch, float* dst, int* abort_word, int flags = 0) {
    constexpr int NG = NG1 * NPL;
    typedef Sweep<NG, PRIV, NTH> S;
    constexpr int NT = S::NT;
    if (!PRIV && (int)threadIdx.x >= NTH) return true;              // (wave-uniform: NTH is a multiple of 64)
    const int tid = PRIV ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    S a;
    unsigned spins = 0;
    for (;;) {
        a.issue(g, epoch);
        if (a.complete(epoch) || (flags & PF_NOWAIT)) break;
        ++spins;
        if ((spins & 127u) == 0u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > PERSIST_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < S::N; ++i) {
        const int idx = tid + NT * i;
        if (NG % NT == 0 || idx < NG) {
            const int pl = idx / NG1, rem = idx % NG1, row = rem / HP, k = rem % HP;
            dst[pl * DSTRIDE + (row * KSPLIT + k / KS) * LDH + (k % KS)] = __uint_as_float((unsigned)a.w[i]);
        }
    }
    return true;
}
// A cluster of ONE work-group (H <= 128) has nobody to hand anything to: the phase vector goes straight into the LDS operand
// buffer and the hand-off is a work-group barrier.
template <int KS, int LDH, int KSPLIT>
__device__ __forceinline__ void lds_publish(float* dst, int row, int k, float v) {
    dst[(row * KSPLIT + k / KS) * LDH + (k % KS)] = v;
}

// what follows a gather: the work-group barrier for a shared buffer; for wave-private buffers only the wave's own LDS order
// matters (lanes of a wave run in lock step; the builtin keeps the compiler from moving the reads above the writes)
template <bool PRIV>
__device__ __forceinline__ void gather_fence() {
    if (PRIV) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}

// acc[r] += sum_x w[x] * v[r][q][x]  over this thread's K slice (pairs of k in one v_pk_fma_f32)
template <int KS, int RB, int LDH, int KSPLIT>
__device__ __forceinline__ void slice_dot(const f32x2 (&w)[KS / 2], const float* buf, int q, float (&out)[RB], int flags = 0) {
    if (flags & PF_NODOT) {
#pragma unroll
        for (int r = 0; r < RB; ++r) out[r] = buf[(r * KSPLIT + q) * LDH];
        return;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const float4* hv = (const float4*)(buf + (r * KSPLIT + q) * LDH);
        f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
        for (int x = 0; x < KS / 4; ++x) {
            const float4 h4 = hv[x];
            const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
            a0 = w[2 * x] * lo + a0;
            a1 = w[2 * x + 1] * hi + a1;
        }
        out[r] = group_sum<KSPLIT>((a0.x + a1.x) + (a0.y + a1.y));
    }
}

// the element of sums[] this lane owns in pass i of the epilogue (rows q, q+KSPLIT, ...), without dynamic indexing
template <int RB, int KSPLIT>
__device__ __forceinline__ float pick_row(const float (&s)[RB], int q, int i) {
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r)
        if (r % KSPLIT == q && r / KSPLIT == i) v = s[r];
    return v;
}

struct PersistGeom { int KS, KSPLIT, HP, UNITS, P, RB, rt, grid, NTH; long long plane; };
static int wide_cluster_capacity();      // defined behind the kernels
// Variant for a hidden size: the smallest padded size HP = KS*KSPLIT >= H among the built ones; RB = the smallest number of
// utterances per cluster (1, 2, 4, 8) whose grid fits the chip with one work-group per CU.
static bool persist_geom(int B, int H, PersistGeom& g) {
    if (H <= 128) { g.KS = 64; g.KSPLIT = 2; }
    else if (H <= 256) { g.KS = 64; g.KSPLIT = 4; }
    else if (H <= 512) { g.KS = 64; g.KSPLIT = 8; }
    else return false;
    g.NTH = 256;
    g.HP = g.KS * g.KSPLIT; g.UNITS = g.NTH / g.KSPLIT; g.P = g.HP / g.UNITS;
    const int want = min(8, lvsr_knob(LVSR_KNOB_PERSIST_ROWS));
    g.RB = 0;
    for (int rb = 1; rb <= 8; rb *= 2) {
        const int rt = (B + rb - 1) / rb;
        if (2 * rt * g.P <= lvsr_max_cluster_wgs() && rb >= want) { g.RB = rb; break; }
    }
    if (!g.RB) return false;
    // One or two utterances per cluster: 512-thread work-groups, two waves per SIMD — half the k-slice per thread (96
    // instead of 192 weight registers), twice the lanes per unit, the same number of work-groups per cluster; 2.03 instead of
    // 2.12 us per step on WSJ-base (the second wave hides the first one's LDS and hand-off latencies; WSJ-base step 21.42 ->
    // 20.79 ms).  1024 threads (four waves, 48 weight registers) were measured too: 26.4 ms — the sweeps and barriers of 16
    // waves cost more than their latency hiding buys.  Knob LVSR_KNOB_PERSIST_THREADS = 256 brings the one-wave version back.
    if (g.RB <= 2 && lvsr_knob(LVSR_KNOB_PERSIST_THREADS) != 256) { g.NTH = 512; g.KS /= 2; g.KSPLIT *= 2; g.UNITS = g.NTH / g.KSPLIT; }
    g.rt = (B + g.RB - 1) / g.RB;
    // 128 < H <= 256, one or two utterances per cluster: clusters of 8 work-groups of 32 units (thread = 16 rows of its unit's
    // columns, 48 weight registers) instead of 4 of 64 — half the LDS operand traffic per CU and contraction (the contractions
    // are LDS-broadcast bound), twice the CUs: 2.01 / 2.20 instead of 2.13 / 2.38 us per forward / BPTT step of the layer probe,
    // WSJ-base step 17.5 -> 16.7 ms (profiles/r03_persist_probe_wide.txt).  At B = 16 that is 256 work-groups: the limit is what the
    // device can hold at once — the occupancy the runtime reports for these kernels (two work-groups per CU) times the CUs
    // (lvsr_max_cluster_wgs) — not one work-group per CU.  PF_NARROW keeps clusters of 4.
    // Round 4: ... and only while the wide clusters still sit ONE per CU.  Two per CU share the CU's LDS ports and issue slots: at a
    // per-GPU batch of 32 (512 work-groups) the step took 2.17 us against 1.93 us for 256 work-groups in clusters of 4 (WSJ-base step
    // 19.2 -> 17.1 ms; batch 64, two utterances per cluster: 31.0 -> 28.7 ms; gpurun session b1).
    if (!(lvsr_knob(LVSR_KNOB_PERSIST_FLAGS) & PF_NARROW) && g.NTH == 512 && g.HP == 256 &&
        2 * g.rt * 8 <= min(wide_cluster_capacity(), lvsr_max_cluster_wgs())) {
        g.KS = 16; g.KSPLIT = 16; g.UNITS = 32; g.P = 8;
    }
    g.grid = cluster_grid(2 * g.rt, g.P, lvsr_knob(LVSR_KNOB_PERSIST_FLAGS));     // (resident: 2 rt P; the padding exits at once)
    g.plane = (long long)g.RB * g.HP;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// LD9 (see enc_pbwd_kernel): a ninth wave does the staging (gate inputs and mask of step n + 2) instead of waves 4..7, which compute too.
template <int KS, int KSPLIT, int RB, bool PRIVOK, int NTH = 256, bool LD9 = false>
__global__ __launch_bounds__(NTH + (LD9 ? 64 : 0), (KS == 16 && NTH == 512) ? 4 : 1) void enc_pfwd_kernel(EncFwd a, u64* planes, u64* hello, int* abort_word, int flags) {
    constexpr int HP = KS * KSPLIT, UNITS = NTH / KSPLIT, P = HP / UNITS, LDH = KS + 4, NG = RB * HP;
    constexpr int NR = (RB + KSPLIT - 1) / KSPLIT;          // rows a lane owns in the epilogue
    constexpr bool PRIV = PRIVOK && NG <= 512 && P > 1 && NTH == 256;     // wave-private operand buffers, see gather_plane
    constexpr int NBUF = PRIV ? 4 : 1;
    __shared__ __attribute__((aligned(16))) float hbuf_all[2][NBUF][RB * KSPLIT * LDH];
    float* const hbuf[2] = {hbuf_all[0][PRIV ? (threadIdx.x >> 6) : 0], hbuf_all[1][PRIV ? (threadIdx.x >> 6) : 0]};
    // STAGED (512-thread kernels): waves 0..3 sweep the granules, waves 4..7 fetch the next step's operands (gate inputs, mask)
    // and hand them over through LDS — a polling wave then issues no other global load (see gather_plane)
    // (only where the plane fits waves 0..3 with one granule per lane — one utterance per cluster, H <= 256: elsewhere all waves
    // sweep and the owners fetch their operands themselves; measured with two granules per lane on waves 0..3: B = 32 22.4 vs 21.8 ms
    // per step, WSJ-deep 67.6 vs 66.6)
    constexpr bool STAGED = NTH == 512 && P > 1 && NG <= 256;
    constexpr int SWEEP = STAGED ? 256 : NTH;
    constexpr int ITEMS = RB * 3 * UNITS + RB, NLD = (ITEMS + 255) / 256;
    __shared__ float opnd[STAGED ? ITEMS : 1];
    const int H = a.H, B = a.B, T = a.T;
    const int rt = (B + RB - 1) / RB;
    int cl, p;
    if (!cluster_of_block(P, 2 * rt, flags, cl, p)) return;
    const bool save = !(flags & PF_NOSAVE);
    const bool plain = P > 1 && cluster_shares_xcd(hello + (size_t)cl * P, P, p, abort_word) && !(flags & PF_SC1);
    const int dir = cl / rt, b0 = (cl % rt) * RB;
    static_assert(!LD9 || (STAGED && RB == 1), "the loader wave serves one utterance per cluster");
    if (LD9 && (int)threadIdx.x >= NTH) {
        // ---- the loader wave: item idx = gate * UNITS + unit, idx = 3 UNITS: the row's mask; two barriers per step = the computing waves' fences
        const int lane = (int)threadIdx.x - NTH;
        // (straight-line loads from clamped addresses, fixed up by selects: loads under branches make the compiler wait for ALL outstanding
        // loads — the ones just issued too — where the older set is handed over)
        auto issue = [&](int ns, float (&dst)[2]) {
            const int nsc = min(ns, T - 1), tn = dir == 0 ? nsc : T - 1 - nsc;
            const float* xrow = a.xg + ((size_t)tn * B + b0) * 6 * H + dir * 3 * H;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const int idx = lane + 64 * l, jj = min(p * UNITS + idx % UNITS, H - 1), gsel = min(idx / UNITS, 2);
                const bool is_mask = idx == 3 * UNITS;
                const float* src = (is_mask && a.mask) ? a.mask + (size_t)tn * B + b0 : xrow + gsel * H + jj;
                const float v = *src;
                dst[l] = is_mask ? (a.mask ? v : 1.f) : ((idx < 3 * UNITS && p * UNITS + idx % UNITS < H) ? v : 0.f);
            }
        };
        // two register sets in turn (a copy at the end of a step would wait for the loads that step has just issued)
        float pa[2], pb[2];
        auto hand = [&](int n, const float (&v)[2]) {             // step n: the operands of step n + 1 between the step's two fences
            __syncthreads();
            if (n + 1 < T) {
                opnd[lane] = v[0];
                if (lane + 64 < ITEMS) opnd[lane + 64] = v[1];
            }
            __syncthreads();
        };
        issue(1, pa);
        for (int n = 0; n < T; n += 2) {
            issue(n + 2, pb);
            hand(n, pa);
            if (n + 1 < T) {
                issue(n + 3, pa);
                hand(n + 1, pb);
            }
        }
        return;
    }
    const int tid = threadIdx.x, q = tid % KSPLIT, j = p * UNITS + tid / KSPLIT, k0 = q * KS;
    const bool junit = j < H;
    const bool staged = STAGED && (LD9 || !(flags & PF_NOSTAGE)); // (PF_NOSTAGE: the owners fetch their operands themselves, for A/B runs)
    const bool loader = !LD9 && staged && tid >= 256;
    // ---- this thread's weight slice, in registers for the whole sequence
    f32x2 wr[KS / 2], wu[KS / 2], wc[KS / 2];
    {
        const float* Whg = a.Whg_p[dir];      // persistent mode: the PLAIN (H,2H) / (H,H) weights
        const float* Whh = a.Whh_p[dir];
        const size_t jc = (size_t)min(j, H - 1);
#pragma unroll
        for (int x = 0; x < KS / 2; ++x) {
            float v[6];
#pragma unroll
            for (int e = 0; e < 2; ++e) {      // clamped addresses, zeroed afterwards: straight-line loads, all in flight at once
                const int k = k0 + 2 * x + e;
                const float keep = (junit && k < H) ? 1.f : 0.f;
                const size_t kc = (size_t)min(k, H - 1);
                v[e] = Whg[kc * 2 * H + jc] * keep;
                v[2 + e] = Whg[kc * 2 * H + H + jc] * keep;
                v[4 + e] = Whh[kc * H + jc] * keep;
            }
            wu[x] = (f32x2){v[0], v[1]}; wr[x] = (f32x2){v[2], v[3]}; wc[x] = (f32x2){v[4], v[5]};
        }
    }
    u64* gh = planes + (size_t)cl * 2 * NG;
    u64* grh = gh + NG;
    // ---- epilogue ownership: lane q of a unit's group handles rows q, q+KSPLIT, ...
    float hown[NR], n_xin[NR], n_gu[NR], n_gr[NR], n_m[NR];
    bool rvalid[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = q + i * KSPLIT;
        rvalid[i] = junit && r < RB && b0 + r < B;
        hown[i] = rvalid[i] ? a.h0[dir][j] : 0.f;
        n_xin[i] = n_gu[i] = n_gr[i] = 0.f; n_m[i] = 1.f;
        if (rvalid[i]) {
            const int t = dir == 0 ? 0 : T - 1;
            const size_t row = (size_t)t * B + b0 + r;
            const float* xr = a.xg + row * 6 * H + dir * 3 * H;
            n_xin[i] = xr[j]; n_gu[i] = xr[H + j]; n_gr[i] = xr[2 * H + j];
            if (a.mask) n_m[i] = a.mask[row];
        }
    }
    // h_{-1} = initial state, broadcast over the rows (every wave fills its own buffer when they are private)
    for (int idx = PRIV ? (tid & 63) : tid; idx < RB * HP; idx += PRIV ? 64 : NTH) {
        const int row = idx / HP, k = idx % HP;
        hbuf[0][(row * KSPLIT + k / KS) * LDH + (k % KS)] = k < H ? a.h0[dir][k] : 0.f;
    }
    // (STAGED) what waves 4..7 fetch for scan step `ns`: item idx = ((row * 3 + gate) * UNITS + unit), then the rows' masks
    auto stage_issue = [&](int ns, float (&dst)[NLD]) {
        const bool fetch = loader && ns < T && !(flags & PF_NOPREFETCH);
        const int tn = dir == 0 ? ns : T - 1 - ns;
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int idx = tid - 256 + 256 * l;
            dst[l] = 0.f;
            if (fetch && idx < RB * 3 * UNITS) {
                const int r = idx / (3 * UNITS), gsel = (idx / UNITS) % 3, jj = p * UNITS + idx % UNITS, bb = b0 + r;
                if (jj < H && bb < B) dst[l] = a.xg[((size_t)tn * B + bb) * 6 * H + dir * 3 * H + gsel * H + jj];
            } else if (fetch && idx < ITEMS) {
                const int bb = b0 + idx - RB * 3 * UNITS;
                dst[l] = (a.mask && bb < B) ? a.mask[(size_t)tn * B + bb] : 1.f;
            }
        }
    };
    float pf[NLD];
    if (STAGED && !LD9) stage_issue(1, pf);                       // (no loads unless `loader`)
    // subsampled output: phase t % sub and row t / sub of the step's time index, carried along instead of divided per step.  Same-box
    // A/B (round 6, two boxes each): 13.34 -> 13.12 ms per WSJ-base step on the pool's usual boxes — every layer's forward kernel 2 %
    // faster, the two with a subsampled output 6 % — but 14.02 -> 14.19 on the one box whose hand-offs are slower (all four layers 5 %
    // slower there): at 1.27 us per step the kernel's time follows how the polls of a step happen to line up with their partners' stores
    // (tools/probes/poll_probe.hip), which a few instructions more or less between publish and first poll shift either way.  The same
    // change in the BPTT kernel's prefetch (pb_dy_at) measured slower on both kinds of box and is not made.
    int sub_ph = 0, sub_row = 0;
    if (a.ysub) { const int t0 = dir == 0 ? 0 : T - 1; sub_ph = t0 % a.sub; sub_row = t0 / a.sub; }
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? n : T - 1 - n;
        float xin[NR], gu[NR], gr[NR], m[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) { xin[i] = n_xin[i]; gu[i] = n_gu[i]; gr[i] = n_gr[i]; m[i] = n_m[i]; }
        // waves 4..7: the operands of step n + 2 are issued here; those of step n + 1 (issued a step ago: long landed) are handed
        // over in front of the second barrier of this step
        float pf_new[NLD];
        if (STAGED && !LD9) stage_issue(n + 2, pf_new);
        if (P > 1 && n > 0) {
            const bool ok = (NG <= 256 || staged) ? gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 1, 0, SWEEP>(gh, (unsigned)n, hbuf[0], abort_word, flags)
                                                  : gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 1, 0, NTH>(gh, (unsigned)n, hbuf[0], abort_word, flags);
            if (!ok) return;
        }
        gather_fence<PRIV>();
        // ---- reset gate: the only thing the next exchange waits for
        float s[RB], rr[NR], uu[NR];
        slice_dot<KS, RB, LDH, KSPLIT>(wr, hbuf[0], q, s, flags);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            if (r < RB) {
                rr[i] = sigmoid_fast(pick_row<RB, KSPLIT>(s, q, i) + gr[i]);
                const float rh = rvalid[i] ? rr[i] * hown[i] : 0.f;
                if (P == 1) lds_publish<KS, LDH, KSPLIT>(hbuf[1], r, j, rh);
                else granule_store(grh + (size_t)r * HP + j, (unsigned)(n + 1), rh, plain);
                if (rvalid[i] && save) {
                    const size_t o = ((size_t)t * B + b0 + r) * 2 * H + dir * H + j;
                    a.r[o] = rr[i]; a.rh[o] = rh;
                }
            }
        }
        // ---- update gate, in the shadow of the hand-off
        slice_dot<KS, RB, LDH, KSPLIT>(wu, hbuf[0], q, s, flags);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            if (r < RB) {
                uu[i] = sigmoid_fast(pick_row<RB, KSPLIT>(s, q, i) + gu[i]);
                if (rvalid[i] && save) a.u[((size_t)t * B + b0 + r) * 2 * H + dir * H + j] = uu[i];
            }
        }
        if (STAGED && !LD9) {
            if (loader && n + 1 < T) {
#pragma unroll
                for (int l = 0; l < NLD; ++l) {
                    const int idx = tid - 256 + 256 * l;
                    if (idx < ITEMS) opnd[idx] = pf[l];
                }
            }
#pragma unroll
            for (int l = 0; l < NLD; ++l) pf[l] = pf_new[l];
        }
        if (P > 1) {
            const bool ok = (NG <= 256 || staged) ? gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 1, 0, SWEEP>(grh, (unsigned)(n + 1), hbuf[1], abort_word, flags)
                                                  : gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 1, 0, NTH>(grh, (unsigned)(n + 1), hbuf[1], abort_word, flags);
            if (!ok) return;
        }
        gather_fence<PRIV>();
        // ---- candidate, state update, mask blend
        slice_dot<KS, RB, LDH, KSPLIT>(wc, hbuf[1], q, s, flags);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            if (r < RB) {
                const float cand = tanh_fast(pick_row<RB, KSPLIT>(s, q, i) + xin[i]);
                float hn = cand * uu[i] + hown[i] * (1.f - uu[i]);
                hn = m[i] * hn + (1.f - m[i]) * hown[i];
                if (!rvalid[i]) hn = 0.f;
                if (P == 1) lds_publish<KS, LDH, KSPLIT>(hbuf[0], r, j, hn);
                else if (n + 1 < T) granule_store(gh + (size_t)r * HP + j, (unsigned)(n + 1), hn, plain);
                if (rvalid[i]) {
                    const size_t o = ((size_t)t * B + b0 + r) * 2 * H + dir * H + j;
                    if (save) a.c[o] = cand;
                    a.y[o] = hn;
                    if (a.ysub && sub_ph == 0) a.ysub[((size_t)sub_row * B + b0 + r) * 2 * H + dir * H + j] = hn;
                }
                hown[i] = hn;
            }
        }
        if (a.ysub) {
            if (dir == 0) { if (++sub_ph == a.sub) { sub_ph = 0; ++sub_row; } }
            else if (sub_ph == 0) { sub_ph = a.sub - 1; --sub_row; }
            else --sub_ph;
        }
        // ---- operands of the next step (independent of the recurrence)
        if (n + 1 < T && !(flags & PF_NOPREFETCH)) {
            const int tn = dir == 0 ? n + 1 : T - 2 - n;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (rvalid[i]) {
                    const int r = q + i * KSPLIT;
                    if (staged) {             // handed over by waves 4..7 in front of the second barrier of this step
                        const int ju = tid / KSPLIT;
                        n_xin[i] = opnd[(r * 3 + 0) * UNITS + ju]; n_gu[i] = opnd[(r * 3 + 1) * UNITS + ju]; n_gr[i] = opnd[(r * 3 + 2) * UNITS + ju];
                        n_m[i] = opnd[RB * 3 * UNITS + r];
                    } else {                  // fetched by the owner itself: in flight during the next hand-off
                        const size_t row = (size_t)tn * B + b0 + r;
                        const float* xr = a.xg + row * 6 * H + dir * 3 * H;
                        n_xin[i] = xr[j]; n_gu[i] = xr[H + j]; n_gr[i] = xr[2 * H + j];
                        if (a.mask) n_m[i] = a.mask[row];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward (BPTT).  Per step (forward direction walks t = T-1..0, backward direction t = 0..T-1):
//   dhn = m*dh; dpre_c = dhn*u*(1-c^2); dpre_u = dhn*(c-h_prev)*u*(1-u)
//   drh = dpre_c @ Whh^T;  dpre_r = drh*h_prev*r*(1-r)
//   dh_prev = dhn*(1-u) + (1-m)*dh + drh*r + dpre_u @ Whg[:, :H]^T + dpre_r @ Whg[:, H:]^T + dy[t_prev]
//   dxg[t] = [dpre_c | dpre_u | dpre_r]
// Thread (unit i, slice q) keeps ROW i of the three weight blocks (the transposed products contract over columns).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pb_dy_at(const EncBwd0& a, int t, int b, int dir, int j) {
    if (t < 0 || t >= a.T || (t % a.sub) != 0) return 0.f;
    return a.dy[((size_t)(t / a.sub) * a.B + b) * 2 * a.H + dir * a.H + j];
}

// (clusters of 8 — KS = 16 — are sized by what the device holds at once, wide_cluster_capacity(): two of their work-groups must fit
// a CU, i.e. 4 waves per SIMD = at most 128 registers)
// LD9 (one utterance per cluster of 8, one work-group per CU): the work-group is launched with a NINTH wave that does nothing but fetch
// the next steps' saved operands (u, c, r, h_prev, mask, dy of every unit of the work-group: 6 x 32 values) two steps ahead and hand them
// over through LDS in front of a step's FIRST fence; the owners pick them up behind it, in the shadow of the second hand-off.  All eight
// computing waves sweep the two-plane first gather of a step, and a global load issued by a sweeping wave sits in front of its sweep in
// the wave's in-order memory queue (0.33 us per step: tools/probe_persist.py, persist_flags 128).
template <int KS, int KSPLIT, int RB, bool PRIVOK, int NTH = 256, bool LD9 = false>
__global__ __launch_bounds__(NTH + (LD9 ? 64 : 0), (KS == 16 && NTH == 512) ? 4 : 1) void enc_pbwd_kernel(EncBwd0 a, u64* planes, u64* hello, int* abort_word, float* dh_out, int Bp, int flags) {
    constexpr int HP = KS * KSPLIT, UNITS = NTH / KSPLIT, P = HP / UNITS, LDH = KS + 4, NG = RB * HP;
    constexpr int NR = (RB + KSPLIT - 1) / KSPLIT;
    constexpr bool PRIV = PRIVOK && NG <= 512 && P > 1 && NTH == 256;
    constexpr int NBUF = PRIV ? 4 : 1;
    static_assert(!LD9 || (RB == 1 && P > 1 && !PRIV && UNITS == 32), "the loader wave serves one utterance per cluster, 32 units per work-group");
    __shared__ float opnd9[LD9 ? 6 * UNITS : 1];
    __shared__ __attribute__((aligned(16))) float vbuf_all[3][NBUF][RB * KSPLIT * LDH];     // dpre_c | dpre_u | dpre_r
    float* const vbuf[3] = {vbuf_all[0][PRIV ? (threadIdx.x >> 6) : 0], vbuf_all[1][PRIV ? (threadIdx.x >> 6) : 0],
                            vbuf_all[2][PRIV ? (threadIdx.x >> 6) : 0]};
    const int H = a.H, B = a.B, T = a.T;
    const int rt = (B + RB - 1) / RB;
    int cl, p;
    if (!cluster_of_block(P, 2 * rt, flags, cl, p)) return;
    const bool save = !(flags & PF_NOSAVE);
    const bool plain = P > 1 && cluster_shares_xcd(hello + (size_t)cl * P, P, p, abort_word) && !(flags & PF_SC1);
    const int dir = cl / rt, b0 = (cl % rt) * RB;
    if (LD9 && (int)threadIdx.x >= NTH) {
        // ---- the loader wave.  Lane = (half, unit): per step three loads — [u | c] of step t, [r | h_prev], [mask | dy] — from pointers
        //      that move by one (T) row per step; two work-group barriers per step, in step with the computing waves' gather fences
        const int lane = (int)threadIdx.x - NTH, hi = lane >> 5, jj = p * UNITS + (lane & 31);
        const bool live = jj < H && b0 < B;
        const size_t jo = (size_t)b0 * 2 * H + dir * H + min(jj, H - 1);
        const long long row = (long long)B * 2 * H, sgn = dir == 0 ? -1 : 1;
        const float* p0 = (hi ? a.c : a.u) + jo;                       // + t * row
        const float* p1 = (hi ? a.y : a.r) + jo;                       // r at t, h_prev = y at tp
        const float h0v = a.h0[dir][min(jj, H - 1)];
        // (straight-line loads from clamped addresses, fixed up by selects: loads under branches make the compiler wait for ALL outstanding
        // loads — the ones just issued too — where the older set is handed over)
        const float* p2 = hi ? a.dy + (size_t)b0 * 2 * H + dir * H + min(jj, H - 1) : (a.mask ? a.mask + b0 : a.h0[dir]);
        const long long row2 = hi ? row : (a.mask ? (long long)B : 0);
        auto issue = [&](int ns, float (&dst)[3]) {
            const int nsc = min(ns, T - 1), t = dir == 0 ? T - 1 - nsc : nsc, tp = t + (int)sgn, tpc = min(max(tp, 0), T - 1);
            const bool edge = tp != tpc, dy_here = !edge && (tp % a.sub) == 0;
            const float v0 = p0[(long long)t * row];
            const float v1 = p1[(long long)(hi ? tpc : t) * row];
            const float v2 = p2[(long long)(hi ? tpc / a.sub : t) * row2];
            dst[0] = live ? v0 : 0.f;
            dst[1] = live ? ((hi && edge) ? h0v : v1) : 0.f;
            dst[2] = hi ? ((live && dy_here) ? v2 : 0.f) : ((live && a.mask) ? v2 : 1.f);
        };
        // two register sets in turn (a copy at the end of a step would wait for the loads that step has just issued)
        float pa[3], pb[3];
        auto hand = [&](int n, const float (&v)[3]) {             // step n: the operands of step n + 1 (fetched a step ago) in front of its first fence
            if (n + 1 < T) { opnd9[lane] = v[0]; opnd9[64 + lane] = v[1]; opnd9[128 + lane] = v[2]; }
            __syncthreads();
            __syncthreads();
        };
        issue(1, pa);
        for (int n = 0; n < T; n += 2) {
            issue(n + 2, pb);
            hand(n, pa);
            if (n + 1 < T) {
                issue(n + 3, pa);
                hand(n + 1, pb);
            }
        }
        return;
    }
    const int tid = threadIdx.x, q = tid % KSPLIT, j = p * UNITS + tid / KSPLIT, k0 = q * KS;
    const bool junit = j < H;
    f32x2 wa[KS / 2], wbu[KS / 2], wbr[KS / 2];
    {
        const float* Whg = a.WhgT_p[dir];     // persistent mode: PLAIN (H,2H) / (H,H) weights, rows read in place
        const float* Whh = a.WhhT_p[dir];
        const size_t jc = (size_t)min(j, H - 1);
#pragma unroll
        for (int x = 0; x < KS / 2; ++x) {
            float v[6];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = k0 + 2 * x + e;
                const float keep = (junit && k < H) ? 1.f : 0.f;
                const size_t kc = (size_t)min(k, H - 1);
                v[e] = Whh[jc * H + kc] * keep;
                v[2 + e] = Whg[jc * 2 * H + kc] * keep;
                v[4 + e] = Whg[jc * 2 * H + H + kc] * keep;
            }
            wa[x] = (f32x2){v[0], v[1]}; wbu[x] = (f32x2){v[2], v[3]}; wbr[x] = (f32x2){v[4], v[5]};
        }
    }
    u64* gc = planes + (size_t)cl * 4 * NG;        // dpre_c, then dpre_u right behind it: published together, swept together
    u64* gu = gc + NG;
    u64* gr = gc + 2 * NG;                         // dpre_r
    const int t_first = dir == 0 ? T - 1 : 0;
    float dh[NR], n_u[NR], n_c[NR], n_r[NR], n_hp[NR], n_m[NR], n_dy[NR];
    bool rvalid[NR];
    auto prefetch = [&](int t, int i) {
        const int b = b0 + q + i * KSPLIT;
        const int tp = dir == 0 ? t - 1 : t + 1;
        const size_t o = ((size_t)t * B + b) * 2 * H + dir * H + j;
        n_u[i] = a.u[o]; n_c[i] = a.c[o]; n_r[i] = a.r[o];
        n_hp[i] = (tp < 0 || tp >= T) ? a.h0[dir][j] : a.y[((size_t)tp * B + b) * 2 * H + dir * H + j];
        n_m[i] = a.mask ? a.mask[(size_t)t * B + b] : 1.f;
        n_dy[i] = pb_dy_at(a, tp, b, dir, j);
    };
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = q + i * KSPLIT;
        rvalid[i] = junit && r < RB && b0 + r < B;
        dh[i] = rvalid[i] ? pb_dy_at(a, t_first, b0 + r, dir, j) : 0.f;
        n_u[i] = n_c[i] = n_r[i] = n_hp[i] = n_dy[i] = 0.f; n_m[i] = 1.f;
        if (rvalid[i]) prefetch(t_first, i);
    }
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? T - 1 - n : n;
        float uu[NR], cc[NR], rr[NR], hp[NR], part[NR];
        // ---- everything of this step that depends on dh elementwise only; publish dpre_c and dpre_u
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            uu[i] = n_u[i]; cc[i] = n_c[i]; rr[i] = n_r[i]; hp[i] = n_hp[i];
            if (r < RB) {
                const float dhn = n_m[i] * dh[i];
                const float dpc = rvalid[i] ? dhn * uu[i] * (1.f - cc[i] * cc[i]) : 0.f;
                const float dpu = rvalid[i] ? dhn * (cc[i] - hp[i]) * uu[i] * (1.f - uu[i]) : 0.f;
                if (P == 1) {
                    lds_publish<KS, LDH, KSPLIT>(vbuf[0], r, j, dpc);
                    lds_publish<KS, LDH, KSPLIT>(vbuf[1], r, j, dpu);
                } else {
                    granule_store(gc + (size_t)r * HP + j, (unsigned)(n + 1), dpc, plain);
                    granule_store(gu + (size_t)r * HP + j, (unsigned)(n + 1), dpu, plain);
                }
                part[i] = dhn * (1.f - uu[i]) + (1.f - n_m[i]) * dh[i] + n_dy[i];
                if (rvalid[i] && save) {
                    float* dx = a.dxg + ((size_t)t * B + b0 + r) * 6 * H + dir * 3 * H;
                    dx[j] = dpc; dx[H + j] = dpu;
                }
            }
        }
        // operands of the next step: in flight during the hand-offs (LD9: the loader wave's business)
        if (!LD9 && n + 1 < T && !(flags & PF_NOPREFETCH)) {
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (rvalid[i]) prefetch(dir == 0 ? t - 1 : t + 1, i);
        }
        if (P > 1 && !gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 2, NBUF * RB * KSPLIT * LDH, NTH>(gc, (unsigned)(n + 1), vbuf[0], abort_word)) return;
        gather_fence<PRIV>();
        if (LD9 && n + 1 < T && rvalid[0]) {          // handed over by the loader wave in front of this fence: [u | c], [r | h_prev], [mask | dy]
            const int ju = tid / KSPLIT;
            n_u[0] = opnd9[ju]; n_c[0] = opnd9[32 + ju]; n_r[0] = opnd9[64 + ju]; n_hp[0] = opnd9[96 + ju];
            n_m[0] = opnd9[128 + ju]; n_dy[0] = opnd9[160 + ju];
        }
        float s[RB], vu[RB];
        slice_dot<KS, RB, LDH, KSPLIT>(wa, vbuf[0], q, s);                     // d(r*h)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            if (r < RB) {
                const float drh = pick_row<RB, KSPLIT>(s, q, i);
                const float dpr = rvalid[i] ? drh * hp[i] * rr[i] * (1.f - rr[i]) : 0.f;
                if (P == 1) lds_publish<KS, LDH, KSPLIT>(vbuf[2], r, j, dpr);
                else granule_store(gr + (size_t)r * HP + j, (unsigned)(n + 1), dpr, plain);
                part[i] += drh * rr[i];
                if (rvalid[i] && save) a.dxg[((size_t)t * B + b0 + r) * 6 * H + dir * 3 * H + 2 * H + j] = dpr;
            }
        }
        // ---- dpre_u @ Whg[:, :H]^T in the shadow of the hand-off (dpre_u came in with dpre_c's sweep)
        slice_dot<KS, RB, LDH, KSPLIT>(wbu, vbuf[1], q, vu);
        if (P > 1 && !gather_plane<NG, HP, KS, LDH, KSPLIT, PRIV, 1, 0, NTH>(gr, (unsigned)(n + 1), vbuf[2], abort_word)) return;
        gather_fence<PRIV>();
        slice_dot<KS, RB, LDH, KSPLIT>(wbr, vbuf[2], q, s);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = q + i * KSPLIT;
            if (r < RB) dh[i] = rvalid[i] ? part[i] + pick_row<RB, KSPLIT>(vu, q, i) + pick_row<RB, KSPLIT>(s, q, i) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (rvalid[i]) dh_out[((size_t)dir * Bp + b0 + q + i * KSPLIT) * H + j] = dh[i];       // d initial state, per utterance
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

// ---------------------------------------------------------------------------------------------------------------
// Unit-blocked forward kernel (one utterance per cluster): a group of G = 16 U lanes serves U units, lane q of the group keeps the KS
// rows [q KS, (q+1) KS) of the three weight columns of ALL U units (the same 3 KS U registers as one unit with U KS rows) and reads
// its slice of the phase vector from LDS ONCE for the U units — the contractions are LDS-broadcast bound, and this divides the
// operand traffic by U (H = 512: 2 instead of 8 ds_read_b128 per lane and contraction).  The U partial sums are folded with a
// butterfly over the group's top log2(U) lane bits (each stage halves the values a lane carries) and a 16-lane DPP sum: lane
// 16 u of the group ends up owning unit u (epilogue, publishing, saved tensors).
// ---------------------------------------------------------------------------------------------------------------
template <int KS, int U>
__device__ __forceinline__ void ub_dot(const f32x2 (&w)[U][KS / 2], const float* buf, int q, float (&out)[U]) {
    const float4* hv = (const float4*)(buf + q * (KS + 4));
    f32x2 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int x = 0; x < KS / 4; ++x) {
        const float4 h4 = hv[x];
        const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u] = w[u][2 * x] * lo + acc[u];
            acc[u] = w[u][2 * x + 1] * hi + acc[u];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = acc[u].x + acc[u].y;
}
// the total of value (q >> 4) over the group's lanes, in every lane of row q >> 4
template <int U>
__device__ __forceinline__ float ub_fold(const float (&v)[U], int q) {
    float r;
    if (U == 4) {
        const bool up = (q & 32) != 0;
        const float a0 = (up ? v[2] : v[0]) + __shfl_xor(up ? v[0] : v[2], 32, 64);
        const float a1 = (up ? v[3] : v[1]) + __shfl_xor(up ? v[1] : v[3], 32, 64);
        const bool up2 = (q & 16) != 0;
        r = (up2 ? a1 : a0) + __shfl_xor(up2 ? a0 : a1, 16, 64);
    } else if (U == 2) {
        const bool up = (q & 16) != 0;
        r = (up ? v[1] : v[0]) + __shfl_xor(up ? v[0] : v[1], 16, 64);
    } else {
        r = v[0];
    }
    return group_sum<16>(r);
}

template <int KS, int U, int NTH>
__global__ __launch_bounds__(NTH) void enc_pfwd_ub_kernel(EncFwd a, u64* planes, u64* hello, int* abort_word, int flags) {
    constexpr int G = 16 * U, HP = KS * G, UNITS = NTH / G * U, P = HP / UNITS, LDH = KS + 4, NG = HP;
    __shared__ __attribute__((aligned(16))) float hbuf[2][G * LDH];
    const int H = a.H, B = a.B, T = a.T;
    int cl, p;
    if (!cluster_of_block(P, 2 * B, flags, cl, p)) return;
    const bool save = !(flags & PF_NOSAVE);
    const bool plain = cluster_shares_xcd(hello + (size_t)cl * P, P, p, abort_word) && !(flags & PF_SC1);
    const int dir = cl / B, b = cl % B;
    const int tid = threadIdx.x, q = tid % G, jb = p * UNITS + (tid / G) * U, k0 = q * KS;
    // ---- the weight slices of the group's U units, in registers for the whole sequence
    f32x2 wr[U][KS / 2], wu[U][KS / 2], wc[U][KS / 2];
    {
        const float* Whg = a.Whg_p[dir];
        const float* Whh = a.Whh_p[dir];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = jb + u;
            const size_t jc = (size_t)min(j, H - 1);
#pragma unroll
            for (int x = 0; x < KS / 2; ++x) {
                float v[6];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int k = k0 + 2 * x + e;
                    const float keep = (j < H && k < H) ? 1.f : 0.f;
                    const size_t kc = (size_t)min(k, H - 1);
                    v[e] = Whg[kc * 2 * H + jc] * keep;
                    v[2 + e] = Whg[kc * 2 * H + H + jc] * keep;
                    v[4 + e] = Whh[kc * H + jc] * keep;
                }
                wu[u][x] = (f32x2){v[0], v[1]}; wr[u][x] = (f32x2){v[2], v[3]}; wc[u][x] = (f32x2){v[4], v[5]};
            }
        }
    }
    u64* gh = planes + (size_t)cl * 2 * NG;
    u64* grh = gh + NG;
    // ---- the owner lane of unit j = jb + (q >> 4): lane 16 (q >> 4) of the group
    const int j = jb + (q >> 4);
    const bool own = (q & 15) == 0 && j < H;
    float hown = own ? a.h0[dir][j] : 0.f, n_xin = 0.f, n_gu = 0.f, n_gr = 0.f, n_m = 1.f;
    auto fetch = [&](int t) {
        const size_t row = (size_t)t * B + b;
        const float* xr = a.xg + row * 6 * H + dir * 3 * H;
        n_xin = xr[j]; n_gu = xr[H + j]; n_gr = xr[2 * H + j];
        if (a.mask) n_m = a.mask[row];
    };
    if (own) fetch(dir == 0 ? 0 : T - 1);
    for (int k = tid; k < HP; k += NTH) hbuf[0][(k / KS) * LDH + (k % KS)] = k < H ? a.h0[dir][k] : 0.f;
    for (int n = 0; n < T; ++n) {
        const int t = dir == 0 ? n : T - 1 - n;
        const float xin = n_xin, gu = n_gu, gr = n_gr, m = n_m;
        if (n > 0 && !gather_plane<NG, HP, KS, LDH, G, false, 1, 0, NTH>(gh, (unsigned)n, hbuf[0], abort_word, flags)) return;
        __syncthreads();
        // ---- reset gate: the only thing the next exchange waits for
        float s[U];
        ub_dot<KS, U>(wr, hbuf[0], q, s);
        const float rr = sigmoid_fast(ub_fold<U>(s, q) + gr);
        const float rh = own ? rr * hown : 0.f;
        if ((q & 15) == 0) granule_store(grh + min(j, HP - 1), (unsigned)(n + 1), rh, plain);
        if (own && save) {
            const size_t o = ((size_t)t * B + b) * 2 * H + dir * H + j;
            a.r[o] = rr; a.rh[o] = rh;
        }
        // ---- update gate, in the shadow of the hand-off
        ub_dot<KS, U>(wu, hbuf[0], q, s);
        const float uu = sigmoid_fast(ub_fold<U>(s, q) + gu);
        if (own && save) a.u[((size_t)t * B + b) * 2 * H + dir * H + j] = uu;
        if (!gather_plane<NG, HP, KS, LDH, G, false, 1, 0, NTH>(grh, (unsigned)(n + 1), hbuf[1], abort_word, flags)) return;
        __syncthreads();
        // ---- candidate, state update, mask blend
        ub_dot<KS, U>(wc, hbuf[1], q, s);
        const float cand = tanh_fast(ub_fold<U>(s, q) + xin);
        float hn = cand * uu + hown * (1.f - uu);
        hn = m * hn + (1.f - m) * hown;
        if (!own) hn = 0.f;
        if ((q & 15) == 0 && n + 1 < T) granule_store(gh + min(j, HP - 1), (unsigned)(n + 1), hn, plain);
        if (own) {
            const size_t o = ((size_t)t * B + b) * 2 * H + dir * H + j;
            if (save) a.c[o] = cand;
            a.y[o] = hn;
            if (a.ysub && (t % a.sub) == 0) a.ysub[((size_t)(t / a.sub) * B + b) * 2 * H + dir * H + j] = hn;
        }
        hown = hn;
        if (own && n + 1 < T && !(flags & PF_NOPREFETCH)) fetch(dir == 0 ? n + 1 : T - 2 - n);
    }
}

// How many work-groups of the wide (8 per cluster) kernels the device holds at once: occupancy per CU as the runtime computes it
// from their registers / LDS (2 on MI355X) x lvsr_max_cluster_wgs(); a launch at most this large is resident as a whole, which the
// clusters' mutual waiting needs.  The forward and BPTT kernels of both row counts are asked; the smallest answer counts.
static int wide_cluster_capacity() {
    static int occ_cached = -1;
    if (occ_cached < 0) {
        int occ = 1 << 20, n = 0;
        const bool ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc_pfwd_kernel<16, 16, 1, false, 512>, 512, 0) == hipSuccess && (occ = min(occ, n)) >= 0 &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc_pfwd_kernel<16, 16, 2, false, 512>, 512, 0) == hipSuccess && (occ = min(occ, n)) >= 0 &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc_pbwd_kernel<16, 16, 1, false, 512>, 512, 0) == hipSuccess && (occ = min(occ, n)) >= 0 &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, enc_pbwd_kernel<16, 16, 2, false, 512>, 512, 0) == hipSuccess && (occ = min(occ, n)) >= 0;
        if (!ok) { (void)hipGetLastError(); occ = 0; }
        occ_cached = min(occ, 2);
    }
    return occ_cached * lvsr_max_cluster_wgs();
}

static int persist_flags() { return lvsr_knob(LVSR_KNOB_PERSIST_FLAGS); }

// utterances per cluster the persistent kernels would use for (B,H); 0 = not available
extern "C" int lvsr_bigru_persist_rows(int B, int H) {
    PersistGeom g;
    if (B <= 0 || H <= 0 || !persist_geom(B, H, g)) return 0;
    return g.RB;
}

extern "C" long long lvsr_bigru_persist_ws_bytes(int B, int H) {
    PersistGeom g;
    if (B <= 0 || H <= 0) return 0;
    // sized for one utterance per cluster (the largest number of clusters) so LVSR_PERSIST_ROWS cannot outgrow it
    if (!persist_geom(B, H, g)) return 0;
    // abort word + 4 planes per (direction, utterance) + one XCC_ID granule per work-group of the largest grid any knob setting
    // can ask for (one utterance per cluster, 16 work-groups per cluster, padded to a multiple of 8 clusters)
    const long long hello = (long long)cluster_grid(2 * B, 16, 0) * 8;
    return 256 + (long long)2 * (B + 16) * 4 * g.HP * 8 + (hello > 8 * 1024 ? hello : 8 * 1024);
}

template <int KS, int KSPLIT>
static void launch_fwd(hipStream_t s, const EncFwd& a, const PersistGeom& g, u64* planes, u64* hello, int* ab, int flags) {
    if (g.NTH == 512 && g.RB == 1 && g.HP == 512 && !(flags & PF_NOUB)) {
        // 256 < H <= 512, one utterance per cluster: four units per thread group (2.88 -> 2.37 us per forward step, WSJ-deep layer
        // probe).  At H <= 256 it measured no gain over the loader-wave kernel (1.78 vs 1.76), and the BPTT counterpart was slower
        // than or equal to the plain kernel at both sizes (2.95 vs 2.90 at H = 512: its contractions are not what bounds it): not built.
        hipLaunchKernelGGL((enc_pfwd_ub_kernel<8, 4, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, flags);
        return;
    }
    if (g.NTH == 512) {                        // two waves per SIMD: half the k-slice per thread (one or two utterances per cluster)
        if (g.P == 8 && g.KS == 16 && g.RB == 1 && !(flags & (PF_NOLD9F | PF_NOSTAGE)) && g.grid <= lvsr_max_cluster_wgs())      // one work-group per CU: nine waves fit
            hipLaunchKernelGGL((enc_pfwd_kernel<16, 16, 1, false, 512, true>), dim3(g.grid), dim3(576), 0, s, a, planes, hello, ab, flags);
        else if (g.P == 8 && g.KS == 16 && g.RB == 1) hipLaunchKernelGGL((enc_pfwd_kernel<16, 16, 1, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, flags);
        else if (g.P == 8 && g.KS == 16) hipLaunchKernelGGL((enc_pfwd_kernel<16, 16, 2, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, flags);
        else if (g.RB == 1) hipLaunchKernelGGL((enc_pfwd_kernel<KS / 2, KSPLIT * 2, 1, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, flags);
        else hipLaunchKernelGGL((enc_pfwd_kernel<KS / 2, KSPLIT * 2, 2, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, flags);
        return;
    }
    switch (g.RB) {
        case 1:
            if (!(flags & PF_PRIVATE)) hipLaunchKernelGGL((enc_pfwd_kernel<KS, KSPLIT, 1, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, flags);
            else hipLaunchKernelGGL((enc_pfwd_kernel<KS, KSPLIT, 1, true>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, flags);
            break;
        case 2: hipLaunchKernelGGL((enc_pfwd_kernel<KS, KSPLIT, 2, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, flags); break;
        case 4: hipLaunchKernelGGL((enc_pfwd_kernel<KS, KSPLIT, 4, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, flags); break;
        default: hipLaunchKernelGGL((enc_pfwd_kernel<KS, KSPLIT, 8, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, flags); break;
    }
}
template <int KS, int KSPLIT>
static void launch_bwd(hipStream_t s, const EncBwd0& a, const PersistGeom& g, u64* planes, u64* hello, int* ab, float* dh, int Bp, int flags) {
    if (g.NTH == 512) {
        if (g.P == 8 && g.KS == 16 && g.RB == 1 && !(flags & PF_NOLD9) && g.grid <= lvsr_max_cluster_wgs())      // one work-group per CU: nine waves fit
            hipLaunchKernelGGL((enc_pbwd_kernel<16, 16, 1, false, 512, true>), dim3(g.grid), dim3(576), 0, s, a, planes, hello, ab, dh, Bp, flags);
        else if (g.P == 8 && g.KS == 16 && g.RB == 1) hipLaunchKernelGGL((enc_pbwd_kernel<16, 16, 1, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, dh, Bp, flags);
        else if (g.P == 8 && g.KS == 16) hipLaunchKernelGGL((enc_pbwd_kernel<16, 16, 2, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, dh, Bp, flags);
        else if (g.RB == 1) {
            if constexpr (KSPLIT == 8) {       // H = 512 (WSJ-deep): 16 work-groups of 32 units per utterance — the loader wave as above
                if (!(flags & PF_NOLD9) && g.grid <= lvsr_max_cluster_wgs()) {
                    hipLaunchKernelGGL((enc_pbwd_kernel<KS / 2, KSPLIT * 2, 1, false, 512, true>), dim3(g.grid), dim3(576), 0, s, a, planes, hello, ab, dh, Bp, flags);
                    return;
                }
            }
            hipLaunchKernelGGL((enc_pbwd_kernel<KS / 2, KSPLIT * 2, 1, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, dh, Bp, flags);
        }
        else hipLaunchKernelGGL((enc_pbwd_kernel<KS / 2, KSPLIT * 2, 2, false, 512>), dim3(g.grid), dim3(512), 0, s, a, planes, hello, ab, dh, Bp, flags);
        return;
    }
    switch (g.RB) {
        case 1:
            if (!(flags & PF_PRIVATE)) hipLaunchKernelGGL((enc_pbwd_kernel<KS, KSPLIT, 1, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, dh, Bp, flags);
            else hipLaunchKernelGGL((enc_pbwd_kernel<KS, KSPLIT, 1, true>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, dh, Bp, flags);
            break;
        case 2: hipLaunchKernelGGL((enc_pbwd_kernel<KS, KSPLIT, 2, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, dh, Bp, flags); break;
        case 4: hipLaunchKernelGGL((enc_pbwd_kernel<KS, KSPLIT, 4, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, dh, Bp, flags); break;
        default: hipLaunchKernelGGL((enc_pbwd_kernel<KS, KSPLIT, 8, false>), dim3(g.grid), dim3(256), 0, s, a, planes, hello, ab, dh, Bp, flags); break;
    }
}

int lvsr_bigru_fwd_persistent(hipStream_t s, const EncFwd& a0, int use_graph) {
    EncFwd a = a0;
    PersistGeom g;
    LVSR_REQUIRE(persist_geom(a.B, a.H, g) && a.sync_ws, "lvsr_bigru_fwd: persistent mode not available for B=%d H=%d", a.B, a.H);
    int* ab = (int*)a.sync_ws;
    u64* planes = (u64*)((char*)a.sync_ws + 256);
    u64* hello = planes + (size_t)2 * g.rt * 2 * g.plane;          // one {1, XCC_ID} granule per work-group, behind the planes
    const size_t bytes = (size_t)2 * g.rt * 2 * g.plane * 8 + (size_t)g.grid * 8;
    if (a.sub == 1) a.ysub = nullptr;
    const int flags = persist_flags();
    auto enqueue = [&]() {
        // planes and XCC_ID granules only: the abort word (first 256 bytes) is STICKY — no launch clears it, so a cluster that gave
        // up in the forward pass of a step is still reported after the backward pass (which shares this workspace) and after any
        // number of replayed steps; the host clears it when it has raised (Encoder.check_persistent)
        (void)hipMemsetAsync(planes, 0, bytes, s);
        switch (g.KSPLIT / (g.NTH / 256)) {
            case 2: launch_fwd<64, 2>(s, a, g, planes, hello, ab, flags); break;
            case 4: launch_fwd<64, 4>(s, a, g, planes, hello, ab, flags); break;
            default: launch_fwd<64, 8>(s, a, g, planes, hello, ab, flags); break;
        }
    };
    GraphKey key("bigru_pfwd");
    key.add(&a, sizeof(a));
    key.add(&g.RB, sizeof(g.RB));
    key.add(&g.NTH, sizeof(g.NTH));
    key.add(&flags, sizeof(flags));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_fwd(persistent)");
}

int lvsr_bigru_bwd_persistent(hipStream_t s, const EncBwd0& a, int use_graph) {
    PersistGeom g;
    LVSR_REQUIRE(persist_geom(a.B, a.H, g) && a.sync_ws, "lvsr_bigru_bwd: persistent mode not available for B=%d H=%d", a.B, a.H);
    int* ab = (int*)a.sync_ws;
    u64* planes = (u64*)((char*)a.sync_ws + 256);
    u64* hello = planes + (size_t)2 * g.rt * 4 * g.plane;
    const size_t bytes = (size_t)2 * g.rt * 4 * g.plane * 8 + (size_t)g.grid * 8;
    const int Bp = ((a.B + 15) / 16) * 16;
    float* dh = a.dh_ws;
    const int flags = persist_flags();
    auto enqueue = [&]() {
        (void)hipMemsetAsync(planes, 0, bytes, s);          // not the abort word: sticky until the host has seen it (see the forward)
        switch (g.KSPLIT / (g.NTH / 256)) {
            case 2: launch_bwd<64, 2>(s, a, g, planes, hello, ab, dh, Bp, flags); break;
            case 4: launch_bwd<64, 4>(s, a, g, planes, hello, ab, dh, Bp, flags); break;
            default: launch_bwd<64, 8>(s, a, g, planes, hello, ab, dh, Bp, flags); break;
        }
        hipLaunchKernelGGL(enc_pbwd_h0_kernel, dim3((a.H + 255) / 256, 1, 2), dim3(256), 0, s, dh, Bp, a.B, a.H, a.dh0[0], a.dh0[1]);
    };
    GraphKey key("bigru_pbwd");
    key.add(&a, sizeof(a));
    key.add(&g.RB, sizeof(g.RB));
    key.add(&g.NTH, sizeof(g.NTH));
    key.add(&flags, sizeof(flags));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_bwd(persistent)");
}
