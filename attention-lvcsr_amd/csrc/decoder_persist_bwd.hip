// Persistent attention-decoder BACKWARD for gfx950: one launch for the reverse walk over the labels instead of four per label.
// Same cluster layout as the forward (decoder_persist.hip, decoder_persist.h PdShape): P = ceil(D/UNITS) work-groups of 512 threads
// serve ONE utterance, work-group p owns decoder units [UNITS p, UNITS p + UNITS), attended positions t = p mod P and match columns
// [MS p, MS p + MS), MS = MC UNITS (clusters of 8: 32 units / 64 columns; of 16: 16 / 32; of up to 32 at D <= 512: 16 / 16).
//
// Per label i = L-1 .. 0 (math as decoder_bwd.hip; AW / QR as its reassociated form):
//   1. GRU:      dsn = ym ds; dpc = dsn u (1-c^2); dpu = dsn (c-s) u (1-u)                          (own units, elementwise)
//                A: all-gather dpc;  drh = dpc @ Whh^T (own units: row of Whh in registers);  dpr = drh s r (1-r)
//   2. gates:    B: all-gather [dpu | dpr];  dsacc = dsn(1-u) + (1-ym)ds + drh r + [dpu|dpr] @ Whg^T + dS_readout
//      glimpse:  q[t] = [dpc|dpu|dpr] . AW[t] + QR[i,t] + dalpha[t]   for the OWN positions (AW rows streamed from L2, fetched
//                behind exchange B);  C: all-gather q;  sd = sum_t alpha_t q_t;  de = alpha (q - sd)          (softmax backward)
//   3. energies: own positions x all match columns on the matrix cores (as attbwd_energy_mfma_kernel): match = PA + sW + cv^T H,
//                dm = de w_e (1 - tanh^2); dPA += dm; dcv = dm H^T (complete: all columns are local); the handler operands
//                and the handler / energy-vector gradient accumulators (over the whole label loop) are per-lane registers; dsW
//                partial over the own positions
//   4. state:    D: reduce-scatter dsW (every work-group publishes 512 partials, gathers the 8 x 64 of its column slice);
//                its slice's contribution dsW[slice] @ Ws^T[slice] to ALL units (the Ws column slice is LDS resident);
//                E: reduce-scatter of those (8 x 32 gathered)
//                ds' = dsacc + sum                                                                     -> next label
//   5. alignment (off the chain, behind D / E): dalpha'[s] = sum_k sum_{own t} dcv[k][t] f[k][c+t-s] for all s;
//                F: reduce-scatter, gathered by the owners of s at the start of the next label.
// Five exchanges on the chain per label (A, B, C, D, E) instead of four kernel boundaries + four kernels.
// Writes what the batched GEMMs after the loop need (DXG, DSW, DCV, dPA, ds) and per-work-group partials of the handler /
// energy-vector / energy-bias gradients (accH, accWe, accEb: B*P rows, folded by lvsr_colsum).  Limits as the forward's.
#include "decoder_persist.h"
#include <stdlib.h>
#include <string.h>

typedef lvsr_attdec_bwd_args AttBwd;

// granule planes of an utterance: A (dpc: 512) | B (dpu | dpr: 2 x 512) | C (q: 512) | XCC_ID granules (64), then per work-group
// D (dsW partials: 512) | E (state-gradient contributions: 512) | F (alignment-gradient partials: 512)
#define PB_SMALL (4 * PD_MAXV + 64)
#define PB_PERWG (3 * 512)
// two-layer launch (lvsr_attdec_bwd_persistent_stack2): + A1 (512) | B1 (2 x 512) | X0 (512: fork_1 part of layer 0's state gradient)
// | Q1 (512: layer 1's share of the glimpse gradient) behind the small planes, and a fourth per-work-group plane E1
#define PB_STACK_EXTRA (5 * PD_MAXV)
#define PB_PERWG_STACK (4 * 512)

struct PbGeom {
    int P, nown, nownp, KC, KCP, FW, RL, AWL, AWS, FTL, shape, NTL, DPAL, DPS, o_dpa, YMD;
    int o_ft, o_nx, o_pa, o_cv, o_dcv, o_al, o_q, o_des, o_dalp, o_dgl, o_dpc, o_dpu, o_dpr, o_dms, o_r8, o_r8b, o_dsw, o_ws, o_aw, o_red, o_clk, prof, total;
    int nb, b0;          // utterances of this launch: [b0, b0 + nb) (pd_pick_passes)
};

static int pb_kc(int K) {
    const int inst[4] = {0, 4, 10, 16};
    for (int i = 0; i < 4; ++i)
        if (K <= inst[i]) return inst[i];
    return -1;
}

static bool pb_geom(const AttDec& a, PbGeom& g, bool allow16 = true, bool stack = false) {
    if ((a.phases & 3) != 3 || a.step_dev != nullptr || a.group_rows != 0) return false;
    if (a.M > PD_MAXV || a.Tp > PD_MAXV) return false;
    g.KC = pb_kc(a.K);
    if (g.KC < 0) return false;
    g.KCP = (g.KC + 3) / 4 * 4;
    PdPick k;
    if (stack) {          // two clusters of 8 per utterance (PdShape8)
        if (a.D > PdShape8::DMAX || (3 * a.D) % 4 != 0) return false;
        k.shape = 0; k.UNITS = PdShape8::UNITS; k.KSPLIT = PdShape8::KSPLIT; k.KD = PdShape8::KD; k.MC = PdShape8::MC; k.AWS = PdShape8::AWS;
        k.P = (a.D + k.UNITS - 1) / k.UNITS;
        if (a.M > k.P * k.MC * k.UNITS || a.B * 2 * k.P > lvsr_max_cluster_wgs()) return false;
    } else if (!pd_pick_passes(a.B, a.D, a.M, a.K == 0 || a.prior_type == 0, k, g.nb, allow16)) return false;
    if (stack) g.nb = a.B;
    g.b0 = 0;
    g.P = k.P; g.shape = k.shape;
    const int DP = k.KSPLIT * k.KD > 256 ? 512 : 256, MS = k.MC * k.UNITS;      // padded decoder width of the exchanges, match columns per work-group
    g.nown = (a.Tp + g.P - 1) / g.P;
    g.nownp = (g.nown + PD_CH - 1) / PD_CH * PD_CH;
    // the q phase maps one lane group (16 lanes at D <= 256, 32 above) per own position, the alignment-gradient gather one granule
    // row of NTL positions per source work-group (P NTL <= 512 threads)
    if (g.nown > (DP == 256 ? 32 : 16)) return allow16 && k.shape == 1 && !stack ? pb_geom(a, g, false) : false;
    g.NTL = (g.nown <= 16 || DP == 512) ? 16 : 32;
    if (g.P * g.NTL > PD_THREADS) return false;
    g.FW = 2 * a.c + 1;
    g.RL = (g.nown + 3) / 4 * 4;
    if (g.RL % 32 == 0) g.RL += 4;
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) / 4 * 4; return at; };
    g.o_pa = take(g.RL * a.M + 64);
    g.o_cv = take(g.nownp * (g.KCP > 0 ? g.KCP : 4));
    g.o_dcv = take(g.nownp * (g.KCP > 0 ? g.KCP : 4));
    g.o_al = take(a.Tp);
    g.o_q = take(a.Tp);
    g.o_des = take(g.nownp);
    g.o_dalp = take(32);                      // one slot per own position (nown <= 32)
    g.o_dgl = take(3 * DP);
    g.o_dpc = take(k.KSPLIT * (k.KD + 4));
    g.o_dpu = take(k.KSPLIT * (k.KD + 4));
    g.o_dpr = take(k.KSPLIT * (k.KD + 4));
    g.o_dms = take(PD_NW * 16 * 17);
    g.o_r8 = take(512);
    g.o_r8b = take(512);
    g.o_dsw = take(64);
    g.o_ws = take(DP * (MS + 4));                         // Ws[unit][own column slice] (+4 pad per row)
    g.o_red = take(2 * PD_NW);
    g.o_clk = take(2 * (PD_NPROF + 1));
    g.o_nx = take(6 * 64);                    // u | r | c | s | dS_readout of the own units | label mask of the utterance for the next label walked
    // the location filters, transposed [tap][filter] (row = one 16-byte-aligned vector of KCP floats, zero beyond K): resident
    // for the whole walk when they fit; else the alignment correlation reads them row-major (staged per label or from L2)
    g.FTL = g.KCP > 0 && o + g.FW * g.KCP <= PD_LDS_FLOATS;
    g.o_ft = g.FTL ? take(g.FW * g.KCP) : 0;
    // the AW rows of the own positions are the same for every label: resident in LDS when they fit beside the rest (short
    // contexts), else streamed from L2 per label
    g.AWS = (3 * a.D + 3) / 4 * 4 + 4;
    g.AWL = o + g.nown * g.AWS <= PD_LDS_FLOATS;
    g.o_aw = g.AWL ? take(g.nown * g.AWS) : 0;
    // the gradient wrt the preprocessed contexts of the own positions, summed over the labels: every element belongs to ONE lane for the whole
    // walk — by no-return L2 atomics label by label (16 per lane and label), or, OPT-IN (persist_flags PF_DPAL), in LDS (ds_add_f32) and added
    // to the caller's buffer once behind the loop.  Same sums in the same order either way.  Measured (round 6, profiles/r06_decoder_bwd_ab.md):
    // in LDS the WSJ-base step is 0.3 % faster (12.42 against 12.46 ms over four same-box pairs), but the energy phase of WSJ-deep's clusters
    // of 32 takes 10.2 instead of 4.3 us per label, TIMIT-tiny's step 1.88 instead of 1.76 ms, the paper model's 12.43 instead of 12.07:
    // the LDS float atomics are the slower ones wherever a tile's lanes are not all live.  Row stride M + 4: the four position groups of a wave
    // fall on different banks
    g.DPS = (a.M + 3) / 4 * 4 + 4;
    g.DPAL = (lvsr_knob(LVSR_KNOB_PERSIST_FLAGS) & PF_DPAL) && o + g.nown * g.DPS <= PD_LDS_FLOATS;
    g.o_dpa = g.DPAL ? take(g.nown * g.DPS) : 0;
    g.YMD = (lvsr_knob(LVSR_KNOB_PERSIST_FLAGS) & PF_NOYMPRE) != 0;
    g.total = o;
    g.prof = lvsr_knob(LVSR_KNOB_PHASE_CLOCK);
    if (o > PD_LDS_FLOATS && k.shape == 1 && allow16 && !stack) return pb_geom(a, g, false);      // clusters of 8 instead, if they fit
    return o <= PD_LDS_FLOATS;
}

// Global accesses as UNIFORM base (scalar registers) + 32-bit per-lane byte offset: the loop-invariant per-lane parts are then one
// register each (as 64-bit pointers they were hoisted out of the label loop two registers apiece and spilled)
template <class T>
__device__ __forceinline__ T pb_ld(const void* sbase, unsigned voff) { return *(const T*)((const char*)sbase + voff); }
template <class T>
__device__ __forceinline__ void pb_st(void* sbase, unsigned voff, T v) { *(T*)((char*)sbase + voff) = v; }

typedef lvsr_attdec_stack2 PbStack;

// The layer-1 cluster of the two-layer reverse walk (lvsr_attdec_bwd_persistent_stack2).  Per label, in walk order: this layer's GRU
// backward (exchanges A1: dpc1, B1: dpu1 | dpr1 inside the cluster) -> X0: the fork_1 part of layer 0's state gradient,
// [dpc1 | dpu1 | dpr1] . F1[unit, :], which the main cluster adds before ITS GRU step of the label -> Q1: this layer's share of the
// glimpse gradient for the own positions, [dpc1 | dpu1 | dpr1] . AW1[t] (the main cluster adds it to its q) -> wait for the main
// cluster's dsW partials of the label (plane D) -> the transform_states#1 part of this layer's state gradient (exchange E1).
template <class SH>
__device__ __forceinline__ void pb_stack_layer1(const AttBwd& gb, const PbGeom& g, const PbStack& k2, float* lds, u64* gbase, int* abort_word,
                                                int b, int p) {
    constexpr int PD_UNITS = SH::UNITS, PD_KSPLIT = SH::KSPLIT, PD_KD = SH::KD;
    constexpr int DP = 256, PMAX = DP / PD_UNITS, MS = SH::MC * PD_UNITS, TPU = PD_THREADS / DP, QL = 16, NB = 2 * DP / PD_THREADS;
    const AttDec& a = gb.f;
    float* const dgl = lds + g.o_dgl;
    float* const dpcs = lds + g.o_dpc;
    float* const dpus = lds + g.o_dpu;
    float* const dprs = lds + g.o_dpr;
    float* const r8 = lds + g.o_r8;
    float* const dsws = lds + g.o_dsw;
    float* const WsL = lds + g.o_ws;
    const int P = g.P, nown = g.nown;
    const int tid0 = threadIdx.x;
    const int D = a.D, M = a.M, Tp = a.Tp, B = a.B, L = a.L, G3 = 3 * a.D;
    const int SLD = a.S_ld ? a.S_ld : 2 * D, DSL = gb.ds_ld ? gb.ds_ld : 2 * D;
    const int AWld = k2.AW1_ld ? k2.AW1_ld : G3, FLD = k2.F1_ld ? k2.F1_ld : G3, G3p = (G3 + 3) & ~3;
    f32x2 whh[PD_KD / 2], whu[PD_KD / 2], whr[PD_KD / 2], wfx[PD_KD / 2], wfu[PD_KD / 2], wfr[PD_KD / 2];
    {
        const int q = tid0 & (PD_KSPLIT - 1), j = p * PD_UNITS + tid0 / PD_KSPLIT;
        const bool junit = j < D;
        const size_t jc = (size_t)min(j, D - 1);
#pragma unroll
        for (int x = 0; x < PD_KD / 2; ++x) {
            float v[6][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = q * PD_KD + 2 * x + e;
                const size_t kc = (size_t)min(k, D - 1);
                const float keep = (junit && k < D) ? 1.f : 0.f;
                v[0][e] = k2.Whh1[jc * D + kc] * keep;
                v[1][e] = k2.Whg1[jc * 2 * D + kc] * keep;
                v[2][e] = k2.Whg1[jc * 2 * D + D + kc] * keep;
                v[3][e] = k2.F1[jc * FLD + kc] * keep;                // row `unit` of fork_1: [to x | to u | to r] of layer 1's units k
                v[4][e] = k2.F1[jc * FLD + D + kc] * keep;
                v[5][e] = k2.F1[jc * FLD + 2 * D + kc] * keep;
            }
            whh[x] = (f32x2){v[0][0], v[0][1]}; whu[x] = (f32x2){v[1][0], v[1][1]}; whr[x] = (f32x2){v[2][0], v[2][1]};
            wfx[x] = (f32x2){v[3][0], v[3][1]}; wfu[x] = (f32x2){v[4][0], v[4][1]}; wfr[x] = (f32x2){v[5][0], v[5][1]};
        }
    }
    for (int x = tid0; x < g.total; x += PD_THREADS) lds[x] = 0.f;
    __syncthreads();
    for (int x = tid0; x < DP * MS; x += PD_THREADS) {
        const int kp = x / MS, mm = x % MS, m = p * MS + mm;
        WsL[kp * (MS + 4) + mm] = (kp < D && m < M) ? k2.Ws1[(size_t)kp * M + m] : 0.f;
    }
    float dsj;
    {
        const int j = p * PD_UNITS + tid0 / PD_KSPLIT;
        dsj = j < D ? gb.ds[(size_t)b * DSL + D + j] : 0.f;
    }
    u64* const gA1 = gbase + PB_SMALL;
    u64* const gB1 = gA1 + PD_MAXV;
    u64* const gX0 = gA1 + 3 * PD_MAXV;
    u64* const gQ1 = gA1 + 4 * PD_MAXV;
    u64* const gD = gbase + PB_SMALL + PB_STACK_EXTRA;
    u64* const gE1 = gD + (size_t)3 * P * 512;
    const bool plain = cluster_shares_xcd(gbase + 4 * PD_MAXV, 2 * P, P + p, abort_word);
    __syncthreads();
    for (int n = 0; n < L; ++n) {
        const int i = L - 1 - n;
        const unsigned epoch = (unsigned)(n + 1);
        const int tid = lvsr_unhoisted((int)threadIdx.x), q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT;
        const int j = p * PD_UNITS + jl;
        const bool junit = j < D;
        const size_t row = (size_t)i * B + b;
        const size_t jc = (size_t)min(j, D - 1);
        const Win wi = attdec_window(a, i);
        const float uu = junit ? k2.U1[row * D + jc] : 0.f, rr = junit ? k2.R1[row * D + jc] : 0.f, cc = junit ? k2.C1[row * D + jc] : 0.f;
        const float sp = junit ? a.S[row * SLD + D + jc] : 0.f;
        const float dsr = (junit && gb.dS_r) ? gb.dS_r[row * SLD + D + jc] : 0.f;
        const float ym = a.ymask ? a.ymask[row] : 1.f;
        // ---- GRU backward of this layer
        const float dsn = ym * dsj;
        const float dpc = junit ? dsn * uu * (1.f - cc * cc) : 0.f;
        const float dpu = junit ? dsn * (cc - sp) * uu * (1.f - uu) : 0.f;
        float part = dsn * (1.f - uu) + (1.f - ym) * dsj + dsr;
        if (q == 0 && junit) granule_store(gA1 + j, epoch, dpc, plain);
        {
            float v[PD_NV];
            if (!pd_gather(gA1, D, epoch, abort_word, v)) return;
            if (tid < D) { dpcs[pd_slot(tid, PD_KD)] = v[0]; dgl[tid] = v[0]; }
        }
        __syncthreads();
        const float drh = pd_dot<PD_KD, PD_KSPLIT>(whh, dpcs, q);
        const float dpr = junit ? drh * sp * rr * (1.f - rr) : 0.f;
        part += drh * rr;
        if (q == 0 && junit) {
            granule_store(gB1 + j, epoch, dpu, plain);
            granule_store(gB1 + DP + j, epoch, dpr, plain);
            float* dx = k2.DXG1 + row * G3;
            dx[j] = dpc; dx[D + j] = dpu; dx[2 * D + j] = dpr;
        }
        {
            u64 wv[NB];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    const int gi = tid + e * PD_THREADS;
                    wv[e] = (u64)epoch << 32;
                    if (gi % DP < D) wv[e] = __hip_atomic_load(gB1 + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(wv[e] >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int gi = tid + e * PD_THREADS, u = gi % DP;
                if (u < D) {
                    const float v = __uint_as_float((unsigned)wv[e]);
                    if (gi < DP) { dpus[pd_slot(u, PD_KD)] = v; dgl[D + u] = v; }
                    else { dprs[pd_slot(u, PD_KD)] = v; dgl[2 * D + u] = v; }
                }
            }
        }
        __syncthreads();
        // ---- X0: what the state of layer 0 this label produced receives through fork_1 (the main cluster waits for it)
        {
            const float x0 = pd_dot<PD_KD, PD_KSPLIT>(wfx, dpcs, q) + pd_dot<PD_KD, PD_KSPLIT>(wfu, dpus, q) + pd_dot<PD_KD, PD_KSPLIT>(wfr, dprs, q);
            if (q == 0 && junit) granule_store(gX0 + j, epoch, x0, plain);
        }
        const float dsacc = part + pd_dot<PD_KD, PD_KSPLIT>(whu, dpus, q) + pd_dot<PD_KD, PD_KSPLIT>(whr, dprs, q);
        // ---- Q1: this layer's share of q for the own positions
        {
            const int qtl = tid / QL, l16 = tid % QL, qt = qtl * P + p;
            const bool qok = qtl < nown && qt < Tp && qt >= wi.begin && qt < wi.end;
            const unsigned awoff = 4u * ((unsigned)min(qt, Tp - 1) * (unsigned)(B * AWld) + 4u * l16);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int col = 4 * l16 + 4 * QL * e;
                const float4 aw = (qok && col < G3p) ? pb_ld<float4>(k2.AW1 + (size_t)b * AWld + 4 * QL * e, awoff) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 dg = (col < G3p) ? *(const float4*)(dgl + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                s0 += aw.x * dg.x + aw.y * dg.y;
                s1 += aw.z * dg.z + aw.w * dg.w;
            }
            const float qs = group_sum<QL>(s0 + s1);
            if (l16 == 0 && qtl < nown && qt < Tp) granule_store(gQ1 + qt, epoch, qok ? qs : 0.f, plain);
        }
        // ---- transform_states#1 part of this layer's state gradient, from the main cluster's dsW partials of the label
        {
            const int src = tid / MS, mm = tid % MS;
            const bool mine = src < P && p * MS + mm < M;
            u64 wv = (u64)epoch << 32;
            unsigned spins = 0;
            for (;;) {
                if (mine) wv = __hip_atomic_load(gD + (size_t)src * 512 + p * MS + mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (unsigned)(wv >> 32) == epoch)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            r8[tid] = mine ? __uint_as_float((unsigned)wv) : 0.f;
        }
        __syncthreads();
        if (tid < MS) {
            float s = 0.f;
#pragma unroll
            for (int src = 0; src < PMAX; ++src) s += r8[src * MS + tid];
            dsws[tid] = s;
        }
        __syncthreads();
        {
            constexpr int NC = MS / TPU, NC4 = NC / 4;
            const int ku = tid / TPU, partn = tid % TPU;
            const float4* dv = (const float4*)(dsws + partn * NC);
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int x = 0; x < NC4; ++x) {
                const float4 d4 = dv[x], w4 = *(const float4*)(WsL + ku * (MS + 4) + partn * NC + 4 * x);
                s0 += d4.x * w4.x + d4.y * w4.y;
                s1 += d4.z * w4.z + d4.w * w4.w;
            }
            float s = s0 + s1;
            if (TPU == 2) s += lvsr_dpp_quad_xor1(s);
            if (partn == 0 && ku < D) granule_store(gE1 + (size_t)p * 512 + ku, epoch, s, plain);
        }
        {
            const int src = tid / PD_UNITS, uu_ = tid % PD_UNITS;
            const bool mine = tid < DP && src < P && p * PD_UNITS + uu_ < D;
            u64 wv = (u64)epoch << 32;
            unsigned spins = 0;
            for (;;) {
                if (mine) wv = __hip_atomic_load(gE1 + (size_t)src * 512 + p * PD_UNITS + uu_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (unsigned)(wv >> 32) == epoch)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            __syncthreads();                      // (every thread is past its reads of r8 above)
            if (tid < DP) r8[tid] = mine ? __uint_as_float((unsigned)wv) : 0.f;
        }
        __syncthreads();
        {
            float s = 0.f;
#pragma unroll
            for (int src = 0; src < PMAX; ++src) s += r8[src * PD_UNITS + jl];
            dsj = junit ? dsacc + s : 0.f;
        }
        __syncthreads();
    }
    {
        const int q = tid0 & (PD_KSPLIT - 1), j = p * PD_UNITS + tid0 / PD_KSPLIT;
        if (q == 0 && j < D) gb.ds[(size_t)b * DSL + D + j] = dsj;
    }
}

template <int KC, class SH, bool STACK = false>
__global__ __launch_bounds__(PD_THREADS) void attdec_pbwd_kernel(AttBwd gb, lvsr_attdec_plain w, PbGeom g, u64* planes, int* abort_word, PbStack k2) {
    constexpr int KCP = (KC + 3) / 4 * 4;
    constexpr int PD_UNITS = SH::UNITS, PD_KSPLIT = SH::KSPLIT, PD_KD = SH::KD;
    constexpr int DP = SH::DMAX > 256 ? 512 : 256;        // padded decoder width: planes A / B / E, rows of the Ws slice
    constexpr int PMAX = DP / PD_UNITS;                    // work-groups per cluster at most (8 / 16 / 32)
    constexpr int MS = SH::MC * PD_UNITS;                  // match columns per work-group (64 / 32 / 16): PMAX MS = 512
    constexpr int TPU = PD_THREADS / DP;                   // threads per unit in the Ws^T contribution (2 / 1)
    constexpr int QL = DP == 256 ? 16 : 32;                // lanes per own position in the q contraction: 12 float4 of the 3 D columns each
    constexpr int NB = 2 * DP / PD_THREADS;                // granules per thread in exchange B (dpu | dpr)
    constexpr int NS = KCP / 4 > 0 ? KCP / 4 : 1;
    const AttDec& a = gb.f;
    __shared__ __attribute__((aligned(16))) float lds[PD_LDS_FLOATS];
    float* const PAs = lds + g.o_pa;      // [M][RL] own positions of the preprocessed attended, transposed, pre-scaled by C2
    float* const cvs = lds + g.o_cv;      // [nownp][KCP] convolution features of the own positions (this label)
    float* const dcvs = lds + g.o_dcv;    // [nownp][KCP] their gradient
    float* const al = lds + g.o_al;       // [T'] alignment produced by this label
    float* const qv = lds + g.o_q;        // [T'] q of all positions
    float* const des = lds + g.o_des;     // [nownp] energy gradients of the own positions
    float* const dgl = lds + g.o_dgl;     // [3 D] dpc | dpu | dpr, as the columns of AW
    float* const dpcs = lds + g.o_dpc;    // sliced copies for the register contractions
    float* const dpus = lds + g.o_dpu;
    float* const dprs = lds + g.o_dpr;
    float* const dms = lds + g.o_dms;     // [PD_NW][16][17] dm tile of a wave; later the cross-wave dcv partials
    float* const r8 = lds + g.o_r8;       // [PMAX][MS] / [PMAX][UNITS]: gathered partials of exchanges D / E
    float* const r8b = lds + g.o_r8b;     // [PMAX][NTL]: gathered partials of exchange F
    float* const dsws = lds + g.o_dsw;    // [MS] dsW of the own column slice
    float* const WsL = lds + g.o_ws;      // [DP][MS + 4] transform_states rows x the own match columns
    float* const AWl = lds + g.o_aw;      // AWL: [nown][AWS] rows of AW of the own positions
    float* const red = lds + g.o_red;
    float* const fT = lds + g.o_ft;       // FTL: [FW][KCP] conv1d.filters, transposed
    float* const nx = lds + g.o_nx;       // [6][64] saved gate values of the own units + the label mask, fetched one label ahead
    float* const dpas = lds + g.o_dpa;    // DPAL: [nown][DPS] gradient wrt the preprocessed contexts of the own positions, summed over the labels
    const int P = g.P, nown = g.nown;
    int b, p;
    if (!cluster_of_block(STACK ? 2 * P : P, g.nb, 0, b, p)) return;  // (work-groups of the grid's padding)
    b += g.b0;
    u64* const gA = planes + (size_t)b * (STACK ? PB_SMALL + PB_STACK_EXTRA + (size_t)P * PB_PERWG_STACK : PB_SMALL + (size_t)P * PB_PERWG);
    if (STACK && p >= P) {            // the second cluster of the utterance: layer 1 of the stack
        pb_stack_layer1<SH>(gb, g, k2, lds, gA, abort_word, b, p - P);
        return;
    }
    const int tid = threadIdx.x, q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const int D = a.D, M = a.M, Tp = a.Tp, B = a.B, K = a.K, L = a.L, G3 = 3 * a.D;
    // row strides of the state slots / the readout's state gradient and of the running state gradient (two-layer launch: 2 D, this
    // layer in front)
    const int SLD = a.S_ld ? a.S_ld : D, DSL = gb.ds_ld ? gb.ds_ld : D;
    // AW rows are read 16 bytes at a time: row stride AWld (a multiple of 4, the host pads when 3D is not), columns up to G3p;
    // the gradient vector they are contracted with is zero beyond 3D
    const int AWld = w.AW_ld ? w.AW_ld : G3, G3p = (G3 + 3) & ~3;
    const int j = p * PD_UNITS + jl;
    const bool junit = j < D;
    const float C2 = 2.885390081777927f;
    // ---- register-resident weights
    f32x2 whh[PD_KD / 2], whu[PD_KD / 2], whr[PD_KD / 2];        // rows j of Whh, Whg[:, :D], Whg[:, D:], slice q of the columns
    {
        const size_t jc = (size_t)min(j, D - 1);
#pragma unroll
        for (int x = 0; x < PD_KD / 2; ++x) {
            float v[3][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = q * PD_KD + 2 * x + e;
                const size_t kc = (size_t)min(k, D - 1);
                const float keep = (junit && k < D) ? 1.f : 0.f;
                v[0][e] = w.Whh[jc * D + kc] * keep;
                v[1][e] = w.Whg[jc * 2 * D + kc] * keep;
                v[2][e] = w.Whg[jc * 2 * D + D + kc] * keep;
            }
            whh[x] = (f32x2){v[0][0], v[0][1]}; whu[x] = (f32x2){v[1][0], v[1][1]}; whr[x] = (f32x2){v[2][0], v[2][1]};
        }
    }
    // MFMA operands of the energy phase: this wave's match columns [64 wave, 64 wave + 64) as four 16-column tiles.  The handler
    // operands and the handler-gradient accumulators are per-lane registers (44 on top of the 48 of the state weights; keeping
    // them in LDS instead was measured slower: 35.5 vs 33.4 us per label, profiles/r03_decoder_bwd_persist_probe.txt)
    float Hb[4][NS], Ht[4][4];
    f32x4 dH[4];
    float wet[4], weacc[4], ebacc = 0.f;
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int m = (4 * wave + tile) * 16 + c16;
#pragma unroll
        for (int sq = 0; sq < NS; ++sq) {
            const int k = 4 * sq + g4;
            Hb[tile][sq] = (KC > 0 && k < K && m < M) ? C2 * a.handler[(size_t)k * M + m] : 0.f;
        }
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
            const int mm = (4 * wave + tile) * 16 + 4 * sq + g4;
            Ht[tile][sq] = (KC > 0 && c16 < K && mm < M) ? a.handler[(size_t)c16 * M + mm] : 0.f;
        }
        dH[tile] = F32X4_ZERO;
        wet[tile] = m < M ? a.w_e[m] : 0.f;
        weacc[tile] = 0.f;
    }
    // ---- LDS residents
    for (int x = tid; x < g.total; x += PD_THREADS) lds[x] = 0.f;
    __syncthreads();
    for (int x = tid; x < nown * M; x += PD_THREADS) {
        const int tl = x / M, m = x % M, t = tl * P + p;
        if (t < Tp) PAs[m * g.RL + tl] = C2 * a.PA[(size_t)t * a.PA_ts + (size_t)b * a.PA_bs + m];
    }
    if (g.AWL)
        for (int x = tid; x < nown * G3; x += PD_THREADS) {
            const int tl = x / G3, col = x % G3, t = tl * P + p;
            AWl[tl * g.AWS + col] = t < Tp ? w.AW[((size_t)t * B + b) * AWld + col] : 0.f;
        }
    for (int x = tid; x < DP * MS; x += PD_THREADS) {
        const int kp = x / MS, mm = x % MS, m = p * MS + mm;
        WsL[kp * (MS + 4) + mm] = (kp < D && m < M) ? w.Ws[(size_t)kp * M + m] : 0.f;
    }
    if (KC > 0 && g.FTL)
        for (int x = tid; x < a.K * g.FW; x += PD_THREADS) fT[(x % g.FW) * KCP + x / g.FW] = a.filters[x];
    float dsj = junit ? gb.ds[(size_t)b * DSL + j] : 0.f;          // running gradient wrt the state (caller: zeros + readout part)
    u64* const gB = gA + PD_MAXV;                     // [2][DP]
    u64* const gC = gA + 3 * PD_MAXV;
    u64* const gX0 = gA + PB_SMALL + 3 * PD_MAXV;     // (two-layer launch) from the layer-1 cluster: fork_1 part of this layer's state gradient
    u64* const gQ1 = gA + PB_SMALL + 4 * PD_MAXV;     // ... and its share of q
    u64* const gD = gA + PB_SMALL + (STACK ? PB_STACK_EXTRA : 0);     // [P][512]
    u64* const gE = gD + (size_t)P * 512;             // [P][512] (DP used)
    u64* const gF = gE + (size_t)P * 512;             // [P][512]
    const bool plain = cluster_shares_xcd(gA + 4 * PD_MAXV, STACK ? 2 * P : P, p, abort_word);
    __syncthreads();
    PdClock clk;
    clk.start(g.prof != 0 && b == g.b0 && p == min(g.prof, P) - 1 && tid == 0, lds + g.o_clk);        // (knob phase_clock = 1 + the work-group of the first cluster that keeps the clock)

    for (int n = 0; n < L; ++n) {
        const int i = L - 1 - n;
        const unsigned epoch = (unsigned)(n + 1);
        // the per-thread indices are re-derived per label from an opaque copy of the thread id: derived from the loop-invariant one,
        // every address / predicate built on them is hoisted out of the label loop and the hoisted values are what gets spilled
        const int tid = lvsr_unhoisted((int)threadIdx.x), q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63, c16 = lane & 15, g4 = lane >> 4;
        const int j = p * PD_UNITS + jl;
        const bool junit = j < D;
        const unsigned jb = 4u * (unsigned)min(j, D - 1);
        const size_t row = (size_t)i * B + b;
        const Win wi = attdec_window(a, i);
        // ---- this label's saved values
        float uu, rr, cc, sp, dsr;
        if (STACK) {
            // the layer above hands down what this layer's state of the label receives through fork_1: every lane polls the granule
            // of its unit (before any other load of the label is issued: a poll behind a load waits for it)
            u64 wv;
            unsigned spins = 0;
            for (;;) {
                wv = __hip_atomic_load(gX0 + min(j, D - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(wv >> 32) == epoch)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            if (junit) dsj += __uint_as_float((unsigned)wv);
        }
        if (n == 0) {
            uu = junit ? pb_ld<float>(a.U + row * D, jb) : 0.f; rr = junit ? pb_ld<float>(a.R + row * D, jb) : 0.f;
            cc = junit ? pb_ld<float>(a.C + row * D, jb) : 0.f; sp = junit ? pb_ld<float>(a.S + row * SLD, jb) : 0.f;
            dsr = (junit && gb.dS_r) ? pb_ld<float>(gb.dS_r + row * SLD, jb) : 0.f;
        } else {       // fetched into LDS behind the previous label's energy phase (below)
            uu = junit ? nx[jl] : 0.f; rr = junit ? nx[64 + jl] : 0.f; cc = junit ? nx[128 + jl] : 0.f; sp = junit ? nx[192 + jl] : 0.f;
            dsr = junit ? nx[256 + jl] : 0.f;
        }
        // the alignment row and the convolution features (first used behind exchange B) are fetched by waves 4-7, which poll neither
        // below (the alignment-gradient gather) nor in exchange A (D <= 256 granules) — a poll issued behind these loads would wait for
        // them (in-order returns) — and are kept in registers until exchange B is over: nothing on the way waits for their latency
        float alv[2] = {0.f, 0.f}, cvv[2] = {0.f, 0.f}, swv[2] = {0.f, 0.f};      // swv: the label's transformed states (energy phase)
        if (tid >= PD_THREADS / 2) {
            const int t2 = tid - PD_THREADS / 2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = t2 + e * (PD_THREADS / 2);
                if (t < Tp) alv[e] = a.W[((size_t)(i + 1) * B + b) * Tp + t];
                if (t < M) swv[e] = a.sW[row * M + t];
                if (KC > 0 && t < nown * K) {
                    const int tl = t / K, k = t % K, tt = tl * P + p;
                    if (tt < Tp) cvv[e] = a.CV[((row * K) + k) * Tp + tt];
                }
            }
        }
        // (the label mask opens the chain — dsn, dpc, the first publish: fetched here it cost every wave a memory round trip per label, and
        // waves 4-7 the latency of the alignment row / feature loads above, which sit in front of it in their in-order queues; it comes
        // with the saved gate values now, one label ahead)
        const float ym = a.ymask ? ((n == 0 || g.YMD) ? a.ymask[row] : nx[320]) : 1.f;
        // ---- 1. GRU
        const float dsn = ym * dsj;
        const float dpc = junit ? dsn * uu * (1.f - cc * cc) : 0.f;
        const float dpu = junit ? dsn * (cc - sp) * uu * (1.f - uu) : 0.f;
        float part = dsn * (1.f - uu) + (1.f - ym) * dsj + dsr;
        if (q == 0 && junit) granule_store(gA + j, epoch, dpc, plain);
        // gradient wrt the alignment this label produced: gathered from the partial correlations of the previous iteration
        if (n > 0 && KC > 0) {
            const int src = tid / g.NTL, tl = tid % g.NTL, t = tl * P + p;
            const bool mine = src < P && tl < nown && t < Tp;
            // (granules of positions this work-group does not own are not waited for: sweep only the valid ones)
            u64 wv = (u64)epoch << 32;
            unsigned spins = 0;
            for (;;) {
                if (mine) wv = __hip_atomic_load(gF + (size_t)src * 512 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (unsigned)(wv >> 32) == (unsigned)n)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            r8b[tid] = mine ? __uint_as_float((unsigned)wv) : 0.f;
        }
        clk.mark(0);
        {
            float v[PD_NV];
            if (!pd_gather(gA, D, epoch, abort_word, v)) return;
            if (tid < D) { dpcs[pd_slot(tid, PD_KD)] = v[0]; dgl[tid] = v[0]; }
        }
        __syncthreads();
        clk.mark(1);
        const float drh = pd_dot<PD_KD, PD_KSPLIT>(whh, dpcs, q);
        const float dpr = junit ? drh * sp * rr * (1.f - rr) : 0.f;
        part += drh * rr;
        if (q == 0 && junit) {
            granule_store(gB + j, epoch, dpu, plain);
            granule_store(gB + DP + j, epoch, dpr, plain);
            float* dx = gb.DXG + row * G3;
            pb_st<float>(dx, jb, dpc); pb_st<float>(dx + D, jb, dpu); pb_st<float>(dx + 2 * D, jb, dpr);
        }
        clk.mark(2);
        // ---- 2. gates
        {
            // granule g = tid + 512 e of the 2 DP of the plane: [dpu | dpr]
            u64 wv[NB];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    const int gi = tid + e * PD_THREADS;
                    wv[e] = (u64)epoch << 32;
                    if (gi % DP < D) wv[e] = __hip_atomic_load(gB + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(wv[e] >> 32) == epoch;
                }
                if (__all(ok)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int gi = tid + e * PD_THREADS, u = gi % DP;
                if (u < D) {
                    const float v = __uint_as_float((unsigned)wv[e]);
                    if (gi < DP) { dpus[pd_slot(u, PD_KD)] = v; dgl[D + u] = v; }
                    else { dprs[pd_slot(u, PD_KD)] = v; dgl[2 * D + u] = v; }
                }
            }
        }
        if (tid >= PD_THREADS / 2) {
            const int t2 = tid - PD_THREADS / 2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int t = t2 + e * (PD_THREADS / 2);
                if (t < Tp) al[t] = alv[e];
                if (KC > 0 && t < nown * K) cvs[(t / K) * KCP + t % K] = cvv[e];
            }
        }
        __syncthreads();
        clk.mark(3);
        // AW rows of the own positions for q: thread (tl = tid / 16, l16): columns 4 l16 + 64 e.  Issued AFTER exchange B: a sweep of
        // the exchange would queue behind them in this wave's memory pipeline (loads return in order) and the exchange would
        // last as long as they do
        const int qtl = tid / QL, l16 = tid % QL, qt = qtl * P + p;
        const bool qok = qtl < nown && qt < Tp && qt >= wi.begin && qt < wi.end;
        const unsigned awoff = 4u * ((unsigned)min(qt, Tp - 1) * (unsigned)(B * AWld) + 4u * l16);
        float4 awv[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int col = 4 * l16 + 4 * QL * e;
            awv[e] = (!g.AWL && qok && col < G3p) ? pb_ld<float4>(w.AW + (size_t)b * AWld + 4 * QL * e, awoff) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float dsacc = part + pd_dot<PD_KD, PD_KSPLIT>(whu, dpus, q) + pd_dot<PD_KD, PD_KSPLIT>(whr, dprs, q);
        clk.mark(11);
        {
            // q of the own positions
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int col = 4 * l16 + 4 * QL * e;             // columns [0,D) dpc, [D,2D) dpu, [2D,3D) dpr: dgl is laid out the same
                const float4 dg = (col < G3p) ? *(const float4*)(dgl + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 aw = !g.AWL ? awv[e] : (qok && col < G3p) ? *(const float4*)(AWl + qtl * g.AWS + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                s0 += aw.x * dg.x + aw.y * dg.y;
                s1 += aw.z * dg.z + aw.w * dg.w;
            }
            clk.mark(12);
            // + the gradient wrt the alignment this label produced: the partials gathered at the top of the label, row src = l16 of the
            // position's column — the lane group's fold IS the sum over the cluster's work-groups (P <= QL; rows of absent work-groups
            // hold zeros; positions this work-group does not own read a neighbour's column and are dropped below).  As a loop of 16
            // lanes over the cluster's runtime size in front of exchange B's publish — 16 dependent LDS reads — it held back wave 0's
            // units of the exchange by 0.5 us on every label
            const float fpart = r8b[l16 * g.NTL + qtl];
            float qs = group_sum<QL>(s0 + s1 + ((n > 0 && KC > 0) ? fpart : 0.f));
            if (qok) qs += gb.QR[row * Tp + qt];
            else qs = 0.f;
            if (l16 == 0 && qtl < nown && qt < Tp) granule_store(gC + qt, epoch, qs, plain);
        }
        clk.mark(4);
        {
            float v[PD_NV];
            if (!pd_gather(gC, Tp, epoch, abort_word, v)) return;
            if (STACK) {          // + the layer-1 cluster's share of q
                float v1[PD_NV];
                if (!pd_gather(gQ1, Tp, epoch, abort_word, v1)) return;
                v[0] += v1[0];
            }
            if (tid < Tp) qv[tid] = v[0];
            // sd = sum_t alpha_t q_t over the window
            const bool inw = tid < Tp && tid >= wi.begin && tid < wi.end;
            const float ws_ = wave_sum_dpp(inw ? al[tid] * v[0] : 0.f);
            if (lane == 0) red[wave] = ws_;
        }
        __syncthreads();
        float sd = 0.f;
#pragma unroll
        for (int x = 0; x < PD_NW; ++x) sd += red[x];
        if (tid < g.nownp) {
            const int t = tid * P + p;
            float de = 0.f;
            if (tid < nown && t >= wi.begin && t < wi.end) {
                if (a.normalizer == 0) {
                    de = al[t] * (qv[t] - sd);
                } else {
                    const float e = a.EN[row * Tp + t], Z = a.ZB[row];
                    const float gq = (qv[t] - sd) / Z * attdec_mask(a, i, b, t);
                    if (a.normalizer == 1) { const float sg = sigmoidf_(e); de = gq * sg * (1.f - sg); }
                    else de = e > 0.f ? gq / 1000.f : 0.f;
                }
            }
            des[tid] = de;
        }
        // the transformed states, staged where dpc | dpu | dpr lay (dead since the q contraction, every wave is past it): the energy
        // phase opens with them and would otherwise open with their memory latency
        if (tid >= PD_THREADS / 2) {
            dgl[tid - PD_THREADS / 2] = swv[0];
            dgl[tid] = swv[1];
        }
        __syncthreads();
        if (a.e_bias && tid == 0) {
            float sb = 0.f;
            for (int x = 0; x < nown; ++x) sb += des[x];
            ebacc += sb;
        }
        clk.mark(5);
        // The saved gate values of the label walked next open that label's chain: fetched at its top, a whole HBM / MALL latency sat
        // on every label.  They are fetched HERE instead — straight into LDS (global_load_lds: no register lives through the energy
        // phase, the register-pressure peak of the kernel), one array per wave, and no wave polls before the phase is over (a poll
        // issued behind them would wait for them: vector-memory results return in order)
        if (i > 0 && wave < 6 && lane < PD_UNITS) {
            const float* arr = wave == 0 ? a.U : wave == 1 ? a.R : wave == 2 ? a.C : wave == 3 ? a.S : wave == 4 ? gb.dS_r : a.ymask;
            const size_t at = wave == 5 ? row - (size_t)B : (row - (size_t)B) * (wave >= 3 ? SLD : D) + min(p * PD_UNITS + lane, D - 1);
            if (arr) __builtin_amdgcn_global_load_lds(arr + at, nx + wave * 64, 4, 0, 0);
        }
        // ---- 3. energies backward on the matrix cores
        float swc[4], dsw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int m = (4 * wave + tile) * 16 + c16;
            swc[tile] = m < M ? C2 * dgl[m] : 0.f;
        }
        float* const dmw = dms + wave * 16 * 17;
        for (int tl0 = 0; tl0 < nown; tl0 += PD_CH) {
            float av[NS];
            const int tla = min(tl0 + c16, nown - 1);
#pragma unroll
            for (int sq = 0; sq < NS; ++sq) av[sq] = KC > 0 ? cvs[tla * KCP + 4 * sq + g4] : 0.f;
            float derow[4];
            bool rowok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tl = tl0 + 4 * g4 + r, t = tl * P + p;
                rowok[r] = tl < nown && t < Tp && t >= wi.begin && t < wi.end;
                derow[r] = rowok[r] ? des[tl] : 0.f;
            }
            f32x4 dcva = F32X4_ZERO;
            // the convolution features as the A operand of the handler-gradient product: the same for every tile of the round; read
            // unconditionally (clamped) and masked by a select — a guarded LDS read becomes a branch and serialises behind its check
            float cva[4];
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const int tlk = min(tl0 + 4 * sq + g4, nown - 1);
                const float raw = KC > 0 ? cvs[tlk * KCP + min(c16, KCP - 1)] : 0.f;
                cva[sq] = (c16 < KCP && tl0 + 4 * sq + g4 < nown) ? raw : 0.f;
            }
            // dPA[t][b][m] of (r, tile): base of (b, r, tile) uniform, the lane's (t of r = 0, m of tile 0) part one offset
            const unsigned dpaoff = 4u * ((unsigned)(min(tl0 + 4 * g4, nown - 1) * P + p) * (unsigned)(B * M) + (unsigned)(64 * wave + c16));
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                const int m = (4 * wave + tile) * 16 + c16, mc = min(m, M - 1);
                const float4 pa = *(const float4*)(PAs + mc * g.RL + tl0 + 4 * g4);
                f32x4 acc = (f32x4){pa.x + swc[tile], pa.y + swc[tile], pa.z + swc[tile], pa.w + swc[tile]};
                if (KC > 0) {
#pragma unroll
                    for (int sq = 0; sq < NS; ++sq)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sq], Hb[tile][sq], acc, 0, 0, 0);
                }
                // Straight-line: derow is 0 outside the window / beyond the own positions and wet is 0 beyond M, so d needs no guard
                // and the four tanh chains interleave; only the memory update itself is predicated (an unconditional update with
                // clamped rows was measured: the 0.0 adds of the masked rows pile up on one address and serialise, 6.9 -> 9.3 us)
                float dv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[r]));
                    const float th = 1.0f - 2.0f * rc;
                    dv[r] = derow[r] * wet[tile] * (1.f - th * th);
                    weacc[tile] += derow[r] * th;
                    dsw[tile] += dv[r];
                    dmw[(4 * g4 + r) * 17 + c16] = dv[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // dPA += dm as a no-return L2 atomic: every element belongs to this lane alone (its adds happen in program
                    // order, label by label: deterministic), and no load has to come back before the store can leave
                    if (rowok[r] && m < M) {
                        if (g.DPAL) unsafeAtomicAdd(dpas + (tl0 + 4 * g4 + r) * g.DPS + m, dv[r]);
                        else unsafeAtomicAdd((float*)((char*)(gb.dPA + (size_t)b * M + (size_t)r * P * B * M + 16 * tile) + dpaoff), dv[r]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (KC > 0) {
                    // dcv[pos][k] += dm[pos][m] handler[k][m];  dH[k][m] += cv[pos][k] dm[pos][m]: all eight LDS operands first,
                    // then the eight products back to back
                    float da[4], db[4];
#pragma unroll
                    for (int sq = 0; sq < 4; ++sq) {
                        da[sq] = dmw[c16 * 17 + 4 * sq + g4];
                        db[sq] = dmw[(4 * sq + g4) * 17 + c16];
                    }
                    f32x4 dHt = dH[tile];
#pragma unroll
                    for (int sq = 0; sq < 4; ++sq) {
                        dcva = __builtin_amdgcn_mfma_f32_16x16x4f32(da[sq], Ht[tile][sq], dcva, 0, 0, 0);
                        dHt = __builtin_amdgcn_mfma_f32_16x16x4f32(cva[sq], db[sq], dHt, 0, 0, 0);
                    }
                    dH[tile] = dHt;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (KC > 0) {
                // (a wave's dm tile is its own until the fold below: program order within the wave is all that has to be kept here)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) dmw[(4 * g4 + r) * 17 + c16] = dcva[r];
                __syncthreads();
                if (tid < 256) {
                    const int pos = tid >> 4, k = tid & 15, tl = tl0 + pos, t = tl * P + p;
                    float s = 0.f;
#pragma unroll
                    for (int wv = 0; wv < PD_NW; ++wv) s += dms[(wv * 16 + pos) * 17 + k];
                    if (k < K && tl < nown && t < Tp) {
                        const float v = (t >= wi.begin && t < wi.end) ? s : 0.f;
                        dcvs[tl * KCP + k] = v;
                        gb.DCV[((row * K) + k) * Tp + t] = v;
                    }
                }
                __syncthreads();
            }
        }
        const bool flt = KC > 0 && !g.FTL && K * g.FW <= PD_NW * 16 * 17;
        if (flt && i > 0)
            for (int x = tid; x < K * g.FW; x += PD_THREADS) dms[x] = a.filters[x];      // (published by the barriers of exchange D)
        clk.mark(6);
        // ---- 4. dsW: fold the four position groups of the wave, publish the 512 partials, gather the own column slice
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            dsw[tile] += __shfl_xor(dsw[tile], 16, 64);
            dsw[tile] += __shfl_xor(dsw[tile], 32, 64);
            if (g4 == 0) granule_store(gD + (size_t)p * 512 + (4 * wave + tile) * 16 + c16, epoch, dsw[tile], plain);
        }
        {
            const int src = tid / MS, mm = tid % MS;
            const bool mine = src < P && p * MS + mm < M;
            u64 wv = (u64)epoch << 32;
            unsigned spins = 0;
            for (;;) {
                if (mine) wv = __hip_atomic_load(gD + (size_t)src * 512 + p * MS + mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (unsigned)(wv >> 32) == epoch)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            r8[tid] = mine ? __uint_as_float((unsigned)wv) : 0.f;
        }
        __syncthreads();
        if (tid < MS) {
            float s = 0.f;
#pragma unroll
            for (int src = 0; src < PMAX; ++src) s += r8[src * MS + tid];          // (rows of absent work-groups hold zeros)
            dsws[tid] = s;
            if (p * MS + tid < M) gb.DSW[row * M + p * MS + tid] = s;
        }
        __syncthreads();
        clk.mark(7);
        {
            // contribution of the own column slice to ALL units: thread (unit k' = tid / TPU, part tid % TPU) takes MS / TPU columns
            constexpr int NC = MS / TPU, NC4 = NC / 4;
            const int ku = tid / TPU, part = tid % TPU;
            const float4* dv = (const float4*)(dsws + part * NC);
            float4 wreg[NC4];
#pragma unroll
            for (int x = 0; x < NC4; ++x) {
                wreg[x] = *(const float4*)(WsL + ku * (MS + 4) + part * NC + 4 * x);
            }
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int x = 0; x < NC4; ++x) {
                const float4 d4 = dv[x], w4 = wreg[x];
                s0 += d4.x * w4.x + d4.y * w4.y;
                s1 += d4.z * w4.z + d4.w * w4.w;
            }
            float s = s0 + s1;
            if (TPU == 2) s += lvsr_dpp_quad_xor1(s);
            if (part == 0 && ku < D) granule_store(gE + (size_t)p * 512 + ku, epoch, s, plain);
        }
        clk.mark(8);
        // ---- 5. alignment gradient for the previous label, partial over the own positions: behind exchange E
        if (KC > 0 && i > 0 && g.FTL) {
            // thread (s = tid % 256 [+256], half = tid / 256): ALL filters over the own positions tl = half, half + 2, ...  The
            // filters of a tap and the dcv of a position are each ONE row of KCP floats in LDS: 2 KCP / 4 vector reads per KCP
            // multiply-adds (the row-major form below needed two scalar reads per multiply-add and was LDS-issue bound); lanes
            // step through the tap rows with a 4 KCP-byte stride, which 16-byte reads serve without bank conflicts for KCP = 4, 12
            const int cn = a.c, hsel = tid >> 8;
            for (int s0 = 0; s0 < Tp; s0 += 256) {
                const int sx = s0 + (tid & 255);
                float acc4[4] = {0.f, 0.f, 0.f, 0.f};
                const bool live = sx < Tp && sx >= wi.begin && sx < wi.end;
                const int base = cn + p - sx;                   // f[k][c + t - s] with t = tl P + p: tap base + tl P
#pragma unroll 2
                for (int tl = hsel; tl < nown; tl += 2) {
                    const int idx = base + tl * P;
                    const bool ok = live && (unsigned)idx < (unsigned)g.FW;
                    const float4* fr = (const float4*)(fT + (ok ? idx : 0) * KCP);      // clamped: unconditional loads
                    const float4* dr = (const float4*)(dcvs + tl * KCP);
                    const float keep = ok ? 1.f : 0.f;
#pragma unroll
                    for (int v = 0; v < KCP / 4; ++v) {
                        const float4 f4 = fr[v], d4 = dr[v];
                        acc4[0] += (d4.x * keep) * f4.x;
                        acc4[1] += (d4.y * keep) * f4.y;
                        acc4[2] += (d4.z * keep) * f4.z;
                        acc4[3] += (d4.w * keep) * f4.w;
                    }
                }
                const float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                if (hsel == 1) r8[tid & 255] = acc;            // (barriers outside any lane-dependent branch: a wave counts once)
                __syncthreads();
                if (hsel == 0 && sx < Tp) granule_store(gF + (size_t)p * 512 + sx, epoch, acc + r8[tid & 255], plain);
                __syncthreads();
            }
        } else if (KC > 0 && i > 0) {
            // thread (s = tid % 256 [+256], half = tid / 256): filters k = half, half + 2, ... over ALL own positions.  Straight-line: the
            // tap index c + t - s is tested instead of bounding the loop per lane (lane-dependent trip counts and the integer
            // divisions that set them cost more than the taps outside the filter), four accumulators break the FMA chain
            const int cn = a.c, hsel = tid >> 8;
            const float* fls = flt ? dms : a.filters;          // the filters: staged in the dm tile buffer when they fit
            for (int s0 = 0; s0 < Tp; s0 += 256) {
                const int sx = s0 + (tid & 255);
                float acc4[4] = {0.f, 0.f, 0.f, 0.f};
                const bool live = sx < Tp && sx >= wi.begin && sx < wi.end;
                const int base = cn + p - sx;                   // f[k][c + t - s] with t = tl P + p: index base + tl P
#pragma unroll 4
                for (int tl = 0; tl < nown; ++tl) {
                    const int idx = base + tl * P;
                    const bool ok = live && (unsigned)idx < (unsigned)g.FW;
                    const float* dr = dcvs + tl * KCP + hsel;
                    const float* fp = fls + (ok ? idx : 0) + hsel * g.FW;       // clamped: the loads below are unconditional
                    const float keep = ok ? 1.f : 0.f;                          // (a guarded load serialises behind its check)
#pragma unroll
                    for (int kk = 0; kk < (KC + 1) / 2; ++kk)
                        if (2 * kk + hsel < K) acc4[kk & 3] += (dr[2 * kk] * keep) * fp[2 * kk * g.FW];
                }
                const float acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
                if (hsel == 1) r8[tid & 255] = acc;            // (barriers outside any lane-dependent branch: a wave counts once)
                __syncthreads();
                if (hsel == 0 && sx < Tp) granule_store(gF + (size_t)p * 512 + sx, epoch, acc + r8[tid & 255], plain);
                __syncthreads();
            }
        }
        clk.mark(9);
        {
            const int src = tid / PD_UNITS, uu_ = tid % PD_UNITS;
            const bool mine = tid < DP && src < P && p * PD_UNITS + uu_ < D;
            u64 wv = (u64)epoch << 32;
            unsigned spins = 0;
            for (;;) {
                if (mine) wv = __hip_atomic_load(gE + (size_t)src * 512 + p * PD_UNITS + uu_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(!mine || (unsigned)(wv >> 32) == epoch)) break;
                if (((++spins) & 127u) == 0u) {
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
                    if (spins > PERSIST_SPIN_LIMIT) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
                }
            }
            if (tid < DP) r8[tid] = mine ? __uint_as_float((unsigned)wv) : 0.f;
        }
        __syncthreads();
        {
            float s = 0.f;
#pragma unroll
            for (int src = 0; src < PMAX; ++src) s += r8[src * PD_UNITS + jl];
            dsj = junit ? dsacc + s : 0.f;
        }
        __syncthreads();
        clk.mark(10);
    }
    if (clk.on) {
        long long* out = (long long*)((char*)abort_word + 64);
        for (int x = 0; x < PD_NPROF; ++x) out[x] = clk.acc[1 + x];
    }
    // ---- epilogue: what the caller folds / uses after the loop
    if (q == 0 && junit) gb.ds[(size_t)b * DSL + j] = dsj;
    const size_t prow = (size_t)b * P + p;
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int m = (4 * wave + tile) * 16 + c16;
        float wsum_ = weacc[tile];
        wsum_ += __shfl_xor(wsum_, 16, 64);
        wsum_ += __shfl_xor(wsum_, 32, 64);
        if (g4 == 0 && m < M) gb.accWe[prow * M + m] = wsum_;
        if (KC > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * g4 + r;
                if (k < K && m < M) gb.accH[(prow * K + k) * M + m] = dH[tile][r];
            }
        }
    }
    if (a.e_bias && tid == 0) gb.accEb[prow] = ebacc;
    if (g.DPAL) {          // (the last label's adds of the other waves: the barriers of its exchanges lie in between)
        for (int x = tid; x < nown * M; x += PD_THREADS) {
            const int tl = x / M, m = x % M, t = tl * P + p;
            if (t < Tp) gb.dPA[((size_t)t * B + b) * M + m] += dpas[tl * g.DPS + m];
        }
    }
}

extern "C" long long lvsr_attdec_bwd_persist_ws_bytes(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PbGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pb_geom(a, g)) return 0;
    return 256 + (long long)a.B * (PB_SMALL + (long long)g.P * PB_PERWG) * 8;
}

extern "C" int lvsr_attdec_bwd_persist_clusters(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PbGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pb_geom(a, g)) return 0;
    return g.P;
}

extern "C" int lvsr_attdec_bwd_persistent(void* stream, const lvsr_attdec_bwd_args* args, const lvsr_attdec_plain* plain, void* ws) {
    LVSR_REQUIRE(args != nullptr && plain != nullptr && ws != nullptr, "lvsr_attdec_bwd_persistent: null argument");
    AttBwd gb;
    memcpy(&gb, args, sizeof(gb));
    const AttDec& a = gb.f;
    if (int rc = attdec_check(a, "lvsr_attdec_bwd_persistent")) return rc;
    LVSR_REQUIRE(a.label0 == 0 && (args->parts & 3) % 3 == 0 && (a.S_ld == 0 || a.S_ld == a.D) && (args->ds_ld == 0 || args->ds_ld == a.D),
                 "lvsr_attdec_bwd_persistent: runs all labels and all parts on contiguous states");
    PbGeom g;
    LVSR_REQUIRE((plain->AW_ld ? plain->AW_ld : 3 * a.D) % 4 == 0 && (plain->AW_ld == 0 || plain->AW_ld >= 3 * a.D),
                 "lvsr_attdec_bwd_persistent: the rows of AW must be a multiple of 4 floats apart (pad them: AW_ld)");
    LVSR_REQUIRE(pb_geom(a, g), "lvsr_attdec_bwd_persistent: configuration outside the persistent kernel's limits "
                 "(lvsr_attdec_bwd_persist_ws_bytes returns 0 for it)");
    LVSR_REQUIRE(a.PA_bs == a.M && a.PA_ts == (long long)a.B * a.M, "lvsr_attdec_bwd_persistent: contexts must be contiguous (Tp,B,*)");
    LVSR_REQUIRE(plain->Ws && plain->Whg && plain->Whh && plain->AW && gb.QR && gb.DXG && gb.DSW && gb.dPA && gb.ds && gb.accWe,
                 "lvsr_attdec_bwd_persistent: missing buffers (AW, QR and the plain weights are required)");
    LVSR_REQUIRE(a.K == 0 || (gb.DCV && gb.accH), "lvsr_attdec_bwd_persistent: DCV / accH missing");
    const lvsr_attdec_plain w = *plain;
    hipStream_t s = (hipStream_t)stream;
    int* ab = (int*)ws;
    u64* planes = (u64*)((char*)ws + 256);
    const size_t bytes = (size_t)a.B * (PB_SMALL + (size_t)g.P * PB_PERWG) * 8;
    (void)hipMemsetAsync(planes, 0, bytes, s);          // the abort word in front of the planes is sticky: cleared by the host only
    const dim3 block(PD_THREADS);
    PbGeom gp = g;
    for (gp.b0 = 0; gp.b0 < a.B; gp.b0 += g.nb) {
    gp.nb = min(g.nb, a.B - gp.b0);
    const dim3 grid(cluster_grid(gp.nb, g.P, 0));
#define PB_LAUNCH(KCV, SHAPE) hipLaunchKernelGGL((attdec_pbwd_kernel<KCV, SHAPE, false>), grid, block, 0, s, gb, w, gp, planes, ab, PbStack())
#define PB_LAUNCH_KC(SHAPE)                       \
    switch (g.KC) {                               \
        case 0: PB_LAUNCH(0, SHAPE); break;       \
        case 4: PB_LAUNCH(4, SHAPE); break;       \
        case 10: PB_LAUNCH(10, SHAPE); break;     \
        default: PB_LAUNCH(16, SHAPE); break;     \
    }
    if (g.shape == 0) { PB_LAUNCH_KC(PdShape8) }
    else if (g.shape == 1) { PB_LAUNCH_KC(PdShape16) }
    else { PB_LAUNCH_KC(PdShape32) }
    }
#undef PB_LAUNCH_KC
#undef PB_LAUNCH
    return lvsr_check_launch("lvsr_attdec_bwd_persistent");
}

extern "C" long long lvsr_attdec_stack2_bwd_persist_ws_bytes(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PbGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pb_geom(a, g, false, true)) return 0;
    return 256 + (long long)a.B * (PB_SMALL + PB_STACK_EXTRA + (long long)g.P * PB_PERWG_STACK) * 8;
}

extern "C" int lvsr_attdec_stack2_bwd_persist_clusters(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PbGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pb_geom(a, g, false, true)) return 0;
    return g.P;
}

extern "C" int lvsr_attdec_bwd_persistent_stack2(void* stream, const lvsr_attdec_bwd_args* args, const lvsr_attdec_plain* plain,
                                                 const lvsr_attdec_stack2* l1, void* ws) {
    LVSR_REQUIRE(args != nullptr && plain != nullptr && l1 != nullptr && ws != nullptr, "lvsr_attdec_bwd_persistent_stack2: null argument");
    AttBwd gb;
    memcpy(&gb, args, sizeof(gb));
    const AttDec& a = gb.f;
    if (int rc = attdec_check(a, "lvsr_attdec_bwd_persistent_stack2")) return rc;
    LVSR_REQUIRE(a.label0 == 0 && (args->parts & 3) % 3 == 0 && a.S_ld >= 2 * a.D && args->ds_ld >= 2 * a.D,
                 "lvsr_attdec_bwd_persistent_stack2: runs all labels; states and state gradients hold both layers side by side (S_ld, ds_ld >= 2 D)");
    PbGeom g;
    LVSR_REQUIRE((plain->AW_ld ? plain->AW_ld : 3 * a.D) % 4 == 0 && (l1->AW1_ld ? l1->AW1_ld : 3 * a.D) % 4 == 0,
                 "lvsr_attdec_bwd_persistent_stack2: the rows of AW / AW1 must be a multiple of 4 floats apart");
    LVSR_REQUIRE(pb_geom(a, g, false, true), "lvsr_attdec_bwd_persistent_stack2: configuration outside the kernel's limits "
                 "(lvsr_attdec_stack2_bwd_persist_ws_bytes returns 0 for it)");
    LVSR_REQUIRE(a.PA_bs == a.M && a.PA_ts == (long long)a.B * a.M, "lvsr_attdec_bwd_persistent_stack2: contexts must be contiguous (Tp,B,*)");
    LVSR_REQUIRE(plain->Ws && plain->Whg && plain->Whh && plain->AW && gb.QR && gb.DXG && gb.DSW && gb.dPA && gb.ds && gb.accWe,
                 "lvsr_attdec_bwd_persistent_stack2: missing buffers of the attention / layer-0 block");
    LVSR_REQUIRE(a.K == 0 || (gb.DCV && gb.accH), "lvsr_attdec_bwd_persistent_stack2: DCV / accH missing");
    LVSR_REQUIRE(l1->Whg1 && l1->Whh1 && l1->Ws1 && l1->F1 && l1->AW1 && l1->U1 && l1->R1 && l1->C1 && l1->DXG1,
                 "lvsr_attdec_bwd_persistent_stack2: layer-1 block incomplete");
    const lvsr_attdec_plain w = *plain;
    const PbStack k2 = *l1;
    hipStream_t s = (hipStream_t)stream;
    int* ab = (int*)ws;
    u64* planes = (u64*)((char*)ws + 256);
    const size_t bytes = (size_t)a.B * (PB_SMALL + PB_STACK_EXTRA + (size_t)g.P * PB_PERWG_STACK) * 8;
    (void)hipMemsetAsync(planes, 0, bytes, s);
    const dim3 grid(cluster_grid(a.B, 2 * g.P, 0)), block(PD_THREADS);
    switch (g.KC) {
        case 0: hipLaunchKernelGGL((attdec_pbwd_kernel<0, PdShape8, true>), grid, block, 0, s, gb, w, g, planes, ab, k2); break;
        case 4: hipLaunchKernelGGL((attdec_pbwd_kernel<4, PdShape8, true>), grid, block, 0, s, gb, w, g, planes, ab, k2); break;
        case 10: hipLaunchKernelGGL((attdec_pbwd_kernel<10, PdShape8, true>), grid, block, 0, s, gb, w, g, planes, ab, k2); break;
        default: hipLaunchKernelGGL((attdec_pbwd_kernel<16, PdShape8, true>), grid, block, 0, s, gb, w, g, planes, ab, k2); break;
    }
    return lvsr_check_launch("lvsr_attdec_bwd_persistent_stack2");
}
