// Persistent attention-decoder forward for gfx950: ONE launch for the whole label loop instead of five launches per label.
//
// The decoder's chain per label is  s -> s W_s -> energies -> softmax -> glimpse -> gates -> r*s -> candidate -> s'; as separate
// launches every all-to-all dependency costs a kernel boundary (~3-5 us, decoder_fwd.hip: 33 us per label on WSJ-base), inside
// one launch it costs a granule hand-off between the CUs of a cluster (~0.85 us, persist.h / encoder_persist.hip).  Utterances
// are independent, so a cluster of P work-groups (512 threads each: two waves per SIMD, one work-group per CU) serves ONE
// utterance and everything the label loop re-reads stays on chip for the whole sequence:
//   * thread (unit jl = tid/16, slice q = tid%16) keeps, in REGISTERS, its 1/16 row slice of the three decoder-GRU state columns
//     of unit j = 32 p + jl (state_to_gates u/r, state_to_state: 3 x 16 values) and of two columns of transform_states (2 x 16);
//     every wave keeps the handler operands of its 64 match columns and the filter taps of its MFMA tap groups;
//   * the glimpse never exists inside the loop: the gate inputs it feeds are linear in it, wa W_d = sum_t alpha_t (A_t W_d), so
//     the caller precomputes AW = attended @ [fork_inputs.W | fork_gate_inputs.W] (one GEMM per batch) and the work-group keeps
//     the rows of ITS 96 gate columns in LDS: the gate inputs become a local contraction with the alignment every work-group
//     already holds (no glimpse exchange, no glimpse weights on chip: the first version kept them, 352 registers per thread of
//     256, and spilled).  The weighted averages themselves (readout, weight gradients) are one batched kernel after the loop
//     (lvsr_attdec_glimpses);
//   * in LDS: the work-group's rows of the preprocessed attended (positions t = p mod P: interleaved, so any window is balanced
//     over the cluster; TRANSPOSED [m][position] for location-aware attention, so that an MFMA accumulator is initialised with
//     one 16-byte read), its AW columns (all positions) and the small vectors;
//   * per label four phase vectors travel as {epoch,value} granules (transformed state M, energies T', r*s D, s' D), everything
//     else is local.  Work that does not sit on the chain runs in the shadow of a hand-off: the location convolution of the NEXT
//     label (needs only the new alignment; on the matrix cores, 16 filters x 16 positions tiles, four taps per MFMA) and the
//     update-gate / candidate-input sums behind the r*s exchange, the saved tensors' plain stores anywhere.
// The energies (T'/P positions x M per work-group and label: K FMAs + tanh each) were the VALU-heavy part (4.1 us per label with
// two columns per thread in v_pk_fma_f32): the handler contraction now runs on the matrix cores and the elementwise part on the
// accumulators (2.4 us), see the comment at the operand registers below.
//
// Same results as decoder_fwd.hip up to float32 rounding (order of additions; the reassociated glimpse); it writes every tensor
// the backward pass reads (sW, CV, EN, ZB, W, U, R, C, RH, S, pos; WA through lvsr_attdec_glimpses).  Limits (else the caller
// uses the step kernels): D <= 512, M <= 512, T' <= 512, M <= MC UNITS P and B P <= the device's CUs for the cluster shape
// (decoder_persist.h PdShape: clusters of 8 / 16 work-groups at D <= 256, up to 32 at D <= 512), conv filter width <= 256, the LDS
// budget below (T' <= ~216 with clusters of 8, ~440 with clusters of 16 at WSJ-base dims).  With the window_around_* priors the window centres of
// all utterances bound the window: one more (B-granule) exchange per label, between all clusters.
#include "decoder_persist.h"
#include <stdlib.h>
#include <string.h>

struct PdGeom {
    int P, nown, nownp, KC, KCP, FW, prof, RL, shape;
    int o_pa, o_a, o_cv, o_al, o_sv, o_rs, o_xw, o_red, o_pos, o_clk, o_cp, o_sw, o_s0, Bp, total;
    int nb, b0;          // utterances of this launch: [b0, b0 + nb) (pd_pick_passes)
};


static int pd_kc(int K) {
    const int inst[4] = {0, 4, 10, 16};
    for (int i = 0; i < 4; ++i)
        if (K <= inst[i]) return inst[i];
    return -1;
}

// stack: the two-layer launch (lvsr_attdec_fwd_persistent_stack2): clusters of 8 (PdShape8), two of them per utterance
static bool pd_geom(const AttDec& a, PdGeom& g, bool allow16 = true, bool stack = false) {
    if ((a.phases & 3) != 3 || a.step_dev != nullptr || a.group_rows != 0) return false;
    if (a.M > PD_MAXV || a.Tp > PD_MAXV) return false;
    g.KC = pd_kc(a.K);
    if (g.KC < 0) return false;
    g.KCP = (g.KC + 3) / 4 * 4;
    PdPick k;
    if (stack) {
        if (a.D > PdShape8::DMAX) return false;
        k.shape = 0; k.UNITS = PdShape8::UNITS; k.KSPLIT = PdShape8::KSPLIT; k.KD = PdShape8::KD; k.MC = PdShape8::MC; k.AWS = PdShape8::AWS;
        k.P = (a.D + k.UNITS - 1) / k.UNITS;
        if (a.M > k.P * k.MC * k.UNITS || a.B * 2 * k.P > lvsr_max_cluster_wgs()) return false;
    } else if (!pd_pick_passes(a.B, a.D, a.M, a.K == 0 || a.prior_type == 0, k, g.nb, allow16)) return false;
    if (stack) g.nb = a.B;
    g.b0 = 0;
    g.P = k.P; g.shape = k.shape;
    g.nown = (a.Tp + g.P - 1) / g.P;
    g.nownp = (g.nown + PD_CH - 1) / PD_CH * PD_CH;
    g.FW = 2 * a.c + 1;
    if (a.K > 0 && g.FW > 4 * 8 * PD_NW) return false;          // tap groups of the convolution: 8 per wave
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) / 4 * 4; return at; };
    g.RL = (g.nown + 3) / 4 * 4;
    if (g.RL % 32 == 0) g.RL += 4;          // transposed table [m][RL] of the MFMA energy path: row stride off the bank period
    g.o_pa = take((g.KC > 0 ? g.RL : g.nown) * a.M + 64);
    g.o_a = take(a.Tp * k.AWS);
    g.o_cv = take(g.nownp * (g.KCP > 0 ? g.KCP : 4));
    take(256);                               // the convolution reads up to 255 floats below `al` (and c + 16 P above) unclamped
    g.o_al = take(a.Tp);
    take(max(0, g.nownp * g.P + a.c + 4 - a.Tp));      // (... and up to nownp P + c above its start)
    g.o_sv = take(k.KSPLIT * (k.KD + 4));
    g.o_rs = take(k.KSPLIT * (k.KD + 4));
    g.o_s0 = stack ? take(k.KSPLIT * (k.KD + 4)) : 0;      // (layer-1 cluster of the two-layer launch: the new state of layer 0)
    g.o_xw = take(PD_NW * PD_CH);
    g.o_sw = take(PD_THREADS);
    g.o_red = take(3 * PD_NW);
    g.Bp = (a.B + 3) / 4 * 4;
    g.o_pos = take(2 * g.Bp);
    g.o_clk = take(2 * (PD_NPROF + 1));
    g.o_cp = take(PD_NW * 16 * 17);
    g.total = o;
    g.prof = lvsr_knob(LVSR_KNOB_PHASE_CLOCK);         // phase clock of work-group 0 (tools/probe_decoder_persist.py); costs ~1 us per label
    if (o > PD_LDS_FLOATS && k.shape == 1 && allow16 && !stack) return pd_geom(a, g, false);      // clusters of 8 instead, if they fit
    return o <= PD_LDS_FLOATS;
}

typedef lvsr_attdec_stack2 PdStack;

// The layer-1 cluster of the two-layer launch (see lvsr_attdec_fwd_persistent_stack2 in lvsr_hip.h).  Work-group p of it owns units
// [UNITS p, UNITS p + UNITS) of layer 1; thread (unit, q) keeps its row slice of the layer's three recurrent columns, of the three
// fork_1 columns (input: the NEW state of layer 0) and of MC transform_states#1 columns.  Per label: own-state gate sums -> the
// label's energies (plane EN of the main cluster) -> softmax (repeated here: cheaper than a hop) -> glimpse part of the gate inputs
// from the AW1 slice in LDS -> new state of layer 0 (plane S of the main cluster) -> fork_1 sums -> r*s exchange -> candidate ->
// new state (plane S1) -> its transformed part for the NEXT label's energies (plane SW1, which the main cluster adds to its own).
template <class SH>
__device__ __forceinline__ void pd_stack_layer1(const AttDec& a, const PdGeom& g, const PdStack& k2, float* lds, u64* planes, int* abort_word,
                                                int b, int p) {
    constexpr int PD_UNITS = SH::UNITS, PD_KSPLIT = SH::KSPLIT, PD_KD = SH::KD, PD_MC = SH::MC, PD_AWS = SH::AWS;
    float* const sv0 = lds + g.o_s0;      // new state of layer 0, sliced
    float* const AWs = lds + g.o_a;       // [T'][PD_AWS] own gate columns of AW1
    float* const al = lds + g.o_al;
    float* const sv = lds + g.o_sv;       // own state (layer 1), sliced
    float* const rsv = lds + g.o_rs;
    float* const red = lds + g.o_red;
    float* const posv = lds + g.o_pos;
    const int P = g.P;
    const int tid = threadIdx.x, q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = a.D, M = a.M, Tp = a.Tp, B = a.B, K = a.K, L = a.L;
    const int SLD = a.S_ld ? a.S_ld : 2 * D, FLD = k2.F1_ld ? k2.F1_ld : 3 * D;
    const int j = p * PD_UNITS + jl;
    const bool junit = j < D;
    f32x2 wsw[PD_MC][PD_KD / 2], whu[PD_KD / 2], whr[PD_KD / 2], whc[PD_KD / 2], wfx[PD_KD / 2], wfu[PD_KD / 2], wfr[PD_KD / 2];
    {
        const size_t jc = (size_t)min(j, D - 1);
#pragma unroll
        for (int x = 0; x < PD_KD / 2; ++x) {
            float v[6][2], s2[PD_MC][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = q * PD_KD + 2 * x + e;
                const size_t kc = (size_t)min(k, D - 1);
                const float keep = (junit && k < D) ? 1.f : 0.f;
                v[0][e] = k2.Whg1[kc * 2 * D + jc] * keep;
                v[1][e] = k2.Whg1[kc * 2 * D + D + jc] * keep;
                v[2][e] = k2.Whh1[kc * D + jc] * keep;
                v[3][e] = k2.F1[kc * FLD + jc] * keep;
                v[4][e] = k2.F1[kc * FLD + D + jc] * keep;
                v[5][e] = k2.F1[kc * FLD + 2 * D + jc] * keep;
#pragma unroll
                for (int c = 0; c < PD_MC; ++c) {
                    const int m = (p * PD_MC + c) * PD_UNITS + jl;
                    s2[c][e] = k2.Ws1[kc * M + (size_t)min(m, M - 1)] * ((m < M && k < D) ? 1.f : 0.f);
                }
            }
            whu[x] = (f32x2){v[0][0], v[0][1]}; whr[x] = (f32x2){v[1][0], v[1][1]}; whc[x] = (f32x2){v[2][0], v[2][1]};
            wfx[x] = (f32x2){v[3][0], v[3][1]}; wfu[x] = (f32x2){v[4][0], v[4][1]}; wfr[x] = (f32x2){v[5][0], v[5][1]};
#pragma unroll
            for (int c = 0; c < PD_MC; ++c) wsw[c][x] = (f32x2){s2[c][0], s2[c][1]};
        }
    }
    const float am = tid < Tp ? a.Am[(size_t)tid * a.Am_ts + (size_t)b * a.Am_bs] : 0.f;
    const float eb = a.e_bias ? a.e_bias[0] : 0.f;
    for (int x = tid; x < g.total - g.o_cv; x += PD_THREADS) lds[g.o_cv + x] = 0.f;
    const int AWL = k2.AW1_ld ? k2.AW1_ld : 3 * D;
    for (int x = tid; x < Tp * 3 * PD_UNITS; x += PD_THREADS) {
        const int t = x / (3 * PD_UNITS), gcol = x % (3 * PD_UNITS), gate = gcol / PD_UNITS, unit = p * PD_UNITS + gcol % PD_UNITS;
        AWs[t * PD_AWS + gcol] = unit < D ? k2.AW1[((size_t)t * B + b) * AWL + (size_t)gate * D + unit] : 0.f;
    }
    __syncthreads();
    for (int t = tid; t < Tp; t += PD_THREADS) al[t] = a.W[(size_t)b * Tp + t];
    for (int k = tid; k < D; k += PD_THREADS) sv[pd_slot(k, PD_KD)] = a.S[(size_t)b * SLD + D + k];
    float sj = junit ? a.S[(size_t)b * SLD + D + j] : 0.f;
    u64* const base = planes + (size_t)b * PD_NPLANE_STACK * PD_MAXV;
    u64* const gEN = base + PD_MAXV;
    u64* const gS0 = base + 3 * PD_MAXV;
    u64* const gSW1 = base + 4 * PD_MAXV;
    u64* const gRS1 = base + 5 * PD_MAXV;
    u64* const gS1 = base + 6 * PD_MAXV;
    u64* const gPOS = planes + (size_t)B * PD_NPLANE_STACK * PD_MAXV;
    const bool plain = cluster_shares_xcd(gPOS + 2 * g.Bp + (size_t)b * PD_MAXP, 2 * P, P + p, abort_word);
    const bool winprior = K > 0 && a.prior_type != 0;
    __syncthreads();
    bool pos_given = false;
    if (winprior && (a.phases & 4)) {
        for (int x = tid; x < B; x += PD_THREADS) posv[x] = a.pos[x];
        pos_given = true;
        __syncthreads();
    }
    // transformed part of the initial state: what the main cluster's first energies wait for
    {
        float sw[PD_MC];
#pragma unroll
        for (int c = 0; c < PD_MC; ++c) sw[c] = pd_dot<PD_KD, PD_KSPLIT>(wsw[c], sv, q);
        if (q < PD_MC) {
            const int m = (p * PD_MC + q) * PD_UNITS + jl;
            float mine = sw[0];
#pragma unroll
            for (int c = 1; c < PD_MC; ++c) mine = q == c ? sw[c] : mine;
            if (m < M) granule_store(gSW1 + m, 1u, mine, plain);
        }
    }
    for (int i = 0; i < L; ++i) {
        const unsigned epoch = (unsigned)(i + 1);
        const size_t row = (size_t)i * B + b;
        const int tid = lvsr_unhoisted((int)threadIdx.x), q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63;
        const int j = p * PD_UNITS + jl;
        const bool junit = j < D;
        const float* xr = k2.xg1 + row * 3 * D;
        const float fx = junit ? xr[j] : 0.f, fu = junit ? xr[D + j] : 0.f, fr = junit ? xr[2 * D + j] : 0.f;
        const float ym = a.ymask ? a.ymask[row] : 1.f;
        const float gu = pd_dot<PD_KD, PD_KSPLIT>(whu, sv, q), gr = pd_dot<PD_KD, PD_KSPLIT>(whr, sv, q);
        if (winprior && !(i == 0 && pos_given)) {
            float v[PD_NV];
            if (!pd_gather(gPOS + (i & 1) * g.Bp, B, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < B) posv[(i & 1) * g.Bp + tid + x * PD_THREADS] = v[x];
            __syncthreads();
        }
        const Win wi = winprior ? pd_window(a, i, posv + (i & 1) * g.Bp) : attdec_window(a, i);
        float amk = am;
        if (winprior) {
            const float pb = posv[(i & 1) * g.Bp + b];
            const float lo = floorf(pb - (float)a.p0), hi = ceilf(pb + (float)a.p1), tf = (float)tid;
            amk *= (tf > lo && tf < hi) ? 1.f : 0.f;
        }
        // ---- the label's alignment: the main cluster's energies, normalised here as there (decoder_persist.hip phase C)
        float eg[PD_NV];
        if (!pd_gather(gEN, Tp, epoch, abort_word, eg)) return;
        {
            const int t = tid;
            const bool inw = t < Tp && t >= wi.begin && t < wi.end;
            const float e = inw ? eg[0] + eb : 0.f;
            const float wmx = wave_max_dpp(inw ? e : -3.0e38f);
            const float wany = wave_max_dpp((inw && 1.f - amk == 0.f) ? 1.f : 0.f);
            if (lane == 0) { red[wave] = wmx; red[PD_NW + wave] = wany; }
            __syncthreads();
            float mx = red[0], anyone = red[PD_NW];
#pragma unroll
            for (int x = 1; x < PD_NW; ++x) { mx = fmaxf(mx, red[x]); anyone = fmaxf(anyone, red[PD_NW + x]); }
            float u = 0.f;
            if (inw) {
                if (a.normalizer == 0) u = __expf(e - mx) * amk;
                else if (a.normalizer == 1) u = sigmoidf_(e) * amk;
                else u = fmaxf(e / 1000.f, 0.f) * amk;
            }
            const float wsm = wave_sum_dpp(u);
            if (lane == 0) red[2 * PD_NW + wave] = wsm;
            __syncthreads();
            float ssum = 0.f;
#pragma unroll
            for (int x = 0; x < PD_NW; ++x) ssum += red[2 * PD_NW + x];
            const float Z = ssum + (anyone > 0.f ? 0.f : 1.f);
            if (t < Tp) al[t] = inw ? u / Z : 0.f;
        }
        __syncthreads();
        // ---- glimpse part of the gate inputs: sum_t alpha_t AW1[t] over the window (as phase D of the main kernel)
        float gx, gu2, gr2;
        {
            float x0 = 0.f, x1 = 0.f, u0 = 0.f, u1 = 0.f, r0 = 0.f, r1 = 0.f;
            for (int t0 = wi.begin + q; t0 < wi.end; t0 += 4 * PD_KSPLIT) {
                float av[4], vx[4], vu[4], vr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tt = t0 + e * PD_KSPLIT, tc = min(tt, Tp - 1);
                    const float araw = al[tc];
                    av[e] = tt < wi.end ? araw : 0.f;
                    const float* rowp = AWs + tc * PD_AWS + jl;
                    vx[e] = rowp[0]; vu[e] = rowp[PD_UNITS]; vr[e] = rowp[2 * PD_UNITS];
                }
                x0 += av[0] * vx[0]; u0 += av[0] * vu[0]; r0 += av[0] * vr[0];
                x1 += av[1] * vx[1]; u1 += av[1] * vu[1]; r1 += av[1] * vr[1];
                x0 += av[2] * vx[2]; u0 += av[2] * vu[2]; r0 += av[2] * vr[2];
                x1 += av[3] * vx[3]; u1 += av[3] * vu[3]; r1 += av[3] * vr[3];
            }
            gr2 = group_sum<PD_KSPLIT>(r0 + r1);
            gx = group_sum<PD_KSPLIT>(x0 + x1); gu2 = group_sum<PD_KSPLIT>(u0 + u1);
        }
        // ---- fork_1 of the new state of layer 0
        {
            float v[PD_NV];
            if (!pd_gather(gS0, D, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < D) sv0[pd_slot(tid + x * PD_THREADS, PD_KD)] = v[x];
        }
        __syncthreads();
        const float ar = pd_dot<PD_KD, PD_KSPLIT>(wfr, sv0, q);
        const float rr = sigmoid_fast(fr + gr + gr2 + ar);
        const float rs = junit ? rr * sj : 0.f;
        if (q == 0 && junit) {
            granule_store(gRS1 + j, epoch, rs, plain);
            k2.R1[row * D + j] = rr;
            k2.RH1[row * D + j] = rs;
        }
        const float uu = sigmoid_fast(fu + gu + gu2 + pd_dot<PD_KD, PD_KSPLIT>(wfu, sv0, q));
        const float xin = fx + gx + pd_dot<PD_KD, PD_KSPLIT>(wfx, sv0, q);
        if (q == 0 && junit) k2.U1[row * D + j] = uu;
        {
            float v[PD_NV];
            if (!pd_gather(gRS1, D, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < D) rsv[pd_slot(tid + x * PD_THREADS, PD_KD)] = v[x];
        }
        __syncthreads();
        const float cand = tanh_fast(xin + pd_dot<PD_KD, PD_KSPLIT>(whc, rsv, q));
        float sn = cand * uu + sj * (1.f - uu);
        sn = ym * sn + (1.f - ym) * sj;
        if (!junit) sn = 0.f;
        if (q == 0 && junit) {
            if (i + 1 < L) granule_store(gS1 + j, epoch, sn, plain);
            k2.C1[row * D + j] = cand;
            a.S[((size_t)(i + 1) * B + b) * SLD + D + j] = sn;
        }
        sj = sn;
        if (i + 1 < L) {
            // the whole new state of this layer: operand of the next label's gate sums, and of the transformed part the main
            // cluster's next energies wait for
            float v[PD_NV];
            if (!pd_gather(gS1, D, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < D) sv[pd_slot(tid + x * PD_THREADS, PD_KD)] = v[x];
            __syncthreads();
            float sw[PD_MC];
#pragma unroll
            for (int c = 0; c < PD_MC; ++c) sw[c] = pd_dot<PD_KD, PD_KSPLIT>(wsw[c], sv, q);
            if (q < PD_MC) {
                const int m = (p * PD_MC + q) * PD_UNITS + jl;
                float mine = sw[0];
#pragma unroll
                for (int c = 1; c < PD_MC; ++c) mine = q == c ? sw[c] : mine;
                if (m < M) granule_store(gSW1 + m, epoch + 1u, mine, plain);
            }
        }
    }
}

template <int KC, class SH, bool STACK = false>
__global__ __launch_bounds__(PD_THREADS) void attdec_pfwd_kernel(AttDec a, lvsr_attdec_plain w, PdGeom g, u64* planes, int* abort_word, PdStack k2) {
    constexpr int KCP = (KC + 3) / 4 * 4;
    constexpr int PD_UNITS = SH::UNITS, PD_KSPLIT = SH::KSPLIT, PD_KD = SH::KD, PD_MC = SH::MC, PD_AWS = SH::AWS;
    __shared__ __attribute__((aligned(16))) float lds[PD_LDS_FLOATS];
    float* const PAs = lds + g.o_pa;      // [nown][M]   own positions (t = tl*P + p) of the preprocessed attended
    float* const AWs = lds + g.o_a;       // [T'][PD_AWS] own gate columns of AW: [x | u | r][32 units]
    float* const cvs = lds + g.o_cv;      // [nownp][KCP] convolution features of the own positions
    float* const al = lds + g.o_al;       // [T']        current alignment
    float* const sv = lds + g.o_sv;       // state, sliced by PD_KD
    float* const rsv = lds + g.o_rs;      // r*s, sliced by PD_KD
    float* const xw = lds + g.o_xw;       // [PD_NW][16] energy partials (one per wave and position of a round)
    float* const swst = lds + g.o_sw;     // [512] transformed state, staged between the sweep's and the energy phase's thread layout
    float* const red = lds + g.o_red;
    float* const posv = lds + g.o_pos;    // [2][Bp] window centres of all utterances, by label parity
    float* const cp = lds + g.o_cp;       // [PD_NW][16][17] convolution partial tiles
    const int P = g.P, nown = g.nown;
    int b, p;
    if (!cluster_of_block(STACK ? 2 * P : P, g.nb, 0, b, p)) return;  // (work-groups of the grid's padding)
    b += g.b0;
    if (STACK && p >= P) {            // the second cluster of the utterance: layer 1 of the stack
        pd_stack_layer1<SH>(a, g, k2, lds, planes, abort_word, b, p - P);
        return;
    }
    const int tid = threadIdx.x, q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: conditions on it become scalar branches
    const int D = a.D, M = a.M, Tp = a.Tp, B = a.B, K = a.K, L = a.L;
    const int SLD = a.S_ld ? a.S_ld : D;          // row stride of the state slots (two-layer launch: 2 D, this layer in front)
    const int j = p * PD_UNITS + jl;
    const bool junit = j < D;
    // ---- weights of this thread, in registers for the whole sequence (clamped addresses, zeroed afterwards: straight-line loads)
    f32x2 wsw[PD_MC][PD_KD / 2], whu[PD_KD / 2], whr[PD_KD / 2], whc[PD_KD / 2];
    {
        const size_t jc = (size_t)min(j, D - 1);
#pragma unroll
        for (int x = 0; x < PD_KD / 2; ++x) {
            float v[3][2], s2[PD_MC][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = q * PD_KD + 2 * x + e;
                const size_t kc = (size_t)min(k, D - 1);
                const float keep = (junit && k < D) ? 1.f : 0.f;
                v[0][e] = w.Whg[kc * 2 * D + jc] * keep;
                v[1][e] = w.Whg[kc * 2 * D + D + jc] * keep;
                v[2][e] = w.Whh[kc * D + jc] * keep;
#pragma unroll
                for (int c = 0; c < PD_MC; ++c) {
                    const int m = (p * PD_MC + c) * PD_UNITS + jl;
                    s2[c][e] = w.Ws[kc * M + (size_t)min(m, M - 1)] * ((m < M && k < D) ? 1.f : 0.f);
                }
            }
            whu[x] = (f32x2){v[0][0], v[0][1]}; whr[x] = (f32x2){v[1][0], v[1][1]}; whc[x] = (f32x2){v[2][0], v[2][1]};
#pragma unroll
            for (int c = 0; c < PD_MC; ++c) wsw[c][x] = (f32x2){s2[c][0], s2[c][1]};
        }
    }
    // energy phase.  tanh(x) = 1 - 2 / (1 + 2^(c x)), c = 2 log2(e): the operands are pre-scaled by c (preprocessed attended and
    // handler once, the transformed state per label), and w_e . tanh = sum(w_e) - 2 w_e . (1 / (1 + 2^y)) with the -2 folded in.
    //  * location-aware attention (KC > 0): on the matrix cores.  Per round of 16 own positions wave w owns match columns
    //    [64 w, 64 w + 64) as four 16 x 16 tiles: C = PA + sW (one 16-byte LDS read per tile from the TRANSPOSED table
    //    PAs[m][RL]), A = convolution features (16 positions x 4 filters), B = handler (resident: Hb), three MFMAs per tile; the
    //    accumulator lane (c16 = lane % 16, g4 = lane / 16) holds positions 4 g4 + r of column c16: exp2 / rcp / w_e on them, a
    //    DPP fold over the 16 column lanes, one partial per (wave, position) through LDS;
    //  * content-only attention (KC = 0): thread (pair mp, half) holds columns mp, mp + 256 on the VALU as before.
    const float C2 = 2.885390081777927f;
    const int mp = tid & (PD_MP - 1), ehalf = wave / (PD_MP / 64);
    const int c16 = lane & 15, g4 = lane >> 4;
    float Hb[KC > 0 ? 4 : 1][KC > 0 ? KCP / 4 : 1], wet[4];          // MFMA path: B operands and -2 w_e of this lane's four columns
    f32x2 wem2 = {0.f, 0.f};
    float wsum = 0.f, am;
    if (KC > 0) {
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int m = (4 * wave + tile) * 16 + c16;
#pragma unroll
            for (int sq = 0; sq < KCP / 4; ++sq) {
                const int k = 4 * sq + g4;
                Hb[tile][sq] = (k < K && m < M) ? C2 * a.handler[(size_t)k * M + m] : 0.f;
            }
            const float wv = m < M ? a.w_e[m] : 0.f;
            wet[tile] = -2.f * wv;
            wsum += wv;
        }
    } else {
        float wv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) wv[x] = (mp + x * PD_MP) < M ? a.w_e[mp + x * PD_MP] : 0.f;
        wem2 = (f32x2){-2.f * wv[0], -2.f * wv[1]};
        wsum = wv[0] + wv[1];
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) wet[tile] = 0.f;
    }
    am = tid < Tp ? a.Am[(size_t)tid * a.Am_ts + (size_t)b * a.Am_bs] : 0.f;
    const float eb = a.e_bias ? a.e_bias[0] : 0.f;
    // ---- LDS residents
    for (int x = tid; x < g.total - g.o_cv; x += PD_THREADS) lds[g.o_cv + x] = 0.f;      // everything behind the big tables
    if (KC > 0) {              // transposed: PAs[m * RL + tl]
        for (int x = tid; x < g.RL * M + 64; x += PD_THREADS) PAs[x] = 0.f;
        __syncthreads();
        for (int x = tid; x < nown * M; x += PD_THREADS) {
            const int tl = x / M, m = x % M, t = tl * P + p;
            if (t < Tp) PAs[m * g.RL + tl] = C2 * a.PA[(size_t)t * a.PA_ts + (size_t)b * a.PA_bs + m];
        }
    } else {
        for (int x = tid; x < nown * M; x += PD_THREADS) {
            const int tl = x / M, m = x % M, t = tl * P + p;
            PAs[x] = t < Tp ? C2 * a.PA[(size_t)t * a.PA_ts + (size_t)b * a.PA_bs + m] : 0.f;
        }
    }
    for (int x = tid; x < Tp * 3 * PD_UNITS; x += PD_THREADS) {
        const int t = x / (3 * PD_UNITS), gcol = x % (3 * PD_UNITS), gate = gcol / PD_UNITS, unit = p * PD_UNITS + gcol % PD_UNITS;
        AWs[t * PD_AWS + gcol] = unit < D ? w.AW[((size_t)t * B + b) * (w.AW_ld ? w.AW_ld : 3 * D) + (size_t)gate * D + unit] : 0.f;
    }
    __syncthreads();
    for (int t = tid; t < Tp; t += PD_THREADS) al[t] = a.W[(size_t)b * Tp + t];
    for (int k = tid; k < D; k += PD_THREADS) sv[pd_slot(k, PD_KD)] = a.S[(size_t)b * SLD + k];
    float sj = junit ? a.S[(size_t)b * SLD + j] : 0.f;
    constexpr int NPL = STACK ? PD_NPLANE_STACK : PD_NPLANE;
    u64* const gSW = planes + (size_t)b * NPL * PD_MAXV;
    u64* const gEN = gSW + PD_MAXV;
    u64* const gRS = gSW + 2 * PD_MAXV;
    u64* const gS = gSW + 3 * PD_MAXV;
    u64* const gSW1 = gSW + 4 * PD_MAXV;                             // (two-layer launch) layer 1's part of the transformed state
    u64* const gPOS = planes + (size_t)B * NPL * PD_MAXV;            // [2][Bp], shared by all clusters
    // plain stores for the exchanges INSIDE the cluster when its work-groups share an XCD (persist.h); the window centres travel
    // between clusters and stay write-through.  The XCC_ID granules: PD_MAXP per utterance behind the window centres.
    const bool plain = cluster_shares_xcd(gPOS + 2 * g.Bp + (size_t)b * PD_MAXP, STACK ? 2 * P : P, p, abort_word);
    const bool winprior = K > 0 && a.prior_type != 0;
    const bool pos_wave = p == P - 1 && wave == PD_NW - 1;           // the wave that derives this utterance's window centre
    __syncthreads();
    bool pos_given = false;                  // window centres of slot 0 supplied by the caller (phases bit 2)
    if (winprior) {
        if (a.phases & 4) {
            for (int x = tid; x < B; x += PD_THREADS) posv[x] = a.pos[x];
            pos_given = true;
            __syncthreads();
        } else if (pos_wave) {               // derived from the initial alignment; gathered by everybody inside label 0
            const float r = attdec_pos_of_row_wave(a, al);
            if (lane == 0) {
                granule_store(gPOS + b, 1u, r);
                a.pos[b] = r;
            }
        }
    }

    PdClock clk;
    clk.start(g.prof != 0 && blockIdx.x == 0 && tid == 0, lds + g.o_clk);

    // Location convolution of the alignment in `al` for the own positions with the window of label i (lvsr/expressions.py:28-54:
    // true convolution of the cut alignment): cv[k][t] = sum_d f[k][c+d] * al[t-d].  On the matrix cores: per block of 16 own
    // positions a 16(filters) x 16(positions) tile accumulates over the 2c+1 taps four at a time (v_mfma_f32_16x16x4_f32: A =
    // filter taps, B = shifted, window-masked alignment values); the four waves split the tap groups and fold through LDS.
    // A operands (filter taps) of this lane's MFMAs: wave w takes the tap groups w, w + 8, ...; the same for every label and block
    float cfa[8];
    {
        const int r16 = lane & 15, kk = lane >> 4;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int u = 4 * (wave + n * PD_NW) + kk;
            cfa[n] = (KC > 0 && r16 < K && u < g.FW) ? a.filters[(size_t)min(r16, K - 1) * g.FW + min(u, g.FW - 1)] : 0.f;
        }
    }
    auto conv = [&](int i, const Win wi) {
        const int cn = a.c;
        const int r16 = lane & 15, kk = lane >> 4;
        for (int bl = 0; bl * 16 < nown; ++bl) {
            const int tlx = bl * 16 + r16, tx = tlx * P + p;          // B operand: this lane's position column
            const bool colok = tlx < nown && tx < Tp;
            // alignment index of tap group n: idx0 - 32 n; in the window <=> (unsigned)(idx - begin) < width.  The reads are
            // not clamped: +-(c + 16 P) around `al` stays inside this work-group's LDS block (pd_geom), the select drops them
            const int idx0 = tx + cn - kk - 4 * wave;
            const unsigned width = colok ? (unsigned)(wi.end - wi.begin) : 0u;
            const float* ap = al + idx0;
            float fb[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const float v = ap[-32 * n];
                fb[n] = ((unsigned)(idx0 - 32 * n - wi.begin) < width) ? v : 0.f;
            }
            f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
#pragma unroll
            for (int n = 0; n < 8; n += 2) {               // tap groups beyond the filter carry zero A operands: no branches
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cfa[n], fb[n], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cfa[n + 1], fb[n + 1], acc1, 0, 0, 0);
            }
            clk.mark(12);
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(wave * 16 + kk * 4 + r) * 17 + r16] = acc0[r] + acc1[r];
            __syncthreads();
            clk.mark(13);
            if (tid < 256) {
                const int k = tid >> 4, x = tid & 15, tl = bl * 16 + x, t = tl * P + p;
                float pv[PD_NW];
#pragma unroll
                for (int wv = 0; wv < PD_NW; ++wv) pv[wv] = cp[(wv * 16 + k) * 17 + x];          // all loads in flight
                const float sum = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
                if (k < K && tl < nown && t < Tp) {
                    const float val = (t >= wi.begin && t < wi.end) ? sum : 0.f;
                    cvs[tl * KCP + k] = val;
                    a.CV[(((size_t)i * B + b) * K + k) * Tp + t] = val;
                }
            }
            if ((bl + 1) * 16 < nown) __syncthreads();
            clk.mark(14);
        }
    };

    Win wnext = attdec_window(a, 0);          // not used with the window_around_* priors
    if (KC > 0 && !winprior) conv(0, wnext);
    for (int i = 0; i < L; ++i) {
        const unsigned epoch = (unsigned)(i + 1);
        const size_t row = (size_t)i * B + b;
        // the per-thread indices are re-derived per label from an opaque copy of the thread id (common.h lvsr_unhoisted): derived
        // from the loop-invariant one, every address / predicate built on them is hoisted out of the label loop and kept (or
        // spilled: 16 registers, 68 bytes of scratch per lane) across it
        const int tid = lvsr_unhoisted((int)threadIdx.x), q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63;
        const int j = p * PD_UNITS + jl;
        const bool junit = j < D;
        const int mp = tid & (PD_MP - 1), c16 = lane & 15, g4 = lane >> 4;
        // per-label operands that do not depend on the recurrence
        const float* xr = a.xg + row * 3 * D;
        const float fx = junit ? xr[j] : 0.f, fu = junit ? xr[D + j] : 0.f, fr = junit ? xr[2 * D + j] : 0.f;
        const float ym = a.ymask ? a.ymask[row] : 1.f;
        // ---- phase A: transformed state (published), state part of the gate pre-activations (kept)
        if (i > 0) {
            float v[PD_NV];
            if (!pd_gather(gS, D, (unsigned)i, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < D) sv[pd_slot(tid + x * PD_THREADS, PD_KD)] = v[x];
        }
        __syncthreads();
        clk.mark(0);
        {
            float sw[PD_MC];
#pragma unroll
            for (int c = 0; c < PD_MC; ++c) sw[c] = pd_dot<PD_KD, PD_KSPLIT>(wsw[c], sv, q);
            if (q < PD_MC) {
                const int m = (p * PD_MC + q) * PD_UNITS + jl;
                float mine = sw[0];
#pragma unroll
                for (int c = 1; c < PD_MC; ++c) mine = q == c ? sw[c] : mine;
                if (m < M) {
                    granule_store(gSW + m, epoch, mine, plain);
                    if (!STACK) a.sW[row * M + m] = mine;
                }
            }
        }
        const float gu = pd_dot<PD_KD, PD_KSPLIT>(whu, sv, q), gr = pd_dot<PD_KD, PD_KSPLIT>(whr, sv, q);
        clk.mark(1);
        // in the shadow of the transformed-state exchange: this label's window centres (published by every cluster during its
        // previous label) and the location convolution of the previous alignment
        if (winprior && !(i == 0 && pos_given)) {
            float v[PD_NV];
            if (!pd_gather(gPOS + (i & 1) * g.Bp, B, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < B) posv[(i & 1) * g.Bp + tid + x * PD_THREADS] = v[x];
            __syncthreads();
        }
        // window of this label: known without the centres for the expanding prior (computed, with the convolution, in the shadow
        // of the previous label's r*s exchange: wnext)
        const Win wi = winprior ? pd_window(a, i, posv + (i & 1) * g.Bp) : wnext;
        float amk[PD_NV];                    // attended mask x window-around mask of this utterance (attdec_mask)
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) amk[x] = am;
        if (winprior) {
            const float pb = posv[(i & 1) * g.Bp + b];
            const float lo = floorf(pb - (float)a.p0), hi = ceilf(pb + (float)a.p1);
#pragma unroll
            for (int x = 0; x < PD_NV; ++x) {
                const float tf = (float)(tid + x * PD_THREADS);
                amk[x] *= (tf > lo && tf < hi) ? 1.f : 0.f;
            }
        }
        clk.mark(9);
        if (KC > 0 && winprior) conv(i, wi);
        clk.mark(7);
        float swv[PD_NV];
        if (!pd_gather(gSW, M, epoch, abort_word, swv)) return;
        // thread tid received granule tid; the energy phase wants columns mp and mp + 256 in one thread: through LDS.  The
        // barrier also publishes the convolution features
        if (STACK) {          // + the layer-1 cluster's part (from that layer's state before this label); saved: the total
            float sw1[PD_NV];
            if (!pd_gather(gSW1, M, epoch, abort_word, sw1)) return;
            swv[0] += sw1[0];
            if (p == 0 && tid < M) a.sW[row * M + tid] = swv[0];
        }
        swst[tid] = swv[0];
        __syncthreads();
        clk.mark(2);
        // ---- phase B: energies of the own positions, 16 per round
        if (KC > 0) {
            float swc[4];                      // transformed state of this lane's four columns, pre-scaled
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) swc[tile] = C2 * swst[(4 * wave + tile) * 16 + c16];
            for (int tl0 = 0; tl0 < nown; tl0 += PD_CH) {
                float av[KCP / 4 > 0 ? KCP / 4 : 1];
                const int tla = min(tl0 + c16, nown - 1);           // A operand: convolution features of position row c16
#pragma unroll
                for (int sq = 0; sq < KCP / 4; ++sq) av[sq] = cvs[tla * KCP + 4 * sq + g4];
                f32x4 acc[4];
#pragma unroll
                for (int tile = 0; tile < 4; ++tile) {
                    const int m = min((4 * wave + tile) * 16 + c16, M - 1);
                    const float4 pa = *(const float4*)(PAs + m * g.RL + tl0 + 4 * g4);      // positions tl0 + 4 g4 .. + 3
                    acc[tile] = (f32x4){pa.x + swc[tile], pa.y + swc[tile], pa.z + swc[tile], pa.w + swc[tile]};
                }
#pragma unroll
                for (int sq = 0; sq < KCP / 4; ++sq)
#pragma unroll
                    for (int tile = 0; tile < 4; ++tile)
                        acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sq], Hb[tile][sq], acc[tile], 0, 0, 0);
                float ra[4] = {wsum, wsum, wsum, wsum};
#pragma unroll
                for (int tile = 0; tile < 4; ++tile) {
                    float ex[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ex[r] = __builtin_amdgcn_exp2f(acc[tile][r]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ex[r] = __builtin_amdgcn_rcpf(1.0f + ex[r]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) ra[r] += wet[tile] * ex[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                      // fold the 16 column lanes of the row group
                    ra[r] += lvsr_dpp_quad_xor1(ra[r]);
                    ra[r] += lvsr_dpp_quad_xor2(ra[r]);
                    ra[r] += lvsr_dpp_half_mirror(ra[r]);
                    ra[r] += lvsr_dpp_mirror(ra[r]);
                }
                if (c16 == 0) *(float4*)(xw + wave * PD_CH + 4 * g4) = make_float4(ra[0], ra[1], ra[2], ra[3]);
                __syncthreads();
                if (tid < PD_CH) {
                    const int tl = tl0 + tid, t = tl * P + p;
                    if (tl < nown && t < Tp) {
                        float e = 0.f;
#pragma unroll
                        for (int wv = 0; wv < PD_NW; ++wv) e += xw[wv * PD_CH + tid];
                        granule_store(gEN + t, epoch, (t >= wi.begin && t < wi.end) ? e : 0.f, plain);
                    }
                }
                if (tl0 + PD_CH < nown) __syncthreads();
            }
        } else {
        // content-only attention: alternate positions per half of the work-group, PD_EG at a time, two columns per thread
        const int m0c = min(mp, M - 1), m1c = min(mp + PD_MP, M - 1);              // columns beyond M carry w_e = 0
        const f32x2 swp = {C2 * swst[mp], C2 * swst[mp + PD_MP]};                   // pre-scaled, see C2
        for (int tl0 = 0; tl0 < nown; tl0 += PD_CH) {
            float v[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = 0.f;
#pragma unroll
            for (int x0 = 0; x0 < 8; x0 += PD_EG) {
                const int tlb = tl0 + 2 * x0 + ehalf;          // the halves take alternate positions: balanced for any T'
                if (tlb < nown) {                              // wave-uniform: padded groups of the last round are skipped
                    float ex[PD_EG][2];
#pragma unroll
                    for (int e = 0; e < PD_EG; ++e) {
                        const int tlc = min(tlb + 2 * e, nown - 1);
                        ex[e][0] = __builtin_amdgcn_exp2f(swp.x + PAs[tlc * M + m0c]);
                        ex[e][1] = __builtin_amdgcn_exp2f(swp.y + PAs[tlc * M + m1c]);
                    }
#pragma unroll
                    for (int e = 0; e < PD_EG; ++e) {
                        ex[e][0] = __builtin_amdgcn_rcpf(1.0f + ex[e][0]);
                        ex[e][1] = __builtin_amdgcn_rcpf(1.0f + ex[e][1]);
                    }
#pragma unroll
                    for (int e = 0; e < PD_EG; ++e) {
                        const int tl = tlb + 2 * e, t = tl * P + p;
                        const float val = wem2.y * ex[e][1] + (wem2.x * ex[e][0] + wsum);
                        v[x0 + e] = (tl < nown && t >= wi.begin && t < wi.end) ? val : 0.f;
                    }
                }
            }
            const float tot = pd_butterfly8(v);
            if ((lane & 7) == 0) xw[wave * 8 + pd_butterfly_index(lane)] = tot;
            __syncthreads();
            if (tid < PD_CH) {
                const int tl = tl0 + tid, t = tl * P + p;
                if (tl < nown && t < Tp) {
                    const float* xr = xw + (tid & 1) * 4 * 8 + (tid >> 1);       // the four waves of this position's half
                    const float e = (xr[0] + xr[8]) + (xr[16] + xr[24]);
                    granule_store(gEN + t, epoch, (t >= wi.begin && t < wi.end) ? e : 0.f, plain);
                }
            }
            if (tl0 + PD_CH < nown) __syncthreads();
        }
        }
        clk.mark(3);
        float eg[PD_NV];
        if (!pd_gather(gEN, Tp, epoch, abort_word, eg)) return;
        clk.mark(4);
        // ---- phase C: normalisation over the window (every work-group, redundantly; one position per thread, three barriers)
        {
            const int t = tid;
            const bool inw = t < Tp && t >= wi.begin && t < wi.end;
            const float e = inw ? eg[0] + eb : 0.f;
            if (p == 0 && t < Tp) a.EN[row * Tp + t] = e;                  // pasted into zeros
            const float wmx = wave_max_dpp(inw ? e : -3.0e38f);
            const float wany = wave_max_dpp((inw && 1.f - amk[0] == 0.f) ? 1.f : 0.f);
            if (lane == 0) { red[wave] = wmx; red[PD_NW + wave] = wany; }
            __syncthreads();
            float mx = red[0], anyone = red[PD_NW];
#pragma unroll
            for (int x = 1; x < PD_NW; ++x) { mx = fmaxf(mx, red[x]); anyone = fmaxf(anyone, red[PD_NW + x]); }
            float u = 0.f;
            if (inw) {
                if (a.normalizer == 0) u = __expf(e - mx) * amk[0];
                else if (a.normalizer == 1) u = sigmoidf_(e) * amk[0];
                else u = fmaxf(e / 1000.f, 0.f) * amk[0];
            }
            const float wsm = wave_sum_dpp(u);
            if (lane == 0) red[2 * PD_NW + wave] = wsm;
            __syncthreads();
            float ssum = 0.f;
#pragma unroll
            for (int x = 0; x < PD_NW; ++x) ssum += red[2 * PD_NW + x];
            const float Z = ssum + (anyone > 0.f ? 0.f : 1.f);
            const float alpha = inw ? u / Z : 0.f;
            if (t < Tp) {
                al[t] = alpha;
                if (p == 0) a.W[((size_t)(i + 1) * B + b) * Tp + t] = alpha;
            }
            if (p == 0 && tid == 0 && a.ZB) a.ZB[row] = Z;
        }
        __syncthreads();
        clk.mark(5);
        // ---- phase D: gate inputs of the own units = sum_t alpha_t AW[t] (the reassociated glimpse): lane q of a unit takes
        // the window positions q, q+8, ...; reset gate published first (the next exchange waits for r*s only)
        float gx, gu2, gr2;
        {
            float x0 = 0.f, x1 = 0.f, u0 = 0.f, u1 = 0.f, r0 = 0.f, r1 = 0.f;
            for (int t0 = wi.begin + q; t0 < wi.end; t0 += 4 * PD_KSPLIT) {         // 4 positions' loads in flight
                float av[4], vx[4], vu[4], vr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tt = t0 + e * PD_KSPLIT, tc = min(tt, Tp - 1);
                    const float araw = al[tc];                            // (unconditional read + select: a guarded read is a branch)
                    av[e] = tt < wi.end ? araw : 0.f;
                    const float* rowp = AWs + tc * PD_AWS + jl;
                    vx[e] = rowp[0]; vu[e] = rowp[PD_UNITS]; vr[e] = rowp[2 * PD_UNITS];
                }
                x0 += av[0] * vx[0]; u0 += av[0] * vu[0]; r0 += av[0] * vr[0];
                x1 += av[1] * vx[1]; u1 += av[1] * vu[1]; r1 += av[1] * vr[1];
                x0 += av[2] * vx[2]; u0 += av[2] * vu[2]; r0 += av[2] * vr[2];
                x1 += av[3] * vx[3]; u1 += av[3] * vu[3]; r1 += av[3] * vr[3];
            }
            gr2 = group_sum<PD_KSPLIT>(r0 + r1);
            gx = x0 + x1; gu2 = u0 + u1;
        }
        const float rr = sigmoid_fast(fr + gr + gr2);
        const float rs = junit ? rr * sj : 0.f;
        if (q == 0 && junit) {
            granule_store(gRS + j, epoch, rs, plain);
            a.R[row * D + j] = rr;
            a.RH[row * D + j] = rs;
        }
        const float uu = sigmoid_fast(fu + gu + group_sum<PD_KSPLIT>(gu2));
        const float xin = fx + group_sum<PD_KSPLIT>(gx);
        if (q == 0 && junit) a.U[row * D + j] = uu;
        clk.mark(6);
        // behind the r*s exchange: the next label's window and convolution features (expanding prior / content-only attention),
        // or the next window centre of this utterance (a sequential scan of the alignment by one wave)
        if (!winprior && i + 1 < L) {
            wnext = attdec_window(a, i + 1);
            if (KC > 0) conv(i + 1, wnext);
        }
        if (winprior && pos_wave && i + 1 < L) {            // next window centre
            const float r = attdec_pos_of_row_wave(a, al);
            if (lane == 0) {
                granule_store(gPOS + ((i + 1) & 1) * g.Bp + b, epoch + 1u, r);
                a.pos[(size_t)(i + 1) * B + b] = r;
            }
        }
        clk.mark(8);
        {
            float v[PD_NV];
            if (!pd_gather(gRS, D, epoch, abort_word, v)) return;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x)
                if (tid + x * PD_THREADS < D) rsv[pd_slot(tid + x * PD_THREADS, PD_KD)] = v[x];
        }
        __syncthreads();
        clk.mark(10);
        // ---- phase E: candidate, state update, label-mask blend
        const float cand = tanh_fast(xin + pd_dot<PD_KD, PD_KSPLIT>(whc, rsv, q));
        float sn = cand * uu + sj * (1.f - uu);
        sn = ym * sn + (1.f - ym) * sj;
        if (!junit) sn = 0.f;
        if (q == 0 && junit) {
            if (STACK || i + 1 < L) granule_store(gS + j, epoch, sn, plain);          // (the layer-1 cluster needs the last one too)
            a.C[row * D + j] = cand;
            a.S[((size_t)(i + 1) * B + b) * SLD + j] = sn;
        }
        sj = sn;
        clk.mark(11);
    }
    if (clk.on) {
        long long* out = (long long*)((char*)abort_word + 64);
        for (int x = 0; x < PD_NPROF; ++x) out[x] = clk.acc[1 + x];
    }
}

// weighted_averages[l,b,:] = sum_t weights[l+1,b,t] * attended[t,b,:] for all labels at once (compute_weighted_averages,
// libs/blocks/blocks/bricks/attention.py:236-256): the persistent loop never forms them.  Grid (ceil(E/256), B, ceil(L/16)): a
// work-group streams its utterance's attended columns once and keeps 16 labels' sums in registers.
#define GL_LAB 16
__global__ __launch_bounds__(256) void attdec_glimpses_kernel(AttDec a) {
    __shared__ float wl[GL_LAB][PD_MAXV];
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, l0 = blockIdx.z * GL_LAB;
    const int Tp = a.Tp, B = a.B, E = a.E, nl = min(GL_LAB, a.L - l0);
    for (int x = threadIdx.x; x < GL_LAB * Tp; x += 256) {
        const int r = x / Tp, t = x % Tp;
        wl[r][t] = r < nl ? a.W[((size_t)(l0 + r + 1) * B + b) * Tp + t] : 0.f;
    }
    __syncthreads();
    if (e >= E) return;
    float acc[GL_LAB];
#pragma unroll
    for (int r = 0; r < GL_LAB; ++r) acc[r] = 0.f;
    const float* Ab = a.A + (size_t)b * a.A_bs + e;
    int t = 0;
    for (; t + 4 <= Tp; t += 4) {
        float av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = Ab[(size_t)(t + u) * a.A_ts];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < GL_LAB; ++r) acc[r] += wl[r][t + u] * av[u];
    }
    for (; t < Tp; ++t) {
        const float av = Ab[(size_t)t * a.A_ts];
#pragma unroll
        for (int r = 0; r < GL_LAB; ++r) acc[r] += wl[r][t] * av;
    }
#pragma unroll
    for (int r = 0; r < GL_LAB; ++r)
        if (r < nl) a.WA[((size_t)(l0 + r) * B + b) * E + e] = acc[r];
}

extern "C" int lvsr_attdec_glimpses(void* stream, const lvsr_attdec_args* args) {
    LVSR_REQUIRE(args != nullptr, "lvsr_attdec_glimpses: null args");
    AttDec a;
    memcpy(&a, args, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_glimpses")) return rc;
    LVSR_REQUIRE(a.Tp <= PD_MAXV, "lvsr_attdec_glimpses: attended length %d > %d", a.Tp, PD_MAXV);
    hipLaunchKernelGGL(attdec_glimpses_kernel, dim3((a.E + 255) / 256, a.B, (a.L + GL_LAB - 1) / GL_LAB), dim3(256), 0,
                       (hipStream_t)stream, a);
    return lvsr_check_launch("lvsr_attdec_glimpses");
}

extern "C" long long lvsr_attdec_persist_ws_bytes(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PdGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pd_geom(a, g)) return 0;
    return 256 + ((long long)a.B * PD_NPLANE * PD_MAXV + 2 * g.Bp + (long long)a.B * PD_MAXP) * 8;
}

extern "C" int lvsr_attdec_fwd_persistent(void* stream, const lvsr_attdec_args* args, const lvsr_attdec_plain* plain, void* ws,
                                          int use_graph) {
    LVSR_REQUIRE(args != nullptr && plain != nullptr && ws != nullptr, "lvsr_attdec_fwd_persistent: null argument");
    AttDec a;
    memcpy(&a, args, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_fwd_persistent")) return rc;
    LVSR_REQUIRE(a.label0 == 0 && (a.S_ld == 0 || a.S_ld == a.D), "lvsr_attdec_fwd_persistent: runs all labels of contiguous state slots (label0 = 0, S_ld = D)");
    PdGeom g;
    LVSR_REQUIRE(pd_geom(a, g), "lvsr_attdec_fwd_persistent: configuration outside the persistent kernel's limits "
                 "(lvsr_attdec_persist_ws_bytes returns 0 for it)");
    LVSR_REQUIRE(plain->Ws && plain->Whg && plain->Whh && plain->AW, "lvsr_attdec_fwd_persistent: plain weights missing");
    const lvsr_attdec_plain w = *plain;
    hipStream_t s = (hipStream_t)stream;
    int* ab = (int*)ws;
    u64* planes = (u64*)((char*)ws + 256);
    const size_t bytes = ((size_t)a.B * PD_NPLANE * PD_MAXV + 2 * g.Bp + (size_t)a.B * PD_MAXP) * 8;
    auto enqueue = [&]() {
        (void)hipMemsetAsync(planes, 0, bytes, s);          // the abort word in front of the planes is sticky: cleared by the host only
        const dim3 block(PD_THREADS);
        PdGeom gp = g;
        for (gp.b0 = 0; gp.b0 < a.B; gp.b0 += g.nb) {
        gp.nb = min(g.nb, a.B - gp.b0);
        const dim3 grid(cluster_grid(gp.nb, g.P, 0));
#define PD_LAUNCH(KCV, SHAPE) hipLaunchKernelGGL((attdec_pfwd_kernel<KCV, SHAPE, false>), grid, block, 0, s, a, w, gp, planes, ab, PdStack())
#define PD_LAUNCH_KC(SHAPE)                       \
        switch (g.KC) {                           \
            case 0: PD_LAUNCH(0, SHAPE); break;   \
            case 4: PD_LAUNCH(4, SHAPE); break;   \
            case 10: PD_LAUNCH(10, SHAPE); break; \
            default: PD_LAUNCH(16, SHAPE); break; \
        }
        if (g.shape == 0) { PD_LAUNCH_KC(PdShape8) }
        else if (g.shape == 1) { PD_LAUNCH_KC(PdShape16) }
        else { PD_LAUNCH_KC(PdShape32) }
        }
#undef PD_LAUNCH_KC
#undef PD_LAUNCH
    };
    GraphKey key("attdec_pfwd");
    key.add(&a, sizeof(a));
    key.add(&w, sizeof(w));
    key.add(&ws, sizeof(ws));
    key.add(&g.prof, sizeof(g.prof));
    key.add(&g.shape, sizeof(g.shape));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_attdec_fwd_persistent");
}

extern "C" long long lvsr_attdec_stack2_persist_ws_bytes(const lvsr_attdec_args* args) {
    if (args == nullptr) return 0;
    AttDec a;
    memcpy(&a, args, sizeof(a));
    PdGeom g;
    if (a.Tp <= 0 || a.B <= 0 || a.L <= 0 || a.E <= 0 || a.D <= 0 || a.M <= 0 || a.K < 0 || !pd_geom(a, g, false, true)) return 0;
    return 256 + ((long long)a.B * PD_NPLANE_STACK * PD_MAXV + 2 * g.Bp + (long long)a.B * PD_MAXP) * 8;
}

extern "C" int lvsr_attdec_fwd_persistent_stack2(void* stream, const lvsr_attdec_args* args, const lvsr_attdec_plain* plain,
                                                 const lvsr_attdec_stack2* l1, void* ws, int use_graph) {
    LVSR_REQUIRE(args != nullptr && plain != nullptr && l1 != nullptr && ws != nullptr, "lvsr_attdec_fwd_persistent_stack2: null argument");
    AttDec a;
    memcpy(&a, args, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_fwd_persistent_stack2")) return rc;
    LVSR_REQUIRE(a.label0 == 0 && a.S_ld >= 2 * a.D, "lvsr_attdec_fwd_persistent_stack2: runs all labels; the state slots hold both layers side by side (S_ld >= 2 D)");
    PdGeom g;
    LVSR_REQUIRE(pd_geom(a, g, false, true), "lvsr_attdec_fwd_persistent_stack2: configuration outside the kernel's limits "
                 "(lvsr_attdec_stack2_persist_ws_bytes returns 0 for it)");
    LVSR_REQUIRE(plain->Ws && plain->Whg && plain->Whh && plain->AW, "lvsr_attdec_fwd_persistent_stack2: plain weights of layer 0 missing");
    LVSR_REQUIRE(l1->Whg1 && l1->Whh1 && l1->Ws1 && l1->F1 && l1->AW1 && l1->xg1 && l1->U1 && l1->R1 && l1->C1 && l1->RH1,
                 "lvsr_attdec_fwd_persistent_stack2: layer-1 block incomplete");
    const lvsr_attdec_plain w = *plain;
    const PdStack k2 = *l1;
    hipStream_t s = (hipStream_t)stream;
    int* ab = (int*)ws;
    u64* planes = (u64*)((char*)ws + 256);
    const size_t bytes = ((size_t)a.B * PD_NPLANE_STACK * PD_MAXV + 2 * g.Bp + (size_t)a.B * PD_MAXP) * 8;
    auto enqueue = [&]() {
        (void)hipMemsetAsync(planes, 0, bytes, s);
        const dim3 grid(cluster_grid(a.B, 2 * g.P, 0)), block(PD_THREADS);
        switch (g.KC) {
            case 0: hipLaunchKernelGGL((attdec_pfwd_kernel<0, PdShape8, true>), grid, block, 0, s, a, w, g, planes, ab, k2); break;
            case 4: hipLaunchKernelGGL((attdec_pfwd_kernel<4, PdShape8, true>), grid, block, 0, s, a, w, g, planes, ab, k2); break;
            case 10: hipLaunchKernelGGL((attdec_pfwd_kernel<10, PdShape8, true>), grid, block, 0, s, a, w, g, planes, ab, k2); break;
            default: hipLaunchKernelGGL((attdec_pfwd_kernel<16, PdShape8, true>), grid, block, 0, s, a, w, g, planes, ab, k2); break;
        }
    };
    GraphKey key("attdec_pfwd_stack2");
    key.add(&a, sizeof(a));
    key.add(&w, sizeof(w));
    key.add(&k2, sizeof(k2));
    key.add(&ws, sizeof(ws));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_attdec_fwd_persistent_stack2");
}
