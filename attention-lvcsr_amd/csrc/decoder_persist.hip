// Persistent attention-decoder forward for gfx950: ONE launch for the whole label loop instead of five launches per label.
//
// The decoder's chain per label is  s -> s W_s -> energies -> softmax -> glimpse -> gates -> r*s -> candidate -> s'; as separate
// launches every all-to-all dependency costs a kernel boundary (~3-5 us, decoder_fwd.hip: 33 us per label on WSJ-base), inside
// one launch it costs a granule hand-off between the CUs of a cluster (~0.85 us, persist.h / encoder_persist.hip).  Utterances
// are independent, so a cluster of P work-groups (256 threads each: one wave per SIMD, one work-group per CU) serves ONE
// utterance and everything the label loop re-reads stays on chip for the whole sequence:
//   * thread (unit jl = tid/8, slice q = tid%8) keeps, in REGISTERS, its 1/8 row slice of the three decoder-GRU state columns of
//     unit j = 32 p + jl (state_to_gates u/r, state_to_state: 3 x 32 values) and of two columns of transform_states (2 x 32);
//     thread m keeps columns m and m + 256 of the location handler (2K values) and of the energy vector;
//   * the glimpse never exists inside the loop: the gate inputs it feeds are linear in it, wa W_d = sum_t alpha_t (A_t W_d), so
//     the caller precomputes AW = attended @ [fork_inputs.W | fork_gate_inputs.W] (one GEMM per batch) and the work-group keeps
//     the rows of ITS 96 gate columns in LDS: the gate inputs become a local contraction with the alignment every work-group
//     already holds (no glimpse exchange, no glimpse weights in registers: 160 instead of 352 per thread).  The weighted
//     averages themselves (readout, weight gradients) are one batched kernel after the loop (lvsr_attdec_glimpses);
//   * in LDS: the work-group's rows of the preprocessed attended (positions t = p mod P: interleaved, so any window is balanced
//     over the cluster), its AW columns (all positions), the convolution filters and the small vectors;
//   * per label four phase vectors travel as {epoch,value} granules (transformed state M, energies T', r*s D, s' D), everything
//     else is local.  Work that does not sit on the chain runs in the shadow of a hand-off: the location convolution of the NEXT
//     label (needs only the new alignment; on the matrix cores) and the update-gate / candidate-input sums behind the r*s
//     exchange, the saved tensors' plain stores anywhere.
// The energies (T'/P positions x M per work-group and label: K FMAs + tanh each) are the VALU-heavy part; position sums over
// the 256 threads use a register butterfly (16 positions per round, 17 shuffles) instead of 16 block reductions.
//
// Same results as decoder_fwd.hip up to float32 rounding (order of additions; the reassociated glimpse); it writes every tensor
// the backward pass reads (sW, CV, EN, ZB, W, U, R, C, RH, S, pos; WA through lvsr_attdec_glimpses).  Limits (else the caller
// uses the step kernels): D <= 256, M <= 512, T' <= 512, M <= 64 P with P = ceil(D/32), B P <= 224 work-groups, the LDS
// budget below (T' <= ~205 at WSJ-base dims).  With the window_around_* priors the window centres of all utterances bound the
// window: one more (B-granule) exchange per label, between all clusters.
#include "decoder.h"
#include "persist.h"
#include <stdlib.h>
#include <string.h>

#define PD_THREADS 256       // one wave per SIMD: 512 registers per lane (VGPR + AGPR), the weights' home
#define PD_KSPLIT 8
#define PD_UNITS 32          // decoder units per work-group
#define PD_KD 32             // state rows per thread
#define PD_MC 2              // transform_states columns per lane group
#define PD_AWS (3 * PD_UNITS + 4)   // LDS row stride of the AW slice: +4 words spreads the 8 position lanes of a unit over the banks
#define PD_CH 16             // attended positions per butterfly round
#define PD_MAXV 512          // longest exchanged vector
#define PD_NV (PD_MAXV / PD_THREADS)     // granules per thread and sweep
#define PD_NW (PD_THREADS / 64)
#define PD_LDS_FLOATS (39 * 1024 + 512)
#define PD_NPLANE 4          // SW | EN | RS | S
#define PD_NPROF 16

struct PdGeom {
    int P, nown, nownp, KC, KCP, FW;
    int o_pa, o_a, o_f, o_cv, o_al, o_sv, o_rs, o_xw, o_red, o_pos, o_clk, o_cp, Bp, total;
};

__host__ __device__ __forceinline__ int pd_slot(int k, int KX) { return (k / KX) * (KX + 4) + (k % KX); }

static int pd_kc(int K) {
    const int inst[4] = {0, 4, 10, 16};
    for (int i = 0; i < 4; ++i)
        if (K <= inst[i]) return inst[i];
    return -1;
}

static bool pd_geom(const AttDec& a, PdGeom& g) {
    if ((a.phases & 3) != 3 || a.step_dev != nullptr) return false;
    if (a.D > PD_KSPLIT * PD_KD || a.M > PD_MAXV || a.Tp > PD_MAXV) return false;
    g.KC = pd_kc(a.K);
    if (g.KC < 0) return false;
    g.KCP = (g.KC + 3) / 4 * 4;
    g.P = (a.D + PD_UNITS - 1) / PD_UNITS;
    if (a.M > g.P * PD_MC * PD_UNITS || a.B * g.P > PERSIST_MAX_WG) return false;
    g.nown = (a.Tp + g.P - 1) / g.P;
    g.nownp = (g.nown + PD_CH - 1) / PD_CH * PD_CH;
    g.FW = 2 * a.c + 1;
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) / 4 * 4; return at; };
    g.o_pa = take(g.nown * a.M);
    g.o_a = take(a.Tp * PD_AWS);
    g.o_f = take(a.K * g.FW);
    g.o_cv = take(g.nownp * (g.KCP > 0 ? g.KCP : 4));
    g.o_al = take(a.Tp);
    g.o_sv = take(PD_KSPLIT * (PD_KD + 4));
    g.o_rs = take(PD_KSPLIT * (PD_KD + 4));
    g.o_xw = take(PD_NW * PD_CH);
    g.o_red = take(16);
    g.Bp = (a.B + 3) / 4 * 4;
    g.o_pos = take(2 * g.Bp);
    g.o_clk = take(2 * (PD_NPROF + 1));
    g.o_cp = take(PD_NW * 16 * 17);
    g.total = o;
    return o <= PD_LDS_FLOATS;
}

// Window of label i (attdec_window) with the window centres taken from LDS: the centres of ALL utterances bound the window
// (lvsr/bricks/attention.py:133-147), which is the one coupling between clusters
__device__ __forceinline__ Win pd_window(const AttDec& a, int i, const float* posv) {
    if (a.K == 0 || a.prior_type == 0) return attdec_window(a, i);
    const float before = (float)a.p0, after = (float)a.p1;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int b = 0; b < a.B; ++b) {
        const float pb = posv[b];
        mn = fminf(mn, floorf(pb - before));
        mx = fmaxf(mx, ceilf(pb + after));
    }
    Win w;
    w.begin = (int)fmaxf(0.f, mn);
    w.end = (int)fminf((float)a.Tp, mx);
    if (w.end < w.begin) w.end = w.begin;
    return w;
}

// sum / max over the waves of the work-group (all threads call; `red` has PD_NW floats)
__device__ __forceinline__ float pd_wg_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float pd_wg_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One sweep over a plane of n <= 512 granules until every granule carries `epoch`; thread tid gets granules tid and tid + 256.
// Returns false when the cluster gave up (spin limit / abort word).
__device__ __forceinline__ bool pd_gather(const u64* g, int n, unsigned epoch, int* abort_word, float (&out)[PD_NV]) {
    const int tid = threadIdx.x;
    u64 wv[PD_NV];
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) {
            wv[x] = (u64)epoch << 32;
            if (tid + x * PD_THREADS < n) wv[x] = __hip_atomic_load(g + tid + x * PD_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) ok = ok && (unsigned)(wv[x] >> 32) == epoch;
        if (__all(ok)) break;
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
    for (int x = 0; x < PD_NV; ++x) out[x] = __uint_as_float((unsigned)wv[x]);
    return true;
}

// sum_x w[x] * v[q][x] over this thread's row slice of a sliced LDS vector, folded over the lanes of the unit
template <int KX>
__device__ __forceinline__ float pd_dot(const f32x2 (&w)[KX / 2], const float* buf, int q) {
    const float4* hv = (const float4*)(buf + q * (KX + 4));
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < KX / 4; ++x) {
        const float4 h4 = hv[x];
        const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
        a0 = w[2 * x] * lo + a0;
        a1 = w[2 * x + 1] * hi + a1;
    }
    return group_sum<PD_KSPLIT>((a0.x + a1.x) + (a0.y + a1.y));
}

// v[x] (x < 16) -> sum over the 64 lanes of the wave; lane l ends with the total of value index
// 8*bit5(l) + 4*bit4(l) + 2*bit3(l) + bit2(l): every stage halves the values a lane carries (17 shuffles instead of 16 x 6)
__device__ __forceinline__ float pd_butterfly16(float (&v)[PD_CH]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int half = 8 >> s, off = 32 >> s;
        const bool up = (lane & off) != 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            if (x < half) {
                const float keep = up ? v[x + half] : v[x];
                const float send = up ? v[x] : v[x + half];
                v[x] = keep + __shfl_xor(send, off, 64);
            }
        }
    }
    float r = v[0];
    r += lvsr_dpp_quad_xor1(r);
    r += lvsr_dpp_quad_xor2(r);
    return r;
}
__device__ __forceinline__ int pd_butterfly_index(int lane) {
    return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}

// Phase clock of work-group 0 (thread 0): accumulated s_memrealtime ticks (100 MHz) per phase of the label loop, left in the
// workspace header (bytes 64..255) for tools/probe_decoder_persist.py.  A dozen scalar clock reads per label in one wave.
struct PdClock {          // accumulators in LDS (2 floats each), touched by thread 0 of work-group 0 only
    long long* acc;
    bool on;
    __device__ __forceinline__ void start(bool enable, float* mem) {
        on = enable;
        acc = (long long*)mem;
        if (on) {
            for (int x = 1; x <= PD_NPROF; ++x) acc[x] = 0;
            acc[0] = wall_clock64();
        }
    }
    __device__ __forceinline__ void mark(int slot) {
        if (on) {
            const long long now = wall_clock64();
            acc[1 + slot] += now - acc[0];
            acc[0] = now;
        }
    }
};

template <int KC>
__global__ __launch_bounds__(PD_THREADS) void attdec_pfwd_kernel(AttDec a, lvsr_attdec_plain w, PdGeom g, u64* planes, int* abort_word) {
    constexpr int KCP = (KC + 3) / 4 * 4;
    __shared__ __attribute__((aligned(16))) float lds[PD_LDS_FLOATS];
    float* const PAs = lds + g.o_pa;      // [nown][M]   own positions (t = tl*P + p) of the preprocessed attended
    float* const AWs = lds + g.o_a;       // [T'][PD_AWS] own gate columns of AW: [x | u | r][32 units]
    float* const Fs = lds + g.o_f;        // [K][FW]     convolution filters
    float* const cvs = lds + g.o_cv;      // [nownp][KCP] convolution features of the own positions
    float* const al = lds + g.o_al;       // [T']        current alignment
    float* const sv = lds + g.o_sv;       // state, sliced by PD_KD
    float* const rsv = lds + g.o_rs;      // r*s, sliced by PD_KD
    float* const xw = lds + g.o_xw;       // [PD_NW][PD_CH]
    float* const red = lds + g.o_red;
    float* const posv = lds + g.o_pos;    // [2][Bp] window centres of all utterances, by label parity
    float* const cp = lds + g.o_cp;       // [PD_NW][16][17] convolution partial tiles
    const int P = g.P, nown = g.nown;
    int b, p;
    cluster_of_block(P, 0, b, p);
    const int tid = threadIdx.x, q = tid & (PD_KSPLIT - 1), jl = tid / PD_KSPLIT, lane = tid & 63, wave = tid >> 6;
    const int D = a.D, M = a.M, Tp = a.Tp, B = a.B, K = a.K, L = a.L;
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
    // energy phase: thread tid holds match columns m = tid and tid + 256 (handler column pair per filter, energy vector pair)
    f32x2 Hk[KC > 0 ? KC : 1], we2;
    float am[PD_NV];
    {
        float h[2][KC > 0 ? KC : 1], wv[2];
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) {
            const int m = tid + x * PD_THREADS, t = m;
#pragma unroll
            for (int k = 0; k < KC; ++k) h[x][k] = (k < K && m < M) ? a.handler[(size_t)k * M + m] : 0.f;
            wv[x] = m < M ? a.w_e[m] : 0.f;
            am[x] = t < Tp ? a.Am[(size_t)t * a.Am_ts + (size_t)b * a.Am_bs] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) Hk[k] = (f32x2){h[0][k], h[1][k]};
        we2 = (f32x2){wv[0], wv[1]};
    }
    const float eb = a.e_bias ? a.e_bias[0] : 0.f;
    // ---- LDS residents
    for (int x = tid; x < g.total - g.o_cv; x += PD_THREADS) lds[g.o_cv + x] = 0.f;      // everything behind the big tables
    for (int x = tid; x < nown * M; x += PD_THREADS) {
        const int tl = x / M, m = x % M, t = tl * P + p;
        PAs[x] = t < Tp ? a.PA[(size_t)t * a.PA_ts + (size_t)b * a.PA_bs + m] : 0.f;
    }
    for (int x = tid; x < Tp * 3 * PD_UNITS; x += PD_THREADS) {
        const int t = x / (3 * PD_UNITS), gcol = x % (3 * PD_UNITS), gate = gcol / PD_UNITS, unit = p * PD_UNITS + gcol % PD_UNITS;
        AWs[t * PD_AWS + gcol] = unit < D ? w.AW[((size_t)t * B + b) * 3 * D + (size_t)gate * D + unit] : 0.f;
    }
    for (int x = tid; x < K * g.FW; x += PD_THREADS) Fs[x] = a.filters[x];
    __syncthreads();
    for (int t = tid; t < Tp; t += PD_THREADS) al[t] = a.W[(size_t)b * Tp + t];
    for (int k = tid; k < D; k += PD_THREADS) sv[pd_slot(k, PD_KD)] = a.S[(size_t)b * D + k];
    float sj = junit ? a.S[(size_t)b * D + j] : 0.f;
    u64* const gSW = planes + (size_t)b * PD_NPLANE * PD_MAXV;
    u64* const gEN = gSW + PD_MAXV;
    u64* const gRS = gSW + 2 * PD_MAXV;
    u64* const gS = gSW + 3 * PD_MAXV;
    u64* const gPOS = planes + (size_t)B * PD_NPLANE * PD_MAXV;      // [2][Bp], shared by all clusters
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

    // Location convolution of the alignment in `al` for the own positions with the window of label i (lvsr/expressions.py:28-54:
    // true convolution of the cut alignment): cv[k][t] = sum_d f[k][c+d] * al[t-d].  On the matrix cores: per block of 16 own
    // positions a 16(filters) x 16(positions) tile accumulates over the 2c+1 taps four at a time (v_mfma_f32_16x16x4_f32: A =
    // filter taps, B = shifted, window-masked alignment values); the four waves split the tap groups and fold through LDS.
    auto conv = [&](int i, const Win wi) {
        const int cn = a.c, FW = g.FW, ng4 = (FW + 3) / 4;
        const int r16 = lane & 15, kk = lane >> 4;
        const int frow = min(r16, K - 1) * FW;
        auto operands = [&](int g4, int tx, bool colok, float& fa, float& fb) {
            const int u = 4 * g4 + kk, idx = tx - (u - cn);
            fa = Fs[frow + min(u, FW - 1)];
            fa = (r16 < K && u < FW) ? fa : 0.f;
            fb = al[min(max(idx, 0), Tp - 1)];
            fb = (colok && idx >= wi.begin && idx < wi.end) ? fb : 0.f;
        };
        for (int bl = 0; bl * 16 < nown; ++bl) {
            const int tlx = bl * 16 + r16, tx = tlx * P + p;          // B operand: this lane's position column
            const bool colok = tlx < nown && tx < Tp;
            f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
            int g4 = wave;
            for (; g4 + PD_NW < ng4; g4 += 2 * PD_NW) {
                float fa0, fb0, fa1, fb1;
                operands(g4, tx, colok, fa0, fb0);
                operands(g4 + PD_NW, tx, colok, fa1, fb1);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0, fb0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fb1, acc1, 0, 0, 0);
            }
            if (g4 < ng4) {
                float fa0, fb0;
                operands(g4, tx, colok, fa0, fb0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0, fb0, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[(wave * 16 + kk * 4 + r) * 17 + r16] = acc0[r] + acc1[r];
            __syncthreads();
            {
                const int k = tid >> 4, x = tid & 15, tl = bl * 16 + x, t = tl * P + p;
                float sum = 0.f;
#pragma unroll
                for (int wv = 0; wv < PD_NW; ++wv) sum += cp[(wv * 16 + k) * 17 + x];
                if (k < K && tl < nown && t < Tp) {
                    const float val = (t >= wi.begin && t < wi.end) ? sum : 0.f;
                    cvs[tl * KCP + k] = val;
                    a.CV[(((size_t)i * B + b) * K + k) * Tp + t] = val;
                }
            }
            if ((bl + 1) * 16 < nown) __syncthreads();
        }
    };
    PdClock clk;
    clk.start(blockIdx.x == 0 && tid == 0, lds + g.o_clk);

    for (int i = 0; i < L; ++i) {
        const unsigned epoch = (unsigned)(i + 1);
        const size_t row = (size_t)i * B + b;
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
            for (int c = 0; c < PD_MC; ++c) sw[c] = pd_dot<PD_KD>(wsw[c], sv, q);
            if (q < PD_MC) {
                const int m = (p * PD_MC + q) * PD_UNITS + jl;
                float mine = sw[0];
#pragma unroll
                for (int c = 1; c < PD_MC; ++c) mine = q == c ? sw[c] : mine;
                if (m < M) {
                    granule_store(gSW + m, epoch, mine);
                    a.sW[row * M + m] = mine;
                }
            }
        }
        const float gu = pd_dot<PD_KD>(whu, sv, q), gr = pd_dot<PD_KD>(whr, sv, q);
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
        const Win wi = pd_window(a, i, posv + (i & 1) * g.Bp);
        float amk[PD_NV];                    // attended mask x window-around mask of this utterance (attdec_mask)
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) amk[x] = am[x];
        if (winprior) {
            const float pb = posv[(i & 1) * g.Bp + b];
            const float lo = floorf(pb - (float)a.p0), hi = ceilf(pb + (float)a.p1);
#pragma unroll
            for (int x = 0; x < PD_NV; ++x) {
                const float tf = (float)(tid + x * PD_THREADS);
                amk[x] *= (tf > lo && tf < hi) ? 1.f : 0.f;
            }
        }
        if (KC > 0) conv(i, wi);
        clk.mark(7);
        float swv[PD_NV];
        if (!pd_gather(gSW, M, epoch, abort_word, swv)) return;
        __syncthreads();                     // the convolution features are in place for everybody
        clk.mark(2);
        // ---- phase B: energies of the own positions, 16 per round; thread tid holds columns tid and tid + 256.  Branch-free
        // (clamped addresses, results masked afterwards): with one wave per SIMD the 16 positions' chains are the only ILP
        const int m0c = min(tid, M - 1), m1c = min(tid + PD_THREADS, M - 1);       // columns beyond M carry w_e = 0
        for (int tl0 = 0; tl0 < nown; tl0 += PD_CH) {
            float v[PD_CH];
#pragma unroll
            for (int x0 = 0; x0 < PD_CH; x0 += 4) {
                // four positions at a time, written operand-first / filter-major so that the four chains interleave
                f32x2 xx[4];
                float4 c4[4][KCP > 0 ? KCP / 4 : 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tlc = min(tl0 + x0 + e, nown - 1);
                    xx[e] = (f32x2){swv[0] + PAs[tlc * M + m0c], swv[1] + PAs[tlc * M + m1c]};
                    const float4* cr = (const float4*)(cvs + tlc * KCP);
#pragma unroll
                    for (int k4 = 0; k4 < KCP / 4; ++k4) c4[e][k4] = cr[k4];
                }
#pragma unroll
                for (int k4 = 0; k4 < KCP / 4; ++k4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 c = c4[e][k4];
                        if (4 * k4 < KC) xx[e] = Hk[4 * k4] * (f32x2){c.x, c.x} + xx[e];
                        if (4 * k4 + 1 < KC) xx[e] = Hk[4 * k4 + 1 < KC ? 4 * k4 + 1 : 0] * (f32x2){c.y, c.y} + xx[e];
                        if (4 * k4 + 2 < KC) xx[e] = Hk[4 * k4 + 2 < KC ? 4 * k4 + 2 : 0] * (f32x2){c.z, c.z} + xx[e];
                        if (4 * k4 + 3 < KC) xx[e] = Hk[4 * k4 + 3 < KC ? 4 * k4 + 3 : 0] * (f32x2){c.w, c.w} + xx[e];
                    }
                }
                float ex[4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) { ex[e][0] = __expf(2.0f * xx[e].x); ex[e][1] = __expf(2.0f * xx[e].y); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ex[e][0] = __builtin_amdgcn_rcpf(1.0f + ex[e][0]);
                    ex[e][1] = __builtin_amdgcn_rcpf(1.0f + ex[e][1]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tl = tl0 + x0 + e, t = tl * P + p;
                    const float val = we2.x * (1.0f - 2.0f * ex[e][0]) + we2.y * (1.0f - 2.0f * ex[e][1]);      // w_e . tanh_fast
                    v[x0 + e] = (tl < nown && t >= wi.begin && t < wi.end) ? val : 0.f;
                }
            }
            const float tot = pd_butterfly16(v);
            if ((lane & 3) == 0) xw[wave * PD_CH + pd_butterfly_index(lane)] = tot;
            __syncthreads();
            if (tid < PD_CH) {
                const int tl = tl0 + tid, t = tl * P + p;
                if (tl < nown && t < Tp) {
                    float e = 0.f;
#pragma unroll
                    for (int wv = 0; wv < PD_NW; ++wv) e += xw[wv * PD_CH + tid];
                    granule_store(gEN + t, epoch, (t >= wi.begin && t < wi.end) ? e : 0.f);
                }
            }
            if (tl0 + PD_CH < nown) __syncthreads();
        }
        clk.mark(3);
        float eg[PD_NV];
        if (!pd_gather(gEN, Tp, epoch, abort_word, eg)) return;
        clk.mark(4);
        // ---- phase C: normalisation over the window (every work-group, redundantly), glimpse of the own columns
        {
            bool inw[PD_NV];
            float e[PD_NV], u[PD_NV], mxl = -3.0e38f, anyl = 0.f;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x) {
                const int t = tid + x * PD_THREADS;
                inw[x] = t < Tp && t >= wi.begin && t < wi.end;
                e[x] = inw[x] ? eg[x] + eb : 0.f;
                if (p == 0 && t < Tp) a.EN[row * Tp + t] = e[x];          // pasted into zeros
                if (inw[x]) mxl = fmaxf(mxl, e[x]);
                if (inw[x] && 1.f - amk[x] == 0.f) anyl = 1.f;
            }
            const float mx = pd_wg_max(mxl, red);
            float sl = 0.f;
#pragma unroll
            for (int x = 0; x < PD_NV; ++x) {
                u[x] = 0.f;
                if (inw[x]) {
                    if (a.normalizer == 0) u[x] = __expf(e[x] - mx) * amk[x];
                    else if (a.normalizer == 1) u[x] = sigmoidf_(e[x]) * amk[x];
                    else u[x] = fmaxf(e[x] / 1000.f, 0.f) * amk[x];
                }
                sl += u[x];
            }
            const float ssum = pd_wg_sum(sl, red);
            const float anyone = pd_wg_max(anyl, red);
            const float Z = ssum + (anyone > 0.f ? 0.f : 1.f);
#pragma unroll
            for (int x = 0; x < PD_NV; ++x) {
                const int t = tid + x * PD_THREADS;
                const float alpha = inw[x] ? u[x] / Z : 0.f;
                if (t < Tp) {
                    al[t] = alpha;
                    if (p == 0) a.W[((size_t)(i + 1) * B + b) * Tp + t] = alpha;
                }
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
                    av[e] = tt < wi.end ? al[tc] : 0.f;
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
            granule_store(gRS + j, epoch, rs);
            a.R[row * D + j] = rr;
            a.RH[row * D + j] = rs;
        }
        const float uu = sigmoid_fast(fu + gu + group_sum<PD_KSPLIT>(gu2));
        const float xin = fx + group_sum<PD_KSPLIT>(gx);
        if (q == 0 && junit) a.U[row * D + j] = uu;
        clk.mark(6);
        // the next label's window centre of this utterance: a sequential scan of the alignment by one wave, behind the r*s exchange
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
        const float cand = tanh_fast(xin + pd_dot<PD_KD>(whc, rsv, q));
        float sn = cand * uu + sj * (1.f - uu);
        sn = ym * sn + (1.f - ym) * sj;
        if (!junit) sn = 0.f;
        if (q == 0 && junit) {
            if (i + 1 < L) granule_store(gS + j, epoch, sn);
            a.C[row * D + j] = cand;
            a.S[((size_t)(i + 1) * B + b) * D + j] = sn;
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
    return 256 + ((long long)a.B * PD_NPLANE * PD_MAXV + 2 * g.Bp) * 8;
}

extern "C" int lvsr_attdec_fwd_persistent(void* stream, const lvsr_attdec_args* args, const lvsr_attdec_plain* plain, void* ws,
                                          int use_graph) {
    LVSR_REQUIRE(args != nullptr && plain != nullptr && ws != nullptr, "lvsr_attdec_fwd_persistent: null argument");
    AttDec a;
    memcpy(&a, args, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_fwd_persistent")) return rc;
    PdGeom g;
    LVSR_REQUIRE(pd_geom(a, g), "lvsr_attdec_fwd_persistent: configuration outside the persistent kernel's limits "
                 "(lvsr_attdec_persist_ws_bytes returns 0 for it)");
    LVSR_REQUIRE(plain->Ws && plain->Whg && plain->Whh && plain->AW, "lvsr_attdec_fwd_persistent: plain weights missing");
    const lvsr_attdec_plain w = *plain;
    hipStream_t s = (hipStream_t)stream;
    int* ab = (int*)ws;
    u64* planes = (u64*)((char*)ws + 256);
    const size_t bytes = 256 + ((size_t)a.B * PD_NPLANE * PD_MAXV + 2 * g.Bp) * 8;
    auto enqueue = [&]() {
        (void)hipMemsetAsync(ws, 0, bytes, s);
        const dim3 grid(a.B * g.P), block(PD_THREADS);
        switch (g.KC) {
            case 0: hipLaunchKernelGGL(attdec_pfwd_kernel<0>, grid, block, 0, s, a, w, g, planes, ab); break;
            case 4: hipLaunchKernelGGL(attdec_pfwd_kernel<4>, grid, block, 0, s, a, w, g, planes, ab); break;
            case 10: hipLaunchKernelGGL(attdec_pfwd_kernel<10>, grid, block, 0, s, a, w, g, planes, ab); break;
            default: hipLaunchKernelGGL(attdec_pfwd_kernel<16>, grid, block, 0, s, a, w, g, planes, ab); break;
        }
    };
    GraphKey key("attdec_pfwd");
    key.add(&a, sizeof(a));
    key.add(&w, sizeof(w));
    key.add(&ws, sizeof(ws));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_attdec_fwd_persistent");
}
