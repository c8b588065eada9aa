// Attention decoder, forward (K5-K9 of SURVEY.md §2.3), step kernels: one decoder step = up to six small kernels
// (location convolution | state projections | energies | masked softmax + glimpse | GRU gates | GRU candidate), the
// whole label loop captured into one hipGraph.  The teacher-forced pass normally runs the persistent kernels instead
// (decoder_persist.hip); these serve generation / beam search (with row groups: several utterances' beams in one set of
// launches, lvsr_attdec_args.group_rows) and the shapes the clusters do not fit.  Same structure as encoder.hip:
// every cross-unit dependency inside a step is a kernel boundary, every contraction with the batch as
// the 16-row MFMA tile (rb_mm), everything else wave-parallel VALU work with shuffle reductions.
//
// Reference semantics (paths relative to the reference root):
//   take_glimpses / compute_energies / compute_weights  lvsr/bricks/attention.py:98-213
//   conv1d (true convolution, 'full' then [c:-c])        lvsr/expressions.py:28-54
//   ShallowEnergyComputer (tanh -> linear, no bias)      libs/blocks/blocks/bricks/attention.py:417-447
//   compute_weighted_averages                            libs/blocks/blocks/bricks/attention.py:236-256
//   compute_states = Distribute + GatedRecurrent step    libs/blocks/blocks/bricks/attention.py:625-662,
//                                                        libs/blocks/blocks/bricks/recurrent.py:608-620
#include "decoder.h"

__global__ __launch_bounds__(64) void attdec_pos_kernel(AttDec a, int slot) {
    const int b = blockIdx.x;          // one wave per alignment row
    const float r = attdec_pos_of_row_wave(a, a.W + ((size_t)slot * a.B + b) * a.Tp);
    if (threadIdx.x == 0) a.pos[(size_t)slot * a.B + b] = r;
}

// Location convolution (attdec_conv_kernel): a work-group serves one row and `kf` of its filters; a thread owns FOUR consecutive
// output positions of one filter and walks the taps four at a time — 16 FMAs on two 16-byte reads of the (zero-padded, cut)
// alignment and one of the filter from LDS.  (One output and one tap per step, as before round 4, is two LDS reads per FMA.)
#define PRE_FL 4096          // filter floats in LDS (kf filters, taps padded to a multiple of 4)
#define PRE_AL (ATT_MAX_T + ATT_MAX_FW + 16)
// ... and the sizes of the SMALL instantiation (T' + taps <= 1520: WSJ's 200..450 positions x 201 taps): 10 KB instead of 36 KB of
// LDS per work-group — the kernel is bound by the waves it has in flight (see the rejected eight-output variant)
#define PRE_FL_S 1024
#define PRE_AL_S 1536
struct PreGrid { int rt, ntS, ntG, nmm, nq, kf, nkg, nconv, small; };
__host__ __device__ __forceinline__ PreGrid attdec_pre_grid(const AttDec& a) {
    PreGrid g;
    g.rt = (a.B + 15) / 16;
    g.ntS = (a.phases & 1) ? (a.M + 15) / 16 : 0;
    g.ntG = (a.phases & 2) ? (2 * a.D + 15) / 16 : 0;
    g.nmm = (g.ntS + g.ntG) * g.rt;
    g.nq = (a.Tp + 3) / 4;                                           // output quads of a row
    const int fw4 = (2 * a.c + 1 + 3) / 4 * 4;
    g.small = a.Tp + fw4 + 16 <= PRE_AL_S && fw4 <= PRE_FL_S;
    g.kf = max(1, min(min(a.K, 256 / g.nq), (g.small ? PRE_FL_S : PRE_FL) / fw4));          // filters per work-group
    g.nkg = a.K > 0 ? (a.K + g.kf - 1) / g.kf : 0;
    g.nconv = ((a.phases & 1) && a.K > 0) ? a.B * g.nkg : 0;
    return g;
}

// pre: sW = s @ W_s (attention), sg = s @ W_hg (gate pre-activation, state part)
__global__ __launch_bounds__(256) void attdec_pre_kernel(AttDec a, int i) {
    const PreGrid g = attdec_pre_grid(a);
    const int D = a.D, B = a.B;
    int blk = blockIdx.x;
    {
        const int tileAll = blk % (g.ntS + g.ntG), b0 = (blk / (g.ntS + g.ntG)) * 16;
        const int ldS = a.S_ld ? a.S_ld : D;
        const float* Srow = a.S + ((size_t)i * B + b0) * ldS;
        const int b = b0 + (threadIdx.x >> 4);
        f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
        if (tileAll < g.ntS) {
            if (attdec_skip(a, b0) && attdec_skip(a, min(b0 + 15, B - 1)) && (a.group_rows == 0 || a.group_rows >= 16)) return;   // (a tile of 16 rows touches at most two groups then)
            rb_mm(acc0, acc1, row_src(Srow, ldS, B - b0, D), a.Ws_p, D, tileAll);
            const float v = rb_reduce(acc0, acc1);
            const int j = tileAll * 16 + (threadIdx.x & 15);
            if (b < B && j < a.M) a.sW[((size_t)i * B + b) * a.M + j] = v;
        } else {
            const int tile = tileAll - g.ntS;
            rb_mm(acc0, acc1, row_src(Srow, ldS, B - b0, D), a.Whg_p, D, tile);
            const float v = rb_reduce(acc0, acc1);
            const int j = tile * 16 + (threadIdx.x & 15);
            if (b < B && j < 2 * D) a.sg[(size_t)b * 2 * D + j] = v;
        }
    }
}

// pre, second part: cv = conv(alpha_prev) (a kernel of its own: its 36 KB of LDS would halve the occupancy of the products above)
template <int AL, int FL>
__global__ __launch_bounds__(256) void attdec_conv_kernel(AttDec a, int i) {
    __shared__ __attribute__((aligned(16))) float al[AL];
    __shared__ __attribute__((aligned(16))) float fl[FL];
    const PreGrid g = attdec_pre_grid(a);
    const int B = a.B, Tp = a.Tp, blk = blockIdx.x;
    const int kg = blk % g.nkg, b = blk / g.nkg, c = a.c, FW = 2 * c + 1, FW4 = (FW + 3) / 4 * 4;
    if (attdec_skip(a, b)) return;
    const int k0 = kg * g.kf, nk = min(g.kf, a.K - k0);
    const Win w = attdec_window_row(a, i, b);
    const float* wprev = a.W + ((size_t)i * B + b) * Tp;
    // al[OFF + t] = cut alignment at t, zeros from OFF - (c + 3) to OFF + Tp + c + 3; OFF such that the reads below are 16-byte aligned
    const int OFF = (c + 3 + 3) / 4 * 4 + ((3 - c) % 4 + 4) % 4;
    for (int x = threadIdx.x; x < OFF + Tp + c + 8; x += 256) {
        const int t = x - OFF;
        al[x] = (t >= w.begin && t < w.end) ? wprev[t] : 0.f;
    }
    for (int x = threadIdx.x; x < nk * FW4; x += 256) {
        const int k = x / FW4, e = x % FW4;
        fl[x] = e < FW ? a.filters[(size_t)(k0 + k) * FW + e] : 0.f;
    }
    __syncthreads();
    // true convolution of the CUT alignment: out[t] = sum_e f[e] * al[t + c - e] (e = c + d), the alignment zero outside the window.
    // Work items = (filter, quad of positions) over the quads that meet the WINDOW (round 6: 28 of the 50 quads of a 200-position
    // row under window_around_median(10, 100) — the threads whose quads lay outside idled while the others took two items each);
    // the positions outside the window are zeroed by a loop of their own.
    {
        float* rowout = a.CV + (((size_t)i * B + b) * a.K + k0) * Tp;
        for (int x = threadIdx.x; x < nk * Tp; x += 256) {
            const int t = x % Tp;
            if (t < w.begin || t >= w.end) rowout[x] = 0.f;
        }
    }
    const int q0 = max(w.begin, 0) >> 2, nqw = w.end > w.begin ? ((w.end + 3) >> 2) - q0 : 0;
    for (int x = threadIdx.x; x < nk * nqw; x += 256) {
        const int k = x / nqw, t = (q0 + x % nqw) * 4;
        const float* f = fl + k * FW4;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        // taps whose source positions t + c - e - 3 .. t + c - e + 4 reach into the window; the others multiply zeros of the cut
        // alignment (adding +-0 to a sum that started at +0 leaves it as it is: the same bits as the full loop).  Under the
        // window_around_* priors the window is ~110 of the 201 taps' reach: half the loop.
        const int e_lo = max(0, (t + c - 3 - (w.end - 1) + 3 - 7) & ~3), e_hi = min(FW4, t + c + 4 - w.begin + 1);
        if (t + 3 >= w.begin && t < w.end)
#pragma unroll 4
            for (int e = e_lo; e < e_hi; e += 4) {
                const float4 fe = *(const float4*)(f + e);
                const float* src = al + (OFF + t + c - e - 3);          // v[n] = al_cut[t + c - e - 3 + n]
                const float4 lo = *(const float4*)src, hi = *(const float4*)(src + 4);
                // out[t + j] += f[e + q] * v[j - q + 3]
                o0 += fe.x * lo.w; o0 += fe.y * lo.z; o0 += fe.z * lo.y; o0 += fe.w * lo.x;
                o1 += fe.x * hi.x; o1 += fe.y * lo.w; o1 += fe.z * lo.z; o1 += fe.w * lo.y;
                o2 += fe.x * hi.y; o2 += fe.y * hi.x; o2 += fe.z * lo.w; o2 += fe.w * lo.z;
                o3 += fe.x * hi.z; o3 += fe.y * hi.y; o3 += fe.z * hi.x; o3 += fe.w * lo.w;
            }
        float* out = a.CV + (((size_t)i * B + b) * a.K + k0 + k) * Tp;
        const float o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (t + j >= w.begin && t + j < w.end) out[t + j] = o[j];
    }
}

// energies: e[b,t] = w_e . tanh(PA[t,b,:] + sW[b,:] + cv[b,:,t] @ handler).  Grid (ceil(M/32), B, ceil(T'/64)): a
// work-group owns a 32-wide slice of the match dimension x 64 attended positions of one utterance; every thread
// issues all of its loads (handler column, 8 PA values, conv features) before the first use, and ~4 work-groups
// share a CU so their latencies overlap.  The slice's partial energies go to `ep`; the glimpse kernel folds the
// slices in a fixed order.
// Row groups (batched beam search: grid.y = groups): the rows of a group share the utterance's preprocessed attended and the
// window, so ONE work-group serves all of them for its (slice, tile) — PA tile, handler column and w_e are fetched once for
// the 16 hypotheses instead of once each (the kernel was bound by exactly that L2 traffic at 512 rows).
// A thread holds 8 CONSECUTIVE positions: its conv features come as two 16-byte LDS reads per filter.
template <int KC>      // compile-time bound of the filter loop, see att_kc()
__global__ __launch_bounds__(256) void attdec_energy_kernel(AttDec a, int i) {
    __shared__ __attribute__((aligned(16))) float cvs[ATT_KMAX][ATT_TT];
    __shared__ float cs[ATT_TT][ATT_MS + 1];
    const int slice = blockIdx.x, nslice = gridDim.x, B = a.B, Tp = a.Tp, M = a.M, K = a.K;
    const int rows = a.group_rows > 0 ? a.group_rows : 1, bfirst = blockIdx.y * rows;
    const int t0 = blockIdx.z * ATT_TT;
    if (attdec_skip(a, bfirst)) return;
    const Win w = attdec_window_row(a, i, bfirst);
    if (t0 >= w.end || t0 + ATT_TT <= w.begin) return;             // tile outside the window: nothing to add
    const int ml = threadIdx.x & 31, tg = threadIdx.x >> 5, m = slice * ATT_MS + ml;
    const bool mok = m < M;
    float pav[8];
    const float* pab = a.PA + (size_t)attdec_ctx(a, bfirst) * a.PA_bs + m;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int t = t0 + 8 * tg + r;
        pav[r] = (t >= w.begin && t < w.end && mok) ? pab[(size_t)t * a.PA_ts] : 0.f;
    }
    float Hk[KC > 0 ? KC : 1];
#pragma unroll
    for (int k = 0; k < KC; ++k) Hk[k] = (k < K && mok) ? a.handler[(size_t)k * M + m] : 0.f;
    const float we_m = mok ? a.w_e[m] : 0.f;
    for (int row = 0; row < rows; ++row) {
        const int b = bfirst + row;
        const float sw_m = mok ? a.sW[((size_t)i * B + b) * M + m] : 0.f;
        if (row > 0) __syncthreads();                              // the previous row's partials have been folded
        for (int x = threadIdx.x; x < K * ATT_TT; x += 256) {
            const int k = x / ATT_TT, tl = x % ATT_TT, t = t0 + tl;
            cvs[k][tl] = (t < Tp) ? a.CV[(((size_t)i * B + b) * K + k) * Tp + t] : 0.f;
        }
        __syncthreads();
        float x8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x8[r] = pav[r] + sw_m;
#pragma unroll
        for (int k = 0; k < KC; ++k)
            if (KC == K || k < K) {
                const float4 c0 = *(const float4*)&cvs[k][8 * tg], c1 = *(const float4*)&cvs[k][8 * tg + 4];
                x8[0] += c0.x * Hk[k]; x8[1] += c0.y * Hk[k]; x8[2] += c0.z * Hk[k]; x8[3] += c0.w * Hk[k];
                x8[4] += c1.x * Hk[k]; x8[5] += c1.y * Hk[k]; x8[6] += c1.z * Hk[k]; x8[7] += c1.w * Hk[k];
            }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int tl = 8 * tg + r, t = t0 + tl;
            cs[tl][ml] = (t >= w.begin && t < w.end && mok) ? we_m * tanh_fast(x8[r]) : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < ATT_TT) {
            const int t = t0 + threadIdx.x;
            if (t >= w.begin && t < w.end) {
                float e = 0.f;
#pragma unroll
                for (int j = 0; j < ATT_MS; ++j) e += cs[threadIdx.x][j];
                a.ep[((size_t)b * nslice + slice) * Tp + t] = e;
            }
        }
    }
}

// The same energies with the convolution-features x handler contraction on the matrix cores (location-aware attention, K > 0).
// Grid as above; wave w of the work-group owns positions [16 w, 16 w + 16) of the 64-position tile and the slice's two 16-column
// match tiles: C = c (PA + sW) (accumulator layout: lane (c16 = lane % 16, g4 = lane / 16) holds positions 4 g4 + r of column c16),
// A = convolution features (16 positions x 4 filters per step, straight from global memory: a feature is used by exactly one
// lane of one wave per slice), B = c * handler (resident), KCP / 4 v_mfma_f32_16x16x4_f32 per tile; then, with c = 2 log2(e),
// tanh(x) = 1 - 2 / (1 + 2^(c x)): e = sum(w_e) - 2 sum_m w_e[m] / (1 + 2^y) — exp2, rcp and one FMA per element, a DPP fold over
// the 16 column lanes, one 16-byte store of four positions' partial energies.  No LDS, no barrier; the PA tile, handler and w_e
// stay in registers for all rows of the group.  (The VALU kernel above spends 10 FMAs + 10 LDS reads per element on the
// contraction: 39.5 us per pass at 512 rows.)
template <int KCP>     // filters padded to a multiple of 4
__global__ __launch_bounds__(256) void attdec_energy_mfma_kernel(AttDec a, int i, int rpw) {
    // rpw = rows of the group this work-group serves (grid.y = groups x ceil(rows / rpw)): all of them keeps the PA tile's traffic
    // lowest, fewer gives more, shorter work-groups — a work-group's loop over 16 rows is the kernel's whole duration when the grid
    // is a single round of work-groups (32 utterances: 2 048)
    const int slice = blockIdx.x, nslice = gridDim.x, B = a.B, Tp = a.Tp, M = a.M, K = a.K;
    const int rows = a.group_rows > 0 ? a.group_rows : 1, nsub = (rows + rpw - 1) / rpw;
    const int bfirst = (blockIdx.y / nsub) * rows, rbeg = (blockIdx.y % nsub) * rpw, rend = min(rows, rbeg + rpw);
    if (attdec_skip(a, bfirst)) return;
    const Win w = attdec_window_row(a, i, bfirst);
    // position tiles count from the window's first position (round 6): a 111-position window is TWO tiles of 64, not the 2.7 it
    // touches on average of tiles laid from position 0 — the kernel is bound by its transcendentals (16 quarter-rate instructions per
    // row and wave: 1 024 rows x 16 slices x 4 waves x ~650 cycles over 1 024 SIMDs = the 35 us it took), a third of which were
    // spent on positions outside the window.  A position's partial energy does not depend on the tile that computes it (same
    // operands, same order over the match columns): identical bits.  38.7 -> 22.8 us per pass at 64 utterances x 16 hypotheses.
    const int t0 = w.begin + blockIdx.z * ATT_TT;
    if (t0 >= w.end) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
    if (t0 + 16 * wave >= w.end) return;          // this wave's 16 positions lie behind the window (no LDS, no barrier: a wave may leave alone)
    const float C2 = 2.885390081777927f;                            // 2 log2(e)
    const int tA = min(t0 + 16 * wave + c16, Tp - 1);               // position whose features this lane feeds (A operand)
    const int tC = t0 + 16 * wave + 4 * g4;                         // first of the four positions this lane accumulates
    float pa[2][4], Hb[2][KCP / 4], wet[2], wsum = 0.f;
    int mcol[2];
    const size_t ctx = (size_t)attdec_ctx(a, bfirst);
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
        const int m = slice * ATT_MS + 16 * tile + c16;
        const bool mok = m < M;
        mcol[tile] = min(m, M - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = min(tC + r, Tp - 1);
            pa[tile][r] = mok ? C2 * a.PA[(size_t)t * a.PA_ts + ctx * a.PA_bs + mcol[tile]] : 0.f;
        }
#pragma unroll
        for (int sq = 0; sq < KCP / 4; ++sq) {
            const int k = 4 * sq + g4;
            Hb[tile][sq] = (k < K && mok) ? C2 * a.handler[(size_t)k * M + mcol[tile]] : 0.f;
        }
        const float wv = mok ? a.w_e[mcol[tile]] : 0.f;
        wet[tile] = -2.f * wv;
        wsum += wv;
    }
    // Rows in chunks of EN_RC: the operands of a whole chunk (convolution features and state projections: the only loads of the row
    // loop) are fetched before its first row is worked on — one exposed round trip per chunk instead of one per row (round 5 fetched
    // one row ahead: the loop of 8-16 rows was a chain of ~1 us load latencies with a hundred instructions of work between them).
    // The arithmetic of a row is unchanged (bit-identical energies).
    constexpr int EN_RC = 8;
    for (int r0 = rbeg; r0 < rend; r0 += EN_RC) {
        float av[EN_RC][KCP / 4], sw[EN_RC][2];
#pragma unroll
        for (int rr = 0; rr < EN_RC; ++rr) {
            const int b = bfirst + min(r0 + rr, rend - 1);
#pragma unroll
            for (int sq = 0; sq < KCP / 4; ++sq) {
                const int k = 4 * sq + g4;
                av[rr][sq] = k < K ? a.CV[(((size_t)i * B + b) * K + k) * Tp + tA] : 0.f;
            }
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) sw[rr][tile] = C2 * a.sW[((size_t)i * B + b) * M + mcol[tile]];
        }
#pragma unroll
        for (int rr = 0; rr < EN_RC; ++rr) {
            if (r0 + rr >= rend) break;
            const int b = bfirst + r0 + rr;
            f32x4 acc[2];
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
                acc[tile] = (f32x4){pa[tile][0] + sw[rr][tile], pa[tile][1] + sw[rr][tile], pa[tile][2] + sw[rr][tile], pa[tile][3] + sw[rr][tile]};
#pragma unroll
            for (int sq = 0; sq < KCP / 4; ++sq)
#pragma unroll
                for (int tile = 0; tile < 2; ++tile)
                    acc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rr][sq], Hb[tile][sq], acc[tile], 0, 0, 0);
            float ra[4] = {wsum, wsum, wsum, wsum};
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
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
            if (c16 == 0) {
                float* ep = a.ep + ((size_t)b * nslice + slice) * Tp;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (tC + r >= w.begin && tC + r < w.end) ep[tC + r] = ra[r];
            }
        }
    }
}

// masked softmax over the window + glimpse; grid (ceil(E/32), B).  Chunk 0 also writes the new alignment
// row (pasted into zeros) and the next window centre.
// FUSED = false (row groups: grid (1, B)): the softmax part alone — the weighted averages of a group's rows share the group's
// attended sequence and are formed per group afterwards (attdec_group_wa_kernel): the attended rows are then read once per
// utterance instead of once per hypothesis and chunk.
template <bool FUSED>
__global__ __launch_bounds__(256) void attdec_glimpse_kernel(AttDec a, int i) {
    __shared__ float al[ATT_MAX_T];
    __shared__ float red[4];
    __shared__ float part[32][33];
    const int b = blockIdx.y, chunk = blockIdx.x, B = a.B, Tp = a.Tp, E = a.E;
    if (attdec_skip(a, b)) return;
    const Win w = attdec_window_row(a, i, b);
    __shared__ float en[ATT_MAX_T];
    // The attended rows do not depend on the alignment: fetch this thread's share (8 float4 = the first 256 positions of
    // the window) before the softmax, so the stream's latency hides behind it.  Block = 32 columns x 32 position groups.
    const int cg = threadIdx.x & 7, tg = threadIdx.x >> 3;
    const int col = chunk * 32 + cg * 4;
    const float* Ab = a.A + (size_t)attdec_ctx(a, b) * a.A_bs + col;
    const bool vec = ((a.A_ts & 3) == 0) && ((a.A_bs & 3) == 0) && ((((size_t)a.A) & 15) == 0);
    const int nvalid = E - col;
    float4 pv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int t = w.begin + tg + 32 * r;
        pv[r] = (FUSED && t < w.end) ? ld4g(Ab + (size_t)t * a.A_ts, nvalid, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int nslice = (a.M + ATT_MS - 1) / ATT_MS;
    const float* ep = a.ep + (size_t)b * nslice * Tp;
    const float eb = a.e_bias ? a.e_bias[0] : 0.f;
    float mx = -3.0e38f;
    for (int t = threadIdx.x; t < Tp; t += 256) {
        float e = 0.f;
        if (t >= w.begin && t < w.end) {
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
            int sl = 0;
            for (; sl + 3 < nslice; sl += 4) {          // independent loads in flight
                e0 += ep[(size_t)sl * Tp + t];
                e1 += ep[(size_t)(sl + 1) * Tp + t];
                e2 += ep[(size_t)(sl + 2) * Tp + t];
                e3 += ep[(size_t)(sl + 3) * Tp + t];
            }
            for (; sl < nslice; ++sl) e0 += ep[(size_t)sl * Tp + t];
            e = ((e0 + e1) + (e2 + e3)) + eb;
            mx = fmaxf(mx, e);
        }
        en[t] = e;
        if (chunk == 0) a.EN[((size_t)i * B + b) * Tp + t] = e;      // pasted into zeros
    }
    mx = block_max(mx, red);
    float s = 0.f, anyone = 0.f;
    for (int t = w.begin + threadIdx.x; t < w.end; t += 256) {
        const float m = attdec_mask(a, i, b, t);
        float u;
        if (a.normalizer == 0) u = expf(en[t] - mx) * m;                        // softmax: shift by the window maximum
        else if (a.normalizer == 1) u = sigmoidf_(en[t]) * m;                     // logistic ("smooth focus")
        else u = fmaxf(en[t] / 1000.f, 0.f) * m;                                  // relu
        al[t] = u;
        s += u;
        if (1.f - m == 0.f) anyone = 1.f;
    }
    s = block_sum(s, red);
    anyone = block_max(anyone, red);
    const float Z = s + (anyone > 0.f ? 0.f : 1.f);      // + all(1 - mask)  (lvsr/bricks/attention.py:210-212)
    __syncthreads();
    for (int t = threadIdx.x; t < Tp; t += 256) al[t] = (t >= w.begin && t < w.end) ? al[t] / Z : 0.f;
    __syncthreads();
    if (chunk == 0) {
        if (threadIdx.x == 0 && a.ZB) a.ZB[(size_t)i * B + b] = Z;
        float* wn = a.W + ((size_t)(i + 1) * B + b) * Tp;
        for (int t = threadIdx.x; t < Tp; t += 256) wn[t] = al[t];
        if (a.K > 0 && a.prior_type != 0 && threadIdx.x < 64) {
            const float r = attdec_pos_of_row_wave(a, al);
            if (threadIdx.x == 0) a.pos[(size_t)(i + 1) * B + b] = r;
        }
    }
    if (!FUSED) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int t = w.begin + tg + 32 * r;
        const float p = t < w.end ? al[t] : 0.f;
        acc.x += p * pv[r].x; acc.y += p * pv[r].y; acc.z += p * pv[r].z; acc.w += p * pv[r].w;
    }
    for (int t = w.begin + tg + 256; t < w.end; t += 32) {          // windows longer than 256 positions
        const float4 v = ld4g(Ab + (size_t)t * a.A_ts, nvalid, vec);
        const float p = al[t];
        acc.x += p * v.x; acc.y += p * v.y; acc.z += p * v.z; acc.w += p * v.w;
    }
    part[tg][cg * 4 + 0] = acc.x; part[tg][cg * 4 + 1] = acc.y; part[tg][cg * 4 + 2] = acc.z; part[tg][cg * 4 + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c2 = chunk * 32 + threadIdx.x;
        if (c2 < E) {
            float r = 0.f;
#pragma unroll
            for (int g = 0; g < 32; ++g) r += part[g][threadIdx.x];
            a.WA[((size_t)i * B + b) * E + c2] = r;
        }
    }
}

// Weighted averages of a row group (batched beam search): WA[r, :] = sum_t alpha[r, t] * attended_g[t, :] for the rows r of group g,
// which share the group's attended sequence.  Grid (ceil(E/64), groups, row chunks): a work-group keeps the group's alignments in LDS
// (rows <= GW_ROWS per pass) and streams 64 columns of the attended rows once for all of them (the fused glimpse kernel reads them
// once per row: 16 times here); thread = (column, row quad).
// The ORDER of the sum is the fused kernel's (attdec_glimpse_kernel<true>): 32 partial sums over the positions begin + p + 32 j of the
// window, j ascending, then the partials p = 0..31 added in turn — so that a hypothesis's glimpse, and with it every hypothesis of a
// search, comes out bit for bit the same whether its utterance is searched alone or beside others.
#define GW_ROWS 16
#define GW_FLOATS 12288
#define GW_CHUNK 32            // rows of a group per work-group (grid.z walks the chunks): beam 200 = 7 work-groups per (group, 64 columns)
                               // instead of ONE running 13 passes — 64 work-groups on 256 CUs, 49 us per call (round-5 profile)
__global__ __launch_bounds__(256) void attdec_group_wa_kernel(AttDec a, int i) {
    // [rows of a pass][partial p][j]: the alignment of window position p + 32 j, zero beyond the window; a partial's terms are consecutive
    __shared__ __attribute__((aligned(16))) float gw_al[GW_FLOATS];
    const int g = blockIdx.y, rows = a.group_rows, B = a.B, Tp = a.Tp, E = a.E;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
    if (attdec_skip(a, g * rows)) return;
    const Win w = attdec_window_row(a, i, g * rows);
    const int span = w.end - w.begin;
    const int J = max(4, (((span + 31) >> 5) + 3) & ~3);                   // terms of a partial, padded to whole 16-byte reads
    const int stride = 32 * J;
    const int per = max(1, min(GW_ROWS, GW_FLOATS / stride));              // rows per pass (16 up to T' = 768)
    const int r_end = min(rows, ((int)blockIdx.z + 1) * GW_CHUNK);
    constexpr int RQ = GW_ROWS / 4;
    const float* Ab = a.A + (size_t)g * a.A_bs + min(col, E - 1) + (size_t)w.begin * a.A_ts;
    for (int r0 = blockIdx.z * GW_CHUNK; r0 < r_end; r0 += per) {
        const int nr = min(per, r_end - r0);
        __syncthreads();
        for (int x = threadIdx.x; x < nr * stride; x += 256) {
            const int r = x / stride, y = x % stride, t = (y / J) + 32 * (y % J);
            gw_al[x] = t < span ? a.W[((size_t)(i + 1) * B + g * rows + r0 + r) * Tp + w.begin + t] : 0.f;
        }
        __syncthreads();
        // rows of this thread that exist in the pass read their own alignments, the others row 0's (and store nothing)
        const float* alr[RQ];
        float sum[RQ];
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            alr[q] = gw_al + (rq * RQ + q < nr ? rq * RQ + q : 0) * stride;
            sum[q] = 0.f;
        }
        for (int p0 = 0; p0 < 32; p0 += 4) {                               // four partials at a time: 16 independent chains
            float acc[RQ][4];
#pragma unroll
            for (int q = 0; q < RQ; ++q)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) acc[q][pp] = 0.f;
            for (int j0 = 0; j0 < J; j0 += 4) {
                float v[4][4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)                         // (beyond the window: any row of it, times a zero alignment)
                        v[pp][jj] = Ab[(size_t)max(min(p0 + pp + 32 * (j0 + jj), span - 1), 0) * a.A_ts];
#pragma unroll
                for (int q = 0; q < RQ; ++q)
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        const float4 al = *(const float4*)(alr[q] + (p0 + pp) * J + j0);
                        acc[q][pp] += al.x * v[pp][0];
                        acc[q][pp] += al.y * v[pp][1];
                        acc[q][pp] += al.z * v[pp][2];
                        acc[q][pp] += al.w * v[pp][3];
                    }
            }
#pragma unroll
            for (int q = 0; q < RQ; ++q)
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) sum[q] += acc[q][pp];
        }
        if (col < E)
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                const int r = rq * RQ + q;
                if (r < nr) a.WA[((size_t)i * B + g * rows + r0 + r) * E + col] = sum[q];
            }
    }
}

// GRU part 1: x_in = fork_x + wa @ W_di;  g = sigmoid(sg + fork_g + wa @ W_dg) -> u, r, rh = r*s
__global__ __launch_bounds__(256) void attdec_gru1_kernel(AttDec a, int i) {
    const int D = a.D, B = a.B, E = a.E;
    const int ntX = (D + 15) / 16;
    const int tileAll = blockIdx.x, b0 = blockIdx.y * 16;
    const int b = b0 + (threadIdx.x >> 4);
    const float* wa = a.WA + ((size_t)i * B + b0) * E;
    const size_t row = (size_t)i * B + b;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    if (tileAll < ntX) {
        const int j = tileAll * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < D;
        const float fx = ok ? a.xg[row * 3 * D + j] : 0.f;
        rb_mm(acc0, acc1, row_src(wa, E, B - b0, E), a.Wdi_p, E, tileAll);
        const float v = rb_reduce(acc0, acc1);
        if (ok) a.xin[(size_t)b * D + j] = fx + v;
    } else {
        const int tile = tileAll - ntX;
        const int j = tile * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < 2 * D;
        const float fg = ok ? a.xg[row * 3 * D + D + j] + a.sg[(size_t)b * 2 * D + j] : 0.f;
        const float sp = (ok && j >= D) ? a.S[row * (a.S_ld ? a.S_ld : D) + (j - D)] : 0.f;
        rb_mm(acc0, acc1, row_src(wa, E, B - b0, E), a.Wdg_p, E, tile);
        const float v = rb_reduce(acc0, acc1);
        if (ok) {
            const float g = sigmoid_fast(v + fg);
            if (j < D) a.U[row * D + j] = g;
            else { a.R[row * D + (j - D)] = g; a.RH[row * D + (j - D)] = g * sp; }
        }
    }
}

// GRU part 2: candidate, state update, label-mask blend; writes state slot i+1
__global__ __launch_bounds__(256) void attdec_gru2_kernel(AttDec a, int i) {
    const int D = a.D, B = a.B;
    const int tile = blockIdx.x, b0 = blockIdx.y * 16;
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < B && j < D;
    const size_t row = (size_t)i * B + b;
    const float xin = ok ? a.xin[(size_t)b * D + j] : 0.f;
    const float uu = ok ? a.U[row * D + j] : 0.f;
    const int ldS = a.S_ld ? a.S_ld : D;
    const float sp = ok ? a.S[row * ldS + j] : 0.f;
    const float m = (ok && a.ymask) ? a.ymask[row] : 1.f;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm(acc0, acc1, row_src(a.RH + ((size_t)i * B + b0) * D, D, B - b0, D), a.Whh_p, D, tile);
    const float v = rb_reduce(acc0, acc1);
    if (ok) {
        const float cand = tanh_fast(v + xin);
        float sn = cand * uu + sp * (1.f - uu);
        sn = m * sn + (1.f - m) * sp;
        a.C[row * D + j] = cand;
        a.S[((size_t)(i + 1) * B + b) * ldS + j] = sn;
    }
}

int attdec_check(const AttDec& a, const char* what) {
    LVSR_REQUIRE(a.Tp > 0 && a.B > 0 && a.L > 0 && a.E > 0 && a.D > 0 && a.M > 0 && a.K >= 0, "%s: bad dims", what);
    LVSR_REQUIRE(a.Tp <= ATT_MAX_T, "%s: attended length %d > %d", what, a.Tp, ATT_MAX_T);
    LVSR_REQUIRE(a.M <= ATT_MAX_M && a.K <= ATT_KMAX && a.K * a.Tp <= ATT_MAX_KT && a.K * (2 * a.c + 1) <= ATT_MAX_KF,
                 "%s: match dim / conv filters / attended length exceed the LDS budget (M<=%d, K<=%d, K*T'<=%d)", what,
                 ATT_MAX_M, ATT_KMAX, ATT_MAX_KT);
    LVSR_REQUIRE(a.K == 0 || 2 * a.c + 1 <= ATT_MAX_FW, "%s: conv filter too wide", what);
    LVSR_REQUIRE(a.prior_type >= 0 && a.prior_type <= 2, "%s: unknown prior type", what);
    LVSR_REQUIRE(a.K == 0 || a.prior_type == 0 || a.pos != nullptr, "%s: window_around_* priors need pos", what);
    LVSR_REQUIRE(a.skip == nullptr || (a.L == 1 && a.skip_stride >= 0), "%s: skip is for single-step (generation) calls", what);
    LVSR_REQUIRE(a.group_rows >= 0 && (a.group_rows == 0 || (a.B % a.group_rows == 0 && (a.step_dev == nullptr || a.step_stride > 0))),
                 "%s: the rows must be whole groups (B = %d, group_rows = %d) with a position counter each (step_stride)", what, a.B, a.group_rows);
    return LVSR_OK;
}

extern "C" int lvsr_attdec_fwd(void* stream, const lvsr_attdec_args* args, int use_graph) {
    LVSR_REQUIRE(args != nullptr, "lvsr_attdec_fwd: null args");
    AttDec a;
    memcpy(&a, args, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_fwd")) return rc;
    LVSR_REQUIRE((a.phases & 3) != 0, "lvsr_attdec_fwd: phases must select attention and/or GRU");
    LVSR_REQUIRE(a.label0 >= 0 && a.label0 < a.L, "lvsr_attdec_fwd: label0 outside [0, L)");
    LVSR_REQUIRE(a.S_ld == 0 || a.S_ld >= a.D, "lvsr_attdec_fwd: S_ld < D");
    hipStream_t s = (hipStream_t)stream;
    const PreGrid g = attdec_pre_grid(a);
    auto enqueue = [&]() {
        if ((a.phases & 1) && !(a.phases & 4) && a.K > 0 && a.prior_type != 0 && a.label0 == 0)
            hipLaunchKernelGGL(attdec_pos_kernel, dim3(a.B), dim3(64), 0, s, a, 0);
        for (int i = a.label0; i < a.L; ++i) {
            // Round 6, measured at 64 utterances x 16 hypotheses and rejected (profiles/r06_decode.md): convolution and state products
            // in ONE launch (21.0 us against 12.7 + 10.7 apart; the merged kernel takes the products' registers and the convolution's
            // LDS); four column tiles per work-group in the 16 x 16 tile products (fragments of the row tile fetched once: gru1
            // 20.7 -> 25.5 us, pre 10.7 -> 11.5, gru2 6.4 -> 7.7, readout merge 11.4 -> 15.0 — these kernels live off the NUMBER of
            // short work-groups in flight, not off their operand traffic)
            if (g.nconv > 0 && g.small) hipLaunchKernelGGL((attdec_conv_kernel<PRE_AL_S, PRE_FL_S>), dim3(g.nconv), dim3(256), 0, s, a, i);
            else if (g.nconv > 0) hipLaunchKernelGGL((attdec_conv_kernel<PRE_AL, PRE_FL>), dim3(g.nconv), dim3(256), 0, s, a, i);
            if (g.nmm > 0) hipLaunchKernelGGL(attdec_pre_kernel, dim3(g.nmm), dim3(256), 0, s, a, i);
            if (a.phases & 1) {
                const dim3 eg((a.M + ATT_MS - 1) / ATT_MS, a.group_rows > 0 ? a.B / a.group_rows : a.B, (a.Tp + ATT_TT - 1) / ATT_TT);
                // rows of a group per work-group of the MFMA kernel: all of them, or 8 while the grid is at most one round of
                // work-groups (measured with two batches in flight, tools/bench_decode.py: 32 utterances 1.52 / 1.47 / 1.52 ms per
                // utterance at 16 / 8 / 4 rows; 64 utterances 1.18 / 1.20 / 1.28 / 1.46 at 16 / 8 / 4 / 2: the chip is busy with
                // the other batch, shorter work-groups only add PA traffic)
                const int rows_g = a.group_rows > 0 ? a.group_rows : 1;
                const int knob_rpw = lvsr_knob(LVSR_KNOB_ENERGY_ROWS);
                // (wide beams — groups of 64 rows and more: 32 rows per work-group; beam 200, 2 x 8 utterances in flight: 18.5 / 16.8 /
                // 15.9 / 16.6 / 17.5 ms per utterance at 8 / 16 / 32 / 64 / 200 rows, round 5)
                const int rpw = knob_rpw > 0 ? min(knob_rpw, rows_g)
                              : rows_g >= 64 ? 32 : ((long long)eg.x * eg.y * eg.z <= 2048 ? min(8, rows_g) : rows_g);
                const dim3 egm(eg.x, eg.y * ((rows_g + rpw - 1) / rpw), eg.z);
                switch (a.K > 0 ? (a.K + 3) / 4 * 4 : 0) {          // location-aware attention: the contraction on the matrix cores
                    case 0: hipLaunchKernelGGL(attdec_energy_kernel<0>, eg, dim3(256), 0, s, a, i); break;
                    case 4: hipLaunchKernelGGL(attdec_energy_mfma_kernel<4>, egm, dim3(256), 0, s, a, i, rpw); break;
                    case 8: hipLaunchKernelGGL(attdec_energy_mfma_kernel<8>, egm, dim3(256), 0, s, a, i, rpw); break;
                    case 12: hipLaunchKernelGGL(attdec_energy_mfma_kernel<12>, egm, dim3(256), 0, s, a, i, rpw); break;
                    default: hipLaunchKernelGGL(attdec_energy_mfma_kernel<16>, egm, dim3(256), 0, s, a, i, rpw); break;
                }
                if (a.group_rows > 0) {
                    hipLaunchKernelGGL(attdec_glimpse_kernel<false>, dim3(1, a.B), dim3(256), 0, s, a, i);
                    hipLaunchKernelGGL(attdec_group_wa_kernel, dim3((a.E + 63) / 64, a.B / a.group_rows, (a.group_rows + GW_CHUNK - 1) / GW_CHUNK), dim3(256), 0, s, a, i);
                } else {
                    hipLaunchKernelGGL(attdec_glimpse_kernel<true>, dim3((a.E + 31) / 32, a.B), dim3(256), 0, s, a, i);
                }
            }
            if (a.phases & 2) {
                hipLaunchKernelGGL(attdec_gru1_kernel, dim3((a.D + 15) / 16 + (2 * a.D + 15) / 16, g.rt), dim3(256), 0, s, a, i);
                hipLaunchKernelGGL(attdec_gru2_kernel, dim3((a.D + 15) / 16, g.rt), dim3(256), 0, s, a, i);
            }
        }
    };
    GraphKey key("attdec_fwd");
    key.add(&a, sizeof(a));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_attdec_fwd");
}
