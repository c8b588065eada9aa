// Bidirectional GRU layer recurrence (K2/K3 forward, K12 backward) for gfx950.
//
// Reference semantics: GatedRecurrent.apply (libs/blocks/blocks/bricks/recurrent.py:608-620) under
// scan (:178-231), Bidirectional.apply (:655-663), time subsampling in Encoder.apply
// (lvsr/bricks/__init__.py:71-78).  Gate columns [:H] = update, [H:] = reset; no bias in the cell;
// the backward direction scans t = T-1..0 WITH THE SAME MASK (so it starts in the padding and holds
// the initial state until the last real frame).
//
// Structure: the two dependent contractions of a GRU step need every hidden unit of the previous
// phase, i.e. a chip-wide exchange twice per time step.  On MI355X the cheapest chip-wide
// synchronisation is a kernel boundary (~1.5 us, MI355X_MICROARCH.md "boundary" row) — cheaper than
// an in-kernel grid barrier (4-5 us, "barrier-xcd") — so a step is two small kernels
// (gates, candidate), both directions in one launch, captured once into a hipGraph per shape so the
// host never paces the loop.  Inside a kernel each 256-thread work-group owns a 16-column tile of the
// packed recurrent weights (L2 resident: the block->XCD placement repeats every step) and the batch
// (<=16 utterances per row tile) is the M dimension of v_mfma_f32_16x16x4_f32 (common.h, rb_mm).
//
// Layouts (all fp32, time-major like the reference):
//   xg   (T,B,6H)  input projections, direction d at column d*3H: [x_in (H) | gate_in_update (H) | gate_in_reset (H)]
//   y    (T,B,2H)  layer output [h_fwd | h_bwd] — also the state sequence used as h_{t-1}
//   u,r,c,rh (T,B,2H)  saved update gate, reset gate, candidate, reset*h_prev (direction d at column d*H)
#include "common.h"
#include "graph_cache.h"
#include "lvsr_hip.h"

typedef lvsr_bigru_fwd_args EncFwd;
typedef lvsr_bigru_bwd_args EncBwd0;

struct HPrev {   // h_{t-1} rows of one direction
    const float* base; long long ld;   // ld = 2H (rows of y) or 0 (initial state broadcast over the batch)
    __device__ __forceinline__ float at(int b, int k) const { return base[(size_t)b * ld + k]; }
};

__device__ __forceinline__ void enc_step_geometry(const EncFwd& a, const float* h0_d, int n, int dir, int& t, HPrev& hp) {
    t = dir == 0 ? n : a.T - 1 - n;
    const int tp = dir == 0 ? t - 1 : t + 1;
    if (n == 0) { hp.base = h0_d; hp.ld = 0; }
    else { hp.base = a.y + (size_t)tp * a.B * 2 * a.H + dir * a.H; hp.ld = 2 * a.H; }
}

// gates: g = sigmoid(h_prev @ W_hg + g_in[t]); u = g[:, :H], r = g[:, H:]; rh = r * h_prev
template <bool FAST>
__global__ __launch_bounds__(256) void enc_gates_kernel(const float* xg, const float* mask, const float* Whh0, const float* Whh1, const float* Whg0, const float* Whg1, const float* h00, const float* h01, float* y, float* ysub, float* u, float* r, float* c, float* rh, int sub, int T, int B, int H, int n) {
    // individual kernel arguments (not a by-value struct): the compiler then fetches all of them with one s_load burst in
    // the entry block instead of one dependent scalar load per first use (5 serialized round trips = ~0.6 us per launch)
    EncFwd a;
    a.xg = xg; a.mask = mask; a.y = y; a.ysub = ysub; a.u = u; a.r = r; a.c = c; a.rh = rh; a.sub = sub; a.T = T; a.B = B; a.H = H;
    const int dir = blockIdx.z;
    const float* Whh_d = dir ? Whh1 : Whh0;
    const float* Whg_d = dir ? Whg1 : Whg0;
    const float* h0_d = dir ? h01 : h00;
    const int tile = blockIdx.x, b0 = blockIdx.y * 16;
    int t; HPrev hp;
    enc_step_geometry(a, h0_d, n, dir, t, hp);
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < a.B && j < 2 * H;
    const size_t row = (size_t)t * a.B + b;
    // epilogue operands first, so their latency overlaps the contraction
    const float gin = ok ? a.xg[row * 6 * H + dir * 3 * H + H + j] : 0.f;
    const float hpj = (ok && j >= H) ? hp.at(b, j - H) : 0.f;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm_sel<FAST>(acc0, acc1, row_src(hp.base + (size_t)b0 * hp.ld, hp.ld, a.B - b0, H), Whg_d, H, tile);
    const float v = rb_reduce_once(acc0, acc1);
    if (ok) {
        const float g = sigmoidf_(v + gin);
        if (j < H) {
            a.u[row * 2 * H + dir * H + j] = g;
        } else {
            const int jj = j - H;
            a.r[row * 2 * H + dir * H + jj] = g;
            a.rh[row * 2 * H + dir * H + jj] = g * hpj;
        }
    }
}

// candidate + state update + mask blend; writes y[t] (and the subsampled copy)
template <bool FAST>
__global__ __launch_bounds__(256) void enc_cand_kernel(const float* xg, const float* mask, const float* Whh0, const float* Whh1, const float* Whg0, const float* Whg1, const float* h00, const float* h01, float* y, float* ysub, float* u, float* r, float* c, float* rh, int sub, int T, int B, int H, int n) {
    // individual kernel arguments (not a by-value struct): the compiler then fetches all of them with one s_load burst in
    // the entry block instead of one dependent scalar load per first use (5 serialized round trips = ~0.6 us per launch)
    EncFwd a;
    a.xg = xg; a.mask = mask; a.y = y; a.ysub = ysub; a.u = u; a.r = r; a.c = c; a.rh = rh; a.sub = sub; a.T = T; a.B = B; a.H = H;
    const int dir = blockIdx.z;
    const float* Whh_d = dir ? Whh1 : Whh0;
    const float* Whg_d = dir ? Whg1 : Whg0;
    const float* h0_d = dir ? h01 : h00;
    const int tile = blockIdx.x, b0 = blockIdx.y * 16;
    int t; HPrev hp;
    enc_step_geometry(a, h0_d, n, dir, t, hp);
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < a.B && j < H;
    const size_t row = (size_t)t * a.B + b;
    const float xin = ok ? a.xg[row * 6 * H + dir * 3 * H + j] : 0.f;
    const float uu = ok ? a.u[row * 2 * H + dir * H + j] : 0.f;
    const float hprev = ok ? hp.at(b, j) : 0.f;
    const float m = (ok && a.mask) ? a.mask[row] : 1.f;
    const float* rh_t = a.rh + ((size_t)t * a.B + b0) * 2 * H + dir * H;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm_sel<FAST>(acc0, acc1, row_src(rh_t, 2 * H, a.B - b0, H), Whh_d, H, tile);
    const float v = rb_reduce_once(acc0, acc1);
    if (ok) {
        const float cand = tanhf(v + xin);
        float hn = cand * uu + hprev * (1.f - uu);
        hn = m * hn + (1.f - m) * hprev;
        a.c[row * 2 * H + dir * H + j] = cand;
        a.y[row * 2 * H + dir * H + j] = hn;
        if (a.ysub && (t % a.sub) == 0) a.ysub[((size_t)(t / a.sub) * a.B + b) * 2 * H + dir * H + j] = hn;
    }
}

// ---------------------------------------------------------------------------------------------
// BPTT.  Per step (fwd direction walks t = T-1..0, bwd direction t = 0..T-1):
//   dhn = m*dh; dc = dhn*u; du = dhn*(c-h_prev); dpre_c = dc*(1-c^2)
//   drh = dpre_c @ Whh^T   and   vu = dpre_u @ Whg[:, :H]^T   (kernel A: two independent tile sets, both K = H)
//   dr = drh*h_prev; dpre_u = du*u*(1-u); dpre_r = dr*r*(1-r)
//   dh_prev = dhn*(1-u) + (1-m)*dh + drh*r + vu + dpre_r @ Whg[:, H:]^T + dy[t_prev]   (kernel B, K = H)
//   dxg[t] = [dpre_c | dpre_u | dpre_r]
// ---------------------------------------------------------------------------------------------
struct EncBwd {
    EncBwd0 a;
    float* dh;       // (2,Bp,H) running dL/dh_t   (Bp = B rounded up to 16)
    float* dhpart;   // (2,Bp,H) elementwise part of dh_prev (+ drh*r)
    float* vu;       // (2,Bp,H) dpre_u @ Whg_u^T (independent of drh, computed next to kernel A)
    float* dpc;      // (2,Bp,H) dpre_c of the CURRENT step, prepared by the previous step's kernel B (or the init kernel)
    float* dpu;      // (2,Bp,H) dpre_u of the current step
    float* base;     // (2,Bp,H) dhn*(1-u) + (1-m)*dh of the current step
    int Bp;
    long long rofs;  // offset of the reset block inside WhgT_p
};

__device__ __forceinline__ float enc_dy_at(const EncBwd0& a, int t, int b, int dir, int j) {
    if (t < 0 || t >= a.T || (t % a.sub) != 0) return 0.f;
    return a.dy[((size_t)(t / a.sub) * a.B + b) * 2 * a.H + dir * a.H + j];
}

// Everything of step t that depends on dh_t elementwise only: dpre_c, dpre_u (also the dxg outputs) and the elementwise
// part of dh_{t-1}.  Called by the init kernel for the first step and by kernel B's epilogue for the following one, so that
// kernel A's contraction operands are plain (B,H) rows.  The saved activations it needs do not depend on the recurrence:
// kernel B fetches them (EncPrep) before its contraction.
struct EncPrep { float uu, cc, hprev, m; };
__device__ __forceinline__ EncPrep enc_bwd_prefetch(const EncBwd& e, const float* h0_d, int dir, int t, int b, int j) {
    const EncBwd0& a = e.a;
    const int H = a.H, tp = dir == 0 ? t - 1 : t + 1;
    const size_t o = ((size_t)t * a.B + b) * 2 * H + dir * H + j;
    EncPrep p;
    p.uu = a.u[o]; p.cc = a.c[o];
    p.hprev = (tp < 0 || tp >= a.T) ? h0_d[j] : a.y[((size_t)tp * a.B + b) * 2 * H + dir * H + j];
    p.m = a.mask ? a.mask[(size_t)t * a.B + b] : 1.f;
    return p;
}
__device__ __forceinline__ void enc_bwd_prepare(const EncBwd& e, const EncPrep& p, int dir, int t, int b, int j, float dhv) {
    const EncBwd0& a = e.a;
    const int H = a.H;
    const float dhn = p.m * dhv;
    const float dpc = dhn * p.uu * (1.f - p.cc * p.cc);
    const float dpu = dhn * (p.cc - p.hprev) * p.uu * (1.f - p.uu);
    const size_t q = ((size_t)dir * e.Bp + b) * H + j;
    e.dpc[q] = dpc;
    e.dpu[q] = dpu;
    e.base[q] = dhn * (1.f - p.uu) + (1.f - p.m) * dhv;
    float* dx = a.dxg + ((size_t)t * a.B + b) * 6 * H + dir * 3 * H;
    dx[j] = dpc;
    dx[H + j] = dpu;
}

__global__ __launch_bounds__(256) void enc_bwd_init_kernel(EncBwd e) {
    const EncBwd0& a = e.a;
    const int dir = blockIdx.z;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= e.Bp * a.H) return;
    const int b = idx / a.H, j = idx % a.H;
    const int t = dir == 0 ? a.T - 1 : 0;
    const size_t q = ((size_t)dir * e.Bp + b) * a.H + j;
    if (b < a.B) {
        const float dhv = enc_dy_at(a, t, b, dir, j);
        e.dh[q] = dhv;
        enc_bwd_prepare(e, enc_bwd_prefetch(e, a.h0[dir], dir, t, b, j), dir, t, b, j, dhv);
    } else {
        e.dh[q] = 0.f; e.dpc[q] = 0.f; e.dpu[q] = 0.f; e.base[q] = 0.f;
    }
}

__device__ __forceinline__ void enc_bwd_geometry(const EncBwd0& a, const float* h0_d, int n, int dir, int& t, int& tp, HPrev& hp) {
    t = dir == 0 ? a.T - 1 - n : n;
    tp = dir == 0 ? t - 1 : t + 1;
    if (tp < 0 || tp >= a.T) { hp.base = h0_d; hp.ld = 0; }
    else { hp.base = a.y + (size_t)tp * a.B * 2 * a.H + dir * a.H; hp.ld = 2 * a.H; }
}

// kernel A: tiles [0,nt): drh = dpre_c @ Whh^T, dpre_r, dhpart;  tiles [nt,2nt): vu = dpre_u @ Whg[:, :H]^T
template <bool FAST>
__global__ __launch_bounds__(256) void enc_bwd_a_kernel(const float* mask, const float* y, const float* u, const float* r, const float* c, const float* WhhT0, const float* WhhT1, const float* WhgT0, const float* WhgT1, const float* h00, const float* h01, const float* dy, float* dxg, float* dh, float* dhpart, float* vu, float* dpc, float* dpu, float* base, long long rofs, int sub, int T, int B, int H, int Bp, int n) {
    EncBwd e;                    // flat kernel arguments, see enc_gates_kernel
    e.a.mask = mask; e.a.y = y; e.a.u = u; e.a.r = r; e.a.c = c; e.a.dy = dy; e.a.dxg = dxg; e.a.sub = sub; e.a.T = T; e.a.B = B; e.a.H = H;
    e.dh = dh; e.dhpart = dhpart; e.vu = vu; e.dpc = dpc; e.dpu = dpu; e.base = base; e.Bp = Bp; e.rofs = rofs;
    const int dir = blockIdx.z;
    const float* WhhT_d = dir ? WhhT1 : WhhT0;
    const float* WhgT_d = dir ? WhgT1 : WhgT0;
    const float* h0_d = dir ? h01 : h00;
    const EncBwd0& a = e.a;
    const int b0 = blockIdx.y * 16, nt = (H + 15) / 16;
    const bool vu_path = (int)blockIdx.x >= nt;
    const int tile = vu_path ? blockIdx.x - nt : blockIdx.x;
    int t, tp; HPrev hp;
    enc_bwd_geometry(a, h0_d, n, dir, t, tp, hp);
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < a.B && j < H;
    const size_t q = ((size_t)dir * e.Bp + b) * H + j;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    if (vu_path) {
        rb_mm_sel<FAST>(acc0, acc1, row_src(e.dpu + ((size_t)dir * e.Bp + b0) * H, H, a.B - b0, H), WhgT_d, H, tile);
        const float v = rb_reduce_once(acc0, acc1);
        if (ok) e.vu[q] = v;
        return;
    }
    const size_t o = ((size_t)t * a.B + b) * 2 * H + dir * H + j;
    const float rr = ok ? a.r[o] : 0.f;
    const float hprev = ok ? hp.at(b, j) : 0.f;
    const float bs = ok ? e.base[q] : 0.f;
    rb_mm_sel<FAST>(acc0, acc1, row_src(e.dpc + ((size_t)dir * e.Bp + b0) * H, H, a.B - b0, H), WhhT_d, H, tile);
    const float drh = rb_reduce_once(acc0, acc1);
    if (ok) {
        a.dxg[((size_t)t * a.B + b) * 6 * H + dir * 3 * H + 2 * H + j] = drh * hprev * rr * (1.f - rr);
        e.dhpart[q] = bs + drh * rr;
    }
}

// kernel B: dh_prev = dhpart + vu + dpre_r @ Whg[:, H:]^T + dy[t_prev]; then prepares the next step's elementwise terms
template <bool FAST>
__global__ __launch_bounds__(256) void enc_bwd_b_kernel(const float* mask, const float* y, const float* u, const float* r, const float* c, const float* WhhT0, const float* WhhT1, const float* WhgT0, const float* WhgT1, const float* h00, const float* h01, const float* dy, float* dxg, float* dh, float* dhpart, float* vu, float* dpc, float* dpu, float* base, long long rofs, int sub, int T, int B, int H, int Bp, int n) {
    EncBwd e;                    // flat kernel arguments, see enc_gates_kernel
    e.a.mask = mask; e.a.y = y; e.a.u = u; e.a.r = r; e.a.c = c; e.a.dy = dy; e.a.dxg = dxg; e.a.sub = sub; e.a.T = T; e.a.B = B; e.a.H = H;
    e.dh = dh; e.dhpart = dhpart; e.vu = vu; e.dpc = dpc; e.dpu = dpu; e.base = base; e.Bp = Bp; e.rofs = rofs;
    const int dir = blockIdx.z;
    const float* WhhT_d = dir ? WhhT1 : WhhT0;
    const float* WhgT_d = dir ? WhgT1 : WhgT0;
    const float* h0_d = dir ? h01 : h00;
    const EncBwd0& a = e.a;
    const int tile = blockIdx.x, b0 = blockIdx.y * 16;
    int t, tp; HPrev hp;
    enc_bwd_geometry(a, h0_d, n, dir, t, tp, hp);
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < a.B && j < H;
    const size_t q = ((size_t)dir * e.Bp + b) * H + j;
    const float part = ok ? e.dhpart[q] + e.vu[q] + enc_dy_at(a, tp, b, dir, j) : 0.f;
    const bool next = ok && tp >= 0 && tp < a.T;
    EncPrep prep;
    prep.uu = prep.cc = prep.hprev = 0.f; prep.m = 1.f;
    if (next) prep = enc_bwd_prefetch(e, h0_d, dir, tp, b, j);        // in flight during the contraction
    const float* dpr = a.dxg + ((size_t)t * a.B + b0) * 6 * H + dir * 3 * H + 2 * H;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm_sel<FAST>(acc0, acc1, row_src(dpr, 6 * H, a.B - b0, H), WhgT_d + e.rofs, H, tile);
    const float v = rb_reduce_once(acc0, acc1);
    if (ok) {
        const float dhp = part + v;
        e.dh[q] = dhp;
        if (next) enc_bwd_prepare(e, prep, dir, tp, b, j, dhp);
    }
}

// d initial_state[dir][j] = sum_b dh[dir][b][j]
__global__ __launch_bounds__(256) void enc_bwd_h0_kernel(EncBwd e) {
    const EncBwd0& a = e.a;
    const int dir = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += e.dh[((size_t)dir * e.Bp + b) * a.H + j];
    a.dh0[dir][j] = s;
}

int lvsr_bigru_fwd_persistent(hipStream_t s, const lvsr_bigru_fwd_args& a, int use_graph);
int lvsr_bigru_bwd_persistent(hipStream_t s, const lvsr_bigru_bwd_args& a, int use_graph);

extern "C" {

int lvsr_bigru_fwd(void* stream, const lvsr_bigru_fwd_args* args, int use_graph) {
    LVSR_REQUIRE(args != nullptr, "lvsr_bigru_fwd: null args");
    EncFwd a = *args;
    const int T = a.T, B = a.B, H = a.H;
    LVSR_REQUIRE(T > 0 && B > 0 && H > 0 && a.sub >= 1, "lvsr_bigru_fwd: bad dims T=%d B=%d H=%d sub=%d", T, B, H, a.sub);
    LVSR_REQUIRE(a.sub == 1 || a.ysub != nullptr, "lvsr_bigru_fwd: subsample>1 needs ysub");
    if (a.persistent) return lvsr_bigru_fwd_persistent((hipStream_t)stream, a, use_graph);
    if (a.sub == 1) a.ysub = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int rt = (B + 15) / 16;
    const int km = a.kernel_mask ? a.kernel_mask : 3;
    // unguarded operand loads: no K padding in either contraction and 16-B aligned rows
    const bool fast = (H % 64 == 0) && ((((size_t)a.y | (size_t)a.rh | (size_t)a.h0[0] | (size_t)a.h0[1]) & 15) == 0);
    auto enqueue = [&]() {
        for (int n = 0; n < T; ++n) {
            if (fast) {
                if (km & 1) hipLaunchKernelGGL(enc_gates_kernel<true>, dim3((2 * H + 15) / 16, rt, 2), dim3(256), 0, s, a.xg, a.mask, a.Whh_p[0], a.Whh_p[1], a.Whg_p[0], a.Whg_p[1], a.h0[0], a.h0[1], a.y, a.ysub, a.u, a.r, a.c, a.rh, a.sub, a.T, a.B, a.H, n);
                if (km & 2) hipLaunchKernelGGL(enc_cand_kernel<true>, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, a.xg, a.mask, a.Whh_p[0], a.Whh_p[1], a.Whg_p[0], a.Whg_p[1], a.h0[0], a.h0[1], a.y, a.ysub, a.u, a.r, a.c, a.rh, a.sub, a.T, a.B, a.H, n);
            } else {
                if (km & 1) hipLaunchKernelGGL(enc_gates_kernel<false>, dim3((2 * H + 15) / 16, rt, 2), dim3(256), 0, s, a.xg, a.mask, a.Whh_p[0], a.Whh_p[1], a.Whg_p[0], a.Whg_p[1], a.h0[0], a.h0[1], a.y, a.ysub, a.u, a.r, a.c, a.rh, a.sub, a.T, a.B, a.H, n);
                if (km & 2) hipLaunchKernelGGL(enc_cand_kernel<false>, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, a.xg, a.mask, a.Whh_p[0], a.Whh_p[1], a.Whg_p[0], a.Whg_p[1], a.h0[0], a.h0[1], a.y, a.ysub, a.u, a.r, a.c, a.rh, a.sub, a.T, a.B, a.H, n);
            }
        }
    };
    GraphKey key("bigru_fwd");
    key.add(&a, sizeof(a));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_fwd");
}

int lvsr_bigru_bwd(void* stream, const lvsr_bigru_bwd_args* args, int use_graph) {
    LVSR_REQUIRE(args != nullptr, "lvsr_bigru_bwd: null args");
    EncBwd e;
    memset(&e, 0, sizeof(e));
    e.a = *args;
    const int T = e.a.T, B = e.a.B, H = e.a.H;
    LVSR_REQUIRE(T > 0 && B > 0 && H > 0 && e.a.sub >= 1, "lvsr_bigru_bwd: bad dims");
    if (e.a.persistent) return lvsr_bigru_bwd_persistent((hipStream_t)stream, e.a, use_graph);
    e.Bp = ((B + 15) / 16) * 16;
    {
        const size_t plane = (size_t)2 * e.Bp * H;                         // workspace: 12*Bp*H floats
        e.dh = e.a.dh_ws; e.dhpart = e.dh + plane; e.vu = e.dh + 2 * plane; e.dpc = e.dh + 3 * plane; e.dpu = e.dh + 4 * plane;
        e.base = e.dh + 5 * plane;
    }
    e.rofs = (long long)((H + 15) / 16) * 16 * 4 * lvsr_pack_kw(H);       // = lvsr_pack_size(H, H)
    hipStream_t s = (hipStream_t)stream;
    const int rt = (B + 15) / 16;
    const int km = e.a.kernel_mask ? e.a.kernel_mask : 3;
    const bool fast = (H % 64 == 0) && ((((size_t)e.a.dxg | (size_t)e.dh) & 15) == 0);
    auto enqueue = [&]() {
        hipLaunchKernelGGL(enc_bwd_init_kernel, dim3((e.Bp * H + 255) / 256, 1, 2), dim3(256), 0, s, e);
        for (int n = 0; n < T; ++n) {
            if (fast) {
                if (km & 1) hipLaunchKernelGGL(enc_bwd_a_kernel<true>, dim3(2 * ((H + 15) / 16), rt, 2), dim3(256), 0, s, e.a.mask, e.a.y, e.a.u, e.a.r, e.a.c, e.a.WhhT_p[0], e.a.WhhT_p[1], e.a.WhgT_p[0], e.a.WhgT_p[1], e.a.h0[0], e.a.h0[1], e.a.dy, e.a.dxg, e.dh, e.dhpart, e.vu, e.dpc, e.dpu, e.base, e.rofs, e.a.sub, e.a.T, e.a.B, e.a.H, e.Bp, n);
                if (km & 2) hipLaunchKernelGGL(enc_bwd_b_kernel<true>, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, e.a.mask, e.a.y, e.a.u, e.a.r, e.a.c, e.a.WhhT_p[0], e.a.WhhT_p[1], e.a.WhgT_p[0], e.a.WhgT_p[1], e.a.h0[0], e.a.h0[1], e.a.dy, e.a.dxg, e.dh, e.dhpart, e.vu, e.dpc, e.dpu, e.base, e.rofs, e.a.sub, e.a.T, e.a.B, e.a.H, e.Bp, n);
            } else {
                if (km & 1) hipLaunchKernelGGL(enc_bwd_a_kernel<false>, dim3(2 * ((H + 15) / 16), rt, 2), dim3(256), 0, s, e.a.mask, e.a.y, e.a.u, e.a.r, e.a.c, e.a.WhhT_p[0], e.a.WhhT_p[1], e.a.WhgT_p[0], e.a.WhgT_p[1], e.a.h0[0], e.a.h0[1], e.a.dy, e.a.dxg, e.dh, e.dhpart, e.vu, e.dpc, e.dpu, e.base, e.rofs, e.a.sub, e.a.T, e.a.B, e.a.H, e.Bp, n);
                if (km & 2) hipLaunchKernelGGL(enc_bwd_b_kernel<false>, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, e.a.mask, e.a.y, e.a.u, e.a.r, e.a.c, e.a.WhhT_p[0], e.a.WhhT_p[1], e.a.WhgT_p[0], e.a.WhgT_p[1], e.a.h0[0], e.a.h0[1], e.a.dy, e.a.dxg, e.dh, e.dhpart, e.vu, e.dpc, e.dpu, e.base, e.rofs, e.a.sub, e.a.T, e.a.B, e.a.H, e.Bp, n);
            }
        }
        hipLaunchKernelGGL(enc_bwd_h0_kernel, dim3((H + 255) / 256, 1, 2), dim3(256), 0, s, e);
    };
    GraphKey key("bigru_bwd");
    key.add(&e, sizeof(e));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_bwd");
}

}  // extern "C"
