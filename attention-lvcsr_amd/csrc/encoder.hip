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
// host never paces the loop.  Inside a kernel each 256-thread work-group owns a 16-column slice of
// the recurrent weights (read from L2, where the 1.5 MiB/layer stays resident because block->XCD
// placement repeats every step) and the batch (<=16 utterances) is the M dimension of
// v_mfma_f32_16x16x4_f32, K split over the 4 waves.
//
// Layouts (all fp32, time-major like the reference):
//   xg   (T,B,6H)  input projections, direction d at column d*3H: [x_in (H) | gate_in_update (H) | gate_in_reset (H)]
//   y    (T,B,2H)  layer output [h_fwd | h_bwd] — also the state sequence used as h_{t-1}
//   u,r,c,rh (T,B,2H)  saved update gate, reset gate, candidate, reset*h_prev (direction d at column d*H)
#include "common.h"
#include "graph_cache.h"

struct EncFwd {
    const float* xg; const float* mask;
    const float* Whh[2]; const float* Whg[2]; const float* h0[2];
    float* y; float* ysub; int sub;
    float* u; float* r; float* c; float* rh;
    int T, B, H;
};

struct HPrev {   // h_{t-1}[b][k] for one direction
    const float* base; int ld;   // ld = 2H (row of y) or 0 (initial state broadcast over the batch)
    int B;
    __device__ __forceinline__ float operator()(int b, int k) const { return b < B ? base[(size_t)b * ld + k] : 0.f; }
};

__device__ __forceinline__ void enc_step_geometry(const EncFwd& a, int n, int dir, int& t, HPrev& hp) {
    t = dir == 0 ? n : a.T - 1 - n;
    const int tp = dir == 0 ? t - 1 : t + 1;
    hp.B = a.B;
    if (n == 0) { hp.base = a.h0[dir]; hp.ld = 0; }
    else { hp.base = a.y + (size_t)tp * a.B * 2 * a.H + dir * a.H; hp.ld = 2 * a.H; }
}

// gates: g = sigmoid(h_prev @ W_hg + g_in[t]); u = g[:, :H], r = g[:, H:]; rh = r * h_prev
__global__ __launch_bounds__(256) void enc_gates_kernel(EncFwd a, int n) {
    const int dir = blockIdx.z, H = a.H, c0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    int t; HPrev hp;
    enc_step_geometry(a, n, dir, t, hp);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = rb_partial(acc, [&](int i, int k) { return hp(b0 + i, k); }, a.Whg[dir], 2 * H, H, c0, 2 * H);
    const float v = rb_reduce(acc);
    const int b = b0 + (threadIdx.x >> 4), j = c0 + (threadIdx.x & 15);
    if (b < a.B && j < 2 * H) {
        const size_t row = (size_t)t * a.B + b;
        const float g = sigmoidf_(v + a.xg[row * 6 * H + dir * 3 * H + H + j]);
        if (j < H) {
            a.u[row * 2 * H + dir * H + j] = g;
        } else {
            const int jj = j - H;
            a.r[row * 2 * H + dir * H + jj] = g;
            a.rh[row * 2 * H + dir * H + jj] = g * hp(b, jj);
        }
    }
}

// candidate + state update + mask blend; writes y[t] (and the subsampled copy)
__global__ __launch_bounds__(256) void enc_cand_kernel(EncFwd a, int n) {
    const int dir = blockIdx.z, H = a.H, c0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    int t; HPrev hp;
    enc_step_geometry(a, n, dir, t, hp);
    const float* rh = a.rh + (size_t)t * a.B * 2 * H + dir * H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = rb_partial(acc, [&](int i, int k) { return (b0 + i) < a.B ? rh[(size_t)(b0 + i) * 2 * H + k] : 0.f; },
                     a.Whh[dir], H, H, c0, H);
    const float v = rb_reduce(acc);
    const int b = b0 + (threadIdx.x >> 4), j = c0 + (threadIdx.x & 15);
    if (b < a.B && j < H) {
        const size_t row = (size_t)t * a.B + b;
        const float cand = tanhf(v + a.xg[row * 6 * H + dir * 3 * H + j]);
        const float uu = a.u[row * 2 * H + dir * H + j];
        const float hprev = hp(b, j);
        float hn = cand * uu + hprev * (1.f - uu);
        if (a.mask) {
            const float m = a.mask[row];
            hn = m * hn + (1.f - m) * hprev;
        }
        a.c[row * 2 * H + dir * H + j] = cand;
        a.y[row * 2 * H + dir * H + j] = hn;
        if (a.ysub && (t % a.sub) == 0) a.ysub[((size_t)(t / a.sub) * a.B + b) * 2 * H + dir * H + j] = hn;
    }
}

// ---------------------------------------------------------------------------------------------
// BPTT.  Per step (fwd direction walks t = T-1..0, bwd direction t = 0..T-1):
//   dhn = m*dh; dc = dhn*u; du = dhn*(c-h_prev); dpre_c = dc*(1-c^2)
//   drh = dpre_c @ Whh^T                                   (kernel A)
//   dr = drh*h_prev; dpre_u = du*u*(1-u); dpre_r = dr*r*(1-r)
//   dh_prev = dhn*(1-u) + (1-m)*dh + drh*r + [dpre_u|dpre_r] @ Whg^T + dy[t_prev]   (kernel B)
//   dxg[t] = [dpre_c | dpre_u | dpre_r]
// ---------------------------------------------------------------------------------------------
struct EncBwd {
    const float* mask; const float* y; const float* u; const float* r; const float* c;
    const float* WhhT[2]; const float* WhgT[2]; const float* h0[2];   // WhhT (H,H) = Whh^T ; WhgT (2H,H) = Whg^T
    const float* dy; int sub;     // gradient wrt the (subsampled) layer output, (ceil(T/sub),B,2H)
    float* dxg;                   // (T,B,6H)
    float* dh;                    // (2,Bp,H) running dL/dh_t   (Bp = B rounded up to 16)
    float* dhpart;                // (2,Bp,H)
    int T, B, H, Bp;
};

__device__ __forceinline__ float enc_dy_at(const EncBwd& a, int t, int b, int dir, int j) {
    if (t < 0 || t >= a.T || (t % a.sub) != 0) return 0.f;
    return a.dy[((size_t)(t / a.sub) * a.B + b) * 2 * a.H + dir * a.H + j];
}

__global__ __launch_bounds__(256) void enc_bwd_init_kernel(EncBwd a) {
    const int dir = blockIdx.z;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.Bp * a.H) return;
    const int b = idx / a.H, j = idx % a.H;
    const int t = dir == 0 ? a.T - 1 : 0;
    a.dh[((size_t)dir * a.Bp + b) * a.H + j] = b < a.B ? enc_dy_at(a, t, b, dir, j) : 0.f;
}

__device__ __forceinline__ void enc_bwd_geometry(const EncBwd& a, int n, int dir, int& t, int& tp, HPrev& hp) {
    t = dir == 0 ? a.T - 1 - n : n;
    tp = dir == 0 ? t - 1 : t + 1;
    hp.B = a.B;
    if (tp < 0 || tp >= a.T) { hp.base = a.h0[dir]; hp.ld = 0; }
    else { hp.base = a.y + (size_t)tp * a.B * 2 * a.H + dir * a.H; hp.ld = 2 * a.H; }
}

__global__ __launch_bounds__(256) void enc_bwd_a_kernel(EncBwd a, int n) {
    const int dir = blockIdx.z, H = a.H, c0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    int t, tp; HPrev hp;
    enc_bwd_geometry(a, n, dir, t, tp, hp);
    const float* dh = a.dh + (size_t)dir * a.Bp * H;
    const size_t trow = (size_t)t * a.B;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = rb_partial(acc, [&](int i, int k) {
        const int b = b0 + i;
        if (b >= a.B) return 0.f;
        const float m = a.mask ? a.mask[trow + b] : 1.f;
        const size_t o = (trow + b) * 2 * H + dir * H + k;
        const float cc = a.c[o];
        return dh[(size_t)b * H + k] * m * a.u[o] * (1.f - cc * cc);
    }, a.WhhT[dir], H, H, c0, H);
    const float drh = rb_reduce(acc);
    const int b = b0 + (threadIdx.x >> 4), j = c0 + (threadIdx.x & 15);
    if (b < a.B && j < H) {
        const float m = a.mask ? a.mask[trow + b] : 1.f;
        const size_t o = (trow + b) * 2 * H + dir * H + j;
        const float uu = a.u[o], rr = a.r[o], cc = a.c[o], hprev = hp(b, j);
        const float dhv = dh[(size_t)b * H + j];
        const float dhn = m * dhv;
        const float dpc = dhn * uu * (1.f - cc * cc);
        const float dpu = dhn * (cc - hprev) * uu * (1.f - uu);
        const float dpr = drh * hprev * rr * (1.f - rr);
        float* dx = a.dxg + (trow + b) * 6 * H + dir * 3 * H;
        dx[j] = dpc; dx[H + j] = dpu; dx[2 * H + j] = dpr;
        a.dhpart[((size_t)dir * a.Bp + b) * H + j] = dhn * (1.f - uu) + (1.f - m) * dhv + drh * rr;
    }
}

__global__ __launch_bounds__(256) void enc_bwd_b_kernel(EncBwd a, int n) {
    const int dir = blockIdx.z, H = a.H, c0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    int t, tp; HPrev hp;
    enc_bwd_geometry(a, n, dir, t, tp, hp);
    const float* dg = a.dxg + (size_t)t * a.B * 6 * H + dir * 3 * H + H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = rb_partial(acc, [&](int i, int k) { return (b0 + i) < a.B ? dg[(size_t)(b0 + i) * 6 * H + k] : 0.f; },
                     a.WhgT[dir], H, 2 * H, c0, H);
    const float v = rb_reduce(acc);
    const int b = b0 + (threadIdx.x >> 4), j = c0 + (threadIdx.x & 15);
    if (b < a.B && j < H) {
        const size_t o = ((size_t)dir * a.Bp + b) * H + j;
        a.dh[o] = a.dhpart[o] + v + enc_dy_at(a, tp, b, dir, j);
    }
}

// d initial_state[dir][j] = sum_b dh[dir][b][j]
__global__ __launch_bounds__(256) void enc_bwd_h0_kernel(EncBwd a, float* dh0_f, float* dh0_b) {
    const int dir = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) s += a.dh[((size_t)dir * a.Bp + b) * a.H + j];
    (dir == 0 ? dh0_f : dh0_b)[j] = s;
}

extern "C" {

int lvsr_bigru_fwd(void* stream, const float* xg, const float* mask, const float* Whh_f, const float* Whg_f,
                   const float* h0_f, const float* Whh_b, const float* Whg_b, const float* h0_b, float* y, float* ysub,
                   int sub, float* u, float* r, float* c, float* rh, int T, int B, int H, int use_graph) {
    LVSR_REQUIRE(T > 0 && B > 0 && H > 0 && sub >= 1, "lvsr_bigru_fwd: bad dims T=%d B=%d H=%d sub=%d", T, B, H, sub);
    EncFwd a;
    memset(&a, 0, sizeof(a));
    a.xg = xg; a.mask = mask; a.Whh[0] = Whh_f; a.Whh[1] = Whh_b; a.Whg[0] = Whg_f; a.Whg[1] = Whg_b;
    a.h0[0] = h0_f; a.h0[1] = h0_b; a.y = y; a.ysub = (sub > 1 ? ysub : nullptr); a.sub = sub;
    a.u = u; a.r = r; a.c = c; a.rh = rh; a.T = T; a.B = B; a.H = H;
    LVSR_REQUIRE(sub == 1 || ysub != nullptr, "lvsr_bigru_fwd: subsample>1 needs ysub");
    hipStream_t s = (hipStream_t)stream;
    const int rt = (B + 15) / 16;
    auto enqueue = [&]() {
        for (int n = 0; n < T; ++n) {
            hipLaunchKernelGGL(enc_gates_kernel, dim3((2 * H + 15) / 16, rt, 2), dim3(256), 0, s, a, n);
            hipLaunchKernelGGL(enc_cand_kernel, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, a, n);
        }
    };
    GraphKey key("bigru_fwd");
    key.add(&a, sizeof(a));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_fwd");
}

int lvsr_bigru_bwd(void* stream, const float* mask, const float* y, const float* u, const float* r, const float* c,
                   const float* WhhT_f, const float* WhgT_f, const float* h0_f, const float* WhhT_b,
                   const float* WhgT_b, const float* h0_b, const float* dy, int sub, float* dxg, float* dh_ws,
                   float* dh0_f, float* dh0_b, int T, int B, int H, int use_graph) {
    LVSR_REQUIRE(T > 0 && B > 0 && H > 0 && sub >= 1, "lvsr_bigru_bwd: bad dims");
    EncBwd a;
    memset(&a, 0, sizeof(a));
    a.mask = mask; a.y = y; a.u = u; a.r = r; a.c = c;
    a.WhhT[0] = WhhT_f; a.WhhT[1] = WhhT_b; a.WhgT[0] = WhgT_f; a.WhgT[1] = WhgT_b; a.h0[0] = h0_f; a.h0[1] = h0_b;
    a.dy = dy; a.sub = sub; a.dxg = dxg; a.T = T; a.B = B; a.H = H; a.Bp = ((B + 15) / 16) * 16;
    a.dh = dh_ws; a.dhpart = dh_ws + (size_t)2 * a.Bp * H;     // workspace: 4*Bp*H floats
    hipStream_t s = (hipStream_t)stream;
    const int rt = (B + 15) / 16;
    auto enqueue = [&]() {
        hipLaunchKernelGGL(enc_bwd_init_kernel, dim3((a.Bp * H + 255) / 256, 1, 2), dim3(256), 0, s, a);
        for (int n = 0; n < T; ++n) {
            hipLaunchKernelGGL(enc_bwd_a_kernel, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, a, n);
            hipLaunchKernelGGL(enc_bwd_b_kernel, dim3((H + 15) / 16, rt, 2), dim3(256), 0, s, a, n);
        }
        hipLaunchKernelGGL(enc_bwd_h0_kernel, dim3((H + 255) / 256, 1, 2), dim3(256), 0, s, a, dh0_f, dh0_b);
    };
    GraphKey key("bigru_bwd");
    key.add(&a, sizeof(a));
    key.add(&dh0_f, sizeof(dh0_f));
    key.add(&dh0_b, sizeof(dh0_b));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_bigru_bwd");
}

}  // extern "C"
