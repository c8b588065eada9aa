// Attention decoder, backward (K12 for K5-K9): reverse walk over the label steps, five kernels per step,
// captured into one hipGraph like the forward.  Math (per step i, all names as in decoder_fwd.hip):
//   GRU:   dsn = ym*ds; dpc = dsn*u*(1-c^2); dpu = dsn*(c-s)*u*(1-u); drh = dpc @ Whh^T; dpr = drh*s*r*(1-r)
//          dwa = [dpc|dpu|dpr] @ [Wdi|Wdg]^T + dWA_readout
//          ds' = dsn*(1-u) + (1-ym)*ds + drh*r + [dpu|dpr] @ Whg^T + dsW @ Ws^T + dS_readout
//   att:   q[b,t] = dwa[b,:].A[t,b,:] + dalpha[b,t];  de = alpha*(q - sum_t alpha*q)      (masked softmax)
//          dm[t,b,:] = de * w_e * (1 - tanh^2(match));  dPA += dm;  dsW = sum_t dm;  dcv = dm @ handler^T
//          dalpha'[b,t'] = sum_k sum_d f[k,c+d] * dcv[b,k,t'+d]   (correlation, inside the step's window)
// Window positions carry no gradient (disconnected_grad / floor, lvsr/bricks/attention.py:141-147).
#include "decoder.h"
#include <stdlib.h>

typedef lvsr_attdec_bwd_args AttBwd;

struct DecDpcSrc {    // A operand: ds * ym * u * (1 - c^2); ds rows ld=ldd, u,c rows ld=D
    const float* ds; const float* u; const float* c; const float* mask;
    int D, ldd, nrows; bool vec, fast;
    template <bool FAST>
    __device__ __forceinline__ float4 get(int i, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) i = min(i, nrows - 1);
        else if (i >= nrows || k >= D) return v;
        const float m = mask ? mask[i] : 1.f;
        const size_t o = (size_t)i * D + k;
        const size_t od = (size_t)i * ldd + k;
        const float4 d4 = FAST ? *(const float4*)(ds + od) : ld4g(ds + od, D - k, vec);
        const float4 u4 = FAST ? *(const float4*)(u + o) : ld4g(u + o, D - k, vec);
        const float4 c4 = FAST ? *(const float4*)(c + o) : ld4g(c + o, D - k, vec);
        v.x = d4.x * m * u4.x * (1.f - c4.x * c4.x);
        v.y = d4.y * m * u4.y * (1.f - c4.y * c4.y);
        v.z = d4.z * m * u4.z * (1.f - c4.z * c4.z);
        v.w = d4.w * m * u4.w * (1.f - c4.w * c4.w);
        return v;
    }
};

// B1: drh = dpc @ Whh^T; writes DXG[i] and dspart
__global__ __launch_bounds__(256) void attbwd_gru_a_kernel(AttBwd g, int i) {
    const AttDec& a = g.f;
    const int D = a.D, B = a.B, tile = blockIdx.x, b0 = blockIdx.y * 16;
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    const bool ok = b < B && j < D;
    const size_t row = (size_t)i * B + b;
    const float m = (ok && a.ymask) ? a.ymask[row] : 1.f;
    const float uu = ok ? a.U[row * D + j] : 0.f, rr = ok ? a.R[row * D + j] : 0.f, cc = ok ? a.C[row * D + j] : 0.f;
    const int ldd = g.ds_ld ? g.ds_ld : D;
    const float sp = ok ? a.S[row * (a.S_ld ? a.S_ld : D) + j] : 0.f;
    const float dsv = ok ? g.ds[(size_t)b * ldd + j] : 0.f;
    DecDpcSrc src;
    src.ldd = ldd;
    src.ds = g.ds + (size_t)b0 * ldd; src.u = a.U + ((size_t)i * B + b0) * D; src.c = a.C + ((size_t)i * B + b0) * D;
    src.mask = a.ymask ? a.ymask + (size_t)i * B + b0 : nullptr; src.D = D; src.nrows = B - b0;
    src.vec = ((D & 3) == 0) && ((ldd & 3) == 0) && ((((size_t)src.ds | (size_t)src.u | (size_t)src.c) & 15) == 0);
    src.fast = src.vec && src.nrows > 0 && rb_no_kpad(D);
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm(acc0, acc1, src, g.WhhT_p, D, tile);
    const float drh = rb_reduce(acc0, acc1);
    if (ok) {
        const float dsn = m * dsv;
        float* dx = g.DXG + row * 3 * D;
        dx[j] = dsn * uu * (1.f - cc * cc);
        dx[D + j] = dsn * (cc - sp) * uu * (1.f - uu);
        dx[2 * D + j] = drh * sp * rr * (1.f - rr);
        g.dspart[(size_t)b * D + j] = dsn * (1.f - uu) + (1.f - m) * dsv + drh * rr;
    }
}

// B2: dwa = DXG[i] @ [Wdi|Wdg]^T + dWA_r (E tiles);  dsacc = dspart + [dpu|dpr] @ Whg^T (D tiles)
__global__ __launch_bounds__(256) void attbwd_gru_b_kernel(AttBwd g, int i) {
    const AttDec& a = g.f;
    const int D = a.D, B = a.B, E = a.E;
    const int ntE = (E + 15) / 16;
    const int tileAll = blockIdx.x, b0 = blockIdx.y * 16;
    const int b = b0 + (threadIdx.x >> 4);
    const size_t row = (size_t)i * B + b;
    const float* dx = g.DXG + ((size_t)i * B + b0) * 3 * D;
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    if (tileAll < ntE) {
        const int j = tileAll * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < E;
        const float add = (ok && g.dWA_r) ? g.dWA_r[row * E + j] : 0.f;
        rb_mm(acc0, acc1, row_src(dx, 3 * D, B - b0, 3 * D), g.WdT_p, 3 * D, tileAll);
        const float v = rb_reduce(acc0, acc1);
        if (ok) g.DWA[row * E + j] = v + add;
    } else {
        const int tile = tileAll - ntE;
        const int j = tile * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < D;
        const float part = ok ? g.dspart[(size_t)b * D + j] : 0.f;
        rb_mm(acc0, acc1, row_src(dx + D, 3 * D, B - b0, 2 * D), g.WhgT_p, 2 * D, tile);
        const float v = rb_reduce(acc0, acc1);
        if (ok) g.dsacc[(size_t)b * (g.ds_ld ? g.ds_ld : D) + j] = part + v;
    }
}

// B3: q[b,t] = dwa[b,:] . A[t,b,:] + sum_k dalp[b,k,t]   (0 outside the window).  One wave per (b,t), 4 per work-group:
// every lane issues its (up to 4) 16-byte loads of the attended row and of dwa at once; the K rows of the alignment
// gradient ride along in lanes 0..K-1 of the same wave reduction.
__global__ __launch_bounds__(256) void attbwd_q_kernel(AttBwd g, int i) {
    const AttDec& a = g.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x * 4 + wave, B = a.B, Tp = a.Tp, E = a.E;
    if (t >= Tp) return;
    const Win w = attdec_window(a, i);
    float q = 0.f;
    if (t >= w.begin && t < w.end) {
        const float* dwa = g.DWA + ((size_t)i * B + b) * E;
        const float* ar = a.A + (size_t)t * a.A_ts + (size_t)b * a.A_bs;
        const bool vec = ((E & 3) == 0) && ((((size_t)ar | (size_t)dwa) & 15) == 0);
        float4 x[4], y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = lane * 4 + r * 256;
            x[r] = ld4g(ar + e, E - e, vec);
            y[r] = ld4g(dwa + e, E - e, vec);
        }
        const float dal = lane < a.K ? g.dalp[((size_t)b * a.K + lane) * Tp + t] : 0.f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            q0 += x[r].x * y[r].x + x[r].y * y[r].y + x[r].z * y[r].z + x[r].w * y[r].w;
            q1 += x[r + 1].x * y[r + 1].x + x[r + 1].y * y[r + 1].y + x[r + 1].z * y[r + 1].z + x[r + 1].w * y[r + 1].w;
        }
        q = q0 + q1;
        for (int e = lane * 4 + 1024; e < E; e += 256) {          // E > 1024: remaining columns
            const float4 xx = ld4g(ar + e, E - e, vec), yy = ld4g(dwa + e, E - e, vec);
            q += xx.x * yy.x + xx.y * yy.y + xx.z * yy.z + xx.w * yy.w;
        }
        q = wave_sum(q + dal);
    }
    if (lane == 0) g.Q[(size_t)b * Tp + t] = q;
}

// B2+B3 merged (reassociated glimpse): with AW = attended @ [Wdi|Wdg] and QR[i,b,t] = dWA_readout[i,b,:] . A[t,b,:] supplied by the
// caller, q[b,t] = DXG[i][b,:] . AW[t,b,:] + QR[i,b,t] + sum_k dalp[b,k,t] needs no dwa, so the D tiles of B2 and the rows of B3
// are independent and share ONE launch (a kernel boundary less per label); DWA for all labels is one GEMM after the loop.
// Blocks [0, ntD*rt): dsacc = dspart + [dpu|dpr] @ Whg^T; the rest: one wave per (b,t), 4 per work-group.
__global__ __launch_bounds__(256) void attbwd_gru_bq_kernel(AttBwd g, int i) {
    const AttDec& a = g.f;
    const int D = a.D, B = a.B, Tp = a.Tp;
    const int rt = (B + 15) / 16, ntD = (D + 15) / 16, nmm = ntD * rt;
    int blk = blockIdx.x;
    if (blk < nmm) {
        const int tile = blk % ntD, b0 = (blk / ntD) * 16;
        const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < D;
        const float* dx = g.DXG + ((size_t)i * B + b0) * 3 * D;
        const float part = ok ? g.dspart[(size_t)b * D + j] : 0.f;
        f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
        rb_mm(acc0, acc1, row_src(dx + D, 3 * D, B - b0, 2 * D), g.WhgT_p, 2 * D, tile);
        const float v = rb_reduce(acc0, acc1);
        if (ok) g.dsacc[(size_t)b * D + j] = part + v;
        return;
    }
    blk -= nmm;
    const int nchunk = (Tp + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blk / nchunk, t = (blk % nchunk) * 4 + wave, G = 3 * D;
    if (t >= Tp) return;
    const Win w = attdec_window(a, i);
    float q = 0.f;
    if (t >= w.begin && t < w.end) {
        const float* dx = g.DXG + ((size_t)i * B + b) * G;
        const float* ar = g.AW + ((size_t)t * B + b) * (g.AW_ld ? g.AW_ld : G);
        const bool vec = ((G & 3) == 0) && ((((size_t)ar | (size_t)dx) & 15) == 0);
        float4 x[4], y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = lane * 4 + r * 256;
            x[r] = ld4g(ar + e, G - e, vec);
            y[r] = ld4g(dx + e, G - e, vec);
        }
        const float dal = lane < a.K ? g.dalp[((size_t)b * a.K + lane) * Tp + t] : 0.f;
        const float qr = lane == 0 ? g.QR[((size_t)i * B + b) * Tp + t] : 0.f;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            q0 += x[r].x * y[r].x + x[r].y * y[r].y + x[r].z * y[r].z + x[r].w * y[r].w;
            q1 += x[r + 1].x * y[r + 1].x + x[r + 1].y * y[r + 1].y + x[r + 1].z * y[r + 1].z + x[r + 1].w * y[r + 1].w;
        }
        q = q0 + q1;
        for (int e = lane * 4 + 1024; e < G; e += 256) {          // 3D > 1024: remaining columns
            const float4 xx = ld4g(ar + e, G - e, vec), yy = ld4g(dx + e, G - e, vec);
            q += xx.x * yy.x + xx.y * yy.y + xx.z * yy.z + xx.w * yy.w;
        }
        q = wave_sum(q + dal + qr);
    }
    if (lane == 0) g.Q[(size_t)b * Tp + t] = q;
}

// B4: softmax backward + energy backward.  Grid (ceil(M/32), B, ceil(T'/64)), same decomposition as the forward
// energy kernel.  Sums over positions (dsW, handler / energy-vector gradients) leave as per-tile partials, sums
// over the match dimension (dcv) as per-slice partials; both are folded in a fixed order by their consumers.
// KC = compile-time bound of the filter loops: K itself when an instantiation exists (no predication: the element loop is
// VALU-issue bound, measured 9.4 of the kernel's 20 us at K = 10 with 16 predicated iterations), else the next larger one.
template <int KC>
__global__ __launch_bounds__(256) void attbwd_energy_kernel(AttBwd g, int i) {
    __shared__ float cvs[ATT_KMAX][ATT_TT];
    __shared__ float des[ATT_TT];
    __shared__ float dms[ATT_TT][ATT_MS + 1];
    __shared__ float Hs[ATT_KMAX][ATT_MS + 1];
    __shared__ float racc[8][ATT_KMAX + 2][ATT_MS + 1];
    __shared__ float red[4];
    const AttDec& a = g.f;
    const int b = blockIdx.y, slice = blockIdx.x, nslice = gridDim.x, tile = blockIdx.z, ntile = gridDim.z;
    const int B = a.B, Tp = a.Tp, M = a.M, K = a.K, t0 = tile * ATT_TT;
    const Win w = attdec_window(a, i);
    const int ml = threadIdx.x & 31, tg = threadIdx.x >> 5, m = slice * ATT_MS + ml;
    const bool mok = m < M;
    const size_t bt = (size_t)b * ntile + tile;
    if (t0 >= w.end || t0 + ATT_TT <= w.begin) {                   // tile outside the window: zero partial
        if (tg == 0 && mok) g.dswp[bt * M + m] = 0.f;
        return;
    }
    // independent loads first
    float pav[8], dpv[8];
    const float* pab = a.PA + (size_t)b * a.PA_bs + m;
    float* dpab = g.dPA + (size_t)b * M + m;
    const size_t dpa_ts = (size_t)B * M;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int t = t0 + tg + 8 * r;
        const bool ok = t >= w.begin && t < w.end && mok;
        pav[r] = ok ? pab[(size_t)t * a.PA_ts] : 0.f;
        dpv[r] = ok ? dpab[(size_t)t * dpa_ts] : 0.f;
    }
    float Hk[KC > 0 ? KC : 1], hacc[KC > 0 ? KC : 1];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        Hk[k] = (k < K && mok) ? a.handler[(size_t)k * M + m] : 0.f;
        hacc[k] = 0.f;
    }
    const float we_m = mok ? a.w_e[m] : 0.f;
    const float sw_m = mok ? a.sW[((size_t)i * B + b) * M + m] : 0.f;
    const float* al = a.W + ((size_t)(i + 1) * B + b) * Tp;       // alignment produced by step i
    const float* qr = g.Q + (size_t)b * Tp;
    // more independent loads, issued before the first barrier so that their latency hides behind the block reduction: the
    // convolution features of this tile, and the running sums this work-group will add to at the very end
    float cvr[(ATT_KMAX * ATT_TT + 255) / 256];
#pragma unroll
    for (int c = 0; c < (ATT_KMAX * ATT_TT + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int k = x / ATT_TT, tl = x % ATT_TT, t = t0 + tl;
        cvr[c] = (x < K * ATT_TT && t < Tp) ? a.CV[(((size_t)i * B + b) * K + k) * Tp + t] : 0.f;
    }
    float oldacc[((2 + ATT_KMAX) * ATT_MS + 255) / 256];
#pragma unroll
    for (int c = 0; c < ((2 + ATT_KMAX) * ATT_MS + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int v = x / ATT_MS, j = x % ATT_MS, mm = slice * ATT_MS + j;
        float o = 0.f;
        if (x < (2 + K) * ATT_MS && mm < M) {
            if (v == 1) o = g.accWe[bt * M + mm];
            else if (v >= 2) o = g.accH[(bt * K + (v - 2)) * M + mm];
        }
        oldacc[c] = o;
    }
    float sd = 0.f;
    for (int t = w.begin + threadIdx.x; t < w.end; t += 256) sd += al[t] * qr[t];
    sd = block_sum(sd, red);
    if (threadIdx.x < ATT_TT) {
        // alpha = u/Z with u = f(e)*mask:  de = (q - sum alpha q)/Z * mask * f'(e);  softmax: f' = u  =>  alpha*(q - sd)
        const int t = t0 + threadIdx.x;
        float de = 0.f;
        if (t >= w.begin && t < w.end) {
            if (a.normalizer == 0) {
                de = al[t] * (qr[t] - sd);
            } else {
                const float e = a.EN[((size_t)i * B + b) * Tp + t], Z = a.ZB[(size_t)i * B + b];
                const float gq = (qr[t] - sd) / Z * attdec_mask(a, i, b, t);
                if (a.normalizer == 1) { const float sg = sigmoidf_(e); de = gq * sg * (1.f - sg); }
                else de = e > 0.f ? gq / 1000.f : 0.f;
            }
        }
        des[threadIdx.x] = de;
    }
#pragma unroll
    for (int c = 0; c < (ATT_KMAX * ATT_TT + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        if (x < K * ATT_TT) cvs[x / ATT_TT][x % ATT_TT] = cvr[c];
    }
    if (tg == 0) {
#pragma unroll
        for (int k = 0; k < KC; ++k) Hs[k][ml] = Hk[k];
    }
    __syncthreads();
    float swacc = 0.f, weacc = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tl = tg + 8 * r, t = t0 + tl;
        float d = 0.f;
        if (t >= w.begin && t < w.end && mok) {
            float x = pav[r] + sw_m;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (KC == K || k < K) x += cvs[k][tl] * Hk[k];
            const float th = tanh_fast(x);
            const float de = des[tl];
            d = de * we_m * (1.f - th * th);
            dpab[(size_t)t * dpa_ts] = dpv[r] + d;
            swacc += d;
            weacc += de * th;
#pragma unroll
            for (int k = 0; k < KC; ++k)
                if (KC == K || k < K) hacc[k] += cvs[k][tl] * d;
        }
        dms[tl][ml] = d;
    }
    if (a.e_bias && slice == 0 && threadIdx.x == 0) {           // d energy bias = sum of de (des is complete: barrier above)
        float sb = 0.f;
        for (int x = 0; x < ATT_TT; ++x) sb += des[x];
        g.accEb[bt] += sb;
    }
    racc[tg][0][ml] = swacc;
    racc[tg][1][ml] = weacc;
#pragma unroll
    for (int k = 0; k < KC; ++k) racc[tg][2 + k][ml] = hacc[k];
    __syncthreads();
    float* dcvp = K > 0 ? g.dcvp + ((size_t)b * nslice + slice) * K * Tp : nullptr;
    for (int x = threadIdx.x; x < ATT_TT * K; x += 256) {
        const int tl = x / K, k = x % K, t = t0 + tl;
        if (t >= w.begin && t < w.end) {
            float p = 0.f;
#pragma unroll
            for (int j = 0; j < ATT_MS; ++j) p += dms[tl][j] * Hs[k][j];
            dcvp[(size_t)k * Tp + t] = p;
        }
    }
#pragma unroll
    for (int c = 0; c < ((2 + ATT_KMAX) * ATT_MS + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int v = x / ATT_MS, j = x % ATT_MS, mm = slice * ATT_MS + j;
        if (x < (2 + K) * ATT_MS && mm < M) {
            float r = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) r += racc[q][v][j];
            if (v == 0) g.dswp[bt * M + mm] = r;
            else if (v == 1) g.accWe[bt * M + mm] = oldacc[c] + r;
            else g.accH[(bt * K + (v - 2)) * M + mm] = oldacc[c] + r;
        }
    }
}

// B4 on the matrix cores (K > 0): the three handler contractions of the kernel above — the forward recomputation
// match += cv^T handler, dcv = dm handler^T and the handler gradient cv dm — as v_mfma_f32_16x16x4_f32 tiles instead of K
// FMAs per element and phase.  Same grid, same inputs / outputs, same per-tile / per-slice partials.
// Work-group = 64 positions x 32 match columns; wave w owns positions [16w, 16w+16): MFMA output lane (c16 = lane%16, g4 =
// lane/16) holds positions 16w + 4 g4 + r (r < 4) x columns ct*16 + c16 (ct < 2), so the elementwise part (tanh, softmax /
// energy backward, dPA) runs directly on the accumulators.  KP = K rounded up to a multiple of 4 (zero filters beyond K).
template <int KP>
__global__ __launch_bounds__(256) void attbwd_energy_mfma_kernel(AttBwd g, int i) {
    constexpr int NS = KP / 4;
    __shared__ float cvs[KP][ATT_TT + 1];
    __shared__ float des[ATT_TT];
    __shared__ float dms[ATT_TT][ATT_MS + 1];
    __shared__ float racc[4][KP + 2][ATT_MS + 1];
    __shared__ float red[4];
    const AttDec& a = g.f;
    const int b = blockIdx.y, slice = blockIdx.x, nslice = gridDim.x, tile = blockIdx.z, ntile = gridDim.z;
    const int B = a.B, Tp = a.Tp, M = a.M, K = a.K, t0 = tile * ATT_TT;
    const Win w = attdec_window(a, i);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
    const size_t bt = (size_t)b * ntile + tile;
    if (t0 >= w.end || t0 + ATT_TT <= w.begin) {                   // tile outside the window: zero partial
        if (threadIdx.x < ATT_MS && slice * ATT_MS + threadIdx.x < M) g.dswp[bt * M + slice * ATT_MS + threadIdx.x] = 0.f;
        return;
    }
    // ---- independent loads first: this lane's 4 positions x 2 columns of PA and of the running dPA, its operand registers
    float pav[2][4], dpv[2][4], we_m[2], sw_m[2];
    bool mok[2];
    const size_t dpa_ts = (size_t)B * M;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int m = slice * ATT_MS + ct * 16 + c16;
        mok[ct] = m < M;
        const float* pab = a.PA + (size_t)b * a.PA_bs + m;
        const float* dpab = g.dPA + (size_t)b * M + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * wv + 4 * g4 + r;
            const bool ok = t >= w.begin && t < w.end && mok[ct];
            pav[ct][r] = ok ? pab[(size_t)t * a.PA_ts] : 0.f;
            dpv[ct][r] = ok ? dpab[(size_t)t * dpa_ts] : 0.f;
        }
        we_m[ct] = mok[ct] ? a.w_e[m] : 0.f;
        sw_m[ct] = mok[ct] ? a.sW[((size_t)i * B + b) * M + m] : 0.f;
    }
    float Hb[NS][2];        // B operand of the forward contraction: handler[4s + g4][column ct*16 + c16]
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int k = 4 * s + g4;
            Hb[s][ct] = (k < K && mok[ct]) ? a.handler[(size_t)k * M + slice * ATT_MS + ct * 16 + c16] : 0.f;
        }
    float Ht[ATT_MS / 4];   // B operand of dcv = dm handler^T: handler[filter c16][column 4 s + g4]
#pragma unroll
    for (int s = 0; s < ATT_MS / 4; ++s) {
        const int m = slice * ATT_MS + 4 * s + g4;
        Ht[s] = (c16 < K && m < M) ? a.handler[(size_t)c16 * M + m] : 0.f;
    }
    const float* al = a.W + ((size_t)(i + 1) * B + b) * Tp;       // alignment produced by step i
    const float* qr = g.Q + (size_t)b * Tp;
    float cvr[(KP * ATT_TT + 255) / 256];
#pragma unroll
    for (int c = 0; c < (KP * ATT_TT + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int k = x / ATT_TT, tl = x % ATT_TT, t = t0 + tl;
        cvr[c] = (x < K * ATT_TT && t < Tp) ? a.CV[(((size_t)i * B + b) * K + k) * Tp + t] : 0.f;
    }
    float oldacc[((2 + KP) * ATT_MS + 255) / 256];
#pragma unroll
    for (int c = 0; c < ((2 + KP) * ATT_MS + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int v = x / ATT_MS, j = x % ATT_MS, mm = slice * ATT_MS + j;
        float o = 0.f;
        if (x < (2 + K) * ATT_MS && mm < M) {
            if (v == 1) o = g.accWe[bt * M + mm];
            else if (v >= 2) o = g.accH[(bt * K + (v - 2)) * M + mm];
        }
        oldacc[c] = o;
    }
    float sd = 0.f;
    for (int t = w.begin + threadIdx.x; t < w.end; t += 256) sd += al[t] * qr[t];
    sd = block_sum(sd, red);
    if (threadIdx.x < ATT_TT) {
        const int t = t0 + threadIdx.x;
        float de = 0.f;
        if (t >= w.begin && t < w.end) {
            if (a.normalizer == 0) {
                de = al[t] * (qr[t] - sd);
            } else {
                const float e = a.EN[((size_t)i * B + b) * Tp + t], Z = a.ZB[(size_t)i * B + b];
                const float gq = (qr[t] - sd) / Z * attdec_mask(a, i, b, t);
                if (a.normalizer == 1) { const float sg = sigmoidf_(e); de = gq * sg * (1.f - sg); }
                else de = e > 0.f ? gq / 1000.f : 0.f;
            }
        }
        des[threadIdx.x] = de;
    }
#pragma unroll
    for (int c = 0; c < (KP * ATT_TT + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        if (x < KP * ATT_TT) cvs[x / ATT_TT][x % ATT_TT] = cvr[c];
    }
    __syncthreads();
    // ---- match = PA + sW + cv^T handler on this wave's 16 positions x 32 columns
    f32x4 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        acc[ct] = (f32x4){pav[ct][0] + sw_m[ct], pav[ct][1] + sw_m[ct], pav[ct][2] + sw_m[ct], pav[ct][3] + sw_m[ct]};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float av = cvs[4 * s + g4][16 * wv + c16];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Hb[s][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Hb[s][1], acc[1], 0, 0, 0);
    }
    // ---- elementwise: dm = de * w_e * (1 - tanh^2); dPA += dm; partial sums over this lane's 4 positions
    float swacc[2] = {0.f, 0.f}, weacc[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tl = 16 * wv + 4 * g4 + r, t = t0 + tl;
        const bool tin = t >= w.begin && t < w.end;
        const float de = des[tl];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float d = 0.f;
            if (tin && mok[ct]) {
                const float th = tanh_fast(acc[ct][r]);
                d = de * we_m[ct] * (1.f - th * th);
                g.dPA[(size_t)t * dpa_ts + (size_t)b * M + slice * ATT_MS + ct * 16 + c16] = dpv[ct][r] + d;
                swacc[ct] += d;
                weacc[ct] += de * th;
            }
            dms[tl][ct * 16 + c16] = d;
        }
    }
    if (a.e_bias && slice == 0 && threadIdx.x == 0) {           // d energy bias = sum of de (des is complete: barrier above)
        float sb = 0.f;
        for (int x = 0; x < ATT_TT; ++x) sb += des[x];
        g.accEb[bt] += sb;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {                             // fold the four position groups of the wave
        swacc[ct] += __shfl_xor(swacc[ct], 16, 64); swacc[ct] += __shfl_xor(swacc[ct], 32, 64);
        weacc[ct] += __shfl_xor(weacc[ct], 16, 64); weacc[ct] += __shfl_xor(weacc[ct], 32, 64);
        if (g4 == 0) { racc[wv][0][ct * 16 + c16] = swacc[ct]; racc[wv][1][ct * 16 + c16] = weacc[ct]; }
    }
    __syncthreads();
    // ---- dcv[t][k] = sum_m dm[t][m] handler[k][m] for this wave's 16 positions (rows) x 16 filters (columns)
    if (K > 0) {
        f32x4 dc = F32X4_ZERO;
#pragma unroll
        for (int s = 0; s < ATT_MS / 4; ++s)
            dc = __builtin_amdgcn_mfma_f32_16x16x4f32(dms[16 * wv + c16][4 * s + g4], Ht[s], dc, 0, 0, 0);
        float* dcvp = g.dcvp + ((size_t)b * nslice + slice) * K * Tp;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * wv + 4 * g4 + r;
            if (c16 < K && t >= w.begin && t < w.end) dcvp[(size_t)c16 * Tp + t] = dc[r];
        }
    }
    // ---- handler gradient partial: dH[k][m] = sum_t cv[k][t] dm[t][m]; wave w takes positions [16w, 16w+16) of the sum
    {
        f32x4 dh[2] = {F32X4_ZERO, F32X4_ZERO};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tl = 16 * wv + 4 * s + g4;
            const float av = c16 < KP ? cvs[c16 < KP ? c16 : 0][tl] : 0.f;
            dh[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dms[tl][c16], dh[0], 0, 0, 0);
            dh[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, dms[tl][16 + c16], dh[1], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g4 + r;
            if (k < KP) { racc[wv][2 + k][c16] = dh[0][r]; racc[wv][2 + k][16 + c16] = dh[1][r]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < ((2 + KP) * ATT_MS + 255) / 256; ++c) {
        const int x = threadIdx.x + 256 * c;
        const int v = x / ATT_MS, j = x % ATT_MS, mm = slice * ATT_MS + j;
        if (x < (2 + K) * ATT_MS && mm < M) {
            const float r = (racc[0][v][j] + racc[1][v][j]) + (racc[2][v][j] + racc[3][v][j]);
            if (v == 0) g.dswp[bt * M + mm] = r;
            else if (v == 1) g.accWe[bt * M + mm] = oldacc[c] + r;
            else g.accH[(bt * K + (v - 2)) * M + mm] = oldacc[c] + r;
        }
    }
}

struct DswSrc {      // A operand of B5: dsW[b][m] = sum over position tiles of the per-work-group partials (read only:
                     // a store inside the functor would order every later operand load behind it)
    const float* __restrict__ dswp; int ntile, M, nrows; bool vec, fast;
    template <bool FAST>
    __device__ __forceinline__ float4 get(int i, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) {
            // unguarded: rows beyond nrows re-read the last valid one (discarded by the epilogue).  No branch between the
            // loads: the guarded path below serialised its ntile x 8 loads per lane behind their bounds checks — 13 us per
            // label for this product alone (tools/r2p.sh ablation)
            const float4* p = (const float4*)(dswp + (size_t)min(i, nrows - 1) * ntile * M + k);
            const int stride = M / 4;
            float4 x0 = p[0], x1 = make_float4(0.f, 0.f, 0.f, 0.f), x2 = x1, x3 = x1;
            if (ntile > 1) x1 = p[stride];
            if (ntile > 2) x2 = p[2 * stride];
            if (ntile > 3) x3 = p[3 * stride];
            v.x = (x0.x + x1.x) + (x2.x + x3.x); v.y = (x0.y + x1.y) + (x2.y + x3.y);
            v.z = (x0.z + x1.z) + (x2.z + x3.z); v.w = (x0.w + x1.w) + (x2.w + x3.w);
            for (int c = 4; c < ntile; ++c) {
                const float4 x = p[(size_t)c * stride];
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            return v;
        }
        if (i >= nrows || k >= M) return v;
        const float* p = dswp + (size_t)i * ntile * M + k;
#pragma unroll 4
        for (int c = 0; c < ntile; ++c) {
            const float4 x = ld4g(p + (size_t)c * M, M - k, vec);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        return v;
    }
};

// B5: new running gradients.  Blocks [0, nmm): ds = dsacc + dsW @ Ws^T + dS_r[i]; blocks [nmm, nmm+nfold): fold the
// per-tile dsW partials into DSW[i] (kept for the transform_states weight gradient);
// remaining blocks, one per (utterance b, filter k): fold the per-slice dcv partials of row k (stored as DCV[i]
// for the filter gradient) and correlate with filter k inside the window:
//   dalp[b,k,t] = sum_d f[k,c+d] * dcv[k,t+d];   the q kernel of the next (earlier) step adds the K rows up.
__global__ __launch_bounds__(256) void attbwd_post_kernel(AttBwd g, int i) {
    __shared__ float row[ATT_MAX_T + ATT_MAX_FW + 272];
    __shared__ float fl[ATT_MAX_FW + 496];
    __shared__ float cpart[4][16][17];
    const AttDec& a = g.f;
    const int D = a.D, B = a.B, Tp = a.Tp, M = a.M, K = a.K;
    const int rt = (B + 15) / 16, ntD = (D + 15) / 16, nmm = ntD * rt;
    const int ntile = (Tp + ATT_TT - 1) / ATT_TT;
    int blk = blockIdx.x;
    // ablation switch of tools/r2p.sh (LVSR_ATTBWD_POST_ABLATE: 1 skip the state product, 2 the folds, 4 the correlations; wrong results)
    if (blk < nmm) {
        const int tile = blk % ntD, b0 = (blk / ntD) * 16;
        const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
        const bool ok = b < B && j < D;
        float base = ok ? g.dsacc[(size_t)b * D + j] : 0.f;
        if (ok && g.dS_r) base += g.dS_r[((size_t)i * B + b) * D + j];
        DswSrc src;
        src.dswp = g.dswp + (size_t)b0 * ntile * M;
        src.ntile = ntile; src.M = M; src.nrows = B - b0;
        src.vec = ((M & 3) == 0) && ((((size_t)src.dswp) & 15) == 0);
        src.fast = src.vec && src.nrows > 0 && rb_no_kpad(M);
        f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
        rb_mm(acc0, acc1, src, g.WsT_p, M, tile);
        const float v = rb_reduce(acc0, acc1);
        if (ok) g.ds[(size_t)b * D + j] = base + v;
        return;
    }
    blk -= nmm;
    const int nfold = (B * M + 255) / 256;
    if (blk < nfold) {
        const int x = blk * 256 + threadIdx.x;
        if (x < B * M) {
            const int b = x / M, m = x % M;
            float v = 0.f;
            for (int c = 0; c < ntile; ++c) v += g.dswp[((size_t)b * ntile + c) * M + m];
            g.DSW[((size_t)i * B + b) * M + m] = v;
        }
        return;
    }
    blk -= nfold;
    const int k = blk % K, b = blk / K, nslice = (M + ATT_MS - 1) / ATT_MS;
    const Win w = attdec_window(a, i);
    const int FW = 2 * a.c + 1, cn = a.c;
    const float* pp = g.dcvp + ((size_t)b * nslice * K + k) * Tp;
    float* dcv = g.DCV + (((size_t)i * B + b) * K + k) * Tp;
    // the folded dcv row, zero outside the window, with cn zeros in front and cn + 272 behind (the correlation reads them unmasked)
    for (int x = threadIdx.x; x < Tp + 2 * cn + 272; x += 256) {
        const int t = x - cn;
        float s = 0.f;
        if (t >= w.begin && t < w.end) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int sl = 0;
            for (; sl + 3 < nslice; sl += 4) {
                s0 += pp[(size_t)sl * K * Tp + t];
                s1 += pp[(size_t)(sl + 1) * K * Tp + t];
                s2 += pp[(size_t)(sl + 2) * K * Tp + t];
                s3 += pp[(size_t)(sl + 3) * K * Tp + t];
            }
            for (; sl < nslice; ++sl) s0 += pp[(size_t)sl * K * Tp + t];
            s = (s0 + s1) + (s2 + s3);
        }
        row[x] = s;
        if (t >= 0 && t < Tp) dcv[t] = s;
    }
    // filter k with 240 (256) zeros either side: tile column j reads it shifted by 16 j
    for (int x = threadIdx.x; x < FW + 496; x += 256) fl[x] = (x >= 240 && x < 240 + FW) ? a.filters[(size_t)k * FW + (x - 240)] : 0.f;
    __syncthreads();
    // dalp[t] = sum_d f[cn+d] * dcv[t+d] on the matrix cores: a 16 x 16 tile holds 256 consecutive outputs, D[i][j] = out[T + i + 16 j]
    // = sum_u dcv[T + i + u] * f[cn + u - 16 j] over the shifted taps u in [-cn, cn + 240]: A is a Hankel view of the row, B a
    // Toeplitz view of the filter, four taps per v_mfma_f32_16x16x4_f32; the waves split the tap groups and fold through LDS.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
    const int ng = (2 * cn + 241 + 3) / 4;
    float* out = g.dalp + ((size_t)b * K + k) * Tp;
    for (int T0 = 0; T0 < Tp; T0 += 256) {
        f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
        const float* xa = row + T0 + c16 + g4;                  // row[] index of position t is t + cn; tap u = -cn + 4 gq + kk
        const float* fb = fl + 240 + g4 - 16 * c16;             // fl[] index of filter tap f[cn + u] is 240 + cn + u
        int gq = wv;
        for (; gq + 4 < ng; gq += 8) {
            const float a0 = xa[4 * gq], b0 = fb[4 * gq], a1 = xa[4 * gq + 16], b1 = fb[4 * gq + 16];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
        }
        if (gq < ng) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * gq], fb[4 * gq], acc0, 0, 0, 0);
        __syncthreads();                                         // (protects cpart against the previous block's readers)
#pragma unroll
        for (int r = 0; r < 4; ++r) cpart[wv][4 * g4 + r][c16] = acc0[r] + acc1[r];
        __syncthreads();
        {
            const int ii = threadIdx.x & 15, jj = threadIdx.x >> 4, t = T0 + ii + 16 * jj;
            if (t < Tp) {
                const float v = (cpart[0][ii][jj] + cpart[1][ii][jj]) + (cpart[2][ii][jj] + cpart[3][ii][jj]);
                out[t] = (t >= w.begin && t < w.end) ? v : 0.f;
            }
        }
    }
}

// gradient wrt conv1d.filters: df[k][j] = sum_{i,b,t in win_i} dcv_i[b,k,t] * alpha_i[b, t-(j-c)]  (alpha index in win_i).
// A product (filters x positions) . (positions x lags) per row (i,b), on the matrix cores: A = the row's dcv (zero outside the
// window), B = a Toeplitz view of the row's alignment (zero outside the window, c zeros either side in LDS, so no operand is
// masked), four positions per v_mfma_f32_16x16x4_f32.  Block = a chunk of R rows; wave w owns the lag tiles w, w+4, ...
// (NTW of them) for ALL filters, so an A operand is fetched once per four positions and reused by the wave's tiles.  Per-chunk
// partials are folded by lvsr_colsum in a fixed order.  (Was: one thread per lag walking rows and positions with two LDS reads
// per multiply-add, one block per filter — 165 us per call on WSJ-base.)
#define FG_LDS 8192
#define FG_PADF 16
template <int NTW>
__global__ __launch_bounds__(256) void attdec_filter_grad_kernel(AttDec a, const float* DCV, float* part, int R) {
    __shared__ float X[FG_LDS];                    // [R][K][Tp] dcv rows, zero outside the row's window
    __shared__ float Wp[FG_LDS];                   // [R][FG_PADF + Tp + 2c + 4] alignment rows, zero outside the window / in the pads
    __shared__ int wb[16], we[16];
    const int chunk = blockIdx.x, FW = 2 * a.c + 1, Tp = a.Tp, K = a.K, nrow = a.L * a.B;
    const int r0 = chunk * R, nr = min(R, nrow - r0), wlen = FG_PADF + Tp + 2 * a.c + 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c16 = lane & 15, g4 = lane >> 4;
    if (threadIdx.x < nr) {
        const Win w = attdec_window(a, (r0 + threadIdx.x) / a.B);
        wb[threadIdx.x] = w.begin; we[threadIdx.x] = w.end;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < nr * K * Tp; x += 256) {
        const int r = x / (K * Tp), t = x % Tp;
        X[x] = (t >= wb[r] && t < we[r]) ? DCV[(size_t)(r0 + r) * K * Tp + (x - r * K * Tp)] : 0.f;
    }
    for (int x = threadIdx.x; x < nr * wlen; x += 256) {
        const int r = x / wlen, t = x % wlen - FG_PADF - a.c;             // alignment slot i = alpha_{i-1}
        Wp[x] = (t >= wb[r] && t < we[r]) ? a.W[(size_t)(r0 + r) * Tp + t] : 0.f;
    }
    __syncthreads();
    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = F32X4_ZERO;
    const int kc = min(c16, K - 1);
    for (int r = 0; r < nr; ++r) {
        const float* xr = X + ((size_t)r * K + kc) * Tp;
        // B[kk][lag j] = alpha[t - (j - c)], t = 4 s + kk: index FG_PADF + c + t - (j - c) - ... = FG_PADF + 2c + t - j in the padded row
        const float* wr = Wp + (size_t)r * wlen + FG_PADF + 2 * a.c - c16;
        const int s0 = wb[r] / 4, s1 = (we[r] + 3) / 4;                    // positions outside the window contribute zeros
        for (int s = s0; s < s1; ++s) {
            const int t = 4 * s + g4;
            const float av = (c16 < K && t < Tp) ? xr[min(t, Tp - 1)] : 0.f;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int j0 = (wv + 4 * n) * 16;
                if (j0 < FW) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[t - j0], acc[n], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int j = (wv + 4 * n) * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g4 + r;
            if (k < K && j < FW) part[((size_t)chunk * K + k) * FW + j] = acc[n][r];
        }
    }
}

extern "C" {

int lvsr_attdec_bwd(void* stream, const lvsr_attdec_bwd_args* args, int use_graph) {
    LVSR_REQUIRE(args != nullptr, "lvsr_attdec_bwd: null args");
    AttBwd g;
    memcpy(&g, args, sizeof(g));
    const AttDec& a = g.f;
    if (int rc = attdec_check(a, "lvsr_attdec_bwd")) return rc;
    const int parts = (g.parts & 3) ? (g.parts & 3) : 3;
    LVSR_REQUIRE((a.phases & 3) == 3 || (parts == 1 && (a.phases & 2)) || (parts == 2 && (a.phases & 1)),
                 "lvsr_attdec_bwd: the forward block does not cover the requested parts");
    LVSR_REQUIRE(parts == 3 || !(g.AW || g.QR), "lvsr_attdec_bwd: parts and the reassociated glimpse (AW / QR) exclude each other");
    LVSR_REQUIRE(a.label0 >= 0 && a.label0 < a.L, "lvsr_attdec_bwd: label0 outside [0, L)");
    LVSR_REQUIRE(parts == 1 || g.ds_ld == 0 || g.ds_ld == a.D, "lvsr_attdec_bwd: ds_ld is for the GRU part alone (parts = 1)");
    LVSR_REQUIRE(a.S_ld == 0 || a.S_ld >= a.D, "lvsr_attdec_bwd: S_ld < D");
    LVSR_REQUIRE(a.PA_bs == a.M && a.PA_ts == (long long)a.B * a.M, "lvsr_attdec_bwd: contexts must be contiguous (Tp,B,*)");
    LVSR_REQUIRE(a.group_rows == 0, "lvsr_attdec_bwd: row groups are a generation-time layout");
    hipStream_t s = (hipStream_t)stream;
    const int rt = (a.B + 15) / 16, ntD = (a.D + 15) / 16, ntE = (a.E + 15) / 16;
    const int nchunk = (a.Tp + ATT_TB - 1) / ATT_TB, ntile = (a.Tp + ATT_TT - 1) / ATT_TT, nslice = (a.M + ATT_MS - 1) / ATT_MS;
    auto enqueue = [&]() {
        for (int i = a.L - 1; i >= a.label0; --i) {
            if (parts & 1) hipLaunchKernelGGL(attbwd_gru_a_kernel, dim3(ntD, rt), dim3(256), 0, s, g, i);
            if (g.AW && g.QR) {
                hipLaunchKernelGGL(attbwd_gru_bq_kernel, dim3(ntD * rt + ((a.Tp + 3) / 4) * a.B), dim3(256), 0, s, g, i);
            } else {
                if (parts & 1) hipLaunchKernelGGL(attbwd_gru_b_kernel, dim3(ntE + ntD, rt), dim3(256), 0, s, g, i);
                if (parts & 2) hipLaunchKernelGGL(attbwd_q_kernel, dim3((a.Tp + 3) / 4, a.B), dim3(256), 0, s, g, i);
            }
            if (!(parts & 2)) continue;
            const dim3 eg(nslice, a.B, ntile);
            switch ((a.K + 3) / 4) {             // K > 0: handler contractions on the matrix cores, filters padded to a multiple of 4
                case 0: hipLaunchKernelGGL(attbwd_energy_kernel<0>, eg, dim3(256), 0, s, g, i); break;
                case 1: hipLaunchKernelGGL(attbwd_energy_mfma_kernel<4>, eg, dim3(256), 0, s, g, i); break;
                case 2: hipLaunchKernelGGL(attbwd_energy_mfma_kernel<8>, eg, dim3(256), 0, s, g, i); break;
                case 3: hipLaunchKernelGGL(attbwd_energy_mfma_kernel<12>, eg, dim3(256), 0, s, g, i); break;
                default: hipLaunchKernelGGL(attbwd_energy_mfma_kernel<16>, eg, dim3(256), 0, s, g, i); break;
            }
            hipLaunchKernelGGL(attbwd_post_kernel, dim3(ntD * rt + (a.B * a.M + 255) / 256 + a.B * a.K), dim3(256), 0, s, g, i);
        }
    };
    GraphKey key("attdec_bwd");
    key.add(&g, sizeof(g));
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_attdec_bwd");
}

int lvsr_attdec_filter_grad(void* stream, const lvsr_attdec_args* f, const float* DCV, float* dfilters, float* ws,
                            long long ws_bytes) {
    LVSR_REQUIRE(f != nullptr && DCV && dfilters, "lvsr_attdec_filter_grad: null argument");
    AttDec a;
    memcpy(&a, f, sizeof(a));
    if (int rc = attdec_check(a, "lvsr_attdec_filter_grad")) return rc;
    if (a.K == 0) return LVSR_OK;
    const int FW = 2 * a.c + 1, wlen = FG_PADF + a.Tp + 2 * a.c + 4;
    int R = FG_LDS / (a.K * a.Tp);
    if (R > 4) R = 4;
    LVSR_REQUIRE(R >= 1 && wlen <= FG_LDS, "lvsr_attdec_filter_grad: conv_num_filters * attended length %d > %d", a.K * a.Tp, FG_LDS);
    while (R > 1 && R * wlen > FG_LDS) --R;
    const int nrow = a.L * a.B, nchunk = (nrow + R - 1) / R, ntile = (FW + 15) / 16, ntw = (ntile + 3) / 4;
    const long long need = (long long)nchunk * a.K * FW * 4;
    LVSR_REQUIRE(ws && ws_bytes >= need, "lvsr_attdec_filter_grad: workspace of %lld bytes needed", need);
    LVSR_REQUIRE(ntw <= 16, "lvsr_attdec_filter_grad: conv filter too wide");
    hipStream_t s = (hipStream_t)stream;
    if (ntw <= 1) hipLaunchKernelGGL(attdec_filter_grad_kernel<1>, dim3(nchunk), dim3(256), 0, s, a, DCV, ws, R);
    else if (ntw <= 2) hipLaunchKernelGGL(attdec_filter_grad_kernel<2>, dim3(nchunk), dim3(256), 0, s, a, DCV, ws, R);
    else if (ntw <= 4) hipLaunchKernelGGL(attdec_filter_grad_kernel<4>, dim3(nchunk), dim3(256), 0, s, a, DCV, ws, R);
    else if (ntw <= 8) hipLaunchKernelGGL(attdec_filter_grad_kernel<8>, dim3(nchunk), dim3(256), 0, s, a, DCV, ws, R);
    else hipLaunchKernelGGL(attdec_filter_grad_kernel<16>, dim3(nchunk), dim3(256), 0, s, a, DCV, ws, R);
    if (int rc = lvsr_check_launch("lvsr_attdec_filter_grad")) return rc;
    return lvsr_colsum(stream, ws, nchunk, a.K * FW, a.K * FW, dfilters, 0.f, nullptr, 0);
}

}  // extern "C"
