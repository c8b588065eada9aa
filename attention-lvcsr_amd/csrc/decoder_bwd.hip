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

typedef lvsr_attdec_bwd_args AttBwd;

struct DecDpcSrc {    // A operand: ds * ym * u * (1 - c^2); ds rows ld=D, u,c rows ld=D
    const float* ds; const float* u; const float* c; const float* mask;
    int D, nrows; bool vec, fast;
    template <bool FAST>
    __device__ __forceinline__ float4 get(int i, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FAST) i = min(i, nrows - 1);
        else if (i >= nrows || k >= D) return v;
        const float m = mask ? mask[i] : 1.f;
        const size_t o = (size_t)i * D + k;
        const float4 d4 = FAST ? *(const float4*)(ds + o) : ld4g(ds + o, D - k, vec);
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
    const float sp = ok ? a.S[row * D + j] : 0.f;
    const float dsv = ok ? g.ds[(size_t)b * D + j] : 0.f;
    DecDpcSrc src;
    src.ds = g.ds + (size_t)b0 * D; src.u = a.U + ((size_t)i * B + b0) * D; src.c = a.C + ((size_t)i * B + b0) * D;
    src.mask = a.ymask ? a.ymask + (size_t)i * B + b0 : nullptr; src.D = D; src.nrows = B - b0;
    src.vec = ((D & 3) == 0) && ((((size_t)src.ds | (size_t)src.u | (size_t)src.c) & 15) == 0);
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
        if (ok) g.dsacc[(size_t)b * D + j] = part + v;
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

struct DswSrc {      // A operand of B5: dsW[b][m] = sum over position tiles of the per-work-group partials (read only:
                     // a store inside the functor would order every later operand load behind it)
    const float* __restrict__ dswp; int ntile, M, nrows; bool vec, fast;
    template <bool FAST>
    __device__ __forceinline__ float4 get(int i, int k) const {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
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
    __shared__ float row[ATT_MAX_T];
    __shared__ float fl[ATT_MAX_FW];
    const AttDec& a = g.f;
    const int D = a.D, B = a.B, Tp = a.Tp, M = a.M, K = a.K;
    const int rt = (B + 15) / 16, ntD = (D + 15) / 16, nmm = ntD * rt;
    const int ntile = (Tp + ATT_TT - 1) / ATT_TT;
    int blk = blockIdx.x;
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
        src.fast = false;
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
    const int FW = 2 * a.c + 1;
    const float* pp = g.dcvp + ((size_t)b * nslice * K + k) * Tp;
    float* dcv = g.DCV + (((size_t)i * B + b) * K + k) * Tp;
    for (int t = threadIdx.x; t < Tp; t += 256) {
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
        row[t] = s;
        dcv[t] = s;
    }
    for (int x = threadIdx.x; x < FW; x += 256) fl[x] = a.filters[(size_t)k * FW + x];
    __syncthreads();
    float* out = g.dalp + ((size_t)b * K + k) * Tp;
    for (int t = threadIdx.x; t < Tp; t += 256) {
        float s = 0.f;
        if (t >= w.begin && t < w.end) {
            const int dlo = max(-a.c, w.begin - t), dhi = min(a.c, w.end - 1 - t);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int d = dlo;
            for (; d + 3 <= dhi; d += 4) {
                s0 += fl[a.c + d] * row[t + d];
                s1 += fl[a.c + d + 1] * row[t + d + 1];
                s2 += fl[a.c + d + 2] * row[t + d + 2];
                s3 += fl[a.c + d + 3] * row[t + d + 3];
            }
            for (; d <= dhi; ++d) s0 += fl[a.c + d] * row[t + d];
            s = (s0 + s1) + (s2 + s3);
        }
        out[t] = s;
    }
}

// gradient wrt conv1d.filters: df[k][j] = sum_{i,b,t in win_i} dcv_i[b,k,t] * alpha_i[b, t-(j-c)]  (alpha index in win_i).
// Block (chunk of R rows (i,b), filter k): the R rows of dcv and alpha are staged in LDS once, thread j owns lag j and
// walks them; per-chunk partials are folded by lvsr_colsum in a fixed order.
#define FG_LDS 8192
__global__ __launch_bounds__(256) void attdec_filter_grad_kernel(AttDec a, const float* DCV, float* part, int R) {
    __shared__ float X[FG_LDS];
    __shared__ float Wr[FG_LDS];
    __shared__ int wb[16], we[16];
    const int chunk = blockIdx.x, k = blockIdx.y, FW = 2 * a.c + 1, Tp = a.Tp, nrow = a.L * a.B;
    const int r0 = chunk * R, nr = min(R, nrow - r0);
    if (threadIdx.x < nr) {
        const Win w = attdec_window(a, (r0 + threadIdx.x) / a.B);
        wb[threadIdx.x] = w.begin; we[threadIdx.x] = w.end;
    }
    for (int x = threadIdx.x; x < nr * Tp; x += 256) {
        const int r = x / Tp, t = x % Tp, row = r0 + r;          // row = i*B + b
        X[x] = DCV[((size_t)row * a.K + k) * Tp + t];
        Wr[x] = a.W[(size_t)row * Tp + t];                        // alignment slot i = alpha_{i-1}
    }
    __syncthreads();
    for (int j = threadIdx.x; j < FW; j += 256) {
        const int d = j - a.c;
        float s0 = 0.f, s1 = 0.f;
        for (int r = 0; r < nr; ++r) {
            const int lo = max(wb[r], wb[r] + d), hi = min(we[r], we[r] + d);     // t and t-d inside [begin,end)
            const float* xr = X + r * Tp;
            const float* ar = Wr + r * Tp - d;
            int t = lo;
            for (; t + 1 < hi; t += 2) { s0 += xr[t] * ar[t]; s1 += xr[t + 1] * ar[t + 1]; }
            if (t < hi) s0 += xr[t] * ar[t];
        }
        part[((size_t)chunk * a.K + k) * FW + j] = s0 + s1;
    }
}

extern "C" {

int lvsr_attdec_bwd(void* stream, const lvsr_attdec_bwd_args* args, int use_graph) {
    LVSR_REQUIRE(args != nullptr, "lvsr_attdec_bwd: null args");
    AttBwd g;
    memcpy(&g, args, sizeof(g));
    const AttDec& a = g.f;
    if (int rc = attdec_check(a, "lvsr_attdec_bwd")) return rc;
    LVSR_REQUIRE((a.phases & 3) == 3, "lvsr_attdec_bwd: needs a full forward (phases = 3)");
    LVSR_REQUIRE(a.PA_bs == a.M && a.PA_ts == (long long)a.B * a.M, "lvsr_attdec_bwd: contexts must be contiguous (Tp,B,*)");
    hipStream_t s = (hipStream_t)stream;
    const int rt = (a.B + 15) / 16, ntD = (a.D + 15) / 16, ntE = (a.E + 15) / 16;
    const int nchunk = (a.Tp + ATT_TB - 1) / ATT_TB, ntile = (a.Tp + ATT_TT - 1) / ATT_TT, nslice = (a.M + ATT_MS - 1) / ATT_MS;
    auto enqueue = [&]() {
        for (int i = a.L - 1; i >= 0; --i) {
            hipLaunchKernelGGL(attbwd_gru_a_kernel, dim3(ntD, rt), dim3(256), 0, s, g, i);
            hipLaunchKernelGGL(attbwd_gru_b_kernel, dim3(ntE + ntD, rt), dim3(256), 0, s, g, i);
            hipLaunchKernelGGL(attbwd_q_kernel, dim3((a.Tp + 3) / 4, a.B), dim3(256), 0, s, g, i);
            switch (att_kc(a.K)) {
                case 0: hipLaunchKernelGGL(attbwd_energy_kernel<0>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                case 1: hipLaunchKernelGGL(attbwd_energy_kernel<1>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                case 2: hipLaunchKernelGGL(attbwd_energy_kernel<2>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                case 4: hipLaunchKernelGGL(attbwd_energy_kernel<4>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                case 8: hipLaunchKernelGGL(attbwd_energy_kernel<8>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                case 10: hipLaunchKernelGGL(attbwd_energy_kernel<10>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
                default: hipLaunchKernelGGL(attbwd_energy_kernel<16>, dim3(nslice, a.B, ntile), dim3(256), 0, s, g, i); break;
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
    int R = FG_LDS / a.Tp;
    if (R > 16) R = 16;
    if (R < 1) R = 1;
    LVSR_REQUIRE(a.Tp <= FG_LDS, "lvsr_attdec_filter_grad: attended length %d > %d", a.Tp, FG_LDS);
    const int nrow = a.L * a.B, nchunk = (nrow + R - 1) / R, FW = 2 * a.c + 1;
    const long long need = (long long)nchunk * a.K * FW * 4;
    LVSR_REQUIRE(ws && ws_bytes >= need, "lvsr_attdec_filter_grad: workspace of %lld bytes needed", need);
    hipLaunchKernelGGL(attdec_filter_grad_kernel, dim3(nchunk, a.K), dim3(256), 0, (hipStream_t)stream, a, DCV, ws, R);
    if (int rc = lvsr_check_launch("lvsr_attdec_filter_grad")) return rc;
    return lvsr_colsum(stream, ws, nchunk, a.K * FW, a.K * FW, dfilters, 0.f, nullptr, 0);
}

}  // extern "C"
