// Small non-recurrent kernels around the decoder: feedback lookup (K11), post-merge activation and the
// softmax emitter cost (K10 epilogues), their gradients, and the deterministic row scatter-add used for
// lookup-table / one-hot fork gradients.
//
// Reference semantics: LookupTable.apply (libs/blocks/blocks/bricks/lookup.py:48-68), OneOfNFeedback
// (lvsr/bricks/__init__.py:97-104: one-hot @ W == row gather), Maxout (libs/blocks/blocks/bricks/simple.py:
// 161-181), Rectifier/Tanh (:184-207), Softmax.log_probabilities / categorical_cross_entropy (:315-371),
// SoftmaxEmitter.cost / costs (libs/blocks/blocks/bricks/sequence_generators.py:780-791).
#include "common.h"
#include "lvsr_hip.h"

// out[r, :] = table[idx[r], :] (+ bias)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* table, int ldt, const long long* idx, int n,
                                                          int nrows, int width, const float* bias, float* out, int ldo) {
    const int r = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (r >= n || c >= width) return;
    long long v = idx[r];
    float x = 0.f;
    if (v >= 0 && v < nrows) x = table[(size_t)v * ldt + c];
    if (bias) x += bias[c];
    out[(size_t)r * ldo + c] = x;
}

// dst[v, :] = beta*dst[v, :] + sum_{r : idx[r] == v} src[r, :]   (fixed r order: deterministic).  The index list is staged in
// LDS a chunk at a time and compacted to the rows that hit v, so a thread walks ~n/V matching rows instead of n dependent
// global index loads (98 -> ~10 us for the 1 600 labels of a WSJ-base minibatch).
#define SCATTER_CHUNK 2048
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* src, int lds, const long long* idx, int n,
                                                               int width, float* dst, int ldd, float beta) {
    __shared__ int hit[SCATTER_CHUNK];
    __shared__ int nhit;
    const int v = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int r0 = 0; r0 < n; r0 += SCATTER_CHUNK) {
        const int m = min(SCATTER_CHUNK, n - r0);
        __syncthreads();
        if (threadIdx.x == 0) nhit = 0;
        // matching rows of this chunk in ascending order: thread t scans a contiguous slice, slices are concatenated in order
        const int per = (m + 255) / 256, x0 = min(m, (int)threadIdx.x * per), x1 = min(m, x0 + per);
        int mine = 0;
        for (int x = x0; x < x1; ++x) mine += idx[r0 + x] == v;
        __shared__ int cnt[256];
        cnt[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int t = 0; t < 256; ++t) { const int k = cnt[t]; cnt[t] = run; run += k; }
            nhit = run;
        }
        __syncthreads();
        int o = cnt[threadIdx.x];
        for (int x = x0; x < x1; ++x)
            if (idx[r0 + x] == v) hit[o++] = r0 + x;
        __syncthreads();
        if (c < width)
            for (int h = 0; h < nhit; ++h) s += src[(size_t)hit[h] * lds + c];
    }
    if (c < width) {
        float* d = dst + (size_t)v * ldd + c;
        *d = (beta != 0.f ? beta * *d : 0.f) + s;
    }
}

// kind: 0 identity, 1 maxout(2 pieces, adjacent pairs), 2 rectifier, 3 tanh
__global__ __launch_bounds__(256) void act_fwd_kernel(int kind, const float* x, int ldx, int n, int P, float* y, int ldy) {
    const int Pout = kind == 1 ? P / 2 : P;
    const size_t total = (size_t)n * Pout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / Pout), j = (int)(idx % Pout);
        const float* xr = x + (size_t)r * ldx;
        float v;
        if (kind == 1) v = fmaxf(xr[2 * j], xr[2 * j + 1]);
        else if (kind == 2) v = xr[j] > 0.f ? xr[j] : 0.f;
        else if (kind == 3) v = tanhf(xr[j]);
        else v = xr[j];
        y[(size_t)r * ldy + j] = v;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(int kind, const float* x, int ldx, const float* dy, int lddy, int n,
                                                      int P, float* dx, int lddx) {
    const int Pout = kind == 1 ? P / 2 : P;
    const size_t total = (size_t)n * Pout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / Pout), j = (int)(idx % Pout);
        const float* xr = x + (size_t)r * ldx;
        float* dr = dx + (size_t)r * lddx;
        const float g = dy[(size_t)r * lddy + j];
        if (kind == 1) {
            // Theano's max gradient goes to every element equal to the maximum
            const float a = xr[2 * j], b = xr[2 * j + 1], m = fmaxf(a, b);
            dr[2 * j] = a == m ? g : 0.f;
            dr[2 * j + 1] = b == m ? g : 0.f;
        } else if (kind == 2) {
            dr[j] = xr[j] > 0.f ? g : 0.f;
        } else if (kind == 3) {
            const float t = tanhf(xr[j]);
            dr[j] = g * (1.f - t * t);
        } else {
            dr[j] = g;
        }
    }
}

// One wave per row: logp = x - max - log(sum exp(x - max)); cost = -logp[label]*mask;
// optional dlogits = (softmax - onehot)*mask*scale; optional neglogp (all classes, beam search).
__global__ __launch_bounds__(256) void softmax_nll_kernel(const float* logits, int ld, const long long* labels,
                                                          const float* mask, int n, int V, float* cost, float* dlogits,
                                                          int ldd, float scale, float* neglogp, int ldn) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const float* x = logits + (size_t)r * ld;
    float mx = -3.0e38f;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, x[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(x[v] - mx);
    s = wave_sum(s);
    const float lse = logf(s);
    const float m = mask ? mask[r] : 1.f;
    const long long y = labels ? labels[r] : -1;
    if (cost && lane == 0) cost[r] = (y >= 0 && y < V) ? -((x[y] - mx) - lse) * m : 0.f;
    for (int v = lane; v < V; v += 64) {
        const float lp = (x[v] - mx) - lse;
        if (neglogp) neglogp[(size_t)r * ldn + v] = -lp;
        if (dlogits) dlogits[(size_t)r * ldd + v] = (expf(lp) - (v == y ? 1.f : 0.f)) * m * scale;
    }
}

// ShallowFusionReadout.readout (lvsr/bricks/language_models.py:92-104); one wave per hypothesis:
//   x = [logsoftmax](am_beta * am) + lm_weight * [logsoftmax](-lm_add)  [-> logsoftmax]
__device__ __forceinline__ float row_lse(const float* x, int V, float scale, int lane) {
    float mx = -3.0e38f;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, scale * x[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(scale * x[v] - mx);
    s = wave_sum(s);
    return mx + logf(s);
}
__global__ __launch_bounds__(256) void shallow_fusion_kernel(const float* am, int ld, const float* lm_add, int n, int V,
                                                             float am_beta, float lm_weight, int norm_am, int norm_lm,
                                                             int norm_tot, float out_scale, float* out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const float* a = am + (size_t)r * ld;
    const float* l = lm_add + (size_t)r * V;
    float* o = out + (size_t)r * V;
    const float lse_a = norm_am ? row_lse(a, V, am_beta, lane) : 0.f;
    const float lse_l = norm_lm ? row_lse(l, V, -1.f, lane) : 0.f;
    float lse_t = 0.f;
    if (norm_tot) {
        float mx = -3.0e38f;
        for (int v = lane; v < V; v += 64) mx = fmaxf(mx, (am_beta * a[v] - lse_a) + lm_weight * (-l[v] - lse_l));
        mx = wave_max(mx);
        float s = 0.f;
        for (int v = lane; v < V; v += 64) s += expf(((am_beta * a[v] - lse_a) + lm_weight * (-l[v] - lse_l)) - mx);
        s = wave_sum(s);
        lse_t = mx + logf(s);
    }
    for (int v = lane; v < V; v += 64) o[v] = out_scale * (((am_beta * a[v] - lse_a) + lm_weight * (-l[v] - lse_l)) - lse_t);
}

// LMEmitter.cost (lvsr/bricks/language_models.py:165-168): cost[r] = -readout[r, label[r]] (* mask[r]); rows whose label is outside
// [0, V) cost 0
__global__ __launch_bounds__(256) void select_cost_kernel(const float* x, int ld, const long long* labels, const float* mask, int n, int V,
                                                          float scale, float* cost) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const long long y = labels[r];
    const float v = (y >= 0 && y < V) ? x[(size_t)r * ld + y] : 0.f;
    cost[r] = scale * v * (mask ? mask[r] : 1.f);
}

// ---------------------------------------------------------------------------------------------------------------
// Generation-time readout + emitter of a few rows (beam hypotheses / sampled utterances) in ONE launch:
//   r1 = bias1 + s @ W_ms + wa @ W_mw  ->  activation  ->  logits = r2 @ W_out + b_out  ->  step costs
// (Readout.readout, libs/blocks/blocks/bricks/sequence_generators.py:614-619 + post-merge, lvsr/bricks/recognizer.py:
// 298-320; SoftmaxEmitter.costs :788-791, or ShallowFusionReadout + LMEmitter, lvsr/bricks/language_models.py:92-184;
// SoftmaxEmitter.emit :770-776 by inverse CDF on a given uniform).  One work-group per row; the three small GEMMs this
// replaces cost 23 us EACH at 16 rows, more than the rest of the beam step together.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float row_lse(const float* x, int V, float scale, int lane);
#define RS_MAX_IN 4096
#define RS_MAX_P 2048
#define RS_MAX_V 2048

// out[j] = bias[j] + sum_k x[k] * W[k*N + j] for j < N; x, out in LDS.  The 256 threads are CP column groups x G slices of K;
// with N % 4 == 0 a thread owns FOUR adjacent columns (one 16-byte load per k) and 8 loads are in flight per thread, which is
// what hides the L2 latency of this weight stream (33 -> 8 us for the 768 x 256 merge at 16 rows).
__device__ void wg_matvec(const float* x, int K, const float* W, int N, const float* bias, float* out, float* part /*[4][256]*/) {
    const bool vec = (N % 4 == 0) && ((((size_t)W) & 15) == 0);
    const int cols = vec ? N / 4 : N;
    int CP = 1;
    while (CP < cols && CP < 256) CP <<= 1;
    const int G = 256 / CP, jl = threadIdx.x % CP, g = threadIdx.x / CP;
    const int per = (K + G - 1) / G, k0 = g * per, k1 = min(K, k0 + per);
    for (int j0 = 0; j0 < cols; j0 += CP) {
        const int j = j0 + jl;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (j < cols) {
            if (vec) {
                const float4* w = (const float4*)W + j;
                const int N4 = N / 4;
                int k = k0;
                for (; k + 7 < k1; k += 8) {
                    float4 wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k + u) * N4];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float xv = x[k + u];
                        a0 += xv * wv[u].x; a1 += xv * wv[u].y; a2 += xv * wv[u].z; a3 += xv * wv[u].w;
                    }
                }
                for (; k < k1; ++k) {
                    const float4 wv = w[(size_t)k * N4];
                    const float xv = x[k];
                    a0 += xv * wv.x; a1 += xv * wv.y; a2 += xv * wv.z; a3 += xv * wv.w;
                }
            } else {
                const float* w = W + j;
                int k = k0;
                for (; k + 3 < k1; k += 4) {
                    a0 += x[k] * w[(size_t)k * N];
                    a1 += x[k + 1] * w[(size_t)(k + 1) * N];
                    a2 += x[k + 2] * w[(size_t)(k + 2) * N];
                    a3 += x[k + 3] * w[(size_t)(k + 3) * N];
                }
                for (; k < k1; ++k) a0 += x[k] * w[(size_t)k * N];
            }
        }
        __syncthreads();
        if (vec) {
            part[threadIdx.x] = a0; part[256 + threadIdx.x] = a1; part[512 + threadIdx.x] = a2; part[768 + threadIdx.x] = a3;
        } else {
            part[threadIdx.x] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        if (vec) {
            // CP * 4 outputs of this pass, folded by the first CP * 4 threads (or in several rounds)
            for (int o = threadIdx.x; o < CP * 4; o += 256) {
                const int q = o >> 2, e = o & 3, jj = (j0 + q) * 4 + e;
                if (j0 + q < cols) {
                    float v = bias ? bias[jj] : 0.f;
                    for (int gg = 0; gg < G; ++gg) v += part[e * 256 + gg * CP + q];
                    out[jj] = v;
                }
            }
        } else if (g == 0 && j < cols) {
            float v = bias ? bias[j] : 0.f;
            for (int gg = 0; gg < G; ++gg) v += part[gg * CP + jl];
            out[j] = v;
        }
    }
    __syncthreads();
}

// lvsr_readout_merge: tile (16 rows x 16 columns) per work-group, the contraction split over the four waves (rb_mm)
__global__ __launch_bounds__(256) void readout_merge_kernel(const float* S, int lds, const float* WA, int ldwa, int n, int D, int E, int P,
                                                            const float* Wms_p, const float* Wmw_p, const float* bias1, float* R1, int ldr1) {
    const int tile = blockIdx.x, b0 = blockIdx.y * 16;
    const int b = b0 + (threadIdx.x >> 4), j = tile * 16 + (threadIdx.x & 15);
    f32x4 acc0 = F32X4_ZERO, acc1 = F32X4_ZERO;
    rb_mm(acc0, acc1, row_src(WA + (size_t)b0 * ldwa, ldwa, n - b0, E), Wmw_p, E, tile);
    if (Wms_p) rb_mm(acc0, acc1, row_src(S + (size_t)b0 * lds, lds, n - b0, D), Wms_p, D, tile);
    const float v = rb_reduce_once(acc0, acc1);
    if (b < n && j < P) R1[(size_t)b * ldr1 + j] = v + (bias1 ? bias1[j] : 0.f);
}

__global__ __launch_bounds__(256) void readout_step_kernel(lvsr_readout_step_args a) {
    __shared__ float xin[RS_MAX_IN];
    __shared__ float r1[RS_MAX_P];
    __shared__ float lg[RS_MAX_V];
    __shared__ float part[4 * 256];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int nin = (a.Wms ? a.D : 0) + a.E;
    // inputs: [s | wa] against the stacked weight [W_ms ; W_mw] — two passes over the same output instead of a stacked copy
    if (a.R1) {                               // merged by lvsr_readout_merge
        for (int j = tid; j < a.P; j += 256) r1[j] = a.R1[(size_t)r * a.ldr1 + j];
        __syncthreads();
    } else {
    for (int k = tid; k < a.E; k += 256) xin[k] = a.WA[(size_t)r * a.ldwa + k];
    for (int k = tid; k < (a.Wms ? a.D : 0); k += 256) xin[a.E + k] = a.S[(size_t)r * a.lds + k];
    __syncthreads();
    wg_matvec(xin, a.E, a.Wmw, a.P, a.bias1, r1, part);
    if (a.Wms) {
        float* tmp = lg;                      // P <= RS_MAX_V is checked by the host when W_ms is given
        wg_matvec(xin + a.E, a.D, a.Wms, a.P, nullptr, tmp, part);
        for (int j = tid; j < a.P; j += 256) r1[j] += tmp[j];
        __syncthreads();
    }
    }
    (void)nin;
    const float* logits = r1;
    if (a.Wout) {
        // activation in place (maxout halves the width)
        const int Pout = a.act == 1 ? a.P / 2 : a.P;
        float v[RS_MAX_P / 256];
        for (int j = tid, q = 0; j < Pout; j += 256, ++q) {
            if (a.act == 1) v[q] = fmaxf(r1[2 * j], r1[2 * j + 1]);
            else if (a.act == 2) v[q] = r1[j] > 0.f ? r1[j] : 0.f;
            else if (a.act == 3) v[q] = tanhf(r1[j]);
            else v[q] = r1[j];
        }
        __syncthreads();
        for (int j = tid, q = 0; j < Pout; j += 256, ++q) r1[j] = v[q];
        __syncthreads();
        int cur = Pout;
        for (int h = 0; h < a.n_hidden; ++h) {            // further post-merge layers: Linear + one-piece activation, via `lg`
            const int w = a.dimh[h];
            wg_matvec(r1, cur, a.Wh[h], w, a.bh[h], lg, part);
            for (int j = tid; j < w; j += 256) r1[j] = a.act == 2 ? fmaxf(lg[j], 0.f) : a.act == 3 ? tanhf(lg[j]) : lg[j];
            __syncthreads();
            cur = w;
        }
        wg_matvec(r1, cur, a.Wout, a.V, a.bout, lg, part);
        logits = lg;
    }
    if (a.logits) for (int v = tid; v < a.V; v += 256) a.logits[(size_t)r * a.V + v] = logits[v];
    // ---- emitter: wave 0 (the others only keep the barriers company)
    const int lane = tid & 63, V = a.V;
    const bool w0 = tid < 64;
    const float* l = a.lm_add ? a.lm_add + (size_t)r * V : nullptr;
    if (w0) {
        float lse_a = 0.f, lse_l = 0.f, lse_t = 0.f, mx0 = 0.f;
        if (!l) {
            // SoftmaxEmitter: the same arithmetic as softmax_nll_kernel ((x - max) - log(sum))
            mx0 = -3.0e38f;
            for (int v = lane; v < V; v += 64) mx0 = fmaxf(mx0, logits[v]);
            mx0 = wave_max(mx0);
            float s0 = 0.f;
            for (int v = lane; v < V; v += 64) s0 += expf(logits[v] - mx0);
            lse_a = logf(wave_sum(s0));
        } else {
            // ShallowFusionReadout: the same arithmetic as shallow_fusion_kernel
            lse_a = a.norm_am ? row_lse(logits, V, a.am_beta, lane) : 0.f;
            lse_l = a.norm_lm ? row_lse(l, V, -1.f, lane) : 0.f;
            if (a.norm_tot) {
                float mx = -3.0e38f;
                for (int v = lane; v < V; v += 64) mx = fmaxf(mx, (a.am_beta * logits[v] - lse_a) + a.lm_weight * (-l[v] - lse_l));
                mx = wave_max(mx);
                float s = 0.f;
                for (int v = lane; v < V; v += 64) s += expf(((a.am_beta * logits[v] - lse_a) + a.lm_weight * (-l[v] - lse_l)) - mx);
                s = wave_sum(s);
                lse_t = mx + logf(s);
            }
        }
        // log-probabilities of the step in xin (free again), costs = their negation
        for (int v = lane; v < V; v += 64) {
            const float lp = l ? ((a.am_beta * logits[v] - lse_a) + a.lm_weight * (-l[v] - lse_l)) - lse_t : (logits[v] - mx0) - lse_a;
            xin[v] = lp;
            if (a.neglogp) a.neglogp[(size_t)r * V + v] = -lp;
        }
    }
    __syncthreads();
    if (a.uniforms && tid == 0) {
        // SoftmaxEmitter.emit: multinomial by inverse CDF over the class order (MultinomialFromUniform: the first class whose
        // running float32 sum of probabilities exceeds the uniform; none -> class 0 as argmax of an all-zero row)
        const float u = a.uniforms[r];
        float cum = 0.f;
        int pick = 0;
        bool hit = false;
        for (int v = 0; v < V; ++v) {
            cum += expf(xin[v]);
            if (!hit && cum > u) { pick = v; hit = true; }
        }
        a.outputs[r] = pick;
        if (a.costs) a.costs[r] = -xin[pick];
    }
}

// SoftmaxEmitter.emit alone (sequence_generators.py:770-776): class by inverse CDF of softmax(logits[r]) at uniforms[r]
// (MultinomialFromUniform: first class whose running float32 sum exceeds the uniform), and its cost; one wave per row.
__global__ __launch_bounds__(256) void softmax_emit_kernel(const float* logits, int ld, const float* uniforms, int n, int V,
                                                           long long* outputs, float* costs) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    const float* x = logits + (size_t)r * ld;
    float mx = -3.0e38f;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, x[v]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(x[v] - mx);
    const float lse = logf(wave_sum(s));
    if (lane == 0) {
        const float u = uniforms[r];
        float cum = 0.f;
        int pick = 0;
        bool hit = false;
        for (int v = 0; v < V; ++v) {
            cum += expf((x[v] - mx) - lse);
            if (!hit && cum > u) { pick = v; hit = true; }
        }
        outputs[r] = pick;
        if (costs) costs[r] = -((x[pick] - mx) - lse);
    }
}

extern "C" {

int lvsr_softmax_emit(void* stream, const float* logits, int ld, const float* uniforms, int n, int V, long long* outputs,
                      float* costs) {
    LVSR_REQUIRE(logits && uniforms && outputs && V > 0, "lvsr_softmax_emit: bad arguments");
    if (n <= 0) return LVSR_OK;
    hipLaunchKernelGGL(softmax_emit_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, uniforms, n, V, outputs,
                       costs);
    return lvsr_check_launch("lvsr_softmax_emit");
}

int lvsr_readout_step(void* stream, const lvsr_readout_step_args* args) {
    LVSR_REQUIRE(args != nullptr, "lvsr_readout_step: null args");
    const lvsr_readout_step_args& a = *args;
    LVSR_REQUIRE(a.n > 0 && a.E > 0 && a.P > 0 && a.V > 0 && a.WA && a.Wmw, "lvsr_readout_step: bad arguments");
    LVSR_REQUIRE(a.D + a.E <= RS_MAX_IN && a.P <= RS_MAX_P && a.V <= RS_MAX_V && (!a.Wms || a.P <= RS_MAX_V),
                 "lvsr_readout_step: sizes exceed the LDS budget (D+E <= %d, P <= %d, V <= %d)", RS_MAX_IN, RS_MAX_P, RS_MAX_V);
    LVSR_REQUIRE(a.Wout != nullptr || a.P == a.V, "lvsr_readout_step: without a post-merge layer the merge width must be V");
    LVSR_REQUIRE(a.act >= 0 && a.act <= 3 && (a.act != 1 || a.P % 2 == 0), "lvsr_readout_step: bad activation");
    LVSR_REQUIRE(!a.uniforms || a.outputs, "lvsr_readout_step: emit needs an output buffer");
    LVSR_REQUIRE(a.n_hidden >= 0 && a.n_hidden <= 3 && (a.n_hidden == 0 || (a.Wout && (a.act == 2 || a.act == 3))),
                 "lvsr_readout_step: further post-merge layers need a one-piece activation and an output layer");
    for (int h = 0; h < a.n_hidden; ++h)
        LVSR_REQUIRE(a.Wh[h] && a.bh[h] && a.dimh[h] > 0 && a.dimh[h] <= RS_MAX_P && a.dimh[h] <= RS_MAX_V,
                     "lvsr_readout_step: bad post-merge layer %d", h);
    hipLaunchKernelGGL(readout_step_kernel, dim3(a.n), dim3(256), 0, (hipStream_t)stream, a);
    return lvsr_check_launch("lvsr_readout_step");
}

int lvsr_readout_merge(void* stream, const float* S, int lds, const float* WA, int ldwa, int n, int D, int E, int P,
                       const float* Wms_p, const float* Wmw_p, const float* bias1, float* R1, int ldr1) {
    LVSR_REQUIRE(WA && Wmw_p && R1 && n > 0 && E > 0 && P > 0 && ldwa >= E && ldr1 >= P && (!Wms_p || (S && D > 0 && lds >= D)),
                 "lvsr_readout_merge: bad arguments");
    hipLaunchKernelGGL(readout_merge_kernel, dim3((P + 15) / 16, (n + 15) / 16), dim3(256), 0, (hipStream_t)stream, S, lds, WA, ldwa, n,
                       D, E, P, Wms_p, Wmw_p, bias1, R1, ldr1);
    return lvsr_check_launch("lvsr_readout_merge");
}

int lvsr_gather_rows(void* stream, const float* table, int ldt, const long long* idx, int n, int nrows, int width,
                     const float* bias, float* out, int ldo) {
    if (n <= 0 || width <= 0) return LVSR_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((width + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, table, ldt, idx, n,
                       nrows, width, bias, out, ldo);
    return lvsr_check_launch("lvsr_gather_rows");
}

int lvsr_scatter_add_rows(void* stream, const float* src, int lds, const long long* idx, int n, int nrows, int width,
                          float* dst, int ldd, float beta) {
    if (nrows <= 0 || width <= 0) return LVSR_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((width + 255) / 256, nrows), dim3(256), 0, (hipStream_t)stream, src, lds,
                       idx, n, width, dst, ldd, beta);
    return lvsr_check_launch("lvsr_scatter_add_rows");
}

int lvsr_act_fwd(void* stream, int kind, const float* x, int ldx, int n, int P, float* y, int ldy) {
    LVSR_REQUIRE(kind >= 0 && kind <= 3 && (kind != 1 || P % 2 == 0), "lvsr_act_fwd: bad activation %d / P=%d", kind, P);
    if (n <= 0 || P <= 0) return LVSR_OK;
    size_t total = (size_t)n * P;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, kind, x, ldx, n, P, y, ldy);
    return lvsr_check_launch("lvsr_act_fwd");
}

int lvsr_act_bwd(void* stream, int kind, const float* x, int ldx, const float* dy, int lddy, int n, int P, float* dx,
                 int lddx) {
    LVSR_REQUIRE(kind >= 0 && kind <= 3 && (kind != 1 || P % 2 == 0), "lvsr_act_bwd: bad activation %d / P=%d", kind, P);
    if (n <= 0 || P <= 0) return LVSR_OK;
    size_t total = (size_t)n * P;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, kind, x, ldx, dy, lddy, n, P, dx, lddx);
    return lvsr_check_launch("lvsr_act_bwd");
}

int lvsr_softmax_nll(void* stream, const float* logits, int ld, const long long* labels, const float* mask, int n, int V,
                     float* cost, float* dlogits, int ldd, float scale, float* neglogp, int ldn) {
    if (n <= 0 || V <= 0) return LVSR_OK;
    hipLaunchKernelGGL(softmax_nll_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, mask, n, V,
                       cost, dlogits, ldd, scale, neglogp, ldn);
    return lvsr_check_launch("lvsr_softmax_nll");
}

int lvsr_select_cost(void* stream, const float* x, int ld, const long long* labels, const float* mask, int n, int V, float scale,
                     float* cost) {
    LVSR_REQUIRE(x && labels && cost, "lvsr_select_cost: null argument");
    if (n <= 0) return LVSR_OK;
    hipLaunchKernelGGL(select_cost_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, ld, labels, mask, n, V, scale, cost);
    return lvsr_check_launch("lvsr_select_cost");
}

int lvsr_shallow_fusion(void* stream, const float* am, int ld, const float* lm_add, int n, int V, float am_beta,
                        float lm_weight, int norm_am, int norm_lm, int norm_tot, float out_scale, float* out) {
    if (n <= 0 || V <= 0) return LVSR_OK;
    hipLaunchKernelGGL(shallow_fusion_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, am, ld, lm_add, n, V,
                       am_beta, lm_weight, norm_am, norm_lm, norm_tot, out_scale, out);
    return lvsr_check_launch("lvsr_shallow_fusion");
}

}  // extern "C"
