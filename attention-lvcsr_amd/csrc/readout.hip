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

// dst[v, :] = beta*dst[v, :] + sum_{r : idx[r] == v} src[r, :]   (fixed r order: deterministic)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* src, int lds, const long long* idx, int n,
                                                               int width, float* dst, int ldd, float beta) {
    const int v = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (int r = 0; r < n; ++r)
        if (idx[r] == v) s += src[(size_t)r * lds + c];
    float* d = dst + (size_t)v * ldd + c;
    *d = (beta != 0.f ? beta * *d : 0.f) + s;
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

extern "C" {

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

int lvsr_shallow_fusion(void* stream, const float* am, int ld, const float* lm_add, int n, int V, float am_beta,
                        float lm_weight, int norm_am, int norm_lm, int norm_tot, float out_scale, float* out) {
    if (n <= 0 || V <= 0) return LVSR_OK;
    hipLaunchKernelGGL(shallow_fusion_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, am, ld, lm_add, n, V,
                       am_beta, lm_weight, norm_am, norm_lm, norm_tot, out_scale, out);
    return lvsr_check_launch("lvsr_shallow_fusion");
}

}  // extern "C"
