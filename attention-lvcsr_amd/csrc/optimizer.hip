// Training step rules on the flat parameter / gradient buffers (K15 of SURVEY.md §2.3): one pass over
// 5.2 M floats per rule group instead of Theano's per-variable update graph.
//
// Rule chain of the reference (lvsr/main.py:480-519):
//   StepClipping(threshold)            libs/blocks/blocks/algorithms/__init__.py:610-643  (global L2 norm)
//   Momentum(scale, momentum)          :431-461  = Scale then BasicMomentum
//   AdaDelta(decay_rate, epsilon)      :464-515
//   Restrict(VariableClipping(max_norm, axis=0), WEIGHT-role parameters)   :646-720, :864-893
//   RemoveNotFinite(scaler)            :829-861  (per parameter tensor)
//   BurnIn(num_steps)                  lvsr/algorithms.py:19-43
// followed by parameter -= step (GradientDescent, :284-287 / :244-256), and the AdaptiveClipping extension
// (lvsr/extensions.py:64-91) that re-tunes the StepClipping threshold after every batch: here a few scalar operations
// of the norm kernel on device-resident state, so the whole step stays graph-replayable.
#include "common.h"
#include "lvsr_hip.h"
#include <string.h>

typedef lvsr_opt_args Opt;

__device__ __forceinline__ float blk_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b] = sum over the block's slice of (grad*grad_scale)^2   (fixed slices: deterministic)
__global__ __launch_bounds__(256) void opt_sqnorm_kernel(Opt o) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < o.n; i += (long long)gridDim.x * 256) {
        const float g = o.grad[i] * o.grad_scale;
        s += g * g;
    }
    s = blk_sum256(s, red);
    if (threadIdx.x == 0) o.scratch[2 + blockIdx.x] = s;
}

// scratch[0] = gradient norm, scratch[1] = StepClipping multiplier
__global__ __launch_bounds__(256) void opt_norm_kernel(Opt o, int nparts) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += o.scratch[2 + i];
    s = blk_sum256(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        // guard: a cluster kernel of this step gave up -> gradients are garbage: skip the step altogether (scratch[3] tells the
        // kernels behind this one, and the host)
        const bool skip = o.guard && o.guard[0] != 0.f;
        o.scratch[3] = skip ? 1.f : 0.f;
        double* cs = skip ? nullptr : o.clip_state;
        const float thr = cs ? (float)cs[0] : o.clip_threshold;        // Theano shared floatX
        o.scratch[0] = norm;
        o.scratch[1] = (thr > 0.f && !(norm < thr)) ? thr / norm : 1.f;
        float burn = 0.f;
        if (cs) {
            if (cs[4] > 0.0) { burn = 1.f; cs[4] -= 1.0; }
            // AdaptiveClipping.after_batch; a non-finite (or zero) norm is skipped: in the reference log(nan) poisons the
            // running statistics and with them every later threshold (deliberate deviation)
            if (o.adaptive_clipping && norm > 0.f && norm - norm == 0.f) {
                const double d = (double)o.adaptive_decay, init = (double)o.clip_threshold, bp = (double)o.adaptive_burnin;
                const double ln = log((double)norm);
                cs[3] += 1.0;
                cs[1] = d * cs[1] + (1.0 - d) * ln;
                cs[2] = d * cs[2] + (1.0 - d) * ln * ln;
                const double var = cs[2] - cs[1] * cs[1];
                double t = exp(cs[1] + sqrt(var > 0.0 ? var : 0.0));
                const double conf = (cs[3] < bp ? cs[3] : bp) / bp;
                t = conf * t + (1.0 - conf) * init;
                cs[0] = t < 5.0 * init ? t : 5.0 * init;
            }
        }
        o.scratch[2] = burn;                                           // the partial sums are consumed
    }
}

// clipping multiplier, Scale, BasicMomentum, AdaDelta -> step
__global__ __launch_bounds__(256) void opt_rules_kernel(Opt o) {
    if (o.scratch[3] != 0.f) return;                                   // guarded step: rule state untouched
    const float mult = o.scratch[1];
    if (blockIdx.x == 0)
        for (int sgi = threadIdx.x; sgi < o.nseg; sgi += 256) o.segflag[sgi] = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < o.n; i += (long long)gridDim.x * 256) {
        float s = o.grad[i] * o.grad_scale * mult;
        if (o.use_momentum) {
            s = o.learning_rate * s;
            s = o.momentum * o.velocity[i] + s;
            o.velocity[i] = s;
        }
        if (o.use_adadelta) {
            const float ms = o.decay_rate * o.ms_step[i] + (1.f - o.decay_rate) * s * s;
            const float dx = sqrtf(o.ms_dx[i] + o.epsilon) / sqrtf(ms + o.epsilon) * s;
            o.ms_step[i] = ms;
            o.ms_dx[i] = o.decay_rate * o.ms_dx[i] + (1.f - o.decay_rate) * dx * dx;
            s = dx;
        }
        o.step[i] = s;
    }
}

// VariableClipping(axis=0) on flagged (rows x cols) segments.  Block (x, seg): 64 columns x 16 row groups (1024 threads: the
// kernel is a latency chain of strided loads, so 16 rows of a column are in flight at once instead of 4: 97 -> ~25 us); the
// column norms of (param - step) are folded through LDS in a fixed order, then the same threads rescale their rows.
#define MAXNORM_GROUPS 16
__global__ __launch_bounds__(64 * MAXNORM_GROUPS) void opt_maxnorm_kernel(Opt o) {
    __shared__ float red[MAXNORM_GROUPS][64];
    const long long* seg = o.segments + 4 * (long long)blockIdx.y;
    const long long off = seg[0], rows = seg[1], cols = seg[2], flags = seg[3];
    if (!(flags & 1) || o.scratch[3] != 0.f) return;
    const long long j = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if ((long long)blockIdx.x * 64 >= cols) return;
    const int g = threadIdx.x >> 6;
    float s = 0.f;
    if (j < cols)
        for (long long r = g; r < rows; r += MAXNORM_GROUPS) {
            const float v = o.param[off + r * cols + j] - o.step[off + r * cols + j];
            s += v * v;
        }
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    const int c = threadIdx.x & 63;
    float tot = 0.f;
#pragma unroll
    for (int gg = 0; gg < MAXNORM_GROUPS; ++gg) tot += red[gg][c];
    const float norm = sqrtf(tot);
    if (j < cols && norm > o.max_norm) {
        const float k = o.max_norm / norm;
        for (long long r = g; r < rows; r += MAXNORM_GROUPS) {
            const long long x = off + r * cols + j;
            o.step[x] = o.param[x] - k * (o.param[x] - o.step[x]);
        }
    }
}

// RemoveNotFinite: segflag[s] = 1 when sum(step of segment s) is nan/inf.  A non-finite sum needs a non-finite element
// or an overflow; either way some partial sum of the segment is non-finite, so blocks (x, seg) test their slice and raise
// the flag (idempotent store, no ordering needed); the flags are cleared by the rules kernel of the same step.
__global__ __launch_bounds__(256) void opt_finite_kernel(Opt o) {
    __shared__ float red[4];
    const long long* seg = o.segments + 4 * (long long)blockIdx.y;
    const long long off = seg[0], n = seg[1] * seg[2];
    if (o.scratch[3] != 0.f) return;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += o.step[off + i];
    s = blk_sum256(s, red);
    if (threadIdx.x == 0 && (s != s || s - s != 0.f)) o.segflag[blockIdx.y] = 1;
}

__global__ __launch_bounds__(256) void opt_apply_kernel(Opt o) {
    const long long* seg = o.segments + 4 * (long long)blockIdx.y;
    const long long off = seg[0], n = seg[1] * seg[2];
    const int bad = o.remove_not_finite ? o.segflag[blockIdx.y] : 0;
    if (o.scratch[3] != 0.f) return;                                   // guarded step
    if (o.clip_state && o.scratch[2] != 0.f) return;                   // BurnIn: the step is multiplied by zero
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = o.param[off + i];
        const float s = bad ? (1.f - o.nonfinite_scaler) * p : o.step[off + i];
        o.param[off + i] = p - s;
    }
}

struct GuardPack { const int* w[64]; int n; };
__global__ void guard_collect_kernel(GuardPack pk, float* out) {
    if (threadIdx.x == 0) {
        int bad = 0;
        for (int i = 0; i < pk.n; ++i)
            if (pk.w[i] && __hip_atomic_load(pk.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ++bad;
        out[0] = (float)bad;
    }
}

extern "C" int lvsr_guard_collect(void* stream, const int* const* words, int n, float* out) {
    LVSR_REQUIRE(n >= 0 && n <= 64 && out && (n == 0 || words), "lvsr_guard_collect: at most 64 words");
    GuardPack pk;
    pk.n = n;
    for (int i = 0; i < 64; ++i) pk.w[i] = i < n ? words[i] : nullptr;
    hipLaunchKernelGGL(guard_collect_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pk, out);
    return lvsr_check_launch("lvsr_guard_collect");
}

extern "C" int lvsr_opt_step(void* stream, const lvsr_opt_args* args) {
    LVSR_REQUIRE(args != nullptr, "lvsr_opt_step: null args");
    Opt o;
    memcpy(&o, args, sizeof(o));
    LVSR_REQUIRE(o.n > 0 && o.nseg > 0 && o.param && o.grad && o.step && o.segments && o.scratch && o.segflag,
                 "lvsr_opt_step: missing buffers");
    LVSR_REQUIRE(!o.use_momentum || o.velocity, "lvsr_opt_step: momentum needs a velocity buffer");
    LVSR_REQUIRE(!o.use_adadelta || (o.ms_step && o.ms_dx), "lvsr_opt_step: AdaDelta needs its two accumulators");
    LVSR_REQUIRE(!o.adaptive_clipping || (o.clip_state && o.clip_threshold > 0.f && o.adaptive_burnin > 0),
                 "lvsr_opt_step: adaptive clipping needs clip_state, an initial threshold and a burn-in period");
    hipStream_t s = (hipStream_t)stream;
    const int nparts = 256;
    int nb = (int)((o.n + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(opt_sqnorm_kernel, dim3(nparts), dim3(256), 0, s, o);
    hipLaunchKernelGGL(opt_norm_kernel, dim3(1), dim3(256), 0, s, o, nparts);
    hipLaunchKernelGGL(opt_rules_kernel, dim3(nb), dim3(256), 0, s, o);
    if (o.max_norm > 0.f)
        hipLaunchKernelGGL(opt_maxnorm_kernel, dim3((o.max_cols + 63) / 64, o.nseg), dim3(64 * MAXNORM_GROUPS), 0, s, o);
    if (o.remove_not_finite)
        hipLaunchKernelGGL(opt_finite_kernel, dim3(16, o.nseg), dim3(256), 0, s, o);
    hipLaunchKernelGGL(opt_apply_kernel, dim3(16, o.nseg), dim3(256), 0, s, o);
    return lvsr_check_launch("lvsr_opt_step");
}
