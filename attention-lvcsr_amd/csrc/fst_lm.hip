// Device-side FST language model walk for shallow-fusion beam search (replaces the host Python
// `perform` loops of FSTTransitionOp / FSTCostsOp, lvsr/ops.py:147-169, 206-225).
//
// A hypothesis carries a set of at most 7 FST states with -log weights (lvsr/ops.py:23 MAX_STATES, padded with -1 / 0).
// One launch does, for every hypothesis of the beam,
//   (1) [if `outputs` given] the transition on the chosen character: arcs with that input label out of every state of
//       the set, weights log-added per destination (FST.transition, ops.py:66-77), then the epsilon closure in the log
//       semiring (FST.expand, ops.py:79-97);
//   (2) for all V characters the look-ahead cost  W(closure(transition(set, c))) - W(set),  W = -log sum exp(-w)
//       (FSTCostsOp.perform), `no_transition_cost` where no arc matches.
// This is pointer-chasing integer work over a CSR arc table, not a GEMM: one wave per hypothesis, lane = character, each
// lane walks its own candidate set held in LDS (12 KB per wave); the arc table is read-only and L2 resident.
// Arithmetic is f64 like the reference's Python floats; `add` is rounded to f32 as FSTCostsOp's output is.
//
// The epsilon closure needs no work-list sort: `topo` is the rank of every state in one fixed topological order of the
// epsilon sub-graph (the reference toposorts the reachable part per call and fails on cycles; the host side refuses
// cyclic automata when it builds the table).  Always expanding the unexpanded member of smallest rank visits every
// state after all of its epsilon predecessors, so each weight is final when it is propagated.
#include "common.h"
#include "lvsr_hip.h"

#define FST_MAX_STATES 7            // lvsr/ops.py:23
#define FST_CAP 16                  // candidate-set capacity while expanding (sets larger than this raise err = 2)

struct FstSet {
    int* st;            // [FST_CAP] this lane's candidate states (LDS)
    double* w;          // [FST_CAP]
    int n;
    bool overflow;
};

// -log(e^-a + e^-b)
__device__ __forceinline__ double fst_logadd(double a, double b) {
    const double lo = a < b ? a : b, hi = a < b ? b : a;
    return lo - log1p(exp(lo - hi));
}

__device__ __forceinline__ void fst_insert(FstSet& s, int state, double w) {
    for (int i = 0; i < s.n; ++i)
        if (s.st[i] == state) {
            s.w[i] = fst_logadd(s.w[i], w);
            return;
        }
    if (s.n == FST_CAP) {
        s.overflow = true;
        return;
    }
    s.st[s.n] = state;
    s.w[s.n] = w;
    ++s.n;
}

// set <- closure(transition(src set, label)); returns false if no labelled arc matched
__device__ __forceinline__ void fst_walk(const lvsr_fst& f, const long long* src_st, const double* src_w, int label, FstSet& s) {
    s.n = 0;
    s.overflow = false;
    for (int i = 0; i < FST_MAX_STATES; ++i) {
        const long long q = src_st[i];
        if (q < 0 || q >= f.num_states) continue;
        // arcs of q are sorted by label: lower bound, then scan the equal range (non-deterministic automata allowed)
        int lo = f.arc_off[q], hi = f.arc_off[q + 1];
        const int end = hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (f.arc_lab[mid] < label) lo = mid + 1;
            else hi = mid;
        }
        for (int a = lo; a < end && f.arc_lab[a] == label; ++a) fst_insert(s, f.arc_dst[a], src_w[i] + f.arc_w[a]);
    }
    unsigned done = 0;
    for (;;) {
        int pick = -1, best = 0x7fffffff;
        for (int i = 0; i < s.n; ++i)
            if (!((done >> i) & 1u)) {
                const int r = f.topo[s.st[i]];
                if (r < best) { best = r; pick = i; }
            }
        if (pick < 0) break;
        done |= 1u << pick;
        const int q = s.st[pick];
        const double wq = s.w[pick];
        for (int e = f.eps_off[q]; e < f.eps_off[q + 1]; ++e) fst_insert(s, f.eps_dst[e], wq + f.eps_w[e]);
    }
}

__device__ __forceinline__ double fst_total(const int n, const double* w) {      // -log sum exp(-w), n >= 1
    double lo = w[0];
    for (int i = 1; i < n; ++i) lo = w[i] < lo ? w[i] : lo;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(lo - w[i]);
    return lo - log(s);
}

__global__ __launch_bounds__(64) void fst_lm_step_kernel(lvsr_fst f, const long long* states, const double* weights,
                                                        const long long* outputs, int n, long long* new_states,
                                                        double* new_weights, float* add, int* err, const int* ctl, int group_rows) {
    // batched beam search: the rows of a search that is over carry stale characters — no walk for them (and no error raised by one)
    if (ctl && ctl[(size_t)(blockIdx.x / group_rows) * 16 + 2] != 0) return;
    __shared__ int l_st[64][FST_CAP + 1];
    __shared__ double l_w[64][FST_CAP];
    __shared__ long long cur_st[FST_MAX_STATES];
    __shared__ double cur_w[FST_MAX_STATES];
    const int b = blockIdx.x, lane = threadIdx.x;
    FstSet s;
    s.st = l_st[lane];
    s.w = l_w[lane];
    if (lane == 0) {
        if (outputs) {
            const long long ch = outputs[b];
            const int label = (ch >= 0 && ch < f.V) ? f.remap[ch] : -1;
            if (label < 0) {
                s.n = 0;
                s.overflow = false;
                atomicMax(err, 3);                                  // character without an FST label
            } else {
                fst_walk(f, states + (size_t)b * FST_MAX_STATES, weights + (size_t)b * FST_MAX_STATES, label, s);
            }
            if (s.overflow) atomicMax(err, 2);
            if (s.n > FST_MAX_STATES) {                             // ValueError of lvsr/ops.py:140-142
                atomicMax(err, 1);
                s.n = FST_MAX_STATES;
            }
            for (int i = 0; i < FST_MAX_STATES; ++i) {
                cur_st[i] = i < s.n ? (long long)s.st[i] : -1;
                cur_w[i] = i < s.n ? s.w[i] : 0.0;
                new_states[(size_t)b * FST_MAX_STATES + i] = cur_st[i];
                new_weights[(size_t)b * FST_MAX_STATES + i] = cur_w[i];
            }
        } else {
            for (int i = 0; i < FST_MAX_STATES; ++i) {
                cur_st[i] = states[(size_t)b * FST_MAX_STATES + i];
                cur_w[i] = weights[(size_t)b * FST_MAX_STATES + i];
            }
        }
    }
    __syncthreads();
    if (!add) return;
    double cw[FST_MAX_STATES];
    int nc = 0;
    for (int i = 0; i < FST_MAX_STATES; ++i)
        if (cur_st[i] >= 0 && cur_st[i] < f.num_states) cw[nc++] = cur_w[i];
    const double total = nc ? fst_total(nc, cw) : 0.0;
    for (int c = lane; c < f.V; c += 64) {
        float cost = (float)f.no_transition_cost;
        const int label = f.remap[c];
        if (nc && label >= 0) {
            fst_walk(f, cur_st, cur_w, label, s);
            if (s.overflow) atomicMax(err, 2);
            if (s.n) cost = (float)(fst_total(s.n, s.w) - total);
        }
        add[(size_t)b * f.V + c] = cost;
    }
}

extern "C" {

int lvsr_fst_lm_step(void* stream, const lvsr_fst* f, const long long* states, const double* weights,
                     const long long* outputs, int n, long long* new_states, double* new_weights, float* add, int* err) {
    LVSR_REQUIRE(f && f->arc_off && f->eps_off && f->topo && f->remap && f->num_states > 0 && f->V > 0,
                 "lvsr_fst_lm_step: incomplete automaton table");
    LVSR_REQUIRE(states && weights && err && (!outputs || (new_states && new_weights)), "lvsr_fst_lm_step: null buffers");
    if (n <= 0) return LVSR_OK;
    hipLaunchKernelGGL(fst_lm_step_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, *f, states, weights, outputs, n,
                       new_states, new_weights, add, err, (const int*)nullptr, 1);
    return lvsr_check_launch("lvsr_fst_lm_step");
}

int lvsr_fst_lm_step_groups(void* stream, const lvsr_fst* f, const long long* states, const double* weights,
                            const long long* outputs, int n, long long* new_states, double* new_weights, float* add, int* err,
                            const int* ctl, int group_rows) {
    LVSR_REQUIRE(f && f->arc_off && f->eps_off && f->topo && f->remap && f->num_states > 0 && f->V > 0,
                 "lvsr_fst_lm_step_groups: incomplete automaton table");
    LVSR_REQUIRE(states && weights && err && (!outputs || (new_states && new_weights)) && ctl && group_rows > 0 && n % group_rows == 0,
                 "lvsr_fst_lm_step_groups: null buffers / rows are not whole groups");
    if (n <= 0) return LVSR_OK;
    hipLaunchKernelGGL(fst_lm_step_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, *f, states, weights, outputs, n,
                       new_states, new_weights, add, err, ctl, group_rows);
    return lvsr_check_launch("lvsr_fst_lm_step_groups");
}

}  // extern "C"
