// Device-resident beam search step (K13 of SURVEY.md §2.3) for gfx950: candidate selection, stopping rules and all
// bookkeeping of the reference's (modified) BeamSearch.search (libs/blocks/blocks/search.py:244-407) in ONE small
// kernel per emitted character, so that a beam step needs no device->host synchronisation; the host only polls a
// `done` word every few steps and reads the finished hypotheses back once, at the end.
//
// What must agree with the reference is the RESULT (hypotheses and costs), so its selection rules are kept:
//   * candidates = cumulative cost of a live hypothesis + step cost of a character (float32 like the reference's
//     `all_costs`, search.py:297); the `beam_size` smallest are taken in ascending order (`_smallest`, :221-242).
//     Ties (equal float32 candidates) are ordered by flat index (hypothesis * V + character): the reference's order
//     among equal values is whatever numpy's introselect + quicksort leave and is not portable; a stable rule is;
//   * a hypothesis that emitted <eol> is finished when its last step cost is below `round_to_inf` (:365-367) and leaves
//     the beam (:368-376; kept at position 0 with ignore_first_eol); ranking of finished hypotheses:
//     cumulative cost - char_discount * (number of rows of its cost column) (:307,382);
//   * stop_on = patience: 30 consecutive steps without a better best finished hypothesis (:306-317), the finished list
//     sorted and cut to beam_size before every step; optimistic_future_cost: stop once the beam_size-th finished
//     hypothesis (in order of completion) beats min(live cumulative cost) - char_discount * max_length (:318-330).
// Hypotheses are kept as back-pointers (position, column) -> (parent column, character, cumulative cost) instead of the
// reference's per-step re-gathering of the whole history (:352-357); the host follows them once at the end.
//
// Live hypotheses occupy rows [0, n_live) of the state buffers in beam order; rows beyond are copies of row 0 so that
// every other kernel of the step can run on all K rows without knowing n_live (a duplicated row changes neither another
// row's result nor the batch-wide window of the location prior, lvsr/bricks/attention.py:148-157).
#include "common.h"
#include "lvsr_hip.h"

#define BEAM_THREADS 256
#define BEAM_MAX_CAND 8192       // K * V candidates held in LDS (32 KiB of keys)
#define BEAM_MAX_K 256
#define BEAM_PATIENCE 30

// ctl words
#define CTL_NLIVE 0
#define CTL_POS 1
#define CTL_DONE 2      // 0 running, 1 stopping rule, 2 beam empty, 3 max_length reached
#define CTL_NFIN 3
#define CTL_PATIENCE 4  // -1 = not assigned yet (the reference's UnboundLocalError case)
#define CTL_NSEL 5
#define CTL_ERR 6       // 1 non-finite step cost, 2 finished list full, 3 patience used before assignment
#define CTL_STEPS 7

__device__ __forceinline__ unsigned f2key(float f) {           // monotone: a < b  <=>  key(a) < key(b)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Exclusive prefix sum of one small count per thread over the 256 threads of the work-group (counts <= 8192: exact in
// float32, so the wave shuffles of common.h serve); `wsum` = 4 floats of LDS.  Also returns the total through *total.
__device__ __forceinline__ int block_excl_scan(int v, float* wsum, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = (float)v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    *total = (int)(((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]);
    return (int)(base + x) - v;
}

// The nsel = min(k, n) smallest of keys[0..n) in ascending (key, index) order -> out_idx[0..nsel).
// All threads of the work-group call this; keys and the scratch arrays live in LDS.
__device__ int select_smallest(const unsigned* keys, int n, int k, int* out_idx, unsigned* hist /*[256]*/, int* scal /*[8]*/,
                               int* tmp_idx /*[k]*/) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nsel = min(k, n);
    if (n > k) {
        // ---- radix select of the k-th smallest key, 8 bits per pass
        unsigned prefix = 0, mask = 0;
        int remaining = k;                     // rank (1-based) of the wanted key among those matching the prefix
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += nt) hist[b] = 0;
            __syncthreads();
            for (int x = tid; x < n; x += nt)
                if ((keys[x] & mask) == prefix) atomicAdd(&hist[(keys[x] >> shift) & 255u], 1u);
            __syncthreads();
            {   // the bin that holds the wanted rank: parallel prefix over the 256 bins (a serial walk of the histogram by
                // one thread is 256 dependent LDS reads = 10 us per pass)
                int total;
                const int mine = (int)hist[tid];
                const int before = block_excl_scan(mine, (float*)(scal + 4), &total);
                if (before < remaining && remaining <= before + mine) { scal[0] = tid; scal[1] = remaining - before; }
            }
            __syncthreads();
            prefix |= (unsigned)scal[0] << shift;
            mask |= 255u << shift;
            remaining = scal[1];
            __syncthreads();
        }
        // every key < prefix is in; of the keys == prefix the `remaining` lowest indices are
        if (tid == 0) { scal[2] = 0; scal[3] = 0; }
        __syncthreads();
        // ordinal of a tied candidate among the ties = number of ties with a smaller index: chunked count + serial scan
        const int per = (n + nt - 1) / nt, x0 = min(n, tid * per), x1 = min(n, x0 + per);
        int mine = 0;
        for (int x = x0; x < x1; ++x) mine += keys[x] == prefix;
        int ties_total;
        int ord = block_excl_scan(mine, (float*)(scal + 4), &ties_total);
        for (int x = x0; x < x1; ++x) {
            const unsigned key = keys[x];
            bool take = key < prefix;
            if (key == prefix) { take = ord < remaining; ++ord; }
            if (take) tmp_idx[atomicAdd(&scal[2], 1)] = x;
        }
        __syncthreads();
    } else {
        for (int x = tid; x < n; x += nt) tmp_idx[x] = x;
        __syncthreads();
    }
    // ---- order the nsel selected candidates by (key, index): rank by counting
    for (int a = tid; a < nsel; a += nt) {
        const int xa = tmp_idx[a];
        const unsigned ka = keys[xa];
        int rank = 0;
        for (int b = 0; b < nsel; ++b) {
            const int xb = tmp_idx[b];
            const unsigned kb = keys[xb];
            rank += (kb < ka) || (kb == ka && xb < xa);
        }
        out_idx[rank] = xa;
    }
    __syncthreads();
    return nsel;
}

// ---------------------------------------------------------------------------------------------------------------
// lvsr_topk_smallest: the k smallest of a flat device array, ascending, ties by index
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BEAM_THREADS) void topk_kernel(const float* costs, int n, int k, long long* idx, float* val) {
    __shared__ unsigned keys[BEAM_MAX_CAND];
    __shared__ unsigned hist[256];
    __shared__ int scal[8];
    __shared__ int tmp[BEAM_MAX_K], out[BEAM_MAX_K];
    for (int x = threadIdx.x; x < n; x += blockDim.x) keys[x] = f2key(costs[x]);
    __syncthreads();
    const int nsel = select_smallest(keys, n, k, out, hist, scal, tmp);
    for (int a = threadIdx.x; a < nsel; a += blockDim.x) { idx[a] = out[a]; val[a] = costs[out[a]]; }
}

// ---------------------------------------------------------------------------------------------------------------
// one beam step
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double fin_score_of(const lvsr_beam_args& a, float cost, int pos) {
    // running[-1] - char_discount * len(running): the column has pos + 2 rows (the initial zero + one per position).  In double:
    // the reference subtracts a Python float from a numpy.float32 (float64 under the numpy it was written for), and the host's
    // final ranking (search.py _collect) does the same — near-equal scores must order identically here and there
    return (double)cost - a.char_discount * (double)(pos + 2);
}

// Argument block of search g of a grouped launch (lvsr_beam_args.groups): every buffer is G consecutive blocks
__device__ __forceinline__ lvsr_beam_args beam_group(lvsr_beam_args a, int g) {
    if (g == 0) return a;
    const size_t K = (size_t)a.K, r = (size_t)g * K, V = (size_t)a.V, D = (size_t)a.D, Tp = (size_t)a.Tp;
#define BG_OFF(field, n) if (a.field) a.field += (n)
    BG_OFF(ctl, (size_t)g * 16); BG_OFF(fctl, (size_t)g * 4);
    BG_OFF(neglogp, r * V); BG_OFF(running, r); BG_OFF(live_col, r);
    BG_OFF(hist_parent, (size_t)g * a.max_length * K); BG_OFF(hist_char, (size_t)g * a.max_length * K); BG_OFF(hist_cost, (size_t)g * a.max_length * K);
    BG_OFF(fin_pos, (size_t)g * a.fin_cap); BG_OFF(fin_col, (size_t)g * a.fin_cap); BG_OFF(fin_cost, (size_t)g * a.fin_cap); BG_OFF(fin_score, (size_t)g * a.fin_cap);
    BG_OFF(keep, r); BG_OFF(chars, r); BG_OFF(parents, r);
    BG_OFF(S_live, r * D); BG_OFF(W_live, r * Tp); BG_OFF(S_sel, r * D); BG_OFF(W_sel, r * Tp);
    BG_OFF(lm_states_live, r * 7); BG_OFF(lm_weights_live, r * 7); BG_OFF(lm_states_sel, r * 7); BG_OFF(lm_weights_sel, r * 7);
    BG_OFF(pos_live, r); BG_OFF(pos_sel, r);
    BG_OFF(fork_xg, r * 3 * D);
    BG_OFF(pos_new, r); BG_OFF(pos_live_out, r);
    BG_OFF(S_new, r * D); BG_OFF(W_new, r * Tp); BG_OFF(S_live_out, r * D); BG_OFF(W_live_out, r * Tp);
    BG_OFF(lm_states_new, r * 7); BG_OFF(lm_weights_new, r * 7); BG_OFF(lm_add_new, r * V);
    BG_OFF(lm_states_live_out, r * 7); BG_OFF(lm_weights_live_out, r * 7); BG_OFF(lm_add_live_out, r * V);
    BG_OFF(WA_live, r * a.E); BG_OFF(WA_sel, r * a.E); BG_OFF(W1_live, r * Tp); BG_OFF(W1_sel, r * Tp);
    BG_OFF(pos1_live, r); BG_OFF(pos1_sel, r);
#undef BG_OFF
    return a;
}

__global__ __launch_bounds__(BEAM_THREADS) void beam_select_kernel(lvsr_beam_args a0) {
    const lvsr_beam_args a = beam_group(a0, blockIdx.x);
    __shared__ unsigned keys[BEAM_MAX_CAND];
    __shared__ unsigned hist[256];
    __shared__ int scal[8];
    __shared__ int tmp[BEAM_MAX_K], sel[BEAM_MAX_K];
    // everything the serial parts touch is staged in LDS: a dependent global access costs ~1 us on this chip
    __shared__ float s_run[BEAM_MAX_K], s_cost[BEAM_MAX_K];
    __shared__ int s_par[BEAM_MAX_K], s_ch[BEAM_MAX_K], s_col[BEAM_MAX_K], s_keep[BEAM_MAX_K], s_finslot[BEAM_MAX_K];
    __shared__ float f_score[2 * BEAM_MAX_K], f_cost[2 * BEAM_MAX_K];
    __shared__ int f_pos[2 * BEAM_MAX_K], f_col[2 * BEAM_MAX_K];
    __shared__ int s_ctl[8];
    __shared__ double s_best, f_scd[2 * BEAM_MAX_K];
    __shared__ int s_stop, s_bad;
    int* ctl = a.ctl;
    const int tid = threadIdx.x, nt = blockDim.x, K = a.K, V = a.V;
    if (tid < 8) s_ctl[tid] = ctl[tid];
    // best finished score so far: a double in fctl[2..3]; before the first position the caller's float in fctl[0] (1000)
    if (tid == 8) s_best = ctl[CTL_POS] == 0 ? (double)a.fctl[0] : *(const double*)(a.fctl + 2);
    __syncthreads();
    if (s_ctl[CTL_DONE] != 0) return;
    const int max_length = a.groups > 1 ? ctl[8] : a.max_length;          // (grouped: every search has its own limit)
    const int n = s_ctl[CTL_NLIVE], p = s_ctl[CTL_POS];
    int nf = s_ctl[CTL_NFIN];
    for (int i = tid; i < n; i += nt) { s_run[i] = a.running[i]; s_col[i] = a.live_col[i]; }
    // ---- stopping rules of the loop head (search.py:300-330), on the state the previous step left
    if (a.stop_on == 0 && n > 0) {          // (an empty beam ends the search BEFORE the list is sorted and cut: search.py:301-302)
        // finished.sort(key=score); del finished[beam_size:] — stable rank sort of the (at most 2K) entries
        for (int i = tid; i < nf; i += nt) {
            f_score[i] = a.fin_score[i]; f_cost[i] = a.fin_cost[i]; f_pos[i] = a.fin_pos[i]; f_col[i] = a.fin_col[i];
            f_scd[i] = fin_score_of(a, f_cost[i], f_pos[i]);
        }
        __syncthreads();
        for (int i = tid; i < nf; i += nt) {
            const double sc = f_scd[i];
            int rank = 0;
            for (int j = 0; j < nf; ++j) rank += (f_scd[j] < sc) || (f_scd[j] == sc && j < i);
            if (rank < K) {
                a.fin_score[rank] = f_score[i]; a.fin_cost[rank] = f_cost[i]; a.fin_pos[rank] = f_pos[i]; a.fin_col[rank] = f_col[i];
            }
            if (rank == 0) scal[4] = i;
        }
        __syncthreads();
        if (nf > K) nf = K;
    }
    __syncthreads();
    if (tid == 0) {
        int stop = 0;
        s_ctl[CTL_NFIN] = nf;
        if (n == 0) stop = 2;
        else if (a.stop_on == 0) {
            if (nf > 0) {
                const double leader = f_scd[scal[4]];
                if (leader < s_best) { s_best = leader; s_ctl[CTL_PATIENCE] = BEAM_PATIENCE; }
                else if (s_ctl[CTL_PATIENCE] < 0) { s_ctl[CTL_ERR] = 3; stop = 1; }
                else if (--s_ctl[CTL_PATIENCE] == 0) stop = 1;
            }
        } else if (nf >= K) {
            float mn = s_run[0];
            for (int i = 1; i < n; ++i) mn = fminf(mn, s_run[i]);
            const double bound = (double)mn - a.char_discount * (double)max_length;
            if (fin_score_of(a, a.fin_cost[K - 1], a.fin_pos[K - 1]) < bound) stop = 1;
        }
        s_stop = stop; s_bad = 0;
    }
    __syncthreads();
    if (s_stop) {
        if (tid == 0) {
            ctl[CTL_DONE] = s_stop; ctl[CTL_NFIN] = s_ctl[CTL_NFIN]; ctl[CTL_PATIENCE] = s_ctl[CTL_PATIENCE];
            ctl[CTL_ERR] = s_ctl[CTL_ERR]; a.fctl[0] = (float)s_best; *(double*)(a.fctl + 2) = s_best;
        }
        return;
    }
    // ---- candidates (search.py:341-347): float32 sums, finite by the reference's assert
    const int N = n * V;
    for (int x = tid; x < N; x += nt) {
        const float c = a.neglogp[x];
        if (!(fabsf(c) <= 3.0e38f)) s_bad = 1;
        keys[x] = f2key(s_run[x / V] + c);
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) { ctl[CTL_ERR] = 1; ctl[CTL_DONE] = 1; }
        return;
    }
    const int nsel = select_smallest(keys, N, K, sel, hist, scal, tmp);
    // ---- history rows of this position; chosen characters; rows of the next-state pass
    for (int k = tid; k < K; k += nt) {
        const int x = sel[k < nsel ? k : 0];                       // rows beyond nsel replicate candidate 0
        const int par = x / V, ch = x % V;
        const float cost = s_run[par] + a.neglogp[x];
        s_par[k] = par; s_ch[k] = ch; s_cost[k] = cost;
        a.chars[k] = ch;
        a.parents[k] = par;
        if (k < nsel) {
            a.hist_parent[(size_t)p * K + k] = s_col[par];
            a.hist_char[(size_t)p * K + k] = ch;
            a.hist_cost[(size_t)p * K + k] = cost;
        }
    }
    __syncthreads();
    // ---- finished hypotheses, the new beam (search.py:358-376); order of columns preserved
    if (tid == 0) {
        int nl = 0, nfn = nf;
        const bool keep_all = a.ignore_first_eol && p == 0;
        for (int k = 0; k < nsel; ++k) {
            const bool ended = s_ch[k] == a.eol;
            int slot = -1;
            if (ended && (s_cost[k] - s_run[s_par[k]]) < a.round_to_inf) {
                if (nfn < a.fin_cap) slot = nfn++;
                else s_ctl[CTL_ERR] = 2;
            }
            s_finslot[k] = slot;
            if (!ended || keep_all) s_keep[nl++] = k;
        }
        for (int i = nl; i < K; ++i) s_keep[i] = nl > 0 ? s_keep[0] : 0;      // rows beyond the beam replicate row 0
        s_ctl[CTL_NFIN] = nfn;
        s_ctl[CTL_NLIVE] = nl;
    }
    __syncthreads();
    const int nl = s_ctl[CTL_NLIVE];
    for (int k = tid; k < nsel; k += nt) {
        const int slot = s_finslot[k];
        if (slot >= 0) {
            a.fin_pos[slot] = p; a.fin_col[slot] = k; a.fin_cost[slot] = s_cost[k]; a.fin_score[slot] = (float)fin_score_of(a, s_cost[k], p);
        }
    }
    for (int i = tid; i < K; i += nt) {
        const int k = s_keep[i];
        a.keep[i] = k;
        if (i < nl) { a.live_col[i] = k; a.running[i] = s_cost[k]; }
    }
    if (tid == 0 && a.WA_live) {
        // do the selected rows span the window centres of the live rows (all K rows of either pass: the tails replicate row 0 /
        // candidate 0)?  Then the second attention pass would repeat the first one row by row: see lvsr_beam_args.WA_live
        int same = 1;
        if (a.pos_live) {
            float mnA = 3.0e38f, mxA = -3.0e38f, mnB = 3.0e38f, mxB = -3.0e38f;
            for (int k = 0; k < K; ++k) {
                const float pa = a.pos_live[k], pb = a.pos_live[s_par[k]];
                mnA = fminf(mnA, pa); mxA = fmaxf(mxA, pa); mnB = fminf(mnB, pb); mxB = fmaxf(mxB, pb);
            }
            same = (mnA == mnB && mxA == mxB) ? 1 : 0;
        }
        ctl[9] = same;
        ctl[10] += same;
    }
    if (tid == 0) {
        ctl[CTL_NLIVE] = nl;
        ctl[CTL_POS] = p + 1;
        ctl[CTL_NFIN] = s_ctl[CTL_NFIN];
        ctl[CTL_PATIENCE] = s_ctl[CTL_PATIENCE];
        ctl[CTL_NSEL] = nsel;
        ctl[CTL_ERR] = s_ctl[CTL_ERR];
        ctl[CTL_STEPS] = s_ctl[CTL_STEPS] + 1;
        if (p + 1 >= max_length) ctl[CTL_DONE] = 3;
        a.fctl[0] = (float)s_best; *(double*)(a.fctl + 2) = s_best;
    }
}

// rows of the next-state pass: sel[k] = live[parent of candidate k] (state, alignment, window centre, language-model state
// set) and, for one-hot feedback, the fork inputs of the chosen character (OneOfNFeedback + Fork = row gathers + biases,
// lvsr/bricks/__init__.py:97-104); one work-group per row
__global__ __launch_bounds__(256) void beam_rows_kernel(lvsr_beam_args a0) {
    const lvsr_beam_args a = beam_group(a0, blockIdx.y);
    const int k = blockIdx.x, par = a.parents[k], tid = threadIdx.x;
    const float* __restrict__ s_src = a.S_live + (size_t)par * a.D;
    const float* __restrict__ w_src = a.W_live + (size_t)par * a.Tp;
    float* __restrict__ s_dst = a.S_sel + (size_t)k * a.D;
    float* __restrict__ w_dst = a.W_sel + (size_t)k * a.Tp;
    for (int j = tid; j < a.D; j += 256) s_dst[j] = s_src[j];
    for (int j = tid; j < a.Tp; j += 256) w_dst[j] = w_src[j];
    if (a.pos_live && tid == 0) a.pos_sel[k] = a.pos_live[par];
    if (a.WA_live && a.ctl[9] != 0) {          // the second attention pass is skipped for this search: its results are the parent's
        const float* __restrict__ wa_src = a.WA_live + (size_t)par * a.E;
        const float* __restrict__ w1_src = a.W1_live + (size_t)par * a.Tp;
        float* __restrict__ wa_dst = a.WA_sel + (size_t)k * a.E;
        float* __restrict__ w1_dst = a.W1_sel + (size_t)k * a.Tp;
        for (int j = tid; j < a.E; j += 256) wa_dst[j] = wa_src[j];
        for (int j = tid; j < a.Tp; j += 256) w1_dst[j] = w1_src[j];
        if (a.pos1_live && tid == 0) a.pos1_sel[k] = a.pos1_live[par];
    }
    if (a.lm_states_live && tid < 7) {
        a.lm_states_sel[(size_t)k * 7 + tid] = a.lm_states_live[(size_t)par * 7 + tid];
        a.lm_weights_sel[(size_t)k * 7 + tid] = a.lm_weights_live[(size_t)par * 7 + tid];
    }
    if (a.fork_xg) {
        const long long ch = a.chars[k];
        const int D = a.D;
        float* __restrict__ xg = a.fork_xg + (size_t)k * 3 * D;
        for (int j = tid; j < 3 * D; j += 256) {
            float v;
            if (j < D) v = a.fork_bi[j] + ((ch >= 0 && ch < a.fork_rows) ? a.fork_Wi[(size_t)ch * D + j] : 0.f);
            else v = a.fork_bg[j - D] + ((ch >= 0 && ch < a.fork_rows) ? a.fork_Wg[(size_t)ch * 2 * D + (j - D)] : 0.f);
            xg[j] = v;
        }
    }
}

// new live rows <- rows keep[i] of the next-state pass (all K rows: the tail replicates row keep[0])
__global__ __launch_bounds__(256) void beam_compact_kernel(lvsr_beam_args a0) {
    const lvsr_beam_args a = beam_group(a0, blockIdx.y);
    const int i = blockIdx.x, k = a.keep[i];
    const float* __restrict__ s_src = a.S_new + (size_t)k * a.D;
    const float* __restrict__ w_src = a.W_new + (size_t)k * a.Tp;
    float* __restrict__ s_dst = a.S_live_out + (size_t)i * a.D;
    float* __restrict__ w_dst = a.W_live_out + (size_t)i * a.Tp;
    for (int j = threadIdx.x; j < a.D; j += 256) s_dst[j] = s_src[j];
    for (int j = threadIdx.x; j < a.Tp; j += 256) w_dst[j] = w_src[j];
    if (a.pos_new && threadIdx.x == 0) a.pos_live_out[i] = a.pos_new[k];
    if (a.lm_states_new) {
        if (threadIdx.x < 7) {
            a.lm_states_live_out[(size_t)i * 7 + threadIdx.x] = a.lm_states_new[(size_t)k * 7 + threadIdx.x];
            a.lm_weights_live_out[(size_t)i * 7 + threadIdx.x] = a.lm_weights_new[(size_t)k * 7 + threadIdx.x];
        }
        for (int j = threadIdx.x; j < a.V; j += 256) a.lm_add_live_out[(size_t)i * a.V + j] = a.lm_add_new[(size_t)k * a.V + j];
    }
}

static int beam_check(const lvsr_beam_args& a, const char* what) {
    LVSR_REQUIRE(a.K > 0 && a.K <= BEAM_MAX_K && a.V > 0 && (long long)a.K * a.V <= BEAM_MAX_CAND,
                 "%s: beam %d x %d characters exceeds the kernel's capacity (K <= %d, K*V <= %d)", what, a.K, a.V, BEAM_MAX_K,
                 BEAM_MAX_CAND);
    LVSR_REQUIRE(a.ctl && a.fctl && a.neglogp && a.running && a.live_col && a.hist_parent && a.hist_char && a.hist_cost &&
                 a.fin_pos && a.fin_col && a.fin_cost && a.fin_score && a.keep && a.chars && a.parents, "%s: null state buffer", what);
    LVSR_REQUIRE(a.stop_on == 0 || a.stop_on == 1, "%s: unknown stopping criterion %d", what, a.stop_on);
    LVSR_REQUIRE(a.fin_cap >= 2 * a.K, "%s: finished list shorter than 2 * beam", what);
    LVSR_REQUIRE(a.groups >= 0 && a.groups <= 65535, "%s: bad number of searches", what);
    LVSR_REQUIRE(!a.WA_live || (a.WA_sel && a.W1_live && a.W1_sel && a.E > 0 && (!a.pos_live || (a.pos1_live && a.pos1_sel))),
                 "%s: incomplete description of the attention results to reuse", what);
    return LVSR_OK;
}

extern "C" {

int lvsr_beam_select(void* stream, const lvsr_beam_args* args) {
    LVSR_REQUIRE(args != nullptr, "lvsr_beam_select: null args");
    if (int rc = beam_check(*args, "lvsr_beam_select")) return rc;
    LVSR_REQUIRE(args->S_live && args->W_live && args->S_sel && args->W_sel && args->D > 0 && args->Tp > 0,
                 "lvsr_beam_select: null row buffers");
    LVSR_REQUIRE(!args->fork_xg || (args->fork_Wi && args->fork_Wg && args->fork_bi && args->fork_bg && args->fork_rows > 0),
                 "lvsr_beam_select: incomplete fork description");
    const int G = args->groups > 1 ? args->groups : 1;
    hipLaunchKernelGGL(beam_select_kernel, dim3(G), dim3(BEAM_THREADS), 0, (hipStream_t)stream, *args);
    hipLaunchKernelGGL(beam_rows_kernel, dim3(args->K, G), dim3(256), 0, (hipStream_t)stream, *args);
    return lvsr_check_launch("lvsr_beam_select");
}

int lvsr_beam_compact(void* stream, const lvsr_beam_args* args) {
    LVSR_REQUIRE(args != nullptr, "lvsr_beam_compact: null args");
    LVSR_REQUIRE(args->keep && args->S_new && args->W_new && args->S_live_out && args->W_live_out, "lvsr_beam_compact: null row buffers");
    hipLaunchKernelGGL(beam_compact_kernel, dim3(args->K, args->groups > 1 ? args->groups : 1), dim3(256), 0, (hipStream_t)stream, *args);
    return lvsr_check_launch("lvsr_beam_compact");
}

int lvsr_topk_smallest(void* stream, const float* costs, int n, int k, long long* idx, float* val) {
    LVSR_REQUIRE(costs && idx && val && n > 0 && k > 0, "lvsr_topk_smallest: bad arguments");
    LVSR_REQUIRE(n <= BEAM_MAX_CAND && k <= BEAM_MAX_K, "lvsr_topk_smallest: n <= %d and k <= %d", BEAM_MAX_CAND, BEAM_MAX_K);
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(BEAM_THREADS), 0, (hipStream_t)stream, costs, n, k, idx, val);
    return lvsr_check_launch("lvsr_topk_smallest");
}

}  // extern "C"
