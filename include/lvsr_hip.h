/* lvsr_hip.h — C ABI of the MI355X (gfx950) hot path of attention-LVCSR.
 *
 * The reference (rizar/attention-lvcsr) has no FFI for this path: the path sits behind Python bricks
 * (lvsr/bricks/recognizer.py:159-562) whose arithmetic Theano generates.  This header is the boundary a
 * maintainer binds instead (ctypes stub: attention-lvcsr_amd/lvsr_amd/native.py).
 * Every entry point cites the reference function whose arithmetic it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - plain C: device pointers + sizes; `stream` is a hipStream_t passed as void*; calls are asynchronous
 *     on that stream; the caller owns every buffer.
 *   - return 0 on success, <0 on error (message: lvsr_last_error()).
 *   - all tensors fp32 row-major in the reference's time-major layouts unless noted;
 *     labels int64; masks fp32 0/1.
 *   - "packed" weights: lvsr_pack_b() output (MFMA operand order), see below.
 *   - `use_graph`: capture the whole time loop once per argument block into a hipGraph and replay it
 *     (argument blocks must therefore be stable across calls; the stream must be capturable, i.e. not
 *     the legacy null stream).
 */
#ifndef LVSR_HIP_H
#define LVSR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ------------------------------------------------------------------------------------ */
const char* lvsr_last_error(void);
int lvsr_abi_version(void);
void lvsr_graph_clear(void);   /* drop cached hipGraphs (call before freeing buffers they reference) */
int lvsr_graph_count(void);
/* Graph regions: everything enqueued on `stream` between begin and end (library calls, copies, fills) becomes ONE cached
 * hipGraph keyed by `key` (the caller's description of every pointer, size and scalar the enqueue code depends on).
 * begin returns 1 = cached graph launched (skip the enqueue code and the end call), 0 = capturing (enqueue, then call
 * end), 2 = not capturable (enqueue eagerly, no end call).  end(keep=1) instantiates, caches and launches;
 * end(keep=0) drops the capture and marks the key not capturable.  Entry points called with use_graph=1 inside a region
 * just record their launches.  Used for the whole training step: one launch per step instead of ~25 graphs + ~150 kernels. */
void lvsr_graph_suppress(int on);   /* 1: use_graph arguments are ignored (eager launches, nothing cached) until switched off */
int lvsr_region_begin(void* stream, const char* key, long long key_bytes);
int lvsr_region_end(void* stream, int keep);
/* Tuning knobs: process-wide integers that select kernel variants for probing and A/B measurements (tools/probe_*.py).  The
 * defaults (all 0) are the benchmarked configuration; the library never reads the environment.  Knob values are part of the
 * key of every cached graph, so a graph captured under one setting is not replayed under another. */
#define LVSR_KNOB_PERSIST_ROWS 0      /* utterances per encoder cluster: 0 = the smallest number whose grid fits the chip; 1, 2, 4, 8 = at least that */
#define LVSR_KNOB_PERSIST_FLAGS 1     /* kernel-variant bits of csrc/persist.h (PF_*; same results): 2 clusters spread over XCDs, 4 write-through publish, 64 clusters of 4,
                                         4096 / 8192 no loader wave in the BPTT / forward encoder kernel, 16384 decoder reverse walk sums dPA in LDS instead of by L2 atomics (opt-in), ...
                                         The timing ablations that change results (1, 8, 16, 128) are refused: they exist only in the probe build (csrc/build.py --probes) */
#define LVSR_KNOB_PERSIST_THREADS 2   /* 0 = 512-thread work-groups when a cluster serves <= 2 utterances; 256 = always 256 */
#define LVSR_KNOB_PHASE_CLOCK 3       /* 1 = work-group 0 of the persistent decoder kernels leaves per-phase times in the workspace header */
#define LVSR_KNOB_MAX_CLUSTER_WGS 4   /* 0 = device CU count (256 on MI355X); else the largest one-work-group-per-CU grid a cluster launch may have */
#define LVSR_KNOB_CLUSTER_RESERVE 5   /* CUs left free by cluster launches for other work on the device (default 0) */
#define LVSR_KNOB_GEMM_MID_TILES 6    /* lvsr_sgemm with K <= 2048 uses 64 x 64 tiles when the output has fewer 128 x 128 tiles than this (0 = 2048; 1 = never) */
#define LVSR_KNOB_DEC_CLUSTER 7       /* persistent decoder at D <= 256: 0 = clusters of 16 work-groups per utterance when they fit the chip, else 8; 8 / 16 = only that */
#define LVSR_KNOB_ENERGY_ROWS 8       /* step-kernel energies (row groups): rows of a group per work-group; 0 = 8 while the grid is at most one round of work-groups, else all */
#define LVSR_KNOB_COUNT 9
int lvsr_set_knob(int knob, int value);
int lvsr_get_knob(int knob);

/* ---- dense helpers (Linear bricks: libs/blocks/blocks/bricks/simple.py:59-76) ------------------ */
/* C[M,N] = alpha*op(A)[M,K]*op(B)[K,N] + beta*C + bias[N]; fp32 MFMA; ws: optional split-K workspace */
int lvsr_sgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
               const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
               long long ws_bytes);
/* `batch` independent products in one launch: problem i uses A + i*strideA, B + i*strideB, C + i*strideC (elements).
 * Here: the per-utterance alignment^T x glimpse-gradient products of the attended-sequence gradient
 * (d compute_weighted_averages, libs/blocks/blocks/bricks/attention.py:236-256). */
int lvsr_sgemm_batched(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                       long long strideA, const float* B, int ldb, long long strideB, float beta, float* C, int ldc,
                       long long strideC, int batch);
/* n transposed-A products C_i = A_i^T B_i + beta_i C_i (A_i: K x M with leading dimension lda, B_i: K x N) in ONE launch (+ one
 * that folds the k-chunk partials in a fixed order): the weight gradients of a training step (`tensor.grad` of the Linear /
 * recurrent bricks, K = T*B or L*B rows), each too small to fill the chip alone.  ws: workspace for the partials (the more,
 * the finer the k-chunks: sum_i ceil(K_i/1024) * M_i * N_i floats is enough); outputs must not overlap. */
typedef struct lvsr_gemm_desc {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    float beta;
} lvsr_gemm_desc;
int lvsr_sgemm_tn_grouped(void* stream, const lvsr_gemm_desc* descs, int n, float* ws, long long ws_bytes);
/* n independent strided 2-D copies (dst[r*ldd + c] = src[r*lds + c]) in ONE launch per 32 descriptors: the concatenated
 * fork weights of the encoder layers ((I,6H) = [Wi_f | Wg_f | Wi_b | Wg_b], refreshed per step) and the scatter of their
 * gradient back into the four parameters were 8 copy kernels of ~5 us per layer and pass. */
typedef struct lvsr_copy_desc {
    const float* src; float* dst;
    int rows, cols, lds, ldd;
    float beta;                           /* dst = src + beta * dst (0: plain copy, dst is not read) */
} lvsr_copy_desc;
int lvsr_copy2d_many(void* stream, const lvsr_copy_desc* descs, int n);
/* out[n] = beta*out[n] + sum_m X[m*ldx+n]  (bias gradients); ws: optional workspace for the row-split partials */
int lvsr_colsum(void* stream, const float* X, int M, int N, int ldx, float* out, float beta, float* ws,
                long long ws_bytes);
/* n column sums in ONE launch (+ one for the row-split partials) per 32 descriptors: the ~12 bias gradients of a training step were 21
 * launches of 5-13 us.  Every member is split over rows exactly as lvsr_colsum would split it given `split_ws_bytes` of workspace (same
 * partial sums, same order: the same bits); ws holds the partials of all members (sum_i S_i * N_i floats). */
typedef struct lvsr_colsum_desc {
    const float* X; float* out;
    int M, N, ldx;
    float beta;
} lvsr_colsum_desc;
int lvsr_colsum_many(void* stream, const lvsr_colsum_desc* descs, int n, float* ws, long long ws_bytes, long long split_ws_bytes);
int lvsr_transpose(void* stream, const float* in, int rows, int cols, float* out);

/* Pack a (K x N) weight (row-major, leading dim ldw; trans=1: the logical weight is W^T of an (N x K)
 * array) into the operand order the recurrent step kernels stream: 16-column tiles, K split over the 4
 * waves of a work-group, 16 B per lane per load.  lvsr_pack_size = number of floats of the packed copy. */
long long lvsr_pack_size(int K, int N);
int lvsr_pack_b(void* stream, const float* W, int ldw, int K, int N, int trans, float* packed);
/* All packed copies of a model in one call (they are refreshed after every optimiser step): n independent lvsr_pack_b
 * jobs; with use_graph the n launches are captured once per descriptor list and replayed as one hipGraph. */
typedef struct lvsr_pack_desc {
    const float* W;
    float* packed;
    int ldw, K, N, trans;
} lvsr_pack_desc;
int lvsr_pack_b_many(void* stream, const lvsr_pack_desc* descs, int n, int use_graph);

/* ---- encoder: one bidirectional GRU layer ---------------------------------------------------------
 * GatedRecurrent.apply (libs/blocks/blocks/bricks/recurrent.py:608-620) under scan (:178-231),
 * Bidirectional.apply (:655-663), time subsampling of Encoder.apply (lvsr/bricks/__init__.py:71-78). */
typedef struct lvsr_bigru_fwd_args {
    const float* xg;        /* (T,B,6H) input projections: direction d at column 3H*d: [x_in | g_update | g_reset] */
    const float* mask;      /* (T,B) or NULL */
    const float* Whh_p[2];  /* packed state_to_state (K=H,N=H) per direction */
    const float* Whg_p[2];  /* packed state_to_gates (K=H,N=2H) per direction */
    const float* h0[2];     /* initial_state (H) */
    float* y;               /* (T,B,2H) [h_fwd | h_bwd] */
    float* ysub;            /* (ceil(T/sub),B,2H) = y[::sub] when sub>1, else NULL */
    float* u; float* r; float* c; float* rh;   /* (T,B,2H) saved gates / candidate / r*h_prev for BPTT */
    int sub, T, B, H;
    int kernel_mask;        /* 0 or 3: both step kernels; 1 / 2: only the gates / candidate kernel (timing probes) */
    int persistent;         /* 1: one persistent launch for the whole time loop (needs sync_ws; Whh_p / Whg_p are then the
                               PLAIN (H,H) / (H,2H) weights), 0: two kernels per step */
    void* sync_ws;          /* lvsr_bigru_persist_ws_bytes(B,H) bytes of device scratch for the persistent mode */
} lvsr_bigru_fwd_args;
int lvsr_bigru_fwd(void* stream, const lvsr_bigru_fwd_args* a, int use_graph);

typedef struct lvsr_bigru_bwd_args {
    const float* mask; const float* y; const float* u; const float* r; const float* c;
    const float* WhhT_p[2];  /* packed state_to_state^T (K=H,N=H) */
    const float* WhgT_p[2];  /* step kernels: packed [update rows | reset rows] of state_to_gates^T, each (K=H,N=H), reset block at
                                offset lvsr_pack_size(H,H); persistent mode: the plain (H,2H) weight */
    const float* h0[2];
    const float* dy;         /* (ceil(T/sub),B,2H) gradient wrt the (subsampled) layer output */
    float* dxg;              /* (T,B,6H) out: gradient wrt xg */
    float* dh_ws;            /* workspace 12*Bp*H floats, Bp = B rounded up to 16 */
    float* dh0[2];           /* (H) out: gradient wrt initial_state */
    int sub, T, B, H;
    int kernel_mask;         /* 0 or 3: both step kernels; 1 / 2: only kernel A / kernel B (timing probes) */
    int persistent;          /* as in lvsr_bigru_fwd_args */
    void* sync_ws;
} lvsr_bigru_bwd_args;
int lvsr_bigru_bwd(void* stream, const lvsr_bigru_bwd_args* a, int use_graph);
/* Persistent mode (csrc/encoder_persist.hip): a cluster of P = ceil(H/64)^2/4 work-groups (1 at H <= 128, 4 at H <= 256, 16 at
 * H <= 512; one per CU) serves `rows` utterances of one direction; every thread keeps 192 recurrent weights in registers
 * for the whole sequence and the cluster exchanges h / r*h (forward) or dpre_c, dpre_u / dpre_r (backward) once per phase
 * through 8-byte {epoch,value} granules.  lvsr_bigru_persist_ws_bytes returns 0 when (B,H) cannot run persistently (H > 512
 * or the clusters do not fit the chip): use the step kernels then.  lvsr_bigru_persist_rows = utterances per cluster that
 * would be used (the per-step cost grows with it: the host prefers the step kernels above 2).  After the stream has drained,
 * the first int of sync_ws is non-zero if a work-group gave up waiting (results invalid). */
int lvsr_bigru_persist_rows(int B, int H);
long long lvsr_bigru_persist_ws_bytes(int B, int H);

/* ---- attention decoder (teacher forced or one generation step) ------------------------------------
 * AttentionRecurrent.do_apply / take_glimpses / compute_states (libs/blocks/blocks/bricks/attention.py:
 * 589-707), SequenceContentAndConvAttention (lvsr/bricks/attention.py:98-230: window prior, location
 * convolution, energies, masked softmax, paste), content-only SequenceContentAttention
 * (libs/blocks/blocks/bricks/attention.py:346-393; K = 0 here), compute_weighted_averages (:236-256),
 * decoder GatedRecurrent step with Distribute'd glimpse (:625-662; recurrent.py:608-620).
 * One call runs steps [0, L): step i reads state slot i (S, W) and writes slot i+1. */
typedef struct lvsr_attdec_args {
    int Tp, B, L, E, D, M, K, c;          /* K = conv_num_filters (0: content-only attention), c = conv_n */
    int prior_type;                       /* 0 expanding, 1 window_around_mean, 2 window_around_median */
    int step0;                            /* value of the reference's `step` state at slot 0 */
    int phases;                           /* bit0: attention (glimpse) part, bit1: GRU part, of every step; bit2: pos slot 0 is
                                             already filled by the caller (window_around_* priors) */
    int normalizer;                       /* energy_normalizer: 0 softmax, 1 logistic, 2 relu (lvsr/bricks/attention.py:191-213) */
    double p0, p1, p2, p3;                /* expanding: initial_begin, initial_end, f32(min_speed), f32(max_speed); window_around_*: before, after */
    /* contexts; element (t,b,x) at base[t*ts + b*bs + x]; bs = 0 broadcasts one utterance (beam search) */
    const float* A; const float* PA; const float* Am;
    long long A_ts, A_bs, PA_ts, PA_bs, Am_ts, Am_bs;
    /* attention parameters */
    const float* Ws_p;                    /* packed state_trans/transform_states.W (K=D,N=M) */
    const float* w_e;                     /* energy_comp/linear.W (M) */
    const float* e_bias;                  /* energy_comp/linear.b (1) for the non-softmax normalisers, else NULL */
    const float* filters;                 /* conv1d.filters (K,2c+1) */
    const float* handler;                 /* handler.W (K,M) */
    /* decoder GRU parameters (packed) */
    const float* Whg_p;                   /* transition.state_to_gates (K=D,N=2D) */
    const float* Whh_p;                   /* transition.state_to_state (K=D,N=D) */
    const float* Wdi_p;                   /* distribute/fork_inputs.W (K=E,N=D) */
    const float* Wdg_p;                   /* distribute/fork_gate_inputs.W (K=E,N=2D) */
    /* per-step inputs */
    const float* xg;                      /* (L,B,3D) fork(feedback) incl. biases: [x_in | g_update | g_reset] */
    const float* ymask;                   /* (L,B) or NULL */
    /* state slots (slot 0 filled by the caller) */
    float* S;                             /* (L+1,B,D) decoder states */
    float* W;                             /* (L+1,B,Tp) alignments (weights) */
    float* pos;                           /* (L+1,B) window centres derived from W (prior_type 1,2), else NULL */
    /* per-step outputs */
    float* WA;                            /* (L,B,E) weighted_averages */
    float* EN;                            /* (L,B,Tp) energies */
    float* ZB;                            /* (L,B) normalisation constants (saved for backward) */
    /* saved for backward */
    float* sW;                            /* (L,B,M) transformed states */
    float* CV;                            /* (L,B,K,Tp) location-convolution features */
    float* U; float* R; float* C; float* RH;   /* (L,B,D) */
    /* scratch */
    float* sg;                            /* (B,2D) state part of the gate pre-activations */
    float* xin;                           /* (B,D) candidate input */
    float* ep;                            /* (B,ceil(M/32),Tp) partial energies of the match-dim slices */
    const int* step_dev;                  /* optional device word added to step0 (graph-replayed generation: the position
                                             counter of lvsr_beam_select), else NULL */
    int S_ld;                             /* row stride of S in floats (0 = D; a slot is B rows): a layer of a stacked decoder keeps its
                                             states as a column block of the concatenated (L+1,B,n*D) array.  Step kernels only */
    int label0;                           /* lvsr_attdec_fwd / lvsr_attdec_bwd run the steps [label0, L) only (0: all).  A stacked decoder
                                             (RecurrentStack, lvsr/bricks/recognizer.py:250-262) is driven label by label with one
                                             block for the attention over the concatenated states and one per GRU layer */
    /* Row groups (batched beam search, step kernels only: several utterances' beams in one set of launches).  group_rows = n > 0:
       rows [g n, g n + n) are the hypotheses of utterance g — they read context column g (element (t,g,x) at base[t*ts + g*bs + x]),
       the batch-wide window of the window_around_* priors is the window of the GROUP's rows, the position counter of group g is
       step_dev[g * step_stride], and the attended length of group g is group_Tp[g] (<= Tp, NULL: Tp) — the windows are clamped
       to it as they would be for the utterance decoded alone.  0: one group of all rows, the fields above as described. */
    int group_rows, step_stride;
    const int* group_Tp;
    /* skip[g * skip_stride] != 0 (device word per row group; one group when group_rows = 0): the attention part of this call is
       not computed for group g — the caller has put the results (WA, slot 1 of W and pos) there already.  The second attention
       pass of a beam-search position repeats the first one on re-arranged rows; lvsr_beam_select raises the word when the
       re-arranged rows see the same window as the live ones (always, with the expanding prior), and copies the results. NULL: compute. */
    const int* skip; int skip_stride;
} lvsr_attdec_args;
int lvsr_attdec_fwd(void* stream, const lvsr_attdec_args* a, int use_graph);

/* The same label loop as ONE persistent launch (csrc/decoder_persist.hip): a cluster of P work-groups per utterance (P =
 * ceil(D/16) when B * P fits the device's CUs, else ceil(D/32), at D <= 256; ceil(D/16) <= 32 at D <= 512; LVSR_KNOB_DEC_CLUSTER)
 * keeps the decoder's state weights in registers and the utterance's contexts in LDS for the whole sequence and exchanges four
 * phase vectors per label through {epoch,value} granules.  Same semantics, inputs and saved tensors as lvsr_attdec_fwd with
 * phases = 3 (the scratch fields sg / xin / ep are not used), except that
 *   - it takes the PLAIN row-major state weights instead of the packed ones, and AW = attended @ [fork_inputs.W |
 *     fork_gate_inputs.W] (T',B,3D) computed by the caller: the gate inputs are formed as sum_t alpha_t AW[t] (the glimpse
 *     contraction reassociated; equal up to float32 rounding);
 *   - it does NOT write WA: call lvsr_attdec_glimpses afterwards (all labels' weighted averages in one launch).
 * lvsr_attdec_persist_ws_bytes returns the workspace size, or 0 when the configuration is outside the kernel's limits (then
 * call lvsr_attdec_fwd). */
typedef struct lvsr_attdec_plain {
    const float* Ws;                      /* transform_states.W (D,M) */
    const float* Whg;                     /* transition.state_to_gates (D,2D) */
    const float* Whh;                     /* transition.state_to_state (D,D) */
    const float* AW;                      /* (T',B,3D) attended @ [distribute/fork_inputs.W | distribute/fork_gate_inputs.W] */
    int AW_ld;                            /* row stride of AW in floats (0 = 3D).  lvsr_attdec_bwd_persistent reads the rows 16 bytes at a
                                             time: it needs AW_ld % 4 == 0 (pad the rows when 3D is not a multiple of 4, e.g. D = 250) */
} lvsr_attdec_plain;
long long lvsr_attdec_persist_ws_bytes(const lvsr_attdec_args* a);
int lvsr_attdec_fwd_persistent(void* stream, const lvsr_attdec_args* a, const lvsr_attdec_plain* w, void* ws, int use_graph);
/* The label loop of a TWO-layer stacked decoder (RecurrentStack with skip connections, lvsr/bricks/recognizer.py:250-262;
 * libs/blocks/blocks/bricks/recurrent.py:677-950) as one persistent launch: per utterance one cluster of 8 work-groups runs the
 * attention and layer 0 exactly as lvsr_attdec_fwd_persistent does, a second cluster of 8 runs layer 1 — glimpse part of its gate
 * inputs from AW1 and the alignment (it repeats the softmax), fork of the NEW state of layer 0, its own recurrence — and hands the
 * layer-1 part of the transformed state (s1 . transform_states#1) back for the next label's energies.  `a` / `w`: the block of
 * layer 0 (D = width of ONE layer; S = (L+1, B, 2D) slots with S_ld = 2D, layer l in columns [l D, (l+1) D); xg, U, R, C, RH, the
 * plain weights and AW of layer 0); sW receives the transformed state of BOTH layers.  WA: lvsr_attdec_glimpses afterwards. */
typedef struct lvsr_attdec_stack2 {
    const float* Whg1; const float* Whh1; /* transition#1 state_to_gates (D,2D), state_to_state (D,D) */
    const float* Ws1;                     /* state_trans/transform_states#1.W (D,M) */
    const float* F1;                      /* recurrentstack/fork_1: [fork_inputs.W | fork_gate_inputs.W] (D,3D), row stride F1_ld */
    const float* AW1;                     /* (T',B,3D) attended @ [distribute/fork_inputs#1.W | fork_gate_inputs#1.W], row stride AW1_ld */
    const float* xg1;                     /* (L,B,3D) fork#1(feedback) incl. biases */
    float* U1; float* R1; float* C1; float* RH1;      /* (L,B,D) saved for the backward pass */
    int F1_ld, AW1_ld;                    /* 0 = 3D */
    float* DXG1;                          /* lvsr_attdec_bwd_persistent_stack2 only: (L,B,3D) out, gradient wrt layer 1's [x_in | gate_in] pre-activations */
} lvsr_attdec_stack2;
long long lvsr_attdec_stack2_persist_ws_bytes(const lvsr_attdec_args* a);      /* 0 = not available (then: the step kernels) */
int lvsr_attdec_fwd_persistent_stack2(void* stream, const lvsr_attdec_args* a, const lvsr_attdec_plain* w,
                                      const lvsr_attdec_stack2* l1, void* ws, int use_graph);
/* WA[l,b,:] = sum_t W[l+1,b,t] * A[t,b,:] for l in [0,L) (compute_weighted_averages, libs/blocks/blocks/bricks/attention.py:236-256) */
int lvsr_attdec_glimpses(void* stream, const lvsr_attdec_args* a);

/* Backward of lvsr_attdec_fwd (what theano.grad derives through the decoder scan,
 * libs/blocks/blocks/algorithms/__init__.py:216-224).  Walks the steps in reverse; per-step tensors needed
 * for the weight gradients (DXG, DWA, DSW, DCV) are left for batched GEMMs by the caller. */
typedef struct lvsr_attdec_bwd_args {
    lvsr_attdec_args f;                   /* the forward argument block (same buffers, contiguous contexts) */
    const float* WhhT_p;                  /* packed state_to_state^T (K=D,N=D) */
    const float* WhgT_p;                  /* packed state_to_gates^T (K=2D,N=D) */
    const float* WdT_p;                   /* packed [distribute/fork_inputs.W | fork_gate_inputs.W]^T (K=3D,N=E) */
    const float* WsT_p;                   /* packed transform_states.W^T (K=M,N=D) */
    const float* dWA_r;                   /* (L,B,E) readout gradient wrt weighted_averages, or NULL */
    const float* dS_r;                    /* (L,B,D) readout gradient wrt state slot i, or NULL */
    float* DXG;                           /* (L,B,3D) out: gradient wrt [x_in | gate_in] pre-activations */
    float* DWA;                           /* (L,B,E) out: total gradient wrt weighted_averages */
    float* DSW;                           /* (L,B,M) out: gradient wrt transformed states */
    float* DCV;                           /* (L,B,K,Tp) out: gradient wrt convolution features */
    float* dPA;                           /* (Tp,B,M) in/out: accumulated gradient wrt preprocessed attended (caller zeroes) */
    float* accH;                          /* (B*ntile, K*M) in/out: per-work-group handler.W gradient partials (caller zeroes); ntile = ceil(Tp/64) */
    float* accWe;                         /* (B*ntile, M) in/out: per-work-group energy vector gradient partials (caller zeroes) */
    float* accEb;                         /* (B*ntile) in/out: energy-bias gradient partials (caller zeroes) */
    float* ds;                            /* (B,D) in/out: running gradient wrt the state (caller zeroes; ends as grad wrt slot 0) */
    float* dalp;                          /* (B,K,Tp) in/out: running gradient wrt the alignment, one row per filter (caller zeroes) */
    float* dspart; float* dsacc;          /* (B,D) scratch */
    float* Q;                             /* (B,Tp) scratch */
    float* dcvp;                          /* (B,ceil(M/32),K,Tp) scratch: per-slice partials of DCV */
    float* dswp;                          /* (B,ntile,M) scratch: per-tile partials of DSW */
    /* optional (both or neither): the reassociated glimpse — q[b,t] = DXG[i][b,:] . AW[t,b,:] + QR[i,b,t] + ..., one launch less
     * per label; DWA is then NOT written: the caller forms DWA = DXG @ [Wdi|Wdg]^T + dWA_r for all labels after the call */
    const float* AW;                      /* (Tp,B,3D) attended @ [fork_inputs.W | fork_gate_inputs.W] */
    const float* QR;                      /* (L,B,Tp) dWA_r[i,b,:] . A[t,b,:] (zeros if dWA_r is NULL) */
    int AW_ld;                            /* row stride of AW in floats (0 = 3D) */
    int ds_ld;                            /* row stride of ds and dsacc in floats (0 = D).  Step kernels only */
    int parts;                            /* 0 or 3: the whole step; 1: the GRU kernels only (ds -> DXG, DWA, dspart, dsacc; the forward
                                             block may then have phases = 2); 2: the attention kernels only (DWA, dalp, dsacc -> ds,
                                             DSW, DCV, dPA, ...; phases = 1).  With parts != 3 the reassociated glimpse (AW / QR) is
                                             not available */
} lvsr_attdec_bwd_args;
int lvsr_attdec_bwd(void* stream, const lvsr_attdec_bwd_args* a, int use_graph);
/* The backward walk as ONE persistent launch (csrc/decoder_persist_bwd.hip; same cluster layout and limits as
 * lvsr_attdec_fwd_persistent).  Takes the plain weights and AW of lvsr_attdec_plain, and in the argument block: QR (required),
 * dS_r, DXG, DSW, DCV, dPA, ds as lvsr_attdec_bwd; accH / accWe / accEb hold ONE ROW PER WORK-GROUP here — (B*P, K*M), (B*P, M),
 * (B*P) with P = lvsr_attdec_bwd_persist_clusters(args), the work-groups per utterance — written, not accumulated.  DWA is not written (see AW / QR above); dalp, dspart, dsacc, Q, dcvp,
 * dswp and the packed weights are not used.  lvsr_attdec_bwd_persist_ws_bytes: workspace size, 0 = not available. */
long long lvsr_attdec_bwd_persist_ws_bytes(const lvsr_attdec_args* a);
int lvsr_attdec_bwd_persist_clusters(const lvsr_attdec_args* a);      /* work-groups per utterance the launch will use (0 = not available) */
int lvsr_attdec_bwd_persistent(void* stream, const lvsr_attdec_bwd_args* a, const lvsr_attdec_plain* w, void* ws);
/* The reverse walk of the two-layer launch (csrc/decoder_persist_bwd.hip): the attention + layer-0 cluster works as
 * lvsr_attdec_bwd_persistent does, the layer-1 cluster runs that layer's GRU backward, hands the fork_1 part of layer 0's state
 * gradient and its share of the glimpse gradient (DXG1 . AW1) to the main cluster and takes its transform_states#1 part from the
 * main cluster's dsW.  `a`: the block of layer 0 as for the forward launch (f.S_ld = 2D); dS_r (L,B,2D) and ds (B,2D, ds_ld = 2D)
 * hold both layers side by side; QR as for lvsr_attdec_bwd_persistent; DXG / l1->DXG1 out. */
long long lvsr_attdec_stack2_bwd_persist_ws_bytes(const lvsr_attdec_args* a);
int lvsr_attdec_stack2_bwd_persist_clusters(const lvsr_attdec_args* a);       /* work-groups of the attention cluster = rows of accH / accWe / accEb per utterance */
int lvsr_attdec_bwd_persistent_stack2(void* stream, const lvsr_attdec_bwd_args* a, const lvsr_attdec_plain* w,
                                      const lvsr_attdec_stack2* l1, void* ws);
/* gradient wrt conv1d.filters (K,2c+1) from DCV and the alignment slots of the forward block; ws: scratch of at least
 * ceil(L*B / R) * K * (2c+1) floats with R = min(4, 8192 / (K*Tp)) rows per work-group (L*B*K*(2c+1) floats always suffice) */
int lvsr_attdec_filter_grad(void* stream, const lvsr_attdec_args* f, const float* DCV, float* dfilters, float* ws,
                            long long ws_bytes);

/* ---- feedback lookup, post-merge activation, softmax emitter ----------------------------------------
 * LookupTable.apply (libs/blocks/blocks/bricks/lookup.py:48-68) / OneOfNFeedback (lvsr/bricks/__init__.py:
 * 97-104); Maxout/Rectifier/Tanh (libs/blocks/blocks/bricks/simple.py:161-207); Softmax.log_probabilities,
 * categorical_cross_entropy (:315-371); SoftmaxEmitter.cost/costs (sequence_generators.py:780-791). */
/* out[r,:] = table[idx[r],:] + bias  (rows with idx outside [0,nrows) read as zero) */
int lvsr_gather_rows(void* stream, const float* table, int ldt, const long long* idx, int n, int nrows, int width,
                     const float* bias, float* out, int ldo);
/* dst[v,:] = beta*dst[v,:] + sum_{r: idx[r]==v} src[r,:]  (deterministic) */
int lvsr_scatter_add_rows(void* stream, const float* src, int lds, const long long* idx, int n, int nrows, int width,
                          float* dst, int ldd, float beta);
/* kind: 0 identity, 1 Maxout(2), 2 Rectifier, 3 Tanh; x (n,P) -> y (n,P or P/2) */
int lvsr_act_fwd(void* stream, int kind, const float* x, int ldx, int n, int P, float* y, int ldy);
int lvsr_act_bwd(void* stream, int kind, const float* x, int ldx, const float* dy, int lddy, int n, int P, float* dx,
                 int lddx);
/* per row: cost = -log_softmax(logits)[label]*mask; optional dlogits = (softmax-onehot)*mask*scale;
 * optional neglogp = -log_softmax for every class (beam search `costs`) */
int lvsr_softmax_nll(void* stream, const float* logits, int ld, const long long* labels, const float* mask, int n, int V,
                     float* cost, float* dlogits, int ldd, float scale, float* neglogp, int ldn);

/* ---- optimiser step on the flat buffers -----------------------------------------------------------
 * StepClipping -> Momentum(scale) -> AdaDelta -> Restrict(VariableClipping(axis=0), WEIGHT params) ->
 * RemoveNotFinite -> [BurnIn] -> parameter -= step  (lvsr/main.py:480-519; libs/blocks/blocks/algorithms/__init__.py:
 * 378-515, 610-720, 829-893).  `grad` holds d(sum cost); grad_scale = 1/batch_size (lvsr/main.py:340-345). */
typedef struct lvsr_opt_args {
    float* param; const float* grad;      /* flat (n) */
    float* velocity; float* ms_step; float* ms_dx;   /* flat (n) rule state (NULL when the rule is off) */
    float* step;                          /* flat (n) out: the applied step (before RemoveNotFinite) */
    const long long* segments;            /* (nseg,4): offset, rows, cols, flags (bit0: max-norm applies) */
    int* segflag;                         /* (nseg) scratch / out: 1 = step was not finite */
    float* scratch;                       /* (2+256): [0] gradient norm, [1] clip multiplier, partial sums */
    long long n;
    int nseg, max_cols;
    int use_momentum, use_adadelta, remove_not_finite, pad0;
    float grad_scale, clip_threshold, learning_rate, momentum, decay_rate, epsilon, max_norm, nonfinite_scaler;
    /* Device-resident schedule state (NULL = off), so that a replayed step graph needs no host round trip:
     *   [0] StepClipping threshold in force, [1] mean log gradient norm, [2] mean squared log norm, [3] iterations done
     *       -- AdaptiveClipping.after_batch (lvsr/extensions.py:64-91; wired with decay 0.998, burn-in 500,
     *       lvsr/main.py:616-619), evaluated on the device right after the norm of this step is known;
     *   [4] BurnIn steps left (lvsr/algorithms.py:19-43): while > 0 the applied step is zero, rule state still updates.
     * The caller initialises [0] = clip_threshold, [1..3] = 0, [4] = burn_in_steps. */
    double* clip_state;
    int adaptive_clipping, adaptive_burnin;
    float adaptive_decay, pad1;
    /* Guard (NULL = off): a device float that must be 0 for the step to be applied.  Non-zero — a persistent cluster kernel of this
     * step gave up waiting (lvsr_guard_collect below; under data parallelism the word rides in front of the gradient bucket, so
     * after the all-reduce every rank sees the same value) — and the whole step is skipped on the device: parameters, rule state,
     * adaptive-clipping statistics and burn-in counter stay as they are; scratch[3] = 1 reports it (else 0). */
    const float* guard;
} lvsr_opt_args;
int lvsr_opt_step(void* stream, const lvsr_opt_args* a);
/* *out = number of non-zero words among words[0..n) (n <= 16 device int pointers; NULL entries are skipped): the abort words of the
 * persistent cluster launches of a step (first int of their workspaces), collected behind the backward pass. */
int lvsr_guard_collect(void* stream, const int* const* words, int n, float* out);

/* Generation-time readout + emitter of n rows in one launch (one work-group per row):
 * Readout.readout (libs/blocks/blocks/bricks/sequence_generators.py:614-619) with the post-merge stack of
 * lvsr/bricks/recognizer.py:298-320, then SoftmaxEmitter.costs (:788-791) or, with lm_add, ShallowFusionReadout + LMEmitter
 * (lvsr/bricks/language_models.py:92-184), and optionally SoftmaxEmitter.emit (:770-776): the class drawn by inverse CDF
 * from uniforms[r] (Theano's MultinomialFromUniform rule) and its cost. */
typedef struct lvsr_readout_step_args {
    const float* S; const float* WA;      /* (n,D) states (row stride lds), (n,E) weighted averages (row stride ldwa) */
    int lds, ldwa, n, D, E, P, V, act;    /* P = merge width; act: 0 identity, 1 Maxout(2), 2 Rectifier, 3 Tanh */
    const float* Wms;                     /* merge/transform_states.W (D,P) or NULL (use_states_for_readout = False) */
    const float* Wmw;                     /* merge/transform_weighted_averages.W (E,P) */
    const float* bias1;                   /* post_merge/bias.b (P), or readout/bias.b when there is no post-merge layer */
    const float* Wout; const float* bout; /* post_merge/mlp/linear_0.{W (P or P/2,V), b (V)} or NULL (then P == V) */
    const float* lm_add;                  /* (n,V) language-model look-ahead costs or NULL */
    float am_beta, lm_weight;
    int norm_am, norm_lm, norm_tot;
    float* neglogp;                       /* (n,V) out: step costs of every class, or NULL */
    float* logits;                        /* (n,V) out, or NULL */
    const float* uniforms;                /* (n) in [0,1): emit, or NULL */
    long long* outputs; float* costs;     /* (n) out: emitted class and its cost (with uniforms) */
    /* further post-merge layers (post_merge_dims of more than one entry, lvsr/bricks/recognizer.py:309-317): layer h maps the
     * previous width to dimh[h] (Wh[h] (prev, dimh[h]), bh[h]) and is followed by `act` (2 or 3 only); Wout then has dimh[last] rows */
    int n_hidden; int dimh[3];
    const float* Wh[3]; const float* bh[3];
    /* R1 != NULL: the merged pre-activations (n,P) = S @ Wms + WA @ Wmw + bias1 have been computed by lvsr_readout_merge (row
     * stride ldr1); the per-row products above are skipped (S, WA, Wms, Wmw are not read) */
    const float* R1; int ldr1;
} lvsr_readout_step_args;
int lvsr_readout_step(void* stream, const lvsr_readout_step_args* a);
/* The merge of lvsr_readout_step for MANY rows (batched beam search: n = searches x beam): R1 (n,P) = S @ Wms + WA @ Wmw + bias1
 * as 16-row MFMA tiles over PACKED weights (lvsr_pack_b of the (D,P) / (E,P) matrices; Wms_p NULL: no state part) — the
 * per-row kernel reads the 0.75 MB of merge weights once per ROW out of L2 (24 us at 512 rows, L2-bandwidth bound). */
int lvsr_readout_merge(void* stream, const float* S, int lds, const float* WA, int ldwa, int n, int D, int E, int P,
                       const float* Wms_p, const float* Wmw_p, const float* bias1, float* R1, int ldr1);
/* SoftmaxEmitter.emit + cost on given readouts (n,V): outputs[r] = class drawn at uniforms[r], costs[r] = -log p (or NULL) */
int lvsr_softmax_emit(void* stream, const float* logits, int ld, const float* uniforms, int n, int V, long long* outputs,
                      float* costs);

/* ---- beam search on the device ------------------------------------------------------------------------
 * The candidate selection, stopping rules and bookkeeping of BeamSearch.search (libs/blocks/blocks/search.py:244-407)
 * as one launch per emitted character; see csrc/beam.hip for the rules that are kept.  State words:
 *   ctl[0] live hypotheses, [1] position, [2] done (0 running, 1 stopping rule, 2 beam empty, 3 max_length), [3] finished
 *   hypotheses, [4] patience left (-1 = unassigned), [5] rows selected by the last step, [6] error (1 non-finite step cost,
 *   2 finished list full, 3 patience used before assignment = the reference's UnboundLocalError), [7] steps executed;
 *   fctl[0] best finished score seen (patience rule; the caller initialises it to 1000), fctl[2..3] the same as a double
 *   (maintained by the kernel; scores are compared in double like the reference's float32 - Python-float arithmetic).
 * The caller zeroes ctl (ctl[0] = 1, ctl[4] = -1), running[0] = 0, live_col[0] = 0 and fills row 0 of the live buffers. */
typedef struct lvsr_beam_args {
    int K, V, eol, ignore_first_eol;      /* beam size, characters, <eol> label, keep <eol> hypotheses alive at position 0 */
    int stop_on;                          /* 0 patience, 1 optimistic_future_cost */
    int max_length, fin_cap, D, Tp;       /* positions, capacity of the finished list (>= 2K), state / alignment widths */
    float round_to_inf;
    double char_discount;
    int* ctl; float* fctl;
    const float* neglogp;                 /* (K,V) step costs of the live hypotheses (rows >= ctl[0] ignored) */
    float* running; int* live_col;        /* (K) cumulative cost / history column of live hypothesis i */
    int* hist_parent; int* hist_char; float* hist_cost;     /* (max_length,K) back-pointers: parent column, character, cost */
    int* fin_pos; int* fin_col; float* fin_cost; float* fin_score;     /* (fin_cap) finished hypotheses */
    int* keep;                            /* (K) out: selected row that becomes live row i (rows >= new ctl[0]: keep[0]) */
    long long* chars;                     /* (K) out: chosen characters = feedback labels of the next-state pass */
    int* parents;                         /* (K) out: live row each chosen candidate extends */
    /* rows of the next-state pass: sel[k] = live[parent of candidate k] (rows >= ctl[5] replicate candidate 0) */
    const float* S_live; const float* W_live; float* S_sel; float* W_sel;          /* (K,D) / (K,Tp) */
    const long long* lm_states_live; const double* lm_weights_live;                /* (K,7) or NULL */
    long long* lm_states_sel; double* lm_weights_sel;
    const float* pos_live; float* pos_sel;                                         /* (K) window centres or NULL */
    /* one-hot feedback (embed_outputs = False): fork inputs of the chosen characters, xg[k] = [Wi[c]+bi | Wg[c]+bg];
     * fork_xg = NULL when the caller computes them itself (lookup feedback) */
    float* fork_xg; const float* fork_Wi; const float* fork_Wg; const float* fork_bi; const float* fork_bg; int fork_rows;
    /* lvsr_beam_compact: live row i <- row keep[i] of the next-state pass */
    const float* pos_new; float* pos_live_out;
    const float* S_new; const float* W_new; float* S_live_out; float* W_live_out;
    const long long* lm_states_new; const double* lm_weights_new; const float* lm_add_new;      /* or NULL */
    long long* lm_states_live_out; double* lm_weights_live_out; float* lm_add_live_out;
    /* groups = G > 1: G independent searches (utterances) advanced by the same launches.  Every buffer above is then G consecutive
     * blocks of the shape given (ctl: 16 words per search, fctl: 4; rows of the state buffers: search g owns rows [g K, g K + K));
     * row indices (parents, keep, live_col, ...) stay local to their search.  The position limit of search g is ctl[16 g + 8]
     * (max_length above is the capacity of the history buffers). 0 / 1: one search. */
    int groups;
    /* Reuse of the first attention pass (optional, all NULL: off).  The next-state pass recomputes the glimpses of the selected
     * rows because the window of the location prior depends on the batch it is computed for (lvsr/bricks/attention.py:133-157); when
     * the selected rows span the same window centres as the live rows (min and max of pos equal; no pos: expanding prior, always)
     * every one of those glimpses equals the one its parent got in the first pass: lvsr_beam_select then sets ctl[9] = 1 (else 0)
     * and copies, row k <- row parents[k]: WA_sel <- WA_live (K,E), W1_sel <- W1_live (K,Tp: the NEW alignments), pos1_sel <-
     * pos1_live.  The caller's second lvsr_attdec_fwd takes skip = ctl + 9 (lvsr_attdec_args.skip). */
    const float* WA_live; float* WA_sel; const float* W1_live; float* W1_sel; const float* pos1_live; float* pos1_sel; int E;
} lvsr_beam_args;
/* two launches: the selection kernel (one work-group) and the row gather / feedback fork of the K chosen candidates */
int lvsr_beam_select(void* stream, const lvsr_beam_args* a);
int lvsr_beam_compact(void* stream, const lvsr_beam_args* a);
/* `_smallest` (libs/blocks/blocks/search.py:221-242) alone: the k smallest of costs[0..n) ascending, equal values in index
 * order; idx (k) / val (k) on the device.  n <= 8192, k <= 256. */
int lvsr_topk_smallest(void* stream, const float* costs, int n, int k, long long* idx, float* val);

/* ShallowFusionReadout.readout (lvsr/bricks/language_models.py:92-104):
 * out = [logsoftmax](am_beta*am) + lm_weight*[logsoftmax](-lm_add) [-> logsoftmax]; with an LM the emitter is
 * LMEmitter (costs = -readout, language_models.py:147-184): out_scale = -1 yields the beam-search costs directly. */
int lvsr_shallow_fusion(void* stream, const float* am, int ld, const float* lm_add, int n, int V, float am_beta,
                        float lm_weight, int norm_am, int norm_lm, int norm_tot, float out_scale, float* out);
/* LMEmitter.cost (lvsr/bricks/language_models.py:165-168) of teacher-forced labels on fused readouts x (n, V):
 * cost[r] = scale * x[r, labels[r]] * mask[r] (scale = -1; mask may be NULL). */
int lvsr_select_cost(void* stream, const float* x, int ld, const long long* labels, const float* mask, int n, int V, float scale,
                     float* cost);

/* ---- FST language model on the device ---------------------------------------------------
 * Replaces the per-hypothesis Python walks of FSTTransitionOp.perform / FSTCostsOp.perform (lvsr/ops.py:147-169,
 * 206-225; FST.transition / FST.expand ops.py:66-97).  The automaton is a CSR arc table in device memory: labelled
 * arcs sorted by (source state, input label), epsilon arcs (label 0, ops.py:22) in their own CSR, `topo` = rank of
 * every state in a topological order of the epsilon sub-graph, `remap` = network character -> FST input label
 * (language_models.py:118-121), weights = -log probabilities in f64 (the reference's Python floats). */
typedef struct lvsr_fst {
    const int* arc_off;                   /* (num_states+1) */
    const int* arc_lab;                   /* (num_arcs) ascending within a state */
    const int* arc_dst;
    const double* arc_w;
    const int* eps_off;                   /* (num_states+1) */
    const int* eps_dst;
    const double* eps_w;
    const int* topo;                      /* (num_states) */
    const int* remap;                     /* (V), -1 = character has no label */
    int num_states, V;
    double no_transition_cost;
} lvsr_fst;
/* states (n,7) int64 padded with -1, weights (n,7) f64 padded with 0 (MAX_STATES = 7, ops.py:23,131-145).
 * outputs != NULL: new_states/new_weights <- epsilon-closure(transition(states, remap[outputs[b]])), then `add` (n,V) =
 * look-ahead costs of the NEW sets; outputs == NULL: `add` = costs of `states` as given (FSTTransition.initial_states,
 * language_models.py:52-62).  `add` may be NULL.  *err (device int, caller zeroes it) is raised to 1 when a new set has
 * more than 7 states (the reference's ValueError), 2 when a candidate set outgrows the kernel's capacity of 16,
 * 3 when a chosen character has no FST label. */
int lvsr_fst_lm_step(void* stream, const lvsr_fst* f, const long long* states, const double* weights,
                     const long long* outputs, int n, long long* new_states, double* new_weights, float* add, int* err);
/* The same for the rows of several beam searches side by side (BeamSearch.search_batch): row b belongs to search b / group_rows,
 * whose control block is ctl[16 * g ...]; the rows of a search whose `done` word (ctl word 2) is set are left alone — they carry the
 * stale characters of its last position, and a walk over them must not raise *err for the searches still running. */
int lvsr_fst_lm_step_groups(void* stream, const lvsr_fst* f, const long long* states, const double* weights,
                            const long long* outputs, int n, long long* new_states, double* new_weights, float* add, int* err,
                            const int* ctl, int group_rows);

/* ---- mel-filterbank front end ------------------------------------------------------------------------
 * The reference runs Kaldi offline (exp/wsj/write_hdf_dataset.sh:94-104: compute-fbank-feats --use-energy=true
 * --num-mel-bins=40 | add-deltas, then global CMVN); Kaldi's source is not part of the reference tree, so these
 * entry points follow Kaldi's documented defaults; the log-mel columns are pinned to an independent Kaldi-compatible
 * implementation (tests/golden/fbank_hf_kaldi.npz), not to Kaldi's own binary (oracle/fbank_oracle.py). */
typedef struct lvsr_fbank_cfg {
    int frame_length, frame_shift;        /* samples (400, 160 at 16 kHz) */
    int num_mel, use_energy, remove_dc, pad0;
    float preemph, pad1;
} lvsr_fbank_cfg;
int lvsr_fbank_num_frames(long long nsamp, const lvsr_fbank_cfg* cfg);
/* wav: int16 PCM; window (frame_length); melw (num_mel,256) dense filter weights; twiddle (2,512) cos|sin table;
 * out (nframes, num_mel + use_energy), energy first */
int lvsr_fbank(void* stream, const short* wav, long long nsamp, const lvsr_fbank_cfg* cfg, const float* window,
               const float* melw, const float* twiddle, float* out);
/* feats (T,dim) -> out (T,3*dim) = [static | delta | delta-delta], optionally (x-mean)*istd with (3*dim) stats */
int lvsr_add_deltas_cmvn(void* stream, const float* feats, int T, int dim, const float* mean, const float* istd, float* out);
/* The same front end for a SET of utterances in one launch (the streaming form: 320 bytes of new PCM in, 164 bytes of features out per
 * frame): `wav` holds the utterances' PCM back to back, utterance u = samples [wav_off[u], wav_off[u+1]) and output rows
 * [frame_off[u], frame_off[u+1]) (device arrays of n + 1 entries; frame_off[u+1] - frame_off[u] = lvsr_fbank_num_frames of its
 * length).  One wave per PAIR of frames (the samples are real: one 512-point complex FFT in registers / LDS transforms two frames)
 * instead of the direct DFT; the mel filters come as n_items <= 64 work items (filter, chunk of 16 bins starting at a multiple of 4):
 * item_bin (n_items) first bin of an item, item_w (n_items, 16) its weights (zero outside the filter), item_first (num_mel + 1): the
 * items of filter j are [item_first[j], item_first[j + 1]) — the 40-filter front end of the recipe is 53 items.  num_mel <= 64.
 * Same arithmetic up to float32 summation order. */
int lvsr_fbank_batch(void* stream, const short* wav, const long long* wav_off, const int* frame_off, int n, int total_frames,
                     const lvsr_fbank_cfg* cfg, const float* window, const int* item_bin, const int* item_first, const float* item_w,
                     int n_items, const float* twiddle, float* out);
/* deltas + CMVN over the same set: edge frames replicated per utterance */
int lvsr_add_deltas_cmvn_batch(void* stream, const float* feats, const int* frame_off, int n, int total_frames, int dim, const float* mean,
                               const float* istd, float* out);

#ifdef __cplusplus
}
#endif
#endif
