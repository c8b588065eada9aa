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

/* ---- dense helpers (Linear bricks: libs/blocks/blocks/bricks/simple.py:59-76) ------------------ */
/* C[M,N] = alpha*op(A)[M,K]*op(B)[K,N] + beta*C + bias[N]; fp32 MFMA; ws: optional split-K workspace */
int lvsr_sgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
               const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
               long long ws_bytes);
/* out[n] = beta*out[n] + sum_m X[m*ldx+n]  (bias gradients) */
int lvsr_colsum(void* stream, const float* X, int M, int N, int ldx, float* out, float beta);
int lvsr_transpose(void* stream, const float* in, int rows, int cols, float* out);

/* Pack a (K x N) weight (row-major, leading dim ldw; trans=1: the logical weight is W^T of an (N x K)
 * array) into the operand order the recurrent step kernels stream: 16-column tiles, K split over the 4
 * waves of a work-group, 16 B per lane per load.  lvsr_pack_size = number of floats of the packed copy. */
long long lvsr_pack_size(int K, int N);
int lvsr_pack_b(void* stream, const float* W, int ldw, int K, int N, int trans, float* packed);

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
} lvsr_bigru_fwd_args;
int lvsr_bigru_fwd(void* stream, const lvsr_bigru_fwd_args* a, int use_graph);

typedef struct lvsr_bigru_bwd_args {
    const float* mask; const float* y; const float* u; const float* r; const float* c;
    const float* WhhT_p[2];  /* packed state_to_state^T (K=H,N=H) */
    const float* WhgT_p[2];  /* packed state_to_gates^T (K=2H,N=H) */
    const float* h0[2];
    const float* dy;         /* (ceil(T/sub),B,2H) gradient wrt the (subsampled) layer output */
    float* dxg;              /* (T,B,6H) out: gradient wrt xg */
    float* dh_ws;            /* workspace 4*Bp*H floats, Bp = B rounded up to 16 */
    float* dh0[2];           /* (H) out: gradient wrt initial_state */
    int sub, T, B, H;
} lvsr_bigru_bwd_args;
int lvsr_bigru_bwd(void* stream, const lvsr_bigru_bwd_args* a, int use_graph);

#ifdef __cplusplus
}
#endif
#endif
