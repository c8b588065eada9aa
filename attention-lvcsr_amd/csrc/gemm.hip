// fp32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32 products, f32 accumulate).
// Used for the dense, non-recurrent contractions of the path: encoder input forks (K1), attended
// preprocess (K4), readout merge (K10) and all their weight/input gradients.
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N]
//
// 64x64x16 work-group tile, 4 waves, each wave a 32x32 sub-tile = 2x2 MFMA 16x16 tiles; operands
// staged k-major in LDS so every MFMA operand read is a conflict-free row of 16 consecutive floats.
// Optional deterministic split-K (partials in a caller-provided workspace, fixed-order reduction)
// for the tall-skinny weight-gradient shapes (K = T*B rows).
#include "common.h"
#include "lvsr_hip.h"
#include "graph_cache.h"

#define BM 64
#define BN 64
#define BK 16
#define LDS_LD (BM + 4)

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, transA, transB;
    float alpha, beta;
    int ksplit;        // number of K splits (grid.z); >1 => write raw partials to `part`
    int kchunk;        // K elements per split (multiple of BK)
    float* part;       // (ksplit, M, N) workspace
    int batch;         // > 1: grid.z indexes independent problems (no split-K), operands advance by sA/sB/sC elements
    long long sA, sB, sC;
};

__global__ __launch_bounds__(256) void lvsr_sgemm_kernel(GemmArgs g) {
    __shared__ float As[BK][LDS_LD];
    __shared__ float Bs[BK][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    int zsplit = blockIdx.z;
    if (g.batch > 1) {
        g.A += blockIdx.z * g.sA; g.B += blockIdx.z * g.sB; g.C += blockIdx.z * g.sC;
        zsplit = 0;
    }
    const int kbeg = zsplit * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int rowbase = (wave >> 1) * 32, colbase = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // ---- stage A tile (BM x BK) -> As[k][m]; thread mapping follows the contiguous memory dim
        if (!g.transA) {            // A[m*lda + k]: k contiguous
            const int kk = tid & 15, mm = tid >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = m0 + mm + p * 16, k = k0 + kk;
                As[kk][mm + p * 16] = (m < g.M && k < kend) ? g.A[(size_t)m * g.lda + k] : 0.f;
            }
        } else {                    // A[k*lda + m]: m contiguous
            const int mm = tid & 63, kk = tid >> 6;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = m0 + mm, k = k0 + kk + p * 4;
                As[kk + p * 4][mm] = (m < g.M && k < kend) ? g.A[(size_t)k * g.lda + m] : 0.f;
            }
        }
        if (!g.transB) {            // B[k*ldb + n]: n contiguous
            const int nn = tid & 63, kk = tid >> 6;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int n = n0 + nn, k = k0 + kk + p * 4;
                Bs[kk + p * 4][nn] = (n < g.N && k < kend) ? g.B[(size_t)k * g.ldb + n] : 0.f;
            }
        } else {                    // B[n*ldb + k]: k contiguous
            const int kk = tid & 15, nn = tid >> 4;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int n = n0 + nn + p * 16, k = k0 + kk;
                Bs[kk][nn + p * 16] = (n < g.N && k < kend) ? g.B[(size_t)n * g.ldb + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
            const int kr = ks + (lane >> 4), li = lane & 15;
            float a[2], b[2];
            a[0] = As[kr][rowbase + li];
            a[1] = As[kr][rowbase + 16 + li];
            b[0] = Bs[kr][colbase + li];
            b[1] = Bs[kr][colbase + 16 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- epilogue: D layout col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + rowbase + i * 16 + (lane >> 4) * 4 + r;
                const int n = n0 + colbase + j * 16 + (lane & 15);
                if (m < g.M && n < g.N) {
                    if (g.ksplit > 1) {
                        g.part[((size_t)zsplit * g.M + m) * g.N + n] = acc[i][j][r];
                    } else {
                        float v = g.alpha * acc[i][j][r];
                        if (g.beta != 0.f) v += g.beta * g.C[(size_t)m * g.ldc + n];
                        if (g.bias) v += g.bias[n];
                        g.C[(size_t)m * g.ldc + n] = v;
                    }
                }
            }
}

// ---- 128x128x32 tiles, v_mfma_f32_32x32x2_f32 (the 64x64x32 tiles: sgemm_tile_kmajor below) ---------------------------------------
// Used when the output is at least one full tile: 4 waves, each a 64x64 sub-tile = 2x2 MFMA 32x32 blocks (64 accumulator
// registers).  Round 6 layout (tools/probes/mfma_rate_probe.hip, profiles/r06_gemm_probe.md): the LDS image of BOTH operands is
// [row][k] with k contiguous (row = m for A, n for B; 36-float rows), and a lane fetches its MFMA operands 16 bytes at a time —
// four k of its row per ds_read_b128, 16 LDS reads per k-tile instead of the 64 ds_read2_b32 of the round-1..5 image ([k][row],
// one k per read), which alone held one wave per SIMD at 0.85 of the MFMA rate (4 831 cycles per 64 MFMAs; 4 270 with b128 reads).
// The instruction contracts k = lane / 32 per step: the columns of a row are permuted (gemm2_col) so that a lane's four values are
// its k of four consecutive steps and the k loop stays ASCENDING — bit for bit the sums of the [k][row] image.
// Staging: the operand whose k is contiguous in memory goes global float4 -> ds_write_b128; the other one (rows contiguous) is
// fetched as float4 along the rows (a wave = eight full 128-byte lines) and transposed on the way in, four scalar writes per float4
// (probe: 0.92 of the MFMA rate for the whole loop against 0.93 with both operands k-contiguous; 16 dword loads + b128 writes 0.84).  The next k-tile is fetched from global memory
// into registers while the current one is consumed; the block -> tile map keeps the tiles that share operand panels on one XCD.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GM 128
#define GN 128
#define GK 32
#define GLD 36                         // floats per LDS row: 16-byte aligned rows, b128 reads of 16 consecutive rows hit 64 distinct banks

// One TX x GK operand tile in registers: TX / 32 float4 per thread.
// CONTIG_K (element (x, k) at p[x * ld + k]): unit u = tid + 256 h -> row u / 8, k-group u % 8 (eight lanes = one 128-byte row).
// otherwise (element (x, k) at p[k * ld + x]): k = 8 * wave + lane % 8, rows 4 * (8 h + (lane / 8) % 8) .. + 3.
template <bool CONTIG_K, int TX, bool GUARD>
__device__ __forceinline__ void gemm2_tile_load(const float* __restrict__ p, int ld, int x0, int X, int k0, int kend, bool vec,
                                                float4 (&r)[TX / 32]) {
    if (CONTIG_K) {
#pragma unroll
        for (int h = 0; h < TX / 32; ++h) {
            const int u = threadIdx.x + h * 256;
            const int x = x0 + (u >> 3), k = k0 + (u & 7) * 4;
            if (!GUARD) r[h] = *(const float4*)(p + (size_t)x * ld + k);
            else r[h] = (x < X && k < kend) ? ld4g(p + (size_t)x * ld + k, kend - k, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        // rows contiguous: float4 along the rows; a wave = 8 k x 8 groups of 4 rows (eight full 128-byte lines per instruction)
        const int k = k0 + (threadIdx.x >> 6) * 8 + (threadIdx.x & 7);
#pragma unroll
        for (int h = 0; h < TX / 32; ++h) {
            const int x = x0 + 4 * (h * 8 + ((threadIdx.x >> 3) & 7));
            if (!GUARD) r[h] = *(const float4*)(p + (size_t)k * ld + x);
            else r[h] = (k < kend && x < X) ? ld4g(p + (size_t)k * ld + x, X - x, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
// column of k in a row of the LDS image: within every group of 8 k the four even ones, then the four odd ones — lanes 0-31 of an
// MFMA step take an even k and lanes 32-63 the odd one behind it, so the 16-byte read of a lane holds ITS k of four consecutive steps
// and the contraction runs over k in ascending order, two per instruction: the order (and the bits) of the [k][row] image of rounds
// 1-5 and of the small-tile kernel's sequential k loop — what an utterance's encoder output rounds to must not depend on the tile shape
// its batch size selects (batched == single searches, tests/test_decode_golden.py)
__device__ __forceinline__ int gemm2_col(int k) { return (k & ~7) + ((k & 1) << 2) + ((k & 7) >> 1); }

template <bool CONTIG_K, int TX>
__device__ __forceinline__ void gemm2_tile_store(float (*S)[GLD], const float4 (&r)[TX / 32]) {
    if (CONTIG_K) {
        // k = 4 (u % 8) .. + 3 -> columns gemm2_col(k): the even k of the float4 side by side, then the odd ones (two 8-byte writes)
#pragma unroll
        for (int h = 0; h < TX / 32; ++h) {
            const int u = threadIdx.x + h * 256;
            const int c0 = ((u & 7) >> 1) * 8 + (u & 1) * 2;
            *(float2*)&S[u >> 3][c0] = make_float2(r[h].x, r[h].z);
            *(float2*)&S[u >> 3][c0 + 4] = make_float2(r[h].y, r[h].w);
        }
    } else {
        // transposed on the way in: four scalar writes per float4; the lanes of a wave (8 row groups x 8 k) fall two to a bank
        const int k = gemm2_col((threadIdx.x >> 6) * 8 + (threadIdx.x & 7));
#pragma unroll
        for (int h = 0; h < TX / 32; ++h) {
            const int x = 4 * (h * 8 + ((threadIdx.x >> 3) & 7));
            S[x][k] = r[h].x; S[x + 1][k] = r[h].y; S[x + 2][k] = r[h].z; S[x + 3][k] = r[h].w;
        }
    }
}

// FAST (host-selected): operands 16-B aligned with ld % 4 == 0.  Blocks whose tile lies inside the matrix then take
// the unguarded loads for every complete k-tile; edge blocks and a ragged last k-tile of a chunk take the guarded ones.
// launch-linear block L of `total` -> position t in the tile sequence: block L runs on XCD L % 8; every XCD gets one contiguous
// run of the sequence, so tiles sharing an operand panel / a k-chunk meet in one L2
__device__ __forceinline__ int gemm_xcd_order(int L, int total) {
    const int xcd = L & 7, per = total >> 3, rem = total & 7;
    return xcd * per + min(xcd, rem) + (L >> 3);
}

// one TM x TN output tile (bx, by) of k-chunk / batch member bz.  TM = TN = 128: four waves of 64 x 64 (2 x 2 MFMA 32x32 blocks);
// TM = TN = 64: four waves of 32 x 32 (one block each) — for outputs of a few hundred 64-tiles that leave most of the chip idle as
// 128-tiles (the decoder's L*B- and T'*B-row products: 13 x 4 tiles of 128 against 25 x 8 of 64 over 512 resident work-groups)
template <int TM, int TN, bool TA, bool TB, bool FAST>
__device__ __forceinline__ void sgemm_tile(GemmArgs g, int bx, int by, int bz) {
    constexpr int MI = TM / 64, NI = TN / 64;                 // MFMA blocks per wave
    __shared__ __attribute__((aligned(16))) float As[2][TM][GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][TN][GLD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (g.batch > 1) {
        g.A += bz * g.sA; g.B += bz * g.sB; g.C += bz * g.sC;
        bz = 0;
    }
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const bool vecA = ((g.lda & 3) == 0) && ((((size_t)g.A) & 15) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((((size_t)g.B) & 15) == 0);
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[TM / 32], rb[TN / 32];
    // A: not transposed -> (m,k) at A[m*lda+k] (contiguous k); transposed -> A[k*lda+m] (contiguous m)
    const bool inside = FAST && m0 + TM <= g.M && n0 + TN <= g.N;
    if (inside && kbeg + GK <= kend) {
        gemm2_tile_load<!TA, TM, false>(g.A, g.lda, m0, g.M, kbeg, kend, true, ra);
        gemm2_tile_load<TB, TN, false>(g.B, g.ldb, n0, g.N, kbeg, kend, true, rb);
    } else {
        gemm2_tile_load<!TA, TM, true>(g.A, g.lda, m0, g.M, kbeg, kend, vecA, ra);
        gemm2_tile_load<TB, TN, true>(g.B, g.ldb, n0, g.N, kbeg, kend, vecB, rb);
    }
    gemm2_tile_store<!TA, TM>(As[0], ra);
    gemm2_tile_store<TB, TN>(Bs[0], rb);
    __syncthreads();
    int cur = 0;
    const int li = lane & 31, kh = (lane >> 5) * 4;           // (gemm2_col: even k of a group of 8 in columns 0-3, odd k in 4-7)
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        // the operands of the next four MFMA steps are fetched before the 4 * MI * NI MFMAs of the current four (the scheduler
        // otherwise emits read -> wait -> MFMAs and the LDS latency is paid per group)
        float4 pa[2][MI], pb[2][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) pa[0][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh];
#pragma unroll
        for (int j = 0; j < NI; ++j) pb[0][j] = *(const float4*)&Bs[cur][wn + 32 * j + li][kh];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q & 1, n = c ^ 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) pa[n][i] = *(const float4*)&As[cur][wm + 32 * i + li][kh + 8 * (q + 1)];
#pragma unroll
                for (int j = 0; j < NI; ++j) pb[n][j] = *(const float4*)&Bs[cur][wn + 32 * j + li][kh + 8 * (q + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const float x = e == 0 ? pa[c][i].x : e == 1 ? pa[c][i].y : e == 2 ? pa[c][i].z : pa[c][i].w;
                        const float y = e == 0 ? pb[c][j].x : e == 1 ? pb[c][j].y : e == 2 ? pb[c][j].z : pb[c][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // The next k-tile is fetched BEHIND the MFMAs of this one and written to the other LDS buffer as it arrives — no register
        // prefetch across the MFMAs.  Measured (tools/probes/gemm_bisect_probe.hip, 16 384 x 512 x 8 192, two work-groups per CU):
        // loads in front of the MFMAs 4.35 us per k-tile (121 TFLOP/s; the k-contiguous operand's loads — 8 rows x 128 bytes per
        // wave instruction — in front cost 5.0 on their own), all loads behind 3.97 (133 TFLOP/s): the latency then lies open in front of
        // the staging writes, where the co-resident work-group's MFMAs cover it; a wave that issues eight 1-KB loads and then MFMAs
        // does not get its MFMAs out.
        if (more) {
            if (inside && k0 + 2 * GK <= kend) {
                gemm2_tile_load<!TA, TM, false>(g.A, g.lda, m0, g.M, k0 + GK, kend, true, ra);
                gemm2_tile_load<TB, TN, false>(g.B, g.ldb, n0, g.N, k0 + GK, kend, true, rb);
            } else {
                gemm2_tile_load<!TA, TM, true>(g.A, g.lda, m0, g.M, k0 + GK, kend, vecA, ra);
                gemm2_tile_load<TB, TN, true>(g.B, g.ldb, n0, g.N, k0 + GK, kend, vecB, rb);
            }
            gemm2_tile_store<!TA, TM>(As[cur ^ 1], ra);
            gemm2_tile_store<TB, TN>(Bs[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    // C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn + j * 32 + (lane & 31);
                if (m < g.M && n < g.N) {
                    if (g.ksplit > 1) {
                        g.part[((size_t)bz * g.M + m) * g.N + n] = acc[i][j][r];
                    } else {
                        float v = g.alpha * acc[i][j][r];
                        if (g.beta != 0.f) v += g.beta * g.C[(size_t)m * g.ldc + n];
                        if (g.bias) v += g.bias[n];
                        g.C[(size_t)m * g.ldc + n] = v;
                    }
                }
            }
}
// ---- the [k][row] LDS image of rounds 1-5, kept for the 64 x 64 tiles --------------------------------------------------------------
// Operands staged k-major ([k][m], [k][n]: an MFMA operand read is a conflict-free row of 32 consecutive floats, one k per read), the next
// k-tile prefetched into registers across the MFMAs.  For a wave of ONE 32 x 32 block at four work-groups per CU this is the faster form
// (in-step products, rocprofv3 totals of 26 steps: transposed-A 77.3 ms against 86.0 with the [row][k] image, plain 81.7 against 87.6 —
// a row-contiguous operand needs no transposition on the way in, and the k-tile's 16 MFMAs are too few to amortise one); the 128 x 128
// tiles (64 MFMAs per wave and k-tile) run the [row][k] image above.  Both walk k in ascending order: the same bits.
#define GU (GK / 8)                    // float4 units per thread and operand tile
// one TX x GK (or GK x TX) operand tile (TX = 128 or 64): GK*TX/4 float4 units, TX/32 per thread.  CONTIG_K: element (x,k) at
// p[x*ld + k].
template <bool CONTIG_K, int TX>
__device__ __forceinline__ void gemm1_tile_load(const float* __restrict__ p, int ld, int x0, int X, int k0, int kend, bool vec,
                                               float4 (&r)[TX / 32]) {
#pragma unroll
    for (int h = 0; h < TX / 32; ++h) {
        const int u = threadIdx.x + h * 256;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CONTIG_K) {
            const int x = x0 + (u % TX), k = k0 + (u / TX) * 4;
            if (x < X && k < kend) v = ld4g(p + (size_t)x * ld + k, kend - k, vec);
        } else {
            const int k = k0 + u / (TX / 4), x = x0 + (u % (TX / 4)) * 4;
            if (k < kend && x < X) v = ld4g(p + (size_t)k * ld + x, X - x, vec);
        }
        r[h] = v;
    }
}
// the same tile when it is known to be complete, in bounds and 16-B aligned: straight-line dwordx4 loads.  (The guarded
// form compiles to exec-masked branches whose results are merged right behind them, i.e. the wave waits for its
// "prefetch" before it starts the MFMAs of the current tile — measured as exactly half the MFMA rate.)
template <bool CONTIG_K, int TX>
__device__ __forceinline__ void gemm1_tile_load_fast(const float* __restrict__ p, int ld, int x0, int k0, float4 (&r)[TX / 32]) {
#pragma unroll
    for (int h = 0; h < TX / 32; ++h) {
        const int u = threadIdx.x + h * 256;
        if (CONTIG_K) r[h] = *(const float4*)(p + (size_t)(x0 + (u % TX)) * ld + k0 + (u / TX) * 4);
        else r[h] = *(const float4*)(p + (size_t)(k0 + u / (TX / 4)) * ld + x0 + (u % (TX / 4)) * 4);
    }
}
template <bool CONTIG_K, int TX>
__device__ __forceinline__ void gemm1_tile_store(float (*S)[TX + 4], const float4 (&r)[TX / 32]) {
#pragma unroll
    for (int h = 0; h < TX / 32; ++h) {
        const int u = threadIdx.x + h * 256;
        if (CONTIG_K) {
            const int x = u % TX, k = (u / TX) * 4;
            S[k + 0][x] = r[h].x; S[k + 1][x] = r[h].y; S[k + 2][x] = r[h].z; S[k + 3][x] = r[h].w;
        } else {
            const int k = u / (TX / 4), x = (u % (TX / 4)) * 4;
            *(float4*)&S[k][x] = r[h];
        }
    }
}

template <int TM, int TN, bool TA, bool TB, bool FAST>
__device__ __forceinline__ void sgemm_tile_kmajor(GemmArgs g, int bx, int by, int bz) {
    constexpr int MI = TM / 64, NI = TN / 64;                 // MFMA blocks per wave
    __shared__ __attribute__((aligned(16))) float As[2][GK][TM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK][TN + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (g.batch > 1) {
        g.A += bz * g.sA; g.B += bz * g.sB; g.C += bz * g.sC;
        bz = 0;
    }
    const int m0 = by * TM, n0 = bx * TN;
    const int kbeg = bz * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int wm = (wave >> 1) * (TM / 2), wn = (wave & 1) * (TN / 2);
    const bool vecA = ((g.lda & 3) == 0) && ((((size_t)g.A) & 15) == 0);
    const bool vecB = ((g.ldb & 3) == 0) && ((((size_t)g.B) & 15) == 0);
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[TM / 32], rb[TN / 32];
    // A: not transposed -> (m,k) at A[m*lda+k] (contiguous k); transposed -> A[k*lda+m] (contiguous m)
    const bool inside = FAST && m0 + TM <= g.M && n0 + TN <= g.N;
    if (inside && kbeg + GK <= kend) {
        gemm1_tile_load_fast<!TA, TM>(g.A, g.lda, m0, kbeg, ra);
        gemm1_tile_load_fast<TB, TN>(g.B, g.ldb, n0, kbeg, rb);
    } else {
        gemm1_tile_load<!TA, TM>(g.A, g.lda, m0, g.M, kbeg, kend, vecA, ra);
        gemm1_tile_load<TB, TN>(g.B, g.ldb, n0, g.N, kbeg, kend, vecB, rb);
    }
    gemm1_tile_store<!TA, TM>(As[0], ra);
    gemm1_tile_store<TB, TN>(Bs[0], rb);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        const bool more = k0 + GK < kend;
        if (more) {
            if (inside && k0 + 2 * GK <= kend) {
                gemm1_tile_load_fast<!TA, TM>(g.A, g.lda, m0, k0 + GK, ra);
                gemm1_tile_load_fast<TB, TN>(g.B, g.ldb, n0, k0 + GK, rb);
            } else {
                gemm1_tile_load<!TA, TM>(g.A, g.lda, m0, g.M, k0 + GK, kend, vecA, ra);
                gemm1_tile_load<TB, TN>(g.B, g.ldb, n0, g.N, k0 + GK, kend, vecB, rb);
            }
        }
        // operand fetch of MFMA step s+1 is issued before the MFMAs of step s (the scheduler otherwise emits
        // read -> wait -> MFMAs per step and the LDS latency is paid 16 times per k-tile)
        const int kr0 = lane >> 5, li = lane & 31;
        float pa[2][MI], pb[2][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) pa[0][i] = As[cur][kr0][wm + 32 * i + li];
#pragma unroll
        for (int j = 0; j < NI; ++j) pb[0][j] = Bs[cur][kr0][wn + 32 * j + li];
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < GK / 2) {
                const int kr = 2 * (s + 1) + kr0;
#pragma unroll
                for (int i = 0; i < MI; ++i) pa[n][i] = As[cur][kr][wm + 32 * i + li];
#pragma unroll
                for (int j = 0; j < NI; ++j) pb[n][j] = Bs[cur][kr][wn + 32 * j + li];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c][i], pb[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            gemm1_tile_store<!TA, TM>(As[cur ^ 1], ra);
            gemm1_tile_store<TB, TN>(Bs[cur ^ 1], rb);
        }
        __syncthreads();
        cur ^= 1;
    }
    // C/D layout of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn + j * 32 + (lane & 31);
                if (m < g.M && n < g.N) {
                    if (g.ksplit > 1) {
                        g.part[((size_t)bz * g.M + m) * g.N + n] = acc[i][j][r];
                    } else {
                        float v = g.alpha * acc[i][j][r];
                        if (g.beta != 0.f) v += g.beta * g.C[(size_t)m * g.ldc + n];
                        if (g.bias) v += g.bias[n];
                        g.C[(size_t)m * g.ldc + n] = v;
                    }
                }
            }
}
template <bool TA, bool TB, bool FAST>
__device__ __forceinline__ void sgemm128_tile(GemmArgs g, int bx, int by, int bz) { sgemm_tile<128, 128, TA, TB, FAST>(g, bx, by, bz); }

template <bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(256) void lvsr_sgemm64_kernel(GemmArgs g) {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    const int t = gemm_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total);
    sgemm_tile_kmajor<64, 64, TA, TB, FAST>(g, t % gridDim.x, (t / gridDim.x) % gridDim.y, t / (gridDim.x * gridDim.y));
}

template <bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(256) void lvsr_sgemm128_kernel(GemmArgs g) {
    // (k-split, m-tile, n-tile) sequence, n fastest
    const int total = gridDim.x * gridDim.y * gridDim.z;
    const int t = gemm_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), total);
    sgemm128_tile<TA, TB, FAST>(g, t % gridDim.x, (t / gridDim.x) % gridDim.y, t / (gridDim.x * gridDim.y));
}

// ---- grouped launch of transposed-A products (weight gradients): C_p = A_p^T B_p (+ beta_p C_p) for up to GROUP_MAX problems.
// The recurrent weight gradients of the encoder (H x H / H x 2H outputs, K = T*B) and the decoder's (K = L*B) are each too small
// to fill the chip: alone they were split 50-fold over K (256-deep chunks: prologue-bound, 13 MB of partials per product) or ran
// on a handful of work-groups at 26 us per launch.  Together their (problem, tile, k-chunk) units fill the chip with ~1 000-deep
// chunks in ONE launch, and one more launch folds all partials in a fixed order (deterministic).
#define GROUP_MAX 40
struct GroupDesc {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    float beta;
    int ksplit, kchunk, gx, gy, unit0;      // tiles gx x gy, ksplit chunks; unit0 = first unit of this problem in the launch
    long long part_off;                     // offset (floats) of this problem's partials in the workspace
};
struct GroupPack { GroupDesc d[GROUP_MAX]; int n, total; float* part; };

template <bool FAST>
__global__ __launch_bounds__(256) void lvsr_sgemm128_grouped_tn_kernel(GroupPack pk) {
    const int t = gemm_xcd_order(blockIdx.x, pk.total);
    int p = 0;
    for (int x = 1; x < pk.n; ++x)
        if (t >= pk.d[x].unit0) p = x;
    const GroupDesc& d = pk.d[p];
    const int local = t - d.unit0;
    GemmArgs g;
    g.A = d.A; g.B = d.B; g.C = d.C; g.bias = nullptr;
    g.M = d.M; g.N = d.N; g.K = d.K; g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc; g.transA = 1; g.transB = 0;
    g.alpha = 1.f; g.beta = d.beta; g.ksplit = d.ksplit; g.kchunk = d.kchunk; g.part = pk.part + d.part_off;
    g.batch = 1; g.sA = g.sB = g.sC = 0;
    sgemm128_tile<true, false, FAST>(g, local % d.gx, (local / d.gx) % d.gy, local / (d.gx * d.gy));
}

// blockIdx.y = problem; its blocks stride over the output elements and fold the k-chunks in order
__global__ __launch_bounds__(256) void lvsr_sgemm_grouped_reduce(GroupPack pk) {
    const GroupDesc& d = pk.d[blockIdx.y];
    if (d.ksplit <= 1) return;
    const size_t total = (size_t)d.M * d.N;
    const float* part = pk.part + d.part_off;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / d.N), n = (int)(idx % d.N);
        float s = 0.f;
        for (int z = 0; z < d.ksplit; ++z) s += part[(size_t)z * total + idx];
        if (d.beta != 0.f) s += d.beta * d.C[(size_t)m * d.ldc + n];
        d.C[(size_t)m * d.ldc + n] = s;
    }
}

__global__ __launch_bounds__(256) void lvsr_sgemm_splitk_reduce(GemmArgs g) {
    const size_t total = (size_t)g.M * g.N;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / g.N), n = (int)(idx % g.N);
        float s = 0.f;
        for (int z = 0; z < g.ksplit; ++z) s += g.part[(size_t)z * total + idx];
        float v = g.alpha * s;
        if (g.beta != 0.f) v += g.beta * g.C[(size_t)m * g.ldc + n];
        if (g.bias) v += g.bias[n];
        g.C[(size_t)m * g.ldc + n] = v;
    }
}

// column sums: out[n] = beta*out[n] + sum_m X[m*ldx + n]   (bias gradients, partial-sum folding)
// grid (ceil(N/64), S): block (x, s) sums rows [s*rows_per, (s+1)*rows_per) of 64 columns; S > 1 writes
// partials to the workspace and lvsr_colsum_finish folds them in a fixed order (deterministic).
__global__ __launch_bounds__(256) void lvsr_colsum_kernel(const float* X, int M, int N, int ldx, float* out, float beta,
                                                          int rows_per, float* part) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int m = m0 + g;
        for (; m + 12 < m1; m += 16) {          // 4 independent loads in flight per thread
            s0 += X[(size_t)m * ldx + n];
            s1 += X[(size_t)(m + 4) * ldx + n];
            s2 += X[(size_t)(m + 8) * ldx + n];
            s3 += X[(size_t)(m + 12) * ldx + n];
        }
        for (; m < m1; m += 4) s0 += X[(size_t)m * ldx + n];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && n < N) {
        const int c = threadIdx.x & 63;
        const float v = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        if (part) part[(size_t)blockIdx.y * N + n] = v;
        else out[n] = (beta != 0.f ? beta * out[n] : 0.f) + v;
    }
}

// ---- many column sums in one launch: block -> (member, column block x, row split s); the member's arithmetic is lvsr_colsum_kernel's
#define COLSUM_MANY_MAX 32
struct ColsumMember { const float* X; float* out; float* part; int M, N, ldx, S, rows_per, nx, block0, fin0; float beta; };
struct ColsumPack { ColsumMember m[COLSUM_MANY_MAX]; int n; };
__global__ __launch_bounds__(256) void lvsr_colsum_many_kernel(ColsumPack pk) {
    __shared__ float red[4][64];
    int p = 0;
    for (int i = 1; i < pk.n; ++i)
        if ((int)blockIdx.x >= pk.m[i].block0) p = i;
    const ColsumMember& c = pk.m[p];
    const int local = blockIdx.x - c.block0, bx = local % c.nx, by = local / c.nx;
    const int n = bx * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int m0 = by * c.rows_per, m1 = min(c.M, m0 + c.rows_per);
    const float* X = c.X;
    const int ldx = c.ldx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < c.N) {
        int m = m0 + g;
        for (; m + 12 < m1; m += 16) {          // 4 independent loads in flight per thread
            s0 += X[(size_t)m * ldx + n];
            s1 += X[(size_t)(m + 4) * ldx + n];
            s2 += X[(size_t)(m + 8) * ldx + n];
            s3 += X[(size_t)(m + 12) * ldx + n];
        }
        for (; m < m1; m += 4) s0 += X[(size_t)m * ldx + n];
    }
    red[g][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && n < c.N) {
        const int col = threadIdx.x & 63;
        const float v = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
        if (c.S > 1) c.part[(size_t)by * c.N + n] = v;
        else c.out[n] = (c.beta != 0.f ? c.beta * c.out[n] : 0.f) + v;
    }
}
__global__ __launch_bounds__(256) void lvsr_colsum_many_finish(ColsumPack pk) {
    int p = 0;
    for (int i = 1; i < pk.n; ++i)
        if ((int)blockIdx.x >= pk.m[i].fin0) p = i;
    const ColsumMember& c = pk.m[p];
    if (c.S <= 1) return;
    const int n = ((int)blockIdx.x - c.fin0) * 256 + threadIdx.x;
    if (n >= c.N) return;
    float v = 0.f;
    for (int s = 0; s < c.S; ++s) v += c.part[(size_t)s * c.N + n];
    c.out[n] = (c.beta != 0.f ? c.beta * c.out[n] : 0.f) + v;
}

__global__ __launch_bounds__(256) void lvsr_colsum_finish(const float* part, int S, int N, float* out, float beta) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += part[(size_t)s * N + n];
    out[n] = (beta != 0.f ? beta * out[n] : 0.f) + v;
}

// out[c*rows + r] = in[r*cols + c]
#define COPY_MANY_MAX 32
struct CopyPack { lvsr_copy_desc d[COPY_MANY_MAX]; int n; };
// blockIdx.y = descriptor; the blocks of a descriptor stride over its rows, the threads over 16-byte pieces of a row
__global__ __launch_bounds__(256) void lvsr_copy2d_many_kernel(CopyPack pk) {
    const lvsr_copy_desc& d = pk.d[blockIdx.y];
    const bool vec = ((d.cols | d.lds | d.ldd) & 3) == 0 && ((((size_t)d.src) | ((size_t)d.dst)) & 15) == 0;
    if (d.rows == 1) {                                       // vectors: all blocks share the one row
        for (int c = blockIdx.x * 256 + threadIdx.x; c < d.cols; c += gridDim.x * 256)
            d.dst[c] = d.beta != 0.f ? d.src[c] + d.beta * d.dst[c] : d.src[c];
        return;
    }
    for (int r = blockIdx.x; r < d.rows; r += gridDim.x) {
        const float* s = d.src + (size_t)r * d.lds;
        float* o = d.dst + (size_t)r * d.ldd;
        if (d.beta != 0.f) {
            for (int c = threadIdx.x; c < d.cols; c += 256) o[c] = s[c] + d.beta * o[c];
        } else if (vec) {
            for (int c = threadIdx.x * 4; c < d.cols; c += 1024) *(float4*)(o + c) = *(const float4*)(s + c);
        } else {
            for (int c = threadIdx.x; c < d.cols; c += 256) o[c] = s[c];
        }
    }
}

__global__ __launch_bounds__(256) void lvsr_transpose_kernel(const float* in, int rows, int cols, float* out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][j];
    }
}

// Pack a weight into the operand order of rb_mm (common.h): P[tile][wave][q][lane][r] with
// k = wave*Kw + (lane>>4)*(Kw/4) + 4q + r, col = tile*16 + (lane&15); zero padded.
__global__ __launch_bounds__(256) void lvsr_pack_b_kernel(const float* W, int ldw, int K, int N, int trans, float* P,
                                                          int Kw, int NQ, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const long long g = idx >> 8;
        const int q = (int)(g % NQ), wave = (int)((g / NQ) & 3), tile = (int)(g / NQ / 4);
        const int k = wave * Kw + (lane >> 4) * (Kw >> 2) + 4 * q + r, col = tile * 16 + (lane & 15);
        float v = 0.f;
        if (k < K && col < N) v = trans ? W[(size_t)col * ldw + k] : W[(size_t)k * ldw + col];
        P[idx] = v;
    }
}

extern "C" {

static int sgemm_launch(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                        const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
                        long long ws_bytes, int batch, long long sA, long long sB, long long sC) {
    LVSR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 1, "lvsr_sgemm: negative dimension");
    if (M == 0 || N == 0) return LVSR_OK;
    GemmArgs g;
    g.batch = batch; g.sA = sA; g.sB = sB; g.sC = sC;
    if (batch > 1) ws = nullptr;
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.transA = transA; g.transB = transB;
    g.alpha = alpha; g.beta = beta; g.ksplit = 1; g.kchunk = 0; g.part = nullptr;
    // at least ~one 128x128 tile of real work — or a long contraction onto a skinny output (the first layer's 40-row weight
    // gradient: K = T*B), which the 64-tile kernel with split-K serves better than the small-tile one (107 -> see profiles)
    const bool big = (M >= 96 && N >= 96) || (min(M, N) >= 32 && max(M, N) >= 256 && K >= 2048);
    int tm = big ? GM : BM, tn = big ? GN : BN;
    const int tk = big ? GK : BK;
    int tiles = ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
    // fewer 128-tiles than CUs: the same kernel on 64 x 64 tiles (four times the work-groups, four of them resident per CU)
    // 64 x 64 tiles: always when the 128-tiles would not fill the chip once; up to a few rounds of them when K is
    // short, where the last, partly filled round costs more than the smaller tile's lower arithmetic intensity
    // (profiles/r03_gemm_sweep.txt: 6400 x 1536 x 512 122 -> 98 us, 12800 x 512 x 1536 NT 229 -> 200 us; 4096^3 loses 3%).
    const int mid_max = lvsr_knob(LVSR_KNOB_GEMM_MID_TILES) > 0 ? lvsr_knob(LVSR_KNOB_GEMM_MID_TILES) : 2048;
    const bool mid = big && batch == 1 && (tiles < 256 || (tiles < mid_max && K <= 2048));
    if (mid) { tm = 64; tn = 64; tiles = ((M + 63) / 64) * ((N + 63) / 64); }
    g.kchunk = ((K + tk - 1) / tk) * tk;
    if (g.kchunk == 0) g.kchunk = tk;
    // deterministic split-K when the output is too small to fill the resident work-group slots (two 128-tiles or four 64-tiles
    // per CU) and K is long
    const int slots = mid ? 1024 : 512;
    if (ws && 2 * tiles <= slots && K >= 1024) {
        // (rounded UP: 192 tiles x 6 chunks = 1 152 work-groups on 1 024 slots measured 199 us for the layer weight gradient
        // (512, 1536, 12 800), 192 x 5 = 960 in a single round 210 us — the longer chunks cost more than the partial second round)
        int want = (slots + tiles - 1) / tiles;
        int maxk = K / 256;
        if (want > maxk) want = maxk;
        long long need = (long long)want * M * N * 4;
        while (want > 1 && need > ws_bytes) { --want; need = (long long)want * M * N * 4; }
        if (want > 1) {
            int chunk = (K + want - 1) / want;
            chunk = ((chunk + tk - 1) / tk) * tk;
            g.ksplit = (K + chunk - 1) / chunk;
            g.kchunk = chunk;
            g.part = ws;
        }
    }
    dim3 grid((N + tn - 1) / tn, (M + tm - 1) / tm, batch > 1 ? batch : g.ksplit);
    hipStream_t st = (hipStream_t)stream;
    if (!big) {
        hipLaunchKernelGGL(lvsr_sgemm_kernel, grid, dim3(256), 0, st, g);
    } else {
        const bool fast = (lda & 3) == 0 && (ldb & 3) == 0 && (((size_t)A) & 15) == 0 && (((size_t)B) & 15) == 0 &&
                          (batch == 1 || ((sA & 3) == 0 && (sB & 3) == 0));
#define LVSR_GEMM128(TA_, TB_)                                                                              \
    do {                                                                                                    \
        if (fast) hipLaunchKernelGGL((lvsr_sgemm128_kernel<TA_, TB_, true>), grid, dim3(256), 0, st, g);    \
        else hipLaunchKernelGGL((lvsr_sgemm128_kernel<TA_, TB_, false>), grid, dim3(256), 0, st, g);        \
    } while (0)
#define LVSR_GEMM64(TA_, TB_)                                                                               \
    do {                                                                                                    \
        if (fast) hipLaunchKernelGGL((lvsr_sgemm64_kernel<TA_, TB_, true>), grid, dim3(256), 0, st, g);     \
        else hipLaunchKernelGGL((lvsr_sgemm64_kernel<TA_, TB_, false>), grid, dim3(256), 0, st, g);         \
    } while (0)
        if (mid) {
            if (!transA && !transB) LVSR_GEMM64(false, false);
            else if (transA && !transB) LVSR_GEMM64(true, false);
            else if (!transA && transB) LVSR_GEMM64(false, true);
            else LVSR_GEMM64(true, true);
        } else if (!transA && !transB) LVSR_GEMM128(false, false);
        else if (transA && !transB) LVSR_GEMM128(true, false);
        else if (!transA && transB) LVSR_GEMM128(false, true);
        else LVSR_GEMM128(true, true);
#undef LVSR_GEMM128
#undef LVSR_GEMM64
    }
    if (g.ksplit > 1) {
        int nb = (int)(((size_t)M * N + 255) / 256);
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(lvsr_sgemm_splitk_reduce, dim3(nb), dim3(256), 0, st, g);
    }
    return lvsr_check_launch("lvsr_sgemm");
}

int lvsr_sgemm(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
               const float* B, int ldb, float beta, float* C, int ldc, const float* bias, float* ws,
               long long ws_bytes) {
    return sgemm_launch(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ws, ws_bytes, 1, 0, 0, 0);
}

int lvsr_sgemm_batched(void* stream, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                       long long strideA, const float* B, int ldb, long long strideB, float beta, float* C, int ldc,
                       long long strideC, int batch) {
    LVSR_REQUIRE(batch >= 0, "lvsr_sgemm_batched: negative batch");
    if (batch == 0) return LVSR_OK;
    return sgemm_launch(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, nullptr, 0, batch,
                        strideA, strideB, strideC);
}

int lvsr_sgemm_tn_grouped(void* stream, const lvsr_gemm_desc* descs, int n, float* ws, long long ws_bytes) {
    LVSR_REQUIRE(n >= 0 && (n == 0 || descs), "lvsr_sgemm_tn_grouped: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i)
        LVSR_REQUIRE(descs[i].A && descs[i].B && descs[i].C && descs[i].M > 0 && descs[i].N > 0 && descs[i].K > 0,
                     "lvsr_sgemm_tn_grouped: bad descriptor %d", i);
    auto aligned = [](const lvsr_gemm_desc& s) {
        return (s.lda & 3) == 0 && (s.ldb & 3) == 0 && (((size_t)s.A) & 15) == 0 && (((size_t)s.B) & 15) == 0;
    };
    // k-chunk depth: ~1024 where the partials of ALL members fit the workspace, else the smallest common depth that fits (starving
    // the members that come last of their splits instead left them as long serial tails: WSJ-deep 108 -> 113 ms)
    int target = 1024;
    if (ws) {
        for (;; target += 512) {
            long long need = 0;
            bool splits = false;
            for (int i = 0; i < n; ++i) {
                const int want = (descs[i].K + target - 1) / target;
                if (want > 1) { need += (long long)want * descs[i].M * descs[i].N; splits = true; }
            }
            if (!splits || need * 4 <= ws_bytes) break;
        }
    }
    // two passes: the problems whose operands allow unguarded 16-byte loads share the fast kernel, the others the guarded one
    // (one misaligned member — the V-wide output-layer gradient — would otherwise put the whole group on the guarded loads)
    long long off = 0;
    for (int pass = 0; pass < 2; ++pass) {
        GroupPack pk;
        pk.n = 0; pk.part = ws;
        int units = 0;
        bool any_split = false;
        auto launch = [&]() {
            if (pk.n == 0) return;
            pk.total = units;
            if (pass == 0) hipLaunchKernelGGL(lvsr_sgemm128_grouped_tn_kernel<true>, dim3(units), dim3(256), 0, st, pk);
            else hipLaunchKernelGGL(lvsr_sgemm128_grouped_tn_kernel<false>, dim3(units), dim3(256), 0, st, pk);
            if (any_split) hipLaunchKernelGGL(lvsr_sgemm_grouped_reduce, dim3(32, pk.n), dim3(256), 0, st, pk);
            pk.n = 0; units = 0; any_split = false;
        };
        for (int i = 0; i < n; ++i) {
            const lvsr_gemm_desc& s = descs[i];
            if (aligned(s) != (pass == 0)) continue;
            if (pk.n == GROUP_MAX) launch();
            GroupDesc& d = pk.d[pk.n++];
            d.A = s.A; d.B = s.B; d.C = s.C; d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.beta = s.beta;
            d.gx = (s.N + GN - 1) / GN; d.gy = (s.M + GM - 1) / GM;
            int want = ws ? (s.K + target - 1) / target : 1;
            int chunk = (s.K + want - 1) / want;
            chunk = ((chunk + GK - 1) / GK) * GK;
            d.kchunk = chunk;
            d.ksplit = (s.K + chunk - 1) / chunk;
            d.part_off = off;
            if (d.ksplit > 1) { off += (long long)d.ksplit * s.M * s.N; any_split = true; }
            d.unit0 = units;
            units += d.gx * d.gy * d.ksplit;
        }
        launch();
    }
    return lvsr_check_launch("lvsr_sgemm_tn_grouped");
}

int lvsr_colsum(void* stream, const float* X, int M, int N, int ldx, float* out, float beta, float* ws,
                long long ws_bytes) {
    if (N <= 0) return LVSR_OK;
    const int nx = (N + 63) / 64;
    int S = 1;
    if (ws && M >= 512) {
        S = (M + 255) / 256;
        const int want = (1024 + nx - 1) / nx;          // aim at ~1024 work-groups
        if (S > want) S = want;
        while (S > 1 && (long long)S * N * 4 > ws_bytes) --S;
    }
    const int rows_per = (M + S - 1) / S;
    hipLaunchKernelGGL(lvsr_colsum_kernel, dim3(nx, S), dim3(256), 0, (hipStream_t)stream, X, M, N, ldx, out, beta, rows_per,
                       S > 1 ? ws : nullptr);
    if (S > 1)
        hipLaunchKernelGGL(lvsr_colsum_finish, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, S, N, out, beta);
    return lvsr_check_launch("lvsr_colsum");
}

int lvsr_colsum_many(void* stream, const lvsr_colsum_desc* descs, int n, float* ws, long long ws_bytes, long long split_ws_bytes) {
    LVSR_REQUIRE(n >= 0 && (n == 0 || descs), "lvsr_colsum_many: bad arguments");
    for (int i = 0; i < n; ++i)
        LVSR_REQUIRE(descs[i].X && descs[i].out && descs[i].M >= 0 && descs[i].N > 0, "lvsr_colsum_many: bad descriptor %d", i);
    for (int i0 = 0; i0 < n; i0 += COLSUM_MANY_MAX) {
        ColsumPack pk;
        pk.n = n - i0 < COLSUM_MANY_MAX ? n - i0 : COLSUM_MANY_MAX;
        int blocks = 0, fin = 0;
        long long off = 0;
        bool any_split = false;
        for (int i = 0; i < pk.n; ++i) {
            const lvsr_colsum_desc& d = descs[i0 + i];
            ColsumMember& c = pk.m[i];
            c.X = d.X; c.out = d.out; c.M = d.M; c.N = d.N; c.ldx = d.ldx; c.beta = d.beta;
            c.nx = (d.N + 63) / 64;
            int S = 1;                                      // the split lvsr_colsum makes, from the workspace IT would have been given
            if (ws && d.M >= 512) {
                S = (d.M + 255) / 256;
                const int want = (1024 + c.nx - 1) / c.nx;
                if (S > want) S = want;
                while (S > 1 && (long long)S * d.N * 4 > split_ws_bytes) --S;
            }
            c.S = S;
            c.rows_per = (d.M + S - 1) / S;
            c.part = ws ? ws + off : nullptr;
            if (S > 1) { off += (long long)S * d.N; any_split = true; }
            c.block0 = blocks; blocks += c.nx * S;
            c.fin0 = fin; fin += (d.N + 255) / 256;
        }
        LVSR_REQUIRE(off * 4 <= ws_bytes, "lvsr_colsum_many: workspace too small for the partial sums (%lld bytes needed)", off * 4);
        if (blocks > 0) hipLaunchKernelGGL(lvsr_colsum_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pk);
        if (any_split) hipLaunchKernelGGL(lvsr_colsum_many_finish, dim3(fin), dim3(256), 0, (hipStream_t)stream, pk);
        // (a second chunk of members would reuse the partials' region: one chunk per call is what the step needs; guard it)
        LVSR_REQUIRE(i0 + COLSUM_MANY_MAX >= n || !any_split, "lvsr_colsum_many: more than %d members with row splits", COLSUM_MANY_MAX);
    }
    return lvsr_check_launch("lvsr_colsum_many");
}

int lvsr_copy2d_many(void* stream, const lvsr_copy_desc* descs, int n) {
    LVSR_REQUIRE(n >= 0 && (n == 0 || descs), "lvsr_copy2d_many: bad arguments");
    for (int i = 0; i < n; ++i)
        LVSR_REQUIRE(descs[i].rows >= 0 && descs[i].cols >= 0 && descs[i].src && descs[i].dst, "lvsr_copy2d_many: bad descriptor %d", i);
    for (int i0 = 0; i0 < n; i0 += COPY_MANY_MAX) {
        CopyPack pk;
        pk.n = n - i0 < COPY_MANY_MAX ? n - i0 : COPY_MANY_MAX;
        for (int i = 0; i < pk.n; ++i) pk.d[i] = descs[i0 + i];
        hipLaunchKernelGGL(lvsr_copy2d_many_kernel, dim3(128, pk.n), dim3(256), 0, (hipStream_t)stream, pk);
    }
    return lvsr_check_launch("lvsr_copy2d_many");
}

int lvsr_transpose(void* stream, const float* in, int rows, int cols, float* out) {
    if (rows <= 0 || cols <= 0) return LVSR_OK;
    hipLaunchKernelGGL(lvsr_transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                       in, rows, cols, out);
    return lvsr_check_launch("lvsr_transpose");
}

long long lvsr_pack_size(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (long long)((N + 15) / 16) * 16 * 4 * lvsr_pack_kw(K);
}

int lvsr_pack_b(void* stream, const float* W, int ldw, int K, int N, int trans, float* packed) {
    LVSR_REQUIRE(K > 0 && N > 0 && W && packed, "lvsr_pack_b: bad arguments K=%d N=%d", K, N);
    const int Kw = lvsr_pack_kw(K), NQ = Kw / 16;
    const long long total = lvsr_pack_size(K, N);
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(lvsr_pack_b_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, W, ldw, K, N, trans, packed, Kw,
                       NQ, total);
    return lvsr_check_launch("lvsr_pack_b");
}

int lvsr_pack_b_many(void* stream, const lvsr_pack_desc* descs, int n, int use_graph) {
    LVSR_REQUIRE(n >= 0 && (n == 0 || descs), "lvsr_pack_b_many: bad arguments");
    if (n == 0) return LVSR_OK;
    for (int i = 0; i < n; ++i)
        LVSR_REQUIRE(descs[i].K > 0 && descs[i].N > 0 && descs[i].W && descs[i].packed, "lvsr_pack_b_many: bad descriptor %d", i);
    hipStream_t s = (hipStream_t)stream;
    auto enqueue = [&]() {
        for (int i = 0; i < n; ++i) {
            const lvsr_pack_desc& d = descs[i];
            const int Kw = lvsr_pack_kw(d.K), NQ = Kw / 16;
            const long long total = lvsr_pack_size(d.K, d.N);
            int nb = (int)((total + 255) / 256);
            if (nb > 4096) nb = 4096;
            hipLaunchKernelGGL(lvsr_pack_b_kernel, dim3(nb), dim3(256), 0, s, d.W, d.ldw, d.K, d.N, d.trans, d.packed, Kw, NQ, total);
        }
    };
    GraphKey key("pack_many");
    key.add(descs, sizeof(lvsr_pack_desc) * (size_t)n);
    return lvsr_run_graph(s, use_graph, key, enqueue, "lvsr_pack_b_many");
}

}  // extern "C"
