// Pieces shared by the persistent attention-decoder kernels (decoder_persist.hip: forward, decoder_persist_bwd.hip: backward).
#pragma once
#include "decoder.h"
#include "persist.h"

#define PD_THREADS 512       // two waves per SIMD: with one, every LDS / transcendental latency of the energy phase is exposed
#define PD_CH 16             // attended positions per energy round (8 per half of the work-group)
#define PD_EG 2              // positions whose chains are written interleaved in the energy phase (registers: 12 per position)
#define PD_MP 256            // match-column pairs (m, m + 256): one per thread of a half
#define PD_MAXV 512          // longest exchanged vector
#define PD_NV (PD_MAXV / PD_THREADS)     // granules per thread and sweep
#define PD_NW (PD_THREADS / 64)
#define PD_LDS_FLOATS (40 * 1024 - 64)      // of the CU's 160 KB
#define PD_NPLANE 4          // SW | EN | RS | S
#define PD_NPLANE_STACK 7    // two-layer launch: + SW1 | RS1 | S1 of the layer-1 cluster
#define PD_NPROF 16
#define PD_MAXP 32           // work-groups per cluster at most (XCC_ID granules per utterance)

// Cluster shapes.  A cluster of P = ceil(D / UNITS) work-groups serves one utterance; thread (unit jl = tid / KSPLIT, slice q =
// tid % KSPLIT) keeps KD rows of its unit's state columns and of MC transform_states columns in registers (UNITS * KSPLIT = 512
// threads, KSPLIT * KD >= D, P * MC * UNITS >= M).
//   PdShape<32, 16, 2>  clusters of  8 at D = 256 (round 3): 128 of the 256 CUs at B = 16, T' <= ~216 / 224 (LDS budget)
//   PdShape<16,  8, 2>  clusters of 16 at D = 256 (round 4): all 256 CUs at B = 16; half the positions, gate columns, match
//                       columns and state rows per work-group: one energy round of 16 positions instead of two, half the
//                       alignment correlation, T' <= ~440
//   PdShape<16, 16, 1>  clusters of up to 32 for 256 < D <= 512 (round 4: WSJ-deep's decoder, B = 8 on all 256 CUs)
template <int UNITS_, int KD_, int MC_>
struct PdShape {
    static constexpr int UNITS = UNITS_, KSPLIT = PD_THREADS / UNITS_, KD = KD_, MC = MC_;
    static constexpr int DMAX = KSPLIT * KD_;                 // widest decoder
    // LDS row stride of the AW slice [x | u | r][UNITS]: lanes q = 0..KSPLIT-1 of a unit read consecutive rows — an odd stride
    // (KSPLIT = 32) spreads them over all banks; +4 words for KSPLIT = 16 (the round-3 layout)
    static constexpr int AWS = UNITS_ == 32 ? 3 * UNITS_ + 4 : 3 * UNITS_ + 1;
};
typedef PdShape<32, 16, 2> PdShape8;
typedef PdShape<16, 8, 2> PdShape16;
typedef PdShape<16, 16, 1> PdShape32;

// Which shape serves (B, D, M): LVSR_KNOB_DEC_CLUSTER 0 = clusters of 16 at D <= 256 when B * ceil(D/16) work-groups fit the chip
// (one per CU), else clusters of 8; 8 / 16 force one of the two (the kernels refuse what does not fit).  D > 256: PdShape32.
// -> 0 / 1 / 2 (PdShape8 / 16 / 32) or -1; fills units, ksplit, kd, mc, P.
struct PdPick { int shape, UNITS, KSPLIT, KD, MC, AWS, P; };
static inline bool pd_pick(int B, int D, int M, PdPick& k, bool allow16 = true) {
    auto set = [&](int shape, int U, int KD, int MC, int AWS) {
        k.shape = shape; k.UNITS = U; k.KSPLIT = PD_THREADS / U; k.KD = KD; k.MC = MC; k.AWS = AWS; k.P = (D + U - 1) / U;
        return M <= k.P * MC * U && k.P <= PD_MAXP && B * k.P <= lvsr_max_cluster_wgs();
    };
    if (D > PdShape32::DMAX) return false;
    if (D > PdShape8::DMAX) return set(2, PdShape32::UNITS, PdShape32::KD, PdShape32::MC, PdShape32::AWS);
    const int want = lvsr_knob(LVSR_KNOB_DEC_CLUSTER);
    if (want != 8 && allow16 && set(1, PdShape16::UNITS, PdShape16::KD, PdShape16::MC, PdShape16::AWS)) return true;
    if (want == 16) return false;          // (allow16 = false: the caller found that clusters of 16 do not fit its LDS budget)
    return set(0, PdShape8::UNITS, PdShape8::KD, PdShape8::MC, PdShape8::AWS);
}

// Utterances per launch.  All of them when their clusters fit the chip at once; otherwise — when nothing couples the utterances of a
// batch (no window prior: the windows' centres are the one thing clusters of different utterances exchange) — the fewest equal
// passes that fit, launched back to back on the stream (per-GPU batches of 64 / 128: 2 / 4 passes of 32 utterances in clusters of 8
// instead of the step kernels).
static inline bool pd_pick_passes(int B, int D, int M, bool independent, PdPick& k, int& nb, bool allow16 = true) {
    nb = B;
    if (pd_pick(B, D, M, k, allow16)) return true;
    if (!independent) return false;
    for (int passes = 2; passes <= 8; ++passes) {
        nb = (B + passes - 1) / passes;
        if (pd_pick(nb, D, M, k, allow16)) return true;
    }
    return false;
}

__host__ __device__ __forceinline__ int pd_slot(int k, int KX) { return (k / KX) * (KX + 4) + (k % KX); }

// Window of label i (attdec_window) with the window centres taken from LDS: the centres of ALL utterances bound the window
// (lvsr/bricks/attention.py:133-147), which is the one coupling between clusters
__device__ __forceinline__ Win pd_window(const AttDec& a, int i, const float* posv) {
    if (a.K == 0 || a.prior_type == 0) return attdec_window(a, i);
    const float before = (float)a.p0, after = (float)a.p1;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int b = 0; b < a.B; ++b) {
        const float pb = posv[b];
        mn = fminf(mn, floorf(pb - before));
        mx = fmaxf(mx, ceilf(pb + after));
    }
    Win w;
    w.begin = (int)fmaxf(0.f, mn);
    w.end = (int)fminf((float)a.Tp, mx);
    if (w.end < w.begin) w.end = w.begin;
    return w;
}

// One sweep over a plane of n <= 512 granules until every granule carries `epoch`; thread tid gets granules tid and tid + 256.
// Returns false when the cluster gave up (spin limit / abort word).
__device__ __forceinline__ bool pd_gather(const u64* g, int n, unsigned epoch, int* abort_word, float (&out)[PD_NV]) {
    const int tid = threadIdx.x;
    u64 wv[PD_NV];
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) {
            wv[x] = (u64)epoch << 32;
            if (tid + x * PD_THREADS < n) wv[x] = __hip_atomic_load(g + tid + x * PD_THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int x = 0; x < PD_NV; ++x) ok = ok && (unsigned)(wv[x] >> 32) == epoch;
        if (__all(ok)) break;
        ++spins;
        if ((spins & 127u) == 0u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (spins > PERSIST_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
#pragma unroll
    for (int x = 0; x < PD_NV; ++x) out[x] = __uint_as_float((unsigned)wv[x]);
    return true;
}

// sum_x w[x] * v[q][x] over this thread's row slice of a sliced LDS vector, folded over the lanes of the unit
template <int KX, int KSPLIT>
__device__ __forceinline__ float pd_dot(const f32x2 (&w)[KX / 2], const float* buf, int q) {
    const float4* hv = (const float4*)(buf + q * (KX + 4));
    f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < KX / 4; ++x) {
        const float4 h4 = hv[x];
        const f32x2 lo = {h4.x, h4.y}, hi = {h4.z, h4.w};
        a0 = w[2 * x] * lo + a0;
        a1 = w[2 * x + 1] * hi + a1;
    }
    return group_sum<KSPLIT>((a0.x + a1.x) + (a0.y + a1.y));
}

// v[x] (x < 8) -> sum over the 64 lanes of the wave; lane l ends with the total of value index 4*bit5(l) + 2*bit4(l) + bit3(l):
// every stage halves the values a lane carries (10 shuffles instead of 8 x 6)
__device__ __forceinline__ float pd_butterfly8(float (&v)[8]) {
    const int lane = threadIdx.x & 63;
    // (named scalars and explicit selects: written as `up ? v[x + half] : v[x]` on the array, the compiler turned the lane-dependent
    // choice into a dynamically indexed private array — 48 bytes of scratch per lane in the content-only kernel)
    const bool up0 = (lane & 32) != 0, up1 = (lane & 16) != 0, up2 = (lane & 8) != 0;
    const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
    const float b0 = (up0 ? a4 : a0) + __shfl_xor(up0 ? a0 : a4, 32, 64);
    const float b1 = (up0 ? a5 : a1) + __shfl_xor(up0 ? a1 : a5, 32, 64);
    const float b2 = (up0 ? a6 : a2) + __shfl_xor(up0 ? a2 : a6, 32, 64);
    const float b3 = (up0 ? a7 : a3) + __shfl_xor(up0 ? a3 : a7, 32, 64);
    const float c0 = (up1 ? b2 : b0) + __shfl_xor(up1 ? b0 : b2, 16, 64);
    const float c1 = (up1 ? b3 : b1) + __shfl_xor(up1 ? b1 : b3, 16, 64);
    float r = (up2 ? c1 : c0) + __shfl_xor(up2 ? c0 : c1, 8, 64);
    r += lvsr_dpp_quad_xor1(r);
    r += lvsr_dpp_quad_xor2(r);
    r += lvsr_dpp_half_mirror(r);
    return r;
}
__device__ __forceinline__ int pd_butterfly_index(int lane) { return ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1); }

// Phase clock of work-group 0 (thread 0): accumulated s_memrealtime ticks (100 MHz) per phase of the label loop, left in the
// workspace header (bytes 64..255) for tools/probe_decoder_persist.py (LVSR_PD_PROF=1).  Every read drains the wave's
// outstanding LDS traffic, so the clock itself costs about a microsecond per label: off by default.
struct PdClock {          // accumulators in LDS (2 floats each), touched by thread 0 of work-group 0 only
    long long* acc;
    bool on;
    __device__ __forceinline__ void start(bool enable, float* mem) {
        on = enable;
        acc = (long long*)mem;
        if (on) {
            for (int x = 1; x <= PD_NPROF; ++x) acc[x] = 0;
            acc[0] = wall_clock64();
        }
    }
    __device__ __forceinline__ void mark(int slot) {
        if (on) {
            const long long now = wall_clock64();
            acc[1 + slot] += now - acc[0];
            acc[0] = now;
        }
    }
};

