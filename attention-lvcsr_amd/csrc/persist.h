// Pieces shared by the persistent cluster kernels (encoder_persist.hip, decoder_persist.hip): the {epoch,value} granule
// hand-off, cluster numbering, lane-group sums.
#pragma once
#include "common.h"

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PERSIST_SPIN_LIMIT (1u << 21)
// (how many work-groups a cluster launch may have: lvsr_max_cluster_wgs(), common.h — derived from the device's CU count)

// The work-groups of a cluster are numbered so that they land on ONE XCD (block b runs on XCD b % 8: observed on MI355X,
// not promised by HIP — nothing depends on it but speed).  Whether they did is CHECKED per launch (cluster_shares_xcd): the
// members exchange their XCC_ID once, and only a cluster that shares an XCD publishes with plain stores (the granules then
// live in the XCD's L2, which the members' sc1 polls read: 0.91 instead of 1.33 us per exchange, profiles/r03_hop_probe.txt);
// any other cluster keeps write-through (sc1) stores, which are correct under every placement.
// Kernel-variant switches (flags argument of the kernels; LVSR_KNOB_PERSIST_FLAGS): every one of them computes the SAME results —
// 2 = consecutive blocks form a cluster instead (members spread over the XCDs), 4 = write-through stores even when the cluster
// shares an XCD (the round-2 hand-off), ...
#define PF_SPREAD 2
#define PF_SC1 4
#define PF_NARROW 64     // encoder, 128 < H <= 256: clusters of 4 work-groups (64 units each) instead of 8 — see persist_geom
#define PF_NOSTAGE 256      // forward kernel: no loader waves (every owner lane fetches its next operands itself, the round-2 form)
#define PF_NOLD9F 8192      // forward kernel, one utterance per cluster of 8: waves 4..7 stage the operands (round 3-5) instead of a ninth wave
#define PF_NOLD9 4096       // BPTT kernel, one utterance per cluster of 8: no ninth (loader) wave — the owners fetch their operands themselves
#define PF_NOYMPRE 32768    // decoder reverse walk: the label mask fetched at the top of its label (rounds 3-6) instead of one label ahead
#define PF_DPAL 16384       // decoder reverse walk: dPA summed in LDS and written once behind the loop instead of by L2 atomics label by label (opt-in:
                            // -0.3 % per WSJ-base step, +3 to +30 % on the reverse walk of every other shape measured; decoder_persist_bwd.hip)
#define PF_NOUB 2048        // forward kernel at 256 < H <= 512: one unit per lane group instead of four (see enc_pfwd_ub_kernel)
#define PF_PRIVATE 32    // every wave sweeps the whole vector into a buffer of its own, no work-group barrier (RB = 1 only):
                         // measured slower, 2.69 vs 2.45 us per step — four times the sc1 loads in the CU's memory queue
// Timing ablations that produce WRONG results (what a step costs without its waits / contractions / operand fetches / saved
// tensors).  They exist only in a probe build of the library (`python csrc/build.py --probes` -> liblvsr_hip_probes.so, for
// tools/probe_persist.py): in the product build the bits are 0, the branches fold away and lvsr_set_knob refuses them.
#ifdef LVSR_PROBES
#define PF_NOSAVE 1
#define PF_NOWAIT 8
#define PF_NODOT 16
#define PF_NOPREFETCH 128
#else
#define PF_NOSAVE 0
#define PF_NOWAIT 0
#define PF_NODOT 0
#define PF_NOPREFETCH 0
#endif
#define PF_WRONG_RESULT_BITS (1 | 8 | 16 | 128)
// plain = true: a store without cache-policy bits (wavefront-scope atomic = global_store_dwordx2; NOT a volatile store, which
// is emitted as flat_store sc0 sc1): it is visible to the other CUs of the SAME XCD once it reaches the XCD's L2 (the vector
// L1 is write-through) — only for granules whose every reader was verified to share the XCD.
__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v, bool plain = false) {
    const u64 w = ((u64)epoch << 32) | (u64)__float_as_uint(v);
    if (plain) __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Do the P work-groups of this cluster run on one XCD?  Member p publishes its XCC_ID (hardware register 20, bits 3:0) as a
// write-through granule {1, id} into hello[p] (zeroed by the launch's memset node) and wave 0 polls the P slots; every member
// sees the same P values, so the answer is uniform over the cluster.  Called once per launch by all threads (it contains a
// work-group barrier).  Returns false as well when a member does not show up within the spin limit (the abort word is raised;
// the first exchange of the sequence then makes everybody leave).
__device__ __forceinline__ bool cluster_shares_xcd(u64* hello, int P, int p, int* abort_word) {
    __shared__ int same_xcd;
    if (threadIdx.x < 64) {
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;       // hwreg(HW_REG_XCC_ID, 0, 4)
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(hello + p, ((u64)1 << 32) | (u64)xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        unsigned spins = 0;
        for (;;) {
            u64 w = ((u64)1 << 32) | (u64)xcc;
            if (lane < P) w = __hip_atomic_load(hello + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(w >> 32) == 1u)) { ok = __all(((unsigned)w & 15u) == xcc); break; }
            if (++spins > PERSIST_SPIN_LIMIT || ((spins & 127u) == 0u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        if (lane == 0) same_xcd = ok ? 1 : 0;
    }
    __syncthreads();
    return same_xcd != 0;
}
// Launch grid for ncl clusters of P work-groups, and the cluster / member index of a work-group in it.  Work-groups are dealt to
// the XCDs round-robin (block b on XCD b % 8), so cluster c = (b % 8) + 8 (b / 8P) takes the P blocks of its XCD column: every
// cluster on ONE XCD for ANY number of clusters — the grid is padded to a multiple of 8 P and the work-groups of the padding
// (cluster index >= ncl: the reference's batch of 10, a single decoding utterance) return at once.  Returns false for those.
static inline int cluster_grid(int ncl, int P, int flags) { return (flags & PF_SPREAD) ? ncl * P : ((ncl + 7) / 8) * 8 * P; }
__device__ __forceinline__ bool cluster_of_block(int P, int ncl, int flags, int& cl, int& p) {
    const int b = blockIdx.x;
    if (flags & PF_SPREAD) { cl = b / P; p = b % P; }
    else { cl = (b % 8) + 8 * (b / (8 * P)); p = (b / 8) % P; }
    return cl < ncl;
}

// sum over the KSPLIT adjacent lanes that share a unit; every lane of the group gets the total
template <int KSPLIT>
__device__ __forceinline__ float group_sum(float v) {
    if (KSPLIT >= 2) v += lvsr_dpp_quad_xor1(v);
    if (KSPLIT >= 4) v += lvsr_dpp_quad_xor2(v);
    if (KSPLIT >= 8) v += lvsr_dpp_half_mirror(v);          // all DPP: the lane groups are aligned to their size
    if (KSPLIT >= 16) v += lvsr_dpp_mirror(v);
    if (KSPLIT >= 32) v = lvsr_swap16_sum(v);
    return v;
}
