// Pieces shared by the persistent cluster kernels (encoder_persist.hip, decoder_persist.hip): the {epoch,value} granule
// hand-off, cluster numbering, lane-group sums.
#pragma once
#include "common.h"

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PERSIST_SPIN_LIMIT (1u << 21)
#define PERSIST_MAX_WG 224

// The work-groups of a cluster are numbered so that they land on ONE XCD (block b runs on XCD b % 8: observed on MI355X,
// not promised by HIP — nothing depends on it but speed: 2.7 instead of 3.2 us per step at H = 256).
// Experiment switches (LVSR_PERSIST_FLAGS, read per call): 1 = do not write the saved tensors (timing only, results unusable
// for BPTT), 2 = consecutive blocks form a cluster instead (members spread over the XCDs), 4 = publish with plain stores
// (they stay in the XCD's L2: only correct when the XCD placement holds; measured SLOWER than write-through sc1 stores)
#define PF_NOSAVE 1
#define PF_SPREAD 2
#define PF_PLAIN 4
#define PF_NOWAIT 8      // ablation: take whatever the first sweep returns (wrong results; what the step costs without hand-off waits)
#define PF_NODOT 16      // ablation: skip the contractions (wrong results; what the hand-offs cost alone)
#define PF_STAGE 64      // encoder_persist1.hip: the next step's operands are fetched by the non-polling waves and staged in LDS
#define PF_PRIVATE 32    // every wave sweeps the whole vector into a buffer of its own, no work-group barrier (RB = 1 only):
                         // measured slower, 2.69 vs 2.45 us per step — four times the sc1 loads in the CU's memory queue
__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v, int flags = 0) {
    const u64 w = ((u64)epoch << 32) | (u64)__float_as_uint(v);
    if (flags & PF_PLAIN) *(volatile u64*)p = w;
    else __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// cluster / member index of a work-group
__device__ __forceinline__ void cluster_of_block(int P, int flags, int& cl, int& p) {
    const int b = blockIdx.x, ncl = gridDim.x / P;
    if (!(flags & PF_SPREAD) && ncl % 8 == 0) { cl = (b % 8) + 8 * (b / (8 * P)); p = (b / 8) % P; }
    else { cl = b / P; p = b % P; }
}

// sum over the KSPLIT adjacent lanes that share a unit; every lane of the group gets the total
template <int KSPLIT>
__device__ __forceinline__ float group_sum(float v) {
    if (KSPLIT >= 2) v += lvsr_dpp_quad_xor1(v);
    if (KSPLIT >= 4) v += lvsr_dpp_quad_xor2(v);
    if (KSPLIT >= 8) v += lvsr_dpp_half_mirror(v);          // all DPP: the lane groups are aligned to their size
    if (KSPLIT >= 16) v += lvsr_dpp_mirror(v);
    return v;
}
