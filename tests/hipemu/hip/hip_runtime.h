// TEST INFRASTRUCTURE ONLY — a host-side stand-in for <hip/hip_runtime.h>.
//
// The product kernels under attention-lvcsr_amd/csrc/ are plain HIP for gfx950 with no
// conditional compilation.  There is no GPU in the build container and only ~90 GPU-minutes per
// round, so tests/hipemu compiles THE SAME SOURCES for x86 with this header first on the include
// path (`clang++ -x c++ -I tests/hipemu`) and runs them on fibers: one fiber per work-item,
// work-groups executed one after another, wave64 collectives (__shfl*, MFMA 16x16x4 f32) and
// __syncthreads implemented as fiber barriers.  It checks indexing / algorithm logic of the kernel
// sources against the oracle at tiny sizes before GPU time is spent.  Nothing in the product loads it.
//
// Work-groups normally run one after another on the calling thread.  hipemu_set_concurrent(1) (tests/hipemu/
// hipemu_support.cpp) makes a launch of 2..64 work-groups run every work-group on its own OS thread (fibers inside),
// so kernels whose work-groups wait for each other (csrc/encoder_persist.hip: granule hand-offs) can be exercised:
// LDS (`__shared__`) and all per-block emulator state are thread_local, the agent-scope atomics are real atomics.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 { unsigned x, y, z; };

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

namespace hipemu {
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    bool done = false;
};
struct State {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int cur = -1;
    int nthreads = 0;
    // barrier bookkeeping: block barrier + one per wave
    int block_arrived = 0;
    unsigned block_gen = 0;
    std::vector<int> wave_arrived;
    std::vector<unsigned> wave_gen;
    std::vector<float> xa, xb;       // per-lane exchange slots (block-wide arrays indexed by tid)
    std::vector<double> xd;
    std::vector<long long> xl;
    std::function<void()> body;
};
inline State& st() { static thread_local State s; return s; }
inline int& concurrent_flag() { static int f = 0; return f; }
inline hipemu_uint3& tidx() { static thread_local hipemu_uint3 v; return v; }
inline hipemu_uint3& bidx() { static thread_local hipemu_uint3 v; return v; }
inline dim3& bdim() { static dim3 v; return v; }
inline dim3& gdim() { static dim3 v; return v; }
struct Saved { hipemu_uint3 t; };
inline std::vector<Saved>& saved() { static std::vector<Saved> v; return v; }

inline void yield() {
    State& s = st();
    int me = s.cur;
    swapcontext(&s.fibers[me].ctx, &s.sched);
}
inline void block_barrier() {
    State& s = st();
    unsigned gen = s.block_gen;
    if (++s.block_arrived == s.nthreads) { s.block_arrived = 0; s.block_gen++; return; }
    while (s.block_gen == gen) yield();
}
inline int lane_id() { return st().cur & 63; }
inline int wave_id() { return st().cur >> 6; }
inline int wave_size_here() {
    State& s = st();
    int w = wave_id();
    return std::min(64, s.nthreads - w * 64);
}
inline void wave_barrier() {
    State& s = st();
    int w = wave_id();
    unsigned gen = s.wave_gen[w];
    if (++s.wave_arrived[w] == wave_size_here()) { s.wave_arrived[w] = 0; s.wave_gen[w]++; return; }
    while (s.wave_gen[w] == gen) yield();
}
inline void trampoline() {
    State& s = st();
    s.body();
    s.fibers[s.cur].done = true;
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}
inline void run_block(int nthreads, const std::function<void()>& body, size_t STK = 256 * 1024) {
    State& s = st();
    s.nthreads = nthreads;
    s.body = body;
    s.block_arrived = 0;
    int nw = (nthreads + 63) / 64;
    s.wave_arrived.assign(nw, 0);
    s.wave_gen.assign(nw, 0);
    s.xa.assign(nthreads, 0.f); s.xb.assign(nthreads, 0.f); s.xd.assign(nthreads, 0.0); s.xl.assign(nthreads, 0);
    if ((int)s.fibers.size() < nthreads) s.fibers.resize(nthreads);
    for (int i = 0; i < nthreads; ++i) {
        Fiber& f = s.fibers[i];
        if (f.stack.size() != STK) f.stack.resize(STK);
        f.done = false;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = STK;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    int remaining = nthreads;
    dim3 bd = bdim();
    while (remaining > 0) {
        for (int i = 0; i < nthreads; ++i) {
            if (s.fibers[i].done) continue;
            s.cur = i;
            tidx().x = i % bd.x; tidx().y = (i / bd.x) % bd.y; tidx().z = i / (bd.x * bd.y);
            swapcontext(&s.sched, &s.fibers[i].ctx);
            if (s.fibers[i].done) --remaining;
        }
    }
    s.cur = -1;
}
template <typename F>
inline void launch(dim3 grid, dim3 block, F&& f) {
    gdim() = grid; bdim() = block;
    int nthreads = block.x * block.y * block.z;
    const unsigned nblocks = grid.x * grid.y * grid.z;
    if (concurrent_flag() && nblocks > 1 && nblocks <= 256) {      // (cluster grids are padded to multiples of 8 P: most of a large one exits at once)
        std::function<void()> body = f;
        std::vector<std::thread> workers;
        for (unsigned z = 0; z < grid.z; ++z)
            for (unsigned y = 0; y < grid.y; ++y)
                for (unsigned x = 0; x < grid.x; ++x)
                    workers.emplace_back([x, y, z, nthreads, &body]() {
                        bidx().x = x; bidx().y = y; bidx().z = z;
                        run_block(nthreads, body, 128 * 1024);
                    });
        for (auto& w : workers) w.join();
        return;
    }
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                bidx().x = x; bidx().y = y; bidx().z = z;
                run_block(nthreads, f);
            }
}
template <typename T> inline std::vector<T>& xslot();
template <> inline std::vector<float>& xslot<float>() { return st().xa; }
template <> inline std::vector<double>& xslot<double>() { return st().xd; }
template <> inline std::vector<long long>& xslot<long long>() { return st().xl; }
template <typename T, typename S>
inline T shfl_generic(T v, int src_lane) {
    State& s = st();
    int base = wave_id() * 64;
    auto& slot = xslot<S>();
    slot[s.cur] = (S)v;
    wave_barrier();
    int n = wave_size_here();
    T r = (src_lane >= 0 && src_lane < n) ? (T)slot[base + src_lane] : v;
    wave_barrier();
    return r;
}
}  // namespace hipemu

#define threadIdx (hipemu::tidx())
#define blockIdx (hipemu::bidx())
#define blockDim (hipemu::bdim())
#define gridDim (hipemu::gdim())
static const int warpSize = 64;

inline void __syncthreads() { hipemu::block_barrier(); }

inline float __shfl(float v, int lane, int width = 64) {
    int l = hipemu::lane_id();
    return hipemu::shfl_generic<float, float>(v, (l / width) * width + (lane % width));
}
inline float __shfl_xor(float v, int m, int width = 64) {
    int l = hipemu::lane_id();
    int src = l ^ m;
    if (src / width != l / width) src = l;
    return hipemu::shfl_generic<float, float>(v, src);
}
inline float __shfl_down(float v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l + (int)d;
    if (src / width != l / width) src = l;
    return hipemu::shfl_generic<float, float>(v, src);
}
inline float __shfl_up(float v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l - (int)d;
    if (src < 0 || src / width != l / width) src = l;
    return hipemu::shfl_generic<float, float>(v, src);
}
inline int __shfl(int v, int lane, int width = 64) {
    int l = hipemu::lane_id();
    return (int)hipemu::shfl_generic<long long, long long>(v, (l / width) * width + (lane % width));
}
inline int __shfl_xor(int v, int m, int width = 64) {
    int l = hipemu::lane_id();
    int src = l ^ m;
    if (src / width != l / width) src = l;
    return (int)hipemu::shfl_generic<long long, long long>(v, src);
}
inline int __shfl_down(int v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l + (int)d;
    if (src / width != l / width) src = l;
    return (int)hipemu::shfl_generic<long long, long long>(v, src);
}

// ---- agent-scope atomics / wave votes used by the persistent kernels.  Kernels that wait for OTHER work-groups need
// hipemu_set_concurrent(1); with work-groups run one after another they would spin until their own bound trips ----
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
// s_getreg_b32 hwreg(HW_REG_XCC_ID): the XCD a work-group runs on.  Here: block b 'runs' on XCD b % HIPEMU_XCDS (environment,
// default 1 = every cluster shares its XCD; 8 mimics MI355X's round-robin dispatch)
inline unsigned hipemu_xcc_id() {
    const char* e = getenv("HIPEMU_XCDS");
    const int n = e ? atoi(e) : 1;
    return n > 1 ? (unsigned)(blockIdx.x % (unsigned)n) : 0u;
}
inline unsigned __builtin_amdgcn_s_getreg(int) { return hipemu_xcc_id(); }
template <typename T, typename V> inline void hipemu_atomic_store(T* p, V v) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T> inline T hipemu_atomic_load(const T* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store(p, v)
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load(p)
inline void __builtin_amdgcn_s_sleep(int) { hipemu::yield(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier(); }      // lanes are fibers here, not lock-step
inline int __all(int pred) {
    int l = hipemu::lane_id();
    int acc = 1;
    for (int src = 0; src < hipemu::wave_size_here(); ++src) acc &= (hipemu::shfl_generic<long long, long long>(pred ? 1 : 0, src) != 0);
    (void)l;
    return acc;
}
// v_mov_b32 with a DPP control: quad_perm (ctrl < 0x100): lane l reads lane (l & ~3) + ((ctrl >> 2*(l&3)) & 3);
// row_ror:n (ctrl 0x121..0x12F): rotation by n lanes inside each row of 16; row_mirror / row_half_mirror (0x140 / 0x141)
inline int hipemu_mov_dpp(int v, int ctrl, int, int, bool) {
    int l = hipemu::lane_id();
    if (ctrl == 0x140) return (int)hipemu::shfl_generic<long long, long long>(v, (l & ~15) | (15 - (l & 15)));      // row_mirror
    if (ctrl == 0x141) return (int)hipemu::shfl_generic<long long, long long>(v, (l & ~7) | (7 - (l & 7)));        // row_half_mirror
    if (ctrl == 0x138) return (int)hipemu::shfl_generic<long long, long long>(v, l > 0 ? l - 1 : l);                 // wave_shr:1 (lane 0 keeps its value)
    if (ctrl == 0x13C) return (int)hipemu::shfl_generic<long long, long long>(v, (l + hipemu::wave_size_here() - 1) % hipemu::wave_size_here());   // wave_ror:1
    if (ctrl >= 0x121 && ctrl <= 0x12F) return (int)hipemu::shfl_generic<long long, long long>(v, (l & ~15) | ((l - (ctrl - 0x120)) & 15));
    if (ctrl >= 0x100) { fprintf(stderr, "hipemu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
    return (int)hipemu::shfl_generic<long long, long long>(v, (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3));
}
#define __builtin_amdgcn_mov_dpp hipemu_mov_dpp
// v_mov_b32 with a DPP control and an `old` operand for the lanes that have no source: wave_shr:1 (0x138: lane 0 keeps `old`),
// wave_ror:1 (0x13C: lane l reads lane (l - 1) mod 64)
inline int hipemu_update_dpp(int old, int v, int ctrl, int, int, bool) {
    int l = hipemu::lane_id();
    const int n = hipemu::wave_size_here();
    if (ctrl == 0x138) { const int got = (int)hipemu::shfl_generic<long long, long long>(v, l > 0 ? l - 1 : l); return l > 0 ? got : old; }
    if (ctrl == 0x13C) return (int)hipemu::shfl_generic<long long, long long>(v, (l + n - 1) % n);
    fprintf(stderr, "hipemu: update_dpp control 0x%x is not emulated\n", ctrl); abort();
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
// v_permlane16_swap_b32 vdst, src0 (gfx950): the odd rows (16 lanes) of vdst are exchanged with the even rows of src0;
// -> {new vdst, new src0}
struct hipemu_u32x2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline hipemu_u32x2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src0, bool, bool) {
    const int l = hipemu::lane_id();
    const bool odd = ((l >> 4) & 1) != 0;
    // lane in an odd row: its vdst <- src0 of the lane 16 below; lane in an even row: its src0 <- vdst of the lane 16 above
    const unsigned from_src_below = (unsigned)hipemu::shfl_generic<long long, long long>((long long)src0, odd ? l - 16 : l);
    const unsigned from_dst_above = (unsigned)hipemu::shfl_generic<long long, long long>((long long)vdst, odd ? l : l + 16);
    hipemu_u32x2 r;
    r.v[0] = odd ? from_src_below : vdst;
    r.v[1] = odd ? src0 : from_dst_above;
    return r;
}
// v_readlane_b32: the value lane `src` holds, for every lane (all lanes of the wave must reach the call)
inline int hipemu_readlane(int v, int src) { return (int)hipemu::shfl_generic<long long, long long>(v, src); }
#define __builtin_amdgcn_readlane hipemu_readlane
// v_readfirstlane_b32: only used on values that are already wave-uniform, so the lane's own value is the answer
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int src = 0; src < hipemu::wave_size_here(); ++src)
        if (hipemu::shfl_generic<long long, long long>(pred ? 1 : 0, src) != 0) m |= 1ull << src;
    return m;
}
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __shfl_up(int v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l - (int)d;
    if (src < 0 || src / width != l / width) src = l;
    return (int)hipemu::shfl_generic<long long, long long>(v, src);
}
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D: col=l&15, row=(l>>4)*4+reg;
// result = k-ordered fmaf chain (cdna_hip_programming.md §3).
inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    hipemu::State& s = hipemu::st();
    int base = hipemu::wave_id() * 64;
    int l = hipemu::lane_id();
    s.xa[s.cur] = a; s.xb[s.cur] = b;
    hipemu::wave_barrier();
    hipemu_f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(s.xa[base + row + 16 * k], s.xb[base + col + 16 * k], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5)
inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    hipemu::State& s = hipemu::st();
    int base = hipemu::wave_id() * 64;
    int l = hipemu::lane_id();
    s.xa[s.cur] = a; s.xb[s.cur] = b;
    hipemu::wave_barrier();
    hipemu_f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = std::fmaf(s.xa[base + row + 32 * k], s.xb[base + col + 32 * k], acc);
        d[r] = acc;
    }
    hipemu::wave_barrier();
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32

inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline float unsafeAtomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
// global_load_lds_dword: LDS address = wave-uniform base + lane * size
inline void __builtin_amdgcn_global_load_lds(const void* src, void* dst, unsigned size, unsigned off, unsigned aux) {
    (void)aux;
    memcpy((char*)dst + off + (threadIdx.x & 63) * size, src, size);
}
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline float atomicMax(int* p, int v) { int o = *p; *p = std::max(o, v); return o; }
inline int atomicMin(int* p, int v) { int o = *p; *p = std::min(o, v); return o; }

inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(16) int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float __expf(float x) { return std::exp(x); }
inline long long wall_clock64() { return 0; }
inline float __fdividef(float a, float b) { return a / b; }
inline float __frcp_rn(float a) { return 1.0f / a; }
inline float __builtin_amdgcn_rcpf(float a) { return 1.0f / a; }
inline float __builtin_amdgcn_exp2f(float a) { return std::exp2(a); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                         \
    do {                                                                                      \
        hipemu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); });             \
    } while (0)

inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }      // an MI355X's CU count
