// Standalone latency probe (not part of the product): what does one dependent step-kernel cost on MI355X?
//   hipcc --offload-arch=gfx950 -O3 -o boundary_probe boundary_probe.hip && ./boundary_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// variant 0: empty; 1: read prev output (16 rows x K) + write own slice; 2: + weight tile loads; 3: + 16 MFMAs + LDS reduce
template <int V>
__global__ __launch_bounds__(256) void step(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w, int K) {
    if (V == 0) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.f; return; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int Kw = K / 4, k0 = wave * Kw + kk * (Kw / 4);
    float4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *(const float4*)(in + (size_t)i * K + k0 + 4 * q);
    if (V >= 2) {
        const float4* p = (const float4*)w + ((size_t)(blockIdx.x * 4 + wave) * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = p[q * 64];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float v;
    if (V >= 3) {
        __shared__ float red[4][16][17];
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc1, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc0[r] + acc1[r];
        __syncthreads();
        const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
        v = red[0][row][col] + red[1][row][col] + red[2][row][col] + red[3][row][col];
    } else {
        v = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) v += a[q].x * b[q].x + a[q].y * b[q].y + a[q].z * b[q].z + a[q].w * b[q].w;
    }
    const int row = threadIdx.x >> 4, col = blockIdx.x * 16 + (threadIdx.x & 15);
    if (V >= 4) {
        // epilogue like the real gates kernel: 2 extra loads, sigmoid, 3 stores to different arrays
        const float g = w[(size_t)row * K + col] + in[(size_t)row * K + col];
        const float sg = 1.0f / (1.0f + expf(-(v + g)));
        if (col < K) {
            out[(size_t)row * K + col] = sg * 1e-3f;
            out[(size_t)(16 + row) * K + col] = sg * 2e-3f;
            out[(size_t)(32 + row) * K + col] = sg * 3e-3f;
        }
        return;
    }
    if (col < K) out[(size_t)row * K + col] = v * 1e-3f;
}

struct Big { const float* in; float* out; const float* w; int K; int pad[9]; const float* p2[8]; float* q2[8]; int tail[6]; };
__global__ __launch_bounds__(256) void step_big(Big a, int n) {
    // same body as variant 3 but operands come from a 200-byte by-value struct, pointers picked from its far end
    const float* in = (n & 1) ? a.p2[7] : a.p2[6];
    float* out = (n & 1) ? a.q2[6] : a.q2[7];
    const float* w = a.w; const int K = a.K + a.tail[5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, kk = lane >> 4;
    const int Kw = K / 4, k0 = wave * Kw + kk * (Kw / 4);
    float4 av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) av[q] = *(const float4*)(in + (size_t)i * K + k0 + 4 * q);
    const float4* p = (const float4*)w + ((size_t)(blockIdx.x * 4 + wave) * 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = p[q * 64];
    __shared__ float red[4][16][17];
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].x, bv[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].y, bv[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].z, bv[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q].w, bv[q].w, acc1, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc0[r] + acc1[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, col = blockIdx.x * 16 + (threadIdx.x & 15);
    const float v = red[0][row][threadIdx.x & 15] + red[1][row][threadIdx.x & 15] + red[2][row][threadIdx.x & 15] + red[3][row][threadIdx.x & 15];
    if (col < K) out[(size_t)row * K + col] = v * 1e-3f;
}

double run_big(hipStream_t s, int nblk, float* bufA, float* bufB, float* w, int K, int steps) {
    Big a; memset(&a, 0, sizeof(a));
    a.w = w; a.K = K; a.p2[6] = bufA; a.p2[7] = bufB; a.q2[6] = bufA; a.q2[7] = bufB;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    for (int n = 0; n < steps; ++n) hipLaunchKernelGGL(step_big, dim3(nblk), dim3(256), 0, s, a, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    double best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(exec, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    return best * 1e3 / steps;
}

template <int V>
double run(hipStream_t s, int nblk, float* bufA, float* bufB, float* w, int K, int steps, bool graph) {
    auto enqueue = [&]() {
        for (int n = 0; n < steps; ++n) {
            hipLaunchKernelGGL(step<V>, dim3(nblk), dim3(256), 0, s, (n & 1) ? bufB : bufA, (n & 1) ? bufA : bufB, w, K);
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraphExec_t exec = nullptr;
    if (graph) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        enqueue();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    }
    double best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0, s));
        if (graph) CK(hipGraphLaunch(exec, s)); else enqueue();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    return best * 1e3 / steps;
}

int main() {
    const int K = 256, steps = 2000;
    hipStream_t s; CK(hipStreamCreate(&s));
    float *A, *B, *W;
    CK(hipMalloc(&A, 64 * K * 4)); CK(hipMalloc(&B, 64 * K * 4)); CK(hipMalloc(&W, 64 * 4 * 4 * 64 * 16));
    CK(hipMemset(A, 0, 64 * K * 4)); CK(hipMemset(B, 0, 64 * K * 4)); CK(hipMemset(W, 0, 64 * 4 * 4 * 64 * 16));
    for (int nblk : {16, 64}) {
        for (int g = 0; g < 2; ++g) {
            printf("blocks=%2d graph=%d  empty %.2f us | dep-load %.2f us | +weights %.2f us | +mfma %.2f us\n", nblk, g,
                   run<0>(s, nblk, A, B, W, K, steps, g), run<1>(s, nblk, A, B, W, K, steps, g),
                   run<2>(s, nblk, A, B, W, K, steps, g), run<3>(s, nblk, A, B, W, K, steps, g));
            if (g) printf("                      mfma variant with a 200-byte by-value kernarg struct %.2f us\n", run_big(s, nblk, A, B, W, K, steps));
            printf("                      +epilogue(2 loads, sigmoid, 3 stores) %.2f us\n", run<4>(s, nblk, A, B, W, K, steps, g));
        }
    }
    return 0;
}
