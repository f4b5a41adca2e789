// Is a latency-bound kernel on MI355X bound by instruction FETCH?  Every launch starts with a cold instruction cache (the
// dispatch's acquire invalidates it), so code that a wave executes once is fetched line by line from L2.  Kernels of NI
// independent fp32 FMAs (8 accumulators round-robin: no dependent-issue stall) as straight-line code vs the same dynamic count
// as a loop over a 64-instruction body, 1 / 2 / 4 waves per SIMD, in a dependent hipGraph chain.  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O3 scripts/icache_probe.hip -o build/icache_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NI>
__global__ void k_straight(const float* in, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8];
    const float v = in[i];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = v + (float)q;
#pragma unroll
    for (int j = 0; j < NI; ++j) a[j & 7] = __builtin_fmaf(a[j & 7], 1.0f + 1e-7f * (float)((j * 7) % 13 + 1), 1e-9f * (float)(j % 5 + 1));
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += a[q];
    out[i] = s;
}
template <int NI>
__global__ void k_looped(const float* in, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8];
    const float v = in[i];
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = v + (float)q;
#pragma unroll 1
    for (int r = 0; r < NI / 64; ++r) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a[j & 7] = __builtin_fmaf(a[j & 7], 1.0f + 1e-7f * (float)((j * 7) % 13 + 1), 1e-9f * (float)(j % 5 + 1));
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += a[q];
    out[i] = s;
}

template <class F>
double chain_us(hipStream_t s, int n, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 1e3 * ms / (5.0 * n);
}

template <int NI>
void one(hipStream_t s, float* a, float* b, int blocks, int threads) {
    const int n = 200;
    const double ts = chain_us(s, n, [&](int i) { hipLaunchKernelGGL((k_straight<NI>), dim3(blocks), dim3(threads), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
    const double tl = chain_us(s, n, [&](int i) { hipLaunchKernelGGL((k_looped<NI>), dim3(blocks), dim3(threads), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
    printf("   %5d FMAs (~%3d KB straight-line): straight %7.2f us   looped %7.2f us   -> %.1f ns per 64-byte line of extra straight-line code\n",
           NI, NI * 8 / 1024, ts, tl, NI > 64 ? 1e3 * (ts - tl) / ((NI - 64) * 8 / 64.0) : 0.0);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 1024 * 1024;
    float *a, *b; CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4));
    CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4));
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int blocks = cfg == 3 ? 32 : 128, threads = cfg == 0 ? 256 : cfg == 1 ? 512 : cfg == 2 ? 1024 : 256;
        printf("%d workgroups x %d threads (%d wave(s) per SIMD):\n", blocks, threads, threads / 256);
        one<64>(s, a, b, blocks, threads);
        one<256>(s, a, b, blocks, threads);
        one<512>(s, a, b, blocks, threads);
        one<1024>(s, a, b, blocks, threads);
        one<2048>(s, a, b, blocks, threads);
        one<4096>(s, a, b, blocks, threads);
    }
    return 0;
}
