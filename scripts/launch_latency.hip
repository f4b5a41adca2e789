// How much does one dependent kernel in a hipGraph chain cost on MI355X, as a function of what the kernel does?
// (empty / load-modify-store of the predecessor's output / + a block barrier / + a second dependent load).
// Measurement tool for DESIGN.md's launch-count arithmetic; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty(const float* in, float* out) {}
__global__ void k_copy(const float* in, float* out) { int i = blockIdx.x * blockDim.x + threadIdx.x; out[i] = in[i] + 1.0f; }
__global__ void k_copy_sync(const float* in, float* out) {
    __shared__ float s[256];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    s[threadIdx.x] = in[i];
    __syncthreads();
    out[i] = s[(threadIdx.x + 1) & 255] + 1.0f;
}
__global__ void k_copy2(const float* in, float* out, const int* idx) {   // two dependent global loads
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = idx[i];
    out[i] = in[j] + 1.0f;
}

// straight-line code of ~NI instructions (8 bytes each) executed once per wave: does a cold instruction stream cost time?
template <int ID, int NI>
__global__ void k_bigcode(const float* in, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[i];
#pragma unroll
    for (int j = 0; j < NI; ++j) v = v * (1.0f + 1e-7f * (float)((j * 7 + ID) % 13)) + 1e-9f * (float)(ID + 1);
    out[i] = v;
}
template <int NI>
void launch_big(int id, int blocks, hipStream_t s, const float* a, float* b) {
    switch (id & 7) {
        case 0: hipLaunchKernelGGL((k_bigcode<0, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 1: hipLaunchKernelGGL((k_bigcode<1, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 2: hipLaunchKernelGGL((k_bigcode<2, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 3: hipLaunchKernelGGL((k_bigcode<3, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 4: hipLaunchKernelGGL((k_bigcode<4, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 5: hipLaunchKernelGGL((k_bigcode<5, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        case 6: hipLaunchKernelGGL((k_bigcode<6, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
        default: hipLaunchKernelGGL((k_bigcode<7, NI>), dim3(blocks), dim3(256), 0, s, a, b); break;
    }
}

template <class F>
double run_chain(hipStream_t s, int n, F launch) {
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

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 400;
    for (int blocks : {32, 256, 1024}) {
        const int N = blocks * 256;
        float *a, *b; int* idx;
        CK(hipMalloc(&a, N * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&idx, N * 4));
        CK(hipMemset(a, 0, N * 4)); CK(hipMemset(b, 0, N * 4));
        int* h = (int*)malloc(N * 4); for (int i = 0; i < N; ++i) h[i] = (i * 7919) % N;
        CK(hipMemcpy(idx, h, N * 4, hipMemcpyHostToDevice));
        double t0 = run_chain(s, n, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double t1 = run_chain(s, n, [&](int i) { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double t2 = run_chain(s, n, [&](int i) { hipLaunchKernelGGL(k_copy_sync, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double t3 = run_chain(s, n, [&](int i) { hipLaunchKernelGGL(k_copy2, dim3(blocks), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, idx); });
        printf("blocks %4d: empty %.2f us | load+store %.2f us | +barrier %.2f us | 2 dependent loads %.2f us   (per kernel in a %d-kernel graph chain)\n",
               blocks, t0, t1, t2, t3, n);
        double b1 = run_chain(s, n, [&](int i) { launch_big<500>(0, blocks, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double b8 = run_chain(s, n, [&](int i) { launch_big<500>(i, blocks, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double c1 = run_chain(s, n, [&](int i) { launch_big<2000>(0, blocks, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        double c8 = run_chain(s, n, [&](int i) { launch_big<2000>(i, blocks, s, (i & 1) ? b : a, (i & 1) ? a : b); });
        printf("             4 KB straight-line code: same kernel %.2f us, 8 alternating kernels %.2f us | 16 KB code: same %.2f us, 8 alternating (128 KB > I-cache) %.2f us\n",
               b1, b8, c1, c8);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(idx)); free(h);
    }
    return 0;
}
