// Where does the sampler's time go?  k_sample (lm_kernels.h) on random bf16 logits of the two real vocabulary sizes, 32 sessions,
// as a dependent chain in one stream: greedy (argmax only), the production path (top-k + counter RNG) and the supplied-noise path
// (softmax, ordered compaction, rank), with and without the next micro-step's input row.  Measurement tool only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Imoshi_amd/csrc scripts/sample_microbench.hip moshi_amd/csrc/api_common.hip -o build/sample_microbench
#include <hip/hip_runtime.h>
__device__ long long g_stamps[16];
// stage stamps of workgroup 0 / thread 0 (shader clock), read back after the run
#define MMI_SAMPLE_STAMP(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = (long long)__builtin_readcyclecounter();
#include "lm_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill_logits(uint16_t* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    unsigned h2 = h * 0x9E3779B9u; h2 ^= h2 >> 16;
    // sum of two uniforms: a bell-ish spread of logits around 0
    float f = (((float)(h & 0xffff) + (float)(h2 & 0xffff)) / 65536.0f - 1.0f) * scale;
    p[i] = mmi_f32_to_bf16(f);
}
__global__ void k_touch(uint16_t* p, int n) {      // the "previous kernel": rewrites a few logits so that the sampler's input is freshly written
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[(size_t)i * 64] ^= 0;
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

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int B = 32, n = 200;
    for (int V : {2048, 32000}) {
        const int k = V == 2048 ? 250 : 25;
        uint16_t *logits, *pre, *emb, *xo; int *out, *use_noise, *forced, *use_forced; float* noise; unsigned long long* rng;
        CK(hipMalloc(&logits, (size_t)B * V * 2)); CK(hipMalloc(&pre, (size_t)B * 8 * 1024 * 2)); CK(hipMalloc(&emb, (size_t)(V + 1) * 1024 * 2));
        CK(hipMalloc(&xo, (size_t)64 * 1024 * 2)); CK(hipMalloc(&out, B * 4)); CK(hipMalloc(&use_noise, 4)); CK(hipMalloc(&forced, B * 4));
        CK(hipMalloc(&use_forced, 4)); CK(hipMalloc(&noise, (size_t)B * 256 * 4)); CK(hipMalloc(&rng, 16));
        CK(hipMemset(use_forced, 0, 4)); CK(hipMemset(forced, 0, B * 4)); CK(hipMemset(pre, 0, (size_t)B * 8 * 1024 * 2));
        CK(hipMemset(emb, 0, (size_t)(V + 1) * 1024 * 2)); CK(hipMemset(rng, 0, 16));
        std::vector<float> hn((size_t)B * 256, 1.0f); CK(hipMemcpy(noise, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
        k_fill_logits<<<(B * V + 255) / 256, 256, 0, s>>>(logits, (size_t)B * V, 99u, 4.0f);
        CK(hipStreamSynchronize(s));
        for (int mode = 0; mode < 3; ++mode) {          // 0 greedy, 1 production, 2 supplied noise
            for (int nx = 0; nx < 2; ++nx) {
                SampleArgs a; memset(&a, 0, sizeof(a));
                a.logits = logits; a.ld = V; a.V = V; a.k = k; a.temp = 0.8f; a.use_sampling = mode != 0;
                a.noise = noise; a.noise_ld = 256; a.use_noise = use_noise; a.rng = rng; a.site = 1; a.out = out; a.out_stride = 1; a.B = B;
                a.forced = forced; a.forced_stride = 1; a.use_forced = use_forced;
                if (nx) { a.nx_pre = pre; a.nx_ld = 8 * 1024; a.nx_emb = emb; a.nx_out = xo; a.nx_D = 1024; a.nx_T = 32; a.nx_ksteps = 64; }
                int un = mode == 2; CK(hipMemcpy(use_noise, &un, 4, hipMemcpyHostToDevice));
                const double t = chain_us(s, n, [&](int) {
                    hipLaunchKernelGGL(k_touch, dim3(B), dim3(64), 0, s, logits, B * 8);
                    if (V <= 2048) hipLaunchKernelGGL((k_sample<256, 8, true>), dim3(B), dim3(256), 0, s, a);
                    else hipLaunchKernelGGL((k_sample<1024, 32, true>), dim3(B), dim3(1024), 0, s, a);
                });
                const double t0 = chain_us(s, n, [&](int) { hipLaunchKernelGGL(k_touch, dim3(B), dim3(64), 0, s, logits, B * 8); });
                printf("V %5d k %3d  %-14s %s: %.2f us per (touch + sample) pair, touch alone %.2f us -> sampler %.2f us\n", V, k,
                       mode == 0 ? "greedy" : mode == 1 ? "production" : "supplied noise", nx ? "+ next input" : "            ", t, t0, t - t0);
            }
        }
        {   // stage stamps of the production path (the last production run was mode 1 nx 1; rerun it once)
            SampleArgs a; memset(&a, 0, sizeof(a));
            a.logits = logits; a.ld = V; a.V = V; a.k = k; a.temp = 0.8f; a.use_sampling = 1;
            a.noise = noise; a.noise_ld = 256; a.use_noise = use_noise; a.rng = rng; a.site = 1; a.out = out; a.out_stride = 1; a.B = B;
            a.forced = forced; a.forced_stride = 1; a.use_forced = use_forced;
            a.nx_pre = pre; a.nx_ld = 8 * 1024; a.nx_emb = emb; a.nx_out = xo; a.nx_D = 1024; a.nx_T = 32; a.nx_ksteps = 64;
            int un = 0; CK(hipMemcpy(use_noise, &un, 4, hipMemcpyHostToDevice));
            for (int r = 0; r < 3; ++r) {
                hipLaunchKernelGGL(k_touch, dim3(B), dim3(64), 0, s, logits, B * 8);
                if (V <= 2048) hipLaunchKernelGGL((k_sample<256, 8, true>), dim3(B), dim3(256), 0, s, a);
                else hipLaunchKernelGGL((k_sample<1024, 32, true>), dim3(B), dim3(1024), 0, s, a);
                CK(hipStreamSynchronize(s));
                long long st[16]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st)));
                printf("   stamps (cycles since entry):");
                for (int i = 1; i <= 12; ++i) printf(" %d:%lld", i, st[i] - st[0]);
                printf("\n");
            }
        }
        int tok[32]; CK(hipMemcpy(tok, out, B * 4, hipMemcpyDeviceToHost));
        printf("   tokens of the last run: %d %d %d %d ...\n", tok[0], tok[1], tok[2], tok[3]);
    }
    return 0;
}
