// Microbenchmark of the weight-streaming skinny GEMM (moshi_amd/csrc/lm_kernels.h: k_gemm_xp) on the Moshi-7B
// layer shapes: sweeps workgroup shapes (waves per workgroup, n-tiles per wave, fragments in flight) and prints the
// achieved weight-streaming rate.  Weights cycle through enough distinct buffers (> 1 GiB) that the 256 MiB
// Infinity Cache cannot serve them.  Measurement tool only - the product picks its plan in lm_engine.hip:plan_gemm.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Imoshi_amd/csrc scripts/gemm_microbench.hip \
//         moshi_amd/csrc/api_common.hip -o gpurun_out/gemm_microbench && gpurun_out/gemm_microbench [B]
#include "lm_kernels.h"

#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

__global__ void k_fill_rand_bf16(uint16_t* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        float f = ((float)(h & 0xffff) / 65536.0f - 0.5f) * 0.05f;
        p[i] = mmi_f32_to_bf16(f);
    }
}

// speed-of-light reference: stream the same bytes with 16-byte non-temporal loads and nothing else
__global__ __launch_bounds__(256) void k_stream_read(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = mmi_load_nt(p + i), b = mmi_load_nt(p + i + stride), c = mmi_load_nt(p + i + 2 * stride), d = mmi_load_nt(p + i + 3 * stride);
        acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    for (; i < n16; i += stride) acc ^= mmi_load_nt(p + i)[0];
    if (acc == 0x12345678u) *sink = acc;
}

// What do the activation-fragment loads cost?  The production main loop (one n-tile per workgroup, K split over 8 waves,
// U fragments per buffer) with XMODE 0 = activation fragments from global/L2 as in k_gemm_xp, 1 = no activation loads at
// all (a constant register): the difference is the price of the operand traffic.  Trivial epilogue (no LDS reduction).
template <int WAVES, int U, int XMODE>
__global__ __launch_bounds__(WAVES * 64) void k_probe_xcost(const u32x4* __restrict__ wpk, const u32x4* __restrict__ xpk, int KSTEPS, float* sink) {
    typedef float acc_t __attribute__((ext_vector_type(16)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int kper = KSTEPS / WAVES, ks0 = wave * kper;
    const u32x4* wp = wpk + ((long)blockIdx.x * KSTEPS + ks0) * 64 + lane;
    const u32x4* xp = xpk + (long)ks0 * 64 + lane;
    acc_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 wA[U], xA[U], wB[U], xB[U];
    const u32x4 xc = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#define P_LOAD(W_, X_, base) _Pragma("unroll") for (int u = 0; u < U; ++u) { W_[u] = mmi_load_nt(wp + ((base) + u) * 64); X_[u] = XMODE == 0 ? xp[((base) + u) * 64] : xc; }
#define P_MMA(W_, X_) _Pragma("unroll") for (int u = 0; u < U; ++u) acc = mmi_mfma_bf16_32x32x16(W_[u], X_[u], acc);
    const int nfull = kper / U;
    P_LOAD(wA, xA, 0);
    int g = 0;
    for (; g + 2 < nfull; g += 2) { P_LOAD(wB, xB, (g + 1) * U); P_MMA(wA, xA); P_LOAD(wA, xA, (g + 2) * U); P_MMA(wB, xB); }
    if (nfull - g == 2) { P_LOAD(wB, xB, (g + 1) * U); P_MMA(wA, xA); P_MMA(wB, xB); } else { P_MMA(wA, xA); }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

struct Shape { const char* name; int N, K, gate; };

typedef void (*launch_fn)(dim3 groups, hipStream_t s, const GemmArgs& a);
struct Variant { const char* name; int TN, MT, NTW, WAVES, U; launch_fn fn; };

template <int TN, int MT, int NTW, int WAVES, int U>
void launch_v(dim3 groups, hipStream_t s, const GemmArgs& a) {
    hipLaunchKernelGGL((k_gemm_xp<TN, MT, NTW, WAVES, U>), groups, dim3(WAVES * 64), 0, s, a);
}
#define V(TN, MT, NTW, WAVES, U) {#TN "x" #MT " ntw" #NTW " w" #WAVES " u" #U, TN, MT, NTW, WAVES, U, launch_v<TN, MT, NTW, WAVES, U>}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32;
    const int ksplit = argc > 2 ? atoi(argv[2]) : 1;   // > 1: the residual GEMMs run split-K with fp32 partial output
    const bool quick = argc > 3;                       // any 3rd argument: only the FFN linear_in shape with the production plan
                                                       // (for rocprofv3 --pmc passes, where every dispatch is serialised)
    const int T = getenv("MB_T") ? atoi(getenv("MB_T")) : (B <= 16 ? 16 : 32);   // MB_T=16 with B=32: 16-row weight tiles, two batch tiles
    const int MT = (B + T - 1) / T;
    const Shape shapes[] = {
        {"ffn_in  22528x4096 (gate)", 11264, 4096, 1}, {"in_proj 12288x4096", 12288, 4096, 0},
        {"ffn_out 4096x11264", 4096, 11264, 0},        {"out_proj 4096x4096", 4096, 4096, 0},
        {"text_linear 32000x4096", 32000, 4096, 0},    {"dep ffn_in 5632x1024 (gate)", 2816, 1024, 1},
        {"dep in_proj 3072x1024", 3072, 1024, 0},      {"dep ffn_out 1024x2816", 1024, 2816, 0},
        {"dep out_proj 1024x1024", 1024, 1024, 0},
    };
    const Variant variants[] = {
        V(32, 1, 1, 4, 4), V(32, 1, 1, 8, 4), V(32, 1, 1, 16, 4), V(32, 1, 2, 4, 4), V(32, 1, 2, 8, 4),
        V(32, 1, 1, 4, 8), V(32, 1, 1, 8, 8), V(32, 1, 2, 4, 2), V(32, 1, 2, 8, 2), V(32, 1, 1, 8, 2), V(32, 1, 1, 16, 2),
        V(32, 2, 1, 4, 4), V(32, 2, 1, 8, 4), V(32, 2, 2, 4, 4), V(32, 2, 2, 4, 2), V(32, 2, 1, 8, 2),
        V(16, 1, 1, 4, 4), V(16, 1, 1, 8, 4), V(16, 1, 1, 16, 4), V(16, 1, 2, 4, 4), V(16, 1, 2, 8, 4),
        V(16, 1, 1, 4, 8), V(16, 1, 1, 8, 8), V(16, 1, 4, 4, 2), V(16, 1, 4, 8, 2),
        V(16, 2, 1, 4, 4), V(16, 2, 1, 8, 4), V(16, 2, 1, 8, 2), V(16, 2, 2, 4, 4),
    };
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("B=%d  tile T=%d  MT=%d  ksplit=%d\n", B, T, MT, ksplit);
    for (const Shape& sh : shapes) {
        if (quick && &sh != &shapes[0]) break;
        const int rows_per_tile = sh.gate ? T / 2 : T;
        const int NT = (sh.N + rows_per_tile - 1) / rows_per_tile;
        const int KS = (sh.K + mmi_kstep(T) - 1) / mmi_kstep(T);
        const size_t welems = (size_t)NT * KS * 512;
        const size_t wbytes = welems * 2;
        int nbuf = (int)((size_t)1400 * 1024 * 1024 / wbytes) + 1;
        if (nbuf > 64) nbuf = 64;
        if (nbuf < 4) nbuf = 4;
        if (getenv("MB_NBUF")) nbuf = atoi(getenv("MB_NBUF"));   // 1: the same weights every launch (L2 / Infinity Cache resident)
        uint16_t* w;
        CK(hipMalloc(&w, wbytes * nbuf));
        k_fill_rand_bf16<<<2048, 256, 0, s>>>(w, welems * nbuf, 12345u);
        uint16_t *x, *out;
        const size_t xelems = (size_t)MT * KS * 512;
        const int out_ks = (sh.N + mmi_kstep(T) - 1) / mmi_kstep(T);
        const size_t oelems = (size_t)MT * out_ks * 512;
        float* partial;
        CK(hipMalloc(&partial, (size_t)4 * 64 * sh.N * sizeof(float)));
        CK(hipMalloc(&x, xelems * 2));
        CK(hipMalloc(&out, oelems * 2));
        k_fill_rand_bf16<<<256, 256, 0, s>>>(x, xelems, 777u);
        CK(hipMemsetAsync(out, 0, oelems * 2, s));
        CK(hipStreamSynchronize(s));
        printf("== %s  (%.1f MB, %d n-tiles x %d k-steps, %d buffers)\n", sh.name, wbytes / 1e6, NT, KS, nbuf);
        for (int blocks : {1024, 2048, 4096}) {      // pure read of the same buffers
            unsigned* sink = (unsigned*)out;
            const int reps = wbytes > 50e6 ? 3 : 10;
            for (int i = 0; i < nbuf; ++i) k_stream_read<<<blocks, 256, 0, s>>>((const u32x4*)(w + welems * i), wbytes / 16, sink);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int i = 0; i < nbuf; ++i) k_stream_read<<<blocks, 256, 0, s>>>((const u32x4*)(w + welems * i), wbytes / 16, sink);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / (reps * nbuf);
            printf("   pure nt read, %4d blocks                 %8.2f us  %7.0f GB/s\n", blocks, us, wbytes / us / 1e3);
        }
        if (T == 32 && MT == 1 && KS % 64 == 0) {   // operand-traffic probe on the same buffers
            for (int mode = 0; mode < 2; ++mode) {
                const int reps = wbytes > 50e6 ? 3 : 10;
                auto go = [&](int i) {
                    const u32x4* wv = (const u32x4*)(w + welems * i);
                    if (mode == 0) hipLaunchKernelGGL((k_probe_xcost<8, 2, 0>), dim3(NT), dim3(512), 0, s, wv, (const u32x4*)x, KS, (float*)out);
                    else hipLaunchKernelGGL((k_probe_xcost<8, 2, 1>), dim3(NT), dim3(512), 0, s, wv, (const u32x4*)x, KS, (float*)out);
                };
                for (int i = 0; i < nbuf; ++i) go(i);
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) for (int i = 0; i < nbuf; ++i) go(i);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = 1e3 * ms / (reps * nbuf);
                printf("   main loop only, %-24s %8.2f us  %7.0f GB/s\n", mode == 0 ? "operand fragments from L2" : "no operand loads", us, wbytes / us / 1e3);
            }
        }
        for (const Variant& v : variants) {
            if (v.TN != T || v.MT != MT) continue;
            if (quick && !(v.NTW == 1 && v.WAVES == 8 && v.U == 2)) continue;
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            a.xp = (const u32x4*)x; a.out = out; a.resid = out; a.B = B; a.N = sh.N; a.KSTEPS = KS; a.NT = NT;
            a.out_mode = MMI_OUT_PACKED; a.out_ld = sh.N; a.out_ksteps = out_ks;
            a.epi = sh.gate ? MMI_EPI_GATE : (ksplit > 1 ? MMI_EPI_PARTIAL : MMI_EPI_RESID);
            a.partial = partial;
            const dim3 groups((NT + v.NTW - 1) / v.NTW, sh.gate ? 1 : ksplit);
            const int reps = wbytes > 50e6 ? 3 : 10;
            for (int i = 0; i < nbuf; ++i) { a.wp = (const u32x4*)(w + welems * i); v.fn(groups, s, a); }   // warm-up
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r)
                for (int i = 0; i < nbuf; ++i) { a.wp = (const u32x4*)(w + welems * i); v.fn(groups, s, a); }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / (reps * nbuf);
            printf("   %-22s groups %5d x %d  %8.2f us  %7.0f GB/s\n", v.name, groups.x, groups.y, us, wbytes / us / 1e3);
        }
        CK(hipFree(w)); CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(partial));
    }
    return 0;
}
