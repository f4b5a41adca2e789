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
#include <string.h>
#include <math.h>
#include <vector>

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


// ---- prototype: activations resident in LDS ----------------------------------------------------------------------
// One workgroup per CU walks a contiguous range of n-tiles.  K is cut into chunks of KC k-steps whose activation fragments
// (MT x KC KiB = 64 KiB) are staged in LDS, double-buffered: every activation byte leaves the L2 ONCE per workgroup instead
// of once per n-tile.  The 8 waves split each chunk's k-steps; per (chunk, tile) segment a wave streams its KPW weight
// fragments from HBM one segment ahead of the MFMAs (register double buffer) and reads the activation fragments from LDS
// (ds_read_b128).  Accumulators of the up-to-3 tiles stay in registers across the chunks.  Epilogue: the per-wave partial tiles
// go out raw (the host sums the 8 waves for the check) - this measures the main loop, like k_probe_xcost.
template <int MT>
__global__ __launch_bounds__(512) void k_probe_lds(const u32x4* __restrict__ wpk, const u32x4* __restrict__ xpk, int KSTEPS, int NT,
                                                   float* __restrict__ out /* [NT][MT][8][64][16] or null */) {
    typedef float acc_t __attribute__((ext_vector_type(16)));
    constexpr int KC = 64 / MT;                 // k-steps per chunk
    constexpr int KPW = KC / 8;                 // k-steps per wave per chunk
    constexpr int XPT = MT * KC * 64 / 512;     // 16-byte activation pieces per thread per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem_raw);          // [2][MT*KC*64]
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int t0 = (int)((long)bid * NT / G), t1 = (int)((long)(bid + 1) * NT / G);
    const int ntiles = t1 - t0;                 // <= 3 (checked by the launcher)
    const int nchunks = KSTEPS / KC;
    acc_t acc[3][MT];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;
    // activation chunk c lives at xpk[(m*KSTEPS + c*KC + k)*64 + lane]; LDS copy at xs[buf][(m*KC + k)*64 + lane]
    auto xsrc = [&](int c, int j) {             // j-th piece of this thread
        const int e = j * 512 + tid;            // 0 .. MT*KC*64
        const int m = e / (KC * 64), rest = e - m * (KC * 64);
        return xpk + ((long)m * KSTEPS + (long)c * KC) * 64 + rest;
    };
    u32x4 xpre[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) xpre[j] = *xsrc(0, j);
#pragma unroll
    for (int j = 0; j < XPT; ++j) xs[j * 512 + tid] = xpre[j];
    auto wsrc = [&](int c, int t) { return wpk + ((long)(t0 + t) * KSTEPS + (long)c * KC + wave * KPW) * 64 + lane; };
    u32x4 cur[KPW], nxt[KPW];
    {
        const u32x4* wp = wsrc(0, 0);
#pragma unroll
        for (int i = 0; i < KPW; ++i) cur[i] = mmi_load_nt(wp + i * 64);
    }
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                        // chunk c is in xs[c & 1]; nobody reads xs[(c + 1) & 1] any more
        const bool more = c + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int j = 0; j < XPT; ++j) xpre[j] = *xsrc(c + 1, j);
        }
        const u32x4* xb = xs + (c & 1) * (MT * KC * 64) + wave * KPW * 64 + lane;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t < ntiles) {
                const bool last_tile = t == ntiles - 1;
                const int nc = last_tile ? c + 1 : c, nt = last_tile ? 0 : t + 1;
                if (nc < nchunks) {
                    const u32x4* wp = wsrc(nc, nt);
#pragma unroll
                    for (int i = 0; i < KPW; ++i) nxt[i] = mmi_load_nt(wp + i * 64);
                }
#pragma unroll
                for (int i = 0; i < KPW; ++i)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[t][m] = mmi_mfma_bf16_32x32x16(cur[i], xb[(m * KC + i) * 64], acc[t][m]);
#pragma unroll
                for (int i = 0; i < KPW; ++i) cur[i] = nxt[i];
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < XPT; ++j) xs[((c + 1) & 1) * (MT * KC * 64) + j * 512 + tid] = xpre[j];
        }
    }
    if (out) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < ntiles)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[((((long)(t0 + t) * MT + m) * 8 + wave) * 64 + lane) * 16 + r] = acc[t][m][r];
    }
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
        if (T == 32 && MT <= 2 && KS % 64 == 0 && NT <= 3 * 256 && NT >= 128 && getenv("MB_LDS")) {   // the LDS-resident prototype
            const int G = 256;
            float* pout;
            const size_t pout_elems = (size_t)NT * MT * 8 * 64 * 16;
            CK(hipMalloc(&pout, pout_elems * sizeof(float)));
            auto launch = [&](const u32x4* wv, float* o) {
                if (MT == 1) hipLaunchKernelGGL((k_probe_lds<1>), dim3(G), dim3(512), 131072, s, wv, (const u32x4*)x, KS, NT, o);
                else hipLaunchKernelGGL((k_probe_lds<2>), dim3(G), dim3(512), 131072, s, wv, (const u32x4*)x, KS, NT, o);
            };
            CK(hipFuncSetAttribute((const void*)k_probe_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            CK(hipFuncSetAttribute((const void*)k_probe_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            launch((const u32x4*)w, pout);
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            {   // check a few tiles against the packed operands on the host
                std::vector<float> ho(pout_elems);
                std::vector<uint16_t> hw((size_t)NT * KS * 512), hx((size_t)MT * KS * 512);
                CK(hipMemcpy(ho.data(), pout, pout_elems * sizeof(float), hipMemcpyDeviceToHost));
                CK(hipMemcpy(hw.data(), w, hw.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx.data(), x, hx.size() * 2, hipMemcpyDeviceToHost));
                auto bf = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
                double worst = 0;
                for (int nt : {0, 1, NT / 2, NT - 1})
                    for (int m = 0; m < MT; ++m)
                        for (int i : {0, 5, 31})
                            for (int j : {0, 17, 31}) {
                                double ref = 0;
                                for (int ks = 0; ks < KS; ++ks)
                                    for (int kq = 0; kq < 2; ++kq)
                                        for (int e = 0; e < 8; ++e)
                                            ref += (double)bf(hw[(((size_t)nt * KS + ks) * 64 + kq * 32 + i) * 8 + e]) *
                                                   bf(hx[(((size_t)m * KS + ks) * 64 + kq * 32 + j) * 8 + e]);
                                // D[i][j] lives in lane j + 32*((i>>2)&1), register (i&3) + 4*(i>>3)
                                const int ln = j + 32 * ((i >> 2) & 1), rg = (i & 3) + 4 * (i >> 3);
                                double got = 0;
                                for (int wv = 0; wv < 8; ++wv) got += ho[((((size_t)nt * MT + m) * 8 + wv) * 64 + ln) * 16 + rg];
                                worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 1e-3));
                            }
                printf("   LDS-resident prototype: worst relative error on sampled outputs %.2e\n", worst);
            }
            const int reps = wbytes > 50e6 ? 3 : 10;
            for (int i = 0; i < nbuf; ++i) launch((const u32x4*)(w + welems * i), nullptr);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) for (int i = 0; i < nbuf; ++i) launch((const u32x4*)(w + welems * i), nullptr);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / (reps * nbuf);
            printf("   LDS-resident prototype, 256 workgroups   %8.2f us  %7.0f GB/s   (main loop only)\n", us, wbytes / us / 1e3);
            CK(hipFree(pout));
            // the product kernel (k_gemm_xlds: the same loop + the common epilogue), plain and staggered tails
            for (int stagger = 0; stagger < 2; ++stagger) {
                GemmArgs a;
                memset(&a, 0, sizeof(a));
                a.xp = (const u32x4*)x; a.out = out; a.B = B; a.N = sh.N; a.KSTEPS = KS; a.NT = NT;
                a.out_mode = MMI_OUT_PACKED; a.out_ld = sh.N; a.out_ksteps = out_ks;
                a.epi = sh.gate ? MMI_EPI_GATE : MMI_EPI_STORE;
                a.gate_rows = sh.gate ? sh.N : 0;
                auto go = [&](const u32x4* wv) {
                    a.wp = wv;
                    if (MT == 1 && !stagger) hipLaunchKernelGGL((k_gemm_xlds<1, 64, 3, false>), dim3(G), dim3(512), 131072, s, a);
                    else if (MT == 1) hipLaunchKernelGGL((k_gemm_xlds<1, 64, 3, true>), dim3(G), dim3(512), 131072, s, a);
                    else if (!stagger) hipLaunchKernelGGL((k_gemm_xlds<2, 32, 3, false>), dim3(G), dim3(512), 131072, s, a);
                    else hipLaunchKernelGGL((k_gemm_xlds<2, 32, 3, true>), dim3(G), dim3(512), 131072, s, a);
                };
                CK(hipFuncSetAttribute((const void*)k_gemm_xlds<1, 64, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
                CK(hipFuncSetAttribute((const void*)k_gemm_xlds<1, 64, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
                CK(hipFuncSetAttribute((const void*)k_gemm_xlds<2, 32, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
                CK(hipFuncSetAttribute((const void*)k_gemm_xlds<2, 32, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
                for (int i = 0; i < nbuf; ++i) go((const u32x4*)(w + welems * i));
                CK(hipStreamSynchronize(s));
                CK(hipGetLastError());
                CK(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) for (int i = 0; i < nbuf; ++i) go((const u32x4*)(w + welems * i));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us2 = 1e3 * ms / (reps * nbuf);
                printf("   k_gemm_xlds %-28s %8.2f us  %7.0f GB/s   (whole kernel)\n", stagger ? "(staggered epilogues)" : "(epilogues at the end)", us2, wbytes / us2 / 1e3);
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
            const int osplit = (getenv("MB_OSPLIT") && !sh.gate && v.NTW == 1) ? atoi(getenv("MB_OSPLIT")) : 1;   // row octets of a tile over several workgroups
            a.osplit = osplit;
            const dim3 groups((NT + v.NTW - 1) / v.NTW * osplit, sh.gate ? 1 : ksplit);
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
