// What does the L2 deliver to the access pattern of the temporal GEMMs' ACTIVATION operand?  In k_gemm_xlds / k_gemm_xp every
// workgroup re-reads the whole packed activation matrix (256 KB at 32 sessions, 512 KB at 64) from its XCD's L2, all 256
// workgroups walking the SAME addresses in the SAME order; under the TCC counters these kernels run at 7.4-9.8 TB/s of L2 requests
// (HISTORY.md round 4 10e), while MI355X_MICROARCH.md puts the L2 at ~34.5 TB/s.  This probe separates the candidate
// explanations with 16-byte-per-lane loads (one 1 KiB line group per wave instruction, as the GEMMs issue them):
//   same      every workgroup reads the same buffer, chunk order identical in all workgroups (the GEMMs' pattern)
//   rotated   same buffer, every workgroup starts at its own offset (requests of one instant spread over the L2 channels)
//   distinct  every workgroup reads its own buffer of the same size (no two workgroups ever want the same line)
// for footprints of 64 / 256 / 512 KiB per workgroup.  Bounded loops only, no inter-workgroup communication; runs in < 1 s.
// Measurement tool for DESIGN.md; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// chunks of 1 KiB; a wave reads chunk c as 64 lanes x 16 B.  mode 0 same, 1 rotated, 2 distinct
template <int MODE, int U>
__global__ __launch_bounds__(512) void k_read(const uint4* __restrict__ buf, int nchunks, int passes, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* base = MODE == 2 ? buf + (size_t)blockIdx.x * nchunks * 64 : buf;
    const int rot = MODE == 1 ? (int)((blockIdx.x * 2654435761u) % (unsigned)nchunks) : 0;
    unsigned acc = 0;
    for (int p = 0; p < passes; ++p) {
#pragma unroll U
        for (int c = wave; c < nchunks; c += 8) {
            int cc = c + rot;
            if (cc >= nchunks) cc -= nchunks;
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base) + (size_t)cc * 64 + lane);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x * 512 + threadIdx.x] = acc;     // never true for the fill pattern; keeps the loads
}

template <int MODE, int U>
__global__ __launch_bounds__(512) void k_read_cached(const uint4* __restrict__ buf, int nchunks, int passes, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* base = MODE == 2 ? buf + (size_t)blockIdx.x * nchunks * 64 : buf;
    const int rot = MODE == 1 ? (int)((blockIdx.x * 2654435761u) % (unsigned)nchunks) : 0;
    unsigned acc = 0;
    for (int p = 0; p < passes; ++p) {
#pragma unroll U
        for (int c = wave; c < nchunks; c += 8) {
            int cc = c + rot;
            if (cc >= nchunks) cc -= nchunks;
            const uint4 v = base[(size_t)cc * 64 + lane];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <class L>
double time_ms(hipStream_t s, L launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 10; ++r) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / 10.0;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int WG = 256;
    const size_t maxbytes = (size_t)WG * 512 * 1024;           // 128 MiB: distinct buffers at the largest footprint
    uint4* buf; unsigned* sink;
    CK(hipMalloc(&buf, maxbytes)); CK(hipMalloc(&sink, (size_t)WG * 512 * 4));
    CK(hipMemset(buf, 0x5A, maxbytes)); CK(hipMemset(sink, 0, (size_t)WG * 512 * 4));
    printf("L2 access-pattern probe: %d workgroups x 512 threads, 16 B per lane, 1 KiB per wave instruction\n", WG);
    for (int kib : {64, 256, 512}) {
        const int nchunks = kib;                                  // 1 KiB chunks
        const int passes = 4096 / kib * 4;                        // 16 MiB read per workgroup in every configuration
        const double bytes = (double)WG * kib * 1024.0 * passes;
        for (int u : {4, 16}) {
            double t[2][3];
            if (u == 4) {
                t[0][0] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<0, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[0][1] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<1, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[0][2] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<2, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][0] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<0, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][1] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<1, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][2] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<2, 4>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
            } else {
                t[0][0] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<0, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[0][1] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<1, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[0][2] = time_ms(s, [&] { hipLaunchKernelGGL((k_read<2, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][0] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<0, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][1] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<1, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
                t[1][2] = time_ms(s, [&] { hipLaunchKernelGGL((k_read_cached<2, 16>), dim3(WG), dim3(512), 0, s, buf, nchunks, passes, sink); });
            }
            for (int k = 0; k < 2; ++k)
                printf("%3d KiB per workgroup, %2d loads in flight per wave, %s: same %.2f TB/s | rotated %.2f TB/s | distinct %.2f TB/s   (%.3f / %.3f / %.3f ms for %.1f GB)\n",
                       kib, u, k ? "plain      " : "nontemporal", bytes / t[k][0] / 1e9, bytes / t[k][1] / 1e9, bytes / t[k][2] / 1e9,
                       t[k][0], t[k][1], t[k][2], bytes / 1e9);
        }
    }
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
