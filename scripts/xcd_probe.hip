// What can ONE XCD (32 CUs, one L2) do on its own?  (a) HBM read bandwidth when only the workgroups that landed on XCC 0
// stream, (b) cost of a barrier among those workgroups through an atomic counter in their shared L2.
// Feasibility probe for an XCD-local persistent depth-transformer kernel; not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf; }   // HW_REG_XCC_ID = 20, bits [3:0]

__global__ void k_census(int* count) { if (threadIdx.x == 0) atomicAdd(&count[xcc_id()], 1); }

// only workgroups on XCC `want` (want < 0: all) read; each takes a contiguous slice
__global__ __launch_bounds__(512) void k_read(const u32x4* p, size_t n16, int want, int nsel, int* ticket, unsigned* sink) {
    __shared__ int my;
    const int x = xcc_id();
    if (want >= 0 && x != want) return;
    if (threadIdx.x == 0) my = atomicAdd(ticket, 1);
    __syncthreads();
    const int me = my;
    const size_t per = n16 / nsel;
    const u32x4* q = p + (size_t)me * per;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + 3 * 512 < per; i += 4 * 512) {
        u32x4 a = __builtin_nontemporal_load(q + i), b = __builtin_nontemporal_load(q + i + 512), c = __builtin_nontemporal_load(q + i + 1024), d = __builtin_nontemporal_load(q + i + 1536);
        acc ^= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    if (acc == 0x12345678u) *sink = acc;
}

// rounds of a counter barrier among the workgroups on XCC `want`; every round each workgroup also publishes a small
// record and reads its neighbour's with non-temporal (L1-bypassing) loads.  Spins are bounded.
__global__ __launch_bounds__(256) void k_barrier(int want, int nsel, int rounds, unsigned* counter, int* ticket, float* rec, int* fail) {
    __shared__ int my;
    const int x = xcc_id();
    if (x != want) return;
    if (threadIdx.x == 0) my = atomicAdd(ticket, 1);
    __syncthreads();
    const int me = my;
    if (me >= nsel) return;
    for (int r = 1; r <= rounds; ++r) {
        rec[me * 256 + threadIdx.x] = (float)(r * 1000 + me);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the (shared) L2
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r * nsel)) {
                if (++spins > (1 << 22)) { *fail = 1; break; }
            }
        }
        __syncthreads();
        const int nb = (me + 1) % nsel;
        const float v = __builtin_nontemporal_load(&rec[nb * 256 + threadIdx.x]);
        if (v != (float)(r * 1000 + nb)) {
            if (atomicAdd(fail + 1, 1) == 0) { fail[2] = r; fail[3] = me; fail[4] = (int)v; fail[5] = (int)threadIdx.x; }   // first stale read: round, reader, value seen
        }
        __syncthreads();
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int *count, *ticket, *fail; unsigned* counter; float* rec; unsigned* sink;
    CK(hipMalloc(&count, 64)); CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&fail, 32)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&rec, 256 * 256 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(count, 0, 64));
    k_census<<<256, 64, 0, s>>>(count);
    int h[16]; CK(hipMemcpy(h, count, 64, hipMemcpyDeviceToHost));
    printf("census of a 256-workgroup launch by XCC id:"); for (int i = 0; i < 8; ++i) printf(" %d", h[i]); printf("\n");
    const int nsel = h[0];
    const size_t bytes = (size_t)2 << 30;
    u32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int want : {-1, 0}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(ticket, 0, 4));
            CK(hipEventRecord(e0, s));
            k_read<<<256, 512, 0, s>>>(buf, bytes / 16, want, want < 0 ? 256 : nsel, ticket, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s read 2 GiB: %.3f ms = %.0f GB/s (%d workgroups x 512 threads)\n", want < 0 ? "whole chip" : "XCC 0 only", ms, bytes / ms / 1e6, want < 0 ? 256 : nsel);
        }
    }
    for (int rounds : {100, 1000, 100, 5000, 100}) {
        CK(hipMemset(ticket, 0, 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 32));
        if (getenv("PROBE_SYNC")) CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s));
        k_barrier<<<256, 256, 0, s>>>(0, nsel, rounds, counter, ticket, rec, fail);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int f[8]; CK(hipMemcpy(f, fail, 32, hipMemcpyDeviceToHost));
        printf("XCC-0 barrier + 1 KiB neighbour hand-off, %d workgroups, %d rounds: %.2f us per round (timeouts %d, stale reads %d", nsel, rounds, 1e3 * ms / rounds, f[0], f[1]);
        if (f[1]) printf("; first: round %d reader %d saw %d thread %d", f[2], f[3], f[4], f[5]);
        printf(")\n");
    }
    return 0;
}
