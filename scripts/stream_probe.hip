// What do cross-stream dependencies and concurrent streams cost on this MI355X / ROCm stack?  Measurement tool for the duplex
// pipeline's design; not part of the product.
//   1. a dependent chain of tiny kernels on one stream (eager and as a hipGraph)
//   2. the same chain ping-ponged between two streams through event record / stream-wait pairs (eager, graph pieces)
//   3. do two streams overlap?  a low-occupancy spinning kernel on A next to a chain on B
//   4. a hipGraph with two parallel branches (fork / join captured through events): do the branches overlap?
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
// every workgroup spins for `us` microseconds (100 MHz constant clock)
__global__ void k_spin(float* p, long us) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < us * 100) {}
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;
}

__global__ void k_touch(float* p, int n_per_wg) {
    float* q = p + (size_t)blockIdx.x * n_per_wg;
    for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) q[i] = q[i] * 1.0001f + 1.0f;
}

__global__ void k_set_flag(int* flag, int v) { if (threadIdx.x == 0) { __threadfence(); __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); } }
__global__ void k_wait_flag(int* flag, int v) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(32);
    }
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t A, B, C, H;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&H, hipStreamNonBlocking, hi));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    std::vector<hipEvent_t> ev(8);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto elapsed = [&](hipStream_t s, auto body) {
        CK(hipDeviceSynchronize());
        const double h0 = now_ms();
        CK(hipEventRecord(t0, s));
        body();
        CK(hipEventRecord(t1, s));
        const double h1 = now_ms();
        CK(hipEventSynchronize(t1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("   [host enqueue %.3f ms]", h1 - h0);
        return (double)ms;
    };
    const int N = 1000;
    // 1. one stream
    for (int rep = 0; rep < 2; ++rep) {
        double ms = elapsed(A, [&] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, 1, 64, 0, A, buf); });
        printf(" 1a. eager chain on one stream: %.2f us per kernel\n", 1e3 * ms / N);
    }
    hipGraph_t g; hipGraphExec_t ge100;
    CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_tiny, 1, 64, 0, C, buf);
    CK(hipStreamEndCapture(C, &g)); CK(hipGraphInstantiate(&ge100, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge100, A)); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        double ms = elapsed(A, [&] { for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge100, A)); });
        printf(" 1b. graph of 100 tiny kernels x 10 on one stream: %.2f us per kernel, %.1f us per graph\n", 1e3 * ms / 1000, 1e3 * ms / 10);
    }
    // 1c. the same graph on every kind of stream: back to back, and one launch from an idle device (host clock)
    {
        hipStream_t L2; CK(hipStreamCreateWithPriority(&L2, hipStreamNonBlocking, lo));
        struct { const char* name; hipStream_t s; } S[] = {{"A (first non-blocking)", A}, {"B (second non-blocking)", B}, {"C (the capture stream)", C},
                                                            {"H (high priority)", H}, {"L2 (low priority)", L2}, {"null stream", nullptr}};
        // a heavier graph: 200 kernels of 256 workgroups that each read + write 64 KiB (a few us each)
        hipGraph_t gh; hipGraphExec_t geh;
        float* big; CK(hipMalloc(&big, 256u << 20));
        CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_touch, 256, 256, 0, C, big + (size_t)(i % 8) * (4u << 20), 16384);
        CK(hipStreamEndCapture(C, &gh)); CK(hipGraphInstantiate(&geh, gh, nullptr, nullptr, 0));
        for (auto& e : S) {
            CK(hipGraphLaunch(ge100, e.s)); CK(hipGraphLaunch(geh, e.s)); CK(hipDeviceSynchronize());
            double ms = elapsed(e.s, [&] { for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge100, e.s)); });
            double msh = elapsed(e.s, [&] { for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(geh, e.s)); });
            CK(hipDeviceSynchronize());
            double idle = 0, idleh = 0;
            for (int r = 0; r < 5; ++r) {
                double h0 = now_ms(); CK(hipGraphLaunch(ge100, e.s)); CK(hipStreamSynchronize(e.s)); idle += now_ms() - h0;
                h0 = now_ms(); CK(hipGraphLaunch(geh, e.s)); CK(hipStreamSynchronize(e.s)); idleh += now_ms() - h0;
            }
            printf(" 1c. %-24s tiny graph %.2f us/kernel back to back, %.1f us per launch from idle | 200 x 256-workgroup kernels: %.2f us/kernel back to back, %.1f us per launch from idle\n",
                   e.name, 1e3 * ms / 1000, 1e3 * idle / 5, 1e3 * msh / 1000, 1e3 * idleh / 5);
        }
        // the heavy graph on A while B / H run the tiny chain
        for (int which = 0; which < 2; ++which) {
            hipStream_t S2 = which ? H : B;
            double solo = elapsed(A, [&] { for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(geh, A)); });
            double both = elapsed(A, [&] {
                CK(hipEventRecord(ev[3], A)); CK(hipStreamWaitEvent(S2, ev[3], 0));
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(geh, A));
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge100, S2));
                CK(hipEventRecord(ev[4], S2)); CK(hipStreamWaitEvent(A, ev[4], 0));
            });
            printf(" 1d. 5 heavy graphs on A alone %.3f ms; with 5 tiny graphs on %s %.3f ms\n", solo, which ? "H" : "B", both);
        }
    }
    // 1e. does a PENDING cross-stream wait slow the stream it waits for?  A runs 10 tiny-kernel graphs; B (or the null stream)
    //     waits for A's end through an event, or through a flag polled by a one-wave kernel
    {
        int* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
        int gen = 0;
        for (int rep = 0; rep < 2; ++rep) {
            double alone = elapsed(A, [&] { for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge100, A)); });
            printf(" 1e. A alone: %.3f ms\n", alone);
            for (int which = 0; which < 3; ++which) {
                hipStream_t W = which == 0 ? B : (which == 1 ? H : (hipStream_t)nullptr);
                const char* wn = which == 0 ? "B" : (which == 1 ? "H" : "null stream");
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, A));
                for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge100, A));
                CK(hipEventRecord(t1, A));
                CK(hipEventRecord(ev[3], A));
                CK(hipStreamWaitEvent(W, ev[3], 0));
                hipLaunchKernelGGL(k_tiny, 1, 64, 0, W, buf);
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                printf(" 1e. A with %s waiting on its end event: %.3f ms\n", wn, ms);
            }
            for (int which = 0; which < 2; ++which) {
                hipStream_t W = which == 0 ? B : H;
                ++gen;
                CK(hipDeviceSynchronize());
                hipLaunchKernelGGL(k_wait_flag, 1, 64, 0, W, flag, gen);       // enqueued FIRST: it is resident while A runs
                hipLaunchKernelGGL(k_tiny, 1, 64, 0, W, buf);
                CK(hipEventRecord(t0, A));
                for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge100, A));
                CK(hipEventRecord(t1, A));
                hipLaunchKernelGGL(k_set_flag, 1, 64, 0, A, flag, gen);
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, t0, t1));
                printf(" 1e. A with a flag-polling kernel resident on %s: %.3f ms\n", which ? "H" : "B", ms);
            }
        }
    }
    // 2. ping-pong
    auto pingpong = [&](hipStream_t X, hipStream_t Y, int hops, auto work, const char* what) {
        double ms = elapsed(X, [&] {
            for (int i = 0; i < hops; ++i) {
                hipStream_t s = (i & 1) ? Y : X, o = (i & 1) ? X : Y;
                work(s);
                CK(hipEventRecord(ev[i & 1], s));
                CK(hipStreamWaitEvent(o, ev[i & 1], 0));
            }
            // t1 is recorded on X: make X wait for the last piece
            CK(hipEventRecord(ev[2], (hops & 1) ? X : Y)); CK(hipStreamWaitEvent(X, ev[2], 0));
        });
        printf(" %s: %.2f us per hop (%d hops)\n", what, 1e3 * ms / hops, hops);
    };
    for (int rep = 0; rep < 2; ++rep)
        pingpong(A, B, 400, [&](hipStream_t s) { hipLaunchKernelGGL(k_tiny, 1, 64, 0, s, buf); }, "2a. eager tiny kernel, A<->B events");
    pingpong(A, H, 400, [&](hipStream_t s) { hipLaunchKernelGGL(k_tiny, 1, 64, 0, s, buf); }, "2b. eager tiny kernel, A<->H (high priority) events");
    for (int rep = 0; rep < 2; ++rep)
        pingpong(A, B, 40, [&](hipStream_t s) { CK(hipGraphLaunch(ge100, s)); }, "2c. graph of 100 tiny kernels per hop, A<->B (subtract 100 x 1b)");
    pingpong(A, H, 40, [&](hipStream_t s) { CK(hipGraphLaunch(ge100, s)); }, "2d. graph of 100 tiny kernels per hop, A<->H");
    pingpong(A, B, 40, [&](hipStream_t s) { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_tiny, 1, 64, 0, s, buf); }, "2e. 100 eager tiny kernels per hop, A<->B");
    // 3. overlap of two streams
    {
        double solo_spin = elapsed(A, [&] { hipLaunchKernelGGL(k_spin, 32, 256, 0, A, buf + 1024, 2000L); });
        printf(" 3a. spin kernel alone (32 workgroups x 2000 us): %.3f ms\n", solo_spin);
        double solo_chain = elapsed(B, [&] { for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge100, B)); });
        printf(" 3b. 5 graphs of 100 tiny kernels alone: %.3f ms\n", solo_chain);
        for (int which = 0; which < 2; ++which) {
            hipStream_t S2 = which ? H : B;
            double both = elapsed(A, [&] {
                CK(hipEventRecord(ev[3], A)); CK(hipStreamWaitEvent(S2, ev[3], 0));
                hipLaunchKernelGGL(k_spin, 32, 256, 0, A, buf + 1024, 2000L);
                for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge100, S2));
                CK(hipEventRecord(ev[4], S2)); CK(hipStreamWaitEvent(A, ev[4], 0));
            });
            printf(" 3c. both at once (spin on A, chain on %s): %.3f ms   (sum %.3f, max %.3f)\n", which ? "H" : "B", both, solo_spin + solo_chain,
                   solo_spin > solo_chain ? solo_spin : solo_chain);
        }
        // a chip-filling kernel next to the chain: 1024 workgroups x 256 threads spinning 1 ms
        double solo_full = elapsed(A, [&] { hipLaunchKernelGGL(k_spin, 1024, 512, 0, A, buf + 1024, 1000L); });
        double both = elapsed(A, [&] {
            CK(hipEventRecord(ev[3], A)); CK(hipStreamWaitEvent(B, ev[3], 0));
            hipLaunchKernelGGL(k_spin, 1024, 512, 0, A, buf + 1024, 1000L);
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge100, B));
            CK(hipEventRecord(ev[4], B)); CK(hipStreamWaitEvent(A, ev[4], 0));
        });
        printf(" 3d. 1024 x 512-thread spinning workgroups alone %.3f ms; with the chain on B %.3f ms (sum %.3f)\n", solo_full, both, solo_full + solo_chain);
    }
    // 4. one graph, two parallel branches
    {
        hipGraph_t g2; hipGraphExec_t ge2, ge1;
        CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_spin, 32, 256, 0, C, buf + 2048, 20L);
        CK(hipStreamEndCapture(C, &g2)); CK(hipGraphInstantiate(&ge1, g2, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(C, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(ev[5], C)); CK(hipStreamWaitEvent(B, ev[5], 0));          // fork: B joins the capture
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_spin, 32, 256, 0, C, buf + 2048, 20L);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_spin, 32, 256, 0, B, buf + 4096, 20L);
        CK(hipEventRecord(ev[6], B)); CK(hipStreamWaitEvent(C, ev[6], 0));          // join
        CK(hipStreamEndCapture(C, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge1, A)); CK(hipGraphLaunch(ge2, A)); CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 2; ++rep) {
            double one = elapsed(A, [&] { CK(hipGraphLaunch(ge1, A)); });
            double two = elapsed(A, [&] { CK(hipGraphLaunch(ge2, A)); });
            printf(" 4. graph of 50 x (32 workgroups spinning 20 us): one branch %.3f ms, two parallel branches %.3f ms\n", one, two);
        }
    }
    return 0;
}
