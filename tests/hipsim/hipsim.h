// hipsim — a tiny fiber-based SIMT simulator used ONLY by tests/ to debug kernel logic on a
// machine without a GPU.  TEST INFRASTRUCTURE, NOT PRODUCT: nothing under moshi_amd/ includes,
// links or loads anything from this directory; the product library is built by hipcc for gfx950
// and fails loudly when it is missing.  The simulator compiles the very same kernel sources for
// the host (clang++, x86) by shadowing `mmi_device.h`, runs every thread of a block as a fiber
// (ucontext) on one OS thread, and implements wave-level operations (64-lane shuffles, MFMA)
// through a per-wave exchange area + rendezvous.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipsim {

extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

// Run `body` once per thread of a grid×block launch (synchronous).
void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);

void sync_block();            // __syncthreads()
void sync_wave();             // rendezvous of the live lanes of the calling wave
int lane_id();                // 0..63
int wave_live_lanes();        // live lanes in the calling wave
// per-wave exchange area: 64 slots x 64 bytes
struct alignas(64) Slot { unsigned char b[64]; };
Slot* wave_slots();
void* dyn_smem();

int num_workers();
void set_num_workers(int n);
// order in which the waves of a block and the lanes of a wave are visited between two synchronisation points:
// 0 forward (default), 1 reverse, 2 random (seed); also HIPSIM_SCHED=forward|reverse|random[:seed]
void set_schedule(int mode, uint64_t seed);
int schedule();

}  // namespace hipsim

// ---- minimal HIP host API surface used by the engines -------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotSupported = 801,
       hipErrorStreamCaptureUnsupported = 900 };
typedef struct hipsimStream* hipStream_t;
typedef struct hipsimEvent* hipEvent_t;
typedef struct hipsimGraph* hipGraph_t;
typedef struct hipsimGraphExec* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2,
                     hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1,
                            hipStreamCaptureModeRelaxed = 2 };

hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* s);
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { return hipStreamCreate(s); }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t* e);
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
// the simulator executes every call synchronously in host order, so a wait on a recorded event is already satisfied
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 10017 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
