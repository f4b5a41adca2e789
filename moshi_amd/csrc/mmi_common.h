// Small helpers shared by the Mimi and LM engines (host + device).
#pragma once
#include <mmi_device.h>  // resolved through -I: csrc/ for the gfx950 build
#include "../../include/moshi_mi.h"

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <functional>

#define MMI_HD __host__ __device__ __forceinline__
MMI_HD uint16_t mmi_f32_to_bf16_bits(float f);

// bf16 <-> f32, round-to-nearest-even, NaN kept quiet: identical to torch's float->bfloat16 cast.
// Device code on gfx950 uses the hardware conversion (v_cvt_pk_bf16_f32, one instruction per pair) through
// mmi_device.h's mmi_cvt_bf16 / mmi_cvt_pk_bf16; the bit-level version below is the host side and the definition of it.
MMI_HD float mmi_bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
MMI_HD uint16_t mmi_f32_to_bf16_bits(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
MMI_HD uint16_t mmi_f32_to_bf16(float f) { return mmi_cvt_bf16(f); }
// two values -> one packed register (lo in bits 0..15)
MMI_HD uint32_t mmi_pack_bf16x2(float lo, float hi) { return mmi_cvt_pk_bf16(lo, hi); }
// round an fp32 value to the nearest bf16 and return it widened again ("bf16 rounding point")
MMI_HD float mmi_round_bf16(float f) { return mmi_bf16_to_f32(mmi_f32_to_bf16(f)); }

static inline int mmi_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t mmi_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- error plumbing ------------------------------------------------------------------------
void mmi_set_error(const std::string& msg);
int mmi_fail(int code, const std::string& msg);

#define MMI_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return mmi_fail(MMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

#define MMI_CHECK_LAUNCH()                                                                           \
    do {                                                                                             \
        hipError_t e_ = hipGetLastError();                                                           \
        if (e_ != hipSuccess)                                                                        \
            return mmi_fail(MMI_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e_));    \
    } while (0)

// A handle binds to the HIP device that was current when it was created; every ABI entry switches the calling thread to that
// device for the duration of the call and restores the caller's (so `device='cuda:1'` works without the caller having made
// device 1 current, and nothing leaks into the caller's - e.g. torch's - notion of the current device).
struct MmiDeviceGuard {
    int prev = -1;
    explicit MmiDeviceGuard(int dev) {
        int cur = -1;
        if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~MmiDeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
    MmiDeviceGuard(const MmiDeviceGuard&) = delete;
    MmiDeviceGuard& operator=(const MmiDeviceGuard&) = delete;
};

// ---- weight table lookup -------------------------------------------------------------------
struct MmiWeights {
    const mmi_tensor_desc* descs;
    int n;
    const mmi_tensor_desc* find(const std::string& name) const {
        for (int i = 0; i < n; ++i)
            if (descs[i].name && name == descs[i].name) return &descs[i];
        return nullptr;
    }
};

// device allocation bookkeeping: every engine frees what it allocated
struct MmiArena {
    std::vector<void*> ptrs;
    std::vector<size_t> sizes;
    size_t bytes = 0;
    template <class T>
    hipError_t alloc(T** p, size_t count) {
        void* q = nullptr;
        size_t nb = count * sizeof(T);
        if (nb == 0) nb = sizeof(T);
        hipError_t e = hipMalloc(&q, nb);
        if (e != hipSuccess) return e;
        // MMI_DEBUG_POISON=1 (tests / scripts/gpu_check.sh): every allocation of a handle starts as 0xFF bytes (bf16 / fp32 NaNs,
        // int -1) instead of whatever the allocator hands out, so that a read of state the engine never initialised shows up as
        // a wrong result on every box, not only on the one whose memory happened to hold something else
        // (MMI_DEBUG_POISON=0: zero-fill instead - two sessions then start from identical memory, which is what MMI_DEBUG_TRACE's
        // per-op checksums need to be comparable line by line)
        const char* pz = getenv("MMI_DEBUG_POISON");
        if (pz && (pz[0] == '1' || pz[0] == '0') && (e = hipMemset(q, pz[0] == '1' ? 0xFF : 0x00, nb)) != hipSuccess) { hipFree(q); return e; }
        ptrs.push_back(q);
        sizes.push_back(nb);
        bytes += nb;
        *p = (T*)q;
        return hipSuccess;
    }
    void release() {
        for (void* p : ptrs) hipFree(p);
        ptrs.clear();
        sizes.clear();
        bytes = 0;
    }
    // snapshot / restore of every allocation, back to back (StreamingModule.get/set_streaming_state, streaming.py:158-181)
    hipError_t save(void* dst, hipStream_t s) const {
        size_t at = 0;
        for (size_t i = 0; i < ptrs.size(); ++i) {
            hipError_t e = hipMemcpyAsync((char*)dst + at, ptrs[i], sizes[i], hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
            at += sizes[i];
        }
        return hipSuccess;
    }
    hipError_t load(const void* src, hipStream_t s) {
        size_t at = 0;
        for (size_t i = 0; i < ptrs.size(); ++i) {
            hipError_t e = hipMemcpyAsync(ptrs[i], (const char*)src + at, sizes[i], hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return e;
            at += sizes[i];
        }
        return hipSuccess;
    }
};
