// hipsim shadow of moshi_amd/csrc/mmi_device.h — TEST INFRASTRUCTURE ONLY (see hipsim.h).
// Same vocabulary, implemented on the host: wave shuffles and MFMA go through the per-wave
// exchange area.  MFMA lane/register maps follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once
#include "hipsim.h"
#include <stdint.h>
#include <cmath>
#include <algorithm>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define threadIdx hipsim::t_threadIdx
#define blockIdx hipsim::t_blockIdx
#define blockDim hipsim::t_blockDim
#define gridDim hipsim::t_gridDim
#define __syncthreads() hipsim::sync_block()

inline void mmi_rope_rotate(float re, float im, float c, float sn, float& out_re, float& out_im) {
    const float a = re * c, b = im * sn, d = re * sn, e = im * c;
    out_re = a - b;
    out_im = d + e;
}
#define MMI_WAVE 64
#define MMI_SHARED static thread_local
#define MMI_DYN_SHARED(T, name) T* name = reinterpret_cast<T*>(hipsim::dyn_smem())
void mmi_note_launch(const char* kernel);
#define MMI_LAUNCH(kern, grid, block, shmem, stream, ...)                                               \
    do {                                                                                                \
        mmi_note_launch(#kern);                                                                         \
        hipsim::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); });        \
    } while (0)

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipsim::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kern(__VA_ARGS__); })

using std::min;
using std::max;

inline uint16_t mmi_f32_to_bf16_bits(float f);
inline uint16_t mmi_cvt_bf16(float f) { return mmi_f32_to_bf16_bits(f); }
inline uint32_t mmi_cvt_pk_bf16(float lo, float hi) {
    return (uint32_t)mmi_f32_to_bf16_bits(lo) | ((uint32_t)mmi_f32_to_bf16_bits(hi) << 16);
}

inline int mmi_lane() { return hipsim::lane_id(); }

template <class T>
inline T mmi_shfl(T v, int src) {
    static_assert(sizeof(T) <= 64, "slot too small");
    hipsim::Slot* s = hipsim::wave_slots();
    memcpy(s[hipsim::lane_id()].b, &v, sizeof(T));
    hipsim::sync_wave();
    T r;
    memcpy(&r, s[src & 63].b, sizeof(T));
    hipsim::sync_wave();
    return r;
}
template <class T>
inline T mmi_shfl_xor(T v, int mask) { return mmi_shfl(v, hipsim::lane_id() ^ mask); }

namespace hipsim_detail {
inline float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
struct OpF32 { float a, b; };
struct OpBF { uint16_t a[8], b[8]; };
}  // namespace hipsim_detail

inline f32x16 mmi_mfma_f32_32x32x2(float a, float b, f32x16 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpF32 me{a, b};
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x16 d;
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            OpF32 oa, ob;
            memcpy(&oa, s[i + 32 * k].b, sizeof(oa));
            memcpy(&ob, s[j + 32 * k].b, sizeof(ob));
            acc = fmaf(oa.a, ob.b, acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

inline f32x4 mmi_mfma_f32_16x16x4(float a, float b, f32x4 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpF32 me{a, b};
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x4 d;
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            OpF32 oa, ob;
            memcpy(&oa, s[i + 16 * k].b, sizeof(oa));
            memcpy(&ob, s[j + 16 * k].b, sizeof(ob));
            acc = fmaf(oa.a, ob.b, acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

inline f32x16 mmi_mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpBF me;
    memcpy(me.a, &a, 16);
    memcpy(me.b, &b, 16);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x16 d;
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            OpBF oa, ob;
            memcpy(&oa, s[i + 32 * (k >> 3)].b, sizeof(oa));
            memcpy(&ob, s[j + 32 * (k >> 3)].b, sizeof(ob));
            acc = fmaf(bf(oa.a[k & 7]), bf(ob.b[k & 7]), acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

inline f32x4 mmi_mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpBF me;
    memcpy(me.a, &a, 16);
    memcpy(me.b, &b, 16);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x4 d;
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            OpBF oa, ob;
            memcpy(&oa, s[i + 16 * (k >> 3)].b, sizeof(oa));
            memcpy(&ob, s[j + 16 * (k >> 3)].b, sizeof(ob));
            acc = fmaf(bf(oa.a[k & 7]), bf(ob.b[k & 7]), acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

// ---- fp8 e4m3fn (OCP): software encode (round-to-nearest-even, clamp to +-448) / decode, and the two fp8 MFMA forms ----
namespace hipsim_detail {
inline uint8_t f32_to_e4m3(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    float a = fabsf(x);
    if (!(a == a)) return (uint8_t)(sign | 0x7f);
    if (a > 448.f) a = 448.f;
    if (a < 0.015625f) {                               // below 2^-6: subnormal grid of 2^-9
        const int q = (int)nearbyintf(a * 512.f);      // 0..8, ties to even (default rounding mode); 8 encodes as the first normal
        return (uint8_t)(sign | q);
    }
    int e; const float m = frexpf(a, &e);              // a = m * 2^e, m in [0.5, 1)
    int q = (int)nearbyintf((m * 2.f - 1.f) * 8.f);    // mantissa of 1.mmm
    int E = e - 1 + 7;
    if (q == 8) { q = 0; E += 1; }
    return (uint8_t)(sign | (E << 3) | q);
}
inline float e4m3_to_f32(uint8_t b) {
    const int E = (b >> 3) & 15, m = b & 7;
    float v = E == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + (float)m / 8.f, E - 7);
    if (E == 15 && m == 7) v = NAN;
    return (b & 0x80) ? -v : v;
}
struct OpF8 { uint8_t a[8], b[8]; };
}  // namespace hipsim_detail

inline uint32_t mmi_cvt_fp8x4(float a, float b, float c, float d) {
    using namespace hipsim_detail;
    return (uint32_t)f32_to_e4m3(a) | ((uint32_t)f32_to_e4m3(b) << 8) | ((uint32_t)f32_to_e4m3(c) << 16) | ((uint32_t)f32_to_e4m3(d) << 24);
}

inline void mmi_fp8x4_to_f32(uint32_t w, float* o) {
    for (int i = 0; i < 4; ++i) o[i] = hipsim_detail::e4m3_to_f32((uint8_t)(w >> (8 * i)));
}

inline f32x16 mmi_mfma_fp8_32x32x16(u32x2 a, u32x2 b, f32x16 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpF8 me;
    memcpy(me.a, &a, 8);
    memcpy(me.b, &b, 8);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x16 d;
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            OpF8 oa, ob;
            memcpy(&oa, s[i + 32 * (k >> 3)].b, sizeof(oa));
            memcpy(&ob, s[j + 32 * (k >> 3)].b, sizeof(ob));
            acc = fmaf(e4m3_to_f32(oa.a[k & 7]), e4m3_to_f32(ob.b[k & 7]), acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

inline f32x4 mmi_mfma_fp8_16x16x32(u32x2 a, u32x2 b, f32x4 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpF8 me;
    memcpy(me.a, &a, 8);
    memcpy(me.b, &b, 8);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    f32x4 d;
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            OpF8 oa, ob;
            memcpy(&oa, s[i + 16 * (k >> 3)].b, sizeof(oa));
            memcpy(&ob, s[j + 16 * (k >> 3)].b, sizeof(ob));
            acc = fmaf(e4m3_to_f32(oa.a[k & 7]), e4m3_to_f32(ob.b[k & 7]), acc);
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}

// int8 x int8 MFMA forms: lane l = (row / column l & (T-1), k-group l / T), 16 bytes per operand; A's byte e of group g meets
// B's byte e of group g (any k order inside the instruction gives the same sum)
namespace hipsim_detail { struct OpI8 { int8_t a[16], b[16]; }; }
inline i32x16 mmi_mfma_i8_32x32x32(u32x4 a, u32x4 b, i32x16 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpI8 me;
    memcpy(me.a, &a, 16);
    memcpy(me.b, &b, 16);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    i32x16 d;
    const int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int acc = c[r];
        for (int g = 0; g < 2; ++g) {
            OpI8 oa, ob;
            memcpy(&oa, s[i + 32 * g].b, sizeof(oa));
            memcpy(&ob, s[j + 32 * g].b, sizeof(ob));
            for (int e = 0; e < 16; ++e) acc += (int)oa.a[e] * (int)ob.b[e];
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}
inline i32x4 mmi_mfma_i8_16x16x64(u32x4 a, u32x4 b, i32x4 c) {
    using namespace hipsim_detail;
    hipsim::Slot* s = hipsim::wave_slots();
    const int l = hipsim::lane_id();
    OpI8 me;
    memcpy(me.a, &a, 16);
    memcpy(me.b, &b, 16);
    memcpy(s[l].b, &me, sizeof(me));
    hipsim::sync_wave();
    i32x4 d;
    const int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (l >> 4) + r;
        int acc = c[r];
        for (int g = 0; g < 4; ++g) {
            OpI8 oa, ob;
            memcpy(&oa, s[i + 16 * g].b, sizeof(oa));
            memcpy(&ob, s[j + 16 * g].b, sizeof(ob));
            for (int e = 0; e < 16; ++e) acc += (int)oa.a[e] * (int)ob.b[e];
        }
        d[r] = acc;
    }
    hipsim::sync_wave();
    return d;
}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// sum over groups of W consecutive lanes, every lane receiving the total (the gfx950 build: DPP row operations)
template <int W>
inline float mmi_group_sum(float x) {
    for (int m = 1; m < W; m <<= 1) x += mmi_shfl_xor(x, m);
    return x;
}
inline float mmi_fma(float a, float b, float c) { return fmaf(a, b, c); }
inline float mmi_dot2_bf16(uint32_t a, uint32_t b, float c) {
    using hipsim_detail::bf;
    return fmaf(bf((uint16_t)(a >> 16)), bf((uint16_t)(b >> 16)), fmaf(bf((uint16_t)a), bf((uint16_t)b), c));
}
inline f32x2 mmi_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
inline float mmi_rint(float x) { return nearbyintf(x); }
inline uint32_t mmi_pack_low_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return (a & 0xffu) | ((b & 0xffu) << 8) | ((c & 0xffu) << 16) | ((d & 0xffu) << 24);
}

inline unsigned mmi_arrive_release(unsigned* counter) { return __atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL); }
inline void mmi_acquire_agent() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
inline void mmi_store_relaxed_agent(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

inline u32x4 mmi_load_nt(const u32x4* p) { return *p; }
inline f32x4 mmi_load_nt(const f32x4* p) { return *p; }
inline float mmi_rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float mmi_fast_logf(float x) { return logf(x); }
inline unsigned mmi_wave_sum_u32(unsigned x) { for (int m = 1; m < 64; m <<= 1) x += mmi_shfl_xor(x, m); return x; }
inline unsigned mmi_wave_max_u32(unsigned x) { for (int m = 1; m < 64; m <<= 1) { const unsigned o = mmi_shfl_xor(x, m); x = o > x ? o : x; } return x; }
inline unsigned mmi_atomic_add(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// hand-off flags (duplex.hip).  The simulator runs every launch synchronously in host order, so a wait whose producer has not
// been launched yet can never be satisfied: abort loudly instead of spinning forever.
inline long mmi_wall_clock() { static long tick = 0; return ++tick; }
inline void mmi_flag_publish(long* flag, long v) { __atomic_store_n(flag, v, __ATOMIC_RELEASE); }
inline void mmi_flag_wait(const long* flag, long v) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < v) {
        fprintf(stderr, "hipsim: mmi_flag_wait on a flag whose producer was not enqueued first (have %ld, want %ld)\n", *flag, v);
        abort();
    }
}
