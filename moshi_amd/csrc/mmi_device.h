// Device-side primitives for the gfx950 (MI355X / CDNA4) build.  Every kernel in csrc/ is written
// against this small vocabulary: 64-lane wave shuffles, the four MFMA shapes we use, non-temporal
// 16-byte weight loads and the launch macro.  This file is the ONLY place the amdgcn builtins are
// named, so the kernels read as algorithms rather than intrinsic soup.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 mmi_bf16x8 __attribute__((ext_vector_type(8)));

#define MMI_WAVE 64
#define MMI_SHARED __shared__
#define MMI_DYN_SHARED(T, name)                                                   \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_[];  \
    T* name = reinterpret_cast<T*>(name##_raw_)
// every launch is announced to the launch-list recorder (mmi_graph.h: which site of the step issued which kernel; a
// no-op unless a program is being recorded)
void mmi_note_launch(const char* kernel);
#define MMI_LAUNCH(kern, grid, block, shmem, stream, ...)                    \
    do {                                                                     \
        mmi_note_launch(#kern);                                              \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)

// float -> bf16, round-to-nearest-even.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32); host code (weight
// import helpers) takes the bit-level definition in mmi_common.h.
typedef __bf16 mmi_bf16x2 __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ uint16_t mmi_f32_to_bf16_bits(float f);
__host__ __device__ __forceinline__ uint16_t mmi_cvt_bf16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
#else
    return mmi_f32_to_bf16_bits(f);
#endif
}
__host__ __device__ __forceinline__ uint32_t mmi_cvt_pk_bf16(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, mmi_bf16x2));
#else
    return (uint32_t)mmi_f32_to_bf16_bits(lo) | ((uint32_t)mmi_f32_to_bf16_bits(hi) << 16);
#endif
}

__device__ __forceinline__ int mmi_lane() { return (int)(threadIdx.x & 63u); }

template <class T>
__device__ __forceinline__ T mmi_shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
template <class T>
__device__ __forceinline__ T mmi_shfl(T v, int src) { return __shfl(v, src, 64); }

// Sum over groups of W consecutive lanes (W = 2, 4, 8, 16; a group never straddles a 16-lane row), every lane of the group
// receiving the total: DPP row operations - quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - which are
// plain VALU operand modifiers.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0) per step: a dependent trip
// through the LDS crossbar each (the decode attention spent most of its issue slots waiting on four of them per key row).
template <int W>
__device__ __forceinline__ float mmi_group_sum(float x) {
    static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16, "group of 1..16 lanes inside a DPP row");
    if constexpr (W >= 2) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, true));
    if constexpr (W >= 4) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, true));
    if constexpr (W >= 8) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));
    if constexpr (W >= 16) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));
    return x;
}
// sum / maximum of an unsigned over the whole wave, every lane gets it: four DPP steps inside the rows of 16, two crossbar steps
// across the rows (a six-step __shfl_xor butterfly is six dependent trips through the LDS crossbar)
__device__ __forceinline__ unsigned mmi_wave_sum_u32(unsigned x) {
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true);
    x += mmi_shfl_xor(x, 16);
    x += mmi_shfl_xor(x, 32);
    return x;
}
__device__ __forceinline__ unsigned mmi_wave_max_u32(unsigned x) {
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, true); x = o > x ? o : x;
    o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, true); x = o > x ? o : x;
    o = mmi_shfl_xor(x, 16); x = o > x ? o : x;
    o = mmi_shfl_xor(x, 32); x = o > x ? o : x;
    return x;
}
__device__ __forceinline__ float mmi_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// c + a.lo * b.lo + a.hi * b.hi on two packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16): no unpacking of either operand
typedef __bf16 mmi_bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float mmi_dot2_bf16(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mmi_bf16x2v, a), __builtin_bit_cast(mmi_bf16x2v, b), c, false);
}
// two fp32 FMAs in one instruction (v_pk_fma_f32)
__device__ __forceinline__ f32x2 mmi_pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// D(32x32) += A(32x2) * B(2x32), exact fp32 fma chain.  lane l: a = A[l&31][l>>5], b = B[l>>5][l&31];
// d[r] = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
__device__ __forceinline__ f32x16 mmi_mfma_f32_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// D(16x16) += A(16x4) * B(4x16).  lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]; d[r] = D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mmi_mfma_f32_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// D(32x32) += A(32x16) * B(16x32), bf16 in / fp32 acc.  lane l: a[e] = A[l&31][8*(l>>5)+e],
// b[e] = B[8*(l>>5)+e][l&31]; d as the 32x32 map above.
__device__ __forceinline__ f32x16 mmi_mfma_bf16_32x32x16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mmi_bf16x8, a),
                                                   __builtin_bit_cast(mmi_bf16x8, b), c, 0, 0, 0);
}
// D(16x16) += A(16x32) * B(32x16).  lane l: a[e] = A[l&15][8*(l>>4)+e], b[e] = B[8*(l>>4)+e][l&15];
// d[r] = D[4*(l>>4)+r][l&15].
__device__ __forceinline__ f32x4 mmi_mfma_bf16_16x16x32(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mmi_bf16x8, a),
                                                   __builtin_bit_cast(mmi_bf16x8, b), c, 0, 0, 0);
}

// One RoPE rotation (re, im) -> (re c - im s, re s + im c) in plain (un-packed) fp32 VALU instructions, each rounded once (no
// fma), pinned by inline asm so that the compiler cannot re-form them into packed-math (v_pk_mul_f32 / v_pk_add_f32 with op_sel
// swizzles) - see the note at its call site in lm_kernels.h (mmi_gemm_epilogue, MMI_EPI_ROPE_KV).  Arithmetic unchanged: the same
// six IEEE operations the C expression means under -ffp-contract=off.
__device__ __forceinline__ void mmi_rope_rotate(float re, float im, float c, float sn, float& out_re, float& out_im) {
    float a, b, d, e;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(re), "v"(c));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(im), "v"(sn));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(re), "v"(sn));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e) : "v"(im), "v"(c));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(out_re) : "v"(a), "v"(b));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(out_im) : "v"(d), "v"(e));
}

// ---- fp8 (OCP e4m3fn on gfx950: 4 exponent bits, bias 7, 3 mantissa bits, max 448, no infinity) -----------------------
// four fp32 -> four fp8 bytes (byte i = value i), round-to-nearest-even, clamped to +-448 first so that the result does not
// depend on the conversion's overflow mode
__device__ __forceinline__ uint32_t mmi_cvt_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
    b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
    d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (uint32_t)r;
}
// four fp8 bytes -> four fp32 (exact)
__device__ __forceinline__ void mmi_fp8x4_to_f32(uint32_t w, float* o) {
    const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
}
// D(32x32) += A(32x16) * B(16x32), fp8 in / fp32 acc.  Same element map as the bf16 32x32x16 form with one byte per
// element: lane l: a[e] = A[l&31][8*(l>>5)+e], b[e] = B[8*(l>>5)+e][l&31] (8 bytes per operand per lane).
__device__ __forceinline__ f32x16 mmi_mfma_fp8_32x32x16(u32x2 a, u32x2 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}
// D(16x16) += A(16x32) * B(32x16): lane l: a[e] = A[l&15][8*(l>>4)+e], b[e] = B[8*(l>>4)+e][l&15].
__device__ __forceinline__ f32x4 mmi_mfma_fp8_16x16x32(u32x2 a, u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
}

// ---- int8 x int8 on the matrix core (BASELINE configs[4]: the reference's bitsandbytes int8 matmul, utils/quantize.py:24-40) ----
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// D(32x32) += A(32x32) * B(32x32), int8 in / int32 acc (v_mfma_i32_32x32x32_i8): 16 bytes per operand per lane.  Both operands
// are packed with the SAME (lane, byte) -> k map (k_pack_w_i8: lane (i, kq), byte e <-> k = 32 kp + 16 (e >> 3) + 8 kq + (e & 7)),
// and the sum over k does not depend on the order the hardware walks it in.  d as the 32x32 map of the bf16 form.
__device__ __forceinline__ i32x16 mmi_mfma_i8_32x32x32(u32x4 a, u32x4 b, i32x16 c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
// D(16x16) += A(16x64) * B(64x16) (v_mfma_i32_16x16x64_i8); d[r] = D[4*(l>>4)+r][l&15].
__device__ __forceinline__ i32x4 mmi_mfma_i8_16x16x64(u32x4 a, u32x4 b, i32x4 c) {
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
// the low bytes of four words -> one word (byte i = low byte of argument i): two v_perm_b32 + one
__device__ __forceinline__ uint32_t mmi_pack_low_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u);      // bytes: [a.0, b.0, 0, 0]  (selector bytes 0-3 index the second operand, 4-7 the first, 0x0c = zero)
    const uint32_t cd = __builtin_amdgcn_perm(d, c, 0x04000c0cu);      // bytes: [0, 0, c.0, d.0]
    return ab | cd;
}
// round-half-even to the nearest integer (v_rndne_f32), the rounding of bitsandbytes' int8_vectorwise_quant
__device__ __forceinline__ float mmi_rint(float x) { return __builtin_rintf(x); }

// "Last arriver finishes" inside one launch (k_lm_attn_wave's ring split): a workgroup publishes its partial result with plain
// stores, then ONE lane releases at agent scope (L2 write-back), drains, and bumps the arrival counter; the workgroup that sees
// the final count acquires at agent scope (invalidates its CU's L1) and reads the others' partials with plain loads - the
// producer / consumer forms MI355X_MICROARCH.md lists as valid across XCDs.  Call from ONE thread, between two __syncthreads().
__device__ __forceinline__ unsigned mmi_arrive_release(unsigned* counter) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the compiler may drop the fence's own wait: keep one it cannot see through)
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mmi_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void mmi_store_relaxed_agent(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// streamed-once weights: non-temporal so they do not evict the activations / KV the other kernels reuse
__device__ __forceinline__ u32x4 mmi_load_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ f32x4 mmi_load_nt(const f32x4* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ float mmi_rsqrtf(float x) { return 1.0f / sqrtf(x); }  // IEEE, matches torch.rsqrt closely
// natural log on the hardware's v_log_f32 (log2, ~1 ulp) - for the sampler's own noise only (nothing the reference's arithmetic
// fixes): the library logf is ~25 dependent instructions on a one-wave-per-SIMD kernel that runs at ~9 cycles per instruction
__device__ __forceinline__ float mmi_fast_logf(float x) { return __logf(x); }
__device__ __forceinline__ unsigned mmi_atomic_add(unsigned* p, unsigned v) { return atomicAdd(p, v); }

// Cross-stream hand-off flags (duplex.hip): a monotonic counter in device memory, published with release semantics by a
// one-thread kernel at the end of a producer stream's work and polled by a one-wave kernel at the head of the consumer's.
// Why not hipStreamWaitEvent: a wait that stays PENDING makes the command processor poll the producer queue's signal, which
// slows every dependent launch of the producer (1.60 -> 2.77 ms for a chain of 1000 tiny kernels on this stack; a resident
// polling wave costs 1.71 ms - scripts/stream_probe.hip test 1e, profiles/r03_logs/stream_probe.txt).
// constant-rate device clock (100 MHz on gfx950): the pipeline's diagnostic stamps
__device__ __forceinline__ long mmi_wall_clock() { return (long)wall_clock64(); }
__device__ __forceinline__ void mmi_flag_publish(long* flag, long v) {
    __threadfence();
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// The poll is a RELAXED agent-scope load (it bypasses the non-coherent cache levels by itself); the acquire fence - on gfx950 an
// invalidation of the polling XCD's L2, which would evict the working set of every kernel running there once per iteration -
// comes once, after the flag was seen.
__device__ __forceinline__ void mmi_flag_wait(const long* flag, long v) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(96);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
