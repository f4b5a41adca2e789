// Standalone reproducer attempt for round 4's non-reproducible launch (HISTORY.md round 5 10a): the RoPE rotation of
// k_gemm_xp<32, 2, 1, 8, 4, 3>'s epilogue as hipcc (ROCm 7.2) compiled it THEN - packed fp32 math with op_sel swizzles, one
// v_pk_mul_f32 writing its own source pair in place - copied instruction for instruction, with the same register numbers, from
// that build's assembly (git a5bec5b, lm_engine.hip -> k_gemm_xp<32, 2, 1, 8, 4, 3>, block %bb.62), run ~10^8 wave executions
// against the same arithmetic written as plain IEEE operations:
//     v_cvt_pk_bf16_f32 x 4, unpack        (the in_proj output is a bf16 tensor)
//     global_load_dwordx4 x 2              (cos, sin of the four pairs)
//     v_pk_mul_f32 x 8 (op_sel forms, the last one IN PLACE and cross-swizzled), v_pk_add_f32 x 7, v_sub_f32, v_mov_b32 x 4
// Variants: 0 = the sequence as it stood; 1 = s_nop 7 after the in-place instruction; 2 = the in-place instruction writing a fresh
// register pair instead; 3 = variant 0 under a partial EXEC mask (the kernel ran it for the q / k features only).
// In the kernel the failure showed in lanes 48-63 only, in the LAST output (imaginary part of the fourth pair), value 0, a few
// times per million launches.  What this tool can say: whether the instruction sequence ALONE misbehaves on this chip.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/pk_hazard_repro.hip -o build/pk_hazard_repro && build/pk_hazard_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ float bf16r(float f) {      // round to nearest even to bf16, as v_cvt_pk_bf16_f32 does for finite values
    unsigned u = __builtin_bit_cast(unsigned, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __builtin_bit_cast(float, u & 0xffff0000u);
}

#define ROT_HEAD                                                                                   \
    "v_mov_b32 v2, %[s0]\n v_mov_b32 v3, %[s1]\n v_mov_b32 v4, %[s2]\n v_mov_b32 v5, %[s3]\n"      \
    "v_mov_b32 v6, %[s4]\n v_mov_b32 v7, %[s5]\n v_mov_b32 v8, %[s6]\n v_mov_b32 v9, %[s7]\n"      \
    "v_cvt_pk_bf16_f32 v12, v2, v3\n v_and_b32 v19, 0xffff0000, v12\n v_lshlrev_b32 v18, 16, v12\n"  \
    "v_cvt_pk_bf16_f32 v12, v4, v5\n v_and_b32 v17, 0xffff0000, v12\n v_lshlrev_b32 v16, 16, v12\n"  \
    "v_cvt_pk_bf16_f32 v12, v6, v7\n v_cvt_pk_bf16_f32 v14, v8, v9\n"                               \
    "v_and_b32 v13, 0xffff0000, v12\n v_lshlrev_b32 v12, 16, v12\n"                                 \
    "v_and_b32 v15, 0xffff0000, v14\n v_lshlrev_b32 v14, 16, v14\n"                                 \
    "v_mov_b32 v32, %[alo]\n v_mov_b32 v33, %[ahi]\n"                                               \
    "global_load_dwordx4 v[24:27], v[32:33], off\n"                                                 \
    "global_load_dwordx4 v[28:31], v[32:33], off offset:16\n"                                       \
    "s_waitcnt vmcnt(1)\n"                                                                          \
    "v_pk_mul_f32 v[32:33], v[24:25], v[18:19] op_sel_hi:[1,0]\n"                                   \
    "v_pk_mul_f32 v[24:25], v[24:25], v[18:19] op_sel:[1,1] op_sel_hi:[0,1]\n"                      \
    "v_pk_mul_f32 v[34:35], v[26:27], v[16:17] op_sel_hi:[1,0]\n"                                   \
    "v_pk_mul_f32 v[26:27], v[26:27], v[16:17] op_sel:[1,1] op_sel_hi:[0,1]\n"                      \
    "s_waitcnt vmcnt(0)\n"                                                                          \
    "v_pk_mul_f32 v[36:37], v[28:29], v[12:13] op_sel_hi:[1,0]\n"                                   \
    "v_pk_mul_f32 v[28:29], v[28:29], v[12:13] op_sel:[1,1] op_sel_hi:[0,1]\n"                      \
    "v_pk_mul_f32 v[38:39], v[30:31], v[14:15]\n"
#define ROT_TAIL(PAIR)                                                                             \
    "v_pk_add_f32 v[18:19], v[32:33], v[24:25]\n"                                                   \
    "v_pk_add_f32 v[16:17], v[34:35], v[26:27] neg_lo:[0,1] neg_hi:[0,1]\n"                         \
    "v_pk_add_f32 v[26:27], v[34:35], v[26:27]\n"                                                   \
    "v_pk_add_f32 v[12:13], v[36:37], v[28:29] neg_lo:[0,1] neg_hi:[0,1]\n"                         \
    "v_pk_add_f32 v[28:29], v[36:37], v[28:29]\n"                                                   \
    "v_pk_add_f32 v[30:31], v[38:39], v[38:39] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_add_f32 v[34:35], " PAIR ", " PAIR " op_sel:[0,1] op_sel_hi:[1,0]\n"                     \
    "v_sub_f32 v18, v32, v24\n v_mov_b32 v17, v27\n v_mov_b32 v13, v29\n v_mov_b32 v14, v30\n v_mov_b32 v15, v34\n" \
    "v_mov_b32 %[o0], v18\n v_mov_b32 %[o1], v19\n v_mov_b32 %[o2], v16\n v_mov_b32 %[o3], v17\n"    \
    "v_mov_b32 %[o4], v12\n v_mov_b32 %[o5], v13\n v_mov_b32 %[o6], v14\n v_mov_b32 %[o7], v15\n"
#define ROT_OPERANDS                                                                               \
    : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5]), [o6] "=&v"(o[6]), [o7] "=&v"(o[7]) \
    : [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [s3] "v"(s[3]), [s4] "v"(s[4]), [s5] "v"(s[5]), [s6] "v"(s[6]), [s7] "v"(s[7]), \
      [alo] "v"(alo), [ahi] "v"(ahi)                                                               \
    : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v24", "v25", "v26", "v27", \
      "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "memory"

template <int VARIANT>
__device__ __forceinline__ void rotate_asm(const float (&s)[8], const float* cs, float (&o)[8]) {
    const unsigned long long ap = (unsigned long long)cs;
    const unsigned alo = (unsigned)ap, ahi = (unsigned)(ap >> 32);
    if constexpr (VARIANT == 1) {
        asm volatile(ROT_HEAD "v_pk_mul_f32 v[14:15], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n s_nop 7\n" ROT_TAIL("v[14:15]") ROT_OPERANDS);
    } else if constexpr (VARIANT == 2) {
        asm volatile(ROT_HEAD "v_pk_mul_f32 v[40:41], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n" ROT_TAIL("v[40:41]") ROT_OPERANDS);
    } else {
        asm volatile(ROT_HEAD "v_pk_mul_f32 v[14:15], v[30:31], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]\n" ROT_TAIL("v[14:15]") ROT_OPERANDS);
    }
}

template <int VARIANT>
__global__ __launch_bounds__(512) void k_repro(const float* __restrict__ table, int table_rows, int iters, unsigned seed,
                                               unsigned long long* __restrict__ bad, unsigned* __restrict__ first) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned h = hash32(gid * 0x9E3779B9u + (unsigned)it * 0x85EBCA6Bu + seed);
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = ((float)(hash32(h + e) & 0xffffu) / 32768.0f - 1.0f) * 3.0f;
        const float* cs = table + (size_t)(h % (unsigned)table_rows) * 8;
        float o[8];
        const bool run = VARIANT != 3 || ((h >> 20) % 3u) != 2u;       // variant 3: a third of the lanes sit out, as the v features did
        if (run) rotate_asm<VARIANT == 3 ? 0 : VARIANT>(s, cs, o);
        if (!run) continue;
        // the same arithmetic, plain IEEE operations (compiled with -ffp-contract=off)
        float v8[8], e_[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = bf16r(s[e]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float c = cs[2 * j], sn = cs[2 * j + 1], re = v8[2 * j], im = v8[2 * j + 1];
            const float a = re * c, b2 = im * sn, c2 = re * sn, d = im * c;
            e_[2 * j] = a - b2;
            e_[2 * j + 1] = c2 + d;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (__builtin_bit_cast(unsigned, o[e]) != __builtin_bit_cast(unsigned, e_[e])) {
                if (nbad == 0 && atomicAdd(&first[0], 1u) < 8u) {
                    const unsigned slot = atomicAdd(&first[1], 1u);
                    if (slot < 8u) { first[2 + 4 * slot] = (threadIdx.x & 63u) | ((unsigned)e << 8) | ((unsigned)it << 12); first[3 + 4 * slot] = __builtin_bit_cast(unsigned, o[e]); first[4 + 4 * slot] = __builtin_bit_cast(unsigned, e_[e]); first[5 + 4 * slot] = gid; }
                }
                ++nbad;
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int VARIANT>
void run(const char* name, const float* table, int rows, unsigned long long* bad, unsigned* first, int blocks, int iters, int launches) {
    CK(hipMemset(bad, 0, 8)); CK(hipMemset(first, 0, 40 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k_repro<VARIANT>), dim3(blocks), dim3(512), 0, 0, table, rows, iters, 0x1234u + 977u * l, bad, first);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hb; unsigned hf[40];
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, sizeof(hf), hipMemcpyDeviceToHost));
    const double waves = (double)blocks * 8 * iters * launches;
    printf("%-58s %.3g wave executions in %d launches (%.0f ms): %llu outputs differ from the IEEE form\n", name, waves, launches, ms, hb);
    for (unsigned i = 0; i < hf[1] && i < 8; ++i)
        printf("     lane %2u output %u iteration %u thread %u: got %08x expected %08x\n", hf[2 + 4 * i] & 63u, (hf[2 + 4 * i] >> 8) & 15u, hf[2 + 4 * i] >> 12, hf[5 + 4 * i], hf[3 + 4 * i], hf[4 + 4 * i]);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, launches = argc > 2 ? atoi(argv[2]) : 20, blocks = 512;
    const int rows = 1 << 16;
    float* table; unsigned long long* bad; unsigned* first;
    CK(hipMalloc(&table, (size_t)rows * 8 * 4)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&first, 40 * 4));
    float* h = (float*)malloc((size_t)rows * 8 * 4);
    for (int r = 0; r < rows; ++r)
        for (int j = 0; j < 4; ++j) { const double ang = 0.37 * r + 1.7 * j; h[r * 8 + 2 * j] = (float)__builtin_cos(ang); h[r * 8 + 2 * j + 1] = (float)__builtin_sin(ang); }
    CK(hipMemcpy(table, h, (size_t)rows * 8 * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d iterations per thread, %d launches of %d x 512 threads per variant\n", p.gcnArchName, iters, launches, blocks);
    run<0>("0 the sequence as compiled in round 4 (in place, swizzled)", table, rows, bad, first, blocks, iters, launches);
    run<1>("1 + s_nop 7 behind the in-place instruction", table, rows, bad, first, blocks, iters, launches);
    run<2>("2 the same instruction writing a fresh register pair", table, rows, bad, first, blocks, iters, launches);
    run<3>("3 variant 0 under a partial EXEC mask", table, rows, bad, first, blocks, iters, launches);
    return 0;
}
