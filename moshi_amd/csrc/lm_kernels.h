// Moshi LM kernels for gfx950 (bf16 weights/activations, fp32 accumulate/norm/rope/softmax - the same rounding
// points as the reference's eager bf16 path: every nn.Linear output, norm output, rope output and residual add
// is rounded to bf16; transformer.py:45-58, rope.py:11-82, gating.py:13-22, lm.py:379-408,450-493).
//
// Data layout in HBM
//   linear weights  packed at load into MFMA A-fragment order, one contiguous 1 KiB fragment per
//                   (n-tile, k-step): TN=32 -> v_mfma_f32_32x32x16_bf16, lane l holds W[nt*32+(l&31)][ks*16+8*(l>>5)+e];
//                   TN=16 -> v_mfma_f32_16x16x32_bf16, lane l holds W[nt*16+(l&15)][ks*32+8*(l>>4)+e].
//                   A wave streams its K-slice of an n-tile as consecutive non-temporal 16-byte-per-lane loads.
//                   The SiLU-gated FFN's linear_in interleaves gate rows and value rows inside each tile so that
//                   the epilogue can form silu(g)*u without another pass (gating.py:19-20).
//   activations     between GEMMs: MFMA B-fragment order Xp[mt][ks][lane][8] (see "fragment-packed layouts" below), written
//                   by the producing epilogue; only q (for attention), the depformer's qkv and the logits are row-major.
//   KV ring         [layer][2][B][H][cap][Dh] bf16 (row = one position of one head: 256 contiguous bytes at Dh=128).
//   token ring      [B][17][max_delay+2] int32 (lm.py:605-613).
#pragma once
#include "mmi_common.h"

// ------------------------------------------------------------------------------------------------
// fragment-packed layouts
// ------------------------------------------------------------------------------------------------
// Every GEMM of the step is out[b][n] = sum_k x[b][k] * W[n][k] with a handful of batch rows (B <= 64) and weights
// that are read exactly once per step, i.e. an HBM stream.  Both operands are kept in HBM in the order the MFMA
// wants them in registers, so that one wave instruction (16 bytes per lane) is one contiguous, fully coalesced
// 1 KiB read that feeds one MFMA without any shuffling:
//   T = 32 (v_mfma_f32_32x32x16_bf16, 17..64 sessions): k-step = 16; lane l holds rows/cols (l & 31), k = 8*(l>>5)+e
//   T = 16 (v_mfma_f32_16x16x32_bf16, <= 16 sessions) : k-step = 32; lane l holds rows/cols (l & 15), k = 8*(l>>4)+e
//   weights     Wp[nt][ks][lane][e] = W[nt*T + (l & (T-1))][ks*KS + 8*(l / T) + e]            (zero padded)
//   activations Xp[mt][ks][lane][e] = x[mt*T + (l & (T-1))][ks*KS + 8*(l / T) + e]            (zero padded)
// A GEMM epilogue writes its result straight into the packed layout of the GEMM that consumes it.
MMI_HD long mmi_xp_index(int T, int b, int k, int ksteps) {
    const int sh = T == 32 ? 4 : 5;                 // log2(k-step)
    const int mt = b / T, bl = b - mt * T;
    const int ks = k >> sh, kq = (k >> 3) & ((1 << (sh - 3)) - 1), e = k & 7;
    return ((((long)mt * ksteps + ks) * 64) + kq * T + bl) * 8 + e;
}
static inline int mmi_kstep(int T) { return T == 32 ? 16 : 32; }

// gate_hidden == 0: plain [N][K] matrix.  gate_hidden == H: rows [0,H) are gates, [H,2H) values; tile nt carries
// gate rows nt*TN/2 .. and, in its second half, the matching value rows (so the epilogue forms silu(g)*u locally).
__global__ void k_pack_w_bf16(const uint16_t* __restrict__ W, uint16_t* __restrict__ P, int N, int K, int TN, int NT,
                              int KSTEPS, int gate_hidden) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NT * KSTEPS * 512;
    if (idx >= total) return;
    int e = (int)(idx & 7);
    int lane = (int)((idx >> 3) & 63);
    long rest = idx >> 9;
    int ks = (int)(rest % KSTEPS);
    int nt = (int)(rest / KSTEPS);
    int i, kq, kstep;
    if (TN == 32) { i = lane & 31; kq = lane >> 5; kstep = 16; } else { i = lane & 15; kq = lane >> 4; kstep = 32; }
    int k = ks * kstep + 8 * kq + e;
    long row;
    bool valid;
    if (gate_hidden > 0) {
        int half = TN / 2;
        int r = nt * half + (i < half ? i : i - half);
        valid = r < gate_hidden;
        row = (i < half ? 0 : gate_hidden) + r;
    } else {
        row = (long)nt * TN + i;
        valid = row < N;
    }
    uint16_t v = 0;
    if (valid && k < K) v = W[row * K + k];
    P[idx] = v;
}

// int8 weights (bitsandbytes-style per-output-row absmax quantisation, utils/quantize.py:13-22: `weight` int8 CB +
// `weight_scb` fp32 row absmax; W ~= CB * SCB / 127).  Same tile order as the bf16 packing with TWO k-steps per 16-byte
// lane entry: Wq[nt][kp][lane][16], entries 0..7 = k-step 2kp, 8..15 = k-step 2kp+1 (zero padded to an even count).
// The GEMM widens the bytes to bf16 in registers (exact: |q| <= 127) and scales the fp32 accumulator by SCB/127.
__global__ void k_pack_w_i8(const int8_t* __restrict__ W, int8_t* __restrict__ P, int N, int K, int TN, int NT, int KP,
                            int gate_hidden) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NT * KP * 1024;
    if (idx >= total) return;
    int e = (int)(idx & 15);
    int lane = (int)((idx >> 4) & 63);
    long rest = idx >> 10;
    int kp = (int)(rest % KP);
    int nt = (int)(rest / KP);
    int i, kq, kstep;
    if (TN == 32) { i = lane & 31; kq = lane >> 5; kstep = 16; } else { i = lane & 15; kq = lane >> 4; kstep = 32; }
    int k = (2 * kp + (e >> 3)) * kstep + 8 * kq + (e & 7);
    long row;
    bool valid;
    if (gate_hidden > 0) {
        int half = TN / 2;
        int r = nt * half + (i < half ? i : i - half);
        valid = r < gate_hidden;
        row = (i < half ? 0 : gate_hidden) + r;
    } else {
        row = (long)nt * TN + i;
        valid = row < N;
    }
    int8_t v = 0;
    if (valid && k < K) v = W[row * K + k];
    P[idx] = v;
}

// 16 int8 -> two bf16 fragments (k-step 2kp, k-step 2kp+1)
__device__ __forceinline__ void mmi_i8x16_to_bf16(u32x4 q, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t w = q[d];          // byte k sign-extended: shift it to the top as unsigned, arithmetic shift back down
        const float f0 = (float)((int)(w << 24) >> 24), f1 = (float)((int)(w << 16) >> 24), f2 = (float)((int)(w << 8) >> 24), f3 = (float)((int)w >> 24);
        const uint32_t p0 = mmi_pack_bf16x2(f0, f1), p1 = mmi_pack_bf16x2(f2, f3);
        if (d < 2) { lo[2 * d] = p0; lo[2 * d + 1] = p1; }
        else { hi[2 * (d - 2)] = p0; hi[2 * (d - 2) + 1] = p1; }
    }
}

// fp8 linears (BASELINE configs[4]: fp8 MFMA GEMMs): `weight` e4m3fn codes + `weight_scale` fp32 per output row (W ~= code *
// scale), packed exactly like the int8 bytes (two k-steps per 16-byte lane entry).  The activations stay bf16 in HBM and are
// converted to e4m3 in registers on their way into v_mfma_f32_{32x32x16,16x16x32}_fp8_fp8: x8 = e4m3(x / input_scale) with a
// static per-linear `input_scale` (a calibration constant, default 1), and the fp32 accumulator is scaled by
// weight_scale[row] * input_scale in the epilogue (folded into one per-row factor at load).
// 8 bf16 (one activation fragment) -> 8 fp8 bytes
__device__ __forceinline__ u32x2 mmi_bf16x8_to_fp8(u32x4 x, float inv) {
    float f[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f[2 * q] = __builtin_bit_cast(float, x[q] << 16) * inv;
        f[2 * q + 1] = __builtin_bit_cast(float, x[q] & 0xffff0000u) * inv;
    }
    u32x2 r;
    r[0] = mmi_cvt_fp8x4(f[0], f[1], f[2], f[3]);
    r[1] = mmi_cvt_fp8x4(f[4], f[5], f[6], f[7]);
    return r;
}

// ---- int8 activations (bitsandbytes int8_vectorwise_quant, utils/quantize.py:24-40; restated in oracle/lm_oracle.py) ----------
// max |v| over 8 packed bf16 values
__device__ __forceinline__ float mmi_absmax_bf16x8(u32x4 v) {
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        m = fmaxf(m, fabsf(__builtin_bit_cast(float, v[q] << 16)));
        m = fmaxf(m, fabsf(__builtin_bit_cast(float, v[q] & 0xffff0000u)));
    }
    return m;
}
// CA = int8(round_half_even(x * (127 / SCA))): 8 bf16 values -> 8 bytes (scale = 127 / SCA, 0 for an all-zero row).
// t = x * scale is rounded to fp32 like bitsandbytes' product, then t + 1.5 * 2^23 rounds it to the nearest integer, ties to
// even - the fp32 adder IS round-half-even at that magnitude, |t| <= 127 - and leaves the two's-complement byte in the low 8 bits
// of the sum: 3 instructions per element + a byte permute per pair instead of rint / convert / mask / shift / or (the
// quantisation sits on the dependent chain of the depth transformer's GEMMs, where every workgroup converts its own copy)
__device__ __forceinline__ u32x2 mmi_quant_i8x8(u32x4 v, float scale) {
    const float magic = 12582912.0f;            // 1.5 * 2^23
    u32x2 r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned x0 = v[2 * h], x1 = v[2 * h + 1];
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x0 << 16) * scale + magic);
        const unsigned b = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x0 & 0xffff0000u) * scale + magic);
        const unsigned c = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x1 << 16) * scale + magic);
        const unsigned d = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, x1 & 0xffff0000u) * scale + magic);
        r[h] = mmi_pack_low_bytes(a, b, c, d);
    }
    return r;
}
__device__ __forceinline__ float mmi_i8_scale(float sca) { return sca > 0.f ? 127.0f / sca : 0.f; }
// The dequantisation of oracle/lm_oracle.py `linear_int8` - a RESTATEMENT of bitsandbytes' int8_mm_dequant, `out32 * (row_stats *
// col_stats) * (1 / 127^2)` left to right in fp32, as its "default" backend writes it; not verified against the library, whose CUDA
// kernel multiplies by a literal constant and returns fp16 (PARITY UNPINNED).  The engine's int8 x int8 linears
// are bit-identical to THAT RESTATEMENT for the same bf16 input row (tests: *_int8_linear_bit_exact_*)
__device__ __forceinline__ float mmi_i8_dequant(int out32, float sca, float scb) {
    return ((float)out32 * (sca * scb)) * (1.0f / (127.0f * 127.0f));
}

// ------------------------------------------------------------------------------------------------
// weight-streaming skinny GEMM
// ------------------------------------------------------------------------------------------------
// grid.x = groups of NTW n-tiles; the block's WAVES waves split K (contiguous slices), each wave streams its slice of
// the NTW weight tiles with U fragments per tile in flight in each of two register buffers (explicit double
// buffering: the loads of group g+1 are issued before the MFMAs of group g), reading the matching activation
// fragments from L2.  The waves' partial tiles are summed through LDS in a fixed order (deterministic), and the
// epilogue (bf16 rounding point of nn.Linear, residual add, SiLU gate, embedding add) writes 8 consecutive features
// of one session as one 16-byte vector.
enum { MMI_EPI_STORE = 0, MMI_EPI_RESID = 1, MMI_EPI_GATE = 2, MMI_EPI_EMB = 3, MMI_EPI_PARTIAL = 4, MMI_EPI_ROPE_KV = 5, MMI_EPI_DEP_QKV0 = 6 };
enum { MMI_OUT_ROWMAJOR = 0, MMI_OUT_PACKED = 1 };

struct GemmArgs {
    const u32x4* wp;        // packed weights [NT][KSTEPS][64]
    const u32x4* xp;        // packed activations [MT][KSTEPS][64]
    uint16_t* out;          // row-major [B][out_ld] or packed [.][out_ksteps][64][8]
    const uint16_t* resid;  // EPI_RESID: same geometry as out
    const uint16_t* emb;    // EPI_EMB: embedding table [rows][N]
    const int* tok;         // EPI_EMB: token per session, tok[(b % tok_rows) * tok_stride] (model rows beyond tok_rows are the
    int tok_stride;         // guidance twins of the sessions and take the same token)
    int tok_rows;
    int B, N, KSTEPS, NT;
    int out_mode, out_ld, out_ksteps;
    int epi;
    const float* wscale;    // int8 / fp8 weights: dequantisation factor per ORIGINAL weight row (gate rows [0,H), value rows [H,2H)); else null
    const float* wscb;      // int8 weights: the raw row absmax SCB (`weight_scb`), read by the int8 x int8 dequantisation
                            // out32 * (SCA[b] * SCB[n]) * (1 / 127^2) - the oracle's restatement of bitsandbytes' int8_mm_dequant, in its order
    float xinv;             // fp8: 1 / input_scale, applied to the activations before the e4m3 conversion
    int wq;                 // host side only: 0 bf16, 1 int8 (widened to bf16), 2 fp8 (fp8 MFMA) weights, 3 int8 weights x int8 activations
    const float* sx;        // WQ = 3 with pre-quantised activations: xp holds int8 entries Xq[mt][kp][lane][16] (k_quant_rows_i8 / the
                            // norm kernel) and sx[b] the row absmax they were scaled by (bitsandbytes' SCA); the epilogue multiplies
                            // the int32 sum by SCA[b] / 127 * SCB[n] / 127.  k_gemm_q8: written here for the epilogue (LDS)
    int gate_rows;          // H of a gated linear_in (value row of feature n is H + n)
    float* partial;         // EPI_PARTIAL: fp32 partial sums [gridDim.y][B][N] (K split over gridDim.y workgroups so that
                            // GEMMs with few n-tiles still cover every CU); summed by k_resid_rmsnorm
    // EPI_ROPE_KV (temporal in_proj): features [q | k | v], each H heads x Dh.  RoPE (rope.py:11-82, interleaved, fp32) on
    // q and k for the single new position offsets[b]; q -> qrot [B][H][Dh]; k, v -> ring slot offsets[b] % cap of
    // [B][H][cap][Dh] (RingKVCache scatter, transformer.py:243-250: written for every row, exec mask or not)
    uint16_t* qrot;
    uint16_t* kc;
    uint16_t* vc;
    const long* offsets;
    int H, Dh, cap;
    float max_period;
    const float* rope;      // [B][Dh/2][2]: (cos, sin) of the new position's angles, filled once per step by k_lm_prepare
    int kv8;                // the ring holds e4m3 bytes instead of bf16 (fp8 KV cache: half the attention stream)
    // k_gemm_xp_norm: RMSNorm of the input rows fused in front of the GEMM (xp holds the un-normalised rows)
    const uint16_t* alpha;  // [D]
    int D;                  // features of a row (the mean is over D, not the padded K)
    float eps;
    int osplit;             // k_gemm_xp: > 1 = every n-tile is shared by `osplit` workgroups, each owning TN / 8 / osplit of its row octets
                            // (grid.x = NT * osplit).  For GEMMs with few n-tiles (N = 1024 of the depth transformer: 32 tiles on
                            // 256 CUs), where one CU cannot pull its 64-180 KB of weights faster than ~25 GB/s: a workgroup's lanes of
                            // octets it does not own repeat an owned lane's address (no extra HBM traffic; those MFMA rows are garbage
                            // that nobody writes), the epilogue writes only the owned 8-feature groups.  Not for gated epilogues.
};

// The residual (or embedding) vector the thread's FIRST epilogue task will add, requested before the weight stream starts
// so that its latency is hidden behind the main loop instead of extending the epilogue.
template <int TN, int MT, int NTW>
__device__ __forceinline__ u32x4 mmi_gemm_prefetch_addend(const GemmArgs& a, int nt0) {
    u32x4 pre = {0u, 0u, 0u, 0u};
    if (a.epi != MMI_EPI_RESID && a.epi != MMI_EPI_EMB) return pre;
    const int G = TN / 8;
    const int q = (int)threadIdx.x;
    if (q >= NTW * MT * G * TN) return pre;
    const int bl = q % TN;
    int rest = q / TN;
    const int gi = rest % G;
    rest /= G;
    const int m = rest % MT, t = rest / MT;
    const int nt = nt0 + t, b = m * TN + bl, n0 = nt * TN + 8 * gi;
    if (nt >= a.NT || b >= a.B || n0 >= a.N) return pre;
    if (a.epi == MMI_EPI_RESID) {
        const uint16_t* rs = a.out_mode == MMI_OUT_PACKED ? a.resid + mmi_xp_index(TN, b, n0, a.out_ksteps) : a.resid + (long)b * a.out_ld + n0;
        pre = *reinterpret_cast<const u32x4*>(rs);
    } else {
        const int tk = a.tok[(long)(b % a.tok_rows) * a.tok_stride];
        if (tk != -1) pre = *reinterpret_cast<const u32x4*>(a.emb + (long)(tk < 0 ? 0 : tk) * a.N + n0);   // lm_utils.py:102-124
    }
    return pre;
}

// Split-K reduction across the workgroup's waves (fixed order -> deterministic) and the epilogue shared by the GEMM
// kernels: one task = 8 consecutive output features of one session, written as one 16-byte vector.
// EXT: the reduction scratch is handed in (`red_ext`, WAVES * NTW * MT * 64 * LS floats) instead of a static LDS array - for
// kernels that own all of the LDS themselves (k_gemm_xlds).
// g_lo / g_hi: only the tile's 8-feature groups [g_lo, g_hi) are written (k_gemm_xlds hands a tile's row octets to two
// workgroups when that balances the chip: 384 in_proj tiles over 256 CUs = 6 octets each); default = the whole tile.
template <int TN, int MT, int NTW, int WAVES, bool EXT = false>
__device__ __forceinline__ void mmi_gemm_epilogue(const GemmArgs& a, float (&accv)[NTW][MT][TN == 32 ? 16 : 4], int wave, int lane,
                                                  int nt0, u32x4 pre, float* red_ext = nullptr, int g_lo = 0, int g_hi = 4,
                                                  const float* sx_local = nullptr) {
    constexpr int R = TN == 32 ? 16 : 4;
    // ---- split-K reduction across the block's waves (fixed order -> deterministic).  LDS layout [wave][tile][lane][LS]:
    // a lane's accumulators are contiguous, so they go out and come back as 16-byte vectors; LS = 20 floats (80 B)
    // spreads the 8 lanes of a ds_*_b128 group over all 32 banks.
    constexpr int LS = R == 4 ? 4 : ((WAVES * NTW * MT * 64 * 20 * 4 <= 65536) ? 20 : 16);
    constexpr int NE = NTW * MT * 64 * LS;
    float* red;
    if constexpr (EXT) {
        red = red_ext;
    } else {
        MMI_SHARED __attribute__((aligned(16))) float red_static[WAVES * NE];
        red = red_static;
    }
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < R / 4; ++c) {
                const f32x4 v4 = {accv[t][m][4 * c], accv[t][m][4 * c + 1], accv[t][m][4 * c + 2], accv[t][m][4 * c + 3]};
                *reinterpret_cast<f32x4*>(&red[wave * NE + ((t * MT + m) * 64 + lane) * LS + 4 * c]) = v4;
            }
    __syncthreads();

    // ---- epilogue: one task = 8 consecutive output features of one session
    const bool gate = a.epi == MMI_EPI_GATE;
    const bool iacc = a.sx || sx_local;              // int8 x int8: accv holds int32 bit patterns
    const int rows_out = gate ? TN / 2 : TN;         // output features per tile
    const int G = rows_out / 8;                      // feature groups per tile
    const int ntasks = NTW * MT * G * TN;
    for (int q = (int)threadIdx.x; q < ntasks; q += WAVES * 64) {
        const int bl = q % TN;
        int rest = q / TN;
        const int gi = rest % G;
        rest /= G;
        const int m = rest % MT, t = rest / MT;
        const int nt = nt0 + t;
        const int b = m * TN + bl;
        const int n0 = nt * rows_out + 8 * gi;
        if (nt >= a.NT || b >= a.B || n0 >= a.N || gi < g_lo || gi >= g_hi) continue;
        const float* rb = red + (t * MT + m) * 64 * LS;
        // features 8*gi .. 8*gi+7 of the tile live in two lanes' accumulator quads (MFMA C layout, lm_kernels.h header):
        //   T = 32: rows i = 8gi+e -> register (e&3) + 4gi of lane bl + 32*(e>>2);  gate partner rows i+16 -> registers + 8
        //   T = 16: rows i = 8gi+e -> register e&3 of lane bl + 16*(2gi + (e>>2));  gate partner rows i+8 -> lanes + 32
        int off_lo, off_hi, off2_lo, off2_hi;
        if constexpr (TN == 32) {
            off_lo = bl * LS + 4 * gi; off_hi = (bl + 32) * LS + 4 * gi;
            off2_lo = off_lo + 8; off2_hi = off_hi + 8;
        } else {
            off_lo = (bl + 32 * gi) * LS; off_hi = (bl + 32 * gi + 16) * LS;
            off2_lo = (bl + 32) * LS; off2_hi = (bl + 48) * LS;
        }
        float s[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; s2[e] = 0.f; }
        if (iacc) {
            // int8 x int8: the waves' accumulators are int32 bit patterns; their sum is the EXACT out32 of bitsandbytes'
            // int8_linear_matmul, converted to fp32 once (round to nearest even, like `out32 * float_tensor` in torch)
            int si[8], si2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { si[e] = 0; si2[e] = 0; }
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const i32x4 lo = *reinterpret_cast<const i32x4*>(rb + w * NE + off_lo);
                const i32x4 hi = *reinterpret_cast<const i32x4*>(rb + w * NE + off_hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) { si[e] += lo[e]; si[4 + e] += hi[e]; }
            }
            if (gate) {
#pragma unroll
                for (int w = 0; w < WAVES; ++w) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(rb + w * NE + off2_lo);
                    const i32x4 hi = *reinterpret_cast<const i32x4*>(rb + w * NE + off2_hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { si2[e] += lo[e]; si2[4 + e] += hi[e]; }
                }
            }
            if (a.epi == MMI_EPI_PARTIAL) {      // split-K: the consumer (k_resid_rmsnorm) adds the integer partials, then dequantises
                int* pd = reinterpret_cast<int*>(a.partial) + ((long)blockIdx.y * a.B + b) * a.N + n0;
                *reinterpret_cast<i32x4*>(pd) = i32x4{si[0], si[1], si[2], si[3]};
                *reinterpret_cast<i32x4*>(pd + 4) = i32x4{si[4], si[5], si[6], si[7]};
                continue;
            }
            // int8_mm_dequant: out32 * (SCA[b] * SCB[n]) * (1 / 127^2), fp32, in that order
            const float sca = sx_local ? sx_local[m * TN + bl] : a.sx[b];      // sx_local: the kernel's own row absmax (LDS)
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.wscb + n0), c1 = *reinterpret_cast<const f32x4*>(a.wscb + n0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] = mmi_i8_dequant(si[e], sca, c0[e]);
                s[4 + e] = mmi_i8_dequant(si[4 + e], sca, c1[e]);
            }
            if (gate) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.wscb + a.gate_rows + n0), g1 = *reinterpret_cast<const f32x4*>(a.wscb + a.gate_rows + n0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s2[e] = mmi_i8_dequant(si2[e], sca, g0[e]);
                    s2[4 + e] = mmi_i8_dequant(si2[4 + e], sca, g1[e]);
                }
            }
        } else {
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(rb + w * NE + off_lo);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(rb + w * NE + off_hi);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += lo[e]; s[4 + e] += hi[e]; }
        }
        if (gate) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(rb + w * NE + off2_lo);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(rb + w * NE + off2_hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s2[e] += lo[e]; s2[4 + e] += hi[e]; }
            }
        }
        }
        if (a.wscale && !iacc) {   // int8 weights, bf16 activations: y = (sum_k q x) * SCB / 127, per original weight row
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.wscale + n0), c1 = *reinterpret_cast<const f32x4*>(a.wscale + n0 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] *= c0[e]; s[4 + e] *= c1[e]; }
            if (gate) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.wscale + a.gate_rows + n0), g1 = *reinterpret_cast<const f32x4*>(a.wscale + a.gate_rows + n0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s2[e] *= g0[e]; s2[4 + e] *= g1[e]; }
            }
        }
        if (a.epi == MMI_EPI_PARTIAL) {
            float* pd = a.partial + ((long)blockIdx.y * a.B + b) * a.N + n0;
            f32x4 lo = {s[0], s[1], s[2], s[3]}, hi = {s[4], s[5], s[6], s[7]};
            *reinterpret_cast<f32x4*>(pd) = lo;
            *reinterpret_cast<f32x4*>(pd + 4) = hi;
            continue;
        }
        if (a.epi == MMI_EPI_DEP_QKV0) {
            // in_proj of the depth transformer's FIRST micro-step: its attention sees one position, softmax over one score is
            // exactly 1, so the attention output IS v (1 * v / 1) - the epilogue writes k and v into position 0 of the frame's
            // cache (transformer.py:243-253) and v as out_proj's packed operand; q is not needed; no attention launch
            const int HD = a.H * a.Dh;
            const int sec = n0 / HD, hn = n0 - sec * HD, h = hn / a.Dh, d0 = hn - h * a.Dh;
            if (sec == 0) continue;
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = mmi_pack_bf16x2(s[2 * e], s[2 * e + 1]);      // nn.Linear output in bf16
            uint16_t* cache = (sec == 1 ? a.kc : a.vc) + (((long)b * a.H + h) * a.cap) * a.Dh + d0;
            *reinterpret_cast<u32x4*>(cache) = ov;
            if (sec == 2) *reinterpret_cast<u32x4*>(a.out + mmi_xp_index(TN, b, hn, a.out_ksteps)) = ov;
            continue;
        }
        if (a.epi == MMI_EPI_ROPE_KV) {
            const int HD = a.H * a.Dh;
            const int sec = n0 / HD, hn = n0 - sec * HD, h = hn / a.Dh, d0 = hn - h * a.Dh;
            const long off = a.offsets[b];
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = mmi_round_bf16(s[e]);          // in_proj output is a bf16 tensor
            if (sec < 2) {
                const float* cs = a.rope + ((long)b * (a.Dh / 2) + d0 / 2) * 2;      // 4 (cos, sin) pairs = 32 contiguous bytes
                const f32x4 cs0 = *reinterpret_cast<const f32x4*>(cs), cs1 = *reinterpret_cast<const f32x4*>(cs + 4);
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float c = j < 2 ? cs0[2 * j] : cs1[2 * j - 4], sn = j < 2 ? cs0[2 * j + 1] : cs1[2 * j - 3];
                    // mmi_rope_rotate = four multiplies, a subtract, an add as separate VALU instructions.  Written as
                    // `re * c - im * sn` the compiler packs the four pairs into v_pk_mul_f32 / v_pk_add_f32 with op_sel swizzles,
                    // one of them in place, and on gfx950 that sequence inside k_gemm_xp<32, 2, ..> (64 sessions) returned, now and
                    // then, a stale last element for lanes 48-63 - q / k of 16 sessions wrong in one feature, a different one
                    // from run to run: the "row 17" failure of the driver's round-4 suite (profiles/r05_logs/c5_trace_*.txt)
                    const float re = v8[2 * j], im = v8[2 * j + 1];
                    mmi_rope_rotate(re, im, c, sn, v8[2 * j], v8[2 * j + 1]);
                }
            }
            if (sec != 0 && a.kv8) {   // fp8 ring: the bf16 k / v values (k after RoPE, as the bf16 ring would hold them) -> e4m3
                uint8_t* d8 = reinterpret_cast<uint8_t*>(sec == 1 ? a.kc : a.vc) + (((long)b * a.H + h) * a.cap + (int)(off % a.cap)) * a.Dh + d0;
                u32x2 o8;
                o8[0] = mmi_cvt_fp8x4(mmi_round_bf16(v8[0]), mmi_round_bf16(v8[1]), mmi_round_bf16(v8[2]), mmi_round_bf16(v8[3]));
                o8[1] = mmi_cvt_fp8x4(mmi_round_bf16(v8[4]), mmi_round_bf16(v8[5]), mmi_round_bf16(v8[6]), mmi_round_bf16(v8[7]));
                *reinterpret_cast<u32x2*>(d8) = o8;
                continue;
            }
            uint16_t* dst;
            if (sec == 0) dst = a.qrot + (long)b * HD + hn;
            else dst = (sec == 1 ? a.kc : a.vc) + (((long)b * a.H + h) * a.cap + (int)(off % a.cap)) * a.Dh + d0;
            u32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = mmi_pack_bf16x2(v8[2 * e], v8[2 * e + 1]);
            *reinterpret_cast<u32x4*>(dst) = ov;
            continue;
        }
        uint16_t* dst = a.out_mode == MMI_OUT_PACKED ? a.out + mmi_xp_index(TN, b, n0, a.out_ksteps)
                                                     : a.out + (long)b * a.out_ld + n0;
        float o[8];
        if (gate) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = mmi_round_bf16(s[e]);
                const float u = mmi_round_bf16(s2[e]);
                const float act = mmi_round_bf16(g / (1.0f + expf(-g)));   // F.silu on a bf16 tensor
                o[e] = act * u;
            }
        } else if (a.epi == MMI_EPI_RESID) {
            const uint16_t* rs = a.out_mode == MMI_OUT_PACKED ? a.resid + mmi_xp_index(TN, b, n0, a.out_ksteps)
                                                              : a.resid + (long)b * a.out_ld + n0;
            const u32x4 rv = q == (int)threadIdx.x ? pre : *reinterpret_cast<const u32x4*>(rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint16_t h = (uint16_t)((e & 1) ? (rv[e >> 1] >> 16) : (rv[e >> 1] & 0xffffu));
                o[e] = mmi_round_bf16(s[e]) + mmi_bf16_to_f32(h);           // x_orig + update
            }
        } else if (a.epi == MMI_EPI_EMB) {
            u32x4 ev = pre;
            if (q != (int)threadIdx.x) {
                const int tk = a.tok[(long)(b % a.tok_rows) * a.tok_stride];
                ev = u32x4{0u, 0u, 0u, 0u};
                if (tk != -1) ev = *reinterpret_cast<const u32x4*>(a.emb + (long)(tk < 0 ? 0 : tk) * a.N + n0);   // lm_utils.py:102-124
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint16_t h = (uint16_t)((e & 1) ? (ev[e >> 1] >> 16) : (ev[e >> 1] & 0xffffu));
                o[e] = mmi_round_bf16(s[e]) + mmi_bf16_to_f32(h);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = s[e];                         // nn.Linear output in bf16
        }
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = mmi_pack_bf16x2(o[2 * e], o[2 * e + 1]);
        *reinterpret_cast<u32x4*>(dst) = ov;
    }
}

// WQ = 1 (int8) / 2 (fp8) weights: one 16-byte weight entry carries two k-steps, so the loop runs over k-step PAIRS
// (a.KSTEPS then counts pairs; the activation buffers hold 2*KSTEPS k-steps, zero padded).
// WQ = 3: int8 weight entries x int8 activation entries (a.xp = Xq, one 16-byte entry per weight entry, a.sx = row absmax) on
// v_mfma_i32_{32x32x32,16x16x64}_i8; the int32 sums are converted to fp32 once per wave and go through the common epilogue.
// The 16-row tile (<= 16 sessions) with bf16 weights is held to 128 registers: two 8-wave workgroups per CU, so that one's
// reduction + epilogue runs under the other's weight stream (136 registers unbounded; 8 spill, outside the loop).  Same-box at
// 1 / 8 / 16 sessions: -0.12 / -0.14 / -0.12 ms per step (in_proj 20.9 -> 19.4 us, linear_in 38.4 -> 35.5, text head 48.9 ->
// 44.2; profiles/r04_logs/call_m_summary.txt).
template <int TN, int MT, int NTW, int WAVES, int U, int WQ = 0>
__global__ __launch_bounds__(WAVES * 64, (TN == 16 && MT == 1 && NTW == 1 && WQ == 0 && WAVES == 8) ? 4 : 1) void k_gemm_xp(GemmArgs a) {
    constexpr bool W8 = WQ == 1;
    constexpr int R = TN == 32 ? 16 : 4;          // accumulator registers per MFMA tile
    constexpr int XS = (WQ == 1 || WQ == 2) ? 2 : 1;   // activation fragments per weight entry
    typedef int iacc_t __attribute__((ext_vector_type(R)));
    typedef float acc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    // octet sharing (a.osplit > 1, NTW == 1): workgroup = (n-tile, part); part owns the 8-feature groups [g_lo, g_hi)
    const int os = (NTW == 1 && a.osplit > 1) ? a.osplit : 1;
    const int bx = (int)blockIdx.x / os, part = (int)blockIdx.x - bx * os;
    const int nt0 = bx * NTW;
    const int g_lo = part * ((TN / 8) / os), g_hi = os > 1 ? g_lo + (TN / 8) / os : TN / 8;
    // weight fragments are stored [k-step][lane], lane = (k-group, row): rows of octets the workgroup does not own are read
    // from the first owned octet instead
    const int ro = (lane >> 3) & (TN / 8 - 1);
    const int wlane = (ro >= g_lo && ro < g_hi) ? lane : ((lane & ~((TN / 8 - 1) << 3)) | (g_lo << 3));
    const u32x4 pre = mmi_gemm_prefetch_addend<TN, MT, NTW>(a, nt0);

    // K range of this workgroup (gridDim.y > 1: split-K over workgroups), then of this wave
    const int kb_per = (a.KSTEPS + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = min(a.KSTEPS, (int)blockIdx.y * kb_per), kb1 = min(a.KSTEPS, kb0 + kb_per);
    const int kper = (kb1 - kb0 + WAVES - 1) / WAVES;
    const int ks0 = min(kb1, kb0 + wave * kper);
    const int nks = min(kb1, ks0 + kper) - ks0;

    const u32x4* wp[NTW];
    const u32x4* xp[MT];
#pragma unroll
    for (int t = 0; t < NTW; ++t) wp[t] = a.wp + ((long)min(nt0 + t, a.NT - 1) * a.KSTEPS + ks0) * 64 + wlane;
#pragma unroll
    for (int m = 0; m < MT; ++m) xp[m] = a.xp + ((long)m * a.KSTEPS + ks0) * XS * 64 + lane;

    acc_t acc[NTW][MT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[t][m][r] = 0.f;

    u32x4 wA[U][NTW], xA[U][MT][XS], wB[U][NTW], xB[U][MT][XS];
#define MMI_G_LOAD(W_, X_, base)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) W_[u][t] = mmi_load_nt(wp[t] + ((base) + u) * 64); \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                        \
            _Pragma("unroll") for (int x = 0; x < XS; ++x) X_[u][m][x] = xp[m][(((base) + u) * XS + x) * 64]; \
    }
#define MMI_G_MFMA(WF, XF, ACC)                                                               \
    if constexpr (TN == 32) ACC = mmi_mfma_bf16_32x32x16(WF, XF, ACC);                        \
    else ACC = mmi_mfma_bf16_16x16x32(WF, XF, ACC);
#define MMI_G_MFMA8(WF, XF, ACC)                                                              \
    if constexpr (TN == 32) ACC = mmi_mfma_fp8_32x32x16(WF, XF, ACC);                         \
    else ACC = mmi_mfma_fp8_16x16x32(WF, XF, ACC);
#define MMI_G_MMA8(W_, X_, u)                                                                 \
    {                                                                                         \
        u32x2 xq_[MT][2];                                                                     \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                      \
            xq_[m][0] = mmi_bf16x8_to_fp8(X_[u][m][0], a.xinv);                               \
            xq_[m][1] = mmi_bf16x8_to_fp8(X_[u][m][XS - 1], a.xinv);                          \
        }                                                                                     \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                     \
            const u32x2 w0_ = {W_[u][t][0], W_[u][t][1]}, w1_ = {W_[u][t][2], W_[u][t][3]};   \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                  \
                MMI_G_MFMA8(w0_, xq_[m][0], acc[t][m])                                        \
                MMI_G_MFMA8(w1_, xq_[m][1], acc[t][m])                                        \
            }                                                                                 \
        }                                                                                     \
    }
#define MMI_G_MMA1(W_, X_, u)                                                                 \
        if constexpr (WQ == 2) MMI_G_MMA8(W_, X_, u)                                          \
        else if constexpr (WQ == 3) {                                                       \
            _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                   \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                              \
                    if constexpr (TN == 32) acc[t][m] = __builtin_bit_cast(acc_t, mmi_mfma_i8_32x32x32(W_[u][t], X_[u][m][0], __builtin_bit_cast(iacc_t, acc[t][m]))); \
                    else acc[t][m] = __builtin_bit_cast(acc_t, mmi_mfma_i8_16x16x64(W_[u][t], X_[u][m][0], __builtin_bit_cast(iacc_t, acc[t][m]))); \
                }                                                                             \
        } else {                                                                              \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                     \
            if constexpr (W8) {                                                               \
                u32x4 wlo_, whi_;                                                             \
                mmi_i8x16_to_bf16(W_[u][t], wlo_, whi_);                                      \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) {                              \
                    MMI_G_MFMA(wlo_, X_[u][m][0], acc[t][m])                                  \
                    MMI_G_MFMA(whi_, X_[u][m][XS - 1], acc[t][m])                             \
                }                                                                             \
            } else {                                                                          \
                _Pragma("unroll") for (int m = 0; m < MT; ++m) { MMI_G_MFMA(W_[u][t], X_[u][m][0], acc[t][m]) } \
            }                                                                                 \
        }                                                                                     \
        }
#define MMI_G_MMA(W_, X_)                                                                     \
    _Pragma("unroll") for (int u = 0; u < U; ++u) MMI_G_MMA1(W_, X_, u)
    const int nfull = nks / U;
    if (nfull > 0) {
        // steady state has no conditional loads, so that the compiler's s_waitcnt vmcnt(N) before each MFMA only waits
        // for the older buffer and the loads of the next group stay in flight behind it
        MMI_G_LOAD(wA, xA, 0);
        int g = 0;
        for (; g + 2 < nfull; g += 2) {
            MMI_G_LOAD(wB, xB, (g + 1) * U);
            MMI_G_MMA(wA, xA);
            MMI_G_LOAD(wA, xA, (g + 2) * U);
            MMI_G_MMA(wB, xB);
        }
        if (nfull - g == 2) {
            MMI_G_LOAD(wB, xB, (g + 1) * U);
            MMI_G_MMA(wA, xA);
            MMI_G_MMA(wB, xB);
        } else {
            MMI_G_MMA(wA, xA);
        }
    }
    // remainder of the slice (fewer than U entries): requested together - one memory round trip, not one per entry - from
    // clamped (valid) entries, and only the live ones meet the matrix core
    const int rem = nks - nfull * U;
    if (rem > 0) {
#pragma unroll
        for (int u = 0; u < U - 1; ++u) {
            const int ks = nfull * U + min(u, rem - 1);
#pragma unroll
            for (int t = 0; t < NTW; ++t) wA[u][t] = mmi_load_nt(wp[t] + ks * 64);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int x = 0; x < XS; ++x) xA[u][m][x] = xp[m][(ks * XS + x) * 64];
        }
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
            if (u < rem) { MMI_G_MMA1(wA, xA, u) }
    }
#undef MMI_G_LOAD
#undef MMI_G_MMA
#undef MMI_G_MMA1
#undef MMI_G_MMA8
#undef MMI_G_MFMA
#undef MMI_G_MFMA8
    float accv[NTW][MT][R];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                accv[t][m][r] = acc[t][m][r];      // WQ == 3: the wave's int32 sum, as its bit pattern (summed as integers by the epilogue)
            }
    mmi_gemm_epilogue<TN, MT, NTW, WAVES>(a, accv, wave, lane, nt0, pre, nullptr, g_lo, g_hi);
}

// k_gemm_xp for GEMMs that are a LATENCY chain, not a stream (the depth transformer's linear_out: 5.8 MB behind 128 workgroups,
// 22 k-steps per wave): the wave's WHOLE K-slice (<= KMAX k-steps: weight fragment + MT activation fragments each) is requested
// before the first MFMA - one memory round trip where the double-buffered loop of k_gemm_xp makes three (U = 4: 8 fragments in
// flight, 22 to fetch).  Same K partition over waves / workgroups, same k order inside a wave, same reduction and epilogue:
// bit-identical to k_gemm_xp<TN, MT, 1, WAVES, U>.  bf16 weights, one n-tile per workgroup, octet sharing and split-K over
// gridDim.y as in k_gemm_xp.  Registers: KMAX * (1 + MT) fragments of 4 (TN = 32, KMAX = 22: 176).
template <int TN, int MT, int WAVES, int KMAX>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_xp_once(GemmArgs a) {
    constexpr int R = TN == 32 ? 16 : 4;
    typedef float acc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int os = a.osplit > 1 ? a.osplit : 1;
    const int nt0 = (int)blockIdx.x / os, part = (int)blockIdx.x - nt0 * os;
    const int g_lo = part * ((TN / 8) / os), g_hi = os > 1 ? g_lo + (TN / 8) / os : TN / 8;
    const int ro = (lane >> 3) & (TN / 8 - 1);
    const int wlane = (ro >= g_lo && ro < g_hi) ? lane : ((lane & ~((TN / 8 - 1) << 3)) | (g_lo << 3));
    const u32x4 pre = mmi_gemm_prefetch_addend<TN, MT, 1>(a, nt0);
    const int kb_per = (a.KSTEPS + (int)gridDim.y - 1) / (int)gridDim.y;
    const int kb0 = min(a.KSTEPS, (int)blockIdx.y * kb_per), kb1 = min(a.KSTEPS, kb0 + kb_per);
    const int kper = (kb1 - kb0 + WAVES - 1) / WAVES;            // <= KMAX (checked by the launcher)
    const int ks0 = min(kb1, kb0 + wave * kper);
    const int nks = min(kb1, ks0 + kper) - ks0;
    const int ksl = min(ks0, a.KSTEPS - 1), last = nks > 0 ? nks - 1 : 0;
    const u32x4* wp = a.wp + ((long)min(nt0, a.NT - 1) * a.KSTEPS + ksl) * 64 + wlane;
    u32x4 wv[KMAX], xv[MT][KMAX];
    // unconditional loads from clamped (valid) addresses; the entries past the slice never meet the matrix core
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        const int uu = min(u, last);
        wv[u] = mmi_load_nt(wp + uu * 64);
#pragma unroll
        for (int m = 0; m < MT; ++m) xv[m][u] = a.xp[((long)m * a.KSTEPS + ksl + uu) * 64 + lane];
    }
    acc_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        if (u < nks) {                                           // wave-uniform
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (TN == 32) acc[m] = mmi_mfma_bf16_32x32x16(wv[u], xv[m][u], acc[m]);
                else acc[m] = mmi_mfma_bf16_16x16x32(wv[u], xv[m][u], acc[m]);
            }
        }
    }
    float accv[1][MT][R];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) accv[0][m][r] = acc[m][r];
    mmi_gemm_epilogue<TN, MT, 1, WAVES>(a, accv, wave, lane, nt0, pre, nullptr, g_lo, g_hi);
}

// RMSNorm fused into the GEMM that consumes it (the depth transformer: norm1 -> in_proj, norm2 -> linear_in; rows of
// 1024 features): y = (x.float() * (alpha.float() * rsqrt(eps + mean(x^2)))).to(bf16) (transformer.py:45-58), then
// out = y @ W^T.  A wave's whole K-slice of activation and weight fragments (<= KMAX k-steps) is loaded in one go -
// everything in flight at once, these GEMMs are latency bound - the waves combine their sums of squares through
// LDS, normalise their own fragments in registers and only then run the MFMAs.  No split-K over workgroups here.
// WQ = 1 / 2: int8 / fp8 weights, a.KSTEPS counts k-step pairs (see k_gemm_xp); KMAX = weight entries per wave.
template <int TN, int MT, int WAVES, int KMAX, int WQ = 0>
__global__ __launch_bounds__(WAVES * 64, (TN == 16 && KMAX == 4 && WQ == 0) ? 4 : 1) void k_gemm_xp_norm(GemmArgs a) {   // 16-row tile, short rows: two workgroups per CU
    constexpr bool W8 = WQ == 1;
    constexpr int R = TN == 32 ? 16 : 4;
    constexpr int KS = TN == 32 ? 16 : 32;
    constexpr int XS = WQ ? 2 : 1;
    constexpr int XMAX = KMAX * XS;                           // activation fragments per wave
    typedef float acc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int nt0 = (int)blockIdx.x;                          // one n-tile per workgroup (octet sharing of these GEMMs measured
    const int g_lo = 0, g_hi = TN / 8, wlane = lane;          // neutral, profiles/r02_logs/ab_osplit_norm*: not built in)
    const int kper = (a.KSTEPS + WAVES - 1) / WAVES;          // <= KMAX (checked by the launcher)
    const int ks0 = min(a.KSTEPS, wave * kper);
    const int nks = min(a.KSTEPS, ks0 + kper) - ks0;
    const int kq = TN == 32 ? (lane >> 5) : (lane >> 4);

    u32x4 wv[KMAX], xv[MT][XMAX], al[XMAX];
    const u32x4 zero = {0u, 0u, 0u, 0u};
    // every load is unconditional from a clamped (valid) address and masked afterwards: conditional loads would be
    // serialised behind s_waitcnt vmcnt(0) by the compiler, and these GEMMs live on having the whole slice in flight
    const int ksl = min(ks0, a.KSTEPS - 1);
    const int dmax = (a.D - 8) > 0 ? (a.D - 8) : 0;
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        const int uu = min(u, nks > 0 ? nks - 1 : 0);
        wv[u] = mmi_load_nt(a.wp + ((long)min(nt0, a.NT - 1) * a.KSTEPS + ksl + uu) * 64 + wlane);
#pragma unroll
        for (int x = 0; x < XS; ++x) {
            const int k = ((ksl + uu) * XS + x) * KS + 8 * kq;
            al[u * XS + x] = *reinterpret_cast<const u32x4*>(a.alpha + min(k, dmax));
#pragma unroll
            for (int m = 0; m < MT; ++m) xv[m][u * XS + x] = a.xp[(((long)m * a.KSTEPS + ksl + uu) * XS + x) * 64 + lane];
        }
    }
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        const bool on = u < nks;
        if (!on) wv[u] = zero;
#pragma unroll
        for (int x = 0; x < XS; ++x) {
            const int k = ((ks0 + u) * XS + x) * KS + 8 * kq;
            if (!on || k >= a.D) al[u * XS + x] = zero;
#pragma unroll
            for (int m = 0; m < MT; ++m) if (!on) xv[m][u * XS + x] = zero;
        }
    }
    // sum of squares of this lane's row over the wave's slice (the kq lane groups hold different k of the same row)
    MMI_SHARED float ssum[WAVES][MT][TN];
    float rs[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < XMAX; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = mmi_bf16_to_f32((uint16_t)(xv[m][u][q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(xv[m][u][q] >> 16));
                ss += lo * lo;
                ss += hi * hi;
            }
        if constexpr (TN == 32) ss += mmi_shfl_xor(ss, 32);
        else { ss += mmi_shfl_xor(ss, 16); ss += mmi_shfl_xor(ss, 32); }
        if (lane < TN) ssum[wave][m][lane] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) tot += ssum[w][m][lane & (TN - 1)];
        rs[m] = mmi_rsqrtf(a.eps + tot / (float)a.D);
    }
    acc_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        u32x4 wf[XS];
        if constexpr (W8) mmi_i8x16_to_bf16(wv[u], wf[0], wf[XS - 1]);
        else wf[0] = wv[u];
        const u32x2 w8f[2] = {{wv[u][0], wv[u][1]}, {wv[u][2], wv[u][3]}};   // fp8: the two k-steps of the entry
#pragma unroll
        for (int x = 0; x < XS; ++x) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                u32x4 xn;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4& xr = xv[m][u * XS + x];
                    const u32x4& ar = al[u * XS + x];
                    const float lo = mmi_bf16_to_f32((uint16_t)(xr[q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(xr[q] >> 16));
                    const float alo = mmi_bf16_to_f32((uint16_t)(ar[q] & 0xffffu)), ahi = mmi_bf16_to_f32((uint16_t)(ar[q] >> 16));
                    xn[q] = mmi_pack_bf16x2(lo * (alo * rs[m]), hi * (ahi * rs[m]));
                }
                if constexpr (WQ == 2) {        // the norm output is a bf16 tensor; it is that tensor the linear quantises
                    const u32x2 x8 = mmi_bf16x8_to_fp8(xn, a.xinv);
                    if constexpr (TN == 32) acc[m] = mmi_mfma_fp8_32x32x16(w8f[x], x8, acc[m]);
                    else acc[m] = mmi_mfma_fp8_16x16x32(w8f[x], x8, acc[m]);
                } else if constexpr (TN == 32) acc[m] = mmi_mfma_bf16_32x32x16(wf[x], xn, acc[m]);
                else acc[m] = mmi_mfma_bf16_16x16x32(wf[x], xn, acc[m]);
            }
        }
    }
    float accv[1][MT][R];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            accv[0][m][r] = acc[m][r];
        }
    mmi_gemm_epilogue<TN, MT, 1, WAVES>(a, accv, wave, lane, nt0, u32x4{0u, 0u, 0u, 0u}, nullptr, g_lo, g_hi);
}

// int8 weights x int8 activations with the row quantisation INSIDE the GEMM (the depth transformer's linears, whose rows are
// short enough - <= 8 * KMAX entries of 32 / 64 k - for one workgroup to hold): bitsandbytes' int8_vectorwise_quant needs the
// absmax of the WHOLE input row, and every workgroup of these N-split GEMMs reads the whole row anyway, so each derives it
// itself - no extra launch on the depth transformer's dependent chain, no cross-workgroup traffic.  Per batch tile: the wave's
// bf16 fragments are loaded in one go, (NORM: RMSNorm'ed in registers exactly as k_gemm_xp_norm does - the norm's output is the
// bf16 tensor the linear quantises,) the block reduces the rows' absmax through LDS, every lane converts its own fragments
// (round half even) and feeds v_mfma_i32_{32x32x32,16x16x64}_i8; the epilogue scales the int32 sums by SCA[b] * SCB[n] / 127^2.
// a.KSTEPS counts weight entries (two bf16 k-steps each).
// gridDim.y > 1 (MT == 1 only): ONE BATCH TILE PER WORKGROUP - blockIdx.y picks the tile, and the workgroup sees the GEMM of those
// TN rows alone (operand, output, residual, frame-cache pointers moved to the tile; B = the rows it holds).  At 33..64 sessions the
// MT = 2 form walks the two tiles one after the other through the same registers, three barriers each, on a latency-bound chain
// (dep.ffn_out 19.5 us against 8.2 for the bf16 kernel at 32 sessions, profiles/r04_q8_b64_sites.csv); the weights are 3-6 MB per
// linear and a tile's second reader finds them in its XCD's L2 (grid.x is a multiple of 8: both readers of an n-tile share an XCD).
template <int TN>
__device__ __forceinline__ void mmi_gemm_batch_tile_view(GemmArgs& a, int y, int x_frags_per_tile) {
    const int r0 = y * TN;
    a.xp += (long)y * x_frags_per_tile * 64;
    const long oshift = a.out_mode == MMI_OUT_PACKED ? (long)y * a.out_ksteps * 512 : (long)r0 * a.out_ld;
    if (a.out) a.out += oshift;
    if (a.resid) a.resid += oshift;
    if (a.kc) { a.kc += (long)r0 * a.H * a.cap * a.Dh; a.vc += (long)r0 * a.H * a.cap * a.Dh; }
    a.B = min(TN, a.B - r0);
}

template <int TN, int MT, int WAVES, int KMAX, bool NORM>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_q8(GemmArgs a_in) {
    GemmArgs a = a_in;
    if constexpr (MT == 1) { if (gridDim.y > 1) mmi_gemm_batch_tile_view<TN>(a, (int)blockIdx.y, a.KSTEPS * 2); }
    constexpr int R = TN == 32 ? 16 : 4;
    constexpr int KS = TN == 32 ? 16 : 32;
    constexpr int XMAX = 2 * KMAX;
    // batch tiles handled together: both while a lane's fragments of both fit its registers (short rows), else one after the other
    constexpr int MG = (MT * XMAX <= 16) ? MT : 1;
    typedef int iacc_t __attribute__((ext_vector_type(R)));
    typedef float facc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    // octet sharing as in k_gemm_xp (a.osplit > 1): N = 1024 linears have 32 n-tiles for 256 CUs; a workgroup = (tile, part) owns
    // TN / 8 / osplit of the tile's 8-row groups, the lanes of the other groups repeat an owned lane's weight address
    const int os = a.osplit > 1 ? a.osplit : 1;
    const int nt0 = (int)blockIdx.x / os, part = (int)blockIdx.x - nt0 * os;
    const int g_lo = part * ((TN / 8) / os), g_hi = os > 1 ? g_lo + (TN / 8) / os : TN / 8;
    const int ro = (lane >> 3) & (TN / 8 - 1);
    const int wlane = (ro >= g_lo && ro < g_hi) ? lane : ((lane & ~((TN / 8 - 1) << 3)) | (g_lo << 3));
    const int kper = (a.KSTEPS + WAVES - 1) / WAVES;          // <= KMAX (checked by the launcher)
    const int ks0 = min(a.KSTEPS, wave * kper);
    const int nks = min(a.KSTEPS, ks0 + kper) - ks0;
    const int kq = TN == 32 ? (lane >> 5) : (lane >> 4);
    const u32x4 pre = mmi_gemm_prefetch_addend<TN, MT, 1>(a, nt0);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const int ksl = min(ks0, a.KSTEPS - 1);
    u32x4 wv[KMAX];
#pragma unroll
    for (int u = 0; u < KMAX; ++u) {
        const int uu = min(u, nks > 0 ? nks - 1 : 0);
        wv[u] = mmi_load_nt(a.wp + ((long)min(nt0, a.NT - 1) * a.KSTEPS + ksl + uu) * 64 + wlane);
    }
#pragma unroll
    for (int u = 0; u < KMAX; ++u) if (u >= nks) wv[u] = zero;
    MMI_SHARED float sxl[MT * TN];                       // the rows' absmax, for the epilogue
    MMI_SHARED float redl[2][WAVES][MG][TN];             // [0] sums of squares, [1] absmax, per wave and tile of the group
    iacc_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[m][r] = 0;
    const int dmax = (a.D - 8) > 0 ? (a.D - 8) : 0;
    u32x4 al[NORM ? XMAX : 1];
    if constexpr (NORM) {
#pragma unroll
        for (int u = 0; u < XMAX; ++u) {
            const int uu = min(u >> 1, nks > 0 ? nks - 1 : 0);
            al[u] = *reinterpret_cast<const u32x4*>(a.alpha + min(((ksl + uu) * 2 + (u & 1)) * KS + 8 * kq, dmax));
        }
#pragma unroll
        for (int u = 0; u < XMAX; ++u)
            if ((u >> 1) >= nks || ((ks0 + (u >> 1)) * 2 + (u & 1)) * KS + 8 * kq >= a.D) al[u] = zero;
    }
#pragma unroll
    for (int m0 = 0; m0 < MT; m0 += MG) {
        // unconditional loads from clamped (valid) addresses, masked afterwards (a load under a branch would be serialised)
        u32x4 xv[MG][XMAX];
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int u = 0; u < XMAX; ++u) {
                const int uu = min(u >> 1, nks > 0 ? nks - 1 : 0);
                xv[g][u] = a.xp[(((long)(m0 + g) * a.KSTEPS + ksl + uu) * 2 + (u & 1)) * 64 + lane];
            }
#pragma unroll
        for (int g = 0; g < MG; ++g)
#pragma unroll
            for (int u = 0; u < XMAX; ++u)
                if ((u >> 1) >= nks) xv[g][u] = zero;
        if constexpr (NORM) {       // y = (x.float() * (alpha.float() * rsqrt(eps + mean(x^2)))).to(bf16)   (transformer.py:45-58)
#pragma unroll
            for (int g = 0; g < MG; ++g) {
                float ss = 0.f;
#pragma unroll
                for (int u = 0; u < XMAX; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = mmi_bf16_to_f32((uint16_t)(xv[g][u][q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(xv[g][u][q] >> 16));
                        ss += lo * lo;
                        ss += hi * hi;
                    }
                if constexpr (TN == 32) ss += mmi_shfl_xor(ss, 32);
                else { ss += mmi_shfl_xor(ss, 16); ss += mmi_shfl_xor(ss, 32); }
                if (lane < TN) redl[0][wave][g][lane] = ss;
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < MG; ++g) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) tot += redl[0][w][g][lane & (TN - 1)];
                const float rs = mmi_rsqrtf(a.eps + tot / (float)a.D);
#pragma unroll
                for (int u = 0; u < XMAX; ++u) {
                    u32x4 xn;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = mmi_bf16_to_f32((uint16_t)(xv[g][u][q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(xv[g][u][q] >> 16));
                        const float alo = mmi_bf16_to_f32((uint16_t)(al[u][q] & 0xffffu)), ahi = mmi_bf16_to_f32((uint16_t)(al[u][q] >> 16));
                        xn[q] = mmi_pack_bf16x2(lo * (alo * rs), hi * (ahi * rs));
                    }
                    xv[g][u] = xn;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            float am = 0.f;
#pragma unroll
            for (int u = 0; u < XMAX; ++u) am = fmaxf(am, mmi_absmax_bf16x8(xv[g][u]));
            if constexpr (TN == 32) am = fmaxf(am, mmi_shfl_xor(am, 32));
            else { am = fmaxf(am, mmi_shfl_xor(am, 16)); am = fmaxf(am, mmi_shfl_xor(am, 32)); }
            if (lane < TN) redl[1][wave][g][lane] = am;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < MG; ++g) {
            float sca = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) sca = fmaxf(sca, redl[1][w][g][lane & (TN - 1)]);
            if (wave == 0 && lane < TN) sxl[(m0 + g) * TN + lane] = sca;
            const float scale = mmi_i8_scale(sca);
#pragma unroll
            for (int u = 0; u < KMAX; ++u) {
                const u32x2 qlo = mmi_quant_i8x8(xv[g][2 * u], scale), qhi = mmi_quant_i8x8(xv[g][2 * u + 1], scale);
                const u32x4 xq = {qlo[0], qlo[1], qhi[0], qhi[1]};
                if constexpr (TN == 32) acc[m0 + g] = mmi_mfma_i8_32x32x32(wv[u], xq, acc[m0 + g]);
                else acc[m0 + g] = mmi_mfma_i8_16x16x64(wv[u], xq, acc[m0 + g]);
            }
        }
        __syncthreads();                                 // redl is rewritten by the next group; sxl is read by the epilogue
    }
    float accv[1][MT][R];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r) accv[0][m][r] = __builtin_bit_cast(facc_t, acc[m])[r];     // int32 bit patterns (see the epilogue)
    mmi_gemm_epilogue<TN, MT, 1, WAVES>(a, accv, wave, lane, nt0, pre, nullptr, g_lo, g_hi, sxl);
}

// ------------------------------------------------------------------------------------------------
// weight-streaming GEMM with the activations resident in LDS
// ------------------------------------------------------------------------------------------------
// k_gemm_xp re-reads the whole activation matrix from L2 in every workgroup (one per n-tile): at 32 sessions that is as many
// L2 bytes as the weights themselves, at 64 twice as many, and the L2 (about 10-11 TB/s of 128-byte requests) - not HBM -
// becomes the bound (profiles/r01_pmc_sq_tcc_ffn_in_b32_b64.csv).  Here ONE workgroup per CU walks a contiguous
// range of up to NTMAX n-tiles.  K is cut into chunks of KC k-steps whose activation fragments (MT x KC KiB) are staged in LDS,
// double-buffered, so every activation byte leaves the L2 once per workgroup; the 8 waves split a chunk's k-steps, stream their
// KPW weight fragments per (chunk, tile) segment from HBM one segment ahead of the MFMAs (register double buffer) and read
// the activation fragments with ds_read_b128.  The tiles' accumulators stay in registers across the chunks; at the end each
// tile goes through the common split-K reduction + epilogue, with the reduction scratch laid over the (then idle) chunk buffers.
// Prototype in scripts/gemm_microbench.hip (main loop only): 31.7 us against 34.8 us at 32 sessions, 37.4 against 51 us at 64.
// bf16 weights, 32-row tiles.  Dynamic LDS = max(2 * MT * KC KiB, reduction scratch); a.KSTEPS % KC == 0.
// STAGGER (MMI_GEMM_LDS=2, not yet measured on hardware): in the LAST chunk each tile's reduction + epilogue runs right after
// its MFMAs, on the chunk buffer that is idle by then, while the next tile's weights (requested before those MFMAs) are still
// streaming - instead of all the tiles' epilogues one after the other behind a finished weight stream.
struct MmiTrue { static constexpr bool value = true; };
struct MmiFalse { static constexpr bool value = false; };
// WQ = 1 / 2: int8 / fp8 weight entries (two k-steps per 16-byte entry, see k_gemm_xp): KC and a.KSTEPS count ENTRIES and every
// entry meets XS = 2 activation fragments, so the production chunk is 32 / MT entries (the same 64 KiB of activations).
// Chunks shorter than 8 entries (the tiny shapes of the tests) leave the upper waves without k-steps: they keep zero
// accumulators and only take part in the barriers and the epilogue.
template <int MT, int KC, int NTMAX = 3, bool STAGGER = false, int WQ = 0>
__global__ __launch_bounds__(512) void k_gemm_xlds(GemmArgs a) {
    typedef float acc_t __attribute__((ext_vector_type(16)));
    constexpr int XS = (WQ == 1 || WQ == 2) ? 2 : 1;   // activation fragments per weight entry (WQ = 3: int8 entries, one each)
    constexpr int KPW = KC >= 8 ? KC / 8 : 1;   // entries per wave per chunk
    constexpr int ACTIVE = KC / KPW;            // waves that own k-steps
    constexpr int XE = MT * KC * XS * 64;       // 16-byte activation pieces per chunk
    constexpr int XPT = (XE + 511) / 512;       // ... per thread
    static_assert(KC % KPW == 0 && ACTIVE <= 8, "a chunk is split over at most 8 waves");
    MMI_DYN_SHARED(u32x4, xs);                  // [2][XE], later the reduction scratch
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool active = ACTIVE == 8 || wave < ACTIVE;
    const int wk = active ? wave : 0;           // idle waves point at valid addresses and never load
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    // The workgroup's share of the n-tiles, in row OCTETS (8 of a tile's 32 weight rows) when the epilogue writes 8-feature
    // groups that map one to one onto them (everything but the gated linear_in, whose tiles interleave gate and value rows):
    // 384 in_proj tiles over 256 workgroups are 6 octets = 1.5 tiles each instead of 1 or 2 whole tiles, i.e. every CU streams
    // the same number of bytes.  A workgroup that owns only part of a tile loads only those octets' lanes (the other lanes
    // repeat a valid lane's address: rows of the MFMA are independent, their results are simply not written).
    const bool by_octet = a.epi != MMI_EPI_GATE;
    const long units = by_octet ? 4L * a.NT : (long)a.NT;
    const long u0 = (long)bid * units / G, u1 = (long)(bid + 1) * units / G;          // [u0, u1) octets or tiles
    const int t0 = by_octet ? (int)(u0 >> 2) : (int)u0;
    const int t1 = by_octet ? (int)((u1 + 3) >> 2) : (int)u1;
    const int ntiles = u1 > u0 ? t1 - t0 : 0;   // <= NTMAX (the launcher sizes the grid)
    // octet window of tile t0 + t
    auto oct_lo = [&](int t) { return by_octet ? (int)max(0L, u0 - 4L * (t0 + t)) : 0; };
    auto oct_hi = [&](int t) { return by_octet ? (int)min(4L, u1 - 4L * (t0 + t)) : 4; };
    const int nchunks = a.KSTEPS / KC;
    acc_t acc[NTMAX][MT];
#pragma unroll
    for (int t = 0; t < NTMAX; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.f;
    // chunk c of the packed activations Xp[m][ks][lane] (XS fragments per entry) -> xs[buf][((m * KC + k) * XS + x) * 64 + lane]
    auto xsrc = [&](int c, int e) {
        const int m = e / (KC * XS * 64), rest = e - m * (KC * XS * 64);
        return a.xp + ((long)m * a.KSTEPS + (long)c * KC) * XS * 64 + rest;
    };
    u32x4 xpre[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
        const int e = min(j * 512 + tid, XE - 1);
        xpre[j] = *xsrc(0, e);
    }
#pragma unroll
    for (int j = 0; j < XPT; ++j)
        if (j * 512 + tid < XE) xs[j * 512 + tid] = xpre[j];
    auto wsrc = [&](int c, int t) {
        const int ro = (lane >> 3) & 3, lo = oct_lo(t), hi = oct_hi(t);
        const int ln = (ro >= lo && ro < hi) ? lane : ((lane & ~0x18) | (lo << 3));       // lanes of octets it does not own
        return a.wp + ((long)min(t0 + t, a.NT - 1) * a.KSTEPS + (long)c * KC + wk * KPW) * 64 + ln;
    };
    u32x4 cur[KPW], nxt[KPW];
    if (active) {
        const u32x4* wp = wsrc(0, 0);
#pragma unroll
        for (int i = 0; i < KPW; ++i) cur[i] = mmi_load_nt(wp + i * 64);
    }
    // one weight entry against the resident activations
    auto mma = [&](const u32x4& w, const u32x4* xe, acc_t (&ac)[MT]) {
        if constexpr (WQ == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m) ac[m] = mmi_mfma_bf16_32x32x16(w, xe[m * KC * XS * 64], ac[m]);
        } else if constexpr (WQ == 3) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                ac[m] = __builtin_bit_cast(acc_t, mmi_mfma_i8_32x32x32(w, xe[m * KC * XS * 64], __builtin_bit_cast(i32x16, ac[m])));
        } else if constexpr (WQ == 1) {
            u32x4 lo, hi;
            mmi_i8x16_to_bf16(w, lo, hi);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ac[m] = mmi_mfma_bf16_32x32x16(lo, xe[m * KC * XS * 64], ac[m]);
                ac[m] = mmi_mfma_bf16_32x32x16(hi, xe[m * KC * XS * 64 + 64], ac[m]);
            }
        } else {
            const u32x2 w0 = {w[0], w[1]}, w1 = {w[2], w[3]};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ac[m] = mmi_mfma_fp8_32x32x16(w0, mmi_bf16x8_to_fp8(xe[m * KC * XS * 64], a.xinv), ac[m]);
                ac[m] = mmi_mfma_fp8_32x32x16(w1, mmi_bf16x8_to_fp8(xe[m * KC * XS * 64 + 64], a.xinv), ac[m]);
            }
        }
    };
    // one chunk: LAST = the final chunk (nothing to prefetch; with STAGGER each tile's epilogue follows its MFMAs at once)
    auto run_chunk = [&](int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        __syncthreads();                        // chunk c is in xs[c & 1]; nobody reads xs[(c + 1) & 1] any more
        if constexpr (!LAST) {
#pragma unroll
            for (int j = 0; j < XPT; ++j) {
                const int e = min(j * 512 + tid, XE - 1);
                xpre[j] = *xsrc(c + 1, e);
            }
        }
        const u32x4* xb = xs + (c & 1) * XE + wk * KPW * XS * 64 + lane;
#pragma unroll
        for (int t = 0; t < NTMAX; ++t) {
            if (t < ntiles) {
                if (active) {
                    // the weights of the NEXT segment - (chunk c, tile t + 1), or (chunk c + 1, first tile) - are requested first
                    const bool last_tile = t == ntiles - 1;
                    const int nc = min(last_tile ? c + 1 : c, nchunks - 1), nt = last_tile ? 0 : t + 1;
                    const u32x4* wp = wsrc(nc, nt);
#pragma unroll
                    for (int i = 0; i < KPW; ++i) nxt[i] = mmi_load_nt(wp + i * 64);
#pragma unroll
                    for (int i = 0; i < KPW; ++i) mma(cur[i], xb + i * XS * 64, acc[t]);
#pragma unroll
                    for (int i = 0; i < KPW; ++i) cur[i] = nxt[i];
                }
                if constexpr (STAGGER && LAST) {
                    // scratch: the other chunk buffer (last read before this chunk's opening barrier, not refilled any more);
                    // the small test chunks are smaller than the scratch, which then sits behind both buffers
                    float* red = reinterpret_cast<float*>(XE * 16 >= 65536 ? xs + (nchunks & 1) * XE : xs + 2 * XE);
                    __syncthreads();            // the previous tile's output tasks are done with the scratch
                    float accv[1][MT][16];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accv[0][m][r] = acc[t][m][r];
                    mmi_gemm_epilogue<32, MT, 1, 8, true>(a, accv, wave, lane, t0 + t, u32x4{0u, 0u, 0u, 0u}, red, oct_lo(t), oct_hi(t));
                }
            }
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int j = 0; j < XPT; ++j)
                if (j * 512 + tid < XE) xs[((c + 1) & 1) * XE + j * 512 + tid] = xpre[j];
        }
    };
    for (int c = 0; c + 1 < nchunks; ++c) run_chunk(c, MmiFalse{});
    run_chunk(nchunks - 1, MmiTrue{});
    if constexpr (STAGGER) return;
    // ---- per tile: split-K reduction over the 8 waves + the common epilogue; the scratch overlays the chunk buffers
    float* red = reinterpret_cast<float*>(xs);
#pragma unroll
    for (int t = 0; t < NTMAX; ++t) {
        if (t < ntiles) {
            __syncthreads();                    // the chunk buffers / the previous tile's scratch are no longer read
            float accv[1][MT][16];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) accv[0][m][r] = acc[t][m][r];     // WQ == 3: int32 bit patterns (see the epilogue)
            mmi_gemm_epilogue<32, MT, 1, 8, true>(a, accv, wave, lane, t0 + t, u32x4{0u, 0u, 0u, 0u}, red, oct_lo(t), oct_hi(t));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm (rms_norm_f32, eps 1e-8): y = (x.float() * (alpha.float() * rsqrt(eps + mean(x^2)))).to(bf16)
// ------------------------------------------------------------------------------------------------
// One block per session row b; thread t owns the 16-byte piece (8 features) t, t + blockDim, ... of the packed row
// (launch with blockDim = D/8 up to 1024 so that a row is a single round of loads held in registers).
// P > 0: first finish the preceding split-K GEMM (out_proj / linear_out) and its residual connection,
//        x <- x + bf16(sum_p partial[p][b][:])      (transformer.py:739-741,772-776: x_orig + update, bf16 tensors)
// then   y <- rms_norm_f32(x) * alpha               (transformer.py:45-58)
#define MMI_NORM_MAXP 4      // pieces per thread kept in registers (D <= 8 * 1024 * MMI_NORM_MAXP)
// yq != null (int8 activations): the normalised row is also stored quantised row-wise for the int8 linears that read it -
// Xq[mt][kp][lane][16] (the bytes of k-steps 2 kp and 2 kp + 1) and sx[b] = its absmax (bitsandbytes' CA / SCA).
__global__ __launch_bounds__(1024) void k_resid_rmsnorm(uint16_t* __restrict__ x, const float* __restrict__ partial, int P,
                                                        int B, const uint16_t* __restrict__ alpha, uint16_t* __restrict__ y,
                                                        int D, int T, int ksteps, float eps, uint8_t* __restrict__ yq = nullptr,
                                                        float* __restrict__ sx = nullptr, const float* __restrict__ psx = nullptr,
                                                        const float* __restrict__ pscb = nullptr) {
    // psx / pscb != null: the pending GEMM was int8 x int8 - its partials are int32 sums (the exact out32 once added), and the
    // dequantisation out32 * (SCA[b] * SCB[n]) / 127^2 happens here (psx = the absmax of that GEMM's input rows, pscb = its SCB)
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    float f[MMI_NORM_MAXP][8];
    u32x4 al[MMI_NORM_MAXP];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        const long at = mmi_xp_index(T, b, i, ksteps);
        u32x4 v = *reinterpret_cast<const u32x4*>(x + at);
        al[j] = *reinterpret_cast<const u32x4*>(alpha + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f[j][2 * q] = mmi_bf16_to_f32((uint16_t)(v[q] & 0xffffu));
            f[j][2 * q + 1] = mmi_bf16_to_f32((uint16_t)(v[q] >> 16));
        }
        if (P > 0) {
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f;
            f32x4 plo[4], phi[4];                                  // P <= 4; unconditional (clamped) loads, all in flight
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float* pp = partial + ((long)min(p, P - 1) * B + b) * D + i;
                plo[p] = *reinterpret_cast<const f32x4*>(pp);
                phi[p] = *reinterpret_cast<const f32x4*>(pp + 4);
            }
            if (psx) {
                int si[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) si[e] = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (p < P) {
                        const i32x4 ilo = __builtin_bit_cast(i32x4, plo[p]), ihi = __builtin_bit_cast(i32x4, phi[p]);   // (whole vectors:
#pragma unroll                                                                                                      // an element-wise
                        for (int e = 0; e < 4; ++e) { si[e] += ilo[e]; si[4 + e] += ihi[e]; }                            // bit_cast is not)
                    }
                }
                const float sca = psx[b];
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(pscb + i), c1 = *reinterpret_cast<const f32x4*>(pscb + i + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { u[e] = mmi_i8_dequant(si[e], sca, c0[e]); u[4 + e] = mmi_i8_dequant(si[4 + e], sca, c1[e]); }
            } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float on = p < P ? 1.f : 0.f;                // exact: x*1 = x, finite x*0 = 0
#pragma unroll
                for (int e = 0; e < 4; ++e) { u[e] += plo[p][e] * on; u[4 + e] += phi[p][e] * on; }
            }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[j][e] = mmi_round_bf16(mmi_round_bf16(u[e]) + f[j][e]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mmi_pack_bf16x2(f[j][2 * q], f[j][2 * q + 1]);
            *reinterpret_cast<u32x4*>(x + at) = v;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += f[j][e] * f[j][e];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += mmi_shfl_xor(ss, m);
    MMI_SHARED float red[16];
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (nth + 63) / 64; ++w) tot += red[w];
    const float rs = mmi_rsqrtf(eps + tot / (float)D);
    u32x4 ov[MMI_NORM_MAXP];
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float alo = mmi_bf16_to_f32((uint16_t)(al[j][q] & 0xffffu)), ahi = mmi_bf16_to_f32((uint16_t)(al[j][q] >> 16));
            o[q] = mmi_pack_bf16x2(f[j][2 * q] * (alo * rs), f[j][2 * q + 1] * (ahi * rs));
        }
        *reinterpret_cast<u32x4*>(y + mmi_xp_index(T, b, i, ksteps)) = o;
        ov[j] = o;
        am = fmaxf(am, mmi_absmax_bf16x8(o));
    }
    if (!yq) return;
    // ---- row-wise int8 copy of the normalised row
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) am = fmaxf(am, mmi_shfl_xor(am, m));
    MMI_SHARED float redm[16];
    if ((tid & 63) == 0) redm[tid >> 6] = am;
    __syncthreads();
    float sca = 0.f;
    for (int w = 0; w < (nth + 63) / 64; ++w) sca = fmaxf(sca, redm[w]);
    if (tid == 0) sx[b] = sca;
    const float scale = mmi_i8_scale(sca);
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        const long at = mmi_xp_index(T, b, i, ksteps);            // bf16 element index of the piece: fragment (mt * ksteps + ks), lane
        const long frag = at >> 9;
        const int ln = (int)((at >> 3) & 63);
        const long mt = frag / ksteps;
        const int ks = (int)(frag - mt * ksteps);
        *reinterpret_cast<u32x2*>(yq + ((mt * (ksteps >> 1) + (ks >> 1)) * 64 + ln) * 16 + (ks & 1) * 8) = mmi_quant_i8x8(ov[j], scale);
    }
}

// A packed bf16 tensor [B][D] -> its row-wise int8 copy Xq[mt][kp][lane][16] + the rows' absmax sx[b] (bitsandbytes'
// int8_vectorwise_quant), for the int8 linears whose input is written by MANY workgroups (the temporal attention output, the
// gated FFN tensor): one workgroup per row holds the row in registers (as the norm kernel does), reduces the absmax, converts.
__global__ __launch_bounds__(1024) void k_quant_rows_i8(const uint16_t* __restrict__ x, int B, int D, int T, int ksteps,
                                                        uint8_t* __restrict__ xq, float* __restrict__ sx) {
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    u32x4 v[MMI_NORM_MAXP];
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        v[j] = *reinterpret_cast<const u32x4*>(x + mmi_xp_index(T, b, i, ksteps));
        am = fmaxf(am, mmi_absmax_bf16x8(v[j]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) am = fmaxf(am, mmi_shfl_xor(am, m));
    MMI_SHARED float redm[16];
    if ((tid & 63) == 0) redm[tid >> 6] = am;
    __syncthreads();
    float sca = 0.f;
    for (int w = 0; w < (nth + 63) / 64; ++w) sca = fmaxf(sca, redm[w]);
    if (tid == 0) sx[b] = sca;
    const float scale = mmi_i8_scale(sca);
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        const long at = mmi_xp_index(T, b, i, ksteps);
        const long frag = at >> 9;
        const int ln = (int)((at >> 3) & 63);
        const long mt = frag / ksteps;
        const int ks = (int)(frag - mt * ksteps);
        *reinterpret_cast<u32x2*>(xq + ((mt * (ksteps >> 1) + (ks >> 1)) * 64 + ln) * 16 + (ks & 1) * 8) = mmi_quant_i8x8(v[j], scale);
    }
}

// The cross-attention block's norm (transformer.py:731-732, 779-783: `norm_cross` is always an nn.LayerNorm with weight and
// bias, eps 1e-5): same row handling as k_resid_rmsnorm - first fold the P pending split-K partials of the preceding out_proj
// into the residual stream - then y = ((x - mean) * rsqrt(var + eps) * w + b).to(bf16), statistics in fp32 over the D features.
__global__ __launch_bounds__(1024) void k_resid_layernorm(uint16_t* __restrict__ x, const float* __restrict__ partial, int P,
                                                          int B, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                                                          uint16_t* __restrict__ y, int D, int T, int ksteps, float eps) {
    const int b = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    float f[MMI_NORM_MAXP][8];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        const long at = mmi_xp_index(T, b, i, ksteps);
        u32x4 v = *reinterpret_cast<const u32x4*>(x + at);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f[j][2 * q] = mmi_bf16_to_f32((uint16_t)(v[q] & 0xffffu));
            f[j][2 * q + 1] = mmi_bf16_to_f32((uint16_t)(v[q] >> 16));
        }
        if (P > 0) {
            float u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = 0.f;
            for (int p = 0; p < P; ++p) {
                const float* pp = partial + ((long)p * B + b) * D + i;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(pp), hi = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { u[e] += lo[e]; u[4 + e] += hi[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[j][e] = mmi_round_bf16(mmi_round_bf16(u[e]) + f[j][e]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = mmi_pack_bf16x2(f[j][2 * q], f[j][2 * q + 1]);
            *reinterpret_cast<u32x4*>(x + at) = v;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += f[j][e];
    }
    MMI_SHARED float red[2][16];
    const int nw = (nth + 63) / 64;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s1 += mmi_shfl_xor(s1, m);
    if ((tid & 63) == 0) red[0][tid >> 6] = s1;
    __syncthreads();
    float tot = 0.f;
    for (int wv = 0; wv < nw; ++wv) tot += red[0][wv];
    const float mean = tot / (float)D;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dlt = f[j][e] - mean; s2 += dlt * dlt; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s2 += mmi_shfl_xor(s2, m);
    if ((tid & 63) == 0) red[1][tid >> 6] = s2;
    __syncthreads();
    float var = 0.f;
    for (int wv = 0; wv < nw; ++wv) var += red[1][wv];
    const float rstd = mmi_rsqrtf(var / (float)D + eps);
#pragma unroll
    for (int j = 0; j < MMI_NORM_MAXP; ++j) {
        const int i = (tid + j * nth) * 8;
        if (i >= D) break;
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + i), bv = *reinterpret_cast<const u32x4*>(bias + i);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float wl = mmi_bf16_to_f32((uint16_t)(wv[q] & 0xffffu)), wh = mmi_bf16_to_f32((uint16_t)(wv[q] >> 16));
            const float bl = mmi_bf16_to_f32((uint16_t)(bv[q] & 0xffffu)), bh = mmi_bf16_to_f32((uint16_t)(bv[q] >> 16));
            o[q] = mmi_pack_bf16x2((f[j][2 * q] - mean) * rstd * wl + bl, (f[j][2 * q + 1] - mean) * rstd * wh + bh);
        }
        *reinterpret_cast<u32x4*>(y + mmi_xp_index(T, b, i, ksteps)) = o;
    }
}

// rows of a row-major bf16 matrix [n][D] -> packed activation operand (columns col0 .. col0 + n of the source become the
// GEMM's batch rows 0 .. n): used once per stream to run the cross-attention source through the key / value projection
__global__ void k_pack_rows(const uint16_t* __restrict__ src, int n, int D, uint16_t* __restrict__ xp, int T, int ksteps) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= n * D) return;
    const int r = idx / D, k = idx - r * D;
    xp[mmi_xp_index(T, r, k, ksteps)] = src[idx];
}

// packed activation operand -> row-major rows [B][D] (the parity tap of the residual stream, mmi_lm_set_hidden_taps)
// debug trace (MMI_DEBUG_TRACE): order-independent 64-bit checksum of a buffer (sum of its 32-bit words, tail bytes included)
__global__ void k_checksum(const uint8_t* __restrict__ p, long nbytes, unsigned long long* __restrict__ out) {
    const long nwords = nbytes >> 2;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    unsigned long long acc = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x)
        acc += (unsigned long long)w[i] * (unsigned long long)(2 * (i % 1021) + 1);       // position-weighted: a swap of two words shows
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long i = nwords << 2; i < nbytes; ++i) acc += (unsigned long long)p[i] << (8 * (i & 3));
    atomicAdd(out, acc);
}

// parity tap: the int8 operand Xq[mt][kp][lane][16] (k_quant_rows_i8 / the norm kernel) -> row-major codes [B][D]
__global__ void k_unpack_q8(const uint8_t* __restrict__ xq, int B, int D, int8_t* __restrict__ dst, int T, int ksteps) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * D) return;
    const int r = idx / D, k = idx - r * D;
    const long at = mmi_xp_index(T, r, k, ksteps);
    const long frag = at >> 9;
    const int ln = (int)((at >> 3) & 63);
    const long mt = frag / ksteps;
    const int ks = (int)(frag - mt * ksteps);
    dst[idx] = (int8_t)xq[((mt * (ksteps >> 1) + (ks >> 1)) * 64 + ln) * 16 + (ks & 1) * 8 + (at & 7)];
}
__global__ void k_unpack_rows(const uint16_t* __restrict__ xp, int B, int D, uint16_t* __restrict__ dst, int T, int ksteps) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * D) return;
    const int r = idx / D, k = idx - r * D;
    dst[idx] = xp[mmi_xp_index(T, r, k, ksteps)];
}

// Cross-attention of the one new query per (model row, head) over the T_c projected condition positions (transformer.py:
// 544-552, 584: no mask, no rope, keys / values fixed for the stream).  kv: [rows][T_c][2 * H * Dh] bf16 (keys, then values, as
// the in_proj's rows dim.. produce them); q: [rows][H * Dh] bf16.  One wave per (row, head); lane (r, c) = position slot r x
// 16-byte chunk c of the head; online softmax per slot, slots merged at the end.
struct CrossAttnArgs {
    const uint16_t* q;
    const uint16_t* kv;
    uint16_t* out;          // packed (T, out_ksteps) operand of the cross out_proj, feature = h*Dh + d
    int B, H, Dh, Tc;
    int T, out_ksteps;
};

__global__ __launch_bounds__(64) void k_lm_cross_attn(CrossAttnArgs a) {
    const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
    const int lane = (int)threadIdx.x;
    const int Dh = a.Dh, HD = a.H * Dh, NC = Dh >> 3;          // NC in {1, 2, 4, 8, 16}
    const int RP = 64 / NC;                                     // position slots per pass
    const int r = lane / NC, c = lane - r * NC;
    const u32x4 qv = *reinterpret_cast<const u32x4*>(a.q + (long)b * HD + h * Dh + 8 * c);
    float qf[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { qf[2 * e] = __builtin_bit_cast(float, qv[e] << 16); qf[2 * e + 1] = __builtin_bit_cast(float, qv[e] & 0xffff0000u); }
    const float scale = 1.0f / sqrtf((float)Dh);
    float m_run = -INFINITY, l_run = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int t0 = 0; t0 < a.Tc; t0 += RP) {
        const int t = t0 + r;
        const bool live = t < a.Tc;
        const uint16_t* row = a.kv + ((long)b * a.Tc + (live ? t : a.Tc - 1)) * 2 * HD + h * Dh + 8 * c;
        const u32x4 kv = *reinterpret_cast<const u32x4*>(row), vv = *reinterpret_cast<const u32x4*>(row + HD);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d += qf[2 * e] * __builtin_bit_cast(float, kv[e] << 16);
            d += qf[2 * e + 1] * __builtin_bit_cast(float, kv[e] & 0xffff0000u);
        }
        for (int m = 1; m < NC; m <<= 1) d += mmi_shfl_xor(d, m);
        const float sc = live ? d * scale : -INFINITY;
        const float m_new = fmaxf(m_run, sc);
        const float resc = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
        const float p = live ? expf(sc - m_new) : 0.f;
        l_run = l_run * resc + p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[2 * e] = acc[2 * e] * resc + p * __builtin_bit_cast(float, vv[e] << 16);
            acc[2 * e + 1] = acc[2 * e + 1] * resc + p * __builtin_bit_cast(float, vv[e] & 0xffff0000u);
        }
        m_run = m_new;
    }
    // merge the RP slots (lanes that share c): butterflies over the slot bits
    for (int m = NC; m < 64; m <<= 1) {
        const float mo = mmi_shfl_xor(m_run, m), lo = mmi_shfl_xor(l_run, m);
        const float mn = fmaxf(m_run, mo);
        const float sa = m_run == -INFINITY ? 0.f : expf(m_run - mn), sb = mo == -INFINITY ? 0.f : expf(mo - mn);
        l_run = l_run * sa + lo * sb;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * sa + mmi_shfl_xor(acc[e], m) * sb;
        m_run = mn;
    }
    if (r == 0) {
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = mmi_pack_bf16x2(acc[2 * e] / l_run, acc[2 * e + 1] / l_run);
        *reinterpret_cast<u32x4*>(a.out + mmi_xp_index(a.T, b, h * Dh + 8 * c, a.out_ksteps)) = ov;
    }
}

// ------------------------------------------------------------------------------------------------
// input embedding sum (lm.py:388-397): ((emb0[t1] + emb1[t2]) + ...) + text_emb[t0], each add rounded to bf16
// ------------------------------------------------------------------------------------------------
// cond: the fuser's summed condition [rows][D] (lm.py:399-400: input_ = input_ + sum_condition), or null
__global__ void k_lm_embed(const int* __restrict__ tokens, int n_codebooks, const uint16_t* __restrict__ emb,
                           int card1, const uint16_t* __restrict__ text_emb, uint16_t* __restrict__ x, int D, int T,
                           int ksteps, const uint16_t* __restrict__ cond) {
    const int b = blockIdx.y;
    const int d = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (d >= D) return;
    const int* tk = tokens + (long)b * n_codebooks;
    float acc = 0.f;
    for (int c = 1; c < n_codebooks; ++c) {
        int t = tk[c];
        float v = 0.f;
        if (t != -1) v = mmi_bf16_to_f32(emb[((long)(c - 1) * card1 + (t < 0 ? 0 : t)) * D + d]);
        acc = c == 1 ? v : mmi_round_bf16(acc + v);
    }
    int t0 = tk[0];
    float tv = 0.f;
    if (t0 != -1) tv = mmi_bf16_to_f32(text_emb[(long)(t0 < 0 ? 0 : t0) * D + d]);
    acc = n_codebooks > 1 ? mmi_round_bf16(acc + tv) : tv;
    if (cond) acc = acc + mmi_bf16_to_f32(cond[(long)b * D + d]);
    x[mmi_xp_index(T, b, d, ksteps)] = mmi_f32_to_bf16(acc);
}

// ------------------------------------------------------------------------------------------------
// temporal attention: RoPE + ring-KV write, split decode attention over the VALID part of the ring, combine
// ------------------------------------------------------------------------------------------------
struct LmAttnArgs {
    uint16_t* qrot;        // [B][H][Dh] roped queries (written by in_proj's epilogue)
    uint16_t* kc;          // [B][H][cap][Dh]
    uint16_t* vc;
    const long* offsets;   // [B]
    float* opart;          // [B][H][NS][Dh]
    float* ml;             // [B][H][NS][2]
    uint16_t* out;         // packed (T, out_ksteps) activations of out_proj, feature = h*Dh + d
    int B, H, Dh, cap, context, NS;
    int T, out_ksteps;
    float max_period;
    unsigned* done;        // k_lm_attn_wave with the ring split over NS workgroups: arrival counter per (session, head), 0 between launches
                           // (the last workgroup to arrive merges); null = leave the partials to k_lm_attn_combine
    int solo_rows;         // ... rings of at most this many rows are walked by workgroup y = 0 alone (the others exit at once); -1 = never
};

#define MMI_ATTN_CHUNK 256
// Decode attention of the one new query per (session, head) over the VALID part of the ring only (the reference reads
// all `cap` slots through a mask, transformer.py:574-585).  grid (B*H, NS); 256 threads.  Block y walks the 256-slot
// chunks y, y+NS, ... of the ring with an online softmax; each 16-byte load covers 8 dims of one key row, DH/8 lanes share
// a row, so one wave instruction reads 64/(DH/8) consecutive rows = 1 KiB contiguous.  NS == 1 (enough (b,h) pairs to
// fill the chip): the normalised output goes straight to out_proj's packed input; otherwise partials for the combine.
// KV8: the ring holds e4m3 bytes (fp8 KV cache): a 16-byte load then covers 16 dims, DH/16 lanes share a row and one wave
// instruction reads twice as many rows; the values are widened exactly (v_cvt_pk_f32_fp8) and everything else is unchanged.
template <int DH, bool KV8 = false>
__global__ __launch_bounds__(256) void k_lm_attn_split(LmAttnArgs a) {
    constexpr int EPL = KV8 ? 16 : 8;  // elements per lane (16 bytes)
    constexpr int ES = KV8 ? 1 : 2;    // bytes per element
    constexpr int LPR = DH / EPL;      // lanes per row
    constexpr int RPW = 64 / LPR;      // rows per wave instruction
    constexpr int CH = MMI_ATTN_CHUNK;
    // (a mirrored session order - every second group of 8 sessions walked backwards, for load balance across CUs under a linear
    // stagger of ring depths - measured neutral, round 2: not built in)
    const int bh = blockIdx.x;
    const int b = bh / a.H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long off = a.offsets[b];
    const long end_new = off + 1;
    const int end_index = (int)(off % a.cap);
    const int L = (int)(end_new < (long)a.cap ? end_new : (long)a.cap);
    MMI_SHARED float sc[CH];
    MMI_SHARED float wred[8];
    MMI_SHARED float ored[4 * DH];
    const int seg = lane % LPR, rsub = lane / LPR;
    float qv[EPL];
#pragma unroll
    for (int v = 0; v < EPL / 8; ++v) {
        u32x4 qq = *reinterpret_cast<const u32x4*>(a.qrot + (long)bh * DH + seg * EPL + v * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qv[v * 8 + 2 * q] = mmi_bf16_to_f32((uint16_t)(qq[q] & 0xffffu));
            qv[v * 8 + 2 * q + 1] = mmi_bf16_to_f32((uint16_t)(qq[q] >> 16));
        }
    }
    // 16 bytes of a ring row -> EPL fp32 values
    auto widen = [](const u32x4& r, float* f) {
        if constexpr (KV8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mmi_fp8x4_to_f32(r[q], f + 4 * q);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f[2 * q] = mmi_bf16_to_f32((uint16_t)(r[q] & 0xffffu));
                f[2 * q + 1] = mmi_bf16_to_f32((uint16_t)(r[q] >> 16));
            }
        }
    };
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(a.kc) + (long)bh * a.cap * DH * ES;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(a.vc) + (long)bh * a.cap * DH * ES;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int PER_WAVE = CH / 4;
    float m_run = -INFINITY, l_run = 0.f;
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int c0 = (int)blockIdx.y * CH; c0 < L; c0 += (int)gridDim.y * CH) {     // block-uniform trip count
        // ---- scores of this chunk: all of the wave's key rows are requested up front (unconditional loads from a clamped
        // slot; a load under `if (valid)` would be serialised behind s_waitcnt vmcnt(0)), then reduced
        constexpr int NIT = PER_WAVE / RPW;
        constexpr int NB = NIT < 8 ? NIT : 8;             // rows in flight per lane: 8 x 16 bytes keeps the kernel at <= 96 VGPRs
#pragma unroll 1
        for (int h0 = 0; h0 < NIT; h0 += NB) {
            u32x4 kk[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int slot = min(c0 + wave * PER_WAVE + (h0 + i) * RPW + rsub, L - 1);
                kk[i] = *reinterpret_cast<const u32x4*>(kbase + ((long)slot * DH + seg * EPL) * ES);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int rl = wave * PER_WAVE + (h0 + i) * RPW + rsub;
                const int slot = c0 + rl;
                bool valid = slot < L;
                if (valid) {   // absolute position of the slot (transformer.py:258-286) and the causal/context mask (:574-582)
                    int delta = slot - end_index;
                    long pos = delta <= 0 ? off + delta : off + delta - a.cap;
                    long dq = off - pos;
                    valid = pos >= 0 && dq >= 0 && dq < a.context;
                }
                float dot = 0.f;
                float kf[EPL];
                widen(kk[i], kf);
#pragma unroll
                for (int e = 0; e < EPL; ++e) dot += qv[e] * kf[e];
#pragma unroll
                for (int m = LPR / 2; m >= 1; m >>= 1) dot += mmi_shfl_xor(dot, m);
                if (seg == 0) sc[rl] = valid ? dot * scale : -INFINITY;
            }
        }
        __syncthreads();
        // ---- online softmax update
        const float sv = sc[tid];
        float mx = sv;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, mmi_shfl_xor(mx, m));
        if (lane == 0) wred[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
        const float m_new = fmaxf(m_run, mx);
        const float p = (m_new == -INFINITY) ? 0.f : expf(sv - m_new);
        float sum = p;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sum += mmi_shfl_xor(sum, m);
        if (lane == 0) wred[4 + wave] = sum;
        sc[tid] = p;
        __syncthreads();
        sum = (wred[4] + wred[5]) + (wred[6] + wred[7]);
        const float resc = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        l_run = l_run * resc + sum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] *= resc;
        // ---- P.V (value rows requested up front as well; rows past L carry p = 0 and finite ring contents)
#pragma unroll 1
        for (int h0 = 0; h0 < NIT; h0 += NB) {
            u32x4 vv[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int slot = min(c0 + wave * PER_WAVE + (h0 + i) * RPW + rsub, L - 1);
                vv[i] = *reinterpret_cast<const u32x4*>(vbase + ((long)slot * DH + seg * EPL) * ES);
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int rl = wave * PER_WAVE + (h0 + i) * RPW + rsub;
                const float pr = (c0 + rl < L) ? sc[rl] : 0.f;
                float vf[EPL];
                widen(vv[i], vf);
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] += pr * vf[e];
            }
        }
        __syncthreads();     // sc / wred are rewritten by the next chunk
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
        for (int m = 32; m >= LPR; m >>= 1) acc[e] += mmi_shfl_xor(acc[e], m);
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) ored[wave * DH + seg * EPL + e] = acc[e];
    }
    __syncthreads();
    if (tid < DH) {
        const float o = (ored[tid] + ored[DH + tid]) + (ored[2 * DH + tid] + ored[3 * DH + tid]);
        if (gridDim.y == 1) {
            a.out[mmi_xp_index(a.T, b, (bh % a.H) * DH + tid, a.out_ksteps)] = mmi_f32_to_bf16(o / l_run);
        } else {
            a.opart[((long)bh * gridDim.y + blockIdx.y) * DH + tid] = o;
        }
    }
    if (gridDim.y > 1 && tid == 0) {
        float* mlp = a.ml + ((long)bh * gridDim.y + blockIdx.y) * 2;
        mlp[0] = m_run;
        mlp[1] = l_run;
    }
}

// The same attention with ONE online softmax per lane group and no workgroup barrier inside the loop (round 4).  k_lm_attn_split
// walks a 256-row chunk in four dependent memory round trips (keys x 2, softmax through LDS, values x 2) with three barriers
// in between, and all of a CU's co-resident workgroups do so in lockstep: at the benchmark's ring depths (150 ... 550 rows per
// head) the kernel is bound by those round trips, not by HBM (23 us for 78 MB).  Here the ring's rows are dealt out in
// row groups (one wave instruction = RPW consecutive rows = 1 KiB) round-robin over ALL waves that serve the (session, head)
// pair - 4 per workgroup x gridDim.y workgroups - and every wave keeps, per row slot (the lanes that share `rsub`), its own
// running (max, sum, accumulator): a round requests NB key groups AND their NB value groups at once (the values do not depend
// on the scores), the next round's 2 NB loads are issued before this round's arithmetic, and nothing but registers and
// cross-lane shuffles is touched until the slots and the waves are merged once at the end.
template <int DH, bool KV8 = false>
__global__ __launch_bounds__(256, KV8 ? 2 : 4) void k_lm_attn_wave(LmAttnArgs a) {      // bf16 ring: <= 128 VGPRs, 4 workgroups per CU
    constexpr int EPL = KV8 ? 16 : 8;  // elements per lane (16 bytes)
    constexpr int ES = KV8 ? 1 : 2;    // bytes per element
    constexpr int LPR = DH / EPL;      // lanes per row
    constexpr int RPW = 64 / LPR;      // rows per wave instruction
    constexpr int NB = 4;              // row groups per round (x 2 buffers x (key + value) = 16 loads of 16 bytes in flight per lane)
    const int bh = blockIdx.x;
    const int b = bh / a.H;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long off = a.offsets[b];
    const int end_index = (int)(off % a.cap);
    const long end_new = off + 1;
    const int L = (int)(end_new < (long)a.cap ? end_new : (long)a.cap);
    const int seg = lane % LPR, rsub = lane / LPR;
    float qv[EPL];
    u32x4 qp = {0u, 0u, 0u, 0u};      // bf16 ring: the lane's 8 query elements stay packed (v_dot2c_f32_bf16 takes both operands packed)
#pragma unroll
    for (int v = 0; v < EPL / 8; ++v) {
        u32x4 qq = *reinterpret_cast<const u32x4*>(a.qrot + (long)bh * DH + seg * EPL + v * 8);
        if (v == 0) qp = qq;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qv[v * 8 + 2 * q] = mmi_bf16_to_f32((uint16_t)(qq[q] & 0xffffu));
            qv[v * 8 + 2 * q + 1] = mmi_bf16_to_f32((uint16_t)(qq[q] >> 16));
        }
    }
    auto widen = [](const u32x4& r, float* f) {
        if constexpr (KV8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mmi_fp8x4_to_f32(r[q], f + 4 * q);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f[2 * q] = mmi_bf16_to_f32((uint16_t)(r[q] & 0xffffu));
                f[2 * q + 1] = mmi_bf16_to_f32((uint16_t)(r[q] >> 16));
            }
        }
    };
    // wave-uniform base (scalar registers) + a 32-bit lane offset: the loads take the saddr form, no 64-bit address per lane
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(a.kc) + (long)bh * a.cap * DH * ES;
    const uint8_t* vbase = reinterpret_cast<const uint8_t*>(a.vc) + (long)bh * a.cap * DH * ES;
    const unsigned loff = (unsigned)(seg * EPL * ES);
    const float scale = 1.0f / sqrtf((float)DH);
    // One workgroup per (session, head) (gridDim.y == 1: which wave walks which row group does not depend on the ring depth): the
    // FIRST round's keys and values are requested before the session's offset - a cold scalar load of its own, ~1 us into a
    // ~40 us launch - has come back: row groups wave, wave + 4, ... of the ring's first slots, any of which lies inside the
    // allocation; rows past the valid part are masked where they are used (slot < L), as the clamped re-reads always were
    u32x4 kA[NB], vA[NB], kB[NB], vB[NB];
    const bool early = gridDim.y == 1;
    if (early) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int slot_ = min((wave + i * 4) * RPW + rsub, a.cap - 1);
            const unsigned o_ = (unsigned)slot_ * (unsigned)(DH * ES) + loff;
            kA[i] = *reinterpret_cast<const u32x4*>(kbase + o_);
            vA[i] = *reinterpret_cast<const u32x4*>(vbase + o_);
        }
    }
    // context >= capacity (the LM's ring: both 3000): every written slot is inside the window - slot s < L holds a position p with
    // 0 <= offset - p < cap - so the mask (transformer.py:574-582) reduces to `slot < L` and the per-row position arithmetic is
    // skipped (wave-uniform branch); a ring longer than its attention window keeps it
    const bool windowed = a.context < a.cap;
    // Fewer (session, head) pairs than it takes to fill the chip (one real-time session: 32): the ring is shared out over gridDim.y
    // workgroups - but only where it is long enough to be worth the merge: a short ring is walked by workgroup y = 0 alone, which
    // then writes the final output itself (no partials, no fence, and no combine launch either way: see the end of the kernel)
    const bool solo = (int)gridDim.y > 1 && L <= a.solo_rows;
    if (solo && blockIdx.y != 0) return;
    const int nsplit = solo ? 1 : (int)gridDim.y;
    const int total = nsplit * 4, wg = (solo ? 0 : (int)blockIdx.y * 4) + wave;  // waves serving this pair, and which one this is
    const int ngroups = (L + RPW - 1) / RPW;
    const int nmine = wg < ngroups ? (ngroups - wg + total - 1) / total : 0;       // row groups wg, wg + total, ...
    const int nrounds = (nmine + NB - 1) / NB;
    float m_run = -INFINITY, l_run = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    // every load is unconditional from a clamped (valid) slot - a load under a branch would be serialised behind
    // s_waitcnt vmcnt(0) - and rows past the end carry probability 0
#define MMI_AW_LOAD(KK, VV, j0)                                                                  \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                            \
        const int slot_ = min((wg + ((j0) + i) * total) * RPW + rsub, L - 1);                   \
        const unsigned o_ = (unsigned)slot_ * (unsigned)(DH * ES) + loff;                       \
        KK[i] = *reinterpret_cast<const u32x4*>(kbase + o_);                                    \
        VV[i] = *reinterpret_cast<const u32x4*>(vbase + o_);                                    \
    }
#define MMI_AW_USE(KK, VV, j0)                                                                   \
    {                                                                                            \
        float s_[NB];                                                                            \
        float mx_ = -INFINITY;                                                                   \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                        \
            const int j_ = (j0) + i;                                                             \
            const int slot_ = (wg + j_ * total) * RPW + rsub;                                    \
            bool valid_ = j_ < nmine && slot_ < L;                                               \
            if (windowed && valid_) {   /* absolute position of the slot (transformer.py:258-286), causal / context mask (:574-582) */ \
                const int delta_ = slot_ - end_index;                                            \
                const long pos_ = delta_ <= 0 ? off + delta_ : off + delta_ - a.cap;             \
                const long dq_ = off - pos_;                                                     \
                valid_ = pos_ >= 0 && dq_ >= 0 && dq_ < a.context;                               \
            }                                                                                    \
            float dot_ = 0.f;                                                                    \
            if constexpr (KV8) {                                                                 \
                float kf_[EPL];                                                                  \
                widen(KK[i], kf_);                                                               \
                _Pragma("unroll") for (int e = 0; e < EPL; ++e) dot_ = mmi_fma(qv[e], kf_[e], dot_); \
            } else {                    /* 4 packed-pair dot products instead of 8 unpacks + 8 FMAs */ \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) dot_ = mmi_dot2_bf16(KK[i][q], qp[q], dot_); \
            }                                                                                    \
            dot_ = mmi_group_sum<LPR>(dot_);                /* DPP row operations, no trip through the LDS crossbar */ \
            s_[i] = valid_ ? dot_ * scale : -INFINITY;                                           \
            mx_ = fmaxf(mx_, s_[i]);                                                             \
        }                                                                                        \
        const float m_new_ = fmaxf(m_run, mx_);                                                  \
        if (m_new_ != -INFINITY) {                                                               \
            const float resc_ = m_run == -INFINITY ? 0.f : expf(m_run - m_new_);                 \
            l_run *= resc_;                                                                      \
            _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] *= resc_;                    \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                    \
                const float p_ = expf(s_[i] - m_new_);        /* exp(-inf) = 0 for masked rows */ \
                l_run += p_;                                                                     \
                if constexpr (KV8) {                                                             \
                    float vf_[EPL];                                                              \
                    widen(VV[i], vf_);                                                           \
                    _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] = mmi_fma(p_, vf_[e], acc[e]); \
                } else {                /* packed FMAs: two accumulator elements per instruction */ \
                    const f32x2 pp_ = {p_, p_};                                                  \
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                             \
                        const f32x2 v2_ = {__builtin_bit_cast(float, VV[i][q] << 16), __builtin_bit_cast(float, VV[i][q] & 0xffff0000u)}; \
                        const f32x2 a2_ = mmi_pk_fma(pp_, v2_, f32x2{acc[2 * q], acc[2 * q + 1]}); \
                        acc[2 * q] = a2_[0];                                                     \
                        acc[2 * q + 1] = a2_[1];                                                 \
                    }                                                                            \
                }                                                                                \
            }                                                                                    \
            m_run = m_new_;                                                                      \
        }                                                                                        \
    }
    if (nrounds > 0) {
        if (!early) { MMI_AW_LOAD(kA, vA, 0) }
        for (int r = 0; r < nrounds; r += 2) {
            MMI_AW_LOAD(kB, vB, (r + 1) * NB)          // clamped past the end: re-reads the last row (cache hit), masked on use
            MMI_AW_USE(kA, vA, r * NB)
            MMI_AW_LOAD(kA, vA, (r + 2) * NB)
            MMI_AW_USE(kB, vB, (r + 1) * NB)
        }
    }
#undef MMI_AW_LOAD
#undef MMI_AW_USE
    // ---- merge the RPW row slots of the wave (lanes that share `seg`): butterflies over the slot bits
#pragma unroll
    for (int mk = LPR; mk < 64; mk <<= 1) {
        const float mo = mmi_shfl_xor(m_run, mk), lo = mmi_shfl_xor(l_run, mk);
        const float mn = fmaxf(m_run, mo);
        const float sa = m_run == -INFINITY ? 0.f : expf(m_run - mn), sb = mo == -INFINITY ? 0.f : expf(mo - mn);
        l_run = l_run * sa + lo * sb;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * sa + mmi_shfl_xor(acc[e], mk) * sb;
        m_run = mn;
    }
    // ---- merge the 4 waves through LDS (the only barrier of the kernel)
    MMI_SHARED float wm[4], wl[4];
    MMI_SHARED float wacc[4 * DH];
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) wacc[wave * DH + seg * EPL + e] = acc[e];
        if (seg == 0) { wm[wave] = m_run; wl[wave] = l_run; }
    }
    __syncthreads();
    if (tid < DH) {
        const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sw = wm[w] == -INFINITY ? 0.f : expf(wm[w] - M);
            num += sw * wacc[w * DH + tid];
            den += sw * wl[w];
        }
        if (nsplit == 1) {
            a.out[mmi_xp_index(a.T, b, (bh % a.H) * DH + tid, a.out_ksteps)] = mmi_f32_to_bf16(num / den);
        } else {
            a.opart[((long)bh * gridDim.y + blockIdx.y) * DH + tid] = num;
            if (tid == 0) {
                float* mlp = a.ml + ((long)bh * gridDim.y + blockIdx.y) * 2;
                mlp[0] = M;
                mlp[1] = den;
            }
        }
    }
    if (nsplit == 1 || a.done == nullptr) return;        // (no counter: k_lm_attn_combine follows in a launch of its own)
    // ---- the workgroup that arrives LAST merges the partials: out = sum_c e^{m_c - M} O_c / sum_c e^{m_c - M} l_c.  The safety net
    // of the engine's shallow-ring program (lm_engine.hip attn_variant), which has no merge launch: the release + arrival + acquire
    // chain costs about 8 us per layer more than that launch would, so rings the host KNOWS may be deep take the other program
    MMI_SHARED unsigned s_last;
    __syncthreads();                                     // the partials of this workgroup are stored
    if (tid == 0) {
        const unsigned seen = mmi_arrive_release(a.done + bh);
        s_last = seen == (unsigned)gridDim.y - 1u ? 1u : 0u;
        if (s_last) {
            mmi_store_relaxed_agent(a.done + bh, 0u);    // everyone has arrived: the counter is ready for the next launch
            mmi_acquire_agent();
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < DH) {
        const int NSg = (int)gridDim.y;
        const float* ml = a.ml + (long)bh * NSg * 2;
        float M = -INFINITY;
        for (int c = 0; c < NSg; ++c) M = fmaxf(M, ml[2 * c]);
        float num = 0.f, den = 0.f;
        for (int c = 0; c < NSg; ++c) {
            const float m = ml[2 * c];
            if (m == -INFINITY) continue;
            const float w = expf(m - M);
            num += w * a.opart[((long)bh * NSg + c) * DH + tid];
            den += w * ml[2 * c + 1];
        }
        a.out[mmi_xp_index(a.T, b, (bh % a.H) * DH + tid, a.out_ksteps)] = mmi_f32_to_bf16(num / den);
    }
}

// merge the chunk partials: out = sum_c e^{m_c-M} O_c / sum_c e^{m_c-M} l_c  -> bf16 [B][H*Dh]
__global__ void k_lm_attn_combine(LmAttnArgs a) {
    const int bh = blockIdx.x;
    const int Dh = a.Dh;
    const float* ml = a.ml + (long)bh * a.NS * 2;
    float M = -INFINITY;
    for (int c = 0; c < a.NS; ++c) M = fmaxf(M, ml[2 * c]);
    for (int d = threadIdx.x; d < Dh; d += blockDim.x) {
        float num = 0.f, den = 0.f;
        for (int c = 0; c < a.NS; ++c) {
            float m = ml[2 * c];
            if (m == -INFINITY) continue;
            float w = expf(m - M);
            num += w * a.opart[((long)bh * a.NS + c) * Dh + d];
            den += w * ml[2 * c + 1];
        }
        a.out[mmi_xp_index(a.T, bh / a.H, (bh % a.H) * Dh + d, a.out_ksteps)] = mmi_f32_to_bf16(num / den);
    }
}

// ------------------------------------------------------------------------------------------------
// depformer attention: step `k` of the 8 per frame; keys 0..k of a cache that is rebuilt every frame
// (no RoPE: depformer_pos_emb "none"; KV ignores the exec mask, transformer.py:251-253,478)
// one wave per (b, head); Dh <= 64
// ------------------------------------------------------------------------------------------------
struct DepAttnArgs {
    const uint16_t* qkv;   // [B][3*H*Dh]
    uint16_t* kc;          // [B][H][steps][Dh]
    uint16_t* vc;
    uint16_t* out;         // packed (T, out_ksteps), feature = h*Dh + lane
    int B, H, Dh, steps, k;
    int T, out_ksteps;
};

__global__ __launch_bounds__(64) void k_dep_attn(DepAttnArgs a) {
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int lane = threadIdx.x, Dh = a.Dh, HD = a.H * Dh;
    const bool on = lane < Dh;
    const uint16_t* row = a.qkv + (long)b * 3 * HD;
    uint16_t* kcb = a.kc + ((long)b * a.H + h) * a.steps * Dh;
    uint16_t* vcb = a.vc + ((long)b * a.H + h) * a.steps * Dh;
    float q = 0.f;
    uint16_t kn = 0, vn = 0;
    if (on) {
        q = mmi_bf16_to_f32(row[h * Dh + lane]);
        kn = row[HD + h * Dh + lane];
        vn = row[2 * HD + h * Dh + lane];
        kcb[(long)a.k * Dh + lane] = kn;
        vcb[(long)a.k * Dh + lane] = vn;
    }
    const float scale = 1.0f / sqrtf((float)Dh);
    // rows 0..k-1 of this frame's cache, all requested up front (unconditional loads from a clamped row); steps <= 16
    float kr[16], vr[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int jc = j < a.k ? j : (a.k > 0 ? a.k - 1 : 0);
        const int lc = on ? lane : 0;
        kr[j] = mmi_bf16_to_f32(kcb[(long)jc * Dh + lc]);
        vr[j] = mmi_bf16_to_f32(vcb[(long)jc * Dh + lc]);
    }
    float sc[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j <= a.k) {
            float kv = 0.f;
            if (on) kv = (j == a.k) ? mmi_bf16_to_f32(kn) : kr[j];
            float d = q * kv;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) d += mmi_shfl_xor(d, m);
            sc[j] = d * scale;
            mx = fmaxf(mx, sc[j]);
        }
    }
    float den = 0.f, o = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j <= a.k) {
            float p = expf(sc[j] - mx);
            den += p;
            float vv = 0.f;
            if (on) vv = (j == a.k) ? mmi_bf16_to_f32(vn) : vr[j];
            o += p * vv;
        }
    }
    if (on) a.out[mmi_xp_index(a.T, b, h * Dh + lane, a.out_ksteps)] = mmi_f32_to_bf16(o / den);
}

// The same attention for Dh a multiple of 8 (<= 64) and <= 8 positions per frame - Moshi's depth transformer - with far fewer
// dependent instructions: one wave per (session, head), lane (j, c) = position j x 16-byte chunk c of the head: it loads its
// chunk of q, of key row j and of value row j once (the newest row straight from the in_proj output, which the lanes of
// position k also copy into the frame's cache), reduces the score over the 8 chunk lanes, the softmax and P.V over the 8
// position groups (3 butterfly steps each), and the lanes of position 0 store 8 output features as one 16-byte vector.
// One (session, head) by the calling wave, in two parts so that a wave serving several pairs (k_dep_attn_out_proj) has all their
// loads in flight before the first dependent instruction.  mmi_dep_attn8_load: the lane's 16-byte chunk of q, of key row j and of
// value row j (write_cache: the lanes of position k also copy the newest key / value row into the frame's cache);
// mmi_dep_attn8_reduce: on return the lanes of position 0 (j == 0, c < Dh / 8) hold output features 8 c .. 8 c + 7 of the head as
// packed bf16.
struct DepAttnOperands { u32x4 q, k, v; };
__device__ __forceinline__ DepAttnOperands mmi_dep_attn8_load(const DepAttnArgs& a, int b, int h, int lane, bool write_cache) {
    const int Dh = a.Dh, HD = a.H * Dh, NC = Dh >> 3;
    const int j = lane >> 3, c = lane & 7;
    const int jc = j < a.k ? j : a.k, cc = c < NC ? c : 0;          // clamped: every lane loads from a valid address
    const uint16_t* row = a.qkv + (long)b * 3 * HD + h * Dh + 8 * cc;
    uint16_t* kcb = a.kc + ((long)b * a.H + h) * a.steps * Dh + 8 * cc;
    uint16_t* vcb = a.vc + ((long)b * a.H + h) * a.steps * Dh + 8 * cc;
    const bool newest = jc == a.k;
    DepAttnOperands o;
    o.q = *reinterpret_cast<const u32x4*>(row);
    o.k = *reinterpret_cast<const u32x4*>(newest ? row + HD : kcb + (long)jc * Dh);
    o.v = *reinterpret_cast<const u32x4*>(newest ? row + 2 * HD : vcb + (long)jc * Dh);
    if (write_cache && j == a.k && c < NC) {                        // this frame's cache, position k (transformer.py:243-253)
        *reinterpret_cast<u32x4*>(kcb + (long)a.k * Dh) = o.k;
        *reinterpret_cast<u32x4*>(vcb + (long)a.k * Dh) = o.v;
    }
    return o;
}
__device__ __forceinline__ u32x4 mmi_dep_attn8_reduce(const DepAttnArgs& a, const DepAttnOperands& in, int lane) {
    const int Dh = a.Dh, NC = Dh >> 3;
    const int j = lane >> 3, c = lane & 7;
    const u32x4 qv = in.q, kv = in.k, vv = in.v;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        d += __builtin_bit_cast(float, qv[e] << 16) * __builtin_bit_cast(float, kv[e] << 16);
        d += __builtin_bit_cast(float, qv[e] & 0xffff0000u) * __builtin_bit_cast(float, kv[e] & 0xffff0000u);
    }
    if (c >= NC) d = 0.f;
    d += mmi_shfl_xor(d, 1); d += mmi_shfl_xor(d, 2); d += mmi_shfl_xor(d, 4);
    const bool live = j <= a.k;
    const float sc = live ? d * (1.0f / sqrtf((float)Dh)) : -INFINITY;
    float mx = sc;
    mx = fmaxf(mx, mmi_shfl_xor(mx, 8)); mx = fmaxf(mx, mmi_shfl_xor(mx, 16)); mx = fmaxf(mx, mmi_shfl_xor(mx, 32));
    const float p = live ? expf(sc - mx) : 0.f;
    float den = p;
    den += mmi_shfl_xor(den, 8); den += mmi_shfl_xor(den, 16); den += mmi_shfl_xor(den, 32);
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[2 * e] = p * __builtin_bit_cast(float, vv[e] << 16);
        o[2 * e + 1] = p * __builtin_bit_cast(float, vv[e] & 0xffff0000u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += mmi_shfl_xor(o[e], 8); o[e] += mmi_shfl_xor(o[e], 16); o[e] += mmi_shfl_xor(o[e], 32);
    }
    u32x4 ov;
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = mmi_pack_bf16x2(o[2 * e] / den, o[2 * e + 1] / den);
    return ov;
}
__device__ __forceinline__ u32x4 mmi_dep_attn8_wave(const DepAttnArgs& a, int b, int h, int lane, bool write_cache) {
    return mmi_dep_attn8_reduce(a, mmi_dep_attn8_load(a, b, h, lane, write_cache), lane);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void k_dep_attn8(DepAttnArgs a) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int bh = (int)blockIdx.x * NW + wave;
    if (bh >= a.B * a.H) return;
    const int b = bh / a.H, h = bh - b * a.H;
    const u32x4 ov = mmi_dep_attn8_wave(a, b, h, lane, true);
    const int j = lane >> 3, c = lane & 7;
    if (j == 0 && c < (a.Dh >> 3)) *reinterpret_cast<u32x4*>(a.out + mmi_xp_index(a.T, b, h * a.Dh + 8 * c, a.out_ksteps)) = ov;
}

// The depth transformer's attention INSIDE its out_proj (one session - the engine instantiates BMAX = 1 - at the 16-row tile,
// micro-steps >= 1): the GEMM's K axis is the attention output [head][Dh], a wave's K-slice is whole heads, so every workgroup's
// wave computes the (session, head) pairs of its own slice itself - k_dep_attn8's arithmetic, straight after requesting its weight
// fragments - and hands the 8-feature pieces to the lanes that hold them in the MFMA operand through LDS: two pairs per wave, a
// few hundred bytes from L2.  What it removes is a 4.5 us launch on the depth transformer's dependent chain, 42 times per step
// (the GEMM grows by 1.6 us).  The workgroups all compute the same thing; workgroup 0 also writes the newest key / value rows to
// the frame's cache.  Bit-identical to the two launches (the same per-pair functions, the same operand values).
template <int WAVES, int HPW, int BMAX>
__global__ __launch_bounds__(WAVES * 64) void k_dep_attn_out_proj(GemmArgs a, DepAttnArgs d) {
    constexpr int TN = 16, R = 4, KPWMAX = 4;
    typedef float acc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int os = a.osplit > 1 ? a.osplit : 1;                     // octet sharing as in k_gemm_xp
    const int bx = (int)blockIdx.x / os, part = (int)blockIdx.x - bx * os;
    const int nt0 = bx;
    const int g_lo = part * ((TN / 8) / os), g_hi = os > 1 ? g_lo + (TN / 8) / os : TN / 8;
    const int ro = (lane >> 3) & (TN / 8 - 1);
    const int wlane = (ro >= g_lo && ro < g_hi) ? lane : ((lane & ~((TN / 8 - 1) << 3)) | (g_lo << 3));
    const u32x4 pre = mmi_gemm_prefetch_addend<TN, 1, 1>(a, nt0);
    const int kper = (a.KSTEPS + WAVES - 1) / WAVES;                // <= KPWMAX, kper * 32 a multiple of Dh (launcher)
    const int ks0 = min(a.KSTEPS, wave * kper);
    const int nks = min(a.KSTEPS, ks0 + kper) - ks0;
    const int last = nks > 0 ? nks - 1 : 0;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4* wp = a.wp + ((long)min(nt0, a.NT - 1) * a.KSTEPS + min(ks0, a.KSTEPS - 1)) * 64 + wlane;
    u32x4 wv[KPWMAX];
#pragma unroll
    for (int u = 0; u < KPWMAX; ++u) wv[u] = mmi_load_nt(wp + min(u, last) * 64);
    // ---- the attention of this wave's heads
    MMI_SHARED __attribute__((aligned(16))) u32x4 xs[WAVES][KPWMAX][64];
    const int h0 = ks0 * 32 / d.Dh, nh = nks * 32 / d.Dh;
    const int j = lane >> 3, c = lane & 7;
    DepAttnOperands in[HPW][BMAX];                                   // every pair's loads first (clamped pairs repeat a live one)
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
        for (int b = 0; b < BMAX; ++b)
            in[hh][b] = mmi_dep_attn8_load(d, min(b, d.B - 1), min(h0 + hh, d.H - 1), lane, false);
    if (blockIdx.x == 0) {                                           // the newest key / value rows into the frame's cache: one workgroup
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
            for (int b = 0; b < BMAX; ++b) {
                if (!(hh < nh && b < d.B) || j != d.k || c >= (d.Dh >> 3)) continue;
                const long at = (((long)b * d.H + h0 + hh) * d.steps + d.k) * d.Dh + 8 * c;
                *reinterpret_cast<u32x4*>(d.kc + at) = in[hh][b].k;
                *reinterpret_cast<u32x4*>(d.vc + at) = in[hh][b].v;
            }
    }
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
        for (int b = 0; b < BMAX; ++b) {
            const bool on = hh < nh && b < d.B;                      // wave-uniform
            const u32x4 ov = mmi_dep_attn8_reduce(d, in[hh][b], lane);
            const int kl = (h0 + hh) * d.Dh + 8 * c - ks0 * 32;      // feature offset inside the wave's slice
            if (on && j == 0 && c < (d.Dh >> 3)) xs[wave][kl >> 5][((kl & 31) >> 3) * 16 + b] = ov;
        }
    __syncthreads();
    acc_t acc;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KPWMAX; ++u) {
        const u32x4 xv = (u < nks && (lane & 15) < d.B) ? xs[wave][u][lane] : zero;
        if (u < nks) acc = mmi_mfma_bf16_16x16x32(wv[u], xv, acc);
    }
    float accv[1][1][R];
#pragma unroll
    for (int r = 0; r < R; ++r) accv[0][0][r] = acc[r];
    mmi_gemm_epilogue<TN, 1, 1, WAVES>(a, accv, wave, lane, nt0, pre, nullptr, g_lo, g_hi);
}

// ------------------------------------------------------------------------------------------------
// sampling (sampling.py:86-106): softmax(logits/temp) -> top-k -> argmax(p / Exp(1)); greedy when disabled
// ------------------------------------------------------------------------------------------------
struct SampleArgs {
    const uint16_t* logits;   // [B][ld] bf16
    int ld, V, k;
    float temp;
    int use_sampling;
    const float* noise;       // [B][noise_ld] Exp(1) draws indexed by rank, used when *use_noise != 0
    int noise_ld;
    const int* use_noise;
    const unsigned long long* rng;   // [0] seed, [1] step counter
    int site;                 // which of the 1 + dep_q sampling sites (RNG stream id)
    int* out;                 // out[b * out_stride]
    int out_stride;
    int B;
    const int* forced;        // teacher forcing (parity taps / depformer_replace_tokens): forced[b * forced_stride]
    int forced_stride;        // is written instead of the sampled token when *use_forced != 0 and the value is >= 0
    const int* use_forced;
    // optional: the sampled token opens the next depth-transformer micro-step (lm.py:465-470) - the workgroup also writes
    // x0[b] = nx_pre[b] + nx_emb[token] (token -1 -> zero row, lm_utils.py:102-124) as the packed activation operand
    const uint16_t* nx_pre;   // [B][nx_ld] bf16: depformer_in[k](transformer_out), computed for all k ahead of the loop
    int nx_ld;
    const uint16_t* nx_emb;   // [card + 1][nx_D]
    uint16_t* nx_out;         // Xp layout (mmi_xp_index), null = nothing to write
    int nx_D, nx_T, nx_ksteps;
    int nx_dup;               // classifier-free guidance: the unconditioned twin of session b is model row b + nx_dup and
                              // opens the micro-step with the same token (lm.py:823-826 `input_.repeat(2, 1, 1)`); 0 = none
};

__device__ __forceinline__ void mmi_sample_next_input(const SampleArgs& a, int b0, int tok) {
    if (!a.nx_out) return;
    const int per = a.nx_D / 8;
    for (int gg = (int)threadIdx.x; gg < (a.nx_dup ? 2 : 1) * per; gg += (int)blockDim.x) {
        const int g = gg % per, b = b0 + (gg / per) * a.nx_dup;
        const int n0 = 8 * g;
        const u32x4 pv = *reinterpret_cast<const u32x4*>(a.nx_pre + (long)b * a.nx_ld + n0);
        u32x4 ev = {0u, 0u, 0u, 0u};
        if (tok != -1) ev = *reinterpret_cast<const u32x4*>(a.nx_emb + (long)(tok < 0 ? 0 : tok) * a.nx_D + n0);
        u32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = mmi_bf16_to_f32((uint16_t)(pv[e] & 0xffffu)) + mmi_bf16_to_f32((uint16_t)(ev[e] & 0xffffu));
            const float hi = mmi_bf16_to_f32((uint16_t)(pv[e] >> 16)) + mmi_bf16_to_f32((uint16_t)(ev[e] >> 16));
            ov[e] = mmi_pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(a.nx_out + mmi_xp_index(a.nx_T, b, n0, a.nx_ksteps)) = ov;
    }
}

// the depth transformer's first input row rebuilt from tok[b] (a step hook changed the text token after the sampler ran)
__global__ void k_dep_next_input(SampleArgs a, const int* __restrict__ tok) {
    mmi_sample_next_input(a, (int)blockIdx.x, tok[(long)blockIdx.x * a.out_stride]);
}

__device__ __forceinline__ int mmi_apply_forced(const SampleArgs& a, int b, int tok) {
    if (*a.use_forced) {
        int f = a.forced[(long)b * a.forced_stride];
        if (f >= 0) return f;
    }
    return tok;
}

__device__ __forceinline__ void mmi_philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                                 unsigned k1) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// Philox4x32-10, all four output words: counter (step, a, b), key = seed
__device__ __forceinline__ void mmi_philox4(unsigned long long seed, unsigned long long step, unsigned a, unsigned b, unsigned (&out)[4]) {
    unsigned c0 = (unsigned)step, c1 = (unsigned)(step >> 32), c2 = a, c3 = b;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        mmi_philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Philox4x32-10 -> one Exp(1) draw
__device__ __forceinline__ float mmi_exp_noise(unsigned long long seed, unsigned long long step, unsigned a, unsigned b) {
    unsigned c0 = (unsigned)step, c1 = (unsigned)(step >> 32), c2 = a, c3 = b;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        mmi_philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    float u = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    return -logf(u);
}

// exclusive prefix sum of one int per thread over the block; returns the block total through *total
template <int NT>
__device__ __forceinline__ int mmi_block_excl_scan(int v, int* wsum /* [NT/64 + 1] shared */, int* total) {
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = mmi_shfl(inc, lane - d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
        int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// One block per session; thread t owns the E consecutive vocabulary entries [t*E, t*E+E) (NT*E >= V; V and the row
// stride multiples of 8), held packed in registers when CACHE (small vocabularies: the 8 depformer heads) or re-read
// from L2 in every pass (the 32000-entry text head).  The logits are bf16, so the top-k threshold is a 16-bit radix
// select (two 256-bin passes) on the order-preserving key of the logit - softmax is monotone in the logit - with ties
// at the threshold resolved towards the lower index; the probabilities p = softmax(logits/temp) are NOT renormalised
// over the top-k set (sampling.py:59-63,95-105), and the token is argmax_i p_i / q_rank(i), q ~ Exp(1) indexed by the
// rank of i in the descending top-k order, exactly the reference's `multinomial` trick (sampling.py:32-46).
__device__ __forceinline__ unsigned mmi_bf16_key(uint16_t h) {       // order-preserving: larger logit <-> larger key
    return (h & 0x8000u) ? (unsigned)(uint16_t)~h : (unsigned)(h | 0x8000u);
}

#ifndef MMI_SAMPLE_STAMP
#define MMI_SAMPLE_STAMP(i)          // scripts/sample_microbench.hip defines it: device-clock stamps at the kernel's stages
#endif
template <int NT, int E, bool CACHE>
__global__ __launch_bounds__(NT) void k_sample(SampleArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    MMI_SAMPLE_STAMP(0)
    const uint16_t* lg = a.logits + (long)b * a.ld;
    const int V = a.V;
    MMI_SHARED float redf[NT / 64];
    MMI_SHARED int redi[NT / 64];
    MMI_SHARED int redr[NT / 64];
    MMI_SHARED int hist[256];
    MMI_SHARED int wsum[NT / 64 + 1];
    MMI_SHARED float sel_val[256];
    MMI_SHARED __attribute__((aligned(16))) unsigned sel_cmp[256];   // (key << 16) | (0xffff - index): larger = earlier rank
    MMI_SHARED int sel_idx[256];
    MMI_SHARED int s_bin;
    MMI_SHARED int s_want;
    MMI_SHARED unsigned redk[NT / 64];
    MMI_SHARED unsigned wcnt[NT / 64][4];
    const int lane = tid & 63, wave = tid >> 6;
    const int i0 = tid * E;
    constexpr int NV = E / 8;
    // the step's control words, requested together with the logits: each is a memory round trip of its own where it is first
    // read otherwise (the RNG words in the middle of the scoring, the forcing words behind the last barrier: ~2 us of an 8 us kernel)
    const int c_use_noise = *a.use_noise;
    const unsigned long long c_seed = a.rng[0], c_step = a.rng[1];
    const int c_forced = *a.use_forced ? a.forced[(long)b * a.forced_stride] : -1;

    u32x4 cache[CACHE ? NV : 1];
    if constexpr (CACHE) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            cache[v] = u32x4{0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};      // bf16 -inf
            if (i0 + v * 8 < V) cache[v] = *reinterpret_cast<const u32x4*>(lg + i0 + v * 8);
        }
    }
    // visit the thread's entries: BODY sees `i` (vocabulary index, < V) and `bits` (the bf16 logit)
#define MMI_S_FOREACH(BODY)                                                                        \
    {                                                                                              \
        _Pragma("unroll") for (int v_ = 0; v_ < NV; ++v_) {                                        \
            if (i0 + v_ * 8 >= V) break;                                                           \
            u32x4 r4_;                                                                             \
            if constexpr (CACHE) r4_ = cache[v_];                                                  \
            else r4_ = *reinterpret_cast<const u32x4*>(lg + i0 + v_ * 8);                          \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                     \
                const int i = i0 + v_ * 8 + e_;                                                    \
                const uint16_t bits = (uint16_t)((e_ & 1) ? (r4_[e_ >> 1] >> 16) : (r4_[e_ >> 1] & 0xffffu)); \
                BODY                                                                               \
            }                                                                                      \
        }                                                                                          \
    }

    // the same, skipping the thread's 8-entry vectors for which SKIP (an expression in v_) holds - the production path keeps the
    // largest key of each vector (vmax) and most vectors of a large vocabulary lie wholly under the top-k threshold
#define MMI_S_FOREACH_UNLESS(SKIP, BODY)                                                           \
    {                                                                                              \
        _Pragma("unroll") for (int v_ = 0; v_ < NV; ++v_) {                                        \
            if (i0 + v_ * 8 >= V) break;                                                           \
            if (SKIP) continue;                                                                    \
            u32x4 r4_;                                                                             \
            if constexpr (CACHE) r4_ = cache[v_];                                                  \
            else r4_ = *reinterpret_cast<const u32x4*>(lg + i0 + v_ * 8);                          \
            _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                     \
                const int i = i0 + v_ * 8 + e_;                                                    \
                const uint16_t bits = (uint16_t)((e_ & 1) ? (r4_[e_ >> 1] >> 16) : (r4_[e_ >> 1] & 0xffffu)); \
                BODY                                                                               \
            }                                                                                      \
        }                                                                                          \
    }

    if (!a.use_sampling || !(a.temp > 0.f)) {
        // torch.argmax(logits): first maximum
        float best = -INFINITY;
        int bi = 0x7fffffff;
        MMI_S_FOREACH({
            const float v = mmi_bf16_to_f32(bits);
            if (v > best || bi == 0x7fffffff) { best = v; bi = i; }
        })
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            float ov = mmi_shfl_xor(best, m);
            int oi = mmi_shfl_xor(bi, m);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { redf[wave] = best; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; ++w)
                if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
            bi = mmi_apply_forced(a, b, bi);
            a.out[(long)b * a.out_stride] = bi;
            redi[0] = bi;
        }
        if (a.nx_out) {
            __syncthreads();
            mmi_sample_next_input(a, b, redi[0]);
        }
        return;
    }

    // The production path (top-k, on-device RNG): argmax_i p_i / q_i over the top-k set does not need the softmax at all -
    // log(p_i / q_i) = logit_i / temp - log q_i + const - and with no recorded draws to replay the Exp(1) draw of an entry may
    // be indexed by the entry instead of by its rank in the sorted top-k (i.i.d. draws assigned independently of their values:
    // the same distribution, checked by the frequency test the reference uses for its own sampler, sampling.py:109-127).  That
    // removes the two softmax passes, the ordered compaction and the rank computation from the critical path of every
    // sampling site; what stays is the radix select of the k-th largest logit.  Supplied noise (the parity taps) keeps the
    // reference's rank-indexed form below.
    const bool fast = a.k > 0 && c_use_noise == 0;
    float mx = -INFINITY, sum = 1.f;
    if (!fast) {
    // ---- softmax statistics of logits / temp (fp32)
    MMI_S_FOREACH({ mx = fmaxf(mx, mmi_bf16_to_f32(bits) / a.temp); })
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, mmi_shfl_xor(mx, m));
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = redf[0];
    for (int w = 1; w < NT / 64; ++w) mx = fmaxf(mx, redf[w]);
    __syncthreads();
    sum = 0.f;
    MMI_S_FOREACH({ sum += expf(mmi_bf16_to_f32(bits) / a.temp - mx); })
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += mmi_shfl_xor(sum, m);
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NT / 64; ++w) sum += redf[w];
    __syncthreads();
    }

    if (a.k <= 0) {
        // top_k = 0: `multinomial(probs)` over the whole vocabulary (sampling.py:98-106 without top-k/top-p): the reference
        // draws one Exp(1) per VOCABULARY ENTRY and takes argmax(p / q) (sampling.py:40-47); here the draw of entry i is the
        // counter RNG at (site, session, i).  Supplied noise (the parity taps) only exists for the top-k form.
        float score = -INFINITY;
        int tok = 0x7fffffff;
        MMI_S_FOREACH({
            const float pr = expf(mmi_bf16_to_f32(bits) / a.temp - mx) / sum;
            const float sc_ = pr / mmi_exp_noise(a.rng[0], a.rng[1], (unsigned)(a.site * a.B + b), (unsigned)i);
            if (sc_ > score || (sc_ == score && i < tok)) { score = sc_; tok = i; }
        })
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float os = mmi_shfl_xor(score, m);
            const int ot = mmi_shfl_xor(tok, m);
            if (os > score || (os == score && ot < tok)) { score = os; tok = ot; }
        }
        if (lane == 0) { redf[wave] = score; redi[wave] = tok; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; ++w)
                if (redf[w] > score || (redf[w] == score && redi[w] < tok)) { score = redf[w]; tok = redi[w]; }
            tok = mmi_apply_forced(a, b, tok);
            a.out[(long)b * a.out_stride] = tok;
            redi[0] = tok;
        }
        if (a.nx_out) {
            __syncthreads();
            mmi_sample_next_input(a, b, redi[0]);
        }
        return;
    }

    const int k = a.k < V ? a.k : V;
    int want = k;
    unsigned Tkey = 0u;
    // ---- production path (round 6): the k-th largest key WITHOUT the two 256-bin LDS histograms over every entry.  The logits of
    // a head share a handful of exponents, so the first histogram is 2048 (text head: 32000) LDS atomics on two or three
    // addresses - serialised - and each pass costs six workgroup barriers.  Instead: the block's largest key, then per-thread
    // PRIVATE counts (8 bits each, packed in 64) of the 8 high bytes at and below it, summed over the wave by shuffles and over
    // the waves through 16 words of LDS; the high byte that holds the k-th largest key is then known to every thread, and only
    // the entries carrying it (typically a few hundred, spread over the low byte's 256 bins) go through an LDS histogram, which
    // every wave scans for itself.  Three barriers instead of twelve.  A set that reaches below those 8 high bytes (more than
    // 2^16 in magnitude under the maximum: never with trained or random-init heads) falls through to the generic select below.
    bool have_t = false;
    int cnt_at_t = -1;                   // entries equal to the threshold key (window select only)
    unsigned vmax[NV];                   // largest key of each of the thread's vectors (production path)
#pragma unroll
    for (int v = 0; v < NV; ++v) vmax[v] = 0xffffu;
    if (fast) {
        unsigned km = 0u;
#pragma unroll
        for (int v = 0; v < NV; ++v) vmax[v] = 0u;
        MMI_S_FOREACH({ const unsigned ky = mmi_bf16_key(bits); vmax[v_] = ky > vmax[v_] ? ky : vmax[v_]; })
#pragma unroll
        for (int v = 0; v < NV; ++v) km = vmax[v] > km ? vmax[v] : km;
        km = mmi_wave_max_u32(km);
        if (lane == 0) redk[wave] = km;
        for (int j = tid; j < 256; j += NT) hist[j] = 0;
        MMI_SAMPLE_STAMP(1)
        __syncthreads();
        MMI_SAMPLE_STAMP(2)
        km = redk[0];
        for (int w = 1; w < NT / 64; ++w) km = redk[w] > km ? redk[w] : km;
        const int hi_max = (int)(km >> 8);
        unsigned long long pc = 0ull;
        MMI_S_FOREACH({
            const int d = hi_max - (int)(mmi_bf16_key(bits) >> 8);
            if (d < 8) pc += 1ull << (8 * d);
        })
        // 8 x 8-bit counters (<= E <= 32 each) -> 4 words of two 16-bit fields: wave sums <= 64 * 32, block sums <= NT * E <= 32768
        unsigned f[4] = {(unsigned)pc & 0x00ff00ffu, ((unsigned)pc >> 8) & 0x00ff00ffu, (unsigned)(pc >> 32) & 0x00ff00ffu, ((unsigned)(pc >> 32) >> 8) & 0x00ff00ffu};
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = mmi_wave_sum_u32(f[q]);
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) wcnt[wave][q] = f[q];
        MMI_SAMPLE_STAMP(3)
        __syncthreads();
        MMI_SAMPLE_STAMP(4)
        unsigned t4[4] = {0u, 0u, 0u, 0u};
        for (int w = 0; w < NT / 64; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) t4[q] += wcnt[w][q];
        // counter d of the packed word: byte d -> word (d >> 2) * 2 + (d & 1), field (d >> 1) & 1
        int cum = 0, dsel = -1, want1 = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned wq = t4[(d >> 2) * 2 + (d & 1)];
            const int c = (int)(((d >> 1) & 1) ? (wq >> 16) : (wq & 0xffffu));
            if (dsel < 0 && cum + c >= k) { dsel = d; want1 = k - cum; }
            cum += c;
        }
        if (dsel >= 0 && hi_max - dsel >= 0) {                     // block-uniform
            const unsigned hi_sel = (unsigned)(hi_max - dsel);
            MMI_S_FOREACH_UNLESS((vmax[v_] >> 8) < hi_sel, {
                const unsigned ky = mmi_bf16_key(bits);
                if ((ky >> 8) == hi_sel) mmi_atomic_add(reinterpret_cast<unsigned*>(&hist[ky & 255u]), 1u);
            })
            MMI_SAMPLE_STAMP(5)
            __syncthreads();
            MMI_SAMPLE_STAMP(6)
            // every wave for itself: lane l looks at bins 255 - 4l .. 252 - 4l (descending keys); `above` = entries in higher bins
            int c4[4], s4 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { c4[j] = hist[255 - 4 * lane - j]; s4 += c4[j]; }
            int inc = s4;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const int o = mmi_shfl(inc, lane - dlt);
                if (lane >= dlt) inc += o;
            }
            int run = inc - s4, fb = 0, fw = 0, fc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (run < want1 && want1 <= run + c4[j]) { fb = 256 - 4 * lane - j; fw = want1 - run; fc = c4[j]; }   // exactly one (lane, j)
                run += c4[j];
            }
            {   // the finder's values are the only non-zero ones: a sum over the wave hands them to every lane
                const unsigned p0 = mmi_wave_sum_u32((unsigned)fb | ((unsigned)fw << 16)), p1 = mmi_wave_sum_u32((unsigned)fc);
                fb = (int)(p0 & 0xffffu); fw = (int)(p0 >> 16); fc = (int)p1;
            }
            Tkey = (hi_sel << 8) | (unsigned)(fb - 1);
            want = fw;
            cnt_at_t = fc;
            have_t = true;
        }
    }
    // ---- radix select of the k-th largest key: high byte, then low byte (supplied noise: the parity taps; and the fallback)
    for (int pass = 0; pass < 2 && !have_t; ++pass) {
        for (int j = tid; j < 256; j += NT) hist[j] = 0;
        __syncthreads();
        MMI_S_FOREACH({
            const unsigned ky = mmi_bf16_key(bits);
            if (pass == 0 || (ky >> 8) == (Tkey >> 8))
                mmi_atomic_add(reinterpret_cast<unsigned*>(&hist[pass == 0 ? (ky >> 8) : (ky & 255u)]), 1u);
        })
        __syncthreads();
        {   // bin holding the want-th largest key: thread t looks at bin 255-t; `above` = entries in higher bins
            const int bin = 255 - tid;
            const int cnt = tid < 256 ? hist[bin] : 0;
            int total;
            const int above = mmi_block_excl_scan<NT>(cnt, wsum, &total);
            if (tid < 256 && above < want && want <= above + cnt) { s_bin = bin; s_want = want - above; }
        }
        __syncthreads();
        Tkey |= (unsigned)s_bin << (pass == 0 ? 8 : 0);
        want = s_want;
        __syncthreads();
    }
    const int want_eq = want;          // how many entries equal to the threshold belong to the set (lowest indices first)
    MMI_SAMPLE_STAMP(7)

    // ---- ordered compaction of the top-k set (index order)
    int n_gt = 0, n_eq = 0;
    MMI_S_FOREACH_UNLESS(vmax[v_] < Tkey, {
        const unsigned ky = mmi_bf16_key(bits);
        n_gt += ky > Tkey ? 1 : 0;
        n_eq += ky == Tkey ? 1 : 0;
    })
    int tot = 0;
    int eq_take = n_eq;                // window select and every entry at the threshold belongs to the set: no scan (block-uniform)
    if (!(have_t && cnt_at_t == want_eq)) {
        const int eq_base = mmi_block_excl_scan<NT>(n_eq, wsum, &tot);
        eq_take = want_eq - eq_base;
        eq_take = eq_take < 0 ? 0 : (eq_take > n_eq ? n_eq : eq_take);
    }
    MMI_SAMPLE_STAMP(8)
    if (fast) {
        // every member of the set scores logit / temp - log(q), q ~ Exp(1); the largest wins.  One Philox4x32-10 call serves the
        // four entries 4j .. 4j + 3 of the vocabulary (counter (step, site * B + session, j), one output word each) and is skipped
        // when none of the four is in the set: two calls per thread for an audio head instead of eight, ~one for the text head
        float best = -INFINITY;
        int bi = 0x7fffffff;
        int eq_seen_f = 0;
        const float inv_t = 1.0f / a.temp;
        const unsigned long long seed = c_seed, step = c_step;
#pragma unroll
        for (int v_ = 0; v_ < NV; ++v_) {
            if (i0 + v_ * 8 >= V) break;
            if (vmax[v_] < Tkey) continue;          // no member of the set in this vector
            u32x4 r4_;
            if constexpr (CACHE) r4_ = cache[v_];
            else r4_ = *reinterpret_cast<const u32x4*>(lg + i0 + v_ * 8);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bool tk[4];
                bool any = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int e_ = 4 * h + e;
                    const uint16_t bits = (uint16_t)((e_ & 1) ? (r4_[e_ >> 1] >> 16) : (r4_[e_ >> 1] & 0xffffu));
                    const unsigned ky = mmi_bf16_key(bits);
                    bool take = ky > Tkey;
                    if (ky == Tkey) { take = eq_seen_f < eq_take; ++eq_seen_f; }
                    tk[e] = take;
                    any = any || take;
                }
                if (!any) continue;
                unsigned r[4];
                mmi_philox4(seed, step, (unsigned)(a.site * a.B + b), (unsigned)((i0 + v_ * 8 + 4 * h) >> 2), r);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!tk[e]) continue;
                    const int e_ = 4 * h + e, i = i0 + v_ * 8 + e_;
                    const uint16_t bits = (uint16_t)((e_ & 1) ? (r4_[e_ >> 1] >> 16) : (r4_[e_ >> 1] & 0xffffu));
                    const float u = ((float)(r[e] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
                    const float sc_ = mmi_bf16_to_f32(bits) * inv_t - mmi_fast_logf(-mmi_fast_logf(u));
                    if (sc_ > best || (sc_ == best && i < bi)) { best = sc_; bi = i; }
                }
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = mmi_shfl_xor(best, m);
            const int oi = mmi_shfl_xor(bi, m);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        MMI_SAMPLE_STAMP(9)
        if (lane == 0) { redf[wave] = best; redi[wave] = bi; }
        __syncthreads();
        MMI_SAMPLE_STAMP(10)
        // only the threads that write the next micro-step's input row (and thread 0, for the token) go on: a 1024-thread workgroup
        // would otherwise run this tail sixteen times over on its four SIMDs
        if (tid != 0 && (!a.nx_out || tid >= (a.nx_dup ? 2 : 1) * (a.nx_D / 8))) return;
        best = redf[0]; bi = redi[0];                   // every remaining thread folds the waves' winners itself: no second barrier
        for (int w = 1; w < NT / 64; ++w)
            if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
        bi = c_forced >= 0 ? c_forced : bi;            // (mmi_apply_forced, on the words requested at the top)
        if (tid == 0) a.out[(long)b * a.out_stride] = bi;
        MMI_SAMPLE_STAMP(11)
        mmi_sample_next_input(a, b, bi);
        MMI_SAMPLE_STAMP(12)
        return;
    }
    int pos = mmi_block_excl_scan<NT>(n_gt + eq_take, wsum, &tot);
    int eq_seen = 0;
    MMI_S_FOREACH({
        const unsigned ky = mmi_bf16_key(bits);
        bool take = ky > Tkey;
        if (ky == Tkey) { take = eq_seen < eq_take; ++eq_seen; }
        if (take) {
            if (pos < 256) {
                sel_val[pos] = expf(mmi_bf16_to_f32(bits) / a.temp - mx) / sum;
                sel_cmp[pos] = (ky << 16) | (unsigned)(0xffff - i);
                sel_idx[pos] = i;
            }
            ++pos;
        }
    })
#undef MMI_S_FOREACH
#undef MMI_S_FOREACH_UNLESS
    __syncthreads();
    // ---- rank inside the set (descending logit, then index) and the noisy argmax
    float score = -INFINITY;
    int rank = 0x7fffffff, tok = 0;
    for (int j = k + tid; j < 256; j += NT) sel_cmp[j] = 0u;      // entries past the set never outrank anything
    __syncthreads();
    if (tid < k && tid < 256) {
        const float v = sel_val[tid];
        const unsigned mine = sel_cmp[tid];
        const int id = sel_idx[tid];
        int r = 0;
        for (int m = 0; m < k; m += 4) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(&sel_cmp[m]);
            r += (o[0] > mine ? 1 : 0) + (o[1] > mine ? 1 : 0) + (o[2] > mine ? 1 : 0) + (o[3] > mine ? 1 : 0);
        }
        float q;
        if (*a.use_noise) q = a.noise[(long)b * a.noise_ld + r];
        else q = mmi_exp_noise(a.rng[0], a.rng[1], (unsigned)(a.site * a.B + b), (unsigned)r);
        score = v / q;
        rank = r;
        tok = id;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        float os = mmi_shfl_xor(score, m);
        int orank = mmi_shfl_xor(rank, m), otok = mmi_shfl_xor(tok, m);
        if (os > score || (os == score && orank < rank)) { score = os; rank = orank; tok = otok; }
    }
    if (lane == 0) { redf[wave] = score; redi[wave] = tok; redr[wave] = rank; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (redf[w] > score || (redf[w] == score && redr[w] < rank)) { score = redf[w]; rank = redr[w]; tok = redi[w]; }
        tok = mmi_apply_forced(a, b, tok);
        a.out[(long)b * a.out_stride] = tok;
        redi[0] = tok;
    }
    if (a.nx_out) {
        __syncthreads();
        mmi_sample_next_input(a, b, redi[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// LMGen delay ring (lm.py:668-783)
// ------------------------------------------------------------------------------------------------
struct TokArgs {
    int* cache;            // [B][NC][CT]
    long* offsets;         // [B]
    const uint8_t* exec;   // [B]
    const int* delays;     // [NC]
    int B, NC, CT, dep_q, max_delay, card, text_card;
};

// 1. write the user's codes at (offset+delay)%CT, 2. gather the model input at offset%CT with init-token substitution
// 3. (threads past B*NC) the RoPE table of the step: (cos, sin)(offset[b] * max_period^(-2j/Dh)) for the new position of
//    every session (rope.py:11-82), computed once here instead of in each of the temporal layers' in_proj epilogues
__global__ void k_lm_prepare(TokArgs t, const int* __restrict__ user, int n_user, int* __restrict__ tokens, float* __restrict__ rope,
                             int Dh, float max_period) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= t.B * t.NC) {
        const int r = idx - t.B * t.NC;
        if (r < t.B * (Dh / 2)) {
            const int b = r / (Dh / 2), j = r - b * (Dh / 2);
            const float freq = expf((float)j * (-logf(max_period) * 2.0f / (float)Dh));
            const float ang = freq * (float)t.offsets[b];
            rope[2 * r] = cosf(ang);
            rope[2 * r + 1] = sinf(ang);
        }
        return;
    }
    const int b = idx / t.NC, c = idx % t.NC;
    const long off = t.offsets[b];
    const bool ex = t.exec[b] != 0;
    int* row = t.cache + ((long)b * t.NC + c) * t.CT;
    const int first_user = t.dep_q + 1;
    if (c >= first_user && ex) row[(int)((off + t.delays[c]) % t.CT)] = user[(long)b * n_user + (c - first_user)];
    const bool is_init = off <= (long)t.delays[c] || !ex;
    int tok = row[(int)(off % t.CT)];
    if (is_init) tok = c == 0 ? t.text_card : t.card;
    tokens[idx] = tok;
}

// 4./5. advance offsets, store the sampled tokens, gather the delayed output frame
__global__ void k_lm_commit(TokArgs t, const int* __restrict__ text_tok, const int* __restrict__ audio_tok,
                            int* __restrict__ out, unsigned long long* __restrict__ rng) {
    int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b == 0) rng[1] += 1ull;
    if (b >= t.B) return;
    const bool ex = t.exec[b] != 0;
    const long off_new = t.offsets[b] + (ex ? 1 : 0);
    t.offsets[b] = off_new;
    const int pos = (int)(off_new % t.CT);
    int* rows = t.cache + (long)b * t.NC * t.CT;
    if (ex) {
        rows[pos] = text_tok[b];
        for (int k = 0; k < t.dep_q; ++k) rows[(1 + k) * t.CT + pos] = audio_tok[(long)b * t.dep_q + k];
    }
    const bool hide = off_new <= (long)t.max_delay || !ex;
    for (int c = 0; c <= t.dep_q; ++c) {
        long i = (off_new - t.max_delay + t.delays[c]) % t.CT;
        if (i < 0) i += t.CT;
        int v = rows[c * t.CT + (int)i];
        out[(long)b * (t.dep_q + 1) + c] = hide ? -2 : v;
    }
}

__global__ void k_i64_to_i32(const long* __restrict__ src, long src_rstride, int* __restrict__ dst, int rows, int cols) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= rows * cols) return;
    int r = idx / cols, c = idx % cols;
    dst[idx] = (int)src[(long)r * src_rstride + c];
}
__global__ void k_i32_to_i64(const int* __restrict__ src, long* __restrict__ dst, int n) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx < n) dst[idx] = (long)src[idx];
}
__global__ void k_bf16_to_f32(const uint16_t* __restrict__ src, float* __restrict__ dst, long n) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] = mmi_bf16_to_f32(src[idx]);
}
__global__ void k_fill_i32(int* __restrict__ p, int v, long n) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) p[idx] = v;
}
__global__ void k_lm_reset(TokArgs t, const uint8_t* __restrict__ mask, uint8_t* __restrict__ exec) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= t.B) return;
    if (mask && !mask[idx]) return;
    t.offsets[idx] = 0;           // lm.py:537-542; transformer.py:329-334 (KV end_offset, MHA offset)
    exec[idx] = 1;                // streaming.py:43-44
}

// ------------------------------------------------------------------------------------------------
// classifier-free guidance (lm.py:646-665, 712-733, 823-832) and extra heads (lm.py:793-807)
// ------------------------------------------------------------------------------------------------
// With cfg_coef != 1 the model runs 2G rows for G sessions: rows [0,G) conditioned, rows [G,2G) their unconditioned twins.
// After k_lm_prepare has built the G conditioned input rows, this kernel derives the twins: the same tokens, except
//   cfg_is_masked_until: every codebook reads the zero token (-1 -> zero embedding) while offset <= delay + masked_until[b],
//   cfg_is_no_text     : the text codebook reads the zero token,
// in both cases only where the row is past its initial token; it also replicates the per-session offset and RoPE angles
// onto the model rows (reset / exec masks reach the twins the same way: `.repeat(2)`, lm.py:655-663).
__global__ void k_lm_cfg_twins(TokArgs t, int* __restrict__ tokens, const int* __restrict__ masked_until, int no_text,
                               long* __restrict__ offsets_m, float* __restrict__ rope, int Dh) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int G = t.B;
    const int n_tok = G * t.NC, n_rope = G * Dh;
    if (idx < n_tok) {
        const int b = idx / t.NC, c = idx % t.NC;
        const long off = t.offsets[b];
        const bool is_init = off <= (long)t.delays[c] || !t.exec[b];
        int tok = tokens[idx];
        if (!is_init) {
            if (masked_until && off <= (long)t.delays[c] + (long)masked_until[b]) tok = -1;
            if (no_text && c == 0) tok = -1;
        }
        tokens[n_tok + idx] = tok;
    } else if (idx < n_tok + n_rope) {
        const int r = idx - n_tok;
        rope[n_rope + r] = rope[r];
    } else if (idx < n_tok + n_rope + G) {
        const int b = idx - n_tok - n_rope;
        offsets_m[b] = t.offsets[b];
        offsets_m[G + b] = t.offsets[b];
    }
}

// logits[b] <- logits_null + (logits - logits_null) * coef, each operation on bf16 tensors (lm.py:733, 830-832);
// logits rows [0,G) conditioned, [G,2G) unconditioned; the mix lands in row b, which the sampler and the taps then read
__global__ void k_cfg_mix(uint16_t* __restrict__ logits, int ld, int V, int G, float coef) {
    const int b = blockIdx.y;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= V) return;
    uint16_t* c = logits + (long)b * ld + i;
    const float l = mmi_bf16_to_f32(*c), n = mmi_bf16_to_f32(logits[(long)(G + b) * ld + i]);
    const float d = mmi_round_bf16(l - n);
    const float m = mmi_round_bf16(d * coef);
    *c = mmi_f32_to_bf16(n + m);
}

// softmax(extra_head(transformer_out)) (lm.py:803-806): one workgroup per (head, model row), one wave per output feature
// group; tout is the packed activation operand of the text head.  probs [rows][n_heads][hdim] fp32 (bf16 values).
__global__ __launch_bounds__(64) void k_extra_heads(const uint16_t* __restrict__ tout, int T, int ksteps, const uint16_t* __restrict__ w,
                                                   int D, int hdim, int n_heads, float* __restrict__ probs) {
    const int b = blockIdx.x, h = blockIdx.y, lane = (int)threadIdx.x;
    MMI_SHARED float lg[64];
    for (int j = 0; j < hdim; ++j) {
        const uint16_t* wr = w + ((long)h * hdim + j) * D;
        float acc = 0.f;
        for (int k = lane; k < D; k += 64) acc += mmi_bf16_to_f32(tout[mmi_xp_index(T, b, k, ksteps)]) * mmi_bf16_to_f32(wr[k]);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc += mmi_shfl_xor(acc, m);
        if (lane == 0) lg[j] = mmi_round_bf16(acc);          // nn.Linear output in bf16
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = 0; j < hdim; ++j) mx = fmaxf(mx, lg[j]);
    float den = 0.f;
    for (int j = 0; j < hdim; ++j) den += expf(lg[j] - mx);
    if (lane < hdim) probs[((long)b * n_heads + h) * hdim + lane] = mmi_round_bf16(expf(lg[lane] - mx) / den);
}
