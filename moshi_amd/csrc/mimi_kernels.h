// Mimi codec kernels for gfx950 (fp32, like the reference: loaders.get_mimi never casts, loaders.py:353).
//
// Data layout in HBM
//   activations   [B][C][H + T] fp32, time contiguous ("conv layout", same as the reference's [B,C,T]).
//                 The first H columns of a buffer are the causal history of its consumer conv
//                 (the reference's `previous` ring state, conv.py:161-169,245-274); the producer writes
//                 its T new columns behind them and k_commit_history shifts the last H columns to the
//                 front once the frame is done (only for rows whose exec mask is set).
//   conv weights  packed at load into MFMA A-fragment order for v_mfma_f32_32x32x2_f32:
//                 P[mt][q][lane][e] = W[mt*32 + (lane&31)][(q*4+e)*2 + (lane>>5)]  (zero padded),
//                 so that a wave's 16-byte-per-lane load is one contiguous 1 KiB read feeding 4 MFMAs.
//   codebooks     E[level][bins][D] fp32 + ||e||^2 in fp64.
//   KV ring       [layer][2][B][H][cap][D] fp32 (transformer.py:196-288).
//
// Every conv / linear of the codec (SEANet convs, the GEMM half of the transposed convs, the 1x1
// RVQ projections, the transformer's linears) is the implicit GEMM
//   out[co][n=(b,t)] = sum_{kd=(ci,k)} W[co][kd] * act(in[b][ci][t*S + k])
// on the fp32 MFMA (an exact fma chain, so results track the fp32 reference to rounding order); see k_conv_wide /
// k_gemm_f32 below.  Activations that feed k_gemm_f32 are additionally kept as a packed B operand:
//   Bp[nt][q][lane][e] = act(in)[kd = (q*4+e)*2 + (lane>>5)][n = nt*32 + (lane&31)]
#pragma once
#include "mmi_common.h"

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// P[mt][q][lane][e] = W[m*sm + kd*sk], m = mt*32 + (lane&31), kd = (q*4+e)*2 + (lane>>5)
__global__ void k_pack_a_f32(const float* __restrict__ W, float* __restrict__ P, int M, int Kd, long sm, long sk,
                             int Mt, int Q) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Mt * Q * 256;
    if (idx >= total) return;
    int e = (int)(idx & 3);
    int lane = (int)((idx >> 2) & 63);
    long rest = idx >> 8;
    int q = (int)(rest % Q);
    int mt = (int)(rest / Q);
    int m = mt * 32 + (lane & 31);
    int kd = (q * 4 + e) * 2 + (lane >> 5);
    float v = 0.f;
    if (m < M && kd < Kd) v = W[(long)m * sm + (long)kd * sk];
    P[idx] = v;
}

// dst[r*dld + c] = src[r*sld + c]  (row-block copy used to concatenate weight matrices / stage I/O)
__global__ void k_copy2d_f32(const float* __restrict__ src, long sld, float* __restrict__ dst, long dld, int rows,
                             int cols) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    int r = (int)(idx / cols), c = (int)(idx % cols);
    dst[(long)r * dld + c] = src[(long)r * sld + c];
}

// codebook = embedding_sum / clamp(cluster_usage, eps)[:, None]   (core_vq.py:178-186); e2 = ||e||^2 (fp64)
__global__ void k_codebook_prepare(const float* __restrict__ esum, const float* __restrict__ usage,
                                   float* __restrict__ E, double* __restrict__ e2, int bins, int D, float eps) {
    int c = blockIdx.x;
    if (c >= bins) return;
    float u = usage[c];
    u = u < eps ? eps : u;
    MMI_SHARED double red[64];
    double acc = 0.0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v = esum[(long)c * D + d] / u;
        E[(long)c * D + d] = v;
        acc += (double)v * (double)v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)blockDim.x; ++i) s += red[i];
        e2[c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// implicit-GEMM conv / linear
// ------------------------------------------------------------------------------------------------
// out[co][n=(b,t)] = sum_{kd=(ci,k)} W[co][kd] * act(in[b][ci][t*S + k]), fp32 MFMA 32x32x2 (exact fma chain).
// Two kernels cover the codec:
//   k_conv_wide  (N = B*T_out > 128 columns: SEANet layers at audio rate).  A wave owns 32 columns x MTB m-tiles x a
//                K-slice; lane (j, h) gathers exactly the B-operand elements its own MFMA lane consumes
//                (column j, reduction index kd = 2q+h), applies ELU once, and reuses them for the MTB m-tiles.
//                The reduction index -> input offset map (ci*x_ld + k) is a per-layer table, 16 bytes per k-quad.
//   k_gemm_f32   (N <= 128: the 12.5/25 Hz layers - transformer linears, last strided convs, resampling, RVQ
//                projections; weights dominate).  Same shape as the LM's weight-streaming GEMM: one m-tile per
//                workgroup, K split over waves (+ workgroups), activations pre-packed into B-fragment order by
//                k_pack_b_f32 (im2col + ELU + replicate padding) or by the producing kernel.
enum { MMI_ACT_NONE = 0, MMI_ACT_GELU = 1, MMI_ACT_ELU = 2 };

// n / d for 0 <= n < 2^17 and 1 <= d < 2^11 (column and tap indices) as one multiply-high: an integer division costs
// ~30 instructions, and the small kernels here are instruction bound.  magic = ceil(2^32 / d), 0 means d == 1.
MMI_HD unsigned mmi_div_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
MMI_HD int mmi_fast_div(int n, unsigned magic) {
    return magic == 0u ? n : (int)(((unsigned long long)(unsigned)n * magic) >> 32);
}
enum { MMI_GOUT_NATURAL = 0, MMI_GOUT_PACKED = 1, MMI_GOUT_PARTIAL = 2 };

struct ConvGemmArgs {
    const float* x;       // input [B][Cin][x_ld]; position p of a row = history (p < H) then the T new columns
    long x_bstride;       // Cin * x_ld
    int x_ld;
    int x_off;            // first column this conv reads (H_buffer - (K - S)); normally 0
    int H;                // history columns in front of the new data (for replicate-first)
    const float* wpk;     // packed weights
    const float* bias;    // [Cout] or null
    float* out;           // [B][Cout][out_ld], written at column out_off + t
    int out_ld, out_off;
    const float* res;     // residual [B][Cout][res_ld] read at res_off + t, or null
    int res_ld, res_off;
    const float* scale;   // LayerScale [Cout] or null: out = res + scale * (acc + bias)
    const uint8_t* first; // replicate-pad flags [B] or null (conv.py:253-259)
    const uint8_t* exec;  // exec mask [B] (only read with `first`)
    int B, Cin, Cout, K, S, T_out;
    unsigned T_magic, K_magic;   // mmi_div_magic(T_out), mmi_div_magic(K)
    int Mt, Q;            // M tiles of 32, packed k-quads (Kdim_pad / 8)
    int Ntot;             // B * T_out
    int elu_in;           // apply ELU(alpha=1) to every loaded input (seanet.py:63,205,222)
    int act_out;          // MMI_ACT_*
    // ELU hoisted out of the consumers: every SEANet conv but the first reads ELU(x) (seanet.py:63,205,222), and an audio-rate
    // conv reads each input K * (Cout / 32) times, so the PRODUCER applies it once per element: elu_out = 1 stores ELU(result)
    // instead of the result (buffers whose only readers are such convs), out2 != null additionally stores ELU(result) there
    // (same geometry as `out`; block inputs, which the residual connection also needs raw).  Same function on the same values
    // as applying it at the consumer's load - bit-identical.
    int elu_out;
    float* out2;
    // k_gemm_f32_ln: nn.LayerNorm over the Cin channels of every column fused in front of the linear (the Mimi transformer's
    // norm1 -> in_proj and norm2 -> linear1): `bp` then holds the RAW residual stream in packed order, ln_w / ln_b the norm's
    // weight and bias in the operand's element order (k_pack_ln), ln_eps its epsilon
    const float* ln_w;
    const float* ln_b;
    float ln_eps;
    // --- filled in by the engine's planner
    const int* koff;      // k_conv_wide: [Q][2][4] input offset of kd = (q*4+e)*2 + h; 0 past Cin*K (the weights there are 0)
    const float* bp;      // k_gemm_f32: packed activations [ceil(N/32)][Q][64][4]
    int x_packed;         // the producer already wrote `bp` (no k_pack_b_f32 launch)
    int out_mode;         // MMI_GOUT_*
    float* outp;          // GOUT_PACKED: packed B operand of the consuming linear, [ceil(N/32)][outQ][64][4]
    int outQ;
    float* partial;       // GOUT_PARTIAL: [gridDim.y][Mt*32][Npad] raw sums, finished by k_conv_finish
    int Npad;
};

__device__ __forceinline__ float mmi_elu(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float mmi_gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// index of element (kd, n) in a packed B operand with Q k-quads
__device__ __forceinline__ long mmi_bp_index(int kd, int n, int Q) {
    return ((((long)(n >> 5) * Q + (kd >> 3)) * 64) + (kd & 1) * 32 + (n & 31)) * 4 + ((kd & 7) >> 1);
}

// bias / activation / LayerScale / residual, then the store to [B][Cout][out_ld] - shared by the epilogues.
// NV results of one thread are finished together: all bias / scale / residual loads are issued first (under branches
// that are uniform for the launch), so that the epilogue costs one memory round trip instead of NV of them.
template <int NV>
__device__ __forceinline__ void mmi_conv_store_n(const ConvGemmArgs& a, const int (&co)[NV], const int (&b)[NV], const int (&t)[NV],
                                                 const bool (&ok)[NV], float (&v)[NV]) {
    float bia[NV], scl[NV], rsv[NV];
    long row[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cc = ok[i] ? co[i] : 0;
        row[i] = (long)(ok[i] ? b[i] : 0) * a.Cout + cc;
        bia[i] = 0.f; scl[i] = 1.f; rsv[i] = 0.f;
    }
    if (a.bias) {
#pragma unroll
        for (int i = 0; i < NV; ++i) bia[i] = a.bias[ok[i] ? co[i] : 0];
    }
    if (a.scale) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scl[i] = a.scale[ok[i] ? co[i] : 0];
    }
    if (a.res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) rsv[i] = a.res[row[i] * a.res_ld + a.res_off + (ok[i] ? t[i] : 0)];
    }
    // launch-uniform flags are tested once per group of NV values, not per value: these kernels are instruction bound, and a
    // per-value maze of scalar branches (with the erf and expm1 bodies in line) was a third of a residual block's time
    if (a.bias) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += bia[i];
    }
    if (a.act_out == MMI_ACT_GELU) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = mmi_gelu_erf(v[i]);
    } else if (a.act_out == MMI_ACT_ELU) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = mmi_elu(v[i]);
    }
    if (a.scale) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] *= scl[i];
    }
    if (a.res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = rsv[i] + v[i];
    }
    long at[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) at[i] = row[i] * a.out_ld + a.out_off + (ok[i] ? t[i] : 0);
    if (a.out2) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) { a.out[at[i]] = v[i]; a.out2[at[i]] = mmi_elu(v[i]); }
    } else if (a.elu_out) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = mmi_elu(v[i]);
            if (ok[i]) a.out[at[i]] = v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) a.out[at[i]] = v[i];
    }
    // also stored as the packed operand of the consuming linear (what `out` holds: the ELU'd value where that is what is stored):
    // the residual stream of the Mimi transformers for the next fused norm + linear, the decoder's first conv for the first
    // transposed-conv GEMM, the latent for the RVQ input projection - no k_pack_b_f32 launch in between
    if (a.outp && a.out_mode == MMI_GOUT_NATURAL) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (ok[i]) a.outp[mmi_bp_index(co[i], b[i] * a.T_out + t[i], a.outQ)] = v[i];
    }
}

template <int MTB, int W, int U, bool ELU_IN = true>
__global__ __launch_bounds__(W * 64) void k_conv_wide(ConvGemmArgs a) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int jl = lane & 31, kh = lane >> 5;
    const int n = (int)blockIdx.x * 32 + jl;
    const int mt0 = (int)blockIdx.y * MTB;
    const bool nvalid = n < a.Ntot;
    const int nn = nvalid ? n : 0;
    const int b = mmi_fast_div(nn, a.T_magic), t = nn - b * a.T_out;
    const float* xb = a.x + (long)b * a.x_bstride + a.x_off + t * a.S;
    const int qper = (a.Q + W - 1) / W;
    const int q0 = min(a.Q, wave * qper), q1 = min(a.Q, q0 + qper);

    f32x16 acc[MTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const f32x4* wp[MTB];
#pragma unroll
    for (int m = 0; m < MTB; ++m) wp[m] = reinterpret_cast<const f32x4*>(a.wpk) + (long)min(mt0 + m, a.Mt - 1) * a.Q * 64 + lane;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4* ko = reinterpret_cast<const i32x4*>(a.koff) + kh;

    // groups of U k-quads, double buffered: the gathers / weight fragments of group g+1 are in flight while group g
    // runs on the matrix core.  Loads past the slice are clamped to its last quad and their MFMAs skipped.
    f32x4 avA[U][MTB], avB[U][MTB];
    float bvA[U][4], bvB[U][4];
#define MMI_W_LOAD(AV, BV, qb)                                                                      \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                 \
        const int qq = min((qb) + u, q1 - 1);                                                       \
        const i32x4 o = ko[qq * 2];                                                                 \
        /* plain unconditional gathers (a conditional load would be serialised behind s_waitcnt vmcnt(0)): reduction   \
           indices past Cin*K point at offset 0 and meet zero weights; columns past Ntot are computed and discarded */  \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) BV[u][e] = xb[o[e]];                          \
        _Pragma("unroll") for (int m = 0; m < MTB; ++m) AV[u][m] = wp[m][(long)qq * 64];           \
    }
#define MMI_W_MMA(AV, BV, qb)                                                                       \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                 \
        if ((qb) + u < q1) {                                                                        \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
                const float v = (ELU_IN && a.elu_in) ? mmi_elu(BV[u][e]) : BV[u][e];                \
                _Pragma("unroll") for (int m = 0; m < MTB; ++m) acc[m] = mmi_mfma_f32_32x32x2(AV[u][m][e], v, acc[m]); \
            }                                                                                       \
        }                                                                                           \
    }
    const int G = (q1 - q0 + U - 1) / U;
    if (G > 0) {
        MMI_W_LOAD(avA, bvA, q0);
        int g = 0;
        for (; g + 2 < G; g += 2) {
            MMI_W_LOAD(avB, bvB, q0 + (g + 1) * U);
            MMI_W_MMA(avA, bvA, q0 + g * U);
            MMI_W_LOAD(avA, bvA, q0 + (g + 2) * U);
            MMI_W_MMA(avB, bvB, q0 + (g + 1) * U);
        }
        if (G - g == 2) {
            MMI_W_LOAD(avB, bvB, q0 + (g + 1) * U);
            MMI_W_MMA(avA, bvA, q0 + g * U);
            MMI_W_MMA(avB, bvB, q0 + (g + 1) * U);
        } else {
            MMI_W_MMA(avA, bvA, q0 + g * U);
        }
    }
#undef MMI_W_LOAD
#undef MMI_W_MMA

    if (W > 1) {   // split-K reduction across the waves, fixed order
        MMI_DYN_SHARED(float, red);
        const int NE = MTB * 16 * 64;
#pragma unroll
        for (int m = 0; m < MTB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * NE + (m * 16 + r) * 64 + lane] = acc[m][r];
        __syncthreads();
        constexpr int NVT = MTB * 16 / W;                      // results per thread (W divides 16)
        constexpr int NV = NVT < 8 ? NVT : 8;                  // finished 8 at a time (register budget)
#pragma unroll 1
        for (int i0 = 0; i0 < NVT; i0 += NV) {
            int co[NV], bb[NV], tt[NV];
            bool ok[NV];
            float v[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e = (int)threadIdx.x + (i0 + i) * W * 64;
                const int le = e & 63, r = (e >> 6) & 15, m = e >> 10;
                const int nn2 = (int)blockIdx.x * 32 + (le & 31);
                co[i] = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (le >> 5);
                ok[i] = nn2 < a.Ntot && co[i] < a.Cout;
                const int n3 = ok[i] ? nn2 : 0;
                bb[i] = mmi_fast_div(n3, a.T_magic);
                tt[i] = n3 - bb[i] * a.T_out;
                float s = 0.f;
                for (int w = 0; w < W; ++w) s += red[w * NE + e];
                v[i] = s;
            }
            mmi_conv_store_n<NV>(a, co, bb, tt, ok, v);
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MTB; ++m) {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {
            int co[8], bb[8], tt[8];
            bool ok[8];
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i;
                co[i] = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                ok[i] = nvalid && co[i] < a.Cout;
                bb[i] = b; tt[i] = t;
                v[i] = acc[m][r];
            }
            mmi_conv_store_n<8>(a, co, bb, tt, ok, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SEANet residual block in ONE launch (seanet.py:31-77 SEANetResnetBlock, true_skip):
//     y = x + conv1x1(ELU(convK(ELU(x))))        convK: C -> C/2 (K = residual_kernel_size, dilation 1), conv1x1: C/2 -> C
// at audio rate (N = B*T > 128 columns).  As two k_conv_wide launches the block writes the hidden tensor to HBM and reads it
// back, and pays two launches for ~1 GFLOP; here a workgroup owns 32 columns and MT1 = ceil((C/2) / 32) waves:
//   stage 0  the tile's input window (Cin x (32 + K - 1) floats of ONE session: T % 32 == 0) is staged in LDS with coalesced row
//            loads that are all in flight at once - one memory round trip for the whole tile, where the gather loop of
//            k_conv_wide pays one per group of k-quads (measured: ~4 us per group at one wave per SIMD)
//   stage 1  wave w computes hidden m-tile w over the whole reduction, B operand from LDS (ds_read at table offsets): acc1
//   hand-off h = ELU(acc1 + b1) stays in the accumulator layout: lane (j, hh) holds hidden rows 32w + 8(r>>2) + 4hh + (r&3),
//            r = 0..15, of its own column j.  The second conv's reduction index is free to be ANY order, so its weights are packed
//            (k_pack_a_f32_hperm) such that MFMA k-slot (qq, hh) of the 32x32x2 B operand IS hidden row 32(qq/16) + 8((qq%16)>>2)
//            + 4hh + (qq&3): the B operand of stage 2 is the lane's own registers.  MT1 > 1: the waves swap their tiles through
//            a per-lane LDS mailbox [MT1][4][64] x 16 bytes (conflict free, one barrier).
//   stage 2  wave w computes output m-tiles w, w + MT1, ... : 16 * MT1 MFMAs each, weights streamed from L2, then the usual
//            epilogue (bias, residual x, ELU'd store / twin).
// a1 / a2 are the ConvGemmArgs the two launches would get (a2.wpk replaced by the permuted packing).
struct ResBlockArgs {
    ConvGemmArgs a1, a2;
};

// P[mt][q][lane][e] = W[m][kperm], m = mt*32 + (lane&31), k-slot qq = q*4+e, hh = lane>>5, kperm as above (K = 1 convs only)
__global__ void k_pack_a_f32_hperm(const float* __restrict__ W, float* __restrict__ P, int M, int Kd, int Mt, int Q) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)Mt * Q * 256;
    if (idx >= total) return;
    int e = (int)(idx & 3);
    int lane = (int)((idx >> 2) & 63);
    long rest = idx >> 8;
    int q = (int)(rest % Q);
    int mt = (int)(rest / Q);
    int m = mt * 32 + (lane & 31);
    int qq = q * 4 + e, hh = lane >> 5;
    int kd = 32 * (qq >> 4) + 8 * ((qq & 15) >> 2) + 4 * hh + (qq & 3);
    float v = 0.f;
    if (m < M && kd < Kd) v = W[(long)m * Kd + kd];
    P[idx] = v;
}

template <int MT1, int U>
__global__ __launch_bounds__(MT1 * 64) void k_resblock(ResBlockArgs ra) {
    const ConvGemmArgs& a = ra.a1;
    const ConvGemmArgs& c = ra.a2;
    constexpr int NTH = MT1 * 64;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int jl = lane & 31, kh = lane >> 5;
    // the tile's 32 columns belong to ONE session (T_out % 32 == 0, checked by the engine): b, t0 are uniform
    const int n0 = (int)blockIdx.x * 32;
    const int b = mmi_fast_div(n0, a.T_magic), t0 = n0 - b * a.T_out;
    const int LD = 32 + a.K - 1;                 // input columns the tile's windows cover (stride 1)
    MMI_DYN_SHARED(float, smem);                 // Xs[Cin][LD] | offs[Q][2][4] | mailbox[MT1][4][64][4]
    float* Xs = smem;
    int* offs = reinterpret_cast<int*>(smem + ((a.Cin * LD + 3) & ~3));
    float* hs = reinterpret_cast<float*>(offs + a.Q * 8);

    // ---- stage 0: the input window -> LDS in ONE memory round trip: wave w stages channels w, w + MT1, ..., one row of LD
    // consecutive floats per load instruction (coalesced; Cin <= 64 * MT1), all requested before the first one is used
    constexpr int NR = 64;
    float xr[NR];
    {
        const float* xw = a.x + (long)b * a.x_bstride + a.x_off + t0;
        const int jj = lane < LD ? lane : LD - 1;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ci = min(wave + i * MT1, a.Cin - 1);
            xr[i] = xw[(long)ci * a.x_ld + jj];
        }
    }
    // meanwhile: the weight fragments of this wave's first output tile, the hidden tile's bias, the first stage-1 fragments
    constexpr int Q2 = 4 * MT1;
    f32x4 w2v[Q2], w2n[Q2];
    {
        const f32x4* w2 = reinterpret_cast<const f32x4*>(c.wpk) + (long)min(wave, c.Mt - 1) * c.Q * 64 + lane;
#pragma unroll
        for (int q = 0; q < Q2; ++q) w2v[q] = w2[(long)min(q, c.Q - 1) * 64];
    }
    float b1v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
        b1v[r] = a.bias ? a.bias[min(row, a.Cout - 1)] : 0.f;
        if (row >= a.Cout) b1v[r] = 0.f;
    }
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk) + (long)min(wave, a.Mt - 1) * a.Q * 64 + lane;
    const int q1 = a.Q;
    f32x4 avA[U], avB[U];
#define MMI_R_LOAD(AV, qb)                                                                          \
    _Pragma("unroll") for (int u = 0; u < U; ++u) AV[u] = wp[(long)min((qb) + u, q1 - 1) * 64];
    MMI_R_LOAD(avA, 0);
    // reduction index -> LDS offset of its window start: kd = (q*4+e)*2 + hh -> (ci, k) -> ci * LD + k; 0 past Cin*K (zero weights)
    for (int i = tid; i < a.Q * 8; i += NTH) {
        const int e = i & 3, hh = (i >> 2) & 1, q = i >> 3;
        const int kd = (q * 4 + e) * 2 + hh;
        const int ci = mmi_fast_div(kd, a.K_magic), k = kd - ci * a.K;
        offs[i] = kd < a.Cin * a.K ? ci * LD + k : 0;
    }
    if (lane < LD) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int ci = wave + i * MT1;
            if (ci < a.Cin) Xs[ci * LD + lane] = a.elu_in ? mmi_elu(xr[i]) : xr[i];
        }
    }
    __syncthreads();

    // ---- stage 1: hidden m-tile `wave` = W1[32 rows][Cin*K] . Xs windows; weights one group of U quads ahead, B operand from LDS
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4* ko = reinterpret_cast<const i32x4*>(offs) + kh;
    const float* xl = Xs + jl;
#define MMI_R_MMA(AV, qb)                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                 \
        if ((qb) + u < q1) {                                                                        \
            const i32x4 o = ko[((qb) + u) * 2];                                                     \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc1 = mmi_mfma_f32_32x32x2(AV[u][e], xl[o[e]], acc1); \
        }                                                                                           \
    }
    {
        const int G = (q1 + U - 1) / U;
        int g = 0;
        for (; g + 2 < G; g += 2) {
            MMI_R_LOAD(avB, (g + 1) * U);
            MMI_R_MMA(avA, g * U);
            MMI_R_LOAD(avA, (g + 2) * U);
            MMI_R_MMA(avB, (g + 1) * U);
        }
        if (G - g == 2) {
            MMI_R_LOAD(avB, (g + 1) * U);
            MMI_R_MMA(avA, g * U);
            MMI_R_MMA(avB, (g + 1) * U);
        } else {
            MMI_R_MMA(avA, g * U);
        }
    }
#undef MMI_R_LOAD
#undef MMI_R_MMA
    // ---- hand-off: h = ELU(acc1 + b1) in the accumulator layout (rows past Cout1 have zero weights on both sides)
    f32x4 hb[MT1][4];
    {
        float h[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = mmi_elu(acc1[r] + b1v[r]);
        if constexpr (MT1 == 1) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) hb[0][c4] = f32x4{h[4 * c4], h[4 * c4 + 1], h[4 * c4 + 2], h[4 * c4 + 3]};
        } else {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
                *reinterpret_cast<f32x4*>(hs + ((wave * 4 + c4) * 64 + lane) * 4) = f32x4{h[4 * c4], h[4 * c4 + 1], h[4 * c4 + 2], h[4 * c4 + 3]};
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT1; ++m)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) hb[m][c4] = *reinterpret_cast<const f32x4*>(hs + ((m * 4 + c4) * 64 + lane) * 4);
        }
    }
    // ---- stage 2: output m-tiles wave, wave + MT1, ...; the epilogue is this block's own (bias, + x, ELU'd store and / or twin),
    // with none of the generic epilogue's per-value branches on launch-uniform flags: these kernels are instruction bound
    const int t = t0 + jl;
    const float* resb = c.res + (long)b * c.Cout * c.res_ld + c.res_off + t;
    float* outb = c.out + (long)b * c.Cout * c.out_ld + c.out_off + t;
    float* out2b = c.out2 ? c.out2 + (long)b * c.Cout * c.out_ld + c.out_off + t : nullptr;
    for (int mt2 = wave; mt2 < c.Mt; mt2 += MT1) {
        {   // the next tile's fragments are requested before this tile's MFMAs and epilogue
            const f32x4* w2 = reinterpret_cast<const f32x4*>(c.wpk) + (long)min(mt2 + MT1, c.Mt - 1) * c.Q * 64 + lane;
#pragma unroll
            for (int q = 0; q < Q2; ++q) w2n[q] = w2[(long)min(q, c.Q - 1) * 64];
        }
        // this tile's bias and residual values travel under the MFMAs (rows clamped into the tensor, masked at the store)
        const int row0 = mt2 * 32 + 4 * kh;
        float b2v[16], rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(row0 + (r & 3) + 8 * (r >> 2), c.Cout - 1);
            b2v[r] = c.bias ? c.bias[row] : 0.f;
            rv[r] = resb[(long)row * c.res_ld];
        }
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            if (q < c.Q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qq = q * 4 + e;                 // k-slot: hidden tile qq / 16, accumulator register qq % 16
                    acc2 = mmi_mfma_f32_32x32x2(w2v[q][e], hb[qq >> 4][(qq & 15) >> 2][qq & 3], acc2);
                }
            }
        }
        {
            const bool full = mt2 * 32 + 32 <= c.Cout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (!full && row >= c.Cout) continue;
                const float x = rv[r] + (acc2[r] + b2v[r]);             // x_orig + (conv + bias)  (seanet.py:76-77)
                const long at = (long)row * c.out_ld;
                if (out2b) { outb[at] = x; out2b[at] = mmi_elu(x); }
                else outb[at] = c.elu_out ? mmi_elu(x) : x;
            }
        }
#pragma unroll
        for (int q = 0; q < Q2; ++q) w2v[q] = w2n[q];
    }
}

// natural [B][Cin][x_ld] -> packed B operand: im2col (K, S), ELU, replicate padding of the first frame.
// One thread per (n-subtile, k-quad, lane): 4 elements kd = (q*4+e)*2 + (lane>>5), column n = nt*32 + (lane&31).
__global__ void k_pack_b_f32(ConvGemmArgs a, float* __restrict__ bp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ntn = (a.Ntot + 31) / 32;
    if (idx >= (long)ntn * a.Q * 64) return;
    const int lane = (int)(idx & 63);
    const int q = (int)((idx >> 6) % a.Q);
    const int nt = (int)((idx >> 6) / a.Q);
    const int n = nt * 32 + (lane & 31), kh = lane >> 5;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    {   // columns past Ntot / reduction indices past Cin*K read a clamped (valid) element: they meet zero weights or are
        // discarded, and unconditional loads are not serialised by the compiler
        const int nc = n < a.Ntot ? n : 0;
        const int b = mmi_fast_div(nc, a.T_magic), t = nc - b * a.T_out;
        const float* xb = a.x + (long)b * a.x_bstride + a.x_off + t * a.S;
        const bool rep = a.first != nullptr && a.first[b] != 0 && a.exec[b] != 0;
        const int hist = a.H - a.x_off;  // columns (relative to x_off) that belong to the history
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kd = (q * 4 + e) * 2 + kh;
            const int ci0 = mmi_fast_div(kd, a.K_magic), k = kd - ci0 * a.K;
            const int ci = ci0 < a.Cin ? ci0 : a.Cin - 1;
            int p = k;
            if (rep && (t * a.S + p) < hist) p = hist - t * a.S;  // replicate x[..., :1]
            float v = xb[(long)ci * a.x_ld + p];
            if (a.elu_in) v = mmi_elu(v);
            o[e] = v;
        }
    }
    reinterpret_cast<f32x4*>(bp)[idx] = o;
}

// split-K reduction over the workgroup's waves (fixed order) + the epilogue of k_gemm_f32 / k_gemm_f32_ln
template <int NSUB, int WAVES>
__device__ __forceinline__ void mmi_gemm_f32_finish(const ConvGemmArgs& a, f32x16 (&acc)[NSUB], int mt, int s0, int wave, int lane) {
    constexpr int NE = NSUB * 16 * 64;
    MMI_SHARED float red[WAVES * NE];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * NE + (s * 16 + r) * 64 + lane] = acc[s][r];
    __syncthreads();
    constexpr int NV = NE / (WAVES * 64);
    int co[NV], nn[NV];
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = (int)threadIdx.x + i * WAVES * 64;
        const int le = e & 63, r = (e >> 6) & 15, s = e >> 10;
        nn[i] = (s0 + s) * 32 + (le & 31);
        co[i] = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (le >> 5);
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) x += red[w * NE + e];
        v[i] = x;
    }
    if (a.out_mode == MMI_GOUT_PARTIAL) {
#pragma unroll
        for (int i = 0; i < NV; ++i) a.partial[((long)blockIdx.y * a.Mt * 32 + co[i]) * a.Npad + nn[i]] = v[i];   // padded rows / columns included
        return;
    }
    if (a.out_mode == MMI_GOUT_PACKED) {
        float bia[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) bia[i] = a.bias ? a.bias[co[i] < a.Cout ? co[i] : 0] : 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (nn[i] >= a.Ntot || co[i] >= a.Cout) continue;
            float x = v[i];
            if (a.bias) x += bia[i];
            if (a.act_out == MMI_ACT_GELU) x = mmi_gelu_erf(x);
            else if (a.act_out == MMI_ACT_ELU) x = mmi_elu(x);
            a.outp[mmi_bp_index(co[i], nn[i], a.outQ)] = x;
        }
        return;
    }
    int bb[NV], tt[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ok[i] = nn[i] < a.Ntot && co[i] < a.Cout;
        const int n3 = ok[i] ? nn[i] : 0;
        bb[i] = mmi_fast_div(n3, a.T_magic);
        tt[i] = n3 - bb[i] * a.T_out;
    }
    mmi_conv_store_n<NV>(a, co, bb, tt, ok, v);
}

// grid (Mt, ksplit, n-subtile groups); a workgroup covers NSUB n-subtiles of 32 columns starting at blockIdx.z*NSUB; its
// WAVES waves split the workgroup's k-quads.  The fp32 MFMA takes 64 cycles, so these GEMMs are matrix-core-latency
// bound unless they are spread over many waves: small weight matrices give each n-subtile its own workgroup.
template <int NSUB, int WAVES, int U>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_f32(ConvGemmArgs a) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int mt = blockIdx.x;
    const int qb_per = (a.Q + (int)gridDim.y - 1) / (int)gridDim.y;
    const int qb0 = min(a.Q, (int)blockIdx.y * qb_per), qb1 = min(a.Q, qb0 + qb_per);
    const int qper = (qb1 - qb0 + WAVES - 1) / WAVES;
    const int q0 = min(qb1, qb0 + wave * qper);
    const int nq = min(qb1, q0 + qper) - q0;
    const int ntn = (a.Ntot + 31) / 32;
    const int s0 = (int)blockIdx.z * NSUB;

    f32x16 acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk) + ((long)mt * a.Q + q0) * 64 + lane;
    const f32x4* bp[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) bp[s] = reinterpret_cast<const f32x4*>(a.bp) + ((long)min(s0 + s, ntn - 1) * a.Q + q0) * 64 + lane;

    f32x4 wA[U], bA[U][NSUB], wB[U], bB[U][NSUB];
#define MMI_F_LOAD(W_, B_, base)                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                           \
        W_[u] = mmi_load_nt(wp + ((base) + u) * 64);                                          \
        _Pragma("unroll") for (int s = 0; s < NSUB; ++s) B_[u][s] = bp[s][((base) + u) * 64]; \
    }
#define MMI_F_MMA(W_, B_)                                                                     \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                         \
            _Pragma("unroll") for (int s = 0; s < NSUB; ++s) acc[s] = mmi_mfma_f32_32x32x2(W_[u][e], B_[u][s][e], acc[s]);
    const int nfull = nq / U;
    if (nfull > 0) {
        MMI_F_LOAD(wA, bA, 0);
        int g = 0;
        for (; g + 2 < nfull; g += 2) {
            MMI_F_LOAD(wB, bB, (g + 1) * U);
            MMI_F_MMA(wA, bA);
            MMI_F_LOAD(wA, bA, (g + 2) * U);
            MMI_F_MMA(wB, bB);
        }
        if (nfull - g == 2) {
            MMI_F_LOAD(wB, bB, (g + 1) * U);
            MMI_F_MMA(wA, bA);
            MMI_F_MMA(wB, bB);
        } else {
            MMI_F_MMA(wA, bA);
        }
    }
    for (int q = nfull * U; q < nq; ++q) {
        wA[0] = mmi_load_nt(wp + q * 64);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) bA[0][s] = bp[s][q * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < NSUB; ++s) acc[s] = mmi_mfma_f32_32x32x2(wA[0][e], bA[0][s][e], acc[s]);
    }
#undef MMI_F_LOAD
#undef MMI_F_MMA

    mmi_gemm_f32_finish<NSUB, WAVES>(a, acc, mt, s0, wave, lane);
}

// LayerNorm fused in front of a Mimi transformer linear (norm1 -> in_proj, norm2 -> linear1; transformer.py:125-126,
// 752-776): grid (Mt, 1, n-subtiles), one 32-column subtile per workgroup; the 8 waves split the Cin / 8 <= 8 * KQ k-quads and
// hold their whole slice - raw residual-stream fragments, weight fragments, the norm's weight / bias - in registers (everything
// in flight at once: these GEMMs are latency bound).  Column statistics: per-wave partial sums of the lane's column, the two
// lane halves folded by one shuffle, the 8 waves through LDS; mean first, then sum (x - mean)^2 (two-pass, like
// k_layernorm_ct), y = (x - mean) * rsqrt(var + eps) * w + b, then the MFMAs.  Needs Q * 8 == Cin (no padded channels).
template <int WAVES, int KQ>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_f32_ln(ConvGemmArgs a) {
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int mt = blockIdx.x, s0 = blockIdx.z;
    const int qper = (a.Q + WAVES - 1) / WAVES;            // <= KQ (checked by the planner)
    const int q0 = min(a.Q, wave * qper);
    const int nq = min(a.Q, q0 + qper) - q0;
    const int kh = lane >> 5;
    const int ql = min(q0, a.Q - 1);
    f32x4 wv[KQ], xv[KQ], gv[KQ], bv[KQ];
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk) + ((long)mt * a.Q + ql) * 64 + lane;
    const f32x4* bp = reinterpret_cast<const f32x4*>(a.bp) + ((long)s0 * a.Q + ql) * 64 + lane;
    const f32x4* gp = reinterpret_cast<const f32x4*>(a.ln_w) + (long)ql * 2 + kh;
    const f32x4* hp = reinterpret_cast<const f32x4*>(a.ln_b) + (long)ql * 2 + kh;
#pragma unroll
    for (int u = 0; u < KQ; ++u) {                         // unconditional loads from clamped (valid) quads, masked afterwards
        const int uu = min(u, nq > 0 ? nq - 1 : 0);
        xv[u] = bp[(long)uu * 64];
        wv[u] = mmi_load_nt(wp + (long)uu * 64);
        gv[u] = gp[(long)uu * 2];
        bv[u] = hp[(long)uu * 2];
    }
    MMI_SHARED float stat[2][WAVES][32];
    float s1 = 0.f;
#pragma unroll
    for (int u = 0; u < KQ; ++u)
        if (u < nq) s1 += (xv[u][0] + xv[u][1]) + (xv[u][2] + xv[u][3]);
    s1 += mmi_shfl_xor(s1, 32);
    if (lane < 32) stat[0][wave][lane] = s1;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) tot += stat[0][w][lane & 31];
    const float mean = tot / (float)a.Cin;
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < KQ; ++u)
        if (u < nq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = xv[u][e] - mean; s2 += dlt * dlt; }
        }
    s2 += mmi_shfl_xor(s2, 32);
    if (lane < 32) stat[1][wave][lane] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) var += stat[1][w][lane & 31];
    const float rstd = mmi_rsqrtf(var / (float)a.Cin + a.ln_eps);
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
#pragma unroll
    for (int u = 0; u < KQ; ++u)
        if (u < nq) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (xv[u][e] - mean) * rstd * gv[u][e] + bv[u][e];
                acc[0] = mmi_mfma_f32_32x32x2(wv[u][e], y, acc[0]);
            }
        }
    mmi_gemm_f32_finish<1, WAVES>(a, acc, mt, s0, wave, lane);
}

// LayerNorm weight / bias [C] -> the element order of the packed operand: out[(q * 2 + kh) * 4 + e] = v[(q * 4 + e) * 2 + kh]
__global__ void k_pack_ln(const float* __restrict__ v, float* __restrict__ out, int C) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= C) return;
    const int e = i & 3, kh = (i >> 2) & 1, q = i >> 3;
    out[i] = v[(q * 4 + e) * 2 + kh];
}

// sum of the split-K partials + epilogue -> [B][Cout][out_ld]
__global__ void k_conv_finish(ConvGemmArgs a, int ksplit) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.Cout * a.Ntot) return;
    const int n = (int)(idx % a.Ntot), co = (int)(idx / a.Ntot);
    float pv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) pv[s] = a.partial[((long)min(s, ksplit - 1) * a.Mt * 32 + co) * a.Npad + n];   // ksplit <= 8, all in flight
    float x = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) x += s < ksplit ? pv[s] : 0.f;
    const int bq = mmi_fast_div(n, a.T_magic);
    int co1[1] = {co}, b1[1] = {bq}, t1[1] = {n - bq * a.T_out};
    bool ok1[1] = {true};
    float v1[1] = {x};
    mmi_conv_store_n<1>(a, co1, b1, t1, ok1, v1);
}

// ------------------------------------------------------------------------------------------------
// transposed conv: overlap-add of the GEMM result with the streaming `partial` (conv.py:340-362)
// tmp [B][Cout*K][tmp_ld >= T_in] holds tmp[b][co*K + k][t] = sum_ci Wtr[ci][co][k] * elu(x[b][ci][t]); K == 2*S.
// ------------------------------------------------------------------------------------------------
// out2: the ELU'd twin of `out` (same geometry) read by the next conv, or null (see ConvGemmArgs::out2)
__global__ void k_convtr_combine(const float* __restrict__ tmp, const float* __restrict__ bias,
                                 float* __restrict__ partial, const uint8_t* __restrict__ exec, float* __restrict__ out,
                                 int out_ld, int out_off, int B, int Cout, int K, int S, int T_in, float* __restrict__ out2,
                                 int tmp_ld) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Tout = T_in * S;
    if (idx >= (long)B * Cout * Tout) return;
    int p = (int)(idx % Tout);
    long row = idx / Tout;  // b*Cout + co
    int co = (int)(row % Cout);
    int b = (int)(row / Cout);
    int t = p / S, r = p - t * S;
    const float* trow = tmp + row * (long)K * tmp_ld;  // [K][T_in], row stride tmp_ld
    float v = trow[(long)r * tmp_ld + t];
    if (t > 0) v += trow[(long)(r + S) * tmp_ld + (t - 1)];
    if (bias) v += bias[co];
    if (t == 0) {
        // y[..., :PT] += partial; then the thread that consumed partial[b][co][r] also refreshes it
        long pi = row * (long)(K - S) + r;
        v += partial[pi];
        if (exec[b]) partial[pi] = trow[(long)(r + S) * tmp_ld + (T_in - 1)];  // tail, bias excluded (conv.py:354-360)
    }
    out[row * (long)out_ld + out_off + p] = v;
    if (out2) out2[row * (long)out_ld + out_off + p] = mmi_elu(v);
}

// depthwise (groups == C) transposed conv, K == 2*S, no bias: ConvTrUpsample1d (resample.py:68-119)
__global__ void k_upsample_dw(const float* __restrict__ x, int x_ld, int x_off, const float* __restrict__ w,
                              float* __restrict__ partial, const uint8_t* __restrict__ exec, float* __restrict__ out,
                              int out_ld, int out_off, int B, int C, int K, int S, int T_in) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Tout = T_in * S;
    if (idx >= (long)B * C * Tout) return;
    int p = (int)(idx % Tout);
    long row = idx / Tout;
    int c = (int)(row % C);
    int b = (int)(row / C);
    int t = p / S, r = p - t * S;
    const float* xr = x + row * (long)x_ld + x_off;
    float v = w[c * K + r] * xr[t];
    if (t > 0) v += w[c * K + r + S] * xr[t - 1];
    if (t == 0) {
        long pi = row * (long)(K - S) + r;
        v += partial[pi];
        if (exec[b]) partial[pi] = w[c * K + r + S] * xr[T_in - 1];
    }
    out[row * (long)out_ld + out_off + p] = v;
}

// ------------------------------------------------------------------------------------------------
// end-of-frame state commit: history shift, transformer offsets, replicate flags
// ------------------------------------------------------------------------------------------------
struct HistDesc {
    float* p;       // buffer [B][C][ld]
    int C, ld, H, T;
    int row_begin;  // prefix of B*C rows over the descriptor table
};

__global__ void k_commit_history(const HistDesc* __restrict__ descs, int ndesc, int total_rows,
                                 const uint8_t* __restrict__ exec) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid >= total_rows) return;
    int d = 0;
    while (d + 1 < ndesc && descs[d + 1].row_begin <= gid) ++d;
    HistDesc h = descs[d];
    int row = gid - h.row_begin;
    int b = row / h.C;
    if (!exec[b]) return;
    float* r = h.p + (long)row * h.ld;
    for (int p = 0; p < h.H; ++p) r[p] = r[p + h.T];  // ascending: source p+T is always ahead of the write
}

// Both commits of a step in ONE launch, with the descriptor table passed by value (kernel arguments: scalar loads) instead of
// searched through global memory by every thread: threads [0, total_rows) shift the histories, the next B advance the counters.
#define MMI_HIST_MAX 16
struct HistTable {
    HistDesc d[MMI_HIST_MAX];
    int n;
};
__global__ void k_commit_all(HistTable tab, int total_rows, const uint8_t* __restrict__ exec, long* __restrict__ counters,
                             int n_counters, int inc, uint8_t* __restrict__ first, int B) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid >= total_rows) {
        gid -= total_rows;
        if (gid >= B || !exec[gid]) return;
        for (int i = 0; i < n_counters; ++i) counters[(long)i * B + gid] += inc;
        if (first) first[gid] = 0;
        return;
    }
    int d = 0;
#pragma unroll
    for (int i = 1; i < MMI_HIST_MAX; ++i)
        if (i < tab.n && tab.d[i].row_begin <= gid) d = i;
    const HistDesc h = tab.d[d];
    int row = gid - h.row_begin;
    int b = row / h.C;
    if (!exec[b]) return;
    float* r = h.p + (long)row * h.ld;
    for (int p = 0; p < h.H; ++p) r[p] = r[p + h.T];  // ascending: source p+T is always ahead of the write
}

// offsets[i][b] += inc where exec; first[b] = 0 where exec
__global__ void k_commit_counters(long* __restrict__ counters, int n_counters, int inc, uint8_t* __restrict__ first,
                                  const uint8_t* __restrict__ exec, int B) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid >= B) return;
    if (!exec[gid]) return;
    for (int i = 0; i < n_counters; ++i) counters[(long)i * B + gid] += inc;
    if (first) first[gid] = 0;
}

__global__ void k_reset_history(const HistDesc* __restrict__ descs, int ndesc, int total_rows,
                                const uint8_t* __restrict__ mask) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid >= total_rows) return;
    int d = 0;
    while (d + 1 < ndesc && descs[d + 1].row_begin <= gid) ++d;
    HistDesc h = descs[d];
    int row = gid - h.row_begin;
    int b = row / h.C;
    if (mask && !mask[b]) return;
    float* r = h.p + (long)row * h.ld;
    for (int p = 0; p < h.H; ++p) r[p] = 0.f;
}

// generic masked fill of per-row state: buf[b][0..n) = 0 where mask
__global__ void k_reset_rows_f32(float* __restrict__ buf, long n_per_row, int B, const uint8_t* __restrict__ mask) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * n_per_row) return;
    int b = (int)(idx / n_per_row);
    if (mask && !mask[b]) return;
    buf[idx] = 0.f;
}

// counters -> 0, first -> 1, exec -> 1 for the reset rows (streaming.py:43-44, conv.py:166-169, transformer.py:329-334)
__global__ void k_reset_counters(long* __restrict__ counters, int n_counters, uint8_t* __restrict__ first,
                                 uint8_t* __restrict__ exec, int B, const uint8_t* __restrict__ mask) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid >= B) return;
    if (mask && !mask[gid]) return;
    for (int i = 0; i < n_counters; ++i) counters[(long)i * B + gid] = 0;
    if (first) first[gid] = 1;
    exec[gid] = 1;
}

__global__ void k_set_mask(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int B) {
    int gid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (gid < B) dst[gid] = src[gid] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// Mimi transformer pieces ([B][C][T] layout)
// ------------------------------------------------------------------------------------------------
// nn.LayerNorm(C, eps) over the channel axis of x[b][:, off + t]  (transformer.py:125-126)
__global__ __launch_bounds__(64) void k_layernorm_ct(const float* __restrict__ x, int x_ld, int x_off,
                                                     const float* __restrict__ w, const float* __restrict__ bvec,
                                                     float* __restrict__ y, int y_ld, int y_off, int C, int T,
                                                     float eps, float* __restrict__ yp, int yQ) {
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int lane = threadIdx.x;
    const float* xc = x + (long)b * C * x_ld + x_off + t;
    constexpr int NR = 16;                 // channels per lane held in registers (C <= 1024), one round of loads
    float xv[NR], wv[NR], bv[NR];          // weight and bias ride along with the input: one memory round trip in all
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + i * 64;
        const int cc = c < C ? c : 0;
        xv[i] = xc[(long)cc * x_ld];
        wv[i] = w[cc];
        bv[i] = bvec[cc];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) s += (lane + i * 64) < C ? xv[i] : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += mmi_shfl_xor(s, m);
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const float d = xv[i] - mean;
        v += (lane + i * 64) < C ? d * d : 0.f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += mmi_shfl_xor(v, m);
    const float rstd = mmi_rsqrtf(v / (float)C + eps);
    float* yc = y + (long)b * C * y_ld + y_off + t;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = lane + i * 64;
        if (c < C) {
            const float o = (xv[i] - mean) * rstd * wv[i] + bv[i];
            if (yp) yp[mmi_bp_index(c, blockIdx.x, yQ)] = o;   // packed B operand of the consuming linear (column n = b*T + t)
            else yc[(long)c * y_ld] = o;
        }
    }
}

// RoPE (rope.py:11-82, interleaved) on q,k + ring-KV write (transformer.py:236-253) + masked attention over the
// valid part of the ring (transformer.py:574-585).  One block per (b, head); T queries (T <= 8).
//   qkv  [B][3*H*D][T]  rows: q = h*D + d, k = H*D + h*D + d, v = 2*H*D + h*D + d   (transformer.py:557-559)
//   kc/vc [B][H][cap][D];  offsets[b] = tokens seen so far (== RingKVCache.end_offset == MHA offset)
//   out  [B][H*D][T]
struct MimiAttnArgs {
    const float* qkv;
    float* kc;
    float* vc;
    const long* offsets;
    float* out;
    float* outp;   // when set: packed B operand of out_proj (feature h*D+d, column b*T+t) instead of `out`
    int outQ;
    int B, H, D, T, cap, context;
    float max_period;
};

// 256 threads; D/4 lanes share a ring row (one 16-byte load each), so a wave instruction reads 64/(D/4) consecutive
// rows.  Only the min(offset + T, cap) valid slots are visited (the reference masks all `cap`).  Head dim D and the
// steps per frame T are compile-time so that the query slices live in registers and every loop unrolls.
// Dynamic LDS: qs[T][D] | sc[T][cap] | red[(256/(D/4)) * T * D].
template <int D, int T>
__global__ __launch_bounds__(256) void k_mimi_attn(MimiAttnArgs a) {
    constexpr int LPR = D / 4;             // lanes per row
    constexpr int RPB = 256 / LPR;         // rows per block pass
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cap = a.cap;
    MMI_DYN_SHARED(float, sm);
    float* qs = sm;               // [T][D] roped queries
    float* sc = qs + T * D;       // [T][cap] scores / probabilities
    float* red = sc + ((T * cap + 3) & ~3);   // [RPB][T][D] partial outputs (16-byte aligned: written as f32x4)
    const long off = a.offsets[b];
    const int HD = a.H * D;
    const float* qrow = a.qkv + (long)b * 3 * HD * T;
    float* kcb = a.kc + ((long)b * a.H + h) * cap * D;
    float* vcb = a.vc + ((long)b * a.H + h) * cap * D;

    // phase 1: rope(q), rope(k) -> ring, v -> ring (written unconditionally, transformer.py:243-250)
    for (int i = tid; i < T * (D / 2); i += 256) {
        const int t = i / (D / 2), j = i % (D / 2);
        const float freq = expf((float)j * (-logf(a.max_period) * 2.0f / (float)D));
        const float ts = (float)(off + t);
        const float ang = freq * ts;
        const float c = cosf(ang), s = sinf(ang);
        const float qr = qrow[(long)(h * D + 2 * j) * T + t], qi = qrow[(long)(h * D + 2 * j + 1) * T + t];
        const float kr = qrow[(long)(HD + h * D + 2 * j) * T + t], ki = qrow[(long)(HD + h * D + 2 * j + 1) * T + t];
        float q_re, q_im, k_re, k_im;          // (plain VALU rotations: see mmi_rope_rotate)
        mmi_rope_rotate(qr, qi, c, s, q_re, q_im);
        mmi_rope_rotate(kr, ki, c, s, k_re, k_im);
        qs[t * D + 2 * j] = q_re;
        qs[t * D + 2 * j + 1] = q_im;
        const int slot = (int)((off + t) % cap);
        kcb[(long)slot * D + 2 * j] = k_re;
        kcb[(long)slot * D + 2 * j + 1] = k_im;
    }
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, d = i % D;
        const int slot = (int)((off + t) % cap);
        vcb[(long)slot * D + d] = qrow[(long)(2 * HD + h * D + d) * T + t];
    }
    __syncthreads();

    // phase 2: scores over the valid slots.  positions of ring slots after this call's write (transformer.py:258-286)
    const long last = off + T - 1;
    const int end_index = (int)(last % cap);
    const long end_new = off + T;
    const int L = (int)(end_new < (long)cap ? end_new : (long)cap);
    const float scale = 1.0f / sqrtf((float)D);
    const int seg = tid % LPR, rsub = tid / LPR;
    f32x4 qv[T];
#pragma unroll
    for (int t = 0; t < T; ++t) qv[t] = *reinterpret_cast<const f32x4*>(qs + t * D + seg * 4);
    constexpr int NP = 8;                             // ring rows in flight per thread (unconditional, clamped loads)
    for (int s0 = 0; s0 < L; s0 += NP * RPB) {        // block-uniform trip count
        f32x4 kk[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) kk[i] = *reinterpret_cast<const f32x4*>(kcb + (long)min(s0 + i * RPB + rsub, L - 1) * D + seg * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int slot = s0 + i * RPB + rsub;
            const int delta = slot - end_index;
            const long pos = delta <= 0 ? last + delta : last + delta - cap;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float dv = (qv[t][0] * kk[i][0] + qv[t][1] * kk[i][1]) + (qv[t][2] * kk[i][2] + qv[t][3] * kk[i][3]);
#pragma unroll
                for (int m = LPR / 2; m >= 1; m >>= 1) dv += mmi_shfl_xor(dv, m);
                if (seg == 0 && slot < L) {
                    const long dq = (off + t) - pos;
                    const bool ok = pos >= 0 && dq >= 0 && dq < a.context;
                    sc[t * cap + slot] = ok ? dv * scale : -INFINITY;
                }
            }
        }
    }
    __syncthreads();

    // phase 3: softmax per query over [0, L): wave t owns query t (T <= 4 waves), so the reductions are wave shuffles
    if (wave < T) {
        const int t = wave;
        float m = -INFINITY;
        for (int slot = lane; slot < L; slot += 64) m = fmaxf(m, sc[t * cap + slot]);
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) m = fmaxf(m, mmi_shfl_xor(m, x));
        float sum = 0.f;
        for (int slot = lane; slot < L; slot += 64) {
            const float e = expf(sc[t * cap + slot] - m);
            sc[t * cap + slot] = e;
            sum += e;
        }
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) sum += mmi_shfl_xor(sum, x);
        const float inv = 1.0f / sum;
        for (int slot = lane; slot < L; slot += 64) sc[t * cap + slot] *= inv;
    }
    __syncthreads();

    // phase 4: out[t][d] = sum_slot p[t][slot] * V[slot][d]
    float acc[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    for (int s0 = 0; s0 < L; s0 += NP * RPB) {
        f32x4 vv[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) vv[i] = *reinterpret_cast<const f32x4*>(vcb + (long)min(s0 + i * RPB + rsub, L - 1) * D + seg * 4);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int slot = s0 + i * RPB + rsub;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float pr = slot < L ? sc[t * cap + slot] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][e] += pr * vv[i][e];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
        *reinterpret_cast<f32x4*>(red + (rsub * T + t) * D + seg * 4) = f32x4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
    __syncthreads();
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, d = i % D;
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < RPB; ++r) s += red[(r * T + t) * D + d];
        if (a.outp) a.outp[mmi_bp_index(h * D + d, b * T + t, a.outQ)] = s;
        else a.out[((long)b * HD + h * D + d) * T + t] = s;
    }
}

// Same operation in ONE memory round trip, for rings of at most 16 * (256 / (D/4)) slots (Mimi: 250 slots of 64 floats):
// every thread first issues its 16 key and 16 value row segments (unconditional, clamped loads: 128 KiB per workgroup in
// flight), ropes the new tokens while they travel, and then keeps scores, exponentials and the weighted value sum in
// registers.  Each wave normalises against its own maximum; the four waves are merged through LDS with the usual
// exp(m_wave - m) rescale.  The slots written by this call are taken from LDS instead of the ring (the ring loads were
// issued before the write).  Static LDS: qs | kn | vn [T][D], red[4][T][D], mw/sw [4][T].
template <int D, int T>
__global__ __launch_bounds__(256) void k_mimi_attn_1pass(MimiAttnArgs a) {
    constexpr int LPR = D / 4;             // lanes per ring row
    constexpr int RPB = 256 / LPR;         // rows per block pass
    constexpr int IT = 16;                 // passes: cap <= IT * RPB
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cap = a.cap;
    MMI_SHARED __attribute__((aligned(16))) float qs[T * D];
    MMI_SHARED __attribute__((aligned(16))) float kn[T * D];
    MMI_SHARED __attribute__((aligned(16))) float vn[T * D];
    MMI_SHARED __attribute__((aligned(16))) float red[4 * T * D];
    MMI_SHARED float mw[4 * T];
    MMI_SHARED float sw[4 * T];
    const long off = a.offsets[b];
    const int HD = a.H * D;
    const float* qrow = a.qkv + (long)b * 3 * HD * T;
    float* kcb = a.kc + ((long)b * a.H + h) * cap * D;
    float* vcb = a.vc + ((long)b * a.H + h) * cap * D;
    const long last = off + T - 1;
    const int end_index = (int)(last % cap);
    const long end_new = off + T;
    const int L = (int)(end_new < (long)cap ? end_new : (long)cap);
    const int seg = tid % LPR, rsub = tid / LPR;

    f32x4 kk[IT], vv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const long r = (long)min(i * RPB + rsub, L - 1) * D + seg * 4;
        kk[i] = *reinterpret_cast<const f32x4*>(kcb + r);
        vv[i] = *reinterpret_cast<const f32x4*>(vcb + r);
    }

    // rope(q), rope(k) -> LDS and ring, v -> LDS and ring (written unconditionally, transformer.py:243-250)
    for (int i = tid; i < T * (D / 2); i += 256) {
        const int t = i / (D / 2), j = i % (D / 2);
        const float freq = expf((float)j * (-logf(a.max_period) * 2.0f / (float)D));
        const float ang = freq * (float)(off + t);
        const float c = cosf(ang), s = sinf(ang);
        const float qr = qrow[(long)(h * D + 2 * j) * T + t], qi = qrow[(long)(h * D + 2 * j + 1) * T + t];
        const float kr = qrow[(long)(HD + h * D + 2 * j) * T + t], ki = qrow[(long)(HD + h * D + 2 * j + 1) * T + t];
        mmi_rope_rotate(qr, qi, c, s, qs[t * D + 2 * j], qs[t * D + 2 * j + 1]);      // (plain VALU rotations: see mmi_rope_rotate)
        mmi_rope_rotate(kr, ki, c, s, kn[t * D + 2 * j], kn[t * D + 2 * j + 1]);
    }
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, d = i % D;
        vn[i] = qrow[(long)(2 * HD + h * D + d) * T + t];
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, d = i % D;
        const long slot = (off + t) % cap;
        kcb[slot * D + d] = kn[i];
        vcb[slot * D + d] = vn[i];
    }

    const float scale = 1.0f / sqrtf((float)D);
    f32x4 qv[T];
    int newslot[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        qv[t] = *reinterpret_cast<const f32x4*>(qs + t * D + seg * 4);
        newslot[t] = (int)((off + t) % cap);
    }
    float sc[T][IT];
    float mx[T];
#pragma unroll
    for (int t = 0; t < T; ++t) mx[t] = -INFINITY;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int slot = i * RPB + rsub;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (slot == newslot[t]) {      // a later token of this call overwrites an earlier one when T > cap never happens (T <= 2 <= cap)
                kk[i] = *reinterpret_cast<const f32x4*>(kn + t * D + seg * 4);
                vv[i] = *reinterpret_cast<const f32x4*>(vn + t * D + seg * 4);
            }
        const int delta = slot - end_index;
        const long pos = delta <= 0 ? last + delta : last + delta - cap;     // position held by the slot (transformer.py:258-286)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float dv = (qv[t][0] * kk[i][0] + qv[t][1] * kk[i][1]) + (qv[t][2] * kk[i][2] + qv[t][3] * kk[i][3]);
#pragma unroll
            for (int m = LPR / 2; m >= 1; m >>= 1) dv += mmi_shfl_xor(dv, m);
            const long dq = (off + t) - pos;
            const bool ok = slot < L && pos >= 0 && dq >= 0 && dq < a.context;
            sc[t][i] = ok ? dv * scale : -INFINITY;
            mx[t] = fmaxf(mx[t], sc[t][i]);
        }
    }
    float acc[T][4], sum[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) mx[t] = fmaxf(mx[t], mmi_shfl_xor(mx[t], m));    // the wave's maximum
        const float mref = mx[t] == -INFINITY ? 0.f : mx[t];
        sum[t] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const float p = expf(sc[t][i] - mref);
            sum[t] += p;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] += p * vv[i][e];
        }
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) {
            sum[t] += mmi_shfl_xor(sum[t], m);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] += mmi_shfl_xor(acc[t][e], m);
        }
        if (lane < LPR) {
            *reinterpret_cast<f32x4*>(red + (wave * T + t) * D + seg * 4) = f32x4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            if (lane == 0) { mw[wave * T + t] = mx[t]; sw[wave * T + t] = sum[t]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < T * D; i += 256) {
        const int t = i / D, d = i % D;
        float m = mw[t];
#pragma unroll
        for (int w = 1; w < 4; ++w) m = fmaxf(m, mw[w * T + t]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = expf(mw[w * T + t] - m);      // 0 for a wave whose slots were all masked
            num += f * red[(w * T + t) * D + d];
            den += f * sw[w * T + t];
        }
        const float o = num / den;
        if (a.outp) a.outp[mmi_bp_index(h * D + d, b * T + t, a.outQ)] = o;
        else a.out[((long)b * HD + h * D + d) * T + t] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// residual vector quantiser (core_vq.py:270-287,507-528; vq.py:126-151,269-287)
// ------------------------------------------------------------------------------------------------
// Distances are evaluated in fp64 (||e||^2 - 2 x.e; the ||x||^2 term is constant per row), i.e. the exact
// nearest centroid of the fp32 inputs, lowest index on ties - the answer the reference's fp32 cdist
// agrees with except at ~1e-7-relative near-ties.
#define MMI_RVQ_CHUNK 32
// grid (chunks of 32 codes, groups of 8 rows); 256 threads: thread (c, g) owns code c0+c and row rg0+g.  The chunk of the
// codebook and the 8 rows are staged in LDS with 16-byte accesses (row stride D+4 floats keeps ds_read_b128 conflict
// free); the dot product runs in four independent fp64 chains.  D % 4 == 0.
// one level's operands; a launch carries two of them (grid.z / grid.y picks): the semantic quantiser's only level and the acoustic
// quantiser's first level work on different halves of the projected latent and do not depend on each other
// (vq.py:269-287 SplitResidualVectorQuantizer.encode), so they share their launches
struct RvqLevel {
    float* x;              // residual rows [Bn][x_rstride], updated in place by k_rvq_select
    const float* E;        // codebook [bins][D]
    const double* e2;      // ||e||^2
    double* best_d;        // [nchunk][Bn]
    int* best_i;
    int level;             // column of `codes` this level writes
};

__global__ __launch_bounds__(256) void k_rvq_dist(RvqLevel l0, RvqLevel l1, int x_rstride, int Bn, int D, int bins) {
    const RvqLevel& L = blockIdx.z == 0 ? l0 : l1;
    const float* __restrict__ x = L.x;
    const float* __restrict__ E = L.E;
    const double* __restrict__ e2 = L.e2;
    double* __restrict__ best_d = L.best_d;
    int* __restrict__ best_i = L.best_i;
    MMI_DYN_SHARED(float, sm);
    const int ldE = D + 4;
    float* Es = sm;                          // [CHUNK][D+4]
    float* xs = sm + MMI_RVQ_CHUNK * ldE;    // [8][D]
    const int chunk = blockIdx.x;
    const int c0 = chunk * MMI_RVQ_CHUNK;
    const int tid = threadIdx.x;
    const int rg0 = (int)blockIdx.y * 8;     // this block's 8 rows (grid.y = ceil(Bn / 8))
    const int D4 = D >> 2;
    {   // 8 threads per code / row, each copying every 8th float4 of it
        const int r = tid >> 3, j0 = tid & 7;
        const int code = min(c0 + r, bins - 1);
        for (int j = j0; j < D4; j += 8)
            *reinterpret_cast<f32x4*>(Es + r * ldE + 4 * j) = *reinterpret_cast<const f32x4*>(E + (long)code * D + 4 * j);
        if (r < 8) {
            const int row = min(rg0 + r, Bn - 1);
            for (int j = j0; j < D4; j += 8)
                *reinterpret_cast<f32x4*>(xs + r * D + 4 * j) = *reinterpret_cast<const f32x4*>(x + (long)row * x_rstride + 4 * j);
        }
    }
    __syncthreads();
    const int c = tid & 31, g = tid >> 5;  // 8 rows, 32 codes
    const bool cvalid = (c0 + c) < bins;
    const double en = cvalid ? e2[c0 + c] : 0.0;
    const int r = rg0 + g;
    const bool rvalid = r < Bn;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = 0; j < D4; ++j) {
        const f32x4 ev = *reinterpret_cast<const f32x4*>(Es + c * ldE + 4 * j);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + g * D + 4 * j);
        a0 += (double)ev[0] * (double)xv[0];
        a1 += (double)ev[1] * (double)xv[1];
        a2 += (double)ev[2] * (double)xv[2];
        a3 += (double)ev[3] * (double)xv[3];
    }
    const double acc = (a0 + a1) + (a2 + a3);
    double dist = (cvalid && rvalid) ? en - 2.0 * acc : INFINITY;
    int idx = c0 + c;
    // argmin over the 32 codes held by this half-wave (xor masks < 32 stay inside it)
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        double od = mmi_shfl_xor(dist, m);
        int oi = mmi_shfl_xor(idx, m);
        if (od < dist || (od == dist && oi < idx)) { dist = od; idx = oi; }
    }
    if (c == 0 && rvalid) { best_d[(long)chunk * Bn + r] = dist; best_i[(long)chunk * Bn + r] = idx; }
}

// pick the winning chunk per row, emit the code, subtract the centroid from the residual
__global__ __launch_bounds__(256) void k_rvq_select(RvqLevel l0, RvqLevel l1, int nchunk, int x_rstride, int* __restrict__ codes,
                                                    int codes_rstride, int Bn, int D) {
    const RvqLevel& L = blockIdx.y == 0 ? l0 : l1;
    const double* __restrict__ best_d = L.best_d;
    const int* __restrict__ best_i = L.best_i;
    float* __restrict__ x = L.x;
    const float* __restrict__ E = L.E;
    const int level = L.level;
    const int r = blockIdx.x;
    MMI_SHARED int widx;
    if (threadIdx.x < 64) {   // first wave: strided scan of the chunk winners, then a wave argmin (lowest index on ties)
        double bd = INFINITY;
        int bi = 0x7fffffff;
        for (int ch = threadIdx.x; ch < nchunk; ch += 64) {
            double d = best_d[(long)ch * Bn + r];
            int i = best_i[(long)ch * Bn + r];
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            double od = mmi_shfl_xor(bd, m);
            int oi = mmi_shfl_xor(bi, m);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        if (threadIdx.x == 0) {
            widx = bi;
            codes[(long)r * codes_rstride + level] = bi;
        }
    }
    __syncthreads();
    const int idx = widx;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long)r * x_rstride + d] -= E[(long)idx * D + d];
}

// decode: q[b][0:D] = E_0[c_0]; q[b][D:2D] = ((0 + E_1[c_1]) + E_2[c_2]) + ...   (core_vq.py:521-528)
// n_codes_dev (optional): the number of codebooks of THIS call, read from device memory so that the captured decoder program
// serves `decode` calls with any K <= n_q (the split RVQ decodes however many codebooks it is given, vq.py:281-287)
__global__ void k_rvq_gather(const int* __restrict__ codes, int codes_rstride, int n_codes,
                             const float* __restrict__ Eall, int bins, int D, int n_sem, float* __restrict__ q, int Bn,
                             const int* __restrict__ n_codes_dev, float* __restrict__ qp, int qpQ) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= Bn * D) return;
    if (n_codes_dev) n_codes = *n_codes_dev;
    int r = idx / D, d = idx % D;
    float first = 0.f, rest = 0.f;
    for (int k = 0; k < n_codes; ++k) {
        int c = codes[(long)r * codes_rstride + k];
        c = c < 0 ? 0 : (c >= bins ? bins - 1 : c);  // the reference crashes on OOB codes (vq.py:144-146); we clamp
        float e = Eall[((long)k * bins + c) * D + d];
        if (k < n_sem) first += e; else rest += e;
    }
    q[(long)r * 2 * D + d] = first;
    q[(long)r * 2 * D + D + d] = rest;
    if (qp) {   // also as the packed B operand of the output projection (column = row r): no k_pack_b_f32 launch in between
        qp[mmi_bp_index(d, r, qpQ)] = first;
        qp[mmi_bp_index(D + d, r, qpQ)] = rest;
    }
}

// int64 <-> int32 code tensors at the ABI boundary: [B][K][F] i64 <-> per-frame [B][K] i32
__global__ void k_codes_out(const int* __restrict__ src, int src_rstride, long* __restrict__ dst, int B, int K, int F,
                            int f) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * K) return;
    int b = idx / K, k = idx % K;
    dst[((long)b * K + k) * F + f] = (long)src[(long)b * src_rstride + k];
}
// src: i64 [B][K][F] with `src_bstride` elements between batch rows (K * F when dense; larger for a column slice of a wider
// tensor such as LMGen's [B][1 + dep_q][1] step output read from its second column on)
__global__ void k_codes_in(const long* __restrict__ src, long src_bstride, int* __restrict__ dst, int dst_rstride, int B, int K,
                           int F, int f) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * K) return;
    int b = idx / K, k = idx % K;
    dst[(long)b * dst_rstride + k] = (int)src[(long)b * src_bstride + (long)k * F + f];
}
