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
// RVQ projections, the transformer's linears) runs through ONE implicit-GEMM kernel, k_conv_gemm:
//   out[co][n=(b,t)] = sum_{kd=(ci,k)} W[co][kd] * act(in[b][ci][t*S + k])
// with the fp32 MFMA (an exact fma chain, so results track the fp32 reference to rounding order).
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
enum { MMI_ACT_NONE = 0, MMI_ACT_GELU = 1 };

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
    int Mt, Q;            // M tiles of 32, packed k-quads (Kdim_pad / 8)
    int Ntot;             // B * T_out
    int elu_in;           // apply ELU(alpha=1) to every loaded input (seanet.py:63,205,222)
    int act_out;          // MMI_ACT_*
};

__device__ __forceinline__ float mmi_elu(float v) { return v > 0.f ? v : expm1f(v); }
__device__ __forceinline__ float mmi_gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int NT, int KSPLIT>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemmArgs a) {
    constexpr int TPB = 4 / KSPLIT;
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int jl = lane & 31, kh = lane >> 5;
    const int Ntiles = (a.Ntot + 32 * NT - 1) / (32 * NT);
    int tile = (int)blockIdx.x * TPB + wave / KSPLIT;
    const bool tile_ok = tile < a.Mt * Ntiles;
    if (!tile_ok) tile = 0;  // keep every lane alive for the MFMAs / barriers; results are discarded
    const int ks = wave % KSPLIT;
    const int mt = tile % a.Mt, nt = tile / a.Mt;

    const float* xb[NT];
    bool nvalid[NT], rep[NT];
    int ob[NT], ot[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        int n = (nt * NT + s) * 32 + jl;
        nvalid[s] = tile_ok && n < a.Ntot;
        int nn = nvalid[s] ? n : 0;
        int b = nn / a.T_out, t = nn - b * a.T_out;
        ob[s] = b; ot[s] = t;
        xb[s] = a.x + (long)b * a.x_bstride + a.x_off + t * a.S;
        rep[s] = a.first != nullptr && a.first[b] != 0 && a.exec[b] != 0;
    }
    const int qper = (a.Q + KSPLIT - 1) / KSPLIT;
    const int q0 = ks * qper;
    const int q1 = min(a.Q, q0 + qper);

    f32x16 acc[NT];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // (ci, k) of this lane's reduction index kd = q*8 + 2*e + kh, advanced incrementally
    int kd = q0 * 8 + kh;
    int ci = kd / a.K, k = kd - ci * a.K;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpk) + ((long)mt * a.Q + q0) * 64 + lane;
    const int hist = a.H - a.x_off;  // columns (relative to x_off) that belong to the history
    for (int q = q0; q < q1; ++q) {
        f32x4 av = mmi_load_nt(wp);
        wp += 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float bv[NT];
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                float v = 0.f;
                if (nvalid[s] && ci < a.Cin) {
                    int p = k;                         // column relative to xb (= x_off + t*S)
                    if (rep[s] && (ot[s] * a.S + p) < hist) p = hist - ot[s] * a.S;  // replicate x[..., :1]
                    v = xb[s][(long)ci * a.x_ld + p];
                    if (a.elu_in) v = mmi_elu(v);
                }
                bv[s] = v;
            }
#pragma unroll
            for (int s = 0; s < NT; ++s) acc[s] = mmi_mfma_f32_32x32x2(av[e], bv[s], acc[s]);
            k += 2;
            while (k >= a.K) { k -= a.K; ++ci; }
        }
    }

    if (KSPLIT > 1) {
        MMI_SHARED float red[4 * NT * 16 * 64];
        if (ks > 0) {
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * NT + s) * 16 + r) * 64 + lane] = acc[s][r];
        }
        __syncthreads();
        if (ks > 0) return;
        for (int w2 = 1; w2 < KSPLIT; ++w2)
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][r] += red[(((wave + w2) * NT + s) * 16 + r) * 64 + lane];
    }

#pragma unroll
    for (int s = 0; s < NT; ++s) {
        if (!nvalid[s]) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co >= a.Cout) continue;
            float v = acc[s][r];
            if (a.bias) v += a.bias[co];
            if (a.act_out == MMI_ACT_GELU) v = mmi_gelu_erf(v);
            if (a.scale) v *= a.scale[co];
            long row = (long)ob[s] * a.Cout + co;
            if (a.res) v = a.res[row * a.res_ld + a.res_off + ot[s]] + v;
            a.out[row * a.out_ld + a.out_off + ot[s]] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// transposed conv: overlap-add of the GEMM result with the streaming `partial` (conv.py:340-362)
// tmp [B][Cout*K][T_in] holds tmp[b][co*K + k][t] = sum_ci Wtr[ci][co][k] * elu(x[b][ci][t]); K == 2*S.
// ------------------------------------------------------------------------------------------------
__global__ void k_convtr_combine(const float* __restrict__ tmp, const float* __restrict__ bias,
                                 float* __restrict__ partial, const uint8_t* __restrict__ exec, float* __restrict__ out,
                                 int out_ld, int out_off, int B, int Cout, int K, int S, int T_in) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Tout = T_in * S;
    if (idx >= (long)B * Cout * Tout) return;
    int p = (int)(idx % Tout);
    long row = idx / Tout;  // b*Cout + co
    int co = (int)(row % Cout);
    int b = (int)(row / Cout);
    int t = p / S, r = p - t * S;
    const float* trow = tmp + row * (long)K * T_in;  // [K][T_in]
    float v = trow[(long)r * T_in + t];
    if (t > 0) v += trow[(long)(r + S) * T_in + (t - 1)];
    if (bias) v += bias[co];
    if (t == 0) {
        // y[..., :PT] += partial; then the thread that consumed partial[b][co][r] also refreshes it
        long pi = row * (long)(K - S) + r;
        v += partial[pi];
        if (exec[b]) partial[pi] = trow[(long)(r + S) * T_in + (T_in - 1)];  // tail, bias excluded (conv.py:354-360)
    }
    out[row * (long)out_ld + out_off + p] = v;
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
                                                     float eps) {
    const int b = blockIdx.x / T, t = blockIdx.x % T;
    const int lane = threadIdx.x;
    const float* xc = x + (long)b * C * x_ld + x_off + t;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xc[(long)c * x_ld];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += mmi_shfl_xor(s, m);
    const float mean = s / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        float d = xc[(long)c * x_ld] - mean;
        v += d * d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += mmi_shfl_xor(v, m);
    const float rstd = mmi_rsqrtf(v / (float)C + eps);
    float* yc = y + (long)b * C * y_ld + y_off + t;
    for (int c = lane; c < C; c += 64) yc[(long)c * y_ld] = (xc[(long)c * x_ld] - mean) * rstd * w[c] + bvec[c];
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
    int B, H, D, T, cap, context;
    float max_period;
};

__global__ __launch_bounds__(256) void k_mimi_attn(MimiAttnArgs a) {
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int D = a.D, T = a.T, cap = a.cap;
    MMI_DYN_SHARED(float, sm);
    float* qs = sm;               // [T][D] roped queries
    float* sc = qs + T * D;       // [T][cap] scores / probabilities
    float* red = sc + T * cap;    // [T][256] reduction scratch... sized [max(T,1)*256]
    const long off = a.offsets[b];
    const int HD = a.H * D;
    const float* qrow = a.qkv + (long)b * 3 * HD * T;
    float* kcb = a.kc + ((long)b * a.H + h) * cap * D;
    float* vcb = a.vc + ((long)b * a.H + h) * cap * D;

    // phase 1: rope(q), rope(k) -> ring, v -> ring (written unconditionally, transformer.py:243-250)
    for (int i = tid; i < T * (D / 2); i += nth) {
        int t = i / (D / 2), j = i % (D / 2);
        float freq = expf((float)j * (-logf(a.max_period) * 2.0f / (float)D));
        float ts = (float)(off + t);
        float ang = freq * ts;
        float c = cosf(ang), s = sinf(ang);
        float qr = qrow[(long)(h * D + 2 * j) * T + t], qi = qrow[(long)(h * D + 2 * j + 1) * T + t];
        float kr = qrow[(long)(HD + h * D + 2 * j) * T + t], ki = qrow[(long)(HD + h * D + 2 * j + 1) * T + t];
        qs[t * D + 2 * j] = qr * c - qi * s;
        qs[t * D + 2 * j + 1] = qr * s + qi * c;
        int slot = (int)((off + t) % cap);
        kcb[(long)slot * D + 2 * j] = kr * c - ki * s;
        kcb[(long)slot * D + 2 * j + 1] = kr * s + ki * c;
    }
    for (int i = tid; i < T * D; i += nth) {
        int t = i / D, d = i % D;
        int slot = (int)((off + t) % cap);
        vcb[(long)slot * D + d] = qrow[(long)(2 * HD + h * D + d) * T + t];
    }
    __syncthreads();

    // phase 2: scores.  positions of ring slots after this call's write (transformer.py:258-286)
    const long last = off + T - 1;
    const int end_index = (int)(last % cap);
    const long end_new = off + T;
    const float scale = 1.0f / sqrtf((float)D);
    for (int slot = tid; slot < cap; slot += nth) {
        int delta = slot - end_index;
        long pos = delta <= 0 ? last + delta : last + delta - cap;
        if ((long)slot >= end_new) pos = -1;
        const float* kr = kcb + (long)slot * D;
        for (int t = 0; t < T; ++t) {
            long dq = (off + t) - pos;
            bool ok = pos >= 0 && dq >= 0 && dq < a.context;
            float s = -INFINITY;
            if (ok) {
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc += qs[t * D + d] * kr[d];
                s = acc * scale;
            }
            sc[t * cap + slot] = s;
        }
    }
    __syncthreads();

    // phase 3: softmax per query (block reductions through `red`)
    for (int t = 0; t < T; ++t) {
        float m = -INFINITY;
        for (int slot = tid; slot < cap; slot += nth) m = fmaxf(m, sc[t * cap + slot]);
        red[tid] = m;
        __syncthreads();
        for (int s2 = nth / 2; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] = fmaxf(red[tid], red[tid + s2]);
            __syncthreads();
        }
        m = red[0];
        __syncthreads();
        float sum = 0.f;
        for (int slot = tid; slot < cap; slot += nth) {
            float e = expf(sc[t * cap + slot] - m);
            sc[t * cap + slot] = e;
            sum += e;
        }
        red[tid] = sum;
        __syncthreads();
        for (int s2 = nth / 2; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        float inv = 1.0f / red[0];
        __syncthreads();
        for (int slot = tid; slot < cap; slot += nth) sc[t * cap + slot] *= inv;
    }
    __syncthreads();

    // phase 4: out[t][d] = sum_slot p[t][slot] * V[slot][d]; thread groups split the slots, then reduce
    const int groups = nth / D > 0 ? nth / D : 1;   // nth is a multiple of D for the configs we run
    const int d = tid % D, g = tid / D;
    for (int t = 0; t < T; ++t) {
        float acc = 0.f;
        if (g < groups)
            for (int slot = g; slot < cap; slot += groups) acc += sc[t * cap + slot] * vcb[(long)slot * D + d];
        red[tid] = acc;
        __syncthreads();
        if (g == 0) {
            float s = red[d];
            for (int g2 = 1; g2 < groups; ++g2) s += red[g2 * D + d];
            a.out[((long)b * HD + h * D + d) * T + t] = s;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// residual vector quantiser (core_vq.py:270-287,507-528; vq.py:126-151,269-287)
// ------------------------------------------------------------------------------------------------
// Distances are evaluated in fp64 (||e||^2 - 2 x.e; the ||x||^2 term is constant per row), i.e. the exact
// nearest centroid of the fp32 inputs, lowest index on ties - the answer the reference's fp32 cdist
// agrees with except at ~1e-7-relative near-ties.
#define MMI_RVQ_CHUNK 32
__global__ __launch_bounds__(256) void k_rvq_dist(const float* __restrict__ x, int x_rstride,
                                                  const float* __restrict__ E, const double* __restrict__ e2,
                                                  double* __restrict__ best_d, int* __restrict__ best_i, int Bn, int D,
                                                  int bins) {
    MMI_DYN_SHARED(float, sm);
    const int ldE = D + 1;
    float* Es = sm;                          // [CHUNK][D+1]
    float* xs = sm + MMI_RVQ_CHUNK * ldE;    // [Bn][D]
    const int chunk = blockIdx.x;
    const int c0 = chunk * MMI_RVQ_CHUNK;
    const int tid = threadIdx.x;
    for (int i = tid; i < MMI_RVQ_CHUNK * D; i += 256) {
        int c = i / D, d = i % D;
        Es[c * ldE + d] = (c0 + c) < bins ? E[(long)(c0 + c) * D + d] : 0.f;
    }
    for (int i = tid; i < Bn * D; i += 256) {
        int r = i / D, d = i % D;
        xs[r * D + d] = x[(long)r * x_rstride + d];
    }
    __syncthreads();
    const int c = tid & 31, g = tid >> 5;  // 8 row groups, 32 codes
    const bool cvalid = (c0 + c) < bins;
    const double en = cvalid ? e2[c0 + c] : 0.0;
    for (int r0 = 0; r0 < Bn; r0 += 8) {   // uniform trip count: both half-waves take part in the shuffles
        const int r = r0 + g;
        const bool rvalid = r < Bn;
        double acc = 0.0;
        if (rvalid)
            for (int d = 0; d < D; ++d) acc += (double)Es[c * ldE + d] * (double)xs[r * D + d];
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
}

// pick the winning chunk per row, emit the code, subtract the centroid from the residual
__global__ __launch_bounds__(256) void k_rvq_select(const double* __restrict__ best_d, const int* __restrict__ best_i,
                                                    int nchunk, float* __restrict__ x, int x_rstride,
                                                    const float* __restrict__ E, int* __restrict__ codes,
                                                    int codes_rstride, int level, int Bn, int D) {
    const int r = blockIdx.x;
    MMI_SHARED int widx;
    if (threadIdx.x == 0) {
        double bd = best_d[r];
        int bi = best_i[r];
        for (int ch = 1; ch < nchunk; ++ch) {
            double d = best_d[(long)ch * Bn + r];
            int i = best_i[(long)ch * Bn + r];
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
        widx = bi;
        codes[(long)r * codes_rstride + level] = bi;
    }
    __syncthreads();
    const int idx = widx;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(long)r * x_rstride + d] -= E[(long)idx * D + d];
}

// decode: q[b][0:D] = E_0[c_0]; q[b][D:2D] = ((0 + E_1[c_1]) + E_2[c_2]) + ...   (core_vq.py:521-528)
__global__ void k_rvq_gather(const int* __restrict__ codes, int codes_rstride, int n_codes,
                             const float* __restrict__ Eall, int bins, int D, int n_sem, float* __restrict__ q, int Bn) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= Bn * D) return;
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
}

// int64 <-> int32 code tensors at the ABI boundary: [B][K][F] i64 <-> per-frame [B][K] i32
__global__ void k_codes_out(const int* __restrict__ src, int src_rstride, long* __restrict__ dst, int B, int K, int F,
                            int f) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * K) return;
    int b = idx / K, k = idx % K;
    dst[((long)b * K + k) * F + f] = (long)src[(long)b * src_rstride + k];
}
__global__ void k_codes_in(const long* __restrict__ src, int* __restrict__ dst, int dst_rstride, int B, int K, int F,
                           int f) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= B * K) return;
    int b = idx / K, k = idx % K;
    dst[(long)b * dst_rstride + k] = (int)src[((long)b * K + k) * F + f];
}
