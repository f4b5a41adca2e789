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
//   activations     [B][features] bf16 row-major.
//   KV ring         [layer][2][B][H][cap][Dh] bf16 (row = one position of one head: 256 contiguous bytes at Dh=128).
//   token ring      [B][17][max_delay+2] int32 (lm.py:605-613).
#pragma once
#include "mmi_common.h"

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
// gate_hidden == 0: plain [N][K] matrix.  gate_hidden == H: rows [0,H) are gates, [H,2H) values; tile nt carries
// gate rows nt*TN/2 .. and, in its second half, the matching value rows.
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

// ------------------------------------------------------------------------------------------------
// skinny GEMM: out[b][n] = sum_k x[b][k] * W[n][k], B <= 16*MT (TN=16) or 32*MT (TN=32) rows, weights streamed once
// ------------------------------------------------------------------------------------------------
enum { MMI_EPI_STORE = 0, MMI_EPI_RESID = 1, MMI_EPI_GATE = 2, MMI_EPI_EMB = 3 };

struct GemmArgs {
    const u32x4* wp;        // packed weights
    const uint16_t* x;      // [B][K] bf16
    uint16_t* out;          // [B][out_ld] bf16
    const uint16_t* resid;  // EPI_RESID: [B][out_ld]
    const uint16_t* emb;    // EPI_EMB: embedding table [rows][N]
    const int* tok;         // EPI_EMB: token per row, tok[b * tok_stride]
    int tok_stride;
    int B, N, K, KSTEPS, out_ld;
    int epi;
};

template <int TN, int MT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_bf16(GemmArgs a) {
    constexpr int R = TN == 32 ? 16 : 4;          // accumulator registers per MFMA tile
    constexpr int BT = TN;                        // batch rows per MFMA tile
    constexpr int KS = TN == 32 ? 16 : 32;        // k per MFMA
    typedef float acc_t __attribute__((ext_vector_type(R)));
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    const int nt = blockIdx.x;
    const int il = TN == 32 ? (lane & 31) : (lane & 15);
    const int kq = TN == 32 ? (lane >> 5) : (lane >> 4);

    const uint16_t* xr[MT];
    bool xv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int b = mt * BT + il;
        xv[mt] = b < a.B;
        xr[mt] = a.x + (long)(xv[mt] ? b : 0) * a.K + 8 * kq;
    }
    const int kper = (a.KSTEPS + WAVES - 1) / WAVES;
    const int ks0 = wave * kper;
    const int ks1 = min(a.KSTEPS, ks0 + kper);

    acc_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[mt][r] = 0.f;

    const u32x4* wp = a.wp + ((long)nt * a.KSTEPS + ks0) * 64 + lane;
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll 4
    for (int ks = ks0; ks < ks1; ++ks) {
        u32x4 wv = mmi_load_nt(wp);
        wp += 64;
        const int k = ks * KS + 8 * kq;
        const bool kin = k < a.K;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x4 xf = zero;
            if (xv[mt] && kin) xf = *reinterpret_cast<const u32x4*>(xr[mt] + ks * KS);
            if constexpr (TN == 32) acc[mt] = mmi_mfma_bf16_32x32x16(wv, xf, acc[mt]);
            else acc[mt] = mmi_mfma_bf16_16x16x32(wv, xf, acc[mt]);
        }
    }

    // split-K reduction across the block's waves (fixed order -> deterministic), then the epilogue
    MMI_SHARED float red[WAVES * MT * R * 64];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < R; ++r) red[((wave * MT + mt) * R + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    constexpr int NE = MT * R * 64;
    for (int e = (int)threadIdx.x; e < NE; e += WAVES * 64) {
        const int le = e & 63;
        const int r = (e >> 6) % R;
        const int mt = (e >> 6) / R;
        int i, j;
        if (TN == 32) { i = (r & 3) + 8 * (r >> 2) + 4 * (le >> 5); j = le & 31; }
        else { i = 4 * (le >> 4) + r; j = le & 15; }
        const int b = mt * BT + j;
        if (b >= a.B) continue;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w * NE + e];
        if (a.epi == MMI_EPI_GATE) {
            if (i >= TN / 2) continue;
            const int n = nt * (TN / 2) + i;
            if (n >= a.N) continue;
            // partner (value) element: same lane, r+8 (TN=32) or lane+32 (TN=16)
            const int e2 = TN == 32 ? e + 8 * 64 : e + 32;
            float u = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) u += red[w * NE + e2];
            const float g = mmi_round_bf16(s);
            u = mmi_round_bf16(u);
            const float act = mmi_round_bf16(g / (1.0f + expf(-g)));     // F.silu on a bf16 tensor
            a.out[(long)b * a.out_ld + n] = mmi_f32_to_bf16(act * u);
            continue;
        }
        const int n = nt * TN + i;
        if (n >= a.N) continue;
        float v = mmi_round_bf16(s);                                      // nn.Linear output in bf16
        if (a.epi == MMI_EPI_RESID) {
            v = v + mmi_bf16_to_f32(a.resid[(long)b * a.out_ld + n]);     // x_orig + update
        } else if (a.epi == MMI_EPI_EMB) {
            int t = a.tok[(long)b * a.tok_stride];
            float ev = 0.f;
            if (t != -1) ev = mmi_bf16_to_f32(a.emb[(long)(t < 0 ? 0 : t) * a.N + n]);   // lm_utils.py:102-124
            v = v + ev;
        }
        a.out[(long)b * a.out_ld + n] = mmi_f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm (rms_norm_f32, eps 1e-8): y = (x.float() * (alpha.float() * rsqrt(eps + mean(x^2)))).to(bf16)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rmsnorm_bf16(const uint16_t* __restrict__ x, const uint16_t* __restrict__ alpha,
                                                      uint16_t* __restrict__ y, int D, float eps) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint16_t* xr = x + (long)b * D;
    float ss = 0.f;
    for (int i = tid * 8; i < D; i += 256 * 8) {
        u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float lo = mmi_bf16_to_f32((uint16_t)(v[q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(v[q] >> 16));
            ss += lo * lo;
            ss += hi * hi;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += mmi_shfl_xor(ss, m);
    MMI_SHARED float red[4];
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    const float rs = mmi_rsqrtf(eps + tot / (float)D);
    uint16_t* yr = y + (long)b * D;
    for (int i = tid * 8; i < D; i += 256 * 8) {
        u32x4 v = *reinterpret_cast<const u32x4*>(xr + i);
        u32x4 al = *reinterpret_cast<const u32x4*>(alpha + i);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float lo = mmi_bf16_to_f32((uint16_t)(v[q] & 0xffffu)), hi = mmi_bf16_to_f32((uint16_t)(v[q] >> 16));
            float alo = mmi_bf16_to_f32((uint16_t)(al[q] & 0xffffu)), ahi = mmi_bf16_to_f32((uint16_t)(al[q] >> 16));
            uint32_t olo = mmi_f32_to_bf16(lo * (alo * rs)), ohi = mmi_f32_to_bf16(hi * (ahi * rs));
            o[q] = olo | (ohi << 16);
        }
        *reinterpret_cast<u32x4*>(yr + i) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// input embedding sum (lm.py:388-397): ((emb0[t1] + emb1[t2]) + ...) + text_emb[t0], each add rounded to bf16
// ------------------------------------------------------------------------------------------------
__global__ void k_lm_embed(const int* __restrict__ tokens, int n_codebooks, const uint16_t* __restrict__ emb,
                           int card1, const uint16_t* __restrict__ text_emb, uint16_t* __restrict__ x, int D) {
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
    x[(long)b * D + d] = mmi_f32_to_bf16(acc);
}

// ------------------------------------------------------------------------------------------------
// temporal attention: RoPE + ring-KV write, split decode attention over the VALID part of the ring, combine
// ------------------------------------------------------------------------------------------------
struct LmAttnArgs {
    const uint16_t* qkv;   // [B][3*H*Dh]
    uint16_t* qrot;        // [B][H][Dh]
    uint16_t* kc;          // [B][H][cap][Dh]
    uint16_t* vc;
    const long* offsets;   // [B]
    float* opart;          // [B][H][NS][Dh]
    float* ml;             // [B][H][NS][2]
    uint16_t* out;         // [B][H*Dh]
    int B, H, Dh, cap, context, NS;
    float max_period;
};

// rope.py:11-82 (interleaved, fp32) for the single new position, then RingKVCache scatter (transformer.py:243-250)
__global__ void k_lm_rope_kv(LmAttnArgs a) {
    const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
    const int Dh = a.Dh, HD = a.H * Dh;
    const long off = a.offsets[b];
    const int slot = (int)(off % a.cap);
    const uint16_t* row = a.qkv + (long)b * 3 * HD;
    uint16_t* kdst = a.kc + (((long)b * a.H + h) * a.cap + slot) * Dh;
    uint16_t* vdst = a.vc + (((long)b * a.H + h) * a.cap + slot) * Dh;
    uint16_t* qdst = a.qrot + ((long)b * a.H + h) * Dh;
    for (int j = threadIdx.x; j < Dh / 2; j += blockDim.x) {
        float freq = expf((float)j * (-logf(a.max_period) * 2.0f / (float)Dh));
        float ang = freq * (float)off;
        float c = cosf(ang), s = sinf(ang);
        float qr = mmi_bf16_to_f32(row[h * Dh + 2 * j]), qi = mmi_bf16_to_f32(row[h * Dh + 2 * j + 1]);
        float kr = mmi_bf16_to_f32(row[HD + h * Dh + 2 * j]), ki = mmi_bf16_to_f32(row[HD + h * Dh + 2 * j + 1]);
        qdst[2 * j] = mmi_f32_to_bf16(qr * c - qi * s);
        qdst[2 * j + 1] = mmi_f32_to_bf16(qr * s + qi * c);
        kdst[2 * j] = mmi_f32_to_bf16(kr * c - ki * s);
        kdst[2 * j + 1] = mmi_f32_to_bf16(kr * s + ki * c);
        vdst[2 * j] = row[2 * HD + h * Dh + 2 * j];
        vdst[2 * j + 1] = row[2 * HD + h * Dh + 2 * j + 1];
    }
}

#define MMI_ATTN_CHUNK 256
// grid (B*H, NS); 256 threads.  Chunk c covers ring slots [c*256, c*256+256).  Each 16-byte load covers 8 dims of one
// key row; DH/8 lanes share a row, so one wave instruction reads 64/(DH/8) consecutive rows = 1 KiB contiguous.
template <int DH>
__global__ __launch_bounds__(256) void k_lm_attn_split(LmAttnArgs a) {
    constexpr int LPR = DH / 8;        // lanes per row
    constexpr int RPW = 64 / LPR;      // rows per wave instruction
    constexpr int CH = MMI_ATTN_CHUNK;
    const int bh = blockIdx.x, b = bh / a.H;
    const int chunk = blockIdx.y;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long off = a.offsets[b];
    const long end_new = off + 1;
    const int end_index = (int)(off % a.cap);
    const int L = (int)(end_new < (long)a.cap ? end_new : (long)a.cap);
    const int c0 = chunk * CH;
    float* mlp = a.ml + ((long)bh * a.NS + chunk) * 2;
    if (c0 >= L) {                      // nothing valid in this chunk (block-uniform)
        if (tid == 0) { mlp[0] = -INFINITY; mlp[1] = 0.f; }
        return;
    }
    MMI_SHARED float sc[CH];
    MMI_SHARED float wred[8];
    MMI_SHARED float ored[4 * DH];
    const int seg = lane % LPR, rsub = lane / LPR;
    float qv[8];
    {
        u32x4 qq = *reinterpret_cast<const u32x4*>(a.qrot + (long)bh * DH + seg * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            qv[2 * q] = mmi_bf16_to_f32((uint16_t)(qq[q] & 0xffffu));
            qv[2 * q + 1] = mmi_bf16_to_f32((uint16_t)(qq[q] >> 16));
        }
    }
    const uint16_t* kbase = a.kc + (long)bh * a.cap * DH;
    const uint16_t* vbase = a.vc + (long)bh * a.cap * DH;
    const float scale = 1.0f / sqrtf((float)DH);
    constexpr int PER_WAVE = CH / 4;
    // ---- scores
    for (int it = 0; it < PER_WAVE / RPW; ++it) {
        const int rl = wave * PER_WAVE + it * RPW + rsub;
        const int slot = c0 + rl;
        bool valid = slot < L;
        if (valid) {   // absolute position of the slot (transformer.py:258-286) and the causal/context mask (:574-582)
            int delta = slot - end_index;
            long pos = delta <= 0 ? off + delta : off + delta - a.cap;
            long dq = off - pos;
            valid = pos >= 0 && dq >= 0 && dq < a.context;
        }
        float dot = 0.f;
        if (valid) {
            u32x4 kk = *reinterpret_cast<const u32x4*>(kbase + (long)slot * DH + seg * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                dot += qv[2 * q] * mmi_bf16_to_f32((uint16_t)(kk[q] & 0xffffu));
                dot += qv[2 * q + 1] * mmi_bf16_to_f32((uint16_t)(kk[q] >> 16));
            }
        }
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) dot += mmi_shfl_xor(dot, m);
        if (seg == 0) sc[rl] = valid ? dot * scale : -INFINITY;
    }
    __syncthreads();
    // ---- chunk softmax statistics
    float s = sc[tid];
    float mx = s;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, mmi_shfl_xor(mx, m));
    if (lane == 0) wred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float p = (mx == -INFINITY) ? 0.f : expf(s - mx);
    float sum = p;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += mmi_shfl_xor(sum, m);
    if (lane == 0) wred[4 + wave] = sum;
    sc[tid] = p;
    __syncthreads();
    sum = (wred[4] + wred[5]) + (wred[6] + wred[7]);
    // ---- P.V
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int it = 0; it < PER_WAVE / RPW; ++it) {
        const int rl = wave * PER_WAVE + it * RPW + rsub;
        const int slot = c0 + rl;
        const float pr = sc[rl];
        if (pr != 0.f && slot < L) {
            u32x4 vv = *reinterpret_cast<const u32x4*>(vbase + (long)slot * DH + seg * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[2 * q] += pr * mmi_bf16_to_f32((uint16_t)(vv[q] & 0xffffu));
                acc[2 * q + 1] += pr * mmi_bf16_to_f32((uint16_t)(vv[q] >> 16));
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int m = 32; m >= LPR; m >>= 1) acc[e] += mmi_shfl_xor(acc[e], m);
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ored[wave * DH + seg * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < DH) {
        float o = (ored[tid] + ored[DH + tid]) + (ored[2 * DH + tid] + ored[3 * DH + tid]);
        a.opart[((long)bh * a.NS + chunk) * DH + tid] = o;
    }
    if (tid == 0) { mlp[0] = mx; mlp[1] = sum; }
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
        a.out[(long)bh * Dh + d] = mmi_f32_to_bf16(num / den);
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
    uint16_t* out;         // [B][H*Dh]
    int B, H, Dh, steps, k;
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
    float sc[16];
    float mx = -INFINITY;
    for (int j = 0; j <= a.k; ++j) {
        float kv = 0.f;
        if (on) kv = (j == a.k) ? mmi_bf16_to_f32(kn) : mmi_bf16_to_f32(kcb[(long)j * Dh + lane]);
        float d = q * kv;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) d += mmi_shfl_xor(d, m);
        sc[j] = d * scale;
        mx = fmaxf(mx, sc[j]);
    }
    float den = 0.f, o = 0.f;
    for (int j = 0; j <= a.k; ++j) {
        float p = expf(sc[j] - mx);
        den += p;
        float vv = 0.f;
        if (on) vv = (j == a.k) ? mmi_bf16_to_f32(vn) : mmi_bf16_to_f32(vcb[(long)j * Dh + lane]);
        o += p * vv;
    }
    if (on) a.out[(long)b * HD + h * Dh + lane] = mmi_f32_to_bf16(o / den);
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
};

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

template <int NT>
__global__ __launch_bounds__(NT) void k_sample(SampleArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint16_t* lg = a.logits + (long)b * a.ld;
    const int V = a.V;
    MMI_SHARED float redf[NT / 64];
    MMI_SHARED int redi[NT / 64];
    MMI_SHARED int hist[256];
    MMI_SHARED int wsum[NT / 64 + 1];
    MMI_SHARED float sel_val[256];
    MMI_SHARED int sel_idx[256];
    MMI_SHARED unsigned s_prefix;
    MMI_SHARED int s_want;
    const int lane = tid & 63, wave = tid >> 6;

    if (!a.use_sampling || !(a.temp > 0.f)) {
        // torch.argmax(logits): first maximum
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += NT) {
            float v = mmi_bf16_to_f32(lg[i]);
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
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
            a.out[(long)b * a.out_stride] = mmi_apply_forced(a, b, bi);
        }
        return;
    }

    // ---- softmax statistics of logits / temp (fp32)
    float mx = -INFINITY;
    for (int i = tid; i < V; i += NT) mx = fmaxf(mx, mmi_bf16_to_f32(lg[i]) / a.temp);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, mmi_shfl_xor(mx, m));
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) mx = fmaxf(mx, redf[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < V; i += NT) sum += expf(mmi_bf16_to_f32(lg[i]) / a.temp - mx);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += mmi_shfl_xor(sum, m);
    if (lane == 0) redf[wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NT / 64; ++w) sum += redf[w];
    __syncthreads();
#define MMI_PROB(i) (expf(mmi_bf16_to_f32(lg[i]) / a.temp - mx) / sum)

    // ---- radix select of the k-th largest probability (bit pattern of a non-negative float is monotonic)
    const int k = a.k < V ? a.k : V;
    if (tid == 0) { s_prefix = 0u; s_want = k; }
    __syncthreads();
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += NT) hist[i] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = tid; i < V; i += NT) {
            unsigned bits = __builtin_bit_cast(unsigned, MMI_PROB(i));
            bool match = shift == 24 ? true : ((bits >> (shift + 8)) == (prefix >> (shift + 8)));
            if (match) mmi_atomic_add(reinterpret_cast<unsigned*>(&hist[(bits >> shift) & 255u]), 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int want = s_want, bin = 255, cum = 0;
            for (; bin > 0; --bin) {
                if (cum + hist[bin] >= want) break;
                cum += hist[bin];
            }
            s_want = want - cum;
            s_prefix = prefix | ((unsigned)bin << shift);
        }
        __syncthreads();
    }
    const unsigned Tbits = s_prefix;   // bits of the k-th largest value
    const int want_eq = s_want;        // how many elements equal to it belong to the top-k (lowest indices first)

    // ---- ordered compaction of the top-k set (index order; ties at the threshold resolved by index)
    int n_sel = 0, n_eq = 0;
    for (int base = 0; base < V; base += NT) {
        const int i = base + tid;
        unsigned bits = 0;
        float pv = 0.f;
        if (i < V) { pv = MMI_PROB(i); bits = __builtin_bit_cast(unsigned, pv); }
        const int is_eq = (i < V && bits == Tbits) ? 1 : 0;
        int tot_eq;
        const int eq_pos = n_eq + mmi_block_excl_scan<NT>(is_eq, wsum, &tot_eq);
        const int take = (i < V && (bits > Tbits || (is_eq && eq_pos < want_eq))) ? 1 : 0;
        int tot_take;
        const int pos = n_sel + mmi_block_excl_scan<NT>(take, wsum, &tot_take);
        if (take && pos < 256) { sel_val[pos] = pv; sel_idx[pos] = i; }
        n_eq += tot_eq;
        n_sel += tot_take;
    }
    __syncthreads();
    // ---- rank inside the set (descending value, then index) and the noisy argmax
    float score = -INFINITY;
    int rank = 0x7fffffff, tok = 0;
    if (tid < k && tid < 256) {
        const float v = sel_val[tid];
        const int id = sel_idx[tid];
        int r = 0;
        for (int m = 0; m < k; ++m) {
            float ov = sel_val[m];
            r += (ov > v || (ov == v && sel_idx[m] < id)) ? 1 : 0;
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
    MMI_SHARED int redr[NT / 64];
    if (lane == 0) { redf[wave] = score; redi[wave] = tok; redr[wave] = rank; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (redf[w] > score || (redf[w] == score && redr[w] < rank)) { score = redf[w]; rank = redr[w]; tok = redi[w]; }
        a.out[(long)b * a.out_stride] = mmi_apply_forced(a, b, tok);
    }
#undef MMI_PROB
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
__global__ void k_lm_prepare(TokArgs t, const int* __restrict__ user, int n_user, int* __restrict__ tokens) {
    int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= t.B * t.NC) return;
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
