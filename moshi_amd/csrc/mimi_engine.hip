// Mimi codec engine behind the C ABI (include/moshi_mi.h): MimiModel.encode / decode in streaming mode
// (reference: moshi/moshi/models/compression.py:338-433) as a fixed per-frame launch list over the kernels
// of mimi_kernels.h, captured into one hipGraph per direction.
#include "mimi_kernels.h"
#include "mmi_graph.h"

#include <math.h>

namespace {

struct ConvW {          // one conv / linear as an implicit GEMM: rows = Cout, reduction = Cin*K
    float* wpk = nullptr;
    float* wpk_h = nullptr;   // 1x1 convs: the same matrix packed in the hidden-tile order of k_resblock's second stage
    float* bias = nullptr;
    int Cin = 0, Cout = 0, K = 1, S = 1, Mt = 0, Q = 0;
};

struct TrLayerW {
    ConvW in_proj, out_proj, lin1, lin2;
    float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr, *ls1 = nullptr, *ls2 = nullptr;
    float *n1w_pk = nullptr, *n1b_pk = nullptr, *n2w_pk = nullptr, *n2b_pk = nullptr;   // operand order, for k_gemm_f32_ln
};

struct Buf {            // activation buffer [B][C][ld]; first H columns = causal history of its consumer
    float* p = nullptr;
    int C = 0, ld = 0, H = 0;
    int lead = 0;          // floats in front of row 0 (alloc_buf: alignment of the new columns)
    // ELU hoisted into the producers (ConvGemmArgs::elu_out / out2, mimi_kernels.h):
    float* pe = nullptr;   // twin of the same geometry holding ELU(values) - what the ELU-fronted consumer conv reads
    bool elu = false;      // `p` itself holds ELU(values) (no reader wants the raw ones)
    float* conv_src() const { return pe ? pe : p; }                    // what an ELU-fronted conv reads
    bool needs_elu_at_load() const { return !pe && !elu; }
};

}  // namespace

struct mmi_mimi {
    int device = -1;                // HIP device the handle lives on (current at create); see MmiDeviceGuard
    int* dec_k_dev = nullptr;       // number of codebooks of the current decode call (read by the captured decoder program)
    int dec_k_host = 0, dec_k_cur = 0;
    mmi_mimi_cfg cfg;
    int max_batch = 0;
    int n_codebooks = 8;
    MmiArena wts;
    // SEANet
    std::vector<ConvW> enc_convs, dec_convs;   // in execution order
    std::vector<TrLayerW> enc_tr, dec_tr;
    ConvW downsample;
    float* upsample_w = nullptr;               // [C][2*stride]
    // RVQ
    ConvW q_in, q_out;                         // stacked first|rest projections
    float* E_all = nullptr;                    // [n_q][bins][D]  (0 = rvq_first.layers.0, 1.. = rvq_rest.layers.*)
    double* e2_all = nullptr;                  // [n_q][bins]
    // RVQ scratch (max_batch rows)
    float* xq = nullptr;                       // [maxB][2*Dq] residuals (first | rest)
    double* best_d = nullptr;
    int* best_i = nullptr;
    int* codes_i32 = nullptr;                  // [maxB][n_q]  encoder side: what the RVQ selected
    int* dec_codes_i32 = nullptr;              // [maxB][n_q]  decoder side: the codes being decoded (its own buffer: an encode of the
                                               //              next frame may run on another stream while a decode is in flight, duplex.hip)
    float* q2 = nullptr;                       // [maxB][2*Dq]
    float* lat_tmp = nullptr;                  // [maxB][dimension]
    float *qin_bp = nullptr, *qin_part = nullptr, *qout_bp = nullptr, *qout_part = nullptr;   // projection scratch (max_batch)
    int nchunk = 0;
    // streaming state
    bool streaming = false;
    int batch = 0;
    MmiArena st;
    uint8_t* exec = nullptr;
    uint8_t* first = nullptr;
    long* counters = nullptr;                  // [2][B]: encoder / decoder transformer offsets
    Buf enc_in, latent, dec_codes_lat, dec_out;
    float* enc_kv = nullptr;
    float* dec_kv = nullptr;
    std::vector<float*> partials;              // conv-transpose partial buffers
    std::vector<long> partial_sizes;           // elements per batch row
    HistDesc* enc_hist = nullptr;
    HistDesc* dec_hist = nullptr;
    int enc_nhist = 0, dec_nhist = 0, enc_hist_rows = 0, dec_hist_rows = 0;
    MmiProgram enc_prog, dec_prog;
    hipStream_t cap_stream = nullptr;
    bool use_graph = true;
};

namespace {

// ---- weight import ---------------------------------------------------------------------------
int need(const MmiWeights& W, const std::string& name, int ndim, const mmi_tensor_desc** out) {
    const mmi_tensor_desc* d = W.find(name);
    if (!d) return mmi_fail(MMI_ERR_MISSING_WEIGHT, "missing weight: " + name);
    if (d->dtype != MMI_F32) return mmi_fail(MMI_ERR_UNSUPPORTED, "Mimi weights must be fp32: " + name);
    if (d->ndim != ndim) return mmi_fail(MMI_ERR_SHAPE, "unexpected rank for " + name);
    *out = d;
    return MMI_OK;
}

int pack_matrix(mmi_mimi* m, const float* W, int M, int Kd, long sm, long sk, ConvW* cw) {
    cw->Mt = mmi_cdiv(M, 32);
    cw->Q = mmi_cdiv(Kd, 8);
    size_t n = (size_t)cw->Mt * cw->Q * 256;
    MMI_HIP_CHECK(m->wts.alloc(&cw->wpk, n));
    int blocks = (int)mmi_cdiv64((int64_t)n, 256);
    MMI_LAUNCH(k_pack_a_f32, blocks, 256, 0, (hipStream_t)0, W, cw->wpk, M, Kd, sm, sk, cw->Mt, cw->Q);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

int copy_vec(mmi_mimi* m, const mmi_tensor_desc* d, int n, float** out) {
    MMI_HIP_CHECK(m->wts.alloc(out, (size_t)n));
    MMI_HIP_CHECK(hipMemcpy(*out, d->data, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice));
    return MMI_OK;
}

// nn.Conv1d weight [Cout][Cin][K] (+ bias)
int load_conv(mmi_mimi* m, const MmiWeights& W, const std::string& prefix, int Cin, int Cout, int K, int S, bool bias,
              ConvW* cw) {
    const mmi_tensor_desc* w;
    int rc = need(W, prefix + ".weight", 3, &w);
    if (rc) return rc;
    if (w->shape[0] != Cout || w->shape[1] != Cin || w->shape[2] != K)
        return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + prefix + ".weight");
    cw->Cin = Cin; cw->Cout = Cout; cw->K = K; cw->S = S;
    rc = pack_matrix(m, (const float*)w->data, Cout, Cin * K, (long)Cin * K, 1, cw);
    if (rc) return rc;
    if (K == 1) {   // the second conv of a residual block (k_resblock reads its reduction index permuted)
        const size_t n = (size_t)cw->Mt * cw->Q * 256;
        MMI_HIP_CHECK(m->wts.alloc(&cw->wpk_h, n));
        MMI_LAUNCH(k_pack_a_f32_hperm, (int)mmi_cdiv64((int64_t)n, 256), 256, 0, (hipStream_t)0, (const float*)w->data, cw->wpk_h, Cout, Cin,
                   cw->Mt, cw->Q);
        MMI_CHECK_LAUNCH();
    }
    if (bias) {
        const mmi_tensor_desc* b;
        rc = need(W, prefix + ".bias", 1, &b);
        if (rc) return rc;
        if (b->shape[0] != Cout) return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + prefix + ".bias");
        rc = copy_vec(m, b, Cout, &cw->bias);
        if (rc) return rc;
    }
    return MMI_OK;
}

// nn.ConvTranspose1d weight [Cin][Cout][K] as a GEMM with rows (co,k): out rows = Cout*K, reduction = Cin
int load_convtr(mmi_mimi* m, const MmiWeights& W, const std::string& prefix, int Cin, int Cout, int K, int S,
                ConvW* cw) {
    const mmi_tensor_desc* w;
    int rc = need(W, prefix + ".weight", 3, &w);
    if (rc) return rc;
    if (w->shape[0] != Cin || w->shape[1] != Cout || w->shape[2] != K)
        return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + prefix + ".weight");
    if (K != 2 * S) return mmi_fail(MMI_ERR_UNSUPPORTED, "transposed conv needs kernel == 2*stride: " + prefix);
    cw->Cin = Cin; cw->Cout = Cout * K; cw->K = 1; cw->S = 1;
    rc = pack_matrix(m, (const float*)w->data, Cout * K, Cin, 1, (long)Cout * K, cw);
    if (rc) return rc;
    const mmi_tensor_desc* b;
    rc = need(W, prefix + ".bias", 1, &b);
    if (rc) return rc;
    return copy_vec(m, b, Cout, &cw->bias);
}

// nn.Linear weight [out][in], no bias
int load_linear(mmi_mimi* m, const MmiWeights& W, const std::string& name, int in, int out, ConvW* cw) {
    const mmi_tensor_desc* w;
    int rc = need(W, name, 2, &w);
    if (rc) return rc;
    if (w->shape[0] != out || w->shape[1] != in) return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + name);
    cw->Cin = in; cw->Cout = out; cw->K = 1; cw->S = 1;
    return pack_matrix(m, (const float*)w->data, out, in, in, 1, cw);
}

int load_vec(mmi_mimi* m, const MmiWeights& W, const std::string& name, int n, float** out) {
    const mmi_tensor_desc* d;
    int rc = need(W, name, 1, &d);
    if (rc) return rc;
    if (d->shape[0] != n) return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + name);
    return copy_vec(m, d, n, out);
}

int load_transformer(mmi_mimi* m, const MmiWeights& W, const std::string& prefix, std::vector<TrLayerW>* layers) {
    const mmi_mimi_cfg& c = m->cfg;
    const int d = c.tr_d_model, ff = c.tr_dim_feedforward;
    layers->resize(c.tr_num_layers);
    for (int l = 0; l < c.tr_num_layers; ++l) {
        TrLayerW& L = (*layers)[l];
        std::string p = prefix + ".transformer.layers." + std::to_string(l);
        int rc;
        if ((rc = load_linear(m, W, p + ".self_attn.in_projs.0.weight", d, 3 * d, &L.in_proj))) return rc;
        if ((rc = load_linear(m, W, p + ".self_attn.out_projs.0.weight", d, d, &L.out_proj))) return rc;
        if ((rc = load_linear(m, W, p + ".linear1.weight", d, ff, &L.lin1))) return rc;
        if ((rc = load_linear(m, W, p + ".linear2.weight", ff, d, &L.lin2))) return rc;
        if ((rc = load_vec(m, W, p + ".norm1.weight", d, &L.n1w))) return rc;
        if ((rc = load_vec(m, W, p + ".norm1.bias", d, &L.n1b))) return rc;
        if ((rc = load_vec(m, W, p + ".norm2.weight", d, &L.n2w))) return rc;
        if ((rc = load_vec(m, W, p + ".norm2.bias", d, &L.n2b))) return rc;
        if ((rc = load_vec(m, W, p + ".layer_scale_1.scale", d, &L.ls1))) return rc;
        if ((rc = load_vec(m, W, p + ".layer_scale_2.scale", d, &L.ls2))) return rc;
        if (d % 8 == 0) {               // the norms' weight / bias in the packed operand's element order (k_gemm_f32_ln)
            struct { const float* src; float** dst; } pk[] = {{L.n1w, &L.n1w_pk}, {L.n1b, &L.n1b_pk}, {L.n2w, &L.n2w_pk}, {L.n2b, &L.n2b_pk}};
            for (auto& e : pk) {
                MMI_HIP_CHECK(m->wts.alloc(e.dst, (size_t)d));
                MMI_LAUNCH(k_pack_ln, mmi_cdiv(d, 256), 256, 0, (hipStream_t)0, e.src, *e.dst, d);
            }
            MMI_CHECK_LAUNCH();
        }
    }
    return MMI_OK;
}

int load_rvq(mmi_mimi* m, const MmiWeights& W) {
    const mmi_mimi_cfg& c = m->cfg;
    const int D = c.q_dimension, dim = c.dimension, bins = c.q_bins, nq = c.q_n_q;
    // stacked input projection [2D][dim]: rows 0..D-1 = rvq_first.input_proj, D..2D-1 = rvq_rest.input_proj
    const mmi_tensor_desc *wf, *wr, *of, *orr;
    int rc;
    if ((rc = need(W, "quantizer.rvq_first.input_proj.weight", 3, &wf))) return rc;
    if ((rc = need(W, "quantizer.rvq_rest.input_proj.weight", 3, &wr))) return rc;
    if ((rc = need(W, "quantizer.rvq_first.output_proj.weight", 3, &of))) return rc;
    if ((rc = need(W, "quantizer.rvq_rest.output_proj.weight", 3, &orr))) return rc;
    if (wf->shape[0] != D || wf->shape[1] != dim || of->shape[0] != dim || of->shape[1] != D)
        return mmi_fail(MMI_ERR_SHAPE, "RVQ projection shape mismatch");
    float* tmp = nullptr;
    MMI_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)2 * D * dim * sizeof(float)));
    MMI_HIP_CHECK(hipMemcpy(tmp, wf->data, (size_t)D * dim * sizeof(float), hipMemcpyDeviceToDevice));
    MMI_HIP_CHECK(hipMemcpy(tmp + (size_t)D * dim, wr->data, (size_t)D * dim * sizeof(float), hipMemcpyDeviceToDevice));
    m->q_in.Cin = dim; m->q_in.Cout = 2 * D; m->q_in.K = 1; m->q_in.S = 1;
    rc = pack_matrix(m, tmp, 2 * D, dim, dim, 1, &m->q_in);
    if (rc) { hipFree(tmp); return rc; }
    MMI_HIP_CHECK(hipDeviceSynchronize());
    // stacked output projection [dim][2D]: columns 0..D-1 = first, D..2D-1 = rest
    int blocks = mmi_cdiv(dim * D, 256);
    MMI_LAUNCH(k_copy2d_f32, blocks, 256, 0, (hipStream_t)0, (const float*)of->data, (long)D, tmp, (long)2 * D, dim, D);
    MMI_LAUNCH(k_copy2d_f32, blocks, 256, 0, (hipStream_t)0, (const float*)orr->data, (long)D, tmp + D, (long)2 * D, dim, D);
    MMI_CHECK_LAUNCH();
    m->q_out.Cin = 2 * D; m->q_out.Cout = dim; m->q_out.K = 1; m->q_out.S = 1;
    rc = pack_matrix(m, tmp, dim, 2 * D, 2 * D, 1, &m->q_out);
    MMI_HIP_CHECK(hipDeviceSynchronize());
    hipFree(tmp);
    if (rc) return rc;

    MMI_HIP_CHECK(m->wts.alloc(&m->E_all, (size_t)nq * bins * D));
    MMI_HIP_CHECK(m->wts.alloc(&m->e2_all, (size_t)nq * bins));
    for (int k = 0; k < nq; ++k) {
        std::string p = k < c.q_n_q_semantic
                            ? "quantizer.rvq_first.vq.layers." + std::to_string(k) + "._codebook."
                            : "quantizer.rvq_rest.vq.layers." + std::to_string(k - c.q_n_q_semantic) + "._codebook.";
        const mmi_tensor_desc *es, *cu;
        if ((rc = need(W, p + "embedding_sum", 2, &es))) return rc;
        if ((rc = need(W, p + "cluster_usage", 1, &cu))) return rc;
        if (es->shape[0] != bins || es->shape[1] != D || cu->shape[0] != bins)
            return mmi_fail(MMI_ERR_SHAPE, "codebook shape mismatch: " + p);
        MMI_LAUNCH(k_codebook_prepare, bins, 64, 0, (hipStream_t)0, (const float*)es->data, (const float*)cu->data,
                   m->E_all + (size_t)k * bins * D, m->e2_all + (size_t)k * bins, bins, D, 1e-5f);
    }
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

// ---- launch helpers ---------------------------------------------------------------------------
// A conv / linear becomes one of two launch sequences, chosen when the frame program is built (mimi_kernels.h):
//   N = B*T_out > 128 columns : k_conv_wide
//   N <= 128                  : [k_pack_b_f32] -> k_gemm_f32 [-> k_conv_finish when K is split over workgroups]
struct ConvPlan {
    bool ln = false;                  // k_gemm_f32_ln: LayerNorm fused in front of the linear
    bool wide = false;
    int MTB = 1, W = 1, U = 1;        // wide
    int NSUB = 1, waves = 4, ksplit = 1, nz = 1;
    bool pack = false;
};

template <int MTB, int U, bool ELU_IN>
void launch_wide_e(hipStream_t s, const ConvGemmArgs& a, int W, dim3 grid) {
    const size_t smem = (size_t)W * MTB * 16 * 64 * sizeof(float);
    switch (W) {
        case 1: MMI_LAUNCH((k_conv_wide<MTB, 1, U, ELU_IN>), grid, 64, 0, s, a); break;
        case 2: MMI_LAUNCH((k_conv_wide<MTB, 2, U, ELU_IN>), grid, 128, smem, s, a); break;
        case 4: MMI_LAUNCH((k_conv_wide<MTB, 4, U, ELU_IN>), grid, 256, smem, s, a); break;
        default: MMI_LAUNCH((k_conv_wide<MTB, 8, U, ELU_IN>), grid, 512, smem, s, a); break;
    }
}
template <int MTB, int U>
void launch_wide_w(hipStream_t s, const ConvGemmArgs& a, int W, dim3 grid) {
    if (a.elu_in) launch_wide_e<MTB, U, true>(s, a, W, grid);      // ELU at the load (inputs nobody pre-activated)
    else launch_wide_e<MTB, U, false>(s, a, W, grid);
}

int launch_conv_plan(hipStream_t s, const ConvGemmArgs& a, const ConvPlan& p) {
    if (p.wide) {
        const dim3 grid(mmi_cdiv(a.Ntot, 32), mmi_cdiv(a.Mt, p.MTB));
        switch (p.MTB) {
            case 1: launch_wide_w<1, 4>(s, a, p.W, grid); break;
            case 2: launch_wide_w<2, 4>(s, a, p.W, grid); break;
            default: launch_wide_w<4, 2>(s, a, p.W, grid); break;
        }
        MMI_CHECK_LAUNCH();
        return MMI_OK;
    }
    if (p.ln) {
        MMI_LAUNCH((k_gemm_f32_ln<8, 8>), dim3(a.Mt, 1, p.nz), 512, 0, s, a);
        MMI_CHECK_LAUNCH();
        return MMI_OK;
    }
    if (p.pack) {
        const long n = (long)mmi_cdiv(a.Ntot, 32) * a.Q * 64;
        MMI_LAUNCH(k_pack_b_f32, (int)mmi_cdiv64(n, 256), 256, 0, s, a, const_cast<float*>(a.bp));
    }
    ConvGemmArgs g = a;
    if (p.ksplit > 1) g.out_mode = MMI_GOUT_PARTIAL;
    const dim3 grid(a.Mt, p.ksplit, p.nz);
    if (p.NSUB == 1) {
        if (p.waves == 8) MMI_LAUNCH((k_gemm_f32<1, 8, 2>), grid, 512, 0, s, g);
        else MMI_LAUNCH((k_gemm_f32<1, 4, 2>), grid, 256, 0, s, g);
    } else if (p.NSUB == 2) {
        if (p.waves == 8) MMI_LAUNCH((k_gemm_f32<2, 8, 2>), grid, 512, 0, s, g);
        else MMI_LAUNCH((k_gemm_f32<2, 4, 2>), grid, 256, 0, s, g);
    } else {
        MMI_LAUNCH((k_gemm_f32<4, 4, 2>), grid, 256, 0, s, g);
    }
    if (p.ksplit > 1) MMI_LAUNCH(k_conv_finish, (int)mmi_cdiv64((int64_t)a.Cout * a.Ntot, 256), 256, 0, s, a, p.ksplit);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

// Decide the launch sequence of one conv and allocate its scratch (offset table / packed operand / partials) from
// `arena`.  `a.x_packed`: the producer writes a.bp itself.  `a.out_mode` PACKED: the result feeds a linear directly.
int plan_conv(MmiArena& arena, ConvGemmArgs& a, ConvPlan* p) {
    const int N = a.Ntot;
    if (a.Cin * a.K >= (1 << 17) || N >= (1 << 17) || a.T_out >= (1 << 11) || a.K >= (1 << 11))
        return mmi_fail(MMI_ERR_UNSUPPORTED, "conv dimensions outside the range of the kernels' index arithmetic (sessions x samples per frame "
                                             "must stay below 131072: 68 sessions per handle at 1920 samples per frame)");
    if (N > 128) {
        if (a.first || a.x_packed || a.out_mode != MMI_GOUT_NATURAL)
            return mmi_fail(MMI_ERR_UNSUPPORTED, "more than 128 columns (sessions x time steps per frame) on a conv with replicate padding or packed "
                                                 "operands: lower max_batch (the 25 Hz layers of the released codec take 64 sessions)");
        p->wide = true;
        const int nsub = mmi_cdiv(N, 32);
        // m-tiles per wave (each gathered + ELU'd operand element is reused MTB times) vs. waves in flight:
        // aim for >= 1024 waves, taking the parallelism from split-K waves (W) before shrinking MTB
        int MTB = 4;                         // 8 m-tiles per wave needs > 256 registers per lane: one wave per SIMD, slower
        while (MTB > a.Mt) MTB >>= 1;
        int W = 1;
        auto waves = [&](int mtb, int w) { return nsub * mmi_cdiv(a.Mt, mtb) * w; };
        auto wmax = [&](int mtb) { int w = 1; while (w < 8 && w * 2 * mtb <= 16 && a.Q / (w * 2) >= 4) w <<= 1; return w; };
        while (MTB > 1 && waves(MTB, wmax(MTB)) < 1024) MTB >>= 1;
        while (W < wmax(MTB) && waves(MTB, W) < 1024) W <<= 1;
        // Measured tilings of the codec's own layer shapes (profiles/r02_logs/conv_wide_tiling_sweep_b{8,32,64}.txt: every
        // (MTB, W) forced on every layer, per-site times from the kernel trace).  The rule above is within a few percent of the
        // best at 8 sessions but leaves 72 us per frame at 32 sessions and 114 us at 64: what it does not see is the workgroup
        // count against the 256 CUs (a strided conv of 16 column tiles at MTB = 2 is 128 workgroups - half the chip) and the
        // 64-value epilogue of an unsplit MTB = 4 wave.  Shapes not listed (other codecs, other batch classes) keep the rule.
        // Same fp32 results up to summation order (W waves split the reduction), like any other tiling.
        {
            struct Tune { int Cin, Cout, K, S, nsub, MTB, W; };
            static const Tune kTune[] = {
                // 32 sessions (column tiles of the layer at 32 sessions)
                {128, 256, 10, 5, 96, 4, 4},    // enc.down1   45.9 -> 34.1 us
                {256, 512, 12, 6, 16, 1, 8},    // enc.down2   47.4 -> 30.8
                {128, 64, 3, 1, 480, 1, 4},     // resblock 128 ch (enc.res1 / dec.res2), first conv
                {64, 128, 1, 1, 480, 1, 4},     //   second conv                      32.9 -> 31.9, 32.7 -> 31.2
                {256, 128, 3, 1, 96, 1, 4},     // resblock 256 ch (enc.res2 / dec.res1)
                {128, 256, 1, 1, 96, 1, 4},     //                                    24.6 -> 23.3, 26.0 -> 23.0
                {512, 256, 3, 1, 16, 1, 8},     // resblock 512 ch (enc.res3 / dec.res0)
                {256, 512, 1, 1, 16, 1, 8},     //                                    21.4 -> 19.4, 21.9 -> 20.1
                {512, 3072, 1, 1, 16, 1, 8},    // dec.convtr1 (rows = Cout * K)      41.5 -> 34.2 (with its combine)
                {256, 1280, 1, 1, 96, 4, 4},    // dec.convtr2                        49.1 -> 47.0
                {128, 512, 1, 1, 480, 1, 2},    // dec.convtr3                        76.6 -> 55.6
                {64, 1, 7, 1, 1920, 1, 4},      // dec.final                          18.9 -> 16.6
                // 64 sessions
                {128, 256, 10, 5, 192, 2, 8},   // enc.down1   78.2 -> 64.1
                {256, 512, 12, 6, 32, 2, 8},    // enc.down2   54.8 -> 49.0
                {256, 128, 3, 1, 192, 1, 4},    // resblock 256 ch                    42.3 -> 34.9, 41.4 -> 34.1
                {128, 256, 1, 1, 192, 1, 4},
                {512, 256, 3, 1, 32, 1, 8},     // resblock 512 ch                    34.1 -> 22.5, 36.3 -> 22.5
                {256, 512, 1, 1, 32, 1, 8},
                {512, 3072, 1, 1, 32, 1, 8},    // dec.convtr1                        61.1 -> 58.9
                {256, 1280, 1, 1, 192, 4, 4},   // dec.convtr2                       100.2 -> 82.3
                {128, 512, 1, 1, 960, 1, 2},    // dec.convtr3                       128.8 -> 99.5
                {64, 1, 7, 1, 3840, 1, 4},      // dec.final                          31.5 -> 28.1
            };
            for (const Tune& t : kTune)
                    if (t.Cin == a.Cin && t.Cout == a.Cout && t.K == a.K && t.S == a.S && 4 * nsub >= 3 * t.nsub && 2 * nsub < 3 * t.nsub) {
                        MTB = t.MTB; W = t.W;
                        while (MTB > a.Mt) MTB >>= 1;
                        break;
                    }
        }
        if (const char* e = getenv("MMI_CONV_MTB")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) MTB = v; }   // test hooks
        if (const char* e = getenv("MMI_CONV_W")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) W = v; }
        while (W * MTB > 16) W >>= 1;          // split-K reduction buffer: W * MTB * 4 KiB of LDS, keep it within 64 KiB
        p->MTB = MTB; p->W = W;
        std::vector<int> tab((size_t)a.Q * 8);
        for (int q = 0; q < a.Q; ++q)
            for (int h = 0; h < 2; ++h)
                for (int e = 0; e < 4; ++e) {
                    const int kd = (q * 4 + e) * 2 + h;
                    tab[((size_t)q * 2 + h) * 4 + e] = kd < a.Cin * a.K ? (kd / a.K) * a.x_ld + kd % a.K : 0;
                }
        int* dev = nullptr;
        MMI_HIP_CHECK(arena.alloc(&dev, tab.size()));
        MMI_HIP_CHECK(hipMemcpy(dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        a.koff = dev;
        return MMI_OK;
    }
    p->wide = false;
    const int nsub = mmi_cdiv(N, 32);
    if (a.ln_w) {                       // fused LayerNorm + linear: one 32-column subtile per workgroup, the whole K in registers
        if (!a.x_packed || a.Q > 64 || a.Q * 8 != a.Cin || a.K != 1 || a.out_mode == MMI_GOUT_PARTIAL)
            return mmi_fail(MMI_ERR_UNSUPPORTED, "fused LayerNorm linear: unsupported shape");
        p->ln = true;
        p->NSUB = 1;
        p->nz = nsub;
        a.Npad = 32;
        return MMI_OK;
    }
    p->NSUB = nsub <= 1 ? 1 : (nsub == 2 ? 2 : 4);
    a.Npad = p->NSUB * 32;
    if (!a.x_packed) {
        if (!a.bp) {
            float* bp = nullptr;
            MMI_HIP_CHECK(arena.alloc(&bp, (size_t)nsub * a.Q * 256));
            a.bp = bp;
        }
        p->pack = true;
    }
    // small weight matrices (the transformer linears, <= 8 MB): one workgroup per n-subtile, up to 16 waves splitting K
    const bool spread = nsub >= 2 && (size_t)a.Mt * a.Q * 1024 <= ((size_t)8 << 20);
    int ks = 1;
    if (a.out_mode == MMI_GOUT_NATURAL) {
        // only the large strided convs (>= 2048-deep reductions, tens of MB of weights) are worth a finishing launch
        while (a.Q >= 256 && ks < 8 && a.Mt * ks * 2 <= 256 && a.Q / (ks * 2) >= 32) ks <<= 1;
    }
    if (const char* e = getenv("MMI_CONV_KSPLIT")) {   // test hook
        const int v = atoi(e);
        if (a.out_mode == MMI_GOUT_NATURAL && v >= 1 && v <= 8 && a.Q >= v) ks = v;
    }
    p->ksplit = ks;
    if (ks > 1 && !a.partial) {
        float* part = nullptr;
        MMI_HIP_CHECK(arena.alloc(&part, (size_t)ks * a.Mt * 32 * a.Npad));
        a.partial = part;
    }
    const int qblk = a.Q / ks;
    if (spread) {
        p->NSUB = 1;
        p->nz = nsub;
        p->waves = qblk >= 16 ? 8 : 4;
    } else {
        p->waves = (p->NSUB <= 2 && qblk >= 32) ? 8 : 4;
    }
    return MMI_OK;
}

ConvGemmArgs conv_args(const ConvW& w, const Buf& in, int x_off, int T_out, const Buf& out, int out_off, int B,
                       bool elu_in) {
    ConvGemmArgs a;
    memset(&a, 0, sizeof(a));
    // an ELU-fronted conv reads the producer's pre-activated values when they exist (Buf::pe / Buf::elu)
    a.x = elu_in ? in.conv_src() : in.p;
    if (elu_in && !in.needs_elu_at_load()) elu_in = false;
    a.x_ld = in.ld; a.x_bstride = (long)in.C * in.ld; a.x_off = x_off; a.H = in.H;
    a.wpk = w.wpk; a.bias = w.bias;
    a.out = out.p; a.out_ld = out.ld; a.out_off = out_off;
    a.out2 = out.pe; a.elu_out = out.elu ? 1 : 0;
    a.B = B; a.Cin = w.Cin; a.Cout = w.Cout; a.K = w.K; a.S = w.S; a.T_out = T_out;
    a.T_magic = mmi_div_magic(T_out); a.K_magic = mmi_div_magic(w.K);
    a.Mt = w.Mt; a.Q = w.Q; a.Ntot = B * T_out;
    a.elu_in = elu_in ? 1 : 0;
    a.act_out = MMI_ACT_NONE;
    a.out_mode = MMI_GOUT_NATURAL;
    return a;
}

// Audio-rate buffers (T >= 32 new columns per row): the row stride is a multiple of 32 floats and the allocation leads with
// `lead` floats so that the T NEW columns of every row start on a 128-byte line (the H history columns sit right in front of
// them).  A 32-column tile of a conv's output is then exactly one cache line per row: consecutive workgroups run on different
// XCDs (different L2s), and with rows starting at arbitrary 8-byte offsets every line was written in part by two of them.
// MMI_MIMI_NO_ALIGN=1 keeps the dense layout (A/B).  Nothing else knows: every kernel takes (pointer, row stride, H).
int alloc_buf(mmi_mimi* m, int B, int C, int H, int T, Buf* b, hipStream_t s) {
    b->C = C; b->H = H; b->ld = H + T; b->lead = 0;
    if (T >= 32 && !getenv("MMI_MIMI_NO_ALIGN")) {
        b->lead = (32 - H % 32) % 32;
        b->ld = ((b->lead + H + T + 31) / 32) * 32;
    }
    size_t n = (size_t)B * C * b->ld + b->lead;
    float* base = nullptr;
    MMI_HIP_CHECK(m->st.alloc(&base, n));
    MMI_HIP_CHECK(hipMemsetAsync(base, 0, n * sizeof(float), s));
    b->p = base + b->lead;
    return MMI_OK;
}

// ELU twin of a buffer (same geometry, zero history: ELU(0) = 0)
int alloc_twin(mmi_mimi* m, int B, Buf* b, hipStream_t s) {
    size_t n = (size_t)B * b->C * b->ld + b->lead;
    float* base = nullptr;
    MMI_HIP_CHECK(m->st.alloc(&base, n));
    MMI_HIP_CHECK(hipMemsetAsync(base, 0, n * sizeof(float), s));
    b->pe = base + b->lead;
    return MMI_OK;
}

// the history a consumer conv reads lives in the buffer it reads: the ELU twin when there is one
HistDesc hist_of(const Buf& b, int T);
HistDesc hist_of_src(const Buf& b, int T) {
    Buf c = b;
    c.p = b.conv_src();
    return hist_of(c, T);
}

int add_conv(mmi_mimi* m, MmiProgram& prog, ConvGemmArgs a, MmiArena* arena = nullptr) {
    ConvPlan p;
    int rc = plan_conv(arena ? *arena : m->st, a, &p);
    if (rc) return rc;
    prog.add([a, p](hipStream_t s) { return launch_conv_plan(s, a, p); });
    return MMI_OK;
}

// SEANet residual block (conv K -> hidden, conv 1x1 -> channels, + input): ONE launch (k_resblock) at audio rate when the block
// has the shape the kernel is built for and enough columns to fill the chip with 32-column workgroups; else the two convs.
int add_resblock(mmi_mimi* m, MmiProgram& prog, ConvGemmArgs a1, ConvGemmArgs a2, const ConvW& w2) {
    const int nblk = mmi_cdiv(a1.Ntot, 32);
    long min_waves = 256;                        // below that the two-launch path (split-K over waves) fills the chip better
    if (const char* e = getenv("MMI_MIMI_RES_FUSION_MIN")) min_waves = atol(e);      // test hook: the tiny shapes
    // Measured on MI355X (profiles/r02_logs/resblock_*): the 64-channel blocks (one hidden tile, 1920 one-wave workgroups) gain
    // 8-13 us each over two launches; the 128 / 256-channel blocks (480 / 96 workgroups of 2 / 4 waves: one wave per SIMD on part of
    // the chip) lost 2-15 us to the two-launch path's split-K over waves and are not instantiated.
    const bool fuse = a1.Ntot > 128 && a1.S == 1 && a2.K == 1 && a2.S == 1 && a1.Cout == a2.Cin && a1.Ntot == a2.Ntot &&
                      a1.T_out == a2.T_out && a1.Mt == 1 && w2.wpk_h && a2.Q <= 4 * a1.Mt &&
                      a2.Mt >= a1.Mt && (long)nblk * a1.Mt >= min_waves && !a1.first && !a2.first && !a1.res && !a1.scale && !a2.scale &&
                      a1.T_out % 32 == 0 && a1.Cin <= 64 * a1.Mt && a1.K <= 33 &&
                      a1.act_out == MMI_ACT_NONE && a2.act_out == MMI_ACT_NONE;
    if (!fuse) {
        int rc = add_conv(m, prog, a1);
        if (rc) return rc;
        return add_conv(m, prog, a2);
    }
    a2.wpk = w2.wpk_h;
    a2.elu_in = 0;                               // the hidden tensor never exists un-activated
    ResBlockArgs ra;
    ra.a1 = a1; ra.a2 = a2;
    const int mt1 = a1.Mt;
    // LDS: input window Cin x (32 + K - 1) | offset table | the waves' hand-off mailbox
    const size_t smem = ((size_t)((a1.Cin * (32 + a1.K - 1) + 3) & ~3) + (size_t)a1.Q * 8 + (mt1 > 1 ? (size_t)mt1 * 1024 : 0)) * sizeof(float);
    if (smem > 65536) return mmi_fail(MMI_ERR_UNSUPPORTED, "k_resblock: input window does not fit in LDS");
    prog.add([ra, nblk, smem](hipStream_t s) {
        MMI_LAUNCH((k_resblock<1, 4>), nblk, 64, smem, s, ra);
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
    return MMI_OK;
}

// residual vector quantiser: latent [B][dim] (column `lat_off` of rows of length lat_ld) -> codes_i32 [B][n_q]
// latent_packed: the producer of the latent also wrote it as the packed operand m->qin_bp (ConvGemmArgs::outp): no pack launch
int add_quantize_ops(mmi_mimi* m, MmiProgram& prog, const float* latent, int lat_ld, int lat_off, int B, bool latent_packed = false) {
    const mmi_mimi_cfg& c = m->cfg;
    const int D = c.q_dimension, bins = c.q_bins;
    Buf in; in.p = const_cast<float*>(latent); in.C = c.dimension; in.ld = lat_ld; in.H = lat_off;
    Buf out; out.p = m->xq; out.C = 2 * D; out.ld = 1; out.H = 0;
    {
        ConvGemmArgs a = conv_args(m->q_in, in, lat_off, 1, out, 0, B, false);
        a.bp = m->qin_bp; a.partial = m->qin_part;
        a.x_packed = latent_packed ? 1 : 0;
        int rc = add_conv(m, prog, a);
        if (rc) return rc;
    }
    const int K = m->n_codebooks;
    const int nchunk = m->nchunk;
    const size_t smem = ((size_t)MMI_RVQ_CHUNK * (D + 4) + (size_t)8 * D) * sizeof(float);
    auto level = [&](int k, int slot) {
        const bool sem = k < c.q_n_q_semantic;
        RvqLevel L;
        L.x = m->xq + (sem ? 0 : D);
        L.E = m->E_all + (size_t)k * bins * D;
        L.e2 = m->e2_all + (size_t)k * bins;
        L.best_d = m->best_d + (size_t)slot * nchunk * m->max_batch;
        L.best_i = m->best_i + (size_t)slot * nchunk * m->max_batch;
        L.level = k;
        return L;
    };
    int* codes = m->codes_i32; const int nq = c.q_n_q;
    // the semantic quantiser's single level and the first acoustic level share their two launches (independent, see RvqLevel)
    const bool pair = c.q_n_q_semantic == 1 && K >= 2 && !getenv("MMI_RVQ_NO_PAIR");
    for (int k = 0; k < K; ++k) {
        const RvqLevel a0 = level(k, 0);
        const bool two = pair && k == 0;
        const RvqLevel a1 = two ? level(1, 1) : a0;
        const int nz = two ? 2 : 1;
        prog.add([=](hipStream_t s) {
            MMI_LAUNCH(k_rvq_dist, dim3(nchunk, mmi_cdiv(B, 8), nz), 256, smem, s, a0, a1, 2 * D, B, D, bins);
            MMI_LAUNCH(k_rvq_select, dim3(B, nz), 256, 0, s, a0, a1, nchunk, 2 * D, codes, nq, B, D);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
        if (two) ++k;
    }
    return MMI_OK;
}

// codes_i32 [B][n_q] (first K used) -> latent written to out buffer column out_off
// k_from_device: the streaming decoder's captured program (K of each call comes from m->dec_k_dev); the one-shot programs of
// decode_latent bake their K in
int add_dequant_ops(mmi_mimi* m, MmiProgram& prog, int K, const Buf& out, int out_off, int B, bool k_from_device = false) {
    const mmi_mimi_cfg& c = m->cfg;
    const int D = c.q_dimension, bins = c.q_bins, nq = c.q_n_q, nsem = c.q_n_q_semantic;
    float* q2 = m->q2; const int* codes = m->dec_codes_i32; const float* E = m->E_all; const int* kdev = k_from_device ? m->dec_k_dev : nullptr;
    const bool packed = B <= 128 && m->q_out.Q * 8 == 2 * D && !getenv("MMI_MIMI_PACK_LAUNCHES");
    float* qp = packed ? m->qout_bp : nullptr; const int qQ = m->q_out.Q;
    prog.add([=](hipStream_t s) {
        MMI_LAUNCH(k_rvq_gather, mmi_cdiv(B * D, 256), 256, 0, s, codes, nq, K, E, bins, D, nsem, q2, B, kdev, qp, qQ);
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
    Buf in; in.p = m->q2; in.C = 2 * D; in.ld = 1; in.H = 0;
    ConvGemmArgs a = conv_args(m->q_out, in, 0, 1, out, out_off, B, false);
    a.bp = m->qout_bp; a.partial = m->qout_part;
    a.x_packed = packed ? 1 : 0;                 // the gather wrote the operand
    return add_conv(m, prog, a);
}

// one streaming transformer (ProjectedTransformer with conv_layout, transformer.py:932-983) working in place
// on x = buf[:, :, off:off+T]
int add_transformer(mmi_mimi* m, MmiProgram& prog, const std::vector<TrLayerW>& layers, const Buf& xb, int off, int T,
                    float* kv, const long* offsets, int B, hipStream_t init_stream) {
    const mmi_mimi_cfg& c = m->cfg;
    const int d = c.tr_d_model, H = c.tr_num_heads, Dh = d / H, ff = c.tr_dim_feedforward, cap = c.tr_context;
    const int N = B * T;
    // At <= 128 columns (every real configuration: T = 2 per session) the linears run on k_gemm_f32 and the producers
    // (LayerNorm, attention, linear1+GELU) write its packed operand directly; otherwise plain [B][C][T] buffers.
    const bool packed = N <= 128;
    Buf y, qkv, att, hb;
    int rc;
    if ((rc = alloc_buf(m, B, d, 0, T, &y, init_stream))) return rc;
    if ((rc = alloc_buf(m, B, 3 * d, 0, T, &qkv, init_stream))) return rc;
    if ((rc = alloc_buf(m, B, d, 0, T, &att, init_stream))) return rc;
    if ((rc = alloc_buf(m, B, ff, 0, T, &hb, init_stream))) return rc;
    float *yp = nullptr, *attp = nullptr, *hbp = nullptr;
    const int nsub = mmi_cdiv(N, 32), Qd = mmi_cdiv(d, 8), Qff = mmi_cdiv(ff, 8);
    if (packed) {
        MMI_HIP_CHECK(m->st.alloc(&yp, (size_t)nsub * Qd * 256));
        MMI_HIP_CHECK(m->st.alloc(&attp, (size_t)nsub * Qd * 256));
        MMI_HIP_CHECK(m->st.alloc(&hbp, (size_t)nsub * Qff * 256));
        MMI_HIP_CHECK(hipMemsetAsync(yp, 0, (size_t)nsub * Qd * 256 * sizeof(float), init_stream));
        MMI_HIP_CHECK(hipMemsetAsync(attp, 0, (size_t)nsub * Qd * 256 * sizeof(float), init_stream));
        MMI_HIP_CHECK(hipMemsetAsync(hbp, 0, (size_t)nsub * Qff * 256 * sizeof(float), init_stream));
    }
    const size_t kv_layer = (size_t)B * H * cap * Dh;
    if (T > 2 || (Dh != 16 && Dh != 32 && Dh != 64))
        return mmi_fail(MMI_ERR_UNSUPPORTED, "Mimi attention: head dim 16/32/64 and 1 or 2 steps per frame are built");
    const size_t attn_smem = ((size_t)T * Dh + (((size_t)T * cap + 3) & ~(size_t)3) + (size_t)(256 / (Dh / 4)) * T * Dh + 8) * sizeof(float);
    auto add_norm = [&](const float* w, const float* bb) {
        const float* xp = xb.p; int xld = xb.ld; float* yn = y.p;
        prog.add([=](hipStream_t s) {
            MMI_LAUNCH(k_layernorm_ct, B * T, 64, 0, s, xp, xld, off, w, bb, yn, T, 0, d, T, 1e-5f, yp, Qd);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    };
    // LayerNorm fused into the linear that consumes it (k_gemm_f32_ln): the producers of the residual stream inside the
    // transformer (out_proj, linear2) also store it in packed operand order (xrp); only the very first norm, whose input
    // comes from the conv stack, stays a launch of its own.  32 -> 2 norm launches per frame for the two transformers.
    const bool fuse_ln = packed && d % 8 == 0 && Qd <= 64 && layers.size() > 0 && layers[0].n1w_pk;
    float* xrp = nullptr;
    if (fuse_ln) {
        MMI_HIP_CHECK(m->st.alloc(&xrp, (size_t)nsub * Qd * 256));
        MMI_HIP_CHECK(hipMemsetAsync(xrp, 0, (size_t)nsub * Qd * 256 * sizeof(float), init_stream));
    }
    bool x_is_packed = false;           // xrp holds the current residual stream
    for (size_t l = 0; l < layers.size(); ++l) {
        const TrLayerW& L = layers[l];
        {   // x = x + ls1 * out_proj(attn(norm1(x)))
            ConvGemmArgs ai = conv_args(L.in_proj, y, 0, T, qkv, 0, B, false);
            if (fuse_ln && x_is_packed) {
                ai.x_packed = 1; ai.bp = xrp; ai.ln_w = L.n1w_pk; ai.ln_b = L.n1b_pk; ai.ln_eps = 1e-5f;
            } else {
                add_norm(L.n1w, L.n1b);
                if (packed) { ai.x_packed = 1; ai.bp = yp; }
            }
            if ((rc = add_conv(m, prog, ai))) return rc;
            MimiAttnArgs aa;
            aa.qkv = qkv.p; aa.kc = kv + (2 * l) * kv_layer; aa.vc = kv + (2 * l + 1) * kv_layer;
            aa.offsets = offsets; aa.out = att.p; aa.B = B; aa.H = H; aa.D = Dh; aa.T = T; aa.cap = cap;
            aa.context = c.tr_context; aa.max_period = c.tr_max_period;
            aa.outp = attp; aa.outQ = Qd;
            // one memory round trip when a thread's 16 row slots cover the ring (the Mimi shape: 250 slots of 64 floats)
            const bool one_pass = cap <= 16 * (256 / (Dh / 4)) && !getenv("MMI_MIMI_ATTN_TWO_PASS");
            prog.add([=](hipStream_t s) {
#define MMI_ATT(D_, T_) do { if (one_pass) MMI_LAUNCH((k_mimi_attn_1pass<D_, T_>), B * H, 256, 0, s, aa); \
                             else MMI_LAUNCH((k_mimi_attn<D_, T_>), B * H, 256, attn_smem, s, aa); } while (0)
                if (Dh == 64 && T == 2) MMI_ATT(64, 2);
                else if (Dh == 64 && T == 1) MMI_ATT(64, 1);
                else if (Dh == 32 && T == 2) MMI_ATT(32, 2);
                else if (Dh == 32 && T == 1) MMI_ATT(32, 1);
                else if (Dh == 16 && T == 2) MMI_ATT(16, 2);
                else if (Dh == 16 && T == 1) MMI_ATT(16, 1);
                else return mmi_fail(MMI_ERR_UNSUPPORTED, "Mimi attention: head dim 16/32/64 and 1 or 2 steps per frame are built");
#undef MMI_ATT
                MMI_CHECK_LAUNCH();
                return (int)MMI_OK;
            });
            ConvGemmArgs a = conv_args(L.out_proj, att, 0, T, xb, off, B, false);
            a.res = xb.p; a.res_ld = xb.ld; a.res_off = off; a.scale = L.ls1;
            if (packed) { a.x_packed = 1; a.bp = attp; }
            if (fuse_ln) { a.outp = xrp; a.outQ = Qd; x_is_packed = true; }      // dual store: [B][C][T] and packed
            if ((rc = add_conv(m, prog, a))) return rc;
        }
        {   // x = x + ls2 * linear2(gelu(linear1(norm2(x))))
            ConvGemmArgs a1 = conv_args(L.lin1, y, 0, T, hb, 0, B, false);
            a1.act_out = MMI_ACT_GELU;
            if (fuse_ln && x_is_packed) {
                a1.x_packed = 1; a1.bp = xrp; a1.ln_w = L.n2w_pk; a1.ln_b = L.n2b_pk; a1.ln_eps = 1e-5f;
                a1.out_mode = MMI_GOUT_PACKED; a1.outp = hbp; a1.outQ = Qff;
            } else {
                add_norm(L.n2w, L.n2b);
                if (packed) { a1.x_packed = 1; a1.bp = yp; a1.out_mode = MMI_GOUT_PACKED; a1.outp = hbp; a1.outQ = Qff; }
            }
            if ((rc = add_conv(m, prog, a1))) return rc;
            ConvGemmArgs a2 = conv_args(L.lin2, hb, 0, T, xb, off, B, false);
            a2.res = xb.p; a2.res_ld = xb.ld; a2.res_off = off; a2.scale = L.ls2;
            if (packed) { a2.x_packed = 1; a2.bp = hbp; }
            if (fuse_ln) { a2.outp = xrp; a2.outQ = Qd; x_is_packed = true; }
            if ((rc = add_conv(m, prog, a2))) return rc;
        }
    }
    return MMI_OK;
}

int upload_hist(mmi_mimi* m, std::vector<HistDesc>& h, int B, HistDesc** dev, int* n, int* rows, HistTable* tab = nullptr) {
    int acc = 0;
    std::vector<HistDesc> keep;
    for (auto& d : h) {
        if (d.H <= 0) continue;
        d.row_begin = acc;
        acc += B * d.C;
        keep.push_back(d);
    }
    *n = (int)keep.size();
    *rows = acc;
    if (tab) {                                   // by-value copy for k_commit_all (n = -1: too many descriptors, two-launch path)
        memset(tab, 0, sizeof(*tab));
        tab->n = keep.size() <= MMI_HIST_MAX ? (int)keep.size() : -1;
        for (size_t i = 0; i < keep.size() && i < MMI_HIST_MAX; ++i) tab->d[i] = keep[i];
    }
    MMI_HIP_CHECK(m->st.alloc(dev, keep.size() ? keep.size() : 1));
    if (!keep.empty())
        MMI_HIP_CHECK(hipMemcpy(*dev, keep.data(), keep.size() * sizeof(HistDesc), hipMemcpyHostToDevice));
    return MMI_OK;
}

HistDesc hist_of(const Buf& b, int T) {
    HistDesc d;
    d.p = b.p; d.C = b.C; d.ld = b.ld; d.H = b.H; d.T = T; d.row_begin = 0;
    return d;
}

int build_encoder(mmi_mimi* m, int B, hipStream_t s0) {
    const mmi_mimi_cfg& c = m->cfg;
    MmiProgram& prog = m->enc_prog;
    std::vector<HistDesc> hist;
    int rc;
    int T = c.frame_size;
    size_t ci = 0;
    // input PCM buffer with the first conv's history
    if ((rc = alloc_buf(m, B, c.channels, c.kernel_size - 1, T, &m->enc_in, s0))) return rc;
    hist.push_back(hist_of(m->enc_in, T));
    Buf cur = m->enc_in;
    int mult = 1;
    // ELU hoisted into the producers (mimi_kernels.h ConvGemmArgs::elu_out / out2): a resblock input is kept raw (residual) AND
    // ELU'd (its first conv reads that, history included); everything else a conv reads through ELU is stored ELU'd only
    {   // conv0: channels -> n_filters, K = kernel_size; consumer = resblock conv (K = residual_kernel_size)
        Buf nxt;
        if ((rc = alloc_buf(m, B, c.n_filters, c.residual_kernel_size - 1, T, &nxt, s0))) return rc;
        if ((rc = alloc_twin(m, B, &nxt, s0))) return rc;
        prog.site("enc.conv0");
        if ((rc = add_conv(m, prog, conv_args(m->enc_convs[ci++], cur, 0, T, nxt, nxt.H, B, false)))) return rc;
        hist.push_back(hist_of_src(nxt, T));
        cur = nxt;
    }
    for (int i = 0; i < c.n_ratios; ++i) {
        const int ratio = c.ratios[c.n_ratios - 1 - i];
        const int ch = mult * c.n_filters;
        // resblock: cur -> (ELU, conv K3 -> hidden) -> (ELU, conv K1 -> ch) + cur ; consumer = strided conv (K=2r, S=r)
        Buf hid, nxt;
        if ((rc = alloc_buf(m, B, ch / c.compress, 0, T, &hid, s0))) return rc;
        if ((rc = alloc_buf(m, B, ch, ratio, T, &nxt, s0))) return rc;
        hid.elu = nxt.elu = true;               // read only by the block's second conv / the strided conv, both behind an ELU
        prog.site("enc.res" + std::to_string(i));
        const ConvGemmArgs a1 = conv_args(m->enc_convs[ci++], cur, 0, T, hid, 0, B, true);
        const ConvW& w2 = m->enc_convs[ci++];
        ConvGemmArgs a = conv_args(w2, hid, 0, T, nxt, nxt.H, B, true);
        a.res = cur.p; a.res_ld = cur.ld; a.res_off = cur.H;
        if ((rc = add_resblock(m, prog, a1, a, w2))) return rc;
        hist.push_back(hist_of(nxt, T));
        cur = nxt;
        // strided conv: ch -> 2ch ; consumer = next resblock conv (K3) or the final conv (last_kernel_size)
        const int Tn = T / ratio;
        const int Hn = (i + 1 < c.n_ratios) ? c.residual_kernel_size - 1 : c.last_kernel_size - 1;
        Buf nx2;
        if ((rc = alloc_buf(m, B, 2 * ch, Hn, Tn, &nx2, s0))) return rc;
        if (i + 1 < c.n_ratios) { if ((rc = alloc_twin(m, B, &nx2, s0))) return rc; }   // the next resblock's input
        else nx2.elu = true;                                                                     // read by the final conv only
        prog.site("enc.down" + std::to_string(i));
        if ((rc = add_conv(m, prog, conv_args(m->enc_convs[ci++], cur, 0, Tn, nx2, nx2.H, B, true)))) return rc;
        hist.push_back(hist_of_src(nx2, Tn));
        cur = nx2;
        T = Tn;
        mult *= 2;
    }
    // final conv -> dimension, written into the downsample conv's input buffer (history = stride)
    const int stride = c.resample_stride;
    Buf dsin;
    if ((rc = alloc_buf(m, B, c.dimension, 2 * stride - stride, T, &dsin, s0))) return rc;
    prog.site("enc.final");
    if ((rc = add_conv(m, prog, conv_args(m->enc_convs[ci++], cur, 0, T, dsin, dsin.H, B, true)))) return rc;
    hist.push_back(hist_of(dsin, T));
    // encoder transformer, in place on dsin[:, :, H:H+T]
    {
        const int Dh = c.tr_d_model / c.tr_num_heads;
        size_t kvn = (size_t)2 * c.tr_num_layers * B * c.tr_num_heads * c.tr_context * Dh;
        MMI_HIP_CHECK(m->st.alloc(&m->enc_kv, kvn));
        MMI_HIP_CHECK(hipMemsetAsync(m->enc_kv, 0, kvn * sizeof(float), s0));
        prog.site("enc.tr");
        if ((rc = add_transformer(m, prog, m->enc_tr, dsin, dsin.H, T, m->enc_kv, m->counters, B, s0))) return rc;
    }
    const int T_tr = T;
    // downsample (ConvDownsample1d learnt, replicate padding on the first frame)
    if (T % stride != 0 || T / stride != 1)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "frame_size must map to exactly one latent column");
    if ((rc = alloc_buf(m, B, c.dimension, 0, 1, &m->latent, s0))) return rc;
    const bool latent_packed = B <= 128 && m->q_in.Q * 8 == c.dimension && !getenv("MMI_MIMI_PACK_LAUNCHES");
    {
        ConvGemmArgs a = conv_args(m->downsample, dsin, 0, 1, m->latent, 0, B, false);
        a.first = m->first; a.exec = m->exec;
        if (latent_packed) { a.outp = m->qin_bp; a.outQ = m->q_in.Q; }   // the latent, also as the RVQ input projection's operand
        prog.site("enc.downsample");
        if ((rc = add_conv(m, prog, a))) return rc;
    }
    prog.site("enc.rvq");
    if ((rc = add_quantize_ops(m, prog, m->latent.p, 1, 0, B, latent_packed))) return rc;
    prog.site("enc.commit");
    // commit
    HistTable tab;
    if ((rc = upload_hist(m, hist, B, &m->enc_hist, &m->enc_nhist, &m->enc_hist_rows, &tab))) return rc;
    {
        HistDesc* hd = m->enc_hist; int nh = m->enc_nhist, rows = m->enc_hist_rows;
        const uint8_t* ex = m->exec; uint8_t* fi = m->first; long* cnt = m->counters;
        const bool one = tab.n >= 0 && !getenv("MMI_MIMI_TWO_COMMITS");
        prog.add([=](hipStream_t s) {
            if (one) {
                MMI_LAUNCH(k_commit_all, mmi_cdiv(rows + B, 256), 256, 0, s, tab, rows, ex, cnt, 1, T_tr, fi, B);
                MMI_CHECK_LAUNCH();
                return (int)MMI_OK;
            }
            if (rows > 0) MMI_LAUNCH(k_commit_history, mmi_cdiv(rows, 256), 256, 0, s, (const HistDesc*)hd, nh, rows, ex);
            MMI_LAUNCH(k_commit_counters, mmi_cdiv(B, 64), 64, 0, s, cnt, 1, T_tr, fi, ex, B);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    return MMI_OK;
}

int build_decoder(mmi_mimi* m, int B, hipStream_t s0) {
    const mmi_mimi_cfg& c = m->cfg;
    MmiProgram& prog = m->dec_prog;
    std::vector<HistDesc> hist;
    int rc;
    const int stride = c.resample_stride;
    // dequantised latent [B][dim][1]
    if ((rc = alloc_buf(m, B, c.dimension, 0, 1, &m->dec_codes_lat, s0))) return rc;
    prog.site("dec.dequant");
    if ((rc = add_dequant_ops(m, prog, m->n_codebooks, m->dec_codes_lat, 0, B, /*k_from_device=*/true))) return rc;
    prog.site("dec.upsample");
    // upsample (depthwise transposed conv) into the decoder transformer buffer = input of decoder conv0
    int T = stride;
    Buf din;
    if ((rc = alloc_buf(m, B, c.dimension, c.kernel_size - 1, T, &din, s0))) return rc;
    hist.push_back(hist_of(din, T));
    {
        float* part = nullptr;
        long pn = (long)c.dimension * stride;
        MMI_HIP_CHECK(m->st.alloc(&part, (size_t)B * pn));
        MMI_HIP_CHECK(hipMemsetAsync(part, 0, (size_t)B * pn * sizeof(float), s0));
        m->partials.push_back(part); m->partial_sizes.push_back(pn);
        const float* x = m->dec_codes_lat.p; const float* w = m->upsample_w; const uint8_t* ex = m->exec;
        float* out = din.p; int old = din.ld, ooff = din.H; int C = c.dimension;
        prog.add([=](hipStream_t s) {
            MMI_LAUNCH(k_upsample_dw, mmi_cdiv(B * C * stride, 256), 256, 0, s, x, 1, 0, w, part, ex, out, old, ooff, B, C,
                       2 * stride, stride, 1);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    {
        const int Dh = c.tr_d_model / c.tr_num_heads;
        size_t kvn = (size_t)2 * c.tr_num_layers * B * c.tr_num_heads * c.tr_context * Dh;
        MMI_HIP_CHECK(m->st.alloc(&m->dec_kv, kvn));
        MMI_HIP_CHECK(hipMemsetAsync(m->dec_kv, 0, kvn * sizeof(float), s0));
        prog.site("dec.tr");
        if ((rc = add_transformer(m, prog, m->dec_tr, din, din.H, T, m->dec_kv, m->counters + B, B, s0))) return rc;
    }
    const int T_tr = T;
    size_t ci = 0;
    int mult = 1 << c.n_ratios;
    Buf cur;
    float* conv0_bp = nullptr;                            // dec.conv0's output in the packed operand order of dec.convtr0's GEMM
    {   // conv0: dimension -> mult*n_filters (no activation before it); consumer = conv-transpose GEMM (no history)
        if ((rc = alloc_buf(m, B, mult * c.n_filters, 0, T, &cur, s0))) return rc;
        cur.elu = true;
        prog.site("dec.conv0");
        ConvGemmArgs a0 = conv_args(m->dec_convs[ci++], din, 0, T, cur, 0, B, false);
        // its (ELU'd) output is read by the first transposed-conv GEMM only: stored as that GEMM's packed operand as well
        const ConvW& wtr0 = m->dec_convs[ci];
        if (B * T <= 128 && wtr0.Q * 8 == mult * c.n_filters && !getenv("MMI_MIMI_PACK_LAUNCHES")) {
            MMI_HIP_CHECK(m->st.alloc(&conv0_bp, (size_t)mmi_cdiv(B * T, 32) * wtr0.Q * 256));
            MMI_HIP_CHECK(hipMemsetAsync(conv0_bp, 0, (size_t)mmi_cdiv(B * T, 32) * wtr0.Q * 256 * sizeof(float), s0));
            a0.outp = conv0_bp; a0.outQ = wtr0.Q;
        }
        if ((rc = add_conv(m, prog, a0))) return rc;
    }
    for (int i = 0; i < c.n_ratios; ++i) {
        const int ratio = c.ratios[i];
        const int cin = mult * c.n_filters, cout = cin / 2, K = 2 * ratio;
        // ELU + ConvTranspose1d: GEMM over channels, then overlap-add with the streaming partial
        Buf tmp, up;
        if ((rc = alloc_buf(m, B, cout * K, 0, T, &tmp, s0))) return rc;
        const int Tn = T * ratio;
        if ((rc = alloc_buf(m, B, cout, c.residual_kernel_size - 1, Tn, &up, s0))) return rc;
        if ((rc = alloc_twin(m, B, &up, s0))) return rc;           // resblock input: raw for the residual, ELU'd for its conv
        const ConvW& wtr = m->dec_convs[ci++];
        prog.site("dec.convtr" + std::to_string(i));
        {
            ConvGemmArgs a = conv_args(wtr, cur, 0, T, tmp, 0, B, true);
            a.bias = nullptr;  // bias is added once, in the combine step
            if (i == 0 && conv0_bp) { a.bp = conv0_bp; a.x_packed = 1; }
            if ((rc = add_conv(m, prog, a))) return rc;
        }
        {
            float* part = nullptr;
            long pn = (long)cout * (K - ratio);
            MMI_HIP_CHECK(m->st.alloc(&part, (size_t)B * pn));
            MMI_HIP_CHECK(hipMemsetAsync(part, 0, (size_t)B * pn * sizeof(float), s0));
            m->partials.push_back(part); m->partial_sizes.push_back(pn);
            const float* tp = tmp.p; const float* bias = wtr.bias; const uint8_t* ex = m->exec;
            float* out = up.p; float* out2 = up.pe; int old = up.ld, ooff = up.H; int Tin = T; const int tld = tmp.ld;
            prog.add([=](hipStream_t s) {
                MMI_LAUNCH(k_convtr_combine, (int)mmi_cdiv64((int64_t)B * cout * Tin * ratio, 256), 256, 0, s, tp, bias, part,
                           ex, out, old, ooff, B, cout, K, ratio, Tin, out2, tld);
                MMI_CHECK_LAUNCH();
                return (int)MMI_OK;
            });
        }
        hist.push_back(hist_of_src(up, Tn));
        T = Tn;
        // resblock; consumer = next conv-transpose GEMM (no history) or the final conv (last_kernel_size)
        Buf hid, nxt;
        const int Hn = (i + 1 < c.n_ratios) ? 0 : c.last_kernel_size - 1;
        if ((rc = alloc_buf(m, B, cout / c.compress, 0, T, &hid, s0))) return rc;
        if ((rc = alloc_buf(m, B, cout, Hn, T, &nxt, s0))) return rc;
        hid.elu = nxt.elu = true;               // read only behind an ELU: by the block's second conv / the next layer
        prog.site("dec.res" + std::to_string(i));
        const ConvGemmArgs a1 = conv_args(m->dec_convs[ci++], up, 0, T, hid, 0, B, true);
        const ConvW& w2 = m->dec_convs[ci++];
        ConvGemmArgs a = conv_args(w2, hid, 0, T, nxt, nxt.H, B, true);
        a.res = up.p; a.res_ld = up.ld; a.res_off = up.H;
        if ((rc = add_resblock(m, prog, a1, a, w2))) return rc;
        if (Hn > 0) hist.push_back(hist_of(nxt, T));
        cur = nxt;
        mult /= 2;
    }
    if (T != c.frame_size) return mmi_fail(MMI_ERR_UNSUPPORTED, "decoder does not reproduce frame_size samples");
    if ((rc = alloc_buf(m, B, c.channels, 0, T, &m->dec_out, s0))) return rc;
    prog.site("dec.final");
    if ((rc = add_conv(m, prog, conv_args(m->dec_convs[ci++], cur, 0, T, m->dec_out, 0, B, true)))) return rc;
    prog.site("dec.commit");
    HistTable tab;
    if ((rc = upload_hist(m, hist, B, &m->dec_hist, &m->dec_nhist, &m->dec_hist_rows, &tab))) return rc;
    {
        HistDesc* hd = m->dec_hist; int nh = m->dec_nhist, rows = m->dec_hist_rows;
        const uint8_t* ex = m->exec; long* cnt = m->counters + B;
        const bool one = tab.n >= 0 && !getenv("MMI_MIMI_TWO_COMMITS");
        prog.add([=](hipStream_t s) {
            if (one) {
                MMI_LAUNCH(k_commit_all, mmi_cdiv(rows + B, 256), 256, 0, s, tab, rows, ex, cnt, 1, T_tr, (uint8_t*)nullptr, B);
                MMI_CHECK_LAUNCH();
                return (int)MMI_OK;
            }
            if (rows > 0) MMI_LAUNCH(k_commit_history, mmi_cdiv(rows, 256), 256, 0, s, (const HistDesc*)hd, nh, rows, ex);
            MMI_LAUNCH(k_commit_counters, mmi_cdiv(B, 64), 64, 0, s, cnt, 1, T_tr, (uint8_t*)nullptr, ex, B);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    return MMI_OK;
}

int check_cfg(const mmi_mimi_cfg& c) {
    if (c.n_ratios < 1 || c.n_ratios > 8) return mmi_fail(MMI_ERR_UNSUPPORTED, "n_ratios out of range");
    int hop = 1;
    for (int i = 0; i < c.n_ratios; ++i) hop *= c.ratios[i];
    if (hop * c.resample_stride != c.frame_size) return mmi_fail(MMI_ERR_UNSUPPORTED, "frame_size != hop*stride");
    if (c.tr_d_model != c.dimension) return mmi_fail(MMI_ERR_UNSUPPORTED, "projected transformer (d_model != dimension)");
    const int Dh = c.tr_d_model / c.tr_num_heads;
    if (Dh * c.tr_num_heads != c.tr_d_model || (Dh & 3) || 256 % Dh != 0)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "head dim must be a multiple of 4 and divide 256");
    if (c.tr_d_model > 1024) return mmi_fail(MMI_ERR_UNSUPPORTED, "transformer width above the LayerNorm kernel's register budget");
    if (c.q_dimension % 4) return mmi_fail(MMI_ERR_UNSUPPORTED, "codebook dimension must be a multiple of 4");
    if (c.compress < 1 || c.q_n_q_semantic < 1 || c.q_n_q < c.q_n_q_semantic)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "bad quantizer/compress config");
    return MMI_OK;
}

int frame_count_ok(const mmi_mimi* m, int batch, int n_frames) {
    if (!m) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!m->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming: call mmi_mimi_streaming_start first");
    if (batch != m->batch) return mmi_fail(MMI_ERR_SHAPE, "batch size does not match the streaming batch");
    if (n_frames <= 0) return mmi_fail(MMI_ERR_SHAPE, "length must be a positive multiple of the frame size");
    return MMI_OK;
}

}  // namespace

// ===============================================================================================
// C ABI
// ===============================================================================================
extern "C" int mmi_mimi_create(const mmi_mimi_cfg* cfg, const mmi_tensor_desc* weights, int32_t n_weights,
                               int32_t max_batch, mmi_mimi** out) {
    if (!cfg || !weights || !out || max_batch <= 0) return mmi_fail(MMI_ERR_INVALID, "mmi_mimi_create: bad argument");
    int rc = check_cfg(*cfg);
    if (rc) return rc;
    mmi_mimi* m = new mmi_mimi();
    if (hipGetDevice(&m->device) != hipSuccess) m->device = -1;
    m->cfg = *cfg;
    m->max_batch = max_batch;
    m->n_codebooks = cfg->q_n_q < 8 ? cfg->q_n_q : 8;
    m->use_graph = mmi_graphs_enabled();
    MmiWeights W{weights, n_weights};
    const mmi_mimi_cfg& c = m->cfg;
    auto fail = [&](int code) { mmi_mimi_destroy(m); return code; };

    // ---- SEANet encoder (seanet.py:169-236): model.{0,3,6,...} strided, model.{1,4,...}.block.{1,3} resblocks
    {
        int idx = 0, mult = 1;
        ConvW cw;
        if ((rc = load_conv(m, W, "encoder.model.0.conv.conv", c.channels, c.n_filters, c.kernel_size, 1, true, &cw))) return fail(rc);
        m->enc_convs.push_back(cw);
        idx = 1;
        for (int i = 0; i < c.n_ratios; ++i) {
            const int ratio = c.ratios[c.n_ratios - 1 - i];
            const int ch = mult * c.n_filters, hid = ch / c.compress;
            std::string rb = "encoder.model." + std::to_string(idx) + ".block.";
            if ((rc = load_conv(m, W, rb + "1.conv.conv", ch, hid, c.residual_kernel_size, 1, true, &cw))) return fail(rc);
            m->enc_convs.push_back(cw);
            if ((rc = load_conv(m, W, rb + "3.conv.conv", hid, ch, 1, 1, true, &cw))) return fail(rc);
            m->enc_convs.push_back(cw);
            idx += 2;  // resblock, ELU
            if ((rc = load_conv(m, W, "encoder.model." + std::to_string(idx) + ".conv.conv", ch, 2 * ch, 2 * ratio, ratio, true, &cw))) return fail(rc);
            m->enc_convs.push_back(cw);
            idx += 1;
            mult *= 2;
        }
        idx += 1;  // ELU
        if ((rc = load_conv(m, W, "encoder.model." + std::to_string(idx) + ".conv.conv", mult * c.n_filters, c.dimension, c.last_kernel_size, 1, true, &cw))) return fail(rc);
        m->enc_convs.push_back(cw);
    }
    // ---- SEANet decoder (seanet.py:315-388)
    {
        int mult = 1 << c.n_ratios;
        ConvW cw;
        if ((rc = load_conv(m, W, "decoder.model.0.conv.conv", c.dimension, mult * c.n_filters, c.kernel_size, 1, true, &cw))) return fail(rc);
        m->dec_convs.push_back(cw);
        int idx = 1;
        for (int i = 0; i < c.n_ratios; ++i) {
            const int ratio = c.ratios[i];
            const int cin = mult * c.n_filters, cout = cin / 2, hid = cout / c.compress;
            idx += 1;  // ELU
            if ((rc = load_convtr(m, W, "decoder.model." + std::to_string(idx) + ".convtr.convtr", cin, cout, 2 * ratio, ratio, &cw))) return fail(rc);
            m->dec_convs.push_back(cw);
            idx += 1;
            std::string rb = "decoder.model." + std::to_string(idx) + ".block.";
            if ((rc = load_conv(m, W, rb + "1.conv.conv", cout, hid, c.residual_kernel_size, 1, true, &cw))) return fail(rc);
            m->dec_convs.push_back(cw);
            if ((rc = load_conv(m, W, rb + "3.conv.conv", hid, cout, 1, 1, true, &cw))) return fail(rc);
            m->dec_convs.push_back(cw);
            idx += 1;
            mult /= 2;
        }
        idx += 1;  // ELU
        if ((rc = load_conv(m, W, "decoder.model." + std::to_string(idx) + ".conv.conv", c.n_filters, c.channels, c.last_kernel_size, 1, true, &cw))) return fail(rc);
        m->dec_convs.push_back(cw);
    }
    if ((rc = load_transformer(m, W, "encoder_transformer", &m->enc_tr))) return fail(rc);
    if ((rc = load_transformer(m, W, "decoder_transformer", &m->dec_tr))) return fail(rc);
    // ---- resampling (resample.py:14-119)
    if ((rc = load_conv(m, W, "downsample.conv.conv.conv", c.dimension, c.dimension, 2 * c.resample_stride, c.resample_stride, false, &m->downsample))) return fail(rc);
    {
        const mmi_tensor_desc* w;
        if ((rc = need(W, "upsample.convtr.convtr.convtr.weight", 3, &w))) return fail(rc);
        if (w->shape[0] != c.dimension || w->shape[1] != 1 || w->shape[2] != 2 * c.resample_stride)
            return fail(mmi_fail(MMI_ERR_SHAPE, "shape mismatch for upsample weight"));
        if ((rc = copy_vec(m, w, c.dimension * 2 * c.resample_stride, &m->upsample_w))) return fail(rc);
    }
    if ((rc = load_rvq(m, W))) return fail(rc);
    // ---- RVQ scratch
    m->nchunk = mmi_cdiv(c.q_bins, MMI_RVQ_CHUNK);
    if (hipSuccess != m->wts.alloc(&m->xq, (size_t)max_batch * 2 * c.q_dimension) ||
        hipSuccess != m->wts.alloc(&m->best_d, (size_t)2 * m->nchunk * max_batch) ||
        hipSuccess != m->wts.alloc(&m->best_i, (size_t)2 * m->nchunk * max_batch) ||
        hipSuccess != m->wts.alloc(&m->codes_i32, (size_t)max_batch * c.q_n_q) ||
        hipSuccess != m->wts.alloc(&m->dec_codes_i32, (size_t)max_batch * c.q_n_q) ||
        hipSuccess != m->wts.alloc(&m->q2, (size_t)max_batch * 2 * c.q_dimension) ||
        hipSuccess != m->wts.alloc(&m->lat_tmp, (size_t)max_batch * c.dimension) ||
        hipSuccess != m->wts.alloc(&m->qin_bp, (size_t)mmi_cdiv(max_batch, 32) * m->q_in.Q * 256) ||
        hipSuccess != m->wts.alloc(&m->qout_bp, (size_t)mmi_cdiv(max_batch, 32) * m->q_out.Q * 256) ||
        hipSuccess != m->wts.alloc(&m->qin_part, (size_t)8 * m->q_in.Mt * 32 * 128) ||
        hipSuccess != m->wts.alloc(&m->qout_part, (size_t)8 * m->q_out.Mt * 32 * 128))
        return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (RVQ scratch)"));
    if (hipDeviceSynchronize() != hipSuccess) return fail(mmi_fail(MMI_ERR_HIP, "weight packing failed"));
    if (m->use_graph && hipStreamCreate(&m->cap_stream) != hipSuccess)
        return fail(mmi_fail(MMI_ERR_HIP, "hipStreamCreate failed"));
    *out = m;
    return MMI_OK;
}

extern "C" void mmi_mimi_destroy(mmi_mimi* m) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m) return;
    mmi_mimi_streaming_stop(m);
    m->wts.release();
    if (m->cap_stream) hipStreamDestroy(m->cap_stream);
    delete m;
}

extern "C" int mmi_mimi_set_num_codebooks(mmi_mimi* m, int32_t n) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (n < 1 || n > m->cfg.q_n_q) return mmi_fail(MMI_ERR_INVALID, "num_codebooks out of range");
    if (m->streaming) return mmi_fail(MMI_ERR_STATE, "set_num_codebooks while streaming");
    m->n_codebooks = n;
    return MMI_OK;
}

extern "C" int mmi_mimi_num_codebooks(const mmi_mimi* m) { return m ? m->n_codebooks : 0; }
extern "C" int mmi_mimi_device(const mmi_mimi* m) { return m ? m->device : -1; }

int64_t mmi_copy_launch_log(const std::vector<std::string>& log, char* buf, int64_t cap);   // api_common.hip

extern "C" int64_t mmi_mimi_launch_list(const mmi_mimi* m, int32_t which, char* buf, int64_t cap) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !m->streaming) return 0;
    const MmiProgram& p = which == 0 ? m->enc_prog : m->dec_prog;
    if (!p.logged()) return 0;
    return mmi_copy_launch_log(p.launch_log(), buf, cap);
}

extern "C" int mmi_mimi_get_cfg(const mmi_mimi* m, mmi_mimi_cfg* out) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    *out = m->cfg;
    return MMI_OK;
}

extern "C" int mmi_mimi_streaming_start(mmi_mimi* m, int32_t batch, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (m->streaming) return mmi_fail(MMI_ERR_STATE, "already streaming");  // streaming.py:113
    if (batch <= 0 || batch > m->max_batch) return mmi_fail(MMI_ERR_SHAPE, "batch exceeds max_batch");
    hipStream_t s = (hipStream_t)stream;
    m->batch = batch;
    int rc = MMI_OK;
    auto fail = [&](int code) { m->streaming = true; mmi_mimi_streaming_stop(m); return code; };
    if (hipSuccess != m->st.alloc(&m->exec, (size_t)batch) || hipSuccess != m->st.alloc(&m->first, (size_t)batch) ||
        hipSuccess != m->st.alloc(&m->counters, (size_t)2 * batch) || hipSuccess != m->st.alloc(&m->dec_k_dev, (size_t)1))
        return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (state)"));
    // every failure from here on goes through fail(): what was allocated / appended so far is released again
    m->dec_k_host = m->n_codebooks;
    m->dec_k_cur = m->n_codebooks;
    if (hipMemsetAsync(m->exec, 1, batch, s) != hipSuccess || hipMemsetAsync(m->first, 1, batch, s) != hipSuccess ||
        hipMemsetAsync(m->counters, 0, (size_t)2 * batch * sizeof(long), s) != hipSuccess ||
        hipMemcpyAsync(m->dec_k_dev, &m->dec_k_host, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess)
        return fail(mmi_fail(MMI_ERR_HIP, "initialising the streaming state failed"));
    if ((rc = build_encoder(m, batch, s))) return fail(rc);
    if ((rc = build_decoder(m, batch, s))) return fail(rc);
    if (hipStreamSynchronize(s) != hipSuccess) return fail(mmi_fail(MMI_ERR_HIP, "hipStreamSynchronize failed"));
    m->streaming = true;
    return MMI_OK;
}

extern "C" int mmi_mimi_streaming_stop(mmi_mimi* m) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!m->streaming) return MMI_OK;
    hipDeviceSynchronize();
    m->enc_prog.clear();
    m->dec_prog.clear();
    m->st.release();
    m->partials.clear();
    m->partial_sizes.clear();
    m->streaming = false;
    m->batch = 0;
    return MMI_OK;
}

extern "C" int mmi_mimi_streaming_batch(const mmi_mimi* m) { return m && m->streaming ? m->batch : 0; }

extern "C" int64_t mmi_mimi_state_bytes(const mmi_mimi* m) { return m && m->streaming ? (int64_t)m->st.bytes : 0; }

extern "C" int mmi_mimi_state_save(mmi_mimi* m, void* dst, int64_t bytes, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !dst) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!m->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    if (bytes != (int64_t)m->st.bytes) return mmi_fail(MMI_ERR_SHAPE, "snapshot buffer has the wrong size");
    MMI_HIP_CHECK(m->st.save(dst, (hipStream_t)stream));
    return MMI_OK;
}

extern "C" int mmi_mimi_state_load(mmi_mimi* m, const void* src, int64_t bytes, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !src) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!m->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    if (bytes != (int64_t)m->st.bytes) return mmi_fail(MMI_ERR_SHAPE, "snapshot taken from a different stream (batch)");
    MMI_HIP_CHECK(m->st.load(src, (hipStream_t)stream));
    m->dec_k_cur = -1;        // the snapshot carries its own device-side K: the next decode re-uploads the caller's
    return MMI_OK;
}

extern "C" int mmi_mimi_set_exec_mask(mmi_mimi* m, const uint8_t* mask, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !mask) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!m->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    MMI_LAUNCH(k_set_mask, mmi_cdiv(m->batch, 64), 64, 0, (hipStream_t)stream, m->exec, mask, m->batch);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

extern "C" int mmi_mimi_reset(mmi_mimi* m, const uint8_t* mask, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!m->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    hipStream_t s = (hipStream_t)stream;
    const int B = m->batch;
    if (m->enc_hist_rows > 0)
        MMI_LAUNCH(k_reset_history, mmi_cdiv(m->enc_hist_rows, 256), 256, 0, s, (const HistDesc*)m->enc_hist, m->enc_nhist, m->enc_hist_rows, mask);
    if (m->dec_hist_rows > 0)
        MMI_LAUNCH(k_reset_history, mmi_cdiv(m->dec_hist_rows, 256), 256, 0, s, (const HistDesc*)m->dec_hist, m->dec_nhist, m->dec_hist_rows, mask);
    for (size_t i = 0; i < m->partials.size(); ++i)
        MMI_LAUNCH(k_reset_rows_f32, (int)mmi_cdiv64((int64_t)B * m->partial_sizes[i], 256), 256, 0, s, m->partials[i], m->partial_sizes[i], B, mask);
    MMI_LAUNCH(k_reset_counters, mmi_cdiv(B, 64), 64, 0, s, m->counters, 2, m->first, m->exec, B, mask);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

static int encode_impl(mmi_mimi* m, const float* pcm, int64_t* codes, float* latent, int32_t batch, int32_t n_frames,
                       hipStream_t s) {
    int rc = frame_count_ok(m, batch, n_frames);
    if (rc) return rc;
    if (!pcm) return mmi_fail(MMI_ERR_INVALID, "null pcm");
    const mmi_mimi_cfg& c = m->cfg;
    const int F = c.frame_size, K = m->n_codebooks;
    for (int f = 0; f < n_frames; ++f) {
        // stage the frame behind the first conv's history
        MMI_LAUNCH(k_copy2d_f32, (int)mmi_cdiv64((int64_t)batch * c.channels * F, 256), 256, 0, s, pcm + (long)f * F,
                   (long)n_frames * F, m->enc_in.p + m->enc_in.H, (long)m->enc_in.ld, batch * c.channels, F);
        if ((rc = m->enc_prog.run(s, m->use_graph, m->cap_stream))) return rc;
        if (codes)
            MMI_LAUNCH(k_codes_out, mmi_cdiv(batch * K, 256), 256, 0, s, (const int*)m->codes_i32, c.q_n_q, (long*)codes, batch, K, n_frames, f);
        if (latent)
            MMI_LAUNCH(k_copy2d_f32, mmi_cdiv(batch * c.dimension, 256), 256, 0, s, (const float*)m->latent.p, 1L, latent + f,
                       (long)n_frames, batch * c.dimension, 1);
        MMI_CHECK_LAUNCH();
    }
    return MMI_OK;
}

extern "C" int mmi_mimi_encode_step(mmi_mimi* m, const float* pcm, int64_t* codes, int32_t batch, int32_t n_frames,
                                    mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!codes) return mmi_fail(MMI_ERR_INVALID, "null codes");
    return encode_impl(m, pcm, codes, nullptr, batch, n_frames, (hipStream_t)stream);
}

extern "C" int mmi_mimi_encode_latent_step(mmi_mimi* m, const float* pcm, float* latent, int32_t batch, int32_t n_frames,
                                           mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!latent) return mmi_fail(MMI_ERR_INVALID, "null latent");
    return encode_impl(m, pcm, nullptr, latent, batch, n_frames, (hipStream_t)stream);
}

extern "C" int mmi_mimi_quantize(mmi_mimi* m, const float* latent, int64_t* codes, int32_t batch, int32_t n_frames,
                                 mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !latent || !codes) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (batch <= 0 || batch > m->max_batch || n_frames <= 0) return mmi_fail(MMI_ERR_SHAPE, "bad batch / frames");
    hipStream_t s = (hipStream_t)stream;
    const mmi_mimi_cfg& c = m->cfg;
    for (int f = 0; f < n_frames; ++f) {
        MmiProgram prog;
        int rc = add_quantize_ops(m, prog, latent, n_frames, f, batch);
        if (rc) return rc;
        rc = prog.run_eager(s);
        if (rc) return rc;
        MMI_LAUNCH(k_codes_out, mmi_cdiv(batch * m->n_codebooks, 256), 256, 0, s, (const int*)m->codes_i32, c.q_n_q, (long*)codes, batch,
                   m->n_codebooks, n_frames, f);
        MMI_CHECK_LAUNCH();
    }
    return MMI_OK;
}

extern "C" int mmi_mimi_decode_latent(mmi_mimi* m, const int64_t* codes, float* latent, int32_t batch, int32_t n_codebooks,
                                      int32_t n_frames, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    if (!m || !latent || !codes) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (batch <= 0 || batch > m->max_batch || n_frames <= 0 || n_codebooks < 1 || n_codebooks > m->cfg.q_n_q)
        return mmi_fail(MMI_ERR_SHAPE, "bad batch / codebooks / frames");
    hipStream_t s = (hipStream_t)stream;
    const mmi_mimi_cfg& c = m->cfg;
    Buf out; out.p = latent; out.C = c.dimension; out.ld = n_frames; out.H = 0;
    for (int f = 0; f < n_frames; ++f) {
        MMI_LAUNCH(k_codes_in, mmi_cdiv(batch * n_codebooks, 256), 256, 0, s, (const long*)codes, (long)n_codebooks * n_frames, m->dec_codes_i32,
                   c.q_n_q, batch, n_codebooks, n_frames, f);
        MmiProgram prog;
        int rc = add_dequant_ops(m, prog, n_codebooks, out, f, batch);
        if (rc) return rc;
        rc = prog.run_eager(s);
        if (rc) return rc;
    }
    return MMI_OK;
}

extern "C" int mmi_mimi_decode_step(mmi_mimi* m, const int64_t* codes, float* pcm, int32_t batch, int32_t n_codebooks,
                                    int32_t n_frames, mmi_stream stream) {
    return mmi_mimi_decode_step_strided(m, codes, (int64_t)n_codebooks * n_frames, pcm, batch, n_codebooks, n_frames, stream);
}

extern "C" int mmi_mimi_decode_step_strided(mmi_mimi* m, const int64_t* codes, int64_t batch_stride, float* pcm, int32_t batch,
                                            int32_t n_codebooks, int32_t n_frames, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(m ? m->device : -1);
    int rc = frame_count_ok(m, batch, n_frames);
    if (rc) return rc;
    if (!codes || !pcm) return mmi_fail(MMI_ERR_INVALID, "null argument");
    // any 1 <= K <= n_q: "the split RVQ decodes however many codebooks it is given" (compression.py:406-429, vq.py:281-287)
    if (n_codebooks < 1 || n_codebooks > m->cfg.q_n_q) return mmi_fail(MMI_ERR_SHAPE, "codes must carry between 1 and n_q codebooks");
    if (batch_stride < (int64_t)n_codebooks * n_frames) return mmi_fail(MMI_ERR_SHAPE, "batch stride smaller than one row of codes");
    hipStream_t s = (hipStream_t)stream;
    const mmi_mimi_cfg& c = m->cfg;
    const int F = c.frame_size;
    if (n_codebooks != m->dec_k_cur) {                  // the captured program reads K from device memory
        m->dec_k_host = n_codebooks;
        MMI_HIP_CHECK(hipMemcpyAsync(m->dec_k_dev, &m->dec_k_host, sizeof(int), hipMemcpyHostToDevice, s));
        MMI_HIP_CHECK(hipStreamSynchronize(s));         // the host word is reused; K changes at most once per model in practice
        m->dec_k_cur = n_codebooks;
    }
    for (int f = 0; f < n_frames; ++f) {
        MMI_LAUNCH(k_codes_in, mmi_cdiv(batch * n_codebooks, 256), 256, 0, s, (const long*)codes, (long)batch_stride, m->dec_codes_i32, c.q_n_q,
                   batch, n_codebooks, n_frames, f);
        MMI_CHECK_LAUNCH();
        if ((rc = m->dec_prog.run(s, m->use_graph, m->cap_stream))) return rc;
        MMI_LAUNCH(k_copy2d_f32, (int)mmi_cdiv64((int64_t)batch * c.channels * F, 256), 256, 0, s, (const float*)m->dec_out.p, (long)m->dec_out.ld,
                   pcm + (long)f * F, (long)n_frames * F, batch * c.channels, F);
        MMI_CHECK_LAUNCH();
    }
    return MMI_OK;
}
