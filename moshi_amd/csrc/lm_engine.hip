// Moshi LM engine behind the C ABI (include/moshi_mi.h): LMGen.step (reference: moshi/moshi/models/lm.py:668-850)
// = delay-ring bookkeeping + LMModel.forward_text (temporal transformer, lm.py:379-408) + text sampling +
// depformer_step (8 sequential depth-transformer micro-steps, lm.py:809-850, 450-493) as one fixed launch list over
// the kernels of lm_kernels.h, captured into a single hipGraph.
#include "lm_kernels.h"
#include "mmi_graph.h"

#include <math.h>
#include <stdio.h>
#include <map>

namespace {

struct GemmW {              // one packed nn.Linear
    u32x4* wp = nullptr;
    int N = 0, K = 0, NT = 0, KSTEPS = 0, gate = 0;   // gate: N = hidden, gate/value rows interleaved per tile
    float* scale = nullptr;   // int8 weights: SCB / 127 per original weight row; fp8: weight_scale * input_scale; KSTEPS then counts k-step PAIRS
    float* scb = nullptr;     // int8 weights: the raw SCB (`weight_scb`), for the int8 x int8 dequantisation (mmi_i8_dequant)
    int wq = 0;               // 0 bf16, 1 int8, 2 fp8
    float xinv = 1.f;         // fp8: 1 / input_scale
    size_t bytes = 0;
    int T = 32;               // the MFMA tile the weight is packed for (lm->T; the depth transformer's linears: lm->Td) - also the
                              // tile of the packed activation operand it reads and of a packed output it writes
};

struct LayerW {
    GemmW in_proj, out_proj, ffn_in, ffn_out;
    uint16_t *n1 = nullptr, *n2 = nullptr;
    // cross-attention block (cfg.cross_attention; transformer.py:727-732): `cross_attention.in_projs.0` packed whole, with two
    // views of it - the query rows [0, dim) and the key / value rows [dim, 3 dim) - `out_projs.0`, and `norm_cross`
    GemmW x_in, x_q, x_kv, x_out;
    uint16_t *nx_w = nullptr, *nx_b = nullptr;
};

struct DepLayerW {
    std::vector<GemmW> in_proj, out_proj, ffn_in, ffn_out;   // per step
    uint16_t *n1 = nullptr, *n2 = nullptr;                   // shared by the steps
};

struct EvPair { hipEvent_t a, b; };

}  // namespace

struct mmi_lm {
    mmi_lm_cfg cfg;
    int device = -1;                // HIP device the handle lives on (current at create); see MmiDeviceGuard
    int max_batch = 0;
    int T = 32;                     // MFMA tile of the temporal transformer and the heads: 16 when max_batch <= 16, else 32 (lm_kernels.h)
    int Td = 32;                    // MFMA tile of the depth transformer (its linears, dx / dxn / datt / dhb): T.  MMI_DEP_TILE=16 (17..32
                                    // sessions, bf16): two 16-row batch tiles per 16-row weight tile - measured in round 6 inside the step
                                    // (profiles/r06_logs/ab_dep_tile_once.txt): in_proj -0.14 us, linear_out -0.6, but the gated
                                    // linear_in +3.0 (352 tiles at 128 registers: spills) and out_proj +0.3 -> +0.15 ms per step; the
                                    // microbenchmark's -0.13 ms (round 5, independent launches) does not survive the dependent chain.
                                    // Kept as an opt-in: the boundaries are row-major either way (dpre, dqkv, the logits)
    int q8 = -1;                    // -1 undecided, 0 bf16 linears, 1 int8 linears (`weight` int8 + `weight_scb`, utils/quantize.py),
                                    // 2 fp8 linears (`weight` e4m3fn + `weight_scale` [+ `input_scale`]) run on the fp8 MFMA
    int NC = 0, CT = 0, max_delay = 0;
    MmiArena wts;
    // weights
    uint16_t* emb = nullptr;        // [n_q][card+1][dim]
    uint16_t* text_emb = nullptr;   // [text_card+1][dim]
    std::vector<LayerW> layers;
    uint16_t* out_norm = nullptr;
    GemmW text_linear;
    std::vector<GemmW> dep_in, dep_lin;
    GemmW dep_in_all;               // the dep_q depformer_in linears as one [dep_q * depformer_dim, dim] GEMM
    bool dep_in_grouped = false;
    std::vector<uint16_t*> dep_emb; // [0] = depformer_text_emb, [k>=1] = depformer_emb[k-1]
    std::vector<DepLayerW> dep_layers;
    // parity tap (mmi_lm_debug_linear): every packed linear / every norm vector by its state-dict key
    std::map<std::string, GemmW> linear_by_name;
    std::map<std::string, const uint16_t*> vector_by_name;
    int* delays_dev = nullptr;
    size_t weight_bytes = 0;
    // streaming state
    bool streaming = false;
    int batch = 0;                  // MODEL rows: the sessions, doubled under classifier-free guidance (lm.py:646-651)
    int gen_batch = 0;              // sessions (rows of the token ring, the samplers and the caller's tensors)
    float cfg_coef = 1.f;           // LMGen(cfg_coef, cfg_is_no_text, cfg_is_masked_until) (lm.py:566-574)
    int cfg_no_text = 0;
    int* masked_until = nullptr;    // [gen_batch] or null
    uint16_t* cond = nullptr;       // [batch][dim] summed `sum` conditions (lm.py:621-628) or null
    uint16_t* xkv = nullptr;        // [layers][batch * cross_len][2 * dim] keys | values of the cross-attention source
    int cross_len = 0;
    long* offsets_m = nullptr;      // [batch] offsets per model row (== offsets without guidance)
    std::vector<uint16_t*> extra_heads;   // nn.Linear(dim, extra_heads_dim) weights, row-major
    uint16_t* extra_heads_all = nullptr;
    mmi_sampling samp;
    int kmax = 0;
    MmiArena st;
    uint8_t* exec = nullptr;
    long* offsets = nullptr;
    int* cache = nullptr;
    int* user_i32 = nullptr;
    int* tokens = nullptr;          // [B][NC] model input
    int* text_tok = nullptr;        // [B]
    int* audio_tok = nullptr;       // [B][dep_q]
    int* out_i32 = nullptr;         // [B][dep_q+1]
    uint16_t *x = nullptr, *xn = nullptr, *qrot = nullptr, *att = nullptr, *hb = nullptr;
    uint16_t *tout = nullptr, *text_logits = nullptr;
    uint16_t *kc = nullptr, *vc = nullptr;          // [layers][B][H][cap][Dh]
    float *opart = nullptr, *ml = nullptr;
    unsigned* attn_done = nullptr;                  // [B][H] arrival counters of the split decode attention (k_lm_attn_wave), 0 between launches
    float* partial = nullptr;                       // [4][B][max(dim, depformer_dim)] split-K partial sums
    float* rope = nullptr;                          // [B][Dh/2][2] (cos, sin) of the step's new position
    // int8 activations (BASELINE configs[4], the reference's own arithmetic: QLinear.forward -> bitsandbytes' int8 x int8 matmul,
    // utils/quantize.py:24-40): on for int8 linears unless MMI_Q8_ACT=bf16 (weight-only, rounds 1-3) or the model has
    // cross-attention layers.  xnq / attq / hbq / toutq: the int8 operand Xq[mt][kp][lane][16] next to its bf16 tensor, sx_*: its
    // rows' absmax (bitsandbytes' SCA).  Who quantises: the norm kernel for its own output (one workgroup per row); a
    // k_quant_rows_i8 launch for the temporal attention output and the gated FFN tensor (rows written by many workgroups); the
    // depth transformer's GEMMs themselves (k_gemm_q8: every workgroup holds the whole short row).  No atomics: absmax slots
    // folded by the producing epilogues were built first and measured - 704 gated tiles x 64 rows hammering 2 cache lines took
    // linear_in from 50 to 94 us (58 with test-then-max), profiles/r04_logs/call_b_summary.txt.
    bool act8 = false;
    uint8_t *xnq = nullptr, *attq = nullptr, *hbq = nullptr, *toutq = nullptr, *dxnq = nullptr;
    float *sx_xn = nullptr, *sx_tout = nullptr, *sx_dxn = nullptr, *sx_att = nullptr, *sx_hb = nullptr;
    bool hidden_taps = false;                       // mmi_lm_set_hidden_taps: the next streaming_start adds the two copies below
    uint16_t* htap = nullptr;                       // [2][B][dim] residual stream after the first / the last temporal layer
    uint16_t *dx = nullptr, *dxn = nullptr, *dqkv = nullptr, *datt = nullptr, *dhb = nullptr, *dlogits = nullptr;
    uint16_t* dpre = nullptr;       // [B][dep_q * depformer_dim] depformer_in[k](transformer_out) of every micro-step
    uint16_t *dkc = nullptr, *dvc = nullptr;        // [dep_layers][B][Hd][dep_q][Dhd]
    float* noise = nullptr;                         // [B][1+dep_q][kmax]
    int* use_noise = nullptr;
    int* forced = nullptr;                          // [B][1+dep_q]
    int* use_forced = nullptr;
    bool forced_armed = false;
    bool noise_on = false;                          // host mirror of *use_noise
    unsigned long long* rng = nullptr;
    // per-step host hooks (mmi_lm_set_hooks): the step is cut at these ops of the launch list
    mmi_lm_hooks hooks{nullptr, nullptr, nullptr, nullptr};
    bool in_hook = false;
    size_t op_text_sample = 0, op_after_text_sample = 0, op_commit = 0;
    size_t op_depformer = 0;        // first op after the temporal transformer + text head: where mmi_lm_set_phase_event's event is recorded
    int (*phase_fn)(void*, mmi_stream) = nullptr;   // mmi_lm_set_phase_callback
    void* phase_user = nullptr;
    SampleArgs text_sample_args;    // to rebuild the depth transformer's first input when a hook changed the text token
    long offset_cpu = 0;
    int attn_ns = 1;                                // workgroups per (session, head) of the decode attention (attn_splits)
    long depth_bound = 0;                           // no session's offset exceeds this (steps since streaming_start / the last seek; resets
                                                    // only lower offsets): picks the step program's variant (attn_variant)
    long xlds_launches = 0;         // launches (or graph nodes captured) that took k_gemm_xlds: mmi_lm_stat(lm, 0)
    bool precapture_failed = false; // capturing the second attention program ahead of time failed once: not tried again for this stream
    bool dominant_xlds = false;     // the profiled (dominant) GEMM ran on k_gemm_xlds
    MmiProgram prog;
    // MMI_DEBUG_TRACE=<prefix> (debug: finding which launch of the step is not reproducible): every step runs its launch list
    // eagerly and, after EVERY op, checksums every allocation of the streaming state; lines "step op site allocation bytes
    // checksum" for the allocations an op changed go to <prefix>.<n> (n = streaming sessions of the process so far).  Two
    // sessions fed the same inputs must write the same file.
    FILE* trace_file = nullptr;
    std::string trace_name;
    unsigned long long* trace_dev = nullptr;
    std::vector<unsigned long long> trace_prev;
    long trace_step = 0;
    hipStream_t cap_stream = nullptr;
    bool use_graph = true;
    // profiling tap
    struct SiteEv { size_t op; hipEvent_t ev; };
    std::vector<SiteEv> site_ev;     // one event in front of every eagerly run op while profiling, one behind the step's last
                                     // (op = SIZE_MAX): an op's time is the distance to the next event (mmi_lm_profile_sites)
    size_t site_ev_used = 0;
    bool profiling = false;
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    hipStream_t prof_stream = nullptr;
    long prof_bytes = 0;
};

namespace {

int need(const MmiWeights& W, const std::string& name, int ndim, const mmi_tensor_desc** out) {
    const mmi_tensor_desc* d = W.find(name);
    if (!d) return mmi_fail(MMI_ERR_MISSING_WEIGHT, "missing weight: " + name);
    if (d->dtype != MMI_BF16) return mmi_fail(MMI_ERR_UNSUPPORTED, "LM weights must be bf16: " + name);
    if (d->ndim != ndim) return mmi_fail(MMI_ERR_SHAPE, "unexpected rank for " + name);
    *out = d;
    return MMI_OK;
}

// nn.Linear weight [N][K] -> packed; gate_hidden > 0: [2*hidden][K] gate|value matrix
__global__ void k_scb_to_scale(const float* __restrict__ scb, float* __restrict__ scale, int n, float mul, float div) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) scale[i] = scb[i] * mul / div;
}

// wp_dst / scale_dst: pack into a slice of a caller-owned allocation instead of a fresh one (see load_dep_in_group)
int load_linear(mmi_lm* lm, const MmiWeights& W, const std::string& name, int N, int K, int gate_hidden, GemmW* g,
                void* wp_dst = nullptr, float* scale_dst = nullptr, float* scb_dst = nullptr, int tile = 0) {
    const mmi_tensor_desc* d = W.find(name);
    if (!d) return mmi_fail(MMI_ERR_MISSING_WEIGHT, "missing weight: " + name);
    if (d->dtype != MMI_BF16 && d->dtype != MMI_I8 && d->dtype != MMI_F8E4M3)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "LM linear weights must be bf16, int8 or fp8 (e4m3fn): " + name);
    if (d->ndim != 2) return mmi_fail(MMI_ERR_SHAPE, "unexpected rank for " + name);
    const int q8 = d->dtype == MMI_I8 ? 1 : (d->dtype == MMI_F8E4M3 ? 2 : 0);
    if (lm->q8 >= 0 && lm->q8 != q8) return mmi_fail(MMI_ERR_UNSUPPORTED, "mixed bf16 / int8 / fp8 linear weights: " + name);
    lm->q8 = q8;
    if (d->shape[0] != N || d->shape[1] != K) return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + name);
    if (K % 8 != 0) return mmi_fail(MMI_ERR_UNSUPPORTED, "in_features must be a multiple of 8: " + name);
    const int TN = tile ? tile : lm->T;
    g->T = TN;
    g->K = K;
    g->gate = gate_hidden > 0 ? 1 : 0;
    g->N = gate_hidden > 0 ? gate_hidden : N;
    if (g->N % 8 != 0) return mmi_fail(MMI_ERR_UNSUPPORTED, "out_features must be a multiple of 8: " + name);
    const int rows_per_tile = gate_hidden > 0 ? TN / 2 : TN;
    g->NT = mmi_cdiv(g->N, rows_per_tile);
    const int ksteps = mmi_cdiv(K, mmi_kstep(TN));
    if (!q8) {
        g->KSTEPS = ksteps;
        size_t n = (size_t)g->NT * g->KSTEPS * 512;
        uint16_t* p = reinterpret_cast<uint16_t*>(wp_dst);
        if (!p) MMI_HIP_CHECK(lm->wts.alloc(&p, n));
        g->wp = reinterpret_cast<u32x4*>(p);
        g->bytes = n * sizeof(uint16_t);
        MMI_LAUNCH(k_pack_w_bf16, (int)mmi_cdiv64((int64_t)n, 256), 256, 0, (hipStream_t)0, (const uint16_t*)d->data, p, N, K,
                   TN, g->NT, g->KSTEPS, gate_hidden);
    } else {
        // int8: `<linear>.weight_scb` = row absmax (utils/quantize.py:20-22).  fp8: `<linear>.weight_scale` = dequantisation
        // factor per row, and an optional scalar `<linear>.input_scale` (static activation scale, default 1)
        const std::string stem = name.size() > 7 && name.compare(name.size() - 7, 7, ".weight") == 0 ? name.substr(0, name.size() - 7) : name;
        const std::string sname = q8 == 1 ? name + "_scb" : name + "_scale";
        const mmi_tensor_desc* sc = W.find(sname);
        if (!sc) return mmi_fail(MMI_ERR_MISSING_WEIGHT, "missing weight: " + sname);
        if (sc->dtype != MMI_F32 || sc->shape[0] != N) return mmi_fail(MMI_ERR_SHAPE, "row scales must be fp32 [out_features]: " + sname);
        float in_scale = 1.f;
        if (q8 == 2) {
            if (const mmi_tensor_desc* is = W.find(stem + ".input_scale")) {
                if (is->dtype != MMI_F32) return mmi_fail(MMI_ERR_SHAPE, "input_scale must be fp32: " + stem);
                MMI_HIP_CHECK(hipMemcpy(&in_scale, is->data, sizeof(float), hipMemcpyDeviceToHost));
                if (!(in_scale > 0.f)) return mmi_fail(MMI_ERR_INVALID, "input_scale must be positive: " + stem);
            }
        }
        g->wq = q8;
        g->xinv = 1.f / in_scale;
        g->KSTEPS = mmi_cdiv(ksteps, 2);                       // pairs of k-steps
        size_t n = (size_t)g->NT * g->KSTEPS * 1024;
        int8_t* p = reinterpret_cast<int8_t*>(wp_dst);
        if (!p) MMI_HIP_CHECK(lm->wts.alloc(&p, n));
        g->wp = reinterpret_cast<u32x4*>(p);
        g->scale = scale_dst;
        if (!g->scale) MMI_HIP_CHECK(lm->wts.alloc(&g->scale, (size_t)N));
        if (q8 == 1) {
            g->scb = scb_dst;
            if (!g->scb) MMI_HIP_CHECK(lm->wts.alloc(&g->scb, (size_t)N));
            MMI_HIP_CHECK(hipMemcpy(g->scb, sc->data, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice));
        }
        g->bytes = n + (size_t)N * sizeof(float);
        MMI_LAUNCH(k_pack_w_i8, (int)mmi_cdiv64((int64_t)n, 256), 256, 0, (hipStream_t)0, (const int8_t*)d->data, p, N, K, TN,
                   g->NT, g->KSTEPS, gate_hidden);
        MMI_LAUNCH(k_scb_to_scale, mmi_cdiv(N, 256), 256, 0, (hipStream_t)0, (const float*)sc->data, g->scale, N,
                   q8 == 1 ? 1.f : in_scale, q8 == 1 ? 127.f : 1.f);
    }
    lm->weight_bytes += g->bytes;
    lm->linear_by_name[name] = *g;
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

int load_copy(mmi_lm* lm, const MmiWeights& W, const std::string& name, int ndim, size_t n, uint16_t** out, uint16_t* into = nullptr) {
    const mmi_tensor_desc* d;
    int rc = need(W, name, ndim, &d);
    if (rc) return rc;
    size_t have = 1;
    for (int i = 0; i < d->ndim; ++i) have *= (size_t)d->shape[i];
    if (have != n) return mmi_fail(MMI_ERR_SHAPE, "shape mismatch for " + name);
    uint16_t* dst = into;
    if (!dst) MMI_HIP_CHECK(lm->wts.alloc(&dst, n));
    MMI_HIP_CHECK(hipMemcpy(dst, d->data, n * sizeof(uint16_t), hipMemcpyDeviceToDevice));
    if (out) *out = dst;
    lm->vector_by_name[name] = dst;
    return MMI_OK;
}

// ---- launch helpers ---------------------------------------------------------------------------
struct GemmPlan { int waves, ntw, ksplit, u, osplit; };

// How a GEMM is cut into workgroups (measured on MI355X with scripts/gemm_microbench.hip): enough workgroups to
// cover the 256 CUs, K split over the waves of a workgroup, more waves per workgroup when there are few n-tiles.
GemmPlan plan_gemm(const GemmW& g, bool may_split) {
    GemmPlan p;
    p.ntw = 1;
    p.ksplit = 1;
    // few n-tiles (N = 4096 at the 32-row tile): split K over workgroups so that every CU streams weights
    if (may_split && g.NT < 200 && g.KSTEPS >= 128) p.ksplit = g.NT <= 64 ? 4 : 2;
    if (const char* ek = getenv("MMI_GEMM_KSPLIT")) {      // test hook: force the split-K path on small shapes
        const int v = atoi(ek);
        if (may_split && v >= 1 && v <= 4 && g.KSTEPS >= v) p.ksplit = v;
    }
    const int ks = g.KSTEPS / p.ksplit;
    p.waves = ks >= 32 ? 8 : 4;
    // fragments in flight per register buffer: 2 for the widest GEMM (the temporal FFN linear_in, 704 n-tiles: fewer
    // registers -> 3 workgroups per CU -> all 704 resident at once; 36.8 vs 39.3 us in the microbenchmark), else 4
    p.u = (g.gate && g.NT >= 512 && p.waves == 8) ? 2 : 4;
    if (g.scale) p.u = 2;            // int8 entries carry two k-steps each
    // (two n-tiles per workgroup for the 384-tile temporal in_proj look 3.5 us faster in the microbenchmark - 22.5 against
    // 26.0 us, profiles/r01_logs/gemm_microbench_b32_v10.txt - and make no difference in the step: 8.176 / 8.186 ms against
    // 8.190 / 8.176 ms in a same-box A/B, profiles/r01_logs/ab_in_proj_ntw2.txt; not adopted)
    p.osplit = 1;
    return p;
}

// Octet sharing of k_gemm_xp (GemmArgs::osplit): GEMMs with so few n-tiles that most CUs would idle while each busy one is
// bound by what a single CU can pull (~25 GB/s) - the depth transformer's N = 1024 linears: 32 tiles of 64-180 KB.
// MMI_GEMM_OSPLIT: "0" = off, "2" / "4" = force (test hook / A-B), default = as many parts as bring the launch to >= 128 workgroups.
int plan_osplit(const GemmW& g, const GemmPlan& p, int epi, int T) {
    if (epi == MMI_EPI_GATE || g.wq != 0 || p.ntw != 1 || (T != 32 && T != 16)) return 1;
    const int octs = T / 8;
    int os = 1;
    const char* e = getenv("MMI_GEMM_OSPLIT");
    if (e && e[0]) {
        const int v = atoi(e);
        if (v <= 1) return 1;
        os = v < octs ? v : octs;
        if (e[1] == 'a') return os;            // "4a": every eligible GEMM (tests)
        return (long)g.NT * p.ksplit <= 64 ? os : 1;
    }
    while (os < octs && (long)g.NT * p.ksplit * os < 128) os *= 2;
    return (long)g.NT * p.ksplit <= 64 ? os : 1;
}

template <int TN, int MT, int NTW, int WQ>
int launch_gemm_q(hipStream_t s, dim3 groups, int waves, const GemmArgs& a) {
    if constexpr (MT * NTW <= 2) {
        if (waves == 8) MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 8, 2, WQ>), groups, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 4, 2, WQ>), groups, 256, 0, s, a);
    } else {
        MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 4, 2, WQ>), groups, 256, 0, s, a);
    }
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

template <int TN, int MT, int NTW>
int launch_gemm_w(hipStream_t s, dim3 groups, int waves, int u, int wq, const GemmArgs& a) {
    if (wq == 1) return launch_gemm_q<TN, MT, NTW, 1>(s, groups, waves, a);
    if (wq == 2) return launch_gemm_q<TN, MT, NTW, 2>(s, groups, waves, a);
    if (wq == 3) {
        // int8 x int8 has no conversion to hold in registers: four entries per register buffer instead of two (MMI_Q8_U=2: the
        // weight-only depth, same-box A/B)
        static const bool u4 = !(getenv("MMI_Q8_U") && getenv("MMI_Q8_U")[0] == '2');
        if (u4 && waves == 8) {
            if constexpr (MT * NTW <= 2) {
                MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 8, 4, 3>), groups, 512, 0, s, a);
                MMI_CHECK_LAUNCH();
                return MMI_OK;
            }
        }
        return launch_gemm_q<TN, MT, NTW, 3>(s, groups, waves, a);
    }
    if (waves == 8 && u == 2 && MT * NTW == 1) {
        if constexpr (MT * NTW == 1) MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 8, 2>), groups, 512, 0, s, a);
        MMI_CHECK_LAUNCH();
        return MMI_OK;
    }
    switch (waves) {
        case 4: MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 4, 4>), groups, 256, 0, s, a); break;
        case 8:
            if constexpr (MT * NTW <= 2) { MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 8, 4>), groups, 512, 0, s, a); break; }
            MMI_LAUNCH((k_gemm_xp<TN, MT, NTW, 4, 4>), groups, 256, 0, s, a); break;
        default: return mmi_fail(MMI_ERR_UNSUPPORTED, "unsupported waves per GEMM workgroup");
    }
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

template <int TN>
int launch_gemm_t(hipStream_t s, const GemmPlan& p, int NT, int mt, const GemmArgs& a) {
    const dim3 groups(mmi_cdiv(NT, p.ntw) * (a.osplit > 1 ? a.osplit : 1), p.ksplit);
    const int w8 = a.wq;
    if (mt == 1) return launch_gemm_w<TN, 1, 1>(s, groups, p.waves, p.u, w8, a);     // one n-tile per workgroup (two measured slower
    if (mt == 2) return launch_gemm_w<TN, 2, 1>(s, groups, p.waves, p.u, w8, a);     // in the step, DESIGN 9e: not instantiated)
    return mmi_fail(MMI_ERR_UNSUPPORTED, "batch too large for the skinny GEMM");
}

// k_gemm_xlds (activations resident in LDS, one workgroup per CU): which GEMMs take it, and how.  MMI_GEMM_LDS: "2" = with
// the staggered tail (each tile's epilogue under the last chunk's weight stream; the DEFAULT for bf16 weights at the 32-row
// tile: same-box A/B of the default benchmark 8.10 / 8.12 ms against 8.17 / 8.20 ms with k_gemm_xp, dominant kernel 38.6
// against 40.2 us live, profiles/r02_logs/ab_gemm_xlds_stagger_in_step.txt), "1" = plain tails (also what int8 / fp8 weight
// entries take when asked for; they stay on k_gemm_xp by default), "0" = k_gemm_xp everywhere.
struct XldsPlan { bool on; int kc, grid; size_t smem; bool stagger; };
XldsPlan plan_xlds(const mmi_lm* lm, const GemmW& g, const GemmArgs& a, int mt) {
    XldsPlan p{false, 0, 0, 0, false};
    const char* en = getenv("MMI_GEMM_LDS");
    const char mode = en && en[0] ? en[0] : (g.wq == 0 ? '2' : '0');
    if (mode == '0' || g.T != 32 || mt > 2) return p;
    if (a.epi != MMI_EPI_GATE && a.epi != MMI_EPI_ROPE_KV && a.epi != MMI_EPI_STORE) return p;   // no prefetched addend, no split-K
    int cus = 256;                                             // MI355X: 256 CUs
    const char* tg = getenv("MMI_GEMM_LDS_GRID");              // test hook: small grids / short chunks for the tiny shapes
    if (tg && atoi(tg) > 0) cus = atoi(tg);
    const int xs = (g.wq && a.wq != 3) ? 2 : 1;                // activation fragments per weight entry (int8 activations: one entry)
    const int big = ((g.wq && a.wq != 3) ? 32 : 64) / mt;      // 64 KiB of activations per chunk buffer (int8 activations: 1 KiB per entry)
    if (g.KSTEPS % big == 0 && g.NT >= 128) p.kc = big;       // the large temporal GEMMs
    else if (tg && g.KSTEPS % 8 == 0) p.kc = 8;
    else if (tg && g.KSTEPS % 4 == 0) p.kc = 4;
    else return p;
    p.grid = g.NT < cus ? g.NT : cus;
    if (mmi_cdiv(g.NT, p.grid) > 3) return p;
    if (a.epi != MMI_EPI_GATE) {   // the kernel shares the tiles out in row octets: no workgroup may touch more than 3 tiles
        for (long b = 0; b < p.grid; ++b) {
            const long u0 = b * 4L * g.NT / p.grid, u1 = (b + 1) * 4L * g.NT / p.grid;
            if (u1 > u0 && ((u1 + 3) >> 2) - (u0 >> 2) > 3) return p;
        }
    }
    p.stagger = mode == '2' && g.wq == 0;                     // per-tile epilogues under the last chunk's stream (bf16)
    const size_t chunks = (size_t)2 * mt * p.kc * xs * 1024, red = mt == 1 ? 40960 : 65536;   // red: the epilogue's reduction scratch
    p.smem = chunks > red ? chunks : red;
    if (p.stagger && chunks < 131072) p.smem = chunks + red; // short test chunks: the scratch sits behind both buffers
    p.on = true;
    return p;
}

template <int MT, int KC, bool STAGGER, int WQ>
int launch_xlds_v(hipStream_t s, const XldsPlan& p, const GemmArgs& a) {
    static bool attr_set = false;
    if (!attr_set) {
        MMI_HIP_CHECK(hipFuncSetAttribute((const void*)k_gemm_xlds<MT, KC, 3, STAGGER, WQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        attr_set = true;
    }
    MMI_LAUNCH((k_gemm_xlds<MT, KC, 3, STAGGER, WQ>), p.grid, 512, p.smem, s, a);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}
// production chunk (64 / MT k-steps, or 32 / MT two-step entries) or one of the short test chunks
template <int MT, bool STAGGER, int WQ>
int launch_xlds_kc(hipStream_t s, const XldsPlan& p, const GemmArgs& a) {
    constexpr int BIG = ((WQ == 1 || WQ == 2) ? 32 : 64) / MT;
    if (p.kc == BIG) return launch_xlds_v<MT, BIG, STAGGER, WQ>(s, p, a);
    if (p.kc == 8) return launch_xlds_v<MT, 8, STAGGER, WQ>(s, p, a);
    if (p.kc == 4) return launch_xlds_v<MT, 4, STAGGER, WQ>(s, p, a);
    return mmi_fail(MMI_ERR_UNSUPPORTED, "k_gemm_xlds: unsupported chunk");
}
template <int MT>
int launch_xlds(hipStream_t s, const XldsPlan& p, const GemmArgs& a) {
    if (a.wq == 1) return launch_xlds_kc<MT, false, 1>(s, p, a);
    if (a.wq == 2) return launch_xlds_kc<MT, false, 2>(s, p, a);
    if (a.wq == 3) return launch_xlds_kc<MT, false, 3>(s, p, a);
    return p.stagger ? launch_xlds_kc<MT, true, 0>(s, p, a) : launch_xlds_kc<MT, false, 0>(s, p, a);
}

// k_gemm_xp_once (a wave's whole K-slice requested before its first MFMA) for the GEMMs k_gemm_xp would walk in three or more
// dependent memory round trips although they are a latency chain, not a stream: bf16 weights behind at most 256 workgroups of 8
// waves whose slice is longer than the two register buffers of k_gemm_xp (2 x 4 k-steps) and short enough for the registers -
// the depth transformer's linear_out (2816 -> 1024: 22 k-steps per wave at the 32-row tile, 11 at the 16-row tile).
// Returns the instantiated slice length, 0 = k_gemm_xp.  MMI_GEMM_ONCE=0: off (same-box A/B; bit-identical either way); "a": every
// bf16 GEMM whose slices fit (test hook: the tiny shapes, always on 8 waves).
int once_kmax(const GemmW& g, const GemmPlan& p, int mt, const GemmArgs& a) {
    const char* e = getenv("MMI_GEMM_ONCE");
    const bool off = e && e[0] == '0', all = e && e[0] == 'a';
    if (off || a.wq != 0 || p.ntw != 1 || a.epi == MMI_EPI_GATE) return 0;
    const int kper = mmi_cdiv(mmi_cdiv(g.KSTEPS, p.ksplit), 8);
    if (!all) {
        if (p.waves != 8 || p.ksplit != 1) return 0;   // (split-K GEMMs are streams: the temporal out_proj lost 0.9 us per launch on this form)
        if ((long)g.NT * p.ksplit * (a.osplit > 1 ? a.osplit : 1) > 256) return 0;      // a stream: keep the double-buffered loop
        if (kper <= 2 * p.u) return 0;                                                   // already one round trip
    }
    if (g.T == 32 && mt == 1) return kper <= 11 ? 11 : (kper <= 22 ? 22 : 0);
    if (g.T == 16 && mt <= 2) return kper <= 11 ? 11 : 0;
    return 0;
}
int launch_once(hipStream_t s, int T, int mt, int kmax, dim3 groups, const GemmArgs& a) {
    if (T == 32 && kmax == 11) MMI_LAUNCH((k_gemm_xp_once<32, 1, 8, 11>), groups, 512, 0, s, a);
    else if (T == 32) MMI_LAUNCH((k_gemm_xp_once<32, 1, 8, 22>), groups, 512, 0, s, a);
    else if (mt == 1) MMI_LAUNCH((k_gemm_xp_once<16, 1, 8, 11>), groups, 512, 0, s, a);
    else MMI_LAUNCH((k_gemm_xp_once<16, 2, 8, 11>), groups, 512, 0, s, a);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

int launch_gemm(mmi_lm* lm, hipStream_t s, const GemmW& g, GemmArgs a, bool is_dominant) {
    a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT;
    a.wscale = g.scale; a.wscb = g.scb; a.gate_rows = g.gate ? g.N : 0;
    a.wq = (g.wq == 1 && a.wq >= 3) ? a.wq : g.wq;             // int8 linears: 3 / 4 = int8 activations (set by the program builder)
    a.xinv = g.xinv;
    const int mt = mmi_cdiv(a.B, g.T);
    const GemmPlan p = plan_gemm(g, a.epi == MMI_EPI_PARTIAL);
    a.osplit = plan_osplit(g, p, a.epi, g.T);
    const XldsPlan xl = plan_xlds(lm, g, a, mt);
    EvPair* ev = nullptr;
    if (lm->profiling && is_dominant) {
        if (lm->ev_used == lm->ev_pool.size()) {
            EvPair pr;
            MMI_HIP_CHECK(hipEventCreate(&pr.a));
            MMI_HIP_CHECK(hipEventCreate(&pr.b));
            lm->ev_pool.push_back(pr);
        }
        ev = &lm->ev_pool[lm->ev_used++];
        lm->prof_stream = s;
        MMI_HIP_CHECK(hipEventRecord(ev->a, s));
    }
    int rc;
    mmi_record_bytes((long)g.bytes);
    if (is_dominant) lm->dominant_xlds = xl.on;
    if (xl.on) {
        lm->xlds_launches += 1;
        rc = mt == 1 ? launch_xlds<1>(s, xl, a) : launch_xlds<2>(s, xl, a);
    } else if (const int kmax = once_kmax(g, p, mt, a)) {
        rc = launch_once(s, g.T, mt, kmax, dim3(g.NT * (a.osplit > 1 ? a.osplit : 1), p.ksplit), a);
    } else {
        rc = g.T == 32 ? launch_gemm_t<32>(s, p, g.NT, mt, a) : launch_gemm_t<16>(s, p, g.NT, mt, a);
    }
    if (rc) return rc;
    if (ev) MMI_HIP_CHECK(hipEventRecord(ev->b, s));
    return MMI_OK;
}

// number of bf16 elements of a packed activation buffer with `features` columns
// k-steps of a packed activation buffer with `features` columns (even when the linears are int8: their weight entries
// carry k-step pairs, and the padding k-step reads as zero)
int packed_ksteps_t(const mmi_lm* lm, int T, int features) {
    const int ks = mmi_cdiv(features, mmi_kstep(T));
    return lm->q8 >= 1 ? 2 * mmi_cdiv(ks, 2) : ks;
}
int packed_ksteps(const mmi_lm* lm, int features) { return packed_ksteps_t(lm, lm->T, features); }
size_t packed_elems_t(const mmi_lm* lm, int T, int features) {
    return (size_t)mmi_cdiv(lm->batch, T) * packed_ksteps_t(lm, T, features) * 512;
}
size_t packed_elems(const mmi_lm* lm, int features) { return packed_elems_t(lm, lm->T, features); }

// x: packed activations.  out: packed with `out_features` columns (out_packed) or row-major with leading dim out_features.
// MMI_EPI_DEP_QKV0 (the depth transformer's in_proj at micro-step 0): where its epilogue writes k / v (frame cache, position 0)
struct DepKv { uint16_t* kc; uint16_t* vc; int H, Dh, steps; };
// int8 activations of one GEMM (lm->act8): `x` is the int8 operand Xq and sx its rows' absmax (wq 3)
struct Q8 { int wq = 0; const float* sx = nullptr; };

void add_gemm(mmi_lm* lm, const GemmW& g, const uint16_t* x, uint16_t* out, int out_features, bool out_packed, int epi,
              const uint16_t* resid, const uint16_t* emb = nullptr, const int* tok = nullptr, int tok_stride = 0,
              bool dominant = false, const DepKv* kv = nullptr, const Q8* q8 = nullptr) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    if (kv) { a.kc = kv->kc; a.vc = kv->vc; a.H = kv->H; a.Dh = kv->Dh; a.cap = kv->steps; }
    if (q8) { a.wq = q8->wq; a.sx = q8->sx; }
    a.xp = reinterpret_cast<const u32x4*>(x); a.out = out; a.epi = epi; a.resid = resid; a.emb = emb; a.tok = tok;
    a.tok_stride = tok_stride; a.tok_rows = lm->gen_batch; a.B = lm->batch;
    a.out_mode = out_packed ? MMI_OUT_PACKED : MMI_OUT_ROWMAJOR;
    a.out_ld = out_features;
    a.out_ksteps = packed_ksteps_t(lm, g.T, out_features);
    GemmW gw = g;
    lm->prog.add([lm, gw, a, dominant](hipStream_t s) { return launch_gemm(lm, s, gw, a, dominant); }, (long)g.bytes);
}

// The depth transformer's attention inside its out_proj (k_dep_attn_out_proj): ONE session at the 16-row tile (the real-time
// configuration), bf16 weights, a wave's K-slice = one or two whole heads.  Same-box at 1 / 2 / 4 sessions: -0.06 / +0.04 / +0.27 ms
// per step (a wave works through its 2 B pairs one after the other; profiles/r04_logs/call_o2_summary.txt), hence one session only.
// MMI_NO_DEP_ATTN_FUSION=1: the two launches (A/B and the bit-equality test)
bool dep_attn_fusable(const mmi_lm* lm, const GemmW& g, int H, int Dh, int steps) {
    if (g.T != 16 || lm->batch != 1 || lm->act8 || g.wq != 0 || getenv("MMI_NO_DEP_ATTN_FUSION")) return false;
    if (Dh % 8 || Dh > 64 || steps > 8 || g.KSTEPS * 32 != H * Dh) return false;
    const GemmPlan p = plan_gemm(g, false);
    const int kper = mmi_cdiv(g.KSTEPS, p.waves);
    return kper <= 4 && (kper * 32) % Dh == 0 && kper * 32 / Dh <= 2;
}

template <int WAVES>
int launch_dep_attn_out_proj(hipStream_t s, int groups, const GemmArgs& a, const DepAttnArgs& d) {
    MMI_LAUNCH((k_dep_attn_out_proj<WAVES, 2, 1>), groups, WAVES * 64, 0, s, a, d);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

// x += out_proj(attention(qkv, frame cache)) (dep_attn_fusable)
void add_dep_attn_out_proj(mmi_lm* lm, const GemmW& g, const DepAttnArgs& da, uint16_t* x, int features) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.out = x; a.epi = MMI_EPI_RESID; a.resid = x; a.B = lm->batch; a.tok_rows = lm->gen_batch;
    a.out_mode = MMI_OUT_PACKED; a.out_ld = features; a.out_ksteps = packed_ksteps_t(lm, g.T, features);
    a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT; a.wscale = g.scale; a.xinv = g.xinv;
    const GemmPlan p = plan_gemm(g, false);
    a.osplit = plan_osplit(g, p, a.epi, g.T);
    const int groups = g.NT * (a.osplit > 1 ? a.osplit : 1), waves = p.waves;
    const long bytes = (long)g.bytes;
    lm->prog.add([=](hipStream_t s) {
        mmi_record_bytes(bytes);
        return waves == 8 ? launch_dep_attn_out_proj<8>(s, groups, a, da) : launch_dep_attn_out_proj<4>(s, groups, a, da);
    }, bytes);
}

// The split-K partials a norm launch still has to fold into the residual stream: P of them; int8 x int8 GEMMs leave int32 sums
// and the norm dequantises (sx = the absmax of the GEMM's input rows, scb = its SCB; k_resid_rmsnorm psx / pscb)
struct Pending { int P = 0; const float* sx = nullptr; const float* scb = nullptr; };

// K-split GEMM whose partial sums (lm->partial) the following add_resid_rmsnorm folds into the residual stream.
// Returns the partials, P = 0 when the GEMM is not split (then it applied the residual itself, in place on x).
Pending add_gemm_resid(mmi_lm* lm, const GemmW& g, const uint16_t* in, uint16_t* x, int features, const Q8* q8 = nullptr) {
    const GemmPlan p = plan_gemm(g, true);
    if (p.ksplit <= 1) {
        add_gemm(lm, g, in, x, features, true, MMI_EPI_RESID, x, nullptr, nullptr, 0, false, nullptr, q8);
        return Pending{};
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    if (q8) { a.wq = q8->wq; a.sx = q8->sx; }
    a.xp = reinterpret_cast<const u32x4*>(in); a.epi = MMI_EPI_PARTIAL; a.partial = lm->partial; a.B = lm->batch;
    GemmW gw = g;
    lm->prog.add([lm, gw, a](hipStream_t s) { return launch_gemm(lm, s, gw, a, false); }, (long)g.bytes);
    Pending pd;
    pd.P = p.ksplit;
    if (q8 && q8->wq == 3 && g.wq == 1) { pd.sx = q8->sx; pd.scb = g.scb; }
    return pd;
}

// x (+= the P pending split-K partials), y = rms_norm(x) * alpha
// yq / sx: also store the row quantised row-wise to int8 (+ its absmax) for the int8 linears that read it (lm->act8)
void add_resid_rmsnorm(mmi_lm* lm, uint16_t* x, Pending pd, const uint16_t* alpha, uint16_t* y, int D, uint8_t* yq = nullptr, float* sx = nullptr,
                       int tile = 0) {
    const int B = lm->batch, T = tile ? tile : lm->T, ksteps = packed_ksteps_t(lm, T, D);
    const float* partial = lm->partial;
    const int P = pd.P;
    const float *psx = pd.sx, *pscb = pd.scb;
    lm->prog.add([=](hipStream_t s) {
        int nth = mmi_cdiv(D / 8, 64) * 64;            // one 16-byte piece per thread where the row allows it
        if (nth > 1024) nth = 1024;
        MMI_LAUNCH(k_resid_rmsnorm, B, nth, 0, s, x, partial, P, B, alpha, y, D, T, ksteps, 1e-8f, yq, sx, psx, pscb);
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
}

// parity tap (mmi_lm_set_hidden_taps; off in production: no op is added): the residual stream, row-major, into htap[which]
void add_hidden_tap(mmi_lm* lm, int which) {
    if (!lm->htap) return;
    const int B = lm->batch, d = lm->cfg.dim, T = lm->T, ksteps = packed_ksteps(lm, d);
    const uint16_t* x = lm->x;
    uint16_t* dst = lm->htap + (size_t)which * B * d;
    lm->prog.add([=](hipStream_t s) {
        MMI_LAUNCH(k_unpack_rows, mmi_cdiv(B * d, 256), 256, 0, s, x, B, d, dst, T, ksteps);
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
}

// k_gemm_q8 at 33..64 sessions: one batch tile per workgroup (grid.y = tiles; the default) or both tiles walked by one workgroup
// (MMI_Q8_TILES=serial: the round-4 form, same-box A/B)
static bool q8_tiles_over_grid() {
    const char* e = getenv("MMI_Q8_TILES");
    return !(e && e[0] == 's');
}

// RMSNorm fused in front of a short-row GEMM (k_gemm_xp_norm; int8 x int8: k_gemm_q8<NORM>): the launch by tile / batch tiles / weight format
int launch_norm_fused(hipStream_t s, int T, int mt, int wq, int NT, const GemmArgs& a) {
    if (wq == 1) {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_xp_norm<32, 1, 8, 4, 1>), NT, 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_xp_norm<32, 2, 8, 4, 1>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_xp_norm<16, 1, 8, 4, 1>), NT, 512, 0, s, a);
    } else if (wq == 3) {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 4, true>), NT, 512, 0, s, a);
        else if (T == 32 && q8_tiles_over_grid()) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 4, true>), dim3(NT, mt), 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_q8<32, 2, 8, 4, true>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_q8<16, 1, 8, 4, true>), NT, 512, 0, s, a);
    } else if (wq == 2) {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_xp_norm<32, 1, 8, 4, 2>), NT, 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_xp_norm<32, 2, 8, 4, 2>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_xp_norm<16, 1, 8, 4, 2>), NT, 512, 0, s, a);
    } else {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_xp_norm<32, 1, 8, 8>), NT, 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_xp_norm<32, 2, 8, 8>), NT, 512, 0, s, a);
        // 16-row tile: rows of <= 32 k-steps (the depth transformer's 1024 features) need 4 fragments per wave, not 8: the
        // smaller register arrays let two workgroups share a CU, so the 352 gated tiles of linear_in are resident at once
        // instead of running as 256 + 96 (same k partition per wave: bit-identical)
        else if (mt == 2 && a.KSTEPS <= 32) MMI_LAUNCH((k_gemm_xp_norm<16, 2, 8, 4>), NT, 512, 0, s, a);   // 17..32 sessions on the depth
        else if (mt == 2) MMI_LAUNCH((k_gemm_xp_norm<16, 2, 8, 8>), NT, 512, 0, s, a);                      // transformer's own tile (lm->Td)
        else if (a.KSTEPS <= 32) MMI_LAUNCH((k_gemm_xp_norm<16, 1, 8, 4>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_xp_norm<16, 1, 8, 8>), NT, 512, 0, s, a);
    }
    MMI_CHECK_LAUNCH();
    return (int)MMI_OK;
}

// k_gemm_q8 without a norm: the row quantisation inside the GEMM (kmax = 4: rows of <= 32 entries, 11: <= 88)
int launch_q8_fused(hipStream_t s, int T, int mt, int kmax, int NT, const GemmArgs& a) {
    if (kmax == 4) {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 4, false>), NT, 512, 0, s, a);
        else if (T == 32 && q8_tiles_over_grid()) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 4, false>), dim3(NT, mt), 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_q8<32, 2, 8, 4, false>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_q8<16, 1, 8, 4, false>), NT, 512, 0, s, a);
    } else {
        if (T == 32 && mt == 1) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 11, false>), NT, 512, 0, s, a);
        else if (T == 32 && q8_tiles_over_grid()) MMI_LAUNCH((k_gemm_q8<32, 1, 8, 11, false>), dim3(NT, mt), 512, 0, s, a);
        else if (T == 32) MMI_LAUNCH((k_gemm_q8<32, 2, 8, 11, false>), NT, 512, 0, s, a);
        else MMI_LAUNCH((k_gemm_q8<16, 1, 8, 11, false>), NT, 512, 0, s, a);
    }
    MMI_CHECK_LAUNCH();
    return (int)MMI_OK;
}
static int q8_fused_kmax(const GemmW& g) { return g.KSTEPS <= 32 ? 4 : (g.KSTEPS <= 88 ? 11 : 0); }
static int q8_fused_osplit(const GemmW& g, int epi, int T) {
    int os = 1;
    if (epi != MMI_EPI_GATE && !getenv("MMI_GEMM_OSPLIT")) {
        const int octs = T / 8;
        while (os < octs && (long)g.NT * os < 128) os *= 2;
        if (g.NT > 64) os = 1;
    }
    return os;
}

// RMSNorm(x) * alpha fused into the GEMM (k_gemm_xp_norm) when a workgroup's 8 waves can hold the whole row slice in
// registers (rows of <= 1024 features at the 32-wide tile: the depth transformer); otherwise norm kernel + GEMM.
// lm->act8: the normalised row is quantised inside the GEMM (k_gemm_q8, fused) or by the norm launch
void add_norm_gemm(mmi_lm* lm, const GemmW& g, uint16_t* x, const uint16_t* alpha, uint16_t* xn_scratch, int D, uint16_t* out,
                   int out_features, bool out_packed, int epi, const DepKv* kv = nullptr) {
    const bool a8 = lm->act8 && g.wq == 1;
    const int wq = a8 ? 3 : g.wq;
    const bool fuse = g.KSTEPS <= (wq ? 32 : 64) && !getenv("MMI_NO_NORM_FUSION");
    if (!fuse) {
        if (a8) {
            add_resid_rmsnorm(lm, x, Pending{}, alpha, xn_scratch, D, lm->dxnq, lm->sx_dxn, g.T);
            Q8 q{3, lm->sx_dxn};
            add_gemm(lm, g, reinterpret_cast<const uint16_t*>(lm->dxnq), out, out_features, out_packed, epi, nullptr, nullptr, nullptr, 0, false, kv, &q);
            return;
        }
        add_resid_rmsnorm(lm, x, Pending{}, alpha, xn_scratch, D, nullptr, nullptr, g.T);
        add_gemm(lm, g, xn_scratch, out, out_features, out_packed, epi, nullptr, nullptr, nullptr, 0, false, kv);
        return;
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    if (kv) { a.kc = kv->kc; a.vc = kv->vc; a.H = kv->H; a.Dh = kv->Dh; a.cap = kv->steps; }
    a.xp = reinterpret_cast<const u32x4*>(x); a.out = out; a.epi = epi; a.B = lm->batch;
    a.out_mode = out_packed ? MMI_OUT_PACKED : MMI_OUT_ROWMAJOR;
    a.out_ld = out_features;
    a.out_ksteps = packed_ksteps_t(lm, g.T, out_features);
    a.alpha = alpha; a.D = D; a.eps = 1e-8f;
    a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT;
    a.wscale = g.scale; a.wscb = g.scb; a.gate_rows = g.gate ? g.N : 0;
    a.wq = wq; a.xinv = g.xinv;
    a.osplit = 1;
    const int T = g.T, mt = mmi_cdiv(lm->batch, g.T), NT = g.NT * (a.osplit > 1 ? a.osplit : 1);
    const long gbytes = (long)g.bytes;
    lm->prog.add([=](hipStream_t s) {
        mmi_record_bytes(gbytes);
        return launch_norm_fused(s, T, mt, wq, NT, a);
    }, gbytes);
}

// bf16 packed tensor -> its row-wise int8 copy + the rows' absmax (lm->act8): one workgroup per row
void add_quant_rows(mmi_lm* lm, const uint16_t* xp, uint8_t* xq, float* sx, int features, int tile = 0) {
    const int B = lm->batch, T = tile ? tile : lm->T, ksteps = packed_ksteps_t(lm, T, features);
    lm->prog.add([=](hipStream_t s) {
        int nth = mmi_cdiv(features / 8, 64) * 64;
        if (nth > 1024) nth = 1024;
        MMI_LAUNCH(k_quant_rows_i8, B, nth, 0, s, xp, B, features, T, ksteps, xq, sx);
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
}

// An int8 linear of the depth transformer on a bf16 input WITHOUT a norm in front (out_proj, linear_out, the logits heads) under
// lm->act8: the row quantisation fused into the GEMM (k_gemm_q8, rows of <= 8 * 11 entries); longer rows: quantisation launch
// (into the gated tensor's scratch) + the plain int8 x int8 GEMM.
void add_q8_gemm(mmi_lm* lm, const GemmW& g, const uint16_t* x, int in_features, uint16_t* out, int out_features, bool out_packed, int epi,
                 const uint16_t* resid) {
    const int T = g.T, mt = mmi_cdiv(lm->batch, g.T);
    const int kmax = q8_fused_kmax(g);
    if (!kmax || getenv("MMI_NO_NORM_FUSION")) {
        add_quant_rows(lm, x, lm->hbq, lm->sx_hb, in_features, g.T);
        Q8 q{3, lm->sx_hb};
        add_gemm(lm, g, reinterpret_cast<const uint16_t*>(lm->hbq), out, out_features, out_packed, epi, resid, nullptr, nullptr, 0, false, nullptr, &q);
        return;
    }
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = reinterpret_cast<const u32x4*>(x); a.out = out; a.epi = epi; a.resid = resid; a.B = lm->batch; a.tok_rows = lm->gen_batch;
    a.out_mode = out_packed ? MMI_OUT_PACKED : MMI_OUT_ROWMAJOR;
    a.out_ld = out_features;
    a.out_ksteps = packed_ksteps_t(lm, g.T, out_features);
    a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT;
    a.wscale = g.scale; a.wscb = g.scb; a.gate_rows = g.gate ? g.N : 0;
    a.wq = 3; a.xinv = g.xinv;
    // few n-tiles (the N = 1024 linears: 32 tiles for 256 CUs): share a tile's row octets out over several workgroups, as
    // plan_osplit does for the bf16 form of these GEMMs
    a.osplit = q8_fused_osplit(g, epi, T);
    const int NT = g.NT * a.osplit;
    const long gbytes = (long)g.bytes;
    lm->prog.add([=](hipStream_t s) {
        mmi_record_bytes(gbytes);
        return launch_q8_fused(s, T, mt, kmax, NT, a);
    }, gbytes);
}

// next_k >= 0: the sampled token opens depth-transformer micro-step next_k, whose input row the sampler writes itself
void add_sample(mmi_lm* lm, uint16_t* logits, int ld, int V, bool text, int site, int* out, int out_stride, int next_k = -1) {
    const bool guided = lm->cfg_coef != 1.f;
    if (guided && !(text && lm->cfg_no_text)) {     // lm.py:727-733 (text; `cfg_is_no_text` keeps the conditioned logits), 828-832
        const int G = lm->gen_batch;
        const float coef = lm->cfg_coef;
        lm->prog.add([=](hipStream_t s) {
            MMI_LAUNCH(k_cfg_mix, dim3(mmi_cdiv(V, 256), G), 256, 0, s, logits, ld, V, G, coef);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    SampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    sa.nx_dup = guided ? lm->gen_batch : 0;
    if (next_k >= 0) {
        const int dd = lm->cfg.depformer_dim;
        sa.nx_pre = lm->dpre + (size_t)next_k * dd; sa.nx_ld = lm->cfg.dep_q * dd;
        sa.nx_emb = lm->dep_emb[next_k]; sa.nx_out = lm->dx;
        sa.nx_D = dd; sa.nx_T = lm->Td; sa.nx_ksteps = packed_ksteps_t(lm, lm->Td, dd);
    }
    sa.logits = logits; sa.ld = ld; sa.V = V;
    sa.k = text ? lm->samp.top_k_text : lm->samp.top_k;
    sa.temp = text ? lm->samp.temp_text : lm->samp.temp;
    sa.use_sampling = lm->samp.use_sampling;
    if (sa.k >= V) sa.k = 0;   // top_k == 0 (or the whole vocabulary): plain multinomial, k_sample's a.k <= 0 branch
    sa.noise = lm->noise + (size_t)site * lm->kmax;
    sa.noise_ld = (1 + lm->cfg.dep_q) * lm->kmax;
    sa.use_noise = lm->use_noise; sa.rng = lm->rng; sa.site = site; sa.out = out; sa.out_stride = out_stride;
    sa.B = lm->gen_batch;
    sa.forced = lm->forced + site; sa.forced_stride = 1 + lm->cfg.dep_q; sa.use_forced = lm->use_forced;
    const int B = lm->gen_batch;
    if (text) {                     // hook boundaries: the guided logits are final here, the sampler is the next op
        lm->op_text_sample = lm->prog.ops.size();
        lm->op_after_text_sample = lm->op_text_sample + 1;
        lm->text_sample_args = sa;
    }
    lm->prog.add([=](hipStream_t s) {
        if (V <= 2048) MMI_LAUNCH((k_sample<256, 8, true>), B, 256, 0, s, sa);
        else if (V <= 8192) MMI_LAUNCH((k_sample<1024, 8, true>), B, 1024, 0, s, sa);
        else MMI_LAUNCH((k_sample<1024, 32, true>), B, 1024, 0, s, sa);   // 32 logits per thread = 16 VGPRs: read once, not once per pass
        MMI_CHECK_LAUNCH();
        return (int)MMI_OK;
    });
}

// MMI_ATTN: "wave" (default since round 4) = k_lm_attn_wave, one online softmax per wave, no barrier in the loop; "split" = the
// chunked kernel of rounds 1-3 (same-box A/Bs)
static bool attn_wave_kernel() {
    const char* e = getenv("MMI_ATTN");
    return !(e && e[0] == 's');
}

// workgroups per (session, head) of the decode attention.  k_lm_attn_wave: ONE from 128 pairs (4 sessions) on - one workgroup per
// pair walking the whole ring beats ring split + merge launch at every depth there (4 / 8 / 16 sessions, 3000 rows: -0.03 / -0.16 /
// -0.16 ms per step, profiles/r04_logs/call_v_summary.txt); below that the ring is split (and still walked by one workgroup while
// it is short: attn_solo_rows).  The chunked kernel of rounds 1-3: 1 once B*H alone fills the chip.
int attn_splits(const mmi_lm_cfg& c, int B) {
    if (const char* e = getenv("MMI_ATTN_NS")) {          // test hook: the split + combine path on rings too short to need it
        const int v = atoi(e);
        if (v >= 1 && v <= 16) return v;
    }
    if (attn_wave_kernel() && B * c.num_heads >= 128) return 1;
    const int chunks = mmi_cdiv(c.context, MMI_ATTN_CHUNK);
    int want = 1024 / (B * c.num_heads);
    if (want < 1) want = 1;
    return want < chunks ? want : chunks;
}

// k_lm_attn_wave, ring split over several workgroups: up to this many rows, workgroup 0 walks the ring alone and writes the output
// itself.  Measured crossover against split + merge launch (profiles/r04_logs/call_u_summary.txt, call_v_summary.txt): one session
// between 600 and 900 rows, two sessions between 600 and 1800.  MMI_ATTN_SOLO: test hook (0 = always the merge launch)
static int attn_solo_rows(int pairs) {
    if (const char* e = getenv("MMI_ATTN_SOLO")) return atoi(e);
    return pairs <= 32 ? 768 : 1200;
}

// The ring split over NS > 1 workgroups (fewer than 32 sessions) has two step programs (MmiProgram::variant):
//   0  while no ring can hold more than solo_rows rows (the host's bound on the offsets): k_lm_attn_wave alone - workgroup 0 of
//      each (session, head) walks the ring and writes the output.  One launch per layer instead of two: 4.4 us x 32 layers of a
//      4.7 ms single-session step.  (Should a ring be longer all the same, the kernel merges in its last-arriving workgroup:
//      correct whatever the host believes, 8 us per layer slower than a launch - measured, profiles/r04_logs/call_j_summary.txt)
//   1  deeper rings: partials from every workgroup + the k_lm_attn_combine launch, as in rounds 1-3.
// MMI_ATTN_MERGE=kernel (test hook) stays on variant 0 whatever the depth.
static int attn_variant(const mmi_lm* lm, int NS) {
    if (NS <= 1 || !attn_wave_kernel()) return 0;
    if (const char* e = getenv("MMI_ATTN_MERGE")) if (e[0] == 'k') return 0;
    return lm->depth_bound + 1 > (long)attn_solo_rows(lm->batch * lm->cfg.num_heads) ? 1 : 0;
}

int launch_attn_split(hipStream_t s, const LmAttnArgs& a, bool kv8) {
    dim3 grid(a.B * a.H, a.NS);
    if (attn_wave_kernel()) {
        if (kv8) {
            switch (a.Dh) {
                case 128: MMI_LAUNCH((k_lm_attn_wave<128, true>), grid, 256, 0, s, a); break;
                case 64: MMI_LAUNCH((k_lm_attn_wave<64, true>), grid, 256, 0, s, a); break;
                case 32: MMI_LAUNCH((k_lm_attn_wave<32, true>), grid, 256, 0, s, a); break;
                default: return mmi_fail(MMI_ERR_UNSUPPORTED, "head dim must be 32, 64 or 128");
            }
        } else {
            switch (a.Dh) {
                case 128: MMI_LAUNCH((k_lm_attn_wave<128>), grid, 256, 0, s, a); break;
                case 64: MMI_LAUNCH((k_lm_attn_wave<64>), grid, 256, 0, s, a); break;
                case 32: MMI_LAUNCH((k_lm_attn_wave<32>), grid, 256, 0, s, a); break;
                default: return mmi_fail(MMI_ERR_UNSUPPORTED, "head dim must be 32, 64 or 128");
            }
        }
        MMI_CHECK_LAUNCH();
        return MMI_OK;
    }
    if (kv8) {
        switch (a.Dh) {
            case 128: MMI_LAUNCH((k_lm_attn_split<128, true>), grid, 256, 0, s, a); break;
            case 64: MMI_LAUNCH((k_lm_attn_split<64, true>), grid, 256, 0, s, a); break;
            case 32: MMI_LAUNCH((k_lm_attn_split<32, true>), grid, 256, 0, s, a); break;
            default: return mmi_fail(MMI_ERR_UNSUPPORTED, "head dim must be 32, 64 or 128");
        }
        MMI_CHECK_LAUNCH();
        return MMI_OK;
    }
    switch (a.Dh) {
        case 128: MMI_LAUNCH((k_lm_attn_split<128>), grid, 256, 0, s, a); break;
        case 64: MMI_LAUNCH((k_lm_attn_split<64>), grid, 256, 0, s, a); break;
        case 32: MMI_LAUNCH((k_lm_attn_split<32>), grid, 256, 0, s, a); break;
        default: return mmi_fail(MMI_ERR_UNSUPPORTED, "head dim must be 32, 64 or 128");
    }
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

TokArgs tok_args(mmi_lm* lm) {
    TokArgs t;
    t.cache = lm->cache; t.offsets = lm->offsets; t.exec = lm->exec; t.delays = lm->delays_dev;
    t.B = lm->gen_batch; t.NC = lm->NC; t.CT = lm->CT; t.dep_q = lm->cfg.dep_q; t.max_delay = lm->max_delay;
    t.card = lm->cfg.card; t.text_card = lm->cfg.text_card;
    return t;
}

int build_program(mmi_lm* lm) {
    const mmi_lm_cfg& c = lm->cfg;
    const int B = lm->batch, d = c.dim, H = c.num_heads, Dh = d / H;
    const int dd = c.depformer_dim, Hd = c.depformer_num_heads, Dhd = dd / Hd;
    const int n_user = c.n_q - c.dep_q;
    MmiProgram& P = lm->prog;
    const bool a8 = lm->act8;
    // ---- token ring in, embeddings
    P.site("prepare");
    {
        TokArgs t = tok_args(lm);
        const int* user = lm->user_i32; int* tokens = lm->tokens;
        const uint16_t *emb = lm->emb, *temb = lm->text_emb; uint16_t* x = lm->x; const int NC = lm->NC, card1 = c.card + 1;
        const int T = lm->T, xks = packed_ksteps(lm, d);
        float* rope = lm->rope; const float max_period = c.max_period;
        const int G = lm->gen_batch;
        const bool guided = lm->cfg_coef != 1.f;
        const int* masked_until = lm->masked_until; const int no_text = lm->cfg_no_text; long* offsets_m = lm->offsets_m;
        const uint16_t* cond = lm->cond;
        P.add([=](hipStream_t s) {
            MMI_LAUNCH(k_lm_prepare, mmi_cdiv(G * NC + G * (Dh / 2), 128), 128, 0, s, t, user, n_user, tokens, rope, Dh, max_period);
            if (guided)
                MMI_LAUNCH(k_lm_cfg_twins, mmi_cdiv(G * NC + G * Dh + G, 128), 128, 0, s, t, tokens, masked_until, no_text, offsets_m, rope, Dh);
            MMI_LAUNCH(k_lm_embed, dim3(mmi_cdiv(d, 256), B), 256, 0, s, (const int*)tokens, NC, emb, card1, temb, x, d, T, xks, cond);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    // ---- temporal transformer
    const int NS = attn_splits(c, B);
    lm->attn_ns = NS;
    const bool kv8 = c.kv_cache_dtype == MMI_F8E4M3;
    const size_t kv_layer = (size_t)B * H * c.context * Dh / (kv8 ? 2 : 1);     // in uint16 units: an fp8 ring is half as large
    Pending pending;   // split-K partials of the previous linear_out still to be folded into x
    for (int l = 0; l < c.num_layers; ++l) {
        const LayerW& L = lm->layers[l];
        P.site("L.norm1");
        add_resid_rmsnorm(lm, lm->x, pending, L.n1, lm->xn, d, a8 ? lm->xnq : nullptr, a8 ? lm->sx_xn : nullptr);
        if (l == 1) add_hidden_tap(lm, 0);     // x is complete (the previous linear_out's partials folded in) right after this norm
        LmAttnArgs a;
        a.qrot = lm->qrot; a.kc = lm->kc + l * kv_layer; a.vc = lm->vc + l * kv_layer;
        a.offsets = lm->offsets_m; a.opart = lm->opart; a.ml = lm->ml; a.out = lm->att;
        a.B = B; a.H = H; a.Dh = Dh; a.cap = c.context; a.context = c.context; a.NS = NS; a.max_period = c.max_period;
        a.T = lm->T; a.out_ksteps = packed_ksteps(lm, d);
        a.done = lm->attn_done; a.solo_rows = attn_solo_rows(B * H);
        P.site("L.in_proj");
        {   // in_proj with RoPE + ring-KV write in its epilogue
            GemmArgs ga;
            memset(&ga, 0, sizeof(ga));
            ga.xp = reinterpret_cast<const u32x4*>(lm->xn); ga.epi = MMI_EPI_ROPE_KV; ga.B = B;
            if (a8) { ga.xp = reinterpret_cast<const u32x4*>(lm->xnq); ga.sx = lm->sx_xn; ga.wq = 3; }
            ga.qrot = a.qrot; ga.kc = a.kc; ga.vc = a.vc; ga.offsets = lm->offsets_m; ga.H = H; ga.Dh = Dh; ga.cap = c.context; ga.kv8 = kv8 ? 1 : 0;
            ga.max_period = c.max_period; ga.rope = lm->rope;
            GemmW gw = L.in_proj;
            P.add([lm, gw, ga](hipStream_t s) { return launch_gemm(lm, s, gw, ga, false); }, (long)gw.bytes);
        }
        P.site("L.attn");
        P.add([=](hipStream_t s) {
            LmAttnArgs aa = a;
            const bool merge_launch = a.NS > 1 && (!attn_wave_kernel() || lm->prog.variant == 1);
            if (merge_launch) { aa.solo_rows = -1; aa.done = nullptr; }     // every workgroup leaves its partial (m, l, O)
            int rc = launch_attn_split(s, aa, kv8);
            if (rc) return rc;
            if (merge_launch) MMI_LAUNCH(k_lm_attn_combine, B * H, Dh < 64 ? 64 : Dh, 0, s, aa);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
        P.site("L.out_proj");
        if (a8) {
            add_quant_rows(lm, lm->att, lm->attq, lm->sx_att, d);
            Q8 q{3, lm->sx_att};
            pending = add_gemm_resid(lm, L.out_proj, reinterpret_cast<const uint16_t*>(lm->attq), lm->x, d, &q);
        } else
        pending = add_gemm_resid(lm, L.out_proj, lm->att, lm->x, d);
        if (c.cross_attention) {   // x = x + cross_attention(norm_cross(x), src, src) (transformer.py:779-786)
            P.site("L.norm_cross");
            {
                const int T = lm->T, ksteps = packed_ksteps(lm, d), Pn = pending.P;
                uint16_t *x = lm->x, *y = lm->xn; const float* partial = lm->partial;
                const uint16_t *w = L.nx_w, *bb = L.nx_b;
                P.add([=](hipStream_t s) {
                    int nth = mmi_cdiv(d / 8, 64) * 64;
                    if (nth > 1024) nth = 1024;
                    MMI_LAUNCH(k_resid_layernorm, B, nth, 0, s, x, partial, Pn, B, w, bb, y, d, T, ksteps, 1e-5f);
                    MMI_CHECK_LAUNCH();
                    return (int)MMI_OK;
                });
            }
            P.site("L.cross_q");
            add_gemm(lm, L.x_q, lm->xn, lm->qrot, d, false, MMI_EPI_STORE, nullptr);
            P.site("L.cross_attn");
            {
                CrossAttnArgs ca;
                ca.q = lm->qrot; ca.kv = lm->xkv + (size_t)l * B * lm->cross_len * 2 * d; ca.out = lm->att;
                ca.B = B; ca.H = H; ca.Dh = Dh; ca.Tc = lm->cross_len; ca.T = lm->T; ca.out_ksteps = packed_ksteps(lm, d);
                P.add([=](hipStream_t s) {
                    MMI_LAUNCH(k_lm_cross_attn, B * H, 64, 0, s, ca);
                    MMI_CHECK_LAUNCH();
                    return (int)MMI_OK;
                });
            }
            P.site("L.cross_out");
            pending = add_gemm_resid(lm, L.x_out, lm->att, lm->x, d);
        }
        P.site("L.norm2");
        add_resid_rmsnorm(lm, lm->x, pending, L.n2, lm->xn, d, a8 ? lm->xnq : nullptr, a8 ? lm->sx_xn : nullptr);
        P.site("L.ffn_in");
        if (a8) {
            Q8 qi{3, lm->sx_xn};
            add_gemm(lm, L.ffn_in, reinterpret_cast<const uint16_t*>(lm->xnq), lm->hb, c.ffn_hidden, true, MMI_EPI_GATE, nullptr, nullptr, nullptr, 0,
                     /*dominant=*/true, nullptr, &qi);
            P.site("L.ffn_out");
            add_quant_rows(lm, lm->hb, lm->hbq, lm->sx_hb, c.ffn_hidden);
            Q8 qo{3, lm->sx_hb};
            pending = add_gemm_resid(lm, L.ffn_out, reinterpret_cast<const uint16_t*>(lm->hbq), lm->x, d, &qo);
            continue;
        }
        add_gemm(lm, L.ffn_in, lm->xn, lm->hb, c.ffn_hidden, true, MMI_EPI_GATE, nullptr, nullptr, nullptr, 0, /*dominant=*/true);
        P.site("L.ffn_out");
        pending = add_gemm_resid(lm, L.ffn_out, lm->hb, lm->x, d);
    }
    P.site("out_norm");
    add_resid_rmsnorm(lm, lm->x, pending, lm->out_norm, lm->tout, d, a8 ? lm->toutq : nullptr, a8 ? lm->sx_tout : nullptr);
    if (c.num_layers == 1) add_hidden_tap(lm, 0);
    add_hidden_tap(lm, 1);
    // the int8 linears that read transformer_out (text head, depformer_in) take its int8 copy
    const uint16_t* tout_in = a8 ? reinterpret_cast<const uint16_t*>(lm->toutq) : lm->tout;
    const Q8 q_tout{3, lm->sx_tout};
    const Q8* qt = a8 ? &q_tout : nullptr;
    P.site("text_linear");
    add_gemm(lm, lm->text_linear, tout_in, lm->text_logits, c.text_card_out, false, MMI_EPI_STORE, nullptr, nullptr, nullptr, 0, false, nullptr, qt);
    // depformer_in[k](transformer_out) for every micro-step in one launch; each sampler then adds its token's embedding row
    // and writes the next micro-step's input (lm.py:465-470) - 8 dependent launches less on the sequential chain
    const bool grouped = lm->dep_in_grouped;
    lm->op_depformer = P.ops.size();
    P.site("dep.in_all");
    if (grouped) add_gemm(lm, lm->dep_in_all, tout_in, lm->dpre, c.dep_q * dd, false, MMI_EPI_STORE, nullptr, nullptr, nullptr, 0, false, nullptr, qt);
    P.site("text_sample");
    add_sample(lm, lm->text_logits, c.text_card_out, c.text_card_out, true, 0, lm->text_tok, 1, grouped ? 0 : -1);
    // ---- depformer: dep_q sequential micro-steps
    const size_t dkv_layer = (size_t)B * Hd * c.dep_q * Dhd;
    for (int k = 0; k < c.dep_q; ++k) {
        const int* prev = k == 0 ? lm->text_tok : lm->audio_tok + (k - 1);
        const int prev_stride = k == 0 ? 1 : c.dep_q;
        P.site("dep.in");
        if (!grouped) add_gemm(lm, lm->dep_in[k], tout_in, lm->dx, dd, true, MMI_EPI_EMB, nullptr, lm->dep_emb[k], prev, prev_stride, false, nullptr, qt);
        for (int l = 0; l < c.depformer_num_layers; ++l) {
            const DepLayerW& L = lm->dep_layers[l];
            P.site("dep.in_proj");
            // micro-step 0 attends to one position: softmax over one score is 1 and the attention output is v itself, so in_proj's
            // epilogue writes k / v into the frame's cache and v as out_proj's operand, and the attention launch is dropped
            // (bit-identical: 1 * v / 1; MMI_DEP_ATTN0_LAUNCH=1 keeps the launch)
            const bool skip_attn0 = k == 0 && Dhd % 8 == 0 && !getenv("MMI_DEP_ATTN0_LAUNCH");
            if (skip_attn0) {
                DepKv kv{lm->dkc + l * dkv_layer, lm->dvc + l * dkv_layer, Hd, Dhd, c.dep_q};
                add_norm_gemm(lm, L.in_proj[k], lm->dx, L.n1, lm->dxn, dd, lm->datt, dd, true, MMI_EPI_DEP_QKV0, &kv);
            } else
            add_norm_gemm(lm, L.in_proj[k], lm->dx, L.n1, lm->dxn, dd, lm->dqkv, 3 * dd, false, MMI_EPI_STORE);
            DepAttnArgs da;
            da.qkv = lm->dqkv; da.kc = lm->dkc + l * dkv_layer; da.vc = lm->dvc + l * dkv_layer; da.out = lm->datt;
            da.B = B; da.H = Hd; da.Dh = Dhd; da.steps = c.dep_q; da.k = k;
            da.T = lm->Td; da.out_ksteps = packed_ksteps_t(lm, lm->Td, dd);
            P.site("dep.attn");
            const bool attn8 = Dhd % 8 == 0 && c.dep_q <= 8;        // else the general one-wave-per-(session, head) kernel
            const bool attn_in_gemm = !skip_attn0 && attn8 && dep_attn_fusable(lm, L.out_proj[k], Hd, Dhd, c.dep_q);
            if (!skip_attn0 && !attn_in_gemm)
            P.add([=](hipStream_t s) {
                if (attn8) MMI_LAUNCH((k_dep_attn8<4>), mmi_cdiv(B * Hd, 4), 256, 0, s, da);
                else MMI_LAUNCH(k_dep_attn, B * Hd, 64, 0, s, da);
                MMI_CHECK_LAUNCH();
                return (int)MMI_OK;
            });
            // int8 activations: the chain's linears quantise their own input row inside the GEMM (k_gemm_q8)
            P.site("dep.out_proj");
            if (attn_in_gemm) add_dep_attn_out_proj(lm, L.out_proj[k], da, lm->dx, dd);
            else if (a8) add_q8_gemm(lm, L.out_proj[k], lm->datt, dd, lm->dx, dd, true, MMI_EPI_RESID, lm->dx);
            else add_gemm(lm, L.out_proj[k], lm->datt, lm->dx, dd, true, MMI_EPI_RESID, lm->dx);
            P.site("dep.ffn_in");
            add_norm_gemm(lm, L.ffn_in[k], lm->dx, L.n2, lm->dxn, dd, lm->dhb, c.depformer_ffn_hidden, true, MMI_EPI_GATE);
            P.site("dep.ffn_out");
            if (a8) add_q8_gemm(lm, L.ffn_out[k], lm->dhb, c.depformer_ffn_hidden, lm->dx, dd, true, MMI_EPI_RESID, lm->dx);
            else add_gemm(lm, L.ffn_out[k], lm->dhb, lm->dx, dd, true, MMI_EPI_RESID, lm->dx);
        }
        uint16_t* lg = lm->dlogits + (size_t)k * B * c.card;
        P.site("dep.lin");
        if (a8) add_q8_gemm(lm, lm->dep_lin[k], lm->dx, dd, lg, c.card, false, MMI_EPI_STORE, nullptr);
        else add_gemm(lm, lm->dep_lin[k], lm->dx, lg, c.card, false, MMI_EPI_STORE, nullptr);
        P.site("dep.sample");
        add_sample(lm, lg, c.card, c.card, false, 1 + k, lm->audio_tok + k, c.dep_q, grouped && k + 1 < c.dep_q ? k + 1 : -1);
    }
    // ---- token ring out
    P.site("commit");
    lm->op_commit = P.ops.size();
    {
        TokArgs t = tok_args(lm);
        const int *tt = lm->text_tok, *at = lm->audio_tok; int* out = lm->out_i32; unsigned long long* rng = lm->rng;
        P.add([=](hipStream_t s) {
            MMI_LAUNCH(k_lm_commit, mmi_cdiv(t.B, 64), 64, 0, s, t, tt, at, out, rng);
            MMI_CHECK_LAUNCH();
            return (int)MMI_OK;
        });
    }
    return MMI_OK;
}

// keys | values of the cross-attention source for every temporal layer (transformer.py:495-505: `linear(src, in_proj.weight[dim:])`,
// bf16), once per stream: the source's rows*T_c positions go through the weight-streaming GEMM as batches of T "sessions"
int project_cross_source(mmi_lm* lm, const uint16_t* src, hipStream_t s) {
    const mmi_lm_cfg& c = lm->cfg;
    const int d = c.dim, T = lm->T, ncol = lm->batch * lm->cross_len, ksteps = packed_ksteps(lm, d);
    uint16_t* xp = lm->xn;                                        // packed scratch of >= one batch tile
    for (int col0 = 0; col0 < ncol; col0 += T) {
        const int n = ncol - col0 < T ? ncol - col0 : T;
        MMI_HIP_CHECK(hipMemsetAsync(xp, 0, (size_t)ksteps * 512 * sizeof(uint16_t), s));
        MMI_LAUNCH(k_pack_rows, mmi_cdiv(n * d, 256), 256, 0, s, src + (size_t)col0 * d, n, d, xp, T, ksteps);
        MMI_CHECK_LAUNCH();
        for (int l = 0; l < c.num_layers; ++l) {
            GemmArgs a;
            memset(&a, 0, sizeof(a));
            a.xp = reinterpret_cast<const u32x4*>(xp);
            a.out = lm->xkv + ((size_t)l * ncol + col0) * 2 * d;
            a.epi = MMI_EPI_STORE; a.B = n; a.out_mode = MMI_OUT_ROWMAJOR; a.out_ld = 2 * d; a.out_ksteps = ksteps;
            a.tok_rows = n;
            int rc = launch_gemm(lm, s, lm->layers[l].x_kv, a, false);
            if (rc) return rc;
        }
    }
    // the packed scratch goes back to all-zero padding (the step's producers only write the live rows)
    MMI_HIP_CHECK(hipMemsetAsync(xp, 0, packed_elems(lm, d) * sizeof(uint16_t), s));
    return MMI_OK;
}

int check_cfg(const mmi_lm_cfg& c) {
    if (c.dim % c.num_heads || c.depformer_dim % c.depformer_num_heads) return mmi_fail(MMI_ERR_UNSUPPORTED, "dim % heads != 0");
    const int Dh = c.dim / c.num_heads, Dhd = c.depformer_dim / c.depformer_num_heads;
    if (Dh != 32 && Dh != 64 && Dh != 128) return mmi_fail(MMI_ERR_UNSUPPORTED, "temporal head dim must be 32/64/128");
    if (Dhd > 64 || Dhd < 1) return mmi_fail(MMI_ERR_UNSUPPORTED, "depformer head dim must be <= 64");
    if (c.kv_cache_dtype != 0 && c.kv_cache_dtype != MMI_BF16 && c.kv_cache_dtype != MMI_F8E4M3)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "kv_cache_dtype must be MMI_BF16 or MMI_F8E4M3");
    if (c.kv_cache_dtype == MMI_F8E4M3 && (c.dim / c.num_heads) % 16) return mmi_fail(MMI_ERR_UNSUPPORTED, "fp8 KV needs a head dim multiple of 16");
    if (c.card % 8 || c.text_card_out % 8 || c.card > 32768 || c.text_card_out > 32768)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "vocabulary sizes must be multiples of 8 and at most 32768");
    if (c.dep_q < 0 || c.dep_q > 16 || c.n_q < c.dep_q || c.n_q + 1 > 64) return mmi_fail(MMI_ERR_UNSUPPORTED, "bad n_q / dep_q");
    if (c.dim > 8 * 1024 * MMI_NORM_MAXP || c.depformer_dim > 8 * 1024 * MMI_NORM_MAXP)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "model width above the RMSNorm kernel's register budget");
    if (c.dim % 8 || c.depformer_dim % 8 || c.ffn_hidden % 8 || c.depformer_ffn_hidden % 8)
        return mmi_fail(MMI_ERR_UNSUPPORTED, "feature sizes must be multiples of 8");
    return MMI_OK;
}

// LMGen's per-step hooks (lm.py:734-757): the launch list cut at the text sampler and at the ring commit, run eagerly, with
// the host callbacks in between.  Nothing synchronises: the hooks enqueue their copies / kernels on the same stream.
int run_step_with_hooks(mmi_lm* lm, hipStream_t s) {
    MmiProgram& P = lm->prog;
    auto call = [&](int (*fn)(void*)) -> int {
        if (!fn) return MMI_OK;
        lm->in_hook = true;
        const int r = fn(lm->hooks.user);
        lm->in_hook = false;
        return r ? mmi_fail(MMI_ERR_INVALID, "a step hook reported an error") : MMI_OK;
    };
    int rc;
    if (lm->phase_fn && lm->op_depformer <= lm->op_text_sample) {       // mmi_lm_set_phase_callback: the same point as in the un-hooked step
        if ((rc = P.run_range(s, 0, lm->op_depformer))) return rc;
        if (lm->phase_fn(lm->phase_user, (mmi_stream)s)) return mmi_fail(MMI_ERR_INVALID, "the phase callback reported an error");
        rc = P.run_range(s, lm->op_depformer, lm->op_text_sample);
    } else rc = P.run_range(s, 0, lm->op_text_sample);
    if (rc || (rc = call(lm->hooks.on_text_logits))) return rc;
    if ((rc = P.run_range(s, lm->op_text_sample, lm->op_after_text_sample))) return rc;
    if (lm->hooks.on_text_token) {
        if ((rc = call(lm->hooks.on_text_token))) return rc;
        // the sampler also wrote the depth transformer's first input row from ITS token (fused, lm.py:465-470): redo that
        // row from the token the hook left behind
        if (lm->text_sample_args.nx_out) {
            MMI_LAUNCH(k_dep_next_input, lm->gen_batch, 128, 0, s, lm->text_sample_args, (const int*)lm->text_tok);
            MMI_CHECK_LAUNCH();
        }
    }
    if ((rc = P.run_range(s, lm->op_after_text_sample, lm->op_commit))) return rc;
    if (lm->cfg.dep_q > 0 && (rc = call(lm->hooks.on_audio_tokens))) return rc;      // no depformer, no audio tokens (lm.py:748-749)
    return P.run_range(s, lm->op_commit, P.ops.size());
}

}  // namespace

// ===============================================================================================
// C ABI
// ===============================================================================================
extern "C" int mmi_lm_create(const mmi_lm_cfg* cfg, const mmi_tensor_desc* weights, int32_t n_weights, int32_t max_batch,
                             mmi_lm** out) {
    if (!cfg || !weights || !out || max_batch <= 0) return mmi_fail(MMI_ERR_INVALID, "mmi_lm_create: bad argument");
    if (max_batch > 64) return mmi_fail(MMI_ERR_UNSUPPORTED, "max_batch > 64 sessions per GPU is not supported yet");
    mmi_lm_cfg norm_cfg = *cfg;
    if (norm_cfg.dep_q == 0) {   // "No-Depformer --- e.g., an ASR model" (lm.py:218-221): text stream only; the depth-transformer
        norm_cfg.depformer_num_layers = 0;   // fields are then unused, give them inert values so that the sizing code stays generic
        norm_cfg.depformer_dim = 8; norm_cfg.depformer_num_heads = 1; norm_cfg.depformer_ffn_hidden = 8;
    }
    int rc = check_cfg(norm_cfg);
    if (rc) return rc;
    mmi_lm* lm = new mmi_lm();
    if (hipGetDevice(&lm->device) != hipSuccess) lm->device = -1;
    lm->cfg = norm_cfg;
    lm->max_batch = max_batch;
    lm->T = max_batch <= 16 ? 16 : 32;
    if (const char* e = getenv("MMI_LM_TILE")) {      // A/B hook: the 32-row tile (and with it k_gemm_xlds / the octet sharing) at small batches
        const int v = atoi(e);
        if (v == 32 || (v == 16 && max_batch <= 16)) lm->T = v;
    }
    lm->use_graph = mmi_graphs_enabled();
    const mmi_lm_cfg& c = lm->cfg;
    lm->NC = c.n_q + 1;
    lm->max_delay = 0;
    for (int i = 0; i < lm->NC; ++i) lm->max_delay = c.delays[i] > lm->max_delay ? c.delays[i] : lm->max_delay;
    lm->CT = lm->max_delay + 2;
    MmiWeights W{weights, n_weights};
    auto fail = [&](int code) { mmi_lm_destroy(lm); return code; };
    const int d = c.dim, dd = c.depformer_dim;
    // embeddings (lm.py:135-139, 189-196): row gathers, kept as stored
    {
        const size_t per = (size_t)(c.card + 1) * d;
        if (hipSuccess != lm->wts.alloc(&lm->emb, per * c.n_q)) return fail(mmi_fail(MMI_ERR_HIP, "out of device memory"));
        for (int i = 0; i < c.n_q; ++i)
            if ((rc = load_copy(lm, W, "emb." + std::to_string(i) + ".weight", 2, per, nullptr, lm->emb + per * i))) return fail(rc);
        if ((rc = load_copy(lm, W, "text_emb.weight", 2, (size_t)(c.text_card + 1) * d, &lm->text_emb))) return fail(rc);
        lm->dep_emb.resize(c.dep_q);
        if (c.dep_q > 0 && (rc = load_copy(lm, W, "depformer_text_emb.weight", 2, (size_t)(c.text_card + 1) * dd, &lm->dep_emb[0]))) return fail(rc);
        for (int k = 1; k < c.dep_q; ++k)
            if ((rc = load_copy(lm, W, "depformer_emb." + std::to_string(k - 1) + ".weight", 2, (size_t)(c.card + 1) * dd, &lm->dep_emb[k]))) return fail(rc);
    }
    // temporal transformer (lm.py:146-158)
    lm->layers.resize(c.num_layers);
    for (int l = 0; l < c.num_layers; ++l) {
        LayerW& L = lm->layers[l];
        std::string p = "transformer.layers." + std::to_string(l);
        if ((rc = load_linear(lm, W, p + ".self_attn.in_projs.0.weight", 3 * d, d, 0, &L.in_proj))) return fail(rc);
        if ((rc = load_linear(lm, W, p + ".self_attn.out_projs.0.weight", d, d, 0, &L.out_proj))) return fail(rc);
        if ((rc = load_linear(lm, W, p + ".gating.linear_in.weight", 2 * c.ffn_hidden, d, c.ffn_hidden, &L.ffn_in))) return fail(rc);
        if ((rc = load_linear(lm, W, p + ".gating.linear_out.weight", d, c.ffn_hidden, 0, &L.ffn_out))) return fail(rc);
        if ((rc = load_copy(lm, W, p + ".norm1.alpha", 3, d, &L.n1))) return fail(rc);
        if ((rc = load_copy(lm, W, p + ".norm2.alpha", 3, d, &L.n2))) return fail(rc);
        if (c.cross_attention) {
            if (d % lm->T) return fail(mmi_fail(MMI_ERR_UNSUPPORTED, "cross-attention needs dim to be a multiple of the GEMM tile"));
            if ((rc = load_linear(lm, W, p + ".cross_attention.in_projs.0.weight", 3 * d, d, 0, &L.x_in))) return fail(rc);
            if ((rc = load_linear(lm, W, p + ".cross_attention.out_projs.0.weight", d, d, 0, &L.x_out))) return fail(rc);
            if (L.x_in.wq) return fail(mmi_fail(MMI_ERR_UNSUPPORTED, "cross-attention layers with quantised linears are not supported"));
            if ((rc = load_copy(lm, W, p + ".norm_cross.weight", 1, d, &L.nx_w))) return fail(rc);
            if ((rc = load_copy(lm, W, p + ".norm_cross.bias", 1, d, &L.nx_b))) return fail(rc);
            const int ntq = d / lm->T;                               // n-tiles of the query rows; the rest are keys | values
            L.x_q = L.x_in;  L.x_q.N = d;      L.x_q.NT = ntq;
            L.x_kv = L.x_in; L.x_kv.N = 2 * d; L.x_kv.NT = 2 * ntq;
            L.x_kv.wp = L.x_in.wp + (size_t)ntq * L.x_in.KSTEPS * 64;
            L.x_q.bytes = L.x_in.bytes / 3; L.x_kv.bytes = L.x_in.bytes - L.x_q.bytes;
        }
    }
    if ((rc = load_copy(lm, W, "out_norm.alpha", 3, d, &lm->out_norm))) return fail(rc);
    if ((rc = load_linear(lm, W, "text_linear.weight", c.text_card_out, d, 0, &lm->text_linear))) return fail(rc);
    if (c.extra_heads_num_heads > 0) {   // lm.py:224-226: nn.Linear(dim, extra_heads_dim, bias=False) read by step_with_extra_heads
        if (c.extra_heads_dim < 1 || c.extra_heads_dim > 64) return fail(mmi_fail(MMI_ERR_UNSUPPORTED, "extra_heads_dim must be 1..64"));
        const size_t per = (size_t)c.extra_heads_dim * d;
        if (hipSuccess != lm->wts.alloc(&lm->extra_heads_all, per * c.extra_heads_num_heads)) return fail(mmi_fail(MMI_ERR_HIP, "out of device memory"));
        for (int i = 0; i < c.extra_heads_num_heads; ++i)
            if ((rc = load_copy(lm, W, "extra_heads." + std::to_string(i) + ".weight", 2, per, nullptr, lm->extra_heads_all + per * i))) return fail(rc);
    }
    // depformer (lm.py:179-232; per-step weights transformer.py:291-318)
    lm->dep_in.resize(c.dep_q);
    lm->dep_lin.resize(c.dep_q);
    {
        // depformer_in[k] all read transformer_out, so they are packed back to back and run as ONE GEMM with
        // dep_q * depformer_dim output features ahead of the micro-step loop (build_program).  Needs whole n-tiles per step.
        const bool group = c.dep_q > 0 && dd % lm->T == 0 && !getenv("MMI_NO_DEP_IN_GROUP");
        uint8_t* wp_all = nullptr;
        float *scale_all = nullptr, *scb_all = nullptr;
        size_t per = 0;
        if (group) {
            const int NT = dd / lm->T, ksteps = mmi_cdiv(d, mmi_kstep(lm->T));
            per = lm->q8 >= 1 ? (size_t)NT * mmi_cdiv(ksteps, 2) * 1024 : (size_t)NT * ksteps * 512 * sizeof(uint16_t);
            if (lm->wts.alloc(&wp_all, per * c.dep_q) != hipSuccess) return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (depformer_in)"));
            if (lm->q8 >= 1 && lm->wts.alloc(&scale_all, (size_t)dd * c.dep_q) != hipSuccess)
                return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (depformer_in scales)"));
            if (lm->q8 == 1 && lm->wts.alloc(&scb_all, (size_t)dd * c.dep_q) != hipSuccess)
                return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (depformer_in scales)"));
        }
        for (int k = 0; k < c.dep_q; ++k) {
            if ((rc = load_linear(lm, W, "depformer_in." + std::to_string(k) + ".weight", dd, d, 0, &lm->dep_in[k],
                                  group ? wp_all + per * k : nullptr, scale_all ? scale_all + (size_t)dd * k : nullptr,
                                  scb_all ? scb_all + (size_t)dd * k : nullptr)))
                return fail(rc);
        }
        for (int k = 1; group && k < c.dep_q; ++k)
            if (lm->dep_in[k].xinv != lm->dep_in[0].xinv)
                return fail(mmi_fail(MMI_ERR_UNSUPPORTED, "the depformer_in linears read the same tensor and must share one input_scale"));
        if (group) {
            lm->dep_in_all = lm->dep_in[0];
            lm->dep_in_all.N = dd * c.dep_q;
            lm->dep_in_all.NT = lm->dep_in[0].NT * c.dep_q;
            lm->dep_in_all.bytes = lm->dep_in[0].bytes * c.dep_q;
            lm->dep_in_grouped = true;
        }
    }
    // the depth transformer's own tile (mmi_lm::Td): two 16-row batch tiles at 17..32 sessions.  Needs the grouped depformer_in
    // (its row-major output is the boundary between the two tiles) and bf16 weights (the int8 / fp8 forms of the 16-row kernels
    // are instantiated for one batch tile only)
    lm->Td = lm->T;
    {
        const char* e = getenv("MMI_DEP_TILE");
        if (lm->T == 32 && max_batch <= 32 && lm->q8 == 0 && lm->dep_in_grouped && dd % 16 == 0 && e && atoi(e) == 16) lm->Td = 16;
    }
    const int Td = lm->Td;
    for (int k = 0; k < c.dep_q; ++k)
        if ((rc = load_linear(lm, W, "linears." + std::to_string(k) + ".weight", c.card, dd, 0, &lm->dep_lin[k], nullptr, nullptr, nullptr, Td))) return fail(rc);
    lm->dep_layers.resize(c.depformer_num_layers);
    for (int l = 0; l < c.depformer_num_layers; ++l) {
        DepLayerW& L = lm->dep_layers[l];
        std::string p = "depformer.layers." + std::to_string(l);
        L.in_proj.resize(c.dep_q); L.out_proj.resize(c.dep_q); L.ffn_in.resize(c.dep_q); L.ffn_out.resize(c.dep_q);
        for (int k = 0; k < c.dep_q; ++k) {
            std::string ks = std::to_string(k);
            if ((rc = load_linear(lm, W, p + ".self_attn.in_projs." + ks + ".weight", 3 * dd, dd, 0, &L.in_proj[k], nullptr, nullptr, nullptr, Td))) return fail(rc);
            if ((rc = load_linear(lm, W, p + ".self_attn.out_projs." + ks + ".weight", dd, dd, 0, &L.out_proj[k], nullptr, nullptr, nullptr, Td))) return fail(rc);
            if ((rc = load_linear(lm, W, p + ".gating." + ks + ".linear_in.weight", 2 * c.depformer_ffn_hidden, dd, c.depformer_ffn_hidden, &L.ffn_in[k], nullptr, nullptr, nullptr, Td))) return fail(rc);
            if ((rc = load_linear(lm, W, p + ".gating." + ks + ".linear_out.weight", dd, c.depformer_ffn_hidden, 0, &L.ffn_out[k], nullptr, nullptr, nullptr, Td))) return fail(rc);
        }
        if ((rc = load_copy(lm, W, p + ".norm1.alpha", 3, dd, &L.n1))) return fail(rc);
        if ((rc = load_copy(lm, W, p + ".norm2.alpha", 3, dd, &L.n2))) return fail(rc);
    }
    if (hipSuccess != lm->wts.alloc(&lm->delays_dev, (size_t)lm->NC)) return fail(mmi_fail(MMI_ERR_HIP, "out of device memory"));
    if (hipSuccess != hipMemcpy(lm->delays_dev, c.delays, lm->NC * sizeof(int), hipMemcpyHostToDevice)) return fail(mmi_fail(MMI_ERR_HIP, "memcpy"));
    if (hipDeviceSynchronize() != hipSuccess) return fail(mmi_fail(MMI_ERR_HIP, "weight packing failed"));
    if (lm->use_graph && hipStreamCreate(&lm->cap_stream) != hipSuccess) return fail(mmi_fail(MMI_ERR_HIP, "hipStreamCreate failed"));
    *out = lm;
    return MMI_OK;
}

extern "C" void mmi_lm_destroy(mmi_lm* lm) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return;
    mmi_lm_streaming_stop(lm);
    lm->wts.release();
    for (auto& e : lm->ev_pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    for (auto& e : lm->site_ev) { if (e.ev) hipEventDestroy(e.ev); }
    if (lm->cap_stream) hipStreamDestroy(lm->cap_stream);
    delete lm;
}

extern "C" int mmi_lm_get_cfg(const mmi_lm* lm, mmi_lm_cfg* out) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    *out = lm->cfg;
    return MMI_OK;
}

extern "C" int mmi_lm_streaming_start(mmi_lm* lm, int32_t batch, const mmi_sampling* sampling, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    return mmi_lm_streaming_start_guided(lm, batch, sampling, nullptr, stream);
}

extern "C" int mmi_lm_streaming_start_guided(mmi_lm* lm, int32_t batch, const mmi_sampling* sampling, const mmi_guidance* guide,
                                             mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !sampling) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (lm->streaming) return mmi_fail(MMI_ERR_STATE, "already streaming");
    const bool guided = guide && guide->cfg_coef != 1.0f;
    const int rows = guided ? 2 * batch : batch;              // lm.py:646-647 `batch_size *= 2`
    if (batch <= 0 || rows > lm->max_batch)
        return mmi_fail(MMI_ERR_SHAPE, guided ? "guidance runs two model rows per session: 2 * batch exceeds max_batch" : "batch exceeds max_batch");
    if (sampling->top_k > 256 || sampling->top_k_text > 256) return mmi_fail(MMI_ERR_UNSUPPORTED, "top_k > 256");
    if (sampling->top_k < 0 || sampling->top_k_text < 0) return mmi_fail(MMI_ERR_INVALID, "top_k must be >= 0 (0 = no top-k)");
    hipStream_t s = (hipStream_t)stream;
    const mmi_lm_cfg& c = lm->cfg;
    lm->batch = rows;
    lm->gen_batch = batch;
    lm->cfg_coef = guided ? guide->cfg_coef : 1.f;
    lm->cfg_no_text = guided && guide->cfg_is_no_text ? 1 : 0;
    lm->masked_until = nullptr;
    lm->cond = nullptr;
    lm->samp = *sampling;
    lm->kmax = sampling->top_k > sampling->top_k_text ? sampling->top_k : sampling->top_k_text;
    if (lm->kmax < 1) lm->kmax = 1;
    lm->offset_cpu = 0;
    lm->depth_bound = 0;
    lm->precapture_failed = false;
    const int G = batch;
    const int B = rows, d = c.dim, H = c.num_heads, Dh = d / H, dd = c.depformer_dim, Hd = c.depformer_num_heads, Dhd = dd / Hd;
    const int NS = attn_splits(c, B);
    auto fail = [&](int code) { lm->streaming = true; mmi_lm_streaming_stop(lm); return code; };
    MmiArena& A = lm->st;
    const size_t kvn = (size_t)c.num_layers * B * H * c.context * Dh / (c.kv_cache_dtype == MMI_F8E4M3 ? 2 : 1);   // uint16 units
    const size_t dkvn = (size_t)c.depformer_num_layers * B * Hd * c.dep_q * Dhd;
    bool ok = true;
    ok &= hipSuccess == A.alloc(&lm->exec, (size_t)G);
    ok &= hipSuccess == A.alloc(&lm->offsets, (size_t)G);
    lm->offsets_m = lm->offsets;
    if (guided) ok &= hipSuccess == A.alloc(&lm->offsets_m, (size_t)B);
    if (guided && guide->cfg_is_masked_until) ok &= hipSuccess == A.alloc(&lm->masked_until, (size_t)G);
    if (guide && guide->condition_sum) ok &= hipSuccess == A.alloc(&lm->cond, (size_t)B * d);
    lm->xkv = nullptr;
    lm->cross_len = 0;
    if (c.cross_attention) {
        if (!guide || !guide->condition_cross || guide->cross_len <= 0)
            return fail(mmi_fail(MMI_ERR_INVALID, "the model has cross-attention layers: mmi_guidance.condition_cross is required"));   // transformer.py:793-795
        lm->cross_len = guide->cross_len;
        ok &= hipSuccess == A.alloc(&lm->xkv, (size_t)c.num_layers * B * lm->cross_len * 2 * d);
    } else if (guide && guide->condition_cross) {
        return fail(mmi_fail(MMI_ERR_INVALID, "a cross-attention condition was given to a model without cross-attention layers"));
    }
    ok &= hipSuccess == A.alloc(&lm->cache, (size_t)G * lm->NC * lm->CT);
    ok &= hipSuccess == A.alloc(&lm->user_i32, (size_t)G * (c.n_q - c.dep_q));
    ok &= hipSuccess == A.alloc(&lm->tokens, (size_t)B * lm->NC);
    ok &= hipSuccess == A.alloc(&lm->text_tok, (size_t)G);
    ok &= hipSuccess == A.alloc(&lm->audio_tok, (size_t)G * c.dep_q);
    ok &= hipSuccess == A.alloc(&lm->out_i32, (size_t)G * (c.dep_q + 1));
    ok &= hipSuccess == A.alloc(&lm->x, packed_elems(lm, d));
    ok &= hipSuccess == A.alloc(&lm->xn, packed_elems(lm, d));
    ok &= hipSuccess == A.alloc(&lm->qrot, (size_t)B * d);
    ok &= hipSuccess == A.alloc(&lm->att, packed_elems(lm, d));
    ok &= hipSuccess == A.alloc(&lm->hb, packed_elems(lm, c.ffn_hidden));
    ok &= hipSuccess == A.alloc(&lm->tout, packed_elems(lm, d));
    ok &= hipSuccess == A.alloc(&lm->text_logits, (size_t)B * c.text_card_out);
    ok &= hipSuccess == A.alloc(&lm->kc, kvn);
    ok &= hipSuccess == A.alloc(&lm->vc, kvn);
    ok &= hipSuccess == A.alloc(&lm->opart, (size_t)B * H * NS * Dh);
    ok &= hipSuccess == A.alloc(&lm->ml, (size_t)B * H * NS * 2);
    ok &= hipSuccess == A.alloc(&lm->attn_done, (size_t)B * H);
    ok &= hipSuccess == A.alloc(&lm->partial, (size_t)4 * B * (d > dd ? d : dd));
    ok &= hipSuccess == A.alloc(&lm->rope, (size_t)B * Dh);
    lm->htap = nullptr;
    if (lm->hidden_taps) ok &= hipSuccess == A.alloc(&lm->htap, (size_t)2 * B * d);
    {   // int8 activations (see mmi_lm::act8)
        const char* e8 = getenv("MMI_Q8_ACT");
        lm->act8 = lm->q8 == 1 && !c.cross_attention && !(e8 && e8[0] == 'b');
        lm->xnq = lm->attq = lm->hbq = lm->toutq = lm->dxnq = nullptr;
        lm->sx_xn = lm->sx_tout = lm->sx_dxn = lm->sx_att = lm->sx_hb = nullptr;
        if (lm->act8) {
            const size_t mtiles = (size_t)mmi_cdiv(B, lm->T);
            auto qbytes = [&](int features) { return mtiles * (size_t)(packed_ksteps(lm, features) / 2) * 1024; };
            const size_t rows = mtiles * lm->T;
            ok &= hipSuccess == A.alloc(&lm->xnq, qbytes(d));
            ok &= hipSuccess == A.alloc(&lm->attq, qbytes(d));
            ok &= hipSuccess == A.alloc(&lm->hbq, qbytes(c.ffn_hidden > c.depformer_ffn_hidden ? c.ffn_hidden : c.depformer_ffn_hidden));
            ok &= hipSuccess == A.alloc(&lm->toutq, qbytes(d));
            ok &= hipSuccess == A.alloc(&lm->dxnq, qbytes(dd > 0 ? dd : 8));
            ok &= hipSuccess == A.alloc(&lm->sx_xn, rows);
            ok &= hipSuccess == A.alloc(&lm->sx_tout, rows);
            ok &= hipSuccess == A.alloc(&lm->sx_dxn, rows);
            ok &= hipSuccess == A.alloc(&lm->sx_att, rows);
            ok &= hipSuccess == A.alloc(&lm->sx_hb, rows);
        }
    }
    ok &= hipSuccess == A.alloc(&lm->dx, packed_elems_t(lm, lm->Td, dd));
    ok &= hipSuccess == A.alloc(&lm->dxn, packed_elems_t(lm, lm->Td, dd));
    ok &= hipSuccess == A.alloc(&lm->dqkv, (size_t)B * 3 * dd);
    ok &= hipSuccess == A.alloc(&lm->datt, packed_elems_t(lm, lm->Td, dd));
    ok &= hipSuccess == A.alloc(&lm->dhb, packed_elems_t(lm, lm->Td, c.depformer_ffn_hidden));
    ok &= hipSuccess == A.alloc(&lm->dlogits, (size_t)c.dep_q * B * c.card);
    ok &= hipSuccess == A.alloc(&lm->dpre, (size_t)B * c.dep_q * dd);
    ok &= hipSuccess == A.alloc(&lm->dkc, dkvn);
    ok &= hipSuccess == A.alloc(&lm->dvc, dkvn);
    ok &= hipSuccess == A.alloc(&lm->noise, (size_t)G * (1 + c.dep_q) * lm->kmax);
    ok &= hipSuccess == A.alloc(&lm->use_noise, (size_t)1);
    ok &= hipSuccess == A.alloc(&lm->forced, (size_t)G * (1 + c.dep_q));
    ok &= hipSuccess == A.alloc(&lm->use_forced, (size_t)1);
    ok &= hipSuccess == A.alloc(&lm->rng, (size_t)2);
    if (!ok) return fail(mmi_fail(MMI_ERR_HIP, "out of device memory (LM streaming state)"));
    MMI_HIP_CHECK(hipMemsetAsync(lm->exec, 1, G, s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->offsets, 0, G * sizeof(long), s));
    if (guided) MMI_HIP_CHECK(hipMemsetAsync(lm->offsets_m, 0, B * sizeof(long), s));
    if (lm->masked_until) {   // host int64[batch] (LMGen(cfg_is_masked_until=[...]), lm.py:637-640)
        std::vector<int> mu(G);
        for (int i = 0; i < G; ++i) mu[i] = (int)guide->cfg_is_masked_until[i];
        MMI_HIP_CHECK(hipMemcpy(lm->masked_until, mu.data(), G * sizeof(int), hipMemcpyHostToDevice));
    }
    if (lm->cond) MMI_HIP_CHECK(hipMemcpyAsync(lm->cond, guide->condition_sum, (size_t)B * d * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
    MMI_LAUNCH(k_fill_i32, mmi_cdiv(G * lm->NC * lm->CT, 256), 256, 0, s, lm->cache, -2, (long)G * lm->NC * lm->CT);   // lm.py:608-613
    MMI_HIP_CHECK(hipMemsetAsync(lm->kc, 0, kvn * sizeof(uint16_t), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->vc, 0, kvn * sizeof(uint16_t), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->attn_done, 0, (size_t)B * H * sizeof(unsigned), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->dkc, 0, dkvn * sizeof(uint16_t), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->dvc, 0, dkvn * sizeof(uint16_t), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->use_noise, 0, sizeof(int), s));
    lm->noise_on = false;
    MMI_HIP_CHECK(hipMemsetAsync(lm->use_forced, 0, sizeof(int), s));
    lm->forced_armed = false;
    {   // packed activations: the padding rows / columns of a fragment are never written and must read as zero
        struct { uint16_t* p; int f; int T; } pk[] = {{lm->x, d, lm->T}, {lm->xn, d, lm->T}, {lm->att, d, lm->T}, {lm->hb, c.ffn_hidden, lm->T},
                                                      {lm->tout, d, lm->T}, {lm->dx, dd, lm->Td}, {lm->dxn, dd, lm->Td}, {lm->datt, dd, lm->Td},
                                                      {lm->dhb, c.depformer_ffn_hidden, lm->Td}};
        for (auto& e : pk) MMI_HIP_CHECK(hipMemsetAsync(e.p, 0, packed_elems_t(lm, e.T, e.f) * sizeof(uint16_t), s));
    }
    if (lm->act8) {   // the int8 operands: like their bf16 twins, padding rows / k-steps are never written and must read as zero
        const size_t mtiles = (size_t)mmi_cdiv(B, lm->T), rows = mtiles * lm->T;
        auto qbytes = [&](int features) { return mtiles * (size_t)(packed_ksteps(lm, features) / 2) * 1024; };
        struct { uint8_t* p; int f; } qk[] = {{lm->xnq, d}, {lm->attq, d}, {lm->hbq, c.ffn_hidden > c.depformer_ffn_hidden ? c.ffn_hidden : c.depformer_ffn_hidden},
                                              {lm->toutq, d}, {lm->dxnq, dd > 0 ? dd : 8}};
        for (auto& e : qk) MMI_HIP_CHECK(hipMemsetAsync(e.p, 0, qbytes(e.f), s));
        for (float* p : {lm->sx_xn, lm->sx_tout, lm->sx_dxn, lm->sx_att, lm->sx_hb}) MMI_HIP_CHECK(hipMemsetAsync(p, 0, rows * sizeof(float), s));
    }
    MMI_HIP_CHECK(hipMemsetAsync(lm->text_tok, 0, G * sizeof(int), s));
    MMI_HIP_CHECK(hipMemsetAsync(lm->audio_tok, 0, (size_t)G * c.dep_q * sizeof(int), s));
    unsigned long long r0[2] = {sampling->seed, 0ull};
    MMI_HIP_CHECK(hipMemcpyAsync(lm->rng, r0, sizeof(r0), hipMemcpyHostToDevice, s));
    MMI_CHECK_LAUNCH();
    int rc = 0;
    if (lm->xkv && (rc = project_cross_source(lm, reinterpret_cast<const uint16_t*>(guide->condition_cross), s))) return fail(rc);
    rc = build_program(lm);
    if (rc) return fail(rc);
    MMI_HIP_CHECK(hipStreamSynchronize(s));
    if (const char* tp = getenv("MMI_DEBUG_TRACE")) {
        static int session = 0;
        lm->trace_name = std::string(tp) + "." + std::to_string(session++);
        lm->trace_file = fopen(lm->trace_name.c_str(), "w");
        lm->trace_prev.assign(lm->st.ptrs.size(), 0ull);
        lm->trace_step = 0;
        if (lm->trace_file && hipMalloc((void**)&lm->trace_dev, lm->st.ptrs.size() * sizeof(unsigned long long)) != hipSuccess) {
            fclose(lm->trace_file);
            lm->trace_file = nullptr;
        }
    }
    lm->streaming = true;
    return MMI_OK;
}

extern "C" int mmi_lm_streaming_stop(mmi_lm* lm) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!lm->streaming) return MMI_OK;
    hipDeviceSynchronize();
    if (lm->trace_file) { fclose(lm->trace_file); lm->trace_file = nullptr; }
    if (lm->trace_dev) { hipFree(lm->trace_dev); lm->trace_dev = nullptr; }
    lm->prog.clear();
    lm->st.release();
    lm->streaming = false;
    lm->batch = 0;
    lm->gen_batch = 0;
    lm->cfg_coef = 1.f;
    return MMI_OK;
}

extern "C" int mmi_lm_set_exec_mask(mmi_lm* lm, const uint8_t* mask, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !mask) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    MMI_HIP_CHECK(hipMemcpyAsync(lm->exec, mask, lm->gen_batch, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MMI_OK;
}

extern "C" int mmi_lm_reset(mmi_lm* lm, const uint8_t* mask, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    TokArgs t = tok_args(lm);
    MMI_LAUNCH(k_lm_reset, mmi_cdiv(lm->gen_batch, 64), 64, 0, (hipStream_t)stream, t, mask, lm->exec);
    MMI_CHECK_LAUNCH();
    lm->offset_cpu = 0;   // lm.py:540: any reset, even partial, zeroes the host-side step counter
    return MMI_OK;
}

extern "C" int mmi_lm_step(mmi_lm* lm, const int64_t* user_codes, int32_t n_user, int64_t* out_tokens, float* opt_text_logits,
                           float* opt_audio_logits, const float* opt_noise, int32_t batch, int32_t* valid, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !user_codes || !out_tokens) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming)
        return mmi_fail(MMI_ERR_STATE, "You should wrap those calls with a `with lm_gen.streaming(): ...`.");   // lm.py:673-676
    if (batch != lm->gen_batch) return mmi_fail(MMI_ERR_SHAPE, "Got a different batch size than the streaming batch");   // lm.py:681
    const mmi_lm_cfg& c = lm->cfg;
    const int need_user = c.n_q - c.dep_q;
    if (n_user < need_user) return mmi_fail(MMI_ERR_SHAPE, "not enough user tokens");   // lm.py:683-686
    hipStream_t s = (hipStream_t)stream;
    const int B = batch;
    MMI_LAUNCH(k_i64_to_i32, mmi_cdiv(B * need_user, 256), 256, 0, s, (const long*)user_codes, (long)n_user, lm->user_i32, B, need_user);
    if (opt_noise && lm->samp.use_sampling && (lm->samp.top_k == 0 || lm->samp.top_k_text == 0))
        return mmi_fail(MMI_ERR_UNSUPPORTED, "supplied noise is indexed by rank in the top-k: not available with top_k = 0");
    if (opt_noise) {
        MMI_HIP_CHECK(hipMemcpyAsync(lm->noise, opt_noise, (size_t)B * (1 + c.dep_q) * lm->kmax * sizeof(float), hipMemcpyDeviceToDevice, s));
        MMI_HIP_CHECK(hipMemsetAsync(lm->use_noise, 1, 1, s));
        lm->noise_on = true;
    } else if (lm->noise_on) {          // the device flag only changes when the caller starts / stops supplying noise: no fill per step
        MMI_HIP_CHECK(hipMemsetAsync(lm->use_noise, 0, sizeof(int), s));
        lm->noise_on = false;
    }
    MMI_CHECK_LAUNCH();
    int rc;
    lm->prog.variant = attn_variant(lm, lm->attn_ns);
    const bool hooked = lm->hooks.on_text_logits || lm->hooks.on_text_token || lm->hooks.on_audio_tokens;
    if (hooked) {
        rc = run_step_with_hooks(lm, s);
        if (rc) {           // a hook aborted the step half-way: it is NOT committed (ring, offsets unchanged), and the forcing armed for
            hipMemsetAsync(lm->use_forced, 0, sizeof(int), s);   // this step does not leak into the next one
            lm->forced_armed = false;
        }
    }
    else if (lm->trace_file) {
        lm->prog.tap = [lm](size_t i, bool begin, hipStream_t st) {
            if (begin) return;
            const size_t n = lm->st.ptrs.size();
            hipMemsetAsync(lm->trace_dev, 0, n * sizeof(unsigned long long), st);
            for (size_t a = 0; a < n; ++a) {
                const long nb = (long)lm->st.sizes[a];
                int blocks = (int)((nb / 4 + 255) / 256);
                if (blocks > 2048) blocks = 2048;
                if (blocks < 1) blocks = 1;
                hipLaunchKernelGGL(k_checksum, blocks, 256, 0, st, (const uint8_t*)lm->st.ptrs[a], nb, lm->trace_dev + a);
            }
            std::vector<unsigned long long> now(n);
            hipStreamSynchronize(st);
            hipMemcpy(now.data(), lm->trace_dev, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            for (size_t a = 0; a < n; ++a)
                if (now[a] != lm->trace_prev[a])
                    fprintf(lm->trace_file, "%ld %zu %s %zu %zu %016llx\n", lm->trace_step, i, lm->prog.sites[i].c_str(), a, lm->st.sizes[a], now[a]);
            lm->trace_prev = now;
            // MMI_DEBUG_DUMP="step:op:alloc[,alloc...]": those allocations, as they are after that op, to <prefix>.<session>.a<alloc>
            if (const char* dd = getenv("MMI_DEBUG_DUMP")) {
                long ds = -1, dop = -1;
                int used = 0;
                if (sscanf(dd, "%ld:%ld:%n", &ds, &dop, &used) >= 2 && ds == lm->trace_step && dop == (long)i) {
                    const char* q = dd + used;
                    while (*q) {
                        const long a = strtol(q, (char**)&q, 10);
                        if (*q == ',') ++q;
                        if (a < 0 || a >= (long)n) break;
                        std::vector<char> host(lm->st.sizes[a]);
                        hipMemcpy(host.data(), lm->st.ptrs[a], host.size(), hipMemcpyDeviceToHost);
                        if (FILE* df = fopen((lm->trace_name + ".a" + std::to_string(a)).c_str(), "wb")) { fwrite(host.data(), 1, host.size(), df); fclose(df); }
                    }
                }
            }
        };
        rc = lm->prog.run_eager(s);
        lm->prog.tap = nullptr;
        lm->trace_step += 1;
    }
    else if (lm->phase_fn) {
        auto fn = lm->phase_fn; void* user = lm->phase_user;
        rc = lm->prog.run_split(s, lm->use_graph && !lm->profiling, lm->cap_stream, lm->op_depformer, [fn, user](hipStream_t st) {
            return fn(user, (mmi_stream)st) ? mmi_fail(MMI_ERR_INVALID, "the phase callback reported an error") : (int)MMI_OK;
        });
    } else rc = lm->prog.run(s, lm->use_graph && !lm->profiling, lm->cap_stream);
    if (rc) return rc;
    if (lm->attn_ns > 1 && lm->use_graph && !lm->profiling && !hooked && attn_wave_kernel() && !getenv("MMI_NO_PRECAPTURE")) {
        // both attention programs exist from the stream's first step on: the switch at depth solo_rows is then a graph launch
        // like any other (without this the deep program was captured + instantiated ~61 s into a live single-session stream)
        // The step above is already committed: a failure to capture the OTHER program (out of memory at instantiate, ...) must not
        // turn it into a failed call - the program is then captured when first needed, as before round 5.  The capture
        // re-runs every op closure: its GEMMs are not launches of this stream (mmi_lm_stat 0).
        const bool split = lm->phase_fn != nullptr;
        const long counted = lm->xlds_launches;
        for (int v = 0; v < MmiProgram::NV && !lm->precapture_failed; ++v)
            if (!lm->prog.ready(v, split) && lm->prog.precapture(v, lm->cap_stream, split, lm->op_depformer) != MMI_OK) {
                lm->precapture_failed = true;
                (void)hipGetLastError();
                fprintf(stderr, "moshi_mi: the second attention step program could not be captured ahead of time (%s); it will be captured when first needed\n",
                        mmi_last_error());
            }
        lm->xlds_launches = counted;
    }
    MMI_LAUNCH(k_i32_to_i64, mmi_cdiv(B * (c.dep_q + 1), 256), 256, 0, s, (const int*)lm->out_i32, (long*)out_tokens, B * (c.dep_q + 1));
    if (opt_text_logits)
        MMI_LAUNCH(k_bf16_to_f32, (int)mmi_cdiv64((int64_t)B * c.text_card_out, 256), 256, 0, s, (const uint16_t*)lm->text_logits, opt_text_logits, (long)B * c.text_card_out);
    if (opt_audio_logits) {
        // internal layout [dep_q][B][card] -> caller's [B][dep_q][card]
        for (int k = 0; k < c.dep_q; ++k)
            for (int b = 0; b < B; ++b)
                MMI_LAUNCH(k_bf16_to_f32, mmi_cdiv(c.card, 256), 256, 0, s, (const uint16_t*)(lm->dlogits + ((size_t)k * lm->batch + b) * c.card),
                           opt_audio_logits + ((size_t)b * c.dep_q + k) * c.card, (long)c.card);
    }
    MMI_CHECK_LAUNCH();
    if (lm->forced_armed) {   // forcing applies to one step only
        MMI_HIP_CHECK(hipMemsetAsync(lm->use_forced, 0, sizeof(int), s));
        lm->forced_armed = false;
    }
    lm->offset_cpu += 1;
    if (lm->depth_bound < (1L << 40)) lm->depth_bound += 1;
    if (valid) *valid = lm->offset_cpu > lm->max_delay ? 1 : 0;   // lm.py:774-776
    return MMI_OK;
}

extern "C" int mmi_lm_set_phase_callback(mmi_lm* lm, int (*fn)(void*, mmi_stream), void* user) {
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    lm->phase_fn = fn;
    lm->phase_user = user;
    return MMI_OK;
}

extern "C" int mmi_lm_set_hooks(mmi_lm* lm, const mmi_lm_hooks* hooks) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (hooks) lm->hooks = *hooks;
    else lm->hooks = mmi_lm_hooks{nullptr, nullptr, nullptr, nullptr};
    return MMI_OK;
}

extern "C" int mmi_lm_set_hidden_taps(mmi_lm* lm, int32_t on) {
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (lm->streaming) return mmi_fail(MMI_ERR_STATE, "mmi_lm_set_hidden_taps: call before streaming_start");
    lm->hidden_taps = on != 0;
    return MMI_OK;
}

extern "C" int mmi_lm_get_hidden_taps(mmi_lm* lm, void* buf, int64_t nbytes, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !buf) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming || !lm->htap) return mmi_fail(MMI_ERR_STATE, "hidden taps were not enabled before streaming_start");
    const int64_t want = (int64_t)2 * lm->batch * lm->cfg.dim * 2;
    if (nbytes != want) return mmi_fail(MMI_ERR_SHAPE, "mmi_lm_get_hidden_taps: the buffer holds " + std::to_string(nbytes) + " bytes, the taps " + std::to_string(want));
    MMI_HIP_CHECK(hipMemcpyAsync(buf, lm->htap, (size_t)want, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MMI_OK;
}

extern "C" int32_t mmi_lm_has_hooks(const mmi_lm* lm) {
    return lm && (lm->hooks.on_text_logits || lm->hooks.on_text_token || lm->hooks.on_audio_tokens) ? 1 : 0;
}

extern "C" int mmi_lm_hook_io(mmi_lm* lm, int32_t which, int32_t write, void* buf, int64_t nbytes, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !buf) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming || !lm->in_hook) return mmi_fail(MMI_ERR_STATE, "mmi_lm_hook_io is only valid inside a step hook");
    hipStream_t s = (hipStream_t)stream;
    const mmi_lm_cfg& c = lm->cfg;
    const int G = lm->gen_batch;
    const int64_t want = which == 0 ? (int64_t)G * c.text_card_out * 2 : which == 1 ? (int64_t)G * 8 : which == 2 ? (int64_t)G * c.dep_q * 8 : -1;
    if (want >= 0 && nbytes != want)
        return mmi_fail(MMI_ERR_SHAPE, "mmi_lm_hook_io: the buffer holds " + std::to_string(nbytes) + " bytes, the tensor " + std::to_string(want));
    if (which == 0) {            // (guided) text logits, rows [0, G) of the model's logits buffer
        const size_t n = (size_t)G * c.text_card_out * sizeof(uint16_t);
        if (write) MMI_HIP_CHECK(hipMemcpyAsync(lm->text_logits, buf, n, hipMemcpyDeviceToDevice, s));
        else MMI_HIP_CHECK(hipMemcpyAsync(buf, lm->text_logits, n, hipMemcpyDeviceToDevice, s));
        return MMI_OK;
    }
    int* tok = which == 1 ? lm->text_tok : (which == 2 ? lm->audio_tok : nullptr);
    if (!tok) return mmi_fail(MMI_ERR_INVALID, "mmi_lm_hook_io: which must be 0, 1 or 2");
    const int cols = which == 1 ? 1 : c.dep_q;
    if (cols == 0) return MMI_OK;
    if (write) MMI_LAUNCH(k_i64_to_i32, mmi_cdiv(G * cols, 256), 256, 0, s, (const long*)buf, (long)cols, tok, G, cols);
    else MMI_LAUNCH(k_i32_to_i64, mmi_cdiv(G * cols, 256), 256, 0, s, (const int*)tok, (long*)buf, G * cols);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

extern "C" int mmi_lm_force_next_tokens(mmi_lm* lm, const int64_t* tokens, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !tokens) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    hipStream_t s = (hipStream_t)stream;
    const int n = lm->gen_batch * (1 + lm->cfg.dep_q);
    MMI_LAUNCH(k_i64_to_i32, mmi_cdiv(n, 256), 256, 0, s, (const long*)tokens, (long)(1 + lm->cfg.dep_q), lm->forced, lm->gen_batch, 1 + lm->cfg.dep_q);
    MMI_CHECK_LAUNCH();
    MMI_HIP_CHECK(hipMemsetAsync(lm->use_forced, 1, 1, s));
    lm->forced_armed = true;
    return MMI_OK;
}

extern "C" int64_t mmi_lm_state_bytes(const mmi_lm* lm) { return lm && lm->streaming ? (int64_t)lm->st.bytes : 0; }

extern "C" int mmi_lm_state_save(mmi_lm* lm, void* dst, int64_t bytes, int64_t* host_word, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !dst || !host_word) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    if (bytes != (int64_t)lm->st.bytes) return mmi_fail(MMI_ERR_SHAPE, "snapshot buffer has the wrong size");
    MMI_HIP_CHECK(lm->st.save(dst, (hipStream_t)stream));
    *host_word = lm->offset_cpu;
    return MMI_OK;
}

extern "C" int mmi_lm_state_load(mmi_lm* lm, const void* src, int64_t bytes, int64_t host_word, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !src) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    if (bytes != (int64_t)lm->st.bytes) return mmi_fail(MMI_ERR_SHAPE, "snapshot taken from a different stream (batch / guidance)");
    MMI_HIP_CHECK(lm->st.load(src, (hipStream_t)stream));
    lm->offset_cpu = (long)host_word;
    {   // the snapshot's offsets bound the ring depth (which step program the next steps take): read them back - a restore is not
        // a per-frame call, and without this a shallow restored session paid the merge launch for the rest of the stream
        std::vector<long> off(lm->batch);
        MMI_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
        MMI_HIP_CHECK(hipMemcpy(off.data(), lm->offsets_m, (size_t)lm->batch * sizeof(long), hipMemcpyDeviceToHost));
        long mx = 0;
        for (long o : off) mx = o > mx ? o : mx;
        lm->depth_bound = mx;
    }
    lm->forced_armed = false;
    lm->noise_on = true;               // the snapshot carries its own use_noise word: the next step rewrites it
    return MMI_OK;
}

extern "C" int mmi_lm_extra_heads(mmi_lm* lm, float* probs, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !probs) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    const mmi_lm_cfg& c = lm->cfg;
    if (c.extra_heads_num_heads <= 0) return MMI_OK;
    MMI_LAUNCH(k_extra_heads, dim3(lm->batch, c.extra_heads_num_heads), 64, 0, (hipStream_t)stream, (const uint16_t*)lm->tout, lm->T,
               packed_ksteps(lm, c.dim), (const uint16_t*)lm->extra_heads_all, c.dim, c.extra_heads_dim, c.extra_heads_num_heads, probs);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

extern "C" int mmi_lm_model_rows(const mmi_lm* lm) { return lm ? lm->batch : 0; }
extern "C" int mmi_lm_streaming_batch(const mmi_lm* lm) { return lm && lm->streaming ? lm->gen_batch : 0; }
extern "C" int mmi_lm_device(const mmi_lm* lm) { return lm ? lm->device : -1; }

int64_t mmi_copy_launch_log(const std::vector<std::string>& log, char* buf, int64_t cap);   // api_common.hip

extern "C" int64_t mmi_lm_launch_list(const mmi_lm* lm, char* buf, int64_t cap) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !lm->streaming || !lm->prog.logged()) return 0;
    return mmi_copy_launch_log(lm->prog.launch_log(), buf, cap);
}

extern "C" int mmi_lm_seek(mmi_lm* lm, const int64_t* offsets, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !offsets) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!lm->streaming) return mmi_fail(MMI_ERR_STATE, "not streaming");
    const int G = lm->gen_batch;
    std::vector<long> off(lm->batch);
    long mx = 0;
    for (int b = 0; b < G; ++b) {
        if (offsets[b] < 0) return mmi_fail(MMI_ERR_INVALID, "negative offset");
        off[b] = (long)offsets[b];
        if (lm->batch > G) off[G + b] = off[b];      // the guidance twins follow their session
        mx = off[b] > mx ? off[b] : mx;
    }
    hipStream_t s = (hipStream_t)stream;
    MMI_HIP_CHECK(hipStreamSynchronize(s));
    MMI_HIP_CHECK(hipMemcpy(lm->offsets, off.data(), (size_t)G * sizeof(long), hipMemcpyHostToDevice));
    if (lm->offsets_m != lm->offsets) MMI_HIP_CHECK(hipMemcpy(lm->offsets_m, off.data(), (size_t)lm->batch * sizeof(long), hipMemcpyHostToDevice));
    lm->offset_cpu = mx;
    lm->depth_bound = mx;
    return MMI_OK;
}

// Parity tap: ONE linear of the model on caller-supplied rows, through the kernels the step uses for it (include/moshi_mi.h).
// Works on its own scratch (a streaming session is not disturbed) and synchronises.
// failures after the scratch arena exists leave through done(): the launches are drained and the arena released
#define MMI_DBG_HIP(call)                                                                                          \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return done(mmi_fail(MMI_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_))); \
    } while (0)
#define MMI_DBG_LAUNCHED()                                                                                         \
    do {                                                                                                           \
        hipError_t e_ = hipGetLastError();                                                                         \
        if (e_ != hipSuccess) return done(mmi_fail(MMI_ERR_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e_))); \
    } while (0)

extern "C" int mmi_lm_debug_linear(mmi_lm* lm, const char* weight_name, const char* alpha_name, int32_t path, const void* x_bf16,
                                   int32_t rows, void* out_bf16, int8_t* codes, float* absmax, void* norm_out, mmi_stream stream) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !weight_name || !x_bf16 || !out_bf16) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (rows <= 0 || rows > lm->max_batch) return mmi_fail(MMI_ERR_SHAPE, "mmi_lm_debug_linear: rows must be 1..max_batch");
    auto it = lm->linear_by_name.find(weight_name);
    if (it == lm->linear_by_name.end()) return mmi_fail(MMI_ERR_MISSING_WEIGHT, std::string("no packed linear named ") + weight_name);
    const GemmW g = it->second;
    const uint16_t* alpha = nullptr;
    const bool normed = path == MMI_DBG_NORM || path == MMI_DBG_NORM_FUSED;
    if (normed) {
        auto ia = alpha_name ? lm->vector_by_name.find(alpha_name) : lm->vector_by_name.end();
        if (ia == lm->vector_by_name.end()) return mmi_fail(MMI_ERR_MISSING_WEIGHT, "mmi_lm_debug_linear: the norm's alpha vector was not found");
        alpha = ia->second;
    }
    const char* e8 = getenv("MMI_Q8_ACT");
    const bool a8 = lm->q8 == 1 && !lm->cfg.cross_attention && !(e8 && e8[0] == 'b');
    if ((codes || absmax) && !(a8 && (path == MMI_DBG_PLAIN || path == MMI_DBG_SPLITK || path == MMI_DBG_NORM)))
        return mmi_fail(MMI_ERR_UNSUPPORTED, "mmi_lm_debug_linear: codes / absmax exist on the int8 x int8 paths that materialise the operand (plain, split-K, norm)");
    hipStream_t s = (hipStream_t)stream;
    const int T = g.T, mt = mmi_cdiv(rows, T), KS = mmi_kstep(T);
    const int K = g.K, N = g.N, kin = packed_ksteps_t(lm, T, K), kout = packed_ksteps_t(lm, T, N);
    const int nthK = [&] { int n = mmi_cdiv(K / 8, 64) * 64; return n > 1024 ? 1024 : n; }();
    MmiArena A;
    uint16_t *xp = nullptr, *yp = nullptr, *xres = nullptr, *zeros = nullptr, *outp = (uint16_t*)out_bf16;
    uint8_t* xq = nullptr;
    float *sx = nullptr, *partial = nullptr;
    const size_t xelems = (size_t)mt * kin * 512, oelems = (size_t)mt * kout * 512;
    bool ok = true;
    ok &= hipSuccess == A.alloc(&xp, xelems);
    ok &= hipSuccess == A.alloc(&yp, xelems);
    ok &= hipSuccess == A.alloc(&xq, (size_t)mt * (kin / 2 + 1) * 1024);
    ok &= hipSuccess == A.alloc(&sx, (size_t)mt * T);
    ok &= hipSuccess == A.alloc(&xres, oelems);
    ok &= hipSuccess == A.alloc(&zeros, (size_t)N + 8);
    ok &= hipSuccess == A.alloc(&partial, (size_t)4 * rows * N);
    auto done = [&](int rc) { hipStreamSynchronize(s); A.release(); return rc; };
    if (!ok) return done(mmi_fail(MMI_ERR_HIP, "out of device memory (mmi_lm_debug_linear)"));
    MMI_DBG_HIP(hipMemsetAsync(xp, 0, xelems * 2, s));
    MMI_DBG_HIP(hipMemsetAsync(yp, 0, xelems * 2, s));
    MMI_DBG_HIP(hipMemsetAsync(xq, 0, (size_t)mt * (kin / 2 + 1) * 1024, s));
    MMI_DBG_HIP(hipMemsetAsync(sx, 0, (size_t)mt * T * sizeof(float), s));
    MMI_DBG_HIP(hipMemsetAsync(xres, 0, oelems * 2, s));
    MMI_DBG_HIP(hipMemsetAsync(zeros, 0, ((size_t)N + 8) * 2, s));
    MMI_LAUNCH(k_pack_rows, mmi_cdiv(rows * K, 256), 256, 0, s, (const uint16_t*)x_bf16, rows, K, xp, T, kin);
    MMI_DBG_LAUNCHED();
    const uint16_t* operand = xp;              // the packed bf16 rows the GEMM (or its quantiser) reads
    const bool gated = g.gate != 0;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.out = outp; a.epi = gated ? MMI_EPI_GATE : MMI_EPI_STORE; a.B = rows; a.tok_rows = rows;
    a.out_mode = MMI_OUT_ROWMAJOR; a.out_ld = N; a.out_ksteps = kout;
    int rc = MMI_OK;
    if (path == MMI_DBG_NORM) {                // the norm launch of the step: y (+ its int8 copy) = rms_norm(x) * alpha
        MMI_LAUNCH(k_resid_rmsnorm, rows, nthK, 0, s, xp, (const float*)nullptr, 0, rows, alpha, yp, K, T, kin, 1e-8f,
                   a8 ? xq : (uint8_t*)nullptr, a8 ? sx : (float*)nullptr, (const float*)nullptr, (const float*)nullptr);
        MMI_DBG_LAUNCHED();
        operand = yp;
        if (norm_out) MMI_LAUNCH(k_unpack_rows, mmi_cdiv(rows * K, 256), 256, 0, s, (const uint16_t*)yp, rows, K, (uint16_t*)norm_out, T, kin);
    } else if (a8 && (path == MMI_DBG_PLAIN || path == MMI_DBG_SPLITK)) {
        MMI_LAUNCH(k_quant_rows_i8, rows, nthK, 0, s, (const uint16_t*)xp, rows, K, T, kin, xq, sx);
        MMI_DBG_LAUNCHED();
    }
    if (path == MMI_DBG_PLAIN || path == MMI_DBG_NORM || path == MMI_DBG_SPLITK) {
        a.xp = reinterpret_cast<const u32x4*>(operand);
        if (a8) { a.xp = reinterpret_cast<const u32x4*>(xq); a.sx = sx; a.wq = 3; }
        Pending pd;
        if (path == MMI_DBG_SPLITK) {
            const GemmPlan p = plan_gemm(g, true);
            if (gated || p.ksplit <= 1) return done(mmi_fail(MMI_ERR_UNSUPPORTED, "mmi_lm_debug_linear: the engine does not split this GEMM over K"));
            a.epi = MMI_EPI_PARTIAL; a.partial = partial; a.out = nullptr;
            pd.P = p.ksplit;
            if (a8) { pd.sx = sx; pd.scb = g.scb; }
        }
        rc = launch_gemm(lm, s, g, a, false);
        if (rc) return done(rc);
        if (path == MMI_DBG_SPLITK) {          // the fold of the next norm launch: x (= 0) + bf16(sum of the partials)
            const int nthN = [&] { int n = mmi_cdiv(N / 8, 64) * 64; return n > 1024 ? 1024 : n; }();
            if (N > 8 * 1024 * MMI_NORM_MAXP) return done(mmi_fail(MMI_ERR_UNSUPPORTED, "row too long for the norm kernel"));
            uint16_t* ytmp = nullptr;
            if (A.alloc(&ytmp, oelems) != hipSuccess) return done(mmi_fail(MMI_ERR_HIP, "out of device memory"));
            MMI_LAUNCH(k_resid_rmsnorm, rows, nthN, 0, s, xres, (const float*)partial, pd.P, rows, (const uint16_t*)zeros, ytmp, N, T, kout, 1e-8f,
                       (uint8_t*)nullptr, (float*)nullptr, pd.sx, pd.scb);
            MMI_LAUNCH(k_unpack_rows, mmi_cdiv(rows * N, 256), 256, 0, s, (const uint16_t*)xres, rows, N, outp, T, kout);
            MMI_DBG_LAUNCHED();
        }
    } else if (path == MMI_DBG_FUSED) {
        const int kmax = q8_fused_kmax(g);
        if (!a8 || !kmax) return done(mmi_fail(MMI_ERR_UNSUPPORTED, "mmi_lm_debug_linear: k_gemm_q8 needs an int8 x int8 model and rows of <= 88 entries"));
        a.xp = reinterpret_cast<const u32x4*>(xp);
        a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT; a.wscale = g.scale; a.wscb = g.scb; a.gate_rows = gated ? g.N : 0;
        a.wq = 3; a.xinv = g.xinv;
        a.osplit = q8_fused_osplit(g, a.epi, T);
        rc = launch_q8_fused(s, T, mt, kmax, g.NT * a.osplit, a);
        if (rc) return done(rc);
    } else if (path == MMI_DBG_NORM_FUSED) {
        const int wq = (a8 && g.wq == 1) ? 3 : g.wq;
        if (g.KSTEPS > (wq ? 32 : 64)) return done(mmi_fail(MMI_ERR_UNSUPPORTED, "mmi_lm_debug_linear: the row is too long for the norm-fused GEMM"));
        a.xp = reinterpret_cast<const u32x4*>(xp);
        a.alpha = alpha; a.D = K; a.eps = 1e-8f;
        a.wp = g.wp; a.N = g.N; a.KSTEPS = g.KSTEPS; a.NT = g.NT; a.wscale = g.scale; a.wscb = g.scb; a.gate_rows = gated ? g.N : 0;
        a.wq = wq; a.xinv = g.xinv; a.osplit = 1;
        rc = launch_norm_fused(s, T, mt, wq, g.NT, a);
        if (rc) return done(rc);
    } else {
        return done(mmi_fail(MMI_ERR_INVALID, "mmi_lm_debug_linear: unknown path"));
    }
    if (codes) MMI_LAUNCH(k_unpack_q8, mmi_cdiv(rows * K, 256), 256, 0, s, (const uint8_t*)xq, rows, K, codes, T, kin);
    if (absmax) MMI_DBG_HIP(hipMemcpyAsync(absmax, sx, (size_t)rows * sizeof(float), hipMemcpyDeviceToDevice, s));
    MMI_DBG_LAUNCHED();
    (void)KS;
    if (hipStreamSynchronize(s) != hipSuccess) { A.release(); return mmi_fail(MMI_ERR_HIP, "mmi_lm_debug_linear: the launches failed"); }
    A.release();
    return MMI_OK;
}

extern "C" int64_t mmi_lm_stat(const mmi_lm* lm, int32_t which) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return -1;
    if (which == 0) return (int64_t)lm->xlds_launches;
    if (which == 1) {       // bit v set: step program v (attn_variant) is captured and instantiated (either graph form)
        int64_t m = 0;
        for (int v = 0; v < MmiProgram::NV; ++v)
            if (lm->prog.ready(v, false) || lm->prog.ready(v, true)) m |= 1ll << v;
        return m;
    }
    if (which == 2) return (int64_t)lm->depth_bound;
    if (which == 3) return (int64_t)lm->Td;               // the depth transformer's MFMA tile (16 / 32)
    return -1;
}

extern "C" int mmi_lm_profile_begin(mmi_lm* lm) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    lm->profiling = true;
    lm->ev_used = 0;
    lm->site_ev_used = 0;
    lm->prog.tap = [lm](size_t i, bool begin, hipStream_t s) {       // one event per op boundary of the un-graphed steps
        if (!begin && i + 1 != lm->prog.ops.size()) return;
        if (lm->site_ev_used == lm->site_ev.size()) {
            mmi_lm::SiteEv e{0, nullptr};
            if (hipEventCreate(&e.ev) != hipSuccess) return;
            lm->site_ev.push_back(e);
        }
        lm->site_ev[lm->site_ev_used].op = begin ? i : (size_t)-1;
        hipEventRecord(lm->site_ev[lm->site_ev_used].ev, s);
        lm->site_ev_used += 1;
    };
    return MMI_OK;
}

// Per-site timings of the steps run since mmi_lm_profile_begin: one line "site<TAB>ops<TAB>total ms<TAB>weight bytes per op" per
// site of the launch list (one hipEvent per op boundary of the un-graphed steps: an op's time runs to the next op's event, so the
// eager dispatch gap and the event's own marker - a few us - are included).  Call before
// mmi_lm_profile_end; synchronises the stream.  Returns the bytes needed including the final NUL.
extern "C" int64_t mmi_lm_profile_sites(mmi_lm* lm, char* buf, int64_t cap) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm || !lm->profiling) return 0;
    if (lm->prof_stream) hipStreamSynchronize(lm->prof_stream);
    else hipDeviceSynchronize();
    std::vector<std::string> order;
    std::vector<double> tot;
    std::vector<long> cnt, bytes;
    for (size_t k = 0; k + 1 < lm->site_ev_used; ++k) {
        const auto& e = lm->site_ev[k];
        if (e.op == (size_t)-1) continue;            // the end of a step: the distance to the next step's first event is host time
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.ev, lm->site_ev[k + 1].ev) != hipSuccess) continue;
        const std::string& site = lm->prog.sites[e.op];
        size_t j = 0;
        while (j < order.size() && order[j] != site) ++j;
        if (j == order.size()) { order.push_back(site); tot.push_back(0.0); cnt.push_back(0); bytes.push_back(0); }
        tot[j] += ms; cnt[j] += 1;
        if (lm->prog.op_bytes[e.op] > bytes[j]) bytes[j] = lm->prog.op_bytes[e.op];
    }
    (void)hipGetLastError();
    std::string all;
    for (size_t j = 0; j < order.size(); ++j)
        all += order[j] + "\t" + std::to_string(cnt[j]) + "\t" + std::to_string(tot[j]) + "\t" + std::to_string(bytes[j]) + "\n";
    const int64_t need = (int64_t)all.size() + 1;
    if (buf && cap > 0) {
        const int64_t n = need <= cap ? need - 1 : cap - 1;
        memcpy(buf, all.data(), (size_t)n);
        buf[n] = 0;
    }
    return need;
}

extern "C" int mmi_lm_profile_end(mmi_lm* lm, double* mean_ms, int64_t* n_launches, int64_t* bytes_per_launch,
                                  const char** kernel_name) {
    MmiDeviceGuard dev_guard_(lm ? lm->device : -1);
    if (!lm) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (!lm->profiling) return mmi_fail(MMI_ERR_STATE, "profiling was not started");
    lm->profiling = false;
    lm->prog.tap = nullptr;
    if (lm->prof_stream || lm->ev_used) MMI_HIP_CHECK(hipStreamSynchronize(lm->prof_stream));
    double tot = 0.0;
    for (size_t i = 0; i < lm->ev_used; ++i) {
        float ms = 0.f;
        MMI_HIP_CHECK(hipEventElapsedTime(&ms, lm->ev_pool[i].a, lm->ev_pool[i].b));
        tot += ms;
    }
    if (mean_ms) *mean_ms = lm->ev_used ? tot / (double)lm->ev_used : 0.0;
    if (n_launches) *n_launches = (int64_t)lm->ev_used;
    if (bytes_per_launch) {
        const mmi_lm_cfg& c = lm->cfg;
        // algorithmic bytes of one FFN linear_in launch: packed weights + activations in + gated activations out
        const int64_t wbytes = lm->q8 >= 1 ? (int64_t)2 * c.ffn_hidden * c.dim + (int64_t)2 * c.ffn_hidden * 4 : (int64_t)2 * c.ffn_hidden * c.dim * 2;
        *bytes_per_launch = wbytes + (int64_t)lm->batch * c.dim * 2 + (int64_t)lm->batch * c.ffn_hidden * 2;
    }
    if (kernel_name)
        *kernel_name = lm->act8      ? "k_gemm_xp<32, MT, 1, 8, 2, 3> (temporal FFN linear_in, int8 weights x int8 activations on v_mfma_i32_32x32x32_i8 + SiLU gate)"
                       : lm->q8 == 1 ? "k_gemm_xp<32, 1, 1, 8, 2, 1> (temporal FFN linear_in, int8 weights + SiLU gate)"
                       : lm->q8 == 2 ? "k_gemm_xp<32, 1, 1, 8, 2, 2> (temporal FFN linear_in, fp8 weights on the fp8 MFMA + SiLU gate)"
                       : lm->dominant_xlds ? (lm->batch > 32 ? "k_gemm_xlds<2, 32, 3, true, 0> (temporal FFN linear_in + SiLU gate)"
                                                             : "k_gemm_xlds<1, 64, 3, true, 0> (temporal FFN linear_in + SiLU gate)")
                                     : "k_gemm_xp<32, 1, 1, 8, 2> (temporal FFN linear_in + SiLU gate)";
    lm->ev_used = 0;
    return MMI_OK;
}
