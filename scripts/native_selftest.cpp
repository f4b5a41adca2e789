// A Python-free self-test of an installed libmoshi_mi.so on a GPU box: a tiny Mimi and a tiny Moshi LM are built through the C ABI
// (include/moshi_mi.h) from weights this program generates itself (an integer hash, the same one
// tests/golden/make_native_selftest.py uses), run for a few frames, and compared with what the numpy oracle computed for them
// (tests/golden/native_selftest/expected.bin): RVQ codes and teacher-forced token-ring outputs bit-exact, PCM within 2e-5, logits
// within the bf16 tolerance of tests/lm_cases.py (5 % max / 1.2 % mean of max|logit| per row and site).  ~1 s, no torch.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -Iinclude scripts/native_selftest.cpp -Lmoshi_amd -lmoshi_mi \
//         -Wl,-rpath,'$ORIGIN/../moshi_amd' -o build/native_selftest && build/native_selftest tests/golden/native_selftest
//
// The same source builds against the CPU simulator (-DMMI_SELFTEST_SIM, tests/test_native_selftest.py), which is how the expected
// values and this program are checked where there is no GPU.  The checker side (oracle) never runs here: only its recorded output.
#include "moshi_mi.h"
#ifdef MMI_SELFTEST_SIM
#include "hipsim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define HCK(x) do { if ((x) != hipSuccess) { printf("HIP call failed at line %d\n", __LINE__); exit(2); } } while (0)
#define MCK(x) do { int rc_ = (x); if (rc_) { printf("%s -> %d: %s\n", #x, rc_, mmi_last_error()); exit(3); } } while (0)

static float u_of(uint32_t seed, uint32_t i) {
    uint32_t h = (i * 2654435761u) ^ seed;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (float)((int)(h & 0xffffu) - 32768) * (1.0f / 32768.0f);
}
static uint16_t bf16_rne(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    x += 0x7fffu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

struct Tensor { std::string model, name; int bf16, ndim; long d[4]; float base, scale; uint32_t seed; void* dev = nullptr; };
struct Expected { std::string name; int i64; long count, off; };

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/native_selftest";
    FILE* mf = fopen((dir + "/manifest.txt").c_str(), "r");
    if (!mf) { printf("cannot open %s/manifest.txt\n", dir.c_str()); return 2; }
    mmi_mimi_cfg mc; mmi_lm_cfg lc;
    memset(&mc, 0, sizeof(mc)); memset(&lc, 0, sizeof(lc));
    std::vector<Tensor> tensors; std::vector<Expected> exps;
    int B = 0, FR = 0, K = 0, ST = 0, fs = 0, n_user = 0, dep_q = 0, card = 0, text_card = 0;
    char tag[64];
    while (fscanf(mf, "%63s", tag) == 1) {
        if (!strcmp(tag, "mimi_cfg")) {
            int32_t* p = (int32_t*)&mc;            // header order: 6 ints, ratios[8], 10 ints, 1 float, 4 ints
            for (int i = 0; i < 6 + 8 + 10; ++i) if (fscanf(mf, "%d", p + i) != 1) return 2;
            if (fscanf(mf, "%f", &mc.tr_max_period) != 1) return 2;
            if (fscanf(mf, "%d %d %d %d", &mc.q_dimension, &mc.q_bins, &mc.q_n_q, &mc.q_n_q_semantic) != 4) return 2;
        } else if (!strcmp(tag, "lm_cfg")) {
            if (fscanf(mf, "%d %d %d %d %d %f", &lc.dim, &lc.num_heads, &lc.num_layers, &lc.ffn_hidden, &lc.context, &lc.max_period) != 6) return 2;
            if (fscanf(mf, "%d %d %d %d %d %d %d %d %d", &lc.n_q, &lc.dep_q, &lc.card, &lc.text_card, &lc.text_card_out, &lc.depformer_dim,
                       &lc.depformer_num_heads, &lc.depformer_num_layers, &lc.depformer_ffn_hidden) != 9) return 2;
            for (int i = 0; i < 64; ++i) if (fscanf(mf, "%d", &lc.delays[i]) != 1) return 2;
            if (fscanf(mf, "%d %d %d %d %d", &lc.existing_text_padding_id, &lc.extra_heads_num_heads, &lc.extra_heads_dim, &lc.kv_cache_dtype,
                       &lc.cross_attention) != 5) return 2;
        } else if (!strcmp(tag, "T")) {
            Tensor t; char model[16], name[256], dt[8], b[64], s[64];
            if (fscanf(mf, "%15s %255s %7s %d %ld %ld %ld %ld %63s %63s %u", model, name, dt, &t.ndim, &t.d[0], &t.d[1], &t.d[2], &t.d[3], b, s, &t.seed) != 11) return 2;
            t.model = model; t.name = name; t.bf16 = !strcmp(dt, "bf16"); t.base = strtof(b, nullptr); t.scale = strtof(s, nullptr);
            tensors.push_back(t);
        } else if (!strcmp(tag, "run")) {
            if (fscanf(mf, "%d %d %d %d %d %d %d %d %d", &B, &FR, &K, &ST, &fs, &n_user, &dep_q, &card, &text_card) != 9) return 2;
        } else if (!strcmp(tag, "E")) {
            Expected e; char name[64], dt[8];
            if (fscanf(mf, "%63s %7s %ld %ld", name, dt, &e.count, &e.off) != 4) return 2;
            e.name = name; e.i64 = !strcmp(dt, "i64");
            exps.push_back(e);
        } else { printf("manifest: unknown line '%s'\n", tag); return 2; }
    }
    fclose(mf);
    std::vector<unsigned char> blob;
    {
        FILE* bf = fopen((dir + "/expected.bin").c_str(), "rb");
        if (!bf) { printf("cannot open expected.bin\n"); return 2; }
        fseek(bf, 0, SEEK_END); long n = ftell(bf); fseek(bf, 0, SEEK_SET);
        blob.resize(n);
        if (fread(blob.data(), 1, n, bf) != (size_t)n) return 2;
        fclose(bf);
    }
    auto expect = [&](const char* name) -> const void* {
        for (auto& e : exps) if (e.name == name) return blob.data() + e.off;
        printf("expected.bin has no %s\n", name); exit(2);
    };
    printf("libmoshi_mi ABI version %d; %zu tensors; B=%d frames=%d K=%d steps=%d\n", mmi_version(), tensors.size(), B, FR, K, ST);

    // ---- weights: generated here, uploaded, described by the reference's state-dict names
    std::vector<mmi_tensor_desc> md, ld;
    for (auto& t : tensors) {
        long n = 1; for (int i = 0; i < t.ndim; ++i) n *= t.d[i];
        std::vector<float> v(n);
        for (long i = 0; i < n; ++i) { const float x = t.scale * u_of(t.seed, (uint32_t)i); v[i] = t.base + x; }
        const size_t bytes = (size_t)n * (t.bf16 ? 2 : 4);
        HCK(hipMalloc(&t.dev, bytes));
        if (t.bf16) {
            std::vector<uint16_t> h(n);
            for (long i = 0; i < n; ++i) h[i] = bf16_rne(v[i]);
            HCK(hipMemcpy(t.dev, h.data(), bytes, hipMemcpyHostToDevice));
        } else HCK(hipMemcpy(t.dev, v.data(), bytes, hipMemcpyHostToDevice));
        mmi_tensor_desc d; memset(&d, 0, sizeof(d));
        d.name = t.name.c_str(); d.data = t.dev; d.dtype = t.bf16 ? MMI_BF16 : MMI_F32; d.ndim = t.ndim;
        for (int i = 0; i < 4; ++i) d.shape[i] = i < t.ndim ? t.d[i] : 0;
        (t.model == "mimi" ? md : ld).push_back(d);
    }
    int failures = 0;

    // ---- Mimi: encode FR frames, decode the oracle's codes
    {
        mmi_mimi* m = nullptr;
        MCK(mmi_mimi_create(&mc, md.data(), (int32_t)md.size(), B, &m));
        MCK(mmi_mimi_set_num_codebooks(m, K));
        MCK(mmi_mimi_streaming_start(m, B, nullptr));
        const int64_t* ecodes = (const int64_t*)expect("mimi_codes");
        const float* epcm = (const float*)expect("mimi_pcm");
        float *dx, *dpcm; int64_t *dc, *dcin;
        HCK(hipMalloc((void**)&dx, (size_t)B * fs * 4)); HCK(hipMalloc((void**)&dpcm, (size_t)B * fs * 4));
        HCK(hipMalloc((void**)&dc, (size_t)B * K * 8)); HCK(hipMalloc((void**)&dcin, (size_t)B * K * 8));
        long code_diffs = 0; double worst_pcm = 0;
        for (int f = 0; f < FR; ++f) {
            std::vector<float> x((size_t)B * fs);
            for (size_t i = 0; i < x.size(); ++i) x[i] = 0.3f * u_of(77u + f, (uint32_t)i);
            HCK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
            MCK(mmi_mimi_encode_step(m, dx, dc, B, 1, nullptr));
            HCK(hipMemcpy(dcin, ecodes + (size_t)f * B * K, (size_t)B * K * 8, hipMemcpyHostToDevice));
            MCK(mmi_mimi_decode_step(m, dcin, dpcm, B, K, 1, nullptr));
            HCK(hipDeviceSynchronize());
            std::vector<int64_t> c((size_t)B * K); std::vector<float> p((size_t)B * fs);
            HCK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
            HCK(hipMemcpy(p.data(), dpcm, p.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < c.size(); ++i) code_diffs += c[i] != ecodes[(size_t)f * B * K + i];
            for (int b = 0; b < B; ++b) {
                double mx = 0, d = 0;
                for (int i = 0; i < fs; ++i) {
                    const float r = epcm[((size_t)f * B + b) * fs + i];
                    mx = fmax(mx, fabs(r)); d = fmax(d, fabs(p[(size_t)b * fs + i] - r));
                }
                worst_pcm = fmax(worst_pcm, d / (2e-5 + 2e-5 * mx));
            }
        }
        printf("mimi: %ld of %d code indices differ from the oracle (must be 0); worst PCM error %.3f of its tolerance (must be <= 1)\n",
               code_diffs, FR * B * K, worst_pcm);
        failures += code_diffs != 0 || !(worst_pcm <= 1.0);
        MCK(mmi_mimi_streaming_stop(m));
        mmi_mimi_destroy(m);
    }
    // ---- LM: greedy steps, teacher-forced with the oracle's tokens
    {
        mmi_lm* lm = nullptr;
        MCK(mmi_lm_create(&lc, ld.data(), (int32_t)ld.size(), B, &lm));
        mmi_sampling sp; memset(&sp, 0, sizeof(sp));
        sp.use_sampling = 0; sp.temp = 0.8f; sp.temp_text = 0.7f; sp.top_k = 250; sp.top_k_text = 25; sp.seed = 0;
        MCK(mmi_lm_streaming_start(lm, B, &sp, nullptr));
        const int NT = dep_q + 1;
        const int64_t* eforced = (const int64_t*)expect("lm_forced");
        const int64_t* eout = (const int64_t*)expect("lm_out");
        const float* etl = (const float*)expect("lm_text_logits");
        const float* eal = (const float*)expect("lm_audio_logits");
        int64_t *duc, *dforced, *dout; float *dtl, *dal;
        HCK(hipMalloc((void**)&duc, (size_t)B * n_user * 8)); HCK(hipMalloc((void**)&dforced, (size_t)B * NT * 8));
        HCK(hipMalloc((void**)&dout, (size_t)B * NT * 8));
        HCK(hipMalloc((void**)&dtl, (size_t)B * text_card * 4)); HCK(hipMalloc((void**)&dal, (size_t)B * dep_q * card * 4));
        long tok_diffs = 0; double worst_max = 0, worst_mean = 0;
        auto site = [&](const float* a, const float* r, int n) {
            double mx = 1e-6, dmax = 0, dsum = 0;
            for (int i = 0; i < n; ++i) { mx = fmax(mx, fabs(r[i])); const double d = fabs(a[i] - r[i]); dmax = fmax(dmax, d); dsum += d; }
            worst_max = fmax(worst_max, dmax / mx); worst_mean = fmax(worst_mean, dsum / n / mx);
        };
        for (int s = 0; s < ST; ++s) {
            std::vector<int64_t> uc((size_t)B * n_user);
            for (size_t i = 0; i < uc.size(); ++i) uc[i] = (int64_t)((u_of(9000u + s, (uint32_t)i) + 1.0f) * 32768.0f) % card;
            HCK(hipMemcpy(duc, uc.data(), uc.size() * 8, hipMemcpyHostToDevice));
            HCK(hipMemcpy(dforced, eforced + (size_t)s * B * NT, (size_t)B * NT * 8, hipMemcpyHostToDevice));
            MCK(mmi_lm_force_next_tokens(lm, dforced, nullptr));
            int32_t valid = 0;
            MCK(mmi_lm_step(lm, duc, n_user, dout, dtl, dal, nullptr, B, &valid, nullptr));
            HCK(hipDeviceSynchronize());
            std::vector<int64_t> o((size_t)B * NT); std::vector<float> tl((size_t)B * text_card), al((size_t)B * dep_q * card);
            HCK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
            HCK(hipMemcpy(tl.data(), dtl, tl.size() * 4, hipMemcpyDeviceToHost));
            HCK(hipMemcpy(al.data(), dal, al.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < o.size(); ++i) tok_diffs += o[i] != eout[(size_t)s * B * NT + i];
            for (int b = 0; b < B; ++b) {
                site(tl.data() + (size_t)b * text_card, etl + ((size_t)s * B + b) * text_card, text_card);
                for (int k = 0; k < dep_q; ++k) site(al.data() + ((size_t)b * dep_q + k) * card, eal + (((size_t)s * B + b) * dep_q + k) * card, card);
            }
        }
        printf("lm: %ld of %d token-ring outputs differ from the oracle (must be 0); logits worst max %.4f (<= 0.05), worst mean %.4f (<= 0.012) of max|logit|\n",
               tok_diffs, ST * B * NT, worst_max, worst_mean);
        failures += tok_diffs != 0 || !(worst_max <= 0.05) || !(worst_mean <= 0.012);
        MCK(mmi_lm_streaming_stop(lm));
        mmi_lm_destroy(lm);
    }
    printf(failures ? "SELFTEST FAILED\n" : "SELFTEST PASSED\n");
    return failures ? 1 : 0;
}
