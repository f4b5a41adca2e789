// Hardware semantics of the gfx950 fp8 path used by the C5 GEMMs, checked against the software definition the oracle and the
// kernel simulator use (OCP e4m3fn, round-to-nearest-even, clamp to +-448):
//   1. v_cvt_pk_fp8_f32 on EVERY finite bf16 value (after the kernels' med3 clamp) and on exact ties
//   2. v_mfma_f32_32x32x16_fp8_fp8 / v_mfma_f32_16x16x32_fp8_fp8 on random fp8 codes incl. subnormals: exact products, fp32 sums
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -I moshi_amd/csrc scripts/fp8_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include "mmi_device.h"
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

static uint8_t sw_e4m3(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    float a = fabsf(x);
    if (a > 448.f) a = 448.f;
    if (a < 0.015625f) return (uint8_t)(sign | (int)nearbyintf(a * 512.f));
    int e; const float m = frexpf(a, &e);
    int q = (int)nearbyintf((m * 2.f - 1.f) * 8.f), E = e - 1 + 7;
    if (q == 8) { q = 0; E += 1; }
    return (uint8_t)(sign | (E << 3) | q);
}
static float sw_dec(uint8_t b) {
    const int E = (b >> 3) & 15, m = b & 7;
    float v = E == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + (float)m / 8.f, E - 7);
    return (b & 0x80) ? -v : v;
}

__global__ void k_cvt(const float* x, uint32_t* y, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mmi_cvt_fp8x4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}
// one wave: A 32x16, B 16x32 as fp8 codes in the lane layout of the GEMM kernels
__global__ void k_mfma32(const uint8_t* A, const uint8_t* B, float* D) {
    const int l = threadIdx.x;
    u32x2 a, b;
    uint8_t ab[8], bb[8];
    for (int e = 0; e < 8; ++e) { ab[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e]; bb[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)]; }
    memcpy(&a, ab, 8); memcpy(&b, bb, 8);
    f32x16 c = {0};
    c = mmi_mfma_fp8_32x32x16(a, b, c);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_mfma16(const uint8_t* A, const uint8_t* B, float* D) {
    const int l = threadIdx.x;
    u32x2 a, b;
    uint8_t ab[8], bb[8];
    for (int e = 0; e < 8; ++e) { ab[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e]; bb[e] = B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
    memcpy(&a, ab, 8); memcpy(&b, bb, 8);
    f32x4 c = {0};
    c = mmi_mfma_fp8_16x16x32(a, b, c);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

int main() {
    // ---- 1. conversion
    std::vector<float> xs;
    for (uint32_t h = 0; h < 65536; ++h) {
        uint32_t u = h << 16; float f; memcpy(&f, &u, 4);
        if (std::isfinite(f)) xs.push_back(f);
    }
    const float ties[] = {0.0009765625f, 0.0029296875f, 0.017578125f, 18.f, 22.f, 416.f, 448.f, 464.f, 1e4f, -1e4f, 0.f, -0.f};
    for (float t : ties) xs.push_back(t);
    while (xs.size() % 4) xs.push_back(0.f);
    const int n4 = (int)xs.size() / 4;
    float* dx; uint32_t* dy;
    hipMalloc(&dx, xs.size() * 4); hipMalloc(&dy, n4 * 4);
    hipMemcpy(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, (n4 + 255) / 256, 256, 0, 0, dx, dy, n4);
    std::vector<uint32_t> ys(n4);
    hipMemcpy(ys.data(), dy, n4 * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < xs.size(); ++i) {
        const uint8_t hw = (uint8_t)(ys[i / 4] >> (8 * (i % 4))), sw = sw_e4m3(xs[i]);
        if (hw != sw && !((hw & 0x7f) == 0 && (sw & 0x7f) == 0)) {
            if (bad < 10) printf("cvt mismatch x=%.9g hw=0x%02x sw=0x%02x\n", xs[i], hw, sw);
            ++bad;
        }
    }
    printf("cvt: %zu values, %ld mismatches\n", xs.size(), bad);
    // ---- 2. MFMA
    uint8_t A[512], B[512];
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int i = 0; i < 512; ++i) {
        uint8_t a = (uint8_t)rnd(), b = (uint8_t)rnd();
        if ((a & 0x7f) == 0x7f) a ^= 1;    // no NaN codes
        if ((b & 0x7f) == 0x7f) b ^= 1;
        if (i % 5 == 0) a &= 0x87;         // subnormals
        if (i % 7 == 0) b &= 0x87;
        A[i] = a; B[i] = b;
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024 * 4);
    hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
    float D[1024];
    hipLaunchKernelGGL(k_mfma32, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, 1024 * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 16; ++k) { double p = (double)sw_dec(A[i * 16 + k]) * sw_dec(B[k * 32 + j]); ref += p; mag += fabs(p); }
        worst = fmax(worst, fabs(D[i * 32 + j] - ref) / (mag + 1e-30));
    }
    printf("mfma 32x32x16 fp8: worst |err| / sum|products| = %.3g\n", worst);
    hipLaunchKernelGGL(k_mfma16, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, 256 * 4, hipMemcpyDeviceToHost);
    worst = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double ref = 0, mag = 0;
        for (int k = 0; k < 32; ++k) { double p = (double)sw_dec(A[i * 32 + k]) * sw_dec(B[k * 16 + j]); ref += p; mag += fabs(p); }
        worst = fmax(worst, fabs(D[i * 16 + j] - ref) / (mag + 1e-30));
    }
    printf("mfma 16x16x32 fp8: worst |err| / sum|products| = %.3g\n", worst);
    // ---- 3. are subnormal inputs honoured?  A = all subnormal codes (0x01..0x07), B = 1.0
    for (int i = 0; i < 512; ++i) { A[i] = (uint8_t)(1 + i % 7); B[i] = 0x38; }
    hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma32, 1, 64, 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, 1024 * 4, hipMemcpyDeviceToHost);
    { double ref = 0; for (int k = 0; k < 16; ++k) ref += sw_dec(A[k]);
      printf("subnormal A x 1.0: got %.9g expected %.9g (0 would mean flush-to-zero)\n", D[0], ref); }
    // ---- 4. one large product + 15 small ones: how many bits below the largest product survive?
    for (int sh = 4; sh <= 20; sh += 2) {
        // A row 0: 448 (0x7e) at k=0, then 1.0 (0x38); B col 0: 1.0 at k=0, then 2^-sh/... use B = small value code
        for (int i = 0; i < 512; ++i) { A[i] = 0x38; B[i] = 0x38; }
        A[0] = 0x7e;                                   // 448
        // small = 448 * 2^-sh  -> choose A[k]=x, B[k]=y with x*y = small: x = 2^-a, y = 2^-b
        const int e_small = 8 - sh;                    // 448*2^-sh ~ 1.75 * 2^(8-sh)
        int ea = e_small / 2, eb = e_small - ea;       // exponents of the two factors (1.0 * 2^e)
        auto code = [](int e) { return (uint8_t)(((e + 7) << 3)); };
        bool ok = ea + 7 >= 1 && eb + 7 >= 1 && ea + 7 <= 15 && eb + 7 <= 15;
        if (!ok) continue;
        for (int k = 1; k < 16; ++k) { A[k] = code(ea); B[k * 32] = code(eb); }
        B[0] = 0x38;
        hipMemcpy(dA, A, 512, hipMemcpyHostToDevice); hipMemcpy(dB, B, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma32, 1, 64, 0, 0, dA, dB, dD);
        hipMemcpy(D, dD, 1024 * 4, hipMemcpyDeviceToHost);
        const double small = ldexp(1.0, ea + eb), ref = 448.0 + 15 * small;
        printf("448 + 15 x 2^%d: got %.9g expected %.9g (small part kept: %.3f)\n", ea + eb, D[0], ref, (D[0] - 448.0) / (15 * small));
    }
    return 0;
}
