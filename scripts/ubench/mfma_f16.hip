// v_mfma_f32_32x32x16_f16 as the multiplier of a two-term fp16 split (r06): three questions.
//  (1) POWER: sustained rate on the operands the split2h kernels feed it (hi / lo fp16 terms of 2^14-scaled N(0,1) floats,
//      products a1 b1, a1 b2, a2 b1) beside the bf16 terms of the six-product scheme -- is the DVFS ceiling the same?
//  (2) ROUNDING: how exact is a chain of K/16 MFMAs on 22-bit products (an fp16 x fp16 product has 22 significant bits, a bf16
//      one 16)?  Same-sign and mixed-sign data, error against fp64 in u = 2^-24 relative to sum |a||b|.
//  (3) SUBNORMALS: are fp16 subnormal inputs multiplied (not flushed)?  The second split term of a small element is one.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f16.hip -o scripts/ubench/mfma_f16 && scripts/ubench/mfma_f16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------- (1) power
template <bool F16>
__global__ __launch_bounds__(256) void kpow(const uint4* __restrict__ src, float* out, int iters) {
    // a[0], b[0]: hi terms; a[1], b[1]: lo terms (of different elements per lane -- only the bit activity matters)
    uint4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = src[(threadIdx.x * 4 + i) & 4095];
        b[i] = src[(threadIdx.x * 4 + 2 + i) & 4095];
    }
    f32x16 hi[2], lo[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) hi[i][r] = lo[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (F16) {
                    lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, b[0]), lo[i], 0, 0, 0);
                    lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[1]), lo[i], 0, 0, 0);
                    hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, b[0]), hi[i], 0, 0, 0);
                } else {
                    lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[0]), lo[i], 0, 0, 0);
                    lo[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[1]), lo[i], 0, 0, 0);
                    hi[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), hi[i], 0, 0, 0);
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += hi[i][r] + lo[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static unsigned short f2bf_trunc(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float gauss() {
    const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <bool F16>
static void run_power(const char* name, const uint4* d, int waves) {
    const int blocks = 256 * waves, iters = 30000 / waves;
    float* o;
    hipMalloc(&o, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kpow<F16><<<blocks, 256>>>(d, o, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kpow<F16><<<blocks, 256>>>(d, o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
    const double tf = flops / ms / 1e9;
    printf("POWER %-44s %d wave/SIMD %8.2f ms %7.1f TFLOP/s = %.3f of 2500 (eff. clock %.2f GHz)\n", name, waves, ms, tf, tf / 2500.0,
           tf / 2500.0 * 2.4);
    hipFree(o);
}

// ---------------------------------------------------------------- (2) rounding: C[32][32] = sum_k A[32][K] B[32][K] as K/16 chained MFMAs
template <bool F16>
__global__ void kdot(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C, int K) {
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const uint4 a = *reinterpret_cast<const uint4*>(A + (size_t)row * K + k0 + half * 8);
        const uint4 b = *reinterpret_cast<const uint4*>(B + (size_t)row * K + k0 + half * 8);
        if (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    // D layout of the 32x32 MFMAs: register r of lane l holds row 8*(r/4) + 4*(l/32) + r%4, column l%32
    for (int r = 0; r < 16; ++r) C[(8 * (r / 4) + 4 * half + (r & 3)) * 32 + row] = acc[r];
}

template <bool F16>
static void run_round(const char* name, int K, int mode) {
    std::vector<unsigned short> A(32 * K), B(32 * K);
    srand(7);
    for (int i = 0; i < 32 * K; ++i) {
        float a, b;
        if (mode == 0) { a = 1.f + rand() / (float)RAND_MAX; b = 1.f + rand() / (float)RAND_MAX; }        // same sign, [1, 2)
        else if (mode == 1) { a = gauss(); b = gauss(); }                                                     // N(0,1)
        else { a = fabsf(gauss()) * 1024.f; b = fabsf(gauss()) * 1024.f; }                                    // same sign, |N| * 2^10
        A[i] = F16 ? f2h(a) : f2bf_trunc(a);
        B[i] = F16 ? f2h(b) : f2bf_trunc(b);
    }
    unsigned short *dA, *dB;
    float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    kdot<F16><<<1, 64>>>(dA, dB, dC, K);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double rms = 0, mx = 0, mean = 0;
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            double ex = 0, ab = 0;
            for (int k = 0; k < K; ++k) {
                const double a = F16 ? h2f(A[m * K + k]) : bf2f(A[m * K + k]), b = F16 ? h2f(B[n * K + k]) : bf2f(B[n * K + k]);
                ex += a * b;
                ab += fabs(a * b);
            }
            const double e = (C[m * 32 + n] - ex) / ab * 16777216.0;      // in u = 2^-24 of sum |a||b|
            rms += e * e; mean += e;
            if (fabs(e) > mx) mx = fabs(e);
        }
    printf("ROUND %-8s K=%5d %-22s err vs fp64 / sum|a||b|: rms %7.2f u  max %7.2f u  mean %+7.2f u\n", F16 ? "f16" : "bf16", K, name,
           sqrt(rms / 1024), mx, mean / 1024);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

// ---------------------------------------------------------------- (3) subnormal inputs, conversion
__global__ void ksub(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    // k = 0 of lane half 0: a = 2^-20 (fp16 subnormal: 0x0010), b = 2^10
    if (lane < 32) { a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010); b[0] = (_Float16)1024.f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = acc[0];                                  // expect 2^-10 = 9.765625e-4
        const float r = 3.0e-6f;                           // below the fp16 normal range (2^-14 = 6.1e-5)
        const _Float16 h = (_Float16)r;                   // expect the nearest subnormal, not 0
        out[1] = (float)h;
        out[2] = __builtin_fmaf(r, 1.0f, -(float)h);       // the residual the second term would be made of
    }
}

int main() {
    // ---- (1)
    std::vector<uint4> h(4096);
    uint4* d;
    hipMalloc(&d, 4096 * 16);
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        // per lane slot i&3: 0 = A hi, 1 = A lo, 2 = B hi, 3 = B lo (kpow reads src[t*4 + i])
        for (int i = 0; i < 4096; ++i) {
            unsigned short w[8];
            for (int j = 0; j < 8; ++j) {
                const float x = gauss();
                if (mode == 0) {                   // fp16 two-term split of 2^14-ish scaled data (amax ~ 4.5 sigma -> scale 2^12)
                    const float sx = x * 4096.f;
                    const unsigned short x1 = f2h(sx);
                    const unsigned short x2 = f2h(sx - h2f(x1));
                    w[j] = (i & 1) ? x2 : x1;
                } else if (mode == 1) {            // bf16 two-term round-to-nearest split (r05 split2)
                    __bf16 b1 = (__bf16)x; float f1 = (float)b1; __bf16 b2 = (__bf16)(x - f1);
                    unsigned short u1, u2; memcpy(&u1, &b1, 2); memcpy(&u2, &b2, 2);
                    w[j] = (i & 1) ? u2 : u1;
                } else {                           // bf16 truncation split, first two of three terms (the six-product scheme)
                    const unsigned short x1 = f2bf_trunc(x);
                    const unsigned short x2 = f2bf_trunc(x - bf2f(x1));
                    w[j] = (i & 1) ? x2 : x1;
                }
            }
            memcpy(&h[i], w, 16);
        }
        hipMemcpy(d, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        for (int waves = 1; waves <= 2; ++waves) {
            if (mode == 0) run_power<true>("f16 two-term split of N(0,1)*2^12 (split2h)", d, waves);
            if (mode == 1) run_power<false>("bf16 two-term rn split of N(0,1) (r05 split2)", d, waves);
            if (mode == 2) run_power<false>("bf16 truncation split, terms 1-2 (split3)", d, waves);
        }
    }
    // ---- (2)
    const char* names[3] = {"same sign [1,2)", "N(0,1)", "same sign |N|*2^10"};
    for (int mode = 0; mode < 3; ++mode)
        for (int K : {16, 512, 4608}) {
            run_round<true>(names[mode], K, mode);
            run_round<false>(names[mode], K, mode);
        }
    // ---- (3)
    float* o;
    hipMalloc(&o, 16);
    ksub<<<1, 64>>>(o);
    float r[3];
    hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
    printf("SUBNORMAL mfma(2^-20 [fp16 subnormal] * 2^10) = %.9g (expect 9.765625e-04); (f16)3.0e-6 = %.9g (expect ~2.98e-06, not 0); residual %.3g\n",
           r[0], r[1], r[2]);
    return 0;
}
