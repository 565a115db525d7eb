// Does the ORDER of MFMA issue change the power-limited rate?  Same loop as mfma_power.hip (random sign / exponent / mantissa
// operands resident in registers, four accumulator chains per wave), but the operand registers are rotated at different rates:
//   mode 0: A and B change with every MFMA                      (no operand shared between neighbours)
//   mode 1: B fixed over 4 consecutive MFMAs, A changes          (the halo kernel: one weight fragment x four row fragments)
//   mode 2: B fixed over 12 consecutive MFMAs, A changes
//   mode 3: A and B fixed over 4 consecutive MFMAs               (only the accumulator changes)
//   mode 4: A and B fixed for the whole loop
//   mode 5: as mode 1 with split-bf16 operands (hi / mid / lo terms of N(0,1) floats) in the kernel's product order
//   mode 6: as mode 5, products ordered by weight term (B changes 3 times per 24 MFMAs instead of 6)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_order.hip -o scripts/ubench/mfma_order && scripts/ubench/mfma_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters) {
    bf16x8 a[4][3], b[3];                                     // four row fragments x three terms, one weight fragment x three terms
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[i][q] = __builtin_bit_cast(bf16x8, src[((threadIdx.x * 5 + i) * 3 + q) & 4095 | (MODE >= 5 ? 0 : 0)]);
#pragma unroll
    for (int q = 0; q < 3; ++q) b[q] = __builtin_bit_cast(bf16x8, src[((threadIdx.x * 5 + 4) * 3 + q) & 4095]);
    f32x16 acc[4], lo[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = lo[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i], a[(i + u) & 3][(i + u) % 3], b[(i + 2 * u + (u >> 1)) % 3]);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i], a[i][(u + i) % 3], b[u % 3]);
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i], a[i][(u + i) % 3], b[u / 3]);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i], a[u & 3][u % 3], b[u % 3]);
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) MFMA(acc[i], a[0][0], b[0]);
        } else if (MODE == 5) {                                // (hi,hi) (hi,mid) (mid,hi) | (mid,mid) (hi,lo) (lo,hi) -> lo accumulator
            constexpr int QA[6] = {0, 0, 1, 1, 0, 2}, QB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (u == 0) MFMA(acc[i], a[i][QA[u]], b[QB[u]]);
                    else MFMA(lo[i], a[i][QA[u]], b[QB[u]]);
                }
        } else {                                               // by weight term: b0 x (hi, mid, lo), b1 x (hi, mid), b2 x hi
            constexpr int QA[6] = {0, 1, 2, 0, 1, 0}, QB[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (u == 0) MFMA(acc[i], a[i][QA[u]], b[QB[u]]);
                    else MFMA(lo[i], a[i][QA[u]], b[QB[u]]);
                }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r] + lo[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, const uint4* d, int blocks, int iters) {
    float* o;
    hipMalloc(&o, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, o, 200);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, o, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;                      // rep 0 warms the clocks down to the sustained state
    }
    const double flops = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
    const double tf = flops / best / 1e9;
    printf("%-72s %8.2f ms  %7.1f TFLOP/s  = %.3f of 2500\n", name, best, tf, tf / 2500.0);
    hipFree(o);
}

static unsigned short bf16_rn(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    static uint4 h[4096];
    uint4* d;
    hipMalloc(&d, sizeof(h));
    srand(1);
    for (int i = 0; i < 4096; ++i) {
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            unsigned lo = ((rand() & 1) << 15) | ((119 + rand() % 17) << 7) | (rand() & 0x7f);
            unsigned hi = ((rand() & 1) << 15) | ((119 + rand() % 17) << 7) | (rand() & 0x7f);
            w[j] = (hi << 16) | lo;
        }
        h[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int blocks = 512, iters = 150000;
    run<0>("random bits: A and B change every MFMA", d, blocks, iters);
    run<1>("random bits: B fixed over 4 MFMAs (one weight fragment x 4 row fragments)", d, blocks, iters);
    run<2>("random bits: B fixed over 12 MFMAs", d, blocks, iters);
    run<3>("random bits: A and B fixed over 4 MFMAs", d, blocks, iters);
    run<4>("random bits: A and B fixed for the loop", d, blocks, iters);
    // split-bf16 operands: slot (x*3 + q) holds term q of eight N(0,1)-ish floats
    for (int x = 0; x < 4096 / 3; ++x) {
        unsigned short t[3][8];
        for (int e = 0; e < 8; ++e) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            for (int q = 0; q < 3; ++q) { t[q][e] = bf16_rn(v); v -= bf16_f(t[q][e]); }
        }
        for (int q = 0; q < 3; ++q) {
            unsigned w[4];
            for (int j = 0; j < 4; ++j) w[j] = t[q][2 * j] | ((unsigned)t[q][2 * j + 1] << 16);
            h[x * 3 + q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    run<5>("split terms of N(0,1): kernel order (hh | hm mh mm hl lh)", d, blocks, iters);
    run<6>("split terms of N(0,1): by weight term (b0: h m l | b1: h m | b2: h)", d, blocks, iters);
    run<1>("split terms of N(0,1): mixed terms, B fixed over 4", d, blocks, iters);
    return 0;
}
