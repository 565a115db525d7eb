// Does the MFMA SHAPE change the power-limited rate of the six-product split scheme?  
// Register-only loops on the operands the conv kernels feed the matrix pipe -- hi / mid / lo bf16 terms of N(0,1) floats, products in
// the kernels' order (hh -> hi accumulator | hm mh mm hl lh -> lo accumulator) -- with the same accumulator footprint (128 VGPRs):
//   mode 0: v_mfma_f32_32x32x16_bf16, 4 row fragments x 1 weight fragment        (what conv3_halo_split<128> issues per tap)
//   mode 1: v_mfma_f32_16x16x32_bf16, 4 row fragments x 4 weight fragments       (same accumulator registers, twice the K per issue)
//   mode 2: as mode 0 with the 24 MFMAs ordered A-major (all products of row fragment 0, then 1, ...): the A operand stays on the bus
//   mode 3: as mode 0, one wave per SIMD (256 blocks)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_shape.hip -o scripts/ubench/mfma_shape && scripts/ubench/mfma_shape
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define MFMA16(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters) {
    constexpr int QA[6] = {0, 0, 1, 1, 0, 2}, QB[6] = {0, 1, 0, 1, 2, 0};
    bf16x8 a[4][3], b[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            a[i][q] = __builtin_bit_cast(bf16x8, src[(((threadIdx.x * 9 + i) % 1365) * 3 + q)]);
            b[i][q] = __builtin_bit_cast(bf16x8, src[(((threadIdx.x * 9 + 4 + i) % 1365) * 3 + q)]);
        }
    float s = 0;
    if (MODE == 1) {
        f32x4 hi[16], lo[16];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) hi[i][r] = lo[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (u == 0) MFMA16(hi[i * 4 + j], a[i][QA[u]], b[j][QB[u]]);
                        else MFMA16(lo[i * 4 + j], a[i][QA[u]], b[j][QB[u]]);
                    }
        }
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) s += hi[i][r] + lo[i][r];
    } else {
        f32x16 hi[4], lo[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) hi[i][r] = lo[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int u = 0; u < 6; ++u) {
                        if (u == 0) MFMA32(hi[i], a[i][QA[u]], b[0][QB[u]]);
                        else MFMA32(lo[i], a[i][QA[u]], b[0][QB[u]]);
                    }
            } else {
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (u == 0) MFMA32(hi[i], a[i][QA[u]], b[0][QB[u]]);
                        else MFMA32(lo[i], a[i][QA[u]], b[0][QB[u]]);
                    }
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += hi[i][r] + lo[i][r];
    }
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
    // MACs per iteration and wave: mode 1 = 96 x 16x16x32, the others 24 x 32x32x16
    const double flops = (double)blocks * 4 * iters * (MODE == 1 ? 96.0 * 16 * 16 * 32 : 24.0 * 32 * 32 * 16) * 2.0;
    const double tf = flops / best / 1e9;
    printf("%-86s %8.2f ms  %7.1f TFLOP/s  = %.3f of 2500\n", name, best, tf, tf / 2500.0);
    hipFree(o);
}

static unsigned short bf16_trunc(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    static uint4 h[4096];
    uint4* d;
    hipMalloc(&d, sizeof(h));
    srand(1);
    // slot (x*3 + q) holds term q (by truncation, as the kernels split) of eight N(0,1) floats
    for (int x = 0; x < 4096 / 3; ++x) {
        unsigned short t[3][8];
        for (int e = 0; e < 8; ++e) {
            float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
            float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            for (int q = 0; q < 3; ++q) { t[q][e] = bf16_trunc(v); v -= bf16_f(t[q][e]); }
        }
        for (int q = 0; q < 3; ++q) {
            unsigned w[4];
            for (int j = 0; j < 4; ++j) w[j] = t[q][2 * j] | ((unsigned)t[q][2 * j + 1] << 16);
            h[x * 3 + q] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 100000;
    run<0>("32x32x16: 4 row fragments x 1 weight fragment, product-major (the halo kernel's tap)", d, 512, iters);
    run<1>("16x16x32: 4 row x 4 weight fragments, same accumulator registers", d, 512, iters / 2);
    run<2>("32x32x16: A-major order (six products of a row fragment back to back)", d, 512, iters);
    run<3>("32x32x16: product-major, ONE wave per SIMD (256 blocks)", d, 256, iters * 2);
    run<0>("32x32x16: product-major again (drift check)", d, 512, iters);
    return 0;
}
