// Sustained v_mfma_f32_32x32x16_bf16 rate with operands of different bit activity: does the power limit (DVFS) cap the
// MFMA pipe below its nominal 2.5 PFLOP/s?  One 256-thread block per CU x `waves` blocks, four independent accumulator
// chains per wave, operands fixed in registers (no memory traffic at all): the pipe is 100 % busy in cycles, so
// TFLOP/s / 2500 = effective clock / 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_power.hip -o scripts/ubench/mfma_power && scripts/ubench/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters) {
    // four A and four B fragments per lane from `src` (zeros, a constant, or random bits)
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + i) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + 4 + i) & 4095]);
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i + 2 * u) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static void run(const char* name, const uint4* d, int blocks, int iters) {
    float* o;
    hipMalloc(&o, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(d, o, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(d, o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
    const double tf = flops / ms / 1e9;
    printf("%-34s blocks=%4d  %8.2f ms  %7.1f TFLOP/s  = %.3f of 2500  (effective clock %.2f GHz)\n", name, blocks, ms, tf, tf / 2500.0,
           tf / 2500.0 * 2.4);
    hipFree(o);
}

int main() {
    uint4 h[4096];
    uint4* d;
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 4; ++mode) {
        srand(1);
        for (int i = 0; i < 4096; ++i) {
            unsigned w[4];
            for (int j = 0; j < 4; ++j) {
                if (mode == 0) w[j] = 0u;                                   // zeros
                else if (mode == 1) w[j] = 0x3f803f80u;                    // 1.0, 1.0
                else if (mode == 2) {                                      // random bf16 in [1, 2): random mantissas only
                    w[j] = 0x3f803f80u | ((rand() & 0x7f) << 16) | (rand() & 0x7f);
                } else {                                                   // random sign / exponent (2^-8 .. 2^8) / mantissa
                    unsigned lo = ((rand() & 1) << 15) | ((119 + rand() % 17) << 7) | (rand() & 0x7f);
                    unsigned hi = ((rand() & 1) << 15) | ((119 + rand() % 17) << 7) | (rand() & 0x7f);
                    w[j] = (hi << 16) | lo;
                }
            }
            h[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        const char* names[4] = {"zeros", "ones", "random mantissa, [1,2)", "random sign / exponent / mantissa"};
        for (int waves = 1; waves <= 2; ++waves) {
            char nm[96];
            snprintf(nm, sizeof(nm), "%s, %d wave/SIMD", names[mode], waves);
            run(nm, d, 256 * waves, 20000 / waves);
        }
    }
    return 0;
}
