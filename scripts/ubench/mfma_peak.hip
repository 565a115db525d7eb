// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate on this MI355X, with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ __attribute__((aligned(16))) float sm[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) sm[i] = a0 + i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = sm + (wave * 64 + (lane & 31)) * 36 + (lane >> 5) * 4;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            float4 af[2], bf[2];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                af[0] = *reinterpret_cast<const float4*>(base + kk * 8);
                af[1] = *reinterpret_cast<const float4*>(base + 32 * 36 + kk * 8);
                bf[0] = *reinterpret_cast<const float4*>(base + kk * 8 + 4 * 36);
                bf[1] = *reinterpret_cast<const float4*>(base + 32 * 36 + kk * 8 + 4 * 36);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float av0 = s == 0 ? af[0].x : s == 1 ? af[0].y : s == 2 ? af[0].z : af[0].w;
                    const float av1 = s == 0 ? af[1].x : s == 1 ? af[1].y : s == 2 ? af[1].z : af[1].w;
                    const float bv0 = s == 0 ? bf[0].x : s == 1 ? bf[0].y : s == 2 ? bf[0].z : bf[0].w;
                    const float bv1 = s == 0 ? bf[1].x : s == 1 ? bf[1].y : s == 2 ? bf[1].z : bf[1].w;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0], 0, 0, 0);
                    acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[1 % NACC], 0, 0, 0);
                    acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[2 % NACC], 0, 0, 0);
                    acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[3 % NACC], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i % NACC], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char* name, int blocks, int iters) {
    float* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDS><<<blocks, 256>>>(d, 64, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDS><<<blocks, 256>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64 * 2.0 * 32 * 32 * 2;
    printf("%-28s blocks=%5d iters=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, iters, ms, flops / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<4, false>("reg-only 4acc 1wave/SIMD", 256, 4000);
        run<4, false>("reg-only 4acc 2wave/SIMD", 512, 4000);
        run<4, false>("reg-only 4acc 4wave/SIMD", 1024, 2000);
        run<1, false>("reg-only 1acc 1wave/SIMD", 256, 4000);
        run<2, false>("reg-only 2acc 1wave/SIMD", 256, 4000);
        run<4, true>("lds-frag 4acc 1wave/SIMD", 256, 4000);
        run<4, true>("lds-frag 4acc 2wave/SIMD", 512, 4000);
        run<4, true>("lds-frag 4acc 3wave/SIMD", 768, 3000);
    }
    return 0;
}
