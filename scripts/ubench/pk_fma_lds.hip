// Reproducer attempt for the r02 "packed-FMA corruption" (profiles/r02_notes.md): SLP-vectorised
//     ds_read2_b32 v[d:d+1], addr offset0:.. offset1:..          (two LDS words into a register pair)
//     v_pk_fma_f32 acc[k:k+1], s[k:k+1], v[d:d+1] op_sel:[0,1,0]  (BOTH halves multiply by the HIGH register d+1)
// produced wrong LOW halves in single 16-lane rows of rd::conv_last_wgrad_tile_kernel, but only while main-stream kernels of
// the two-stream backward ran beside it.  This program isolates the instruction pair: kernel `victim` runs it in a loop
// against a scalar v_fma_f32 reference on the same operands; `stress_lds` / `stress_mfma` run on a second stream on the
// other half of every CU's resources.  Reports mismatches alone and under each stressor.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/pk_fma_lds.hip -o scripts/ubench/pk_fma_lds && scripts/ubench/pk_fma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[(blockIdx.x * 4096 + i) & 65535];
    __syncthreads();
    const float w0 = in[blockIdx.x & 255] + 1.25f, w1 = in[(blockIdx.x + 7) & 255] - 0.75f;
    v2f wv;                                                    // wave-uniform: an SGPR pair in the asm
    wv.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w0)));
    wv.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, w1)));
    v2f acc = {0.f, 0.f};
    float r0 = 0.f, r1 = 0.f;
    unsigned nbad = 0;
    const unsigned base = (threadIdx.x * 8) & 16383;           // byte address; offset1 = +1 dword, as in the original code
    for (int it = 0; it < iters; ++it) {
        const unsigned a = (base + it * 264) & 16376;
        v2f d;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory");
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wv), "v"(d));
        const float dh = lds[(a >> 2) + 1];
        r0 = __builtin_fmaf(wv.x, dh, r0);
        r1 = __builtin_fmaf(wv.y, dh, r1);
        if ((it & 63) == 63) {
            nbad += (acc.x != r0) + (acc.y != r1);
            acc.x = r0 = 0.f; acc.y = r1 = 0.f;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void stress_lds(float* out, int iters) {          // LDS bandwidth hog: b128 reads + writes
    __shared__ float4 l4[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) l4[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    float4 s = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const float4 v = l4[(threadIdx.x * 5 + it * 33) & 2047];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        l4[(threadIdx.x + it * 17) & 2047] = s;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
}

__global__ __launch_bounds__(256) void stress_mfma(const uint4* src, float* out, int iters) {   // matrix-pipe + power hog
    bf16x8 a = __builtin_bit_cast(bf16x8, src[threadIdx.x]), b = __builtin_bit_cast(bf16x8, src[threadIdx.x + 256]);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    unsigned* bad;
    uint4* src;
    hipMalloc(&in, 65536 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&bad, 4); hipMalloc(&src, 512 * 16);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipMemcpy(src, h, 512 * 16, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    const char* names[3] = {"alone", "beside an LDS-bound kernel", "beside an MFMA-bound kernel"};
    for (int mode = 0; mode < 3; ++mode) {
        unsigned total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemsetAsync(bad, 0, 4, s1);
            if (mode == 1) for (int q = 0; q < 4; ++q) stress_lds<<<512, 256, 0, s2>>>(out, 40000);
            if (mode == 2) for (int q = 0; q < 4; ++q) stress_mfma<<<512, 256, 0, s2>>>(src, out, 20000);
            for (int q = 0; q < 8; ++q) victim<<<1024, 256, 0, s1>>>(in, bad, 20000);
            hipDeviceSynchronize();
            unsigned b = 0;
            hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
            total += b;
        }
        printf("%-30s mismatching 64-FMA chains: %u of %.3g\n", names[mode], total, 20.0 * 8 * 1024 * 256 * (20000 / 64) * 2);
    }
    return 0;
}
