// Do VALU instructions of a SECOND wave on the same SIMD overlap with bf16 MFMAs?  And does it depend on where the MFMA
// accumulators live (arch VGPRs vs AccVGPRs)?  512-thread blocks, one per CU: waves 0-3 run an MFMA loop (inline asm so
// the accumulator register class is pinned), waves 4-7 a VALU loop (the split3 instruction mix).  Reported: kernel time
// with only the MFMA waves working, only the VALU waves working, and both.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_overlap.hip -o scripts/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

template <bool AGPR, bool SAMEWAVE, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, int mfma_iters, int valu_iters, float seed) {
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float res = 0.f;
    if (wave < 4) {
        f32x16 a0, a1, a2, a3;
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(seed + t * 0.001f + j); fb[j] = (__bf16)(seed * 0.5f + j); }
        float x = seed + t;
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (AGPR) { MFMA_A(a0, fa, fb); MFMA_A(a1, fa, fb); MFMA_A(a2, fa, fb); MFMA_A(a3, fa, fb); }
                else { MFMA_V(a0, fa, fb); MFMA_V(a1, fa, fb); MFMA_V(a2, fa, fb); MFMA_V(a3, fa, fb); }
                if (SAMEWAVE) {          // the same VALU mix inside the MFMA wave (5 VALU per MFMA)
#pragma unroll
                    for (int v = 0; v < 20; ++v) {
                        const unsigned h = __float_as_uint(x) & 0xffff0000u;
                        x = (x - __uint_as_float(h)) * 1.0001f + 1.f;
                    }
                }
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int r = 0; r < 16; ++r) res += a0[r] + a1[r] + a2[r] + a3[r];
        res += x;
    } else {
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
        float x0 = seed + t, x1 = seed - t, x2 = seed * t, x3 = seed + 2 * t;
        for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {       // split3-like chain: and, sub, mul-add (dependent per stream, 4 streams)
                unsigned h;
                h = __float_as_uint(x0) & 0xffff0000u; x0 = (x0 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x1) & 0xffff0000u; x1 = (x1 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x2) & 0xffff0000u; x2 = (x2 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x3) & 0xffff0000u; x3 = (x3 - __uint_as_float(h)) * 1.0001f + 1.f;
            }
        }
        res = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + t] = res;
}

template <bool AGPR, bool SAMEWAVE, int PRIO = 0>
static float run(int mi, int vi) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<AGPR, SAMEWAVE, PRIO><<<256, 512>>>(d, mi / 8 + 1, vi / 8 + 1, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<AGPR, SAMEWAVE, PRIO><<<256, 512>>>(d, mi, vi, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}

int main() {
    const int MI = 4000;          // x16 MFMAs per wave
    // VALU iterations chosen so that the VALU waves alone take about 60 % of the MFMA waves' time
    for (int vi : {0, 1500, 3000}) {
        const float mv = run<false, false>(MI, vi), ma = run<true, false>(MI, vi), v0 = run<true, false>(0, vi);
        printf("VALU iters %5d | VALU waves alone %.3f ms | MFMA(VGPR acc) + VALU waves %.3f ms | MFMA(AGPR acc) + VALU waves %.3f ms\n", vi, v0, mv, ma);
    }
    for (int vi : {1500, 3000})
        printf("VALU iters %5d with s_setprio on the VALU waves: prio1 %.3f ms, prio3 %.3f ms (AGPR acc); prio3 VGPR acc %.3f ms\n", vi,
               run<true, false, 1>(MI, vi), run<true, false, 3>(MI, vi), run<false, false, 3>(MI, vi));
    printf("MFMA alone: VGPR acc %.3f ms, AGPR acc %.3f ms (%.0f TF bf16)\n", run<false, false>(MI, 0), run<true, false>(MI, 0),
           256.0 * 4 * MI * 16 * 32 * 32 * 16 * 2 / (run<true, false>(MI, 0) * 1e-3) / 1e12);
    printf("same-wave VALU (20 VALU per 4 MFMA): VGPR acc %.3f ms, AGPR acc %.3f ms\n", run<false, true>(MI, 0), run<true, true>(MI, 0));
    return 0;
}
