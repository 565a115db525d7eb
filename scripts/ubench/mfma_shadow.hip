// What hides in the shadow of back-to-back v_mfma_f32_32x32x16_bf16 on gfx950?  One 256-thread block per CU (one wave per
// SIMD) or 512 threads (two waves per SIMD).  The MFMA wave issues 4 independent accumulator chains; between MFMAs it
// issues NV VALU ops (v_and / v_sub / v_fma mix = the split arithmetic), NL ds_read_b128 or NW ds_write_b64.
// Prints cycles per MFMA (at the measured time, nominal 2.4 GHz is NOT assumed: also prints MFMA-only time for scale).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NV, int NL, int NW, int THREADS, int OTHER>   // OTHER: what waves 4-7 do (0 nothing/absent, 1 VALU, 2 ds_read, 3 ds_write)
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float sm[16384];
    const int t = threadIdx.x;
    for (int i = t; i < 16384; i += THREADS) sm[i] = seed + i;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float res = 0.f;
    const float* rp = sm + (t & 255) * 36;            // padded rows: conflict-free b128 reads
    float* wp = sm + 9216 + (t & 255) * 2;
    if (wave < 4) {
        f32x16 a0, a1, a2, a3;
        for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
        bf16x8 fa, fb;
        for (int j = 0; j < 8; ++j) { fa[j] = (__bf16)(seed + t * 0.001f + j); fb[j] = (__bf16)(seed * 0.5f + j); }
        float x0 = seed + t, x1 = seed - t;
        float4 l0 = make_float4(0, 0, 0, 0);
        for (int it = 0; it < iters; ++it) {
            v4f q[4 * (NL > 0 ? NL : 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x16& acc = u == 0 ? a0 : u == 1 ? a1 : u == 2 ? a2 : a3;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (v % 3 == 0) x0 = __uint_as_float(__float_as_uint(x0) & 0xffff0fffu);
                    else if (v % 3 == 1) x1 = x1 - x0;
                    else x0 = fmaf(x0, 1.0001f, x1);
                }
#pragma unroll
                for (int v = 0; v < NL; ++v) q[u * NL + v] = *reinterpret_cast<const v4f*>(rp + ((it + u * NL + v) & 7) * 4);
#pragma unroll
                for (int v = 0; v < NW; ++v) { v2f w2; w2.x = x0 + it; w2.y = x1; *reinterpret_cast<v2f*>(wp + ((it + u + v) & 3) * 512) = w2; }
            }
            if (NL > 0) {           // one wait for the 4*NL reads of this group (after the 4 MFMAs were issued)
#pragma unroll
                for (int v = 0; v < 4 * NL; ++v) l0.x += q[v].x;
            }
            if (NW > 0) asm volatile("" ::: "memory");
        }
        for (int r = 0; r < 16; ++r) res += a0[r] + a1[r] + a2[r] + a3[r];
        res += x0 + x1 + l0.x;
    } else if (OTHER == 1) {
        float x0 = seed + t, x1 = seed - t, x2 = seed * t, x3 = seed + 2 * t;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {         // 4 x 12 = 48 VALU per iteration = 12 per MFMA of the partner wave
                unsigned h;
                h = __float_as_uint(x0) & 0xffff0000u; x0 = (x0 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x1) & 0xffff0000u; x1 = (x1 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x2) & 0xffff0000u; x2 = (x2 - __uint_as_float(h)) * 1.0001f + 1.f;
                h = __float_as_uint(x3) & 0xffff0000u; x3 = (x3 - __uint_as_float(h)) * 1.0001f + 1.f;
            }
        }
        res = x0 + x1 + x2 + x3;
    } else if (OTHER == 2) {
        float4 l0 = make_float4(0, 0, 0, 0);
        for (int it = 0; it < iters; ++it) {
            v4f q[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) q[v] = *reinterpret_cast<const v4f*>(rp + ((it + v) & 7) * 4);   // 2 per partner MFMA
#pragma unroll
            for (int v = 0; v < 8; ++v) l0.x += q[v].x;
        }
        res = l0.x;
    } else if (OTHER == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int v = 0; v < 4; ++v)            // 4 ds_write_b64 per iteration = 1 per MFMA of the partner wave
                { v2f w2; w2.x = seed + it; w2.y = seed; *reinterpret_cast<v2f*>(wp + ((it + v) & 3) * 512) = w2; }
            asm volatile("" ::: "memory");
        }
    }
    out[blockIdx.x * THREADS + t] = res;
}

template <int NV, int NL, int NW, int THREADS, int OTHER>
static float run(int iters) {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, NL, NW, THREADS, OTHER><<<256, THREADS>>>(d, iters / 8 + 1, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NV, NL, NW, THREADS, OTHER><<<256, THREADS>>>(d, iters, 1.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}

int main() {
    const int IT = 16000;                   // x4 MFMAs per wave
    const float base = run<0, 0, 0, 256, 0>(IT);
    auto rep = [&](const char* name, float ms) { printf("%-58s %.3f ms  = %5.1f x MFMA-only; extra per MFMA = %5.1f %% of an MFMA slot\n", name, ms, ms / base, (ms / base - 1) * 100); };
    rep("MFMA only (1 wave/SIMD)", base);
    rep("same wave +1 VALU per MFMA", run<1, 0, 0, 256, 0>(IT));
    rep("same wave +2 VALU per MFMA", run<2, 0, 0, 256, 0>(IT));
    rep("same wave +3 VALU per MFMA", run<3, 0, 0, 256, 0>(IT));
    rep("same wave +4 VALU per MFMA", run<4, 0, 0, 256, 0>(IT));
    rep("same wave +6 VALU per MFMA", run<6, 0, 0, 256, 0>(IT));
    rep("same wave +9 VALU per MFMA", run<9, 0, 0, 256, 0>(IT));
    rep("same wave +12 VALU per MFMA", run<12, 0, 0, 256, 0>(IT));
    rep("same wave +1 ds_read_b128 per MFMA", run<0, 1, 0, 256, 0>(IT));
    rep("same wave +2 ds_read_b128 per MFMA", run<0, 2, 0, 256, 0>(IT));
    rep("same wave +4 ds_read_b128 per MFMA", run<0, 4, 0, 256, 0>(IT));
    rep("same wave +1 ds_write_b64 per MFMA", run<0, 0, 1, 256, 0>(IT));
    rep("same wave +2 ds_write_b64 per MFMA", run<0, 0, 2, 256, 0>(IT));
    rep("same wave +3 VALU +1 ds_read +1 ds_write per MFMA", run<3, 1, 1, 256, 0>(IT));
    rep("2 waves/SIMD, partner idle", run<0, 0, 0, 512, 0>(IT));
    rep("2 waves/SIMD, partner 12 VALU per MFMA", run<0, 0, 0, 512, 1>(IT));
    rep("2 waves/SIMD, partner 2 ds_read_b128 per MFMA", run<0, 0, 0, 512, 2>(IT));
    rep("2 waves/SIMD, partner 1 ds_write_b64 per MFMA", run<0, 0, 0, 512, 3>(IT));
    return 0;
}
