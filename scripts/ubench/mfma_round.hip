// How does v_mfma_f32_32x32x16_bf16 round?  (diagnosis behind DESIGN.md 3.1b: accumulation error of the split kernels)
// One wave; A[i][k], B[k][j] chosen so that output (0,0) sees chosen products.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned short f2bf(float f) { return (unsigned short)(__float_as_uint(f) >> 16); }

// a_k, b_k: 16 k-values for row 0 / column 0 (all other rows/cols zero); c0: accumulator input at (0,0); bf16 MFMA
__global__ void k_bf16(const float* a, const float* b, float c0, float* out) {
    const int lane = threadIdx.x;
    union { bf16x8 v; unsigned short u[8]; } A, B;
    for (int j = 0; j < 8; ++j) {
        const int k = (lane >> 5) * 8 + j;          // lane half g owns k = 8g..8g+7
        A.u[j] = (lane & 31) == 0 ? f2bf(a[k]) : 0;
        B.u[j] = (lane & 31) == 0 ? f2bf(b[k]) : 0;
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (lane == 0) c[0] = c0;                        // D[i][j]: lane -> column j = lane&31, rows (r&3)+8*(r>>2)+4*(lane>>5)
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.v, B.v, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}
__global__ void k_f32(const float* a, const float* b, float c0, float* out) {   // v_mfma_f32_32x32x2_f32, k = 0,1
    const int lane = threadIdx.x;
    const float av = (lane & 31) == 0 ? a[lane >> 5] : 0.f, bv = (lane & 31) == 0 ? b[lane >> 5] : 0.f;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    if (lane == 0) c[0] = c0;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}

static float run(bool bf, const float* a, const float* b, float c0) {
    float *da, *db, *dout, h;
    hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
    if (bf) hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
    else hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, da, db, c0, dout);
    hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return h;
}

int main() {
    const float T = 16777216.f;   // 2^24: ulp = 2 above, 1 below
    float a[16], b[16];
    auto clear = [&]() { memset(a, 0, sizeof a); memset(b, 0, sizeof b); };
    struct Case { const char* name; float c0; int n; float av; float bv; };
    const Case cases[] = {
        {"C=2^24 + one product 1.5         (RN:+2 RZ:+0)", T, 1, 1.5f, 1.f},
        {"C=2^24 + one product 1.0 (tie)   (RNE:+0 RZ:+0 RU:+2)", T, 1, 1.0f, 1.f},
        {"C=2^24 + one product 3.0 (tie)   (RNE:+4 RZ:+2)", T, 1, 3.0f, 1.f},
        {"C=2^24 + 16 products of 0.25     (exact sum first:+4, sequential:+0)", T, 16, 0.25f, 1.f},
        {"C=2^24 + 16 products of 0.125    (sum 2: exact)", T, 16, 0.125f, 1.f},
        {"C=2^24 + 2 products of 0.75      (sum 1.5 -> RN:+2)", T, 2, 0.75f, 1.f},
        {"C=2^24 + 8 products of 0.1875    (sum 1.5; k in one lane half)", T, 8, 0.1875f, 1.f},
        {"C=-2^24 + one product 1.25       (exact -(2^24-1.25): RN:-(2^24-1) RZ:-(2^24-2))", -T, 1, 1.25f, 1.f},
        {"C=-2^24 + one product 1.75       (RN:-(2^24-2) RZ:-(2^24-2))", -T, 1, 1.75f, 1.f},
        {"C=2^24 + one product -0.75       (exact 2^24-0.75: RN:2^24-1 RZ:2^24-1)", T, 1, -0.75f, 1.f},
        {"C=2^24 + one product -0.25       (exact 2^24-0.25: RN:2^24 RZ:2^24-1)", T, 1, -0.25f, 1.f},
        {"C=1 + one product 2^-25          (RN:1 RZ:1)", 1.f, 1, 2.9802322e-08f, 1.f},
        {"C=1 + one product 1.5*2^-24      (RN:1+2^-23 RZ:1)", 1.f, 1, 8.9406967e-08f, 1.f},
        {"C=0 + subnormal bf16 input 2^-130 * 2^10 (flushed -> 0?)", 0.f, 1, 7.3468e-40f, 1024.f},
        {"C=0 + 2^-100 * 2^-40 (subnormal result 2^-140)", 0.f, 1, 7.8886e-31f, 9.0949e-13f},
    };
    for (const Case& cs : cases) {
        clear();
        for (int k = 0; k < cs.n; ++k) { a[k] = cs.av; b[k] = cs.bv; }
        const float r = run(true, a, b, cs.c0);
        float r32 = 0.f / 0.f;
        if (cs.n <= 2) r32 = run(false, a, b, cs.c0);
        printf("%-86s bf16: %.9g (delta %+g)   f32-mfma: %.9g\n", cs.name, r, (double)r - (double)cs.c0, r32);
    }
    // 16 products with cancellation: +2^24*1, -2^24*1, then 14 x 0.25 -> exact 3.5; C = 0
    clear();
    a[0] = T; b[0] = 1.f; a[1] = -T; b[1] = 1.f;
    for (int k = 2; k < 16; ++k) { a[k] = 0.25f; b[k] = 1.f; }
    printf("%-86s bf16: %.9g\n", "C=0 + (2^24 - 2^24 + 14*0.25): exact 3.5", run(true, a, b, 0.f));
    // order dependence: big product at k=15 instead
    clear();
    a[15] = T; b[15] = 1.f; a[14] = -T; b[14] = 1.f;
    for (int k = 0; k < 14; ++k) { a[k] = 0.25f; b[k] = 1.f; }
    printf("%-86s bf16: %.9g\n", "C=0 + (14*0.25 + 2^24 - 2^24): exact 3.5", run(true, a, b, 0.f));
    clear();
    a[0] = T; b[0] = 1.f;
    for (int k = 1; k < 16; ++k) { a[k] = 0.25f; b[k] = 1.f; }
    printf("%-86s bf16: %.9g\n", "C=0 + (2^24 + 15*0.25): exact 2^24+3.75 -> RN 2^24+4", run(true, a, b, 0.f));
    // alignment window inside one 8-product group: +2^m - 2^m + q (exact result q): which bits of q survive?
    for (int half = 0; half < 2; ++half) {
        for (int m = 16; m <= 40; m += 1) {
            clear();
            const float big = ldexpf(1.f, m);
            a[0] = big; b[0] = 1.f; a[1] = -big; b[1] = 1.f;
            const int kq = half ? 8 : 2;
            a[kq] = 1.9921875f; b[kq] = 1.9921875f;          // (2 - 2^-7)^2 = 3.96881103515625: 16 significant bits
            const float r = run(true, a, b, 0.f);
            clear();
            a[0] = big; b[0] = 1.f; a[1] = -big; b[1] = 1.f;
            a[kq] = -1.9921875f; b[kq] = 1.9921875f;
            const float rn = run(true, a, b, 0.f);
            printf("cancel 2^%d in k=0,1; q=+-3.96881103515625 at k=%d: got %+.11f / %+.11f\n", m, kq, r, rn);
        }
    }
    return 0;
}
