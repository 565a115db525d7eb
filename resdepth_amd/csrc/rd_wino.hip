// conv3x3 forward / data gradient as Winograd F(2x2, 3x3) on the split-bf16 MFMA pipe.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      d: 4x4 input tile (stride 2, pad 1), g: 3x3 filter, Y: 2x2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// 16 multiplies per 2x2 outputs and channel pair instead of 36: 2.25x fewer MFMA products than the direct (halo) kernel.
// In fp32 the result differs from the direct convolution by ~2x its rounding error (measured: rel-L2 5e-7 vs 2.3e-7).
//
// Block = 8x16 output pixels (32 Winograd tiles) x 128 output channels; wave w owns channels 32 w .. 32 w + 31 and keeps
// hi / lo accumulators for TWO transform positions (64 registers) + the 2x2 outputs (64): 246 VGPRs, two waves per SIMD.
// Eight passes (transform row r = 0..3) x (position pair 0-1 / 2-3); per pass the K loop runs over 32-channel chunks:
//   staging   thread (tile, channel quad): 2 input rows x 3 columns (6 x 16-byte loads), the B^T row combination, the two
//             column combinations V[r][c], each split into three bf16 terms -> LDS row (c, tile), 240 bytes (2 sub-chunks x
//             3 terms x 16 bf16 + pad: a ds_read_b128 of 16 consecutive rows is bank-conflict free);
//   MFMAs     per (sub-chunk, c): A fragments from LDS, U = G g G^T fragments (pre-split, fragment order) straight from
//             global, six products: a1 b1 -> hi[c], the five low-order ones -> lo[c];
//   after the K loop: M = hi + lo; pair 0 contributes (M0 + M1, M1) to (T_0, T_1), pair 1 (M2, -M2 - M3);
//             Y[i][j] += A^T[i][r] T_j.
// Epilogue: rd_nt.h (patch mode 2) -- stores, BN forward statistics, BN-backward statistics hook.
#include <stdlib.h>

#include <type_traits>

#include "rd_common.h"
#include "rd_mfma_dev.h"
#include "rd_nt.h"

namespace rd {

template <int EPI>
__global__ __launch_bounds__(256) void conv3_wino_split_kernel(NtParams p) {
    constexpr int BM = 128, BN = 128, WM = 1, WN = 4;
    constexpr int RS = 60;                         // LDS row stride in words (240 B: 16 consecutive rows never share a bank)
    constexpr int STAGE = 64 * RS;                 // 2 positions x 32 tiles
    constexpr int EPI_WORDS = 32 * (BN + 4) + 512;
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    static_assert(SMEM >= 256 * 16, "BN-backward statistics scratch");
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int n0 = tile_n * BN;
    const int H = p.H, W = p.W;
    const int pxs = W >> 4, pys = H >> 3;
    const int pbx = tile_m % pxs, pby = (tile_m / pxs) % pys, img = tile_m / (pxs * pys);
    const int x0 = pbx * 16, y0 = pby * 8;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lrow = lane & 31, half = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.Bwino, p.b_bytes);

    // ---- staging task of this thread: Winograd tile (ty, tx), channel quad q8 of the 32-channel chunk
    const int stile = t & 31, q8 = t >> 5;
    const int sty = stile >> 3, stx = stile & 7;
    const int w_lds = stile * RS + (q8 >> 2) * 24 + (q8 & 3) * 2;       // + (c & 1) * 32 * RS + term * 8

    const int nb = (n0 >> 5) + wave;
    const int NB = (p.N + 31) >> 5;
    const int nk1 = p.chunks;                     // 16-channel K-steps per position
    const bool nb_ok = nb * 32 < p.N;

    // A pass = (transform row R, position pair HP), positions c = 2 HP, 2 HP + 1 of row R, pass index P = 2 R + HP at RUN time
    // (eight compile-time instances of the stage loop made hipcc's allocator spill: 256 + 256 registers + scratch; one instance
    // needs 144).  B^T row R combines input rows (ra, rb) as d[ra] + sg d[rb]:  R0: d0 - d2, R1: d1 + d2, R2: d2 - d1,
    // R3: d1 - d3;  with rc[b] that combination of input column b:  V0 = rc0 - rc2, V1 = rc1 + rc2, V2 = rc2 - rc1,
    // V3 = rc1 - rc3, so pair 0 needs columns 0..2 and pair 1 columns 1..3: local columns l = b - HP.
    unsigned offa[3], offb[3];      // byte offsets of the two input rows x three columns of the current pass (kOOB outside)
    float sg = -1.f;
    int hp = 0;
    auto set_pass = [&](int P) {
        const int R = P >> 1;
        hp = P & 1;
        const int ra = R == 0 ? 0 : (R == 2 ? 2 : 1), rb = R == 0 ? 2 : (R == 1 ? 2 : (R == 2 ? 1 : 3));
        sg = R == 1 ? 1.f : -1.f;
        const int ya = y0 + 2 * sty - 1 + ra, yb = y0 + 2 * sty - 1 + rb;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int x = x0 + 2 * stx - 1 + l + hp;
            const bool xok = (unsigned)x < (unsigned)W;
            offa[l] = (xok && (unsigned)ya < (unsigned)H) ? (unsigned)(((((long)img * H + ya) * W + x) * p.Cin + q8 * 4) * 4) : kOOB;
            offb[l] = (xok && (unsigned)yb < (unsigned)H) ? (unsigned)(((((long)img * H + yb) * W + x) * p.Cin + q8 * 4) * 4) : kOOB;
        }
    };
    float4 raw[2][3];
    auto load_rows = [&](int cp) {
        const bool cok = cp * 32 + q8 * 4 < p.Cin;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            raw[0][l] = buf_load4(rsA, cok ? offa[l] : kOOB, (unsigned)(cp * 32 * 4));
            raw[1][l] = buf_load4(rsA, cok ? offb[l] : kOOB, (unsigned)(cp * 32 * 4));
        }
    };
    auto f4sub = [](const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    auto f4add = [](const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    auto f4sel = [](bool c, const float4 a, const float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); };
    // (sg_, hp_: the pass the raw rows were loaded for -- the prefetch across a pass boundary stores with the NEXT pass's values)
    auto store_rows = [&](float* stage, float sg_, int hp_) {
        float4 rc[3];
#pragma unroll
        for (int l = 0; l < 3; ++l)
            rc[l] = make_float4(fmaf(sg_, raw[1][l].x, raw[0][l].x), fmaf(sg_, raw[1][l].y, raw[0][l].y),
                                fmaf(sg_, raw[1][l].z, raw[0][l].z), fmaf(sg_, raw[1][l].w, raw[0][l].w));
        // pair 0: (V0, V1) = (rc0 - rc2, rc1 + rc2);  pair 1 (local columns = 1..3): (V2, V3) = (rc2 - rc1, rc1 - rc3) = (l1 - l0, l0 - l2)
        const float4 d02 = f4sub(rc[0], rc[2]);
        const float4 v[2] = {f4sel(hp_ != 0, f4sub(rc[1], rc[0]), d02), f4sel(hp_ != 0, d02, f4add(rc[1], rc[2]))};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint2 ph, pm, pl;
            split_pack4(v[c], ph, pm, pl);
            float* row = stage + c * 32 * RS + w_lds;
            *reinterpret_cast<uint2*>(row) = ph;
            *reinterpret_cast<uint2*>(row + 8) = pm;
            *reinterpret_cast<uint2*>(row + 16) = pl;
        }
    };

    const int a_rd = lrow * RS + half * 4;        // + c * 32 * RS + s * 24 + q * 8
    // U fragment (position xi, row block nb, K-step kt, term q): the wave / lane part of the address is the per-lane
    // offset, the (xi, kt, q) part a scalar offset (a lane-dependent scalar offset would cost a waterfall loop)
    const unsigned b_voff = nb_ok ? (unsigned)((long)nb * nk1 * 3 * 1024) + lane * 16 : kOOB;
    auto load_b = [&](int xi, int kt, uint4 (&rb)[3]) {
        const unsigned voff = kt < nk1 ? b_voff : kOOB;
        const unsigned base = (unsigned)(((long)xi * NB * nk1 + kt) * 3 * 1024);
#pragma unroll
        for (int q = 0; q < 3; ++q) rb[q] = buf_load4u(rsB, voff, base + (unsigned)(q * 1024));
    };

    f32x16 hi[2], lo[2], Y[4][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        hi[0][e] = hi[1][e] = lo[0][e] = lo[1][e] = 0.f;
        Y[0][0][e] = Y[1][0][e] = Y[2][0][e] = Y[3][0][e] = 0.f;
    }
    const int ncp = (p.Cin + 31) / 32;

    // one (pass, chunk pair) stage: 4 (sub-chunk, position) steps of 6 MFMAs; operands of step k + 1 are requested before
    // the MFMAs of step k
    auto mma_stage = [&](const float* stage, int xi0, int cp) {
        uint4 b0[3], b1[3];
        bf16x8 a0[3], a1[3];
        auto read_a = [&](int k, bf16x8 (&af)[3]) {
            const int s = k >> 1, c = k & 1;
#pragma unroll
            for (int q = 0; q < 3; ++q) af[q] = *reinterpret_cast<const bf16x8*>(stage + c * 32 * RS + a_rd + s * 24 + q * 8);
        };
        load_b(xi0, 2 * cp, b0);
        read_a(0, a0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = k & 1;
            uint4(&bc)[3] = (k & 1) ? b1 : b0;
            uint4(&bn)[3] = (k & 1) ? b0 : b1;
            bf16x8(&ac)[3] = (k & 1) ? a1 : a0;
            bf16x8(&an)[3] = (k & 1) ? a0 : a1;
            if (k < 3) {
                load_b(xi0 + ((k + 1) & 1), 2 * cp + ((k + 1) >> 1), bn);
                read_a(k + 1, an);
            }
            bf16x8 bf[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) bf[q] = __builtin_bit_cast(bf16x8, bc[q]);
#pragma unroll
            for (int t6 = 0; t6 < 5; ++t6)
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[PA6[t6]], bf[PB6[t6]], lo[c], 0, 0, 0);
            hi[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[0], bf[0], hi[c], 0, 0, 0);
        }
    };

    // software pipeline: while the MFMAs of one stage run, the next stage's input rows are in flight; the first stage of
    // the next pass is prefetched during the last stage of the current one
    int buf = 0;
    set_pass(0);
    load_rows(0);
    store_rows(smem, sg, hp);
    __syncthreads();
    for (int P = 0; P < 8; ++P) {
        const int R = P >> 1, HP = P & 1;
        for (int cp = 0; cp < ncp; ++cp) {
            const bool more = cp + 1 < ncp, nextpass = !more && P < 7;
            if (nextpass) set_pass(P + 1);
            if (more) load_rows(cp + 1);
            else if (nextpass) load_rows(0);
            mma_stage(smem + buf * STAGE, 4 * R + 2 * HP, cp);
            if (more || nextpass) store_rows(smem + (buf ^ 1) * STAGE, sg, hp);
            __syncthreads();
            buf ^= 1;
        }
        // output transform.  Row R, columns:  T_0 = M0 + M1 + M2,  T_1 = M1 - M2 - M3  =>  pair 0 contributes (M0 + M1, M1),
        // pair 1 (M2, -M2 - M3);   Y[i][j] += A^T[i][R] T_j with A^T = [1 1 1 0; 0 1 -1 -1]
        const float a0 = R < 3 ? 1.f : 0.f, a1 = R == 0 ? 0.f : (R == 1 ? 1.f : -1.f);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float ma = merge_hi_lo(hi[0][e], lo[0][e]), mb = merge_hi_lo(hi[1][e], lo[1][e]);
            const float t0 = HP ? ma : ma + mb, t1 = HP ? -ma - mb : mb;
            Y[0][0][e] = fmaf(a0, t0, Y[0][0][e]);
            Y[1][0][e] = fmaf(a0, t1, Y[1][0][e]);
            Y[2][0][e] = fmaf(a1, t0, Y[2][0][e]);
            Y[3][0][e] = fmaf(a1, t1, Y[3][0][e]);
            hi[0][e] = hi[1][e] = lo[0][e] = lo[1][e] = 0.f;
        }
    }
    const int m0 = ((img * H + y0) * W) + x0;
    nt_epilogue<BM, BN, WM, WN, EPI, SMEM, 1>(Y, smem, p, m0, n0, tile_m);
}

// host side -----------------------------------------------------------------------------------------------------------------
bool wino_channels_ok(int n, int cin) { return tune(TUNE_NT_WINO) != 0 && wino_shape(n, cin); }

int wino_launch(NtParams p, hipStream_t s) {
    p.patch = 2;
    p.tiles_n = p.N / 128;
    p.chunks = (p.Cin + 15) / 16;
    const int tiles_m = p.M / 128;
    hipLaunchKernelGGL((conv3_wino_split_kernel<EPI_STORE>), dim3(tiles_m * p.tiles_n), dim3(256), 0, s, p);
    RD_LAUNCH_CHECK("conv3_wino");
    return RD_OK;
}

}  // namespace rd
