// Implicit-GEMM convolution kernels on the gfx950 matrix cores.
//
//  NT form  C[M][N] = A[M][K] * B[N][K]^T      M = pixels, K = taps*Cin (im2col on the fly)
//      conv3x3 forward        A = x   (3x3 taps, zero padding),   B = wf[co][tap][ci]
//      conv3x3 data-gradient  A = dz  (3x3 taps),                 B = wd[ci][tap][co] (flipped)
//      convT2x2 forward       A = x   (1 tap),                    B = wtf[(ab,co)][ci], scatter epilogue
//      convT2x2 data-gradient A = dout(4 taps on the fine grid),  B = wtd[ci][(ab,co)]
//      conv1x1 (bilinear up-mode)  A = x (1 tap), plain epilogue
//  TN form  C[M][N] = sum_p A[p][M] * B[p][N]   reduction over pixels, split-K slabs
//      conv3x3 weight-gradient   A = dz[p][co],         B = x[p+tap][ci]   -> [co][(tap,ci)]
//      convT2x2 weight-gradient  A = dout[fine(p,ab)][co], B = x[p][ci]    -> [(ab,co)][ci]
//
// Replaces the ATen/cuDNN kernels behind nn.Conv2d / nn.ConvTranspose2d forward and autograd
// (reference call sites lib/UNet.py:4-5, 21, 44, 63-65, 85, 181).
//
// Two arithmetic modes, same fp32 results to rounding:
//   * split-bf16 (default): operands split exactly into three bf16 terms, six products per multiply on
//     v_mfma_f32_32x32x16_bf16 -- igemm_nt_split_kernel (generic NT), conv3_halo_split_kernel (3x3 forward / data
//     gradient with halo reuse), wgrad_tn_split_kernel (generic TN); the 3x3 weight gradient lives in
//     rd_wgrad_strip.hip.  Weights are pre-split by the pack kernels into MFMA-fragment order.
//   * exact f32 (RD_MFMA=f32, and always for short-K transposed convolutions): v_mfma_f32_32x32x2_f32 --
//     igemm_nt_kernel, wgrad_tn_kernel.  256 threads = 4 waves, block tile BMxBN, K-step 32, operands staged
//     global -> VGPR (one K-step ahead) -> LDS rows padded to 36 floats (conflict-free b128 fragment reads, k-pairing
//     permuted so that one read feeds four MFMA k-steps).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "rd_common.h"
#include "rd_mfma_dev.h"
#include "rd_nt.h"

namespace rd {

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(256) void igemm_nt_kernel(NtParams p) {
    constexpr int BK = 32, LS = BK + 4;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LS];
    float* As = smem;
    float* Bs = smem + BM * LS;

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int c4 = t & 7, r0 = t >> 3;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[AI], rb[BI];
    const int H = p.H, W = p.W;

    // ---- operand addressing, hoisted out of the K loop.  Loads are raw buffer loads: every lane carries a
    // 32-bit byte offset, lanes that must read zero (image border taps, rows/columns beyond M/N/Cin) carry an
    // out-of-range offset and the hardware returns 0 -- no exec-mask branches, no 64-bit pointer math.
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
    unsigned a_off[AI], a_val[AI], b_off[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool inm = m < p.M;
        unsigned val = 0;
        long pix = m;
        if (AMODE == A_CONV3) {
            int y, x, img_;
            pix_split(m, p.pd, img_, y, x);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
                if (((unsigned)(y + dy) < (unsigned)H) && ((unsigned)(x + dx) < (unsigned)W)) val |= 1u << t9;
            }
        } else if (AMODE == A_PLAIN) {
            val = 1u;
        } else {
            int jj, ii, img;
                    pix_split(m, p.pd, img, ii, jj);
            pix = ((long)img * (2 * H) + 2 * ii) * (2 * W) + 2 * jj;
            val = 0xFu;
        }
        a_val[i] = inm ? val : 0u;
        a_off[i] = (unsigned)((pix * p.Cin + c4 * 4) * 4);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_off[i] = n < p.N ? (unsigned)(((long)n * p.K + c4 * 4) * 4) : kOOB;
    }

    auto load_tile = [&](int kt) {
        // K order = channel-chunk outer, tap inner: the taps re-touch the same 128-byte row segments of
        // neighbouring pixels within consecutive K-steps (L1/L2 hits)
        const int chunk = kt / p.taps;
        const int tap = kt - chunk * p.taps;
        int shift;  // pixel shift of this tap, in A pixels
        if (AMODE == A_CONV3) shift = (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        else if (AMODE == A_PLAIN) shift = 0;
        else shift = (tap >> 1) * (2 * W) + (tap & 1);
        const unsigned toff = (unsigned)((shift * p.Cin + chunk * BK) * 4);
        const unsigned koff = (unsigned)((tap * p.Cin + chunk * BK) * 4);
        const bool cok = chunk * BK + c4 * 4 < p.Cin;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const bool ok = cok && ((a_val[i] >> tap) & 1u);
            ra[i] = buf_load4(rsA, ok ? a_off[i] + toff : kOOB, 0);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[i] = buf_load4(rsB, cok ? b_off[i] : kOOB, koff);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * LS + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * LS + c4 * 4]) = rb[i];
    };

    const int lrow = lane & 31, half = lane >> 5;
    const float* a_base = As + (wm * TM * 32 + lrow) * LS + half * 4;
    const float* b_base = Bs + (wn * TN * 32 + lrow) * LS + half * 4;

    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < p.nk; ++kt) {
        const bool more = kt + 1 < p.nk;
        if (more) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_base + i * 32 * LS + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_base + j * 32 * LS + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

    nt_epilogue<BM, BN, WM, WN, EPI, (BM + BN) * LS>(acc, smem, p, m0, n0, tile_m);
}

// ------------------------------------------------------------------------------------------
//  Split-bf16 NT kernel: the same implicit GEMM on v_mfma_f32_32x32x16_bf16
// ------------------------------------------------------------------------------------------
// Every fp32 operand is split EXACTLY into three bf16 terms x = x1 + x2 + x3 (8 significant bits each, by truncation)
// and a*b is evaluated as the six bf16 products of weight >= 2^-16 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1) with fp32
// accumulation.  Each bf16 product is exact in fp32; the dropped terms (a2b3, a3b2, a3b3) are <= 2^-23 |ab|, below the
// rounding error of one fp32 multiply -- results agree with the exact-f32 MFMA kernel to fp32 rounding (same parity
// tolerances in tests/).  Six MFMAs at 16x the f32-MFMA rate = 2.67x the fp32 matrix roofline (419 TFLOP/s).
//   A (activations / gradients): split while staged into LDS (4 VALU per element + 3 v_perm per pair).
//   B (weights): split ONCE by the pack kernels into the LDS row layout, staged by plain 16-byte copies.
//   K-step = 16 k-values (one MFMA K), LDS double-buffered (one barrier per K-step), two K-steps of global loads
//   in flight.  LDS row = 3 terms x 16 bf16 (96 B) + 16 B pad = 112 B: the 16 rows of a ds_read_b128 lane group start
//   on 16 distinct bank quads.
// LDS image of the A tile: one 128-byte row per pixel and stage = 8 chunks of 16 B, chunk c = 2*term + khalf (6 used).
// Chunks are XOR-swizzled by swz(row) so that (a) the 16 rows of a ds_read_b128 lane group hit 16 distinct bank quads
// and (b) the 4 rows of a ds_write_b64 lane group (rows r, r+2, r+4, r+6 by the thread->row map below) hit 4 distinct
// 32-byte spans: no LDS bank conflicts on either side.
__device__ __forceinline__ int swz(int r) { return (r ^ (r >> 3)) & 7; }

template <int NP, int BM, int BN, int WM, int WN, int AMODE, int EPI>
__device__ __forceinline__ void igemm_nt_split_body(const NtParams& p, float* smem, const Quant qz) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;               // split terms per operand
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TN == 1, "each wave owns one 32-column block: its B fragments come straight from global memory");
    constexpr int AI = BM / 64;                       // A: 4 float4 per 16-k row, 64 rows per pass
    constexpr int STAGE = BM * 32;                    // words per LDS stage (A only)
    constexpr int EPI_WORDS = 32 * (BN + 4) + 512;    // epilogue staging: one 32-row block per pass
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // staging map: 4 lanes per row (c4 = 16-byte quarter of the 64-byte k-row); within 16 lanes the rows are r, r+2, r+4, r+6
    const int c4 = t & 3;
    const int r0 = wave * 16 + ((lane >> 5) << 3) + (((lane >> 2) & 3) << 1) + ((lane >> 4) & 1);

    f32x16 acc[TM][1], lo[TM];       // hi / lo accumulators (rd_mfma_dev.h: PA6)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = lo[i][r] = 0.f;

    const int H = p.H, W = p.W;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(NP == 3 ? p.Bsplit3 : p.Bsplit, NP == 3 ? p.b_bytes3 : p.b_bytes);
    unsigned a_off[AI], a_val[AI];
    int a_wr[AI];       // LDS word address of this thread's 8 bytes of term 0 (terms 1, 2: chunk +2, +4 before the swizzle)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = r0 + 64 * i;
        const int m = m0 + row;
        const bool inm = m < p.M;
        unsigned val = 0;
        long pix = m;
        if (AMODE == A_CONV3) {
            int y, x, img_;
            pix_split(m, p.pd, img_, y, x);
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9) {
                const int dy = t9 / 3 - 1, dx = t9 % 3 - 1;
                if (((unsigned)(y + dy) < (unsigned)H) && ((unsigned)(x + dx) < (unsigned)W)) val |= 1u << t9;
            }
        } else if (AMODE == A_PLAIN) {
            val = 1u;
        } else {
            int jj, ii, img;
                    pix_split(m, p.pd, img, ii, jj);
            pix = ((long)img * (2 * H) + 2 * ii) * (2 * W) + 2 * jj;
            val = 0xFu;
        }
        a_val[i] = inm ? val : 0u;
        a_off[i] = (unsigned)((pix * p.Cin + c4 * 4) * 4);
        a_wr[i] = row * 32 + (c4 & 1) * 2;
    }
    // B fragments: packed layout [row-block of 32][kt][term][lane][16 B] (split_pack_kernel): one fully coalesced
    // 1 KB load per (wave, term, K-step)
    const int nb = (n0 >> 5) + wn;
    const unsigned b_off = (nb * 32 < p.N) ? (unsigned)(((long)nb * p.nk * NT) * 1024 + lane * 16) : kOOB;

    auto load_a = [&](int kt, float4 (&ra)[AI]) {
        // K order = channel-chunk outer, tap inner (see the f32 kernel).  kt >= nk (prefetch running past the end):
        // every lane carries the out-of-range offset -- keeps the number of outstanding loads static for s_waitcnt
        const int chunk = kt / p.taps;
        const int tap = kt - chunk * p.taps;
        int shift;
        if (AMODE == A_CONV3) shift = (tap / 3 - 1) * W + (tap - (tap / 3) * 3 - 1);
        else if (AMODE == A_PLAIN) shift = 0;
        else shift = (tap >> 1) * (2 * W) + (tap & 1);
        const unsigned toff = (unsigned)((shift * p.Cin + chunk * SK) * 4);
        const bool cok = kt < p.nk && chunk * SK + c4 * 4 < p.Cin;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const bool ok = cok && ((a_val[i] >> tap) & 1u);
            ra[i] = buf_load4(rsA, ok ? a_off[i] + toff : kOOB, 0);
        }
    };
    auto load_b = [&](int kt, uint4 (&rb)[3]) {
        const unsigned voff = kt < p.nk ? b_off : kOOB;
#pragma unroll
        for (int q = 0; q < NT; ++q) rb[q] = buf_load4u(rsB, voff, (unsigned)((kt * NT + q) * 1024));
    };
    auto store_a = [&](float* stage, const float4 (&ra)[AI]) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int sw = swz(r0 + 64 * i), hi = c4 >> 1;
            uint2 ph, pm, pl;
            split_pack4<NP>(ra[i], qz.sa, ph, pm, pl);
            *reinterpret_cast<uint2*>(stage + a_wr[i] + ((0 + hi) ^ sw) * 4) = ph;
            *reinterpret_cast<uint2*>(stage + a_wr[i] + ((2 + hi) ^ sw) * 4) = pm;
            if (kterm3<NP>()) *reinterpret_cast<uint2*>(stage + a_wr[i] + ((4 + hi) ^ sw) * 4) = pl;
        }
    };

    const int lrow = lane & 31, half = lane >> 5;
    int a_rd[TM][3];    // LDS word address of this lane's fragment of (row block i, term q) within a stage
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 3; ++q) a_rd[i][q] = row * 32 + ((2 * q + half) ^ swz(row)) * 4;
    }
    FR af[TM][3];
    auto read_a = [&](const float* stage, int i) {
#pragma unroll
        for (int q = 0; q < NT; ++q) af[i][q] = *reinterpret_cast<const FR*>(stage + a_rd[i][q]);
    };
    // six products per (a, b) pair, smallest terms first; lane half g owns k = 8g .. 8g+7 (the same 8 k for A and B)
    // One K-step.  Software pipeline per tile j:  global load (step j-4) -> split + LDS write (step j-2) -> fragment
    // read (step j-1, right after the MFMAs that last used the registers) -> MFMA (step j).  B: global load straight
    // into fragment registers at step j-3.  One barrier per step; LDS stage of tile j = j & 1.
    auto step = [&](int kt, float4 (&ra)[AI], uint4 (&bcur)[3], uint4 (&bnew)[3], float* wstage, const float* rstage) {
        store_a(wstage, ra);          // tile kt+2
        load_a(kt + 4, ra);
        load_b(kt + 3, bnew);
        FR bf[3];
#pragma unroll
        for (int q = 0; q < NT; ++q) bf[q] = __builtin_bit_cast(FR, bcur[q]);
        constexpr int GP = TM >= 2 ? 2 : 1;       // row blocks interleaved per group (independent accumulators)
#pragma unroll
        for (int g = 0; g < TM; g += GP) {
#pragma unroll
            for (int t6 = lo0<NP>(); t6 < 5; ++t6)
#pragma unroll
                for (int i = g; i < g + GP; ++i) lo[i] = mfma16<NP>(af[i][PA6[t6]], bf[PB6[t6]], lo[i]);
#pragma unroll
            for (int i = g; i < g + GP; ++i) acc[i][0] = mfma16<NP>(af[i][0], bf[0], acc[i][0]);
#pragma unroll
            for (int i = g; i < g + GP; ++i) read_a(rstage, i);      // tile kt+1, consumed one step later
        }
        __syncthreads();
    };

    float* st0 = smem;
    float* st1 = smem + STAGE;
    float4 ra0[AI], ra1[AI];
    uint4 b0[3], b1[3], b2[3], b3[3];
    load_a(0, ra0);
    load_a(1, ra1);
    load_b(0, b0);
    load_b(1, b1);
    load_b(2, b2);
    store_a(st0, ra0);
    load_a(2, ra0);
    store_a(st1, ra1);
    load_a(3, ra1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) read_a(st0, i);
    __syncthreads();
    for (int kt = 0; kt < p.nk; kt += 4) {      // nk is rounded up to a multiple of 4 with all-zero K-steps
        step(kt, ra0, b0, b3, st0, st1);
        step(kt + 1, ra1, b1, b0, st1, st0);
        step(kt + 2, ra0, b2, b1, st0, st1);
        step(kt + 3, ra1, b3, b2, st1, st0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = merge_q<NP>(acc[i][0][r], lo[i][r], qz.dexp);
    nt_epilogue<BM, BN, WM, WN, EPI, SMEM, 1>(acc, smem, p, m0, n0, tile_m);
}

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
__global__ __launch_bounds__(256) void igemm_nt_split_kernel(NtParams p) {
    constexpr int STAGE = BM * 32, EPI_WORDS = 32 * (BN + 4) + 512;
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    const Quant qz = quant_select(p.a_amax, p.b_amax);
    if (qz.use3) igemm_nt_split_body<3, BM, BN, WM, WN, AMODE, EPI>(p, smem, qz);
    else igemm_nt_split_body<6, BM, BN, WM, WN, AMODE, EPI>(p, smem, qz);
}

// ---- conv3x3 forward / data gradient with halo reuse (split-bf16) ----------------------------------------------------
// The general split kernel above stages the activation tile once per (tap, 16-channel chunk).  For the 3x3 convolution
// this kernel makes the block tile an 8x16-pixel PATCH and stages the patch with its one-pixel halo (10x18 pixels) once
// per 16-channel chunk; the nine taps are nine fragment reads at constant row offsets (LDS row = halo pixel, 112-byte
// stride -> the tap shift is an instruction immediate).  Activation staging (global loads, split arithmetic, LDS
// writes) drops ~6x and there is one barrier per chunk instead of per K-step.  Weights: unchanged (pre-split fragment
// layout straight from global memory, kt = chunk*9 + tap, three register sets).
// W8 = 1: 8 x 8 images (the cfg-S bottleneck, lib/UNet.py:210: 2048 pixels x 512 channels, K = 4608).  A patch is TWO images
// side by side -- tile row r = pixel (y = r >> 4, x = r & 7) of image 2 * tile + ((r & 15) >> 3) -- each with its own zero border
// in LDS (halo rows of 2 x 10 pixels), so the tap offsets stay instruction immediates; only the per-lane base address and the
// row -> pixel map of the epilogue change.  K is accumulated in ranges of eight chunks, each merged into the tile total in
// turn.  16 patches x 8 column tiles are too few blocks, so small grids give every range its own block (p.ksplit = number of
// ranges, p.chunks_per = 8): the blocks park their range in a scratch slab and draw a ticket, the last one adds the ranges in
// the same order and runs the epilogue (statistics, BN-backward hook) on the finished tile -- the same bits as the unsplit form.
template <int NP, int BN, int WM, int WN, int EPI, int SKEW, int W8>
__device__ __forceinline__ void conv3_halo_split_body(const NtParams& p, float* smem, int& sk_last, const Quant qz, const int amax_off) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;               // split terms per operand
    constexpr int BM = 128, PH = 8, PW = 16, HW_ = W8 ? 20 : PW + 2, HROWS = (PH + 2) * HW_;   // 180 (200) halo pixels
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TN == 1, "one 32-column block per wave");
    constexpr int RS = 28;                            // LDS row stride in words (3 terms x 16 bf16 + 16 B pad)
    // halo-row pitch = 18 rows + 32 bytes of skew: a 16-lane group of a ds_read_b128 fragment read spans two patch rows
    // (e.g. pixels 0-3,12-15 of row y and 4-11 of row y+1); with the plain pitch 18*RS two of its 16-byte slots fell
    // into the same banks for ANY RS (SQ_LDS_BANK_CONFLICT 19 % of the kernel's cycles), with the skew all 16 differ
    constexpr int HP = HW_ * RS + SKEW;
    constexpr int STAGE = (PH + 2) * HP;
    constexpr int NLD = (HROWS * 4 + 255) / 256;      // staging float4 per thread and chunk
    constexpr int EPI_WORDS = 32 * (BN + 4) + 512;
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    const int lb0 = xcd_remap(blockIdx.x, gridDim.x);
    const int ksp = W8 && p.ksplit > 1 ? p.ksplit : 1;
    const int ksplit_id = lb0 % ksp, lb = lb0 / ksp;           // the K ranges of one tile are neighbours (same XCD / L2)
    const int cbeg = W8 ? ksplit_id * p.chunks_per : 0;
    const int cend = W8 ? (cbeg + p.chunks_per < p.chunks ? cbeg + p.chunks_per : p.chunks) : p.chunks;
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int n0 = tile_n * BN;
    const int H = p.H, W = p.W;
    const int pxs = W8 ? 1 : W >> 4, pys = W8 ? 1 : H >> 3;   // patches per image row / column
    const int pbx = tile_m % pxs, pby = (tile_m / pxs) % pys, img = W8 ? 2 * tile_m : tile_m / (pxs * pys);
    const int x0 = pbx * PW, y0 = pby * PH;
    const int n_img = p.M / (H * W);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][1], lo[TM];       // hi / lo accumulators (rd_mfma_dev.h: PA6)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = lo[i][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(NP == 3 ? p.Bsplit3 : p.Bsplit, NP == 3 ? p.b_bytes3 : p.b_bytes);
    // staging tasks: element e = t + 256 k -> halo pixel e >> 2, 16-byte quarter e & 3 of its 64-byte channel chunk
    unsigned s_off[NLD];
    int s_lds[NLD];
    const int c4 = t & 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        // halo pixel of this task: aligned blocks of eight pixels are dealt out so that a 16-lane group of a ds_write_b64 gets
        // pixels p, p+2, p+4, p+6 (bank windows 0 / 24 / 16 / 8 at the 28-word pitch: disjoint) instead of p .. p+3 (windows
        // overlapping by half: a 2-way conflict on every staging write, the whole of this kernel's SQ_LDS_BANK_CONFLICT in r04);
        // blocks that straddle two halo rows (7 of 23) keep one overlapping pair, the ragged last block its natural order
        const int e = t + 256 * k, hr0 = e >> 2;
        const int hr = (hr0 | 7) < HROWS ? ((hr0 & ~7) | ((hr0 & 3) << 1) | ((hr0 >> 2) & 1)) : hr0;
        const int hy = hr / HW_, hx = hr - hy * HW_;
        const int sub = W8 ? hx / 10 : 0;                     // W8: which of the patch's two images
        const int y = y0 - 1 + hy, x = W8 ? hx - sub * 10 - 1 : x0 - 1 + hx;
        const bool ok = hr < HROWS && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && img + sub < n_img;
        s_off[k] = ok ? (unsigned)(((((long)(img + sub) * H + y) * W + x) * p.Cin + c4 * 4) * 4) : kOOB;
        s_lds[k] = hr < HROWS ? hy * HP + hx * RS + c4 * 2 : -1;
    }
    const int nb = (n0 >> 5) + wn;
    const unsigned b_off = (nb * 32 < p.N) ? (unsigned)(((long)nb * p.nk * NT) * 1024 + lane * 16) : kOOB;

    auto load_halo = [&](int chunk, float4 (&rh)[NLD]) {
        const bool cok = chunk < cend && chunk * SK + c4 * 4 < p.Cin;
#pragma unroll
        for (int k = 0; k < NLD; ++k) rh[k] = buf_load4(rsA, cok ? s_off[k] : kOOB, (unsigned)(chunk * SK * 4));
    };
    auto store_halo = [&](float* stage, const float4 (&rh)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (s_lds[k] < 0) continue;
            uint2 ph, pm, pl;
            split_pack4<NP>(rh[k], qz.sa, ph, pm, pl);
            float* row = stage + s_lds[k];
            *reinterpret_cast<uint2*>(row) = ph;
            *reinterpret_cast<uint2*>(row + 8) = pm;
            if (kterm3<NP>()) *reinterpret_cast<uint2*>(row + 16) = pl;
        }
    };
    auto load_b = [&](int kt, uint4 (&rb)[3]) {
        const unsigned voff = kt < cend * 9 ? b_off : kOOB;
#pragma unroll
        for (int q = 0; q < NT; ++q) rb[q] = buf_load4u(rsB, voff, (unsigned)((kt * NT + q) * 1024));
    };

    const int lrow = lane & 31, half = lane >> 5;
    int a_rd[TM];       // word address of (top-left tap, term 0) of this lane's pixel in a stage
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;            // tile row -> patch pixel (row >> 4, row & 15)
        a_rd[i] = (row >> 4) * HP + ((row & 15) + (W8 ? 2 * ((row & 15) >> 3) : 0)) * RS + half * 4;
    }
    FR af[TM][3];

    float* stage_cur = smem;
    float* stage_nxt = smem + STAGE;
    float4 rh[NLD];
    uint4 b0[3], b1[3], b2[3];
    load_halo(cbeg, rh);
    load_b(cbeg * 9, b0);
    load_b(cbeg * 9 + 1, b1);
    store_halo(stage_cur, rh);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) af[i][q] = *reinterpret_cast<const FR*>(stage_cur + a_rd[i] + q * 8);

    // tap step: MFMAs of (chunk, TAP) with the fragments in af / bcur; prefetch the weights two steps ahead; reload af
    // for the next step right after its last use (next tap of this chunk, or tap 0 of the next chunk's stage)
    auto tap_step = [&](int kt, auto tap_c, uint4 (&bcur)[3], uint4 (&bnew)[3]) {
        constexpr int TAP = decltype(tap_c)::value;
        constexpr int NEXT = TAP == 8 ? 0 : (TAP + 1) / 3 * HP + (TAP + 1) % 3 * RS;
        load_b(kt + 2, bnew);
        FR bf[3];
#pragma unroll
        for (int q = 0; q < NT; ++q) bf[q] = __builtin_bit_cast(FR, bcur[q]);
        const float* nstage = TAP == 8 ? stage_nxt : stage_cur;
        // Raised wave priority over the MFMA cluster of a tap: with two blocks per CU there are two waves per SIMD, and the
        // sibling's staging VALU / LDS instructions otherwise win issue slots between this wave's MFMAs.  r03, interleaved:
        // nine layers alone 2.74 -> 2.65 ms, end to end 2747.7 / 2758.3 / 2751.0 -> 2817.3 / 2821.7 / 2827.1 tiles/s (+2.5 %).
        // Priority 3 instead of 1: same.  Priority for the WHOLE kernel (set once): no gain -- it is the toggling that
        // orders the two waves, not a priority over other kernels.  The same two lines in the weight-gradient kernels (one
        // wave per SIMD at 256 blocks, on the second stream) take the gain away again; in the transposed-convolution and
        // generic NT kernels they cost 1.2 % (profiles/r03_notes.md section 7).
        __builtin_amdgcn_s_setprio(1);
        // products in the order of PA6 / PB6 over ALL row blocks; every A term is re-loaded for the next tap right after its
        // last use (term 2 after the first product, term 1 after the fourth, term 0 after the sixth), and the scheduler is
        // pinned to that interleaving (sched_group_barrier: 0x008 = MFMA, 0x100 = LDS read) -- left alone, hipcc sinks all
        // 3 * TM fragment reads behind the ~19th MFMA and the next tap starts with four lgkmcnt waits
        auto rd = [&](int q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const FR*>(nstage + a_rd[i] + NEXT + q * 8);
        };
#if RD_TAP_FINE
        if constexpr (NP == 3) {
            // every fragment is re-read for the next tap right after ITS last use, pinned by scheduling fences
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                lo[i] = mfma16<NP>(af[i][1], bf[0], lo[i]);
                __builtin_amdgcn_sched_barrier(0);
                af[i][1] = *reinterpret_cast<const FR*>(nstage + a_rd[i] + NEXT + 8);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][0], bf[1], lo[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i][0] = mfma16<NP>(af[i][0], bf[0], acc[i][0]);
                __builtin_amdgcn_sched_barrier(0);
                af[i][0] = *reinterpret_cast<const FR*>(nstage + a_rd[i] + NEXT);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#endif
        if constexpr (NP == 3) {
            // three products (a2 b1, a1 b2, a1 b1): term 2 does not exist
#pragma unroll
            for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][1], bf[0], lo[i]);
            rd(1);
#pragma unroll
            for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][0], bf[1], lo[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = mfma16<NP>(af[i][0], bf[0], acc[i][0]);
            rd(0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][2], bf[0], lo[i]);
            rd(2);
#pragma unroll
            for (int t6 = 1; t6 < 4; ++t6)
#pragma unroll
                for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][PA6[t6]], bf[PB6[t6]], lo[i]);
            rd(1);
#pragma unroll
            for (int i = 0; i < TM; ++i) lo[i] = mfma16<NP>(af[i][0], bf[1], lo[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = mfma16<NP>(af[i][0], bf[0], acc[i][0]);
            rd(0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // W8: K is accumulated in RANGES of eight chunks (128 channels), each merged into `tot` in turn -- whether the ranges of a
    // tile run in one block (ksplit = 1) or one per block (ksplit = number of ranges, added in the same order by the fix-up),
    // so the result does not depend on the batch size that decides between the two
    f32x16 tot[W8 ? TM : 1];
#pragma unroll
    for (int i = 0; i < (W8 ? TM : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;
    for (int c0 = cbeg; c0 < cend; c0 += W8 ? 8 : (cend - cbeg)) {
    const int c1 = W8 ? c0 + 8 : cend;                        // W8: always eight steps (chunks past cend load zeros)
    for (int chunk = c0; chunk < c1; ++chunk) {
        const int kt = chunk * 9;
        load_halo(chunk + 1, rh);                             // next chunk's patch: in flight during taps 0..4
        tap_step(kt + 0, std::integral_constant<int, 0>(), b0, b2);
        tap_step(kt + 1, std::integral_constant<int, 1>(), b1, b0);
        tap_step(kt + 2, std::integral_constant<int, 2>(), b2, b1);
        tap_step(kt + 3, std::integral_constant<int, 3>(), b0, b2);
        tap_step(kt + 4, std::integral_constant<int, 4>(), b1, b0);
        store_halo(stage_nxt, rh);                            // the other stage was last read before the previous barrier
        tap_step(kt + 5, std::integral_constant<int, 5>(), b2, b1);
        tap_step(kt + 6, std::integral_constant<int, 6>(), b0, b2);
        tap_step(kt + 7, std::integral_constant<int, 7>(), b1, b0);
        __syncthreads();                                      // next stage complete before tap 8 prefetches from it
        tap_step(kt + 8, std::integral_constant<int, 8>(), b2, b1);
        float* tmp = stage_cur; stage_cur = stage_nxt; stage_nxt = tmp;
    }
    if (W8) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tot[i][r] += merge_q<NP>(acc[i][0][r], lo[i][r], qz.dexp);
                acc[i][0][r] = lo[i][r] = 0.f;
            }
    }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = W8 ? tot[i][r] : merge_q<NP>(acc[i][0][r], lo[i][r], qz.dexp);
    if (ksp > 1) {
        // All traffic through the slab is agent-scope (sc1) relaxed atomics -- coherent across the XCDs' L2s by themselves --
        // ordered by completion: the partial stores are acknowledged (vmcnt = 0) before the block's ticket is drawn.  (The
        // portable spelling, __threadfence() on both sides, costs a buffer_wbl2 = a write-back of the XCD's whole L2 per block.)
        float* base = p.sk_slab + ((size_t)lb * ksp) * (size_t)(BM * BN);
        float* mine = base + (size_t)ksplit_id * (BM * BN) + (wave * TM) * 1024 + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __hip_atomic_store(mine + (i * 16 + r) * 64, acc[i][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): every partial of this wave has reached the coherence point
        __syncthreads();
        if (t == 0) {
            const unsigned prev = __hip_atomic_fetch_add(p.sk_ticket + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sk_last = prev == (unsigned)(ksp - 1);
            if (sk_last) __hip_atomic_store(p.sk_ticket + lb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
        }
        __syncthreads();
        if (!sk_last) return;
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        for (int sp = 0; sp < ksp; ++sp) {
            const float* part = base + (size_t)sp * (BM * BN) + (wave * TM) * 1024 + lane;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][0][r] += __hip_atomic_load(part + (i * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int m0 = ((img * H + y0) * W) + x0;
    const long pool_base = ((long)img * (H >> 1) + (y0 >> 1)) * (W >> 1) + (x0 >> 1);
    nt_epilogue<BM, BN, WM, WN, EPI, SMEM, 1>(acc, smem, p, m0, n0, tile_m, pool_base, amax_off);
}

template <int BN, int WM, int WN, int EPI, int SKEW = 8, int W8 = 0>
__global__ __launch_bounds__(256) void conv3_halo_split_kernel(NtParams p) {
    constexpr int HW_ = W8 ? 20 : 18, HP = HW_ * 28 + SKEW, STAGE = 10 * HP, EPI_WORDS = 32 * (BN + 4) + 512;
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    __shared__ int sk_last;
    // per-image magnitude slots (inference; never with W8, whose patches hold two images: launch_nt): this block's image, as the
    // body derives it from the block index
    int amax_off = 0;
    if (!W8 && p.amax_img_stride) {
        const int lb = xcd_remap(blockIdx.x, gridDim.x);
        const int tile_m = lb / p.tiles_n;
        amax_off = (tile_m / ((p.W >> 4) * (p.H >> 3))) * p.amax_img_stride;
    }
    const Quant qz = quant_select(p.a_amax ? p.a_amax + amax_off : nullptr, p.b_amax);
    if (qz.use3) conv3_halo_split_body<3, BN, WM, WN, EPI, SKEW, W8>(p, smem, sk_last, qz, amax_off);
    else conv3_halo_split_body<6, BN, WM, WN, EPI, SKEW, W8>(p, smem, sk_last, qz, amax_off);
}

// one 16-byte fragment piece per term of row n, K-step kt, k-half g: the six-product form (three bf16 terms) at `out`, and --
// out3 != nullptr -- the three-product form (two fp16 terms of s3 * v, rd_mfma_dev.h) at `out3`
__device__ __forceinline__ void write_split_piece(uint4* out, long n, int nk, int kt, int g, const float (&v)[8], uint4* out3 = nullptr,
                                                  float s3 = 1.f) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
    const long blk = ((n >> 5) * nk + kt) * 3;
    const int lane = g * 32 + (int)(n & 31);
    const unsigned sel = 0x07060302u;
    out[blk * 64 + lane] = make_uint4(__builtin_amdgcn_perm(h[1], h[0], sel), __builtin_amdgcn_perm(h[3], h[2], sel),
                                      __builtin_amdgcn_perm(h[5], h[4], sel), __builtin_amdgcn_perm(h[7], h[6], sel));
    out[(blk + 1) * 64 + lane] = make_uint4(__builtin_amdgcn_perm(m[1], m[0], sel), __builtin_amdgcn_perm(m[3], m[2], sel),
                                            __builtin_amdgcn_perm(m[5], m[4], sel), __builtin_amdgcn_perm(m[7], m[6], sel));
    out[(blk + 2) * 64 + lane] = make_uint4(__builtin_amdgcn_perm(l[1], l[0], sel), __builtin_amdgcn_perm(l[3], l[2], sel),
                                            __builtin_amdgcn_perm(l[5], l[4], sel), __builtin_amdgcn_perm(l[7], l[6], sel));
    if (out3) {
        uint4 ph, pm;
        split2h_pair_v(v[0], v[1], s3, ph.x, pm.x);
        split2h_pair_v(v[2], v[3], s3, ph.y, pm.y);
        split2h_pair_v(v[4], v[5], s3, ph.z, pm.z);
        split2h_pair_v(v[6], v[7], s3, ph.w, pm.w);
        const long blk3 = ((n >> 5) * nk + kt) * 2;
        out3[blk3 * 64 + lane] = ph;
        out3[(blk3 + 1) * 64 + lane] = pm;
    }
}
// three-product form of a packed operand: follows the six-product form (rows32 * nk * 96 bytes = rows32 * nk * 6 uint4)
__device__ __forceinline__ uint4* form3_of(uint4* out6, long rows, int nk) { return out6 + rows32_of_dev(rows) * nk * 6; }
// scale of a weight tensor's three-product form from its magnitude slot (the kernel that filled it ran just before)
__device__ __forceinline__ float pack_scale(const unsigned* amax) {       // any thread, any tensor
    return __uint_as_float((unsigned)scale_bexp(amax_read(amax)) << 23);
}
__device__ __forceinline__ float pack_scale_wave(const unsigned* amax) {  // every lane of the wave asks for the same tensor
    return __uint_as_float((unsigned)scale_bexp(amax_read_wave(amax)) << 23);
}

// max |w| of weight tensors into their magnitude slots: block b of an item strides over the item's elements
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ w, long n, unsigned* slot, const float* __restrict__ row_scale,
                                                   long row_len) {
    float m = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        m = amax_acc(m, row_scale ? w[e] * row_scale[e / row_len] : w[e]);
    amax_commit(slot, m);
}

// fp32 GEMM-layout B[N][K] (K = taps*Cin, tap-major) -> split-bf16 fragment layout
//   [row block nb = n/32][kt][term q][lane = 32*(j/8) + n%32][8 bf16: k = 8*(j/8) .. +7]      (16 bytes per lane)
// kt = chunk*taps + tap, j = channel within the 16-channel chunk; rows beyond N and channels beyond Cin are zero.
__global__ void split_pack_kernel(const float* __restrict__ B, uint4* __restrict__ out, int N, int K, int Cin,
                                  int taps, int nk, const unsigned* amax) {
    // one thread per (row n, K-step kt, k-half g): 8 consecutive channels -> one 16-byte fragment piece per term
    const long rows32 = (long)((N + 31) / 32) * 32;
    const long total = rows32 * nk * 2;
    uint4* out3 = amax ? form3_of(out, N, nk) : nullptr;
    const float s3 = amax ? pack_scale_wave(amax) : 1.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int g = (int)(e & 1);
        const int kt = (int)((e >> 1) % nk);
        const long n = (e >> 1) / nk;
        const int chunk = kt / taps, tap = kt - chunk * taps, ci0 = chunk * SK + g * 8;
        float v[8];
        const float* src = B + n * K + (long)tap * Cin + ci0;
        if (n < N && ci0 + 8 <= Cin && (((size_t)src & 15) == 0)) {
            const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (n < N && ci0 + j < Cin) ? src[j] : 0.f;
        }
        write_split_piece(out, n, nk, kt, g, v, out3, s3);
    }
}

// conv3x3 weights straight from the torch layout w[co][ci][3][3] into BOTH split-bf16 fragment tensors in one launch
// (split mode never reads the fp32 GEMM layouts of a 3x3 convolution):
//   pieces [0, Tf)      forward operand   rows n = co, k = (tap, ci):        w[co][ci0+j][tap]
//   pieces [Tf, Tf+Td)  data-gradient operand rows n = ci, k = (tap', co):   w[co0+j][ci][8-tap']
// one thread per 16-byte fragment piece (row n, K-step kt = chunk*9 + tap, k-half g); gathered, cached loads.
__global__ void split_pack_conv3x3_kernel(const float* __restrict__ w, uint4* __restrict__ outf, uint4* __restrict__ outd,
                                          int cout, int cin, const float* __restrict__ row_scale, const unsigned* amax) {
    const int nkf = 9 * ((cin + SK - 1) / SK), nkd = 9 * ((cout + SK - 1) / SK);
    const long rf = (long)((cout + 31) / 32) * 32, rd_ = (long)((cin + 31) / 32) * 32;
    const long Tf = rf * nkf * 2, Td = outd ? rd_ * nkd * 2 : 0;
    const float s3 = amax ? pack_scale_wave(amax) : 1.f;
    for (long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x; e0 < Tf + Td; e0 += (long)gridDim.x * blockDim.x) {
        const bool fwd = e0 < Tf;
        const long e = fwd ? e0 : e0 - Tf;
        const int nk = fwd ? nkf : nkd, N = fwd ? cout : cin, Kc = fwd ? cin : cout;
        const int g = (int)(e & 1);
        const int kt = (int)((e >> 1) % nk);
        const long n = (e >> 1) / nk;
        const int chunk = kt / 9, tap = kt - chunk * 9, c0 = chunk * SK + g * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            float x = 0.f;
            if (n < N && c < Kc) x = fwd ? w[((long)n * cin + c) * 9 + tap] : w[((long)c * cin + n) * 9 + (8 - tap)];
            if (fwd && row_scale && n < N) x *= row_scale[n];       // eval-mode BatchNorm folded into the forward operand
            v[j] = x;
        }
        uint4* out = fwd ? outf : outd;
        write_split_piece(out, n, nk, kt, g, v, amax ? form3_of(out, N, nk) : nullptr, s3);
    }
}

// ---- every packed operand of the network in ONE launch ------------------------------------------------------------------
// The per-layer pack kernels were ~30 launches per step (9 conv3x3 layers, 5 transposed convolutions x 4 launches) of 4-60 us
// each on the second stream.  Here a device-resident item table describes the layers and one grid covers all 16-byte
// fragment pieces: thread -> (item, operand, row n, K-step kt, k-half g) -> gathers its 8 weights straight from the torch
// layouts, splits them and writes the three fragment pieces.  Items (8 x int64 each): w pointer, forward-operand buffer,
// data-gradient-operand buffer, kind (0 conv3x3, 1 convT2x2), Cout, Cin, first piece, write-f32-layout flag.
struct PackItem {
    long long w, outf, outd, kind, cout, cin, begin, f32, tile_begin, amax;   // amax: magnitude slot of w (0: six-product form only)
};

__global__ __launch_bounds__(256) void pack_all_kernel(const PackItem* __restrict__ items, int n_items, long total) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        int it = 0;
        while (it + 1 < n_items && items[it + 1].begin <= p) ++it;
        const PackItem I = items[it];
        const float* __restrict__ w = reinterpret_cast<const float*>(I.w);
        const int cout = (int)I.cout, cin = (int)I.cin;
        long e = p - I.begin;
        float v[8];
        const unsigned* am = reinterpret_cast<const unsigned*>(I.amax);
        const float s3 = am ? pack_scale(am) : 1.f;
        auto put = [&](long long outp, long rows, long n, int nk, int kt, int g) {
            uint4* out = reinterpret_cast<uint4*>(outp);
            write_split_piece(out, n, nk, kt, g, v, am ? form3_of(out, rows, nk) : nullptr, s3);
        };
        if (I.kind == 0) {
            // conv3x3 w[co][ci][3][3]: forward rows n = co, k = (tap, ci); data gradient rows n = ci, k = (8 - tap, co)
            const int nkf = 9 * ((cin + SK - 1) / SK), nkd = 9 * ((cout + SK - 1) / SK);
            const long Tf = rows32_of_dev(cout) * nkf * 2;
            const bool fwd = e < Tf;
            if (!fwd) e -= Tf;
            const int nk = fwd ? nkf : nkd, N = fwd ? cout : cin, Kc = fwd ? cin : cout;
            const int g = (int)(e & 1), kt = (int)((e >> 1) % nk);
            const long n = (e >> 1) / nk;
            const int chunk = kt / 9, tap = kt - chunk * 9, c0 = chunk * SK + g * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                v[j] = (n < N && c < Kc) ? (fwd ? w[((long)n * cin + c) * 9 + tap] : w[((long)c * cin + n) * 9 + (8 - tap)]) : 0.f;
            }
            put(fwd ? I.outf : I.outd, N, n, nk, kt, g);
        } else {
            // convT2x2 w[ci][co][2][2]: forward rows n = (ab, co), k = ci; data gradient rows n = ci, k = (ab, co);
            // optional third segment: the fp32 forward operand wtf[(ab, co)][ci] (exact-f32 kernel of the short-K levels)
            const int nkf = (cin + SK - 1) / SK, nkd = 4 * ((cout + SK - 1) / SK);
            const long Tf = rows32_of_dev(4L * cout) * nkf * 2, Td = rows32_of_dev(cin) * nkd * 2;
            if (e < Tf) {
                const int g = (int)(e & 1), kt = (int)((e >> 1) % nkf);
                const long n = (e >> 1) / nkf;
                const int ab = (int)(n / cout), co = (int)(n - (long)ab * cout), c0 = kt * SK + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (n < 4L * cout && c0 + j < cin) ? w[((long)(c0 + j) * cout + co) * 4 + ab] : 0.f;
                put(I.outf, 4L * cout, n, nkf, kt, g);
            } else if (e < Tf + Td) {
                e -= Tf;
                const int g = (int)(e & 1), kt = (int)((e >> 1) % nkd);
                const long n = (e >> 1) / nkd;
                const int chunk = kt / 4, ab = kt - chunk * 4, c0 = chunk * SK + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (n < cin && c0 + j < cout) ? w[((long)n * cout + c0 + j) * 4 + ab] : 0.f;
                put(I.outd, cin, n, nkd, kt, g);
            } else {
                e -= Tf + Td;                    // 8 consecutive elements of wtf[(ab*Cout + co)][ci]
                float* wtf = reinterpret_cast<float*>(I.f32);
                const long tot = 4L * cout * cin;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long q = e * 8 + j;
                    if (q < tot) {
                        const int ci = (int)(q % cin), co = (int)((q / cin) % cout), ab = (int)(q / ((long)cin * cout));
                        wtf[q] = w[((long)ci * cout + co) * 4 + ab];
                    }
                }
            }
        }
    }
}


// magnitude slots of every item of a fused pack (first pass of rd_pack_weights_fused in three-product mode): AMAX_PARTS blocks per
// item, four elements per thread and trip in flight (r06: 32 blocks per item took 126 us -- a 9.4 MB layer over 32 blocks is
// latency-bound -- ahead of the pack itself on the stream the first 3 x 3 convolution waits for)
constexpr int AMAX_PARTS = 256;
__global__ __launch_bounds__(256) void pack_items_amax_kernel(const PackItem* __restrict__ items, int n_items) {
    const int it = blockIdx.x / AMAX_PARTS, part = blockIdx.x % AMAX_PARTS;
    const PackItem I = items[it];
    if (!I.amax) return;
    const float* __restrict__ w = reinterpret_cast<const float*>(I.w);
    const long n = (long)I.cout * I.cin * (I.kind == 0 ? 9 : 4);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    const long stride = (long)AMAX_PARTS * 256;
    long e = (long)part * 256 + threadIdx.x;
    for (; e + 3 * stride < n; e += 4 * stride) {
        const float a = w[e], b = w[e + stride], c = w[e + 2 * stride], d = w[e + 3 * stride];
        m0 = amax_acc(m0, a); m1 = amax_acc(m1, b); m2 = amax_acc(m2, c); m3 = amax_acc(m3, d);
    }
    for (; e < n; e += stride) m0 = amax_acc(m0, w[e]);
    amax_commit(reinterpret_cast<unsigned*>(I.amax), fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
}

// ---- tile packer: layers whose channel counts are multiples of 32 (every MFMA layer of cfg-S / cfg-M) -----------------------
// pack_all_kernel gathers every 16-byte fragment piece straight from the torch layouts: 4-byte loads 36 B (conv3x3) or 16 B
// (convT) apart, 2.2x the algorithmic HBM traffic in sector over-fetch (560 MB per step measured) and 0.22 ms at 14 % of the
// HBM rate -- on the second stream, where it slowed the first convolution of the step from 0.16 to 0.23 ms.  Here one block
// takes a [32 x 32 x taps] tile of a weight tensor, loads it with coalesced 16-byte loads into LDS once, and builds BOTH
// operands (forward and data gradient) from it: every fragment piece row of 64 lanes is one contiguous 1 KB store.
//   conv3x3 w[co][ci][9]:  tile = 32 co rows x (32 ci x 9 taps = 288 contiguous floats), LDS row stride 289
//   convT   w[ci][co][4]:  tile = 32 ci rows x (32 co x 4 = 128 contiguous floats), de-interleaved into four (a, b) planes
//                          [ci][co] of row stride 33 (conflict-free along either index)
__global__ __launch_bounds__(256) void pack_tiles_kernel(const PackItem* __restrict__ items, int n_items, long total_tiles) {
    __shared__ float tile[32 * 289];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = lane >> 5, nl = lane & 31;
    for (long T = blockIdx.x; T < total_tiles; T += gridDim.x) {
        int it = 0;
        while (it + 1 < n_items && items[it + 1].tile_begin <= T) ++it;
        const PackItem I = items[it];
        const float* __restrict__ w = reinterpret_cast<const float*>(I.w);
        const int cout = (int)I.cout, cin = (int)I.cin;
        const int lt = (int)(T - I.tile_begin);
        float v[8];
        const unsigned* am = reinterpret_cast<const unsigned*>(I.amax);
        const float s3 = am ? pack_scale_wave(am) : 1.f;      // one item per block
        auto put = [&](long long outp, long rows, long n, int nk, int kt) {
            uint4* out = reinterpret_cast<uint4*>(outp);
            write_split_piece(out, n, nk, kt, g, v, am ? form3_of(out, rows, nk) : nullptr, s3);
        };
        if (I.kind == 0) {
            const int tiles_ci = cin >> 5, co0 = (lt / tiles_ci) * 32, ci0 = (lt % tiles_ci) * 32;
            // (a parameter is a view into the flat buffer at an arbitrary float offset -- e.g. behind a 1-element PReLU slope:
            // 16-byte loads only when the tensor happens to be 16-byte aligned)
            const bool al16 = ((size_t)w & 15) == 0;
            for (int e = t; e < 32 * 72; e += 256) {
                const int row = e / 72, q = e - row * 72;
                const float* src = w + ((long)(co0 + row) * cin + ci0) * 9 + q * 4;
                float* d = tile + row * 289 + q * 4;
                if (al16) {
                    const float4 x = *reinterpret_cast<const float4*>(src);
                    d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
                } else {
                    d[0] = src[0]; d[1] = src[1]; d[2] = src[2]; d[3] = src[3];
                }
            }
            __syncthreads();
            const int nkf = 9 * (cin >> 4), nkd = 9 * (cout >> 4);
            for (int task = wave; task < 36; task += 4) {          // (operand, 16-channel chunk, tap): one piece row of 64 lanes
                const int dg = task / 18, rest = task - dg * 18, chunk_l = rest / 9, tap = rest - chunk_l * 9;
                if (dg == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = tile[nl * 289 + (chunk_l * 16 + g * 8 + j) * 9 + tap];
                    put(I.outf, cout, co0 + nl, nkf, ((ci0 >> 4) + chunk_l) * 9 + tap);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = tile[(chunk_l * 16 + g * 8 + j) * 289 + nl * 9 + (8 - tap)];
                    put(I.outd, cin, ci0 + nl, nkd, ((co0 >> 4) + chunk_l) * 9 + tap);
                }
            }
        } else {
            const int tiles_co = cout >> 5, ci0 = (lt / tiles_co) * 32, co0 = (lt % tiles_co) * 32;
            const bool al16 = ((size_t)w & 15) == 0;
            for (int e = t; e < 32 * 32; e += 256) {
                const int row = e >> 5, c = e & 31;
                const float* src = w + ((long)(ci0 + row) * cout + co0 + c) * 4;
                float* d = tile + row * 33 + c;
                if (al16) {
                    const float4 x = *reinterpret_cast<const float4*>(src);
                    d[0] = x.x; d[1056] = x.y; d[2 * 1056] = x.z; d[3 * 1056] = x.w;
                } else {
                    d[0] = src[0]; d[1056] = src[1]; d[2 * 1056] = src[2]; d[3 * 1056] = src[3];
                }
            }
            __syncthreads();
            const int nkf = cin >> 4, nkd = 4 * (cout >> 4);
            for (int task = wave; task < 16; task += 4) {          // (operand, (a, b) plane, 16-channel chunk)
                const int dg = task >> 3, ab = (task >> 1) & 3, chunk_l = task & 1;
                const float* pl = tile + ab * 1056;
                if (dg == 0) {          // forward rows n = (ab, co), k = ci
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = pl[(chunk_l * 16 + g * 8 + j) * 33 + nl];
                    put(I.outf, 4L * cout, (long)ab * cout + co0 + nl, nkf, (ci0 >> 4) + chunk_l);
                } else {                // data gradient rows n = ci, k = (co chunk, ab)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = pl[nl * 33 + chunk_l * 16 + g * 8 + j];
                    put(I.outd, cin, ci0 + nl, nkd, ((co0 >> 4) + chunk_l) * 4 + ab);
                }
            }
            if (I.f32) {                // fp32 forward operand wtf[(ab, co)][ci] of the short-K levels (exact-f32 NT kernel)
                float* wtf = reinterpret_cast<float*>(I.f32);
                for (int e = t; e < 4 * 32 * 32; e += 256) {
                    const int ab = e >> 10, co_l = (e >> 5) & 31, ci_l = e & 31;
                    wtf[((long)ab * cout + co0 + co_l) * cin + ci0 + ci_l] = tile[ab * 1056 + ci_l * 33 + co_l];
                }
            }
        }
        __syncthreads();
    }
}

// ---- packed weight buffers -------------------------------------------------------------------------------
// One opaque buffer per GEMM operand B[rows][K = taps*Cin]:  [rows*K floats, fp32 GEMM layout] [pad to 16 B]
// [rows * nk16 * 96 bytes, split-bf16 layout], nk16 = taps * ceil(Cin/16).  rd_packed_weight_bytes() sizes it.
static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
static inline int nk16_of(int taps, int cin) { return taps * cdiv(cin, SK); }
static inline size_t packed_f32_bytes(long rows, int taps, int cin) { return align16((size_t)rows * taps * cin * 4); }
static inline long rows32_of(long rows) { return (rows + 31) / 32 * 32; }
static inline size_t packed_bytes(long rows, int taps, int cin) {
    return packed_f32_bytes(rows, taps, cin) + (size_t)rows32_of(rows) * nk16_of(taps, cin) * (SROWB + SROWB3);
}

static int grid_for(long total, int block, int cap);
// max |w| of a weight tensor -> its magnitude slot (zeroed by the caller); row_scale: the folded operand's rows
static void launch_amax(const float* w, long n, unsigned* slot, const float* row_scale, long row_len, hipStream_t s) {
    RD_LAUNCH(amax_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, s, w, n, slot, row_scale, row_len);
}

static int split_pack(const float* b_f32, long rows, int taps, int cin, hipStream_t s, const unsigned* amax = nullptr) {
    uint4* out = (uint4*)((char*)b_f32 + packed_f32_bytes(rows, taps, cin));
    const int nk = nk16_of(taps, cin);
    const long total = rows32_of(rows) * nk * 2;
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    RD_LAUNCH(split_pack_kernel, dim3((int)g), dim3(256), 0, s, b_f32, out, (int)rows, taps * cin, cin, taps, nk, amax);
    RD_LAUNCH_CHECK("split_pack");
    return RD_OK;
}

static inline void set_quant(NtParams& p, const QuantArgs& q) {
    p.a_amax = q.a; p.b_amax = q.b; p.out_amax = q.out; p.pool_amax = q.out2; p.amax_img_stride = q.img_stride;
}
// per-image slots reach a kernel that cannot index them (generic row tiles, the two-images-per-patch form): the launch goes
// without -- six-product body, nothing committed (the untouched slots read as "unknown" downstream)
static inline void drop_img_quant(NtParams& p) {
    if (p.amax_img_stride) { p.a_amax = p.b_amax = nullptr; p.out_amax = p.pool_amax = nullptr; p.amax_img_stride = 0; }
}

template <int AMODE, int EPI>
static int launch_nt(NtParams p, hipStream_t s, const char* cls, int* tiles_m_out = nullptr) {
    const long flops = 2L * p.M * p.N * p.K;
    const double bytes = 4.0 * ((double)p.M * p.Cin * (AMODE == A_UP2 ? 4 : 1) + (double)p.N * p.K + (double)p.M * p.N);
    const int taps = p.K / p.Cin;
    // short-K transposed convolutions are bound by their scatter epilogue, not by the contraction: exact-f32 MFMA kernel
    const int split = mfma_split() && !(EPI == EPI_CONVT && p.K <= 128);
    p.taps = taps;
    p.chunks = cdiv(p.Cin, split ? SK : 32);
    p.nk = taps * p.chunks;
    p.vec = (p.N % 4 == 0) && (EPI != EPI_CONVT || p.Cout % 4 == 0);
    const double a_bytes = 4.0 * p.M * p.Cin * (AMODE == A_UP2 ? 4 : 1);
    const double b_bytes = split ? (double)rows32_of(p.N) * p.nk * SROWB : 4.0 * p.N * p.K;
    if (a_bytes >= 4294967040.0 || b_bytes >= 4294967040.0) {
        set_error("%s: operand larger than the 4 GiB buffer-descriptor range (A %.0f B, B %.0f B)", cls, a_bytes, b_bytes);
        return RD_ERR_ARG;
    }
    p.a_bytes = (unsigned)a_bytes;
    p.b_bytes = (unsigned)b_bytes;
    p.Bsplit = (const char*)p.B + packed_f32_bytes(p.N, taps, p.Cin);
    p.Bsplit3 = (const char*)p.Bsplit + (size_t)rows32_of(p.N) * p.nk * SROWB;
    p.b_bytes3 = (unsigned)((double)rows32_of(p.N) * p.nk * SROWB3);
    if (!split || mfma_products() != 3 || !p.a_amax || !p.b_amax) p.a_amax = p.b_amax = nullptr;    // six products
    const int force = tune(TUNE_NT_TILE);
    const int tiles_128x64 = cdiv(p.M, 128) * cdiv(p.N, 64);
    int cfg;
    if (split) {
        // split kernel: two 128x128 blocks (57 KB LDS each) per CU; measured per layer with scripts/bench_layers.py
        const bool halo_shape = AMODE == A_CONV3 && EPI == EPI_STORE && p.W % 16 == 0 && p.H % 8 == 0;
        // (the pooling epilogue exists in the patch kernels only: small batches take them too.  So do launches with per-image
        // magnitude slots, which only the patch kernels index: whether a layer runs on three or six products must follow from
        // its SHAPE alone -- a ragged last batch of a sweep must not change a tile's bits)
        if (halo_shape && (tiles_128x64 >= 512 || p.pool_out || p.amax_img_stride)) cfg = (p.N >= 128 && tiles_128x64 >= 1024) ? 0 : 1;   // 16x16 levels: 128x64 patches
        else if (p.M < 128 || tiles_128x64 < 1024) cfg = 2;
        else if (EPI == EPI_CONVT && p.K <= 256) cfg = 2;   // scatter epilogue dominates: small tiles keep more in flight
        else if (p.N >= 128) cfg = AMODE == A_UP2 ? 1 : 0;  // gathered operand (convT data gradient): 128x64 runs two waves per SIMD (0.130 -> 0.116 ms)
        else cfg = 1;
    } else {
        // Tile choice (measured per layer on MI355X, scripts/bench_layers.py): with the lean buffer-load loader the
        // launch BALANCE matters more than per-block efficiency -- 1024 tiles of 128x128 at 3 resident blocks/CU run
        // 1.33 rounds (~105 TF) where 2048 tiles of 128x64 run 130+ TF.  128x128 only pays for short-K problems
        // (few K-steps per tile -> amortise the epilogue over a bigger tile); tiny grids take 64x64.
        if (p.M < 128 || tiles_128x64 < 1024) cfg = 2;
        else if (EPI == EPI_CONVT && p.K <= 256) cfg = 2;   // scatter epilogue dominates: small tiles keep more in flight
        else if (p.N >= 128 && p.K <= 640 && cdiv(p.M, 128) * cdiv(p.N, 128) >= 1536) cfg = 0;
        else cfg = 1;
    }
    if (force >= 0 && force <= 2 && !(force == 0 && p.N <= 64)) cfg = force;
    const int halo_force = tune(TUNE_NT_HALO);
    const bool halo = split && AMODE == A_CONV3 && EPI == EPI_STORE && p.W % 16 == 0 && p.H % 8 == 0 && cfg != 2 && halo_force != 0;
    // 8 x 8 images: two-images-per-patch halo kernel; K in ranges of 8 chunks, split over blocks when the grid is small and the
    // stream has the scratch of rd_set_splitk_workspace.  Chosen by the LAYER shape only (never by the pixel count), and the
    // split / unsplit forms add the ranges in the same order: results do not depend on the batch size.
    p.ksplit = 1;
    p.chunks_per = p.chunks;
    if (split && AMODE == A_CONV3 && EPI == EPI_STORE && p.W == 8 && p.H == 8 && !p.pool_out && halo_force != 0 && force < 0 &&
        tune(TUNE_NT_SPLITK) != 0 && p.N % 64 == 0 && p.chunks >= 8) {
        // 128 pixels x 64 channels per block (2 x 2 waves, 172 VGPRs; the 128-column variant needs 270 with the 200-pixel halo)
        const int tiles_m = cdiv(p.M, 128), tiles_n = p.N / 64;
        const long tiles = (long)tiles_m * tiles_n;
        const int ranges = cdiv(p.chunks, 8);
        unsigned* tickets = nullptr;
        float* slab = nullptr;
        int n_tickets = 0;
        size_t slab_bytes = 0;
        if (ranges >= 2 && tiles * ranges <= 1024 && splitk_workspace(s, &tickets, &n_tickets, &slab, &slab_bytes) &&
            tiles <= n_tickets && (size_t)tiles * ranges * 128 * 64 * 4 <= slab_bytes) {
            p.ksplit = ranges;
            p.chunks_per = 8;
            p.sk_slab = slab;
            p.sk_ticket = tickets;
        }
        p.patch = 2;
        drop_img_quant(p);
        p.direct = tune(TUNE_NT_EPI) != 0 && p.N % 32 == 0;
        p.tiles_n = tiles_n;
        if (tiles_m_out) *tiles_m_out = tiles_m;
        char pc8[64];
        snprintf(pc8, sizeof(pc8), "%s|conv3_halo_split<64,w8>", cls);
        ProfScope ps8(s, pc8, (double)flops, bytes, true);
        RD_LAUNCH((conv3_halo_split_kernel<64, 2, 2, EPI_STORE, 8, 1>), dim3((unsigned)(tiles * p.ksplit)), dim3(256), 0, s, p);
        RD_LAUNCH_CHECK(cls);
        return RD_OK;
    }
    if (p.pool_out && !halo) {
        set_error("%s: the pooling epilogue exists in the patch (halo) kernels only", cls);
        return RD_ERR_ARG;
    }
    if (!halo) drop_img_quant(p);
    char pcls[64];   // "<operation>|<kernel symbol>": the kernel symbol is what rocprofv3 reports
    if (halo)
        snprintf(pcls, sizeof(pcls), "%s|conv3_halo_split<%d>", cls, cfg == 0 ? 128 : 64);
    else
        snprintf(pcls, sizeof(pcls), "%s|igemm_nt%s<%s,%d,%d>", cls, split ? "_split" : "",
                 cfg == 0 ? "128,128" : cfg == 1 ? "128,64" : "64,64", AMODE, EPI);
    ProfScope ps(s, pcls, (double)flops, bytes, true);
    if (tiles_m_out) *tiles_m_out = cdiv(p.M, cfg == 2 ? 64 : 128);
    const int bm = cfg == 2 ? 64 : 128, bn = cfg == 0 ? 128 : 64;
    p.tiles_n = cdiv(p.N, bn);
    const int grid = cdiv(p.M, bm) * p.tiles_n;
    if (halo) {
        // 3x3 convolution on 8x16-pixel patches with halo reuse
        p.patch = 1;
        p.direct = tune(TUNE_NT_EPI) != 0 && p.N % 32 == 0;
        const int tiles_m = (p.M >> 7);       // (H/8) * (W/16) patches per image, 128 pixels each
        if (tiles_m_out) *tiles_m_out = tiles_m;
        if (cfg == 0) {
            p.tiles_n = cdiv(p.N, 128);
            if (tune(TUNE_NT_SKEW) == 0)
                RD_LAUNCH((conv3_halo_split_kernel<128, 1, 4, EPI_STORE, 0>), dim3(tiles_m * p.tiles_n), dim3(256), 0, s, p);
            else
                RD_LAUNCH((conv3_halo_split_kernel<128, 1, 4, EPI_STORE>), dim3(tiles_m * p.tiles_n), dim3(256), 0, s, p);
        } else {
            p.tiles_n = cdiv(p.N, 64);
            if (tune(TUNE_NT_SKEW) == 0)
                RD_LAUNCH((conv3_halo_split_kernel<64, 2, 2, EPI_STORE, 0>), dim3(tiles_m * p.tiles_n), dim3(256), 0, s, p);
            else
                RD_LAUNCH((conv3_halo_split_kernel<64, 2, 2, EPI_STORE>), dim3(tiles_m * p.tiles_n), dim3(256), 0, s, p);
        }
        RD_LAUNCH_CHECK(cls);
        return RD_OK;
    }
    // plain row tiles that are all inside M take the register-direct epilogue too (rd_nt.h: the W = 16 case of its row formula).
    // NOTE on batch-size independence: C is the same bits under either epilogue, but the BN-statistics / BN-backward partial
    // rows are summed in a different (fixed) order by the direct and the staged form, and `p.M % bm == 0` depends on the pixel
    // count, i.e. on the batch.  The "an image's result does not depend on the batch it is in" statement of the 8 x 8 split-K
    // path therefore covers C (and everything inference uses); training-mode statistics of these generic row-tile layers can
    // differ in the last bits between batch sizes that flip this predicate -- as they already do between batch sizes that
    // select different kernels.
    p.direct = EPI == EPI_STORE && tune(TUNE_NT_EPI) != 0 && p.M % bm == 0 && p.N % 32 == 0 && !p.shift && !p.pool_out &&
               (long)bm * p.N * 4 < 0x7fffffffL;
    if (split) {
        if (cfg == 2) RD_LAUNCH((igemm_nt_split_kernel<64, 64, 2, 2, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
        else if (cfg == 0) RD_LAUNCH((igemm_nt_split_kernel<128, 128, 1, 4, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
        else RD_LAUNCH((igemm_nt_split_kernel<128, 64, 2, 2, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (cfg == 2) RD_LAUNCH((igemm_nt_kernel<64, 64, 2, 2, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
        else if (cfg == 0) RD_LAUNCH((igemm_nt_kernel<128, 128, 2, 2, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
        else RD_LAUNCH((igemm_nt_kernel<128, 64, 2, 2, AMODE, EPI>), dim3(grid), dim3(256), 0, s, p);
    }
    RD_LAUNCH_CHECK(cls);
    return RD_OK;
}

// ------------------------------------------------------------------------------------------
//  TN split-K kernel (weight gradients)
// ------------------------------------------------------------------------------------------
enum { WA_PLAIN = 0, WA_UP2 = 1 };
enum { WB_PLAIN = 0, WB_CONV3 = 1 };

struct TnParams {
    const float* A;
    const float* B;
    float* slab;  // [splits][M][N]
    int M, N;
    long Kp;       // pixels reduced over
    int lda, ldb;  // channels per pixel of the A / B source tensors
    int H, W;
    PixDiv pd;
    int Cout;  // WA_UP2: channels per (a,b) quadrant of the A columns
    int Cin;   // WB_CONV3: channels per tap of the B columns
    int kchunk;  // pixels per split, multiple of 32
    int tiles_n, tiles_mn;
    unsigned a_bytes, b_bytes;
    const unsigned* a_amax;   // magnitude slots of A / B (both non-null: three-product body, rd_mfma_dev.h); split kernel only
    const unsigned* b_amax;
};

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void wgrad_tn_kernel(TnParams p) {   // (256,3) fits 128 VGPRs but measured slower
    constexpr int BK = 32;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AQ = BM / 4, BQ = BN / 4;          // float4 per row
    constexpr int AR = 256 / AQ, BR = 256 / BQ;      // rows per pass
    constexpr int AP = BK / AR, BP = BK / BR;        // passes
    __shared__ __attribute__((aligned(16))) float smem[BK * (BM + BN)];
    float* As = smem;
    float* Bs = smem + BK * BM;

    const int nb_mn = p.tiles_mn;
    const int gb = xcd_remap(blockIdx.x, gridDim.x);   // the (tap, ci) tiles of one pixel chunk share an XCD's L2
    const int split = gb / nb_mn;
    const int lb = gb - split * nb_mn;
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, W = p.W;

    // per-thread fixed column descriptors
    const int ca = t % AQ, ra0 = t / AQ;
    const int cb = t % BQ, rb0 = t / BQ;
    const int ma = m0 + ca * 4, nbcol = n0 + cb * 4;
    const bool a_ok = ma < p.M, b_ok = nbcol < p.N;
    int a_col = ma, a_qa = 0, a_qb = 0;
    if (AMODE == WA_UP2) {
        const int ab = ma / p.Cout;
        a_col = ma - ab * p.Cout;
        a_qa = ab >> 1;
        a_qb = ab & 1;
    }
    int b_col = nbcol, b_dy = 0, b_dx = 0;
    if (BMODE == WB_CONV3) {
        const int tap = nbcol / p.Cin;
        b_col = nbcol - tap * p.Cin;
        b_dy = tap / 3 - 1;
        b_dx = tap - (tap / 3) * 3 - 1;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[AP], rb[BP];
    const long k_begin = (long)split * p.kchunk;
    long k_end = k_begin + p.kchunk;
    if (k_end > p.Kp) k_end = p.Kp;

    // raw buffer loads: 32-bit byte offsets, out-of-range offset => the hardware returns zeros (see NT kernel)
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);
    const int b_shift = b_dy * W + b_dx;
    auto load_tile = [&](long kbase) {
        const int rem = (int)(k_end - kbase);   // rows of this K-step that exist (>= 32 except at the very end)
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = ra0 + AR * i;
            const bool ok = a_ok && row < rem;
            unsigned voff;
            if (AMODE == WA_UP2) {
                const int m = (int)kbase + row;
                int j, ii, img;
                pix_split(m, p.pd, img, ii, j);
                const long src = ((long)img * (2 * H) + 2 * ii + a_qa) * (2 * W) + 2 * j + a_qb;
                voff = (unsigned)((src * p.lda + a_col) * 4);
                ra[i] = buf_load4(rsA, ok ? voff : kOOB, 0);
            } else {
                voff = (unsigned)((row * p.lda + a_col) * 4);
                ra[i] = buf_load4(rsA, ok ? voff : kOOB, (unsigned)(kbase * p.lda * 4));
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int row = rb0 + BR * i;
            bool ok = b_ok && row < rem;
            if (BMODE == WB_CONV3) {
                const int m = (int)kbase + row;
                int y, x, img_;
                pix_split(m, p.pd, img_, y, x);
                ok = ok && ((unsigned)(y + b_dy) < (unsigned)H) && ((unsigned)(x + b_dx) < (unsigned)W);
                const unsigned voff = (unsigned)((((long)m + b_shift) * p.ldb + b_col) * 4);
                rb[i] = buf_load4(rsB, ok ? voff : kOOB, 0);
            } else {
                const unsigned voff = (unsigned)((row * p.ldb + b_col) * 4);
                rb[i] = buf_load4(rsB, ok ? voff : kOOB, (unsigned)(kbase * p.ldb * 4));
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(&As[(ra0 + AR * i) * BM + ca * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < BP; ++i) *reinterpret_cast<float4*>(&Bs[(rb0 + BR * i) * BN + cb * 4]) = rb[i];
    };

    const int lrow = lane & 31, half = lane >> 5;
    const float* a_base = As + half * BM + wm * TM * 32 + lrow;
    const float* b_base = Bs + half * BN + wn * TN * 32 + lrow;

    if (k_begin < k_end) {
        load_tile(k_begin);
        store_tile();
        __syncthreads();
        for (long kb = k_begin; kb < k_end; kb += BK) {
            const bool more = kb + BK < k_end;
            if (more) load_tile(kb + BK);
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = a_base[ks * 2 * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = b_base[ks * 2 * BN + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }

    float* out = p.slab + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + lrow;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) out[(long)m * p.N + n] = acc[i][j][r];
            }
        }
}

// ---- split-bf16 version of the weight-gradient kernel (see igemm_nt_split_kernel for the arithmetic) -----------------
// Both operands are k-major in memory (pixels x channels) while the bf16 MFMA wants 8 consecutive k per lane, so the
// transpose happens in registers on the way into LDS: a staging task = 4 pixels x 4 channels (four 16-byte loads);
// for each channel the 4 pixel values of one term are packed into 8 bytes and written to that channel's LDS row
// (128-byte rows: chunk = 2*term + khalf, XOR-swizzled by (row >> 1) & 7 -- conflict-free b64 writes and b128 reads).
// Threads [0, BM) stage A, [BM, BM+BN) stage B (wave-uniform roles).  K-step = 16 pixels, LDS double-buffered, two
// K-steps of global loads in flight.
template <int NP, int BM, int BN, int WM, int WN, int AMODE, int BMODE>
__device__ __forceinline__ void wgrad_tn_split_body(const TnParams& p, float* smem, const Quant qz) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STAGE = (BM + BN) * 32;            // words

    const int nb_mn = p.tiles_mn;
    const int gb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = gb / nb_mn;
    const int lb = gb - split * nb_mn;
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int H = p.H, W = p.W;

    // staging role of this thread
    const bool isA = t < BM, active = t < BM + BN;
    const int idx = isA ? t : t - BM;
    const int kq = idx & 3, quad = idx >> 2;          // pixel quarter of the 16-pixel K-step, channel quad
    const int col0 = (isA ? m0 : n0) + quad * 4;
    const bool col_ok = active && col0 < (isA ? p.M : p.N);
    int a_col = col0, a_qa = 0, a_qb = 0;
    if (AMODE == WA_UP2 && isA) {
        const int ab = col0 / p.Cout;
        a_col = col0 - ab * p.Cout;
        a_qa = ab >> 1;
        a_qb = ab & 1;
    }
    int b_col = col0, b_dy = 0, b_dx = 0;
    if (BMODE == WB_CONV3 && !isA) {
        const int tap = col0 / p.Cin;
        b_col = col0 - tap * p.Cin;
        b_dy = tap / 3 - 1;
        b_dx = tap - (tap / 3) * 3 - 1;
    }
    const int b_shift = b_dy * W + b_dx;
    const int lds_row0 = (isA ? 0 : BM) + quad * 4;   // LDS rows of this task's 4 channels

    f32x16 acc[TM][TN], lo[TM][TN];       // hi / lo accumulators (rd_mfma_dev.h: PA6)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = lo[i][j][r] = 0.f;

    const long k_begin = (long)split * p.kchunk;
    long k_end = k_begin + p.kchunk;
    if (k_end > p.Kp) k_end = p.Kp;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(p.B, p.b_bytes);

    const bool w4 = (W & 3) == 0;     // the task's 4 consecutive pixels share an image row: one index decomposition
    auto load_task = [&](long kbase, float4 (&x)[4]) {
        const int rem = (int)(k_end - kbase);   // pixels of this K-step that exist (may be <= 0 past the end)
        int img0 = 0, y0 = 0, x0 = 0;
        if ((AMODE == WA_UP2 && isA) || (BMODE == WB_CONV3 && !isA)) pix_split((int)kbase + kq * 4, p.pd, img0, y0, x0);
        // WA_UP2, image width a multiple of 4: the task's four coarse pixels are consecutive in one row, their fine-grid
        // sources two pixels apart -- one address, then a constant stride
        unsigned up_off0 = 0;
        const unsigned up_step = (unsigned)(2 * p.lda * 4);
        if (AMODE == WA_UP2 && isA && w4)
            up_off0 = (unsigned)(((((long)img0 * (2 * H) + 2 * y0 + a_qa) * (2 * W) + 2 * x0 + a_qb) * p.lda + a_col) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = kq * 4 + j;
            const bool ok = col_ok && row < rem;
            if (isA) {
                if (AMODE == WA_UP2) {
                    unsigned off = up_off0 + j * up_step;
                    if (!w4) {
                        int jj, ii, img;
                        pix_split((int)kbase + row, p.pd, img, ii, jj);
                        off = (unsigned)(((((long)img * (2 * H) + 2 * ii + a_qa) * (2 * W) + 2 * jj + a_qb) * p.lda + a_col) * 4);
                    }
                    x[j] = buf_load4(rsA, ok ? off : kOOB, 0);
                } else {
                    x[j] = buf_load4(rsA, ok ? (unsigned)((row * p.lda + a_col) * 4) : kOOB, (unsigned)(kbase * p.lda * 4));
                }
            } else {
                if (BMODE == WB_CONV3) {
                    const int m = (int)kbase + row;
                    int y = y0, xx = x0 + j, img_ = img0;
                    if (!w4) pix_split(m, p.pd, img_, y, xx);
                    const bool in = ok && ((unsigned)(y + b_dy) < (unsigned)H) && ((unsigned)(xx + b_dx) < (unsigned)W);
                    x[j] = buf_load4(rsB, in ? (unsigned)((((long)m + b_shift) * p.ldb + b_col) * 4) : kOOB, 0);
                } else {
                    x[j] = buf_load4(rsB, ok ? (unsigned)((row * p.ldb + b_col) * 4) : kOOB, (unsigned)(kbase * p.ldb * 4));
                }
            }
        }
    };
    const float qs = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(isA ? qz.sa : qz.sb)));   // wave-uniform role
    auto store_task = [&](float* stage, const float4 (&x)[4]) {
        if (!active) return;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint2 ph, pm, pl;               // the 4 pixels of channel c
            split_pack4v<NP, false>(c == 0 ? x[0].x : c == 1 ? x[0].y : c == 2 ? x[0].z : x[0].w, c == 0 ? x[1].x : c == 1 ? x[1].y : c == 2 ? x[1].z : x[1].w,
                                c == 0 ? x[2].x : c == 1 ? x[2].y : c == 2 ? x[2].z : x[2].w, c == 0 ? x[3].x : c == 1 ? x[3].y : c == 2 ? x[3].z : x[3].w,
                                qs, ph, pm, pl);
            const int row = lds_row0 + c;
            const int sw = (row >> 1) & 7;
            float* base = stage + row * 32 + (kq & 1) * 2;
            const int hi = kq >> 1;
            *reinterpret_cast<uint2*>(base + ((0 + hi) ^ sw) * 4) = ph;
            *reinterpret_cast<uint2*>(base + ((2 + hi) ^ sw) * 4) = pm;
            if (kterm3<NP>()) *reinterpret_cast<uint2*>(base + ((4 + hi) ^ sw) * 4) = pl;
        }
    };

    const int lrow = lane & 31, half = lane >> 5;
    int a_rd[TM][3], b_rd[TN][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 3; ++q) a_rd[i][q] = row * 32 + ((2 * q + half) ^ ((row >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = BM + (wn * TN + j) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 3; ++q) b_rd[j][q] = row * 32 + ((2 * q + half) ^ ((row >> 1) & 7)) * 4;
    }
    auto mma_tile = [&](const float* stage) {
        FR af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q) af[i][q] = *reinterpret_cast<const FR*>(stage + a_rd[i][q]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < NT; ++q) bf[j][q] = *reinterpret_cast<const FR*>(stage + b_rd[j][q]);
#pragma unroll
        for (int t6 = lo0<NP>(); t6 < 5; ++t6)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) lo[i][j] = mfma16<NP>(af[i][PA6[t6]], bf[j][PB6[t6]], lo[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma16<NP>(af[i][0], bf[j][0], acc[i][j]);
    };

    if (k_begin < k_end) {
        float* st0 = smem;
        float* st1 = smem + STAGE;
        float4 x0[4], x1[4];
        load_task(k_begin, x0);
        load_task(k_begin + SK, x1);
        store_task(st0, x0);
        __syncthreads();
        for (long kb = k_begin; kb < k_end; kb += 2 * SK) {     // kchunk is a multiple of 32
            load_task(kb + 2 * SK, x0);
            mma_tile(st0);
            store_task(st1, x1);
            __syncthreads();
            load_task(kb + 3 * SK, x1);
            mma_tile(st1);
            store_task(st0, x0);
            __syncthreads();
        }
    }

    float* out = p.slab + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + lrow;
            if (n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < p.M) out[(long)m * p.N + n] = merge_q<NP>(acc[i][j][r], lo[i][j][r], qz.dexp);
            }
        }
}

template <int BM, int BN, int WM, int WN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void wgrad_tn_split_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * 32];
    const Quant qz = quant_select(p.a_amax, p.b_amax);
    if (qz.use3) wgrad_tn_split_body<3, BM, BN, WM, WN, AMODE, BMODE>(p, smem, qz);
    else wgrad_tn_split_body<6, BM, BN, WM, WN, AMODE, BMODE>(p, smem, qz);
}

struct TnPlan {
    int bm, bn, tiles_m, tiles_n, splits, kchunk;
};

static TnPlan plan_tn(int M, int N, long Kp) {
    TnPlan pl;
    pl.bm = M > 64 ? 128 : 64;
    pl.bn = N > 64 ? 128 : 64;
    const int tforce = tune(TUNE_TN_TILE);
    if (tforce > 0) {
        pl.bm = tforce / 1000;
        pl.bn = tforce % 1000;
    }
    pl.tiles_m = cdiv(M, pl.bm);
    pl.tiles_n = cdiv(N, pl.bn);
    const int tiles = pl.tiles_m * pl.tiles_n;
    long ktiles = (Kp + 31) / 32;
    const int target = tune(TUNE_TN_BLOCKS);
    long want = target / tiles;                   // whole rounds of resident blocks
    if (want < 1) want = 1;
    long maxs = ktiles / 8 > 0 ? ktiles / 8 : 1;  // at least 8 K-steps per split
    long s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    long per = (ktiles + s - 1) / s;
    pl.kchunk = (int)(per * 32);
    pl.splits = (int)((ktiles + per - 1) / per);
    return pl;
}

template <int AMODE, int BMODE>
static int launch_tn(TnParams p, const TnPlan& pl, hipStream_t s, const char* cls) {
    char pcls[64];
    // the split kernel stages both operands through registers + LDS (two transposes): it only pays on full 128x128 tiles
    const int tn_force = tune(TUNE_TN_SPLIT);
    const int split = tn_force >= 0 ? tn_force : (mfma_split() && pl.bm == 128 && pl.bn == 128);
    snprintf(pcls, sizeof(pcls), "%s|wgrad_tn%s<%d,%d,%d,%d>", cls, split ? "_split" : "", pl.bm, pl.bn, AMODE, BMODE);
    ProfScope ps(s, pcls, 2.0 * p.M * p.N * (double)p.Kp,
                 4.0 * ((double)p.Kp * p.lda * (AMODE == WA_UP2 ? 4 : 1) + (double)p.Kp * p.ldb + (double)p.M * p.N), true);
    const double a_bytes = 4.0 * (double)p.Kp * p.lda * (AMODE == WA_UP2 ? 4 : 1), b_bytes = 4.0 * (double)p.Kp * p.ldb;
    if (a_bytes >= 4294967040.0 || b_bytes >= 4294967040.0) {
        set_error("%s: operand larger than the 4 GiB buffer-descriptor range", cls);
        return RD_ERR_ARG;
    }
    p.a_bytes = (unsigned)a_bytes;
    p.b_bytes = (unsigned)b_bytes;
    p.kchunk = pl.kchunk;
    p.tiles_n = pl.tiles_n;
    p.tiles_mn = pl.tiles_m * pl.tiles_n;
    if (!split || mfma_products() != 3 || !p.a_amax || !p.b_amax) p.a_amax = p.b_amax = nullptr;    // six products
    const int grid = p.tiles_mn * pl.splits;
    if (split) {
        if (pl.bm == 128 && pl.bn == 128)
            RD_LAUNCH((wgrad_tn_split_kernel<128, 128, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
        else if (pl.bm == 128)
            RD_LAUNCH((wgrad_tn_split_kernel<128, 64, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
        else if (pl.bn == 128)
            RD_LAUNCH((wgrad_tn_split_kernel<64, 128, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
        else
            RD_LAUNCH((wgrad_tn_split_kernel<64, 64, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
        RD_LAUNCH_CHECK(cls);
        return RD_OK;
    }
    if (pl.bm == 128 && pl.bn == 128)
        RD_LAUNCH((wgrad_tn_kernel<128, 128, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
    else if (pl.bm == 128)
        RD_LAUNCH((wgrad_tn_kernel<128, 64, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
    else if (pl.bn == 128)
        RD_LAUNCH((wgrad_tn_kernel<64, 128, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
    else
        RD_LAUNCH((wgrad_tn_kernel<64, 64, 2, 2, AMODE, BMODE>), dim3(grid), dim3(256), 0, s, p);
    RD_LAUNCH_CHECK(cls);
    return RD_OK;
}

// slab[s][m][n] summed over s (fixed order, fp64) and scattered into the torch weight layout.
//  mode 0 (conv3x3): m = co, n = tap*Cin + ci  ->  dw[(co*Cin + ci)*9 + tap]
//  mode 1 (convT)  : m = ab*Cout + co, n = ci  ->  dw[(ci*Cout + co)*4 + ab]
//  mode 2 (conv1x1): m = co, n = ci             ->  dw[co*Cin + ci]
//  mode 3 (conv3x3, swapped strip kernel): m = ci, n = (8-tap)*Cout + co  ->  dw[(co*Cin + ci)*9 + tap]
// One thread owns four consecutive n (16-byte coalesced slab reads, four splits in flight).
// SPT = lanes that share one output quad: each sums the slabs s = part, part + SPT, ... in fp64 and the partial sums are
// combined by a fixed butterfly (deterministic).  Few outputs x many slabs (e.g. the 64-channel transposed convolution:
// 4096 quads x 256 slabs) would otherwise be a handful of latency-bound blocks.
template <int SPT>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M,
                                                          int N, int splits, int mode, int Cin, int Cout) {
    const long total = (long)M * N, quads = total >> 2;   // N % 4 == 0 (channels are multiples of 4)
    const long nthreads = (long)gridDim.x * blockDim.x;
    // lane groups are aligned (256 % SPT == 0, total thread count a multiple of SPT): the SPT lanes of a quad always
    // run the same iterations, so the shuffles below only ever read live partners
    for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < quads * SPT; gid += nthreads) {
        const long q = gid / SPT;
        const int part = (int)(gid - q * SPT);
        const float4* src = reinterpret_cast<const float4*>(slab) + q;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int s = part;
        for (; s + 3 * SPT < splits; s += 4 * SPT) {
            // (plain loads: the non-temporal form measured 4 % SLOWER here -- 0.265 -> 0.276 ms per step, profiles/r05_notes.md
            // section 14 -- the slabs were written a moment ago by the strip kernel and part of them still sits in L2)
            const float4 v0 = src[(long)s * quads], v1 = src[(long)(s + SPT) * quads];
            const float4 v2 = src[(long)(s + 2 * SPT) * quads], v3 = src[(long)(s + 3 * SPT) * quads];
            a0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            a1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
            a2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
            a3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
        }
        for (; s < splits; s += SPT) {
            const float4 v = src[(long)s * quads];
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
#pragma unroll
        for (int o = 1; o < SPT; o <<= 1) {
            a0 += __shfl_xor(a0, o);
            a1 += __shfl_xor(a1, o);
            a2 += __shfl_xor(a2, o);
            a3 += __shfl_xor(a3, o);
        }
        if (part != 0) continue;
        const long e = q << 2;
        const int m = (int)(e / N), n = (int)(e - (long)m * N);
        const float r[4] = {(float)a0, (float)a1, (float)a2, (float)a3};
        if (mode == 3) {                                      // mirrored transpose of the strip kernel: m = ci, n = (8-tap)*Cout + co
            const int tq = n / Cout, co = n - tq * Cout;     // the quad stays inside one tap (Cout % 4 == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) dw[((long)(co + k) * Cin + m) * 9 + (8 - tq)] = r[k];
        } else if (mode == 2) {                               // plain [M][N] (conv1x1: dw[co][ci])
            *reinterpret_cast<float4*>(dw + e) = make_float4(r[0], r[1], r[2], r[3]);
        } else if (mode == 0) {
            const int tap = n / Cin, ci = n - tap * Cin;     // the quad stays inside one tap (Cin % 4 == 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) dw[((long)m * Cin + ci + k) * 9 + tap] = r[k];
        } else {
            const int ab = m / Cout, co = m - ab * Cout;
#pragma unroll
            for (int k = 0; k < 4; ++k) dw[((long)(n + k) * Cout + co) * 4 + ab] = r[k];
        }
    }
}

static void launch_slab_reduce(const float* slab, float* dw, int M, int N, int splits, int mode, int Cin, int Cout, hipStream_t s) {
#if RD_DIAG_SLAB_READ_DIV
    // DIAGNOSIS BUILD (wrong results): the reduction reads only every RD_DIAG_SLAB_READ_DIV-th slab -- the upper bound of what a
    // grouped fix-up inside the weight-gradient kernels could take off the reduce (r05 verdict item 5; profiles/r06_notes.md)
    splits = (splits + RD_DIAG_SLAB_READ_DIV - 1) / RD_DIAG_SLAB_READ_DIV;
#endif
    const long quads = (long)M * N / 4;
    int spt = 1;
    while (spt < 16 && quads * spt < 262144 && 2 * spt <= splits) spt *= 2;      // ~1024 blocks of work, at most 16 lanes per quad
    const int grid = grid_for(quads * spt, 256, 4096);
    if (spt == 1) RD_LAUNCH(slab_reduce_kernel<1>, dim3(grid), dim3(256), 0, s, slab, dw, M, N, splits, mode, Cin, Cout);
    else if (spt == 2) RD_LAUNCH(slab_reduce_kernel<2>, dim3(grid), dim3(256), 0, s, slab, dw, M, N, splits, mode, Cin, Cout);
    else if (spt == 4) RD_LAUNCH(slab_reduce_kernel<4>, dim3(grid), dim3(256), 0, s, slab, dw, M, N, splits, mode, Cin, Cout);
    else if (spt == 8) RD_LAUNCH(slab_reduce_kernel<8>, dim3(grid), dim3(256), 0, s, slab, dw, M, N, splits, mode, Cin, Cout);
    else RD_LAUNCH(slab_reduce_kernel<16>, dim3(grid), dim3(256), 0, s, slab, dw, M, N, splits, mode, Cin, Cout);
}

__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                    int cout, int cin) {
    const long total = (long)cout * cin * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e indexes wf[co][tap][ci]
        const int ci = (int)(e % cin);
        const int tap = (int)((e / cin) % 9);
        const int co = (int)(e / ((long)cin * 9));
        wf[e] = w[((long)co * cin + ci) * 9 + tap];
        if (wd) {
            // the same e read as an index into wd[ci'][tap'][co']: coalesced stores, gathered (cached) loads
            const int co2 = (int)(e % cout);
            const int tap2 = (int)((e / cout) % 9);
            const int ci2 = (int)(e / ((long)cout * 9));
            wd[e] = w[((long)co2 * cin + ci2) * 9 + (8 - tap2)];
        }
    }
}

// out[c][g'][r] = in[r][g][c] for a matrix in[rows][groups][cols] (g' = groups-1-g when flip, else g): 32x32 LDS tiles,
// coalesced on both sides.  conv3x3: wd[ci][8-tap][co] = wf[co][tap][ci]; plain transposes use groups = 1.
__global__ __launch_bounds__(256) void tiled_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                              int groups, int cols, int flip) {
    __shared__ float tile[32][33];
    const int tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (long b = blockIdx.x; b < (long)groups * tiles_r * tiles_c; b += gridDim.x) {
        const int tc = (int)(b % tiles_c), tr = (int)((b / tiles_c) % tiles_r), g = (int)(b / ((long)tiles_c * tiles_r));
        const int r0 = tr * 32, c0 = tc * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            tile[ty + 8 * k][tx] = (r < rows && c < cols) ? in[((long)r * groups + g) * cols + c] : 0.f;
        }
        __syncthreads();
        const int go = flip ? groups - 1 - g : g;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < rows && c < cols) out[((long)c * groups + go) * rows + r] = tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}


__global__ void pack_convt_kernel(const float* __restrict__ w, float* __restrict__ wtf, float* __restrict__ wtd,
                                  int cin, int cout) {
    const long total = (long)cin * cout * 4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e indexes wtf[(ab*Cout + co)][ci]
        const int ci = (int)(e % cin);
        const int co = (int)((e / cin) % cout);
        const int ab = (int)(e / ((long)cin * cout));
        wtf[e] = w[((long)ci * cout + co) * 4 + ab];
        if (wtd) {
            // e as an index into wtd[ci'][(ab', co')]: coalesced stores
            const int co2 = (int)(e % cout);
            const int ab2 = (int)((e / cout) % 4);
            const int ci2 = (int)(e / ((long)cout * 4));
            wtd[e] = w[((long)ci2 * cout + co2) * 4 + ab2];
        }
    }
}

static int grid_for(long total, int block = 256, int cap = 4096) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static int check_conv_args(int n, int h, int w, int cin, int cout) {
    RD_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "bad conv shape n=%d h=%d w=%d cin=%d cout=%d", n, h, w,
               cin, cout);
    RD_REQUIRE((long)n * h * w * 4L < (1L << 31), "pixel count too large for 32-bit tile indices");
    return RD_OK;
}

}  // namespace rd

using namespace rd;

extern "C" {

// 3: operands that carry magnitude slots (rd_quant_next) are multiplied as two fp16 terms / three products (rd_mfma_dev.h);
// 6: three bf16 terms / six products everywhere.  The knob `mfma_products` (RD_MFMA=split2h | split3) of ONE library.
int rd_mfma_products(void) { return mfma_products(); }

int rd_amax(const float* x, long long n, unsigned* slot, rd_stream_t s) {
    RD_REQUIRE(x && slot && n > 0, "rd_amax: bad arguments");
    launch_amax(x, (long)n, slot, nullptr, 1, (hipStream_t)s);
    RD_LAUNCH_CHECK("rd_amax");
    return RD_OK;
}


size_t rd_packed_weight_bytes(int rows, int taps, int cin) {
    if (rows <= 0 || taps <= 0 || cin <= 0) return 0;
    return packed_bytes(rows, taps, cin);
}

int rd_pack_conv3x3_weight(const float* w, float* wf, float* wd, int cout, int cin, rd_stream_t s) {
    RD_REQUIRE(w && wf && cout > 0 && cin > 0, "rd_pack_conv3x3_weight: bad arguments");
    ProfScope ps((hipStream_t)s, "pack_weights", 0, 12.0 * cout * cin * 9);
    unsigned* amax = mfma_products() == 3 ? quant_take().out2 : (quant_take(), nullptr);   // the weight's magnitude slot (zeroed)
    if (amax) launch_amax(w, (long)cout * cin * 9, amax, nullptr, 1, (hipStream_t)s);
    if (mfma_split()) {     // split kernels only: both fragment tensors in one launch, fp32 GEMM layouts left unwritten
        uint4* of = (uint4*)((char*)wf + packed_f32_bytes(cout, 9, cin));
        uint4* od = wd ? (uint4*)((char*)wd + packed_f32_bytes(cin, 9, cout)) : nullptr;
        const long pieces = rows32_of(cout) * nk16_of(9, cin) * 2 + (wd ? rows32_of(cin) * nk16_of(9, cout) * 2 : 0);
        long g = (pieces + 255) / 256;
        if (g > 8192) g = 8192;
        RD_LAUNCH(split_pack_conv3x3_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, w, of, od, cout, cin,
                           (const float*)nullptr, (const unsigned*)amax);
        RD_LAUNCH_CHECK("pack_conv3x3");
        return RD_OK;
    }
    RD_LAUNCH(pack_conv3x3_kernel, dim3(grid_for((long)cout * cin * 9)), dim3(256), 0, (hipStream_t)s, w, wf,
                       (float*)nullptr, cout, cin);
    if (wd) {
        const long tiles = 9L * cdiv(cout, 32) * cdiv(cin, 32);
        RD_LAUNCH(tiled_transpose_kernel, dim3((int)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)s,
                           (const float*)wf, wd, cout, 9, cin, 1);
    }
    RD_LAUNCH_CHECK("pack_conv3x3");
    if (int e = split_pack(wf, cout, 9, cin, (hipStream_t)s, amax)) return e;
    if (wd) return split_pack(wd, cin, 9, cout, (hipStream_t)s, amax);
    return RD_OK;
}

int rd_pack_conv3x3_weight_folded(const float* w, const float* row_scale, float* wf, int cout, int cin, rd_stream_t s) {
    RD_REQUIRE(w && wf && row_scale && cout > 0 && cin > 0, "rd_pack_conv3x3_weight_folded: bad arguments");
    RD_REQUIRE(mfma_split(), "rd_pack_conv3x3_weight_folded: only the split-bf16 kernels implement the folded inference path");
    ProfScope ps((hipStream_t)s, "pack_weights", 0, 10.0 * cout * cin * 9);
    unsigned* amax = mfma_products() == 3 ? quant_take().out2 : (quant_take(), nullptr);   // magnitude slot of the SCALED rows
    if (amax) launch_amax(w, (long)cout * cin * 9, amax, row_scale, (long)cin * 9, (hipStream_t)s);
    uint4* of = (uint4*)((char*)wf + packed_f32_bytes(cout, 9, cin));
    const long pieces = rows32_of(cout) * nk16_of(9, cin) * 2;
    long g = (pieces + 255) / 256;
    if (g > 8192) g = 8192;
    RD_LAUNCH(split_pack_conv3x3_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, w, of, (uint4*)nullptr, cout, cin,
                       row_scale, (const unsigned*)amax);
    RD_LAUNCH_CHECK("pack_conv3x3_folded");
    return RD_OK;
}

long long rd_pack_item_pieces(int kind, int cout, int cin, int with_f32) {
    if (cout <= 0 || cin <= 0) return 0;
    if (kind == 0) return rows32_of(cout) * nk16_of(9, cin) * 2 + rows32_of(cin) * nk16_of(9, cout) * 2;
    return rows32_of(4L * cout) * nk16_of(1, cin) * 2 + rows32_of(cin) * nk16_of(4, cout) * 2 + (with_f32 ? (4L * cout * cin + 7) / 8 : 0);
}

long long rd_pack_item_tiles(int kind, int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cout % 32 != 0 || cin % 32 != 0 || (kind != 0 && kind != 1)) return 0;
    return (long long)(cout / 32) * (cin / 32);
}

int rd_pack_weights_fused(const void* items_dev, int n_items, long long total_pieces, long long total_tiles, rd_stream_t s) {
    RD_REQUIRE(items_dev && n_items > 0 && total_pieces >= 0 && total_tiles >= 0 && total_pieces + total_tiles > 0,
               "rd_pack_weights_fused: bad arguments");
    RD_REQUIRE(mfma_split(), "rd_pack_weights_fused: the fused packer writes the split-bf16 operands only");
    ProfScope ps((hipStream_t)s, "pack_weights", 0, 10.0 * 8.0 * (double)total_pieces + 14.0 * 9216.0 * (double)total_tiles);
    // items with a magnitude slot (column 9 of the table; zeroed by the caller) get the three-product form too: first their maxima
    if (mfma_products() == 3)
        RD_LAUNCH(pack_items_amax_kernel, dim3(AMAX_PARTS * n_items), dim3(256), 0, (hipStream_t)s, (const PackItem*)items_dev, n_items);
    if (total_tiles > 0) {
        const long gt = total_tiles < 4096 ? total_tiles : 4096;
        RD_LAUNCH(pack_tiles_kernel, dim3((int)gt), dim3(256), 0, (hipStream_t)s, (const PackItem*)items_dev, n_items,
                           (long)total_tiles);
    }
    if (total_pieces > 0) {
        long g = (total_pieces + 255) / 256;
        if (g > 16384) g = 16384;
        RD_LAUNCH(pack_all_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)s, (const PackItem*)items_dev, n_items,
                           (long)total_pieces);
    }
    RD_LAUNCH_CHECK("pack_weights_fused");
    return RD_OK;
}

int rd_pack_convt2x2_weight(const float* w, float* wtf, float* wtd, int cin, int cout, rd_stream_t s) {
    RD_REQUIRE(w && wtf && cout > 0 && cin > 0, "rd_pack_convt2x2_weight: bad arguments");
    ProfScope ps((hipStream_t)s, "pack_weights", 0, 12.0 * cout * cin * 4);
    unsigned* amax = mfma_products() == 3 ? quant_take().out2 : (quant_take(), nullptr);
    if (amax) launch_amax(w, (long)cout * cin * 4, amax, nullptr, 1, (hipStream_t)s);
    RD_LAUNCH(pack_convt_kernel, dim3(grid_for((long)cout * cin * 4)), dim3(256), 0, (hipStream_t)s, w, wtf,
                       (float*)nullptr, cin, cout);
    if (wtd) {
        const long tiles = (long)cdiv(4 * cout, 32) * cdiv(cin, 32);
        RD_LAUNCH(tiled_transpose_kernel, dim3((int)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)s,
                           (const float*)wtf, wtd, 4 * cout, 1, cin, 0);
    }
    RD_LAUNCH_CHECK("pack_convt");
    if (int e = split_pack(wtf, 4L * cout, 1, cin, (hipStream_t)s, amax)) return e;
    if (wtd) return split_pack(wtd, cin, 4, cout, (hipStream_t)s, amax);
    return RD_OK;
}

int rd_conv3x3_fwd(const float* x, const float* wf, float* z, int n, int h, int w, int cin, int cout, rd_stream_t s) {
    return rd_conv3x3_fwd_stats(x, wf, z, nullptr, n, h, w, cin, cout, nullptr, 0, s);
}

size_t rd_conv3x3_fwd_stats_ws_bytes(int n, int h, int w, int cin, int cout) {
    (void)cin;
    return (size_t)cdiv((long)n * h * w, 64) * 2 * cout * sizeof(float);
}

int rd_conv3x3_fwd_stats(const float* x, const float* wf, float* z, double* sums, int n, int h, int w, int cin, int cout,
                         void* ws, size_t ws_bytes, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && wf && z, "rd_conv3x3_fwd: null pointer");
    RD_REQUIRE(cin % 4 == 0, "rd_conv3x3_fwd: Cin must be a multiple of 4 (got %d); use rd_conv3x3_first_fwd", cin);
    if (sums && (!ws || ws_bytes < rd_conv3x3_fwd_stats_ws_bytes(n, h, w, cin, cout))) {
        set_error("rd_conv3x3_fwd_stats: workspace too small");
        return RD_ERR_WS;
    }
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = x; p.B = wf; p.C = z;
    p.M = n * h * w; p.N = cout; p.K = 9 * cin; p.Cin = cin;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    p.stats = sums ? (float*)ws : nullptr;
    int tiles_m = 0;
    if (int e = launch_nt<A_CONV3, EPI_STORE>(p, (hipStream_t)s, "conv3x3_fwd", &tiles_m)) return e;
    if (sums) return reduce_partials_f32((const float*)ws, sums, tiles_m, 2 * cout, (hipStream_t)s);
    return RD_OK;
}

int rd_conv3x3_fwd_bn(const float* x, const float* wf, float* z, double count, float eps, float momentum, float* mean,
                      float* invstd, float* running_mean, float* running_var, int64_t* nbt, int n, int h, int w, int cin,
                      int cout, void* ws, size_t ws_bytes, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && wf && z && mean && invstd && count > 0, "rd_conv3x3_fwd_bn: bad arguments");
    RD_REQUIRE(cin % 4 == 0, "rd_conv3x3_fwd_bn: Cin must be a multiple of 4 (got %d); use rd_conv3x3_first_fwd_bn", cin);
    if (!ws || ws_bytes < rd_conv3x3_fwd_stats_ws_bytes(n, h, w, cin, cout)) {
        set_error("rd_conv3x3_fwd_bn: workspace too small");
        return RD_ERR_WS;
    }
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = x; p.B = wf; p.C = z;
    p.M = n * h * w; p.N = cout; p.K = 9 * cin; p.Cin = cin;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    p.stats = (float*)ws;
    int tiles_m = 0;
    if (int e = launch_nt<A_CONV3, EPI_STORE>(p, (hipStream_t)s, "conv3x3_fwd", &tiles_m)) return e;
    return bn_reduce_finalize((const float*)ws, tiles_m, cout, count, eps, momentum, mean, invstd, running_mean, running_var, nbt,
                              (hipStream_t)s);
}

int rd_conv3x3_fwd_act(const float* x, const float* wf_folded, const float* shift, float slope, float* a, float* pooled, int n,
                       int h, int w, int cin, int cout, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && wf_folded && shift && a, "rd_conv3x3_fwd_act: null pointer");
    RD_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "rd_conv3x3_fwd_act: channels must be multiples of 4 (%d, %d)", cin, cout);
    RD_REQUIRE(mfma_split(), "rd_conv3x3_fwd_act: only the split-bf16 kernels implement the folded inference path");
    RD_REQUIRE(!pooled || (w % 16 == 0 && h % 8 == 0),
               "rd_conv3x3_fwd_act: the pooling epilogue needs W a multiple of 16 and H a multiple of 8 (got %dx%d)", h, w);
    NtParams p = {};
    set_quant(p, quant_take_img());
    p.A = x; p.B = wf_folded; p.C = a;
    p.M = n * h * w; p.N = cout; p.K = 9 * cin; p.Cin = cin;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    p.shift = shift; p.act_slope = slope; p.pool_out = pooled;
    return launch_nt<A_CONV3, EPI_STORE>(p, (hipStream_t)s, "conv3x3_fwd");
}

int rd_conv3x3_bwd_data(const float* dz, const float* wd, float* dx, int n, int h, int w, int cin, int cout,
                        rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(dz && wd && dx, "rd_conv3x3_bwd_data: null pointer");
    RD_REQUIRE(cout % 4 == 0, "rd_conv3x3_bwd_data: Cout must be a multiple of 4 (got %d)", cout);
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = dz; p.B = wd; p.C = dx;
    p.M = n * h * w; p.N = cin; p.K = 9 * cout; p.Cin = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    return launch_nt<A_CONV3, EPI_STORE>(p, (hipStream_t)s, "conv3x3_dgrad");
}

// BN-backward statistics from the epilogue of the kernel that produces the gradient operand (NtParams::bn_part)
static int set_bn_hook(NtParams& p, const char* who, const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, float slope, const float* slope_dev, int mode, float* part, size_t part_floats) {
    RD_REQUIRE(bn_z && mean && invstd && gamma && beta && part && (mode == 1 || mode == 2), "%s: bad BN-statistics arguments", who);
    RD_REQUIRE(p.N % 4 == 0, "%s: the BN-statistics epilogue needs a multiple of 4 channels (got %d)", who, p.N);
    RD_REQUIRE(part_floats >= rd_bn_bwd_part_floats(p.M, p.N), "%s: statistics buffer too small (%zu < %zu floats)", who,
               part_floats, rd_bn_bwd_part_floats(p.M, p.N));
    p.bn_z = bn_z; p.bn_mean = mean; p.bn_invstd = invstd; p.bn_gamma = gamma; p.bn_beta = beta;
    p.bn_slope = slope; p.bn_slope_dev = slope_dev; p.bn_mode = mode; p.bn_part = part;
    return RD_OK;
}

size_t rd_bn_bwd_part_floats(long long pixels, int c) { return (size_t)cdiv(pixels, 64) * 4 * (size_t)c; }

int rd_conv3x3_bwd_data_bnstats(const float* dz, const float* wd, float* dx, int n, int h, int w, int cin, int cout,
                                const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float slope, const float* slope_dev, int mode, float* part,
                                size_t part_floats, int* rows_out, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(dz && wd && dx && rows_out, "rd_conv3x3_bwd_data_bnstats: null pointer");
    RD_REQUIRE(cout % 4 == 0, "rd_conv3x3_bwd_data_bnstats: Cout must be a multiple of 4 (got %d)", cout);
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = dz; p.B = wd; p.C = dx;
    p.M = n * h * w; p.N = cin; p.K = 9 * cout; p.Cin = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    if (int e = set_bn_hook(p, "rd_conv3x3_bwd_data_bnstats", bn_z, mean, invstd, gamma, beta, slope, slope_dev, mode, part,
                            part_floats))
        return e;
    return launch_nt<A_CONV3, EPI_STORE>(p, (hipStream_t)s, "conv3x3_dgrad", rows_out);
}

size_t rd_conv3x3_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout) {
    if (const int ss = wgrad_strip_splits(n, h, w, cin, cout)) return (size_t)ss * cout * 9 * cin * sizeof(float);
    TnPlan pl = plan_tn(cout, 9 * cin, (long)n * h * w);
    return (size_t)pl.splits * cout * 9 * cin * sizeof(float);
}

int rd_conv3x3_bwd_weight(const float* x, const float* dz, float* dw, int n, int h, int w, int cin, int cout, void* ws,
                          size_t ws_bytes, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && dz && dw, "rd_conv3x3_bwd_weight: null pointer");
    RD_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "rd_conv3x3_bwd_weight: channels must be multiples of 4 (%d, %d)", cin,
               cout);
    const QuantArgs wq = quant_take();       // a: slot of dz, b: slot of x
    const size_t need = rd_conv3x3_bwd_weight_ws_bytes(n, h, w, cin, cout);
    if (ws_bytes < need || !ws) {
        set_error("rd_conv3x3_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    int strip_splits = 0, strip_swapped = 0;
    if (int e = wgrad_strip_launch(x, dz, (float*)ws, n, h, w, cin, cout, (hipStream_t)s, &strip_splits, &strip_swapped, wq.b, wq.a)) return e;
    if (strip_splits > 0) {
        ProfScope ps((hipStream_t)s, "wgrad_reduce", 0, 4.0 * (strip_splits + 1) * (double)cout * 9 * cin);
        if (strip_swapped)
            launch_slab_reduce((const float*)ws, dw, cin, 9 * cout, strip_splits, 3, cin, cout, (hipStream_t)s);
        else
            launch_slab_reduce((const float*)ws, dw, cout, 9 * cin, strip_splits, 0, cin, cout, (hipStream_t)s);
        RD_LAUNCH_CHECK("slab_reduce");
        return RD_OK;
    }
    TnPlan pl = plan_tn(cout, 9 * cin, (long)n * h * w);
    TnParams p = {};
    p.a_amax = wq.a; p.b_amax = wq.b;
    p.A = dz; p.B = x; p.slab = (float*)ws;
    p.M = cout; p.N = 9 * cin; p.Kp = (long)n * h * w;
    p.lda = cout; p.ldb = cin; p.Cin = cin; p.Cout = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    if (int e = launch_tn<WA_PLAIN, WB_CONV3>(p, pl, (hipStream_t)s, "conv3x3_wgrad")) return e;
    ProfScope ps((hipStream_t)s, "wgrad_reduce", 0, 4.0 * (pl.splits + 1) * (double)p.M * p.N);
    launch_slab_reduce((const float*)ws, dw, p.M, p.N, pl.splits, 0, cin, cout, (hipStream_t)s);
    RD_LAUNCH_CHECK("slab_reduce");
    return RD_OK;
}

int rd_convt2x2_fwd(const float* x, const float* wtf, const float* bias, const float* skip, float* out, int n, int h,
                    int w, int cin, int cout, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && wtf && out, "rd_convt2x2_fwd: null pointer");
    RD_REQUIRE(cin % 4 == 0, "rd_convt2x2_fwd: Cin must be a multiple of 4 (got %d)", cin);
    const QuantArgs qa = quant_take_img();
    {
        int launched = 0;
        const size_t sb = (size_t)rows32_of(4L * cout) * nk16_of(1, cin) * SROWB;
        if (int e = convt_fwd_launch(x, (const char*)wtf + packed_f32_bytes(4L * cout, 1, cin), sb, bias, skip, nullptr, nullptr,
                                     nullptr, nullptr, 0.f, nullptr, out, n, h, w, cin, cout, (hipStream_t)s, &launched, qa))
            return e;
        if (launched) return RD_OK;
    }
    NtParams p = {};
    set_quant(p, qa);
    p.A = x; p.B = wtf; p.C = out; p.bias = bias; p.skip = skip;
    p.M = n * h * w; p.N = 4 * cout; p.K = cin; p.Cin = cin; p.Cout = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    return launch_nt<A_PLAIN, EPI_CONVT>(p, (hipStream_t)s, "convt2x2_fwd");
}

int rd_convt2x2_fwd_bnskip(const float* x, const float* wtf, const float* bias, const float* z_skip, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float slope, const float* slope_dev,
                           float* out, int n, int h, int w, int cin, int cout, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && wtf && out && z_skip && mean && invstd && gamma && beta, "rd_convt2x2_fwd_bnskip: null pointer");
    RD_REQUIRE(cin % 4 == 0, "rd_convt2x2_fwd_bnskip: Cin must be a multiple of 4 (got %d)", cin);
    const QuantArgs qa = quant_take_img();
    {
        int launched = 0;
        const size_t sb = (size_t)rows32_of(4L * cout) * nk16_of(1, cin) * SROWB;
        if (int e = convt_fwd_launch(x, (const char*)wtf + packed_f32_bytes(4L * cout, 1, cin), sb, bias, z_skip, mean, invstd,
                                     gamma, beta, slope, slope_dev, out, n, h, w, cin, cout, (hipStream_t)s, &launched, qa))
            return e;
        if (launched) return RD_OK;
    }
    NtParams p = {};
    set_quant(p, qa);
    p.A = x; p.B = wtf; p.C = out; p.bias = bias; p.skip = z_skip;
    p.sk_mean = mean; p.sk_invstd = invstd; p.sk_gamma = gamma; p.sk_beta = beta; p.sk_slope = slope; p.sk_slope_dev = slope_dev;
    p.M = n * h * w; p.N = 4 * cout; p.K = cin; p.Cin = cin; p.Cout = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    return launch_nt<A_PLAIN, EPI_CONVT>(p, (hipStream_t)s, "convt2x2_fwd");
}

// dedicated transposed-convolution data-gradient kernel (rd_convt.hip): -1 = launched, 0 = not handled, > 0 = error code
static int convt_dgrad_try(NtParams p, hipStream_t s, int* rows_out) {
    const int taps = 4;
    const double b_bytes = (double)rows32_of(p.N) * (taps * cdiv(p.Cin, SK)) * SROWB;
    if (b_bytes >= 4294967040.0) return 0;
    p.b_bytes = (unsigned)b_bytes;
    p.Bsplit = (const char*)p.B + packed_f32_bytes(p.N, taps, p.Cin);
    p.Bsplit3 = (const char*)p.Bsplit + (size_t)b_bytes;
    p.b_bytes3 = (unsigned)((double)rows32_of(p.N) * (taps * cdiv(p.Cin, SK)) * SROWB3);
    int launched = 0;
    if (int e = convt_dgrad_launch(p, s, &launched, rows_out)) return e;
    return launched ? -1 : 0;
}

int rd_convt2x2_bwd_data(const float* dout, const float* wtd, float* dx, int n, int h, int w, int cin, int cout,
                         rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(dout && wtd && dx, "rd_convt2x2_bwd_data: null pointer");
    RD_REQUIRE(cout % 4 == 0, "rd_convt2x2_bwd_data: Cout must be a multiple of 4 (got %d)", cout);
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = dout; p.B = wtd; p.C = dx;
    p.M = n * h * w; p.N = cin; p.K = 4 * cout; p.Cin = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    if (int e = convt_dgrad_try(p, (hipStream_t)s, nullptr)) return e > 0 ? e : RD_OK;
    return launch_nt<A_UP2, EPI_STORE>(p, (hipStream_t)s, "convt2x2_dgrad");
}

int rd_convt2x2_bwd_data_bnstats(const float* dout, const float* wtd, float* dx, int n, int h, int w, int cin, int cout,
                                 const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, float slope, const float* slope_dev, float* part, size_t part_floats,
                                 int* rows_out, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(dout && wtd && dx && rows_out, "rd_convt2x2_bwd_data_bnstats: null pointer");
    RD_REQUIRE(cout % 4 == 0, "rd_convt2x2_bwd_data_bnstats: Cout must be a multiple of 4 (got %d)", cout);
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = dout; p.B = wtd; p.C = dx;
    p.M = n * h * w; p.N = cin; p.K = 4 * cout; p.Cin = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    if (int e = set_bn_hook(p, "rd_convt2x2_bwd_data_bnstats", bn_z, mean, invstd, gamma, beta, slope, slope_dev, 1, part,
                            part_floats))
        return e;
    if (int e = convt_dgrad_try(p, (hipStream_t)s, rows_out)) return e > 0 ? e : RD_OK;
    return launch_nt<A_UP2, EPI_STORE>(p, (hipStream_t)s, "convt2x2_dgrad", rows_out);
}

size_t rd_convt2x2_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout) {
    if (const size_t own = convt_wgrad_ws_bytes(n, h, w, cin, cout)) return own;
    TnPlan pl = plan_tn(4 * cout, cin, (long)n * h * w);
    return (size_t)pl.splits * 4 * cout * cin * sizeof(float);
}

int rd_convt2x2_bwd_weight(const float* x, const float* dout, float* dw, int n, int h, int w, int cin, int cout,
                           void* ws, size_t ws_bytes, rd_stream_t s) {
    if (int e = check_conv_args(n, h, w, cin, cout)) return e;
    RD_REQUIRE(x && dout && dw, "rd_convt2x2_bwd_weight: null pointer");
    RD_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "rd_convt2x2_bwd_weight: channels must be multiples of 4 (%d, %d)", cin,
               cout);
    const QuantArgs wq = quant_take();       // a: slot of dout, b: slot of x
    const size_t need = rd_convt2x2_bwd_weight_ws_bytes(n, h, w, cin, cout);
    if (ws_bytes < need || !ws) {
        set_error("rd_convt2x2_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    {
        int splits = 0;
        if (int e = convt_wgrad_launch(x, dout, (float*)ws, n, h, w, cin, cout, (hipStream_t)s, &splits, wq.b, wq.a)) return e;
        if (splits > 0) {
            ProfScope ps((hipStream_t)s, "wgrad_reduce", 0, 4.0 * (splits + 1) * 4.0 * cout * cin);
            launch_slab_reduce((const float*)ws, dw, 4 * cout, cin, splits, 1, cin, cout, (hipStream_t)s);
            RD_LAUNCH_CHECK("slab_reduce");
            return RD_OK;
        }
    }
    TnPlan pl = plan_tn(4 * cout, cin, (long)n * h * w);
    TnParams p = {};
    p.a_amax = wq.a; p.b_amax = wq.b;
    p.A = dout; p.B = x; p.slab = (float*)ws;
    p.M = 4 * cout; p.N = cin; p.Kp = (long)n * h * w;
    p.lda = cout; p.ldb = cin; p.Cin = cin; p.Cout = cout;
    p.H = h; p.W = w; p.pd = make_pixdiv(h, w);
    if (int e = launch_tn<WA_UP2, WB_PLAIN>(p, pl, (hipStream_t)s, "convt2x2_wgrad")) return e;
    ProfScope ps((hipStream_t)s, "wgrad_reduce", 0, 4.0 * (pl.splits + 1) * (double)p.M * p.N);
    launch_slab_reduce((const float*)ws, dw, p.M, p.N, pl.splits, 1, cin, cout, (hipStream_t)s);
    RD_LAUNCH_CHECK("slab_reduce");
    return RD_OK;
}

// ---- conv1x1 of the bilinear up-mode (lib/UNet.py:8-9,20).  A 1x1 convolution commutes with the bilinear
// interpolation (both linear; the interpolation weights sum to 1 so the bias commutes too), so the engine applies it on
// the COARSE grid -- a plain [pixels x Cin] x [Cin x Cout] GEMM with a quarter of the reference's work.
int rd_pack_conv1x1_weight(const float* w, float* wf, float* wt, int cout, int cin, rd_stream_t s) {
    RD_REQUIRE(w && wf && cout > 0 && cin > 0, "rd_pack_conv1x1_weight: bad arguments");
    ProfScope ps((hipStream_t)s, "pack_weights", 0, 16.0 * cout * cin);
    unsigned* amax = mfma_products() == 3 ? quant_take().out2 : (quant_take(), nullptr);
    if (amax) launch_amax(w, (long)cout * cin, amax, nullptr, 1, (hipStream_t)s);
    plan_poison((hipStream_t)s, "rd_pack_conv1x1_weight (bilinear up-mode) copies with hipMemcpyAsync");
    if (int e = check_hip(hipMemcpyAsync(wf, w, (size_t)cout * cin * 4, hipMemcpyDeviceToDevice, (hipStream_t)s),
                          "pack_conv1x1 copy"))
        return e;
    if (int e = split_pack(wf, cout, 1, cin, (hipStream_t)s, amax)) return e;
    if (!wt) return RD_OK;
    {
        const long tiles = (long)cdiv(cout, 32) * cdiv(cin, 32);
        RD_LAUNCH(tiled_transpose_kernel, dim3((int)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)s, w, wt,
                           cout, 1, cin, 0);
    }
    RD_LAUNCH_CHECK("pack_conv1x1");
    return split_pack(wt, cin, 1, cout, (hipStream_t)s, amax);
}

int rd_conv1x1_fwd(const float* x, const float* w, float* out, long long pixels, int cin, int cout, rd_stream_t s) {
    RD_REQUIRE(x && w && out && pixels > 0 && cin > 0 && cout > 0, "rd_conv1x1_fwd: bad arguments");
    RD_REQUIRE(cin % 4 == 0, "rd_conv1x1_fwd: Cin must be a multiple of 4 (got %d)", cin);
    RD_REQUIRE(pixels * 4LL < (1LL << 31), "rd_conv1x1_fwd: pixel count too large for 32-bit tile indices");
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = x; p.B = w; p.C = out;
    p.M = (int)pixels; p.N = cout; p.K = cin; p.Cin = cin;
    p.H = 1; p.W = 1; p.pd = make_pixdiv(1, 1);
    return launch_nt<A_PLAIN, EPI_STORE>(p, (hipStream_t)s, "conv1x1_fwd");
}

int rd_conv1x1_bwd_data(const float* dy, const float* wt, float* dx, long long pixels, int cin, int cout, rd_stream_t s) {
    RD_REQUIRE(dy && wt && dx && pixels > 0 && cin > 0 && cout > 0, "rd_conv1x1_bwd_data: bad arguments");
    RD_REQUIRE(cout % 4 == 0, "rd_conv1x1_bwd_data: Cout must be a multiple of 4 (got %d)", cout);
    RD_REQUIRE(pixels * 4LL < (1LL << 31), "rd_conv1x1_bwd_data: pixel count too large for 32-bit tile indices");
    NtParams p = {};
    set_quant(p, quant_take());
    p.A = dy; p.B = wt; p.C = dx;
    p.M = (int)pixels; p.N = cin; p.K = cout; p.Cin = cout;
    p.H = 1; p.W = 1; p.pd = make_pixdiv(1, 1);
    return launch_nt<A_PLAIN, EPI_STORE>(p, (hipStream_t)s, "conv1x1_dgrad");
}

size_t rd_conv1x1_bwd_weight_ws_bytes(long long pixels, int cin, int cout) {
    TnPlan pl = plan_tn(cout, cin, (long)pixels);
    return (size_t)pl.splits * cout * cin * sizeof(float);
}

int rd_conv1x1_bwd_weight(const float* x, const float* dy, float* dw, long long pixels, int cin, int cout, void* ws,
                          size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(x && dy && dw && pixels > 0, "rd_conv1x1_bwd_weight: bad arguments");
    RD_REQUIRE(cin % 4 == 0 && cout % 4 == 0, "rd_conv1x1_bwd_weight: channels must be multiples of 4 (%d, %d)", cin,
               cout);
    const QuantArgs wq = quant_take();       // a: slot of dy, b: slot of x
    const size_t need = rd_conv1x1_bwd_weight_ws_bytes(pixels, cin, cout);
    if (ws_bytes < need || !ws) {
        set_error("rd_conv1x1_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    TnPlan pl = plan_tn(cout, cin, (long)pixels);
    TnParams p = {};
    p.a_amax = wq.a; p.b_amax = wq.b;
    p.A = dy; p.B = x; p.slab = (float*)ws;
    p.M = cout; p.N = cin; p.Kp = (long)pixels;
    p.lda = cout; p.ldb = cin; p.Cin = cin; p.Cout = cout;
    p.H = 1; p.W = 1; p.pd = make_pixdiv(1, 1);
    if (int e = launch_tn<WA_PLAIN, WB_PLAIN>(p, pl, (hipStream_t)s, "conv1x1_wgrad")) return e;
    ProfScope ps((hipStream_t)s, "wgrad_reduce", 0, 4.0 * (pl.splits + 1) * (double)p.M * p.N);
    launch_slab_reduce((const float*)ws, dw, p.M, p.N, pl.splits, 2, cin, cout, (hipStream_t)s);
    RD_LAUNCH_CHECK("slab_reduce");
    return RD_OK;
}

}  // extern "C"
