// NT-kernel parameter block and the epilogues shared by the implicit-GEMM kernels (rd_igemm.hip: exact-f32 / split / halo
// kernels; rd_convt.hip: transposed-convolution data gradient) -- fused BatchNorm forward statistics, inference shift /
// activation / pooling, BN-backward statistics hook, transposed-convolution scatter + bias + skip.  Two forms: the
// register-direct one (nt_epilogue_direct: patch kernels and full plain-row tiles, r04) and the LDS-staged one (16-byte
// row-contiguous stores; ragged tiles, the EPI_CONVT scatter, `nt_epi = 0`).
#pragma once
#include "rd_common.h"
#include "rd_mfma_dev.h"

namespace rd {

__host__ __device__ inline long rows32_of_dev(long rows) { return (rows + 31) / 32 * 32; }

enum { A_CONV3 = 0, A_PLAIN = 1, A_UP2 = 2 };
enum { EPI_STORE = 0, EPI_CONVT = 1 };

struct NtParams {
    const float* A;
    const float* B;        // fp32 GEMM layout [N][K]            (exact-f32 MFMA kernel)
    const void* Bsplit;    // split-bf16 layout [N][nk][3][16]   (split kernel; follows B in the packed buffer)
    float* C;
    const float* bias;
    const float* skip;
    // EPI_CONVT, optional: `skip` holds the PRE-BatchNorm conv output z of the encoder level and the skip value is
    // recomputed here as act(gamma*(z-mean)*invstd + beta) -- the encoder then never writes its full-resolution
    // activation (same arithmetic as bn_act_pool_fwd_kernel, bit-identical values)
    const float* sk_mean;
    const float* sk_invstd;
    const float* sk_gamma;
    const float* sk_beta;
    const float* sk_slope_dev;
    float sk_slope;
    int M, N, K;
    int Cin;  // channels per tap (A row length)
    int H, W;
    PixDiv pd;  // pixel index -> (img, y, x): tiles need not be powers of two
    int Cout;  // EPI_CONVT: channels per (a,b) quadrant
    int chunks, nk, taps;
    int tiles_n;
    int patch;  // halo kernel: 1 = tile rows are an 8x16-pixel patch (row r -> pixel m0 + (r >> 4) * W + (r & 15)); 2 = two 8x8 images
    int vec;  // epilogue may use 16-byte accesses (N % 4 == 0 and, for the transposed conv, Cout % 4 == 0)
    unsigned a_bytes, b_bytes;  // extents of the A / B tensors for the buffer descriptors
    float* stats;  // EPI_STORE only, nullable: per-(tile_m) column sums / sums of squares [tiles_m][2][N] (BN statistics)
    // EPI_STORE, inference with eval-mode BatchNorm folded into the convolution (the weight rows carry gamma*invstd):
    // C = act(acc + shift[n]); pool_out (patch kernels only) additionally receives the 2x2/2 max-pool of C
    const float* shift;
    float act_slope;
    float* pool_out;
    // EPI_STORE, backward: C is the gradient g w.r.t. the activation a = act(BN(z)) of a conv block.  With bn_part != NULL
    // the block's BN-backward statistics come out of this epilogue as per-(tile_m) column sums [tiles_m][4][N] of
    //   g' = g act'(y),  g' xhat,  g,  g y [y <= 0]        (y = gamma xhat + beta, xhat = (z - mean) invstd)
    // -- the arithmetic of bn_act_bwd_kernel<*, false>, whose pass over z and g this replaces.  bn_mode 1: bn_z = z at
    // C's resolution; 2: C is the POOLED gradient and bn_z = z at the arg-max positions (rd_bn_act_pool_fwd zpool): the
    // third sum is then left 0 (it belongs to the un-pooled operand only)
    const float* bn_z;
    const float* bn_mean;
    const float* bn_invstd;
    const float* bn_gamma;
    const float* bn_beta;
    const float* bn_slope_dev;
    float bn_slope;
    int bn_mode;
    float* bn_part;
    // split-K (halo kernel on 8 x 8 images): ksplit blocks per output tile each take chunks_per 16-channel chunks, park their
    // partial accumulators in sk_slab and draw a ticket; the LAST block to arrive adds the ksplit partials in split order
    // (bit-reproducible whatever the arrival order) and runs the epilogue with all its fusions
    int ksplit, chunks_per;
    float* sk_slab;
    unsigned* sk_ticket;
    // training-mode EPI_STORE (no shift / pooling epilogue), N % 32 == 0, patch kernels or plain row tiles with M a multiple
    // of the tile height: the register-direct epilogue (nt_epilogue_direct) instead of the LDS-staged one
    int direct;
    // ---- NP = 3 arithmetic (rd_mfma_dev.h).  a_amax / b_amax: 16-word magnitude slots of the A / B operand tensors, both
    // non-null asks for the three-product body (the kernel still takes the six-product one when a maximum is infinite);
    // Bsplit3 / b_bytes3: the two-term fp16 fragments of B, scaled by the scale of *b_amax (they follow Bsplit in the packed
    // buffer).  out_amax / pool_amax (nullable): slots that receive max |C| / max |pool_out| from the epilogue.
    const unsigned* a_amax;
    const unsigned* b_amax;
    const void* Bsplit3;
    unsigned b_bytes3;
    unsigned* out_amax;
    unsigned* pool_amax;
    // per-image slots (rd_quant_next_img; patch kernels with one image per patch only): a_amax, out_amax, pool_amax are arrays,
    // this many words per image; 0 = one slot per tensor
    int amax_img_stride;
};

// transposed-convolution data gradient (rd_convt.hip); *launched = 0 when the shape is left to the generic NT kernel
int convt_dgrad_launch(NtParams p, hipStream_t s, int* launched, int* tiles_m_out);

__device__ __forceinline__ float nt_act_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }


__device__ __forceinline__ float skip_act(float y, float slope) { return y > 0.f ? y : y * slope; }

// ---- register-direct epilogue of the patch (halo) kernels in training (r04).  The LDS-staged epilogue below costs the 3 x 3
// convolutions 6 % with the BatchNorm statistics and 4 % with the BN-backward hook on top of the plain store, which is
// itself ~1000 VALU instructions per wave and eight barriers (scripts/epilogue_cost.py: enc1 forward 0.383 -> 0.449 ms with
// statistics) -- VALU work that a sibling wave's MFMAs on the same SIMD wait behind.  Here nothing goes through LDS:
//  * C: a lane holds column n = lane & 31 of sixteen rows per 32 x 32 block; lanes 0-31 of a store instruction cover 128
//    contiguous bytes (one cache line) of one pixel, lanes 32-63 the same columns four pixels on.  The row part of the
//    address is wave-uniform (scalar registers), the lane part one 32-bit offset computed once: no address VALU per element;
//  * BatchNorm statistics (forward): column sums straight from the accumulator registers (64 adds + 64 FMAs per 32-column
//    block), halves combined by one cross-lane exchange -- replaces 4 passes of scalar LDS reads with per-row bounds checks;
//  * BN-backward hook (data gradient): z is loaded in the accumulator layout (all loads issued before the C stores), the
//    sums stay in registers, one cross-lane exchange.
// Summation order differs from the staged epilogue's (still fixed: deterministic); rows of a patch are always inside the
// image, only the second image of a two-image patch (patch == 2) can be absent.
template <int BM, int BN, int WM, int WN, int SMEM_WORDS>
__device__ __forceinline__ void nt_epilogue_direct(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem, const NtParams& p,
                                                   int m0, int n0, int tile_m, long pool_base, int amax_off = 0) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, half = lane >> 5;
    const int N = p.N;
    const bool two = p.patch == 2;
    const bool img2_ok = !two || m0 + 64 < p.M;
    // pixel offset (from m0) of accumulator element r of row block i: wave-uniform part; the lane adds 4 * half pixels.
    //   8 x 16 patch:            (tile row >> 4) * W + (tile row & 15)           two 8 x 8 images: ((row & 15) >> 3) * 64 + (row >> 4) * 8 + (row & 7)
    // with tile row = 32 (wm TM + i) + (r & 3) + 8 (r >> 2) + 4 half, i.e. patch row 2 (wm TM + i) + (r >> 3), bit 3 of the column = (r >> 2) & 1
    // plain row tiles (patch == 0, every row inside M: the host only asks for this epilogue then) are the W = 16 case of the formula
    const int SA = two ? 8 : p.patch ? p.W : 16, SB = two ? 64 : 8;   // pixels per patch row / per column-bit-3
    const int rowN = __builtin_amdgcn_readfirstlane(N * 4);
    const int pu0 = __builtin_amdgcn_readfirstlane(2 * wm * TM * SA);
    auto pix_u = [&](int i, int r) { return pu0 + (2 * i + (r >> 3)) * SA + (r & 3) + ((r >> 2) & 1) * SB; };
    const int ncol0 = n0 + wn * TN * 32;                        // first column of this wave (N % 32 == 0: a block is in or out)
    // buffer addressing: descriptor base = first pixel of the tile (wave-uniform), scalar offset = the element's pixel row
    // (wave-uniform, SGPR), lane offset = (4 * half pixels, column) computed once; an absent second image / column block gets
    // the out-of-extent offset, which the hardware drops (stores) or answers with zeros (loads).  A tile spans fewer than
    // 8 * W + 16 pixels: its byte extent stays far below 2^32
    const unsigned span = (unsigned)((two ? 128 : p.patch ? 7 * p.W + 16 : BM) * N * 4);
    const unsigned lane_off = (unsigned)(((4 * half) * N + ncol0 + lrow) * 4);
    unsigned vo1[TN], vo2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        vo1[j] = ncol0 + j * 32 < N ? lane_off + j * 128 : kOOB;
        vo2[j] = img2_ok ? vo1[j] : kOOB;                       // elements of the second image of a two-image patch
    }
    auto soff = [&](int i, int r) { return (unsigned)(pix_u(i, r) * rowN); };
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(p.C + (long)m0 * N, span);
    const bool bn_on = p.bn_part != nullptr;
    const bool st_on = p.stats != nullptr;

    // ---- BN-backward hook: request z first (accumulator layout), its latency hides behind the C stores
    float zr[TM][TN][16];
    if (bn_on) {
        const __amdgpu_buffer_rsrc_t rsZ = make_rsrc(p.bn_z + (long)m0 * N, span);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zr[i][j][r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, ((r >> 2) & 1) ? vo2[j] : vo1[j], soff(i, r), 0));
    }
    // ---- inference (eval-mode BatchNorm folded into the weights): C = act(acc + shift[n]) -- one shift value per lane -- and,
    // on request, MaxPool2d(2, 2) of it from the same registers: the four pixels of a window are elements r, r + 1 (next
    // column) and r + 8, r + 9 (next patch row) of ONE lane, r in {0, 2, 4, 6}; the pooled row of a 32 x 32 block is again
    // 128 contiguous bytes per pixel.  Window order and the NaN rule are torch's max_pool2d (first maximum, NaN wins).
    if (p.shift) {
        const float slope = p.act_slope;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ncol0 + j * 32 + lrow;
            const float sh = n < N ? p.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = skip_act(acc[i][j][r] + sh, slope);
        }
        if (p.pool_out && pool_base >= 0) {
            const int Wp = p.W >> 1;
            float pmax = 0.f;
            const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.pool_out + pool_base * N, (unsigned)((3 * Wp + 8) * N * 4));
            const unsigned lane_off_p = (unsigned)(((2 * half) * N + ncol0 + lrow) * 4);
            const int prow0 = __builtin_amdgcn_readfirstlane(wm * TM * Wp);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 8; r += 2) {
                        float mx = acc[i][j][r];
#pragma unroll
                        for (int k = 1; k < 4; ++k) {
                            const float y = acc[i][j][r + (k & 1) + 8 * (k >> 1)];
                            if (y > mx || y != y) mx = y;
                        }
                        const unsigned so = (unsigned)((prow0 + i * Wp + ((r >> 1) & 1) + 4 * ((r >> 2) & 1)) * rowN);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(mx), rsP, ncol0 + j * 32 < N ? lane_off_p + j * 128 : kOOB, so, 0);
                        pmax = amax_acc(pmax, mx);
                    }
            if (p.pool_amax) amax_commit(p.pool_amax + amax_off, pmax);
        }
    }
    // ---- C
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[i][j][r]), rsC, ((r >> 2) & 1) ? vo2[j] : vo1[j], soff(i, r), 0);
    if (p.out_amax) {                                           // magnitude of C for the GEMM that takes it as an operand
        float cmax = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (img2_ok || !((r >> 2) & 1)) cmax = amax_acc(cmax, acc[i][j][r]);
        amax_commit(p.out_amax + amax_off, cmax);
    }
    // ---- forward BatchNorm statistics: per-(tile_m) column sums / sums of squares [tiles_m][2][N]
    if (st_on) {
        float* red = smem;                                      // WM > 1: [wm][2][BN]
        if (WM > 1) __syncthreads();                            // every wave is done with the operand stages this scratch overlays
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (img2_ok || !((r >> 2) & 1)) ? acc[i][j][r] : 0.f;
                    ss += v;
                    qq = fmaf(v, v, qq);
                }
            ss += __shfl_xor(ss, 32);
            qq += __shfl_xor(qq, 32);
            const int col = wn * TN * 32 + j * 32 + lrow;        // column inside the block tile
            if (WM == 1) {
                if (half == 0 && n0 + col < N) {
                    float* out = p.stats + (long)tile_m * 2 * N + n0 + col;
                    out[0] = ss;
                    out[N] = qq;
                }
            } else if (half == 0) {
                red[(wm * 2 + 0) * BN + col] = ss;
                red[(wm * 2 + 1) * BN + col] = qq;
            }
        }
        if (WM > 1) {
            static_assert(WM * 2 * BN <= SMEM_WORDS, "statistics scratch must fit the operand buffers");
            __syncthreads();
            if (t < BN && n0 + t < N) {
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WM; ++w2) {
                    ss += red[(w2 * 2 + 0) * BN + t];
                    qq += red[(w2 * 2 + 1) * BN + t];
                }
                float* out = p.stats + (long)tile_m * 2 * N + n0 + t;
                out[0] = ss;
                out[N] = qq;
            }
        }
    }
    // ---- BN-backward statistics [tile_m][4][N]: g' = g act'(y), g' xhat, g (mode 1), g y [y <= 0]
    if (bn_on) {
        const float bslope = p.bn_slope_dev ? p.bn_slope_dev[0] : p.bn_slope;
        const bool mode1 = p.bn_mode == 1;
        float* red = smem;                                      // WM > 1: [wm][4][BN]
        if (WM > 1) __syncthreads();                            // operand stages / the statistics scratch above are done with
        // 10 VALU per element: y, y > 0, g * slope, select, xhat as one FMA, three accumulations, and the fourth sum (the
        // learnable slope's gradient: select + FMA; part of the entry points' contract whatever the activation)
        {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * TN * 32 + j * 32 + lrow;
                const int n = n0 + col;
                float sc = 0.f, sh = 0.f, is = 0.f, nmi = 0.f;
                if (n < N) {
                    const float mu = p.bn_mean[n];
                    is = p.bn_invstd[n];
                    sc = is * p.bn_gamma[n];
                    sh = p.bn_beta[n] - mu * sc;
                    nmi = -mu * is;
                }
                float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool ok = img2_ok || !((r >> 2) & 1);
                        const float g = ok ? acc[i][j][r] : 0.f, z = zr[i][j][r];
                        const float y = fmaf(z, sc, sh);
                        const float gm = y > 0.f ? g : g * bslope;
                        const float xh = fmaf(z, is, nmi);
                        b0 += gm;
                        b1 = fmaf(gm, xh, b1);
                        b2 += g;
                        if (!(y > 0.f)) b3 = fmaf(g, y, b3);
                    }
                if (!mode1) b2 = 0.f;
                b0 += __shfl_xor(b0, 32);
                b1 += __shfl_xor(b1, 32);
                b2 += __shfl_xor(b2, 32);
                b3 += __shfl_xor(b3, 32);
                if (WM == 1) {
                    if (half == 0 && n < N) {
                        float* out = p.bn_part + (long)tile_m * 4 * N + n;
                        out[0] = b0;
                        out[N] = b1;
                        out[2 * N] = b2;
                        out[3 * N] = b3;
                    }
                } else if (half == 0) {
                    red[(wm * 4 + 0) * BN + col] = b0;
                    red[(wm * 4 + 1) * BN + col] = b1;
                    red[(wm * 4 + 2) * BN + col] = b2;
                    red[(wm * 4 + 3) * BN + col] = b3;
                }
            }
        }
        if (WM > 1) {
            static_assert(WM * 4 * BN <= SMEM_WORDS, "statistics scratch must fit the operand buffers");
            __syncthreads();
            for (int o = t; o < 4 * BN; o += 256) {
                const int sidx = o / BN, col = o - sidx * BN;
                float sum = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WM; ++w2) sum += red[(w2 * 4 + sidx) * BN + col];
                if (n0 + col < N) p.bn_part[((long)tile_m * 4 + sidx) * N + n0 + col] = sum;
            }
        }
    }
}

// ---- epilogue shared by the NT kernels.  D[i][j]: lane -> column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5) (the C/D
// map is the same for the f32 and the bf16 MFMA shapes).
template <int BM, int BN, int WM, int WN, int EPI, int SMEM_WORDS, int EB = BM / WM / 32>
__device__ __forceinline__ void nt_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem, const NtParams& p,
                                            int m0, int n0, int tile_m, long pool_base = -1, int amax_off = 0) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    if constexpr (EPI == EPI_STORE) {
        if (p.direct) {
            nt_epilogue_direct<BM, BN, WM, WN, SMEM_WORDS>(acc, smem, p, m0, n0, tile_m, pool_base, amax_off);
            return;
        }
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, half = lane >> 5;
    const int H = p.H, W = p.W;
    // ---- epilogue.  D[i][j]: lane -> column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5).  The accumulators
    // are staged through LDS (EB 32-row blocks of one wave row-band per pass) so that HBM sees 16 B per lane and whole
    // contiguous rows per wave; bias / skip-add of the transposed convolution ride the same pass.
    // tile row -> pixel: plain | 8 x 16 patch of one image | two 8 x 8 images side by side (m0 = first pixel of the pair)
    auto row_to_m = [&](int r) {
        return p.patch == 2 ? m0 + ((r & 15) >> 3) * 64 + (r >> 4) * 8 + (r & 7) : p.patch ? m0 + (r >> 4) * W + (r & 15) : m0 + r;
    };
    constexpr int CS = BN + 4, ROWS = EB * 32, Q = BN / 4, PPB = TM / EB;   // PPB passes per wave row-band
    static_assert(TM % EB == 0, "EB must divide TM");
    static_assert(ROWS * CS + 512 <= SMEM_WORDS, "epilogue staging (+ statistics scratch) must fit the operand buffers");
    float* Cs = smem;
    float* red = smem + ROWS * CS;          // 2 x 256 floats for the fused BatchNorm statistics
    float tot_s = 0.f, tot_q = 0.f;         // threads t < BN: column totals over the passes
    // BN-backward statistics hook: a thread keeps one column quad (256 % Q == 0) over all passes
    static_assert(256 % Q == 0, "a thread must keep its column quad across the store loop");
    const bool bn_on = EPI == EPI_STORE && p.bn_part != nullptr;
    float bsc[4] = {0, 0, 0, 0}, bsh[4] = {0, 0, 0, 0}, bmu[4] = {0, 0, 0, 0}, bis[4] = {0, 0, 0, 0}, bacc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) bacc[k] = 0.f;
    float bslope = 0.f;
    float cmax = 0.f, pmax = 0.f;           // max |stored value| of this thread (p.out_amax / p.pool_amax)
    if (bn_on) {
        bslope = p.bn_slope_dev ? p.bn_slope_dev[0] : p.bn_slope;
        const int n = n0 + (t % Q) * 4;
        if (n < p.N) {
            const float4 m4 = *reinterpret_cast<const float4*>(p.bn_mean + n), i4 = *reinterpret_cast<const float4*>(p.bn_invstd + n);
            const float4 g4 = *reinterpret_cast<const float4*>(p.bn_gamma + n), b4 = *reinterpret_cast<const float4*>(p.bn_beta + n);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bmu[q] = mm[q];
                bis[q] = ii[q];
                bsc[q] = ii[q] * gg[q];
                bsh[q] = bb[q] - mm[q] * bsc[q];
            }
        }
    }
#pragma unroll
    for (int pass = 0; pass < WM * PPB; ++pass) {
        const int rowbase = (pass / PPB) * (TM * 32) + (pass % PPB) * ROWS;   // first tile row of this pass
        // BN-backward hook: this pass's z values are requested before the accumulators go through LDS, so their HBM
        // latency hides behind the staging (element e = t + 256 k of the store loop below)
        constexpr int NIT = (ROWS * Q + 255) / 256;
        float4 zpre[NIT];
        if (bn_on) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int e = t + 256 * k, row = e / Q, q4 = e - row * Q;
                const int m = row_to_m(rowbase + row), n = n0 + q4 * 4;
                zpre[k] = (e < ROWS * Q && m < p.M && n < p.N) ? *reinterpret_cast<const float4*>(p.bn_z + (long)m * p.N + n)
                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        if (wm == pass / PPB) {
#pragma unroll
            for (int ii = 0; ii < EB; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CS + wn * TN * 32 + j * 32 + lrow] =
                            acc[(pass % PPB) * EB + ii][j][r];
        }
        __syncthreads();
        if (EPI == EPI_STORE && p.stats) {
            // fused BN statistics: column sums over this pass's rows (fixed order), combined over the row groups
            constexpr int G = 256 / BN;      // row groups
            const int col = t % BN, grp = t / BN;
            float ss = 0.f, qq = 0.f;
            for (int row = grp; row < ROWS; row += G) {
                if (row_to_m(rowbase + row) < p.M) {
                    const float v = Cs[row * CS + col];
                    ss += v;
                    qq = fmaf(v, v, qq);
                }
            }
            red[t] = ss;
            red[256 + t] = qq;
            __syncthreads();
            if (t < BN) {
#pragma unroll
                for (int g2 = 0; g2 < G; ++g2) {
                    tot_s += red[g2 * BN + t];
                    tot_q += red[256 + g2 * BN + t];
                }
            }
        }
        if (p.vec) {
#pragma unroll
            for (int kk = 0; kk < NIT; ++kk) {
                const int e = t + 256 * kk;
                if (e >= ROWS * Q) break;
                const int row = e / Q, q4 = e - row * Q;
                const int m = row_to_m(rowbase + row), n = n0 + q4 * 4;
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(&Cs[row * CS + q4 * 4]);
                if (EPI == EPI_STORE) {
                    if (p.shift) {
                        const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + n);
                        v.x = skip_act(v.x + sh4.x, p.act_slope); v.y = skip_act(v.y + sh4.y, p.act_slope);
                        v.z = skip_act(v.z + sh4.z, p.act_slope); v.w = skip_act(v.w + sh4.w, p.act_slope);
                    }
                    *reinterpret_cast<float4*>(p.C + (long)m * p.N + n) = v;
                    cmax = amax_acc(amax_acc(amax_acc(amax_acc(cmax, v.x), v.y), v.z), v.w);
                    if (bn_on) {
                        const float4 z4 = zpre[kk];
                        const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, gv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float y = fmaf(zz[q], bsc[q], bsh[q]);
                            const float gm = gv[q] * nt_act_grad(y, bslope);
                            const float xh = (zz[q] - bmu[q]) * bis[q];
                            bacc[q] += gm;
                            bacc[4 + q] = fmaf(gm, xh, bacc[4 + q]);
                            if (p.bn_mode == 1) bacc[8 + q] += gv[q];
                            if (!(y > 0.f)) bacc[12 + q] = fmaf(gv[q], y, bacc[12 + q]);
                        }
                    }
                } else {
                    const int ab = n / p.Cout, co = n - ab * p.Cout;
                    int jj, ii, img;
                    pix_split(m, p.pd, img, ii, jj);
                    const long opix = ((long)img * (2 * H) + 2 * ii + (ab >> 1)) * (2 * W) + 2 * jj + (ab & 1);
                    const long o = opix * p.Cout + co;
                    if (p.bias) {
                        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
                    if (p.skip) {
                        float4 s4 = *reinterpret_cast<const float4*>(p.skip + o);
                        if (p.sk_mean) {
                            const float4 mu = *reinterpret_cast<const float4*>(p.sk_mean + co);
                            const float4 is = *reinterpret_cast<const float4*>(p.sk_invstd + co);
                            const float4 ga = *reinterpret_cast<const float4*>(p.sk_gamma + co);
                            const float4 be = *reinterpret_cast<const float4*>(p.sk_beta + co);
                            const float sl = p.sk_slope_dev ? p.sk_slope_dev[0] : p.sk_slope;
                            const float sc0 = is.x * ga.x, sc1 = is.y * ga.y, sc2 = is.z * ga.z, sc3 = is.w * ga.w;
                            s4.x = skip_act(fmaf(s4.x, sc0, be.x - mu.x * sc0), sl);
                            s4.y = skip_act(fmaf(s4.y, sc1, be.y - mu.y * sc1), sl);
                            s4.z = skip_act(fmaf(s4.z, sc2, be.z - mu.z * sc2), sl);
                            s4.w = skip_act(fmaf(s4.w, sc3, be.w - mu.w * sc3), sl);
                        }
                        v.x = s4.x + v.x; v.y = s4.y + v.y; v.z = s4.z + v.z; v.w = s4.w + v.w;
                    }
                    *reinterpret_cast<float4*>(p.C + o) = v;
                    cmax = amax_acc(amax_acc(amax_acc(amax_acc(cmax, v.x), v.y), v.z), v.w);
                }
            }
            if (EPI == EPI_STORE && p.pool_out && pool_base >= 0) {
                // this pass = two patch rows x 16 pixels = 8 complete 2x2 windows per channel quad (ROWS = 32)
                for (int e = t; e < 8 * Q; e += 256) {
                    const int j = e / Q, q4 = e - j * Q, n = n0 + q4 * 4;
                    if (n >= p.N) continue;
                    const float4 sh4 = *reinterpret_cast<const float4*>(p.shift + n);
                    float mx[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 c4 = *reinterpret_cast<const float4*>(&Cs[((k >> 1) * 16 + 2 * j + (k & 1)) * CS + q4 * 4]);
                        const float y[4] = {skip_act(c4.x + sh4.x, p.act_slope), skip_act(c4.y + sh4.y, p.act_slope),
                                            skip_act(c4.z + sh4.z, p.act_slope), skip_act(c4.w + sh4.w, p.act_slope)};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (k == 0 || y[q] > mx[q] || y[q] != y[q]) mx[q] = y[q];      // torch max_pool2d: NaN wins
                    }
                    // pooled pixel: patch origin + (pass, j); rowbase / 32 = pass index = pooled row inside the patch
                    const long pp = pool_base + (long)(rowbase >> 5) * (p.W >> 1) + j;
                    *reinterpret_cast<float4*>(p.pool_out + pp * p.N + n) = make_float4(mx[0], mx[1], mx[2], mx[3]);
                    pmax = amax_acc(amax_acc(amax_acc(amax_acc(pmax, mx[0]), mx[1]), mx[2]), mx[3]);
                }
            }
        } else {
            for (int e = t; e < ROWS * BN; e += 256) {
                const int row = e / BN, c = e - row * BN;
                const int m = row_to_m(rowbase + row), n = n0 + c;
                if (m >= p.M || n >= p.N) continue;
                float v = Cs[row * CS + c];
                if (EPI == EPI_STORE) {
                    if (p.shift) v = skip_act(v + p.shift[n], p.act_slope);
                    p.C[(long)m * p.N + n] = v;
                    cmax = amax_acc(cmax, v);
                } else {
                    const int ab = n / p.Cout, co = n - ab * p.Cout;
                    int jj, ii, img;
                    pix_split(m, p.pd, img, ii, jj);
                    const long opix = ((long)img * (2 * H) + 2 * ii + (ab >> 1)) * (2 * W) + 2 * jj + (ab & 1);
                    const long o = opix * p.Cout + co;
                    if (p.bias) v += p.bias[co];
                    if (p.skip) {
                        float sv = p.skip[o];
                        if (p.sk_mean) {
                            const float sc0 = p.sk_invstd[co] * p.sk_gamma[co];
                            sv = skip_act(fmaf(sv, sc0, p.sk_beta[co] - p.sk_mean[co] * sc0),
                                          p.sk_slope_dev ? p.sk_slope_dev[0] : p.sk_slope);
                        }
                        v = sv + v;
                    }
                    p.C[o] = v;
                    cmax = amax_acc(cmax, v);
                }
            }
        }
    }
    if (p.out_amax) amax_commit(p.out_amax + amax_off, cmax);
    if (EPI == EPI_STORE && p.pool_amax && p.pool_out) amax_commit(p.pool_amax + amax_off, pmax);
    if (EPI == EPI_STORE && p.stats && t < BN && n0 + t < p.N) {
        float* out = p.stats + (long)tile_m * 2 * p.N;
        out[n0 + t] = tot_s;
        out[p.N + n0 + t] = tot_q;
    }
    if (bn_on) {
        // combine the 256 / Q threads of each column quad in a fixed order; [tile_m][4][N]
        static_assert(256 * 16 <= SMEM_WORDS, "BN-backward statistics scratch must fit the operand buffers");
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) smem[t * 16 + k] = bacc[k];
        __syncthreads();
        for (int o = t; o < 4 * BN; o += 256) {
            const int sidx = o / BN, col = o - sidx * BN;
            float sum = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 256 / Q; ++g2) sum += smem[(g2 * Q + (col >> 2)) * 16 + sidx * 4 + (col & 3)];
            if (n0 + col < p.N) p.bn_part[((long)tile_m * 4 + sidx) * p.N + n0 + col] = sum;
        }
    }
}

}  // namespace rd
