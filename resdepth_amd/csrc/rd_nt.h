// NT-kernel parameter block and the epilogue shared by the implicit-GEMM kernels (rd_igemm.hip: exact-f32 / split / halo
// kernels) -- staging through LDS for 16-byte row-contiguous stores, fused BatchNorm
// forward statistics, transposed-convolution scatter + bias + skip, inference shift / activation / pooling, BN-backward
// statistics hook.
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
};

// transposed-convolution data gradient (rd_convt.hip); *launched = 0 when the shape is left to the generic NT kernel
int convt_dgrad_launch(NtParams p, hipStream_t s, int* launched, int* tiles_m_out);

__device__ __forceinline__ float nt_act_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }


__device__ __forceinline__ float skip_act(float y, float slope) { return y > 0.f ? y : y * slope; }

// ---- epilogue shared by the NT kernels.  D[i][j]: lane -> column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5) (the C/D
// map is the same for the f32 and the bf16 MFMA shapes).
template <int BM, int BN, int WM, int WN, int EPI, int SMEM_WORDS, int EB = BM / WM / 32>
__device__ __forceinline__ void nt_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem, const NtParams& p,
                                            int m0, int n0, int tile_m, long pool_base = -1) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
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
                }
            }
        }
    }
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
