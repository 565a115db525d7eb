// HBM-bound kernels of the ResDepth hot path (gfx950): BatchNorm statistics / apply (+ activation
// + 2x2 max-pool with argmax) forward and backward, the Cin<=6 first convolution, the C->1 last
// convolution with the outer residual add, the masked de-normalised L1 loss, the flat Adam step
// and layout helpers.  Replaces the ATen kernels behind nn.BatchNorm2d / nn.ReLU / nn.LeakyReLU /
// nn.MaxPool2d (lib/UNet.py:27-33, 45, 66, 86, 161, 167), the first / last nn.Conv2d
// (lib/UNet.py:159, 184), Trainer._compute_denormalized_loss (lib/Trainer.py:87-100) and
// torch.optim.Adam.step (lib/utils.py:329-331).
//
// Mapping used by every NHWC kernel here: a thread owns one float4 of channels (cq = t % CQ,
// CQ = C/4) of one pixel row (pr = t / CQ); consecutive lanes read consecutive 16-byte chunks, so a
// wave touches 1 KiB of contiguous HBM per load instruction.  Per-channel reductions accumulate in
// fp64 and go through fixed-order block partials (deterministic, no float atomics).
#include <stdlib.h>

#include "rd_common.h"
#include "rd_mfma_dev.h"

namespace rd {

static inline int grid_cap(long g, int cap) {
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float act_fn(float y, float slope) { return y > 0.f ? y : y * slope; }
__device__ __forceinline__ float act_grad(float y, float slope) { return y > 0.f ? 1.f : slope; }

// ---- block-level fixed-order reduction over the pixel-row slots of a block --------------------
// vals[K] per thread (K = 4 * quantities); red = LDS [K][256] doubles; result valid for t < CQ.
template <int K>
__device__ __forceinline__ void reduce_rows(double (&vals)[K], double* red, int t, int CQ, int RP, bool active) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[k * 256 + t] = active ? vals[k] : 0.0;
    __syncthreads();
    if (t < CQ) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double s = 0.0;
            for (int r = 0; r < RP; ++r) s += red[k * 256 + r * CQ + t];
            vals[k] = s;
        }
    }
    __syncthreads();
}

// Second-stage reduction helper: one block = 16 columns x 16 slices of the partial rows; every thread sums its
// slice with four loads in flight, the 16 slice sums are combined in fixed order (deterministic).  Result valid
// for threadIdx.x < 16 (column = blockIdx.x*16 + threadIdx.x).
template <typename T>
__device__ __forceinline__ double sliced_column_sum(const T* __restrict__ partial, int nb, long stride, int ncols,
                                                    double* red) {
    const int t = threadIdx.x, slice = t >> 4, col = blockIdx.x * 16 + (t & 15);
    double s = 0.0;
    if (col < ncols) {
        int b = slice;
        for (; b + 48 < nb; b += 64) {
            const T v0 = partial[(long)b * stride + col], v1 = partial[(long)(b + 16) * stride + col];
            const T v2 = partial[(long)(b + 32) * stride + col], v3 = partial[(long)(b + 48) * stride + col];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nb; b += 16) s += (double)partial[(long)b * stride + col];
    }
    red[t] = s;
    __syncthreads();
    double r = 0.0;
    if (t < 16) {
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) r += red[sl * 16 + t];
    }
    return r;
}

// sums[q*C + c] = sum_b partial[(b*Q + q)*C + c]
// dbeta / dgamma / dextra (nullable): the first three C-column groups of the BN-backward sums are parameter gradients
// (sum g', sum g' * xhat, sum g_full = the bias gradient of the ConvTranspose2d feeding the skip add) -- written here
// instead of by separate launches
__global__ __launch_bounds__(256) void partial_reduce_kernel(const double* __restrict__ partial,
                                                             double* __restrict__ sums, int nb, int QC, int C = 0,
                                                             float* __restrict__ dbeta = nullptr,
                                                             float* __restrict__ dgamma = nullptr,
                                                             float* __restrict__ dextra = nullptr) {
    __shared__ double red[256];
    const double r = sliced_column_sum<double>(partial, nb, QC, QC, red);
    const int col = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && col < QC) {
        sums[col] = r;
        if (dbeta && col < C) dbeta[col] = (float)r;
        if (dgamma && col >= C && col < 2 * C) dgamma[col - C] = (float)r;
        if (dextra && col >= 2 * C && col < 3 * C) dextra[col - 2 * C] = (float)r;
    }
}

__global__ __launch_bounds__(256) void partial_reduce_f32_kernel(const float* __restrict__ partial,
                                                                 double* __restrict__ sums, int nb, int QC) {
    __shared__ double red[256];
    const double r = sliced_column_sum<float>(partial, nb, QC, QC, red);
    const int col = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && col < QC) sums[col] = r;
}

int reduce_partials_f32(const float* partial, double* sums, int nb, int qc, hipStream_t s) {
    RD_LAUNCH(partial_reduce_f32_kernel, dim3(cdiv(qc, 16)), dim3(256), 0, s, partial, sums, nb, qc);
    RD_LAUNCH_CHECK("partial_reduce_f32");
    return RD_OK;
}

struct RowPlan {
    int CQ, RP, nb;
    long rows_per_block;
};

static bool plan_rows(long rows, int C, RowPlan* pl, int maxblocks = 0) {
    if (maxblocks <= 0) maxblocks = tune(TUNE_ROWS_BLOCKS);
    if (C % 4 != 0 || C / 4 > 256 || C <= 0) return false;
    pl->CQ = C / 4;
    pl->RP = 256 / pl->CQ;
    long rpb = (rows + maxblocks - 1) / maxblocks;   // <= 512 first-stage blocks by default; the row loops are unrolled for MLP
    const long minr = (long)pl->RP * 8;
    if (rpb < minr) rpb = minr;
    rpb = (rpb + pl->RP - 1) / pl->RP * pl->RP;
    pl->rows_per_block = rpb;
    pl->nb = (int)((rows + rpb - 1) / rpb);
    if (pl->nb < 1) pl->nb = 1;
    return true;
}

// ---- per-channel sum / sum of squares ---------------------------------------------------------
template <bool SQ>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ z, double* __restrict__ partial,
                                                            long P, int C, int CQ, int RP, long rows_per_block) {
    __shared__ double red[8 * 256];
    const int t = threadIdx.x, cq = t % CQ, pr = t / CQ;
    const bool active = pr < RP;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.0;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > P) r1 = P;
    if (active) {
        // 8 independent 16-byte loads in flight per lane; fp32 partials over 8 rows, fp64 across groups
        for (long r = r0 + pr; r < r1; r += 8L * RP) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long rr = r + (long)u * RP;
                x[u] = rr < r1 ? *reinterpret_cast<const float4*>(z + rr * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s0 += x[u].x; s1 += x[u].y; s2 += x[u].z; s3 += x[u].w;
                if (SQ) {
                    q0 = fmaf(x[u].x, x[u].x, q0); q1 = fmaf(x[u].y, x[u].y, q1);
                    q2 = fmaf(x[u].z, x[u].z, q2); q3 = fmaf(x[u].w, x[u].w, q3);
                }
            }
            v[0] += s0; v[1] += s1; v[2] += s2; v[3] += s3;
            if (SQ) { v[4] += q0; v[5] += q1; v[6] += q2; v[7] += q3; }
        }
    }
    reduce_rows<8>(v, red, t, CQ, RP, active);
    if (t < CQ) {
        const int Q = SQ ? 2 : 1;
        double* out = partial + (long)blockIdx.x * Q * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out[cq * 4 + k] = v[k];
            if (SQ) out[C + cq * 4 + k] = v[4 + k];
        }
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                   float* __restrict__ rvar, int64_t* nbt, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
    if (rvar) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// BatchNorm batch statistics in ONE launch: per-tile partials [nb][2][C] (fp32, written by the convolution epilogues) ->
// fixed-order fp64 column sums -> mean / invstd / running statistics (the arithmetic of bn_finalize_kernel).  One block =
// 4 channels x (sum, sum of squares) x 32 row slices; replaces partial_reduce_f32_kernel + bn_finalize_kernel.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ partial, int nb, int C, double count,
                                                                 float eps, float momentum, float* __restrict__ mean,
                                                                 float* __restrict__ invstd, float* __restrict__ rmean,
                                                                 float* __restrict__ rvar, int64_t* nbt) {
    __shared__ double red[256];
    const int t = threadIdx.x, j = t & 7, slice = t >> 3;          // j: 0..3 sums, 4..7 sums of squares
    const int c = blockIdx.x * 4 + (j & 3);
    const long col = (j < 4 ? 0 : C) + c;
    double sacc = 0.0;
    if (c < C) {
        int b = slice;
        for (; b + 96 < nb; b += 128) {
            const float v0 = partial[(long)b * 2 * C + col], v1 = partial[(long)(b + 32) * 2 * C + col];
            const float v2 = partial[(long)(b + 64) * 2 * C + col], v3 = partial[(long)(b + 96) * 2 * C + col];
            sacc += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; b < nb; b += 32) sacc += (double)partial[(long)b * 2 * C + col];
    }
    red[t] = sacc;
    __syncthreads();
    if (t == 0 && blockIdx.x == 0 && nbt) *nbt += 1;
    if (t < 4 && c < C) {
        double su = 0.0, sq = 0.0;
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) {
            su += red[sl * 8 + t];
            sq += red[sl * 8 + 4 + t];
        }
        const double m = su / count;
        double var = sq / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        if (rvar) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
}

// BN-backward statistics from the dgrad epilogues: per-tile partials [rows][4][C] (fp32) of one or two producers (the
// un-pooled and the pooled gradient operand) -> fixed-order fp64 column sums `sums[4C]` (layout of bn_act_bwd_kernel's
// reduction) + the fp32 parameter gradients.  One block = one channel quad of one of the four sums; its 256 threads are 256
// row slices (16-byte loads, four rows per thread in flight), combined by a fixed shuffle butterfly per wave and 128 bytes
// of LDS across the four waves.  This launch sits on the critical path of the backward (data gradient -> statistics -> BN
// apply): the r02 version -- one wave per quad, two loads in flight, sized to squeeze in beside a two-waves-per-SIMD
// weight-gradient kernel -- took 50-72 us at the 256 x 256 levels (8192 partial rows); the weight-gradient stream now runs
// one block per CU and leaves room for a real block.
__global__ __launch_bounds__(256) void bn_bwd_stats_finalize_kernel(const float* __restrict__ pa, int ra,
                                                                    const float* __restrict__ pb, int rb, int C,
                                                                    double* __restrict__ sums, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, float* __restrict__ dextra) {
    __shared__ double wred[4][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, CQ = C >> 2;
    const int sidx = blockIdx.x / CQ, cq = blockIdx.x - sidx * CQ;
    const long col = (long)sidx * C + cq * 4, stride = 4L * C;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int src = 0; src < 2; ++src) {
        const float* part = src ? pb : pa;
        const int nb = src ? rb : ra;
        if (!part) continue;
        int b = t;
        for (; b + 768 < nb; b += 1024) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (long)b * stride + col);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (long)(b + 256) * stride + col);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (long)(b + 512) * stride + col);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (long)(b + 768) * stride + col);
            a0 += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
            a1 += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
            a2 += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
            a3 += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
        }
        for (; b < nb; b += 256) {
            const float4 v = *reinterpret_cast<const float4*>(part + (long)b * stride + col);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
        a3 += __shfl_xor(a3, o);
    }
    if (lane == 0) {
        wred[wave][0] = a0; wred[wave][1] = a1; wred[wave][2] = a2; wred[wave][3] = a3;
    }
    __syncthreads();
    if (t < 4) {
        const double r = ((wred[0][t] + wred[1][t]) + wred[2][t]) + wred[3][t];
        const int c = cq * 4 + t;
        sums[col + t] = r;
        if (sidx == 0 && dbeta) dbeta[c] = (float)r;
        if (sidx == 1 && dgamma) dgamma[c] = (float)r;
        if (sidx == 2 && dextra) dextra[c] = (float)r;
    }
}

// Same, C % 4 == 0 (every layer of the network): one block = one channel quad, 512 threads = 256 row slices x (sums | sums of
// squares), 16-byte loads, eight rows per thread in flight -- the partials of a 256 x 256 level are 4096 rows, and this launch
// sits between the convolution and its BN / activation pass with nothing to overlap it in the forward.
__global__ __launch_bounds__(512) void bn_reduce_finalize_quad_kernel(const float* __restrict__ partial, int nb, int C, double count,
                                                                      float eps, float momentum, float* __restrict__ mean,
                                                                      float* __restrict__ invstd, float* __restrict__ rmean,
                                                                      float* __restrict__ rvar, int64_t* nbt) {
    __shared__ double red[512][4];
    const int t = threadIdx.x, kind = t & 1, slice = t >> 1, cq = blockIdx.x;
    const float* base = partial + (long)kind * C + cq * 4;
    const long stride = 2L * C;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int b = slice;
    for (; b + 7 * 256 < nb; b += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(base + (long)(b + 256 * k) * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a0 += v[k].x; a1 += v[k].y; a2 += v[k].z; a3 += v[k].w;
        }
    }
    for (; b < nb; b += 256) {
        const float4 v = *reinterpret_cast<const float4*>(base + (long)b * stride);
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
    }
    red[t][0] = a0; red[t][1] = a1; red[t][2] = a2; red[t][3] = a3;
    __syncthreads();
    for (int sl = 128; sl >= 1; sl >>= 1) {           // fixed tree over the 256 slices (thread = 2 * slice + kind)
        if (slice < sl) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[t][k] += red[t + 2 * sl][k];
        }
        __syncthreads();
    }
    if (t == 0 && blockIdx.x == 0 && nbt) *nbt += 1;
    if (t < 4) {
        const int c = cq * 4 + t;
        const double su = red[0][t], sq = red[1][t];
        const double m = su / count;
        double var = sq / count - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)m;
        if (rvar) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    }
}

int bn_reduce_finalize(const float* partial, int nb, int c, double count, float eps, float momentum, float* mean, float* invstd,
                       float* rmean, float* rvar, int64_t* nbt, hipStream_t s) {
    if (c % 4 == 0)
        RD_LAUNCH(bn_reduce_finalize_quad_kernel, dim3(c / 4), dim3(512), 0, s, partial, nb, c, count, eps, momentum, mean,
                           invstd, rmean, rvar, nbt);
    else
        RD_LAUNCH(bn_reduce_finalize_kernel, dim3(cdiv(c, 4)), dim3(256), 0, s, partial, nb, c, count, eps, momentum, mean,
                           invstd, rmean, rvar, nbt);
    RD_LAUNCH_CHECK("bn_reduce_finalize");
    return RD_OK;
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

// ---- BN apply + activation (+ 2x2 max-pool with argmax) ---------------------------------------
template <bool POOL>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope_val,
                                                              const float* __restrict__ slope_dev,
                                                              float* __restrict__ a, float* __restrict__ pooled,
                                                              uint8_t* __restrict__ idx, float* __restrict__ zpool, long rows,
                                                              int H, int W, int C, int CQ, unsigned* a_amax, unsigned* p_amax) {
    // rows = pooled pixels (POOL) or pixels; one thread per (row, cq).  zpool (nullable): z at the arg-max position -- the
    // pooled part of the BN-backward statistics is then a pass over quarter-size tensors (rd_conv3x3_bwd_data_bnstats)
    const float slope = slope_dev ? slope_dev[0] : slope_val;   // PReLU: learnable slope read on the device
    const long total = rows * CQ;
    float amx = 0.f, pmx = 0.f;     // a_amax / p_amax (nullable): magnitude slots of a / pooled (operands of three-product GEMMs)
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(e % CQ);
        const long row = e / CQ;
        const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + cq * 4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + cq * 4);
        const float4 be = *reinterpret_cast<const float4*>(beta + cq * 4);
        const float sc[4] = {is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w};
        const float sh[4] = {be.x - mu.x * sc[0], be.y - mu.y * sc[1], be.z - mu.z * sc[2], be.w - mu.w * sc[3]};
        if (!POOL) {
            const float4 x = ld_nt4(z + row * C + cq * 4);
            float4 y;
            y.x = act_fn(fmaf(x.x, sc[0], sh[0]), slope);
            y.y = act_fn(fmaf(x.y, sc[1], sh[1]), slope);
            y.z = act_fn(fmaf(x.z, sc[2], sh[2]), slope);
            y.w = act_fn(fmaf(x.w, sc[3], sh[3]), slope);
            *reinterpret_cast<float4*>(a + row * C + cq * 4) = y;
            amx = amax_acc(amax_acc(amax_acc(amax_acc(amx, y.x), y.y), y.z), y.w);
        } else {
            const int W2 = W >> 1, H2 = H >> 1;
            const int j = (int)(row % W2);
            const int i = (int)((row / W2) % H2);
            const long img = row / ((long)W2 * H2);
            const long base = (img * H + 2 * i) * W + 2 * j;
            float m[4], zm[4] = {0.f, 0.f, 0.f, 0.f};
            int mi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long pix = base + (k >> 1) * W + (k & 1);
                const float4 x = ld_nt4(z + pix * C + cq * 4);        // z: read once here, next in the backward
                const float xv[4] = {x.x, x.y, x.z, x.w};
                float y[4];
                y[0] = act_fn(fmaf(x.x, sc[0], sh[0]), slope);
                y[1] = act_fn(fmaf(x.y, sc[1], sh[1]), slope);
                y[2] = act_fn(fmaf(x.z, sc[2], sh[2]), slope);
                y[3] = act_fn(fmaf(x.w, sc[3], sh[3]), slope);
                if (a) *reinterpret_cast<float4*>(a + pix * C + cq * 4) = make_float4(y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // torch max_pool2d: (val > maxval) || isnan(val) replaces -> first max wins, NaN wins
                    if (k == 0 || y[q] > m[q] || y[q] != y[q]) {
                        m[q] = y[q];
                        mi[q] = k;
                        zm[q] = xv[q];
                    }
                }
            }
            *reinterpret_cast<float4*>(pooled + row * C + cq * 4) = make_float4(m[0], m[1], m[2], m[3]);
            pmx = amax_acc(amax_acc(amax_acc(amax_acc(pmx, m[0]), m[1]), m[2]), m[3]);
            if (zpool) st_nt4(zpool + row * C + cq * 4, make_float4(zm[0], zm[1], zm[2], zm[3]));      // read in the backward
            st_nt_u8x4(idx + row * C + cq * 4,
                       make_uchar4((unsigned char)mi[0], (unsigned char)mi[1], (unsigned char)mi[2], (unsigned char)mi[3]));
        }
    }
    // (with pooling, max |a| = max |pooled| whenever a maximum exists: the pooled value IS the window's largest, and the slot of
    // `a` is only asked for without pooling)
    if (a_amax) amax_commit(a_amax, POOL ? pmx : amx);
    if (p_amax) amax_commit(p_amax, pmx);
}

// ---- BN + activation (+ pool) backward ---------------------------------------------------------
// g(a) = g_full (nullable) + unpool(g_pool) (nullable).  POOL: rows are pooled pixels, each thread
// visits the 2x2 window; otherwise rows are pixels.
template <bool POOL, bool APPLY>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float slope_val, const float* __restrict__ slope_dev,
                                                         const float* __restrict__ g_full,
                                                         const float* __restrict__ g_pool,
                                                         const uint8_t* __restrict__ idx, double* __restrict__ partial,
                                                         const double* __restrict__ sums, double count, int training,
                                                         float* __restrict__ dz, long rows, int H, int W, int C, int CQ,
                                                         int RP, long rows_per_block, unsigned* dz_amax) {
    __shared__ double red[APPLY ? 1 : 16 * 256];
    float dmx = 0.f;                // dz_amax (APPLY, nullable): magnitude slot of dz
    const int t = threadIdx.x, cq = t % CQ, pr = t / CQ;
    const bool active = pr < RP;
    const float slope = slope_dev ? slope_dev[0] : slope_val;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    float sc[4], sh[4], mu[4], is[4], k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
    {
        const float4 m4 = *reinterpret_cast<const float4*>(mean + cq * 4);
        const float4 i4 = *reinterpret_cast<const float4*>(invstd + cq * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + cq * 4);
        const float4 b4 = *reinterpret_cast<const float4*>(beta + cq * 4);
        mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
        is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = is[q] * gg[q];
            sh[q] = bb[q] - mu[q] * sc[q];
            if (APPLY && training) {
                k1[q] = (float)(sums[cq * 4 + q] / count);
                k2[q] = (float)(sums[C + cq * 4 + q] / count);
            }
        }
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (active) {
#pragma unroll 2
        for (long row = r0 + pr; row < r1; row += RP) {
            long base = row;
            float gp[4] = {0, 0, 0, 0};
            int pi[4] = {-1, -1, -1, -1};
            if (POOL) {
                const int W2 = W >> 1, H2 = H >> 1;
                const int j = (int)(row % W2);
                const int i = (int)((row / W2) % H2);
                const long img = row / ((long)W2 * H2);
                base = (img * H + 2 * i) * W + 2 * j;
                if (g_pool) {
                    // the apply pass is the last reader of z, of both gradient operands and of the arg-max bytes
                    const float4 g4 = APPLY ? ld_nt4(g_pool + row * C + cq * 4) : *reinterpret_cast<const float4*>(g_pool + row * C + cq * 4);
                    const uchar4 i4 = APPLY ? ld_nt_u8x4(idx + row * C + cq * 4) : *reinterpret_cast<const uchar4*>(idx + row * C + cq * 4);
                    gp[0] = g4.x; gp[1] = g4.y; gp[2] = g4.z; gp[3] = g4.w;
                    pi[0] = i4.x; pi[1] = i4.y; pi[2] = i4.z; pi[3] = i4.w;
                }
            }
            float pacc[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < (POOL ? 4 : 1); ++k) {
                const long pix = POOL ? base + (k >> 1) * W + (k & 1) : base;
                const float4 x4 = APPLY ? ld_nt4(z + pix * C + cq * 4) : *reinterpret_cast<const float4*>(z + pix * C + cq * 4);
                float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g_full) f4 = APPLY ? ld_nt4(g_full + pix * C + cq * 4) : *reinterpret_cast<const float4*>(g_full + pix * C + cq * 4);
                const float x[4] = {x4.x, x4.y, x4.z, x4.w}, gf[4] = {f4.x, f4.y, f4.z, f4.w};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float y = fmaf(x[q], sc[q], sh[q]);
                    float g = gf[q];
                    if (POOL && pi[q] == k) g += gp[q];
                    const float gm = g * act_grad(y, slope);
                    const float xh = (x[q] - mu[q]) * is[q];
                    if (!APPLY) {
                        pacc[q] += gm;
                        pacc[4 + q] = fmaf(gm, xh, pacc[4 + q]);
                        pacc[8 + q] += gf[q];
                        if (!(y > 0.f)) pacc[12 + q] = fmaf(g, y, pacc[12 + q]);   // d/d(slope) of PReLU
                    } else {
                        o[q] = training ? sc[q] * (gm - k1[q] - xh * k2[q]) : sc[q] * gm;
                    }
                }
                if (APPLY) {
                    *reinterpret_cast<float4*>(dz + pix * C + cq * 4) = make_float4(o[0], o[1], o[2], o[3]);
                    dmx = amax_acc(amax_acc(amax_acc(amax_acc(dmx, o[0]), o[1]), o[2]), o[3]);
                }
            }
            if (!APPLY) {
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += (double)pacc[q];
            }
        }
    }
    if (APPLY && dz_amax) amax_commit(dz_amax, dmx);
    if (!APPLY) {
        reduce_rows<16>(acc, red, t, CQ, RP, active);
        if (t < CQ) {
            double* out = partial + (long)blockIdx.x * 4 * C;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                out[cq * 4 + q] = acc[q];
                out[C + cq * 4 + q] = acc[4 + q];
                out[2 * C + cq * 4 + q] = acc[8 + q];
                out[3 * C + cq * 4 + q] = acc[12 + q];
            }
        }
    }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sums, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dbeta) dbeta[c] = (float)sums[c];
    if (dgamma) dgamma[c] = (float)sums[C + c];
}

// ---- first encoder convolution (NCHW input, CIN <= 6) ----------------------------------------
constexpr int FT_H = 8, FT_W = 32;

// WGRAD=false: forward (optionally also per-block BN statistics partials [grid][2][Cout] in `partial`)
template <int CIN, bool WGRAD>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ z_or_null, const float* __restrict__ dz,
                                                         float* __restrict__ partial, int N, int H, int W, int Cout,
                                                         int CQ, int PS, int tiles_x, int tiles_y, int ntiles) {
    constexpr int TW2 = FT_W + 2, TH2 = FT_H + 2, NT = 9 * CIN;
    __shared__ float tile[TH2 * TW2 * CIN];
    __shared__ float red[256 * 4];
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};   // forward: running BN sums of this thread
    const int t = threadIdx.x, cq = t % CQ, ps = t / CQ;
    const bool active = ps < PS;
    float wreg[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tap = j / CIN, ci = j % CIN;
            wreg[j][k] = WGRAD ? 0.f : (active ? w[((long)(cq * 4 + k) * CIN + ci) * 9 + tap] : 0.f);
        }
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * FT_H, x0 = tx * FT_W;
        __syncthreads();
        for (int e = t; e < TH2 * TW2 * CIN; e += 256) {
            const int ci = e / (TH2 * TW2), rem = e % (TH2 * TW2), yy = rem / TW2, xx = rem % TW2;
            const int gy = y0 + yy - 1, gx = x0 + xx - 1;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = x[(((long)n * CIN + ci) * H + gy) * W + gx];
            tile[(yy * TW2 + xx) * CIN + ci] = v;
        }
        __syncthreads();
        if (active) {
            for (int pix = ps; pix < FT_H * FT_W; pix += PS) {
                const int py = pix / FT_W, px = pix % FT_W;
                const int gy = y0 + py, gx = x0 + px;
                if (gy >= H || gx >= W) continue;
                const long o = (((long)n * H + gy) * W + gx) * Cout + cq * 4;
                if (!WGRAD) {
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int tap = j / CIN, ci = j % CIN;
                        const float xv = tile[((py + tap / 3) * TW2 + px + tap % 3) * CIN + ci];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k] = fmaf(xv, wreg[j][k], acc[k]);
                    }
                    *reinterpret_cast<float4*>(z_or_null + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        st_s[k] += acc[k];
                        st_q[k] = fmaf(acc[k], acc[k], st_q[k]);
                    }
                } else {
                    const float4 d4 = *reinterpret_cast<const float4*>(dz + o);
                    const float d[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int tap = j / CIN, ci = j % CIN;
                        const float xv = tile[((py + tap / 3) * TW2 + px + tap % 3) * CIN + ci];
#pragma unroll
                        for (int k = 0; k < 4; ++k) wreg[j][k] = fmaf(xv, d[k], wreg[j][k]);
                    }
                }
            }
        }
    }
    if (!WGRAD && partial) {
        // BN statistics of this block: fixed-order reduction over the pixel slots -> partial[block][2][Cout]
        float* out = partial + (long)blockIdx.x * 2 * Cout;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) red[t * 4 + k] = active ? (q == 0 ? st_s[k] : st_q[k]) : 0.f;
            __syncthreads();
            if (t < CQ) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float sacc = 0.f;
                    for (int r = 0; r < PS; ++r) sacc += red[(r * CQ + t) * 4 + k];
                    out[(long)q * Cout + t * 4 + k] = sacc;
                }
            }
        }
    }
    if (WGRAD) {
        // fixed-order reduction over the pixel slots, one (tap,ci) row at a time
        float* out = partial + (long)blockIdx.x * NT * Cout;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) red[t * 4 + k] = active ? wreg[j][k] : 0.f;
            __syncthreads();
            if (t < CQ) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float s = 0.f;
                    for (int r = 0; r < PS; ++r) s += red[(r * CQ + t) * 4 + k];
                    out[(long)j * Cout + t * 4 + k] = s;
                }
            }
        }
    }
}

// dw[(co*CIN + ci)*9 + tap] = sum_b partial[b][tap*CIN + ci][co]
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                 float* __restrict__ dw, int nb, int CIN, int Cout) {
    __shared__ double red[256];
    const int NT = 9 * CIN, tot = NT * Cout;
    const double r = sliced_column_sum<float>(partial, nb, tot, tot, red);
    const int e = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && e < tot) {
        const int j = e / Cout, co = e % Cout, tap = j / CIN, ci = j % CIN;
        dw[((long)co * CIN + ci) * 9 + tap] = (float)r;
    }
}

// ---- last convolution C -> 1 (+bias, + x0) ----------------------------------------------------
constexpr int LT_H = 8, LT_W = 32, LT_CH = 16, LT_STRIDE = LT_CH + 4;   // (32-channel chunks measured slower: 0.27 vs 0.21 ms)

__global__ __launch_bounds__(256) void conv_last_fwd_kernel(const float* __restrict__ s_in, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ x_nchw, int xc,
                                                            float* __restrict__ out, int N, int H, int W, int C,
                                                            int tiles_x, int tiles_y) {
    constexpr int TW2 = LT_W + 2, TH2 = LT_H + 2;
    __shared__ __attribute__((aligned(16))) float tile[TH2 * TW2 * LT_STRIDE];
    const int t = threadIdx.x;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * LT_H, x0 = tx * LT_W;
    const int py = t / LT_W, px = t % LT_W;
    float acc = 0.f;
    for (int c0 = 0; c0 < C; c0 += LT_CH) {
        const int nq = (C - c0 < LT_CH ? C - c0 : LT_CH) / 4;  // float4 per pixel in this chunk
        __syncthreads();
        for (int e = t; e < TH2 * TW2 * nq; e += 256) {
            const int q = e % nq, pp = e / nq, yy = pp / TW2, xx = pp % TW2;
            const int gy = y0 + yy - 1, gx = x0 + xx - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v = *reinterpret_cast<const float4*>(s_in + (((long)n * H + gy) * W + gx) * C + c0 + q * 4);
            *reinterpret_cast<float4*>(&tile[pp * LT_STRIDE + q * 4]) = v;
        }
        __syncthreads();
        for (int tap = 0; tap < 9; ++tap) {
            const float* tp = &tile[((py + tap / 3) * TW2 + px + tap % 3) * LT_STRIDE];
            for (int q = 0; q < nq; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(tp + q * 4);
                const int c = c0 + q * 4;
                acc = fmaf(v.x, w[(c + 0) * 9 + tap], acc);
                acc = fmaf(v.y, w[(c + 1) * 9 + tap], acc);
                acc = fmaf(v.z, w[(c + 2) * 9 + tap], acc);
                acc = fmaf(v.w, w[(c + 3) * 9 + tap], acc);
            }
        }
    }
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W) {
        float v = acc;
        if (bias) v += bias[0];
        if (x_nchw) v = x_nchw[(((long)n * xc) * H + gy) * W + gx] + v;
        out[((long)n * H + gy) * W + gx] = v;
    }
}

// ds[q][ci] = sum_tap dout[q - tap] * w[ci][tap]
__global__ __launch_bounds__(256) void conv_last_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                              float* __restrict__ ds, long P, int H, int W, int C, int CQ,
                                                              int RP, long rows_per_block) {
    const int t = threadIdx.x, cq = t % CQ, pr = t / CQ;
    if (pr >= RP) return;
    float wr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[tap][k] = w[(cq * 4 + k) * 9 + tap];
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > P) r1 = P;
    // 32-bit pixel arithmetic (P < 2^31, checked by the launcher); four pixels per iteration keep four independent
    // gather/FMA chains and four 16-byte stores in flight per thread
    const unsigned uW = (unsigned)W, uH = (unsigned)H;
    for (long pb = r0 + pr; pb < r1; pb += 4L * RP) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long p = pb + (long)u * RP;
            if (p >= r1) break;
            const unsigned pu = (unsigned)p;
            const int xx = (int)(pu % uW), yy = (int)((pu / uW) % uH);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const int sy = yy - dy, sx = xx - dx;
                float d = 0.f;
                if ((unsigned)sy < uH && (unsigned)sx < uW) d = dout[p - dy * W - dx];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = fmaf(d, wr[tap][k], acc[k]);
            }
            *reinterpret_cast<float4*>(ds + p * C + cq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

// dw[ci][tap] = sum_q s[q][ci] * dout[q - tap];  db = sum dout
__global__ __launch_bounds__(256) void conv_last_wgrad_kernel(const float* __restrict__ s_in,
                                                              const float* __restrict__ dout,
                                                              double* __restrict__ partial, long P, int H, int W, int C,
                                                              int CQ, int RP, long rows_per_block) {
    __shared__ double red[4 * 256];
    const int t = threadIdx.x, cq = t % CQ, pr = t / CQ;
    const bool active = pr < RP;
    float acc[9][4];
    float accb = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[tap][k] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > P) r1 = P;
    if (active) {
        // four pixels per iteration: four 16-byte loads of s in flight per thread (the kernel streams 4*P*C bytes once;
        // with one load per iteration it was latency-bound at ~2 TB/s); 32-bit pixel arithmetic
        const unsigned uW = (unsigned)W, uH = (unsigned)H;
        for (long pb = r0 + pr; pb < r1; pb += 4L * RP) {
            float4 s4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long p = pb + (long)u * RP;
                s4[u] = p < r1 ? *reinterpret_cast<const float4*>(s_in + p * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long p = pb + (long)u * RP;
                if (p >= r1) break;
                const unsigned pu = (unsigned)p;
                const int xx = (int)(pu % uW), yy = (int)((pu / uW) % uH);
                const float sv[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    const int sy = yy - dy, sx = xx - dx;
                    float d = 0.f;
                    if ((unsigned)sy < uH && (unsigned)sx < uW) d = dout[p - dy * W - dx];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[tap][k] = fmaf(sv[k], d, acc[tap][k]);
                    if (tap == 4 && cq == 0) accb += d;
                }
            }
        }
    }
    // partial layout per block: [9][C] then [1] (bias)
    double* out = partial + (long)blockIdx.x * (9 * C + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = acc[tap][k];
        reduce_rows<4>(v, red, t, CQ, RP, active);
        if (t < CQ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[tap * C + cq * 4 + k] = v[k];
        }
    }
    double vb[4] = {(double)accb, 0.0, 0.0, 0.0};
    reduce_rows<4>(vb, red, t, CQ, RP, active);
    if (t == 0) out[9 * C] = vb[0];
}

// dw[ci*9 + tap] = sum_b partial[b][tap*C + ci]; dbias = sum_b partial[b][9*C]
__global__ __launch_bounds__(256) void last_wgrad_reduce_kernel(const double* __restrict__ partial,
                                                                float* __restrict__ dw, float* __restrict__ dbias,
                                                                int nb, int C) {
    __shared__ double red[256];
    const int tot = 9 * C + 1;
    const double r = sliced_column_sum<double>(partial, nb, tot, tot, red);
    const int e = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x < 16 && e < tot) {
        if (e == 9 * C) {
            if (dbias) dbias[0] = (float)r;
        } else {
            const int tap = e / C, ci = e % C;
            dw[ci * 9 + tap] = (float)r;
        }
    }
}

// ---- masked de-normalised L1 ---------------------------------------------------------------------
__device__ __forceinline__ float denorm(float v, float sd, float mu) { return __fadd_rn(__fmul_rn(v, sd), mu); }

__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ yp, const float* __restrict__ y,
                                                         const uint8_t* __restrict__ mask, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, double* __restrict__ partial,
                                                         long total, long pps) {
    __shared__ double red[2 * 256];
    double s = 0.0, c = 0.0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        if (mask[e]) {
            const int n = (int)(e / pps);
            const float sd = stdv[n], mu = mean[n];
            s += (double)fabsf(denorm(yp[e], sd, mu) - denorm(y[e], sd, mu));
            c += 1.0;
        }
    }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x] += red[threadIdx.x + off];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2] = red[0];
        partial[blockIdx.x * 2 + 1] = red[256];
    }
}

__global__ __launch_bounds__(256) void l1_reduce_kernel(const double* __restrict__ partial, double* __restrict__ sums,
                                                        int nb) {
    __shared__ double red[2 * 256];
    double s = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        s += partial[b * 2];
        c += partial[b * 2 + 1];
    }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x] += red[threadIdx.x + off];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[0] = red[0];
        sums[1] = red[256];
    }
}

__global__ __launch_bounds__(256) void l1_finish_kernel(const float* __restrict__ yp, const float* __restrict__ y,
                                                        const uint8_t* __restrict__ mask, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, const double* __restrict__ sums,
                                                        double numel_total, const float* __restrict__ gout_dev,
                                                        float* __restrict__ loss, float* __restrict__ dyp, long total,
                                                        long pps) {
    const float cnt = (float)sums[1], numel = (float)numel_total;
    const float gout = gout_dev ? gout_dev[0] : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) {
        float l = (float)(sums[0] / numel_total);  // L1Loss(mean) over all elements
        l = l * numel;                             // * loss_mask.numel()
        loss[0] = l / cnt;                         // / loss_mask.sum()
    }
    if (!dyp) return;
    // autograd of the reference expression: ((gout / cnt) * numel) / numel * sign(p - t) * std_i, 0 where masked
    const float g = ((gout / cnt) * numel) / numel;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float o = 0.f;
        if (mask[e]) {
            const int n = (int)(e / pps);
            const float sd = stdv[n], mu = mean[n];
            const float d = denorm(yp[e], sd, mu) - denorm(y[e], sd, mu);
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            o = g * sg * sd;
        }
        dyp[e] = o;
    }
}

// ---- Adam -----------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_elements(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float w1, float b2, float w2, float eps, float wd,
                                              float step_size, float bc2_sqrt, float gscale) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        // the gradient and the two moments stream through once per step (non-temporal); the parameter is read again by the pack
        const float pe = p[e];
        float ge = ld_nt1(g + e) * gscale;
        ge = fmaf(wd, pe, ge);
        float me = ld_nt1(m + e), ve = ld_nt1(v + e);
        me = me + w1 * (ge - me);
        ve = fmaf(w2 * ge, ge, ve * b2);
        const float denom = sqrtf(ve) / bc2_sqrt + eps;
        p[e] = pe - step_size * (me / denom);
        st_nt1(m + e, me);
        st_nt1(v + e, ve);
    }
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float w1,
                                                   float b2, float w2, float eps, float wd, float step_size,
                                                   float bc2_sqrt, float gscale) {
    adam_elements(p, g, m, v, n, w1, b2, w2, eps, wd, step_size, bc2_sqrt, gscale);
}
// the same step with its eight scalars read from device memory (sc = {1-beta1, beta2, 1-beta2, eps, weight_decay, step_size,
// sqrt(bias_correction2), grad_scale}): the form a captured HIP graph replays -- the host refreshes sc before every replay
// (step count, learning-rate schedule), the launch itself never changes
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long n, const float* __restrict__ sc) {
    adam_elements(p, g, m, v, n, sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7]);
}

// ---- SGD (torch.optim.SGD, lib/utils.py:332-334: lr + coupled weight decay; momentum / dampening / nesterov as torch) ----
template <bool MOM>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long n, float lr, float wd, float momentum, float damp1, int nesterov,
                                                  int first, float gscale) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float pe = p[e];
        float ge = g[e] * gscale;
        ge = fmaf(wd, pe, ge);
        if (MOM) {
            // torch: buf = clone(g) on the first step, else buf = buf * momentum + (1 - dampening) * g
            const float be = first ? ge : fmaf(damp1, ge, buf[e] * momentum);
            buf[e] = be;
            ge = nesterov ? fmaf(momentum, be, ge) : be;
        }
        p[e] = fmaf(-lr, ge, pe);
    }
}

// ---- tiled inference: linear blend --------------------------------------------------------------
__device__ __forceinline__ double blend_axis(int c, int lo, int hi, int T, int overlap, double step) {
    // np.linspace(0, 1, overlap): ramp[i] = i * step, last element exactly 1
    double w = 1.0;
    if (lo > 0) {
        if (c < lo - overlap) return 0.0;
        if (c < lo) {
            const int i = c - (lo - overlap);
            w *= (i == overlap - 1 && overlap > 1) ? 1.0 : i * step;
        }
    }
    if (hi < T - 1 && c > hi) {
        const int i = overlap - 1 - (c - hi - 1);
        w *= (i < 0) ? 0.0 : ((i == overlap - 1 && overlap > 1) ? 1.0 : i * step);
    }
    return w;
}

__global__ __launch_bounds__(256) void blend_tile_kernel(const float* __restrict__ pred, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, const int* __restrict__ pos,
                                                         const int* __restrict__ reg, int tile, int T, int stride,
                                                         double* __restrict__ raster, int rows, int cols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= T * T) return;
    const int r = e / T, c = e - r * T;
    const int y0 = pos[tile * 2], x0 = pos[tile * 2 + 1];
    const int uly = reg[tile * 4], ulx = reg[tile * 4 + 1], lry = reg[tile * 4 + 2], lrx = reg[tile * 4 + 3];
    const int y = y0 + r, x = x0 + c;
    if ((unsigned)y >= (unsigned)rows || (unsigned)x >= (unsigned)cols) return;
    const int overlap = T - stride;
    const double step = overlap > 1 ? 1.0 / (double)(overlap - 1) : 0.0;
    const double w = blend_axis(r, uly, lry, T, overlap, step) * blend_axis(c, ulx, lrx, T, overlap, step);
    const float den = __fadd_rn(__fmul_rn(pred[(long)tile * T * T + e], stdv[tile]), mean[tile]);
    raster[(long)y * cols + x] += (double)den * w;
}

// All tiles of a batch in ONE launch with the accumulation order of the per-tile launches (tile 0, 1, 2, ... per raster
// pixel: the reference's order, lib/evaluation.py:497-511).  Thread (tile i, pixel e) OWNS its raster pixel iff no earlier
// tile of the batch covers it; the owner adds the contributions of tiles i, i+1, ... that cover the pixel, in order, and
// every other thread exits -- no atomics, no inter-thread ordering needed, any tile placement (overlapping areas, shifted
// border tiles) works.
__global__ __launch_bounds__(256) void blend_batch_kernel(const float* __restrict__ pred, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, const int* __restrict__ pos,
                                                          const int* __restrict__ reg, int n, int T, int stride,
                                                          double* __restrict__ raster, int rows, int cols) {
    const long e_all = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e_all >= (long)n * T * T) return;
    const int i = (int)(e_all / ((long)T * T)), e = (int)(e_all - (long)i * T * T);
    const int r = e / T, c = e - r * T;
    const int y = pos[i * 2] + r, x = pos[i * 2 + 1] + c;
    if ((unsigned)y >= (unsigned)rows || (unsigned)x >= (unsigned)cols) return;
    for (int j = 0; j < i; ++j)
        if ((unsigned)(y - pos[j * 2]) < (unsigned)T && (unsigned)(x - pos[j * 2 + 1]) < (unsigned)T) return;   // not the owner
    const int overlap = T - stride;
    const double step = overlap > 1 ? 1.0 / (double)(overlap - 1) : 0.0;
    double acc = raster[(long)y * cols + x];
    for (int j = i; j < n; ++j) {
        const int rj = y - pos[j * 2], cj = x - pos[j * 2 + 1];
        if ((unsigned)rj >= (unsigned)T || (unsigned)cj >= (unsigned)T) continue;
        const double w = blend_axis(rj, reg[j * 4], reg[j * 4 + 2], T, overlap, step) *
                         blend_axis(cj, reg[j * 4 + 1], reg[j * 4 + 3], T, overlap, step);
        const float den = __fadd_rn(__fmul_rn(pred[((long)j * T + rj) * T + cj], stdv[j]), mean[j]);
        acc += (double)den * w;
    }
    raster[(long)y * cols + x] = acc;
}

// ---- bilinear 2x upsampling (up_mode='bilinear', lib/UNet.py:20) ---------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False): src = max(0.5*(dst+0.5)-0.5, 0), i0 = floor(src),
// i1 = i0 + (i0 < size-1), l1 = src - i0, l0 = 1 - l1   (same arithmetic as ATen's upsample_bilinear2d).
__device__ __forceinline__ void up2_src(int dst, int size, int& i0, int& i1, float& l0, float& l1) {
    const float src = fmaxf(0.5f * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    i1 = i0 + (i0 < size - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// out[n,2h,2w,c] = skip + bias + up2(t[n,h,w,c]); one thread per (fine pixel, 4 channels)
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ t, const float* __restrict__ bias,
                                                             const float* __restrict__ skip, float* __restrict__ out,
                                                             long total, int h, int w, int CQ) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(e % CQ);
        const long pix = e / CQ;
        const int x = (int)(pix % (2 * w)), y = (int)((pix / (2 * w)) % (2 * h));
        const long img = pix / ((long)4 * w * h);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        up2_src(y, h, y0, y1, ly0, ly1);
        up2_src(x, w, x0, x1, lx0, lx1);
        const long C = (long)CQ * 4;
        const float* base = t + img * h * w * C + cq * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x0) * C);
        const float4 v01 = *reinterpret_cast<const float4*>(base + ((long)y0 * w + x1) * C);
        const float4 v10 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x0) * C);
        const float4 v11 = *reinterpret_cast<const float4*>(base + ((long)y1 * w + x1) * C);
        float4 r;
        r.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        r.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        r.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        r.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        if (bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + cq * 4);
            r.x += b4.x; r.y += b4.y; r.z += b4.z; r.w += b4.w;
        }
        if (skip) {
            const float4 s4 = *reinterpret_cast<const float4*>(skip + e * 4);
            r.x = s4.x + r.x; r.y = s4.y + r.y; r.z = s4.z + r.z; r.w = s4.w + r.w;
        }
        *reinterpret_cast<float4*>(out + e * 4) = r;
    }
}

// adjoint: dt[n,h,w,c] = up2^T(g[n,2h,2w,c]).  Gather form (no atomics, fixed summation order): coarse pixel i receives
// from the fine rows 2i-1 .. 2i+2 whichever interpolation weights name it (clamped borders fold two weights into one).
__device__ __forceinline__ void up2_adj_weights(int i, int size, int r0, float wgt[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + k;
        float wk = 0.f;
        if (r >= 0 && r < 2 * size) {
            int i0, i1;
            float l0, l1;
            up2_src(r, size, i0, i1, l0, l1);
            if (i0 == i) wk += l0;
            if (i1 == i) wk += l1;
        }
        wgt[k] = wk;
    }
}

__global__ __launch_bounds__(256) void upsample2x_adj_kernel(const float* __restrict__ g, float* __restrict__ dt,
                                                             long total, int h, int w, int CQ) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(e % CQ);
        const long pix = e / CQ;
        const int j = (int)(pix % w), i = (int)((pix / w) % h);
        const long img = pix / ((long)w * h);
        float wy[4], wx[4];
        up2_adj_weights(i, h, 2 * i - 1, wy);
        up2_adj_weights(j, w, 2 * j - 1, wx);
        const long C = (long)CQ * 4;
        const float* base = g + img * 4 * h * w * C + cq * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int r = 2 * i - 1 + a;
            if (wy[a] == 0.f) continue;
            float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int cc = 2 * j - 1 + b;
                if (wx[b] == 0.f) continue;
                const float4 v = *reinterpret_cast<const float4*>(base + ((long)r * 2 * w + cc) * C);
                row.x = fmaf(wx[b], v.x, row.x); row.y = fmaf(wx[b], v.y, row.y);
                row.z = fmaf(wx[b], v.z, row.z); row.w = fmaf(wx[b], v.w, row.w);
            }
            acc.x = fmaf(wy[a], row.x, acc.x); acc.y = fmaf(wy[a], row.y, acc.y);
            acc.z = fmaf(wy[a], row.z, acc.z); acc.w = fmaf(wy[a], row.w, acc.w);
        }
        *reinterpret_cast<float4*>(dt + e * 4) = acc;
    }
}

// ---- training-sample assembly -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_sums_kernel(const float* __restrict__ planes, long plane_stride,
                                                         const int* __restrict__ plane_idx, int P,
                                                         const int* __restrict__ pos, int T, int W, float nodata,
                                                         int use_nodata, double* __restrict__ sums) {
    __shared__ double red[2 * 256];
    const int i = blockIdx.x, y0 = pos[i * 2], x0 = pos[i * 2 + 1];
    double s = 0.0, c = 0.0;
    for (int p = 0; p < P; ++p) {
        const float* pl = planes + (long)plane_idx[i * P + p] * plane_stride;
        for (int e = threadIdx.x; e < T * T; e += 256) {
            const float v = pl[(long)(y0 + e / T) * W + x0 + e % T];
            if (!use_nodata || v != nodata) {
                s += v;
                c += 1.0;
            }
        }
    }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x] += red[threadIdx.x + off];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[i * 2] = red[0];
        sums[i * 2 + 1] = red[256];
    }
}

__global__ __launch_bounds__(256) void assemble_patches_kernel(
    const float* __restrict__ dsm_in, const float* __restrict__ dsm_gt, const float* __restrict__ ortho, long plane_stride,
    const int* __restrict__ pair_idx, int V, const int* __restrict__ pos, const int* __restrict__ aug,
    const float* __restrict__ dsm_mean, float dsm_std, const float* __restrict__ ortho_mean, float ortho_std, float nodata,
    int n, int T, int W, float* __restrict__ input, float* __restrict__ target, uint8_t* __restrict__ mask) {
    const int C = 1 + V;
    const long total = (long)n * (C + 1) * T * T;    // channel C = the target / mask plane
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % T), r = (int)((e / T) % T);
        const int ch = (int)((e / ((long)T * T)) % (C + 1));
        const int i = (int)(e / ((long)T * T * (C + 1)));
        if (ch == C && !dsm_gt) continue;
        // inverse of rot90(k) -> flipud -> fliplr
        const int a = aug ? aug[i] : 0, k = a & 3;
        const int c1 = (a & 8) ? T - 1 - c : c, r1 = (a & 4) ? T - 1 - r : r;
        int sr, sc;
        if (k == 0) { sr = r1; sc = c1; }
        else if (k == 1) { sr = c1; sc = T - 1 - r1; }
        else if (k == 2) { sr = T - 1 - r1; sc = T - 1 - c1; }
        else { sr = T - 1 - c1; sc = r1; }
        const long src = (long)(pos[i * 2] + sr) * W + pos[i * 2 + 1] + sc;
        if (ch == 0) {
            input[((long)i * C) * T * T + (long)r * T + c] = __fdiv_rn(__fsub_rn(dsm_in[src], dsm_mean[i]), dsm_std);
        } else if (ch < C) {
            const float v = ortho[(long)pair_idx[i * V + ch - 1] * plane_stride + src];
            input[((long)i * C + ch) * T * T + (long)r * T + c] = __fdiv_rn(__fsub_rn(v, ortho_mean[i]), ortho_std);
        } else {
            const float g = dsm_gt[src];
            target[(long)i * T * T + (long)r * T + c] = __fdiv_rn(__fsub_rn(g, dsm_mean[i]), dsm_std);
            mask[(long)i * T * T + (long)r * T + c] = (g != 0.f && g != nodata) ? 1 : 0;
        }
    }
}

// ---- layout ---------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
    const long total = (long)N * C * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const long pix = e / C;
        const long n = pix / HW, hw = pix % HW;
        dst[e] = src[(n * C + c) * HW + hw];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
    const long total = (long)N * C * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long hw = e % HW;
        const int c = (int)((e / HW) % C);
        const long n = e / ((long)HW * C);
        dst[e] = src[(n * HW + hw) * C + c];
    }
}

}  // namespace rd

using namespace rd;

extern "C" {

int rd_upsample2x_add_fwd(const float* t, const float* bias, const float* skip, float* out, int n, int h, int w, int c,
                          rd_stream_t s) {
    RD_REQUIRE(t && out && n > 0 && h > 0 && w > 0, "rd_upsample2x_add_fwd: bad arguments");
    RD_REQUIRE(c > 0 && c % 4 == 0, "rd_upsample2x_add_fwd: C must be a multiple of 4 (got %d)", c);
    const long total = (long)n * 4 * h * w * (c / 4);
    ProfScope ps((hipStream_t)s, "upsample2x_add", 0, 4.0 * ((double)n * h * w * c * (1 + 4 + (skip ? 4 : 0))));
    RD_LAUNCH(upsample2x_add_kernel, dim3(grid_cap((total + 255) / 256, 16384)), dim3(256), 0, (hipStream_t)s, t, bias, skip, out,
                       total, h, w, c / 4);
    RD_LAUNCH_CHECK("upsample2x_add");
    return RD_OK;
}

int rd_upsample2x_bwd(const float* g, float* dt, int n, int h, int w, int c, rd_stream_t s) {
    RD_REQUIRE(g && dt && n > 0 && h > 0 && w > 0, "rd_upsample2x_bwd: bad arguments");
    RD_REQUIRE(c > 0 && c % 4 == 0, "rd_upsample2x_bwd: C must be a multiple of 4 (got %d)", c);
    const long total = (long)n * h * w * (c / 4);
    ProfScope ps((hipStream_t)s, "upsample2x_bwd", 0, 4.0 * ((double)n * h * w * c * 5));
    RD_LAUNCH(upsample2x_adj_kernel, dim3(grid_cap((total + 255) / 256, 16384)), dim3(256), 0, (hipStream_t)s, g, dt, total, h, w,
                       c / 4);
    RD_LAUNCH_CHECK("upsample2x_bwd");
    return RD_OK;
}

size_t rd_channel_sum_ws_bytes(long long pixels, int c) {
    RowPlan pl;
    if (!plan_rows(pixels, c, &pl)) return 0;
    return (size_t)pl.nb * c * sizeof(double) + (size_t)c * sizeof(double);
}

int rd_channel_sum(const float* g, float* out, long long pixels, int c, void* ws, size_t ws_bytes, rd_stream_t s) {
    // result returned as float; internally doubles.  out may alias nothing in ws.
    RowPlan pl;
    RD_REQUIRE(plan_rows(pixels, c, &pl), "rd_channel_sum: C must be a multiple of 4 and <= 1024 (got %d)", c);
    const size_t need = (size_t)pl.nb * c * sizeof(double) + (size_t)c * sizeof(double);
    if (!ws || ws_bytes < need) {
        set_error("rd_channel_sum: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "channel_sum", 0, 4.0 * pixels * c);
    double* partial = (double*)ws;
    double* sums = partial + (size_t)pl.nb * c;
    RD_LAUNCH((channel_stats_kernel<false>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, g, partial,
                       (long)pixels, c, pl.CQ, pl.RP, pl.rows_per_block);
    RD_LAUNCH(partial_reduce_kernel, dim3(cdiv(c, 16)), dim3(256), 0, (hipStream_t)s, partial, sums, pl.nb, c, 0,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr);
    RD_LAUNCH(bn_param_grad_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)s, sums, (float*)nullptr,
                       out, c);
    RD_LAUNCH_CHECK("channel_sum");
    return RD_OK;
}

size_t rd_bn_stats_ws_bytes(long long pixels, int c) {
    RowPlan pl;
    if (!plan_rows(pixels, c, &pl)) return 0;
    return (size_t)pl.nb * 2 * c * sizeof(double);
}

int rd_bn_stats_partial(const float* z, double* sums, long long pixels, int c, void* ws, size_t ws_bytes,
                        rd_stream_t s) {
    RowPlan pl;
    RD_REQUIRE(z && sums, "rd_bn_stats_partial: null pointer");
    RD_REQUIRE(plan_rows(pixels, c, &pl), "rd_bn_stats_partial: C must be a multiple of 4 and <= 1024 (got %d)", c);
    const size_t need = (size_t)pl.nb * 2 * c * sizeof(double);
    if (!ws || ws_bytes < need) {
        set_error("rd_bn_stats_partial: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "bn_stats", 0, 4.0 * pixels * c);
    RD_LAUNCH((channel_stats_kernel<true>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, z, (double*)ws,
                       (long)pixels, c, pl.CQ, pl.RP, pl.rows_per_block);
    RD_LAUNCH(partial_reduce_kernel, dim3(cdiv(2 * c, 16)), dim3(256), 0, (hipStream_t)s, (const double*)ws,
                       sums, pl.nb, 2 * c, 0, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    RD_LAUNCH_CHECK("bn_stats");
    return RD_OK;
}

int rd_bn_stats_finalize(const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, int64_t* nbt, int c, rd_stream_t s) {
    RD_REQUIRE(sums && mean && invstd && c > 0 && count > 0, "rd_bn_stats_finalize: bad arguments");
    RD_LAUNCH(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)s, sums, count, eps, momentum,
                       mean, invstd, running_mean, running_var, nbt, c);
    RD_LAUNCH_CHECK("bn_finalize");
    return RD_OK;
}

int rd_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* invstd, int c,
                     rd_stream_t s) {
    RD_REQUIRE(running_mean && running_var && mean && invstd && c > 0, "rd_bn_eval_stats: bad arguments");
    RD_LAUNCH(bn_eval_stats_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)s, running_mean,
                       running_var, eps, mean, invstd, c);
    RD_LAUNCH_CHECK("bn_eval_stats");
    return RD_OK;
}

int rd_bn_act_pool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, const float* slope_dev, float* a, float* pooled, uint8_t* idx, float* zpool, int n, int h,
                       int w, int c, rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && (a || pooled), "rd_bn_act_pool_fwd: null pointer");
    RD_REQUIRE(c % 4 == 0 && c > 0, "rd_bn_act_pool_fwd: C must be a multiple of 4 (got %d)", c);
    const QuantArgs qa = quant_take();      // out: magnitude slot of a (un-pooled form), out2: of pooled
    const int CQ = c / 4;
    const long pixels = (long)n * h * w;
    if (pooled) {
        RD_REQUIRE(idx && h % 2 == 0 && w % 2 == 0, "rd_bn_act_pool_fwd: pooling needs idx and even H, W");
        const long rows = pixels / 4;
        ProfScope ps((hipStream_t)s, "bn_act_pool_fwd", 0,
                     4.0 * pixels * c * (a ? 2.25 : 1.25) + 0.25 * pixels * c + (zpool ? 1.0 * pixels * c : 0.0));
        RD_LAUNCH((bn_act_pool_fwd_kernel<true>), dim3(grid_cap((rows * CQ + 255) / 256, 8192)), dim3(256), 0,
                           (hipStream_t)s, z, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, idx, zpool, rows, h, w, c,
                           CQ, (unsigned*)nullptr, qa.out2);
    } else {
        ProfScope ps((hipStream_t)s, "bn_act_fwd", 0, 8.0 * pixels * c);
        RD_LAUNCH((bn_act_pool_fwd_kernel<false>), dim3(grid_cap((pixels * CQ + 255) / 256, 8192)), dim3(256),
                           0, (hipStream_t)s, z, mean, invstd, gamma, beta, slope, slope_dev, a, (float*)nullptr,
                           (uint8_t*)nullptr, (float*)nullptr, pixels, h, w, c, CQ, qa.out, (unsigned*)nullptr);
    }
    RD_LAUNCH_CHECK("bn_act_pool_fwd");
    return RD_OK;
}

size_t rd_bn_act_bwd_ws_bytes(int n, int h, int w, int c) {
    RowPlan pl;
    if (!plan_rows((long)n * h * w, c, &pl)) return 0;
    return (size_t)pl.nb * 4 * c * sizeof(double);  // upper bound (pooled variant uses fewer rows)
}

int rd_bn_act_bwd_reduce(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         float slope, const float* slope_dev, const float* g_full, const float* g_pool,
                         const uint8_t* idx, double* sums, float* dgamma, float* dbeta, float* dextra, int n, int h, int w, int c,
                         void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && sums, "rd_bn_act_bwd_reduce: null pointer");
    RD_REQUIRE(g_full || g_pool, "rd_bn_act_bwd_reduce: no gradient source");
    RD_REQUIRE(!g_pool || idx, "rd_bn_act_bwd_reduce: g_pool needs idx");
    const bool pool = g_pool != nullptr;
    const long pixels = (long)n * h * w;
    const long rows = pool ? pixels / 4 : pixels;
    RowPlan pl;
    RD_REQUIRE(plan_rows(rows, c, &pl), "rd_bn_act_bwd_reduce: C must be a multiple of 4 and <= 1024 (got %d)", c);
    const size_t need = (size_t)pl.nb * 4 * c * sizeof(double);
    if (!ws || ws_bytes < need) {
        set_error("rd_bn_act_bwd_reduce: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "bn_act_bwd_reduce", 0, 4.0 * pixels * c * (1 + (g_full ? 1 : 0) + (pool ? 0.3 : 0)));
    if (pool)
        RD_LAUNCH((bn_act_bwd_kernel<true, false>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, z, mean, invstd,
                           gamma, beta, slope, slope_dev, g_full, g_pool, idx, (double*)ws, (const double*)nullptr, 1.0, 1,
                           (float*)nullptr, rows, h, w, c, pl.CQ, pl.RP, pl.rows_per_block, (unsigned*)nullptr);
    else
        RD_LAUNCH((bn_act_bwd_kernel<false, false>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, z, mean,
                           invstd, gamma, beta, slope, slope_dev, g_full, g_pool, idx, (double*)ws, (const double*)nullptr, 1.0, 1,
                           (float*)nullptr, rows, h, w, c, pl.CQ, pl.RP, pl.rows_per_block, (unsigned*)nullptr);
    RD_LAUNCH(partial_reduce_kernel, dim3(cdiv(4 * c, 16)), dim3(256), 0, (hipStream_t)s, (const double*)ws,
                       sums, pl.nb, 4 * c, c, dbeta, dgamma, dextra);
    RD_LAUNCH_CHECK("bn_act_bwd_reduce");
    return RD_OK;
}

int rd_bn_bwd_stats_finalize(const float* part_a, int rows_a, const float* part_b, int rows_b, int c, double* sums,
                             float* dgamma, float* dbeta, float* dextra, rd_stream_t s) {
    RD_REQUIRE(part_a && rows_a > 0 && sums && c > 0 && c % 4 == 0 && (!part_b || rows_b > 0), "rd_bn_bwd_stats_finalize: bad arguments");
    ProfScope ps((hipStream_t)s, "bn_act_bwd_reduce", 0, 16.0 * c * ((double)rows_a + (part_b ? rows_b : 0)));
    RD_LAUNCH(bn_bwd_stats_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)s, part_a, rows_a, part_b,
                       part_b ? rows_b : 0, c, sums, dgamma, dbeta, dextra);
    RD_LAUNCH_CHECK("bn_bwd_stats_finalize");
    return RD_OK;
}

int rd_bn_act_bwd_apply(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        float slope, const float* slope_dev, const float* g_full, const float* g_pool,
                        const uint8_t* idx, const double* sums,
                        double count, int training, float* dz, float* dgamma, float* dbeta, int n, int h, int w, int c,
                        rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && sums && dz, "rd_bn_act_bwd_apply: null pointer");
    RD_REQUIRE(g_full || g_pool, "rd_bn_act_bwd_apply: no gradient source");
    RD_REQUIRE(!g_pool || idx, "rd_bn_act_bwd_apply: g_pool needs idx");
    RD_REQUIRE(count > 0, "rd_bn_act_bwd_apply: count must be positive");
    const QuantArgs qa = quant_take();      // out: magnitude slot of dz
    const bool pool = g_pool != nullptr;
    const long pixels = (long)n * h * w;
    const long rows = pool ? pixels / 4 : pixels;
    RowPlan pl;
    RD_REQUIRE(plan_rows(rows, c, &pl), "rd_bn_act_bwd_apply: C must be a multiple of 4 and <= 1024 (got %d)", c);
    {
        ProfScope ps((hipStream_t)s, "bn_act_bwd_apply", 0,
                     4.0 * pixels * c * (2 + (g_full ? 1 : 0) + (pool ? 0.3 : 0)));
        if (pool)
            RD_LAUNCH((bn_act_bwd_kernel<true, true>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, z, mean,
                               invstd, gamma, beta, slope, slope_dev, g_full, g_pool, idx, (double*)nullptr, sums, count,
                               training, dz, rows, h, w, c, pl.CQ, pl.RP, pl.rows_per_block, qa.out);
        else
            RD_LAUNCH((bn_act_bwd_kernel<false, true>), dim3(pl.nb), dim3(256), 0, (hipStream_t)s, z, mean,
                               invstd, gamma, beta, slope, slope_dev, g_full, g_pool, idx, (double*)nullptr, sums, count, training,
                               dz, rows, h, w, c, pl.CQ, pl.RP, pl.rows_per_block, qa.out);
    }
    if (dgamma || dbeta)
        RD_LAUNCH(bn_param_grad_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)s, sums, dgamma, dbeta,
                           c);
    RD_LAUNCH_CHECK("bn_act_bwd_apply");
    return RD_OK;
}

}  // extern "C"

// ---- first conv ------------------------------------------------------------------------------------
static int first_grid(int n, int h, int w, int* tiles_x, int* tiles_y, int* ntiles) {
    *tiles_x = cdiv(w, FT_W);
    *tiles_y = cdiv(h, FT_H);
    *ntiles = n * (*tiles_x) * (*tiles_y);
    return *ntiles < 512 ? *ntiles : 512;
}

template <bool WGRAD>
static int launch_first(const float* x, const float* wt, float* z, const float* dz, float* partial, int n, int h, int w,
                        int cin, int cout, int grid, int tiles_x, int tiles_y, int ntiles, hipStream_t s) {
    const int CQ = cout / 4, PS = 256 / CQ;
#define RD_FIRST_CASE(CI)                                                                                              \
    case CI:                                                                                                           \
        RD_LAUNCH((conv_first_kernel<CI, WGRAD>), dim3(grid), dim3(256), 0, s, x, wt, z, dz, partial, n, h, w, \
                           cout, CQ, PS, tiles_x, tiles_y, ntiles);                                                    \
        break;
    switch (cin) {
        RD_FIRST_CASE(1)
        RD_FIRST_CASE(2)
        RD_FIRST_CASE(3)
        RD_FIRST_CASE(4)
        RD_FIRST_CASE(5)
        RD_FIRST_CASE(6)
        default:
            set_error("first conv supports 1..6 input channels (got %d)", cin);
            return RD_ERR_ARG;
    }
#undef RD_FIRST_CASE
    return RD_OK;
}

extern "C" {

int rd_conv3x3_first_fwd(const float* x, const float* wt, float* z, int n, int h, int w, int cin, int cout,
                         rd_stream_t s) {
    return rd_conv3x3_first_fwd_stats(x, wt, z, nullptr, n, h, w, cin, cout, nullptr, 0, s);
}

int rd_conv3x3_first_fwd_act_available(int n, int h, int w, int cin, int cout) {
    return conv_first_seg_tiles(n, h, w, cin, cout) > 0 && h % 2 == 0 && w % 2 == 0;
}

int rd_conv3x3_first_fwd_act(const float* x, const float* wt, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float slope, const float* slope_dev, float* a, float* pooled, int n, int h, int w,
                             int cin, int cout, rd_stream_t s) {
    RD_REQUIRE(x && wt && mean && invstd && gamma && beta && a, "rd_conv3x3_first_fwd_act: null pointer");
    RD_REQUIRE(rd_conv3x3_first_fwd_act_available(n, h, w, cin, cout),
               "rd_conv3x3_first_fwd_act: shape not covered (1-4 input channels, 32 / 64 / 128 output channels, even H and W)");
    ProfScope ps((hipStream_t)s, "conv_first_fwd", 2.0 * n * h * w * cout * 9.0 * cin,
                 4.0 * n * h * w * (double)(cin + cout * (pooled ? 1.25 : 1.0)));
    const QuantArgs qa = quant_take_img();     // out2: slot of `pooled` -- or, with a stride, one slot per image
    return conv_first_fwd_act_launch(x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, cin, cout, (hipStream_t)s,
                                     qa.out2, qa.img_stride);
}

size_t rd_conv3x3_first_fwd_stats_ws_bytes(int n, int h, int w, int cin, int cout) {
    if (const int nt2 = conv_first_seg_tiles(n, h, w, cin, cout)) return (size_t)nt2 * 2 * cout * sizeof(float);
    int tx, ty, nt;
    const int grid = first_grid(n, h, w, &tx, &ty, &nt);
    return (size_t)grid * 2 * cout * sizeof(float);
}

int rd_conv3x3_first_fwd_stats(const float* x, const float* wt, float* z, double* sums, int n, int h, int w, int cin,
                               int cout, void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(x && wt && z, "rd_conv3x3_first_fwd: null pointer");
    RD_REQUIRE(cout % 4 == 0 && cout / 4 <= 256 && cout > 0, "rd_conv3x3_first_fwd: Cout must be a multiple of 4, <= 1024");
    int tx, ty, nt;
    int grid = first_grid(n, h, w, &tx, &ty, &nt);
    const int seg_tiles = conv_first_seg_tiles(n, h, w, cin, cout);
    if (seg_tiles) grid = seg_tiles;
    if (sums && (!ws || ws_bytes < (size_t)grid * 2 * cout * sizeof(float))) {
        set_error("rd_conv3x3_first_fwd_stats: workspace too small");
        return RD_ERR_WS;
    }
    if (seg_tiles) {
        ProfScope ps((hipStream_t)s, "conv_first_fwd", 2.0 * n * h * w * cout * 9.0 * cin, 4.0 * n * h * w * (double)(cin + cout));
        if (int e = conv_first_seg_launch(false, x, wt, z, nullptr, sums ? (float*)ws : nullptr, n, h, w, cin, cout, (hipStream_t)s))
            return e;
    } else {
        ProfScope ps((hipStream_t)s, "conv_first_fwd", 2.0 * n * h * w * cout * 9.0 * cin,
                     4.0 * n * h * w * (double)(cin + cout));
        if (int e = launch_first<false>(x, wt, z, nullptr, sums ? (float*)ws : nullptr, n, h, w, cin, cout, grid, tx, ty,
                                        nt, (hipStream_t)s))
            return e;
        RD_LAUNCH_CHECK("conv_first_fwd");
    }
    if (sums) return reduce_partials_f32((const float*)ws, sums, grid, 2 * cout, (hipStream_t)s);
    return RD_OK;
}

int rd_conv3x3_first_fwd_bn(const float* x, const float* wt, float* z, double count, float eps, float momentum, float* mean,
                            float* invstd, float* running_mean, float* running_var, int64_t* nbt, int n, int h, int w, int cin,
                            int cout, void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(x && wt && z && mean && invstd && count > 0, "rd_conv3x3_first_fwd_bn: bad arguments");
    RD_REQUIRE(cout % 4 == 0 && cout / 4 <= 256 && cout > 0, "rd_conv3x3_first_fwd_bn: Cout must be a multiple of 4, <= 1024");
    int tx, ty, nt;
    int grid = first_grid(n, h, w, &tx, &ty, &nt);
    const int seg_tiles = conv_first_seg_tiles(n, h, w, cin, cout);
    if (seg_tiles) grid = seg_tiles;
    if (!ws || ws_bytes < (size_t)grid * 2 * cout * sizeof(float)) {
        set_error("rd_conv3x3_first_fwd_bn: workspace too small");
        return RD_ERR_WS;
    }
    if (seg_tiles) {
        ProfScope ps((hipStream_t)s, "conv_first_fwd", 2.0 * n * h * w * cout * 9.0 * cin, 4.0 * n * h * w * (double)(cin + cout));
        if (int e = conv_first_seg_launch(false, x, wt, z, nullptr, (float*)ws, n, h, w, cin, cout, (hipStream_t)s)) return e;
    } else {
        ProfScope ps((hipStream_t)s, "conv_first_fwd", 2.0 * n * h * w * cout * 9.0 * cin, 4.0 * n * h * w * (double)(cin + cout));
        if (int e = launch_first<false>(x, wt, z, nullptr, (float*)ws, n, h, w, cin, cout, grid, tx, ty, nt, (hipStream_t)s))
            return e;
        RD_LAUNCH_CHECK("conv_first_fwd");
    }
    return bn_reduce_finalize((const float*)ws, grid, cout, count, eps, momentum, mean, invstd, running_mean, running_var, nbt,
                              (hipStream_t)s);
}

size_t rd_conv3x3_first_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout) {
    if (const int nb = conv_first_wgrad_seg_blocks(n, h, w, cin, cout)) return (size_t)nb * 9 * cin * cout * sizeof(float);
    int tx, ty, nt;
    const int grid = first_grid(n, h, w, &tx, &ty, &nt);
    return (size_t)grid * 9 * cin * cout * sizeof(float);
}

int rd_conv3x3_first_bwd_weight_bn_available(int n, int h, int w, int cin, int cout) {
    return conv_first_wgrad_seg_blocks(n, h, w, cin, cout) != 0 && cin <= 3 && h % 2 == 0 && w % 2 == 0;
}

int rd_conv3x3_first_bwd_weight_bn(const float* x, const float* z, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float slope, const float* slope_dev, const float* g_full, const float* g_pool,
                                   const uint8_t* idx, const double* sums, double count, int training, const float* dout,
                                   const float* w_last, float* dw, int n, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                                   rd_stream_t s) {
    RD_REQUIRE(x && z && mean && invstd && gamma && beta && sums && dw, "rd_conv3x3_first_bwd_weight_bn: null pointer");
    RD_REQUIRE(g_full || g_pool || dout, "rd_conv3x3_first_bwd_weight_bn: no gradient source");
    RD_REQUIRE(!dout || (w_last && !g_full), "rd_conv3x3_first_bwd_weight_bn: dout needs w_last and excludes g_full");
    RD_REQUIRE(!g_pool || idx, "rd_conv3x3_first_bwd_weight_bn: g_pool needs idx");
    RD_REQUIRE(count > 0, "rd_conv3x3_first_bwd_weight_bn: count must be positive");
    RD_REQUIRE(rd_conv3x3_first_bwd_weight_bn_available(n, h, w, cin, cout),
               "rd_conv3x3_first_bwd_weight_bn: shape not handled (Cin <= 3, Cout in {32, 64, 128}, H and W even; got %d -> %d, %dx%d)",
               cin, cout, h, w);
    const int grid = conv_first_wgrad_seg_blocks(n, h, w, cin, cout);
    const size_t need = (size_t)grid * 9 * cin * cout * sizeof(float);
    if (!ws || ws_bytes < need) {
        set_error("rd_conv3x3_first_bwd_weight_bn: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "conv_first_wgrad", 2.0 * n * h * w * cout * 9.0 * cin,
                 4.0 * n * h * w * ((double)cin + cout * (1.0 + (g_full ? 1.0 : 0.0) + (g_pool ? 0.3 : 0.0))));
    const FirstBnBwd bn = {z, mean, invstd, gamma, beta, slope_dev, slope, g_full, g_pool, idx, sums, count, training, dout, w_last};
    if (int e = conv_first_seg_launch(true, x, nullptr, nullptr, nullptr, (float*)ws, n, h, w, cin, cout, (hipStream_t)s, &bn)) return e;
    RD_LAUNCH(first_wgrad_reduce_kernel, dim3(cdiv(9 * cin * cout, 16)), dim3(256), 0, (hipStream_t)s, (const float*)ws, dw,
                       grid, cin, cout);
    RD_LAUNCH_CHECK("conv_first_wgrad");
    return RD_OK;
}

int rd_conv3x3_first_bwd_weight(const float* x, const float* dz, float* dw, int n, int h, int w, int cin, int cout,
                                void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(x && dz && dw, "rd_conv3x3_first_bwd_weight: null pointer");
    RD_REQUIRE(cout % 4 == 0 && cout / 4 <= 256 && cout > 0,
               "rd_conv3x3_first_bwd_weight: Cout must be a multiple of 4, <= 1024");
    int tx, ty, nt;
    int grid = first_grid(n, h, w, &tx, &ty, &nt);
    const int seg_blocks = conv_first_wgrad_seg_blocks(n, h, w, cin, cout);
    if (seg_blocks) grid = seg_blocks;
    const size_t need = (size_t)grid * 9 * cin * cout * sizeof(float);
    if (!ws || ws_bytes < need) {
        set_error("rd_conv3x3_first_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "conv_first_wgrad", 2.0 * n * h * w * cout * 9.0 * cin,
                 4.0 * n * h * w * (double)(cin + cout));
    if (seg_blocks) {
        if (int e = conv_first_seg_launch(true, x, nullptr, nullptr, dz, (float*)ws, n, h, w, cin, cout, (hipStream_t)s)) return e;
        RD_LAUNCH(first_wgrad_reduce_kernel, dim3(cdiv(9 * cin * cout, 16)), dim3(256), 0, (hipStream_t)s,
                           (const float*)ws, dw, grid, cin, cout);
        RD_LAUNCH_CHECK("conv_first_wgrad");
        return RD_OK;
    }
    if (int e = launch_first<true>(x, nullptr, nullptr, dz, (float*)ws, n, h, w, cin, cout, grid, tx, ty, nt,
                                   (hipStream_t)s))
        return e;
    RD_LAUNCH(first_wgrad_reduce_kernel, dim3(cdiv(9 * cin * cout, 16)), dim3(256), 0, (hipStream_t)s,
                       (const float*)ws, dw, grid, cin, cout);
    RD_LAUNCH_CHECK("conv_first_wgrad");
    return RD_OK;
}

// ---- last conv -------------------------------------------------------------------------------------
static int last_blocks() { return tune(TUNE_LAST_BLOCKS); }   // first-stage blocks of the last-conv streaming kernels

int rd_conv3x3_last_fwd(const float* s_in, const float* wt, const float* bias, const float* x_nchw, int x_channels,
                        float* out, int n, int h, int w, int c, rd_stream_t s) {
    RD_REQUIRE(s_in && wt && out, "rd_conv3x3_last_fwd: null pointer");
    RD_REQUIRE(c % 4 == 0 && c > 0, "rd_conv3x3_last_fwd: C must be a multiple of 4 (got %d)", c);
    const int tx = cdiv(w, LT_W), ty = cdiv(h, LT_H);
    ProfScope ps((hipStream_t)s, "conv_last_fwd", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(c + 2));
    {
        int launched = 0;
        if (int e = conv_last_fwd_launch(s_in, wt, bias, x_nchw, x_channels, out, n, h, w, c, (hipStream_t)s, &launched)) return e;
        if (launched) return RD_OK;
    }
    RD_LAUNCH(conv_last_fwd_kernel, dim3(n * tx * ty), dim3(256), 0, (hipStream_t)s, s_in, wt, bias, x_nchw,
                       x_channels, out, n, h, w, c, tx, ty);
    RD_LAUNCH_CHECK("conv_last_fwd");
    return RD_OK;
}

int rd_conv3x3_last_bwd_data(const float* dout, const float* wt, float* ds, int n, int h, int w, int c, rd_stream_t s) {
    RD_REQUIRE(dout && wt && ds, "rd_conv3x3_last_bwd_data: null pointer");
    RowPlan pl;
    RD_REQUIRE(plan_rows((long)n * h * w, c, &pl, last_blocks()), "rd_conv3x3_last_bwd_data: C must be a multiple of 4, <= 1024");
    RD_REQUIRE((long)n * h * w < (1L << 31), "rd_conv3x3_last_bwd_data: pixel count must be < 2^31");
    ProfScope ps((hipStream_t)s, "conv_last_dgrad", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(c + 1));
    {
        int launched = 0;
        if (int e = conv_last_dgrad_launch(dout, wt, ds, n, h, w, c, (hipStream_t)s, &launched)) return e;
        if (launched) return RD_OK;
    }
    RD_LAUNCH(conv_last_dgrad_kernel, dim3(pl.nb), dim3(256), 0, (hipStream_t)s, dout, wt, ds, (long)n * h * w,
                       h, w, c, pl.CQ, pl.RP, pl.rows_per_block);
    RD_LAUNCH_CHECK("conv_last_dgrad");
    return RD_OK;
}

int rd_conv3x3_last_bwd_data_bnstats(const float* dout, const float* wt, float* ds, int n, int h, int w, int c,
                                     const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, float slope, const float* slope_dev, float* part, size_t part_floats,
                                     int* rows_out, rd_stream_t s) {
    RD_REQUIRE(dout && wt && bn_z && mean && invstd && gamma && beta && part && rows_out,
               "rd_conv3x3_last_bwd_data_bnstats: null pointer");
    RD_REQUIRE(part_floats >= rd_bn_bwd_part_floats((long long)n * h * w, c),
               "rd_conv3x3_last_bwd_data_bnstats: statistics buffer too small");
    {
        ProfScope ps((hipStream_t)s, "conv_last_dgrad", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(2 * c + 1));
        if (int e = conv_last_dgrad_bn_launch(dout, wt, ds, n, h, w, c, bn_z, mean, invstd, gamma, beta, slope, slope_dev, part,
                                              (hipStream_t)s, rows_out))
            return e;
        if (*rows_out) return RD_OK;
    }
    // channel counts without a tile kernel: plain data gradient, the caller runs the stand-alone reduction
    RD_REQUIRE(ds, "rd_conv3x3_last_bwd_data_bnstats: ds = NULL (statistics only) needs C in {16, 32, 64} (got %d)", c);
    return rd_conv3x3_last_bwd_data(dout, wt, ds, n, h, w, c, s);
}

int rd_tail_available(int cin, int c0) { return tail_shape_ok(cin) && c0 > 0 && c0 <= 1024 ? 1 : 0; }

int rd_tail_compose(const float* wt_iohw, const float* bias_t, const float* wl, float* M, float* V, float* VT, float* B9, int cin,
                    int c0, rd_stream_t s) {
    RD_REQUIRE(wt_iohw && wl && M && V && cin > 0 && c0 > 0, "rd_tail_compose: bad arguments");
    return tail_compose_launch(wt_iohw, bias_t, wl, M, V, VT, B9, cin, c0, (hipStream_t)s);
}

int rd_conv3x3_last_fwd_tail(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float slope, const float* slope_dev, const float* t16, const float* b9, const float* w_last,
                             const float* bias, const float* x_nchw, int x_channels, float* out, int n, int h, int w, int c,
                             rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && t16 && b9 && w_last && out, "rd_conv3x3_last_fwd_tail: null pointer");
    RD_REQUIRE((c == 16 || c == 32 || c == 64) && h % 2 == 0 && w % 2 == 0,
               "rd_conv3x3_last_fwd_tail: C in {16, 32, 64}, H and W even (got %d, %dx%d)", c, h, w);
    const TailSkip sk = {z, mean, invstd, gamma, beta, slope_dev, slope};
    ProfScope ps((hipStream_t)s, "conv_last_fwd", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(c + 2 + 4));
    return conv_last_fwd_tail_launch(sk, t16, b9, w_last, bias, x_nchw, x_channels, out, n, h, w, c, (hipStream_t)s);
}

int rd_conv3x3_last_bwd_tail_blocks(int n, int h, int w) { return conv_last_tail_blocks(n, h, w); }

int rd_conv3x3_last_bwd_tail_fused(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   float slope, const float* slope_dev, const float* dout, const float* w_last, double* wpartial,
                                   float* part, size_t part_floats, int* rows_out, int n, int h, int w, int c, rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && dout && w_last && wpartial && part && rows_out,
               "rd_conv3x3_last_bwd_tail_fused: null pointer");
    RD_REQUIRE(c == 16 || c == 32 || c == 64, "rd_conv3x3_last_bwd_tail_fused: C in {16, 32, 64} (got %d)", c);
    const int nb = conv_last_tail_blocks(n, h, w);
    RD_REQUIRE(part_floats >= (size_t)nb * 4 * c, "rd_conv3x3_last_bwd_tail_fused: statistics buffer too small");
    const TailSkip sk = {z, mean, invstd, gamma, beta, slope_dev, slope};
    ProfScope ps((hipStream_t)s, "conv_last_dgrad", 2.0 * n * h * w * 18.0 * c, 4.0 * n * h * w * (double)(c + 1));
    if (int e = conv_last_bwd_tail_fused_launch(sk, dout, w_last, wpartial, part, n, h, w, c, (hipStream_t)s)) return e;
    *rows_out = nb;
    return RD_OK;
}

int rd_tail_wl_finish(const double* wpartial, int nb, const double* c16, const float* wt_iohw, const float* bias_t, float* dw,
                      float* dbias, int cin, int c, rd_stream_t s) {
    RD_REQUIRE(wpartial && nb > 0 && c16 && wt_iohw && dw && cin > 0 && c > 0, "rd_tail_wl_finish: bad arguments");
    ProfScope ps((hipStream_t)s, "conv_last_wgrad", 0, 8.0 * nb * (9.0 * c + 9));
    return tail_wl_finish_launch(wpartial, nb, c16, wt_iohw, bias_t, dw, dbias, cin, c, (hipStream_t)s);
}

size_t rd_conv3x3_last_bwd_weight_tail_ws_bytes(int n, int h, int w, int c) {
    const long nt = (long)n * cdiv(w, 32) * cdiv(h, 16);
    return (size_t)(nt < 1024 ? nt : 1024) * (9 * (size_t)c + 9) * sizeof(double);
}

int rd_conv3x3_last_bwd_weight_tail(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    float slope, const float* slope_dev, const float* dout, const double* c16, const float* wt_iohw,
                                    const float* bias_t, float* dw, float* dbias, int n, int h, int w, int cin, int c, void* ws,
                                    size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(z && mean && invstd && gamma && beta && dout && c16 && wt_iohw && dw, "rd_conv3x3_last_bwd_weight_tail: null pointer");
    RD_REQUIRE((c == 16 || c == 32 || c == 64) && cin > 0, "rd_conv3x3_last_bwd_weight_tail: C in {16, 32, 64} (got %d)", c);
    const size_t need = rd_conv3x3_last_bwd_weight_tail_ws_bytes(n, h, w, c);
    if (!ws || ws_bytes < need) {
        set_error("rd_conv3x3_last_bwd_weight_tail: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    const TailSkip sk = {z, mean, invstd, gamma, beta, slope_dev, slope};
    const long nt = (long)n * cdiv(w, 32) * cdiv(h, 16);
    const int nb = (int)(nt < 1024 ? nt : 1024);
    ProfScope ps((hipStream_t)s, "conv_last_wgrad", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(c + 1));
    if (int e = conv_last_wgrad_tail_launch(sk, dout, (double*)ws, n, h, w, c, (hipStream_t)s)) return e;
    return tail_wl_finish_launch((const double*)ws, nb, c16, wt_iohw, bias_t, dw, dbias, cin, c, (hipStream_t)s);
}

int rd_convt_last_bwd_data(const float* dout, const float* V, float* dprev, int n, int hc, int wc, int cin, const float* bn_z,
                           const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                           const float* slope_dev, float* part, size_t part_floats, int* rows_out, rd_stream_t s) {
    RD_REQUIRE(dout && V && dprev && n > 0 && hc > 0 && wc > 0, "rd_convt_last_bwd_data: bad arguments");
    RD_REQUIRE(tail_shape_ok(cin), "rd_convt_last_bwd_data: Cin must be 16, 32, 64, 128 or 256 (got %d)", cin);
    RD_REQUIRE(!bn_z || (mean && invstd && gamma && beta && part && rows_out &&
                         part_floats >= rd_bn_bwd_part_floats((long long)n * hc * wc, cin)),
               "rd_convt_last_bwd_data: statistics hook arguments");
    int rows = 0;
    ProfScope ps((hipStream_t)s, "convt_last_dgrad", 2.0 * n * hc * wc * 16.0 * cin,
                 4.0 * n * hc * wc * (double)(cin * (bn_z ? 2 : 1) + 4));
    if (int e = convt_last_dgrad_launch(dout, V, dprev, n, hc, wc, cin, bn_z, mean, invstd, gamma, beta, slope, slope_dev, part,
                                        (hipStream_t)s, &rows))
        return e;
    if (rows_out) *rows_out = rows;
    return RD_OK;
}

size_t rd_convt_last_bwd_weight_ws_bytes(int n, int hc, int wc, int cin) {
    return ((size_t)tail_corr_blocks(n, hc, wc) + 1) * 16 * (size_t)cin * sizeof(double);      // block partials + C16
}

int rd_tail_t16(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                const float* slope_dev, const float* V, float* t16, long long pixels, int cin, rd_stream_t s) {
    RD_REQUIRE(z && V && t16 && pixels > 0, "rd_tail_t16: bad arguments");
    RD_REQUIRE(!mean || (invstd && gamma && beta), "rd_tail_t16: incomplete BN descriptor");
    RD_REQUIRE(tail_shape_ok(cin), "rd_tail_t16: Cin must be 16, 32, 64, 128 or 256 (got %d)", cin);
    const TailSkip sk = {z, mean, invstd, gamma, beta, slope_dev, slope};
    ProfScope ps((hipStream_t)s, "tail_t16", 2.0 * pixels * cin * 16.0, 4.0 * pixels * (double)(cin + 16));
    return tail_t16_launch(sk, V, t16, (long)pixels, cin, (hipStream_t)s);
}

int rd_convt_last_bwd_weight(const float* x, const float* dout, const float* w_last, float* dwt_iohw, double* c16_out, int n, int hc,
                             int wc, int cin, int c0, void* ws, size_t ws_bytes, rd_stream_t s) {
    return rd_convt_last_bwd_weight_bn(x, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, dout, w_last, dwt_iohw, c16_out, n, hc, wc,
                                       cin, c0, ws, ws_bytes, s);
}

int rd_convt_last_bwd_weight_bn(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float slope, const float* slope_dev, const float* dout, const float* w_last, float* dwt_iohw,
                                double* c16_out, int n, int hc, int wc, int cin, int c0, void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(x && dout && w_last && dwt_iohw && n > 0 && hc > 0 && wc > 0 && c0 > 0, "rd_convt_last_bwd_weight: bad arguments");
    RD_REQUIRE(!mean || (invstd && gamma && beta), "rd_convt_last_bwd_weight_bn: incomplete BN descriptor");
    RD_REQUIRE(tail_shape_ok(cin), "rd_convt_last_bwd_weight: Cin must be 16, 32, 64, 128 or 256 (got %d)", cin);
    const size_t need = rd_convt_last_bwd_weight_ws_bytes(n, hc, wc, cin);
    if (!ws || ws_bytes < need) {
        set_error("rd_convt_last_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    const int nb = tail_corr_blocks(n, hc, wc);
    double* partial = (double*)ws;
    double* c16 = c16_out ? c16_out : partial + (size_t)nb * 16 * cin;
    ProfScope ps((hipStream_t)s, "convt_last_wgrad", 2.0 * n * hc * wc * 16.0 * cin, 4.0 * n * hc * wc * (double)(cin + 4));
    const TailSkip sk = {x, mean, invstd, gamma, beta, slope_dev, slope};
    return convt_last_wgrad_launch(x, sk, dout, w_last, dwt_iohw, partial, c16, n, hc, wc, cin, c0, (hipStream_t)s);
}

size_t rd_conv3x3_last_bwd_weight_ws_bytes(int n, int h, int w, int c) {
    if (const int nb = conv_last_wgrad_blocks(n, h, w, c)) return (size_t)nb * (9 * c + 1) * sizeof(double);
    RowPlan pl;
    if (!plan_rows((long)n * h * w, c, &pl, last_blocks())) return 0;
    return (size_t)pl.nb * (9 * c + 1) * sizeof(double);
}

int rd_conv3x3_last_bwd_weight(const float* s_in, const float* dout, float* dw, float* dbias, int n, int h, int w,
                               int c, void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(s_in && dout && dw, "rd_conv3x3_last_bwd_weight: null pointer");
    RowPlan pl;
    RD_REQUIRE(plan_rows((long)n * h * w, c, &pl, last_blocks()), "rd_conv3x3_last_bwd_weight: C must be a multiple of 4, <= 1024");
    RD_REQUIRE((long)n * h * w < (1L << 31), "rd_conv3x3_last_bwd_weight: pixel count must be < 2^31");
    const size_t need = rd_conv3x3_last_bwd_weight_ws_bytes(n, h, w, c);
    if (!ws || ws_bytes < need) {
        set_error("rd_conv3x3_last_bwd_weight: workspace too small (%zu < %zu)", ws_bytes, need);
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "conv_last_wgrad", 2.0 * n * h * w * 9.0 * c, 4.0 * n * h * w * (double)(c + 1));
    if (const int nb = conv_last_wgrad_blocks(n, h, w, c)) {
        if (int e = conv_last_wgrad_launch(s_in, dout, (double*)ws, n, h, w, c, (hipStream_t)s)) return e;
        RD_LAUNCH(last_wgrad_reduce_kernel, dim3(cdiv(9 * c + 1, 16)), dim3(256), 0, (hipStream_t)s,
                           (const double*)ws, dw, dbias, nb, c);
        RD_LAUNCH_CHECK("conv_last_wgrad");
        return RD_OK;
    }
    RD_LAUNCH(conv_last_wgrad_kernel, dim3(pl.nb), dim3(256), 0, (hipStream_t)s, s_in, dout, (double*)ws,
                       (long)n * h * w, h, w, c, pl.CQ, pl.RP, pl.rows_per_block);
    RD_LAUNCH(last_wgrad_reduce_kernel, dim3(cdiv(9 * c + 1, 16)), dim3(256), 0, (hipStream_t)s,
                       (const double*)ws, dw, dbias, pl.nb, c);
    RD_LAUNCH_CHECK("conv_last_wgrad");
    return RD_OK;
}

// ---- loss ------------------------------------------------------------------------------------------
static int l1_grid(long total) { return grid_cap((total + 255) / 256, 1024); }

size_t rd_masked_l1_ws_bytes(long long numel) { return (size_t)l1_grid(numel) * 2 * sizeof(double); }

int rd_masked_l1_partial(const float* yp, const float* y, const uint8_t* mask, const float* mean, const float* stdv,
                         double* sums, int n, long long pps, void* ws, size_t ws_bytes, rd_stream_t s) {
    RD_REQUIRE(yp && y && mask && mean && stdv && sums && n > 0 && pps > 0, "rd_masked_l1_partial: bad arguments");
    const long total = (long)n * pps;
    const int nb = l1_grid(total);
    if (!ws || ws_bytes < (size_t)nb * 2 * sizeof(double)) {
        set_error("rd_masked_l1_partial: workspace too small");
        return RD_ERR_WS;
    }
    ProfScope ps((hipStream_t)s, "masked_l1", 0, 9.0 * total);
    RD_LAUNCH(l1_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)s, yp, y, mask, mean, stdv, (double*)ws,
                       total, (long)pps);
    RD_LAUNCH(l1_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const double*)ws, sums, nb);
    RD_LAUNCH_CHECK("masked_l1_partial");
    return RD_OK;
}

int rd_masked_l1_finish(const float* yp, const float* y, const uint8_t* mask, const float* mean, const float* stdv,
                        const double* sums, double numel_total, const float* gout, float* loss, float* dyp, int n,
                        long long pps, rd_stream_t s) {
    RD_REQUIRE(yp && y && mask && mean && stdv && sums && n > 0 && pps > 0, "rd_masked_l1_finish: bad arguments");
    const long total = (long)n * pps;
    ProfScope ps((hipStream_t)s, "masked_l1", 0, 13.0 * total);
    RD_LAUNCH(l1_finish_kernel, dim3(dyp ? l1_grid(total) : 1), dim3(256), 0, (hipStream_t)s, yp, y, mask, mean,
                       stdv, sums, numel_total, gout, loss, dyp, total, (long)pps);
    RD_LAUNCH_CHECK("masked_l1_finish");
    return RD_OK;
}

// ---- Adam ------------------------------------------------------------------------------------------
int rd_adam_step(float* p, const float* g, float* m, float* v, long long numel, double beta1, double beta2, float eps,
                 float weight_decay, float step_size, float bc2_sqrt, float grad_scale, rd_stream_t s) {
    RD_REQUIRE(p && g && m && v && numel > 0, "rd_adam_step: bad arguments");
    ProfScope ps((hipStream_t)s, "adam", 0, 28.0 * numel);
    RD_LAUNCH(adam_kernel, dim3(grid_cap((numel + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)s, p, g, m, v,
                       (long)numel, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, step_size, bc2_sqrt,
                       grad_scale);
    RD_LAUNCH_CHECK("adam");
    return RD_OK;
}

int rd_adam_step_dev(float* p, const float* g, float* m, float* v, long long numel, const float* scalars_dev, rd_stream_t s) {
    RD_REQUIRE(p && g && m && v && scalars_dev && numel > 0, "rd_adam_step_dev: bad arguments");
    ProfScope ps((hipStream_t)s, "adam", 0, 28.0 * numel);
    RD_LAUNCH(adam_dev_kernel, dim3(grid_cap((numel + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)s, p, g, m, v,
                       (long)numel, scalars_dev);
    RD_LAUNCH_CHECK("adam_dev");
    return RD_OK;
}

int rd_sgd_step(float* p, const float* g, float* momentum_buf, long long numel, float lr, float weight_decay, float momentum,
                float dampening, int nesterov, int first_step, float grad_scale, rd_stream_t s) {
    RD_REQUIRE(p && g && numel > 0, "rd_sgd_step: bad arguments");
    RD_REQUIRE(momentum == 0.f || momentum_buf, "rd_sgd_step: momentum != 0 needs a momentum buffer");
    ProfScope ps((hipStream_t)s, "sgd", 0, (momentum != 0.f ? 20.0 : 12.0) * numel);
    const dim3 grid(grid_cap((numel + 255) / 256, 8192));
    if (momentum != 0.f)
        RD_LAUNCH(sgd_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, p, g, momentum_buf, (long)numel, lr, weight_decay,
                           momentum, 1.f - dampening, nesterov, first_step, grad_scale);
    else
        RD_LAUNCH(sgd_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, p, g, (float*)nullptr, (long)numel, lr,
                           weight_decay, 0.f, 1.f, 0, 0, grad_scale);
    RD_LAUNCH_CHECK("sgd");
    return RD_OK;
}

int rd_blend_accumulate(const float* pred, const float* mean, const float* stdv, const int* pos, const int* reg, int n,
                        int tile_size, int stride, double* raster, int rows, int cols, rd_stream_t s) {
    RD_REQUIRE(pred && mean && stdv && pos && reg && raster, "rd_blend_accumulate: null pointer");
    RD_REQUIRE(n > 0 && tile_size > 0 && stride > 0 && stride <= tile_size && rows > 0 && cols > 0,
               "rd_blend_accumulate: bad shape (n=%d tile=%d stride=%d raster=%dx%d)", n, tile_size, stride, rows, cols);
    ProfScope ps((hipStream_t)s, "blend_accumulate", 0, 20.0 * n * tile_size * tile_size);
    if (n <= 64) {      // one launch for the whole batch (owner threads walk the covering tiles in order)
        const long total = (long)n * tile_size * tile_size;
        RD_LAUNCH(blend_batch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, pred, mean, stdv,
                           pos, reg, n, tile_size, stride, raster, rows, cols);
        RD_LAUNCH_CHECK("blend_accumulate");
        return RD_OK;
    }
    const int blocks = cdiv((long)tile_size * tile_size, 256);
    for (int i = 0; i < n; ++i)   // one launch per tile, in order: overlapping tiles never race, order is fixed
        RD_LAUNCH(blend_tile_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, pred, mean, stdv, pos, reg, i,
                           tile_size, stride, raster, rows, cols);
    RD_LAUNCH_CHECK("blend_accumulate");
    return RD_OK;
}

int rd_patch_sums(const float* planes, long long plane_stride, const int* plane_idx, int p_per_patch, const int* pos,
                  int n, int tile, int width, float nodata, int use_nodata, double* sums, rd_stream_t s) {
    RD_REQUIRE(planes && plane_idx && pos && sums && n > 0 && tile > 0 && width >= tile && p_per_patch > 0,
               "rd_patch_sums: bad arguments");
    ProfScope ps((hipStream_t)s, "patch_sums", 0, 4.0 * n * p_per_patch * tile * tile);
    RD_LAUNCH(patch_sums_kernel, dim3(n), dim3(256), 0, (hipStream_t)s, planes, (long)plane_stride, plane_idx,
                       p_per_patch, pos, tile, width, nodata, use_nodata, sums);
    RD_LAUNCH_CHECK("patch_sums");
    return RD_OK;
}

int rd_assemble_patches(const float* dsm_in, const float* dsm_gt, const float* ortho_planes, long long plane_stride,
                        const int* pair_idx, int views, const int* pos, const int* aug, const float* dsm_mean,
                        float dsm_std, const float* ortho_mean, float ortho_std, float nodata, int n, int tile, int width,
                        float* input, float* target, uint8_t* mask, rd_stream_t s) {
    RD_REQUIRE(dsm_in && pos && dsm_mean && input && n > 0 && tile > 0 && width >= tile && views >= 0,
               "rd_assemble_patches: bad arguments");
    RD_REQUIRE(views == 0 || (ortho_planes && pair_idx && ortho_mean), "rd_assemble_patches: ortho inputs missing");
    RD_REQUIRE(!dsm_gt || (target && mask), "rd_assemble_patches: target / mask outputs missing");
    const long total = (long)n * (views + 2) * tile * tile;
    ProfScope ps((hipStream_t)s, "assemble_patches", 0, 8.0 * total);
    RD_LAUNCH(assemble_patches_kernel, dim3(grid_cap((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)s,
                       dsm_in, dsm_gt, ortho_planes, (long)plane_stride, pair_idx, views, pos, aug, dsm_mean, dsm_std,
                       ortho_mean, ortho_std, nodata, n, tile, width, input, target, mask);
    RD_LAUNCH_CHECK("assemble_patches");
    return RD_OK;
}

int rd_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, rd_stream_t s) {
    RD_REQUIRE(src && dst, "rd_nchw_to_nhwc: null pointer");
    const long total = (long)n * c * h * w;
    RD_LAUNCH(nchw_to_nhwc_kernel, dim3(grid_cap((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)s, src,
                       dst, n, c, h * w);
    RD_LAUNCH_CHECK("nchw_to_nhwc");
    return RD_OK;
}

int rd_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, rd_stream_t s) {
    RD_REQUIRE(src && dst, "rd_nhwc_to_nchw: null pointer");
    const long total = (long)n * c * h * w;
    RD_LAUNCH(nhwc_to_nchw_kernel, dim3(grid_cap((total + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)s, src,
                       dst, n, c, h * w);
    RD_LAUNCH_CHECK("nhwc_to_nchw");
    return RD_OK;
}

}  // extern "C"
