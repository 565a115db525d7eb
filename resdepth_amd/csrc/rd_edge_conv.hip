// Streaming kernels of the last convolution C -> 1 (+ bias + outer residual x[:, 0:1]) for the channel counts the network
// uses (C = 16 / 32 / 64: in the forward a pixel is 2 / 4 / 8 lanes of 8 channels, in the gradients 4 / 8 / 16 lanes of 4),
// and of the first convolution (1..4 input channels -> 32 / 64 / 128, segment kernels further down).  Replaces the ATen kernels behind the reference's
// `last_layer` + outer SkipConnection and their autograd (lib/UNet.py:184, 227-244).  All three are HBM-bound: they move
// the 64-channel full-resolution tensor exactly once (537 MB at cfg-S) and nothing else of size.
//
//  forward   every input pixel is read ONCE as a whole C-channel row (256 contiguous bytes at C = 64; the previous kernel
//            re-read a 10x34 halo per 16-channel chunk in 64-byte pieces: 1.78x the algorithmic HBM traffic).  The 8 lanes
//            of a pixel (8 channels each) form its nine tap dot products V[q][tap] = sum_c s[q][c] w[c][tap] (weights in registers, lane
//            reduction with DPP adds -- no LDS, no barrier), V goes to LDS (612 halo pixels x 9 floats per 16x32 tile), and
//            out[p] = bias + x0[p] + sum_tap V[p + tap][tap] is nine conflict-free LDS reads per output pixel.
//  dgrad     ds[q][c] = sum_tap dout[q - tap] w[c][tap]: the 1-channel dout tile (+ halo) sits in LDS, every lane keeps its
//            4 channels x 9 taps of weights in registers and streams 16-byte stores (a wave writes 1 KB contiguous).
//            Template BN = true: it also reads the consumer block's z and emits that block's BN-backward sums (BnHook).
//  wgrad     dw[c][tap] = sum_q s[q][c] dout[q - tap]: same LDS tile of dout, four 16-byte loads of s in flight per lane,
//            36 accumulators per lane, ONE block-level reduction at the end (LDS, two barriers) instead of 9 x 2.
#include "rd_common.h"
#include "rd_mfma_dev.h"

namespace rd {

// TUNE_EDGE_CONV: -1 all kernels of this file, else a bit mask (1 last fwd, 2 last dgrad, 4 last wgrad, 8 first fwd, 16 first wgrad,
// 32 the composed tail kernels); 128 the first convolution's fused weight gradient on the matrix pipe; bits 64 / 256 (opt-in only, NOT implied by -1): the first
// convolution's forward / the fused head of the backward there)
static bool edge_on(int bit) {
    const int v = tune(TUNE_EDGE_CONV);
    return v < 0 || (v & bit);
}

constexpr int ET_H = 16, ET_W = 32, EH_W = ET_W + 2, EH_H = ET_H + 2, EH_NP = EH_H * EH_W;   // 16x32 tile, 612 halo pixels
constexpr int EVS = 9;          // V row stride in floats (odd: column reads of one tap are bank-conflict free)

template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the LP consecutive lanes of a pixel (LP = 2, 4, 8, 16; groups are aligned inside a 16-lane DPP row); every lane
// ends up with the total.  Fixed pairing order => deterministic.
template <int LP>
__device__ __forceinline__ float lane_group_sum(float v) {
    v = dpp_xadd<0xB1>(v);                        // quad_perm [1,0,3,2]
    if (LP >= 4) v = dpp_xadd<0x4E>(v);           // quad_perm [2,3,0,1]
    if (LP >= 8) v = dpp_xadd<0x141>(v);          // row_half_mirror: quad 0 <-> quad 1
    if (LP >= 16) v = dpp_xadd<0x140>(v);         // row_mirror: half 0 <-> half 1
    return v;
}

// LP = lanes per pixel, each lane owns 8 consecutive channels (two 16-byte loads): C = 8 * LP
template <int LP>
__global__ __launch_bounds__(256) void conv_last_fwd_dpp_kernel(const float* __restrict__ s_in, const float* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ x_nchw, int xc,
                                                                float* __restrict__ out, int N, int H, int W, int tiles_x,
                                                                int tiles_y) {
    constexpr int C = LP * 8, PPI = 256 / LP;     // pixels per block iteration
    __shared__ float V[EH_NP * EVS];
    const int t = threadIdx.x, q = t % LP, slot = t / LP;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    float wr[9][8];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[tap][k] = w[(q * 8 + k) * 9 + tap];
    constexpr int UN = 2;                          // pixels (2 x 16-byte loads each) in flight per lane
    for (int p0 = 0; p0 < EH_NP; p0 += PPI * UN) {
        float4 v[UN][2];
        int hp[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            hp[u] = p0 + u * PPI + slot;
            const int hy = hp[u] / EH_W, hx = hp[u] - hy * EH_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding / tile overhang
            if (hp[u] < EH_NP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const float* src = s_in + (((long)n * H + gy) * W + gx) * C + q * 8;
                v[u][0] = *reinterpret_cast<const float4*>(src);
                v[u][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const float sv[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
            float pt[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float a = sv[0] * wr[tap][0];
#pragma unroll
                for (int k = 1; k < 8; ++k) a = fmaf(sv[k], wr[tap][k], a);
                pt[tap] = lane_group_sum<LP>(a);
            }
            if (q == 0 && hp[u] < EH_NP) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) V[hp[u] * EVS + tap] = pt[tap];
            }
        }
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
    for (int e = t; e < ET_H * ET_W; e += 256) {
        const int py = e / ET_W, px = e - py * ET_W;
        const int gy = y0 + py, gx = x0 + px;
        if (gy >= H || gx >= W) continue;
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc += V[((py + tap / 3) * EH_W + px + tap % 3) * EVS + tap];
        acc += b0;
        if (x_nchw) acc = x_nchw[(((long)n * xc) * H + gy) * W + gx] + acc;
        out[((long)n * H + gy) * W + gx] = acc;
    }
}

// The same convolution when its input s = up-convolution(x_coarse) + bias_t + act(BN(z)) is NOT a tensor (tail of the network):
//   out[q] = conv_last(act(BN(z)))[q]                                   -- z read here, BN + activation applied on load
//          + sum over the <= 4 coarse pixels p' with d = q - 2p' in [-1, 2]^2 of T[p'][d]     -- T = x_coarse . V (1x1 conv, 16 ch)
//          + sum over the taps whose pixel q + off(tap) lies inside the image of B9[tap]       -- the up-convolution's bias
//          + bias + x[:, 0]
template <int LP>
__global__ __launch_bounds__(256) void conv_last_fwd_tail_kernel(TailSkip sk, const float* __restrict__ t16, const float* __restrict__ b9,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 const float* __restrict__ x_nchw, int xc, float* __restrict__ out,
                                                                 int N, int H, int W, int tiles_x, int tiles_y) {
    constexpr int C = LP * 8, PPI = 256 / LP;
    __shared__ float V[EH_NP * EVS];
    const int t = threadIdx.x, q = t % LP, slot = t / LP;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    float wr[9][8];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[tap][k] = w[(q * 8 + k) * 9 + tap];
    float sc[8], sh[8];
    const float slope = sk.slope_dev ? sk.slope_dev[0] : sk.slope;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = q * 8 + k;
        sc[k] = sk.invstd[c] * sk.gamma[c];
        sh[k] = sk.beta[c] - sk.mean[c] * sc[k];
    }
    constexpr int UN = 2;
    for (int p0 = 0; p0 < EH_NP; p0 += PPI * UN) {
        float4 v[UN][2];
        int hp[UN];
        bool in[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            hp[u] = p0 + u * PPI + slot;
            const int hy = hp[u] / EH_W, hx = hp[u] - hy * EH_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            in[u] = hp[u] < EH_NP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            if (in[u]) {
                const float* src = sk.z + (((long)n * H + gy) * W + gx) * C + q * 8;
                v[u][0] = *reinterpret_cast<const float4*>(src);
                v[u][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float sv[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {          // zero padding applies to the ACTIVATION: pixels outside stay 0
                const float y = fmaf(sv[k], sc[k], sh[k]);
                sv[k] = in[u] ? (y > 0.f ? y : y * slope) : 0.f;
            }
            float pt[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float a = sv[0] * wr[tap][0];
#pragma unroll
                for (int k = 1; k < 8; ++k) a = fmaf(sv[k], wr[tap][k], a);
                pt[tap] = lane_group_sum<LP>(a);
            }
            if (q == 0 && hp[u] < EH_NP) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) V[hp[u] * EVS + tap] = pt[tap];
            }
        }
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
    const int Hc = H >> 1, Wc = W >> 1;
    float bq[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bq[tap] = b9[tap];
    for (int e = t; e < ET_H * ET_W; e += 256) {
        const int py = e / ET_W, px = e - py * ET_W;
        const int gy = y0 + py, gx = x0 + px;
        if (gy >= H || gx >= W) continue;
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc += V[((py + tap / 3) * EH_W + px + tap % 3) * EVS + tap];
        // up-convolution part: q even -> (p', d) = (q/2, 0), (q/2 - 1, 2); q odd -> ((q-1)/2, 1), ((q+1)/2, -1)
        float up = 0.f;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const int cy = (gy >> 1) + ((gy & 1) ? jy : -jy), dy = (gy & 1) ? (jy ? -1 : 1) : (jy ? 2 : 0);
            if ((unsigned)cy >= (unsigned)Hc) continue;
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int cx = (gx >> 1) + ((gx & 1) ? jx : -jx), dx = (gx & 1) ? (jx ? -1 : 1) : (jx ? 2 : 0);
                if ((unsigned)cx >= (unsigned)Wc) continue;
                up += t16[(((long)n * Hc + cy) * Wc + cx) * 16 + (dy + 1) * 4 + dx + 1];
            }
        }
        float bsum = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ry = gy + tap / 3 - 1, rx = gx + tap % 3 - 1;
            if ((unsigned)ry < (unsigned)H && (unsigned)rx < (unsigned)W) bsum += bq[tap];
        }
        acc = ((acc + up) + bsum) + b0;
        if (x_nchw) acc = x_nchw[(((long)n * xc) * H + gy) * W + gx] + acc;
        out[((long)n * H + gy) * W + gx] = acc;
    }
}

// dout tile (+ one-pixel halo, zero outside the image) -> LDS; D[hy][hx] = dout[y0 - 1 + hy][x0 - 1 + hx]
__device__ __forceinline__ void load_dout_tile(float* D, const float* __restrict__ dout, int n, int y0, int x0, int H, int W,
                                               int t) {
    for (int e = t; e < EH_NP; e += 256) {
        const int hy = e / EH_W, hx = e - hy * EH_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        D[e] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? dout[((long)n * H + gy) * W + gx] : 0.f;
    }
}

// BN-backward statistics of the block whose activation gradient the kernel writes (the arithmetic of bn_act_bwd_kernel's
// reduction: sum g', sum g' xhat, sum g, sum_{y<=0} g y per channel); per-tile partial rows [tile][4][C]
struct BnHook {
    const float* z;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* slope_dev;
    float slope;
    float* part;
};

// ds[q][c] = sum_tap dout[q - off(tap)] * w[c][tap], off(tap) = (tap/3 - 1, tap%3 - 1)
template <bool BN>
__global__ __launch_bounds__(256) void conv_last_dgrad_tile_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                   float* __restrict__ ds, int N, int H, int W, int C, int CQ,
                                                                   int tiles_x, int tiles_y, BnHook bn) {
    __shared__ float D[EH_NP];
    __shared__ float red[BN ? 256 * 16 : 1];
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    float wr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 4; ++k) wr[tap][k] = w[(q * 4 + k) * 9 + tap];
    float sc[4], sh[4], mu[4], is[4], bacc[16], slope = 0.f;
    if (BN) {
        slope = bn.slope_dev ? bn.slope_dev[0] : bn.slope;
        const float4 m4 = *reinterpret_cast<const float4*>(bn.mean + q * 4), i4 = *reinterpret_cast<const float4*>(bn.invstd + q * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(bn.gamma + q * 4), b4 = *reinterpret_cast<const float4*>(bn.beta + q * 4);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = mm[k];
            is[k] = ii[k];
            sc[k] = ii[k] * gg[k];
            sh[k] = bb[k] - mm[k] * sc[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) bacc[k] = 0.f;
    }
    load_dout_tile(D, dout, n, y0, x0, H, W, t);
    __syncthreads();
    for (int e = slot; e < ET_H * ET_W; e += PPI) {
        const int py = e / ET_W, px = e - py * ET_W;
        const int gy = y0 + py, gx = x0 + px;
        if (gy >= H || gx >= W) continue;
        const long o = (((long)n * H + gy) * W + gx) * C + q * 4;
        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BN) z4 = *reinterpret_cast<const float4*>(bn.z + o);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // dout at q - off(tap): halo coordinates (py + 1 - dy, px + 1 - dx) = (py + 2 - tap/3, px + 2 - tap%3)
            const float d = D[(py + 2 - tap / 3) * EH_W + px + 2 - tap % 3];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(d, wr[tap][k], acc[k]);
        }
        if (ds) *reinterpret_cast<float4*>(ds + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);   // NULL: statistics only
        if (BN) {
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y = fmaf(zz[k], sc[k], sh[k]);
                const float gm = acc[k] * (y > 0.f ? 1.f : slope);
                const float xh = (zz[k] - mu[k]) * is[k];
                bacc[k] += gm;
                bacc[4 + k] = fmaf(gm, xh, bacc[4 + k]);
                bacc[8 + k] += acc[k];
                if (!(y > 0.f)) bacc[12 + k] = fmaf(acc[k], y, bacc[12 + k]);
            }
        }
    }
    if (BN) {
        // fixed-order combination of the PPI pixel slots of every channel quad
#pragma unroll
        for (int k = 0; k < 16; ++k) red[t * 16 + k] = bacc[k];
        __syncthreads();
        for (int o = t; o < 4 * C; o += 256) {
            const int sidx = o / C, c = o - sidx * C;
            float sum = 0.f;
            for (int sl = 0; sl < PPI; ++sl) sum += red[(sl * CQ + (c >> 2)) * 16 + sidx * 4 + (c & 3)];
            bn.part[((long)tl * 4 + sidx) * C + c] = sum;
        }
    }
}

// ---- the tail of the network: last up-convolution (ConvTranspose2d k2 s2, Cin -> C0, lib/UNet.py:21,218-225) followed by the
// last convolution (C0 -> 1, 3x3, lib/UNet.py:227).  Both are linear, so the gradient w.r.t. the up-convolution's INPUT is a
// 16-tap stride-2 stencil on the 1-channel output gradient:
//   g[q][co]     = sum_tap dout[q - off(tap)] wl[co][tap]                       (last conv, data gradient; off = (tap/3-1, tap%3-1))
//   dprev[p][ci] = sum_{a,b,co} g[2p + (a,b)][co] Wt[ci][co][a][b]              (up-convolution, data gradient)
//                = sum_{d in 4x4} dout[2p + d] V[ci][d],   d = (a,b) - off(tap) in [-1, 2]^2,
//   M[ci][ab][tap] = sum_co Wt[ci][co][a][b] wl[co][tap],   V[ci][d] = sum over the (ab, tap) with (a,b) - off(tap) = d of M.
// The C0-channel gradient g at full resolution (537 MB in cfg-S) is then not an operand of this layer at all.
// tail_compose_kernel: one block per ci; M [Cin][4][9] and V [Cin][16] (fp64 accumulation, fp32 results).
// VT [16][Cin] (nullable): V transposed = the weight of a 1x1 convolution Cin -> 16 (forward use); B9 [9] (nullable):
// B9[tap] = sum_co wl[co][tap] bt[co], the up-convolution's bias seen through the last convolution.
__global__ __launch_bounds__(64) void tail_compose_kernel(const float* __restrict__ wt, const float* __restrict__ bt,
                                                          const float* __restrict__ wl, float* __restrict__ M, float* __restrict__ V,
                                                          float* __restrict__ VT, float* __restrict__ B9, int Cin, int C0) {
    __shared__ float m[36];
    const int ci = blockIdx.x, t = threadIdx.x;
    if (ci == 0 && B9 && t >= 48 && t < 57) {
        const int tap = t - 48;
        double acc = 0.0;
        if (bt)
            for (int co = 0; co < C0; ++co) acc += (double)wl[co * 9 + tap] * (double)bt[co];
        B9[tap] = (float)acc;
    }
    if (t < 36) {
        const int ab = t / 9, tap = t - ab * 9;
        double acc = 0.0;
        for (int co = 0; co < C0; ++co) acc += (double)wt[((long)ci * C0 + co) * 4 + ab] * (double)wl[co * 9 + tap];
        m[t] = (float)acc;
        M[(long)ci * 36 + t] = (float)acc;
    }
    __syncthreads();
    if (t < 16) {
        const int dy = t / 4 - 1, dx = t % 4 - 1;
        double acc = 0.0;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const int oy = a - dy, ox = b - dx;          // off(tap) = (a, b) - d
                if (oy < -1 || oy > 1 || ox < -1 || ox > 1) continue;
                acc += (double)m[(a * 2 + b) * 9 + (oy + 1) * 3 + (ox + 1)];
            }
        V[(long)ci * 16 + t] = (float)acc;
        if (VT) VT[(long)t * Cin + ci] = (float)acc;
    }
}

// T[p][d] = sum_ci a[p][ci] V[ci][d], a = act(BN(z)) evaluated on load (identity when sk.mean == NULL: z IS the activation):
// the up-convolution's input contracted with the 16 stencil columns, on the exact-f32 matrix pipe.  One wave = 16 pixels per
// iteration: lane (m = l % 16, j = l / 16) loads the CPL = Cin / 4 contiguous channels [j CPL, (j + 1) CPL) of pixel m (16-byte
// loads) and feeds v_mfma_f32_16x16x4_f32 number i with A[m][j] = a[m][j CPL + i], B[j][n] = V[j CPL + i][n] -- any partition
// of the channels into groups of four is a valid K order.  HBM-bound: reads the tensor once, writes 64 bytes per pixel.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int CPL>
__global__ __launch_bounds__(256) void tail_t16_kernel(TailSkip sk, const float* __restrict__ V, float* __restrict__ t16, long P) {
    constexpr int C = CPL * 4;
    __shared__ float scs[C], shs[C];
    const int t = threadIdx.x, lane = t & 63, m = lane & 15, j = lane >> 4;
    const bool ident = sk.mean == nullptr;
    for (int c = t; c < C; c += 256) {
        const float scv = ident ? 1.f : sk.invstd[c] * sk.gamma[c];
        scs[c] = scv;
        shs[c] = ident ? 0.f : sk.beta[c] - sk.mean[c] * scv;
    }
    const float slope = ident ? 1.f : (sk.slope_dev ? sk.slope_dev[0] : sk.slope);
    float bv[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) bv[i] = V[(long)(j * CPL + i) * 16 + m];        // B[k = j][n = m] of MFMA i
    __syncthreads();
    const long ntile = (P + 15) / 16, wave0 = (long)blockIdx.x * 4 + (t >> 6), nwave = (long)gridDim.x * 4;
    for (long tile = wave0; tile < ntile; tile += nwave) {
        const long pix = tile * 16 + m;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        float a[CPL];
#pragma unroll
        for (int g = 0; g < CPL / 4; ++g) {
            const float4 z4 = pix < P ? ld_nt4(sk.z + pix * C + j * CPL + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            a[g * 4 + 0] = z4.x; a[g * 4 + 1] = z4.y; a[g * 4 + 2] = z4.z; a[g * 4 + 3] = z4.w;
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float y = fmaf(a[i], scs[j * CPL + i], shs[j * CPL + i]);
            const float av = pix < P ? (y > 0.f ? y : y * slope) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[i], acc, 0, 0, 0);
        }
        // C[row = 4 (l / 16) + r][col = l % 16]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long op = tile * 16 + 4 * j + r;
            if (op < P) t16[op * 16 + m] = acc[r];
        }
    }
}

constexpr int TL_W = 2 * ET_W + 3, TL_H = 2 * ET_H + 3;     // dout region of a 16 x 32 coarse tile: rows 2 y0 - 1 .. 2 y0 + 2 ET_H + 1

// dprev [N][Hc][Wc][C] = stencil above; BN: + the BN-backward statistics of the block whose activation gradient this is
// (its z at the coarse resolution), per-tile partial rows like conv_last_dgrad_tile_kernel
template <bool BN>
__global__ __launch_bounds__(256) void convt_last_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ V,
                                                               float* __restrict__ dprev, int N, int Hc, int Wc, int C, int CQ,
                                                               int tiles_x, int tiles_y, BnHook bn) {
    __shared__ float D[TL_H * TL_W];
    __shared__ float red[BN ? 256 * 16 : 1];
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W, H = 2 * Hc, W = 2 * Wc;
    float v[16][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int d = 0; d < 16; ++d) v[d][k] = V[(long)(q * 4 + k) * 16 + d];
    float sc[4], sh[4], mu[4], is[4], bacc[16], slope = 0.f;
    if (BN) {
        slope = bn.slope_dev ? bn.slope_dev[0] : bn.slope;
        const float4 m4 = *reinterpret_cast<const float4*>(bn.mean + q * 4), i4 = *reinterpret_cast<const float4*>(bn.invstd + q * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(bn.gamma + q * 4), b4 = *reinterpret_cast<const float4*>(bn.beta + q * 4);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = mm[k];
            is[k] = ii[k];
            sc[k] = ii[k] * gg[k];
            sh[k] = bb[k] - mm[k] * sc[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) bacc[k] = 0.f;
    }
    for (int e = t; e < TL_H * TL_W; e += 256) {
        const int hy = e / TL_W, hx = e - hy * TL_W;
        const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
        D[e] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? dout[((long)n * H + gy) * W + gx] : 0.f;
    }
    __syncthreads();
    for (int e = slot; e < ET_H * ET_W; e += PPI) {
        const int py = e / ET_W, px = e - py * ET_W;
        const int gy = y0 + py, gx = x0 + px;
        if (gy >= Hc || gx >= Wc) continue;
        const long o = (((long)n * Hc + gy) * Wc + gx) * C + q * 4;
        float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BN) z4 = *reinterpret_cast<const float4*>(bn.z + o);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const float dv = D[(2 * py + d / 4) * TL_W + 2 * px + d % 4];       // dout[2p + (d/4 - 1, d%4 - 1)]
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = fmaf(dv, v[d][k], acc[k]);
        }
        *reinterpret_cast<float4*>(dprev + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (BN) {
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y = fmaf(zz[k], sc[k], sh[k]);
                const float gm = acc[k] * (y > 0.f ? 1.f : slope);
                const float xh = (zz[k] - mu[k]) * is[k];
                bacc[k] += gm;
                bacc[4 + k] = fmaf(gm, xh, bacc[4 + k]);
                bacc[8 + k] += acc[k];
                if (!(y > 0.f)) bacc[12 + k] = fmaf(acc[k], y, bacc[12 + k]);
            }
        }
    }
    if (BN) {
#pragma unroll
        for (int k = 0; k < 16; ++k) red[t * 16 + k] = bacc[k];
        __syncthreads();
        for (int o = t; o < 4 * C; o += 256) {
            const int sidx = o / C, c = o - sidx * C;
            float sum = 0.f;
            for (int sl = 0; sl < PPI; ++sl) sum += red[(sl * CQ + (c >> 2)) * 16 + sidx * 4 + (c & 3)];
            bn.part[((long)tl * 4 + sidx) * C + c] = sum;
        }
    }
}

// Weight gradient of the last up-convolution, same composition:
//   dWt[ci][co][a][b] = sum_p x[p][ci] g[2p + (a,b)][co] = sum_tap wl[co][tap] C16[ci][(a,b) - off(tap)],
//   C16[ci][d] = sum_p x[p][ci] dout[2p + d]                        (16 correlations per input channel)
// tail_corr_kernel: persistent blocks over 16 x 32 coarse tiles, per block one partial row [16][C] of doubles.
// sk.mean != NULL: x = act(BN(sk.z)) evaluated on load (x itself unused)
__global__ __launch_bounds__(256) void tail_corr_kernel(const float* __restrict__ x, TailSkip sk, const float* __restrict__ dout,
                                                        double* __restrict__ partial, int N, int Hc, int Wc, int C, int CQ,
                                                        int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];     // D[TL_H * TL_W] (padded to 2368), then the reduction scratch
    float* D = tsm;
    float* red = tsm + 2368;                       // [PPI][16][C] floats = 16384
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    const int H = 2 * Hc, W = 2 * Wc;
    const bool lazy = sk.mean != nullptr;
    const float* src = lazy ? sk.z : x;
    float scq[4] = {1.f, 1.f, 1.f, 1.f}, shq[4] = {0.f, 0.f, 0.f, 0.f}, slope = 1.f;
    if (lazy) {
        slope = sk.slope_dev ? sk.slope_dev[0] : sk.slope;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = q * 4 + k;
            scq[k] = sk.invstd[c] * sk.gamma[c];
            shq[k] = sk.beta[c] - sk.mean[c] * scq[k];
        }
    }
    float acc[16][4];
#pragma unroll
    for (int d = 0; d < 16; ++d)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[d][k] = 0.f;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        for (int e = t; e < TL_H * TL_W; e += 256) {
            const int hy = e / TL_W, hx = e - hy * TL_W;
            const int gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
            D[e] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? dout[((long)n * H + gy) * W + gx] : 0.f;
        }
        __syncthreads();
        constexpr int UN = 4;
        for (int e0 = 0; e0 < ET_H * ET_W; e0 += PPI * UN) {
            float4 x4[UN];
            int py[UN], px[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * PPI + slot;
                py[u] = e / ET_W;
                px[u] = e - py[u] * ET_W;
                const bool ok = y0 + py[u] < Hc && x0 + px[u] < Wc;
                x4[u] = ok ? *reinterpret_cast<const float4*>(src + (((long)n * Hc + y0 + py[u]) * Wc + x0 + px[u]) * C + q * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
                if (lazy) {
                    float* xe = reinterpret_cast<float*>(&x4[u]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float y = fmaf(xe[k], scq[k], shq[k]);
                        xe[k] = ok ? (y > 0.f ? y : y * slope) : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float xv[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
                for (int d = 0; d < 16; ++d) {
                    const float dv = D[(2 * py[u] + d / 4) * TL_W + 2 * px[u] + d % 4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[d][k] = fmaf(xv[k], dv, acc[d][k]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 16; ++d)
        *reinterpret_cast<float4*>(red + ((slot * 16 + d) * C) + q * 4) = make_float4(acc[d][0], acc[d][1], acc[d][2], acc[d][3]);
    __syncthreads();
    double* out = partial + (long)blockIdx.x * (16 * C);
    for (int e = t; e < 16 * C; e += 256) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 16 * C + e];
        out[e] = sum;
    }
}

// one block per input channel: C16[ci][d] = sum_b partial[b][d * C + ci] (fixed order), then
// dWt[ci][co][ab] = sum_tap wl[co][tap] C16[ci][(a,b) - off(tap)]; c16 (nullable) receives the correlations
__global__ __launch_bounds__(256) void tail_wgrad_finish_kernel(const double* __restrict__ partial, int nb, const float* __restrict__ wl,
                                                                float* __restrict__ dwt, double* __restrict__ c16, int C, int C0) {
    __shared__ double red[256];
    __shared__ double cs[16];
    const int ci = blockIdx.x, t = threadIdx.x, d = t & 15, slice = t >> 4;
    double acc = 0.0;
    for (int b = slice; b < nb; b += 16) acc += partial[(long)b * 16 * C + d * C + ci];
    red[t] = acc;
    __syncthreads();
    if (t < 16) {
        double sum = 0.0;
        for (int sl = 0; sl < 16; ++sl) sum += red[sl * 16 + t];
        cs[t] = sum;
        if (c16) c16[(long)ci * 16 + t] = sum;
    }
    __syncthreads();
    for (int e = t; e < C0 * 4; e += 256) {
        const int co = e >> 2, ab = e & 3, a = ab >> 1, b = ab & 1;
        double sum = 0.0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = a - (tap / 3 - 1), dx = b - (tap % 3 - 1);      // d = (a, b) - off(tap), each in [-1, 2]
            sum += (double)wl[co * 9 + tap] * cs[(dy + 1) * 4 + dx + 1];
        }
        dwt[((long)ci * C0 + co) * 4 + ab] = (float)sum;
    }
}

// dw[c][tap] = sum_q s[q][c] * dout[q - off(tap)];  dbias = sum dout.  Persistent blocks over tiles; per block one partial
// row [9][C] + [1] (doubles, layout of last_wgrad_reduce_kernel).
__global__ __launch_bounds__(256) void conv_last_wgrad_tile_kernel(const float* __restrict__ s_in, const float* __restrict__ dout,
                                                                   double* __restrict__ partial, int N, int H, int W, int C,
                                                                   int CQ, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float esm[];     // D[EH_NP] (padded to 640), then the reduction scratch
    float* D = esm;
    float* red = esm + 640;                        // [PPI][9][C] floats
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    float acc[9][4];
    float accb = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[tap][k] = 0.f;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_dout_tile(D, dout, n, y0, x0, H, W, t);
        __syncthreads();
        constexpr int UN = 4;
        for (int e0 = 0; e0 < ET_H * ET_W; e0 += PPI * UN) {
            float4 s4[UN];
            int py[UN], px[UN];
            bool ok[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * PPI + slot;
                py[u] = e / ET_W;
                px[u] = e - py[u] * ET_W;
                ok[u] = y0 + py[u] < H && x0 + px[u] < W;
                s4[u] = ok[u] ? *reinterpret_cast<const float4*>(s_in + (((long)n * H + y0 + py[u]) * W + x0 + px[u]) * C + q * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float sv[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float d = D[(py[u] + 2 - tap / 3) * EH_W + px[u] + 2 - tap % 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[tap][k] = fmaf(sv[k], d, acc[tap][k]);
                    if (tap == 4 && q == 0 && ok[u]) accb += d;
                }
            }
        }
    }
    // one block-level reduction over the PPI pixel slots (fixed order)
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        *reinterpret_cast<float4*>(red + ((slot * 9 + tap) * C) + q * 4) = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
    __syncthreads();
    double* out = partial + (long)blockIdx.x * (9 * C + 1);
    for (int e = t; e < 9 * C; e += 256) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 9 * C + e];
        out[e] = sum;
    }
    __syncthreads();
    if (q == 0) red[slot] = accb;
    __syncthreads();
    if (t == 0) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl];
        out[9 * C] = sum;
    }
}

// The same when the convolution's input is not a tensor (tail of the network): the act(BN(z)) part of it is recomputed from z on
// load; per block one partial row [9][C] + [9] (doubles): the second part = S[tap] = sum_{q inside} dout[q - off(tap)], which
// multiplies the up-convolution's bias (and S[4] = sum dout is the last convolution's bias gradient).  The up-convolution part
// of the input enters through the correlations C16 (tail_wl_finish_kernel).
__global__ __launch_bounds__(256) void conv_last_wgrad_tail_kernel(TailSkip sk, const float* __restrict__ dout,
                                                                   double* __restrict__ partial, int N, int H, int W, int C,
                                                                   int CQ, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float esm[];     // D[EH_NP] (padded to 640), then the reduction scratch
    float* D = esm;
    float* red = esm + 640;                        // [PPI][9][C] floats
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    float acc[9][4];
    float accs[9];
    float sc[4], sh[4];
    const float slope = sk.slope_dev ? sk.slope_dev[0] : sk.slope;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        sc[k] = sk.invstd[c] * sk.gamma[c];
        sh[k] = sk.beta[c] - sk.mean[c] * sc[k];
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) accs[tap] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[tap][k] = 0.f;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_dout_tile(D, dout, n, y0, x0, H, W, t);
        __syncthreads();
        constexpr int UN = 4;
        for (int e0 = 0; e0 < ET_H * ET_W; e0 += PPI * UN) {
            float4 s4[UN];
            int py[UN], px[UN];
            bool ok[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * PPI + slot;
                py[u] = e / ET_W;
                px[u] = e - py[u] * ET_W;
                ok[u] = y0 + py[u] < H && x0 + px[u] < W;
                s4[u] = ok[u] ? ld_nt4(sk.z + (((long)n * H + y0 + py[u]) * W + x0 + px[u]) * C + q * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                float sv[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float y = fmaf(sv[k], sc[k], sh[k]);
                    sv[k] = ok[u] ? (y > 0.f ? y : y * slope) : 0.f;
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float d = D[(py[u] + 2 - tap / 3) * EH_W + px[u] + 2 - tap % 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[tap][k] = fmaf(sv[k], d, acc[tap][k]);
                    if (q == 0 && ok[u]) accs[tap] += d;
                }
            }
        }
    }
    // one block-level reduction over the PPI pixel slots (fixed order)
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        *reinterpret_cast<float4*>(red + ((slot * 9 + tap) * C) + q * 4) = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
    __syncthreads();
    double* out = partial + (long)blockIdx.x * (9 * C + 9);
    for (int e = t; e < 9 * C; e += 256) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 9 * C + e];
        out[e] = sum;
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) red[slot * 9 + tap] = accs[tap];
    }
    __syncthreads();
    if (t < 9) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 9 + t];
        out[9 * C + t] = sum;
    }
}

// Head of the backward in ONE pass over level 0's z (tail of the network): the partial sums of conv_last_wgrad_tail_kernel AND the
// BN-backward statistics of level 0 that conv_last_dgrad_tile_kernel<true> emits (g = conv_last^T(dout) evaluated per element, never
// stored) -- both need exactly z and the dout tile.  Persistent blocks; per block one row [9][C] + [9] of doubles (wpartial) and
// one row [4][C] of floats (bn_part, the layout bn_bwd_stats_finalize_kernel takes).
__global__ __launch_bounds__(256) void conv_last_bwd_tail_fused_kernel(TailSkip sk, const float* __restrict__ dout,
                                                                       const float* __restrict__ w, double* __restrict__ wpartial,
                                                                       float* __restrict__ bn_part, int N, int H, int W, int C, int CQ,
                                                                       int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float esm[];     // D[EH_NP] (padded to 640), then the reduction scratch
    float* D = esm;
    float* red = esm + 640;                        // [PPI][9][C] floats (>= 256 * 16)
    const int t = threadIdx.x, q = t % CQ, slot = t / CQ, PPI = 256 / CQ;
    float acc[9][4], wr[9][4], accs[9], bacc[16];
    float sc[4], sh[4], mu[4], is[4];
    const float slope = sk.slope_dev ? sk.slope_dev[0] : sk.slope;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = q * 4 + k;
        mu[k] = sk.mean[c];
        is[k] = sk.invstd[c];
        sc[k] = is[k] * sk.gamma[c];
        sh[k] = sk.beta[c] - mu[k] * sc[k];
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        accs[tap] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[tap][k] = 0.f;
            wr[tap][k] = w[(q * 4 + k) * 9 + tap];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) bacc[k] = 0.f;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_dout_tile(D, dout, n, y0, x0, H, W, t);
        __syncthreads();
        constexpr int UN = 4;
        for (int e0 = 0; e0 < ET_H * ET_W; e0 += PPI * UN) {
            float4 z4[UN];
            int py[UN], px[UN];
            bool ok[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * PPI + slot;
                py[u] = e / ET_W;
                px[u] = e - py[u] * ET_W;
                ok[u] = y0 + py[u] < H && x0 + px[u] < W;
                z4[u] = ok[u] ? ld_nt4(sk.z + (((long)n * H + y0 + py[u]) * W + x0 + px[u]) * C + q * 4)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (!ok[u]) continue;
                const float zz[4] = {z4[u].x, z4[u].y, z4[u].z, z4[u].w};
                float yv[4], av[4], g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    yv[k] = fmaf(zz[k], sc[k], sh[k]);
                    av[k] = yv[k] > 0.f ? yv[k] : yv[k] * slope;
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float d = D[(py[u] + 2 - tap / 3) * EH_W + px[u] + 2 - tap % 3];      // dout[q - off(tap)]
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc[tap][k] = fmaf(av[k], d, acc[tap][k]);
                        g[k] = fmaf(d, wr[tap][k], g[k]);
                    }
                    if (q == 0) accs[tap] += d;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gm = g[k] * (yv[k] > 0.f ? 1.f : slope);
                    const float xh = (zz[k] - mu[k]) * is[k];
                    bacc[k] += gm;
                    bacc[4 + k] = fmaf(gm, xh, bacc[4 + k]);
                    bacc[8 + k] += g[k];
                    if (!(yv[k] > 0.f)) bacc[12 + k] = fmaf(g[k], yv[k], bacc[12 + k]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        *reinterpret_cast<float4*>(red + ((slot * 9 + tap) * C) + q * 4) = make_float4(acc[tap][0], acc[tap][1], acc[tap][2], acc[tap][3]);
    __syncthreads();
    double* out = wpartial + (long)blockIdx.x * (9 * C + 9);
    for (int e = t; e < 9 * C; e += 256) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 9 * C + e];
        out[e] = sum;
    }
    __syncthreads();
    if (q == 0) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) red[slot * 9 + tap] = accs[tap];
    }
    __syncthreads();
    if (t < 9) {
        double sum = 0.0;
        for (int sl = 0; sl < PPI; ++sl) sum += (double)red[sl * 9 + t];
        out[9 * C + t] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) red[t * 16 + k] = bacc[k];
    __syncthreads();
    for (int o = t; o < 4 * C; o += 256) {
        const int sidx = o / C, c = o - sidx * C;
        float sum = 0.f;
        for (int sl = 0; sl < PPI; ++sl) sum += red[(sl * CQ + (c >> 2)) * 16 + sidx * 4 + (c & 3)];
        bn_part[((long)blockIdx.x * 4 + sidx) * C + c] = sum;
    }
}

// dw[co][tap] = sum_b partial[b][tap * C0 + co]                                              (act(BN(z)) part)
//             + sum_ci sum_ab Wt[ci][co][ab] C16[ci][(a,b) - off(tap)]                        (up-convolution part)
//             + bt[co] S[tap]                                                                 (its bias);   dbias = S[4]
// one block per output channel co, fixed orders
__global__ __launch_bounds__(256) void tail_wl_finish_kernel(const double* __restrict__ partial, int nb, const double* __restrict__ c16,
                                                             const float* __restrict__ wt, const float* __restrict__ bt,
                                                             float* __restrict__ dw, float* __restrict__ dbias, int Cin, int C0) {
    __shared__ double red[256];
    __shared__ double tot[18];
    const int co = blockIdx.x, t = threadIdx.x;
    const int row = 9 * C0 + 9;
    // columns tap * C0 + co (9) and 9 * C0 + tap (9): 18 columns x 14 row slices
    const int col = t % 18, slice = t / 18;
    double acc = 0.0;
    if (slice < 14) {
        const int e = col < 9 ? col * C0 + co : 9 * C0 + (col - 9);
        for (int b = slice; b < nb; b += 14) acc += partial[(long)b * row + e];
    }
    red[t] = acc;
    __syncthreads();
    if (t < 18) {
        double sum = 0.0;
        for (int sl = 0; sl < 14; ++sl) sum += red[sl * 18 + t];
        tot[t] = sum;
    }
    __syncthreads();
    // up-convolution part: 9 taps x 28 slices of the input channels (fixed order), combined through LDS
    {
        const int tap = t % 9, sl = t / 9;          // 252 threads
        double up = 0.0;
        if (sl < 28) {
            for (int ci = sl; ci < Cin; ci += 28)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab) {
                    const int dy = (ab >> 1) - (tap / 3 - 1), dx = (ab & 1) - (tap % 3 - 1);
                    up += (double)wt[((long)ci * C0 + co) * 4 + ab] * c16[(long)ci * 16 + (dy + 1) * 4 + dx + 1];
                }
        }
        __syncthreads();
        red[t] = up;
        __syncthreads();
    }
    if (t < 9) {
        const int tap = t;
        double up = 0.0;
        for (int sl = 0; sl < 28; ++sl) up += red[sl * 9 + tap];
        const double bias_part = bt ? (double)bt[co] * tot[9 + tap] : 0.0;
        dw[co * 9 + tap] = (float)((tot[tap] + up) + bias_part);
        if (co == 0 && tap == 4 && dbias) dbias[0] = (float)tot[9 + 4];
    }
}


// ---- first convolution (NCHW input with 1..6 channels -> NHWC, Cout = 32 / 64 / 128) ---------------------------------------
// Tile = 16 x 32 output pixels; the input halo (18 x 34 per channel plane, row pitch 36 floats = 16-byte aligned rows) sits in
// LDS.  A thread owns 4 output channels (lane cq) and works on 8-pixel row SEGMENTS: per (ky, ci) it reads the 10 input
// values the 8 pixels share with three LDS instructions (the old kernel read 27 scalars per pixel) and keeps the weights
// (forward) or the 27 x 4 weight-gradient accumulators (wgrad) in registers.  Forward: 16-byte stores, the 16 lanes of a pixel
// write its 256 contiguous bytes; BN statistics of the block leave through one LDS reduction.  Wgrad: the 8 dz loads of a
// segment are issued before its arithmetic; one three-chunk LDS reduction per block at the end (was 27 x 2 barriers).
constexpr int FP = 36;                      // LDS row pitch of the input halo, floats
constexpr int FH_PLANE = EH_H * FP;         // floats per channel plane

template <int CIN>
__device__ __forceinline__ void load_x_halo(float* X, const float* __restrict__ x, int n, int y0, int x0, int H, int W, int t) {
    for (int e = t; e < CIN * EH_NP; e += 256) {
        const int ci = e / EH_NP, rem = e - ci * EH_NP, hy = rem / EH_W, hx = rem - hy * EH_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        X[ci * FH_PLANE + hy * FP + hx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                                              ? x[(((long)n * CIN + ci) * H + gy) * W + gx] : 0.f;
    }
}

// the 10 input values under an 8-pixel segment starting at halo column c0 (multiple of 8) of halo row hy, channel ci
__device__ __forceinline__ void read_row10(const float* X, int ci, int hy, int c0, float (&v)[10]) {
    const float* p = X + ci * FH_PLANE + hy * FP + c0;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    const float2 c = *reinterpret_cast<const float2*>(p + 8);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y;
}

template <int CIN, int CQ>
__global__ __launch_bounds__(256, 4) void conv_first_fwd_seg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ z, float* __restrict__ partial, int N, int H,
                                                                 int W, int tiles_x, int tiles_y) {
    constexpr int Cout = CQ * 4, SLOTS = 256 / CQ, NSEG = (ET_H * ET_W / 8) / SLOTS;
    // halo planes, then the statistics scratch [8][256], then the weights [tap * CIN + ci][Cout] (16-byte broadcast reads:
    // with the weights in registers the kernel needed 256 VGPRs = one wave per SIMD)
    __shared__ __attribute__((aligned(16))) float X[CIN * FH_PLANE + 8 * 256 + 9 * CIN * Cout];
    float* Wl = X + CIN * FH_PLANE + 8 * 256;
    const int t = threadIdx.x, cq = t % CQ, slot = t / CQ;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    for (int e = t; e < 9 * CIN * Cout; e += 256) {
        const int j = e / Cout, co = e - j * Cout;
        Wl[e] = w[((long)co * CIN + j % CIN) * 9 + j / CIN];
    }
    load_x_halo<CIN>(X, x, n, y0, x0, H, W, t);
    __syncthreads();
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int sg = 0; sg < NSEG; ++sg) {
        const int seg = sg * SLOTS + slot;         // 64 segments: row = seg / 4, first column = (seg % 4) * 8
        const int py = seg >> 2, c0 = (seg & 3) * 8;
        float acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky)             // not unrolled: fully unrolled, hipcc hoists all 27*CIN LDS reads and spills
#pragma unroll 1
            for (int ci = 0; ci < CIN; ++ci) {
                float v[10];
                read_row10(X, ci, py + ky, c0, v);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 w4 = *reinterpret_cast<const float4*>(Wl + ((ky * 3 + kx) * CIN + ci) * Cout + cq * 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i][0] = fmaf(v[i + kx], w4.x, acc[i][0]);
                        acc[i][1] = fmaf(v[i + kx], w4.y, acc[i][1]);
                        acc[i][2] = fmaf(v[i + kx], w4.z, acc[i][2]);
                        acc[i][3] = fmaf(v[i + kx], w4.w, acc[i][3]);
                    }
                }
            }
        const int gy = y0 + py;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gx = x0 + c0 + i;
            if (gy < H && gx < W) {
                st_nt4(z + (((long)n * H + gy) * W + gx) * Cout + cq * 4, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    st_s[k] += acc[i][k];
                    st_q[k] = fmaf(acc[i][k], acc[i][k], st_q[k]);
                }
            }
        }
    }
    if (partial) {      // BN statistics of this tile: fixed-order sum over the pixel slots -> partial[tile][2][Cout]
        float* red = X + CIN * FH_PLANE;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[k * 256 + t] = st_s[k];
            red[(4 + k) * 256 + t] = st_q[k];
        }
        __syncthreads();
        for (int e = t; e < 2 * Cout; e += 256) {
            const int qq = e / Cout, c = e - qq * Cout, k = c & 3, lane_cq = c >> 2;
            float sum = 0.f;
            for (int sl = 0; sl < SLOTS; ++sl) sum += red[(qq * 4 + k) * 256 + sl * CQ + lane_cq];
            partial[(long)tl * 2 * Cout + e] = sum;
        }
    }
}

// Inference form of the above (r04): eval-mode BatchNorm + activation + MaxPool2d(2, 2) of the first block in the convolution's
// own epilogue -- a = act(fma(z, invstd*gamma, beta - mean*invstd*gamma)) is what leaves (the same expression, in the same order, as
// bn_act_pool_fwd_kernel evaluates on a stored z: bit-identical a and pooled values), z itself is never written and the separate
// pass that read its 537 MB back (cfg-G: 4.6 % of the sweep) is gone.  A thread's segments are ordered so that rows 2k and
// 2k + 1 follow each other: the 2 x 2 windows of a row pair are the thread's own eight pixels twice (first maximum in window
// order, NaN wins: torch's max_pool2d).
template <int CIN, int CQ>
__global__ __launch_bounds__(256, 2) void conv_first_fwd_act_seg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float slope_val, const float* __restrict__ slope_dev,
                                                                     float* __restrict__ a, float* __restrict__ pooled, int N, int H,
                                                                     int W, int tiles_x, int tiles_y, unsigned* p_amax, int amax_img_stride) {
    constexpr int Cout = CQ * 4, SLOTS = 256 / CQ, NSEG = (ET_H * ET_W / 8) / SLOTS;
    static_assert(NSEG % 2 == 0 && SLOTS % 4 == 0, "row pairs per thread");
    float pmx = 0.f;        // p_amax (nullable): magnitude slot of `pooled`, the operand of the next level's three-product GEMM
    __shared__ __attribute__((aligned(16))) float X[CIN * FH_PLANE + 9 * CIN * Cout];
    float* Wl = X + CIN * FH_PLANE;
    const int t = threadIdx.x, cq = t % CQ, slot = t / CQ;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    for (int e = t; e < 9 * CIN * Cout; e += 256) {
        const int j = e / Cout, co = e - j * Cout;
        Wl[e] = w[((long)co * CIN + j % CIN) * 9 + j / CIN];
    }
    load_x_halo<CIN>(X, x, n, y0, x0, H, W, t);
    float sc[4], sh[4];
    {
        const float4 m4 = *reinterpret_cast<const float4*>(mean + cq * 4), i4 = *reinterpret_cast<const float4*>(invstd + cq * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(gamma + cq * 4), b4 = *reinterpret_cast<const float4*>(beta + cq * 4);
        sc[0] = i4.x * g4.x; sc[1] = i4.y * g4.y; sc[2] = i4.z * g4.z; sc[3] = i4.w * g4.w;
        sh[0] = b4.x - m4.x * sc[0]; sh[1] = b4.y - m4.y * sc[1]; sh[2] = b4.z - m4.z * sc[2]; sh[3] = b4.w - m4.w * sc[3];
    }
    const float slope = slope_dev ? slope_dev[0] : slope_val;
    __syncthreads();
    const int W2 = W >> 1, H2 = H >> 1;
    float prev[8][4];
#pragma unroll 1
    for (int sg = 0; sg < NSEG; ++sg) {
        // row pair (sg >> 1) * (SLOTS / 4) + (slot >> 2), row sg & 1 of it; first column (slot & 3) * 8
        const int py = 2 * ((sg >> 1) * (SLOTS / 4) + (slot >> 2)) + (sg & 1), c0 = (slot & 3) * 8;
        float acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll 1
            for (int ci = 0; ci < CIN; ++ci) {
                float v[10];
                read_row10(X, ci, py + ky, c0, v);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 w4 = *reinterpret_cast<const float4*>(Wl + ((ky * 3 + kx) * CIN + ci) * Cout + cq * 4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i][0] = fmaf(v[i + kx], w4.x, acc[i][0]);
                        acc[i][1] = fmaf(v[i + kx], w4.y, acc[i][1]);
                        acc[i][2] = fmaf(v[i + kx], w4.z, acc[i][2]);
                        acc[i][3] = fmaf(v[i + kx], w4.w, acc[i][3]);
                    }
                }
            }
        const int gy = y0 + py;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y = fmaf(acc[i][k], sc[k], sh[k]);
                acc[i][k] = y > 0.f ? y : y * slope;
            }
            const int gx = x0 + c0 + i;
            if (gy < H && gx < W)
                st_nt4(a + (((long)n * H + gy) * W + gx) * Cout + cq * 4, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));   // next read: the tail
        }
        if ((sg & 1) == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) prev[i][k] = acc[i][k];
        } else if (pooled) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    m[k] = prev[2 * j][k];
                    const float c1 = prev[2 * j + 1][k], c2 = acc[2 * j][k], c3 = acc[2 * j + 1][k];
                    if (c1 > m[k] || c1 != c1) m[k] = c1;
                    if (c2 > m[k] || c2 != c2) m[k] = c2;
                    if (c3 > m[k] || c3 != c3) m[k] = c3;
                }
                const int gx = x0 + c0 + 2 * j;
                if (gy < H && gx < W) {
                    *reinterpret_cast<float4*>(pooled + (((long)n * H2 + (gy >> 1)) * W2 + (gx >> 1)) * Cout + cq * 4) =
                        make_float4(m[0], m[1], m[2], m[3]);
                    pmx = amax_acc(amax_acc(amax_acc(amax_acc(pmx, m[0]), m[1]), m[2]), m[3]);
                }
            }
        }
    }
    if (p_amax) amax_commit(p_amax + n * amax_img_stride, pmx);       // (per-image slots: stride > 0, rd_quant_next_img)
}

// FUSED: there is no dz tensor.  The first block's BN + activation + pool backward (bn_act_bwd_kernel<true, true> in
// rd_elementwise.hip, the same expressions in the same order) is evaluated on the operands as they are loaded: dz of level 0 has
// this kernel as its only reader, so its 537 MB (cfg-S) are neither written nor read back.
template <int CIN, int CQ, bool FUSED>
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_seg_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                   float* __restrict__ partial, int N, int H, int W, int tiles_x,
                                                                   int tiles_y, int ntiles, FirstBnBwd bn) {
    constexpr int Cout = CQ * 4, SLOTS = 256 / CQ, NSEG = (ET_H * ET_W / 8) / SLOTS, NT = 9 * CIN;
    extern __shared__ __attribute__((aligned(16))) float fsm[];       // halo planes, then the reduction scratch [9][256][4]
    float* X = fsm;
    float* red = fsm + CIN * FH_PLANE;
    float* Dl = red + 9 * 256 * 4;            // FUSED with bn.dout: dout tile + halo [EH_NP] (padded to 640), then w_last [9][Cout]
    float* Wl = Dl + 640;
    const int t = threadIdx.x, cq = t % CQ, slot = t / CQ;
    const bool lazy_g = FUSED && bn.dout != nullptr;
    if (lazy_g)
        for (int e = t; e < 9 * Cout; e += 256) Wl[e] = bn.w_last[(e % Cout) * 9 + e / Cout];
    float wg[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) wg[j][k] = 0.f;
    float sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0}, mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0}, k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
    float slope = 0.f;
    if (FUSED) {
        slope = bn.slope_dev ? bn.slope_dev[0] : bn.slope;
        const float4 m4 = *reinterpret_cast<const float4*>(bn.mean + cq * 4), i4 = *reinterpret_cast<const float4*>(bn.invstd + cq * 4);
        const float4 g4 = *reinterpret_cast<const float4*>(bn.gamma + cq * 4), b4 = *reinterpret_cast<const float4*>(bn.beta + cq * 4);
        mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
        is[0] = i4.x; is[1] = i4.y; is[2] = i4.z; is[3] = i4.w;
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = is[q] * gg[q];
            sh[q] = bb[q] - mu[q] * sc[q];
            if (bn.training) {
                k1[q] = (float)(bn.sums[cq * 4 + q] / bn.count);
                k2[q] = (float)(bn.sums[Cout + cq * 4 + q] / bn.count);
            }
        }
    }
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_x_halo<CIN>(X, x, n, y0, x0, H, W, t);
        if (lazy_g) load_dout_tile(Dl, bn.dout, n, y0, x0, H, W, t);
        __syncthreads();
#pragma unroll 1
        for (int sg = 0; sg < NSEG; ++sg) {
            const int seg = sg * SLOTS + slot;
            const int py = seg >> 2, c0 = (seg & 3) * 8;
            const int gy = y0 + py;
            float4 d[8];
            if (!FUSED) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {          // eight 16-byte loads of dz in flight
                    const int gx = x0 + c0 + i;
                    d[i] = (gy < H && gx < W) ? *reinterpret_cast<const float4*>(dz + (((long)n * H + gy) * W + gx) * Cout + cq * 4)
                                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
                // H, W even (pooled level): the segment's eight pixels are four window halves of pooled row gy / 2
                const bool rowok = gy < H;
                float4 gpv[4];
                uchar4 piv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = x0 + c0 + 2 * j;
                    const bool ok = rowok && gx < W && bn.g_pool;
                    const long pp = (((long)n * (H >> 1) + (gy >> 1)) * (W >> 1) + (gx >> 1)) * Cout + cq * 4;
                    gpv[j] = ok ? *reinterpret_cast<const float4*>(bn.g_pool + pp) : make_float4(0.f, 0.f, 0.f, 0.f);
                    piv[j] = ok ? *reinterpret_cast<const uchar4*>(bn.idx + pp) : make_uchar4(255, 255, 255, 255);
                }
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf) {    // four pixels at a time: z and the skip gradient in flight together
                    float4 zv[4], gv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = hlf * 4 + u, gx = x0 + c0 + i;
                        const bool ok = rowok && gx < W;
                        const long o = (((long)n * H + gy) * W + gx) * Cout + cq * 4;
                        zv[u] = ok ? *reinterpret_cast<const float4*>(bn.z + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                        gv[u] = (ok && bn.g_full) ? *reinterpret_cast<const float4*>(bn.g_full + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (lazy_g) {                      // g[q][c] = sum_tap dout[q - off(tap)] w_last[c][tap] (conv_last_dgrad_tile_kernel)
                        float ga[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int k = 0; k < 4; ++k) ga[u][k] = 0.f;
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) {
                            const float4 w4 = *reinterpret_cast<const float4*>(Wl + tap * Cout + cq * 4);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float dv = Dl[(py + 2 - tap / 3) * EH_W + c0 + hlf * 4 + u + 2 - tap % 3];
                                ga[u][0] = fmaf(dv, w4.x, ga[u][0]);
                                ga[u][1] = fmaf(dv, w4.y, ga[u][1]);
                                ga[u][2] = fmaf(dv, w4.z, ga[u][2]);
                                ga[u][3] = fmaf(dv, w4.w, ga[u][3]);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) gv[u] = make_float4(ga[u][0], ga[u][1], ga[u][2], ga[u][3]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = hlf * 4 + u, gx = x0 + c0 + i;
                        const int k = ((gy & 1) << 1) | (i & 1);                  // position of this pixel in its 2x2 window
                        const float4 gp4 = gpv[i >> 1];
                        const uchar4 pi4 = piv[i >> 1];
                        const float xz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w}, gf[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
                        const float gpq[4] = {gp4.x, gp4.y, gp4.z, gp4.w};
                        const int piq[4] = {pi4.x, pi4.y, pi4.z, pi4.w};
                        float o4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float y = fmaf(xz[q], sc[q], sh[q]);
                            float g = gf[q];
                            if (piq[q] == k) g += gpq[q];
                            const float gm = g * (y > 0.f ? 1.f : slope);
                            const float xh = (xz[q] - mu[q]) * is[q];
                            o4[q] = bn.training ? sc[q] * (gm - k1[q] - xh * k2[q]) : sc[q] * gm;
                        }
                        d[i] = (rowok && gx < W) ? make_float4(o4[0], o4[1], o4[2], o4[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    float v[10];
                    read_row10(X, ci, py + ky, c0, v);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int j = (ky * 3 + kx) * CIN + ci;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            wg[j][0] = fmaf(v[i + kx], d[i].x, wg[j][0]);
                            wg[j][1] = fmaf(v[i + kx], d[i].y, wg[j][1]);
                            wg[j][2] = fmaf(v[i + kx], d[i].z, wg[j][2]);
                            wg[j][3] = fmaf(v[i + kx], d[i].w, wg[j][3]);
                        }
                    }
                }
        }
    }
    // block reduction over the pixel slots in chunks of 9 (tap, ci) rows: partial[block][NT][Cout]
    float* out = partial + (long)blockIdx.x * NT * Cout;
#pragma unroll
    for (int ch = 0; ch < CIN; ++ch) {
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 9; ++jj)
            *reinterpret_cast<float4*>(red + (jj * 256 + t) * 4) = make_float4(wg[ch * 9 + jj][0], wg[ch * 9 + jj][1], wg[ch * 9 + jj][2], wg[ch * 9 + jj][3]);
        __syncthreads();
        for (int e = t; e < 9 * Cout; e += 256) {
            const int jj = e / Cout, c = e - jj * Cout;
            float sum = 0.f;
            for (int sl = 0; sl < SLOTS; ++sl) sum += red[(jj * 256 + sl * CQ + (c >> 2)) * 4 + (c & 3)];
            out[(long)(ch * 9 + jj) * Cout + c] = sum;
        }
    }
}

// ---- first convolution on the matrix pipe, in exact fp32 (r05) ---------------------------------------------------------------
// The segment kernels above are bound by vector-ALU issue: 27 fused multiply-adds per output element (CIN = 3), 0.91-0.97 of
// the cycles of conv_first_fwd_seg / conv_first_fwd_act_seg are VALU instructions (profiles/r05_notes.md section 6), and the
// packed form that would halve them is banned in this library.  `v_mfma_f32_32x32x2_f32` does the same arithmetic -- an fp32
// fused multiply-add per k, accumulated in k order, bit for bit the chain `acc = fmaf(x, w, acc)` -- at twice the vector rate
// and off the vector ALU.  As a GEMM: rows = the 32 pixels of one image row of the 16 x 32 tile, columns = output channels,
// k = (ky, ci, kx) in the order the segment kernels add their products, so z is THE SAME BITS (asserted by
// tests/test_ops_gpu.py::test_first_convolution_on_the_matrix_pipe_is_bit_identical).
//   A: lane (x = lane % 32, khalf = lane / 32) reads X[ci][py + ky][x + kx] for k = 2 s + khalf -- one conflict-free ds_read_b32;
//   B: w[co = lane % 32 + 32 j][k] sits in registers (2 s + khalf: NS values per 32-channel block, loaded once per block);
//   C: lane = one output channel, sixteen pixels x = (r & 3) + 8 (r >> 2) + 4 khalf of the row: a store instruction writes the 128
//      contiguous bytes of 32 channels of one pixel (lanes 0-31) and of the pixel four columns on (lanes 32-63); the BatchNorm
//      statistics are sums over the lane's own registers; the 2 x 2 max-pool windows of the inference form are registers
//      (r, r + 1) of the two rows of a row pair, which one wave owns.
// A wave takes four image rows as two row pairs (4 x NB accumulators live at a time).
template <int CIN, int NB, int MODE>      // NB = Cout / 32; MODE 0: z (+ per-tile BN statistics), 1: a = act(BN(z)) (+ 2 x 2 max-pool)
__global__ __launch_bounds__(256, 2) void conv_first_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  float* __restrict__ out, float* __restrict__ aux,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float slope_val, const float* __restrict__ slope_dev, int N, int H,
                                                                  int W, int tiles_x, int tiles_y) {
    constexpr int Cout = NB * 32, K = 9 * CIN, NS = (K + 1) / 2;
    __shared__ __attribute__((aligned(16))) float X[CIN * FH_PLANE + 2 * NS * Cout + 4 * 2 * Cout];
    float* Wk = X + CIN * FH_PLANE;               // [k][co], k = (ky * CIN + ci) * 3 + kx, zero beyond K
    float* red = Wk + 2 * NS * Cout;              // MODE 0: per-wave statistics [4][2][Cout]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, xl = lane & 31, khalf = lane >> 5;
    const int tl = blockIdx.x;
    const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
    const int y0 = ty * ET_H, x0 = tx * ET_W;
    for (int e = t; e < 2 * NS * Cout; e += 256) {
        const int k = e / Cout, co = e - k * Cout;
        const int ky = k / (3 * CIN), ci = (k / 3) % CIN, kx = k % 3;
        Wk[e] = k < K ? w[((long)co * CIN + ci) * 9 + ky * 3 + kx] : 0.f;
    }
    load_x_halo<CIN>(X, x, n, y0, x0, H, W, t);
    __syncthreads();
    float wb[NS][NB];
    int aoff[NS];                                  // LDS word offset of this lane's A value of step s, relative to (row py, column xl)
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        const int k = 2 * s_ + khalf;
#pragma unroll
        for (int j = 0; j < NB; ++j) wb[s_][j] = Wk[k * Cout + j * 32 + xl];
        const int kk = k < K ? k : 0;              // the padding k multiplies a zero weight: any finite A value will do
        const int ky = kk / (3 * CIN), ci = (kk / 3) % CIN, kx = kk % 3;
        aoff[s_] = ci * FH_PLANE + ky * FP + kx;
    }
    float sc[NB], sh[NB], slope = 0.f;
    float st_s[NB], st_q[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        st_s[j] = st_q[j] = 0.f;
        sc[j] = sh[j] = 0.f;
        if (MODE == 1) {
            const int co = j * 32 + xl;
            sc[j] = invstd[co] * gamma[co];
            sh[j] = beta[co] - mean[co] * sc[j];
        }
    }
    if (MODE == 1) slope = slope_dev ? slope_dev[0] : slope_val;
    const int H2 = H >> 1, W2 = W >> 1;
#pragma unroll 1
    for (int rp = 0; rp < 2; ++rp) {
        const int py = wave * 4 + rp * 2;          // rows py, py + 1 of the tile
        f32x16 acc[2][NB];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;
        const float* xr = X + py * FP + xl;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const float a0 = xr[aoff[s_]], a1 = xr[aoff[s_] + FP];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wb[s_][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wb[s_][j], acc[1][j], 0, 0, 0);
            }
        }
        // ---- epilogue of the row pair: buffer stores, the row / pixel part of the address in a scalar register, one per-lane offset
        // (masked lanes carry an out-of-extent offset: no exec-mask branch per store)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gy = y0 + py + q;
            const bool rowok = gy < H;
            const long rbase = (((long)n * H + (rowok ? gy : 0)) * W + x0) * Cout;
            const int cols_left = W - x0;                  // pixels of this tile row inside the image
            const __amdgpu_buffer_rsrc_t rsO = make_rsrc(out + rbase, (unsigned)((cols_left < ET_W ? cols_left : ET_W) * Cout * 4));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned lane_off = rowok ? (unsigned)((4 * khalf * Cout + j * 32 + xl) * 4) : kOOB;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = (r & 3) + 8 * (r >> 2);          // + 4 khalf in the lane offset
                    float v = acc[q][j][r];
                    if (MODE == 1) {
                        const float y = fmaf(v, sc[j], sh[j]);
                        v = y > 0.f ? y : y * slope;
                        acc[q][j][r] = v;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rsO, lane_off, px * Cout * 4, 0);
                    if (MODE == 0) {
                        const bool in = rowok && px + 4 * khalf < cols_left;
                        const float vm = in ? v : 0.f;
                        st_s[j] += vm;
                        st_q[j] = fmaf(vm, vm, st_q[j]);
                    }
                }
            }
        }
        if (MODE == 1 && aux) {                    // 2 x 2 max-pool of the row pair: first maximum in window order, NaN wins
            const int gy = y0 + py;
            const bool rowok = gy < H;
            const long pbase = (((long)n * H2 + ((rowok ? gy : 0) >> 1)) * W2 + (x0 >> 1)) * Cout;
            const int pcols = (W - x0) >> 1;
            const __amdgpu_buffer_rsrc_t rsP = make_rsrc(aux + pbase, (unsigned)((pcols < ET_W / 2 ? pcols : ET_W / 2) * Cout * 4));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned lane_off = rowok ? (unsigned)((2 * khalf * Cout + j * 32 + xl) * 4) : kOOB;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int px = (r & 3) + 8 * (r >> 2);
                    float m = acc[0][j][r];
                    const float c1 = acc[0][j][r + 1], c2 = acc[1][j][r], c3 = acc[1][j][r + 1];
                    if (c1 > m || c1 != c1) m = c1;
                    if (c2 > m || c2 != c2) m = c2;
                    if (c3 > m || c3 != c3) m = c3;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(m), rsP, lane_off, (px >> 1) * Cout * 4, 0);
                }
            }
        }
    }
    if (MODE == 0 && aux) {                        // BN statistics of the tile -> aux[tile][2][Cout], fixed order: halves, then waves
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float s2 = st_s[j] + __shfl_xor(st_s[j], 32), q2 = st_q[j] + __shfl_xor(st_q[j], 32);
            if (khalf == 0) {
                red[(wave * 2 + 0) * Cout + j * 32 + xl] = s2;
                red[(wave * 2 + 1) * Cout + j * 32 + xl] = q2;
            }
        }
        __syncthreads();
        for (int e = t; e < 2 * Cout; e += 256) {
            const int qq = e / Cout, c = e - qq * Cout;
            aux[(long)tl * 2 * Cout + e] = ((red[(0 * 2 + qq) * Cout + c] + red[(1 * 2 + qq) * Cout + c]) + red[(2 * 2 + qq) * Cout + c]) +
                                           red[(3 * 2 + qq) * Cout + c];
        }
    }
}

template <int CIN, int MODE>
static void launch_first_mfma(const float* x, const float* wt, float* out, float* aux, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float slope, const float* slope_dev, int n, int h, int w,
                              int cout, hipStream_t s) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H), nt = n * tx * ty;
    if (cout == 64) RD_LAUNCH((conv_first_fwd_mfma_kernel<CIN, 2, MODE>), dim3(nt), dim3(256), 0, s, x, wt, out, aux, mean, invstd, gamma, beta, slope, slope_dev, n, h, w, tx, ty);
    else RD_LAUNCH((conv_first_fwd_mfma_kernel<CIN, 1, MODE>), dim3(nt), dim3(256), 0, s, x, wt, out, aux, mean, invstd, gamma, beta, slope, slope_dev, n, h, w, tx, ty);
}
// (Cout = 128 would need 8 accumulators + 4 x NS weights per lane: it stays on the segment kernels)
// OPT-IN (edge_conv >= 0 with bit 64 set), not part of "-1 = everything": measured SLOWER than the segment kernels although it takes
// the 27 multiply-adds per element off the vector ALU (profiles/r05_notes.md section 11: cfg-S level 0 alone 192 vs 175 us; with the
// stores removed 113 us -- of which the matrix pipe is busy 47 us --, with the MFMAs removed 144 us: one tile per block leaves the halo
// load, the 3 us of MFMAs and 128 four-byte stores per lane of a tile in sequence, and the old kernel's 16-byte stores reach the HBM
// write rate the 4-byte ones of the C layout do not).  Kept because it pins a fact the design leans on elsewhere: the fp32 MFMA is
// the SAME BITS as the fmaf chain (tests/test_ops_gpu.py::test_first_convolution_on_the_matrix_pipe_is_bit_identical).
static bool first_mfma_ok(int cout) {
    const int v = tune(TUNE_EDGE_CONV);
    return (cout == 32 || cout == 64) && v >= 0 && (v & 64);
}

// ---- the first convolution's fused weight gradient on the matrix pipe, in exact fp32 (r05) -----------------------------------------
// conv_first_wgrad_seg_kernel<.., true> is bound by vector-ALU issue (0.81 of its cycles; profiles/r05_notes.md section 6): 54 vector
// instructions per dz element, 27 of them the products dW[co][k] += dz[p][co] * xpatch[p][k].  Here those products are a GEMM on
// `v_mfma_f32_32x32x2_f32` with K = pixels:
//   A[co][p]: the lane (co = lane % 32 + 32 j, khalf = lane / 32) EVALUATES dz of pixel (row, 2 xs + khalf) for its channel -- the
//             BatchNorm / activation / un-pool / skip-add backward of the segment kernel, expression for expression -- from 4-byte
//             loads (lanes 0-31 = 128 contiguous bytes of one pixel); the operand never exists anywhere but in that register;
//   B[p][k]:  lane (k = lane % 32 -> (ky, kx, ci), khalf) reads X[ci][row + ky][2 xs + khalf + kx] from the LDS halo: one
//             conflict-free ds_read_b32 (bank = 8 ci + 4 ky + kx + khalf);
//   C[co][k]: NB accumulators of 16 registers instead of 27 x 4 per lane.
// Per step (2 pixels x 32 NB channels): 2 NB x ~24 vector instructions + 9 shared LDS reads of the dout tile (composed tail) against
// NB MFMAs of 64 cycles -- measured 67.3 M vector instructions per launch against the segment kernel's 123.7 M.  A wave walks four image rows of the 16 x 32 tile;
// the four waves' accumulators are added in wave order through LDS at the end of the block's tile loop: partial[block][k][Cout], the
// segment kernel's layout, reduced over blocks by first_wgrad_reduce_kernel as before.  Deterministic; not bit-identical to the segment
// kernel (another summation order), same fp32 products.
template <int CIN, int NB, bool GF>      // GF: the full-resolution gradient operand is a tensor (else: absent, or evaluated from dout)
__global__ __launch_bounds__(256, 4) void conv_first_wgrad_mfma_kernel(const float* __restrict__ x, float* __restrict__ partial, int N,
                                                                    int H, int W, int tiles_x, int tiles_y, int ntiles, FirstBnBwd bn,
                                                                    unsigned z_bytes, unsigned p_bytes, unsigned i_bytes) {
    constexpr int Cout = NB * 32, NT = 9 * CIN, GS = NB == 2 ? 2 : 4;       // GS steps (pixel pairs) x two rows per batch of loads
    static_assert(NT <= 32, "one 32-column block of (tap, ci)");
    extern __shared__ __attribute__((aligned(16))) float msm[];
    float* X = msm;                               // halo planes
    float* Dl = X + CIN * FH_PLANE;               // dout tile + halo (640)
    float* red = msm;                             // after the tile loop: [4 waves][NB][16][64], over the halo planes
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, xl = lane & 31, khalf = lane >> 5;
    const bool lazy_g = bn.dout != nullptr;
    float sc[NB], sh[NB], mu[NB], is[NB], k1[NB], k2[NB], wl[NB][9];
    const float slope = bn.slope_dev ? bn.slope_dev[0] : bn.slope;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int co = j * 32 + xl;
        mu[j] = bn.mean[co];
        is[j] = bn.invstd[co];
        sc[j] = is[j] * bn.gamma[co];
        sh[j] = bn.beta[co] - mu[j] * sc[j];
        k1[j] = k2[j] = 0.f;
        if (bn.training) {
            k1[j] = (float)(bn.sums[co] / bn.count);
            k2[j] = (float)(bn.sums[Cout + co] / bn.count);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wl[j][tap] = lazy_g ? bn.w_last[co * 9 + tap] : 0.f;
    }
    // B operand: column k = xl -> row j_ = (ky * 3 + kx) * CIN + ci of the partial layout; columns >= NT are never written out
    int boff;
    {
        const int j_ = xl < NT ? xl : 0, tap = j_ / CIN, ci = j_ - tap * CIN;
        boff = ci * FH_PLANE + (tap / 3) * FP + tap % 3 + khalf;
    }
    const __amdgpu_buffer_rsrc_t rsZ = make_rsrc(bn.z, z_bytes), rsG = make_rsrc(bn.g_full ? bn.g_full : bn.z, z_bytes);
    const __amdgpu_buffer_rsrc_t rsP = make_rsrc(bn.g_pool ? bn.g_pool : bn.z, p_bytes), rsI = make_rsrc(bn.idx ? (const void*)bn.idx : (const void*)bn.z, i_bytes);
    const bool has_gp = bn.g_pool != nullptr;
    const unsigned zlane = (unsigned)((khalf * Cout + xl) * 4), plane = (unsigned)(xl * 4), ilane = (unsigned)xl;
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_x_halo<CIN>(X, x, n, y0, x0, H, W, t);
        if (lazy_g) load_dout_tile(Dl, bn.dout, n, y0, x0, H, W, t);
        __syncthreads();
        // a wave takes image rows 4 wave .. 4 wave + 3 as two ROW PAIRS: the four pixels of a 2 x 2 pooling window (two steps of two
        // rows) share one load of the pooled gradient and of the arg-max byte
#pragma unroll 1
        for (int rp = 0; rp < 2; ++rp) {
            const int py = wave * 4 + rp * 2, gy = y0 + py;                                  // gy even
            const bool rok0 = gy < H, rok1 = gy + 1 < H;
            // byte offsets of the tile rows (scalar); < 4 GB checked by the launcher
            const unsigned zrow = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)n * H + gy) * W + x0) * Cout * 4));
            const unsigned zrow1 = zrow + (unsigned)(W * Cout * 4);
            const unsigned prow = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)n * (H >> 1) + (gy >> 1)) * (W >> 1) + (x0 >> 1)) * Cout));
            const float* xb = X + py * FP + boff;
            const float* db = Dl + (py + 2) * EH_W + khalf + 2;
#pragma unroll 1
            for (int x4 = 0; x4 < 16; x4 += GS) {
                float zv[2][GS][NB], gf[2][GS][GF ? NB : 1], gp[GS][NB];
                int pi[GS][NB];
                bool ok[2][GS];
                const unsigned zs = (unsigned)(2 * x4 * Cout * 4), ps = (unsigned)(x4 * Cout);
#pragma unroll
                for (int u = 0; u < GS; ++u) {         // the loads of GS steps x two rows go out together; masked lanes carry an
                    const int px = 2 * (x4 + u) + khalf;                    // out-of-extent offset and read zeros (no branch, no wait)
                    const bool cok = x0 + px < W;
                    ok[0][u] = rok0 && cok;
                    ok[1][u] = rok1 && cok;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const unsigned zi = (unsigned)((2 * u * Cout + j * 32) * 4), pq = (unsigned)(u * Cout + j * 32);
                        const unsigned v0 = ok[0][u] ? zlane + zi : kOOB, v1 = ok[1][u] ? zlane + zi : kOOB;
                        zv[0][u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, v0, zrow + zs, RD_AUX_NT));
                        zv[1][u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, v1, zrow1 + zs, RD_AUX_NT));
                        if (GF) {
                            gf[0][u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsG, v0, zrow + zs, RD_AUX_NT));
                            gf[1][u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsG, v1, zrow1 + zs, RD_AUX_NT));
                        }
                        const bool pok = ok[0][u] && has_gp;                  // H even: row gy + 1 exists whenever gy does
                        gp[u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsP, pok ? plane + pq * 4 : kOOB, (prow + ps) * 4, RD_AUX_NT));
                        pi[u][j] = (int)__builtin_amdgcn_raw_buffer_load_b8(rsI, pok ? ilane + pq : kOOB, prow + ps, RD_AUX_NT);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int kwin = q << 1 | khalf;                                          // this pixel's place in its 2 x 2 window
#pragma unroll
                    for (int u = 0; u < GS; ++u) {
                        const int pxe = 2 * (x4 + u);                                        // even pixel of the pair
                        const float b = xb[q * FP + pxe];
                        float dv[9];
                        if (lazy_g) {
#pragma unroll
                            for (int tap = 0; tap < 9; ++tap) dv[tap] = db[q * EH_W + pxe - (tap / 3) * EH_W - tap % 3];
                        }
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            float g = GF ? gf[q][u][GF ? j : 0] : 0.f;
                            if (lazy_g) {              // g[q][c] = sum_tap dout[q - off(tap)] w_last[c][tap], taps in order from zero
                                float ga = 0.f;
#pragma unroll
                                for (int tap = 0; tap < 9; ++tap) ga = fmaf(dv[tap], wl[j][tap], ga);
                                g = ga;
                            }
                            const float xz = zv[q][u][j];
                            const float y = fmaf(xz, sc[j], sh[j]);
                            if (pi[u][j] == kwin) g += gp[u][j];
                            const float gm = g * (y > 0.f ? 1.f : slope);
                            const float xh = (xz - mu[j]) * is[j];
                            const float o = bn.training ? sc[j] * (gm - k1[j] - xh * k2[j]) : sc[j] * gm;
                            const float dzv = ok[q][u] ? o : 0.f;
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dzv, b, acc[j], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- the four waves' accumulators, added in wave order; rows = co (C layout), columns = k -> partial[block][k][Cout]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NB + j) * 16 + r) * 64 + lane] = acc[j][r];
    __syncthreads();
    float* out = partial + (long)blockIdx.x * NT * Cout;
    for (int e = t; e < NB * 16 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, j = e >> 10;
        const int k = l & 31, co = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (k < NT) {
            const float v = ((red[((0 * NB + j) * 16 + r) * 64 + l] + red[((1 * NB + j) * 16 + r) * 64 + l]) +
                             red[((2 * NB + j) * 16 + r) * 64 + l]) + red[((3 * NB + j) * 16 + r) * 64 + l];
            out[(long)k * Cout + co] = v;
        }
    }
}

// Default since r05 (cfg-S level 0: 0.272 -> 0.217 ms in the step, +0.6 % end to end, profiles/r05_notes.md section 12); edge_conv
// without bit 128 (e.g. 63) selects the segment kernel.
static bool first_wgrad_mfma_ok(int cin, int cout, int n, int h, int w) {
    return cin <= 3 && (cout == 32 || cout == 64) && edge_on(128) && 4.0 * n * h * (double)w * cout < 4294967040.0;
}

// (5 and 6 input channels would need > 256 registers for the 216 weight-gradient accumulators: they stay on the generic kernel)
static bool first_shape_ok(int cin, int cout) { return cin >= 1 && cin <= 4 && (cout == 32 || cout == 64 || cout == 128); }

int conv_first_seg_tiles(int n, int h, int w, int cin, int cout) {        // 0 = shape stays on the generic kernel
    if (!first_shape_ok(cin, cout) || !edge_on(8)) return 0;
    return n * cdiv(w, ET_W) * cdiv(h, ET_H);
}

int conv_first_wgrad_seg_blocks(int n, int h, int w, int cin, int cout) {
    if (!first_shape_ok(cin, cout) || !edge_on(16)) return 0;
    const int nt = n * cdiv(w, ET_W) * cdiv(h, ET_H);
    return nt < 1024 ? nt : 1024;
}

template <int CIN>
static int launch_first_seg(bool wgrad, const float* x, const float* wt, float* z, const float* dz, float* partial, int n, int h,
                            int w, int cout, hipStream_t s, const FirstBnBwd* bn) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H), nt = n * tx * ty;
    if (!wgrad && first_mfma_ok(cout)) {
        launch_first_mfma<CIN, 0>(x, wt, z, partial, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, n, h, w, cout, s);
    } else if (!wgrad) {
        if (cout == 64) RD_LAUNCH((conv_first_fwd_seg_kernel<CIN, 16>), dim3(nt), dim3(256), 0, s, x, wt, z, partial, n, h, w, tx, ty);
        else if (cout == 32) RD_LAUNCH((conv_first_fwd_seg_kernel<CIN, 8>), dim3(nt), dim3(256), 0, s, x, wt, z, partial, n, h, w, tx, ty);
        else RD_LAUNCH((conv_first_fwd_seg_kernel<CIN, 32>), dim3(nt), dim3(256), 0, s, x, wt, z, partial, n, h, w, tx, ty);
    } else if (bn) {
        if constexpr (CIN > 3) {      // 144 accumulators + the fused operands do not fit 256 registers
            set_error("conv_first_wgrad (fused BN backward): Cin <= 3 only");
            return RD_ERR_ARG;
        } else {
        const int nb = nt < 1024 ? nt : 1024;
        if (first_wgrad_mfma_ok(CIN, cout, n, h, w)) {
            const unsigned zb = (unsigned)(4.0 * n * h * (double)w * cout), pb = zb / 4, ib = zb / 16;
            const size_t halo = CIN * FH_PLANE + 640, redw = (size_t)4 * (cout / 32) * 16 * 64;
            const size_t sm = (halo > redw ? halo : redw) * sizeof(float);
            if (bn->g_full) {
                if (cout == 64) RD_LAUNCH((conv_first_wgrad_mfma_kernel<CIN, 2, true>), dim3(nb), dim3(256), sm, s, x, partial, n, h, w, tx, ty, nt, *bn, zb, pb, ib);
                else RD_LAUNCH((conv_first_wgrad_mfma_kernel<CIN, 1, true>), dim3(nb), dim3(256), sm, s, x, partial, n, h, w, tx, ty, nt, *bn, zb, pb, ib);
            } else {
                if (cout == 64) RD_LAUNCH((conv_first_wgrad_mfma_kernel<CIN, 2, false>), dim3(nb), dim3(256), sm, s, x, partial, n, h, w, tx, ty, nt, *bn, zb, pb, ib);
                else RD_LAUNCH((conv_first_wgrad_mfma_kernel<CIN, 1, false>), dim3(nb), dim3(256), sm, s, x, partial, n, h, w, tx, ty, nt, *bn, zb, pb, ib);
            }
            RD_LAUNCH_CHECK("conv_first_wgrad_mfma");
            return RD_OK;
        }
        const size_t smem = (size_t)(CIN * FH_PLANE + 9 * 256 * 4 + 640 + 9 * cout) * sizeof(float);
        if (cout == 64) RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 16, true>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, *bn);
        else if (cout == 32) RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 8, true>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, *bn);
        else RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 32, true>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, *bn);
        }
    } else {
        const int nb = nt < 1024 ? nt : 1024;
        const size_t smem = (size_t)(CIN * FH_PLANE + 9 * 256 * 4) * sizeof(float);
        const FirstBnBwd none = {};
        if (cout == 64) RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 16, false>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, none);
        else if (cout == 32) RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 8, false>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, none);
        else RD_LAUNCH((conv_first_wgrad_seg_kernel<CIN, 32, false>), dim3(nb), dim3(256), smem, s, x, dz, partial, n, h, w, tx, ty, nt, none);
    }
    RD_LAUNCH_CHECK("conv_first_seg");
    return RD_OK;
}

// forward (partial = per-tile BN statistics [tiles][2][Cout], nullable) / weight gradient (partial = [blocks][9*Cin][Cout])
int conv_first_seg_launch(bool wgrad, const float* x, const float* wt, float* z, const float* dz, float* partial, int n, int h,
                          int w, int cin, int cout, hipStream_t s, const FirstBnBwd* bn) {
    switch (cin) {
        case 1: return launch_first_seg<1>(wgrad, x, wt, z, dz, partial, n, h, w, cout, s, bn);
        case 2: return launch_first_seg<2>(wgrad, x, wt, z, dz, partial, n, h, w, cout, s, bn);
        case 3: return launch_first_seg<3>(wgrad, x, wt, z, dz, partial, n, h, w, cout, s, bn);
        default: return launch_first_seg<4>(wgrad, x, wt, z, dz, partial, n, h, w, cout, s, bn);
    }
}

template <int CIN>
static int launch_first_act(const float* x, const float* wt, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float slope, const float* slope_dev, float* a, float* pooled, int n, int h, int w,
                            int cout, hipStream_t s, unsigned* p_amax, int amax_img_stride) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H), nt = n * tx * ty;
    if (first_mfma_ok(cout)) {      // (opt-in A/B variant: no magnitude slot -- the next level then runs the six-product body)
        launch_first_mfma<CIN, 1>(x, wt, a, pooled, mean, invstd, gamma, beta, slope, slope_dev, n, h, w, cout, s);
        RD_LAUNCH_CHECK("conv_first_fwd_act");
        return RD_OK;
    }
    if (cout == 64) RD_LAUNCH((conv_first_fwd_act_seg_kernel<CIN, 16>), dim3(nt), dim3(256), 0, s, x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, tx, ty, p_amax, amax_img_stride);
    else if (cout == 32) RD_LAUNCH((conv_first_fwd_act_seg_kernel<CIN, 8>), dim3(nt), dim3(256), 0, s, x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, tx, ty, p_amax, amax_img_stride);
    else RD_LAUNCH((conv_first_fwd_act_seg_kernel<CIN, 32>), dim3(nt), dim3(256), 0, s, x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, tx, ty, p_amax, amax_img_stride);
    RD_LAUNCH_CHECK("conv_first_fwd_act");
    return RD_OK;
}

// inference: convolution + BN (given mean / invstd) + activation (+ 2x2 max-pool) in one kernel; shapes of conv_first_seg_tiles
int conv_first_fwd_act_launch(const float* x, const float* wt, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float slope, const float* slope_dev, float* a, float* pooled, int n, int h, int w,
                              int cin, int cout, hipStream_t s, unsigned* p_amax, int amax_img_stride) {
    switch (cin) {
        case 1: return launch_first_act<1>(x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, cout, s, p_amax, amax_img_stride);
        case 2: return launch_first_act<2>(x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, cout, s, p_amax, amax_img_stride);
        case 3: return launch_first_act<3>(x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, cout, s, p_amax, amax_img_stride);
        default: return launch_first_act<4>(x, wt, mean, invstd, gamma, beta, slope, slope_dev, a, pooled, n, h, w, cout, s, p_amax, amax_img_stride);
    }
}

static bool edge_shape_ok(int c) { return c == 16 || c == 32 || c == 64; }

int conv_last_fwd_launch(const float* s_in, const float* wt, const float* bias, const float* x_nchw, int xc, float* out, int n,
                         int h, int w, int c, hipStream_t s, int* launched) {
    *launched = 0;
    if (!edge_shape_ok(c) || !edge_on(1)) return RD_OK;
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const dim3 grid(n * tx * ty);
    if (c == 64) RD_LAUNCH(conv_last_fwd_dpp_kernel<8>, grid, dim3(256), 0, s, s_in, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    else if (c == 32) RD_LAUNCH(conv_last_fwd_dpp_kernel<4>, grid, dim3(256), 0, s, s_in, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    else RD_LAUNCH(conv_last_fwd_dpp_kernel<2>, grid, dim3(256), 0, s, s_in, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    RD_LAUNCH_CHECK("conv_last_fwd");
    *launched = 1;
    return RD_OK;
}

int conv_last_fwd_tail_launch(const TailSkip& sk, const float* t16, const float* b9, const float* wt, const float* bias,
                              const float* x_nchw, int xc, float* out, int n, int h, int w, int c, hipStream_t s) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const dim3 grid(n * tx * ty);
    if (c == 64) RD_LAUNCH(conv_last_fwd_tail_kernel<8>, grid, dim3(256), 0, s, sk, t16, b9, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    else if (c == 32) RD_LAUNCH(conv_last_fwd_tail_kernel<4>, grid, dim3(256), 0, s, sk, t16, b9, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    else RD_LAUNCH(conv_last_fwd_tail_kernel<2>, grid, dim3(256), 0, s, sk, t16, b9, wt, bias, x_nchw, xc, out, n, h, w, tx, ty);
    RD_LAUNCH_CHECK("conv_last_fwd_tail");
    return RD_OK;
}

int conv_last_wgrad_tail_launch(const TailSkip& sk, const float* dout, double* partial, int n, int h, int w, int c, hipStream_t s) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const long nt = (long)n * tx * ty;
    const int nb = (int)(nt < 1024 ? nt : 1024);
    const size_t smem = (640 + (size_t)(256 / (c / 4)) * 9 * c) * sizeof(float);
    RD_LAUNCH(conv_last_wgrad_tail_kernel, dim3(nb), dim3(256), smem, s, sk, dout, partial, n, h, w, c, c / 4, tx, ty, (int)nt);
    RD_LAUNCH_CHECK("conv_last_wgrad_tail");
    return RD_OK;
}

int conv_last_tail_blocks(int n, int h, int w) {
    const long nt = (long)n * cdiv(w, ET_W) * cdiv(h, ET_H);
    return (int)(nt < 1024 ? nt : 1024);
}

// ---- the head of the backward on the matrix pipe, in exact fp32 (r05; cf. conv_first_wgrad_mfma_kernel) -------------------------------
// conv_last_bwd_tail_fused_kernel spends 18 of its ~33 vector instructions per element on two sets of nine products: the gradient
// g = conv_last^T(dout) it evaluates per element, and the last convolution's weight gradient dw[c][tap] += act(BN(z))[p][c] dout[p - off].
// Both are small GEMMs on `v_mfma_f32_32x32x2_f32`, per 32-pixel image row of the 16 x 32 tile and 32-channel block:
//   G[p][c]   = sum_tap D[p - off(tap)] w[c][tap]   K = 9 taps (+ one zero): 5 MFMAs; lands in the C layout -- lane = channel,
//               sixteen pixels P(r) = (r & 3) + 8 (r >> 2) + 4 khalf;
//   dW[c][tap] += sum_p a[p][c] D[p - off(tap)]      K = pixels, taken in exactly that order: step r pairs pixel P(r) (lanes 0-31) with
//               P(r) + 4 (lanes 32-63), so the lane that holds g of a pixel also loads its z (128 contiguous bytes per half wave),
//               forms a = act(BN(z)) as the A operand and the BatchNorm-backward statistics of that element; B = the dout value of
//               (pixel, tap = lane % 32), one LDS read, whose masked sum over the steps IS S[tap] (the bias part).
// Per block the same rows as the vector kernel: wpartial[block][9 C + 9] doubles, bn_part[block][4][C] floats; the four waves are added
// in wave order.  Deterministic; another summation order than the vector kernel's, the same fp32 products.
template <int NB>
__global__ __launch_bounds__(256, 4) void conv_last_bwd_tail_mfma_kernel(TailSkip sk, const float* __restrict__ dout,
                                                                      const float* __restrict__ w, double* __restrict__ wpartial,
                                                                      float* __restrict__ bn_part, int N, int H, int W, int tiles_x,
                                                                      int tiles_y, int ntiles, unsigned z_bytes) {
    constexpr int C = NB * 32;
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* D = tsm;                                // dout tile + halo (640)
    float* red = tsm;                              // after the tile loop: [4][NB][16][64] accumulators, then the small sums
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, xl = lane & 31, khalf = lane >> 5;
    const float slope = sk.slope_dev ? sk.slope_dev[0] : sk.slope;
    float sc[NB], sh[NB], mu[NB], is[NB], wr5[NB][5];
    int goff[5];
#pragma unroll
    for (int s_ = 0; s_ < 5; ++s_) {
        const int k = 2 * s_ + khalf, kk = k < 9 ? k : 8;
        goff[s_] = (2 - kk / 3) * EH_W + 2 - kk % 3;
#pragma unroll
        for (int j = 0; j < NB; ++j) wr5[j][s_] = k < 9 ? w[(j * 32 + xl) * 9 + k] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = j * 32 + xl;
        mu[j] = sk.mean[c];
        is[j] = sk.invstd[c];
        sc[j] = is[j] * sk.gamma[c];
        sh[j] = sk.beta[c] - mu[j] * sc[j];
    }
    const int tapl = xl < 9 ? xl : 0;
    const int boff = (2 - tapl / 3) * EH_W + 2 - tapl % 3 + 4 * khalf;      // B operand of the weight gradient: (tap = xl, pixel half)
    const __amdgpu_buffer_rsrc_t rsZ = make_rsrc(sk.z, z_bytes);
    const unsigned zlane = (unsigned)((4 * khalf * C + xl) * 4);
    f32x16 acc[NB];
    float bacc[NB][4], accs = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) bacc[j][k] = 0.f;
    }
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int tx = tl % tiles_x, ty = (tl / tiles_x) % tiles_y, n = tl / (tiles_x * tiles_y);
        const int y0 = ty * ET_H, x0 = tx * ET_W;
        __syncthreads();
        load_dout_tile(D, dout, n, y0, x0, H, W, t);
        __syncthreads();
#pragma unroll 1
        for (int r4 = 0; r4 < 4; ++r4) {
            const int py = wave * 4 + r4, gy = y0 + py;
            const bool rowok = gy < H;
            const unsigned zrow = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)n * H + gy) * W + x0) * C * 4));
            const float* dr = D + py * EH_W;
            f32x16 gt[NB];                         // g of this image row: lane = channel, register r = pixel P(r) + 4 khalf
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) gt[j][r] = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < 5; ++s_) {
                const float a = dr[xl + goff[s_]];
#pragma unroll
                for (int j = 0; j < NB; ++j) gt[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr5[j][s_], gt[j], 0, 0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {
                float zv[8][NB];
                bool ok[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {          // masked lanes carry an out-of-extent offset and read zeros
                    const int r = rb + u, p0 = (r & 3) + 8 * (r >> 2);
                    ok[u] = rowok && x0 + p0 + 4 * khalf < W;
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        zv[u][j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsZ, ok[u] ? zlane + (unsigned)((p0 * C + j * 32) * 4) : kOOB, zrow, 0));
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = rb + u, p0 = (r & 3) + 8 * (r >> 2);
                    const float b = dr[p0 + boff];
                    accs += ok[u] ? b : 0.f;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const float zz = zv[u][j];
                        const float yv = fmaf(zz, sc[j], sh[j]);
                        const float av = yv > 0.f ? yv : yv * slope;
                        const float g = ok[u] ? gt[j][r] : 0.f;
                        const float gm = g * (yv > 0.f ? 1.f : slope);
                        const float xh = (zz - mu[j]) * is[j];
                        bacc[j][0] += gm;
                        bacc[j][1] = fmaf(gm, xh, bacc[j][1]);
                        bacc[j][2] += g;
                        if (!(yv > 0.f)) bacc[j][3] = fmaf(g, yv, bacc[j][3]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ok[u] ? av : 0.f, b, acc[j], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- block reduction, fixed order: waves 0..3 (and the two pixel halves of a lane pair for the per-channel sums)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NB + j) * 16 + r) * 64 + lane] = acc[j][r];
    __syncthreads();
    double* out = wpartial + (long)blockIdx.x * (9 * C + 9);
    for (int e = t; e < NB * 16 * 64; e += 256) {
        const int l = e & 63, r = (e >> 6) & 15, j = e >> 10;
        const int tap = l & 31, co = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (tap < 9)
            out[tap * C + co] = (((double)red[((0 * NB + j) * 16 + r) * 64 + l] + (double)red[((1 * NB + j) * 16 + r) * 64 + l]) +
                                 (double)red[((2 * NB + j) * 16 + r) * 64 + l]) + (double)red[((3 * NB + j) * 16 + r) * 64 + l];
    }
    __syncthreads();
    // per-channel statistics [4][C] and S[tap]: [wave][half][..] in LDS
    float* rs = red;                               // [4 waves][2 halves][NB * 4 * 32]
    float* ra = red + 4 * 2 * NB * 128;            // [4 waves][64 lanes]
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) rs[((wave * 2 + khalf) * NB + j) * 128 + k * 32 + xl] = bacc[j][k];
    ra[wave * 64 + lane] = accs;
    __syncthreads();
    for (int o = t; o < 4 * C; o += 256) {
        const int k = o / C, c = o - k * C, j = c >> 5, cl = c & 31;
        float sum = 0.f;
        for (int wv = 0; wv < 4; ++wv)
            for (int hf = 0; hf < 2; ++hf) sum += rs[((wv * 2 + hf) * NB + j) * 128 + k * 32 + cl];
        bn_part[((long)blockIdx.x * 4 + k) * C + c] = sum;
    }
    if (t < 9) {
        double sum = 0.0;
        for (int wv = 0; wv < 4; ++wv)
            for (int hf = 0; hf < 2; ++hf) sum += (double)ra[wv * 64 + hf * 32 + t];
        out[9 * C + t] = sum;
    }
}

int conv_last_bwd_tail_fused_launch(const TailSkip& sk, const float* dout, const float* wl, double* wpartial, float* bn_part, int n,
                                    int h, int w, int c, hipStream_t s) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H), nb = conv_last_tail_blocks(n, h, w);
    {
        // matrix-pipe form (r05): OPT-IN (edge_conv >= 0 with bit 256) -- 0.181 -> 0.161 ms for the kernel, nothing end to end
        // (profiles/r05_notes.md section 13)
        const int v = tune(TUNE_EDGE_CONV);
        const double zb = 4.0 * n * h * (double)w * c;
        if ((c == 32 || c == 64) && v >= 0 && (v & 256) && zb < 4294967040.0) {
            const size_t words = (size_t)4 * (c / 32) * 16 * 64;      // >= 640 + the small sums
            if (c == 64) RD_LAUNCH(conv_last_bwd_tail_mfma_kernel<2>, dim3(nb), dim3(256), words * sizeof(float), s, sk, dout, wl, wpartial, bn_part, n, h, w, tx, ty, n * tx * ty, (unsigned)zb);
            else RD_LAUNCH(conv_last_bwd_tail_mfma_kernel<1>, dim3(nb), dim3(256), words * sizeof(float), s, sk, dout, wl, wpartial, bn_part, n, h, w, tx, ty, n * tx * ty, (unsigned)zb);
            RD_LAUNCH_CHECK("conv_last_bwd_tail_mfma");
            return RD_OK;
        }
    }
    const size_t smem = (640 + (size_t)(256 / (c / 4)) * 9 * c) * sizeof(float);
    RD_LAUNCH(conv_last_bwd_tail_fused_kernel, dim3(nb), dim3(256), smem, s, sk, dout, wl, wpartial, bn_part, n, h, w, c, c / 4,
                       tx, ty, n * tx * ty);
    RD_LAUNCH_CHECK("conv_last_bwd_tail_fused");
    return RD_OK;
}

int tail_wl_finish_launch(const double* partial, int nb, const double* c16, const float* wt, const float* bt, float* dw, float* dbias,
                          int cin, int c0, hipStream_t s) {
    RD_LAUNCH(tail_wl_finish_kernel, dim3(c0), dim3(256), 0, s, partial, nb, c16, wt, bt, dw, dbias, cin, c0);
    RD_LAUNCH_CHECK("tail_wl_finish");
    return RD_OK;
}

int conv_last_dgrad_launch(const float* dout, const float* wt, float* ds, int n, int h, int w, int c, hipStream_t s, int* launched) {
    *launched = 0;
    if (!edge_shape_ok(c) || !edge_on(2)) return RD_OK;
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const BnHook none = {};
    RD_LAUNCH(conv_last_dgrad_tile_kernel<false>, dim3(n * tx * ty), dim3(256), 0, s, dout, wt, ds, n, h, w, c, c / 4, tx,
                       ty, none);
    RD_LAUNCH_CHECK("conv_last_dgrad");
    *launched = 1;
    return RD_OK;
}

// same + the BN-backward statistics of the last decoder level's consumer; *rows = partial rows written (0: shape not handled)
int conv_last_dgrad_bn_launch(const float* dout, const float* wt, float* ds, int n, int h, int w, int c, const float* bn_z,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                              const float* slope_dev, float* part, hipStream_t s, int* rows) {
    *rows = 0;
    if (!edge_shape_ok(c) || !edge_on(2)) return RD_OK;
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const BnHook bn = {bn_z, mean, invstd, gamma, beta, slope_dev, slope, part};
    RD_LAUNCH(conv_last_dgrad_tile_kernel<true>, dim3(n * tx * ty), dim3(256), 0, s, dout, wt, ds, n, h, w, c, c / 4, tx,
                       ty, bn);
    RD_LAUNCH_CHECK("conv_last_dgrad");
    *rows = n * tx * ty;
    return RD_OK;
}

int tail_compose_launch(const float* wt, const float* bt, const float* wl, float* M, float* V, float* VT, float* B9, int cin, int c0,
                        hipStream_t s) {
    RD_LAUNCH(tail_compose_kernel, dim3(cin), dim3(64), 0, s, wt, bt, wl, M, V, VT, B9, cin, c0);
    RD_LAUNCH_CHECK("tail_compose");
    return RD_OK;
}

// 256 / (Cin / 4) pixel slots; knob edge_conv bit 32 switches the composed tail kernels off
bool tail_shape_ok(int cin) { return (cin == 16 || cin == 32 || cin == 64 || cin == 128 || cin == 256) && edge_on(32); }

// coarse grid hc x wc; *rows = statistics rows written (0 without a hook)
int convt_last_dgrad_launch(const float* dout, const float* V, float* dprev, int n, int hc, int wc, int cin, const float* bn_z,
                            const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                            const float* slope_dev, float* part, hipStream_t s, int* rows) {
    const int tx = cdiv(wc, ET_W), ty = cdiv(hc, ET_H);
    *rows = 0;
    if (bn_z) {
        const BnHook bn = {bn_z, mean, invstd, gamma, beta, slope_dev, slope, part};
        RD_LAUNCH(convt_last_dgrad_kernel<true>, dim3(n * tx * ty), dim3(256), 0, s, dout, V, dprev, n, hc, wc, cin, cin / 4,
                           tx, ty, bn);
        *rows = n * tx * ty;
    } else {
        const BnHook none = {};
        RD_LAUNCH(convt_last_dgrad_kernel<false>, dim3(n * tx * ty), dim3(256), 0, s, dout, V, dprev, n, hc, wc, cin, cin / 4,
                           tx, ty, none);
    }
    RD_LAUNCH_CHECK("convt_last_dgrad");
    return RD_OK;
}

int tail_corr_blocks(int n, int hc, int wc) {
    const long nt = (long)n * cdiv(wc, ET_W) * cdiv(hc, ET_H);
    return (int)(nt < 512 ? nt : 512);
}

int tail_t16_launch(const TailSkip& sk, const float* V, float* t16, long pixels, int cin, hipStream_t s) {
    const long ntile = (pixels + 15) / 16;
    const int grid = (int)(ntile / 4 < 2048 ? (ntile + 3) / 4 : 2048);
    switch (cin) {
        case 16: RD_LAUNCH(tail_t16_kernel<4>, dim3(grid), dim3(256), 0, s, sk, V, t16, pixels); break;
        case 32: RD_LAUNCH(tail_t16_kernel<8>, dim3(grid), dim3(256), 0, s, sk, V, t16, pixels); break;
        case 64: RD_LAUNCH(tail_t16_kernel<16>, dim3(grid), dim3(256), 0, s, sk, V, t16, pixels); break;
        case 128: RD_LAUNCH(tail_t16_kernel<32>, dim3(grid), dim3(256), 0, s, sk, V, t16, pixels); break;
        default: RD_LAUNCH(tail_t16_kernel<64>, dim3(grid), dim3(256), 0, s, sk, V, t16, pixels); break;
    }
    RD_LAUNCH_CHECK("tail_t16");
    return RD_OK;
}

int convt_last_wgrad_launch(const float* x, const TailSkip& sk, const float* dout, const float* wl, float* dwt, double* partial,
                            double* c16, int n, int hc, int wc, int cin, int c0, hipStream_t s) {
    const int tx = cdiv(wc, ET_W), ty = cdiv(hc, ET_H), nb = tail_corr_blocks(n, hc, wc);
    const size_t smem = (2368 + 16384) * sizeof(float);
    RD_LAUNCH(tail_corr_kernel, dim3(nb), dim3(256), smem, s, x, sk, dout, partial, n, hc, wc, cin, cin / 4, tx, ty, n * tx * ty);
    RD_LAUNCH(tail_wgrad_finish_kernel, dim3(cin), dim3(256), 0, s, (const double*)partial, nb, wl, dwt, c16, cin, c0);
    RD_LAUNCH_CHECK("convt_last_wgrad");
    return RD_OK;
}

int conv_last_wgrad_blocks(int n, int h, int w, int c) {       // 0 = shape not handled here
    if (!edge_shape_ok(c) || !edge_on(4)) return 0;
    const long nt = (long)n * cdiv(w, ET_W) * cdiv(h, ET_H);
    return (int)(nt < 1024 ? nt : 1024);
}

int conv_last_wgrad_launch(const float* s_in, const float* dout, double* partial, int n, int h, int w, int c, hipStream_t s) {
    const int tx = cdiv(w, ET_W), ty = cdiv(h, ET_H);
    const int nb = conv_last_wgrad_blocks(n, h, w, c);
    const size_t smem = (640 + (size_t)(256 / (c / 4)) * 9 * c) * sizeof(float);       // 640 + 9 * 256 * 4 floats = 39.4 KB
    RD_LAUNCH(conv_last_wgrad_tile_kernel, dim3(nb), dim3(256), smem, s, s_in, dout, partial, n, h, w, c, c / 4, tx, ty,
                       n * tx * ty);
    RD_LAUNCH_CHECK("conv_last_wgrad");
    return RD_OK;
}

}  // namespace rd
