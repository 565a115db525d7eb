// ConvTranspose2d(k=2, s=2) family on the gfx950 matrix cores (split-bf16, hi/lo accumulators; rd_mfma_dev.h).
// Replaces the ATen kernels behind nn.ConvTranspose2d forward / autograd and the SkipConnection ADD
// (reference: lib/UNet.py:21, 63, 96-101, 181, 213-224).
//
// A k2s2 transposed convolution has no overlap between taps: out[2g+a][2x+b][co] = bias[co] + sum_ci x[g][x][ci] * W[ci][co][a][b]
// (g = n*H + y: the images of a batch are stacked rows, so an image border needs no special case).  It is four 1x1
// convolutions sharing the input; as a GEMM: M = input pixels, K = Cin, N' = (a, b, co).
//
// convt_fwd_kernel: block tile = a PATCH of TR x PW input pixels (32*TM rows) x 128 consecutive N' columns.  For Cout >= 128
// the 128 columns are one (a, b) quadrant and 128 output channels; for Cout = 64 they are one output row parity `a` with both
// `b` -- in both cases a tile row (one input pixel) owns 512 CONTIGUOUS bytes of the output, and for Cout = 64 the PW pixels
// of a patch row own one contiguous PW * 512-byte segment of an output row (the generic NT kernel scattered 16-byte pieces).
// Epilogue (r04): straight from the accumulator registers -- a lane holds one output column, lanes 0-31 of a buffer access cover
// 128 contiguous bytes of one output pixel, the pixel part of the address is a scalar offset: bias, skip tensor and the lazy
// act(BN(z)) recomputation of the encoder skip without an LDS round trip or a barrier (all skip loads issued first).  The r03
// epilogue (32 rows x 128 columns staged through LDS, 16 bytes per lane) is kept behind `nt_epi = 0` for A/B runs.
// Main loop = the split NT pipeline of rd_igemm.hip (global load -> split + LDS write -> fragment read -> MFMA over four
// K-steps, one barrier per 16-channel step, weights straight from global memory in fragment order).
#include "rd_common.h"
#include "rd_mfma_dev.h"
#include "rd_nt.h"

namespace rd {

struct CtParams {
    const float* x;
    const void* wsplit;      // split-bf16 fragment layout of wtf[(ab, co)][ci] (rd_pack_convt2x2_weight)
    const float* bias;
    const float* skip;
    const float* sk_mean;    // non-null: `skip` is the pre-BatchNorm conv output z; skip value = act(gamma*(z-mean)*invstd + beta)
    const float* sk_invstd;
    const float* sk_gamma;
    const float* sk_beta;
    const float* sk_slope_dev;
    float sk_slope;
    float* out;
    int G, W, Cin, Cout;     // G = N * H stacked input rows
    int PW, logPW, TR;       // patch = TR rows x PW columns, TR * PW = 32 * TM
    int tiles_x, ngroups, nk;
    unsigned x_bytes, w_bytes;
    int direct;              // register-direct epilogue (r04) instead of the LDS-staged one (tune nt_epi = 0)
    // three-product arithmetic (rd_mfma_dev.h): magnitude slots of x / of the weights (both non-null: NP = 3 body unless a
    // maximum is infinite), the two-term fp16 fragments of the weights, and the slot that receives max |out|
    const unsigned* a_amax;
    const unsigned* b_amax;
    const void* wsplit3;
    unsigned w_bytes3;
    unsigned* out_amax;
    // per-image slots (rd_quant_next_img): a_amax / out_amax are arrays, amax_img_stride words per image, and a tile's rows lie
    // inside one image of img_rows stacked rows (the launcher checks TR | H); 0 = one slot per tensor
    int amax_img_stride, img_rows;
};

__device__ __forceinline__ float ct_act(float y, float slope) { return y > 0.f ? y : y * slope; }

// Staging task -> tile row.  A ds_write_b64 is serviced in groups of 16 CONTIGUOUS lanes with 32 dword banks; with the natural
// assignment (task idx -> row idx) a group writes rows r .. r+3, whose 8-bank windows at the 28-word row pitch start at banks
// 0, 28, 24, 20: neighbouring windows overlap by half (2-way conflict on every staging write: SQ_LDS_BANK_CONFLICT = 25 % of the
// LDS cycles of convt_fwd / convt_dgrad in profiles/r04_summary.json, 9-15 % in the halo kernel, whose fragment READS were
// conflict-free already).  Rows r, r+2, r+4, r+6 start at banks 0, 24, 16, 8: disjoint.  So the 32 tasks of an aligned block of
// eight rows are dealt out as (low bits w2 w1 w0 of idx) -> row (w1 w0 w2): lane group {p = w2} takes rows p, p+2, p+4, p+6.
// Which pixel a 4-lane cluster loads does not matter to the global loads (64 contiguous bytes per cluster either way).
__device__ __forceinline__ int stage_row(int idx) { return (idx & ~7) | ((idx & 3) << 1) | ((idx >> 2) & 1); }

template <int NP, int TM>
__device__ __forceinline__ void convt_fwd_body(const CtParams& p, float* smem, const Quant qz, const int amax_off) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;
    constexpr int BM = 32 * TM, RS = 28;                  // LDS row = 3 terms x 16 bf16 + 16 B pad (conflict-free b128 reads)
    constexpr int STAGE = BM * RS;
    constexpr int NLD = (BM * 4 + 255) / 256;             // staging float4 per thread and K-step
    constexpr int CS = 128 + 4;

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = lb % p.ngroups, tile_m = lb / p.ngroups;     // the column groups of one patch run back to back (A from L2)
    const int tx = tile_m % p.tiles_x, ty = tile_m / p.tiles_x;
    const int g0 = ty * p.TR, x0 = tx * p.PW;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int W = p.W, logPW = p.logPW, pwm = p.PW - 1;

    f32x16 acc[TM], lo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = lo[i][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.x, p.x_bytes), rsB = make_rsrc(NP == 3 ? p.wsplit3 : p.wsplit, NP == 3 ? p.w_bytes3 : p.w_bytes);
    unsigned s_off[NLD];
    int s_lds[NLD];
    const int c4 = t & 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int row = stage_row((t + 256 * k) >> 2);
        const int g = g0 + (row >> logPW), xx = x0 + (row & pwm);
        const bool ok = row < BM && g < p.G;
        s_off[k] = ok ? (unsigned)((((long)g * W + xx) * p.Cin + c4 * 4) * 4) : kOOB;
        s_lds[k] = row < BM ? row * RS + c4 * 2 : -1;
    }
    const int nb = cg * 4 + wave;                         // 32-column block of this wave
    const unsigned b_off = (unsigned)(((long)nb * p.nk * NT) * 1024 + lane * 16);

    auto load_a = [&](int kt, float4 (&ra)[NLD]) {
        const bool cok = kt < p.nk;
#pragma unroll
        for (int k = 0; k < NLD; ++k) ra[k] = buf_load4(rsA, cok ? s_off[k] : kOOB, (unsigned)(kt * SK * 4));
    };
    auto store_a = [&](float* stage, const float4 (&ra)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (s_lds[k] < 0) continue;
            uint2 ph, pm, pl;
            split_pack4<NP>(ra[k], qz.sa, ph, pm, pl);
            float* row = stage + s_lds[k];
            *reinterpret_cast<uint2*>(row) = ph;
            *reinterpret_cast<uint2*>(row + 8) = pm;
            if (kterm3<NP>()) *reinterpret_cast<uint2*>(row + 16) = pl;
        }
    };
    auto load_b = [&](int kt, uint4 (&rb)[3]) {
        const unsigned voff = kt < p.nk ? b_off : kOOB;
#pragma unroll
        for (int q = 0; q < NT; ++q) rb[q] = buf_load4u(rsB, voff, (unsigned)((kt * NT + q) * 1024));
    };
    const int lrow = lane & 31, half = lane >> 5;
    FR af[TM][3];
    auto read_a = [&](const float* stage) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q)
                af[i][q] = *reinterpret_cast<const FR*>(stage + (i * 32 + lrow) * RS + half * 4 + q * 8);
    };
    constexpr int GP = TM >= 2 ? 2 : 1;
    constexpr int NB_ = TM >= 4 ? 3 : 4;
    // One K-step (16 input channels).  Pipeline per tile j: global load (step j-3) -> split + LDS write (step j-1) ->
    // fragment read (end of step j-1, after the barrier) -> MFMA (step j); LDS stage of tile j = j & 1.
    auto step = [&](int kt, float4 (&ra)[NLD], uint4 (&bcur)[3], uint4 (&bnew)[3], float* stage_next) {
        store_a(stage_next, ra);          // tile kt+1 (this stage was last read before the previous barrier)
        load_a(kt + 3, ra);
        load_b(kt + NB_ - 1, bnew);        // into the register set of tile kt-1
        FR bf[3];
#pragma unroll
        for (int q = 0; q < NT; ++q) bf[q] = __builtin_bit_cast(FR, bcur[q]);
#pragma unroll
        for (int g = 0; g < TM; g += GP) {
#pragma unroll
            for (int t6 = lo0<NP>(); t6 < 5; ++t6)
#pragma unroll
                for (int i = g; i < g + GP; ++i) lo[i] = mfma16<NP>(af[i][PA6[t6]], bf[PB6[t6]], lo[i]);
#pragma unroll
            for (int i = g; i < g + GP; ++i) acc[i] = mfma16<NP>(af[i][0], bf[0], acc[i]);
        }
        __syncthreads();
        read_a(stage_next);               // tile kt+1, consumed by the next step
    };

    float* st0 = smem;
    float* st1 = smem + STAGE;
    // A register sets alternate with period 2; weight-fragment ring of NB sets: 3 for the 128-row tile (period 6, exits after
    // every pair of steps: nk is even), 4 for the 64-row tile (period 4 divides nk = Cin / 16 when Cin % 64 == 0, no exit
    // inside the body -- with exits hipcc keeps several copies of the accumulators and the 64-row tile spills, cf. convt_dgrad)
    constexpr int NB = TM >= 4 ? 3 : 4, PER = TM >= 4 ? 6 : 4;
    float4 ra[2][NLD];
    uint4 bq[NB][3];
    load_a(0, ra[0]);
    load_a(1, ra[1]);
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) load_b(j, bq[j]);
    store_a(st0, ra[0]);
    load_a(2, ra[0]);
    __syncthreads();
    read_a(st0);
#pragma unroll 1
    for (int kt = 0; kt < p.nk; kt += PER) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (NB == 3 && u > 0 && (u & 1) == 0 && kt + u >= p.nk) break;
            step(kt + u, ra[(u + 1) & 1], bq[u % NB], bq[(u + NB - 1) % NB], (u & 1) ? st0 : st1);
        }
    }
    __syncthreads();

    if (p.direct) {
        // ---- register-direct epilogue (r04; cf. nt_epilogue_direct in rd_nt.h): no LDS round trip, no barriers.  A lane holds
        // ONE output column (32-column block `wave` of the tile's 128 contiguous floats per pixel) of sixteen tile rows per
        // 32 x 32 block -- so one bias value and one (scale, shift) pair per lane instead of four; lanes 0-31 of a load / store
        // cover 128 contiguous bytes of one output pixel, lanes 32-63 the pixel four input columns on.  The pixel part of the
        // address is wave-uniform (SGPR offset of a buffer access), the lane part one offset computed once.  All skip loads of
        // the tile are issued before the first use.
        const int col0d = cg * 128;
        const int ab0d = col0d / p.Cout, co0d = col0d - ab0d * p.Cout;
        const long obased = (((long)(2 * g0 + (ab0d >> 1)) * (2 * W)) + 2 * x0 + (ab0d & 1)) * p.Cout + co0d;
        const int ystr = __builtin_amdgcn_readfirstlane(4 * W * p.Cout * 4), xstr = __builtin_amdgcn_readfirstlane(2 * p.Cout * 4);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        int cod = co0d + wv * 32 + lrow;
        if (cod >= p.Cout) cod -= p.Cout;                   // Cout = 64: columns 64..127 are the b = 1 pixel
        const float bias1 = p.bias ? p.bias[cod] : 0.f;
        // skip value = act(sc * s + sh): the lazy act(BN(z)) recomputation; a materialised skip is the identity case
        // (sc, sh, slope) = (1, 0, 1), no skip at all loads zeros (out-of-extent offset) -- one code path, no per-element selects
        float sc1 = 1.f, sh1 = 0.f, slope1 = 1.f;
        if (p.sk_mean) {
            sc1 = p.sk_invstd[cod] * p.sk_gamma[cod];
            sh1 = p.sk_beta[cod] - p.sk_mean[cod] * sc1;
            slope1 = p.sk_slope_dev ? p.sk_slope_dev[0] : p.sk_slope;
        }

        const unsigned span = (unsigned)(p.TR * (4 * W * p.Cout * 4));
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(p.out + obased, span), rsS = make_rsrc(p.skip ? p.skip + obased : p.out + obased, span);
        const unsigned lane_off = (unsigned)(4 * half * xstr + (wv * 32 + lrow) * 4);
        const unsigned lane_off_s = p.skip ? lane_off : kOOB;
        const int rows_left = __builtin_amdgcn_readfirstlane(p.G - g0);         // patch rows inside the stacked image rows
        auto soffd = [&](int i, int r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2);                      // + 4 * half in the lane offset
            return (unsigned)((row >> logPW) * ystr + (row & pwm) * xstr);
        };
        auto rowok = [&](int i, int r) { return ((i * 32 + (r & 3) + 8 * (r >> 2)) >> logPW) < rows_left; };
        // FULL: every patch row of the tile exists (all tiles but the last row band of a ragged grid) -- no per-row select
        // BNSKIP = false: the skip is a materialised tensor (or absent): a plain add -- the identity case of the lazy form would
        // spend a multiply-add, a compare, a multiply and a select per element on (1, 0, slope 1) (inference: every level but
        // the first hands its activation on as a tensor; 4 of the ~10 vector instructions per output element of this epilogue)
        auto emit = [&](auto full_c, auto bn_c) {
            constexpr bool FULL = decltype(full_c)::value, BNSKIP = decltype(bn_c)::value;
            auto voff = [&](int i, int r) { return FULL || rowok(i, r) ? lane_off : kOOB; };
            auto voff_s = [&](int i, int r) { return FULL || rowok(i, r) ? lane_off_s : kOOB; };
            float skv[TM][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    skv[i][r] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsS, voff_s(i, r), soffd(i, r), 0));
            float omax = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = merge_q<NP>(acc[i][r], lo[i][r], qz.dexp) + bias1;
                    const float s1 = BNSKIP ? ct_act(fmaf(skv[i][r], sc1, sh1), slope1) : skv[i][r];
                    const float o = s1 + v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(o), rsO, voff(i, r), soffd(i, r), 0);
                    if (FULL || rowok(i, r)) omax = amax_acc(omax, o);
                }
            if (p.out_amax) amax_commit(p.out_amax + amax_off, omax);
        };
        if (p.sk_mean) {
            if (rows_left >= p.TR) emit(std::true_type(), std::true_type());
            else emit(std::false_type(), std::true_type());
        } else {
            if (rows_left >= p.TR) emit(std::true_type(), std::false_type());
            else emit(std::false_type(), std::false_type());
        }
        return;
    }
    // ---- epilogue: 32-row passes through LDS; tile row r = (gy, px) owns 128 contiguous floats of the output
    const int col0 = cg * 128;
    const int ab0 = col0 / p.Cout, co0 = col0 - ab0 * p.Cout;
    const long obase = (((long)(2 * g0 + (ab0 >> 1)) * (2 * W)) + 2 * x0 + (ab0 & 1)) * p.Cout + co0;
    const long ystride = (long)4 * W * p.Cout;
    const int xstride = 2 * p.Cout;
    const int q4 = t & 31, erow = t >> 5;                 // this thread's 16-byte column and first row of a pass (rows erow + 8k)
    int co = co0 + q4 * 4;
    if (co >= p.Cout) co -= p.Cout;                       // Cout = 64: columns 64..127 are the b = 1 pixel
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, slope = 0.f;
    if (p.sk_mean) {
        const float4 mu = *reinterpret_cast<const float4*>(p.sk_mean + co), is = *reinterpret_cast<const float4*>(p.sk_invstd + co);
        const float4 ga = *reinterpret_cast<const float4*>(p.sk_gamma + co), be = *reinterpret_cast<const float4*>(p.sk_beta + co);
        sc[0] = is.x * ga.x; sc[1] = is.y * ga.y; sc[2] = is.z * ga.z; sc[3] = is.w * ga.w;
        sh[0] = be.x - mu.x * sc[0]; sh[1] = be.y - mu.y * sc[1]; sh[2] = be.z - mu.z * sc[2]; sh[3] = be.w - mu.w * sc[3];
        slope = p.sk_slope_dev ? p.sk_slope_dev[0] : p.sk_slope;
    }
    float* Cs = smem;
    float omax = 0.f;
    // all skip loads of the tile go out first (the fragment / weight registers of the main loop are dead by now): the
    // HBM latency is paid once per block instead of once per 32-row pass
    long off[TM][4];
    bool ok[TM][4];
    float4 sk[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = i * 32 + erow + 8 * k;
            const int gy = r >> logPW, px = r & pwm;
            ok[i][k] = g0 + gy < p.G;
            off[i][k] = obase + gy * ystride + px * xstride + q4 * 4;
            sk[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.skip && ok[i][k]) sk[i][k] = *reinterpret_cast<const float4*>(p.skip + off[i][k]);
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Cs[((r & 3) + 8 * (r >> 2) + 4 * half) * CS + wave * 32 + lrow] = merge_q<NP>(acc[i][r], lo[i][r], qz.dexp);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[i][k]) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[(erow + 8 * k) * CS + q4 * 4]);
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (p.skip) {
                float4 s4 = sk[i][k];
                if (p.sk_mean) {
                    s4.x = ct_act(fmaf(s4.x, sc[0], sh[0]), slope);
                    s4.y = ct_act(fmaf(s4.y, sc[1], sh[1]), slope);
                    s4.z = ct_act(fmaf(s4.z, sc[2], sh[2]), slope);
                    s4.w = ct_act(fmaf(s4.w, sc[3], sh[3]), slope);
                }
                v.x = s4.x + v.x; v.y = s4.y + v.y; v.z = s4.z + v.z; v.w = s4.w + v.w;
            }
            *reinterpret_cast<float4*>(p.out + off[i][k]) = v;
            omax = amax_acc(amax_acc(amax_acc(amax_acc(omax, v.x), v.y), v.z), v.w);
        }
        __syncthreads();
    }
    if (p.out_amax) amax_commit(p.out_amax + amax_off, omax);
}

template <int TM>
__global__ __launch_bounds__(256, 2) void convt_fwd_kernel(CtParams p) {
    constexpr int STAGE = 32 * TM * 28, EPI_WORDS = 32 * (128 + 4);
    constexpr int SMEM = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    int amax_off = 0;
    if (p.amax_img_stride) {                                  // this block's image (the body's tile -> row map)
        const int lb = xcd_remap(blockIdx.x, gridDim.x);
        const int ty = (lb / p.ngroups) / p.tiles_x;
        amax_off = ((ty * p.TR) / p.img_rows) * p.amax_img_stride;
    }
    const Quant qz = quant_select(p.a_amax ? p.a_amax + amax_off : nullptr, p.b_amax);
    if (qz.use3) convt_fwd_body<3, TM>(p, smem, qz, amax_off);
    else convt_fwd_body<6, TM>(p, smem, qz, amax_off);
}

// ---- data gradient:  dx[p][ci] = sum_{a,b,co} dout[2g+a][2x+b][co] * W[ci][co][a][b]   (p = (g, x), g = n*H + y) --------
// GEMM with M = coarse pixels, N = Cin, K = (a, b, co).  For a fixed `a` the K range (b, co) of pixel (g, x) is the 2*Cd
// CONTIGUOUS floats dout[2g+a][2x .. 2x+1][:] -- and consecutive x continue the same run -- so with K ordered (a, b, co)
// ("tap outer") the A operand is a plain row-major matrix per `a`: every staging load is 64 contiguous bytes per row and
// consecutive K-steps walk along the same DRAM rows (the generic NT kernel orders K chunk-outer / tap-inner, which makes
// consecutive K-steps of a row jump by Cd floats and 2*W*Cd floats).  The packed weights keep their chunk-outer order
// (rd_pack_convt2x2_weight: kt' = chunk * 4 + tap); this kernel just asks for fragment kt' when it is at (tap, chunk).
// Pipeline = convt_fwd_kernel's (global load -> split + LDS write -> fragment read -> MFMA, one barrier per K-step, weight
// fragments straight from global memory into a three-deep register ring); epilogue = the shared NT epilogue (rd_nt.h:
// 16-byte row-contiguous stores through LDS + the BatchNorm-backward statistics hook).
//   <TM=4, WM=1, WN=4>: 128 pixels x 128 channels (Cin >= 128, big grids)     <TM=2, WM=1, WN=4>: 64 x 128 (small grids)
//   <TM=2, WM=2, WN=2>: 128 pixels x 64 channels (Cin = 64: the 128^2 -> 256^2 level, HBM-bound)
template <int NP, int TM, int WM, int WN>
__device__ __forceinline__ void convt_dgrad_body(const NtParams& p, float* smem, const Quant qz) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;
    constexpr int BM = 32 * TM * WM, BN = 32 * WN, RS = 28;
    constexpr int STAGE = BM * RS;
    constexpr int NLD = (BM * 4 + 255) / 256;
    constexpr int EPI_WORDS = 32 * (BN + 4) + 512;
    constexpr int SM0 = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    constexpr int SMEM = SM0 > 4096 ? SM0 : 4096;           // BN-backward statistics scratch of the epilogue

    const int lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;    // the column groups of one pixel tile run back to back
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int Cd = p.Cin, W = p.W;                          // Cd = channels of dout (the transposed convolution's Cout)

    f32x16 acc[TM][1], lo[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = lo[i][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, p.a_bytes), rsB = make_rsrc(NP == 3 ? p.Bsplit3 : p.Bsplit, NP == 3 ? p.b_bytes3 : p.b_bytes);
    unsigned s_off[NLD];
    int s_lds[NLD];
    const int c4 = t & 3;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int row = stage_row((t + 256 * k) >> 2);
        const int m = m0 + row;
        const int g = (int)fd_div((unsigned)m, p.pd.w), x = m - g * W;       // stacked image rows: g = n*H + y
        const bool ok = row < BM && m < p.M;
        s_off[k] = ok ? (unsigned)((((long)(4 * g) * W + 2 * x) * Cd + c4 * 4) * 4) : kOOB;
        s_lds[k] = row < BM ? row * RS + c4 * 2 : -1;
    }
    const int nb = (n0 >> 5) + wn;                          // 32-column block of this wave
    const unsigned b_off = (unsigned)(((long)nb * p.nk * NT) * 1024 + lane * 16);
    const int cpt = p.chunks;                               // 16-channel chunks per tap
    const int half_nk = 2 * cpt;
    const unsigned arow = (unsigned)__builtin_amdgcn_readfirstlane(2 * W * Cd * 4);     // bytes from fine row 2g to fine row 2g+1

    auto load_a = [&](int kt, float4 (&ra)[NLD]) {
        const bool cok = kt < p.nk;
        const int a = kt >= half_nk ? 1 : 0;
        // (readfirstlane: the scalar offset must be an SGPR -- left to its divergence analysis hipcc computed it in a VGPR and
        // wrapped every load in a waterfall loop)
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
            cok ? (int)((a ? arow : 0u) + (unsigned)((kt - a * half_nk) * (SK * 4))) : 0);
#pragma unroll
        for (int k = 0; k < NLD; ++k) ra[k] = buf_load4(rsA, cok ? s_off[k] : kOOB, soff);
    };
    auto store_a = [&](float* stage, const float4 (&ra)[NLD]) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (s_lds[k] < 0) continue;
            uint2 ph, pm, pl;
            split_pack4<NP>(ra[k], qz.sa, ph, pm, pl);
            float* row = stage + s_lds[k];
            *reinterpret_cast<uint2*>(row) = ph;
            *reinterpret_cast<uint2*>(row + 8) = pm;
            if (kterm3<NP>()) *reinterpret_cast<uint2*>(row + 16) = pl;
        }
    };
    // weight fragments: K-step kt = (tap, chunk) in tap-outer order lives at kt' = chunk * 4 + tap of the packed operand
    int b_tap = 0, b_chunk = 0;                             // position of the NEXT load_b call (called with kt = 0, 1, 2, ...)
    auto load_b = [&](int kt, uint4 (&rb)[3]) {
        const unsigned voff = kt < p.nk ? b_off : kOOB;
        const int ktp = __builtin_amdgcn_readfirstlane(kt < p.nk ? b_chunk * 4 + b_tap : 0);
#pragma unroll
        for (int q = 0; q < NT; ++q) rb[q] = buf_load4u(rsB, voff, (unsigned)((ktp * NT + q) * 1024));
        if (++b_chunk == cpt) { b_chunk = 0; ++b_tap; }
    };
    const int lrow = lane & 31, half = lane >> 5;
    FR af[TM][3];
    auto read_a = [&](const float* stage) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q)
                af[i][q] = *reinterpret_cast<const FR*>(stage + ((wm * TM + i) * 32 + lrow) * RS + half * 4 + q * 8);
    };
    constexpr int GP = TM >= 2 ? 2 : 1;

    float* st0 = smem;
    float* st1 = smem + STAGE;
    // A register sets alternate with period 2; weight-fragment ring of NB sets: 3 for the 128-row tiles (period 6, exits
    // after every pair of steps: nk = 4 * cpt is even) -- the 64-row tiles take 4 (period 4 divides nk, no exit inside the
    // loop body): with the exits hipcc failed to coalesce their four accumulators across the steps (12 accumulator register
    // blocks, 256 VGPRs + scratch)
    constexpr int NB = TM >= 4 ? 3 : 4, PER = TM >= 4 ? 6 : 4;
    float4 ra[2][NLD];
    uint4 bq[NB][3];
    load_a(0, ra[0]);
    load_a(1, ra[1]);
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) load_b(j, bq[j]);
    store_a(st0, ra[0]);
    load_a(2, ra[0]);
    __syncthreads();
    read_a(st0);
    auto step = [&](int kt, float4 (&ra_)[NLD], uint4 (&bcur)[3], uint4 (&bnew)[3], float* stage_next) {
        store_a(stage_next, ra_);         // tile kt+1 (this stage was last read before the previous barrier)
        load_a(kt + 3, ra_);
        load_b(kt + NB - 1, bnew);
        FR bf[3];
#pragma unroll
        for (int q = 0; q < NT; ++q) bf[q] = __builtin_bit_cast(FR, bcur[q]);
#pragma unroll
        for (int g = 0; g < TM; g += GP) {
#pragma unroll
            for (int t6 = lo0<NP>(); t6 < 5; ++t6)
#pragma unroll
                for (int i = g; i < g + GP; ++i) lo[i] = mfma16<NP>(af[i][PA6[t6]], bf[PB6[t6]], lo[i]);
#pragma unroll
            for (int i = g; i < g + GP; ++i) acc[i][0] = mfma16<NP>(af[i][0], bf[0], acc[i][0]);
        }
        __syncthreads();
        read_a(stage_next);               // tile kt+1, consumed by the next step
    };
#pragma unroll 1
    for (int kt = 0; kt < p.nk; kt += PER) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (NB == 3 && u > 0 && (u & 1) == 0 && kt + u >= p.nk) break;
            step(kt + u, ra[(u + 1) & 1], bq[u % NB], bq[(u + NB - 1) % NB], (u & 1) ? st0 : st1);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = merge_q<NP>(acc[i][0][r], lo[i][r], qz.dexp);
    nt_epilogue<BM, BN, WM, WN, EPI_STORE, SMEM, 1>(acc, smem, p, m0, n0, tile_m);
}

template <int TM, int WM, int WN>
__global__ __launch_bounds__(256, 2) void convt_dgrad_kernel(NtParams p) {
    constexpr int STAGE = 32 * TM * WM * 28, EPI_WORDS = 32 * (32 * WN + 4) + 512;
    constexpr int SM0 = 2 * STAGE > EPI_WORDS ? 2 * STAGE : EPI_WORDS;
    constexpr int SMEM = SM0 > 4096 ? SM0 : 4096;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    const Quant qz = quant_select(p.a_amax, p.b_amax);
    if (qz.use3) convt_dgrad_body<3, TM, WM, WN>(p, smem, qz);
    else convt_dgrad_body<6, TM, WM, WN>(p, smem, qz);
}

// p: A = dout, Bsplit / b_bytes = the split operand of wtd[ci][(ab, co)], C = dx, M = coarse pixels, N = Cin, Cin = Cd (channels
// of dout), H, W, pd = coarse grid, bn_* hook optional.  *launched = 0: shape left to the generic NT kernel.
int convt_dgrad_launch(NtParams p, hipStream_t s, int* launched, int* tiles_m_out) {
    *launched = 0;
    if (!mfma_split() || tune(TUNE_CONVT_PATCH) == 0) return RD_OK;
    const int Cd = p.Cin;
    if (Cd % 16 != 0 || Cd < 64 || p.N % 32 != 0 || p.N < 64 || p.M < 4096) return RD_OK;
    const double ab = 16.0 * (double)p.M * Cd;
    if (ab >= 4294967040.0) return RD_OK;
    p.taps = 4;
    p.chunks = Cd / 16;
    p.nk = 4 * p.chunks;
    p.K = 4 * Cd;
    p.vec = 1;
    p.patch = 0;
    p.a_bytes = (unsigned)ab;
    if (mfma_products() != 3 || !p.a_amax || !p.b_amax) p.a_amax = p.b_amax = nullptr;    // six products
    // 64-row tiles (152 VGPRs: three waves per SIMD) beat 128-row tiles at every cfg-S level (r03: 176 vs 165 TFLOP/s at
    // 32^2 x 256, equal at 64^2 x 128; profiles/r03_notes.md); convt_patch = 3 forces the 128-row variant for A/B runs
    int cfg = p.N < 128 ? 2 : 1;
    if (p.N >= 128 && tune(TUNE_CONVT_PATCH) == 3) cfg = 0;
    const int bm = cfg == 1 ? 64 : 128, bn = cfg == 2 ? 64 : 128;
    p.tiles_n = cdiv(p.N, bn);
    const long tiles_m = cdiv(p.M, bm);
    const long grid = tiles_m * p.tiles_n;
    if (grid >= (1L << 31)) return RD_OK;
    if (tiles_m_out) *tiles_m_out = (int)tiles_m;
    p.direct = tune(TUNE_NT_EPI) != 0 && p.M % bm == 0 && p.N % 32 == 0 && !p.shift && !p.pool_out;     // rd_nt.h: nt_epilogue_direct
    char pcls[64];
    snprintf(pcls, sizeof(pcls), "convt2x2_dgrad|convt_dgrad<%s>", cfg == 0 ? "4,1,4" : cfg == 1 ? "2,1,4" : "2,2,2");
    ProfScope ps(s, pcls, 2.0 * p.M * (double)p.N * p.K, 4.0 * (4.0 * p.M * Cd + (double)p.N * p.K + (double)p.M * p.N * (p.bn_part ? 2 : 1)), true);
    if (cfg == 0) RD_LAUNCH((convt_dgrad_kernel<4, 1, 4>), dim3((unsigned)grid), dim3(256), 0, s, p);
    else if (cfg == 1) RD_LAUNCH((convt_dgrad_kernel<2, 1, 4>), dim3((unsigned)grid), dim3(256), 0, s, p);
    else RD_LAUNCH((convt_dgrad_kernel<2, 2, 2>), dim3((unsigned)grid), dim3(256), 0, s, p);
    RD_LAUNCH_CHECK("convt_dgrad");
    *launched = 1;
    return RD_OK;
}

// ---- weight gradient:  dW[ci][co][a][b] = sum_p x[p][ci] * dout[2g+a][2x+b][co] -----------------------------------------------
// TN GEMM  C[(ab, co)][ci] = sum over coarse pixels, split-K slabs + the fixed-order slab reduction of rd_igemm.hip.
// Both operands are activations, so BOTH are split into bf16 terms on the way into LDS (no pre-split weights here): the cost
// per MFMA is set by how many MFMAs consume one staged element.  The generic TN kernel (128 x 128 tile, 4 waves, hi / lo
// accumulators) stages 4096 elements per 96 MFMAs; this kernel takes a 256 x (64 * TN') tile with EIGHT waves and one
// accumulator per 32 x 32 block (like wgrad_strip_kernel): 8192 elements per 384 MFMAs at TN = 4 -- half the split
// arithmetic, LDS writes and barriers per MFMA.
//   tile rows   = the four (a, b) quadrants x 64 output channels: wave row wm IS quadrant ab, and the A operand of a K-step
//                 (16 consecutive coarse pixels of one coarse row) is two runs of 32 fine pixels x 256 contiguous bytes;
//   tile cols   = 64 * (BN / 64) input channels; K-step = 16 coarse pixels; staging task = 4 pixels x 4 channels (register
//                 transpose, as in the other weight-gradient kernels); waves 0-3 stage A, waves 4-7 stage B;
//   LDS         = 128-byte rows [chunk = 2 * term + k-half], XOR-swizzled (conflict-free b64 writes / b128 reads), two stages.
struct CtwParams {
    const float* dout;
    const float* x;
    float* slab;             // [splits][4 * Cd][Cin]
    int Cd, Cin;             // channels of dout (transposed convolution's Cout) / of x
    int W;                   // coarse width (multiple of 16)
    int spr;                 // K-steps per coarse row = W / 16
    int total_ks, kps;       // K-steps in all / per split
    int tiles_n, tiles_mn;
    unsigned a_bytes, b_bytes;
    const unsigned* a_amax;  // magnitude slots of dout / x (both non-null: three-product body, rd_mfma_dev.h)
    const unsigned* b_amax;
};

template <int NP, int TN>
__device__ __forceinline__ void convt_wgrad_body(const CtwParams& p, float* smem, const Quant qz) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;
    constexpr int BM = 256, BN = 64 * TN, ROWS = BM + BN;
    constexpr int STAGE = ROWS * 32;                         // words

    const int gb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = gb / p.tiles_mn;
    const int lb = gb - split * p.tiles_mn;
    const int tile_n = lb % p.tiles_n, tile_m = lb / p.tiles_n;
    const int co0 = tile_m * 64, ci0 = tile_n * BN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int Cd = p.Cd, Cin = p.Cin, W = p.W;

    // ---- staging role (wave-uniform): waves 0-3 the dout tile (wave = quadrant ab), waves 4-7 the x tile
    const bool isA = __builtin_amdgcn_readfirstlane(t) < 256;
    const int idx = isA ? t : t - 256;
    const int kq = idx & 3, rq = idx >> 2;                   // pixel quarter of the K-step, row quad of the operand tile
    const bool active = isA || rq * 4 < BN;
    const int lds_row0 = isA ? rq * 4 : BM + rq * 4;
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int px = kq * 4 + j;                           // coarse pixel within the K-step
        if (isA) {
            const int ab = rq >> 4, cq = rq & 15;
            voff[j] = (unsigned)(((((ab >> 1) * 2 * W + 2 * px + (ab & 1)) * Cd) + co0 + cq * 4) * 4);
        } else {
            voff[j] = active ? (unsigned)((px * Cin + ci0 + rq * 4) * 4) : kOOB;
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.dout, p.a_bytes), rsB = make_rsrc(p.x, p.b_bytes);
    const int ks0 = split * p.kps;
    const int ks1 = ks0 + p.kps < p.total_ks ? ks0 + p.kps : p.total_ks;

    auto load_task = [&](int ks, float4 (&v)[4]) {
        // K-step ks = 16 coarse pixels starting at (g, x0): g = ks / spr stacked coarse row, x0 = 16 * (ks % spr)
        const bool ok = ks < ks1;
        const int g = ks / p.spr, x0 = (ks - g * p.spr) * 16;
        const unsigned soffA = (unsigned)__builtin_amdgcn_readfirstlane(ok ? (int)(((unsigned)(4 * g) * W + 2 * x0) * (unsigned)Cd * 4u) : 0);
        const unsigned soffB = (unsigned)__builtin_amdgcn_readfirstlane(ok ? (int)(((unsigned)g * W + x0) * (unsigned)Cin * 4u) : 0);
        if (isA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = buf_load4(rsA, ok ? voff[j] : kOOB, soffA);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = buf_load4(rsB, ok ? voff[j] : kOOB, soffB);
        }
    };
    int wr_e[3];
    {
        const int sw = (lds_row0 >> 1) & 7, hi = kq >> 1;
#pragma unroll
        for (int q = 0; q < 3; ++q) wr_e[q] = lds_row0 * 32 + (kq & 1) * 2 + ((2 * q + hi) ^ sw) * 4;
    }
    // one channel (c = 0..3) of a staging task: 4 pixels -> three 8-byte pieces of that channel's LDS row.  The four pieces
    // of a task are issued BETWEEN the MFMA groups of a K-step (mma_tile): all eight waves leave the barrier together, so a
    // store phase in front of the MFMAs would idle the matrix pipe of every SIMD at the same time, while VALU work
    // interleaved with MFMAs is nearly free up to ~1 instruction per MFMA (profiles/r02_notes.md section 1)
    const float qs = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(isA ? qz.sa : qz.sb)));
    auto store_piece = [&](float* stage, const float4 (&v)[4], int c) {
        if (!active) return;
        uint2 ph, pm, pl;                   // the 4 pixels of channel c
        split_pack4v<NP, false>(c == 0 ? v[0].x : c == 1 ? v[0].y : c == 2 ? v[0].z : v[0].w, c == 0 ? v[1].x : c == 1 ? v[1].y : c == 2 ? v[1].z : v[1].w,
                            c == 0 ? v[2].x : c == 1 ? v[2].y : c == 2 ? v[2].z : v[2].w, c == 0 ? v[3].x : c == 1 ? v[3].y : c == 2 ? v[3].z : v[3].w,
                            qs, ph, pm, pl);
        const int flip = (c >> 1) * 4;      // rows lds_row0 + 2, + 3 carry the next swizzle value (chunk index ^ 1)
        float* rowp = stage + c * 32;
        *reinterpret_cast<uint2*>(rowp + (wr_e[0] ^ flip)) = ph;
        *reinterpret_cast<uint2*>(rowp + (wr_e[1] ^ flip)) = pm;
        if (kterm3<NP>()) *reinterpret_cast<uint2*>(rowp + (wr_e[2] ^ flip)) = pl;
    };
    auto store_task = [&](float* stage, const float4 (&v)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) store_piece(stage, v, c);
    };

    const int lrow = lane & 31, half = lane >> 5;
    int a_rd[2][3], b_rd[TN][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wm * 2 + i) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 3; ++q) a_rd[i][q] = row * 32 + ((2 * q + half) ^ ((row >> 1) & 7)) * 4;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = BM + (wn * TN + j) * 32 + lrow;
#pragma unroll
        for (int q = 0; q < 3; ++q) b_rd[j][q] = row * 32 + ((2 * q + half) ^ ((row >> 1) & 7)) * 4;
    }
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one K-step: multiply stage `cur`; meanwhile split + store the next tile (registers v) into stage `nxt`, one channel
    // piece after each group of 48 / 4 MFMAs
    auto mma_tile = [&](const float* cur, float* nxt, const float4 (&v)[4]) {
        // (TN < 4: 12 / 24 MFMAs per wave and K-step -- interleaving measured slower there: 133 vs 154 and 89 vs 98 TFLOP/s
        // at the 64^2 x 128 / 128^2 x 64 levels, where it gains 165 -> 175 at 16^2 x 512; the whole task goes first)
        FR af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q) af[i][q] = *reinterpret_cast<const FR*>(cur + a_rd[i][q]);
#pragma unroll
        for (int q = 0; q < NT; ++q) bf[0][q] = *reinterpret_cast<const FR*>(cur + b_rd[0][q]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (j + 1 < TN) {                                // next column block's fragments while this one multiplies
#pragma unroll
                for (int q = 0; q < NT; ++q) bf[(j + 1) & 1][q] = *reinterpret_cast<const FR*>(cur + b_rd[j + 1][q]);
            }
#pragma unroll
            for (int t6 = lo0<NP>(); t6 < 6; ++t6) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mfma16<NP>(af[i][PA6[t6]], bf[j & 1][PB6[t6]], acc[i][j]);
            }
            if (TN == 4) store_piece(nxt, v, j);
        }
    };

    if (ks0 < ks1) {
        float* st0 = smem;
        float* st1 = smem + STAGE;
        float4 v0[4], v1[4];
        load_task(ks0, v0);
        load_task(ks0 + 1, v1);
        store_task(st0, v0);
        load_task(ks0 + 2, v0);
        __syncthreads();
        // step i multiplies stage i & 1, splits + stores tile i + 1 into the other stage between its MFMAs and then requests
        // tile i + 3 into the registers just consumed.  Steps come in pairs with no exit in between (an odd count
        // multiplies one all-zero tile: loads past ks1 return zeros) -- an exit inside the body made hipcc keep several
        // copies of the accumulators (cf. convt_dgrad_kernel)
        for (int ks = ks0; ks < ks1; ks += 2) {
            if (TN < 4) {                    // whole task first, and the next global loads go out before the MFMAs
                store_task(st1, v1);
                load_task(ks + 3, v1);
            }
            mma_tile(st0, st1, v1);
            if (TN == 4) load_task(ks + 3, v1);
            __syncthreads();
            if (TN < 4) {
                store_task(st0, v0);
                load_task(ks + 4, v0);
            }
            mma_tile(st1, st0, v0);
            if (TN == 4) load_task(ks + 4, v0);
            __syncthreads();
        }
    }

    // slab rows: m' = ab * Cd + co (wave row wm = quadrant ab), columns ci
    float* out = p.slab + (long)split * (4L * Cd) * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ci0 + (wn * TN + j) * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * Cd + co0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                out[(long)m * Cin + n] = scale_q<NP>(acc[i][j][r], qz.dexp);
            }
        }
}

template <int TN>
__global__ __launch_bounds__(512, 2) void convt_wgrad_kernel(CtwParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (256 + 64 * TN) * 32];
    const Quant qz = quant_select(p.a_amax, p.b_amax);
    if (qz.use3) convt_wgrad_body<3, TN>(p, smem, qz);
    else convt_wgrad_body<6, TN>(p, smem, qz);
}

struct CtwPlan {
    int ok, tn, tiles_m, tiles_n, splits, kps, total_ks;
};

static CtwPlan plan_convt_wgrad(int n, int h, int w, int cin, int cout) {
    CtwPlan pl = {};
    if (!mfma_split() || tune(TUNE_CONVT_PATCH) == 0) return pl;
    if (w % 16 != 0 || cout % 64 != 0 || cin % 64 != 0) return pl;
    if (16.0 * n * h * (double)w * cout >= 4294967040.0 || 4.0 * n * h * (double)w * cin >= 4294967040.0) return pl;
    pl.tn = cin % 256 == 0 ? 4 : cin % 128 == 0 ? 2 : 1;
    pl.tiles_m = cout / 64;
    pl.tiles_n = cin / (64 * pl.tn);
    const long total = (long)n * h * (w / 16);
    if (total >= (1L << 30)) return pl;
    pl.total_ks = (int)total;
    // one 8-wave block per CU: aim at ~256 blocks (more splits = more slab traffic), at most 128 K-steps (2048 pixels) per
    // block so that the single accumulator sums a bounded number of products (rd_mfma_dev.h: hi / lo discussion)
    const int tiles = pl.tiles_m * pl.tiles_n;
    long splits = cdiv(pl.tn == 1 ? 512 : 256, tiles);      // TN = 1: 80 KB of LDS, 123 VGPRs -- two blocks per CU
    if (splits < cdiv(total, 128)) splits = cdiv(total, 128);
    if (splits > total) splits = total;
    pl.kps = (int)cdiv(total, splits);
    pl.splits = (int)cdiv(total, pl.kps);
    pl.ok = 1;
    return pl;
}

size_t convt_wgrad_ws_bytes(int n, int h, int w, int cin, int cout) {
    const CtwPlan pl = plan_convt_wgrad(n, h, w, cin, cout);
    return pl.ok ? (size_t)pl.splits * 4 * cout * cin * sizeof(float) : 0;
}

// *splits_out = 0: shape left to the generic TN kernel; else the slab [splits][4 * cout][cin] was written
int convt_wgrad_launch(const float* x, const float* dout, float* slab, int n, int h, int w, int cin, int cout, hipStream_t s,
                       int* splits_out, const unsigned* x_amax, const unsigned* dout_amax) {
    *splits_out = 0;
    const CtwPlan pl = plan_convt_wgrad(n, h, w, cin, cout);
    if (!pl.ok) return RD_OK;
    CtwParams q = {};
    q.dout = dout; q.x = x; q.slab = slab;
    q.Cd = cout; q.Cin = cin; q.W = w; q.spr = w / 16;
    q.total_ks = pl.total_ks; q.kps = pl.kps;
    q.tiles_n = pl.tiles_n; q.tiles_mn = pl.tiles_m * pl.tiles_n;
    q.a_bytes = (unsigned)(16.0 * n * h * (double)w * cout);
    q.b_bytes = (unsigned)(4.0 * n * h * (double)w * cin);
    if (mfma_products() == 3 && x_amax && dout_amax) { q.a_amax = dout_amax; q.b_amax = x_amax; }
    const double px = (double)n * h * w;
    char pcls[64];
    snprintf(pcls, sizeof(pcls), "convt2x2_wgrad|convt_wgrad<%d>", pl.tn);
    ProfScope ps(s, pcls, 2.0 * 4.0 * cout * cin * px, 4.0 * px * (4.0 * cout + cin) + 4.0 * pl.splits * 4.0 * cout * cin, true);
    const dim3 grid((unsigned)(q.tiles_mn * pl.splits));
    if (pl.tn == 4) RD_LAUNCH(convt_wgrad_kernel<4>, grid, dim3(512), 0, s, q);
    else if (pl.tn == 2) RD_LAUNCH(convt_wgrad_kernel<2>, grid, dim3(512), 0, s, q);
    else RD_LAUNCH(convt_wgrad_kernel<1>, grid, dim3(512), 0, s, q);
    RD_LAUNCH_CHECK("convt_wgrad");
    *splits_out = pl.splits;
    return RD_OK;
}

// The patch kernel handles Cin % 32 == 0, Cout % 64 == 0, W a power of two >= 8; everything else stays on the generic NT
// kernel (rd_igemm.hip).  Returns 0 when it did not launch.
int convt_fwd_launch(const float* x, const void* wsplit, size_t wsplit_bytes, const float* bias, const float* skip,
                     const float* sk_mean, const float* sk_invstd, const float* sk_gamma, const float* sk_beta, float sk_slope,
                     const float* sk_slope_dev, float* out, int n, int h, int w, int cin, int cout, hipStream_t s, int* launched,
                     const QuantArgs& qa) {
    *launched = 0;
    if (!mfma_split() || tune(TUNE_CONVT_PATCH) == 0) return RD_OK;
    if (cin % 32 != 0 || cout % 64 != 0 || (w != 8 && w % 16 != 0)) return RD_OK;     // nk = Cin/16 even; 16- (or 8-) pixel patch rows
    const double xb = 4.0 * n * h * (double)w * cin;
    if (xb >= 4294967040.0 || (double)wsplit_bytes >= 4294967040.0) return RD_OK;
    const long G = (long)n * h;
    const long M = G * w;
    // 128-row tiles; 64-row tiles (151 VGPRs, three waves per SIMD) where the 128-row grid would not fill the chip twice (the
    // 8^2 -> 16^2 level: 0.038 -> 0.032 ms).  Everywhere else they measured +3 % alone but -0.3 % end to end (r03 notes);
    // convt_patch = 4 forces them wherever Cin % 64 == 0
    const long blocks128 = ((G + 128 / (w < 16 ? w : 16) - 1) / (128 / (w < 16 ? w : 16))) * (w / (w < 16 ? w : 16)) * (4L * cout / 128);
    int tm = (cin % 64 == 0 && (tune(TUNE_CONVT_PATCH) == 4 || blocks128 < 512)) ? 2 : 4;
    // per-image magnitude slots: a tile must lie inside one image, and whether it does must follow from the layer's SHAPE alone
    // (never from the batch size that picks the tile height above): images too small for 128-row tiles take the 64-row ones
    const int pw_ = w < 16 ? w : 16;
    if (qa.img_stride && h % (128 / pw_) != 0 && cin % 64 == 0) tm = 2;
    CtParams p = {};
    p.x = x; p.wsplit = wsplit; p.bias = bias; p.skip = skip; p.out = out;
    p.sk_mean = sk_mean; p.sk_invstd = sk_invstd; p.sk_gamma = sk_gamma; p.sk_beta = sk_beta; p.sk_slope = sk_slope;
    p.sk_slope_dev = sk_slope_dev;
    p.G = (int)G; p.W = w; p.Cin = cin; p.Cout = cout;
    p.PW = w < 16 ? w : 16;
    p.logPW = ilog2_exact(p.PW);
    p.TR = 32 * tm / p.PW;
    p.tiles_x = w / p.PW;
    p.ngroups = 4 * cout / 128;
    p.nk = cin / 16;
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wsplit_bytes;
    p.wsplit3 = (const char*)wsplit + wsplit_bytes;                  // the two-term form follows the three-term form
    p.w_bytes3 = (unsigned)(wsplit_bytes / SROWB * SROWB3);
    p.out_amax = qa.out;
    if (mfma_products() == 3 && qa.a && qa.b) { p.a_amax = qa.a; p.b_amax = qa.b; }
    if (qa.img_stride) {
        // per-image slots: a tile (TR stacked rows) must lie inside one image; otherwise the launch goes without (six products,
        // nothing committed -- the untouched slots read as "unknown" downstream)
        if (h % p.TR == 0) { p.amax_img_stride = qa.img_stride; p.img_rows = h; }
        else { p.a_amax = p.b_amax = nullptr; p.out_amax = nullptr; }
    }
    const long tiles_y = (G + p.TR - 1) / p.TR;
    const long grid = tiles_y * p.tiles_x * p.ngroups;
    if (grid >= (1L << 31)) return RD_OK;
    char pcls[64];
    snprintf(pcls, sizeof(pcls), "convt2x2_fwd|convt_fwd<%d>", tm);
    ProfScope ps(s, pcls, 2.0 * M * 4.0 * cout * cin,
                 4.0 * ((double)M * cin + 4.0 * cout * cin + (skip ? 2.0 : 1.0) * 4.0 * M * cout), true);
    p.direct = tune(TUNE_NT_EPI) != 0;
    if (tm == 2) RD_LAUNCH(convt_fwd_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, p);
    else RD_LAUNCH(convt_fwd_kernel<4>, dim3((unsigned)grid), dim3(256), 0, s, p);
    RD_LAUNCH_CHECK("convt_fwd");
    *launched = 1;
    return RD_OK;
}

}  // namespace rd
