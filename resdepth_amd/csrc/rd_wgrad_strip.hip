// conv3x3 weight gradient, strip kernel (split-bf16 MFMA).  Own translation unit: it is built WITHOUT
// -amdgpu-mfma-vgpr-form -- its 9 accumulators (144 registers) live in AGPRs, the MFMA's native C/D file, while the
// other kernels (<= 64 accumulator registers) are faster with everything in arch VGPRs (see build.sh).
#include <stdlib.h>

#include "rd_common.h"
#include "rd_mfma_dev.h"

namespace rd {

// dW[co][tap][ci] = sum_{img,y,x} dz[img,y,x,co] * X[img,y+dy,x+dx,ci].  The TN kernel above stages the x tile once
// per tap and the dz tile once per 128 output columns; here one block owns (128 co) x (all 9 taps) x (32 ci) and walks
// down a 16-pixel-wide column strip of one image, one image row (= one K-step of 16 pixels) at a time:
//   * dz row y            -> LDS once (double-buffered), shared by the 9 taps;
//   * x halo row y+1      -> LDS once per dx in {-1,0,+1} (the bf16 fragments need 16-byte aligned k, so the three
//                            horizontal shifts are three staged images; the vertical shifts are ring slots), kept in a
//                            4-row ring and used by the three rows y-1, y, y+1 that need it;
//   * wave w owns output-channel block w (32 co): 9 accumulators (one per tap), 54 MFMAs per K-step.
// Staging traffic per MFMA is ~2.5x lower than in the TN kernel.  Same register-transpose staging tasks (4 px x 4 ch)
// and swizzled 128-byte LDS rows as wgrad_tn_split_kernel.
struct WsParams {
    const float* dz;
    const float* x;
    float* slab;        // [splits][M][N]
    int Cin, Cout, M, N;
    int H, W;
    int strips_x, chunks_y, rows_per_chunk, reps;
    int tiles_ci, tiles_mn;
    unsigned a_bytes, b_bytes;
    const unsigned* a_amax;   // magnitude slots of dz / x (both non-null: three-product body, rd_mfma_dev.h)
    const unsigned* b_amax;
    int n_img;          // W8 form: images in the batch (a strip is a PAIR of 8 x 8 images; the last pair may lack its second one)
};

// ---- the same strip schedule with the gfx950 transpose read (ds_read_b64_tr_b16) ------------------------------------------
// wgrad_strip_kernel transposes pixel-major data into k-contiguous LDS rows in registers, which forces the x halo row to be
// staged (and split) once per dx: the bf16 fragments need 16-byte aligned k, so a one-pixel shift is another image.  Here both
// operands are stored in their NATURAL [pixel][term][channel] order (straight 8-byte copies of split float4s) and the
// transposition happens in the fragment read: a 16-lane group of ds_read_b64_tr_b16 reads a [4 pixels][16 channels] bf16 block
// (lane i: 8 bytes at its own address = pixel i >> 2, channels 4 (i & 3) ..) and returns column i to lane i -- two reads give
// the 8 consecutive k (pixels) of a lane's MFMA row.  A horizontal tap shift is then just a pixel-row offset:
//   * the x halo row (18 pixels x 32 ci) is staged ONCE instead of three times: 2624 instead of 3584 split elements per
//     K-step (54 MFMAs per wave), and every staging load is a coalesced run (512 B per dz pixel, 128 B per x pixel);
//   * LDS: pixel strides of 832 B (dz, 64 B pad) and 192 B (x) put the four pixel rows of a transpose read 16 banks apart --
//     the two 16-lane groups of a 32-lane half then cover all 64 banks; 40 KB per block instead of 80.
typedef short v4s16 __attribute__((ext_vector_type(4)));
template <typename FR>
__device__ __forceinline__ FR lds_tr2(const float* smem_base, int byte_off, int second) {
    // two transpose reads (k = 0..3 and 4..7 of the lane's octet) -> one 8 x bf16 MFMA operand
    typedef __attribute__((address_space(3))) v4s16* lds_p;
    const char* b = reinterpret_cast<const char*>(smem_base) + byte_off;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b));
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(b + second));
    typedef short v8s16 __attribute__((ext_vector_type(8)));
    const v8s16 r = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(FR, r);
}

// NCO x NCI = 4 waves: wave (cw, cb) owns output channels [32 cw, +32) x input channels [32 cb, +32) of the block tile.
//   <4, 1>: 128 co x 32 ci (the r02 tile): 2048 + 576 = 2624 staged elements per K-step
//   <2, 2>:  64 co x 64 ci               : 1024 + 1152 = 2176 (-17 %), needs Cin % 64 == 0
//
// W8 (r04): 8 x 8 images (the bottleneck level).  A "strip" is a PAIR of images side by side: row y of the strip = row y of both
// images, so a K-step is still 16 pixels (k 0-7: first image, 8-15: second).  Each image brings its own zero border: the halo row
// has 2 x 10 pixels, and the fragment reads of the second image's pixels start two halo pixels later (a per-lane constant, as in
// the W8 form of conv3_halo_split_kernel) -- the tap shifts stay compile-time row offsets.  Replaces the generic TN kernel for this
// level (two register transposes per operand, 7 VALU per MFMA: 99 TFLOP/s).
template <int NP, int NCO, int NCI, bool W8>
__device__ __forceinline__ void wgrad_strip_tr_body(const WsParams& p, float* smem, const Quant qz) {
    typedef typename frag_of<NP>::type FR;
    constexpr int NT = NP == 3 ? 2 : 3;
    static_assert(NCO * NCI == 4, "four waves");
    constexpr int TA = 64 * NCO, TB = 64 * NCI;                // bytes of one term of one pixel (32 channels x 2 B per wave column)
    // pixel strides: 3 terms + padding so that (stride / 4) mod 64 is 16 or 48 -- the four pixel rows of a transpose read then
    // sit 16 banks apart and the two 16-lane groups of a 32-lane half cover all 64 banks
    constexpr int APX = 3 * TA + (NCO == 4 ? 64 : NCO == 2 ? 64 : 0), BPX = 3 * TB + (NCI == 2 ? 64 : 0);
    static_assert((APX / 4) % 64 == 16 || (APX / 4) % 64 == 48, "dz pixel stride");
    static_assert((BPX / 4) % 64 == 16 || (BPX / 4) % 64 == 48, "x pixel stride");
    constexpr int HPX = W8 ? 20 : 18;                          // halo pixels of an x row (W8: two images with their own borders)
    constexpr int ASTAGE = 16 * APX, BSLOT = HPX * BPX;        // bytes per dz stage / per halo ring slot
    constexpr int RINGB = 2 * ASTAGE;                          // byte offset of the halo ring
    constexpr int AQ = 8 * NCO, BQ = 8 * NCI;                  // channel quads per pixel
    constexpr int AI = 16 * AQ / 256, BI = (HPX * BQ + 255) / 256;     // staging items (float4) per thread

    const int gb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = gb / p.tiles_mn;
    const int lb = gb - split * p.tiles_mn;
    const int tile_ci = lb % p.tiles_ci, tile_m = lb / p.tiles_ci;
    const int m0 = tile_m * (32 * NCO), ci0 = tile_ci * (32 * NCI);
    int img = 0, x0 = 0, ya = 0, yb = 0;
    const int H = p.H, W = p.W;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cw = wave / NCI, cb = wave % NCI;

    // staging items (one float4 = 4 channels of one pixel): dz row 16 px x AQ quads, x halo row 18 px x BQ quads
    int a_wr[AI], b_wr[BI];
    int a_px[AI], a_ch[AI], b_hp[BI], b_ch[BI];
    bool b_act[BI];
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int e = t + 256 * k, cq = e % AQ, px = e / AQ;
        a_px[k] = px;
        a_ch[k] = m0 + cq * 4 < p.Cout ? m0 + cq * 4 : -1;
        a_wr[k] = px * APX + cq * 8;
    }
#pragma unroll
    for (int k = 0; k < BI; ++k) {
        const int e = t + 256 * k, cq = e % BQ, hp = e / BQ;
        b_act[k] = e < HPX * BQ;
        b_hp[k] = hp;
        b_ch[k] = (b_act[k] && ci0 + cq * 4 < p.Cin) ? ci0 + cq * 4 : -1;
        b_wr[k] = hp * BPX + cq * 8;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.dz, p.a_bytes), rsB = make_rsrc(p.x, p.b_bytes);
    const unsigned rowA = (unsigned)W * p.Cout * 4, rowB = (unsigned)W * p.Cin * 4;
    unsigned baseA = 0, baseB = 0, voffA[AI], voffB[BI];
    auto set_strip = [&](int sid) {
        const int sx = sid % p.strips_x;
        const int cy = (sid / p.strips_x) % p.chunks_y;
        img = sid / (p.strips_x * p.chunks_y);
        x0 = sx * 16;
        ya = cy * p.rows_per_chunk;
        yb = ya + p.rows_per_chunk;
        if (W8) img *= 2;                                       // first image of the pair
        baseA = (unsigned)img * H * rowA;
        baseB = (unsigned)img * H * rowB;
        if (W8) {
            const bool two = img + 1 < p.n_img;
#pragma unroll
            for (int k = 0; k < AI; ++k) {
                const int sub = a_px[k] >> 3, xx = a_px[k] & 7;
                voffA[k] = (a_ch[k] >= 0 && (sub == 0 || two)) ? (unsigned)(((sub * 64 + xx) * p.Cout + a_ch[k]) * 4) : kOOB;
            }
#pragma unroll
            for (int k = 0; k < BI; ++k) {
                const int sub = b_hp[k] >= 10 ? 1 : 0, px = b_hp[k] - 10 * sub - 1;
                voffB[k] = (b_ch[k] >= 0 && (unsigned)px < 8u && (sub == 0 || two)) ? (unsigned)(((sub * 64 + px) * p.Cin + b_ch[k]) * 4) : kOOB;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < AI; ++k) voffA[k] = a_ch[k] >= 0 ? (unsigned)(((x0 + a_px[k]) * p.Cout + a_ch[k]) * 4) : kOOB;
#pragma unroll
        for (int k = 0; k < BI; ++k) {
            const int px = x0 - 1 + b_hp[k];
            voffB[k] = (b_ch[k] >= 0 && (unsigned)px < (unsigned)W) ? (unsigned)((px * p.Cin + b_ch[k]) * 4) : kOOB;
        }
    };
    // virtual step yy stages T(yy) = { dz row yy+1, x halo row yy+2 } and multiplies row yy
    auto load_task = [&](int yy, float4 (&v)[AI + BI]) {
        const int ra = yy + 1, rb = yy + 2;
        const bool oka = ra >= ya && ra < yb, okb = rb >= 0 && rb < H && rb <= yb;
        const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(oka ? (int)(baseA + (unsigned)ra * rowA) : 0);
        const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane(okb ? (int)(baseB + (unsigned)rb * rowB) : 0);
#pragma unroll
        for (int k = 0; k < AI; ++k) v[k] = buf_load4(rsA, oka ? voffA[k] : kOOB, sa);
#pragma unroll
        for (int k = 0; k < BI; ++k) v[AI + k] = buf_load4(rsB, okb ? voffB[k] : kOOB, sb);
    };
    // float4 (4 channels of a pixel) -> three 8-byte pieces (one per term) at dst, dst + term stride, dst + 2 term strides
    auto put = [&](char* dst, int tstride, const float4 x, float qs) {
        uint2 ph, pm, pl;
        split_pack4v<NP, false>(x.x, x.y, x.z, x.w, qs, ph, pm, pl);
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + tstride) = pm;
        if (kterm3<NP>()) *reinterpret_cast<uint2*>(dst + 2 * tstride) = pl;
    };
    auto store_task = [&](int yy, const float4 (&v)[AI + BI]) {
        char* sa = reinterpret_cast<char*>(smem) + ((yy + 1) & 1) * ASTAGE;
#pragma unroll
        for (int k = 0; k < AI; ++k) put(sa + a_wr[k], TA, v[k], qz.sa);
        char* sb = reinterpret_cast<char*>(smem) + RINGB + ((yy + 3) & 3) * BSLOT;
#pragma unroll
        for (int k = 0; k < BI; ++k)
            if (b_act[k]) put(sb + b_wr[k], TB, v[AI + k], qz.sb);
    };

    // fragment addressing: lane -> 16-lane group g (n-block nb = g & 1, k-octet h = g >> 1), i = lane & 15
    const int li = lane & 15, lg = lane >> 4, nb = lg & 1, hk = lg >> 1;
    const int a_rd = (8 * hk + (li >> 2)) * APX + (32 * cw + 16 * nb + 4 * (li & 3)) * 2;
    const int b_rd = (8 * hk + (li >> 2) + (W8 ? 2 * hk : 0)) * BPX + (32 * cb + 16 * nb + 4 * (li & 3)) * 2;   // W8: second image's halo
    const int lrow = lane & 31, half = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    auto mma_row = [&](int y) {
        const int a_stage = (y & 1) * ASTAGE;
        FR af[3], bf[2][3][3];
        auto read_b = [&](int dy, FR (&dst)[3][3]) {
            const int slot = RINGB + ((y + dy) & 3) * BSLOT;    // image row y+dy-1 lives in slot (row+1)&3
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int q = 0; q < NT; ++q) dst[d][q] = lds_tr2<FR>(smem, slot + b_rd + d * BPX + q * TB, 4 * BPX);
        };
#pragma unroll
        for (int q = 0; q < NT; ++q) af[q] = lds_tr2<FR>(smem, a_stage + a_rd + q * TA, 4 * APX);
        read_b(0, bf[0]);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (dy < 2) read_b(dy + 1, bf[(dy + 1) & 1]);
#if RD_WG_FENCE
            if constexpr (NP == 3) __builtin_amdgcn_sched_barrier(0);      // the next group's fragment reads stay AHEAD of this group's MFMAs
#endif
#pragma unroll
            for (int t6 = lo0<NP>(); t6 < 6; ++t6)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc[dy * 3 + d] = mfma16<NP>(af[PA[t6]], bf[dy & 1][d][PB[t6]], acc[dy * 3 + d]);
        }
    };

    float4 v0[AI + BI], v1[AI + BI];
    for (int rep = 0; rep < p.reps; ++rep) {
        set_strip(split * p.reps + rep);
        const int ys = ya - 3;                   // three warm-up steps fill the halo ring and the first dz row
        load_task(ys, v0);
        load_task(ys + 1, v1);
        store_task(ys, v0);
        load_task(ys + 2, v0);
        __syncthreads();
        store_task(ys + 1, v1);
        load_task(ys + 3, v1);
        __syncthreads();
        store_task(ys + 2, v0);
        load_task(ys + 4, v0);
        __syncthreads();
        for (int yy = ya; yy < yb; yy += 2) {    // rows_per_chunk is even; no branches around the MFMAs
            store_task(yy, v1);
            load_task(yy + 2, v1);
            mma_row(yy);
            __syncthreads();
            store_task(yy + 1, v0);
            load_task(yy + 3, v0);
            mma_row(yy + 1);
            __syncthreads();
        }
    }

    float* out = p.slab + (long)split * p.M * p.N;
    const int cwu = __builtin_amdgcn_readfirstlane(cw), cbu = __builtin_amdgcn_readfirstlane(cb);
    if (m0 + 32 * NCO <= p.M && (long)(32 * NCO) * p.N * 4 < 0x7fffffffL) {
        // whole tile inside the slab (every shape of the network): buffer stores with the row / tap part of the address in
        // scalar registers and ONE per-lane offset -- 144 stores and no address arithmetic in the vector ALU (r04; the 64-bit
        // per-element addresses below cost ~6 VALU instructions per store beside a sibling wave's MFMAs)
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(out + (long)m0 * p.N + ci0, (unsigned)((long)(32 * NCO) * p.N * 4));
        const unsigned lane_off = (unsigned)(((4 * half) * p.N + lrow) * 4);
        const int rowN = __builtin_amdgcn_readfirstlane(p.N * 4);
        const int base = (cwu * 32) * rowN + cbu * 128;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int tapo = base + tp * p.Cin * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(scale_q<NP>(acc[tp][r], qz.dexp)), rsO, lane_off, (unsigned)(tapo + ((r & 3) + 8 * (r >> 2)) * rowN), 0);
        }
        return;
    }
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int n = tp * p.Cin + ci0 + cb * 32 + lrow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < p.M) out[(long)m * p.N + n] = scale_q<NP>(acc[tp][r], qz.dexp);
        }
    }
}

// ---- three-product form with the x fragments ROTATING through registers (r06) ---------------------------------------------------
// In wgrad_strip_tr_body every K-step (image row y) reads the fragments of the three x halo rows y-1, y, y+1 from LDS: 36 of its
// 40 fragment reads, each row three steps in a row, and each group of reads sits right in front of the MFMAs that need it (hipcc
// sinks them there whatever the source order; the wave then waits out the LDS latency three times per step: MFMA pipe 0.58 at
// 1.96 GHz in profiles/r06_summary.json -- stall-bound, not power-bound like the halo kernel).  With two fp16 terms per operand a
// row's fragments are 24 registers, so two rows stay in registers across steps:
//     step y:   set P = x row y-1, set Q = x row y, af = dz row y     (all read during EARLIER steps)
//               MFMAs of dy = 0 on P  ->  P <- x row y+1, af' <- dz row y+1 (16 reads, behind 18 MFMAs of cover)
//               MFMAs of dy = 1 on Q,  MFMAs of dy = 2 on P,  barrier;  step y+1 runs with (Q, P, af') in the roles of (P, Q, af).
// 16 fragment reads per step instead of 40, none of them waited for.  Staging runs one row further ahead (task T(k) = dz row k+2
// and x halo row k+2), so LDS holds four dz stages and only TWO halo rows (the one being read and the one being written).  Per
// accumulator the MFMAs arrive in the same order as in wgrad_strip_tr_body<3>: bit-identical results.
template <int NCO, int NCI, bool W8>
__device__ __forceinline__ void wgrad_strip_rot_body(const WsParams& p, float* smem, const Quant qz) {
    typedef f16x8 FR;
    static_assert(NCO * NCI == 4, "four waves");
    constexpr int TA = 64 * NCO, TB = 64 * NCI;
    constexpr int APX = 3 * TA + (NCO == 4 ? 64 : NCO == 2 ? 64 : 0), BPX = 3 * TB + (NCI == 2 ? 64 : 0);   // pixel strides of the
    constexpr int HPX = W8 ? 20 : 18;                                                                      // six-product layout: same banks
    constexpr int ASTAGE = 16 * APX, BSLOT = HPX * BPX;
    constexpr int RINGB = 4 * ASTAGE;                          // four dz stages, then two halo-row slots
    constexpr int AQ = 8 * NCO, BQ = 8 * NCI;
    constexpr int AI = 16 * AQ / 256, BI = (HPX * BQ + 255) / 256;

    const int gb = xcd_remap(blockIdx.x, gridDim.x);
    const int split = gb / p.tiles_mn;
    const int lb = gb - split * p.tiles_mn;
    const int tile_ci = lb % p.tiles_ci, tile_m = lb / p.tiles_ci;
    const int m0 = tile_m * (32 * NCO), ci0 = tile_ci * (32 * NCI);
    int img = 0, x0 = 0, ya = 0, yb = 0;
    const int H = p.H, W = p.W;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cw = wave / NCI, cb = wave % NCI;

    int a_wr[AI], b_wr[BI];
    int a_px[AI], a_ch[AI], b_hp[BI], b_ch[BI];
    bool b_act[BI];
#pragma unroll
    for (int k = 0; k < AI; ++k) {
        const int e = t + 256 * k, cq = e % AQ, px = e / AQ;
        a_px[k] = px;
        a_ch[k] = m0 + cq * 4 < p.Cout ? m0 + cq * 4 : -1;
        a_wr[k] = px * APX + cq * 8;
    }
#pragma unroll
    for (int k = 0; k < BI; ++k) {
        const int e = t + 256 * k, cq = e % BQ, hp = e / BQ;
        b_act[k] = e < HPX * BQ;
        b_hp[k] = hp;
        b_ch[k] = (b_act[k] && ci0 + cq * 4 < p.Cin) ? ci0 + cq * 4 : -1;
        b_wr[k] = hp * BPX + cq * 8;
    }
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.dz, p.a_bytes), rsB = make_rsrc(p.x, p.b_bytes);
    const unsigned rowA = (unsigned)W * p.Cout * 4, rowB = (unsigned)W * p.Cin * 4;
    unsigned baseA = 0, baseB = 0, voffA[AI], voffB[BI];
    auto set_strip = [&](int sid) {
        const int sx = sid % p.strips_x;
        const int cy = (sid / p.strips_x) % p.chunks_y;
        img = sid / (p.strips_x * p.chunks_y);
        x0 = sx * 16;
        ya = cy * p.rows_per_chunk;
        yb = ya + p.rows_per_chunk;
        if (W8) img *= 2;
        baseA = (unsigned)img * H * rowA;
        baseB = (unsigned)img * H * rowB;
        if (W8) {
            const bool two = img + 1 < p.n_img;
#pragma unroll
            for (int k = 0; k < AI; ++k) {
                const int sub = a_px[k] >> 3, xx = a_px[k] & 7;
                voffA[k] = (a_ch[k] >= 0 && (sub == 0 || two)) ? (unsigned)(((sub * 64 + xx) * p.Cout + a_ch[k]) * 4) : kOOB;
            }
#pragma unroll
            for (int k = 0; k < BI; ++k) {
                const int sub = b_hp[k] >= 10 ? 1 : 0, px = b_hp[k] - 10 * sub - 1;
                voffB[k] = (b_ch[k] >= 0 && (unsigned)px < 8u && (sub == 0 || two)) ? (unsigned)(((sub * 64 + px) * p.Cin + b_ch[k]) * 4) : kOOB;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < AI; ++k) voffA[k] = a_ch[k] >= 0 ? (unsigned)(((x0 + a_px[k]) * p.Cout + a_ch[k]) * 4) : kOOB;
#pragma unroll
        for (int k = 0; k < BI; ++k) {
            const int px = x0 - 1 + b_hp[k];
            voffB[k] = (b_ch[k] >= 0 && (unsigned)px < (unsigned)W) ? (unsigned)((px * p.Cin + b_ch[k]) * 4) : kOOB;
        }
    };
    // task T(k) = { dz row k+2, x halo row k+2 }: loaded two steps before it is stored, stored during step k, read during step k+1
    auto load_task = [&](int k, float4 (&v)[AI + BI]) {
        const int r = k + 2;
        const bool oka = r >= ya && r < yb, okb = r >= 0 && r < H && r <= yb;
        const unsigned sa = (unsigned)__builtin_amdgcn_readfirstlane(oka ? (int)(baseA + (unsigned)r * rowA) : 0);
        const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane(okb ? (int)(baseB + (unsigned)r * rowB) : 0);
#pragma unroll
        for (int i = 0; i < AI; ++i) v[i] = buf_load4(rsA, oka ? voffA[i] : kOOB, sa);
#pragma unroll
        for (int i = 0; i < BI; ++i) v[AI + i] = buf_load4(rsB, okb ? voffB[i] : kOOB, sb);
    };
    auto put = [&](char* dst, int tstride, const float4 x, float qs) {
        uint2 ph, pm, pl;
        split_pack4v<3, false>(x.x, x.y, x.z, x.w, qs, ph, pm, pl);
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + tstride) = pm;
    };
    auto store_task = [&](int k, const float4 (&v)[AI + BI]) {
        char* sa = reinterpret_cast<char*>(smem) + ((k + 2) & 3) * ASTAGE;
#pragma unroll
        for (int i = 0; i < AI; ++i) put(sa + a_wr[i], TA, v[i], qz.sa);
        char* sb = reinterpret_cast<char*>(smem) + RINGB + ((k + 2) & 1) * BSLOT;
#pragma unroll
        for (int i = 0; i < BI; ++i)
            if (b_act[i]) put(sb + b_wr[i], TB, v[AI + i], qz.sb);
    };

    const int li = lane & 15, lg = lane >> 4, nb = lg & 1, hk = lg >> 1;
    const int a_rd = (8 * hk + (li >> 2)) * APX + (32 * cw + 16 * nb + 4 * (li & 3)) * 2;
    const int b_rd = (8 * hk + (li >> 2) + (W8 ? 2 * hk : 0)) * BPX + (32 * cb + 16 * nb + 4 * (li & 3)) * 2;
    const int lrow = lane & 31, half = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    auto read_x = [&](int row, FR (&dst)[3][2]) {              // fragments of x halo row `row` (three horizontal shifts x two terms)
        const int slot = RINGB + (row & 1) * BSLOT;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int q = 0; q < 2; ++q) dst[d][q] = lds_tr2<FR>(smem, slot + b_rd + d * BPX + q * TB, 4 * BPX);
    };
    auto read_dz = [&](int row, FR (&dst)[2]) {
        const int stage = (row & 3) * ASTAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) dst[q] = lds_tr2<FR>(smem, stage + a_rd + q * TA, 4 * APX);
    };
    // the nine MFMAs of one vertical tap row: a2 b1, a1 b2, a1 b1 for the three horizontal shifts (order of wgrad_strip_tr_body<3>)
    auto group = [&](int dy, const FR (&af)[2], const FR (&xf)[3][2]) {
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[dy * 3 + d] = mfma16<3>(af[1], xf[d][0], acc[dy * 3 + d]);
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[dy * 3 + d] = mfma16<3>(af[0], xf[d][1], acc[dy * 3 + d]);
#pragma unroll
        for (int d = 0; d < 3; ++d) acc[dy * 3 + d] = mfma16<3>(af[0], xf[d][0], acc[dy * 3 + d]);
    };
    // one K-step (image row y): P holds x row y-1, Q x row y, af dz row y; leaves x row y+1 in P and dz row y+1 in afn
    auto step = [&](int y, const FR (&af)[2], FR (&afn)[2], FR (&P)[3][2], const FR (&Q)[3][2]) {
        group(0, af, P);
        __builtin_amdgcn_sched_barrier(0);                      // the sixteen reads go out HERE, behind eighteen MFMAs of cover
        read_x(y + 1, P);
        read_dz(y + 1, afn);
        __builtin_amdgcn_sched_barrier(0);
        group(1, af, Q);
        __builtin_amdgcn_sched_barrier(0);                      // (hipcc would pull the a2 products of dy = 2 -- which wait for P -- forward)
        group(2, af, P);
        __builtin_amdgcn_sched_barrier(0);                      // the step's barrier stays BEHIND its MFMAs (hipcc hoists it -- and the
    };                                                          // wait for all sixteen reads with it -- to right behind the reads)

    float4 v0[AI + BI], v1[AI + BI];
    FR xa[3][2], xb[3][2], af0[2], af1[2];
    for (int rep = 0; rep < p.reps; ++rep) {
        set_strip(split * p.reps + rep);
        // warm-up: x rows ya-1, ya and dz row ya into registers, T(ya-1) into LDS, T(ya) / T(ya+1) in flight
        load_task(ya - 3, v0);
        load_task(ya - 2, v1);
        store_task(ya - 3, v0);
        load_task(ya - 1, v0);
        store_task(ya - 2, v1);
        load_task(ya, v1);
        __syncthreads();
        read_x(ya - 1, xa);
        read_x(ya, xb);
        read_dz(ya, af0);
        __syncthreads();                                        // (the compiler waits for the reads before the barrier) slot of row ya-1 is free
        store_task(ya - 1, v0);
        load_task(ya + 1, v0);
        __syncthreads();
        for (int yy = ya; yy < yb; yy += 2) {                   // rows_per_chunk is even
            store_task(yy, v1);
            load_task(yy + 2, v1);
            step(yy, af0, af1, xa, xb);
            __syncthreads();
            store_task(yy + 1, v0);
            load_task(yy + 3, v0);
            step(yy + 1, af1, af0, xb, xa);
            __syncthreads();
        }
    }

    float* out = p.slab + (long)split * p.M * p.N;
    const int cwu = __builtin_amdgcn_readfirstlane(cw), cbu = __builtin_amdgcn_readfirstlane(cb);
    if (m0 + 32 * NCO <= p.M && (long)(32 * NCO) * p.N * 4 < 0x7fffffffL) {
        const __amdgpu_buffer_rsrc_t rsO = make_rsrc(out + (long)m0 * p.N + ci0, (unsigned)((long)(32 * NCO) * p.N * 4));
        const unsigned lane_off = (unsigned)(((4 * half) * p.N + lrow) * 4);
        const int rowN = __builtin_amdgcn_readfirstlane(p.N * 4);
        const int base = (cwu * 32) * rowN + cbu * 128;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int tapo = base + tp * p.Cin * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(scale_q<3>(acc[tp][r], qz.dexp)), rsO, lane_off, (unsigned)(tapo + ((r & 3) + 8 * (r >> 2)) * rowN), 0);
        }
        return;
    }
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int n = tp * p.Cin + ci0 + cb * 32 + lrow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < p.M) out[(long)m * p.N + n] = scale_q<3>(acc[tp][r], qz.dexp);
        }
    }
}

template <int OCC, int NCO, int NCI, bool W8 = false>
__global__ __launch_bounds__(256, OCC) void wgrad_strip_tr_kernel(WsParams p) {
    constexpr int TA = 64 * NCO, TB = 64 * NCI;
    constexpr int APX = 3 * TA + (NCO == 4 ? 64 : NCO == 2 ? 64 : 0), BPX = 3 * TB + (NCI == 2 ? 64 : 0);
    constexpr int HPX = W8 ? 20 : 18, ASTAGE = 16 * APX, BSLOT = HPX * BPX;
    constexpr int SM6 = 2 * ASTAGE + 4 * BSLOT, SM3 = 4 * ASTAGE + 2 * BSLOT;      // six-product ring | rotating three-product form
    __shared__ __attribute__((aligned(16))) float smem[(SM6 > SM3 ? SM6 : SM3) / 4];
    const Quant qz = quant_select(p.a_amax, p.b_amax);
#if RD_WG_NOROT
    if (qz.use3) wgrad_strip_tr_body<3, NCO, NCI, W8>(p, smem, qz);
#else
    if (qz.use3) wgrad_strip_rot_body<NCO, NCI, W8>(p, smem, qz);
#endif
    else wgrad_strip_tr_body<6, NCO, NCI, W8>(p, smem, qz);
}

struct WsPlan {
    int ok, swapped, rows_per_chunk, chunks_y, strips_x, tiles_m, tiles_ci, reps, splits;
    int sq;         // 1: 64 co x 64 ci block tile (transpose-read kernel <2, 2>), 0: 128 co x 32 ci
    int w8;         // 1: 8 x 8 images, strips are image pairs (wgrad_strip_tr_kernel<.., W8>)
};

// The strip kernel needs W >= 16, the shifted operand's channels % 32 == 0 and a full 128-channel block on the other
// side.  When Cout < 128 <= Cin the roles are swapped (pl.swapped): dW[co][tap][ci] = sum_p x[p][ci] * dz[p - tap][co],
// i.e. the same kernel with x as the un-shifted operand, dz as the shifted one and the taps mirrored; the slab is then
// [Cin][(8 - tap) * Cout + co] and the reduction kernel un-mirrors it.
static WsPlan plan_strip(int n, int h, int w, int cin, int cout) {
    WsPlan pl = {};
    const int force = tune(TUNE_WG_STRIP);
    // 8 x 8 images (W8 form of the transpose-read kernel): pairs of images are the strips; square 64 x 64 tiles only
    const bool w8 = w == 8 && h == 8 && cout % 64 == 0 && cin % 64 == 0 && force != 1 && force != 3 && tune(TUNE_WG_STRIP) != 4;
    if (!mfma_split() || force == 0 || (!w8 && w % 16 != 0) || h < 4 || h % 2 != 0) return pl;   // 16-pixel strips, rows in pairs
    // square tile: both channel counts multiples of 64 (every cfg-S / cfg-M layer) -- fewer staged elements per MFMA and no
    // operand swap; wg_strip = 3 keeps the 128 x 32 tile of the transpose-read kernel for A/B runs
    if (force != 1 && force != 3 && cout % 64 == 0 && cin % 64 == 0) {
        pl.sq = 1;
    } else if (cout >= 128 && cout % 4 == 0 && cin % 32 == 0) {
        pl.swapped = 0;
    } else if (cin >= 128 && cin % 4 == 0 && cout % 32 == 0) {
        pl.swapped = 1;
        const int tmp = cin; cin = cout; cout = tmp;
    } else {
        return pl;
    }
    pl.ok = 1;
    pl.w8 = w8;
    if (w8) n = (n + 1) / 2;                      // strips = image pairs
    pl.strips_x = w8 ? 1 : w / 16;
    pl.tiles_m = cdiv(cout, pl.sq ? 64 : 128);
    pl.tiles_ci = cin / (pl.sq ? 64 : 32);
    const long base = (long)pl.tiles_m * pl.tiles_ci * n * pl.strips_x;
    int rows = h;
    const int minb = tune(TUNE_WG_MINBLOCKS);
    while (base * (h / rows) < minb && rows > 32 && rows % 4 == 0) rows >>= 1;   // chunks stay an even number of rows
    pl.rows_per_chunk = rows;
    pl.chunks_y = h / rows;
    // every split costs one [Cout][9*Cin] slab of HBM traffic (written here, read by the reduction): with more than
    // ~2048 blocks, give each block several strips instead
    const long strips = (long)n * pl.strips_x * pl.chunks_y;
    // measured (r03, transpose-read kernel, interleaved end to end): 256 blocks 2747.9 / 2748.5 tiles/s, 384: 2716.7 / 2728.4,
    // 512 (the r02 optimum of the 80 KB kernel): 2715.3 / 2715.1, 768: 2712.3 / 2720.6, 128: 2499 -- half the split-K slabs to
    // write and reduce, and one block per CU leaves room beside the main-stream kernels of the two-stream backward
    const int target = tune(TUNE_WG_BLOCKS);
    int reps = 1;
    while (base * pl.chunks_y / (2 * reps) >= target && strips % (2 * reps) == 0) reps *= 2;
    pl.reps = reps;
    pl.splits = (int)(strips / reps);
    return pl;
}


int wgrad_strip_splits(int n, int h, int w, int cin, int cout) {
    const WsPlan pl = plan_strip(n, h, w, cin, cout);
    return pl.ok ? pl.splits : 0;
}

int wgrad_strip_launch(const float* x, const float* dz, float* slab, int n, int h, int w, int cin, int cout, hipStream_t s,
                       int* splits_out, int* swapped_out, const unsigned* x_amax, const unsigned* dz_amax) {
    const WsPlan wp = plan_strip(n, h, w, cin, cout);
    *splits_out = 0;
    *swapped_out = 0;
    if (!wp.ok) return RD_OK;
    if (wp.swapped) {
        const float* tp = x; x = dz; dz = tp;
        const int tc = cin; cin = cout; cout = tc;
        const unsigned* ta = x_amax; x_amax = dz_amax; dz_amax = ta;
    }
    WsParams q = {};
    q.dz = dz; q.x = x; q.slab = slab;
    q.Cin = cin; q.Cout = cout; q.M = cout; q.N = 9 * cin;
    q.H = h; q.W = w;
    q.strips_x = wp.strips_x; q.chunks_y = wp.chunks_y; q.rows_per_chunk = wp.rows_per_chunk; q.reps = wp.reps;
    q.tiles_ci = wp.tiles_ci; q.tiles_mn = wp.tiles_m * wp.tiles_ci;
    q.n_img = n;
    if (mfma_products() == 3 && x_amax && dz_amax) { q.a_amax = dz_amax; q.b_amax = x_amax; }
    const double ab = 4.0 * n * h * w * (double)cout, bb = 4.0 * n * h * w * (double)cin;
    RD_REQUIRE(ab < 4294967040.0 && bb < 4294967040.0, "rd_conv3x3_bwd_weight: operand beyond the 4 GiB descriptor range");
    q.a_bytes = (unsigned)ab; q.b_bytes = (unsigned)bb;
    const int occ_ = tune(TUNE_WG_OCC) == 2 ? 2 : 1;
    char pcls[64];      // "<operation>|<kernel symbol as rocprofv3 prints it, summarize_prof.py form>"
    if (wp.w8) snprintf(pcls, sizeof(pcls), "conv3x3_wgrad|wgrad_strip_tr<%d,2,2,w8>", occ_);
    else if (wp.sq) snprintf(pcls, sizeof(pcls), "conv3x3_wgrad|wgrad_strip_tr<%d,2,2>", occ_);
    else snprintf(pcls, sizeof(pcls), "conv3x3_wgrad|wgrad_strip_tr<%d,4,1>", occ_);
    ProfScope ps(s, pcls, 2.0 * cout * 9.0 * cin * (double)n * h * w, ab + bb + 4.0 * cout * 9.0 * cin, true);
    // (the r02 register-transpose kernel this replaced -- 2.80 -> 2.69 ms over the nine layers, profiles/r03_notes.md -- left the
    // library in r06)
    const dim3 grid(q.tiles_mn * wp.splits);
    const int occ = tune(TUNE_WG_OCC);
    if (wp.w8 && occ == 2) RD_LAUNCH((wgrad_strip_tr_kernel<2, 2, 2, true>), grid, dim3(256), 0, s, q);
    else if (wp.w8) RD_LAUNCH((wgrad_strip_tr_kernel<1, 2, 2, true>), grid, dim3(256), 0, s, q);
    else if (wp.sq && occ == 2) RD_LAUNCH((wgrad_strip_tr_kernel<2, 2, 2>), grid, dim3(256), 0, s, q);
    else if (wp.sq) RD_LAUNCH((wgrad_strip_tr_kernel<1, 2, 2>), grid, dim3(256), 0, s, q);
    else if (occ == 2) RD_LAUNCH((wgrad_strip_tr_kernel<2, 4, 1>), grid, dim3(256), 0, s, q);
    else RD_LAUNCH((wgrad_strip_tr_kernel<1, 4, 1>), grid, dim3(256), 0, s, q);
    RD_LAUNCH_CHECK("wgrad_strip");
    *splits_out = wp.splits;
    *swapped_out = wp.swapped;
    return RD_OK;
}

}  // namespace rd
