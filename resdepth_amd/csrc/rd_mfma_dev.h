// Device-side helpers shared by the MFMA kernel translation units (rd_igemm.hip, rd_wgrad_strip.hip).
#pragma once
#include "rd_common.h"

namespace rd {

__device__ __forceinline__ unsigned fd_div(unsigned n, FastDiv f) { return f.sh < 0 ? n : (__umulhi(n, f.mul) >> f.sh); }
// pixel index m of an [img][H][W] grid (m < 2^31) -> img, y, x
__device__ __forceinline__ void pix_split(int m, const PixDiv& d, int& img, int& y, int& x) {
    img = (int)fd_div((unsigned)m, d.hw);
    const int rem = m - img * d.HW;
    y = (int)fd_div((unsigned)rem, d.w);
    x = rem - y * d.W;
}

typedef int v4i32 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0xFFFFFF00u;  // voffset beyond any descriptor extent: the hardware returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}

__device__ __forceinline__ int xcd_remap(int b, int nb) {
    // blocks are dispatched round-robin over the 8 XCDs; give every XCD a contiguous range of
    // logical tiles so neighbouring tiles (which share A rows / B panels) share one L2.
    const int q = nb >> 3, r = nb & 7, x = b & 7, within = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + within;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int SK = 16;      // k-values per K-step
constexpr int SROWB = 96;   // bytes of one (row, K-step) in the packed split-B tensor, six-product form (3 bf16 terms)
constexpr int SROWB3 = 64;  // ... three-product form (2 fp16 terms), which follows the six-product form in the packed buffer

// =====================================================================================================================
//  Two arithmetic forms of the same fp32 GEMM, chosen PER LAUNCH ON THE DEVICE (one library, template parameter NP):
//
//  NP = 6  "split3": x = x1 + x2 + x3 exactly (three bf16 terms by truncation), six products on v_mfma_f32_32x32x16_bf16.
//          No assumption about the operands: the form every kernel falls back to.
//  NP = 3  "split2h" (r06): x ~ (x1 + x2) / s with x1 = rn16(s x), x2 = rn16(s x - x1) two FP16 terms (11 significant bits
//          each: |s x - x1 - x2| <= 2^-22 |s x|, and <= 2^-25 absolute once x2 is an fp16 subnormal), s = a power of two per
//          operand TENSOR that puts the tensor's largest magnitude into [2^14, 2^15); three products a1 b1 (hi), a1 b2,
//          a2 b1 (lo) on v_mfma_f32_32x32x16_f16 -- half the matrix instructions -- and the result scaled back by
//          2^-(ea + eb) (v_ldexp: exact).  Per product: |ab - (a1 b1 + a1 b2 + a2 b1)/(sa sb)| <= 3 * 2^-22 |ab| for
//          elements within 2^-18 of their tensor's maximum (64 x tighter than two bf16 terms), and <= 2^-39 amax(a) |b| below.
//          It needs the tensor maxima: producers max-accumulate |x| into a 16-word device slot from their epilogues
//          (amax_commit; an integer max of IEEE bit patterns is order-independent, so results stay run-to-run identical) and
//          the consumer reads them at kernel start (quant_select).  An operand without a slot, or whose maximum is Inf (NaN
//          elements are fine: they stay NaN in both terms), sends the launch to the NP = 6 body -- which therefore keeps
//          fp32's non-finite semantics.
// =====================================================================================================================
// one slot = 16 words, each on a 128-byte line of its own (2 KB): a block commits to word (blockIdx & 15).  Device-scope
// atomics serialise per LINE at ~11 ns each (MI355X_MICROARCH.md "fanin"; measured here: 2048 commits into one line = 22 us
// behind a 14 us element-wise kernel), sixteen lines take them in parallel
constexpr int AMAX_WORDS = 16, AMAX_STRIDE = 32;

template <int NP> struct frag_of { typedef bf16x8 type; };
template <> struct frag_of<3> { typedef f16x8 type; };
template <int NP>
__device__ __forceinline__ f32x16 mfma16(typename frag_of<NP>::type a, typename frag_of<NP>::type b, f32x16 c) {
    if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// what a launch was given / decided about its operands (wave-uniform, scalar registers)
struct Quant {
    int use3;       // 1: NP = 3 body
    float sa, sb;   // operand scales 2^ea, 2^eb (sb only where B is split on the fly: the weight-gradient kernels)
    int dexp;       // -(ea + eb): exponent of the result's scale-back
};
// per-thread form (pack kernels whose threads serve different tensors)
__device__ __forceinline__ unsigned amax_read(const unsigned* __restrict__ slot) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < AMAX_WORDS; ++i) {
        const unsigned v = slot[i * AMAX_STRIDE];
        m = v > m ? v : m;
    }
    return m;
}
// wave-uniform form: lanes 0..15 fetch one word each (ONE load instruction, sixteen lines in flight), then a cross-lane maximum
__device__ __forceinline__ unsigned amax_read_wave(const unsigned* __restrict__ slot) {
    const int lane = threadIdx.x & 63;
    unsigned m = lane < AMAX_WORDS ? slot[lane * AMAX_STRIDE] : 0u;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)m, o);
        m = other > m ? other : m;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
// biased exponent of the scale that maps a tensor maximum with bits `am` into [2^14, 2^15) (clamped to a finite float)
__device__ __forceinline__ int scale_bexp(unsigned am) {
    const int b = 268 - (int)(am >> 23);
    return b > 254 ? 254 : b;
}
__device__ __forceinline__ Quant quant_select(const unsigned* a_amax, const unsigned* b_amax) {
    Quant q = {0, 1.f, 1.f, 0};
    if (a_amax == nullptr || b_amax == nullptr) return q;
    const unsigned am = amax_read_wave(a_amax), bm = amax_read_wave(b_amax);
    if (am >= 0x7f800000u || bm >= 0x7f800000u) return q;        // an infinite element: exact non-finite semantics live in NP = 6
    if (am == 0u || bm == 0u) return q;                          // a slot nobody wrote (or an all-zero operand): magnitude unknown
    const int ea = scale_bexp(am), eb = scale_bexp(bm);
    q.use3 = 1;
    q.sa = __uint_as_float((unsigned)ea << 23);
    q.sb = __uint_as_float((unsigned)eb << 23);
    q.dexp = 254 - ea - eb;
    return q;
}
// producer side: m = max |v| over what this lane wrote (>= 0; a NaN element is ignored by v_max, see above) -> slot
__device__ __forceinline__ float amax_acc(float m, float v) {
    float r;
    asm("v_max_f32_e64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(v));
    return r;
}
// EVERY thread of the block calls this (block-uniform control flow): wave maximum by cross-lane exchange, block maximum through
// 8 words of LDS, then ONE agent-scope atomic per block.  Measured r06: device-scope atomics on one 128-byte line retire at
// ~11 ns each whatever word they hit (MI355X_MICROARCH.md "fanin"), so one per WAVE of an 8192-block element-wise kernel cost
// 0.23 ms per launch (bn_act_pool_fwd 0.27 -> 1.44 ms per step) -- hence one per block, spread over the slot's sixteen lines.
__device__ __forceinline__ void amax_commit(unsigned* slot, float m) {
    __shared__ unsigned amax_red[16];
    unsigned b = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)b, o);
        b = other > b ? other : b;
    }
    const int nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) b = amax_red[w] > b ? amax_red[w] : b;
        __hip_atomic_fetch_max(slot + (blockIdx.x & (AMAX_WORDS - 1)) * AMAX_STRIDE, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- NP = 6: the six products of a*b = (a1+a2+a3)(b1+b2+b3) that are kept, smallest first.  The five products of weight <= 2^-8
// go into a SEPARATE accumulator (lo) and only a1*b1 into the main one (hi); the two are added once in the epilogue.
// With a single accumulator a 2^-16-class group sum (8 products) that is below half an ulp of a large running sum is
// rounded away EVERY time -- a one-sided loss of up to 2^-15 |ab| on same-sign data (measured 163 u rms at K = 4608,
// u = 2^-24; scripts/split_numerics.py) -- whereas `lo` only ever holds terms of its own size.  `hi` then behaves like an
// fp32 dot product with one rounding per 16 products.  NP = 3 keeps entries 3..5 of the table: a2 b1, a1 b2 (lo), a1 b1 (hi).
constexpr int PA6[6] = {2, 1, 0, 1, 0, 0}, PB6[6] = {0, 1, 2, 0, 1, 0};
template <int NP> constexpr int lo0() { return NP == 3 ? 3 : 0; }      // first kept entry of PA6 / PB6
// hi + lo.  NP = 6: an infinite operand lives in its first term only (split3), so hi = Inf * b1 carries the correct +-Inf (or
// NaN for Inf * 0 / Inf - Inf, as in fp32) while lo may have picked up Inf * 0 = NaN from a ZERO lower term of the other
// operand: an infinite hi therefore wins.  (Only difference to an fp32 product left: Inf * b with 0 < |b| < 2^-133.)
// NP = 3: no infinite operand reaches this form (quant_select); the sum is scaled back by 2^dexp.
__device__ __forceinline__ float merge_hi_lo(float hi, float lo) { return __builtin_fabsf(hi) == __builtin_inff() ? hi : hi + lo; }
template <int NP>
__device__ __forceinline__ float merge_q(float hi, float lo, int dexp) {
    if constexpr (NP == 3) return __builtin_amdgcn_ldexpf(hi + lo, dexp);
    else return merge_hi_lo(hi, lo);
}
template <int NP>
__device__ __forceinline__ float scale_q(float v, int dexp) {       // single-accumulator kernels (weight gradients)
    if constexpr (NP == 3) return __builtin_amdgcn_ldexpf(v, dexp);
    else return v;
}

// x = h + m + l exactly (|x| >= 2^-110; below that the third term is a bf16 subnormal and absorbs an absolute error
// <= 2^-133); h, m, l have <= 8 significant bits (bf16-representable), returned as fp32 bit patterns.  Split by
// truncation: m and l carry the sign of x, |m| < 2^-7 |x|, |l| < 2^-15 |x|.
// Non-finite x: h = x and the residual x - h (Inf - Inf = NaN) is replaced by 0, so +-Inf stays one exact term (h) and
// propagates through the products exactly like in an fp32 multiply (Inf * 0 = NaN, Inf + -Inf = NaN); NaN stays NaN.
// GUARD = false drops the non-finite handling (2 VALU per element): used by the weight-gradient staging, where every
// VALU instruction costs ~3 cycles of MFMA time (scripts/ubench/mfma_shadow.hip) and the single accumulator could not
// keep an Inf apart from Inf * 0 anyway -- there a non-finite operand yields NaN in every output it touches.
template <bool GUARD = true>
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xffff0000u;
    float r = x - __uint_as_float(h);
    if (GUARD) r = (r == r) ? r : 0.f;
    m = __float_as_uint(r) & 0xffff0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}
// NP = 3: two values -> {rn16(s x1) : rn16(s x0)} and the fp16 of the two remainders, 2 VALU per element: v_fma_mixlo/hi_f16
// evaluate fma(x, s, -h) with ONE rounding, to fp16 (s x is exact, so is s x - h: the remainder of an 11-bit rounding of a
// 24-bit value has <= 13 significant bits), and write one half of the destination each -- scale, convert, subtract and pack in
// four instructions (the compiler's own lowering of the C expression takes seven).
__device__ __forceinline__ void split2h_pair(float x0, float x1, float s, unsigned& ph, unsigned& pm) {
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(ph) : "v"(x0), "v"(x1), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(pm)
        : "v"(x0), "v"(x1), "s"(s), "v"(ph));
}
// the same with the scale in a vector register (pack kernels: threads of one wave may serve different tensors)
__device__ __forceinline__ void split2h_pair_v(float x0, float x1, float s, unsigned& ph, unsigned& pm) {
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(ph) : "v"(x0), "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(pm)
        : "v"(x0), "v"(x1), "v"(s), "v"(ph));
}
// four values (consecutive k) -> 8-byte groups of 4 x 16 bit, one per term; NP = 6: v_perm_b32 -> {hi16(odd), hi16(even)}.
// `s`: the operand's scale (NP = 3 only).  kterm3<NP>: the third term exists.
template <int NP> constexpr bool kterm3() { return NP != 3; }
template <int NP, bool GUARD = true>
__device__ __forceinline__ void split_pack4v(float x0, float x1, float x2, float x3, float s, uint2& ph, uint2& pm, uint2& pl) {
    if constexpr (NP == 3) {
        split2h_pair(x0, x1, s, ph.x, pm.x);
        split2h_pair(x2, x3, s, ph.y, pm.y);
        pl = make_uint2(0u, 0u);
    } else {
        unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3<GUARD>(x0, h0, m0, l0);
        split3<GUARD>(x1, h1, m1, l1);
        split3<GUARD>(x2, h2, m2, l2);
        split3<GUARD>(x3, h3, m3, l3);
        ph = make_uint2(__builtin_amdgcn_perm(h1, h0, 0x07060302u), __builtin_amdgcn_perm(h3, h2, 0x07060302u));
        pm = make_uint2(__builtin_amdgcn_perm(m1, m0, 0x07060302u), __builtin_amdgcn_perm(m3, m2, 0x07060302u));
        pl = make_uint2(__builtin_amdgcn_perm(l1, l0, 0x07060302u), __builtin_amdgcn_perm(l3, l2, 0x07060302u));
    }
}
template <int NP>
__device__ __forceinline__ void split_pack4(const float4 v, float s, uint2& ph, uint2& pm, uint2& pl) {
    split_pack4v<NP, true>(v.x, v.y, v.z, v.w, s, ph, pm, pl);
}
__device__ __forceinline__ uint4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
}


}  // namespace rd
