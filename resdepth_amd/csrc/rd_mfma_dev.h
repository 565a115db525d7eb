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
constexpr int SK = 16;      // k-values per K-step
constexpr int SROWB = 96;   // bytes of one (row, K-step) in the packed split-B tensor

// ---- the six products of a*b = (a1+a2+a3)(b1+b2+b3) that are kept, smallest first.  The five products of weight <= 2^-8
// go into a SEPARATE accumulator (lo) and only a1*b1 into the main one (hi); the two are added once in the epilogue.
// With a single accumulator a 2^-16-class group sum (8 products) that is below half an ulp of a large running sum is
// rounded away EVERY time -- a one-sided loss of up to 2^-15 |ab| on same-sign data (measured 163 u rms at K = 4608,
// u = 2^-24; scripts/split_numerics.py) -- whereas `lo` only ever holds terms of its own size.  `hi` then behaves like an
// fp32 dot product with one rounding per 16 products.
constexpr int PA6[6] = {2, 1, 0, 1, 0, 0}, PB6[6] = {0, 1, 2, 0, 1, 0};
#ifndef RD_NPROD
#define RD_NPROD 6
#endif
constexpr int LO0 = RD_NPROD == 3 ? 3 : 0;      // first kept entry of PA6 / PB6
// hi + lo.  An infinite operand lives in its first term only (split3), so hi = Inf * b1 carries the correct +-Inf (or NaN
// for Inf * 0 / Inf - Inf, as in fp32) while lo may have picked up Inf * 0 = NaN from a ZERO lower term of the other
// operand: an infinite hi therefore wins.  (Only difference to an fp32 product left: Inf * b with 0 < |b| < 2^-133.)
__device__ __forceinline__ float merge_hi_lo(float hi, float lo) { return __builtin_fabsf(hi) == __builtin_inff() ? hi : hi + lo; }

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
#if RD_NPROD == 3
    h = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x) << 16;
    float r = x - __uint_as_float(h);
    if (GUARD) r = (r == r) ? r : 0.f;
    m = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r) << 16;
    l = 0u;
#else
    h = __float_as_uint(x) & 0xffff0000u;
    float r = x - __uint_as_float(h);
    if (GUARD) r = (r == r) ? r : 0.f;
    m = __float_as_uint(r) & 0xffff0000u;
    l = __float_as_uint(r - __uint_as_float(m));
#endif
}
// four values (consecutive k) -> 8-byte groups of 4 bf16, one per term; v_perm_b32 -> {hi16(odd), hi16(even)}.
// kTerm3: the third term exists (six-product build); the three-product build writes / reads two terms only.
constexpr bool kTerm3 = RD_NPROD != 3;
#if RD_NPROD == 3
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two values -> {rn_bf16(x1) : rn_bf16(x0)} and the same of the remainders: v_cvt_pk_bf16_f32, 2 unpack, 2 sub, v_cvt_pk_bf16_f32
template <bool GUARD>
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& ph, unsigned& pm) {
    const f32x2_t v = {x0, x1};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    float r0 = x0 - __uint_as_float(ph << 16), r1 = x1 - __uint_as_float(ph & 0xffff0000u);
    if (GUARD) {
        r0 = (r0 == r0) ? r0 : 0.f;
        r1 = (r1 == r1) ? r1 : 0.f;
    }
    const f32x2_t r = {r0, r1};
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}
#endif
template <bool GUARD = true>
__device__ __forceinline__ void split_pack4v(float x0, float x1, float x2, float x3, uint2& ph, uint2& pm, uint2& pl) {
#if RD_NPROD == 3
    split2_pair<GUARD>(x0, x1, ph.x, pm.x);
    split2_pair<GUARD>(x2, x3, ph.y, pm.y);
    pl = make_uint2(0u, 0u);
#else
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3<GUARD>(x0, h0, m0, l0);
    split3<GUARD>(x1, h1, m1, l1);
    split3<GUARD>(x2, h2, m2, l2);
    split3<GUARD>(x3, h3, m3, l3);
    ph = make_uint2(__builtin_amdgcn_perm(h1, h0, 0x07060302u), __builtin_amdgcn_perm(h3, h2, 0x07060302u));
    pm = make_uint2(__builtin_amdgcn_perm(m1, m0, 0x07060302u), __builtin_amdgcn_perm(m3, m2, 0x07060302u));
    pl = make_uint2(__builtin_amdgcn_perm(l1, l0, 0x07060302u), __builtin_amdgcn_perm(l3, l2, 0x07060302u));
#endif
}
__device__ __forceinline__ void split_pack4(const float4 v, uint2& ph, uint2& pm, uint2& pl) { split_pack4v<true>(v.x, v.y, v.z, v.w, ph, pm, pl); }
__device__ __forceinline__ uint4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
}


}  // namespace rd
