// Device-side helpers shared by the MFMA kernel translation units (rd_igemm.hip, rd_wgrad_strip.hip).
#pragma once
#include "rd_common.h"

namespace rd {

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

// x = h + m + l exactly; h, m, l have <= 8 significant bits (bf16-representable), returned as fp32 bit patterns
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xffff0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}
// float4 (4 consecutive k) -> three 8-byte groups of 4 bf16 (one per term); v_perm_b32 -> {hi16(odd), hi16(even)}
__device__ __forceinline__ void split_pack4(const float4 v, uint2& ph, uint2& pm, uint2& pl) {
    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
    split3(v.x, h0, m0, l0);
    split3(v.y, h1, m1, l1);
    split3(v.z, h2, m2, l2);
    split3(v.w, h3, m3, l3);
    ph = make_uint2(__builtin_amdgcn_perm(h1, h0, 0x07060302u), __builtin_amdgcn_perm(h3, h2, 0x07060302u));
    pm = make_uint2(__builtin_amdgcn_perm(m1, m0, 0x07060302u), __builtin_amdgcn_perm(m3, m2, 0x07060302u));
    pl = make_uint2(__builtin_amdgcn_perm(l1, l0, 0x07060302u), __builtin_amdgcn_perm(l3, l2, 0x07060302u));
}
__device__ __forceinline__ uint4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4((unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w);
}


}  // namespace rd
