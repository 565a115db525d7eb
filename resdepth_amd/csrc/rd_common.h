// Internal helpers shared by the kernel translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <tuple>
#include <utility>

#include "../../include/resdepth_hip.h"

namespace rd {

void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// ---- profiling (rd_prof_*) -----------------------------------------------------------
struct QuantArgs;
int prof_level();   // 0 off, 1 = MFMA (roofline) kernel classes only, 2 = every kernel class
void prof_begin(hipStream_t s, const char* cls, double flops, double bytes);
void prof_end(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(hipStream_t s_, const char* cls, double flops, double bytes, bool mfma_class = false)
        : s(s_), on(prof_level() >= (mfma_class ? 1 : 2)) {
        if (on) prof_begin(s, cls, flops, bytes);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};

// BN statistics: per-tile fp32 partials [nb][2][C] -> mean / invstd / running statistics in one launch (rd_elementwise.hip)
int bn_reduce_finalize(const float* partial, int nb, int c, double count, float eps, float momentum, float* mean, float* invstd,
                       float* rmean, float* rvar, int64_t* nbt, hipStream_t s);
// second stage of the per-channel reductions (rd_elementwise.hip): sums[c] = sum_b partial[b*qc + c], fixed order
int reduce_partials_f32(const float* partial, double* sums, int nb, int qc, hipStream_t s);

// ---- tuning / diagnosis knobs (registry lives in rd_runtime.hip; set through rd_tune_set() or the RD_TUNE
// environment variable "name=value,name=value" read once at load time).  The kernel translation units only read them.
enum TuneKey {
    TUNE_MFMA_F32 = 0,     // 1: exact-f32 MFMA kernels instead of split-bf16 (also RD_MFMA=f32)
    TUNE_NT_TILE,          // -1 auto | 0: 128x128, 1: 128x64, 2: 64x64 tiles of the NT kernels
    TUNE_NT_HALO,          // -1 auto | 0: never use the halo-reuse conv3x3 kernel
    TUNE_NT_SKEW,          // halo kernel: 1 = skewed halo-row pitch (no LDS bank conflicts) | 0 = plain pitch (r01 layout)
    TUNE_TN_TILE,          // -1 auto | bm*1000 + bn
    TUNE_TN_BLOCKS,        // target block count of the split-K TN kernels
    TUNE_TN_SPLIT,         // -1 auto | 0/1 force the split-bf16 TN kernel off/on
    TUNE_WG_STRIP,         // -1 auto | 0: never use the strip weight-gradient kernel | 1: its register-transpose version (r02) | 4: no image-pair form for 8 x 8 images
    TUNE_WG_MINBLOCKS,     // strip kernel: minimum blocks before image rows are chunked
    TUNE_WG_BLOCKS,        // strip kernel: target block count
    TUNE_WG_OCC,           // strip kernel: 2 (default) = register budget for two waves per SIMD, 1 = one wave (accumulators in AGPRs)
    TUNE_CONVT_PATCH,      // -1 auto | 0: never use the patch transposed-convolution kernels
    TUNE_EDGE_CONV,        // -1 auto | 0: never use the tile kernels of the first / last convolution (rd_edge_conv.hip)
    TUNE_ROWS_BLOCKS,      // first-stage blocks of the per-channel reductions
    TUNE_LAST_BLOCKS,      // first-stage blocks of the last-conv gradient kernels
    TUNE_NT_SPLITK,        // -1 auto | 0: never use the split-K patch kernel for 8 x 8 images (needs rd_set_splitk_workspace)
    TUNE_NT_EPI,           // -1 auto | 0: patch kernels keep the LDS-staged epilogue (r03) instead of the register-direct one
    TUNE_MFMA_PRODUCTS,    // 3: two-term fp16 split, three products, where the operands carry magnitude slots (rd_quant_next; RD_MFMA=split2h) | 6: three-term bf16 split everywhere (RD_MFMA=split3)
    TUNE_D2H_BLOCKS,       // rd_copy_to_host_async: 0 (default) = hipMemcpyAsync (the runtime's blit kernel) | n > 0: own copy kernel with n workgroups (r05 experiment: slower)
    TUNE_COUNT
};
int tune(int key);
// split MFMA arithmetic enabled (default) or exact f32 (TUNE_MFMA_F32)
inline int mfma_split() { return !tune(TUNE_MFMA_F32); }
// products per fp32 multiply of the split kernels: 3 (operands with magnitude slots) or 6
inline int mfma_products() { return tune(TUNE_MFMA_PRODUCTS) == 3 ? 3 : 6; }
// rd_quant_next(): the magnitude slots the NEXT entry point of this host thread takes (cleared by the take)
struct QuantArgs {
    const unsigned* a;     // operand A (activations / gradients); a slot = RD_AMAX_SLOT_BYTES
    const unsigned* b;     // operand B (packed weights; the second activation operand of a weight gradient)
    unsigned* out;         // receives max |output|
    unsigned* out2;        // second output of the call (pooled tensor), or the weight slot a pack call fills
    int img_stride;        // 0: one slot per tensor | > 0 (rd_quant_next_img): a, out, out2 are ARRAYS of slots, one per image of the
                           // batch, this many words apart (inference: a tile's scale then depends on that tile alone); b stays per tensor
};
// the slots of the next call.  quant_take(): entry points that know nothing of per-image slots -- they get NO slots when the
// caller set per-image ones (six-product body, nothing committed: an untouched slot reads as "magnitude unknown", quant_select).
// quant_take_img(): the entry points of the inference path that index the arrays by image.
QuantArgs quant_take();
QuantArgs quant_take_img();
// conv3x3 weight-gradient strip kernel (rd_wgrad_strip.hip): number of split-K slabs it will write for this shape
// (0 = shape not handled, use the TN kernel), and its launcher (*splits_out = 0 when it did not run; *swapped_out = 1
// when the slab is the mirrored transpose [Cin][(8 - tap) * Cout + co], see plan_strip)
int wgrad_strip_splits(int n, int h, int w, int cin, int cout);
int wgrad_strip_launch(const float* x, const float* dz, float* slab, int n, int h, int w, int cin, int cout, hipStream_t s,
                       int* splits_out, int* swapped_out, const unsigned* x_amax = nullptr, const unsigned* dz_amax = nullptr);

// transposed-convolution patch kernels (rd_convt.hip); *launched = 0 when the shape is left to the generic NT kernel
int convt_fwd_launch(const float* x, const void* wsplit, size_t wsplit_bytes, const float* bias, const float* skip,
                     const float* sk_mean, const float* sk_invstd, const float* sk_gamma, const float* sk_beta, float sk_slope,
                     const float* sk_slope_dev, float* out, int n, int h, int w, int cin, int cout, hipStream_t s, int* launched,
                     const QuantArgs& qa);

// transposed-convolution weight gradient (rd_convt.hip): workspace of its split-K slabs (0: shape left to the generic TN kernel)
size_t convt_wgrad_ws_bytes(int n, int h, int w, int cin, int cout);
int convt_wgrad_launch(const float* x, const float* dout, float* slab, int n, int h, int w, int cin, int cout, hipStream_t s,
                       int* splits_out, const unsigned* x_amax = nullptr, const unsigned* dout_amax = nullptr);

// last convolution, tile kernels (rd_edge_conv.hip); *launched = 0 / blocks = 0 when the shape stays on the generic kernels
int conv_last_fwd_launch(const float* s_in, const float* wt, const float* bias, const float* x_nchw, int xc, float* out, int n,
                         int h, int w, int c, hipStream_t s, int* launched);
int conv_last_dgrad_launch(const float* dout, const float* wt, float* ds, int n, int h, int w, int c, hipStream_t s, int* launched);
int conv_last_dgrad_bn_launch(const float* dout, const float* wt, float* ds, int n, int h, int w, int c, const float* bn_z,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                              const float* slope_dev, float* part, hipStream_t s, int* rows);
int conv_last_wgrad_blocks(int n, int h, int w, int c);
// tail of the network (last up-convolution composed with the last convolution, rd_edge_conv.hip)
// level 0's activation act(BN(z)) as an operand that is recomputed from z where it is used (the descriptor the transposed
// convolution's lazy skip takes); `t16` / `b9`: the up-convolution's contribution through the composed stencil (see TailSkip use)
struct TailSkip {
    const float* z;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* slope_dev;
    float slope;
};
int tail_compose_launch(const float* wt, const float* bt, const float* wl, float* M, float* V, float* VT, float* B9, int cin, int c0,
                        hipStream_t s);
int conv_last_fwd_tail_launch(const TailSkip& sk, const float* t16, const float* b9, const float* wt, const float* bias,
                              const float* x_nchw, int xc, float* out, int n, int h, int w, int c, hipStream_t s);
int conv_last_wgrad_tail_launch(const TailSkip& sk, const float* dout, double* partial, int n, int h, int w, int c, hipStream_t s);
int tail_t16_launch(const TailSkip& sk, const float* V, float* t16, long pixels, int cin, hipStream_t s);
int convt_last_wgrad_launch(const float* x, const TailSkip& sk, const float* dout, const float* wl, float* dwt, double* partial,
                            double* c16, int n, int hc, int wc, int cin, int c0, hipStream_t s);
int conv_last_tail_blocks(int n, int h, int w);
int conv_last_bwd_tail_fused_launch(const TailSkip& sk, const float* dout, const float* wl, double* wpartial, float* bn_part, int n,
                                    int h, int w, int c, hipStream_t s);
int tail_wl_finish_launch(const double* partial, int nb, const double* c16, const float* wt, const float* bt, float* dw, float* dbias,
                          int cin, int c0, hipStream_t s);
bool tail_shape_ok(int cin);
int tail_corr_blocks(int n, int hc, int wc);

int convt_last_dgrad_launch(const float* dout, const float* V, float* dprev, int n, int hc, int wc, int cin, const float* bn_z,
                            const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                            const float* slope_dev, float* part, hipStream_t s, int* rows);
// first convolution, segment kernels: tiles / blocks = 0 when the shape stays on the generic kernel
int conv_first_seg_tiles(int n, int h, int w, int cin, int cout);
int conv_first_wgrad_seg_blocks(int n, int h, int w, int cin, int cout);
// weight gradient with dz computed on the fly: the arguments of rd_bn_act_bwd_apply for the first block (pooled form)
struct FirstBnBwd {
    const float* z;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* slope_dev;
    float slope;
    const float* g_full;
    const float* g_pool;
    const unsigned char* idx;
    const double* sums;
    double count;
    int training;
    // g_full == NULL and these set: the full-resolution gradient operand is the last convolution's data gradient of the
    // network's output gradient (lib/UNet.py:227: skip ADD + conv C0 -> 1), evaluated from dout [N][H][W] and w_last [C0][9]
    // per element -- g[q][c] = sum_tap dout[q - off(tap)] w_last[c][tap] -- instead of being read
    const float* dout;
    const float* w_last;
};
int conv_first_fwd_act_launch(const float* x, const float* wt, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float slope, const float* slope_dev, float* a, float* pooled, int n, int h, int w,
                              int cin, int cout, hipStream_t s, unsigned* p_amax = nullptr, int amax_img_stride = 0);
int conv_first_seg_launch(bool wgrad, const float* x, const float* wt, float* z, const float* dz, float* partial, int n, int h,
                          int w, int cin, int cout, hipStream_t s, const FirstBnBwd* bn = nullptr);
int conv_last_wgrad_launch(const float* s_in, const float* dout, double* partial, int n, int h, int w, int c, hipStream_t s);

// split-K scratch registered for this stream (rd_set_splitk_workspace): tile tickets (all zero between launches) + slab area
bool splitk_workspace(hipStream_t s, unsigned** tickets, int* n_tickets, float** slab, size_t* slab_bytes);

inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Division of 0 <= n < 2^31 by a launch-time constant d (image width, pixels per image) as one v_mul_hi + shift:
// q = (n * mul) >> (32 + sh) with mul = ceil(2^(31+s) / d), s = ceil(log2 d), sh = s - 1  (error term n * e < 2^(31+s)
// because e < d <= 2^s and n < 2^31).  Powers of two give mul = 2^31, i.e. a plain shift.  d = 1: sh = -1 (identity).
struct FastDiv {
    unsigned mul;
    int sh;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f = {0u, -1};
    if (d <= 1) return f;
    int s = 0;
    while ((1u << s) < d) ++s;
    f.mul = (unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d);
    f.sh = s - 1;
    return f;
}
// pixel index m of an [img][H][W] grid -> (img, y, x); hw = H * W
struct PixDiv {
    FastDiv w, hw;
    int W, HW;
};
inline PixDiv make_pixdiv(int h, int w) {
    PixDiv d;
    d.w = make_fastdiv((unsigned)w);
    d.hw = make_fastdiv((unsigned)h * (unsigned)w);
    d.W = w;
    d.HW = h * w;
    return d;
}

// ---- launch plans (include/resdepth_hip.h: rd_plan_*) -------------------------------------------------------------------
// Every kernel of the library is launched through rd::launch (RD_LAUNCH): while a plan is being recorded, a launch on one of
// the plan's two streams is appended to it -- kernel function, grid, block, dynamic LDS and a COPY of the argument values --
// and a recorded plan is replayed by rd_plan_replay with one hipLaunchKernel per entry, no Python, no argument marshalling.
bool plan_recording();
void plan_note_launch(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, void** args, const size_t* sizes,
                      const size_t* aligns, int nargs);
// a stream operation of the library that a plan cannot hold (memcpy / memset nodes): the recording is marked unusable
void plan_poison(hipStream_t s, const char* why);

template <typename... KArgs, size_t... I>
inline void launch_impl(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, std::tuple<KArgs...>& vals,
                        std::index_sequence<I...>) {
    void* ptrs[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {(void*)&std::get<I>(vals)...};
    if (plan_recording()) {
        const size_t sizes[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {sizeof(KArgs)...};
        const size_t aligns[sizeof...(KArgs) > 0 ? sizeof...(KArgs) : 1] = {alignof(KArgs)...};
        plan_note_launch((const void*)kern, grid, block, shmem, s, ptrs, sizes, aligns, (int)sizeof...(KArgs));
    }
    (void)hipLaunchKernel((const void*)kern, grid, block, ptrs, shmem, s);
}
template <typename... KArgs, typename... Args>
inline void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    std::tuple<KArgs...> vals(static_cast<KArgs>(std::forward<Args>(args))...);      // the kernel's own parameter types
    launch_impl(kern, grid, block, shmem, s, vals, std::index_sequence_for<KArgs...>());
}
#define RD_LAUNCH(kernel, grid, block, shmem, stream, ...) rd::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)

#define RD_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            rd::set_error(__VA_ARGS__);    \
            return RD_ERR_ARG;             \
        }                                  \
    } while (0)

#define RD_LAUNCH_CHECK(what)                                   \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return rd::check_hip(e__, what); \
    } while (0)

}  // namespace rd

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Non-temporal (streaming) 16-byte accesses for tensors a kernel touches exactly once and nobody reads again soon: the loads /
// stores carry the `nt` cache policy, so the lines do not displace what the NEXT kernel needs from the 32 MB of L2.  Measured r05
// (profiles/r05_notes.md section 14): bn_act_pool_fwd 0.322 -> 0.270 ms per step (4.9 -> 5.9 TB/s of algorithmic bytes) with z read
// this way.  -DRD_NO_NT builds the plain accesses for A/B runs.
// buffer intrinsics: auxiliary cache-policy operand; bit 1 = nt on gfx940+
#ifdef RD_NO_NT
#define RD_AUX_NT 0
#else
#define RD_AUX_NT 2
#endif
typedef float rd_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned char rd_u8x4 __attribute__((ext_vector_type(4)));
#if defined(__HIPCC__)
__device__ __forceinline__ float4 ld_nt4(const float* p) {
#ifdef RD_NO_NT
    return *reinterpret_cast<const float4*>(p);
#else
    const rd_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rd_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st_nt4(float* p, float4 v) {
#ifdef RD_NO_NT
    *reinterpret_cast<float4*>(p) = v;
#else
    __builtin_nontemporal_store(rd_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<rd_f32x4*>(p));
#endif
}
__device__ __forceinline__ float ld_nt1(const float* p) {
#ifdef RD_NO_NT
    return *p;
#else
    return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ void st_nt1(float* p, float v) {
#ifdef RD_NO_NT
    *p = v;
#else
    __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ uchar4 ld_nt_u8x4(const unsigned char* p) {
#ifdef RD_NO_NT
    return *reinterpret_cast<const uchar4*>(p);
#else
    const rd_u8x4 v = __builtin_nontemporal_load(reinterpret_cast<const rd_u8x4*>(p));
    return make_uchar4(v.x, v.y, v.z, v.w);
#endif
}
__device__ __forceinline__ void st_nt_u8x4(unsigned char* p, uchar4 v) {
#ifdef RD_NO_NT
    *reinterpret_cast<uchar4*>(p) = v;
#else
    __builtin_nontemporal_store(rd_u8x4{v.x, v.y, v.z, v.w}, reinterpret_cast<rd_u8x4*>(p));
#endif
}
#endif
