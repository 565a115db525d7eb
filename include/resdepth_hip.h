/*
 * resdepth_hip.h -- C ABI of libresdepth_hip.so (gfx950 / MI355X).
 *
 * The reference (prs-eth/ResDepth) has no FFI / plugin interface: its hot path is a
 * chain of torch.nn modules.  This header is the boundary UNDER the reference's Python
 * surface (UNet / Trainer): every entry point replaces one implicit
 * ATen/cuDNN kernel family the reference reaches through torch.nn, cited per function
 * as <reference file>:<line>.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; the library never allocates,
 *    frees or keeps device memory; scratch comes in through (ws, ws_bytes) and the
 *    matching rd_*_ws_bytes() query;
 *  - activations are NHWC fp32 ("pixel-major, channel-contiguous"): index
 *    ((n*H + y)*W + x)*C + c.  Only the network input (NCHW, few channels) and the
 *    1-channel output / target / mask keep the reference's NCHW layout;
 *  - H and W are arbitrary positive sizes at the op level (the patch / strip kernels take W % 16 == 0 and H % 8 == 0,
 *    every other shape runs the generic kernels); a depth-d network needs H and W to be multiples of 2^d.  The reference
 *    itself only ever feeds square tiles of 2^k >= 2^(depth+2) pixels (lib/validate_arguments.py:143-171), which is the
 *    rule resdepth_amd.validate_tile_size restates for callers;
 *  - all work is enqueued on `stream` (a hipStream_t); nothing synchronises;
 *  - return 0 on success, non-zero on error; rd_last_error_string() (thread-local)
 *    describes the last failure of the calling thread;
 *  - re-entrant from several host threads (forward thread + autograd thread).
 */
#ifndef RESDEPTH_HIP_H
#define RESDEPTH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rd_stream_t; /* hipStream_t */

#define RD_OK 0
#define RD_ERR_ARG 1
#define RD_ERR_WS 2
#define RD_ERR_HIP 3

int rd_version(void); /* 100: r01-r03; 101: rd_set_splitk_workspace registrations belong to (current device, stream); 102: rd_host_register & co; 103: rd_mfma_products; 104: rd_adam_step_dev; 105 (r06): rd_quant_next / rd_amax, packed operands hold both split forms; 106: rd_plan_*; 107: rd_quant_next_img */
/* Arithmetic of the split MFMA kernels -- ONE library, chosen per launch (csrc/rd_mfma_dev.h):
 *   6  "split3"   x = x1 + x2 + x3 (three bf16 terms, exact), six products per multiply on v_mfma_f32_32x32x16_bf16.  No
 *                 assumption about the operands; what every launch falls back to.
 *   3  "split2h"  x ~ (x1 + x2) / s: two FP16 terms of s x (11 bits each, |error| <= 2^-22 |s x|), s = a power of two per
 *                 operand tensor that maps the tensor's largest magnitude into [2^14, 2^15); three products (a1 b1, a1 b2,
 *                 a2 b1) on v_mfma_f32_32x32x16_f16, scaled back exactly: <= 3 * 2^-22 relative error per product, half the
 *                 matrix instructions.  Needs the operands' magnitudes: see rd_quant_next.
 * rd_mfma_products() returns the mode of the process: the knob `mfma_products` (rd_tune_set, RD_TUNE, RD_MFMA=split2h | split3).
 * In mode 3 a launch still runs the six-product body when an operand comes without a magnitude slot or its maximum is +-Inf
 * (decided on the device at kernel start, no host round trip) -- fp32's non-finite semantics are those of mode 6. */
int rd_mfma_products(void);
/* Magnitude slots.  A slot is RD_AMAX_SLOT_BYTES (2 KB) of device memory, 128-byte aligned, zeroed by the caller before its
 * producer runs: sixteen 32-bit words, one at the start of each 128-byte line (same-line device atomics serialise; sixteen
 * lines take them in parallel).  The producer max-accumulates the IEEE bit patterns of |x| into them (atomic integer max:
 * order-independent, so run-to-run identical) and a consumer takes the largest of the 16 words as the tensor's maximum.
 *   rd_quant_next(a, b, out, out2): the slots the NEXT rd_* call of this host thread takes (any may be NULL); that call clears
 *     them again, whether it uses them or not.  Consumers (rd_conv3x3_fwd*, rd_conv3x3_bwd_data*, rd_convt2x2_fwd*,
 *     rd_convt2x2_bwd_data*, rd_conv1x1_*): a = slot of the activation / gradient operand, b = slot of the packed weight;
 *     weight gradients (rd_*_bwd_weight): a = slot of the gradient operand (dz / dout / dy), b = slot of x.  Producers: out =
 *     slot that receives max |primary output| (z is not an operand and gets none: rd_conv3x3_fwd_act -> a, rd_conv3x3_bwd_data*
 *     -> dx, rd_convt2x2_fwd* -> out, rd_bn_act_pool_fwd -> a (un-pooled form), rd_bn_act_bwd_apply -> dz), out2 = slot of
 *     the pooled output (rd_conv3x3_fwd_act, rd_bn_act_pool_fwd, rd_conv3x3_first_fwd_act); rd_pack_*: out2 = the weight's
 *     slot, which the call fills itself before it writes the three-product form of the operand.
 *   rd_amax(x, n, slot): max |x[0..n)| into a (zeroed) slot: operands that no producer of this library wrote. */
int rd_quant_next(const unsigned* a_amax, const unsigned* b_amax, unsigned* out_amax, unsigned* out2_amax);
/* Per-IMAGE slots (inference, r06): a, out and out2 are arrays of slots, one per image of the batch, `img_stride_words` 32-bit
 * words apart (a multiple of 32, >= RD_AMAX_SLOT_BYTES / 4); b (the packed weight) stays one slot.  A block of a consumer scales
 * its activation operand by ITS image's maximum, a block of a producer commits to its image's slot: a tile's result then depends
 * on that tile alone -- not on the tiles that share its batch, as with one scale per tensor -- which the tiled sweep of
 * lib/evaluation.py:460-567 promises (same raster however the tiles are batched or sharded).  Taken by rd_conv3x3_fwd_act
 * (its patch kernels; the 8 x 8 level, whose patches hold two images, and any launch that falls to a generic row-tile kernel
 * run the six-product body and commit nothing), rd_convt2x2_fwd / rd_convt2x2_fwd_bnskip (tiles inside one image, else the
 * same) and rd_conv3x3_first_fwd_act (out2); every other entry point ignores slots set this way.  A slot nobody wrote reads as
 * zero = "magnitude unknown": the consumer runs the six-product body (so does an all-zero operand, in either slot form). */
int rd_quant_next_img(const unsigned* a_amax, const unsigned* b_amax, unsigned* out_amax, unsigned* out2_amax, int img_stride_words);
int rd_amax(const float* x, long long n, unsigned* slot, rd_stream_t s);
/* zero `bytes` (a multiple of 16, 16-byte aligned) with a kernel launch -- which a launch plan records, unlike a memset node */
int rd_zero(void* p, size_t bytes, rd_stream_t s);
/* n <= 8 device-to-device copies (16-byte aligned ranges of whole 16-byte units) in ONE launch: a batch's tensors into the static
 * input buffers of a captured / planned iteration (lib/Trainer.py:212-215 hands the step a new batch every iteration) */
int rd_copy_segments(void* const* dst, const void* const* src, const size_t* bytes, int n, rd_stream_t s);
#define RD_AMAX_SLOT_BYTES 2048
const char* rd_last_error_string(void);

/* ---- launch plans: the iteration's launch list recorded once, replayed from C (r06) ------------------------------------------
 * A training iteration (lib/Trainer.py:212-222) is ~110 kernel launches on two HIP streams, enqueued through Python + ctypes at
 * ~30-50 us each: 3-6 ms of host time per step against 7 ms of GPU time.  A plan holds that list -- for every launch of THIS
 * library on one of two streams: the kernel, its grid / block / dynamic LDS and a copy of its argument values; the event
 * records / waits that order the two streams; and segment ends, where the host does something between replays (a collective of
 * torch.distributed, which stays in Python).  rd_plan_replay enqueues one segment with one hipLaunchKernel per entry (~4 us).
 * The pointers inside the arguments are the ones of the recorded iteration: the caller records while the iteration's memory
 * comes from a pool that stays reserved (torch: under a CUDA-graph capture, whose private pool outlives it) and feeds the batch
 * through fixed input tensors.  What a plan cannot hold (a hipMemcpyAsync / hipMemsetAsync of the library on a plan stream)
 * marks the recording unusable: rd_plan_end then returns NULL with the reason in rd_last_error_string().
 *   rd_plan_begin(main, side)        start recording launches on these two streams (one recording per process at a time; launches
 *                                    on other streams are ignored)
 *   rd_plan_event_record(stream)     -> event index >= 0 (-1: not recording / not a plan stream)
 *   rd_plan_event_wait(stream, ev)   the stream waits for that event (ev < 0 on a plan stream marks the recording unusable)
 *   rd_plan_segment()                ends the current segment -> its index
 *   rd_plan_poison(why)              mark the recording unusable (host logic that knows the iteration cannot be replayed)
 *   rd_plan_end(&launches, &segs)    -> plan handle | NULL
 *   rd_plan_replay(plan, seg, main, side)   enqueue segment `seg` on the given streams (they need not be the recorded ones)
 *   rd_plan_free(plan) */
int rd_plan_begin(rd_stream_t main_stream, rd_stream_t side_stream);
int rd_plan_event_record(rd_stream_t stream);
int rd_plan_event_wait(rd_stream_t stream, int ev);
int rd_plan_segment(void);
int rd_plan_poison(const char* why);
void* rd_plan_end(int* n_launches, int* n_segments);
int rd_plan_replay(void* plan, int segment, rd_stream_t main_stream, rd_stream_t side_stream);
int rd_plan_free(void* plan);
int rd_plan_dump(void* plan);   /* diagnosis: the operations in enqueue order, one line each, to stderr */

/* ---- weight (re)packing: torch layouts -> GEMM operand layouts ------------------- */
/* Every packed operand B[rows][K = taps*Cin] is ONE opaque caller-owned buffer of
 * rd_packed_weight_bytes(rows, taps, Cin) bytes: the fp32 GEMM layout (rows*K floats, described
 * below) followed by the same matrix pre-split in the fragment order of the split MFMA kernels, in BOTH
 * forms: three bf16 terms per element (96 bytes per row and 16-k step), then two fp16 terms of the
 * scaled element (64 bytes; written only when the pack call was given the weight's magnitude slot:
 * rd_quant_next(..., out2) in mode 3).  The conv entry points take the buffer's base pointer.  (rows, taps, Cin) per operand:
 *   conv3x3  wf: (Cout, 9, Cin)   wd: (Cin, 9, Cout)
 *   convT2x2 wtf: (4*Cout, 1, Cin)  wtd: (Cin, 4, Cout)
 *   conv1x1  wf: (Cout, 1, Cin)   wt: (Cin, 1, Cout) */
size_t rd_packed_weight_bytes(int rows, int taps, int cin);
/* Optional scratch for the 3x3 convolutions on 8 x 8 images (the bottleneck of cfg-S, lib/UNet.py:210: few output tiles,
 * K = 9 * 512).  That kernel accumulates K in ranges of 128 channels; with `ws` registered (256-byte aligned, >= 1.1 MB;
 * 64 MB covers cfg-S / cfg-M) every rd_conv3x3_fwd* / rd_conv3x3_bwd_data* launch on `stream` with a small grid runs one
 * block per range -- partial tiles + one ticket per output tile live in `ws` -- and the last block adds the ranges in the
 * same order.  Results are the same bits with or without a registration, at any batch size; only the block count differs.
 * A registration belongs to (the device that is current at the call, stream) -- a stream handle alone does not name a device
 * (the null stream is handle 0 on all of them) -- and `ws` must be memory of that device (RD_ERR_ARG otherwise); launches look
 * the scratch up under (their current device, their stream).  32 MB + 64 KB is the most any launch uses.
 * The caller keeps `ws` alive and must not use it for anything else; ws = NULL un-registers (current device, stream). */
int rd_set_splitk_workspace(void* ws, size_t bytes, rd_stream_t stream);

/* nn.Conv2d weight [Cout][Cin][3][3] (lib/UNet.py:4-5) ->
 *   wf[co][tap][ci]              B operand of the forward implicit GEMM
 *   wd[ci][tap][co] = w[co][ci][8-tap]   B operand of the data-gradient GEMM (nullable) */
int rd_pack_conv3x3_weight(const float* w_oihw, float* wf, float* wd, int cout, int cin, rd_stream_t s);
/* All packed operands of a network in ONE or TWO launches (split-bf16 mode).  items_dev: device array of n_items records of
 * 10 int64:
 *   { w (device pointer, torch layout), forward-operand buffer, data-gradient-operand buffer, kind (0 = conv3x3 [Cout][Cin][3][3],
 *     1 = ConvTranspose2d [Cin][Cout][2][2]), Cout, Cin, first piece index, fp32 forward-operand pointer (convT only: the buffer
 *     base again when the fp32 layout wtf is wanted, else 0), first tile index, magnitude slot of w (device pointer to a zeroed
 *     RD_AMAX_SLOT_BYTES block; 0 = six-product form only) }
 * A layer whose channel counts are both multiples of 32 is packed by the TILE kernel (rd_pack_item_tiles() > 0 tiles, coalesced
 * loads through LDS; it then owns no pieces: its `first piece index` is the running piece count), every other layer piece by
 * piece (rd_pack_item_pieces() pieces, gathered loads; it owns no tiles).  The buffers are the opaque packed buffers of
 * rd_packed_weight_bytes().  Same results as rd_pack_conv3x3_weight / rd_pack_convt2x2_weight layer by layer. */
long long rd_pack_item_pieces(int kind, int cout, int cin, int with_f32);
long long rd_pack_item_tiles(int kind, int cout, int cin);
int rd_pack_weights_fused(const void* items_dev, int n_items, long long total_pieces, long long total_tiles, rd_stream_t s);
/* Inference: eval-mode BatchNorm folded into the forward operand -- wf[co][tap][ci] = w[co][ci][tap] * row_scale[co] with
 * row_scale = gamma / sqrt(running_var + eps) (lib/UNet.py:45 in eval mode, reached from lib/evaluation.py:497). */
int rd_pack_conv3x3_weight_folded(const float* w_oihw, const float* row_scale, float* wf, int cout, int cin, rd_stream_t s);
/* nn.ConvTranspose2d weight [Cin][Cout][2][2] (lib/UNet.py:21) ->
 *   wtf[(a*2+b)*Cout + co][ci]   forward B operand
 *   wtd[ci][(a*2+b)*Cout + co]   data-gradient B operand (nullable) */
int rd_pack_convt2x2_weight(const float* w_iohw, float* wtf, float* wtd, int cin, int cout, rd_stream_t s);

/* ---- 3x3 / stride 1 / pad 1 convolution, Cin % 4 == 0 (lib/UNet.py:4-5,44,65,85) - */
/* z[N,H,W,Cout] = conv(x[N,H,W,Cin], w)          (replaces nn.Conv2d.forward, no bias) */
int rd_conv3x3_fwd(const float* x, const float* wf, float* z, int n, int h, int w, int cin, int cout, rd_stream_t s);
/* same, and the BatchNorm batch statistics of z come out of the GEMM epilogue: sums[2*Cout] doubles
 * (sum, sum of squares per channel) -- the input of rd_bn_stats_finalize -- without re-reading z */
size_t rd_conv3x3_fwd_stats_ws_bytes(int n, int h, int w, int cin, int cout);
int rd_conv3x3_fwd_stats(const float* x, const float* wf, float* z, double* sums, int n, int h, int w, int cin, int cout,
                         void* ws, size_t ws_bytes, rd_stream_t s);
/* Convolution + the complete training-mode BatchNorm statistics step (lib/UNet.py:44-45) in two launches: the per-tile
 * partials of the epilogue are reduced and turned into mean / invstd, the running statistics are updated
 * (momentum, unbiased variance) and num_batches_tracked incremented (all three nullable).  count = N*H*W. */
int rd_conv3x3_fwd_bn(const float* x, const float* wf, float* z, double count, float eps, float momentum, float* mean,
                      float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked, int n, int h,
                      int w, int cin, int cout, void* ws, size_t ws_bytes, rd_stream_t s);
/* Inference (eval-mode BN folded, rd_pack_conv3x3_weight_folded): a = act(conv(x) + shift[co]), shift = beta -
 * running_mean * row_scale; `pooled` (nullable, needs W >= 16, H >= 8) also receives MaxPool2d(2,2)(a) from the same epilogue:
 * conv -> BN -> act -> pool of an encoder level (lib/UNet.py:201-207) in ONE kernel, no pre-BN tensor is ever written. */
int rd_conv3x3_fwd_act(const float* x, const float* wf_folded, const float* shift, float slope, float* a, float* pooled, int n,
                       int h, int w, int cin, int cout, rd_stream_t s);
/* dx[N,H,W,Cin] = conv^T(dz)                     (autograd data gradient of the above) */
int rd_conv3x3_bwd_data(const float* dz, const float* wd, float* dx, int n, int h, int w, int cin, int cout,
                        rd_stream_t s);
/* dw[Cout][Cin][3][3] (torch layout) = sum_p dz[p] (x) x[p+tap]   (autograd weight gradient) */
size_t rd_conv3x3_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout);
int rd_conv3x3_bwd_weight(const float* x, const float* dz, float* dw_oihw, int n, int h, int w, int cin, int cout,
                          void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- first encoder conv: NCHW input with 1..6 channels -> NHWC (lib/UNet.py:159) -- */
int rd_conv3x3_first_fwd(const float* x_nchw, const float* w_oihw, float* z, int n, int h, int w, int cin, int cout,
                         rd_stream_t s);
/* Inference: the same convolution with the block's eval-mode BatchNorm (mean / invstd from rd_bn_eval_stats), activation and
 * MaxPool2d(2,2) in its epilogue (lib/UNet.py:44-47,161 in eval mode, reached from lib/evaluation.py:497):
 * a[N,H,W,Cout] = act(gamma * (conv(x) - mean) * invstd + beta) and, if `pooled` is non-NULL, pooled[N,H/2,W/2,Cout] = max over
 * 2x2 windows of a -- the pre-BN tensor is never written.  Same arithmetic, in the same order, as rd_conv3x3_first_fwd followed
 * by rd_bn_act_pool_fwd (bit-identical a / pooled).  rd_conv3x3_first_fwd_act_available: 1 for the shapes it covers. */
int rd_conv3x3_first_fwd_act_available(int n, int h, int w, int cin, int cout);
int rd_conv3x3_first_fwd_act(const float* x_nchw, const float* w_oihw, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float slope, const float* slope_dev, float* a, float* pooled, int n, int h, int w,
                             int cin, int cout, rd_stream_t s);
size_t rd_conv3x3_first_fwd_stats_ws_bytes(int n, int h, int w, int cin, int cout);
int rd_conv3x3_first_fwd_stats(const float* x_nchw, const float* w_oihw, float* z, double* sums, int n, int h, int w,
                               int cin, int cout, void* ws, size_t ws_bytes, rd_stream_t s);
int rd_conv3x3_first_fwd_bn(const float* x_nchw, const float* w_oihw, float* z, double count, float eps, float momentum,
                            float* mean, float* invstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, int n, int h, int w, int cin, int cout, void* ws, size_t ws_bytes,
                            rd_stream_t s);
size_t rd_conv3x3_first_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout);
int rd_conv3x3_first_bwd_weight(const float* x_nchw, const float* dz, float* dw_oihw, int n, int h, int w, int cin,
                                int cout, void* ws, size_t ws_bytes, rd_stream_t s);
/* The same weight gradient with dz = rd_bn_act_bwd_apply(z, ..., g_full, g_pool, idx, sums, count, training) of the first
 * block (lib/UNet.py:44-47,159-161 differentiated) evaluated on the fly: this kernel is dz's only reader, so the largest
 * gradient tensor of the network (64 channels at full resolution) is never written.  Arguments as rd_bn_act_bwd_apply (pooled
 * form) + rd_conv3x3_first_bwd_weight; same workspace size.  dout / w_last (with g_full = NULL): the full-resolution gradient
 * operand is rd_conv3x3_last_bwd_data(dout, w_last) -- the skip gradient of level 0, lib/UNet.py:218-227 -- evaluated per
 * element from the 1-channel dout instead of being read.  Cin <= 3, Cout in {32, 64, 128}, H and W even
 * (rd_conv3x3_first_bwd_weight_bn_available). */
int rd_conv3x3_first_bwd_weight_bn_available(int n, int h, int w, int cin, int cout);
int rd_conv3x3_first_bwd_weight_bn(const float* x_nchw, const float* z, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float slope, const float* slope_dev, const float* g_full,
                                   const float* g_pool, const uint8_t* idx, const double* sums, double count, int training,
                                   const float* dout, const float* w_last, float* dw_oihw, int n, int h, int w, int cin, int cout,
                                   void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- last conv C -> 1 (+bias) with the outer residual add fused (lib/UNet.py:184,227-244)
 * out[N,1,H,W] = conv(s[N,H,W,C], w[1][C][3][3]) + bias[0] + x0, x0 = x_nchw[:,0] (both nullable) */
int rd_conv3x3_last_fwd(const float* s_in, const float* w_oihw, const float* bias, const float* x_nchw, int x_channels,
                        float* out, int n, int h, int w, int c, rd_stream_t s);
int rd_conv3x3_last_bwd_data(const float* dout, const float* w_oihw, float* ds, int n, int h, int w, int c,
                             rd_stream_t s);
/* ---- the tail: last up-convolution composed with the last convolution (lib/UNet.py:21,218-227) ----
 * The last decoder level is ConvTranspose2d(Cin -> C0, k2 s2) + skip add, followed directly by the 3x3 convolution C0 -> 1.
 * Both are linear, so the backward of the up-convolution does not need the C0-channel gradient g = conv_last^T(dout) at full
 * resolution (the largest gradient tensor of the network) as an operand:
 *   M[ci][ab][tap] = sum_co Wt[ci][co][a][b] wl[co][tap]                    (Cin x 4 x 9)
 *   V[ci][d]       = sum of M over the (ab, tap) with (a, b) - off(tap) = d   (Cin x 16, d in [-1, 2]^2, off = (tap/3-1, tap%3-1))
 *   d loss / d input[p][ci] = sum_d dout[2p + d] V[ci][d]                     (rd_convt_last_bwd_data)
 * rd_tail_compose writes M and V (tiny; once per step), and on request VT [16][Cin] = V transposed (the weight of a 1x1
 * convolution Cin -> 16) and B9[tap] = sum_co w_last[co][tap] bias_t[co] (bias_t nullable) for the forward below.  rd_convt_last_bwd_data(hc, wc = the COARSE grid) == rd_convt2x2_bwd_data
 * applied to rd_conv3x3_last_bwd_data(dout) up to fp32 rounding; bn_z != NULL: the BN-backward statistics hook of
 * rd_convt2x2_bwd_data_bnstats (mode 1), part / rows_out as there.  Cin in {16, 32, 64, 128, 256} (rd_tail_available). */
int rd_tail_available(int cin, int c0);
int rd_tail_compose(const float* wt_iohw, const float* bias_t, const float* w_last, float* M, float* V, float* VT, float* B9, int cin,
                    int c0, rd_stream_t s);
int rd_convt_last_bwd_data(const float* dout, const float* V, float* dprev, int n, int hc, int wc, int cin, const float* bn_z,
                           const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                           const float* slope_dev, float* part, size_t part_floats, int* rows_out, rd_stream_t s);
/* Weight gradient of the last up-convolution from dout: C16[ci][d] = sum_p x[p][ci] dout[2p + d] (16 correlations per input
 * channel, one pass over the up-convolution's INPUT x [N, hc, wc, Cin]), then dWt[ci][co][a][b] = sum_tap w_last[co][tap]
 * C16[ci][(a,b) - off(tap)]  == rd_convt2x2_bwd_weight(x, rd_conv3x3_last_bwd_data(dout)) up to fp32 rounding. */
size_t rd_convt_last_bwd_weight_ws_bytes(int n, int hc, int wc, int cin);
int rd_convt_last_bwd_weight(const float* x, const float* dout, const float* w_last, float* dwt_iohw, double* c16_out /* [Cin][16],
                             nullable */, int n, int hc, int wc, int cin, int c0, void* ws, size_t ws_bytes, rd_stream_t s);
/* The same with x = act(BN(z)) evaluated on load (x = z, the pre-BN tensor of the block that feeds the up-convolution): that
 * block's activation need not be a tensor either.  rd_tail_t16: T [pixels][16] = act(BN(z)) . V on the exact-f32 matrix pipe
 * (mean = NULL: z is the activation itself) -- the 1x1 convolution of the forward below without a packed operand. */
int rd_convt_last_bwd_weight_bn(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float slope, const float* slope_dev, const float* dout, const float* w_last, float* dwt_iohw,
                                double* c16_out, int n, int hc, int wc, int cin, int c0, void* ws, size_t ws_bytes, rd_stream_t s);
int rd_tail_t16(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                const float* slope_dev, const float* V, float* t16, long long pixels, int cin, rd_stream_t s);
/* FORWARD of the tail without its input tensor.  s = up-convolution(x_coarse) + bias_t + act(BN(z)) is what the reference feeds
 * the last convolution (lib/UNet.py:218-227); here
 *   out[q] = conv_last(act(BN(z)))[q] + sum_{p': d = q - 2p' in [-1,2]^2} T[p'][d] + sum_{tap: q + off(tap) inside} B9[tap] + bias
 *            (+ x[:, 0], the outer residual)
 * with T [N, H/2, W/2, 16] = rd_conv1x1_fwd(x_coarse, VT): z (level 0's pre-BN tensor, which exists anyway) is read once with BN
 * + activation applied on load, the up-convolution's 64-channel full-resolution output is neither written nor read.
 * rd_conv3x3_last_bwd_weight_tail: the last convolution's weight / bias gradient in the same terms -- the act(BN(z)) part from
 * z and dout, the up-convolution part from the correlations C16 of rd_convt_last_bwd_weight, the bias part from dout alone.
 * C in {16, 32, 64}; == rd_conv3x3_last_fwd / rd_conv3x3_last_bwd_weight on the materialised s up to fp32 rounding. */
int rd_conv3x3_last_fwd_tail(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float slope, const float* slope_dev, const float* t16, const float* b9, const float* w_last,
                             const float* bias, const float* x_nchw, int x_channels, float* out, int n, int h, int w, int c,
                             rd_stream_t s);
/* The head of the backward in one pass over z: the partial sums of rd_conv3x3_last_bwd_weight_tail (wpartial:
 * rd_conv3x3_last_bwd_tail_blocks() rows of 9*C + 9 doubles, finished by rd_tail_wl_finish once C16 is there) AND the BN-backward
 * statistics of level 0 that rd_conv3x3_last_bwd_data_bnstats emits (part / rows_out as there; the data gradient itself is
 * evaluated per element and not stored). */
int rd_conv3x3_last_bwd_tail_blocks(int n, int h, int w);
int rd_conv3x3_last_bwd_tail_fused(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                   float slope, const float* slope_dev, const float* dout, const float* w_last, double* wpartial,
                                   float* part, size_t part_floats, int* rows_out, int n, int h, int w, int c, rd_stream_t s);
int rd_tail_wl_finish(const double* wpartial, int nb, const double* c16, const float* wt_iohw, const float* bias_t, float* dw,
                      float* dbias, int cin, int c, rd_stream_t s);
size_t rd_conv3x3_last_bwd_weight_tail_ws_bytes(int n, int h, int w, int c);
int rd_conv3x3_last_bwd_weight_tail(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                    float slope, const float* slope_dev, const float* dout, const double* c16, const float* wt_iohw,
                                    const float* bias_t, float* dw, float* dbias, int n, int h, int w, int cin, int c, void* ws,
                                    size_t ws_bytes, rd_stream_t s);
size_t rd_conv3x3_last_bwd_weight_ws_bytes(int n, int h, int w, int c);
/* dw[1][C][3][3], dbias[1] (nullable) */
int rd_conv3x3_last_bwd_weight(const float* s_in, const float* dout, float* dw_oihw, float* dbias, int n, int h, int w,
                               int c, void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- ConvTranspose2d(k=2, s=2) + bias, skip ADD fused (lib/UNet.py:21,96-101,219-224)
 * out[N,2H,2W,Cout] = convT(x[N,H,W,Cin]) + bias + skip   (skip nullable)               */
int rd_convt2x2_fwd(const float* x, const float* wtf, const float* bias, const float* skip, float* out, int n, int h,
                    int w, int cin, int cout, rd_stream_t s);
/* Same, but the skip operand is given as the encoder level's PRE-BatchNorm conv output z_skip[N,2H,2W,Cout] plus that
 * level's BN/activation parameters; the epilogue recomputes skip = act(gamma*(z-mean)*invstd + beta) (identical
 * arithmetic to rd_bn_act_pool_fwd), so the encoder never has to write its full-resolution activation
 * (rd_bn_act_pool_fwd with a == NULL).  slope / slope_dev as in rd_bn_act_pool_fwd. */
int rd_convt2x2_fwd_bnskip(const float* x, const float* wtf, const float* bias, const float* z_skip, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float slope, const float* slope_dev,
                           float* out, int n, int h, int w, int cin, int cout, rd_stream_t s);
/* dx[N,H,W,Cin] from dout[N,2H,2W,Cout] */
int rd_convt2x2_bwd_data(const float* dout, const float* wtd, float* dx, int n, int h, int w, int cin, int cout,
                         rd_stream_t s);
size_t rd_convt2x2_bwd_weight_ws_bytes(int n, int h, int w, int cin, int cout);
/* dw[Cin][Cout][2][2] (torch layout) */
int rd_convt2x2_bwd_weight(const float* x, const float* dout, float* dw_iohw, int n, int h, int w, int cin, int cout,
                           void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- bilinear up-mode: nn.Upsample(scale_factor=2, 'bilinear') -> conv1x1 (lib/UNet.py:8-9,17-24) -----
 * The 1x1 convolution is applied on the coarse grid (it commutes with the interpolation), then
 * rd_upsample2x_add_fwd interpolates, adds the conv bias and the skip tensor (SkipConnection, lib/UNet.py:96-101).
 * w: torch layout [Cout][Cin][1][1] -> packed wf ([Cout][Cin]) and wt (its transpose [Cin][Cout], nullable). */
int rd_pack_conv1x1_weight(const float* w, float* wf, float* wt, int cout, int cin, rd_stream_t s);
int rd_conv1x1_fwd(const float* x, const float* wf, float* out, long long pixels, int cin, int cout, rd_stream_t s);
int rd_conv1x1_bwd_data(const float* dy, const float* wt, float* dx, long long pixels, int cin, int cout, rd_stream_t s);
size_t rd_conv1x1_bwd_weight_ws_bytes(long long pixels, int cin, int cout);
int rd_conv1x1_bwd_weight(const float* x, const float* dy, float* dw, long long pixels, int cin, int cout, void* ws,
                          size_t ws_bytes, rd_stream_t s);
/* out[N,2H,2W,C] = skip (nullable) + bias (nullable) + bilinear2x(t[N,H,W,C]), align_corners=False */
int rd_upsample2x_add_fwd(const float* t, const float* bias, const float* skip, float* out, int n, int h, int w, int c,
                          rd_stream_t s);
/* adjoint of the interpolation: dt[N,H,W,C] = bilinear2x^T(g[N,2H,2W,C]) (gather form, deterministic) */
int rd_upsample2x_bwd(const float* g, float* dt, int n, int h, int w, int c, rd_stream_t s);

/* per-channel sum over pixels: out[c] = sum_p g[p][c]   (bias gradients) */
size_t rd_channel_sum_ws_bytes(long long pixels, int c);
int rd_channel_sum(const float* g, float* out, long long pixels, int c, void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- BatchNorm2d (training statistics) (lib/UNet.py:45,66,86) ---------------------- */
/* Phase 1: per-channel sums over z[P][C] -> sums[2*C] doubles (sum, sum of squares), so a
 * data-parallel caller can all-reduce them (SyncBN) before phase 2. */
size_t rd_bn_stats_ws_bytes(long long pixels, int c);
int rd_bn_stats_partial(const float* z, double* sums, long long pixels, int c, void* ws, size_t ws_bytes,
                        rd_stream_t s);
/* Phase 2: mean / invstd = 1/sqrt(biased var + eps); running stats <- (1-m)*old + m*new
 * (unbiased var); num_batches_tracked (int64, nullable) += 1.  count = pixels summed.  */
int rd_bn_stats_finalize(const double* sums, double count, float eps, float momentum, float* mean, float* invstd,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, int c, rd_stream_t s);
/* eval mode: mean = running_mean, invstd = 1/sqrt(running_var + eps) */
int rd_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* invstd, int c,
                     rd_stream_t s);

/* a = act(gamma*(z-mean)*invstd + beta), act = LeakyReLU(slope) (slope 0 = ReLU, lib/UNet.py:27-33); with pooling
 * `a` may be NULL (the full-resolution activation is then not written, see rd_convt2x2_fwd_bnskip);
 * if pooled != NULL also the 2x2/2 max-pool of a (lib/UNet.py:161,167): pooled[N,H/2,W/2,C] and
 * idx (uint8, window position 0..3 = dy*2+dx; first maximum in row-major order, NaN wins).
 * zpool (nullable, with pooling): z at the arg-max position [N,H/2,W/2,C] -- lets the backward take the pooled part of
 * the BN statistics from quarter-size tensors (rd_conv3x3_bwd_data_bnstats, mode 2). */
int rd_bn_act_pool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, const float* slope_dev, float* a, float* pooled, uint8_t* idx, float* zpool, int n, int h,
                       int w, int c, rd_stream_t s);

/* Backward of conv -> BN -> act [-> pool].  The gradient wrt `a` is g_full (same resolution,
 * nullable) + unpool(g_pool via idx) (nullable).
 * Phase 1 -> sums[4*C] doubles: sum g', sum g'*xhat (g' = gradient after the activation mask),
 *            sum g_full (un-masked; the bias gradient of the ConvTranspose2d feeding a skip add), and
 *            sum_{y<=0} g*y (per channel; its total is the gradient of nn.PReLU()'s single slope).
 * slope_dev (nullable): device pointer to a learnable PReLU slope (lib/UNet.py:29); when non-NULL it
 *            overrides the immediate `slope` (ReLU = 0, LeakyReLU = 0.01).
 * Phase 2 -> dz; dgamma/dbeta are sums[C..2C) / sums[0..C): written as fp32 by whichever of the two calls is handed
 *            non-NULL dgamma / dbeta (phase 1 saves a launch; phase 2 is for SyncBN, where the LOCAL sums are the parameter
 *            gradients and the all-reduced ones feed dz).  dextra (phase 1, nullable): sums[2C..3C) as fp32, the bias gradient
 *            of the ConvTranspose2d whose output was added to this block's activation.
 * training=0 treats mean/invstd as constants (eval-mode BN). */
/* ---- BN-backward statistics WITHOUT their own pass over z and g: the kernels that PRODUCE the gradient operand g of a
 * conv block (the 3x3 / transposed / last convolution's data gradient) read the block's pre-BN output z in their
 * epilogue and emit per-tile partial rows [rows][4][C] of (sum g', sum g' xhat, sum g, sum_{y<=0} g y) -- the phase-1
 * sums of rd_bn_act_bwd_reduce.  rd_bn_bwd_stats_finalize adds the rows of one or two producers (an encoder block has
 * two: the un-pooled skip gradient, mode 1, and the pooled gradient, mode 2, whose `bn_z` is the `zpool` tensor of
 * rd_bn_act_pool_fwd) in fixed order into `sums[4C]` (fp64) and writes the fp32 parameter gradients.
 * `part` must hold rd_bn_bwd_part_floats(pixels of the OUTPUT, C) floats; *rows_out = rows written (0: this shape has no
 * statistics epilogue -- then only the data gradient was computed and the caller runs rd_bn_act_bwd_reduce). */
size_t rd_bn_bwd_part_floats(long long pixels, int c);
int rd_conv3x3_bwd_data_bnstats(const float* dz, const float* wd, float* dx, int n, int h, int w, int cin, int cout,
                                const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float slope, const float* slope_dev, int mode, float* part,
                                size_t part_floats, int* rows_out, rd_stream_t s);
int rd_convt2x2_bwd_data_bnstats(const float* dout, const float* wtd, float* dx, int n, int h, int w, int cin, int cout,
                                 const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, float slope, const float* slope_dev, float* part, size_t part_floats,
                                 int* rows_out, rd_stream_t s);
int rd_conv3x3_last_bwd_data_bnstats(const float* dout, const float* wt, float* ds, int n, int h, int w, int c,
                                     const float* bn_z, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, float slope, const float* slope_dev, float* part, size_t part_floats,
                                     int* rows_out, rd_stream_t s);
int rd_bn_bwd_stats_finalize(const float* part_a, int rows_a, const float* part_b, int rows_b, int c, double* sums,
                             float* dgamma, float* dbeta, float* dextra, rd_stream_t s);

size_t rd_bn_act_bwd_ws_bytes(int n, int h, int w, int c);
int rd_bn_act_bwd_reduce(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         float slope, const float* slope_dev, const float* g_full, const float* g_pool,
                         const uint8_t* idx, double* sums, float* dgamma, float* dbeta, float* dextra, int n, int h, int w, int c,
                         void* ws, size_t ws_bytes, rd_stream_t s);
int rd_bn_act_bwd_apply(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        float slope, const float* slope_dev, const float* g_full, const float* g_pool,
                        const uint8_t* idx, const double* sums, double count, int training, float* dz, float* dgamma,
                        float* dbeta, int n, int h, int w, int c, rd_stream_t s);

/* ---- masked, de-normalised L1 (lib/Trainer.py:87-100, lib/data_normalization.py:29-38) */
/* sums[0] = sum over valid pixels |(yp*std_i+mean_i) - (y*std_i+mean_i)|, sums[1] = #valid.
 * yp, y: [N,1,H,W]; mask: uint8 (torch.bool) [N,1,H,W]; mean, std: [N] fp32. */
size_t rd_masked_l1_ws_bytes(long long numel);
int rd_masked_l1_partial(const float* yp, const float* y, const uint8_t* mask, const float* mean, const float* std,
                         double* sums, int n, long long pixels_per_sample, void* ws, size_t ws_bytes, rd_stream_t s);
/* loss[0] = (float)(sums[0]/numel_total) * numel_total / sums[1];
 * dyp = gout * std_i * sign(p-t) * mask / sums[1]     (dyp nullable; gout = *gout_dev, a DEVICE scalar so
 * autograd's upstream gradient never has to be read on the host; NULL means 1) */
int rd_masked_l1_finish(const float* yp, const float* y, const uint8_t* mask, const float* mean, const float* std,
                        const double* sums, double numel_total, const float* gout_dev, float* loss, float* dyp, int n,
                        long long pixels_per_sample, rd_stream_t s);

/* ---- torch.optim.Adam step over a flat buffer (lib/utils.py:329-331) ---------------- */
/* g += wd*p; m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps) */
/* beta1/beta2 are doubles so that (1-beta) is formed from the host's double scalars exactly like torch does */
int rd_adam_step(float* p, const float* g, float* m, float* v, long long numel, double beta1, double beta2, float eps,
                 float weight_decay, float step_size, float bc2_sqrt, float grad_scale, rd_stream_t s);

/* The same step with its scalars in DEVICE memory: scalars_dev[8] = {(float)(1-beta1), (float)beta2, (float)(1-beta2), eps, weight_decay,
 * step_size, bc2_sqrt, grad_scale} -- bit-identical to rd_adam_step called with those values.  This is the launch a captured HIP graph
 * replays (resdepth_amd/graph.py: one graph launch per lib/Trainer.py:212-222 iteration); the host rewrites the scalars before each
 * replay (step count of the bias corrections, learning-rate schedule), the graph itself never changes. */
int rd_adam_step_dev(float* p, const float* g, float* m, float* v, long long numel, const float* scalars_dev, rd_stream_t s);

/* ---- torch.optim.SGD step over a flat buffer (lib/utils.py:332-334: SGD(lr, weight_decay)) ---- */
/* g = grad_scale*g + wd*p;  momentum != 0: buf = first_step ? g : momentum*buf + (1-dampening)*g;  g = nesterov ? g + momentum*buf : buf;
 * p -= lr*g.   momentum_buf may be NULL when momentum == 0 (the reference's configuration). */
int rd_sgd_step(float* p, const float* g, float* momentum_buf, long long numel, float lr, float weight_decay, float momentum,
                float dampening, int nesterov, int first_step, float grad_scale, rd_stream_t s);

/* ---- tiled inference: linear blend of overlapping tiles (lib/evaluation.py:460-567) ----------- */
/* For every tile i (in tile order per raster pixel => deterministic fp64 accumulation order, no atomics; batches of up to
 * 64 tiles are ONE launch, larger ones one launch per tile):
 *   raster[y_i + r][x_i + c] += (double)(float)(pred[i][r][c] * std[i] + mean[i]) * w_i(r, c)
 * with w_i = the reference's _get_blend_weights(tile_size, stride, ulx, uly, lrx, lry) (separable linear ramps,
 * np.linspace(0,1,overlap)).  pos = int32 [n][2] (offset_y, offset_x); reg = int32 [n][4] (uly, ulx, lry, lrx). */
int rd_blend_accumulate(const float* pred, const float* mean, const float* std, const int* pos, const int* reg, int n,
                        int tile_size, int stride, double* raster, int rows, int cols, rd_stream_t s);

/* Host side of the sweep's raster read-back (lib/evaluation.py:510-513 returns the blended raster as a host array).  A multi-GPU
 * sweep lets every rank write ITS band of the raster straight into one host buffer shared by the ranks of the node (POSIX
 * shared memory mapped by each of them): rd_host_register page-locks and maps [p, p + bytes) of such a mapping for the
 * current device (hipHostRegister, portable + mapped), rd_copy_to_host_async enqueues device -> host on `s` (hipMemcpyAsync:
 * asynchronous for page-locked destinations; knob d2h_blocks > 0 selects a small-grid copy kernel storing into the mapped range
 * instead, measured slower), rd_host_unregister undoes the registration.  Plain pointers; no torch types. */
int rd_host_register(void* p, size_t bytes);
int rd_host_unregister(void* p);
int rd_copy_to_host_async(void* dst_host, const void* src_dev, size_t bytes, rd_stream_t s);

/* ---- training-sample assembly from rasters resident in HBM (lib/DsmOrthoDataset.py:161-291, lib/torch_transforms.py) */
/* sums[i] = (sum, count) over the T x T patch at pos[i] = (y, x) of the planes plane_idx[i*P .. i*P+P-1] of a planar
 * raster stack (plane stride in floats), skipping pixels equal to `nodata` when use_nodata (the masked patch mean the
 * reference takes with np.ma.mean).  One block per patch, fixed-order reduction. */
int rd_patch_sums(const float* planes, long long plane_stride, const int* plane_idx, int p_per_patch, const int* pos,
                  int n, int tile, int width, float nodata, int use_nodata, double* sums, rd_stream_t s);
/* input[i] = cat((dsm_in - dsm_mean[i]) / dsm_std, (ortho[pair] - ortho_mean[i]) / ortho_std), target[i] =
 * (dsm_gt - dsm_mean[i]) / dsm_std, mask[i] = (dsm_gt != 0) & (dsm_gt != nodata), all augmented by
 * rot90(k) -> flipud -> fliplr with aug[i] = k | flip_v << 2 | flip_h << 3 (dsm_gt / target / mask nullable together). */
int rd_assemble_patches(const float* dsm_in, const float* dsm_gt, const float* ortho_planes, long long plane_stride,
                        const int* pair_idx, int views, const int* pos, const int* aug, const float* dsm_mean,
                        float dsm_std, const float* ortho_mean, float ortho_std, float nodata, int n, int tile, int width,
                        float* input, float* target, uint8_t* mask, rd_stream_t s);

/* ---- masked residual statistics of a refined DSM (lib/evaluation.py:11-131) ------------------------ */
/* residual r = raster - gt where raster != nodata, gt != nodata, mask (nullable) != 0 and, if threshold > 0,
 * |r| <= threshold.  out[8] (doubles, device): count, max, min, MAE, RMSE, absolute median, median,
 * NMAD = 1.4826 * median|r - absolute median| (exact medians; even counts average the two middle values). */
size_t rd_residual_stats_ws_bytes(long long n);
int rd_residual_stats(const double* raster, const float* gt, const uint8_t* mask, long long n, double nodata,
                      double threshold, double* out, void* ws, size_t ws_bytes, rd_stream_t s);

/* ---- layout helpers ----------------------------------------------------------------- */
int rd_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, rd_stream_t s);
int rd_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, rd_stream_t s);

/* ---- built-in kernel timing (HIP events on the launch stream) ----------------------- */
/* When enabled every kernel launch is bracketed by hipEventRecord on its own stream and
 * accumulated per kernel class; rd_prof_collect synchronises the recorded events. */
#define RD_PROF_MAX_CLASSES 64
typedef struct {
    char name[64];   /* "<operation>|<kernel symbol / tile>" */
    long long launches;
    double ms;      /* summed event-to-event duration */
    double flops;   /* summed algorithmic FLOPs declared at launch */
    double bytes;   /* summed algorithmic HBM bytes declared at launch */
} rd_prof_entry;
/* Diagnosis knobs (tile-shape / kernel-selection overrides used by scripts/; never needed in production).  Names:
 * mfma_f32 nt_tile nt_halo nt_skew nt_splitk tn_tile tn_blocks tn_split wg_strip wg_minblocks wg_blocks wg_occ convt_patch edge_conv
 * rows_blocks last_blocks nt_epi d2h_blocks
 * (resdepth_amd/csrc/rd_common.h: TuneKey).  Also settable at load time: RD_TUNE="name=value,..." */
int rd_tune_set(const char* name, int value);
int rd_tune_get(const char* name, int* value);
int rd_prof_enable(int level);   /* 0 off; 1 = the MFMA (roofline) kernel classes only; 2 = every kernel class */
int rd_prof_reset(void);
int rd_prof_collect(rd_prof_entry* out, int max_entries); /* returns number of classes, <0 on error */

#ifdef __cplusplus
}
#endif
#endif /* RESDEPTH_HIP_H */
