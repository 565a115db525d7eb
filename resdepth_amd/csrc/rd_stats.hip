// Masked residual statistics of a refined DSM against the ground truth on the GPU (lib/evaluation.py:11-131:
// compute_residuals + get_statistics): count, max, min, MAE, RMSE and the three medians (absolute median, median,
// NMAD = 1.4826 * median|r - absolute_median|), optionally after truncating |r| > threshold.
//
// Medians are exact: an 8-pass radix select over the order-preserving 64-bit keys of the fp64 values, both middle
// ranks at once (even counts average the two middle values like np.ma.median).  Histograms are integer counters, so
// the result is independent of scheduling; sums go through fixed-order block partials.  No host synchronisation.
#include "rd_common.h"

namespace rd {

struct SelState {
    unsigned long long prefix[2];
    long long rank[2];
    long long count;
    double shift;
};

__device__ __forceinline__ unsigned long long key_of(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ double pick_value(double r, int mode, double shift) {
    return mode == 0 ? r : (mode == 1 ? fabs(r) : fabs(r - shift));
}

__global__ __launch_bounds__(256) void residual_kernel(const double* __restrict__ raster, const float* __restrict__ gt,
                                                       const uint8_t* __restrict__ mask, long n, double nodata,
                                                       double thr, double* __restrict__ r, uint8_t* __restrict__ valid,
                                                       double* __restrict__ partial) {
    __shared__ double red[5 * 256];
    double cnt = 0.0, sa = 0.0, sq = 0.0, mn = INFINITY, mx = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double a = raster[i], g = (double)gt[i];
        const double d = a - g;
        bool ok = (a != nodata) && (g != nodata) && (!mask || mask[i]);
        if (thr > 0.0) ok = ok && (fabs(d) <= thr);
        r[i] = d;
        valid[i] = ok ? 1 : 0;
        if (ok) {
            cnt += 1.0;
            sa += fabs(d);
            sq += d * d;
            mn = fmin(mn, d);
            mx = fmax(mx, d);
        }
    }
    const int t = threadIdx.x;
    red[t] = cnt; red[256 + t] = sa; red[512 + t] = sq; red[768 + t] = mn; red[1024 + t] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) {
            red[t] += red[t + off];
            red[256 + t] += red[256 + t + off];
            red[512 + t] += red[512 + t + off];
            red[768 + t] = fmin(red[768 + t], red[768 + t + off]);
            red[1024 + t] = fmax(red[1024 + t], red[1024 + t + off]);
        }
        __syncthreads();
    }
    if (t == 0)
        for (int q = 0; q < 5; ++q) partial[blockIdx.x * 5 + q] = red[q * 256];
}

__global__ void moments_finish_kernel(const double* __restrict__ partial, int nb, double* __restrict__ out,
                                      SelState* __restrict__ st) {
    if (threadIdx.x || blockIdx.x) return;
    double cnt = 0.0, sa = 0.0, sq = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int b = 0; b < nb; ++b) {
        cnt += partial[b * 5];
        sa += partial[b * 5 + 1];
        sq += partial[b * 5 + 2];
        mn = fmin(mn, partial[b * 5 + 3]);
        mx = fmax(mx, partial[b * 5 + 4]);
    }
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    out[0] = cnt;
    out[1] = cnt > 0 ? mx : nanv;
    out[2] = cnt > 0 ? mn : nanv;
    out[3] = cnt > 0 ? sa / cnt : nanv;
    out[4] = cnt > 0 ? sqrt(sq / cnt) : nanv;
    st->count = (long long)cnt;
}

__global__ void select_init_kernel(SelState* st, unsigned* hist, const double* shift_src) {
    const int t = threadIdx.x;
    for (int i = t; i < 512; i += blockDim.x) hist[i] = 0u;
    if (t == 0) {
        st->prefix[0] = st->prefix[1] = 0ull;
        st->rank[0] = (st->count - 1) / 2;
        st->rank[1] = st->count / 2;
        st->shift = shift_src ? *shift_src : 0.0;
    }
}

__global__ __launch_bounds__(256) void select_hist_kernel(const double* __restrict__ r, const uint8_t* __restrict__ valid,
                                                          long n, int mode, int pass, const SelState* __restrict__ st,
                                                          unsigned* __restrict__ hist) {
    __shared__ unsigned lh[512];
    for (int i = threadIdx.x; i < 512; i += 256) lh[i] = 0u;
    __syncthreads();
    const double shift = st->shift;
    const unsigned long long p0 = st->prefix[0], p1 = st->prefix[1];
    const int hs = 8 * (pass + 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (!valid[i]) continue;
        const unsigned long long k = key_of(pick_value(r[i], mode, shift));
        const unsigned long long hi = pass == 7 ? 0ull : (k >> hs);
        const unsigned b = (unsigned)((k >> (8 * pass)) & 255ull);
        if (hi == p0) atomicAdd(&lh[b], 1u);
        if (hi == p1) atomicAdd(&lh[256 + b], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

__global__ void select_pick_kernel(SelState* st, unsigned* hist) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int s = 0; s < 2; ++s) {
            long long rank = st->rank[s], below = 0;
            int bin = 255;
            for (int b = 0; b < 256; ++b) {
                const long long c = hist[s * 256 + b];
                if (rank < below + c) {
                    bin = b;
                    break;
                }
                below += c;
            }
            st->prefix[s] = (st->prefix[s] << 8) | (unsigned long long)bin;
            st->rank[s] = rank - below;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) hist[i] = 0u;
}

__global__ void select_finish_kernel(const SelState* st, double* dst, double scale) {
    if (threadIdx.x || blockIdx.x) return;
    const double nanv = __longlong_as_double(0x7ff8000000000000ll);
    dst[0] = st->count > 0 ? scale * 0.5 * (unkey(st->prefix[0]) + unkey(st->prefix[1])) : nanv;
}

static int stats_grid(long n) {
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace rd

using namespace rd;

extern "C" {

size_t rd_residual_stats_ws_bytes(long long n) {
    const size_t nn = (size_t)((n + 15) / 16 * 16);
    return nn * sizeof(double) + nn + (size_t)stats_grid(n) * 5 * sizeof(double) + 512 * sizeof(unsigned) + 256;
}

int rd_residual_stats(const double* raster, const float* gt, const uint8_t* mask, long long n, double nodata,
                      double threshold, double* out, void* ws, size_t ws_bytes, rd_stream_t s_) {
    RD_REQUIRE(raster && gt && out && n > 0, "rd_residual_stats: bad arguments");
    if (!ws || ws_bytes < rd_residual_stats_ws_bytes(n)) {
        set_error("rd_residual_stats: workspace too small (%zu < %zu)", ws_bytes, rd_residual_stats_ws_bytes(n));
        return RD_ERR_WS;
    }
    hipStream_t s = (hipStream_t)s_;
    const size_t nn = (size_t)((n + 15) / 16 * 16);
    const int nb = stats_grid(n);
    char* base = (char*)ws;
    double* r = (double*)base;
    uint8_t* valid = (uint8_t*)(base + nn * sizeof(double));
    double* partial = (double*)(base + nn * sizeof(double) + nn);
    unsigned* hist = (unsigned*)((char*)partial + (size_t)nb * 5 * sizeof(double));
    SelState* st = (SelState*)((char*)hist + 512 * sizeof(unsigned));
    ProfScope ps(s, "residual_stats", 0, 13.0 * n + 24.0 * 9.0 * n);
    RD_LAUNCH(residual_kernel, dim3(nb), dim3(256), 0, s, raster, gt, mask, (long)n, nodata, threshold, r, valid,
                       partial);
    RD_LAUNCH(moments_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)partial, nb, out, st);
    // out: 0 count, 1 max, 2 min, 3 MAE, 4 RMSE, 5 absolute_median, 6 median, 7 NMAD
    const int modes[3] = {1, 0, 2};
    const int dst[3] = {5, 6, 7};
    for (int m = 0; m < 3; ++m) {
        RD_LAUNCH(select_init_kernel, dim3(1), dim3(256), 0, s, st, hist, m == 2 ? (const double*)(out + 5) : nullptr);
        for (int pass = 7; pass >= 0; --pass) {
            RD_LAUNCH(select_hist_kernel, dim3(nb), dim3(256), 0, s, (const double*)r, (const uint8_t*)valid,
                               (long)n, modes[m], pass, (const SelState*)st, hist);
            RD_LAUNCH(select_pick_kernel, dim3(1), dim3(256), 0, s, st, hist);
        }
        RD_LAUNCH(select_finish_kernel, dim3(1), dim3(64), 0, s, (const SelState*)st, out + dst[m],
                           m == 2 ? 1.4826 : 1.0);
    }
    RD_LAUNCH_CHECK("residual_stats");
    return RD_OK;
}

}  // extern "C"
