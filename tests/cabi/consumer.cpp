// A consumer of libresdepth_hip.so that is NOT PyTorch: plain HIP runtime + include/resdepth_hip.h, the way a C / C++ /
// cgo / JNI host would bind the library.  One 3x3 convolution layer (lib/UNet.py:4-5,44) forward, data
// gradient and weight gradient on device buffers it allocates itself, then the block behind it (training-mode BatchNorm
// statistics from the convolution's epilogue, BN + LeakyReLU + max-pool), checked against direct loops in double on the host;
// the optional split-K scratch of the 8 x 8 convolution (register for (device, stream) / run / un-register: same bits each way);
// then the error contract (non-zero return + rd_last_error_string).  Test infrastructure: built and run by
// tests/test_cabi_consumer_gpu.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "resdepth_hip.h"

#define HIP_OK(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            return 2;                                                                         \
        }                                                                                     \
    } while (0)
#define RD_OK_(call)                                                                          \
    do {                                                                                      \
        int e_ = (call);                                                                      \
        if (e_ != 0) {                                                                        \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, e_, rd_last_error_string());        \
            return 3;                                                                         \
        }                                                                                     \
    } while (0)

static unsigned g_seed = 12345u;
static float rnd() {      // uniform in (-1, 1), the same on every host
    g_seed = g_seed * 1664525u + 1013904223u;
    return ((g_seed >> 8) & 0xffff) / 32768.0f - 1.0f;
}

template <class T>
static T* dev_alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) ? n * sizeof(T) : 4) != hipSuccess) return nullptr;
    return static_cast<T*>(p);
}

static double max_rel(const std::vector<float>& got, const std::vector<double>& want) {
    double scale = 0, err = 0;
    for (double v : want) scale = std::fmax(scale, std::fabs(v));
    for (size_t i = 0; i < want.size(); ++i) err = std::fmax(err, std::fabs(got[i] - want[i]));
    return err / (scale + 1e-30);
}

int main() {
    const int N = 2, H = 16, W = 24, CI = 8, CO = 12;
    std::vector<float> x((size_t)N * H * W * CI), wt((size_t)CO * CI * 9), dz((size_t)N * H * W * CO);
    for (float& v : x) v = rnd();
    for (float& v : wt) v = rnd() * 0.2f;
    for (float& v : dz) v = rnd();

    // host reference in double: z[n,y,x,co] = sum_{ci,ky,kx} x[n,y+ky-1,x+kx-1,ci] w[co,ci,ky,kx]   (zero padding)
    std::vector<double> z_ref((size_t)N * H * W * CO, 0.0), dx_ref((size_t)N * H * W * CI, 0.0), dw_ref((size_t)CO * CI * 9, 0.0);
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int xx = 0; xx < W; ++xx)
                for (int co = 0; co < CO; ++co) {
                    const double g = dz[(((size_t)n * H + y) * W + xx) * CO + co];
                    double acc = 0;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yy = y + ky - 1, xs = xx + kx - 1;
                            if (yy < 0 || yy >= H || xs < 0 || xs >= W) continue;
                            for (int ci = 0; ci < CI; ++ci) {
                                const size_t xi = (((size_t)n * H + yy) * W + xs) * CI + ci, wi = (((size_t)co * CI + ci) * 3 + ky) * 3 + kx;
                                acc += (double)x[xi] * wt[wi];
                                dx_ref[xi] += g * wt[wi];
                                dw_ref[wi] += g * x[xi];
                            }
                        }
                    z_ref[(((size_t)n * H + y) * W + xx) * CO + co] = acc;
                }

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *d_x = dev_alloc<float>(x.size()), *d_w = dev_alloc<float>(wt.size()), *d_dz = dev_alloc<float>(dz.size());
    float *d_z = dev_alloc<float>(z_ref.size()), *d_dx = dev_alloc<float>(dx_ref.size()), *d_dw = dev_alloc<float>(dw_ref.size());
    const size_t wf_bytes = rd_packed_weight_bytes(CO, 9, CI), wd_bytes = rd_packed_weight_bytes(CI, 9, CO);
    const size_t ws_bytes = rd_conv3x3_bwd_weight_ws_bytes(N, H, W, CI, CO);
    char *d_wf = dev_alloc<char>(wf_bytes), *d_wd = dev_alloc<char>(wd_bytes), *d_ws = dev_alloc<char>(ws_bytes);
    if (!d_x || !d_w || !d_dz || !d_z || !d_dx || !d_dw || !d_wf || !d_wd || !d_ws) {
        std::fprintf(stderr, "hipMalloc failed\n");
        return 2;
    }
    HIP_OK(hipMemcpyAsync(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_w, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_dz, dz.data(), dz.size() * 4, hipMemcpyHostToDevice, stream));

    RD_OK_(rd_pack_conv3x3_weight(d_w, (float*)d_wf, (float*)d_wd, CO, CI, stream));
    RD_OK_(rd_conv3x3_fwd(d_x, (const float*)d_wf, d_z, N, H, W, CI, CO, stream));
    RD_OK_(rd_conv3x3_bwd_data(d_dz, (const float*)d_wd, d_dx, N, H, W, CI, CO, stream));
    RD_OK_(rd_conv3x3_bwd_weight(d_x, d_dz, d_dw, N, H, W, CI, CO, d_ws, ws_bytes, stream));

    std::vector<float> z(z_ref.size()), dx(dx_ref.size()), dw(dw_ref.size());
    HIP_OK(hipMemcpyAsync(z.data(), d_z, z.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(dx.data(), d_dx, dx.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(dw.data(), d_dw, dw.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    int bad = 0;
    const double e_z = max_rel(z, z_ref), e_dx = max_rel(dx, dx_ref), e_dw = max_rel(dw, dw_ref);
    std::printf("conv3x3 forward       max |err| / max |ref| = %.3g\n", e_z);
    std::printf("conv3x3 data gradient max |err| / max |ref| = %.3g\n", e_dx);
    std::printf("conv3x3 weight grad.  max |err| / max |ref| = %.3g\n", e_dw);
    const double tol = 2e-6;      // fp32 results (the tolerance of tests/test_ops_gpu.py)
    if (!(e_z <= tol) || !(e_dx <= tol) || !(e_dw <= tol)) bad = 1;

    // ---- the block behind the convolution: training-mode BatchNorm statistics from the convolution's epilogue, BN apply +
    // LeakyReLU + 2x2 max-pool (lib/UNet.py:44-47,161), against the host in double
    {
        float *d_mean = dev_alloc<float>(CO), *d_invstd = dev_alloc<float>(CO), *d_gamma = dev_alloc<float>(CO), *d_beta = dev_alloc<float>(CO);
        float *d_p = dev_alloc<float>((size_t)N * (H / 2) * (W / 2) * CO), *d_z2 = dev_alloc<float>(z_ref.size());
        unsigned char* d_idx = dev_alloc<unsigned char>((size_t)N * (H / 2) * (W / 2) * CO);
        const size_t sws = rd_conv3x3_fwd_stats_ws_bytes(N, H, W, CI, CO);
        char* d_sws = dev_alloc<char>(sws);
        std::vector<float> gamma(CO), beta(CO);
        for (int c = 0; c < CO; ++c) { gamma[c] = 1.0f + 0.5f * rnd(); beta[c] = 0.3f * rnd(); }
        HIP_OK(hipMemcpyAsync(d_gamma, gamma.data(), CO * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_beta, beta.data(), CO * 4, hipMemcpyHostToDevice, stream));
        const float eps = 1e-5f, slope = 0.01f;
        RD_OK_(rd_conv3x3_fwd_bn(d_x, (const float*)d_wf, d_z2, (double)N * H * W, eps, 0.1f, d_mean, d_invstd, nullptr, nullptr, nullptr,
                                 N, H, W, CI, CO, d_sws, sws, stream));
        RD_OK_(rd_bn_act_pool_fwd(d_z2, d_mean, d_invstd, d_gamma, d_beta, slope, nullptr, nullptr, d_p, d_idx, nullptr, N, H, W, CO, stream));
        std::vector<float> mean(CO), invstd(CO), pooled((size_t)N * (H / 2) * (W / 2) * CO);
        HIP_OK(hipMemcpyAsync(mean.data(), d_mean, CO * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(invstd.data(), d_invstd, CO * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(pooled.data(), d_p, pooled.size() * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        const double cnt = (double)N * H * W;
        double e_stat = 0, e_pool = 0, pscale = 0;
        std::vector<double> mu(CO), is(CO);
        for (int c = 0; c < CO; ++c) {
            double s1 = 0, s2 = 0;
            for (size_t p = 0; p < (size_t)N * H * W; ++p) { const double v = z_ref[p * CO + c]; s1 += v; s2 += v * v; }
            mu[c] = s1 / cnt;
            is[c] = 1.0 / std::sqrt(s2 / cnt - mu[c] * mu[c] + (double)eps);
            e_stat = std::fmax(e_stat, std::fabs(mean[c] - mu[c]) * is[c]);
            e_stat = std::fmax(e_stat, std::fabs(invstd[c] / is[c] - 1.0));
        }
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < H / 2; ++y)
                for (int xx = 0; xx < W / 2; ++xx)
                    for (int c = 0; c < CO; ++c) {
                        double best = -1e300;
                        for (int k = 0; k < 4; ++k) {
                            const size_t q = (((size_t)n * H + 2 * y + (k >> 1)) * W + 2 * xx + (k & 1)) * CO + c;
                            double a = (z_ref[q] - mu[c]) * is[c] * gamma[c] + beta[c];
                            a = a > 0 ? a : a * slope;
                            best = std::fmax(best, a);
                        }
                        const double got = pooled[(((size_t)n * (H / 2) + y) * (W / 2) + xx) * CO + c];
                        e_pool = std::fmax(e_pool, std::fabs(got - best));
                        pscale = std::fmax(pscale, std::fabs(best));
                    }
        std::printf("BatchNorm statistics   max relative deviation = %.3g\n", e_stat);
        std::printf("BN + act + max-pool    max |err| / max |ref| = %.3g\n", e_pool / pscale);
        if (!(e_stat <= 1e-5) || !(e_pool / pscale <= 1e-5)) bad = 1;
        for (void* p : {(void*)d_mean, (void*)d_invstd, (void*)d_gamma, (void*)d_beta, (void*)d_p, (void*)d_z2, (void*)d_idx, (void*)d_sws})
            (void)hipFree(p);
    }

    // ---- rd_set_splitk_workspace: the 3x3 convolution on 8 x 8 images (the bottleneck shape, lib/UNet.py:78-93) without a
    // registration, with one for (current device, stream), and after un-registering: the header promises the same bits
    {
        const int n8 = 4, h8 = 8, ci8 = 512, co8 = 64;
        std::vector<float> x8((size_t)n8 * h8 * h8 * ci8), w8((size_t)co8 * ci8 * 9);
        for (float& v : x8) v = rnd();
        for (float& v : w8) v = rnd() * 0.05f;
        std::vector<double> ref((size_t)n8 * h8 * h8 * co8, 0.0);
        for (int n = 0; n < n8; ++n)
            for (int y = 0; y < h8; ++y)
                for (int xx = 0; xx < h8; ++xx)
                    for (int co = 0; co < co8; ++co) {
                        double acc = 0;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int yy = y + ky - 1, xs = xx + kx - 1;
                                if (yy < 0 || yy >= h8 || xs < 0 || xs >= h8) continue;
                                const float* xp = &x8[(((size_t)n * h8 + yy) * h8 + xs) * ci8];
                                for (int ci = 0; ci < ci8; ++ci) acc += (double)xp[ci] * w8[(((size_t)co * ci8 + ci) * 3 + ky) * 3 + kx];
                            }
                        ref[(((size_t)n * h8 + y) * h8 + xx) * co8 + co] = acc;
                    }
        float *d_x8 = dev_alloc<float>(x8.size()), *d_w8 = dev_alloc<float>(w8.size()), *d_z8 = dev_alloc<float>(ref.size());
        const size_t pk = rd_packed_weight_bytes(co8, 9, ci8), sk_bytes = (32u << 20) + (64u << 10);
        char *d_pk = dev_alloc<char>(pk), *d_sk = dev_alloc<char>(sk_bytes);
        if (!d_x8 || !d_w8 || !d_z8 || !d_pk || !d_sk) return 2;
        HIP_OK(hipMemcpyAsync(d_x8, x8.data(), x8.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(d_w8, w8.data(), w8.size() * 4, hipMemcpyHostToDevice, stream));
        RD_OK_(rd_pack_conv3x3_weight(d_w8, (float*)d_pk, nullptr, co8, ci8, stream));
        std::vector<float> z[3] = {std::vector<float>(ref.size()), std::vector<float>(ref.size()), std::vector<float>(ref.size())};
        for (int pass = 0; pass < 3; ++pass) {
            if (pass == 1) RD_OK_(rd_set_splitk_workspace(d_sk, sk_bytes, stream));      // register
            if (pass == 2) RD_OK_(rd_set_splitk_workspace(nullptr, 0, stream));           // un-register
            HIP_OK(hipMemsetAsync(d_z8, 0xff, ref.size() * 4, stream));
            RD_OK_(rd_conv3x3_fwd(d_x8, (const float*)d_pk, d_z8, n8, h8, h8, ci8, co8, stream));
            HIP_OK(hipMemcpyAsync(z[pass].data(), d_z8, ref.size() * 4, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
        }
        const double e8 = max_rel(z[1], ref);
        const bool same = !std::memcmp(z[0].data(), z[1].data(), ref.size() * 4) && !std::memcmp(z[1].data(), z[2].data(), ref.size() * 4);
        std::printf("split-K scratch: 8 x 8 convolution unregistered / registered / un-registered: %s, deviation from the host %.3g\n",
                    same ? "same bits" : "DIFFERENT BITS", e8);
        if (!same || !(e8 <= tol)) bad = 1;
        // the scratch must be device memory of the current device, 256-byte aligned and large enough
        std::vector<char> host_buf(sk_bytes);
        const int r_host = rd_set_splitk_workspace(host_buf.data(), sk_bytes, stream);
        const int r_align = rd_set_splitk_workspace(d_sk + 4, sk_bytes - 4, stream);
        const int r_small = rd_set_splitk_workspace(d_sk, 4096, stream);
        std::printf("split-K scratch: host pointer -> %d, misaligned -> %d, too small -> %d (%s)\n", r_host, r_align, r_small,
                    rd_last_error_string());
        if (r_host != RD_ERR_ARG || r_align != RD_ERR_ARG || r_small != RD_ERR_ARG) bad = 1;
        for (void* p : {(void*)d_x8, (void*)d_w8, (void*)d_z8, (void*)d_pk, (void*)d_sk}) (void)hipFree(p);
    }

    // ---- host side of a sweep's raster read-back (rd_host_register / rd_copy_to_host_async / rd_host_unregister): a plain
    // malloc'ed (page-aligned) host range is page-locked, receives a device buffer asynchronously -- through the runtime's copy and
    // through the library's own small-grid copy kernel (knob d2h_blocks) -- and is released again
    {
        const size_t n = 1u << 20;                                   // doubles: 8 MB
        std::vector<double> src(n);
        for (size_t i = 0; i < n; ++i) src[i] = (double)rnd() * 1000.0 + (double)i;
        double* d_r = dev_alloc<double>(n);
        void* raw = nullptr;
        if (!d_r || posix_memalign(&raw, 4096, n * sizeof(double))) return 2;
        double* h_r = static_cast<double*>(raw);
        HIP_OK(hipMemcpyAsync(d_r, src.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
        RD_OK_(rd_host_register(h_r, n * sizeof(double)));
        int same = 1;
        for (int blocks : {0, 32}) {
            RD_OK_(rd_tune_set("d2h_blocks", blocks));
            std::memset(h_r, 0, n * sizeof(double));
            RD_OK_(rd_copy_to_host_async(h_r, d_r, n * sizeof(double), stream));
            HIP_OK(hipStreamSynchronize(stream));
            same = same && !std::memcmp(h_r, src.data(), n * sizeof(double));
        }
        RD_OK_(rd_tune_set("d2h_blocks", 0));
        RD_OK_(rd_host_unregister(h_r));
        const int r_null = rd_host_register(nullptr, 4096), r_copy = rd_copy_to_host_async(nullptr, d_r, 16, stream);
        std::printf("host read-back: registered range filled by the runtime's copy and by the copy kernel: %s; null range -> %d, null "
                    "destination -> %d\n", same ? "same bytes" : "DIFFERENT BYTES", r_null, r_copy);
        if (!same || r_null != RD_ERR_ARG || r_copy != RD_ERR_ARG) bad = 1;
        std::free(raw);
        (void)hipFree(d_r);
    }

    // error contract: Cin must be a multiple of 4 here -> non-zero return, message names the argument, nothing launched
    const int rc = rd_conv3x3_fwd(d_x, (const float*)d_wf, d_z, N, H, W, 3, CO, stream);
    const char* msg = rd_last_error_string();
    std::printf("rd_conv3x3_fwd(cin = 3) -> %d: %s\n", rc, msg ? msg : "(null)");
    if (rc == 0 || !msg || !std::strstr(msg, "Cin")) bad = 1;
    HIP_OK(hipStreamSynchronize(stream));

    for (void* p : {(void*)d_x, (void*)d_w, (void*)d_dz, (void*)d_z, (void*)d_dx, (void*)d_dw, (void*)d_wf, (void*)d_wd, (void*)d_ws}) (void)hipFree(p);
    (void)hipStreamDestroy(stream);
    std::puts(bad ? "FAILED" : "OK");
    return bad;
}
