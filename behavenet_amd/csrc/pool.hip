// nn.MaxPool2d(return_indices=True, ceil_mode) / nn.MaxUnpool2d of the 'max_pooling' architectures
// (reference aes.py:99-110,196-208,281-294,460-464).  HBM-bound element-wise kernels.
//
// Indices follow torch: position h * W + w inside the (H x W) input plane of the pooled tensor;
// they are kept as int32 on the device.  Ties go to the first maximum in row-major window order,
// NaNs propagate (torch's `val > max || isnan(val)`).
#include "bn_common.h"
#include <math.h>

// y[plane][ho][wo] = max over the (clipped) window; idx = its position in the input plane
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x,
                                                     float* __restrict__ y, int* __restrict__ idx,
                                                     size_t total, int H, int W, int Ho, int Wo,
                                                     int k, int s, int pt, int pl) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int wo = (int)(i % Wo);
    const int ho = (int)((i / Wo) % Ho);
    const size_t plane = i / ((size_t)Wo * Ho);
    const float* xp = x + plane * H * W;
    int h0 = ho * s - pt, w0 = wo * s - pl;
    const int h1 = min(h0 + k, H), w1 = min(w0 + k, W);
    h0 = max(h0, 0);
    w0 = max(w0, 0);
    float best = -INFINITY;
    int bi = h0 * W + w0;
    for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) {
            const float v = xp[h * W + w];
            if (v > best || isnan(v)) { best = v; bi = h * W + w; }
        }
    y[i] = best;
    idx[i] = bi;
}

// dx[plane][h][w] = sum of dy over the windows whose maximum sits at (h, w): gather form, no
// atomics, fixed order
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ dy,
                                                     const int* __restrict__ idx,
                                                     float* __restrict__ dx, size_t total, int H,
                                                     int W, int Ho, int Wo, int k, int s, int pt,
                                                     int pl) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % W);
    const int h = (int)((i / W) % H);
    const size_t plane = i / ((size_t)W * H);
    const int me = h * W + w;
    // windows (ho, wo) that contain (h, w): ho*s - pt <= h < ho*s - pt + k
    const int ho_lo = max(0, (h + pt - k + s) / s), ho_hi = min(Ho - 1, (h + pt) / s);
    const int wo_lo = max(0, (w + pl - k + s) / s), wo_hi = min(Wo - 1, (w + pl) / s);
    const size_t ob = plane * Ho * Wo;
    float v = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
        for (int wo = wo_lo; wo <= wo_hi; ++wo)
            if (idx[ob + ho * Wo + wo] == me) v += dy[ob + ho * Wo + wo];
    dx[i] = v;
}

// y (pre-zeroed, planes of Ho x Wo) : y[plane][idx] = x[plane][i]
__global__ __launch_bounds__(256) void k_maxunpool_fwd(const float* __restrict__ x,
                                                       const int* __restrict__ idx,
                                                       float* __restrict__ y, size_t total,
                                                       int in_plane, int out_plane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t plane = i / in_plane;
    const int t = idx[i];
    if (t >= 0 && t < out_plane) y[plane * out_plane + t] = x[i];
}

// dx[plane][i] = dy[plane][idx]
__global__ __launch_bounds__(256) void k_maxunpool_bwd(const float* __restrict__ dy,
                                                       const int* __restrict__ idx,
                                                       float* __restrict__ dx, size_t total,
                                                       int in_plane, int out_plane) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t plane = i / in_plane;
    const int t = idx[i];
    dx[i] = (t >= 0 && t < out_plane) ? dy[plane * out_plane + t] : 0.f;
}

static inline unsigned pool_blocks(size_t total) { return (unsigned)((total + 255) / 256); }

extern "C" int bn_maxpool2d_fwd(const float* x, float* y, int* idx, int planes, int H, int W,
                                int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                                bn_stream_t stream) {
    if (!x || !y || !idx) return BN_E_BADARG;
    if (planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || k <= 0 || stride <= 0 ||
        pad_t < 0 || pad_l < 0)
        return BN_E_BADARG;
    const size_t total = (size_t)planes * Ho * Wo;
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(pool_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       x, y, idx, total, H, W, Ho, Wo, k, stride, pad_t, pad_l);
    BN_LAUNCH_CHECK();
    return 0;
}

extern "C" int bn_maxpool2d_bwd(const float* dy, const int* idx, float* dx, int planes, int H,
                                int W, int Ho, int Wo, int k, int stride, int pad_t, int pad_l,
                                bn_stream_t stream) {
    if (!dy || !dx || !idx) return BN_E_BADARG;
    if (planes <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || k <= 0 || stride <= 0)
        return BN_E_BADARG;
    const size_t total = (size_t)planes * H * W;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(pool_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       dy, idx, dx, total, H, W, Ho, Wo, k, stride, pad_t, pad_l);
    BN_LAUNCH_CHECK();
    return 0;
}

extern "C" int bn_maxunpool2d_fwd(const float* x, const int* idx, float* y, int planes,
                                  int in_plane, int out_plane, bn_stream_t stream) {
    if (!x || !y || !idx) return BN_E_BADARG;
    if (planes <= 0 || in_plane <= 0 || out_plane <= 0) return BN_E_BADARG;
    hipError_t e = hipMemsetAsync(y, 0, (size_t)planes * out_plane * sizeof(float),
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const size_t total = (size_t)planes * in_plane;
    hipLaunchKernelGGL(k_maxunpool_fwd, dim3(pool_blocks(total)), dim3(256), 0,
                       (hipStream_t)stream, x, idx, y, total, in_plane, out_plane);
    BN_LAUNCH_CHECK();
    return 0;
}

extern "C" int bn_maxunpool2d_bwd(const float* dy, const int* idx, float* dx, int planes,
                                  int in_plane, int out_plane, bn_stream_t stream) {
    if (!dy || !dx || !idx) return BN_E_BADARG;
    if (planes <= 0 || in_plane <= 0 || out_plane <= 0) return BN_E_BADARG;
    const size_t total = (size_t)planes * in_plane;
    hipLaunchKernelGGL(k_maxunpool_bwd, dim3(pool_blocks(total)), dim3(256), 0,
                       (hipStream_t)stream, dy, idx, dx, total, in_plane, out_plane);
    BN_LAUNCH_CHECK();
    return 0;
}
