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

// ---- 2x2 windows, stride 2, no padding, even maps (round 4: the only pooling the reference's architecture generator
// draws -- possible_max_pool_sizes = [2] -- and the test architecture's): two windows per thread, 16-byte accesses on the
// big map, no divisions per element.  Same winner as the loops above (row-major scan, a later value wins only if
// strictly larger or NaN).  The kernels above: 79-84 us forward, 217 us backward, 80 + 26 us (memset) unpooling for 16
// channels of 128x128 and 256 frames.
__device__ __forceinline__ void pool2_pick(float v, int i, float& best, int& bi) {
    if (v > best || isnan(v)) { best = v; bi = i; }
}
// (act / slope: the activation that follows the pooling -- aes.py:204-211 -- applied on the way out; it is monotonic,
// so the winner is the same)
__global__ __launch_bounds__(256) void k_maxpool_fwd_k2(const float* __restrict__ x, float* __restrict__ y,
                                                        int* __restrict__ idx, size_t pairs, int Ho, int Wo2,
                                                        int W, int act, float slope) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    const int q2 = (int)(i % Wo2);
    const size_t t = i / Wo2;
    const int ho = (int)(t % Ho);
    const size_t plane = t / Ho;
    const int me = 2 * ho * W + 4 * q2;
    const float* r0 = x + plane * (size_t)(2 * Ho) * W + me;
    const float4 a = *reinterpret_cast<const float4*>(r0), b = *reinterpret_cast<const float4*>(r0 + W);
    float b0 = -INFINITY, b1 = -INFINITY;
    int i0 = me, i1 = me + 2;
    pool2_pick(a.x, me, b0, i0); pool2_pick(a.y, me + 1, b0, i0);
    pool2_pick(b.x, me + W, b0, i0); pool2_pick(b.y, me + W + 1, b0, i0);
    pool2_pick(a.z, me + 2, b1, i1); pool2_pick(a.w, me + 3, b1, i1);
    pool2_pick(b.z, me + W + 2, b1, i1); pool2_pick(b.w, me + W + 3, b1, i1);
    reinterpret_cast<float2*>(y)[i] = make_float2(bn_apply_act(b0, act, slope), bn_apply_act(b1, act, slope));
    reinterpret_cast<int2*>(idx)[i] = make_int2(i0, i1);
}
// big[plane][h][w] = small[plane][p][q] where idx[plane][p][q] == h W + w inside window (p, q), 0.0f elsewhere: the
// backward pass of the pooling (small = dy) and the forward pass of the unpooling that undoes it (small = x)
// (yact: the saved output of the activation behind the pooling; small is multiplied by its derivative first)
__global__ __launch_bounds__(256) void k_pool_spread_k2(const float* __restrict__ small, const int* __restrict__ idx,
                                                        float* __restrict__ big, size_t pairs, int Ho, int Wo2,
                                                        int W, const float* __restrict__ yact, int act, float slope) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    const int q2 = (int)(i % Wo2);
    const size_t t = i / Wo2;
    const int ho = (int)(t % Ho);
    const size_t plane = t / Ho;
    const int me = 2 * ho * W + 4 * q2;
    float2 v = reinterpret_cast<const float2*>(small)[i];
    if (yact) {
        const float2 ya = reinterpret_cast<const float2*>(yact)[i];
        v.x *= bn_act_grad_from_output(ya.x, act, slope);
        v.y *= bn_act_grad_from_output(ya.y, act, slope);
    }
    const int2 id = reinterpret_cast<const int2*>(idx)[i];
    float* r0 = big + plane * (size_t)(2 * Ho) * W + me;
    *reinterpret_cast<float4*>(r0) = make_float4(id.x == me ? v.x : 0.f, id.x == me + 1 ? v.x : 0.f,
                                                 id.y == me + 2 ? v.y : 0.f, id.y == me + 3 ? v.y : 0.f);
    *reinterpret_cast<float4*>(r0 + W) = make_float4(id.x == me + W ? v.x : 0.f, id.x == me + W + 1 ? v.x : 0.f,
                                                     id.y == me + W + 2 ? v.y : 0.f, id.y == me + W + 3 ? v.y : 0.f);
}
static inline bool pool_k2_ok(int H, int W, int Ho, int Wo, int k, int stride, int pad_t, int pad_l, const void* a,
                              const void* b, const void* c) {
    return k == 2 && stride == 2 && pad_t == 0 && pad_l == 0 && H == 2 * Ho && W == 2 * Wo && (Wo & 1) == 0 &&
           ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15u) == 0;
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
    if (pool_k2_ok(H, W, Ho, Wo, k, stride, pad_t, pad_l, x, y, idx)) {
        hipLaunchKernelGGL(k_maxpool_fwd_k2, dim3(pool_blocks(total / 2)), dim3(256), 0, (hipStream_t)stream, x, y,
                           idx, total / 2, Ho, Wo / 2, W, BN_ACT_NONE, 0.f);
        BN_LAUNCH_CHECK();
        return 0;
    }
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
    if (pool_k2_ok(H, W, Ho, Wo, k, stride, pad_t, pad_l, dy, dx, idx)) {
        const size_t pairs = (size_t)planes * Ho * Wo / 2;
        hipLaunchKernelGGL(k_pool_spread_k2, dim3(pool_blocks(pairs)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx,
                           pairs, Ho, Wo / 2, W, (const float*)nullptr, BN_ACT_NONE, 0.f);
        BN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(pool_blocks(total)), dim3(256), 0, (hipStream_t)stream,
                       dy, idx, dx, total, H, W, Ho, Wo, k, stride, pad_t, pad_l);
    BN_LAUNCH_CHECK();
    return 0;
}

// 2x2 / stride-2 / unpadded pooling of an even map with the activation that follows it in one pass each way
// (BN_E_SHAPE where the two-windows-per-thread kernels do not apply: pool, then activate)
extern "C" int bn_maxpool2d_act_fwd(const float* x, float* y, int* idx, int planes, int H, int W, int act,
                                    float slope, bn_stream_t stream) {
    if (!x || !y || !idx || planes <= 0 || H <= 0 || W <= 0) return BN_E_BADARG;
    if ((H & 1) || (W & 1) || !pool_k2_ok(H, W, H / 2, W / 2, 2, 2, 0, 0, x, y, idx)) return BN_E_SHAPE;
    const size_t pairs = (size_t)planes * (H / 2) * (W / 2) / 2;
    hipLaunchKernelGGL(k_maxpool_fwd_k2, dim3(pool_blocks(pairs)), dim3(256), 0, (hipStream_t)stream, x, y, idx, pairs,
                       H / 2, W / 4, W, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}
// dx = spread( dy * act'(y) ) with y the saved output of bn_maxpool2d_act_fwd
extern "C" int bn_maxpool2d_act_bwd(const float* dy, const float* y, const int* idx, float* dx, int planes, int H,
                                    int W, int act, float slope, bn_stream_t stream) {
    if (!dy || !y || !dx || !idx || planes <= 0 || H <= 0 || W <= 0) return BN_E_BADARG;
    if ((H & 1) || (W & 1) || !pool_k2_ok(H, W, H / 2, W / 2, 2, 2, 0, 0, dy, dx, idx) || (((uintptr_t)y) & 15u))
        return BN_E_SHAPE;
    const size_t pairs = (size_t)planes * (H / 2) * (W / 2) / 2;
    hipLaunchKernelGGL(k_pool_spread_k2, dim3(pool_blocks(pairs)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx,
                       pairs, H / 2, W / 4, W, y, act, slope);
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

// the unpooling of a 2x2 / stride-2 pooling with THAT pooling's indices (every index inside its own window; the caller
// vouches for it): one pass, no memset.  BN_E_SHAPE when the maps do not qualify (call bn_maxunpool2d_fwd then)
extern "C" int bn_maxunpool2d_fwd_k2(const float* x, const int* idx, float* y, int planes, int Hi, int Wi,
                                     bn_stream_t stream) {
    if (!x || !y || !idx) return BN_E_BADARG;
    if (planes <= 0 || Hi <= 0 || Wi <= 0) return BN_E_BADARG;
    if (!pool_k2_ok(2 * Hi, 2 * Wi, Hi, Wi, 2, 2, 0, 0, x, y, idx)) return BN_E_SHAPE;
    const size_t pairs = (size_t)planes * Hi * Wi / 2;
    hipLaunchKernelGGL(k_pool_spread_k2, dim3(pool_blocks(pairs)), dim3(256), 0, (hipStream_t)stream, x, idx, y, pairs,
                       Hi, Wi / 2, 2 * Wi, (const float*)nullptr, BN_ACT_NONE, 0.f);
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
