// Geometries the specialised kernels were not written for, served BY them:
//
//  * 5x5 stride-2 layers whose small-side map is no power of two (the tiles of the MFMA families
//    need power-of-two rows and columns): both tensors are embedded, top-left, in zero-padded
//    copies of the next power-of-two size.  TF-"same" convolutions and cropped transposed
//    convolutions read zeros outside the frame, the padding supplies exactly those zeros, so every
//    element of the original region is unchanged (gather-down and gather-up: the result is cropped
//    back; weight gradient: padded positions add 0).  Costs one copy in and one copy out per call
//    and (padded area / area) of the arithmetic -- against direct loops that ran 50-100x slower
//    (the reference's integration-test shape 1x64x48: 42.5 ms per training step, 0.015 of peak).
//  * 5x5 stride-5 gather-down onto maps of a few pixels other than the benchmark's 2x2 (enc.conv4 /
//    dec.convT0 data gradient on other frame sizes): the windows do not overlap, every window is a
//    row of an im2col matrix and the layer a dense GEMM on the matrix cores.
#include "bn_common.h"
#include "bn_launch.h"

#define PD_THREADS 256

// dst (planes, Hp, Wp) <- src (planes, H, W) at row / column offset (oh, ow), zeros elsewhere.
// (The offset turns a layer whose first tap sits pt = 2 pixels outside the frame into the pt = 1
// form the specialised kernels are written for: big'[k] = big[k - (pt - 1)].)
__global__ __launch_bounds__(PD_THREADS) void k_pad2d(const float* __restrict__ src,
                                                       float* __restrict__ dst, size_t planes, int H,
                                                       int W, int Hp, int Wp, int oh, int ow) {
    const size_t total = planes * Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int w = (int)(i % Wp) - ow;
        const size_t t = i / Wp;
        const int h = (int)(t % Hp) - oh;
        const size_t pl = t / Hp;
        dst[i] = (h >= 0 && h < H && w >= 0 && w < W) ? src[(pl * H + h) * W + w] : 0.f;
    }
}

// dst (planes, H, W) <- src (planes, Hp, Wp) from offset (oh, ow), times act'(dact_src) when given
__global__ __launch_bounds__(PD_THREADS) void k_crop2d(const float* __restrict__ src,
                                                        float* __restrict__ dst, size_t planes, int H,
                                                        int W, int Hp, int Wp, int oh, int ow,
                                                        const float* __restrict__ dact_src, int dact,
                                                        float slope) {
    const size_t total = planes * H * W;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int w = (int)(i % W);
        const size_t t = i / W;
        const int h = (int)(t % H);
        const size_t pl = t / H;
        float v = src[(pl * Hp + h + oh) * Wp + w + ow];
        if (dact_src) v *= bn_act_grad_from_output(dact_src[i], dact, slope);
        dst[i] = v;
    }
}

static int pd_blocks(size_t n) {
    const size_t b = (n + PD_THREADS - 1) / PD_THREADS;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

int bn_launch_pad2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                    int oh, int ow, hipStream_t st) {
    hipLaunchKernelGGL(k_pad2d, dim3(pd_blocks(planes * Hp * Wp)), dim3(PD_THREADS), 0, st, src, dst,
                       planes, H, W, Hp, Wp, oh, ow);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_crop2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                     int oh, int ow, const float* dact_src, int dact, float slope, hipStream_t st) {
    hipLaunchKernelGGL(k_crop2d, dim3(pd_blocks(planes * H * W)), dim3(PD_THREADS), 0, st, src, dst,
                       planes, H, W, Hp, Wp, oh, ow, dact_src, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// stride-5 gather-down onto a map of a few pixels (the windows do not overlap):
//   out[n, m, p, q] = act(b[m] + sum_{c,r,s} big[n, c, 5p + r - pt, 5q + s - pl] W[m, c, r, s]) * act'(dact_src)
// = (N P Q) x (C 25) im2col matrix times W^T, then the rows go back to NCHW.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PD_THREADS) void k_im2col_s5(const float* __restrict__ big,
                                                           float* __restrict__ col, BnGeom g) {
    const int PQ = g.Hs * g.Ws;
    const size_t total = (size_t)g.N * PQ * g.Cb * 25;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int tap = (int)(i % 25);
        size_t t = i / 25;
        const int c = (int)(t % g.Cb);
        t /= g.Cb;
        const int pq = (int)(t % PQ);
        const size_t n = t / PQ;
        const int h = 5 * (pq / g.Ws) + tap / 5 - g.pt, w = 5 * (pq % g.Ws) + tap % 5 - g.pl;
        col[i] = (h >= 0 && h < g.Hb && w >= 0 && w < g.Wb)
            ? big[((n * g.Cb + c) * g.Hb + h) * g.Wb + w] : 0.f;
    }
}

// tmp[(n, pq)][m] -> out[n][m][pq] with the activation and the derivative mask
__global__ __launch_bounds__(PD_THREADS) void k_s5_rows_to_nchw(const float* __restrict__ tmp,
                                                                 float* __restrict__ out,
                                                                 const float* __restrict__ dact_src,
                                                                 int N, int M, int PQ, int act, int dact,
                                                                 float slope) {
    const size_t total = (size_t)N * M * PQ;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int pq = (int)(i % PQ);
        const size_t t = i / PQ;
        const int m = (int)(t % M);
        const size_t n = t / M;
        float v = bn_apply_act(tmp[(n * PQ + pq) * M + m], act, slope);
        if (dact_src) v *= bn_act_grad_from_output(dact_src[i], dact, slope);
        out[i] = v;
    }
}

bool bn_s5_down_small_ok(const BnGeom& g) {
    return g.R == 5 && g.S == 5 && g.stride == 5 && g.Hs * g.Ws <= 16 && g.pt <= 4 && g.pl <= 4 &&
           g.Cs >= 16 && g.Cb >= 16 && (size_t)g.N * g.Hs * g.Ws * g.Cb * 25 * 4 < 0x7fffffffull;
}

static size_t s5_col_bytes(const BnGeom& g) {
    return ((size_t)g.N * g.Hs * g.Ws * g.Cb * 25 * 4 + 255) & ~(size_t)255;
}
static size_t s5_tmp_bytes(const BnGeom& g) {
    return ((size_t)g.N * g.Hs * g.Ws * g.Cs * 4 + 255) & ~(size_t)255;
}

size_t bn_s5_down_small_ws_bytes(const BnGeom& g) {
    return s5_col_bytes(g) + s5_tmp_bytes(g) + bn_gemm_ws_bytes(g.N * g.Hs * g.Ws, g.Cs, g.Cb * 25);
}

int bn_launch_s5_down_small(const float* big, const float* w, const float* bias, float* out,
                            const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    if (!ws || ws_bytes < bn_s5_down_small_ws_bytes(g)) return BN_E_WORKSPACE;
    const int PQ = g.Hs * g.Ws, rows = g.N * PQ, K = g.Cb * 25;
    float* col = (float*)ws;
    float* tmp = (float*)((char*)ws + s5_col_bytes(g));
    const size_t used = s5_col_bytes(g) + s5_tmp_bytes(g);
    hipLaunchKernelGGL(k_im2col_s5, dim3(pd_blocks((size_t)rows * K)), dim3(PD_THREADS), 0, st, big, col, g);
    BN_LAUNCH_CHECK();
    GemmArgs a;                                   // tmp[i, m] = b[m] + sum_k col[i, k] W[m, k]
    a.A = col; a.sai = K; a.sak = 1;
    a.B = w; a.sbk = 1; a.sbj = K;
    a.C = tmp; a.sci = g.Cs; a.scj = 1;
    a.M = rows; a.N = g.Cs; a.K = K;
    a.bias_j = bias; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = slope; a.accumulate = 0;
    const int rc = bn_launch_gemm(a, st, (char*)ws + used, ws_bytes - used);
    if (rc) return rc;
    hipLaunchKernelGGL(k_s5_rows_to_nchw, dim3(pd_blocks((size_t)rows * g.Cs)), dim3(PD_THREADS), 0, st, tmp,
                       out, dact_src, g.N, g.Cs, PQ, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}
