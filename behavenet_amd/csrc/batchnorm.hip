// BatchNorm2d (+ fused LeakyReLU) for architectures with ae_batch_norm = 1
// (reference aes.py:90-97,113-114,332-341).  Statistics are per chunk, over (N, H, W) per channel,
// exactly like nn.BatchNorm2d in train mode: biased variance for the normalisation, unbiased
// variance for the running estimate, momentum None = cumulative average.
//
// All kernels are HBM streams with a per-channel reduction: (channel, frame-slice) partial sums
// by wavefront shuffles + LDS, combined in fixed order (deterministic).  The variance is
// computed in a second pass around the mean (as torch does), not as E[x^2] - mean^2.
#include "bn_common.h"
#include "bn_launch.h"

#define BNK_THREADS 256

__device__ __forceinline__ float bnk_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[c][sp] = sum over frames [n_beg, n_end) and pixels of (x - shift[c])^p   (p = 1 or 2)
template <int POW>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_moment_part(
    const float* __restrict__ x, const float* __restrict__ shift, float* __restrict__ part, int N,
    int C, int HW, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int n_beg = (int)((long)sp * N / S), n_end = (int)((long)(sp + 1) * N / S);
    const float sh = shift ? shift[c] : 0.f;
    float acc = 0.f;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)x) & 15u) == 0);
    for (int n = n_beg; n < n_end; ++n) {
        const float* xp = x + ((size_t)n * C + c) * HW;
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xp);
            for (int i = threadIdx.x; i < (HW >> 2); i += BNK_THREADS) {
                float4 v = x4[i];
                v.x -= sh; v.y -= sh; v.z -= sh; v.w -= sh;
                if (POW == 2) { v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w; }
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += BNK_THREADS) {
                float v = xp[i] - sh;
                acc += POW == 2 ? v * v : v;
            }
        }
    }
    const float s = bnk_block_sum(acc, red);
    if (threadIdx.x == 0) part[(size_t)c * S + sp] = s;
}

// out[c] = scale * sum_s part[c][s]
__global__ void k_bn_combine(const float* __restrict__ part, float* __restrict__ out, int C, int S,
                             float scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float v = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) v += part[(size_t)c * S + s];
    out[c] = v * scale;
}

// invstd = 1/sqrt(var + eps); running stats (momentum < 0: the caller passes the cumulative
// average factor 1/num_batches_tracked instead, as nn.BatchNorm2d(momentum=None) does)
__global__ void k_bn_finalize(const float* __restrict__ mean, const float* __restrict__ var,
                              float* __restrict__ invstd, float* __restrict__ running_mean,
                              float* __restrict__ running_var, int C, float eps, float momentum,
                              float unbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = mean[c], v = var[c];
    invstd[c] = 1.0f / sqrtf(v + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * v * unbias;
}

// y = act( (x - mean) * invstd * gamma + beta )
__global__ __launch_bounds__(BNK_THREADS) void k_bn_act_fwd(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int C,
    int HW, int act, float slope) {
    const int nc = blockIdx.x;           // n * C + c
    const int c = nc % C;
    const float sc = invstd[c] * (gamma ? gamma[c] : 1.f);
    const float sh = (beta ? beta[c] : 0.f) - mean[c] * sc;
    const float* xp = x + (size_t)nc * HW;
    float* yp = y + (size_t)nc * HW;
    if ((HW & 3) == 0 && (((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0)) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        float4* y4 = reinterpret_cast<float4*>(yp);
        for (int i = threadIdx.x; i < (HW >> 2); i += BNK_THREADS) {
            const float4 v = x4[i];
            float4 o;
            o.x = bn_apply_act(fmaf(v.x, sc, sh), act, slope);
            o.y = bn_apply_act(fmaf(v.y, sc, sh), act, slope);
            o.z = bn_apply_act(fmaf(v.z, sc, sh), act, slope);
            o.w = bn_apply_act(fmaf(v.w, sc, sh), act, slope);
            y4[i] = o;
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += BNK_THREADS)
            yp[i] = bn_apply_act(fmaf(xp[i], sc, sh), act, slope);
    }
}

// backward reductions: part0[c][sp] = sum dz, part1[c][sp] = sum dz * xhat,
// dz = dy * act'(y), xhat = (x - mean) * invstd
__global__ __launch_bounds__(BNK_THREADS) void k_bn_bwd_part(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ part0,
    float* __restrict__ part1, int N, int C, int HW, int S, int act, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int n_beg = (int)((long)sp * N / S), n_end = (int)((long)(sp + 1) * N / S);
    const float m = mean[c], is = invstd[c];
    float a0 = 0.f, a1 = 0.f;
    for (int n = n_beg; n < n_end; ++n) {
        const size_t base = ((size_t)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += BNK_THREADS) {
            const float dz = dy[base + i] * bn_act_grad_from_output(y[base + i], act, slope);
            a0 += dz;
            a1 += dz * ((x[base + i] - m) * is);
        }
    }
    const float s0 = bnk_block_sum(a0, red);
    const float s1 = bnk_block_sum(a1, red);
    if (threadIdx.x == 0) {
        part0[(size_t)c * S + sp] = s0;
        part1[(size_t)c * S + sp] = s1;
    }
}

// dx = gamma * invstd * (dz - dbeta/n - xhat * dgamma/n);  dgamma/dbeta (+)= the sums
__global__ __launch_bounds__(BNK_THREADS) void k_bn_bwd_apply(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ sum_dz,
    const float* __restrict__ sum_dzx, float* __restrict__ dx, int C, int HW, float inv_n, int act,
    float slope) {
    const int nc = blockIdx.x;
    const int c = nc % C;
    const float m = mean[c], is = invstd[c];
    const float g = (gamma ? gamma[c] : 1.f) * is;
    const float k0 = sum_dz[c] * inv_n, k1 = sum_dzx[c] * inv_n;
    const size_t base = (size_t)nc * HW;
    for (int i = threadIdx.x; i < HW; i += BNK_THREADS) {
        const float dz = dy[base + i] * bn_act_grad_from_output(y[base + i], act, slope);
        const float xh = (x[base + i] - m) * is;
        dx[base + i] = g * (dz - k0 - xh * k1);
    }
}

__global__ void k_bn_param_grads(const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx,
                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                 int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + sum_dzx[c] : sum_dzx[c];
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + sum_dz[c] : sum_dz[c];
}

// ---------------------------------------------------------------------------------------------
static int bn_splits(int N, int C) {
    int s = 1024 / C;
    if (s > 64) s = 64;
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

size_t bn_batchnorm_ws_bytes_impl(int N, int C) {
    // two partial arrays [C][S] + two combined vectors [C]
    return ((size_t)2 * C * bn_splits(N, C) + 2 * C) * sizeof(float);
}

int bn_launch_bn_stats(const float* x, float* mean, float* var, int N, int C, int HW, void* ws,
                       hipStream_t st) {
    const int S = bn_splits(N, C);
    float* part = (float*)ws;
    const float inv_n = 1.0f / ((float)N * (float)HW);
    hipLaunchKernelGGL(k_bn_moment_part<1>, dim3(C, S), dim3(BNK_THREADS), 0, st, x,
                       (const float*)nullptr, part, N, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part, mean, C, S, inv_n);
    hipLaunchKernelGGL(k_bn_moment_part<2>, dim3(C, S), dim3(BNK_THREADS), 0, st, x,
                       (const float*)mean, part, N, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part, var, C, S, inv_n);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_finalize(const float* mean, const float* var, float* invstd, float* running_mean,
                          float* running_var, int C, float eps, float momentum, float unbias,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, st, mean, var, invstd,
                       running_mean, running_var, C, eps, momentum, unbias);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_act_fwd(const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* y, int N, int C, int HW,
                         int act, float slope, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_act_fwd, dim3(N * C), dim3(BNK_THREADS), 0, st, x, mean, invstd, gamma,
                       beta, y, C, HW, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_act_bwd(const float* x, const float* y, const float* dy, const float* mean,
                         const float* invstd, const float* gamma, float* dx, float* dgamma,
                         float* dbeta, int accumulate, int batch_stats, int N, int C, int HW,
                         int act, float slope, void* ws, hipStream_t st) {
    const int S = bn_splits(N, C);
    float* part0 = (float*)ws;
    float* part1 = part0 + (size_t)C * S;
    float* sum0 = part1 + (size_t)C * S;
    float* sum1 = sum0 + C;
    hipLaunchKernelGGL(k_bn_bwd_part, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,
                       part0, part1, N, C, HW, S, act, slope);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part0, sum0, C, S, 1.0f);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part1, sum1, C, S, 1.0f);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(N * C), dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,
                       gamma, sum0, sum1, dx, C, HW,
                       batch_stats ? 1.0f / ((float)N * (float)HW) : 0.0f, act, slope);
    hipLaunchKernelGGL(k_bn_param_grads, dim3((C + 63) / 64), dim3(64), 0, st, sum0, sum1, dgamma,
                       dbeta, C, accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Split forms for statistics synchronised over ranks (frame-sharded data parallelism): the
// caller all-reduces the per-channel SUMS between the passes.
// ---------------------------------------------------------------------------------------------
// sums[c] = sum over (n, pixels) of x (center == nullptr) or of (x - center[c])^2; not divided
int bn_launch_bn_moment(const float* x, const float* center, float* sums, int N, int C, int HW,
                        void* ws, hipStream_t st) {
    const int S = bn_splits(N, C);
    float* part = (float*)ws;
    if (center)
        hipLaunchKernelGGL(k_bn_moment_part<2>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, center,
                           part, N, C, HW, S);
    else
        hipLaunchKernelGGL(k_bn_moment_part<1>, dim3(C, S), dim3(BNK_THREADS), 0, st, x,
                           (const float*)nullptr, part, N, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part, sums, C, S, 1.0f);
    BN_LAUNCH_CHECK();
    return 0;
}

// sum_dz[c] = sum dy act'(y), sum_dzx[c] = sum dy act'(y) xhat  (this rank's frames)
int bn_launch_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                            const float* invstd, float* sum_dz, float* sum_dzx, int N, int C,
                            int HW, int act, float slope, void* ws, hipStream_t st) {
    const int S = bn_splits(N, C);
    float* part0 = (float*)ws;
    float* part1 = part0 + (size_t)C * S;
    hipLaunchKernelGGL(k_bn_bwd_part, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,
                       part0, part1, N, C, HW, S, act, slope);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part0, sum_dz, C, S, 1.0f);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part1, sum_dzx, C, S,
                       1.0f);
    BN_LAUNCH_CHECK();
    return 0;
}

// dx from the (global) sums; inv_count = 1 / (frames * pixels the statistics were taken over)
int bn_launch_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                           const float* invstd, const float* gamma, const float* sum_dz,
                           const float* sum_dzx, float* dx, int N, int C, int HW, float inv_count,
                           int act, float slope, hipStream_t st) {
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(N * C), dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,
                       gamma, sum_dz, sum_dzx, dx, C, HW, inv_count, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}
