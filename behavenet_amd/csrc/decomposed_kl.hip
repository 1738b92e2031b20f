// Decomposed KL of the beta-TC-VAE / PS-VAE (reference behavenet/fitting/losses.py:284-372):
// with L[j,i,l] = log N(z[j,l]; mu[i,l], exp(logvar[i,l]))   (N x N x D, never materialised)
//   joint[j,i] = sum_l L[j,i,l]            log_qz[j]  = logsumexp_i joint[j,i]
//   cond[j]    = joint[j,j]                lqp[j]     = sum_l logsumexp_i L[j,i,l]
//   lpz[j]     = sum_l -0.5 (z[j,l]^2 + ln 2pi)
//   MI = mean_j (cond - log_qz)    TC = mean_j (log_qz - lqp)    DWKL = mean_j (lqp - lpz)
// The batch is one 200-frame chunk and D <= 32, so the whole thing is a few hundred thousand
// exp() evaluations: one workgroup per sample j (threads over i) for the forward and the z
// gradient, one per i (threads over j) for the mu / logvar gradients; the softmax weights of the
// backward pass are recomputed from the saved log_qz[j] and lse[j,l].  logsumexp is evaluated as
// torch does (max, then sum of exp(x - max)) in a fixed reduction order.
#include "bn_common.h"
#include "bn_launch.h"

#define DK_THREADS 256
#define DK_MAXD 32
#define DK_LN2PI 1.8378770664093453f

__device__ __forceinline__ float dk_block_reduce(float v, bool is_max, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(v, off, 64);
        v = is_max ? fmaxf(v, o) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < DK_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
}

// per-sample terms: terms[j] = cond - log_qz, terms[N + j] = log_qz - lqp, terms[2N + j] = lqp - lpz
__global__ __launch_bounds__(DK_THREADS) void k_dkl_fwd(
    const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ lv,
    float* __restrict__ terms, float* __restrict__ log_qz, float* __restrict__ lse, int N, int D) {
    __shared__ float red[DK_THREADS / 64];
    __shared__ float zj[DK_MAXD];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (tid < D) zj[tid] = z[(size_t)j * D + tid];
    __syncthreads();

    // pass 1: maxima over i of joint[j,i] and of every L[j,i,l]
    float mx[DK_MAXD + 1];
#pragma unroll
    for (int l = 0; l <= DK_MAXD; ++l) mx[l] = -INFINITY;
    for (int i = tid; i < N; i += DK_THREADS) {
        float joint = 0.f;
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l) {
            if (l < D) {
                const float m = mu[(size_t)i * D + l], v = lv[(size_t)i * D + l];
                const float d = zj[l] - m;
                const float L = -0.5f * (expf(-v) * d * d + v + DK_LN2PI);
                mx[l] = fmaxf(mx[l], L);
                joint += L;
            }
        }
        mx[DK_MAXD] = fmaxf(mx[DK_MAXD], joint);
    }
#pragma unroll
    for (int l = 0; l <= DK_MAXD; ++l)
        if (l < D || l == DK_MAXD) mx[l] = dk_block_reduce(mx[l], true, red);

    // pass 2: sums of exp(x - max)
    float sm[DK_MAXD + 1];
#pragma unroll
    for (int l = 0; l <= DK_MAXD; ++l) sm[l] = 0.f;
    float cond = 0.f;
    for (int i = tid; i < N; i += DK_THREADS) {
        float joint = 0.f;
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l) {
            if (l < D) {
                const float m = mu[(size_t)i * D + l], v = lv[(size_t)i * D + l];
                const float d = zj[l] - m;
                const float L = -0.5f * (expf(-v) * d * d + v + DK_LN2PI);
                sm[l] += expf(L - mx[l]);
                joint += L;
            }
        }
        sm[DK_MAXD] += expf(joint - mx[DK_MAXD]);
        if (i == j) cond = joint;
    }
    float lqp = 0.f;
#pragma unroll
    for (int l = 0; l < DK_MAXD; ++l) {
        if (l < D) {
            const float s = dk_block_reduce(sm[l], false, red);
            const float e = logf(s) + mx[l];
            lqp += e;
            if (tid == 0) lse[(size_t)j * D + l] = e;
        }
    }
    const float sj = dk_block_reduce(sm[DK_MAXD], false, red);
    const float lq = logf(sj) + mx[DK_MAXD];
    const float cj = dk_block_reduce(cond, false, red);
    if (tid == 0) {
        float lpz = 0.f;
        for (int l = 0; l < D; ++l) lpz += -0.5f * (zj[l] * zj[l] + DK_LN2PI);
        log_qz[j] = lq;
        terms[j] = cj - lq;
        terms[N + j] = lq - lqp;
        terms[2 * N + j] = lqp - lpz;
    }
}

// out[t] = mean_j terms[t*N + j], t = 0..2   (fixed order)
__global__ __launch_bounds__(DK_THREADS) void k_dkl_means(const float* __restrict__ terms,
                                                          float* __restrict__ out, int N) {
    __shared__ float red[DK_THREADS / 64];
    const int t = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += DK_THREADS) s += terms[(size_t)t * N + j];
    s = dk_block_reduce(s, false, red);
    if (threadIdx.x == 0) out[t] = s / (float)N;
}

// G[j,i,l] = a*[i==j] + b*P[j,i] + c*S[j,i,l],   a = g_mi/N, b = (g_tc-g_mi)/N, c = (g_dw-g_tc)/N
//   P[j,i] = exp(joint[j,i] - log_qz[j]),  S[j,i,l] = exp(L[j,i,l] - lse[j,l])
// dz[j,l] = sum_i G * dL/dz + (g_dw/N) z[j,l],   dL/dz = -exp(-lv[i,l]) (z[j,l]-mu[i,l])
__global__ __launch_bounds__(DK_THREADS) void k_dkl_bwd_z(
    const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ lv,
    const float* __restrict__ log_qz, const float* __restrict__ lse, const float* __restrict__ g3,
    float* __restrict__ dz, int N, int D) {
    __shared__ float red[DK_THREADS / 64];
    __shared__ float zj[DK_MAXD], ej[DK_MAXD];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (tid < D) {
        zj[tid] = z[(size_t)j * D + tid];
        ej[tid] = lse[(size_t)j * D + tid];
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)N;
    const float a = g3[0] * inv_n, b = (g3[1] - g3[0]) * inv_n, c = (g3[2] - g3[1]) * inv_n;
    const float lq = log_qz[j];
    float acc[DK_MAXD];
#pragma unroll
    for (int l = 0; l < DK_MAXD; ++l) acc[l] = 0.f;
    for (int i = tid; i < N; i += DK_THREADS) {
        float L[DK_MAXD], wd[DK_MAXD];
        float joint = 0.f;
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l) {
            if (l < D) {
                const float m = mu[(size_t)i * D + l], v = lv[(size_t)i * D + l];
                const float d = zj[l] - m, w = expf(-v);
                L[l] = -0.5f * (w * d * d + v + DK_LN2PI);
                wd[l] = w * d;
                joint += L[l];
            }
        }
        const float base = (i == j ? a : 0.f) + b * expf(joint - lq);
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l)
            if (l < D) acc[l] -= (base + c * expf(L[l] - ej[l])) * wd[l];
    }
#pragma unroll
    for (int l = 0; l < DK_MAXD; ++l) {
        if (l < D) {
            const float s = dk_block_reduce(acc[l], false, red);
            if (tid == 0) dz[(size_t)j * D + l] = s + g3[2] * inv_n * zj[l];
        }
    }
}

// dmu[i,l] = sum_j G * w d,   dlogvar[i,l] = sum_j G * 0.5 (w d^2 - 1)
__global__ __launch_bounds__(DK_THREADS) void k_dkl_bwd_q(
    const float* __restrict__ z, const float* __restrict__ mu, const float* __restrict__ lv,
    const float* __restrict__ log_qz, const float* __restrict__ lse, const float* __restrict__ g3,
    float* __restrict__ dmu, float* __restrict__ dlv, int N, int D) {
    __shared__ float red[DK_THREADS / 64];
    __shared__ float mi[DK_MAXD], vi[DK_MAXD];
    const int i = blockIdx.x, tid = threadIdx.x;
    if (tid < D) {
        mi[tid] = mu[(size_t)i * D + tid];
        vi[tid] = lv[(size_t)i * D + tid];
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)N;
    const float a = g3[0] * inv_n, b = (g3[1] - g3[0]) * inv_n, c = (g3[2] - g3[1]) * inv_n;
    float am[DK_MAXD], av[DK_MAXD];
#pragma unroll
    for (int l = 0; l < DK_MAXD; ++l) am[l] = av[l] = 0.f;
    for (int j = tid; j < N; j += DK_THREADS) {
        float L[DK_MAXD], wd[DK_MAXD], wdd[DK_MAXD];
        float joint = 0.f;
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l) {
            if (l < D) {
                const float d = z[(size_t)j * D + l] - mi[l], w = expf(-vi[l]);
                L[l] = -0.5f * (w * d * d + vi[l] + DK_LN2PI);
                wd[l] = w * d;
                wdd[l] = 0.5f * (w * d * d - 1.0f);
                joint += L[l];
            }
        }
        const float base = (i == j ? a : 0.f) + b * expf(joint - log_qz[j]);
#pragma unroll
        for (int l = 0; l < DK_MAXD; ++l) {
            if (l < D) {
                const float G = base + c * expf(L[l] - lse[(size_t)j * D + l]);
                am[l] += G * wd[l];
                av[l] += G * wdd[l];
            }
        }
    }
#pragma unroll
    for (int l = 0; l < DK_MAXD; ++l) {
        if (l < D) {
            const float s0 = dk_block_reduce(am[l], false, red);
            const float s1 = dk_block_reduce(av[l], false, red);
            if (tid == 0) {
                dmu[(size_t)i * D + l] = s0;
                dlv[(size_t)i * D + l] = s1;
            }
        }
    }
}

int bn_launch_dkl_fwd(const float* z, const float* mu, const float* lv, float* out3,
                      float* log_qz, float* lse, float* terms, int N, int D, hipStream_t st) {
    if (D > DK_MAXD) return BN_E_SHAPE;
    hipLaunchKernelGGL(k_dkl_fwd, dim3(N), dim3(DK_THREADS), 0, st, z, mu, lv, terms, log_qz, lse,
                       N, D);
    hipLaunchKernelGGL(k_dkl_means, dim3(3), dim3(DK_THREADS), 0, st, terms, out3, N);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_dkl_bwd(const float* z, const float* mu, const float* lv, const float* log_qz,
                      const float* lse, const float* g3, float* dz, float* dmu, float* dlv, int N,
                      int D, hipStream_t st) {
    if (D > DK_MAXD) return BN_E_SHAPE;
    hipLaunchKernelGGL(k_dkl_bwd_z, dim3(N), dim3(DK_THREADS), 0, st, z, mu, lv, log_qz, lse, g3,
                       dz, N, D);
    hipLaunchKernelGGL(k_dkl_bwd_q, dim3(N), dim3(DK_THREADS), 0, st, z, mu, lv, log_qz, lse, g3,
                       dmu, dlv, N, D);
    BN_LAUNCH_CHECK();
    return 0;
}
