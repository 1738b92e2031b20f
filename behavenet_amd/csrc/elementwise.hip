// HBM-bound elementwise / reduction kernels: activation backward, pixel losses, the variational
// tail, Adam(amsgrad) and the uint8 -> unit-float frame conversion.
// All streams are float4-vectorised with a scalar tail; grids are capped and grid-strided.
#include "bn_common.h"
#include "bn_launch.h"

#define EW_THREADS 256
#define EW_MAX_BLOCKS 2048   // 256 CUs x 8

static inline int ew_blocks(size_t nvec) {
    size_t b = (nvec + EW_THREADS - 1) / EW_THREADS;
    if (b < 1) b = 1;
    if (b > EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    return (int)b;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------- activation backward
__global__ __launch_bounds__(EW_THREADS) void k_act_bwd(
    const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dpre, size_t n,
    int vec, int act, float slope) {
    const size_t tid = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    if (vec) {
        const size_t n4 = n >> 2;
        const float4* dy4 = reinterpret_cast<const float4*>(dy);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        float4* o4 = reinterpret_cast<float4*>(dpre);
        for (size_t i = tid; i < n4; i += nthreads) {
            const float4 a = dy4[i], b = y4[i];
            float4 o;
            o.x = a.x * bn_act_grad_from_output(b.x, act, slope);
            o.y = a.y * bn_act_grad_from_output(b.y, act, slope);
            o.z = a.z * bn_act_grad_from_output(b.z, act, slope);
            o.w = a.w * bn_act_grad_from_output(b.w, act, slope);
            o4[i] = o;
        }
        for (size_t i = (n4 << 2) + tid; i < n; i += nthreads)
            dpre[i] = dy[i] * bn_act_grad_from_output(y[i], act, slope);
    } else {
        for (size_t i = tid; i < n; i += nthreads)
            dpre[i] = dy[i] * bn_act_grad_from_output(y[i], act, slope);
    }
}

// ---------------------------------------------------------------- squared-error frame sums
// one workgroup per frame; float4 loads; wave shuffle tree + LDS combine (fixed order)
__global__ __launch_bounds__(EW_THREADS) void k_sqerr_frame_sums(
    const float* __restrict__ pred, const float* __restrict__ target,
    const float* __restrict__ mask, float* __restrict__ frame_sums, size_t D, int vec) {
    __shared__ float red[EW_THREADS / BN_WAVE];
    const size_t base = (size_t)blockIdx.x * D;
    const float* p = pred + base;
    const float* t = target + base;
    const float* m = mask ? mask + base : nullptr;
    float acc = 0.f;
    if (vec) {
        const size_t D4 = D >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        const float4* t4 = reinterpret_cast<const float4*>(t);
        const float4* m4 = reinterpret_cast<const float4*>(m);
        for (size_t i = threadIdx.x; i < D4; i += EW_THREADS) {
            const float4 a = p4[i], b = t4[i];
            float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
            dx *= dx; dy *= dy; dz *= dz; dw *= dw;
            if (m) { const float4 k = m4[i]; dx *= k.x; dy *= k.y; dz *= k.z; dw *= k.w; }
            acc += (dx + dy) + (dz + dw);
        }
        for (size_t i = (D4 << 2) + threadIdx.x; i < D; i += EW_THREADS) {
            float d = p[i] - t[i]; d *= d; if (m) d *= m[i]; acc += d;
        }
    } else {
        for (size_t i = threadIdx.x; i < D; i += EW_THREADS) {
            float d = p[i] - t[i]; d *= d; if (m) d *= m[i]; acc += d;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) frame_sums[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(EW_THREADS) void k_sqerr_bwd(
    const float* __restrict__ pred, const float* __restrict__ target,
    const float* __restrict__ mask, float* __restrict__ dpred, size_t n, float scale,
    const float* __restrict__ gscale, int vec) {
    const float sc = 2.f * scale * (gscale ? gscale[0] : 1.f);
    const size_t tid = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    if (vec) {
        const size_t n4 = n >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(pred);
        const float4* t4 = reinterpret_cast<const float4*>(target);
        const float4* m4 = reinterpret_cast<const float4*>(mask);
        float4* o4 = reinterpret_cast<float4*>(dpred);
        for (size_t i = tid; i < n4; i += nthreads) {
            const float4 a = p4[i], b = t4[i];
            float4 o;
            o.x = sc * (a.x - b.x); o.y = sc * (a.y - b.y);
            o.z = sc * (a.z - b.z); o.w = sc * (a.w - b.w);
            if (mask) { const float4 k = m4[i]; o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w; }
            o4[i] = o;
        }
        for (size_t i = (n4 << 2) + tid; i < n; i += nthreads) {
            float o = sc * (pred[i] - target[i]); if (mask) o *= mask[i]; dpred[i] = o;
        }
    } else {
        for (size_t i = tid; i < n; i += nthreads) {
            float o = sc * (pred[i] - target[i]); if (mask) o *= mask[i]; dpred[i] = o;
        }
    }
}

// out[0] = scale * sum(in[0..n)); single workgroup, fixed order
__global__ __launch_bounds__(EW_THREADS) void k_reduce_sum(
    const float* __restrict__ in, float* __restrict__ out, size_t n, float scale) {
    __shared__ float red[EW_THREADS / BN_WAVE];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += EW_THREADS) acc += in[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

// ---------------------------------------------------------------- variational tail
__global__ __launch_bounds__(EW_THREADS) void k_reparam_fwd(
    const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
    float* __restrict__ z, size_t n) {
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += nthreads)
        z[i] = fmaf(eps[i], expf(logvar[i]), mu[i]);   // std = exp(logvar): vaes.py:33
}

// one wave per row; D <= a few hundred
__global__ __launch_bounds__(EW_THREADS) void k_kl_rows(
    const float* __restrict__ mu, const float* __restrict__ logvar, float* __restrict__ kl_rows,
    int N, int D) {
    const int row = blockIdx.x * (EW_THREADS / BN_WAVE) + (threadIdx.x >> 6);
    if (row >= N) return;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int d = lane; d < D; d += BN_WAVE) {
        const float lv = logvar[(size_t)row * D + d], m = mu[(size_t)row * D + d];
        acc += expf(lv) - lv + m * m - 1.f;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) kl_rows[row] = 0.5f * acc;
}

__global__ __launch_bounds__(EW_THREADS) void k_reparam_bwd(
    const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ mu,
    float* __restrict__ dlogvar, size_t n) {
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += nthreads)
        dlogvar[i] = dz[i] * (z[i] - mu[i]);
}

__global__ __launch_bounds__(EW_THREADS) void k_kl_bwd(
    const float* __restrict__ mu, const float* __restrict__ logvar, float* __restrict__ dmu,
    float* __restrict__ dlogvar, size_t n, float scale, const float* __restrict__ gscale) {
    const float s = scale * (gscale ? gscale[0] : 1.f);
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += nthreads) {
        dmu[i] = s * mu[i];
        dlogvar[i] = s * 0.5f * (expf(logvar[i]) - 1.f);
    }
}

// ---------------------------------------------------------------- Adam (amsgrad)
// Follows torch.optim.Adam single-tensor update order (amsgrad=True, maximize=False):
//   g += wd*p; m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; vmax = max(vmax, v);
//   denom = sqrt(vmax)/sqrt(bc2) + eps; p -= (lr/bc1) * m/denom
typedef unsigned int adam_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(EW_THREADS) void k_adam_amsgrad(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, float* __restrict__ vmax, size_t n, float step_size, float b1,
    float b2, float bc2_sqrt, float eps, float wd, int vec) {
    const size_t tid = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
#define ADAM_ONE(P, G, M, V, X)                                  \
    {                                                            \
        float gg = (G);                                          \
        if (wd != 0.f) gg = fmaf(wd, (P), gg);                   \
        (M) = (M) + omb1 * (gg - (M));                           \
        (V) = fmaf(omb2 * gg, gg, (V) * b2);                     \
        (X) = fmaxf((X), (V));                                   \
        const float denom = sqrtf(X) / bc2_sqrt + eps;           \
        (P) = (P) - step_size * ((M) / denom);                   \
    }
#define ADAM_ST(VAL, PTR)                                                                       \
    __builtin_amdgcn_raw_buffer_store_b128(                                                     \
        __builtin_bit_cast(adam_u4, VAL),                                                       \
        __builtin_amdgcn_make_buffer_rsrc((void*)(PTR), 0, (int)(n * 4), 0x00020000), o, 0, 16);
    if (vec) {
        const size_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        float4* x4 = reinterpret_cast<float4*>(vmax);
        for (size_t i = tid; i < n4; i += nthreads) {
            float4 P = p4[i], M = m4[i], V = v4[i], X = x4[i];
            const float4 G = g4[i];
            ADAM_ONE(P.x, G.x, M.x, V.x, X.x)
            ADAM_ONE(P.y, G.y, M.y, V.y, X.y)
            ADAM_ONE(P.z, G.z, M.z, V.z, X.z)
            ADAM_ONE(P.w, G.w, M.w, V.w, X.w)
            if (vec == 2) {
                // write-through (sc1) 16-byte stores: the updated arenas are not read again before
                // the next step, so their lines need not sit dirty in the L2s while the first
                // kernels of that step stream their own output
                const int o = (int)(i << 4);
                ADAM_ST(P, p) ADAM_ST(M, m) ADAM_ST(V, v) ADAM_ST(X, vmax)
            } else {
                p4[i] = P; m4[i] = M; v4[i] = V; x4[i] = X;
            }
        }
        for (size_t i = (n4 << 2) + tid; i < n; i += nthreads)
            ADAM_ONE(p[i], g[i], m[i], v[i], vmax[i])
    } else {
        for (size_t i = tid; i < n; i += nthreads) ADAM_ONE(p[i], g[i], m[i], v[i], vmax[i])
    }
#undef ADAM_ONE
#undef ADAM_ST
}

// ---------------------------------------------------------------- uint8 -> float/255
__global__ __launch_bounds__(EW_THREADS) void k_u8_to_unit_float(
    const unsigned char* __restrict__ in, float* __restrict__ out, size_t n, int vec) {
    const size_t tid = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    if (vec) {
        const size_t n4 = n >> 2;
        const uchar4* i4 = reinterpret_cast<const uchar4*>(in);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (size_t i = tid; i < n4; i += nthreads) {
            const uchar4 b = i4[i];
            // true division keeps bit-parity with numpy's astype(float32)/255
            o4[i] = make_float4(b.x / 255.f, b.y / 255.f, b.z / 255.f, b.w / 255.f);
        }
        for (size_t i = (n4 << 2) + tid; i < n; i += nthreads) out[i] = in[i] / 255.f;
    } else {
        for (size_t i = tid; i < n; i += nthreads) out[i] = in[i] / 255.f;
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers (called from capi.hip)
// ---------------------------------------------------------------------------------------------
int bn_launch_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, float slope,
                      hipStream_t st) {
    const int vec = aligned16(dy) && aligned16(y) && aligned16(dpre);
    hipLaunchKernelGGL(k_act_bwd, dim3(ew_blocks(vec ? n / 4 + 1 : n)), dim3(EW_THREADS), 0, st, dy,
                       y, dpre, n, vec, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// y = act(x): the sigmoid after the optional last dense decoder layer (aes.py:345-359)
__global__ __launch_bounds__(EW_THREADS) void k_act_fwd(const float* __restrict__ x,
                                                        float* __restrict__ y, size_t n, int act,
                                                        float slope) {
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += nthreads)
        y[i] = bn_apply_act(x[i], act, slope);
}

int bn_launch_act_fwd(const float* x, float* y, size_t n, int act, float slope, hipStream_t st) {
    hipLaunchKernelGGL(k_act_fwd, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, x, y, n, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_sqerr_frame_sums(const float* pred, const float* target, const float* mask,
                               float* frame_sums, int N, size_t D, hipStream_t st) {
    const int vec = aligned16(pred) && aligned16(target) && (!mask || aligned16(mask)) &&
                    (D % 4 == 0);
    hipLaunchKernelGGL(k_sqerr_frame_sums, dim3(N), dim3(EW_THREADS), 0, st, pred, target, mask,
                       frame_sums, D, vec);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_sqerr_bwd(const float* pred, const float* target, const float* mask, float* dpred,
                        size_t n, float scale, const float* gscale, hipStream_t st) {
    const int vec = aligned16(pred) && aligned16(target) && aligned16(dpred) &&
                    (!mask || aligned16(mask));
    hipLaunchKernelGGL(k_sqerr_bwd, dim3(ew_blocks(vec ? n / 4 + 1 : n)), dim3(EW_THREADS), 0, st,
                       pred, target, mask, dpred, n, scale, gscale, vec);
    BN_LAUNCH_CHECK();
    return 0;
}

// t[n, :] *= frame_scale[n] * group_scale[group_of_frame[n]]: the per-chunk loss normalisation and
// upstream gradient applied to the fused loss epilogue's dL/dpre (grid.y = frame)
__global__ __launch_bounds__(EW_THREADS) void k_scale_frames(
    float* __restrict__ t, const float* __restrict__ frame_scale,
    const float* __restrict__ group_scale, const int* __restrict__ group_of_frame, size_t D,
    int vec) {
    const int n = blockIdx.y;
    float sc = frame_scale[n];
    if (group_scale) sc *= group_scale[group_of_frame[n]];
    float* row = t + (size_t)n * D;
    const size_t tid = (size_t)blockIdx.x * EW_THREADS + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * EW_THREADS;
    if (vec) {
        float4* r4 = reinterpret_cast<float4*>(row);
        for (size_t i = tid; i < (D >> 2); i += nthreads) {
            float4 v = r4[i];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            r4[i] = v;
        }
    } else {
        for (size_t i = tid; i < D; i += nthreads) row[i] *= sc;
    }
}

int bn_launch_scale_frames(float* t, const float* frame_scale, const float* group_scale,
                           const int* group_of_frame, int N, size_t D, hipStream_t st) {
    const int vec = aligned16(t) && (D % 4 == 0);
    const size_t per = vec ? D / 4 : D;
    int bx = (int)((per + EW_THREADS - 1) / EW_THREADS);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_scale_frames, dim3(bx, N), dim3(EW_THREADS), 0, st, t, frame_scale,
                       group_scale, group_of_frame, D, vec);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_reduce_sum(const float* in, float* out, size_t n, float scale, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(EW_THREADS), 0, st, in, out, n, scale);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                          size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_reparam_fwd, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, mu, logvar, eps,
                       z, n);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_kl_rows(const float* mu, const float* logvar, float* kl_rows, int N, int D,
                      hipStream_t st) {
    const int rows_per_block = EW_THREADS / BN_WAVE;
    hipLaunchKernelGGL(k_kl_rows, dim3((N + rows_per_block - 1) / rows_per_block),
                       dim3(EW_THREADS), 0, st, mu, logvar, kl_rows, N, D);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_reparam_bwd(const float* dz, const float* z, const float* mu, float* dlogvar,
                          size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_reparam_bwd, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, dz, z, mu,
                       dlogvar, n);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_kl_bwd(const float* mu, const float* logvar, float* dmu, float* dlogvar, size_t n,
                     float scale, const float* gscale, hipStream_t st) {
    hipLaunchKernelGGL(k_kl_bwd, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, st, mu, logvar, dmu,
                       dlogvar, n, scale, gscale);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_adam(float* p, const float* g, float* m, float* v, float* vmax, size_t n, float lr,
                   float b1, float b2, float eps, float wd, int step, hipStream_t st) {
    // bias corrections in double on the host, as torch does with python floats
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    int vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && aligned16(vmax);
    if (vec && n * 4 < 0x7fffffffull) {
        vec = 2;                                   // arenas addressable by a buffer descriptor
        const char* e = bn_tune_env("BN_ADAM_WT");
        if (e && e[0] == '0') vec = 1;
    }
    BN_LAUNCH_MAIN(k_adam_amsgrad, dim3(ew_blocks(vec ? n / 4 + 1 : n)), dim3(EW_THREADS), 0,
                       st, p, g, m, v, vmax, n, step_size, b1, b2, bc2_sqrt, eps, wd, vec);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_u8_to_unit_float(const unsigned char* in, float* out, size_t n, hipStream_t st) {
    const int vec = ((((uintptr_t)in) & 3u) == 0) && aligned16(out);
    hipLaunchKernelGGL(k_u8_to_unit_float, dim3(ew_blocks(vec ? n / 4 + 1 : n)), dim3(EW_THREADS),
                       0, st, in, out, n, vec);
    BN_LAUNCH_CHECK();
    return 0;
}
