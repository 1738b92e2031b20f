// Generic (any kernel size <= 16, any stride) direct convolution kernels.
//
// These are the shape-agnostic fallbacks of the three kernel families (bn_common.h).  They are
// im2col-free direct loops with per-thread register blocking over output channels; the
// benchmark layer shapes (k5 s2 / k5 s5) are served by the MFMA kernels in conv_mfma.hip and
// the HBM-bound kernels in conv_edge.hip -- the dispatcher in capi.hip picks.
#include "bn_common.h"
#include "bn_launch.h"

#define GEN_THREADS 256
#define GEN_MT 8          // output channels per thread
#define GEN_MAXS 16

// out = small side; weights [Cs][Cb][R][S]
__global__ __launch_bounds__(GEN_THREADS) void k_down_generic(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, int act, int dact,
    float slope) {
    const int pix = blockIdx.x * GEN_THREADS + threadIdx.x;
    const int npix = g.Hs * g.Ws;
    const int m0 = blockIdx.y * GEN_MT;
    const int n = blockIdx.z;
    if (pix >= npix) return;
    const int p = pix / g.Ws, q = pix - p * g.Ws;
    const int RS = g.R * g.S;
    float acc[GEN_MT];
#pragma unroll
    for (int j = 0; j < GEN_MT; ++j) acc[j] = 0.f;

    const int h0 = p * g.stride - g.pt, w0 = q * g.stride - g.pl;
    for (int c = 0; c < g.Cb; ++c) {
        const float* bp = big + ((size_t)n * g.Cb + c) * g.Hb * g.Wb;
        for (int r = 0; r < g.R; ++r) {
            const int hb = h0 + r;
            if (hb < 0 || hb >= g.Hb) continue;
            for (int s = 0; s < g.S; ++s) {
                const int wb = w0 + s;
                if (wb < 0 || wb >= g.Wb) continue;
                const float v = bp[(size_t)hb * g.Wb + wb];
#pragma unroll
                for (int j = 0; j < GEN_MT; ++j) {
                    const int m = min(m0 + j, g.Cs - 1);
                    acc[j] = fmaf(v, w[((size_t)m * g.Cb + c) * RS + r * g.S + s], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GEN_MT; ++j) {
        const int m = m0 + j;
        if (m >= g.Cs) break;
        const size_t idx = ((size_t)n * g.Cs + m) * npix + pix;
        float v = acc[j] + (bias ? bias[m] : 0.f);
        v = bn_apply_act(v, act, slope);
        if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
        out[idx] = v;
    }
}

// out = big side; weights [Cs][Cb][R][S]; out channel m indexes the big side
__global__ __launch_bounds__(GEN_THREADS) void k_up_generic(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, int act, int dact,
    float slope) {
    const int pix = blockIdx.x * GEN_THREADS + threadIdx.x;
    const int npix = g.Hb * g.Wb;
    const int m0 = blockIdx.y * GEN_MT;
    const int n = blockIdx.z;
    if (pix >= npix) return;
    const int h = pix / g.Wb, x = pix - h * g.Wb;
    const int RS = g.R * g.S;
    float acc[GEN_MT];
#pragma unroll
    for (int j = 0; j < GEN_MT; ++j) acc[j] = 0.f;

    const int hh = h + g.pt, ww = x + g.pl;
    // p*stride + r == hh with 0 <= r < R  ->  p in [ceil((hh-R+1)/st), floor(hh/st)]
    int p_lo = hh - (g.R - 1);
    p_lo = p_lo <= 0 ? 0 : (p_lo + g.stride - 1) / g.stride;
    const int p_hi = min(g.Hs - 1, hh / g.stride);
    int q_lo = ww - (g.S - 1);
    q_lo = q_lo <= 0 ? 0 : (q_lo + g.stride - 1) / g.stride;
    const int q_hi = min(g.Ws - 1, ww / g.stride);

    for (int c = 0; c < g.Cs; ++c) {
        const float* sp = small + ((size_t)n * g.Cs + c) * g.Hs * g.Ws;
        const float* wp = w + (size_t)c * g.Cb * RS;
        for (int p = p_lo; p <= p_hi; ++p) {
            const int r = hh - p * g.stride;
            for (int q = q_lo; q <= q_hi; ++q) {
                const int s = ww - q * g.stride;
                const float v = sp[(size_t)p * g.Ws + q];
#pragma unroll
                for (int j = 0; j < GEN_MT; ++j) {
                    const int m = min(m0 + j, g.Cb - 1);
                    acc[j] = fmaf(v, wp[(size_t)m * RS + r * g.S + s], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GEN_MT; ++j) {
        const int m = m0 + j;
        if (m >= g.Cb) break;
        const size_t idx = ((size_t)n * g.Cb + m) * npix + pix;
        float v = acc[j] + (bias ? bias[m] : 0.f);
        v = bn_apply_act(v, act, slope);
        if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
        out[idx] = v;
    }
}

__device__ __forceinline__ float bn_block_reduce_256(float v, float* red) {
    // wave64 shuffle tree, then 4 partials through LDS; every thread gets the total
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// dW[a][b][r][0..S) for one (b, a, r) per block; deterministic strided partial sums + tree
__global__ __launch_bounds__(GEN_THREADS) void k_wgrad_generic(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ dw,
    BnGeom g, int accumulate) {
    __shared__ float red[4];
    const int b = blockIdx.x, a = blockIdx.y, r = blockIdx.z;
    const int npix = g.Hs * g.Ws;
    const long total = (long)g.N * npix;
    float acc[GEN_MAXS];
#pragma unroll
    for (int s = 0; s < GEN_MAXS; ++s) acc[s] = 0.f;

    for (long i = threadIdx.x; i < total; i += GEN_THREADS) {
        const int n = (int)(i / npix);
        const int pix = (int)(i - (long)n * npix);
        const int p = pix / g.Ws, q = pix - p * g.Ws;
        const int hb = p * g.stride + r - g.pt;
        if (hb < 0 || hb >= g.Hb) continue;
        const float v = small[((size_t)n * g.Cs + a) * npix + pix];
        const float* bp = big + (((size_t)n * g.Cb + b) * g.Hb + hb) * g.Wb;
        const int w0 = q * g.stride - g.pl;
#pragma unroll
        for (int s = 0; s < GEN_MAXS; ++s) {
            const int wb = w0 + s;
            if (s < g.S && wb >= 0 && wb < g.Wb) acc[s] = fmaf(v, bp[wb], acc[s]);
        }
    }
    float* dst = dw + (((size_t)a * g.Cb + b) * g.R + r) * g.S;
#pragma unroll
    for (int s = 0; s < GEN_MAXS; ++s) {
        if (s < g.S) {   // block-uniform
            const float t = bn_block_reduce_256(acc[s], red);
            if (threadIdx.x == 0) dst[s] = accumulate ? dst[s] + t : t;
        }
    }
}

// db[c] (+)= sum_{n,pix} t[n,c,pix]
__global__ __launch_bounds__(GEN_THREADS) void k_channel_sum(
    const float* __restrict__ t, float* __restrict__ db, int N, int C, int npix, int accumulate) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    if (npix >= GEN_THREADS) {
        for (int n = 0; n < N; ++n) {
            const float* tp = t + ((size_t)n * C + c) * npix;
            for (int i = threadIdx.x; i < npix; i += GEN_THREADS) acc += tp[i];
        }
    } else {
        // few pixels per frame (e.g. 2x2 maps): spread (frame, pixel) pairs over the threads
        const int total = N * npix;
        for (int i = threadIdx.x; i < total; i += GEN_THREADS) {
            const int n = i / npix, px = i - n * npix;
            acc += t[((size_t)n * C + c) * npix + px];
        }
    }
    const float s = bn_block_reduce_256(acc, red);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
int bn_launch_down_generic(const float* big, const float* w, const float* bias, float* out,
                           const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                           hipStream_t st) {
    if (g.R > GEN_MAXS || g.S > GEN_MAXS) return BN_E_SHAPE;
    dim3 grid((g.Hs * g.Ws + GEN_THREADS - 1) / GEN_THREADS, (g.Cs + GEN_MT - 1) / GEN_MT, g.N);
    hipLaunchKernelGGL(k_down_generic, grid, dim3(GEN_THREADS), 0, st, big, w, bias, out, dact_src,
                       g, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_up_generic(const float* small, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         hipStream_t st) {
    if (g.R > GEN_MAXS || g.S > GEN_MAXS) return BN_E_SHAPE;
    dim3 grid((g.Hb * g.Wb + GEN_THREADS - 1) / GEN_THREADS, (g.Cb + GEN_MT - 1) / GEN_MT, g.N);
    hipLaunchKernelGGL(k_up_generic, grid, dim3(GEN_THREADS), 0, st, small, w, bias, out, dact_src,
                       g, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_wgrad_generic(const float* small, const float* big, float* dw, const BnGeom& g,
                            int accumulate, hipStream_t st) {
    if (g.R > GEN_MAXS || g.S > GEN_MAXS) return BN_E_SHAPE;
    dim3 grid(g.Cb, g.Cs, g.R);
    hipLaunchKernelGGL(k_wgrad_generic, grid, dim3(GEN_THREADS), 0, st, small, big, dw, g,
                       accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}

// two-pass variant for large tensors: (channel, frame-slice) partial sums, then a fixed-order
// combine -- one block per channel cannot pull a 100 MB gradient tensor through 32..64 CUs
__global__ __launch_bounds__(GEN_THREADS) void k_channel_sum_part(
    const float* __restrict__ t, float* __restrict__ part, int N, int C, int npix, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int n_beg = (int)((long)sp * N / S), n_end = (int)((long)(sp + 1) * N / S);
    float acc = 0.f;
    const bool vec = (npix & 3) == 0 && ((((uintptr_t)t) & 15u) == 0);
    const int per = npix >> 2;                       // float4 groups of one plane
    if (vec && per < GEN_THREADS && (GEN_THREADS % per) == 0) {
        // small planes (8x8 maps): a thread keeps its group position and walks the frames, the
        // block covers GEN_THREADS / per frames per pass (one plane per frame would leave most
        // lanes idle)
        const int fpp = GEN_THREADS / per, q = threadIdx.x % per;
        for (int n = n_beg + threadIdx.x / per; n < n_end; n += fpp) {
            const float4 v = reinterpret_cast<const float4*>(t + ((size_t)n * C + c) * npix)[q];
            acc += (v.x + v.y) + (v.z + v.w);
        }
    } else
    for (int n = n_beg; n < n_end; ++n) {
        const float* tp = t + ((size_t)n * C + c) * npix;
        if (vec) {
            const float4* t4 = reinterpret_cast<const float4*>(tp);
            for (int i = threadIdx.x; i < (npix >> 2); i += GEN_THREADS) {
                const float4 v = t4[i];
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = threadIdx.x; i < npix; i += GEN_THREADS) acc += tp[i];
        }
    }
    const float s = bn_block_reduce_256(acc, red);
    if (threadIdx.x == 0) part[(size_t)c * S + sp] = s;
}

// one wave per channel: a lane adds its <= S / 64 slices, the lanes are added as a tree (fixed order).  (Round 6: it was
// one thread walking all S slices -- 256 sequential additions; the single-channel sum of a zero-mean gradient, whose
// cancellation multiplies every rounding error by ~200, came out at 6e-6 where the CPU's pairwise sum gives 2e-7.)
__global__ __launch_bounds__(64) void k_channel_sum_final(const float* __restrict__ part, float* __restrict__ db,
                                                          int C, int S, int accumulate) {
    const int c = blockIdx.x;
    float v = 0.f;
    for (int s = threadIdx.x; s < S; s += 64) v += part[(size_t)c * S + s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + v : v;
}

static int channel_sum_splits(int N, int C, int npix) {
    if ((long)N * npix < 8192) return 1;
    int s = 1024 / C;
    if (s > 256) s = 256;     // one channel (dec.convT4's bias): 256 frame slices
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

size_t bn_channel_sum_ws_bytes(int N, int C, int npix) {
    const int s = channel_sum_splits(N, C, npix);
    return s > 1 ? (size_t)C * s * sizeof(float) : 0;
}

int bn_launch_channel_sum(const float* t, float* db, int N, int C, int npix, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    const int S = channel_sum_splits(N, C, npix);
    if (S > 1 && ws && ws_bytes >= (size_t)C * S * sizeof(float)) {
        hipLaunchKernelGGL(k_channel_sum_part, dim3(C, S), dim3(GEN_THREADS), 0, st, t, (float*)ws,
                           N, C, npix, S);
        BN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_channel_sum_final, dim3(C), dim3(64), 0, st,
                           (const float*)ws, db, C, S, accumulate);
        BN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_channel_sum, dim3(C), dim3(GEN_THREADS), 0, st, t, db, N, C, npix,
                       accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// second pass of split reductions (declared in bn_reduce.h)
// ---------------------------------------------------------------------------------------------
#include "bn_reduce.h"
__global__ __launch_bounds__(1024) void k_sum_partials(const float* __restrict__ part,
                                                       float* __restrict__ out, int total,
                                                       int splits, int accumulate, int ab_elems,
                                                       int ntap, int row_len, int row_stride) {
    __shared__ float red[16][64];
    const int il = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const int nz = blockDim.x >> 6;                 // z-lanes: 4, or 16 for many partials
    const int i = blockIdx.x * 64 + il;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int z = zl;
        for (; z + 3 * nz < splits; z += 4 * nz) {
            s0 += part[(size_t)z * total + i];
            s1 += part[(size_t)(z + nz) * total + i];
            s2 += part[(size_t)(z + 2 * nz) * total + i];
            s3 += part[(size_t)(z + 3 * nz) * total + i];
        }
        for (; z < splits; z += nz) s0 += part[(size_t)z * total + i];
    }
    red[zl][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (nz == 16 && zl < 4)
        red[zl][il] = (red[zl][il] + red[zl + 4][il]) + (red[zl + 8][il] + red[zl + 12][il]);
    if (nz == 16) __syncthreads();
    if (zl == 0 && i < total) {
        const float v = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
        size_t o = i;
        if (ab_elems > 0) {
            const int tap = i / ab_elems;
            o = (size_t)(i - tap * ab_elems) * ntap + tap;
        } else if (row_len > 0) {
            const int row = i / row_len;
            o = (size_t)row * row_stride + (i - row * row_len);
        }
        out[o] = accumulate ? out[o] + v : v;
    }
}

__global__ __launch_bounds__(256) void k_sum_partials_pair(
    const float* __restrict__ part_a, float* __restrict__ out_a, int total_a, int ab_elems, int ntap,
    const float* __restrict__ part_b, float* __restrict__ out_b, int total_b, int splits,
    int accumulate, int blocks_a) {
    __shared__ float red[4][64];
    const bool job_b = (int)blockIdx.x >= blocks_a;                  // block-uniform
    const float* part = job_b ? part_b : part_a;
    float* out = job_b ? out_b : out_a;
    const int total = job_b ? total_b : total_a;
    const int il = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const int i = (job_b ? (int)blockIdx.x - blocks_a : (int)blockIdx.x) * 64 + il;
    // same association as k_sum_partials with 4 z-lanes
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int z = zl;
        for (; z + 12 < splits; z += 16) {
            s0 += part[(size_t)z * total + i];
            s1 += part[(size_t)(z + 4) * total + i];
            s2 += part[(size_t)(z + 8) * total + i];
            s3 += part[(size_t)(z + 12) * total + i];
        }
        for (; z < splits; z += 4) s0 += part[(size_t)z * total + i];
    }
    red[zl][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zl == 0 && i < total) {
        const float v = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
        size_t o = i;
        if (!job_b && ab_elems > 0) {
            const int tap = i / ab_elems;
            o = (size_t)(i - tap * ab_elems) * ntap + tap;
        }
        out[o] = accumulate ? out[o] + v : v;
    }
}

