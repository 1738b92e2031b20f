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
#define BNK_MAX_CHUNKS 4

// Row ranges whose statistics are separate (the reference's 200-frame chunks inside one pass over
// the batch): blockIdx.z / a frame's range selects the chunk, statistics are laid out [chunk][C].
struct BnChunks {
    int n;
    int beg[BNK_MAX_CHUNKS], end[BNK_MAX_CHUNKS];
    float scale[BNK_MAX_CHUNKS];          // per-chunk factor (1 / count, running-estimate factor, ...)
    float aux[BNK_MAX_CHUNKS];            // second per-chunk factor (unbiasing)
    // frame slices of the (channel, slice) grids: chunk z owns the global slices [sl_beg[z], sl_beg[z] + sl_n[z]),
    // their number proportional to the chunk's length (round 6: with S slices PER chunk the 56-frame chunk's
    // workgroups finished in a quarter of the time of the 200-frame chunk's and left the tail half empty)
    int sl_beg[BNK_MAX_CHUNKS], sl_n[BNK_MAX_CHUNKS];
};
static BnChunks bnk_one_chunk(int N, float scale = 1.f, float aux = 1.f) {
    BnChunks ch;
    ch.n = 1;
    for (int i = 0; i < BNK_MAX_CHUNKS; ++i) {
        ch.beg[i] = 0; ch.end[i] = 0; ch.scale[i] = scale; ch.aux[i] = aux; ch.sl_beg[i] = 0; ch.sl_n[i] = 0;
    }
    ch.end[0] = N;
    return ch;
}
// S slices in total over the chunks, by length, at least one each, none longer than its chunk; -> the total
static int bnk_set_slices(BnChunks* ch, int S) {
    int N = 0;
    for (int z = 0; z < ch->n; ++z) N += ch->end[z] - ch->beg[z];
    if (S < ch->n) S = ch->n;
    int used = 0;
    for (int z = 0; z < ch->n; ++z) {
        const int len = ch->end[z] - ch->beg[z];
        int k = N > 0 ? (int)(((long)S * len + N / 2) / N) : 1;
        if (k < 1) k = 1;
        if (k > len && len > 0) k = len;
        ch->sl_n[z] = k;
        used += k;
    }
    // the rounding's surplus / deficit goes to the longest chunk
    int big = 0;
    for (int z = 1; z < ch->n; ++z)
        if (ch->end[z] - ch->beg[z] > ch->end[big] - ch->beg[big]) big = z;
    if (used > S && ch->sl_n[big] - (used - S) >= 1) { ch->sl_n[big] -= used - S; used = S; }
    if (used < S) {
        int add = S - used;
        const int room = (ch->end[big] - ch->beg[big]) - ch->sl_n[big];
        if (add > room) add = room > 0 ? room : 0;
        ch->sl_n[big] += add;
        used += add;
    }
    int pos = 0;
    for (int z = 0; z < ch->n; ++z) { ch->sl_beg[z] = pos; pos += ch->sl_n[z]; }
    return pos;
}
// global slice -> chunk; its frames are n_beg + k * n_step for k < n_end - n_beg: the slices of a chunk INTERLEAVE
// (slice sp takes frames sp, sp + S, sp + 2 S, ... of the chunk).  With contiguous frame ranges per slice the ~1000
// workgroups of a launch were ~1000 separate streams through the tensor (4.3-4.6 TB/s for the two reductions against
// 6.2 for the flat normalise pass); interleaved, the workgroups that run at the same time read the same frames'
// planes, i.e. one region that moves through memory in address order (round 6).
__device__ __forceinline__ int bnk_slice(const BnChunks& ch, int gs, int* n_beg, int* n_end, int* n_step) {
    int z = 0;
#pragma unroll
    for (int i = 1; i < BNK_MAX_CHUNKS; ++i)
        if (i < ch.n && gs >= ch.sl_beg[i]) z = i;
    const int sp = gs - ch.sl_beg[z], S = ch.sl_n[z], N = ch.end[z] - ch.beg[z];
    *n_beg = ch.beg[z] + sp;
    *n_end = *n_beg + (N - sp + S - 1) / S;              // (count of frames, not the last frame)
    *n_step = S;
    return z;
}

__device__ __forceinline__ int bnk_chunk_of(const BnChunks& ch, int n) {
    int z = 0;
#pragma unroll
    for (int i = 1; i < BNK_MAX_CHUNKS; ++i)
        if (i < ch.n && n >= ch.beg[i]) z = i;
    return z;
}

// 16-byte store that does not allocate in the L2s: the normalised activations / input gradients of a large layer are
// read next by another kernel from HBM anyway
__device__ __forceinline__ void bnk_store_nt(float4* p, const float4 v) {
    typedef float bnk_f4 __attribute__((ext_vector_type(4)));
    bnk_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<bnk_f4*>(p));
}

__device__ __forceinline__ float bnk_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// part[c][sp] = sum over frames [n_beg, n_end) and pixels of (x - shift[c])^p   (p = 1 or 2).
// The (frame, pixel) pairs of a slice are walked as ONE index range: with a loop over frames
// around a loop over pixels the 2x2 maps of the deepest layer kept 4 of 256 threads busy for 100
// dependent iterations (80-100 us for 400 KB; round 3).
template <int POW>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_moment_part(
    const float* __restrict__ x, const float* __restrict__ shift, float* __restrict__ part,
    BnChunks ch, int C, int HW, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, sp = blockIdx.y, z = blockIdx.z;
    const int N = ch.end[z] - ch.beg[z];
    const int n_beg = ch.beg[z] + (int)((long)sp * N / S), n_end = ch.beg[z] + (int)((long)(sp + 1) * N / S);
    const float sh = shift ? shift[z * C + c] : 0.f;
    float acc = 0.f, acc2 = 0.f;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)x) & 15u) == 0);
    if (vec) {
        const unsigned hw4 = HW >> 2, cnt = (unsigned)(n_end - n_beg) * hw4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (unsigned e = threadIdx.x; e < cnt; e += 2 * BNK_THREADS) {
            const unsigned e2 = e + BNK_THREADS;
            const unsigned n = e / hw4, i = e - n * hw4;
            float4 v = x4[((size_t)(n_beg + n) * C + c) * hw4 + i];
            float4 w = make_float4(sh, sh, sh, sh);
            if (e2 < cnt) {
                const unsigned n2 = e2 / hw4, i2 = e2 - n2 * hw4;
                w = x4[((size_t)(n_beg + n2) * C + c) * hw4 + i2];
            }
            v.x -= sh; v.y -= sh; v.z -= sh; v.w -= sh;
            w.x -= sh; w.y -= sh; w.z -= sh; w.w -= sh;
            if (POW == 2) {
                v.x *= v.x; v.y *= v.y; v.z *= v.z; v.w *= v.w;
                w.x *= w.x; w.y *= w.y; w.z *= w.z; w.w *= w.w;
            }
            acc += (v.x + v.y) + (v.z + v.w);
            acc2 += (w.x + w.y) + (w.z + w.w);
        }
    } else {
        const unsigned cnt = (unsigned)(n_end - n_beg) * HW;
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / HW, i = e - n * HW;
            const float v = x[((size_t)(n_beg + n) * C + c) * HW + i] - sh;
            acc += POW == 2 ? v * v : v;
        }
    }
    const float s = bnk_block_sum(acc + acc2, red);
    if (threadIdx.x == 0) part[((size_t)z * C + c) * S + sp] = s;
}

// y = x * sc + sh with sc = invstd * gamma, sh = beta - mean * sc: ONE spelling (explicit fused
// multiply-adds) for the forward pass and for the backward kernels that rebuild the sign of the
// activation's input from x instead of reading y back (round 4) -- bit-identical in all of them
__device__ __forceinline__ void bnk_affine(float mean, float invstd, const float* gamma, const float* beta,
                                           int c, float* sc, float* sh) {
    *sc = invstd * (gamma ? gamma[c] : 1.f);
    *sh = fmaf(-mean, *sc, beta ? beta[c] : 0.f);
}

// The shift of the one-pass statistics: the mean of eight samples of the channel spread over the chunk's
// frames and pixels.  (Round 4 took the chunk's FIRST value -- the corner pixel of the first frame, which
// after a zero-padded convolution is often an outlier: the variance then loses a factor
// 1 + (mean - s)^2 / var of precision, ADVICE r4.  One outlier among eight samples moves the shift by an
// eighth of its distance.)  The same function in the statistics kernel and in its finalize kernel.
__device__ __forceinline__ float bnk_shift(const float* __restrict__ x, int beg, int end, int c, int C, int HW) {
    const int N = end - beg;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int n = beg + (int)(((long)k * N) >> 3);
        const int i = (int)(((long)(2 * k + 1) * HW) >> 4);
        s += x[((size_t)n * C + c) * HW + i];
    }
    return s * 0.125f;
}

// Statistics in ONE pass over x (round 4; they were two: mean, then the centred second moment):
// part1 = sum (x - s), part2 = sum (x - s)^2 around a per-channel shift s inside the data (bnk_shift).  With s inside the data the subtraction var = E[(x-s)^2] - (E[x-s])^2 loses a
// factor (1 + (mean - s)^2 / var) of precision -- a few units in the last place for any channel
// that is not constant, against the catastrophic E[x^2] - mean^2 -- and 0.5 GB less traffic per
// training step of the batch-norm model.
__global__ __launch_bounds__(BNK_THREADS) void k_bn_stats_part(
    const float* __restrict__ x, float* __restrict__ part1, float* __restrict__ part2, BnChunks ch,
    int C, int HW, int S) {
    __shared__ float red[4];
    const int c = blockIdx.x, gs = blockIdx.y;
    int n_beg, n_end, n_step;
    const int z = bnk_slice(ch, gs, &n_beg, &n_end, &n_step);
    const float sh = bnk_shift(x, ch.beg[z], ch.end[z], c, C, HW);
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)x) & 15u) == 0);
    if (vec) {
        const unsigned hw4 = HW >> 2, cnt = (unsigned)(n_end - n_beg) * hw4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (unsigned e = threadIdx.x; e < cnt; e += 2 * BNK_THREADS) {
            const unsigned e2 = e + BNK_THREADS;
            const unsigned n = e / hw4, i = e - n * hw4;
            float4 v = x4[((size_t)(n_beg + n * n_step) * C + c) * hw4 + i];
            float4 w = make_float4(sh, sh, sh, sh);
            if (e2 < cnt) {
                const unsigned n2 = e2 / hw4, i2 = e2 - n2 * hw4;
                w = x4[((size_t)(n_beg + n2 * n_step) * C + c) * hw4 + i2];
            }
            v.x -= sh; v.y -= sh; v.z -= sh; v.w -= sh;
            w.x -= sh; w.y -= sh; w.z -= sh; w.w -= sh;
            a1 += (v.x + v.y) + (v.z + v.w);
            b1 += (w.x + w.y) + (w.z + w.w);
            a2 += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
            b2 += fmaf(w.x, w.x, w.y * w.y) + fmaf(w.z, w.z, w.w * w.w);
        }
    } else {
        const unsigned cnt = (unsigned)(n_end - n_beg) * HW;
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / HW, i = e - n * HW;
            const float v = x[((size_t)(n_beg + n * n_step) * C + c) * HW + i] - sh;
            a1 += v;
            a2 = fmaf(v, v, a2);
        }
    }
    const float s1 = bnk_block_sum(a1 + b1, red);
    const float s2 = bnk_block_sum(a2 + b2, red);
    if (threadIdx.x == 0) {
        part1[(size_t)c * S + gs] = s1;
        part2[(size_t)c * S + gs] = s2;
    }
}

// mean / biased variance / invstd of every chunk from the one-pass partial sums, the running
// estimates (chunks in order: one update per chunk, factor scale[z], unbiasing aux[z]) and the batch
// counter in ONE launch (they were two combine launches, the finalize kernel and torch's add_)
__global__ void k_bn_stats_finalize(const float* __restrict__ x, const float* __restrict__ part1,
                                    const float* __restrict__ part2, float* __restrict__ mean,
                                    float* __restrict__ var, float* __restrict__ invstd,
                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                    long long* __restrict__ num_batches, int C, int HW, int S, float eps,
                                    BnChunks ch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches) *num_batches += ch.n;
    if (c >= C) return;
    for (int z = 0; z < ch.n; ++z) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) {
            s1 += part1[((size_t)z * C + c) * S + s];
            s2 += part2[((size_t)z * C + c) * S + s];
        }
        const float inv_n = 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW);
        const float sh = bnk_shift(x, ch.beg[z], ch.end[z], c, C, HW);
        const float d = s1 * inv_n;
        const float m = sh + d;
        const float v = fmaxf(fmaf(-d, d, s2 * inv_n), 0.f);
        mean[z * C + c] = m;
        if (var) var[z * C + c] = v;
        if (invstd) invstd[z * C + c] = 1.0f / sqrtf(v + eps);
        const float momentum = ch.scale[z];
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * v * ch.aux[z];
    }
}

// out[z][c] = scale[z] * sum_s part[z][c][s]
__global__ void k_bn_combine(const float* __restrict__ part, float* __restrict__ out, int C, int S,
                             BnChunks ch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * ch.n) return;
    float v = 0.f;
#pragma unroll 8
    for (int s = 0; s < S; ++s) v += part[(size_t)i * S + s];
    out[i] = v * ch.scale[i / C];
}

// invstd = 1/sqrt(var + eps); running stats (momentum < 0: the caller passes the cumulative
// average factor 1/num_batches_tracked instead, as nn.BatchNorm2d(momentum=None) does)
// (chunks in order: the running estimates see one update per chunk, factor scale[z], unbiasing aux[z])
__global__ void k_bn_finalize(const float* __restrict__ mean, const float* __restrict__ var,
                              float* __restrict__ invstd, float* __restrict__ running_mean,
                              float* __restrict__ running_var, int C, float eps, BnChunks ch) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int z = 0; z < ch.n; ++z) {
        const float m = mean[z * C + c], v = var[z * C + c], momentum = ch.scale[z];
        invstd[z * C + c] = 1.0f / sqrtf(v + eps);
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * v * ch.aux[z];
    }
}

static int bnk_flat_blocks(size_t n) {
    const size_t b = (n + BNK_THREADS - 1) / BNK_THREADS;
    return (int)(b < 16384 ? (b ? b : 1) : 16384);
}

// y = act( (x - mean) * invstd * gamma + beta ), flat over all (frame, channel, pixel group)s: a
// workgroup per (frame, channel) left 252 of 256 threads idle on the 2x2 maps
template <int VEC>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_act_fwd(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    unsigned NC, int C, int HW, int act, float slope, BnChunks ch) {
    const unsigned hwv = HW / VEC, total = NC * hwv;
    for (unsigned e = blockIdx.x * BNK_THREADS + threadIdx.x; e < total; e += gridDim.x * BNK_THREADS) {
        const unsigned nc = e / hwv;
        const int c = nc % C, zc = bnk_chunk_of(ch, nc / C) * C + c;
        float sc, sh;
        bnk_affine(mean[zc], invstd[zc], gamma, beta, c, &sc, &sh);
        if (VEC == 4) {
            const float4 v = reinterpret_cast<const float4*>(x)[e];
            float4 o;
            o.x = bn_apply_act(fmaf(v.x, sc, sh), act, slope);
            o.y = bn_apply_act(fmaf(v.y, sc, sh), act, slope);
            o.z = bn_apply_act(fmaf(v.z, sc, sh), act, slope);
            o.w = bn_apply_act(fmaf(v.w, sc, sh), act, slope);
            reinterpret_cast<float4*>(y)[e] = o;
        } else {
            y[e] = bn_apply_act(fmaf(x[e], sc, sh), act, slope);
        }
    }
}

// backward reductions: part0[c][sp] = sum dz, part1[c][sp] = sum dz * xhat,
// dz = dy * act'(y), xhat = (x - mean) * invstd   (one index range per slice, see above)
// FROMX (identity / LeakyReLU): the sign of the activation's input is rebuilt from x with the forward
// pass's own affine map (bnk_affine) instead of reading y: 0.5 GB less per pass over the batch
template <bool FROMX>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_bwd_part(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ part0,
    float* __restrict__ part1, BnChunks ch, int C, int HW, int S, int act, float slope) {
    __shared__ float red[4];
    const int c = blockIdx.x, gs = blockIdx.y;
    int n_beg, n_end, n_step;
    const int z = bnk_slice(ch, gs, &n_beg, &n_end, &n_step);
    const float m = mean[z * C + c], is = invstd[z * C + c];
    float sc = 1.f, sh = 0.f;
    if (FROMX) bnk_affine(m, is, gamma, beta, c, &sc, &sh);
    const float neg = act == BN_ACT_LRELU ? slope : 1.f;
    float a0 = 0.f, a1 = 0.f;
    const bool vec = (HW & 3) == 0 &&
                     (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy)) & 15u) == 0);
    if (vec) {
        const unsigned hw4 = HW >> 2, cnt = (unsigned)(n_end - n_beg) * hw4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        const float4* d4 = reinterpret_cast<const float4*>(dy);
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / hw4, i = e - n * hw4;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * hw4 + i;
            const float4 xv = x4[o], dv = d4[o];
            float z0, z1, z2, z3;
            if (FROMX) {
                z0 = dv.x * (fmaf(xv.x, sc, sh) > 0.f ? 1.f : neg);
                z1 = dv.y * (fmaf(xv.y, sc, sh) > 0.f ? 1.f : neg);
                z2 = dv.z * (fmaf(xv.z, sc, sh) > 0.f ? 1.f : neg);
                z3 = dv.w * (fmaf(xv.w, sc, sh) > 0.f ? 1.f : neg);
            } else {
                const float4 yv = y4[o];
                z0 = dv.x * bn_act_grad_from_output(yv.x, act, slope);
                z1 = dv.y * bn_act_grad_from_output(yv.y, act, slope);
                z2 = dv.z * bn_act_grad_from_output(yv.z, act, slope);
                z3 = dv.w * bn_act_grad_from_output(yv.w, act, slope);
            }
            a0 += (z0 + z1) + (z2 + z3);
            a1 += (z0 * ((xv.x - m) * is) + z1 * ((xv.y - m) * is)) +
                  (z2 * ((xv.z - m) * is) + z3 * ((xv.w - m) * is));
        }
    } else {
        const unsigned cnt = (unsigned)(n_end - n_beg) * HW;
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / HW, i = e - n * HW;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * HW + i;
            const float dz = dy[o] * (FROMX ? (fmaf(x[o], sc, sh) > 0.f ? 1.f : neg)
                                            : bn_act_grad_from_output(y[o], act, slope));
            a0 += dz;
            a1 += dz * ((x[o] - m) * is);
        }
    }
    const float s0 = bnk_block_sum(a0, red);
    const float s1 = bnk_block_sum(a1, red);
    if (threadIdx.x == 0) {
        part0[(size_t)c * S + gs] = s0;
        part1[(size_t)c * S + gs] = s1;
    }
}

// both sums of the backward pass and the parameter gradients in one launch (they were three)
__global__ void k_bn_bwd_combine(const float* __restrict__ part0, const float* __restrict__ part1,
                                 float* __restrict__ sum0, float* __restrict__ sum1,
                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int S,
                                 int accumulate, int n_chunks) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = (accumulate && dgamma) ? dgamma[c] : 0.f, b = (accumulate && dbeta) ? dbeta[c] : 0.f;
    for (int z = 0; z < n_chunks; ++z) {          // chunk order, as separate backward passes would add
        float v0 = 0.f, v1 = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) {
            v0 += part0[((size_t)z * C + c) * S + s];
            v1 += part1[((size_t)z * C + c) * S + s];
        }
        sum0[z * C + c] = v0;
        sum1[z * C + c] = v1;
        g += v1;
        b += v0;
    }
    if (dgamma) dgamma[c] = g;
    if (dbeta) dbeta[c] = b;
}

// dx = gamma * invstd * (dz - dbeta/n - xhat * dgamma/n), flat like the forward
template <int VEC, bool FROMX>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_bwd_apply(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ sum_dz,
    const float* __restrict__ sum_dzx, float* __restrict__ dx, unsigned NC, int C, int HW,
    int act, float slope, BnChunks ch) {
    const unsigned hwv = HW / VEC, total = NC * hwv;
    const float neg = act == BN_ACT_LRELU ? slope : 1.f;
    for (unsigned e = blockIdx.x * BNK_THREADS + threadIdx.x; e < total; e += gridDim.x * BNK_THREADS) {
        const unsigned nc = e / hwv;
        const int c = nc % C, z = bnk_chunk_of(ch, nc / C), zc = z * C + c;
        const float m = mean[zc], is = invstd[zc], inv_n = ch.scale[z];
        const float g = (gamma ? gamma[c] : 1.f) * is;
        const float k0 = sum_dz[zc] * inv_n, k1 = sum_dzx[zc] * inv_n;
        float sc = 1.f, sh = 0.f;
        if (FROMX) bnk_affine(m, is, gamma, beta, c, &sc, &sh);
        if (VEC == 4) {
            const float4 xv = reinterpret_cast<const float4*>(x)[e];
            const float4 dv = reinterpret_cast<const float4*>(dy)[e];
            float4 f;
            if (FROMX) {
                f.x = fmaf(xv.x, sc, sh) > 0.f ? 1.f : neg;
                f.y = fmaf(xv.y, sc, sh) > 0.f ? 1.f : neg;
                f.z = fmaf(xv.z, sc, sh) > 0.f ? 1.f : neg;
                f.w = fmaf(xv.w, sc, sh) > 0.f ? 1.f : neg;
            } else {
                const float4 yv = reinterpret_cast<const float4*>(y)[e];
                f.x = bn_act_grad_from_output(yv.x, act, slope);
                f.y = bn_act_grad_from_output(yv.y, act, slope);
                f.z = bn_act_grad_from_output(yv.z, act, slope);
                f.w = bn_act_grad_from_output(yv.w, act, slope);
            }
            float4 o;
            o.x = g * (dv.x * f.x - k0 - ((xv.x - m) * is) * k1);
            o.y = g * (dv.y * f.y - k0 - ((xv.y - m) * is) * k1);
            o.z = g * (dv.z * f.z - k0 - ((xv.z - m) * is) * k1);
            o.w = g * (dv.w * f.w - k0 - ((xv.w - m) * is) * k1);
            reinterpret_cast<float4*>(dx)[e] = o;
        } else {
            const float dz = dy[e] * (FROMX ? (fmaf(x[e], sc, sh) > 0.f ? 1.f : neg)
                                            : bn_act_grad_from_output(y[e], act, slope));
            dx[e] = g * (dz - k0 - ((x[e] - m) * is) * k1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6: the per-channel finalize / combine launches folded into their neighbours.  The apply kernels
// run on the (channel, frame-slice, chunk) grid of the partial-sum kernels; a workgroup first adds up
// the S partials of ITS (chunk, channel) -- one wave, a fixed shuffle tree, so every workgroup of the
// channel derives bit-identical statistics -- and then streams its slice.  Slice 0 of a channel
// publishes what the backward pass / the running estimates need.  54 -> 36 batch-norm launches per
// training step of the default architecture.
// ---------------------------------------------------------------------------------------------
// totals of two partial arrays p1[0..S), p2[0..S) (S <= 64) for every thread of the workgroup
__device__ __forceinline__ void bnk_reduce_parts(const float* __restrict__ p1, const float* __restrict__ p2,
                                                 int S, float* red, float* o1, float* o2) {
    __syncthreads();                                   // (red may still be read from a previous call)
    if (threadIdx.x < 64) {
        float a = (int)threadIdx.x < S ? p1[threadIdx.x] : 0.f;
        float b = (int)threadIdx.x < S ? p2[threadIdx.x] : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_xor(a, off, 64);
            b += __shfl_xor(b, off, 64);
        }
        if (threadIdx.x == 0) { red[0] = a; red[1] = b; }
    }
    __syncthreads();
    *o1 = red[0];
    *o2 = red[1];
}

// The arithmetic between the reduced sums and (mean, variance, invstd) -- ONE piece of machine code (noinline) for
// every caller.  The forward pass normalises with values a workgroup derives itself, the backward pass rebuilds the
// activation's sign from the STORED mean / invstd: the two must agree to the last bit, and an inlined copy per call
// site leaves that to the compiler's contraction choices (`shift + s1 * inv_n` fused at one site and not at the
// other moved one pre-activation in 10^8 across zero: tools/fuzz_archs.py, seed 410 with batch norm -- a branch the
// backward pass then took differently from the forward pass).
__device__ __noinline__ void bnk_finish_stats(float s1, float s2, float inv_n, float shift, float eps, float* m,
                                              float* v, float* is) {
    const float d = s1 * inv_n;
    *m = shift + d;
    *v = fmaxf(fmaf(-d, d, s2 * inv_n), 0.f);
    *is = 1.0f / sqrtf(*v + eps);
}

// mean / biased variance / invstd of (chunk z, channel c) from the one-pass partial sums (k_bn_stats_part)
__device__ __forceinline__ void bnk_stats_of(const float* __restrict__ x, const float* __restrict__ part1,
                                             const float* __restrict__ part2, const BnChunks& ch, int z, int c,
                                             int C, int HW, int S, float eps, float* red, float* m, float* v,
                                             float* is) {
    float s1, s2;
    bnk_reduce_parts(part1 + (size_t)c * S + ch.sl_beg[z], part2 + (size_t)c * S + ch.sl_beg[z], ch.sl_n[z], red,
                     &s1, &s2);
    const float inv_n = 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW);
    const float sh = bnk_shift(x, ch.beg[z], ch.end[z], c, C, HW);
    bnk_finish_stats(s1, s2, inv_n, sh, eps, m, v, is);
}

// statistics finalize + y = act((x - mean) * invstd * gamma + beta) of one (channel, slice, chunk)
__global__ __launch_bounds__(BNK_THREADS) void k_bn_act_fwd_fin(
    const float* __restrict__ x, const float* __restrict__ part1, const float* __restrict__ part2,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches, BnChunks ch, int C, int HW, int S,
    float eps, int act, float slope) {
    __shared__ float red[2];
    const int c = blockIdx.x, gs = blockIdx.y;
    int n_beg, n_end, n_step;
    const int z = bnk_slice(ch, gs, &n_beg, &n_end, &n_step);
    float m, v, is;
    if (gs == 0) {
        // this workgroup also owns the channel's running estimates: one update per chunk, in chunk order
        // (factor scale[z], unbiasing aux[z]), and the batch counter
        float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
        float m0 = 0.f, v0 = 0.f, is0 = 0.f;
        for (int zz = 0; zz < ch.n; ++zz) {
            float mz, vz, iz;
            bnk_stats_of(x, part1, part2, ch, zz, c, C, HW, S, eps, red, &mz, &vz, &iz);
            if (zz == 0) { m0 = mz; v0 = vz; is0 = iz; }
            const float momentum = ch.scale[zz];
            rm = (1.f - momentum) * rm + momentum * mz;
            rv = (1.f - momentum) * rv + momentum * vz * ch.aux[zz];
            if (threadIdx.x == 0) {
                mean[zz * C + c] = mz;
                invstd[zz * C + c] = iz;
            }
        }
        if (threadIdx.x == 0) {
            if (running_mean) running_mean[c] = rm;
            if (running_var) running_var[c] = rv;
            if (c == 0 && num_batches) *num_batches += ch.n;
        }
        m = m0; v = v0; is = is0;
    } else {
        bnk_stats_of(x, part1, part2, ch, z, c, C, HW, S, eps, red, &m, &v, &is);
    }
    (void)v;
    float sc, sh;
    bnk_affine(m, is, gamma, beta, c, &sc, &sh);
    const bool vec = (HW & 3) == 0 && (((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0);
    if (vec) {
        const unsigned hw4 = HW >> 2, cnt = (unsigned)(n_end - n_beg) * hw4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (unsigned e = threadIdx.x; e < cnt; e += 2 * BNK_THREADS) {
            const unsigned e2 = e + BNK_THREADS;
            const unsigned n = e / hw4, i = e - n * hw4;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * hw4 + i;
            size_t o2 = o;
            if (e2 < cnt) {
                const unsigned n2 = e2 / hw4, i2 = e2 - n2 * hw4;
                o2 = ((size_t)(n_beg + n2 * n_step) * C + c) * hw4 + i2;
            }
            const float4 a = x4[o], b = x4[o2];
            float4 p, q;
            p.x = bn_apply_act(fmaf(a.x, sc, sh), act, slope);
            p.y = bn_apply_act(fmaf(a.y, sc, sh), act, slope);
            p.z = bn_apply_act(fmaf(a.z, sc, sh), act, slope);
            p.w = bn_apply_act(fmaf(a.w, sc, sh), act, slope);
            q.x = bn_apply_act(fmaf(b.x, sc, sh), act, slope);
            q.y = bn_apply_act(fmaf(b.y, sc, sh), act, slope);
            q.z = bn_apply_act(fmaf(b.z, sc, sh), act, slope);
            q.w = bn_apply_act(fmaf(b.w, sc, sh), act, slope);
            bnk_store_nt(y4 + o, p);
            if (e2 < cnt) bnk_store_nt(y4 + o2, q);
        }
    } else {
        const unsigned cnt = (unsigned)(n_end - n_beg) * HW;
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / HW, i = e - n * HW;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * HW + i;
            y[o] = bn_apply_act(fmaf(x[o], sc, sh), act, slope);
        }
    }
}

// combine of the backward sums + dx = gamma * invstd * (dz - sum_dz / n - xhat * sum_dzx / n) of one
// (channel, slice, chunk); slice 0 of chunk 0 adds the parameter gradients (chunks in order, as separate
// backward passes would)
template <bool FROMX>
__global__ __launch_bounds__(BNK_THREADS) void k_bn_bwd_apply_fin(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ part0, const float* __restrict__ part1,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, BnChunks ch,
    int C, int HW, int S, int act, float slope) {
    __shared__ float red[2];
    const int c = blockIdx.x, gs = blockIdx.y;
    int n_beg, n_end, n_step;
    const int z = bnk_slice(ch, gs, &n_beg, &n_end, &n_step);
    float v0, v1;
    if (gs == 0 && (dgamma || dbeta)) {
        float g = (accumulate && dgamma) ? dgamma[c] : 0.f, b = (accumulate && dbeta) ? dbeta[c] : 0.f;
        float f0 = 0.f, f1 = 0.f;
        for (int zz = 0; zz < ch.n; ++zz) {
            float a0, a1;
            bnk_reduce_parts(part0 + (size_t)c * S + ch.sl_beg[zz], part1 + (size_t)c * S + ch.sl_beg[zz],
                             ch.sl_n[zz], red, &a0, &a1);
            if (zz == 0) { f0 = a0; f1 = a1; }
            g += a1;
            b += a0;
        }
        if (threadIdx.x == 0) {
            if (dgamma) dgamma[c] = g;
            if (dbeta) dbeta[c] = b;
        }
        v0 = f0; v1 = f1;
    } else {
        bnk_reduce_parts(part0 + (size_t)c * S + ch.sl_beg[z], part1 + (size_t)c * S + ch.sl_beg[z], ch.sl_n[z],
                         red, &v0, &v1);
    }
    const int zc = z * C + c;
    const float m = mean[zc], is = invstd[zc], inv_n = ch.scale[z];
    const float g = (gamma ? gamma[c] : 1.f) * is;
    const float k0 = v0 * inv_n, k1 = v1 * inv_n;
    float sc = 1.f, sh = 0.f;
    if (FROMX) bnk_affine(m, is, gamma, beta, c, &sc, &sh);
    const float neg = act == BN_ACT_LRELU ? slope : 1.f;
    const bool vec = (HW & 3) == 0 &&
                     (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15u) == 0);
    if (vec) {
        const unsigned hw4 = HW >> 2, cnt = (unsigned)(n_end - n_beg) * hw4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* y4 = reinterpret_cast<const float4*>(y);
        const float4* d4 = reinterpret_cast<const float4*>(dy);
        float4* o4 = reinterpret_cast<float4*>(dx);
        // two groups per thread and iteration: four 16-byte loads in flight
        for (unsigned e0 = threadIdx.x; e0 < cnt; e0 += 2 * BNK_THREADS)
        for (unsigned e = e0; e < e0 + 2 * BNK_THREADS && e < cnt; e += BNK_THREADS) {
            const unsigned n = e / hw4, i = e - n * hw4;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * hw4 + i;
            const float4 xv = x4[o], dv = d4[o];
            float4 f;
            if (FROMX) {
                f.x = fmaf(xv.x, sc, sh) > 0.f ? 1.f : neg;
                f.y = fmaf(xv.y, sc, sh) > 0.f ? 1.f : neg;
                f.z = fmaf(xv.z, sc, sh) > 0.f ? 1.f : neg;
                f.w = fmaf(xv.w, sc, sh) > 0.f ? 1.f : neg;
            } else {
                const float4 yv = y4[o];
                f.x = bn_act_grad_from_output(yv.x, act, slope);
                f.y = bn_act_grad_from_output(yv.y, act, slope);
                f.z = bn_act_grad_from_output(yv.z, act, slope);
                f.w = bn_act_grad_from_output(yv.w, act, slope);
            }
            float4 r;
            r.x = g * (dv.x * f.x - k0 - ((xv.x - m) * is) * k1);
            r.y = g * (dv.y * f.y - k0 - ((xv.y - m) * is) * k1);
            r.z = g * (dv.z * f.z - k0 - ((xv.z - m) * is) * k1);
            r.w = g * (dv.w * f.w - k0 - ((xv.w - m) * is) * k1);
            bnk_store_nt(o4 + o, r);
        }
    } else {
        const unsigned cnt = (unsigned)(n_end - n_beg) * HW;
        for (unsigned e = threadIdx.x; e < cnt; e += BNK_THREADS) {
            const unsigned n = e / HW, i = e - n * HW;
            const size_t o = ((size_t)(n_beg + n * n_step) * C + c) * HW + i;
            const float dz = dy[o] * (FROMX ? (fmaf(x[o], sc, sh) > 0.f ? 1.f : neg)
                                            : bn_act_grad_from_output(y[o], act, slope));
            dx[o] = g * (dz - k0 - ((x[o] - m) * is) * k1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 6, the deep layers (many channels, small maps: E2-E4 / D0-D1 of the default architecture, a third of the
// batch-norm launches and none of the big tensors): ONE workgroup of 1024 threads owns a channel.  A chunk's
// values of that channel (<= 32 K floats: 200 frames of an 8x8 map four times over) are loaded ONCE into registers; mean, then the centred second moment
// (torch's own two-pass form: no shift, no cancellation) by two workgroup reductions; the normalised values leave
// from the registers.  Chunks one after the other inside the workgroup, so the running estimates see their updates
// in order.  Forward = 1 launch and x read once (was stats + normalise: 2 launches, x read twice); backward = 1
// launch (was sums + apply), x and dz held in registers.  Larger chunks (the 16x16 maps: 51 K floats) stay on the
// two-launch forms: one workgroup per channel is a chain of four dependent memory phases and loses there.
// ---------------------------------------------------------------------------------------------
#define BNO_THREADS 1024
#define BNO_NV 8                                         // float4 groups per thread: 1024 x 8 x 4 = 32 K floats

__device__ __forceinline__ float bno_block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                                     // (red may still be read from the previous reduction)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    const int nw = blockDim.x >> 6;                                  // (256 threads for the smallest maps)
    for (int w = 0; w < nw; ++w) t += red[w];                        // fixed order, every thread the same
    return t;
}

// 256 threads when every chunk of a channel fits them (2x2 maps: 200 groups), else 1024
static int bno_threads(const BnChunks& ch, int HW) {
    long m = 0;
    for (int z = 0; z < ch.n; ++z) {
        const long cnt = (long)(ch.end[z] - ch.beg[z]) * (HW >> 2);
        m = cnt > m ? cnt : m;
    }
    return m <= 256 * BNO_NV ? 256 : BNO_THREADS;
}

static bool bno_fits(const BnChunks& ch, int C, int HW, const void* a, const void* b, const void* c2,
                     const void* d) {
    if ((HW & 3) != 0) return false;
    if ((size_t)ch.end[ch.n - 1] * C * HW >= 0xffffffffull) return false;
    if (((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c2) | ((uintptr_t)d)) & 15u) != 0) return false;
    int total = 0;
    for (int z = 0; z < ch.n; ++z) {
        const long cnt = (long)(ch.end[z] - ch.beg[z]) * (HW >> 2);
        if (cnt > (long)BNO_THREADS * BNO_NV) return false;
        total += ch.end[z] - ch.beg[z];
    }
    static int off = -1;                                 // BN_BN_OWNED=0: the two-launch forms (tuning builds)
    if (off < 0) { const char* e = bn_tune_env("BN_BN_OWNED"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off) return false;
    // few channels: only when the tensor is small anyway (a workgroup per channel must fill the chip)
    return C >= 64 || (size_t)total * C * HW * 4 <= ((size_t)2 << 20);
}

__global__ __launch_bounds__(BNO_THREADS) void k_bn_fwd_owned(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ y, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, long long* __restrict__ num_batches, BnChunks ch, int C, int HW, float eps,
    int act, float slope) {
    __shared__ float red[BNO_THREADS / 64];
    const int c = blockIdx.x;
    const unsigned hw4 = HW >> 2, nthr = blockDim.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
#pragma unroll
    for (int z = 0; z < BNK_MAX_CHUNKS; ++z) {
        if (z >= ch.n) break;                            // (static indices into the chunk table: it stays in scalar registers)
        const unsigned cnt = (unsigned)(ch.end[z] - ch.beg[z]) * hw4;
        float4 v[BNO_NV];
        // float4 index of group k (the launcher checked N C HW < 2^32); not kept: 14 offsets next to 56 data
        // registers spilled.  Maps whose group count divides 1024 (powers of two): linear in k
        const unsigned n0 = threadIdx.x / hw4, i0 = threadIdx.x - n0 * hw4;
        const unsigned off0 = ((unsigned)(ch.beg[z] + n0) * C + c) * hw4 + i0;
        const bool lin = (nthr % hw4) == 0;
        const unsigned kstride = (nthr / hw4) * C * hw4;
        auto off_of = [&](int k) __attribute__((always_inline)) {
            if (lin) return off0 + k * kstride;
            const unsigned e = threadIdx.x + nthr * k;
            const unsigned n = e / hw4, i = e - n * hw4;
            return ((unsigned)(ch.beg[z] + n) * C + c) * hw4 + i;
        };
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < BNO_NV; ++k)
            v[k] = threadIdx.x + nthr * k < cnt ? x4[off_of(k)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < BNO_NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const float inv_n = 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW);
        const float m = bno_block_sum(s, red) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < BNO_NV; ++k) {
            if (threadIdx.x + nthr * k < cnt) {
                const float a0 = v[k].x - m, a1 = v[k].y - m, a2 = v[k].z - m, a3 = v[k].w - m;
                q += fmaf(a0, a0, a1 * a1) + fmaf(a2, a2, a3 * a3);
            }
        }
        const float var = bno_block_sum(q, red) * inv_n;
        const float is = 1.0f / sqrtf(var + eps);
        float sc, sh;
        bnk_affine(m, is, gamma, beta, c, &sc, &sh);
#pragma unroll
        for (int k = 0; k < BNO_NV; ++k) {
            if (threadIdx.x + nthr * k < cnt) {
                float4 o;
                o.x = bn_apply_act(fmaf(v[k].x, sc, sh), act, slope);
                o.y = bn_apply_act(fmaf(v[k].y, sc, sh), act, slope);
                o.z = bn_apply_act(fmaf(v[k].z, sc, sh), act, slope);
                o.w = bn_apply_act(fmaf(v[k].w, sc, sh), act, slope);
                y4[off_of(k)] = o;
            }
        }
        const float momentum = ch.scale[z];
        rm = (1.f - momentum) * rm + momentum * m;
        rv = (1.f - momentum) * rv + momentum * var * ch.aux[z];
        if (threadIdx.x == 0) {
            mean[z * C + c] = m;
            invstd[z * C + c] = is;
        }
    }
    if (threadIdx.x == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
        if (c == 0 && num_batches) *num_batches += ch.n;
    }
}

// KEEP: dz stays in registers next to xhat (both fit); else dy is read again for the second pass
template <bool FROMX, bool KEEP>
__global__ __launch_bounds__(BNO_THREADS) void k_bn_bwd_owned(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
    int accumulate, BnChunks ch, int C, int HW, int act, float slope) {
    __shared__ float red[BNO_THREADS / 64];
    constexpr int NV = BNO_NV;
    const int c = blockIdx.x;
    const unsigned hw4 = HW >> 2, nthr = blockDim.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    float4* o4 = reinterpret_cast<float4*>(dx);
    const float neg = act == BN_ACT_LRELU ? slope : 1.f;
    float gsum = (accumulate && dgamma) ? dgamma[c] : 0.f, bsum = (accumulate && dbeta) ? dbeta[c] : 0.f;
#pragma unroll
    for (int z = 0; z < BNK_MAX_CHUNKS; ++z) {
        if (z >= ch.n) break;
        const unsigned cnt = (unsigned)(ch.end[z] - ch.beg[z]) * hw4;
        const int zc = z * C + c;
        const float m = mean[zc], is = invstd[zc], inv_n = ch.scale[z];
        float sc = 1.f, sh = 0.f;
        if (FROMX) bnk_affine(m, is, gamma, beta, c, &sc, &sh);
        float4 xh[NV], dz[KEEP ? NV : 1];
        const unsigned n0 = threadIdx.x / hw4, i0 = threadIdx.x - n0 * hw4;
        const unsigned off0 = ((unsigned)(ch.beg[z] + n0) * C + c) * hw4 + i0;
        const bool lin = (nthr % hw4) == 0;
        const unsigned kstride = (nthr / hw4) * C * hw4;
        auto off_of = [&](int k) __attribute__((always_inline)) {
            if (lin) return off0 + k * kstride;
            const unsigned e = threadIdx.x + nthr * k;
            const unsigned n = e / hw4, i = e - n * hw4;
            return ((unsigned)(ch.beg[z] + n) * C + c) * hw4 + i;
        };
        float a0 = 0.f, a1 = 0.f;
        auto dz_of = [&](const float4& xv, const float4& dv, unsigned o) __attribute__((always_inline)) {
            float4 f;
            if (FROMX) {
                f.x = fmaf(xv.x, sc, sh) > 0.f ? 1.f : neg;
                f.y = fmaf(xv.y, sc, sh) > 0.f ? 1.f : neg;
                f.z = fmaf(xv.z, sc, sh) > 0.f ? 1.f : neg;
                f.w = fmaf(xv.w, sc, sh) > 0.f ? 1.f : neg;
            } else {
                const float4 yv = y4[o];
                f.x = bn_act_grad_from_output(yv.x, act, slope);
                f.y = bn_act_grad_from_output(yv.y, act, slope);
                f.z = bn_act_grad_from_output(yv.z, act, slope);
                f.w = bn_act_grad_from_output(yv.w, act, slope);
            }
            return make_float4(dv.x * f.x, dv.y * f.y, dv.z * f.z, dv.w * f.w);
        };
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KEEP) dz[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (threadIdx.x + nthr * k < cnt) {
                const unsigned o = off_of(k);
                const float4 xv = x4[o], dv = d4[o];
                const float4 g = dz_of(xv, dv, o);
                // (FROMX: the sign is rebuilt from x in the second pass as well: keep x, not xhat)
                xh[k] = xv;
                if (KEEP) dz[k] = g;
                a0 += (g.x + g.y) + (g.z + g.w);
                a1 += (g.x * ((xv.x - m) * is) + g.y * ((xv.y - m) * is)) +
                      (g.z * ((xv.z - m) * is) + g.w * ((xv.w - m) * is));
            }
        }
        const float s0 = bno_block_sum(a0, red);
        const float s1 = bno_block_sum(a1, red);
        gsum += s1;
        bsum += s0;
        const float gm = (gamma ? gamma[c] : 1.f) * is;
        const float k0 = s0 * inv_n, k1 = s1 * inv_n;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (threadIdx.x + nthr * k < cnt) {
                const float4 xv = xh[k];
                const unsigned o = off_of(k);
                const float4 g = KEEP ? dz[KEEP ? k : 0] : dz_of(xv, d4[o], o);
                float4 r;
                r.x = gm * (g.x - k0 - ((xv.x - m) * is) * k1);
                r.y = gm * (g.y - k0 - ((xv.y - m) * is) * k1);
                r.z = gm * (g.z - k0 - ((xv.z - m) * is) * k1);
                r.w = gm * (g.w - k0 - ((xv.w - m) * is) * k1);
                o4[o] = r;
            }
        }
    }
    if (threadIdx.x == 0) {
        if (dgamma) dgamma[c] = gsum;
        if (dbeta) dbeta[c] = bsum;
    }
}

// ---------------------------------------------------------------------------------------------
static int bn_splits(int N, int C) {
    int s = 1024 / C;
    if (s > 64) s = 64;
    if (s > N) s = N;
    return s < 1 ? 1 : s;
}

size_t bn_batchnorm_ws_bytes_impl(int N, int C) {
    // per chunk (up to BNK_MAX_CHUNKS): two partial arrays [C][S] + two combined vectors [C]
    return (size_t)BNK_MAX_CHUNKS * ((size_t)2 * C * bn_splits(N, C) + 2 * C) * sizeof(float);
}

static int bnk_max_len(const BnChunks& ch) {
    int m = 0;
    for (int z = 0; z < ch.n; ++z) m = ch.end[z] - ch.beg[z] > m ? ch.end[z] - ch.beg[z] : m;
    return m;
}

// mean / var of every chunk: [ch.n][C]
static int bn_launch_stats_chunks(const float* x, float* mean, float* var, BnChunks ch, int C, int HW,
                                  void* ws, hipStream_t st) {
    const int S = bn_splits(bnk_max_len(ch), C);
    float* part = (float*)ws;
    for (int z = 0; z < ch.n; ++z) ch.scale[z] = 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW);
    const dim3 cgrid((C * ch.n + 63) / 64);
    hipLaunchKernelGGL(k_bn_moment_part<1>, dim3(C, S, ch.n), dim3(BNK_THREADS), 0, st, x,
                       (const float*)nullptr, part, ch, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, cgrid, dim3(64), 0, st, part, mean, C, S, ch);
    hipLaunchKernelGGL(k_bn_moment_part<2>, dim3(C, S, ch.n), dim3(BNK_THREADS), 0, st, x,
                       (const float*)mean, part, ch, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, cgrid, dim3(64), 0, st, part, var, C, S, ch);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_stats(const float* x, float* mean, float* var, int N, int C, int HW, void* ws,
                       hipStream_t st) {
    return bn_launch_stats_chunks(x, mean, var, bnk_one_chunk(N), C, HW, ws, st);
}

int bn_launch_bn_finalize(const float* mean, const float* var, float* invstd, float* running_mean,
                          float* running_var, int C, float eps, float momentum, float unbias,
                          hipStream_t st) {
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, st, mean, var, invstd,
                       running_mean, running_var, C, eps, bnk_one_chunk(0, momentum, unbias));
    BN_LAUNCH_CHECK();
    return 0;
}

static int bn_launch_act_fwd_chunks(const float* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, float* y, int N, int C, int HW,
                                    int act, float slope, const BnChunks& ch, hipStream_t st) {
    if ((size_t)N * C * HW >= 0xffffffffull) return BN_E_SHAPE;
    if ((HW & 3) == 0 && (((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0))
        hipLaunchKernelGGL(k_bn_act_fwd<4>, dim3(bnk_flat_blocks((size_t)N * C * (HW >> 2))), dim3(BNK_THREADS),
                           0, st, x, mean, invstd, gamma, beta, y, (unsigned)(N * C), C, HW, act, slope, ch);
    else
        hipLaunchKernelGGL(k_bn_act_fwd<1>, dim3(bnk_flat_blocks((size_t)N * C * HW)), dim3(BNK_THREADS), 0,
                           st, x, mean, invstd, gamma, beta, y, (unsigned)(N * C), C, HW, act, slope, ch);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_act_fwd(const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* y, int N, int C, int HW,
                         int act, float slope, hipStream_t st) {
    return bn_launch_act_fwd_chunks(x, mean, invstd, gamma, beta, y, N, C, HW, act, slope, bnk_one_chunk(N), st);
}

// y == nullptr: the activation's sign from x through (gamma, beta) -- identity / LeakyReLU only
static int bn_launch_bwd_apply_chunks(const float* x, const float* y, const float* dy, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta,
                                      const float* sum_dz, const float* sum_dzx, float* dx, int N, int C,
                                      int HW, int act, float slope, const BnChunks& ch, hipStream_t st) {
    if ((size_t)N * C * HW >= 0xffffffffull) return BN_E_SHAPE;
    if (!y && act != BN_ACT_NONE && act != BN_ACT_LRELU) return BN_E_BADARG;
    const bool vec = (HW & 3) == 0 &&
                     (((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15u) == 0);
    const dim3 grid(bnk_flat_blocks((size_t)N * C * (vec ? HW >> 2 : HW)));
#define BN_APPLY(V, F)                                                                                  \
    hipLaunchKernelGGL((k_bn_bwd_apply<V, F>), grid, dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,   \
                       gamma, beta, sum_dz, sum_dzx, dx, (unsigned)(N * C), C, HW, act, slope, ch)
    if (vec) { if (y) BN_APPLY(4, false); else BN_APPLY(4, true); }
    else { if (y) BN_APPLY(1, false); else BN_APPLY(1, true); }
#undef BN_APPLY
    BN_LAUNCH_CHECK();
    return 0;
}

// backward of all chunks (statistics [ch.n][C]); x / y / dy / dx hold N frames in total
static int bn_launch_act_bwd_chunks(const float* x, const float* y, const float* dy, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float* dx,
                                    float* dgamma, float* dbeta, int accumulate, int batch_stats, int N,
                                    BnChunks ch, int C, int HW, int act, float slope, void* ws,
                                    hipStream_t st) {
    if (!y && act != BN_ACT_NONE && act != BN_ACT_LRELU) return BN_E_BADARG;
    if (bno_fits(ch, C, HW, x, y, dy, dx)) {
        for (int z = 0; z < ch.n; ++z)
            ch.scale[z] = batch_stats ? 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW) : 0.0f;
        const int thr = bno_threads(ch, HW);
#define BNO_BWD(F)                                                                                             \
        hipLaunchKernelGGL((k_bn_bwd_owned<F, true>), dim3(C), dim3(thr), 0, st, x, y, dy, mean, invstd, gamma,  \
                           beta, dx, dgamma, dbeta, accumulate, ch, C, HW, act, slope)
        if (y) BNO_BWD(false); else BNO_BWD(true);
#undef BNO_BWD
        BN_LAUNCH_CHECK();
        return 0;
    }
    const int S = bnk_set_slices(&ch, bn_splits(N, C));
    float* part0 = (float*)ws;
    float* part1 = part0 + (size_t)C * S;
    if (y)
        hipLaunchKernelGGL(k_bn_bwd_part<false>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean,
                           invstd, gamma, beta, part0, part1, ch, C, HW, S, act, slope);
    else
        hipLaunchKernelGGL(k_bn_bwd_part<true>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean,
                           invstd, gamma, beta, part0, part1, ch, C, HW, S, act, slope);
    if ((size_t)N * C * HW >= 0xffffffffull) return BN_E_SHAPE;
    for (int z = 0; z < ch.n; ++z)
        ch.scale[z] = batch_stats ? 1.0f / ((float)(ch.end[z] - ch.beg[z]) * (float)HW) : 0.0f;
    // combine (both sums, parameter gradients) inside the launch that writes dx
    if (y)
        hipLaunchKernelGGL(k_bn_bwd_apply_fin<false>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean,
                           invstd, gamma, beta, (const float*)part0, (const float*)part1, dx, dgamma, dbeta,
                           accumulate, ch, C, HW, S, act, slope);
    else
        hipLaunchKernelGGL(k_bn_bwd_apply_fin<true>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean,
                           invstd, gamma, beta, (const float*)part0, (const float*)part1, dx, dgamma, dbeta,
                           accumulate, ch, C, HW, S, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_act_bwd(const float* x, const float* y, const float* dy, const float* mean,
                         const float* invstd, const float* gamma, float* dx, float* dgamma,
                         float* dbeta, int accumulate, int batch_stats, int N, int C, int HW,
                         int act, float slope, void* ws, hipStream_t st) {
    return bn_launch_act_bwd_chunks(x, y, dy, mean, invstd, gamma, nullptr, dx, dgamma, dbeta, accumulate,
                                    batch_stats, N, bnk_one_chunk(N), C, HW, act, slope, ws, st);
}

// ---- up to BNK_MAX_CHUNKS chunks of one batch in the launches of one (the chunked entry points) ----
static bool bnk_make_chunks(const int* bounds, int n_chunks, BnChunks* ch, int* N) {
    if (n_chunks < 1 || n_chunks > BNK_MAX_CHUNKS) return false;
    *ch = bnk_one_chunk(0);
    ch->n = n_chunks;
    int pos = 0;
    for (int i = 0; i < n_chunks; ++i) {
        if (bounds[2 * i] != pos || bounds[2 * i + 1] <= pos) return false;     // contiguous, in order
        ch->beg[i] = bounds[2 * i];
        ch->end[i] = pos = bounds[2 * i + 1];
    }
    *N = pos;
    return true;
}

int bn_launch_bn_train_fwd_chunks(const float* x, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches,
                                  float* y, float* mean, float* invstd, const int* bounds,
                                  const float* factors, int n_chunks, int C, int HW, float eps, int act,
                                  float slope, void* ws, hipStream_t st) {
    BnChunks ch;
    int N = 0;
    if (!bnk_make_chunks(bounds, n_chunks, &ch, &N)) return BN_E_SHAPE;
    if (bno_fits(ch, C, HW, x, y, nullptr, nullptr)) {
        for (int z = 0; z < ch.n; ++z) {
            const double cnt = (double)(ch.end[z] - ch.beg[z]) * HW;
            ch.scale[z] = factors ? factors[z] : 0.f;
            ch.aux[z] = cnt > 1 ? (float)(cnt / (cnt - 1.0)) : 1.f;
        }
        hipLaunchKernelGGL(k_bn_fwd_owned, dim3(C), dim3(bno_threads(ch, HW)), 0, st, x, gamma, beta, y, mean, invstd,
                           running_mean, running_var, num_batches, ch, C, HW, eps, act, slope);
        BN_LAUNCH_CHECK();
        return 0;
    }
    const int S = bnk_set_slices(&ch, bn_splits(N, C));
    float* part1 = (float*)ws;
    float* part2 = part1 + (size_t)C * S;
    // one pass over x for both moments
    hipLaunchKernelGGL(k_bn_stats_part, dim3(C, S), dim3(BNK_THREADS), 0, st, x, part1, part2, ch, C,
                       HW, S);
    for (int z = 0; z < ch.n; ++z) {
        const double cnt = (double)(ch.end[z] - ch.beg[z]) * HW;
        ch.scale[z] = factors ? factors[z] : 0.f;
        ch.aux[z] = cnt > 1 ? (float)(cnt / (cnt - 1.0)) : 1.f;
    }
    if ((size_t)N * C * HW >= 0xffffffffull) return BN_E_SHAPE;
    // finalize (mean / invstd / running estimates / batch counter) inside the normalising launch
    hipLaunchKernelGGL(k_bn_act_fwd_fin, dim3(C, S), dim3(BNK_THREADS), 0, st, x, (const float*)part1,
                       (const float*)part2, gamma, beta, y, mean, invstd, running_mean, running_var, num_batches,
                       ch, C, HW, S, eps, act, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_bn_act_bwd_chunks(const float* x, const float* y, const float* dy, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* dx,
                                float* dgamma, float* dbeta, int accumulate, const int* bounds,
                                int n_chunks, int C, int HW, int act, float slope, void* ws,
                                hipStream_t st) {
    BnChunks ch;
    int N = 0;
    if (!bnk_make_chunks(bounds, n_chunks, &ch, &N)) return BN_E_SHAPE;
    return bn_launch_act_bwd_chunks(x, y, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta, accumulate, 1, N,
                                    ch, C, HW, act, slope, ws, st);
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
    const BnChunks ch = bnk_one_chunk(N);
    if (center)
        hipLaunchKernelGGL(k_bn_moment_part<2>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, center,
                           part, ch, C, HW, S);
    else
        hipLaunchKernelGGL(k_bn_moment_part<1>, dim3(C, S), dim3(BNK_THREADS), 0, st, x,
                           (const float*)nullptr, part, ch, C, HW, S);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part, sums, C, S, ch);
    BN_LAUNCH_CHECK();
    return 0;
}

// sum_dz[c] = sum dy act'(y), sum_dzx[c] = sum dy act'(y) xhat  (this rank's frames)
int bn_launch_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                            const float* invstd, float* sum_dz, float* sum_dzx, int N, int C,
                            int HW, int act, float slope, void* ws, hipStream_t st) {
    BnChunks ch = bnk_one_chunk(N);
    const int S = bnk_set_slices(&ch, bn_splits(N, C));
    float* part0 = (float*)ws;
    float* part1 = part0 + (size_t)C * S;
    hipLaunchKernelGGL(k_bn_bwd_part<false>, dim3(C, S), dim3(BNK_THREADS), 0, st, x, y, dy, mean, invstd,
                       (const float*)nullptr, (const float*)nullptr, part0, part1, ch, C, HW, S, act, slope);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part0, sum_dz, C, S, ch);
    hipLaunchKernelGGL(k_bn_combine, dim3((C + 63) / 64), dim3(64), 0, st, part1, sum_dzx, C, S, ch);
    BN_LAUNCH_CHECK();
    return 0;
}

// dx from the (global) sums; inv_count = 1 / (frames * pixels the statistics were taken over)
int bn_launch_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                           const float* invstd, const float* gamma, const float* sum_dz,
                           const float* sum_dzx, float* dx, int N, int C, int HW, float inv_count,
                           int act, float slope, hipStream_t st) {
    return bn_launch_bwd_apply_chunks(x, y, dy, mean, invstd, gamma, nullptr, sum_dz, sum_dzx, dx, N, C, HW,
                                      act, slope, bnk_one_chunk(N, inv_count), st);
}
