// Deterministic second pass of the split reductions: out[i] (+)= sum_z part[z][i].
// 64 outputs x 4 (or 16) z-lanes per workgroup: each thread sums every 4th (16th) split with 4
// independent running sums (loads stay in flight), the z-lanes are combined through LDS in fixed
// order.
#pragma once
#include "bn_common.h"

// ab_elems > 0: part is laid out [z][tap][ab] and out is [ab][ntap] (weight-gradient layout)
// row_len > 0: output element i goes to (i / row_len) * row_stride + i % row_len
__global__ __launch_bounds__(1024) void k_sum_partials(const float* __restrict__ part,
                                                       float* __restrict__ out, int total,
                                                       int splits, int accumulate, int ab_elems,
                                                       int ntap, int row_len, int row_stride);

// the same for two reductions of one launch (a weight gradient and its bias gradient): blocks
// [0, blocks_a) do job A, the rest job B (plain layout); `splits` is common
__global__ __launch_bounds__(256) void k_sum_partials_pair(
    const float* __restrict__ part_a, float* __restrict__ out_a, int total_a, int ab_elems, int ntap,
    const float* __restrict__ part_b, float* __restrict__ out_b, int total_b, int splits,
    int accumulate, int blocks_a);

static inline int bn_launch_sum_partials_pair(const float* part_a, float* out_a, int total_a,
                                              int ab_elems, int ntap, const float* part_b,
                                              float* out_b, int total_b, int splits, int accumulate,
                                              hipStream_t st) {
    const int blocks_a = (total_a + 63) / 64, blocks_b = (total_b + 63) / 64;
    hipLaunchKernelGGL(k_sum_partials_pair, dim3(blocks_a + blocks_b), dim3(256), 0, st, part_a, out_a,
                       total_a, ab_elems, ntap, part_b, out_b, total_b, splits, accumulate, blocks_a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline int bn_launch_sum_partials(const float* part, float* out, int total, int splits,
                                         int accumulate, int ab_elems, int ntap, hipStream_t st,
                                         int row_len = 0, int row_stride = 0) {
    // many partials of few outputs (the single-channel edge layers: 768 x 800): 16 z-lanes
    const int threads = (splits >= 128 && total <= 16384) ? 1024 : 256;
    hipLaunchKernelGGL(k_sum_partials, dim3((total + 63) / 64), dim3(threads), 0, st, part, out, total,
                       splits, accumulate, ab_elems, ntap, row_len, row_stride);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
