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
