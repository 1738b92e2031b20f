// Dense latent projections on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate,
// bit-identical to an fmaf chain).  One strided kernel serves nn.Linear forward, its data
// gradient and its weight gradient:   C[i,j] (+)= epi( sum_k A(i,k) * B(k,j) ).
//
// Lane mapping (cdna_hip_programming.md section 3): lane l feeds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator register t of lane l is C[(t&3)+8*(t>>2)+4*(l>>5)][l&31].
// The latent GEMMs are skinny (N=12..64 or K=12..64), so a workgroup owns one 32x32 output tile
// and its SPLITK waves split the reduction dimension, combined in fixed order through LDS.
#include "bn_common.h"
#include "bn_launch.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));


template <int SPLITK>
__global__ __launch_bounds__(64 * SPLITK) void k_gemm_mfma(GemmArgs a) {
    __shared__ float red[SPLITK > 1 ? SPLITK * 16 * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int ws = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int li = lane & 31, lk = lane >> 5;

    // this workgroup's slice (gridDim.z slices of a.kslice, a multiple of 16) and this wave's
    // part of it (even length so k pairs stay aligned)
    const int zbeg = blockIdx.z * a.kslice;
    const int zend = min(a.K, zbeg + a.kslice);
    int kper = (zend - zbeg + SPLITK - 1) / SPLITK;
    kper = (kper + 7) & ~7;
    const int kbeg = zbeg + ws * kper;
    const int kend = min(zend, kbeg + kper);

    const int ia = i0 + li, jb = j0 + li;
    const bool a_ok = ia < a.M, b_ok = jb < a.N;
    const float* ap = a.A + (long)(a_ok ? ia : 0) * a.sai;
    const float* bp = a.B + (long)(b_ok ? jb : 0) * a.sbj;

    floatx16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;

    // Unit-stride operands: each lane loads 16 bytes (4 consecutive k of ITS row / column) per
    // operand and feeds 4 MFMA steps from them -- step u pairs k = base+u (lane half 0) with
    // k = base+4+u (lane half 1).  A per-lane 4-byte gather would touch 64 cache lines per
    // instruction for 2 useful floats each; this touches them once per 8 k.
    const bool vec = (a.sak == 1) && (a.sbk == 1) && ((kbeg & 7) == 0) && ((kend & 7) == 0) &&
                     ((a.sai & 3) == 0) && ((a.sbj & 3) == 0) &&
                     ((((uintptr_t)a.A) | ((uintptr_t)a.B)) & 15u) == 0;
    if (vec) {
        for (int k0 = kbeg; k0 < kend; k0 += 16) {      // 2 float4 per operand per trip
            float4 av4[2], bv4[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = min(k0 + 8 * h + 4 * lk, kend - 4);
                av4[h] = *reinterpret_cast<const float4*>(ap + k);
                bv4[h] = *reinterpret_cast<const float4*>(bp + k);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = (k0 + 8 * h) < kend;
                const float ax[4] = {av4[h].x, av4[h].y, av4[h].z, av4[h].w};
                const float bx[4] = {bv4[h].x, bv4[h].y, bv4[h].z, bv4[h].w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && ok) ? ax[u] : 0.f,
                                                               (b_ok && ok) ? bx[u] : 0.f, acc, 0,
                                                               0, 0);
            }
        }
    }
    // 8 reduction steps per trip: 16 independent (clamped, unconditional) loads are in flight
    // before the first MFMA needs one -- these skinny GEMMs are latency-, not bandwidth-bound
    const int klast = max(kend - 1, kbeg);
    for (int k0 = kbeg + lk; k0 < (vec ? 0 : kend + lk); k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + 2 * u, klast);
            av[u] = ap[(long)k * a.sak];
            bv[u] = bp[(long)k * a.sbk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = (k0 + 2 * u) < kend;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && ok) ? av[u] : 0.f,
                                                       (b_ok && ok) ? bv[u] : 0.f, acc, 0, 0, 0);
        }
    }

    if (SPLITK > 1) {
#pragma unroll
        for (int t = 0; t < 16; ++t) red[(ws * 16 + t) * 64 + lane] = acc[t];
        __syncthreads();
        if (ws != 0) return;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float s = red[t * 64 + lane];
            for (int w2 = 1; w2 < SPLITK; ++w2) s += red[(w2 * 16 + t) * 64 + lane];
            acc[t] = s;
        }
    }

    const int j = j0 + (lane & 31);
    if (j >= a.N) return;
    if (a.part) {       // cross-workgroup split: raw partial tile, combined by k_gemm_combine
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
            if (i < a.M) a.part[((size_t)blockIdx.z * a.M + i) * a.N + j] = acc[t];
        }
        return;
    }
    const float bj = a.bias_j ? a.bias_j[j] : 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = i0 + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
        if (i >= a.M) continue;
        const long off = (long)i * a.sci + (long)j * a.scj;
        float v = acc[t] + bj;
        if (a.dact_src) v *= bn_act_grad_from_output(a.dact_src[off], a.dact, a.slope);
        a.C[off] = a.accumulate ? a.C[off] + v : v;
    }
}

// C[i,j] (+)= epi( sum_z part[z][i][j] )   (fixed order)
__global__ __launch_bounds__(256) void k_gemm_combine(GemmArgs a, int slices) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.M * a.N) return;
    const int i = idx / a.N, j = idx - i * a.N;
    float v = 0.f;
    for (int z = 0; z < slices; ++z) v += a.part[((size_t)z * a.M + i) * a.N + j];
    if (a.bias_j) v += a.bias_j[j];
    const long off = (long)i * a.sci + (long)j * a.scj;
    if (a.dact_src) v *= bn_act_grad_from_output(a.dact_src[off], a.dact, a.slope);
    a.C[off] = a.accumulate ? a.C[off] + v : v;
}

// db[n] (+)= sum_m dy[m,n]   (dy row-major M x N); one wave per column
__global__ __launch_bounds__(64) void k_col_sum(const float* __restrict__ dy,
                                                float* __restrict__ db, int M, int N,
                                                int accumulate) {
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int m = threadIdx.x; m < M; m += 64) acc += dy[(size_t)m * N + n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) db[n] = accumulate ? db[n] + acc : acc;
}

// a long reduction feeding only a handful of output tiles (the 2048 -> n_latents projections:
// 7 tiles) is split over workgroups as well, so that it does not run on 7 of the 256 CUs
static int gemm_slices(int M, int N, int K) {
    const int tiles = ((N + 31) / 32) * ((M + 31) / 32);
    if (K < 1024 || tiles > 32) return 1;
    int s = K / 128;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

size_t bn_gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_slices(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

int bn_launch_gemm(const GemmArgs& a0, hipStream_t st, void* ws, size_t ws_bytes) {
    GemmArgs a = a0;
    int slices = gemm_slices(a.M, a.N, a.K);
    if (slices > 1 && (!ws || ws_bytes < (size_t)slices * a.M * a.N * sizeof(float))) slices = 1;
    a.kslice = ((a.K + slices - 1) / slices + 15) & ~15;
    a.part = slices > 1 ? (float*)ws : nullptr;
    dim3 grid((a.N + 31) / 32, (a.M + 31) / 32, slices);
    if (a.K >= 512) {
        hipLaunchKernelGGL(k_gemm_mfma<8>, grid, dim3(512), 0, st, a);
    } else if (a.K >= 128) {
        // e.g. the FF weight gradients (reduction over the 256 frames of a batch): four waves
        // share the reduction instead of one wave walking it alone (20 -> 7 us)
        hipLaunchKernelGGL(k_gemm_mfma<4>, grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_gemm_mfma<1>, grid, dim3(64), 0, st, a);
    }
    if (slices > 1)
        hipLaunchKernelGGL(k_gemm_combine, dim3((a.M * a.N + 255) / 256), dim3(256), 0, st, a,
                           slices);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_col_sum(const float* dy, float* db, int M, int N, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(k_col_sum, dim3(N), dim3(64), 0, st, dy, db, M, N, accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}
