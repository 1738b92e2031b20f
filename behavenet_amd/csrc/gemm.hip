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

// ---------------------------------------------------------------------------------------------
// The same product for LARGE operands (the im2col detours of conv_pad.hip: thousands of rows, K in
// the hundreds or thousands): 64 x 128 tile per workgroup of four waves (a wave owns 32 x 64 = two
// accumulators), 32-deep stages through LDS (rows of 33 words: the 32 rows x 2 k of an MFMA operand
// read hit 64 distinct banks), the next stage's 16-byte global loads in flight during the matrix
// work.  k_gemm_mfma above reads every operand element from global memory once per 32x32 tile:
// 8 FLOP per byte of L2 traffic, ~40 TFLOP/s; this one 43 FLOP per byte.
//   TA = 0: A(i, k) is k-contiguous (sak == 1)    TA = 1: i-contiguous (sai == 1)
//   TB = 0: B(k, j) is k-contiguous (sbk == 1)    TB = 1: j-contiguous (sbj == 1)
// The contiguous extents must be multiples of 8 floats and 16-byte aligned (checked by the launcher).
// ---------------------------------------------------------------------------------------------
#define GT_M 64
#define GT_N 128
#define GT_K 32
#define GT_LD 33
template <int TA, int TB>
__global__ __launch_bounds__(256) void k_gemm_tiled(GemmArgs a) {
    __shared__ float As[GT_M * GT_LD];
    __shared__ float Bs[GT_N * GT_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int i0 = blockIdx.y * GT_M, j0 = blockIdx.x * GT_N;
    const int kbeg = blockIdx.z * a.kslice;
    const int kend = min(a.K, kbeg + a.kslice);

    // staging: 8 floats of the A tile and 2 x 8 of the B tile per thread, two 16-byte loads each
    //   TA = 0: row ar = tid / 4, k = 8 (tid % 4) ..       TA = 1: k row tid / 8, rows 8 (tid % 8) ..
    //   TB = 0: col br = tid / 4 (+ 64 h), k = 8 (tid % 4) ..  TB = 1: k row tid / 8, cols 8 (tid % 8) .. (+ 64 h)
    const int ar = TA ? (tid & 7) * 8 : tid >> 2, ak = TA ? tid >> 3 : (tid & 3) * 8;
    const int br = TB ? (tid & 7) * 8 : tid >> 2, bk = TB ? tid >> 3 : (tid & 3) * 8;
    float ra[8], rb[2][8];
    auto fetch = [&](const int k0) __attribute__((always_inline)) {
        {
            const int i = i0 + ar, k = k0 + ak;
            const bool ok = TA ? (k < kend && i < a.M) : (i < a.M && k < kend);
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (ok) {
                const float* p = a.A + (long)i * a.sai + (long)k * a.sak;
                v0 = *reinterpret_cast<const float4*>(p);
                v1 = *reinterpret_cast<const float4*>(p + 4);
            }
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v0.z; ra[3] = v0.w;
            ra[4] = v1.x; ra[5] = v1.y; ra[6] = v1.z; ra[7] = v1.w;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + 64 * h + br, k = k0 + bk;
            const bool ok = j < a.N && k < kend;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (ok) {
                const float* p = a.B + (long)k * a.sbk + (long)j * a.sbj;
                v0 = *reinterpret_cast<const float4*>(p);
                v1 = *reinterpret_cast<const float4*>(p + 4);
            }
            rb[h][0] = v0.x; rb[h][1] = v0.y; rb[h][2] = v0.z; rb[h][3] = v0.w;
            rb[h][4] = v1.x; rb[h][5] = v1.y; rb[h][6] = v1.z; rb[h][7] = v1.w;
        }
    };

    floatx16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[h][t] = 0.f;
    // wave (wv >> 1, wv & 1): rows 32 (wv >> 1) .., columns 32 (wv & 1) .. of both 64-column halves
    const float* ap = As + ((wv >> 1) * 32 + li) * GT_LD + lk;
    const float* bp = Bs + ((wv & 1) * 32 + li) * GT_LD + lk;

    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GT_K) {
        __syncthreads();                       // everyone is done reading the previous stage
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (TA) As[(ar + e) * GT_LD + ak] = ra[e];
            else    As[ar * GT_LD + ak + e] = ra[e];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (TB) Bs[(64 * h + br + e) * GT_LD + bk] = rb[h][e];
                else    Bs[(64 * h + br) * GT_LD + bk + e] = rb[h][e];
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < GT_K / 2; ++t) {
            const float av = ap[2 * t];
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[64 * h * GT_LD + 2 * t], acc[h], 0, 0, 0);
            if (t == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (k0 + GT_K < kend) fetch(k0 + GT_K);      // rides in this MFMA stream
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = j0 + 64 * h + (wv & 1) * 32 + li;
        if (j >= a.N) continue;
        const float bj = (a.bias_j && !a.part) ? a.bias_j[j] : 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + (wv >> 1) * 32 + (t & 3) + 8 * (t >> 2) + 4 * lk;
            if (i >= a.M) continue;
            if (a.part) {
                a.part[((size_t)blockIdx.z * a.M + i) * a.N + j] = acc[h][t];
                continue;
            }
            const long off = (long)i * a.sci + (long)j * a.scj;
            float v = acc[h][t] + bj;
            if (a.dact_src) v *= bn_act_grad_from_output(a.dact_src[off], a.dact, a.slope);
            a.C[off] = a.accumulate ? a.C[off] + v : v;
        }
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

// split of the reduction over workgroups: as many slices as fill the chip twice, 128-deep at least
static int gemm_tiled_slices(int M, int N, int K) {
    const int tiles = ((N + GT_N - 1) / GT_N) * ((M + GT_M - 1) / GT_M);
    if (tiles >= 256 || K < 512) return 1;
    int s = 512 / tiles;
    if (s > K / 128) s = K / 128;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

size_t bn_gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_slices(M, N, K);
    size_t need = s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
    if (M >= 64 && N >= 64 && K >= 64) {                 // (the tiled kernel's split, if it is chosen)
        const size_t t = (size_t)gemm_tiled_slices(M, N, K) * M * N * sizeof(float);
        if (gemm_tiled_slices(M, N, K) > 1 && t > need) need = t;
    }
    return need;
}

// the tiled kernel: large products whose contiguous extents are 8-float multiples and aligned
static bool gemm_tiled_ok(const GemmArgs& a, int* ta, int* tb) {
    if (a.M < 64 || a.N < 64 || a.K < 64 || (size_t)a.M * a.N < 64 * 1024) return false;
    if ((((uintptr_t)a.A) | ((uintptr_t)a.B)) & 15u) return false;
    if (a.sak == 1) { *ta = 0; if ((a.K & 7) || (a.sai & 3)) return false; }
    else if (a.sai == 1) { *ta = 1; if ((a.M & 7) || (a.sak & 3)) return false; }
    else return false;
    if (a.sbk == 1) { *tb = 0; if ((a.K & 7) || (a.sbj & 3)) return false; }
    else if (a.sbj == 1) { *tb = 1; if ((a.N & 7) || (a.sbk & 3)) return false; }
    else return false;
    return true;
}
int bn_launch_gemm(const GemmArgs& a0, hipStream_t st, void* ws, size_t ws_bytes) {
    GemmArgs a = a0;
    int ta = 0, tb = 0;
    if (gemm_tiled_ok(a, &ta, &tb)) {
        int slices = gemm_tiled_slices(a.M, a.N, a.K);
        if (slices > 1 && (!ws || ws_bytes < (size_t)slices * a.M * a.N * sizeof(float))) slices = 1;
        a.kslice = ((a.K + slices - 1) / slices + GT_K - 1) / GT_K * GT_K;
        a.part = slices > 1 ? (float*)ws : nullptr;
        dim3 grid((a.N + GT_N - 1) / GT_N, (a.M + GT_M - 1) / GT_M, slices);
        if (grid.y < 65536) {
            if (ta == 0 && tb == 0) hipLaunchKernelGGL((k_gemm_tiled<0, 0>), grid, dim3(256), 0, st, a);
            else if (ta == 0) hipLaunchKernelGGL((k_gemm_tiled<0, 1>), grid, dim3(256), 0, st, a);
            else if (tb == 0) hipLaunchKernelGGL((k_gemm_tiled<1, 0>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_gemm_tiled<1, 1>), grid, dim3(256), 0, st, a);
            if (slices > 1)
                hipLaunchKernelGGL(k_gemm_combine, dim3((a.M * a.N + 255) / 256), dim3(256), 0, st, a, slices);
            BN_LAUNCH_CHECK();
            return 0;
        }
    }
    a = a0;
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
