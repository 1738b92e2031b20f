// Dense latent projections on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate,
// bit-identical to an fmaf chain).  One strided kernel serves nn.Linear forward, its data
// gradient and its weight gradient:   C[i,j] (+)= epi( sum_k A(i,k) * B(k,j) ).
//
// Lane mapping (cdna_hip_programming.md section 3): lane l feeds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; accumulator register t of lane l is C[(t&3)+8*(t>>2)+4*(l>>5)][l&31].
// The latent GEMMs are skinny (N=12..64 or K=12..64), so a workgroup owns one 32x32 output tile
// and its SPLITK waves split the reduction dimension, combined in fixed order through LDS.
#include <string.h>
#include "bn_common.h"
#include "bn_launch.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));


// One 32x32 tile of C at tile coordinates (bx, by), reduction slice bz of a.kslice, shared by `nw`
// waves (wave `ws` of them; combined through `red` = nw * 1024 floats in the fixed order of the wave
// index).  Every wave of the tile must call it; the caller's block may hold other tiles.
__device__ __forceinline__ void gemm_tile(const GemmArgs& a, const int bx, const int by, const int bz,
                                          const int nw, const int ws, float* red,
                                          const bool live = true) {
    const int lane = threadIdx.x & 63;
    const int i0 = by * 32, j0 = bx * 32;
    const int li = lane & 31, lk = lane >> 5;

    // this tile's slice (slices of a.kslice, a multiple of 16) and this wave's part of it (even
    // length so k pairs stay aligned)
    const int zbeg = bz * a.kslice;
    const int zend = min(a.K, zbeg + a.kslice);
    int kper = (zend - zbeg + nw - 1) / nw;
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
    // These skinny products are bound by memory LATENCY: a trip requests 32 k of both operands
    // (8 independent 16-byte loads) before its first MFMA, and the trips of a wave are few.
    // (a reduction that ends on a multiple of 4, K = 12: the second lane half of the last group of 8 is masked)
    const bool vec = (a.sak == 1) && (a.sbk == 1) && ((kbeg & 7) == 0) && ((kend & 3) == 0) && kend - kbeg >= 4 &&
                     ((a.sai & 3) == 0) && ((a.sbj & 3) == 0) &&
                     ((((uintptr_t)a.A) | ((uintptr_t)a.B)) & 15u) == 0;
    if (vec) {
        for (int k0 = kbeg; k0 < kend; k0 += 32) {      // 4 float4 per operand per trip
            float4 av4[4], bv4[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int k = min(k0 + 8 * h + 4 * lk, kend - 4);
                av4[h] = *reinterpret_cast<const float4*>(ap + k);
                bv4[h] = *reinterpret_cast<const float4*>(bp + k);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const bool ok = (k0 + 8 * h + 4 * lk) < kend;
                if (k0 + 8 * h >= kend) break;                    // wave-uniform
                const float ax[4] = {av4[h].x, av4[h].y, av4[h].z, av4[h].w};
                const float bx4[4] = {bv4[h].x, bv4[h].y, bv4[h].z, bv4[h].w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && ok) ? ax[u] : 0.f,
                                                               (b_ok && ok) ? bx4[u] : 0.f, acc, 0,
                                                               0, 0);
            }
        }
    }
    // A alone unit-stride (the data gradient of a wide layer: A = dy (M x 2048) in rows of 8 KB,
    // B = the 12-column weight): A by 16-byte loads as above -- a 4-byte gather along a row-strided
    // operand touches 64 cache lines per instruction, and the CU's line rate, not bytes, sets the
    // pace -- B by four 4-byte loads from its small, cache-resident matrix
    const bool veca = !vec && (a.sak == 1) && ((kbeg & 7) == 0) && ((kend & 3) == 0) && kend - kbeg >= 4 &&
                      ((a.sai & 3) == 0) && (((uintptr_t)a.A) & 15u) == 0;
    if (veca) {
        for (int k0 = kbeg; k0 < kend; k0 += 32) {
            float4 av4[4];
            float bs[4][4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int k = min(k0 + 8 * h + 4 * lk, kend - 4);
                av4[h] = *reinterpret_cast<const float4*>(ap + k);
#pragma unroll
                for (int u = 0; u < 4; ++u) bs[h][u] = bp[(long)(k + u) * a.sbk];
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const bool ok = (k0 + 8 * h + 4 * lk) < kend;
                if (k0 + 8 * h >= kend) break;                    // wave-uniform
                const float ax[4] = {av4[h].x, av4[h].y, av4[h].z, av4[h].w};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && ok) ? ax[u] : 0.f,
                                                               (b_ok && ok) ? bs[h][u] : 0.f, acc, 0,
                                                               0, 0);
            }
        }
    }
    // 16 reduction steps per trip: 32 independent (clamped, unconditional) loads are in flight
    // before the first MFMA needs one
    const int klast = max(kend - 1, kbeg);
    for (int k0 = kbeg + lk; k0 < ((vec || veca) ? 0 : kend + lk); k0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = min(k0 + 2 * u, klast);
            av[u] = ap[(long)k * a.sak];
            bv[u] = bp[(long)k * a.sbk];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool ok = (k0 + 2 * u) < kend;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((a_ok && ok) ? av[u] : 0.f,
                                                       (b_ok && ok) ? bv[u] : 0.f, acc, 0, 0, 0);
        }
    }

    if (nw > 1) {
#pragma unroll
        for (int t = 0; t < 16; ++t) red[(ws * 16 + t) * 64 + lane] = acc[t];
        __syncthreads();      // (the LAST barrier of the workgroup: waves may leave from here on)
        if (ws != 0) return;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float s = red[t * 64 + lane];
            for (int w2 = 1; w2 < nw; ++w2) s += red[(w2 * 16 + t) * 64 + lane];
            acc[t] = s;
        }
    }

    const int j = j0 + (lane & 31);
    if (j >= a.N || !live) return;
    if (a.part) {       // cross-workgroup split: raw partial tile, combined by k_gemm_combine
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
            if (i < a.M) a.part[((size_t)bz * a.M + i) * a.N + j] = acc[t];
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

template <int SPLITK>
__global__ __launch_bounds__(64 * SPLITK) void k_gemm_mfma(GemmArgs a) {
    __shared__ float red[SPLITK > 1 ? SPLITK * 16 * 64 : 1];
    gemm_tile(a, blockIdx.x, blockIdx.y, blockIdx.z, SPLITK, threadIdx.x >> 6, red);
}

// ---------------------------------------------------------------------------------------------
// nn.Linear: the jobs of a direction side by side in ONE grid.  A workgroup of four waves serves
// 4 / nw units (a unit = one 32x32 tile x one reduction slice), nw waves sharing a unit's reduction
// (nw = 4 for the deep products, 1 for the 12-deep ones, whose units then share a workgroup); the
// backward pass runs its three jobs -- data gradient, weight gradient, bias column sums -- in the same
// launch (they read the same two operands and depend on nothing else).  A 2048-deep product onto a
// handful of tiles is still split over workgroups (its row-strided operand reads are bound by the
// CU's cache-line rate: 8 workgroups took 25 us, 128 take 6) and combined in fixed order by
// k_gemm_combine.  The backward pass used to be three or four launches of 5-11 us for 12.6 MFLOP.
// ---------------------------------------------------------------------------------------------
__global__ void k_gemm_combine(GemmArgs a, int slices);
#define GJ_WAVES 4
struct GemmJob { GemmArgs a; int tiles_x, tiles, slices, nw, blocks; };
struct LinearJobs {
    GemmJob g[2];
    const float* dy; float* db; int M, N, accumulate, db_blocks;     // bias column sums (db nullable)
};

__global__ __launch_bounds__(64 * GJ_WAVES) void k_linear_jobs(LinearJobs p) {
    __shared__ float red[GJ_WAVES * 16 * 64];
    int b = blockIdx.x;
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const GemmJob& job = p.g[q];
        if (b < job.blocks) {
            const int upb = GJ_WAVES / job.nw;
            // the waves of a unit past the last one (two units per workgroup, odd unit count) stay
            // alive through gemm_tile's barrier -- they redo the last unit and store nothing --
            // instead of returning in front of it (a barrier that other waves of the
            // workgroup never reach is undefined in HIP, whatever gfx950's s_barrier does)
            const int units = job.tiles * job.slices;
            const int unit_w = b * upb + wave / job.nw;
            const bool live = unit_w < units;
            const int unit = live ? unit_w : units - 1;
            const int tile = unit / job.slices, slice = unit - tile * job.slices;
            gemm_tile(job.a, tile % job.tiles_x, tile / job.tiles_x, slice, job.nw, wave % job.nw,
                      red + (wave / job.nw) * job.nw * 1024, live);
            return;
        }
        b -= job.blocks;
    }
    // db[n] (+)= sum_m dy[m][n]: 64 columns per workgroup, the rows dealt to the four waves and added
    // up in the fixed order of the wave index
    const int col = b * 64 + (threadIdx.x & 63);
    float acc = 0.f;
    if (col < p.N) {
        // sixteen loads in flight, added in row order (one load per trip was a chain of M / 4
        // memory latencies: 20 us for 256 rows)
        for (int m0 = wave; m0 < p.M; m0 += 16 * GJ_WAVES) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int m = m0 + u * GJ_WAVES;
                v[u] = p.dy[(size_t)(m < p.M ? m : wave) * p.N + col];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (m0 + u * GJ_WAVES < p.M) acc += v[u];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (wave == 0 && col < p.N) {
        float v = red[threadIdx.x];
        for (int w2 = 1; w2 < GJ_WAVES; ++w2) v += red[w2 * 64 + threadIdx.x];
        p.db[col] = p.accumulate ? p.db[col] + v : v;
    }
}

// a long reduction feeding only a handful of output tiles (the 2048 -> n_latents projections:
// 8 tiles) is split over workgroups, so that it does not run on 8 of the 256 CUs
static int gemm_slices(int M, int N, int K) {
    const int tiles = ((N + 31) / 32) * ((M + 31) / 32);
    if (K < 1024 || tiles > 32) return 1;
    int s = K / 128;
    // (round 6) a very deep reduction -- the 32768-wide FF layer of the max-pooling test architecture: 8 tiles x 16
    // slices were 128 workgroups walking 2048 elements each, 33 us for 33.5 MB -- takes slices of >= 256 elements
    // until the units fill the chip twice; K = 2048 (the default architecture) keeps its 16
    int cap = 16;
    while (cap < 128 && tiles * cap < 512 && K / (2 * cap) >= 256) cap *= 2;
    if (s > cap) s = cap;
    return s < 1 ? 1 : s;
}

static void gemm_job(GemmJob* j, const GemmArgs& a, float* part) {
    j->a = a;
    j->tiles_x = (a.N + 31) / 32;
    j->tiles = j->tiles_x * ((a.M + 31) / 32);
    j->slices = part ? gemm_slices(a.M, a.N, a.K) : 1;
    j->a.part = j->slices > 1 ? part : nullptr;
    j->a.kslice = ((a.K + j->slices - 1) / j->slices + 15) & ~15;
    const int kw = j->a.kslice / 32;        // >= 32 reduction steps per wave
    j->nw = kw >= 4 ? 4 : kw >= 2 ? 2 : 1;
    const int upb = GJ_WAVES / j->nw, units = j->tiles * j->slices;
    j->blocks = (units + upb - 1) / upb;
}

size_t bn_linear_jobs_ws_bytes(int M, int N, int K) {
    const int s = gemm_slices(M, N, K);
    return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}

// up to two products and the column sums of `dy` (M x N, nullable with db) in one launch; a product
// that is split over workgroups takes its partial tiles through `ws` (g0 first) and is finished by
// k_gemm_combine
int bn_launch_linear_jobs(const GemmArgs* g0, const GemmArgs* g1, const float* dy, float* db, int M,
                          int N, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    LinearJobs p;
    memset(&p, 0, sizeof(p));
    int blocks = 0;
    const GemmArgs* gs[2] = {g0, g1};
    char* wp = (char*)ws;
    for (int q = 0; q < 2; ++q)
        if (gs[q]) {
            const size_t need = bn_linear_jobs_ws_bytes(gs[q]->M, gs[q]->N, gs[q]->K);
            float* part = nullptr;
            if (need && wp && ws_bytes >= need) { part = (float*)wp; wp += need; ws_bytes -= need; }
            gemm_job(&p.g[q], *gs[q], part);
            blocks += p.g[q].blocks;
        }
    if (db) {
        p.dy = dy; p.db = db; p.M = M; p.N = N; p.accumulate = accumulate;
        p.db_blocks = (N + 63) / 64;
        blocks += p.db_blocks;
    }
    if (blocks == 0) return 0;
    BN_LAUNCH_MAIN(k_linear_jobs, dim3(blocks), dim3(64 * GJ_WAVES), 0, st, p);
    for (int q = 0; q < 2; ++q)
        if (gs[q] && p.g[q].slices > 1)
            hipLaunchKernelGGL(k_gemm_combine, dim3((gs[q]->M * gs[q]->N + 255) / 256), dim3(256), 0, st,
                               p.g[q].a, p.g[q].slices);
    BN_LAUNCH_CHECK();
    return 0;
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
    // sixteen loads in flight, added in slice order (one load per trip is a chain of `slices` memory latencies:
    // 17 us for the 64 slices of a 32768-deep product)
    const size_t zs = (size_t)a.M * a.N;
    const float* p0 = a.part + (size_t)i * a.N + j;
    for (int z0 = 0; z0 < slices; z0 += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = p0[(size_t)(z0 + u < slices ? z0 + u : z0) * zs];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (z0 + u < slices) v += t[u];
    }
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
