// Layers whose stride equals the kernel size (the default arch's last encoder / first decoder
// layer: k5 s5).  The 5x5 windows do not overlap, so every input element is used by exactly one
// (output pixel, tap): the "gather" is a pure permutation and the op is a plain GEMM.  Operands
// are short (K = 512 channels or N*4 pixels) and L2-resident, so these kernels feed
// v_mfma_f32_32x32x2_f32 straight from global memory (per-lane gathers, no LDS staging):
//
//   up    (convT fwd / conv bwd-data):  D[(c,tap)][pixel] = sum_k W[k][c][tap] * small[n][k][p][q]
//                                       scattered to out[n][c][5p+r-pt][5q+s-pl]
//   wgrad (both weight gradients):      D[a][(b,tap)] = sum_pixel small[n][a][p][q] *
//                                                       big[n][b][5p+r-pt][5q+s-pl]
//
// Out-of-range reads (padding, tails) go through raw buffer descriptors and return 0.
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define S5_THREADS 256
#define S5_OOB 0x7fffffff
#define S5_U 8                 // reduction steps (MFMA pairs) per load batch

__device__ __forceinline__ float ldbuf(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// ---------------------------------------------------------------------------------------------
// up: rows = (c, tap) in [0, Cb*25), columns = small pixels in [0, N*Hs*Ws), K = Cs
// workgroup: 4 waves, each 64 rows x 32 columns; grid (row tiles of 64, column tiles of 128)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(S5_THREADS) void k_up_s5(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, int act, int dact,
    float slope) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int rows = g.Cb * 25, HWs = g.Hs * g.Ws, npix = g.N * HWs;
    const int row0 = blockIdx.x * 64;
    const int col = (blockIdx.y * 4 + wv) * 32 + li;

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * rows * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)npix * g.Cs * 4), 0x00020000);

    // A(i, k) = w[k*rows + row];  B(k, j) = small[(n*Cs + k)*HWs + pq]
    int aoff[2];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
        const int r = row0 + mr * 32 + li;
        aoff[mr] = r < rows ? (kk * rows + r) * 4 : S5_OOB;
    }
    const int n = col / HWs, pq = col - n * HWs;
    int boff = col < npix ? ((n * g.Cs + kk) * HWs + pq) * 4 : S5_OOB;
    const int astep = 2 * rows * 4, bstep = 2 * HWs * 4;

    floatx16 acc[2];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mr][e] = 0.f;

    // batches of S5_U reduction steps: all 3*S5_U loads are issued (L2 round trip ~600-900
    // cycles) before the first MFMA of the batch consumes one
    const int ksteps = (g.Cs + 1) / 2;
    for (int k0 = 0; k0 < ksteps; k0 += S5_U) {
        float bq[S5_U], aq[S5_U][2];
#pragma unroll
        for (int u = 0; u < S5_U; ++u) {
            const int ks = k0 + u;
            const bool kok = ks < ksteps && 2 * ks + kk < g.Cs;
            bq[u] = ldbuf(rx, (kok && boff != S5_OOB) ? boff + ks * bstep : S5_OOB);
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
                aq[u][mr] =
                    ldbuf(rw, (kok && aoff[mr] != S5_OOB) ? aoff[mr] + ks * astep : S5_OOB);
        }
#pragma unroll
        for (int u = 0; u < S5_U; ++u)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
                acc[mr] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][mr], bq[u], acc[mr], 0, 0, 0);
    }

    // scatter: row -> (c, r, s), column -> (n, p, q) -> out[n][c][5p+r-pt][5q+s-pl]
    if (col >= npix) return;
    const int p = pq / g.Ws, q = pq - p * g.Ws;
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = row0 + mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (r >= rows) continue;
            const int c = r / 25, tap = r - c * 25;
            const int tr = tap / 5, ts = tap - tr * 5;
            const int h = 5 * p + tr - g.pt, x = 5 * q + ts - g.pl;
            if (h < 0 || h >= g.Hb || x < 0 || x >= g.Wb) continue;
            const size_t idx = (((size_t)n * g.Cb + c) * g.Hb + h) * g.Wb + x;
            float v = acc[mr][e] + (bias ? bias[c] : 0.f);
            v = bn_apply_act(v, act, slope);
            if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
            out[idx] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad: rows = a in [0, Cs), columns = (b, tap) in [0, Cb*25), K = N*Hs*Ws pixels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(S5_THREADS) void k_wgrad_s5(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ dw,
    BnGeom g, int accumulate) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int cols = g.Cb * 25, HWs = g.Hs * g.Ws, npix = g.N * HWs, HWb = g.Hb * g.Wb;
    const int row0 = blockIdx.x * 64;
    const int col = (blockIdx.y * 4 + wv) * 32 + li;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)npix * g.Cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HWb * 4), 0x00020000);

    // column -> (b, r, s): lane-constant part of the big-side address
    const bool cok = col < cols;
    const int b = cok ? col / 25 : 0;
    const int tap = col - b * 25;
    const int tr = tap / 5, ts = tap - tr * 5;
    const int a_lane[2] = {row0 + li, row0 + 32 + li};

    floatx16 acc[2];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mr][e] = 0.f;

    const int ksteps = (npix + 1) / 2;
    // pixel (n, p, q) of this lane's k index, advanced by 2 pixels per step without divisions
    int n = kk / HWs, p = (kk - n * HWs) / g.Ws, q = kk - n * HWs - p * g.Ws;
    for (int k0 = 0; k0 < ksteps; k0 += S5_U) {
        float bq[S5_U], aq[S5_U][2];
#pragma unroll
        for (int u = 0; u < S5_U; ++u) {
            const int pix = 2 * (k0 + u) + kk;
            const int pq = p * g.Ws + q;
            const int h = 5 * p + tr - g.pt, x = 5 * q + ts - g.pl;
            const bool bok = cok && pix < npix && h >= 0 && h < g.Hb && x >= 0 && x < g.Wb;
            bq[u] = ldbuf(rb, bok ? (((n * g.Cb + b) * g.Hb + h) * g.Wb + x) * 4 : S5_OOB);
#pragma unroll
            for (int mr = 0; mr < 2; ++mr) {
                const bool aok = pix < npix && a_lane[mr] < g.Cs;
                aq[u][mr] = ldbuf(rs, aok ? ((n * g.Cs + a_lane[mr]) * HWs + pq) * 4 : S5_OOB);
            }
            q += 2;
            while (q >= g.Ws) { q -= g.Ws; ++p; }
            while (p >= g.Hs) { p -= g.Hs; ++n; }
        }
#pragma unroll
        for (int u = 0; u < S5_U; ++u)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
                acc[mr] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][mr], bq[u], acc[mr], 0, 0, 0);
    }

    if (!cok) return;
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int a = row0 + mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a >= g.Cs) continue;
            float* o = dw + (size_t)a * cols + col;
            *o = accumulate ? *o + acc[mr][e] : acc[mr][e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wgrad, frame-streaming variant for 2x2 small images (the default arch: 8x8 -> 2x2):
// per frame n the update is  dW[a][(b,tap)] += sum_{pq<4} small[n][a][pq] * big[n][b][win(pq,tap)]
// and both operands of a frame are CONTIGUOUS in global memory (small[n][a0..a0+63] = 256
// floats, big[n][b] = Hb*Wb floats), so a stage of W5_F frames is staged with 16-byte LDS-DMA
// (coalesced, no per-lane gathers from global) and the window gather happens in LDS with two
// lane-constant offsets.  Workgroup = 4 waves, tile 64 a x 128 columns (5-7 big channels), the
// whole batch is reduced inside the workgroup (no split, no partials; deterministic).
// ---------------------------------------------------------------------------------------------
#define W5_F 8                 // frames per stage
#define W5_MAXB 7              // big-side channels a 128-column tile can touch
typedef float floatx4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(S5_THREADS, 2) void k_wgrad_s5f(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ dw,
    BnGeom g, int accumulate, int buf_floats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int cols = g.Cb * 25, HWb = g.Hb * g.Wb;
    const int row0 = blockIdx.x * 64;
    const int ctile = blockIdx.y * 128;
    const int bmin = ctile / 25;
    const int bmax = min(g.Cb - 1, (ctile + 127) / 25);
    const int nb = bmax - bmin + 1;
    const int a_gr = 64;                       // 16-byte groups of the A block per frame (64 a x 4)
    const int b_gr = nb * HWb / 4;             // groups of the B block per frame
    const int a_groups = W5_F * a_gr, groups = W5_F * (a_gr + b_gr);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * 4 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HWb * 4), 0x00020000);

    // column -> (b, r, s); the two pixels of this lane's k index (pq = 2u + kk) -> LDS word of
    // the B block of a frame, or -1 (padding / column past the tensor): operand 0
    const int col = ctile + 32 * wv + li;
    const bool cok = col < cols;
    const int b = cok ? col / 25 : bmin;
    const int tap = col - b * 25;
    const int tr = tap / 5, ts = tap - tr * 5;
    int boff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pq = 2 * u + kk;
        const int pp = pq / g.Ws, qq = pq - pp * g.Ws;
        const int h = 5 * pp + tr - g.pt, x = 5 * qq + ts - g.pl;
        const bool ok = cok && h >= 0 && h < g.Hb && x >= 0 && x < g.Wb;
        boff[u] = ok ? (b - bmin) * HWb + h * g.Wb + x : -1;
    }

    floatx16 acc[2];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mr][e] = 0.f;

    auto issue_dma = [&](int n0, int buf) {
        float* img = smem + buf * buf_floats;
        for (int e0 = 64 * wv; e0 < groups; e0 += S5_THREADS) {       // wave-uniform trip count
            const int e = e0 + lane;
            int off = S5_OOB;
            if (e0 < a_groups) {               // a_groups is a multiple of 64: whole waves
                const int f = e / a_gr, ga = e - f * a_gr;
                const bool ok = (n0 + f < g.N) && (row0 + ga < g.Cs);
                if (ok) off = (((n0 + f) * g.Cs + row0) * 4 + 4 * ga) * 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, img + 4 * e0, 16, off, 0, 0, 0);
            } else {
                const int eb = e - a_groups;
                const int f = eb / b_gr, gb = eb - f * b_gr;
                const bool ok = (e < groups) && (n0 + f < g.N);
                if (ok) off = (((n0 + f) * g.Cb + bmin) * HWb + 4 * gb) * 4;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, img + 4 * e0, 16, off, 0, 0, 0);
            }
        }
    };

    const int n_stages = (g.N + W5_F - 1) / W5_F;
    int cur = 0;
    issue_dma(0, 0);
    for (int st = 0; st < n_stages; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < n_stages) issue_dma((st + 1) * W5_F, cur ^ 1);
        const float* ab = smem + cur * buf_floats;
        const float* bb = ab + 4 * a_groups;
#pragma unroll
        for (int f = 0; f < W5_F; ++f) {
            // A: the four pixels of channel a in one 16-byte read; k index pq = 2u + kk
            floatx4s a4[2];
#pragma unroll
            for (int mr = 0; mr < 2; ++mr)
                a4[mr] = *reinterpret_cast<const floatx4s*>(ab + (f * 64 + mr * 32 + li) * 4);
            const float* bf = bb + f * nb * HWb;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float bv = boff[u] >= 0 ? bf[boff[u]] : 0.f;
#pragma unroll
                for (int mr = 0; mr < 2; ++mr) {
                    const float av = kk ? (u ? a4[mr].w : a4[mr].y) : (u ? a4[mr].z : a4[mr].x);
                    acc[mr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mr], 0, 0, 0);
                }
            }
        }
        cur ^= 1;
    }

    if (!cok) return;
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int a = row0 + mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a >= g.Cs) continue;
            float* o = dw + (size_t)a * cols + col;
            *o = accumulate ? *o + acc[mr][e] : acc[mr][e];
        }
    }
}

// frame-streaming variant applies: 2x2 small image, 16-byte aligned big frames, tiles fit
static bool s5f_ok(const BnGeom& g, int* buf_floats) {
    if (g.Hs * g.Ws != 4 || ((g.Hb * g.Wb) & 3) != 0) return false;
    if ((g.Cs & 63) != 0) return false;                 // whole 64-row A blocks (DMA rows)
    const int groups = W5_F * (64 + W5_MAXB * g.Hb * g.Wb / 4);
    *buf_floats = 4 * ((groups + 63) & ~63);            // whole wave rows of 64 groups
    return (size_t)2 * *buf_floats * 4 <= 64 * 1024;
}

// ---------------------------------------------------------------------------------------------
static bool s5_geom(const BnGeom& g) {
    if (g.R != 5 || g.S != 5 || g.stride != 5) return false;
    // every big-side pixel must map to a window inside the small image
    if (g.Hb + g.pt > 5 * g.Hs || g.Wb + g.pl > 5 * g.Ws) return false;
    if (g.pt > 4 || g.pl > 4) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.Cs * g.Cb * 25 * 4 >= 0x7fffffffull) return false;
    return g.Cs >= 32 && g.Cb >= 16;
}

BnFastPlan bn_s5_up_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_up_generic", 0, 0, 0, 0, 0, 0};
    if (!s5_geom(g)) return p;
    p.supported = true;
    p.kernel_name = "k_up_s5";
    return p;
}

BnFastPlan bn_s5_wgrad_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (!s5_geom(g)) return p;
    p.supported = true;
    p.kernel_name = "k_wgrad_s5";
    return p;
}

int bn_launch_up_s5(const float* small, const float* w, const float* bias, float* out,
                    const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                    hipStream_t st) {
    const int rows = g.Cb * 25, npix = g.N * g.Hs * g.Ws;
    dim3 grid((rows + 63) / 64, (npix + 127) / 128);
    hipLaunchKernelGGL(k_up_s5, grid, dim3(S5_THREADS), 0, st, small, w, bias, out, dact_src, g,
                       act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_wgrad_s5(const float* small, const float* big, float* dw, const BnGeom& g,
                       int accumulate, hipStream_t st) {
    const int cols = g.Cb * 25;
    dim3 grid((g.Cs + 63) / 64, (cols + 127) / 128);
    int buf_floats = 0;
    if (s5f_ok(g, &buf_floats)) {
        hipLaunchKernelGGL(k_wgrad_s5f, grid, dim3(S5_THREADS), (size_t)2 * buf_floats * 4, st,
                           small, big, dw, g, accumulate, buf_floats);
        BN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_wgrad_s5, grid, dim3(S5_THREADS), 0, st, small, big, dw, g, accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}
