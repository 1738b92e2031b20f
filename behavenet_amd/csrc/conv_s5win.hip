// Stride == kernel (5x5, stride 5) layers between ANY pair of maps: the last layer of the default
// architecture (reference ae_model_architecture_generator.py:707-720) on every frame size -- 12x12 <-> 3x3
// (192x192 frames), 12x10 <-> 3x2 (192x160), 4x3 <-> 1x1 (64x48, offsets (0, 1)), 2x2 <-> 1x1 (32x32,
// offsets (1, 1) with two rows / columns of padding behind) ... -- all three roles, im2col-free.  (The
// benchmark's 8x8 <-> 2x2 geometry keeps its own kernels, conv_qgemm.hip / conv_qgemm2.hip.)
//
// The windows of the small-side pixels do not overlap.  Window z = (p, q) sees big rows 5p + r - pt for
// the taps r in [r0(p), r1(p)) that fall ON the map, likewise columns: a rectangle of T = nr x ns taps
// (9 .. 25 of 25).  Per role:
//
//   down  (conv fwd, convT bwd-data)  S[n][m][z]         = sum_{c,t} B[n][c][pix(z,t)] W[m][c][tap(z,t)]
//         one dense GEMM per window over ITS taps only (the padded taps are never multiplied):
//         rows = frames, columns = small-side channels, K = Cb x T; reduction split over workgroups
//         when the windows' tiles do not fill the chip, finished by k_s5w_finish_down;
//   up    (convT fwd, conv bwd-data)  B[n][c][pix(z,t)]  = sum_m S[n][m][z] W[m][c][tap(z,t)]
//         one dense GEMM per window: rows = frames, columns = (c, t), K = Cs; bias / activation /
//         derivative of the layer below in the epilogue, scattered straight to NCHW (every big pixel
//         belongs to exactly one window);
//   wgrad                             dW[m][c][r][s]    (+)= sum_{n,z} S[n][m][z] B[n][c][5p+r-pt][5q+s-pl]
//         ONE GEMM over K = (frame, window) with zeros for taps off the map: rows = small-side
//         channels, columns = (c, tap) = dW as it lies in memory -- no partial tiles, no finish pass.
//
// No column matrix is ever materialised: operands are gathered from the NCHW tensors by buffer loads
// whose lane offsets are computed ONCE per workgroup -- a stage of the reduction is a whole number of
// channels (down), 32 small-side channels (up) or a whole number of frames (wgrad), so a stage only adds
// a scalar offset -- and staged through LDS (rows of KS + 1 words: conflict-free MFMA operand reads),
// the next stage's loads issued from inside the MFMA stream of the current one (DESIGN.md section 4,
// issue rules).  Tile 64 x 128 per workgroup of four waves, v_mfma_f32_32x32x2_f32, fp32 throughout.
//
// Lab hooks (compile-time, `tools/build_variant.sh <name> conv_s5win.hip -D...`, measured with tools/bench_s5.py):
// SW_ABL_NOFETCH (no gathers after the first stage), SW_ABL_NOLDSW (no LDS writes), SW_ABL_NOBAR (no barriers),
// SW_ABL_NOEPI (no epilogue), SW_SPLIT_TARGET=<workgroups> (reduction slices of the down role).  Results: DESIGN.md 4.
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define SW_T 64          // tile rows
#define SW_NB 2          // column granules of 64 per tile
enum { SW_DOWN = 0, SW_UP = 1, SW_WGRAD = 2 };

struct SWArgs {
    const float* small;
    const float* big;
    const float* w;
    float* part;         // down: scratch [split][Z][N][Cs]
    float* out;          // up: the big-side tensor; wgrad: dW
    const float* bias;
    const float* dact_src;
    int act, dact;
    float slope;
    int N, Cs, Cb, Hs, Ws, Hb, Wb, pt, pl;
    int Z;               // windows = Hs * Ws
    int per;             // wgrad: frames per stage (down: channels per stage = KS / T of the window)
    int kper;            // down: channels per reduction split
    int accumulate;      // wgrad
    int nz;              // down / up: windows in launch order (heaviest first)
    int nrow, ncol, splits, maxper;      // launch geometry (see the kernels' index decode)
    unsigned char zlist[44];
    unsigned char zdepth[44];            // down: stage depth of window zlist[i]
    unsigned char ztiles[44];            // up: column tiles of window zlist[i]
};

// x / d for the small non-negative values of the set-up code (x < 65536, d < 65536) without the ~35
// instructions of a 32-bit division: a workgroup's 30 gather elements took two or three of them each
struct SWDiv { unsigned m; int d; };
__host__ __device__ static inline SWDiv sw_div(int d) { SWDiv v; v.d = d; v.m = d > 1 ? 0xffffffffu / (unsigned)d + 1u : 0u; return v; }
__device__ __forceinline__ int sw_quot(int x, const SWDiv& v) { return v.d == 1 ? x : (int)__umulhi((unsigned)x, v.m); }

struct SWWin { int r0, nr, s0, ns, y0, x0, T; };
__host__ __device__ static inline SWWin sw_window(int p, int q, int Hb, int Wb, int pt, int pl) {
    SWWin v;
    v.r0 = pt - 5 * p > 0 ? pt - 5 * p : 0;
    const int r1 = Hb + pt - 5 * p < 5 ? Hb + pt - 5 * p : 5;
    v.s0 = pl - 5 * q > 0 ? pl - 5 * q : 0;
    const int s1 = Wb + pl - 5 * q < 5 ? Wb + pl - 5 * q : 5;
    v.nr = r1 - v.r0; v.ns = s1 - v.s0;
    v.y0 = 5 * p + v.r0 - pt; v.x0 = 5 * q + v.s0 - pl;
    v.T = v.nr * v.ns;
    return v;
}

template <int MODE, int KS>
__device__ __forceinline__ void sw_body(const SWArgs& a, const int z, const int ks, const int i0, const int j0,
                                        float* __restrict__ As, float* __restrict__ Bs) {
    constexpr int LD = KS + 1;                              // odd row stride
    constexpr int EA = (SW_T * KS + 255) / 256;             // operand elements per thread and stage
    constexpr int EB = (SW_NB * SW_T * KS + 255) / 256;
    constexpr int OOB = 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int HW = a.Hb * a.Wb;
    const SWWin win = sw_window(z / a.Ws, z % a.Ws, a.Hb, a.Wb, a.pt, a.pl);
    const int T = win.T;
    if (MODE == SW_UP && j0 >= a.Cb * T) return;           // (the grid is sized for 25-tap windows)
    const SWDiv dT = sw_div(T), dns = sw_div(win.ns), dZ = sw_div(a.Z), dWs = sw_div(a.Ws),
                dblk = sw_div(SW_T * a.Z);

    // the reduction in stage units: channels (down), small-side channels (up), frames (wgrad)
    int ubeg, uend, ustep;
    if (MODE == SW_DOWN) { ubeg = ks * a.kper; uend = min(a.Cb, ubeg + a.kper); ustep = KS / T; }
    else if (MODE == SW_UP) { ubeg = 0; uend = a.Cs; ustep = KS; }
    else { ubeg = 0; uend = a.N; ustep = a.per; }
    const int kused = MODE == SW_DOWN ? ustep * T : (MODE == SW_UP ? KS : a.per * a.Z);   // <= KS
    if (MODE == SW_WGRAD) {
        // the stage's padding columns of the small-side tile are written by nobody: zero once
        for (int i = tid; i < SW_T * LD; i += 256) As[i] = 0.f;
    }

    // ---- per-thread gather lists, computed once: byte offset (or OOB), the element's stage unit
    //      (for the tail test) and its LDS word ---------------------------------------------------
    int avo[EA], aun[EA], als[EA];
    int bvo[EB], bun[EB], bls[EB];
    size_t a_bytes, b_bytes;
    const void *a_base, *b_base;
    int a_ustride, b_ustride;                               // bytes per stage unit
    if (MODE == SW_DOWN) {
        a_base = a.big;  a_bytes = (size_t)a.N * a.Cb * HW * 4;  a_ustride = HW * 4;
        b_base = a.w;    b_bytes = (size_t)a.Cs * a.Cb * 100;    b_ustride = 100;
    } else if (MODE == SW_UP) {
        a_base = a.small; a_bytes = (size_t)a.N * a.Cs * a.Z * 4; a_ustride = a.Z * 4;
        b_base = a.w;     b_bytes = (size_t)a.Cs * a.Cb * 100;    b_ustride = a.Cb * 100;
    } else {
        a_base = a.small; a_bytes = (size_t)a.N * a.Cs * a.Z * 4; a_ustride = a.Cs * a.Z * 4;
        b_base = a.big;   b_bytes = (size_t)a.N * a.Cb * HW * 4;   b_ustride = a.Cb * HW * 4;
    }
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        const int L = e * 256 + tid;
        int row, kl, vo = OOB, un = 0;
        if (MODE == SW_WGRAD) {
            // a frame's small map [Cs][Z] is contiguous: (row, window) runs along the threads
            const int blk = SW_T * a.Z;
            const int f = sw_quot(L, dblk), b = L - f * blk;
            row = sw_quot(b, dZ);
            const int zz = b - row * a.Z;
            kl = f * a.Z + zz;
            un = f;
            if (f < a.per && i0 + row < a.Cs) vo = ((f * a.Cs + i0 + row) * a.Z + zz) * 4;
            if (f >= a.per) { row = SW_T; kl = 0; }        // (no such element: not stored)
        } else {
            row = L / KS;
            kl = L - row * KS;
            if (MODE == SW_DOWN) {
                const int cc = sw_quot(kl, dT), t = kl - cc * T;
                const int rr = sw_quot(t, dns), ss = t - rr * win.ns;
                un = cc;
                if (row < SW_T && kl < kused && i0 + row < a.N)
                    vo = (((i0 + row) * a.Cb + cc) * HW + (win.y0 + rr) * a.Wb + win.x0 + ss) * 4;
            } else {
                un = kl;
                if (row < SW_T && i0 + row < a.N) vo = (((i0 + row) * a.Cs + kl) * a.Z + z) * 4;
            }
        }
        avo[e] = vo; aun[e] = un;
        als[e] = row < SW_T ? row * LD + kl : -1;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
        const int L = e * 256 + tid;
        int col, kl, vo = OOB, un = 0;
        if (MODE == SW_DOWN) {
            // W[m][c][taps]: the reduction index runs along the threads
            col = L / KS;
            kl = L - col * KS;
            const int cc = sw_quot(kl, dT), t = kl - cc * T;
            const int rr = sw_quot(t, dns), ss = t - rr * win.ns;
            un = cc;
            if (col < SW_NB * SW_T && kl < kused && j0 + col < a.Cs)
                vo = (((j0 + col) * a.Cb + cc) * 25 + (win.r0 + rr) * 5 + win.s0 + ss) * 4;
        } else {
            // columns (channel, tap / pixel) run along the threads
            kl = L / (SW_NB * SW_T);
            col = L - kl * (SW_NB * SW_T);
            const int j = j0 + col;
            if (MODE == SW_UP) {
                const int c = sw_quot(j, dT), t = j - c * T;
                const int rr = sw_quot(t, dns), ss = t - rr * win.ns;
                un = kl;
                if (kl < KS && c < a.Cb) vo = ((kl * a.Cb + c) * 25 + (win.r0 + rr) * 5 + win.s0 + ss) * 4;
            } else {
                const int c = j / 25, tap = j - c * 25;
                const int r = tap / 5, s = tap - r * 5;
                const int f = sw_quot(kl, dZ), zz = kl - f * a.Z;
                const int p = sw_quot(zz, dWs), q = zz - p * a.Ws;
                const int y = 5 * p + r - a.pt, x = 5 * q + s - a.pl;
                un = f;
                if (kl < kused && c < a.Cb && y >= 0 && y < a.Hb && x >= 0 && x < a.Wb)
                    vo = ((f * a.Cb + c) * HW + y * a.Wb + x) * 4;
            }
            if (kl >= KS) col = SW_NB * SW_T;
        }
        bvo[e] = vo; bun[e] = un;
        bls[e] = col < SW_NB * SW_T ? col * LD + kl : -1;
    }
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);

    float ra[EA], rb[EB];
    auto fetch = [&](const int u0) __attribute__((always_inline)) {
        const bool tail = u0 + ustep > uend;                // wave-uniform: units past the end read 0.0f
        const int sa = u0 * a_ustride, sb = u0 * b_ustride;
        if (!tail) {
            // (the common stage: no per-element select -- 30 v_cndmask inside the MFMA stream otherwise)
#pragma unroll
            for (int e = 0; e < EA; ++e)
                ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, avo[e], sa, 0));
#pragma unroll
            for (int e = 0; e < EB; ++e)
                rb[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, bvo[e], sb, 0));
        } else {
#pragma unroll
            for (int e = 0; e < EA; ++e) {
                const int vo = (u0 + aun[e] >= uend) ? OOB : avo[e];
                ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, vo, sa, 0));
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int vo = (u0 + bun[e] >= uend) ? OOB : bvo[e];
                rb[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, vo, sb, 0));
            }
        }
    };

    floatx16 acc[SW_NB];
#pragma unroll
    for (int h = 0; h < SW_NB; ++h)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[h][t] = 0.f;
    // wave (wv >> 1, wv & 1): rows 32 (wv >> 1).., columns 32 (wv & 1).. of EVERY granule
    const float* ap = As + ((wv >> 1) * 32 + li) * LD + lk;
    const float* bp = Bs + ((wv & 1) * 32 + li) * LD + lk;

    if (ubeg < uend) fetch(ubeg);
    for (int u0 = ubeg; u0 < uend; u0 += ustep) {
#ifndef SW_ABL_NOBAR
        __syncthreads();                       // everyone is done reading the previous stage
#endif
#ifndef SW_ABL_NOLDSW
#pragma unroll
        for (int e = 0; e < EA; ++e)
            if (als[e] >= 0) As[als[e]] = ra[e];
#pragma unroll
        for (int e = 0; e < EB; ++e)
            if (bls[e] >= 0) Bs[bls[e]] = rb[e];
#endif
#ifndef SW_ABL_NOBAR
        __syncthreads();
#endif
#pragma unroll
        for (int t = 0; t < KS / 2; ++t) {
            const float av = ap[2 * t];
#pragma unroll
            for (int h = 0; h < SW_NB; ++h)
                acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[SW_T * h * LD + 2 * t], acc[h], 0, 0, 0);
            if (t == 1) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef SW_ABL_NOFETCH
                if (u0 + ustep < uend) fetch(u0 + ustep);   // rides in this MFMA stream
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // lane holds C[(t & 3) + 8 (t >> 2) + 4 lk][li] of its 32 x 32 block of granule h
    const int irow = i0 + (wv >> 1) * 32 + 4 * lk;
#ifdef SW_ABL_NOEPI
    if (acc[0][0] != 12345.678f) return;
#endif
    if (MODE == SW_DOWN) {
        float* dst = a.part + ((size_t)(ks * a.Z + z) * a.N) * a.Cs;
#pragma unroll
        for (int h = 0; h < SW_NB; ++h) {
            const int j = j0 + SW_T * h + (wv & 1) * 32 + li;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = irow + (t & 3) + 8 * (t >> 2);
                if (i < a.N && j < a.Cs) dst[(size_t)i * a.Cs + j] = acc[h][t];
            }
        }
    } else if (MODE == SW_UP) {
        const size_t frame = (size_t)a.Cb * HW;
#pragma unroll
        for (int h = 0; h < SW_NB; ++h) {
            const int j = j0 + SW_T * h + (wv & 1) * 32 + li;
            const int c = sw_quot(j, dT), tt = j - c * T;
            if (c >= a.Cb) continue;
            const int rr = sw_quot(tt, dns), ss = tt - rr * win.ns;
            const size_t col = (size_t)c * HW + (win.y0 + rr) * a.Wb + win.x0 + ss;
            const float bj = a.bias ? a.bias[c] : 0.f;
            float d[16];
            if (a.dact_src) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int i = irow + (t & 3) + 8 * (t >> 2);
                    d[t] = i < a.N ? a.dact_src[(size_t)i * frame + col] : 0.f;
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = irow + (t & 3) + 8 * (t >> 2);
                if (i >= a.N) continue;
                float v = bn_apply_act(acc[h][t] + bj, a.act, a.slope);
                if (a.dact_src) v *= bn_act_grad_from_output(d[t], a.dact, a.slope);
                a.out[(size_t)i * frame + col] = v;
            }
        }
    } else {
        const int ncol = a.Cb * 25;
#pragma unroll
        for (int h = 0; h < SW_NB; ++h) {
            const int j = j0 + SW_T * h + (wv & 1) * 32 + li;
            if (j >= ncol) continue;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = irow + (t & 3) + 8 * (t >> 2);
                if (i >= a.Cs) continue;
                float* o = a.out + (size_t)i * ncol + j;
                *o = a.accumulate ? *o + acc[h][t] : acc[h][t];
            }
        }
    }
}

// ---- launch geometry: 1-D grids decoded so that the workgroups sharing an operand slab sit on ONE XCD
// (block b runs on XCD b % 8 -- observed, MI355X_MICROARCH.md "Workgroup dispatch": a speed choice only, any
// placement computes the same thing).  The operands of these layers are tens of MB gathered 4 bytes at a time:
// spread over the XCDs every L2 saw every line and the gathers ran at the Infinity-Cache rate.
#define SW_LDS_FLOATS ((SW_T + SW_NB * SW_T) * 41)

// down: a GROUP = the windows x column tiles of one (row tile, reduction slice): they read the same 64 frames x
// `kper` channels of the big map (1.2 MB at 12x12 maps) and the same channel slice of the weights
__global__ __launch_bounds__(256) void k_s5win_down(SWArgs a) {
    __shared__ float lds[SW_LDS_FLOATS];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int gsize = a.nz * a.ncol;
    const int group = xcd + 8 * (slot / gsize), member = slot % gsize;
    if (group >= a.nrow * a.splits) return;
    const int y = group % a.nrow, ks = group / a.nrow;
    const int zi = member / a.ncol, x = member - zi * a.ncol;
    const int z = a.zlist[zi];
    const int i0 = y * SW_T, j0 = x * (SW_NB * SW_T);
    switch (a.zdepth[zi]) {                                 // (uniform: the window's stage depth)
    case 26: sw_body<SW_DOWN, 26>(a, z, ks, i0, j0, lds, lds + SW_T * 27); break;
    case 32: sw_body<SW_DOWN, 32>(a, z, ks, i0, j0, lds, lds + SW_T * 33); break;
    case 36: sw_body<SW_DOWN, 36>(a, z, ks, i0, j0, lds, lds + SW_T * 37); break;
    default: sw_body<SW_DOWN, 40>(a, z, ks, i0, j0, lds, lds + SW_T * 41); break;
    }
}

// up: XCD k takes the column tiles of EVERY window that lie in the k-th eighth of the big-side channels, row
// tile by row tile: the windows of a (frames, channels) slab of the output complete each other's cache lines in
// that XCD's L2 (a window writes runs of 3-5 floats)
__global__ __launch_bounds__(256) void k_s5win_up(SWArgs a) {
    __shared__ float lds[SW_LDS_FLOATS];
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int y = slot / a.maxper;
    int idx = slot - y * a.maxper;
    int z = -1, x = 0;
    for (int zi = 0; zi < a.nz; ++zi) {
        const int nt = a.ztiles[zi];
        const int lo = (xcd * nt) >> 3, hi = ((xcd + 1) * nt) >> 3;
        if (idx < hi - lo) { z = a.zlist[zi]; x = lo + idx; break; }
        idx -= hi - lo;
    }
    if (z < 0) return;
    sw_body<SW_UP, 32>(a, z, 0, y * SW_T, x * (SW_NB * SW_T), lds, lds + SW_T * 33);
}

// wgrad: the row tiles (small-side channels) of one column tile on one XCD: they gather the same (frame,
// window) x (channel, tap) slab of the big map
template <int KS>
__global__ __launch_bounds__(256) void k_s5win_wgrad(SWArgs a) {
    __shared__ float lds[SW_LDS_FLOATS];
    const int id = blockIdx.x, xk = id & 7, slot = id >> 3;
    const int y = slot % a.nrow, x = (slot / a.nrow) * 8 + xk;
    if (x >= a.ncol) return;
    sw_body<SW_WGRAD, KS>(a, 0, 0, y * SW_T, x * (SW_NB * SW_T), lds, lds + SW_T * (KS + 1));
}

// out_s[n][m][z] = epi( sum_split P[split][z][n][m] + bias[m] ), fixed summation order
__global__ __launch_bounds__(256) void k_s5w_finish_down(
    const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
    const float* __restrict__ dact_src, unsigned N, unsigned Cs, unsigned Z, int splits, int act, int dact,
    float slope) {
    const size_t total = (size_t)N * Cs * Z;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const unsigned z = (unsigned)(idx % Z);
    const size_t nm = idx / Z;                          // n * Cs + m
    const size_t plane = (size_t)N * Cs;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[((size_t)s * Z + z) * plane + nm];
    if (bias) v += bias[nm % Cs];
    v = bn_apply_act(v, act, slope);
    if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
    out[idx] = v;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool bn_s5win_supported(const BnGeom& g) {
    static int disabled = -1;                          // BN_S5WIN=0: the previous paths (tuning builds)
    if (disabled < 0) { const char* e = bn_tune_env("BN_S5WIN"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (g.R != 5 || g.S != 5 || g.stride != 5 || g.CsS != 0 || g.KV != 0 || g.K0 != 0) return false;
    if (g.pt > 4 || g.pl > 4 || g.Hs < 1 || g.Ws < 1) return false;
    // the windows tile the big map: every big pixel in exactly one window (what TF-"same" padding plans),
    // every window with at least one tap on the map
    if (5 * g.Hs - g.pt < g.Hb || 5 * g.Ws - g.pl < g.Wb) return false;
    if (5 * (g.Hs - 1) - g.pt >= g.Hb || 5 * (g.Ws - 1) - g.pl >= g.Wb) return false;
    if (g.Hs * g.Ws > 40 || g.Cb > 1280) return false;   // (a wgrad stage is at least one frame's windows; tile counts in a byte)
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.Cs * g.Cb * 100 >= 0x7fffffffull) return false;
    return true;
}

// stage depth for a stage unit of `unit` reduction elements: the instantiated depth (26 / 32 / 36 / 40)
// that holds a whole number of units with the least padding; -> units per stage
static int sw_stage(int unit, int* ks_out) {
    static const int depths[4] = {32, 36, 40, 26};
    int best_ks = 0, best_per = 0;
    float best_fill = 0.f;
    for (int d = 0; d < 4; ++d) {
        const int per = depths[d] / unit;
        if (per < 1) continue;
        const float fill = (float)(per * unit) / (float)depths[d];
        if (fill > best_fill + 1e-6f) { best_fill = fill; best_ks = depths[d]; best_per = per; }
    }
    *ks_out = best_ks;
    return best_per;
}

// down role: a window of T taps runs KS / T channels per stage; the instantiated depth (26 / 32 / 36 / 40)
// that its stages fill best -- 25 taps: 26 (one channel), 20: 40 (two), 16: 32, 15: 32 (30 of 32), 12: 36
// (three), 9: 36 (four) ... -- the windows of a map are launched in groups of equal depth (a run-time
// step count inside one launch was tried: the scalar branches between the MFMAs doubled the kernel time)
static int sw_window_depth(int T) {
    static const int depths[4] = {40, 36, 32, 26};
    int best = 40;
    float fill = 0.f;
    for (int d = 0; d < 4; ++d) {
        const int per = depths[d] / T;
        if (per < 1) continue;
        const float f = (float)(per * T) / (float)depths[d];
        if (f > fill + 1e-6f) { fill = f; best = depths[d]; }
    }
    return best;
}

// reduction splits of the down role: a function of the geometry only, NOT of the number of frames beyond
// the tile count of small batches (tile rows are independent: a frame's result does not depend on the
// batch it is in, as for k_qgemm)
static int sw_down_splits(const BnGeom& g) {
    // the windows' work differs by their tap counts (9 .. 25 of 25) and the workgroups of a window are few
    // (3x3 windows, 512 channels, 256 frames: 16 tiles each): enough slices that the chip is filled about
    // four times over and the dispatcher evens the windows out -- slices of at least 32 channels
    const int tiles = ((g.Cs + SW_NB * SW_T - 1) / (SW_NB * SW_T)) * g.Hs * g.Ws * 4;   // at 256 frames
    int s = 1;
#ifndef SW_SPLIT_TARGET
#define SW_SPLIT_TARGET 1024
#endif
    while (tiles * s < SW_SPLIT_TARGET && s < 16 && g.Cb / (2 * s) >= 32) s *= 2;
    return s;
}

size_t bn_s5win_ws_bytes(int role, const BnGeom& g) {
    if (role != SW_DOWN) return 0;
    return (size_t)sw_down_splits(g) * g.Hs * g.Ws * g.N * g.Cs * sizeof(float);
}

// windows in launch order: heaviest (most taps) first, so that the dispatcher's tail is made of light ones
static void sw_fill(SWArgs* a, const BnGeom& g) {
    a->N = g.N; a->Cs = g.Cs; a->Cb = g.Cb; a->Hs = g.Hs; a->Ws = g.Ws; a->Hb = g.Hb; a->Wb = g.Wb;
    a->pt = g.pt; a->pl = g.pl; a->Z = g.Hs * g.Ws;
    a->nz = a->Z;
    int T[44];
    for (int z = 0; z < a->Z; ++z) {
        a->zlist[z] = (unsigned char)z;
        T[z] = sw_window(z / g.Ws, z % g.Ws, g.Hb, g.Wb, g.pt, g.pl).T;
    }
    for (int i = 1; i < a->Z; ++i)                         // insertion sort by descending tap count (stable)
        for (int j = i; j > 0 && T[a->zlist[j]] > T[a->zlist[j - 1]]; --j) {
            const unsigned char t = a->zlist[j]; a->zlist[j] = a->zlist[j - 1]; a->zlist[j - 1] = t;
        }
    for (int i = 0; i < a->Z; ++i) {
        const int t = T[a->zlist[i]];
        a->zdepth[i] = (unsigned char)sw_window_depth(t);
        a->ztiles[i] = (unsigned char)((g.Cb * t + SW_NB * SW_T - 1) / (SW_NB * SW_T));
    }
}

int bn_launch_s5win_down(const float* big, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         void* ws, hipStream_t st) {
    SWArgs a = {};
    sw_fill(&a, g);
    a.big = big; a.w = w; a.part = (float*)ws;
    a.splits = sw_down_splits(g);
    a.kper = (g.Cb + a.splits - 1) / a.splits;
    a.nrow = (g.N + SW_T - 1) / SW_T;
    a.ncol = (g.Cs + SW_NB * SW_T - 1) / (SW_NB * SW_T);
    const int groups = a.nrow * a.splits;
    const dim3 grid((unsigned)(8 * ((groups + 7) / 8) * a.nz * a.ncol));
    BN_LAUNCH_MAIN(k_s5win_down, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    const size_t total = (size_t)g.N * g.Cs * a.Z;
    hipLaunchKernelGGL(k_s5w_finish_down, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, bias, out, dact_src, (unsigned)g.N, (unsigned)g.Cs, (unsigned)a.Z,
                       a.splits, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_s5win_up(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       hipStream_t st) {
    SWArgs a = {};
    sw_fill(&a, g);
    a.small = small; a.w = w; a.out = out; a.bias = bias; a.dact_src = dact_src;
    a.act = act; a.dact = dact; a.slope = slope;
    a.nrow = (g.N + SW_T - 1) / SW_T;
    a.maxper = 0;
    for (int k = 0; k < 8; ++k) {
        int n = 0;
        for (int i = 0; i < a.nz; ++i) n += (((k + 1) * a.ztiles[i]) >> 3) - ((k * a.ztiles[i]) >> 3);
        if (n > a.maxper) a.maxper = n;
    }
    const dim3 grid((unsigned)(8 * a.nrow * a.maxper));
    BN_LAUNCH_MAIN(k_s5win_up, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_s5win_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                          int accumulate, hipStream_t st) {
    SWArgs a = {};
    sw_fill(&a, g);
    a.small = small; a.big = big; a.out = dw; a.accumulate = accumulate;
    int ks = 32;
    a.per = sw_stage(a.Z, &ks);
    a.nrow = (g.Cs + SW_T - 1) / SW_T;
    a.ncol = (g.Cb * 25 + SW_NB * SW_T - 1) / (SW_NB * SW_T);
    const dim3 grid((unsigned)(8 * a.nrow * ((a.ncol + 7) / 8)));
    if (ks == 26) BN_LAUNCH_MAIN(k_s5win_wgrad<26>, grid, dim3(256), 0, st, a);
    else if (ks == 36) BN_LAUNCH_MAIN(k_s5win_wgrad<36>, grid, dim3(256), 0, st, a);
    else if (ks == 40) BN_LAUNCH_MAIN(k_s5win_wgrad<40>, grid, dim3(256), 0, st, a);
    else BN_LAUNCH_MAIN(k_s5win_wgrad<32>, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    return 0;
}
