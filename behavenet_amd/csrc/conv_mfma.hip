// FLOP-bound convolution layers on the matrix cores: im2col-free direct convolutions whose
// inner products run on v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 (f32 in, f32
// accumulate: bit-identical to an fmaf chain, 157 TF peak = the fp32 vector peak, but fed with
// one VGPR per operand and leaving the VALU free for addressing).
//
// No im2col matrix is ever materialised: a workgroup stages an input tile (with its halo and
// the zero padding) and a slice of the weights in LDS, and every lane gathers its MFMA operand
// straight from that tile.  See DESIGN.md "Kernel families" for the tilings.
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define MF_THREADS 256
#define MF_MAX_LDS (64 * 1024)

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

// =============================================================================================
// family 1: "gather-down" (conv forward, transposed-conv data gradient)
//   out[n,m,p,q] = sum_{c,r,s} big[n,c,p*ST+r-pt,q*ST+s-pl] * W[m][c][r][s]
// MFMA roles: A = weights (row i = output channel), B = input pixels (col j = output pixel),
// the reduction index runs over (tap, channel pair): lane half kk = l>>5 takes channel c+kk.
// Workgroup tile: TM = 32*MR output channels x TP = 128*NR output pixels (4 waves, each wave
// owns NR pixel blocks and all MR channel blocks); pixels are F frames x PT_H rows x Q columns.
// =============================================================================================
struct DownTile {
    int F, PT_H, PTQ;                 // frames / small-map rows per workgroup tile, PT_H * Ws
    int IH, IWp, FS, CHS;
    int tiles_per_frame;
    int TMP;
    int xl_floats;
    int c_per_split;     // channels of the reduction handled by one blockIdx.z
    int splits;
    int dbg;             // BN_DOWN_DBG experiments (0 in production)
};

// Software pipeline: the global loads of chunk i+1 (input tile + weight slice) are issued into
// registers right after chunk i has been published to LDS, and stay in flight behind chunk i's
// MFMAs; the gather offsets of the input tile are chunk-invariant and live in registers.
template <int MR, int NR, int CC, int ST, int R, int S, int KIN>
__global__ __launch_bounds__(MF_THREADS, 2) void k_down_mfma(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, DownTile t, int act,
    int dact, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xl = smem;
    float* wl = smem + t.xl_floats;
    constexpr int RS = R * S;
    constexpr int TM = 32 * MR;
    constexpr int TMP = TM + 1;                       // odd row stride: conflict-free transpose
    constexpr int WROWS = TM / 4;                     // weight rows per wave
    constexpr int WPASS = (CC * RS + 63) / 64;        // 64-lane passes along (channel, tap)
    constexpr int WK = WROWS * WPASS;                 // weight loads per thread per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    const int grp = blockIdx.x / t.tiles_per_frame;
    const int rowt = blockIdx.x - grp * t.tiles_per_frame;
    const int n0 = grp * t.F;
    const int p0 = rowt * t.PT_H;
    const int m0 = blockIdx.y * TM;
    const int Q = g.Ws, PQ = g.Hs * g.Ws;
    const int HW = g.Hb * g.Wb;

    int base[NR];
    size_t opix[NR];
    bool pvalid[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        // (maps that are no powers of two leave pixels of a tile without an output: they multiply the
        // tile's first pixel and store nothing)
        int pix = 32 * (wv * NR + nr) + li;
        const bool inside = pix < t.F * t.PTQ;
        if (!inside) pix = 0;
        const int f = pix / t.PTQ;
        const int rem = pix - f * t.PTQ;
        const int pj = rem / Q, qj = rem - pj * Q;
        base[nr] = f * t.FS + (ST * pj) * t.IWp + ST * qj + kk * t.CHS;
        pvalid[nr] = inside && (n0 + f) < g.N && (p0 + pj) < g.Hs;
        opix[nr] = (size_t)(n0 + f) * g.Cs * PQ + (size_t)(p0 + pj) * Q + qj;
    }

    // chunk-invariant gather offsets of this thread's input-tile elements, relative to
    // big[n0][c][0][0]:  >= 0 source offset, -1 zero (padding / halo / frame tail), -2 no element
    int ioff[KIN];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const int e = tid + MF_THREADS * k;
        int off = -2;
        if (e < t.CHS) {
            const int f = e / t.FS;
            const int r2 = e - f * t.FS;
            const int y = r2 / t.IWp;
            const int x = r2 - y * t.IWp;
            const int hb = ST * p0 - g.pt + y, wb = x - g.pl;
            const bool ok = (n0 + f < g.N) && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            off = ok ? f * (g.Cb * HW) + hb * g.Wb + wb : -1;
        }
        ioff[k] = off;
    }

    floatx16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    const int c_beg = blockIdx.z * t.c_per_split;
    const int c_end = min(g.Cb, c_beg + t.c_per_split);

    float xr[CC][KIN];
    float wr[WK];

    auto issue_loads = [&](int c0) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int c = min(c0 + cc, g.Cb - 1);
            const float* bp = big + ((size_t)n0 * g.Cb + c) * HW;
#pragma unroll
            for (int k = 0; k < KIN; ++k) xr[cc][k] = bp[max(ioff[k], 0)];
        }
        // weights: wave wv fetches rows m = wv, wv+4, ... of the (TM x CC*RS) slice; lanes run
        // along the contiguous (channel, tap) axis, so the k-dependent address part is scalar
        const float* wp = w + ((size_t)(m0 + wv) * g.Cb + c0) * RS + lane;
#pragma unroll
        for (int k = 0; k < WROWS; ++k) {
            const int m = min(m0 + wv + 4 * k, g.Cs - 1) - (m0 + wv);
            const float* rp = wp + (size_t)m * g.Cb * RS;
#pragma unroll
            for (int ps = 0; ps < WPASS; ++ps) {
                const int r2 = lane + 64 * ps;
                const bool ok = (r2 < CC * RS) && (c0 + r2 / RS < g.Cb);
                wr[k * WPASS + ps] = rp[ok ? 64 * ps : -lane];
            }
        }
    };

    issue_loads(c_beg);
    for (int c0 = c_beg; c0 < c_end; c0 += CC) {
        __syncthreads();   // the previous chunk's MFMA reads of LDS are complete
        if (!(t.dbg & 1) || c0 == c_beg) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const bool cok = c0 + cc < c_end;
#pragma unroll
            for (int k = 0; k < KIN; ++k) {
                if (ioff[k] != -2)
                    xl[cc * t.CHS + tid + MF_THREADS * k] = (cok && ioff[k] >= 0) ? xr[cc][k] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < WROWS; ++k) {
            const bool mok = m0 + wv + 4 * k < g.Cs;
#pragma unroll
            for (int ps = 0; ps < WPASS; ++ps) {
                const int r2 = lane + 64 * ps;
                if (r2 < CC * RS) {
                    const bool ok = mok && (c0 + r2 / RS < c_end);
                    wl[r2 * TMP + wv + 4 * k] = ok ? wr[k * WPASS + ps] : 0.f;
                }
            }
        }
        }
        __syncthreads();
        if (c0 + CC < c_end && !(t.dbg & 1)) issue_loads(c0 + CC);   // behind the MFMAs below

        // MFMA loop, one (channel pair, kernel row) = S taps per "row".  Operands are double
        // buffered by hand: the LDS reads of row i+1 are issued BEFORE the MFMAs of row i (the
        // sched_barriers pin that order), so an MFMA never waits on an LDS round trip.
        constexpr int NIT = (CC / 2) * R;
        float a0[S][MR], b0[S][NR], a1[S][MR], b1[S][NR];
        auto load_row = [&](int it, float (&av)[S][MR], float (&bv)[S][NR]) {
            const int cp = it / R, r = it - cp * R;
            const float* wa = wl + ((2 * cp + kk) * RS + r * S) * TMP + li;
            const float* xb = xl + (2 * cp) * t.CHS + r * t.IWp;
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) av[s][mr] = wa[s * TMP + mr * 32];
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) bv[s][nr] = xb[base[nr] + s];
            }
        };
        auto mfma_row = [&](float (&av)[S][MR], float (&bv)[S][NR]) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr)
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            av[s][mr], bv[s][nr], acc[mr][nr], 0, 0, 0);
        };
        load_row(0, a0, b0);
#pragma unroll 1
        for (int it = 0; it < NIT; it += 2) {
            if (it + 1 < NIT) load_row(it + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 2 < NIT) load_row(it + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < NIT) mfma_row(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: lane holds channel (e&3)+8*(e>>2)+4*kk of pixel li for each register e
    const bool partial = t.splits > 1;
    float* dst = partial ? out + (size_t)blockIdx.z * g.N * g.Cs * PQ : out;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            if (!pvalid[nr]) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (m >= g.Cs) continue;
                const size_t idx = opix[nr] + (size_t)m * PQ;
                float v = acc[mr][nr][e];
                if (!partial) {
                    if (bias) v += bias[m];
                    v = bn_apply_act(v, act, slope);
                    if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
                }
                dst[idx] = v;
            }
        }
    }
}

// out = epilogue(sum_z part[z]) for split reductions; fixed summation order
__global__ __launch_bounds__(256) void k_split_epilogue(
    const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
    const float* __restrict__ dact_src, size_t total, int splits, int C, int npix, int act,
    int dact, float slope) {
    const size_t nthreads = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += nthreads) {
        float v = part[i];
        for (int z = 1; z < splits; ++z) v += part[(size_t)z * total + i];
        if (bias) v += bias[(i / npix) % C];
        v = bn_apply_act(v, act, slope);
        if (dact_src) v *= bn_act_grad_from_output(dact_src[i], dact, slope);
        out[i] = v;
    }
}

// the same in 16-byte groups (maps of a multiple of 4 pixels: a group lies in one channel), one group
// per thread: the slabs of all slices are requested before the first add (round 5: the scalar form took
// 6.6 us per call for 12 MB, eight calls per step of a 32-frame shard)
template <int Z>
__global__ __launch_bounds__(256) void k_split_epilogue4(
    const float4* __restrict__ part, const float* __restrict__ bias, float4* __restrict__ out,
    const float4* __restrict__ dact_src, unsigned total4, unsigned C, unsigned npix4, int act, int dact,
    float slope) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 p[Z];
#pragma unroll
    for (int z = 0; z < Z; ++z) p[z] = part[(size_t)z * total4 + i];
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dact_src) d = dact_src[i];
    const float b = bias ? bias[(i / npix4) % C] : 0.f;
    float4 v = p[0];
#pragma unroll
    for (int z = 1; z < Z; ++z) { v.x += p[z].x; v.y += p[z].y; v.z += p[z].z; v.w += p[z].w; }
    v.x = bn_apply_act(v.x + b, act, slope); v.y = bn_apply_act(v.y + b, act, slope);
    v.z = bn_apply_act(v.z + b, act, slope); v.w = bn_apply_act(v.w + b, act, slope);
    if (dact_src) {
        v.x *= bn_act_grad_from_output(d.x, dact, slope); v.y *= bn_act_grad_from_output(d.y, dact, slope);
        v.z *= bn_act_grad_from_output(d.z, dact, slope); v.w *= bn_act_grad_from_output(d.w, dact, slope);
    }
    out[i] = v;
}

int bn_launch_split_epilogue(const float* part, const float* bias, float* out, const float* dact_src,
                             size_t total, int splits, int C, int npix, int act, int dact, float slope,
                             hipStream_t st) {
    const bool al = ((((uintptr_t)part) | ((uintptr_t)out) | ((uintptr_t)dact_src)) & 15u) == 0;
    if (al && (npix & 3) == 0 && (total & 3) == 0 && total / 4 < 0xffffffffull &&
        (splits == 2 || splits == 4 || splits == 8)) {
        const unsigned total4 = (unsigned)(total / 4);
        const dim3 grid((total4 + 255) / 256);
#define BN_SE4(Z) hipLaunchKernelGGL(k_split_epilogue4<Z>, grid, dim3(256), 0, st, (const float4*)part, bias, \
                                     (float4*)out, (const float4*)dact_src, total4, (unsigned)C,               \
                                     (unsigned)(npix / 4), act, dact, slope)
        if (splits == 2) BN_SE4(2); else if (splits == 4) BN_SE4(4); else BN_SE4(8);
#undef BN_SE4
        BN_LAUNCH_CHECK();
        return 0;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_split_epilogue, dim3(blocks), dim3(256), 0, st, part, bias, out, dact_src, total,
                       splits, C, npix, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// planning: plan.variant = (MR<<8)|(NR<<4)|CC... kept explicit in a,b,c,d
//   a = MR, b = NR, c = CC, d = splits
// ---------------------------------------------------------------------------------------------
static bool down_tile(const BnGeom& g, int MR, int NR, int CC, DownTile* t, int* n_wg_xy) {
    const int TP = 128 * NR;
    const bool pow2 = ilog2_exact(g.Ws) >= 0 && ilog2_exact(g.Hs) >= 0;
    // maps that are no powers of two: stride 1 only (round 4; stride 2 has the second generation, the
    // stride-5 layers their own families) -- a tile is F whole frames or PT_H rows of one, rows spread
    // evenly over a frame's tiles, fewer of them until the tile fits LDS and the register budget
    if (!pow2 && g.stride != 1) return false;
    const int PQ = g.Hs * g.Ws;
    if (g.Ws > TP) return false;
    t->F = PQ >= TP ? 1 : TP / PQ;
    int rows = PQ >= TP ? TP / g.Ws : g.Hs;
    t->TMP = 32 * MR + 1;
    t->IWp = g.stride * (g.Ws - 1) + g.S;
    t->IWp += (t->IWp & 1);                 // even row stride
    for (;;) {
        t->tiles_per_frame = (g.Hs + rows - 1) / rows;
        t->PT_H = (g.Hs + t->tiles_per_frame - 1) / t->tiles_per_frame;
        t->IH = g.stride * (t->PT_H - 1) + g.R;
        t->FS = t->IH * t->IWp;
        t->CHS = t->F * t->FS;
        t->xl_floats = (CC * t->CHS + 3) & ~3;
        const size_t lds = ((size_t)t->xl_floats + (size_t)CC * g.R * g.S * t->TMP) * 4;
        const bool fits = lds <= MF_MAX_LDS &&
                          t->CHS <= MF_THREADS * (g.stride == 1 ? 3 : g.stride == 2 ? 6 : 13);   // KIN register budget
        if (fits) break;
        if (pow2) return false;             // (the tiles of power-of-two maps are as measured)
        if (t->F > 1) --t->F;
        else if (rows > 1) --rows;
        else return false;
    }
    t->PTQ = t->PT_H * g.Ws;
    const int groups = (g.N + t->F - 1) / t->F;
    *n_wg_xy = groups * t->tiles_per_frame * ((g.Cs + 32 * MR - 1) / (32 * MR));
    t->splits = 1;
    t->c_per_split = g.Cb;
    return true;
}

BnFastPlan bn_fast_down_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_down_generic", 0, 0, 0, 0, 0, 0};
    // stride 1 (round 3): the first-generation kernel has no stride-specific assumption in its body;
    // 3x3 and 5x5 kernels are instantiated (max-pooling architectures, ae_arch_2.json's last layer
    // through zero-extended taps).  Forward and -- through flipped weights, capi.hip -- data gradient;
    // the weight gradient of these layers stays on im2col + GEMM.
    const bool s1 = g.stride == 1 && g.R == g.S && (g.R == 3 || g.R == 5);
    if (!s1 && (g.R != 5 || g.S != 5)) return p;
    if (!s1 && g.stride != 2 && g.stride != 5) return p;
    // single-channel inputs: conv_edge.hip (stride 2).  Stride 1 (round 4): the first layer of a
    // max-pooling architecture runs here with three of its chunk's four channels masked to zero --
    // four times the arithmetic of a layer that has almost none, against im2col + a GEMM over 4 M rows
    if (g.Cb < 2 && !s1) return p;
    if (g.Cs < 16) return p;
    const int CC = (g.stride == 5) ? 2 : 4;
    if (g.Cs == 16 && g.R == 5 && g.S == 5 && (g.stride == 1 || g.stride == 2) && g.K0 == 0) {
        // round 6: a 16-channel small side on the 16-row MFMA tile (k_down2_m16) -- the 32-row tiles below would
        // multiply 16 rows of zeros; the 256-pixel tile unless the 128-pixel one wastes fewer pixels
        static int off = -1;
        if (off < 0) { const char* e = bn_tune_env("BN_DOWN_M16"); off = (e && e[0] == '0') ? 1 : 0; }
        float fill = 0.f;
        int bn = 0;
        for (int nr = 2; nr >= 1 && !off; --nr) {
            if (!bn_down2_m16_supported(g, nr)) continue;
            const float f = bn_down2_fill(g, 1, nr);
            if (f > fill + 0.02f) { fill = f; bn = nr; }
        }
        if (bn) {
            static const char* const names16[2][2] = {{"k_down2_m16<2, 1>", "k_down2_m16<4, 1>"},
                                                      {"k_down2_m16<2, 2>", "k_down2_m16<4, 2>"}};
            p.supported = true;
            p.a = 0; p.b = bn; p.c = CC; p.d = 1; p.variant = 2;
            p.kernel_name = names16[g.stride - 1][bn - 1];
            const int s2 = bn_down2_splits(g, 1, bn);
            if (s2 > 1) {
                p.d = s2;
                p.ws_bytes = (size_t)s2 * g.N * g.Cs * g.Hs * g.Ws * sizeof(float);
            }
            return p;
        }
    }
    if (g.stride == 1 && g.R == 5 && g.S == 5) {
        // round 4: the streamlined kernel's stride-1 instantiation (16-byte LDS-DMA, double-buffered images, pinned
        // issue order) where its tile serves the map; the tile shape that wastes the fewest pixels
        float fill = 0.f;
        int bi = -1;
        // (32 channels x 256 pixels for layers with fewer than 64 small-side channels: half the weight traffic and
        // half the halo rows per pixel of the 32 x 128 tile)
        static const int cand1[4][2] = {{2, 1}, {2, 2}, {1, 2}, {1, 1}};
        for (int i = 0; i < 4; ++i) {
            if (cand1[i][0] == 2 && g.Cs < 64) continue;
            const float f = bn_down2_fill(g, cand1[i][0], cand1[i][1]);
            if (f > fill + 0.02f) { fill = f; bi = i; }
        }
        if (bi >= 0) {
            static const char* const names2s[4] = {"k_down2_mfma<2, 1, 5, 0, 1>", "k_down2_mfma<2, 2, 5, 0, 1>",
                                                   "k_down2_mfma<1, 2, 5, 0, 1>", "k_down2_mfma<1, 1, 5, 0, 1>"};
            p.supported = true;
            p.a = cand1[bi][0]; p.b = cand1[bi][1]; p.c = CC; p.d = 1; p.variant = 2;
            p.kernel_name = names2s[bi];
            const int s2 = bn_down2_splits(g, p.a, p.b);
            if (s2 > 1) {
                p.d = s2;
                p.ws_bytes = (size_t)s2 * g.N * g.Cs * g.Hs * g.Ws * sizeof(float);
            }
            return p;
        }
    }
    // stride 2: 64 ch x 128 px tiles whenever they give ~100 workgroups -- measured in situ
    // (bench.py, other streams' kernels fill the machine) they beat both the larger 64x256 tiles
    // and the 32x128 tiles with more workgroups; stride 5: 64x128 tiles, the reduction split
    static const int cand[3][2] = {{2, 1}, {2, 2}, {1, 1}};
    const int want = g.stride <= 2 ? 96 : 1;   // stride 5 (K = 6400): the 64x128 tile + split-K
    int best = -1, best_wg = 0;
    DownTile t;
    // whole-batch launches (256 frames) of the layers with 16-pixel or wider small maps: the
    // 64 ch x 256 px tile halves the per-workgroup prologue / epilogue share and still gives
    // >= 2 workgroups per CU (E1 322 -> 290 us, E2 286 -> 260 us at 256 frames; the 8x8 maps of
    // E3 / D1 are faster with 128-pixel tiles)
    if (g.stride == 2 && g.Ws >= 16 && g.Cs >= 64 && !bn_tune_env("BN_DOWN_TILE")) {
        int nwg = 0;
        if (down_tile(g, 2, 2, CC, &t, &nwg) && nwg >= 512) {
            best = 1;
            best_wg = nwg;
        }
    }
    const bool preset = best >= 0;
    for (int i = 0; i < 3 && !preset; ++i) {
        if (cand[i][0] == 2 && g.Cs < 64) continue;
        int nwg = 0;
        if (!down_tile(g, cand[i][0], cand[i][1], CC, &t, &nwg)) continue;
        if (best < 0 || (best_wg < want && nwg > best_wg)) {
            best = i;
            best_wg = nwg;
        }
        if (best_wg >= want) break;
    }
    if (best >= 0) down_tile(g, cand[best][0], cand[best][1], CC, &t, &best_wg);
    // tuning hook (tools/kbench.py): BN_DOWN_TILE=<candidate index 0..2> pins the tile shape
    if (const char* e = bn_tune_env("BN_DOWN_TILE")) {
        const int i = e[0] - '0';
        int nwg = 0;
        if (i >= 0 && i < 3 && !(cand[i][0] == 2 && g.Cs < 64) &&
            down_tile(g, cand[i][0], cand[i][1], CC, &t, &nwg)) {
            best = i;
            best_wg = nwg;
        }
    }
    if (best < 0 && g.stride == 2 && g.R == 5 && g.S == 5) {
        // maps whose sizes are no powers of two (64x48 or 192x160 frames): the second-generation kernel
        // takes any even width and any height; the tile shape that wastes the fewest pixels
        float fill = 0.f;
        for (int i = 0; i < 3; ++i) {
            if (cand[i][0] == 2 && g.Cs < 64) continue;
            const float f = bn_down2_fill(g, cand[i][0], cand[i][1]);
            if (f > fill + 0.02f) { fill = f; best = i; }
        }
        if (best < 0) return p;
        static const char* const names2n[3] = {"k_down2_mfma<2, 1>", "k_down2_mfma<2, 2>", "k_down2_mfma<1, 1>"};
        p.supported = true;
        p.a = cand[best][0]; p.b = cand[best][1]; p.c = CC; p.d = 1; p.variant = 2;
        p.kernel_name = names2n[best];
        const int s2 = bn_down2_splits(g, p.a, p.b);
        if (s2 > 1) {
            p.d = s2;
            p.ws_bytes = (size_t)s2 * g.N * g.Cs * g.Hs * g.Ws * sizeof(float);
        }
        return p;
    }
    if (best < 0) return p;
    p.supported = true;
    p.a = cand[best][0];
    p.b = cand[best][1];
    p.c = CC;
    // split the reduction over channels when the grid cannot fill the chip
    // (stride 2: the second-generation kernel without split is preferred down to ~100
    // workgroups -- the other chunk's and the weight-gradient stream's kernels fill the machine)
    int splits = 1;
    if (best_wg < (g.stride <= 2 ? want : 384)) {
        const int max_splits = g.Cb / (4 * CC) > 0 ? g.Cb / (4 * CC) : 1;
        splits = 512 / best_wg;            // all workgroups resident at once (2 per CU)
        if (splits > max_splits) splits = max_splits;
        if (splits > 16) splits = 16;
        if (splits < 1) splits = 1;
    }
    if (const char* e = bn_tune_env("BN_DOWN_SPLITS")) {       // tuning hook
        const int v = atoi(e);
        if (v >= 1 && v <= 16 && v <= (g.Cb / CC)) splits = v;
    }
    p.d = splits;
    p.ws_bytes = splits > 1 ? (size_t)splits * g.N * g.Cs * g.Hs * g.Ws * sizeof(float) : 0;
    // names as rocprofv3 prints the instantiations (leading template arguments)
    static const char* const names1[2][3] = {
        {"k_down_mfma<2, 1, 4, 2>", "k_down_mfma<2, 2, 4, 2>", "k_down_mfma<1, 1, 4, 2>"},
        {"k_down_mfma<2, 1, 2, 5>", "k_down_mfma<2, 2, 2, 5>", "k_down_mfma<1, 1, 2, 5>"}};
    static const char* const names2[3] = {"k_down2_mfma<2, 1>", "k_down2_mfma<2, 2>",
                                          "k_down2_mfma<1, 1>"};
    static const char* const names_s1[2][3] = {
        {"k_down_mfma<2, 1, 4, 1, 3, 3>", "k_down_mfma<2, 2, 4, 1, 3, 3>", "k_down_mfma<1, 1, 4, 1, 3, 3>"},
        {"k_down_mfma<2, 1, 4, 1, 5, 5>", "k_down_mfma<2, 2, 4, 1, 5, 5>", "k_down_mfma<1, 1, 4, 1, 5, 5>"}};
    p.kernel_name = g.stride == 1 ? names_s1[g.R == 5 ? 1 : 0][best] : names1[g.stride == 2 ? 0 : 1][best];
    if (g.stride == 2 && splits == 1 && CC == 4 && bn_down2_supported(g, p.a, p.b)) {
        p.variant = 2;
        p.kernel_name = names2[best];
    }
    // small batches: the streamlined kernel on the LARGEST tile that fits, its reduction split over
    // workgroups (round 4; the first generation's split path is 20 % slower per FLOP, and the
    // unsplit 32-channel tile ran 128 workgroups of 54 us for a 32-frame shard)
    if (g.stride == 2 && CC == 4 && !bn_tune_env("BN_DOWN_SPLITS") && !bn_tune_env("BN_DOWN_TILE")) {
        for (int i = 0; i < 3; ++i) {
            const int mr = cand[i][0], nr = cand[i][1];
            if (mr == 2 && g.Cs < 64) continue;
            if (!bn_down2_supported(g, mr, nr)) continue;
            const int s2 = bn_down2_splits(g, mr, nr);
            if (s2 > 1) {
                p.a = mr; p.b = nr; p.d = s2; p.variant = 2;
                p.kernel_name = names2[i];
                p.ws_bytes = (size_t)s2 * g.N * g.Cs * g.Hs * g.Ws * sizeof(float);
            }
            break;          // only the first (preferred) tile shape that the kernel serves
        }
    }
    return p;
}

template <int MR, int NR, int CC, int ST>
static int launch_down(const DownTile& t, dim3 grid, size_t lds, const float* big, const float* w,
                       const float* bias, float* out, const float* dact_src, const BnGeom& g,
                       int act, int dact, float slope, hipStream_t st) {
    BN_LAUNCH_MAIN((k_down_mfma<MR, NR, CC, ST, 5, 5, (ST == 2 ? 6 : 13)>), grid,
                       dim3(MF_THREADS), lds, st, big, w, bias, out, dact_src, g, t, act, dact,
                       slope);
    BN_LAUNCH_CHECK();
    return 0;
}

template <int MR, int NR, int R>
static int launch_down_s1(const DownTile& t, dim3 grid, size_t lds, const float* big, const float* w,
                          const float* bias, float* out, const float* dact_src, const BnGeom& g,
                          int act, int dact, float slope, hipStream_t st) {
    BN_LAUNCH_MAIN((k_down_mfma<MR, NR, 4, 1, R, R, 3>), grid, dim3(MF_THREADS), lds, st, big, w,
                       bias, out, dact_src, g, t, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_down_fast(const BnFastPlan& plan, const float* big, const float* w,
                        const float* bias, float* out, const float* dact_src, const BnGeom& g,
                        int act, int dact, float slope, void* ws, hipStream_t st) {
    const int MR = plan.a, NR = plan.b, CC = plan.c, splits = plan.d;
    if (plan.variant == 2)
        return bn_launch_down2(MR, NR, big, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws);
    DownTile t;
    int nwg = 0;
    if (!down_tile(g, MR, NR, CC, &t, &nwg)) return BN_E_SHAPE;
    t.splits = splits;
    static int dbg = -1;
    if (dbg < 0) { const char* e = bn_tune_env("BN_DOWN_DBG"); dbg = e ? atoi(e) : 0; }
    t.dbg = dbg;
    if (splits > 1) {
        int cps = (g.Cb + splits - 1) / splits;
        cps = (cps + CC - 1) / CC * CC;
        t.c_per_split = cps;
    }
    const int groups = (g.N + t.F - 1) / t.F;
    dim3 grid(groups * t.tiles_per_frame, (g.Cs + 32 * MR - 1) / (32 * MR), splits);
    const size_t lds = ((size_t)t.xl_floats + (size_t)CC * g.R * g.S * t.TMP) * 4;
    float* dst = splits > 1 ? (float*)ws : out;
    int rc = BN_E_SHAPE;
#define DOWN_CASE(mr, nr, cc, s)                                                                 \
    if (MR == mr && NR == nr && CC == cc && g.stride == s)                                        \
        rc = launch_down<mr, nr, cc, s>(t, grid, lds, big, w, bias, dst, dact_src, g, act, dact, \
                                        slope, st);
    DOWN_CASE(2, 2, 4, 2) DOWN_CASE(2, 1, 4, 2) DOWN_CASE(1, 1, 4, 2)
    DOWN_CASE(2, 2, 2, 5) DOWN_CASE(2, 1, 2, 5) DOWN_CASE(1, 1, 2, 5)
#undef DOWN_CASE
#define DOWN_S1(mr, nr, r)                                                                       \
    if (g.stride == 1 && CC == 4 && MR == mr && NR == nr && g.R == r)                            \
        rc = launch_down_s1<mr, nr, r>(t, grid, lds, big, w, bias, dst, dact_src, g, act, dact, slope, st);
    DOWN_S1(2, 2, 3) DOWN_S1(2, 1, 3) DOWN_S1(1, 1, 3)
    DOWN_S1(2, 2, 5) DOWN_S1(2, 1, 5) DOWN_S1(1, 1, 5)
#undef DOWN_S1
    if (rc) return rc;
    if (splits > 1) {
        const size_t total = (size_t)g.N * g.Cs * g.Hs * g.Ws;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_split_epilogue, dim3(blocks), dim3(256), 0, st, (const float*)ws, bias,
                           out, dact_src, total, splits, g.Cs, g.Hs * g.Ws, act, dact, slope);
        BN_LAUNCH_CHECK();
    }
    return 0;
}

