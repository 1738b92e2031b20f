// family 2: "gather-up" on the matrix cores (transposed-conv forward, conv data gradient),
// kernel 5x5, stride 2, offset (1,1):
//   out[n,m,h,w] = sum_{c,r,s} small[n,c,p,q] * W[c][m][r][s],   2p + r = h + 1,  2q + s = w + 1
//
// Output pixels split into four parity classes (h&1, w&1); within a class the op is a stride-1
// correlation of the small image with a 3x3 / 3x2 / 2x3 / 2x2 sub-kernel (9+6+6+4 = 25 taps, no
// wasted MACs).  A wave owns 32 positions (a,b) of the small image -- i.e. a 2x2 output block
// per lane and 128 output pixels per wave -- and all four classes, so the nine neighbour values
// small[c][a+dy][b+dx] (dy,dx in -1..1) are read from LDS once and shared by the classes:
// per channel pair 9 B-reads + 25*MR A-reads feed 25*MR MFMAs (v_mfma_f32_32x32x2_f32).
// The two column classes of a lane are adjacent in memory, so the epilogue stores float2.
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define MF_THREADS 256
#define MF_MAX_LDS (64 * 1024)

static inline int ilog2_exact_up(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

struct UpTile {
    int F, AT_H, lgW, lgATW;      // frames / small-image rows per workgroup tile, log2 sizes
    int SWp, FS, CHS;             // LDS strides of the small tile (floats)
    int tiles_per_frame;
    int xl_floats;
    int dbg;              // BN_UP_DBG experiments (0 in production)
};

template <int MR, int CC>
__global__ __launch_bounds__(MF_THREADS, 2) void k_up_mfma(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, UpTile t, int act,
    int dact, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xl = smem;
    float* wl = smem + t.xl_floats;
    constexpr int RS = 25;
    constexpr int TM = 32 * MR;
    // Weight slice of a chunk: for each of the CC channels the TM*25 floats W[c][m0..m0+TM)[tap]
    // are contiguous in global memory and 16-byte aligned (Cb, m0 multiples of 4); they are
    // copied AS THEY ARE into LDS by buffer_load_dwordx4 ... lds (no registers, no transpose:
    // a lane's A operand is word m*25 + tap, and a stride of 25 words over 32 lanes touches
    // every bank exactly once).  Double buffered: chunk i+1 lands behind chunk i's MFMAs.
    constexpr int WCH = TM * RS;                        // floats per channel
    constexpr int WGRP = CC * WCH / 4;                  // 16-byte groups per chunk
    constexpr int WDMA = (WGRP + MF_THREADS - 1) / MF_THREADS;
    constexpr int WBUF = ((WDMA * MF_THREADS * 4) + 3) & ~3;   // floats per buffer (whole waves)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    const int grp = blockIdx.x / t.tiles_per_frame;
    const int rowt = blockIdx.x - grp * t.tiles_per_frame;
    const int n0 = grp * t.F;
    const int a0 = rowt * t.AT_H;
    const int m0 = blockIdx.y * TM;
    const int Ws = g.Ws, HWs = g.Hs * g.Ws;

    // lane -> position (frame f, row aj, col bj) of the small image
    const int pos = 32 * wv + li;
    const int pf = pos >> t.lgATW;
    const int prem = pos & ((1 << t.lgATW) - 1);
    const int aj = prem >> t.lgW, bj = prem & (Ws - 1);
    const int base = pf * t.FS + (aj + 1) * t.SWp + (bj + 1) + kk * t.CHS;
    const bool pvalid = (n0 + pf) < g.N;

    // this thread's element of the small tile (CHS <= 256): chunk-invariant gather offset
    int ioff = -2;
    if (tid < t.CHS) {
        const int f = tid / t.FS;
        const int r2 = tid - f * t.FS;
        const int y = r2 / t.SWp;
        const int x = r2 - y * t.SWp;
        const int p = a0 - 1 + y, q = x - 1;
        const bool ok = (n0 + f < g.N) && p >= 0 && p < g.Hs && q >= 0 && q < g.Ws;
        ioff = ok ? f * (g.Cs * HWs) + p * Ws + q : -1;
    }

    floatx16 acc[MR][4];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][cl][e] = 0.f;

    float xr[CC];
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * g.Cb * RS * 4), 0x00020000);

    auto issue_loads = [&](int c0, int buf) {
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int c = min(c0 + cc, g.Cs - 1);
            xr[cc] = small[((size_t)n0 * g.Cs + c) * HWs + max(ioff, 0)];
        }
        // group e = tid + 256 k of the chunk image [cc][m][tap]; a channel past Cs (or a row
        // past the weight tensor) is out of range and reads 0.0f
#pragma unroll
        for (int k = 0; k < WDMA; ++k) {
            const int e = tid + MF_THREADS * k;
            const int cc = e / (WCH / 4);
            const int within = e - cc * (WCH / 4);
            const bool ok = (e < WGRP) && (c0 + cc < g.Cs);
            const int off = (((c0 + cc) * g.Cb + m0) * RS + 4 * within) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_w, wl + buf * WBUF + 4 * (MF_THREADS * k + 64 * wv), 16, ok ? off : 0x7fffffff,
                0, 0, 0);
        }
    };

    int cur = 0;
    issue_loads(0, 0);
    for (int c0 = 0; c0 < g.Cs; c0 += CC) {
        __syncthreads();   // the previous chunk's MFMA reads of xl are complete
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const bool cok = c0 + cc < g.Cs;
            if (ioff != -2) xl[cc * t.CHS + tid] = (cok && ioff >= 0) ? xr[cc] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own weight DMA of this chunk landed
        __syncthreads();
        if (c0 + CC < g.Cs) issue_loads(c0 + CC, cur ^ 1);
        const float* wcur = wl + cur * WBUF;

#pragma unroll 1
        for (int cp = 0; cp < CC / 2; ++cp) {
            const float* xb = xl + (2 * cp) * t.CHS + base;
            const float* wa = wcur + ((2 * cp + kk) * TM + li) * RS;
            float bv[3][3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) bv[dy][dx] = xb[(dy - 1) * t.SWp + (dx - 1)];
            // classes (rho, sigma) = output (row, col) parity; with offset 1:
            //   rho = 0: r in {1,3}, dy = -u      rho = 1: r in {0,2,4}, dy = 1 - u
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int r = ((rho + 1) & 1) + 2 * u;
                    if (r >= 5) continue;
                    const int dy = ((rho + 1) >> 1) - u + 1;      // index into bv
#pragma unroll
                    for (int sig = 0; sig < 2; ++sig) {
#pragma unroll
                        for (int v = 0; v < 3; ++v) {
                            const int s = ((sig + 1) & 1) + 2 * v;
                            if (s >= 5) continue;
                            const int dx = ((sig + 1) >> 1) - v + 1;
#pragma unroll
                            for (int mr = 0; mr < MR; ++mr) {
                                const float av = wa[mr * 32 * RS + r * 5 + s];
                                acc[mr][rho * 2 + sig] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                    av, bv[dy][dx], acc[mr][rho * 2 + sig], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        cur ^= 1;
    }

    // ---- epilogue: lane owns output pixels (2a+rho, 2b+sig) of channel m(e, kk); the two
    // column classes are adjacent -> one float2 store per (channel, row class)
    if (!pvalid) return;
    const int n = n0 + pf;
    const int Wb = g.Wb, HWb = g.Hb * g.Wb;
    const int h0 = 2 * (a0 + aj), w0 = 2 * bj;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (m >= g.Cb) continue;
            const float bm = bias ? bias[m] : 0.f;
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) {
                const size_t idx = ((size_t)n * g.Cb + m) * HWb + (size_t)(h0 + rho) * Wb + w0;
                float2 v;
                v.x = bn_apply_act(acc[mr][rho * 2 + 0][e] + bm, act, slope);
                v.y = bn_apply_act(acc[mr][rho * 2 + 1][e] + bm, act, slope);
                if (dact_src) {
                    const float2 d = *reinterpret_cast<const float2*>(dact_src + idx);
                    v.x *= bn_act_grad_from_output(d.x, dact, slope);
                    v.y *= bn_act_grad_from_output(d.y, dact, slope);
                }
                *reinterpret_cast<float2*>(out + idx) = v;
            }
        }
    }
}

// xl + two weight buffers of whole-wave DMA rows (see WBUF in the kernel)
static size_t up_lds_bytes(int xl_floats, int MR, int CC) {
    const int wgrp = CC * 32 * MR * 25 / 4;
    const int wdma = (wgrp + MF_THREADS - 1) / MF_THREADS;
    return ((size_t)xl_floats + 2 * (size_t)wdma * MF_THREADS * 4) * 4;
}

static bool up_tile(const BnGeom& g, int MR, int CC, UpTile* t, int* n_wg) {
    const int TP = 128;   // positions per workgroup
    const int lgW = ilog2_exact_up(g.Ws), lgH = ilog2_exact_up(g.Hs);
    if (lgW < 0 || lgH < 0) return false;
    if (g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return false;
    const int HW = g.Hs * g.Ws;
    if (g.Ws > TP) return false;
    if (HW >= TP) {
        t->F = 1;
        t->AT_H = TP / g.Ws;
    } else {
        t->F = TP / HW;
        t->AT_H = g.Hs;
    }
    t->lgW = lgW;
    t->lgATW = ilog2_exact_up(t->AT_H * g.Ws);
    t->SWp = g.Ws + 2;
    t->FS = (t->AT_H + 2) * t->SWp;
    t->CHS = t->F * t->FS;
    if (t->CHS > MF_THREADS) return false;          // one tile element per thread
    t->tiles_per_frame = (t->F == 1) ? g.Hs / t->AT_H : 1;
    t->xl_floats = (CC * t->CHS + 3) & ~3;
    const size_t lds = up_lds_bytes(t->xl_floats, MR, CC);
    if (lds > MF_MAX_LDS) return false;
    if ((g.Cb & 3) != 0) return false;               // 16-byte aligned weight rows for the DMA
    if ((size_t)g.Cs * g.Cb * 25 * 4 >= 0x7fffffffull) return false;
    const int groups = (g.N + t->F - 1) / t->F;
    *n_wg = groups * t->tiles_per_frame * ((g.Cb + 32 * MR - 1) / (32 * MR));
    return true;
}

BnFastPlan bn_fast_up_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_up_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || g.stride != 2 || g.pt != 1 || g.pl != 1) return p;
    if (g.Cb < 16 || g.Cs < 2) return p;
    UpTile t;
    int nwg2 = 0, nwg1 = 0;
    static int env_mr = -1, env_cc = -1;          // tuning hooks: BN_UP_MR=1|2, BN_UP_CC=4|8
    if (env_mr < 0) { const char* e = bn_tune_env("BN_UP_MR"); env_mr = e ? atoi(e) : 0; }
    if (env_cc < 0) { const char* e = bn_tune_env("BN_UP_CC"); env_cc = e ? atoi(e) : 0; }
    const int cc = (env_cc == 8 && (g.Cs % 8) == 0) ? 8 : 4;
    const bool ok2 = g.Cb >= 64 && up_tile(g, 2, cc, &t, &nwg2);
    const bool ok1 = up_tile(g, 1, cc, &t, &nwg1);
    if (!ok1 && !ok2) return p;
    p.supported = true;
    p.a = (ok2 && (nwg2 >= 768 || !ok1)) ? 2 : 1;
    if (env_mr == 2 && ok2) p.a = 2;
    if (env_mr == 1 && ok1) p.a = 1;
    p.c = cc;
    p.kernel_name = p.a == 2 ? (cc == 8 ? "k_up_mfma<2, 8>" : "k_up_mfma<2, 4>")
                             : (cc == 8 ? "k_up_mfma<1, 8>" : "k_up_mfma<1, 4>");
    return p;
}

int bn_launch_up_fast(const BnFastPlan& plan, const float* small, const float* w,
                      const float* bias, float* out, const float* dact_src, const BnGeom& g,
                      int act, int dact, float slope, void* ws, hipStream_t st) {
    (void)ws;
    const int MR = plan.a, CC = plan.c;
    UpTile t;
    int nwg = 0;
    if (!up_tile(g, MR, CC, &t, &nwg)) return BN_E_SHAPE;
    static int dbg = -1;
    if (dbg < 0) { const char* e = bn_tune_env("BN_UP_DBG"); dbg = e ? atoi(e) : 0; }
    t.dbg = dbg;
    const int groups = (g.N + t.F - 1) / t.F;
    dim3 grid(groups * t.tiles_per_frame, (g.Cb + 32 * MR - 1) / (32 * MR));
    const size_t lds = up_lds_bytes(t.xl_floats, MR, CC);
    if (MR == 2 && CC == 8) {
        hipLaunchKernelGGL((k_up_mfma<2, 8>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else if (MR == 2) {
        hipLaunchKernelGGL((k_up_mfma<2, 4>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else if (CC == 8) {
        hipLaunchKernelGGL((k_up_mfma<1, 8>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else {
        hipLaunchKernelGGL((k_up_mfma<1, 4>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    }
    BN_LAUNCH_CHECK();
    return 0;
}
