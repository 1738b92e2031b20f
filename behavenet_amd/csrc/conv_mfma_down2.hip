// family 1, second generation: "gather-down" (conv forward, transposed-conv data gradient) for
// kernel 5x5, stride 2, left offset 1:
//   out[n,m,p,q] = sum_{c,r,s} big[n,c,2p+r-pt,2q+s-1] * W[m][c][r][s]
//
// Same MFMA roles, tile shapes and epilogue as k_down_mfma (conv_mfma.hip): A = weights (row i =
// output channel), B = input pixels (col j = output pixel), reduction over (tap, channel pair),
// workgroup tile 32*MR channels x 128*NR pixels, chunks of 4 input channels.  What changed is how
// the INPUT tile gets into LDS and out of it again:
//
//  * rows are stored with image column wb at LDS column wb + 4, so a patch row is a run of
//    16-byte groups that are 16-byte aligned in global memory too: the tile of chunk i+1 is
//    copied by buffer_load_dwordx4 ... lds (6-7 per thread, no registers, no ds_write) into the
//    second of two LDS images while chunk i is being multiplied (was: 20-24 dword loads + as
//    many ds_write_b32 per thread and chunk);
//  * a lane needs columns 2q+s-1, s = 0..4: the aligned pairs (2q-2,2q-1) (2q,2q+1) (2q+2,2q+3)
//    -> three ds_read_b64 per kernel row instead of five ds_read_b32 whose stride-2 addresses
//    were 2-way bank conflicted; the row stride is chosen == Q (mod 32) so that the 32 pixels of
//    a half-wave (several image rows when Q < 32) cover the 64 banks exactly once.
//
// The weight slice still goes through registers (global order [m][c][tap] has a 100-word row
// per output channel: no 16-byte-granular copy of it is bank-conflict free for per-lane rows).
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2d __attribute__((ext_vector_type(2)));

#define D2_THREADS 256
#define D2_CC 4
#define D2_X0 4                 // LDS column of image column 0
#define D2_XK 8                 // max 16-byte DMA groups per thread per chunk
#define D2_MAX_LDS (80 * 1024)

static inline int ilog2_exact_d2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

struct Down2Tile {
    int F, PT_H, lgQ, lgPTQ;
    int IH, RW, FS, CHS;          // patch rows per frame, row stride, per-frame / per-channel floats
    int tiles_per_frame;
    int groups;                   // 16-byte groups of one chunk image (D2_CC * CHS / 4)
    int xbuf_floats;              // one LDS input image (whole wave rows)
    float inv_chs4, inv_fs4, inv_c4;
};

template <int MR, int NR>
__global__ __launch_bounds__(D2_THREADS, 2) void k_down2_mfma(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, Down2Tile t, int act,
    int dact, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CC = D2_CC, R = 5, S = 5, RS = 25;
    constexpr int TM = 32 * MR;
    constexpr int TMP = TM + 1;                       // odd row stride: conflict-free transpose
    constexpr int WROWS = TM / 4;                     // weight rows per wave
    constexpr int WPASS = (CC * RS + 63) / 64;        // 64-lane passes along (channel, tap)
    constexpr int WK = WROWS * WPASS;                 // weight loads per thread per chunk
    float* wl = smem + 2 * t.xbuf_floats;
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
        const int pix = 32 * (wv * NR + nr) + li;
        const int f = pix >> t.lgPTQ;
        const int rem = pix & ((1 << t.lgPTQ) - 1);
        const int pj = rem >> t.lgQ, qj = rem & (Q - 1);
        // pair 0 = columns (2q-2, 2q-1) of the image = LDS columns 2q+2, 2q+3
        base[nr] = f * t.FS + (2 * pj) * t.RW + 2 * qj + (D2_X0 - 2) + kk * t.CHS;
        pvalid[nr] = (n0 + f) < g.N;
        opix[nr] = (size_t)(n0 + f) * g.Cs * PQ + (size_t)(p0 + pj) * Q + qj;
    }

    // chunk-invariant part of this thread's DMA groups: byte offset relative to channel c0 of
    // frame n0 (>= 0), or -1 for padding / halo / frame-tail groups (they read 0.0f)
    const __amdgpu_buffer_rsrc_t rbig = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HW * 4), 0x00020000);
    const int C4 = t.RW / 4;
    int xoff[D2_XK];
#pragma unroll
    for (int k = 0; k < D2_XK; ++k) {
        const int e = tid + D2_THREADS * k;
        const int cc = (int)(((float)e + 0.5f) * t.inv_chs4);
        const int within = e - cc * (t.CHS / 4);
        const int f = (int)(((float)within + 0.5f) * t.inv_fs4);
        const int r2 = within - f * (t.FS / 4);
        const int y = (int)(((float)r2 + 0.5f) * t.inv_c4);
        const int c4 = r2 - y * C4;
        const int hb = 2 * p0 - g.pt + y, wb = 4 * c4 - D2_X0;
        const bool ok = e < t.groups && (n0 + f < g.N) && hb >= 0 && hb < g.Hb && wb >= 0 &&
                        wb < g.Wb;
        xoff[k] = ok ? ((f * g.Cb + cc) * HW + hb * g.Wb + wb) * 4 : -1;
    }

    floatx16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    float wr[WK];
    auto issue_loads = [&](int c0, int buf) {
        // input tile: 16-byte LDS-DMA straight into image `buf`
        const int cbase = (n0 * g.Cb + c0) * HW * 4;
#pragma unroll
        for (int k = 0; k < D2_XK; ++k) {
            if (D2_THREADS * k + 64 * wv < t.groups)              // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rbig, smem + buf * t.xbuf_floats + 4 * (D2_THREADS * k + 64 * wv), 16,
                    xoff[k] >= 0 ? cbase + xoff[k] : 0x7fffffff, 0, 0, 0);
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
                wr[k * WPASS + ps] = rp[(r2 < CC * RS) ? 64 * ps : -lane];
            }
        }
    };

    int cur = 0;
    issue_loads(0, 0);
    for (int c0 = 0; c0 < g.Cb; c0 += CC) {
        __syncthreads();   // the previous chunk's MFMA reads of wl (and of image cur^1) are done
#pragma unroll
        for (int k = 0; k < WROWS; ++k) {
            const bool mok = m0 + wv + 4 * k < g.Cs;
#pragma unroll
            for (int ps = 0; ps < WPASS; ++ps) {
                const int r2 = lane + 64 * ps;
                if (r2 < CC * RS) wl[r2 * TMP + wv + 4 * k] = mok ? wr[k * WPASS + ps] : 0.f;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA groups of this chunk landed
        __syncthreads();
        if (c0 + CC < g.Cb) issue_loads(c0 + CC, cur ^ 1);   // in flight behind the MFMAs below
        const float* xcur = smem + cur * t.xbuf_floats;

        // MFMA loop, one (channel pair, kernel row) = 5 taps per "row"; operands double buffered
        // by hand: the LDS reads of row i+1 are issued BEFORE the MFMAs of row i
        constexpr int NIT = (CC / 2) * R;
        float a0[S][MR], b0[S][NR], a1[S][MR], b1[S][NR];
        auto load_row = [&](int it, float (&av)[S][MR], float (&bv)[S][NR]) {
            const int cp = it / R, r = it - cp * R;
            const float* wa = wl + ((2 * cp + kk) * RS + r * S) * TMP + li;
            const float* xb = xcur + (2 * cp) * t.CHS + r * t.RW;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) av[s][mr] = wa[s * TMP + mr * 32];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                const floatx2d c0p = *reinterpret_cast<const floatx2d*>(xb + base[nr]);
                const floatx2d c1p = *reinterpret_cast<const floatx2d*>(xb + base[nr] + 2);
                const floatx2d c2p = *reinterpret_cast<const floatx2d*>(xb + base[nr] + 4);
                bv[0][nr] = c0p.y; bv[1][nr] = c1p.x; bv[2][nr] = c1p.y;
                bv[3][nr] = c2p.x; bv[4][nr] = c2p.y;
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
        cur ^= 1;
    }

    // ---- epilogue: lane holds channel (e&3)+8*(e>>2)+4*kk of pixel li for each register e
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
                if (bias) v += bias[m];
                v = bn_apply_act(v, act, slope);
                if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
                out[idx] = v;
            }
        }
    }
}

static bool down2_tile(const BnGeom& g, int MR, int NR, Down2Tile* t, size_t* lds_bytes) {
    if (g.R != 5 || g.S != 5 || g.stride != 2 || g.pl != 1 || g.pt < 0) return false;
    if ((g.Cb % D2_CC) != 0 || (g.Wb & 3) != 0) return false;
    const int TP = 128 * NR;
    const int lgQ = ilog2_exact_d2(g.Ws), lgP = ilog2_exact_d2(g.Hs);
    if (lgQ < 2 || lgP < 0 || g.Ws > TP) return false;
    const int PQ = g.Hs * g.Ws;
    if (PQ >= TP) {
        t->F = 1;
        t->PT_H = TP / g.Ws;
    } else {
        t->F = TP / PQ;
        t->PT_H = g.Hs;
    }
    t->lgQ = lgQ;
    t->lgPTQ = ilog2_exact_d2(t->PT_H * g.Ws);
    t->IH = 2 * (t->PT_H - 1) + 5;
    int rw = 2 * g.Ws + 8;
    if (g.Ws < 32)
        while ((rw & 31) != (g.Ws & 31)) rw += 4;        // half-wave rows cover the 64 banks once
    t->RW = rw;
    t->FS = t->IH * rw;
    t->CHS = t->F * t->FS;
    t->tiles_per_frame = (t->F == 1) ? g.Hs / t->PT_H : 1;
    t->groups = D2_CC * t->CHS / 4;
    if (t->groups > D2_THREADS * D2_XK) return false;
    t->xbuf_floats = 4 * ((t->groups + 63) & ~63);
    t->inv_chs4 = 1.0f / (float)(t->CHS / 4);
    t->inv_fs4 = 1.0f / (float)(t->FS / 4);
    t->inv_c4 = 1.0f / (float)(rw / 4);
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    *lds_bytes = ((size_t)2 * t->xbuf_floats + (size_t)D2_CC * 25 * (32 * MR + 1)) * 4;
    return *lds_bytes <= D2_MAX_LDS;
}

bool bn_down2_supported(const BnGeom& g, int MR, int NR) {
    static int disabled = -1;                      // BN_DOWN2=0: first-generation kernel only
    if (disabled < 0) { const char* e = bn_tune_env("BN_DOWN2"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    Down2Tile t;
    size_t lds = 0;
    return down2_tile(g, MR, NR, &t, &lds);
}

template <int MR, int NR>
static int launch_down2(const Down2Tile& t, dim3 grid, size_t lds, const float* big, const float* w,
                        const float* bias, float* out, const float* dact_src, const BnGeom& g,
                        int act, int dact, float slope, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_down2_mfma<MR, NR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, D2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_down2_mfma<MR, NR>), grid, dim3(D2_THREADS), lds, st, big, w, bias, out,
                       dact_src, g, t, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_down2(int MR, int NR, const float* big, const float* w, const float* bias,
                    float* out, const float* dact_src, const BnGeom& g, int act, int dact,
                    float slope, hipStream_t st) {
    Down2Tile t;
    size_t lds = 0;
    if (!down2_tile(g, MR, NR, &t, &lds)) return BN_E_SHAPE;
    const int groups = (g.N + t.F - 1) / t.F;
    dim3 grid(groups * t.tiles_per_frame, (g.Cs + 32 * MR - 1) / (32 * MR));
    if (MR == 2 && NR == 2)
        return launch_down2<2, 2>(t, grid, lds, big, w, bias, out, dact_src, g, act, dact, slope, st);
    if (MR == 2 && NR == 1)
        return launch_down2<2, 1>(t, grid, lds, big, w, bias, out, dact_src, g, act, dact, slope, st);
    if (MR == 1 && NR == 1)
        return launch_down2<1, 1>(t, grid, lds, big, w, bias, out, dact_src, g, act, dact, slope, st);
    return BN_E_SHAPE;
}
