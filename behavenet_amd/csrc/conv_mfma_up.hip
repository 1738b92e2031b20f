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
#ifndef UP2_ST_AUX
#define UP2_ST_AUX 2        // cache policy of the 8-byte output stores of k_up2_mfma: nt (streaming).  In the
                            // training step dec.convT3 fwd (134 MB of output) 224 -> 206 us, step 4.42 -> 4.39 ms;
                            // sc1 (16): no change; plain (0): the output evicts the weights and tiles from the L2s
#endif
#ifndef UP2_ST_AUX_D
#define UP2_ST_AUX_D UP2_ST_AUX   // the same for the data-gradient epilogue (dact_src given), whose output
                                  // is read back by the very next kernel (the layer's weight gradient)
#endif


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

// ---------------------------------------------------------------------------------------------
// Streamlined variant for square small images of 8x8 / 16x16 / 32x32 (the bench layers; LGW = 3 / 4 / 5) and,
// since round 4, for any other small map up to 83 columns wide (LGW = 0: tile geometry in a kernel
// argument, see UP2<0> below), 32 output channels per workgroup.  Same tiling, MFMA roles and summation order as k_up_mfma<1, CC> above
// (results are bit-identical); what changed is everything around the MFMAs:
//  * a VALU instruction that is not an MFMA costs the matrix pipe 6-13 cycles on this chip, and a
//    wave that is NOT multiplying cannot issue vector-memory instructions at all while another
//    wave of its SIMD streams MFMAs (tools/lab/issue_probe.hip, coissue_probe.hip).  The kernel
//    above spends ~1.5 VALU instructions per MFMA (operand addresses, the group decode of the
//    weight DMA, the register round trip of the input tile) and issues its loads between two
//    barriers, outside the MFMA stream;
//  * here the tile geometry is a compile-time function of LGW, so an operand read is
//    `ds_read_b32 vbase offset:imm`; the input tile arrives by dword LDS-DMA (halo and padding
//    = out-of-range reads = 0.0f) next to the 16-byte weight DMA, both double buffered, both with
//    per-thread offsets computed once plus a scalar chunk offset, and both issued from INSIDE the
//    MFMA stream of the previous chunk; a chunk boundary is one s_waitcnt and one barrier;
//  * the operand reads of channel pair i+1 sit between the MFMAs of pair i (A values refresh in
//    place, an MFMA has taken its operands when it issues); sched_barrier pins the order;
//  * the epilogue batches its loads and is branch-free (buffer stores with out-of-range lanes).
// ---------------------------------------------------------------------------------------------
#define UP2_STR2(x) #x
#define UP2_STR(x) UP2_STR2(x)
// Placement of k_up2_mfma's chunk loop in the code object: `.p2align UP2_LOOP_ALIGN` plus UP2_LOOP_SHIFT s_nop in front of
// it.  The loop is the same instruction stream wherever it lies, but not the same speed: the 16x16-map instantiation
// (dec.convT2 forward, enc.conv2's data gradient) ran 206-239 us in the training step depending on the shift alone --
// it had gone from 202 to 232 us when two kernel arguments were added in front of it (the small-batch reduction split)
// and nothing in the loop changed (ISA diffed, round 4).  A sweep of the shift in steps of 32 bytes inside the step
// (tools/ab_layers.sh): 0: 227, 32 B: 237, 64: 228, 96: 208, 128: 207, 160: 217, 192: 218, 224: 206 us; the 32x32-map
// instantiation 213 -> 204.  256-byte alignment + 32 s_nop = 128 bytes: the six launches of the step 1296 -> 1218 us.
// k_down2_mfma and k_wgrad4s_mfma have the same hook (D2_LOOP_ALIGN, W4_LOOP_ALIGN) and did not react to it.
#ifndef UP2_LOOP_ALIGN
#define UP2_LOOP_ALIGN 8
#endif
#ifndef UP2_LOOP_SHIFT
#define UP2_LOOP_SHIFT 32
#endif
#ifndef UP2_WGS
#define UP2_WGS 2      // min workgroups per CU for the register budget (4 = 128 VGPRs measured no faster)
#endif
template <int LGW>
struct UP2 {
    static constexpr int Ws = 1 << LGW, HW = Ws * Ws, TP = 128;
    static constexpr int F = HW >= TP ? 1 : TP / HW;
    static constexpr int AT_H = HW >= TP ? TP / Ws : Ws;
    static constexpr int lgATW = (HW >= TP) ? 7 : 2 * LGW;          // log2(AT_H * Ws)
    static constexpr int SWp = Ws + 2, FS = (AT_H + 2) * SWp, CHS = F * FS;
    static constexpr int CHSP = (CHS + 63) / 64 * 64;                // channel stride in LDS
    static constexpr int TPF = (F == 1) ? Ws / AT_H : 1;             // tiles per frame
};

// LGW == 0: the tile geometry is a RUNTIME function of the map (any width up to 83 columns, any
// height: 64x48 or 192x160 frames).  One tile element per thread as above (CHS <= 256 = CHSP); what
// cannot be an immediate any more is the row stride of the LDS image, so the nine neighbour reads of
// a lane go through three row base registers instead of one (nothing else enters the MFMA stream).
// A tile is F whole frames or AT_H rows of one; positions past F * AT_H * Ws, and rows of a frame's
// last tile below the map, multiply the tile's first position and store nothing.
template <>
struct UP2<0> {
    static constexpr int CHSP = 256;
};
// LGW == 1 (round 6): the same with room for 320 elements -- eight 4x4 maps with their halos are 288, ten 4x3 maps 300;
// the elements past 256 are copied by wave 0's second DMA of a channel.  Its own instantiation, chosen only where it
// fills the tile better: with the four extra slots in the DMA schedule the 256-element tiles of 192x160 frames ran
// 2.7 % slower (A/B on one box).
template <>
struct UP2<1> {
    static constexpr int CHSP = 320;
};
struct Up2Geo {
    int F, AT_H, ATW;             // UNITS per tile, rows per unit, AT_H * Ws
    int SWp, FS, CHS;             // LDS row / unit strides of the small tile, elements of a channel
    int UPF;                      // units (row blocks) per frame: a tile holds F units -- whole frames (UPF = 1),
                                  // the row blocks of one frame (F = 1) or row blocks of adjacent frames
};

__device__ __forceinline__ void up_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ void up_dma4(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 4, voffset, soffset, 0, 0);
}

template <int KV>
__device__ __forceinline__ constexpr bool up2_tap_live(int r, int s) {
    return r >= KV / 10 && r < KV % 10 && s >= KV / 10 && s < KV % 10;
}

// KV = 4: taps with r >= 4 or s >= 4 are zeros (BnGeom::KV, a 4x4 layer on 5x5 taps): neither read nor
// multiplied -- 16 of the 25 products of a channel pair; the slots of the issue order keep their places.
// KV = 14: row 0 / column 0 are zeros as well (BnGeom::K0 = 1, a 3x3 layer embedded at (1, 1)): 9 products
template <int LGW, int CC, int KV>
__global__ __launch_bounds__(MF_THREADS, UP2_WGS) void k_up2_mfma(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, int act, int dact,
    float slope, int cper, size_t zstride, Up2Geo tg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using T = UP2<LGW>;
    constexpr bool RT = LGW <= 1;                         // runtime tile geometry (tg)
    // reduction split over workgroups (small batches: gridDim.z slices of cper input channels, raw
    // sums into slab blockIdx.z of the scratch, finished by k_split_epilogue); gridDim.z == 1: all
    const int c_beg = blockIdx.z * cper;
    const int c_end = min(g.Cs, c_beg + cper);
    out += blockIdx.z * zstride;
    constexpr int RS = 25, TM = 32;
    const int Ws = RT ? g.Ws : (1 << LGW), HWs = RT ? g.Hs * g.Ws : (1 << (2 * LGW)), SWp = RT ? tg.SWp : Ws + 2;
    constexpr int XBUF = CC * T::CHSP;                    // floats per input image
    constexpr int WCH = TM * RS;                          // weight floats per channel
    constexpr int WGRP = CC * WCH / 4;                    // 16-byte groups per chunk
    constexpr int WDMA = (WGRP + MF_THREADS - 1) / MF_THREADS;
    constexpr int WBUF = WDMA * MF_THREADS * 4;           // floats per weight image (whole waves)
    constexpr int XW = T::CHSP > MF_THREADS ? MF_THREADS / 64 : T::CHSP / 64;   // waves that copy input elements
    constexpr int OOB = 0x7fffffff;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    int grp, rowt, n0, a0, pf, aj, bj, tFS, tCHS;
    bool pvalid;
    const int m0 = blockIdx.y * TM;
    // lane -> position (frame f, row aj, col bj) of the small image
    int u0 = 0, lidx = 0;                                 // RT: first unit of the tile; the lane's slot in LDS
    if constexpr (RT) {
        grp = rowt = 0;
        u0 = blockIdx.x * tg.F;
        n0 = u0 / tg.UPF;                                 // frame of the first unit: scalar part of the offsets
        int pos = 32 * wv + li;
        const bool inside = pos < tg.F * tg.ATW;
        if (!inside) pos = 0;
        const int uf = pos / tg.ATW;                      // unit inside the tile -> its frame and first row
        const int prem = pos - uf * tg.ATW;
        const int un = (u0 + uf) / tg.UPF;
        a0 = (u0 + uf - un * tg.UPF) * tg.AT_H;
        aj = prem / Ws;
        bj = prem - aj * Ws;
        pvalid = inside && un < g.N && (a0 + aj) < g.Hs;
        pf = un - n0;                                     // frames behind n0 (the epilogue's n = n0 + pf)
        tFS = tg.FS; tCHS = tg.CHS;
        lidx = uf;                                        // (its unit's index in the tile, not its frame)
    } else {
        grp = blockIdx.x / T::TPF;
        rowt = blockIdx.x - grp * T::TPF;
        n0 = grp * T::F;
        a0 = rowt * T::AT_H;
        const int pos = 32 * wv + li;
        pf = pos >> T::lgATW;
        const int prem = pos & ((1 << T::lgATW) - 1);
        aj = prem >> LGW; bj = prem & (Ws - 1);
        pvalid = (n0 + pf) < g.N;
        tFS = T::FS; tCHS = T::CHS;
        lidx = pf;
    }
    const int Hs_rt = RT ? g.Hs : Ws;

    // ---- DMA descriptors (chunk independent) ---------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * HWs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * g.Cb * RS * 4), 0x00020000);
    auto tile_elem = [&](const int el) {                  // element el of a channel's tile -> byte offset or OOB
        if (el >= tCHS) return OOB;
        const int f = el / tFS, r2 = el - f * tFS;
        const int y = r2 / SWp, x = r2 - y * SWp;
        int fn = f, fa0 = a0;                             // frame behind n0 and first row of element's unit
        if constexpr (RT) {
            const int un = (u0 + f) / tg.UPF;
            fn = un - n0;
            fa0 = (u0 + f - un * tg.UPF) * tg.AT_H;
        }
        const int p = fa0 - 1 + y, q = x - 1;
        const bool ok = (n0 + fn < g.N) && p >= 0 && p < Hs_rt && q >= 0 && q < Ws;
        return ok ? (fn * (g.Cs * HWs) + p * Ws + q) * 4 : OOB;
    };
    const int xvo = tile_elem(tid);                       // element tid of a channel's tile
    constexpr int NX2 = T::CHSP > MF_THREADS ? CC : 0;    // second DMA of a channel: elements 256 .. CHSP - 1 (wave 0)
    int xvo2 = OOB;
    if constexpr (NX2 > 0) xvo2 = tile_elem(tid + MF_THREADS);
    int wvo[WDMA];                                        // group tid + 256 k of [cc][m][tap]
#pragma unroll
    for (int k = 0; k < WDMA; ++k) {
        const int e = tid + MF_THREADS * k;
        const int cc = e / (WCH / 4), within = e - cc * (WCH / 4);
        wvo[k] = (e < WGRP) ? ((cc * g.Cb + m0) * RS + 4 * within) * 4 : OOB;
    }
    // DMA instruction d of chunk c0 into image pair `buf`: d < CC input channel d, else weight slice
    auto issue_dma = [&](const int d, const int buf, const int c0) __attribute__((always_inline)) {
        if (d < CC) {
            if (wv < XW)
                up_dma4(rs_x, smem + buf * XBUF + d * T::CHSP + 64 * wv, xvo,
                        (n0 * g.Cs + c0 + d) * HWs * 4);
        } else if (d < CC + NX2) {
            if (wv == 0 && tCHS > MF_THREADS)             // (uniform: most tiles end below element 256)
                up_dma4(rs_x, smem + buf * XBUF + (d - CC) * T::CHSP + MF_THREADS, xvo2,
                        (n0 * g.Cs + c0 + d - CC) * HWs * 4);
        } else {
            const int k = d - CC - NX2;
            up_dma16(rs_w, smem + 2 * XBUF + buf * WBUF + 4 * (MF_THREADS * k + 64 * wv), wvo[k],
                     c0 * g.Cb * RS * 4);
        }
    };
    constexpr int NDMA = CC + NX2 + WDMA;

    floatx16 acc[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[cl][e] = 0.f;

    // ---- operand read bases (byte offsets from smem), one per LDS image --------------------------
    int xb_cur = (lidx * tFS + (aj + 1) * SWp + (bj + 1) + kk * T::CHSP) * 4;
    int xb_oth = xb_cur + XBUF * 4;
    int wa_cur = (2 * XBUF + (kk * TM + li) * RS) * 4;
    int wa_oth = wa_cur + WBUF * 4;
    asm volatile("" : "+v"(xb_cur)); asm volatile("" : "+v"(xb_oth));
    asm volatile("" : "+v"(wa_cur)); asm volatile("" : "+v"(wa_oth));
    // runtime row stride: the rows above / below have their own base registers
    int xu_cur = xb_cur - SWp * 4, xd_cur = xb_cur + SWp * 4;
    int xu_oth = xb_oth - SWp * 4, xd_oth = xb_oth + SWp * 4;
    if constexpr (RT) {
        asm volatile("" : "+v"(xu_cur)); asm volatile("" : "+v"(xd_cur));
        asm volatile("" : "+v"(xu_oth)); asm volatile("" : "+v"(xd_oth));
    }
    const char* sm = reinterpret_cast<const char*>(smem);

    // tap list of a channel pair in the order of the MFMAs (classes (rho, sigma) = output parity):
    //   rho = 0: r in {1,3}, dy = -u      rho = 1: r in {0,2,4}, dy = 1 - u   (same for columns)
    struct Tap { int r, s, dy, dx, cl; };
    auto tap_of = [](const int j) {
        int n = 0;
        Tap tp = {0, 0, 0, 0, 0};
        for (int rho = 0; rho < 2; ++rho)
            for (int u = 0; u < 3; ++u) {
                const int r = ((rho + 1) & 1) + 2 * u;
                if (r >= 5) continue;
                for (int sig = 0; sig < 2; ++sig)
                    for (int v = 0; v < 3; ++v) {
                        const int s = ((sig + 1) & 1) + 2 * v;
                        if (s >= 5) continue;
                        if (n == j) tp = Tap{r, s, ((rho + 1) >> 1) - u + 1, ((sig + 1) >> 1) - v + 1, rho * 2 + sig};
                        ++n;
                    }
            }
        return tp;
    };
    // issue order: the taps of one class keep their order (same sums as k_up_mfma, bit for bit) but
    // consecutive MFMAs go to different accumulators (a dependent MFMA waits for its predecessor)
    constexpr int ORDER[25] = {12, 10, 13, 2, 14, 11, 17, 3, 18, 15, 19, 4, 0, 22, 16, 7, 1, 23, 20, 8, 5, 24, 21, 9, 6};
    auto load_a = [&](const int wa, const int cp, const int jj, float (&a)[25]) __attribute__((always_inline)) {
        const int j = ORDER[jj];
        const Tap tp = tap_of(j);
        a[j] = *reinterpret_cast<const float*>(sm + wa + ((2 * cp) * TM * RS + tp.r * 5 + tp.s) * 4);
    };
    auto load_b = [&](const int xb, const int xu, const int xd, const int cp, const int i, float (&b)[9]) __attribute__((always_inline)) {
        const int dy = i / 3, dx = i - 3 * dy;
        if constexpr (RT) {
            const int xr = dy == 0 ? xu : dy == 1 ? xb : xd;
            b[i] = *reinterpret_cast<const float*>(sm + xr + ((2 * cp) * T::CHSP + (dx - 1)) * 4);
        } else {
            constexpr int SW = (1 << LGW) + 2;
            b[i] = *reinterpret_cast<const float*>(sm + xb + ((2 * cp) * T::CHSP + (dy - 1) * SW + (dx - 1)) * 4);
        }
    };
    auto chunk_body = [&](const int xb, const int xu, const int xd, const int wa, const int nbuf, const bool more, const int c0n) __attribute__((always_inline)) {
        float av[25], bv[2][9];
#pragma unroll
        for (int i = 0; i < 9; ++i) load_b(xb, xu, xd, 0, i, bv[0]);
#pragma unroll
        for (int j = 0; j < 25; ++j) {
            const Tap t0 = tap_of(ORDER[j]);
            if (up2_tap_live<KV>(t0.r, t0.s)) load_a(wa, 0, j, av);
        }
#pragma unroll
        for (int cp = 0; cp < CC / 2; ++cp) {
#pragma unroll
            for (int j = 0; j < 25; ++j) {
                const Tap tp = tap_of(ORDER[j]);
                const bool live = up2_tap_live<KV>(tp.r, tp.s);
                if (live)
                    acc[tp.cl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ORDER[j]], bv[cp & 1][tp.dy * 3 + tp.dx],
                                                                      acc[tp.cl], 0, 0, 0);
                if (cp + 1 < CC / 2) {
                    if (live) load_a(wa, cp + 1, j, av);             // refresh in place
                    if (j >= 8 && j < 17) load_b(xb, xu, xd, cp + 1, j - 8, bv[(cp + 1) & 1]);
                }
                // the next chunk's DMA rides in this wave's own MFMA stream, one instruction per slot
                if (more && cp * 25 + j >= 2 && cp * 25 + j < 2 + NDMA) issue_dma(cp * 25 + j - 2, nbuf, c0n);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    static_assert(NDMA + 2 <= (CC / 2) * 25, "DMA slots");
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue_dma(d, 0, c_beg);
    int cur = 0;
#ifdef UP2_LOOP_ALIGN
    asm volatile(".p2align " UP2_STR(UP2_LOOP_ALIGN) "\n .rept " UP2_STR(UP2_LOOP_SHIFT) "\n s_nop 0\n .endr" ::: "memory");
#endif
    for (int c0 = c_beg; c0 < c_end; c0 += CC) {
        // own DMA of this chunk landed; after the barrier everyone's has, and every wave is done
        // reading the other image pair
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        chunk_body(xb_cur, xu_cur, xd_cur, wa_cur, cur ^ 1, c0 + CC < c_end, c0 + CC);
        cur ^= 1;
        int tmp = xb_cur; xb_cur = xb_oth; xb_oth = tmp;
        tmp = wa_cur; wa_cur = wa_oth; wa_oth = tmp;
        if constexpr (RT) {
            tmp = xu_cur; xu_cur = xu_oth; xu_oth = tmp;
            tmp = xd_cur; xd_cur = xd_oth; xd_oth = tmp;
        }
    }

    // ---- epilogue: lane owns output pixels (2a+rho, 2b+sig) of channel m(e, kk); the two column
    // classes are adjacent -> one 8-byte store per (channel, row class)
    const int Wb = g.Wb, HWb = g.Hb * g.Wb;
    const int obytes = (int)((size_t)g.N * g.Cb * HWb * 4);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)dact_src, 0, dact_src ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)bias, 0, bias ? g.Cb * 4 : 0, 0x00020000);
    const int mlane = m0 + 4 * kk;
    const int n = n0 + pf;
    const int vo = pvalid ? (int)((((size_t)n * g.Cb + mlane) * HWb + (size_t)(2 * (a0 + aj)) * Wb + 2 * bj) * 4) : OOB;
    typedef float floatx2u __attribute__((ext_vector_type(2)));
    typedef unsigned int uintx2u __attribute__((ext_vector_type(2)));
    if (act != BN_ACT_SIGMOID && dact != BN_ACT_SIGMOID) {             // wave-uniform
        const float es = (act == BN_ACT_LRELU) ? slope : 1.f;           // identity = slope 1
        const float ds = (dact == BN_ACT_LRELU) ? slope : 1.f;
        float bz[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            bz[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rbs, (mlane + (e & 3) + 8 * (e >> 2)) * 4, 0, 0));
#pragma unroll
        for (int rho = 0; rho < 2; ++rho) {
            floatx2u d[16];
            if (dact_src) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mo = (e & 3) + 8 * (e >> 2);
                    d[e] = __builtin_bit_cast(floatx2u, __builtin_amdgcn_raw_buffer_load_b64(
                        rd, (mlane + mo < g.Cb) ? vo : OOB, (mo * HWb + rho * Wb) * 4, 0));
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int mo = (e & 3) + 8 * (e >> 2);
                floatx2u v;
                v.x = acc[rho * 2 + 0][e] + bz[e];
                v.y = acc[rho * 2 + 1][e] + bz[e];
                v.x = v.x > 0.f ? v.x : v.x * es;
                v.y = v.y > 0.f ? v.y : v.y * es;
                if (dact_src) {
                    v.x *= d[e].x > 0.f ? 1.f : ds;
                    v.y *= d[e].y > 0.f ? 1.f : ds;
                }
                if (UP2_ST_AUX_D != UP2_ST_AUX && dact_src)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uintx2u, v), ro,
                                                          (mlane + mo < g.Cb) ? vo : OOB,
                                                          (mo * HWb + rho * Wb) * 4, UP2_ST_AUX_D);
                else
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uintx2u, v), ro,
                                                          (mlane + mo < g.Cb) ? vo : OOB,
                                                          (mo * HWb + rho * Wb) * 4, UP2_ST_AUX);
            }
        }
    } else {
        if (!pvalid) return;
        const int h0 = 2 * (a0 + aj), w0 = 2 * bj;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (m >= g.Cb) continue;
            const float bm = bias ? bias[m] : 0.f;
#pragma unroll
            for (int rho = 0; rho < 2; ++rho) {
                const size_t idx = ((size_t)n * g.Cb + m) * HWb + (size_t)(h0 + rho) * Wb + w0;
                float2 v;
                v.x = bn_apply_act(acc[rho * 2 + 0][e] + bm, act, slope);
                v.y = bn_apply_act(acc[rho * 2 + 1][e] + bm, act, slope);
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

// streamlined kernel: square 8x8 / 16x16 / 32x32 small images, whole chunks of channels
static bool up2_ok(const BnGeom& g, int* cc_out) {
    static int mode = -1;                 // BN_UP2=0: off; BN_UP2=8: 8-channel chunks
    if (mode < 0) { const char* e = bn_tune_env("BN_UP2"); mode = e ? atoi(e) : 4; }
    if (mode == 0) return false;
    const int lgw = ilog2_exact_up(g.Ws);
    if (g.Hs != g.Ws || lgw < 3 || lgw > 5) return false;
    const int cc = (mode == 8 && (g.Cs % 8) == 0) ? 8 : 4;
    if ((g.Cs % cc) != 0 || (g.Cb & 3) != 0) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    *cc_out = cc;
    return true;
}

// reduction splits of the streamlined gather-up kernel: a grid that fills less than half of the chip
// (a 32-frame shard of a trial: 64 workgroups of 105 us each) is cut over the input channels into
// up to 8 slices of >= 32 channels, raw sums through scratch, finished by k_split_epilogue
static int up2_splits(const BnGeom& g, int lgw, int cc) {
    const int F = (1 << (2 * lgw)) >= 128 ? 1 : 128 >> (2 * lgw);
    const int tpf = (1 << (2 * lgw)) >= 128 ? (1 << (2 * lgw)) / 128 : 1;
    const int wgs = ((g.N + F - 1) / F) * tpf * ((g.Cb + 31) / 32);
    if (wgs > 128) return 1;
    int s = 256 / wgs;
    while (s > 1 && (g.Cs % (s * cc) != 0 || g.Cs / s < 32)) --s;
    return s > 8 ? 8 : (s < 1 ? 1 : s);
}

template <int LGW, int CC, int KV>
static int launch_up2(const float* small, const float* w, const float* bias, float* out,
                      const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                      hipStream_t st, int splits, void* ws) {
    using T = UP2<LGW>;
    constexpr int WDMA = (CC * 32 * 25 / 4 + MF_THREADS - 1) / MF_THREADS;
    constexpr size_t lds = ((size_t)2 * CC * T::CHSP + (size_t)2 * WDMA * MF_THREADS * 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_up2_mfma<LGW, CC, KV>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int groups = (g.N + T::F - 1) / T::F;
    dim3 grid(groups * T::TPF, (g.Cb + 31) / 32, splits);
    if (splits > 1) {
        if (!ws) return BN_E_WORKSPACE;
        const size_t total = (size_t)g.N * g.Cb * g.Hb * g.Wb;
        BN_LAUNCH_MAIN((k_up2_mfma<LGW, CC, KV>), grid, dim3(MF_THREADS), lds, st, small, w,
                       (const float*)nullptr, (float*)ws, (const float*)nullptr, g, BN_ACT_NONE,
                       BN_ACT_NONE, slope, g.Cs / splits, total, Up2Geo{});
        BN_LAUNCH_CHECK();
        return bn_launch_split_epilogue((const float*)ws, bias, out, dact_src, total, splits, g.Cb,
                                        g.Hb * g.Wb, act, dact, slope, st);
    }
    BN_LAUNCH_MAIN((k_up2_mfma<LGW, CC, KV>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                       dact_src, g, act, dact, slope, g.Cs, (size_t)0, Up2Geo{});
    BN_LAUNCH_CHECK();
    return 0;
}

// ---- runtime tile geometry (k_up2_mfma<0, CC, KV>): maps that are no power-of-two squares ----------
// tile = F whole frames, or AT_H rows of one frame (rows spread evenly over a frame's tiles), at most 128
// positions and 256 elements of the haloed LDS image
static bool up2g_geo_lim(const BnGeom& g, Up2Geo* t, int chsp, float* fill_out);
// the 256-element image unless the 320-element one fills the tile's 128 positions at least a tenth better
static bool up2g_geo(const BnGeom& g, Up2Geo* t) {
    float f0 = 0.f, f1 = 0.f;
    Up2Geo t1;
    const bool ok0 = up2g_geo_lim(g, t, UP2<0>::CHSP, &f0);
    // (only maps the small image serves too: the dispatch keeps wider maps on tiles of the square instantiations)
    const bool ok1 = ok0 && up2g_geo_lim(g, &t1, UP2<1>::CHSP, &f1);
    if (ok1 && f1 > 1.1f * f0) { *t = t1; return true; }
    return ok0;
}
static bool up2g_geo_lim(const BnGeom& g, Up2Geo* t, int chsp, float* fill_out) {
    if (g.R != 5 || g.S != 5 || g.stride != 2 || g.pt != 1 || g.pl != 1) return false;
    if (g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return false;
    if ((g.Cs % 4) != 0 || (g.Cb & 3) != 0 || g.Cb < 16) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.Cs * g.Cb * 25 * 4 >= 0x7fffffffull) return false;
    const int TP = 128, HW = g.Hs * g.Ws;
    if (g.Ws < 2 || g.Ws > TP) return false;
    t->SWp = g.Ws + 2;
    (void)HW;
    // units of AT_H rows, F of them per tile (whole frames, row blocks of one frame, or row blocks of adjacent
    // frames): the split of a frame into UPF blocks that fills the 128 positions best within the 256 elements
    // of the haloed LDS image; ties: fewer blocks
    float best = 0.f;
    int b_upf = 0, b_ath = 0, b_F = 0;
    for (int upf = 1; upf <= g.Hs; ++upf) {
        const int ath = (g.Hs + upf - 1) / upf;
        if ((g.Hs + ath - 1) / ath != upf || ath * g.Ws > TP) continue;
        int F = TP / (ath * g.Ws);
        const int fs = (ath + 2) * t->SWp;
        while (F >= 1 && F * fs > chsp) --F;
        if (F < 1) continue;
        const float fill = (float)(F * ath * g.Ws) / (float)TP * (float)g.Hs / (float)(upf * ath);
        if (fill > best + 1e-6f) { best = fill; b_upf = upf; b_ath = ath; b_F = F; }
        if (best >= 0.999f) break;
    }
    if (b_upf == 0) return false;
    *fill_out = best;
    t->UPF = b_upf; t->AT_H = b_ath; t->F = b_F;
    t->FS = (b_ath + 2) * t->SWp;
    t->CHS = b_F * t->FS;
    t->ATW = t->AT_H * g.Ws;
    return true;
}

static int up2g_splits(const BnGeom& g, const Up2Geo& t, int cc) {
    const int wgs = ((g.N * t.UPF + t.F - 1) / t.F) * ((g.Cb + 31) / 32);
    if (wgs > 128) return 1;
    int s = 256 / wgs;
    while (s > 1 && (g.Cs % (s * cc) != 0 || g.Cs / s < 32)) --s;
    return s > 8 ? 8 : (s < 1 ? 1 : s);
}

template <int KV, int LG>
static int launch_up2g_img(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       hipStream_t st, int splits, void* ws) {
    constexpr int CC = 4;
    Up2Geo t;
    if (!up2g_geo(g, &t)) return BN_E_SHAPE;
    constexpr int WDMA = (CC * 32 * 25 / 4 + MF_THREADS - 1) / MF_THREADS;
    constexpr size_t lds = ((size_t)2 * CC * UP2<LG>::CHSP + (size_t)2 * WDMA * MF_THREADS * 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_up2_mfma<LG, CC, KV>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((g.N * t.UPF + t.F - 1) / t.F, (g.Cb + 31) / 32, splits);
    if (splits > 1) {
        if (!ws) return BN_E_WORKSPACE;
        const size_t total = (size_t)g.N * g.Cb * g.Hb * g.Wb;
        BN_LAUNCH_MAIN((k_up2_mfma<LG, CC, KV>), grid, dim3(MF_THREADS), lds, st, small, w,
                       (const float*)nullptr, (float*)ws, (const float*)nullptr, g, BN_ACT_NONE,
                       BN_ACT_NONE, slope, g.Cs / splits, total, t);
        BN_LAUNCH_CHECK();
        return bn_launch_split_epilogue((const float*)ws, bias, out, dact_src, total, splits, g.Cb,
                                        g.Hb * g.Wb, act, dact, slope, st);
    }
    BN_LAUNCH_MAIN((k_up2_mfma<LG, CC, KV>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                       dact_src, g, act, dact, slope, g.Cs, (size_t)0, t);
    BN_LAUNCH_CHECK();
    return 0;
}

template <int KV>
static int launch_up2g(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       hipStream_t st, int splits, void* ws) {
    Up2Geo t;
    if (!up2g_geo(g, &t)) return BN_E_SHAPE;
    return t.CHS > UP2<0>::CHSP
        ? launch_up2g_img<KV, 1>(small, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws)
        : launch_up2g_img<KV, 0>(small, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws);
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
    if (!ok1 && !ok2) {
        // no power-of-two map: the streamlined kernel with its tile geometry at run time
        Up2Geo tg;
        static int off = -1;                      // BN_UP2G=0: off
        if (off < 0) { const char* e = bn_tune_env("BN_UP2G"); off = (e && e[0] == '0') ? 1 : 0; }
        if (off || !up2g_geo(g, &tg)) return p;
        p.supported = true;
        p.a = 1; p.c = 4; p.variant = 3;
        p.kernel_name = tg.CHS > UP2<0>::CHSP ? "k_up2_mfma<1, 4>" : "k_up2_mfma<0, 4>";
        p.d = up2g_splits(g, tg, 4);
        p.ws_bytes = p.d > 1 ? (size_t)p.d * g.N * g.Cb * g.Hb * g.Wb * sizeof(float) : 0;
        return p;
    }
    p.supported = true;
    p.a = (ok2 && (nwg2 >= 768 || !ok1)) ? 2 : 1;
    if (env_mr == 2 && ok2) p.a = 2;
    if (env_mr == 1 && ok1) p.a = 1;
    p.c = cc;
    p.kernel_name = p.a == 2 ? (cc == 8 ? "k_up_mfma<2, 8>" : "k_up_mfma<2, 4>")
                             : (cc == 8 ? "k_up_mfma<1, 8>" : "k_up_mfma<1, 4>");
    // the streamlined kernel wherever it applies (round 3: 64-channel big sides used to take the
    // two-block first-generation kernel once the grid was large: 530 against 2 x 206 us)
    if ((p.a == 1 || (ok1 && env_mr != 2)) && up2_ok(g, &p.c)) {
        p.a = 1;
        p.variant = 2;
        const int lgw = ilog2_exact_up(g.Ws);
        static const char* const n4[6] = {"", "", "", "k_up2_mfma<3, 4>", "k_up2_mfma<4, 4>", "k_up2_mfma<5, 4>"};
        static const char* const n8[6] = {"", "", "", "k_up2_mfma<3, 8>", "k_up2_mfma<4, 8>", "k_up2_mfma<5, 8>"};
        p.kernel_name = p.c == 8 ? n8[lgw] : n4[lgw];
        p.d = up2_splits(g, lgw, p.c);
        p.ws_bytes = p.d > 1 ? (size_t)p.d * g.N * g.Cb * g.Hb * g.Wb * sizeof(float) : 0;
    }
    return p;
}

int bn_launch_up_fast(const BnFastPlan& plan, const float* small, const float* w,
                      const float* bias, float* out, const float* dact_src, const BnGeom& g,
                      int act, int dact, float slope, void* ws, hipStream_t st) {
    const int MR = plan.a, CC = plan.c;
    if (plan.variant == 3)
        return g.KV == 4 && g.K0 == 1
            ? launch_up2g<14>(small, w, bias, out, dact_src, g, act, dact, slope, st, plan.d > 1 ? plan.d : 1, ws)
            : g.KV == 4
            ? launch_up2g<4>(small, w, bias, out, dact_src, g, act, dact, slope, st, plan.d > 1 ? plan.d : 1, ws)
            : launch_up2g<5>(small, w, bias, out, dact_src, g, act, dact, slope, st, plan.d > 1 ? plan.d : 1, ws);
    if (plan.variant == 2) {
        const int lgw = ilog2_exact_up(g.Ws);
        const int splits = plan.d > 1 ? plan.d : 1;
#define UP2_CASE(L, C)                                                                          \
    if (lgw == L && CC == C)                                                                    \
        return launch_up2<L, C, 5>(small, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws);
#define UP2_CASE4(L)                                                                            \
    if (lgw == L && CC == 4 && g.KV == 4 && g.K0 == 1)                                          \
        return launch_up2<L, 4, 14>(small, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws); \
    if (lgw == L && CC == 4 && g.KV == 4)                                                       \
        return launch_up2<L, 4, 4>(small, w, bias, out, dact_src, g, act, dact, slope, st, splits, ws);
        UP2_CASE4(3) UP2_CASE4(4) UP2_CASE4(5)
#undef UP2_CASE4
        UP2_CASE(3, 4) UP2_CASE(4, 4) UP2_CASE(5, 4) UP2_CASE(3, 8) UP2_CASE(4, 8) UP2_CASE(5, 8)
#undef UP2_CASE
        return BN_E_SHAPE;
    }
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
        BN_LAUNCH_MAIN((k_up_mfma<2, 8>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else if (MR == 2) {
        BN_LAUNCH_MAIN((k_up_mfma<2, 4>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else if (CC == 8) {
        BN_LAUNCH_MAIN((k_up_mfma<1, 8>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    } else {
        BN_LAUNCH_MAIN((k_up_mfma<1, 4>), grid, dim3(MF_THREADS), lds, st, small, w, bias, out,
                           dact_src, g, t, act, dact, slope);
    }
    BN_LAUNCH_CHECK();
    return 0;
}
