// family 3, second generation: weight gradients on the matrix cores (kernel 5x5, stride 2,
// left/top offset 1) with 16-byte LDS-DMA staging and 8-byte operand reads.
//   dW[a][b][r][s] = sum_{n,p,q} small[n,a,p,q] * big[n,b,2p+r-1,2q+s-1]
//
// Same GEMM view and wave layout as conv_mfma_wgrad.hip (v_mfma_f32_16x16x4_f32, 25 tap
// accumulators per wave, 8 waves = 64 x 32 tile of dW, 64-pixel stages, double-buffered LDS
// images filled by `buffer_load ... lds` one stage ahead).  What changed is the data layout:
//
//  * big tile: rows are stored UNSPLIT, image column wb at LDS column wb + 4, so a patch row is a
//    run of 16-byte groups that are 16-byte aligned in global memory as well: one
//    buffer_load_dwordx4 ... lds per lane moves four pixels (a stage needs ~7 per thread instead
//    of ~31 dword DMAs; the address unit, not the matrix cores, was the bottleneck).  Groups left
//    of column 0 / right of the image and padding rows are out-of-range reads = 0.0f.
//  * B operand: a lane needs columns 2q+s-1, s = 0..4, of its pixel q: the three aligned pairs
//    (2q-2,2q-1) (2q,2q+1) (2q+2,2q+3) -> three ds_read_b64 per kernel row instead of five
//    ds_read_b32.  The channel stride is 4*odd (mod 64) words, which makes the b64 reads of the
//    16 channels x 2 adjacent pixels of a half-wave hit 64 distinct banks.
//  * small tile: 64-pixel rows stored without padding, the 4-pixel groups of row a XOR-swizzled
//    by (a & 15); four rows per DMA instruction, A reads are at worst 2-way conflicted (1 of 16
//    LDS reads per k-step).
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"
#include "bn_reduce.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

#define W4_STR2(x) #x
#define W4_STR(x) W4_STR2(x)
#ifndef W4_LOOP_SHIFT
#define W4_LOOP_SHIFT 0
#endif
#define W4_THREADS 512
#define W4_TA 64            // a-channels per workgroup tile
#define W4_TB 32            // b-channels per workgroup tile
#define W4_TPX 64           // small-image pixels per stage
#define W4_X0 4             // LDS column of image column 0
#define W4_SLICES 8         // DMA slices per stage = MFMA-loop trips (2 k-steps each)
#define W4_MAX_LDS (160 * 1024)
#define W4_OOB 0x7fffffff

static inline int ilog2_exact_w4(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

struct Wgrad4Tile {
    int F, PT_H, lgPTQ;            // pixel stage = F frames x PT_H rows x Q columns (64 pixels)
    int tiles_per_frame, n_stages;
    int IH, rows_per_b;            // patch rows per frame / per channel (F * IH)
    int GPB;                       // 16-byte groups per channel image; BCH = 4 * GPB words
    int row_groups;                // groups of a channel that carry data (rows_per_b * RW / 4)
    int big_groups;                // W4_TB * GPB
    int FSb;                       // per-frame stride inside a channel (IH * RW)
    float inv_gpb, inv_c4, inv_ih; // reciprocals for the group decode
    int splits;                    // reduction splits (gridDim.y)
    int buf_floats;                // one LDS stage image
    int bias_side;                 // fused bias gradient: 0 none, 1 sum of `small` per a-channel
                                   // (Conv2d), 2 sum of `big` per b-channel (ConvTranspose2d)
    int nbias;                     // channels of that side (stride of the partial bias rows)
};

template <int LGQ>
__global__ __launch_bounds__(W4_THREADS, 2) void k_wgrad4_mfma(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    float* __restrict__ bias_part, BnGeom g, Wgrad4Tile t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Q = 1 << LGQ, RW = 2 * Q + 8, C4 = RW / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ablk = wv >> 1, bblk = wv & 1;
    const int lj = lane & 15, kk = lane >> 4;

    const int n_btiles = (g.Cb + W4_TB - 1) / W4_TB;
    const int atile = blockIdx.x / n_btiles, btile = blockIdx.x - atile * n_btiles;
    const int a0 = atile * W4_TA, b0 = btile * W4_TB;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;
    const int BCH = 4 * t.GPB;

    floatx4 acc[25];
#pragma unroll
    for (int tp = 0; tp < 25; ++tp) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    // fused bias gradient: the operand values pass through this lane's registers anyway.
    // side 1: every small pixel is the A operand of exactly one k-step; side 2: the taps
    // (r,s) in {1,2}^2 of all small pixels tile the big image (2p+r-1, 2q+s-1) exactly once.
    float bsum = 0.f;

    const __amdgpu_buffer_rsrc_t rs_small = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HWb * 4), 0x00020000);

    // queue slice j (of W4_SLICES) of one stage's DMA into image `buf`: every thread moves one
    // 16-byte group of the big tile (group e = tid + 512 j of the [b][BCH] image) and, for
    // j < 2, one group of the small tile.  Lane L of a wave instruction lands at the wave's
    // base + 16 L bytes, i.e. the images are filled in plain group order.
    auto issue_dma = [&](int st, int buf, int j) {
        const int grp = st / t.tiles_per_frame;
        const int n0 = grp * t.F;
        const int p0 = (st - grp * t.tiles_per_frame) * t.PT_H;
        float* sl = smem + buf * t.buf_floats;
        float* bl = sl + W4_TA * W4_TPX;
        if (j < (W4_TA * W4_TPX / 4) / W4_THREADS) {
            const int e = tid + W4_THREADS * j;
            const int a = e >> 4;
            const int pix0 = 4 * ((e & 15) ^ (a & 15));          // swizzled source group
            const int f = pix0 >> t.lgPTQ;
            const int rem = pix0 & ((1 << t.lgPTQ) - 1);
            const bool ok = (a0 + a < g.Cs) && (n0 + f < g.N);
            const int off = (((n0 + f) * g.Cs + a0 + a) * PQ + p0 * Q + rem) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_small, sl + 4 * (W4_THREADS * j + 64 * wv), 16, ok ? off : W4_OOB, 0, 0, 0);
        }
        if (W4_THREADS * j + 64 * wv < t.big_groups) {               // wave-uniform
            int e = tid + W4_THREADS * j;
            asm volatile("" : "+v"(e));   // keep the decode next to its load (no hoisting)
            const int b = (int)(((float)e + 0.5f) * t.inv_gpb);
            const int within = e - b * t.GPB;
            const int rr = (int)(((float)within + 0.5f) * t.inv_c4);
            const int c4 = within - rr * C4;
            const int f = (t.F == 1) ? 0 : (int)(((float)rr + 0.5f) * t.inv_ih);
            const int y = rr - f * t.IH;
            const int hb = 2 * p0 - g.pt + y, wb = 4 * c4 - W4_X0;
            const bool ok = (b < W4_TB) && (within < t.row_groups) && (b0 + b < g.Cb) &&
                            (n0 + f < g.N) && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            const int off = ((((n0 + f) * g.Cb + b0 + b) * g.Hb + hb) * g.Wb + wb) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rs_big, bl + 4 * (W4_THREADS * j + 64 * wv), 16, ok ? off : W4_OOB, 0, 0, 0);
        }
    };

    const int a_off = (ablk * 16 + lj) * W4_TPX + kk;
    const int b_off = W4_TA * W4_TPX + (bblk * 16 + lj) * BCH + (W4_X0 - 2);

    int st = blockIdx.y;
    int cur = 0;
    if (st < t.n_stages) {
#pragma unroll 1
        for (int j = 0; j < W4_SLICES; ++j) issue_dma(st, 0, j);
    }
    for (; st < t.n_stages; st += t.splits) {
        // own DMAs of this stage have landed; after the barrier everyone's have, and every wave
        // is done reading the other image (it was computed from in the previous trip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = st + t.splits < t.n_stages;

        const float* ap = smem + cur * t.buf_floats + a_off;
        const float* bp = smem + cur * t.buf_floats + b_off;
#pragma unroll 1
        for (int j = 0; j < W4_SLICES; ++j) {
            if (more) issue_dma(st + t.splits, cur ^ 1, j);
#pragma unroll
            for (int ks = 2 * j; ks < 2 * j + 2; ++ks) {
                const int pix = 4 * ks + kk;
                const int f = pix >> t.lgPTQ;
                const int rem = pix & ((1 << t.lgPTQ) - 1);
                const int pj = rem >> LGQ, qj = rem & (Q - 1);
                const float av = ap[(ks ^ lj) << 2];
                if (t.bias_side == 1) bsum += av;
                // columns (2q-2, 2q-1 | 2q, 2q+1 | 2q+2, 2q+3) of patch row 2p + r
                const float* bq = bp + f * t.FSb + (2 * pj) * RW + 2 * qj;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const floatx2 c0 = *reinterpret_cast<const floatx2*>(bq + r * RW);
                    const floatx2 c1 = *reinterpret_cast<const floatx2*>(bq + r * RW + 2);
                    const floatx2 c2 = *reinterpret_cast<const floatx2*>(bq + r * RW + 4);
                    if (t.bias_side == 2 && (r == 1 || r == 2)) bsum += c1.x + c1.y;
                    acc[r * 5 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c0.y, acc[r * 5 + 0],
                                                                          0, 0, 0);
                    acc[r * 5 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c1.x, acc[r * 5 + 1],
                                                                          0, 0, 0);
                    acc[r * 5 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c1.y, acc[r * 5 + 2],
                                                                          0, 0, 0);
                    acc[r * 5 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c2.x, acc[r * 5 + 3],
                                                                          0, 0, 0);
                    acc[r * 5 + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, c2.y, acc[r * 5 + 4],
                                                                          0, 0, 0);
                }
            }
        }
        cur ^= 1;
    }

    if (t.bias_side != 0) {
        // lanes lj + 16 kk hold the four pixel phases of one channel: fixed-order butterfly
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        float* bdst = bias_part + (size_t)blockIdx.y * t.nbias;
        if (t.bias_side == 1 && btile == 0 && bblk == 0 && kk == 0) {
            const int a = a0 + ablk * 16 + lj;
            if (a < g.Cs) bdst[a] = bsum;
        }
        if (t.bias_side == 2 && atile == 0 && ablk == 0 && kk == 0) {
            const int bb = b0 + bblk * 16 + lj;
            if (bb < g.Cb) bdst[bb] = bsum;
        }
    }

    // partial tile -> scratch [split][tap][a][b]; lane holds D[i = 4*kk + e][j = lj]
    float* dst = part + (size_t)blockIdx.y * 25 * g.Cs * g.Cb;
    const int b = b0 + bblk * 16 + lj;
    if (b < g.Cb) {
#pragma unroll
        for (int tp = 0; tp < 25; ++tp) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = a0 + ablk * 16 + 4 * kk + e;
                if (a < g.Cs) dst[((size_t)tp * g.Cs + a) * g.Cb + b] = acc[tp][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Streamlined variant for stages that lie inside one frame (F == 1: Hs * Ws >= 64, the bench
// layers; since round 4 templated on the map WIDTH: 8-44 in steps of 4 with stride 2, 8-64 with stride 1,
// any height -- see W4S and the GEN / KV / ST / PTH template arguments of the kernel).  Same tiles, LDS images and MFMA roles as k_wgrad4_mfma above -- and the same partial
// sums, bit for bit -- but nothing is left for the vector ALU inside the multiply loop:
//  * on this chip every VALU instruction that is not an MFMA takes 6-13 cycles away from the matrix
//    pipe (tools/lab/issue_probe.hip), LDS reads and scalar instructions none; the loop above spends
//    about one VALU instruction per MFMA (pixel decode, operand addresses, the float-multiply group
//    decode of the DMA) and reaches 65 % matrix utilisation;
//  * here the geometry of a stage is a compile-time function of LGQ: an operand read is
//    `ds_read vbase offset:imm` (one base register per LDS image; the A read adds one v_xor for
//    the swizzle), the DMA source offsets of a thread are computed once and a stage contributes a
//    scalar offset (buffer soffset) plus a two-compare row test per group;
//  * the reads of k-step i+1 and the DMA instructions of the next stage are placed by hand between
//    the 25 MFMAs of k-step i (sched_barrier pins the order), everything unrolled over the 16
//    k-steps of a stage and over the two LDS images;
//  * one workgroup per CU by __launch_bounds__ (it needs 150 KB of LDS anyway): 256 registers.
// ---------------------------------------------------------------------------------------------
// W4S<Q>: any map width that is a multiple of 4 (the k-step's four pixels lie in one row); a stage is
// the PT_H = 64 / Q whole rows that fit into 64 pixels, KS k-steps of four pixels
// ST = 1 (round 4): stride-1 layers (5x5 taps, any offsets up to 4) -- big patch rows PT_H + 4, row stride
// Q + 8, a lane's five columns are consecutive words (five 4-byte reads instead of 1 + 2 x 8 bytes)
// PTH_ (0 = 64 / Q): rows of a stage -- 4 instead of 8 for 8-column maps of at most four rows (the 4x3
// maps of 64x48 frames on their 4x8 zero-padded copies: half the k-steps of a stage were zeros)
template <int Q_, int ST_ = 2, int PTH_ = 0>
struct W4S {
    static constexpr int Q = Q_, ST = ST_, PT_H = PTH_ ? PTH_ : W4_TPX / Q, IH = ST * (PT_H - 1) + 5;
    static constexpr int KS = PT_H * Q / 4;
    static constexpr int RW = ST * Q + 8, C4 = RW / 4;
    static constexpr int ROWG = IH * C4;                         // data groups of a channel image
    static constexpr int GPB = (ROWG & 1) ? ROWG : ROWG + 1;     // odd: conflict-free b64 reads
    static constexpr int BCH = 4 * GPB;
    static constexpr int BIGG = W4_TB * GPB;
    static constexpr int SMALLW = W4_TA * W4_TPX;
    static constexpr int BUFW = SMALLW + 4 * ((BIGG + 63) / 64 * 64);
    static constexpr int NBIG = (BIGG + W4_THREADS - 1) / W4_THREADS;
    static constexpr int NSM = SMALLW / 4 / W4_THREADS;
};

__device__ __forceinline__ void w4_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voffset,
                                         int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voffset, soffset, 0, 0);
}

// GEN: maps that are no powers of two -- the stages of a frame are ceil(Hs / PT_H) (lg_tpf then carries
// the magic multiplier 2^32 / tiles + 1 of the stage -> frame division), a frame's last stage may hang
// over its lower edge: the small rows below the map and the big rows below it are left out of the DMA
// (out-of-range source offset = 0.0f) by a per-group row number against a per-stage scalar limit.
// KV = 4: taps with r >= 4 or s >= 4 belong to the zero extension of a smaller kernel (BnGeom::KV): their
// products are skipped and their (zero) tiles are dropped again by the caller's crop of dW; KV = 10 K0 + K1
// in general: the taps [K0, K1)^2 are the layer's (14 = a 3x3 kernel embedded at (1, 1), BnGeom::K0 = 1)
// CW (round 4): COLUMN WINDOWS -- a map wider than any instantiated width (48 = 2 x 24, 64 = 2 x 32, 96 = 4 x 24:
// the first matrix-core layer of 192- / 256-pixel-wide frames) is walked in windows of QQ columns: a stage is
// PT_H rows of ONE window (its small rows lie g.Ws apart, its big patch starts 2 cb QQ columns into the row and
// has real neighbours on the inner sides: only the first window's left groups and the last window's right
// groups are padding).  `ncb` windows per row; stages are ordered (frame, window, row block).
// NA x NB (round 4): 16-channel blocks of the tile that hold channels (default 4 x 2 = the whole 64 x 32 tile).  A layer
// with at most 32 small-side / 16 big-side channels (the 16- and 32-channel layers of max-pooling architectures) left
// six of the eight waves multiplying zero rows; with NA x NB < 8 the waves form NG = 8 / (NA NB) groups, every group owns
// the SAME blocks for its own 1 / NG of a stage's k-steps (an offset in its two LDS base registers) and writes its own
// partial tile: the caller's sum over splits runs over NG times as many slabs.
template <int QQ, int BIAS, bool GEN, int KV, int ST = 2, int PTH = 0, bool CW = false, int NA = 4, int NB = 2>
__global__ __launch_bounds__(W4_THREADS, 1) void k_wgrad4s_mfma(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    float* __restrict__ bias_part, BnGeom g, int n_stages, int splits, int lg_tpf, int nbias, int ncb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using T = W4S<QQ, ST, PTH>;
    static_assert(ST == 2 || GEN, "stride 1: the general row limits");
    static_assert(!CW || (GEN && ST == 2 && (QQ & 3) == 0), "column windows: general stride-2 stages");
    constexpr int Q = T::Q, RW = T::RW, BUFW = T::BUFW;
    // widths == 2 (mod 4) (round 4: the 12x10 / 8x6 maps of 160- / 96-pixel-wide frames): every other row of a
    // stage starts in the middle of a k-step -- pixels 2, 3 of such a k-step lie in the next row, whose
    // lanes (kk >= 2) read through a second base register that is 2 RW - 2 Q words further on; needs an
    // even number of rows per stage and per map (the row limits of the DMA are per 16-byte group)
    constexpr bool HALF = (Q & 3) == 2;
    static_assert(((Q & 3) == 0 || (HALF && ST == 2 && GEN && (T::PT_H & 1) == 0)) && T::KS >= 1 && T::KS <= 16,
                  "stage geometry");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NG = 8 / (NA * NB), KSG = T::KS / NG;      // wave groups, k-steps of a stage per group
    static_assert(NA * NB * NG == 8 && KSG * NG == T::KS && KSG >= 1 && NG <= 4, "wave groups");
    static_assert(NG == 1 || (!HALF && !CW && (KSG & (KSG - 1)) == 0 &&
                              ((4 * KSG) % Q == 0 || Q % (4 * KSG) == 0)), "wave groups: whole rows or parts of one");
    const int blk = wv % (NA * NB), grp = wv / (NA * NB);
    const int ablk = NG == 1 ? wv >> 1 : blk / NB, bblk = NG == 1 ? wv & 1 : blk % NB;
    const int lj = lane & 15, kk = lane >> 4;

    const int n_btiles = (g.Cb + W4_TB - 1) / W4_TB;
    const int atile = blockIdx.x / n_btiles, btile = blockIdx.x - atile * n_btiles;
    const int a0 = atile * W4_TA, b0 = btile * W4_TB;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;

    floatx4 acc[25];
#pragma unroll
    for (int tp = 0; tp < 25; ++tp) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // ---- DMA descriptors of this thread (stage independent) --------------------------------
    // the big image is addressed from one row above its start (patch row y is image row
    // 2 p0 - 1 + y): lane offsets stay non-negative, the stage adds a scalar offset
    const __amdgpu_buffer_rsrc_t rs_small = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    const int pt_rows = ST == 2 ? 1 : g.pt;                // patch row 0 = image row ST p0 - pt_rows
    // (CW: also from W4_X0 columns to the left, so that a window's left neighbour groups have offsets >= 0)
    const int lead = pt_rows * g.Wb + (CW ? W4_X0 : 0);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(big - lead), 0, (int)(((size_t)g.N * g.Cb * HWb + lead) * 4), 0x00020000);
    int soff[T::NSM], srow = 0;
#pragma unroll
    for (int j = 0; j < T::NSM; ++j) {
        const int e = tid + W4_THREADS * j;
        const int a = e >> 4;
        const int pix0 = 4 * ((e & 15) ^ (a & 15));                  // swizzled source group
        // (a stage's rows are g.Ws apart: the whole row, or one window of it)
        const int spix = CW ? (pix0 / Q) * g.Ws + (pix0 - (pix0 / Q) * Q) : pix0;
        soff[j] = (a0 + a < g.Cs && pix0 < 4 * T::KS) ? ((a0 + a) * PQ + spix) * 4 : W4_OOB;
        srow |= (pix0 / Q) << (8 * j);                               // GEN: its row inside the stage
    }
    // rows above the image exist only in a frame's first tile (patch row 0), rows below it only in
    // its last tile (the last two patch rows; Hb == 2 Hs): two class bits per group, all groups of
    // a thread packed into one register
    int voff[T::NBIG], rowcls = 0, ypack[2] = {0, 0}, colcls = 0;
    static_assert(T::NBIG <= 8, "row numbers of a thread's groups in two registers");
#pragma unroll
    for (int j = 0; j < T::NBIG; ++j) {
        const int e = tid + W4_THREADS * j;
        const int b = e / T::GPB, within = e - b * T::GPB;
        const int y = within / T::C4, c4 = within - y * T::C4;
        const int wb = 4 * c4 - W4_X0;
        if constexpr (CW) {
            // columns relative to the window (2 cb Q columns into the row): left of it only the first
            // window is padding, from 2 Q on only the last one
            const bool ok = (e < T::BIGG) && (within < T::ROWG) && (b0 + b < g.Cb);
            voff[j] = ok ? (((b0 + b) * g.Hb + y) * g.Wb + wb + W4_X0) * 4 : W4_OOB;
            colcls |= ((wb < 0 ? 1 : 0) | (wb >= 2 * Q ? 2 : 0)) << (2 * j);
        } else {
            const bool ok = (e < T::BIGG) && (within < T::ROWG) && (b0 + b < g.Cb) && wb >= 0 && wb < g.Wb;
            voff[j] = ok ? (((b0 + b) * g.Hb + y) * g.Wb + wb) * 4 : W4_OOB;
        }
        rowcls |= ((y == 0 ? 1 : 0) | (y >= T::IH - 2 ? 2 : 0)) << (2 * j);
        ypack[j >> 2] |= (y & 255) << (8 * (j & 3));
    }
    // one DMA instruction of stage `st` into image `buf`: d < NSM small tile, else big tile
    auto issue_dma = [&](const int d, const int buf, const int n0w, const int p0) __attribute__((always_inline)) {
        float* sl = smem + buf * BUFW;
        // CW: n0w = frame * ncb + window
        const int n0 = CW ? n0w / ncb : n0w;
        const int cb = CW ? n0w - n0 * ncb : 0;
        if (d < T::NSM) {
            int so = soff[d];
            if constexpr (GEN) {
                // rows of the stage below the map (a frame's last stage): 0.0f
                const int row = (srow >> (8 * d)) & 255;
                so = row < g.Hs - p0 ? so : W4_OOB;
            }
            w4_dma16(rs_small, sl + 4 * (W4_THREADS * d + 64 * wv), so,
                     (n0 * g.Cs * PQ + p0 * (CW ? g.Ws : Q) + cb * Q) * 4);
        } else {
            const int j = d - T::NSM;
            if (W4_THREADS * j + 64 * wv < T::BIGG) {                 // wave-uniform
                int vo;
                if constexpr (GEN) {
                    // patch row y is image row ST p0 - pt + y (stride 2: pt = 1): rows above the image
                    // (a frame's first stages) and from Hb + pt - ST p0 on below it read 0.0f
                    const int y = (ypack[j >> 2] >> (8 * (j & 3))) & 255;
                    const int ymin = pt_rows - ST * p0, ylim = g.Hb + pt_rows - ST * p0;
                    vo = (y >= ymin && y < ylim) ? voff[j] : W4_OOB;
                    if constexpr (CW) {
                        const int cmask = ((cb == 0 ? 1 : 0) | (cb == ncb - 1 ? 2 : 0)) << (2 * j);
                        vo = (colcls & cmask) ? W4_OOB : vo;
                    }
                } else {
                    // rows above the image (first tile) and below it (last tile) read 0.0f
                    const int smask = ((p0 == 0 ? 1 : 0) | (p0 + T::PT_H >= g.Hs ? 2 : 0)) << (2 * j);
                    vo = (rowcls & smask) ? W4_OOB : voff[j];
                }
                w4_dma16(rs_big, sl + T::SMALLW + 4 * (W4_THREADS * j + 64 * wv), vo,
                         ((n0 * g.Cb * g.Hb + ST * p0) * g.Wb + 2 * cb * Q) * 4);
            }
        }
    };
    constexpr int NDMA = T::NSM + T::NBIG;

    // ---- operand read addresses (A: byte, B: word offsets from smem), one base per LDS image ---
    int abase[2], bbase[2];
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        abase[bf] = (bf * BUFW + (ablk * 16 + lj) * W4_TPX + 4 * lj + kk) * 4;      // bytes
        // (a lane's first tap column ST q - pl sits at LDS column ST q - pl + X0; the reads are words 1..5)
        bbase[bf] = bf * BUFW + T::SMALLW + (bblk * 16 + lj) * T::BCH +
                    (ST == 2 ? (W4_X0 - 2) + 2 * kk : (W4_X0 - 1 - g.pl) + kk);
        if (NG > 1) {
            // this group's first k-step: (4 lj) ^ (4 (ks0 + k)) = ((4 lj) ^ (4 ks0)) ^ (4 k) for k < KSG, and the pixels of
            // k-step ks0 start pj0 rows / q00 columns into the stage
            const int ks0 = grp * KSG, pj0 = (4 * ks0) / Q, q00 = 4 * ks0 - pj0 * Q;
            abase[bf] ^= 16 * ks0;
            bbase[bf] += ST * (pj0 * RW + q00);
        }
        asm volatile("" : "+v"(abase[bf]));
        asm volatile("" : "+v"(bbase[bf]));
    }
    int bsp_cur = bbase[0] + (kk >= 2 ? 2 * RW - 2 * Q : 0), bsp_oth = bbase[1] + (kk >= 2 ? 2 * RW - 2 * Q : 0);
    if (HALF) { asm volatile("" : "+v"(bsp_cur)); asm volatile("" : "+v"(bsp_oth)); }

    // Operand registers: the 25 B values of a k-step live in ONE set -- kernel row r of the next
    // k-step is read into the registers of row r while the MFMAs of row r+1 run (an MFMA has taken
    // its operands when it issues; the row is needed again 20 MFMAs = 640 cycles later).  The A
    // value is double buffered.  Two reads per row: word 1, and words 2..5 as two b64.
    auto load_a = [&](const int ab, const int ks, float& a) __attribute__((always_inline)) {
        // (4 lj) ^ (4 ks) = 4 (lj ^ ks); volatile: computed here, every time (hoisted out of the
        // stage loop the 32 addresses would be spilled)
        int ao;                                                    // byte offsets
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ao) : "n"(16 * ks), "v"(ab));
        a = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + ao);
    };
    auto load_row = [&](const int bb_in, const int bs_in, const int ks, const int r, const int half, float (&bq)[25]) __attribute__((always_inline)) {
        const int pj = (4 * ks) / Q, q0 = (4 * ks) - pj * Q;
        const int bb = (HALF && q0 + 4 > Q) ? bs_in : bb_in;       // k-step across a row boundary
        const float* bp = smem + bb + (ST * pj + r) * RW + ST * q0;
        if (ST == 1) {
            // consecutive words of either parity: 4-byte reads
            if (half == 0) { bq[r * 5 + 0] = bp[1]; bq[r * 5 + 1] = bp[2]; }
            else { bq[r * 5 + 2] = bp[3]; bq[r * 5 + 3] = bp[4]; bq[r * 5 + 4] = bp[5]; }
        } else if (half == 0) {
            bq[r * 5 + 0] = bp[1];
        } else {
            const floatx2 c1 = *reinterpret_cast<const floatx2*>(bp + 2);
            const floatx2 c2 = *reinterpret_cast<const floatx2*>(bp + 4);
            bq[r * 5 + 1] = c1.x; bq[r * 5 + 2] = c1.y; bq[r * 5 + 3] = c2.x; bq[r * 5 + 4] = c2.y;
        }
    };
    auto stage_body = [&](const int ab, const int bb, const int bs, const int nbuf, const bool more, const int n0n,
                          const int p0n) __attribute__((always_inline)) {
        float av[2], bv[25];
        load_a(ab, 0, av[0]);
#pragma unroll
        for (int r = 0; r < 5; ++r) { load_row(bb, bs, 0, r, 0, bv); load_row(bb, bs, 0, r, 1, bv); }
        constexpr int KS = KSG;
        // (a 16-pixel stage -- 4-column maps -- has four k-steps for five DMA instructions: two slots per k-step)
        constexpr bool DMA2 = NG == 1 && NDMA > T::KS;
        static_assert(NDMA <= (DMA2 ? 2 : 1) * T::KS, "DMA slots of the stage");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (BIAS == 1) bsum += av[ks & 1];
#pragma unroll
            for (int tp = 0; tp < 25; ++tp) {
                const int r = tp / 5, sx = tp - 5 * r;
                if (r >= KV / 10 && r < KV % 10 && sx >= KV / 10 && sx < KV % 10)
                    acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks & 1], bv[tp], acc[tp], 0, 0, 0);
                // bias side 2: taps (1,1) (1,2), then (2,1) (2,2), right behind their last MFMA
                if (BIAS == 2 && tp == 7) bsum += bv[6] + bv[7];
                if (BIAS == 2 && tp == 12) bsum += bv[11] + bv[12];
                // row r-1 of the NEXT k-step goes into its registers during row r (r >= 1); row 4
                // follows during row 0 of the next k-step
                if (r >= 1 && ks + 1 < KS && (sx == 0 || sx == 2)) load_row(bb, bs, ks + 1, r - 1, sx >> 1, bv);
                if (r == 0 && ks >= 1 && (sx == 0 || sx == 2)) load_row(bb, bs, ks, 4, sx >> 1, bv);
                if (r == 2 && sx == 4 && ks + 1 < KS) load_a(ab, ks + 1, av[(ks + 1) & 1]);
                if (NG == 1 && DMA2) {
                    if ((r == 3 || r == 1) && sx == 4 && 2 * ks + (r == 3 ? 1 : 0) < NDMA) {
                        if (more) issue_dma(2 * ks + (r == 3 ? 1 : 0), nbuf, n0n, p0n);
                    }
                } else if (NG == 1) {
                    if (r == 3 && sx == 4 && ks < NDMA) {
                        if (more) issue_dma(ks, nbuf, n0n, p0n);
                    }
                } else if (sx == 4 && r < NG && ks * NG + r < NDMA) {
                    // (a group has 1 / NG of the k-steps for the same DMA slots: up to NG of them per k-step, one per
                    // row of taps)
                    if (more) issue_dma(ks * NG + r, nbuf, n0n, p0n);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const int tpf_mask = (1 << lg_tpf) - 1;
    // stage -> (frame, first row): shift / mask, or (GEN) the multiply-high division by the tiles per frame
    const int tpf_gen = (g.Hs + T::PT_H - 1) / T::PT_H;
    auto frame_of = [&](const int s) __attribute__((always_inline)) {
        if constexpr (GEN) return lg_tpf == 0 ? s : (int)__umulhi((unsigned)s, (unsigned)lg_tpf);
        else return s >> lg_tpf;
    };
    auto row_of = [&](const int s, const int f) __attribute__((always_inline)) {
        if constexpr (GEN) return (s - f * tpf_gen) * T::PT_H;
        else return (s & tpf_mask) * T::PT_H;
    };
    int st = blockIdx.y;
    if (st < n_stages) {
        const int f0 = frame_of(st);
#pragma unroll
        for (int d = 0; d < NDMA; ++d) issue_dma(d, 0, f0, row_of(st, f0));
    }
    // one loop trip = one stage; the base registers of the two LDS images swap after each trip
    int ab_cur = abase[0], ab_oth = abase[1], bb_cur = bbase[0], bb_oth = bbase[1];
    int cur = 0;
#ifdef W4_LOOP_ALIGN
    asm volatile(".p2align " W4_STR(W4_LOOP_ALIGN) "\n .rept " W4_STR(W4_LOOP_SHIFT) "\n s_nop 0\n .endr" ::: "memory");
#endif
    for (; st < n_stages; st += splits) {
        // own DMAs of this stage have landed; after the barrier everyone's have, and every wave
        // is done reading the other image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nx = st + splits;
        const int fx = frame_of(nx);
        stage_body(ab_cur, bb_cur, bsp_cur, cur ^ 1, nx < n_stages, fx, row_of(nx, fx));
        cur ^= 1;
        int tmp = ab_cur; ab_cur = ab_oth; ab_oth = tmp;
        tmp = bb_cur; bb_cur = bb_oth; bb_oth = tmp;
        if (HALF) { tmp = bsp_cur; bsp_cur = bsp_oth; bsp_oth = tmp; }
    }

    if (BIAS != 0) {
        // lanes lj + 16 kk hold the four pixel phases of one channel: fixed-order butterfly
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        float* bdst = bias_part + ((size_t)blockIdx.y * NG + grp) * nbias;
        if (BIAS == 1 && btile == 0 && bblk == 0 && kk == 0) {
            const int a = a0 + ablk * 16 + lj;
            if (a < g.Cs) bdst[a] = bsum;
        }
        if (BIAS == 2 && atile == 0 && ablk == 0 && kk == 0) {
            const int bb = b0 + bblk * 16 + lj;
            if (bb < g.Cb) bdst[bb] = bsum;
        }
    }

    // partial tile -> scratch [split][tap][a][b]; lane holds D[i = 4*kk + e][j = lj]
    float* dst = part + ((size_t)blockIdx.y * NG + grp) * 25 * g.Cs * g.Cb;
    const int b = b0 + bblk * 16 + lj;
    if (b < g.Cb) {
#pragma unroll
        for (int tp = 0; tp < 25; ++tp) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = a0 + ablk * 16 + 4 * kk + e;
                if (a < g.Cs) dst[((size_t)tp * g.Cs + a) * g.Cb + b] = acc[tp][e];
            }
        }
    }
}

static bool wgrad4_tile(const BnGeom& g, Wgrad4Tile* t, size_t* lds_bytes) {
    const int lgQ = ilog2_exact_w4(g.Ws), lgP = ilog2_exact_w4(g.Hs);
    if (lgQ < 2 || lgQ > 5 || lgP < 0) return false;
    if (g.pt != 1 || g.pl != 1) return false;           // the pair layout assumes offset 1
    if ((g.Wb & 3) != 0 || g.Wb < 2 * g.Ws) return false;   // 16-byte aligned image rows
    const int PQ = g.Hs * g.Ws;
    if (PQ >= W4_TPX) {
        t->F = 1;
        t->PT_H = W4_TPX / g.Ws;
    } else {
        t->F = W4_TPX / PQ;
        t->PT_H = g.Hs;
    }
    t->lgPTQ = ilog2_exact_w4(t->PT_H * g.Ws);
    t->tiles_per_frame = (t->F == 1) ? g.Hs / t->PT_H : 1;
    t->n_stages = ((g.N + t->F - 1) / t->F) * t->tiles_per_frame;
    t->IH = 2 * (t->PT_H - 1) + 5;
    const int RW = 2 * g.Ws + 8;
    t->FSb = t->IH * RW;
    t->rows_per_b = t->F * t->IH;
    t->row_groups = t->rows_per_b * (RW / 4);
    int gpb = t->row_groups;
    while ((gpb & 1) == 0 || gpb < t->row_groups) ++gpb;    // BCH = 4 * odd: conflict-free b64
    // (4*odd) mod 64 in {4,12,...,60}: 16 channels x 4 words cover the 64 banks exactly once
    t->GPB = gpb;
    t->big_groups = W4_TB * gpb;
    if (t->big_groups > W4_THREADS * W4_SLICES) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;   // 32-bit offsets
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    t->inv_gpb = 1.0f / (float)gpb;
    t->inv_c4 = 1.0f / (float)(RW / 4);
    t->inv_ih = 1.0f / (float)t->IH;
    // the DMA writes whole wave rows of 64 groups: round the image up so the last one stays inside
    const int words = W4_TA * W4_TPX + 4 * ((t->big_groups + 63) & ~63);
    t->buf_floats = words;
    *lds_bytes = (size_t)2 * words * 4;
    return *lds_bytes <= W4_MAX_LDS;
}

static int wgrad4_splits(const BnGeom& g, const Wgrad4Tile& t) {
    const int tiles = ((g.Cs + W4_TA - 1) / W4_TA) * ((g.Cb + W4_TB - 1) / W4_TB);
    int splits = (256 + tiles - 1) / tiles;         // one workgroup per CU
    if (splits > t.n_stages) splits = t.n_stages;
    if (splits < 1) splits = 1;
    return splits;
}

// stages inside one frame, geometry = the compile-time one of W4S<LGQ>
static bool wgrad4s_ok(const BnGeom& g, const Wgrad4Tile& t) {
    static int disabled = -1;                          // BN_WGRAD4S=0: the general kernel only
    if (disabled < 0) { const char* e = bn_tune_env("BN_WGRAD4S"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    const int lgq = ilog2_exact_w4(g.Ws);
    if (t.F != 1 || lgq < 3 || lgq > 5) return false;
    if (g.Hb != 2 * g.Hs) return false;                // the row classes of the DMA assume it
    if (ilog2_exact_w4(t.tiles_per_frame) < 0) return false;
    return true;
}

// column windows (CW instantiations): the widest instantiated window that divides the map's width, 0 = none
static inline int w4g_window(const BnGeom& g) {
    static const int wins[] = {44, 40, 36, 32, 28, 24};
    if (g.stride != 2 || g.Ws <= 44 || (g.Ws & 3)) return 0;
    for (int q : wins)
        if (g.Ws % q == 0 && g.Ws / q <= 8) return q;
    return 0;
}
// rows of a stage of the GEN instantiations (W4S::PT_H)
static inline int w4g_pth(const BnGeom& g) {
    if (w4g_window(g)) return W4_TPX / w4g_window(g);
    if ((g.Ws & 3) == 2) return (W4_TPX / g.Ws) & ~1;          // an even number of rows (W4S<Q, 2, PTH>)
    return (g.stride == 2 && (g.Ws == 8 || g.Ws == 4) && g.Hs <= 4) ? 4 : W4_TPX / g.Ws;
}
// maps that are no powers of two on the streamlined kernel (GEN instantiations): widths the kernel is
// instantiated for, any height, big map exactly twice the small one
static const int W4G_WIDTHS[] = {4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40, 44};   // (4: maps of at most four rows)
static bool wgrad4g_ok(const BnGeom& g) {
    static int disabled = -1;                          // BN_WGRAD4G=0: off
    if (disabled < 0) { const char* e = bn_tune_env("BN_WGRAD4G"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (g.pt != 1 || g.pl != 1 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return false;
    bool width = w4g_window(g) != 0;
    for (int q : W4G_WIDTHS) width = width || q == g.Ws;
    if (!width) return false;
    if ((g.Ws & 3) == 2 && (g.Hs & 1)) return false;   // half-row k-steps: rows come in pairs
    if (g.Ws == 4 && g.Hs > 4) return false;           // (round 6: the 4-column instantiation has 4-row stages)
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;   // 32-bit offsets
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    const int tpf = (g.Hs + w4g_pth(g) - 1) / w4g_pth(g);
    const int ncb = w4g_window(g) ? g.Ws / w4g_window(g) : 1;
    return (size_t)g.N * tpf * ncb < (1u << 20);       // multiply-high division of the stage index
}
// stride 1 (5x5 taps, offsets up to 4): power-of-two widths up to 64, any height
static bool wgrad4g1_ok(const BnGeom& g) {
    static int disabled = -1;                          // BN_WGRAD4G1=0: off
    if (disabled < 0) { const char* e = bn_tune_env("BN_WGRAD4G1"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (g.stride != 1 || g.R != 5 || g.S != 5 || g.pt > 4 || g.pl > 4) return false;
    static const int widths[] = {8, 12, 16, 20, 24, 32, 40, 48, 64};
    bool width = false;
    for (int q : widths) width = width || q == g.Ws;
    if (!width) return false;
    if ((g.Wb & 3) != 0) return false;                 // 16-byte rows of the big map
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7ffffff0ull - (size_t)4 * g.Wb * 4) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    const int tpf = (g.Hs + w4g_pth(g) - 1) / w4g_pth(g);
    return (size_t)g.N * tpf < (1u << 20);
}
static int wgrad4g_stages(const BnGeom& g) {
    const int pth = w4g_pth(g);
    const int ncb = (g.stride == 2 && w4g_window(g)) ? g.Ws / w4g_window(g) : 1;
    return g.N * ncb * ((g.Hs + pth - 1) / pth);
}

// wave groups of the stride-1 instantiations (k_wgrad4s_mfma<.., NA, NB>): layers with at most 32 small-side channels
// on 16- / 32- / 64-pixel-wide maps; 1 = none
static int w4g1_groups(const BnGeom& g, int* na, int* nb) {
    *na = 4; *nb = 2;
    static int off = -1;                               // BN_W4_GROUPS=0: off (tuning build)
    if (off < 0) { const char* e = bn_tune_env("BN_W4_GROUPS"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || g.Cs > 32) return 1;
    // (stride 2: the power-of-two instantiations, 8 / 16 / 32-wide small maps -- the caller has checked wgrad4s_ok)
    if (g.stride == 1 ? (g.Ws != 16 && g.Ws != 32 && g.Ws != 64) : (g.Ws != 8 && g.Ws != 16 && g.Ws != 32)) return 1;
    if (g.KV == 4 && g.K0 != 1) return 1;              // (4x4 kernels: instantiated without groups; 3x3: with)
    *na = 2;
    *nb = g.Cb <= 16 ? 1 : 2;
    return 8 / (*na * *nb);
}

BnFastPlan bn_wgrad4_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || (g.stride != 2 && g.stride != 1)) return p;
    if (g.Cs < 16 || g.Cb < 16) return p;
    static int disabled = -1;                          // BN_WGRAD4=0: fall back to the dword-DMA kernel
    if (disabled < 0) { const char* e = bn_tune_env("BN_WGRAD4"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return p;
    Wgrad4Tile t;
    size_t lds = 0;
    if (g.stride == 1) {
        if (!wgrad4g1_ok(g)) return p;
        t.n_stages = wgrad4g_stages(g);
        p.supported = true;
        p.variant = 6;
        int na, nb;
        p.d = wgrad4_splits(g, t) * w4g1_groups(g, &na, &nb);     // partial tiles: one per (split, wave group)
        p.ws_bytes = (size_t)p.d * (25 * g.Cs * g.Cb + (g.Cs > g.Cb ? g.Cs : g.Cb)) * sizeof(float);
        static char names_1[16][40];
        const int slot1 = (g.Ws / 4 - 1) & 15;
        snprintf(names_1[slot1], sizeof(names_1[slot1]), "k_wgrad4s_mfma<%d, stride 1>", g.Ws);
        p.kernel_name = names_1[slot1];
        return p;
    }
    if (!wgrad4_tile(g, &t, &lds)) {
        if (!wgrad4g_ok(g)) return p;
        t.n_stages = wgrad4g_stages(g);
        p.supported = true;
        p.variant = 5;
        p.d = wgrad4_splits(g, t);
        p.ws_bytes = (size_t)p.d * (25 * g.Cs * g.Cb + (g.Cs > g.Cb ? g.Cs : g.Cb)) * sizeof(float);
        static char names_g[32][64];
        const int slot = (g.Ws / 4) & 31;
        if (w4g_window(g))
            snprintf(names_g[slot], sizeof(names_g[slot]), "k_wgrad4s_mfma<%d, gen> x %d windows", w4g_window(g),
                     g.Ws / w4g_window(g));
        else
            snprintf(names_g[slot], sizeof(names_g[slot]), "k_wgrad4s_mfma<%d, gen>", g.Ws);
        p.kernel_name = names_g[slot];
        return p;
    }
    p.supported = true;
    p.variant = 4;
    p.d = wgrad4_splits(g, t);
    if (wgrad4s_ok(g, t)) {
        int na, nb;
        p.d *= w4g1_groups(g, &na, &nb);                // partial tiles: one per (split, wave group)
    }
    // partial dW tiles + partial bias rows (either side) of every split
    p.ws_bytes = (size_t)p.d * (25 * g.Cs * g.Cb + (g.Cs > g.Cb ? g.Cs : g.Cb)) * sizeof(float);
    static const char* const names[6] = {"k_wgrad4_mfma<?>", "k_wgrad4_mfma<?>", "k_wgrad4_mfma<2>",
                                         "k_wgrad4_mfma<3>", "k_wgrad4_mfma<4>", "k_wgrad4_mfma<5>"};
    // (the second template argument of the streamlined kernel, the fused bias side, is not part
    // of the plan's name)
    static const char* const names_s[6] = {"", "", "", "k_wgrad4s_mfma<8>", "k_wgrad4s_mfma<16>",
                                           "k_wgrad4s_mfma<32>"};
    const int lgq = ilog2_exact_w4(g.Ws);
    p.kernel_name = wgrad4s_ok(g, t) ? names_s[lgq] : names[(lgq >= 2 && lgq <= 5) ? lgq : 0];
    return p;
}

template <int LGQ>
static int launch_wgrad4(dim3 grid, size_t lds, hipStream_t st, const float* small,
                         const float* big, float* part, float* bias_part, const BnGeom& g,
                         const Wgrad4Tile& t) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_wgrad4_mfma<LGQ>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, W4_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    BN_LAUNCH_MAIN(k_wgrad4_mfma<LGQ>, grid, dim3(W4_THREADS), lds, st, small, big, part,
                       bias_part, g, t);
    BN_LAUNCH_CHECK();
    return 0;
}

template <int Q, int BIAS, bool GEN, int KV, int ST = 2, int PTH = 0, bool CW = false, int NA = 4, int NB = 2>
static int launch_wgrad4s(dim3 grid, hipStream_t st, const float* small, const float* big,
                          float* part, float* bias_part, const BnGeom& g, int n_stages, int splits,
                          int lg_tpf, int nbias, int ncb = 1) {
    using TS = W4S<Q, ST, PTH>;
    static_assert((size_t)2 * TS::BUFW * 4 <= W4_MAX_LDS, "two stage images in LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_wgrad4s_mfma<Q, BIAS, GEN, KV, ST, PTH, CW, NA, NB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, W4_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    constexpr size_t lds = (size_t)2 * TS::BUFW * 4;
    BN_LAUNCH_MAIN((k_wgrad4s_mfma<Q, BIAS, GEN, KV, ST, PTH, CW, NA, NB>), grid, dim3(W4_THREADS), lds, st, small, big, part,
                       bias_part, g, n_stages, splits, lg_tpf, nbias, ncb);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_wgrad4(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                     const BnGeom& g, int accumulate, void* ws, hipStream_t st, float* db,
                     int bias_side, bool* bias_done) {
    Wgrad4Tile t;
    size_t lds = 0;
    const bool s1 = plan.variant == 6;
    const bool gen = plan.variant == 5 || s1;
    if (gen) {
        if (s1 ? !wgrad4g1_ok(g) : !wgrad4g_ok(g)) return BN_E_SHAPE;
        t.n_stages = wgrad4g_stages(g);
    } else if (!wgrad4_tile(g, &t, &lds)) return BN_E_SHAPE;
    t.splits = plan.d;
    // the taps {1,2}^2 cover the big image only if it is not larger than 2x the small one
    // (the GEN instantiations leave the big side's sums to the caller's channel-sum kernels: with the
    // row numbers of the DMA groups on top, the fused form runs out of registers)
    const bool fuse = db && (bias_side == 1 || (bias_side == 2 && !gen && g.Hb <= 2 * g.Hs &&
                                                g.Wb <= 2 * g.Ws));
    t.bias_side = fuse ? bias_side : 0;
    t.nbias = bias_side == 1 ? g.Cs : g.Cb;
    float* bias_part = (float*)ws + (size_t)t.splits * 25 * g.Cs * g.Cb;
    const int tiles = ((g.Cs + W4_TA - 1) / W4_TA) * ((g.Cb + W4_TB - 1) / W4_TB);
    dim3 grid(tiles, t.splits);
    int rc = BN_E_SHAPE;
    if (s1) {
        const int pth = w4g_pth(g), tpf = (g.Hs + pth - 1) / pth;
        const int magic = tpf == 1 ? 0 : (int)(unsigned)((1ull << 32) / (unsigned)tpf + 1ull);
        int na = 4, nb = 2;
        const int ng = w4g1_groups(g, &na, &nb);
        if (ng > 1) {
            if (t.splits % ng) return BN_E_BADARG;
            dim3 gridg(tiles, t.splits / ng);
            const bool k3g = g.KV == 4;
#define W4GG_CASE(QV, B, A_, B_)                                                                 \
    if (g.Ws == QV && t.bias_side == B && na == A_ && nb == B_)                                  \
        rc = k3g ? launch_wgrad4s<QV, B, true, 14, 1, 0, false, A_, B_>(gridg, st, small, big, (float*)ws, bias_part, g, \
                                                                        t.n_stages, t.splits / ng, magic, t.nbias) \
                 : launch_wgrad4s<QV, B, true, 5, 1, 0, false, A_, B_>(gridg, st, small, big, (float*)ws, bias_part, g, \
                                                                       t.n_stages, t.splits / ng, magic, t.nbias);
#define W4GG_ALL(QV) W4GG_CASE(QV, 0, 2, 2) W4GG_CASE(QV, 1, 2, 2) W4GG_CASE(QV, 0, 2, 1) W4GG_CASE(QV, 1, 2, 1)
            W4GG_ALL(16) W4GG_ALL(32) W4GG_ALL(64)
#undef W4GG_ALL
#undef W4GG_CASE
        } else {
        const bool k3 = g.KV == 4 && g.K0 == 1;
#define W4G1_CASE(QV, B)                                                                         \
    if (g.Ws == QV && t.bias_side == B)                                                          \
        rc = k3 ? launch_wgrad4s<QV, B, true, 14, 1>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                     t.n_stages, t.splits, magic, t.nbias)       \
                : launch_wgrad4s<QV, B, true, 5, 1>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                    t.n_stages, t.splits, magic, t.nbias);
#define W4G1_ALL(QV) W4G1_CASE(QV, 0) W4G1_CASE(QV, 1)
        W4G1_ALL(8) W4G1_ALL(12) W4G1_ALL(16) W4G1_ALL(20) W4G1_ALL(24) W4G1_ALL(32) W4G1_ALL(40) W4G1_ALL(48)
        W4G1_ALL(64)
#undef W4G1_ALL
#undef W4G1_CASE
        }
    } else if (gen) {
        // stage -> frame by multiply-high: 2^32 / tiles + 1 (exact below 2^32 / tiles stages)
        const int pth = w4g_pth(g), tpf = (g.Hs + pth - 1) / pth;
        const int magic = tpf == 1 ? 0 : (int)(unsigned)((1ull << 32) / (unsigned)tpf + 1ull);
        // (the 3x3 window of taps -- BnGeom::K0 -- is instantiated for the plain widths only)
        const bool k3 = g.KV == 4 && g.K0 == 1;
#define W4G_CASE(QV, B)                                                                          \
    if (g.Ws == QV && t.bias_side == B)                                                          \
        rc = k3 ? launch_wgrad4s<QV, B, true, 14>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                  t.n_stages, t.splits, magic, t.nbias)          \
                : launch_wgrad4s<QV, B, true, 5>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                 t.n_stages, t.splits, magic, t.nbias);
        if (w4g_window(g)) {
            const int qw = w4g_window(g), ncb = g.Ws / qw;
#define W4W_CASE(QV, B)                                                                          \
    if (qw == QV && t.bias_side == B)                                                            \
        rc = launch_wgrad4s<QV, B, true, 5, 2, 0, true>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                        t.n_stages, t.splits, magic, t.nbias, ncb);
            W4W_CASE(24, 0) W4W_CASE(24, 1) W4W_CASE(28, 0) W4W_CASE(28, 1) W4W_CASE(32, 0) W4W_CASE(32, 1)
            W4W_CASE(36, 0) W4W_CASE(36, 1) W4W_CASE(40, 0) W4W_CASE(40, 1) W4W_CASE(44, 0) W4W_CASE(44, 1)
#undef W4W_CASE
        } else if ((g.Ws & 3) == 2) {
#define W4H_CASE(QV, B)                                                                          \
    if (g.Ws == QV && t.bias_side == B)                                                          \
        rc = launch_wgrad4s<QV, B, true, 5, 2, ((W4_TPX / QV) & ~1)>(grid, st, small, big, (float*)ws, \
                                                 bias_part, g, t.n_stages, t.splits, magic, t.nbias);
            W4H_CASE(6, 0) W4H_CASE(6, 1) W4H_CASE(10, 0) W4H_CASE(10, 1) W4H_CASE(14, 0) W4H_CASE(14, 1)
#undef W4H_CASE
        } else if (g.Ws == 4 && pth == 4) {
#define W4Q_CASE(B, K)                                                                           \
    if (t.bias_side == B && (k3 ? 14 : 5) == K)                                                  \
        rc = launch_wgrad4s<4, B, true, K, 2, 4>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                 t.n_stages, t.splits, magic, t.nbias);
            W4Q_CASE(0, 5) W4Q_CASE(1, 5) W4Q_CASE(0, 14) W4Q_CASE(1, 14)
#undef W4Q_CASE
        } else if (g.Ws == 8 && pth == 4) {
#define W4P_CASE(B, K)                                                                           \
    if (t.bias_side == B && (k3 ? 14 : 5) == K)                                                  \
        rc = launch_wgrad4s<8, B, true, K, 2, 4>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                 t.n_stages, t.splits, magic, t.nbias);
            W4P_CASE(0, 5) W4P_CASE(1, 5) W4P_CASE(0, 14) W4P_CASE(1, 14)
#undef W4P_CASE
        } else {
#define W4G_ALL(QV) W4G_CASE(QV, 0) W4G_CASE(QV, 1)
        W4G_ALL(8) W4G_ALL(12) W4G_ALL(16) W4G_ALL(20) W4G_ALL(24) W4G_ALL(28) W4G_ALL(32) W4G_ALL(36)
        W4G_ALL(40) W4G_ALL(44)
#undef W4G_ALL
#undef W4G_CASE
        }
    } else if (wgrad4s_ok(g, t)) {
        const int lgq = ilog2_exact_w4(g.Ws), lg_tpf = ilog2_exact_w4(t.tiles_per_frame);
        int na = 4, nb = 2;
        const int ng = w4g1_groups(g, &na, &nb);
        if (ng > 1) {
            if (t.splits % ng) return BN_E_BADARG;
            dim3 gridg(tiles, t.splits / ng);
            const bool k3g = g.KV == 4;
#define W4SG_CASE(L, B, A_, B_)                                                                  \
    if (lgq == L && t.bias_side == B && na == A_ && nb == B_)                                    \
        rc = k3g ? launch_wgrad4s<(1 << L), B, false, 14, 2, 0, false, A_, B_>(gridg, st, small, big, (float*)ws,      \
                                                          bias_part, g, t.n_stages, t.splits / ng, lg_tpf, t.nbias) \
                 : launch_wgrad4s<(1 << L), B, false, 5, 2, 0, false, A_, B_>(gridg, st, small, big, (float*)ws,       \
                                                          bias_part, g, t.n_stages, t.splits / ng, lg_tpf, t.nbias);
#define W4SG_ALL(L) W4SG_CASE(L, 0, 2, 2) W4SG_CASE(L, 1, 2, 2) W4SG_CASE(L, 2, 2, 2) W4SG_CASE(L, 0, 2, 1)      \
                    W4SG_CASE(L, 1, 2, 1) W4SG_CASE(L, 2, 2, 1)
            W4SG_ALL(3) W4SG_ALL(4) W4SG_ALL(5)
#undef W4SG_ALL
#undef W4SG_CASE
        } else {
#define W4S_CASE(L, B)                                                                         \
    if (lgq == L && t.bias_side == B)                                                          \
        rc = g.KV == 4 && g.K0 == 1                                                            \
            ? launch_wgrad4s<(1 << L), B, false, 14>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                     t.n_stages, t.splits, lg_tpf, t.nbias)    \
            : g.KV == 4                                                                        \
            ? launch_wgrad4s<(1 << L), B, false, 4>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                    t.n_stages, t.splits, lg_tpf, t.nbias)     \
            : launch_wgrad4s<(1 << L), B, false, 5>(grid, st, small, big, (float*)ws, bias_part, g, \
                                                    t.n_stages, t.splits, lg_tpf, t.nbias);
        W4S_CASE(3, 0) W4S_CASE(3, 1) W4S_CASE(3, 2) W4S_CASE(4, 0) W4S_CASE(4, 1) W4S_CASE(4, 2)
        W4S_CASE(5, 0) W4S_CASE(5, 1) W4S_CASE(5, 2)
#undef W4S_CASE
        }
    } else
    switch (ilog2_exact_w4(g.Ws)) {
        case 2: rc = launch_wgrad4<2>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 3: rc = launch_wgrad4<3>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 4: rc = launch_wgrad4<4>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 5: rc = launch_wgrad4<5>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        default: break;
    }
    if (rc) return rc;
    if (fuse) {
        // one launch adds up the partial tiles and the partial bias rows
        rc = bn_launch_sum_partials_pair((const float*)ws, dw, 25 * g.Cs * g.Cb, g.Cs * g.Cb, 25,
                                         bias_part, db, t.nbias, t.splits, accumulate, st);
        if (bias_done) *bias_done = true;
        return rc;
    }
    return bn_launch_sum_partials((const float*)ws, dw, 25 * g.Cs * g.Cb, t.splits, accumulate,
                                  g.Cs * g.Cb, 25, st);
}
