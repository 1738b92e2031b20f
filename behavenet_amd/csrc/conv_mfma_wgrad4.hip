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

BnFastPlan bn_wgrad4_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || g.stride != 2) return p;
    if (g.Cs < 16 || g.Cb < 16) return p;
    static int disabled = -1;                          // BN_WGRAD4=0: fall back to the dword-DMA kernel
    if (disabled < 0) { const char* e = bn_tune_env("BN_WGRAD4"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return p;
    Wgrad4Tile t;
    size_t lds = 0;
    if (!wgrad4_tile(g, &t, &lds)) return p;
    p.supported = true;
    p.variant = 4;
    p.d = wgrad4_splits(g, t);
    // partial dW tiles + partial bias rows (either side) of every split
    p.ws_bytes = (size_t)p.d * (25 * g.Cs * g.Cb + (g.Cs > g.Cb ? g.Cs : g.Cb)) * sizeof(float);
    static const char* const names[6] = {"k_wgrad4_mfma<?>", "k_wgrad4_mfma<?>", "k_wgrad4_mfma<2>",
                                         "k_wgrad4_mfma<3>", "k_wgrad4_mfma<4>", "k_wgrad4_mfma<5>"};
    const int lgq = ilog2_exact_w4(g.Ws);
    p.kernel_name = names[(lgq >= 2 && lgq <= 5) ? lgq : 0];
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
    hipLaunchKernelGGL(k_wgrad4_mfma<LGQ>, grid, dim3(W4_THREADS), lds, st, small, big, part,
                       bias_part, g, t);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_wgrad4(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                     const BnGeom& g, int accumulate, void* ws, hipStream_t st, float* db,
                     int bias_side, bool* bias_done) {
    Wgrad4Tile t;
    size_t lds = 0;
    if (!wgrad4_tile(g, &t, &lds)) return BN_E_SHAPE;
    t.splits = plan.d;
    // the taps {1,2}^2 cover the big image only if it is not larger than 2x the small one
    const bool fuse = db && (bias_side == 1 || (bias_side == 2 && g.Hb <= 2 * g.Hs &&
                                                g.Wb <= 2 * g.Ws));
    t.bias_side = fuse ? bias_side : 0;
    t.nbias = bias_side == 1 ? g.Cs : g.Cb;
    float* bias_part = (float*)ws + (size_t)t.splits * 25 * g.Cs * g.Cb;
    const int tiles = ((g.Cs + W4_TA - 1) / W4_TA) * ((g.Cb + W4_TB - 1) / W4_TB);
    dim3 grid(tiles, t.splits);
    int rc = BN_E_SHAPE;
    switch (ilog2_exact_w4(g.Ws)) {
        case 2: rc = launch_wgrad4<2>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 3: rc = launch_wgrad4<3>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 4: rc = launch_wgrad4<4>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        case 5: rc = launch_wgrad4<5>(grid, lds, st, small, big, (float*)ws, bias_part, g, t); break;
        default: break;
    }
    if (rc) return rc;
    rc = bn_launch_sum_partials((const float*)ws, dw, 25 * g.Cs * g.Cb, t.splits, accumulate,
                                g.Cs * g.Cb, 25, st);
    if (rc) return rc;
    if (fuse) {
        rc = bn_launch_sum_partials(bias_part, db, t.nbias, t.splits, accumulate, 0, 0, st);
        if (bias_done) *bias_done = true;
    }
    return rc;
}
