// family 3: weight gradients on the matrix cores (kernel 5x5, stride 2):
//   dW[a][b][r][s] = sum_{n,p,q} small[n,a,p,q] * big[n,b,2p+r-pt,2q+s-pl]
//
// GEMM view per tap: D[a][b] += A[a][pixel] * B[pixel][b], reduction over the N*P*Q pixels.
// v_mfma_f32_16x16x4_f32 (4 accumulator registers) lets one wave keep all 25 taps of a 16x16
// (a,b) block resident: 100 accumulator registers, one A read + 25 B reads per 25 MFMAs.
// A workgroup is 8 waves = 4(a) x 2(b) blocks -> a 64 x 32 tile of dW, and walks a strided set
// of 64-pixel "stages".
//
// Staging is pure LDS-DMA (buffer_load ... lds): every thread's k-th element IS LDS word
// tid + 512*k of the stage image, so a wave's 64 lanes land contiguously and only the per-lane
// SOURCE offset is computed; out-of-range offsets (zero padding, halo, channel / frame tails)
// arrive as 0.0f.  No staging registers, no ds_write pass; the image is double buffered
// (2 x 78 KB of the 160 KB LDS) so the DMA of stage i+1 runs behind the MFMAs of stage i with
// one barrier per stage.  The big-side tile is stored column-parity-split
// (x -> (x&1)*HALF + x/2) so the stride-2 gather of four consecutive pixels is bank-conflict
// free.  Partial tiles go to scratch as [split][tap][a][b]; k_sum_partials combines them in
// fixed order (deterministic, no atomics) into dW[a][b][tap] (+= when accumulating).
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"
#include "bn_reduce.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

#define WG_THREADS 512
#define WG_TA 64            // a-channels per workgroup tile
#define WG_TB 32            // b-channels per workgroup tile
#define WG_TPX 64           // small-image pixels per stage
#define WG_SP (WG_TPX + 2)  // small tile row stride (== 2 mod 32: conflict-free A reads)
#define WG_SLICES 8         // DMA slices per stage = MFMA-loop trips (2 k-steps of 4 pixels each)
#define WG_KBS 4            // big-tile words per thread per slice
#define WG_KB (WG_SLICES * WG_KBS)   // max big-tile words per thread per stage (32)
#define WG_KS ((WG_TA * WG_TPX) / WG_THREADS)   // small-tile words per thread per stage (8)
#define WG_MAX_LDS (160 * 1024)
#define WG_OOB 0x7fffffff

static inline int ilog2_exact_wg(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

struct WgradTile {
    int F, PT_H, lgQ, lgPTQ;       // pixel stage = F frames x PT_H rows x Q columns (64 pixels)
    int tiles_per_frame;           // P / PT_H (1 when F > 1)
    int n_stages;                  // total stages = ceil(N/F) * tiles_per_frame
    int IH, HALF, RW;              // big tile rows per frame, parity-half length, row stride
    int FSb;                       // per-frame stride inside a channel (IH * RW)
    int BCH;                       // per-channel stride (== 2 mod 32: conflict-free B reads)
    int big_words;                 // WG_TB * BCH
    float inv_rw, inv_bch, inv_ih; // reciprocals for the source-offset decode
    int rows_per_b;                // F * IH
    int splits;                    // reduction splits (gridDim.y)
    int buf_floats;                // one LDS stage image: WG_TA*WG_SP + WG_TB*BCH (16-B multiple)
    int dbg;                       // BN_WGRAD_DBG experiments (0 in production)
};

template <int LGQ>
__global__ __launch_bounds__(WG_THREADS, 2) void k_wgrad_mfma(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    BnGeom g, WgradTile t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ablk = wv >> 1, bblk = wv & 1;
    const int lj = lane & 15, kk = lane >> 4;

    const int n_btiles = (g.Cb + WG_TB - 1) / WG_TB;
    const int atile = blockIdx.x / n_btiles, btile = blockIdx.x - atile * n_btiles;
    const int a0 = atile * WG_TA, b0 = btile * WG_TB;
    // small-image width is a template parameter: the 25 tap offsets r*RW + (s&1)*HALF + (s>>1)
    // of the B reads become instruction immediates on one base address per k-step
    constexpr int Q = 1 << LGQ, T_HALF = Q + 2, T_RW = 2 * Q + 4;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;

    floatx4 acc[25];
#pragma unroll
    for (int tp = 0; tp < 25; ++tp) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rs_small = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HWb * 4), 0x00020000);

    // queue slice `j` (of WG_SLICES) of the LDS-DMA of one stage into image `buf`: small-tile
    // word k = j and big-tile words k = WG_KBS*j .. WG_KBS*j + WG_KBS-1 of every thread.  The
    // slices are spread over the MFMA loop of the previous stage: a dword-granular DMA costs the
    // CU's address unit tens of cycles per wave instruction, and 8 waves queueing all ~39 of
    // theirs at the top of a stage kept the matrix cores waiting for the queue to drain.
    auto issue_dma = [&](int st, int buf, int j) {
        const int grp = st / t.tiles_per_frame;
        const int n0 = grp * t.F;
        const int p0 = (st - grp * t.tiles_per_frame) * t.PT_H;
        float* sl = smem + buf * t.buf_floats;
        float* bl = sl + WG_TA * WG_SP;
        // small tile: word (a = 8k + wv, pix = lane); a stage's pixels are contiguous per frame
        {
            const int k = j;
            const int a = 8 * k + wv;
            const int f = lane >> t.lgPTQ;
            const int rem = lane & ((1 << t.lgPTQ) - 1);
            const bool ok = (a0 + a < g.Cs) && (n0 + f < g.N);
            const int off = (((n0 + f) * g.Cs + a0 + a) * PQ + p0 * Q + rem) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_small, sl + a * WG_SP, 4,
                                                     ok ? off : WG_OOB, 0, 0, 0);
        }
        // big tile: word e = tid + 512k of bl[b][BCH]:  b = e / BCH, within = e % BCH ->
        // (row rr = f*IH + y, parity-split column xx) -> source pixel (hb, wb)
#pragma unroll
        for (int kb = 0; kb < WG_KBS; ++kb) {
            const int k = WG_KBS * j + kb;
            if (WG_THREADS * k + 64 * wv >= t.big_words) break;      // wave-uniform
            int e = tid + WG_THREADS * k;
            asm volatile("" : "+v"(e));   // keep the decode next to its load (no hoisting)
            const int b = (int)(((float)e + 0.5f) * t.inv_bch);
            const int within = e - b * t.BCH;
            const int rr = (int)(((float)within + 0.5f) * t.inv_rw);
            const int xx = within - rr * T_RW;
            const int f = (t.F == 1) ? 0 : (int)(((float)rr + 0.5f) * t.inv_ih);
            const int y = rr - f * t.IH;
            const int par = xx >= T_HALF ? 1 : 0;
            const int x = 2 * (xx - par * T_HALF) + par;
            const int hb = 2 * p0 - g.pt + y, wb = x - g.pl;
            const bool ok = (b < WG_TB) && (rr < t.rows_per_b) && (b0 + b < g.Cb) &&
                            (n0 + f < g.N) && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            const int off = ((((n0 + f) * g.Cb + b0 + b) * g.Hb + hb) * g.Wb + wb) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, bl + WG_THREADS * k + 64 * wv, 4,
                                                     ok ? off : WG_OOB, 0, 0, 0);
        }
    };

    const int a_off = (ablk * 16 + lj) * WG_SP + kk;
    const int b_off = WG_TA * WG_SP + (bblk * 16 + lj) * t.BCH;

    int st = blockIdx.y;
    int cur = 0;
    if (st < t.n_stages) {
#pragma unroll 1
        for (int j = 0; j < WG_SLICES; ++j) issue_dma(st, 0, j);
    }
    for (; st < t.n_stages; st += t.splits) {
        // own DMAs of this stage have landed; after the barrier everyone's have, and every wave
        // is done reading the other image (it was computed from in the previous trip)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(t.dbg & 2)) __syncthreads();
        const bool more = (st + t.splits < t.n_stages) && !(t.dbg & 1);

        const float* ap = smem + cur * t.buf_floats + a_off;
        const float* bp = smem + cur * t.buf_floats + b_off;
#pragma unroll 1
        for (int j = 0; j < WG_SLICES; ++j) {
        if (more) issue_dma(st + t.splits, cur ^ 1, j);
#pragma unroll
        for (int ks = 2 * j; ks < 2 * j + 2; ++ks) {
            const int pix = 4 * ks + kk;
            const int f = pix >> t.lgPTQ;
            const int rem = pix & ((1 << t.lgPTQ) - 1);
            const int pj = rem >> LGQ, qj = rem & (Q - 1);
            const float av = ap[4 * ks];
            const float* bq = bp + f * t.FSb + (2 * pj) * T_RW + qj;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    const float bv = bq[r * T_RW + (s & 1) * T_HALF + (s >> 1)];
                    acc[r * 5 + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[r * 5 + s],
                                                                          0, 0, 0);
                }
            }
        }
        }
        cur ^= 1;
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

static bool wgrad_tile(const BnGeom& g, WgradTile* t, size_t* lds_bytes) {
    const int lgQ = ilog2_exact_wg(g.Ws), lgP = ilog2_exact_wg(g.Hs);
    if (lgQ < 0 || lgP < 0 || g.Ws < 4) return false;
    const int PQ = g.Hs * g.Ws;
    if (g.Ws > WG_TPX) return false;
    if (PQ >= WG_TPX) {
        t->F = 1;
        t->PT_H = WG_TPX / g.Ws;
    } else {
        t->F = WG_TPX / PQ;
        t->PT_H = g.Hs;
    }
    t->lgQ = lgQ;
    t->lgPTQ = ilog2_exact_wg(t->PT_H * g.Ws);
    t->tiles_per_frame = (t->F == 1) ? g.Hs / t->PT_H : 1;
    t->n_stages = ((g.N + t->F - 1) / t->F) * t->tiles_per_frame;
    t->IH = 2 * (t->PT_H - 1) + 5;
    const int IW = 2 * (g.Ws - 1) + 5;
    t->HALF = (IW + 1) / 2;
    t->RW = 2 * t->HALF;
    t->FSb = t->IH * t->RW;
    t->rows_per_b = t->F * t->IH;
    int bch = t->F * t->FSb;
    while ((bch & 31) != 2) ++bch;
    t->BCH = bch;
    t->big_words = WG_TB * t->BCH;
    if (t->big_words > WG_THREADS * WG_KB) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;   // 32-bit offsets
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    t->inv_rw = 1.0f / (float)t->RW;
    t->inv_bch = 1.0f / (float)t->BCH;
    t->inv_ih = 1.0f / (float)t->IH;
    // the DMA writes whole 64-word wave rows: round the image up so the last row stays inside
    int words = WG_TA * WG_SP + ((t->big_words + 63) & ~63);
    t->buf_floats = (words + 3) & ~3;
    *lds_bytes = (size_t)2 * t->buf_floats * 4;
    return *lds_bytes <= WG_MAX_LDS;
}

static int wgrad_splits(const BnGeom& g, const WgradTile& t) {
    const int tiles = ((g.Cs + WG_TA - 1) / WG_TA) * ((g.Cb + WG_TB - 1) / WG_TB);
    int splits = (256 + tiles - 1) / tiles;         // one workgroup per CU
    if (splits > t.n_stages) splits = t.n_stages;
    if (splits < 1) splits = 1;
    return splits;
}

template <int LGQ>
static int launch_wgrad(dim3 grid, size_t lds, hipStream_t st, const float* small, const float* big,
                        float* part, const BnGeom& g, const WgradTile& t) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_wgrad_mfma<LGQ>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, WG_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_wgrad_mfma<LGQ>, grid, dim3(WG_THREADS), lds, st, small, big, part, g, t);
    BN_LAUNCH_CHECK();
    return 0;
}

BnFastPlan bn_fast_wgrad_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || (g.stride != 2 && g.stride != 1)) return p;
    if (g.Cs < 16 || g.Cb < 16) return p;
    const BnFastPlan p4 = bn_wgrad4_plan(g);        // 16-byte DMA generation where it fits
    if (p4.supported) return p4;
    if (g.stride != 2) return p;
    WgradTile t;
    size_t lds = 0;
    if (!wgrad_tile(g, &t, &lds)) return p;
    p.supported = true;
    p.d = wgrad_splits(g, t);
    p.ws_bytes = (size_t)p.d * 25 * g.Cs * g.Cb * sizeof(float);
    p.kernel_name = "k_wgrad_mfma";
    return p;
}

int bn_launch_wgrad_fast(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                         const BnGeom& g, int accumulate, void* ws, hipStream_t st, float* db,
                         int bias_side, bool* bias_done) {
    if (plan.variant == 4 || plan.variant == 5 || plan.variant == 6)
        return bn_launch_wgrad4(plan, small, big, dw, g, accumulate, ws, st, db, bias_side,
                                bias_done);
    WgradTile t;
    size_t lds = 0;
    if (!wgrad_tile(g, &t, &lds)) return BN_E_SHAPE;
    t.splits = plan.d;
    static int dbg = -1;
    if (dbg < 0) { const char* e = bn_tune_env("BN_WGRAD_DBG"); dbg = e ? atoi(e) : 0; }
    t.dbg = dbg;
    const int tiles = ((g.Cs + WG_TA - 1) / WG_TA) * ((g.Cb + WG_TB - 1) / WG_TB);
    dim3 grid(tiles, t.splits);
    int rc = BN_E_SHAPE;
    switch (t.lgQ) {
        case 2: rc = launch_wgrad<2>(grid, lds, st, small, big, (float*)ws, g, t); break;
        case 3: rc = launch_wgrad<3>(grid, lds, st, small, big, (float*)ws, g, t); break;
        case 4: rc = launch_wgrad<4>(grid, lds, st, small, big, (float*)ws, g, t); break;
        case 5: rc = launch_wgrad<5>(grid, lds, st, small, big, (float*)ws, g, t); break;
        case 6: rc = launch_wgrad<6>(grid, lds, st, small, big, (float*)ws, g, t); break;
        default: break;
    }
    if (rc) return rc;
    return bn_launch_sum_partials((const float*)ws, dw, 25 * g.Cs * g.Cb, t.splits, accumulate,
                                  g.Cs * g.Cb, 25, st);
}
