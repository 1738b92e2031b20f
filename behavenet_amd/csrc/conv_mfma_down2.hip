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
// The weight slice of a chunk is copied by 16-byte LDS-DMA as well, in its global order (see the
// note at `woff`): the chunk boundary is then a handful of scalar-addressed DMA instructions.  It
// used to be ~250 vector instructions (32 loads, 32 transposing ds_write, selects), and a wave
// that is not in its MFMA loop gets about one instruction issued per MFMA of the wave it shares
// the SIMD with (s_memtime trace: boundary 2.6 k cycles alone, 15.8 k next to a multiplying wave
// -- longer than the 14.4 k cycle MFMA loop it was meant to hide behind).
//
// Round 4: the tile geometry (Down2Tile, computed on the host) takes maps that are no powers of two -- any
// even width (16-byte rows of the big map), any height: a tile is F whole frames or PT_H rows of one frame,
// lanes past the tile's pixels store nothing (down2_tile); KV = 4 skips the zero taps of a smaller kernel.
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"
#ifndef DOWN2_ST_AUX
#define DOWN2_ST_AUX 0      // cache policy of the dword output stores of k_down2_mfma
#endif


typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2d __attribute__((ext_vector_type(2)));

#define D2_STR2(x) #x
#define D2_STR(x) D2_STR2(x)
#ifndef D2_LOOP_SHIFT
#define D2_LOOP_SHIFT 0
#endif
#define D2_THREADS 256
#define D2_CC 4
#define D2_X0 4                 // LDS column of image column 0
#define D2_XK 8                 // max 16-byte DMA groups per thread per chunk
#define D2_MAX_LDS (80 * 1024)
#define D2_XBUF_FLOATS 6912         // fixed LDS distance between the two input images (27 KB: two
                                   // images + the 25.6 KB weight slice = 80 KB, two workgroups per CU)

static inline int ilog2_exact_d2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

// 16-byte LDS-DMA (a plain function: inside the kernel template the builtin's size argument would be
// checked at instantiation time, where the host pass rejects 16 and silently drops the kernel)
__device__ __forceinline__ void d2_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voffset,
                                         int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voffset, soffset, 0, 0);
}

struct Down2Tile {
    int F, PT_H, PTQ;             // frames / small-map rows of a workgroup tile, PT_H * Ws
    int IH, RW, FS, CHS;          // patch rows per frame, row stride, per-frame / per-channel floats
    int UPF;                      // units per frame: a tile is F UNITS of PT_H rows -- whole frames (UPF = 1),
                                  // the row blocks of one frame (F = 1), or (round 4) row blocks of adjacent
                                  // frames: a 12x12 map fills 3 x 72 = 216 of 256 pixels instead of 144
    int groups;                   // 16-byte groups of one chunk image (D2_CC * CHS / 4)
    int xbuf_floats;              // one LDS input image (whole wave rows)
    float inv_chs4, inv_fs4, inv_c4;
};

// KV = 4: the 5x5 taps are a smaller kernel zero-extended (BnGeom::KV): rows / columns of taps from KV on
// are neither read nor multiplied (a 4x4 layer: 16 of 25 products); LDS layouts stay the 5x5 ones.
// K0 = 1: so are row 0 / column 0 (BnGeom::K0, a 3x3 layer embedded at (1, 1): 9 of 25 products)
// ST = 1 (round 4): the stride-1 layers -- max-pooling architectures, and every 7x7 / 9x9 layer after its rewrite as a
// 5x5 layer on phases / shifted copies (capi.hip) -- on the same schedule: image rows `Ws + 8` words, a unit's image
// PT_H + 4 rows, a lane's five columns are consecutive words of either parity (4-byte reads paired by ds_read2_b32
// instead of 8-byte ones), any column offset 0..4.  KV = 5 only.
// POOL (round 6, stride 1 only): the 2x2 / stride-2 max pooling and the activation behind a max-pooling architecture's
// layer in the epilogue -- the tile's sums + bias go through LDS (rows of a window belong to different waves), a thread
// picks the winner of a window in row-major order (a later element wins only if strictly larger or NaN: torch's
// indices) and stores the activated maximum and its index h Ws + w; `dact_src` carries the index tensor.
template <int MR, int NR, int KV, int K0 = 0, int ST = 2, bool POOL = false>
__global__ __launch_bounds__(D2_THREADS, 2) void k_down2_mfma(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, Down2Tile t, int act,
    int dact, float slope, int cper, size_t zstride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // reduction split over workgroups (small batches: gridDim.z slices of cper input channels, raw
    // sums into slab blockIdx.z of the scratch, finished by k_split_epilogue); gridDim.z == 1: all
    const int c_beg = blockIdx.z * cper;
    const int c_end = min(g.Cb, c_beg + cper);
    out += blockIdx.z * zstride;
#ifdef D2_TRACE
    unsigned long long* trc = d2_trace + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * D2_TRACE_SLOTS;
#define D2_MARK(slot) do { if (threadIdx.x == 0) trc[slot] = __builtin_readcyclecounter(); } while (0)
    if (threadIdx.x == 0) {
        trc[0] = __builtin_readcyclecounter();
        trc[1] = __builtin_amdgcn_s_memrealtime();
        trc[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    }
#else
#define D2_MARK(slot)
#endif
    constexpr int CC = D2_CC, R = 5, S = 5, RS = 25;
    constexpr int RE = KV - K0, SE = KV - K0;         // rows / columns of taps that are multiplied
    constexpr int TM = 32 * MR;
    constexpr int WS = CC * RS;                       // weight row of one output channel (100 words)
    constexpr int WG = TM * WS / 4;                   // 16-byte groups of the weight slice
    constexpr int WK = (WG + D2_THREADS - 1) / D2_THREADS;
    float* wl = smem + 2 * D2_XBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;

    const int u0 = blockIdx.x * t.F;                 // first unit of this tile
    const int n0 = u0 / t.UPF;                       // its frame: the scalar part of the DMA offsets
    const int m0 = blockIdx.y * TM;
    const int Q = g.Ws, PQ = g.Hs * g.Ws;
    const int HW = g.Hb * g.Wb;

    int base[NR];
    size_t opix[NR];
    bool pvalid[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        // pixel -> (frame, row, column) of the tile; maps whose sizes are no powers of two leave the
        // last pixels of a tile (and the rows of a frame's last tile below the map) without an
        // output: those lanes multiply the tile's first pixel and store nothing
        int pix = 32 * (wv * NR + nr) + li;
        const bool inside = pix < t.F * t.PTQ;
        if (!inside) pix = 0;
        const int f = pix / t.PTQ;
        const int rem = pix - f * t.PTQ;
        const int pj = rem / Q, qj = rem - pj * Q;
        const int un = (u0 + f) / t.UPF, up0 = (u0 + f - un * t.UPF) * t.PT_H;     // the unit's frame, first row
        // pair 0 = columns (2q-2, 2q-1) of the image = LDS columns 2q+2, 2q+3
        // (stride 1: word 1 of the lane's run = image column q - pl = LDS column q - pl + D2_X0)
        base[nr] = ST == 2 ? f * t.FS + (2 * pj) * t.RW + 2 * qj + (D2_X0 - 2) + kk * t.CHS
                           : f * t.FS + pj * t.RW + qj + (D2_X0 - 1 - g.pl) + kk * t.CHS;
        pvalid[nr] = inside && un < g.N && (up0 + pj) < g.Hs;
        opix[nr] = (size_t)un * g.Cs * PQ + (size_t)(up0 + pj) * Q + qj;
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
        const int un = (u0 + f) / t.UPF, up0 = (u0 + f - un * t.UPF) * t.PT_H;
        const int hb = ST * up0 - g.pt + y, wb = 4 * c4 - D2_X0;
        const bool ok = e < t.groups && un < g.N && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
        xoff[k] = ok ? (((un - n0) * g.Cb + cc) * HW + hb * g.Wb + wb) * 4 : 0x7fffffff;
    }
    // weight slice of a chunk: rows m0.. of W[m][c0..c0+3][25 taps] = 100 contiguous words per
    // output channel, copied in that order (25 groups per row).  A lane later reads word
    // (2cp+kk)*25 + tap of row li: li * 100 words = li * 36 (mod 64) banks -> the 32 lanes of a
    // half wave fall on 16 bank quads, 2 lanes each -- the best any 16-byte-granular image of
    // per-lane rows can do, and 2 extra LDS cycles per read are nothing next to the MFMAs.
    // The two halves (kk) are 25 words apart: disjoint banks.
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * g.Cb * RS * 4), 0x00020000);
    int woff[WK];
#pragma unroll
    for (int k = 0; k < WK; ++k) {
        const int e = tid + D2_THREADS * k;
        const int m = e / (WS / 4), j = e - m * (WS / 4);
        woff[k] = (e < WG && m0 + m < g.Cs) ? ((m0 + m) * g.Cb * RS + 4 * j) * 4 : 0x7fffffff;
    }

    floatx16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mr][nr][e] = 0.f;

    // 16-byte LDS-DMA; the chunk-dependent part of the address is the scalar offset (unsigned,
    // added after the range check: padding / tail groups keep reading 0.0f)
    auto issue_image = [&](const int k, int c0, int buf) __attribute__((always_inline)) {
        if (D2_THREADS * k + 64 * wv < t.groups)                      // wave-uniform
            d2_dma16(rbig, smem + buf * D2_XBUF_FLOATS + 4 * (D2_THREADS * k + 64 * wv), xoff[k],
                     (n0 * g.Cb + c0) * HW * 4);
    };
    // The weight slice of the NEXT chunk travels through registers: 16-byte buffer loads issued from
    // inside this wave's MFMA stream, written to LDS at the chunk boundary.  A wave that is not
    // multiplying cannot issue vector-memory instructions while the wave it shares the SIMD with is
    // (tools/lab/coissue_probe.hip) -- with the slice fetched by LDS-DMA at the boundary, the two
    // workgroups of a CU alternated and every hand-over exposed the DMA latency (1.1 k cycles per
    // 13.2 k-cycle chunk); LDS writes and barriers are not held up, so a boundary made of those is
    // finished long before the other workgroup's loop is.
    typedef unsigned int d2_u4 __attribute__((ext_vector_type(4)));
    d2_u4 wreg[WK];
    auto load_weights = [&](const int k, int c0) __attribute__((always_inline)) {
        if (D2_THREADS * k + 64 * wv < WG)                            // wave-uniform
            wreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[k], c0 * RS * 4, 0);
    };
    auto store_weights = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            if (D2_THREADS * k + 64 * wv < WG)
                *reinterpret_cast<d2_u4*>(wl + 4 * (D2_THREADS * k + tid)) = wreg[k];
        }
    };

    // LDS addresses of the operand reads, computed ONCE: inside the MFMA loop every vector
    // instruction that is not an MFMA costs the matrix pipe 6-13 cycles (tools/lab/issue_probe.hip:
    // 10 v_add per 20 MFMAs = -8 %; LDS reads and scalar instructions are free), and the address
    // arithmetic of the reads was 10-13 such instructions per 20 MFMAs.  Now a row's reads are
    // `ds_read vaddr offset:imm` only: the image rows of (pixel block, channel pair, kernel row)
    // each have an address register, the second LDS image lies a compile-time distance behind the
    // first, and the weight reads share one lane base.
    constexpr int NIT = (CC / 2) * RE;
    // (ds_read2 offsets reach 255 words / double words only: one register per image and per
    // 32-channel weight block, so that no read needs an address add; the asm keeps the compiler
    // from re-deriving them from one another with an add in front of every read)
    int xro[2][NR][CC / 2][R];                       // float offsets from smem
#pragma unroll
    for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int cp = 0; cp < CC / 2; ++cp)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    xro[bf][nr][cp][r] = bf * D2_XBUF_FLOATS + base[nr] + (2 * cp) * t.CHS + r * t.RW;
                    asm volatile("" : "+v"(xro[bf][nr][cp][r]));
                }
    int wao[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        wao[mr] = 2 * D2_XBUF_FLOATS + (li + 32 * mr) * WS + kk * RS;
        asm volatile("" : "+v"(wao[mr]));
    }

    // One chunk = NIT rows of 5 taps x (MR x NR) MFMAs, everything unrolled, operand registers
    // double buffered.  The reads of row i+1 are cut into U units of one LDS instruction each and
    // placed by hand between the MFMAs of row i; sched_barrier pins that order (the scheduler's
    // own interleaving put dependent MFMAs back to back).
    constexpr int NM = SE * MR * NR;                 // MFMAs per row
    constexpr int WU = SE == 5 ? 3 : 2;              // weight read units per 32-channel block
    constexpr int NU = WU * MR + 2 * NR;             // read units per row
    static_assert(NU <= NM || K0 > 0, "one read unit per MFMA at most");
    auto load_unit = [&](const int BUF, const int it, const int u, float (&a)[S][MR],
                         float (&bq)[S][NR]) __attribute__((always_inline)) {
        const int cp = it / RE, r = K0 + it - cp * RE;
        if (u < WU * MR) {
            const int mr = u / WU, k = u - WU * mr;
            const float* wp = smem + wao[mr] + (2 * cp) * RS + r * S;
            if (K0 == 1) {
                if (k == 0) { a[1][mr] = wp[1]; a[2][mr] = wp[2]; }
                else a[3][mr] = wp[3];
            } else if (k == 0) { a[0][mr] = wp[0]; a[1][mr] = wp[1]; }
            else if (k == 1) { a[2][mr] = wp[2]; a[3][mr] = wp[3]; }
            else a[4][mr] = wp[4];
        } else {
            // columns 2q-1 .. 2q+3 = words 1..5 of the aligned six-word run
            const int v = u - WU * MR, nr = v / 2;
            const float* xb = smem + xro[BUF][nr][cp][r];
            if (ST == 1) {
                // either parity: 4-byte words (the compiler pairs them into ds_read2_b32)
                if ((v & 1) == 0) bq[0][nr] = xb[1];
                else { bq[1][nr] = xb[2]; bq[2][nr] = xb[3]; bq[3][nr] = xb[4]; bq[4][nr] = xb[5]; }
            } else if (K0 == 1) {
                if ((v & 1) == 0) {
                    const floatx2d c1p = *reinterpret_cast<const floatx2d*>(xb + 2);
                    bq[1][nr] = c1p.x; bq[2][nr] = c1p.y;
                } else {
                    bq[3][nr] = xb[4];
                }
            } else if ((v & 1) == 0) bq[0][nr] = xb[1];
            else {
                const floatx2d c1p = *reinterpret_cast<const floatx2d*>(xb + 2);
                bq[1][nr] = c1p.x; bq[2][nr] = c1p.y;
                if (SE == 5) {
                    const floatx2d c2p = *reinterpret_cast<const floatx2d*>(xb + 4);
                    bq[3][nr] = c2p.x; bq[4][nr] = c2p.y;
                } else {
                    bq[3][nr] = xb[4];
                }
            }
        }
    };
    auto chunk_rows = [&](const int BUF, const int c0) __attribute__((always_inline)) {
        float av[2][S][MR], bv[2][S][NR];
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(BUF, 0, u, av[0], bv[0]);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const int s = K0 + j / (MR * NR), mr = (j / NR) % MR, nr = j % NR;
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    av[it & 1][s][mr], bv[it & 1][s][nr], acc[mr][nr], 0, 0, 0);
                if (it + 1 < NIT) {
#pragma unroll
                    for (int u = (j * NU) / NM; u < ((j + 1) * NU) / NM; ++u)
                        load_unit(BUF, it + 1, u, av[(it + 1) & 1], bv[(it + 1) & 1]);
                }
                // the next chunk's loads ride in this stream, one instruction per slot
                const int sl = it * NM + j;
                if (sl >= 1 && sl < 1 + D2_XK) { if (c0 + CC < c_end) issue_image(sl - 1, c0 + CC, BUF ^ 1); }
                if (sl >= 1 + D2_XK && sl < 1 + D2_XK + WK) { if (c0 + CC < c_end) load_weights(sl - 1 - D2_XK, c0 + CC); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(1 + D2_XK + WK <= NIT * NM, "load slots");
    auto boundary = [&](int c0) __attribute__((always_inline)) {
        if (c0 == 2 * CC) D2_MARK(22);
        __syncthreads();   // the previous chunk's MFMA reads of wl (and of the other image) are done
        if (c0 == 2 * CC) D2_MARK(23);
        store_weights();
        if (c0 == 2 * CC) D2_MARK(24);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own image groups landed
        if (c0 == 2 * CC) D2_MARK(25);
        __syncthreads();
        if (c0 == 2 * CC) D2_MARK(26);
        if (c0 == 2 * CC) D2_MARK(27);
    };

    D2_MARK(3);
#pragma unroll
    for (int k = 0; k < D2_XK; ++k) issue_image(k, c_beg, 0);
#pragma unroll
    for (int k = 0; k < WK; ++k) load_weights(k, c_beg);
#ifdef D2_LOOP_ALIGN
    asm volatile(".p2align " D2_STR(D2_LOOP_ALIGN) "\n .rept " D2_STR(D2_LOOP_SHIFT) "\n s_nop 0\n .endr" ::: "memory");
#endif
    for (int c0 = c_beg; c0 < c_end; c0 += 2 * CC) {
        if (c0 < 6 * CC) D2_MARK(4 + 2 * (c0 / CC));
        boundary(c0);
        if (c0 < 6 * CC) D2_MARK(5 + 2 * (c0 / CC));
        chunk_rows(0, c0);
        if (c0 + CC < c_end) {
            if (c0 < 4 * CC) D2_MARK(6 + 2 * (c0 / CC));
            boundary(c0 + CC);
            if (c0 < 4 * CC) D2_MARK(7 + 2 * (c0 / CC));
            chunk_rows(1, c0 + CC);
        }
    }

    // ---- epilogue: lane holds channel (e&3)+8*(e>>2)+4*kk of pixel li for each register e.
    // Loads are batched ahead of the stores (bias: all at once; activation-derivative source: 16 per
    // accumulator block): with a load -> wait -> store chain per element the 64 elements of a lane
    // cost 64 memory round trips (59 k of the kernel's 490 k cycles, s_memtime trace).
    D2_MARK(16);
    if constexpr (POOL) {
        constexpr int TP = 128 * NR;
        __syncthreads();                                   // every wave is done reading the images
        float* ps = smem;                                  // [TM][TP] sums + bias
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int chn = mr * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                const float bzv = (bias && m0 + chn < g.Cs) ? bias[m0 + chn] : 0.f;
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) ps[chn * TP + 32 * (wv * NR + nr) + li] = acc[mr][nr][e] + bzv;
            }
        __syncthreads();
        int* pidx = reinterpret_cast<int*>(const_cast<float*>(dact_src));
        const int Wo = Q >> 1, Ho = g.Hs >> 1, hp = t.PT_H >> 1;
        const int per_unit = hp * Wo, per_tile = t.F * per_unit;      // <= TP / 4 windows of a channel
        // thread -> one window position (decoded once), TM * 4 / TP ... channels c0, c0 + 256 / (TP / 4), ...
        constexpr int WPT = TP / 4, CSTEP = D2_THREADS / WPT;
        const int pp = tid % WPT, c0 = tid / WPT;
        if (pp < per_tile) {
            const int f = pp / per_unit, r2 = pp - f * per_unit;
            const int pr = r2 / Wo, pc = r2 - pr * Wo;
            const int un = (u0 + f) / t.UPF, up0 = (u0 + f - un * t.UPF) * t.PT_H;
            const int h = up0 + 2 * pr;
            if (un < g.N && h < g.Hs) {
                const int me = h * Q + 2 * pc;
                const float* pv0 = ps + f * t.PTQ + 2 * pr * Q + 2 * pc;
                const size_t o0 = (((size_t)un * g.Cs + m0) * Ho + (h >> 1)) * Wo + pc;
                const size_t ostep = (size_t)Ho * Wo;
#pragma unroll 4
                for (int chn = c0; chn < TM; chn += CSTEP) {
                    if (m0 + chn >= g.Cs) break;
                    const float* pv = pv0 + chn * TP;
                    float best = -INFINITY;
                    int bi = me;
                    const float v00 = pv[0], v01 = pv[1], v10 = pv[Q], v11 = pv[Q + 1];
                    if (v00 > best || isnan(v00)) { best = v00; bi = me; }
                    if (v01 > best || isnan(v01)) { best = v01; bi = me + 1; }
                    if (v10 > best || isnan(v10)) { best = v10; bi = me + Q; }
                    if (v11 > best || isnan(v11)) { best = v11; bi = me + Q + 1; }
                    const size_t oi = o0 + chn * ostep;
                    out[oi] = bn_apply_act(best, act, slope);
                    pidx[oi] = bi;
                }
            }
        }
        return;
    }
    if (act != BN_ACT_SIGMOID && dact != BN_ACT_SIGMOID) {             // wave-uniform
        // branch-free: buffer loads / stores whose lane offset is out of range for lanes (or
        // channels) that do not exist -- per-element branches made the compiler drain the memory
        // queue (s_waitcnt vmcnt(0)) in front of every single store
        const float es = (act == BN_ACT_LRELU) ? slope : 1.f;           // identity = slope 1
        const float ds = (dact == BN_ACT_LRELU) ? slope : 1.f;
        const int obytes = (int)((size_t)g.N * g.Cs * PQ * 4);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, obytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)dact_src, 0, dact_src ? obytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)bias, 0, bias ? g.Cs * 4 : 0, 0x00020000);
        const int mlane = m0 + 4 * kk;
        float bz[MR][16];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                bz[mr][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rbs, mlane * 4 + (mr * 32 + (e & 3) + 8 * (e >> 2)) * 4, 0, 0));
#ifdef D2_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        D2_MARK(17);
#endif
        int vo[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
            vo[nr] = pvalid[nr] ? (int)((opix[nr] + (size_t)mlane * PQ) * 4) : 0x7fffffff;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                float d[16];
                if (dact_src) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int mo = mr * 32 + (e & 3) + 8 * (e >> 2);
                        d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rd, (mlane + mo < g.Cs) ? vo[nr] : 0x7fffffff, mo * PQ * 4, 0));
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int mo = mr * 32 + (e & 3) + 8 * (e >> 2);
                    float v = acc[mr][nr][e] + bz[mr][e];
                    v = v > 0.f ? v : v * es;
                    if (dact_src) v *= d[e] > 0.f ? 1.f : ds;
                    __builtin_amdgcn_raw_buffer_store_b32(
                        __builtin_bit_cast(int, v), ro, (mlane + mo < g.Cs) ? vo[nr] : 0x7fffffff,
                        mo * PQ * 4, DOWN2_ST_AUX);
                }
                D2_MARK(18 + mr * NR + nr);
            }
        }
    } else {
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
#ifdef D2_TRACE
    if (threadIdx.x == 0) {
        trc[14] = __builtin_readcyclecounter();
        trc[15] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---- 16-row tile (round 6): layers whose small side has 16 channels (the max-pooling test architecture's 32 -> 16
// layers, a drawn architecture's first block).  k_down2_mfma's 32-row MFMA multiplied 16 rows of zeros for them; here
// the products run on v_mfma_f32_16x16x4_f32: A = 16 channels x the chunk's 4 input channels of one tap, B = those 4
// channels x 16 pixels, so a chunk is 5 kernel rows x 5 taps x NR pixel blocks of ONE instruction each (32 cycles, the
// same rate as the 32x32x2 one).  Tile = 16 channels x 64 NR pixels on Down2Tile's geometry (NR = 4: the 256-pixel
// tile), images / weights / boundary / issue order exactly as above.  Lane = (li = pixel or channel 0..15, kk = input
// channel 0..3 of the chunk); the weight rows' li * 100 + kk * 25 words fall on 64 different banks.
typedef float floatx4d __attribute__((ext_vector_type(4)));

template <int NR, int ST>
__global__ __launch_bounds__(D2_THREADS, 2) void k_down2_m16(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, Down2Tile t, int act,
    int dact, float slope, int cper, size_t zstride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int c_beg = blockIdx.z * cper;
    const int c_end = min(g.Cb, c_beg + cper);
    out += blockIdx.z * zstride;
    constexpr int CC = D2_CC, R = 5, S = 5, RS = 25;
    constexpr int TM = 16;
    constexpr int WS = CC * RS;
    constexpr int WG = TM * WS / 4;                   // 400 16-byte groups
    constexpr int WK = (WG + D2_THREADS - 1) / D2_THREADS;
    float* wl = smem + 2 * D2_XBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;

    const int u0 = blockIdx.x * t.F;
    const int n0 = u0 / t.UPF;
    const int m0 = blockIdx.y * TM;
    const int Q = g.Ws, PQ = g.Hs * g.Ws;
    const int HW = g.Hb * g.Wb;

    int base[NR];
    size_t opix[NR];
    bool pvalid[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        int pix = 16 * (wv * NR + nr) + li;
        const bool inside = pix < t.F * t.PTQ;
        if (!inside) pix = 0;
        const int f = pix / t.PTQ;
        const int rem = pix - f * t.PTQ;
        const int pj = rem / Q, qj = rem - pj * Q;
        const int un = (u0 + f) / t.UPF, up0 = (u0 + f - un * t.UPF) * t.PT_H;
        base[nr] = ST == 2 ? f * t.FS + (2 * pj) * t.RW + 2 * qj + (D2_X0 - 2) + kk * t.CHS
                           : f * t.FS + pj * t.RW + qj + (D2_X0 - 1 - g.pl) + kk * t.CHS;
        pvalid[nr] = inside && un < g.N && (up0 + pj) < g.Hs;
        opix[nr] = (size_t)un * g.Cs * PQ + (size_t)(up0 + pj) * Q + qj;
    }

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
        const int un = (u0 + f) / t.UPF, up0 = (u0 + f - un * t.UPF) * t.PT_H;
        const int hb = ST * up0 - g.pt + y, wb = 4 * c4 - D2_X0;
        const bool ok = e < t.groups && un < g.N && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
        xoff[k] = ok ? (((un - n0) * g.Cb + cc) * HW + hb * g.Wb + wb) * 4 : 0x7fffffff;
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * g.Cb * RS * 4), 0x00020000);
    int woff[WK];
#pragma unroll
    for (int k = 0; k < WK; ++k) {
        const int e = tid + D2_THREADS * k;
        const int m = e / (WS / 4), j = e - m * (WS / 4);
        woff[k] = (e < WG && m0 + m < g.Cs) ? ((m0 + m) * g.Cb * RS + 4 * j) * 4 : 0x7fffffff;
    }

    floatx4d acc[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nr][e] = 0.f;

    auto issue_image = [&](const int k, int c0, int buf) __attribute__((always_inline)) {
        if (D2_THREADS * k + 64 * wv < t.groups)                      // wave-uniform
            d2_dma16(rbig, smem + buf * D2_XBUF_FLOATS + 4 * (D2_THREADS * k + 64 * wv), xoff[k],
                     (n0 * g.Cb + c0) * HW * 4);
    };
    typedef unsigned int d2_u4 __attribute__((ext_vector_type(4)));
    d2_u4 wreg[WK];
    auto load_weights = [&](const int k, int c0) __attribute__((always_inline)) {
        if (D2_THREADS * k + 64 * wv < WG)                            // wave-uniform
            wreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[k], c0 * RS * 4, 0);
    };
    auto store_weights = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < WK; ++k) {
            if (D2_THREADS * k + 64 * wv < WG)
                *reinterpret_cast<d2_u4*>(wl + 4 * (D2_THREADS * k + tid)) = wreg[k];
        }
    };

    int xro[2][NR][R];
#pragma unroll
    for (int bf = 0; bf < 2; ++bf)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xro[bf][nr][r] = bf * D2_XBUF_FLOATS + base[nr] + r * t.RW;
                asm volatile("" : "+v"(xro[bf][nr][r]));
            }
    int wao = 2 * D2_XBUF_FLOATS + li * WS + kk * RS;
    asm volatile("" : "+v"(wao));

    constexpr int NIT = R;                            // one MFMA takes the chunk's four channels
    constexpr int NM = S * NR;                        // MFMAs per kernel row
    constexpr int WU = 3;
    constexpr int NU = WU + 2 * NR;
    static_assert(NU <= NM, "one read unit per MFMA at most");
    auto load_unit = [&](const int BUF, const int r, const int u, float (&a)[S],
                         float (&bq)[S][NR]) __attribute__((always_inline)) {
        if (u < WU) {
            const float* wp = smem + wao + r * S;
            if (u == 0) { a[0] = wp[0]; a[1] = wp[1]; }
            else if (u == 1) { a[2] = wp[2]; a[3] = wp[3]; }
            else a[4] = wp[4];
        } else {
            const int v = u - WU, nr = v / 2;
            const float* xb = smem + xro[BUF][nr][r];
            if (ST == 1) {
                if ((v & 1) == 0) bq[0][nr] = xb[1];
                else { bq[1][nr] = xb[2]; bq[2][nr] = xb[3]; bq[3][nr] = xb[4]; bq[4][nr] = xb[5]; }
            } else if ((v & 1) == 0) bq[0][nr] = xb[1];
            else {
                const floatx2d c1p = *reinterpret_cast<const floatx2d*>(xb + 2);
                const floatx2d c2p = *reinterpret_cast<const floatx2d*>(xb + 4);
                bq[1][nr] = c1p.x; bq[2][nr] = c1p.y;
                bq[3][nr] = c2p.x; bq[4][nr] = c2p.y;
            }
        }
    };
    auto chunk_rows = [&](const int BUF, const int c0) __attribute__((always_inline)) {
        float av[2][S], bv[2][S][NR];
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(BUF, 0, u, av[0], bv[0]);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const int s = j / NR, nr = j % NR;
                acc[nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[it & 1][s], bv[it & 1][s][nr], acc[nr], 0, 0, 0);
                if (it + 1 < NIT) {
#pragma unroll
                    for (int u = (j * NU) / NM; u < ((j + 1) * NU) / NM; ++u)
                        load_unit(BUF, it + 1, u, av[(it + 1) & 1], bv[(it + 1) & 1]);
                }
                const int sl = it * NM + j;
                if (sl >= 1 && sl < 1 + D2_XK) { if (c0 + CC < c_end) issue_image(sl - 1, c0 + CC, BUF ^ 1); }
                if (sl >= 1 + D2_XK && sl < 1 + D2_XK + WK) { if (c0 + CC < c_end) load_weights(sl - 1 - D2_XK, c0 + CC); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    static_assert(1 + D2_XK + WK <= NIT * NM, "load slots");
    auto boundary = [&]() __attribute__((always_inline)) {
        __syncthreads();
        store_weights();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };

#pragma unroll
    for (int k = 0; k < D2_XK; ++k) issue_image(k, c_beg, 0);
#pragma unroll
    for (int k = 0; k < WK; ++k) load_weights(k, c_beg);
    for (int c0 = c_beg; c0 < c_end; c0 += 2 * CC) {
        boundary();
        chunk_rows(0, c0);
        if (c0 + CC < c_end) {
            boundary();
            chunk_rows(1, c0 + CC);
        }
    }

    // ---- epilogue: register e of block nr = channel m0 + 4 kk + e of pixel li
    const int mlane = m0 + 4 * kk;
    if (act != BN_ACT_SIGMOID && dact != BN_ACT_SIGMOID) {             // wave-uniform
        const float es = (act == BN_ACT_LRELU) ? slope : 1.f;
        const float ds = (dact == BN_ACT_LRELU) ? slope : 1.f;
        const int obytes = (int)((size_t)g.N * g.Cs * PQ * 4);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, obytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)dact_src, 0, dact_src ? obytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rbs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)bias, 0, bias ? g.Cs * 4 : 0, 0x00020000);
        float bz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bz[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbs, (mlane + e) * 4, 0, 0));
        int vo[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
            vo[nr] = pvalid[nr] ? (int)((opix[nr] + (size_t)mlane * PQ) * 4) : 0x7fffffff;
        float d[NR][4];
        if (dact_src) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    d[nr][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rd, (mlane + e < g.Cs) ? vo[nr] : 0x7fffffff, e * PQ * 4, 0));
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[nr][e] + bz[e];
                v = v > 0.f ? v : v * es;
                if (dact_src) v *= d[nr][e] > 0.f ? 1.f : ds;
                __builtin_amdgcn_raw_buffer_store_b32(
                    __builtin_bit_cast(int, v), ro, (mlane + e < g.Cs) ? vo[nr] : 0x7fffffff, e * PQ * 4, DOWN2_ST_AUX);
            }
    } else {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            if (!pvalid[nr]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = mlane + e;
                if (m >= g.Cs) continue;
                const size_t idx = opix[nr] + (size_t)m * PQ;
                float v = acc[nr][e];
                if (bias) v += bias[m];
                v = bn_apply_act(v, act, slope);
                if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
                out[idx] = v;
            }
        }
    }
}

static bool down2_tile(const BnGeom& g, int MR, int NR, Down2Tile* t, size_t* lds_bytes) {
    const int ST = g.stride;
    if (g.R != 5 || g.S != 5 || (ST != 2 && ST != 1) || g.pt < 0) return false;
    if (ST == 2 ? g.pl != 1 : (g.pl < 0 || g.pl > 4 || g.KV == 4)) return false;
    if ((g.Cb % D2_CC) != 0 || (g.Wb & 3) != 0) return false;
    const int TP = 128 * NR;
    // any even map width (16-byte rows of the big map), any height: a tile is F whole frames or PT_H
    // rows of one (the rows of a frame spread evenly over its tiles); powers of two fill it exactly
    if (g.Ws < 4 || (ST == 2 && (g.Ws & 1)) || g.Ws > TP || g.Hs < 1) return false;
    const bool pow2 = ilog2_exact_d2(g.Ws) >= 0 && ilog2_exact_d2(g.Hs) >= 0;
    const int rw0 = ST == 2 ? 2 * g.Ws + 8 : (g.Ws + 8 + 3) & ~3;
    int rw = rw0;
    // the 32 pixels of a half wave (several image rows when Ws < 32 or no power of two) cover the 64
    // banks once with their 8-byte reads if the row stride == Ws (mod 32)
    if ((g.Ws < 32 || !pow2) && (g.Ws & 3) == 0)
        while ((rw & 31) != (g.Ws & 31)) rw += 4;
    // units of PT_H rows, F of them per tile: among the splits of a frame into UPF row blocks the one that
    // fills the tile best (ties: fewer blocks = less halo); an LDS image that does not fit first drops the
    // bank-friendly row stride (conflicts on the operand reads cost less than idle pixels), then units
    auto fits = [&](int pth, int stride, int F) {
        t->RW = stride; t->PT_H = pth; t->F = F;
        t->IH = ST * (pth - 1) + 5;
        t->FS = t->IH * stride;
        t->CHS = F * t->FS;
        t->groups = D2_CC * t->CHS / 4;
        t->xbuf_floats = 4 * ((t->groups + 63) & ~63);
        return t->groups <= D2_THREADS * D2_XK && t->xbuf_floats <= D2_XBUF_FLOATS;
    };
    float best = 0.f;
    int b_upf = 0, b_pth = 0, b_F = 0, b_rw = 0;
    for (int upf = 1; upf <= g.Hs; ++upf) {
        const int pth = (g.Hs + upf - 1) / upf;
        if ((g.Hs + pth - 1) / pth != upf || pth * g.Ws > TP) continue;
        // (maps that are powers of two keep the tiles they were measured with: whole frames or row blocks)
        int F = TP / (pth * g.Ws);
        if (pow2 && upf > 1) F = 1;
        for (int pass = 0; pass < 2; ++pass) {
            const int stride = pass == 0 ? rw : rw0;
            int Fp = F;
            while (Fp >= 1 && !fits(pth, stride, Fp)) { if (pass == 0) { Fp = 0; break; } --Fp; }
            if (Fp < 1) continue;
            const float fill = (float)(Fp * pth * g.Ws) / (float)TP * (float)g.Hs / (float)(upf * pth);
            if (fill > best + 1e-6f) { best = fill; b_upf = upf; b_pth = pth; b_F = Fp; b_rw = stride; }
            break;
        }
        if (best >= 0.999f) break;
    }
    if (b_upf == 0) return false;
    fits(b_pth, b_rw, b_F);
    t->UPF = b_upf;
    t->PTQ = t->PT_H * g.Ws;
    t->inv_chs4 = 1.0f / (float)(t->CHS / 4);
    t->inv_fs4 = 1.0f / (float)(t->FS / 4);
    t->inv_c4 = 1.0f / (float)(t->RW / 4);
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    *lds_bytes = ((size_t)2 * D2_XBUF_FLOATS + (size_t)D2_CC * 25 * 32 * MR + 256) * 4;
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

template <int MR, int NR, int KV, int K0 = 0, int ST = 2>
static int launch_down2(const Down2Tile& t, dim3 grid, size_t lds, const float* big, const float* w,
                        const float* bias, float* out, const float* dact_src, const BnGeom& g,
                        int act, int dact, float slope, hipStream_t st, int cper, size_t zstride) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_down2_mfma<MR, NR, KV, K0, ST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, D2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    BN_LAUNCH_MAIN((k_down2_mfma<MR, NR, KV, K0, ST>), grid, dim3(D2_THREADS), lds, st, big, w, bias, out,
                       dact_src, g, t, act, dact, slope, cper, zstride);
    BN_LAUNCH_CHECK();
    return 0;
}

template <int NR, int ST>
static int launch_down2_m16(const Down2Tile& t, dim3 grid, size_t lds, const float* big, const float* w,
                            const float* bias, float* out, const float* dact_src, const BnGeom& g,
                            int act, int dact, float slope, hipStream_t st, int cper, size_t zstride) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_down2_m16<NR, ST>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, D2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    BN_LAUNCH_MAIN((k_down2_m16<NR, ST>), grid, dim3(D2_THREADS), lds, st, big, w, bias, out,
                       dact_src, g, t, act, dact, slope, cper, zstride);
    BN_LAUNCH_CHECK();
    return 0;
}

// the 16-row tile serves 5x5 layers (all 25 taps) of either stride; MR = 0 names it in the plans
bool bn_down2_m16_supported(const BnGeom& g, int NR) {
    if (g.K0 != 0) return false;        // (zero-extended 4x4 taps are multiplied as they are)
    return bn_down2_supported(g, 1, NR);
}

// Conv2d + 2x2 pooling + activation (POOL instantiation): stride-1 5x5 layers with 32 k output channels on even maps
// whose tiles are whole even row blocks (32 channels x 256 pixels)
bool bn_down2_pool_ok(const BnGeom& g) {
    Down2Tile t;
    size_t lds = 0;
    if (g.stride != 1 || g.R != 5 || g.S != 5 || g.KV != 0 || g.K0 != 0 || (g.Cs & 31) || (g.Hs & 1) || (g.Ws & 1)) return false;
    if (!bn_down2_supported(g, 1, 2) || !down2_tile(g, 1, 2, &t, &lds)) return false;
    if ((t.PT_H & 1) || bn_down2_splits(g, 1, 2) != 1) return false;
    return (size_t)32 * 256 * 4 <= lds && (size_t)g.Hs * g.Ws < 0x7fffffffull;
}
int bn_launch_down2_pool(const float* big, const float* w, const float* bias, float* y, int* idx, const BnGeom& g,
                         int act, float slope, hipStream_t st) {
    Down2Tile t;
    size_t lds = 0;
    if (!bn_down2_pool_ok(g) || !down2_tile(g, 1, 2, &t, &lds)) return BN_E_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_down2_mfma<1, 2, 5, 0, 1, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, D2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid((g.N * t.UPF + t.F - 1) / t.F, (g.Cs + 31) / 32, 1);
    BN_LAUNCH_MAIN((k_down2_mfma<1, 2, 5, 0, 1, true>), grid, dim3(D2_THREADS), lds, st, big, w, bias, y,
                       reinterpret_cast<const float*>(idx), g, t, act, BN_ACT_NONE, slope, g.Cb, (size_t)0);
    BN_LAUNCH_CHECK();
    return 0;
}

// share of a tile's pixels that are pixels of the map (1 for powers of two), 0 if not served
float bn_down2_fill(const BnGeom& g, int MR, int NR) {
    Down2Tile t;
    size_t lds = 0;
    if (!bn_down2_supported(g, MR, NR) || !down2_tile(g, MR, NR, &t, &lds)) return 0.f;
    return (float)(t.F * t.PT_H * g.Ws) / (float)(128 * NR) * (float)g.Hs / (float)(t.UPF * t.PT_H);
}

// reduction splits: a grid that fills less than a quarter of the chip's 512 workgroup slots (a
// 32-frame shard of a trial) is cut over the input channels into up to 8 slices of >= 16 channels
int bn_down2_splits(const BnGeom& g, int MR, int NR) {
    Down2Tile t;
    size_t lds = 0;
    if (!down2_tile(g, MR, NR, &t, &lds)) return 1;
    const int wgs = ((g.N * t.UPF + t.F - 1) / t.F) * ((g.Cs + 32 * MR - 1) / (32 * MR));
    if (wgs > 160) return 1;
    int s = 512 / wgs;
    if (s > 8) s = 8;
    while (s > 1 && (g.Cb % (s * 2 * D2_CC) != 0 || g.Cb / s < 16)) --s;
    return s < 1 ? 1 : s;
}

int bn_launch_down2(int MR, int NR, const float* big, const float* w, const float* bias,
                    float* out, const float* dact_src, const BnGeom& g, int act, int dact,
                    float slope, hipStream_t st, int splits, void* ws) {
    Down2Tile t;
    size_t lds = 0;
    const bool m16 = MR == 0;
    if (!down2_tile(g, m16 ? 1 : MR, NR, &t, &lds)) return BN_E_SHAPE;
    const int tiles = (g.N * t.UPF + t.F - 1) / t.F;
    if (splits < 1) splits = 1;
    dim3 grid(tiles, m16 ? (g.Cs + 15) / 16 : (g.Cs + 32 * MR - 1) / (32 * MR), splits);
    const size_t total = (size_t)g.N * g.Cs * g.Hs * g.Ws;
    if (splits > 1 && !ws) return BN_E_WORKSPACE;
    // (split: raw sums into the slabs, no bias / activation / mask in the kernel)
    const float* kb = splits > 1 ? nullptr : bias;
    const float* kd = splits > 1 ? nullptr : dact_src;
    float* ko = splits > 1 ? (float*)ws : out;
    const int ka = splits > 1 ? BN_ACT_NONE : act, kda = splits > 1 ? BN_ACT_NONE : dact;
    const int cper = g.Cb / splits;
    const size_t zs = splits > 1 ? total : 0;
    int rc = BN_E_SHAPE;
    const bool k4 = g.KV == 4, k3 = k4 && g.K0 == 1;
#define D2_CASE(mr, nr)                                                                                       \
    if (MR == mr && NR == nr)                                                                                 \
        rc = g.stride == 1 ? launch_down2<mr, nr, 5, 0, 1>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs) \
           : k3 ? launch_down2<mr, nr, 4, 1>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs) \
           : k4 ? launch_down2<mr, nr, 4>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs)   \
                : launch_down2<mr, nr, 5>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs);
    D2_CASE(2, 2) D2_CASE(2, 1) D2_CASE(1, 1) D2_CASE(1, 2)
#undef D2_CASE
    if (m16) {
        if (!bn_down2_m16_supported(g, NR)) return BN_E_SHAPE;
        if (NR == 2)
            rc = g.stride == 1 ? launch_down2_m16<4, 1>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs)
                               : launch_down2_m16<4, 2>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs);
        else if (NR == 1)
            rc = g.stride == 1 ? launch_down2_m16<2, 1>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs)
                               : launch_down2_m16<2, 2>(t, grid, lds, big, w, kb, ko, kd, g, ka, kda, slope, st, cper, zs);
    }
    if (rc || splits == 1) return rc;
    return bn_launch_split_epilogue((const float*)ws, bias, out, dact_src, total, splits, g.Cs,
                                    g.Hs * g.Ws, act, dact, slope, st);
}
