// HBM-bound edge layers of the autoencoder (single-channel image side), kernel 5x5 stride 2:
//
//   enc.conv0  fwd        1 -> 32   read 64 KB, write 512 KB per frame   (k_down_c1)
//   dec.convT4 bwd-data   same kernel with the LeakyReLU' mask epilogue
//   dec.convT4 fwd        32 -> 1   read 512 KB, write 64 KB per frame   (k_up_c1)
//   enc.conv0 / dec.convT4 weight gradients (32 x 25 outputs, 580 KB read per frame)
//                                                                       (k_wgrad_c1)
//
// All three are bound by the 8 TB/s HBM stream, not by arithmetic (11 FLOP/B); the matrix cores
// are used as the FMA engine (v_mfma_f32_32x32x2_f32, exact fp32) only because that keeps the
// VALU and LDS out of the way of the stream.  Every global access is a full 128/256-byte
// wavefront row; padding, halos and tails come from raw-buffer out-of-range zeros.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"
#include "bn_reduce.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define ED_THREADS 256
#define ED_OOB 0x7fffffff

__device__ __forceinline__ float ed_ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// =============================================================================================
// weight gradient with one big-side channel:
//   dW[a][tap] = sum_{n,p,q} small[n,a,p,q] * big[n,0,2p+r-pt,2q+s-pl],   a < 32
// MFMA roles: rows = a (32), columns = taps (25 of 32), reduction = pixels (2 per instruction).
// Stage = 4 rows x 64 columns of the small image (256 pixels, one row per wave).
// =============================================================================================
#define WC_ROWS 4
#define WC_W 64
#define WC_TPX (WC_ROWS * WC_W)
#define WC_SP (WC_TPX + 2)              // small-tile row stride (== 2 mod 32)
#define WC_IH (2 * WC_ROWS + 3)
#define WC_RW (2 * WC_W + 4)            // big-tile row stride (131 used)
#define WC_KS (32 * WC_TPX / ED_THREADS)            // 32 small elements per thread per stage

// Grid of the two weight-gradient kernels below: G workgroups per big-side channel.  `pair` = 0: grid (G, Cb).
// `pair` = 1 (round 5, two-channel frames -- PS-VAE, BASELINE configs[3]): a 1-D grid of G * Cb workgroups decoded
// so that the Cb workgroups walking the SAME stages sit on ONE XCD, 8 blocks apart in dispatch order (block b runs on
// XCD b % 8): they read the same 134 MB small-side tensor, which used to come from HBM once per channel -- the second
// reader now finds the stage in that XCD's L2.
#define WC_DECODE_GRID()                                                                    \
    int bch, bx, G;                                                                         \
    if (pair) {                                                                             \
        const int id = blockIdx.x, r = id >> 3;                                             \
        G = gridDim.x / g.Cb;                                                               \
        bch = r % g.Cb;                                                                     \
        bx = (r / g.Cb) * 8 + (id & 7);                                                     \
    } else {                                                                                \
        bch = blockIdx.y; bx = blockIdx.x; G = gridDim.x;                                   \
    }

// ST = 1 (round 4): stride-1 layers with a single-channel side (the first / last layer of a max-pooling
// architecture: im2col + a GEMM over 4 M rows before).  A frame's stages are blocks of 4 rows x 64
// columns of the small map (any size: the blocks at the right and lower edge are masked), so the
// same kernel serves maps wider than 64 columns.
// VEC: the small tile by 16-byte loads (rows of the small map are 16-byte multiples and the tensor is aligned): four
// pixels of one channel per load instead of one -- 8 load instructions per thread and stage instead of 32 (the
// stride-1 launch of the max-pooling architecture: 122 us for 285 MB with the dword loads)
template <int ST, bool VEC = false>
__global__ __launch_bounds__(ED_THREADS) void k_wgrad_c1(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    float* __restrict__ bias_part, BnGeom g, int n_stages, int stages_per_frame, int cblocks, int pair) {
    constexpr int IH = ST * (WC_ROWS - 1) + 5, RW = ST * WC_W + 4;
    constexpr int KB = (IH * RW + ED_THREADS - 1) / ED_THREADS;
    __shared__ float sl[32 * WC_SP];
    __shared__ float bl[IH * RW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * g.Cb * HWb * 4), 0x00020000);
    // big-side channel (Cb <= 4: 2-channel frames) and this workgroup's slot among the G of its channel
    WC_DECODE_GRID();

    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // fused bias gradient of a Conv2d (sum of `small` per channel): every small element is the A
    // operand of exactly one lane and step, so it is summed on its way through the registers
    float bsum = 0.f;

    float sr[WC_KS];
    float br[KB];

    auto issue_loads = [&](int st) {
        const int n = st / stages_per_frame;
        const int rem = st - n * stages_per_frame;
        const int rblk = rem / cblocks;
        const int p0 = rblk * WC_ROWS, q00 = (rem - rblk * cblocks) * WC_W;
        if (VEC) {
#pragma unroll
            for (int k = 0; k < WC_KS / 4; ++k) {
                const int e = tid + ED_THREADS * k;               // 16-byte group: channel a, pixels 4 g4 ..
                const int a = e >> 6, g4 = e & 63;
                const int row = p0 + (g4 >> 4), col = q00 + 4 * (g4 & 15);
                const bool ok = a < g.Cs && row < g.Hs && col < g.Ws;
                typedef float wc_f4 __attribute__((ext_vector_type(4)));
                const wc_f4 v = __builtin_bit_cast(wc_f4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs, ok ? (((n * g.Cs + a) * g.Hs + row) * g.Ws + col) * 4 : ED_OOB, 0, 0));
                sr[4 * k] = v.x; sr[4 * k + 1] = v.y; sr[4 * k + 2] = v.z; sr[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
        for (int k = 0; k < WC_KS; ++k) {
            const int e = tid + ED_THREADS * k;
            const int a = e >> 8, pix = e & (WC_TPX - 1);
            const int row = p0 + (pix >> 6), col = q00 + (pix & (WC_W - 1));
            const bool ok = a < g.Cs && row < g.Hs && col < g.Ws;
            sr[k] = ed_ld(rs, ok ? (((n * g.Cs + a) * g.Hs + row) * g.Ws + col) * 4 : ED_OOB);
        }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = tid + ED_THREADS * k;
            const int y = e / RW, x = e - y * RW;
            const int hb = ST * p0 - g.pt + y, wb = ST * q00 + x - g.pl;
            const bool ok = y < IH && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            br[k] = ed_ld(rb, ok ? (((n * g.Cb + bch) * g.Hb + hb) * g.Wb + wb) * 4 : ED_OOB);
        }
    };

    // lane-constant part of the B gather: tap j = li -> (r, s); pixel (row wv, column 2t+kk)
    const int tap = li < 25 ? li : 0;
    const int tr = tap / 5, ts = tap - tr * 5;
    const float* bq = bl + (ST * wv + tr) * RW + ts + ST * kk;
    const float* aq = sl + li * WC_SP + wv * WC_W + kk;

    int st = bx;
    if (st < n_stages) issue_loads(st);
    for (; st < n_stages; st += G) {
        __syncthreads();
        if (VEC) {
#pragma unroll
            for (int k = 0; k < WC_KS / 4; ++k) {
                const int e = tid + ED_THREADS * k;
                float* d = sl + (e >> 6) * WC_SP + 4 * (e & 63);     // (WC_SP is even: 8-byte aligned)
                *reinterpret_cast<float2*>(d) = make_float2(sr[4 * k], sr[4 * k + 1]);
                *reinterpret_cast<float2*>(d + 2) = make_float2(sr[4 * k + 2], sr[4 * k + 3]);
            }
        } else {
#pragma unroll
        for (int k = 0; k < WC_KS; ++k) {
            const int e = tid + ED_THREADS * k;
            sl[(e >> 8) * WC_SP + (e & (WC_TPX - 1))] = sr[k];
        }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int e = tid + ED_THREADS * k;
            if (e < IH * RW) bl[e] = br[k];
        }
        __syncthreads();
        if (st + G < n_stages) issue_loads(st + G);
#pragma unroll 8
        for (int t = 0; t < WC_W / 2; ++t) {
            const float av = aq[2 * t];
            if (bias_part) bsum += av;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[2 * ST * t], acc, 0, 0, 0);
        }
    }

    // combine the four waves (fixed order) and emit this workgroup's partial [a][tap]
    __syncthreads();
    float* red = sl;    // 4 x 16 x 64 floats
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (wv == 0 && li < 25) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = (red[e * 64 + lane] + red[(16 + e) * 64 + lane]) +
                            (red[(32 + e) * 64 + lane] + red[(48 + e) * 64 + lane]);
            const int a = (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a < g.Cs)
                part[((size_t)bch * G + bx) * (g.Cs * 25) + a * 25 + li] = v;
        }
    }
    if (bias_part && bch == 0) {
        // lanes (li, kk) of the four waves hold partial sums of channel li: fixed-order combine
        bsum += __shfl_xor(bsum, 32, 64);
        __syncthreads();
        if (lane < 32) red[wv * 32 + lane] = bsum;
        __syncthreads();
        if (tid < 32 && tid < g.Cs)
            bias_part[(size_t)bx * g.Cs + tid] =
                (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
    }
}

// ---------------------------------------------------------------------------------------------
// Second generation of the weight gradient above (offsets pt = pl = 1): same stages, MFMA roles and
// partial-sum layout, but the tiles arrive by 16-byte LDS-DMA, double buffered, issued from inside
// the MFMA stream of the previous stage -- the first generation moved every element with a dword
// load, five address instructions and a ds_write (38 loads per thread and stage: the vector ALU,
// not the HBM stream, set its pace: 4.0 TB/s).  Two accumulators break the 32-long dependent MFMA
// chain of a stage.
//   small tile: 32 rows (channels) of 256 contiguous floats, row stride 260 words (16-byte multiple;
//               the A read of 32 rows x 2 pixels is 2-way conflicted, one ds_read_b32 per MFMA);
//   big tile:   11 patch rows, image column wb at LDS column wb + 4, row stride 136 words; border
//               groups and rows outside the image are out-of-range reads = 0.0f.
// ---------------------------------------------------------------------------------------------
#define WD_SP 260
#define WD_RW (2 * WC_W + 8)
#define WD_BG (WC_IH * WD_RW / 4)                     // 374 groups of the big tile
#define WD_BGP ((WD_BG + 63) / 64 * 64)               // whole wave rows: 384
#define WD_BUF (32 * WD_SP + 4 * WD_BGP)              // floats per stage image
#define WD_ONES 128                                   // floats of 1.0f behind the two images (BIAS)
#define WD_LDS ((2 * WD_BUF + WD_ONES) * 4)

__device__ __forceinline__ void wd_dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voffset, soffset, 0, 0);
}

template <bool BIAS>
__global__ __launch_bounds__(ED_THREADS, 2) void k_wgrad_c1d(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    float* __restrict__ bias_part, BnGeom g, int n_stages, int stages_per_frame, int pair) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;
    WC_DECODE_GRID();
    const bool do_bias = BIAS && bch == 0;
    // (the small side may be a window of Cs channels in frames of css: the last frame's window ends
    // the buffer range)
    const int css = bn_cs_stride(g);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((((size_t)g.N - 1) * css + g.Cs) * PQ * 4), 0x00020000);
    // the big image is addressed from one row above its start (patch row y = image row 2 p0 - 1 + y)
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(big - g.Wb), 0, (int)(((size_t)g.N * g.Cb * HWb + g.Wb) * 4), 0x00020000);

    // rows of channels that do not exist are never copied: zero them once (both images)
    for (int e = tid; e < 2 * WD_BUF; e += ED_THREADS) {
        const int w = e % WD_BUF;
        if (w < 32 * WD_SP && w / WD_SP >= g.Cs) wsm[e] = 0.f;
    }
    // Conv2d bias gradient = sum over pixels of the small side: tap column 25 of the MFMA is not a
    // tap, its lanes read 1.0f as their B operand (a constant region behind the images, the same
    // immediate offsets as the tap lanes), so accumulator column 25 IS the bias partial -- no
    // vector instruction in the MFMA stream.  (Measured neutral: 34.7 us with either form.  The
    // 3 us between this variant (enc.conv0: dy fresh from the non-temporal stores of the kernel
    // before it) and the other (dec.convT4: 31.2 us) is not the bias arithmetic.)
    if (BIAS) for (int e = tid; e < WD_ONES; e += ED_THREADS) wsm[2 * WD_BUF + e] = 1.f;
    // DMA descriptors: small rows a = wv + 4k, group = lane; big groups e = lane + 64 (wv + 4k)
    const int svo = (wv * PQ + 4 * lane) * 4;
    int bvo[2], bcls = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = lane + 64 * (wv + 4 * k);
        const int y = e / (WD_RW / 4), c = e - y * (WD_RW / 4);
        const bool ok = e < WD_BG && c >= 1 && c <= 2 * WC_W / 4;
        bvo[k] = ok ? (y * g.Wb + 4 * (c - 1)) * 4 : ED_OOB;
        bcls |= ((y == 0 ? 1 : 0) | (y >= WC_IH - 2 ? 2 : 0)) << (2 * k);
    }
    auto issue_dma = [&](const int d, const int buf, const int st) __attribute__((always_inline)) {
        const int n = st / stages_per_frame;
        const int p0 = (st - n * stages_per_frame) * WC_ROWS;
        float* img = wsm + buf * WD_BUF;
        if (d < 8) {
            if (wv + 4 * d < g.Cs)                                   // wave-uniform
                wd_dma16(rs, img + (wv + 4 * d) * WD_SP, svo, ((n * css + 4 * d) * g.Hs + p0) * g.Ws * 4);
        } else {
            const int k = d - 8;
            if (64 * (wv + 4 * k) < WD_BGP) {                        // wave-uniform
                const int smask = ((p0 == 0 ? 1 : 0) | (p0 + WC_ROWS >= g.Hs ? 2 : 0)) << (2 * k);
                wd_dma16(rb, img + 32 * WD_SP + 4 * 64 * (wv + 4 * k), (bcls & smask) ? ED_OOB : bvo[k],
                         ((n * g.Cb + bch) * g.Hb + 2 * p0) * g.Wb * 4);
            }
        }
    };

    floatx16 acc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][e] = 0.f;

    // operand addresses: tap j = li -> (r, s); pixel (row wv, column 2t + kk)
    const int tap = li < 25 ? li : 0;
    const int tr = tap / 5, ts = tap - tr * 5;
    int aq_cur = (li * WD_SP + wv * WC_W + kk) * 4;                              // bytes
    int bq_cur = (32 * WD_SP + (2 * wv + tr) * WD_RW + ts + 2 * kk + 3) * 4;     // column wb + 4, pl = 1
    int aq_oth = aq_cur + WD_BUF * 4, bq_oth = bq_cur + WD_BUF * 4;
    if (do_bias && li >= 25) bq_cur = bq_oth = 2 * WD_BUF * 4;
    const char* sm = reinterpret_cast<const char*>(wsm);

    int st = bx;
    if (st < n_stages) {
#pragma unroll
        for (int d = 0; d < 10; ++d) issue_dma(d, 0, st);
    }
    int cur = 0;
    for (; st < n_stages; st += G) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nx = st + G;
        const bool more = nx < n_stages;
#pragma unroll
        for (int t = 0; t < WC_W / 2; ++t) {
            const float av = *reinterpret_cast<const float*>(sm + aq_cur + 8 * t);
            const float bv = *reinterpret_cast<const float*>(sm + bq_cur + 16 * t);
            acc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t & 1], 0, 0, 0);
            if (t >= 2 && t < 12) { if (more) issue_dma(t - 2, cur ^ 1, nx); }
        }
        cur ^= 1;
        int tmp = aq_cur; aq_cur = aq_oth; aq_oth = tmp;
        tmp = bq_cur; bq_cur = bq_oth; bq_oth = tmp;
    }

    // combine the two accumulators, then the four waves (fixed order): partial [a][tap]; column 25
    // of the bias variant holds the channel sums
    __syncthreads();
    float* red = wsm;    // 4 x 16 x 64 floats
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = acc[0][e] + acc[1][e];
    __syncthreads();
    if (wv == 0 && (li < 25 || (do_bias && li == 25))) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = (red[e * 64 + lane] + red[(16 + e) * 64 + lane]) +
                            (red[(32 + e) * 64 + lane] + red[(48 + e) * 64 + lane]);
            const int a = (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a < g.Cs) {
                if (li < 25) part[((size_t)bch * G + bx) * (g.Cs * 25) + a * 25 + li] = v;
                else         bias_part[(size_t)bx * g.Cs + a] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The DMA generation for STRIDE 1 (round 6): the first / last layer of a max-pooling architecture (1 <-> 16 channels on
// the full frame).  k_wgrad_c1<1, true> moved its stage through registers (8 16-byte loads, 19 ds_writes, two barriers
// per stage: 90 us for 285 MB).  Same stages (4 rows x 64 columns of the small map = the big map), MFMA roles and
// partial layout as k_wgrad_c1d; what differs is the geometry of the tiles:
//   small tile: a stage's four rows lie Ws apart in memory (Ws = 64 k): lane = 16-byte group (row lane / 16, column
//               4 (lane % 16)), offset computed once, a stage adds a scalar;
//   big tile:   8 patch rows (4 + 4) of 72 words: image column wb at LDS column wb - q00 + 4, any offsets pt, pl <= 4;
//               the rows above / below the frame and the groups left / right of it are out-of-range reads (per-lane
//               row number and column class against the stage's scalars), interior column blocks read their neighbours'.
// ---------------------------------------------------------------------------------------------
#define WE_IH (WC_ROWS + 4)
#define WE_RW (WC_W + 8)
#define WE_BG (WE_IH * WE_RW / 4)                     // 144 groups of the big tile
#define WE_BGP ((WE_BG + 63) / 64 * 64)               // whole wave rows: 192
#define WE_BUF (32 * WD_SP + 4 * WE_BGP)
#define WE_LDS ((2 * WE_BUF + WD_ONES) * 4)

// M16: at most 16 small-side channels (the max-pooling test architecture's 1 <-> 16 layers) on v_mfma_f32_16x16x4_f32 --
// rows = 16 channels, reduction = 4 pixels, columns = 16 taps, two instructions (taps 0-15, 16-31) per 4 pixels: half the
// matrix time of the 32-row form, whose 32 x 64-cycle instructions per stage row were the pace (4.9 TB/s at best).
template <bool BIAS, bool M16>
__global__ __launch_bounds__(ED_THREADS, 2) void k_wgrad_c1e(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    float* __restrict__ bias_part, BnGeom g, int n_stages, int stages_per_frame, int cblocks, int pair) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;
    WC_DECODE_GRID();
    const bool do_bias = BIAS && bch == 0;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    // the big image is addressed from pt rows and 4 columns in front of its start (patch row y = image row
    // p0 - pt + y, LDS column c = image column q00 - 4 + c): what lies in front of a frame is masked, not read
    const int shift = g.pt * g.Wb + 4;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(big - shift), 0, (int)(((size_t)g.N * g.Cb * HWb + shift) * 4), 0x00020000);

    for (int e = tid; e < 2 * WE_BUF; e += ED_THREADS) {
        const int w = e % WE_BUF;
        if (w < 32 * WD_SP && w / WD_SP >= g.Cs) wsm[e] = 0.f;
    }
    if (BIAS) for (int e = tid; e < WD_ONES; e += ED_THREADS) wsm[2 * WE_BUF + e] = 1.f;
    // DMA descriptors: small rows a = wv + 4 d, group = lane; big groups e = lane + 64 wv (waves 0 .. 2)
    const int svo = (wv * PQ + (lane >> 4) * g.Ws + 4 * (lane & 15)) * 4;
    const int be = lane + 64 * wv;
    const int by = be / (WE_RW / 4), bc = be - by * (WE_RW / 4);
    const int bvo = (by * g.Wb + 4 * bc) * 4;
    const bool b_in = be < WE_BG;
    auto issue_dma = [&](const int d, const int buf, const int st) __attribute__((always_inline)) {
        const int n = st / stages_per_frame;
        const int rem = st - n * stages_per_frame;
        const int rblk = rem / cblocks;
        const int p0 = rblk * WC_ROWS, q00 = (rem - rblk * cblocks) * WC_W;
        float* img = wsm + buf * WE_BUF;
        if (d < 8) {
            if (wv + 4 * d < g.Cs)                                   // wave-uniform
                wd_dma16(rs, img + (wv + 4 * d) * WD_SP, svo, (((n * g.Cs + 4 * d) * g.Hs + p0) * g.Ws + q00) * 4);
        } else if (64 * wv < WE_BGP) {                               // wave-uniform
            const int hb = p0 - g.pt + by;                           // image row of this lane's group
            const bool ok = b_in && hb >= 0 && hb < g.Hb && (bc > 0 || q00 > 0) &&
                            (bc < WE_RW / 4 - 1 || q00 + WC_W < g.Wb);
            wd_dma16(rb, img + 32 * WD_SP + 4 * 64 * wv, ok ? bvo : ED_OOB,
                     (((n * g.Cb + bch) * g.Hb + p0) * g.Wb + q00) * 4);
        }
    };

    typedef float floatx4w __attribute__((ext_vector_type(4)));
    floatx16 acc[2];
    floatx4w acc4[2][2];                              // M16: [tap block][even / odd step]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][e] = 0.f;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) acc4[h][b2] = (floatx4w){0.f, 0.f, 0.f, 0.f};
    }

    // operand addresses: tap j -> (r, s); 32-row form: tap li, pixel (row wv, column 2t + kk);
    // M16: channel / tap lane & 15, pixel (row wv, column 4t + lane / 16), tap blocks lane & 15 and 16 + lane & 15
    const int l16 = lane & 15, kq = lane >> 4;
    const int tap = M16 ? l16 : (li < 25 ? li : 0);
    const int tr = tap / 5, ts = tap - tr * 5;
    const int tap2 = 16 + l16 < 25 ? 16 + l16 : 0;
    const int tr2 = tap2 / 5, ts2 = tap2 - tr2 * 5;
    int aq_cur = M16 ? (l16 * WD_SP + wv * WC_W + kq) * 4 : (li * WD_SP + wv * WC_W + kk) * 4;   // bytes
    int bq_cur = (32 * WD_SP + (wv + tr) * WE_RW + ts + (M16 ? kq : kk) + 4 - g.pl) * 4;         // column wb - q00 + 4
    int b2_cur = (32 * WD_SP + (wv + tr2) * WE_RW + ts2 + kq + 4 - g.pl) * 4;
    int aq_oth = aq_cur + WE_BUF * 4, bq_oth = bq_cur + WE_BUF * 4, b2_oth = b2_cur + WE_BUF * 4;
    if (!M16 && do_bias && li >= 25) bq_cur = bq_oth = 2 * WE_BUF * 4;
    if (M16 && do_bias && l16 == 9) b2_cur = b2_oth = 2 * WE_BUF * 4;             // tap column 25 = the bias sums
    const char* sm = reinterpret_cast<const char*>(wsm);

    int st = bx;
    if (st < n_stages) {
#pragma unroll
        for (int d = 0; d < 9; ++d) issue_dma(d, 0, st);
    }
    int cur = 0;
    for (; st < n_stages; st += G) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nx = st + G;
        const bool more = nx < n_stages;
        if constexpr (M16) {
#pragma unroll
            for (int t = 0; t < WC_W / 4; ++t) {
                const float av = *reinterpret_cast<const float*>(sm + aq_cur + 16 * t);
                const float bv = *reinterpret_cast<const float*>(sm + bq_cur + 16 * t);
                const float b2 = *reinterpret_cast<const float*>(sm + b2_cur + 16 * t);
                acc4[t & 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc4[t & 1][0], 0, 0, 0);
                acc4[t & 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2, acc4[t & 1][1], 0, 0, 0);
                if (t >= 2 && t < 11) { if (more) issue_dma(t - 2, cur ^ 1, nx); }
            }
        } else {
#pragma unroll
            for (int t = 0; t < WC_W / 2; ++t) {
                const float av = *reinterpret_cast<const float*>(sm + aq_cur + 8 * t);
                const float bv = *reinterpret_cast<const float*>(sm + bq_cur + 8 * t);     // (bias lanes: inside the 1.0f region)
                acc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t & 1], 0, 0, 0);
                if (t >= 2 && t < 11) { if (more) issue_dma(t - 2, cur ^ 1, nx); }
            }
        }
        cur ^= 1;
        int tmp = aq_cur; aq_cur = aq_oth; aq_oth = tmp;
        tmp = bq_cur; bq_cur = bq_oth; bq_oth = tmp;
        tmp = b2_cur; b2_cur = b2_oth; b2_oth = tmp;
    }

    if constexpr (M16) {
        // register e of block b2 = channel 4 (lane / 16) + e, tap 16 b2 + lane % 16; the four waves in fixed order
        __syncthreads();
        float* red = wsm;    // 4 waves x 2 blocks x 4 x 64 floats
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[((wv * 2 + b2) * 4 + e) * 64 + lane] = acc4[0][b2][e] + acc4[1][b2][e];
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int tp = 16 * b2 + l16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = (b2 * 4 + e) * 64 + lane;
                    const float v = (red[o] + red[512 + o]) + (red[1024 + o] + red[1536 + o]);
                    const int a = 4 * kq + e;
                    if (a < g.Cs) {
                        if (tp < 25) part[((size_t)bch * G + bx) * (g.Cs * 25) + a * 25 + tp] = v;
                        else if (do_bias && tp == 25) bias_part[(size_t)bx * g.Cs + a] = v;
                    }
                }
            }
        }
        return;
    }

    __syncthreads();
    float* red = wsm;    // 4 x 16 x 64 floats
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = acc[0][e] + acc[1][e];
    __syncthreads();
    if (wv == 0 && (li < 25 || (do_bias && li == 25))) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = (red[e * 64 + lane] + red[(16 + e) * 64 + lane]) +
                            (red[(32 + e) * 64 + lane] + red[(48 + e) * 64 + lane]);
            const int a = (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a < g.Cs) {
                if (li < 25) part[((size_t)bch * G + bx) * (g.Cs * 25) + a * 25 + li] = v;
                else         bias_part[(size_t)bx * g.Cs + a] = v;
            }
        }
    }
}

// stride-1 layers the DMA generation takes: whole stages (maps of 64 k columns, 4 k rows), "same"-sized maps
static bool wgrad_c1e_ok(const BnGeom& g) {
    return g.stride == 1 && g.R == 5 && g.S == 5 && g.Cs <= 32 && g.CsS == 0 && (g.Ws % WC_W) == 0 &&
           (g.Hs % WC_ROWS) == 0 && g.Hb == g.Hs && g.Wb == g.Ws && g.pt <= 4 && g.pl <= 4;
}

static inline int wgrad_c1_rblocks(const BnGeom& g) { return (g.Hs + WC_ROWS - 1) / WC_ROWS; }
static inline int wgrad_c1_cblocks(const BnGeom& g) { return (g.Ws + WC_W - 1) / WC_W; }
static int wgrad_c1_grid(const BnGeom& g) {
    const int n_stages = g.N * wgrad_c1_rblocks(g) * wgrad_c1_cblocks(g);
    // resident workgroups per CU in total: 3 (first generation), 2 (DMA generation: 79 KB of LDS)
    const int cap = (((g.pt == 1 && g.pl == 1) || wgrad_c1e_ok(g)) ? 512 : 768) / (g.Cb > 0 ? g.Cb : 1);
    return n_stages < cap ? n_stages : cap;
}

BnFastPlan bn_edge_wgrad_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || (g.stride != 2 && g.stride != 1) || g.Cb > 4) return p;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return p;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return p;
    if (g.stride == 1) {
        // stride 1 (round 4): any map size, offsets up to the kernel size, the first-generation kernel
        if (g.Cs > 32 || g.pt > 4 || g.pl > 4 || (g.CsS > 0 && g.CsS != g.Cs)) return p;
        p.supported = true;
        p.variant = 1;
        p.d = wgrad_c1_grid(g);
        p.ws_bytes = ((size_t)g.Cb * p.d * g.Cs * 25 + (size_t)p.d * 32) * sizeof(float);
        p.kernel_name = wgrad_c1e_ok(g) ? "k_wgrad_c1e" : "k_wgrad_c1<1>";
        return p;
    }
    if (g.Cs > 32) return p;
    if (g.Ws != WC_W || (g.Hs % WC_ROWS) != 0 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) {
        // round 4: any other map (64x48, 192x160 frames) in row / column blocks on the first-generation
        // kernel -- no zero-padded or tiled copies of the two operands
        static int off = -1;                          // BN_WGRAD_C1G=0: off
        if (off < 0) { const char* e = bn_tune_env("BN_WGRAD_C1G"); off = (e && e[0] == '0') ? 1 : 0; }
        if (off || g.pt > 4 || g.pl > 4 || (g.CsS > 0 && g.CsS != g.Cs)) return p;
        p.supported = true;
        p.variant = 1;
        p.d = wgrad_c1_grid(g);
        p.ws_bytes = ((size_t)g.Cb * p.d * g.Cs * 25 + (size_t)p.d * 32) * sizeof(float);
        p.kernel_name = "k_wgrad_c1<2>";
        return p;
    }
    p.supported = true;
    p.d = wgrad_c1_grid(g);
    p.ws_bytes = ((size_t)g.Cb * p.d * g.Cs * 25 + (size_t)p.d * 32) * sizeof(float);
    p.kernel_name = (g.pt == 1 && g.pl == 1) ? "k_wgrad_c1d" : "k_wgrad_c1<2>";
    return p;
}

int bn_launch_edge_wgrad(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                         const BnGeom& g, int accumulate, void* ws, hipStream_t st, float* db,
                         int bias_side, bool* bias_done) {
    const int spf = wgrad_c1_rblocks(g) * wgrad_c1_cblocks(g), n_stages = g.N * spf;
    // Conv2d bias gradient (sum of the small side) as a by-product
    float* bias_part = (db && bias_side == 1)
        ? (float*)ws + (size_t)g.Cb * plan.d * g.Cs * 25 : nullptr;
    if (g.CsS > 0 && g.CsS != g.Cs && !(g.pt == 1 && g.pl == 1)) return BN_E_SHAPE;   // k_wgrad_c1d only
    const bool vec = (g.Ws & 3) == 0 && (((uintptr_t)small) & 15u) == 0 && g.CsS == 0;
    // several big-side channels: their workgroups of a stage together on one XCD (WC_DECODE_GRID)
    const int pair = (g.Cb > 1 && (plan.d & 7) == 0) ? 1 : 0;
    const dim3 wgrid = pair ? dim3(plan.d * g.Cb) : dim3(plan.d, g.Cb);
    if (g.stride == 1 && wgrad_c1e_ok(g) && vec) {
        static bool attr_set = false;
        if (!attr_set) {
            const void* fns[4] = {(const void*)k_wgrad_c1e<true, false>, (const void*)k_wgrad_c1e<false, false>,
                                  (const void*)k_wgrad_c1e<true, true>, (const void*)k_wgrad_c1e<false, true>};
            for (const void* fn : fns) {
                hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WE_LDS);
                if (e1 != hipSuccess) return (int)e1;
            }
            attr_set = true;
        }
#define WE_LAUNCH(B, M)                                                                                     \
        BN_LAUNCH_MAIN((k_wgrad_c1e<B, M>), wgrid, dim3(ED_THREADS), WE_LDS, st, small, big, (float*)ws,   \
                           bias_part, g, n_stages, spf, wgrad_c1_cblocks(g), pair)
        if (g.Cs <= 16) { if (bias_part) WE_LAUNCH(true, true); else WE_LAUNCH(false, true); }
        else { if (bias_part) WE_LAUNCH(true, false); else WE_LAUNCH(false, false); }
#undef WE_LAUNCH
    } else if (g.stride == 1) {
        if (vec)
            BN_LAUNCH_MAIN((k_wgrad_c1<1, true>), wgrid, dim3(ED_THREADS), 0, st, small, big,
                               (float*)ws, bias_part, g, n_stages, spf, wgrad_c1_cblocks(g), pair);
        else
            BN_LAUNCH_MAIN((k_wgrad_c1<1, false>), wgrid, dim3(ED_THREADS), 0, st, small, big,
                               (float*)ws, bias_part, g, n_stages, spf, wgrad_c1_cblocks(g), pair);
    } else if (g.pt == 1 && g.pl == 1 && plan.variant != 1) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e1 = hipFuncSetAttribute((const void*)k_wgrad_c1d<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
            hipError_t e2 = hipFuncSetAttribute((const void*)k_wgrad_c1d<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
            if (e1 != hipSuccess) return (int)e1;
            if (e2 != hipSuccess) return (int)e2;
            attr_set = true;
        }
        if (bias_part)
            BN_LAUNCH_MAIN(k_wgrad_c1d<true>, wgrid, dim3(ED_THREADS), WD_LDS, st, small,
                               big, (float*)ws, bias_part, g, n_stages, g.Hs / WC_ROWS, pair);
        else
            BN_LAUNCH_MAIN(k_wgrad_c1d<false>, wgrid, dim3(ED_THREADS), WD_LDS, st, small,
                               big, (float*)ws, bias_part, g, n_stages, g.Hs / WC_ROWS, pair);
    } else if (vec) {
        BN_LAUNCH_MAIN((k_wgrad_c1<2, true>), wgrid, dim3(ED_THREADS), 0, st, small, big,
                           (float*)ws, bias_part, g, n_stages, spf, wgrad_c1_cblocks(g), pair);
    } else {
        BN_LAUNCH_MAIN((k_wgrad_c1<2, false>), wgrid, dim3(ED_THREADS), 0, st, small, big,
                           (float*)ws, bias_part, g, n_stages, spf, wgrad_c1_cblocks(g), pair);
    }
    BN_LAUNCH_CHECK();
    if (bias_part) {
        const int rc = bn_launch_sum_partials(bias_part, db, g.Cs, plan.d, accumulate, 0, 0, st);
        if (rc) return rc;
        if (bias_done) *bias_done = true;
    }
    // per big-side channel b: partial rows [split][a][tap] -> dW[a][b][tap]
    for (int b = 0; b < g.Cb; ++b) {
        const int rc = bn_launch_sum_partials(
            (const float*)ws + (size_t)b * plan.d * g.Cs * 25, dw + b * 25, g.Cs * 25, plan.d,
            accumulate, 0, 0, st, 25, g.Cb * 25);
        if (rc) return rc;
    }
    return 0;
}

// =============================================================================================
// gather-down with one big-side channel (enc.conv0 forward, dec.convT4 data gradient):
//   out[n,m,p,q] = epi( sum_{r,s} big[n,0,2p+r-pt,2q+s-pl] * W[m][0][r][s] ),   m < 32
// Persistent, wave-autonomous kernel: a workgroup is ONE wave and owns a private LDS arena, so
// there is no barrier anywhere.  The work unit is a pair of output rows of one frame
// (2 x 64 pixels x 32 channels = 16 KB of output, from a 7 x 128 input patch); wave g handles
// units g, g + G, g + 2G, ...  and the grid size G is chosen by the host so that every CU gets
// the same number of units (N = 200: 25 units per CU = 9 waves x <= 3 units).
// Per unit:
//  * the patch of the NEXT unit is already in flight (float4 buffer loads into registers issued
//    one unit ahead); the current one goes to LDS with ds_write_b128, zero borders included
//    (out-of-range loads return 0);
//  * MFMA roles: rows = 32 output pixels of one image row (A = gathered input), columns =
//    output channels (B = weights, lane-resident), reduction = 25 taps (13 steps, the 26th tap
//    is a zero weight);
//  * the accumulators of one output row (64 pixels x 32 channels) are transposed through the
//    wave's LDS slab so that 16 consecutive lanes store one 256-byte image row of one channel:
//    every store instruction writes four full rows (8 HBM lines).
// =============================================================================================
#define DC_W 64
#define DC_ROWS 2                        // output rows per unit
#define DC_IH (2 * DC_ROWS + 3)          // patch rows (7)
#define DC_X0 4                          // LDS column of image column 0
#define DC_RW (2 * DC_W + 8)             // 4 zero columns | 128 image columns | 4 zero columns
#define DC_C4 (DC_RW / 4)                // float4 slots per patch row (34)
#define DC_NLD ((DC_IH * DC_C4 + 63) / 64)   // float4 loads per lane per unit (4)
#define DC_TS (DC_W + 4)                 // transpose slab row stride (68: b128-conflict-free)
#define DC_SLAB (32 * DC_TS)
#ifndef DC_HALF
#define DC_HALF 1                        // 1: half-row transposition slab, 4 waves per SIMD
#endif
#if DC_HALF
#define DC_MAX_WAVES_PER_CU 16           // 8.4 KB of LDS per wave, <= 128 VGPRs
#define DC_WPE 4
#else
#define DC_MAX_WAVES_PER_CU 12           // 12.5 KB of LDS per wave, <= 168 VGPRs
#define DC_WPE 3
#endif
#define DC_HTS 36                        // half-row slab row stride
#ifndef DC_VARIANT
#define DC_VARIANT 5                     // product build, float frames, forward, 32 channels: fourth
                                         // generation on 8-row strips (k_down_c1p_*_s8; 4 = the same on
                                         // 16-row strips, 3 = third generation k_down_c1w) where the
                                         // geometry holds, else the first generation (0).  Same bits.
                                         // tools/lab/e0_lab.hip, 256 frames, inputs and outputs rotated
                                         // through > 256 MB, behind clean L2s / a 35 MB memset:
                                         // generation 1 31.5 / 32.8 us, 2 36.6 / 37.1, 3 31.4 / 33.5,
                                         // 4 on 16-row strips 30.9 / 31.9, on 8-row strips 27.3 / 27.9.
                                         // uint8 frames and the masked data gradient: second generation
#endif

#ifndef DC_ST_AUX
#define DC_ST_AUX 16                     // cache policy of the first generation's 16-byte output stores
                                         // (buffer stores; -1 = plain global stores).  16 = sc1, write-
                                         // through: the 134 MB output stream does not park dirty lines in
                                         // the 4 MB L2s.  In the training step: plain 33.0-33.3 us, nt (2)
                                         // 34.9, sc1 31.5-31.9, sc1+nt (18) 34.2
#endif

// LeakyReLU of four values, max(v, slope v) for 0 <= slope <= 1: two packed multiplies and four
// v_max.  `fmaxf(v, v * slope)` costs THREE vector instructions per value (the compiler quiets a
// possible signalling NaN with an extra v_max): 96 of them per 26 MFMAs in enc.conv0's loop, where
// every vector instruction takes its cycles from the matrix pipe.
typedef float floatx2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ed_lrelu4(float (&v)[4], float slope) {
    const floatx2p s2 = {slope, slope};
    floatx2p lo = {v[0], v[1]}, hi = {v[2], v[3]}, mlo, mhi;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(mlo) : "v"(lo), "v"(s2));
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(mhi) : "v"(hi), "v"(s2));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[0]) : "v"(lo.x), "v"(mlo.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[1]) : "v"(lo.y), "v"(mlo.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[2]) : "v"(hi.x), "v"(mhi.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[3]) : "v"(hi.y), "v"(mhi.y));
}

// tools/lab/e0_lab.hip -DE0_SMALLOUT: every store lands in one 1 MB window (no HBM write stream): what
// the kernel costs when the output is free
#ifdef E0_SMALLOUT
#define E0_OUT_OFFSET(o) ((o) & 0xffff0)
#else
#define E0_OUT_OFFSET(o) (o)
#endif

typedef int intx4 __attribute__((ext_vector_type(4)));
typedef float floatx4e __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4e __attribute__((ext_vector_type(4)));

// GENW (round 4): any map with a width that is a multiple of 4 -- a unit is two output rows of ONE block of 64
// columns (`ncb` blocks per row, the last one masked): the patch starts 128 cb columns into the row and has
// real neighbours on its inner sides (the four-column borders are loaded, not zeros), the rows of the output
// lie g.Ws apart, an odd last row is masked.  No zero-padded or tiled copies of 64x48 / 192x192 frames.
template <int ACT, bool MASK, bool GENW = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DC_WPE, DC_WPE))) void k_down_c1(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, float slope,
    int units) {
    __shared__ __attribute__((aligned(16))) float bl[DC_IH * DC_RW];
    __shared__ __attribute__((aligned(16))) float tw[DC_HALF ? 32 * DC_HTS : DC_SLAB];
    const int lane = threadIdx.x;
    const int li = lane & 31, kk = lane >> 5;
    const int ncb = GENW ? (g.Ws + DC_W - 1) / DC_W : 1;         // column blocks per row
    const int rpf = GENW ? (g.Hs + DC_ROWS - 1) / DC_ROWS : g.Hs / DC_ROWS;   // row pairs per frame
    const int upf = rpf * ncb;                       // units per frame
    const int HWb = g.Hb * g.Wb, PQ = g.Hs * g.Ws;
#ifdef E0_TRACE      // tools/lab/e0_lab.hip: s_memrealtime marks per wave (100 MHz)
    unsigned long long* trc = e0_trace + (size_t)blockIdx.x * 8;
#define E0_MARK(slot) do { if (threadIdx.x == 0) trc[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
    E0_MARK(0);
    if (threadIdx.x == 0) trc[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
#else
#define E0_MARK(slot)
#endif

    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * HWb * 4), 0x00020000);

    // lane-constant part of the patch decode: slot e = lane + 64k -> (row y, float4 column c4)
    int ld_y[DC_NLD], ld_off[DC_NLD];
#pragma unroll
    for (int k = 0; k < DC_NLD; ++k) {
        const int e = lane + 64 * k;
        const int y = e / DC_C4, c4 = e - y * DC_C4;
        // (GENW: the border slots are image columns of the neighbouring blocks, tested per unit)
        const bool col_ok = e < DC_IH * DC_C4 && (GENW || (c4 >= 1 && c4 <= DC_W / 2));
        ld_y[k] = col_ok ? y : -0x10000;            // fails the row test below
        ld_off[k] = (y * g.Wb + 4 * (c4 - 1)) * 4;
    }
    int ld_col[DC_NLD];                              // GENW: image column of the slot inside its block
#pragma unroll
    for (int k = 0; k < DC_NLD; ++k) {
        const int e = lane + 64 * k;
        ld_col[k] = 4 * (e - (e / DC_C4) * DC_C4 - 1);
    }
    auto issue = [&](int u, intx4 (&st)[DC_NLD]) {
        const int n = u / upf;
        const int ur = u - n * upf;
        const int cb = GENW ? ur % ncb : 0;
        const int rp = GENW ? ur / ncb : ur;
        const int hb0 = 2 * DC_ROWS * rp - g.pt;                  // image row of patch row 0
        const int base = ((n * g.Hb + hb0) * g.Wb + 2 * DC_W * cb) * 4;
#pragma unroll
        for (int k = 0; k < DC_NLD; ++k) {
            const int hb = hb0 + ld_y[k];
            bool ok = hb >= 0 && hb < g.Hb;
            if (GENW) {
                const int wb = 2 * DC_W * cb + ld_col[k];
                ok = ok && wb >= 0 && wb < g.Wb;
            }
            st[k] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? base + ld_off[k] : ED_OOB, 0, 0);
        }
    };

    // B operand: weights of output channel li for taps (2t + kk); lane-constant tap offsets of
    // the A gather (pixel li of the block, tap 2t + kk)
    float wv_[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) {
        const int tap = 2 * t + kk;
        wv_[t] = (tap < 25 && li < g.Cs) ? w[li * 25 + tap] : 0.f;
    }
    // patch offset of tap 2t + kk = offset of tap 2t (a compile-time constant, folded into the
    // ds_read offset field) + kk * (1, or DC_RW - 4 when tap 2t ends a kernel row): two
    // lane-dependent base pointers instead of thirteen offset registers
    const int kkA = kk, kkB = kk * (DC_RW - 4);
    const float bz = (bias && li < g.Cs) ? bias[li] : 0.f;
    const int a_col = DC_X0 - g.pl + 2 * li;

    intx4 stage[DC_NLD];
    int u = blockIdx.x;
    if (u < units) issue(u, stage);
#pragma unroll 1
    for (; u < units; u += gridDim.x) {
        // LDS operations of one wave execute in order: the previous unit's reads are done
#pragma unroll
        for (int k = 0; k < DC_NLD; ++k) {
            const int e = lane + 64 * k;
            if (e < DC_IH * DC_C4) *reinterpret_cast<intx4*>(bl + 4 * e) = stage[k];
        }
        if (u + (int)gridDim.x < units) issue(u + gridDim.x, stage);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (u == (int)blockIdx.x) E0_MARK(1);        // first patch in LDS

        const int n = u / upf;
        const int ur = u - n * upf;
        const int cbo = GENW ? ur % ncb : 0;                     // this unit's column block
        const int p0 = DC_ROWS * (GENW ? ur / ncb : ur);
        const int opitch = GENW ? g.Ws : DC_W;                   // floats between output rows
        const float* aq = bl + a_col;
#pragma unroll 1
        for (int pr = 0; pr < DC_ROWS; ++pr) {
            // two independent accumulators (the half-rows), interleaved step by step: a 13-long
            // dependent chain would stall the issue after every MFMA.  The stores of this row
            // drain while the next row's (and the other waves') MFMAs run.
            floatx16 acc[2];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[qh][e] = bz;   // a lane owns ONE channel: bias = init
            const float* ar = aq + (2 * pr) * DC_RW;
            const float* arA = ar + kkA;
            const float* arB = ar + kkB;
#pragma unroll
            for (int t = 0; t < 13; ++t) {
                // tap 24 + kk = 25 (t = 12, kk = 1) has a zero weight but must still read a FINITE
                // value: it takes the next column of the last patch row, which is always written
                const int t0 = ((2 * t) / 5) * DC_RW + (2 * t) % 5;
                const float* at = ((2 * t) % 5 == 4 && t != 12) ? arB : arA;
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
                    acc[qh] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[t0 + 64 * qh], wv_[t],
                                                                   acc[qh], 0, 0, 0);
            }
            if (u == (int)blockIdx.x && pr == 0) E0_MARK(2);     // first row multiplied
#if DC_HALF
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    floatx4e v = {acc[qh][4 * grp], acc[qh][4 * grp + 1], acc[qh][4 * grp + 2],
                                  acc[qh][4 * grp + 3]};
                    if (ACT == BN_ACT_LRELU) {
                        float q[4] = {v.x, v.y, v.z, v.w};
                        ed_lrelu4(q, slope);
                        v = (floatx4e){q[0], q[1], q[2], q[3]};
                    }
                    *reinterpret_cast<floatx4e*>(tw + li * DC_HTS + 8 * grp + 4 * kk) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // 8 lanes x 16 B = one 128-byte half row; 8 channels per store instruction
                const int ocol = DC_W * cbo + 32 * qh + 4 * (lane & 7);
                const size_t row0 = ((size_t)n * g.Cs * g.Hs + (p0 + pr)) * opitch + ocol;
                const bool oin = !GENW || (ocol < g.Ws && p0 + pr < g.Hs);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 8 * i + (lane >> 3);
                    floatx4e v = *reinterpret_cast<const floatx4e*>(tw + ch * DC_HTS + 4 * (lane & 7));
                    if (ch < g.Cs && oin) {
                        const size_t o = row0 + (size_t)ch * PQ;
                        if (MASK) {
                            const floatx4e d = *reinterpret_cast<const floatx4e*>(dact_src + o);
                            v.x *= d.x > 0.f ? 1.f : slope; v.y *= d.y > 0.f ? 1.f : slope;
                            v.z *= d.z > 0.f ? 1.f : slope; v.w *= d.w > 0.f ? 1.f : slope;
                        }
#if DC_ST_AUX >= 0
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(uintx4e, v),
                            __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffffc, 0x00020000),
                            E0_OUT_OFFSET((int)(o * 4)), 0, DC_ST_AUX);
#else
                        *reinterpret_cast<floatx4e*>(out + o) = v;
#endif
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#else
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                // lane holds, for channel li, pixels 32*qh + 8*grp + 4*kk + {0..3}
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    floatx4e v = {acc[qh][4 * grp], acc[qh][4 * grp + 1], acc[qh][4 * grp + 2],
                                  acc[qh][4 * grp + 3]};
                    if (ACT == BN_ACT_LRELU) {
                        float q[4] = {v.x, v.y, v.z, v.w};
                        ed_lrelu4(q, slope);
                        v = (floatx4e){q[0], q[1], q[2], q[3]};
                    }
                    *reinterpret_cast<floatx4e*>(tw + li * DC_TS + 32 * qh + 8 * grp + 4 * kk) = v;
                }
            }
            // the asm only keeps the compiler from moving the slab reads above the writes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const size_t row0 = ((size_t)n * g.Cs * g.Hs + (p0 + pr)) * DC_W + 4 * (lane & 15);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = 4 * i + (lane >> 4);
                floatx4e v = *reinterpret_cast<const floatx4e*>(tw + ch * DC_TS + 4 * (lane & 15));
                if (ch < g.Cs) {
                    const size_t o = row0 + (size_t)ch * PQ;
                    if (MASK) {
                        const floatx4e d = *reinterpret_cast<const floatx4e*>(dact_src + o);
                        v.x *= d.x > 0.f ? 1.f : slope; v.y *= d.y > 0.f ? 1.f : slope;
                        v.z *= d.z > 0.f ? 1.f : slope; v.w *= d.w > 0.f ? 1.f : slope;
                    }
                    *reinterpret_cast<floatx4e*>(out + o) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            if (u == (int)blockIdx.x && pr == 0) E0_MARK(3);     // first row's stores issued
        }
        if (u == (int)blockIdx.x) E0_MARK(4);                    // first unit done
    }
    E0_MARK(5);                                                  // all stores issued
#ifdef E0_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    E0_MARK(6);                                                  // all stores acknowledged
#endif
}

// ---------------------------------------------------------------------------------------------
// second generation of the gather-down edge kernel: the MFMA roles are swapped -- rows = output
// CHANNELS (A = weights, lane-resident), columns = 32 output PIXELS of one image row (B = the
// gathered input) -- so that an accumulator register holds one channel's value for 32 adjacent
// pixels across 32 lanes: every accumulator goes to HBM with ONE dword store whose two lane halves
// are two full 128-byte lines (channels ch and ch + 4).  No transposition slab, no LDS traffic in
// the epilogue: the wave's LDS arena is the input patch only (3.8 / 6 KB for 2- / 4-row units).
// U8: the frames are read as stored on disk (uint8, reference data_generator.py:251-263) and
// converted in flight, value / 255 with an IEEE division = numpy's astype(float32) / 255.
// ---------------------------------------------------------------------------------------------
// CB = 2 (two-channel frames, e.g. the PS-VAE's two camera views): the reduction index of the
// 32x32x2 MFMA is the CHANNEL (lane half kk), one step per tap; a patch per channel in LDS.
// GENW (round 4, two-channel float frames of any size with a width that is a multiple of 4): a unit is ROWS
// output rows of ONE block of 64 columns, as in k_down_c1<.., GENW> -- the four-column borders of the patch are
// image columns of the neighbouring blocks (34 instead of 32 16-byte slots per patch row, tested per unit),
// output rows lie g.Ws apart, the last block / rows below the map are masked.
template <int ACT, bool MASK, bool U8, int ROWS, int CB = 1, bool GENW = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CB == 1 ? 4 : 3, CB == 1 ? 4 : 3))) void k_down_c1s(
    const void* __restrict__ big_, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, BnGeom g, float slope,
    int units) {
    constexpr int IH = 2 * ROWS + 3;                     // patch rows
    constexpr int PER_ROW = U8 ? 8 : (GENW ? DC_W / 2 + 2 : DC_W / 2);   // 16-byte loads per patch row
    constexpr int NLD = (IH * PER_ROW + 63) / 64;
    static_assert(CB == 1 || !U8, "uint8 frames: one channel");
    static_assert(!GENW || !U8, "column blocks: float frames");
    __shared__ __attribute__((aligned(16))) float bl[CB * IH * DC_RW];
    const int lane = threadIdx.x;
    const int li = lane & 31, kk = lane >> 5;
    const int ncb = GENW ? (g.Ws + DC_W - 1) / DC_W : 1;
    const int upf = GENW ? ((g.Hs + ROWS - 1) / ROWS) * ncb : g.Hs / ROWS;   // units per frame
    const int HWb = g.Hb * g.Wb, PQ = g.Hs * g.Ws;

    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big_, 0, (int)((size_t)g.N * CB * HWb * (U8 ? 1 : 4)), 0x00020000);
    const int css = bn_cs_stride(g);         // output (and mask): a window of Cs channels in frames of css
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)out, 0, (int)((((size_t)g.N - 1) * css + g.Cs) * PQ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(MASK ? dact_src : out), 0, (int)((((size_t)g.N - 1) * css + g.Cs) * PQ * 4), 0x00020000);

    // the zero columns left and right of the image never change: written once
    for (int e = lane; e < CB * IH * 8; e += 64) {
        const int y = e >> 3, c = e & 7;                 // y runs over the rows of all patches
        bl[y * DC_RW + (c < 4 ? c : DC_X0 + 2 * DC_W + c - 4)] = 0.f;
    }

    // lane-constant part of the patch decode: slot e = lane + 64k -> (row y, 16-byte column)
    int ld_y[NLD], ld_off[NLD], ld_lds[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = lane + 64 * k;
        const int y = e / PER_ROW, c = e - y * PER_ROW - (GENW ? 1 : 0);   // GENW: slot -1 = the left border
        ld_y[k] = e < IH * PER_ROW ? y : -0x10000;       // fails the row test below
        ld_off[k] = U8 ? (y * g.Wb + 16 * c) : (y * g.Wb + 4 * c) * 4;
        ld_lds[k] = y * DC_RW + DC_X0 + (U8 ? 16 : 4) * c;
    }
    auto issue = [&](int u, intx4 (&st)[CB * NLD]) {
        const int n = u / upf;
        const int ur = u - n * upf;
        const int cb = GENW ? ur % ncb : 0;
        const int hb0 = 2 * ROWS * (GENW ? ur / ncb : ur) - g.pt;    // image row of patch row 0
        const int base = ((n * CB * g.Hb + hb0) * g.Wb + (GENW ? 2 * DC_W * cb : 0)) * (U8 ? 1 : 4);
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int hb = hb0 + ld_y[k];
                bool ok = hb >= 0 && hb < g.Hb;
                if (GENW) {
                    const int wb = 2 * DC_W * cb + (ld_lds[k] - ld_y[k] * DC_RW - DC_X0);   // image column
                    ok = ok && wb >= 0 && wb < g.Wb;
                }
                st[c * NLD + k] = __builtin_amdgcn_raw_buffer_load_b128(
                    rb, ok ? base + ld_off[k] : ED_OOB, c * HWb * (U8 ? 1 : 4), 0);
            }
    };
    auto to_lds = [&](const intx4 (&st)[CB * NLD]) {
#pragma unroll
        for (int c = 1; c < CB; ++c)
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                if (ld_y[k] >= 0) *reinterpret_cast<intx4*>(bl + c * IH * DC_RW + ld_lds[k]) = st[c * NLD + k];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (ld_y[k] < 0) continue;
            if (U8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned v = (unsigned)st[k][j];
                    floatx4e f = {(float)(v & 255u) / 255.f, (float)((v >> 8) & 255u) / 255.f,
                                  (float)((v >> 16) & 255u) / 255.f, (float)(v >> 24) / 255.f};
                    *reinterpret_cast<floatx4e*>(bl + ld_lds[k] + 4 * j) = f;
                }
            } else {
                *reinterpret_cast<intx4*>(bl + ld_lds[k]) = st[k];
            }
        }
    };

    // A operand: weights of output channel li for taps (2t + kk); CB = 2: for channel kk, tap t
    constexpr int NT = CB == 1 ? 13 : 25;
    float wv_[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (CB == 1) {
            const int tap = 2 * t + kk;
            wv_[t] = (tap < 25 && li < g.Cs) ? w[li * 25 + tap] : 0.f;
        } else {
            wv_[t] = (li < g.Cs) ? w[(li * CB + kk) * 25 + t] : 0.f;
        }
    }
    // accumulator register e of this lane = channel (e&3) + 8*(e>>2) + 4*kk, pixel li
    float bz[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ch = (e & 3) + 8 * (e >> 2) + 4 * kk;
        bz[e] = (bias && ch < g.Cs) ? bias[ch] : 0.f;
    }
    const int kkA = kk, kkB = kk * (DC_RW - 4);
    const int a_col = DC_X0 - g.pl + 2 * li;
    const int st_lane = (4 * kk * PQ + li) * 4;          // byte offset of this lane's column

    intx4 stage[CB * NLD];
    int u = blockIdx.x;
    if (u < units) issue(u, stage);
#pragma unroll 1
    for (; u < units; u += gridDim.x) {
        // LDS operations of one wave execute in order: the previous unit's reads are done
        to_lds(stage);
        if (u + (int)gridDim.x < units) issue(u + gridDim.x, stage);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        const int n = u / upf;
        const int urr = u - n * upf;
        const int cbo = GENW ? urr % ncb : 0;                        // this unit's column block
        const int p0 = ROWS * (GENW ? urr / ncb : urr);
        const int opitch = GENW ? g.Ws : DC_W;                       // floats between output rows
        const float* aq = bl + a_col;
#pragma unroll 1
        for (int pr = 0; pr < ROWS; ++pr) {
            if (GENW && p0 + pr >= g.Hs) break;                      // (wave-uniform: rows below the map)
            floatx16 acc[2];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[qh][e] = bz[e];
            const float* ar = aq + (2 * pr) * DC_RW;
            const float* arA = ar + kkA;
            const float* arB = ar + kkB;
            if (CB == 2) {
                const float* ac = ar + kk * (IH * DC_RW);           // this lane half's channel
#pragma unroll
                for (int t = 0; t < 25; ++t) {
#pragma unroll
                    for (int qh = 0; qh < 2; ++qh)
                        acc[qh] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            wv_[t], ac[(t / 5) * DC_RW + t % 5 + 64 * qh], acc[qh], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int t = 0; t < 13; ++t) {
                // tap 24 + kk = 25 (t = 12, kk = 1) has a zero weight but must still read a FINITE
                // value: it takes the next column of the last patch row, which is always written
                const int t0 = ((2 * t) / 5) * DC_RW + (2 * t) % 5;
                const float* at = ((2 * t) % 5 == 4 && t != 12) ? arB : arA;
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
                    acc[qh] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[t], at[t0 + 64 * qh],
                                                                   acc[qh], 0, 0, 0);
            }
            // 16 x 2 dword stores: lanes 0-31 one 128-byte line of channel ch, lanes 32-63 of ch+4
            const int row_off = ((n * css * g.Hs + (p0 + pr)) * opitch + DC_W * cbo) * 4;
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                const bool cin = !GENW || DC_W * cbo + 32 * qh + li < g.Ws;   // this lane's column exists
                float d[16];
                if (MASK) {         // all 16 mask loads of the half row in flight together
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int chb = (e & 3) + 8 * (e >> 2);
                        const bool ok = chb + 4 * kk < g.Cs && cin;
                        d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rd, ok ? st_lane : ED_OOB, row_off + (chb * PQ + 32 * qh) * 4, 0));
                    }
                }
                float av[16];
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    float q[4] = {acc[qh][4 * e4], acc[qh][4 * e4 + 1], acc[qh][4 * e4 + 2], acc[qh][4 * e4 + 3]};
                    if (ACT == BN_ACT_LRELU) ed_lrelu4(q, slope);
                    av[4 * e4] = q[0]; av[4 * e4 + 1] = q[1]; av[4 * e4 + 2] = q[2]; av[4 * e4 + 3] = q[3];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int chb = (e & 3) + 8 * (e >> 2);           // + 4*kk in st_lane
                    const int so = row_off + (chb * PQ + 32 * qh) * 4;
                    const bool ok = chb + 4 * kk < g.Cs && cin;
                    float v = av[e];
                    if (MASK) v *= d[e] > 0.f ? 1.f : slope;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ro,
                                                          ok ? st_lane : ED_OOB, so, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// third generation of the gather-down edge kernel (float frames, forward): ALL input of a launch
// is requested in its first microsecond.
//
// What the per-wave s_memrealtime trace of the first generation showed (tools/lab/e0_lab.hip,
// 256 frames): the store stream itself is not the limit -- a pure 134 MB write-through stream with
// this kernel's address pattern runs at 6.5 TB/s, the same as a linear fill -- but the READS are:
// a wave's 3.5 KB patch took 1.9 us on the idle chip, 5.8 us (median) once the other waves'
// patches were queued and up to 25 us behind the saturated store queues, and every wave sat idle
// for that long, twice (2 units per wave).  The input is only 10 % of the bytes but it was on the
// critical path of every unit.
//
// Here a workgroup (4 waves) owns a STRIP of 16 output rows of one frame: its 35 input rows
// (18.6 KB) are copied into LDS by 16-byte LDS-DMA (`buffer_load ... lds`, no registers), all of
// them issued before anything else, zero borders included (out-of-range groups and rows read 0).
// With 4 workgroups per CU the whole launch's input (16.8 MB + 9 % halo) is in flight within the
// first microsecond, while HBM is otherwise idle; after that the memory system carries only the
// output stream.  Wave w multiplies strip rows w, w+4, w+8, w+12 (the first ones need only the
// first two DMA rounds, which are waited for separately); MFMA roles, LDS transposition slab,
// store instructions (8 channels x 128 B, write-through) and summation order are the first
// generation's: outputs are bit-identical to k_down_c1.
// ---------------------------------------------------------------------------------------------
#define DW_SROWS 16                          // output rows per strip
#define DW_WAVES 4
#define DW_IH (2 * DW_SROWS + 3)             // 35 input rows
#define DW_NG (DW_IH * DC_C4)                // 16-byte groups of the strip image (34 per row)
#define DW_NDMA ((DW_NG + 64 * DW_WAVES - 1) / (64 * DW_WAVES))      // DMA instructions per wave (5)
#define DW_IMG (DW_NDMA * DW_WAVES * 64 * 4) // floats of the image incl. the tail of the last round
#define DW_LDS ((DW_IMG + DW_WAVES * 32 * DC_HTS) * 4)

// s_barrier WITHOUT the memory fence of __syncthreads() (which drains vmcnt: the staged waits
// below would wait for every DMA round).  LDS-DMA data are visible to the other waves once the
// issuing wave's vmcnt covers them and a barrier has passed.
__device__ __forceinline__ void ed_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int ACT>
__global__ __launch_bounds__(64 * DW_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_down_c1w(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, BnGeom g, float slope, int units) {
    extern __shared__ __attribute__((aligned(16))) float wsm_[];
    float* img = wsm_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tw = wsm_ + DW_IMG + wv * (32 * DC_HTS);
    const int li = lane & 31, kk = lane >> 5;
    const int spf = g.Hs / DW_SROWS;                 // strips per frame
    const int HWb = g.Hb * g.Wb, PQ = g.Hs * g.Ws;
#ifdef E0_TRACE
    unsigned long long* trc = e0_trace + (size_t)(blockIdx.x * DW_WAVES + wv) * 8;
#undef E0_MARK
#define E0_MARK(slot) do { if (lane == 0) trc[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
    E0_MARK(0);
#endif

    // the image is addressed from `pt` rows above its start: patch row y of strip (n, s) is image
    // row 2 * 16 s - pt + y, and the scalar offset of a strip stays non-negative
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(big - g.pt * g.Wb), 0, (int)(((size_t)g.N * HWb + g.pt * g.Wb) * 4), 0x00020000);

    // B operand first (it is needed with the first rows): weights of output channel li, taps 2t + kk
    float wv_[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) {
        const int tap = 2 * t + kk;
        wv_[t] = (tap < 25 && li < g.Cs) ? w[li * 25 + tap] : 0.f;
    }
    const float bz = (bias && li < g.Cs) ? bias[li] : 0.f;

    // DMA round k of wave wv: groups e = 64 (wv + 4k) + lane -> (patch row y, 16-byte column c);
    // columns 0 and 33 are the zero borders (out-of-range source)
    int dvo[DW_NDMA], dy[DW_NDMA];
#pragma unroll
    for (int k = 0; k < DW_NDMA; ++k) {
        const int e = 64 * (wv + DW_WAVES * k) + lane;
        const int y = e / DC_C4, c = e - y * DC_C4;
        const bool ok = e < DW_NG && c >= 1 && c <= DC_W / 2;
        dy[k] = ok ? y : -0x10000;                    // fails the row test below
        dvo[k] = (y * g.Wb + 4 * (c - 1)) * 4;
    }
    auto issue_dma = [&](int u) __attribute__((always_inline)) {
        const int n = u / spf;
        const int hb0 = 2 * DW_SROWS * (u - n * spf) - g.pt;         // image row of patch row 0
        const int soff = ((n * g.Hb + hb0 + g.pt) * g.Wb) * 4;
#pragma unroll
        for (int k = 0; k < DW_NDMA; ++k) {
            const int hb = hb0 + dy[k];
            const bool ok = hb >= 0 && hb < g.Hb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, img + 4 * 64 * (wv + DW_WAVES * k), 16,
                                                     ok ? dvo[k] : ED_OOB, soff, 0, 0);
        }
    };

    const int kkA = kk, kkB = kk * (DC_RW - 4);
    const int a_col = DC_X0 - g.pl + 2 * li;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffffc, 0x00020000);

    int u = blockIdx.x;
    if (u < units) issue_dma(u);
#pragma unroll 1
    for (; u < units; u += gridDim.x) {
        const int n = u / spf;
        const int p0 = DW_SROWS * (u - n * spf);
        // Rows 0 .. 3 of the strip need patch rows 0 .. 10 = DMA rounds 0 and 1 (15 rows).  Only
        // loads are in flight here (they return in order), so the count is exact.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DW_NDMA - 2) : "memory");
        ed_barrier();
        if (u == (int)blockIdx.x) E0_MARK(1);
#pragma unroll 1
        for (int j = 0; j < DW_SROWS / DW_WAVES; ++j) {
            const int r = wv + DW_WAVES * j;                 // strip row of this wave
            if (j == 1) {
                // the rest of the image, before the first row that needs it.  (The stores of row 0
                // are in flight too and may retire out of order with loads: wait for everything.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ed_barrier();
            }
            floatx16 acc[2];
#pragma unroll
            for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[qh][e] = bz;
            const float* ar = img + (2 * r) * DC_RW + a_col;
            const float* arA = ar + kkA;
            const float* arB = ar + kkB;
#pragma unroll
            for (int t = 0; t < 13; ++t) {
                // (tap 25 = t 12, kk 1: zero weight, reads the next column of the last patch row)
                const int t0 = ((2 * t) / 5) * DC_RW + (2 * t) % 5;
                const float* at = ((2 * t) % 5 == 4 && t != 12) ? arB : arA;
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
                    acc[qh] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[t0 + 64 * qh], wv_[t], acc[qh], 0, 0, 0);
            }
            if (u == (int)blockIdx.x && j == 0) E0_MARK(2);
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    floatx4e v = {acc[qh][4 * grp], acc[qh][4 * grp + 1], acc[qh][4 * grp + 2],
                                  acc[qh][4 * grp + 3]};
                    if (ACT == BN_ACT_LRELU) {
                        float q[4] = {v.x, v.y, v.z, v.w};
                        ed_lrelu4(q, slope);
                        v = (floatx4e){q[0], q[1], q[2], q[3]};
                    }
                    *reinterpret_cast<floatx4e*>(tw + li * DC_HTS + 8 * grp + 4 * kk) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // 8 lanes x 16 B = one 128-byte half row; 8 channels per store instruction
                const size_t row0 = ((size_t)n * g.Cs * g.Hs + (p0 + r)) * DC_W + 32 * qh + 4 * (lane & 7);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 8 * i + (lane >> 3);
                    const floatx4e v = *reinterpret_cast<const floatx4e*>(tw + ch * DC_HTS + 4 * (lane & 7));
                    if (ch < g.Cs)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4e, v), ro,
                                                               E0_OUT_OFFSET((int)((row0 + (size_t)ch * PQ) * 4)), 0, DC_ST_AUX);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (u == (int)blockIdx.x && j == 0) E0_MARK(3);
        }
        if (u == (int)blockIdx.x) E0_MARK(4);
        if (u + (int)gridDim.x < units) {
            // more strips than resident workgroups: the image is reused once every wave has
            // finished reading it (all stores retired first: the next wait counts loads only)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ed_barrier();
            issue_dma(u + gridDim.x);
        }
    }
    E0_MARK(5);
#ifdef E0_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    E0_MARK(6);
#endif
}

// ---------------------------------------------------------------------------------------------
// fourth generation (float frames, forward, 32 output channels): the third generation's workgroup
// strips + swapped MFMA roles + a software pipeline inside every wave.
//
// What the s_memrealtime traces and two ablations of the earlier generations showed
// (tools/lab/e0_lab.hip, buffers rotated through > 256 MB so that the Infinity Cache serves neither
// stream): with every store redirected into a 1 MB window -- no HBM write stream at all -- the first
// generation still took 30.4 us of its 31.6: the kernel was never paced by the memory system (a
// skeleton that only loads the patches and issues the same stores runs in 27.4 us).  It was paced by
// ISSUE on the SIMDs: a row is 26 MFMAs (1664 cycles) followed by an epilogue (LeakyReLU, LDS
// transposition with four LDS round trips, 8 stores) that took 2.2 us of wall time per row, because
// while ONE wave of a SIMD streams MFMAs the other three cannot issue a single vector-ALU or
// vector-memory instruction (issue rule 2) -- the epilogues of four waves per SIMD and
// their MFMA bursts excluded each other; and the first patch of a wave arrived 7 us (median) after
// the launch, 2.5 us of that behind the 13 gathered weight loads of 16 waves in the CU's one
// vector-memory pipe.
//
// Here: (1) weights, bias and all 35 input rows of a strip are requested by LDS-DMA in the first
// instructions of the kernel; (2) MFMA rows = output channels (A = weights, from LDS into 13
// registers), columns = 32 pixels, so an accumulator register is one channel x 32 adjacent pixels
// and goes to HBM as it is -- one dword store instruction = two full 128-byte lines, no LDS
// transposition, no LDS round trip; (3) the epilogue of a half row (LeakyReLU + 16 stores) is issued from
// INSIDE the MFMA stream of the next half row of the same wave (LDS reads and stores cost nothing
// in the shadow of an MFMA; the vector-ALU instructions cost what they would cost anyway), two
// accumulators alternating -- the first stores leave 13 MFMAs after the data arrived; (4) strips of
// 8 output rows (19 patch rows, 10 KB), one per workgroup, 2048 workgroups for 256 frames of which
// 1024 are resident: the dispatcher hands out the second half as the first finishes, which evens
// out the tail that a static assignment showed (the slowest 10 % of the waves finished 4 us after
// the rest).  Summation order = the earlier generations': bit-identical outputs.
// ---------------------------------------------------------------------------------------------
#ifndef DP_ST_AUX
#define DP_ST_AUX 16                         // cache policy of the dword output stores: sc1 (write-through),
                                             // 27.3 / 27.9 us against 29.9 / 31.4 plain (0) behind clean / memset L2s
#endif
// STRIP = output rows per strip (16: 35 patch rows in 5 DMA rounds, 4 rows per wave; 8: 19 patch rows in 3
// rounds, 2 rows per wave)
template <int STRIP> struct DpGeom {
    static constexpr int IH = 2 * STRIP + 3;                    // patch rows
    static constexpr int NG = IH * DC_C4;                    // 16-byte groups of the strip image (34 per row)
    static constexpr int NDMA = (NG + 64 * DW_WAVES - 1) / (64 * DW_WAVES);    // DMA rounds
    static constexpr int IMG = NDMA * DW_WAVES * 64 * 4;     // floats of the image incl. the last round's tail
    // (an LDS-DMA instruction writes all 64 lanes x 16 B, zeros for out-of-range lanes: the four
    // weight instructions cover 1024 floats, the bias instruction 256)
    static constexpr int WL = IMG;                           // weights [32][25] behind the image
    static constexpr int BL = IMG + 1024;                    // bias [32]
    static constexpr int LDS = (IMG + 1024 + 256) * 4;
    static constexpr int ROWS = STRIP / DW_WAVES;               // rows per wave
};

template <int ACT, int STRIP>
__device__ __forceinline__ void dp_body(
    const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const BnGeom& g, const float slope, const int units) {
    typedef DpGeom<STRIP> G;
    extern __shared__ __attribute__((aligned(16))) float wsm_[];
    float* img = wsm_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int spf = g.Hs / STRIP;
    const int HWb = g.Hb * g.Wb, PQ = g.Hs * g.Ws;
#ifdef E0_TRACE
    unsigned long long* trc = e0_trace + (size_t)(blockIdx.x * DW_WAVES + wv) * 8;
#undef E0_MARK
#define E0_MARK(slot) do { if (lane == 0) trc[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
    E0_MARK(0);
#endif
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(big - g.pt * g.Wb), 0, (int)(((size_t)g.N * HWb + g.pt * g.Wb) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 800 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbi = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(bias ? bias : w), 0, bias ? 32 * 4 : 0, 0x00020000);          // no bias: reads 0
    const int css = bn_cs_stride(g);         // output: a window of Cs channels in frames of css
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
        (void*)out, 0, (int)((((size_t)g.N - 1) * css + g.Cs) * PQ * 4), 0x00020000);

    // first instructions: weights (wave wv copies groups 64 wv ..), bias (wave 3's free lanes), image
    {
        const int e = 64 * wv + lane;                               // 16-byte group of the weights
        // (plain ints: an argument of this builtin that depends on a template parameter makes this
        // hipcc drop the host-side stub of the kernel without a diagnostic)
        int wl_off = G::WL, bl_off = G::BL;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, wsm_ + wl_off + 4 * 64 * wv, 16,
                                                 e < 200 ? 16 * e : ED_OOB, 0, 0, 0);
        if (wv == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbi, wsm_ + bl_off, 16, lane < 8 ? 16 * lane : ED_OOB,
                                                     0, 0, 0);
    }
    int dvo[G::NDMA], dy[G::NDMA];
#pragma unroll
    for (int k = 0; k < G::NDMA; ++k) {
        const int e = 64 * (wv + DW_WAVES * k) + lane;
        const int y = e / DC_C4, c = e - y * DC_C4;
        const bool ok = e < G::NG && c >= 1 && c <= DC_W / 2;
        dy[k] = ok ? y : -0x10000;
        dvo[k] = (y * g.Wb + 4 * (c - 1)) * 4;
    }
    // rounds [k0, k1) of strip u; the row-validity masks of all rounds are computed once per strip
    auto issue_dma = [&](const int u, const int k0, const int k1) __attribute__((always_inline)) {
        const int n = u / spf;
        const int hb0 = 2 * STRIP * (u - n * spf) - g.pt;
        const int soff = ((n * g.Hb + hb0 + g.pt) * g.Wb) * 4;
#pragma unroll
        for (int k = 0; k < G::NDMA; ++k) {
            if (k < k0 || k >= k1) continue;
            const int hb = hb0 + dy[k];
            const bool ok = hb >= 0 && hb < g.Hb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, img + 4 * 64 * (wv + DW_WAVES * k), 16,
                                                     ok ? dvo[k] : ED_OOB, soff, 0, 0);
        }
    };
    int u = blockIdx.x;
    if (u < units) issue_dma(u, 0, G::NDMA);

    const int kkA = kk, kkB = kk * (DC_RW - 4);
    const int a_col = DC_X0 - g.pl + 2 * li;
    const int st_lane = (4 * kk * PQ + li) * 4;          // byte offset of this lane's column

    // weights, bias and the whole strip image (10 KB for 8-row strips) before the first MFMA: the
    // other resident workgroups of the CU are multiplying meanwhile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ed_barrier();
    E0_MARK(1);
    // A operand: weights of output channel li for taps (2t + kk); bias per accumulator register
    float wv_[13];
#pragma unroll
    for (int t = 0; t < 13; ++t) {
        const int tap = 2 * t + kk;
        wv_[t] = wsm_[G::WL + li * 25 + (tap < 25 ? tap : 24)];
        if (tap >= 25) wv_[t] = 0.f;
    }
    float bz[16];
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
        const floatx4e b4 = *reinterpret_cast<const floatx4e*>(wsm_ + G::BL + 8 * e4 + 4 * kk);
        bz[4 * e4] = b4.x; bz[4 * e4 + 1] = b4.y; bz[4 * e4 + 2] = b4.z; bz[4 * e4 + 3] = b4.w;
    }

    // one HALF row (32 pixels x 32 channels): 13 MFMAs on one accumulator.  With PREV the epilogue of
    // the previous half row (accumulator `pv`, byte offset `poff` of its first pixel in channel 0) is
    // issued from inside the stream: 4 registers -> LeakyReLU (6 vector-ALU instructions) and 4
    // dword stores behind every third MFMA.
    auto half = [&](floatx16& cur, const floatx16& pv, const int r, const int qh, const bool PREV,
                    const int poff) __attribute__((always_inline)) {
        const float* ar = img + (2 * r) * DC_RW + a_col + 64 * qh;
        const float* arA = ar + kkA;
        const float* arB = ar + kkB;
        float bv[13];
        auto rd = [&](int t) __attribute__((always_inline)) {
            // (tap 25 = t 12, kk 1: zero weight, reads the next column of the last patch row)
            const int t0 = ((2 * t) / 5) * DC_RW + (2 * t) % 5;
            const float* at = ((2 * t) % 5 == 4 && t != 12) ? arB : arA;
            bv[t] = at[t0];
        };
        rd(0); rd(1);
#pragma unroll
        for (int t = 0; t < 13; ++t) {
            if (t + 2 < 13) rd(t + 2);
            if (t == 0) {
                floatx16 init;
#pragma unroll
                for (int e = 0; e < 16; ++e) init[e] = bz[e];
                cur = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[t], bv[t], init, 0, 0, 0);
            } else {
                cur = __builtin_amdgcn_mfma_f32_32x32x2f32(wv_[t], bv[t], cur, 0, 0, 0);
            }
            if (PREV && (t % 3) == 0 && t / 3 < 4) {
                const int e4 = t / 3;
                float q[4] = {pv[4 * e4], pv[4 * e4 + 1], pv[4 * e4 + 2], pv[4 * e4 + 3]};
                if (ACT == BN_ACT_LRELU) ed_lrelu4(q, slope);
#pragma unroll
                for (int i = 0; i < 4; ++i)       // register e = 4 e4 + i: channel (e & 3) + 8 (e >> 2) (+ 4 kk)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i]), ro, st_lane,
                                                          E0_OUT_OFFSET(poff + (i + 8 * e4) * PQ * 4), DP_ST_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto flush = [&](const floatx16& pv, const int poff) __attribute__((always_inline)) {
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            float q[4] = {pv[4 * e4], pv[4 * e4 + 1], pv[4 * e4 + 2], pv[4 * e4 + 3]};
            if (ACT == BN_ACT_LRELU) ed_lrelu4(q, slope);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, q[i]), ro, st_lane,
                                                      E0_OUT_OFFSET(poff + (i + 8 * e4) * PQ * 4), DP_ST_AUX);
        }
    };

#pragma unroll 1
    for (; u < units; u += gridDim.x) {
        const int n = u / spf;
        const int p0 = STRIP * (u - n * spf);
        // byte offset of (strip row j of this wave, half qh) in channel 0
#define DP_OFF(j, qh) (((n * css * g.Hs + p0 + wv + DW_WAVES * (j)) * DC_W + 32 * (qh)) * 4)
        floatx16 a0, a1;
        half(a0, a0, wv, 0, false, 0);                                  // (no previous half row yet)
        if (u == (int)blockIdx.x) E0_MARK(2);
        half(a1, a0, wv, 1, true, DP_OFF(0, 0));                        // + 16 stores
#pragma unroll
        for (int j = 1; j < G::ROWS; ++j) {
            half(a0, a1, wv + DW_WAVES * j, 0, true, DP_OFF(j - 1, 1));
            if (u == (int)blockIdx.x && j == 1) E0_MARK(3);
            half(a1, a0, wv + DW_WAVES * j, 1, true, DP_OFF(j, 0));
        }
        flush(a1, DP_OFF(G::ROWS - 1, 1));
#undef DP_OFF
        if (u == (int)blockIdx.x) E0_MARK(4);
        if (u + (int)gridDim.x < units) {
            // more strips than resident workgroups: the image is reused once every wave is done with it
            ed_barrier();
            issue_dma(u + gridDim.x, 0, G::NDMA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ed_barrier();
        }
    }
    E0_MARK(5);
#ifdef E0_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    E0_MARK(6);
#endif
}

// (Plain kernels around the template body: this hipcc drops the host-side stub of a __global__
// TEMPLATE without a diagnostic when an argument of the LDS-DMA builtin in its body depends on a
// template parameter.)
#define DP_KERNEL(name, ACT, STRIP)                                                                    \
    __global__ __launch_bounds__(64 * DW_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void name(  \
        const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,    \
        float* __restrict__ out, BnGeom g, float slope, int units) {                                   \
        dp_body<ACT, STRIP>(big, w, bias, out, g, slope, units);                                       \
    }
DP_KERNEL(k_down_c1p_lrelu_s16, BN_ACT_LRELU, 16)
DP_KERNEL(k_down_c1p_none_s16, BN_ACT_NONE, 16)
DP_KERNEL(k_down_c1p_lrelu_s8, BN_ACT_LRELU, 8)
DP_KERNEL(k_down_c1p_none_s8, BN_ACT_NONE, 8)
#undef DP_KERNEL

template <int ACT, int STRIP>
static int launch_down_c1p(const float* big, const float* w, const float* bias, float* out,
                           const BnGeom& g, float slope, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    typedef void (*kernel_t)(const float*, const float*, const float*, float*, BnGeom, float, int);
    const kernel_t kernel = STRIP == 16
        ? (ACT == BN_ACT_LRELU ? k_down_c1p_lrelu_s16 : k_down_c1p_none_s16)
        : (ACT == BN_ACT_LRELU ? k_down_c1p_lrelu_s8 : k_down_c1p_none_s8);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DpGeom<STRIP>::LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int units = g.N * (g.Hs / STRIP);
    int grid = units;                                  // one strip per workgroup; the dispatcher balances
    if (const char* e = bn_tune_env("BN_E0_WGRID")) grid = atoi(e);     // (tuning build only)
    if (grid > units) grid = units;
    hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(64 * DW_WAVES), DpGeom<STRIP>::LDS, st, e0, e1, 0,
                          big, w, bias, out, g, slope, units);
    BN_LAUNCH_CHECK();
    return 0;
}

static bool down_c1w_ok(const BnGeom& g) {
    return g.Cb == 1 && (g.Hs % DW_SROWS) == 0 && g.pt <= 2 && DC_ST_AUX >= 0;
}
static bool down_c1p_ok(const BnGeom& g, int strip) {
    return g.Cb == 1 && g.Cs == 32 && (g.Hs % strip) == 0 && g.pt <= 2;
}

template <int ACT>
static int launch_down_c1w(const float* big, const float* w, const float* bias, float* out,
                           const BnGeom& g, float slope, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_down_c1w<ACT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int units = g.N * (g.Hs / DW_SROWS);
    int grid = 256 * 4;                               // 4 workgroups of 38 KB per CU
    if (const char* e = bn_tune_env("BN_E0_WGRID")) grid = atoi(e);     // (tuning build only)
    if (grid > units) grid = units;
    hipExtLaunchKernelGGL((k_down_c1w<ACT>), dim3(grid), dim3(64 * DW_WAVES), DW_LDS, st, e0, e1, 0,
                          big, w, bias, out, g, slope, units);
    BN_LAUNCH_CHECK();
    return 0;
}

BnFastPlan bn_edge_down_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_down_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || g.stride != 2 || (g.Cb != 1 && g.Cb != 2)) return p;
    if (g.pl < 0 || g.pl > DC_X0 || g.pt < 0) return p;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return p;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return p;
    if (g.Cs > 32) return p;
    if (g.Ws != DC_W || (g.Hs % DC_ROWS) != 0 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) {
        // round 4: single-channel frames of any other size (widths in multiples of 4) on the first-generation
        // kernel in blocks of 64 columns -- no zero-padded / tiled copies (variant 9: no uint8 input, no
        // channel window)
        static int off = -1;                          // BN_DOWN_C1G=0: off
        if (off < 0) { const char* e = bn_tune_env("BN_DOWN_C1G"); off = (e && e[0] == '0') ? 1 : 0; }
        if (off || (g.Ws & 3) || (g.Wb & 3) || (g.CsS > 0 && g.CsS != g.Cs)) return p;
        p.supported = true;
        p.variant = 9;
        p.kernel_name = g.Cb == 2 ? "k_down_c1s<.., 2, 2, gen>" : "k_down_c1<gen>";
        return p;
    }
    p.supported = true;
    p.kernel_name = "k_down_c1";
    return p;
}

// grid = waves; every CU should get the same number of units in the fewest rounds
static int down_c1_grid(int units) {
#ifdef BN_TUNING
    static int env_grid = -1;                       // BN_E0_GRID=<waves>
    if (env_grid < 0) { const char* e = bn_tune_env("BN_E0_GRID"); env_grid = e ? atoi(e) : 0; }
    if (env_grid > 0) return env_grid < units ? env_grid : units;
#endif
    const int n_cu = 256;
    const int per_cu = (units + n_cu - 1) / n_cu;
    const int rounds = (per_cu + DC_MAX_WAVES_PER_CU - 1) / DC_MAX_WAVES_PER_CU;
    const int waves = (per_cu + rounds - 1) / rounds;
    const int grid = n_cu * waves;
    return grid < units ? grid : units;
}

template <int ACT, bool MASK, bool U8, int ROWS, int CB = 1>
static int launch_down_c1s(const void* big, const float* w, const float* bias, float* out,
                           const float* dact_src, const BnGeom& g, float slope, hipStream_t st,
                           hipEvent_t e0, hipEvent_t e1) {
    const int units = g.N * (g.Hs / ROWS);
    int grid = 256 * DC_MAX_WAVES_PER_CU;
    if (const char* e = bn_tune_env("BN_E0_SGRID")) grid = atoi(e);     // (tuning build only)
    if (grid > units) grid = units;
    hipExtLaunchKernelGGL((k_down_c1s<ACT, MASK, U8, ROWS, CB>), dim3(grid), dim3(64), 0, st, e0, e1,
                          0, big, w, bias, out, dact_src, g, slope, units);
    BN_LAUNCH_CHECK();
    return 0;
}

// name of the kernel bn_launch_edge_down dispatches to (profiling scopes, tests)
const char* bn_edge_down_kernel_name(const BnGeom& g, int act, bool has_dact, bool u8) {
    const bool lrelu = act == BN_ACT_LRELU;
    if (g.Ws != DC_W || (g.Hs % DC_ROWS) != 0 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) {
        if (g.Cb == 2) return has_dact ? "k_down_c1s<.., 2, 2, gen, mask>" : "k_down_c1s<.., 2, 2, gen>";
        return has_dact ? "k_down_c1<gen, mask>" : "k_down_c1<gen>";
    }
    if (g.Cb == 2)
        return has_dact ? "k_down_c1s<0, true, false, 2, 2>"
                        : (lrelu ? "k_down_c1s<1, false, false, 2, 2>" : "k_down_c1s<0, false, false, 2, 2>");
    if (u8) {
        if ((g.Hs % 4) == 0)
            return lrelu ? "k_down_c1s<1, false, true, 4, 1>" : "k_down_c1s<0, false, true, 4, 1>";
        return lrelu ? "k_down_c1s<1, false, true, 2, 1>" : "k_down_c1s<0, false, true, 2, 1>";
    }
    if (has_dact) return "k_down_c1s<0, true, false, 2, 1>";
    if (DC_VARIANT == 5 && down_c1p_ok(g, 8)) return lrelu ? "k_down_c1p_lrelu_s8" : "k_down_c1p_none_s8";
    if (DC_VARIANT >= 4 && down_c1p_ok(g, 16)) return lrelu ? "k_down_c1p_lrelu_s16" : "k_down_c1p_none_s16";
    if (DC_VARIANT >= 3 && down_c1w_ok(g)) return lrelu ? "k_down_c1w<1>" : "k_down_c1w<0>";
    return lrelu ? "k_down_c1<1, false>" : "k_down_c1<0, false>";
}

// u8 != nullptr: the frames are uint8 (converted in flight, value / 255); else `big` is float.
int bn_launch_edge_down(const float* big, const float* w, const float* bias, float* out,
                        const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                        hipStream_t st, const unsigned char* u8) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bn_prof_take_dispatch_events(&e0, &e1);   // stay null unless bench.py's hook is armed
    if (g.Ws != DC_W || (g.Hs % DC_ROWS) != 0 || g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) {
        // any other map of a single-channel frame: blocks of 64 columns (k_down_c1<.., GENW>)
        if (u8 || (g.Ws & 3) || (g.Wb & 3) || (g.CsS > 0 && g.CsS != g.Cs)) return BN_E_SHAPE;
        const int units = g.N * ((g.Hs + DC_ROWS - 1) / DC_ROWS) * ((g.Ws + DC_W - 1) / DC_W);
        if (g.Cb == 2) {     // two-channel frames: the swapped-role kernel (2-row units) in column blocks
            int grid2 = 256 * DC_MAX_WAVES_PER_CU;
            if (grid2 > units) grid2 = units;
#define C1S_GEN(A, M)                                                                                     \
    hipExtLaunchKernelGGL((k_down_c1s<A, M, false, 2, 2, true>), dim3(grid2), dim3(64), 0, st, e0, e1, 0,  \
                          (const void*)big, w, bias, out, dact_src, g, slope, units)
            if (act == BN_ACT_LRELU && !dact_src) C1S_GEN(BN_ACT_LRELU, false);
            else if (!dact_src) C1S_GEN(BN_ACT_NONE, false);
            else C1S_GEN(BN_ACT_NONE, true);
#undef C1S_GEN
            BN_LAUNCH_CHECK();
            return 0;
        }
        const dim3 grid(down_c1_grid(units));
        if (act == BN_ACT_LRELU && !dact_src) {
            hipExtLaunchKernelGGL((k_down_c1<BN_ACT_LRELU, false, true>), grid, dim3(64), 0, st, e0, e1, 0,
                                  big, w, bias, out, dact_src, g, slope, units);
        } else if (act == BN_ACT_NONE && !dact_src) {
            hipExtLaunchKernelGGL((k_down_c1<BN_ACT_NONE, false, true>), grid, dim3(64), 0, st, e0, e1, 0,
                                  big, w, bias, out, dact_src, g, slope, units);
        } else {
            hipExtLaunchKernelGGL((k_down_c1<BN_ACT_NONE, true, true>), grid, dim3(64), 0, st, e0, e1, 0,
                                  big, w, bias, out, dact_src, g, slope, units);
        }
        BN_LAUNCH_CHECK();
        return 0;
    }
    // Product choice, measured INSIDE the training step (rocprofv3, 256 frames; the isolated
    // ranking differs): plain forward -> first generation (31.7-32.8 us vs 33.7-34.6 us);
    // data gradient with the LeakyReLU' mask -> second generation, 2-row units (48.2 vs 52.6 us).
    if (g.Cb == 2) {     // two-channel frames: swapped-role kernel, 2-row units
        if (u8) return BN_E_SHAPE;
        if (act == BN_ACT_LRELU && !dact_src)
            return launch_down_c1s<BN_ACT_LRELU, false, false, 2, 2>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        if (!dact_src)
            return launch_down_c1s<BN_ACT_NONE, false, false, 2, 2>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        return launch_down_c1s<BN_ACT_NONE, true, false, 2, 2>(big, w, bias, out, dact_src, g, slope, st, e0, e1);
    }
    int variant = dact_src ? 1 : DC_VARIANT;
    if (const char* e = bn_tune_env("BN_E0_V")) variant = atoi(e);   // (tuning build only)
    if (variant == 2 && (g.Hs % 4) != 0) variant = 1;
    if (u8) {
        if ((g.Hs % 4) == 0)       // uint8 frames: second generation only, 4-row units
            return act == BN_ACT_LRELU
                ? launch_down_c1s<BN_ACT_LRELU, false, true, 4>(u8, w, bias, out, nullptr, g, slope, st, e0, e1)
                : launch_down_c1s<BN_ACT_NONE, false, true, 4>(u8, w, bias, out, nullptr, g, slope, st, e0, e1);
        return act == BN_ACT_LRELU
            ? launch_down_c1s<BN_ACT_LRELU, false, true, 2>(u8, w, bias, out, nullptr, g, slope, st, e0, e1)
            : launch_down_c1s<BN_ACT_NONE, false, true, 2>(u8, w, bias, out, nullptr, g, slope, st, e0, e1);
    }
    if (variant == 5 && !down_c1p_ok(g, 8)) variant = 4;
    if ((variant == 4 || variant == 5) && !dact_src && down_c1p_ok(g, variant == 5 ? 8 : 16)) {
        if (variant == 5)
            return act == BN_ACT_LRELU
                ? launch_down_c1p<BN_ACT_LRELU, 8>(big, w, bias, out, g, slope, st, e0, e1)
                : launch_down_c1p<BN_ACT_NONE, 8>(big, w, bias, out, g, slope, st, e0, e1);
        return act == BN_ACT_LRELU
            ? launch_down_c1p<BN_ACT_LRELU, 16>(big, w, bias, out, g, slope, st, e0, e1)
            : launch_down_c1p<BN_ACT_NONE, 16>(big, w, bias, out, g, slope, st, e0, e1);
    }
    if (variant == 4 || variant == 5) variant = 3;
    // (a channel window in wider frames, BnGeom::CsS: only k_down_c1p and k_down_c1s address it)
    const bool windowed = g.CsS > 0 && g.CsS != g.Cs;
    if (windowed && variant != 1 && variant != 2) variant = 1;
    if (variant == 3 && !dact_src && down_c1w_ok(g))
        return act == BN_ACT_LRELU
            ? launch_down_c1w<BN_ACT_LRELU>(big, w, bias, out, g, slope, st, e0, e1)
            : launch_down_c1w<BN_ACT_NONE>(big, w, bias, out, g, slope, st, e0, e1);
    if (variant == 3) variant = dact_src ? 1 : 0;
    if (variant == 2) {
        if (act == BN_ACT_LRELU && !dact_src)
            return launch_down_c1s<BN_ACT_LRELU, false, false, 4>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        if (!dact_src)
            return launch_down_c1s<BN_ACT_NONE, false, false, 4>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        return launch_down_c1s<BN_ACT_NONE, true, false, 4>(big, w, bias, out, dact_src, g, slope, st, e0, e1);
    }
    if (variant == 1) {
        if (act == BN_ACT_LRELU && !dact_src)
            return launch_down_c1s<BN_ACT_LRELU, false, false, 2>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        if (!dact_src)
            return launch_down_c1s<BN_ACT_NONE, false, false, 2>(big, w, bias, out, nullptr, g, slope, st, e0, e1);
        return launch_down_c1s<BN_ACT_NONE, true, false, 2>(big, w, bias, out, dact_src, g, slope, st, e0, e1);
    }
    const int units = g.N * (g.Hs / DC_ROWS);
    const dim3 grid(down_c1_grid(units));
    if (act == BN_ACT_LRELU && !dact_src) {
        hipExtLaunchKernelGGL((k_down_c1<BN_ACT_LRELU, false>), grid, dim3(64), 0, st, e0, e1, 0,
                              big, w, bias, out, dact_src, g, slope, units);
    } else if (act == BN_ACT_NONE && !dact_src) {
        hipExtLaunchKernelGGL((k_down_c1<BN_ACT_NONE, false>), grid, dim3(64), 0, st, e0, e1, 0,
                              big, w, bias, out, dact_src, g, slope, units);
    } else {   // data gradient: no activation of its own, LeakyReLU' mask of the layer below
        hipExtLaunchKernelGGL((k_down_c1<BN_ACT_NONE, true>), grid, dim3(64), 0, st, e0, e1, 0,
                              big, w, bias, out, dact_src, g, slope, units);
    }
    BN_LAUNCH_CHECK();
    return 0;
}

// =============================================================================================
// gather-up with one big-side channel (dec.convT4 forward):
//   out[n,0,h,w] = act( b + sum_{c,r,s} small[n,c,p,q] * W[c][0][r][s] ),  2p+r = h+1, 2q+s = w+1
// Two phases per workgroup (8 small-image rows -> 16 x 128 output pixels):
//  1. a skinny GEMM on the matrix cores  T[tap][pos] = sum_c W[c][tap] * small[c][pos]
//     (rows = 25 taps, K = Cs channels, columns = positions incl. a one-pixel halo), small read
//     straight from HBM in wavefront rows, T kept in LDS;
//  2. every output pixel gathers its <= 9 contributions T[(r,s)][(p,q)] from LDS, adds the
//     bias, applies the activation and is stored as float2 (both column parities per lane).
// =============================================================================================
#define UC_TH 8
#define UC_W 64
#define UC_PW (UC_W + 2)                      // positions per row incl. halo
#define UC_NPOS ((UC_TH + 2) * UC_PW)         // 660
#define UC_NBLK ((UC_NPOS + 31) / 32)         // 21
#define UC_DP (UC_NBLK * 32)                  // 672: row stride of T in LDS
#define UC_LDS (25 * UC_DP * 4)

__global__ __launch_bounds__(ED_THREADS) void k_up_c1(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, BnGeom g, int act, float slope) {
    extern __shared__ __attribute__((aligned(16))) float dl[];     // T[25][UC_DP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int tiles_per_frame = g.Hs / UC_TH;
    const int n = blockIdx.x / tiles_per_frame;
    const int a0 = (blockIdx.x - n * tiles_per_frame) * UC_TH;
    const int HWs = g.Hs * g.Ws;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * HWs * 4), 0x00020000);

    // A operand: W[c = 2t+kk][tap = li]   (Cs <= 32 -> 16 steps)
    float av[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int c = 2 * t + kk;
        av[t] = (li < 25 && c < g.Cs) ? w[(c * g.Cb + blockIdx.y) * 25 + li] : 0.f;
    }

    for (int blk = wv; blk < UC_NBLK; blk += 4) {
        const int f = blk * 32 + li;
        const int py = f / UC_PW, px = f - py * UC_PW;
        const int p = a0 - 1 + py, q = px - 1;
        const bool ok = f < UC_NPOS && p >= 0 && p < g.Hs && q >= 0 && q < g.Ws;
        const int boff = ok ? ((n * g.Cs + kk) * HWs + p * g.Ws + q) * 4 : ED_OOB;
        float bv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t)
            bv[t] = ed_ld(rs, (ok && 2 * t + kk < g.Cs) ? boff + t * (2 * HWs * 4) : ED_OOB);
        floatx16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tap = (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (tap < 25) dl[tap * UC_DP + f] = acc[e];
        }
    }
    __syncthreads();

    // phase 2: thread -> column pair b (w = 2b, 2b+1), wave -> row h_l = 4k + wv
    const int b = lane;
    const float bs = bias ? bias[blockIdx.y] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int hl = 4 * k + wv;                 // 0..15
        const int h = 2 * a0 + hl;
        const int hh = hl + 1;                     // (h + pt) relative to row 2*a0, pt = 1
        float o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = (hh & 1) + 2 * u;
            if (r >= 5) continue;
            const int py = ((hh - r) >> 1) + 1;     // position row inside the tile (halo = 1)
            const float* row = dl + py * UC_PW;
            // w = 2b   -> ww = 2b+1: s in {1,3}, q = b - {0,1}   -> px = q + 1
            o0 += row[(r * 5 + 1) * UC_DP + b + 1] + row[(r * 5 + 3) * UC_DP + b];
            // w = 2b+1 -> ww = 2b+2: s in {0,2,4}, q = b + 1 - {0,1,2}
            o1 += (row[(r * 5 + 0) * UC_DP + b + 2] + row[(r * 5 + 2) * UC_DP + b + 1]) +
                  row[(r * 5 + 4) * UC_DP + b];
        }
        float2 v;
        v.x = bn_apply_act(o0 + bs, act, slope);
        v.y = bn_apply_act(o1 + bs, act, slope);
        *reinterpret_cast<float2*>(
            out + (((size_t)n * g.Cb + blockIdx.y) * g.Hb + h) * g.Wb + 2 * b) = v;
    }
}

// =============================================================================================
// gather-up with few big-side channels, second generation (dec.convT4 forward [+ loss epilogue]):
//   out[n,b,h,w] = act( bias[b] + sum_{c,r,s} small[n,c,p,q] * W[c][b][r][s] ),
//   2p + r = h + 1, 2q + s = w + 1
// 11 FLOP per byte and ONE output channel: there is no matrix shape to feed, so this generation
// is a register-resident VALU kernel that touches neither LDS nor the matrix cores:
//  * a workgroup is ONE wave, lane q = column q of the 64-wide small image; a unit is a strip of
//    R small-image rows of one frame (plus one halo row above and below, of which only the taps
//    that reach the strip's 2R output rows are evaluated: 7 % extra FMAs at R = 8);
//  * the wave's 2R x 128 output pixels live in 4R accumulator registers per lane (columns 2q and
//    2q+1 of every row) for the whole unit;
//  * per input channel the 25 weights are wave-uniform SGPR operands of v_fmac (s_load from the
//    scalar cache), the R + 2 input rows are full 256-byte wave rows (one buffer_load_dword per
//    row, the next channel's rows in flight while this one is multiplied), and the column
//    neighbours q-1 / q+1 come from DPP wave shifts (zero shifted in at the image border);
//  * epilogue: bias, activation, one 512-byte float2 store per output row.  LOSS variant
//    (training): the target (and mask) rows are read instead, the squared error is summed per
//    unit, and dL/dpre = 2 (xhat - target) mask xhat (1 - xhat) is written -- xhat itself never
//    goes to memory unless asked for (reference aes.py:330 + losses.py:56-59).
// =============================================================================================
#define UV_W 64
#ifndef UP_C1_VARIANT
#define UP_C1_VARIANT 1                  // 1: k_up_c1m (matrix cores) where its geometry holds (in the
                                         // training step 41.6 us against 51.0 for k_up_c1v on one box)
#endif

__device__ __forceinline__ float uv_shift_from_left(float v) {      // lane q <- lane q-1, 0 at q=0
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
        __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float uv_shift_from_right(float v) {     // lane q <- lane q+1, 0 at q=63
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(
        __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// GENW (round 4): any small map (dec.convT4 of frames other than 128 columns wide, no tiles / zero-padded
// copies) -- a unit is a strip of R input rows of ONE block of 62 input columns: the wave's 64 lanes hold columns
// 62 cb - 1 .. 62 cb + 62, so that every lane that stores (1..62) finds both neighbours in the wave and the
// DPP row shifts stay what they are; the outer two lanes only carry the neighbours (zeros at the frame's edges).
template <int R, bool LOSS, bool GENW = false>
__global__ __launch_bounds__(64) void k_up_c1v(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ target, const float* __restrict__ mask,
    float* __restrict__ dpre, float* __restrict__ partial, BnGeom g, int act, float slope,
    int units) {
    // (round 5: the fused pixel loss serves the column blocks too -- 2x192x160 / 1x192x192 frames ran
    // k_sqerr_frame_sums + k_sqerr_bwd + k_act_bwd behind this layer: 80-125 us per step)
    constexpr int NR = R + 2;                        // strip rows + halo above / below
    const int lane = threadIdx.x;
    const int strips = GENW ? (g.Hs + R - 1) / R : g.Hs / R;
    const int ncb = GENW ? (g.Ws + 61) / 62 : 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * g.Hs * g.Ws * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)g.Cs * g.Cb * 25 * 4), 0x00020000);

    // one channel's operands: the R + 2 input rows (lane = column) and, in lanes 0..24, the 25
    // weights of that channel (broadcast to SGPRs by v_readlane when the channel is multiplied:
    // no scalar-memory latency anywhere in the loop)
    static_assert((NR & 1) == 0, "input rows are kept as register pairs");
    struct Chan { floatx2p xx[NR / 2]; float wl; };          // rows (2k, 2k+1) = one register pair

#pragma unroll 1
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int cbk = GENW ? u % ncb : 0;            // column block
        const int us = GENW ? u / ncb : u;
        const int strip = us % strips;
        const int nb = us / strips;
        const int bch = nb % g.Cb, n = nb / g.Cb;
        const int p0 = strip * R;
        const int col = GENW ? 62 * cbk - 1 + lane : lane;     // this lane's input column

        floatx2p acc[2 * R];                           // .x = column 2q, .y = column 2q+1
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) acc[j] = (floatx2p){0.f, 0.f};

        // Addresses: the lane part (column, 4 * lane, plus the row's compile-time 256 * i in the
        // instruction's offset field) is per unit, the channel part is ONE scalar offset per
        // channel.  Rows above / below the image (first / last strip) use an out-of-range lane
        // offset: they read as zero without touching memory.
        // (the scalar offset is unsigned: it points at row p0-1, or at row p0 for the first strip,
        // whose rows then sit one row earlier in the lane offsets)
        const int sh = p0 > 0 ? 0 : -(UV_W * 4);
        const int vo_mid = lane * 4 + sh;
        const int vo_top = p0 > 0 ? lane * 4 : ED_OOB;
        const int vo_bot = p0 + R < g.Hs ? lane * 4 + sh + (NR - 1) * (UV_W * 4) : ED_OOB;
        const int vo_w = lane < 25 ? lane * 4 : ED_OOB;
        const int frame_row0 = GENW ? n * g.Cs * g.Hs * g.Ws * 4
                                    : (n * g.Cs * g.Hs + (p0 > 0 ? p0 - 1 : 0)) * (UV_W * 4);
        // GENW: strip row i is image row p0 - 1 + i of a map of g.Ws columns; rows / columns outside read 0.0f
        int vog[NR];
        if (GENW) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = p0 - 1 + i;
                vog[i] = (row >= 0 && row < g.Hs && col >= 0 && col < g.Ws) ? (row * g.Ws + col) * 4 : ED_OOB;
            }
        }
        auto load_chan = [&](int c, Chan& ch) {
            // channels past the last one (the 3-way unrolled loop overshoots): zero weights, and
            // the rows of the last channel again (finite whenever the frame is)
            const int cx = c < g.Cs ? c : g.Cs - 1;
            const int so = frame_row0 + cx * g.Hs * (GENW ? g.Ws * 4 : UV_W * 4);
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int vo = GENW ? vog[i]
                                    : (i == 0 ? vo_top : (i == NR - 1 ? vo_bot : vo_mid + i * (UV_W * 4)));
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
                if (i & 1) ch.xx[i >> 1].y = v; else ch.xx[i >> 1].x = v;
            }
            ch.wl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rw, c < g.Cs ? vo_w : ED_OOB, (cx * g.Cb + bch) * 100, 0));
        };
        auto fma_chan = [&](const Chan& ch) {
            float wk[25];
#pragma unroll
            for (int t = 0; t < 25; ++t)
                wk[t] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                    __builtin_bit_cast(int, ch.wl), t));
            // The two columns of an output row are one register pair, and taps (1,2) resp. (3,4)
            // of a kernel row multiply the SAME input value (column q resp. q-1): each is one
            // v_pk_fma_f32 with the weight pair as the scalar operand and the input half picked by
            // op_sel -- 3 instead of 5 vector instructions per (input row, kernel row); the kernel
            // is bound by vector issue (200 multiply-adds per channel and lane), not by HBM.
#pragma unroll
            for (int ip = 0; ip < NR / 2; ++ip) {
                const floatx2p xc2 = ch.xx[ip];
                floatx2p xm2, xp2;
                xm2.x = uv_shift_from_left(xc2.x);  xm2.y = uv_shift_from_left(xc2.y);     // small[.., q-1]
                xp2.x = uv_shift_from_right(xc2.x); xp2.y = uv_shift_from_right(xc2.y);    // small[.., q+1]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * ip + h;
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        const int j = 2 * i + r - 3;               // output row 2*p0 + j
                        if (j < 0 || j >= 2 * R) continue;         // (compile-time)
                        // column 2q: s = 1 -> q, s = 3 -> q-1; column 2q+1: s = 2 -> q, s = 4 -> q-1
                        const unsigned long long w12 =
                            ((unsigned long long)__builtin_bit_cast(unsigned, wk[r * 5 + 2]) << 32) |
                            __builtin_bit_cast(unsigned, wk[r * 5 + 1]);
                        const unsigned long long w34 =
                            ((unsigned long long)__builtin_bit_cast(unsigned, wk[r * 5 + 4]) << 32) |
                            __builtin_bit_cast(unsigned, wk[r * 5 + 3]);
                        if (h == 0) {
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "s"(w12), "v"(xc2));
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[j]) : "s"(w34), "v"(xm2));
                        } else {
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[j]) : "s"(w12), "v"(xc2));
                            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[j]) : "s"(w34), "v"(xm2));
                        }
                        // column 2q+1, s = 0 -> q+1
                        acc[j].y = fmaf(wk[r * 5 + 0], h ? xp2.y : xp2.x, acc[j].y);
                    }
                }
            }
        };

        // three register sets: channel c is multiplied while c+1 and c+2 are in flight
        Chan ca, cb, cc;
        load_chan(0, ca);
        load_chan(1, cb);
#pragma unroll 1
        for (int c = 0; c < g.Cs; c += 3) {
            load_chan(c + 2, cc);
            fma_chan(ca);
            __builtin_amdgcn_sched_barrier(0);     // keep each channel's 25 readlanes with its FMAs
            load_chan(c + 3, ca);
            fma_chan(cb);
            __builtin_amdgcn_sched_barrier(0);
            load_chan(c + 4, cb);
            fma_chan(cc);
            __builtin_amdgcn_sched_barrier(0);
        }

        const float bs = bias ? bias[bch] : 0.f;
        const size_t o0 = (((size_t)n * g.Cb + bch) * g.Hb + 2 * p0) * g.Wb + 2 * (GENW ? (col > 0 ? col : 0) : lane);
        const bool lane_out = !GENW || (lane >= 1 && lane <= 62 && col < g.Ws);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) {
            if (GENW && (!lane_out || 2 * p0 + j >= g.Hb)) continue;
            float2 v;
            v.x = acc[j].x + bs;
            v.y = acc[j].y + bs;
            if (act == BN_ACT_SIGMOID) {
                // rcp instead of the IEEE division of bn_apply_act: 1 ulp, a tenth of the code
                v.x = __builtin_amdgcn_rcpf(1.f + __expf(-v.x));
                v.y = __builtin_amdgcn_rcpf(1.f + __expf(-v.y));
            } else {
                v.x = bn_apply_act(v.x, act, slope);
                v.y = bn_apply_act(v.y, act, slope);
            }
            const size_t o = o0 + (size_t)j * g.Wb;
            if (out) *reinterpret_cast<float2*>(out + o) = v;
            if (LOSS) {
                const float2 t = *reinterpret_cast<const float2*>(target + o);
                float dx = v.x - t.x, dy = v.y - t.y;
                float mx = 1.f, my = 1.f;
                if (mask) {
                    const float2 m = *reinterpret_cast<const float2*>(mask + o);
                    mx = m.x; my = m.y;
                }
                sq += dx * dx * mx + dy * dy * my;
                float2 d;
                d.x = 2.f * dx * mx * bn_act_grad_from_output(v.x, act, slope);
                d.y = 2.f * dy * my * bn_act_grad_from_output(v.y, act, slope);
                *reinterpret_cast<float2*>(dpre + o) = d;
            }
        }
        if (LOSS) {
            // fixed-order butterfly over the 64 lanes, then one value per unit
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
            if (lane == 0) partial[u] = sq;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gather-up with <= 32 small-side channels on the matrix cores (dec.convT4 forward, optionally with
// the pixel loss): the 32 -> 1 channel reduction is a skinny GEMM
//     T[tap][pos] = sum_c W[c][tap] * small[c][pos]          (25 taps x 32 channels x positions)
// and the transposed convolution is then a sum of <= 9 SHIFTED entries of T per output pixel:
//     out[2P + a][2Q + b] = sum_{(r, dp) in R_a} sum_{(s, dq) in S_b} T[r][s][P + dp][Q + dq],
//     R_0 = S_0 = {(1, 0), (3, -1)},  R_1 = S_1 = {(0, +1), (2, 0), (4, -1)}        (pt = pl = 1).
// k_up_c1v does the whole thing on the vector ALU (200 multiply-adds per channel and lane: bound by
// vector issue at 35 TFLOP/s, 48 us); here the multiply-adds are MFMAs (14 us of matrix time) and
// the vector ALU only adds the 25 shifted entries (~30 instructions per input row of 64 columns).
//
// Workgroup = one frame x one output channel, wave = 8 input rows, 64 lanes = the 64 columns:
//  * MFMA 32x32x2, rows = taps, columns = 32 positions, reduction = channels (16 steps).  Two column
//    blocks per input row: A = even columns q = 2 li, B = odd columns q = 2 li + 1 (one 8-byte
//    load per lane and channel pair feeds both).  The tap rows are ordered so that lane half kk = 0
//    holds the taps of output column parity b = 0 (s = 1, 3) and half 1 those of b = 1 (s = 2, 4, 0),
//    register e of both halves having the same column shift: e < 5 none (r = e), 5 <= e < 10 from
//    q - 1, 10 <= e < 15 from q + 1.  With the even / odd blocks "q - 1" of A is B's lane li - 1 and
//    of B is A's own lane: per (r, block) two plain adds and ONE v_fmac_f32_dpp (wave shift by one
//    lane, times a lane mask that clears the value crossing the lane-31 / 32 boundary).
//  * rows: input row p adds its five kernel rows into a sliding window of output rows
//    2p - 1 .. 2p + 3; rows 2p - 1 and 2p are complete afterwards.  The first four output rows of a
//    wave also need the last rows of the wave above: they are kept, every wave leaves its unfinished
//    window (3 rows) in LDS at the end, one barrier, the wave below adds them and emits.
//  * emission: two output rows at a time; two v_permlane32_swap give every lane 4 ADJACENT pixels of
//    one row (lanes 0-31 one row, 32-63 the other): bias, activation, target / mask, squared error,
//    dL/dpre with 16-byte accesses of full 512-byte rows.
// ---------------------------------------------------------------------------------------------
#define UM_R 8                                  // input rows per wave
typedef unsigned int uintx2m __attribute__((ext_vector_type(2)));

template <bool LOSS>
__global__ __launch_bounds__(512) void k_up_c1m(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ target, const float* __restrict__ mask,
    float* __restrict__ dpre, float* __restrict__ partial, BnGeom g, int act, float slope) {
    __shared__ float xch[8 * 6 * 64];                // per wave: 3 window rows x 2 blocks x 64 lanes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_waves = g.Hs / UM_R;
    const int li = lane & 31, kk = lane >> 5;
    const int n = blockIdx.x, bch = blockIdx.y;
    const int p0 = wv * UM_R;
    const int HWs = g.Hs * g.Ws;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * HWs * 4), 0x00020000);
    const int obytes = (int)((size_t)g.N * g.Cb * g.Hb * g.Wb * 4);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)target, 0, LOSS ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)mask, 0, (LOSS && mask) ? obytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dpre, 0, LOSS ? obytes : 0, 0x00020000);

    // A operand: lane (li, kk) = row m = li of the tap matrix for channel 2t + kk
    float wA[16];
    {
        const int e = (li & 3) + 4 * (li >> 3), half = (li >> 2) & 1;
        int r = -1, sx = 0;
        if (e < 5) { r = e; sx = half ? 2 : 1; }
        else if (e < 10) { r = e - 5; sx = half ? 4 : 3; }
        else if (e < 15 && half) { r = e - 10; sx = 0; }
        // through LDS: the weights of this output channel are read once per workgroup with
        // coalesced loads (16 gathers per lane and wave kept the CU's one vector-memory pipe busy for
        // the first microseconds of enc.conv0's first generation)
        for (int e2 = tid; e2 < 32 * 25; e2 += blockDim.x) {
            const int c = e2 / 25;
            xch[e2] = c < g.Cs ? w[(c * g.Cb + bch) * 25 + (e2 - c * 25)] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) wA[t] = r >= 0 ? xch[(2 * t + kk) * 25 + r * 5 + sx] : 0.f;
        __syncthreads();                               // (xch is reused for the window exchange)
    }
    // lane masks of the two shifted classes: the wave shift carries lane 31 into lane 32 (and 32
    // into 31), which are different column blocks' ends
    const float m_shr = lane == 32 ? 0.f : 1.f;
    const float m_shl = lane == 31 ? 0.f : 1.f;

    // B operand: small[n, 2t + kk, p, 2 li .. 2 li + 1]
    const int x_vo = (kk * HWs + 2 * li) * 4;
    auto load_row = [&](floatx2p (&x)[16], const int p) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const bool ok = 2 * t + kk < g.Cs;
            x[t] = __builtin_bit_cast(floatx2p, __builtin_amdgcn_raw_buffer_load_b64(
                rs, ok ? x_vo : ED_OOB, ((n * g.Cs + 2 * t) * g.Hs + p) * g.Ws * 4, 0));
        }
    };

    const float bs = bias ? bias[bch] : 0.f;
    float sq = 0.f;
    // two complete output rows (h1: registers a1 / b1 of blocks A / B, h2: a2 / b2) -> HBM
    // byte offset of this lane's four pixels in rows (h1 | h2), out of range for rows outside the frame
    auto row_vo = [&](const int h1, const int h2) __attribute__((always_inline)) {
        const int h = kk ? h2 : h1;
        return (h >= 0 && h < g.Hb) ? (((n * g.Cb + bch) * g.Hb + h) * g.Wb + 4 * li) * 4 : ED_OOB;
    };
    // target and mask of a row pair, requested ahead of the multiplications that complete the rows
    auto load_tm = [&](floatx4e& t4, floatx4e& m4, const int vo) __attribute__((always_inline)) {
        if (LOSS) {
            t4 = __builtin_bit_cast(floatx4e, __builtin_amdgcn_raw_buffer_load_b128(rt, vo, 0, 0));
            if (mask) m4 = __builtin_bit_cast(floatx4e, __builtin_amdgcn_raw_buffer_load_b128(rm, vo, 0, 0));
        }
    };
    auto emit2 = [&](const float a1, const float b1, const float a2, const float b2, const int vo,
                     const floatx4e t4, const floatx4e m4in) __attribute__((always_inline)) {
        // v_permlane32_swap x, y: x = [x.lo, y.lo], y = [x.hi, y.hi] (lanes 0-31 | 32-63).  Inline
        // asm: with two __builtin_amdgcn_permlane32_swap in a row this hipcc re-used the register of
        // the first one's second result before it was read (tools/lab/swap_probe.hip has the
        // semantics; the s_nop covers the VALU-write -> permlane hazard the compiler pads itself)
        float v[4] = {a1, a2, b1, b2};
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[1]));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[2]), "+v"(v[3]));
        // lanes 0-31: row h1, lanes 32-63: row h2; columns 4 li .. 4 li + 3
        const bool ok = vo != ED_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] += bs;
            if (act == BN_ACT_SIGMOID) v[i] = __builtin_amdgcn_rcpf(1.f + __expf(-v[i]));
            else v[i] = bn_apply_act(v[i], act, slope);
        }
        if (out) {
            const floatx4e o4 = {v[0], v[1], v[2], v[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4e, o4), ro, vo, 0, 0);
        }
        if (LOSS) {
            floatx4e m4 = {1.f, 1.f, 1.f, 1.f};
            if (mask) m4 = m4in;
            floatx4e d4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dx = v[i] - t4[i];
                if (ok) sq = fmaf(dx * dx, m4[i], sq);
                d4[i] = 2.f * dx * m4[i] * bn_act_grad_from_output(v[i], act, slope);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4e, d4), rd, vo, 0, 0);
        }
    };

    // sliding window of output rows 2p - 1 .. 2p + 3 (index = kernel row r), per column block
    float oa[5], ob[5], ha[4], hb[4];
#pragma unroll
    for (int r = 0; r < 5; ++r) oa[r] = ob[r] = 0.f;

    // three register sets of input rows: row p is multiplied while p + 1 and p + 2 are in flight
    floatx2p x[3][16];
    load_row(x[0], p0);
    load_row(x[1], p0 + 1);
#pragma unroll
    for (int i = 0; i < UM_R; ++i) {
        const int p = p0 + i;
        if (i + 2 < UM_R) load_row(x[(i + 2) % 3], p + 2);
        floatx4e t4 = {0.f, 0.f, 0.f, 0.f}, m4 = {1.f, 1.f, 1.f, 1.f};
        const int vo = row_vo(2 * p - 1, 2 * p);
        if (i >= 2) load_tm(t4, m4, vo);
        floatx16 accA, accB;
#pragma unroll
        for (int e = 0; e < 16; ++e) accA[e] = accB[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[t], x[i % 3][t].x, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[t], x[i % 3][t].y, accB, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            // block A (q = 2 li): same q own block; q + 1 = B's lane; q - 1 = B's lane li - 1
            oa[r] += accA[r] + accB[10 + r];
            asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                         : "+v"(oa[r]) : "v"(accB[5 + r]), "v"(m_shr));
            // block B (q = 2 li + 1): q - 1 = A's lane; q + 1 = A's lane li + 1
            ob[r] += accB[r] + accA[5 + r];
            asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                         : "+v"(ob[r]) : "v"(accA[10 + r]), "v"(m_shl));
        }
        // rows 2p - 1 (index 0) and 2p (index 1) are complete, except in the first two rows of the
        // wave, whose output rows wait for the wave above
        if (i < 2) {
            ha[2 * i] = oa[0]; hb[2 * i] = ob[0];
            ha[2 * i + 1] = oa[1]; hb[2 * i + 1] = ob[1];
        } else {
            emit2(oa[0], ob[0], oa[1], ob[1], vo, t4, m4);
        }
        oa[0] = oa[2]; oa[1] = oa[3]; oa[2] = oa[4]; oa[3] = 0.f; oa[4] = 0.f;
        ob[0] = ob[2]; ob[1] = ob[3]; ob[2] = ob[4]; ob[3] = 0.f; ob[4] = 0.f;
    }
    // targets of the rows that waited, requested before the barrier
    floatx4e th0 = {0.f, 0.f, 0.f, 0.f}, mh0 = {1.f, 1.f, 1.f, 1.f}, th1 = th0, mh1 = mh0, th2 = th0, mh2 = mh0;
    const int vo_h0 = row_vo(2 * p0 - 1, 2 * p0), vo_h1 = row_vo(2 * p0 + 1, 2 * p0 + 2);
    const int vo_h2 = row_vo(2 * (p0 + UM_R - 1) + 1, -1);
    load_tm(th0, mh0, vo_h0);
    load_tm(th1, mh1, vo_h1);
    if (wv == n_waves - 1) load_tm(th2, mh2, vo_h2);
    // unfinished window (rows 2 pe + 1 .. 2 pe + 3) -> the wave below
    float* slot = xch + wv * (6 * 64);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        slot[(2 * r) * 64 + lane] = oa[r];
        slot[(2 * r + 1) * 64 + lane] = ob[r];
    }
    __syncthreads();
    if (wv > 0) {
        const float* up = xch + (wv - 1) * (6 * 64);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ha[r] += up[(2 * r) * 64 + lane];
            hb[r] += up[(2 * r + 1) * 64 + lane];
        }
    }
    emit2(ha[0], hb[0], ha[1], hb[1], vo_h0, th0, mh0);
    emit2(ha[2], hb[2], ha[3], hb[3], vo_h1, th1, mh1);
    if (wv == n_waves - 1) emit2(oa[0], ob[0], 0.f, 0.f, vo_h2, th2, mh2);          // last row of the frame
    if (LOSS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if (lane == 0) partial[(n * g.Cb + bch) * n_waves + wv] = sq;
    }
}

static bool up_c1m_ok(const BnGeom& g) {
    // (one workgroup per frame and output channel: small batches keep the finer-grained k_up_c1v)
    return g.N * g.Cb >= 128 &&
           g.Cs <= 32 && (g.Hs % UM_R) == 0 && g.Hs / UM_R <= 8 && g.Ws == 64 && g.pt == 1 && g.pl == 1 &&
           (size_t)g.N * g.Cb * g.Hb * g.Wb * 4 < 0x7fffffffull;
}

BnFastPlan bn_edge_up_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_up_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || g.stride != 2 || g.Cb > 4 || g.pt != 1 || g.pl != 1) return p;
    if (g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return p;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return p;
    if (g.Ws != UV_W || (g.Hs % 8) != 0) {
        // round 4: any other small map in blocks of 62 columns (k_up_c1v<8, .., GENW>; variant 9; the fused
        // pixel loss since round 5)
        static int off = -1;                          // BN_UP_C1G=0: off
        if (off < 0) { const char* e = bn_tune_env("BN_UP_C1G"); off = (e && e[0] == '0') ? 1 : 0; }
        if (off) return p;
        p.supported = true;
        p.variant = 9;
        p.kernel_name = "k_up_c1v<8, false, gen>";
        return p;
    }
    p.supported = true;
    p.kernel_name = bn_edge_up_kernel_name(g, false);
    return p;
}

const char* bn_edge_up_kernel_name(const BnGeom& g, bool loss) {
    if (g.Ws != UV_W || (g.Hs % 8) != 0) return loss ? "k_up_c1v<8, true, gen>" : "k_up_c1v<8, false, gen>";
    if (UP_C1_VARIANT == 1 && up_c1m_ok(g)) return loss ? "k_up_c1m<true>" : "k_up_c1m<false>";
    return loss ? "k_up_c1v<8, true>" : "k_up_c1v<8, false>";
}

#define UV_R 8          // strip height of the production instantiation

static bool up_c1_gen(const BnGeom& g) { return g.Ws != UV_W || (g.Hs % 8) != 0; }
int bn_edge_up_parts_per_frame(const BnGeom& g) {
    if (up_c1_gen(g)) return g.Cb * ((g.Hs + UV_R - 1) / UV_R) * ((g.Ws + 61) / 62);     // strips x column blocks
    return g.Cb * (g.Hs / UV_R);
}

// target == nullptr: plain forward (out required).  Otherwise the loss epilogue: `dpre` and
// `partial` (N * bn_edge_up_parts_per_frame floats) are written, `out` only if non-null.
int bn_launch_edge_up(const float* small, const float* w, const float* bias, float* out,
                      const BnGeom& g, int act, float slope, hipStream_t st, const float* target,
                      const float* mask, float* dpre, float* partial) {
#ifdef BN_TUNING
    static int old = -1;
    if (old < 0) { const char* e = bn_tune_env("BN_UP_C1_OLD"); old = (e && e[0] == '1') ? 1 : 0; }
    if (old && !target) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)k_up_c1,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, UC_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        BN_LAUNCH_MAIN(k_up_c1, dim3(g.N * (g.Hs / UC_TH), g.Cb), dim3(ED_THREADS), UC_LDS, st,
                           small, w, bias, out, g, act, slope);
        BN_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (up_c1_gen(g)) {
        const int unitsg = g.N * g.Cb * ((g.Hs + UV_R - 1) / UV_R) * ((g.Ws + 61) / 62);
        // (the loss epilogue reads / writes pixel pairs: 8-byte accesses need an even row length, which
        // Wb = 2 Ws is, and 8-byte aligned tensors)
        if (target && (((((uintptr_t)target) | ((uintptr_t)dpre) | ((uintptr_t)mask) | ((uintptr_t)out)) & 7u) != 0))
            return BN_E_SHAPE;
        if (target)
            BN_LAUNCH_MAIN((k_up_c1v<UV_R, true, true>), dim3(unitsg < 256 * 16 ? unitsg : 256 * 16), dim3(64), 0, st,
                               small, w, bias, out, target, mask, dpre, partial, g, act, slope, unitsg);
        else
            BN_LAUNCH_MAIN((k_up_c1v<UV_R, false, true>), dim3(unitsg < 256 * 16 ? unitsg : 256 * 16), dim3(64), 0, st,
                               small, w, bias, out, nullptr, nullptr, nullptr, nullptr, g, act, slope, unitsg);
        BN_LAUNCH_CHECK();
        return 0;
    }
    int use_m = UP_C1_VARIANT == 1 && up_c1m_ok(g);
    if (const char* e = bn_tune_env("BN_UP_C1_M")) use_m = atoi(e) && up_c1m_ok(g);     // (tuning build only)
    if (use_m) {
        const dim3 grid_m(g.N, g.Cb), block_m(64 * (g.Hs / UM_R));
        if (target)
            BN_LAUNCH_MAIN((k_up_c1m<true>), grid_m, block_m, 0, st, small, w, bias, out, target, mask,
                               dpre, partial, g, act, slope);
        else
            BN_LAUNCH_MAIN((k_up_c1m<false>), grid_m, block_m, 0, st, small, w, bias, out, nullptr,
                               nullptr, nullptr, nullptr, g, act, slope);
        BN_LAUNCH_CHECK();
        return 0;
    }
    const int units = g.N * g.Cb * (g.Hs / UV_R);
    const int grid = units < 256 * 16 ? units : 256 * 16;
    if (target) {
        BN_LAUNCH_MAIN((k_up_c1v<UV_R, true>), dim3(grid), dim3(64), 0, st, small, w, bias, out,
                           target, mask, dpre, partial, g, act, slope, units);
    } else {
#ifdef BN_TUNING
        static int r4 = -1;
        if (r4 < 0) { const char* e = bn_tune_env("BN_UP_C1_R4"); r4 = e ? atoi(e) : 0; }
        if (r4 == 1) {
            const int units4 = g.N * g.Cb * (g.Hs / 4);
            BN_LAUNCH_MAIN((k_up_c1v<4, false>), dim3(units4 < 256 * 24 ? units4 : 256 * 24),
                               dim3(64), 0, st, small, w, bias, out, nullptr, nullptr, nullptr,
                               nullptr, g, act, slope, units4);
            BN_LAUNCH_CHECK();
            return 0;
        }
        if (r4 == 16) {
            const int units16 = g.N * g.Cb * (g.Hs / 16);
            BN_LAUNCH_MAIN((k_up_c1v<16, false>), dim3(units16), dim3(64), 0, st, small, w, bias,
                               out, nullptr, nullptr, nullptr, nullptr, g, act, slope, units16);
            BN_LAUNCH_CHECK();
            return 0;
        }
#endif
        BN_LAUNCH_MAIN((k_up_c1v<UV_R, false>), dim3(grid), dim3(64), 0, st, small, w, bias, out,
                           nullptr, nullptr, nullptr, nullptr, g, act, slope, units);
    }
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Stride-1 gather-down onto ONE or TWO channels (round 4): the last transposed layer of a max-pooling architecture
// (its forward is the gather-down of the flipped layer, capi.hip) with any odd kernel the reference's search draws
// (3 / 5 / 7 / 9).  One output channel leaves 31 of the 32 rows of a matrix-core tile idle, so this is a vector
// kernel: a workgroup owns 16 x 64 output pixels of a frame, walks the input channels four at a time through LDS
// ((16 + K - 1) x (64 + K - 1) pixels each, 0.0f off the frame: any padding, any map) and every thread keeps four
// adjacent pixels of a row per output channel -- a row of K + 3 LDS words feeds 4 K multiply-adds per output channel;
// the taps are wave-uniform (scalar loads).  It ran as k_gemm_mfma + k_col2im: 1.6 ms per launch for 16 -> 1 channels,
// 9x9 on 128x128 frames.
// ---------------------------------------------------------------------------------------------
#define S1C_TH 16
#define S1C_TW 64
#define S1C_CC 4
template <int KS, int NOUT>
__global__ __launch_bounds__(256) void k_down_s1_c1(const float* __restrict__ big, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     const float* __restrict__ dact_src, BnGeom g, int act, int dact,
                                                     float slope, int tiles_h, int tiles_w) {
    constexpr int IH = S1C_TH + KS - 1, IWP = (S1C_TW + KS - 1 + 3) & ~3, NV = (KS + 3 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[S1C_CC * IH * IWP];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int h0 = th * S1C_TH, w0 = tw * S1C_TW;
    const size_t HWb = (size_t)g.Hb * g.Wb;
    float acc[NOUT][4];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
    // the tile of the NEXT four channels travels through registers while this one is multiplied (its element ->
    // (channel, row, column) decode and the frame-edge test are done once per thread, not once per chunk)
    constexpr int NL = (S1C_CC * IH * IWP + 255) / 256;
    int goff[NL];                                         // offset inside the frame's channel block, -1 = off the frame
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = tid + 256 * i;
        const int cc = e / (IH * IWP), rem = e - cc * (IH * IWP);
        const int y = rem / IWP, xx = rem - y * IWP;
        const int hb = h0 - g.pt + y, wb = w0 - g.pl + xx;
        const bool ok = e < S1C_CC * IH * IWP && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
        goff[i] = ok ? (cc << 24) | (hb * g.Wb + wb) : -1;
    }
    const float* frame = big + (size_t)n * g.Cb * HWb;
    float nxt[NL];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int cc = goff[i] >> 24;
            nxt[i] = (goff[i] >= 0 && c0 + cc < g.Cb) ? frame[(size_t)(c0 + cc) * HWb + (goff[i] & 0xffffff)] : 0.f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < g.Cb; c0 += S1C_CC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (tid + 256 * i < S1C_CC * IH * IWP) tile[tid + 256 * i] = nxt[i];
        __syncthreads();
        if (c0 + S1C_CC < g.Cb) fetch(c0 + S1C_CC);
#pragma unroll
        for (int cc = 0; cc < S1C_CC; ++cc) {
            const int c = min(c0 + cc, g.Cb - 1);          // (past the last channel the tile holds zeros)
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                float row[4 * NV];
                const float4* rp = reinterpret_cast<const float4*>(tile + (cc * IH + ty + r) * IWP + 4 * tx);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 q = rp[v];
                    row[4 * v] = q.x; row[4 * v + 1] = q.y; row[4 * v + 2] = q.z; row[4 * v + 3] = q.w;
                }
#pragma unroll
                for (int o = 0; o < NOUT; ++o) {
                    const float* wr = w + (((size_t)o * g.Cb + c) * KS + r) * KS;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float wv = wr[s];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(row[s + j], wv, acc[o][j]);
                    }
                }
            }
        }
    }
    const int h = h0 + ty, wq = w0 + 4 * tx;
    if (h >= g.Hs) return;
    const size_t PQ = (size_t)g.Hs * g.Ws;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float bz = bias ? bias[o] : 0.f;
        const size_t base = ((size_t)n * NOUT + o) * PQ + (size_t)h * g.Ws + wq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wq + j >= g.Ws) continue;
            float v = bn_apply_act(acc[o][j] + bz, act, slope);
            if (dact_src) v *= bn_act_grad_from_output(dact_src[base + j], dact, slope);
            out[base + j] = v;
        }
    }
}

bool bn_s1c1_ok(const BnGeom& g) {
    if (g.stride != 1 || g.R != g.S || (g.R != 3 && g.R != 5 && g.R != 7 && g.R != 9)) return false;
    // (four channels with 5x5 taps: the four phases of a 7x7 / 9x9 stride-2 gather-up onto one channel, capi.hip)
    if ((g.Cs > 2 && !(g.Cs == 4 && g.R == 5)) || g.CsS) return false;
    if ((size_t)g.Hb * g.Wb >= (1u << 24)) return false;        // (a pixel's offset in its plane rides in 24 bits)
    const size_t tiles = (size_t)g.N * ((g.Hs + S1C_TH - 1) / S1C_TH) * ((g.Ws + S1C_TW - 1) / S1C_TW);
    return tiles < 0x7fffffffull;
}

int bn_launch_s1c1(const float* big, const float* w, const float* bias, float* out, const float* dact_src,
                   const BnGeom& g, int act, int dact, float slope, hipStream_t st) {
    if (!bn_s1c1_ok(g)) return BN_E_SHAPE;
    const int tiles_h = (g.Hs + S1C_TH - 1) / S1C_TH, tiles_w = (g.Ws + S1C_TW - 1) / S1C_TW;
    const dim3 grid((unsigned)((size_t)g.N * tiles_h * tiles_w));
#define S1C_CASE(K, O)                                                                                      \
    if (g.R == K && g.Cs == O) {                                                                            \
        BN_LAUNCH_MAIN((k_down_s1_c1<K, O>), grid, dim3(256), 0, st, big, w, bias, out, dact_src, g, act,   \
                       dact, slope, tiles_h, tiles_w);                                                      \
        BN_LAUNCH_CHECK();                                                                                  \
        return 0;                                                                                           \
    }
    S1C_CASE(3, 1) S1C_CASE(3, 2) S1C_CASE(5, 1) S1C_CASE(5, 2) S1C_CASE(5, 4) S1C_CASE(7, 1) S1C_CASE(7, 2) S1C_CASE(9, 1)
    S1C_CASE(9, 2)
#undef S1C_CASE
    return BN_E_SHAPE;
}

// ---------------------------------------------------------------------------------------------
// Stride-1 gather-down FROM one or two channels (round 4): the first layer of every max-pooling architecture (1 -> 16
// channels on the full frame) and the data gradient of its mirror image in the decoder.  It ran on the matrix-core
// kernel with three of the four channels of a chunk masked to zero (321 us for 1 -> 16 channels, 5x5, 256 frames of
// 128x128: 3.4 GFLOP of arithmetic and 268 MB of output).  A vector kernel again: a workgroup loads ONE
// (16 + K - 1) x (64 + K - 1) input tile per input channel and walks the output channels eight at a time over it;
// a thread keeps four adjacent pixels of eight channels (a row of K + 3 LDS words feeds 32 K multiply-adds), the
// taps are wave-uniform (scalar loads), the four pixels leave as one 16-byte store where the row allows it.
// ---------------------------------------------------------------------------------------------
#define S1I_MG 8
template <int KS>
__global__ __launch_bounds__(256) void k_down_s1_in1(const float* __restrict__ big, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      const float* __restrict__ dact_src, BnGeom g, int act, int dact,
                                                      float slope, int tiles_h, int tiles_w) {
    constexpr int IH = S1C_TH + KS - 1, IWP = (S1C_TW + KS - 1 + 3) & ~3, NV = (KS + 3 + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[2 * IH * IWP];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int h0 = th * S1C_TH, w0 = tw * S1C_TW;
    const size_t HWb = (size_t)g.Hb * g.Wb;
    for (int e = tid; e < g.Cb * IH * IWP; e += 256) {
        const int cc = e / (IH * IWP), rem = e - cc * (IH * IWP);
        const int y = rem / IWP, xx = rem - y * IWP;
        const int hb = h0 - g.pt + y, wb = w0 - g.pl + xx;
        const bool ok = hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
        tile[e] = ok ? big[((size_t)n * g.Cb + cc) * HWb + (size_t)hb * g.Wb + wb] : 0.f;
    }
    __syncthreads();
    const int h = h0 + ty, wq = w0 + 4 * tx;
    const size_t PQ = (size_t)g.Hs * g.Ws;
    const bool vec = (g.Ws & 3) == 0 && ((((uintptr_t)out) | ((uintptr_t)dact_src)) & 15u) == 0;
    for (int m0 = 0; m0 < g.Cs; m0 += S1I_MG) {
        float acc[S1I_MG][4];
#pragma unroll
        for (int o = 0; o < S1I_MG; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[o][j] = 0.f;
        for (int c = 0; c < g.Cb; ++c) {
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                float row[4 * NV];
                const float4* rp = reinterpret_cast<const float4*>(tile + (c * IH + ty + r) * IWP + 4 * tx);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 q = rp[v];
                    row[4 * v] = q.x; row[4 * v + 1] = q.y; row[4 * v + 2] = q.z; row[4 * v + 3] = q.w;
                }
#pragma unroll
                for (int o = 0; o < S1I_MG; ++o) {
                    const int m = min(m0 + o, g.Cs - 1);            // (channels past the last one are not stored)
                    const float* wr = w + (((size_t)m * g.Cb + c) * KS + r) * KS;
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float wv = wr[s];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(row[s + j], wv, acc[o][j]);
                    }
                }
            }
        }
        if (h < g.Hs) {
#pragma unroll
            for (int o = 0; o < S1I_MG; ++o) {
                const int m = m0 + o;
                if (m >= g.Cs) break;
                const float bz = bias ? bias[m] : 0.f;
                const size_t base = ((size_t)n * g.Cs + m) * PQ + (size_t)h * g.Ws + wq;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = bn_apply_act(acc[o][j] + bz, act, slope);
                if (vec && wq + 3 < g.Ws) {
                    if (dact_src) {
                        const float4 d = *reinterpret_cast<const float4*>(dact_src + base);
                        v[0] *= bn_act_grad_from_output(d.x, dact, slope);
                        v[1] *= bn_act_grad_from_output(d.y, dact, slope);
                        v[2] *= bn_act_grad_from_output(d.z, dact, slope);
                        v[3] *= bn_act_grad_from_output(d.w, dact, slope);
                    }
                    *reinterpret_cast<float4*>(out + base) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (wq + j >= g.Ws) continue;
                        float u = v[j];
                        if (dact_src) u *= bn_act_grad_from_output(dact_src[base + j], dact, slope);
                        out[base + j] = u;
                    }
                }
            }
        }
    }
}

// Round 6: the same layer with 5x5 taps and 16 k output channels on the matrix cores.  The vector kernel above spends
// 3.4 GFLOP of v_fma on the first layer of the max-pooling test architecture (83 us, vector-ALU bound, for 285 MB).
// v_mfma_f32_16x16x4_f32 with the TAPS as the reduction: A = 16 channels x 4 taps (28 = 7 steps for one input channel,
// 52 = 13 for two; held in registers), B = 4 taps x 16 adjacent pixels of a row, read from the same LDS tile through
// per-lane tap offsets computed once (one ds_read_b32 per instruction: 19 consecutive words, no conflicts); a wave owns
// 4 rows x 64 columns, four pixel blocks in flight (independent accumulators).  Output: lane = (pixel, 4 channels).
typedef float floatx4e __attribute__((ext_vector_type(4)));
// POOL: the 2x2 / stride-2 max pooling and the activation that follow the first layer of a max-pooling architecture in
// the epilogue (ref aes.py:200-211: conv -> pool -> LeakyReLU).  A lane pair (j, j ^ 1) holds a window's two columns,
// two rows are multiplied before the epilogue; the even lane picks the winner in row-major window order (a later
// element wins only if it is strictly larger or NaN: k_maxpool_fwd_k2's rule, torch's indices h W + w) and stores the
// activated maximum and its index: the 268 MB of the layer's output are neither written nor read back (`pidx` != null).
template <int CIN, bool POOL = false>
__global__ __launch_bounds__(256) void k_down_s1_in1m(const float* __restrict__ big, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const float* __restrict__ dact_src, BnGeom g, int act, int dact,
                                                       float slope, int tiles_h, int tiles_w, int* __restrict__ pidx = nullptr) {
    constexpr int KS = 5, IH = S1C_TH + KS - 1, IWP = (S1C_TW + KS - 1 + 3) & ~3;
    constexpr int KT = CIN * KS * KS, NK = (KT + 3) / 4;
    __shared__ __attribute__((aligned(16))) float tile[CIN * IH * IWP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kg = lane >> 4;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int h0 = th * S1C_TH, w0 = tw * S1C_TW;
    const size_t HWb = (size_t)g.Hb * g.Wb;
    for (int e = tid; e < CIN * IH * IWP; e += 256) {
        const int cc = e / (IH * IWP), rem = e - cc * (IH * IWP);
        const int y = rem / IWP, xx = rem - y * IWP;
        const int hb = h0 - g.pt + y, wb = w0 - g.pl + xx;
        const bool ok = hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
        tile[e] = ok ? big[((size_t)n * g.Cb + cc) * HWb + (size_t)hb * g.Wb + wb] : 0.f;
    }
    int toff[NK];                                    // tap 4 t + kg -> its word offset in the tile (0 for no tap)
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        const int tap = 4 * t + kg;
        const int cc = tap / (KS * KS), rem = tap - cc * (KS * KS);
        const int r = rem / KS, sx = rem - r * KS;
        toff[t] = tap < KT ? (cc * IH + r) * IWP + sx : 0;
    }
    __syncthreads();
    const size_t PQ = (size_t)g.Hs * g.Ws;
    for (int m0 = 0; m0 < g.Cs; m0 += 16) {
        float a[NK];
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const int tap = 4 * t + kg;
            a[t] = (tap < KT && m0 + j < g.Cs) ? w[(size_t)(m0 + j) * KT + tap] : 0.f;
        }
        float bz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bz[e] = (bias && m0 + 4 * kg + e < g.Cs) ? bias[m0 + 4 * kg + e] : 0.f;
        if constexpr (POOL) {
            const int Ho = g.Hs >> 1, Wo = g.Ws >> 1;
#pragma unroll 1
            for (int rp = 0; rp < 2; ++rp) {
                const int row = 4 * wv + 2 * rp;
                floatx4e a0[4], a1[4];
#pragma unroll
                for (int xb = 0; xb < 4; ++xb) a0[xb] = a1[xb] = (floatx4e){0.f, 0.f, 0.f, 0.f};
                const float* tp = tile + row * IWP + j;
#pragma unroll
                for (int t = 0; t < NK; ++t) {
#pragma unroll
                    for (int xb = 0; xb < 4; ++xb) {
                        a0[xb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tp[toff[t] + 16 * xb], a0[xb], 0, 0, 0);
                        a1[xb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tp[toff[t] + IWP + 16 * xb], a1[xb], 0, 0, 0);
                    }
                }
                const int h = h0 + row;
#pragma unroll
                for (int xb = 0; xb < 4; ++xb) {
                    const int wq = w0 + 16 * xb + j;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v0 = a0[xb][e] + bz[e], v1 = a1[xb][e] + bz[e];
                        const float n0 = __shfl_xor(v0, 1, 64), n1 = __shfl_xor(v1, 1, 64);
                        const int m = m0 + 4 * kg + e;
                        if ((j & 1) || h >= g.Hs || wq >= g.Ws || m >= g.Cs) continue;
                        const int me = h * g.Ws + wq;
                        float best = -INFINITY;
                        int bi = me;
                        if (v0 > best || isnan(v0)) { best = v0; bi = me; }
                        if (n0 > best || isnan(n0)) { best = n0; bi = me + 1; }
                        if (v1 > best || isnan(v1)) { best = v1; bi = me + g.Ws; }
                        if (n1 > best || isnan(n1)) { best = n1; bi = me + g.Ws + 1; }
                        const size_t o = (((size_t)n * g.Cs + m) * Ho + (h >> 1)) * Wo + (wq >> 1);
                        out[o] = bn_apply_act(best, act, slope);
                        pidx[o] = bi;
                    }
                }
            }
            continue;
        }
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * wv + rr;
            floatx4e acc[4];
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) acc[xb] = (floatx4e){0.f, 0.f, 0.f, 0.f};
            const float* tp = tile + row * IWP + j;
#pragma unroll
            for (int t = 0; t < NK; ++t) {
#pragma unroll
                for (int xb = 0; xb < 4; ++xb)
                    acc[xb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], tp[toff[t] + 16 * xb], acc[xb], 0, 0, 0);
            }
            const int h = h0 + row;
            if (h >= g.Hs) continue;
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) {
                const int wq = w0 + 16 * xb + j;
                if (wq >= g.Ws) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + 4 * kg + e;
                    if (m >= g.Cs) continue;
                    const size_t idx = ((size_t)n * g.Cs + m) * PQ + (size_t)h * g.Ws + wq;
                    float v = bn_apply_act(acc[xb][e] + bz[e], act, slope);
                    if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
                    out[idx] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the FIRST layer of a max-pooling architecture straight from the POOLED gradient (round 6).  The
// layer's output gradient is 3/4 zeros -- one winner per 2x2 window -- and was materialised (bn_maxpool2d_act_bwd:
// 268 MB written for 256 frames of 16 x 128 x 128) only to be read back by the weight gradient (the first layer has no
// data gradient).  Here
//   dW[a][c][r][s] = sum_{n,ph,pw} g[n,a,ph,pw] x[n, c, h* + r - pt, w* + s - pl],   g = dy * act'(y),
// with (h*, w*) the window's winner (idx = h* Ws + w*): a workgroup loads the (16 + 4) x (64 + 4) input tile of a frame
// per input channel, every thread owns ONE output channel and 8 x 32 / (256 / Cs) window positions of the tile, reads
// dy / y / idx there (67 MB each instead of 268 + 268) and gathers its 25 taps from LDS at the winner's position;
// 25 CIN accumulators per thread over all the tiles of the workgroup, combined across the channel's threads at the end
// (shuffles), one partial row per workgroup as k_wgrad_c1 leaves them (+ the bias sums).
// ---------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void k_wgrad_pool_c1(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ y, const int* __restrict__ idx,
                                                        float* __restrict__ part, float* __restrict__ bias_part,
                                                        BnGeom g, int n_tiles, int tiles_h, int tiles_w, int act,
                                                        float slope) {
    constexpr int KS = 5, IH = S1C_TH + KS - 1, IWP = (S1C_TW + KS - 1 + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float tile[CIN * IH * IWP];
    const int tid = threadIdx.x;
    const int tpc = 256 / g.Cs;                           // threads of a channel (Cs in {16, 32, 64})
    const int a = tid / tpc, lt = tid - a * tpc;
    const int Ho = g.Hs >> 1, Wo = g.Ws >> 1;
    const size_t HWb = (size_t)g.Hb * g.Wb;
    float acc[CIN][25];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int t = 0; t < 25; ++t) acc[c][t] = 0.f;
    float bsum = 0.f;
    const float dslope = (act == BN_ACT_LRELU) ? slope : 1.f;     // (identity or LeakyReLU behind the pooling)
    for (int tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        int b = tl;
        const int tw = b % tiles_w; b /= tiles_w;
        const int th = b % tiles_h;
        const int n = b / tiles_h;
        const int h0 = th * S1C_TH, w0 = tw * S1C_TW;
        __syncthreads();                                  // the previous tile's gathers are done
        for (int e = tid; e < CIN * IH * IWP; e += 256) {
            const int cc = e / (IH * IWP), rem = e - cc * (IH * IWP);
            const int yy = rem / IWP, xx = rem - yy * IWP;
            const int hb = h0 - g.pt + yy, wb = w0 - g.pl + xx;
            const bool ok = hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            tile[e] = ok ? x[((size_t)n * g.Cb + cc) * HWb + (size_t)hb * g.Wb + wb] : 0.f;
        }
        __syncthreads();
        // window positions of the tile: 8 rows x 32 columns, dealt to the channel's threads
        const size_t pbase = ((size_t)n * g.Cs + a) * Ho * Wo;
        // sixteen positions' loads in flight before the first gather (one position per trip was a chain of 16 x 3
        // dependent global loads per tile: 111 us for 201 MB)
        constexpr int UN = 16;
        for (int p0 = lt; p0 < (S1C_TH / 2) * (S1C_TW / 2); p0 += UN * tpc) {
            float gq[UN], yq[UN];
            int iq[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int p = p0 + u * tpc;
                const int ph = (h0 >> 1) + (p >> 5), pw = (w0 >> 1) + (p & 31);
                const bool ok = ph < Ho && pw < Wo;
                const size_t o = pbase + (size_t)(ok ? ph : 0) * Wo + (ok ? pw : 0);
                gq[u] = ok ? dy[o] : 0.f;
                yq[u] = y[o];
                iq[u] = ok ? idx[o] : (2 * ph) * g.Ws + 2 * pw;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int p = p0 + u * tpc;
                const int pr = p >> 5, pc = p & 31;
                const int me = (2 * ((h0 >> 1) + pr)) * g.Ws + 2 * ((w0 >> 1) + pc);
                const float gv = gq[u] * (yq[u] > 0.f ? 1.f : dslope);
                const int d = iq[u] - me;
                const int dh = d >= g.Ws ? 1 : 0, dw = d - dh * g.Ws;
                bsum += gv;
                const float* xp = tile + (2 * pr + dh) * IWP + 2 * pc + dw;
#pragma unroll
                for (int c = 0; c < CIN; ++c)
#pragma unroll
                    for (int r = 0; r < KS; ++r)
#pragma unroll
                        for (int sx = 0; sx < KS; ++sx)
                            acc[c][r * KS + sx] = fmaf(gv, xp[(c * IH + r) * IWP + sx], acc[c][r * KS + sx]);
            }
        }
    }
    // the channel's threads are tpc adjacent lanes of one wave (tpc <= 16): butterfly, fixed order
    for (int off = tpc >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int t = 0; t < 25; ++t) acc[c][t] += __shfl_xor(acc[c][t], off, 64);
        bsum += __shfl_xor(bsum, off, 64);
    }
    if (lt == 0) {
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int t = 0; t < 25; ++t)
                part[((size_t)c * gridDim.x + blockIdx.x) * (g.Cs * 25) + a * 25 + t] = acc[c][t];
        if (bias_part) bias_part[(size_t)blockIdx.x * g.Cs + a] = bsum;
    }
}

bool bn_wgrad_pool_c1_ok(const BnGeom& g) {
    return g.stride == 1 && g.R == 5 && g.S == 5 && g.Cb >= 1 && g.Cb <= 2 && (g.Cs == 16 || g.Cs == 32 || g.Cs == 64) &&
           !(g.Hs & 1) && !(g.Ws & 1) && g.pt <= 4 && g.pl <= 4 && g.CsS == 0 && (size_t)g.Hs * g.Ws < 0x7fffffffull;
}
static int wgrad_pool_c1_grid(const BnGeom& g) {
    const size_t tiles = (size_t)g.N * ((g.Hs + S1C_TH - 1) / S1C_TH) * ((g.Ws + S1C_TW - 1) / S1C_TW);
    return (int)(tiles < 1024 ? tiles : 1024);
}
size_t bn_wgrad_pool_c1_ws_bytes(const BnGeom& g) {
    const int G = wgrad_pool_c1_grid(g);
    return ((size_t)g.Cb * G * g.Cs * 25 + (size_t)G * g.Cs) * sizeof(float);
}
// dw[a][c][25] (+)= ..., db[a] (+)= sum g  (db nullable); ws >= bn_wgrad_pool_c1_ws_bytes
int bn_launch_wgrad_pool_c1(const float* x, const float* dy, const float* y, const int* idx, float* dw, float* db,
                            const BnGeom& g, int act, float slope, int accumulate, void* ws, hipStream_t st) {
    if (!bn_wgrad_pool_c1_ok(g)) return BN_E_SHAPE;
    if (!ws) return BN_E_WORKSPACE;
    const int tiles_h = (g.Hs + S1C_TH - 1) / S1C_TH, tiles_w = (g.Ws + S1C_TW - 1) / S1C_TW;
    const int n_tiles = g.N * tiles_h * tiles_w, G = wgrad_pool_c1_grid(g);
    float* part = (float*)ws;
    float* bias_part = db ? part + (size_t)g.Cb * G * g.Cs * 25 : nullptr;
    if (g.Cb == 1)
        BN_LAUNCH_MAIN((k_wgrad_pool_c1<1>), dim3(G), dim3(256), 0, st, x, dy, y, idx, part, bias_part, g, n_tiles,
                       tiles_h, tiles_w, act, slope);
    else
        BN_LAUNCH_MAIN((k_wgrad_pool_c1<2>), dim3(G), dim3(256), 0, st, x, dy, y, idx, part, bias_part, g, n_tiles,
                       tiles_h, tiles_w, act, slope);
    BN_LAUNCH_CHECK();
    if (bias_part) {
        const int rc = bn_launch_sum_partials(bias_part, db, g.Cs, G, accumulate, 0, 0, st);
        if (rc) return rc;
    }
    for (int b = 0; b < g.Cb; ++b) {
        const int rc = bn_launch_sum_partials(part + (size_t)b * G * g.Cs * 25, dw + b * 25, g.Cs * 25, G, accumulate,
                                              0, 0, st, 25, g.Cb * 25);
        if (rc) return rc;
    }
    return 0;
}

bool bn_s1in1_ok(const BnGeom& g) {
    if (g.stride != 1 || g.R != g.S || (g.R != 3 && g.R != 5 && g.R != 7 && g.R != 9)) return false;
    if (g.Cb > 2 || g.Cs < 3 || g.CsS) return false;      // (one / two OUTPUT channels: k_down_s1_c1)
    const size_t tiles = (size_t)g.N * ((g.Hs + S1C_TH - 1) / S1C_TH) * ((g.Ws + S1C_TW - 1) / S1C_TW);
    return tiles < 0x7fffffffull;
}

// the pooled form: 5x5 taps, 16 k channels, even maps (every window inside one workgroup tile of 16 x 64 pixels)
bool bn_s1in1_pool_ok(const BnGeom& g) {
    return bn_s1in1_ok(g) && g.R == 5 && (g.Cs & 15) == 0 && !(g.Hs & 1) && !(g.Ws & 1) &&
           (size_t)g.Hs * g.Ws < 0x7fffffffull;
}
int bn_launch_s1in1_pool(const float* big, const float* w, const float* bias, float* y, int* idx, const BnGeom& g,
                         int act, float slope, hipStream_t st) {
    if (!bn_s1in1_pool_ok(g)) return BN_E_SHAPE;
    const int tiles_h = (g.Hs + S1C_TH - 1) / S1C_TH, tiles_w = (g.Ws + S1C_TW - 1) / S1C_TW;
    const dim3 grid((unsigned)((size_t)g.N * tiles_h * tiles_w));
    if (g.Cb == 1)
        BN_LAUNCH_MAIN((k_down_s1_in1m<1, true>), grid, dim3(256), 0, st, big, w, bias, y, (const float*)nullptr, g, act,
                       BN_ACT_NONE, slope, tiles_h, tiles_w, idx);
    else
        BN_LAUNCH_MAIN((k_down_s1_in1m<2, true>), grid, dim3(256), 0, st, big, w, bias, y, (const float*)nullptr, g, act,
                       BN_ACT_NONE, slope, tiles_h, tiles_w, idx);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_s1in1(const float* big, const float* w, const float* bias, float* out, const float* dact_src,
                    const BnGeom& g, int act, int dact, float slope, hipStream_t st) {
    if (!bn_s1in1_ok(g)) return BN_E_SHAPE;
    const int tiles_h = (g.Hs + S1C_TH - 1) / S1C_TH, tiles_w = (g.Ws + S1C_TW - 1) / S1C_TW;
    const dim3 grid((unsigned)((size_t)g.N * tiles_h * tiles_w));
    if (g.R == 5 && (g.Cs & 15) == 0) {             // 16 k output channels: the taps on the matrix cores
        if (g.Cb == 1)
            BN_LAUNCH_MAIN((k_down_s1_in1m<1>), grid, dim3(256), 0, st, big, w, bias, out, dact_src, g, act, dact, slope,
                           tiles_h, tiles_w, (int*)nullptr);
        else
            BN_LAUNCH_MAIN((k_down_s1_in1m<2>), grid, dim3(256), 0, st, big, w, bias, out, dact_src, g, act, dact, slope,
                           tiles_h, tiles_w, (int*)nullptr);
        BN_LAUNCH_CHECK();
        return 0;
    }
#define S1I_CASE(K)                                                                                         \
    if (g.R == K) {                                                                                         \
        BN_LAUNCH_MAIN((k_down_s1_in1<K>), grid, dim3(256), 0, st, big, w, bias, out, dact_src, g, act,     \
                       dact, slope, tiles_h, tiles_w);                                                      \
        BN_LAUNCH_CHECK();                                                                                  \
        return 0;                                                                                           \
    }
    S1I_CASE(3) S1I_CASE(5) S1I_CASE(7) S1I_CASE(9)
#undef S1I_CASE
    return BN_E_SHAPE;
}
