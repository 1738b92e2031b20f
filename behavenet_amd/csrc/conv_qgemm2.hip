// Stride == kernel (5x5, stride 5) layers between an 8x8 and a 2x2 map with offset 1 (enc.conv4 /
// dec.convT0 of the default architecture at 128x128 frames), second generation.
//
// Same arithmetic as conv_qgemm.hip (only 16 of the 25 taps of a window ever meet data: four dense
// quadrant GEMMs per role, z = 2p + q the small-side pixel), different decomposition: a workgroup
// walks ALL FOUR quadrants of its tile, so that
//   * every operand is a contiguous run of the tensor as it lies in memory -- the 8x8 map of a
//     (frame, channel) is the four quadrants' 4x4 blocks, the 25 taps of a (small, big) channel pair
//     hold the four 16-tap slices, the 2x2 map of a (frame, channel) is one 16-byte group -- and goes
//     to LDS by 16-byte LDS-DMA in that order (no registers, no ds_write, no z-major copy of the
//     small side: k_qg_split_small is gone), double buffered, requested from inside the MFMA stream
//     of the previous stage (issue rules): a stage boundary is one s_waitcnt
//     and one barrier;
//   * the weight gradient's quadrants that share a tap meet in the workgroup (k_qg_finish_wgrad and
//     its 33 MB of partial tiles are gone), and its bias gradients come from the operand tiles that
//     pass through LDS anyway;
//   * one workgroup per CU (256 workgroups at the bench size), launch ramp and first-operand
//     latency paid once.
// MFMA: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the reference's arithmetic).
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float q2x16 __attribute__((ext_vector_type(16)));
typedef float q2x4 __attribute__((ext_vector_type(4)));

#define Q2_THREADS 256
#define Q2_OOB 0x7fffffff
#define Q2_MAX_LDS (160 * 1024)

// (a plain function: inside a kernel template the builtin's size argument would be checked at
// instantiation time, where the host pass rejects 16 and silently drops the kernel)
__device__ __forceinline__ void q2_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voffset,
                                         int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (float*)lds, 16, voffset, soffset, 0, 0);
}

// quadrant z = 2 zp + zq: first tap row / column of its 4x4 block inside the 5x5 kernel
#define Q2_R0(z) (((z) >> 1) ? 0 : 1)
#define Q2_S0(z) (((z) & 1) ? 0 : 1)

// ---------------------------------------------------------------------------------------------
// gather-up (convT forward, conv data gradient):
//   out[n][c][pix(z,jj)] = epi( sum_m small[n][m][z] W[m][c][tap(z,jj)] + bias[c] )
// Workgroup = 32 frames x 8 big-side channels x the whole 8x8 map (all four quadrants); wave w owns
// the channel pair 2 w, 2 w + 1.  MFMA rows = 32 columns (2 channels x 16 taps of quadrant z, A =
// weights), MFMA columns = the 32 frames (B = small side), one 32x32 accumulator per quadrant;
// reduction over the small-side channels m in stages of 32.  Per k-step (two m) a lane reads ONE
// 16-byte group of the small side (the four quadrant values of its frame: the B operands of the
// four MFMAs) and four weight words.
//   LDS stage image: small side as 16 row PAIRS of 64 groups + 1 pad group (1040 bytes: one DMA
//   instruction fills a pair; the pad makes the b128 reads of 16 consecutive frames 2-way instead
//   of 16-way conflicted), then the weight rows W[m][8 channels][25] (200 words = 50 groups per m,
//   contiguous and 16-byte aligned in the tensor).  41 DMA instructions per stage and workgroup
//   (~10 per wave per 64 MFMAs: an LDS-DMA instruction costs tens of cycles of issue).
// ---------------------------------------------------------------------------------------------
#define Q2U_KS 32
#define Q2U_PAIR 1040
#define Q2U_SBYTES (16 * Q2U_PAIR)
#define Q2U_WROW 200
#define Q2U_WINSTR 25                             // 32 m x 50 groups = 1600 groups
#define Q2U_WBYTES (Q2U_WINSTR * 1024)
#define Q2U_STAGE (Q2U_SBYTES + Q2U_WBYTES)
#define Q2U_NDMA 11                               // per wave and stage: 4 small-side pairs + 7 (6) weight pieces

__global__ __launch_bounds__(Q2_THREADS, 1) void k_qg2_up(
    const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, const float* __restrict__ dact_src, int N, int Cs, int Cb, int act,
    int dact, float slope) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int cw0 = blockIdx.x * 8, c0 = cw0 + 2 * wv, n0 = blockIdx.y * 32;

    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)N * Cs * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)w, 0, (int)((size_t)Cs * Cb * 100), 0x00020000);

    // DMA lane offsets (stage independent).  Small side: lane = (row of the pair, m within the
    // stage), wave w fills pairs 4 w .. 4 w + 3; frames beyond N read 0.0f (buffer range).
    // Weights: piece d of the 25, waves take d = wv + 4 h.
    const int vo_s = ((n0 + (lane >> 5)) * Cs + (lane & 31)) * 16;
    int vo_w[7];
#pragma unroll
    for (int h = 0; h < 7; ++h) {
        const int grp = 64 * (wv + 4 * h) + lane;
        const int ml = grp / 50, gi = grp - 50 * ml;
        vo_w[h] = (ml * Cb + cw0) * 100 + gi * 16;
    }
    auto issue_dma = [&](const int d, const int buf, const int stage) __attribute__((always_inline)) {
        char* base = smem + buf * Q2U_STAGE;
        if (d < 4) {
            q2_dma16(rs_s, base + (4 * wv + d) * Q2U_PAIR, vo_s, (2 * (4 * wv + d) * Cs + stage * Q2U_KS) * 16);
        } else {
            const int h = d - 4;
            if (wv + 4 * h < Q2U_WINSTR)
                q2_dma16(rs_w, base + Q2U_SBYTES + (wv + 4 * h) * 1024, vo_w[h], stage * Q2U_KS * Cb * 100);
        }
    };

    // operand read addresses (bytes from smem, buffer 0, k-step 0)
    const int ad_s = (li >> 1) * Q2U_PAIR + ((li & 1) * 32 + kh) * 16;
    const int jj = li & 15;
    const int ad_w = Q2U_SBYTES + (kh * Q2U_WROW + (2 * wv + (li >> 4)) * 25 + (jj >> 2) * 5 + (jj & 3)) * 4;

    q2x16 acc[4];
#pragma unroll
    for (int z = 0; z < 4; ++z)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[z][t] = 0.f;

    const int n_stages = Cs / Q2U_KS;
#pragma unroll
    for (int d = 0; d < Q2U_NDMA; ++d) issue_dma(d, 0, 0);

#ifdef Q2U_LOOP_SHIFT
    BN_LOOP_PLACE(8, Q2U_LOOP_SHIFT);
#endif
    for (int st = 0; st < n_stages; ++st) {
        // own DMAs of this stage have landed; behind the barrier everyone's have, and every wave is
        // done reading the other image
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int buf = st & 1;
#ifdef Q2_ABL_NODMA
        const bool more = false;
#else
        const bool more = st + 1 < n_stages;
#endif
        const char* sp = smem + buf * Q2U_STAGE + ad_s;
        // one base register per quadrant, laundered: the compiler then cannot fuse the reads of
        // adjacent taps into ds_read2_b32 (whose 8-bit offsets would cost a v_add per read; every
        // VALU instruction in the stream takes 6-13 cycles from the matrix pipe)
        int wo[4];                   // (integer offsets: a laundered POINTER loses its address space)
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            wo[z] = buf * Q2U_STAGE + ad_w + (Q2_R0(z) * 5 + Q2_S0(z)) * 4;
            asm volatile("" : "+v"(wo[z]));
        }
        q2x4 sv[2];
        float wq[2][4];
        sv[0] = *reinterpret_cast<const q2x4*>(sp);
#pragma unroll
        for (int z = 0; z < 4; ++z) wq[0][z] = *reinterpret_cast<const float*>(smem + wo[z]);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int cu = t & 1, nx = cu ^ 1;
            // operands of k-step t + 1 are requested a whole k-step (256 matrix cycles) ahead
            if (t + 1 < 16) {
                sv[nx] = *reinterpret_cast<const q2x4*>(sp + (t + 1) * 32);
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    wq[nx][z] = *reinterpret_cast<const float*>(smem + wo[z] + (t + 1) * 2 * Q2U_WROW * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next stage's DMA goes out in the first eleven k-steps: the last piece then has five
            // k-steps = 1.3 k cycles of matrix work to land behind
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[cu][0], sv[cu].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[cu][1], sv[cu].y, acc[1], 0, 0, 0);
            if (more && t < Q2U_NDMA) issue_dma(t, buf ^ 1, st + 1);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[cu][2], sv[cu].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[cu][3], sv[cu].w, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: lane = frame n0 + li; register group tq = t >> 2 of quadrant z holds
    // the 16-byte run (channel c0 + (tq >> 1), map row 4 zp + kh + 2 (tq & 1), columns 4 zq .. + 3)
    const int n = n0 + li;
    const int obytes = (int)((size_t)N * Cb * 64 * 4);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)dact_src, 0, dact_src ? obytes : 0, 0x00020000);
    const int vo = (n < N) ? (n * Cb + c0) * 256 : Q2_OOB;
    const float es = (act == BN_ACT_LRELU) ? slope : 1.f;           // identity = slope 1
    const float ds = (dact == BN_ACT_LRELU) ? slope : 1.f;
    float bz[2] = {0.f, 0.f};
    if (bias) { bz[0] = bias[c0]; bz[1] = bias[c0 + 1]; }
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        q2x4 d[4];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const int so = (tq >> 1) * 256 + ((4 * (z >> 1) + 2 * (tq & 1)) * 8 + 4 * (z & 1)) * 4;
            if (dact_src)
                d[tq] = __builtin_bit_cast(q2x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vo + kh * 32, so, 0));
        }
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const int so = (tq >> 1) * 256 + ((4 * (z >> 1) + 2 * (tq & 1)) * 8 + 4 * (z & 1)) * 4;
            q2x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[z][4 * tq + e] + bz[tq >> 1];
                x = x > 0.f ? x : x * es;
                if (dact_src) x *= d[tq][e] > 0.f ? 1.f : ds;
                v[e] = x;
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v),
                                                   ro, vo + kh * 32, so, 0);
        }
    }
}

bool bn_qg2_up_supported(const BnGeom& g, int act, int dact) {
    static int disabled = -1;                          // BN_QG2=0: first generation (conv_qgemm.hip)
    if (disabled < 0) { const char* e = bn_tune_env("BN_QG2"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (!bn_qgemm_supported(g)) return false;
    if (act == BN_ACT_SIGMOID || dact == BN_ACT_SIGMOID) return false;
    if ((g.Cs % Q2U_KS) != 0 || (g.Cb & 7) != 0) return false;
    // one workgroup per 32 frames x 8 channels, each as long as the whole launch: below ~128 of them
    // (a 32-frame shard of a trial: 32) the first generation's finer grid is faster (24 against 40 us)
    if (((g.N + 31) / 32) * (g.Cb / 8) < 128) return false;
    return true;
}

int bn_launch_qg2_up(const float* small, const float* w, const float* bias, float* out,
                     const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                     hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_qg2_up, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Q2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid(g.Cb / 8, (g.N + 31) / 32);
    BN_LAUNCH_MAIN(k_qg2_up, grid, dim3(Q2_THREADS), (size_t)2 * Q2U_STAGE, st, small, w, bias, out,
                   dact_src, g.N, g.Cs, g.Cb, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// weight gradient:
//   dW[m][c][tap] (+)= sum_{z : tap in window(z)} sum_n small[n][m][z] big[n][c][pix_z(tap)]
// Workgroup = 64 small-side channels x 8 big-side channels, all four quadrants; reduction over the
// frames in stages of 16.  MFMA rows = 32 small-side channels (A = small side: ONE 16-byte read per
// k-step gives the A operands of the four quadrants), MFMA columns = 2 channels x 16 taps (B = big
// side, from the 8x8 maps as they lie in memory); a wave owns (m-block w & 1) x (channels
// 4 (w >> 1) .. + 3) x four quadrants = eight 32x32 accumulators.  Three LDS stage images of 48 KB
// (per frame: 64 groups of the small side + the 2 KB of eight 8x8 maps, both contiguous in the
// tensors), requested two stages ahead.  The quadrants that share a tap are added up in the
// workgroup (through LDS, fixed order z = 0..3): no partial tiles, no second kernel.  The bias
// gradient of either side is a by-product: the workgroups of the first column (row) of the grid add
// up the small (big) operand tiles as they pass through LDS.
// ---------------------------------------------------------------------------------------------
#define Q2W_KS 16
#define Q2W_SBYTES (Q2W_KS * 1024)
#define Q2W_BBYTES (Q2W_KS * 2048)
#define Q2W_STAGE (Q2W_SBYTES + Q2W_BBYTES)
#define Q2W_NBUF 3
#define Q2W_NDMA 12                               // per wave and stage: 4 small-side + 8 big-side pieces

__global__ __launch_bounds__(Q2_THREADS, 1) void k_qg2_wgrad(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ dw,
    float* __restrict__ db, int N, int Cs, int Cb, int accumulate, int bias_side) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const int mb = wv & 1, cq = wv >> 1;
    const int c0 = blockIdx.x * 8, m0 = blockIdx.y * 64;

    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)N * Cs * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)N * Cb * 256), 0x00020000);
    // DMA: wave w moves frames 4 w .. 4 w + 3 of a stage: one piece of the small side and two of
    // the big side each; frames beyond N read 0.0f (buffer range)
    const int vo_s = (m0 + lane) * 16;
    const int vo_b = c0 * 256 + lane * 16;
    auto issue_dma = [&](const int d, const int buf, const int stage) __attribute__((always_inline)) {
        char* base = smem + buf * Q2W_STAGE;
        if (d < 4) {
            const int nl = 4 * wv + d;
            q2_dma16(rs_s, base + nl * 1024, vo_s, (stage * Q2W_KS + nl) * Cs * 16);
        } else {
            const int nl = 4 * wv + ((d - 4) >> 1), half = (d - 4) & 1;
            q2_dma16(rs_b, base + Q2W_SBYTES + nl * 2048 + half * 1024, vo_b,
                     (stage * Q2W_KS + nl) * Cb * 256 + half * 1024);
        }
    };

    const int jj = li & 15;
    const int ad_a = (kh * 64 + 32 * mb + li) * 16;
    const int ad_b = Q2W_SBYTES + (kh * 512 + (4 * cq + (li >> 4)) * 64 + (jj >> 2) * 8 + (jj & 3)) * 4;

    q2x16 acc[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int z = 0; z < 4; ++z)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[cb][z][t] = 0.f;
    float bsum = 0.f;
    const bool do_bias1 = (bias_side == 1) && db && blockIdx.x == 0;
    const bool do_bias2 = (bias_side == 2) && db && blockIdx.y == 0;

    const int n_stages = (N + Q2W_KS - 1) / Q2W_KS;
#pragma unroll
    for (int d = 0; d < Q2W_NDMA; ++d) issue_dma(d, 0, 0);
    if (n_stages > 1) {
#pragma unroll
        for (int d = 0; d < Q2W_NDMA; ++d) issue_dma(d, 1, 1);
    }

    int buf = 0;
#ifdef Q2W_LOOP_SHIFT
    BN_LOOP_PLACE(8, Q2W_LOOP_SHIFT);
#endif
    for (int st = 0; st < n_stages; ++st) {
        // this stage's pieces have landed (the next stage's twelve may still be in flight)
        if (st + 1 < n_stages) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = st + 2 < n_stages;
        const int nbuf = (buf + 2 >= Q2W_NBUF) ? buf + 2 - Q2W_NBUF : buf + 2;
        const char* base = smem + buf * Q2W_STAGE;
        if (do_bias1) {                 // sum of the small tile: thread = (m, quarter of the frames)
            const char* p = base + ((tid >> 6) * 4 * 64 + (tid & 63)) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const q2x4 v = *reinterpret_cast<const q2x4*>(p + i * 1024);
                bsum += (v.x + v.y) + (v.z + v.w);
            }
        }
        if (do_bias2) {                 // sum of the big tile: thread = (frame parity, c, 4 pixels)
            const char* p = base + Q2W_SBYTES + tid * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const q2x4 v = *reinterpret_cast<const q2x4*>(p + i * 4096);
                bsum += (v.x + v.y) + (v.z + v.w);
            }
        }
        const char* ap = base + ad_a;
        int bo[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            bo[z] = buf * Q2W_STAGE + ad_b + (32 * (z >> 1) + 4 * (z & 1)) * 4;
            asm volatile("" : "+v"(bo[z]));
        }
        q2x4 sv[2];
        float bq[2][2][4];
        sv[0] = *reinterpret_cast<const q2x4*>(ap);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int z = 0; z < 4; ++z) bq[0][cb][z] = *reinterpret_cast<const float*>(smem + bo[z] + cb * 512);
#pragma unroll
        for (int t = 0; t < Q2W_KS / 2; ++t) {
            const int cu = t & 1, nx = cu ^ 1;
            if (t + 1 < Q2W_KS / 2) {
                sv[nx] = *reinterpret_cast<const q2x4*>(ap + (t + 1) * 2048);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        bq[nx][cb][z] = *reinterpret_cast<const float*>(smem + bo[z] + cb * 512 + (t + 1) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int z = 0; z < 4; ++z) {
                acc[0][z] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[cu][z], bq[cu][0][z], acc[0][z], 0, 0, 0);
                acc[1][z] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[cu][z], bq[cu][1][z], acc[1][z], 0, 0, 0);
                if (more && z < 2 && t < 6) issue_dma(2 * t + z, nbuf, st + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = (buf + 1 >= Q2W_NBUF) ? 0 : buf + 1;
    }

    // ---- quadrants -> taps, one m-block (32 small-side channels) at a time: its two waves put
    // their accumulators into P[z][m 32][col 128] (64 KB over the stage images), every thread adds
    // up the quadrants of one (m, c) pair in the fixed order z = 0..3 and puts the 25 taps into
    // R[m 32][c 8][25], which is the layout of dW: rows of 800 contiguous, 16-byte aligned bytes
    // that leave (and, when accumulating, arrive) as 16-byte runs
    float* P = reinterpret_cast<float*>(smem);
    float* R = reinterpret_cast<float*>(smem + 4 * 32 * 128 * 4);
    float* bred = reinterpret_cast<float*>(smem + 4 * 32 * 128 * 4 + 32 * 200 * 4);
    __syncthreads();
    if (do_bias1 || do_bias2) bred[tid] = bsum;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (mb == h) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int z = 0; z < 4; ++z)
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int m = (t & 3) + 8 * (t >> 2) + 4 * kh;
                        P[(z * 32 + m) * 128 + 64 * cq + 32 * cb + li] = acc[cb][z][t];
                    }
        }
        __syncthreads();
        {
            const int m = tid >> 3, c = tid & 7;
            float q[4][16];
#pragma unroll
            for (int z = 0; z < 4; ++z)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const q2x4 v = *reinterpret_cast<const q2x4*>(P + (z * 32 + m) * 128 + c * 16 + 4 * g4);
                    q[z][4 * g4 + 0] = v.x; q[z][4 * g4 + 1] = v.y; q[z][4 * g4 + 2] = v.z; q[z][4 * g4 + 3] = v.w;
                }
#pragma unroll
            for (int tap = 0; tap < 25; ++tap) {
                const int r = tap / 5, s = tap - 5 * r;
                float v = 0.f;
#pragma unroll
                for (int zp = 0; zp < 2; ++zp) {
                    const int yy = r - (zp ? 0 : 1);
                    if (yy < 0 || yy > 3) continue;
#pragma unroll
                    for (int zq = 0; zq < 2; ++zq) {
                        const int xx = s - (zq ? 0 : 1);
                        if (xx < 0 || xx > 3) continue;
                        v += q[2 * zp + zq][4 * yy + xx];
                    }
                }
                R[m * 200 + c * 25 + tap] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < 32 * 50; i += Q2_THREADS) {
            const int m = i / 50, g4 = i - 50 * m;
            q2x4 v = *reinterpret_cast<const q2x4*>(R + m * 200 + 4 * g4);
            float* dst = dw + ((size_t)(m0 + 32 * h + m) * Cb + c0) * 25 + 4 * g4;
            if (accumulate) {
                const q2x4 d = *reinterpret_cast<const q2x4*>(dst);
                v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            }
            *reinterpret_cast<q2x4*>(dst) = v;
        }
        // (the next block's P writes touch neither R nor bred; its R writes come behind a barrier
        // that every thread reaches only after its stores above)
    }
    if (do_bias1 && tid < 64) {
        const float v = ((bred[tid] + bred[64 + tid]) + bred[128 + tid]) + bred[192 + tid];
        db[m0 + tid] = accumulate ? db[m0 + tid] + v : v;
    }
    if (do_bias2 && tid < 8) {
        // threads (parity 0 / 1) x channel tid x 16 pixel groups, fixed order
        float v = 0.f;
        for (int par = 0; par < 2; ++par)
            for (int g4 = 0; g4 < 16; ++g4) v += bred[par * 128 + tid * 16 + g4];
        db[c0 + tid] = accumulate ? db[c0 + tid] + v : v;
    }
}

bool bn_qg2_wgrad_supported(const BnGeom& g) {
    static int disabled = -1;                          // BN_QG2=0: first generation (conv_qgemm.hip)
    if (disabled < 0) { const char* e = bn_tune_env("BN_QG2"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (!bn_qgemm_supported(g)) return false;
    return (g.Cs % 64) == 0 && (g.Cb % 8) == 0;
}

int bn_launch_qg2_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                        int accumulate, float* db, int bias_side, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_qg2_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Q2_MAX_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid(g.Cb / 8, g.Cs / 64);
    BN_LAUNCH_MAIN(k_qg2_wgrad, grid, dim3(Q2_THREADS), (size_t)Q2W_NBUF * Q2W_STAGE, st, small, big, dw,
                   db, g.N, g.Cs, g.Cb, accumulate, bias_side);
    BN_LAUNCH_CHECK();
    return 0;
}
