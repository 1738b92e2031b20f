// Stride == kernel (5x5, stride 5) layers between an 8x8 and a 2x2 map with offset 1: enc.conv4 and
// dec.convT0 of the default architecture at 128x128 frames, all three roles.
//
// The windows of the four small-side pixels do not overlap: small pixel (p, q) sees the 4x4 block
// (4p..4p+3, 4q..4q+3) of the 8x8 map through the taps r = y - 5p + 1, s = x - 5q + 1, and the
// fifth row / column of every window lies in the padding.  Only 16 of the 25 taps of a window
// ever meet data, so each role is four DENSE GEMMs (one per quadrant z = 2p + q) over 16-tap
// weight slices -- 64 % of the multiply-adds of the zero-padded formulation (4.3 instead of 6.7
// GFLOP per role at 256 frames), every operand element used exactly once:
//
//   down  (conv fwd, convT bwd-data)   S[n][m][z]      = sum_{c,j} B[n][c][pix(z,j)] W[m][c][tap(z,j)]
//   up    (convT fwd, conv bwd-data)   B[n][c][pix]    = sum_m     S[n][m][z]        W[m][c][tap(z,j)]
//   wgrad                              dWz[z][m][c][j] = sum_n     S[n][m][z]        B[n][c][pix(z,j)]
//
// One tiled MFMA kernel (v_mfma_f32_32x32x2_f32, 64x128 tile per workgroup of four waves, 32-deep
// stages through LDS, next stage's global loads in flight during the matrix work) serves the
// three roles; only the address functions differ.  The up role applies its epilogue in the kernel
// and scatters straight to NCHW; the down role (reduction split over workgroups) and the weight
// gradient write raw tiles to a packed scratch [split][z][M][N], and a small second kernel adds
// the splits in fixed order and applies bias / activation / derivative (weight gradient: adds the
// quadrants that share a tap, in fixed order).
#include <stdlib.h>
#include "bn_common.h"
#include "bn_fast.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define QG_T 64          // tile rows (and the column granule)
#define QG_NB 2          // column granules per workgroup tile: 64 x 128, a wave owns 32 x 64
#define QG_KS 32         // reduction depth of a stage
#define QG_LD 33         // LDS row stride: odd -> the 32 rows x 2 k of an MFMA operand read hit
                         // 64 distinct banks
enum { QG_DOWN = 0, QG_UP = 1, QG_WGRAD = 2 };

struct QGArgs {
    const float* small;
    const float* big;
    const float* w;
    float* part;
    int N, Cs, Cb;
    int M, Nc, K;        // GEMM sizes of one quadrant
    int kper;            // reduction elements per split (multiple of QG_KS)
    // up role: the epilogue is applied in the kernel and the result scattered straight to NCHW
    float* out;
    const float* bias;
    const float* dact_src;
    int act, dact;
    float slope;
};

__device__ __forceinline__ int qg_pix(int z, int j) {
    return (4 * (z >> 1) + (j >> 2)) * 8 + 4 * (z & 1) + (j & 3);
}
__device__ __forceinline__ int qg_tap(int z, int j) {
    return (((z >> 1) ? 0 : 1) + (j >> 2)) * 5 + ((z & 1) ? 0 : 1) + (j & 3);
}

typedef float floatx4a __attribute__((ext_vector_type(4)));
typedef float floatx4u __attribute__((ext_vector_type(4), aligned(4)));   // tap runs: 4-byte aligned

template <int MODE>
__global__ __launch_bounds__(256) void k_qgemm(QGArgs a) {
    __shared__ float As[QG_T * QG_LD];
    __shared__ float Bs[QG_NB * QG_T * QG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    const int z = blockIdx.z & 3, ks = blockIdx.z >> 2;
    const int i0 = blockIdx.y * QG_T, j0 = blockIdx.x * (QG_NB * QG_T);
    const int kbeg = ks * a.kper;
    const int kend = min(a.K, kbeg + a.kper);
    const int zp = z >> 1, zq = z & 1;

    // staging assignment: 8 elements of each operand tile per thread, as two 16-byte loads along
    // the contiguous direction of the tensor wherever there is one.  The up role reads the small
    // side from the z-major copy [z][n][m] written by k_qg_split_small; the weight gradient, whose
    // loads are scalar either way, reads small[n][m][z] in place.
    //   A tile As[row][k]: thread (ar, ak..ak+7);  B tile Bs[col][k]: down (br, bk..bk+7),
    //   up / wgrad: k row bk, columns br..br+7
    int ar, ak, br, bk;
    if (MODE == QG_WGRAD) { ar = tid & 63; ak = (tid >> 6) * 8; }
    else                  { ar = tid >> 2; ak = (tid & 3) * 8; }
    if (MODE == QG_DOWN)  { br = tid >> 2; bk = (tid & 3) * 8; }
    else                  { bk = tid >> 3; br = (tid & 7) * 8; }
    const bool a_ok = (i0 + ar) < a.M;

    // Global loads of a stage: buffer loads whose lane offsets are computed ONCE; a stage only adds
    // a scalar offset (every operand advances by a constant number of bytes per 32-deep stage).
    // They are issued from inside the MFMA stream of the previous stage: between two barriers, as
    // before, their ~35 address instructions and 6 loads were vector work that no wave can issue
    // while another wave of its SIMD multiplies (issue rules).
    constexpr int OOB = 0x7fffffff;
    const void* a_base = (MODE == QG_DOWN) ? (const void*)a.big : (const void*)a.small;
    const size_t a_bytes = (MODE == QG_DOWN) ? (size_t)a.N * a.Cb * 64 * 4 : (size_t)a.N * a.Cs * 4 * 4;
    const void* b_base = (MODE == QG_WGRAD) ? (const void*)a.big : (const void*)a.w;
    const size_t b_bytes = (MODE == QG_WGRAD) ? (size_t)a.N * a.Cb * 64 * 4 : (size_t)a.Cs * a.Cb * 25 * 4;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, (int)b_bytes, 0x00020000);
    int avo, bvo[QG_NB], a_step, b_step, a_e = 0;                 // bytes
    if (MODE == QG_DOWN) {
        // A = big[n][c][4x4 block of quadrant z]: k = 16 c + 4 y' + x'
        avo = (((i0 + ar) * a.Cb + (ak >> 4)) * 64 + (4 * zp + ((ak & 15) >> 2)) * 8 + 4 * zq) * 4;
        a_step = 2 * 64 * 4;
    } else if (MODE == QG_UP) {
        // z-major copy [z][n][m] written by k_qg_split_small
        avo = (z * a.N * a.Cs + (i0 + ar) * a.Cs + ak) * 4;
        a_step = QG_KS * 4;
    } else {
        // small[n][m][z] in place: rows k = frames
        avo = ((ak * a.Cs + i0 + ar) * 4 + z) * 4;
        a_e = a.Cs * 16;
        a_step = QG_KS * a.Cs * 16;
    }
    if (!a_ok) avo = OOB;
#pragma unroll
    for (int h = 0; h < QG_NB; ++h) {
        const int jc = j0 + QG_T * h + br;              // this thread's (first) column of granule h
        if (MODE == QG_DOWN)
            // B = W[m][c][16 taps of quadrant z]: two runs of four taps, five floats apart
            bvo[h] = ((jc * a.Cb + (bk >> 4)) * 25 + ((zp ? 0 : 1) + ((bk & 15) >> 2)) * 5 + (zq ? 0 : 1)) * 4;
        else if (MODE == QG_UP)
            bvo[h] = ((bk * a.Cb + (jc >> 4)) * 25 + ((zp ? 0 : 1) + ((jc & 15) >> 2)) * 5 + (zq ? 0 : 1)) * 4;
        else
            bvo[h] = ((bk * a.Cb + (jc >> 4)) * 64 + (4 * zp + ((jc & 15) >> 2)) * 8 + 4 * zq) * 4;
    }
    b_step = (MODE == QG_DOWN) ? 2 * 25 * 4 : (MODE == QG_UP ? QG_KS * a.Cb * 25 * 4 : QG_KS * a.Cb * 64 * 4);
    constexpr int B2ND = (MODE == QG_WGRAD) ? 32 : 20;  // second 16-byte run of the B operand

    float ra[8], rb[QG_NB][8];
    // stage number sidx (0 = the split's first); tail: the weight gradient's last stage may reach
    // past the last frame (K = N): lanes beyond it read 0.0f
    auto fetch = [&](const int sidx, const int k0) __attribute__((always_inline)) {
        const int sa = (MODE == QG_DOWN ? (kbeg >> 4) * 256 : (MODE == QG_UP ? kbeg * 4 : kbeg * a.Cs * 16)) + sidx * a_step;
        const int sb = (MODE == QG_DOWN ? (kbeg >> 4) * 100 : (MODE == QG_UP ? kbeg * a.Cb * 100 : kbeg * a.Cb * 256)) + sidx * b_step;
        const bool tail = (MODE == QG_WGRAD) && (k0 + QG_KS > kend);          // wave-uniform
        if (MODE == QG_WGRAD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int vo = (tail && k0 + ak + e >= kend) ? OOB : avo;
                ra[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsa, vo, sa + e * a_e, 0));
            }
        } else {
            // (whole-vector casts: extracting the components of the builtin's result one by one
            // makes this compiler emit a single dword load and splat it)
            const floatx4a v0 = __builtin_bit_cast(floatx4a, __builtin_amdgcn_raw_buffer_load_b128(rsa, avo, sa, 0));
            const floatx4a v1 = __builtin_bit_cast(floatx4a, __builtin_amdgcn_raw_buffer_load_b128(
                rsa, avo, sa + (MODE == QG_DOWN ? 32 : 16), 0));
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v0.z; ra[3] = v0.w;
            ra[4] = v1.x; ra[5] = v1.y; ra[6] = v1.z; ra[7] = v1.w;
        }
#pragma unroll
        for (int h = 0; h < QG_NB; ++h) {
            const int vo = (tail && k0 + bk >= kend) ? OOB : bvo[h];
            const floatx4a w0 = __builtin_bit_cast(floatx4a, __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, sb, 0));
            const floatx4a w1 = __builtin_bit_cast(floatx4a, __builtin_amdgcn_raw_buffer_load_b128(rsb, vo, sb + B2ND, 0));
            rb[h][0] = w0.x; rb[h][1] = w0.y; rb[h][2] = w0.z; rb[h][3] = w0.w;
            rb[h][4] = w1.x; rb[h][5] = w1.y; rb[h][6] = w1.z; rb[h][7] = w1.w;
        }
    };

    floatx16 acc[QG_NB];
#pragma unroll
    for (int h = 0; h < QG_NB; ++h)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[h][t] = 0.f;
    // wave (wv >> 1, wv & 1): rows 32 (wv >> 1).., columns 32 (wv & 1).. of EVERY granule
    const float* ap = As + ((wv >> 1) * 32 + li) * QG_LD + lk;
    const float* bp = Bs + ((wv & 1) * 32 + li) * QG_LD + lk;

    if (kbeg < kend) fetch(0, kbeg);
    int sidx = 0;
#ifdef QG_LOOP_SHIFT
    BN_LOOP_PLACE(8, QG_LOOP_SHIFT);
#endif
    for (int k0 = kbeg; k0 < kend; k0 += QG_KS, ++sidx) {
        __syncthreads();                       // everyone is done reading the previous stage
#pragma unroll
        for (int e = 0; e < 8; ++e) As[ar * QG_LD + ak + e] = ra[e];
#pragma unroll
        for (int h = 0; h < QG_NB; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (MODE == QG_DOWN) Bs[(QG_T * h + br) * QG_LD + bk + e] = rb[h][e];
                else                 Bs[(QG_T * h + br + e) * QG_LD + bk] = rb[h][e];
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < QG_KS / 2; ++t) {
            const float av = ap[2 * t];
#pragma unroll
            for (int h = 0; h < QG_NB; ++h)
                acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[QG_T * h * QG_LD + 2 * t],
                                                              acc[h], 0, 0, 0);
            if (t == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (k0 + QG_KS < kend) fetch(sidx + 1, k0 + QG_KS);   // rides in this MFMA stream
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    if (MODE == QG_UP) {
        // out_b[n][c][pix(z, j & 15)] = epi(acc + bias[c]): 16-byte runs per lane group; the other
        // half of each 32-byte sector belongs to the neighbouring quadrant's workgroup.  Loads are
        // batched ahead of the stores and everything is branch-free (buffer accesses whose lane
        // offset is out of range for rows that do not exist): a load -> wait -> store chain per
        // element drains the memory queue 32 times per lane.
        const int obytes = (int)((size_t)a.M * a.Cb * 64 * 4);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, obytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)a.dact_src, 0, a.dact_src ? obytes : 0, 0x00020000);
        const int row_bytes = a.Cb * 64 * 4;
        const int irow = i0 + (wv >> 1) * 32 + 4 * lk;                 // + (t&3) + 8*(t>>2)
        if (a.act != BN_ACT_SIGMOID && a.dact != BN_ACT_SIGMOID) {     // wave-uniform
            const float es = (a.act == BN_ACT_LRELU) ? a.slope : 1.f;   // identity = slope 1
            const float ds = (a.dact == BN_ACT_LRELU) ? a.slope : 1.f;
#pragma unroll
            for (int h = 0; h < QG_NB; ++h) {
                const int j = j0 + QG_T * h + (wv & 1) * 32 + li;
                const int c = j >> 4;
                const float bj = a.bias ? a.bias[c] : 0.f;
                const int vo = (irow * a.Cb * 64 + c * 64 + qg_pix(z, j & 15)) * 4;
                float d[16];
                if (a.dact_src) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int ro_t = (t & 3) + 8 * (t >> 2);
                        d[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rd, (irow + ro_t < a.M) ? vo : 0x7fffffff, ro_t * row_bytes, 0));
                    }
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int ro_t = (t & 3) + 8 * (t >> 2);
                    float v = acc[h][t] + bj;
                    v = v > 0.f ? v : v * es;
                    if (a.dact_src) v *= d[t] > 0.f ? 1.f : ds;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), ro,
                                                          (irow + ro_t < a.M) ? vo : 0x7fffffff,
                                                          ro_t * row_bytes, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < QG_NB; ++h) {
            const int j = j0 + QG_T * h + (wv & 1) * 32 + li;
            const int c = j >> 4;
            const float bj = a.bias ? a.bias[c] : 0.f;
            const size_t col = (size_t)c * 64 + qg_pix(z, j & 15);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int i = i0 + (wv >> 1) * 32 + (t & 3) + 8 * (t >> 2) + 4 * lk;
                if (i >= a.M) continue;
                const size_t o = (size_t)i * a.Cb * 64 + col;
                float v = bn_apply_act(acc[h][t] + bj, a.act, a.slope);
                if (a.dact_src) v *= bn_act_grad_from_output(a.dact_src[o], a.dact, a.slope);
                a.out[o] = v;
            }
        }
        return;
    }
    // raw tile -> scratch [ks][z][M][Nc]; lane holds C[(t&3) + 8*(t>>2) + 4*lk][li]
    float* dst = a.part + ((size_t)(ks * 4 + z) * a.M) * a.Nc;
#pragma unroll
    for (int h = 0; h < QG_NB; ++h) {
        const int j = j0 + QG_T * h + (wv & 1) * 32 + li;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + (wv >> 1) * 32 + (t & 3) + 8 * (t >> 2) + 4 * lk;
            if (i < a.M) dst[(size_t)i * a.Nc + j] = acc[h][t];
        }
    }
}

// small[n][m][z] -> z-major copy [z][n][m] (the A operand of the up role is then contiguous along
// its reduction index)
__global__ __launch_bounds__(256) void k_qg_split_small(const float* __restrict__ small,
                                                        float* __restrict__ dst, size_t nm) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nm) return;
    const floatx4a v = *reinterpret_cast<const floatx4a*>(small + 4 * idx);
    dst[idx] = v.x; dst[nm + idx] = v.y; dst[2 * nm + idx] = v.z; dst[3 * nm + idx] = v.w;
}

// ---------------------------------------------------------------------------------------------
// second pass
// ---------------------------------------------------------------------------------------------
// out_s[n][m][z] = epi( sum_split P[split][z][n][m] + bias[m] )
__global__ __launch_bounds__(256) void k_qg_finish_down(
    const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ out,
    const float* __restrict__ dact_src, int N, int Cs, int splits, int act, int dact,
    float slope) {
    const size_t total = (size_t)N * Cs * 4;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int z = (int)(idx & 3);
    const size_t nm = idx >> 2;                       // n * Cs + m
    const size_t plane = (size_t)N * Cs;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[(size_t)(s * 4 + z) * plane + nm];
    if (bias) v += bias[nm % Cs];
    v = bn_apply_act(v, act, slope);
    if (dact_src) v *= bn_act_grad_from_output(dact_src[idx], dact, slope);
    out[idx] = v;
}

// dW[m][c][r][s] (+)= sum over the quadrants whose window has tap (r, s), fixed order.
// A thread owns one (m, c): it reads its four 16-float quadrant blocks with 16-byte loads (rows of
// consecutive threads are adjacent), the 25 results of a block of 256 pairs go through LDS so that
// the weight-gradient tensor is read (accumulate) and written in 16-byte runs.
__global__ __launch_bounds__(256) void k_qg_finish_wgrad(
    const float* __restrict__ part, float* __restrict__ dw, int Cs, int Cb, int accumulate) {
    __shared__ __attribute__((aligned(16))) float stage[256 * 25];
    const size_t npair = (size_t)Cs * Cb;
    const size_t mc = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t plane = npair * 16;
    if (mc < npair) {
        float q[4][16];
#pragma unroll
        for (int zq = 0; zq < 4; ++zq)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const floatx4a v = *reinterpret_cast<const floatx4a*>(part + zq * plane + mc * 16 + 4 * g4);
                q[zq][4 * g4 + 0] = v.x; q[zq][4 * g4 + 1] = v.y; q[zq][4 * g4 + 2] = v.z; q[zq][4 * g4 + 3] = v.w;
            }
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            const int r = tap / 5, s = tap - 5 * r;
            float v = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int yy = r - (p ? 0 : 1);
                if (yy < 0 || yy > 3) continue;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const int xx = s - (qq ? 0 : 1);
                    if (xx < 0 || xx > 3) continue;
                    v += q[2 * p + qq][4 * yy + xx];
                }
            }
            stage[threadIdx.x * 25 + tap] = v;
        }
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 256 * 25;           // multiple of 4 floats
    const size_t total = npair * 25;
    for (int i = threadIdx.x; i < 256 * 25 / 4; i += 256) {
        const size_t o = base + 4 * (size_t)i;
        if (o + 3 < total) {
            floatx4a v = *reinterpret_cast<const floatx4a*>(stage + 4 * i);
            if (accumulate) {
                const floatx4a d = *reinterpret_cast<const floatx4a*>(dw + o);
                v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            }
            *reinterpret_cast<floatx4a*>(dw + o) = v;
        } else {
            for (int e = 0; e < 4; ++e)
                if (o + e < total) dw[o + e] = accumulate ? dw[o + e] + stage[4 * i + e] : stage[4 * i + e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool bn_qgemm_supported(const BnGeom& g) {
    static int disabled = -1;                          // BN_QGEMM=0: previous s5 kernels
    if (disabled < 0) { const char* e = bn_tune_env("BN_QGEMM"); disabled = (e && e[0] == '0') ? 1 : 0; }
    if (disabled) return false;
    if (g.R != 5 || g.S != 5 || g.stride != 5) return false;
    if (g.Hs != 2 || g.Ws != 2 || g.Hb != 8 || g.Wb != 8 || g.pt != 1 || g.pl != 1) return false;
    if ((g.Cs % (QG_NB * QG_T)) != 0 || (g.Cb % (4 * QG_NB)) != 0) return false;
    if ((size_t)g.N * g.Cb * 64 * 4 >= 0x7fffffffull) return false;
    return true;
}

// reduction splits of the down role: a function of the channel counts only, NOT of the number of
// frames -- a frame's result then does not depend on the batch it is in (tile rows are independent),
// which keeps whole-batch and chunked passes bit-identical through these layers
static int qg_down_splits(const BnGeom& g) {
    static int env = -1;                             // tuning hook: BN_QG_SPLITS
    if (env < 0) { const char* e = bn_tune_env("BN_QG_SPLITS"); env = e ? atoi(e) : 0; }
    if (env > 0) return env;
    if ((g.Cs / QG_T) * 4 >= 64) return 1;
    int s = g.Cb * 16 >= 4096 ? 4 : (g.Cb * 16 >= 2048 ? 2 : 1);
    // small batches (round 4; a 32-frame shard of a trial ran 64 workgroups of 40 us): more slices
    // until ~256 workgroups, 256 reduction elements per slice at least.  Batches of more than 64
    // frames keep the channel-count rule, so the 200 / 56 / 256-frame passes of the benchmark
    // are unaffected beyond the 56-frame chunk.
    if (g.N <= 64) {
        const int tiles = (g.Cs / (QG_NB * QG_T)) * ((g.N + QG_T - 1) / QG_T) * 4;
        while (tiles * s < 256 && s < 16 && (g.Cb * 16) / (2 * s) >= 256) s *= 2;
    }
    return s;
}

size_t bn_qgemm_ws_bytes(int role, const BnGeom& g) {
    if (role == QG_DOWN) return (size_t)qg_down_splits(g) * 4 * g.N * g.Cs * sizeof(float);
    if (role == QG_UP) return (size_t)4 * g.N * g.Cs * sizeof(float);
    return (size_t)4 * g.Cs * g.Cb * 16 * sizeof(float);
}

int bn_launch_qgemm_down(const float* big, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         void* ws, hipStream_t st) {
    const int splits = qg_down_splits(g);
    QGArgs a = {nullptr, big, w, (float*)ws, g.N, g.Cs, g.Cb, g.N, g.Cs, g.Cb * 16, 0,
                nullptr, nullptr, nullptr, 0, 0, 0.f};
    a.kper = ((a.K / splits + QG_KS - 1) / QG_KS) * QG_KS;
    const dim3 grid(a.Nc / (QG_NB * QG_T), (a.M + QG_T - 1) / QG_T, 4 * splits);
    BN_LAUNCH_MAIN(k_qgemm<QG_DOWN>, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    const size_t total = (size_t)g.N * g.Cs * 4;
    hipLaunchKernelGGL(k_qg_finish_down, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, bias, out, dact_src, g.N, g.Cs, splits, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_qgemm_up(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       void* ws, hipStream_t st) {
    float* zsmall = (float*)ws;
    const size_t nm = (size_t)g.N * g.Cs;
    hipLaunchKernelGGL(k_qg_split_small, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, st, small,
                       zsmall, nm);
    BN_LAUNCH_CHECK();
    QGArgs a = {zsmall, nullptr, w, nullptr, g.N, g.Cs, g.Cb, g.N, g.Cb * 16, g.Cs, 0,
                out, bias, dact_src, act, dact, slope};
    a.kper = ((a.K + QG_KS - 1) / QG_KS) * QG_KS;
    const dim3 grid(a.Nc / (QG_NB * QG_T), (a.M + QG_T - 1) / QG_T, 4);
    BN_LAUNCH_MAIN(k_qgemm<QG_UP>, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_qgemm_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                          int accumulate, void* ws, hipStream_t st) {
    QGArgs a = {small, big, nullptr, (float*)ws, g.N, g.Cs, g.Cb, g.Cs, g.Cb * 16, g.N, 0,
                nullptr, nullptr, nullptr, 0, 0, 0.f};
    a.kper = ((a.K + QG_KS - 1) / QG_KS) * QG_KS;
    const dim3 grid(a.Nc / (QG_NB * QG_T), (a.M + QG_T - 1) / QG_T, 4);
    BN_LAUNCH_MAIN(k_qgemm<QG_WGRAD>, grid, dim3(256), 0, st, a);
    BN_LAUNCH_CHECK();
    const size_t npair = (size_t)g.Cs * g.Cb;
    hipLaunchKernelGGL(k_qg_finish_wgrad, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, st,
                       (const float*)ws, dw, g.Cs, g.Cb, accumulate);
    BN_LAUNCH_CHECK();
    return 0;
}
