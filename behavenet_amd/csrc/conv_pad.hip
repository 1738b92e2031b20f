// Geometries the specialised kernels were not written for, served BY them:
//
//  * 5x5 stride-2 layers whose small-side map is no power of two (the tiles of the MFMA families
//    need power-of-two rows and columns): both tensors are embedded, top-left, in zero-padded
//    copies of the next power-of-two size.  TF-"same" convolutions and cropped transposed
//    convolutions read zeros outside the frame, the padding supplies exactly those zeros, so every
//    element of the original region is unchanged (gather-down and gather-up: the result is cropped
//    back; weight gradient: padded positions add 0).  Costs one copy in and one copy out per call
//    and (padded area / area) of the arithmetic -- against direct loops that ran 50-100x slower
//    (the reference's integration-test shape 1x64x48: 42.5 ms per training step, 0.015 of peak).
//  * 5x5 stride-5 gather-down onto maps of a few pixels other than the benchmark's 2x2 (enc.conv4 /
//    dec.convT0 data gradient on other frame sizes): the windows do not overlap, every window is a
//    row of an im2col matrix and the layer a dense GEMM on the matrix cores.
#include "bn_common.h"
#include "bn_launch.h"

#define PD_THREADS 256

// dst (planes, Hp, Wp) <- src (planes, H, W) at row / column offset (oh, ow), zeros elsewhere.
// (The offset turns a layer whose first tap sits pt = 2 pixels outside the frame into the pt = 1
// form the specialised kernels are written for: big'[k] = big[k - (pt - 1)].)
// Every tensor here is below 2 GB (the kernels around these copies address them through buffer
// descriptors), so all index arithmetic is 32-bit; the padded side moves in 16-byte groups (Wp is
// a multiple of 4), and so does the cropped side when its rows are.
__global__ __launch_bounds__(PD_THREADS) void k_pad2d(const float* __restrict__ src,
                                                       float* __restrict__ dst, unsigned planes, int H,
                                                       int W, int Hp, int Wp, int oh, int ow) {
    const unsigned wq = Wp >> 2, groups = Hp * wq, total = planes * groups;
    float4* dp = reinterpret_cast<float4*>(dst);
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pl = q / groups, e = q - pl * groups;
        const int hp = e / wq, w0 = 4 * (int)(e - hp * wq) - ow;
        const int h = hp - oh;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H) {
            const float* row = src + (pl * H + h) * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (w0 + k >= 0 && w0 + k < W) v[k] = row[w0 + k];
        }
        dp[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// dst (planes, H, W) <- src (planes, Hp, Wp) from offset (oh, ow), times act'(dact_src) when given;
// VEC = 4: rows of the destination are multiples of four floats
template <int VEC>
__global__ __launch_bounds__(PD_THREADS) void k_crop2d(const float* __restrict__ src,
                                                        float* __restrict__ dst, unsigned planes, int H,
                                                        int W, int Hp, int Wp, int oh, int ow,
                                                        const float* __restrict__ dact_src, int dact,
                                                        float slope) {
    const unsigned wq = W / VEC, groups = H * wq, total = planes * groups;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pl = q / groups, e = q - pl * groups;
        const int h = e / wq, w0 = VEC * (int)(e - h * wq);
        const float* row = src + (pl * Hp + h + oh) * Wp + w0 + ow;
        float v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = row[k];
        if (dact_src) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                v[k] *= bn_act_grad_from_output(dact_src[(size_t)q * VEC + k], dact, slope);
        }
        if (VEC == 4) reinterpret_cast<float4*>(dst)[q] = make_float4(v[0], v[1], v[2], v[VEC - 1]);
        else dst[q] = v[0];
    }
}

static int pd_blocks(size_t n) {
    const size_t b = (n + PD_THREADS - 1) / PD_THREADS;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

int bn_launch_pad2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                    int oh, int ow, hipStream_t st) {
    if (Wp & 3) return BN_E_BADARG;
    hipLaunchKernelGGL(k_pad2d, dim3(pd_blocks(planes * Hp * (Wp >> 2))), dim3(PD_THREADS), 0, st, src,
                       dst, (unsigned)planes, H, W, Hp, Wp, oh, ow);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_crop2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                     int oh, int ow, const float* dact_src, int dact, float slope, hipStream_t st) {
    if ((W & 3) == 0 && ((((uintptr_t)dst) | ((uintptr_t)dact_src)) & 15u) == 0)
        hipLaunchKernelGGL(k_crop2d<4>, dim3(pd_blocks(planes * H * (W >> 2))), dim3(PD_THREADS), 0, st,
                           src, dst, (unsigned)planes, H, W, Hp, Wp, oh, ow, dact_src, dact, slope);
    else
        hipLaunchKernelGGL(k_crop2d<1>, dim3(pd_blocks(planes * H * W)), dim3(PD_THREADS), 0, st, src,
                           dst, (unsigned)planes, H, W, Hp, Wp, oh, ow, dact_src, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Kernels smaller than 5x5 (ae_arch_2.json: 4x4) on the 5x5 stride-2 families: the weights are
// embedded top-left in 5x5 taps, the added taps are zero.  Same index relation (small pixel p meets
// big row 2p - pt + r), so every role is exact; the weight gradient of the added taps is dropped.
// ---------------------------------------------------------------------------------------------
// w5[pair][5][5] <- w[pair][R][S]
// (dr, ds): where tap (0, 0) sits inside the 5x5 taps.  Small pixel p meets big row st p + r - pt = st p + (r + dr) -
// (pt + dr): a 3x3 stride-2 layer with TF-"same" padding (first tap ON the frame, pt = pl = 0) embedded at (1, 1) is a
// 5x5 layer with pt = pl = 1 -- the offsets the streamlined stride-2 families are built for (round 4: it ran on the
// first-generation kernels and the col2im detour, 8.0 ms/step for 36 % of the default architecture's arithmetic).
__global__ __launch_bounds__(PD_THREADS) void k_pad_taps(const float* __restrict__ w,
                                                          float* __restrict__ w5, unsigned pairs,
                                                          int R, int S, int dr, int ds) {
    const unsigned total = pairs * 25;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pr = q / 25, t = q - pr * 25;
        const int r = (int)(t / 5) - dr, c = (int)(t - 5 * (t / 5)) - ds;
        w5[q] = (r >= 0 && r < R && c >= 0 && c < S) ? w[pr * (R * S) + r * S + c] : 0.f;
    }
}
// dw[pair][R][S] (+)= dw5[pair][5][5];  db[i] (+)= db5[i] (the bias gradient the inner kernel left in
// scratch: it writes, the caller accumulates)
__global__ __launch_bounds__(PD_THREADS) void k_crop_taps(const float* __restrict__ dw5,
                                                           float* __restrict__ dw, unsigned pairs,
                                                           int R, int S, int accumulate,
                                                           const float* __restrict__ db5,
                                                           float* __restrict__ db, unsigned nb, int dr, int ds) {
    const unsigned total = pairs * R * S;
    if (db5 && blockIdx.x == 0)
        for (unsigned i = threadIdx.x; i < nb; i += PD_THREADS) db[i] = accumulate ? db[i] + db5[i] : db5[i];
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pr = q / (R * S), t = q - pr * (R * S);
        const int r = t / S, c = t - S * r;
        const float v = dw5[pr * 25 + (r + dr) * 5 + c + ds];
        dw[q] = accumulate ? dw[q] + v : v;
    }
}
int bn_launch_pad_taps(const float* w, float* w5, size_t pairs, int R, int S, hipStream_t st, int dr, int ds) {
    hipLaunchKernelGGL(k_pad_taps, dim3(pd_blocks(pairs * 25)), dim3(PD_THREADS), 0, st, w, w5,
                       (unsigned)pairs, R, S, dr, ds);
    BN_LAUNCH_CHECK();
    return 0;
}
// the same for several layers in ONE launch (round 6): a fused conv stack pads the taps of all its small-kernel layers
// when its forward pass starts and hands the copies to the forward and data-gradient entry points (bn_conv_taps_hint);
// k_pad_taps ran once per layer and role, 19 launches of 5-6 us in a step of ae_arch_2.json
__global__ __launch_bounds__(PD_THREADS) void k_pad_taps_jobs(BnPadTapsJobs p) {
    int b = blockIdx.x;
    for (int j = 0; j < p.n; ++j) {
        const BnPadTapsJob& jb = p.job[j];
        if (b < jb.blocks) {
            const unsigned total = jb.pairs * 25;
            for (unsigned q = b * PD_THREADS + threadIdx.x; q < total; q += jb.blocks * PD_THREADS) {
                const unsigned pr = q / 25, t = q - pr * 25;
                const int r = (int)(t / 5) - jb.dr, c = (int)(t - 5 * (t / 5)) - jb.ds;
                jb.w5[q] = (r >= 0 && r < jb.R && c >= 0 && c < jb.S) ? jb.w[pr * (jb.R * jb.S) + r * jb.S + c] : 0.f;
            }
            return;
        }
        b -= jb.blocks;
    }
}
int bn_launch_pad_taps_jobs(BnPadTapsJobs* p, hipStream_t st) {
    int blocks = 0;
    for (int j = 0; j < p->n; ++j) {
        const int b = pd_blocks((size_t)p->job[j].pairs * 25);
        p->job[j].blocks = b;
        blocks += b;
    }
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(k_pad_taps_jobs, dim3(blocks), dim3(PD_THREADS), 0, st, *p);
    BN_LAUNCH_CHECK();
    return 0;
}
// ---------------------------------------------------------------------------------------------
// Stride-1 gather-up as a gather-down (round 4, no im2col / col2im): with stride 1
//   out[n,m,h,w] = sum_{c,r,s} small[n,c,h+pt-r,w+pl-s] W[c][m][r][s]
//                = sum_{c,r',s'} small[n,c,h+r'-(R-1-pt),w+s'-(S-1-pl)] W'[m][c][r'][s']
// with W'[m][c][r'][s'] = W[c][m][R-1-r'][S-1-s']: the channel roles swapped, the taps reversed.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PD_THREADS) void k_flip_taps(const float* __restrict__ w, float* __restrict__ wf,
                                                           unsigned Cs, unsigned Cb, unsigned RS) {
    const unsigned total = Cs * Cb * RS;
    for (unsigned e = blockIdx.x * PD_THREADS + threadIdx.x; e < total; e += gridDim.x * PD_THREADS) {
        const unsigned t = e % RS, mc = e / RS;
        const unsigned c = mc % Cs, m = mc / Cs;
        wf[e] = w[(c * Cb + m) * RS + (RS - 1 - t)];
    }
}
int bn_launch_flip_taps(const float* w, float* wf, int Cs, int Cb, int RS, hipStream_t st) {
    hipLaunchKernelGGL(k_flip_taps, dim3(pd_blocks((size_t)Cs * Cb * RS)), dim3(PD_THREADS), 0, st, w, wf,
                       (unsigned)Cs, (unsigned)Cb, (unsigned)RS);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_crop_taps(const float* dw5, float* dw, size_t pairs, int R, int S, int accumulate,
                        hipStream_t st, const float* db5, float* db, int nb, int dr, int ds) {
    hipLaunchKernelGGL(k_crop_taps, dim3(pd_blocks(pairs * R * S)), dim3(PD_THREADS), 0, st, dw5, dw,
                       (unsigned)pairs, R, S, accumulate, db5, db, (unsigned)nb, dr, ds);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Kernels LARGER than 5x5 with stride 2 (7x7, 9x9: the reference's architecture search draws 3 / 5 / 7 / 9 with equal
// probability) on the stride-1 5x5 kernels, without im2col / col2im: big pixel (2i + rho, 2j + sig) only meets the taps
// of one parity per axis -- at most 5 x 5 of them for kernels up to 10 x 10 -- and meets them at stride 1 on the
// small map.  With the big map cut into its four phases X[(rho, sig) C + c][i][j] = x[c][2i + rho][2j + sig]
// (k_space_to_depth; zero padding of X = zero padding of x), every role of the layer is the same role of ONE stride-1
// 5x5 layer between the small map and the 4 Cb phase channels:
//   gather-down      out[m][p] = sum_{(rho sig) c, u'} X[..][p - pt1 + u'] W1[m][(rho sig) c][u']
//   weight gradient  dW1[m][(rho sig) c][u'] -> dW (k_bigk_phase_unpack)
//   gather-up        Y[(rho sig) c][i] = sum_{m, u'} small[m][i - pt1 + u'] W1'[(rho sig) c][m][u'], interleaved
//                    by k_depth_to_space, which also carries the epilogue (bias, activation, mask of the layer below)
// 7x7: 16 / 12 / 12 / 9 taps in the four phases, 9x9: 25 / 20 / 20 / 16 -- of 25 multiplied each.
// ---------------------------------------------------------------------------------------------
// X[n][(2 rho + sig) C + c][i][j] = x[n][c][2i + rho][2j + sig]; one 16-byte read, two 8-byte writes per thread
__global__ __launch_bounds__(PD_THREADS) void k_space_to_depth(const float* __restrict__ x, float* __restrict__ X,
                                                                unsigned N, unsigned C, int Hy, int Wy) {
    const int Hb = 2 * Hy;
    const unsigned wq = Wy / 2, total = N * C * Hb * wq;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned w4 = q % wq, t = q / wq;
        const int h = t % Hb;
        const unsigned nc = t / Hb, c = nc % C, n = nc / C;
        const int rho = h & 1, i = h >> 1;
        const float4 v = reinterpret_cast<const float4*>(x)[q];
        float* p0 = X + (((size_t)(n * 4 + 2 * rho) * C + c) * Hy + i) * Wy + 2 * w4;
        *reinterpret_cast<float2*>(p0) = make_float2(v.x, v.z);
        *reinterpret_cast<float2*>(p0 + (size_t)C * Hy * Wy) = make_float2(v.y, v.w);
    }
}
int bn_launch_space_to_depth(const float* x, float* X, int N, int C, int Hy, int Wy, hipStream_t st) {
    const size_t total = (size_t)N * C * 4 * Hy * Wy;
    if (total >= (1ull << 32) || (Wy & 1)) return BN_E_SHAPE;
    hipLaunchKernelGGL(k_space_to_depth, dim3(pd_blocks(total / 4)), dim3(PD_THREADS), 0, st, x, X, (unsigned)N,
                       (unsigned)C, Hy, Wy);
    BN_LAUNCH_CHECK();
    return 0;
}
// tap (u', v') of phase (rho, sig) is tap (2u + kr[rho], 2v + kc[sig]) of the layer, u = sgn u' + ofr[rho] (sgn = +1:
// the phases are the layer's big side -- gather-down, weight gradient; -1: gather-up, the taps reversed), 0.0f where
// that tap does not exist.
//   phase_out = 0: w1[m][(2 rho + sig) Cb + c][u'][v']     1: w1[(2 rho + sig) Cb + c][m][u'][v']
struct BigKTaps { int kr[2], ofr[2], kc[2], ofc[2], sgn; };
__global__ __launch_bounds__(PD_THREADS) void k_bigk_phase_pack(const float* __restrict__ w, float* __restrict__ w1,
                                                                 unsigned Cs, unsigned Cb, int R, int S,
                                                                 BigKTaps k, int phase_out) {
    const unsigned total = 4 * Cb * Cs * 25;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned t = q % 25;
        unsigned m, pc;
        if (phase_out) { m = (q / 25) % Cs; pc = q / (25 * Cs); }
        else { pc = (q / 25) % (4 * Cb); m = q / (25 * 4 * Cb); }
        const unsigned c = pc % Cb, ph = pc / Cb;
        const int rho = ph >> 1, sig = ph & 1;
        const int u = k.sgn * (int)(t / 5) + k.ofr[rho], v = k.sgn * (int)(t % 5) + k.ofc[sig];
        const int r = 2 * u + k.kr[rho], sx = 2 * v + k.kc[sig];
        w1[q] = (u >= 0 && v >= 0 && r < R && sx < S) ? w[((size_t)(m * Cb + c) * R + r) * S + sx] : 0.f;
    }
}
static BigKTaps bigk_taps(const int* kr, const int* ofr, const int* kc, const int* ofc, int sgn) {
    BigKTaps k;
    for (int i = 0; i < 2; ++i) { k.kr[i] = kr[i]; k.ofr[i] = ofr[i]; k.kc[i] = kc[i]; k.ofc[i] = ofc[i]; }
    k.sgn = sgn;
    return k;
}
int bn_launch_bigk_phase_pack(const float* w, float* w1, int Cs, int Cb, int R, int S, const int* kr, const int* ofr,
                              const int* kc, const int* ofc, int sgn, int phase_out, hipStream_t st) {
    hipLaunchKernelGGL(k_bigk_phase_pack, dim3(pd_blocks((size_t)4 * Cb * Cs * 25)), dim3(PD_THREADS), 0, st, w, w1,
                       (unsigned)Cs, (unsigned)Cb, R, S, bigk_taps(kr, ofr, kc, ofc, sgn), phase_out);
    BN_LAUNCH_CHECK();
    return 0;
}
// dw[m][c][r][s] (+)= dw1[m][(2 rho + sig) Cb + c][u'][v'] of the phase that holds tap (r, s);  db[i] (+)= db5[i] as
// in k_crop_taps
__global__ __launch_bounds__(PD_THREADS) void k_bigk_phase_unpack(const float* __restrict__ dw1,
                                                                   float* __restrict__ dw, unsigned Cs, unsigned Cb,
                                                                   int R, int S, BigKTaps k, int accumulate,
                                                                   const float* __restrict__ db5,
                                                                   float* __restrict__ db, unsigned nb) {
    const unsigned total = Cs * Cb * R * S;
    if (db5 && blockIdx.x == 0)
        for (unsigned i = threadIdx.x; i < nb; i += PD_THREADS) db[i] = accumulate ? db[i] + db5[i] : db5[i];
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned t = q % (R * S), mc = q / (R * S);
        const unsigned c = mc % Cb, m = mc / Cb;
        const int r = t / S, sx = t - S * r;
        const int rho = (k.kr[0] == (r & 1)) ? 0 : 1, sig = (k.kc[0] == (sx & 1)) ? 0 : 1;
        const int up = (r >> 1) - k.ofr[rho], vp = (sx >> 1) - k.ofc[sig];        // sgn = +1
        const float v = dw1[((size_t)(m * 4 + 2 * rho + sig) * Cb + c) * 25 + up * 5 + vp];
        dw[q] = accumulate ? dw[q] + v : v;
    }
}
int bn_launch_bigk_phase_unpack(const float* dw1, float* dw, int Cs, int Cb, int R, int S, const int* kr,
                                const int* ofr, const int* kc, const int* ofc, int accumulate, const float* db5,
                                float* db, int nb, hipStream_t st) {
    hipLaunchKernelGGL(k_bigk_phase_unpack, dim3(pd_blocks((size_t)Cs * Cb * R * S)), dim3(PD_THREADS), 0, st, dw1,
                       dw, (unsigned)Cs, (unsigned)Cb, R, S, bigk_taps(kr, ofr, kc, ofc, 1), accumulate, db5, db,
                       (unsigned)nb);
    BN_LAUNCH_CHECK();
    return 0;
}
// ---------------------------------------------------------------------------------------------
// ... and with stride 1 (under max pooling the search draws the same kernel sizes): the taps are cut into 2 x 2 blocks
// of at most 5 x 5 (rows [0, L0r) and [L0r, R), columns likewise); small pixel p meets big row p - pt + r0 + r' = row
// p + r' of the big map SHIFTED by r0 - pt rows.  The four shifted copies, each (Hs + 4) x (Ws + 4) so that every row a
// 5-tap window can reach is IN the copy (k_shift_cat writes 0.0f where the frame ends), concatenated along the channel
// axis, make the layer ONE 5x5 layer without padding over 4 Cb channels (weights k_bigk_pack / k_bigk_unpack).  The
// gather-up role is the gather-down of the flipped layer (k_flip_taps above).
// ---------------------------------------------------------------------------------------------
// y[n][b C + c][h][w] = x[n][c][h + dr[b >> 1]][w + dc[b & 1]], h < Ho, w < Wo (0.0f off the frame)
__global__ __launch_bounds__(PD_THREADS) void k_shift_cat(const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned N, unsigned C, int H, int W, int Ho, int Wo,
                                                           int dr0, int dr1, int dc0, int dc1) {
    const unsigned wq = Wo / 4, total = N * 4 * C * Ho * wq;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned w4 = q % wq, t = q / wq;
        const int h = t % Ho;
        const unsigned pc = t / Ho;                      // n * 4C + b * C + c
        const unsigned c = pc % C, b = (pc / C) & 3, n = pc / (4 * C);
        const int hs = h + ((b >> 1) ? dr1 : dr0), w0 = 4 * (int)w4 + ((b & 1) ? dc1 : dc0);
        const float* row = x + ((size_t)(n * C + c) * H + hs) * W;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = (hs >= 0 && hs < H && w0 + k >= 0 && w0 + k < W) ? row[w0 + k] : 0.f;
        reinterpret_cast<float4*>(y)[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
int bn_launch_shift_cat(const float* x, float* y, int N, int C, int H, int W, int Ho, int Wo, int dr0, int dr1,
                        int dc0, int dc1, hipStream_t st) {
    const size_t total = (size_t)N * 4 * C * Ho * Wo;
    if (total >= (1ull << 32) || (Wo & 3)) return BN_E_SHAPE;
    hipLaunchKernelGGL(k_shift_cat, dim3(pd_blocks(total / 4)), dim3(PD_THREADS), 0, st, x, y, (unsigned)N,
                       (unsigned)C, H, W, Ho, Wo, dr0, dr1, dc0, dc1);
    BN_LAUNCH_CHECK();
    return 0;
}
// w5[m][b Cb + c][r'][s'] = w[m][c][r0(b) + r'][s0(b) + s'] inside block b, 0.0f past it
__global__ __launch_bounds__(PD_THREADS) void k_bigk_pack(const float* __restrict__ w, float* __restrict__ w5,
                                                           unsigned Cs, unsigned Cb, int R, int S, int L0r,
                                                           int L0c) {
    const unsigned total = Cs * 4 * Cb * 25;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned t = q % 25, bc = (q / 25) % (4 * Cb), m = q / (25 * 4 * Cb);
        const unsigned b = bc / Cb, c = bc - b * Cb;
        const int r = (int)(t / 5) + ((b >> 1) ? L0r : 0), sx = (int)(t % 5) + ((b & 1) ? L0c : 0);
        const int rl = (b >> 1) ? R : L0r, sl = (b & 1) ? S : L0c;
        w5[q] = (r < rl && sx < sl) ? w[((size_t)(m * Cb + c) * R + r) * S + sx] : 0.f;
    }
}
int bn_launch_bigk_pack(const float* w, float* w5, int Cs, int Cb, int R, int S, int L0r, int L0c, hipStream_t st) {
    hipLaunchKernelGGL(k_bigk_pack, dim3(pd_blocks((size_t)Cs * 4 * Cb * 25)), dim3(PD_THREADS), 0, st, w, w5,
                       (unsigned)Cs, (unsigned)Cb, R, S, L0r, L0c);
    BN_LAUNCH_CHECK();
    return 0;
}
// dw[m][c][r][s] (+)= dw5[m][b(r, s) Cb + c][r - r0][s - s0];  db[i] (+)= db5[i] as in k_crop_taps
__global__ __launch_bounds__(PD_THREADS) void k_bigk_unpack(const float* __restrict__ dw5, float* __restrict__ dw,
                                                             unsigned Cs, unsigned Cb, int R, int S, int L0r,
                                                             int L0c, int accumulate,
                                                             const float* __restrict__ db5, float* __restrict__ db,
                                                             unsigned nb) {
    const unsigned total = Cs * Cb * R * S;
    if (db5 && blockIdx.x == 0)
        for (unsigned i = threadIdx.x; i < nb; i += PD_THREADS) db[i] = accumulate ? db[i] + db5[i] : db5[i];
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned t = q % (R * S), mc = q / (R * S);
        const unsigned c = mc % Cb, m = mc / Cb;
        const int r = t / S, sx = t - S * r;
        const int br = r >= L0r, bs = sx >= L0c;
        const float v = dw5[((size_t)(m * 4 + 2 * br + bs) * Cb + c) * 25 + (r - (br ? L0r : 0)) * 5 +
                            (sx - (bs ? L0c : 0))];
        dw[q] = accumulate ? dw[q] + v : v;
    }
}
int bn_launch_bigk_unpack(const float* dw5, float* dw, int Cs, int Cb, int R, int S, int L0r, int L0c,
                          int accumulate, const float* db5, float* db, int nb, hipStream_t st) {
    hipLaunchKernelGGL(k_bigk_unpack, dim3(pd_blocks((size_t)Cs * Cb * R * S)), dim3(PD_THREADS), 0, st, dw5, dw,
                       (unsigned)Cs, (unsigned)Cb, R, S, L0r, L0c, accumulate, db5, db, (unsigned)nb);
    BN_LAUNCH_CHECK();
    return 0;
}
// out[n][c][2i + rho][2j + sig] = epilogue(y[n][(2 rho + sig) C + c][i][j] + bias[c]); four output columns per thread
__global__ __launch_bounds__(PD_THREADS) void k_depth_to_space(const float* __restrict__ y, float* __restrict__ out,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ dact_src, unsigned N,
                                                                unsigned C, int Hy, int Wy, int act, int dact,
                                                                float slope) {
    const int Hb = 2 * Hy, Wb = 2 * Wy;
    const unsigned wq = Wb / 4, total = N * C * Hb * wq;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned w4 = q % wq, t = q / wq;
        const int h = t % Hb;
        const unsigned nc = t / Hb, c = nc % C, n = nc / C;
        const int rho = h & 1, i = h >> 1;
        const float* p0 = y + (((size_t)(n * 4 + 2 * rho) * C + c) * Hy + i) * Wy + 2 * w4;
        const float* p1 = p0 + (size_t)C * Hy * Wy;
        const float2 a = *reinterpret_cast<const float2*>(p0), b = *reinterpret_cast<const float2*>(p1);
        const float bz = bias ? bias[c] : 0.f;
        float v[4] = {a.x + bz, b.x + bz, a.y + bz, b.y + bz};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = bn_apply_act(v[k], act, slope);
        if (dact_src) {
            const float4 d = reinterpret_cast<const float4*>(dact_src)[q];
            v[0] *= bn_act_grad_from_output(d.x, dact, slope);
            v[1] *= bn_act_grad_from_output(d.y, dact, slope);
            v[2] *= bn_act_grad_from_output(d.z, dact, slope);
            v[3] *= bn_act_grad_from_output(d.w, dact, slope);
        }
        reinterpret_cast<float4*>(out)[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
int bn_launch_depth_to_space(const float* y, float* out, const float* bias, const float* dact_src, int N, int C,
                             int Hy, int Wy, int act, int dact, float slope, hipStream_t st) {
    const size_t total = (size_t)N * C * 4 * Hy * Wy;
    if (total >= (1ull << 32) || (Wy & 1)) return BN_E_SHAPE;
    hipLaunchKernelGGL(k_depth_to_space, dim3(pd_blocks(total / 4)), dim3(PD_THREADS), 0, st, y, out, bias, dact_src,
                       (unsigned)N, (unsigned)C, Hy, Wy, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Channel groups: the single-channel edge kernels take up to 32 channels on their many-channel
// side; a layer with more (ae_arch_2.json: 1 -> 64) is run group by group on contiguous copies
// (gather-down and weight gradient are independent per small-side channel).
//   dst[n][c0d + c][i] <- src[n][c0s + c][i] (times act'(dact_src at dst) when given), c < Cg
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PD_THREADS) void k_chan_copy(const float* __restrict__ src,
                                                           float* __restrict__ dst, unsigned N,
                                                           unsigned Csrc, unsigned c0s, unsigned Cdst,
                                                           unsigned c0d, unsigned Cg, unsigned HW4,
                                                           const float* __restrict__ dact_src, int dact,
                                                           float slope) {
    const unsigned total = N * Cg * HW4;                    // 16-byte groups (HW is a multiple of 4)
    const float4* sp = reinterpret_cast<const float4*>(src);
    const float4* ap = reinterpret_cast<const float4*>(dact_src);
    float4* dp = reinterpret_cast<float4*>(dst);
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned i = q % HW4, t = q / HW4;
        const unsigned c = t % Cg, n = t / Cg;
        float4 v = sp[(n * Csrc + c0s + c) * HW4 + i];
        const unsigned o = (n * Cdst + c0d + c) * HW4 + i;
        if (ap) {
            const float4 a = ap[o];
            v.x *= bn_act_grad_from_output(a.x, dact, slope);
            v.y *= bn_act_grad_from_output(a.y, dact, slope);
            v.z *= bn_act_grad_from_output(a.z, dact, slope);
            v.w *= bn_act_grad_from_output(a.w, dact, slope);
        }
        dp[o] = v;
    }
}
int bn_launch_chan_copy(const float* src, float* dst, int N, int Csrc, int c0s, int Cdst, int c0d,
                        int Cg, int HW, const float* dact_src, int dact, float slope, hipStream_t st) {
    if (HW & 3) return BN_E_BADARG;
    hipLaunchKernelGGL(k_chan_copy, dim3(pd_blocks((size_t)N * Cg * (HW >> 2))), dim3(PD_THREADS), 0, st,
                       src, dst, (unsigned)N, (unsigned)Csrc, (unsigned)c0s, (unsigned)Cdst, (unsigned)c0d,
                       (unsigned)Cg, (unsigned)(HW >> 2), dact_src, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Maps LARGER than the specialised kernels take (small side wider than 32, or 64 for the
// single-channel edge layers: frames beyond 128 pixels): spatial tiles with halos, every tile a
// pseudo-frame of the size the kernels are compiled for.
//
// Per axis, tile k of a 32-long small window (S = 32) holds small coordinates 30 k - 1 + p' and the
// matching 64-long big window holds big coordinates 60 k - 1 - pt + i, which is exactly the
// correspondence i = 2 p' - 1 + r  <->  b = 2 p - pt + r of the pt = 1 layer the kernels implement.
//   gather-down: outputs p' in [1, 31) see only rows inside the window -> the tile OWNS 30 outputs;
//   gather-up:   big rows i in [1 + pt, 61 + pt) get every contribution from inside the window -> the
//                tile owns 60 big rows;
//   weight gradient: the small window carries its owned 30 rows only (the rest zero), so every dy
//                element meets its inputs exactly once over all tiles.
// Everything outside the map reads as zero (the layer's own padding).  An axis that fits is one
// tile holding the whole map (the zero-padded embedding above).  Exact, like the padding: only
// the order of the additions differs from an untiled kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PD_THREADS) void k_tile_gather(const float* __restrict__ src,
                                                             float* __restrict__ dst, int N, int C,
                                                             int H, int W, BnTileAxis th,
                                                             BnTileAxis tw, int masked) {
    // 16-byte groups of the tiles (D is a multiple of 4), 32-bit indices (tensors below 2 GB)
    const unsigned T = th.T * tw.T, wq = tw.D >> 2, groups = th.D * wq;
    const unsigned total = (unsigned)N * T * C * groups;
    float4* dp = reinterpret_cast<float4*>(dst);
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pl = q / groups, e = q - pl * groups;
        const unsigned nt = pl / C, c = pl - nt * C;
        const unsigned n = nt / T, tix = nt - n * T;
        const int kh = tix / tw.T, kw = tix - kh * tw.T;
        const int i = e / wq, j0 = 4 * (int)(e - i * wq);
        const int y = th.step * kh - th.v0 + i, x0 = tw.step * kw - tw.v0;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && (!masked || (i >= th.v0 && i < th.v0 + th.V))) {
            const float* row = src + ((n * C + c) * H + y) * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + k, x = x0 + j;
                if (x >= 0 && x < W && (!masked || (j >= tw.v0 && j < tw.v0 + tw.V))) v[k] = row[x];
            }
        }
        dp[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int VEC>
__global__ __launch_bounds__(PD_THREADS) void k_tile_scatter(const float* __restrict__ src,
                                                              float* __restrict__ dst, int N, int C,
                                                              int H, int W, BnTileAxis th,
                                                              BnTileAxis tw,
                                                              const float* __restrict__ dact_src,
                                                              int dact, float slope) {
    const unsigned T = th.T * tw.T, wq = W / VEC, groups = H * wq, DD = th.D * tw.D;
    const unsigned total = (unsigned)N * C * groups;
    for (unsigned q = blockIdx.x * PD_THREADS + threadIdx.x; q < total; q += gridDim.x * PD_THREADS) {
        const unsigned pl = q / groups, e = q - pl * groups;
        const unsigned n = pl / C, c = pl - n * C;
        const int y = e / wq, x0 = VEC * (int)(e - y * wq);
        const int kh = th.T > 1 ? y / th.step : 0;
        const int i = th.v0 + y - th.step * kh;
        float v[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int x = x0 + k;
            const int kw = tw.T > 1 ? x / tw.step : 0;
            const int j = tw.v0 + x - tw.step * kw;
            v[k] = src[(((n * T + kh * tw.T + kw) * C + c) * th.D + i) * tw.D + j];
        }
        (void)DD;
        if (dact_src) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                v[k] *= bn_act_grad_from_output(dact_src[(size_t)q * VEC + k], dact, slope);
        }
        if (VEC == 4) reinterpret_cast<float4*>(dst)[q] = make_float4(v[0], v[1], v[2], v[VEC - 1]);
        else dst[q] = v[0];
    }
}

int bn_launch_tile_gather(const float* src, float* dst, int N, int C, int H, int W, BnTileAxis th,
                          BnTileAxis tw, int masked, hipStream_t st) {
    if (tw.D & 3) return BN_E_BADARG;
    hipLaunchKernelGGL(k_tile_gather, dim3(pd_blocks((size_t)N * th.T * tw.T * C * th.D * (tw.D >> 2))),
                       dim3(PD_THREADS), 0, st, src, dst, N, C, H, W, th, tw, masked);
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_tile_scatter(const float* src, float* dst, int N, int C, int H, int W, BnTileAxis th,
                           BnTileAxis tw, const float* dact_src, int dact, float slope,
                           hipStream_t st) {
    if ((W & 3) == 0 && ((((uintptr_t)dst) | ((uintptr_t)dact_src)) & 15u) == 0)
        hipLaunchKernelGGL(k_tile_scatter<4>, dim3(pd_blocks((size_t)N * C * H * (W >> 2))), dim3(PD_THREADS),
                           0, st, src, dst, N, C, H, W, th, tw, dact_src, dact, slope);
    else
        hipLaunchKernelGGL(k_tile_scatter<1>, dim3(pd_blocks((size_t)N * C * H * W)), dim3(PD_THREADS), 0, st,
                           src, dst, N, C, H, W, th, tw, dact_src, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// stride-5 gather-down onto a map of a few pixels (the windows do not overlap):
//   out[n, m, p, q] = act(b[m] + sum_{c,r,s} big[n, c, 5p + r - pt, 5q + s - pl] W[m, c, r, s]) * act'(dact_src)
// = (N P Q) x (C 25) im2col matrix times W^T, then the rows go back to NCHW.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PD_THREADS) void k_im2col_s5(const float* __restrict__ big,
                                                           float* __restrict__ col, BnGeom g) {
    const int PQ = g.Hs * g.Ws;
    const size_t total = (size_t)g.N * PQ * g.Cb * 25;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int tap = (int)(i % 25);
        size_t t = i / 25;
        const int c = (int)(t % g.Cb);
        t /= g.Cb;
        const int pq = (int)(t % PQ);
        const size_t n = t / PQ;
        const int h = 5 * (pq / g.Ws) + tap / 5 - g.pt, w = 5 * (pq % g.Ws) + tap % 5 - g.pl;
        col[i] = (h >= 0 && h < g.Hb && w >= 0 && w < g.Wb)
            ? big[((n * g.Cb + c) * g.Hb + h) * g.Wb + w] : 0.f;
    }
}

// tmp[(n, pq)][m] -> out[n][m][pq] with the activation and the derivative mask
__global__ __launch_bounds__(PD_THREADS) void k_s5_rows_to_nchw(const float* __restrict__ tmp,
                                                                 float* __restrict__ out,
                                                                 const float* __restrict__ dact_src,
                                                                 int N, int M, int PQ, int act, int dact,
                                                                 float slope) {
    const size_t total = (size_t)N * M * PQ;
    for (size_t i = (size_t)blockIdx.x * PD_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * PD_THREADS) {
        const int pq = (int)(i % PQ);
        const size_t t = i / PQ;
        const int m = (int)(t % M);
        const size_t n = t / M;
        float v = bn_apply_act(tmp[(n * PQ + pq) * M + m], act, slope);
        if (dact_src) v *= bn_act_grad_from_output(dact_src[i], dact, slope);
        out[i] = v;
    }
}

bool bn_s5_down_small_ok(const BnGeom& g) {
    return g.R == 5 && g.S == 5 && g.stride == 5 && g.Hs * g.Ws <= 16 && g.pt <= 4 && g.pl <= 4 &&
           g.Cs >= 16 && g.Cb >= 16 && (size_t)g.N * g.Hs * g.Ws * g.Cb * 25 * 4 < 0x7fffffffull;
}

static size_t s5_col_bytes(const BnGeom& g) {
    return ((size_t)g.N * g.Hs * g.Ws * g.Cb * 25 * 4 + 255) & ~(size_t)255;
}
static size_t s5_tmp_bytes(const BnGeom& g) {
    return ((size_t)g.N * g.Hs * g.Ws * g.Cs * 4 + 255) & ~(size_t)255;
}

size_t bn_s5_down_small_ws_bytes(const BnGeom& g) {
    return s5_col_bytes(g) + s5_tmp_bytes(g) + bn_gemm_ws_bytes(g.N * g.Hs * g.Ws, g.Cs, g.Cb * 25);
}

int bn_launch_s5_down_small(const float* big, const float* w, const float* bias, float* out,
                            const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    if (!ws || ws_bytes < bn_s5_down_small_ws_bytes(g)) return BN_E_WORKSPACE;
    const int PQ = g.Hs * g.Ws, rows = g.N * PQ, K = g.Cb * 25;
    float* col = (float*)ws;
    float* tmp = (float*)((char*)ws + s5_col_bytes(g));
    const size_t used = s5_col_bytes(g) + s5_tmp_bytes(g);
    hipLaunchKernelGGL(k_im2col_s5, dim3(pd_blocks((size_t)rows * K)), dim3(PD_THREADS), 0, st, big, col, g);
    BN_LAUNCH_CHECK();
    GemmArgs a;                                   // tmp[i, m] = b[m] + sum_k col[i, k] W[m, k]
    a.A = col; a.sai = K; a.sak = 1;
    a.B = w; a.sbk = 1; a.sbj = K;
    a.C = tmp; a.sci = g.Cs; a.scj = 1;
    a.M = rows; a.N = g.Cs; a.K = K;
    a.bias_j = bias; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = slope; a.accumulate = 0;
    const int rc = bn_launch_gemm(a, st, (char*)ws + used, ws_bytes - used);
    if (rc) return rc;
    hipLaunchKernelGGL(k_s5_rows_to_nchw, dim3(pd_blocks((size_t)rows * g.Cs)), dim3(PD_THREADS), 0, st, tmp,
                       out, dact_src, g.N, g.Cs, PQ, act, dact, slope);
    BN_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Everything else (kernel sizes up to 9, stride 1, odd channel counts -- what the reference's
// architecture search draws, ae_arch_2.json's last 4x4 stride-1 layer): im2col + the MFMA GEMM,
// a few dozen frames per pass so that the column matrix stays below 512 MB.  40 TFLOP/s class
// against ~1 TFLOP/s for the direct loops of conv_generic.hip.
//   rows = (n, p, q) of the small side, columns = (c, r, s) of the big side:
//   down:   out_rows = col W^T               (+ bias, activation, act' mask on the way to NCHW)
//   wgrad:  dW = small_rows^T col            (reduction over the rows, accumulated over passes)
//   up:     colg = small_rows W, then every big pixel gathers its <= ceil(R/s) ceil(S/s) terms
// ---------------------------------------------------------------------------------------------
#define COL_MAX_BYTES ((size_t)512 << 20)

__global__ __launch_bounds__(PD_THREADS) void k_im2col(const float* __restrict__ big,
                                                        float* __restrict__ col, BnGeom g, int n0,
                                                        int nf) {
    const unsigned PQ = g.Hs * g.Ws, RS = g.R * g.S, K = g.Cb * RS;
    const unsigned total = (unsigned)nf * PQ * K;
    for (unsigned i = blockIdx.x * PD_THREADS + threadIdx.x; i < total; i += gridDim.x * PD_THREADS) {
        const unsigned tap = i % RS;
        unsigned t = i / RS;
        const unsigned c = t % g.Cb;
        t /= g.Cb;
        const unsigned pq = t % PQ, n = t / PQ;
        const int p = pq / g.Ws, q = pq - p * g.Ws;
        const int r = tap / g.S, sx = tap - r * g.S;
        const int h = g.stride * p + r - g.pt, w = g.stride * q + sx - g.pl;
        col[i] = (h >= 0 && h < g.Hb && w >= 0 && w < g.Wb)
            ? big[(((size_t)(n0 + n) * g.Cb + c) * g.Hb + h) * g.Wb + w] : 0.f;
    }
}

// rows[(n, pq)][m] <- small[n0 + n][m][pq]
__global__ __launch_bounds__(PD_THREADS) void k_nchw_to_rows(const float* __restrict__ small,
                                                              float* __restrict__ rows, int n0, int nf,
                                                              int M, int PQ) {
    const unsigned total = (unsigned)nf * M * PQ;
    for (unsigned i = blockIdx.x * PD_THREADS + threadIdx.x; i < total; i += gridDim.x * PD_THREADS) {
        const unsigned m = i % M, t = i / M;
        const unsigned pq = t % PQ, n = t / PQ;
        rows[i] = small[((size_t)(n0 + n) * M + m) * PQ + pq];
    }
}

// out[n0 + n][cb][y][x] = act(bias[cb] + sum over the taps that reach (y, x)) * act'(dact_src)
__global__ __launch_bounds__(PD_THREADS) void k_col2im(const float* __restrict__ colg,
                                                        float* __restrict__ out,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ dact_src, BnGeom g,
                                                        int n0, int nf, int act, int dact,
                                                        float slope) {
    const unsigned HW = g.Hb * g.Wb, RS = g.R * g.S, K = g.Cb * RS, PQ = g.Hs * g.Ws;
    const unsigned total = (unsigned)nf * g.Cb * HW;
    for (unsigned i = blockIdx.x * PD_THREADS + threadIdx.x; i < total; i += gridDim.x * PD_THREADS) {
        const unsigned yx = i % HW, t = i / HW;
        const unsigned cb = t % g.Cb, n = t / g.Cb;
        const int y = yx / g.Wb, x = yx - y * g.Wb;
        float v = bias ? bias[cb] : 0.f;
        for (int r = 0; r < g.R; ++r) {
            const int pn = y + g.pt - r;
            if (pn < 0 || pn % g.stride) continue;
            const int p = pn / g.stride;
            if (p >= g.Hs) continue;
            for (int sx = 0; sx < g.S; ++sx) {
                const int qn = x + g.pl - sx;
                if (qn < 0 || qn % g.stride) continue;
                const int q = qn / g.stride;
                if (q >= g.Ws) continue;
                v += colg[((size_t)n * PQ + p * g.Ws + q) * K + cb * RS + r * g.S + sx];
            }
        }
        v = bn_apply_act(v, act, slope);
        const size_t o = (size_t)(n0 + n) * g.Cb * HW + (size_t)cb * HW + yx;
        if (dact_src) v *= bn_act_grad_from_output(dact_src[o], dact, slope);
        out[o] = v;
    }
}

static int col_frames(const BnGeom& g) {
    const size_t per = (size_t)g.Hs * g.Ws * g.Cb * g.R * g.S * 4;
    size_t nb = COL_MAX_BYTES / (per ? per : 1);
    const size_t rows_cap = ((size_t)1 << 21) / ((size_t)g.Hs * g.Ws);     // GEMM grid: rows / 32 < 65536
    if (nb > rows_cap) nb = rows_cap;
    if (nb > (size_t)g.N) nb = g.N;
    return (int)nb;
}
bool bn_col_ok(const BnGeom& g) {
    if (g.R > 9 || g.S > 9 || g.stride < 1) return false;
    if ((size_t)g.N * g.Cb * g.Hb * g.Wb * 4 >= 0x7fffffffull) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    return col_frames(g) >= 1;
}
static size_t col_col_bytes(const BnGeom& g) {
    return ((size_t)col_frames(g) * g.Hs * g.Ws * g.Cb * g.R * g.S * 4 + 255) & ~(size_t)255;
}
static size_t col_rows_bytes(const BnGeom& g) {
    return ((size_t)col_frames(g) * g.Hs * g.Ws * g.Cs * 4 + 255) & ~(size_t)255;
}
size_t bn_col_ws_bytes(const BnGeom& g) {
    const int rows = col_frames(g) * g.Hs * g.Ws, K = g.Cb * g.R * g.S;
    size_t gw = bn_gemm_ws_bytes(rows, g.Cs, K);
    const size_t g2 = bn_gemm_ws_bytes(g.Cs, K, rows), g3 = bn_gemm_ws_bytes(rows, K, g.Cs);
    if (g2 > gw) gw = g2;
    if (g3 > gw) gw = g3;
    return col_col_bytes(g) + col_rows_bytes(g) + gw;
}

int bn_launch_col_down(const float* big, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    if (!ws || ws_bytes < bn_col_ws_bytes(g)) return BN_E_WORKSPACE;
    const int PQ = g.Hs * g.Ws, K = g.Cb * g.R * g.S, nb = col_frames(g);
    float* col = (float*)ws;
    float* tmp = (float*)((char*)ws + col_col_bytes(g));
    const size_t used = col_col_bytes(g) + col_rows_bytes(g);
    for (int n0 = 0; n0 < g.N; n0 += nb) {
        const int nf = g.N - n0 < nb ? g.N - n0 : nb, rows = nf * PQ;
        hipLaunchKernelGGL(k_im2col, dim3(pd_blocks((size_t)rows * K)), dim3(PD_THREADS), 0, st, big, col, g, n0, nf);
        GemmArgs a;                               // tmp[i, m] = b[m] + sum_k col[i, k] W[m, k]
        a.A = col; a.sai = K; a.sak = 1;
        a.B = w; a.sbk = 1; a.sbj = K;
        a.C = tmp; a.sci = g.Cs; a.scj = 1;
        a.M = rows; a.N = g.Cs; a.K = K;
        a.bias_j = bias; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = slope; a.accumulate = 0;
        const int rc = bn_launch_gemm(a, st, (char*)ws + used, ws_bytes - used);
        if (rc) return rc;
        const size_t os = (size_t)n0 * g.Cs * PQ;
        hipLaunchKernelGGL(k_s5_rows_to_nchw, dim3(pd_blocks((size_t)rows * g.Cs)), dim3(PD_THREADS), 0, st, tmp,
                           out + os, dact_src ? dact_src + os : nullptr, nf, g.Cs, PQ, act, dact, slope);
    }
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_col_wgrad(const float* small, const float* big, float* dw, const BnGeom& g, int accumulate,
                        void* ws, size_t ws_bytes, hipStream_t st, float* db, int bias_side,
                        bool* bias_done) {
    if (!ws || ws_bytes < bn_col_ws_bytes(g)) return BN_E_WORKSPACE;
    const int PQ = g.Hs * g.Ws, K = g.Cb * g.R * g.S, nb = col_frames(g);
    float* col = (float*)ws;
    float* srows = (float*)((char*)ws + col_col_bytes(g));
    const size_t used = col_col_bytes(g) + col_rows_bytes(g);
    for (int n0 = 0; n0 < g.N; n0 += nb) {
        const int nf = g.N - n0 < nb ? g.N - n0 : nb, rows = nf * PQ;
        const int acc = (accumulate || n0 > 0) ? 1 : 0;
        hipLaunchKernelGGL(k_im2col, dim3(pd_blocks((size_t)rows * K)), dim3(PD_THREADS), 0, st, big, col, g, n0, nf);
        hipLaunchKernelGGL(k_nchw_to_rows, dim3(pd_blocks((size_t)rows * g.Cs)), dim3(PD_THREADS), 0, st, small,
                           srows, n0, nf, g.Cs, PQ);
        GemmArgs a;                               // dW[m, k] (+)= sum_i srows[i, m] col[i, k]
        a.A = srows; a.sai = 1; a.sak = g.Cs;
        a.B = col; a.sbk = K; a.sbj = 1;
        a.C = dw; a.sci = K; a.scj = 1;
        a.M = g.Cs; a.N = K; a.K = rows;
        a.bias_j = nullptr; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = 0.f; a.accumulate = acc;
        const int rc = bn_launch_gemm(a, st, (char*)ws + used, ws_bytes - used);
        if (rc) return rc;
    }
    // (the bias gradient is left to the caller's channel sums over the NCHW tensor: a column sum
    // over millions of rows with one wave per column took 3.5 ms per call)
    (void)db; (void)bias_side; (void)bias_done;
    BN_LAUNCH_CHECK();
    return 0;
}

int bn_launch_col_up(const float* small, const float* w, const float* bias, float* out,
                     const float* dact_src, const BnGeom& g, int act, int dact, float slope, void* ws,
                     size_t ws_bytes, hipStream_t st) {
    if (!ws || ws_bytes < bn_col_ws_bytes(g)) return BN_E_WORKSPACE;
    const int PQ = g.Hs * g.Ws, K = g.Cb * g.R * g.S, nb = col_frames(g);
    float* colg = (float*)ws;
    float* srows = (float*)((char*)ws + col_col_bytes(g));
    const size_t used = col_col_bytes(g) + col_rows_bytes(g);
    for (int n0 = 0; n0 < g.N; n0 += nb) {
        const int nf = g.N - n0 < nb ? g.N - n0 : nb, rows = nf * PQ;
        hipLaunchKernelGGL(k_nchw_to_rows, dim3(pd_blocks((size_t)rows * g.Cs)), dim3(PD_THREADS), 0, st, small,
                           srows, n0, nf, g.Cs, PQ);
        GemmArgs a;                               // colg[i, k] = sum_m srows[i, m] W[m, k]
        a.A = srows; a.sai = g.Cs; a.sak = 1;
        a.B = w; a.sbk = K; a.sbj = 1;
        a.C = colg; a.sci = K; a.scj = 1;
        a.M = rows; a.N = K; a.K = g.Cs;
        a.bias_j = nullptr; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = 0.f; a.accumulate = 0;
        const int rc = bn_launch_gemm(a, st, (char*)ws + used, ws_bytes - used);
        if (rc) return rc;
        hipLaunchKernelGGL(k_col2im, dim3(pd_blocks((size_t)nf * g.Cb * g.Hb * g.Wb)), dim3(PD_THREADS), 0, st,
                           colg, out, bias, dact_src, g, n0, nf, act, dact, slope);
    }
    BN_LAUNCH_CHECK();
    return 0;
}
