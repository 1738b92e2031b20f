// Internal launcher prototypes (one per kernel family); implemented in the .hip files,
// dispatched from capi.hip.
#pragma once
#include "bn_common.h"

struct GemmArgs {
    const float* A; long sai, sak;
    const float* B; long sbk, sbj;
    float* C; long sci, scj;
    int M, N, K;               // C is M x N, reduction length K
    const float* bias_j;       // nullable, added per column j
    const float* dact_src;     // nullable, same layout as C
    int dact; float slope;
    int accumulate;
    float* part;               // set by bn_launch_gemm: partial tiles of a cross-workgroup split
    int kslice;                // reduction length per workgroup slice
};

// conv_generic.hip
int bn_launch_down_generic(const float* big, const float* w, const float* bias, float* out,
                           const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                           hipStream_t st);
int bn_launch_up_generic(const float* small, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         hipStream_t st);
int bn_launch_wgrad_generic(const float* small, const float* big, float* dw, const BnGeom& g,
                            int accumulate, hipStream_t st);
size_t bn_channel_sum_ws_bytes(int N, int C, int npix);
int bn_launch_channel_sum(const float* t, float* db, int N, int C, int npix, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st);

// gemm.hip
int bn_launch_gemm(const GemmArgs& a, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
size_t bn_gemm_ws_bytes(int M, int N, int K);
int bn_launch_col_sum(const float* dy, float* db, int M, int N, int accumulate, hipStream_t st);
// nn.Linear in one launch: up to two products (nullable) and the column sums of dy (M x N) into db
// (nullable), side by side in one grid
int bn_launch_linear_jobs(const GemmArgs* g0, const GemmArgs* g1, const float* dy, float* db, int M,
                          int N, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
size_t bn_linear_jobs_ws_bytes(int M, int N, int K);

// elementwise.hip
int bn_launch_act_fwd(const float* x, float* y, size_t n, int act, float slope, hipStream_t st);
int bn_launch_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act, float slope,
                      hipStream_t st);
int bn_launch_sqerr_frame_sums(const float* pred, const float* target, const float* mask,
                               float* frame_sums, int N, size_t D, hipStream_t st);
int bn_launch_sqerr_bwd(const float* pred, const float* target, const float* mask, float* dpred,
                        size_t n, float scale, const float* gscale, hipStream_t st);
int bn_launch_scale_frames(float* t, const float* frame_scale, const float* group_scale,
                           const int* group_of_frame, int N, size_t D, hipStream_t st);
int bn_launch_reduce_sum(const float* in, float* out, size_t n, float scale, hipStream_t st);
int bn_launch_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                          size_t n, hipStream_t st);
int bn_launch_kl_rows(const float* mu, const float* logvar, float* kl_rows, int N, int D,
                      hipStream_t st);
int bn_launch_reparam_bwd(const float* dz, const float* z, const float* mu, float* dlogvar,
                          size_t n, hipStream_t st);
int bn_launch_kl_bwd(const float* mu, const float* logvar, float* dmu, float* dlogvar, size_t n,
                     float scale, const float* gscale, hipStream_t st);
int bn_launch_adam(float* p, const float* g, float* m, float* v, float* vmax, size_t n, float lr,
                   float b1, float b2, float eps, float wd, int step, hipStream_t st);
int bn_launch_u8_to_unit_float(const unsigned char* in, float* out, size_t n, hipStream_t st);

// batchnorm.hip
size_t bn_batchnorm_ws_bytes_impl(int N, int C);
int bn_launch_bn_stats(const float* x, float* mean, float* var, int N, int C, int HW, void* ws,
                       hipStream_t st);
int bn_launch_bn_finalize(const float* mean, const float* var, float* invstd, float* running_mean,
                          float* running_var, int C, float eps, float momentum, float unbias,
                          hipStream_t st);
int bn_launch_bn_act_fwd(const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* y, int N, int C, int HW,
                         int act, float slope, hipStream_t st);
int bn_launch_bn_act_bwd(const float* x, const float* y, const float* dy, const float* mean,
                         const float* invstd, const float* gamma, float* dx, float* dgamma,
                         float* dbeta, int accumulate, int batch_stats, int N, int C, int HW,
                         int act, float slope, void* ws, hipStream_t st);

int bn_launch_bn_moment(const float* x, const float* center, float* sums, int N, int C, int HW,
                        void* ws, hipStream_t st);
int bn_launch_bn_bwd_reduce(const float* x, const float* y, const float* dy, const float* mean,
                            const float* invstd, float* sum_dz, float* sum_dzx, int N, int C,
                            int HW, int act, float slope, void* ws, hipStream_t st);
int bn_launch_bn_train_fwd_chunks(const float* x, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches,
                                  float* y, float* mean, float* invstd, const int* bounds,
                                  const float* factors, int n_chunks, int C, int HW, float eps, int act,
                                  float slope, void* ws, hipStream_t st);
int bn_launch_bn_act_bwd_chunks(const float* x, const float* y, const float* dy, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* dx,
                                float* dgamma, float* dbeta, int accumulate, const int* bounds,
                                int n_chunks, int C, int HW, int act, float slope, void* ws,
                                hipStream_t st);
int bn_launch_bn_bwd_apply(const float* x, const float* y, const float* dy, const float* mean,
                           const float* invstd, const float* gamma, const float* sum_dz,
                           const float* sum_dzx, float* dx, int N, int C, int HW, float inv_count,
                           int act, float slope, hipStream_t st);

// decomposed_kl.hip
int bn_launch_dkl_fwd(const float* z, const float* mu, const float* lv, float* out3,
                      float* log_qz, float* lse, float* terms, int N, int D, hipStream_t st);
int bn_launch_dkl_bwd(const float* z, const float* mu, const float* lv, const float* log_qz,
                      const float* lse, const float* g3, float* dz, float* dmu, float* dlv, int N,
                      int D, hipStream_t st);

// conv_pad.hip: zero-padded copies for geometries off the fast paths, stride-5 1x1 maps as a GEMM
int bn_launch_pad2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                    int oh, int ow, hipStream_t st);
int bn_launch_crop2d(const float* src, float* dst, size_t planes, int H, int W, int Hp, int Wp,
                     int oh, int ow, const float* dact_src, int dact, float slope, hipStream_t st);
// spatial tiles with halos (maps larger than the specialised kernels take): one axis of the tiling.
// Tile k holds source coordinates step * k - v0 + i for i in [0, D); its elements [v0, v0 + V) are
// the ones it OWNS (every source coordinate is owned by exactly one tile: V == step when T > 1).
struct BnTileAxis { int T, step, v0, V, D; };
// src (N, C, H, W) -> dst ((N Th Tw), C, Dh, Dw), zeros outside the map; masked: owned elements only
int bn_launch_tile_gather(const float* src, float* dst, int N, int C, int H, int W, BnTileAxis th,
                          BnTileAxis tw, int masked, hipStream_t st);
// dst (N, C, H, W) <- the owning tile's element, times act'(dact_src) when given
int bn_launch_tile_scatter(const float* src, float* dst, int N, int C, int H, int W, BnTileAxis th,
                           BnTileAxis tw, const float* dact_src, int dact, float slope,
                           hipStream_t st);
// dst[n][c0d + c][i] <- src[n][c0s + c][i], c < Cg, i < HW (a multiple of 4); optional act' mask
int bn_launch_chan_copy(const float* src, float* dst, int N, int Csrc, int c0s, int Cdst, int c0d,
                        int Cg, int HW, const float* dact_src, int dact, float slope, hipStream_t st);
// im2col + GEMM for whatever no specialised kernel (or detour onto one) serves
bool bn_col_ok(const BnGeom& g);
size_t bn_col_ws_bytes(const BnGeom& g);
int bn_launch_col_down(const float* big, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope, void* ws,
                       size_t ws_bytes, hipStream_t st);
int bn_launch_col_up(const float* small, const float* w, const float* bias, float* out,
                     const float* dact_src, const BnGeom& g, int act, int dact, float slope, void* ws,
                     size_t ws_bytes, hipStream_t st);
int bn_launch_col_wgrad(const float* small, const float* big, float* dw, const BnGeom& g, int accumulate,
                        void* ws, size_t ws_bytes, hipStream_t st, float* db, int bias_side,
                        bool* bias_done);
// kernels smaller than 5x5 embedded in 5x5 taps (weights [pairs][R][S] <-> [pairs][5][5])
#define BN_PAD_TAPS_MAX_JOBS 16
struct BnPadTapsJob { const float* w; float* w5; unsigned pairs; int R, S, dr, ds, blocks; };
struct BnPadTapsJobs { int n; BnPadTapsJob job[BN_PAD_TAPS_MAX_JOBS]; };
int bn_launch_pad_taps_jobs(BnPadTapsJobs* p, hipStream_t st);
int bn_launch_pad_taps(const float* w, float* w5, size_t pairs, int R, int S, hipStream_t st, int dr = 0,
                       int ds = 0);
int bn_launch_flip_taps(const float* w, float* wf, int Cs, int Cb, int RS, hipStream_t st);
int bn_launch_crop_taps(const float* dw5, float* dw, size_t pairs, int R, int S, int accumulate,
                        hipStream_t st, const float* db5 = nullptr, float* db = nullptr, int nb = 0,
                        int dr = 0, int ds = 0);
// kernels larger than 5x5 with stride 2 as stride-1 5x5 layers on the phases of the big map (conv_pad.hip)
int bn_launch_space_to_depth(const float* x, float* X, int N, int C, int Hy, int Wy, hipStream_t st);
int bn_launch_bigk_phase_pack(const float* w, float* w1, int Cs, int Cb, int R, int S, const int* kr, const int* ofr,
                              const int* kc, const int* ofc, int sgn, int phase_out, hipStream_t st);
int bn_launch_bigk_phase_unpack(const float* dw1, float* dw, int Cs, int Cb, int R, int S, const int* kr,
                                const int* ofr, const int* kc, const int* ofc, int accumulate, const float* db5,
                                float* db, int nb, hipStream_t st);
// ... and with stride 1: four shifted copies of the big map against the four blocks of taps
int bn_launch_shift_cat(const float* x, float* y, int N, int C, int H, int W, int Ho, int Wo, int dr0, int dr1,
                        int dc0, int dc1, hipStream_t st);
int bn_launch_bigk_pack(const float* w, float* w5, int Cs, int Cb, int R, int S, int L0r, int L0c, hipStream_t st);
int bn_launch_bigk_unpack(const float* dw5, float* dw, int Cs, int Cb, int R, int S, int L0r, int L0c,
                          int accumulate, const float* db5, float* db, int nb, hipStream_t st);
int bn_launch_depth_to_space(const float* y, float* out, const float* bias, const float* dact_src, int N, int C,
                             int Hy, int Wy, int act, int dact, float slope, hipStream_t st);
bool bn_s5_down_small_ok(const BnGeom& g);
size_t bn_s5_down_small_ws_bytes(const BnGeom& g);
int bn_launch_s5_down_small(const float* big, const float* w, const float* bias, float* out,
                          const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                          void* ws, size_t ws_bytes, hipStream_t st);
