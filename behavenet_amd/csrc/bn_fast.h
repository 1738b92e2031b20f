// Shape-specialised fast paths (conv_mfma.hip / conv_edge.hip).  `bn_fast_*_plan` decides
// whether a geometry is served by a fast kernel, names that kernel (for the profiling hook) and
// reports the scratch it needs; it never launches anything.
#pragma once
#include "bn_common.h"

struct BnFastPlan {
    bool supported;
    const char* kernel_name;
    size_t ws_bytes;
    int variant;          // kernel-family specific selector
    int a, b, c, d;       // kernel-family specific tile parameters
};

BnFastPlan bn_fast_down_plan(const BnGeom& g);
BnFastPlan bn_fast_up_plan(const BnGeom& g);
BnFastPlan bn_fast_wgrad_plan(const BnGeom& g);

int bn_launch_down_fast(const BnFastPlan& plan, const float* big, const float* w,
                        const float* bias, float* out, const float* dact_src, const BnGeom& g,
                        int act, int dact, float slope, void* ws, hipStream_t st);
int bn_launch_up_fast(const BnFastPlan& plan, const float* small, const float* w,
                      const float* bias, float* out, const float* dact_src, const BnGeom& g,
                      int act, int dact, float slope, void* ws, hipStream_t st);
// db / bias_side (1: sum `small` per a-channel, 2: sum `big` per b-channel) / bias_done: the
// kernel may produce the bias gradient as a by-product; *bias_done tells whether it did
int bn_launch_wgrad_fast(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                         const BnGeom& g, int accumulate, void* ws, hipStream_t st,
                         float* db = nullptr, int bias_side = 0, bool* bias_done = nullptr);

// conv_mfma_down2.hip: 16-byte-DMA generation of the stride-2 gather-down kernel (chosen by
// bn_fast_down_plan when the geometry fits and no split-K is needed; plan.variant == 2)
bool bn_wgrad_pool_c1_ok(const BnGeom& g);
size_t bn_wgrad_pool_c1_ws_bytes(const BnGeom& g);
int bn_launch_wgrad_pool_c1(const float* x, const float* dy, const float* y, const int* idx, float* dw, float* db,
                            const BnGeom& g, int act, float slope, int accumulate, void* ws, hipStream_t st);
bool bn_s1in1_pool_ok(const BnGeom& g);
int bn_launch_s1in1_pool(const float* big, const float* w, const float* bias, float* y, int* idx, const BnGeom& g,
                         int act, float slope, hipStream_t st);
bool bn_down2_pool_ok(const BnGeom& g);
int bn_launch_down2_pool(const float* big, const float* w, const float* bias, float* y, int* idx, const BnGeom& g,
                         int act, float slope, hipStream_t st);
bool bn_down2_supported(const BnGeom& g, int MR, int NR);
bool bn_down2_m16_supported(const BnGeom& g, int NR);   // the 16-row tile (MR = 0 in the plans)
int bn_down2_splits(const BnGeom& g, int MR, int NR);
float bn_down2_fill(const BnGeom& g, int MR, int NR);
int bn_launch_down2(int MR, int NR, const float* big, const float* w, const float* bias,
                    float* out, const float* dact_src, const BnGeom& g, int act, int dact,
                    float slope, hipStream_t st, int splits = 1, void* ws = nullptr);

// conv_mfma_wgrad4.hip: 16-byte-DMA generation of the stride-2 weight gradient (tried first by
// bn_fast_wgrad_plan; plan.variant == 4)
BnFastPlan bn_wgrad4_plan(const BnGeom& g);
int bn_launch_wgrad4(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                     const BnGeom& g, int accumulate, void* ws, hipStream_t st, float* db,
                     int bias_side, bool* bias_done);

// conv_s5.hip: stride == kernel size (non-overlapping windows), direct-from-global MFMA GEMMs
BnFastPlan bn_s5_up_plan(const BnGeom& g);
BnFastPlan bn_s5_wgrad_plan(const BnGeom& g);
int bn_launch_up_s5(const float* small, const float* w, const float* bias, float* out,
                    const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                    hipStream_t st);
int bn_launch_wgrad_s5(const float* small, const float* big, float* dw, const BnGeom& g,
                       int accumulate, hipStream_t st);

// conv_edge.hip: HBM-bound single-channel-side layers (enc.conv0 / dec.convT4)
BnFastPlan bn_edge_wgrad_plan(const BnGeom& g);
int bn_launch_edge_wgrad(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                         const BnGeom& g, int accumulate, void* ws, hipStream_t st,
                         float* db = nullptr, int bias_side = 0, bool* bias_done = nullptr);
// stride-1 gather-down onto one or two channels, kernels 3 / 5 / 7 / 9 (conv_edge.hip, k_down_s1_c1)
bool bn_s1c1_ok(const BnGeom& g);
int bn_launch_s1c1(const float* big, const float* w, const float* bias, float* out, const float* dact_src,
                   const BnGeom& g, int act, int dact, float slope, hipStream_t st);
// stride-1 gather-down FROM one or two channels, kernels 3 / 5 / 7 / 9 (conv_edge.hip, k_down_s1_in1)
bool bn_s1in1_ok(const BnGeom& g);
int bn_launch_s1in1(const float* big, const float* w, const float* bias, float* out, const float* dact_src,
                    const BnGeom& g, int act, int dact, float slope, hipStream_t st);
BnFastPlan bn_edge_down_plan(const BnGeom& g);
const char* bn_edge_down_kernel_name(const BnGeom& g, int act, bool has_dact, bool u8);
const char* bn_edge_up_kernel_name(const BnGeom& g, bool loss);
int bn_launch_edge_down(const float* big, const float* w, const float* bias, float* out,
                        const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                        hipStream_t st, const unsigned char* u8 = nullptr);
BnFastPlan bn_edge_up_plan(const BnGeom& g);
int bn_launch_edge_up(const float* small, const float* w, const float* bias, float* out,
                      const BnGeom& g, int act, float slope, hipStream_t st,
                      const float* target = nullptr, const float* mask = nullptr,
                      float* dpre = nullptr, float* partial = nullptr);
int bn_edge_up_parts_per_frame(const BnGeom& g);

// conv_qgemm.hip: stride == kernel (5x5 s5) between an 8x8 and a 2x2 map: four dense 16-tap
// quadrant GEMMs per role (skips the taps that only ever meet padding)
bool bn_qgemm_supported(const BnGeom& g);
size_t bn_qgemm_ws_bytes(int role, const BnGeom& g);     // role: 0 down, 1 up, 2 wgrad
int bn_launch_qgemm_down(const float* big, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         void* ws, hipStream_t st);
int bn_launch_qgemm_up(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       void* ws, hipStream_t st);
int bn_launch_qgemm_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                          int accumulate, void* ws, hipStream_t st);

// conv_qgemm2.hip: second generation of the gather-up role and the weight gradient of the same
// layers (a workgroup walks all four quadrants; operands by LDS-DMA as they lie in memory; the weight
// gradient adds the quadrants that share a tap itself and yields the bias gradient of either side)
bool bn_qg2_up_supported(const BnGeom& g, int act, int dact);
int bn_launch_qg2_up(const float* small, const float* w, const float* bias, float* out,
                     const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                     hipStream_t st);
bool bn_qg2_wgrad_supported(const BnGeom& g);
int bn_launch_qg2_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                        int accumulate, float* db, int bias_side, hipStream_t st);

// conv_s5win.hip: stride == kernel (5x5 s5) between ANY pair of maps whose windows tile the big one: one dense
// GEMM per window over the taps that fall on the map (down, up), one GEMM over (frame, window) for the weight
// gradient; operands gathered from the NCHW tensors, no column matrix
bool bn_s5win_supported(const BnGeom& g);
size_t bn_s5win_ws_bytes(int role, const BnGeom& g);     // role: 0 down, 1 up, 2 wgrad
int bn_launch_s5win_down(const float* big, const float* w, const float* bias, float* out,
                         const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                         void* ws, hipStream_t st);
int bn_launch_s5win_up(const float* small, const float* w, const float* bias, float* out,
                       const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                       hipStream_t st);
int bn_launch_s5win_wgrad(const float* small, const float* big, float* dw, const BnGeom& g,
                          int accumulate, hipStream_t st);

// conv_mfma.hip: out = epilogue(sum_z part[z]) of a reduction split over workgroups (NCHW, C channels
// of npix pixels; fixed summation order)
int bn_launch_split_epilogue(const float* part, const float* bias, float* out, const float* dact_src,
                             size_t total, int splits, int C, int npix, int act, int dact, float slope,
                             hipStream_t st);
