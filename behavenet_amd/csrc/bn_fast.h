// Shape-specialised fast paths (conv_mfma.hip / conv_edge.hip).  `*_supported` names the device
// kernel that will serve the geometry (for the profiling hook) and never launches anything.
#pragma once
#include "bn_common.h"

bool bn_fast_down_supported(const BnGeom& g, const char** kernel_name);
bool bn_fast_up_supported(const BnGeom& g, const char** kernel_name);
bool bn_fast_wgrad_supported(const BnGeom& g, const char** kernel_name);
size_t bn_fast_wgrad_ws_bytes(const BnGeom& g);

int bn_launch_down_fast(const float* big, const float* w, const float* bias, float* out,
                        const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                        hipStream_t st);
int bn_launch_up_fast(const float* small, const float* w, const float* bias, float* out,
                      const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                      hipStream_t st);
int bn_launch_wgrad_fast(const float* small, const float* big, float* dw, const BnGeom& g,
                         int accumulate, void* ws, hipStream_t st);
