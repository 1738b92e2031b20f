// Fast-path dispatcher (placeholder until the MFMA kernels land).
#include "bn_common.h"
#include "bn_fast.h"

bool bn_fast_down_supported(const BnGeom&, const char**) { return false; }
bool bn_fast_up_supported(const BnGeom&, const char**) { return false; }
bool bn_fast_wgrad_supported(const BnGeom&, const char**) { return false; }
size_t bn_fast_wgrad_ws_bytes(const BnGeom&) { return 0; }
int bn_launch_down_fast(const float*, const float*, const float*, float*, const float*,
                        const BnGeom&, int, int, float, hipStream_t) { return BN_E_SHAPE; }
int bn_launch_up_fast(const float*, const float*, const float*, float*, const float*,
                      const BnGeom&, int, int, float, hipStream_t) { return BN_E_SHAPE; }
int bn_launch_wgrad_fast(const float*, const float*, float*, const BnGeom&, int, void*,
                         hipStream_t) { return BN_E_SHAPE; }
