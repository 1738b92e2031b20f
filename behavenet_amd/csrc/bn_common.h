// Shared host/device helpers for libbehavenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/behavenet_hip.h"

#define BN_WAVE 64

// Experiment hooks.  The dispatch code consults a few environment variables (tile shapes, split
// counts, kernel generations: tools/README.md) ONLY in the tuning build (`make tuning`,
// -DBN_TUNING -> ../libbehavenet_hip_tuning.so, selected with BN_HIP_LIB); in the product library
// this is a constant null pointer, every hook folds to its default and no environment is read.
#include <stdlib.h>
#ifdef BN_TUNING
static inline const char* bn_tune_env(const char* name) { return getenv(name); }
#else
static inline const char* bn_tune_env(const char*) { return nullptr; }
#endif

// ---------------------------------------------------------------------------------------------
// geometry shared by the three kernel families (see DESIGN.md "Kernel families")
//
//  "gather-down" (conv-like):   out[n,m,p,q] = sum_{c,r,s} in[n,c,p*st+r-pt,q*st+s-pl] * W(m,c,r,s)
//  "gather-up"   (convT-like):  out[n,m,h,w] = sum_{c,r,s} in[n,c,(h+pt-r)/st,(w+pl-s)/st] * W(c,m,r,s)
//  "wgrad":                     dW[a,b,r,s]  = sum_{n,p,q} small[n,a,p,q] * big[n,b,p*st+r-pt,q*st+s-pl]
//
// `big` is the spatially larger tensor (Hb x Wb), `small` the smaller one (Hs x Ws); weights are
// always indexed [small-side channel][big-side channel][r][s] when WT == false (conv layout
// w[k][c][r][s] with k on the small side) -- and a ConvTranspose2d weight w[ci][co][r][s] has
// exactly the same layout (ci is on the small side).
// ---------------------------------------------------------------------------------------------
struct BnGeom {
    int N;
    int Cs, Hs, Ws;   // small side: channels, height, width
    int Cb, Hb, Wb;   // big side
    int R, S, stride;
    int pt, pl;       // offset of tap (0,0) of small pixel (0,0) in big coordinates is (-pt,-pl)
};

static inline int bn_geom_ok(const BnGeom& g) {
    return g.N > 0 && g.Cs > 0 && g.Hs > 0 && g.Ws > 0 && g.Cb > 0 && g.Hb > 0 && g.Wb > 0 &&
           g.R > 0 && g.S > 0 && g.stride > 0 && g.pt >= 0 && g.pl >= 0;
}

__device__ __forceinline__ float bn_apply_act(float v, int act, float slope) {
    if (act == BN_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == BN_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// derivative of the activation expressed through its saved OUTPUT y
__device__ __forceinline__ float bn_act_grad_from_output(float y, int act, float slope) {
    if (act == BN_ACT_LRELU) return y > 0.f ? 1.f : slope;
    if (act == BN_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// ---------------------------------------------------------------------------------------------
// profiling hook (bn_prof_*): brackets launches of one kernel family with hipEvents
// ---------------------------------------------------------------------------------------------
struct BnProfScope {
    bool active, on_dispatch;
    hipStream_t stream;
    hipEvent_t e0, e1;
    // on_dispatch: the op is ONE kernel whose launcher attaches the events to the dispatch
    // itself (hipExtLaunchKernelGGL start/stop events = the kernel's own begin/end timestamps,
    // what rocprofv3 reports) instead of bracketing it with two event records on the stream
    BnProfScope(int family, int C, int K, const char* kernel_name, hipStream_t s,
                bool on_dispatch = false);
    ~BnProfScope();
};
// launcher side of on_dispatch: true (and the pair) if a profiling scope is waiting for it
bool bn_prof_take_dispatch_events(hipEvent_t* e0, hipEvent_t* e1);

#define BN_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)
