// Shared host/device helpers for libbehavenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
// Placement of a hot loop in the code object: `.p2align A` plus S s_nop in front of it (conv_mfma_up.hip, UP2_LOOP_SHIFT,
// has the story: the same instruction stream runs 206-239 us depending on where it lies)
#define BN_STR2(x) #x
#define BN_STR(x) BN_STR2(x)
#define BN_LOOP_PLACE(A, S) asm volatile(".p2align " BN_STR(A) "\n .rept " BN_STR(S) "\n s_nop 0\n .endr" ::: "memory")

struct BnGeom {
    int N;
    int Cs, Hs, Ws;   // small side: channels, height, width
    int Cb, Hb, Wb;   // big side
    int R, S, stride;
    int pt, pl;       // offset of tap (0,0) of small pixel (0,0) in big coordinates is (-pt,-pl)
    int CsS;          // channels per FRAME of the small tensor in memory when it is a channel window of
                      // a wider tensor (0 = Cs: contiguous).  Honoured by the single-channel edge kernels
                      // that serve channel groups in place (k_down_c1p, k_down_c1s, k_wgrad_c1d); every
                      // other kernel requires 0.
    int KV;           // 5x5 taps zero-extended from a smaller kernel (capi.hip, taps_plan): taps with r >= KV
                      // or s >= KV are zero and the stride-2 families skip their products (0 = all 25)
    int K0;           // ... and taps with r < K0 or s < K0 (a 3x3 kernel embedded at (1, 1): K0 = 1, KV = 4)
};
static inline __host__ __device__ int bn_cs_stride(const BnGeom& g) { return g.CsS > 0 ? g.CsS : g.Cs; }

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
    bool active;
    hipStream_t stream;
    int slot;
    // brackets everything the op launches with two event records on the stream AND offers a
    // second event pair to the launcher of the op's main kernel (BN_LAUNCH_MAIN), which attaches
    // it to that dispatch (hipExtLaunchKernelGGL start / stop events = the kernel's own begin /
    // end timestamps, what rocprofv3 reports): bn_prof_read = the op, bn_prof_read_main = the kernel
    BnProfScope(int family, int C, int K, const char* kernel_name, hipStream_t s);
    ~BnProfScope();
};
// launcher side: true (and the pair) if a profiling scope is waiting for its main kernel
bool bn_prof_take_dispatch_events(hipEvent_t* e0, hipEvent_t* e1);

// launch of an op's MAIN kernel: with bench.py's hook armed the dispatch carries the events
#define BN_LAUNCH_MAIN(kernel, grid, block, lds, st, ...)                                        \
    do {                                                                                         \
        hipEvent_t _e0 = nullptr, _e1 = nullptr;                                                 \
        if (bn_prof_take_dispatch_events(&_e0, &_e1))                                            \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, _e0, _e1, 0, __VA_ARGS__);       \
        else                                                                                     \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                       \
    } while (0)

#define BN_LAUNCH_CHECK()                                   \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)
