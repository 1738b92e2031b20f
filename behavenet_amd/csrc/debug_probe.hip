// Hardware probes used by tools/ (not part of the product path): pure-MFMA issue-rate ceiling.
#include "bn_common.h"
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_probe_mfma(float* out, int iters, float a0, float b0) {
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// launches `blocks` workgroups of 4 waves, each wave issuing 4*iters MFMAs (4096 FLOP each)
extern "C" int bn_debug_probe_mfma(float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_probe_mfma, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters,
                       1.0f, 0.5f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
