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

// write-only stream: out[i] = v as float4 (HBM write ceiling for the edge kernels)
__global__ __launch_bounds__(256) void k_probe_fill(float4* out, size_t n4, float v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        out[i] = make_float4(v, v, v, v);
}
extern "C" int bn_debug_probe_fill(float* out, size_t n, int blocks, void* stream) {
    hipLaunchKernelGGL(k_probe_fill, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (float4*)out, n / 4, 1.0f);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// store-pattern variants: mode 0 plain, 1 nontemporal, 2 plain 4x float4 per thread contiguous,
// 3 nontemporal 4x contiguous
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_fill2(vf4* out, size_t n4, float v, int mode) {
    const vf4 val = {v, v, v, v};
    if (mode < 2) {
        const size_t stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            if (mode == 1) __builtin_nontemporal_store(val, &out[i]); else out[i] = val;
        }
    } else {
        const size_t stride = (size_t)gridDim.x * 1024;
        for (size_t b = (size_t)blockIdx.x * 1024; b < n4; b += stride) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t i = b + k * 256 + threadIdx.x;
                if (i < n4) {
                    if (mode == 3) __builtin_nontemporal_store(val, &out[i]); else out[i] = val;
                }
            }
        }
    }
}
extern "C" int bn_debug_probe_fill2(float* out, size_t n, int blocks, int mode, void* stream) {
    hipLaunchKernelGGL(k_probe_fill2, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (vf4*)out, n / 4, 1.0f, mode);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// mode 4: the k_down_c1 store pattern -- a wave writes 32-byte pieces to 32 channel rows
// (16 KB apart), four store instructions complete a 128-byte line per row
__global__ __launch_bounds__(256) void k_probe_fill3(float* out, int n_frames) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int n = blockIdx.x >> 3, p0 = (blockIdx.x & 7) * 8;
    if (n >= n_frames) return;
    const size_t chan = ((size_t)n * 32 + li) * 4096;
    for (int bk = 0; bk < 4; ++bk) {
        const int blk = wv * 4 + bk, pr = blk >> 1, q0 = (blk & 1) * 32;
        const size_t row = chan + (size_t)(p0 + pr) * 64 + q0 + 4 * kk;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(out + row + 8 * g) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
extern "C" int bn_debug_probe_fill3(float* out, int n_frames, void* stream) {
    hipLaunchKernelGGL(k_probe_fill3, dim3(n_frames * 8), dim3(256), 0, (hipStream_t)stream, out,
                       n_frames);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// LDS-DMA semantics probe: odd lanes use an out-of-range offset; LDS is pre-filled with 7.0
__global__ void k_probe_lds_dma(const float* p, float* o, int n) {
    __shared__ float lds[256];
    lds[threadIdx.x] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 4, 0x00020000);
    int off = threadIdx.x * 4;
    if (threadIdx.x & 1) off = 0x7fffffff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + (threadIdx.x >> 6) * 64, 4, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    o[threadIdx.x] = lds[threadIdx.x];
}
extern "C" int bn_debug_probe_lds_dma(const float* p, float* o, int n, void* stream) {
    hipLaunchKernelGGL(k_probe_lds_dma, dim3(1), dim3(256), 0, (hipStream_t)stream, p, o, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
