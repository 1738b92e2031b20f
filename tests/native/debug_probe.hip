// Test-only library (tests/native/libbn_debug.so, include/behavenet_hip_debug.h): hardware probes used
// by tools/ and the LDS-poisoning aid of the GPU tests.  NOT linked into libbehavenet_hip.so.
#include "../../behavenet_amd/csrc/bn_common.h"
#include "../../include/behavenet_hip_debug.h"
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

// E0-like store streams with different contiguity per store instruction (one wave = one workgroup,
// persistent over "units" of ROWS output rows x 64 px x 32 channels, as k_down_c1):
//   mode 0: 4 channels x 256 B (one image row each)       -- what k_down_c1 does
//   mode 1: 2 channels x 512 B (two adjacent rows each)
//   mode 2: 1 channel  x 1 KB  (four adjacent rows)
__global__ __launch_bounds__(64) void k_probe_fill4(float* out, int n_frames, int mode) {
    const int lane = threadIdx.x;
    const int rows = mode == 0 ? 1 : (mode == 1 ? 2 : 4);     // rows per store instruction
    const int upf = 64 / 4;                                   // units of 4 rows per frame
    const int units = n_frames * upf;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int n = u / upf, p0 = 4 * (u - n * upf);
        // 4 rows x 32 channels x 256 B = 32 KB per unit = 32 store instructions of 1 KB
        for (int r0 = 0; r0 < 4; r0 += rows) {
            const int chans = 4 / rows;                        // channels per instruction
            for (int c0 = 0; c0 < 32; c0 += chans) {
                const int per = 64 / chans;                    // lanes per channel
                const int ch = c0 + lane / per;
                const size_t off = (((size_t)n * 32 + ch) * 64 + (p0 + r0)) * 64 + 4 * (lane % per);
                *reinterpret_cast<float4*>(out + off) = v;
            }
        }
    }
}
extern "C" int bn_debug_probe_fill4(float* out, int n_frames, int mode, int grid, void* stream) {
    hipLaunchKernelGGL(k_probe_fill4, dim3(grid), dim3(64), 0, (hipStream_t)stream, out, n_frames,
                       mode);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// test aid: leave NaNs in (most of) every CU's LDS, so that a kernel reading LDS it never wrote
// (e.g. a padded tap with a zero weight: 0 * NaN) shows up as NaN in the parity tests
__global__ __launch_bounds__(256) void k_poison_lds(float* sink) {
    extern __shared__ float l[];
    const float qnan = __builtin_nanf("");
    for (int i = threadIdx.x; i < 160 * 256 - 64; i += 256) l[i] = qnan;
    __syncthreads();
    if (sink && l[threadIdx.x] == 0.f) sink[0] = 1.f;      // keeps the stores alive
}
extern "C" int bn_debug_poison_lds(float* sink, void* stream) {
    static bool attr_set = false;
    const size_t bytes = (160 * 256 - 64) * sizeof(float);      // just under the 160 KB of a CU
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_poison_lds,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(256), bytes, (hipStream_t)stream, sink);
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

// ---------------------------------------------------------------------------------------------
// MFMA + LDS-operand ceiling in the shapes the conv kernels use (no global traffic):
//  mode 0: 32x32x2, 2x2 register blocking (2 A + 2 B ds_read_b32 per 4 MFMAs)   [k_down_mfma]
//  mode 1: 16x16x4, 25 accumulators (1 A + 25 B ds_read_b32 per 25 MFMAs)        [k_wgrad_mfma]
//  mode 2: as 0 but operands stay in registers (pure issue rate, data still random)
// ---------------------------------------------------------------------------------------------
typedef float floatx4p __attribute__((ext_vector_type(4)));
__global__ void k_probe_mfma_lds(float* out, int iters, int mode) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x)
        lds[i] = __sinf(0.37f * i + blockIdx.x) + 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float s = 0.f;
    if (mode == 0 || mode == 2) {
        floatx16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const float* ap = lds + wv * 128 + (lane & 31) + 66 * (lane >> 5);
        const float* bp = lds + 4096 + (lane & 31) + 130 * (lane >> 5);
        float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
        for (int it = 0; it < iters; ++it) {
            if (mode == 0) {
                const int o = (it & 15) * 132;
                a0 = ap[o]; a1 = ap[o + 32]; b0 = bp[o]; b1 = bp[o + 32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        floatx4p acc[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) acc[t] = (floatx4p){0.f, 0.f, 0.f, 0.f};
        const float* ap = lds + wv * 64 + (lane & 15) * 66 + (lane >> 4);
        const float* bp = lds + 4096 + (lane & 15) * 34 + (lane >> 4);
        for (int it = 0; it < iters; ++it) {
            const int o = (it & 15) * 4;
            const float av = ap[o];
#pragma unroll
            for (int t = 0; t < 25; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[o + 40 * (t / 5) + (t % 5)],
                                                             acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 25; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// returns FLOP per workgroup-iteration through *flop_per_iter_per_wave
extern "C" int bn_debug_probe_mfma_lds(float* out, int blocks, int threads, int iters, int mode,
                                       void* stream) {
    hipLaunchKernelGGL(k_probe_mfma_lds, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out,
                       iters, mode);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
