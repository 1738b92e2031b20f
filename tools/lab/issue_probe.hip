// How much MFMA issue time do other instructions of the same wave (or of the wave sharing the SIMD)
// cost?  Loop body: 20 independent-enough v_mfma_f32_32x32x2_f32 (4 accumulators round robin) plus
// K filler instructions of one kind spread between them.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND, int K, int SHAPE>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters, float a0, float b0) {
    __shared__ float lds[8192];
    floatx16 acc[4];
    floatx4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    int x = threadIdx.x, y = 3;
    int sx = iters, sy = 5;
    lds[threadIdx.x] = a;
    __syncthreads();
    const int lp = (threadIdx.x & 63) * 4;
    float l = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            if (SHAPE == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            else {
                acc4[(2 * m) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * m) & 7], 0, 0, 0);
                acc4[(2 * m + 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[(2 * m + 1) & 7], 0, 0, 0);
            }
            // K fillers per 20 MFMAs, evenly spread
            if ((m * K) / 20 != ((m + 1) * K) / 20) {
                const int reps = ((m + 1) * K) / 20 - (m * K) / 20;
#pragma unroll
                for (int r = 0; r < reps; ++r) {
                    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if (KIND == 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sx) : "s"(sy));
                    if (KIND == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(lp)); l = t; }
                    if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(l) : "v"(b));
                    if (KIND == 4) { double t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(lp * 2)); l = (float)(long long)__double_as_longlong(t); }
                    if (KIND == 5) { floatx4 t; asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(t) : "v"(lp * 4)); l = t.x; }
                    if (KIND == 6) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((lp * 100) & 0x7ffc)); l = t; }
                    if (KIND == 7) { floatx2 t; asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(t) : "v"((lp * 100) & 0x7ffc)); l = t.x; }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 2 || KIND >= 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = l + x + sx;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 4; ++e) s += acc4[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int K, int SHAPE>
static void run(float* out, int blocks, const char* name) {
    const int iters = 4000;
    hipLaunchKernelGGL((k_probe<KIND, K, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, 50, 1.f, 2.f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<KIND, K, SHAPE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)blocks * 4 * iters * 20 * 4096.0;
    printf("%-10s K=%2d shape=%s blocks=%4d: %.3f ms %.1f TFLOP/s  (%.1f ns per 20-MFMA iteration per wave-slot)\n", name, K,
           SHAPE ? "16x16x4" : "32x32x2", blocks, ms, flop / ms / 1e9, ms * 1e6 / iters);
}

int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    for (int blocks : {256, 512}) {
        run<0, 0, 0>(out, blocks, "none");
        run<0, 10, 0>(out, blocks, "valu");
        run<2, 10, 0>(out, blocks, "lds b32"); run<2, 20, 0>(out, blocks, "lds b32");
        run<4, 5, 0>(out, blocks, "lds b64"); run<4, 10, 0>(out, blocks, "lds b64"); run<4, 20, 0>(out, blocks, "lds b64");
        run<5, 5, 0>(out, blocks, "lds 2xb64"); run<5, 10, 0>(out, blocks, "lds 2xb64");
        run<6, 10, 0>(out, blocks, "b32 cnfl"); run<6, 20, 0>(out, blocks, "b32 cnfl");
        run<7, 5, 0>(out, blocks, "2xb32 cnfl"); run<7, 10, 0>(out, blocks, "2xb32 cnfl");
        run<0, 0, 0>(out, blocks, "none");
    }
    return 0;
}
