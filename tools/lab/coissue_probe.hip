// What can a wave issue while the wave it shares a SIMD with runs a dense MFMA stream?
// 512 workgroups of 4 waves: workgroup b < 256 multiplies (role A), b >= 256 (same CU, dispatch order)
// issues batches of memory instructions and timestamps them (role B).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds, int vo, int so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, vo, so, 0, 0);
}

// AMODE: 0 idle, 1 MFMA only, 2 MFMA + ds_read per 2 MFMAs, 3 MFMA + v_add per 2 MFMAs, 4 MFMA + s_nop gaps
// BMODE: 0 LDS-DMA batch of 7, 1 global_load batch of 7, 2 ds_write batch, 3 salu batch, 4 valu batch
template <int AMODE, int BMODE>
__global__ __launch_bounds__(256) void k_co(const float* src, float* out, unsigned long long* tout, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (blockIdx.x < 256) {
        if (AMODE == 0) return;
        floatx16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        float a = a0 + threadIdx.x, b = 2.f, l = 0.f;
        int x = lane, la = lane * 4;
        lds[threadIdx.x] = a;
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 20; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
                if (m & 1) {
                    if (AMODE == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(la)); l = t; }
                    if (AMODE == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(la));
                    if (AMODE == 4) asm volatile("s_nop 7");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (AMODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        float s = l + x;
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 24, 0x00020000);
        const int vo = (threadIdx.x * 16 + blockIdx.x * 4096) & ((1 << 22) - 1);
        float accv = 0.f; int sx = iters, vx = lane;
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int bt = 0; bt < 64; ++bt) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                if (BMODE == 0) dma16(r, lds + 4 * (256 * k + 64 * wv), vo, k * 65536 + bt * 1024);
                if (BMODE == 1) accv += src[(vo >> 2) + k * 16384 + bt * 256];
                if (BMODE == 2) lds[threadIdx.x + 256 * k] = accv;
                if (BMODE == 3) asm volatile("s_add_u32 %0, %0, %0" : "+s"(sx));
                if (BMODE == 4) asm volatile("v_add_u32 %0, %0, %0" : "+v"(vx));
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) tout[blockIdx.x - 256] = t1 - t0;
        out[blockIdx.x * 256 + threadIdx.x] = accv + lds[threadIdx.x] + sx + vx;
    }
}

template <int AMODE, int BMODE>
static void run(const float* src, float* out, unsigned long long* tout, const char* an, const char* bn) {
    hipLaunchKernelGGL((k_co<AMODE, BMODE>), dim3(512), dim3(256), 0, 0, src, out, tout, 3000, 1.f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), tout, 256 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("A: %-22s B: %-14s  64 batches of 7: median %8llu cycles (min %llu max %llu) = %.0f cycles per batch\n", an, bn, h[128], h[0], h[255], h[128] / 64.0);
}

int main() {
    float *src, *out; unsigned long long* tout;
    CK(hipMalloc(&src, 1 << 25)); CK(hipMalloc(&out, 512 * 256 * 4)); CK(hipMalloc(&tout, 256 * 8));
    CK(hipMemset(src, 0, 1 << 25));
#define ROW(B, bn) run<0, B>(src, out, tout, "idle", bn); run<1, B>(src, out, tout, "mfma", bn); run<2, B>(src, out, tout, "mfma + ds_read", bn); run<3, B>(src, out, tout, "mfma + v_add", bn); run<4, B>(src, out, tout, "mfma + s_nop 7", bn);
    ROW(0, "lds-dma x7") ROW(1, "global_load x7") ROW(2, "ds_write x7") ROW(3, "s_add x7") ROW(4, "v_add x7")
    return 0;
}
