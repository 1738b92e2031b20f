// Does a 16-byte LDS-DMA (buffer_load_dwordx4 ... lds) honour a 4-byte aligned global offset?
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds, int vo, int so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, vo, so, 0, 0);
}
__global__ void k(const float* src, float* out, int shift) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096, 0x00020000);
    dma16(r, lds, 20 * threadIdx.x + 4 * shift, 0);           // 5-float stride + shift: all alignments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}
int main() {
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 64 * 4 * 4); hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 2; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o, shift);
        float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) for (int i = 0; i < 4; ++i) if (r[t * 4 + i] != (float)(5 * t + shift + i)) ++bad;
        printf("shift %d: %d of 256 words wrong; lane 1 got %g %g %g %g (want %d..)\n", shift, bad, r[4], r[5], r[6], r[7], 5 + shift);
    }
    return 0;
}
