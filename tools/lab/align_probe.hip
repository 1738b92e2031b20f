// Does a raw buffer load of 8 / 16 bytes honour a 4-byte aligned offset on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* src, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 4096, 0x00020000);
    const int off = (blockIdx.x ? 16 : 4) * (threadIdx.x & 7);
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    const u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, off, 64, 0);
    out[threadIdx.x * 6 + 0] = __builtin_bit_cast(float, v.x); out[threadIdx.x * 6 + 1] = __builtin_bit_cast(float, v.y);
    out[threadIdx.x * 6 + 2] = __builtin_bit_cast(float, v.z); out[threadIdx.x * 6 + 3] = __builtin_bit_cast(float, v.w);
    out[threadIdx.x * 6 + 4] = __builtin_bit_cast(float, w.x); out[threadIdx.x * 6 + 5] = __builtin_bit_cast(float, w.y);
}
int main() {
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 64 * 6 * 4); hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o); hipDeviceSynchronize(); { float r2[64*6]; hipMemcpy(r2, o, sizeof(r2), hipMemcpyDeviceToHost); } hipLaunchKernelGGL(k, dim3(2), dim3(64), 0, 0, s, o + 0);
    float r[64 * 6]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t) printf("offset %2d B: b128 -> %g %g %g %g   b64(+64) -> %g %g\n", 4 * t, r[t*6], r[t*6+1], r[t*6+2], r[t*6+3], r[t*6+4], r[t*6+5]);
    return 0;
}
