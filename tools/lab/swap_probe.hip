// v_permlane32_swap semantics through the clang builtin: prints what lanes 0, 5, 32, 37 hold afterwards
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    const unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
    const uintx2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 128 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 5, 32, 37}) printf("lane %2d: r.x = %u  r.y = %u   (a = 1000 + lane, b = 2000 + lane)\n", l, h[l], h[64 + l]);
    return 0;
}
