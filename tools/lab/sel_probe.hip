#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    const int lane = threadIdx.x, li = lane & 31;
    const int e = (li & 3) + 4 * (li >> 3), half = (li >> 2) & 1;
    int r = -1, sx = 0;
    if (e < 5) { r = e; sx = half ? 2 : 1; }
    else if (e < 10) { r = e - 5; sx = half ? 4 : 3; }
    else if (e < 15 && half) { r = e - 10; sx = 0; }
    o[lane] = r * 10 + sx + 1000 * half + 100000 * e;
}
int main() {
    int* d; hipMalloc(&d, 64 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 32; ++l) printf("row %2d: %d\n", l, h[l]);
    return 0;
}
