// Stand-alone bench / comparison of the two generations of the stride-5 quadrant GEMMs
// (conv_qgemm.hip, conv_qgemm2.hip #included as they are):  qg_lab up|wgrad|down [N] [iters]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../../behavenet_amd/csrc/conv_qgemm.hip"
#include "../../behavenet_amd/csrc/conv_qgemm2.hip"
bool bn_prof_take_dispatch_events(hipEvent_t*, hipEvent_t*) { return false; }

// calibration: the MFMA count of one role (1024 per wave, four accumulators, one wave per SIMD, 256
// workgroups) with random operands from LDS and nothing else: the ceiling under the lab's clocks
__global__ __launch_bounds__(256, 1) void k_cal(const float* src, float* dst, int reps) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    q2x16 acc[4];
    for (int z = 0; z < 4; ++z) for (int t = 0; t < 16; ++t) acc[z][t] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float a = lds[(lane + 64 * t) & 4095], b = lds[(lane * 3 + 64 * t + 1) & 4095];
#pragma unroll
            for (int z = 0; z < 4; ++z) acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[z], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int z = 0; z < 4; ++z) for (int t = 0; t < 16; ++t) s += acc[z][t];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev(const std::vector<float>& h) {
    float* d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return d;
}
static std::vector<float> rnd(size_t n, float scale) {
    std::vector<float> v(n); for (auto& x : v) x = ((rand() / (float)RAND_MAX) - 0.5f) * scale; return v;
}
static void compare(const char* what, const float* a, const float* b, size_t n) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; size_t bad = 0, nan = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!(ha[i] == ha[i]) || !(hb[i] == hb[i])) { ++nan; continue; }
        const double d = fabs((double)ha[i] - hb[i]); maxd = std::max(maxd, d); maxv = std::max(maxv, fabs((double)ha[i]));
        if (d > 1e-4 * (1.0 + fabs((double)ha[i]))) ++bad;
    }
    printf("  %s: max |old - new| %.3g (max |old| %.3g), %zu of %zu beyond 1e-4, %zu NaN\n", what, maxd, maxv, bad, n, nan);
}

int main(int argc, char** argv) {
    const char* role = argc > 1 ? argv[1] : "up";
    const int N = argc > 2 ? atoi(argv[2]) : 256;
    const int iters = argc > 3 ? atoi(argv[3]) : 30;
    BnGeom g; g.CsS = 0; g.N = N; g.Cs = 512; g.Hs = g.Ws = 2; g.Cb = 256; g.Hb = g.Wb = 8; g.R = g.S = 5; g.stride = 5; g.pt = g.pl = 1;
    srand(3);
    const size_t n_small = (size_t)N * g.Cs * 4, n_big = (size_t)N * g.Cb * 64, n_w = (size_t)g.Cs * g.Cb * 25;
    float* small = dev(rnd(n_small, 1.f)); float* big = dev(rnd(n_big, 1.f)); float* w = dev(rnd(n_w, 0.05f));
    float* bias_b = dev(rnd(g.Cb, 0.1f)); float* bias_s = dev(rnd(g.Cs, 0.1f));
    float* dsrc_b = dev(rnd(n_big, 1.f)); float* dsrc_s = dev(rnd(n_small, 1.f));
    const size_t ws_bytes = std::max(std::max(bn_qgemm_ws_bytes(0, g), bn_qgemm_ws_bytes(1, g)), bn_qgemm_ws_bytes(2, g)) + (1 << 20);
    void *ws0, *ws1; CK(hipMalloc(&ws0, ws_bytes)); CK(hipMalloc(&ws1, ws_bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const double flop = 2.0 * N * g.Cs * g.Cb * 64;
    { float* cd; CK(hipMalloc(&cd, 256 * 256 * 4));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_cal, dim3(256), dim3(256), 0, st, small, cd, 16);
      CK(hipStreamSynchronize(st)); }
    float *o0 = nullptr, *o1 = nullptr; size_t n_out = 0;
    float *db0 = nullptr, *db1 = nullptr; size_t n_db = 0;
    // variant: 0 = plain (bias + lrelu for fwd roles), 1 = data-gradient epilogue (dact)
    for (int variant = 0; variant < (strcmp(role, "wgrad") ? 2 : 3); ++variant) {

        for (int which = 0; which < 2; ++which) {
            std::vector<float> ts;
            for (int i = 0; i < iters; ++i) {
                int rc = 0;
                if (!strcmp(role, "up")) {
                    n_out = n_big;
                    if (!o0) { CK(hipMalloc(&o0, n_out * 4)); CK(hipMalloc(&o1, n_out * 4)); }
                    float* o = which ? o1 : o0;
                    CK(hipEventRecord(e0, st));
                    if (which == 0) rc = bn_launch_qgemm_up(small, w, variant ? nullptr : bias_b, o, variant ? dsrc_b : nullptr, g, variant ? 0 : 1, variant ? 1 : 0, 0.05f, ws0, st);
                    else rc = bn_launch_qg2_up(small, w, variant ? nullptr : bias_b, o, variant ? dsrc_b : nullptr, g, variant ? 0 : 1, variant ? 1 : 0, 0.05f, st);
                    CK(hipEventRecord(e1, st));
                }
#if 1
                else if (!strcmp(role, "wgrad")) {
                    n_out = n_w; n_db = g.Cs;
                    if (!o0) { CK(hipMalloc(&o0, n_out * 4)); CK(hipMalloc(&o1, n_out * 4)); CK(hipMalloc(&db0, 4096 * 4)); CK(hipMalloc(&db1, 4096 * 4)); }
                    float* o = which ? o1 : o0;
                    CK(hipMemsetAsync(o, 0, n_out * 4, st));
                    CK(hipEventRecord(e0, st));
                    if (which == 0) rc = bn_launch_qgemm_wgrad(small, big, o, g, 0, ws0, st);
                    else rc = bn_launch_qg2_wgrad(small, big, o, g, 0, variant == 1 ? db0 : db1, variant, st);
                    CK(hipEventRecord(e1, st));
                }
#endif
#ifdef QG2_HAVE_DOWN
                else if (!strcmp(role, "down")) {
                    n_out = n_small;
                    if (!o0) { CK(hipMalloc(&o0, n_out * 4)); CK(hipMalloc(&o1, n_out * 4)); }
                    float* o = which ? o1 : o0;
                    CK(hipEventRecord(e0, st));
                    if (which == 0) rc = bn_launch_qgemm_down(big, w, variant ? nullptr : bias_s, o, variant ? dsrc_s : nullptr, g, variant ? 0 : 1, variant ? 1 : 0, 0.05f, ws0, st);
                    else rc = bn_launch_qg2_down(big, w, variant ? nullptr : bias_s, o, variant ? dsrc_s : nullptr, g, variant ? 0 : 1, variant ? 1 : 0, 0.05f, ws1, st);
                    CK(hipEventRecord(e1, st));
                }
#endif
                else { printf("role %s not built\n", role); return 1; }
                if (rc) { printf("launch failed %d\n", rc); return 1; }
                if (getenv("QG_SYNC")) {
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 3) ts.push_back(ms * 1e3f);
                } else if (i == 2) CK(hipEventRecord(e2, st));     // back to back: total time of launches 3..
            }
            if (!getenv("QG_SYNC")) {
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e2, e1)); ts.push_back(ms * 1e3f / (iters - 3));
            }
            std::sort(ts.begin(), ts.end());
            printf("%s N=%d variant %d %s: median %.1f us (min %.1f)  %.1f TFLOP/s executed\n", role, N, variant, which ? "new" : "old",
                   ts[ts.size() / 2], ts[0], flop / ts[ts.size() / 2] / 1e6);
        }
        compare("output", o0, o1, n_out);
        if (!strcmp(role, "wgrad") && variant > 0) {
            // bias gradients against host sums (variant 0: small side, variant 1: big side)
            std::vector<float> hs(n_small), hb(n_big), got(512);
            CK(hipMemcpy(hs.data(), small, n_small * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), big, n_big * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), variant == 1 ? db0 : db1, 512 * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            if (variant == 1) for (int m = 0; m < g.Cs; ++m) { double r = 0; for (int n = 0; n < N; ++n) for (int z = 0; z < 4; ++z) r += hs[((size_t)n * g.Cs + m) * 4 + z]; worst = std::max(worst, fabs(r - got[m])); }
            else for (int c = 0; c < g.Cb; ++c) { double r = 0; for (int n = 0; n < N; ++n) for (int p = 0; p < 64; ++p) r += hb[((size_t)n * g.Cb + c) * 64 + p]; worst = std::max(worst, fabs(r - got[c])); }
            printf("  bias side %d: max |host - device| %.3g\n", variant, worst);
        }
    }
    return 0;
}
