// Stand-alone bench / bit-comparison of the two stride-2 gather-up kernels (conv_mfma_up.hip is
// #included as is):  up_lab D1|D2|D3 [N] [cc]      (D1: 256->128 ch 8x8->16x16, D2: 128->64, D3: 64->32)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "../../behavenet_amd/csrc/conv_mfma_up.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const char* layer = argc > 1 ? argv[1] : "D2";
    const int N = argc > 2 ? atoi(argv[2]) : 256;
    const int cc = argc > 3 ? atoi(argv[3]) : 4;
    const int use_dact = argc > 4 ? atoi(argv[4]) : 0;
    BnGeom g; g.CsS = 0;
    g.N = N; g.R = g.S = 5; g.stride = 2; g.pt = 1; g.pl = 1;
    if (!strcmp(layer, "D1")) { g.Cs = 256; g.Hs = g.Ws = 8; g.Cb = 128; }
    else if (!strcmp(layer, "D2")) { g.Cs = 128; g.Hs = g.Ws = 16; g.Cb = 64; }
    else { g.Cs = 64; g.Hs = g.Ws = 32; g.Cb = 32; }
    g.Hb = 2 * g.Hs; g.Wb = 2 * g.Ws;
    const size_t nb = (size_t)N * g.Cb * g.Hb * g.Wb, ns = (size_t)N * g.Cs * g.Hs * g.Ws, nw = (size_t)g.Cs * g.Cb * 25;
    std::vector<float> hs(ns), hw(nw), hbias(g.Cb), hd(nb);
    srand(1);
    for (auto& v : hs) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (auto& v : hbias) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hd) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *ds, *dw, *dbias, *o0, *o1, *dd;
    CK(hipMalloc(&ds, ns * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dbias, g.Cb * 4)); CK(hipMalloc(&o0, nb * 4)); CK(hipMalloc(&o1, nb * 4)); CK(hipMalloc(&dd, nb * 4));
    CK(hipMemcpy(ds, hs.data(), ns * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), g.Cb * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemset(o0, 0xff, nb * 4)); CK(hipMemset(o1, 0xff, nb * 4));
    BnFastPlan pnew = bn_fast_up_plan(g);
    BnFastPlan pold = pnew; pold.variant = 0; pold.c = 4; pold.kernel_name = "k_up_mfma<1, 4>";
    pnew.c = cc;
    printf("%s N=%d: plan %s variant %d a=%d c=%d\n", layer, N, pnew.kernel_name, pnew.variant, pnew.a, pnew.c);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * N * g.Cs * g.Hs * g.Ws * g.Cb * 25;
    const float* dsrc = use_dact ? dd : nullptr;
    for (int which = 0; which < 2; ++which) {
        std::vector<float> ts;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0, st));
            int rc = bn_launch_up_fast(which ? pnew : pold, ds, dw, dbias, which ? o1 : o0, dsrc, g, BN_ACT_LRELU, use_dact ? BN_ACT_LRELU : 0, 0.05f, nullptr, st);
            if (rc) { printf("launch failed %d\n", rc); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("  %s: median %.1f us (min %.1f)  %.1f TFLOP/s\n", which ? "new (up2)" : "old (up) ", ts[ts.size() / 2], ts[0], flop / ts[ts.size() / 2] / 1e6);
    }
    std::vector<float> h0(nb), h1(nb);
    CK(hipMemcpy(h0.data(), o0, nb * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, nb * 4, hipMemcpyDeviceToHost));
    size_t diff = 0; double maxd = 0;
    for (size_t i = 0; i < nb; ++i) if (memcmp(&h0[i], &h1[i], 4)) { ++diff; maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i])); }
    printf("  outputs: %zu of %zu words differ (max |d| %.3g)\n", diff, nb, maxd);
    return 0;
}
