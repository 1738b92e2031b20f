// Stand-alone bench of the stride-2 gather-down kernel (conv_mfma_down2.hip is #included as is):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DD2_TRACE] [-D...] tools/lab/down2_lab.hip -o /tmp/d2lab
//   /tmp/d2lab E2 256 [MR NR]
// With -DD2_TRACE the kernel records per-workgroup timestamps (s_memtime / s_memrealtime / HW_ID).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#ifdef D2_TRACE
#include <hip/hip_runtime.h>
#define D2_TRACE_SLOTS 32
__device__ unsigned long long d2_trace[8192 * D2_TRACE_SLOTS];
#endif
#include "../../behavenet_amd/csrc/conv_mfma_down2.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const char* layer = argc > 1 ? argv[1] : "E2";
    const int N = argc > 2 ? atoi(argv[2]) : 256;
    int MR = argc > 3 ? atoi(argv[3]) : 2, NR = argc > 4 ? atoi(argv[4]) : 2;
    BnGeom g; g.CsS = 0;
    g.N = N; g.R = g.S = 5; g.stride = 2; g.pt = 1; g.pl = 1;
    if (!strcmp(layer, "E1")) { g.Cb = 32; g.Hb = g.Wb = 64; g.Cs = 64; }
    else if (!strcmp(layer, "E2")) { g.Cb = 64; g.Hb = g.Wb = 32; g.Cs = 128; }
    else { g.Cb = 128; g.Hb = g.Wb = 16; g.Cs = 256; if (argc <= 4) { NR = 1; } }
    g.Hs = g.Hb / 2; g.Ws = g.Wb / 2;
    const size_t nb = (size_t)N * g.Cb * g.Hb * g.Wb, ns = (size_t)N * g.Cs * g.Hs * g.Ws;
    const size_t nw = (size_t)g.Cs * g.Cb * 25;
    std::vector<float> hb(nb), hw(nw), hbias(g.Cs);
    srand(1);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.05f;
    for (auto& v : hbias) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *db, *dw, *dbias, *dout;
    CK(hipMalloc(&db, nb * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dbias, g.Cs * 4));
    CK(hipMalloc(&dout, ns * 4));
    CK(hipMemcpy(db, hb.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), g.Cs * 4, hipMemcpyHostToDevice));
    if (!bn_down2_supported(g, MR, NR)) { printf("unsupported\n"); return 1; }
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 3; ++i)
        if (bn_launch_down2(MR, NR, db, dw, dbias, dout, nullptr, g, BN_ACT_LRELU, 0, 0.05f, st)) { printf("launch failed\n"); return 1; }
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20;
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, st));
        bn_launch_down2(MR, NR, db, dw, dbias, dout, nullptr, g, BN_ACT_LRELU, 0, 0.05f, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const double flop = 2.0 * N * g.Cs * g.Hs * g.Ws * g.Cb * 25;
    printf("%s N=%d MR=%d NR=%d: median %.1f us (min %.1f)  %.1f TFLOP/s\n", layer, N, MR, NR, ts[iters / 2], ts[0], flop / ts[iters / 2] / 1e6);
    // checksum of the output (compare variants)
    std::vector<float> ho(ns);
    CK(hipMemcpy(ho.data(), dout, ns * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (size_t i = 0; i < ns; ++i) cs += ho[i] * (double)((i % 97) + 1);
    printf("checksum %.10e\n", cs);
#ifdef D2_TRACE
    const int nwg = 512;
    std::vector<unsigned long long> tr((size_t)8192 * D2_TRACE_SLOTS);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(d2_trace), tr.size() * 8));
    if (getenv("D2_DUMP")) { FILE* f = fopen(getenv("D2_DUMP"), "wb"); fwrite(tr.data(), 8, (size_t)nwg * D2_TRACE_SLOTS, f); fclose(f); }
    // slots: 0 start clk, 1 start real, 2 hwid, 3 after prologue, 4.. chunk marks, 14 end clk, 15 end real
    unsigned long long t0 = ~0ull, r0 = ~0ull, t1 = 0, r1 = 0;
    for (int b = 0; b < nwg; ++b) { auto* p = &tr[(size_t)b * D2_TRACE_SLOTS]; t0 = std::min(t0, p[0]); r0 = std::min(r0, p[1]); t1 = std::max(t1, p[14]); r1 = std::max(r1, p[15]); }
    printf("kernel span: %llu clk, %llu real(100MHz) -> %.3f GHz, %.1f us\n", t1 - t0, r1 - r0, (double)(t1 - t0) / ((double)(r1 - r0) * 10.0), (r1 - r0) / 100.0);
    {   // shader clock seen by every workgroup that left marks: s_memtime ticks per s_memrealtime tick (100 MHz)
        std::vector<double> ghz;
        for (int b = 0; b < 8192; ++b) {
            auto* q = &tr[(size_t)b * D2_TRACE_SLOTS];
            if (q[0] && q[14] > q[0] && q[15] > q[1]) ghz.push_back((double)(q[14] - q[0]) / ((double)(q[15] - q[1]) * 10.0));
        }
        std::sort(ghz.begin(), ghz.end());
        if (!ghz.empty())
            printf("shader clock over the lifetime of %zu workgroups (s_memtime / s_memrealtime): min %.3f  median %.3f  max %.3f GHz\n",
                   ghz.size(), ghz[0], ghz[ghz.size() / 2], ghz.back());
    }
    for (int b = 0; b < nwg; b += 37) {
        auto* p = &tr[(size_t)b * D2_TRACE_SLOTS];
        printf("wg %3d hwid %08llx start %7llu prologue %6llu", b, p[2], p[0] - t0, p[3] - p[0]);
        for (int k = 4; k < 13; ++k) printf(" %6llu", p[k + 1] - p[k]);
        printf(" | bnd2: bar1 %5llu issueW %5llu wait %5llu bar2 %5llu issueX %5llu", p[23]-p[22], p[24]-p[23], p[25]-p[24], p[26]-p[25], p[27]-p[26]); printf(" | epi: bias %5llu blocks %5llu %5llu %5llu %5llu", p[17]-p[16], p[18]-p[17], p[19]-p[18], p[20]-p[19], p[21]-p[20]); printf(" | mainloop end %7llu  epilogue %6llu total %7llu\n", p[16] - p[0], p[14] - p[16], p[14] - p[0]);
    }
#endif
    return 0;
}
