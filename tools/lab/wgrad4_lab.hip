// Stand-alone bench / bit-comparison of the two stride-2 weight-gradient kernels
// (conv_mfma_wgrad4.hip is #included as is):  wgrad4_lab E1|E2|E3 [N] [bias_side]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../behavenet_amd/csrc/conv_mfma_wgrad4.hip"
__global__ __launch_bounds__(1024) void k_sum_partials(const float*, float*, int, int, int, int, int, int, int) {}
__global__ __launch_bounds__(256) void k_sum_partials_pair(const float*, float*, int, int, int, const float*, float*, int, int, int, int) {}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int L>
static int run_old(dim3 grid, size_t lds, hipStream_t st, const float* s, const float* b, float* part, float* bp, const BnGeom& g, const Wgrad4Tile& t) {
    return launch_wgrad4<L>(grid, lds, st, s, b, part, bp, g, t);
}
template <int L>
static int run_new(int bias, dim3 grid, hipStream_t st, const float* s, const float* b, float* part, float* bp, const BnGeom& g, const Wgrad4Tile& t, int lg_tpf) {
    if (bias == 0) return launch_wgrad4s<L, 0>(grid, st, s, b, part, bp, g, t.n_stages, t.splits, lg_tpf, t.nbias);
    if (bias == 1) return launch_wgrad4s<L, 1>(grid, st, s, b, part, bp, g, t.n_stages, t.splits, lg_tpf, t.nbias);
    return launch_wgrad4s<L, 2>(grid, st, s, b, part, bp, g, t.n_stages, t.splits, lg_tpf, t.nbias);
}

int main(int argc, char** argv) {
    const char* layer = argc > 1 ? argv[1] : "E2";
    const int N = argc > 2 ? atoi(argv[2]) : 256;
    const int bias = argc > 3 ? atoi(argv[3]) : 1;
    BnGeom g; g.CsS = 0;
    g.N = N; g.R = g.S = 5; g.stride = 2; g.pt = 1; g.pl = 1;
    if (!strcmp(layer, "E1")) { g.Cb = 32; g.Hb = g.Wb = 64; g.Cs = 64; }
    else if (!strcmp(layer, "E2")) { g.Cb = 64; g.Hb = g.Wb = 32; g.Cs = 128; }
    else { g.Cb = 128; g.Hb = g.Wb = 16; g.Cs = 256; }
    g.Hs = g.Hb / 2; g.Ws = g.Wb / 2;
    const size_t nb = (size_t)N * g.Cb * g.Hb * g.Wb, ns = (size_t)N * g.Cs * g.Hs * g.Ws;
    std::vector<float> hb(nb), hs(ns);
    srand(1);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    for (auto& v : hs) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *db, *dsm; CK(hipMalloc(&db, nb * 4)); CK(hipMalloc(&dsm, ns * 4));
    CK(hipMemcpy(db, hb.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsm, hs.data(), ns * 4, hipMemcpyHostToDevice));
    BnFastPlan plan = bn_wgrad4_plan(g);
    if (!plan.supported) { printf("unsupported\n"); return 1; }
    Wgrad4Tile t; size_t lds = 0; wgrad4_tile(g, &t, &lds);
    t.splits = plan.d; t.bias_side = bias; t.nbias = bias == 1 ? g.Cs : g.Cb;
    const size_t npart = (size_t)t.splits * 25 * g.Cs * g.Cb, nbp = (size_t)t.splits * std::max(g.Cs, g.Cb);
    float *p0, *p1, *bp0, *bp1;
    CK(hipMalloc(&p0, npart * 4)); CK(hipMalloc(&p1, npart * 4)); CK(hipMalloc(&bp0, nbp * 4)); CK(hipMalloc(&bp1, nbp * 4));
    CK(hipMemset(p0, 0xff, npart * 4)); CK(hipMemset(p1, 0xff, npart * 4)); CK(hipMemset(bp0, 0, nbp * 4)); CK(hipMemset(bp1, 0, nbp * 4));
    const int tiles = ((g.Cs + W4_TA - 1) / W4_TA) * ((g.Cb + W4_TB - 1) / W4_TB);
    dim3 grid(tiles, t.splits);
    const int lgq = ilog2_exact_w4(g.Ws), lg_tpf = ilog2_exact_w4(t.tiles_per_frame);
    printf("%s N=%d: kernel %s, tiles %d x splits %d, stages %d, wgrad4s_ok %d\n", layer, N, plan.kernel_name, tiles, t.splits, t.n_stages, (int)wgrad4s_ok(g, t));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * N * g.Cs * g.Hs * g.Ws * g.Cb * 25;
    for (int which = 0; which < 2; ++which) {
        std::vector<float> ts;
        for (int i = 0; i < 12; ++i) {
            CK(hipEventRecord(e0, st));
            int rc;
            if (which == 0) rc = lgq == 3 ? run_old<3>(grid, lds, st, dsm, db, p0, bp0, g, t) : lgq == 4 ? run_old<4>(grid, lds, st, dsm, db, p0, bp0, g, t) : run_old<5>(grid, lds, st, dsm, db, p0, bp0, g, t);
            else rc = lgq == 3 ? run_new<3>(bias, grid, st, dsm, db, p1, bp1, g, t, lg_tpf) : lgq == 4 ? run_new<4>(bias, grid, st, dsm, db, p1, bp1, g, t, lg_tpf) : run_new<5>(bias, grid, st, dsm, db, p1, bp1, g, t, lg_tpf);
            if (rc) { printf("launch failed %d\n", rc); return 1; }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("  %s: median %.1f us (min %.1f)  %.1f TFLOP/s\n", which ? "new (wgrad4s)" : "old (wgrad4) ", ts[ts.size() / 2], ts[0], flop / ts[ts.size() / 2] / 1e6);
    }
    std::vector<float> h0(npart), h1(npart), b0(nbp), b1(nbp);
    CK(hipMemcpy(h0.data(), p0, npart * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), p1, npart * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b0.data(), bp0, nbp * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b1.data(), bp1, nbp * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, bdiff = 0; double maxd = 0;
    for (size_t i = 0; i < npart; ++i) if (memcmp(&h0[i], &h1[i], 4)) { ++diff; maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i])); }
    for (size_t i = 0; i < nbp; ++i) if (memcmp(&b0[i], &b1[i], 4)) ++bdiff;
    printf("  partial tiles: %zu of %zu words differ (max |d| %.3g); bias partials: %zu of %zu differ\n", diff, npart, maxd, bdiff, nbp);
    return 0;
}
