// Stand-alone laboratory for enc.conv0 forward (conv_edge.hip is #included as it is):
//   e0_lab [variant 0|1|2|3 ...]      0 = k_down_c1, 1 / 2 = k_down_c1s with 2- / 4-row units, 3 = k_down_c1w
// 256 frames, 1x128x128 -> 32x64x64, LeakyReLU.  Every timed launch follows a kernel that streams
// 140 MB of plain stores through the L2s (what Adam leaves behind in the training step); times are
// the dispatch-attached event intervals bench.py uses.  The outputs of all variants are compared
// word for word with variant 0.  Built with -DE0_TRACE the kernels also leave per-wave
// s_memrealtime marks (start, first patch in LDS, first row multiplied, first row stored, first
// unit done, all stores issued, all stores acknowledged): the ramp of the store stream.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBN_TUNING [-DE0_TRACE] tools/lab/e0_lab.hip -o tools/lab/bin/e0_lab
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#ifdef E0_TRACE
__device__ unsigned long long* e0_trace;
#endif
#include "../../behavenet_amd/csrc/conv_edge.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// what the library provides elsewhere
__global__ __launch_bounds__(1024) void k_sum_partials(const float* __restrict__, float* __restrict__, int, int, int, int, int, int, int) {}
__global__ __launch_bounds__(256) void k_sum_partials_pair(const float* __restrict__, float* __restrict__, int, int, int, const float* __restrict__, float* __restrict__, int, int, int, int) {}
static hipEvent_t g_e0, g_e1;
bool bn_prof_take_dispatch_events(hipEvent_t* e0, hipEvent_t* e1) { *e0 = g_e0; *e1 = g_e1; return true; }

__global__ __launch_bounds__(256) void k_dirty(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// the same stream with write-through (sc1) stores: nothing stays dirty in the L2s
__global__ __launch_bounds__(256) void k_dirty_wt(float4* p, size_t n4) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffffc, 0x00020000);
    const uintx4e d = {1u, 2u, 3u, 4u};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)(i * 16), 0, 16);
}

// E0_HEAT=<iterations>: a dense fp32-MFMA kernel on every SIMD in front of each timed pair, with random
// operands -- the power state the training step leaves the chip in (its clock settles near 2.05 GHz
// under the stride-2 layers, against 2.4 GHz for a kernel that follows idle time)
__global__ __launch_bounds__(256) void k_heat(float* sink, int iters, float seed) {
    floatx16 acc[4];
    for (int h = 0; h < 4; ++h) for (int e = 0; e < 16; ++e) acc[h][e] = 0.f;
    float a = seed + threadIdx.x * 0.37f, b = seed - threadIdx.x * 0.11f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[h], 0, 0, 0);
        a = a * 1.0001f + 0.001f; b = b * 0.9999f - 0.001f;
    }
    float t = 0.f;
    for (int h = 0; h < 4; ++h) for (int e = 0; e < 16; ++e) t += acc[h][e];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

int main(int argc, char** argv) {
    const int N = 256;
    BnGeom g; g.CsS = 0;
    g.N = N; g.R = g.S = 5; g.stride = 2; g.pt = 1; g.pl = 1;
    g.Cs = 32; g.Hs = g.Ws = 64; g.Cb = 1; g.Hb = g.Wb = 128;
    const size_t nb = (size_t)N * 128 * 128, ns = (size_t)N * 32 * 64 * 64;
    std::vector<float> hb(nb), hw(32 * 25), hbias(32);
    srand(1);
    for (auto& v : hb) v = (float)(rand() % 256) / 255.f;
    for (auto& v : hw) v = ((rand() / (float)RAND_MAX) - 0.5f) * 0.4f;
    for (auto& v : hbias) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *db, *dw, *dbias, *o_ref, *o; float4* dirty;
    const size_t dirty_n4 = 140u * 1000 * 1000 / 16;
    // E0_RING=K (default 20): K copies of the input and 3 output buffers, rotated per launch, so that the
    // 256 MB Infinity Cache serves neither the reads nor the writes (in the training step 20 trials of
    // 16.8 MB and gigabytes of other traffic pass between two launches)
    const int ring = getenv("E0_RING") ? atoi(getenv("E0_RING")) : 20;
    CK(hipMalloc(&db, nb * 4 * ring)); CK(hipMalloc(&dw, 800 * 4)); CK(hipMalloc(&dbias, 32 * 4));
    CK(hipMalloc(&o_ref, ns * 4 * 3)); CK(hipMalloc(&o, ns * 4)); CK(hipMalloc(&dirty, dirty_n4 * 16));
    for (int r = 0; r < ring; ++r) CK(hipMemcpy(db + r * nb, hb.data(), nb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), 800 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), 32 * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipEventCreate(&g_e0)); CK(hipEventCreate(&g_e1));
#ifdef E0_TRACE
    const size_t trace_words = 8192 * 8;
    unsigned long long* dtrace; CK(hipMalloc(&dtrace, trace_words * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(e0_trace), &dtrace, sizeof(dtrace)));
#endif
    std::vector<float> h_ref(ns), h(ns);
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
        const char* variant = argc > 1 ? argv[a] : "0";
        setenv("BN_E0_V", variant, 1);
        float* out = a == 1 ? o_ref : o;
        CK(hipMemsetAsync(out, 0xff, ns * 4, st));
        std::vector<float> ts;
        for (int i = 0; i < 23; ++i) {
#ifdef E0_TRACE
            CK(hipMemsetAsync(dtrace, 0, trace_words * 8, st));
#endif
            // E0_DIRTY: what runs before the timed launch.  1 (default) 140 MB of plain stores, 0
            // nothing, 2 the same bytes write-through, 3 35 MB of plain stores, 4 hipMemsetAsync of
            // 35 MB (the optimizer's zero_grad in the training step)
            static const int dmode = getenv("E0_DIRTY") ? atoi(getenv("E0_DIRTY")) : 1;
            static const int heat = getenv("E0_HEAT") ? atoi(getenv("E0_HEAT")) : 0;
            if (heat) hipLaunchKernelGGL(k_heat, dim3(512), dim3(256), 0, st, (float*)dirty, heat, 0.5f + i);
            if (dmode == 1) hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, st, dirty, dirty_n4);
            else if (dmode == 2) hipLaunchKernelGGL(k_dirty_wt, dim3(2048), dim3(256), 0, st, dirty, dirty_n4);
            else if (dmode == 3) hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, st, dirty, dirty_n4 / 4);
            else if (dmode == 4) CK(hipMemsetAsync(dirty, 0, 35u * 1000 * 1000, st));
            // (the last launch writes `out`, which is compared; the others rotate)
            float* dst = i == 22 ? out : o_ref + (size_t)(1 + i % 2) * ns;
            int rc = bn_launch_edge_down(db + (size_t)(i % ring) * nb, dw, dbias, dst, nullptr, g, BN_ACT_LRELU, BN_ACT_NONE, 0.05f, st, nullptr);
            if (rc) { printf("launch failed %d\n", rc); return 1; }
            CK(hipEventSynchronize(g_e1));
            float ms; CK(hipEventElapsedTime(&ms, g_e0, g_e1));
            if (i >= 3) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        const double bytes = (double)N * (65536 + 524288);
        printf("variant %s: median %.2f us (min %.2f, max %.2f)  %.2f TB/s = %.3f of 8 TB/s\n", variant, ts[ts.size() / 2],
               ts[0], ts.back(), bytes / ts[ts.size() / 2] / 1e6, bytes / ts[ts.size() / 2] / 1e6 / 8.0);
        CK(hipMemcpy(a == 1 ? h_ref.data() : h.data(), out, ns * 4, hipMemcpyDeviceToHost));
        if (a > 1) {
            size_t diff = 0;
            for (size_t i = 0; i < ns; ++i) if (memcmp(&h_ref[i], &h[i], 4)) ++diff;
            printf("  output vs first variant: %zu of %zu words differ\n", diff, ns);
            if (diff) {      // where: (frame, row) pairs with differences, and a sample
                int shown = 0;
                for (int n = 0; n < N && shown < 12; ++n)
                    for (int p = 0; p < 64 && shown < 12; ++p) {
                        size_t cnt = 0, first = 0;
                        for (int c = 0; c < 32; ++c)
                            for (int q = 0; q < 64; ++q) {
                                const size_t i = (((size_t)n * 32 + c) * 64 + p) * 64 + q;
                                if (memcmp(&h_ref[i], &h[i], 4)) { if (!cnt) first = i; ++cnt; }
                            }
                        if (cnt) { printf("    frame %d row %d: %zu words, e.g. [%zu] %g vs %g\n", n, p, cnt, first, h_ref[first], h[first]); ++shown; }
                    }
            }
        } else {
            double cs = 0; for (size_t i = 0; i < ns; i += 997) cs += h_ref[i];
            printf("  checksum %.6f\n", cs);
        }
#ifdef E0_TRACE
        std::vector<unsigned long long> tr(trace_words);
        CK(hipMemcpy(tr.data(), dtrace, trace_words * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; int waves = 0;
        for (size_t w = 0; w < 8192; ++w) if (tr[w * 8]) { t0 = std::min(t0, tr[w * 8]); ++waves; }
        static const char* names[7] = {"wave start", "first patch in LDS", "first row multiplied", "first row stored",
                                       "first unit done", "all stores issued", "all stores acknowledged"};
        printf("  trace of the last launch, %d waves, us after the first wave's start (min / 10%% / median / 90%% / max):\n", waves);
        for (int m = 0; m < 7; ++m) {
            std::vector<double> v;
            for (size_t w = 0; w < 8192; ++w) if (tr[w * 8] && tr[w * 8 + m]) v.push_back((double)(tr[w * 8 + m] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("    %-24s %6.2f %6.2f %6.2f %6.2f %6.2f\n", names[m], v[0], v[v.size() / 10], v[v.size() / 2],
                   v[v.size() * 9 / 10], v.back());
        }
#endif
    }
    return 0;
}
