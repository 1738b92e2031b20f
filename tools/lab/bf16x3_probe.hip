// Split-bf16 question, speed half: what would a gather-down kernel of the stride-2 middle layers reach
// if every fp32 operand were pre-split into three bf16 terms and a K-chunk were six
// v_mfma_f32_32x32x16_bf16 (x1 w1, x1 w2, x2 w1, x2 w2, x1 w3, x3 w1; fp32 accumulate)?
//
// NOT a convolution: an instruction-mix model of the kernel one would build,
// with the byte counts, instruction counts, LDS access patterns and barriers of that design and real
// (random) data everywhere, so that it can only be FASTER than the real thing:
//   * workgroup = 4 waves = 64 output channels x 128 output pixels (wave: 64 x 32 = two 32x32
//     accumulators), one workgroup per CU, persistent over its tiles;
//   * a stage = 8 input channels: K-chunk = 2 taps x 8 channels (a lane's 16 bytes = the 8 channels of
//     ONE term of one pixel, channel-last split layout [pixel][term][8 x bf16] = 48 B per pixel), 13
//     chunks per stage (25 taps in pairs: 4 % of the MFMA work multiplies zeros);
//   * per chunk and wave: 6 A reads (2 blocks x 3 terms, conflict-free) + 3 B reads (3 terms of the
//     stride-2 pixel gather: 96-byte lane stride = 2-way conflicts), all ds_read_b128, 12 MFMAs;
//   * LDS-DMA per stage and workgroup: the input patch (11 rows x 64 pixels x 48 B = 33 KB, double
//     buffered) and the weight slice of the stage (13 chunks x 64 m x 2 x 3 x 16 B = 78 KB, streamed in
//     three sub-stages of 5 / 4 / 4 chunks through three 30 KB buffers, requested two sub-stages
//     ahead): 31 pieces per wave per 156 MFMAs, issued inside the MFMA stream; three counted waits +
//     barriers per stage;
//   * per tile (every NCG stages): the fp32 output (32 KB) AND its three-term split for the next layer
//     (48 KB; the split arithmetic is done), written to HBM.
// Reported: time for the E1 / E2 / E3 workloads at 256 frames, TFLOP/s-equivalent (fp32 FLOPs of the
// layer / time), bytes moved.   bf16x3_probe [E1|E2|E3] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define PATCH_BYTES (33 * 1024)
#define WSUB_BYTES (30 * 1024)
#define LDS_BYTES (2 * PATCH_BYTES + 3 * WSUB_BYTES)

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int vo, int so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (float*)lds, 16, vo, so, 0, 0);
}
__device__ __forceinline__ unsigned short bf(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

struct Args {
    const char* xs; size_t xs_bytes;     // split input, walked linearly tile by tile
    const char* ws; int ncg;             // split weights [cg][78 KB]
    float* out; char* outs;              // fp32 output and its split
    int tiles_per_wg;
};

__global__ __launch_bounds__(256, 1) void k_probe(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.xs, 0, (int)a.xs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws, 0, a.ncg * 78 * 1024, 0x00020000);
#define PBUF(i) (smem + (i) * PATCH_BYTES)
#define WBUF(i) (smem + 2 * PATCH_BYTES + (i) * WSUB_BYTES)

    f32x16 acc[2];
    for (int b = 0; b < 2; ++b) for (int t = 0; t < 16; ++t) acc[b][t] = 0.f;

    // B reads: pixel (row 2 wv + r, column 2 li + s) of the patch, 48 B per pixel, 64-pixel rows;
    // vertical tap pairs (kh = second row), the last kernel row in horizontal pairs
    const int bV = ((2 * wv) * 64 + 2 * li) * 48 + kh * 64 * 48;
    const int bH = ((2 * wv + 4) * 64 + 2 * li) * 48 + kh * 48;
    // A reads: [chunk][term][kh][m 64] groups of 16 B
    const int aB = (kh * 64 + li) * 16;

    const int n_stages = a.tiles_per_wg * a.ncg;
    const size_t tile_bytes = (size_t)a.ncg * PATCH_BYTES;
    const int wg_x0 = (int)(((size_t)blockIdx.x * a.tiles_per_wg * tile_bytes) % (a.xs_bytes - (size_t)n_stages * PATCH_BYTES - 4096));
    // DMA schedule (every wave the same counts, so that the waits can be counted): a stage has three
    // sub-stages g = 3 st + u of 5 / 4 / 4 chunks; the weights of sub-stage g are read from weight
    // buffer g % 3 and were requested during sub-stage g - 2 (7 pieces per wave = 28 KB >= the 26 KB
    // average), the next stage's patch (33 KB) during sub-stages 0 and 1 (5 pieces per wave each):
    // 12 / 12 / 7 pieces per wave and sub-stage, all in its first chunks.  At the head of a sub-stage
    // only the pieces of the sub-stage before may still be in flight.
    auto issue = [&](const int i, const int g, const int st, const int u) __attribute__((always_inline)) {
        if (i < 7) {
            const int piece = wv + 4 * i;                                    // of the 28 of sub-stage g + 2
            const int g2 = g + 2, st2 = g2 / 3, u2 = g2 - 3 * st2;
            dma16(rw, WBUF(g2 % 3) + piece * 1024, lane * 16, (st2 % a.ncg) * 78 * 1024 + (u2 * 26 + piece) * 1024 % (78 * 1024));
        } else {
            const int piece = wv + 4 * (i - 7) + 20 * u;                     // of the 40 (33 used) of the patch
            dma16(rx, PBUF((st + 1) & 1) + (piece % 33) * 1024, lane * 16, wg_x0 + (st + 1) * PATCH_BYTES + (piece % 33) * 1024);
        }
    };
    // prologue: patch 0, weights of sub-stages 0 and 1
    for (int d = wv; d < 33; d += 4) dma16(rx, PBUF(0) + d * 1024, lane * 16, wg_x0 + d * 1024);
    for (int d = wv; d < 28; d += 4) dma16(rw, WBUF(0) + d * 1024, lane * 16, d * 1024);
    for (int d = wv; d < 28; d += 4) dma16(rw, WBUF(1) + d * 1024, lane * 16, (26 + d) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int tile = 0;
    for (int st = 0; st < n_stages; ++st) {
        const int pb = st & 1;
        bf16x8 av[2][2][3], bv[2][3];
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            const int g = 3 * st + sub;
            const int wsel = g % 3;
            auto load = [&](const int slot, const int chunk, const int csub) __attribute__((always_inline)) {
                const char* wb = WBUF(wsel) + aB;
                const char* pp = PBUF(pb);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
                        av[slot][mb][t] = *reinterpret_cast<const bf16x8*>(wb + ((csub * 3 + t) * 2 * 64 + mb * 32) * 16);
                    const int off = (chunk < 10) ? bV + ((chunk / 5) * 2 * 64 + (chunk % 5)) * 48
                                                 : bH + ((chunk - 10) * 2) * 48;
                    bv[slot][t] = *reinterpret_cast<const bf16x8*>(pp + off + t * 16);
                }
            };
            // the pieces of the sub-stage before may be in flight: 7 behind a sub-stage 2, else 12
#ifndef ABL_NODMA
            if (sub == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#endif
            __builtin_amdgcn_s_barrier();
            const int c0 = sub == 0 ? 0 : (sub == 1 ? 5 : 9), nc = sub == 0 ? 5 : 4;
            const int npieces = sub < 2 ? 12 : 7;
            int ip = 0;
            load(0, c0, 0);
#pragma unroll
            for (int cs = 0; cs < 5; ++cs) {
                if (cs >= nc) break;
                const int cu = cs & 1;
#ifndef ABL_NOLDS
                if (cs + 1 < nc) load(cu ^ 1, c0 + cs + 1, cs + 1);
#else
                if (cs == 0) load(1, c0 + 1, 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
                // six products per block, the two blocks alternating; one DMA piece behind every pair
                constexpr int ta[6] = {0, 0, 1, 1, 0, 2}, tb[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cu][0][ta[p]], bv[cu][tb[p]], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cu][1][ta[p]], bv[cu][tb[p]], acc[1], 0, 0, 0);
#ifndef ABL_NODMA
                    if (cs * 6 + p < npieces) { issue(cs * 6 + p, g, st, sub); ++ip; }
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            (void)ip;
        }
        if ((st + 1) % a.ncg == 0) {
            // tile epilogue: lane = pixel li of output row wv, registers = channels; fp32 NCHW + the
            // three-term split, channel-last (4 consecutive channels of a term = 8 bytes)
            const size_t t_glob = (size_t)blockIdx.x * a.tiles_per_wg + tile;
            float* o = a.out + t_glob * (64 * 128) + wv * 32 + li;
            char* os = a.outs + t_glob * (64 * 128 * 6) + (size_t)(wv * 32 + li) * (64 * 6);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    unsigned short s1[4], s2[4], s3[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[mb][4 * tq + e];
                        v = v > 0.f ? v : v * 0.05f;
                        const int m = mb * 32 + e + 8 * tq + 4 * kh;
                        o[m * 128] = v;
                        const __bf16 h1 = (__bf16)v; const float r1 = v - (float)h1;
                        const __bf16 h2 = (__bf16)r1; const float r2 = r1 - (float)h2;
                        s1[e] = __builtin_bit_cast(unsigned short, h1); s2[e] = __builtin_bit_cast(unsigned short, h2);
                        s3[e] = bf(r2);
                        acc[mb][4 * tq + e] = 0.f;
                    }
                    const int mg = (mb * 32 + 8 * tq + 4 * kh) / 8, half = ((8 * tq + 4 * kh) & 4) ? 8 : 0;
                    u32x2 w1 = {(unsigned)s1[0] | ((unsigned)s1[1] << 16), (unsigned)s1[2] | ((unsigned)s1[3] << 16)};
                    u32x2 w2 = {(unsigned)s2[0] | ((unsigned)s2[1] << 16), (unsigned)s2[2] | ((unsigned)s2[3] << 16)};
                    u32x2 w3 = {(unsigned)s3[0] | ((unsigned)s3[1] << 16), (unsigned)s3[2] | ((unsigned)s3[3] << 16)};
                    *reinterpret_cast<u32x2*>(os + mg * 48 + 0 + half) = w1;
                    *reinterpret_cast<u32x2*>(os + mg * 48 + 16 + half) = w2;
                    *reinterpret_cast<u32x2*>(os + mg * 48 + 32 + half) = w3;
                }
            ++tile;
        }
    }
}

int main(int argc, char** argv) {
    const char* layer = argc > 1 ? argv[1] : "E1";
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    int cin, cout, hw;
    if (!strcmp(layer, "E1")) { cin = 32; cout = 64; hw = 64; }
    else if (!strcmp(layer, "E2")) { cin = 64; cout = 128; hw = 32; }
    else { cin = 128; cout = 256; hw = 16; }
    const int N = 256, ncg = cin / 8;
    const int px_out = (hw / 2) * (hw / 2);
    const long tiles = (long)N * px_out / 128 * (cout / 64);
    const int tiles_per_wg = (int)(tiles / 256);
    const double flop = 2.0 * N * px_out * cout * cin * 25;
    // split input: N x cin x hw x hw x 6 B (walked linearly; at least the patches of all stages)
    size_t xs_bytes = (size_t)N * cin * hw * hw * 6;
    const size_t need = (size_t)256 * tiles_per_wg * ncg * PATCH_BYTES + (1 << 20);
    // (a layer with several channel tiles re-reads its input once per tile: those reads hit the caches)
    if (xs_bytes > 0x7ff00000ull) xs_bytes = 0x7ff00000ull;
    std::vector<unsigned short> h(16 << 20);
    srand(5);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) - 0.5f; v = (unsigned short)(__builtin_bit_cast(unsigned int, f) >> 16); }
    char *xs, *ws, *outs; float* out;
    CK(hipMalloc(&xs, xs_bytes)); CK(hipMalloc(&ws, (size_t)ncg * 78 * 1024));
    for (size_t o = 0; o < xs_bytes; o += h.size() * 2) CK(hipMemcpy(xs + o, h.data(), std::min(h.size() * 2, xs_bytes - o), hipMemcpyHostToDevice));
    CK(hipMemcpy(ws, h.data(), (size_t)ncg * 78 * 1024, hipMemcpyHostToDevice));
    const size_t out_f = (size_t)256 * tiles_per_wg * 64 * 128;
    CK(hipMalloc(&out, out_f * 4)); CK(hipMalloc(&outs, out_f * 6));
    Args a = {xs, xs_bytes, ws, ncg, out, outs, tiles_per_wg};
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), LDS_BYTES, 0, a);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 3) ts.push_back(ms * 1e3f);
    }
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    const double us = ts[ts.size() / 2];
    const double in_mb = (double)std::min(need, (size_t)N * cin * hw * hw * 6 * (cout / 64)) / 1e6;
    printf("%s (256 frames, %d -> %d channels, %dx%d -> %dx%d): %ld tiles of 64 x 128, %d per workgroup, %d stages each\n", layer, cin, cout, hw, hw,
           hw / 2, hw / 2, tiles, tiles_per_wg, ncg);
    printf("  model kernel: median %.1f us (min %.1f)  = %.0f TFLOP/s fp32-equivalent (%.2f GFLOP of the layer)\n", us, ts[0], flop / us / 1e6, flop / 1e9);
    printf("  HBM-side bytes per launch: split input %.0f MB (fp32: %.0f), fp32 output %.0f MB + split output %.0f MB (fp32 alone: %.0f); LDS-DMA per stage and workgroup 111 KB\n",
           (double)N * cin * hw * hw * 6 / 1e6, (double)N * cin * hw * hw * 4 / 1e6, out_f * 4 / 1e6, out_f * 6 / 1e6, out_f * 4 / 1e6);
    (void)in_mb;
    return 0;
}
