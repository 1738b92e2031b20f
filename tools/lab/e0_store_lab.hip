// Store-pattern laboratory for enc.conv0 (1 -> 32 channels, 128x128 -> 64x64, 256 frames):
// what does the SHAPE of the 134 MB output stream cost, everything else being equal?
//
// Every variant writes the same bytes (256 x 32 x 64 x 64 floats, write-through `sc1` 16-byte buffer
// stores like the product kernel) from the same persistent one-wave workgroups, and reads its
// input patch (16.8 MB in total + halo rows) before a unit's stores; only the mapping of a store
// instruction's 64 lanes to addresses differs:
//   mode 0  plain fill: unit = 16 KB contiguous, 1 KB per instruction (the ceiling)
//   mode 1  product pattern (k_down_c1, DC_HALF): unit = 2 rows x 32 ch; per half row 4 instructions of
//           8 channels x 128 B (pieces 16 KB apart)
//   mode 2  full rows: unit = 2 rows; per row 8 instructions of 4 channels x 256 B
//   mode 3  unit = 2 rows; 16 instructions of 2 channels x (2 rows = 512 B contiguous)
//   mode 4  unit = 4 rows; 32 instructions of 1 channel x (4 rows = 1 KB contiguous)
//   mode 5  unit = 8 rows; 64 instructions, two back to back per channel (2 KB contiguous)
//   mode 6  unit = 4 rows, dword stores as the second-generation kernel: 2 x 128 B per instruction
// Before every timed launch a "dirtying" kernel streams plain stores through the L2s (E0_DIRTY_MB, default
// 35: the optimizer's zero_grad fill), and inputs / outputs rotate through rings larger than the 256 MB
// Infinity Cache (E0_RING=1 switches that off: the first version of this lab measured 6.5 TB/s for
// every pattern because the one 134 MB output buffer lived in that cache).
// usage: e0_store_lab [grid_waves_per_cu=16] [iters=20]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));

#define NF 256
#define CH 32
#define HS 64
#define WS 64
#define PQ (HS * WS)

template <int MODE, int READ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_store(
    const float* __restrict__ in, float* __restrict__ out, int units) {
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffffc, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, NF * 128 * 128 * 4, 0x00020000);
    constexpr int ROWS = MODE == 0 ? 2 : MODE == 4 || MODE == 6 ? 4 : MODE == 5 ? 8 : 2;
    constexpr int UPF = HS / ROWS;
    constexpr int IH = 2 * ROWS + 3;
    constexpr int NLD = (IH * 32 + 63) / 64;
    float v = (float)lane;
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int n = u / UPF, p0 = ROWS * (u - n * UPF);
        if (READ) {
            // the unit's input patch: IH rows of 512 B, 16 B per lane
            intx4 st[NLD];
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int e = lane + 64 * k;
                const int y = e >> 5, c = e & 31;
                const int hb = 2 * p0 - 1 + y;
                const bool ok = y < IH && hb >= 0 && hb < 128;
                st[k] = __builtin_amdgcn_raw_buffer_load_b128(ri, ok ? ((n * 128 + hb) * 128 + 4 * c) * 4 : 0x7fffffff, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) v += __builtin_bit_cast(float, st[k].x);
        }
        const uintx4 d = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
        if (MODE == 0) {
            // 16 KB contiguous per unit
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(d, ro, ((u * 16 + i) * 64 + lane) * 16, 0, 16);
        } else if (MODE == 1) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ch = 8 * i + (lane >> 3);
                        const int o = ((n * CH + ch) * HS + p0 + pr) * WS + 32 * qh + 4 * (lane & 7);
                        __builtin_amdgcn_raw_buffer_store_b128(d, ro, o * 4, 0, 16);
                    }
        } else if (MODE == 2) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = 4 * i + (lane >> 4);
                    const int o = ((n * CH + ch) * HS + p0 + pr) * WS + 4 * (lane & 15);
                    __builtin_amdgcn_raw_buffer_store_b128(d, ro, o * 4, 0, 16);
                }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ch = 2 * i + (lane >> 5);
                const int o = ((n * CH + ch) * HS + p0) * WS + 4 * (lane & 31);
                __builtin_amdgcn_raw_buffer_store_b128(d, ro, o * 4, 0, 16);
            }
        } else if (MODE == 4) {
#pragma unroll
            for (int ch = 0; ch < 32; ++ch) {
                const int o = ((n * CH + ch) * HS + p0) * WS + 4 * lane;
                __builtin_amdgcn_raw_buffer_store_b128(d, ro, o * 4, 0, 16);
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int ch = 0; ch < 32; ++ch)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int o = ((n * CH + ch) * HS + p0 + 4 * h) * WS + 4 * lane;
                    __builtin_amdgcn_raw_buffer_store_b128(d, ro, o * 4, 0, 16);
                }
        } else if (MODE == 6) {
#pragma unroll
            for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ch = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int o = ((n * CH + ch) * HS + p0 + pr) * WS + 32 * qh + (lane & 31);
                        __builtin_amdgcn_raw_buffer_store_b32(d.x, ro, o * 4, 0, 0);
                    }
        }
    }
}

__global__ __launch_bounds__(256) void k_dirty(float4* p, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <int MODE, int READ>
static void run(const char* name, const float* in_ring, float* out_ring, float4* dirty, size_t dirty_n4, int wpc, int iters,
                hipStream_t st) {
    constexpr int ROWS = MODE == 0 ? 2 : MODE == 4 || MODE == 6 ? 4 : MODE == 5 ? 8 : 2;
    const int units = NF * (HS / ROWS);
    int grid = 256 * wpc;
    if (grid > units) grid = units;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < iters + 3; ++i) {
        // rotate inputs (20 x 16.8 MB) and outputs (3 x 134 MB): the 256 MB Infinity Cache must not
        // absorb the streams (E0_RING=1: one buffer each, the cache-flattered figure)
        static const int ring = getenv("E0_RING") ? atoi(getenv("E0_RING")) : 20;
        const float* in = in_ring + (size_t)(i % ring) * NF * 128 * 128;
        float* out = out_ring + (size_t)(ring > 1 ? i % 3 : 0) * NF * CH * PQ;
        if (dirty_n4) hipLaunchKernelGGL(k_dirty, dim3(2048), dim3(256), 0, st, dirty, dirty_n4);
        hipExtLaunchKernelGGL((k_store<MODE, READ>), dim3(grid), dim3(64), 0, st, e0, e1, 0, in, out, units);
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)NF * CH * PQ * 4 + (READ ? (double)NF * 128 * 128 * 4 : 0.0);
    printf("  %-44s read %d grid %5d: median %6.2f us (min %6.2f)  %.2f TB/s algorithmic\n", name, READ, grid,
           ts[ts.size() / 2], ts[0], bytes / ts[ts.size() / 2] / 1e6);
}

int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 16;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    float *in, *out; float4* dirty;
    const size_t n_out = (size_t)NF * CH * PQ, n_in = (size_t)NF * 128 * 128;
    // E0_DIRTY_MB: plain stores streamed through the L2s before every timed launch (default 35: the
    // optimizer's zero_grad fill in the training step; 0: none)
    const size_t dirty_mb = getenv("E0_DIRTY_MB") ? atoi(getenv("E0_DIRTY_MB")) : 35;
    const size_t dirty_n4 = dirty_mb * 1000 * 1000 / 16;
    CK(hipMalloc(&in, n_in * 4 * 20)); CK(hipMalloc(&out, n_out * 4 * 3)); CK(hipMalloc(&dirty, 160u * 1000 * 1000));
    CK(hipMemset(in, 0, n_in * 4 * 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("enc.conv0 output stream, %d waves per CU\n", wpc);
    run<0, 0>("0 fill, 1 KB/instr, 16 KB/unit", in, out, dirty, dirty_n4, wpc, iters, st);
    run<0, 1>("0 fill, 1 KB/instr, 16 KB/unit", in, out, dirty, dirty_n4, wpc, iters, st);
    run<1, 0>("1 product: 8 ch x 128 B /instr", in, out, dirty, dirty_n4, wpc, iters, st);
    run<1, 1>("1 product: 8 ch x 128 B /instr", in, out, dirty, dirty_n4, wpc, iters, st);
    run<2, 1>("2 4 ch x 256 B /instr", in, out, dirty, dirty_n4, wpc, iters, st);
    run<3, 1>("3 2 ch x 512 B /instr (2 rows)", in, out, dirty, dirty_n4, wpc, iters, st);
    run<4, 0>("4 1 ch x 1 KB /instr (4-row units)", in, out, dirty, dirty_n4, wpc, iters, st);
    run<4, 1>("4 1 ch x 1 KB /instr (4-row units)", in, out, dirty, dirty_n4, wpc, iters, st);
    run<5, 1>("5 1 ch x 2 KB / 2 instr (8-row units)", in, out, dirty, dirty_n4, wpc, iters, st);
    run<6, 1>("6 dword stores 2 x 128 B /instr (4-row units)", in, out, dirty, dirty_n4, wpc, iters, st);
    return 0;
}
