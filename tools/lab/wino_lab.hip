// Winograd question, speed half: what would enc.conv2's forward pass (64 -> 128 channels,
// 32x32 -> 16x16, 5x5 stride 2, 256 frames; k_down2_mfma<2, 1>: 200 us in the training step) reach as
// F(2, 3) / F(2, 2) Winograd on its four phase convolutions -- 16 + 12 + 12 + 9 = 49 products per 2x2
// output block instead of 100 (numerics: tools/lab/wino_numerics.py, 3.1e-7 against 2.4e-7 for the direct
// float32 sum)?
//
// NOT a convolution: an instruction-mix model of the kernel one would build, with that design's MFMA count,
// LDS traffic and access patterns, transform arithmetic, LDS-DMA byte counts, barriers and output stores, on
// real (random) data -- it can only be FASTER than the real thing (no border handling, no index arithmetic
// beyond what the access patterns need):
//   * workgroup = 4 waves (one per SIMD, 512 registers each) = 64 output channels x 32 tiles (the 2x2 output
//     blocks of HALF a frame), wave = 32 channels x 16 tiles on v_mfma_f32_16x16x4_f32: the 16 point
//     accumulators of the 3x3 phase are 16 x 2 blocks x 4 = 128 registers.  (Tried first: 32x32x2 tiles, 16
//     accumulators = 256 registers = every accumulation register: 900 spilled registers; then eight waves of
//     256 registers on 64 tiles: 240 spilled.)  Grid = 256 frames x 2 halves x 2 channel blocks;
//   * the four phase groups one after the other (P = 16, 12, 12, 9 points); a group runs over the 64 input
//     channels in 8 chunks of 8 (two k-steps): per chunk and wave P x 4 MFMAs, operands by three conflict-free
//     ds_read_b32 per pair of MFMAs (A = transformed filters U[p][c][k], rows of 80 floats; B = transformed
//     input V[p][c][t], rows of 48) -- NO register blocking is possible: the points take the registers;
//   * per chunk and thread one (tile, channel) input transform: (rt + 1) x (st + 1) raw values from the LDS image
//     of the phase rows (stride-2 column gather), B^T d B by the real add counts, P ds_write_b32 into V;
//   * LDS-DMA per chunk: the raw rows of 8 channels (10 rows x 40 floats each: 12.5 KB) and the U slice
//     (P x 8 x 80 floats <= 40 KB; U is precomputed per step, 1.6 MB per layer, L2-resident), requested one chunk
//     ahead (double buffers: 154 KB of LDS); two barriers per chunk.  The U slice is re-read by every workgroup:
//     49 / 25 of the weights per 32 tiles = 1 GB per launch from the L2s, where the direct kernel's 64 x 256-pixel
//     tiles read 0.2 GB;
//   * per group: the inverse transform A^T m A of the P accumulators into the four output accumulators (real
//     add counts); per tile: bias + LeakyReLU + 8-byte stores.
// Reported: us per launch against k_down2_mfma's 200, with and without the transforms / the DMA, and the floor:
// a pure MFMA stream of the model's product count (1568 per wave on a 512-workgroup grid).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/wino_lab.hip -o tools/lab/bin/wino_lab
//   tools/lab/bin/wino_lab [iters]          (-DWL_NO_XFORM, -DWL_NO_DMA, -DWL_NO_OUTX: ablations)
//   results and the go / no-go: profiles/r06_wino_lab.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define NFRAMES 256
#define CIN 64
#define KOUT 128
#define CC 8                                   // channels per chunk = two k-steps of v_mfma_f32_16x16x4_f32
#define NCHUNK (CIN / CC)
#define ROWU 80                                // floats per (point, channel) row of U: 64 + 16 / of V: 32 + 16 (the
#define ROWV 48                                // four channel groups of an operand read fall on disjoint banks)
#define RAW_BUF 3328                           // 8 channels x 10 rows x 40 floats = 3200 (whole kilobytes: 13 KB)
#define U_BUF (16 * CC * ROWU)                 // 10240 floats = 40 KB
#define V_BUF (16 * CC * ROWV)                 // 6144 floats = 24 KB
#define LDS_FLOATS (2 * RAW_BUF + 2 * U_BUF + 2 * V_BUF)     // 26 + 80 + 48 = 154 KB
#define WG_THREADS 256

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float* lds, int vo, int so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, vo, so, 0, 0);
}

struct Args {
    const float* x;          // input frames, NCHW
    const float* u;          // transformed filters [kblock][group][chunk][p][c][80]
    const float* bias;
    float* out;
    size_t x_bytes, u_bytes;
};

// one phase group: RT x ST sub-kernel, P = (RT + 1) (ST + 1) points.  Four waves, one per SIMD (512 registers):
// wave = 32 channels x 16 tiles = two 16x16 blocks per point
template <int RT, int ST, int GI>
__device__ __forceinline__ void group(const Args& a, float* smem, f32x4 (&o)[4][2], int frame, int half,
                                      __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ru, int& ubase) {
    constexpr int PR = RT + 1, PS = ST + 1, P = PR * PS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, cg = lane >> 4;
    const int kb = wv & 1, tb = wv >> 1;
    float* raw = smem;
    float* ub = smem + 2 * RAW_BUF;
    float* vb = smem + 2 * RAW_BUF + 2 * U_BUF;

    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[P][2];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[p][0][e] = 0.f; acc[p][1][e] = 0.f; }

    // input transform: one (tile, channel) of a chunk per thread: tile = tid & 31 (4 x 8 tiles of the half frame)
    const int tile = tid & 31, tc = tid >> 5;
    const int ty = tile >> 3, tx = tile & 7;
    // raw image of a channel: 10 phase rows x 40 floats (columns 4.. hold image columns 0..31)
    const int rbase = tc * 400 + (2 * ty) * 40 + 4 * tx + 3 + (GI & 1);
    const int vwr = tc * ROWV + tile;                             // V[p][c][t]
    const int ard = cg * ROWU + kb * 32 + l16;                    // U[p][c = 4 step + cg][k] (+ 16: second block)
    const int brd = cg * ROWV + tb * 16 + l16;                    // V[p][c = 4 step + cg][t]

    auto issue = [&](int ch) __attribute__((always_inline)) {
#ifndef WL_NO_DMA
        if (ch < NCHUNK) {
            const int buf = ch & 1;
            // raw rows: 800 groups of 16 B (8 channels x 10 rows x 10 groups) in 832 slots = 13 wave requests
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + WG_THREADS * k;
                if (k < 3 || wv == 0) {                             // wave-uniform
                    const int c = e / 100, r = (e - c * 100) / 10, q = e - c * 100 - r * 10;
                    const int row = 16 * half + 2 * r + (GI >> 1);
                    const int off = e < 800 ? (((ch * CC + c) * 32 + (row & 31)) * 32 + ((4 * q) & 31)) * 4 : 0x7fffffff;
                    dma16(rx, raw + buf * RAW_BUF + 4 * (WG_THREADS * k + 64 * wv), off, frame * CIN * 1024 * 4);
                }
            }
            // U slice: P x 8 x 80 floats = P x 160 groups (<= 2560 = 40 wave requests)
#pragma unroll
            for (int k = 0; k < (P * 160 + WG_THREADS - 1) / WG_THREADS; ++k) {
                const int e = tid + WG_THREADS * k;
                if (WG_THREADS * k + 64 * wv < P * 160)              // wave-uniform
                    dma16(ru, ub + buf * U_BUF + 4 * (WG_THREADS * k + 64 * wv), e * 16, (ubase + ch * U_BUF) * 4);
            }
        }
#endif
    };
    auto transform = [&](int ch) __attribute__((always_inline)) {
#ifndef WL_NO_XFORM
        const float* rp = raw + (ch & 1) * RAW_BUF + rbase;
        float d[PR][PS];
#pragma unroll
        for (int i = 0; i < PR; ++i)
#pragma unroll
            for (int j = 0; j < PS; ++j) d[i][j] = rp[i * 40 + 2 * j];      // stride-2 column gather
        float t[PR][PS];
#pragma unroll
        for (int j = 0; j < PS; ++j) {
            if (PR == 4) {
                t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
            } else {
                t[0][j] = d[0][j] - d[1][j]; t[1][j] = d[1][j]; t[2][j] = d[2][j] - d[1][j];
            }
        }
        float* vp = vb + (ch & 1) * V_BUF + vwr;
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            float v[PS];
            if (PS == 4) {
                v[0] = t[i][0] - t[i][2]; v[1] = t[i][1] + t[i][2];
                v[2] = t[i][2] - t[i][1]; v[3] = t[i][1] - t[i][3];
            } else {
                v[0] = t[i][0] - t[i][1]; v[1] = t[i][1]; v[2] = t[i][2] - t[i][1];
            }
#pragma unroll
            for (int j = 0; j < PS; ++j) vp[(i * PS + j) * (CC * ROWV)] = v[j];
        }
#endif
    };
    auto products = [&](int ch) __attribute__((always_inline)) {
        const float* up = ub + (ch & 1) * U_BUF + ard;
        const float* vp = vb + (ch & 1) * V_BUF + brd;
#pragma unroll
        for (int s = 0; s < CC / 4; ++s)
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float a0 = up[p * (CC * ROWU) + s * 4 * ROWU], a1 = up[p * (CC * ROWU) + s * 4 * ROWU + 16];
            const float bv = vp[p * (CC * ROWV) + s * 4 * ROWV];
            acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[p][0], 0, 0, 0);
            acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[p][1], 0, 0, 0);
            if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // operand reads at most four points ahead
        }
    };

    // chunk ch: its products run while chunk ch + 1 is requested (DMA) -- V of chunk ch + 1 is transformed behind the
    // products, once its raw rows have landed
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform(0);
    issue(1);
    __syncthreads();
#pragma nounroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
        products(ch);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                            // raw / U of chunk ch + 1 are in LDS, V[ch & 1] and
        if (ch + 1 < NCHUNK) transform(ch + 1);                     // raw[ch & 1], U[ch & 1] are free
        issue(ch + 2);
        __syncthreads();
    }
    ubase += NCHUNK * U_BUF;
    __builtin_amdgcn_sched_barrier(0);

    // inverse transform A^T m A of this lane's (channel, tile) elements: P accumulators -> 2 x 2 outputs
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifndef WL_NO_OUTX
        float r0[PS], r1[PS];
#pragma unroll
        for (int j = 0; j < PS; ++j) {
            if (PR == 4) {
                r0[j] = acc[0 * PS + j][blk][e] + acc[1 * PS + j][blk][e] + acc[2 * PS + j][blk][e];
                r1[j] = acc[1 * PS + j][blk][e] - acc[2 * PS + j][blk][e] - acc[3 * PS + j][blk][e];
            } else {
                r0[j] = acc[0 * PS + j][blk][e] + acc[1 * PS + j][blk][e];
                r1[j] = acc[1 * PS + j][blk][e] + acc[2 * PS + j][blk][e];
            }
        }
        if (PS == 4) {
            o[0][blk][e] += r0[0] + r0[1] + r0[2]; o[1][blk][e] += r0[1] - r0[2] - r0[3];
            o[2][blk][e] += r1[0] + r1[1] + r1[2]; o[3][blk][e] += r1[1] - r1[2] - r1[3];
        } else {
            o[0][blk][e] += r0[0] + r0[1]; o[1][blk][e] += r0[1] + r0[2];
            o[2][blk][e] += r1[0] + r1[1]; o[3][blk][e] += r1[1] + r1[2];
        }
#else
#pragma unroll
        for (int p = 0; p < P; ++p) o[p & 3][blk][e] += acc[p][blk][e];
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_wino_model(Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int frame = blockIdx.x >> 2, kblk = blockIdx.x & 1, half = (blockIdx.x >> 1) & 1;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, (int)a.u_bytes, 0x00020000);
    f32x4 o[4][2];
#pragma unroll
    for (int z = 0; z < 4; ++z)
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[z][0][e] = 0.f; o[z][1][e] = 0.f; }
    int ubase = kblk * 4 * NCHUNK * U_BUF;
    group<3, 3, 0>(a, smem, o, frame, half, rx, ru, ubase);
    group<3, 2, 1>(a, smem, o, frame, half, rx, ru, ubase);
    group<2, 3, 2>(a, smem, o, frame, half, rx, ru, ubase);
    group<2, 2, 3>(a, smem, o, frame, half, rx, ru, ubase);
    // epilogue: lane holds channels 4 (lane >> 4) + e of block blk for tile lane & 15: bias, LeakyReLU, the
    // tile's two rows as 8-byte stores
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, cg = lane >> 4, kb = wv & 1, tb = wv >> 1;
    const int tile = tb * 16 + l16, ty = 4 * half + (tile >> 3), tx = tile & 7;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = kblk * 64 + kb * 32 + blk * 16 + 4 * cg + e;
        const float b = a.bias[k];
        float v0 = o[0][blk][e] + b, v1 = o[1][blk][e] + b, v2 = o[2][blk][e] + b, v3 = o[3][blk][e] + b;
        v0 = v0 > 0.f ? v0 : 0.05f * v0; v1 = v1 > 0.f ? v1 : 0.05f * v1;
        v2 = v2 > 0.f ? v2 : 0.05f * v2; v3 = v3 > 0.f ? v3 : 0.05f * v3;
        float* op = a.out + (((size_t)frame * KOUT + k) * 16 + 2 * ty) * 16 + 2 * tx;
        *reinterpret_cast<float2*>(op) = make_float2(v0, v1);
        *reinterpret_cast<float2*>(op + 16) = make_float2(v2, v3);
    }
}

// calibration: the direct kernel's MFMA count (100 products per 2x2 block: 3200 per wave for the same tile) as a
// pure stream with operands from LDS = what "MFMA-bound" means under this lab's clocks
__global__ __launch_bounds__(256, 1) void k_cal(const float* src, float* dst, int per_wave) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x16 acc[8];
    for (int z = 0; z < 8; ++z) for (int t = 0; t < 16; ++t) acc[z][t] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < per_wave / 8; ++r) {
        const float av = lds[(lane + 64 * r) & 4095], bv = lds[(lane * 3 + 64 * r + 1) & 4095];
#pragma unroll
        for (int z = 0; z < 8; ++z) acc[z] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[z], 0, 0, 0);
    }
    float s = 0.f;
    for (int z = 0; z < 8; ++z) for (int t = 0; t < 16; ++t) s += acc[z][t];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    srand(5);
    const size_t nx = (size_t)NFRAMES * CIN * 32 * 32, nu = (size_t)2 * 4 * NCHUNK * U_BUF, no = (size_t)NFRAMES * KOUT * 256;
    std::vector<float> hx(nx), hu(nu), hb(KOUT);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.3f;
    for (auto& v : hu) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (auto& v : hb) v = rand() / (float)RAND_MAX - 0.5f;
    // inputs / outputs rotated through > 256 MB, as the training step does (the Infinity Cache flatters a lab
    // that re-uses its buffers)
    const int ROT = 4;
    float *dx[ROT], *dout[ROT], *du, *db, *dcal;
    for (int r = 0; r < ROT; ++r) {
        CK(hipMalloc(&dx[r], nx * 4)); CK(hipMemcpy(dx[r], hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dout[r], no * 4));
    }
    CK(hipMalloc(&du, nu * 4)); CK(hipMemcpy(du, hu.data(), nu * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, KOUT * 4)); CK(hipMemcpy(db, hb.data(), KOUT * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dcal, 512 * 256 * 4));
    CK(hipFuncSetAttribute((const void*)k_wino_model, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int r) {
        Args a{dx[r % ROT], du, db, dout[r % ROT], nx * 4, nu * 4};
        hipLaunchKernelGGL(k_wino_model, dim3(NFRAMES * 4), dim3(WG_THREADS), LDS_FLOATS * 4, 0, a);
    };
    for (int i = 0; i < 5; ++i) run(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) run(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    std::vector<float> ho(4096);
    CK(hipMemcpy(ho.data(), dout[0], 4096 * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : ho) cs += v;
    // calibration: the same grid with the direct kernel's MFMA count (3200 per wave) and the model's (1568)
    float ms_c[2];
    const int counts[2] = {3200, 1568};
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_cal, dim3(512), dim3(256), 0, 0, dx[0], dcal, counts[k]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_cal, dim3(512), dim3(256), 0, 0, dx[0], dcal, counts[k]);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms_c[k], e0, e1));
    }
    const char* variant =
#if defined(WL_NO_XFORM)
        "no input transform";
#elif defined(WL_NO_DMA)
        "no LDS-DMA";
#elif defined(WL_NO_OUTX)
        "no inverse transform";
#else
        "full model";
#endif
    printf("wino model (%s): %.1f us per launch (enc.conv2 forward, 256 frames; k_down2_mfma<2, 1>: 200 us in the step)"
           "  checksum %.6g\n", variant, us, cs);
    printf("  pure MFMA streams on the same grid: 3200 per wave (direct) %.1f us, 1568 per wave (49 points) %.1f us\n",
           ms_c[0] * 1e3 / iters, ms_c[1] * 1e3 / iters);
    printf("  layer FLOPs (direct count, 26.84 GFLOP) / time = %.1f TFLOP/s-equivalent\n", 26.84e9 / (us * 1e-6) / 1e12);
    return 0;
}
