// HBM-bound edge layers of the autoencoder (single-channel image side), kernel 5x5 stride 2:
//
//   enc.conv0  fwd        1 -> 32   read 64 KB, write 512 KB per frame   (k_down_c1)
//   dec.convT4 bwd-data   same kernel with the LeakyReLU' mask epilogue
//   dec.convT4 fwd        32 -> 1   read 512 KB, write 64 KB per frame   (k_up_c1)
//   enc.conv0 / dec.convT4 weight gradients (32 x 25 outputs, 580 KB read per frame)
//                                                                       (k_wgrad_c1)
//
// All three are bound by the 8 TB/s HBM stream, not by arithmetic (11 FLOP/B); the matrix cores
// are used as the FMA engine (v_mfma_f32_32x32x2_f32, exact fp32) only because that keeps the
// VALU and LDS out of the way of the stream.  Every global access is a full 128/256-byte
// wavefront row; padding, halos and tails come from raw-buffer out-of-range zeros.
#include "bn_common.h"
#include "bn_fast.h"
#include "bn_reduce.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

#define ED_THREADS 256
#define ED_OOB 0x7fffffff

__device__ __forceinline__ float ed_ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// =============================================================================================
// weight gradient with one big-side channel:
//   dW[a][tap] = sum_{n,p,q} small[n,a,p,q] * big[n,0,2p+r-pt,2q+s-pl],   a < 32
// MFMA roles: rows = a (32), columns = taps (25 of 32), reduction = pixels (2 per instruction).
// Stage = 4 rows x 64 columns of the small image (256 pixels, one row per wave).
// =============================================================================================
#define WC_ROWS 4
#define WC_W 64
#define WC_TPX (WC_ROWS * WC_W)
#define WC_SP (WC_TPX + 2)              // small-tile row stride (== 2 mod 32)
#define WC_IH (2 * WC_ROWS + 3)
#define WC_RW (2 * WC_W + 4)            // big-tile row stride (131 used)
#define WC_KS (32 * WC_TPX / ED_THREADS)            // 32 small elements per thread per stage
#define WC_KB ((WC_IH * WC_RW + ED_THREADS - 1) / ED_THREADS)

__global__ __launch_bounds__(ED_THREADS) void k_wgrad_c1(
    const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ part,
    BnGeom g, int n_stages, int stages_per_frame) {
    __shared__ float sl[32 * WC_SP];
    __shared__ float bl[WC_IH * WC_RW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kk = lane >> 5;
    const int PQ = g.Hs * g.Ws, HWb = g.Hb * g.Wb;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)small, 0, (int)((size_t)g.N * g.Cs * PQ * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)big, 0, (int)((size_t)g.N * HWb * 4), 0x00020000);

    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    float sr[WC_KS];
    float br[WC_KB];

    auto issue_loads = [&](int st) {
        const int n = st / stages_per_frame;
        const int p0 = (st - n * stages_per_frame) * WC_ROWS;
#pragma unroll
        for (int k = 0; k < WC_KS; ++k) {
            const int e = tid + ED_THREADS * k;
            const int a = e >> 8, pix = e & (WC_TPX - 1);
            const bool ok = a < g.Cs && (p0 + (pix >> 6)) < g.Hs;
            sr[k] = ed_ld(rs, ok ? (((n * g.Cs + a) * g.Hs + p0) * g.Ws + pix) * 4 : ED_OOB);
        }
#pragma unroll
        for (int k = 0; k < WC_KB; ++k) {
            const int e = tid + ED_THREADS * k;
            const int y = e / WC_RW, x = e - y * WC_RW;
            const int hb = 2 * p0 - g.pt + y, wb = x - g.pl;
            const bool ok = y < WC_IH && hb >= 0 && hb < g.Hb && wb >= 0 && wb < g.Wb;
            br[k] = ed_ld(rb, ok ? ((n * g.Hb + hb) * g.Wb + wb) * 4 : ED_OOB);
        }
    };

    // lane-constant part of the B gather: tap j = li -> (r, s); pixel (row wv, column 2t+kk)
    const int tap = li < 25 ? li : 0;
    const int tr = tap / 5, ts = tap - tr * 5;
    const float* bq = bl + (2 * wv + tr) * WC_RW + ts + 2 * kk;
    const float* aq = sl + li * WC_SP + wv * WC_W + kk;

    int st = blockIdx.x;
    if (st < n_stages) issue_loads(st);
    for (; st < n_stages; st += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WC_KS; ++k) {
            const int e = tid + ED_THREADS * k;
            sl[(e >> 8) * WC_SP + (e & (WC_TPX - 1))] = sr[k];
        }
#pragma unroll
        for (int k = 0; k < WC_KB; ++k) {
            const int e = tid + ED_THREADS * k;
            if (e < WC_IH * WC_RW) bl[e] = br[k];
        }
        __syncthreads();
        if (st + (int)gridDim.x < n_stages) issue_loads(st + gridDim.x);
#pragma unroll 8
        for (int t = 0; t < WC_W / 2; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[2 * t], bq[4 * t], acc, 0, 0, 0);
    }

    // combine the four waves (fixed order) and emit this workgroup's partial [a][tap]
    __syncthreads();
    float* red = sl;    // 4 x 16 x 64 floats
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wv * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (wv == 0 && li < 25) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = (red[e * 64 + lane] + red[(16 + e) * 64 + lane]) +
                            (red[(32 + e) * 64 + lane] + red[(48 + e) * 64 + lane]);
            const int a = (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (a < g.Cs) part[(size_t)blockIdx.x * (g.Cs * 25) + a * 25 + li] = v;
        }
    }
}

static int wgrad_c1_grid(const BnGeom& g) {
    const int n_stages = g.N * (g.Hs / WC_ROWS);
    return n_stages < 768 ? n_stages : 768;     // 3 resident workgroups per CU
}

BnFastPlan bn_edge_wgrad_plan(const BnGeom& g) {
    BnFastPlan p = {false, "k_wgrad_generic", 0, 0, 0, 0, 0, 0};
    if (g.R != 5 || g.S != 5 || g.stride != 2 || g.Cb != 1) return p;
    if (g.Cs > 32 || g.Ws != WC_W || (g.Hs % WC_ROWS) != 0) return p;
    if (g.Hb != 2 * g.Hs || g.Wb != 2 * g.Ws) return p;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return p;
    p.supported = true;
    p.d = wgrad_c1_grid(g);
    p.ws_bytes = (size_t)p.d * g.Cs * 25 * sizeof(float);
    p.kernel_name = "k_wgrad_c1";
    return p;
}

int bn_launch_edge_wgrad(const BnFastPlan& plan, const float* small, const float* big, float* dw,
                         const BnGeom& g, int accumulate, void* ws, hipStream_t st) {
    const int n_stages = g.N * (g.Hs / WC_ROWS);
    hipLaunchKernelGGL(k_wgrad_c1, dim3(plan.d), dim3(ED_THREADS), 0, st, small, big, (float*)ws, g,
                       n_stages, g.Hs / WC_ROWS);
    BN_LAUNCH_CHECK();
    return bn_launch_sum_partials((const float*)ws, dw, g.Cs * 25, plan.d, accumulate, 0, 0, st);
}
