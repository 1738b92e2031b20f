// extern "C" surface of libbehavenet_hip.so (include/behavenet_hip.h): argument checks, the
// mapping of the six convolution roles onto the three kernel families, fast-path dispatch and
// the hipEvent profiling hook.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_ext.h>
#include "bn_common.h"
#include "bn_launch.h"
#include "bn_fast.h"

// ------------------------------------------------------------------------------------------
// profiling hook
// ------------------------------------------------------------------------------------------
namespace {
// One slot per profiled op: a BRACKET pair (hipEventRecord before / after everything the op
// launches: the whole op, event overhead and launch gaps included) and a DISPATCH pair that the
// launcher of the op's main kernel attaches to that one dispatch (hipExtLaunchKernelGGL start /
// stop events = the kernel's own begin / end timestamps, what rocprofv3 reports).
struct ProfState {
    int family = BN_PROF_NONE;
    int C = 0, K = 0;
    static const int kMaxSlots = 2048;
    hipEvent_t ev[4 * kMaxSlots];
    bool taken[kMaxSlots], bracketed[kMaxSlots];
    int created = 0;       // slots whose four events exist
    int used = 0;          // slots recorded since the last read/reset
    double total_ms = 0.0, main_ms = 0.0;
    long launches = 0, main_launches = 0;
    bool bracket = true;   // bn_prof_set_bracket(0): dispatch pair only (nothing extra on the stream)
    int nth = -1, seen = 0;   // bn_prof_select_nth: only the nth matching call since the selection
    char kernel_name[96] = "";
} g_prof;

void prof_drain() {
    for (int i = 0; i < g_prof.used; ++i) {
        float ms = 0.f;
        if (g_prof.bracketed[i] && hipEventSynchronize(g_prof.ev[4 * i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_prof.ev[4 * i], g_prof.ev[4 * i + 1]) == hipSuccess) {
            g_prof.total_ms += ms;
            g_prof.launches += 1;
        }
        if (g_prof.taken[i] && hipEventSynchronize(g_prof.ev[4 * i + 3]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_prof.ev[4 * i + 2], g_prof.ev[4 * i + 3]) == hipSuccess) {
            g_prof.main_ms += ms;
            g_prof.main_launches += 1;
        }
    }
    g_prof.used = 0;
}
}  // namespace

static int g_dispatch_slot = -1;            // slot whose dispatch pair waits for a launcher

bool bn_prof_take_dispatch_events(hipEvent_t* e0, hipEvent_t* e1) {
    if (g_dispatch_slot < 0) return false;
    *e0 = g_prof.ev[4 * g_dispatch_slot + 2];
    *e1 = g_prof.ev[4 * g_dispatch_slot + 3];
    g_prof.taken[g_dispatch_slot] = true;
    g_dispatch_slot = -1;
    return true;
}

BnProfScope::BnProfScope(int family, int C, int K, const char* kernel_name, hipStream_t s)
    : active(false), stream(s), slot(-1) {
    if (g_prof.family == BN_PROF_NONE || g_prof.family != family) return;
    if (g_prof.C > 0 && g_prof.C != C) return;
    if (g_prof.K > 0 && g_prof.K != K) return;
    if (g_dispatch_slot >= 0) return;                 // nested scope (a detour re-entering run_*)
    if (g_prof.nth >= 0 && g_prof.seen++ != g_prof.nth) return;
    if (g_prof.used >= ProfState::kMaxSlots) prof_drain();
    const int i = g_prof.used;
    while (g_prof.created <= i) {
        for (int e = 0; e < 4; ++e)
            if (hipEventCreate(&g_prof.ev[4 * g_prof.created + e]) != hipSuccess) return;
        g_prof.created++;
    }
    if (kernel_name) {
        strncpy(g_prof.kernel_name, kernel_name, sizeof(g_prof.kernel_name) - 1);
        g_prof.kernel_name[sizeof(g_prof.kernel_name) - 1] = 0;
    }
    g_prof.taken[i] = false;
    g_prof.bracketed[i] = g_prof.bracket;
    if (g_prof.bracket && hipEventRecord(g_prof.ev[4 * i], stream) != hipSuccess) return;
    slot = i;
    g_dispatch_slot = i;
    active = true;
}

BnProfScope::~BnProfScope() {
    if (!active) return;
    g_dispatch_slot = -1;
    if (!g_prof.bracketed[slot]) { if (g_prof.taken[slot]) g_prof.used++; return; }
    if (hipEventRecord(g_prof.ev[4 * slot + 1], stream) == hipSuccess) g_prof.used++;
}

extern "C" int bn_prof_select(int family, int C, int K) {
    if (family < BN_PROF_NONE || family > BN_PROF_LINEAR_BWD) return BN_E_BADARG;
    g_prof.family = family;
    g_prof.C = C;
    g_prof.K = K;
    g_prof.used = 0;
    g_prof.total_ms = g_prof.main_ms = 0.0;
    g_prof.launches = g_prof.main_launches = 0;
    g_prof.kernel_name[0] = 0;
    g_prof.nth = -1;
    g_prof.seen = 0;
    g_dispatch_slot = -1;
    return 0;
}

extern "C" int bn_prof_select_nth(int family, int C, int K, int nth) {
    const int rc = bn_prof_select(family, C, K);
    if (rc == 0) g_prof.nth = nth;
    return rc;
}

extern "C" int bn_prof_read(double* total_ms, long* launches) {
    if (!total_ms || !launches) return BN_E_BADARG;
    prof_drain();
    *total_ms = g_prof.total_ms;
    *launches = g_prof.launches;
    return 0;
}

extern "C" int bn_prof_set_bracket(int on) {
    const int prev = g_prof.bracket ? 1 : 0;
    g_prof.bracket = on != 0;
    return prev;
}

extern "C" int bn_prof_read_main(double* total_ms, long* launches) {
    if (!total_ms || !launches) return BN_E_BADARG;
    prof_drain();
    *total_ms = g_prof.main_ms;
    *launches = g_prof.main_launches;
    return 0;
}

extern "C" const char* bn_prof_kernel_name(void) { return g_prof.kernel_name; }

// Constant part of a dispatch-attached event interval: the same start/stop events around an EMPTY
// kernel (minimum over `iters` launches, microseconds).  bench.py reports it next to the raw
// enc.conv0 interval; rocprofv3's kernel timestamps do not contain it.
__global__ void k_prof_empty() {}
extern "C" double bn_prof_dispatch_overhead_us(int iters, bn_stream_t stream) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    double best = 1e30;
    for (int i = 0; i < iters; ++i) {
        hipExtLaunchKernelGGL(k_prof_empty, dim3(1), dim3(64), 0, (hipStream_t)stream, e0, e1, 0);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { best = -1.0; break; }
        if (ms * 1e3 < best) best = ms * 1e3;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

// ------------------------------------------------------------------------------------------
// info
// ------------------------------------------------------------------------------------------
extern "C" int bn_version(void) { return 1; }
extern "C" const char* bn_build_arch(void) { return "gfx950"; }
extern "C" const char* bn_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case BN_E_BADARG: return "BN_E_BADARG: null pointer or non-positive size";
        case BN_E_SHAPE: return "BN_E_SHAPE: geometry not supported";
        case BN_E_WORKSPACE: return "BN_E_WORKSPACE: workspace too small";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

// ------------------------------------------------------------------------------------------
// convolution roles -> kernel families
// ------------------------------------------------------------------------------------------
static inline BnGeom conv_geom(int N, int C, int H, int W, int K, int R, int S, int stride,
                               int pad_t, int pad_l, int P, int Q) {
    BnGeom g;
    g.N = N; g.Cs = K; g.Hs = P; g.Ws = Q; g.Cb = C; g.Hb = H; g.Wb = W;
    g.R = R; g.S = S; g.stride = stride; g.pt = pad_t; g.pl = pad_l; g.CsS = 0; g.KV = 0; g.K0 = 0;
    return g;
}
static inline BnGeom convT_geom(int N, int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                                int crop_t, int crop_l, int Ho, int Wo) {
    BnGeom g;
    g.N = N; g.Cs = Ci; g.Hs = Hi; g.Ws = Wi; g.Cb = Co; g.Hb = Ho; g.Wb = Wo;
    g.R = R; g.S = S; g.stride = stride; g.pt = crop_t; g.pl = crop_l; g.CsS = 0; g.KV = 0; g.K0 = 0;
    return g;
}

// BN_FORCE_GENERIC=1 in the environment disables every fast path (used by tests to cross-check
// the specialised kernels against the shape-agnostic ones on the device).
static int g_force_generic = -1;
static bool force_generic() {
    if (g_force_generic < 0) {
        const char* e = bn_tune_env("BN_FORCE_GENERIC");
        g_force_generic = (e && e[0] == '1') ? 1 : 0;
    }
    return g_force_generic == 1;
}
extern "C" int bn_set_force_generic(int on) {
    const int prev = force_generic() ? 1 : 0;
    g_force_generic = on ? 1 : 0;
    return prev;
}

// the fast kernels move 16-byte groups (LDS-DMA, float4 / float2 accesses): tensors that are not
// 16-byte aligned (a view at an odd offset) take the shape-agnostic kernels
static inline bool aligned16_all(const void* a, const void* b, const void* c, const void* d = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15u) == 0;
}

static inline bool ws_ok(const BnFastPlan& p, void* ws, size_t ws_bytes) {
    return p.ws_bytes == 0 || (ws != nullptr && ws_bytes >= p.ws_bytes);
}

// ---- zero-padded detour for 5x5 stride-2 layers whose small map is no power of two (conv_pad.hip)
struct PadPlan {
    bool ok;
    BnGeom gp;             // the padded geometry the fast kernel runs on
    BnFastPlan inner;
    size_t big_bytes, small_bytes, inner_ws;
    bool edge;             // inner = one of the single-channel edge kernels (own launchers)
    int oh, ow;            // where the big tensor sits inside its padded copy
    char name[96];
};
static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// role: 0 gather-down, 1 gather-up, 2 weight gradient
static PadPlan pad_plan(int role, const BnGeom& g) {
    PadPlan p;
    p.ok = false;
    p.edge = false;
    if (force_generic() || g.R != 5 || g.S != 5 || g.stride != 2) return p;
    // first tap 1 or 2 pixels outside the frame (TF-"same" padding of 3 or 4 in total); the big
    // tensor sits (pt - 1, pl - 1) inside its padded copy, which is then a pt = pl = 1 layer
    if (g.pt < 1 || g.pt > 2 || g.pl < 1 || g.pl > 2) return p;
    const int oh = g.pt - 1, ow = g.pl - 1;
    p.oh = oh; p.ow = ow;
    if (g.Cb <= 2) {
        // single- / two-channel frames (enc.conv0, dec.convT4): the edge kernels want 64-column
        // small maps and row counts in multiples of 16
        if (g.Ws > 64 || g.Cs > 32) return p;
        const int hs = (g.Hs + 15) / 16 * 16;
        if (hs == g.Hs && g.Ws == 64 && g.Hb == 2 * g.Hs && g.Wb == 128 && !oh && !ow) return p;
        if (2 * hs < g.Hb + oh || 128 < g.Wb + ow) return p;
        p.gp = g;
        p.gp.Hs = hs; p.gp.Ws = 64; p.gp.Hb = 2 * hs; p.gp.Wb = 128; p.gp.pt = p.gp.pl = 1;
        p.inner = role == 0 ? bn_edge_down_plan(p.gp) : role == 1 ? bn_edge_up_plan(p.gp)
                                                                  : bn_edge_wgrad_plan(p.gp);
        if (!p.inner.supported) return p;
        p.edge = true;
    } else {
        // next power of two; maps the specialised kernels do not take at that size (4x4) once or
        // twice more (8x8 small maps are the smallest every family covers)
        int hs = next_pow2(g.Hs), wsm = next_pow2(g.Ws);
        if (hs == g.Hs && wsm == g.Ws && g.Hb == 2 * g.Hs && g.Wb == 2 * g.Ws && !oh && !ow) { hs *= 2; wsm *= 2; }
        bool found = false;
        // round 4: the fast families take any height and many widths that are no powers of two -- the
        // narrowest zero-padded copy one of them serves, rows only as far as the shifted big map needs
        {
            const int hmin = 2 * g.Hs >= g.Hb + oh ? g.Hs : (g.Hb + oh + 1) / 2;
            int w0 = 2 * g.Ws >= g.Wb + ow ? g.Ws : (g.Wb + ow + 1) / 2;
            for (int wq = w0; wq <= (wsm > 8 ? wsm : 8) && wq <= w0 + 8 && !found; ++wq) {
                if (wq == g.Ws && hmin == g.Hs && g.Hb == 2 * g.Hs && g.Wb == 2 * g.Ws && !oh && !ow) continue;
                // (gather-up: the small map is the operand that is copied -- 16-byte rows for k_pad2d, or
                // no copy at all when only the output is larger)
                if (role >= 1 && (wq & 3) && !(wq == g.Ws && hmin == g.Hs)) continue;
                p.gp = g;
                p.gp.Hs = hmin; p.gp.Ws = wq; p.gp.Hb = 2 * hmin; p.gp.Wb = 2 * wq; p.gp.pt = p.gp.pl = 1;
                if ((size_t)g.N * g.Cb * p.gp.Hb * p.gp.Wb * 4 >= 0x7fffffffull) break;
                if ((size_t)g.N * g.Cs * p.gp.Hs * p.gp.Ws * 4 >= 0x7fffffffull) break;
                p.inner = role == 0 ? bn_fast_down_plan(p.gp) : role == 1 ? bn_fast_up_plan(p.gp)
                                                                          : bn_fast_wgrad_plan(p.gp);
                found = p.inner.supported;
            }
        }
        // as measured first, then square (several families take square maps only), then larger
        for (int grow = 0; grow < 4 && !found; ++grow) {
            if (grow == 1) { if (hs == wsm) continue; hs = wsm = (hs > wsm ? hs : wsm); }
            if (grow >= 2) { hs *= 2; wsm *= 2; }
            if (2 * hs < g.Hb + oh || 2 * wsm < g.Wb + ow) continue;
            p.gp = g;
            p.gp.Hs = hs; p.gp.Ws = wsm; p.gp.Hb = 2 * hs; p.gp.Wb = 2 * wsm; p.gp.pt = p.gp.pl = 1;
            if ((size_t)g.N * g.Cb * p.gp.Hb * p.gp.Wb * 4 >= 0x7fffffffull) break;
            if ((size_t)g.N * g.Cs * p.gp.Hs * p.gp.Ws * 4 >= 0x7fffffffull) break;
            p.inner = role == 0 ? bn_fast_down_plan(p.gp) : role == 1 ? bn_fast_up_plan(p.gp)
                                                                      : bn_fast_wgrad_plan(p.gp);
            found = p.inner.supported;
        }
        if (!found) return p;
    }
    if ((size_t)g.N * g.Cb * p.gp.Hb * p.gp.Wb * 4 >= 0x7fffffffull) return p;
    p.big_bytes = align256((size_t)g.N * g.Cb * p.gp.Hb * p.gp.Wb * 4);
    p.small_bytes = align256((size_t)g.N * g.Cs * p.gp.Hs * p.gp.Ws * 4);
    p.inner_ws = p.inner.ws_bytes;
    snprintf(p.name, sizeof(p.name), "%s on zero-padded %dx%d", p.inner.kernel_name, p.gp.Hs, p.gp.Ws);
    p.ok = true;
    return p;
}
static inline size_t pad_ws_bytes(const PadPlan& p) { return p.big_bytes + p.small_bytes + p.inner_ws; }

// ---- spatial tiles with halos for 5x5 stride-2 layers whose maps exceed what the specialised
// kernels take (conv_pad.hip, k_tile_gather / k_tile_scatter): frames wider than 128 pixels
struct TilePlan {
    bool ok;
    BnTileAxis sh, sw, bh, bw;   // small / big side, rows / columns
    int T;                       // tiles per frame
    int nb;                      // frames per pass (the tiled copies stay below 2 GB each)
    int Dsh, Dsw;                // small-side tile
    bool edge;
    size_t big_bytes, small_bytes, inner_ws;
    char name[96];
};
static BnGeom tile_geom(const BnGeom& g, const TilePlan& p, int frames) {
    BnGeom gp = g;
    gp.N = frames * p.T;
    gp.Hs = p.Dsh; gp.Ws = p.Dsw; gp.Hb = 2 * p.Dsh; gp.Wb = 2 * p.Dsw; gp.pt = gp.pl = 1;
    return gp;
}
static BnFastPlan tile_inner(int role, const TilePlan& p, const BnGeom& gp) {
    if (p.edge)
        return role == 0 ? bn_edge_down_plan(gp) : role == 1 ? bn_edge_up_plan(gp) : bn_edge_wgrad_plan(gp);
    return role == 0 ? bn_fast_down_plan(gp) : role == 1 ? bn_fast_up_plan(gp) : bn_fast_wgrad_plan(gp);
}
// one axis: small length Ls, big length Lb, first tap pt outside, tile length S (small side)
static bool tile_axis(int Ls, int Lb, int pt, int S, bool may_fit, BnTileAxis* sm, BnTileAxis* bg) {
    if (may_fit && Ls <= S && Lb + pt - 1 <= 2 * S) {
        *sm = BnTileAxis{1, Ls, 0, Ls, S};
        *bg = BnTileAxis{1, Lb, pt - 1, Lb, 2 * S};
        return true;
    }
    const int V = S - 2, T = (Ls + V - 1) / V;
    if (Lb > 2 * V * T) return false;
    *sm = BnTileAxis{T, V, 1, V, S};
    *bg = BnTileAxis{T, 2 * V, 1 + pt, 2 * V, 2 * S};
    return true;
}
static TilePlan tile_plan(int role, const BnGeom& g) {
    TilePlan p;
    p.ok = false;
    if (force_generic() || g.R != 5 || g.S != 5 || g.stride != 2) return p;
    if (g.pt < 1 || g.pt > 2 || g.pl < 1 || g.pl > 2) return p;
    p.edge = g.Cb <= 2;
    if (p.edge) {
        if (g.Cs > 32) return p;
        const int hs16 = (g.Hs + 15) / 16 * 16;
        const int Sh = hs16 <= 64 ? hs16 : 64;
        if (!tile_axis(g.Hs, g.Hb, g.pt, Sh, hs16 <= 64, &p.sh, &p.bh)) return p;
        if (!tile_axis(g.Ws, g.Wb, g.pl, 64, true, &p.sw, &p.bw)) return p;
    } else {
        if (!tile_axis(g.Hs, g.Hb, g.pt, 32, true, &p.sh, &p.bh)) return p;
        if (!tile_axis(g.Ws, g.Wb, g.pl, 32, true, &p.sw, &p.bw)) return p;
    }
    p.T = p.sh.T * p.sw.T;
    if (p.T == 1) return p;                    // fits: the zero-padded embedding serves it
    p.Dsh = p.sh.D; p.Dsw = p.sw.D;
    const size_t big_frame = (size_t)p.T * g.Cb * 4 * p.Dsh * p.Dsw * 4;
    const size_t small_frame = (size_t)p.T * g.Cs * p.Dsh * p.Dsw * 4;
    const size_t per = big_frame > small_frame ? big_frame : small_frame;
    size_t nb = 0x7ff00000ull / per;
    if (nb < 1) return p;
    if (nb > (size_t)g.N) nb = g.N;
    p.nb = (int)nb;
    const BnGeom gp = tile_geom(g, p, p.nb);
    const BnFastPlan inner = tile_inner(role, p, gp);
    if (!inner.supported) return p;
    p.big_bytes = align256(big_frame * nb);
    p.small_bytes = align256(small_frame * nb);
    p.inner_ws = inner.ws_bytes;
    // (a last, shorter pass may pick another tiling of the same family: size for both)
    if (g.N % p.nb) {
        const BnFastPlan last = tile_inner(role, p, tile_geom(g, p, g.N % p.nb));
        if (!last.supported) return p;
        if (last.ws_bytes > p.inner_ws) p.inner_ws = last.ws_bytes;
    }
    snprintf(p.name, sizeof(p.name), "%s on %dx%d tiles of %dx%d", inner.kernel_name, p.sh.T, p.sw.T,
             p.Dsh, p.Dsw);
    p.ok = true;
    return p;
}
static inline size_t tile_ws_bytes(const TilePlan& p) { return p.big_bytes + p.small_bytes + p.inner_ws; }

// ---- kernels smaller than 5x5 with stride 2 (ae_arch_2.json: 4x4): 5x5 taps, the added ones zero
static size_t role_ws_need(int role, const BnGeom& g);
static bool served_fast(int role, const BnGeom& g);
static bool flip_plan(const BnGeom& g, BnGeom* gf, BnFastPlan* inner);
static bool bigk1_flip_plan(const BnGeom& g, BnGeom* gf);
// ---- single- / two-channel edge layers with more than 32 channels on the other side (1 -> 64):
// groups of 32 small-side channels on contiguous copies (gather-down, weight gradient)
static bool chan_plan(int role, const BnGeom& g, BnGeom* gg) {
    if (force_generic() || role == 1 || g.Cb > 2 || g.Cs <= 32 || (g.Cs & 31)) return false;
    if (g.R != 5 || g.S != 5 || g.stride != 2 || ((g.Hs * g.Ws) & 3)) return false;
    if ((size_t)g.N * g.Cs * g.Hs * g.Ws * 4 >= 0x7fffffffull) return false;
    *gg = g;
    gg->Cs = 32;
    return served_fast(role, *gg);
}
static inline size_t chan_bytes(const BnGeom& g) { return align256((size_t)g.N * 32 * g.Hs * g.Ws * 4); }
static bool served_fast(int role, const BnGeom& g) {
    BnGeom gg;
    if (role == 0 && (bn_s1c1_ok(g) || bn_s1in1_ok(g))) return true;
    if (chan_plan(role, g, &gg)) return true;
    if (g.Cb <= 4) {
        const BnFastPlan ed = role == 0 ? bn_edge_down_plan(g) : role == 1 ? bn_edge_up_plan(g)
                                                                          : bn_edge_wgrad_plan(g);
        if (ed.supported) return true;
    }
    const BnFastPlan pl = role == 0 ? bn_fast_down_plan(g) : role == 1 ? bn_fast_up_plan(g)
                                                                      : bn_fast_wgrad_plan(g);
    if (pl.supported) return true;
    if (role == 1) {
        BnGeom gf;
        BnFastPlan in;
        if (flip_plan(g, &gf, &in)) return true;
        if (bigk1_flip_plan(g, &gf)) return true;
    }
    return pad_plan(role, g).ok || tile_plan(role, g).ok;
}
// (a stride-1 3x3 layer comes here for its weight gradient only -- taps_plan -- and goes in at (1, 1) as well:
// the 3x3 window of taps is the one instantiation both strides share)
static inline bool taps_s1k3(const BnGeom& g) { return g.stride == 1 && g.R == 3 && g.S == 3 && g.pt <= 3 && g.pl <= 3; }
static inline int taps_dr(const BnGeom& g) { return ((g.stride == 2 && g.pt == 0 && g.R <= 4) || taps_s1k3(g)) ? 1 : 0; }
static inline int taps_ds(const BnGeom& g) { return ((g.stride == 2 && g.pl == 0 && g.S <= 4) || taps_s1k3(g)) ? 1 : 0; }
static bool taps_plan(int role, const BnGeom& g, BnGeom* g5) {
    // (stride 1 too: the index relation p * stride - pt + r does not care, and the stride-1 gather-down
    // kernel is instantiated for 3x3 and 5x5 -- a 4x4 layer becomes a 5x5 one where that kernel serves it)
    if (force_generic() || (g.stride != 2 && g.stride != 1) || g.R > 5 || g.S > 5 || (g.R == 5 && g.S == 5))
        return false;
    if (g.stride == 1 && g.R == 3 && g.S == 3 && role != 2) return false;      // served as it is
    if (g.R < 2 || g.S < 2) return false;
    *g5 = g;
    g5->R = g5->S = 5;
    // stride 2, first tap ON the frame (TF-"same" padding of a 3x3 kernel: pt = pl = 0): the taps go one row /
    // column into the 5x5 ones, which makes it a layer with the offsets (1, 1) the streamlined families take
    const int dr = taps_dr(g), ds = taps_ds(g);
    g5->pt = g.pt + dr;
    g5->pl = g.pl + ds;
    // stride 2: the fifth row and column of taps are zeros the 16-byte-DMA kernels need not multiply
    g5->KV = ((g.stride == 2 || taps_s1k3(g)) && g.R + dr <= 4 && g.S + ds <= 4) ? 4 : 0;
    // ... and neither the first ones of a kernel that went in at (1, 1): 9 products of 25 for a 3x3 layer
    g5->K0 = (g5->KV == 4 && dr == 1 && ds == 1) ? 1 : 0;
    return served_fast(role, *g5);
}
static inline size_t taps_bytes(const BnGeom& g) { return align256((size_t)g.Cs * g.Cb * 25 * sizeof(float)); }
// One-shot hint (round 6, bn_conv_taps_hint): the caller already holds the 5x5 copy of `w` (bn_conv_taps_pad, one
// launch for all the layers of a stack).  The next forward / data-gradient entry of THIS thread takes it if the
// weight pointer matches and drops it otherwise -- a hint never outlives one call.
static thread_local const float* g_taps_hint_w = nullptr;
static thread_local const float* g_taps_hint_w5 = nullptr;
struct TapsHintDrop { ~TapsHintDrop() { g_taps_hint_w = g_taps_hint_w5 = nullptr; } };   // every conv entry ends with no hint
static const float* taps_hint_take(const float* w) {
    const float* w5 = (w && g_taps_hint_w == w) ? g_taps_hint_w5 : nullptr;
    g_taps_hint_w = g_taps_hint_w5 = nullptr;
    return w5;
}

// ---- kernels larger than 5x5 with stride 2 (7x7, 9x9): conv_pad.hip, "Kernels LARGER than 5x5"
struct BigK {
    int kr[2], ofr[2], kc[2], ofc[2];   // parity of a phase's taps; tap u' of the phase is tap u = sgn u' + of
    int sgn;                            // +1: gather-down / weight gradient, -1: gather-up
    int Hy, Wy;                         // the phase maps
};
static bool bigk_enabled() {
    static int mode = -1;                              // BN_BIGK=0: off (the im2col detour, tuning builds)
    if (mode < 0) { const char* e = bn_tune_env("BN_BIGK"); mode = (e && e[0] == '0') ? 0 : 1; }
    return mode == 1;
}
// one axis: phase rho of the big map meets the taps 2u + k (k = parity of rho + pad, u < U) at small pixel
// p = i + o - u.  As a stride-1 5-tap kernel with padding p1 that is tap u' = u - o + p1 (gather-down, sgn +1) or
// u' = o + p1 - u (gather-up, sgn -1); false if no p1 in 0..4 holds both phases in 5 taps
static bool bigk_phase_axis(int R, int pad, int sgn, int* k, int* of, int* p1) {
    for (int q = 0; q <= 4; ++q) {
        bool ok = true;
        for (int rho = 0; rho < 2; ++rho) {
            k[rho] = (rho + pad) & 1;
            const int U = (R - k[rho] + 1) / 2, o = (rho + pad - k[rho]) / 2;
            if (sgn > 0) { of[rho] = o - q; ok = ok && q >= o && (U - 1) - o + q <= 4; }
            else { of[rho] = o + q; ok = ok && of[rho] <= 4 && of[rho] - (U - 1) >= 0; }
        }
        if (ok) { *p1 = q; return true; }
    }
    return false;
}
static bool bigk_plan(int role, const BnGeom& g, BnGeom* g5, BigK* k) {
    if (force_generic() || !bigk_enabled() || g.stride != 2 || g.CsS) return false;
    if (g.R < 6 || g.R > 10 || g.S < 6 || g.S > 10) return false;
    if ((g.Hb & 1) || (g.Wb & 3)) return false;
    int pt1 = 0, pl1 = 0;
    k->sgn = role == 1 ? -1 : 1;
    if (!bigk_phase_axis(g.R, g.pt, k->sgn, k->kr, k->ofr, &pt1)) return false;
    if (!bigk_phase_axis(g.S, g.pl, k->sgn, k->kc, k->ofc, &pl1)) return false;
    k->Hy = g.Hb / 2; k->Wy = g.Wb / 2;
    if ((size_t)g.N * 4 * g.Cb * k->Hy * k->Wy * 4 >= 0x7fffffffull) return false;
    *g5 = g;
    g5->R = g5->S = 5;
    g5->stride = 1; g5->pt = pt1; g5->pl = pl1; g5->KV = 0; g5->K0 = 0;
    if (role == 1) {
        // the phases are the OUTPUT of a gather-down from the small map
        g5->Cs = 4 * g.Cb; g5->Hs = k->Hy; g5->Ws = k->Wy;
        g5->Cb = g.Cs; g5->Hb = g.Hs; g5->Wb = g.Ws;
        return served_fast(0, *g5);
    }
    g5->Cb = 4 * g.Cb; g5->Hb = k->Hy; g5->Wb = k->Wy;
    return served_fast(role, *g5);
}
static inline size_t bigk_map_bytes(const BnGeom& g) { return align256((size_t)g.N * g.Cb * g.Hb * g.Wb * sizeof(float)); }
// ---- ... and with stride 1 (conv_pad.hip, k_shift_cat): 2 x 2 blocks of taps on four shifted copies of the big map,
// frames in blocks that keep the copies below 2 GB (the kernels' 32-bit offsets)
struct BigK1 { int L0r, L0c, dr[2], dc[2], Ho, Wo, nb; };
static size_t g_bigk1_block_bytes = 0x70000000ull;    // bytes of shifted copies per block of frames (test hook below)
extern "C" size_t bn_set_bigk1_block_bytes(size_t bytes) {
    const size_t prev = g_bigk1_block_bytes;
    g_bigk1_block_bytes = (bytes && bytes < 0x70000000ull) ? bytes : 0x70000000ull;
    return prev;
}
static bool bigk1_plan(int role, const BnGeom& g, BnGeom* g5, BigK1* k) {
    if (force_generic() || !bigk_enabled() || g.stride != 1 || g.CsS || role == 1) return false;
    if (g.R < 6 || g.R > 10 || g.S < 6 || g.S > 10) return false;
    k->L0r = (g.R + 1) / 2; k->L0c = (g.S + 1) / 2;
    k->dr[0] = -g.pt; k->dr[1] = k->L0r - g.pt;
    k->dc[0] = -g.pl; k->dc[1] = k->L0c - g.pl;
    k->Ho = g.Hs + 4; k->Wo = (g.Ws + 4 + 3) & ~3;
    const size_t per_frame = (size_t)4 * g.Cb * k->Ho * k->Wo * sizeof(float);
    if (per_frame >= 0x40000000ull) return false;
    const size_t nb = g_bigk1_block_bytes / per_frame > 0 ? g_bigk1_block_bytes / per_frame : 1;
    k->nb = nb < (size_t)g.N ? (int)nb : g.N;
    *g5 = g;
    g5->N = k->nb;
    g5->Cb = 4 * g.Cb; g5->Hb = k->Ho; g5->Wb = k->Wo;
    g5->R = g5->S = 5; g5->pt = g5->pl = 0; g5->KV = 0; g5->K0 = 0;
    if (!served_fast(role, *g5)) return false;
    if (g.N % k->nb) {                                  // the last, shorter block of frames
        BnGeom gl = *g5;
        gl.N = g.N % k->nb;
        if (!served_fast(role, gl)) return false;
    }
    return true;
}
static inline size_t bigk1_map_bytes(const BnGeom& g, const BigK1& k) {
    return align256((size_t)k.nb * 4 * g.Cb * k.Ho * k.Wo * sizeof(float));
}
static size_t bigk1_inner_ws(int role, const BnGeom& g, const BnGeom& g5, const BigK1& k) {
    size_t need = role_ws_need(role, g5);
    if (g.N % k.nb) {
        BnGeom gl = g5;
        gl.N = g.N % k.nb;
        const size_t l = role_ws_need(role, gl);
        if (l > need) need = l;
    }
    return need;
}
// the gather-up role of such a layer: the gather-down of the flipped layer (as flip_plan, whose inner plan is the
// shifted-copies one)
static bool bigk1_flip_plan(const BnGeom& g, BnGeom* gf) {
    if (force_generic() || g.stride != 1 || g.pt > g.R - 1 || g.pl > g.S - 1) return false;
    *gf = g;
    gf->Cs = g.Cb; gf->Hs = g.Hb; gf->Ws = g.Wb;
    gf->Cb = g.Cs; gf->Hb = g.Hs; gf->Wb = g.Ws;
    gf->pt = g.R - 1 - g.pt; gf->pl = g.S - 1 - g.pl;
    if (bn_s1c1_ok(*gf)) return true;                   // onto one / two channels: the vector kernel (conv_edge.hip)
    BnGeom g5;
    BigK1 k;
    return g.R >= 6 && g.S >= 6 && bigk1_plan(0, *gf, &g5, &k);
}
static inline size_t bigk_w_bytes(const BnGeom& g) { return align256((size_t)g.Cs * 4 * g.Cb * 25 * sizeof(float)); }

// stride == kernel layers between maps other than 8x8 and 2x2 (the last layer of 64x48 / 192x160 frames): the
// weight gradient as ONE GEMM over the (permuted) windows instead of the first-generation direct kernel
static bool s5_wgrad_by_col(const BnGeom& g) {
    static int mode = -1;                              // BN_S5_WGRAD_COL=0: off (tuning build)
    if (mode < 0) { const char* e = bn_tune_env("BN_S5_WGRAD_COL"); mode = (e && e[0] == '0') ? 0 : 1; }
    if (!mode || force_generic() || g.stride != 5 || g.R != 5 || g.S != 5) return false;
    if (bn_qg2_wgrad_supported(g) || bn_qgemm_supported(g)) return false;
    return bn_s5_wgrad_plan(g).supported && bn_col_ok(g) && (size_t)g.N * g.Hs * g.Ws >= 512;
}

// ---- stride-1 gather-up (transposed-conv forward, conv data gradient) as a gather-down with the channel
// roles swapped and the taps reversed (conv_pad.hip, k_flip_taps): no im2col / col2im
static bool flip_plan(const BnGeom& g, BnGeom* gf, BnFastPlan* inner) {
    if (force_generic() || g.stride != 1 || g.R != g.S || (g.R != 3 && g.R != 5)) return false;
    if (g.pt > g.R - 1 || g.pl > g.S - 1) return false;
    *gf = g;
    gf->Cs = g.Cb; gf->Hs = g.Hb; gf->Ws = g.Wb;
    gf->Cb = g.Cs; gf->Hb = g.Hs; gf->Wb = g.Ws;
    gf->pt = g.R - 1 - g.pt; gf->pl = g.S - 1 - g.pl;
    *inner = bn_fast_down_plan(*gf);
    return inner->supported;
}
static inline size_t flip_bytes(const BnGeom& g) { return align256((size_t)g.Cs * g.Cb * g.R * g.S * sizeof(float)); }

static int run_down(int family, const float* big, const float* w, const float* bias, float* out,
                    const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                    void* ws, size_t ws_bytes, hipStream_t st) {
    const bool generic = force_generic() || !aligned16_all(big, w, out, dact_src);
    const float* w5 = taps_hint_take(w);
    BnGeom g5;
    if (!generic && w5 && aligned16_all(w5, w5, w5) && taps_plan(0, g, &g5))        // the caller's padded copy
        return run_down(family, big, w5, bias, out, dact_src, g5, act, dact, slope, ws, ws_bytes, st);
    if (!generic && taps_plan(0, g, &g5)) {
        const size_t wb = taps_bytes(g);
        if (!ws || ws_bytes < wb + role_ws_need(0, g5)) return BN_E_WORKSPACE;
        const int rc = bn_launch_pad_taps(w, (float*)ws, (size_t)g.Cs * g.Cb, g.R, g.S, st, taps_dr(g), taps_ds(g));
        if (rc) return rc;
        return run_down(family, big, (const float*)ws, bias, out, dact_src, g5, act, dact, slope,
                        (char*)ws + wb, ws_bytes - wb, st);
    }
    if (!generic && bn_s1in1_ok(g)) {
        static const char* names1[4] = {"k_down_s1_in1<3>", "k_down_s1_in1<5>", "k_down_s1_in1<7>", "k_down_s1_in1<9>"};
        BnProfScope prof(family, g.Cb, g.Cs, (g.R == 5 && (g.Cs & 15) == 0) ? "k_down_s1_in1m" : names1[(g.R - 3) / 2], st);
        return bn_launch_s1in1(big, w, bias, out, dact_src, g, act, dact, slope, st);
    }
    if (!generic && bn_s1c1_ok(g)) {
        static const char* names[4] = {"k_down_s1_c1<3>", "k_down_s1_c1<5>", "k_down_s1_c1<7>", "k_down_s1_c1<9>"};
        BnProfScope prof(family, g.Cb, g.Cs, names[(g.R - 3) / 2], st);
        return bn_launch_s1c1(big, w, bias, out, dact_src, g, act, dact, slope, st);
    }
    BigK bk;
    if (!generic && bigk_plan(0, g, &g5, &bk)) {
        const size_t xb = bigk_map_bytes(g), wb = bigk_w_bytes(g);
        if (!ws || ws_bytes < xb + wb + role_ws_need(0, g5)) return BN_E_WORKSPACE;
        float* X = (float*)ws;
        float* w1 = (float*)((char*)ws + xb);
        int rc = bn_launch_space_to_depth(big, X, g.N, g.Cb, bk.Hy, bk.Wy, st);
        if (rc) return rc;
        rc = bn_launch_bigk_phase_pack(w, w1, g.Cs, g.Cb, g.R, g.S, bk.kr, bk.ofr, bk.kc, bk.ofc, 1, 0, st);
        if (rc) return rc;
        return run_down(family, X, w1, bias, out, dact_src, g5, act, dact, slope, (char*)ws + xb + wb,
                        ws_bytes - xb - wb, st);
    }
    BigK1 b1;
    if (!generic && bigk1_plan(0, g, &g5, &b1)) {
        const size_t xb = bigk1_map_bytes(g, b1), wb = bigk_w_bytes(g), iw = bigk1_inner_ws(0, g, g5, b1);
        if (!ws || ws_bytes < xb + wb + iw) return BN_E_WORKSPACE;
        float* xcat = (float*)ws;
        float* w5 = (float*)((char*)ws + xb);
        int rc = bn_launch_bigk_pack(w, w5, g.Cs, g.Cb, g.R, g.S, b1.L0r, b1.L0c, st);
        if (rc) return rc;
        const size_t fb = (size_t)g.Cb * g.Hb * g.Wb, fs = (size_t)g.Cs * g.Hs * g.Ws;
        for (int n0 = 0; n0 < g.N; n0 += b1.nb) {
            BnGeom gb = g5;
            gb.N = g.N - n0 < b1.nb ? g.N - n0 : b1.nb;
            rc = bn_launch_shift_cat(big + n0 * fb, xcat, gb.N, g.Cb, g.Hb, g.Wb, b1.Ho, b1.Wo, b1.dr[0], b1.dr[1],
                                     b1.dc[0], b1.dc[1], st);
            if (rc) return rc;
            rc = run_down(family, xcat, w5, bias, out + n0 * fs, dact_src ? dact_src + n0 * fs : nullptr, gb, act,
                          dact, slope, (char*)ws + xb + wb, ws_bytes - xb - wb, st);
            if (rc) return rc;
        }
        return 0;
    }
    if (!generic && chan_plan(0, g, &g5)) {
        const size_t cb = chan_bytes(g);
        if (!ws || ws_bytes < cb + role_ws_need(0, g5)) return BN_E_WORKSPACE;
        const int PQ = g.Hs * g.Ws;
        // the edge kernels write a group straight into its channel window of the output (frames of
        // Cs channels, BnGeom::CsS) and read the mask there: no contiguous copy, no k_chan_copy
        const bool epi_ok = dact_src ? (act == BN_ACT_NONE && dact == BN_ACT_LRELU)
                                     : (act == BN_ACT_NONE || act == BN_ACT_LRELU);
        if (g.Cb == 1 && bn_edge_down_plan(g5).supported && bn_edge_down_plan(g5).variant != 9 && epi_ok) {
            BnGeom gw = g5;
            gw.CsS = g.Cs;
            BnProfScope prof(family, g.Cb, g.Cs, bn_edge_down_kernel_name(g5, act, dact_src != nullptr, false), st);
            for (int c0 = 0; c0 < g.Cs; c0 += 32) {
                const int rc = bn_launch_edge_down(big, w + (size_t)c0 * g.Cb * 25, bias ? bias + c0 : nullptr,
                                                   out + (size_t)c0 * PQ,
                                                   dact_src ? dact_src + (size_t)c0 * PQ : nullptr, gw, act,
                                                   dact, slope, st);
                if (rc) return rc;
            }
            return 0;
        }
        for (int c0 = 0; c0 < g.Cs; c0 += 32) {
            int rc = run_down(family, big, w + (size_t)c0 * g.Cb * 25, bias ? bias + c0 : nullptr, (float*)ws,
                              nullptr, g5, act, BN_ACT_NONE, slope, (char*)ws + cb, ws_bytes - cb, st);
            if (rc) return rc;
            rc = bn_launch_chan_copy((const float*)ws, out, g.N, 32, 0, g.Cs, c0, 32, PQ, dact_src, dact, slope, st);
            if (rc) return rc;
        }
        return 0;
    }
    if (!generic) {
        const BnFastPlan ed = bn_edge_down_plan(g);
        // epilogues the edge kernel is instantiated for: plain / LeakyReLU forward, or a data
        // gradient carrying the LeakyReLU' mask of the layer below
        const bool epi_ok = dact_src ? (act == BN_ACT_NONE && dact == BN_ACT_LRELU)
                                     : (act == BN_ACT_NONE || act == BN_ACT_LRELU);
        if (ed.supported && epi_ok) {
            const char* name = bn_edge_down_kernel_name(g, act, dact_src != nullptr, false);
            BnProfScope prof(family, g.Cb, g.Cs, name, st);
            return bn_launch_edge_down(big, w, bias, out, dact_src, g, act, dact, slope, st);
        }
    }
    if (!generic && bn_qgemm_supported(g)) {
        if (!ws || ws_bytes < bn_qgemm_ws_bytes(0, g)) return BN_E_WORKSPACE;
        BnProfScope prof(family, g.Cb, g.Cs, "k_qgemm<0>", st);
        return bn_launch_qgemm_down(big, w, bias, out, dact_src, g, act, dact, slope, ws, st);
    }
    if (!generic && bn_s5win_supported(g)) {
        if (!ws || ws_bytes < bn_s5win_ws_bytes(0, g)) return BN_E_WORKSPACE;
        BnProfScope prof(family, g.Cb, g.Cs, "k_s5win<down>", st);
        return bn_launch_s5win_down(big, w, bias, out, dact_src, g, act, dact, slope, ws, st);
    }
    BnFastPlan plan = bn_fast_down_plan(g);
    if (generic) plan.supported = false;
    if (!generic && !plan.supported && bn_s5_down_small_ok(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_im2col_s5 + k_gemm_mfma", st);
        return bn_launch_s5_down_small(big, w, bias, out, dact_src, g, act, dact, slope, ws, ws_bytes, st);
    }
    if (!generic && !plan.supported) {
        const PadPlan pp = pad_plan(0, g);
        // (the edge kernel is instantiated for plain / LeakyReLU epilogues)
        if (pp.ok && !(pp.edge && act != BN_ACT_NONE && act != BN_ACT_LRELU)) {
            if (!ws || ws_bytes < pad_ws_bytes(pp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, pp.name, st);
            float* bigp = (float*)ws;
            float* outp = (float*)((char*)ws + pp.big_bytes);
            void* iws = (char*)ws + pp.big_bytes + pp.small_bytes;
            int rc = bn_launch_pad2d(big, bigp, (size_t)g.N * g.Cb, g.Hb, g.Wb, pp.gp.Hb, pp.gp.Wb, pp.oh, pp.ow, st);
            if (rc) return rc;
            rc = pp.edge ? bn_launch_edge_down(bigp, w, bias, outp, nullptr, pp.gp, act, BN_ACT_NONE, slope, st)
                         : bn_launch_down_fast(pp.inner, bigp, w, bias, outp, nullptr, pp.gp, act, BN_ACT_NONE,
                                               slope, iws, st);
            if (rc) return rc;
            return bn_launch_crop2d(outp, out, (size_t)g.N * g.Cs, g.Hs, g.Ws, pp.gp.Hs, pp.gp.Ws, 0, 0,
                                    dact_src, dact, slope, st);
        }
    }
    if (!generic && !plan.supported) {
        const TilePlan tp = tile_plan(0, g);
        if (tp.ok && !(tp.edge && act != BN_ACT_NONE && act != BN_ACT_LRELU)) {
            if (!ws || ws_bytes < tile_ws_bytes(tp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, tp.name, st);
            float* bigp = (float*)ws;
            float* outp = (float*)((char*)ws + tp.big_bytes);
            void* iws = (char*)ws + tp.big_bytes + tp.small_bytes;
            for (int n0 = 0; n0 < g.N; n0 += tp.nb) {
                const int nf = g.N - n0 < tp.nb ? g.N - n0 : tp.nb;
                const BnGeom gp = tile_geom(g, tp, nf);
                const BnFastPlan inner = tile_inner(0, tp, gp);
                const size_t ob = (size_t)n0 * g.Cb * g.Hb * g.Wb, os = (size_t)n0 * g.Cs * g.Hs * g.Ws;
                int rc = bn_launch_tile_gather(big + ob, bigp, nf, g.Cb, g.Hb, g.Wb, tp.bh, tp.bw, 0, st);
                if (rc) return rc;
                rc = tp.edge ? bn_launch_edge_down(bigp, w, bias, outp, nullptr, gp, act, BN_ACT_NONE, slope, st)
                             : bn_launch_down_fast(inner, bigp, w, bias, outp, nullptr, gp, act, BN_ACT_NONE,
                                                   slope, iws, st);
                if (rc) return rc;
                rc = bn_launch_tile_scatter(outp, out + os, nf, g.Cs, g.Hs, g.Ws, tp.sh, tp.sw,
                                            dact_src ? dact_src + os : nullptr, dact, slope, st);
                if (rc) return rc;
            }
            return 0;
        }
    }
    if (!generic && !plan.supported && bn_col_ok(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_im2col + k_gemm_mfma", st);
        return bn_launch_col_down(big, w, bias, out, dact_src, g, act, dact, slope, ws, ws_bytes, st);
    }
    BnProfScope prof(family, g.Cb, g.Cs, plan.supported ? plan.kernel_name : "k_down_generic", st);
    if (plan.supported) {
        if (!ws_ok(plan, ws, ws_bytes)) return BN_E_WORKSPACE;
        return bn_launch_down_fast(plan, big, w, bias, out, dact_src, g, act, dact, slope, ws, st);
    }
    return bn_launch_down_generic(big, w, bias, out, dact_src, g, act, dact, slope, st);
}

static int run_up(int family, const float* small, const float* w, const float* bias, float* out,
                  const float* dact_src, const BnGeom& g, int act, int dact, float slope,
                  void* ws, size_t ws_bytes, hipStream_t st) {
    const bool generic = force_generic() || !aligned16_all(small, w, out, dact_src);
    const float* w5 = taps_hint_take(w);
    BnGeom g5;
    if (!generic && w5 && aligned16_all(w5, w5, w5) && taps_plan(1, g, &g5))        // the caller's padded copy
        return run_up(family, small, w5, bias, out, dact_src, g5, act, dact, slope, ws, ws_bytes, st);
    if (!generic && taps_plan(1, g, &g5)) {
        const size_t wb = taps_bytes(g);
        if (!ws || ws_bytes < wb + role_ws_need(1, g5)) return BN_E_WORKSPACE;
        const int rc = bn_launch_pad_taps(w, (float*)ws, (size_t)g.Cs * g.Cb, g.R, g.S, st, taps_dr(g), taps_ds(g));
        if (rc) return rc;
        return run_up(family, small, (const float*)ws, bias, out, dact_src, g5, act, dact, slope,
                      (char*)ws + wb, ws_bytes - wb, st);
    }
    BigK bk;
    if (!generic && bigk_plan(1, g, &g5, &bk)) {
        const size_t yb = bigk_map_bytes(g), wb = bigk_w_bytes(g);
        if (!ws || ws_bytes < yb + wb + role_ws_need(0, g5)) return BN_E_WORKSPACE;
        float* y = (float*)ws;
        float* w1 = (float*)((char*)ws + yb);
        int rc = bn_launch_bigk_phase_pack(w, w1, g.Cs, g.Cb, g.R, g.S, bk.kr, bk.ofr, bk.kc, bk.ofc, -1, 1, st);
        if (rc) return rc;
        rc = run_down(family, small, w1, nullptr, y, nullptr, g5, BN_ACT_NONE, BN_ACT_NONE, slope,
                      (char*)ws + yb + wb, ws_bytes - yb - wb, st);
        if (rc) return rc;
        return bn_launch_depth_to_space(y, out, bias, dact_src, g.N, g.Cb, bk.Hy, bk.Wy, act, dact, slope, st);
    }
    if (!generic && bn_qg2_up_supported(g, act, dact_src ? dact : BN_ACT_NONE)) {
        BnProfScope prof(family, g.Cs, g.Cb, "k_qg2_up", st);
        return bn_launch_qg2_up(small, w, bias, out, dact_src, g, act, dact, slope, st);
    }
    if (!generic && bn_qgemm_supported(g)) {
        if (bn_qgemm_ws_bytes(1, g) && (!ws || ws_bytes < bn_qgemm_ws_bytes(1, g))) return BN_E_WORKSPACE;
        BnProfScope prof(family, g.Cs, g.Cb, "k_qgemm<1>", st);
        return bn_launch_qgemm_up(small, w, bias, out, dact_src, g, act, dact, slope, ws, st);
    }
    if (!generic && bn_s5win_supported(g)) {
        BnProfScope prof(family, g.Cs, g.Cb, "k_s5win<up>", st);
        return bn_launch_s5win_up(small, w, bias, out, dact_src, g, act, dact, slope, st);
    }
    if (!generic) {
        const BnFastPlan s5 = bn_s5_up_plan(g);
        if (s5.supported) {
            BnProfScope prof(family, g.Cs, g.Cb, s5.kernel_name, st);
            return bn_launch_up_s5(small, w, bias, out, dact_src, g, act, dact, slope, st);
        }
    }
    if (!generic && !dact_src) {
        const BnFastPlan ed = bn_edge_up_plan(g);
        if (ed.supported) {
            BnProfScope prof(family, g.Cs, g.Cb, ed.kernel_name, st);
            return bn_launch_edge_up(small, w, bias, out, g, act, slope, st);
        }
    }
    BnFastPlan plan = bn_fast_up_plan(g);
    if (generic) plan.supported = false;
    if (!generic && !plan.supported) {
        const PadPlan pp = pad_plan(1, g);
        if (pp.ok) {
            if (!ws || ws_bytes < pad_ws_bytes(pp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cs, g.Cb, pp.name, st);
            float* outp = (float*)ws;
            float* smallp = (float*)((char*)ws + pp.big_bytes);
            void* iws = (char*)ws + pp.big_bytes + pp.small_bytes;
            int rc = 0;
            if (pp.gp.Hs == g.Hs && pp.gp.Ws == g.Ws) smallp = const_cast<float*>(small);   // only the output grows
            else rc = bn_launch_pad2d(small, smallp, (size_t)g.N * g.Cs, g.Hs, g.Ws, pp.gp.Hs, pp.gp.Ws, 0, 0, st);
            if (rc) return rc;
            rc = pp.edge ? bn_launch_edge_up(smallp, w, bias, outp, pp.gp, act, slope, st)
                         : bn_launch_up_fast(pp.inner, smallp, w, bias, outp, nullptr, pp.gp, act, BN_ACT_NONE,
                                             slope, iws, st);
            if (rc) return rc;
            return bn_launch_crop2d(outp, out, (size_t)g.N * g.Cb, g.Hb, g.Wb, pp.gp.Hb, pp.gp.Wb, pp.oh,
                                    pp.ow, dact_src, dact, slope, st);
        }
    }
    if (!generic && !plan.supported) {
        const TilePlan tp = tile_plan(1, g);
        if (tp.ok) {
            if (!ws || ws_bytes < tile_ws_bytes(tp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cs, g.Cb, tp.name, st);
            float* outp = (float*)ws;
            float* smallp = (float*)((char*)ws + tp.big_bytes);
            void* iws = (char*)ws + tp.big_bytes + tp.small_bytes;
            for (int n0 = 0; n0 < g.N; n0 += tp.nb) {
                const int nf = g.N - n0 < tp.nb ? g.N - n0 : tp.nb;
                const BnGeom gp = tile_geom(g, tp, nf);
                const BnFastPlan inner = tile_inner(1, tp, gp);
                const size_t ob = (size_t)n0 * g.Cb * g.Hb * g.Wb, os = (size_t)n0 * g.Cs * g.Hs * g.Ws;
                int rc = bn_launch_tile_gather(small + os, smallp, nf, g.Cs, g.Hs, g.Ws, tp.sh, tp.sw, 0, st);
                if (rc) return rc;
                rc = tp.edge ? bn_launch_edge_up(smallp, w, bias, outp, gp, act, slope, st)
                             : bn_launch_up_fast(inner, smallp, w, bias, outp, nullptr, gp, act, BN_ACT_NONE,
                                                 slope, iws, st);
                if (rc) return rc;
                rc = bn_launch_tile_scatter(outp, out + ob, nf, g.Cb, g.Hb, g.Wb, tp.bh, tp.bw,
                                            dact_src ? dact_src + ob : nullptr, dact, slope, st);
                if (rc) return rc;
            }
            return 0;
        }
    }
    if (!generic && !plan.supported) {
        BnGeom gf;
        BnFastPlan in;
        if (flip_plan(g, &gf, &in)) {
            const size_t fb = flip_bytes(g);
            if (!ws || ws_bytes < fb + in.ws_bytes) return BN_E_WORKSPACE;
            static char names[4][96];
            static int slot = 0;
            char* nm = names[slot = (slot + 1) & 3];
            snprintf(nm, 96, "%s on reversed taps", in.kernel_name);
            BnProfScope prof(family, g.Cs, g.Cb, nm, st);
            const int rc = bn_launch_flip_taps(w, (float*)ws, g.Cs, g.Cb, g.R * g.S, st);
            if (rc) return rc;
            return bn_launch_down_fast(in, small, (const float*)ws, bias, out, dact_src, gf, act, dact, slope,
                                       (char*)ws + fb, st);
        }
        if (bigk1_flip_plan(g, &gf)) {
            const size_t fb = flip_bytes(g);
            if (!ws || ws_bytes < fb + role_ws_need(0, gf)) return BN_E_WORKSPACE;
            const int rc = bn_launch_flip_taps(w, (float*)ws, g.Cs, g.Cb, g.R * g.S, st);
            if (rc) return rc;
            return run_down(family, small, (const float*)ws, bias, out, dact_src, gf, act, dact, slope,
                            (char*)ws + fb, ws_bytes - fb, st);
        }
    }
    if (!generic && !plan.supported && bn_col_ok(g)) {
        BnProfScope prof(family, g.Cs, g.Cb, "k_gemm_mfma + k_col2im", st);
        return bn_launch_col_up(small, w, bias, out, dact_src, g, act, dact, slope, ws, ws_bytes, st);
    }
    BnProfScope prof(family, g.Cs, g.Cb, plan.supported ? plan.kernel_name : "k_up_generic", st);
    if (plan.supported) {
        if (!ws_ok(plan, ws, ws_bytes)) return BN_E_WORKSPACE;
        return bn_launch_up_fast(plan, small, w, bias, out, dact_src, g, act, dact, slope, ws, st);
    }
    return bn_launch_up_generic(small, w, bias, out, dact_src, g, act, dact, slope, st);
}

static int run_wgrad(int family, const float* small, const float* big, float* dw,
                     const BnGeom& g, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                     float* db, int bias_side, bool* bias_done) {
    const bool generic = force_generic() || !aligned16_all(small, big, dw);
    BnGeom g5;
    if (!generic && taps_plan(2, g, &g5)) {
        // dw5 and (when the inner kernel yields it) the bias gradient are WRITTEN into scratch; the
        // crop kernel adds both to the caller's tensors (the bias row sits behind dw5, inside the
        // 256-byte rounding of taps_bytes or in the 4 KB the plan adds for it)
        const size_t wb = taps_bytes(g) + 4096;
        if (!ws || ws_bytes < wb + role_ws_need(2, g5)) return BN_E_WORKSPACE;
        const int nb = bias_side == 1 ? g.Cs : g.Cb;
        float* db5 = (db && nb <= 1024) ? (float*)((char*)ws + taps_bytes(g)) : nullptr;
        bool done = false;
        int rc = run_wgrad(family, small, big, (float*)ws, g5, 0, (char*)ws + wb, ws_bytes - wb, st,
                           db5, bias_side, &done);
        if (rc) return rc;
        if (done && bias_done) *bias_done = true;
        return bn_launch_crop_taps((const float*)ws, dw, (size_t)g.Cs * g.Cb, g.R, g.S, accumulate, st,
                                   done ? db5 : nullptr, db, nb, taps_dr(g), taps_ds(g));
    }
    BigK bk;
    if (!generic && bigk_plan(2, g, &g5, &bk)) {
        // as above: dw1 and the small side's bias gradient are written into scratch, k_bigk_phase_unpack adds them;
        // the big side's channel sums are left to the caller
        const size_t xb = bigk_map_bytes(g), wb = bigk_w_bytes(g) + 4096;
        if (!ws || ws_bytes < xb + wb + role_ws_need(2, g5)) return BN_E_WORKSPACE;
        float* X = (float*)ws;
        float* dw1 = (float*)((char*)ws + xb);
        float* db5 = (db && bias_side == 1 && g.Cs <= 1024) ? (float*)((char*)ws + xb + bigk_w_bytes(g)) : nullptr;
        int rc = bn_launch_space_to_depth(big, X, g.N, g.Cb, bk.Hy, bk.Wy, st);
        if (rc) return rc;
        bool done = false;
        rc = run_wgrad(family, small, X, dw1, g5, 0, (char*)ws + xb + wb, ws_bytes - xb - wb, st, db5,
                       bias_side, &done);
        if (rc) return rc;
        if (done && bias_done) *bias_done = true;
        return bn_launch_bigk_phase_unpack(dw1, dw, g.Cs, g.Cb, g.R, g.S, bk.kr, bk.ofr, bk.kc, bk.ofc, accumulate,
                                           done ? db5 : nullptr, db, g.Cs, st);
    }
    BigK1 b1;
    if (!generic && bigk1_plan(2, g, &g5, &b1)) {
        const size_t xb = bigk1_map_bytes(g, b1), wb = bigk_w_bytes(g) + 4096, iw = bigk1_inner_ws(2, g, g5, b1);
        if (!ws || ws_bytes < xb + wb + iw) return BN_E_WORKSPACE;
        float* xcat = (float*)ws;
        float* dw5 = (float*)((char*)ws + xb);
        // (the small side's bias sums ride with the inner kernel only when one block of frames holds the batch)
        float* db5 = (db && bias_side == 1 && g.Cs <= 1024 && b1.nb == g.N)
                         ? (float*)((char*)ws + xb + bigk_w_bytes(g)) : nullptr;
        const size_t fb = (size_t)g.Cb * g.Hb * g.Wb, fs = (size_t)g.Cs * g.Hs * g.Ws;
        bool done = false;
        for (int n0 = 0; n0 < g.N; n0 += b1.nb) {
            BnGeom gb = g5;
            gb.N = g.N - n0 < b1.nb ? g.N - n0 : b1.nb;
            int rc = bn_launch_shift_cat(big + n0 * fb, xcat, gb.N, g.Cb, g.Hb, g.Wb, b1.Ho, b1.Wo, b1.dr[0],
                                         b1.dr[1], b1.dc[0], b1.dc[1], st);
            if (rc) return rc;
            rc = run_wgrad(family, small + n0 * fs, xcat, dw5, gb, n0 > 0, (char*)ws + xb + wb, ws_bytes - xb - wb,
                           st, db5, bias_side, &done);
            if (rc) return rc;
        }
        if (done && bias_done) *bias_done = true;
        return bn_launch_bigk_unpack(dw5, dw, g.Cs, g.Cb, g.R, g.S, b1.L0r, b1.L0c, accumulate,
                                     done ? db5 : nullptr, db, g.Cs, st);
    }
    if (!generic && chan_plan(2, g, &g5)) {
        const size_t cb = chan_bytes(g);
        if (!ws || ws_bytes < cb + role_ws_need(2, g5)) return BN_E_WORKSPACE;
        const int PQ = g.Hs * g.Ws;
        bool all_done = db && bias_side == 1;
        // k_wgrad_c1d reads a group from its channel window of the small tensor (BnGeom::CsS)
        const BnFastPlan edw = bn_edge_wgrad_plan(g5);
        if (g.Cb == 1 && edw.supported && edw.variant != 1 && g5.pt == 1 && g5.pl == 1) {
            BnGeom gw = g5;
            gw.CsS = g.Cs;
            if (!ws_ok(edw, (char*)ws + cb, ws_bytes - cb)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, edw.kernel_name, st);
            for (int c0 = 0; c0 < g.Cs; c0 += 32) {
                bool done = false;
                const int rc = bn_launch_edge_wgrad(edw, small + (size_t)c0 * PQ, big, dw + (size_t)c0 * g.Cb * 25,
                                                    gw, accumulate, (char*)ws + cb, st,
                                                    (db && bias_side == 1) ? db + c0 : nullptr, bias_side, &done);
                if (rc) return rc;
                if (db && bias_side == 1 && !done) {
                    if (c0 > 0) return BN_E_BADARG;
                    all_done = false;
                    db = nullptr;
                }
            }
            if (bias_done && all_done) *bias_done = true;
            return 0;
        }
        for (int c0 = 0; c0 < g.Cs; c0 += 32) {
            int rc = bn_launch_chan_copy(small, (float*)ws, g.N, g.Cs, c0, 32, 0, 32, PQ, nullptr, 0, 0.f, st);
            if (rc) return rc;
            bool done = false;
            // (a bias gradient over the big side would be the same sum once per group)
            rc = run_wgrad(family, (const float*)ws, big, dw + (size_t)c0 * g.Cb * 25, g5, accumulate,
                           (char*)ws + cb, ws_bytes - cb, st, (db && bias_side == 1) ? db + c0 : nullptr,
                           bias_side, &done);
            if (rc) return rc;
            if (db && bias_side == 1 && !done) {
                if (c0 > 0) return BN_E_BADARG;
                all_done = false;
                db = nullptr;
            }
        }
        if (bias_done && all_done) *bias_done = true;
        return 0;
    }
    if (!generic && bn_qg2_wgrad_supported(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_qg2_wgrad", st);
        // the bias gradient of either side is a by-product (the operand tiles pass through LDS)
        const int rc = bn_launch_qg2_wgrad(small, big, dw, g, accumulate, db, db ? bias_side : 0, st);
        if (rc == 0 && db && (bias_side == 1 || bias_side == 2) && bias_done) *bias_done = true;
        return rc;
    }
    if (!generic && bn_qgemm_supported(g)) {
        if (!ws || ws_bytes < bn_qgemm_ws_bytes(2, g)) return BN_E_WORKSPACE;
        BnProfScope prof(family, g.Cb, g.Cs, "k_qgemm<2>", st);
        return bn_launch_qgemm_wgrad(small, big, dw, g, accumulate, ws, st);
    }
    if (!generic && bn_s5win_supported(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_s5win<wgrad>", st);
        return bn_launch_s5win_wgrad(small, big, dw, g, accumulate, st);
    }
    if (!generic && s5_wgrad_by_col(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_im2col + k_gemm_mfma (dW, stride 5)", st);
        return bn_launch_col_wgrad(small, big, dw, g, accumulate, ws, ws_bytes, st, db, bias_side, bias_done);
    }
    if (!generic) {
        const BnFastPlan s5 = bn_s5_wgrad_plan(g);
        if (s5.supported) {
            BnProfScope prof(family, g.Cb, g.Cs, s5.kernel_name, st);
            return bn_launch_wgrad_s5(small, big, dw, g, accumulate, st);
        }
    }
    if (!generic) {
        const BnFastPlan ed = bn_edge_wgrad_plan(g);
        if (ed.supported) {
            if (!ws_ok(ed, ws, ws_bytes)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, ed.kernel_name, st);
            return bn_launch_edge_wgrad(ed, small, big, dw, g, accumulate, ws, st, db, bias_side,
                                        bias_done);
        }
    }
    BnFastPlan plan = bn_fast_wgrad_plan(g);
    if (generic) plan.supported = false;
    if (!generic && !plan.supported) {
        const PadPlan pp = pad_plan(2, g);
        if (pp.ok) {
            if (!ws || ws_bytes < pad_ws_bytes(pp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, pp.name, st);
            float* bigp = (float*)ws;
            float* smallp = (float*)((char*)ws + pp.big_bytes);
            void* iws = (char*)ws + pp.big_bytes + pp.small_bytes;
            int rc = bn_launch_pad2d(big, bigp, (size_t)g.N * g.Cb, g.Hb, g.Wb, pp.gp.Hb, pp.gp.Wb, pp.oh, pp.ow, st);
            if (rc) return rc;
            if (pp.gp.Hs == g.Hs && pp.gp.Ws == g.Ws) smallp = const_cast<float*>(small);   // only the big map grows
            else rc = bn_launch_pad2d(small, smallp, (size_t)g.N * g.Cs, g.Hs, g.Ws, pp.gp.Hs, pp.gp.Ws, 0, 0, st);
            if (rc) return rc;
            if (pp.edge)
                return bn_launch_edge_wgrad(pp.inner, smallp, bigp, dw, pp.gp, accumulate, iws, st, db,
                                            bias_side, bias_done);
            return bn_launch_wgrad_fast(pp.inner, smallp, bigp, dw, pp.gp, accumulate, iws, st, db, bias_side,
                                        bias_done);
        }
    }
    if (!generic && !plan.supported) {
        const TilePlan tp = tile_plan(2, g);
        if (tp.ok) {
            if (!ws || ws_bytes < tile_ws_bytes(tp)) return BN_E_WORKSPACE;
            BnProfScope prof(family, g.Cb, g.Cs, tp.name, st);
            float* bigp = (float*)ws;
            float* smallp = (float*)((char*)ws + tp.big_bytes);
            void* iws = (char*)ws + tp.big_bytes + tp.small_bytes;
            // the big windows overlap (halos): a bias gradient summed over the big side would count
            // them twice -- left to the caller's channel sums (bias_done stays false)
            float* dbt = bias_side == 1 ? db : nullptr;
            bool done_all = dbt != nullptr;
            for (int n0 = 0; n0 < g.N; n0 += tp.nb) {
                const int nf = g.N - n0 < tp.nb ? g.N - n0 : tp.nb;
                const BnGeom gp = tile_geom(g, tp, nf);
                const BnFastPlan inner = tile_inner(2, tp, gp);
                const size_t ob = (size_t)n0 * g.Cb * g.Hb * g.Wb, os = (size_t)n0 * g.Cs * g.Hs * g.Ws;
                int rc = bn_launch_tile_gather(big + ob, bigp, nf, g.Cb, g.Hb, g.Wb, tp.bh, tp.bw, 0, st);
                if (rc) return rc;
                rc = bn_launch_tile_gather(small + os, smallp, nf, g.Cs, g.Hs, g.Ws, tp.sh, tp.sw, 1, st);
                if (rc) return rc;
                const int acc = (accumulate || n0 > 0) ? 1 : 0;
                bool done = false;
                rc = tp.edge ? bn_launch_edge_wgrad(inner, smallp, bigp, dw, gp, acc, iws, st, dbt, bias_side, &done)
                             : bn_launch_wgrad_fast(inner, smallp, bigp, dw, gp, acc, iws, st, dbt, bias_side, &done);
                if (rc) return rc;
                if (dbt && !done) {
                    // (a family that leaves the bias to the caller does so for every pass)
                    if (n0 > 0) return BN_E_BADARG;
                    dbt = nullptr;
                    done_all = false;
                }
            }
            if (bias_done && done_all) *bias_done = true;
            return 0;
        }
    }
    if (!generic && !plan.supported && bn_col_ok(g)) {
        BnProfScope prof(family, g.Cb, g.Cs, "k_im2col + k_gemm_mfma (dW)", st);
        return bn_launch_col_wgrad(small, big, dw, g, accumulate, ws, ws_bytes, st, db, bias_side, bias_done);
    }
    BnProfScope prof(family, g.Cb, g.Cs, plan.supported ? plan.kernel_name : "k_wgrad_generic",
                     st);
    if (plan.supported) {
        if (!ws_ok(plan, ws, ws_bytes)) return BN_E_WORKSPACE;
        return bn_launch_wgrad_fast(plan, small, big, dw, g, accumulate, ws, st, db, bias_side,
                                    bias_done);
    }
    return bn_launch_wgrad_generic(small, big, dw, g, accumulate, st);
}

// scratch of one role on geometry g (without the bias gradient's)
static size_t role_ws_need(int role, const BnGeom& g) {
    BnGeom g5;
    if (taps_plan(role, g, &g5)) return taps_bytes(g) + (role == 2 ? 4096 : 0) + role_ws_need(role, g5);
    if (role == 0 && (bn_s1c1_ok(g) || bn_s1in1_ok(g))) return 0;
    BigK bk;
    if (bigk_plan(role, g, &g5, &bk))
        return bigk_map_bytes(g) + bigk_w_bytes(g) + (role == 2 ? 4096 : 0) + role_ws_need(role == 1 ? 0 : role, g5);
    BigK1 b1;
    if (bigk1_plan(role, g, &g5, &b1))
        return bigk1_map_bytes(g, b1) + bigk_w_bytes(g) + (role == 2 ? 4096 : 0) + bigk1_inner_ws(role, g, g5, b1);
    if (chan_plan(role, g, &g5)) return chan_bytes(g) + role_ws_need(role, g5);
    if (bn_qgemm_supported(g)) return bn_qgemm_ws_bytes(role, g);
    if (bn_s5win_supported(g)) return bn_s5win_ws_bytes(role, g);
    if (role == 2 && s5_wgrad_by_col(g)) return bn_col_ws_bytes(g);
    BnFastPlan plan;
    if (role == 0) plan = bn_fast_down_plan(g);
    else if (role == 1) plan = bn_fast_up_plan(g);
    else {
        plan = bn_edge_wgrad_plan(g);
        if (!plan.supported) plan = bn_fast_wgrad_plan(g);
    }
    if (plan.supported) return plan.ws_bytes;
    if (role == 1) {
        BnGeom gf;
        BnFastPlan in;
        if (flip_plan(g, &gf, &in)) return flip_bytes(g) + in.ws_bytes;
        if (bigk1_flip_plan(g, &gf)) return flip_bytes(g) + role_ws_need(0, gf);
    }
    if (role == 0 && bn_s5_down_small_ok(g)) return bn_s5_down_small_ws_bytes(g);
    const PadPlan pp = pad_plan(role, g);
    if (pp.ok) return pad_ws_bytes(pp);
    const TilePlan tp = tile_plan(role, g);
    if (tp.ok) return tile_ws_bytes(tp);
    if (bn_col_ok(g)) return bn_col_ws_bytes(g);
    return 0;
}

extern "C" size_t bn_conv_ws_bytes(int op, int N, int C, int H, int W, int K, int R, int S,
                                   int stride, int off_t, int off_l, int P, int Q) {
    BnGeom g;
    switch (op) {
        case BN_OP_CONV_FWD: case BN_OP_CONV_BWD_D: case BN_OP_CONV_BWD_W:
            g = conv_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); break;
        case BN_OP_CONVT_FWD: case BN_OP_CONVT_BWD_D: case BN_OP_CONVT_BWD_W:
            g = convT_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); break;
        default: return 0;
    }
    if (!bn_geom_ok(g)) return 0;
    // the bias gradient (channel sums of dy) reuses the scratch after the weight gradient
    size_t bias_ws = 0;
    if (op == BN_OP_CONV_BWD_W) bias_ws = bn_channel_sum_ws_bytes(N, K, P * Q);
    if (op == BN_OP_CONVT_BWD_W) bias_ws = bn_channel_sum_ws_bytes(N, K, P * Q);
    if (force_generic()) return bias_ws;
    const int role = (op == BN_OP_CONV_FWD || op == BN_OP_CONVT_BWD_D) ? 0 :
                     (op == BN_OP_CONV_BWD_D || op == BN_OP_CONVT_FWD) ? 1 : 2;
    const size_t need = role_ws_need(role, g) + (role == 2 ? bias_ws : 0);
    return need > bias_ws ? need : bias_ws;
}

static bool taps_op_geom(int op, int N, int C, int H, int W, int K, int R, int S, int stride, int off_t,
                         int off_l, int P, int Q, BnGeom* g, int* role) {
    switch (op) {
        case BN_OP_CONV_FWD: *g = conv_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); *role = 0; break;
        case BN_OP_CONV_BWD_D: *g = conv_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); *role = 1; break;
        case BN_OP_CONVT_FWD: *g = convT_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); *role = 1; break;
        case BN_OP_CONVT_BWD_D: *g = convT_geom(N, C, H, W, K, R, S, stride, off_t, off_l, P, Q); *role = 0; break;
        default: return false;
    }
    return bn_geom_ok(*g);
}

// Bytes of the 5x5 copy of a small-kernel layer's weights if the forward / data-gradient op pads them (0: it does not)
extern "C" size_t bn_conv_taps_bytes(int op, int N, int C, int H, int W, int K, int R, int S, int stride,
                                     int off_t, int off_l, int P, int Q) {
    BnGeom g, g5;
    int role = 0;
    if (!taps_op_geom(op, N, C, H, W, K, R, S, stride, off_t, off_l, P, Q, &g, &role)) return 0;
    return taps_plan(role, g, &g5) ? taps_bytes(g) : 0;
}

// w5[j] <- the 5x5 copy of w[j] for n layers in one launch; geoms = n x (op, N, C, H, W, K, R, S, stride, off_t,
// off_l, P, Q) as bn_conv_taps_bytes takes them.  A layer whose op does not pad is an error (BN_E_SHAPE).
extern "C" int bn_conv_taps_pad(int n, const float* const* w, float* const* w5, const int* geoms,
                                bn_stream_t stream) {
    if (n < 0 || (n && (!w || !w5 || !geoms))) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    for (int j0 = 0; j0 < n; j0 += BN_PAD_TAPS_MAX_JOBS) {
        BnPadTapsJobs p;
        p.n = 0;
        for (int j = j0; j < n && j < j0 + BN_PAD_TAPS_MAX_JOBS; ++j) {
            const int* q = geoms + 13 * j;
            BnGeom g, g5;
            int role = 0;
            if (!w[j] || !w5[j]) return BN_E_BADARG;
            if (!taps_op_geom(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11], q[12], &g, &role))
                return BN_E_BADARG;
            if (!taps_plan(role, g, &g5)) return BN_E_SHAPE;
            BnPadTapsJob& jb = p.job[p.n++];
            jb.w = w[j]; jb.w5 = w5[j]; jb.pairs = (unsigned)((size_t)g.Cs * g.Cb);
            jb.R = g.R; jb.S = g.S; jb.dr = taps_dr(g); jb.ds = taps_ds(g); jb.blocks = 0;
        }
        const int rc = bn_launch_pad_taps_jobs(&p, st);
        if (rc) return rc;
    }
    return 0;
}

// One-shot: the next forward / data-gradient call of this thread on weights `w` reads their 5x5 copy from `w5`
// (written by bn_conv_taps_pad for the same layer) instead of padding them again; any other call drops the hint.
extern "C" int bn_conv_taps_hint(const float* w, const float* w5) {
    g_taps_hint_w = w5 ? w : nullptr;
    g_taps_hint_w5 = w ? w5 : nullptr;
    return 0;
}

extern "C" int bn_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N,
                             int C, int H, int W, int K, int R, int S, int stride, int pad_t,
                             int pad_l, int P, int Q, int act, float slope, void* ws,
                             size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !w || !y) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    return run_down(BN_PROF_CONV_FWD, x, w, b, y, nullptr, g, act, BN_ACT_NONE, slope, ws,
                    ws_bytes, (hipStream_t)stream);
}

// Conv2d + 2x2 / stride-2 max pooling + activation in one kernel (round 6): y:(N,K,P/2,Q/2), idx int32 = h Q + w of
// the window's winner in the (P, Q) plane of the convolution's output, which is never written.  Served for the first
// layer of a max-pooling architecture (stride 1, 5x5 taps, one or two input channels, 16 k output channels, even
// maps); BN_E_SHAPE otherwise: convolve, then bn_maxpool2d_act_fwd.
extern "C" int bn_conv2d_pool2_act_fwd(const float* x, const float* w, const float* b, float* y, int* idx, int N,
                                       int C, int H, int W, int K, int R, int S, int stride, int pad_t, int pad_l,
                                       int P, int Q, int act, float slope, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !w || !y || !idx) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    if (force_generic() || !aligned16_all(x, w, y, idx)) return BN_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (bn_s1in1_pool_ok(g)) {
        BnProfScope prof(BN_PROF_CONV_FWD, g.Cb, g.Cs, "k_down_s1_in1m<pool>", st);
        return bn_launch_s1in1_pool(x, w, b, y, idx, g, act, slope, st);
    }
    if (bn_down2_pool_ok(g)) {
        BnProfScope prof(BN_PROF_CONV_FWD, g.Cb, g.Cs, "k_down2_mfma<1, 2, 5, 0, 1, pool>", st);
        return bn_launch_down2_pool(x, w, b, y, idx, g, act, slope, st);
    }
    return BN_E_SHAPE;
}

// 1 if bn_conv2d_pool2_act_fwd serves this layer (aligned operands assumed), else 0
extern "C" int bn_conv2d_pool2_act_ok(int N, int C, int H, int W, int K, int R, int S, int stride, int pad_t,
                                      int pad_l, int P, int Q) {
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g) || force_generic()) return 0;
    return (bn_s1in1_pool_ok(g) || bn_down2_pool_ok(g)) ? 1 : 0;
}

// Weight (+ bias) gradient of a layer run by bn_conv2d_pool2_act_fwd straight from the POOLED gradient: dy, y, idx are
// the (N,K,P/2,Q/2) tensors of the pooling's output side; the dense gradient of the convolution's output (3/4 zeros)
// is never built.  Served for the first layer (1 or 2 input channels; 16 / 32 / 64 output channels); the scratch
// query returns 0 where it is not (the caller then runs bn_maxpool2d_act_bwd and bn_conv2d_bwd_weight).
extern "C" size_t bn_conv2d_pool2_bwd_weight_ws_bytes(int N, int C, int H, int W, int K, int R, int S, int stride,
                                                      int pad_t, int pad_l, int P, int Q) {
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g) || force_generic() || !bn_wgrad_pool_c1_ok(g)) return 0;
    return bn_wgrad_pool_c1_ws_bytes(g);
}
extern "C" int bn_conv2d_pool2_bwd_weight(const float* x, const float* dy, const float* y, const int* idx, float* dw,
                                          float* db, int N, int C, int H, int W, int K, int R, int S, int stride,
                                          int pad_t, int pad_l, int P, int Q, int act, float slope, int accumulate,
                                          void* ws, size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !dy || !y || !idx || !dw) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    if (force_generic() || !bn_wgrad_pool_c1_ok(g) || (act != BN_ACT_NONE && act != BN_ACT_LRELU)) return BN_E_SHAPE;
    if (!ws || ws_bytes < bn_wgrad_pool_c1_ws_bytes(g)) return BN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    BnProfScope prof(BN_PROF_CONV_BWD_W, g.Cb, g.Cs, "k_wgrad_pool_c1", st);
    return bn_launch_wgrad_pool_c1(x, dy, y, idx, dw, db, g, act, slope, accumulate, ws, st);
}

static bool u8_fast(const BnGeom& g, int act) {
    return !force_generic() && g.Cb == 1 && bn_edge_down_plan(g).supported && bn_edge_down_plan(g).variant != 9 &&
           (act == BN_ACT_NONE || act == BN_ACT_LRELU);
}

extern "C" size_t bn_conv2d_fwd_u8_ws_bytes(int N, int C, int H, int W, int K, int R, int S,
                                            int stride, int pad_t, int pad_l, int P, int Q,
                                            int act) {
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return 0;
    if (u8_fast(g, act)) return 0;      // the SAME predicate the launch dispatches on
    size_t conv = bn_conv_ws_bytes(BN_OP_CONV_FWD, N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    conv = (conv + 255) & ~(size_t)255;
    return conv + (size_t)N * C * H * W * sizeof(float);
}

extern "C" int bn_conv2d_fwd_u8(const unsigned char* x, const float* w, const float* b, float* y,
                                int N, int C, int H, int W, int K, int R, int S, int stride,
                                int pad_t, int pad_l, int P, int Q, int act, float slope, void* ws,
                                size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !w || !y) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (u8_fast(g, act)) {
        BnProfScope prof(BN_PROF_CONV_FWD, g.Cb, g.Cs, bn_edge_down_kernel_name(g, act, false, true), st);
        return bn_launch_edge_down(nullptr, w, b, y, nullptr, g, act, BN_ACT_NONE, slope, st, x);
    }
    const size_t need =
        bn_conv2d_fwd_u8_ws_bytes(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q, act);
    const size_t n_in = (size_t)N * C * H * W;
    if (!ws || ws_bytes < need || need < n_in * sizeof(float)) return BN_E_WORKSPACE;
    const size_t conv_ws = need - n_in * sizeof(float);
    float* xf = (float*)((char*)ws + conv_ws);
    int rc = bn_launch_u8_to_unit_float(x, xf, n_in, st);
    if (rc) return rc;
    return run_down(BN_PROF_CONV_FWD, xf, w, b, y, nullptr, g, act, BN_ACT_NONE, slope, ws, conv_ws,
                    st);
}

extern "C" int bn_conv2d_bwd_data(const float* dy, const float* w, float* dx,
                                  const float* dact_src, int N, int C, int H, int W, int K, int R,
                                  int S, int stride, int pad_t, int pad_l, int P, int Q, int dact,
                                  float slope, void* ws, size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!dy || !w || !dx) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    return run_up(BN_PROF_CONV_BWD_D, dy, w, nullptr, dx, dact_src, g, BN_ACT_NONE, dact, slope,
                  ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int bn_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N,
                                    int C, int H, int W, int K, int R, int S, int stride,
                                    int pad_t, int pad_l, int P, int Q, int accumulate, void* ws,
                                    size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !dy || !dw) return BN_E_BADARG;
    const BnGeom g = conv_geom(N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    bool bias_done = false;   // the MFMA kernel sums dy (its `small` operand) on the way
    int rc = run_wgrad(BN_PROF_CONV_BWD_W, dy, x, dw, g, accumulate, ws, ws_bytes, st, db, 1,
                       &bias_done);
    if (rc) return rc;
    if (db && !bias_done)
        rc = bn_launch_channel_sum(dy, db, N, K, P * Q, accumulate, ws, ws_bytes, st);
    return rc;
}

extern "C" int bn_convT2d_fwd(const float* x, const float* w, const float* b, float* y, int N,
                              int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                              int crop_t, int crop_l, int Ho, int Wo, int act, float slope,
                              void* ws, size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !w || !y) return BN_E_BADARG;
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    return run_up(BN_PROF_CONVT_FWD, x, w, b, y, nullptr, g, act, BN_ACT_NONE, slope, ws,
                  ws_bytes, (hipStream_t)stream);
}

extern "C" int bn_convT2d_bwd_data(const float* dy, const float* w, float* dx,
                                   const float* dact_src, int N, int Ci, int Hi, int Wi, int Co,
                                   int R, int S, int stride, int crop_t, int crop_l, int Ho,
                                   int Wo, int dact, float slope, void* ws, size_t ws_bytes,
                                   bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!dy || !w || !dx) return BN_E_BADARG;
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    return run_down(BN_PROF_CONVT_BWD_D, dy, w, nullptr, dx, dact_src, g, BN_ACT_NONE, dact, slope,
                    ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int bn_convT2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N,
                                     int Ci, int Hi, int Wi, int Co, int R, int S, int stride,
                                     int crop_t, int crop_l, int Ho, int Wo, int accumulate,
                                     void* ws, size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !dy || !dw) return BN_E_BADARG;
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    bool bias_done = false;   // here dy is the kernel's `big` operand
    int rc = run_wgrad(BN_PROF_CONVT_BWD_W, x, dy, dw, g, accumulate, ws, ws_bytes, st, db, 2,
                       &bias_done);
    if (rc) return rc;
    if (db && !bias_done)
        rc = bn_launch_channel_sum(dy, db, N, Co, Ho * Wo, accumulate, ws, ws_bytes, st);
    return rc;
}

extern "C" int bn_act_fwd(const float* x, float* y, size_t n, int act, float slope,
                          bn_stream_t stream) {
    if (!x || !y || n == 0) return BN_E_BADARG;
    return bn_launch_act_fwd(x, y, n, act, slope, (hipStream_t)stream);
}

extern "C" int bn_act_bwd(const float* dy, const float* y, float* dpre, size_t n, int act,
                          float slope, bn_stream_t stream) {
    if (!dy || !y || !dpre) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_act_bwd(dy, y, dpre, n, act, slope, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// batch norm
// ------------------------------------------------------------------------------------------
extern "C" size_t bn_batchnorm_ws_bytes(int N, int C) {
    return (N > 0 && C > 0) ? bn_batchnorm_ws_bytes_impl(N, C) : 0;
}

extern "C" int bn_batchnorm_stats(const float* x, float* mean, float* var, int N, int C, int HW,
                                  void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !mean || !var || N <= 0 || C <= 0 || HW <= 0) return BN_E_BADARG;
    if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(N, C)) return BN_E_WORKSPACE;
    return bn_launch_bn_stats(x, mean, var, N, C, HW, ws, (hipStream_t)stream);
}

extern "C" int bn_batchnorm_finalize(const float* mean, const float* var, float* invstd,
                                     float* running_mean, float* running_var, int C, float eps,
                                     float momentum, float unbias, bn_stream_t stream) {
    if (!mean || !var || !invstd || C <= 0) return BN_E_BADARG;
    return bn_launch_bn_finalize(mean, var, invstd, running_mean, running_var, C, eps, momentum,
                                 unbias, (hipStream_t)stream);
}

extern "C" int bn_batchnorm_act_fwd(const float* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, float* y, int N, int C,
                                    int HW, int act, float slope, bn_stream_t stream) {
    if (!x || !mean || !invstd || !y || N <= 0 || C <= 0 || HW <= 0) return BN_E_BADARG;
    return bn_launch_bn_act_fwd(x, mean, invstd, gamma, beta, y, N, C, HW, act, slope,
                                (hipStream_t)stream);
}

extern "C" int bn_batchnorm_act_bwd(const float* x, const float* y, const float* dy,
                                    const float* mean, const float* invstd, const float* gamma,
                                    float* dx, float* dgamma, float* dbeta, int accumulate,
                                    int batch_stats, int N, int C, int HW, int act, float slope,
                                    void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !y || !dy || !mean || !invstd || !dx || N <= 0 || C <= 0 || HW <= 0)
        return BN_E_BADARG;
    if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(N, C)) return BN_E_WORKSPACE;
    return bn_launch_bn_act_bwd(x, y, dy, mean, invstd, gamma, dx, dgamma, dbeta, accumulate,
                                batch_stats, N, C, HW, act, slope, ws, (hipStream_t)stream);
}

// Train-mode BatchNorm2d + activation over a batch whose statistics are taken PER CHUNK of frames
// (rows [bounds[2i], bounds[2i+1]) of x, in order; the running estimates see one update per chunk
// with factors[i]): one call per layer instead of three per chunk.  mean / invstd: [n_chunks][C].
extern "C" int bn_batchnorm_train_fwd_chunks(const float* x, const float* gamma, const float* beta,
                                             float* running_mean, float* running_var,
                                             long long* num_batches_tracked, float* y,
                                             float* mean, float* invstd, const int* bounds,
                                             const float* factors, int n_chunks, int C, int HW,
                                             float eps, int act, float slope, void* ws,
                                             size_t ws_bytes, bn_stream_t stream) {
    if (!x || !y || !mean || !invstd || !bounds || n_chunks <= 0 || C <= 0 || HW <= 0) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    {   // all chunks in the launches of one (up to four contiguous chunks)
        int nmax = 0;
        for (int i = 0; i < n_chunks; ++i) nmax = bounds[2 * i + 1] - bounds[2 * i] > nmax ? bounds[2 * i + 1] - bounds[2 * i] : nmax;
        if (nmax > 0 && ws && ws_bytes >= bn_batchnorm_ws_bytes_impl(nmax, C)) {
            const int rc = bn_launch_bn_train_fwd_chunks(x, gamma, beta, running_mean, running_var,
                                                         num_batches_tracked, y, mean, invstd, bounds,
                                                         factors, n_chunks, C, HW, eps, act, slope, ws, st);
            if (rc != BN_E_SHAPE) return rc;
        }
    }
    // (more than four chunks, or chunks that are not contiguous: one at a time)
    for (int i = 0; i < n_chunks; ++i) {
        const int bi[2] = {0, bounds[2 * i + 1] - bounds[2 * i]};
        if (bi[1] <= 0 || bounds[2 * i] < 0) return BN_E_BADARG;
        if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(bi[1], C)) return BN_E_WORKSPACE;
        const size_t o = (size_t)bounds[2 * i] * C * HW;
        const float fi = factors ? factors[i] : 0.f;
        const int rc = bn_launch_bn_train_fwd_chunks(x + o, gamma, beta, running_mean, running_var,
                                                     num_batches_tracked, y + o, mean + (size_t)i * C,
                                                     invstd + (size_t)i * C, bi, &fi, 1, C, HW, eps, act,
                                                     slope, ws, st);
        if (rc) return rc;
    }
    return 0;
}

// the backward pass of the same: dgamma / dbeta are accumulated over the chunks (accumulate = 0:
// they are overwritten by the first chunk)
// y == NULL (identity / LeakyReLU): the sign of the activation's input is rebuilt from x with the
// forward pass's affine map through (gamma, beta) -- the saved output is not read back
extern "C" int bn_batchnorm_act_bwd_chunks(const float* x, const float* y, const float* dy,
                                           const float* mean, const float* invstd,
                                           const float* gamma, const float* beta, float* dx,
                                           float* dgamma, float* dbeta, int accumulate,
                                           const int* bounds, int n_chunks, int C, int HW, int act,
                                           float slope, void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !dx || !bounds || n_chunks <= 0 || C <= 0 || HW <= 0)
        return BN_E_BADARG;
    if (!y && act != BN_ACT_NONE && act != BN_ACT_LRELU) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    {
        int nmax = 0;
        for (int i = 0; i < n_chunks; ++i) nmax = bounds[2 * i + 1] - bounds[2 * i] > nmax ? bounds[2 * i + 1] - bounds[2 * i] : nmax;
        if (nmax > 0 && ws && ws_bytes >= bn_batchnorm_ws_bytes_impl(nmax, C)) {
            const int rc = bn_launch_bn_act_bwd_chunks(x, y, dy, mean, invstd, gamma, beta, dx, dgamma, dbeta,
                                                       accumulate, bounds, n_chunks, C, HW, act, slope, ws, st);
            if (rc != BN_E_SHAPE) return rc;
        }
    }
    for (int i = 0; i < n_chunks; ++i) {
        const int bi[2] = {0, bounds[2 * i + 1] - bounds[2 * i]};
        if (bi[1] <= 0 || bounds[2 * i] < 0) return BN_E_BADARG;
        if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(bi[1], C)) return BN_E_WORKSPACE;
        const size_t o = (size_t)bounds[2 * i] * C * HW;
        const int rc = bn_launch_bn_act_bwd_chunks(x + o, y ? y + o : nullptr, dy + o, mean + (size_t)i * C,
                                                   invstd + (size_t)i * C, gamma, beta, dx + o, dgamma, dbeta,
                                                   (accumulate || i > 0) ? 1 : 0, bi, 1, C, HW, act, slope, ws,
                                                   st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int bn_batchnorm_moment(const float* x, const float* center, float* sums, int N, int C,
                                   int HW, void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !sums || N <= 0 || C <= 0 || HW <= 0) return BN_E_BADARG;
    if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(N, C)) return BN_E_WORKSPACE;
    return bn_launch_bn_moment(x, center, sums, N, C, HW, ws, (hipStream_t)stream);
}

extern "C" int bn_batchnorm_bwd_reduce(const float* x, const float* y, const float* dy,
                                       const float* mean, const float* invstd, float* sum_dz,
                                       float* sum_dzx, int N, int C, int HW, int act, float slope,
                                       void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !y || !dy || !mean || !invstd || !sum_dz || !sum_dzx || N <= 0 || C <= 0 || HW <= 0)
        return BN_E_BADARG;
    if (!ws || ws_bytes < bn_batchnorm_ws_bytes_impl(N, C)) return BN_E_WORKSPACE;
    return bn_launch_bn_bwd_reduce(x, y, dy, mean, invstd, sum_dz, sum_dzx, N, C, HW, act, slope, ws,
                                   (hipStream_t)stream);
}

extern "C" int bn_batchnorm_bwd_apply(const float* x, const float* y, const float* dy,
                                      const float* mean, const float* invstd, const float* gamma,
                                      const float* sum_dz, const float* sum_dzx, float* dx, int N,
                                      int C, int HW, float inv_count, int act, float slope,
                                      bn_stream_t stream) {
    if (!x || !y || !dy || !mean || !invstd || !sum_dz || !sum_dzx || !dx || N <= 0 || C <= 0 ||
        HW <= 0)
        return BN_E_BADARG;
    return bn_launch_bn_bwd_apply(x, y, dy, mean, invstd, gamma, sum_dz, sum_dzx, dx, N, C, HW,
                                  inv_count, act, slope, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// linear
// ------------------------------------------------------------------------------------------
extern "C" size_t bn_linear_ws_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    // forward (reduce over K) and data gradient (reduce over N) may split their reduction
    const size_t f = bn_gemm_ws_bytes(M, N, K), d = bn_gemm_ws_bytes(M, K, N);
    return f > d ? f : d;
}

extern "C" int bn_linear_fwd(const float* x, const float* w, const float* b, float* y, int M,
                             int K, int N, void* ws, size_t ws_bytes, bn_stream_t stream) {
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0) return BN_E_BADARG;
    GemmArgs a;
    a.A = x; a.sai = K; a.sak = 1;
    a.B = w; a.sbk = 1; a.sbj = K;
    a.C = y; a.sci = N; a.scj = 1;
    a.M = M; a.N = N; a.K = K;
    a.bias_j = b; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = 0.f; a.accumulate = 0;
    BnProfScope prof(BN_PROF_LINEAR_FWD, K, N, "k_linear_jobs", (hipStream_t)stream);
    return bn_launch_linear_jobs(&a, nullptr, nullptr, nullptr, 0, 0, 0, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int bn_linear_bwd(const float* x, const float* w, const float* dy, float* dx,
                             const float* dact_src, int dact, float slope, float* dw, float* db,
                             int accumulate, int M, int K, int N, void* ws, size_t ws_bytes,
                             bn_stream_t stream) {
    if (!dy || M <= 0 || K <= 0 || N <= 0) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // one launch: data gradient, weight gradient and bias column sums side by side
    BnProfScope prof(BN_PROF_LINEAR_BWD, K, N, "k_linear_jobs (dx, dW, db)", st);
    GemmArgs gx, gw;
    if (dx) {
        if (!w) return BN_E_BADARG;
        GemmArgs& a = gx;                 // dx[m,k] = sum_n dy[m,n] w[n,k]
        a.A = dy; a.sai = N; a.sak = 1;
        a.B = w; a.sbk = K; a.sbj = 1;
        a.C = dx; a.sci = K; a.scj = 1;
        a.M = M; a.N = K; a.K = N;
        a.bias_j = nullptr; a.dact_src = dact_src; a.dact = dact; a.slope = slope;
        a.accumulate = 0; a.part = nullptr; a.kslice = 0;
    }
    if (dw) {
        if (!x) return BN_E_BADARG;
        GemmArgs& a = gw;                 // dw[n,k] = sum_m dy[m,n] x[m,k]
        a.A = dy; a.sai = 1; a.sak = N;
        a.B = x; a.sbk = K; a.sbj = 1;
        a.C = dw; a.sci = K; a.scj = 1;
        a.M = N; a.N = K; a.K = M;
        a.bias_j = nullptr; a.dact_src = nullptr; a.dact = BN_ACT_NONE; a.slope = 0.f;
        a.accumulate = accumulate; a.part = nullptr; a.kslice = 0;
    }
    return bn_launch_linear_jobs(dx ? &gx : nullptr, dw ? &gw : nullptr, dy, db, M, N, accumulate, ws,
                                 ws_bytes, st);
}

// ------------------------------------------------------------------------------------------
// losses / variational tail / optimiser / input conversion
// ------------------------------------------------------------------------------------------
extern "C" int bn_sqerr_frame_sums(const float* pred, const float* target, const float* mask,
                                   float* frame_sums, int N, size_t D, bn_stream_t stream) {
    if (!pred || !target || !frame_sums || N <= 0 || D == 0) return BN_E_BADARG;
    return bn_launch_sqerr_frame_sums(pred, target, mask, frame_sums, N, D, (hipStream_t)stream);
}

extern "C" int bn_sqerr_bwd(const float* pred, const float* target, const float* mask,
                            float* dpred, size_t n, float scale, const float* gscale,
                            bn_stream_t stream) {
    if (!pred || !target || !dpred) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_sqerr_bwd(pred, target, mask, dpred, n, scale, gscale, (hipStream_t)stream);
}

// Last decoder layer + pixel loss.  Fast path: the VALU edge kernel with the loss epilogue; any
// other geometry: the same result composed from the stand-alone kernels (xhat goes through `ws`
// if the caller does not want it).
static bool fused_sqerr_fast(const BnGeom& g) {
    return !force_generic() && !bn_qgemm_supported(g) && !bn_s5win_supported(g) && !bn_s5_up_plan(g).supported &&
           bn_edge_up_plan(g).supported;
}

extern "C" int bn_convT2d_fwd_sqerr_parts(int N, int Ci, int Hi, int Wi, int Co, int R, int S,
                                          int stride, int crop_t, int crop_l, int Ho, int Wo) {
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g)) return 0;
    return fused_sqerr_fast(g) ? bn_edge_up_parts_per_frame(g) : 1;
}

extern "C" size_t bn_convT2d_fwd_sqerr_ws_bytes(int N, int Ci, int Hi, int Wi, int Co, int R,
                                                int S, int stride, int crop_t, int crop_l, int Ho,
                                                int Wo, int with_xhat) {
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g) || fused_sqerr_fast(g)) return 0;
    size_t conv = bn_conv_ws_bytes(BN_OP_CONVT_FWD, N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l,
                                   Ho, Wo);
    conv = (conv + 255) & ~(size_t)255;
    return conv + (with_xhat ? 0 : (size_t)N * Co * Ho * Wo * sizeof(float));
}

extern "C" int bn_convT2d_fwd_sqerr(const float* x, const float* w, const float* b,
                                    const float* target, const float* mask, float* xhat,
                                    float* dpre, float* part, int N, int Ci, int Hi, int Wi,
                                    int Co, int R, int S, int stride, int crop_t, int crop_l,
                                    int Ho, int Wo, int act, float slope, void* ws,
                                    size_t ws_bytes, bn_stream_t stream) {
    TapsHintDrop hint_drop;
    if (!x || !w || !target || !dpre || !part) return BN_E_BADARG;
    const BnGeom g = convT_geom(N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo);
    if (!bn_geom_ok(g)) return BN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (fused_sqerr_fast(g)) {
        BnProfScope prof(BN_PROF_CONVT_FWD, g.Cs, g.Cb, bn_edge_up_kernel_name(g, true), st);
        return bn_launch_edge_up(x, w, b, xhat, g, act, slope, st, target, mask, dpre, part);
    }
    const size_t need = bn_convT2d_fwd_sqerr_ws_bytes(N, Ci, Hi, Wi, Co, R, S, stride, crop_t,
                                                      crop_l, Ho, Wo, xhat != nullptr);
    if (need && (!ws || ws_bytes < need)) return BN_E_WORKSPACE;
    const size_t n_out = (size_t)N * Co * Ho * Wo;
    const size_t conv_ws = need - (xhat ? 0 : n_out * sizeof(float));
    float* xh = xhat ? xhat : (float*)((char*)ws + conv_ws);
    int rc = run_up(BN_PROF_CONVT_FWD, x, w, b, xh, nullptr, g, act, BN_ACT_NONE, slope, ws,
                    conv_ws, st);
    if (rc) return rc;
    rc = bn_launch_sqerr_frame_sums(xh, target, mask, part, N, (size_t)Co * Ho * Wo, st);
    if (rc) return rc;
    rc = bn_launch_sqerr_bwd(xh, target, mask, dpre, n_out, 1.f, nullptr, st);
    if (rc) return rc;
    if (act != BN_ACT_NONE) rc = bn_launch_act_bwd(dpre, xh, dpre, n_out, act, slope, st);
    return rc;
}

extern "C" int bn_scale_frames(float* t, const float* frame_scale, const float* group_scale,
                               const int* group_of_frame, int N, size_t D, bn_stream_t stream) {
    if (!t || !frame_scale || N <= 0 || D == 0) return BN_E_BADARG;
    if ((group_scale == nullptr) != (group_of_frame == nullptr)) return BN_E_BADARG;
    return bn_launch_scale_frames(t, frame_scale, group_scale, group_of_frame, N, D,
                                  (hipStream_t)stream);
}

extern "C" int bn_reduce_sum(const float* in, float* out, size_t n, float scale,
                             bn_stream_t stream) {
    if (!in || !out) return BN_E_BADARG;
    return bn_launch_reduce_sum(in, out, n, scale, (hipStream_t)stream);
}

extern "C" int bn_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z,
                              size_t n, bn_stream_t stream) {
    if (!mu || !logvar || !eps || !z) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_reparam_fwd(mu, logvar, eps, z, n, (hipStream_t)stream);
}

extern "C" int bn_kl_rows(const float* mu, const float* logvar, float* kl_rows, int N, int D,
                          bn_stream_t stream) {
    if (!mu || !logvar || !kl_rows || N <= 0 || D <= 0) return BN_E_BADARG;
    return bn_launch_kl_rows(mu, logvar, kl_rows, N, D, (hipStream_t)stream);
}

extern "C" int bn_reparam_bwd(const float* dz, const float* z, const float* mu, float* dlogvar,
                              size_t n, bn_stream_t stream) {
    if (!dz || !z || !mu || !dlogvar) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_reparam_bwd(dz, z, mu, dlogvar, n, (hipStream_t)stream);
}

extern "C" int bn_kl_bwd(const float* mu, const float* logvar, float* dmu, float* dlogvar,
                         size_t n, float scale, const float* gscale, bn_stream_t stream) {
    if (!mu || !logvar || !dmu || !dlogvar) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_kl_bwd(mu, logvar, dmu, dlogvar, n, scale, gscale, (hipStream_t)stream);
}

extern "C" int bn_decomposed_kl_fwd(const float* z, const float* mu, const float* logvar,
                                    float* out3, float* log_qz, float* lse, float* terms, int N,
                                    int D, bn_stream_t stream) {
    if (!z || !mu || !logvar || !out3 || !log_qz || !lse || !terms || N <= 0 || D <= 0)
        return BN_E_BADARG;
    return bn_launch_dkl_fwd(z, mu, logvar, out3, log_qz, lse, terms, N, D, (hipStream_t)stream);
}

extern "C" int bn_decomposed_kl_bwd(const float* z, const float* mu, const float* logvar,
                                    const float* log_qz, const float* lse, const float* g3,
                                    float* dz, float* dmu, float* dlogvar, int N, int D,
                                    bn_stream_t stream) {
    if (!z || !mu || !logvar || !log_qz || !lse || !g3 || !dz || !dmu || !dlogvar || N <= 0 ||
        D <= 0)
        return BN_E_BADARG;
    return bn_launch_dkl_bwd(z, mu, logvar, log_qz, lse, g3, dz, dmu, dlogvar, N, D,
                             (hipStream_t)stream);
}

extern "C" int bn_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax,
                                    size_t n, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int step, bn_stream_t stream) {
    if (!p || !g || !m || !v || !vmax || step < 1) return BN_E_BADARG;
    if (n == 0) return 0;
    BnProfScope prof(BN_PROF_ADAM, 0, 0, "k_adam_amsgrad", (hipStream_t)stream);
    return bn_launch_adam(p, g, m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step,
                          (hipStream_t)stream);
}

extern "C" int bn_u8_to_unit_float(const unsigned char* in, float* out, size_t n,
                                   bn_stream_t stream) {
    if (!in || !out) return BN_E_BADARG;
    if (n == 0) return 0;
    return bn_launch_u8_to_unit_float(in, out, n, (hipStream_t)stream);
}
