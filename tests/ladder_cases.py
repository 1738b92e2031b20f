"""Geometries for the dispatch-ladder contract: the named conv cases of the device tests plus
the rungs nobody asks for by name -- odd widths, stride 3, kernels past 9, unaligned operand views -- and the
function that walks them through the three roles with the profiling hook armed."""
import torch

from behavenet_amd import _hip

SLOPE = 0.05

# (name, N, C, H, W, K, R, stride, (pt, pb), (pl, pr)) as tests/test_gpu_kernels.py CONV_CASES
LADDER_EXTRA = [
    ('odd_w_31x29', 2, 32, 31, 29, 64, 5, 2, (1, 2), (1, 2)),
    ('odd_w_E0_63x65', 2, 1, 63, 65, 32, 5, 2, (1, 2), (1, 2)),
    ('w_mod4_2_30x30', 2, 32, 60, 60, 64, 5, 2, (1, 2), (1, 2)),
    ('wide_90x90', 2, 32, 180, 180, 64, 5, 2, (1, 2), (1, 2)),
    ('stride3_k5', 2, 16, 30, 30, 32, 5, 3, (1, 1), (1, 1)),
    ('stride3_k3', 2, 16, 27, 27, 32, 3, 3, (0, 0), (0, 0)),
    ('stride4_k5', 2, 16, 32, 32, 32, 5, 4, (1, 2), (1, 2)),
    ('k11_s2', 2, 8, 32, 32, 16, 11, 2, (4, 5), (4, 5)),
    ('k11_s1', 2, 8, 24, 24, 16, 11, 1, (5, 5), (5, 5)),
    ('k7_odd_map_s2', 2, 16, 31, 27, 32, 7, 2, (2, 3), (2, 3)),
    ('k5_s1_odd_19x23', 2, 12, 19, 23, 20, 5, 1, (2, 2), (2, 2)),
    ('k3_s1_c3', 2, 3, 20, 20, 5, 3, 1, (1, 1), (1, 1)),
    ('k1_s1', 2, 32, 16, 16, 64, 1, 1, (0, 0), (0, 0)),
    ('s5_17x17', 2, 64, 17, 17, 96, 5, 5, (1, 2), (1, 2)),
    ('s1_16_to_1_k5', 2, 16, 64, 64, 1, 5, 1, (2, 2), (2, 2)),
    ('c1_to_64', 2, 1, 128, 128, 64, 5, 2, (1, 2), (1, 2)),
    ('c1_to_48', 2, 1, 128, 128, 48, 5, 2, (1, 2), (1, 2)),
]

DETOUR_TOKENS = ('im2col', 'col2im', 'generic')

# The ONLY geometries (of the named cases + the ones above) that may still reach the column-matrix or the
# shape-agnostic rungs of csrc/capi.hip's run_down / run_up / run_wgrad, role by role -- DESIGN.md section 8's list of
# detours as a checked contract.  Everything else must land on a specialised kernel (possibly on zero-padded /
# tiled / phase-split copies, which the kernel name says).
DETOURS = {
    # operands that are not 16-byte aligned: the first rung sends every role to the direct loops
    'E1 @unaligned': ('fwd', 'bwd_data', 'bwd_weight'),
    'E4 @unaligned': ('fwd', 'bwd_data', 'bwd_weight'),
    'pad_24x20 @unaligned': ('fwd', 'bwd_data', 'bwd_weight'),
    # kernels past 10 taps: direct loops (the architecture search draws <= 9)
    'k11_s1': ('fwd', 'bwd_data', 'bwd_weight'),
    'k11_s2': ('fwd', 'bwd_data', 'bwd_weight'),
    'k11s2': ('fwd', 'bwd_data', 'bwd_weight'),
    'k13s1_valid': ('fwd', 'bwd_data', 'bwd_weight'),
    # strides other than 1, 2 and the kernel size: column matrix + GEMM
    'stride3_k3': ('fwd', 'bwd_data', 'bwd_weight'),
    'stride3_k5': ('fwd', 'bwd_data', 'bwd_weight'),
    'stride4_k5': ('fwd', 'bwd_data', 'bwd_weight'),
    # 1x1 kernels, odd channel counts under 3x3 / 4x4 kernels, odd big maps under 7x7
    'k1_s1': ('fwd', 'bwd_data', 'bwd_weight'),
    'k3_s1_c3': ('fwd', 'bwd_data'),
    'k3s1': ('fwd', 'bwd_data'),
    'k4s2': ('fwd', 'bwd_data', 'bwd_weight'),
    'k7_odd_map_s2': ('fwd', 'bwd_data', 'bwd_weight'),
    'k7s2_valid': ('fwd', 'bwd_data', 'bwd_weight'),
    # a single channel against 48 (no multiple of 32) on the other side
    'c1_to_48': ('fwd', 'bwd_weight'),
    # stride-1 5x5 on odd maps / odd channel counts: gather-up and weight gradient
    'k5_s1_odd_19x23': ('bwd_data', 'bwd_weight'),
    'odd_channels': ('bwd_data',),
    # weight gradients of wide-kernel layers whose small side has few channels / odd widths
    'k7s2_same_24x20': ('bwd_weight',),
    's1_16_to_1_k5': ('bwd_weight',),
    's1_k7_2ch_50x70': ('bwd_weight',),
}


def _operands(case, misalign=False, dev='cuda'):
    name, N, C, H, W, K, R, st, (pt, pb), (pl, pr) = case
    g = torch.Generator().manual_seed(0)
    P = (H + pt + pb - R) // st + 1
    Q = (W + pl + pr - R) // st + 1
    geom = (N, C, H, W, K, R, R, st, pt, pl, P, Q)

    def make(shape):
        n = 1
        for s in shape:
            n *= s
        flat = torch.rand((n + 4,), generator=g).to(dev)
        # (misalign: a contiguous view that starts 4 bytes into the allocation)
        return flat[1:1 + n].view(shape) if misalign else flat[:n].view(shape)
    return make((N, C, H, W)), make((K, C, R, R)), make((K,)), make((N, K, P, Q)), geom


def kernel_names(case, misalign=False):
    """{role: the kernel name bn_prof_read reports} for the three roles of bn_conv2d_* on this geometry."""
    x, w, b, dy, geom = _operands(case, misalign)
    dw = torch.zeros(tuple(w.shape), device=x.device)
    db = torch.zeros(tuple(b.shape), device=x.device)
    out = {}
    for role, prof, fn in (
            ('fwd', _hip.PROF_CONV_FWD, lambda: _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE)),
            ('bwd_data', _hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy, w, geom, None, _hip.ACT_NONE, SLOPE)),
            ('bwd_weight', _hip.PROF_CONV_BWD_W, lambda: _hip.conv2d_bwd_weight(x, dy, dw, db, geom, False))):
        _hip.prof_select(prof, 0, 0)
        try:
            fn()
            torch.cuda.synchronize()
            _, n, name = _hip.prof_read()
        except _hip.HipLibraryError as err:
            # the library's own refusal (a negative code: nothing was launched) is part of the contract
            n, name = 1, 'refused: ' + str(err).split('failed: ')[-1].split(':')[0]
        finally:
            _hip.prof_select(_hip.PROF_NONE)
        out[role] = name if n >= 1 else 'no scope'
    return out


def all_cases():
    from tests.test_gpu_kernels import CONV_CASES
    return list(CONV_CASES) + list(LADDER_EXTRA)


def ladder_table():
    table = {}
    for case in all_cases():
        table[case[0]] = kernel_names(case)
    # unaligned operand views: the first rung of the ladder (capi.hip `aligned16_all`) sends them to the
    # shape-agnostic kernels
    for case in all_cases():
        if case[0] in ('E1', 'E4', 'pad_24x20'):
            table[case[0] + ' @unaligned'] = kernel_names(case, misalign=True)
    return table
