"""Host side of the convolution dispatch (csrc/capi.hip) WITHOUT a GPU: every geometry of a seeded sweep must
get past plan selection, scratch sizing and argument checks of all three roles -- i.e. reach a kernel launch,
which on a machine without a device fails with a (positive) HIP error code.  A negative code is the library's
own refusal (BN_E_BADARG / BN_E_SHAPE / BN_E_WORKSPACE): the class of bug where a plan picks a detour whose
copy kernel does not take the padded shape (found by this sweep in round 4: a zero-padded copy of a
gather-up operand with a width that is no multiple of 4).

The sweep is the one of tests/test_gpu_kernels.py::test_random_geometries_all_roles (which checks the numbers
on the device) with more draws."""
import ctypes

import numpy as np
import pytest
import torch

from behavenet_amd import _hip


def _cases(seed, count):
    """The seeded sweep plus every named conv case of the device tests."""
    from tests.test_gpu_kernels import _random_conv_cases, _random_stride5_cases, CONV_CASES
    # (round 5: plus a sweep of the stride-5 last layer on maps of 1..17 pixels -- csrc/conv_s5win.hip)
    return list(CONV_CASES) + _random_conv_cases(seed, count) + _random_conv_cases(seed + 1, count // 4, big=True) + \
        _random_stride5_cases(seed + 2, count // 4)


@pytest.mark.skipif(torch.cuda.is_available(), reason='launches real kernels on dummy pointers when a GPU is present')
def test_every_role_of_a_geometry_sweep_reaches_a_launch():
    lib = _hip.load()
    buf = (ctypes.c_char * 256)()
    p = ctypes.addressof(buf)
    big = ctypes.c_size_t(1 << 42)
    refused = []
    for name, N, C, H, W, K, R, st, (pt, pb), (pl, pr) in _cases(7, 600):
        P, Q = (H + pt + pb - R) // st + 1, (W + pl + pr - R) // st + 1
        conv = (N, C, H, W, K, R, R, st, pt, pl, P, Q)
        rcs = {
            'conv fwd': lib.bn_conv2d_fwd(p, p, p, p, *conv, 1, 0.05, p, big, None),
            'conv bwd_data': lib.bn_conv2d_bwd_data(p, p, p, None, *conv, 0, 0.05, p, big, None),
            'conv bwd_data*lrelu': lib.bn_conv2d_bwd_data(p, p, p, p, *conv, 1, 0.05, p, big, None),
            'conv bwd_weight': lib.bn_conv2d_bwd_weight(p, p, p, p, *conv, 0, p, big, None),
            # the transposed layer with the same maps: small (K, P, Q) -> big (C, H, W)
            'convT fwd': lib.bn_convT2d_fwd(p, p, p, p, N, K, P, Q, C, R, R, st, pt, pl, H, W, 1, 0.05, p, big,
                                            None),
            'convT bwd_data': lib.bn_convT2d_bwd_data(p, p, p, p, N, K, P, Q, C, R, R, st, pt, pl, H, W, 1,
                                                      0.05, p, big, None),
            'convT bwd_weight': lib.bn_convT2d_bwd_weight(p, p, p, p, N, K, P, Q, C, R, R, st, pt, pl, H, W, 0,
                                                          p, big, None),
        }
        for role, rc in rcs.items():
            if rc < 0:
                refused.append((name, role, rc))
    assert not refused, refused[:10]


def test_scratch_sizes_of_the_sweep_are_finite():
    """bn_conv_ws_bytes of every role: below 4 GB for these (<= 6 M element) operands."""
    lib = _hip.load()
    for name, N, C, H, W, K, R, st, (pt, pb), (pl, pr) in _cases(11, 300):
        P, Q = (H + pt + pb - R) // st + 1, (W + pl + pr - R) // st + 1
        for op in (1, 2, 3):          # BN_OP_CONV_*
            ws = lib.bn_conv_ws_bytes(op, N, C, H, W, K, R, R, st, pt, pl, P, Q)
            assert 0 <= ws < (1 << 32), (name, op, ws)
        for op in (4, 5, 6):          # BN_OP_CONVT_*: small (K, P, Q) -> big (C, H, W)
            ws = lib.bn_conv_ws_bytes(op, N, K, P, Q, C, R, R, st, pt, pl, H, W)
            assert 0 <= ws < (1 << 32), (name, op, ws)


def test_padded_taps_queries_and_argument_checks():
    """bn_conv_taps_bytes / _pad / _hint (round 6) on the host: a 5x5 layer pads nothing, the 3x3 / 4x4 stride-2 layers of
    the architecture search / ae_arch_2.json ask for a [pairs][5][5] copy in both of their roles, the weight-gradient ops
    and unknown ops for none; the batched copy checks its arguments before it launches anything."""
    lib = _hip.load()
    five = (8, 32, 32, 32, 64, 5, 5, 2, 1, 1, 16, 16)
    three = (8, 32, 32, 32, 64, 3, 3, 2, 0, 0, 16, 16)          # TF-"same": first tap on the frame
    four = (8, 64, 32, 32, 64, 4, 4, 2, 1, 1, 16, 16)           # ae_arch_2.json
    for op in (_hip.OP_CONV_FWD, _hip.OP_CONV_BWD_D):
        assert lib.bn_conv_taps_bytes(op, *five) == 0
        assert lib.bn_conv_taps_bytes(op, *three) >= 64 * 32 * 25 * 4
        assert lib.bn_conv_taps_bytes(op, *four) >= 64 * 64 * 25 * 4
    # the transposed layer with the same maps: small (64, 16, 16) -> big (32, 32, 32)
    threeT = (8, 64, 16, 16, 32, 3, 3, 2, 0, 0, 32, 32)
    assert lib.bn_conv_taps_bytes(_hip.OP_CONVT_FWD, *threeT) == lib.bn_conv_taps_bytes(_hip.OP_CONV_FWD, *three)
    assert lib.bn_conv_taps_bytes(_hip.OP_CONVT_BWD_D, *threeT) == lib.bn_conv_taps_bytes(_hip.OP_CONV_FWD, *three)
    assert lib.bn_conv_taps_bytes(_hip.OP_CONV_BWD_W, *three) == 0 and lib.bn_conv_taps_bytes(99, *three) == 0
    assert lib.bn_conv_taps_pad(0, None, None, None, None) == 0
    buf = (ctypes.c_char * 256)()
    ptrs = (ctypes.c_void_p * 1)(ctypes.addressof(buf))
    geoms = (ctypes.c_int * 13)(_hip.OP_CONV_FWD, *five)
    assert lib.bn_conv_taps_pad(1, ptrs, ptrs, geoms, None) < 0             # a layer that does not pad: refused
    assert lib.bn_conv_taps_pad(1, None, ptrs, geoms, None) < 0
    assert lib.bn_conv_taps_hint(None, None) == 0
