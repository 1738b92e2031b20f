"""Kernel-level parity on the MI355X: every C-ABI entry point (called through ctypes) against the
CPU oracle's operators on the same seeded inputs.

Tolerance (fp32 path, BASELINE.json north_star: 1e-4 relative).  fp32 sums in a different order
than mkldnn's differ by rounding noise that depends on cancellation, so each check is made
against a float64 evaluation of the same operator on the CPU:
  * err_hip  = max|hip - f64| / max|f64|  must be <= 1e-4 (the stated tolerance), and
  * err_hip <= max(8 * err_cpu32, 3e-6) where err_cpu32 is the fp32 CPU oracle's own error,
    i.e. the HIP result is as close to the exact answer as the reference's arithmetic is.
Where no float64 evaluation is supplied the scale-normalised error vs the fp32 oracle must be
<= 1e-4.
"""

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from behavenet_amd import _hip

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SLOPE = 0.05


def close(got, want, want64=None, rtol=1e-4, norm_tol=1e-4, name='', sum_of=None):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert np.all(np.isfinite(got)), name
    if want64 is None:
        scale = max(np.abs(want).max(), 1e-30)
        err = np.abs(got - want).max() / scale
        assert err <= norm_tol, '%s: normalised max err %.3e' % (name, err)
        return
    w64 = want64.detach().cpu().double().numpy()
    scale = max(np.abs(w64).max(), 1e-30)
    e_hip = np.abs(got - w64).max() / scale
    e_cpu = np.abs(want - w64).max() / scale
    assert e_hip <= norm_tol, '%s: err vs f64 %.3e' % (name, e_hip)
    # `sum_of` (a reduction's terms: the bias gradient of one or two channels is a sum of 10^4..10^5 zero-mean values
    # that cancel to a thousandth of their magnitudes): an error of four roundings of sum |terms| is what ANY fp32
    # summation order may leave -- the gate relative to the result would measure the cancellation, not the kernel
    floor = 3e-6
    if sum_of is not None:
        mags = sum_of.detach().cpu().double().abs()
        floor = max(floor, 4 * 2.0 ** -24 * float(mags.sum(dim=[d for d in range(mags.dim()) if d != 1]).max()) / scale)
    assert e_hip <= max(8 * e_cpu, floor), \
        '%s: hip err %.3e vs f64, cpu fp32 oracle err %.3e' % (name, e_hip, e_cpu)


def act_ref(t, act):
    if act == _hip.ACT_LRELU:
        return F.leaky_relu(t, SLOPE)
    if act == _hip.ACT_SIGMOID:
        return torch.sigmoid(t)
    return t


# (name, N, C, H, W, K, R, stride, (pad_t, pad_b), (pad_l, pad_r))
CONV_CASES = [
    ('E0', 3, 1, 128, 128, 32, 5, 2, (1, 2), (1, 2)),
    ('E1', 2, 32, 64, 64, 64, 5, 2, (1, 2), (1, 2)),
    ('E2', 2, 64, 32, 32, 128, 5, 2, (1, 2), (1, 2)),
    ('E3', 3, 128, 16, 16, 256, 5, 2, (1, 2), (1, 2)),
    ('E4', 5, 256, 8, 8, 512, 5, 5, (1, 1), (1, 1)),
    ('E4_cfg1', 4, 256, 2, 2, 512, 5, 5, (1, 2), (1, 2)),
    ('E0_2ch', 2, 2, 128, 128, 32, 5, 2, (1, 2), (1, 2)),
    ('k4s2', 2, 5, 17, 19, 7, 4, 2, (1, 2), (1, 1)),
    ('k3s1', 2, 4, 9, 11, 6, 3, 1, (1, 1), (1, 1)),
    ('k7s2_valid', 2, 3, 21, 18, 9, 7, 2, (0, 0), (0, 0)),
    ('nonsquare_last', 3, 256, 4, 3, 512, 5, 5, (0, 1), (1, 1)),
    ('odd_channels', 2, 33, 20, 20, 65, 5, 2, (1, 2), (1, 2)),
    # kernels past 9 (a handcrafted architecture json may ask for any size; the search draws <= 9): direct loops
    ('k11s2', 2, 3, 33, 29, 5, 11, 2, (4, 5), (4, 5)),
    ('k13s1_valid', 2, 4, 20, 18, 6, 13, 1, (0, 0), (0, 0)),
    # geometries served by the specialised kernels on zero-padded copies / by the im2col GEMM
    # (csrc/conv_pad.hip): 24x20 and 4x3 small maps, a 64x48 single-channel frame, stride-5
    # windows onto 2x1 and 1x1 maps
    ('pad_24x20', 3, 32, 48, 40, 64, 5, 2, (1, 2), (1, 2)),
    ('pad_4x3', 3, 128, 8, 6, 256, 5, 2, (1, 2), (1, 2)),
    ('pad_10x8_pl2', 3, 128, 20, 15, 256, 5, 2, (1, 2), (2, 2)),
    ('pad_E0_64x48', 3, 1, 64, 48, 32, 5, 2, (1, 2), (1, 2)),
    ('s5_2x1', 4, 256, 6, 5, 512, 5, 5, (2, 2), (0, 0)),
    ('s5_1x1', 4, 256, 4, 3, 512, 5, 5, (0, 1), (1, 1)),
    # round 5: the stride-5 last layer on ANY pair of maps without im2col (csrc/conv_s5win.hip: one dense GEMM
    # per window over the taps that fall on the map): 12x12 <-> 3x3 (192x192 frames; 70 frames: a second,
    # partly filled row tile), 12x10 <-> 3x2 (192x160), channel counts that are no multiples of the tile,
    # the benchmark's 8x8 <-> 2x2 maps with channel counts its own kernels do not take
    ('s5_3x3', 70, 256, 12, 12, 512, 5, 5, (1, 2), (1, 2)),
    ('s5_3x2', 5, 256, 12, 10, 512, 5, 5, (1, 2), (0, 0)),
    ('s5_oddch_2x2', 3, 40, 7, 9, 72, 5, 5, (1, 2), (0, 1)),
    ('s5_2x2_c64_c96', 4, 64, 8, 8, 96, 5, 5, (1, 1), (1, 1)),
    # maps larger than the specialised kernels take: spatial tiles with halos (conv_pad.hip):
    # 2x2 tiles of 32x32 (enc.conv1 of a 192x160 frame), 4x1 tiles, an odd-sized map with the
    # first tap 2 pixels outside, single- and two-channel 192x160 frames on the edge kernels
    ('tile_48x40', 3, 32, 96, 80, 64, 5, 2, (1, 2), (1, 2)),
    ('tile_100x24', 2, 32, 200, 48, 64, 5, 2, (1, 2), (1, 2)),
    ('tile_odd_pl2', 2, 32, 93, 71, 64, 5, 2, (1, 2), (2, 2)),
    ('tile_E0_96x80', 2, 1, 192, 160, 32, 5, 2, (1, 2), (1, 2)),
    ('tile_E0c2_80x128', 2, 2, 160, 256, 32, 5, 2, (1, 2), (1, 2)),
    # kernels smaller than 5x5 on the 5x5 families (zero taps added): ae_arch_2.json's 4x4 layers
    ('k4_64ch_32x32', 3, 64, 64, 64, 64, 4, 2, (1, 1), (1, 1)),
    ('k4_64ch_8x8', 3, 64, 16, 16, 64, 4, 2, (1, 1), (1, 1)),
    ('k3_16x16', 2, 32, 32, 32, 64, 3, 2, (1, 1), (1, 1)),
    ('k4x3_24x20', 2, 32, 48, 40, 64, 4, 2, (1, 1), (1, 1)),
    # 3x3 stride 2 with TF-"same" padding (first tap ON the frame: what the reference's random architecture search
    # draws a quarter of the time): the taps sit at (1, 1) of the 5x5 ones, the layer runs with offsets (1, 1)
    ('k3s2_same_32x32', 3, 32, 64, 64, 64, 3, 2, (0, 1), (0, 1)),
    ('k3s2_same_12x10', 5, 64, 24, 20, 128, 3, 2, (0, 1), (0, 1)),
    ('k3s2_pt0_pl1', 2, 32, 32, 32, 64, 3, 2, (0, 1), (1, 0)),
    ('k3s2_same_E0', 2, 1, 128, 128, 32, 3, 2, (0, 1), (0, 1)),
    ('k3s2_same_8x8_4x4', 9, 64, 8, 8, 96, 3, 2, (0, 1), (0, 1)),
    ('k3s2_same_16x16_8x8', 5, 64, 16, 16, 96, 3, 2, (0, 1), (0, 1)),
    ('k2s2_same_16x16', 3, 32, 16, 16, 64, 2, 2, (0, 0), (0, 0)),
    # kernels larger than 5x5 with stride 2 (the architecture search draws 7 and 9): tap blocks on shifted copies /
    # phases at stride 1 (csrc/conv_pad.hip)
    ('k7s2_same_32x32', 3, 32, 32, 32, 64, 7, 2, (2, 3), (2, 3)),
    ('k9s2_same_32x32', 2, 64, 32, 32, 128, 9, 2, (3, 4), (3, 4)),
    ('k9s2_same_16x16', 4, 32, 16, 16, 64, 9, 2, (3, 4), (3, 4)),
    ('k7s2_same_24x20', 3, 32, 24, 20, 64, 7, 2, (2, 3), (2, 3)),
    ('k7s2_same_E0', 2, 1, 64, 64, 32, 7, 2, (2, 3), (2, 3)),
    ('k9s2_same_E0', 2, 1, 128, 128, 16, 9, 2, (3, 4), (3, 4)),
    ('k6s2_k8', 2, 32, 16, 16, 32, 8, 2, (3, 3), (3, 3)),
    # ... and with stride 1 (under max pooling): 2 x 2 blocks of taps on four shifted copies of the big map
    ('s1_k7_32x32', 3, 32, 32, 32, 64, 7, 1, (3, 3), (3, 3)),
    ('s1_k9_24x20', 2, 16, 24, 20, 32, 9, 1, (4, 4), (4, 4)),
    ('s1_k9_1ch_64x64', 2, 1, 64, 64, 16, 9, 1, (4, 4), (4, 4)),
    ('s1_k7_valid_16x12', 2, 32, 22, 18, 32, 7, 1, (0, 0), (0, 0)),
    ('s1_k9_16x16_n9', 9, 64, 16, 16, 32, 9, 1, (4, 4), (4, 4)),
    ('s1_k9_1ch_128x128', 2, 1, 128, 128, 16, 9, 1, (4, 4), (4, 4)),
    # from one / two channels at stride 1 (the first layer under max pooling): k_down_s1_in1
    ('s1_k7_2ch_50x70', 2, 2, 50, 70, 24, 7, 1, (3, 3), (3, 3)),
    ('s1_k3_1ch_33x20', 3, 1, 33, 20, 5, 3, 1, (1, 1), (1, 1)),
    ('s1_k5_1ch_valid_60x64', 2, 1, 64, 68, 32, 5, 1, (0, 0), (0, 0)),
    # at most 32 small-side channels on 16 / 32 / 64-wide maps: the weight gradient's wave groups (k_wgrad4s_mfma<.., NA, NB>)
    ('s1_k5_32x32_c16', 3, 16, 32, 32, 32, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_16x16_c32', 5, 32, 16, 16, 16, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_64x64_c48', 2, 48, 64, 64, 24, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_60x64_c16', 3, 16, 60, 64, 32, 5, 1, (2, 2), (2, 2)),
    ('s2_c16_to_32_64x64', 3, 16, 64, 64, 32, 5, 2, (1, 2), (1, 2)),
    ('s2_c32_to_32_32x32', 5, 32, 32, 32, 32, 5, 2, (1, 2), (1, 2)),
    ('s2_c48_to_24_16x16', 9, 48, 16, 16, 24, 5, 2, (1, 2), (1, 2)),
    ('k3s2_same_c16_to_32', 3, 16, 64, 64, 32, 3, 2, (0, 1), (0, 1)),
    ('k3s1_c16_to_32_64x64', 2, 16, 64, 64, 32, 3, 1, (1, 1), (1, 1)),
    ('k3s1_c32_to_16_32x32', 3, 32, 32, 32, 16, 3, 1, (1, 1), (1, 1)),
    ('s1_k5_valid_4ch_128x128', 2, 4, 132, 132, 16, 5, 1, (0, 0), (0, 0)),
    ('s1_k7_64x64', 2, 16, 64, 64, 32, 7, 1, (3, 3), (3, 3)),
    ('s1_k5_valid_64ch_64x64', 2, 64, 68, 68, 32, 5, 1, (0, 0), (0, 0)),
    # round 6: a 16-channel small side on the 16-row MFMA tile (k_down2_m16): stride 2 and stride 1, a map that is no
    # power of two, 4x4 taps (zero-extended, multiplied as they are), a batch small enough for the reduction split
    ('m16_s2_c32_to_16_32x32', 3, 32, 64, 64, 16, 5, 2, (1, 2), (1, 2)),
    ('m16_s2_c64_to_16_12x10', 5, 64, 24, 20, 16, 5, 2, (1, 2), (1, 2)),
    ('m16_s1_c32_to_16_64x64', 2, 32, 64, 64, 16, 5, 1, (2, 2), (2, 2)),
    ('m16_s1_c48_to_16_24x20_n7', 7, 48, 24, 20, 16, 5, 1, (2, 2), (2, 2)),
    ('m16_k4s2_c32_to_16', 3, 32, 32, 32, 16, 4, 2, (1, 1), (1, 1)),
    ('m16_s2_c64_to_16_8x8_n40', 40, 64, 16, 16, 16, 5, 2, (1, 2), (1, 2)),
    ('m16_s1_c16_to_16_64x64', 2, 16, 64, 64, 16, 5, 1, (2, 2), (2, 2)),
    # round 6: the stride-1 single-channel weight gradient's DMA generation (k_wgrad_c1e): three column blocks and an
    # odd number of frames, offsets that are not the kernel's half, two-channel frames, 32 channels
    ('s1_k5_1ch_64x192_n5', 5, 1, 64, 192, 16, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_1ch_pad13_128x64', 3, 1, 128, 64, 32, 5, 1, (1, 3), (3, 1)),
    ('s1_k5_1ch_pad40_64x64', 3, 1, 64, 64, 16, 5, 1, (4, 0), (0, 4)),
    ('s1_k5_2ch_64x128', 2, 2, 64, 128, 16, 5, 1, (2, 2), (2, 2)),
    # (fuzz, round 6: one output channel, 256 frames -- its bias gradient is a sum of 59,136 terms that cancel)
    ('fuzz_db_cancel_18x14_n256', 256, 2, 18, 14, 1, 5, 1, (3, 4), (0, 1)),
    # round 6: the weight gradient of maps with four columns on its own instantiation (16-pixel stages; they ran on
    # zero-padded 4x8 copies): 4x4, 2x4, 3x4 and -- zero-padded to 4x4 -- 4x3 maps, an odd number of frames
    ('w4_4x4_n7', 7, 64, 8, 8, 128, 5, 2, (1, 2), (1, 2)),
    ('w4_2x4_n5', 5, 64, 4, 8, 96, 5, 2, (1, 2), (1, 2)),
    ('w4_3x4_n3', 3, 32, 6, 8, 64, 5, 2, (1, 2), (1, 2)),
    ('w4_4x3_n9', 9, 128, 8, 6, 256, 5, 2, (1, 2), (1, 2)),
    # single-channel frames onto 64 channels: two groups of 32 on the edge kernels
    ('E0_64ch', 2, 1, 128, 128, 64, 5, 2, (1, 2), (1, 2)),
    ('E0_k4_64ch', 2, 1, 128, 128, 64, 4, 2, (1, 1), (1, 1)),
    # stride 1 on the first-generation gather-down kernel (3x3, 5x5; 4x4 through zero-extended taps)
    ('s1_k5_64x64', 2, 16, 64, 64, 32, 5, 1, (2, 2), (2, 2)),
    ('s1_k3_32x32', 3, 32, 32, 32, 64, 3, 1, (1, 1), (1, 1)),
    ('s1_k4_8x8', 3, 64, 8, 8, 64, 4, 1, (1, 2), (1, 2)),
    # round 4: the other two roles of stride-1 layers without im2col -- data gradient on the gather-down
    # kernel with reversed taps, weight gradient on k_wgrad4s_mfma<Q, stride 1> (rows of a frame's last
    # stage below the map, offsets other than 2)
    ('s1_k5_24x16', 5, 32, 24, 16, 32, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_8x8_n9', 9, 64, 8, 8, 64, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_valid_60x64', 2, 16, 64, 68, 32, 5, 1, (0, 0), (0, 0)),
    ('s1_k5_pad13', 3, 32, 16, 16, 64, 5, 1, (1, 3), (3, 1)),
    ('s1_k5_48x40', 2, 16, 48, 40, 32, 5, 1, (2, 2), (2, 2)),
    ('s1_k3_18x12', 5, 32, 18, 12, 48, 3, 1, (1, 1), (1, 1)),
    ('s1_k5_7x20', 4, 64, 7, 20, 32, 5, 1, (2, 2), (2, 2)),
    ('s1_k5_1ch_128x128', 2, 1, 128, 128, 16, 5, 1, (2, 2), (2, 2)),
    # round 4: maps that are no powers of two DIRECTLY on the stride-2 families (runtime tile geometry:
    # any even width / any height for the gather-down role, any width for the gather-up role, widths
    # that are multiples of 4 up to 44 for the weight gradient).  Frame counts that leave the last
    # workgroup's frame group short, tiles with rows below the map, stages that hang over a frame's edge
    ('np2_16x12', 5, 32, 32, 24, 64, 5, 2, (1, 2), (1, 2)),
    ('np2_12x10', 7, 64, 24, 20, 128, 5, 2, (1, 2), (1, 2)),
    ('np2_8x6', 9, 64, 16, 12, 128, 5, 2, (1, 2), (1, 2)),
    ('np2_6x8', 5, 64, 12, 16, 128, 5, 2, (1, 2), (1, 2)),
    ('np2_6x5', 11, 128, 12, 10, 256, 5, 2, (1, 2), (1, 2)),
    ('np2_21x28', 3, 32, 42, 56, 64, 5, 2, (1, 2), (1, 2)),
    ('np2_5x36', 4, 32, 10, 72, 96, 5, 2, (1, 2), (1, 2)),
    ('np2_48x40_n5', 5, 32, 96, 80, 64, 5, 2, (1, 2), (1, 2)),
    # maps wider than any instantiated weight-gradient width: column windows (48 = 2 x 24, 64 = 2 x 32, 96 = 3 x 32)
    ('cw_48x48', 3, 32, 96, 96, 64, 5, 2, (1, 2), (1, 2)),
    ('cw_30x64', 2, 16, 60, 128, 64, 5, 2, (1, 2), (1, 2)),
    ('cw_9x96', 3, 16, 18, 192, 32, 5, 2, (1, 2), (1, 2)),
    ('cw_5x56_n7', 7, 32, 10, 112, 32, 5, 2, (1, 2), (1, 2)),
    # single-channel frames of other sizes on the first-generation edge kernel in blocks of 64 columns: an odd
    # number of output rows, a last block of 4 columns, first taps two pixels outside
    ('edge_gen_35x36', 3, 1, 70, 72, 32, 5, 2, (1, 2), (1, 2)),
    ('edge_gen_9x68', 5, 1, 18, 136, 16, 5, 2, (1, 2), (1, 2)),
    ('edge_gen_pt2', 2, 1, 47, 96, 32, 5, 2, (2, 2), (2, 1)),
    ('edge_gen_2ch_35x36', 3, 2, 70, 72, 32, 5, 2, (1, 2), (1, 2)),
    ('edge_gen_2ch_9x68', 5, 2, 18, 136, 16, 5, 2, (1, 2), (1, 2)),
]


def _conv_setup(case, seed=0):
    name, N, C, H, W, K, R, st, (pt, pb), (pl, pr) = case
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((N, C, H, W), generator=g) - 0.3
    w = (torch.rand((K, C, R, R), generator=g) - 0.5) * (2.0 / np.sqrt(C * R * R))
    b = torch.rand((K,), generator=g) - 0.5
    P = (H + pt + pb - R) // st + 1
    Q = (W + pl + pr - R) // st + 1
    geom = (N, C, H, W, K, R, R, st, pt, pl, P, Q)
    return x, w, b, geom, (pl, pr, pt, pb)


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
@pytest.mark.parametrize('act', [_hip.ACT_LRELU, _hip.ACT_NONE])
def test_conv2d_fwd(case, act):
    x, w, b, geom, pad = _conv_setup(case)
    want = act_ref(F.conv2d(F.pad(x, pad), w, b, stride=geom[7]), act)
    want64 = act_ref(F.conv2d(F.pad(x.double(), pad), w.double(), b.double(), stride=geom[7]), act)
    got = _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, act, SLOPE)
    close(got, want, want64, name=case[0])


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_bwd(case):
    x0, w0, b0, geom, pad = _conv_setup(case, seed=1)
    N, C, H, W, K, R, S, st, pt, pl, P, Q = geom
    g = torch.Generator().manual_seed(5)
    dy = torch.rand((N, K, P, Q), generator=g) - 0.5

    def grads(dt):
        x, w, b = (t.detach().clone().to(dt).requires_grad_(True) for t in (x0, w0, b0))
        # x plays the role of a post-LeakyReLU activation of the layer below
        xin = F.leaky_relu(x, SLOPE)
        xin.retain_grad()
        F.conv2d(F.pad(xin, pad), w, b, stride=st).backward(dy.to(dt))
        return xin, xin.grad, x.grad, w.grad, b.grad
    xin, dxin, dx, dwr, dbr = grads(torch.float32)
    _, dxin64, dx64, dw64, db64 = grads(torch.float64)
    w = w0

    dyd, wd = dy.to(DEV), w.detach().to(DEV)
    xind = xin.detach().to(DEV).contiguous()
    # plain data gradient, and the fused form that applies lrelu'(input) in the epilogue
    close(_hip.conv2d_bwd_data(dyd, wd, geom, None, _hip.ACT_NONE, SLOPE), dxin, dxin64,
          name=case[0] + ' dx')
    close(_hip.conv2d_bwd_data(dyd, wd, geom, xind, _hip.ACT_LRELU, SLOPE), dx, dx64,
          name=case[0] + ' dx*lrelu')
    dw = torch.full_like(wd, 7.0)
    db = torch.full((K,), 7.0, device=DEV)
    _hip.conv2d_bwd_weight(xind, dyd, dw, db, geom, False)
    close(dw, dwr, dw64, name=case[0] + ' dw')
    close(db, dbr, db64, name=case[0] + ' db', sum_of=dy)
    # accumulate: a second call adds on top (cross-chunk accumulation)
    _hip.conv2d_bwd_weight(xind, dyd, dw, db, geom, True)
    close(dw, 2 * dwr, 2 * dw64, name=case[0] + ' dw acc')
    close(db, 2 * dbr, 2 * db64, name=case[0] + ' db acc')


# (name, N, Ci, Hi, Wi, Co, R, stride, torch_padding, crop(l,r,t,b) or None, output_padding)
CONVT_CASES = [
    ('D0', 5, 512, 2, 2, 256, 5, 5, (1, 1), None, 0),
    ('D0_cfg1', 4, 512, 1, 1, 256, 5, 5, 0, (1, 2, 1, 2), 0),
    ('D1', 3, 256, 8, 8, 128, 5, 2, 0, (1, 2, 1, 2), 0),
    ('D2', 2, 128, 16, 16, 64, 5, 2, 0, (1, 2, 1, 2), 0),
    ('D3', 2, 64, 32, 32, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('D4', 3, 32, 64, 64, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('D4_2ch', 2, 32, 64, 64, 2, 5, 2, 0, (1, 2, 1, 2), 0),
    ('k4s2', 2, 7, 9, 10, 5, 4, 2, (1, 1), None, 0),
    ('k3s1', 2, 6, 9, 11, 4, 3, 1, (1, 1), None, 0),
    ('valid_outpad', 2, 9, 8, 6, 3, 7, 2, 0, None, (1, 0)),
    ('nonsquare_first', 3, 512, 1, 1, 256, 5, 5, 0, (1, 1, 0, 1), 0),
    ('s5_3x3', 70, 512, 3, 3, 256, 5, 5, 0, (1, 2, 1, 2), 0),
    ('s5_3x2', 5, 512, 3, 2, 256, 5, 5, 0, (0, 0, 1, 2), 0),
    ('s5_oddch_2x2', 3, 72, 2, 2, 40, 5, 5, 0, (0, 1, 1, 2), 0),
    ('odd_channels', 2, 65, 10, 10, 33, 5, 2, 0, (1, 2, 1, 2), 0),
    ('pad_24x20', 3, 64, 24, 20, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('pad_4x3', 3, 256, 4, 3, 128, 5, 2, 0, (1, 2, 1, 2), 0),
    ('pad_10x8_pl2', 3, 256, 10, 8, 128, 5, 2, 0, (2, 2, 1, 2), 0),
    ('pad_D4_64x48', 3, 32, 32, 24, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('tile_48x40', 3, 64, 48, 40, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('tile_47x36_pt2', 2, 64, 47, 36, 32, 5, 2, 0, (1, 2, 2, 2), 0),
    ('tile_D4_96x80', 2, 32, 96, 80, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('tile_D4c2_80x128', 2, 32, 80, 128, 2, 5, 2, 0, (1, 2, 1, 2), 0),
    ('k4_64ch_16x16', 3, 64, 16, 16, 64, 4, 2, 0, (1, 1, 1, 1), 0),
    ('k3_8x8', 2, 128, 8, 8, 64, 3, 2, 0, (1, 0, 1, 0), 0),
    ('k3s2_same_16x16', 3, 128, 16, 16, 64, 3, 2, 0, (0, 1, 0, 1), 0),
    ('k3s2_same_D4', 2, 32, 64, 64, 1, 3, 2, 0, (0, 1, 0, 1), 0),
    ('k3s2_same_4x4_8x8', 9, 96, 4, 4, 64, 3, 2, 0, (0, 1, 0, 1), 0),
    ('k3s2_same_10x12', 4, 64, 10, 12, 32, 3, 2, 0, (0, 1, 0, 1), 0),
    ('k7s2_same_16x16', 3, 64, 16, 16, 32, 7, 2, 0, (2, 3, 2, 3), 0),
    ('s2_c32_to_16_16x16', 3, 32, 16, 16, 16, 5, 2, 0, (1, 2, 1, 2), 0),
    ('s2_c16_to_16_32x32', 2, 16, 32, 32, 16, 5, 2, 0, (1, 2, 1, 2), 0),
    ('k9s2_same_8x8', 3, 128, 8, 8, 64, 9, 2, 0, (3, 4, 3, 4), 0),
    ('k9s2_same_D4', 2, 32, 64, 64, 1, 9, 2, 0, (3, 4, 3, 4), 0),
    ('k7s2_same_10x12', 3, 64, 10, 12, 32, 7, 2, 0, (2, 3, 2, 3), 0),
    ('D4_64ch', 2, 64, 64, 64, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('D4_k4_64ch', 2, 64, 64, 64, 1, 4, 2, 0, (1, 1, 1, 1), 0),
    # round 4: stride 1 -- forward on the gather-down kernel with reversed taps, weight gradient direct
    ('s1_k5_16x16', 3, 32, 16, 16, 64, 5, 1, (2, 2), None, 0),
    ('s1_k3_32x32', 3, 64, 32, 32, 32, 3, 1, (1, 1), None, 0),
    ('s1_k4_8x8', 5, 64, 8, 8, 64, 4, 1, 0, (1, 2, 1, 2), 0),
    ('s1_k5_64x64', 2, 32, 64, 64, 16, 5, 1, (2, 2), None, 0),
    ('s1_k5_to_1ch_42x72', 3, 16, 42, 72, 1, 5, 1, (2, 2), None, 0),
    ('s1_k7_16x16', 3, 64, 16, 16, 32, 7, 1, (3, 3), None, 0),
    ('s1_k9_32x32', 2, 32, 32, 32, 16, 9, 1, (4, 4), None, 0),
    ('s1_k9_to_1ch_64x64', 2, 16, 64, 64, 1, 9, 1, (4, 4), None, 0),
    ('s1_k7_to_2ch_40x72', 3, 18, 40, 72, 2, 7, 1, (3, 3), None, 0),
    ('s1_k3_to_1ch_50x30', 2, 7, 50, 30, 1, 3, 1, (1, 1), None, 0),
    ('s1_k5_to_1ch_pad04', 2, 16, 20, 24, 1, 5, 1, 0, (0, 4, 4, 0), 0),
    ('s1_k5_to_2ch_128x128', 2, 16, 128, 128, 2, 5, 1, (2, 2), None, 0),
    ('s1_k5_to_1ch_128x128', 2, 16, 128, 128, 1, 5, 1, (2, 2), None, 0),
    ('s1_k5_32_to_1ch_64x64_n3', 3, 32, 64, 64, 1, 5, 1, (2, 2), None, 0),
    # round 4: no powers of two, directly on the stride-2 families (see CONV_CASES)
    ('np2_16x12', 5, 64, 16, 12, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_12x10', 7, 128, 12, 10, 64, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_8x6', 9, 128, 8, 6, 64, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_6x8', 5, 128, 6, 8, 64, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_6x5', 11, 256, 6, 5, 128, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_21x28', 3, 64, 21, 28, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_5x36', 4, 96, 5, 36, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('np2_48x40_n5', 5, 64, 48, 40, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('cw_48x48', 3, 64, 48, 48, 32, 5, 2, 0, (1, 2, 1, 2), 0),
    ('cw_12x72', 4, 32, 12, 72, 16, 5, 2, 0, (1, 2, 1, 2), 0),
    ('edge_gen_35x36', 3, 32, 35, 36, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('edge_gen_9x68', 5, 16, 9, 68, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    # ... and its forward on k_up_c1v<8, false, gen>: blocks of 62 columns (1, 2 and 3 blocks; a last block of
    # one column), strips whose last rows lie below the map, two output channels
    ('edge_gen_up_13x62', 3, 32, 13, 62, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('edge_gen_up_24x125', 2, 16, 24, 125, 2, 5, 2, 0, (1, 2, 1, 2), 0),
    ('edge_gen_up_5x3', 7, 32, 5, 3, 1, 5, 2, 0, (1, 2, 1, 2), 0),
    ('edge_gen_2ch_35x36', 3, 32, 35, 36, 2, 5, 2, 0, (1, 2, 1, 2), 0),
]


def _convT_setup(case, seed=0):
    name, N, Ci, Hi, Wi, Co, R, st, tpad, crop, opad = case
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((N, Ci, Hi, Wi), generator=g) - 0.3
    w = (torch.rand((Ci, Co, R, R), generator=g) - 0.5) * (2.0 / np.sqrt(Ci * R * R / st / st))
    b = torch.rand((Co,), generator=g) - 0.5
    tp = (tpad, tpad) if isinstance(tpad, int) else tpad
    op = (opad, opad) if isinstance(opad, int) else opad
    if crop is not None:
        crop_t, crop_l = crop[2], crop[0]
        Ho = (Hi - 1) * st + R - crop[2] - crop[3]
        Wo = (Wi - 1) * st + R - crop[0] - crop[1]
    else:
        crop_t, crop_l = tp
        Ho = (Hi - 1) * st + R - 2 * tp[0] + op[0]
        Wo = (Wi - 1) * st + R - 2 * tp[1] + op[1]
    geom = (N, Ci, Hi, Wi, Co, R, R, st, crop_t, crop_l, Ho, Wo)

    def ref(xx, ww, bb):
        y = F.conv_transpose2d(xx, ww, bb, stride=st, padding=tp, output_padding=op)
        if crop is not None:
            y = F.pad(y, [-c for c in crop])
        return y
    return x, w, b, geom, ref


@pytest.mark.parametrize('case', CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
@pytest.mark.parametrize('act', [_hip.ACT_LRELU, _hip.ACT_SIGMOID])
def test_convT2d_fwd(case, act):
    x, w, b, geom, ref = _convT_setup(case)
    want = act_ref(ref(x, w, b), act)
    want64 = act_ref(ref(x.double(), w.double(), b.double()), act)
    assert tuple(want.shape[2:]) == (geom[10], geom[11])
    got = _hip.convT2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, act, SLOPE)
    close(got, want, want64, name=case[0])


@pytest.mark.parametrize('case', CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_convT2d_bwd(case):
    x0, w0, b0, geom, ref = _convT_setup(case, seed=1)
    N, Co, Ho, Wo = geom[0], geom[4], geom[10], geom[11]
    g = torch.Generator().manual_seed(5)
    dy = torch.rand((N, Co, Ho, Wo), generator=g) - 0.5

    def grads(dt):
        x, w, b = (t.detach().clone().to(dt).requires_grad_(True) for t in (x0, w0, b0))
        xin = F.leaky_relu(x, SLOPE)
        xin.retain_grad()
        ref(xin, w, b).backward(dy.to(dt))
        return xin, xin.grad, x.grad, w.grad, b.grad
    xin, dxin, dx, dwr, dbr = grads(torch.float32)
    _, dxin64, dx64, dw64, db64 = grads(torch.float64)
    dyd, wd = dy.to(DEV), w0.to(DEV)
    xind = xin.detach().to(DEV).contiguous()
    close(_hip.convT2d_bwd_data(dyd, wd, geom, None, _hip.ACT_NONE, SLOPE), dxin, dxin64,
          name=case[0] + ' dx')
    close(_hip.convT2d_bwd_data(dyd, wd, geom, xind, _hip.ACT_LRELU, SLOPE), dx, dx64,
          name=case[0] + ' dx*lrelu')
    dw = torch.full_like(wd, -3.0)
    db = torch.full((Co,), -3.0, device=DEV)
    _hip.convT2d_bwd_weight(xind, dyd, dw, db, geom, False)
    close(dw, dwr, dw64, name=case[0] + ' dw')
    close(db, dbr, db64, name=case[0] + ' db', sum_of=dy)
    _hip.convT2d_bwd_weight(xind, dyd, dw, db, geom, True)
    close(dw, 2 * dwr, 2 * dw64, name=case[0] + ' dw acc')


@pytest.mark.parametrize('case_name', ['D4', 'D4_2ch', 'k4s2', 'D1', 'pad_D4_64x48', 'tile_D4_96x80',
                                       'tile_D4c2_80x128'])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('act', [_hip.ACT_SIGMOID, _hip.ACT_LRELU])
def test_convT2d_fwd_sqerr(case_name, masked, act):
    """Last decoder layer fused with the pixel loss (bn_convT2d_fwd_sqerr; reference
    aes.py:315-330,466-470 + losses.py:56-59): x_hat, the per-frame squared-error sums and
    d(frame sum)/d(pre-activation) against the oracle's operators -- on the fused VALU kernel
    (D4 geometries; round 5: also frames that are not 128 columns wide, in blocks of 62 input columns:
    64x48, 192x160 and two-channel 160x256 frames) and on the composed fallback (any other geometry)."""
    case = [c for c in CONVT_CASES if c[0] == case_name][0]
    x, w, b, geom, ref = _convT_setup(case, seed=3)
    N, Co, Ho, Wo = geom[0], geom[4], geom[10], geom[11]
    g = torch.Generator().manual_seed(9)
    target = torch.rand((N, Co, Ho, Wo), generator=g)
    mask = (torch.rand((N, Co, Ho, Wo), generator=g) > 0.3).float() if masked else None

    def oracle(dt):
        pre = ref(x.to(dt), w.to(dt), b.to(dt)).requires_grad_(True)
        xh = act_ref(pre, act)
        d = (xh - target.to(dt)) ** 2
        if mask is not None:
            d = d * mask.to(dt)
        sums = d.reshape(N, -1).sum(dim=1)
        sums.sum().backward()
        return xh.detach(), sums.detach(), pre.grad
    xh32, s32, d32 = oracle(torch.float32)
    xh64, s64, d64 = oracle(torch.float64)
    md = mask.to(DEV) if mask is not None else None
    for want in (True, False):
        xh, dpre, part = _hip.convT2d_fwd_sqerr(
            x.to(DEV), w.to(DEV), b.to(DEV), target.to(DEV), md, geom, act, SLOPE, want)
        assert (xh is not None) == want
        if want:
            close(xh, xh32, xh64, name=case_name + ' xhat')
        assert part.shape[0] == N
        close(part.sum(dim=1), s32, s64, name=case_name + ' frame sums')
        close(dpre, d32, d64, name=case_name + ' dpre')
    # backward-side helper: per-frame scale x per-chunk upstream gradient, in place
    fs = torch.rand((N,), generator=g) + 0.5
    gs = torch.rand((2,), generator=g) + 0.5
    grp = torch.tensor([0 if i < (N + 1) // 2 else 1 for i in range(N)], dtype=torch.int32)
    want_scaled = d32 * (fs * gs[grp.long()]).reshape(N, 1, 1, 1)
    _hip.scale_frames(dpre, fs.to(DEV), gs.to(DEV), grp.to(DEV))
    close(dpre, want_scaled, name=case_name + ' scaled dpre')
    t = d32.clone().to(DEV)
    _hip.scale_frames(t, fs.to(DEV))
    close(t, d32 * fs.reshape(N, 1, 1, 1), name='scale_frames without groups')


def test_conv_stack_sq_err_matches_unfused_path():
    """ConvStackSqErrFn (decoder stack + fused loss) against ConvStackFn + ChunkedSqErrFn on the
    same parameters: chunk losses, x_hat and every gradient (incl. the stack's input)."""
    from behavenet_amd import hip_functions as hf
    g = torch.Generator().manual_seed(2)
    n = 7
    plan = [hf.ConvLayerPlan('convT', 8, 32, 32, 32, 64, 64, 5, 5, 2, 1, 1, _hip.ACT_LRELU),
            hf.ConvLayerPlan('convT', 32, 64, 64, 1, 128, 128, 5, 5, 2, 1, 1, _hip.ACT_SIGMOID)]
    mk = lambda *sh: ((torch.rand(sh, generator=g) - 0.5) * 0.2).to(DEV)
    base = [mk(8, 32, 5, 5), mk(32), mk(32, 1, 5, 5), mk(1)]
    x0 = (torch.rand((n, 8, 32, 32), generator=g) - 0.3).to(DEV)
    target = torch.rand((n, 1, 128, 128), generator=g).to(DEV)
    mask = (torch.rand((n, 1, 128, 128), generator=g) > 0.2).float().to(DEV)
    bounds = [(0, 4), (4, 7)]
    scales = [1.0 / (4 * 128 * 128), 1.0 / (3 * 128 * 128)]
    wts = torch.tensor([1.3, 0.6], device=DEV)
    res = []
    for fused in (False, True):
        params = [p.clone().requires_grad_(True) for p in base]
        x = x0.clone().requires_grad_(True)
        if fused:
            terms, xh = hf.conv_stack_sq_err(plan, x, params, target, mask, bounds, scales, True)
        else:
            xh = hf.conv_stack(plan, x, params)
            terms = hf.chunked_sq_err(xh, target, mask, bounds, scales)
        (terms * wts).sum().backward()
        res.append((terms.detach(), xh.detach(), x.grad, [p.grad for p in params]))
    (t0, xh0, dx0, g0), (t1, xh1, dx1, g1) = res
    close(t1, t0, norm_tol=1e-6, name='chunk terms')
    close(xh1, xh0, norm_tol=1e-6, name='x_hat')
    close(dx1, dx0, norm_tol=2e-5, name='dx')
    for a, b_ in zip(g1, g0):
        close(a, b_, norm_tol=2e-5, name='param grad')


BN_CASES = [  # (N, C, H, W, momentum, training, affine)
    (200, 32, 16, 16, 0.1, True, True),
    (7, 64, 9, 5, 0.1, True, True),
    (3, 512, 2, 2, None, True, True),
    (2, 33, 7, 3, 0.3, True, False),
    (5, 16, 8, 8, 0.1, False, True),
    (1, 1200, 1, 3, 0.1, True, True),
]


@pytest.mark.parametrize('case', BN_CASES)
@pytest.mark.parametrize('act', [_hip.ACT_LRELU, _hip.ACT_NONE])
def test_batchnorm_act(case, act):
    """nn.BatchNorm2d (+LeakyReLU) forward, running statistics and backward (aes.py:90-97)."""
    from behavenet_amd.hip_functions import BatchNormActFn
    N, C, H, W, mom, training, affine = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, C, H, W), generator=g) * 1.7 + 0.4
    gy = torch.randn((N, C, H, W), generator=g)
    mods = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64), ('hip', torch.float32)):
        m = torch.nn.BatchNorm2d(C, momentum=mom, affine=affine)
        with torch.no_grad():
            if affine:
                m.weight.copy_(torch.rand((C,), generator=torch.Generator().manual_seed(5)) + 0.5)
                m.bias.copy_(torch.rand((C,), generator=torch.Generator().manual_seed(6)) - 0.5)
            m.running_mean.copy_(torch.rand((C,), generator=torch.Generator().manual_seed(7)))
            m.running_var.copy_(torch.rand((C,), generator=torch.Generator().manual_seed(8)) + 0.5)
        m = m.to(dt)
        m.train(training)
        mods[key] = m
    mods['hip'] = mods['hip'].to(DEV)

    outs = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64)):
        xi = x.detach().clone().to(dt).requires_grad_(True)
        for rep in range(2):   # two calls: running statistics move twice (momentum=None path)
            y = act_ref(mods[key](xi), act)
        y.backward(gy.to(dt))
        outs[key] = (y, xi.grad)
    xh = x.detach().clone().to(DEV).requires_grad_(True)
    mh = mods['hip']
    for rep in range(2):
        yh = BatchNormActFn.apply(xh, mh.weight, mh.bias, mh, act)
    yh.backward(gy.to(DEV))

    close(yh, outs['f32'][0], outs['f64'][0], name='bn y')
    close(xh.grad, outs['f32'][1], outs['f64'][1], name='bn dx')
    if affine:
        close(mh.weight.grad, mods['f32'].weight.grad, mods['f64'].weight.grad, name='bn dgamma')
        close(mh.bias.grad, mods['f32'].bias.grad, mods['f64'].bias.grad, name='bn dbeta')
    close(mh.running_mean, mods['f32'].running_mean, mods['f64'].running_mean, name='bn rmean')
    close(mh.running_var, mods['f32'].running_var, mods['f64'].running_var, name='bn rvar')
    assert int(mh.num_batches_tracked) == int(mods['f32'].num_batches_tracked)


@pytest.mark.parametrize('offset,spread', [(3.0, 0.05), (-40.0, 0.5), (0.0, 1e-3)])
def test_batchnorm_one_pass_statistics_are_well_conditioned(offset, spread):
    """Round 4: both moments in one pass (sums shifted by the chunk's first value of the channel).
    Channels whose mean is many standard deviations away from zero -- where E[x^2] - mean^2 loses
    everything -- against a float64 nn.BatchNorm2d, two chunks with their own statistics."""
    from behavenet_amd.hip_functions import BatchNormActFn, bn_chunks
    N, C, H, W = 24, 16, 16, 12
    g = torch.Generator().manual_seed(11)
    x = torch.randn((N, C, H, W), generator=g) * spread + offset
    gy = torch.randn((N, C, H, W), generator=g)
    bounds = [(0, 17), (17, 24)]
    outs = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64)):
        m = torch.nn.BatchNorm2d(C, momentum=None).to(dt).train()
        xi = x.detach().clone().to(dt).requires_grad_(True)
        y = torch.cat([F.leaky_relu(m(xi[b:e]), SLOPE) for b, e in bounds])
        y.backward(gy.to(dt))
        outs[key] = (y, xi.grad, m)
    mh = torch.nn.BatchNorm2d(C, momentum=None).to(DEV).train()
    xh = x.detach().clone().to(DEV).requires_grad_(True)
    with bn_chunks(bounds):
        yh = BatchNormActFn.apply(xh, mh.weight, mh.bias, mh, _hip.ACT_LRELU)
    yh.backward(gy.to(DEV))
    close(yh, outs['f32'][0], outs['f64'][0], name='bn y')
    close(xh.grad, outs['f32'][1], outs['f64'][1], name='bn dx')
    close(mh.running_mean, outs['f32'][2].running_mean, outs['f64'][2].running_mean, name='bn rmean')
    close(mh.running_var, outs['f32'][2].running_var, outs['f64'][2].running_var, name='bn rvar')
    assert int(mh.num_batches_tracked) == 2


def test_batchnorm_statistics_with_an_outlier_at_the_first_pixel():
    """the one-pass statistics used to be shifted by the chunk's FIRST value of a channel -- the
    corner pixel of the first frame, often an outlier behind a zero-padded convolution.  With the corner 500
    standard deviations away the variance lost 2.5e5 x of its precision (1 + (mean - s)^2 / var).  The shift is
    now the mean of eight samples spread over the chunk: same comparison as above, against float64."""
    from behavenet_amd.hip_functions import BatchNormActFn, bn_chunks
    N, C, H, W = 24, 16, 16, 12
    g = torch.Generator().manual_seed(12)
    x = torch.randn((N, C, H, W), generator=g) * 0.2 + 1.5
    bounds = [(0, 17), (17, 24)]
    for b, _ in bounds:
        x[b, :, 0, 0] = 100.0                    # the corner pixel of each chunk's first frame
    gy = torch.randn((N, C, H, W), generator=g)
    outs = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64)):
        m = torch.nn.BatchNorm2d(C, momentum=None).to(dt).train()
        xi = x.detach().clone().to(dt).requires_grad_(True)
        y = torch.cat([F.leaky_relu(m(xi[b:e]), SLOPE) for b, e in bounds])
        y.backward(gy.to(dt))
        outs[key] = (y, xi.grad, m)
    mh = torch.nn.BatchNorm2d(C, momentum=None).to(DEV).train()
    xh = x.detach().clone().to(DEV).requires_grad_(True)
    with bn_chunks(bounds):
        yh = BatchNormActFn.apply(xh, mh.weight, mh.bias, mh, _hip.ACT_LRELU)
    yh.backward(gy.to(DEV))
    close(yh, outs['f32'][0], outs['f64'][0], name='bn y (outlier)')
    close(xh.grad, outs['f32'][1], outs['f64'][1], name='bn dx (outlier)')
    close(mh.running_var, outs['f32'][2].running_var, outs['f64'][2].running_var, name='bn rvar (outlier)')


@pytest.mark.parametrize('shape,bounds', [
    # the (channel, interleaved frame-slice) grid with finalize / combine folded into the apply launches (round 6):
    # a chunk of a channel is > 32 K floats, so the owner-workgroup kernels do not take it
    ((40, 8, 32, 32), [(0, 33), (33, 34), (34, 40)]),           # a one-frame chunk between two others
    ((40, 8, 30, 36), [(0, 31), (31, 40)]),                     # 270 16-byte groups per plane: no power of two
    ((70, 4, 64, 32), [(0, 17), (17, 34), (34, 51), (51, 70)]),  # four chunks (the kernels' maximum)
    ((36, 3, 33, 31), [(0, 35), (35, 36)]),                     # odd plane size: the scalar paths
    # one workgroup per channel, a chunk in registers (deep layers): 1024 and 256 threads, odd group counts
    ((210, 64, 8, 8), [(0, 200), (200, 210)]),
    ((210, 96, 2, 2), [(0, 200), (200, 210)]),
    ((23, 70, 6, 4), [(0, 9), (9, 10), (10, 23)]),
    # more chunks than one launch takes (four): the entry point falls back to one call per chunk
    ((15, 8, 8, 8), [(0, 3), (3, 6), (6, 9), (9, 12), (12, 15)]),
    ((45, 4, 64, 32), [(0, 17), (17, 18), (18, 30), (30, 31), (31, 45)]),
    # many channels: two slices for two chunks of very different lengths (found by tools/fuzz_archs.py, seed 410)
    ((210, 512, 3, 3), [(0, 200), (200, 210)]),
    ((210, 512, 6, 4), [(0, 200), (200, 210)]),
    ((210, 512, 9, 13), [(0, 200), (200, 210)]),
    ((210, 256, 29, 21), [(0, 200), (200, 210)]),
    ((210, 512, 27, 19), [(0, 200), (200, 210)]),
    ((210, 512, 29, 21), [(0, 200), (200, 210)]),
])
def test_batchnorm_chunked_paths(shape, bounds):
    """Both round-6 forms of the chunked train-mode batch norm against a float64 nn.BatchNorm2d run chunk by chunk:
    outputs, input gradient, parameter gradients, running estimates (cumulative average: one update per chunk, in
    order) and the batch counter.  The gradients are compared ON THE DEVICE'S LeakyReLU BRANCHES (the float64
    reference back-propagates through the mask ``y > 0`` of the device's output): among the tens of millions of
    pre-activations of the larger cases a few lie within fp32 rounding of zero and may fall on either side -- a tie,
    not an error (their outputs agree to 1e-7 either way), but one flipped element moves dx there by ~|dy|."""
    from behavenet_amd.hip_functions import BatchNormActFn, bn_chunks
    N, C, H, W = shape
    g = torch.Generator().manual_seed(21)
    x = torch.randn((N, C, H, W), generator=g) * 0.7 + torch.randn((1, C, 1, 1), generator=g)
    gy = torch.randn((N, C, H, W), generator=g)
    gamma = torch.rand((C,), generator=torch.Generator().manual_seed(5)) + 0.5
    beta = torch.rand((C,), generator=torch.Generator().manual_seed(6)) - 0.5

    def device_run():
        mh = torch.nn.BatchNorm2d(C, momentum=None).to(DEV).train()
        with torch.no_grad():
            mh.weight.copy_(gamma)
            mh.bias.copy_(beta)
        xh = x.detach().clone().to(DEV).requires_grad_(True)
        with bn_chunks(bounds):
            yh = BatchNormActFn.apply(xh, mh.weight, mh.bias, mh, _hip.ACT_LRELU)
        yh.backward(gy.to(DEV))
        return mh, xh, yh
    mh, xh, yh = device_run()
    mask = (yh.detach() > 0).cpu()
    outs = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64)):
        m = torch.nn.BatchNorm2d(C, momentum=None).to(dt).train()
        with torch.no_grad():
            m.weight.copy_(gamma.to(dt))
            m.bias.copy_(beta.to(dt))
        xi = x.detach().clone().to(dt).requires_grad_(True)
        z = torch.cat([m(xi[b:e]) for b, e in bounds])
        y = torch.where(mask, z, SLOPE * z)
        y.backward(gy.to(dt))
        outs[key] = (y.detach(), xi.grad, m)
        if key == 'f64':
            ties = (z.detach() > 0) != mask
            assert int(ties.sum()) <= max(2, 1e-6 * z.numel())
            if bool(ties.any()):
                assert float(z.detach()[ties].abs().max()) <= 2e-6 * float(z.detach().abs().max())
    r32, r64 = outs['f32'], outs['f64']
    close(yh, r32[0], r64[0], name='bn y')
    close(xh.grad, r32[1], r64[1], name='bn dx')
    close(mh.weight.grad, r32[2].weight.grad, r64[2].weight.grad, name='bn dgamma')
    close(mh.bias.grad, r32[2].bias.grad, r64[2].bias.grad, name='bn dbeta')
    close(mh.running_mean, r32[2].running_mean, r64[2].running_mean, name='bn rmean')
    close(mh.running_var, r32[2].running_var, r64[2].running_var, name='bn rvar')
    assert int(mh.num_batches_tracked) == len(bounds)
    # run-to-run bit-identical (fixed-order reductions, no atomics)
    mh2, xh2, yh2 = device_run()
    assert torch.equal(yh2, yh) and torch.equal(xh2.grad, xh.grad) and torch.equal(mh2.weight.grad, mh.weight.grad)


def test_batchnorm_backward_rebuilds_the_forward_branches():
    """The backward kernels rebuild the LeakyReLU branch from x through the STORED mean / invstd (no y read-back); the
    forward kernels normalise with statistics every workgroup derives itself.  Both must land on the same side of zero
    for EVERY element: dx with the mask taken from y and dx with the mask rebuilt from x agree to rounding (round 6: an
    inlined copy of the finalize arithmetic per call site left that to the compiler's contraction choices)."""
    for shape, bounds in (((210, 512, 27, 19), [(0, 200), (200, 210)]), ((256, 64, 32, 32), [(0, 200), (200, 256)]),
                          ((256, 256, 8, 8), [(0, 200), (200, 256)]), ((64, 32, 64, 64), [(0, 64)]),
                          ((10, 512, 27, 19), [(0, 10)])):
        N, C, H, W = shape
        g = torch.Generator().manual_seed(31)
        # (conv-like data: channel means well away from zero, so that shift + s1 / n has something to round)
        x = (torch.randn(shape, generator=g) * 0.35 + 3.0 * torch.randn((1, C, 1, 1), generator=g)).to(DEV)
        dy = torch.randn(shape, generator=g).to(DEV)
        gamma = (torch.rand((C,), generator=g) + 0.5).to(DEV)
        beta = (torch.rand((C,), generator=g) - 0.5).to(DEV)
        y, mean, invstd = _hip.batchnorm_train_fwd_chunks(x, gamma, beta, None, None, [0.0] * len(bounds), 1e-5,
                                                          _hip.ACT_LRELU, SLOPE, bounds)
        outs = []
        for y_arg, beta_arg in ((y, None), (None, beta)):
            dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
            outs.append(_hip.batchnorm_bwd_chunks(x, y_arg, dy, mean, invstd, gamma, dg, db, False, _hip.ACT_LRELU,
                                                  SLOPE, bounds, beta=beta_arg))
        # (the two instantiations may round their arithmetic differently -- the compiler's contraction choices again --
        # but a flipped branch moves an element by ~|dy| (1 - slope) gamma invstd: orders of magnitude above this gate)
        diff = float((outs[0] - outs[1]).abs().max())
        assert diff <= 2e-6 * float(outs[0].abs().max()), (shape, diff, int((outs[0] != outs[1]).sum()))


@pytest.mark.parametrize('act', [_hip.ACT_LRELU, _hip.ACT_SIGMOID, _hip.ACT_NONE])
def test_act_bwd(act):
    g = torch.Generator().manual_seed(0)
    pre = (torch.rand((7, 33, 5), generator=g) - 0.5).requires_grad_(True)
    y = act_ref(pre, act)
    dy = torch.rand(y.shape, generator=g)
    y.backward(dy)
    got = _hip.act_bwd(dy.to(DEV), y.detach().to(DEV), act, SLOPE)
    close(got, pre.grad, name='act_bwd')


@pytest.mark.parametrize('M,K,N', [(200, 2048, 12), (56, 2048, 12), (200, 12, 2048), (7, 512, 8),
                                   (33, 16, 4), (5, 37, 65), (200, 2048, 16)])
def test_linear(M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = (torch.rand((M, K), generator=g) - 0.4).requires_grad_(True)
    w = ((torch.rand((N, K), generator=g) - 0.5) / np.sqrt(K)).requires_grad_(True)
    b = (torch.rand((N,), generator=g) - 0.5).requires_grad_(True)
    y = F.linear(x, w, b)
    dy = torch.rand(y.shape, generator=g) - 0.5
    y.backward(dy)
    xd, wd, bd, dyd = x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), dy.to(DEV)
    close(_hip.linear_fwd(xd, wd, bd), y, name='linear fwd')
    close(_hip.linear_fwd(xd, wd, None), F.linear(x, w), name='linear fwd nobias')
    dw = torch.empty_like(wd)
    db = torch.empty((N,), device=DEV)
    dx = _hip.linear_bwd(xd, wd, dyd, True, None, _hip.ACT_NONE, 0.0, dw, db, False)
    close(dx, x.grad, name='linear dx')
    close(dw, w.grad, name='linear dw')
    close(db, b.grad, name='linear db')
    _hip.linear_bwd(xd, wd, dyd, False, None, _hip.ACT_NONE, 0.0, dw, db, True)
    close(dw, 2 * w.grad, name='linear dw acc')


@pytest.mark.parametrize('shape,masked', [((5, 1, 32, 32), False), ((3, 2, 128, 128), True),
                                          ((6, 4), True), ((7, 3), False), ((200, 1, 128, 128), False)])
def test_sqerr(shape, masked):
    g = torch.Generator().manual_seed(3)
    a = torch.rand(shape, generator=g)
    b = torch.rand(shape, generator=g)
    m = (torch.rand(shape, generator=g) > 0.3).float() if masked else None
    d = (a - b) ** 2
    if m is not None:
        d = d * m
    want = d.reshape(shape[0], -1).sum(dim=1)
    md = m.to(DEV) if m is not None else None
    got = _hip.sqerr_frame_sums(a.to(DEV), b.to(DEV), md)
    close(got, want, name='frame sums')
    close(_hip.reduce_sum(got, 0.25), 0.25 * want.sum(), name='reduce')
    gs = torch.tensor(0.7, device=DEV)
    dpred = _hip.sqerr_bwd(a.to(DEV), b.to(DEV), md, 0.1, gs)
    want_d = 0.7 * 0.1 * 2 * (a - b)
    if m is not None:
        want_d = want_d * m
    close(dpred, want_d, name='sqerr bwd')


def test_reparam_and_kl():
    g = torch.Generator().manual_seed(4)
    mu = (torch.rand((37, 12), generator=g) - 0.5).requires_grad_(True)
    lv = (torch.rand((37, 12), generator=g) - 0.5).requires_grad_(True)
    eps = torch.randn((37, 12), generator=g)
    z = eps * torch.exp(lv) + mu
    dz = torch.rand(z.shape, generator=g)
    z.backward(dz)
    mud, lvd = mu.detach().to(DEV), lv.detach().to(DEV)
    zd = _hip.reparam_fwd(mud, lvd, eps.to(DEV))
    close(zd, z, name='reparam')
    close(_hip.reparam_bwd(dz.to(DEV), zd, mud), lv.grad, name='reparam dlogvar')

    mu.grad = None
    lv.grad = None
    kl = torch.mean(0.5 * torch.sum(lv.exp() - lv + mu.pow(2) - 1, dim=1))
    kl.backward()
    rows = _hip.kl_rows(mud, lvd)
    close(_hip.reduce_sum(rows, 1.0 / 37), kl, name='kl')
    dmu, dlv = _hip.kl_bwd(mud, lvd, 1.0 / 37, None)
    close(dmu, mu.grad, name='kl dmu')
    close(dlv, lv.grad, name='kl dlogvar')


@pytest.mark.parametrize('N,D', [(200, 12), (56, 12), (7, 3), (200, 32), (1, 4), (300, 16)])
def test_decomposed_kl(N, D):
    """bn_decomposed_kl_fwd/bwd against the oracle's (N, N, D) formulation (losses.py:284-351),
    values and all three input gradients, for arbitrary upstream weights of the three terms."""
    from oracle import ref_cpu
    from behavenet_amd.hip_functions import decomposed_kl_terms
    g = torch.Generator().manual_seed(N * 100 + D)
    z = torch.randn((N, D), generator=g)
    mu = torch.randn((N, D), generator=g) * 0.7
    lv = torch.randn((N, D), generator=g) * 0.5 - 0.3
    wts = torch.tensor([1.3, -0.4, 2.1])
    res = {}
    for key, dt in (('f32', torch.float32), ('f64', torch.float64)):
        zi, mi, li = (t.detach().clone().to(dt).requires_grad_(True) for t in (z, mu, lv))
        terms = torch.stack(ref_cpu.decomposed_kl(zi, mi, li))
        (terms * wts.to(dt)).sum().backward()
        res[key] = (terms.detach(), zi.grad, mi.grad, li.grad)
    zh, mh, lh = (t.detach().clone().to(DEV).requires_grad_(True) for t in (z, mu, lv))
    th = decomposed_kl_terms(zh, mh, lh)
    (th * wts.to(DEV)).sum().backward()
    close(th, res['f32'][0], res['f64'][0], name='dkl terms')
    close(zh.grad, res['f32'][1], res['f64'][1], name='dkl dz')
    close(mh.grad, res['f32'][2], res['f64'][2], name='dkl dmu')
    close(lh.grad, res['f32'][3], res['f64'][3], name='dkl dlogvar')


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_adam_amsgrad_trajectory(wd):
    g = torch.Generator().manual_seed(6)
    n = 100003
    p0 = torch.rand(n, generator=g) - 0.5
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=wd, amsgrad=True)
    pd = p0.to(DEV)
    m, v, vmax = (torch.zeros(n, device=DEV) for _ in range(3))
    for step in range(1, 6):
        grad = torch.rand(n, generator=g) - 0.5
        if step == 3:
            grad = grad * 0.01   # makes max_exp_avg_sq matter
        ref.grad = grad.clone()
        opt.step()
        _hip.adam_amsgrad_step(pd, grad.to(DEV), m, v, vmax, 1e-3, 0.9, 0.999, 1e-8, wd, step)
    close(pd, ref.detach(), norm_tol=1e-6, name='adam p')
    st = opt.state[ref]
    close(m, st['exp_avg'], name='adam m')
    close(v, st['exp_avg_sq'], name='adam v')
    close(vmax, st['max_exp_avg_sq'], name='adam vmax')


def test_u8_to_unit_float_bit_exact():
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, size=(3, 1, 37, 41), dtype=np.uint8)
    want = u8.astype(np.float32) / 255
    got = _hip.u8_to_unit_float(torch.from_numpy(u8).to(DEV)).cpu().numpy()
    assert np.array_equal(got, want)
    allv = np.arange(256, dtype=np.uint8)
    got = _hip.u8_to_unit_float(torch.from_numpy(allv).to(DEV)).cpu().numpy()
    assert np.array_equal(got, allv.astype(np.float32) / 255)


@pytest.mark.parametrize('case_name', ['E0', 'E0_2ch', 'k3s1', 'E1'])
@pytest.mark.parametrize('act', [_hip.ACT_LRELU, _hip.ACT_NONE, _hip.ACT_SIGMOID])
def test_conv2d_fwd_u8(case_name, act):
    """First encoder layer from uint8 frames (bn_conv2d_fwd_u8): bit-identical to the float
    kernel on ``u8.astype(float32) / 255`` (reference data_generator.py:251-263) and within the
    kernel tolerance of the oracle's operator; E0 = the fused kernel, others = the fallback."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    _, w, b, geom, pad = _conv_setup(case)
    N, C, H, W = geom[:4]
    rng = np.random.default_rng(5)
    u8 = rng.integers(0, 256, size=(N, C, H, W), dtype=np.uint8)
    u8[0, 0, 0, :4] = [0, 1, 254, 255]
    x = torch.from_numpy(u8.astype(np.float32) / 255)
    want = act_ref(F.conv2d(F.pad(x, pad), w, b, stride=geom[7]), act)
    want64 = act_ref(F.conv2d(F.pad(x.double(), pad), w.double(), b.double(), stride=geom[7]), act)
    got = _hip.conv2d_fwd_u8(torch.from_numpy(u8).to(DEV), w.to(DEV), b.to(DEV), geom, act, SLOPE)
    close(got, want, want64, name=case_name + ' u8')
    via_float = _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, act, SLOPE)
    assert torch.equal(got, via_float), 'u8 path differs from the float kernel on u8/255'


def _misaligned(t):
    """Contiguous copy of ``t`` whose storage starts 4 bytes past a 16-byte boundary."""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=DEV)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4 and v.is_contiguous()
    return v


@pytest.mark.parametrize('case_name', ['E1', 'E3'])
def test_conv2d_on_unaligned_views(case_name):
    """Tensors that are not 16-byte aligned (views at an odd offset) must not reach the kernels
    that move 16-byte groups: forward, data and weight gradient equal the aligned call's oracle."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, pad = _conv_setup(case, seed=3)
    want = act_ref(F.conv2d(F.pad(x, pad), w, b, stride=geom[7]), _hip.ACT_LRELU)
    want64 = act_ref(F.conv2d(F.pad(x.double(), pad), w.double(), b.double(), stride=geom[7]),
                     _hip.ACT_LRELU)
    for which in ('x', 'w'):
        xd = _misaligned(x) if which == 'x' else x.to(DEV)
        wd = _misaligned(w) if which == 'w' else w.to(DEV)
        got = _hip.conv2d_fwd(xd, wd, b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)
        close(got, want, want64, name='%s unaligned %s' % (case_name, which))
    # transposed-conv direction through the same weights (gather-up family)
    dy = torch.rand(want.shape, generator=torch.Generator().manual_seed(4)) - 0.5
    dx_want = torch.nn.grad.conv2d_input(F.pad(x, pad).shape, w, dy, stride=geom[7])
    pl, pr, pt, pb = pad
    H, W = geom[2], geom[3]
    dx_want = dx_want[:, :, pt:pt + H, pl:pl + W]
    dx64 = torch.nn.grad.conv2d_input(F.pad(x, pad).shape, w.double(), dy.double(), stride=geom[7])
    dx64 = dx64[:, :, pt:pt + H, pl:pl + W]
    got = _hip.conv2d_bwd_data(_misaligned(dy), w.to(DEV), geom, None, _hip.ACT_NONE, SLOPE)
    close(got, dx_want, dx64, name=case_name + ' unaligned dy')


def test_errors_are_loud():
    x = torch.zeros((1, 1, 8, 8))
    with pytest.raises(_hip.HipLibraryError):
        _hip.conv2d_fwd(x, x, None, (1, 1, 8, 8, 1, 3, 3, 1, 1, 1, 8, 8), 0, 0.0)   # CPU tensor
    xd = torch.zeros((1, 1, 8, 8), device=DEV)
    xd = torch.zeros((1, 1, 24, 24), device=DEV)
    wd = torch.zeros((1, 1, 17, 17), device=DEV)
    with pytest.raises(_hip.HipLibraryError):   # kernel larger than supported (16 taps since round 6) -> BN_E_SHAPE
        _hip.conv2d_fwd(xd, wd, None, (1, 1, 24, 24, 1, 17, 17, 1, 8, 8, 24, 24), 0, 0.0)
    with pytest.raises(_hip.HipLibraryError):
        _hip.conv2d_fwd(xd.double(), wd, None, (1, 1, 24, 24, 1, 17, 17, 1, 8, 8, 24, 24), 0, 0.0)


@pytest.mark.parametrize('shape,k,s,pad', [
    ((3, 5, 32, 32), 2, 2, (0, 0)),       # the non-overlapping pooling of the reference archs
    ((4, 16, 64, 48), 2, 2, (0, 0)),      # ... on the two-windows-per-thread kernels (even output width)
    ((2, 3, 8, 12), 2, 2, (0, 0)),
    ((2, 3, 6, 10), 2, 2, (0, 0)),        # odd output width: the general kernels
    ((2, 4, 17, 13), 2, 2, (0, 0)),       # ceil_mode: clipped last windows
    ((2, 3, 16, 20), 3, 2, (1, 1)),       # overlapping windows with padding
    ((1, 2, 9, 9), 3, 3, (0, 0)),
])
def test_maxpool_unpool_vs_torch(shape, k, s, pad):
    """bn_maxpool2d_fwd/bwd and bn_maxunpool2d_fwd/bwd against F.max_pool2d(return_indices=True,
    ceil_mode=True) / F.max_unpool2d and their autograd (bit-exact: selections and copies)."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g)
    if shape[1] == 16:
        x = torch.round(x * 2) / 2          # many exact ties inside a window: the first one in scan order wins
    xr = x.clone().requires_grad_(True)
    y_ref, idx_ref = F.max_pool2d(xr, k, s, padding=pad, return_indices=True, ceil_mode=True)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    y, idx = _hip.maxpool2d_fwd(x.to(DEV), k, s, pad, tuple(y_ref.shape[2:]))
    assert idx.dtype == torch.int32
    assert torch.equal(y.cpu(), y_ref.detach())
    assert torch.equal(idx.cpu().long(), idx_ref)
    dx = _hip.maxpool2d_bwd(dy.to(DEV), idx, tuple(shape[2:]), k, s, pad)
    close(dx, xr.grad, norm_tol=1e-6, name='maxpool bwd')
    if k == s and pad == (0, 0):
        # unpooling (indices unique for non-overlapping windows)
        v = torch.randn(y_ref.shape, generator=g)
        vr = v.clone().requires_grad_(True)
        u_ref = F.max_unpool2d(vr, idx_ref, k, s, output_size=shape[2:])
        du = torch.randn(u_ref.shape, generator=g)
        u_ref.backward(du)
        u = _hip.maxunpool2d_fwd(v.to(DEV), idx, tuple(shape[2:]))
        assert torch.equal(u.cpu(), u_ref.detach())
        if k == 2 and shape[2] % 2 == 0 and shape[3] % 2 == 0:
            # the one-pass form for indices that lie in their own windows (what max_pool hands to max_unpool)
            u2 = _hip.maxunpool2d_fwd(v.to(DEV), idx, tuple(shape[2:]), True)
            assert torch.equal(u2.cpu(), u_ref.detach())
        dv = _hip.maxunpool2d_bwd(du.to(DEV), idx)
        assert torch.equal(dv.cpu(), vr.grad)


@pytest.mark.parametrize('N,C,H,W,K,pad', [(3, 1, 128, 128, 16, (2, 2)), (2, 2, 64, 192, 32, (2, 2)), (5, 1, 36, 60, 16, (2, 2)),
                                           (2, 1, 20, 24, 16, (0, 4)), (3, 1, 66, 68, 48, (1, 3)),
                                           # the second layer: the matrix-core kernel's tile through LDS
                                           (3, 16, 64, 64, 32, (2, 2)), (5, 16, 32, 32, 64, (2, 2)), (2, 32, 24, 20, 32, (2, 2)),
                                           (7, 16, 16, 16, 96, (1, 3)), (2, 64, 12, 36, 32, (2, 2))])
def test_conv_pool_activation_in_one_kernel(N, C, H, W, K, pad):
    """Round 6: the first layer of a max-pooling architecture -- Conv2d, 2x2 / stride-2 pooling, LeakyReLU -- in ONE kernel
    (bn_conv2d_pool2_act_fwd): the same bits and the same winners as the convolution followed by bn_maxpool2d_act_fwd
    (the matrix-core kernel of the layer with the pooling in its epilogue), torch's values and indices, and through
    ConvPoolActFn the same gradients as the two nodes it replaces."""
    from behavenet_amd import hip_functions as hf
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = (torch.rand((N, C, H, W), generator=g) - 0.4)
    w = (torch.rand((K, C, 5, 5), generator=g) - 0.5) * 0.4
    b = torch.rand((K,), generator=g) - 0.5
    pt, pl = pad
    geom = (N, C, H, W, K, 5, 5, 1, pt, pl, H, W)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    got = _hip.conv2d_pool_act_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE)
    layer = hf.ConvLayerPlan('conv', C, H, W, K, H, W, 5, 5, 1, pt, pl, _hip.ACT_NONE)
    if not _hip.conv2d_pool_act_ok(geom):
        # (a map whose tiles are no whole even row blocks: refused, the model convolves and pools in two launches)
        assert got is None and hf.conv_pool_act(layer, xd, (wd, bd), 2, 2, (0, 0), (H // 2, W // 2), _hip.ACT_LRELU) is None
        return
    assert got is not None
    y, idx = got
    conv = _hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_NONE, SLOPE)
    y2, idx2 = _hip.maxpool2d_act_fwd(conv, _hip.ACT_LRELU, SLOPE)
    assert torch.equal(y, y2) and torch.equal(idx, idx2)
    ref = F.conv2d(F.pad(x.double(), (pl, 4 - pl, pt, 4 - pt)), w.double(), b.double())
    yr, ir = F.max_pool2d(ref, 2, 2, return_indices=True)
    close(y, F.leaky_relu(yr, SLOPE).float(), F.leaky_relu(yr, SLOPE), name='conv+pool+act')
    # (indices: fp32 and float64 may pick different winners among values within rounding -- compare the VALUES they name)
    flat = conv.flatten(2)
    assert torch.equal(flat.gather(2, idx.flatten(2).long()), F.max_pool2d(conv, 2, 2).flatten(2))
    # gradients: the fused node against conv_stack + max_pool_act
    dy = (torch.rand(y.shape, generator=g) - 0.5).to(DEV)
    grads = []
    for fused in (True, False):
        wp, bp = wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
        xp = xd.clone().requires_grad_(True)
        if fused:
            out = hf.conv_pool_act(layer, xp, (wp, bp), 2, 2, (0, 0), (H // 2, W // 2), _hip.ACT_LRELU)
            assert out is not None
            yy = out[0]
        else:
            yy, _ = hf.max_pool_act(hf.conv_stack([layer], xp, (wp, bp)), 2, 2, (0, 0), (H // 2, W // 2), _hip.ACT_LRELU)
        yy.backward(dy)
        hf.join_side_streams()
        torch.cuda.synchronize()
        grads.append((xp.grad.clone(), wp.grad.clone(), bp.grad.clone()))
    for a, bb in zip(*grads):
        assert torch.equal(a, bb)
    # the first layer (no data gradient wanted): weight / bias gradient straight from the pooled gradient and the
    # winners (bn_conv2d_pool2_bwd_weight) -- against float64 on the DEVICE's winners (dense gradient = dy act'(y) at idx)
    if not _hip.conv2d_pool_bwd_weight_ws_bytes(geom):
        return
    wp, bp = wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
    yy, ii = hf.conv_pool_act(layer, xd, (wp, bp), 2, 2, (0, 0), (H // 2, W // 2), _hip.ACT_LRELU)
    yy.backward(dy)
    hf.join_side_streams()
    torch.cuda.synchronize()
    gsm = (dy * torch.where(yy.detach() > 0, 1.0, SLOPE)).cpu()
    dense = torch.zeros((N, K, H * W)).scatter_(2, ii.cpu().flatten(2).long(), gsm.flatten(2)).view(N, K, H, W)
    refs = []
    for dt in (torch.float32, torch.float64):
        w_ = w.detach().clone().to(dt).requires_grad_(True)
        b_ = b.detach().clone().to(dt).requires_grad_(True)
        F.conv2d(F.pad(x.detach().to(dt), (pl, 4 - pl, pt, 4 - pt)), w_, b_).backward(dense.to(dt))
        refs.append((w_.grad, b_.grad))
    close(wp.grad, refs[0][0], refs[1][0], name='pooled-side dw')
    close(bp.grad, refs[0][1], refs[1][1], name='pooled-side db', sum_of=dense)
    assert torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize('shape', [(3, 16, 64, 48), (2, 5, 8, 12), (2, 3, 6, 10)])
def test_maxpool_with_activation_vs_torch(shape):
    """max_pool_act (2x2 / stride-2 pooling + LeakyReLU in one pass each way where the pooled width is even, the two
    separate ops elsewhere) against F.max_pool2d + F.leaky_relu and their autograd: bit-exact."""
    from behavenet_amd import hip_functions as hf
    g = torch.Generator().manual_seed(5)
    x = torch.round(torch.randn(shape, generator=g) * 4) / 4
    xr = x.clone().requires_grad_(True)
    p_ref, idx_ref = F.max_pool2d(xr, 2, 2, return_indices=True)
    y_ref = F.leaky_relu(p_ref, SLOPE)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    xd = x.to(DEV).requires_grad_(True)
    y, idx = hf.max_pool_act(xd, 2, 2, (0, 0), tuple(y_ref.shape[2:]), _hip.ACT_LRELU)
    assert torch.equal(y.detach().cpu(), y_ref.detach())
    assert torch.equal(idx.cpu().long(), idx_ref)
    assert getattr(idx, 'bn_own_window', False)
    y.backward(dy.to(DEV))
    assert torch.equal(xd.grad.cpu(), xr.grad)


def _random_conv_cases(seed, count, big=False):
    """Seeded sweep over what the round-4 tile logic has to get right: map sizes that are no powers of
    two (tiles with masked lanes, frames' last tiles / stages hanging over the edge, frame groups cut short
    by the batch), channel counts off the 32 / 64 tile sizes, 3x3 / 4x4 / 5x5 kernels, strides 1 and 2."""
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        st = int(rng.choice([1, 2, 2])) if big else int(rng.choice([1, 2, 2, 2]))
        R = int(rng.choice([6, 7, 7, 8, 9, 9])) if big else int(rng.choice([3, 4, 5, 5, 5]))
        P, Q = int(rng.randint(1, 41)), int(rng.randint(1, 41))
        if rng.rand() < 0.4:
            Q = int(rng.choice([4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48]))
        C, K = int(rng.choice([16, 32, 48, 64, 96, 128])), int(rng.choice([16, 32, 64, 80, 128]))
        N = int(rng.randint(1, 10))
        if st == 2:
            pt, pl = int(rng.choice([1, 1, 1, 2])), int(rng.choice([1, 1, 1, 2]))
            if big:
                # TF-"same" offsets most of the time ((R - 2) // 2 above / left), any other now and then
                pt = (R - 2) // 2 if rng.rand() < 0.6 else int(rng.randint(0, R - 1))
                pl = (R - 2) // 2 if rng.rand() < 0.6 else int(rng.randint(0, R - 1))
            H, W = 2 * P - int(rng.rand() < 0.15), 2 * Q - int(rng.rand() < 0.15)
            pb, pr = (P - 1) * 2 + R - H - pt, (Q - 1) * 2 + R - W - pl
        else:
            pt, pl = int(rng.randint(0, R)), int(rng.randint(0, R))
            H, W = P, Q
            pb, pr = R - 1 - pt, R - 1 - pl
        if H < 1 or W < 1 or pb < 0 or pr < 0 or N * C * H * W > 6e6 or N * K * P * Q > 6e6:
            continue
        cases.append(('rnd%d_s%d_k%d_%dx%d_c%d_k%d_n%d' % (len(cases), st, R, P, Q, C, K, N),
                      N, C, H, W, K, R, st, (pt, pb), (pl, pr)))
    return cases


RANDOM_CASES = _random_conv_cases(2026, 72)
RANDOM_BIG_CASES = _random_conv_cases(2027, 32, big=True)


@pytest.mark.parametrize('case', RANDOM_CASES + RANDOM_BIG_CASES, ids=[c[0] for c in RANDOM_CASES + RANDOM_BIG_CASES])
def test_random_geometries_all_roles(case):
    """Forward, both data-gradient forms and the weight / bias gradients of a seeded sweep of geometries
    against float64 (same gate as the named cases)."""
    x, w, b, geom, pad = _conv_setup(case)
    st = geom[7]
    want = act_ref(F.conv2d(F.pad(x, pad), w, b, stride=st), _hip.ACT_LRELU)
    want64 = act_ref(F.conv2d(F.pad(x.double(), pad), w.double(), b.double(), stride=st), _hip.ACT_LRELU)
    assert tuple(want.shape[2:]) == (geom[10], geom[11]), (want.shape, geom)
    got = _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)
    close(got, want, want64, name=case[0] + ' fwd')
    test_conv2d_bwd(case)


def _random_stride5_cases(seed, n):
    """5x5 stride-5 layers as the reference's planner pads them (TF-"same": out = ceil(in / 5), the total padding
    split before / after, ae_model_architecture_generator.py:379-383) on maps of 1..17 pixels per side, plus
    offsets a planner would not produce (any split of the padding)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        H, W = int(rng.integers(1, 18)), int(rng.integers(1, 18))
        C = int(rng.choice([3, 8, 17, 40, 64]))
        K = int(rng.choice([5, 16, 33, 72, 130]))
        N = int(rng.choice([1, 2, 5, 67]))
        pads = []
        for size in (H, W):
            o = -(-size // 5)
            tot = max(0, (o - 1) * 5 + 5 - size)
            before = tot // 2 if i % 3 else int(rng.integers(0, tot + 1))
            pads.append((before, tot - before))
        out.append(('rs5_%d_%dx%d_c%d_k%d_n%d_p%d%d' % (i, H, W, C, K, N, pads[0][0], pads[1][0]),
                    N, C, H, W, K, 5, 5, pads[0], pads[1]))
    return out


RANDOM_S5_CASES = _random_stride5_cases(5051, 40)


@pytest.mark.parametrize('case', RANDOM_S5_CASES, ids=[c[0] for c in RANDOM_S5_CASES])
def test_random_stride5_geometries_all_roles(case):
    """Round 5: the window GEMMs of csrc/conv_s5win.hip on a seeded sweep of maps, paddings, channel counts and
    batch sizes (tile tails in every dimension), through the Conv2d entry points and -- same maps -- the
    ConvTranspose2d ones, against float64."""
    x, w, b, geom, pad = _conv_setup(case)
    want = act_ref(F.conv2d(F.pad(x, pad), w, b, stride=5), _hip.ACT_LRELU)
    want64 = act_ref(F.conv2d(F.pad(x.double(), pad), w.double(), b.double(), stride=5), _hip.ACT_LRELU)
    assert tuple(want.shape[2:]) == (geom[10], geom[11]), (want.shape, geom)
    got = _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)
    close(got, want, want64, name=case[0] + ' fwd')
    test_conv2d_bwd(case)
    c = case
    tcase = (c[0] + 'T', c[1], c[5], (c[3] + sum(c[8]) - 5) // 5 + 1, (c[4] + sum(c[9]) - 5) // 5 + 1, c[2], 5, 5, 0,
             (c[9][0], c[9][1], c[8][0], c[8][1]), 0)
    xt, wt, bt, geomt, ref = _convT_setup(tcase)
    wantt = act_ref(ref(xt, wt, bt), _hip.ACT_LRELU)
    wantt64 = act_ref(ref(xt.double(), wt.double(), bt.double()), _hip.ACT_LRELU)
    gott = _hip.convT2d_fwd(xt.to(DEV), wt.to(DEV), bt.to(DEV), geomt, _hip.ACT_LRELU, SLOPE)
    close(gott, wantt, wantt64, name=tcase[0] + ' fwd')
    test_convT2d_bwd(tcase)


# the transposed layers between the same maps: small (K, P, Q) -> big (C, H, W), cropped by the conv's pads
RANDOM_T_CASES = [(c[0] + 'T', c[1], c[5], (c[3] + sum(c[8]) - c[6]) // c[7] + 1,
                   (c[4] + sum(c[9]) - c[6]) // c[7] + 1, c[2], c[6], c[7], 0,
                   (c[9][0], c[9][1], c[8][0], c[8][1]), 0)
                  for c in _random_conv_cases(4052, 48) + _random_conv_cases(4053, 20, big=True)]


@pytest.mark.parametrize('case', RANDOM_T_CASES, ids=[c[0] for c in RANDOM_T_CASES])
def test_random_geometries_all_roles_transposed(case):
    """The same sweep through the ConvTranspose2d entry points: gather-up with bias + activation, gather-down
    with the activation derivative of the layer below, weight gradient with the bias sums over the big side."""
    x, w, b, geom, ref = _convT_setup(case)
    want = act_ref(ref(x, w, b), _hip.ACT_LRELU)
    want64 = act_ref(ref(x.double(), w.double(), b.double()), _hip.ACT_LRELU)
    assert tuple(want.shape[2:]) == (geom[10], geom[11]), (want.shape, geom)
    got = _hip.convT2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)
    close(got, want, want64, name=case[0] + ' fwd')
    test_convT2d_bwd(case)


@pytest.mark.parametrize('case_name', ['k4_64ch_32x32', 'k3s2_same_32x32', 'k3_16x16', 'k3s2_pt0_pl1', 'k4x3_24x20',
                                       'k3s2_same_8x8_4x4', 'k2s2_same_16x16'])
def test_padded_taps_made_once_serve_both_roles(case_name):
    """Round 6: a conv stack makes the 5x5 copies of its small-kernel layers' taps in one launch (bn_conv_taps_pad) and
    hands them to the forward and data-gradient entry points (bn_conv_taps_hint, one-shot) -- same bits as the
    copies the entry points make themselves; a hint for other weights is dropped, never used later."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, _ = _conv_setup(case)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    N, C, H, W, K, R, S, st, pt, pl, P, Q = geom
    dy = (torch.rand((N, K, P, Q), generator=torch.Generator().manual_seed(3)) - 0.5).to(DEV)
    assert _hip.conv_taps_bytes(_hip.OP_CONV_FWD, geom) >= K * C * 25 * 4
    assert _hip.conv_taps_bytes(_hip.OP_CONV_BWD_D, geom) == _hip.conv_taps_bytes(_hip.OP_CONV_FWD, geom)
    w5, w5b = _hip.conv_taps_pad([(_hip.OP_CONV_FWD, geom, wd), (_hip.OP_CONV_BWD_D, geom, wd)], DEV)
    t5 = w5[:K * C * 25]
    assert torch.equal(t5, w5b[:K * C * 25])
    assert torch.equal(t5[t5 != 0].sort().values, wd.flatten().sort().values)       # every tap once, zeros elsewhere
    y0 = _hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE)
    y1 = _hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE, w5=w5)
    assert torch.equal(y0, y1)
    dx0 = _hip.conv2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE)
    dx1 = _hip.conv2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE, w5=w5)
    assert torch.equal(dx0, dx1)
    # the copy IS what the kernels read ...
    poison = torch.full_like(w5, float('nan'))
    assert bool(torch.isnan(_hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE, w5=poison)).all())
    # ... only for the weights it was made for, and only in the very next call
    other = wd.clone()
    _hip.load().bn_conv_taps_hint(other.data_ptr(), poison.data_ptr())
    assert torch.equal(_hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE), y0)
    assert torch.equal(_hip.conv2d_fwd(xd, other, bd, geom, _hip.ACT_LRELU, SLOPE), y0)
    _hip.load().bn_conv_taps_hint(wd.data_ptr(), poison.data_ptr())
    dwt = torch.zeros_like(wd)
    _hip.conv2d_bwd_weight(xd, dy, dwt, None, geom, False)          # (an entry point that takes no hint drops it)
    assert torch.equal(_hip.conv2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE), y0)


@pytest.mark.parametrize('case_name', ['k4_64ch_16x16', 'k3_8x8', 'k3s2_same_16x16', 'k3s2_same_10x12'])
def test_padded_taps_made_once_serve_both_roles_transposed(case_name):
    case = [c for c in CONVT_CASES if c[0] == case_name][0]
    x, w, b, geom, _ = _convT_setup(case)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    N, Ci, Hi, Wi, Co, R, S, st, ct, cl, Ho, Wo = geom
    dy = (torch.rand((N, Co, Ho, Wo), generator=torch.Generator().manual_seed(3)) - 0.5).to(DEV)
    assert _hip.conv_taps_bytes(_hip.OP_CONVT_FWD, geom) >= Ci * Co * 25 * 4
    (w5,) = _hip.conv_taps_pad([(_hip.OP_CONVT_FWD, geom, wd)], DEV)
    y0 = _hip.convT2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE)
    assert torch.equal(y0, _hip.convT2d_fwd(xd, wd, bd, geom, _hip.ACT_LRELU, SLOPE, w5=w5))
    dx0 = _hip.convT2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE)
    assert torch.equal(dx0, _hip.convT2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE, w5=w5))
    poison = torch.full_like(w5, float('nan'))
    assert bool(torch.isnan(_hip.convT2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE, w5=poison)).all())
    assert torch.equal(_hip.convT2d_bwd_data(dy, wd, geom, None, _hip.ACT_NONE, SLOPE), dx0)


@pytest.mark.parametrize('case_name', ['s1_k5_64x64', 's1_k3_32x32', 's1_k4_8x8', 's1_k5_24x16', 's1_k5_pad13',
                                       's1_k5_48x40', 's1_k3_18x12', 's1_k5_7x20'])
def test_stride1_roles_run_without_im2col(case_name):
    """Round 4: data gradient = gather-down kernel on reversed taps, weight gradient = the streamlined
    MFMA kernel's stride-1 instantiation (no k_im2col / k_col2im detour)."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, ref = _conv_setup(case)
    N, K, P, Q = geom[0], geom[4], geom[10], geom[11]
    dy = torch.ones((N, K, P, Q), device=DEV)
    for prof, fn, want in (
            (_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy, w.to(DEV), geom, None, _hip.ACT_NONE, SLOPE),
             'on reversed taps'),
            (_hip.PROF_CONV_BWD_W, lambda: _hip.conv2d_bwd_weight(
                x.to(DEV), dy, torch.empty_like(w, device=DEV), torch.empty_like(b, device=DEV), geom, False),
             'stride 1')):
        _hip.prof_select(prof, 0, 0)
        try:
            fn()
            torch.cuda.synchronize()
            _, n, name = _hip.prof_read()
        finally:
            _hip.prof_select(_hip.PROF_NONE)
        assert n >= 1 and want in name and 'im2col' not in name and 'col2im' not in name, name


@pytest.mark.parametrize('case_name', ['s1_k9_16x16_n9', 's1_k7_32x32', 's1_k9_24x20'])
def test_stride1_large_kernels_in_blocks_of_frames(case_name):
    """The shifted copies of a stride-1 7x7 / 9x9 layer are made for blocks of frames that keep them below 2 GB
    (256 frames of 32 channels at 128x128 are more): with the block size turned down to two frames' worth, the
    forward and all gradients of 9 / 3 / 2 frames (a shorter last block, an accumulated weight gradient)."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    N, C, H, W = case[1:5]
    per_frame = 4 * C * (H + 4) * ((W + 4 + 3) // 4 * 4) * 4
    prev = _hip.set_bigk1_block_bytes(2 * per_frame + 64)
    try:
        test_conv2d_fwd(case, _hip.ACT_LRELU)
        test_conv2d_bwd(case)
    finally:
        _hip.set_bigk1_block_bytes(prev)


@pytest.mark.parametrize('case_name', ['k7s2_same_32x32', 'k9s2_same_32x32', 'k9s2_same_16x16', 's1_k7_32x32',
                                       's1_k9_16x16_n9', 's1_k9_24x20'])
def test_kernels_larger_than_5x5_run_without_im2col(case_name):
    """Round 4: 7x7 / 9x9 stride-2 layers are stride-1 5x5 layers on the four phases of the big map
    (csrc/conv_pad.hip, k_space_to_depth / k_depth_to_space), stride-1 ones 5x5 layers on four shifted copies
    (k_shift_cat): every role on the matrix-core kernels."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, ref = _conv_setup(case)
    N, K, P, Q = geom[0], geom[4], geom[10], geom[11]
    dy = torch.ones((N, K, P, Q), device=DEV)
    for prof, fn, want in (
            (_hip.PROF_CONV_FWD, lambda: _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE),
             'k_down2_m'),              # (k_down2_mfma<..> or the 16-row k_down2_m16<..>)
            (_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy, w.to(DEV), geom, None, _hip.ACT_NONE, SLOPE),
             'k_down2_m'),
            (_hip.PROF_CONV_BWD_W, lambda: _hip.conv2d_bwd_weight(
                x.to(DEV), dy, torch.empty_like(w, device=DEV), torch.empty_like(b, device=DEV), geom, False),
             'k_wgrad4s_mfma<')):
        _hip.prof_select(prof, 0, 0)
        try:
            fn()
            torch.cuda.synchronize()
            _, n, name = _hip.prof_read()
        finally:
            _hip.prof_select(_hip.PROF_NONE)
        assert n >= 1 and want in name and 'im2col' not in name and 'col2im' not in name, name


@pytest.mark.parametrize('case_name', ['s5_3x3', 's5_3x2', 's5_oddch_2x2', 's5_2x2_c64_c96', 's5_2x1', 's5_1x1',
                                       'E4_cfg1', 'nonsquare_last'])
def test_stride5_layers_run_without_im2col_on_any_map(case_name):
    """Round 5: the stride-5 last layer of the default architecture, whatever the frame size
    makes of its maps, runs all three roles on the window GEMMs of csrc/conv_s5win.hip -- no k_im2col*, k_col2im,
    k_gemm_tiled, k_up_s5, k_wgrad_s5."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, ref = _conv_setup(case)
    N, K, P, Q = geom[0], geom[4], geom[10], geom[11]
    dy = torch.ones((N, K, P, Q), device=DEV)
    for prof, fn in (
            (_hip.PROF_CONV_FWD, lambda: _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)),
            (_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(dy, w.to(DEV), geom, None, _hip.ACT_NONE, SLOPE)),
            (_hip.PROF_CONV_BWD_W, lambda: _hip.conv2d_bwd_weight(
                x.to(DEV), dy, torch.empty_like(w, device=DEV), torch.empty_like(b, device=DEV), geom, False))):
        _hip.prof_select(prof, 0, 0)
        try:
            fn()
            torch.cuda.synchronize()
            _, n, name = _hip.prof_read()
        finally:
            _hip.prof_select(_hip.PROF_NONE)
        assert n >= 1 and 'k_s5win<' in name, name


@pytest.mark.parametrize('case_name, want', [
    ('tile_48x40', 'k_down2_mfma<'), ('tile_100x24', 'k_down2_mfma<'),
    ('tile_E0_96x80', 'k_down_c1<gen>'), ('tile_E0c2_80x128', 'k_down_c1s<.., 2, 2, gen>'), ('pad_24x20', 'k_down2_mfma<'),
    ('k4_64ch_32x32', 'mfma'), ('k4x3_24x20', 'k_down2_mfma<'), ('np2_6x5', 'on zero-padded 6x6'), ('pad_4x3', 'k_down2_mfma<2, 1> on zero-padded 4x4'),
    ('tile_odd_pl2', 'on zero-padded 47x36'),
    ('s1_k5_64x64', 'k_down2_mfma<1, 2, 5, 0, 1>'), ('s1_k4_8x8', 'k_down2_mfma<')])
def test_large_and_odd_maps_are_served_by_the_specialised_kernels(case_name, want):
    """The dispatch takes the tiled / zero-padded detour (conv_pad.hip), not the direct loops."""
    case = [c for c in CONV_CASES if c[0] == case_name][0]
    x, w, b, geom, ref = _conv_setup(case)
    _hip.prof_select(_hip.PROF_CONV_FWD, 0, 0)
    try:
        y = _hip.conv2d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), geom, _hip.ACT_LRELU, SLOPE)
        torch.cuda.synchronize()
        _, n, name = _hip.prof_read()
    finally:
        _hip.prof_select(_hip.PROF_NONE)
    assert n >= 1 and want in name, name
    # the other two roles: gather-up and weight gradient (tiles where the padded map is too large
    # for their families)
    for prof, fn in ((_hip.PROF_CONV_BWD_D, lambda: _hip.conv2d_bwd_data(
                          torch.ones_like(y), w.to(DEV), geom, None, _hip.ACT_NONE, SLOPE)),
                     (_hip.PROF_CONV_BWD_W, lambda: _hip.conv2d_bwd_weight(
                          x.to(DEV), torch.ones_like(y), torch.empty_like(w, device=DEV),
                          torch.empty_like(b, device=DEV), geom, False))):
        _hip.prof_select(prof, 0, 0)
        try:
            fn()
            torch.cuda.synchronize()
            _, n, name = _hip.prof_read()
        finally:
            _hip.prof_select(_hip.PROF_NONE)
        assert n >= 1 and ('tiles of' in name or 'zero-padded' in name or 'mfma' in name or 'k_down2_m16<' in name or
                           'k_wgrad_c1<' in name or 'k_up_c1v<8, false, gen>' in name), name


def test_dispatch_ladder_is_pinned():
    """csrc/capi.hip's run_down / run_up / run_wgrad are ladders of ten rungs.  Every named case
    and the odd ones of tests/ladder_cases.py (odd widths, strides 3 / 4, kernels of 1 and past 9 taps, unaligned
    views) must land on the kernel tests/golden/dispatch_ladder.json names for it (regenerate with
    tools/ladder_probe.py when a dispatch change is MEANT), and the column-matrix / direct-loop rungs may be reached
    by the (case, role) pairs of ladder_cases.DETOURS only."""
    import json
    from tests import ladder_cases
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'dispatch_ladder.json')) as f:
        want = json.load(f)
    got = ladder_cases.ladder_table()
    assert sorted(got) == sorted(want)
    wrong = [(c, r, got[c][r], want[c][r]) for c in want for r in want[c] if got[c].get(r) != want[c][r]]
    assert not wrong, wrong[:8]
    on_detour = {(c, r) for c in got for r, name in got[c].items()
                 if any(t in name for t in ladder_cases.DETOUR_TOKENS)}
    allowed = {(c, r) for c, roles in ladder_cases.DETOURS.items() for r in roles}
    assert on_detour == allowed, (sorted(on_detour - allowed), sorted(allowed - on_detour))
