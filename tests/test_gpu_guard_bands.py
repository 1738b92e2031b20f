"""Every operand of every C-ABI entry point is surrounded by NaN guard bands: a kernel that reads
beyond (or before) a tensor and multiplies what it finds by a zero weight / a padded operand
produces NaN here instead of passing on whatever finite values happen to lie next to the tensor.

(Found in the field: raw uint8 staging buffers of the prefetching feed next to float activations
-- 0.4 % of random byte quadruples are NaN bit patterns.)  Shapes: the awkward small batches of
BASELINE configs[0] (1x32x32, 5-7 frames) and one benchmark-shaped case per role."""

import numpy as np
import pytest
import torch

from behavenet_amd import _hip

pytestmark = pytest.mark.gpu
DEV = 'cuda'
SLOPE = 0.05
GUARD = 1 << 15            # floats of NaN on either side (128 KB)


_bands = []


def guarded(t):
    """A device copy of ``t`` with NaN-filled memory directly before and after it."""
    flat = torch.full((t.numel() + 2 * GUARD,), float('nan'), dtype=torch.float32, device=DEV)
    view = flat[GUARD:GUARD + t.numel()].view(t.shape)
    view.copy_(t)
    _bands.append((flat, t.numel()))
    return view


@pytest.fixture(autouse=True)
def _bands_stay_untouched():
    """No kernel may WRITE outside its output tensors either: every guard band made during the
    test is still all-NaN afterwards."""
    _bands.clear()
    yield
    torch.cuda.synchronize()
    for flat, n in _bands:
        assert bool(torch.isnan(flat[:GUARD]).all()), 'write before a tensor'
        assert bool(torch.isnan(flat[GUARD + n:]).all()), 'write past the end of a tensor'
    _bands.clear()


def finite(t, name):
    assert bool(torch.isfinite(t).all()), '%s: non-finite values (over-read into a guard band)' % name


# cfg1 geometry (1x32x32 -> 512x1x1) plus the cfg2 edge layers
CONV = [  # (name, C, H, W, K, P, Q, stride, pad_t, pad_l)
    ('E0', 1, 32, 32, 32, 16, 16, 2, 1, 1), ('E1', 32, 16, 16, 64, 8, 8, 2, 1, 1),
    ('E2', 64, 8, 8, 128, 4, 4, 2, 1, 1), ('E3', 128, 4, 4, 256, 2, 2, 2, 1, 1),
    ('E4', 256, 2, 2, 512, 1, 1, 5, 1, 1), ('E0_128', 1, 128, 128, 32, 64, 64, 2, 1, 1),
    ('E1_128', 32, 64, 64, 64, 32, 32, 2, 1, 1), ('E4_128', 256, 8, 8, 512, 2, 2, 5, 1, 1),
    # 1 -> 64: two groups of 32 channels served in place through channel windows (BnGeom::CsS)
    ('E0_64ch', 1, 128, 128, 64, 64, 64, 2, 1, 1),
]
CONVT = [  # (name, Ci, Hi, Wi, Co, Ho, Wo, stride, crop_t, crop_l)
    ('D0', 512, 1, 1, 256, 2, 2, 5, 1, 1), ('D1', 256, 2, 2, 128, 4, 4, 2, 1, 1),
    ('D2', 128, 4, 4, 64, 8, 8, 2, 1, 1), ('D3', 64, 8, 8, 32, 16, 16, 2, 1, 1),
    ('D4', 32, 16, 16, 1, 32, 32, 2, 1, 1), ('D4_128', 32, 64, 64, 1, 128, 128, 2, 1, 1),
    ('D3_128', 64, 32, 32, 32, 64, 64, 2, 1, 1), ('D0_128', 512, 2, 2, 256, 8, 8, 5, 1, 1),
    ('D4_64ch', 64, 64, 64, 1, 128, 128, 2, 1, 1),
]


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) - 0.4


@pytest.mark.parametrize('n', [5, 6, 7])
@pytest.mark.parametrize('case', CONV, ids=[c[0] for c in CONV])
def test_conv_roles_do_not_read_outside_their_operands(case, n):
    name, C, H, W, K, P, Q, st, pt, pl = case
    geom = (n, C, H, W, K, 5, 5, st, pt, pl, P, Q)
    x, w, b = guarded(_rand(n, C, H, W)), guarded(_rand(K, C, 5, 5) * 0.1), guarded(_rand(K))
    dy = guarded(_rand(n, K, P, Q, seed=1))
    finite(_hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE), name + ' fwd')
    finite(_hip.conv2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, SLOPE), name + ' bwd-data')
    finite(_hip.conv2d_bwd_data(dy, w, geom, None, _hip.ACT_NONE, SLOPE), name + ' bwd-data plain')
    dw, db = guarded(torch.zeros(K, C, 5, 5)), guarded(torch.zeros(K))
    _hip.conv2d_bwd_weight(x, dy, dw, db, geom, False)
    finite(dw, name + ' dw')
    finite(db, name + ' db')
    _hip.conv2d_bwd_weight(x, dy, dw, db, geom, True)
    finite(dw, name + ' dw (accumulate)')


@pytest.mark.parametrize('n', [5, 6, 7])
@pytest.mark.parametrize('case', CONVT, ids=[c[0] for c in CONVT])
def test_convT_roles_do_not_read_outside_their_operands(case, n):
    name, Ci, Hi, Wi, Co, Ho, Wo, st, ct, cl = case
    geom = (n, Ci, Hi, Wi, Co, 5, 5, st, ct, cl, Ho, Wo)
    x, w, b = guarded(_rand(n, Ci, Hi, Wi)), guarded(_rand(Ci, Co, 5, 5) * 0.1), guarded(_rand(Co))
    dy = guarded(_rand(n, Co, Ho, Wo, seed=1))
    for act in (_hip.ACT_LRELU, _hip.ACT_SIGMOID):
        finite(_hip.convT2d_fwd(x, w, b, geom, act, SLOPE), name + ' fwd')
    finite(_hip.convT2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, SLOPE), name + ' bwd-data')
    dw, db = guarded(torch.zeros(Ci, Co, 5, 5)), guarded(torch.zeros(Co))
    _hip.convT2d_bwd_weight(x, dy, dw, db, geom, False)
    finite(dw, name + ' dw')
    finite(db, name + ' db')
    target = guarded(torch.rand(n, Co, Ho, Wo))
    xh, dpre, part = _hip.convT2d_fwd_sqerr(x, w, b, target, None, geom, _hip.ACT_SIGMOID, SLOPE,
                                            True)
    finite(xh, name + ' fused xhat')
    finite(dpre, name + ' fused dpre')
    finite(part, name + ' fused partial sums')


@pytest.mark.parametrize('n', [130, 160])
def test_stride5_second_generation_does_not_read_outside_its_operands(n):
    """The gather-up kernel of the stride-5 layers (k_qg2_up) only serves batches of >= 128 frames:
    130 frames = four full 32-frame tiles and a ragged one, under guard bands, in both of its roles
    (conv data gradient with and without the LeakyReLU' mask, convT forward), and the weight gradient
    with either bias side next to it."""
    geom = (n, 256, 8, 8, 512, 5, 5, 5, 1, 1, 2, 2)
    x, w = guarded(_rand(n, 256, 8, 8)), guarded(_rand(512, 256, 5, 5) * 0.05)
    dy = guarded(_rand(n, 512, 2, 2, seed=1))
    finite(_hip.conv2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, SLOPE), 'E4 bwd-data')
    finite(_hip.conv2d_bwd_data(dy, w, geom, None, _hip.ACT_NONE, SLOPE), 'E4 bwd-data plain')
    dw, db = guarded(torch.zeros(512, 256, 5, 5)), guarded(torch.zeros(512))
    _hip.conv2d_bwd_weight(x, dy, dw, db, geom, False)
    finite(dw, 'E4 dw')
    finite(db, 'E4 db')
    geom_t = (n, 512, 2, 2, 256, 5, 5, 5, 1, 1, 8, 8)
    xt, wt, bt = guarded(_rand(n, 512, 2, 2)), guarded(_rand(512, 256, 5, 5) * 0.05), guarded(_rand(256))
    dyt = guarded(_rand(n, 256, 8, 8, seed=2))
    finite(_hip.convT2d_fwd(xt, wt, bt, geom_t, _hip.ACT_LRELU, SLOPE), 'D0 fwd')
    dwt, dbt = guarded(torch.zeros(512, 256, 5, 5)), guarded(torch.zeros(256))
    _hip.convT2d_bwd_weight(xt, dyt, dwt, dbt, geom_t, True)
    finite(dwt, 'D0 dw')
    finite(dbt, 'D0 db')


@pytest.mark.parametrize('M,K,N', [(5, 512, 4), (6, 512, 8), (7, 2048, 12), (5, 12, 2048),
                                   (7, 4, 512), (200, 2048, 12), (56, 2048, 16), (3, 37, 65)])
def test_linear_does_not_read_outside_its_operands(M, K, N):
    x, w, b = guarded(_rand(M, K)), guarded(_rand(N, K) * 0.1), guarded(_rand(N))
    dy = guarded(_rand(M, N, seed=1))
    finite(_hip.linear_fwd(x, w, b), 'linear fwd')
    dw, db = guarded(torch.zeros(N, K)), guarded(torch.zeros(N))
    dx = _hip.linear_bwd(x, w, dy, True, None, _hip.ACT_NONE, 0.0, dw, db, False)
    finite(dx, 'linear dx')
    finite(dw, 'linear dw')
    finite(db, 'linear db')
    dx = _hip.linear_bwd(x, w, dy, True, x, _hip.ACT_LRELU, SLOPE, None, None, False)
    finite(dx, 'linear dx * lrelu\'')


@pytest.mark.parametrize('shape', [(5, 1, 32, 32), (7, 2, 33, 31), (6, 4)])
def test_elementwise_and_losses_do_not_read_outside_their_operands(shape):
    a, b = guarded(torch.rand(shape)), guarded(torch.rand(shape))
    m = guarded((torch.rand(shape) > 0.3).float())
    finite(_hip.sqerr_frame_sums(a, b, m), 'frame sums')
    finite(_hip.sqerr_bwd(a, b, m, 0.1, None), 'sqerr bwd')
    finite(_hip.act_bwd(a, b, _hip.ACT_SIGMOID, SLOPE), 'act bwd')
    finite(_hip.reduce_sum(guarded(torch.rand(shape[0]))), 'reduce sum')
    if len(shape) == 2:
        mu, lv, eps = guarded(_rand(*shape)), guarded(_rand(*shape)), guarded(_rand(*shape))
        z = _hip.reparam_fwd(mu, lv, eps)
        finite(z, 'reparam')
        finite(_hip.kl_rows(mu, lv), 'kl rows')
        out3, log_qz, lse = _hip.decomposed_kl_fwd(guarded(z), mu, lv)
        finite(out3, 'decomposed kl')


@pytest.mark.parametrize('n,c,hw', [(5, 32, 256), (7, 512, 1), (6, 33, 21)])
def test_batchnorm_does_not_read_outside_its_operands(n, c, hw):
    x = guarded(_rand(n, c, hw, 1))
    gamma, beta = guarded(torch.rand(c) + 0.5), guarded(_rand(c))
    rm, rv = guarded(torch.zeros(c)), guarded(torch.ones(c))
    y, mean, invstd = _hip.batchnorm_train_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, _hip.ACT_LRELU,
                                               SLOPE)
    for t, k in ((y, 'y'), (mean, 'mean'), (invstd, 'invstd'), (rm, 'running mean'),
                 (rv, 'running var')):
        finite(t, 'batch norm ' + k)
    dgamma, dbeta = guarded(torch.zeros(c)), guarded(torch.zeros(c))
    dx = _hip.batchnorm_bwd(x, guarded(y), guarded(_rand(n, c, hw, 1, seed=2)), mean, invstd,
                            gamma, dgamma, dbeta, False, True, _hip.ACT_LRELU, SLOPE)
    finite(dx, 'batch norm dx')
    finite(dgamma, 'batch norm dgamma')
