"""Stride-1 conv layers through the C ABI: the kernel the dispatch chose and its time per launch (256 frames).\nusage: python tools/probe_stride1.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
SLOPE = 0.05
def probe(label, N, C, H, W, K, R, st, pt, pl, P, Q):
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((N, C, H, W), generator=g) - 0.3).cuda()
    w = ((torch.rand((K, C, R, R), generator=g) - 0.5) * 0.1).cuda()
    b = (torch.rand((K,), generator=g) - 0.5).cuda()
    geom = (N, C, H, W, K, R, R, st, pt, pl, P, Q)
    for _ in range(5):
        _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE)
    _hip.prof_select(_hip.PROF_CONV_FWD, 0, 0)
    for _ in range(10):
        _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE)
    torch.cuda.synchronize()
    ms, n, name = _hip.prof_read()
    _hip.prof_select(_hip.PROF_NONE)
    print('%-34s %-44s %7.1f us' % (label, name, ms * 1e3 / n))
# ae_arch_2.json's last encoder layer: 64 -> 64, 4x4, stride 1, 8x8 maps, 256 frames
probe('arch_2 E4 (k4 s1, 8x8 maps)', 256, 64, 8, 8, 64, 4, 1, 1, 1, 8, 8)
# the same layer with native 3x3 and 5x5 kernels, and on larger maps
probe('k5 s1, 8x8 maps', 256, 64, 8, 8, 64, 5, 1, 2, 2, 8, 8)
probe('k3 s1, 8x8 maps', 256, 64, 8, 8, 64, 3, 1, 1, 1, 8, 8)
probe('k5 s1, 32x32 maps', 256, 64, 32, 32, 64, 5, 1, 2, 2, 32, 32)
probe('k5 s1, 64x64 maps (maxpool conv2)', 256, 16, 64, 64, 32, 5, 1, 2, 2, 64, 64)
