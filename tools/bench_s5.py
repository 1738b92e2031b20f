"""The stride-5 last layer (256 <-> 512 channels, 5x5 stride 5) on the maps a frame size gives it, all three
roles through the C ABI, HIP-event times and TFLOP/s on the products that meet data.
    python tools/bench_s5.py [frames]      (BN_HIP_LIB=<other build> for an A/B on one box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from behavenet_amd import _hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_hip.load()


def pads(n):
    out = -(-n // 5)
    tot = max(0, (out - 1) * 5 + 5 - n)
    return out, tot // 2, tot - tot // 2


def timed(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for H, W in ((12, 12), (12, 10), (4, 3), (2, 2), (8, 8)):
    P, pt, pb = pads(H)
    Q, pl, pr = pads(W)
    C, K = 256, 512
    x = torch.rand((N, C, H, W), device='cuda')
    w = torch.randn((K, C, 5, 5), device='cuda') * 0.01
    b = torch.zeros(K, device='cuda')
    geom = (N, C, H, W, K, 5, 5, 5, pt, pl, P, Q)
    y = _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, 0.05)
    dy = torch.rand_like(y)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    flop = 2.0 * N * C * K * H * W            # every big pixel meets every (c, m) pair once
    t_f = timed(lambda: _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, 0.05))
    t_d = timed(lambda: _hip.conv2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, 0.05))
    t_w = timed(lambda: _hip.conv2d_bwd_weight(x, dy, dw, db, geom, True))
    print('%2dx%-2d -> %dx%d pads (%d,%d)(%d,%d): fwd %7.1f us %5.1f TF | bwd-data %7.1f us %5.1f TF | bwd-weight(+bias) %7.1f us %5.1f TF'
          % (H, W, P, Q, pt, pb, pl, pr, t_f, flop / t_f / 1e6, t_d, flop / t_d / 1e6, t_w, flop / t_w / 1e6))
