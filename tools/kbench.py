#!/usr/bin/env python
"""Per-kernel micro-benchmark + on-device cross-check (fast path vs shape-agnostic kernels).

    python tools/kbench.py [--n 200] [--ops fwd,bwd_d,bwd_w] [--layers E1,E2,...]

For each benchmark layer and role it times the C-ABI call with HIP events on the launch stream
and prints achieved TFLOP/s (2*MAC, zero-padded taps counted) and GB/s of algorithmic bytes.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip  # noqa: E402

SLOPE = 0.05
# name: (kind, Cin, Hin, Win, Cout, Hout, Wout, stride, off)
LAYERS = {
    'E0': ('conv', 1, 128, 128, 32, 64, 64, 2, 1),
    'E1': ('conv', 32, 64, 64, 64, 32, 32, 2, 1),
    'E2': ('conv', 64, 32, 32, 128, 16, 16, 2, 1),
    'E3': ('conv', 128, 16, 16, 256, 8, 8, 2, 1),
    'E4': ('conv', 256, 8, 8, 512, 2, 2, 5, 1),
    'D0': ('convT', 512, 2, 2, 256, 8, 8, 5, 1),
    'D1': ('convT', 256, 8, 8, 128, 16, 16, 2, 1),
    'D2': ('convT', 128, 16, 16, 64, 32, 32, 2, 1),
    'D3': ('convT', 64, 32, 32, 32, 64, 64, 2, 1),
    'D4': ('convT', 32, 64, 64, 1, 128, 128, 2, 1),
}


RING = 0   # keep this many previous outputs alive so that results rotate through > 256 MB (MALL)


def timeit(fn, iters):
    keep = []
    keep.append(fn())
    for _ in range(RING):
        keep.append(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        keep.append(fn())
        if len(keep) > RING + 1:
            keep.pop(0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _poison():
    from tests import debug_lib
    debug_lib.poison_lds('cuda')


def relerr(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=200)
    ap.add_argument('--ops', default='fwd,bwd_d,bwd_w')
    ap.add_argument('--layers', default=','.join(LAYERS))
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--ring', type=int, default=0,
                    help='outputs kept alive (defeats the 256 MB Infinity Cache for small layers)')
    args = ap.parse_args()
    global RING
    RING = args.ring
    N = args.n
    dev = 'cuda'
    rows = []
    for name in args.layers.split(','):
        kind, ci, hi, wi, co, ho, wo, st, off = LAYERS[name]
        g = torch.Generator(device='cpu').manual_seed(0)
        x = (torch.rand((N, ci, hi, wi), generator=g) - 0.3).to(dev)
        if kind == 'conv':
            w = ((torch.rand((co, ci, 5, 5), generator=g) - 0.5) / (ci * 25) ** 0.5).to(dev)
        else:
            w = ((torch.rand((ci, co, 5, 5), generator=g) - 0.5) / (ci * 6) ** 0.5).to(dev)
        b = (torch.rand((co,), generator=g) - 0.5).to(dev)
        dy = (torch.rand((N, co, ho, wo), generator=g) - 0.5).to(dev)
        geom = (N, ci, hi, wi, co, 5, 5, st, off, off, ho, wo)
        small_pix = ho * wo if kind == 'conv' else hi * wi
        flop = 2.0 * N * ci * co * 25 * small_pix
        byt = 4.0 * N * (ci * hi * wi + co * ho * wo)
        dw = torch.empty_like(w)
        db = torch.empty_like(b)
        if kind == 'conv':
            ops = {
                'fwd': lambda: _hip.conv2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE),
                'bwd_d': lambda: _hip.conv2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, SLOPE),
                'bwd_w': lambda: (_hip.conv2d_bwd_weight(x, dy, dw, db, geom, False), dw)[1],
            }
        else:
            ops = {
                'fwd': lambda: _hip.convT2d_fwd(x, w, b, geom, _hip.ACT_LRELU, SLOPE),
                'bwd_d': lambda: _hip.convT2d_bwd_data(dy, w, geom, x, _hip.ACT_LRELU, SLOPE),
                'bwd_w': lambda: (_hip.convT2d_bwd_weight(x, dy, dw, db, geom, False), dw)[1],
            }
            if co <= 4:
                tgt = torch.rand((N, co, ho, wo), device=dev)
                ops['fwd_sqerr'] = lambda: _hip.convT2d_fwd_sqerr(
                    x, w, b, tgt, None, geom, _hip.ACT_SIGMOID, SLOPE, False)[1]
        for op in args.ops.split(','):
            if op not in ops:
                continue
            if name == 'E0' and op == 'bwd_d':
                continue
            fn = ops[op]
            err = None
            if not args.no_check:
                # the checked run starts from NaN-filled LDS (see tests/test_gpu_kernels.py)
                _poison()
                got = fn().clone()
                prev = _hip.set_force_generic(True)
                want = fn().clone()
                _hip.set_force_generic(prev)
                err = relerr(got, want)
            ms = timeit(fn, args.iters)
            row = {'layer': name, 'op': op, 'N': N, 'ms': round(ms, 4),
                   'tflops': round(flop / ms / 1e9, 2), 'gbs': round(byt / ms / 1e6, 1),
                   'err_vs_generic': err}
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == '__main__':
    main()
