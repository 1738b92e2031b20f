#!/usr/bin/env python
"""Run one benchmark layer/op a few times (target for rocprofv3 --pmc passes)."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
from tools.kbench import LAYERS, SLOPE

ap = argparse.ArgumentParser()
ap.add_argument('--layer', default='E0')
ap.add_argument('--op', default='fwd')
ap.add_argument('--n', type=int, default=200)
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
kind, ci, hi, wi, co, ho, wo, st, off = LAYERS[a.layer]
g = torch.Generator().manual_seed(0)
x = (torch.rand((a.n, ci, hi, wi), generator=g) - 0.3).cuda()
w = (torch.rand((co, ci, 5, 5) if kind == 'conv' else (ci, co, 5, 5), generator=g) - 0.5).cuda()
b = (torch.rand((co,), generator=g) - 0.5).cuda()
dy = (torch.rand((a.n, co, ho, wo), generator=g) - 0.5).cuda()
geom = (a.n, ci, hi, wi, co, 5, 5, st, off, off, ho, wo)
dw, db = torch.empty_like(w), torch.empty_like(b)
P = 'conv2d' if kind == 'conv' else 'convT2d'
for _ in range(a.iters):
    if a.op == 'fwd':
        getattr(_hip, P + '_fwd')(x, w, b, geom, _hip.ACT_LRELU, SLOPE)
    elif a.op == 'fwd_sqerr':      # last decoder layer + sigmoid + squared error (training)
        _hip.convT2d_fwd_sqerr(x, w, b, dy.abs(), None, geom, _hip.ACT_SIGMOID, SLOPE, False)
    elif a.op == 'fwd_u8':         # first encoder layer from uint8 frames
        _hip.conv2d_fwd_u8((x.clamp(0, 1) * 255).to(torch.uint8), w, b, geom, _hip.ACT_LRELU, SLOPE)
    elif a.op == 'bwd_d':
        getattr(_hip, P + '_bwd_data')(dy, w, geom, x, _hip.ACT_LRELU, SLOPE)
    else:
        getattr(_hip, P + '_bwd_weight')(x, dy, dw, db, geom, False)
torch.cuda.synchronize()
