"""Compute side of the parity-exact strong-scaling mode ('frames') on ONE GPU:
the step of rank 0 of R over its slice of the 256-frame trial (emulated rank: collectives are
identities), i.e. what each GPU of an R-GPU job executes between the gradient all-reduces.
    python tools/bench_frames_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models import AE
import bench

hp = bench.build_hparams()
torch.manual_seed(0)
model = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4)
x = torch.rand((1, 256, 1, 128, 128), device='cuda')
data = {'images': x}
prev = bdist.set_shard_mode('frames')
for R in (1, 2, 4, 8):
    def step():
        opt.zero_grad()
        with bdist.emulate_rank(0, R):
            model.loss(data, dataset=0, accumulate_grad=True)
        opt.step()
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('R=%d: rank step over %3d frames %.3f ms -> compute-side bound %.0f frames/s for the %d-GPU job '
          '(%.2f of linear)' % (R, 256 // R, dt * 1e3, 256 / dt, R, (256 / dt) / (R * 58000.0)))
bdist.set_shard_mode(prev)
