"""Compute side of the parity-exact strong-scaling mode ('frames') on ONE GPU:
the step of rank 0 of R over its slice of the 256-frame trial (emulated rank: collectives are
identities), i.e. what each GPU of an R-GPU job executes between the gradient exchanges.  As the
product runs it: the optimizer sharded for R >= 4 (fitting/distributed.py default_shard_optimizer:
Adam on 1/R of the arena), eager launches and the recorded HIP graph side by side.
    python tools/bench_frames_shard.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.fitting.graph_step import GraphedLoss
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models import AE
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = [int(v) for v in os.environ.get('BN_R', '1,2,4,8').split(',')]
x = torch.rand((1, 256, 1, 128, 128), device='cuda')
data = {'images': x}
prev = bdist.set_shard_mode('frames')
for R in only:
    res = {}
    for graphed in (False, True):
        hp = bench.build_hparams()
        torch.manual_seed(0)
        model = AE(hp).to('cuda')
        sharded = bdist.default_shard_optimizer(R, 'frames')
        opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4, shard_over=R if sharded else 1)
        fn = GraphedLoss(model) if graphed else model.loss
        lo, hi = opt.shard_range(0)
        pend = []

        def step():
            opt.zero_grad()
            with bdist.emulate_rank(0, R):
                pend.append(fn(data, dataset=0, accumulate_grad=True))
            if len(pend) > 3:
                pend.pop(0)['loss']
            opt.step_range(lo, hi)
        from behavenet_amd import hip_functions as hf
        was = hf.set_lazy_losses(True)
        for _ in range(15):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[graphed] = (time.perf_counter() - t0) / steps
        hf.set_lazy_losses(was)
        if graphed and fn.n_replays < steps:
            print('  (graph NOT recorded: %d eager calls)' % fn.n_eager)
        del model, opt
    best = min(res.values())
    if R == only[0]:
        base = best * R          # the one-GPU step this run is compared with (R = 1 when it is in the list)
    print('R=%d: rank step over %3d frames: eager %.3f ms, HIP graph %.3f ms%s -> compute-side bound %.0f '
          'frames/s for the %d-GPU job (%.2f of linear)' % (
              R, 256 // R, res[False] * 1e3, res[True] * 1e3, ', Adam on 1/%d of the arena' % R if sharded else '',
              256 / best, R, base / (R * best)))
bdist.set_shard_mode(prev)
