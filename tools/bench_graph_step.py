"""Eager launches against the recorded HIP graph (behavenet_amd/fitting/graph_step.py): ms per
training step (zero_grad + loss fwd/bwd + Adam) for the headline shape, the reference's 64x48
integration shape, a small batch, and rank 0 of 8 of a frame-sharded 256-frame trial (emulated
rank: collectives are identities, the compute side of strong scaling).
usage: python tools/bench_graph_step.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.fitting.graph_step import GraphedLoss
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.data.synthetic import base_hparams, make_frames

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def run(label, dim, batch, shard=None, extra=None):
    arch = load_handcrafted_arch(list(dim), 12, None, check_memory=False)
    hp = base_hparams(arch, 'ae', extra or {})
    hp['device'] = 'cuda'
    out = []
    for graphed in (False, True):
        np.random.seed(0); torch.manual_seed(0)
        m = AE(hp).to('cuda')
        opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
        data = {'images': [torch.from_numpy(make_frames(batch, list(dim), seed=1)).cuda()]}
        fn = GraphedLoss(m) if graphed else m.loss
        pend = []

        def step():
            m.train(); opt.zero_grad()
            pend.append(fn(data, dataset=0, accumulate_grad=True))
            if len(pend) > 3:
                pend.pop(0)['loss']
            opt.step()

        ctx = bdist.emulate_rank(*shard) if shard else None
        prev = bdist.set_shard_mode('frames') if shard else None
        if ctx: ctx.__enter__()
        try:
            for _ in range(12): step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps): step()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps * 1e3)
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
                bdist.set_shard_mode(prev)
        if graphed and fn.n_replays < steps:
            label += ' [NOT recorded: %d eager]' % fn.n_eager
    print('%-44s eager %7.3f ms   graph %7.3f ms   (%.2fx)' % (label, out[0], out[1], out[0] / out[1]))


run('1x128x128, 256 frames (headline)', (1, 128, 128), 256)
run('1x128x128, 256 frames, batch norm', (1, 128, 128), 256, extra={'ae_batch_norm': True})
run('1x64x48, 256 frames', (1, 64, 48), 256)
run('1x128x128, 32 frames', (1, 128, 128), 32)
run('1x32x32, 32 frames (configs[0])', (1, 32, 32), 32)
run('rank 0 of 8 of a 256-frame trial (frames)', (1, 128, 128), 256, shard=(0, 8))
run('rank 0 of 2 of a 256-frame trial (frames)', (1, 128, 128), 256, shard=(0, 2))
