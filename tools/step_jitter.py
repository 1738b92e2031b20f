"""Per-step GPU time (event deltas) and host time of the bench loop: where are the hiccups?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from behavenet_amd.models import AE
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
hp = bench.build_hparams()
torch.manual_seed(0)
model = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4, weight_decay=0)
sess = SyntheticSession(20, 256, bench.DIM, seed=100, trial_splits='8;1;1;0')
gen = SyntheticSessionsGenerator([sess], device='cuda', placement='device')
gen.reset_iterators('train')
W, K = int(sys.argv[1]) if len(sys.argv) > 1 else 5, 60
for _ in range(W):
    bench.one_step(model, opt, gen)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
host = []
evs[0].record()
for i in range(K):
    t = time.perf_counter()
    bench.one_step(model, opt, gen)
    host.append((time.perf_counter() - t) * 1e3)
    evs[i + 1].record()
torch.cuda.synchronize()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(K)]
print('gpu ms/step :', ' '.join('%.2f' % g for g in gpu))
print('host ms/step:', ' '.join('%.2f' % h for h in host))
print('mean gpu %.3f  median %.3f  max %.3f' % (np.mean(gpu), np.median(gpu), np.max(gpu)))
