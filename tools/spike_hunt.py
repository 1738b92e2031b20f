"""Find what makes one early step of bench.py's loop slow: per-component host time of one_step."""
import os, sys, time, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from behavenet_amd import _hip
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.models import AE
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
rank, world = bdist.init_from_env()
hp = bench.build_hparams()
torch.manual_seed(hp['rng_seed_model'])
model = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4, weight_decay=0)
sess = SyntheticSession(20, 256, bench.DIM, seed=100, trial_splits='8;1;1;0')
gen = SyntheticSessionsGenerator([sess], device='cuda', placement='device')
torch.manual_seed(1); np.random.seed(1)
gen.reset_iterators('train')
gc.callbacks.append(lambda phase, info: print('  [gc %s gen%d]' % (phase, info['generation'])) if info['generation'] == 2 else None)
for i in range(30):
    t = [time.perf_counter()]
    model.train(); opt.zero_grad(); t.append(time.perf_counter())
    data, ds = gen.next_batch('train'); t.append(time.perf_counter())
    if data is None:
        gen.reset_iterators('train'); t.append(time.perf_counter())
        data, ds = gen.next_batch('train')
    t.append(time.perf_counter())
    loss = model.loss(data, dataset=ds, accumulate_grad=True); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])]
    if sum(d) > 12 or i < 2:
        print('step %d total %.2f parts %s' % (i, sum(d), ' '.join('%.2f' % x for x in d)))
