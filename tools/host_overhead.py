"""Host-side cost of one training step: time spent issuing the forward and the backward launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from behavenet_amd import hip_functions as hf
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
for _ in range(5):
    bench.one_step(model, opt, gen)
torch.cuda.synchronize()
orig_bwd = hf.backward_chunks
acc = {'bwd': 0.0, 'n': 0}
def timed_bwd(d, *a, **k):
    t = time.perf_counter(); orig_bwd(d, *a, **k); acc['bwd'] += time.perf_counter() - t; acc['n'] += 1
import behavenet_amd.models.aes as aes
aes.backward_chunks = timed_bwd
x = gen.next_batch('train')[0]
# forward issue time: run the forwards only (no_grad off) without syncing
T = 10
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(T):
    with torch.enable_grad():
        for beg, end in ((0, 200), (200, 256)):
            xh, _ = model(x['images'][0][beg:end], dataset=0)
t_fwd = (time.perf_counter() - t0) / T
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(T):
    bench.one_step(model, opt, gen)
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / T
print('step %.2f ms | host issue: forwards %.2f ms, backwards %.2f ms' % (
    t_step * 1e3, t_fwd * 1e3, acc['bwd'] / acc['n'] * 1e3))
