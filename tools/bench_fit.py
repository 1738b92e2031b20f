"""Throughput of the product entry point itself: fit(hparams, model, data_generator, exp) on the
headline workload (synthetic 1x128x128 trials of 256 frames resident in HBM), wall time per epoch
against the bench step.   python tools/bench_fit.py [n_epochs]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from behavenet_amd.models import AE
from behavenet_amd.fitting.training import fit
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator


class Exp(object):
    def __init__(self): self.rows, self.version = [], 0
    def log(self, row): self.rows.append(dict(row))
    def save(self): pass


n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hp = bench.build_hparams()
tmp = tempfile.mkdtemp()
hp.update({'max_n_epochs': n_epochs, 'min_n_epochs': n_epochs, 'enable_early_stop': False,
           'val_check_interval': 1, 'expt_dir': tmp, 'version': 0, 'device': 'cuda',
           'rng_seed_train': 0, 'export_latents': False, 'early_stop_history': 10})
os.makedirs(os.path.join(tmp, 'version_0'), exist_ok=True)
torch.manual_seed(0)
model = AE(hp).to('cuda')
model.version = 0
sess = SyntheticSession(40, bench.BATCH, bench.DIM, seed=100, trial_splits='8;1;1;0')
gen = SyntheticSessionsGenerator([sess], device='cuda', placement='device')
exp = Exp()
torch.cuda.synchronize()
t0 = time.perf_counter()
fit(hp, model, gen, exp, method='ae')
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n_tr, n_val, n_test = gen.n_tot_batches['train'], gen.n_tot_batches['val'], gen.n_tot_batches['test']
print('fit: %d epochs (+ epoch 0 without steps) over %d train / %d val / %d test trials in %.3f s' % (
    n_epochs, n_tr, n_val, n_test, dt))
train_steps = n_epochs * n_tr
print('  if everything were training steps at 4.40 ms: %.3f s; logged rows: %d' % (
    ((n_epochs + 1) * n_tr + (n_epochs + 1) * n_val + n_test) * 4.4e-3, len(exp.rows)))
if os.environ.get('BN_CPROFILE'):
    import cProfile, pstats
    hp['max_n_epochs'] = hp['min_n_epochs'] = 3
    pr = cProfile.Profile(); pr.enable()
    fit(hp, model, gen, Exp(), method='ae')
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
