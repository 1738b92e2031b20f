"""PS-VAE training step (BASELINE configs[3]) on its own: wall time per step and, under
rocprofv3 --kernel-trace --stats, the per-kernel breakdown.  usage: python tools/bench_psvae.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.models import PSVAE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.synthetic import base_hparams, make_frames, make_labels

B = 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dim4 = [2, 128, 128]
arch = load_handcrafted_arch(list(dim4), 16, None, check_memory=False)
hp = base_hparams(arch, 'ps-vae', {'ps_vae.alpha': 1000, 'ps_vae.beta': 5,
                                   'ps_vae.anneal_epochs': 100, 'max_n_epochs': 200})
hp['n_labels'] = 4
hp['device'] = 'cuda'
np.random.seed(0); torch.manual_seed(0)
m = PSVAE(hp).to('cuda'); m.curr_epoch = 3
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
data = {'images': torch.from_numpy(make_frames(B, dim4, seed=1)).cuda()[None],
        'labels': torch.from_numpy(make_labels(B, 4, seed=2)).cuda()[None]}

def step():
    m.train(); opt.zero_grad(); m.loss(data, dataset=0, accumulate_grad=True); opt.step()

for _ in range(30): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
host = 0.0
for _ in range(steps):
    h0 = time.perf_counter(); step(); host += time.perf_counter() - h0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print('ps-vae: %.3f ms/step (%.1f frames/s); host issue time %.3f ms/step' % (dt * 1e3, B / dt, host / steps * 1e3))
if os.environ.get('BN_CPROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(steps): step()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)
