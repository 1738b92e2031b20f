"""Training step of one model class at 1x128x128 / 256 frames (for rocprofv3):
python tools/step_class.py ae|vae [bn] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.models import AE, VAE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.synthetic import base_hparams, make_frames

cls = {'ae': AE, 'vae': VAE}[sys.argv[1]]
bn = len(sys.argv) > 2 and sys.argv[2] == 'bn'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
arch = load_handcrafted_arch([1, 128, 128], 12, None, check_memory=False)
hp = base_hparams(arch, sys.argv[1], {'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10,
                                      'ae_batch_norm': bn})
hp['device'] = 'cuda'
np.random.seed(0); torch.manual_seed(0)
m = cls(hp).to('cuda'); m.curr_epoch = 1
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
data = {'images': torch.from_numpy(make_frames(256, [1, 128, 128], seed=1)).cuda()[None]}


def step():
    m.train(); opt.zero_grad(); m.loss(data, dataset=0, accumulate_grad=True); opt.step()


for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print('%s%s: %.3f ms/step' % (sys.argv[1], ' + batch norm' if bn else '', (time.perf_counter() - t0) / steps * 1e3))
