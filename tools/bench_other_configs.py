"""Secondary measurements (not the headline): cfg4 PS-VAE training and cfg5 encode-only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd.models import AE, PSVAE
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from tests.golden_utils import base_hparams, make_frames, make_labels

def timed(fn, warm, steps):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

# cfg4: PS-VAE, 2x128x128, 16 latents, 4 labels, batch 256
arch = load_handcrafted_arch([2, 128, 128], 16, None, check_memory=False)
hp = base_hparams(arch, 'ps-vae', {'ps_vae.alpha': 1000, 'ps_vae.beta': 5, 'ps_vae.anneal_epochs': 100,
                                   'max_n_epochs': 200})
hp['n_labels'] = 4
np.random.seed(0); torch.manual_seed(0)
m = PSVAE(hp).to('cuda'); m.curr_epoch = 3
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
x = torch.from_numpy(make_frames(256, [2, 128, 128], seed=1)).cuda()
y = torch.from_numpy(make_labels(256, 4, seed=2)).cuda()
data = {'images': x[None], 'labels': y[None]}
def step():
    m.train(); opt.zero_grad(); m.loss(data, dataset=0, accumulate_grad=True); opt.step()
t = timed(step, 30, 30)
print('cfg4 PS-VAE training (2x128x128, 16 latents, 4 labels, batch 256): %.2f ms/step, %.0f frames/s' % (t * 1e3, 256 / t))

# cfg5: encode-only, 1x128x128, 12 latents, 256-frame trials (uint8 resident -> float on device)
from behavenet_amd import _hip
arch = load_handcrafted_arch([1, 128, 128], 12, None, check_memory=False)
hp = base_hparams(arch, 'ae', None)
torch.manual_seed(0); ae = AE(hp).to('cuda'); ae.eval()
xu = torch.randint(0, 255, (256, 1, 128, 128), dtype=torch.uint8, device='cuda')
def enc():
    with torch.no_grad():
        ae.encoding(_hip.u8_to_unit_float(xu), dataset=0)
t = timed(enc, 20, 50)
print('cfg5 encode-only (uint8 trial of 256 frames -> 12 latents): %.3f ms/trial, %.0f frames/s' % (t * 1e3, 256 / t))
