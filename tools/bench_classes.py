"""Training step of every model class at the headline size (1x128x128, 256 frames, 12 latents; two
labels where the class takes them): ms per step and the host's share of it.
usage: python tools/bench_classes.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.models import AE, ConditionalAE, AEMSP, VAE, ConditionalVAE, BetaTCVAE, PSVAE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.synthetic import base_hparams, make_frames, make_labels

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, DIM = 256, [1, 128, 128]
EXTRA = {'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0, 'beta_tcvae.beta': 5.0,
         'beta_tcvae.beta_anneal_epochs': 0, 'ps_vae.alpha': 1000, 'ps_vae.beta': 5,
         'ps_vae.anneal_epochs': 0, 'max_n_epochs': 10, 'msp.alpha': 1e-4,
         'conditional_encoder': False}
CASES = [('ae', AE, 0, {}), ('ae + batch norm', AE, 0, {'ae_batch_norm': True}),
         ('cond-ae', ConditionalAE, 2, {}), ('cond-ae-msp', AEMSP, 2, {}),
         ('vae', VAE, 0, {}), ('vae + batch norm', VAE, 0, {'ae_batch_norm': True}),
         ('cond-vae', ConditionalVAE, 2, {}), ('beta-tcvae', BetaTCVAE, 0, {}),
         ('ps-vae', PSVAE, 2, {})]
NAMES = {AE: 'ae', ConditionalAE: 'cond-ae', AEMSP: 'cond-ae-msp', VAE: 'vae', ConditionalVAE: 'cond-vae',
         BetaTCVAE: 'beta-tcvae', PSVAE: 'ps-vae'}
x = torch.from_numpy(make_frames(B, DIM, seed=1)).cuda()
for label, cls, n_labels, extra in CASES:
    arch = load_handcrafted_arch(list(DIM), 12, None, check_memory=False)
    hp = base_hparams(arch, NAMES[cls], dict(EXTRA, **extra))
    hp['device'] = 'cuda'
    if n_labels:
        hp['n_labels'] = n_labels
    np.random.seed(0); torch.manual_seed(0)
    m = cls(hp).to('cuda'); m.curr_epoch = 1
    opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
    data = {'images': x[None]}
    if n_labels:
        data['labels'] = torch.from_numpy(make_labels(B, n_labels, seed=2)).cuda()[None]

    def step():
        m.train(); opt.zero_grad(); m.loss(data, dataset=0, accumulate_grad=True); opt.step()
    for _ in range(12): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter(); step(); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%-18s %7.3f ms/step  %8.0f frames/s   host issue %6.3f ms' % (label, dt * 1e3, B / dt, host / steps * 1e3))
    del m, opt
    torch.cuda.empty_cache()
