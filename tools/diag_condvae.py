import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd.models import ConditionalVAE
from behavenet_amd.models import vaes as hip_vaes
from oracle import ref_cpu
from tests.cases import case_hparams, case_data, seeded_build, EpsReplay
n_lat = 8
extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10, 'conditional_encoder': False}
meta = {'dim': [1, 32, 32], 'n_lat': n_lat, 'model_class': 'cond-vae', 'extra_hp': extra, 'n_labels': 4,
        'n_frames': int(sys.argv[1]) if len(sys.argv) > 1 else 210}
hip = seeded_build(ConditionalVAE, case_hparams(meta)).to('cuda')
o32 = seeded_build(ref_cpu.build_model, case_hparams(meta))
o64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
d = case_data(meta)
g = torch.Generator().manual_seed(9)
sizes = [200, meta['n_frames'] - 200] if meta['n_frames'] > 200 else [meta['n_frames']]
eps = [torch.randn((n, n_lat), generator=g).numpy() for n in sizes]
for m in (hip, o32, o64):
    m.train(); m.curr_epoch = 3
o32.eps_fn = EpsReplay(eps); o64.eps_fn = EpsReplay([e.astype(np.float64) for e in eps])
hip_vaes.set_eps_provider(EpsReplay(eps, 'cuda'))
l32 = o32.loss(d, dataset=0, accumulate_grad=True)
l64 = o64.loss({k: v.double() for k, v in d.items()}, dataset=0, accumulate_grad=True)
lh = hip.loss({k: v.cuda() for k, v in d.items()}, dataset=0, accumulate_grad=True)
print(lh, l32)
for (k, ph), (_, p32), (_, p64) in zip(hip.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    w = p64.grad.numpy(); s = max(np.abs(w).max(), 1e-30)
    eh = np.abs(ph.grad.cpu().double().numpy() - w).max() / s
    ec = np.abs(p32.grad.double().numpy() - w).max() / s
    if True: print('  %-40s hip %.2e  cpu32 %.2e' % (k, eh, ec))
hp = dict(hip.named_parameters()); p6 = dict(o64.named_parameters())
for k in ['decoding.decoder.convtranspose3.bias', 'decoding.FF.bias']:
    a = hp[k].grad.cpu().double().numpy().ravel(); b = p6[k].grad.numpy().ravel()
    print(k, '\n hip', a[:8], '\n f64', b[:8], '\n ratio', (a / b)[:8])
a = hp['decoding.FF.weight'].grad.cpu().double().numpy(); b = p6['decoding.FF.weight'].grad.numpy()
print('FF.weight err per column', np.abs(a - b).max(0) / np.abs(b).max())
