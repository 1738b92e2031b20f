import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd.models import ConditionalVAE
from behavenet_amd.models import vaes as hip_vaes
from behavenet_amd import hip_functions as hf
from oracle import ref_cpu
from tests.cases import case_hparams, case_data, seeded_build, EpsReplay
n_lat = 8; n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10, 'conditional_encoder': False}
meta = {'dim': [1, 32, 32], 'n_lat': n_lat, 'model_class': 'cond-vae', 'extra_hp': extra, 'n_labels': 4, 'n_frames': n}
hip = seeded_build(ConditionalVAE, case_hparams(meta)).to('cuda')
o32 = seeded_build(ref_cpu.build_model, case_hparams(meta))
o64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
d = case_data(meta)
eps = [torch.randn((n, n_lat), generator=torch.Generator().manual_seed(9)).numpy()]
for m in (hip, o32, o64): m.train()
x = d['images'][0]; y = d['labels'][0]
R = torch.randn(x.shape, generator=torch.Generator().manual_seed(4))
mode = sys.argv[2] if len(sys.argv) > 2 else 'lin'
def run(model, x, y, r, e):
    if model is hip: hip_vaes.set_eps_provider(EpsReplay(e, 'cuda'))
    else: model.eps_fn = EpsReplay(e)
    for p in model.parameters(): p.grad = None
    xh, z, mu, lv = model.forward(x, dataset=0, labels=y, use_mean=False)
    if mode == 'lin': l = (xh * r).sum()
    else: l = ((xh - x) ** 2).sum() * 0.5
    l.backward()
    return xh.detach(), z.detach(), {k: p.grad for k, p in model.named_parameters()}
xh64, z64, g64 = run(o64, x.double(), y.double(), R.double(), [e.astype(np.float64) for e in eps])
xh32, z32, g32 = run(o32, x, y, R, eps)
xhh, zh, gh = run(hip, x.cuda(), y.cuda(), R.cuda(), eps)
hf.join_side_streams(); torch.cuda.synchronize()
def e(a, b): return (a.cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
print('z: hip %.2e cpu32 %.2e   xhat: hip %.2e cpu32 %.2e' % (e(zh, z64), e(z32, z64), e(xhh, xh64), e(xh32, xh64)))
for k in g64: print('  %-40s hip %.2e cpu32 %.2e' % (k, e(gh[k], g64[k]), e(g32[k], g64[k])))
from behavenet_amd import _hip
taps64, taps32 = [], []
with torch.no_grad():
    o64.decoding(torch.cat((z64, y.double()), 1), None, None, dataset=0, taps=taps64)
    o32.decoding(torch.cat((z32, y), 1), None, None, dataset=0, taps=taps32)
    dec = hip.decoding
    h = hf.linear(torch.cat((zh, y.cuda()), 1), dec.FF.weight, dec.FF.bias)
    st = dec.hparams['ae_decoding_starting_dim']; h = h.view(n, st[0], st[1], st[2])
    params = dec._stack_params(0)
    for i, layer in enumerate(dec._plan):
        h = hf._fwd(layer, h, params[2 * i].detach(), params[2 * i + 1].detach())
        ph = h.cpu().double(); r64 = taps64[i]; r32 = taps32[i].double()
        bad = ((ph > 0) != (r64 > 0))
        print('layer %d flips hip %d cpu32 %d' % (i, bad.sum().item(), ((r32 > 0) != (r64 > 0)).sum().item()),
              'values f64', r64[bad][:4].tolist(), 'hip', ph[bad][:4].tolist(), 'max', r64.abs().max().item())
