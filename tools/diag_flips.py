"""Counts LeakyReLU sign flips of the HIP decoder forward against the float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd.models import ConditionalVAE, VAE
from behavenet_amd import hip_functions as hf, _hip
from oracle import ref_cpu
from tests.cases import case_hparams, case_data, seeded_build
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mc = sys.argv[2] if len(sys.argv) > 2 else 'cond-vae'
n_lat = 8
extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10, 'conditional_encoder': False}
meta = {'dim': [1, 32, 32], 'n_lat': n_lat, 'model_class': mc, 'extra_hp': extra,
        'n_labels': 4 if mc == 'cond-vae' else 0, 'n_frames': n}
cls = ConditionalVAE if mc == 'cond-vae' else VAE
hip = seeded_build(cls, case_hparams(meta)).to('cuda')
o32 = seeded_build(ref_cpu.build_model, case_hparams(meta))
o64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
g = torch.Generator().manual_seed(3)
kin = n_lat + meta['n_labels']
zin = torch.randn((n, kin), generator=g)
if len(sys.argv) > 3:      # the decoder input of the multi-chunk parity test
    d = case_data(meta)
    eps = torch.randn((200, n_lat), generator=torch.Generator().manual_seed(9)).double()[:n]
    with torch.no_grad():
        mu, logvar, _, _ = o64.encoding(d['images'][0].double(), dataset=0)
    z = mu + eps * torch.exp(logvar)
    zin = (torch.cat((z, d['labels'][0].double()), 1) if mc == 'cond-vae' else z).float()
outs = {'o32': [], 'o64': []}
with torch.no_grad():
    o32.decoding(zin, None, None, dataset=0, taps=outs['o32']); o64.decoding(zin.double(), None, None, dataset=0, taps=outs['o64'])
    dec = hip.decoding
    h = hf.linear(zin.cuda(), dec.FF.weight, dec.FF.bias)
    st = dec.hparams['ae_decoding_starting_dim']
    h = h.view(n, st[0], st[1], st[2])
    params = dec._stack_params(0)
    ff64 = torch.nn.functional.linear(zin.double(), o64.decoding.FF.weight, o64.decoding.FF.bias)
    print('FF out err', (h.cpu().double().view(n, -1) - ff64).abs().max().item() / ff64.abs().max().item())
    for i, layer in enumerate(dec._plan):
        h = hf._fwd(layer, h, params[2 * i].detach(), params[2 * i + 1].detach())
        r64 = outs['o64'][i]; r32 = outs['o32'][i].double()
        # oracle modules apply the crop after the conv: compare only when shapes agree
        ph = h.cpu().double()
        if ph.shape != r64.shape:
            print(i, 'shape', tuple(ph.shape), tuple(r64.shape)); continue
        s = r64.abs().max().item()
        fl_h = ((ph > 0) != (r64 > 0)).sum().item(); fl_c = ((r32 > 0) != (r64 > 0)).sum().item()
        print('layer %d %s: max err hip %.2e cpu32 %.2e | sign flips hip %d cpu32 %d | |pre|<1e-6*max: %d of %d' % (
            i, tuple(ph.shape), (ph - r64).abs().max().item() / s, (r32 - r64).abs().max().item() / s,
            fl_h, fl_c, (r64.abs() < 1e-6 * s).sum().item(), r64.numel()))
# ---- decoder-only backward: loss = sum(x_hat * R)
R = torch.randn((n, 1, 32, 32), generator=g)
def run(model, z, r):
    for p in model.parameters(): p.grad = None
    z = z.clone().requires_grad_(True)
    out = model.decoding(z, None, None, dataset=0)
    (out * r).sum().backward()
    return z.grad, {k: p.grad for k, p in model.decoding.named_parameters()}
dz64, g64 = run(o64, zin.double(), R.double())
dz32, g32 = run(o32, zin, R)
dzh, gh = run(hip, zin.cuda(), R.cuda())
hf.join_side_streams(); torch.cuda.synchronize()
def e(a, b): return (a.cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
print('dz: hip %.2e cpu32 %.2e' % (e(dzh, dz64), e(dz32, dz64)))
for k in g64: print('  %-34s hip %.2e cpu32 %.2e' % (k, e(gh[k], g64[k]), e(g32[k], g64[k])))
