"""Every conv call of one AE training step (json architecture) against the shape-agnostic kernels on the SAME inputs,
then the step's gradients fast vs generic vs fast again (run-to-run):  python tools/diag_fast_vs_generic.py <arch.json> <frames>
(how the 3e-3 gradient difference of a max-pooling architecture was traced to ONE LeakyReLU sign flip, round 4)."""
import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from behavenet_amd import _hip
from behavenet_amd import hip_functions
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.data.synthetic import base_hparams
js = sys.argv[1]; n = int(sys.argv[2])
arch = load_handcrafted_arch([1, 128, 128], 12, js, check_memory=False)
torch.manual_seed(0); hip = AE(base_hparams(arch, 'ae', None)).to('cuda')
x = torch.rand((n, 1, 128, 128), generator=torch.Generator().manual_seed(31)).cuda()

def wrap(name):
    orig = getattr(_hip, name)
    def f(*a, **k):
        if 'bwd_weight' in name:
            x_in, dy, dw, db, geom, acc = a[:6]
            dw0 = dw.clone(); db0 = db.clone() if db is not None else None
            xc, dyc = x_in.clone(), dy.clone()
            out = orig(*a, **k)
            torch.cuda.synchronize()
            dwf = dw.clone(); dbf = db.clone() if db is not None else None
            dw.copy_(dw0)
            if db is not None: db.copy_(db0)
            prev = _hip.set_force_generic(True)
            orig(*a, **k)
            _hip.set_force_generic(prev)
            torch.cuda.synchronize()
            print(name, geom, 'acc', acc, 'dw diff %.2e' % float((dwf.double() - dw.double()).abs().max() / dw.double().abs().max()),
                  'db diff %.2e' % (float((dbf.double() - db.double()).abs().max() / db.double().abs().max()) if db is not None else -1),
                  'inputs changed', bool((xc != x_in).any()), bool((dyc != dy).any()), flush=True)
            return out
        out = orig(*a, **k)
        torch.cuda.synchronize()
        prev = _hip.set_force_generic(True)
        ref = orig(*a, **k)
        _hip.set_force_generic(prev)
        o = out[0] if isinstance(out, tuple) else out
        r = ref[0] if isinstance(ref, tuple) else ref
        geom = [v for v in a if isinstance(v, tuple)]
        print(name, geom, 'sign flips', int(((o > 0) != (r > 0)).sum()), 'zeros', int((o == 0).sum()), 'diff %.2e' % float((o.double() - r.double()).abs().max() / r.double().abs().max()),
              'ptrs', [hex(v.data_ptr() % 256) for v in a if torch.is_tensor(v)], flush=True)
        return out
    setattr(_hip, name, f)
for nm in ('conv2d_fwd', 'conv2d_bwd_data', 'convT2d_fwd', 'convT2d_bwd_data', 'conv2d_bwd_weight', 'convT2d_bwd_weight'):
    wrap(nm)
hip.train(); hip.zero_grad(set_to_none=True)
l = hip.loss({'images': x[None]}, dataset=0, accumulate_grad=True)['loss']
torch.cuda.synchronize()
gw = {k: p.grad.detach().clone() for k, p in hip.named_parameters() if p.grad is not None}
for nm in ('conv2d_fwd', 'conv2d_bwd_data', 'convT2d_fwd', 'convT2d_bwd_data', 'conv2d_bwd_weight', 'convT2d_bwd_weight'):
    pass
import importlib
def run(gen):
    prev = _hip.set_force_generic(gen)
    hip.zero_grad(set_to_none=True)
    hip.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
    torch.cuda.synchronize()
    _hip.set_force_generic(prev)
    return {k: p.grad.detach().clone() for k, p in hip.named_parameters() if p.grad is not None}
print('--- wrapped-fast vs generic, fast vs generic, fast vs fast')
import builtins
_print = builtins.print
builtins.print = lambda *a, **k: None
gg = run(True); gf1 = run(False); gf2 = run(False)
builtins.print = _print
for k in gg:
    d = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    print('%-42s %.2e %.2e %.2e' % (k, d(gw[k], gg[k]), d(gf1[k], gg[k]), d(gf1[k], gf2[k])))
