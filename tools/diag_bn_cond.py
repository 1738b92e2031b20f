"""ae_cfg1_bn_b210 (batch norm over 10 values per channel in the second chunk: ill-conditioned):
error of every parameter gradient against the float64 oracle for the fp32 CPU oracle, the HIP
model on the dispatched kernels and the HIP model on the shape-agnostic kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd import _hip
from behavenet_amd.models import AE
from oracle import ref_cpu
from tests.cases import load_case, case_hparams, case_data, seeded_build
z, meta = load_case('ae_cfg1_bn_b210')
data_c = case_data(meta)
data_g = {k: v.cuda() for k, v in data_c.items()}
from tests.branches import record_branches, BranchReplay
def grads_hip(generic):
    """-> gradients, and the float64 oracle's gradients on the same LeakyReLU branches"""
    prev = _hip.set_force_generic(generic)
    try:
        m = seeded_build(AE, case_hparams(meta)).cuda(); m.train(); m.zero_grad()
        with record_branches(m) as rec:
            m.loss(data_g, dataset=0, accumulate_grad=True)
        o = seeded_build(ref_cpu.build_model, case_hparams(meta)).double(); o.train(); o.zero_grad()
        with BranchReplay(rec) as br:
            o.loss({k: v.double() for k, v in data_c.items()}, dataset=0, accumulate_grad=True)
        print('generic' if generic else 'fast', 'branch differences:', len(br.flips), 'of', br.n_elements,
              'max |x|/max', max([f[2] for f in br.flips], default=0.0))
        return ({k: p.grad.cpu().double().numpy() for k, p in m.named_parameters()},
                {k: p.grad.numpy() for k, p in o.named_parameters()})
    finally:
        _hip.set_force_generic(prev)
def grads_ora(dt):
    m = seeded_build(ref_cpu.build_model, case_hparams(meta)).to(dt); m.train(); m.zero_grad()
    m.loss({k: v.to(dt) for k, v in data_c.items()}, dataset=0, accumulate_grad=True)
    return {k: p.grad.double().numpy() for k, p in m.named_parameters()}
g64, g32 = grads_ora(torch.float64), grads_ora(torch.float32)
(gh, gh64), (gg, gg64) = grads_hip(False), grads_hip(True)
print('%-44s %10s %10s %10s %10s %10s' % ('parameter', 'cpu fp32', 'hip fast', 'hip generic', 'fast@own', 'generic@own'))
for k in g64:
    s = max(np.abs(g64[k]).max(), 1e-30)
    print('%-44s %10.2e %10.2e %10.2e %10.2e %10.2e' % (
        k, np.abs(g32[k] - g64[k]).max() / s, np.abs(gh[k] - g64[k]).max() / s, np.abs(gg[k] - g64[k]).max() / s,
        np.abs(gh[k] - gh64[k]).max() / max(np.abs(gh64[k]).max(), 1e-30),
        np.abs(gg[k] - gg64[k]).max() / max(np.abs(gg64[k]).max(), 1e-30)))
