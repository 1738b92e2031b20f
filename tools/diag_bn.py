"""Diagnostic: BN model gradients, HIP vs fp32 oracle vs float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.cases import load_case, case_hparams, case_data, seeded_build
from behavenet_amd.models import AE
from oracle import ref_cpu
name = sys.argv[1] if len(sys.argv) > 1 else 'ae_cfg1_bn_b210'
z, meta = load_case(name)
hip = seeded_build(AE, case_hparams(meta)).to('cuda')
o32 = seeded_build(ref_cpu.build_model, case_hparams(meta))
o64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
d = case_data(meta)
for m in (hip, o32, o64):
    m.train()
hip.loss({k: v.cuda() for k, v in d.items()}, dataset=0, accumulate_grad=True)
o32.loss(d, dataset=0, accumulate_grad=True)
o64.loss({k: v.double() for k, v in d.items()}, dataset=0, accumulate_grad=True)
for (k, ph), (_, p32), (_, p64) in zip(hip.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    if p64.grad is None: continue
    g64 = p64.grad.numpy(); s = max(np.abs(g64).max(), 1e-30)
    eh = np.abs(ph.grad.cpu().double().numpy() - g64).max() / s
    ec = np.abs(p32.grad.double().numpy() - g64).max() / s
    print('%-45s scale %.3e  hip %.2e  cpu32 %.2e' % (k, s, eh, ec))
