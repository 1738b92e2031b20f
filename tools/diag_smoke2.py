import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd import _hip, hip_functions as hf
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from oracle import ref_cpu
from tests.golden_utils import base_hparams, make_frames
dim = [1, 128, 128]; n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
arch = load_handcrafted_arch(list(dim), 12, None, check_memory=False)
torch.manual_seed(0); hip = AE(base_hparams(arch, 'ae')).to('cuda:0')
torch.manual_seed(0); o32 = ref_cpu.AE(base_hparams(dict(arch), 'ae'))
torch.manual_seed(0); o64 = ref_cpu.AE(base_hparams(dict(arch), 'ae')).double()
x = torch.from_numpy(make_frames(n, dim, seed=7))
o32.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
o64.loss({'images': x.double()[None]}, dataset=0, accumulate_grad=True)
if len(sys.argv) > 2: _hip.set_force_generic(True)
hip.loss({'images': x.cuda()[None]}, dataset=0, accumulate_grad=True)
torch.cuda.synchronize()
for (k, ph), (_, p32), (_, p64) in zip(hip.named_parameters(), o32.named_parameters(), o64.named_parameters()):
    w = p64.grad.numpy(); s = max(np.abs(w).max(), 1e-30)
    eh = np.abs(ph.grad.cpu().double().numpy() - w).max() / s
    ec = np.abs(p32.grad.double().numpy() - w).max() / s
    if max(eh, ec) > 1e-5: print('  %-40s hip %.2e  cpu32 %.2e' % (k, eh, ec))
