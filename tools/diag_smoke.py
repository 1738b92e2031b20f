import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from behavenet_amd import _hip, hip_functions as hf
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from oracle import ref_cpu
from tests.golden_utils import base_hparams, make_frames
dim = [1, 128, 128]; n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
arch = load_handcrafted_arch(list(dim), 12, None, check_memory=False)
torch.manual_seed(0); hip = AE(base_hparams(arch, 'ae')).to('cuda:0')
torch.manual_seed(0); ora = ref_cpu.AE(base_hparams(dict(arch), 'ae'))
x = torch.from_numpy(make_frames(n, dim, seed=7))
ora.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
opt = FlatAdamAMSGrad(hip.get_parameters(), lr=1e-4)
for mode in ('default', 'nohoist', 'noside'):
    if mode == 'nohoist': hip.encoding.prepare_first_layer = lambda *a, **k: None
    if mode == 'noside': hf._use_side_stream = False
    opt.zero_grad()
    hip.loss({'images': x.cuda()[None]}, dataset=0, accumulate_grad=True)
    torch.cuda.synchronize()
    print('==', mode)
    for (k, ph), (_, po) in zip(hip.named_parameters(), ora.named_parameters()):
        g, w = ph.grad.cpu().double().numpy(), po.grad.double().numpy()
        err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)
        if err > 1e-5: print('  %-40s %.2e' % (k, err))
