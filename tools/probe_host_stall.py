"""What a host hiccup costs the training step: 20 AE steps (headline shape) with one time.sleep(stall) at step 10,
plain loss dicts against lazy ones (hip_functions.set_lazy_losses).  python tools/probe_host_stall.py [stall_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from behavenet_amd import hip_functions as hf
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.synthetic import base_hparams, make_frames

stall = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.008
arch = load_handcrafted_arch([1, 128, 128], 12, None, check_memory=False)
hp = base_hparams(arch, 'ae', {}); hp['device'] = 'cuda'
torch.manual_seed(0)
m = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
data = {'images': [torch.from_numpy(make_frames(256, [1, 128, 128], seed=1)).cuda()]}


def step():
    m.train(); opt.zero_grad(); out = m.loss(data, dataset=0, accumulate_grad=True); opt.step(); return out


for lazy in (False, True, False, True):
    hf.set_lazy_losses(lazy)
    for hiccup in (False, True):
        for _ in range(30): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            if hiccup and i == 10:
                time.sleep(stall)
            last = step()
        torch.cuda.synchronize()
        print('lazy %-5s hiccup %-5s %.3f ms/step  (loss %.6f)' % (lazy, hiccup, (time.perf_counter() - t0) / 20 * 1e3, last['loss']))
