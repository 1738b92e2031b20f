"""AE training step of a json architecture (for rocprofv3): python tools/step_arch.py <arch.json> C H W batch [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.data.synthetic import base_hparams, make_frames

js = sys.argv[1]
C, H, W, B = [int(v) for v in sys.argv[2:6]]
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
arch = load_handcrafted_arch([C, H, W], 12, js, check_memory=False)
hp = base_hparams(arch, 'ae', {})
hp['device'] = 'cuda'
torch.manual_seed(0)
m = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
data = {'images': [torch.from_numpy(make_frames(B, [C, H, W], seed=1)).cuda()]}


def step():
    m.train(); opt.zero_grad(); m.loss(data, dataset=0, accumulate_grad=True); opt.step()


for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print('%s on %dx%dx%d batch %d: %.3f ms/step' % (os.path.basename(js), C, H, W, B, (time.perf_counter() - t0) / steps * 1e3))
