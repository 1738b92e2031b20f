"""AE training step at an arbitrary frame shape / batch size (for rocprofv3 --kernel-trace --stats):
    python tools/step_shape.py C H W batch [steps] [rank world]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.data.synthetic import base_hparams, make_frames

C, H, W, B = [int(v) for v in sys.argv[1:5]]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
shard = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else None
arch = load_handcrafted_arch([C, H, W], 12, None, check_memory=False)
hp = base_hparams(arch, 'ae', {})
hp['device'] = 'cuda'
torch.manual_seed(0)
m = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
data = {'images': [torch.from_numpy(make_frames(B, [C, H, W], seed=1)).cuda()]}
if shard:
    bdist.set_shard_mode('frames')


def step():
    m.train(); opt.zero_grad()
    if shard:
        with bdist.emulate_rank(*shard):
            m.loss(data, dataset=0, accumulate_grad=True)
    else:
        m.loss(data, dataset=0, accumulate_grad=True)
    opt.step()


for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print('%dx%dx%d batch %d%s: %.3f ms/step' % (C, H, W, B, ' rank %d of %d' % shard if shard else '',
                                            (time.perf_counter() - t0) / steps * 1e3))
