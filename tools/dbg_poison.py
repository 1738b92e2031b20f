"""Poison the LDS with NaNs before EVERY C-ABI call of a few training steps (device placement):
finds kernels that read LDS they never wrote at the batch sizes the call sees."""
import os, sys, torch, numpy as np, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
from tests import debug_lib
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from tests.golden_utils import base_hparams
lib = _hip.load(); dbg = debug_lib.load()
sink = torch.zeros(1, device='cuda')
calls = []
class Wrapped(object):
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        dbg.bn_debug_poison_lds(sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        calls.append((self.name, a))
        return self.fn(*a)
class LibProxy(object):
    def __getattr__(self, name):
        fn = getattr(lib, name)
        if name.startswith('bn_') and not name.endswith('_bytes') and 'prof' not in name and name not in ('bn_error_string', 'bn_version', 'bn_build_arch', 'bn_convT2d_fwd_sqerr_parts', 'bn_set_force_generic'):
            return Wrapped(name, fn)
        return fn
_hip._lib = LibProxy()
dim = [1, int(os.environ.get('HW', 32)), int(os.environ.get('HW', 32))]
arch = load_handcrafted_arch(list(dim), 4, None, check_memory=False)
hp = base_hparams(arch, 'ae', None); hp['device'] = 'cuda'
torch.manual_seed(0)
model = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4)
names = [(k, p.numel()) for k, p in model.named_parameters() if p.requires_grad]
rng = np.random.default_rng(0)
for n in [int(v) for v in os.environ.get('NS', '5,6,7,200,56').split(',')]:
    x = torch.from_numpy(rng.integers(0, 255, size=(n,) + tuple(dim), dtype=np.uint8).astype(np.float32) / 255).cuda()
    opt.zero_grad()
    calls.clear()
    out = model.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
    torch.cuda.synchronize()
    bad = []
    for (k, cnt), o in zip(names, opt.offsets):
        if not torch.isfinite(opt.flat_g[o:o + cnt]).all(): bad.append(k)
    print('n=%d loss=%s bad grads: %s' % (n, out['loss'], bad))
