"""How long the float64 CPU oracle takes on this host as a function of torch's thread count
(the GPU suite's wall time is mostly this; tests/conftest.py picks its thread count from it)."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from behavenet_amd.data.synthetic import base_hparams, make_frames  # noqa: E402


def run(dim, n_lat, batch, dtype):
    arch = load_handcrafted_arch(list(dim), n_lat, None, check_memory=False)
    torch.manual_seed(0)
    m = ref_cpu.AE(base_hparams(arch, 'ae')).to(dtype)
    x = torch.from_numpy(make_frames(batch, dim, seed=1)).to(dtype)
    t0 = time.perf_counter()
    m.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
    return time.perf_counter() - t0


if __name__ == '__main__':
    print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)),
          'default threads', torch.get_num_threads())
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
        torch.set_num_threads(nt)
        a = run([1, 32, 32], 8, 210, torch.float64)
        b = run([1, 128, 128], 12, 64, torch.float64)
        c = run([1, 128, 128], 12, 64, torch.float32)
        print('threads %3d: cfg1 x210 f64 %.2fs | cfg2 x64 f64 %.2fs | cfg2 x64 f32 %.2fs' % (nt, a, b, c),
              flush=True)
