"""Run by tests/test_gpu_model.py in a subprocess with BN_DIST_FORCE=1: a process group of ONE
rank over RCCL on the GPU, so that the bucketed, overlapped gradient all-reduce issues its real
collectives and stream hand-offs.  With one rank the sum is the identity: parameters after a few
optimizer steps must be bit-identical with and without the reducer.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np   # noqa: E402
import torch         # noqa: E402

from behavenet_amd.fitting import distributed as bdist           # noqa: E402
from behavenet_amd.fitting.optim import FlatAdamAMSGrad          # noqa: E402
from behavenet_amd import hip_functions as hf                    # noqa: E402
from behavenet_amd.models import AE                              # noqa: E402
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch  # noqa: E402
from tests.golden_utils import base_hparams, make_frames         # noqa: E402


def run(with_reducer, n_frames):
    arch = load_handcrafted_arch([1, 32, 32], 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    torch.manual_seed(0)
    model = AE(hp).to('cuda')
    opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-3)
    hf.set_grad_ready_callback(None)
    reducer = bdist.attach_reducer(opt) if with_reducer else None
    x = torch.from_numpy(make_frames(n_frames, [1, 32, 32], seed=4)).cuda()
    early, losses = [], []
    for step in range(4):
        opt.zero_grad()
        losses.append(model.loss({'images': x[None]}, dataset=0, accumulate_grad=True)['loss'])
        if step == 2:
            continue          # a step that is not applied (fit()'s epoch 0): collectives drained
        if reducer is not None:
            early.append(reducer.n_overlapped)
        bdist.reduce_gradients(opt)
        opt.step()
    torch.cuda.synchronize()
    return opt.flat_p.cpu().numpy().copy(), early, losses, \
        (len(reducer.buckets) if reducer else 0)


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ['WORLD_SIZE'] = '1'
    os.environ['RANK'] = '0'
    bdist.init_from_env()
    assert bdist.is_active() and bdist.world_size() == 1
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 210
    p_ref, _, l_ref, _ = run(False, n_frames)
    p_red, early, l_red, n_buckets = run(True, n_frames)
    print(json.dumps({'identical': bool(np.array_equal(p_ref, p_red)),
                      'max_abs_diff': float(np.abs(p_ref - p_red).max()),
                      'losses_identical': l_ref == l_red, 'n_buckets': n_buckets,
                      'overlapped_per_step': early}))
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
