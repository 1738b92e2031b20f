"""Whole-model fuzz over the architectures the reference's random search draws:

    python tools/fuzz_archs.py [first_seed] [n_seeds] [C H W] [frames]

for every seed: ``get_possible_arch`` -> AE on the device and the float64 CPU oracle with the same parameters -> one
``loss(accumulate_grad=True)`` -> the loss to 1e-5 and every parameter gradient to 2e-5 of its maximum on the device's
LeakyReLU branch pattern (the gate of tests/test_gpu_bench_sizes.py).  Prints one line per seed; exit status = failures."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dim = [int(v) for v in sys.argv[3:6]] if len(sys.argv) > 5 else [1, 64, 64]
    n_frames = int(sys.argv[6]) if len(sys.argv) > 6 else 8
    from behavenet_amd.models import AE
    from behavenet_amd.models.ae_model_architecture_generator import get_possible_arch
    from behavenet_amd.hostinfo import limit_host_threads
    from oracle import ref_cpu
    from tests.branches import record_branches, BranchReplay
    from tests.golden_utils import base_hparams, make_frames
    from tests.test_gpu_model import grads_close_on_same_branches
    limit_host_threads(cap=32)
    bad = 0
    for seed in range(first, first + count):
        arch = get_possible_arch(list(dim), 12, arch_seed=seed)
        arch.update(n_input_channels=dim[0], y_pixels=dim[1], x_pixels=dim[2])
        desc = '%s c%s k%s s%s' % (arch['ae_padding_type'], [int(v) for v in arch['ae_encoding_n_channels']],
                                   [int(v) for v in arch['ae_encoding_kernel_size']],
                                   [int(v) for v in arch['ae_encoding_stride_size']])
        n_par = sum(int(c) for c in arch['ae_encoding_n_channels'])
        t0 = time.time()
        try:
            torch.manual_seed(0)
            hip = AE(base_hparams(dict(arch), 'ae')).to('cuda')
            torch.manual_seed(0)
            ora64 = ref_cpu.AE(base_hparams(dict(arch), 'ae')).double()
            x = torch.from_numpy(make_frames(n_frames, dim, seed=500 + seed))
            hip.train()
            hip.zero_grad(set_to_none=True)
            with record_branches(hip) as rec:
                lh = hip.loss({'images': x.to('cuda')[None]}, dataset=0, accumulate_grad=True)['loss']
            with BranchReplay(rec) as br:
                l64 = ora64.loss({'images': x.double()[None]}, dataset=0, accumulate_grad=True)['loss']
            br.assert_only_ties()
            assert abs(lh - l64) <= 1e-5 * abs(l64), (lh, l64)
            grads_close_on_same_branches(hip, ora64, 'seed %d' % seed)
            print('ok   seed %d  %s  (%.1f s)' % (seed, desc, time.time() - t0), flush=True)
        except BaseException as err:                                  # noqa: BLE001
            bad += 1
            print('FAIL seed %d  %s: %s' % (seed, desc, (str(err).splitlines() or [type(err).__name__])[0][:300]),
                  flush=True)
            torch.cuda.synchronize()
        del n_par
    print('%d architectures on %s, %d failures' % (count, dim, bad))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
