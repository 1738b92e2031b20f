"""Frame-sharded data parallelism on drawn architectures (one GPU, ranks emulated one after another):

    python tools/fuzz_shards.py [first_seed] [n_seeds] [C H W] [frames] [chunk] [R ...]

for every seed and every R: the sum over the R emulated ranks of the frame-sharded loss and gradients (every rank takes
its slice of every chunk, chunk terms normalised globally -- reference aes.py:751-771) against the float64 oracle's
single-device step on the branch pattern assembled from the ranks' passes; the gate of
tests/test_gpu_sharding.py::test_ae_frame_shards_add_up_to_the_single_device_step.  Small slices (a few frames per rank)
put every layer on the small-batch forms of its kernels (reduction splits, split epilogues)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dim = [int(v) for v in sys.argv[3:6]] if len(sys.argv) > 5 else [1, 64, 64]
    batch = int(sys.argv[6]) if len(sys.argv) > 6 else 67
    chunk = int(sys.argv[7]) if len(sys.argv) > 7 else 40
    ranks = [int(v) for v in sys.argv[8:]] or [2, 3, 8]
    from behavenet_amd.models import AE
    from behavenet_amd.models.ae_model_architecture_generator import get_possible_arch
    from behavenet_amd.hostinfo import limit_host_threads
    from oracle import ref_cpu
    from tests.branches import BranchReplay
    from tests.golden_utils import base_hparams, make_frames
    from tests.test_gpu_sharding import _sum_over_emulated_ranks, _grads_match_oracle_on_branches
    limit_host_threads(cap=32)
    bad = 0
    for seed in range(first, first + count):
        arch = get_possible_arch(list(dim), 12, arch_seed=seed)
        arch.update(n_input_channels=dim[0], y_pixels=dim[1], x_pixels=dim[2])
        desc = '%s c%s k%s s%s' % (arch['ae_padding_type'], [int(v) for v in arch['ae_encoding_n_channels']],
                                   [int(v) for v in arch['ae_encoding_kernel_size']],
                                   [int(v) for v in arch['ae_encoding_stride_size']])
        try:
            torch.manual_seed(0)
            model = AE(base_hparams(dict(arch), 'ae')).to('cuda')
            x = torch.from_numpy(make_frames(batch, dim, seed=900 + seed))
            data = {'images': x.to('cuda')[None]}
            model.zero_grad(set_to_none=True)
            whole = model.loss(data, dataset=0, accumulate_grad=True, chunk_size=chunk)
            for R in ranks:
                shard_loss, g_sum, pattern = _sum_over_emulated_ranks(model, data, R, chunk)
                assert abs(shard_loss['loss'] - whole['loss']) <= 1e-6 * abs(whole['loss']), (R, shard_loss, whole)
                torch.manual_seed(0)
                ora64 = ref_cpu.AE(base_hparams(dict(arch), 'ae')).double()
                with BranchReplay(pattern) as br:
                    l64 = ora64.loss({'images': x.double()[None]}, dataset=0, accumulate_grad=True, chunk_size=chunk)
                br.assert_only_ties()
                assert abs(shard_loss['loss'] - l64['loss']) <= 1e-5 * abs(l64['loss'])
                _grads_match_oracle_on_branches(g_sum, ora64, 'seed %d R=%d' % (seed, R))
            print('ok   seed %d  %s  R=%s' % (seed, desc, ranks), flush=True)
        except BaseException as err:                                  # noqa: BLE001
            bad += 1
            print('FAIL seed %d  %s: %s' % (seed, desc, (str(err).splitlines() or [type(err).__name__])[0][:300]),
                  flush=True)
            torch.cuda.synchronize()
    print('%d architectures on %s, %d frames in chunks of %d, R in %s: %d failures' % (count, dim, batch, chunk, ranks, bad))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
