"""Whole-model fuzz over the architectures the reference's random search draws:

    python tools/fuzz_archs.py [first_seed] [n_seeds] [C H W] [frames]          (BN_FUZZ_BN=1: with ae_batch_norm)

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
    with_bn = os.environ.get('BN_FUZZ_BN') == '1'
    extra = {'ae_batch_norm': True} if with_bn else None
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
            hip = AE(base_hparams(dict(arch), 'ae', extra)).to('cuda')
            torch.manual_seed(0)
            ora64 = ref_cpu.AE(base_hparams(dict(arch), 'ae', extra)).double()
            ora64.train()
            x = torch.from_numpy(make_frames(n_frames, dim, seed=500 + seed))
            hip.train()
            hip.zero_grad(set_to_none=True)
            with record_branches(hip) as rec:
                lh = hip.loss({'images': x.to('cuda')[None]}, dataset=0, accumulate_grad=True)['loss']
            with BranchReplay(rec) as br:
                l64 = ora64.loss({'images': x.double()[None]}, dataset=0, accumulate_grad=True)['loss']
            # (batch norm over few values amplifies fp32 rounding: pre-activations up to 1e-5 of the layer's maximum
            # may land on the other side of zero)
            br.assert_only_ties(max_rel=3e-5 if with_bn else 2e-6)
            assert abs(lh - l64) <= 1e-5 * abs(l64), (lh, l64)
            if not with_bn:
                grads_close_on_same_branches(hip, ora64, 'seed %d' % seed)
            else:
                # batch norm: statistics over few values amplify fp32 rounding (the golden batch-norm cases' 2e-4), and
                # a conv bias in front of a batch norm has an analytically zero gradient: rounding noise on both sides
                import numpy as np
                from tests.test_gpu_model import _bias_before_batchnorm
                names = {k for k, _ in hip.named_parameters()}
                top = max(float(po.grad.abs().max()) for po in ora64.parameters() if po.grad is not None)
                # how far a float32 CPU run of the same model (same branches) is from float64: deep 'same' architectures
                # end in 1x1 maps, and a 5-frame chunk then normalises over FIVE values per channel -- whatever computes
                # that in float32 is 1e-4 .. 1e-3 away from float64
                torch.manual_seed(0)
                ora32 = ref_cpu.AE(base_hparams(dict(arch), 'ae', extra))
                ora32.train()
                with BranchReplay(rec):
                    ora32.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
                cpu32 = {k: p_.grad for k, p_ in ora32.named_parameters()}
                for (k, ph), (_, po) in zip(hip.named_parameters(), ora64.named_parameters()):
                    if po.grad is None or _bias_before_batchnorm(k, names):
                        continue
                    w = po.grad.numpy()
                    # another analytically-zero class: when the decoder starts from 1x1 maps, a constant added to the
                    # latents of every frame (enc.FF.bias, dec.FF.bias) is removed by the first batch norm's batch mean
                    if np.abs(w).max() < 1e-9 * top and float(ph.grad.abs().max()) < 1e-6 * top:
                        continue
                    err = np.abs(ph.grad.cpu().double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
                    e32 = np.abs(cpu32[k].double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
                    assert err <= max(2e-4, 16 * e32), 'seed %d grad %s: normalised max err %.3e (float32 CPU: %.3e)' % (
                        seed, k, err, e32)
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
