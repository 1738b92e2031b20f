"""Whole-model fuzz over the architectures the reference's random search draws:

    python tools/fuzz_archs.py [first_seed] [n_seeds] [C H W] [frames]          (BN_FUZZ_BN=1: with ae_batch_norm;
                                                                                  BN_FUZZ_POOL=1: max-pooling architectures)

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
    from behavenet_amd.models.ae_model_architecture_generator import (
        get_possible_arch, get_encoding_conv_block, get_decoding_conv_block, default_search_options)
    import numpy as np
    from behavenet_amd.hostinfo import limit_host_threads
    from oracle import ref_cpu
    from tests.branches import record_branches, BranchReplay
    from tests.golden_utils import base_hparams, make_frames
    from tests.test_gpu_model import grads_close_on_same_branches
    limit_host_threads(cap=32)
    with_bn = os.environ.get('BN_FUZZ_BN') == '1'
    pooled = os.environ.get('BN_FUZZ_POOL') == '1'
    skipped = 0
    extra = {'ae_batch_norm': True} if with_bn else None
    bad = 0
    for seed in range(first, first + count):
        if pooled:
            # the search itself never draws max pooling (ref :117-119, commented out); a handcrafted json may: the block
            # generator with ae_network_type = 'max_pooling' (conv k x k stride 1 -> 2x2 pooling), 'same' or 'valid'
            np.random.seed(seed)
            arch = {'ae_input_dim': list(dim), 'model_type': 'conv', 'n_ae_latents': 12, 'ae_decoding_last_FF_layer': 0,
                    'ae_batch_norm': 0, 'ae_batch_norm_momentum': None, 'ae_network_type': 'max_pooling',
                    'ae_padding_type': ('valid', 'same')[np.random.randint(2)]}
            opts = default_search_options()
            opts['possible_n_channels'] = np.asarray([16, 32, 64, 128])
            arch = get_decoding_conv_block(get_encoding_conv_block(arch, opts))
            if not arch['ae_encoding_n_channels']:
                continue
        else:
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
            if pooled:
                # the float64 oracle replays LeakyReLU branches, not pooling winners: two window values within fp32
                # rounding of each other route the gradient elsewhere -- a tie of its own kind, skipped here
                with torch.no_grad():
                    _, idx_h, _ = hip.encoding(x.to('cuda'), dataset=0)
                    _, idx_o, _ = ora64.encoding(x.double(), dataset=0)
                keys = sorted(idx_h.keys()) if isinstance(idx_h, dict) else range(len(idx_h))
                if any(int((idx_h[k].cpu().long() != idx_o[k].long()).sum()) for k in keys):
                    skipped += 1
                    print('tie  seed %d  %s: pooling winners differ (values within rounding)' % (seed, desc), flush=True)
                    continue
            br.assert_only_ties(max_rel=3e-5 if with_bn else 2e-6)
            assert abs(lh - l64) <= 1e-5 * abs(l64), (lh, l64)
            # gate per tensor: 2e-5 of its maximum (2e-4 with batch norm), or 16 x what a float32 CPU run of the same
            # model on the same branches is away from float64 -- a scalar bias gradient that is the sum of 25 k terms of
            # both signs, or batch-norm statistics over five values (deep 'same' architectures end in 1x1 maps and a
            # 5-frame chunk), are 1e-4 .. 1e-3 away from float64 in ANY float32 implementation
            import numpy as np
            from tests.test_gpu_model import _bias_before_batchnorm
            names = {k for k, _ in hip.named_parameters()}
            top = max(float(po.grad.abs().max()) for po in ora64.parameters() if po.grad is not None)
            torch.manual_seed(0)
            ora32 = ref_cpu.AE(base_hparams(dict(arch), 'ae', extra))
            ora32.train()
            with BranchReplay(rec):
                ora32.loss({'images': x[None]}, dataset=0, accumulate_grad=True)
            cpu32 = {k: p_.grad for k, p_ in ora32.named_parameters()}
            tol = 2e-4 if with_bn else 2e-5
            for (k, ph), (_, po) in zip(hip.named_parameters(), ora64.named_parameters()):
                if po.grad is None or (with_bn and _bias_before_batchnorm(k, names)):
                    continue
                w = po.grad.numpy()
                # analytically zero gradients (a conv bias in front of a batch norm; enc.FF.bias / dec.FF.bias when the
                # decoder's first batch norm sees 1x1 maps): rounding noise on both sides
                if with_bn and np.abs(w).max() < 1e-9 * top and float(ph.grad.abs().max()) < 1e-6 * top:
                    continue
                err = np.abs(ph.grad.cpu().double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
                e32 = np.abs(cpu32[k].double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
                assert err <= max(tol, 16 * e32), 'seed %d grad %s: normalised max err %.3e (float32 CPU: %.3e)' % (
                    seed, k, err, e32)
            print('ok   seed %d  %s  (%.1f s)' % (seed, desc, time.time() - t0), flush=True)
        except BaseException as err:                                  # noqa: BLE001
            bad += 1
            print('FAIL seed %d  %s: %s' % (seed, desc, (str(err).splitlines() or [type(err).__name__])[0][:300]),
                  flush=True)
            torch.cuda.synchronize()
        del n_par
    print('%d architectures on %s, %d failures%s' % (count, dim, bad, ', %d skipped for pooling ties' % skipped if pooled else ''))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
