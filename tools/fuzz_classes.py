"""Every model class of the path on drawn architectures, two chunks per batch:

    python tools/fuzz_classes.py [first_seed] [n_seeds] [C H W] [frames] [chunk]

for each seed (``get_possible_arch``) and each of vae / beta-tcvae / cond-vae / ps-vae / cond-ae / cond-ae-msp: the loss
dict against the float32 CPU oracle's chunk loop (same eps per chunk) and every parameter gradient against the float64
oracle on the device's LeakyReLU branches -- 2e-5 of the tensor's maximum, or 16 x the float32 CPU oracle's own distance
from float64."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

CLASSES = ('vae', 'beta-tcvae', 'cond-vae', 'ps-vae', 'cond-ae', 'cond-ae-msp')


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    dim = [int(v) for v in sys.argv[3:6]] if len(sys.argv) > 5 else [1, 48, 48]
    n_frames = int(sys.argv[6]) if len(sys.argv) > 6 else 45
    chunk = int(sys.argv[7]) if len(sys.argv) > 7 else 30
    from behavenet_amd.models import vaes as hip_vaes
    from behavenet_amd.models.ae_model_architecture_generator import get_possible_arch
    from behavenet_amd.hostinfo import limit_host_threads
    from oracle import ref_cpu
    from tests.branches import record_branches, BranchReplay
    from tests.cases import seeded_build, EpsReplay
    from tests.golden_utils import base_hparams, make_frames, make_labels
    from tests.test_gpu_model import BUILDERS
    limit_host_threads(cap=32)
    n_lat = 8
    extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10, 'beta_tcvae.beta': 3.0,
             'beta_tcvae.beta_anneal_epochs': 5, 'ps_vae.alpha': 10, 'ps_vae.beta': 5, 'ps_vae.anneal_epochs': 5,
             'msp.alpha': 0.05, 'conditional_encoder': False}
    sizes = [min(chunk, n_frames - b) for b in range(0, n_frames, chunk)]
    bad = 0
    total = 0
    for seed in range(first, first + count):
        arch = get_possible_arch(list(dim), n_lat, arch_seed=seed)
        arch.update(n_input_channels=dim[0], y_pixels=dim[1], x_pixels=dim[2])
        desc = '%s c%s k%s s%s' % (arch['ae_padding_type'], [int(v) for v in arch['ae_encoding_n_channels']],
                                   [int(v) for v in arch['ae_encoding_kernel_size']],
                                   [int(v) for v in arch['ae_encoding_stride_size']])
        for cls in CLASSES:
            total += 1
            n_labels = 4 if cls in ('ps-vae', 'cond-vae', 'cond-ae', 'cond-ae-msp') else 0
            try:
                def hp():
                    h = base_hparams(dict(arch), cls, dict(extra))
                    if n_labels:
                        h['n_labels'] = n_labels
                    return h
                hip = seeded_build(BUILDERS[cls], hp()).to('cuda')
                ora = seeded_build(ref_cpu.build_model, hp())
                ora64 = seeded_build(ref_cpu.build_model, hp()).double()
                x = torch.from_numpy(make_frames(n_frames, dim, seed=300 + seed))
                data_c = {'images': x[None]}
                if n_labels:
                    data_c['labels'] = torch.from_numpy(make_labels(n_frames, n_labels, seed=2))[None]
                data_g = {k: v.to('cuda') for k, v in data_c.items()}
                data64 = {k: v.double() for k, v in data_c.items()}
                g = torch.Generator().manual_seed(9)
                eps = [torch.randn((n, n_lat), generator=g).numpy() for n in sizes]
                for m in (hip, ora, ora64):
                    m.train()
                    m.curr_epoch = 3
                ora.eps_fn = EpsReplay(eps)
                ora64.eps_fn = EpsReplay([e.astype(np.float64) for e in eps])
                hip_vaes.set_eps_provider(EpsReplay(eps, 'cuda'))
                try:
                    hip.zero_grad()
                    with record_branches(hip) as rec:
                        loss_h = hip.loss(data_g, dataset=0, accumulate_grad=True, chunk_size=chunk)
                    with BranchReplay(rec):
                        loss_o = ora.loss(data_c, dataset=0, accumulate_grad=True, chunk_size=chunk)
                    with BranchReplay(rec) as br:
                        ora64.loss(data64, dataset=0, accumulate_grad=True, chunk_size=chunk)
                finally:
                    hip_vaes.set_eps_provider(None)
                br.assert_only_ties()
                assert sorted(loss_h.keys()) == sorted(loss_o.keys()), (sorted(loss_h), sorted(loss_o))
                for k in loss_o:
                    assert abs(loss_h[k] - loss_o[k]) <= 1e-4 * abs(loss_o[k]) + 1e-6, (k, loss_h[k], loss_o[k])
                cpu32 = dict(ora.named_parameters())
                for (k, ph), (_, po) in zip(hip.named_parameters(), ora64.named_parameters()):
                    if po.grad is None:
                        continue
                    w = po.grad.numpy()
                    scale = max(np.abs(w).max(), 1e-30)
                    err = np.abs(ph.grad.cpu().double().numpy() - w).max() / scale
                    e32 = np.abs(cpu32[k].grad.double().numpy() - w).max() / scale
                    assert err <= max(2e-5, 16 * e32), 'grad %s: normalised max err %.3e (float32 CPU: %.3e)' % (k, err, e32)
                print('ok   seed %d %-11s %s' % (seed, cls, desc), flush=True)
            except BaseException as err:                              # noqa: BLE001
                bad += 1
                print('FAIL seed %d %-11s %s: %s' % (seed, cls, desc,
                                                     (str(err).splitlines() or [type(err).__name__])[0][:300]), flush=True)
                torch.cuda.synchronize()
    print('%d class x architecture runs on %s (%d frames in chunks of %d): %d failures' % (total, dim, n_frames, chunk, bad))
    return bad


if __name__ == '__main__':
    sys.exit(min(main(), 255))
