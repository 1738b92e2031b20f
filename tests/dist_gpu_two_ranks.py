"""One rank of a 2-process data-parallel run on ONE GPU (gloo rendezvous, collectives staged
through the host): launched twice by tests/test_gpu_sharding.py with RANK=0/1.  Also imported by
that test for the single-process reference of the same cases."""

import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator  # noqa: E402
from behavenet_amd.fitting import distributed as bdist  # noqa: E402
from behavenet_amd.fitting.optim import FlatAdamAMSGrad  # noqa: E402
from behavenet_amd.fitting.training import fit  # noqa: E402
from behavenet_amd.models import AE, BetaTCVAE, PSVAE, VAE  # noqa: E402
from behavenet_amd.models import vaes as hip_vaes  # noqa: E402
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch  # noqa: E402
from tests.golden_utils import base_hparams, make_frames, make_labels  # noqa: E402

DEV = 'cuda'
DIM = [1, 32, 32]


class _Eps(object):
    """Deterministic eps per call index (the same on every rank and in the reference run)."""

    def __init__(self):
        self.i = 0

    def __call__(self, like):
        g = torch.Generator().manual_seed(1000 + self.i)
        self.i += 1
        return torch.randn(like.shape, generator=g).to(device=like.device, dtype=like.dtype)


def build_case(case):
    """-> (model on the GPU, data dict, loss kwargs): batch 44 in chunks of 30 + 14."""
    arch = load_handcrafted_arch(list(DIM), 8, None, check_memory=False)
    x = torch.from_numpy(make_frames(44, DIM, seed=8)).to(DEV)
    data = {'images': x[None]}
    kw = {'chunk_size': 30}
    torch.manual_seed(0)
    np.random.seed(0)
    if case == 'ae_bn':
        model = AE(base_hparams(arch, 'ae', {'ae_batch_norm': True}))
    elif case == 'vae_bn':
        model = VAE(base_hparams(arch, 'vae', {'ae_batch_norm': True, 'vae.beta': 2.0,
                                               'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10}))
        hip_vaes.set_eps_provider(_Eps())
    elif case in ('psvae', 'psvae_bn'):
        hp = base_hparams(arch, 'ps-vae', {'ps_vae.alpha': 10.0, 'ps_vae.beta': 3.0,
                                           'ps_vae.anneal_epochs': 0, 'max_n_epochs': 10,
                                           'ae_batch_norm': case == 'psvae_bn'})
        hp['n_labels'] = 2
        model = PSVAE(hp)
        data['labels'] = torch.from_numpy(make_labels(44, 2, seed=2)).to(DEV)[None]
        hip_vaes.set_eps_provider(_Eps())
    elif case == 'betatc':
        hp = base_hparams(arch, 'beta-tcvae', {'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0,
                                               'beta_tcvae.beta': 4.0,
                                               'beta_tcvae.beta_anneal_epochs': 0,
                                               'max_n_epochs': 10})
        model = BetaTCVAE(hp)
        hip_vaes.set_eps_provider(_Eps())
    elif case == 'aemsp':
        from behavenet_amd.models import AEMSP
        hp = base_hparams(arch, 'cond-ae-msp', {'msp.alpha': 0.05, 'conditional_encoder': False})
        hp['n_labels'] = 4
        model = AEMSP(hp)
        data['labels'] = torch.from_numpy(make_labels(44, 4, seed=2)).to(DEV)[None]
    else:
        raise ValueError(case)
    model = model.to(DEV)
    model.train()
    return model, data, kw


def build_oracle(case, dtype=torch.float64):
    """The CPU oracle of the same case (same seeds => same parameters, same eps per chunk)."""
    from oracle import ref_cpu
    arch = load_handcrafted_arch(list(DIM), 8, None, check_memory=False)
    data = {'images': torch.from_numpy(make_frames(44, DIM, seed=8)).to(dtype)[None]}
    torch.manual_seed(0)
    np.random.seed(0)
    if case == 'ae_bn':
        model = ref_cpu.AE(base_hparams(arch, 'ae', {'ae_batch_norm': True}))
    elif case == 'vae_bn':
        model = ref_cpu.VAE(base_hparams(arch, 'vae', {'ae_batch_norm': True, 'vae.beta': 2.0,
                                                       'vae.beta_anneal_epochs': 0,
                                                       'max_n_epochs': 10}))
        model.eps_fn = _Eps()
    elif case in ('psvae', 'psvae_bn'):
        hp = base_hparams(arch, 'ps-vae', {'ps_vae.alpha': 10.0, 'ps_vae.beta': 3.0,
                                           'ps_vae.anneal_epochs': 0, 'max_n_epochs': 10,
                                           'ae_batch_norm': case == 'psvae_bn'})
        hp['n_labels'] = 2
        model = ref_cpu.PSVAE(hp)
        data['labels'] = torch.from_numpy(make_labels(44, 2, seed=2)).to(dtype)[None]
        model.eps_fn = _Eps()
    elif case == 'betatc':
        hp = base_hparams(arch, 'beta-tcvae', {'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0,
                                               'beta_tcvae.beta': 4.0,
                                               'beta_tcvae.beta_anneal_epochs': 0,
                                               'max_n_epochs': 10})
        model = ref_cpu.BetaTCVAE(hp)
        model.eps_fn = _Eps()
    elif case == 'aemsp':
        hp = base_hparams(arch, 'cond-ae-msp', {'msp.alpha': 0.05, 'conditional_encoder': False})
        hp['n_labels'] = 4
        model = ref_cpu.AEMSP(hp)
        data['labels'] = torch.from_numpy(make_labels(44, 4, seed=2)).to(dtype)[None]
    else:
        raise ValueError(case)
    model = model.to(dtype)
    model.train()
    return model, data, {'chunk_size': 30}


def flat_grad(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad])


def run_fit(tmp):
    """fit() of the conv AE on 10 trials x 32 frames, 2 epochs (frames mode when distributed)."""
    arch = load_handcrafted_arch(list(DIM), 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    hp.update({'expt_dir': tmp, 'max_n_epochs': 2, 'min_n_epochs': 0, 'val_check_interval': 1,
               'enable_early_stop': False, 'early_stop_history': 10, 'rng_seed_train': 0,
               'export_latents': False, 'progress_bar': False, 'device': 'cuda'})
    os.makedirs(os.path.join(tmp, 'version_0'), exist_ok=True)
    sess = SyntheticSession(10, 32, DIM, seed=0, trial_splits='8;1;1;0')
    gen = SyntheticSessionsGenerator([sess], device=DEV, placement='device_u8')
    torch.manual_seed(0)
    model = AE(hp).to(DEV)
    model.version = 0

    class Exp(object):
        version = 0
        rows = []

        def log(self, row):
            self.rows.append(dict(row))

        def save(self):
            pass
    exp = Exp()
    fit(hp, model, gen, exp, method='ae')
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    idx = torch.linspace(0, flat.numel() - 1, 4096).long()
    rows = [{k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in r.items()}
            for r in exp.rows]
    return {'rows': rows, 'param_sample': flat[idx.to(flat.device)].cpu().tolist()}


def run_refusals():
    """AEMSP with batch norm and MSPSVAE under frame sharding: EVERY rank must raise before it issues a kernel or
    a collective (a rank that went on alone would hang the others in their next collective).  ->
    {class name: message} of what this rank raised."""
    from behavenet_amd.models import AEMSP
    from tests.test_gpu_model import _pair
    from tests.test_oracle_golden import _msps_case
    got = {}
    arch = load_handcrafted_arch(list(DIM), 8, None, check_memory=False)
    # (AEMSP is served since round 4 -- case 'aemsp'; its batch-norm variant is not)
    hp = base_hparams(arch, 'cond-ae-msp', {'msp.alpha': 0.05, 'conditional_encoder': False,
                                            'ae_batch_norm': True})
    hp['n_labels'] = 4
    torch.manual_seed(0)
    model = AEMSP(hp).to(DEV)
    data = {'images': torch.from_numpy(make_frames(44, DIM, seed=8)).to(DEV)[None],
            'labels': torch.from_numpy(make_labels(44, 4, seed=2)).to(DEV)[None]}
    try:
        model.loss(data, dataset=0, accumulate_grad=True, chunk_size=30)
        got['AEMSP'] = None
    except NotImplementedError as err:
        got['AEMSP'] = str(err)
    _, meta, datas_c = _msps_case()
    msps, _, _ = _pair(meta)
    datas_g = [{k: v.to(DEV) for k, v in d.items()} for d in datas_c]
    try:
        msps.loss(datas_g[0], dataset=meta['sess'][0] if isinstance(meta.get('sess'), list) else 0,
                  accumulate_grad=True)
        got['MSPSVAE'] = None
    except NotImplementedError as err:
        got['MSPSVAE'] = str(err)
    return got


def run_sharded_optimizer():
    """Three optimizer steps of an AE on this rank's frames: reduce-scatter -> Adam on this rank's
    half of the arena -> all-gather (`bdist.sharded_step`) against all-reduce + the full step, on the
    device kernel.  -> max |difference| of the parameters (two ranks: must be 0)."""
    import copy
    model, data, kw = build_case('ae_bn')
    twin = copy.deepcopy(model)
    opt_s = FlatAdamAMSGrad(model.get_parameters(), lr=1e-3, shard_over=2)
    opt_r = FlatAdamAMSGrad(twin.get_parameters(), lr=1e-3)
    for _ in range(3):
        for m, opt in ((model, opt_s), (twin, opt_r)):
            opt.zero_grad()
            m.loss(data, dataset=0, accumulate_grad=True, **kw)
        bdist.sharded_step(opt_s)
        bdist.reduce_gradients(opt_r)
        opt_r.step()
    n = opt_r.flat_p.numel()
    diff = float((opt_s.flat_p[:n] - opt_r.flat_p).abs().max().item())
    return {'max_abs_diff': diff, 'params_finite': bool(torch.isfinite(opt_s.flat_p).all().item()),
            'steps': opt_s.step_count}


def run_case(case, tmp, rank):
    if case == 'shardopt':
        out = run_sharded_optimizer()
        with open(os.path.join(tmp, 'shardopt_rank%d.json' % rank), 'w') as f:
            json.dump(out, f)
    elif case == 'refuse':
        out = run_refusals()
        with open(os.path.join(tmp, 'refuse_rank%d.json' % rank), 'w') as f:
            json.dump(out, f)
    elif case == 'fit':
        out = run_fit(os.path.join(tmp, 'rank%d' % rank))
    else:
        from tests.branches import record_branches
        model, data, kw = build_case(case)
        opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4)
        opt.zero_grad()
        try:
            with record_branches(model) as rec:
                loss = model.loss(data, dataset=0, accumulate_grad=True, **kw)
        finally:
            hip_vaes.set_eps_provider(None)
        bdist.reduce_gradients(opt)
        g = flat_grad(model).cpu().double().numpy()
        # this rank's LeakyReLU branch pattern (its frames, its processing order) and, from rank
        # 0, the whole all-reduced gradient: the test runs the float64 oracle on the assembled
        # pattern (tests/branches.py)
        torch.save(rec, os.path.join(tmp, '%s_branches_rank%d.pt' % (case, rank)))
        if rank == 0:
            np.save(os.path.join(tmp, case + '_grad.npy'), g)
        idx = np.linspace(0, g.size - 1, 8192).astype(np.int64)
        out = {'loss': loss, 'grad_norm': float(np.linalg.norm(g)), 'grad_index': idx.tolist(),
               'grad_sample': g[idx].tolist(),
               'buffers': {k: v.float().reshape(-1)[:64].cpu().tolist()
                           for k, v in model.named_buffers() if 'running_' in k}}
    if rank == 0:
        with open(os.path.join(tmp, case + '_rank0.json'), 'w') as f:
            json.dump(out, f)


def main():
    """argv: comma-separated cases, output directory.  ONE rendezvous serves all cases (process
    start-up, HIP context and the first launches of every kernel are paid once per rank, not once
    per case); a barrier separates them."""
    cases, tmp = sys.argv[1].split(','), sys.argv[2]
    torch.cuda.set_device(0)
    rank, world = bdist.init_from_env(backend='gloo')
    assert world == 2 and bdist.shard_mode() == 'frames' and bdist.frames_sharded()
    for case in cases:
        print('rank %d: case %s' % (rank, case), flush=True)
        run_case(case, tmp, rank)
        torch.distributed.barrier()
        if rank == 0:       # marks the case complete for the waiting test process
            open(os.path.join(tmp, case + '.done'), 'w').close()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
