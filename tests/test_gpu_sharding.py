"""Frame-sharded data parallelism (BASELINE configs[2]) on the HIP models.

* one GPU, ranks emulated one after another (``bdist.emulate_rank``): the sum over ranks of the
  per-rank loss contributions and gradients equals the unsharded HIP step and the CPU oracle
  (reference aes.py:751-771: one mean per 200-frame chunk, gradients accumulated over chunks);
* two real processes on the one GPU with a gloo rendezvous (collectives staged through the
  host): the batch-coupled terms that emulation cannot provide -- batch-norm statistics over all
  ranks' frames (aes.py:90-97,332-336) and the decomposed KL on the all-gathered chunk
  (losses.py:321-341) -- and ``fit`` in 'frames' mode against the single-process run."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.models import AE, VAE
from behavenet_amd.models import vaes as hip_vaes
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from oracle import ref_cpu
from tests.golden_utils import base_hparams, make_frames
from tests.test_gpu_kernels import close

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _grads(model):
    return [p.grad.detach().clone() for p in model.parameters() if p.requires_grad]


def _sum_over_emulated_ranks(model, data, R, chunk_size, **loss_kw):
    total, loss = None, {}
    for r in range(R):
        model.zero_grad(set_to_none=True)
        with bdist.emulate_rank(r, R):
            out = model.loss(data, dataset=0, accumulate_grad=True, chunk_size=chunk_size, **loss_kw)
        g = _grads(model)
        total = g if total is None else [a + b for a, b in zip(total, g)]
        for k, v in out.items():
            loss[k] = loss.get(k, 0.0) + v
    return loss, total


@pytest.mark.parametrize('dim,n_lat,batch,chunk,R', [
    ([1, 32, 32], 8, 210, 200, 2), ([1, 32, 32], 8, 210, 200, 8),
    ([1, 128, 128], 12, 256, 200, 2), ([1, 128, 128], 12, 256, 200, 8)])
def test_ae_frame_shards_add_up_to_the_single_device_step(dim, n_lat, batch, chunk, R):
    arch = load_handcrafted_arch(list(dim), n_lat, None, check_memory=False)
    torch.manual_seed(0)
    model = AE(base_hparams(arch, 'ae')).to(DEV)
    x = torch.from_numpy(make_frames(batch, dim, seed=5))
    data = {'images': x.to(DEV)[None]}
    model.zero_grad(set_to_none=True)
    whole = model.loss(data, dataset=0, accumulate_grad=True, chunk_size=chunk)
    g_whole = _grads(model)
    shard_loss, g_sum = _sum_over_emulated_ranks(model, data, R, chunk)
    assert shard_loss['loss'] == pytest.approx(whole['loss'], rel=1e-6)
    for a, b in zip(g_sum, g_whole):
        # same kernels on fewer frames: fp32 summation order (and LeakyReLU ties, see
        # tests/branches.py) -- L2 at rounding level, single elements within 1e-3 of the max
        err = (a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)
        assert float(err) <= 1e-3
        close(a, b, norm_tol=2e-3, name='sum of shard gradients')
    if dim[1] == 32:
        # and against the oracle's chunk loop (fp32, CPU)
        torch.manual_seed(0)
        ora = ref_cpu.AE(base_hparams(dict(arch), 'ae'))
        lo = ora.loss({'images': x[None]}, dataset=0, accumulate_grad=True, chunk_size=chunk)
        assert shard_loss['loss'] == pytest.approx(lo['loss'], rel=1e-5)
        for a, p in zip(g_sum, ora.parameters()):
            # (fp32 oracle, own LeakyReLU branches at ties: tests/branches.py has the exact form)
            err = (a.cpu().double() - p.grad.double()).norm() / p.grad.double().norm()
            assert float(err) <= 2e-3


def test_vae_frame_shards_use_the_single_device_eps():
    dim, batch, chunk, R = [1, 32, 32], 210, 200, 4
    arch = load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, 'vae', {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0,
                                    'max_n_epochs': 10})
    torch.manual_seed(0)
    model = VAE(hp).to(DEV)
    x = torch.from_numpy(make_frames(batch, dim, seed=6))
    data = {'images': x.to(DEV)[None]}
    g = torch.Generator().manual_seed(1)
    eps_chunks = [torch.randn((200, 8), generator=g), torch.randn((10, 8), generator=g)]

    class Replay(object):
        def __init__(self):
            self.i = 0

        def __call__(self, like):
            t = eps_chunks[self.i % 2].to(like.device)
            self.i += 1
            assert t.shape == like.shape
            return t
    try:
        hip_vaes.set_eps_provider(Replay())
        model.zero_grad(set_to_none=True)
        whole = model.loss(data, dataset=0, accumulate_grad=True, chunk_size=chunk)
        g_whole = _grads(model)
        hip_vaes.set_eps_provider(Replay())
        shard_loss, g_sum = _sum_over_emulated_ranks(model, data, R, chunk)
    finally:
        hip_vaes.set_eps_provider(None)
    # (loss_mse is an affine function of loss_ll evaluated AFTER the sum over ranks: not additive)
    for k in ('loss', 'loss_ll', 'loss_kl'):
        assert shard_loss[k] == pytest.approx(whole[k], rel=2e-5, abs=1e-6), k
    for a, b in zip(g_sum, g_whole):
        err = (a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)
        assert float(err) <= 1e-4


def test_batch_coupled_terms_refuse_emulation():
    """Batch-norm statistics and the decomposed KL need the other ranks' data inside the step."""
    dim = [1, 32, 32]
    arch = load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    torch.manual_seed(0)
    model = AE(base_hparams(arch, 'ae', {'ae_batch_norm': True})).to(DEV)
    data = {'images': torch.from_numpy(make_frames(12, dim, seed=1)).to(DEV)[None]}
    with bdist.emulate_rank(0, 2):
        with pytest.raises(RuntimeError):
            model.loss(data, dataset=0, accumulate_grad=True)


def _run_two_ranks(tmp_path, case):
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), BN_DP_SHARD='frames',
                   PYTHONPATH=REPO)
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(REPO, 'tests', 'dist_gpu_two_ranks.py'), case,
             str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    with open(os.path.join(str(tmp_path), case + '_rank0.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('case', ['ae_bn', 'psvae', 'betatc'])
def test_two_ranks_on_one_gpu_match_the_single_process_step(tmp_path, case):
    """Real collectives (gloo, host-staged) between two processes sharing the GPU."""
    got = _run_two_ranks(tmp_path, case)
    from tests.dist_gpu_two_ranks import build_case, flat_grad
    model, data, kw = build_case(case)
    model.zero_grad(set_to_none=True)
    want = model.loss(data, dataset=0, accumulate_grad=True, **kw)
    for k, v in want.items():
        assert got['loss'][k] == pytest.approx(v, rel=5e-5, abs=1e-6), k
    g = flat_grad(model).cpu().double().numpy()
    gg = np.asarray(got['grad_sample'])
    idx = np.asarray(got['grad_index'])
    scale = np.abs(g).max()
    assert np.abs(gg - g[idx]).max() <= 2e-4 * scale
    assert got['grad_norm'] == pytest.approx(float(np.linalg.norm(g)), rel=1e-4)
    for k, v in got.get('buffers', {}).items():          # batch-norm running statistics
        b = dict(model.named_buffers())[k].float().cpu().numpy()
        np.testing.assert_allclose(np.asarray(v), b.reshape(-1)[:len(v)], rtol=1e-4, atol=1e-6)


def test_fit_in_frames_mode_matches_single_process(tmp_path):
    got = _run_two_ranks(tmp_path, 'fit')
    from tests.dist_gpu_two_ranks import run_fit
    want = run_fit(os.path.join(str(tmp_path), 'single'))
    assert len(got['rows']) == len(want['rows'])
    for a, b in zip(got['rows'], want['rows']):
        assert a.keys() == b.keys()
        for k, v in b.items():
            if isinstance(v, float):
                assert a[k] == pytest.approx(v, rel=1e-4), k
            else:
                assert a[k] == v, k
    # 16 Adam steps.  Adam normalises every gradient element by its own magnitude, so where the
    # gradient is rounding noise (or a LeakyReLU tie fell the other way on the 16-frame shards)
    # a weight moves by up to lr per step in either run: bound every element by the full travel
    # and all but a few by 2 % of it
    travel = 16 * 1e-4
    diff = np.abs(np.asarray(got['param_sample']) - np.asarray(want['param_sample']))
    assert diff.max() <= travel
    assert np.mean(diff > 0.02 * travel + 1e-3 * np.abs(want['param_sample'])) <= 0.02


def test_bench_two_ranks_control_flow():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank),
    with both ranks on the one GPU over gloo: parameter broadcast, per-step gradient all-reduce,
    barrier-bracketed timing, max over ranks, one JSON line from rank 0 with the whole-job value."""
    env = dict(os.environ, BN_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29517', os.path.join(REPO, 'bench.py'),
           '--gpus', '2', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-secondary']
    out = subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=600).stdout.decode()
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak'
    assert d['config']['global_frames_per_step'] == 512
    assert abs(d['value'] - 512 * 1e3 / d['ms_per_step']) <= 0.01 * d['value']
    assert np.isfinite(d['final_loss'])
