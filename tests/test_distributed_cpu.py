"""world_size-2 gloo checks of the data-parallel host logic (runs on CPU).

With the CPU oracle model standing in for the HIP model: frame-sharded chunks whose local loss
is scaled to the global chunk mean, followed by ONE all-reduce (sum) of the flat gradient,
must reproduce the single-process gradient of AE.loss."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from oracle import ref_cpu
from tests.golden_utils import base_hparams, make_frames


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _sharded_grad(model, x, chunk_size):
    """'frames' mode: local slice of every chunk, loss scaled to the global chunk mean."""
    B = x.shape[0]
    per_elem = int(np.prod(x.shape[1:]))
    for beg in range(0, B, chunk_size):
        end = min(beg + chunk_size, B)
        lb, le = bdist.shard_bounds(beg, end)
        if le <= lb:
            continue
        x_in = x[lb:le]
        x_hat, _ = model(x_in, dataset=0)
        loss = ((x_in - x_hat) ** 2).sum() / ((end - beg) * per_elem)
        loss.backward()


def _worker(rank, world, port, out):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                       'RANK': str(rank), 'WORLD_SIZE': str(world)})
    torch.set_num_threads(2)
    r, w = bdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) and bdist.is_active()
    arch = load_handcrafted_arch([1, 32, 32], 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    torch.manual_seed(0)
    model = ref_cpu.AE(hp)
    x = torch.from_numpy(make_frames(50, [1, 32, 32], seed=3))
    _sharded_grad(model, x, chunk_size=30)
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    bdist.all_reduce_flat_(flat)
    tot = bdist.all_reduce_scalars([1.0, float(rank)])
    if rank == 0:
        out.put((flat.numpy(), tot))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_gradient_matches_single_process():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    flat, tot = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tot == [2.0, 1.0]

    arch = load_handcrafted_arch([1, 32, 32], 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    torch.manual_seed(0)
    model = ref_cpu.AE(hp)
    x = torch.from_numpy(make_frames(50, [1, 32, 32], seed=3))
    model.loss({'images': x[None]}, dataset=0, accumulate_grad=True, chunk_size=30)
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    scale = np.abs(want).max()
    np.testing.assert_allclose(flat, want, rtol=1e-4, atol=1e-6 * scale)


def test_shard_bounds_partition():
    for n in [1, 7, 56, 200]:
        for R in [1, 2, 3, 8]:
            edges = [bdist.shard_bounds(10, 10 + n, r, R) for r in range(R)]
            assert edges[0][0] == 10 and edges[-1][1] == 10 + n
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0]


class _ArenaStub(object):
    """What BucketedGradReducer reads of FlatAdamAMSGrad: params, offsets, flat_g."""

    def __init__(self, sizes, align=64):
        self.params = [torch.nn.Parameter(torch.zeros(n)) for n in sizes]
        self.offsets, total = [], 0
        for n in sizes:
            self.offsets.append(total)
            total += (n + align - 1) // align * align
        self.flat_g = torch.zeros(total)


def _reducer_worker(rank, world, port, out):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                       'RANK': str(rank), 'WORLD_SIZE': str(world)})
    torch.set_num_threads(1)
    bdist.init_from_env(backend='gloo')
    # encoder-like then decoder-like sizes (floats); 4 KiB buckets
    sizes = [100, 800, 3000, 50, 60, 2500, 900, 30]
    opt = _ArenaStub(sizes)
    red = bdist.BucketedGradReducer(opt, bucket_bytes=4096)
    spans = sorted((lo, hi) for lo, hi, _ in red.buckets)
    assert spans[0][0] == 0 and spans[-1][1] == opt.flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    results = []
    for step, order in enumerate([[7, 6, 5, 4, 3, 2, 1, 0],       # everything reported, in order
                                  [5, 7, 6, 3, 4],                # out of order, three missing
                                  []]):                           # chunked schedule: none
        red.begin()
        g = torch.Generator().manual_seed(10 * step + rank)
        opt.flat_g.copy_(torch.randn(opt.flat_g.shape, generator=g))
        launched_before_finish = []
        for i in order:
            red.grad_ready(opt.params[i])
            launched_before_finish.append(sum(red._launched))
        n_early = sum(red._launched)
        red.finish()
        assert all(red._launched)
        results.append((opt.flat_g.clone().numpy(), n_early, launched_before_finish))
    if rank == 0:
        out.put(results)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_reducer_equals_flat_all_reduce():
    """Overlapped bucket launches + finish() == one all-reduce of the arena, whatever subset of
    parameters was reported and in whatever order (collectives stay in bucket order)."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = results[0][0].shape[0]
    for step, (got, n_early, trace) in enumerate(results):
        want = sum(torch.randn((n,), generator=torch.Generator().manual_seed(10 * step + r))
                   for r in range(2)).numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=0)
        assert trace == sorted(trace)
    assert results[0][1] == max(results[0][2]) >= 3     # every bucket went out before finish()
    assert 0 < results[1][1] < results[0][1]     # a missing parameter holds its bucket and later ones
    assert results[2][1] == 0


class _CpuShardedAdam(object):
    """FlatAdamAMSGrad's sharding interface (flat_p, flat_g, shard_over, shard_range, step,
    step_range) on the CPU, with the element-wise Adam(amsgrad) update written out."""

    def __init__(self, n, shard_over, lr=1e-2):
        per = ((n + shard_over - 1) // shard_over + 3) // 4 * 4
        total = per * shard_over
        self.shard_over, self.lr, self.t = shard_over, lr, 0
        g = torch.Generator().manual_seed(5)
        self.flat_p = torch.randn(total, generator=g)
        self.flat_g = torch.zeros(total)
        self.m, self.v, self.vmax = torch.zeros(total), torch.zeros(total), torch.zeros(total)

    def shard_range(self, r):
        per = self.flat_p.numel() // self.shard_over
        return r * per, (r + 1) * per

    def step(self):
        self.step_range(0, self.flat_p.numel())

    def step_range(self, lo, hi):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        g = self.flat_g[lo:hi]
        self.m[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
        self.v[lo:hi].mul_(b2).addcmul_(g, g, value=1 - b2)
        torch.maximum(self.vmax[lo:hi], self.v[lo:hi], out=self.vmax[lo:hi])
        denom = (self.vmax[lo:hi].sqrt() / (1 - b2 ** self.t) ** 0.5).add_(eps)
        self.flat_p[lo:hi].addcdiv_(self.m[lo:hi], denom, value=-self.lr / (1 - b1 ** self.t))


def _sharded_step_worker(rank, world, port, out):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                       'RANK': str(rank), 'WORLD_SIZE': str(world)})
    torch.set_num_threads(1)
    bdist.init_from_env(backend='gloo')
    n = 10007                                     # not a multiple of anything
    sharded, replicated = _CpuShardedAdam(n, world), _CpuShardedAdam(n, world)
    for step in range(3):
        g = torch.randn(sharded.flat_g.shape, generator=torch.Generator().manual_seed(100 * step + rank))
        sharded.flat_g.copy_(g)
        replicated.flat_g.copy_(g)
        bdist.sharded_step(sharded, average=(step == 1), divide_by=3.0 if step == 2 else None)
        # the replicated step: all-reduce, the same scalings, a full step on every rank
        bdist.all_reduce_flat_(replicated.flat_g, average=(step == 1))
        if step == 2:
            replicated.flat_g.div_(3.0)
        replicated.step()
    lo, hi = sharded.shard_range(rank)
    others_untouched = bool((sharded.m[:lo] == 0).all() and (sharded.m[hi:] == 0).all())
    out.put((rank, sharded.flat_p.numpy().copy(), replicated.flat_p.numpy().copy(), others_untouched))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_optimizer_step_equals_the_replicated_step(world):
    """reduce-scatter -> Adam on this rank's shard -> all-gather (fitting/distributed.py
    sharded_step) against all-reduce + identical full steps: the same
    parameters on every rank -- bit for bit with two ranks (a + b in either form), to rounding
    with three (the two reductions may add the ranks' terms in different orders) -- and the
    moments of the other ranks' shards never touched."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_step_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = results[0][1]
    for rank, got, want, untouched in results:
        assert untouched, rank
        np.testing.assert_array_equal(got, ref)            # every rank holds the same parameters
        if world == 2:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
