"""``model.loss`` as a HIP graph: record once per input signature, replay per batch.

A training step of the conv autoencoder is ~75 kernel launches, issued by autograd and ctypes in
2-3 ms of host time per step on an idle host.  Where the GPU finishes them faster than the host
issues them (a busy host, many ranks per socket, small frames and batches) the step is as slow as
the host.  ``GraphedLoss`` records ``model.loss(data, dataset, accumulate_grad)``
-- forward, loss, backward into the gradient arena -- into a ``torch.cuda.CUDAGraph`` (a HIP graph
on ROCm) the first time an input signature has been seen ``warmup`` times, and from then on a step
is: copy the batch into the graph's static input tensors, one graph launch, the eager optimizer
kernel.  Same kernels in the same order on the same operands: the gradients are bit-identical to
the eager step (tests/test_gpu_graph_step.py).

Nothing inside a recording may wait on the host, so the loss VALUES come back differently: the
models' ``loss`` returns a ``DeferredLoss`` (hip_functions.finish_loss) while recording; after
every replay the recorded device tensors are read back asynchronously and ``LazyLoss`` turns them
into the loss dict when somebody looks at it -- ``fit``'s logger a few steps later, so the host
never waits for a step it has just launched.

What is recorded must not depend on host state that changes from step to step.  The signature
therefore holds everything ``loss`` reads from the host: tensor shapes and dtypes, the dataset
index, train / eval mode, ``accumulate_grad``, the sharding state and ``model.curr_epoch`` for the
classes whose loss weights are annealed.  Classes whose ``loss`` has a host tail that is not routed
through ``finish_loss`` are run eagerly (``supported``).
"""

import collections.abc
import os
import warnings

import torch

from behavenet_amd import hip_functions as hf
from behavenet_amd.fitting import distributed as bdist

__all__ = ['GraphedLoss', 'LazyLoss', 'enabled_by_default']


def enabled_by_default():
    """``fit`` replays graphs when ``hparams['hip_graph']`` is true, ``BN_GRAPH=1``, or -- round 5 -- the
    fit is frame-sharded over four or more ranks.  Measured on the MI355X (tools/bench_graph_step.py,
    tools/bench_frames_shard.py): with a whole trial per step the GPU is the bound -- 4.32 / 4.33 ms
    eager / graph at 256 frames of 128x128 -- so a graph only pays where the host is slower than
    here; the 32-frame shard of an 8-rank frame-sharded step is ~75 launches of 5-40 us, which the
    host issues in 1.41 ms and the graph replays in 1.04 (bit-identical results either way)."""
    env = os.environ.get('BN_GRAPH')
    if env is not None:
        return env == '1'
    return bdist.frames_sharded() and bdist.shard_rank_world()[1] >= 4


LazyLoss = hf.LazyLoss      # (the class lives next to Readback: eager steps hand it out as well, set_lazy_losses)


def _tensor_slots(data):
    """[(key, index)] of the device tensors of a batch dict, in a fixed order."""
    slots = []
    for k in sorted(data.keys()):
        v = data[k]
        if isinstance(v, (list, tuple)):
            for i, t in enumerate(v):
                if torch.is_tensor(t) and t.is_cuda:
                    slots.append((k, i))
        elif torch.is_tensor(v) and v.is_cuda:
            slots.append((k, None))
    return slots


def _get(data, slot):
    k, i = slot
    return data[k] if i is None else data[k][i]


class GraphedLoss(object):

    def __init__(self, model, warmup=2, max_graphs=6):
        self.model = model
        self.warmup = int(warmup)
        self.max_graphs = int(max_graphs)
        self._graphs = {}
        self._seen = {}
        self._refused = set()
        self._pool = None
        self.n_replays = 0
        self.n_eager = 0

    # ------------------------------------------------------------------------------------------
    def supported(self, data):
        m = self.model
        if not getattr(m, 'graph_capturable', False):
            return False
        if isinstance(data, (list, tuple)):         # multi-session batches (MSPSVAE)
            return False
        x = data['images'][0]
        if not x.is_cuda:
            return False
        return bool(m.graph_capturable_for(x)) if hasattr(m, 'graph_capturable_for') else True

    def _key(self, data, dataset, accumulate_grad):
        sig = []
        for slot in _tensor_slots(data):
            t = _get(data, slot)
            sig.append((slot, tuple(t.shape), t.dtype, tuple(t.stride())))
        other = tuple(sorted((k, repr(v)) for k, v in data.items()
                             if not isinstance(v, (list, tuple)) and not torch.is_tensor(v)))
        epoch = self.model.curr_epoch if getattr(self.model, 'graph_epoch_dependent', False) \
            else None
        return (tuple(sig), other, repr(dataset), bool(accumulate_grad), bool(self.model.training),
                epoch, bdist.shard_mode(), bdist.shard_rank_world())

    # ------------------------------------------------------------------------------------------
    def __call__(self, data, dataset=0, accumulate_grad=True):
        if not self.supported(data):
            self.n_eager += 1
            return self.model.loss(data, dataset=dataset, accumulate_grad=accumulate_grad)
        key = self._key(data, dataset, accumulate_grad)
        rec = self._graphs.get(key)
        if rec is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n <= self.warmup or key in self._refused or len(self._graphs) >= self.max_graphs:
                self.n_eager += 1
                return self.model.loss(data, dataset=dataset, accumulate_grad=accumulate_grad)
            rec = self._record(key, data, dataset, accumulate_grad)
            self._compare_with_peers(rec is not None)
            if rec is None:
                self.n_eager += 1
                return self.model.loss(data, dataset=dataset, accumulate_grad=accumulate_grad)
        for slot, dst in zip(rec.slots, rec.static_tensors()):
            src = _get(data, slot)
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        rec.graph.replay()
        self.n_replays += 1
        d = rec.deferred
        tensors = d.tensors
        if d.reduce_over_ranks:
            # frame-sharded step of real ranks: the recording holds this rank's chunk terms; their
            # sum over ranks is the one collective of the step, issued here, behind the replay
            tensors = [None if t is None else bdist.all_reduce_(t.clone()) for t in tensors]
        return LazyLoss([None if t is None else hf.Readback(t) for t in tensors], d.fn)

    # ------------------------------------------------------------------------------------------
    def _compare_with_peers(self, recorded):
        """Real ranks record at the same step (same warm-up count, same signatures).  A rank whose capture
        failed keeps launching eagerly while its peers replay: the collectives still line up (the loss-table
        all-reduce sits behind the backward pass either way), but the job then runs at the eager rank's pace
        -- say so on every rank instead of leaving a slow job unexplained."""
        self.peers_recorded = None
        if not bdist.is_active() or bdist._emulated is not None or not bdist.frames_sharded():
            return
        import torch.distributed as dist
        flags = [None] * dist.get_world_size()
        dist.all_gather_object(flags, bool(recorded))
        self.peers_recorded = flags
        if any(flags) and not all(flags):
            warnings.warn('HIP graph of the frame-sharded step: ranks %s replay a graph, ranks %s stayed on eager '
                          'launches (capture failed there); results are the same, the step runs at the eager '
                          'ranks\' pace' % ([r for r, f in enumerate(flags) if f],
                                            [r for r, f in enumerate(flags) if not f]))

    # ------------------------------------------------------------------------------------------
    def _record(self, key, data, dataset, accumulate_grad):
        slots = _tensor_slots(data)
        static = {}
        for k, v in data.items():
            static[k] = list(v) if isinstance(v, (list, tuple)) else v
        for slot in slots:
            k, i = slot
            t = torch.empty_like(_get(data, slot))
            t.copy_(_get(data, slot))
            if i is None:
                static[k] = t
            else:
                static[k][i] = t
        rec = _RecordedGraph()
        rec.static, rec.slots = static, slots
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        hf._capturing = True
        # test hook: BN_GRAPH_FAULT_RANK=<r> -- that rank's capture fails (mixed eager / replaying ranks)
        fault = os.environ.get('BN_GRAPH_FAULT_RANK')
        try:
            if fault is not None and bdist.is_active() and int(fault) == bdist.rank():
                raise RuntimeError('injected capture failure (BN_GRAPH_FAULT_RANK)')
            # thread_local: the data generator's prefetch thread may be copying on its own stream
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode='thread_local'):
                out = self.model.loss(static, dataset=dataset, accumulate_grad=accumulate_grad)
        except Exception as err:                        # noqa: BLE001 (reported, then eager)
            hf._capturing = False
            torch.cuda.synchronize()
            self._refused.add(key)
            warnings.warn('HIP graph capture of %s.loss failed (%s: %s); this input signature '
                          'stays on eager launches' % (type(self.model).__name__,
                                                       type(err).__name__, err))
            return None
        finally:
            hf._capturing = False
        if not isinstance(out, hf.DeferredLoss):
            self._refused.add(key)
            warnings.warn('%s.loss returned values while being recorded; eager launches kept'
                          % type(self.model).__name__)
            return None
        rec.graph, rec.deferred = graph, out
        self._graphs[key] = rec
        return rec


class _RecordedGraph(object):
    __slots__ = ('graph', 'static', 'slots', 'deferred')

    def static_tensors(self):
        return [_get(self.static, s) for s in self.slots]
