"""Data-parallel training over RCCL/xGMI: one process per GPU.

The reference has no working multi-GPU path (its ``nn.DataParallel`` wrapper is bypassed by
``fit``).  Here frames shard across ranks and the only exchange is ONE
all-reduce (sum) of the flat gradient arena per optimizer step (35 MB for the default arch).

Two sharding modes:

* ``'trial'`` (weak scaling, default): the training trials of an epoch are dealt to the ranks
  in groups of ``world_size`` (every rank walks the same order and keeps the trial at its own
  position), gradients are AVERAGED over the trials of the step (``fit``; a short last group
  divides by its own size), i.e. one optimizer step consumes up to ``world_size`` trials.  Not
  step-for-step identical to the single-GPU reference (which steps once per trial).
* ``'frames'`` (strong scaling, parity-exact): all ranks see the same trial; inside each
  200-frame chunk rank r takes the contiguous slice [r*n_c/R, (r+1)*n_c/R) and its loss is
  scaled to the *global* chunk mean, so the summed gradient equals the single-GPU gradient.

Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` on CPU (tests).
"""

import os

import torch
import torch.distributed as dist


# BN_DIST_FORCE=1 keeps the collective code paths on for a world of ONE rank (a 1-GPU box then
# runs the real RCCL launches and stream hand-offs; used by tests/test_gpu_model.py)
_FORCE = os.environ.get('BN_DIST_FORCE', '0') == '1'


def is_active():
    return dist.is_available() and dist.is_initialized() and \
        (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def _loopback(addr):
    return addr in ('127.0.0.1', 'localhost', '::1')


def init_from_env(backend=None, timeout_s=None, rendezvous_timeout_s=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun env).

    Two time limits (one short limit for both aborted jobs whose rank 0 spent more than
    ten minutes writing a checkpoint between two collectives):

    * ``rendezvous_timeout_s`` (env BN_DIST_RDZV_TIMEOUT_S; default 120): how long the ranks wait
      for each other at start-up.  A rank that never comes up (a GPU that is not there, a crashed
      import) then fails the job within two minutes instead of torch's thirty.
    * ``timeout_s`` (env BN_DIST_TIMEOUT_S; default 1800, torch's own): how long a collective may
      wait for a missing rank.  ``bench.py --gpus N`` sets a short one for itself.

    When the rendezvous address is the loopback interface (one node: what ``torchrun --master-addr
    127.0.0.1`` and bench.py use) gloo and the RCCL bootstrap are pinned to ``lo`` unless the user
    chose an interface: both otherwise pick theirs from the HOST NAME, which on a container need
    not resolve to anything reachable."""
    if dist.is_initialized():
        return rank(), world_size()
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1 and not _FORCE:
        return 0, 1
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if _loopback(os.environ['MASTER_ADDR']):
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    if backend is None:
        # BN_DIST_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL refuses two ranks per device)
        backend = os.environ.get('BN_DIST_BACKEND') or (
            'nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        # a collective that times out (or a peer that died) takes this rank's PROCESS down (mode 1 =
        # tear down: it does not raise) instead of leaving it spinning inside a kernel; the launcher
        # then stops the peers (bench.py's SIGTERM handler / watchdog print the error line)
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
    if timeout_s is None:
        timeout_s = float(os.environ.get('BN_DIST_TIMEOUT_S', '1800'))
    if rendezvous_timeout_s is None:
        rendezvous_timeout_s = float(os.environ.get('BN_DIST_RDZV_TIMEOUT_S', '120'))
    import datetime
    r = int(os.environ['RANK'])
    # the store (= the rendezvous) with its own, short limit.  Under torchrun the agent already
    # serves a store on MASTER_PORT (TORCHELASTIC_USE_AGENT_STORE): every worker is a client of it,
    # with a per-attempt key prefix, as torch's own env:// handler does
    agent_store = os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True'
    tcp_store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']), ws,
                              (r == 0) and not agent_store,
                              timeout=datetime.timedelta(seconds=float(rendezvous_timeout_s)))
    store = tcp_store
    if agent_store:
        store = dist.PrefixStore('/bn/attempt_%s' % os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'),
                                 store)
    dist.init_process_group(backend=backend, store=store, rank=r, world_size=ws,
                            timeout=datetime.timedelta(seconds=float(timeout_s)))
    # everybody is here (waits at most the rendezvous limit: the store's)
    store.add('bn_ranks_up', 1)
    store.wait(['bn_ranks_up'])
    import time
    t_end = time.time() + float(rendezvous_timeout_s)
    while int(store.add('bn_ranks_up', 0)) < ws:
        if time.time() > t_end:
            raise RuntimeError('rendezvous: %d of %d ranks came up within %.0f s' % (
                int(store.add('bn_ranks_up', 0)), ws, float(rendezvous_timeout_s)))
        time.sleep(0.01)
    # from here on the store carries the COLLECTIVE limit: torch only calls set_timeout on a store it
    # made itself, and the store-backed waits that come later (the lazy RCCL communicator bootstrap at
    # the first collective, new_group, gloo pair set-up) would otherwise keep the short rendezvous
    # limit -- ranks that reach their first collective more than two minutes apart (rank 0 in
    # create_experiment, one rank compiling the library) aborted the job
    for s in {id(tcp_store): tcp_store, id(store): store}.values():
        s.set_timeout(datetime.timedelta(seconds=float(timeout_s)))
    return rank(), world_size()


# ------------------------------------------------------------------------------------------
# sharding mode
# ------------------------------------------------------------------------------------------
_MODES = ('trial', 'frames')
_mode = os.environ.get('BN_DP_SHARD', 'trial')
_emulated = None          # (rank, world): tests run the ranks of a 'frames' step one after another


def set_shard_mode(mode):
    """'trial' (weak scaling: one trial per rank per step) or 'frames' (strong scaling,
    parity-exact: every rank sees the same trial and takes its slice of every 200-frame chunk);
    returns the previous mode.  Also read from hparams['dp_shard'] by ``fit``."""
    global _mode
    if mode not in _MODES:
        raise ValueError('dp shard mode must be one of %s, got "%s"' % (_MODES, mode))
    prev, _mode = _mode, mode
    return prev


def shard_mode():
    return _mode


class emulate_rank(object):
    """Context manager (tests): behave like rank ``r`` of ``R`` in 'frames' mode WITHOUT a process
    group.  Collectives are identities, so what a model returns / accumulates inside is that
    rank's local contribution; the test adds the contributions of all ranks itself.  Terms that
    need other ranks' data inside the step (batch-norm statistics, the decomposed KL) cannot be
    emulated this way and raise."""

    def __init__(self, r, R):
        self.pair = (int(r), int(R))

    def __enter__(self):
        global _emulated
        self._prev, _emulated = _emulated, self.pair
        return self

    def __exit__(self, *exc):
        global _emulated
        _emulated = self._prev
        return False


def frames_sharded():
    """Is the current ``loss()`` call one rank's share of a frame-sharded step?"""
    if _emulated is not None:
        return _emulated[1] > 1
    return _mode == 'frames' and is_active() and world_size() > 1


def shard_rank_world():
    if _emulated is not None:
        return _emulated
    return rank(), world_size()


def shard_bounds(beg, end, r=None, R=None):
    """Contiguous slice of frames [beg, end) owned by rank r of R ('frames' mode)."""
    if r is None or R is None:
        r0, R0 = shard_rank_world()
        r = r0 if r is None else r
        R = R0 if R is None else R
    n = end - beg
    return beg + (r * n) // R, beg + ((r + 1) * n) // R


def shard_chunks(batch_size, chunk_size):
    """The reference's chunks of a batch (aes.py:748-753) and this rank's slice of each.

    -> (bounds, local, sizes): ``bounds`` the global [beg, end) of every chunk, ``local`` this
    rank's contiguous [beg, end) inside it (== bounds when not sharded), ``sizes`` the GLOBAL chunk
    lengths that normalise the chunk's loss terms (the chunk mean is the global one, so that the
    sum over ranks of the local gradients is the single-device gradient)."""
    bounds = [(b, min(b + chunk_size, batch_size)) for b in range(0, batch_size, chunk_size)]
    sizes = [e - b for b, e in bounds]
    if not frames_sharded():
        return bounds, list(bounds), sizes
    return bounds, [shard_bounds(b, e) for b, e in bounds], sizes


def _backend():
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def all_reduce_(t, op=None):
    """In-place sum over ranks of a (device or host) tensor; identity without a process group or
    under ``emulate_rank``.  Under gloo (CPU rendezvous, e.g. two test processes sharing one GPU)
    device tensors are staged through the host."""
    if _emulated is not None or not is_active():
        return t
    op = dist.ReduceOp.SUM if op is None else op
    if _backend() == 'gloo' and t.is_cuda:
        tmp = t.detach().cpu()
        dist.all_reduce(tmp, op=op)
        t.copy_(tmp)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_gather_rows(t):
    """Concatenation over ranks (rank order) of 2-d tensors that may differ in their row count.
    -> (gathered, row offset of this rank's block)."""
    if _emulated is not None:
        raise RuntimeError('emulate_rank cannot provide the other ranks\' rows (all-gather)')
    if not is_active() or world_size() == 1:
        return t, 0
    W = world_size()
    staged = _backend() == 'gloo' and t.is_cuda
    src = t.detach().cpu() if staged else t.detach().contiguous()
    counts = torch.zeros(W, dtype=torch.int64, device=src.device)
    counts[rank()] = src.shape[0]
    dist.all_reduce(counts)
    counts = [int(c) for c in counts.tolist()]
    parts = [torch.empty((c,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
             for c in counts]
    dist.all_gather(parts, src)
    out = torch.cat(parts, dim=0)
    if staged:
        out = out.to(t.device)
    return out, sum(counts[:rank()])


def all_reduce_flat_(flat, average=False):
    """In-place sum (or mean) of one flat buffer over all ranks; no-op for world_size 1."""
    if not is_active():
        return flat
    if flat.is_cuda:
        from behavenet_amd.hip_functions import join_side_streams
        join_side_streams()
    all_reduce_(flat)
    if average:
        flat.div_(world_size())
    return flat


class BucketedGradReducer(object):
    """All-reduce of the flat gradient arena in buckets, overlapped with the backward pass.

    The arena (``FlatAdamAMSGrad.flat_g``) holds the parameters in ``get_parameters()`` order --
    encoder first, decoder last -- and the backward pass finishes them in (nearly) the reverse
    order, so buckets are contiguous arena segments cut from the END of the arena every
    ``bucket_bytes``.  A bucket goes out as soon as the kernels that write its last gradient have
    been ISSUED (``hip_functions`` reports each parameter through ``grad_ready``): the collective
    is launched on a stream that waits for the main and the weight-gradient streams at that
    point, and RCCL moves it over xGMI while the rest of the backward pass computes.  Whatever
    was not reported (chunked schedules, gradients produced by plain autograd) is reduced by
    ``finish()``, in arena order, so every rank issues the same sequence of collectives.
    """

    def __init__(self, optimizer, bucket_bytes=None):
        self.opt = optimizer
        self.flat = optimizer.flat_g
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get('BN_BUCKET_MB', '4')) * (1 << 20))
        # cut from the end of the arena
        self.buckets = []            # [lo, hi) element ranges, in launch (= reverse arena) order
        self._bucket_of = {}
        hi, acc, members = self.flat.numel(), 0, []
        params = list(zip(optimizer.params, optimizer.offsets))
        for p, off in reversed(params):
            members.append(p)
            acc += p.numel() * 4
            if acc >= bucket_bytes or off == 0:
                self._add_bucket(off, hi, members)
                hi, acc, members = off, 0, []
        self._pending = []
        self._missing = None
        self._launched = None
        self._stream = None
        # overlap = False: nothing goes out during the backward pass, finish() launches the buckets
        # (in order) behind it.  A caller can switch per step: persistent kernels whose grids fill
        # the chip exactly (one or two LDS-filling workgroups per CU) lose a whole round when an
        # RCCL block holds a CU, so which is faster depends on how many channels RCCL uses --
        # bench.py measures both during its warm-up and keeps the faster one
        self.overlap = True
        self.begin()

    def _add_bucket(self, lo, hi, members):
        index = len(self.buckets)
        self.buckets.append((lo, hi, len(members)))
        for p in members:
            self._bucket_of[id(p)] = index

    def begin(self):
        """Start of a step (before any backward pass): nothing reported, nothing in flight."""
        self._drain()
        self._missing = [n for _, _, n in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen = set()
        self.n_overlapped = 0        # buckets of this step that went out during the backward pass
        if self.flat.is_cuda:
            from behavenet_amd import hip_functions as hf
            hf.reset_grad_ready()

    def grad_ready(self, p):
        """The kernels writing ``p.grad`` for this step have all been issued."""
        if not is_active() or not self.overlap:
            return
        b = self._bucket_of.get(id(p))
        if b is None or id(p) in self._seen or self._launched[b]:
            return
        self._seen.add(id(p))
        self._missing[b] -= 1
        # in order only: every rank must issue the same sequence of collectives
        while True:
            nxt = self._launched.index(False) if False in self._launched else None
            if nxt is None or self._missing[nxt] > 0:
                break
            self._launch(nxt, overlapped=True)

    def _launch(self, b, overlapped):
        lo, hi, _ = self.buckets[b]
        seg = self.flat[lo:hi]
        self._launched[b] = True
        self.n_overlapped += int(overlapped)
        if seg.is_cuda and overlapped:
            from behavenet_amd import hip_functions as hf
            dev = seg.device
            main = torch.cuda.current_stream(dev)
            ev_main = torch.cuda.Event()
            ev_main.record(main)
            side = hf._side_streams.get(dev.index)
            if os.environ.get('BN_COMM_VIA_SIDE', '0') == '1' and side is not None:
                launch = side
                launch.wait_event(ev_main)
            else:
                if self._stream is None:
                    self._stream = torch.cuda.Stream(device=dev)
                launch = self._stream
                launch.wait_event(ev_main)
                if side is not None:
                    ev_side = torch.cuda.Event()
                    ev_side.record(side)
                    launch.wait_event(ev_side)
            with torch.cuda.stream(launch):
                work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append(work)

    def _drain(self):
        for work in self._pending:
            work.wait()
        self._pending = []

    def finish(self):
        """Before ``optimizer.step()``: reduce what is left and wait for everything."""
        if not is_active():
            return
        if self.flat.is_cuda:
            from behavenet_amd.hip_functions import join_side_streams
            join_side_streams()
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b, overlapped=False)
        self._drain()


def attach_reducer(optimizer):
    """Give a flat-arena optimizer an overlapped bucketed reducer (BN_OVERLAP_ALLREDUCE=0: one
    flat all-reduce after the backward pass instead)."""
    if not is_active() or getattr(optimizer, 'flat_g', None) is None:
        return None
    if os.environ.get('BN_OVERLAP_ALLREDUCE', '1') == '0':
        return None
    if _backend() == 'gloo' and optimizer.flat_g.is_cuda:
        return None      # host-staged collectives (tests): one flat all-reduce after the backward
    reducer = BucketedGradReducer(optimizer)
    optimizer.reducer = reducer
    if optimizer.flat_g.is_cuda:
        from behavenet_amd import hip_functions as hf
        hf.set_grad_ready_callback(reducer.grad_ready)
    return reducer


def reduce_gradients(optimizer, average=False):
    """Sum (or average) the gradients over ranks before ``optimizer.step()``.

    'frames' mode sums: every rank holds its share of ONE trial's gradient.  'trial' mode averages:
    every rank holds the gradient of its own trial, and the mean keeps the step size and the
    weight decay (added by the optimizer after this) those of a single-trial step."""
    reducer = getattr(optimizer, 'reducer', None)
    if reducer is not None:
        reducer.finish()
        if average and is_active():
            optimizer.flat_g.div_(world_size())
    elif getattr(optimizer, 'flat_g', None) is not None:
        all_reduce_flat_(optimizer.flat_g, average=average)


def sharded_step(optimizer, average=False, divide_by=None):
    """The optimizer step of SURVEY.md section 8e's alternative: reduce-scatter of the flat gradient
    arena -> Adam on THIS rank's 1/R shard -> all-gather of the parameter arena, instead of
    all-reduce + R identical full steps.  Same bytes on the links (an all-reduce is a
    reduce-scatter and an all-gather), 1/R of the optimizer's 9 x 4 bytes per parameter of HBM
    traffic per rank; but the all-gather sits BEHIND the step where nothing is left to overlap it,
    while the bucketed all-reduce rides under the backward pass -- which is why it is an option
    (``hparams['shard_optimizer']``, ``BN_SHARD_OPTIMIZER=1``) and not the default.

    The update is element-wise, so the parameters come out as after the replicated step wherever
    the two reductions add the ranks' terms in the same order (two ranks: always).  Needs an
    optimizer built with ``shard_over == world_size()``; the moments of the other ranks' shards
    are never touched on this rank.

    What a caller must know: after the step ``flat_g`` holds the reduced gradient in
    THIS rank's shard only (the rest is this rank's local gradient, or what gloo's reduce left
    there) -- gradient norms and the like must be taken before the step or over the owned shard;
    and every rank holds 1/R of the Adam moments: ``gather_optimizer_state_`` makes them whole
    before an optimizer checkpoint or a switch to the replicated step (``fit`` saves the model's
    ``state_dict`` only, like the reference, training.py:390)."""
    if not is_active() or world_size() == 1 or _emulated is not None:
        reduce_gradients(optimizer, average=average)
        if divide_by is not None:
            optimizer.flat_g.div_(float(divide_by))
        optimizer.step()
        return
    W, r = world_size(), rank()
    if getattr(optimizer, 'shard_over', 1) != W:
        raise ValueError('sharded_step: the optimizer was built for %d shards, the group has %d '
                         'ranks' % (getattr(optimizer, 'shard_over', 1), W))
    flat_g, flat_p = optimizer.flat_g, optimizer.flat_p
    if flat_g.is_cuda:
        from behavenet_amd.hip_functions import join_side_streams
        join_side_streams()
    lo, hi = optimizer.shard_range(r)
    staged = _backend() == 'gloo' and flat_g.is_cuda
    if _backend() == 'nccl':
        dist.reduce_scatter_tensor(flat_g[lo:hi], flat_g, op=dist.ReduceOp.SUM)
    else:
        # gloo has no reduce-scatter: one reduce per shard to its owner (the same sums)
        src = flat_g.detach().cpu() if staged else flat_g
        for q in range(W):
            qlo, qhi = optimizer.shard_range(q)
            dist.reduce(src[qlo:qhi], dst=q, op=dist.ReduceOp.SUM)
        if staged:
            flat_g[lo:hi].copy_(src[lo:hi])
    if average:
        flat_g[lo:hi].div_(W)
    if divide_by is not None:
        flat_g[lo:hi].div_(float(divide_by))
    optimizer.step_range(lo, hi)
    if _backend() == 'nccl':
        dist.all_gather_into_tensor(flat_p, flat_p[lo:hi].clone())
    else:
        mine = flat_p[lo:hi].detach().cpu() if staged else flat_p[lo:hi].detach().clone()
        parts = [torch.empty_like(mine) for _ in range(W)]
        dist.all_gather(parts, mine)
        with torch.no_grad():
            for q, part in enumerate(parts):
                qlo, qhi = optimizer.shard_range(q)
                flat_p[qlo:qhi].copy_(part)


def gather_optimizer_state_(optimizer):
    """All-gather the Adam moments of a sharded optimizer so that every rank holds the whole state
    (before saving it, or before continuing with replicated steps / another world size)."""
    if not is_active() or world_size() == 1 or getattr(optimizer, 'shard_over', 1) == 1:
        return optimizer
    lo, hi = optimizer.shard_range(rank())
    for arena in (optimizer.exp_avg, optimizer.exp_avg_sq, optimizer.max_exp_avg_sq):
        if _backend() == 'nccl':
            dist.all_gather_into_tensor(arena, arena[lo:hi].clone())
        else:
            staged = arena.is_cuda
            mine = arena[lo:hi].detach().cpu() if staged else arena[lo:hi].detach().clone()
            parts = [torch.empty_like(mine) for _ in range(world_size())]
            dist.all_gather(parts, mine)
            for q, part in enumerate(parts):
                qlo, qhi = optimizer.shard_range(q)
                arena[qlo:qhi].copy_(part)
    return optimizer


def default_shard_optimizer(world=None, mode=None):
    """Is the sharded optimizer step (``sharded_step``) the default for this job?

    ``BN_SHARD_OPTIMIZER=0/1`` decides when set.  Otherwise: ON for frame sharding ('frames') over
    four or more ranks, OFF elsewhere.  Why there: a rank of an 8-rank frame-sharded step runs 32
    frames in ~1 ms, of which the replicated Adam(amsgrad) -- 9 streams x 35 MB, 49 us, the same on
    every rank whatever the shard -- is 4-5 %; on 1/R of the arena it is 6 us (R = 8), and the
    backward pass of so small a shard is too short to hide the bucketed all-reduce behind anyway.
    In 'trial' mode (one whole trial per rank, 4.4 ms steps) the overlapped all-reduce is hidden
    completely and the all-gather behind a sharded step would be exposed: replicated stays."""
    env = os.environ.get('BN_SHARD_OPTIMIZER')
    if env is not None:
        return env == '1'
    world = world_size() if world is None else world
    mode = shard_mode() if mode is None else mode
    return mode == 'frames' and world >= 4


def all_reduce_scalars(values):
    """Sum a short list of python floats over ranks (loss bookkeeping)."""
    if not is_active() or _emulated is not None:
        return list(values)
    dev = torch.device('cuda', torch.cuda.current_device()) \
        if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


def broadcast_object(obj, src=0):
    """A small python object from rank ``src`` to everyone."""
    if not is_active() or _emulated is not None:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_parameters_(flat, src=0):
    """Make every rank start from rank ``src``'s parameters."""
    if is_active():
        if _backend() == 'gloo' and flat.is_cuda:
            tmp = flat.detach().cpu()
            dist.broadcast(tmp, src=src)
            flat.copy_(tmp)
        else:
            dist.broadcast(flat, src=src)
    return flat
