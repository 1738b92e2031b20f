"""autograd glue between the model classes and the HIP kernels.

Each ``torch.autograd.Function`` here stands for one fused block of the reference network and
calls the C ABI (behavenet_amd/_hip.py) for both directions; torch autograd only chains them.

* :class:`ConvStackFn` -- a whole encoder or decoder convolution stack
  (ZeroPad2d/Conv2d/LeakyReLU x L, or ConvTranspose2d/crop/LeakyReLU|Sigmoid x L;
  reference aes.py:203-211 and aes.py:460-476).  The backward pass is orchestrated by hand so
  that the activation derivative of layer l-1 is applied in the epilogue of layer l's data
  gradient kernel instead of in a separate pass.
* :class:`LinearFn` -- nn.Linear on the matrix cores (aes.py:121,125,266).
* :class:`SqErrFn` -- masked sum of squared differences (losses.py:56-59,84-96).
* :class:`ReparamFn`, :class:`KLFn` -- the variational tail (vaes.py:33-35, losses.py:146-147).
"""

import collections.abc
import os

import torch

from behavenet_amd import _hip

LRELU_SLOPE = 0.05  # aes.py:114,341

# When set (by the models' loss() around loss.backward()), weight/bias gradients are accumulated
# by the kernels straight into existing `param.grad` buffers (accumulate=1 of the C ABI) and the
# autograd node returns None for them: this is the reference's cross-chunk `+=` without the
# temporary + add-kernel per parameter.  Off by default so torch.autograd.grad() keeps working.
_direct_grads = False


class accumulate_into_param_grads(object):
    """Context manager: let backward kernels add into `param.grad` in place."""

    def __enter__(self):
        global _direct_grads
        self._prev = _direct_grads
        _direct_grads = True

    def __exit__(self, *exc):
        global _direct_grads
        _direct_grads = self._prev
        return False


# Single-pass schedule only (ONE backward per step): number of bottom layers of the LAST stack in
# the backward order (the encoder) whose weight gradients are issued on the main stream, after
# its data-gradient chain, instead of on the side stream.  The side stream lags the main one, so
# without this the last ~0.6 ms of a step run one weight-gradient kernel at a time.
_tail_on_main = 0
_TAIL_LAYERS = int(os.environ.get('BN_WGRAD_TAIL', '2'))
_SINGLE_PASS_SIDE = os.environ.get('BN_SINGLE_PASS_SIDE', '0') == '1'

# Data-parallel runs: fitting/distributed.BucketedGradReducer asks to be told when the kernels
# that complete a parameter's gradient have been issued, so that it can start that bucket's
# all-reduce under the rest of the backward pass.  Reported only in the single-pass schedule (one
# backward per step) and only for gradients the kernels write in place; a parameter used by
# several nodes of one graph is reported by the last of them.
_single_pass = False
_grad_ready_cb = None
_param_uses = {}


def set_grad_ready_callback(fn):
    global _grad_ready_cb
    _grad_ready_cb = fn
    _param_uses.clear()


def reset_grad_ready():
    _param_uses.clear()


def _note_use(params, needed):
    """Called from Function.forward: `needed` is false under no_grad (no backward will come)."""
    if _grad_ready_cb is not None and needed and not _capturing:
        for p in params:
            if p is not None:
                _param_uses[id(p)] = _param_uses.get(id(p), 0) + 1


def _report_ready(params):
    # (a step recorded into a HIP graph reports nothing: its gradients are exchanged after the
    # replay, by the reducer's flat path)
    if _grad_ready_cb is None or not _single_pass or _capturing:
        return
    for p in params:
        if p is None:
            continue
        left = _param_uses.get(id(p), 0) - 1
        _param_uses[id(p)] = left
        if left == 0:
            _grad_ready_cb(p)


_ones_cache = {}


def _ones_like(t):
    key = (t.device, tuple(t.shape))
    o = _ones_cache.get(key)
    if o is None:
        o = torch.ones_like(t.detach())
        _ones_cache[key] = o
    return o


def backward_chunks(chunk_losses, streams=None, single_pass=False):
    """Run the per-chunk backward passes, in chunk order, after ALL chunk forwards.

    The reference interleaves forward and backward per chunk (aes.py:748-769); the parameters do
    not change in between, so running the forwards first gives bit-identical gradients (same
    kernels, same accumulation order) while the forwards run with the weight-gradient side
    stream idle and the tail of chunk c's weight gradients overlaps the head of chunk c+1's data
    gradients.
    """
    global _tail_on_main, _single_pass, _use_side_stream
    _single_pass = bool(single_pass and len(chunk_losses) == 1)
    _tail_on_main = _TAIL_LAYERS if _single_pass else 0
    # single-pass schedule: every kernel sees the whole batch and fills the chip on its own; a
    # second stream of weight-gradient kernels then only competes for LDS and wave slots (5.45 ->
    # 5.43 ms for the AE, 6.39 -> 6.19 ms for the PS-VAE).  BN_SINGLE_PASS_SIDE=1 keeps it.
    saved_side = _use_side_stream
    if _single_pass and not _SINGLE_PASS_SIDE:
        _use_side_stream = False
    try:
        with accumulate_into_param_grads():
            for i, loss in enumerate(chunk_losses):
                # called from the chunk's own stream: autograd orders the graph's streams after
                # the CALLING stream, so a backward() issued from the main stream would make the
                # auxiliary pipeline wait for everything the previous chunk queued there
                stream = streams[i] if streams is not None and i < len(streams) else None
                # a vector of per-chunk losses stands for their sum: seeding its backward with
                # ones saves the sum kernel and its expand in the graph
                seed = _ones_like(loss) if loss.dim() > 0 else None
                if stream is not None:
                    with torch.cuda.stream(stream):
                        loss.backward(seed)
                else:
                    loss.backward(seed)
    finally:
        _tail_on_main = 0
        _single_pass = False
        _use_side_stream = saved_side


# Graph capture (fitting/graph_step.py): while a step is being recorded into a HIP graph nothing
# may wait on the host.  `Readback` then only remembers its device tensor (whose address is fixed
# inside the graph's memory pool) and `finish_loss` returns a `DeferredLoss` instead of the loss
# dict: after every replay the recorded tensors are read back and the same host function turns
# them into the dict.
_capturing = False


def capturing():
    return _capturing


class DeferredLoss(object):
    """What ``model.loss`` returns while a step is being captured: the device tensors its
    read-backs would have copied and the host function that turns their values into the dict."""

    def __init__(self, tensors, fn, reduce_over_ranks=False):
        self.tensors, self.fn = tensors, fn
        # the tensors are this rank's terms of a frame-sharded step: whoever replays the recording
        # sums them over ranks (one collective behind the replay) before reading them back
        self.reduce_over_ranks = bool(reduce_over_ranks)


class LazyLoss(collections.abc.Mapping):
    """A loss dict whose values are still on their way from the device: fetched (one event wait) when
    somebody first looks at it.  ``fit``'s logger adds such dicts up a few steps later, so the host
    never waits for a step it has just launched (fitting/training.py, Logger.update_metrics)."""

    def __init__(self, readbacks, fn):
        self._rbs, self._fn, self._dict = readbacks, fn, None

    def resolve(self):
        if self._dict is None:
            self._dict = self._fn(*[None if rb is None else rb.numpy() for rb in self._rbs])
            self._rbs = self._fn = None
        return self._dict

    def __getitem__(self, key):
        return self.resolve()[key]

    def __iter__(self):
        return iter(self.resolve())

    def __len__(self):
        return len(self.resolve())


# Eager steps: ``loss`` returns a plain dict (it waits for the read-back of the forward's chunk losses
# after it has queued the backward launches -- the host is at most one backward pass ahead of the
# device, and a host hiccup longer than that idles the device).  With lazy losses on (``fit``,
# bench.py) it returns a LazyLoss instead and the host runs ahead as far as the HIP queue lets it.
_lazy_losses = False


def set_lazy_losses(flag):
    """-> the previous setting."""
    global _lazy_losses
    prev, _lazy_losses = _lazy_losses, bool(flag)
    return prev


def finish_loss(readbacks, fn, reduce_over_ranks=False):
    """``fn(*arrays)`` -> loss dict, with arrays = the values of `readbacks` (None entries stay
    None).  Eager: waits for the read-backs and calls `fn` (or hands out a LazyLoss that will, see
    set_lazy_losses); under capture: defers both.  ``reduce_over_ranks``: only read under capture --
    the eager caller has summed its tensors over ranks already, a recording must not."""
    if _capturing:
        return DeferredLoss([None if rb is None else rb.tensor for rb in readbacks], fn,
                            reduce_over_ranks)
    if _lazy_losses:
        return LazyLoss(readbacks, fn)
    return fn(*[None if rb is None else rb.numpy() for rb in readbacks])


class Readback(object):
    """Asynchronous device -> host copy of a small tensor through a pooled pinned buffer.

    ``Readback(t)`` enqueues the copy on the current stream; ``.numpy()`` waits for THAT copy only
    (an event), not for work enqueued afterwards.  The models start the read-back of the chunk
    losses after the forwards, enqueue every backward launch, and only then wait: the value is
    long there, and the host returns with the backward kernels still queued, so the optimizer
    step and the next batch's forward follow them without a bubble.
    """
    _pool = {}
    _dropped = {}

    def __init__(self, t):
        t = t.detach()
        self.tensor = t
        if _capturing:
            return
        self._key = (t.dtype, tuple(t.shape))
        free = Readback._pool.setdefault(self._key, [])
        if not free:
            # buffers of read-backs nobody looked at (a LazyLoss that was dropped): back once their copy is done
            flying = Readback._dropped.get(self._key)
            while flying and flying[0][0].query():
                free.append(flying.pop(0)[1])
        self._buf = free.pop() if free else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self._buf.copy_(t, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def __del__(self):
        buf = getattr(self, '_buf', None)
        if buf is not None and Readback is not None:
            Readback._dropped.setdefault(self._key, []).append((self._event, buf))
            self._buf = None

    def numpy(self):
        if _capturing:
            raise RuntimeError('a loss value was asked for while the step is being captured '
                               'into a HIP graph: route the host tail through finish_loss()')
        self.tensor = None
        self._event.synchronize()
        out = self._buf.numpy().copy()
        Readback._pool[self._key].append(self._buf)
        self._buf = None
        return out


_use_side_stream = os.environ.get('BN_SIDE_STREAM', '1') != '0'
_side_streams = {}

# Chunk pipelines: the 200-frame chunks of one batch are independent until their gradients meet
# in `param.grad`.  Odd chunks run (forward and, through autograd's stream affinity, backward)
# on an auxiliary HIP stream so that the workgroups of the 56-frame chunk fill the tails the
# 200-frame chunk leaves on the 256 CUs.  Every in-place gradient accumulation goes through the
# single weight-gradient side stream in issue order (chunk 0 first), so the result is the same
# sum in the same order as the one-stream schedule.
_use_chunk_streams = os.environ.get('BN_CHUNK_STREAMS', '1') != '0'
_aux_streams = {}
_chunk_epoch = {}


def begin_chunks(device):
    """Mark the point on the current stream that the auxiliary chunk stream has to wait for."""
    if not _use_chunk_streams or torch.device(device).type != 'cuda':
        return      # (a CPU tensor fails loudly in the first kernel call, not here)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    _chunk_epoch[torch.device(device).index] = ev


class chunk_stream(object):
    """Context manager: run the enclosed chunk on its pipeline's stream."""

    def __init__(self, index, device, enabled=True):
        self._ctx = None
        key = torch.device(device).index
        if enabled and _use_chunk_streams and _use_side_stream and (index % 2) == 1 \
                and key in _chunk_epoch:
            aux = _aux_streams.get(key)
            if aux is None:
                aux = torch.cuda.Stream(device=device)
                _aux_streams[key] = aux
            aux.wait_event(_chunk_epoch[key])
            self._ctx = torch.cuda.stream(aux)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            return self._ctx.__exit__(*exc)
        return False


def reserve_device_pools(model, frames_per_batch, device='cuda', headroom=6.0):
    """Grow torch's caching-allocator pools of the main and the auxiliary chunk stream ONCE, up
    front, to `headroom` x the activation bytes of one batch.

    Which cached block a tensor gets depends on when the side streams release theirs, so without
    this the pools keep growing by an occasional hipMalloc (a 30-60 ms device-wide stall) for the
    first dozens of steps.  288 GB of HBM make the reservation free: cfg2 at batch 256 asks for
    ~6 GB + ~3 GB.
    """
    per_frame = 0
    for part in (getattr(model, 'encoding', None), getattr(model, 'decoding', None)):
        for layer in getattr(part, '_plan', []) or []:
            per_frame += 4 * (layer.cin * layer.hin * layer.win + layer.cout * layer.hout * layer.wout)
    if per_frame == 0 or torch.device(device).type != 'cuda':
        return 0
    main_bytes = int(headroom * per_frame * frames_per_batch)
    blocks = [torch.empty(main_bytes, dtype=torch.uint8, device=device)]
    if _use_chunk_streams:
        key = torch.device(device).index
        if key is None:
            key = torch.cuda.current_device()
        aux = _aux_streams.get(key)
        if aux is None:
            aux = torch.cuda.Stream(device=device)
            _aux_streams[key] = aux
        with torch.cuda.stream(aux):
            blocks.append(torch.empty(main_bytes // 2, dtype=torch.uint8, device=device))
    total = sum(b.numel() for b in blocks)
    del blocks
    return total


class ChunkScalars(object):
    """The per-chunk loss scalars of one ``loss()`` call, still on the device.

    ``add(t)`` is called inside the chunk's stream context with a small 1-d tensor; ``finish``
    enqueues one read-back per chunk on that chunk's stream (after ALL forwards have been
    enqueued, so that no device->host copy sits between two forwards), then the deferred
    backwards, joins every stream and returns the values as a float64 (n_chunks, n_values) array.
    """

    def __init__(self):
        self._items = []

    def add(self, t):
        self._items.append((torch.cuda.current_stream(t.device), t.detach()))

    def finish(self, chunk_losses):
        import numpy as np
        rbs = []
        for stream, t in self._items:
            with torch.cuda.stream(stream):
                rbs.append(Readback(t))
        # chunk c's backward is issued from chunk c's stream (only when every chunk has one,
        # i.e. gradients were requested for all of them)
        streams = [s for s, _ in self._items] if len(chunk_losses) == len(self._items) else None
        backward_chunks(chunk_losses, streams)
        join_chunk_streams()
        join_side_streams()
        return np.stack([r.numpy() for r in rbs]).astype(np.float64)


def join_chunk_streams():
    """Make the current stream wait for the auxiliary chunk stream(s)."""
    for key, s in _aux_streams.items():
        torch.cuda.current_stream(s.device).wait_stream(s)


def _side_stream(device):
    key = torch.device(device).index
    s = _side_streams.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side_streams[key] = s
    return s


def join_side_streams():
    """Make the current stream wait for all weight-gradient work queued on the side streams.

    Called by the models at the end of ``loss()`` and by the optimizer before ``step()``.
    """
    for key, s in _side_streams.items():
        torch.cuda.current_stream(s.device).wait_stream(s)


def _grad_buffer(p):
    """`p.grad` if the kernels may accumulate into it directly, else None."""
    if not _direct_grads or not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or not g.is_cuda:
        return None
    return g


class ConvLayerPlan(object):
    """Geometry of one fused layer, independent of the batch size.

    kind 'conv':  (C, H, W) -> (K, P, Q), zero padding (pad_t, pad_l) folded into the kernel
    kind 'convT': (Ci, Hi, Wi) -> (Co, Ho, Wo), crop (crop_t, crop_l) folded into the kernel
    """

    __slots__ = ('kind', 'cin', 'hin', 'win', 'cout', 'hout', 'wout', 'R', 'S', 'stride',
                 'off_t', 'off_l', 'act')

    def __init__(self, kind, cin, hin, win, cout, hout, wout, R, S, stride, off_t, off_l, act):
        self.kind = kind
        self.cin, self.hin, self.win = int(cin), int(hin), int(win)
        self.cout, self.hout, self.wout = int(cout), int(hout), int(wout)
        self.R, self.S, self.stride = int(R), int(S), int(stride)
        self.off_t, self.off_l = int(off_t), int(off_l)
        self.act = act

    def geom(self, n):
        """Argument tuple in the order of include/behavenet_hip.h."""
        return (int(n), self.cin, self.hin, self.win, self.cout, self.R, self.S, self.stride,
                self.off_t, self.off_l, self.hout, self.wout)

    def with_act(self, act):
        return ConvLayerPlan(self.kind, self.cin, self.hin, self.win, self.cout, self.hout,
                             self.wout, self.R, self.S, self.stride, self.off_t, self.off_l, act)

    def __repr__(self):
        return '%s(%dx%dx%d->%dx%dx%d k%dx%d s%d off(%d,%d) act%d)' % (
            self.kind, self.cin, self.hin, self.win, self.cout, self.hout, self.wout, self.R,
            self.S, self.stride, self.off_t, self.off_l, self.act)


def _fwd(layer, x, w, b, w5=None):
    g = layer.geom(x.shape[0])
    if layer.kind == 'conv':
        if x.dtype == torch.uint8:
            # frames as stored on disk: value / 255 is fused into the first layer's patch load
            return _hip.conv2d_fwd_u8(x, w, b, g, layer.act, LRELU_SLOPE)
        return _hip.conv2d_fwd(x, w, b, g, layer.act, LRELU_SLOPE, w5=w5)
    return _hip.convT2d_fwd(x, w, b, g, layer.act, LRELU_SLOPE, w5=w5)


_STACK_TAPS = os.environ.get('BN_STACK_TAPS', '1') != '0'      # 0: every entry point pads for itself (A/B switch)
_FWD_OP = {'conv': _hip.OP_CONV_FWD, 'convT': _hip.OP_CONVT_FWD}
_BWD_OP = {'conv': _hip.OP_CONV_BWD_D, 'convT': _hip.OP_CONVT_BWD_D}


def _stack_taps(plan, n, params, first=0):
    """Layers with kernels smaller than 5x5 run on the 5x5 kernel families with their taps embedded in 5x5 ones
    (csrc/capi.hip, taps_plan).  The entry points make that copy per call -- once for the forward pass and once for
    the data gradient of every such layer, 19 launches in a step of ae_arch_2.json; a stack makes the copies of ALL
    its layers in one launch when its forward pass starts and keeps them for its backward pass (same weights: the
    node's saved tensors).  -> [w5 | None] per layer, None for a stack without such layers."""
    if not _STACK_TAPS:
        return None
    jobs, idx = [], []
    for i in range(first, len(plan)):
        layer = plan[i]
        g = layer.geom(n)
        for op in (_FWD_OP[layer.kind], _BWD_OP[layer.kind]):
            if _hip.conv_taps_bytes(op, g):
                jobs.append((op, g, params[2 * i].detach()))
                idx.append(i)
                break
    if not jobs:
        return None
    taps = [None] * len(plan)
    for i, w5 in zip(idx, _hip.conv_taps_pad(jobs, params[0].device)):
        taps[i] = w5
    return taps


def first_layer_forward(plan, x, params):
    """Layer 1 of a stack for a whole batch, outside autograd (see ConvStackFn's `h1`)."""
    return _fwd(plan[0], x.contiguous(), params[0].detach(), params[1].detach())


# Tests only (tests/branches.py): when a dict, every ConvStackFn.forward records the sign of its
# LeakyReLU outputs, keyed by id(plan), one list per layer in call order.
_sign_tap = None


class ConvStackFn(torch.autograd.Function):
    """y = layer_L(...layer_1(x)); params = (w_1, b_1, ..., w_L, b_L).

    ``h1`` (optional): the output of layer 1 for these frames, computed beforehand for the whole
    batch in one launch -- the frames are independent, so a slice of it is bit-identical to what
    this node would compute; the chunk then starts at layer 2 (backward is unchanged).
    """

    @staticmethod
    def forward(ctx, plan, x, h1, *params):
        if x.shape[1:] != (plan[0].cin, plan[0].hin, plan[0].win):
            raise ValueError('conv stack expects input (N,%d,%d,%d), got %s' % (
                plan[0].cin, plan[0].hin, plan[0].win, tuple(x.shape)))
        x = x.contiguous()
        acts = [x]
        h = x
        skip_first = h1 is not None and not x.requires_grad
        taps = _stack_taps(plan, x.shape[0], params, first=1 if skip_first else 0) if x.is_cuda else None
        for i, layer in enumerate(plan):
            if i == 0 and h1 is not None:
                h = h1
            else:
                h = _fwd(layer, h, params[2 * i].detach(), params[2 * i + 1].detach(),
                         w5=taps[i] if taps else None)
            acts.append(h)
        if _sign_tap is not None:
            rec = _sign_tap.setdefault(id(plan), [[] for _ in plan])
            for i, layer in enumerate(plan):
                if layer.act == _hip.ACT_LRELU:
                    rec[i].append((acts[i + 1] > 0).cpu())
        ctx.plan = plan
        ctx.taps = taps
        ctx.need_dx = x.requires_grad
        ctx.param_refs = params
        _note_use(params, any(ctx.needs_input_grad[3:]))
        ctx.save_for_backward(*acts, *params[0::2])
        return h

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        dout = dout.contiguous()
        top = plan[-1]
        if top.act != _hip.ACT_NONE:
            dpre = _hip.act_bwd(dout, ctx.saved_tensors[len(plan)], top.act, LRELU_SLOPE)
        else:
            dpre = dout
        dx, grads = _stack_backward(ctx, dpre, first_param=3)
        return (None, dx, None) + tuple(grads)


_DGRAD_FIRST = os.environ.get('BN_DGRAD_FIRST', '0') == '1'


def _stack_backward(ctx, dpre, first_param):
    """Backward pass of a fused conv stack from ``dpre`` = dL/d(pre-activation of the top layer):
    per layer the weight (+bias) gradient, then the data gradient with the activation derivative
    of the layer below fused into its epilogue.  -> (dx | None, [dw_1, db_1, ..., dw_L, db_L])
    with None for gradients the kernels accumulated straight into ``param.grad``."""
    plan = ctx.plan
    n_layers = len(plan)
    saved = ctx.saved_tensors
    acts, weights = saved[:n_layers + 1], saved[n_layers + 1:2 * n_layers + 1]
    n = dpre.shape[0]
    grads = [None] * (2 * n_layers)
    taps = getattr(ctx, 'taps', None)       # the forward pass's 5x5 copies of small-kernel taps (_stack_taps)
    tail = []     # weight gradients deferred to the main stream (see _tail_on_main)
    for i in range(n_layers - 1, -1, -1):
        layer = plan[i]
        g = layer.geom(n)
        w = weights[i]
        x_in = acts[i]
        if x_in.dtype == torch.uint8:
            # uint8 frames went straight into the first layer; its weight gradient multiplies the
            # float values (the only place a float copy of the batch is made)
            x_in = _hip.u8_to_unit_float(x_in) \
                if ctx.needs_input_grad[first_param + 2 * i] else None
        need_w = ctx.needs_input_grad[first_param + 2 * i]
        need_b = ctx.needs_input_grad[first_param + 1 + 2 * i]
        # BN_DGRAD_FIRST=1: data gradient FIRST when the layer below is a single-channel edge layer,
        # so that this layer's matrix-bound weight gradient runs between the 134 MB of (non-temporal)
        # data-gradient writes and the HBM-bound weight gradient that reads them back.  Measured
        # (round 4, one box): no difference (35.2 against 35.1 us for enc.conv0's weight gradient,
        # step 4.330 against 4.337 ms); that kernel takes 35 us on some boxes and 41 on others in
        # either order.  Off by default.
        dpre_below = None
        if i > 0 and min(plan[i - 1].cin, plan[i - 1].cout) <= 4 and _DGRAD_FIRST:
            dact_src, dact = acts[i], plan[i - 1].act
            bwd_data = _hip.conv2d_bwd_data if layer.kind == 'conv' else _hip.convT2d_bwd_data
            dpre_below = bwd_data(dpre, w, g, dact_src, dact, LRELU_SLOPE, w5=taps[i] if taps else None)
        if need_w:
            gw = _grad_buffer(ctx.param_refs[2 * i])
            gb = _grad_buffer(ctx.param_refs[2 * i + 1]) if need_b else None
            direct = gw is not None and (gb is not None or not need_b)
            if direct:
                dw, db = gw, gb
            else:
                dw = torch.empty_like(w)
                db = torch.empty((layer.cout,), dtype=w.dtype, device=w.device) if need_b \
                    else None
            wgrad = _hip.conv2d_bwd_weight if layer.kind == 'conv' else _hip.convT2d_bwd_weight
            side = _side_stream(w.device) if (direct and _use_side_stream) else None
            if side is not None and not ctx.need_dx and i < _tail_on_main:
                tail.append(((wgrad, x_in, dpre, dw, db, g), ctx.param_refs[2 * i:2 * i + 2]))
            elif side is not None:
                # weight gradients go to a second HIP stream: they only depend on dpre and the
                # saved input, while the main stream continues down the data-gradient chain;
                # the tail of one kernel is filled by workgroups of the other.  Gradients
                # land in param.grad in stream order; join_side_streams() publishes them.
                main = torch.cuda.current_stream(w.device)
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    wgrad(x_in, dpre, dw, db, g, True)
                dpre.record_stream(side)
                x_in.record_stream(side)
                _report_ready(ctx.param_refs[2 * i:2 * i + 2])
            else:
                wgrad(x_in, dpre, dw, db, g, direct)
                if direct:
                    _report_ready(ctx.param_refs[2 * i:2 * i + 2])
            if not direct:
                grads[2 * i], grads[2 * i + 1] = dw, db
        if dpre_below is not None:
            dpre = dpre_below
        elif i > 0 or ctx.need_dx:
            # fuse the derivative of the layer below into this kernel's epilogue
            dact_src = acts[i] if i > 0 else None
            dact = plan[i - 1].act if i > 0 else _hip.ACT_NONE
            w5 = taps[i] if taps else None
            if layer.kind == 'conv':
                dpre = _hip.conv2d_bwd_data(dpre, w, g, dact_src, dact, LRELU_SLOPE, w5=w5)
            else:
                dpre = _hip.convT2d_bwd_data(dpre, w, g, dact_src, dact, LRELU_SLOPE, w5=w5)
    for (wgrad, x_in, dy, dw, db, g), refs in tail:
        wgrad(x_in, dy, dw, db, g, True)
        _report_ready(refs)
    return (dpre if ctx.need_dx else None), grads


def conv_stack(plan, x, params, h1=None):
    return ConvStackFn.apply(plan, x, h1, *params)


_frame_scale_cache = {}


def _frame_scales(n, bounds, scales, device):
    """(per-frame loss scale (N,) float32, chunk index of every frame (N,) int32) on `device`."""
    key = (torch.device(device), int(n), tuple(bounds), tuple(float(s) for s in scales))
    hit = _frame_scale_cache.get(key)
    if hit is None:
        fs = torch.zeros(n, dtype=torch.float32)
        ci = torch.zeros(n, dtype=torch.int32)
        for c, ((beg, end), sc) in enumerate(zip(bounds, scales)):
            fs[beg:end] = float(sc)
            ci[beg:end] = c
        hit = (fs.to(device), ci.to(device))
        if len(_frame_scale_cache) > 64:
            _frame_scale_cache.clear()
        _frame_scale_cache[key] = hit
    return hit


class ConvStackSqErrFn(torch.autograd.Function):
    """A decoder stack whose LAST layer is fused with the pixel loss (csrc: ``k_up_c1v<R, true>``;
    reference aes.py:460-476 + losses.py:56-59 / 84-96):

        out[c] = scales[c] * sum_{frames of chunk c} sum_pixels (x_hat - target)^2 * mask

    for the contiguous frame ranges ``bounds``; also returns ``x_hat`` if ``want_xhat`` (not
    differentiable through this node), else None.  The forward kernel already leaves dL/dpre of
    the last layer (up to the per-chunk factor), so the backward pass starts at the top layer's
    weight gradient: no sigmoid / squared-error backward kernels, and x_hat is never re-read.
    """

    @staticmethod
    def forward(ctx, plan, x, target, mask, bounds, scales, want_xhat, *params):
        if plan[-1].kind != 'convT':
            raise ValueError('the fused pixel loss follows a transposed convolution')
        x = x.contiguous()
        target = target.contiguous()
        mask = mask.contiguous() if mask is not None else None
        acts = [x]
        h = x
        taps = _stack_taps(plan, x.shape[0], params) if x.is_cuda else None
        for i, layer in enumerate(plan[:-1]):
            h = _fwd(layer, h, params[2 * i].detach(), params[2 * i + 1].detach(), w5=taps[i] if taps else None)
            acts.append(h)
        top = plan[-1]
        n = h.shape[0]
        xhat, dpre, part = _hip.convT2d_fwd_sqerr(
            h, params[-2].detach(), params[-1].detach(), target, mask, top.geom(n), top.act,
            LRELU_SLOPE, bool(want_xhat))
        if _sign_tap is not None:
            rec = _sign_tap.setdefault(id(plan), [[] for _ in plan])
            for i, layer in enumerate(plan[:-1]):
                if layer.act == _hip.ACT_LRELU:
                    rec[i].append((acts[i + 1] > 0).cpu())
        out = torch.empty((len(bounds),), dtype=torch.float32, device=x.device)
        for c, ((beg, end), sc) in enumerate(zip(bounds, scales)):
            _hip.reduce_sum(part[beg:end], float(sc), out=out[c:c + 1])
        ctx.plan = plan
        ctx.taps = taps
        ctx.need_dx = x.requires_grad
        ctx.param_refs = params
        ctx.bounds, ctx.scales = list(bounds), [float(sc) for sc in scales]
        ctx.consumed = False
        _note_use(params, any(ctx.needs_input_grad[7:]))
        # slot n_layers of `acts` (the top layer's output) is not needed by the backward pass:
        # dpre stands in for it so that _stack_backward's indexing stays the ConvStackFn one
        ctx.save_for_backward(*acts, dpre, *params[0::2])
        if xhat is not None:
            ctx.mark_non_differentiable(xhat)
        return out, xhat

    @staticmethod
    def backward(ctx, g, _g_xhat):
        if ctx.consumed:
            raise RuntimeError('ConvStackSqErrFn: backward twice (dL/dpre is scaled in place)')
        ctx.consumed = True
        n_layers = len(ctx.plan)
        dpre = ctx.saved_tensors[n_layers]
        fs, ci = _frame_scales(dpre.shape[0], ctx.bounds, ctx.scales, dpre.device)
        _hip.scale_frames(dpre, fs, g.contiguous(), ci)
        dx, grads = _stack_backward(ctx, dpre, first_param=7)
        return (None, dx, None, None, None, None, None) + tuple(grads)


def conv_stack_sq_err(plan, x, params, target, mask, bounds, scales, want_xhat=False):
    """-> (per-chunk scaled squared-error sums (n_chunks,), x_hat | None)."""
    return ConvStackSqErrFn.apply(plan, x, target, mask, tuple(bounds), tuple(scales),
                                  bool(want_xhat), *params)


class FusedPixelLoss(object):
    """What a decoder returns in place of ``x_hat`` when it was asked for the pixel loss of a
    batch (``pixel_loss=`` argument): the per-chunk loss terms (already normalised:
    ``kind='mse'`` -> mean squared error of each chunk, ``kind='ll'`` -> the data-dependent part
    of ``losses.gaussian_ll``) and ``x_hat`` itself only if it was requested."""

    def __init__(self, x_hat, chunk_terms, kind, bounds):
        self.x_hat = x_hat
        self.chunk_terms = chunk_terms
        self.kind = kind
        self.bounds = list(bounds)


_DEVICE_CONSTANTS = {}


def device_constant(values, device):
    """A small float32 device tensor holding ``values``, built once per (values, device) and
    reused (never written to): ``torch.tensor(list, device='cuda')`` is a synchronous pageable
    copy, which drains the launch queue when done every step."""
    key = (tuple(float(v) for v in values), str(device))
    t = _DEVICE_CONSTANTS.get(key)
    if t is None:
        if len(_DEVICE_CONSTANTS) > 256:
            _DEVICE_CONSTANTS.clear()
        t = torch.tensor(key[0], dtype=torch.float32, device=device)
        _DEVICE_CONSTANTS[key] = t
    return t


def pixel_loss_scales(kind, bounds, per_frame, chunk_sizes=None):
    """Per-chunk factors of the squared-error sums: 'mse' -> 1 / (frames * pixels) (reference
    losses.py:56-59: mean over ALL elements), 'll' -> -0.5 / frames (losses.py:84-96, std = 1).
    ``chunk_sizes``: the frame counts to normalise by if they are not the lengths of ``bounds``
    (frame-sharded data parallelism: the GLOBAL chunk lengths)."""
    sizes = chunk_sizes if chunk_sizes is not None else [end - beg for beg, end in bounds]
    if kind == 'mse':
        return [1.0 / (n * per_frame) for n in sizes]
    if kind == 'll':
        return [-0.5 / n for n in sizes]
    raise ValueError('unknown pixel loss kind "%s"' % kind)


# Batch-norm statistics are per 200-frame chunk (the reference runs its chunks one after another
# through the whole network); everything else in the network is per frame.  Inside ``bn_chunks``
# the batch-norm nodes take their statistics -- and fold them into the running estimates -- chunk
# by chunk over the given row ranges of ONE pass over the whole batch, so that every convolution
# sees all frames at once (AE + batch norm at 256 frames: 8.2 -> see DESIGN.md ms/step).
_bn_bounds = None


class bn_chunks(object):
    def __init__(self, bounds):
        self.bounds = [(int(b), int(e)) for b, e in bounds if e > b]

    def __enter__(self):
        global _bn_bounds
        self._prev, _bn_bounds = _bn_bounds, (self.bounds if len(self.bounds) > 1 else None)
        return self

    def __exit__(self, *exc):
        global _bn_bounds
        _bn_bounds = self._prev
        return False


def _batches_tracked(module, added=1):
    """``int(module.num_batches_tracked)`` BEFORE this node's update, without reading the device
    counter back every step (momentum=None, the reference's default: the cumulative-average factor
    is 1 / count -- nine host synchronisations per training step otherwise).  The counter itself is
    advanced by ``added`` on the device by the statistics kernel (round 4: it was two torch
    element-wise launches per layer); a host mirror follows it and is re-read from the device
    whenever something else has touched the tensor (load_state_dict, a new tensor, a reset).

    The mirror is NOT advanced here: the caller does that with ``_batches_advance`` once the library
    call that bumps the device counter has returned (a call that raised used to leave the
    mirror ahead of the device, and every later cumulative-average factor silently wrong)."""
    t = module.num_batches_tracked
    mirror = getattr(module, '_bn_count_mirror', None)
    if mirror is not None and mirror[0] is t and mirror[1] == t._version:
        return mirror[2]
    value = int(t.item())
    module._bn_count_mirror = (t, t._version, value)
    return value


def _batches_advance(module, added):
    """The device counter has been advanced by ``added`` (inside a kernel: torch's version counter of
    the tensor did not move)."""
    t = module.num_batches_tracked
    mirror = getattr(module, '_bn_count_mirror', None)
    if mirror is not None and mirror[0] is t:
        module._bn_count_mirror = (t, t._version, mirror[2] + added)


class BatchNormActFn(torch.autograd.Function):
    """y = act(BatchNorm2d(x)) with nn.BatchNorm2d's train / eval semantics (aes.py:90-97,113).

    Train mode (or no running statistics): normalise with the batch's own biased variance and
    fold the batch statistics into ``running_mean`` / ``running_var`` (unbiased variance,
    ``momentum=None`` -> cumulative average).  Eval mode: normalise with the running statistics.
    """

    @staticmethod
    def forward(ctx, x, gamma, beta, module, act):
        x = x.contiguous()
        g = gamma.detach() if gamma is not None else None
        b = beta.detach() if beta is not None else None
        rm, rv = module.running_mean, module.running_var
        batch_stats = module.training or rm is None
        if batch_stats:
            from behavenet_amd.fitting import distributed as bdist
            ctx.sync_count = None
            ctx.chunks = None
            tracking = module.training and module.track_running_stats and rm is not None
            if not tracking:
                rm = rv = None
            if bdist.frames_sharded():
                # the chunk's frames are spread over the ranks: statistics over all of them
                # (per-channel sums all-reduced), as the single device sees them
                if bdist._emulated is not None:
                    raise RuntimeError('batch-norm statistics need the other ranks\' frames: '
                                       'not available under emulate_rank')
                factor = 0.0
                if tracking:
                    if module.momentum is None:
                        factor = 1.0 / float(_batches_tracked(module, 1) + 1)
                    else:
                        factor = float(module.momentum)
                    module.num_batches_tracked.add_(1)
                    mirror = getattr(module, '_bn_count_mirror', None)
                    if mirror is not None and module.momentum is None:
                        # (torch's add_ moved the version counter: the mirror follows both)
                        t = module.num_batches_tracked
                        module._bn_count_mirror = (t, t._version, mirror[2] + 1)
                    else:
                        module._bn_count_mirror = None
                y, mean, invstd, ctx.sync_count = _hip.batchnorm_sync_train_fwd(
                    x, g, b, rm, rv, factor, float(module.eps), act, LRELU_SLOPE,
                    bdist.all_reduce_)
            else:
                bounds = _bn_bounds if _bn_bounds is not None else [(0, int(x.shape[0]))]
                if x.shape[0] != bounds[-1][1]:
                    # whole-batch statistics here would silently differ from the reference's
                    # per-chunk ones (and advance num_batches_tracked by 1 instead of n_chunks)
                    raise RuntimeError(
                        'batch norm inside bn_chunks(%s): the input has %d frames, the chunk '
                        'bounds cover %d' % (bounds, x.shape[0], bounds[-1][1]))
                # one pass over the whole batch, statistics per chunk (in chunk order: the
                # running estimates see the same sequence of updates as in the reference); the
                # device counter advances by the number of chunks inside the same library call
                k = len(bounds)
                factors = [0.0] * k
                if tracking:
                    if module.momentum is None:
                        first = _batches_tracked(module, k) + 1
                        factors = [1.0 / (first + i) for i in range(k)]
                    else:
                        factors = [float(module.momentum)] * k
                        module._bn_count_mirror = None
                try:
                    y, mean, invstd = _hip.batchnorm_train_fwd_chunks(
                        x, g, b, rm, rv, factors, float(module.eps), act, LRELU_SLOPE, bounds,
                        num_batches_tracked=module.num_batches_tracked if tracking else None)
                except Exception:
                    module._bn_count_mirror = None      # (whatever the device counter is now: re-read it)
                    raise
                if tracking and module.momentum is None:
                    _batches_advance(module, k)
                ctx.chunks = list(bounds)
        else:
            ctx.sync_count = None
            ctx.chunks = None
            mean = rm
            y, invstd = _hip.batchnorm_eval_fwd(x, g, b, rm, rv, float(module.eps), act,
                                                LRELU_SLOPE)
            mean = mean.clone()  # the running buffers may move before backward
        ctx.batch_stats = bool(batch_stats)
        ctx.act = act
        ctx.param_refs = (gamma, beta)
        # the chunked backward rebuilds the sign of the activation's input from x (identity /
        # LeakyReLU): y is not kept for it
        ctx.from_x = ctx.chunks is not None and act in (_hip.ACT_NONE, _hip.ACT_LRELU)
        ctx.beta = b
        if ctx.from_x:
            ctx.save_for_backward(x, mean, invstd, g)
        else:
            ctx.save_for_backward(x, y, mean, invstd, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.from_x:
            x, mean, invstd, g = ctx.saved_tensors
            y = None
        else:
            x, y, mean, invstd, g = ctx.saved_tensors
        gamma, beta = ctx.param_refs
        need_g = gamma is not None and ctx.needs_input_grad[1]
        need_b = beta is not None and ctx.needs_input_grad[2]
        gg = _grad_buffer(gamma) if need_g else None
        gb = _grad_buffer(beta) if need_b else None
        direct = (need_g or need_b) and (gg is not None or not need_g) and \
            (gb is not None or not need_b)
        if getattr(ctx, 'sync_count', None) is not None:
            from behavenet_amd.fitting import distributed as bdist
            dx, sum_dz, sum_dzx = _hip.batchnorm_sync_bwd(
                x, y, dy.contiguous(), mean, invstd, g, ctx.sync_count, ctx.act, LRELU_SLOPE,
                bdist.all_reduce_)
            if direct:
                if need_g:
                    gg.add_(sum_dzx)
                if need_b:
                    gb.add_(sum_dz)
                return (dx if ctx.needs_input_grad[0] else None), None, None, None, None
            return (dx if ctx.needs_input_grad[0] else None), \
                (sum_dzx if need_g else None), (sum_dz if need_b else None), None, None
        chunks = getattr(ctx, 'chunks', None)
        if chunks is not None:
            dy = dy.contiguous()
            if direct:
                dgamma, dbeta = gg, gb
            else:
                dgamma = torch.zeros_like(mean[0]) if need_g else None
                dbeta = torch.zeros_like(mean[0]) if need_b else None
            dx = _hip.batchnorm_bwd_chunks(x, y, dy, mean, invstd, g, dgamma, dbeta, True, ctx.act,
                                           LRELU_SLOPE, chunks, beta=ctx.beta)
            if direct:
                dgamma = dbeta = None
            return (dx if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None
        if direct:
            dgamma, dbeta = gg, gb
        else:
            dgamma = torch.empty_like(mean) if need_g else None
            dbeta = torch.empty_like(mean) if need_b else None
        dx = _hip.batchnorm_bwd(x, y, dy.contiguous(), mean, invstd, g, dgamma, dbeta, direct,
                                ctx.batch_stats, ctx.act, LRELU_SLOPE)
        if direct:
            dgamma = dbeta = None
        return (dx if ctx.needs_input_grad[0] else None), dgamma, dbeta, None, None


def conv_stack_bn(plan, x, params, bn_modules):
    """Conv stack whose layer i is followed by ``bn_modules[i]`` (or None) before its activation.

    Runs of layers without batch norm stay one fused :class:`ConvStackFn`; a batch-normed layer
    is convolution (no activation) -> :class:`BatchNormActFn` (normalisation + activation fused).
    """
    h = x
    run_plan, run_params = [], []
    for i, layer in enumerate(plan):
        run_params += [params[2 * i], params[2 * i + 1]]
        if bn_modules[i] is None:
            run_plan.append(layer)
            continue
        run_plan.append(layer.with_act(_hip.ACT_NONE))
        h = ConvStackFn.apply(run_plan, h, None, *run_params)
        bn = bn_modules[i]
        h = BatchNormActFn.apply(h, bn.weight, bn.bias, bn, layer.act)
        if _sign_tap is not None and layer.act == _hip.ACT_LRELU:
            _sign_tap.setdefault(id(plan), [[] for _ in plan])[i].append((h.detach() > 0).cpu())
        run_plan, run_params = [], []
    if run_plan:
        h = ConvStackFn.apply(run_plan, h, None, *run_params)
    return h


class LinearFn(torch.autograd.Function):
    """y = x w^T + b on v_mfma_f32_32x32x2_f32."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        ctx.param_refs = (w, b)
        _note_use((w, b), ctx.needs_input_grad[1])
        return _hip.linear_fwd(x, w.detach().contiguous(),
                               b.detach() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        gw = _grad_buffer(ctx.param_refs[0]) if need_dw else None
        gb = _grad_buffer(ctx.param_refs[1]) if need_db else None
        direct = need_dw and gw is not None and (gb is not None or not need_db)
        if direct:
            dw, db = gw, gb
        else:
            dw = torch.empty_like(w) if need_dw else None
            db = torch.empty((w.shape[0],), dtype=w.dtype, device=w.device) if need_db else None
        wc = w.contiguous()
        if direct and _use_side_stream:
            # in-place accumulation into param.grad: on the weight-gradient side stream, like the
            # convolutions', so that chunks running on different streams never race on it
            dx = _hip.linear_bwd(x, wc, dy, need_dx, None, _hip.ACT_NONE, 0.0, None, None, False) \
                if need_dx else None
            side = _side_stream(w.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(w.device))
            with torch.cuda.stream(side):
                side.wait_event(ev)
                _hip.linear_bwd(x, wc, dy, False, None, _hip.ACT_NONE, 0.0, dw, db, True)
            dy.record_stream(side)
            x.record_stream(side)
            _report_ready(ctx.param_refs)
            return dx, None, None
        dx = _hip.linear_bwd(x, wc, dy, need_dx, None, _hip.ACT_NONE, 0.0, dw, db, direct)
        if direct:
            _report_ready(ctx.param_refs)
            return dx, None, None
        return dx, dw, db


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


class SqErrFn(torch.autograd.Function):
    """scale * sum_i (a_i - b_i)^2 * mask_i, returned as a 0-dim device tensor."""

    @staticmethod
    def forward(ctx, a, b, mask, scale):
        a, b = a.contiguous(), b.contiguous()
        mask = mask.contiguous() if mask is not None else None
        ctx.save_for_backward(a, b, mask)
        ctx.scale = float(scale)
        sums = _hip.sqerr_frame_sums(a, b, mask)
        return _hip.reduce_sum(sums, scale)

    @staticmethod
    def backward(ctx, g):
        a, b, mask = ctx.saved_tensors
        g = g.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _hip.sqerr_bwd(a, b, mask, ctx.scale, g)
        if ctx.needs_input_grad[1]:
            db = _hip.sqerr_bwd(b, a, mask, ctx.scale, g)
        return da, db, None, None


def sq_err(a, b, mask, scale):
    return SqErrFn.apply(a, b, mask, scale)


class ChunkedSqErrFn(torch.autograd.Function):
    """out[c] = scales[c] * sum_{frames in chunk c} (a - b)^2 * mask for contiguous frame ranges
    ``bounds`` -- the per-chunk pixel losses of a batch whose forward ran in ONE pass."""

    @staticmethod
    def forward(ctx, a, b, mask, bounds, scales):
        a, b = a.contiguous(), b.contiguous()
        mask = mask.contiguous() if mask is not None else None
        ctx.save_for_backward(a, b, mask)
        ctx.bounds, ctx.scales = list(bounds), [float(s) for s in scales]
        sums = _hip.sqerr_frame_sums(a, b, mask)
        out = torch.empty((len(ctx.bounds),), dtype=torch.float32, device=a.device)
        for i, ((beg, end), s) in enumerate(zip(ctx.bounds, ctx.scales)):
            _hip.reduce_sum(sums[beg:end], s, out=out[i:i + 1])
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, mask = ctx.saved_tensors
        g = g.contiguous()
        grads = []
        for need, p, t in ((ctx.needs_input_grad[0], a, b), (ctx.needs_input_grad[1], b, a)):
            if not need:
                grads.append(None)
                continue
            d = torch.empty_like(p)
            for i, ((beg, end), s) in enumerate(zip(ctx.bounds, ctx.scales)):
                _hip.sqerr_bwd(p[beg:end], t[beg:end], mask[beg:end] if mask is not None else None,
                               s, g[i:i + 1], out=d[beg:end])
            grads.append(d)
        return grads[0], grads[1], None, None, None


def chunked_sq_err(a, b, mask, bounds, scales):
    return ChunkedSqErrFn.apply(a, b, mask, bounds, scales)


class ReparamFn(torch.autograd.Function):
    """z = mu + eps * exp(logvar)  (std = exp(logvar) as in the reference, vaes.py:33)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        z = _hip.reparam_fwd(mu, logvar, eps)
        ctx.save_for_backward(mu, z)
        return z

    @staticmethod
    def backward(ctx, dz):
        mu, z = ctx.saved_tensors
        # dz/dmu = 1, dz/dlogvar = eps*exp(logvar) = z - mu
        dz = dz.contiguous()
        dlogvar = _hip.reparam_bwd(dz, z, mu) if ctx.needs_input_grad[1] else None
        return (dz if ctx.needs_input_grad[0] else None), dlogvar, None


def reparameterize_with_eps(mu, logvar, eps):
    return ReparamFn.apply(mu, logvar, eps)


class KLFn(torch.autograd.Function):
    """mean_n 0.5 * sum_d (exp(logvar) - logvar + mu^2 - 1)  (losses.py:146-147)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        ctx.save_for_backward(mu, logvar)
        rows = _hip.kl_rows(mu, logvar)
        return _hip.reduce_sum(rows, 1.0 / mu.shape[0])

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu, dlogvar = _hip.kl_bwd(mu, logvar, 1.0 / mu.shape[0], g.contiguous())
        return dmu, dlogvar


class ActFn(torch.autograd.Function):
    """y = act(x) as its own node (the Sigmoid behind the optional dense last decoder layer)."""

    @staticmethod
    def forward(ctx, x, act):
        y = _hip.act_fwd(x.contiguous(), act, LRELU_SLOPE)
        ctx.act = act
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _hip.act_bwd(dy.contiguous(), y, ctx.act, LRELU_SLOPE), None


def activation(x, act):
    return ActFn.apply(x, act)


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(return_indices=True) (ref aes.py:99-110,204-207) -> (y, idx int32)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, out_hw):
        x = x.contiguous()
        y, idx = _hip.maxpool2d_fwd(x, k, stride, pad, out_hw)
        ctx.save_for_backward(idx)
        ctx.args = (tuple(x.shape[2:]), k, stride, pad)
        ctx.mark_non_differentiable(idx)
        # (autograd would otherwise materialise a zero "gradient" of the int32 index output for every backward pass:
        # a 33-67 MB fill per pooling layer of the max-pooling test architecture)
        ctx.set_materialize_grads(False)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        if dy is None:
            return None, None, None, None, None
        idx, = ctx.saved_tensors
        in_hw, k, stride, pad = ctx.args
        return _hip.maxpool2d_bwd(dy.contiguous(), idx, in_hw, k, stride, pad), None, None, None, \
            None


class MaxPoolActFn(torch.autograd.Function):
    """2x2 / stride-2 / unpadded max pooling and the activation behind it (ref aes.py:204-211) in one pass each
    way -> (y, idx int32); only built by max_pool_act where bn_maxpool2d_act_fwd applies."""

    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        out = _hip.maxpool2d_act_fwd(x, act, LRELU_SLOPE)
        if out is None:
            raise RuntimeError('max_pool_act: geometry not served')
        y, idx = out
        ctx.save_for_backward(idx, y)
        ctx.args = (tuple(x.shape[2:]), act)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        if dy is None:
            return None, None
        idx, y = ctx.saved_tensors
        in_hw, act = ctx.args
        return _hip.maxpool2d_act_bwd(dy.contiguous(), y, idx, in_hw, act, LRELU_SLOPE), None


def max_pool_act(x, k, stride, pad, out_hw, act):
    """max_pool followed by ``activation(., act)``; fused where the pooling is 2x2 / stride 2 / unpadded on an even
    map whose pooled width is even (the only pooling the reference's generator draws)."""
    fused = (int(k) == 2 and int(stride) == 2 and int(pad[0]) == 0 and int(pad[1]) == 0 and x.is_cuda and
             x.shape[2] == 2 * int(out_hw[0]) and x.shape[3] == 2 * int(out_hw[1]) and x.shape[3] % 4 == 0 and
             x.data_ptr() % 16 == 0)
    if not fused:
        y, idx = max_pool(x, k, stride, pad, out_hw)
        return activation(y, act), idx
    y, idx = MaxPoolActFn.apply(x, int(act))
    idx.bn_own_window = True
    return y, idx


class ConvPoolActFn(torch.autograd.Function):
    """One conv layer, the 2x2 / stride-2 max pooling behind it and the activation behind that in ONE kernel
    (bn_conv2d_pool2_act_fwd: the first layer of a max-pooling architecture, whose 268 MB of output per 256 frames of
    128x128 were written, read back by the pooling and never looked at again) -> (y, idx int32).  Backward: the pooling's
    own (dy * act'(y) spread to the winners), then the layer's weight gradient through _stack_backward."""

    @staticmethod
    def forward(ctx, plan, x, act, *params):
        layer = plan[0]
        x = x.contiguous()
        out = _hip.conv2d_pool_act_fwd(x, params[0].detach(), params[1].detach(), layer.geom(x.shape[0]), act, LRELU_SLOPE)
        if out is None:
            raise RuntimeError('conv_pool_act: geometry not served')
        y, idx = out
        ctx.plan = plan
        ctx.taps = None
        ctx.need_dx = x.requires_grad
        ctx.param_refs = params
        ctx.pool_args = ((layer.hout, layer.wout), act)
        _note_use(params, any(ctx.needs_input_grad[3:]))
        # (slot 1 = the layer's output in ConvStackFn's layout: the pooled output stands in, _stack_backward does
        # not read it for a one-layer plan whose activation is applied here)
        ctx.save_for_backward(x, y, params[0], idx)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        if dy is None:
            return (None,) * (3 + len(ctx.param_refs))
        x, y, w, idx = ctx.saved_tensors
        in_hw, act = ctx.pool_args
        dy = dy.contiguous()
        layer = ctx.plan[0]
        g = layer.geom(dy.shape[0])
        need_w, need_b = ctx.needs_input_grad[3], ctx.needs_input_grad[4]
        if (_CONV_POOL_WGRAD and not ctx.need_dx and need_w and x.dtype == torch.float32 and
                _hip.conv2d_pool_bwd_weight_ws_bytes(g)):
            # the first layer: no data gradient, and its weight gradient needs the winners only -- straight from the
            # pooled gradient (the dense one, 3/4 zeros, is never built: bn_conv2d_pool2_bwd_weight)
            gw = _grad_buffer(ctx.param_refs[0])
            gb = _grad_buffer(ctx.param_refs[1]) if need_b else None
            direct = gw is not None and (gb is not None or not need_b)
            if direct:
                dw, db = gw, gb
            else:
                dw = torch.empty_like(w)
                db = torch.empty((layer.cout,), dtype=w.dtype, device=w.device) if need_b else None
            _hip.conv2d_pool_bwd_weight(x, dy, y, idx, dw, db, g, act, LRELU_SLOPE, direct)
            if direct:
                _report_ready(ctx.param_refs[0:2])
                return (None, None, None, None, None)
            return (None, None, None, dw, db)
        dpre = _hip.maxpool2d_act_bwd(dy, y, idx, in_hw, act, LRELU_SLOPE)
        dx, grads = _stack_backward(ctx, dpre, first_param=3)
        return (None, dx, None) + tuple(grads)


_CONV_POOL = os.environ.get('BN_CONV_POOL', '1') != '0'      # 0: convolve, then pool (A/B switch)
_CONV_POOL_WGRAD = os.environ.get('BN_CONV_POOL_WGRAD', '1') != '0'      # 0: dense gradient, then the layer's weight gradient


def conv_pool_act(layer, x, params, k, stride, pad, out_hw, act):
    """``max_pool_act(conv_stack([layer], x, params), ...)`` in one kernel where bn_conv2d_pool2_act_fwd serves the layer
    (bn_conv2d_pool2_act_ok; pooling 2x2 / stride 2 / unpadded on an even map) -> (y, idx), else None."""
    if not (_CONV_POOL and x.is_cuda and layer.kind == 'conv' and layer.act == _hip.ACT_NONE and int(k) == 2 and
            int(stride) == 2 and int(pad[0]) == 0 and int(pad[1]) == 0 and layer.hout == 2 * int(out_hw[0]) and
            layer.wout == 2 * int(out_hw[1]) and layer.wout % 4 == 0 and x.dtype == torch.float32 and
            x.data_ptr() % 16 == 0 and params[0].data_ptr() % 16 == 0 and
            _hip.conv2d_pool_act_ok(layer.geom(x.shape[0]))):
        return None
    y, idx = ConvPoolActFn.apply([layer], x, int(act), *params)
    idx.bn_own_window = True
    return y, idx


def max_pool(x, k, stride, pad, out_hw):
    y, idx = MaxPoolFn.apply(x, int(k), int(stride), (int(pad[0]), int(pad[1])),
                             (int(out_hw[0]), int(out_hw[1])))
    # (a 2x2 / stride-2 / unpadded pooling of an even map: every index lies in its own window, which lets the
    # unpooling that receives THIS tensor run in one pass -- max_unpool)
    idx.bn_own_window = (int(k) == 2 and int(stride) == 2 and int(pad[0]) == 0 and int(pad[1]) == 0 and
                         x.shape[2] == 2 * int(out_hw[0]) and x.shape[3] == 2 * int(out_hw[1]))
    return y, idx


class MaxUnpoolFn(torch.autograd.Function):
    """nn.MaxUnpool2d with the encoder's indices and pre-pool size (ref aes.py:281-294,460-464)."""

    @staticmethod
    def forward(ctx, x, idx, out_hw, own_window=False):
        ctx.save_for_backward(idx)
        return _hip.maxunpool2d_fwd(x.contiguous(), idx, out_hw, own_window)

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        return _hip.maxunpool2d_bwd(dy.contiguous(), idx), None, None, None


def max_unpool(x, idx, out_hw):
    if tuple(idx.shape) != tuple(x.shape):
        raise ValueError('unpool indices %s do not match the input %s' % (
            tuple(idx.shape), tuple(x.shape)))
    return MaxUnpoolFn.apply(x, idx, (int(out_hw[0]), int(out_hw[1])), bool(getattr(idx, 'bn_own_window', False)))


class DecomposedKLFn(torch.autograd.Function):
    """(MI, TC, DWKL) of losses.py:284-351 as one (3,) tensor; N x N x D never materialised."""

    @staticmethod
    def forward(ctx, z, mu, logvar):
        z, mu, logvar = z.contiguous(), mu.contiguous(), logvar.contiguous()
        out3, log_qz, lse = _hip.decomposed_kl_fwd(z, mu, logvar)
        ctx.save_for_backward(z, mu, logvar, log_qz, lse)
        return out3

    @staticmethod
    def backward(ctx, g3):
        z, mu, logvar, log_qz, lse = ctx.saved_tensors
        dz, dmu, dlogvar = _hip.decomposed_kl_bwd(z, mu, logvar, log_qz, lse, g3.contiguous())
        return dz, dmu, dlogvar


def decomposed_kl_terms(z, mu, logvar):
    return DecomposedKLFn.apply(z, mu, logvar)


class KLChunksFn(torch.autograd.Function):
    """(n_chunks,) tensor of weights[c] * kl_div_to_std_normal(rows of chunk c) for contiguous row
    ranges `bounds` that tile the batch.  One node instead of a slice / contiguous / KLFn / scale
    chain per chunk: row slices of a contiguous tensor are pointer offsets for the kernels, and the
    backward pass writes every chunk's gradient into its rows of ONE buffer (autograd's slice
    backward allocates a zero tensor of the full size per slice and adds them up)."""

    @staticmethod
    def forward(ctx, mu, logvar, bounds, weights):
        mu, logvar = mu.contiguous(), logvar.contiguous()
        out = torch.empty((len(bounds),), dtype=torch.float32, device=mu.device)
        for c, (b, e) in enumerate(bounds):
            if e > b:
                _hip.reduce_sum(_hip.kl_rows(mu[b:e], logvar[b:e]), float(weights[c]) / (e - b), out=out[c])
            else:
                out[c].zero_()
        ctx.save_for_backward(mu, logvar)
        ctx.bounds, ctx.weights = list(bounds), [float(w) for w in weights]
        return out

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlogvar = torch.empty_like(mu), torch.empty_like(mu)
        covered = 0
        for c, (b, e) in enumerate(ctx.bounds):
            if e > b:
                _hip.kl_bwd(mu[b:e], logvar[b:e], ctx.weights[c] / (e - b), g[c],
                            out=(dmu[b:e], dlogvar[b:e]))
                covered += e - b
        assert covered == mu.shape[0], 'chunk bounds must tile the rows'
        return dmu, dlogvar, None, None


def kl_chunks(mu, logvar, bounds, weights):
    return KLChunksFn.apply(mu, logvar, bounds, weights)


class DecomposedKLChunksFn(torch.autograd.Function):
    """(n_chunks, 3) tensor of the decomposed-KL terms of every contiguous row range in `bounds`
    (they tile the batch); see KLChunksFn for why this is one node."""

    @staticmethod
    def forward(ctx, z, mu, logvar, bounds):
        z, mu, logvar = z.contiguous(), mu.contiguous(), logvar.contiguous()
        out = torch.empty((len(bounds), 3), dtype=torch.float32, device=z.device)
        saved = []
        for c, (b, e) in enumerate(bounds):
            _, log_qz, lse = _hip.decomposed_kl_fwd(z[b:e], mu[b:e], logvar[b:e], out3=out[c])
            saved += [log_qz, lse]
        ctx.save_for_backward(z, mu, logvar, *saved)
        ctx.bounds = list(bounds)
        return out

    @staticmethod
    def backward(ctx, g):
        z, mu, logvar = ctx.saved_tensors[:3]
        saved = ctx.saved_tensors[3:]
        g = g.contiguous()
        dz, dmu, dlogvar = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        covered = 0
        for c, (b, e) in enumerate(ctx.bounds):
            _hip.decomposed_kl_bwd(z[b:e], mu[b:e], logvar[b:e], saved[2 * c], saved[2 * c + 1], g[c],
                                   out=(dz[b:e], dmu[b:e], dlogvar[b:e]))
            covered += e - b
        assert covered == z.shape[0], 'chunk bounds must tile the rows'
        return dz, dmu, dlogvar, None


def decomposed_kl_chunks(z, mu, logvar, bounds):
    return DecomposedKLChunksFn.apply(z, mu, logvar, bounds)


class SplitColsFn(torch.autograd.Function):
    """(t[:, :k], t[:, k:]) as two contiguous tensors; the backward pass is ONE concatenation
    (autograd's slice backward would fill a zero tensor of the full size per slice, copy into it
    and add the two up)."""

    @staticmethod
    def forward(ctx, t, k):
        ctx.k, ctx.shape = int(k), t.shape
        return t[:, :k].contiguous(), t[:, k:].contiguous()

    @staticmethod
    def backward(ctx, g1, g2):
        n, d = ctx.shape
        if g1 is None:
            g1 = torch.zeros((n, ctx.k), dtype=g2.dtype, device=g2.device)
        if g2 is None:
            g2 = torch.zeros((n, d - ctx.k), dtype=g1.dtype, device=g1.device)
        return torch.cat([g1, g2], dim=1), None


def split_cols(t, k):
    return SplitColsFn.apply(t, k)


class CombineChunkTermsFn(torch.autograd.Function):
    """total[c] = sum_i sum_j coefs[i][j] * terms[i][c, j] for per-chunk term tensors of shape
    (n_chunks,) or (n_chunks, k_i) -> (total (n_chunks,), matrix (n_chunks, sum k_i) of the raw
    terms, for the metric read-back).  One node and three small kernels instead of a chain of
    scalar multiplies, adds and negations with a backward node each."""

    @staticmethod
    def forward(ctx, coefs, *terms):
        cols = [t[:, None] if t.dim() == 1 else t for t in terms]
        mat = torch.cat(cols, dim=1)
        cvec = device_constant([c for cs in coefs for c in cs], mat.device)
        ctx.widths = [c.shape[1] for c in cols]
        ctx.flat = [t.dim() == 1 for t in terms]
        ctx.save_for_backward(cvec)
        ctx.mark_non_differentiable(mat)
        ctx.set_materialize_grads(False)
        return torch.mv(mat, cvec), mat

    @staticmethod
    def backward(ctx, g, _gmat):
        if g is None:
            return (None,) * (1 + len(ctx.widths))
        (cvec,) = ctx.saved_tensors
        G = g[:, None] * cvec[None, :]                     # (n_chunks, sum k_i)
        out, pos = [], 0
        for width, flat in zip(ctx.widths, ctx.flat):
            blk = G[:, pos:pos + width]
            out.append(blk[:, 0] if flat else blk)
            pos += width
        return (None,) + tuple(out)


def combine_chunk_terms(terms, coefs):
    """-> (total (n_chunks,), matrix of the terms (n_chunks, K), detached)."""
    return CombineChunkTermsFn.apply([[float(c) for c in cs] for cs in coefs], *terms)


def kl_to_std_normal(mu, logvar):
    return KLFn.apply(mu, logvar)


_DEVICE_INT_CONSTANTS = {}


def device_int_constant(values, device):
    """int32 counterpart of :func:`device_constant` (chunk bounds handed to kernels)."""
    key = (tuple(int(v) for v in values), str(device))
    t = _DEVICE_INT_CONSTANTS.get(key)
    if t is None:
        if len(_DEVICE_INT_CONSTANTS) > 256:
            _DEVICE_INT_CONSTANTS.clear()
        t = torch.tensor(key[0], dtype=torch.int32, device=device)
        _DEVICE_INT_CONSTANTS[key] = t
    return t


class PSVAEHeadFn(torch.autograd.Function):
    """Everything between the PS encoder's heads and the decoder as ONE autograd node (reference
    vaes.py:571-601 forward, :669-704 loss terms; csrc/psvae_head.hip):

        (y, w, logvar, D.weight, D.bias) -> z (N, L + U), T (n_chunks,), y_hat (N, L), cols5

    with T[c] = -alpha ll_labels + KL(z_s) + kl MI + beta TC + kl DWKL of chunk c (rows
    ``bounds[c]``; means over the chunk) and cols5 = the five raw terms for the metric table.
    Forward: 1 + 2 n_chunks + 1 launches (rows, decomposed KL per chunk, combination); backward:
    1 + 2 n_chunks + 1.  It replaces the reparameterisation, DiagLinear, label log-likelihood, KL,
    column-split, concatenation and combination nodes (~75 element-wise launches per step)."""

    @staticmethod
    def forward(ctx, y, w, logvar, Dw, Db, labels, lmask, eps, bounds, alpha, kl, beta):
        y, w, logvar = y.contiguous(), w.contiguous(), logvar.contiguous()
        labels, eps = labels.contiguous(), eps.contiguous()
        lmask = lmask.contiguous() if lmask is not None else None
        dwv, dbv = Dw.detach(), (Db.detach() if Db is not None else None)
        L = y.shape[1]
        z, z_u, lv_u, yhat, row_sq, row_kl = _hip.psvae_head_fwd(y, w, logvar, eps, dwv, dbv, labels,
                                                                 lmask)
        n_chunks = len(bounds)
        dkl3 = torch.empty((n_chunks, 3), dtype=torch.float32, device=y.device)
        saved = []
        for c, (b, e) in enumerate(bounds):
            _, log_qz, lse = _hip.decomposed_kl_fwd(z_u[b:e], w[b:e], lv_u[b:e], out3=dkl3[c])
            saved += [log_qz, lse]
        bdev = device_int_constant([v for be in bounds for v in be], y.device)
        T, cols5 = _hip.psvae_head_combine(row_sq, row_kl, dkl3, bdev, n_chunks, alpha, kl, beta, L)
        ctx.save_for_backward(y, w, logvar, eps, yhat, labels, lmask, dwv, z_u, lv_u, bdev, *saved)
        ctx.bounds = list(bounds)
        ctx.coefs = (float(alpha), float(kl), float(beta))
        ctx.param_refs = (Dw, Db)
        ctx.has_bias = Db is not None
        _note_use((Dw, Db), ctx.needs_input_grad[3])
        ctx.mark_non_differentiable(yhat, cols5)
        ctx.set_materialize_grads(False)      # (no zero "gradients" of y_hat / cols5: two fill launches per backward pass)
        return z, T, yhat, cols5

    @staticmethod
    def backward(ctx, dz, gT, _gy, _gc):
        y, w, logvar, eps, yhat, labels, lmask, dwv, z_u, lv_u, bdev = ctx.saved_tensors[:11]
        saved = ctx.saved_tensors[11:]
        alpha, kl, beta = ctx.coefs
        n_chunks = len(ctx.bounds)
        dz = dz.contiguous() if dz is not None else torch.zeros_like(logvar)
        gT = gT.contiguous() if gT is not None else torch.zeros((n_chunks,), device=y.device)
        g3 = gT[:, None] * device_constant([kl, beta, kl], y.device)[None, :]
        gz_u, gmu_u, glv_u = torch.empty_like(z_u), torch.empty_like(z_u), torch.empty_like(z_u)
        for c, (b, e) in enumerate(ctx.bounds):
            _hip.decomposed_kl_bwd(z_u[b:e], w[b:e], lv_u[b:e], saved[2 * c], saved[2 * c + 1], g3[c],
                                   out=(gz_u[b:e], gmu_u[b:e], glv_u[b:e]))
        Dw, Db = ctx.param_refs
        need_d = ctx.needs_input_grad[3]
        gw = _grad_buffer(Dw) if need_d else None
        gb = _grad_buffer(Db) if (need_d and ctx.has_bias) else None
        direct = need_d and gw is not None and (gb is not None or not ctx.has_bias)
        if direct:
            dDw, dDb = gw, gb
        else:
            dDw = torch.empty_like(dwv) if need_d else None
            dDb = torch.empty_like(dwv) if (need_d and ctx.has_bias) else None
        dy, dw, dlogvar = _hip.psvae_head_bwd(dz, gT, bdev, n_chunks, y, logvar, eps, yhat, labels,
                                              lmask, dwv, gz_u, gmu_u, glv_u, alpha, dDw, dDb, direct)
        if direct:
            _report_ready(ctx.param_refs)
            dDw = dDb = None
        return dy, dw, dlogvar, dDw, dDb, None, None, None, None, None, None, None


def psvae_head(y, w, logvar, D, labels, lmask, eps, bounds, alpha, kl, beta):
    """-> (z, T (n_chunks,), y_hat, cols5 (n_chunks, 5)); D = the DiagLinear label map."""
    return PSVAEHeadFn.apply(y, w, logvar, D.weight, D.bias, labels, lmask, eps, bounds, alpha, kl, beta)
