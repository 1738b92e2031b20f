"""Adam(amsgrad) over one flat fp32 arena, stepped by a single HIP kernel.

Replaces ``torch.optim.Adam(model.get_parameters(), lr, weight_decay=l2_reg, amsgrad=True)`` of
the reference (training.py:284-286).  All trainable parameters are re-homed into ONE contiguous
device buffer (each tensor 16-byte aligned) with a parallel gradient arena, so that

* ``zero_grad`` is one memset,
* the data-parallel gradient exchange is one RCCL all-reduce of one buffer
  (fitting/distributed.py),
* ``step`` is one launch of ``bn_adam_amsgrad_step`` streaming p, g, m, v, vmax once.

``state_dict()`` of the model is unaffected (parameters stay ``nn.Parameter`` objects, now
views into the arena).
"""

import torch

from behavenet_amd import _hip
from behavenet_amd.hip_functions import join_side_streams

_ALIGN = 4  # floats (16 bytes)


class FlatAdamAMSGrad(object):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 shard_over=1):
        """``shard_over`` = R > 1: the arenas are padded to R equal 16-byte aligned shards, so that
        ``fitting.distributed.sharded_step`` can reduce-scatter the gradients, step shard r on rank r
        (``step_range``) and all-gather the parameters."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError('all parameters must be fp32 on one device')
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0

        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.shard_over = max(1, int(shard_over))
        if self.shard_over > 1:
            per = (total + self.shard_over - 1) // self.shard_over
            per = (per + _ALIGN - 1) // _ALIGN * _ALIGN
            total = per * self.shard_over
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.max_exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[off:off + p.numel()].view_as(p)

    def _grads_in_arena(self):
        """Re-attach gradient views if something replaced ``p.grad`` (e.g. set_to_none)."""
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            view = self.flat_g[off:off + p.numel()].view_as(p)
            if g is None:
                p.grad = view
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view

    def zero_grad(self):
        join_side_streams()   # pending side-stream accumulations must not race the memset
        if getattr(self, 'reducer', None) is not None:
            self.reducer.begin()   # collectives of a step that was not applied are waited for
        self._grads_in_arena()
        self.flat_g.zero_()

    def step(self):
        self.step_range(0, self.flat_p.numel())

    def shard_range(self, r):
        """[lo, hi) of shard ``r`` of ``shard_over`` in the arenas."""
        per = self.flat_p.numel() // self.shard_over
        return r * per, (r + 1) * per

    def step_range(self, lo, hi):
        """One Adam(amsgrad) step of the arena elements [lo, hi) only (the update is element-wise: a
        range stepped alone gets bit for bit what a step of the whole arena gives it).  ``lo`` and
        ``hi`` must be multiples of 4 (16-byte groups)."""
        if lo % _ALIGN or hi % _ALIGN:
            raise ValueError('step_range: [%d, %d) is not 16-byte aligned' % (lo, hi))
        join_side_streams()   # weight gradients queued on the side stream are complete
        self._grads_in_arena()
        self.step_count += 1
        if self.flat_p.is_cuda:
            _hip.adam_amsgrad_step(
                self.flat_p[lo:hi], self.flat_g[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                self.max_exp_avg_sq[lo:hi],
                self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                self.step_count)
        else:
            raise _hip.HipLibraryError(
                'FlatAdamAMSGrad.step: parameters are on %s; the optimizer kernel only runs on '
                'the GPU (no CPU fallback)' % self.flat_p.device)

    def state_tensors(self, index):
        """(exp_avg, exp_avg_sq, max_exp_avg_sq) views for parameter ``index`` (for tests)."""
        p, off = self.params[index], self.offsets[index]
        sl = slice(off, off + p.numel())
        return (self.exp_avg[sl].view_as(p), self.exp_avg_sq[sl].view_as(p),
                self.max_exp_avg_sq[sl].view_as(p))
