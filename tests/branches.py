"""LeakyReLU branch bookkeeping for gradient parity tests (see oracle/ref_cpu.py LRELU_BRANCH).

``record_branches(model)`` captures, during the HIP model's own ``loss()`` call, which side of
zero every LeakyReLU output of its conv stacks fell on; ``BranchReplay`` hands that pattern to
the oracle (chunk by chunk, in call order) and counts the elements where the oracle's own
arithmetic would have chosen the other branch, with their magnitudes: a legitimate difference is
a TIE -- |pre-activation| within fp32 rounding of zero relative to its layer.
"""

import contextlib

import torch

from behavenet_amd import hip_functions as hf
from oracle import ref_cpu


@contextlib.contextmanager
def record_branches(model):
    """-> dict: {'encoding'|'decoding': [bool tensor (N,C,H,W) or None per layer]} (filled on
    exit; frames in the order the model processed them, i.e. batch order)."""
    out = {}
    hf._sign_tap = {}
    try:
        yield out
    finally:
        tap, hf._sign_tap = hf._sign_tap, None
        for stack in ('encoding', 'decoding'):
            mod = getattr(model, stack, None)
            plan = getattr(mod, '_plan', None)
            if plan is None or id(plan) not in tap:
                continue
            out[stack] = [torch.cat(parts, 0) if parts else None for parts in tap[id(plan)]]


class BranchReplay(object):
    """Serves recorded branch patterns to the oracle; ``with BranchReplay(rec) as br: ...``."""

    def __init__(self, recorded):
        self.rec = recorded
        self.cursor = {}
        self.flips = []          # (stack, layer, |value| / layer max) of every differing element
        self.n_elements = 0

    def take(self, stack, layer, x):
        layers = self.rec.get(stack)
        if layers is None or layer >= len(layers) or layers[layer] is None:
            return None
        beg = self.cursor.get((stack, layer), 0)
        pos = layers[layer][beg:beg + x.shape[0]]
        assert pos.shape == x.shape, (stack, layer, tuple(pos.shape), tuple(x.shape))
        self.cursor[(stack, layer)] = beg + x.shape[0]
        own = x.detach() > 0
        diff = own != pos
        self.n_elements += x.numel()
        if bool(diff.any()):
            scale = float(x.detach().abs().max())
            for v in x.detach()[diff].abs().tolist():
                self.flips.append((stack, layer, v / scale))
        return pos

    def __enter__(self):
        ref_cpu.LRELU_BRANCH = self
        return self

    def __exit__(self, *exc):
        ref_cpu.LRELU_BRANCH = None
        return False

    def assert_only_ties(self, max_rel=2e-6, max_fraction=2e-5):
        """Every element on a different branch is a tie, and there are few of them."""
        for stack, layer, rel in self.flips:
            assert rel <= max_rel, 'branch differs at a non-tie: %s layer %d, |x|/max = %.2e' % (
                stack, layer, rel)
        assert len(self.flips) <= max(2, max_fraction * self.n_elements), len(self.flips)
