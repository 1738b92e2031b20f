"""Module/model templates; mirror of the reference ``behavenet/models/base.py``."""


import torch
from torch import nn

__all__ = ['BaseModule', 'BaseModel', 'DiagLinear', 'CustomDataParallel']


class BaseModule(nn.Module):
    """Template for encoder/decoder modules (ref base.py:10-36)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def __str__(self):
        raise NotImplementedError

    def build_model(self):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def freeze(self):
        """Exclude every parameter of this module from gradient updates."""
        for p in self.parameters():
            p.requires_grad = False

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True


class BaseModel(nn.Module):
    """Template for models (ref base.py:39-67)."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def __str__(self):
        raise NotImplementedError

    def build_model(self):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def loss(self, *args, **kwargs):
        raise NotImplementedError

    def save(self, filepath):
        """``torch.save`` of the state dict, key-compatible with reference checkpoints.

        Parameters may live in one flat arena (fitting/optim.py); every entry is cloned so the
        file holds independent tensors like the reference's (base.py:61-63).
        """
        torch.save({k: v.detach().clone() for k, v in self.state_dict().items()}, filepath)

    def get_parameters(self):
        """Parameters with gradient updates switched on (frozen PS-VAE A/B are skipped)."""
        return filter(lambda p: p.requires_grad, self.parameters())


class DiagLinear(nn.Module):
    """Element-wise affine map ``y_i = d_i * x_i + b_i``: the PS-VAE's label head ``D``, which
    rescales every supervised latent on its own (reference base.py:70-103).

    ``state_dict`` keys ``weight`` / ``bias`` (both of length ``features``); initial values are
    uniform in +-1/sqrt(features), the weight drawn before the bias -- the order in which the
    reference consumes the seeded generator, so that a seed reproduces its parameters."""

    def __init__(self, features, bias=True):
        super().__init__()
        self.features = int(features)
        self.weight = nn.Parameter(torch.empty(self.features))
        self.bias = nn.Parameter(torch.empty(self.features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        half_width = self.features ** -0.5
        with torch.no_grad():
            for p in (self.weight, self.bias):
                if p is not None:
                    p.uniform_(-half_width, half_width)

    def forward(self, x):
        y = x * self.weight
        return y if self.bias is None else y + self.bias

    def extra_repr(self):
        return 'features=%d, bias=%s' % (self.features, self.bias is not None)


class CustomDataParallel(nn.Module):
    """Attribute-forwarding wrapper kept for config compatibility (ref base.py:106-116).

    In the reference this wraps ``nn.DataParallel`` but ``fit`` only ever calls ``model.loss``,
    which forwards to the inner module, so no scatter/gather happens.  Here
    multi-GPU training is one process per GPU with an RCCL gradient all-reduce
    (behavenet_amd/fitting/distributed.py); this wrapper only forwards attributes.
    """

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)
