"""Data-parallel training over RCCL/xGMI: one process per GPU.

The reference has no working multi-GPU path (its ``nn.DataParallel`` wrapper is bypassed by
``fit``).  Here frames shard across ranks and the only exchange is ONE
all-reduce (sum) of the flat gradient arena per optimizer step (35 MB for the default arch).

Two sharding modes:

* ``'trial'`` (weak scaling, default): every rank draws its own trial per step; gradients are
  summed, i.e. one optimizer step consumes ``world_size`` trials.  Not step-for-step identical
  to the single-GPU reference (which steps once per trial).
* ``'frames'`` (strong scaling, parity-exact): all ranks see the same trial; inside each
  200-frame chunk rank r takes the contiguous slice [r*n_c/R, (r+1)*n_c/R) and its loss is
  scaled to the *global* chunk mean, so the summed gradient equals the single-GPU gradient.

Backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` on CPU (tests).
"""

import os

import torch
import torch.distributed as dist


def is_active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun env)."""
    if dist.is_initialized():
        return rank(), world_size()
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1:
        return 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=ws)
    return rank(), world_size()


def shard_bounds(beg, end, r=None, R=None):
    """Contiguous slice of frames [beg, end) owned by rank r of R ('frames' mode)."""
    r = rank() if r is None else r
    R = world_size() if R is None else R
    n = end - beg
    return beg + (r * n) // R, beg + ((r + 1) * n) // R


def all_reduce_flat_(flat, average=False):
    """In-place sum (or mean) of one flat buffer over all ranks; no-op for world_size 1."""
    if not is_active():
        return flat
    if flat.is_cuda:
        from behavenet_amd.hip_functions import join_side_streams
        join_side_streams()
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat.div_(world_size())
    return flat


def all_reduce_scalars(values):
    """Sum a short list of python floats over ranks (loss bookkeeping)."""
    if not is_active():
        return list(values)
    dev = torch.device('cuda', torch.cuda.current_device()) \
        if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor(list(values), dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().tolist()


def broadcast_parameters_(flat, src=0):
    """Make every rank start from rank ``src``'s parameters."""
    if is_active():
        dist.broadcast(flat, src=src)
    return flat
