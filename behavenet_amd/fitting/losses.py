"""Loss terms of the autoencoder hot path, on the HIP kernels.

Host-side mirror of the reference ``behavenet/fitting/losses.py`` (same function names,
argument order and return conventions).  Inputs must be fp32 tensors on the GPU; there is no
CPU implementation here (the CPU restatement used for checking lives in ``oracle/``).
"""

import numpy as np
import torch

from behavenet_amd import hip_functions as hf

__all__ = [
    'mse', 'gaussian_ll', 'gaussian_ll_to_mse', 'kl_div_to_std_normal', 'index_code_mi',
    'total_correlation', 'dimension_wise_kl_to_std_normal', 'decomposed_kl', 'subspace_overlap',
    'triplet_loss']

LN2PI = np.log(2 * np.pi)


def mse(y_pred, y_true, masks=None):
    """mean((y_pred - y_true)^2 [* masks]) over ALL elements (ref losses.py:36-59).

    With a mask the divisor is still the element count, not the mask sum.
    """
    return hf.sq_err(y_pred, y_true, masks, 1.0 / y_pred.numel())


def mse_chunks(y_pred, y_true, masks, bounds, chunk_sizes=None):
    """``mse`` of every contiguous frame range in ``bounds`` (same divisor convention) as one
    (n_chunks,) device tensor -- for a batch whose forward ran in a single pass.
    ``chunk_sizes``: frame counts to divide by when ``bounds`` hold only this rank's share of
    each chunk (frame-sharded data parallelism)."""
    fused = y_pred if isinstance(y_pred, hf.FusedPixelLoss) else y_true
    if isinstance(fused, hf.FusedPixelLoss):
        # the decoder evaluated it in the epilogue of its last layer (csrc: k_up_c1v<R, true>)
        assert fused.kind == 'mse' and fused.bounds == list(bounds)
        return fused.chunk_terms
    per_frame = y_pred[0].numel()
    return hf.chunked_sq_err(y_pred, y_true, masks, bounds,
                             hf.pixel_loss_scales('mse', bounds, per_frame, chunk_sizes))


def gaussian_ll(y_pred, y_mean, masks=None, std=1):
    """Diagonal-Gaussian log-likelihood, summed over dims, averaged over frames (ref :62-96)."""
    n_frames = y_pred.shape[0]
    n_dims = int(np.prod(y_pred.shape[1:]))
    log_var = np.log(std ** 2)
    const = -(0.5 * LN2PI + 0.5 * log_var) * n_dims
    return hf.sq_err(y_pred, y_mean, masks, -(0.5 / (std ** 2)) / n_frames) + float(const)


def gaussian_ll_chunks(y_pred, y_mean, masks, bounds, std=1, chunk_sizes=None, const_share=None):
    """``gaussian_ll`` of every contiguous frame range in ``bounds`` as one (n_chunks,) tensor.
    Frame-sharded data parallelism: ``chunk_sizes`` = the global chunk lengths to average over,
    ``const_share`` = per-chunk fraction of the additive constant this rank accounts for (its
    share of the chunk's frames), so that the sum over ranks is the single-device value."""
    fused = y_pred if isinstance(y_pred, hf.FusedPixelLoss) else y_mean
    images = y_mean if fused is y_pred else y_pred
    n_dims = int(np.prod(images.shape[1:]))
    log_var = np.log(std ** 2)
    const = -(0.5 * LN2PI + 0.5 * log_var) * n_dims
    if const_share is None:
        const_t = float(const)
    else:
        const_t = hf.device_constant([float(const) * float(f) for f in const_share], images.device)
    if isinstance(fused, hf.FusedPixelLoss):
        assert fused.kind == 'll' and fused.bounds == list(bounds) and std == 1
        return fused.chunk_terms + const_t
    sizes = chunk_sizes if chunk_sizes is not None else [end - beg for beg, end in bounds]
    return hf.chunked_sq_err(y_pred, y_mean, masks, bounds,
                             [-(0.5 / (std ** 2)) / n for n in sizes]) + const_t


def gaussian_ll_to_mse(ll, n_dims, gaussian_std=1, mse_std=1):
    """Strip the Gaussian constants from a log-likelihood value (ref :99-127); host floats."""
    llc = np.copy(ll)
    llc += (0.5 * LN2PI + 0.5 * np.log(gaussian_std ** 2)) * n_dims
    llc *= -(gaussian_std ** 2) / 0.5
    llc /= n_dims
    llc *= 1.0 / (mse_std ** 2)
    return llc


def kl_div_to_std_normal(mu, logvar):
    """KL(N(mu, exp(logvar)) || N(0, 1)) summed over dims, averaged over frames (ref :130-147)."""
    return hf.kl_to_std_normal(mu, logvar)


# ------------------------------------------------------------------------------------------
# beta-TC decomposition (ref :150-372): one HIP kernel pair (csrc/decomposed_kl.hip) evaluates the
# three terms -- and their gradients -- from the (N, D) inputs; the reference's (N, N, D) pairwise
# tensor and its autograd graph are never materialised.
# ------------------------------------------------------------------------------------------
def decomposed_kl(z, mu, logvar):
    """(index-code MI, total correlation, dimension-wise KL) batch estimates (ref :284-351)."""
    t = hf.decomposed_kl_terms(z, mu, logvar)
    return t[0], t[1], t[2]


def index_code_mi(z, mu, logvar):
    """ref :150-192"""
    return decomposed_kl(z, mu, logvar)[0]


def total_correlation(z, mu, logvar):
    """ref :195-240"""
    return decomposed_kl(z, mu, logvar)[1]


def dimension_wise_kl_to_std_normal(z, mu, logvar):
    """ref :243-281"""
    return decomposed_kl(z, mu, logvar)[2]


def subspace_overlap(A, B, C=None):
    """mean((U U^T - I)^2) for the stacked subspace bases (ref :375-399)."""
    U = torch.cat([A, B] if C is None else [A, B, C], dim=0)
    eye = torch.eye(U.shape[0], device=U.device)
    return torch.mean((torch.matmul(U, U.t()) - eye).pow(2))


def triplet_loss(triplet_loss_obj, z, datasets):
    """Session-separation loss on the background latents of the MSPS-VAE (ref losses.py:402-511).

    ``z`` (N, d) on the device, ``datasets`` (N,) numpy session ids.  Per session its shuffled
    sample indices (``np.random.permutation``, the reference's host RNG consumption: one draw per
    session in id order) are dealt into 3*(n-1) interleaved chunks of equal length; chunks
    (2j, 2j+1) are anchor / positive for the j-th other session, whose next unused chunk from
    index 2*(n-1) on is the negative; the mean anchor-positive distance of every pair is added.
    Normalised by n*(n-1) terms -- except the reference's 3 for two sessions.  A few hundred
    floats: evaluated with torch's elementwise device ops, not a dedicated kernel.
    """
    ids = np.unique(datasets)
    n = len(ids)
    if n not in (2, 3, 4):
        raise NotImplementedError
    n_chunks = 3 * (n - 1)
    perms = [np.random.permutation(np.where(datasets == i)[0]) for i in ids]
    m = int(np.min([len(p) // n_chunks for p in perms]))

    def rows(sess, chunk):
        idx = torch.from_numpy(np.ascontiguousarray(perms[sess][chunk::n_chunks][:m]))
        return z.index_select(0, idx.to(z.device))
    next_neg = [2 * (n - 1)] * n
    loss = 0
    pairs = []
    for a in range(n):
        j = 0
        for b in range(n):
            if b == a:
                continue
            anchor, positive = rows(a, 2 * j), rows(a, 2 * j + 1)
            loss = loss + triplet_loss_obj(anchor, positive, rows(b, next_neg[b]))
            next_neg[b] += 1
            pairs.append((anchor, positive))
            j += 1
    for anchor, positive in pairs:
        loss = loss + torch.pairwise_distance(anchor, positive).mean()
    return loss / (3 if n == 2 else n * (n - 1))
