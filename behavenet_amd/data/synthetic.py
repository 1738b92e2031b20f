"""Synthetic inputs in the recipe of the reference's own integration test
(``tests/integration.py:105-116``: uint8 noise frames consumed as ``float32 / 255``, standard-normal
labels) and the hparams a model needs on top of a planned architecture.  Used by ``bench.py``, the
tools and the tests (``tests/golden_utils.py`` re-exports these): the benchmark must not import the
test package."""

import numpy as np


def make_frames(n_frames, dim, seed):
    """uint8 noise frames consumed as float32/255 (reference tests/integration.py:105-109)."""
    rng = np.random.default_rng(seed)
    u8 = rng.integers(0, 255, size=(n_frames,) + tuple(dim), dtype=np.uint8)
    return u8.astype(np.float32) / 255


def make_frames_u8(n_frames, dim, seed):
    """The same frames as they are stored on disk (``images/trial_%04i``, uint8)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 255, size=(n_frames,) + tuple(dim), dtype=np.uint8)


def make_labels(n_frames, n_labels, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n_frames, n_labels)).astype(np.float32)


def make_masks(n_frames, dim, seed):
    """0/1 pixel masks, ~70 % ones (the `masks` signal, ref losses.py:56-59)."""
    rng = np.random.default_rng(seed)
    return (rng.random((n_frames,) + tuple(dim)) < 0.7).astype(np.float32)


def make_labels_sc(n_frames, n_maps, dim, seed):
    """One-hot label maps (N, n_maps, H, W): one pixel set per frame and map (the `labels_sc`
    signal of the conditional encoder, ref aes.py:818-826)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_frames, n_maps, dim[1], dim[2]), dtype=np.float32)
    ys = rng.integers(0, dim[1], size=(n_frames, n_maps))
    xs = rng.integers(0, dim[2], size=(n_frames, n_maps))
    for n in range(n_frames):
        for k in range(n_maps):
            out[n, k, ys[n, k], xs[n, k]] = 1.0
    return out


def base_hparams(arch, model_class, extra=None):
    hp = dict(arch)
    hp.update({
        'model_class': model_class, 'device': 'cpu', 'learning_rate': 1e-4, 'l2_reg': 0.0,
        'fit_sess_io_layers': False, 'n_datasets': 1, 'rng_seed_model': 0})
    if extra:
        hp.update(extra)
    return hp
