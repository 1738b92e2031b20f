"""Helpers shared by the golden-vector generator and the tests that consume the fixtures."""

import numpy as np


def checksum(a):
    """[sum, sum|.|, sum(.^2), size] in float64 -- pins a large tensor without storing it."""
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.size], dtype=np.float64)


def strided_sample(a, n=64):
    a = np.asarray(a).ravel()
    idx = np.linspace(0, a.size - 1, n).astype(np.int64)
    return a[idx].astype(np.float32)


def checksum_close(got, want, rtol):
    """Compare two checksum vectors; the abs-sum sets the scale for the signed sum."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if got[3] != want[3]:
        return False
    scale = max(want[1], 1e-30)
    return (abs(got[0] - want[0]) <= rtol * scale and
            abs(got[1] - want[1]) <= rtol * scale and
            abs(got[2] - want[2]) <= 2 * rtol * max(want[2], 1e-30))


# the synthetic-data recipes live in the package (bench.py and the tools use them too)
from behavenet_amd.data.synthetic import (  # noqa: E402,F401
    base_hparams, make_frames, make_frames_u8, make_labels, make_labels_sc, make_masks)
