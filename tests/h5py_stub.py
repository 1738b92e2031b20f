"""npz-backed stand-in for the part of ``h5py`` the data path uses (the package is not in the
image): read-only ``h5py.File(path)`` as a context manager, ``f[signal]['trial_%04i'][()]``,
``len(f[signal])``, ``f.keys()`` -- served from the ``data.npz`` mirror next to ``path``.
Used by tests/golden/make_golden.py to import the REFERENCE's data generator, and by
tests/test_fit_host.py to execute this repo's HDF5 trial-store backend."""

import os
import sys
import types

import numpy as np


def install():
    """``h5py`` is not in the image; the reference's data generator only opens files read-only and
    indexes ``f[signal]['trial_%04i'][()]`` / ``len(f[signal])``.  This stand-in serves exactly
    that from the ``data.npz`` mirror next to the requested ``data.hdf5`` (same member names,
    behavenet_amd/data/trial_store.py)."""
    shim = types.ModuleType('h5py')

    class _Dataset(object):
        def __init__(self, arr):
            self._arr = arr

        def __getitem__(self, key):
            assert key == ()
            return self._arr

    class _Group(object):
        def __init__(self, npz, signal):
            self._npz, self._signal = npz, signal
            self._names = [n for n in npz.files if n.startswith(signal + '/')]

        def __len__(self):
            return len(self._names)

        def __getitem__(self, key):
            return _Dataset(self._npz['%s/%s' % (self._signal, key)])

    class File(object):
        def __init__(self, path, mode='r', **kwargs):
            assert mode == 'r'
            self._npz = np.load(os.path.splitext(str(path))[0] + '.npz', allow_pickle=False)

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            self._npz.close()
            return False

        def __getitem__(self, signal):
            return _Group(self._npz, signal)

        def keys(self):
            return sorted(set(n.split('/')[0] for n in self._npz.files))
    shim.File = File
    sys.modules['h5py'] = shim


