"""On-disk trial stores behind the data generator.

The reference keeps one ``data.hdf5`` per session with one dataset per trial,
``<signal>/trial_%04i`` (``images`` uint8 (T, C, H, W), ``labels`` / ``labels_sc`` /
``labels_masks`` / ``masks`` float; docs/source/data_structure.rst:17-75, read by
``SingleSessionDatasetBatchedLoad.__getitem__``, data_generator.py:233-300).  Two backends with the
same key space:

* ``.hdf5`` / ``.h5`` -- through ``h5py`` when it is installed (it is not in the build image:
  opening such a file then raises ImportError with this explanation);
* ``.npz`` -- a numpy zip mirror with IDENTICAL member names (``images/trial_0000`` ...), readable
  member by member without loading the file; ``write_npz_session`` / ``hdf5_to_npz`` produce it.
"""

import os
import zipfile

import numpy as np

__all__ = ['open_trial_store', 'write_npz_session', 'hdf5_to_npz', 'TRIAL_KEY']

TRIAL_KEY = 'trial_%04i'


class _NpzStore(object):
    def __init__(self, path):
        self.path = path
        self._npz = np.load(path, allow_pickle=False)
        self._members = {}
        for name in self._npz.files:
            signal, _, trial = name.partition('/')
            self._members.setdefault(signal, []).append(trial)

    def signals(self):
        return sorted(self._members)

    def n_trials(self, signal):
        return len(self._members[signal])

    def read(self, signal, trial):
        return self._npz['%s/%s' % (signal, TRIAL_KEY % trial)]

    def close(self):
        self._npz.close()


class _Hdf5Store(object):
    def __init__(self, path):
        try:
            import h5py
        except ImportError as e:            # pragma: no cover - h5py is absent from the image
            raise ImportError(
                'reading %s needs h5py, which is not installed; convert the session once with '
                'behavenet_amd.data.trial_store.hdf5_to_npz on a machine that has it' % path) from e
        self.path = path
        self._h5py = h5py

    def _open(self):
        return self._h5py.File(self.path, 'r', libver='latest', swmr=True)

    def signals(self):
        with self._open() as f:
            return sorted(f.keys())

    def n_trials(self, signal):
        with self._open() as f:
            return len(f[signal])

    def read(self, signal, trial):
        with self._open() as f:
            return f[signal][TRIAL_KEY % trial][()]

    def close(self):
        pass


def open_trial_store(path):
    """Store for ``path``; a missing ``data.hdf5`` falls back to a ``data.npz`` next to it."""
    if not os.path.exists(path):
        mirror = os.path.splitext(path)[0] + '.npz'
        if os.path.exists(mirror):
            path = mirror
        else:
            raise FileNotFoundError(path)
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        return _NpzStore(path)
    if ext in ('.hdf5', '.h5'):
        return _Hdf5Store(path)
    raise ValueError('unknown trial store format "%s"' % path)


def write_npz_session(path, signals):
    """``signals``: {'images': [uint8 (T,C,H,W) per trial], 'labels': [...], ...} -> ``path``.

    Stored uncompressed so that a trial is one contiguous read."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with zipfile.ZipFile(path, 'w', zipfile.ZIP_STORED, allowZip64=True) as zf:
        for signal, trials in signals.items():
            for i, arr in enumerate(trials):
                with zf.open('%s/%s.npy' % (signal, TRIAL_KEY % i), 'w', force_zip64=True) as f:
                    np.lib.format.write_array(f, np.ascontiguousarray(arr), allow_pickle=False)
    return path


def hdf5_to_npz(hdf5_path, npz_path=None):
    """One-off conversion of a reference ``data.hdf5`` (needs h5py)."""
    src = _Hdf5Store(hdf5_path)
    npz_path = npz_path or os.path.splitext(hdf5_path)[0] + '.npz'
    signals = {s: [src.read(s, t) for t in range(src.n_trials(s))] for s in src.signals()}
    return write_npz_session(npz_path, signals)
