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
import threading
import zipfile

import numpy as np

__all__ = ['open_trial_store', 'write_npz_session', 'hdf5_to_npz', 'TRIAL_KEY']

TRIAL_KEY = 'trial_%04i'


class _NpzStore(object):
    """Uncompressed members (what ``write_npz_session`` produces) are read STRAIGHT from the file:
    the member's payload is one contiguous run at a known offset (zip local header + npy header),
    so a trial is one ``pread`` into the caller's buffer -- a pinned staging tensor, say -- with
    the GIL released; numpy's own ``NpzFile`` path goes through ``zipfile`` (a CRC pass and two
    copies per member, ~4 ms per 4.2 MB trial, all under the GIL).  Compressed members fall back
    to it."""

    def __init__(self, path):
        self.path = path
        self._npz = np.load(path, allow_pickle=False)
        self._members = {}
        for name in self._npz.files:
            signal, _, trial = name.partition('/')
            self._members.setdefault(signal, []).append(trial)
        self._info = {}
        for zi in self._npz.zip.infolist():
            self._info[zi.filename] = zi
        self._layout = {}
        # opened here, not on first use: the reader pool calls layout() / read_into() from several threads
        # (two of them opening at once leaked a descriptor per store)
        self._fd = os.open(self.path, os.O_RDONLY)
        self._lock = threading.Lock()
        self._readers = 0                   # preadv calls in flight
        self._idle = threading.Condition(self._lock)

    def signals(self):
        return sorted(self._members)

    def n_trials(self, signal):
        return len(self._members[signal])

    def _member_layout(self, key):
        """(dtype, shape, payload offset in the file) of a stored, C-ordered member, else None."""
        got = self._layout.get(key, False)
        if got is not False:
            return got
        out = None
        zi = self._info.get(key + '.npy')
        if zi is not None and zi.compress_type == zipfile.ZIP_STORED:
            if self._fd is None:
                raise ValueError('%s: trial store is closed' % self.path)
            head = os.pread(self._fd, 30, zi.header_offset)
            if len(head) == 30 and head[:4] == b'PK\x03\x04':
                n_name = int.from_bytes(head[26:28], 'little')
                n_extra = int.from_bytes(head[28:30], 'little')
                npy_at = zi.header_offset + 30 + n_name + n_extra
                import io
                buf = io.BytesIO(os.pread(self._fd, min(4096, zi.file_size), npy_at))
                try:
                    version = np.lib.format.read_magic(buf)
                    if version == (1, 0):
                        shape, fortran, dtype = np.lib.format.read_array_header_1_0(buf)
                    else:
                        shape, fortran, dtype = np.lib.format.read_array_header_2_0(buf)
                    if not fortran and not dtype.hasobject and \
                            buf.tell() + int(np.prod(shape)) * dtype.itemsize == zi.file_size:
                        out = (dtype, tuple(shape), npy_at + buf.tell())
                except ValueError:
                    out = None
        self._layout[key] = out
        return out

    def layout(self, signal, trial):
        """(dtype, shape) of a trial without reading it, or None (compressed / foreign member)."""
        lay = self._member_layout('%s/%s' % (signal, TRIAL_KEY % trial))
        return None if lay is None else lay[:2]

    def read_into(self, signal, trial, out):
        """Fill ``out`` (a C-contiguous numpy array of the trial's dtype and shape, e.g. the numpy
        view of a pinned tensor) with the trial; -> ``out``.  One ``preadv`` for stored members."""
        key = '%s/%s' % (signal, TRIAL_KEY % trial)
        lay = self._member_layout(key)
        if lay is None:
            out[...] = self._npz[key]
            return out
        dtype, shape, offset = lay
        if out.dtype != dtype or tuple(out.shape) != shape or not out.flags['C_CONTIGUOUS']:
            raise ValueError('read_into: buffer %s %s does not match the stored trial %s %s' % (
                out.dtype, out.shape, dtype, shape))
        view = memoryview(out).cast('B')
        done, total = 0, view.nbytes
        with self._lock:
            if self._fd is None:
                raise ValueError('%s: trial store is closed' % self.path)
            fd = self._fd
            self._readers += 1
        try:
            while done < total:
                n = os.preadv(fd, [view[done:]], offset + done)
                if n <= 0:
                    raise IOError('%s: short read of %s' % (self.path, key))
                done += n
        finally:
            with self._idle:
                self._readers -= 1
                self._idle.notify_all()
        return out

    def read(self, signal, trial):
        lay = self._member_layout('%s/%s' % (signal, TRIAL_KEY % trial))
        if lay is None:
            return self._npz['%s/%s' % (signal, TRIAL_KEY % trial)]
        return self.read_into(signal, trial, np.empty(lay[1], dtype=lay[0]))

    def close(self):
        """Waits for the reads in flight (a reader thread inside ``preadv``) before the descriptor goes."""
        with self._idle:
            fd, self._fd = self._fd, None
            while self._readers:
                self._idle.wait()
        self._npz.close()
        if fd is not None:
            os.close(fd)


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
