"""In-memory trial generator with the duck type ``fit()`` / ``export_latents`` consume.

The reference's HDF5-backed ``ConcatSessionsGenerator`` (behavenet/data/data_generator.py:432-633)
is I/O and out of scope for the kernels; what the hot path needs is its
*interface*: ``n_datasets``, ``n_tot_batches[dtype]``, ``reset_iterators(dtype)``,
``next_batch(dtype) -> (dict, int)`` with ``data[key]`` shaped ``(1, B, ...)`` and
``datasets[i].{n_trials, batch_idxs, lab, expt, animal, session}``.  One batch = one trial.

Frames are synthesised like the reference's integration test (tests/integration.py:105-109):
``uint8`` noise, consumed as ``float32 / 255``.  Where the trial lives is a choice:

* ``placement='host'``      -- float32 on the host, copied per batch (what the reference does,
                              data_generator.py:251-263,630-631)
* ``placement='device_u8'`` -- uint8 resident in HBM, converted per batch by a HIP kernel
* ``placement='device'``    -- float32 resident in HBM (bench default: inputs already resident)
* ``placement='host_u8'``   -- uint8 in PINNED host memory (the reference's on-disk dtype,
                              docs/source/data_structure.rst:17-21): the trial that heads the
                              queue is copied to the device on a copy stream while the current
                              one is being trained on (4.2 MB instead of the 16.8 MB float32 copy
                              of the reference), and converted by ``bn_u8_to_unit_float``
"""

import os
from collections import OrderedDict

import numpy as np
import torch

from behavenet_amd import _hip

__all__ = ['split_trials', 'SyntheticSession', 'SyntheticSessionsGenerator',
           'SingleSessionDatasetBatchedLoad', 'SingleSessionDataset', 'ConcatSessionsGenerator',
           'ConcatSessionsGeneratorMulti']


def split_trials(n_trials, rng_seed=0, train_tr=8, val_tr=1, test_tr=1, gap_tr=0):
    """Blocks of ``train | gap | val | gap | test | gap`` trials (ref data_generator.py:42-134)."""
    np.random.seed(rng_seed)
    per_block = train_tr + val_tr + test_tr + 3 * gap_tr
    n_blocks = n_trials // per_block
    if n_blocks == 0:
        raise ValueError(
            'Not enough trials (n=%i) for the train/test/val/gap values %i/%i/%i/%i' %
            (n_trials, train_tr, val_tr, test_tr, gap_tr))
    leftover = n_trials - per_block * n_blocks
    offset = np.random.randint(0, high=leftover) if leftover > 0 else 0
    out = {'train': [], 'test': [], 'val': []}
    for block in np.random.permutation(n_blocks):
        cur = block * per_block + offset
        out['train'].append(np.arange(cur, cur + train_tr))
        cur += train_tr + gap_tr
        out['val'].append(np.arange(cur, cur + val_tr))
        cur += val_tr + gap_tr
        out['test'].append(np.arange(cur, cur + test_tr))
    return {k: np.concatenate(v, axis=0) for k, v in out.items()}


class _Skipped(object):
    """What ``next_batch(skip=True)`` returns in place of a sample (not None: None = exhausted)."""

    def __repr__(self):
        return 'SKIPPED'


SKIPPED = _Skipped()


class SyntheticSession(object):
    """One session of synthetic trials (stands in for ``SingleSessionDatasetBatchedLoad``)."""

    def __init__(self, n_trials, frames_per_trial, img_shape, seed=0, n_labels=0,
                 trial_splits='8;1;1;0', rng_seed_data=0, name=('lab', 'expt', 'animal', 'sess')):
        self.lab, self.expt, self.animal, self.session = name
        self.n_trials = n_trials
        rng = np.random.default_rng(seed)
        if np.isscalar(frames_per_trial):
            frames_per_trial = [int(frames_per_trial)] * n_trials
        self.images_u8 = [
            rng.integers(0, 255, size=(int(t),) + tuple(img_shape), dtype=np.uint8)
            for t in frames_per_trial]
        self.labels = None
        if n_labels > 0:
            self.labels = [rng.standard_normal((int(t), n_labels)).astype(np.float32)
                           for t in frames_per_trial]
        tr, va, te, gap = [int(v) for v in trial_splits.split(';')]
        self.batch_idxs = split_trials(n_trials, rng_seed_data, tr, va, te, gap)
        self.n_batches = {k: len(v) for k, v in self.batch_idxs.items()}


class SyntheticSessionsGenerator(object):
    """Serves one trial per ``next_batch`` call; sessions are drawn with ``np.random.choice``
    and trials in ``torch.randperm`` order, so ``fit``'s per-epoch reseeding controls both."""

    _dtypes = ['train', 'val', 'test']

    def __init__(self, sessions, device='cuda', placement='device', n_sessions_per_batch=1):
        if placement not in ('host', 'device_u8', 'device', 'host_u8'):
            raise ValueError('unknown placement "%s"' % placement)
        if n_sessions_per_batch > 4:
            raise NotImplementedError   # losses.triplet_loss covers 2-4 sessions (ref :687-689)
        self.n_sessions_per_batch = int(n_sessions_per_batch)
        self.datasets = list(sessions)
        self.n_datasets = len(self.datasets)
        self.device = device
        self.placement = placement
        self.n_tot_batches = {
            k: sum(ds.n_batches[k] for ds in self.datasets) for k in self._dtypes}
        tot = float(sum(ds.n_batches['train'] for ds in self.datasets))
        self.batch_ratios = [ds.n_batches['train'] / tot for ds in self.datasets]
        if self.n_sessions_per_batch > 1:
            # several batches are served per training iteration (ref data_generator.py:697-699)
            self.n_tot_batches['train'] = int(
                self.n_tot_batches['train'] / self.n_sessions_per_batch)
        self._store = [self._load_session(ds) for ds in self.datasets]
        # host_u8 prefetcher: two device staging buffers per trial shape, one copy stream
        self._pf = None          # (key, device uint8 tensor, ready event) of the prefetched trial
        self.lookahead = 0       # queue position of this rank's next trial ('trial' mode: W - 1)
        # uint8 placements: hand out the device uint8 frames as they are (inference: the first conv
        # layer converts in flight, no float copy of the trial is made; fitting/eval.py sets it)
        self.serve_uint8 = False
        self.read_ahead = 4      # file-backed sessions: trials requested from the reader threads ahead of use
        self._skip_pred = None   # predicate of the last next_batch(skip=callable): steers the look-ahead
        self._last_slot = None
        self._pf_bufs = {}
        self._pf_stream = None
        self._queues = [{k: [] for k in self._dtypes} for _ in self.datasets]
        for k in self._dtypes:
            self.reset_iterators(k)

    def _load_session(self, ds):
        """(image trials, label trials | None) in the chosen placement."""
        placement, device = self.placement, self.device
        trials = []
        for u8 in ds.images_u8:
            if placement == 'host':
                t = torch.from_numpy(u8.astype(np.float32) / 255)
                if device == 'cuda':
                    t = t.pin_memory()
            elif placement == 'device_u8':
                t = torch.from_numpy(u8).to(device)
            elif placement == 'host_u8':
                t = torch.from_numpy(u8).pin_memory()
            else:
                t = torch.from_numpy(u8.astype(np.float32) / 255).to(device)
            trials.append(t)
        labels = None
        if ds.labels is not None:
            labels = [torch.from_numpy(l).to(device) for l in ds.labels]
        return trials, labels

    def __len__(self):
        return self.n_datasets

    def reset_iterators(self, dtype):
        """All trials of ``dtype`` available again, in a fresh random order per session.

        Consumes the random streams exactly as the reference does (pinned by
        tests/golden/generator.json): its ``reset_iterators`` builds ``iter(DataLoader(...,
        sampler=SubsetRandomSampler(batch_idxs)))`` per session, which draws ONE int64 from
        torch's global generator on creation (the loader's base seed), while the permutation
        itself (``torch.randperm``) is only drawn when the first trial of that session is asked
        for (data_generator.py:585-599 + torch.utils.data)."""
        kinds = self._dtypes if dtype == 'all' else [dtype]
        for i, ds in enumerate(self.datasets):
            for k in kinds:
                torch.empty((), dtype=torch.int64).random_()
                self._queues[i][k] = None          # permutation pending

    def _queue(self, sess, dtype):
        q = self._queues[sess][dtype]
        if q is None:
            idxs = self.datasets[sess].batch_idxs[dtype]
            q = [int(idxs[j]) for j in torch.randperm(len(idxs)).tolist()]
            self._queues[sess][dtype] = q
        return q

    def _n_left(self, sess, dtype):
        q = self._queues[sess][dtype]
        return len(self.datasets[sess].batch_idxs[dtype]) if q is None else len(q)

    def next_batch(self, dtype, return_multiple=True, skip=False):
        """One trial, or -- for training with ``n_sessions_per_batch`` > 1 -- a list of trials
        from that many DIFFERENT sessions plus the list of their ids (the multi-session batches
        of the MSPS-VAE, ref data_generator.py:712-790): sessions are drawn without replacement
        by ``batch_ratios``; ``(None, None)`` once too few sessions have trials left.

        ``skip=True`` (data-parallel 'trial' mode: the trial belongs to another rank) advances
        the generator exactly as a normal call would -- same queue pops, same RNG draws -- but
        does not read, copy or convert the trial: returns ``(SKIPPED, session)``.  ``skip`` may
        also be a predicate ``skip(session, trial) -> bool`` asked once the trial is known
        (``export_latents``: a trial belongs to a rank by its identity); the look-ahead of the
        uint8 feed then requests only trials the predicate lets through."""
        self._skip_pred = skip if callable(skip) else None
        if self.n_sessions_per_batch > 1 and dtype == 'train' and return_multiple:
            samples, sessions = [], []
            ratios = np.array(self.batch_ratios, dtype=np.float64)
            for k in range(self.n_sessions_per_batch):
                while True:
                    if np.sum(ratios > 0) < self.n_sessions_per_batch - k:
                        return None, None
                    sess = int(np.random.choice(np.arange(self.n_datasets), p=ratios))
                    ratios[sess] = 0
                    if np.sum(ratios) > 0:
                        ratios = ratios / np.sum(ratios)
                    if self._n_left(sess, dtype):
                        trial = self._queue(sess, dtype).pop(0)
                        break
                if callable(skip):      # (multi-session batches are skipped whole: True / False only)
                    raise ValueError('next_batch: a skip predicate needs one trial per batch '
                                     '(return_multiple=False)')
                samples.append(SKIPPED if skip else self._sample(sess, trial, dtype))
                sessions.append(sess)
            return (SKIPPED if skip else samples), sessions
        if all(self._n_left(i, dtype) == 0 for i in range(self.n_datasets)):
            return None, None
        while True:
            sess = int(np.random.choice(np.arange(self.n_datasets), p=self.batch_ratios))
            if self._n_left(sess, dtype):
                trial = self._queue(sess, dtype).pop(0)
                break
        if (skip(sess, trial) if callable(skip) else skip):
            return SKIPPED, sess
        return self._sample(sess, trial, dtype), sess

    def _sample(self, sess, trial, dtype):
        trials, labels = self._store[sess]
        img = trials[trial]
        if self.placement == 'host':
            img = img.to(self.device, non_blocking=True)
        elif self.placement == 'device_u8':
            img = img if self.serve_uint8 else _hip.u8_to_unit_float(img)
        elif self.placement == 'host_u8':
            img = self._fetch_host_u8(sess, trial, dtype)
        sample = {'images': img[None], 'batch_idx': torch.tensor([trial])}
        if labels is not None:
            sample['labels'] = labels[trial][None]
        self._extra_signals(sample, sess, trial)
        return sample

    def _extra_signals(self, sample, sess, trial):
        pass

    # -- pinned uint8 feed with one-trial look-ahead ------------------------------------------
    def _staging(self, shape, slot):
        """Device staging buffer of one trial shape and slot (kept for the generator's life).

        Allocated under the COPY stream: the caching allocator hands a stream's freed blocks only
        to later allocations of the same stream, so a block that backed an activation of the
        previous step on the compute stream (whose kernels may still be running) can never
        become a staging buffer that the copy stream overwrites out of order.  (It did, once per
        buffer, when these were allocated on the compute stream: raw frame bytes landed in live
        activations -- 0.4 % of random byte quadruples are NaN bit patterns.)"""
        key = (tuple(shape), slot)
        buf = self._pf_bufs.get(key)
        if buf is None:
            with torch.cuda.stream(self._pf_stream):
                buf = torch.empty(shape, dtype=torch.uint8, device=self.device)
            self._pf_bufs[key] = buf
        return buf

    def _upcoming(self, sess, dtype, n):
        """Up to ``n`` trials of this session's queue that THIS process will ask for next, in order:
        those the skip predicate lets through (export_latents), every W-th one ('trial' mode of
        ``fit``: ``lookahead`` = W - 1 trials in between go to the other ranks), or simply the head."""
        queue = self._queues[sess][dtype]      # (None: the session's order is not drawn yet)
        if not queue:
            return []
        pred = self._skip_pred
        if pred is not None:
            out = []
            for t in queue:
                if not pred(sess, t):
                    out.append(t)
                    if len(out) == n:
                        break
            return out
        # (only the TRAINING loop deals trials out to the ranks; validation and test loops consume
        # every trial on every rank, their next request is the head of the queue)
        ahead = int(getattr(self, 'lookahead', 0)) if dtype == 'train' else 0
        return list(queue[ahead::ahead + 1][:n])

    def _fetch_host_u8(self, sess, trial, dtype):
        main = torch.cuda.current_stream()
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream()
            self._pf_slot = 0
            self._pf_done = [None, None]      # main-stream events: staging slot consumed
        if self._last_slot is not None:
            # whatever consumed the trial served last (the conversion kernel, or -- serve_uint8 --
            # the encoder's first layer reading the staging buffer itself) has been queued on the
            # main stream by now: the slot is free once THIS point of the stream is reached
            done = torch.cuda.Event()
            done.record(main)
            self._pf_done[self._last_slot] = done
        store = self._store[sess][0]
        pf = self._pf
        self._pf = None
        if pf is not None and pf[0] == (sess, trial):
            _, dev_u8, ready, slot = pf
            main.wait_event(ready)
        else:
            # nothing (or something else) was prefetched: copy on the main stream
            host = store[trial]
            slot = self._pf_slot
            dev_u8 = self._staging(host.shape, slot)
            if pf is not None:
                # the discarded look-ahead copy targets this slot: let it land first, or it could
                # overwrite the trial copied below
                main.wait_event(pf[2])
            if self._pf_done[slot] is not None:
                main.wait_event(self._pf_done[slot])
            dev_u8.copy_(host, non_blocking=True)
        img = dev_u8 if self.serve_uint8 else _hip.u8_to_unit_float(dev_u8)
        done = torch.cuda.Event()
        done.record(main)
        self._pf_done[slot] = done
        self._last_slot = slot
        self._pf_slot = slot ^ 1
        # look ahead: the next trials this process will ask this session for.  File-backed sessions
        # hand them to their reader threads now (one pread per trial into pinned memory, off the
        # main thread); the first of them is also copied to the device on the copy stream.
        coming = self._upcoming(sess, dtype, max(1, int(self.read_ahead)))
        if coming and hasattr(store, 'prefetch'):
            for t in coming:
                store.prefetch(t)
        if coming:
            nxt = coming[0]
            host = store[nxt]
            nslot = self._pf_slot
            buf = self._staging(host.shape, nslot)
            with torch.cuda.stream(self._pf_stream):
                if self._pf_done[nslot] is not None:
                    self._pf_stream.wait_event(self._pf_done[nslot])
                buf.copy_(host, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(self._pf_stream)
            self._pf = ((sess, nxt), buf, ready, nslot)
        return img


# ---------------------------------------------------------------------------------------------
# file-backed sessions
# ---------------------------------------------------------------------------------------------
_FRAME_SIGNALS = ('images', 'masks', 'labels', 'labels_sc', 'labels_masks')


class _LazyTrials(object):
    """List-like over the trials of one signal: read from the store on first use, then kept (the
    images as PINNED uint8 -- 1 byte per pixel, a quarter of the reference's float32 batches).

    ``prefetch(trial)`` hands the read to a small pool of reader threads (the generator's
    look-ahead calls it for the next few trials): with ``direct`` set the trial goes from the file
    into a pinned tensor by one ``pread`` (``trial_store._NpzStore.read_into``: no zipfile pass, no
    intermediate copy, GIL released), so the disk side of trial k + 1 .. k + 4 runs underneath the
    device's work on trial k."""

    _pool = None

    def __init__(self, dataset, signal, convert, direct=None):
        self.dataset, self.signal, self.convert, self.direct = dataset, signal, convert, direct
        self.cache = {}
        self._pending = {}

    def __len__(self):
        return self.dataset.n_trials

    def _load(self, trial):
        if self.direct is not None:
            t = self.direct(trial)
            if t is not None:
                return t
        return self.convert(self.dataset.read(self.signal, trial))

    def prefetch(self, trial):
        if trial in self.cache or trial in self._pending:
            return
        if _LazyTrials._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _LazyTrials._pool = ThreadPoolExecutor(
                max_workers=int(os.environ.get('BN_READ_THREADS', '2')),      # (1: 332 k, 2: 341 k, 3: 327 k frames/s end to end)
                thread_name_prefix='bn-trial-reader')
        self._pending[trial] = _LazyTrials._pool.submit(self._load, trial)

    def __getitem__(self, trial):
        t = self.cache.get(trial)
        if t is None:
            fut = self._pending.pop(trial, None)
            t = fut.result() if fut is not None else self._load(trial)
            if self.dataset.keep_in_memory:
                self.cache[trial] = t
            elif len(self._pending) > 64:
                # (requested ahead but never asked for: do not let them pile up)
                for k in list(self._pending)[:-32]:
                    self._pending.pop(k).cancel()
        return t


class SingleSessionDatasetBatchedLoad(object):
    """One session on disk, one trial per read (ref data_generator.py:137-343).

    ``paths`` may name the reference's ``data.hdf5`` (needs h5py) or its ``data.npz`` mirror
    (behavenet_amd/data/trial_store.py); if the hdf5 is absent the mirror next to it is used.
    Signals on the autoencoder path only: images, masks, labels, labels_sc, labels_masks.
    """

    def __init__(self, data_dir, lab='', expt='', animal='', session='', signals=None,
                 transforms=None, paths=None, device='cpu', as_numpy=False, keep_in_memory=True):
        from behavenet_amd.data.trial_store import open_trial_store
        self.lab, self.expt, self.animal, self.session = lab, expt, animal, session
        self.data_dir = os.path.join(data_dir, lab, expt, animal, session)
        self.name = os.path.join(lab, expt, animal, session)
        self.sess_str = '%s_%s_%s_%s' % (lab, expt, animal, session)
        self.signals = list(signals) if signals is not None else ['images']
        transforms = transforms if transforms is not None else [None] * len(self.signals)
        paths = paths if paths is not None else \
            [os.path.join(self.data_dir, 'data.hdf5')] * len(self.signals)
        self.transforms = OrderedDict(zip(self.signals, transforms))
        self.paths = OrderedDict(zip(self.signals, paths))
        for signal in self.signals:
            if signal not in _FRAME_SIGNALS:
                raise NotImplementedError(
                    'signal "%s" is outside the autoencoder path (supported: %s)' % (
                        signal, ', '.join(_FRAME_SIGNALS)))
        self._stores = {}
        self.n_trials = None
        for signal in self.signals:
            if self.n_trials is None and signal != 'masks':
                self.n_trials = self._store(signal).n_trials(signal)
        if self.n_trials is None:
            self.n_trials = self._store(self.signals[0]).n_trials(self.signals[0])
        self.batch_idxs = None
        self.n_batches = None
        self.device = device
        self.as_numpy = as_numpy
        self.keep_in_memory = keep_in_memory

    def _store(self, signal):
        from behavenet_amd.data.trial_store import open_trial_store
        path = self.paths[signal]
        if path not in self._stores:
            self._stores[path] = open_trial_store(path)
        return self._stores[path]

    def __str__(self):
        out = '%s\n' % self.sess_str
        out += '    signals: {}\n'.format(self.signals)
        out += '    transforms: {}\n'.format(self.transforms)
        out += '    paths: {}\n'.format(self.paths)
        return out

    def __len__(self):
        return self.n_trials

    def read(self, signal, trial):
        """Raw array of one trial as stored (images: uint8)."""
        return np.asarray(self._store(signal).read(signal, trial))

    def __getitem__(self, idx):
        """The reference's sample dict for one trial on the HOST: images float32/255, the other
        signals float32, transforms applied, ``batch_idx`` (ref :222-300)."""
        if idx is None:
            raise NotImplementedError('Cannot currently load all data as torch tensors')
        sample = OrderedDict()
        for signal in self.signals:
            arr = self.read(signal, idx).astype('float32')
            if signal == 'images':
                arr = arr / 255
            if signal == 'masks':
                arr = arr[0]         # the reference's indexing (see ConcatSessionsGenerator)
            if self.transforms[signal]:
                arr = self.transforms[signal](arr)
            sample[signal] = arr if self.as_numpy else torch.from_numpy(arr).float()
        sample['batch_idx'] = idx
        return sample


class ConcatSessionsGenerator(SyntheticSessionsGenerator):
    """File-backed sessions behind the same iteration logic (ref data_generator.py:432-633).

    Same constructor surface as the reference (``ids_list`` of {'lab','expt','animal','session'}
    dicts, ``signals_list`` / ``transforms_list`` / ``paths_list`` per session, ``trial_splits``
    dict, ``train_frac``, ``rng_seed``).  Images stay uint8 end to end: read from the store,
    pinned on the host, copied to the device one trial ahead on a copy stream, converted by
    ``bn_u8_to_unit_float`` (``placement='host_u8'``, the default); transforms on the images are
    therefore not supported on this path (``SingleSessionDatasetBatchedLoad.__getitem__`` applies
    them on the host for other uses).
    """

    def __init__(self, data_dir, ids_list, signals_list=None, transforms_list=None,
                 paths_list=None, device='cuda', as_numpy=False, batch_load=True, rng_seed=0,
                 trial_splits=None, train_frac=1.0, placement='host_u8', n_sessions_per_batch=1,
                 keep_in_memory=True):
        if isinstance(ids_list, dict):
            ids_list = [ids_list]
        n = len(ids_list)
        self.ids = ids_list
        self.as_numpy, self.batch_load = as_numpy, batch_load
        # host path: trials go through SingleSessionDatasetBatchedLoad.__getitem__ (float32 / 255 on
        # the host, transforms applied) as in the reference, instead of the pinned-uint8 feed --
        # for generators that hand out numpy arrays (as_numpy, ref :251-263,625-628) and for
        # sessions with an image transform
        self._host_path = bool(as_numpy)
        self.signals = signals_list if signals_list is not None else [['images']] * n
        self.transforms = transforms_list if transforms_list is not None else \
            [[None] * len(sig) for sig in self.signals]
        self.paths = paths_list if paths_list is not None else [None] * n
        datasets = []
        self.datasets_info = []
        for ids, signals, transforms, paths in zip(ids_list, self.signals, self.transforms,
                                                   self.paths):
            if transforms is not None and 'images' in signals and \
                    transforms[list(signals).index('images')] is not None:
                # a transform works on the float frames (ref :315-317): such sessions -- like
                # as_numpy generators -- are served through the host path below
                self._host_path = True
            datasets.append(SingleSessionDatasetBatchedLoad(
                data_dir, lab=ids['lab'], expt=ids['expt'], animal=ids['animal'],
                session=ids['session'], signals=signals, transforms=transforms, paths=paths,
                device=device, as_numpy=False, keep_in_memory=keep_in_memory))
            self.datasets_info.append({k: ids[k] for k in ('lab', 'expt', 'animal', 'session')})
        if trial_splits is None:
            trial_splits = {'train_tr': 8, 'val_tr': 1, 'test_tr': 1, 'gap_tr': 0}
        for ds in datasets:
            ds.batch_idxs = split_trials(len(ds), rng_seed=rng_seed, **trial_splits)
            n_train = len(ds.batch_idxs['train'])
            if train_frac != 1.0:
                # subsample the training trials (ref :520-534; numpy global RNG as seeded by
                # split_trials)
                if train_frac < 1.0:
                    n_idxs = int(np.floor(train_frac * n_train))
                    if n_idxs <= 0:
                        print('warning: attempting to use invalid number of training ' +
                              'batches; defaulting to all training batches')
                        n_idxs = n_train
                else:
                    n_idxs = int(min(train_frac, n_train))
                keep = np.random.choice(n_train, size=n_idxs, replace=False)
                ds.batch_idxs['train'] = ds.batch_idxs['train'][keep]
            ds.n_batches = {k: len(v) for k, v in ds.batch_idxs.items()}
        super().__init__(datasets, device=device, placement=placement,
                         n_sessions_per_batch=n_sessions_per_batch)

    def _load_session(self, ds):
        device = self.device

        def images(u8):
            if u8.dtype != np.uint8:
                raise ValueError('images must be stored as uint8 (got %s)' % u8.dtype)
            if self.placement == 'host_u8':
                t = torch.from_numpy(np.ascontiguousarray(u8))
                return t.pin_memory() if device == 'cuda' else t
            if self.placement == 'device_u8':
                return torch.from_numpy(np.ascontiguousarray(u8)).to(device)
            t = torch.from_numpy(u8.astype(np.float32) / 255)
            if self.placement == 'host':
                return t.pin_memory() if device == 'cuda' else t
            return t.to(device)

        def floats(signal):
            tr = ds.transforms[signal]

            def conv(a):
                a = a.astype(np.float32)
                if signal == 'masks':
                    # as the reference serves them (data_generator.py:264-277,337-340: the array
                    # is not wrapped in a list before ``sample[signal][0]``): the mask of the
                    # trial's FIRST frame, (C, H, W); the losses broadcast it over the frames
                    a = a[0]
                if tr:
                    a = tr(a)
                return torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
            return conv
        self._extra = getattr(self, '_extra', [])
        extra = {sig: _LazyTrials(ds, sig, floats(sig)) for sig in ds.signals
                 if sig not in ('images', 'labels')}
        self._extra.append(extra)
        labels = _LazyTrials(ds, 'labels', floats('labels')) if 'labels' in ds.signals else None
        direct = None
        if self.placement == 'host_u8' and device == 'cuda':
            store = ds._store('images')

            def direct(trial):
                # file -> pinned tensor in one read (the caching host allocator recycles the block
                # once the tensor is dropped and the copies queued from it have completed)
                lay = store.layout('images', trial) if hasattr(store, 'layout') else None
                if lay is None:
                    return None
                if lay[0] != np.uint8:
                    raise ValueError('images must be stored as uint8 (got %s)' % lay[0])
                t = torch.empty(lay[1], dtype=torch.uint8, pin_memory=True)
                store.read_into('images', trial, t.numpy())
                return t
        return _LazyTrials(ds, 'images', images, direct), labels

    def _extra_signals(self, sample, sess, trial):
        for signal, trials in self._extra[sess].items():
            sample[signal] = trials[trial][None]

    def _sample(self, sess, trial, dtype):
        if not self._host_path:
            return super()._sample(sess, trial, dtype)
        host = self.datasets[sess][int(trial)]          # signals as float32 host tensors
        sample = OrderedDict()
        for signal, value in host.items():
            if signal == 'batch_idx':
                continue
            if self.as_numpy:
                # the reference's form: a list over the loader's batch dimension of 1
                sample[signal] = [value.cpu().detach().numpy() if torch.is_tensor(value)
                                  else np.asarray(value)]
            else:
                sample[signal] = value[None].to(self.device)
        sample['batch_idx'] = torch.tensor([trial])
        return sample

    def __str__(self):
        out = 'Generator contains %i SingleSessionDatasetBatchedLoad objects:\n' % self.n_datasets
        for ds in self.datasets:
            out += ds.__str__()
        return out


class SingleSessionDataset(SingleSessionDatasetBatchedLoad):
    """The reference's in-memory variant (ref data_generator.py:345-429: every trial read at construction).  Here
    the trials of a session are read on first use and then kept (``keep_in_memory``), which serves the same
    purpose without the start-up pass; the name exists so that code written against the reference finds it."""

    def __init__(self, data_dir, lab='', expt='', animal='', session='', signals=None, transforms=None,
                 paths=None, device='cuda', as_numpy=False):
        super().__init__(data_dir, lab=lab, expt=expt, animal=animal, session=session, signals=signals,
                         transforms=transforms, paths=paths, device=device, as_numpy=as_numpy, keep_in_memory=True)


class ConcatSessionsGeneratorMulti(ConcatSessionsGenerator):
    """Several sessions per training batch (ref data_generator.py:636-790, the MSPS-VAE's generator): the
    ``n_sessions_per_batch`` form of :class:`ConcatSessionsGenerator` under the reference's name and default."""

    def __init__(self, data_dir, ids_list, signals_list=None, transforms_list=None, paths_list=None,
                 device='cuda', as_numpy=False, batch_load=True, rng_seed=0, trial_splits=None, train_frac=1.0,
                 n_sessions_per_batch=2, **kwargs):
        super().__init__(data_dir, ids_list, signals_list=signals_list, transforms_list=transforms_list,
                         paths_list=paths_list, device=device, as_numpy=as_numpy, batch_load=batch_load,
                         rng_seed=rng_seed, trial_splits=trial_splits, train_frac=train_frac,
                         n_sessions_per_batch=n_sessions_per_batch, **kwargs)
