"""Training loop for the autoencoder hot path.

Host-side mirror of the reference ``behavenet/fitting/training.py``: ``fit(hparams, model,
data_generator, exp, method='ae')`` with the same duck-typed generator / experiment protocol,
the same metric rows, early stopping and checkpoint files.  Differences, all below the
reference's Python surface:

* the optimizer is :class:`behavenet_amd.fitting.optim.FlatAdamAMSGrad` (one HIP launch);
* ``model.loss`` synchronises with the host once per call instead of once per chunk;
* with ``torch.distributed`` initialised the fit is data parallel (fitting/distributed.py),
  ``hparams['dp_shard']``:
    'frames' -- every rank sees the same trial and runs its slice of every 200-frame chunk; the
                summed gradients, the metric rows and the fitted weights are the single-device
                ones (up to fp32 summation order);
    'trial'  -- (default) the training trials of an epoch are dealt round-robin to the ranks, W
                at a time: one optimizer step per W trials on the AVERAGE of their gradients
                (not step-for-step the reference, which steps once per trial); validation and
                test trials are evaluated by every rank.
"""

import copy
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from behavenet_amd.fitting import distributed as bdist

__all__ = ['Logger', 'EarlyStopping', 'fit']

_PREFIX = {'train': 'tr', 'val': 'val', 'test': 'test'}


class Logger(object):
    """Accumulates every key of the loss dicts, in aggregate and per dataset (ref :16-170)."""

    def __init__(self, n_datasets=1):
        self.n_datasets = n_datasets
        kinds = ['train', 'val', 'test', 'curr']
        self.metrics = {k: {} for k in kinds}
        self.metrics_by_dataset = []
        if n_datasets > 1:
            self.metrics_by_dataset = [{k: {} for k in kinds} for _ in range(n_datasets)]

    def flush(self):
        """Add up the loss dicts that were handed over unresolved (graph_step.LazyLoss)."""
        pending, self._pending = getattr(self, '_pending', []), []
        for dtype, lazy, dataset in pending:
            self._add(dtype, lazy.resolve(), dataset)

    def reset_metrics(self, dtype):
        self.flush()
        for key in self.metrics[dtype]:
            self.metrics[dtype][key] = 0
        for per in self.metrics_by_dataset:
            for key in per[dtype]:
                per[dtype][key] = 0

    def update_metrics(self, dtype, loss_dict, dataset=None):
        if hasattr(loss_dict, 'resolve'):
            # values still on their way from the device: added up, in order, when something
            # reads the metrics (or a few steps later) -- the host does not wait per step
            self._pending = getattr(self, '_pending', []) + [(dtype, loss_dict, dataset)]
            if len(self._pending) > 4:
                head, self._pending = self._pending[0], self._pending[1:]
                self._add(head[0], head[1].resolve(), head[2])
            return
        self.flush()
        self._add(dtype, loss_dict, dataset)

    def _add(self, dtype, loss_dict, dataset):
        for key, val in {**loss_dict, 'batches': 1}.items():
            self.metrics[dtype][key] = self.metrics[dtype].get(key, 0) + val
            if isinstance(dataset, int) and self.n_datasets > 1:
                per = self.metrics_by_dataset[dataset][dtype]
                per[key] = per.get(key, 0) + val

    def create_metric_row(
            self, dtype, epoch, batch, dataset, trial, best_epoch=None, by_dataset=False):
        if dtype not in _PREFIX:
            raise ValueError("%s is an invalid data type" % dtype)
        self.flush()
        prefix = _PREFIX[dtype]
        row = {'epoch': epoch, 'batch': batch, 'trial': trial}
        if dtype == 'val':
            row['best_val_epoch'] = best_epoch
        if by_dataset and self.n_datasets > 1:
            source = self.metrics_by_dataset[dataset][dtype]
        else:
            dataset = -1
            source = self.metrics[dtype]
        norm = source['batches']
        for key, val in source.items():
            if key != 'batches':
                row['%s_%s' % (prefix, key)] = val / norm
        row['dataset'] = dataset
        return row

    def get_loss(self, dtype):
        self.flush()
        return self.metrics[dtype]['loss'] / self.metrics[dtype]['batches']


class EarlyStopping(object):
    """Patience-based stopping rule behind ``hparams['enable_early_stop']`` (behaviour of reference
    training.py:173-241, re-stated): a validation check that does not undercut the best loss seen
    so far by more than ``delta`` is a strike; ``patience`` strikes in a row -- counted only once
    ``min_epochs`` epochs have passed -- raise ``should_stop``, which ``fit`` polls after every
    epoch.  The attributes ``fit`` and the reference's callers read keep their names
    (``best_loss``, ``best_epoch``, ``counter``, ``stopped_epoch``, ``should_stop``)."""

    _REPORT = ('\n== early stopping criteria met; exiting train loop ==\n'
               'training epochs: %d\nend cost: %04f\nbest epoch: %i\nbest cost: %04f\n')

    def __init__(self, patience=10, min_epochs=10, delta=0):
        self.patience, self.min_epochs, self.delta = patience, min_epochs, delta
        self.best_loss, self.best_epoch = np.inf, 0
        self.counter = 0                 # strikes since the last improvement
        self.stopped_epoch = 0
        self.should_stop = False

    def on_val_check(self, epoch, curr_loss):
        improved = curr_loss < self.best_loss - self.delta
        self.counter = 0 if improved else self.counter + 1
        if improved:
            self.best_loss, self.best_epoch = curr_loss, epoch
        if self.counter >= self.patience and epoch > self.min_epochs:
            print(self._REPORT % (epoch, curr_loss, self.best_epoch, self.best_loss))
            self.stopped_epoch, self.should_stop = epoch, True


def _snapshot(model, hparams, into=None):
    """The reference's ``copy.deepcopy`` of the model with ``hparams`` detached (training.py:393-396).

    ``into``: the previous snapshot.  While it still has the model's tensors (same state-dict keys,
    shapes, dtypes) it is REFRESHED -- one multi-tensor device copy, queued like any kernel, nothing
    allocated, no module tree walked -- instead of being rebuilt: a fresh deepcopy costs a few
    milliseconds of host time at a moment when the device has just run dry (the decision to take a
    snapshot needs the validation loss on the host)."""
    if into is not None:
        src, dst = model.state_dict(), into.state_dict()
        if list(src.keys()) == list(dst.keys()) and all(
                a.shape == b.shape and a.dtype == b.dtype and a.device == b.device
                for a, b in zip(src.values(), dst.values())):
            with torch.no_grad():
                torch._foreach_copy_([t for t in dst.values()], [t.detach() for t in src.values()])
            for name in ('curr_epoch', 'version', 'training'):
                if hasattr(model, name):
                    try:
                        setattr(into, name, getattr(model, name))
                    except AttributeError:
                        pass
            if into.training != model.training:
                into.train(model.training)
            into.hparams = hparams
            return into
    model.hparams = None
    snap = copy.deepcopy(model)
    model.hparams = hparams
    snap.hparams = hparams
    return snap


class _CheckpointWriter(object):
    """``model.save(path)`` without stalling the device: the state dict is copied device-side on
    the compute stream (microseconds), brought to pinned host memory on a side stream, and pickled
    to ``path`` by a background thread (written next to it, then renamed: a reader never sees a
    half-written file).  The reference writes the same file synchronously
    (training.py:388-397); nothing says the device has to wait for the disk.  ``wait()`` before
    anybody reads the file; one write at a time per writer."""

    def __init__(self):
        self._thread = None
        self._error = None
        self._stream = None
        self._plan = None           # (dtype, ((key, shape), ...)) groups the flat buffers were laid out for
        self._flat = {}             # dtype -> (flat device buffer, flat pinned host buffer)

    def wait(self, reraise=True):
        """Join the write in flight; its failure (disk full, pickling error) is raised here -- or, with
        ``reraise=False`` (fit() is already on its way out with another exception), only dropped: the
        thread has reported it on stderr when it happened."""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            err, self._error = self._error, None
            if reraise:
                raise err

    def save(self, model, path):
        from behavenet_amd.models.base import BaseModel
        state = model.state_dict()
        plain = getattr(type(model), 'save', None) is BaseModel.save
        if not plain or not any(t.is_cuda for t in state.values()):
            # classes whose ``save`` does more than write the state dict (AEMSP), host models
            self.wait()
            model.save(path)
            return
        self.wait()
        import threading
        dev = next(t.device for t in state.values() if t.is_cuda)
        main = torch.cuda.current_stream(dev)
        # ONE flat device buffer and ONE flat pinned host buffer per dtype, kept across saves: the state
        # dict is packed device-side by a multi-tensor copy in stream order (the snapshot), crosses
        # PCIe as one transfer per dtype on the side stream, and is cut back into tensors by the
        # writer thread.  (Round 5, first form: a clone and a pinned allocation PER TENSOR -- 7 ms of
        # host time per checkpoint with the device idle, tools/gaps.py on the rocprofv3 trace of fit().)
        names = list(state.keys())
        groups = {}
        for k in names:
            v = state[k]
            if v.is_cuda:
                groups.setdefault(v.dtype, []).append(k)
        plan = tuple((dt, tuple((k, tuple(state[k].shape)) for k in ks)) for dt, ks in groups.items())
        if self._plan != plan:
            self._plan = plan
            self._flat = {}
            for dt, ks in groups.items():
                n = sum(state[k].numel() for k in ks)
                self._flat[dt] = (torch.empty(n, dtype=dt, device=dev),
                                  torch.empty(n, dtype=dt, pin_memory=True))
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        layout = {}
        with torch.no_grad():
            for dt, ks in groups.items():
                flat_d, _ = self._flat[dt]
                views, srcs, pos = [], [], 0
                for k in ks:
                    v = state[k].detach()
                    views.append(flat_d[pos:pos + v.numel()].view(v.shape))
                    srcs.append(v)
                    layout[k] = (dt, pos, v.numel(), tuple(v.shape))
                    pos += v.numel()
                torch._foreach_copy_(views, srcs)
        ev = torch.cuda.Event()
        ev.record(main)
        self._stream.wait_event(ev)
        with torch.cuda.stream(self._stream):
            for dt in groups:
                flat_d, flat_h = self._flat[dt]
                flat_h.copy_(flat_d, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._stream)
        # (the next save's packing copy cannot overtake this transfer: save() starts with wait(), and the
        # writer thread has waited for `done` by then)
        on_host = {k: state[k].detach().clone() for k in names if not state[k].is_cuda}
        metadata = getattr(state, '_metadata', None)
        flats = {dt: pair[1] for dt, pair in self._flat.items()}

        def write():
            try:
                done.synchronize()
                # the structure BaseModel.save writes: an OrderedDict that carries the modules' version
                # metadata (the file must not depend on which of the two paths wrote it)
                out = OrderedDict()
                if metadata is not None:
                    out._metadata = metadata
                for k in names:
                    if k in layout:
                        dt, pos, n, shape = layout[k]
                        out[k] = flats[dt][pos:pos + n].view(shape).clone()     # plain, unpinned
                    else:
                        out[k] = on_host[k]
                tmp = '%s.tmp.%d' % (path, os.getpid())
                torch.save(out, tmp)
                os.replace(tmp, path)
            except BaseException as err:            # noqa: BLE001 (re-raised by wait())
                self._error = err
                # said at once: if fit() fails before its next save() / wait(), nobody would hear of it
                print('checkpoint writer: %s was NOT written (%s: %s)' % (path, type(err).__name__, err),
                      file=sys.stderr, flush=True)
        self._thread = threading.Thread(target=write, name='bn-checkpoint', daemon=False)
        self._thread.start()


def _save_checkpoint(model, path, is_main, writer=None):
    """``model.save`` on the main rank.  Classes whose ``save`` also finalises derived state
    (AEMSP builds its orthogonal matrix U there, ref aes.py:1062-1065) do that on EVERY rank, so
    that the snapshots the other ranks keep (and export latents through) carry it too."""
    if is_main:
        if writer is not None:
            writer.save(model, path)
        else:
            model.save(path)
    elif hasattr(model, 'create_orthogonal_matrix'):
        model.create_orthogonal_matrix()


def _merge_rank_metrics(logger, dtype):
    """'trial' mode: every rank logged the trials it trained on; give every rank the totals."""
    import torch.distributed as dist
    logger.flush()
    mine = [logger.metrics[dtype]] + [per[dtype] for per in logger.metrics_by_dataset]
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    for slot, target in enumerate(mine):
        keys = []
        for other in everyone:
            for k in other[slot]:
                if k not in keys:
                    keys.append(k)
        for k in keys:
            target[k] = sum(other[slot].get(k, 0) for other in everyone)


def _progress(iterable, enabled):
    if not enabled:
        return iterable
    try:
        from tqdm import tqdm
        return tqdm(iterable)
    except ImportError:
        return iterable


def _log_rows(exp, logger, dtype, i_epoch, i_batch, dataset, n_datasets, best_epoch):
    """The metric rows of one check (ref training.py:358-368 train, :400-417 val): the row pooled over
    sessions (dataset -1), then -- when several sessions are fit and batches come from one session at a
    time -- one row per session; ``dataset`` is what the last ``next_batch`` returned."""
    single = dataset is not None and (isinstance(dataset, int) or len(dataset) == 1)
    targets = [(-1, False)]
    if n_datasets > 1 and single:
        targets.extend((d, True) for d in range(n_datasets))
    for d, per_session in targets:
        exp.log(logger.create_metric_row(dtype, i_epoch, i_batch, d, trial=-1, by_dataset=per_session,
                                         best_epoch=best_epoch))
    exp.save()


def fit(hparams, model, data_generator, exp, method='ae', optimizer=None):
    """Fit a model with Adam(amsgrad) SGD and early stopping (ref training.py:244-461).

    ``optimizer`` (optional) injects an object with ``zero_grad()``/``step()``; by default a
    :class:`FlatAdamAMSGrad` over ``model.get_parameters()`` is built.

    The loss dicts of the eager steps are handed to the logger unresolved (hip_functions.set_lazy_losses;
    ``hparams['lazy_losses'] = False`` turns that off): same values in the same order in the metric rows,
    but the host does not wait for the forward pass of a step before it queues the next one.
    """
    from behavenet_amd import hip_functions as hf
    prev = hf.set_lazy_losses(bool(hparams.get('lazy_losses', True)))
    # checkpoints leave through a background writer unless hparams['async_checkpoint'] is False
    writer = _CheckpointWriter() if hparams.get('async_checkpoint', True) else None
    try:
        out = _fit(hparams, model, data_generator, exp, method=method, optimizer=optimizer, writer=writer)
    except BaseException:
        if writer is not None:
            writer.wait(reraise=False)          # a write in flight still finishes: the best model so far stays valid
        raise
    else:
        if writer is not None:
            writer.wait()                       # the checkpoint files are complete before fit() returns
        return out
    finally:
        hf.set_lazy_losses(prev)


def _fit(hparams, model, data_generator, exp, method='ae', optimizer=None, writer=None):
    if hparams.get('dp_shard') is not None:
        bdist.set_shard_mode(hparams['dp_shard'])
    # hparams['shard_optimizer'] (BN_SHARD_OPTIMIZER=0/1): reduce-scatter -> Adam on this rank's 1/R
    # shard of the arena -> all-gather, instead of all-reduce + R identical steps
    # (fitting/distributed.py sharded_step, default_shard_optimizer: on for frame sharding over >= 4 ranks)
    shard_opt = hparams.get('shard_optimizer')
    if shard_opt is None:
        shard_opt = bdist.default_shard_optimizer()
    shard_opt = bool(shard_opt) and bdist.is_active() and bdist.world_size() > 1
    if optimizer is None:
        from behavenet_amd.fitting.optim import FlatAdamAMSGrad
        optimizer = FlatAdamAMSGrad(
            model.get_parameters(), lr=hparams['learning_rate'],
            weight_decay=hparams.get('l2_reg', 0),
            shard_over=bdist.world_size() if shard_opt else 1)
    shard_opt = shard_opt and getattr(optimizer, 'shard_over', 1) == bdist.world_size()
    flat_g = getattr(optimizer, 'flat_g', None)
    world = bdist.world_size() if bdist.is_active() else 1
    rank = bdist.rank()
    trial_mode = world > 1 and bdist.shard_mode() == 'trial'
    if bdist.is_active() and getattr(optimizer, 'flat_p', None) is not None:
        bdist.broadcast_parameters_(optimizer.flat_p)
        if not shard_opt:
            bdist.attach_reducer(optimizer)

    reducer = getattr(optimizer, 'reducer', None)
    overlap_pref = reducer.overlap if reducer is not None else False
    can_skip = False
    if trial_mode:
        import inspect
        try:
            can_skip = 'skip' in inspect.signature(data_generator.next_batch).parameters
        except (TypeError, ValueError):
            can_skip = False
        if can_skip and hasattr(data_generator, 'lookahead'):
            data_generator.lookahead = world - 1

    # the step as a HIP graph (fitting/graph_step.py): per input signature, after two eager steps;
    # opt-in with hparams['hip_graph'] / BN_GRAPH=1 (the step is GPU-bound on an idle host).
    # Needs the flat gradient arena (the recorded kernels write into fixed addresses).
    loss_fn = model.loss
    if flat_g is not None and flat_g.is_cuda:
        from behavenet_amd.fitting import graph_step
        if hparams.get('hip_graph', graph_step.enabled_by_default()):
            loss_fn = graph_step.GraphedLoss(model)

    logger = Logger(n_datasets=data_generator.n_datasets)
    early_stop = None
    if hparams['enable_early_stop']:
        early_stop = EarlyStopping(
            patience=hparams['early_stop_history'], min_epochs=hparams['min_n_epochs'])

    n_trials_train = data_generator.n_tot_batches['train']
    # optimizer steps per epoch: in 'trial' mode W trials are consumed per step
    n_train = -(-n_trials_train // world) if trial_mode else n_trials_train
    max_epochs = hparams['max_n_epochs']
    interval = hparams['val_check_interval']
    val_check_batch = np.append(
        interval * n_train * np.arange(1, int((max_epochs + 1) / interval)),
        [n_train * max_epochs, n_train * (max_epochs + 1)]).astype('int')

    best_val_loss = np.inf
    best_val_epoch = None
    best_val_model = None
    best_model_saved = False

    if hparams.get('rng_seed_train', None) is None:
        rng_train = np.random.randint(0, 10000)
        if world > 1:       # every rank must walk the same trial order: rank 0's draw
            rng_train = int(bdist.broadcast_object(rng_train))
    else:
        rng_train = int(hparams['rng_seed_train'])
    torch.manual_seed(rng_train)
    np.random.seed(rng_train)

    expt_dir = os.path.join(hparams['expt_dir'], 'version_%i' % exp.version)
    is_main = bdist.rank() == 0
    show_bar = hparams.get('progress_bar', True) and is_main

    i_epoch = 0
    for i_epoch in range(max_epochs + 1):
        # epoch 0 evaluates the randomly initialised model: forward/backward, no step (:320-322)
        if is_main:
            print_epoch(i_epoch, max_epochs)
        torch.manual_seed(rng_train + i_epoch)
        np.random.seed(rng_train + i_epoch)
        logger.reset_metrics('train')
        data_generator.reset_iterators('train')
        model.curr_epoch = i_epoch
        if reducer is not None:
            # Epoch 0 takes no optimizer step, so nothing is reduced in it -- and nothing may be
            # LAUNCHED either: the overlapped reducer sends a bucket as soon as the backward pass
            # has completed it, a rank without a trial in the last (short) group of the epoch
            # runs no backward pass, and the ranks' collective sequences would then differ
            # (RCCL: hang or mixed-up payloads).  From epoch 1 on every step ends in
            # `reduce_gradients`, which launches whatever a rank has not sent yet in bucket
            # order, so ranks with and without a trial issue the same sequence.
            reducer.overlap = overlap_pref and i_epoch > 0

        for i_train in _progress(range(n_train), show_bar):
            model.train()
            optimizer.zero_grad()
            if trial_mode:
                # every rank walks the same W trials (same seeds, same generator state) and
                # keeps the one at its own position: disjoint trials, nothing communicated; the
                # other ranks' trials only advance the generator (no read / copy / conversion)
                n_group = min(world, n_trials_train - i_train * world)
                if can_skip:
                    group = [data_generator.next_batch('train', skip=(j != rank))
                             for j in range(n_group)]
                else:
                    group = [data_generator.next_batch('train') for _ in range(n_group)]
                data, dataset = group[rank] if rank < len(group) else (None, None)
                stepping = any(d is not None for d, _ in group)
                n_in_step = sum(d is not None for d, _ in group)
            else:
                data, dataset = data_generator.next_batch('train')
                stepping = data is not None
                n_in_step = 1
            if data is not None:
                loss_dict = loss_fn(data, dataset=dataset, accumulate_grad=True)
                logger.update_metrics('train', loss_dict, dataset=dataset)
            if stepping and i_epoch > 0:
                if flat_g is not None and shard_opt:
                    bdist.sharded_step(optimizer, divide_by=n_in_step if trial_mode else None)
                else:
                    if flat_g is not None:
                        if trial_mode:
                            # mean over the trials of this step (a rank without one adds zeros)
                            bdist.reduce_gradients(optimizer)
                            optimizer.flat_g.div_(float(n_in_step))
                        else:
                            bdist.reduce_gradients(optimizer)
                    optimizer.step()

            if (i_train + 1) % n_train == 0:
                if trial_mode:
                    _merge_rank_metrics(logger, 'train')
                _log_rows(exp, logger, 'train', i_epoch, i_train, dataset, data_generator.n_datasets,
                          best_val_epoch)

            curr_batch = (i_train + 1) + i_epoch * n_train
            if np.any(curr_batch == val_check_batch):
                logger.reset_metrics('val')
                data_generator.reset_iterators('val')
                model.eval()
                for _ in range(data_generator.n_tot_batches['val']):
                    data, dataset = data_generator.next_batch('val')
                    loss_dict = loss_fn(data, dataset=dataset, accumulate_grad=False)
                    logger.update_metrics('val', loss_dict, dataset=dataset)

                if logger.get_loss('val') < best_val_loss:
                    best_val_loss = logger.get_loss('val')
                    _save_checkpoint(model, os.path.join(expt_dir, 'best_val_model.pt'), is_main,
                                     writer)
                    best_model_saved = True
                    best_val_model = _snapshot(model, hparams, into=best_val_model)
                    best_val_epoch = i_epoch

                _log_rows(exp, logger, 'val', i_epoch, i_train, dataset, data_generator.n_datasets,
                          best_val_epoch)

        if hparams['enable_early_stop']:
            early_stop.on_val_check(i_epoch, logger.get_loss('val'))
            if early_stop.should_stop:
                break

    if not best_model_saved:
        _save_checkpoint(model, os.path.join(expt_dir, 'best_val_model.pt'), is_main, writer)
        best_val_model = _snapshot(model, hparams)

    if hparams.get('save_last_model', False):
        _save_checkpoint(model, os.path.join(expt_dir, 'last_model.pt'), is_main, writer)

    # test loss, one row per test trial.  NB the reference evaluates `model`, not
    # `best_val_model`, here (training.py:433,442) -- kept.
    # (one row per test trial: the logger is cleared in front of every trial)
    data_generator.reset_iterators('test')
    best_val_model.eval()
    n_test = data_generator.n_tot_batches['test']
    for i_test, (data, dataset) in enumerate(data_generator.next_batch('test') for _ in range(n_test)):
        logger.reset_metrics('test')
        logger.update_metrics('test', model.loss(data, dataset=dataset, accumulate_grad=False), dataset=dataset)
        idx = data['batch_idx']
        exp.log(logger.create_metric_row('test', i_epoch, i_test, dataset, by_dataset=True,
                                         trial=idx.item() if hasattr(idx, 'item') else int(idx)))
    exp.save()
    if writer is not None:
        writer.wait()           # the checkpoint files are complete before fit() returns

    if method == 'ae' and hparams['export_latents']:
        if is_main:
            print('exporting latents')
        from behavenet_amd.fitting.eval import export_latents
        export_latents(data_generator, best_val_model)
    elif method == 'nll' and hparams.get('export_predictions', False):
        raise NotImplementedError('neural decoders are outside the MI355X hot path')
    return best_val_model


def print_epoch(curr, total):
    """Zero-padded epoch counter."""
    width = 1 if total < 10 else min(len(str(total)), 5) if total < 100000 else 0
    if width:
        print('epoch %0*i/%0*i' % (width, curr, width, total))
    else:
        print('epoch %i/%i' % (curr, total))
