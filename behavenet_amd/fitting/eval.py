"""Inference-only encode of every trial -> ``*_latents.pkl`` (BASELINE config 5).

Mirror of ``export_latents`` in the reference ``behavenet/fitting/eval.py:6-118``: same pickle
schema ``{'latents': [per-trial (T x D) arrays, empty for gap trials], 'trials': batch_idxs}``
and file name ``<lab>_<expt>_<animal>_<session>_latents.pkl`` so the ARHMM stage can read it.
Differences below the surface: runs under ``torch.no_grad()`` (the reference builds an unused
graph), accepts device-resident ``uint8`` frames (converted by ``bn_u8_to_unit_float``), and
under ``torch.distributed`` shards trials round-robin over ranks and gathers on rank 0.
"""

import os
import pickle

import numpy as np
import torch

from behavenet_amd import _hip
from behavenet_amd.fitting import distributed as bdist

__all__ = ['export_latents', 'encode_trial', 'encode_trial_device', 'get_reconstruction']


def encode_trial(model, y, sess=None, labels_2d=None, chunk_size=200):
    """Latents (T x D numpy) of one trial, encoded in 200-frame chunks (ref eval.py:51-97)."""
    return encode_trial_device(model, y, sess, labels_2d, chunk_size).cpu().numpy()


def encode_trial_device(model, y, sess=None, labels_2d=None, chunk_size=200):
    """The same latents as a DEVICE tensor: nothing waits for the host (``export_latents`` keeps
    the trials' latents on the device and fetches them once at the end)."""
    mc = model.hparams['model_class']
    if y.dtype == torch.uint8 and (labels_2d is not None or
                                   model.hparams.get('model_type', 'conv') != 'conv'):
        # (conv encoders take the stored uint8 frames as they are: value / 255 is fused into the
        # first layer, csrc k_down_c1s<.., U8>; extra input channels need the float tensor)
        y = _hip.u8_to_unit_float(y.contiguous())
    n = y.shape[0]
    parts = []
    with torch.no_grad():
        for beg in range(0, n, chunk_size):
            end = min(beg + chunk_size, n)
            y_in = y[beg:end]
            if labels_2d is not None:
                y_in = torch.cat((y_in, labels_2d[beg:end]), dim=1)
            out = model.encoding(y_in, dataset=sess)
            if mc == 'ps-vae':
                cur = torch.cat([out[0], out[1]], dim=1)
            elif mc == 'msps-vae':
                cur = torch.cat([out[0], out[1], out[2]], dim=1)
            else:
                cur = out[0]
            if mc == 'cond-ae-msp':
                cur = model.U(cur)
            parts.append(cur)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)


class _GraphedTrialEncoder(object):
    """``encode_trial_device`` as a HIP graph per trial shape: a trial's encode is ~25 launches of
    20-200 us issued through autograd-free Python and ctypes in 0.6-0.9 ms of host time -- longer than
    the device needs for them (0.69 ms for 256 frames of 128x128) once the trials come from a file-backed
    generator that also keeps the main thread busy.  Recorded after the first trial of a shape (same
    kernels on the same operands: bit-identical latents), replayed with the trial copied into the
    graph's static input; trials with extra input channels, host tensors and shapes beyond
    ``max_graphs`` run eagerly.  Opt-in: ``hparams['hip_graph_encode'] = True`` / ``BN_GRAPH_ENCODE=1``."""

    def __init__(self, model, warmup=1, max_graphs=6):
        self.model, self.warmup, self.max_graphs = model, int(warmup), int(max_graphs)
        self._graphs, self._seen, self._refused, self._pool = {}, {}, set(), None
        self.n_replays = self.n_eager = 0
        # opt-in: measured (tools/probe_export.py, one MI355X, idle host) 0.72 ms per 256-frame trial replayed
        # against 0.70 ms launched eagerly -- the encoder is bound by the device; for hosts that are not idle
        self.enabled = bool(model.hparams.get('hip_graph_encode', os.environ.get('BN_GRAPH_ENCODE', '0') == '1'))
        # The reference encodes a trial in chunks of 200 frames (eval.py:51-97): that bounds ITS memory
        # use, nothing else -- in eval mode frames are independent through every encoder (batch norm
        # normalises with the running estimates).  A 256-frame trial as 200 + 56 frames is two passes,
        # the second on a quarter-filled chip: 0.91 ms against 0.69 ms for one pass (tools/probe_export.py).
        # Device-resident trials are therefore encoded whole, up to ``export_chunk_frames`` frames a pass.
        self.chunk = int(model.hparams.get('export_chunk_frames', 1024))

    def __call__(self, y, sess=None, labels_2d=None):
        if not self.enabled or labels_2d is not None or not y.is_cuda or not y.is_contiguous():
            self.n_eager += 1
            return encode_trial_device(self.model, y, sess, labels_2d, self.chunk)
        key = (tuple(y.shape), y.dtype, repr(sess))
        rec = self._graphs.get(key)
        if rec is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n <= self.warmup or key in self._refused or len(self._graphs) >= self.max_graphs:
                self.n_eager += 1
                return encode_trial_device(self.model, y, sess, labels_2d, self.chunk)
            rec = self._record(key, y, sess)
            if rec is None:
                self.n_eager += 1
                return encode_trial_device(self.model, y, sess, labels_2d, self.chunk)
        static_in, graph, out = rec
        static_in.copy_(y, non_blocking=True)
        graph.replay()
        self.n_replays += 1
        return out.clone()

    def _record(self, key, y, sess):
        import warnings
        static_in = torch.empty_like(y)
        static_in.copy_(y)
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        try:
            # thread_local: the generator's reader threads and copy stream keep working meanwhile
            with torch.cuda.graph(graph, pool=self._pool, capture_error_mode='thread_local'):
                out = encode_trial_device(self.model, static_in, sess, None, self.chunk)
        except Exception as err:                        # noqa: BLE001 (reported, then eager)
            torch.cuda.synchronize()
            self._refused.add(key)
            warnings.warn('HIP graph capture of the trial encoder failed (%s: %s); trials of shape %s '
                          'stay on eager launches' % (type(err).__name__, err, key[0]))
            return None
        rec = (static_in, graph, out)
        self._graphs[key] = rec
        return rec


def export_latents(data_generator, model, filename=None):
    """Encode train/val/test trials of every session and pickle them (ref eval.py:6-118)."""
    # multi-session generators serve lists of batches for training; latents are exported trial by
    # trial (the reference's MSPSVAE.export_latents rebuilds a one-session-per-batch generator,
    # vaes.py:1198-1216)
    single = {'return_multiple': False} \
        if getattr(data_generator, 'n_sessions_per_batch', 1) > 1 else {}
    model.eval()
    rank, world = bdist.rank(), bdist.world_size()

    latents = [[np.array([]) for _ in range(ds.n_trials)] for ds in data_generator.datasets]
    cond_enc = model.hparams['model_class'] == 'cond-ae' and \
        model.hparams.get('conditional_encoder', False)
    # which rank encodes a trial is a function of the trial's identity (its position in the
    # sorted list of all (session, trial) pairs), not of the order in which the generator
    # happens to serve it on this rank
    wanted = sorted((s_, int(t)) for s_, ds in enumerate(data_generator.datasets)
                    for dt in ('train', 'val', 'test') for t in ds.batch_idxs[dt])
    owner = {key: i % world for i, key in enumerate(wanted)}
    # generators that can be told to pass over a trial (this package's) are asked not to read,
    # copy or convert the other ranks' trials -- round 4 fetched every trial on every rank -- and to
    # hand out stored uint8 frames as they are (the first conv layer converts in flight)
    import inspect
    try:
        can_skip = 'skip' in inspect.signature(data_generator.next_batch).parameters
    except (TypeError, ValueError):
        can_skip = False
    skip = {'skip': (lambda s_, t_: owner.get((s_, int(t_)), rank) != rank)} \
        if (can_skip and world > 1) else {}
    conv_u8 = model.hparams.get('model_type', 'conv') == 'conv' and not cond_enc
    serve_prev = getattr(data_generator, 'serve_uint8', None)
    if serve_prev is not None and conv_u8:
        data_generator.serve_uint8 = True
    on_device = {}          # (session, trial) -> device tensor: ONE transfer to the host at the end
    encode = _GraphedTrialEncoder(model)
    try:
        for dtype in ['train', 'val', 'test']:
            data_generator.reset_iterators(dtype)
            n_batches = data_generator.n_tot_batches[dtype]
            if single and dtype == 'train':      # the multi generator counts training ITERATIONS
                n_batches = sum(ds.n_batches['train'] for ds in data_generator.datasets)
            for _ in range(n_batches):
                data, sess = data_generator.next_batch(dtype, **single, **skip)
                if data is None:
                    # the generator ran dry before its own count: latents of the remaining trials would be
                    # exported EMPTY without a word
                    raise RuntimeError('export_latents: the generator ended after %d of %d %s batches'
                                       % (_, n_batches, dtype))
                if not isinstance(data, dict):       # SKIPPED: another rank's trial
                    continue
                idx = data['batch_idx']
                idx = idx.item() if hasattr(idx, 'item') else int(idx)
                if owner[(sess, idx)] != rank:
                    continue
                labels_2d = data['labels_sc'][0] if cond_enc else None
                images = data['images'][0]
                if not torch.is_tensor(images):      # generators serving numpy arrays (as_numpy)
                    images = torch.from_numpy(np.asarray(images))
                if images.is_cuda:
                    on_device[(sess, idx)] = encode(images, sess, labels_2d)
                else:
                    latents[sess][idx] = encode_trial(model, images, sess, labels_2d)
    finally:
        if serve_prev is not None:
            data_generator.serve_uint8 = serve_prev
    if on_device:
        keys = list(on_device)
        flat = torch.cat([on_device[k] for k in keys], dim=0).cpu().numpy()
        pos = 0
        for k in keys:
            n = on_device[k].shape[0]
            latents[k[0]][k[1]] = flat[pos:pos + n].copy()
            pos += n
        del on_device

    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(latents, gathered, dst=0)
        if rank != 0:
            return []
        for other in gathered[1:]:
            for s, per_sess in enumerate(other):
                for i, arr in enumerate(per_sess):
                    if arr.size:
                        latents[s][i] = arr
        missing = [key for key in wanted if latents[key[0]][key[1]].size == 0]
        if missing:
            raise RuntimeError('export_latents: %d trials were encoded by no rank, e.g. %s' % (
                len(missing), missing[:3]))

    filenames = []
    for sess, dataset in enumerate(data_generator.datasets):
        if filename is None:
            sess_id = '%s_%s_%s_%s_latents.pkl' % (
                dataset.lab, dataset.expt, dataset.animal, dataset.session)
            out = os.path.join(model.hparams['expt_dir'], 'version_%i' % model.version, sess_id)
        else:
            out = filename
        print('saving latents %i of %i:\n%s' % (sess + 1, data_generator.n_datasets, out))
        with open(out, 'wb') as f:
            pickle.dump({'latents': latents[sess], 'trials': dataset.batch_idxs}, f)
        filenames.append(out)
    return filenames


# which element of ``model(x)`` holds the latents (the reconstruction is always element 0): AE / cond-AE -> (x_hat, z);
# AEMSP -> (x_hat, z, y); VAE / beta-TC-VAE / cond-VAE -> (x_hat, z, mu, logvar); PS-VAE / MSPS-VAE -> (x_hat, sample,
# mu, logvar, y_hat), where the reference takes mu
_LATENTS_AT = {'ae': 1, 'cond-ae': 1, 'cond-ae-msp': 1, 'vae': 1, 'beta-tcvae': 1, 'cond-vae': 1, 'ps-vae': 2,
               'msps-vae': 2}


def get_reconstruction(model, inputs, dataset=None, return_latents=False, labels=None, labels_2d=None,
                       apply_inverse_transform=True, use_mean=True):
    """Images from images (through the whole model) or from latents (through the decoder) as numpy arrays
    (ref fitting/eval.py:286-374; the caller of ``forward`` / ``decoding`` behind the reference's plotting code).

    ``inputs`` with two dimensions are latents, anything else images.  Latents of the label-aware classes are
    completed first: cond-AE / cond-VAE get ``labels`` appended, AEMSP / PS-VAE / MSPS-VAE latents given in the
    transformed space are mapped back (``apply_inverse_transform``).  ``use_mean`` picks the posterior mean for the
    variational classes -- except cond-VAE, which the reference calls without it (kept: it samples there).
    Runs under ``no_grad`` in eval mode."""
    cls = model.hparams['model_class']
    model.eval()
    # (arrays go where the model's parameters are -- the reference sends them to hparams['device'], which a model
    # moved with .to() no longer matches)
    t = inputs if torch.is_tensor(inputs) else torch.Tensor(inputs).to(next(model.parameters()).device)
    with torch.no_grad():
        if t.dim() != 2:
            if cls not in _LATENTS_AT:
                raise ValueError('Invalid model class %s' % cls)
            kwargs = {'dataset': dataset}
            if cls in ('vae', 'beta-tcvae', 'ps-vae', 'msps-vae'):
                kwargs['use_mean'] = use_mean
            elif cls in ('cond-ae', 'cond-vae'):
                kwargs.update(labels=labels, labels_2d=labels_2d)
            out = model(t, **kwargs)
            recon, latents = out[0], out[_LATENTS_AT[cls]]
        else:
            if cls in ('cond-ae', 'cond-vae'):
                t = torch.cat((t, labels), dim=1)
            elif cls in ('cond-ae-msp', 'ps-vae', 'msps-vae') and apply_inverse_transform:
                t = model.get_inverse_transformed_latents(t, as_numpy=False)
            recon, latents = model.decoding(t, None, None, dataset=None), t
    recon, latents = recon.detach().cpu().numpy(), latents.detach().cpu().numpy()
    return (recon, latents) if return_latents else recon
