"""Variational autoencoder models on the MI355X HIP kernels.

Host-side mirror of the reference ``behavenet/models/vaes.py`` for the classes on the hot path:
``VAE``, ``ConditionalVAE``, ``BetaTCVAE``, ``PSVAE`` (+ ``ConvAEPSEncoder``), ``MSPSVAE``
(+ ``ConvAEMSPSEncoder``).  Quirks of the
reference that are reproduced on purpose:

* ``reparameterize`` uses ``std = exp(logvar)`` while the KL terms treat ``logvar`` as a
  log-variance;
* with ``vae.beta_anneal_epochs > 0`` the beta table continues with ones, not with beta;
* ``loss_data_mse`` / ``loss_mse`` of PS-VAE / beta-TC-VAE divide the *accumulated*
  log-likelihood by the current chunk size.
"""

import os

import numpy as np
import torch
from torch import nn

import behavenet_amd.fitting.losses as losses
from behavenet_amd import hip_functions as hf
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.hip_functions import linear
from behavenet_amd.models.aes import (
    AE, ConvAEDecoder, ConvAEEncoder, _no_sharded_chunk_loop, frame_masks)
from behavenet_amd.models.base import DiagLinear

__all__ = [
    'reparameterize', 'VAE', 'ConditionalVAE', 'BetaTCVAE', 'PSVAE', 'ConvAEPSEncoder',
    'MSPSVAE', 'ConvAEMSPSEncoder', 'set_eps_provider']

_eps_provider = None


def set_eps_provider(fn):
    """Install ``fn(like_tensor) -> eps`` used instead of ``torch.randn_like``.

    The reference draws eps from torch's CPU generator; the device generator produces a
    different stream, so parity tests inject the oracle's eps here.  ``None`` restores the
    default (device RNG).
    """
    global _eps_provider
    _eps_provider = fn


# BN_FUSED_PS_HEAD=0: the PS-VAE's latent head as separate autograd nodes (the first implementation)
_FUSED_PS_HEAD = os.environ.get('BN_FUSED_PS_HEAD', '1') != '0'


def _draw_eps(logvar, bounds):
    """eps ~ N(0, 1) of the shape of ``logvar``.  With an eps provider (tests): one call per chunk
    in chunk order with that chunk's shape, as the reference's chunk loop consumes them; else one
    draw for the whole batch."""
    if _eps_provider is None or bounds is None:
        return _eps_provider(logvar) if _eps_provider is not None else torch.randn_like(logvar)
    return torch.cat([_eps_provider(logvar[b:e]) for b, e in bounds], dim=0)


def reparameterize(mu, logvar, eps=None):
    """Sample ``mu + eps * exp(logvar)`` (ref vaes.py:17-35; std = exp(logvar), sic)."""
    if eps is None:
        eps = _eps_provider(logvar) if _eps_provider is not None else torch.randn_like(logvar)
    return hf.reparameterize_with_eps(mu, logvar, eps)


def _sample(mu, logvar, use_mean, bounds, shards=None):
    """Latents for a whole batch: the mean, or one ``reparameterize`` draw per chunk (the
    reference samples chunk by chunk; keeping that keeps the RNG / eps-provider sequence).

    ``shards`` (frame-sharded data parallelism): ``mu`` holds only this rank's rows of every
    chunk; eps is still drawn for the WHOLE chunk (every rank consumes the same random stream)
    and this rank's rows are cut out of it, so the ranks together use the single-device eps."""
    if use_mean:
        return mu
    if shards is not None and shards.sharded:
        parts = []
        for (b, e), (lb, le), (pb, pe) in zip(shards.bounds, shards.local, shards.bounds_l):
            like = torch.empty((e - b, logvar.shape[1]), dtype=logvar.dtype, device=logvar.device)
            eps = _eps_provider(like) if _eps_provider is not None else torch.randn_like(like)
            if pe > pb:
                parts.append(hf.reparameterize_with_eps(
                    mu[pb:pe].contiguous(), logvar[pb:pe].contiguous(),
                    eps[lb - b:le - b].contiguous()))
        return torch.cat(parts, dim=0) if parts else mu
    if bounds is None:
        return reparameterize(mu, logvar)
    return torch.cat([reparameterize(mu[b:e].contiguous(), logvar[b:e].contiguous())
                      for b, e in bounds], dim=0)


class _FrameShards(object):
    """The chunks of one batch and this rank's part of them (fitting/distributed.py, 'frames'
    mode).  ``bounds``: global [beg, end) of every chunk; ``local``: this rank's slice of it;
    ``bounds_l``: where those slices sit once packed back to back (what the model is run on);
    ``sizes``: global chunk lengths (every chunk term is a mean over the GLOBAL chunk);
    ``share``: this rank's fraction of each chunk (weights terms that every rank evaluates)."""

    def __init__(self, batch_size, chunk_size):
        self.bounds, self.local, self.sizes = bdist.shard_chunks(batch_size, chunk_size)
        self.sharded = self.local != self.bounds
        self.bounds_l, pos = [], 0
        for b, e in self.local:
            self.bounds_l.append((pos, pos + e - b))
            pos += e - b
        self.n_local = pos
        if self.sharded:
            # every rank evaluates this for ALL ranks, so that all of them refuse together (a
            # rank that raised alone would leave the others waiting in their collectives)
            _, R = bdist.shard_rank_world()
            for q in range(R):
                if all(bdist.shard_bounds(b, e, q, R)[0] == bdist.shard_bounds(b, e, q, R)[1]
                       for b, e in self.bounds):
                    raise NotImplementedError(
                        'frame sharding: a batch of %d frames (chunks of %d) leaves rank %d of '
                        '%d without a frame; the variational models need one frame per rank'
                        % (batch_size, chunk_size, q, R))
        self.share = [(le - lb) / float(n) for (lb, le), n in zip(self.local, self.sizes)]

    def take(self, *tensors):
        """This rank's rows of batch-leading tensors, packed (None stays None)."""
        if not self.sharded:
            return list(tensors)
        return [None if t is None else
                torch.cat([t[b:e] for b, e in self.local], dim=0).contiguous() for t in tensors]

    def share_t(self, device):
        """Per-chunk share as a device tensor (cached: building one from host floats is a
        synchronous copy that would stall the launch queue every step); 1.0 when not sharded."""
        if not self.sharded:
            return 1.0
        return hf.device_constant(self.share, device)

    def all_rows(self, t):
        """The rows of every rank (metrics over the whole batch); identity when not sharded."""
        return bdist.all_gather_rows(t.contiguous())[0] if self.sharded else t

    def row_terms(self, fn, *tensors):
        """(n_chunks,) tensor of ``fn(rows of chunk c)`` (a mean over those rows) weighted by
        this rank's share, i.e. its part of the global chunk mean; 0 for an empty slice."""
        out = []
        for (pb, pe), w in zip(self.bounds_l, self.share):
            if pe > pb:
                out.append(fn(*[t[pb:pe].contiguous() for t in tensors]) * float(w))
            else:
                out.append(torch.zeros((), dtype=torch.float32, device=tensors[0].device))
        return torch.stack(out)

    def kl_terms(self, mu, logvar):
        """(n_chunks,) tensor: this rank's part of every chunk's mean KL to N(0, 1) (one autograd
        node; same numbers as ``row_terms(losses.kl_div_to_std_normal, mu, logvar)``)."""
        return hf.kl_chunks(mu, logvar, self.bounds_l, self.share)

    def decomposed_kl_terms(self, z, mu, logvar):
        """(n_chunks, 3) tensor of (MI, TC, DWKL) per chunk.  Not sharded: one autograd node over
        the row ranges; sharded: every rank evaluates the gathered chunk (see ``gathered``)."""
        if not self.sharded:
            return hf.decomposed_kl_chunks(z, mu, logvar, self.bounds_l)
        return torch.stack([torch.stack(losses.decomposed_kl(*self.gathered(c, z, mu, logvar)))
                            for c in range(len(self.bounds_l))])

    def gathered(self, c, *tensors):
        """Chunk ``c``'s rows of every rank (rank order = frame order), with THIS rank's rows
        still attached to the graph: batch-coupled terms (the decomposed KL) are evaluated on
        the whole chunk by every rank and back-propagated through the local rows only -- the
        ranks' gradients then add up to the single-device gradient."""
        pb, pe = self.bounds_l[c]
        if not self.sharded:
            return [t[pb:pe] for t in tensors]
        out = []
        for t in tensors:
            mine = t[pb:pe]
            full, off = bdist.all_gather_rows(mine)
            out.append(torch.cat([full[:off], mine, full[off + mine.shape[0]:]], dim=0))
        return out


def _bounds(batch_size, chunk_size):
    return [(beg, min(beg + chunk_size, batch_size)) for beg in range(0, batch_size, chunk_size)]


def _finish_whole(table, total, accumulate_grad):
    """Read back the (n_chunks, n_keys) scalar table (summed over ranks when the frames are
    sharded: every entry is this rank's part of a global chunk mean), run the single backward,
    join streams."""
    rb = hf.Readback(bdist.all_reduce_(table.detach().clone()))
    if accumulate_grad:
        hf.backward_chunks([total], single_pass=True)
    hf.join_side_streams()
    return rb.numpy().astype(np.float64)


def _r2_variance_weighted(y_true, y_pred):
    """sklearn.metrics.r2_score(..., multioutput='variance_weighted') in numpy."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    if y_true.ndim == 1:
        y_true, y_pred = y_true[:, None], y_pred[:, None]
    num = ((y_true - y_pred) ** 2).sum(axis=0)
    den = ((y_true - y_true.mean(axis=0)) ** 2).sum(axis=0)
    nonzero = den != 0
    scores = np.ones(y_true.shape[1])
    scores[nonzero] = 1 - num[nonzero] / den[nonzero]
    scores[(num != 0) & ~nonzero] = 0.0
    if not np.any(nonzero):
        return float(np.mean(scores))
    return float(np.average(scores, weights=den))


def _anneal_tables(beta, anneal_epochs, max_n_epochs, tail_is_beta):
    tail = (beta if tail_is_beta else 1.0) * np.ones(max_n_epochs + 1)
    if anneal_epochs > 0:
        beta_vals = np.append(np.linspace(0, beta, anneal_epochs), tail)
        kl_vals = np.append(np.linspace(0, 1, anneal_epochs), np.ones(max_n_epochs + 1))
    else:
        beta_vals = beta * np.ones(max_n_epochs + 1)
        kl_vals = np.ones(max_n_epochs + 1)
    return beta_vals, kl_vals


def _stack_scalars(t):
    """One chunk's 0-dim loss terms as one small device tensor (dict order)."""
    return torch.stack([v.detach().float() for v in t.values()])


class VAE(AE):
    """Variational autoencoder / beta-VAE (ref vaes.py:38-208)."""

    graph_capturable = False      # (loss tail not routed through finish_loss yet)

    def __init__(self, hparams):
        if hparams['model_type'] == 'linear':
            raise NotImplementedError
        hparams['variational'] = True
        super().__init__(hparams)
        self.curr_epoch = 0  # set by fit()
        anneal = self.hparams.get('vae.beta_anneal_epochs', 0)
        # NB: after annealing the reference continues with ones, not with beta (vaes.py:93-100)
        self.beta_vals, _ = _anneal_tables(
            hparams['vae.beta'], anneal, hparams['max_n_epochs'], tail_is_beta=anneal <= 0)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        """-> (x_hat, z, mu, logvar)."""
        mu, logvar, pool_idx, outsize = self.encoding(x, dataset=dataset)
        z = _sample(mu, logvar, use_mean, kwargs.get('sample_bounds'), kwargs.get('sample_shards'))
        x_hat = self.decoding(z, pool_idx, outsize, dataset=dataset,
                              pixel_loss=kwargs.get('pixel_loss'))
        return x_hat, z, mu, logvar

    def _elbo_loss(self, data, dataset, accumulate_grad, chunk_size, fwd_kwargs_fn):
        x = data['images'][0]
        m = frame_masks(data, x)
        beta = self.beta_vals[self.curr_epoch]
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        rbs, sizes, deferred = hf.ChunkScalars(), [], []
        keys = ['loss', 'loss_ll', 'loss_kl']
        self._reserve_pools(x)
        groups = self._pass_groups(x, chunk_size)
        whole = groups is not None
        if whole:
            # one pass over the whole batch (batch norm under frame sharding: one per chunk),
            # latents sampled and losses normalised per chunk (see AE._loss_whole_batch)
            vals, sizes = [], []
            for gb, ge in groups:
                sh = _FrameShards(ge - gb, chunk_size)
                bounds = sh.bounds_l
                kw = fwd_kwargs_fn(gb, ge)
                xl, ml, *kw_l = sh.take(x[gb:ge], m[gb:ge] if m is not None else None,
                                        *kw.values())
                kw = dict(zip(kw.keys(), kw_l))
                with torch.set_grad_enabled(bool(accumulate_grad)), hf.bn_chunks(bounds):
                    x_hat, _, mu, logvar = self.forward(
                        xl, dataset=dataset, use_mean=False, sample_bounds=bounds,
                        sample_shards=sh,
                        pixel_loss={'target': xl, 'mask': ml, 'bounds': bounds, 'kind': 'll',
                                    'chunk_sizes': sh.sizes}, **kw)
                    ll = losses.gaussian_ll_chunks(xl, x_hat, ml, bounds, chunk_sizes=sh.sizes,
                                                   const_share=sh.share if sh.sharded else None)
                    klv = sh.kl_terms(mu, logvar)
                    lossv = -ll + float(beta) * klv
                vals.extend(_finish_whole(torch.stack([lossv, ll, klv], dim=1), lossv,
                                          accumulate_grad))
                sizes.extend(sh.sizes)
            n_chunks = 0
        else:
            _no_sharded_chunk_loop(self)
            self._prepare_first_layer(x, dataset)
            hf.begin_chunks(x.device)
        for chunk in range(n_chunks):
            beg = chunk * chunk_size
            end = min((chunk + 1) * chunk_size, batch_size)
            x_in = x[beg:end]
            m_in = m[beg:end] if m is not None else None
            with hf.chunk_stream(chunk, x.device, self._chunk_streams_ok()):
                with torch.set_grad_enabled(bool(accumulate_grad)):
                    x_hat, _, mu, logvar = self.forward(
                        x_in, dataset=dataset, use_mean=False, **fwd_kwargs_fn(beg, end))
                    loss_ll = losses.gaussian_ll(x_in, x_hat, m_in)
                    loss_kl = losses.kl_div_to_std_normal(mu, logvar)
                    loss = -loss_ll + float(beta) * loss_kl
                rbs.add(_stack_scalars({'loss': loss, 'loss_ll': loss_ll, 'loss_kl': loss_kl}))
            if accumulate_grad:
                deferred.append(loss)
            sizes.append(end - beg)
        if not whole:
            # read-backs enqueued with the forwards, collected after the deferred backwards
            vals = rbs.finish(deferred)
            self._release_first_layer()
        out = {k: 0.0 for k in keys}
        out['loss_mse'] = 0.0
        n_dims = np.prod(x.shape[1:])
        for row, bs in zip(vals, sizes):
            for k, v in zip(keys, row):
                out[k] += v * bs
            out['loss_mse'] += losses.gaussian_ll_to_mse(row[keys.index('loss_ll')], n_dims) * bs
        for k in out:
            out[k] = float(out[k] / batch_size)
        out['beta'] = beta
        return out

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        """-> {'loss','loss_ll','loss_kl','loss_mse','beta'} (ref vaes.py:131-208)."""
        return self._elbo_loss(data, dataset, accumulate_grad, chunk_size, lambda b, e: {})


class ConditionalVAE(VAE):
    """Conditional VAE: labels appended to the sampled latents (ref vaes.py:211-364)."""

    def __init__(self, hparams):
        super().__init__(hparams)

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents'] + self.hparams['n_labels']
        self.encoding = ConvAEEncoder(self.hparams)
        self.decoding = ConvAEDecoder(self.hparams)

    def forward(self, x, dataset=None, labels=None, labels_2d=None, use_mean=False, **kwargs):
        if self.hparams['conditional_encoder']:
            x = torch.cat((x, labels_2d), dim=1)
        mu, logvar, pool_idx, outsize = self.encoding(x, dataset=dataset)
        z = _sample(mu, logvar, use_mean, kwargs.get('sample_bounds'), kwargs.get('sample_shards'))
        z_aug = torch.cat((z, labels), dim=1)
        x_hat = self.decoding(z_aug, pool_idx, outsize, dataset=dataset,
                              pixel_loss=kwargs.get('pixel_loss'))
        return x_hat, z, mu, logvar

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        y = data['labels'][0]
        y_2d = data['labels_sc'][0] if self.hparams['conditional_encoder'] else None

        def kwargs(beg, end):
            return {'labels': y[beg:end],
                    'labels_2d': y_2d[beg:end] if y_2d is not None else None}
        return self._elbo_loss(data, dataset, accumulate_grad, chunk_size, kwargs)


class BetaTCVAE(VAE):
    """beta-TC-VAE: KL term decomposed into MI + beta*TC + dim-wise KL (ref vaes.py:367-503)."""

    def __init__(self, hparams):
        if hparams['model_type'] == 'linear':
            raise NotImplementedError
        super().__init__(hparams)
        self.curr_epoch = 0
        self.beta_vals, self.kl_anneal_vals = _anneal_tables(
            hparams['beta_tcvae.beta'], self.hparams.get('beta_tcvae.beta_anneal_epochs', 0),
            hparams['max_n_epochs'], tail_is_beta=True)

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x = data['images'][0]
        m = frame_masks(data, x)
        beta = self.beta_vals[self.curr_epoch]
        kl = self.kl_anneal_vals[self.curr_epoch]
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        rbs, sizes, deferred = hf.ChunkScalars(), [], []
        keys = ['loss', 'loss_ll', 'loss_mi', 'loss_tc', 'loss_dwkl']
        self._reserve_pools(x)
        groups = self._pass_groups(x, chunk_size)
        whole = groups is not None
        if whole:
            vals, sizes = [], []
            for gb, ge in groups:
                sh = _FrameShards(ge - gb, chunk_size)
                bounds = sh.bounds_l
                xl, ml = sh.take(x[gb:ge], m[gb:ge] if m is not None else None)
                with torch.set_grad_enabled(bool(accumulate_grad)), hf.bn_chunks(bounds):
                    x_hat, sample, mu, logvar = self.forward(
                        xl, dataset=dataset, use_mean=False, sample_bounds=bounds,
                        sample_shards=sh,
                        pixel_loss={'target': xl, 'mask': ml, 'bounds': bounds, 'kind': 'll',
                                    'chunk_sizes': sh.sizes})
                    ll = losses.gaussian_ll_chunks(xl, x_hat, ml, bounds, chunk_sizes=sh.sizes,
                                                   const_share=sh.share if sh.sharded else None)
                    # the decomposed KL couples all samples of a chunk: every rank evaluates it
                    # on the gathered chunk (gradients flow through its own rows only)
                    dk = sh.decomposed_kl_terms(sample, mu, logvar)              # (n_chunks, 3)
                    kl_terms = float(kl) * dk[:, 0] + float(beta) * dk[:, 1] + float(kl) * dk[:, 2]
                    lossv = -ll + kl_terms
                    # what is REPORTED is summed over ranks: the shared terms enter with this
                    # rank's share of the chunk
                    w = sh.share_t(x.device)
                    table = torch.cat([(-ll + w * kl_terms)[:, None], ll[:, None],
                                       dk * (w if isinstance(w, float) else w[:, None])], dim=1)
                vals.extend(_finish_whole(table, lossv, accumulate_grad))
                sizes.extend(sh.sizes)
            n_chunks = 0
        else:
            _no_sharded_chunk_loop(self)
            self._prepare_first_layer(x, dataset)
            hf.begin_chunks(x.device)
        for chunk in range(n_chunks):
            beg = chunk * chunk_size
            end = min((chunk + 1) * chunk_size, batch_size)
            x_in = x[beg:end]
            m_in = m[beg:end] if m is not None else None
            with hf.chunk_stream(chunk, x.device, self._chunk_streams_ok()):
                with torch.set_grad_enabled(bool(accumulate_grad)):
                    x_hat, sample, mu, logvar = self.forward(x_in, dataset=dataset,
                                                             use_mean=False)
                    ll = losses.gaussian_ll(x_in, x_hat, m_in)
                    mi, tc, dwkl = losses.decomposed_kl(sample, mu, logvar)
                    loss = -ll + float(kl) * mi + float(beta) * tc + float(kl) * dwkl
                rbs.add(_stack_scalars({'loss': loss, 'loss_ll': ll, 'loss_mi': mi,
                                            'loss_tc': tc, 'loss_dwkl': dwkl}))
            if accumulate_grad:
                deferred.append(loss)
            sizes.append(end - beg)
        if not whole:
            vals = rbs.finish(deferred)
            self._release_first_layer()
        out = {k: 0.0 for k in keys}
        out['loss_mse'] = 0.0
        n_dims = np.prod(x.shape[1:])
        for row, bs in zip(vals, sizes):
            for k, v in zip(keys, row):
                out[k] += v * bs
            # reference bookkeeping (vaes.py:494-495): accumulated ll / current chunk size
            out['loss_mse'] += losses.gaussian_ll_to_mse(out['loss_ll'] / bs, n_dims) * bs
        for k in out:
            out[k] = float(out[k] / batch_size)
        out['beta'] = beta
        return out


class ConvAEPSEncoder(ConvAEEncoder):
    """Encoder with fixed orthogonal projections A (labels) / B (rest) and the diagonal label
    map D (ref vaes.py:1276-1363)."""

    def __init__(self, hparams):
        super().__init__(hparams)
        n_latents = self.hparams['n_ae_latents']
        n_labels = self.hparams['n_labels']
        self.A = nn.Linear(n_latents, n_labels, bias=False)
        self.B = nn.Linear(n_latents, n_latents - n_labels, bias=False)
        self.D = DiagLinear(n_labels, bias=True)
        # rows of one random orthogonal matrix, frozen; seeded by numpy's global RNG like the
        # reference (scipy.stats.ortho_group.rvs, vaes.py:1296-1302)
        from scipy.stats import ortho_group
        m = ortho_group.rvs(dim=n_latents).astype('float32')
        with torch.no_grad():
            self.A.weight = nn.Parameter(torch.from_numpy(m[:n_labels, :]), requires_grad=False)
            self.B.weight = nn.Parameter(torch.from_numpy(m[n_labels:, :]), requires_grad=False)

    def __str__(self):
        out = 'Encoder architecture:\n'
        i = 0
        for i, module in enumerate(self.encoder):
            out += '    {:02d}: {}\n'.format(i, module)
        i += 1
        out += '    {:02d}: {}\n'.format(i, self.FF)
        out += '    {:02d}: {} (to constrained latents)\n'.format(i, self.A)
        out += '    {:02d}: {} (to unconstrained latents)\n'.format(i, self.B)
        out += '    {:02d}: {} (constrained latents to labels)\n'.format(i, self.D)
        return out

    def forward(self, x, dataset=None):
        """-> (y, w, logvar, pool_idx, output_sizes)."""
        x1 = self._features(x, dataset)
        h = linear(x1, self.FF.weight, self.FF.bias)
        y = linear(h, self.A.weight, None)
        w = linear(h, self.B.weight, None)
        pool_idx, sizes = self._pool_out()
        return y, w, linear(x1, self.logvar.weight, self.logvar.bias), pool_idx, sizes


class PSVAE(AE):
    """Partitioned-subspace VAE (ref vaes.py:506-846)."""

    graph_capturable = False      # (loss tail not routed through finish_loss yet)

    def __init__(self, hparams):
        if hparams['model_type'] == 'linear':
            raise NotImplementedError
        if hparams['n_ae_latents'] < hparams['n_labels']:
            raise ValueError('PS-VAE model must contain at least as many latents as labels')
        self.n_latents = hparams['n_ae_latents']
        self.n_labels = hparams['n_labels']
        hparams['variational'] = True
        super().__init__(hparams)
        self.curr_epoch = 0
        self.beta_vals, self.kl_anneal_vals = _anneal_tables(
            hparams['ps_vae.beta'], self.hparams.get('ps_vae.anneal_epochs', 0),
            hparams['max_n_epochs'], tail_is_beta=True)

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        if self.model_type == 'conv':
            self.encoding = ConvAEPSEncoder(self.hparams)
            self.decoding = ConvAEDecoder(self.hparams)
        elif self.model_type == 'linear':
            raise NotImplementedError
        else:
            raise ValueError('"%s" is an invalid model_type' % self.model_type)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        """-> (x_hat, z, mu, logvar, y_hat)."""
        return self._forward_parts(x, dataset, use_mean, **kwargs)[:5]

    def _forward_parts(self, x, dataset=None, use_mean=False, **kwargs):
        """forward() plus the two blocks (y, w) that mu is concatenated from: the loss takes the
        supervised / unsupervised means from them instead of slicing mu apart again."""
        y, w, logvar, pool_idx, outsize = self.encoding(x, dataset=dataset)
        mu = torch.cat([y, w], dim=1)
        z = _sample(mu, logvar, use_mean, kwargs.get('sample_bounds'), kwargs.get('sample_shards'))
        x_hat = self.decoding(z, pool_idx, outsize, dataset=dataset,
                              pixel_loss=kwargs.get('pixel_loss'))
        y_hat = self.encoding.D(y)
        return x_hat, z, mu, logvar, y_hat, y, w

    def _loss_terms_unfused(self, xl, yl, ml, nl, sh, bounds, share, dataset, alpha, beta, kl):
        """Per-chunk total, metric table and y_hat from separate autograd nodes per term: the path
        of frame-sharded data parallelism (the decomposed KL is evaluated on the gathered chunk)
        and of BN_FUSED_PS_HEAD=0."""
        n_labels = self.hparams['n_labels']
        x_hat, sample, mu, logvar, y_hat, mu_s, mu_u = self._forward_parts(
            xl, dataset=dataset, use_mean=False, sample_bounds=bounds, sample_shards=sh,
            pixel_loss={'target': xl, 'mask': ml, 'bounds': bounds, 'kind': 'll',
                        'chunk_sizes': sh.sizes})
        ll_x = losses.gaussian_ll_chunks(xl, x_hat, ml, bounds, chunk_sizes=sh.sizes,
                                         const_share=share)
        ll_y = losses.gaussian_ll_chunks(yl, y_hat, nl, bounds, chunk_sizes=sh.sizes,
                                         const_share=share)
        # column blocks once for the whole batch: the means are the encoder's own two heads,
        # log-variance and sample are split by one node each
        logvar_s, logvar_u = hf.split_cols(logvar, n_labels)
        _, sample_u = hf.split_cols(sample, n_labels)
        zs = sh.kl_terms(mu_s, logvar_s)
        # batch-coupled: evaluated on the gathered chunk by every rank (see BetaTCVAE)
        dk = sh.decomposed_kl_terms(sample_u, mu_u, logvar_u)          # (n_chunks, 3)
        w = sh.share_t(xl.device)
        if isinstance(w, float):
            # not sharded: the total and the metric table from one node
            lossv, terms = hf.combine_chunk_terms(
                [ll_x, ll_y, zs, dk], [[-1.0], [-float(alpha)], [1.0],
                                       [float(kl), float(beta), float(kl)]])
            table = torch.cat([terms, lossv.detach()[:, None]], dim=1)
        else:
            kl_terms = float(kl) * dk[:, 0] + float(beta) * dk[:, 1] + float(kl) * dk[:, 2]
            lossv = -ll_x - float(alpha) * ll_y + zs + kl_terms
            table = torch.cat([ll_x[:, None], ll_y[:, None], zs[:, None], dk * w[:, None],
                               (-ll_x - float(alpha) * ll_y + zs + w * kl_terms)[:, None]], dim=1)
        return lossv, table, y_hat

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        """Modified ELBO of the PS-VAE (ref vaes.py:603-729); returns the same 11 keys."""
        x = data['images'][0]
        y = data['labels'][0]
        m = frame_masks(data, x)
        n = data['labels_masks'][0] if 'labels_masks' in data else None
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        n_labels = self.hparams['n_labels']
        alpha = self.hparams['ps_vae.alpha']
        beta = self.beta_vals[self.curr_epoch]
        kl = self.kl_anneal_vals[self.curr_epoch]

        self._reserve_pools(x)
        rbs, sizes, y_hat_all, deferred = hf.ChunkScalars(), [], [], []
        keys = ['loss_data_ll', 'loss_label_ll', 'loss_zs_kl', 'loss_zu_mi', 'loss_zu_tc',
                'loss_zu_dwkl', 'loss']
        groups = self._pass_groups(x, chunk_size)
        whole = groups is not None
        if whole:
            # one pass over the whole batch (batch norm under frame sharding: one per chunk);
            # latents sampled, and every term normalised, per chunk (see AE._loss_whole_batch)
            vals, sizes, y_hat_rbs, y_rbs, n_rbs = [], [], [], [], []
            for gb, ge in groups:
                sh = _FrameShards(ge - gb, chunk_size)
                bounds = sh.bounds_l
                xl, yl, ml, nl = sh.take(x[gb:ge], y[gb:ge], m[gb:ge] if m is not None else None,
                                         n[gb:ge] if n is not None else None)
                share = sh.share if sh.sharded else None
                fused_head = not sh.sharded and _FUSED_PS_HEAD and xl.is_cuda
                with torch.set_grad_enabled(bool(accumulate_grad)), hf.bn_chunks(bounds):
                    if fused_head:
                        # encoder heads -> ONE node (latents, label head, label / KL /
                        # decomposed-KL terms of every chunk) -> decoder with the pixel loss in
                        # its last kernel
                        mu_s, mu_u, logvar, pool_idx, outsize = self.encoding(xl, dataset=dataset)
                        eps = _draw_eps(logvar, bounds)
                        sample, lat_terms, y_hat, cols5 = hf.psvae_head(
                            mu_s, mu_u, logvar, self.encoding.D, yl, nl, eps, bounds, alpha, kl,
                            beta)
                        x_hat = self.decoding(sample, pool_idx, outsize, dataset=dataset,
                                              pixel_loss={'target': xl, 'mask': ml,
                                                          'bounds': bounds, 'kind': 'll',
                                                          'chunk_sizes': sh.sizes})
                        ll_x = losses.gaussian_ll_chunks(xl, x_hat, ml, bounds,
                                                         chunk_sizes=sh.sizes)
                        lossv = lat_terms - ll_x
                        table = torch.cat([ll_x.detach()[:, None], cols5,
                                           lossv.detach()[:, None]], dim=1)
                    else:
                        lossv, table, y_hat = self._loss_terms_unfused(
                            xl, yl, ml, nl, sh, bounds, share, dataset, alpha, beta, kl)
                # label r^2 over the whole batch: the ranks' rows are gathered (a few KB)
                y_hat_rbs.append(hf.Readback(sh.all_rows(y_hat.detach())))
                y_rbs.append(hf.Readback(sh.all_rows(yl)))
                if n is not None:
                    n_rbs.append(hf.Readback(sh.all_rows(nl)))
                vals.extend(_finish_whole(table, lossv, accumulate_grad))
                sizes.extend(sh.sizes)
            n_chunks = 0
        else:
            _no_sharded_chunk_loop(self)
            self._prepare_first_layer(x, dataset)
        for chunk in range(n_chunks):
            beg = chunk * chunk_size
            end = min((chunk + 1) * chunk_size, batch_size)
            x_in, y_in = x[beg:end], y[beg:end]
            m_in = m[beg:end] if m is not None else None
            n_in = n[beg:end] if n is not None else None
            with torch.set_grad_enabled(bool(accumulate_grad)):
                x_hat, sample, mu, logvar, y_hat = self.forward(
                    x_in, dataset=dataset, use_mean=False)
                t = {}
                t['loss_data_ll'] = losses.gaussian_ll(x_in, x_hat, m_in)
                t['loss_label_ll'] = losses.gaussian_ll(y_in, y_hat, n_in)
                t['loss_zs_kl'] = losses.kl_div_to_std_normal(
                    mu[:, :n_labels].contiguous(), logvar[:, :n_labels].contiguous())
                mi, tc, dwkl = losses.decomposed_kl(
                    sample[:, n_labels:], mu[:, n_labels:], logvar[:, n_labels:])
                t['loss_zu_mi'], t['loss_zu_tc'], t['loss_zu_dwkl'] = mi, tc, dwkl
                t['loss'] = -t['loss_data_ll'] - float(alpha) * t['loss_label_ll'] \
                    + t['loss_zs_kl'] + float(kl) * mi + float(beta) * tc + float(kl) * dwkl
            if accumulate_grad:
                deferred.append(t['loss'])
            assert list(t.keys()) == keys
            rbs.add(_stack_scalars(t))
            sizes.append(end - beg)
            y_hat_all.append(y_hat.detach())

        if not whole:
            # read-backs enqueued between the forwards and the deferred backwards (see AE.loss).
            # One stream only: the diagonal label head D accumulates through AccumulateGrad.
            y_hat_rb = hf.Readback(torch.cat(y_hat_all, dim=0))
            y_rb = hf.Readback(y)
            n_rb = hf.Readback(n) if n is not None else None
            vals = rbs.finish(deferred)
            self._release_first_layer()
        if whole:
            y_hat_np = np.concatenate([rb.numpy() for rb in y_hat_rbs], axis=0)
            y_np = np.concatenate([rb.numpy() for rb in y_rbs], axis=0)
            n_np = np.concatenate([rb.numpy() for rb in n_rbs], axis=0) if n is not None else None
        else:
            y_hat_np, y_np = y_hat_rb.numpy(), y_rb.numpy()
            n_np = n_rb.numpy() if n_rb is not None else None
        order = ['loss', 'loss_data_ll', 'loss_label_ll', 'loss_zs_kl', 'loss_zu_mi',
                 'loss_zu_tc', 'loss_zu_dwkl']
        out = {k: 0.0 for k in order}
        out['loss_data_mse'] = 0.0
        n_dims = np.prod(x.shape[1:])
        for row, bs in zip(vals, sizes):
            for k, v in zip(keys, row):
                out[k] += v * bs
            # reference bookkeeping (vaes.py:705-706): accumulated ll / current chunk size
            out['loss_data_mse'] += losses.gaussian_ll_to_mse(
                out['loss_data_ll'] / bs, n_dims) * bs

        if n is not None:
            r2 = _r2_variance_weighted(y_np[n_np == 1], y_hat_np[n_np == 1])
        else:
            r2 = _r2_variance_weighted(y_np, y_hat_np)

        for k in out:
            out[k] = float(out[k] / batch_size)
        out['alpha'] = alpha
        out['beta'] = beta
        out['label_r2'] = r2
        return out

    def get_predicted_labels(self, x, dataset=None, use_mean=True):
        y, w, logvar, _, _ = self.encoding(x, dataset=dataset)
        if not use_mean:
            y = reparameterize(y, logvar[:, :self.n_labels].contiguous())
        return self.encoding.D(y)

    def get_transformed_latents(self, inputs, dataset=None, as_numpy=True):
        """Latents with the supervised block mapped to label space by D (ref vaes.py:755-800)."""
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.Tensor(inputs)
        inputs = inputs.to(self.encoding.D.bias.device)
        if len(inputs.shape) == 2:
            y_og = inputs[:, :self.hparams['n_labels']]
            w_og = inputs[:, self.hparams['n_labels']:]
        else:
            y_og, w_og, _, _, _ = self.encoding(inputs, dataset=dataset)
        out = torch.cat([self.encoding.D(y_og), w_og], dim=1)
        return out.cpu().detach().numpy() if as_numpy else out

    def get_inverse_transformed_latents(self, inputs, dataset=None, as_numpy=True):
        """Inverse of :meth:`get_transformed_latents` for latent inputs (ref vaes.py:802-846)."""
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.Tensor(inputs)
        if len(inputs.shape) != 2:
            raise NotImplementedError
        inputs = inputs.to(self.encoding.D.bias.device)
        y_og = inputs[:, :self.hparams['n_labels']]
        w_og = inputs[:, self.hparams['n_labels']:]
        y_new = torch.div(torch.sub(y_og, self.encoding.D.bias), self.encoding.D.weight)
        out = torch.cat([y_new, w_og], dim=1)
        return out.cpu().detach().numpy() if as_numpy else out


class ConvAEMSPSEncoder(ConvAEEncoder):
    """PS encoder with a background head: frozen orthogonal rows A (labels), C (background,
    trainable bias) and B (rest), diagonal label map D (ref vaes.py:1366-1470)."""

    def __init__(self, hparams):
        super().__init__(hparams)
        n_latents = self.hparams['n_ae_latents']
        n_labels = self.hparams['n_labels']
        n_background = self.hparams['n_background']
        # construction order = the reference's: it fixes the torch RNG stream of C.bias
        self.A = nn.Linear(n_latents, n_labels, bias=False)
        self.B = nn.Linear(n_latents, n_latents - n_labels - n_background, bias=False)
        self.C = nn.Linear(n_latents, n_background, bias=True)
        self.D = DiagLinear(n_labels, bias=True)
        from scipy.stats import ortho_group
        m = ortho_group.rvs(dim=n_latents).astype('float32')
        with torch.no_grad():
            self.A.weight = nn.Parameter(torch.from_numpy(m[:n_labels, :]), requires_grad=False)
            self.B.weight = nn.Parameter(
                torch.from_numpy(m[n_labels + n_background:, :]), requires_grad=False)
            self.C.weight = nn.Parameter(
                torch.from_numpy(m[n_labels:n_labels + n_background, :]), requires_grad=False)

    def __str__(self):
        out = 'Encoder architecture:\n'
        i = 0
        for i, module in enumerate(self.encoder):
            out += '    {:02d}: {}\n'.format(i, module)
        i += 1
        out += '    {:02d}: {}\n'.format(i, self.FF)
        out += '    {:02d}: {} (to supervised latents)\n'.format(i, self.A)
        out += '    {:02d}: {} (to unsupervised latents)\n'.format(i, self.B)
        out += '    {:02d}: {} (to background latents)\n'.format(i, self.C)
        out += '    {:02d}: {} (supervised latents to labels)\n'.format(i, self.D)
        return out

    def forward(self, x, dataset=None):
        """-> (z_s, z_b, z_u, logvar, pool_idx, output_sizes)."""
        x1 = self._features(x, dataset)
        h = linear(x1, self.FF.weight, self.FF.bias)
        z_s = linear(h, self.A.weight, None)
        z_u = linear(h, self.B.weight, None)
        z_b = linear(h, self.C.weight, self.C.bias)
        pool_idx, sizes = self._pool_out()
        return z_s, z_b, z_u, linear(x1, self.logvar.weight, self.logvar.bias), pool_idx, sizes


class MSPSVAE(PSVAE):
    """Multi-session PS-VAE (ref vaes.py:849-1273): a training batch is a LIST of per-session
    batches, concatenated and run through the network in one pass (the reference does not chunk
    this model); a triplet loss on the background latents separates the sessions."""

    def __init__(self, hparams):
        if hparams['n_sessions_per_batch'] == 1:
            raise ValueError('must choose "n_sessions_per_batch" > 1 in hparams')
        hparams['n_background'] = hparams.get('n_background', 4)   # saved with the hparams
        super().__init__(hparams)
        self.TripletLoss = nn.TripletMarginLoss(margin=1.0, p=2)

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        if self.model_type == 'conv':
            self.encoding = ConvAEMSPSEncoder(self.hparams)
            self.decoding = ConvAEDecoder(self.hparams)
        elif self.model_type == 'linear':
            raise NotImplementedError
        else:
            raise ValueError('"%s" is an invalid model_type' % self.model_type)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        """-> (x_hat, z, mu, logvar, y_hat); mu = [z_s | z_b | z_u]."""
        z_s, z_b, z_u, logvar, pool_idx, outsize = self.encoding(x, dataset=dataset)
        mu = torch.cat([z_s, z_b, z_u], dim=1)
        z = _sample(mu, logvar, use_mean, None)
        x_hat = self.decoding(z, pool_idx, outsize, dataset=dataset)
        y_hat = self.encoding.D(z_s)
        return x_hat, z, mu, logvar, y_hat

    def loss(self, datas, dataset=None, accumulate_grad=True, chunk_size=None):
        """Modified ELBO + triplet term (ref vaes.py:916-1098).  ``datas``: list of data dicts
        (training; ``dataset`` = list of their session ids) or one dict (validation / test; no
        triplet term, the key is reported as 0 like the reference)."""
        if bdist.frames_sharded():
            raise NotImplementedError('frame-sharded data parallelism of MSPSVAE is not '
                                      'implemented (use dp_shard="trial")')
        multi = isinstance(datas, list)
        if multi:
            x = torch.cat([d['images'][0] for d in datas], dim=0)
            y = torch.cat([d['labels'][0] for d in datas], dim=0)
            m = torch.cat([frame_masks(d, d['images'][0]) for d in datas], dim=0) \
                if 'masks' in datas[0] else None
            n = torch.cat([d['labels_masks'][0] for d in datas], dim=0) \
                if 'labels_masks' in datas[0] else None
            sess_ids = np.concatenate(
                [d * np.ones(datas[i]['images'].shape[1]) for i, d in enumerate(dataset)])
        else:
            x, y = datas['images'][0], datas['labels'][0]
            m = frame_masks(datas, datas['images'][0])
            n = datas['labels_masks'][0] if 'labels_masks' in datas else None
            sess_ids = None
        n_labels = self.hparams['n_labels']
        n_bg = self.hparams['n_background']
        alpha = self.hparams['ps_vae.alpha']
        delta = self.hparams['ps_vae.delta']
        beta = self.beta_vals[self.curr_epoch]
        kl = self.kl_anneal_vals[self.curr_epoch]

        self._reserve_pools(x)
        u0 = n_labels + n_bg
        with torch.set_grad_enabled(bool(accumulate_grad)):
            x_hat, sample, mu, logvar, y_hat = self.forward(x, dataset=None, use_mean=False)
            t = {}
            t['loss_data_ll'] = losses.gaussian_ll(x, x_hat, m)
            t['loss_label_ll'] = losses.gaussian_ll(y, y_hat, n)
            t['loss_zs_kl'] = losses.kl_div_to_std_normal(
                mu[:, :n_labels].contiguous(), logvar[:, :n_labels].contiguous())
            mi, tc, dwkl = losses.decomposed_kl(sample[:, u0:], mu[:, u0:], logvar[:, u0:])
            t['loss_zu_mi'], t['loss_zu_tc'], t['loss_zu_dwkl'] = mi, tc, dwkl
            total = -t['loss_data_ll'] - float(alpha) * t['loss_label_ll'] + t['loss_zs_kl'] \
                + float(kl) * mi + float(beta) * tc + float(kl) * dwkl
            if multi:
                t['loss_triplet'] = losses.triplet_loss(
                    self.TripletLoss, mu[:, n_labels:u0], sess_ids)
                total = total + float(delta) * t['loss_triplet']
            t['loss'] = total
        keys = list(t.keys())
        table = _stack_scalars(t)
        y_hat_rb, y_rb = hf.Readback(y_hat), hf.Readback(y)
        n_rb = hf.Readback(n) if n is not None else None
        vals = _finish_whole(table.reshape(1, -1), total, accumulate_grad)[0]
        out = {'loss': 0.0}
        out.update({k: float(v) for k, v in zip(keys, vals)})
        out.setdefault('loss_triplet', 0)
        out['loss_data_mse'] = losses.gaussian_ll_to_mse(out['loss_data_ll'], np.prod(x.shape[1:]))
        y_hat_np, y_np = y_hat_rb.numpy(), y_rb.numpy()
        if n is not None:
            n_np = n_rb.numpy()
            r2 = _r2_variance_weighted(y_np[n_np == 1], y_hat_np[n_np == 1])
        else:
            r2 = _r2_variance_weighted(y_np, y_hat_np)
        out.update({'alpha': alpha, 'beta': beta, 'delta': delta, 'label_r2': r2})
        return out

    def get_predicted_labels(self, x, dataset=None, use_mean=True):
        z_s, _, _, logvar, _, _ = self.encoding(x, dataset=dataset)
        if not use_mean:
            z_s = reparameterize(z_s, logvar[:, :self.n_labels].contiguous())
        return self.encoding.D(z_s)

    def _split_latents(self, inputs, dataset):
        n_labels, n_bg = self.hparams['n_labels'], self.hparams['n_background']
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.Tensor(inputs)
        inputs = inputs.to(self.encoding.D.bias.device)
        if len(inputs.shape) == 2:
            return inputs[:, :n_labels], inputs[:, n_labels:n_labels + n_bg], \
                inputs[:, n_labels + n_bg:], True
        z_s, z_b, z_u, _, _, _ = self.encoding(inputs, dataset=dataset)
        return z_s, z_b, z_u, False

    def get_transformed_latents(self, inputs, dataset=None, as_numpy=True):
        """Latents with the supervised block mapped to label space by D (ref vaes.py:1100-1147)."""
        z_s, z_b, z_u, _ = self._split_latents(inputs, dataset)
        out = torch.cat([self.encoding.D(z_s), z_b, z_u], dim=1)
        return out.cpu().detach().numpy() if as_numpy else out

    def get_inverse_transformed_latents(self, inputs, dataset=None, as_numpy=True):
        """Inverse of :meth:`get_transformed_latents`, latent inputs only (ref :1149-1196)."""
        if not isinstance(inputs, torch.Tensor):
            inputs = torch.Tensor(inputs)
        if len(inputs.shape) != 2:
            raise NotImplementedError
        z_s, z_b, z_u, _ = self._split_latents(inputs, dataset)
        z_new = torch.div(torch.sub(z_s, self.encoding.D.bias), self.encoding.D.weight)
        out = torch.cat([z_new, z_b, z_u], dim=1)
        return out.cpu().detach().numpy() if as_numpy else out

    def export_latents(self, data_generator, filename=None):
        """Latents [z_s | z_b | z_u] of every trial of every session, one pickle per session
        (ref vaes.py:1198-1273; the reference rebuilds a one-session-per-batch generator from
        disk, here the generator is asked for single batches: ``return_multiple=False``)."""
        from behavenet_amd.fitting.eval import export_latents
        return export_latents(data_generator, self, filename=filename)
