"""Autoencoder models on the MI355X HIP kernels.

Host-side mirror of the reference ``behavenet/models/aes.py``: same class names, constructor
hparams, ``forward/encoding/decoding/loss`` signatures and ``state_dict`` keys, so a reference
checkpoint loads here and vice versa.  The ``nn.Conv2d`` / ``nn.ConvTranspose2d`` /
``nn.Linear`` children are kept as *parameter containers only* (identical registration order
=> identical initialisation from ``torch.manual_seed``); their own ``forward`` is never used.
The arithmetic runs through :mod:`behavenet_amd.hip_functions`.
"""

import os

import numpy as np
import torch
from torch import nn

import behavenet_amd.fitting.losses as losses
from behavenet_amd import _hip
from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.models.base import BaseModule, BaseModel
from behavenet_amd.hip_functions import (
    ChunkScalars, ConvLayerPlan, FusedPixelLoss, Readback, activation, backward_chunks, bn_chunks,
    capturing, finish_loss,
    chunked_sq_err, conv_stack, conv_stack_bn, conv_stack_sq_err, first_layer_forward,
    join_side_streams, linear, begin_chunks, chunk_stream, max_pool, max_pool_act, max_unpool, conv_pool_act,
    pixel_loss_scales, reserve_device_pools)

__all__ = [
    'ConvAEEncoder', 'ConvAEDecoder', 'LinearAEEncoder', 'LinearAEDecoder', 'AE', 'ConditionalAE',
    'AEMSP',
    'load_pretrained_ae']


def _mark_footprint(module):
    module._bn_counts_for_footprint = True
    return module


def _unsupported_on_hip(what):
    raise NotImplementedError(
        '%s is not implemented by the MI355X kernels yet (SURVEY.md section 8(f), rank 3)' % what)


def _no_sharded_chunk_loop(model):
    """The chunk-by-chunk schedules (batch-norm variants) of the classes other than the plain AE
    are not frame-sharded: refuse instead of silently using per-rank statistics."""
    if bdist.frames_sharded():
        raise NotImplementedError(
            'frame-sharded data parallelism of %s with ae_batch_norm=1 is not implemented '
            '(use dp_shard="trial", or the plain AE)' % type(model).__name__)


def frame_masks(data, x):
    """``data['masks'][0]`` as an (N, C, H, W) tensor or None.  The reference's generator serves
    ONE (C, H, W) mask per trial (its first frame's) that the losses broadcast over the frames
    (losses.py:56-59 with data_generator.py:264-277); per-frame masks pass through."""
    if 'masks' not in data:
        return None
    m = data['masks'][0]
    if m.dim() == x.dim() - 1:
        m = m.unsqueeze(0)
    if m.shape[0] == 1 and x.shape[0] != 1:
        m = m.expand_as(x)
    return m.contiguous()


def _bn_modules(container, layer_names):
    """nn.BatchNorm2d following each (transposed) convolution of the stack, or None."""
    out = []
    for name in layer_names:
        num = ''.join(ch for ch in name.split('_')[0] if ch.isdigit())
        out.append(getattr(container, 'batchnorm%s' % num, None))
    return out



def _tap_signs(plan, j, layer, h):
    """Tests only (tests/branches.py): the layer-by-layer paths of max-pooling architectures run every layer as
    its own one-layer stack, so the LeakyReLU signs are filed under the module's plan here."""
    from behavenet_amd import hip_functions as hf
    if hf._sign_tap is not None and layer.act == _hip.ACT_LRELU:
        hf._sign_tap.setdefault(id(plan), [[] for _ in plan])[j].append((h.detach() > 0).cpu())


class ConvAEEncoder(BaseModule):
    """Convolutional encoder (ref aes.py:17-218)."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.encoder = None
        self.build_model()

    def __str__(self):
        out = 'Encoder architecture:\n'
        i = 0
        for i, module in enumerate(self.encoder):
            out += '    {:02d}: {}\n'.format(i, module)
        out += '    {:02d}: {}\n'.format(i + 1 if len(self.encoder) else 0, self.FF)
        return out

    def build_model(self):
        hp = self.hparams
        self.encoder = nn.ModuleList()
        self._layer_names = []   # conv module name per fused layer
        self._plan = []
        self._pool_after = []    # per conv layer: None or (kernel, stride, (pad_t, pad_l), (Ho, Wo))
        n_layers = len(hp['ae_encoding_n_channels'])
        gnum = 0
        for i in range(n_layers):
            if hp['ae_encoding_layer_type'][i] != 'conv':
                continue
            args, pads = self._get_conv2d_args(i, gnum)
            if hp.get('fit_sess_io_layers', False) and i == 0:
                name = 'conv%i_sess_io_layers' % gnum
                self.encoder.add_module(name, nn.ModuleList([
                    _mark_footprint(nn.Conv2d(**args)) for _ in range(hp['n_datasets'])]))
            else:
                name = 'conv%i' % gnum
                self.encoder.add_module(name, _mark_footprint(nn.Conv2d(**args)))
            if hp['ae_batch_norm']:
                self.encoder.add_module('batchnorm%i' % gnum, nn.BatchNorm2d(
                    hp['ae_encoding_n_channels'][i],
                    momentum=hp.get('ae_batch_norm_momentum', 0.1),
                    track_running_stats=hp.get('track_running_stats', True)))
            pool = None
            if i < n_layers - 1 and hp['ae_encoding_layer_type'][i + 1] == 'maxpool':
                pargs = self._get_maxpool2d_args(i)
                self.encoder.add_module('maxpool%i' % gnum, _mark_footprint(nn.MaxPool2d(**pargs)))
                # ceil_mode (same) or not (valid): either way the planner's output size says how
                # many windows there are, and the kernel clips windows at the border
                pool = (pargs['kernel_size'], pargs['stride'], pargs['padding'],
                        (hp['ae_encoding_y_dim'][i + 1], hp['ae_encoding_x_dim'][i + 1]))
            self._pool_after.append(pool)
            self.encoder.add_module('relu%i' % gnum, nn.LeakyReLU(0.05))

            if i == 0:
                hin, win = hp['ae_input_dim'][1], hp['ae_input_dim'][2]
            else:
                hin, win = hp['ae_encoding_y_dim'][i - 1], hp['ae_encoding_x_dim'][i - 1]
            k = args['kernel_size']
            self._plan.append(ConvLayerPlan(
                'conv', args['in_channels'], hin, win, args['out_channels'],
                hp['ae_encoding_y_dim'][i], hp['ae_encoding_x_dim'][i], k, k, args['stride'],
                pads[0], pads[1], _hip.ACT_LRELU))
            self._layer_names.append(name)
            gnum += 1

        last = hp['ae_encoding_n_channels'][-1] * hp['ae_encoding_y_dim'][-1] * \
            hp['ae_encoding_x_dim'][-1]
        self.FF = _mark_footprint(nn.Linear(last, hp['n_ae_latents']))
        if hp.get('variational', False):
            self.logvar = _mark_footprint(nn.Linear(last, hp['n_ae_latents']))

    def _get_conv2d_args(self, layer, global_layer):
        """nn.Conv2d kwargs + the (top, left) zero padding folded into the kernel (ref :127-163)."""
        hp = self.hparams
        if layer == 0:
            extra = 0
            if hp['model_class'] == 'cond-ae' and hp.get('conditional_encoder', False):
                extra = int(hp['n_labels'] / 2)  # x/y label coordinates share one 2d map
            in_channels = hp['ae_input_dim'][0] + extra
        else:
            in_channels = hp['ae_encoding_n_channels'][layer - 1]
        x0, x1 = hp['ae_encoding_x_padding'][layer]
        y0, y1 = hp['ae_encoding_y_padding'][layer]
        if x0 == x1 and y0 == y1:
            padding = (y0, x0)
        else:
            # asymmetric TF-"same" padding: the reference inserts a ZeroPad2d; the module is kept
            # (names / printing) but the padding is folded into the conv kernel's tile load
            self.encoder.add_module('zero_pad%i' % global_layer, nn.ZeroPad2d((x0, x1, y0, y1)))
            padding = 0
        args = {
            'in_channels': in_channels,
            'out_channels': hp['ae_encoding_n_channels'][layer],
            'kernel_size': hp['ae_encoding_kernel_size'][layer],
            'stride': hp['ae_encoding_stride_size'][layer],
            'padding': padding}
        return args, (y0, x0)

    def _get_maxpool2d_args(self, layer):
        hp = self.hparams
        return {
            'kernel_size': int(hp['ae_encoding_kernel_size'][layer + 1]),
            'stride': int(hp['ae_encoding_stride_size'][layer + 1]),
            'padding': (hp['ae_encoding_y_padding'][layer + 1][0],
                        hp['ae_encoding_x_padding'][layer + 1][0]),
            'return_indices': True,
            'ceil_mode': hp['ae_padding_type'] != 'valid'}

    def _stack_params(self, dataset):
        params = []
        for name in self._layer_names:
            mod = getattr(self.encoder, name)
            if isinstance(mod, nn.ModuleList):
                mod = mod[dataset]
            params += [mod.weight, mod.bias]
        return params

    def _features(self, x, dataset=None):
        """Run the conv stack -> (N, C*H*W) post-LeakyReLU features.  For max-pooling
        architectures the pooling indices and pre-pool sizes are left in ``self._pool_state``
        (see :meth:`_pool_out`)."""
        hp = self.hparams
        self._pool_state = ([], [])
        if any(p is not None for p in self._pool_after):
            return self._features_pooled(x, dataset)
        if hp['ae_batch_norm']:
            h = conv_stack_bn(self._plan, x, self._stack_params(dataset),
                              _bn_modules(self.encoder, self._layer_names))
        else:
            h = conv_stack(self._plan, x, self._stack_params(dataset),
                           h1=self._first_layer_slice(x, dataset))
        return h.view(h.size(0), -1)

    def _features_pooled(self, x, dataset):
        """conv [-> batch norm] -> max pool (indices kept) -> LeakyReLU, layer by layer
        (ref aes.py:99-114,200-211: the activation follows the pooling)."""
        params = self._stack_params(dataset)
        bns = _bn_modules(self.encoder, self._layer_names)
        pool_idx, sizes = [], []
        h = x
        for j, layer in enumerate(self._plan):
            pool = self._pool_after[j]
            one = [layer.with_act(_hip.ACT_NONE) if pool is not None else layer]
            fused = None
            if pool is not None and bns[j] is None:
                # conv -> pool -> LeakyReLU of the first layer in one kernel where the library serves it
                fused = conv_pool_act(one[0], h, params[2 * j:2 * j + 2], *pool, _hip.ACT_LRELU)
            if fused is not None:
                sizes.append(torch.Size((h.size(0), layer.cout, layer.hout, layer.wout)))
                h, idx = fused
                pool_idx.append(idx)
                _tap_signs(self._plan, j, layer, h)
                continue
            if bns[j] is not None:
                h = conv_stack_bn(one, h, params[2 * j:2 * j + 2], [bns[j]])
            else:
                h = conv_stack(one, h, params[2 * j:2 * j + 2])
            if pool is not None:
                k, stride, pad, out_hw = pool
                sizes.append(h.size())
                h, idx = max_pool_act(h, k, stride, pad, out_hw, _hip.ACT_LRELU)
                pool_idx.append(idx)
            _tap_signs(self._plan, j, layer, h)
        self._pool_state = (pool_idx, sizes)
        return h.reshape(h.size(0), -1)

    def _pool_out(self):
        """(pool_idx, output_sizes) of the last forward: what the decoder's unpooling needs."""
        state = getattr(self, '_pool_state', ([], []))
        self._pool_state = ([], [])
        return state

    # -- whole-batch first layer ----------------------------------------------------------
    # enc.conv0 is HBM-bound (512 KB written per frame) and frames are independent, so the
    # models' loss() runs it ONCE for the whole batch (one large launch on an otherwise idle
    # GPU) before the 200-frame chunks fork; each chunk then picks its rows out of that output.
    def prepare_first_layer(self, x_all, dataset=None):
        """Run layer 1 for every frame of ``x_all`` (a contiguous (B, C, H, W) device tensor)."""
        self._h1_cache = None
        if self.hparams['ae_batch_norm'] or not x_all.is_cuda or not x_all.is_contiguous() \
                or tuple(x_all.shape[1:]) != (self._plan[0].cin, self._plan[0].hin,
                                              self._plan[0].win):
            return
        with torch.no_grad():
            h1 = first_layer_forward(self._plan, x_all, self._stack_params(dataset))
        self._h1_cache = (x_all, h1, dataset)

    def release_first_layer(self):
        self._h1_cache = None

    def _first_layer_slice(self, x, dataset):
        cache = getattr(self, '_h1_cache', None)
        if cache is None:
            return None
        x_all, h1, ds = cache
        frame_bytes = x_all[0].numel() * x_all.element_size()
        delta = x.data_ptr() - x_all.data_ptr()
        if ds != dataset or not x.is_contiguous() or x.shape[1:] != x_all.shape[1:] or \
                delta < 0 or delta % frame_bytes != 0:
            return None
        beg = delta // frame_bytes
        if beg + x.shape[0] > x_all.shape[0]:
            return None
        return h1[beg:beg + x.shape[0]]

    def forward(self, x, dataset=None):
        """-> (latents, pool_idx, output_sizes) or (mu, logvar, pool_idx, output_sizes)."""
        x1 = self._features(x, dataset)
        pool_idx, sizes = self._pool_out()
        if self.hparams.get('variational', False):
            return (linear(x1, self.FF.weight, self.FF.bias),
                    linear(x1, self.logvar.weight, self.logvar.bias), pool_idx, sizes)
        return linear(x1, self.FF.weight, self.FF.bias), pool_idx, sizes


class ConvAEDecoder(BaseModule):
    """Convolutional decoder (ref aes.py:221-488)."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.decoder = None
        self.build_model()

    def __str__(self):
        out = 'Decoder architecture:\n'
        out += '    {:02d}: {}\n'.format(0, self.FF)
        for i, module in enumerate(self.decoder):
            out += '    {:02d}: {}\n'.format(i + 1, module)
        return out

    def build_model(self):
        hp = self.hparams
        start = hp['ae_decoding_starting_dim']
        self.FF = _mark_footprint(nn.Linear(hp['hidden_layer_size'], start[0] * start[1] * start[2]))
        self.decoder = nn.ModuleList()
        self.conv_t_pads = {}
        self._layer_names = []
        self._plan = []
        self._unpool_before = []     # per convT layer: a MaxUnpool2d precedes it
        n_layers = len(hp['ae_decoding_n_channels'])
        gnum = 0
        for i in range(n_layers):
            if hp['ae_decoding_layer_type'][i] != 'convtranspose':
                continue
            self._unpool_before.append(i > 0 and hp['ae_decoding_layer_type'][i - 1] == 'unpool')
            if i > 0 and hp['ae_decoding_layer_type'][i - 1] == 'unpool':
                k = int(hp['ae_decoding_kernel_size'][i - 1])
                s = int(hp['ae_decoding_stride_size'][i - 1])
                self.decoder.add_module('maxunpool%i' % gnum, _mark_footprint(nn.MaxUnpool2d(
                    kernel_size=(k, k), stride=(s, s),
                    padding=(hp['ae_decoding_y_padding'][i - 1][0],
                             hp['ae_decoding_x_padding'][i - 1][0]))))
            args, crop, out_hw = self._get_convtranspose2d_args(i, gnum)
            is_last = i == n_layers - 1 and not hp['ae_decoding_last_FF_layer']
            if hp.get('fit_sess_io_layers', False) and is_last:
                name = 'convtranspose%i_sess_io_layers' % gnum
                self.decoder.add_module(name, nn.ModuleList([
                    _mark_footprint(nn.ConvTranspose2d(**args))
                    for _ in range(hp['n_datasets'])]))
                self.conv_t_pads[name] = self.conv_t_pads['convtranspose%i' % gnum]
            else:
                name = 'convtranspose%i' % gnum
                self.decoder.add_module(name, _mark_footprint(nn.ConvTranspose2d(**args)))
            if is_last:
                self.decoder.add_module('sigmoid%i' % gnum, nn.Sigmoid())
                act = _hip.ACT_SIGMOID
            else:
                if hp['ae_batch_norm']:
                    self.decoder.add_module('batchnorm%i' % gnum, nn.BatchNorm2d(
                        hp['ae_decoding_n_channels'][i],
                        momentum=hp.get('ae_batch_norm_momentum', 0.1),
                        track_running_stats=hp.get('track_running_stats', True)))
                self.decoder.add_module('relu%i' % gnum, nn.LeakyReLU(0.05))
                act = _hip.ACT_LRELU

            if i == 0:
                hin, win = start[1], start[2]
            else:
                hin, win = hp['ae_decoding_y_dim'][i - 1], hp['ae_decoding_x_dim'][i - 1]
            k = args['kernel_size'][0]
            self._plan.append(ConvLayerPlan(
                'convT', args['in_channels'], hin, win, args['out_channels'], out_hw[0], out_hw[1],
                k, k, args['stride'][0], crop[0], crop[1], act))
            self._layer_names.append(name)
            gnum += 1

        if hp['ae_decoding_last_FF_layer']:
            if hp.get('fit_sess_io_layers', False):
                raise NotImplementedError
            self.decoder.add_module('last_ff%i' % gnum, _mark_footprint(nn.Linear(
                hp['ae_decoding_x_dim'][-1] * hp['ae_decoding_y_dim'][-1] *
                hp['ae_decoding_n_channels'][-1],
                hp['ae_input_dim'][0] * hp['ae_input_dim'][1] * hp['ae_input_dim'][2])))
            self.decoder.add_module('sigmoid%i' % gnum, nn.Sigmoid())
            self._last_ff_name = 'last_ff%i' % gnum

    def _get_convtranspose2d_args(self, layer, global_layer):
        """nn.ConvTranspose2d kwargs, the (top, left) crop and the output size (ref :361-430).

        Full transposed-conv size is ``(in-1)*stride + k``; the reference either passes
        ``padding`` (symmetric), crops afterwards with ``F.pad(x, [-l,-r,-t,-b])`` (asymmetric),
        or extends with ``output_padding`` ('valid').  All three reduce to "output pixel (h, w)
        is full-size pixel (h+crop_t, w+crop_l), for h < Ho, w < Wo".
        """
        hp = self.hparams
        start = hp['ae_decoding_starting_dim']
        in_channels = start[0] if layer == 0 else hp['ae_decoding_n_channels'][layer - 1]
        k = hp['ae_decoding_kernel_size'][layer]
        s = hp['ae_decoding_stride_size'][layer]
        x0, x1 = hp['ae_decoding_x_padding'][layer]
        y0, y1 = hp['ae_decoding_y_padding'][layer]
        in_y = start[1] if layer == 0 else hp['ae_decoding_y_dim'][layer - 1]
        in_x = start[2] if layer == 0 else hp['ae_decoding_x_dim'][layer - 1]
        name = 'convtranspose%i' % global_layer
        if hp['ae_padding_type'] == 'valid':
            out_pad = (hp['ae_decoding_y_dim'][layer] - ((in_y - 1) * s + k),
                       hp['ae_decoding_x_dim'][layer] - ((in_x - 1) * s + k))
            in_pad = (y0, x0)
            self.conv_t_pads[name] = None
            crop = (y0, x0)
            out_hw = ((in_y - 1) * s + k - 2 * y0 + out_pad[0],
                      (in_x - 1) * s + k - 2 * x0 + out_pad[1])
        elif hp['ae_padding_type'] == 'same':
            out_pad = 0
            if x0 == x1 and y0 == y1:
                in_pad = (y0, x0)
                self.conv_t_pads[name] = None
            else:
                in_pad = 0
                self.conv_t_pads[name] = [x0, x1, y0, y1]
            crop = (y0, x0)
            out_hw = ((in_y - 1) * s + k - y0 - y1, (in_x - 1) * s + k - x0 - x1)
        else:
            raise ValueError('"%s" is not a valid padding type' % hp['ae_padding_type'])
        args = {
            'in_channels': in_channels,
            'out_channels': hp['ae_decoding_n_channels'][layer],
            'kernel_size': (k, k),
            'stride': (s, s),
            'padding': in_pad,
            'output_padding': out_pad}
        return args, crop, out_hw

    def _stack_params(self, dataset):
        params = []
        for name in self._layer_names:
            mod = getattr(self.decoder, name)
            if isinstance(mod, nn.ModuleList):
                mod = mod[dataset]
            params += [mod.weight, mod.bias]
        return params

    def forward(self, x, pool_idx=None, target_output_size=None, dataset=None, pixel_loss=None):
        """-> x_hat; or, with ``pixel_loss = {'target', 'mask', 'bounds', 'kind', 'want_xhat'}``
        (the single-pass training schedule), a :class:`FusedPixelLoss`: the per-chunk pixel loss
        evaluated in the epilogue of the last transposed convolution (x_hat is then not written
        unless asked for)."""
        hp = self.hparams
        start = hp['ae_decoding_starting_dim']
        h = linear(x, self.FF.weight, self.FF.bias)
        h = h.view(h.size(0), start[0], start[1], start[2])
        params = self._stack_params(dataset)
        if pixel_loss is not None:
            target, bounds, kind = pixel_loss['target'], pixel_loss['bounds'], pixel_loss['kind']
            # (chunk_sizes: the GLOBAL chunk lengths when the frames are sharded over ranks)
            scales = pixel_loss_scales(kind, bounds, target[0].numel(),
                                       pixel_loss.get('chunk_sizes'))
            if not any(self._unpool_before) and not hp['ae_batch_norm'] and \
                    not hp['ae_decoding_last_FF_layer'] and \
                    os.environ.get('BN_FUSED_LOSS', '1') != '0':
                terms, x_hat = conv_stack_sq_err(
                    self._plan, h, params, target, pixel_loss.get('mask'), bounds, scales,
                    pixel_loss.get('want_xhat', False))
                return FusedPixelLoss(x_hat, terms, kind, bounds)
            x_hat = self.forward(x, pool_idx, target_output_size, dataset=dataset)
            return FusedPixelLoss(x_hat, chunked_sq_err(
                x_hat, target, pixel_loss.get('mask'), bounds, scales), kind, bounds)
        if any(self._unpool_before):
            # max-pooling architectures: MaxUnpool2d with the encoder's indices (last pooled first)
            # in front of its transposed convolution, layer by layer (ref aes.py:460-476)
            pool_idx = list(pool_idx) if pool_idx is not None else []
            sizes = list(target_output_size) if target_output_size is not None else []
            bns = _bn_modules(self.decoder, self._layer_names)
            for j, layer in enumerate(self._plan):
                if self._unpool_before[j]:
                    idx = pool_idx.pop(-1)
                    outsize = sizes.pop(-1)
                    h = max_unpool(h, idx, (outsize[2], outsize[3]))
                if bns[j] is not None:
                    h = conv_stack_bn([layer], h, params[2 * j:2 * j + 2], [bns[j]])
                else:
                    h = conv_stack([layer], h, params[2 * j:2 * j + 2])
                _tap_signs(self._plan, j, layer, h)
        elif hp['ae_batch_norm']:
            h = conv_stack_bn(self._plan, h, params, _bn_modules(self.decoder, self._layer_names))
        else:
            h = conv_stack(self._plan, h, params)
        if hp['ae_decoding_last_FF_layer']:
            # dense last layer + Sigmoid (ref aes.py:345-359,478-486)
            ff = getattr(self.decoder, self._last_ff_name)
            h = activation(linear(h.reshape(h.size(0), -1), ff.weight, ff.bias),
                           _hip.ACT_SIGMOID)
            h = h.view(-1, hp['ae_input_dim'][0], hp['ae_input_dim'][1], hp['ae_input_dim'][2])
        return h


class LinearAEEncoder(BaseModule):
    """Single dense layer encoder (ref aes.py:491-544)."""

    def __init__(self, n_latents, input_size):
        super().__init__()
        self.n_latents = n_latents
        self.input_size = input_size
        self.encoder = None
        self.decoder = None
        self.build_model()

    def __str__(self):
        return 'Encoder architecture:\n    {}\n'.format(self.encoder)

    def build_model(self):
        self.encoder = nn.Linear(
            out_features=self.n_latents, in_features=int(np.prod(self.input_size)), bias=True)

    def forward(self, x, dataset=None):
        x = x.reshape(x.size(0), -1)
        return linear(x, self.encoder.weight, self.encoder.bias), None, None


class LinearAEDecoder(BaseModule):
    """Dense decoder, optionally tied to the encoder weights (ref aes.py:547-613)."""

    def __init__(self, n_latents, output_size, encoder=None):
        super().__init__()
        self.n_latents = n_latents
        self.output_size = output_size
        self.encoder = encoder
        self.decoder = None
        self.build_model()

    def __str__(self):
        out = 'Decoder architecture:\n'
        if self.bias is not None:
            out += '    Encoder weights transposed (plus independent bias)\n'
        else:
            out += '    {}\n'.format(self.decoder)
        return out

    def build_model(self):
        if self.encoder is None:
            self.decoder = nn.Linear(
                out_features=int(np.prod(self.output_size)), in_features=self.n_latents, bias=True)
        else:
            self.bias = nn.Parameter(
                torch.zeros(int(np.prod(self.output_size))), requires_grad=True)

    def forward(self, x, dataset=None):
        if self.encoder is None:
            x = linear(x, self.decoder.weight, self.decoder.bias)
        else:
            # tied weights: y = x W_enc + b   (W_enc is (n_latents, n_pixels))
            x = linear(x, self.encoder.encoder.weight.t().contiguous(), self.bias)
        return x.view(x.size(0), *self.output_size)


class AE(BaseModel):
    """Base autoencoder class (ref aes.py:616-773)."""

    _whole_batch = True      # loss() implements the single-pass schedule (see _loss_whole_batch)

    # fitting/graph_step.py: ``loss`` can be recorded into a HIP graph on the single-pass path
    # (its host tail goes through hip_functions.finish_loss); subclasses with their own ``loss``
    # opt in one by one
    graph_capturable = True
    graph_epoch_dependent = False

    def graph_capturable_for(self, x):
        # (batch norm with momentum=None reads its batch counter on the host for the cumulative
        # average factor: not recordable.  A frame-sharded step of REAL ranks is recordable since
        # round 5 -- its one collective, the per-chunk loss table, is issued behind the replay --
        # unless a rank could end up without frames: that rank takes another branch than the ones
        # that record, so the decision is taken from what every rank knows, the batch size)
        from behavenet_amd.fitting import distributed as bdist
        if bdist.frames_sharded() and bdist._emulated is None and \
                x.shape[0] < bdist.world_size():
            return False
        return self._whole_batch_ok(x) and not (
            self.hparams.get('ae_batch_norm', False) and
            self.hparams.get('ae_batch_norm_momentum', 0.1) is None)

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.model_type = self.hparams['model_type']
        self.img_size = (
            self.hparams['n_input_channels'],
            self.hparams['y_pixels'],
            self.hparams['x_pixels'])
        self.encoding = None
        self.decoding = None
        self.build_model()

    def __str__(self):
        out = '\nAutoencoder architecture\n'
        out += '------------------------\n'
        out += self.encoding.__str__()
        out += self.decoding.__str__()
        out += '\n'
        return out

    def build_model(self):
        hp = self.hparams
        hp['hidden_layer_size'] = hp['n_ae_latents']
        kind = self.model_type
        if kind == 'conv':
            # (encoder first: the order the seeded generator is consumed in, ref aes.py:680-693)
            self.encoding, self.decoding = ConvAEEncoder(hp), ConvAEDecoder(hp)
            return
        if kind != 'linear':
            raise ValueError('"%s" is an invalid model_type' % kind)
        if hp.get('fit_sess_io_layers', False):
            raise NotImplementedError
        # tied weights: the decoder shares the encoder's matrix
        self.encoding = LinearAEEncoder(hp['n_ae_latents'], self.img_size)
        self.decoding = LinearAEDecoder(hp['n_ae_latents'], self.img_size, self.encoding)

    def _reserve_pools(self, x):
        """First call (or a larger batch than seen so far): pre-grow the allocator pools."""
        if x.shape[0] > getattr(self, '_reserved_frames', 0) and x.is_cuda:
            reserve_device_pools(self, x.shape[0], x.device)
            self._reserved_frames = int(x.shape[0])

    def _prepare_first_layer(self, x, dataset):
        if self.model_type == 'conv' and x.shape[0] > 0 and \
                not (self.hparams['model_class'] == 'cond-ae' and
                     self.hparams.get('conditional_encoder', False)):
            self.encoding.prepare_first_layer(x, dataset)

    def _release_first_layer(self):
        if self.model_type == 'conv':
            self.encoding.release_first_layer()

    def _whole_batch_ok(self, x):
        """Frames are independent through this model (no batch norm): the chunks of the reference
        only bound ITS memory use, and only the loss normalisation depends on them."""
        # (batch norm: its statistics are per chunk -- hip_functions.bn_chunks takes them chunk by
        # chunk inside the one pass; under frame sharding they also span ranks: _pass_groups)
        return self.model_type == 'conv' and x.is_cuda and self._whole_batch and \
            not (self.hparams.get('ae_batch_norm', False) and bdist.frames_sharded()) and \
            os.environ.get('BN_WHOLE_BATCH', '1') != '0'

    def _pass_groups(self, x, chunk_size):
        """Frame ranges that are run as ONE forward / backward pass each, or None for the
        chunk-by-chunk pipelines: the whole batch when frames are independent through the
        model; with batch norm under frame-sharded data parallelism every 200-frame chunk is
        its own pass (the statistics are per chunk, taken over all ranks' frames:
        hip_functions.BatchNormActFn), so that the sharding machinery of the single-pass
        schedule (global chunk means, shared eps, gathered decomposed KL) serves it too."""
        n = x.shape[0]
        if self._whole_batch_ok(x):
            return [(0, n)]
        if bdist.frames_sharded() and self.model_type == 'conv' and x.is_cuda and self._whole_batch:
            return [(beg, min(beg + chunk_size, n)) for beg in range(0, n, chunk_size)]
        return None

    @staticmethod
    def _local_frames(local, *tensors):
        """This rank's frames of a frame-sharded batch: the local slice of every chunk, packed
        (-> packed tensors, the chunks' [beg, end) in the packed order)."""
        packed_bounds, pos = [], 0
        for b, e in local:
            packed_bounds.append((pos, pos + e - b))
            pos += e - b
        out = []
        for t in tensors:
            out.append(None if t is None else
                       torch.cat([t[b:e] for b, e in local], dim=0).contiguous())
        return out, packed_bounds

    def _loss_whole_batch(self, x, m, dataset, accumulate_grad, chunk_size, **fwd_kwargs):
        """ONE forward and ONE backward pass over the whole batch with the reference's per-chunk
        loss normalisation (ref aes.py:748-771): the gradient is the same
        sum_chunks grad(mean_chunk) and the reported loss the same frame-weighted mean, but
        every kernel sees all frames at once (256 frames on 256 CUs) and there is one set of
        weight-gradient launches and partial sums per step instead of one per chunk.

        Frame-sharded data parallelism (fitting/distributed.py, 'frames' mode): this rank runs
        its slice of every chunk, every chunk term is normalised by the GLOBAL chunk length, and
        the chunk losses are summed over ranks -- so the gradients all-reduced before the
        optimizer step, and the returned loss, are those of the single-device step."""
        batch_size = x.shape[0]
        bounds, local, sizes = bdist.shard_chunks(batch_size, chunk_size)
        if local != bounds:
            keys = [k for k in ('labels', 'labels_2d') if fwd_kwargs.get(k) is not None]
            (x, m, *rest), bounds_l = self._local_frames(local, x, m,
                                                         *[fwd_kwargs[k] for k in keys])
            fwd_kwargs = dict(fwd_kwargs, **dict(zip(keys, rest)))
        else:
            bounds_l = bounds
        self._reserve_pools(x)
        if x.shape[0] > 0:
            with torch.set_grad_enabled(bool(accumulate_grad)):
                # the pixel loss rides in the epilogue of the last decoder layer
                with bn_chunks(bounds_l):
                    x_hat, _ = self.forward(
                        x, dataset=dataset,
                        pixel_loss={'target': x, 'mask': m, 'bounds': bounds_l, 'kind': 'mse',
                                    'chunk_sizes': sizes},
                        **fwd_kwargs)
                chunk_losses = losses.mse_chunks(x, x_hat, m, bounds_l, sizes)
        else:       # more ranks than frames: nothing local, but the collectives still line up
            chunk_losses = None
        sharded = local != bounds
        local_vals = chunk_losses.detach() if chunk_losses is not None else \
            torch.zeros(len(bounds), device=x.device)
        # one device: the read-back of the chunk terms is queued in front of the backward pass
        # (a caller that looks at the value waits for the forward pass only)
        vals = None if sharded else Readback(local_vals)
        if accumulate_grad and chunk_losses is not None:
            backward_chunks([chunk_losses], single_pass=True)
        join_side_streams()
        if sharded:
            # The ranks' chunk terms are summed by ONE collective BEHIND the backward pass (it used
            # to sit between the forward and the backward pass: a launch gap on every rank, and a
            # step that could not be recorded into a HIP graph).  Under capture the collective is
            # left to the caller: graph_step.GraphedLoss issues it after every replay on the
            # recorded tensor (DeferredLoss.reduce_over_ranks) -- a recorded step contains no
            # collective at all.
            vals = Readback(local_vals if capturing() else bdist.all_reduce_(local_vals.clone()))

        def to_dict(v):
            v = v.astype(np.float64)
            return {'loss': float(np.sum(v * np.asarray(sizes, dtype=np.float64)) / batch_size)}
        return finish_loss([vals], to_dict, reduce_over_ranks=sharded)

    def _chunk_streams_ok(self):
        """Chunks may run on two HIP streams unless a layer accumulates outside the weight-
        gradient side stream (batch-norm scale/shift gradients and running statistics) ..."""
        if self.hparams.get('ae_batch_norm', False):
            return False
        # ... and only when every gradient is accumulated in place (param.grad exists, as under
        # FlatAdamAMSGrad): a first-touch `param.grad = dw` by autograd would not be ordered
        # against the side stream's in-place adds of the other chunk
        return all(p.grad is not None for p in self.parameters() if p.requires_grad)

    def forward(self, x, dataset=None, **kwargs):
        """-> (x_hat (N,C,H,W), latents (N,n_latents))."""
        if self.model_type == 'conv':
            z, pool_idx, outsize = self.encoding(x, dataset=dataset)
            y = self.decoding(z, pool_idx, outsize, dataset=dataset,
                              pixel_loss=kwargs.get('pixel_loss'))
        elif self.model_type == 'linear':
            z, _, _ = self.encoding(x)
            y = self.decoding(z)
        else:
            raise ValueError('"%s" is an invalid model_type' % self.model_type)
        return y, z

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        """Pixel MSE over 200-frame chunks with one backward per chunk (ref aes.py:722-773).

        Each chunk's loss is the mean over *that chunk*, so the accumulated gradient is
        sum_chunks grad(mean_chunk) exactly as in the reference; the returned
        value is the frame-weighted mean (G3).  Losses stay on the device until the end of the
        call: one host synchronisation per ``loss()`` instead of one per chunk.
        """
        x = data['images'][0]
        m = frame_masks(data, x)
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        if self._whole_batch_ok(x):
            return self._loss_whole_batch(x, m, dataset, accumulate_grad, chunk_size)

        # chunk loop (batch-norm models: the statistics are per chunk).  Frame-sharded data
        # parallelism: this rank runs its slice of every chunk, the batch-norm layers take their
        # statistics over all ranks' frames (hip_functions.BatchNormActFn), the chunk loss is this
        # rank's part of the global chunk mean.
        bounds, local, csizes = bdist.shard_chunks(batch_size, chunk_size)
        sharded = local != bounds
        vals, sizes, deferred = ChunkScalars(), [], []
        self._reserve_pools(x)
        if not sharded:
            self._prepare_first_layer(x, dataset)
        begin_chunks(x.device)
        for chunk, ((beg, end), n_c) in enumerate(zip(local, csizes)):
            if end == beg:
                raise NotImplementedError(
                    'a chunk of %d frames cannot be sharded over %d ranks' % (
                        n_c, bdist.shard_rank_world()[1]))
            x_in = x[beg:end]
            m_in = m[beg:end] if m is not None else None
            with chunk_stream(chunk, x.device, self._chunk_streams_ok() and not sharded):
                with torch.set_grad_enabled(bool(accumulate_grad)):
                    x_hat, _ = self.forward(x_in, dataset=dataset)
                    loss = losses.mse(x_in, x_hat, m_in)
                    if sharded:
                        loss = loss * ((end - beg) / float(n_c))
                vals.add(loss.detach().reshape(1))
            if accumulate_grad:
                deferred.append(loss)
            sizes.append(n_c)
        # the loss values only need the forwards: their read-back is enqueued before the
        # (deferred) backwards and waited for after every backward launch is queued
        vals = vals.finish(deferred)[:, 0]
        if sharded:
            vals = np.asarray(bdist.all_reduce_scalars(vals.tolist()))
        self._release_first_layer()
        loss_val = float(np.sum(vals * np.asarray(sizes, dtype=np.float64)) / batch_size)
        return {'loss': loss_val}


class ConditionalAE(AE):
    """Conditional autoencoder: labels are appended to the latents (ref aes.py:776-898)."""

    def __init__(self, hparams):
        if hparams['model_type'] == 'linear':
            raise NotImplementedError
        super().__init__(hparams)

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents'] + self.hparams['n_labels']
        self.encoding = ConvAEEncoder(self.hparams)
        self.decoding = ConvAEDecoder(self.hparams)

    def forward(self, x, dataset=None, labels=None, labels_2d=None, **kwargs):
        if self.hparams.get('conditional_encoder', False):
            x = torch.cat((x, labels_2d), dim=1)
        z, pool_idx, outsize = self.encoding(x, dataset=dataset)
        z_aug = torch.cat((z, labels), dim=1)
        y = self.decoding(z_aug, pool_idx, outsize, dataset=dataset,
                          pixel_loss=kwargs.get('pixel_loss'))
        return y, z

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x = data['images'][0]
        y = data['labels'][0]
        m = frame_masks(data, x)
        labels_2d = data['labels_sc'][0] if self.hparams.get('conditional_encoder', False) \
            else None
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        groups = self._pass_groups(x, chunk_size)
        if groups is not None:
            # labels ride along frame by frame: same single-pass schedule as AE.loss (batch norm
            # under frame sharding: one pass per chunk, see _pass_groups)
            total = 0.0
            for gb, ge in groups:
                part = self._loss_whole_batch(
                    x[gb:ge], m[gb:ge] if m is not None else None, dataset, accumulate_grad,
                    chunk_size, labels=y[gb:ge],
                    labels_2d=labels_2d[gb:ge] if labels_2d is not None else None)
                if len(groups) == 1:
                    return part
                total += part['loss'] * (ge - gb)
            return {'loss': float(total / batch_size)}
        _no_sharded_chunk_loop(self)
        vals, sizes, deferred = ChunkScalars(), [], []
        self._reserve_pools(x)
        self._prepare_first_layer(x, dataset)
        begin_chunks(x.device)
        for chunk in range(n_chunks):
            beg = chunk * chunk_size
            end = min((chunk + 1) * chunk_size, batch_size)
            x_in, y_in = x[beg:end], y[beg:end]
            m_in = m[beg:end] if m is not None else None
            l2d = labels_2d[beg:end] if labels_2d is not None else None
            with chunk_stream(chunk, x.device, self._chunk_streams_ok()):
                with torch.set_grad_enabled(bool(accumulate_grad)):
                    x_hat, _ = self.forward(x_in, dataset=dataset, labels=y_in, labels_2d=l2d)
                    loss = losses.mse(x_in, x_hat, m_in)
                vals.add(loss.detach().reshape(1))
            if accumulate_grad:
                deferred.append(loss)
            sizes.append(end - beg)
        # the loss values only need the forwards: their read-back is enqueued before the
        # (deferred) backwards and waited for after every backward launch is queued
        vals = vals.finish(deferred)[:, 0]
        self._release_first_layer()
        loss_val = float(np.sum(vals * np.asarray(sizes, dtype=np.float64)) / batch_size)
        return {'loss': loss_val}


class AEMSP(AE):
    """Autoencoder with matrix subspace projection (ref aes.py:901-1217, Li et al. 2019).

    ``y_hat = P z`` (no bias) predicts the labels from the latents; the MSP loss
    ``mse(y, y_hat) + mse(z, y_hat P)`` pushes the label information into the row space of P.
    ``model_class = 'cond-ae-msp'``; conv encoder/decoder only.
    """

    graph_capturable = False

    def __init__(self, hparams):
        n_lat, n_lab = hparams['n_ae_latents'], hparams['n_labels']
        if hparams['model_type'] == 'linear':
            raise NotImplementedError
        if n_lab > n_lat:
            raise ValueError('AEMSP model must contain at least as many latents as labels')
        # P (`projection`: latents -> labels) and U (`U`: latents <-> transformed latents) are made by
        # build_model, which the base constructor calls; nn.Module wants the names to exist before that
        self.n_latents, self.n_labels = n_lat, n_lab
        self.projection = self.U = None
        super().__init__(hparams)

    def build_model(self):
        hp = self.hparams
        hp['hidden_layer_size'] = hp['n_ae_latents']
        # construction order = the order the seeded generator is consumed in (ref :943-950): conv encoder,
        # conv decoder, P, then U -- U only so that its key is in the state dict from the start
        # (create_orthogonal_matrix replaces its weight)
        self.encoding, self.decoding = ConvAEEncoder(hp), ConvAEDecoder(hp)
        heads = [nn.Linear(self.n_latents, n_out, bias=False) for n_out in (self.n_labels, self.n_latents)]
        self.projection, self.U = heads

    def _chunk_streams_ok(self):
        return False     # `projection` is used twice per chunk and accumulates through torch

    def forward(self, x, dataset=None, **kwargs):
        """-> (x_hat, z, y_hat)."""
        z, pool_idx, outsize = self.encoding(x, dataset=dataset)
        y = linear(z, self._P(), None)
        x_hat = self.decoding(z, pool_idx, outsize, dataset=dataset)
        return x_hat, z, y

    def _P(self):
        # a non-leaf alias of the projection: it enters two GEMMs per chunk, and both gradients
        # must reach `projection.weight.grad` through ONE torch accumulation, not through the
        # kernels' in-place side-stream path and torch's AccumulateGrad at the same time
        return self.projection.weight.clone()

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        """mse + msp.alpha * (mse(y, y_hat) + mse(z, y_hat P)) per 200-frame chunk (ref :979-1060);
        returns ``loss, loss_mse, loss_msp, labels_r2``."""
        from behavenet_amd.models.vaes import _r2_variance_weighted
        x = data['images'][0]
        y = data['labels'][0]
        m = frame_masks(data, x)
        batch_size = x.shape[0]
        n_chunks = int(np.ceil(batch_size / chunk_size))
        alpha = self.hparams['msp.alpha']
        sharded = bdist.frames_sharded()
        if sharded and (self.hparams.get('ae_batch_norm', False) or not self._whole_batch or
                        self.model_type != 'conv' or not x.is_cuda):
            # (every rank takes this branch together: it depends on hparams only)
            raise NotImplementedError('frame-sharded data parallelism of AEMSP serves the '
                                      'single-pass conv model without batch norm '
                                      '(use dp_shard="trial")')
        self._reserve_pools(x)
        if self._whole_batch_ok(x) or sharded:
            # single pass (see AE._loss_whole_batch): every map here is frame-wise, the two MSP
            # terms are means over a chunk's rows like the pixel term.  Frame-sharded (round 4):
            # this rank's slice of every chunk, every chunk term divided by the GLOBAL chunk
            # length (the pixel term through `chunk_sizes`, the MSP terms through the rank's share
            # of the chunk), the table summed over ranks; labels_r2 over all ranks' rows.
            from behavenet_amd.models.vaes import _FrameShards
            sh = _FrameShards(batch_size, chunk_size)
            x_l, y_l, m_l = sh.take(x, y, m)
            bounds = sh.bounds_l
            with torch.set_grad_enabled(bool(accumulate_grad)), bn_chunks(bounds):
                P = self._P()
                z, pool_idx, outsize = self.encoding(x_l, dataset=dataset)
                y_hat = linear(z, P, None)
                x_hat = self.decoding(
                    z, pool_idx, outsize, dataset=dataset,
                    pixel_loss={'target': x_l, 'mask': m_l, 'bounds': bounds, 'kind': 'mse',
                                'chunk_sizes': sh.sizes})
                z_back = linear(y_hat, P.t().contiguous(), None)
                l_mse = losses.mse_chunks(x_l, x_hat, m_l, bounds, sh.sizes)
                l_msp = sh.row_terms(
                    lambda yy, yh, zz, zb: losses.mse(yy, yh) + losses.mse(zz, zb),
                    y_l, y_hat, z, z_back)
                lossv = l_mse + float(alpha) * l_msp
            table_t = torch.stack([lossv.detach(), l_mse.detach(), l_msp.detach()], dim=1)
            if sh.sharded:
                table_t = bdist.all_reduce_(table_t.clone())
            table = Readback(table_t)
            # (labels_r2 is a metric over the whole batch: all ranks' rows, in the same -- rank-major
            # -- order for both; an emulated rank has only its own)
            rows = sh.all_rows if bdist._emulated is None else (lambda t: t)
            y_hat_rb, y_rb = Readback(rows(y_hat.detach())), Readback(rows(y_l))
            if accumulate_grad:
                backward_chunks([lossv], single_pass=True)
            join_side_streams()
            vals = table.numpy().astype(np.float64)
            w = np.asarray(sh.sizes, dtype=np.float64)[:, None]
            tot = (vals * w).sum(axis=0) / batch_size
            r2 = _r2_variance_weighted(y_rb.numpy(), y_hat_rb.numpy())
            return {'loss': float(tot[0]), 'loss_mse': float(tot[1]), 'loss_msp': float(tot[2]),
                    'labels_r2': r2}
        _no_sharded_chunk_loop(self)
        self._prepare_first_layer(x, dataset)
        rbs, sizes, deferred, y_hat_all = ChunkScalars(), [], [], []
        for chunk in range(n_chunks):
            beg = chunk * chunk_size
            end = min((chunk + 1) * chunk_size, batch_size)
            x_in, y_in = x[beg:end], y[beg:end]
            m_in = m[beg:end] if m is not None else None
            with torch.set_grad_enabled(bool(accumulate_grad)):
                P = self._P()
                z, pool_idx, outsize = self.encoding(x_in, dataset=dataset)
                y_hat = linear(z, P, None)
                x_hat = self.decoding(z, pool_idx, outsize, dataset=dataset)
                loss_mse = losses.mse(x_in, x_hat, m_in)
                # y_hat P: (N, n_labels) x (n_labels, n_latents); nn.Linear stores P as
                # (n_labels, n_latents), so the "transposed layer" is P itself (ref :1035-1037)
                z_back = linear(y_hat, P.t().contiguous(), None)
                loss_msp = losses.mse(y_in, y_hat) + losses.mse(z, z_back)
                loss = loss_mse + float(alpha) * loss_msp
            if accumulate_grad:
                deferred.append(loss)
            rbs.add(torch.stack([loss.detach(), loss_mse.detach(), loss_msp.detach()]))
            sizes.append(end - beg)
            y_hat_all.append(y_hat.detach())
        y_hat_rb = Readback(torch.cat(y_hat_all, dim=0))
        y_rb = Readback(y)
        vals = rbs.finish(deferred)
        self._release_first_layer()
        w = np.asarray(sizes, dtype=np.float64)[:, None]
        tot = (vals * w).sum(axis=0) / batch_size
        r2 = _r2_variance_weighted(y_rb.numpy(), y_hat_rb.numpy())
        return {'loss': float(tot[0]), 'loss_mse': float(tot[1]), 'loss_msp': float(tot[2]),
                'labels_r2': r2}

    def save(self, filepath):
        """Build the complete orthogonal matrix U before saving (ref :1062-1065)."""
        self.create_orthogonal_matrix()
        super().save(filepath)

    def create_orthogonal_matrix(self):
        """U = the rows of P stacked on an orthonormal basis of P's null space (ref :1067-1080; scipy's
        ``null_space`` on the float32 matrix, as there, so that the basis -- any would do -- is the same one)."""
        from scipy.linalg import null_space
        P = self.projection.weight.detach().cpu().numpy()
        complement = null_space(P).T                           # (n_latents - n_labels, n_latents)
        full = torch.from_numpy(np.vstack((P, complement))).to(torch.float32)
        # (next to P, wherever the model was moved; the reference sends it to hparams['device'])
        self.U.weight = nn.Parameter(full.to(self.projection.weight.device), requires_grad=False)

    def get_transformed_latents(self, inputs, dataset=None, as_numpy=True):
        """U z (ref :1082-1122) of latents (2-d input) or of the latents of images (anything else)."""
        t = inputs if torch.is_tensor(inputs) else torch.Tensor(inputs)
        z = t if t.dim() == 2 else self.encoding(t, dataset=dataset)[0]
        out = linear(z, self.U.weight, None)
        return out.detach().cpu().numpy() if as_numpy else out

    def get_inverse_transformed_latents(self, latents, as_numpy=True):
        """Back to the original latent space: latents U (ref :1124-1146)."""
        if not isinstance(latents, torch.Tensor):
            latents = torch.Tensor(latents)
        latents_og = linear(latents, self.U.weight.t().contiguous(), None)
        return latents_og.cpu().detach().numpy() if as_numpy else latents_og

    def sample(self, x=None, dataset=None, latents=None, labels=None, labels_2d=None):
        """Decode user-defined labels and/or transformed latents (ref :1148-1217)."""
        if latents is None or labels is None:
            latents_tr = self.get_transformed_latents(x, dataset)
        else:
            latents_tr = np.full(shape=(latents.shape[0], self.n_latents), fill_value=np.nan)
        if labels is not None:
            latents_tr[:, :self.n_labels] = labels
        if latents is not None:
            latents_tr[:, self.n_labels:] = latents
        dev = self.U.weight.device
        latents_og = self.get_inverse_transformed_latents(
            torch.from_numpy(latents_tr).float().to(dev), as_numpy=False)
        return self.decoding(latents_og, None, None, dataset=dataset)


def load_pretrained_ae(model, hparams):
    """Initialise ``model`` from ``hparams['pretrained_weights_path']`` (ref aes.py:1220-1274).

    Tensors whose shapes differ (typically the FF layers when the latent count changed) are
    dropped and keep their fresh initialisation.
    """
    path = hparams.get('pretrained_weights_path', None)
    if path is None or path is False or path == '':
        print('Initializing with random weights')
        return model
    if hparams['model_type'] == 'linear':
        raise NotImplementedError('Loading pretrained weights with linear AE')
    print('Loading pretrained weights')
    loaded = torch.load(path, map_location=lambda storage, loc: storage)
    if loaded['encoding.FF.weight'].shape != model.encoding.FF.weight.shape:
        print('PRETRAINED MODEL HAS DIFFERENT SPATIAL DIMENSIONS OR N LATENTS: '
              'NOT LOADING FF PARAMETERS')
        for key in ('encoding.FF.weight', 'encoding.FF.bias',
                    'decoding.FF.weight', 'decoding.FF.bias'):
            loaded.pop(key, None)
    model.load_state_dict(loaded, strict=False)
    return model
