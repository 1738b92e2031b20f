"""Image decoders driven by behavioural labels (ref behavenet/models/decoders.py:355-496).

Only ``ConvDecoder`` lives on the conv-autoencoder hot path: it is the AE's ``ConvAEDecoder``
stack fed with the labels instead of latents.  The neural-activity decoders of the reference's
``decoders.py`` (``Decoder``, ``MLP``, ``LSTM``) are out of scope.
"""

import numpy as np
import torch

from behavenet_amd.fitting import losses
from behavenet_amd.hip_functions import (
    Readback, backward_chunks, join_side_streams, reserve_device_pools, ChunkScalars)
from behavenet_amd.models.aes import frame_masks
from behavenet_amd.models.aes import ConvAEDecoder, LinearAEDecoder
from behavenet_amd.models.base import BaseModel

__all__ = ['ConvDecoder']


class ConvDecoder(BaseModel):
    """Decode images from labels with the convolutional decoder (ref decoders.py:355-496)."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.model_type = self.hparams['model_type']
        self.img_size = (
            self.hparams['n_input_channels'],
            self.hparams['y_pixels'],
            self.hparams['x_pixels'])
        self.decoding = None
        self.build_model()

    def __str__(self):
        out = '\nConvolutional decoder architecture\n'
        out += '------------------------\n'
        out += self.decoding.__str__()
        out += '\n'
        return out

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_labels']
        if self.model_type == 'conv':
            self.decoding = ConvAEDecoder(self.hparams)
        elif self.model_type == 'linear':
            if self.hparams.get('fit_sess_io_layers', False):
                raise NotImplementedError
            self.decoding = LinearAEDecoder(self.hparams['n_labels'], self.img_size)
        else:
            raise ValueError('"%s" is an invalid model_type' % self.model_type)

    def forward(self, x, dataset=None, **kwargs):
        """labels (N, n_labels) -> images (N, C, H, W)  (a tensor, not a tuple: ref :431)."""
        if self.model_type == 'conv':
            return self.decoding(x, None, None, dataset=dataset)
        if self.model_type == 'linear':
            return self.decoding(x)
        raise ValueError('"%s" is an invalid model_type' % self.model_type)

    def _single_pass_ok(self, x):
        return self.model_type == 'conv' and x.is_cuda and \
            not self.hparams.get('ae_batch_norm', False)

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        """Pixel MSE with the reference's per-chunk normalisation (ref decoders.py:433-496)."""
        x = data['images'][0]
        y = data['labels'][0]
        m = frame_masks(data, data['images'][0])
        batch_size = x.shape[0]
        bounds = [(beg, min(beg + chunk_size, batch_size))
                  for beg in range(0, batch_size, chunk_size)]
        sizes = np.asarray([end - beg for beg, end in bounds], dtype=np.float64)
        if x.is_cuda and batch_size > getattr(self, '_reserved_frames', 0):
            reserve_device_pools(self, batch_size, x.device)
            self._reserved_frames = int(batch_size)
        if self._single_pass_ok(x):
            # frames are independent through the decoder: one forward / one backward over the
            # whole batch, loss normalised per chunk (same schedule as AE._loss_whole_batch)
            with torch.set_grad_enabled(bool(accumulate_grad)):
                x_hat = self.forward(y, dataset=dataset)
                chunk_losses = losses.mse_chunks(x, x_hat, m, bounds)
            vals = Readback(chunk_losses.detach())
            if accumulate_grad:
                backward_chunks([chunk_losses], single_pass=True)
            join_side_streams()
            vals = vals.numpy().astype(np.float64)
        else:
            scalars, deferred = ChunkScalars(), []
            for beg, end in bounds:
                m_in = m[beg:end] if m is not None else None
                with torch.set_grad_enabled(bool(accumulate_grad)):
                    x_hat = self.forward(y[beg:end], dataset=dataset)
                    loss = losses.mse(x[beg:end], x_hat, m_in)
                scalars.add(loss.detach().reshape(1))
                if accumulate_grad:
                    deferred.append(loss)
            vals = scalars.finish(deferred)[:, 0]
        return {'loss': float(np.sum(vals * sizes) / batch_size)}
