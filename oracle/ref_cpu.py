"""CPU oracle: a plain PyTorch-CPU restatement of BehaveNet's conv-autoencoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (behavenet_amd/) imports this file;
it is imported by tests/, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline``
leg, always as the checker / the timed CPU baseline, never as the thing shipped.

What it restates (reference file:line -> here):
  * ConvAEEncoder.forward   behavenet/models/aes.py:181-218   -> ConvEncoder.forward
  * ConvAEDecoder.forward   behavenet/models/aes.py:432-488   -> ConvDecoder.forward
  * AE / AE.loss            behavenet/models/aes.py:616-773   -> AE
  * ConditionalAE           behavenet/models/aes.py:776-898   -> ConditionalAE
  * AEMSP                   behavenet/models/aes.py:901-1060  -> AEMSP
  * reparameterize, VAE, ConditionalVAE, BetaTCVAE, PSVAE, ConvAEPSEncoder
                            behavenet/models/vaes.py:17-35,38-208,211-364,367-503,506-729,1276-1363
  * mse, gaussian_ll, gaussian_ll_to_mse, kl_div_to_std_normal, decomposed_kl
                            behavenet/fitting/losses.py:36-147,284-372
  * MSPSVAE, ConvAEMSPSEncoder behavenet/models/vaes.py:849-1098,1366-1470 -> MSPSVAE, ConvMSPSEncoder
  * triplet_loss            behavenet/fitting/losses.py:402-511 -> triplet_loss
  * ConvDecoder (labels -> images) behavenet/models/decoders.py:355-496 -> ConvDecoderModel
  * LinearAEEncoder/Decoder behavenet/models/aes.py:491-613   -> LinearEncoder / LinearDecoder
  * MaxPool2d / MaxUnpool2d architectures behavenet/models/aes.py:99-110,196-208,281-294,460-464
  * DiagLinear              behavenet/models/base.py:70-103
  * the SGD inner loop      behavenet/fitting/training.py:284-286,336-352 -> train_step / Adam

The arithmetic of the reference lives in the un-vendored third-party PyTorch (pinned
torch==1.3.1 in requirements.txt:15); like the reference, this file calls torch's CPU operators
(F.conv2d, F.conv_transpose2d, F.pad, F.linear, torch.optim.Adam).  PINNING: the oracle is
checked against golden vectors produced by importing the real reference in the build
container (tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py):
identical parameters from the same seed, identical outputs, losses, gradients and a 3-step
Adam(amsgrad) trajectory.  Modules are registered in the reference's order and under the
reference's names so ``torch.manual_seed(s); AE(hparams)`` draws the same initial weights and
``state_dict`` keys match.
"""

import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

LN2PI = np.log(2 * np.pi)
SLOPE = 0.05

# Tests only.  LeakyReLU is continuous but its slope is not: on noise frames about one
# pre-activation in 10^6 lies within fp32 rounding of zero, and two correct fp32 implementations
# may put it on different branches, which moves the gradients by ~1e-3.  A parity test can hand
# the oracle the branch pattern the implementation under test took (an object with
# ``take(stack, layer, x) -> bool tensor | None``): the forward value is unaffected (|x| ~ 1e-9
# there), and the gradients become comparable at rounding level.  The test then separately
# checks that the pattern differs from the oracle's own only at such ties.
LRELU_BRANCH = None


def _lrelu(x, stack, layer):
    if LRELU_BRANCH is not None:
        pos = LRELU_BRANCH.take(stack, layer, x)
        if pos is not None:
            return torch.where(pos, x, SLOPE * x)
    return F.leaky_relu(x, SLOPE)


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------
def mse(y_pred, y_true, masks=None):
    d = (y_pred - y_true) ** 2
    if masks is not None:
        d = d * masks
    return torch.mean(d)


def gaussian_ll(y_pred, y_mean, masks=None, std=1):
    n_dims = np.prod(y_pred.shape[1:])
    d = (y_pred - y_mean) ** 2
    if masks is not None:
        d = d * masks
    per_frame = d.sum(dim=tuple(range(1, y_pred.dim())))
    ll = -(0.5 * LN2PI + 0.5 * np.log(std ** 2)) * n_dims - (0.5 / std ** 2) * per_frame
    return torch.mean(ll)


def gaussian_ll_to_mse(ll, n_dims, gaussian_std=1, mse_std=1):
    out = np.copy(ll)
    out += (0.5 * LN2PI + 0.5 * np.log(gaussian_std ** 2)) * n_dims
    out *= -(gaussian_std ** 2) / 0.5
    out /= n_dims
    out *= 1.0 / (mse_std ** 2)
    return out


def kl_div_to_std_normal(mu, logvar):
    return torch.mean(0.5 * torch.sum(logvar.exp() - logvar + mu.pow(2) - 1, dim=1))


def _log_q_pairs(z, mu, logvar):
    zz, mm, lv = z[:, None], mu[None, :], logvar[None, :]
    return -0.5 * (torch.exp(-lv) * (zz - mm) ** 2 + lv + LN2PI)   # [j, i, l]


def decomposed_kl(z, mu, logvar):
    lq = _log_q_pairs(z, mu, logvar)
    joint = torch.sum(lq, dim=2)
    log_qz = torch.logsumexp(joint, dim=1)
    log_qz_cond = torch.diag(joint)
    log_qz_prod = torch.sum(torch.logsumexp(lq, dim=1), dim=1)
    log_pz_prod = torch.sum(-0.5 * (z ** 2 + LN2PI), dim=1)
    return (torch.mean(log_qz_cond - log_qz), torch.mean(log_qz - log_qz_prod),
            torch.mean(log_qz_prod - log_pz_prod))


def index_code_mi(z, mu, logvar):
    return decomposed_kl(z, mu, logvar)[0]


def total_correlation(z, mu, logvar):
    return decomposed_kl(z, mu, logvar)[1]


def dimension_wise_kl_to_std_normal(z, mu, logvar):
    return decomposed_kl(z, mu, logvar)[2]


def reparameterize(mu, logvar, eps=None):
    std = torch.exp(logvar)          # sic: not exp(0.5*logvar)
    if eps is None:
        eps = torch.randn_like(std)
    return eps.mul(std).add_(mu)


# ------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------
class DiagLinear(nn.Module):
    def __init__(self, features, bias=True):
        super().__init__()
        self.features = features
        self.weight = nn.Parameter(torch.empty(features))
        self.bias = nn.Parameter(torch.empty(features)) if bias else None
        bound = 1 / math.sqrt(features)
        nn.init.uniform_(self.weight, -bound, bound)
        if bias:
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        y = x * self.weight
        return y + self.bias if self.bias is not None else y


class ConvEncoder(nn.Module):
    """ZeroPad2d/Conv2d/[BatchNorm2d]/LeakyReLU stack + FF (+ logvar)."""

    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        self.encoder = nn.ModuleList()
        self.layers = []           # (name, pad(l,r,t,b) or None, bn_name or None)
        g = 0
        n = len(hp['ae_encoding_n_channels'])
        for i in range(n):
            if hp['ae_encoding_layer_type'][i] != 'conv':
                continue                      # 'maxpool' entries are attached to the conv before
            if i == 0:
                extra = 0
                if hp['model_class'] == 'cond-ae' and hp.get('conditional_encoder', False):
                    extra = int(hp['n_labels'] / 2)
                cin = hp['ae_input_dim'][0] + extra
            else:
                cin = hp['ae_encoding_n_channels'][i - 1]
            x0, x1 = hp['ae_encoding_x_padding'][i]
            y0, y1 = hp['ae_encoding_y_padding'][i]
            sym = (x0 == x1) and (y0 == y1)
            if not sym:
                self.encoder.add_module('zero_pad%i' % g, nn.ZeroPad2d((x0, x1, y0, y1)))
            mk = lambda: nn.Conv2d(                                             # noqa: E731
                cin, hp['ae_encoding_n_channels'][i], hp['ae_encoding_kernel_size'][i],
                stride=hp['ae_encoding_stride_size'][i], padding=(y0, x0) if sym else 0)
            if hp.get('fit_sess_io_layers', False) and i == 0:
                name = 'conv%i_sess_io_layers' % g
                self.encoder.add_module(name, nn.ModuleList(
                    [mk() for _ in range(hp['n_datasets'])]))
            else:
                name = 'conv%i' % g
                self.encoder.add_module(name, mk())
            bn = None
            if hp['ae_batch_norm']:
                bn = 'batchnorm%i' % g
                self.encoder.add_module(bn, nn.BatchNorm2d(
                    hp['ae_encoding_n_channels'][i],
                    momentum=hp.get('ae_batch_norm_momentum', 0.1),
                    track_running_stats=hp.get('track_running_stats', True)))
            pool = None
            if i < n - 1 and hp['ae_encoding_layer_type'][i + 1] == 'maxpool':
                # ref aes.py:99-110,165-179: kernel / stride / padding of the NEXT list entry
                pool = 'maxpool%i' % g
                self.encoder.add_module(pool, nn.MaxPool2d(
                    kernel_size=int(hp['ae_encoding_kernel_size'][i + 1]),
                    stride=int(hp['ae_encoding_stride_size'][i + 1]),
                    padding=(hp['ae_encoding_y_padding'][i + 1][0],
                             hp['ae_encoding_x_padding'][i + 1][0]),
                    return_indices=True, ceil_mode=hp['ae_padding_type'] != 'valid'))
            self.encoder.add_module('relu%i' % g, nn.LeakyReLU(SLOPE))
            self.layers.append((name, None if sym else (x0, x1, y0, y1), bn, pool))
            g += 1
        last = hp['ae_encoding_n_channels'][-1] * hp['ae_encoding_y_dim'][-1] * \
            hp['ae_encoding_x_dim'][-1]
        self.FF = nn.Linear(last, hp['n_ae_latents'])
        if hp.get('variational', False):
            self.logvar = nn.Linear(last, hp['n_ae_latents'])

    def features(self, x, dataset=None, taps=None):
        self.pool_idx, self.pool_sizes = [], []
        for li, (name, pad, bn, pool) in enumerate(self.layers):
            conv = getattr(self.encoder, name)
            if isinstance(conv, nn.ModuleList):
                conv = conv[dataset]
            if pad is not None:
                x = F.pad(x, pad)
            x = F.conv2d(x, conv.weight, conv.bias, stride=conv.stride, padding=conv.padding)
            if bn is not None:
                x = getattr(self.encoder, bn)(x)
            if pool is not None:                    # conv -> [bn] -> pool -> relu (ref :200-211)
                self.pool_sizes.append(x.size())
                x, idx = getattr(self.encoder, pool)(x)
                self.pool_idx.append(idx)
            x = _lrelu(x, 'encoding', li)
            if taps is not None:
                taps.append(x)
        return x.reshape(x.size(0), -1)

    def forward(self, x, dataset=None, taps=None):
        x1 = self.features(x, dataset, taps)
        if self.hp.get('variational', False):
            return F.linear(x1, self.FF.weight, self.FF.bias), \
                F.linear(x1, self.logvar.weight, self.logvar.bias), self.pool_idx, self.pool_sizes
        return F.linear(x1, self.FF.weight, self.FF.bias), self.pool_idx, self.pool_sizes


class ConvPSEncoder(ConvEncoder):
    def __init__(self, hp):
        super().__init__(hp)
        n_lat, n_lab = hp['n_ae_latents'], hp['n_labels']
        self.A = nn.Linear(n_lat, n_lab, bias=False)
        self.B = nn.Linear(n_lat, n_lat - n_lab, bias=False)
        self.D = DiagLinear(n_lab, bias=True)
        from scipy.stats import ortho_group
        m = ortho_group.rvs(dim=n_lat).astype('float32')
        with torch.no_grad():
            self.A.weight = nn.Parameter(torch.from_numpy(m[:n_lab, :]), requires_grad=False)
            self.B.weight = nn.Parameter(torch.from_numpy(m[n_lab:, :]), requires_grad=False)

    def forward(self, x, dataset=None, taps=None):
        x1 = self.features(x, dataset, taps)
        h = F.linear(x1, self.FF.weight, self.FF.bias)
        return F.linear(h, self.A.weight), F.linear(h, self.B.weight), \
            F.linear(x1, self.logvar.weight, self.logvar.bias), self.pool_idx, self.pool_sizes


class ConvDecoder(nn.Module):
    """FF + ConvTranspose2d/crop/[BatchNorm2d]/LeakyReLU stack, Sigmoid on the last layer."""

    def __init__(self, hp):
        super().__init__()
        self.hp = hp
        start = hp['ae_decoding_starting_dim']
        self.FF = nn.Linear(hp['hidden_layer_size'], start[0] * start[1] * start[2])
        self.decoder = nn.ModuleList()
        self.layers = []   # (name, crop or None, bn or None, is_last)
        last_ff = bool(hp['ae_decoding_last_FF_layer'])        # ref aes.py:326-330,345-359
        if last_ff and hp.get('fit_sess_io_layers', False):
            raise NotImplementedError
        n = len(hp['ae_decoding_n_channels'])
        g = 0
        for i in range(n):
            if hp['ae_decoding_layer_type'][i] != 'convtranspose':
                continue                      # 'unpool' entries precede the next convtranspose
            unpool = None
            if i > 0 and hp['ae_decoding_layer_type'][i - 1] == 'unpool':     # ref aes.py:281-294
                unpool = 'maxunpool%i' % g
                ku, su = int(hp['ae_decoding_kernel_size'][i - 1]), \
                    int(hp['ae_decoding_stride_size'][i - 1])
                self.decoder.add_module(unpool, nn.MaxUnpool2d(
                    kernel_size=(ku, ku), stride=(su, su),
                    padding=(hp['ae_decoding_y_padding'][i - 1][0],
                             hp['ae_decoding_x_padding'][i - 1][0])))
            cin = start[0] if i == 0 else hp['ae_decoding_n_channels'][i - 1]
            k, s = hp['ae_decoding_kernel_size'][i], hp['ae_decoding_stride_size'][i]
            x0, x1 = hp['ae_decoding_x_padding'][i]
            y0, y1 = hp['ae_decoding_y_padding'][i]
            in_y = start[1] if i == 0 else hp['ae_decoding_y_dim'][i - 1]
            in_x = start[2] if i == 0 else hp['ae_decoding_x_dim'][i - 1]
            crop = None
            if hp['ae_padding_type'] == 'valid':
                pad = (y0, x0)
                opad = (hp['ae_decoding_y_dim'][i] - ((in_y - 1) * s + k),
                        hp['ae_decoding_x_dim'][i] - ((in_x - 1) * s + k))
            elif hp['ae_padding_type'] == 'same':
                opad = 0
                if x0 == x1 and y0 == y1:
                    pad = (y0, x0)
                else:
                    pad = 0
                    crop = [x0, x1, y0, y1]
            else:
                raise ValueError('"%s" is not a valid padding type' % hp['ae_padding_type'])
            mk = lambda: nn.ConvTranspose2d(                                     # noqa: E731
                cin, hp['ae_decoding_n_channels'][i], (k, k), stride=(s, s), padding=pad,
                output_padding=opad)
            is_last = i == n - 1 and not last_ff
            if hp.get('fit_sess_io_layers', False) and is_last:
                name = 'convtranspose%i_sess_io_layers' % g
                self.decoder.add_module(name, nn.ModuleList(
                    [mk() for _ in range(hp['n_datasets'])]))
            else:
                name = 'convtranspose%i' % g
                self.decoder.add_module(name, mk())
            bn = None
            if is_last:
                self.decoder.add_module('sigmoid%i' % g, nn.Sigmoid())
            else:
                if hp['ae_batch_norm']:
                    bn = 'batchnorm%i' % g
                    self.decoder.add_module(bn, nn.BatchNorm2d(
                        hp['ae_decoding_n_channels'][i],
                        momentum=hp.get('ae_batch_norm_momentum', 0.1),
                        track_running_stats=hp.get('track_running_stats', True)))
                self.decoder.add_module('relu%i' % g, nn.LeakyReLU(SLOPE))
            self.layers.append((name, crop, bn, is_last, unpool))
            g += 1
        if last_ff:
            # "have last layer be feedforward if this is 1" (ref aes.py:345-359)
            self.decoder.add_module('last_ff%i' % g, nn.Linear(
                hp['ae_decoding_x_dim'][-1] * hp['ae_decoding_y_dim'][-1] *
                hp['ae_decoding_n_channels'][-1],
                hp['ae_input_dim'][0] * hp['ae_input_dim'][1] * hp['ae_input_dim'][2]))
            self.decoder.add_module('sigmoid%i' % g, nn.Sigmoid())
            self.last_ff = 'last_ff%i' % g
        else:
            self.last_ff = None

    def forward(self, z, pool_idx=None, target_output_size=None, dataset=None, taps=None):
        start = self.hp['ae_decoding_starting_dim']
        x = F.linear(z, self.FF.weight, self.FF.bias).view(-1, start[0], start[1], start[2])
        pool_idx = list(pool_idx) if pool_idx is not None else []
        sizes = list(target_output_size) if target_output_size is not None else []
        for li, (name, crop, bn, is_last, unpool) in enumerate(self.layers):
            if unpool is not None:                                  # ref aes.py:460-464
                x = getattr(self.decoder, unpool)(x, pool_idx.pop(-1), sizes.pop(-1))
            ct = getattr(self.decoder, name)
            if isinstance(ct, nn.ModuleList):
                ct = ct[dataset]
            x = F.conv_transpose2d(x, ct.weight, ct.bias, stride=ct.stride, padding=ct.padding,
                                   output_padding=ct.output_padding)
            if crop is not None:
                x = F.pad(x, [-c for c in crop])
            if is_last:
                x = torch.sigmoid(x)
            else:
                if bn is not None:
                    x = getattr(self.decoder, bn)(x)
                x = _lrelu(x, 'decoding', li)
            if taps is not None:
                taps.append(x)
        if self.last_ff is not None:                     # ref aes.py:478-486
            ff = getattr(self.decoder, self.last_ff)
            x = torch.sigmoid(F.linear(x.reshape(x.shape[0], -1), ff.weight, ff.bias))
            x = x.view(-1, self.hp['ae_input_dim'][0], self.hp['ae_input_dim'][1],
                       self.hp['ae_input_dim'][2])
        return x


def _chunks(batch_size, chunk_size):
    for beg in range(0, batch_size, chunk_size):
        yield beg, min(beg + chunk_size, batch_size)


class LinearEncoder(nn.Module):
    """Dense encoder (ref aes.py:491-544)."""

    def __init__(self, n_latents, input_size):
        super().__init__()
        self.encoder = nn.Linear(int(np.prod(input_size)), n_latents, bias=True)

    def forward(self, x, dataset=None, taps=None):
        return self.encoder(x.view(x.size(0), -1)), None, None


class LinearDecoder(nn.Module):
    """Dense decoder on the transposed encoder weights plus its own bias (ref aes.py:547-613)."""

    def __init__(self, n_latents, output_size, encoder):
        super().__init__()
        self.output_size = tuple(output_size)
        self.encoder = encoder
        self.bias = nn.Parameter(torch.zeros(int(np.prod(output_size))), requires_grad=True)

    def forward(self, x, dataset=None):
        x = F.linear(x, self.encoder.encoder.weight.t()) + self.bias
        return x.view(x.size(0), *self.output_size)


class AE(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        self.model_type = hparams['model_type']
        if self.model_type == 'linear' and type(self) is not AE:
            raise NotImplementedError('oracle covers model_type="linear" for the plain AE only')
        self.build_model()

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        if self.model_type == 'linear':
            size = (self.hparams['n_input_channels'], self.hparams['y_pixels'],
                    self.hparams['x_pixels'])
            self.encoding = LinearEncoder(self.hparams['n_ae_latents'], size)
            self.decoding = LinearDecoder(self.hparams['n_ae_latents'], size, self.encoding)
            return
        self.encoding = ConvEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)

    def get_parameters(self):
        return filter(lambda p: p.requires_grad, self.parameters())

    def forward(self, x, dataset=None, **kwargs):
        z, pi, os_ = self.encoding(x, dataset=dataset)
        if self.model_type == 'linear':
            return self.decoding(z), z
        return self.decoding(z, pi, os_, dataset=dataset), z

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x = data['images'][0]
        m = data['masks'][0] if 'masks' in data else None
        B = x.shape[0]
        total = 0
        for beg, end in _chunks(B, chunk_size):
            x_in = x[beg:end]
            m_in = m[beg:end] if m is not None else None
            x_hat, _ = self.forward(x_in, dataset=dataset)
            loss = mse(x_in, x_hat, m_in)
            if accumulate_grad:
                loss.backward()
            total += loss.item() * (end - beg)
        return {'loss': total / B}


class ConditionalAE(AE):
    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents'] + self.hparams['n_labels']
        self.encoding = ConvEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)

    def forward(self, x, dataset=None, labels=None, labels_2d=None, **kwargs):
        if self.hparams['conditional_encoder']:
            x = torch.cat((x, labels_2d), dim=1)
        z, pi, os_ = self.encoding(x, dataset=dataset)
        return self.decoding(torch.cat((z, labels), dim=1), pi, os_, dataset=dataset), z

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x, y = data['images'][0], data['labels'][0]
        m = data['masks'][0] if 'masks' in data else None
        y2 = data['labels_sc'][0] if self.hparams['conditional_encoder'] else None
        B = x.shape[0]
        total = 0
        for beg, end in _chunks(B, chunk_size):
            x_in = x[beg:end]
            x_hat, _ = self.forward(x_in, dataset=dataset, labels=y[beg:end],
                                    labels_2d=y2[beg:end] if y2 is not None else None)
            loss = mse(x_in, x_hat, m[beg:end] if m is not None else None)
            if accumulate_grad:
                loss.backward()
            total += loss.item() * (end - beg)
        return {'loss': total / B}


def _tables(beta, anneal, max_n_epochs, tail_is_beta):
    tail = (beta if tail_is_beta else 1.0) * np.ones(max_n_epochs + 1)
    if anneal > 0:
        return (np.append(np.linspace(0, beta, anneal), tail),
                np.append(np.linspace(0, 1, anneal), np.ones(max_n_epochs + 1)))
    return beta * np.ones(max_n_epochs + 1), np.ones(max_n_epochs + 1)


class VAE(AE):
    def __init__(self, hparams):
        hparams['variational'] = True
        super().__init__(hparams)
        self.curr_epoch = 0
        anneal = hparams.get('vae.beta_anneal_epochs', 0)
        self.beta_vals, _ = _tables(hparams['vae.beta'], anneal, hparams['max_n_epochs'],
                                    tail_is_beta=anneal <= 0)
        self.eps_fn = None      # tests inject eps to share it with the device run

    def _sample(self, mu, logvar, use_mean):
        if use_mean:
            return mu
        return reparameterize(mu, logvar, self.eps_fn(logvar) if self.eps_fn else None)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        mu, logvar, pi, os_ = self.encoding(x, dataset=dataset)
        z = self._sample(mu, logvar, use_mean)
        return self.decoding(z, pi, os_, dataset=dataset), z, mu, logvar

    def _fwd_kwargs(self, data, beg, end):
        return {}

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x = data['images'][0]
        m = data['masks'][0] if 'masks' in data else None
        beta = self.beta_vals[self.curr_epoch]
        B = x.shape[0]
        acc = {'loss': 0, 'loss_ll': 0, 'loss_kl': 0, 'loss_mse': 0}
        for beg, end in _chunks(B, chunk_size):
            x_in = x[beg:end]
            x_hat, _, mu, logvar = self.forward(
                x_in, dataset=dataset, use_mean=False, **self._fwd_kwargs(data, beg, end))
            ll = gaussian_ll(x_in, x_hat, m[beg:end] if m is not None else None)
            kl = kl_div_to_std_normal(mu, logvar)
            loss = -ll + beta * kl
            if accumulate_grad:
                loss.backward()
            bs = end - beg
            acc['loss'] += loss.item() * bs
            acc['loss_ll'] += ll.item() * bs
            acc['loss_kl'] += kl.item() * bs
            acc['loss_mse'] += gaussian_ll_to_mse(ll.item(), np.prod(x.shape[1:])) * bs
        out = {k: float(v / B) for k, v in acc.items()}
        out['beta'] = beta
        return out


class ConditionalVAE(VAE):
    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents'] + self.hparams['n_labels']
        self.encoding = ConvEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)

    def forward(self, x, dataset=None, labels=None, labels_2d=None, use_mean=False, **kwargs):
        if self.hparams['conditional_encoder']:
            x = torch.cat((x, labels_2d), dim=1)
        mu, logvar, pi, os_ = self.encoding(x, dataset=dataset)
        z = self._sample(mu, logvar, use_mean)
        x_hat = self.decoding(torch.cat((z, labels), dim=1), pi, os_, dataset=dataset)
        return x_hat, z, mu, logvar

    def _fwd_kwargs(self, data, beg, end):
        y2 = data['labels_sc'][0] if self.hparams['conditional_encoder'] else None
        return {'labels': data['labels'][0][beg:end],
                'labels_2d': y2[beg:end] if y2 is not None else None}


class BetaTCVAE(VAE):
    def __init__(self, hparams):
        super().__init__(hparams)
        self.beta_vals, self.kl_anneal_vals = _tables(
            hparams['beta_tcvae.beta'], hparams.get('beta_tcvae.beta_anneal_epochs', 0),
            hparams['max_n_epochs'], True)

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x = data['images'][0]
        m = data['masks'][0] if 'masks' in data else None
        beta, kl = self.beta_vals[self.curr_epoch], self.kl_anneal_vals[self.curr_epoch]
        B = x.shape[0]
        keys = ['loss', 'loss_ll', 'loss_mi', 'loss_tc', 'loss_dwkl']
        acc = {k: 0 for k in keys}
        acc['loss_mse'] = 0
        for beg, end in _chunks(B, chunk_size):
            x_in = x[beg:end]
            x_hat, sample, mu, logvar = self.forward(x_in, dataset=dataset, use_mean=False)
            t = {'loss_ll': gaussian_ll(x_in, x_hat, m[beg:end] if m is not None else None)}
            t['loss_mi'], t['loss_tc'], t['loss_dwkl'] = decomposed_kl(sample, mu, logvar)
            t['loss'] = -t['loss_ll'] + kl * t['loss_mi'] + beta * t['loss_tc'] \
                + kl * t['loss_dwkl']
            if accumulate_grad:
                t['loss'].backward()
            bs = end - beg
            for k in keys:
                acc[k] += t[k].item() * bs
            acc['loss_mse'] += gaussian_ll_to_mse(acc['loss_ll'] / bs, np.prod(x.shape[1:])) * bs
        out = {k: float(v / B) for k, v in acc.items()}
        out['beta'] = beta
        return out


class PSVAE(AE):
    def __init__(self, hparams):
        if hparams['n_ae_latents'] < hparams['n_labels']:
            raise ValueError('PS-VAE model must contain at least as many latents as labels')
        hparams['variational'] = True
        super().__init__(hparams)
        self.curr_epoch = 0
        self.beta_vals, self.kl_anneal_vals = _tables(
            hparams['ps_vae.beta'], hparams.get('ps_vae.anneal_epochs', 0),
            hparams['max_n_epochs'], True)
        self.eps_fn = None

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        self.encoding = ConvPSEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        y, w, logvar, pi, os_ = self.encoding(x, dataset=dataset)
        mu = torch.cat([y, w], dim=1)
        if use_mean:
            z = mu
        else:
            z = reparameterize(mu, logvar, self.eps_fn(logvar) if self.eps_fn else None)
        return self.decoding(z, pi, os_, dataset=dataset), z, mu, logvar, self.encoding.D(y)

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        from sklearn.metrics import r2_score
        x, y = data['images'][0], data['labels'][0]
        m = data['masks'][0] if 'masks' in data else None
        n = data['labels_masks'][0] if 'labels_masks' in data else None
        B = x.shape[0]
        L = self.hparams['n_labels']
        alpha = self.hparams['ps_vae.alpha']
        beta, kl = self.beta_vals[self.curr_epoch], self.kl_anneal_vals[self.curr_epoch]
        keys = ['loss', 'loss_data_ll', 'loss_label_ll', 'loss_zs_kl', 'loss_zu_mi',
                'loss_zu_tc', 'loss_zu_dwkl']
        acc = {k: 0 for k in keys}
        acc['loss_data_mse'] = 0
        y_hats = []
        for beg, end in _chunks(B, chunk_size):
            x_in, y_in = x[beg:end], y[beg:end]
            x_hat, sample, mu, logvar, y_hat = self.forward(x_in, dataset=dataset)
            t = {}
            t['loss_data_ll'] = gaussian_ll(x_in, x_hat, m[beg:end] if m is not None else None)
            t['loss_label_ll'] = gaussian_ll(y_in, y_hat, n[beg:end] if n is not None else None)
            t['loss_zs_kl'] = kl_div_to_std_normal(mu[:, :L], logvar[:, :L])
            t['loss_zu_mi'], t['loss_zu_tc'], t['loss_zu_dwkl'] = decomposed_kl(
                sample[:, L:], mu[:, L:], logvar[:, L:])
            t['loss'] = -t['loss_data_ll'] - alpha * t['loss_label_ll'] + t['loss_zs_kl'] \
                + kl * t['loss_zu_mi'] + beta * t['loss_zu_tc'] + kl * t['loss_zu_dwkl']
            if accumulate_grad:
                t['loss'].backward()
            bs = end - beg
            for k in keys:
                acc[k] += t[k].item() * bs
            acc['loss_data_mse'] += gaussian_ll_to_mse(
                acc['loss_data_ll'] / bs, np.prod(x.shape[1:])) * bs
            y_hats.append(y_hat.detach().numpy())
        y_hat_all = np.concatenate(y_hats, axis=0)
        y_all = y.detach().numpy()
        if n is not None:
            n_np = n.detach().numpy()
            r2 = r2_score(y_all[n_np == 1], y_hat_all[n_np == 1],
                          multioutput='variance_weighted')
        else:
            r2 = r2_score(y_all, y_hat_all, multioutput='variance_weighted')
        out = {k: float(v / B) for k, v in acc.items()}
        out['alpha'], out['beta'], out['label_r2'] = alpha, beta, r2
        return out


def triplet_loss(triplet_loss_obj, z, datasets):
    """Session-separation loss on the background latents (ref losses.py:402-511).

    Per session: its shuffled sample indices are dealt into 3*(n-1) interleaved chunks of equal
    length; chunks (2j, 2j+1) are anchor/positive for the j-th OTHER session, whose next unused
    chunk from index 2*(n-1) on is the negative; plus the mean anchor-positive distance of every
    pair.  Divided by n*(n-1) terms, except the reference's "legacy" 3 for two sessions.
    """
    ids = np.unique(datasets)
    n = len(ids)
    if n not in (2, 3, 4):
        raise NotImplementedError
    n_chunks = 3 * (n - 1)
    perms = [np.random.permutation(np.where(datasets == i)[0]) for i in ids]
    m = np.min([len(p) // n_chunks for p in perms])
    idxs = [[p[i::n_chunks][:m] for i in range(n_chunks)] for p in perms]
    next_neg = [2 * (n - 1)] * n
    loss = 0
    pairs = []
    for a in range(n):
        j = 0
        for b in range(n):
            if b == a:
                continue
            loss = loss + triplet_loss_obj(z[idxs[a][2 * j]], z[idxs[a][2 * j + 1]],
                                           z[idxs[b][next_neg[b]]])
            next_neg[b] += 1
            pairs.append((a, j))
            j += 1
    for a, j in pairs:
        loss = loss + F.pairwise_distance(z[idxs[a][2 * j]], z[idxs[a][2 * j + 1]]).mean()
    return loss / (3 if n == 2 else n * (n - 1))


class ConvMSPSEncoder(ConvEncoder):
    """PS encoder with a background head C (ref vaes.py:1366-1470)."""

    def __init__(self, hp):
        super().__init__(hp)
        n_lat, n_lab, n_bg = hp['n_ae_latents'], hp['n_labels'], hp['n_background']
        self.A = nn.Linear(n_lat, n_lab, bias=False)
        self.B = nn.Linear(n_lat, n_lat - n_lab - n_bg, bias=False)
        self.C = nn.Linear(n_lat, n_bg, bias=True)
        self.D = DiagLinear(n_lab, bias=True)
        from scipy.stats import ortho_group
        m = ortho_group.rvs(dim=n_lat).astype('float32')
        with torch.no_grad():
            self.A.weight = nn.Parameter(torch.from_numpy(m[:n_lab, :]), requires_grad=False)
            self.B.weight = nn.Parameter(torch.from_numpy(m[n_lab + n_bg:, :]),
                                         requires_grad=False)
            self.C.weight = nn.Parameter(torch.from_numpy(m[n_lab:n_lab + n_bg, :]),
                                         requires_grad=False)

    def forward(self, x, dataset=None, taps=None):
        x1 = self.features(x, dataset, taps)
        h = F.linear(x1, self.FF.weight, self.FF.bias)
        return F.linear(h, self.A.weight), F.linear(h, self.C.weight, self.C.bias), \
            F.linear(h, self.B.weight), F.linear(x1, self.logvar.weight, self.logvar.bias), \
            self.pool_idx, self.pool_sizes


class MSPSVAE(PSVAE):
    """Multi-session PS-VAE (ref vaes.py:849-1098): one pass over the concatenated sessions."""

    def __init__(self, hparams):
        if hparams['n_sessions_per_batch'] == 1:
            raise ValueError('must choose "n_sessions_per_batch" > 1 in hparams')
        hparams['n_background'] = hparams.get('n_background', 4)
        super().__init__(hparams)
        self.TripletLoss = nn.TripletMarginLoss(margin=1.0, p=2)

    def build_model(self):
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        self.encoding = ConvMSPSEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)

    def forward(self, x, dataset=None, use_mean=False, **kwargs):
        z_s, z_b, z_u, logvar, pi, os_ = self.encoding(x, dataset=dataset)
        mu = torch.cat([z_s, z_b, z_u], dim=1)
        if use_mean:
            z = mu
        else:
            z = reparameterize(mu, logvar, self.eps_fn(logvar) if self.eps_fn else None)
        return self.decoding(z, pi, os_, dataset=dataset), z, mu, logvar, self.encoding.D(z_s)

    def loss(self, datas, dataset=None, accumulate_grad=True, chunk_size=None):
        from sklearn.metrics import r2_score
        multi = isinstance(datas, list)
        if multi:
            x = torch.cat([d['images'][0] for d in datas], dim=0)
            y = torch.cat([d['labels'][0] for d in datas], dim=0)
            m = torch.cat([d['masks'][0] for d in datas], dim=0) if 'masks' in datas[0] else None
            n = torch.cat([d['labels_masks'][0] for d in datas], dim=0) \
                if 'labels_masks' in datas[0] else None
            sess = np.concatenate([d * np.ones(datas[i]['images'].shape[1])
                                   for i, d in enumerate(dataset)])
        else:
            x, y = datas['images'][0], datas['labels'][0]
            m = datas['masks'][0] if 'masks' in datas else None
            n = datas['labels_masks'][0] if 'labels_masks' in datas else None
        L, G = self.hparams['n_labels'], self.hparams['n_background']
        alpha, delta = self.hparams['ps_vae.alpha'], self.hparams['ps_vae.delta']
        beta, kl = self.beta_vals[self.curr_epoch], self.kl_anneal_vals[self.curr_epoch]
        x_hat, sample, mu, logvar, y_hat = self.forward(x, dataset=None, use_mean=False)
        t = {}
        t['loss_data_ll'] = gaussian_ll(x, x_hat, m)
        t['loss_label_ll'] = gaussian_ll(y, y_hat, n)
        t['loss_zs_kl'] = kl_div_to_std_normal(mu[:, :L], logvar[:, :L])
        t['loss_zu_mi'], t['loss_zu_tc'], t['loss_zu_dwkl'] = decomposed_kl(
            sample[:, L + G:], mu[:, L + G:], logvar[:, L + G:])
        t['loss'] = -t['loss_data_ll'] - alpha * t['loss_label_ll'] + t['loss_zs_kl'] \
            + kl * t['loss_zu_mi'] + beta * t['loss_zu_tc'] + kl * t['loss_zu_dwkl']
        if multi:
            t['loss_triplet'] = triplet_loss(self.TripletLoss, mu[:, L:L + G], sess)
            t['loss'] = t['loss'] + delta * t['loss_triplet']
        if accumulate_grad:
            t['loss'].backward()
        out = {k: v.item() for k, v in t.items()}
        # the key stays in the reference's value dict (as 0) when the term is not computed
        out.setdefault('loss_triplet', 0)
        out['loss_data_mse'] = gaussian_ll_to_mse(out['loss_data_ll'], np.prod(x.shape[1:]))
        y_hat_np, y_np = y_hat.detach().numpy(), y.detach().numpy()
        if n is not None:
            n_np = n.detach().numpy()
            r2 = r2_score(y_np[n_np == 1], y_hat_np[n_np == 1], multioutput='variance_weighted')
        else:
            r2 = r2_score(y_np, y_hat_np, multioutput='variance_weighted')
        out.update({'alpha': alpha, 'beta': beta, 'delta': delta, 'label_r2': r2})
        return out


class AEMSP(AE):
    """Matrix subspace projection AE (ref aes.py:901-1060)."""

    def build_model(self):
        self.n_latents = self.hparams['n_ae_latents']
        self.n_labels = self.hparams['n_labels']
        self.hparams['hidden_layer_size'] = self.hparams['n_ae_latents']
        self.encoding = ConvEncoder(self.hparams)
        self.decoding = ConvDecoder(self.hparams)
        self.projection = nn.Linear(self.n_latents, self.n_labels, bias=False)
        with torch.no_grad():
            self.U = nn.Linear(self.n_latents, self.n_latents, bias=False)

    def forward(self, x, dataset=None, **kwargs):
        z, pi, os_ = self.encoding(x, dataset=dataset)
        y = self.projection(z)
        return self.decoding(z, pi, os_, dataset=dataset), z, y

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        from sklearn.metrics import r2_score
        x, y = data['images'][0], data['labels'][0]
        m = data['masks'][0] if 'masks' in data else None
        B = x.shape[0]
        tot = np.zeros(3)
        y_hat_all = []
        for beg, end in _chunks(B, chunk_size):
            x_in, y_in = x[beg:end], y[beg:end]
            m_in = m[beg:end] if m is not None else None
            x_hat, z, y_hat = self.forward(x_in, dataset=dataset)
            loss_mse = mse(x_in, x_hat, m_in)
            loss_msp = mse(y_in, y_hat) + mse(z, torch.matmul(y_hat, self.projection.weight))
            loss = loss_mse + self.hparams['msp.alpha'] * loss_msp
            if accumulate_grad:
                loss.backward()
            tot += np.array([loss.item(), loss_mse.item(), loss_msp.item()]) * (end - beg)
            y_hat_all.append(y_hat.detach().numpy())
        tot /= B
        r2 = r2_score(y.detach().numpy(), np.concatenate(y_hat_all, axis=0),
                      multioutput='variance_weighted')
        return {'loss': tot[0], 'loss_mse': tot[1], 'loss_msp': tot[2], 'labels_r2': r2}


class ConvDecoderModel(nn.Module):
    """Images from labels with the AE's decoder stack (ref decoders.py:355-496)."""

    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        if hparams['model_type'] != 'conv':
            raise NotImplementedError('oracle covers model_type="conv"')
        self.hparams['hidden_layer_size'] = self.hparams['n_labels']     # ref :404
        self.decoding = ConvDecoder(self.hparams)

    def get_parameters(self):
        return filter(lambda p: p.requires_grad, self.parameters())

    def forward(self, x, dataset=None, **kwargs):
        return self.decoding(x, None, None, dataset=dataset)

    def loss(self, data, dataset=0, accumulate_grad=True, chunk_size=200):
        x, y = data['images'][0], data['labels'][0]
        m = data['masks'][0] if 'masks' in data else None
        B = x.shape[0]
        total = 0
        for beg, end in _chunks(B, chunk_size):
            m_in = m[beg:end] if m is not None else None
            loss = mse(x[beg:end], self.forward(y[beg:end], dataset=dataset), m_in)
            if accumulate_grad:
                loss.backward()
            total += loss.item() * (end - beg)
        return {'loss': total / B}


MODEL_CLASSES = {'conv-decoder': ConvDecoderModel, 'ae': AE, 'cond-ae': ConditionalAE, 'vae': VAE, 'cond-vae': ConditionalVAE,
                 'beta-tcvae': BetaTCVAE, 'ps-vae': PSVAE, 'msps-vae': MSPSVAE,
                 'cond-ae-msp': AEMSP}


def build_model(hparams):
    return MODEL_CLASSES[hparams['model_class']](hparams)


def make_optimizer(model, hparams):
    """The reference's optimizer (training.py:284-286)."""
    return torch.optim.Adam(
        model.get_parameters(), lr=hparams['learning_rate'],
        weight_decay=hparams.get('l2_reg', 0), amsgrad=True)


def train_step(model, optimizer, data, dataset=0, do_step=True):
    """One iteration of the reference hot loop (training.py:336-352)."""
    model.train()
    optimizer.zero_grad()
    loss_dict = model.loss(data, dataset=dataset, accumulate_grad=True)
    if do_step:
        optimizer.step()
    return loss_dict
