"""Layer planner: architecture dict/json -> fully specified conv-AE geometry.

Host-side mirror of the reference module of the same name
(``behavenet/models/ae_model_architecture_generator.py``); function names, argument meaning,
returned keys and error behaviour follow the reference so the parity tests read like the
reference's own ``tests/test_models/test_ae_model_architecture_generator.py``.

The rules are re-stated from SURVEY.md Appendix C:

* conv/same:  ``out = ceil(in / stride)``, ``total = max(0, (out-1)*stride + k - in)``,
  ``before = total // 2``, ``after = total - before``          (ref :379-383)
* conv/valid: ``out = floor((in - k) / stride + 1)``, no padding                 (ref :384-387)
* maxpool (k == 2 only): same -> ceil, valid -> floor, no padding                (ref :391-405)
* the decoder mirrors the encoder, last encoder layer first                      (ref :271-344)
"""

import copy
import json
import math

import numpy as np

__all__ = [
    'calculate_output_dim', 'get_decoding_conv_block', 'get_handcrafted_dims',
    'load_default_arch', 'load_handcrafted_arch', 'load_handcrafted_arches',
    'estimate_model_footprint', 'load_commented_json',
    'default_search_options', 'get_encoding_conv_block', 'get_possible_arch', 'draw_archs']


def load_commented_json(path):
    """Parse a json file that may carry ``# ...`` comments (the reference uses ``commentjson``).

    A ``#`` only starts a comment outside of a string literal.
    """
    lines = []
    with open(path, 'r') as f:
        for raw in f:
            out, in_str, esc = [], False, False
            for ch in raw:
                if in_str:
                    out.append(ch)
                    if esc:
                        esc = False
                    elif ch == '\\':
                        esc = True
                    elif ch == '"':
                        in_str = False
                elif ch == '"':
                    in_str = True
                    out.append(ch)
                elif ch == '#':
                    break
                else:
                    out.append(ch)
            lines.append(''.join(out))
    return json.loads('\n'.join(lines))


def calculate_output_dim(input_dim, kernel, stride, padding_type, layer_type):
    """Output size and (before, after) zero padding of one spatial dim (ref :347-410)."""
    if layer_type == 'conv':
        if padding_type == 'same':
            out = -(-input_dim // stride)
            total = max(0, (out - 1) * stride + kernel - input_dim)
            before = total // 2
            return out, before, total - before
        if padding_type == 'valid':
            return int(math.floor((input_dim - kernel) / stride + 1)), 0, 0
        raise NotImplementedError
    if layer_type == 'maxpool':
        if kernel != 2:
            raise NotImplementedError
        frac = (input_dim - kernel) / stride + 1
        if padding_type == 'same':
            return int(math.ceil(frac)), 0, 0
        if padding_type == 'valid':
            return int(math.floor(frac)), 0, 0
        raise NotImplementedError
    raise NotImplementedError


def get_decoding_conv_block(arch):
    """Fill the ``ae_decoding_*`` keys as the mirror image of the encoder (ref :271-344)."""
    enc_c = arch['ae_encoding_n_channels']
    n_layers = len(enc_c)
    arch['ae_decoding_starting_dim'] = [
        enc_c[-1], arch['ae_encoding_y_dim'][-1], arch['ae_encoding_x_dim'][-1]]
    for key in ('x_dim', 'y_dim', 'x_padding', 'y_padding', 'n_channels', 'kernel_size',
                'stride_size', 'layer_type'):
        arch['ae_decoding_' + key] = []
    for src in range(n_layers - 1, -1, -1):
        if src == 0:
            ch, ydim, xdim = arch['ae_input_dim']
        else:
            ch = enc_c[src - 1]
            ydim = arch['ae_encoding_y_dim'][src - 1]
            xdim = arch['ae_encoding_x_dim'][src - 1]
        arch['ae_decoding_n_channels'].append(ch)
        arch['ae_decoding_y_dim'].append(ydim)
        arch['ae_decoding_x_dim'].append(xdim)
        for key in ('kernel_size', 'stride_size', 'x_padding', 'y_padding'):
            arch['ae_decoding_' + key].append(arch['ae_encoding_' + key][src])
        kind = arch['ae_encoding_layer_type'][src]
        if kind == 'maxpool':
            arch['ae_decoding_layer_type'].append('unpool')
        elif kind == 'conv':
            arch['ae_decoding_layer_type'].append('convtranspose')
    if arch['ae_decoding_last_FF_layer']:
        # a final dense layer follows: keep its fan-in small (ref :339-341)
        arch['ae_decoding_n_channels'][-1] = 16
    return arch


def get_handcrafted_dims(arch, symmetric=True):
    """Per-layer output dims and paddings for a handcrafted architecture (ref :482-592)."""
    arch['model_type'] = 'conv'
    for key in ('x_dim', 'y_dim', 'x_padding', 'y_padding'):
        arch['ae_encoding_' + key] = []
    ydim, xdim = arch['ae_input_dim'][1], arch['ae_input_dim'][2]
    for k, s, kind in zip(arch['ae_encoding_kernel_size'], arch['ae_encoding_stride_size'],
                          arch['ae_encoding_layer_type']):
        xdim, xb, xa = calculate_output_dim(xdim, k, s, arch['ae_padding_type'], kind)
        ydim, yb, ya = calculate_output_dim(ydim, k, s, arch['ae_padding_type'], kind)
        arch['ae_encoding_x_dim'].append(xdim)
        arch['ae_encoding_y_dim'].append(ydim)
        arch['ae_encoding_x_padding'].append((xb, xa))
        arch['ae_encoding_y_padding'].append((yb, ya))

    if symmetric:
        return get_decoding_conv_block(arch)

    # user-specified decoder: un-pooling cannot be matched up with pooling layers (ref :543-546)
    if arch['ae_network_type'] == 'max_pooling' or \
            any(t == 'unpool' for t in arch['ae_decoding_layer_type']):
        raise NotImplementedError
    for key in ('x_dim', 'y_dim', 'x_padding', 'y_padding'):
        arch['ae_decoding_' + key] = []
    ydim, xdim = arch['ae_decoding_starting_dim'][1], arch['ae_decoding_starting_dim'][2]
    for k, s in zip(arch['ae_decoding_kernel_size'], arch['ae_decoding_stride_size']):
        if arch['ae_padding_type'] == 'valid':
            continue  # reference leaves these lists empty (marked "TODO: not correct" there)
        if arch['ae_padding_type'] != 'same':
            raise NotImplementedError
        x_out = xdim * s - s + 1
        x_tot = max(0, (xdim - 1) * s + k - x_out)
        y_out = ydim * s - s + 1
        y_tot = max(0, (ydim - 1) * s + k - y_out)
        xb, yb = x_tot // 2, y_tot // 2
        arch['ae_decoding_x_dim'].append(x_out)
        arch['ae_decoding_y_dim'].append(y_out)
        arch['ae_decoding_x_padding'].append((xb, x_tot - xb))
        # the reference derives the trailing y pad from the x total (ref :578); kept as is
        arch['ae_decoding_y_padding'].append((yb, x_tot - yb))
        xdim, ydim = x_out, y_out
    return arch


def load_default_arch():
    """Default architecture of the BehaveNet paper (ref :707-720)."""
    n = 5
    return {
        'ae_network_type': 'strides_only',
        'ae_padding_type': 'same',
        'ae_batch_norm': 0,
        'ae_batch_norm_momentum': None,
        'symmetric_arch': 1,
        'ae_encoding_n_channels': [32 << i for i in range(n)],
        'ae_encoding_kernel_size': [5] * n,
        'ae_encoding_stride_size': [2] * (n - 1) + [5],
        'ae_encoding_layer_type': ['conv'] * n,
        'ae_decoding_last_FF_layer': 0}


def estimate_model_footprint(model, input_dim, cutoff_size=20):
    """Bytes needed for input + parameters + activations/gradients, x1.2 (ref :413-479).

    Activations are counted analytically from the layer plan (the reference pushes an
    uninitialised tensor through the encoder; only the sizes matter).
    """
    nbytes = 4
    total = float(np.prod(input_dim)) * nbytes
    for mod in model.modules():
        if getattr(mod, '_bn_counts_for_footprint', False):
            for p in mod.parameters(recurse=False):
                total += p.numel() * nbytes
    hp = model.hparams
    n = input_dim[0]
    for i, kind in enumerate(hp['ae_encoding_layer_type']):
        numel = n * hp['ae_encoding_n_channels'][i] * hp['ae_encoding_y_dim'][i] * \
            hp['ae_encoding_x_dim'][i]
        # the reference adds one term per module in the ModuleList whose output has this size
        n_modules = _modules_per_layer(hp, i, kind)
        # x2 symmetric decoder, x2 values + gradients
        for _ in range(n_modules):
            total += numel * nbytes * 2 * 2
            if total / 1e9 > cutoff_size:
                return total * 1.2
    return total * 1.2


def _modules_per_layer(hp, i, kind):
    """Number of encoder modules emitting a tensor of layer i's output size (ref aes.py:55-115).

    zero_pad emits the padded input (counted separately below is not needed: the reference adds
    the padded tensor too, which this mirrors by size).
    """
    if kind == 'maxpool':
        return 0  # accounted with its conv layer
    n = 1  # conv
    if hp['ae_batch_norm']:
        n += 1
    n += 1  # leaky relu
    return n


def load_handcrafted_arch(
        input_dim, n_ae_latents, ae_arch_json, batch_size=None, check_memory=True,
        mem_limit_gb=10):
    """Load one handcrafted architecture (``None`` -> default) and plan it (ref :595-660)."""
    if ae_arch_json is None:
        arch = load_default_arch()
    else:
        try:
            arch = load_commented_json(ae_arch_json)
        except FileNotFoundError:
            print('Warning! could not find ae arch defined in %s; using default architecture' %
                  ae_arch_json)
            arch = load_default_arch()

    arch['ae_batch_norm'] = arch['ae_batch_norm'] == 1
    arch['n_input_channels'], arch['y_pixels'], arch['x_pixels'] = input_dim
    arch['ae_input_dim'] = input_dim
    arch['n_ae_latents'] = n_ae_latents
    arch = get_handcrafted_dims(arch, symmetric=arch['symmetric_arch'] == 1)

    if check_memory:
        from behavenet_amd.models.aes import AE
        probe = copy.deepcopy(arch)
        probe['model_class'] = 'ae'
        probe['n_input_channels'], probe['y_pixels'], probe['x_pixels'] = input_dim
        probe['device'] = 'meta_plan'  # plan only: no device buffers are touched
        model = AE(probe)
        mem_gb = estimate_model_footprint(model, tuple([batch_size] + list(input_dim))) / 1e9
        if mem_gb > mem_limit_gb:
            raise ValueError('Handcrafted architecture from %s too big for memory' % ae_arch_json)
        arch['mem_size_gb'] = mem_gb
    return arch


def load_handcrafted_arches(
        input_dim, n_ae_latents, ae_arch_json, batch_size=None, check_memory=True,
        mem_limit_gb=10):
    """One planned architecture per latent count; accepts int, list or "[a,b]" (ref :663-704)."""
    if isinstance(n_ae_latents, int):
        n_ae_latents = [n_ae_latents]
    elif isinstance(n_ae_latents, str):
        if ',' in n_ae_latents:
            n_ae_latents = [int(v) for v in n_ae_latents[1:-1].split(',')]
        else:
            n_ae_latents = [int(n_ae_latents)]
    return [
        load_handcrafted_arch(input_dim, n, ae_arch_json, batch_size, check_memory, mem_limit_gb)
        for n in n_ae_latents]


# ---------------------------------------------------------------------------------------------
# Random architecture search (ref :7-268).  The SEQUENCE of draws from numpy's global generator is the
# reference's -- seed, padding type, then per layer: kernel size, stride, channel count, [pool size], stop
# flag -- so a seed names the same architecture in both code bases (tests/golden/drawn_archs.json, recorded
# from the imported reference).  The construction differs: layers are collected as records and the
# ``ae_encoding_*`` lists are written once at the end.
# ---------------------------------------------------------------------------------------------
def default_search_options():
    """The search space of ``get_possible_arch`` (ref :90-100): what a layer may be drawn from."""
    return {
        'possible_kernel_sizes': np.asarray([3, 5, 7, 9]),
        'possible_strides': np.asarray([1, 2]),                 # always 1 under max pooling
        'possible_strides_probs': np.asarray([0.1, 0.9]),
        'possible_max_pool_sizes': np.asarray([2]),             # (larger windows: not implemented there either)
        'possible_n_channels': np.asarray([16, 32, 64, 128, 256, 512]),
        'prob_stopping': np.arange(0, 1, .05),                  # chance to stop after the k-th block
        'max_latents': 64,
    }


_ENC_KEYS = ('n_channels', 'kernel_size', 'stride_size', 'x_dim', 'y_dim', 'x_padding', 'y_padding', 'layer_type')


def get_encoding_conv_block(arch, opts):
    """Draw the encoder of ``arch`` (needs ``ae_input_dim``, ``ae_network_type``, ``ae_padding_type``) layer by
    layer from ``opts`` until the feature map would fall below ``opts['max_latents']`` values, a side below one
    pixel, or the stop flag comes up; fills the ``ae_encoding_*`` lists (ref :132-268)."""
    c_in, y_in, x_in = arch['ae_input_dim']
    pooled = arch['ae_network_type'] == 'max_pooling'
    pad = arch['ae_padding_type']
    floor = opts['max_latents']
    menu = opts['possible_n_channels']

    def survives(c, y, x):
        return c * x * y >= floor and min(x, y) >= 1

    layers = []          # (channels, kernel, stride, x_dim, y_dim, x_padding, y_padding, kind)
    cur_c, cur_y, cur_x = c_in, y_in, x_in
    blocks = 0
    while cur_c * cur_y * cur_x >= floor and min(cur_y, cur_x) >= 1:
        kernel = np.random.choice(opts['possible_kernel_sizes'])
        stride = 1 if pooled else np.random.choice(opts['possible_strides'], p=opts['possible_strides_probs'])
        y_out, y_lo, y_hi = calculate_output_dim(cur_y, kernel, stride, padding_type=pad, layer_type='conv')
        x_out, x_lo, x_hi = calculate_output_dim(cur_x, kernel, stride, padding_type=pad, layer_type='conv')
        # channel counts never shrink; the smallest admissible one is favoured 3 : 1 over all the others together
        wider = menu[menu >= cur_c]
        weights = [1.0] if len(wider) == 1 else [.75] + [.25 / (len(wider) - 1)] * (len(wider) - 1)
        n_ch = np.random.choice(wider, p=weights)
        if not survives(n_ch, y_out, x_out):
            break
        layers.append((n_ch, kernel, stride, x_out, y_out, (x_lo, x_hi), (y_lo, y_hi), 'conv'))
        cur_c, cur_y, cur_x = n_ch, y_out, x_out
        if pooled:
            window = np.random.choice(opts['possible_max_pool_sizes'])
            y_out, y_lo, y_hi = calculate_output_dim(cur_y, window, window, padding_type=pad, layer_type='maxpool')
            x_out, x_lo, x_hi = calculate_output_dim(cur_x, window, window, padding_type=pad, layer_type='maxpool')
            if not survives(n_ch, y_out, x_out):
                layers.pop()                      # a convolution is never left without its pooling layer
                break
            layers.append((n_ch, window, window, x_out, y_out, (x_lo, x_hi), (y_lo, y_hi), 'maxpool'))
            cur_y, cur_x = y_out, x_out
        p_stop = opts['prob_stopping'][blocks]
        if np.random.choice([0, 1], p=[1 - p_stop, p_stop]):
            break
        blocks += 1
    for pos, key in enumerate(_ENC_KEYS):
        arch['ae_encoding_' + key] = [rec[pos] for rec in layers]
    return arch


def get_possible_arch(input_dim, n_ae_latents, arch_seed=0):
    """One random strides-only conv architecture, reproducible from ``arch_seed`` (ref :70-129)."""
    np.random.seed(arch_seed)
    opts = default_search_options()
    if n_ae_latents > opts['max_latents']:
        raise ValueError('Number of latents higher than max latents')
    arch = {'ae_input_dim': input_dim, 'model_type': 'conv', 'n_ae_latents': n_ae_latents,
            'ae_decoding_last_FF_layer': 0, 'ae_batch_norm': 0, 'ae_batch_norm_momentum': None,
            'ae_network_type': 'strides_only'}
    arch['ae_padding_type'] = ('valid', 'same')[np.random.randint(2)]
    return get_decoding_conv_block(get_encoding_conv_block(arch, opts))


def draw_archs(batch_size, input_dim, n_ae_latents, n_archs=100, check_memory=True, mem_limit_gb=5.0):
    """``n_archs`` distinct random architectures with ``n_ae_latents`` latents (seeds 0, 1, 2, ... in order);
    with ``check_memory`` those whose estimated footprint at ``batch_size`` exceeds ``mem_limit_gb`` are skipped
    and the others carry ``mem_size_gb`` (ref :7-67)."""
    kept, seed = [], 0
    while len(kept) < n_archs:
        cand = get_possible_arch(input_dim, n_ae_latents, arch_seed=seed)
        seed += 1
        if check_memory:
            from behavenet_amd.models import AE
            probe = copy.deepcopy(cand)
            probe.update(model_class='ae', n_input_channels=input_dim[0], y_pixels=input_dim[1],
                         x_pixels=input_dim[2])
            gb = estimate_model_footprint(AE(probe), tuple([batch_size] + list(input_dim))) / 1e9
            if gb > mem_limit_gb:
                print('Model size of %02.3f GB is larger than limit of %1.3f GB; skipping model' % (gb, mem_limit_gb))
                continue
            cand['mem_size_gb'] = gb
        if not any(cand == other for other in kept):
            kept.append(cand)
    return kept
