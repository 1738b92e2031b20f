"""Records what the REFERENCE's random architecture search draws (run in the build container, where /root/reference is
importable): tests/golden/drawn_archs.json = {"<C>x<H>x<W>/<latents>/<seed>": architecture dict} for seeds 0..29 at four
input sizes, plus get_encoding_conv_block under max pooling for seeds 0..11.  Data only -- inputs and expected outputs.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_drawn_archs.py
"""
import json
import os
import sys
import types

import numpy as np


def _jsonable(v):
    if isinstance(v, dict):
        return {k: _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def main():
    # (the reference imports commentjson for its json configs only; absent here)
    shim = types.ModuleType('commentjson')
    shim.load = json.load
    shim.loads = json.loads
    sys.modules.setdefault('commentjson', shim)
    import behavenet.models.ae_model_architecture_generator as ref
    out = {}
    for dims, n_lat in (([1, 128, 128], 12), ([2, 32, 32], 6), ([1, 64, 48], 8), ([3, 200, 160], 16)):
        for seed in range(30):
            arch = ref.get_possible_arch(list(dims), n_lat, arch_seed=seed)
            out['%dx%dx%d/%d/%d' % (dims[0], dims[1], dims[2], n_lat, seed)] = _jsonable(arch)
    opts = {'possible_kernel_sizes': np.asarray([3, 5]), 'possible_strides': np.asarray([1, 2]),
            'possible_strides_probs': np.asarray([0.1, 0.9]), 'possible_max_pool_sizes': np.asarray([2]),
            'possible_n_channels': np.asarray([16, 32, 64, 128]), 'prob_stopping': np.arange(0, 1, .05),
            'max_latents': 64}
    for pad in ('valid', 'same'):
        for seed in range(12):
            arch = {'ae_input_dim': [2, 32, 32], 'model_type': 'conv', 'n_ae_latents': 6,
                    'ae_decoding_last_FF_layer': 0, 'ae_network_type': 'max_pooling', 'ae_padding_type': pad}
            np.random.seed(seed)
            out['maxpool/%s/%d' % (pad, seed)] = _jsonable(ref.get_encoding_conv_block(arch, opts))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'drawn_archs.json')
    with open(path, 'w') as f:
        json.dump(out, f, sort_keys=True)
    print('%d architectures -> %s' % (len(out), path))


if __name__ == '__main__':
    main()
