"""Layer planner vs the reference: golden outputs (tests/golden/planner.json, captured from the
imported reference) and the closed-form answers the reference's own test file asserts
(reference tests/test_models/test_ae_model_architecture_generator.py:156-338,374-468)."""

import json
import os

import pytest

from behavenet_amd.models import ae_model_architecture_generator as gen

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONFIGS = os.path.join(os.path.dirname(GOLDEN), '..', 'behavenet_amd', 'configs', 'ae_jsons')


def _norm(o):
    if isinstance(o, dict):
        return {k: _norm(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_norm(v) for v in o]
    return o


def test_calculate_output_dim_grid():
    with open(os.path.join(GOLDEN, 'planner.json')) as f:
        grid = json.load(f)['calculate_output_dim']
    assert len(grid) == 280
    for inp, k, s, pad, kind, out, before, after in grid:
        assert gen.calculate_output_dim(inp, k, s, pad, kind) == (out, before, after), \
            (inp, k, s, pad, kind)


def test_calculate_output_dim_known_answers():
    # reference test file :163-217 (conv/same, even/odd combos)
    assert gen.calculate_output_dim(16, 4, 3, 'same', 'conv') == (6, 1, 2)
    assert gen.calculate_output_dim(17, 4, 3, 'same', 'conv') == (6, 1, 1)
    assert gen.calculate_output_dim(16, 3, 3, 'same', 'conv') == (6, 1, 1)
    assert gen.calculate_output_dim(17, 3, 3, 'same', 'conv') == (6, 0, 1)
    assert gen.calculate_output_dim(16, 4, 2, 'same', 'conv') == (8, 1, 1)
    assert gen.calculate_output_dim(17, 4, 2, 'same', 'conv') == (9, 1, 2)
    with pytest.raises(NotImplementedError):
        gen.calculate_output_dim(16, 3, 2, 'same', 'maxpool')
    with pytest.raises(NotImplementedError):
        gen.calculate_output_dim(16, 3, 2, 'test', 'conv')
    with pytest.raises(NotImplementedError):
        gen.calculate_output_dim(16, 2, 2, 'same', 'test')


def test_default_arch_dims():
    # reference test file :374-387
    a = gen.load_default_arch()
    a['ae_input_dim'] = [2, 128, 128]
    a = gen.get_handcrafted_dims(a, symmetric=True)
    assert a['ae_encoding_x_dim'] == [64, 32, 16, 8, 2]
    assert a['ae_encoding_y_dim'] == [64, 32, 16, 8, 2]
    assert a['ae_encoding_x_padding'] == [(1, 2), (1, 2), (1, 2), (1, 2), (1, 1)]
    assert a['ae_encoding_y_padding'] == [(1, 2), (1, 2), (1, 2), (1, 2), (1, 1)]
    assert a['ae_decoding_x_dim'] == [8, 16, 32, 64, 128]
    assert a['ae_decoding_y_dim'] == [8, 16, 32, 64, 128]
    assert a['ae_decoding_x_padding'] == [(1, 1), (1, 2), (1, 2), (1, 2), (1, 2)]
    assert a['ae_decoding_y_padding'] == [(1, 1), (1, 2), (1, 2), (1, 2), (1, 2)]


def test_asymmetric_decoder_dims():
    # reference test file :389-405
    a = gen.load_default_arch()
    a['ae_input_dim'] = [2, 128, 128]
    a['ae_decoding_n_channels'] = [64, 32, 32]
    a['ae_decoding_kernel_size'] = [5, 5, 5]
    a['ae_decoding_stride_size'] = [2, 2, 2]
    a['ae_decoding_layer_type'] = ['conv', 'conv', 'conv']
    a['ae_decoding_starting_dim'] = [1, 8, 8]
    a = gen.get_handcrafted_dims(a, symmetric=False)
    assert a['ae_decoding_x_dim'] == [15, 29, 57]
    assert a['ae_decoding_y_dim'] == [15, 29, 57]
    assert a['ae_decoding_x_padding'] == [(2, 2), (2, 2), (2, 2)]
    assert a['ae_decoding_y_padding'] == [(2, 2), (2, 2), (2, 2)]
    b = gen.load_default_arch()
    b['ae_input_dim'] = [2, 128, 128]
    b['ae_network_type'] = 'max_pooling'
    with pytest.raises(NotImplementedError):
        gen.get_handcrafted_dims(b, symmetric=False)


def test_decoding_block_last_ff():
    a = gen.load_default_arch()
    a['ae_input_dim'] = [2, 128, 128]
    a = gen.get_handcrafted_dims(a)
    a['ae_decoding_last_FF_layer'] = True
    a = gen.get_decoding_conv_block(a)
    assert a['ae_decoding_n_channels'][-1] == 16


@pytest.mark.parametrize('case', [
    'default_1x32x32', 'default_1x128x128', 'default_2x128x128', 'default_1x64x48',
    'arch2_2x128x128', 'archdefault_1x128x128'])
def test_load_handcrafted_arch_matches_reference(case):
    with open(os.path.join(GOLDEN, 'planner.json')) as f:
        ref = json.load(f)[case]
    path = None if ref['arch_json'] is None else os.path.join(CONFIGS, ref['arch_json'])
    arch = gen.load_handcrafted_arch(
        list(ref['input_dim']), ref['n_ae_latents'], path, check_memory=False)
    want = ref['arch']
    got = _norm(arch)
    for key, val in want.items():
        if val is None and key not in got:
            continue
        assert got[key] == val, key
    assert set(got.keys()) == set(want.keys())


def test_missing_json_falls_back_to_default():
    arch = gen.load_handcrafted_arch(
        [2, 128, 128], 12, os.path.join(CONFIGS, 'ae_arch_3.json'), check_memory=False)
    assert arch['ae_encoding_n_channels'] == [32, 64, 128, 256, 512]


def test_load_handcrafted_arches_latent_formats():
    assert len(gen.load_handcrafted_arches([1, 32, 32], 8, None, check_memory=False)) == 1
    archs = gen.load_handcrafted_arches([1, 32, 32], '[4,8]', None, check_memory=False)
    assert [a['n_ae_latents'] for a in archs] == [4, 8]
    archs = gen.load_handcrafted_arches([1, 32, 32], '12', None, check_memory=False)
    assert archs[0]['n_ae_latents'] == 12


# ---------------------------------------------------------------------------------------------------------------
# random architecture search (ref ae_model_architecture_generator.py:7-268): a seed must name the same architecture
# as in the reference -- tests/golden/drawn_archs.json was recorded from the imported reference
# (tests/golden/make_drawn_archs.py)
# ---------------------------------------------------------------------------------------------------------------
def _drawn():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'drawn_archs.json')) as f:
        return json.load(f)


def _plain(v):
    import numpy as np
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def test_get_possible_arch_draws_the_references_architectures():
    from behavenet_amd.models import ae_model_architecture_generator as gen
    drawn = _drawn()
    keys = [k for k in drawn if not k.startswith('maxpool/')]
    assert len(keys) == 120
    kinds = set()
    for key in keys:
        dims, n_lat, seed = key.split('/')
        arch = gen.get_possible_arch([int(v) for v in dims.split('x')], int(n_lat), arch_seed=int(seed))
        assert _plain(arch) == drawn[key], key
        kinds.add((arch['ae_padding_type'], len(arch['ae_encoding_n_channels'])))
    assert len(kinds) >= 8                      # both padding types, depths from one layer to many
    with pytest.raises(ValueError):
        gen.get_possible_arch([2, 32, 32], 65, 0)


def test_get_encoding_conv_block_under_max_pooling_matches_the_reference():
    import numpy as np
    from behavenet_amd.models import ae_model_architecture_generator as gen
    drawn = _drawn()
    opts = {'possible_kernel_sizes': np.asarray([3, 5]), 'possible_strides': np.asarray([1, 2]),
            'possible_strides_probs': np.asarray([0.1, 0.9]), 'possible_max_pool_sizes': np.asarray([2]),
            'possible_n_channels': np.asarray([16, 32, 64, 128]), 'prob_stopping': np.arange(0, 1, .05),
            'max_latents': 64}
    n = 0
    for key in drawn:
        if not key.startswith('maxpool/'):
            continue
        _, pad, seed = key.split('/')
        arch = {'ae_input_dim': [2, 32, 32], 'model_type': 'conv', 'n_ae_latents': 6,
                'ae_decoding_last_FF_layer': 0, 'ae_network_type': 'max_pooling', 'ae_padding_type': pad}
        np.random.seed(int(seed))
        got = gen.get_encoding_conv_block(arch, opts)
        assert _plain(got) == drawn[key], key
        kinds = got['ae_encoding_layer_type']
        assert all(a == 'conv' and b == 'maxpool' for a, b in zip(kinds[0::2], kinds[1::2])) and len(kinds) % 2 == 0
        n += 1
    assert n == 24


def test_draw_archs_distinct_and_memory_checked():
    """The reference's own test (tests/test_models/test_ae_model_architecture_generator.py:7-38)."""
    from behavenet_amd.models import ae_model_architecture_generator as gen
    archs = gen.draw_archs(batch_size=100, input_dim=[2, 32, 32], n_ae_latents=6, n_archs=3, check_memory=False,
                           mem_limit_gb=None)
    assert len(archs) == 3 and all(a['n_ae_latents'] == 6 for a in archs)
    assert all(sum(a == b for b in archs) == 1 for a in archs)
    # seeds in order: the first three distinct draws
    assert archs[0] == gen.get_possible_arch([2, 32, 32], 6, arch_seed=0)
    archs = gen.draw_archs(batch_size=100, input_dim=[2, 32, 32], n_ae_latents=6, n_archs=3, check_memory=True,
                           mem_limit_gb=1)
    assert len(archs) == 3
    assert all(a['mem_size_gb'] < 1 and a['n_ae_latents'] == 6 for a in archs)
    assert all(sum(a == b for b in archs) == 1 for a in archs)
