"""Host-side model construction (no kernels run): parameter names, initial values and the fused
layer plan must agree with the reference (golden checksums) on the CPU."""

import numpy as np
import pytest
import torch

from behavenet_amd import _hip
from behavenet_amd.models import AE, VAE, ConditionalVAE, BetaTCVAE, PSVAE
from tests.cases import load_case, case_hparams, case_data, seeded_build
from tests.golden_utils import checksum

BUILDERS = {'ae': AE, 'vae': VAE, 'cond-vae': ConditionalVAE, 'beta-tcvae': BetaTCVAE,
            'ps-vae': PSVAE}


@pytest.mark.parametrize('name', ['ae_cfg1', 'ae_cfg2', 'ae_1x64x48', 'vae_cfg1',
                                  'condvae_cfg1', 'psvae_cfg4'])
def test_initial_parameters_bit_exact(name):
    z, meta = load_case(name)
    hp = case_hparams(meta)
    model = seeded_build(BUILDERS[meta['model_class']], hp)
    sd = model.state_dict()
    want = sorted(k[len('param0/'):-len('/checksum')] for k in z.files
                  if k.startswith('param0/') and k.endswith('/checksum'))
    assert sorted(sd.keys()) == want
    for k, v in sd.items():
        np.testing.assert_array_equal(checksum(v.numpy()), z['param0/' + k + '/checksum'], k)


def test_default_plan_geometry():
    z, meta = load_case('ae_cfg2')
    model = seeded_build(AE, case_hparams(meta))
    enc = [(p.kind, p.cin, p.hin, p.cout, p.hout, p.stride, p.off_t, p.off_l)
           for p in model.encoding._plan]
    assert enc == [('conv', 1, 128, 32, 64, 2, 1, 1), ('conv', 32, 64, 64, 32, 2, 1, 1),
                   ('conv', 64, 32, 128, 16, 2, 1, 1), ('conv', 128, 16, 256, 8, 2, 1, 1),
                   ('conv', 256, 8, 512, 2, 5, 1, 1)]
    dec = [(p.kind, p.cin, p.hin, p.cout, p.hout, p.stride, p.off_t, p.act)
           for p in model.decoding._plan]
    assert dec == [('convT', 512, 2, 256, 8, 5, 1, _hip.ACT_LRELU),
                   ('convT', 256, 8, 128, 16, 2, 1, _hip.ACT_LRELU),
                   ('convT', 128, 16, 64, 32, 2, 1, _hip.ACT_LRELU),
                   ('convT', 64, 32, 32, 64, 2, 1, _hip.ACT_LRELU),
                   ('convT', 32, 64, 1, 128, 2, 1, _hip.ACT_SIGMOID)]
    # non-square input: x and y are planned independently
    z, meta = load_case('ae_1x64x48')
    model = seeded_build(AE, case_hparams(meta))
    last = model.encoding._plan[-1]
    assert (last.hin, last.win, last.hout, last.wout, last.off_t, last.off_l) == (4, 3, 1, 1, 0, 1)
    assert 'zero_pad4' in dict(model.encoding.encoder.named_children())


def test_str_lists_reference_modules():
    z, meta = load_case('ae_cfg2')
    s = str(seeded_build(AE, case_hparams(meta)))
    assert '00: ZeroPad2d((1, 2, 1, 2))' in s
    assert '12: Conv2d(256, 512, kernel_size=(5, 5), stride=(5, 5), padding=(1, 1))' in s
    assert '14: Linear(in_features=2048, out_features=12, bias=True)' in s
    assert '09: ConvTranspose2d(32, 1, kernel_size=(5, 5), stride=(2, 2))' in s
    assert '10: Sigmoid()' in s


def test_no_cpu_fallback():
    """The product path must fail loudly without the GPU instead of computing on the CPU."""
    z, meta = load_case('ae_cfg1')
    model = seeded_build(AE, case_hparams(meta))
    data = case_data(meta)
    with pytest.raises(_hip.HipLibraryError):
        model.loss(data, dataset=0, accumulate_grad=False)


def test_bad_hparams_raise_like_reference():
    z, meta = load_case('ae_cfg1')
    hp = case_hparams(meta)
    hp['model_type'] = 'bogus'
    with pytest.raises(ValueError):
        AE(hp)
    hp = case_hparams(meta)
    hp['model_type'] = 'linear'
    hp['max_n_epochs'] = 1
    hp['vae.beta'] = 1
    with pytest.raises(NotImplementedError):
        VAE(hp)
    hp = case_hparams(meta)
    hp.update({'n_labels': 20, 'max_n_epochs': 1, 'ps_vae.beta': 1, 'ps_vae.alpha': 1})
    with pytest.raises(ValueError):
        PSVAE(hp)


def test_deepcopy_and_state_dict_roundtrip(tmp_path):
    import copy
    z, meta = load_case('ae_cfg1')
    hp = case_hparams(meta)
    model = seeded_build(AE, hp)
    model.hparams = None
    clone = copy.deepcopy(model)
    model.hparams = hp
    clone.hparams = hp
    path = str(tmp_path / 'm.pt')
    model.save(path)
    loaded = torch.load(path)
    assert sorted(loaded.keys()) == sorted(clone.state_dict().keys())
    other = AE(case_hparams(meta))
    other.load_state_dict(loaded)
    for k, v in other.state_dict().items():
        assert torch.equal(v, model.state_dict()[k])


def test_conv_decoder_initial_parameters_bit_exact():
    from behavenet_amd.models import ConvDecoder
    z, meta = load_case('convdecoder_cfg1')
    model = seeded_build(ConvDecoder, case_hparams(meta))
    assert model.hparams['hidden_layer_size'] == meta['n_labels']
    sd = model.state_dict()
    want = sorted(k[len('param0/'):-len('/checksum')] for k in z.files
                  if k.startswith('param0/') and k.endswith('/checksum'))
    assert sorted(sd.keys()) == want
    for k, v in sd.items():
        np.testing.assert_array_equal(checksum(v.numpy()), z['param0/' + k + '/checksum'], k)
    assert 'Convolutional decoder architecture' in str(model)


class _LabelGen(object):
    """Minimal generator: what build_model reads of it."""
    n_datasets = 3

    def __init__(self, n_labels):
        self.n_labels, self.calls = n_labels, []

    def next_batch(self, dtype):
        self.calls.append(dtype)
        return {'labels': torch.zeros((1, 5, self.n_labels))}, 0


def test_model_dispatch_like_ae_grid_search(tmp_path):
    """ref ae_grid_search.py:52-95: class by ``model_class``, seeding, n_datasets, n_labels from
    one validation batch, pretrained weights, NotImplementedError for unknown classes."""
    from behavenet_amd.fitting import ae_grid_search as ags
    from behavenet_amd import models
    z, meta = load_case('ae_cfg1')
    extra = {'device': 'cpu', 'max_n_epochs': 2, 'vae.beta': 1, 'vae.beta_anneal_epochs': 0,
             'beta_tcvae.beta': 1, 'beta_tcvae.beta_anneal_epochs': 0, 'ps_vae.alpha': 1,
             'ps_vae.beta': 1, 'ps_vae.anneal_epochs': 0, 'msp.alpha': 0.1,
             'conditional_encoder': False, 'rng_seed_model': 0, 'n_sessions_per_batch': 2,
             'n_background': 2, 'ps_vae.delta': 1}
    want = {'ae': models.AE, 'vae': models.VAE, 'beta-tcvae': models.BetaTCVAE,
            'ps-vae': models.PSVAE, 'msps-vae': models.MSPSVAE,
            'cond-vae': models.ConditionalVAE,
            'cond-ae': models.ConditionalAE, 'cond-ae-msp': models.AEMSP,
            'conv-decoder': models.ConvDecoder}
    assert sorted(want) == sorted(ags.MODEL_CLASSES)
    for mc, cls in want.items():
        hp = case_hparams(meta)
        hp.update(extra)
        hp['model_class'] = mc
        gen = _LabelGen(4)
        model = ags.build_model(hp, gen)
        assert type(model) is cls
        assert hp['n_datasets'] == 3
        assert gen.calls == (['val'] if mc in ags.NEEDS_LABELS else [])
        if mc in ags.NEEDS_LABELS:
            assert hp['n_labels'] == 4
        assert 'model_build_rng_seed' in hp and 'training_rng_seed' in hp
    # same seed -> same parameters as a direct, seeded construction
    hp = case_hparams(meta)
    hp.update(extra)
    model = ags.build_model(hp, n_datasets=1)
    ref = seeded_build(models.AE, case_hparams(meta))
    for (k, a), (_, b) in zip(model.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a, b), k
    # pretrained weights (ref aes.py:1220-1265)
    path = str(tmp_path / 'pre.pt')
    with torch.no_grad():
        ref.encoding.FF.bias.add_(1.0)
    ref.save(path)
    hp = case_hparams(meta)
    hp.update(extra)
    hp['pretrained_weights_path'] = path
    model = ags.build_model(hp, n_datasets=1)
    assert torch.equal(model.encoding.FF.bias, ref.encoding.FF.bias)
    hp = case_hparams(meta)
    hp.update(extra)
    hp['model_class'] = 'labels-images'
    with pytest.raises(NotImplementedError):
        ags.build_model(hp, n_datasets=2)
    hp['model_class'] = 'cond-vae'
    hp.pop('n_labels', None)
    with pytest.raises(ValueError):
        ags.build_model(hp, n_datasets=1)
