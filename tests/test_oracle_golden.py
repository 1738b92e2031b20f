"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors captured from the reference.

The reference's own tests do not pin forward/backward numerics of the models
(reference tests/README.md:13), so the golden vectors come from running the imported reference
in the build container (tests/golden/make_golden.py).
"""

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from tests.cases import forward_kwargs, load_case, case_hparams, case_data, seeded_build, EpsReplay, eps_list
from tests.golden_utils import checksum, checksum_close, strided_sample

CASES = ['ae_cfg1', 'ae_cfg1_b210', 'ae_cfg2', 'ae_1x64x48', 'vae_cfg1', 'betatc_cfg1',
         'condvae_cfg1', 'psvae_cfg4',
         'ae_cfg1_bn', 'ae_cfg1_bn_b210', 'vae_1x64x48_bn', 'ae_cfg1_lastff', 'aemsp_cfg1',
         'condae_cfg1', 'condae_enc_cfg1', 'ae_sessio_masks', 'ae_linear', 'ae_valid_1x30x26', 'ae_arch2_1x128x128', 'ae_2x192x160', 'ae_maxpool',
         'ae_maxpool_valid']


def _check_tensor(z, prefix, t, rtol, atol=1e-7):
    a = t.detach().cpu().numpy()
    assert checksum_close(checksum(a), z[prefix + '/checksum'], rtol), prefix
    want = z[prefix + '/full'] if prefix + '/full' in z.files else z[prefix + '/sample']
    got = a if prefix + '/full' in z.files else strided_sample(a)
    # element tolerance relative to the tensor's own scale (tiny entries carry rounding noise)
    atol = max(atol, rtol * float(np.abs(want).max())) if rtol > 0 else atol
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=prefix)


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference(name):
    torch.set_num_threads(8)
    z, meta = load_case(name)
    DS = meta.get('dataset', 0)
    hp = case_hparams(meta)
    model = seeded_build(ref_cpu.build_model, hp)
    variational = meta['model_class'] in ('vae', 'ps-vae', 'cond-vae', 'beta-tcvae')

    # (i) identical initial parameters from the same seed (bit-exact)
    sd = model.state_dict()
    want_keys = sorted(k[len('param0/'):-len('/checksum')] for k in z.files
                       if k.startswith('param0/') and k.endswith('/checksum'))
    assert sorted(sd.keys()) == want_keys
    for k, v in sd.items():
        _check_tensor(z, 'param0/' + k, v, rtol=0.0, atol=0.0)

    data = case_data(meta)
    x = data['images'][0]
    n_fwd = meta['n_fwd']

    # (ii) forward: per-layer activation checksums, x_hat, latents
    model.train()
    taps_e, taps_d = [], []
    with torch.no_grad():
        if variational:
            model.eps_fn = EpsReplay([z['fwd/eps']])
        # the extra encoder pass (for the per-layer taps) must not count as a batch-norm update
        bufs = {k: v.clone() for k, v in model.named_buffers()}
        x_enc = x[:n_fwd]
        if 'labels_sc' in data:
            x_enc = torch.cat((x_enc, data['labels_sc'][0][:n_fwd]), dim=1)
        enc_out = model.encoding(x_enc, dataset=DS, taps=taps_e)
        for k, v in model.named_buffers():
            v.copy_(bufs[k])
        kw = forward_kwargs(meta, data, n_fwd)
        out = model(x[:n_fwd], dataset=DS, **kw)
    act_keys = [k for k in z.files if k.startswith('act/encoding')]
    assert len(act_keys) == len(taps_e)
    for i, t in enumerate(taps_e):
        assert checksum_close(checksum(t.numpy()),
                              z['act/encoding.encoder.relu%d/checksum' % i], 1e-6)
    if 'fwd/x_hat' in z.files:
        np.testing.assert_allclose(out[0].numpy(), z['fwd/x_hat'], rtol=1e-5, atol=1e-6)
    else:
        np.testing.assert_allclose(out[0][:1].numpy(), z['fwd/x_hat_first'], rtol=1e-5, atol=1e-6)
    assert checksum_close(checksum(out[0].numpy()), z['fwd/x_hat/checksum'], 1e-6)
    names = {2: ['z'], 3: ['z', 'y_hat'], 4: ['z', 'mu', 'logvar'],
             5: ['z', 'mu', 'logvar', 'y_hat']}[len(out)]
    for nm, t in zip(names, out[1:]):
        np.testing.assert_allclose(t.numpy(), z['fwd/' + nm], rtol=1e-5, atol=1e-6, err_msg=nm)

    # (iii) loss dict + accumulated gradients (two chunks for the b210 case: SURVEY G2)
    model.zero_grad()
    if variational:
        model.curr_epoch = meta['curr_epoch']
        model.eps_fn = EpsReplay(eps_list(z, 'loss/eps'))
    loss = model.loss(data, dataset=DS, accumulate_grad=True)
    keys = [str(k) for k in z['loss/keys']]
    assert sorted(loss.keys()) == keys
    got = np.array([float(loss[k]) for k in keys])
    np.testing.assert_allclose(got, z['loss/vals'], rtol=1e-6, atol=1e-9)
    for k, p in model.named_parameters():
        if p.grad is not None:
            _check_tensor(z, 'grad/' + k, p.grad, rtol=2e-5, atol=1e-9)

    # (iv) three Adam(amsgrad) steps
    opt = ref_cpu.make_optimizer(model, hp)
    losses = []
    for step in range(3):
        if variational:
            model.eps_fn = EpsReplay(eps_list(z, 'adam/eps_step%d_' % step))
        losses.append(ref_cpu.train_step(model, opt, data, dataset=DS)['loss'])
    np.testing.assert_allclose(losses, z['adam/losses'], rtol=1e-6)
    for k, p in model.named_parameters():
        if p.requires_grad:
            _check_tensor(z, 'adam/param/' + k, p, rtol=1e-6, atol=1e-8)
            st = opt.state[p]
            if not st:        # never receives a gradient (AEMSP.U)
                assert 'adam/exp_avg/%s/checksum' % k not in z.files
                continue
            for sk in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
                assert checksum_close(checksum(st[sk].numpy()),
                                      z['adam/%s/%s/checksum' % (sk, k)], 1e-4), (sk, k)
    # batch-norm running statistics (train-mode forward + loss call + 3 steps)
    n_buf = 0
    for k, v in model.named_buffers():
        if 'adam/buffer/' + k in z.files:
            np.testing.assert_allclose(v.numpy().astype(np.float64), z['adam/buffer/' + k],
                                       rtol=1e-5, atol=1e-8, err_msg=k)
            n_buf += 1
    assert n_buf == len([k for k in z.files if k.startswith('adam/buffer/')])


def test_oracle_loss_known_answers():
    """The closed-form answers of the reference's tests/test_fitting/test_losses.py:8-94."""
    LN2PI = np.log(2 * np.pi)
    x = torch.rand((5, 3))
    assert ref_cpu.mse(x, x) == 0
    a = torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.float)
    b = torch.tensor([2, 3, 4, 5, 6, 7], dtype=torch.float)
    m = torch.tensor([1, 0, 1, 0, 1, 0], dtype=torch.float)
    assert ref_cpu.mse(a, b, m) == 0.5

    n_batch, n_dims = 5, 3
    x = torch.rand((n_batch, n_dims))
    assert ref_cpu.gaussian_ll(x, x) == -(0.5 * LN2PI) * n_dims
    ones, zeros = torch.ones(n_batch, n_dims), torch.zeros(n_batch, n_dims)
    mask = torch.zeros(n_batch, n_dims)
    mask[:, 0] = 1
    assert ref_cpu.gaussian_ll(ones, zeros, masks=mask) == -(0.5 * LN2PI) * n_dims - 0.5

    ll = ref_cpu.gaussian_ll(ones, zeros)
    mse_ = 2 * (-ll - 0.5 * LN2PI * n_dims) / n_dims
    assert np.allclose(ref_cpu.gaussian_ll_to_mse(ll.numpy(), n_dims), mse_.numpy())

    assert ref_cpu.kl_div_to_std_normal(torch.zeros(1, 1), torch.zeros(1, 1)) == 0

    zz, mu, lv = torch.rand(5, 3), torch.rand(5, 3), torch.rand(5, 3)
    mi, tc, dw = ref_cpu.decomposed_kl(zz, mu, lv)
    assert ref_cpu.index_code_mi(zz, mu, lv).item() == mi.item()
    assert ref_cpu.total_correlation(zz, mu, lv).item() == tc.item()
    assert ref_cpu.dimension_wise_kl_to_std_normal(zz, mu, lv).item() == dw.item()


def test_oracle_conv_decoder_matches_reference():
    """ConvDecoder (labels -> images, ref decoders.py:355-496) on a two-chunk batch."""
    torch.set_num_threads(8)
    z, meta = load_case('convdecoder_cfg1')
    hp = case_hparams(meta)
    model = seeded_build(ref_cpu.build_model, hp)
    sd = model.state_dict()
    want_keys = sorted(k[len('param0/'):-len('/checksum')] for k in z.files
                       if k.startswith('param0/') and k.endswith('/checksum'))
    assert sorted(sd.keys()) == want_keys
    for k, v in sd.items():
        _check_tensor(z, 'param0/' + k, v, rtol=0.0, atol=0.0)
    data = case_data(meta)
    model.train()
    with torch.no_grad():
        out = model(data['labels'][0][:meta['n_fwd']], dataset=0)
    np.testing.assert_allclose(out.numpy(), z['fwd/x_hat'], rtol=1e-5, atol=1e-6)
    model.zero_grad()
    loss = model.loss(data, dataset=0, accumulate_grad=True)
    assert sorted(loss.keys()) == [str(k) for k in z['loss/keys']] == ['loss']
    np.testing.assert_allclose(loss['loss'], z['loss/vals'][0], rtol=1e-6)
    for k, p in model.named_parameters():
        _check_tensor(z, 'grad/' + k, p.grad, rtol=2e-5, atol=1e-9)
    opt = ref_cpu.make_optimizer(model, hp)
    traj = [ref_cpu.train_step(model, opt, data)['loss'] for _ in range(3)]
    np.testing.assert_allclose(traj, z['adam/losses'], rtol=1e-6)
    for k, p in model.named_parameters():
        _check_tensor(z, 'adam/param/' + k, p, rtol=1e-6, atol=1e-8)


def _msps_case():
    from tests.golden_utils import make_labels, make_frames
    z, meta = load_case('mspsvae_cfg1')
    datas = []
    for i, t in enumerate(meta['n_frames']):
        datas.append({
            'images': torch.from_numpy(make_frames(t, meta['dim'], seed=1 + i))[None],
            'labels': torch.from_numpy(make_labels(t, meta['n_labels'], seed=5 + i))[None]})
    return z, meta, datas


def test_oracle_triplet_loss_known_answers():
    """losses.triplet_loss (ref losses.py:402-511) for 2, 3 and 4 sessions."""
    z, _ = load_case('mspsvae_cfg1')
    obj = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    for n_sess in (2, 3, 4):
        np.random.seed(30 + n_sess)
        got = ref_cpu.triplet_loss(obj, torch.from_numpy(z['triplet/%d/z' % n_sess]),
                                   z['triplet/%d/ids' % n_sess]).item()
        np.testing.assert_allclose(got, float(z['triplet/%d/val' % n_sess]), rtol=1e-6)
    with pytest.raises(NotImplementedError):
        ref_cpu.triplet_loss(obj, torch.zeros((10, 3)), np.arange(10) % 5)


def test_oracle_mspsvae_matches_reference():
    """Multi-session PS-VAE (ref vaes.py:849-1098): parameters, forward, the 13-key loss dict of a
    two-session batch (with the triplet term), gradients, the single-session (validation) loss
    and three Adam steps."""
    torch.set_num_threads(8)
    z, meta, datas = _msps_case()
    hp = case_hparams(meta)
    model = seeded_build(ref_cpu.build_model, hp)
    sd = model.state_dict()
    want_keys = sorted(k[len('param0/'):-len('/checksum')] for k in z.files
                       if k.startswith('param0/') and k.endswith('/checksum'))
    assert sorted(sd.keys()) == want_keys
    for k, v in sd.items():
        _check_tensor(z, 'param0/' + k, v, rtol=0.0, atol=0.0)
    model.train()
    model.eps_fn = EpsReplay([z['fwd/eps']])
    with torch.no_grad():
        out = model(datas[0]['images'][0][:meta['n_fwd']], dataset=None)
    for nm, t in zip(['x_hat', 'z', 'mu', 'logvar', 'y_hat'], out):
        np.testing.assert_allclose(t.numpy(), z['fwd/' + nm], rtol=1e-5, atol=1e-6, err_msg=nm)
    model.curr_epoch = meta['curr_epoch']
    model.eps_fn = EpsReplay([z['loss/eps0']])
    np.random.seed(11)
    model.zero_grad()
    loss = model.loss(datas, dataset=meta['sess'], accumulate_grad=True)
    keys = [str(k) for k in z['loss/keys']]
    assert sorted(loss.keys()) == keys and 'loss_triplet' in keys and 'delta' in keys
    np.testing.assert_allclose([float(loss[k]) for k in keys], z['loss/vals'], rtol=2e-6,
                               atol=1e-9)
    for k, p in model.named_parameters():
        if p.grad is not None:
            _check_tensor(z, 'grad/' + k, p.grad, rtol=2e-5, atol=1e-9)
        else:
            assert 'grad/' + k + '/checksum' not in z.files
    model.eps_fn = EpsReplay([z['loss1/eps0']])
    loss1 = model.loss(datas[0], dataset=1, accumulate_grad=False)
    keys1 = [str(k) for k in z['loss1/keys']]
    assert sorted(loss1.keys()) == keys1 and loss1['loss_triplet'] == 0
    np.testing.assert_allclose([float(loss1[k]) for k in keys1], z['loss1/vals'], rtol=2e-6,
                               atol=1e-9)
    opt = ref_cpu.make_optimizer(model, hp)
    traj = []
    for step in range(3):
        model.eps_fn = EpsReplay([z['adam/eps_step%d_0' % step]])
        np.random.seed(20 + step)
        model.train()
        opt.zero_grad()
        traj.append(model.loss(datas, dataset=meta['sess'], accumulate_grad=True)['loss'])
        opt.step()
    np.testing.assert_allclose(traj, z['adam/losses'], rtol=2e-6)
    for k, p in model.named_parameters():
        if p.requires_grad:
            _check_tensor(z, 'adam/param/' + k, p, rtol=1e-6, atol=1e-8)
