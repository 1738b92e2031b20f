"""Model-level parity on the MI355X: the HIP-backed models (behavenet_amd.models) against

  (a) the CPU oracle (oracle/ref_cpu.py) on the same seeded inputs -- full tensors;
  (b) the committed golden vectors captured from the imported reference (tests/golden/*.npz);
  (c) size-independent properties at BASELINE's full size (batch 256, two chunks).

Tolerance: 1e-4 relative (BASELINE.json north_star), applied as in tests/test_gpu_kernels.py.
"""

import copy
import json
import os
import pickle

import numpy as np
import pytest
import torch

from behavenet_amd import _hip
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.fitting.training import fit
from behavenet_amd.fitting import losses
from behavenet_amd.models import AE, ConditionalAE, AEMSP, VAE, ConditionalVAE, BetaTCVAE, PSVAE, MSPSVAE, ConvDecoder
from behavenet_amd.models import vaes as hip_vaes
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from oracle import ref_cpu
from tests.cases import forward_kwargs, load_case, case_hparams, case_data, seeded_build, EpsReplay, eps_list
from tests.golden_utils import base_hparams, checksum, checksum_close, make_frames
from tests.test_gpu_kernels import close
from tests.branches import record_branches, BranchReplay

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BUILDERS = {'ae': AE, 'cond-ae': ConditionalAE, 'vae': VAE, 'cond-vae': ConditionalVAE, 'beta-tcvae': BetaTCVAE,
            'ps-vae': PSVAE, 'msps-vae': MSPSVAE, 'cond-ae-msp': AEMSP,
            'conv-decoder': ConvDecoder}
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _pair(meta):
    """(hip model on the GPU, oracle on the CPU) with identical parameters."""
    hp_h, hp_o = case_hparams(meta), case_hparams(meta)
    hip = seeded_build(BUILDERS[meta['model_class']], hp_h).to(DEV)
    ora = seeded_build(ref_cpu.build_model, hp_o)
    for (k1, v1), (k2, v2) in zip(hip.state_dict().items(), ora.state_dict().items()):
        assert k1 == k2 and torch.equal(v1.cpu(), v2)
    return hip, ora, hp_h


def _bias_before_batchnorm(key, names):
    """A conv bias that feeds a batch norm: its gradient is analytically zero (the layer subtracts
    the channel mean again), so what any implementation computes for it is rounding noise."""
    if not key.endswith('.bias') or 'conv' not in key:
        return False
    parts = key.split('.')
    head = '.'.join(parts[:-2])
    num = ''.join(ch for ch in parts[-2].split('_')[0] if ch.isdigit())
    return '%s.batchnorm%s.weight' % (head, num) in names


def grads_close_on_same_branches(hip, ora64, name, tol=2e-5):
    """Gradients of the HIP model against a float64 oracle that was run on the branch pattern the
    HIP forward took (tests/branches.py): what remains is rounding, so the bar is 2e-5 of each
    tensor's maximum instead of the 1e-4 of the plain comparisons."""
    for (k, ph), (_, po) in zip(hip.named_parameters(), ora64.named_parameters()):
        if po.grad is None:
            assert ph.grad is None or not ph.requires_grad
            continue
        w = po.grad.numpy()
        err = np.abs(ph.grad.cpu().double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
        assert err <= tol, '%s grad %s: normalised max err %.3e' % (name, k, err)


def _assert_grads_on_device_branches(hip, meta, data_c, data_g, dataset, name):
    ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
    ora64.train()
    hip.zero_grad()
    with record_branches(hip) as rec:
        hip.loss(data_g, dataset=dataset, accumulate_grad=True)
    with BranchReplay(rec) as br:
        ora64.loss({k: v.double() for k, v in data_c.items()}, dataset=dataset,
                   accumulate_grad=True)
    br.assert_only_ties()
    assert len(br.flips) > 0, 'gradient mismatch without a branch difference'
    grads_close_on_same_branches(hip, ora64, name)


CASES = ['ae_cfg1', 'ae_cfg1_b210', 'ae_cfg2', 'ae_1x64x48', 'vae_cfg1', 'betatc_cfg1',
         'condvae_cfg1', 'psvae_cfg4', 'ae_cfg1_bn', 'ae_cfg1_bn_b210', 'vae_1x64x48_bn', 'ae_cfg1_lastff', 'aemsp_cfg1',
         'condae_cfg1', 'condae_enc_cfg1', 'ae_sessio_masks', 'ae_linear', 'ae_valid_1x30x26', 'ae_arch2_1x128x128', 'ae_2x192x160', 'ae_maxpool',
         'ae_maxpool_valid']


# which cases took the tie fallback of the test below in this run, and which are known to
TIE_FALLBACK_TAKEN = []
TIE_FALLBACK_KNOWN = {'condae_cfg1'}     # one flip in dec.convT1 on the current tilings (round 4)
TIE_FALLBACK_SEEN_ALL = []


@pytest.mark.parametrize('name', CASES)
def test_forward_loss_grads_vs_oracle_and_golden(name):
    if name == CASES[-1]:
        TIE_FALLBACK_SEEN_ALL.append(True)
    z, meta = load_case(name)
    DS = meta.get('dataset', 0)
    hip, ora, hp = _pair(meta)
    variational = meta['model_class'] in ('vae', 'ps-vae', 'cond-vae', 'beta-tcvae')
    data_c = case_data(meta)
    data_g = {k: v.to(DEV) for k, v in data_c.items()}
    n_fwd = meta['n_fwd']
    kw_c, kw_g = forward_kwargs(meta, data_c, n_fwd), forward_kwargs(meta, data_g, n_fwd)

    # forward (same eps on both sides, the one the reference drew)
    hip.train()
    ora.train()
    if variational:
        ora.eps_fn = EpsReplay([z['fwd/eps']])
        hip_vaes.set_eps_provider(EpsReplay([z['fwd/eps']], DEV))
    try:
        with torch.no_grad():
            out_o = ora(data_c['images'][0][:n_fwd], dataset=DS, **kw_c)
            out_h = hip(data_g['images'][0][:n_fwd], dataset=DS, **kw_g)
        for i, (a, b) in enumerate(zip(out_h, out_o)):
            close(a, b, name='%s fwd out%d' % (name, i))
        # against the reference itself
        if 'fwd/x_hat' in z.files:
            close(out_h[0], torch.from_numpy(z['fwd/x_hat']), name=name + ' x_hat golden')
        assert checksum_close(checksum(out_h[0].cpu().numpy()), z['fwd/x_hat/checksum'], 2e-5)
        close(out_h[1], torch.from_numpy(z['fwd/z']), name=name + ' z golden')

        # loss dict and accumulated gradients
        hip.zero_grad()
        ora.zero_grad()
        if variational:
            hip.curr_epoch = ora.curr_epoch = meta['curr_epoch']
            ora.eps_fn = EpsReplay(eps_list(z, 'loss/eps'))
            hip_vaes.set_eps_provider(EpsReplay(eps_list(z, 'loss/eps'), DEV))
        loss_o = ora.loss(data_c, dataset=DS, accumulate_grad=True)
        with record_branches(hip) as rec:
            loss_h = hip.loss(data_g, dataset=DS, accumulate_grad=True)
    finally:
        hip_vaes.set_eps_provider(None)
    assert sorted(loss_h.keys()) == sorted(loss_o.keys()) == [str(k) for k in z['loss/keys']]
    for k, want in zip([str(k) for k in z['loss/keys']], z['loss/vals']):
        assert loss_h[k] == pytest.approx(loss_o[k], rel=1e-4, abs=1e-7), k
        assert loss_h[k] == pytest.approx(float(want), rel=1e-4, abs=1e-7), k
    # batch-norm cases (ae_cfg1_bn_b210 normalises 10 values per channel in its second chunk): ONE
    # LeakyReLU output at a tie (|x| ~ 1e-7 of its layer) on the other branch moves every upstream
    # gradient by ~1e-3, in float64 as much as in fp32 (tools/diag_bn_cond.py) -- the gradient is a
    # discontinuous function there, not an ill-conditioned one.  So these cases are compared on the
    # branch pattern the device took: float64 oracle on that pattern, 2e-5 of each tensor's
    # maximum, and the pattern may differ from the oracle's own at ties only.
    g64 = None
    if meta['extra_hp'].get('ae_batch_norm') and not variational:
        ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
        ora64.train()
        with BranchReplay(rec) as br:
            ora64.loss({k: v.double() for k, v in data_c.items()}, dataset=DS, accumulate_grad=True)
        br.assert_only_ties()
        g64 = {k: p.grad for k, p in ora64.named_parameters()}
    names = set(k for k, _ in ora.named_parameters())
    for (k, ph), (_, po) in zip(hip.named_parameters(), ora.named_parameters()):
        if po.grad is None:
            assert ph.grad is None or not ph.requires_grad
            continue
        if _bias_before_batchnorm(k, names):
            wscale = float(dict(ora.named_parameters())[k[:-4] + 'weight'].grad.abs().max())
            assert float(ph.grad.abs().max()) <= 1e-4 * wscale, k
            assert float(po.grad.abs().max()) <= 1e-4 * wscale, k
            continue
        if g64 is not None:
            w64 = g64[k].numpy()
            err = np.abs(ph.grad.cpu().double().numpy() - w64).max() / max(np.abs(w64).max(), 1e-30)
            assert err <= 2e-5, '%s grad %s: normalised max err %.3e on the device branches' % (
                name, k, err)
        else:
            try:
                close(ph.grad, po.grad, name='%s grad %s' % (name, k))
                ctol = 1e-4
                assert checksum_close(checksum(ph.grad.cpu().numpy()),
                                      z['grad/' + k + '/checksum'], ctol), k
            except AssertionError:
                # a LeakyReLU pre-activation at a tie took the other branch (see
                # tests/branches.py): accept only if the gradients agree with the float64 oracle
                # on the device's branch pattern and that pattern differs at ties only
                if variational or meta['extra_hp'].get('ae_batch_norm'):
                    raise
                _assert_grads_on_device_branches(hip, meta, data_c, data_g, DS, name)
                TIE_FALLBACK_TAKEN.append(name)
                break
    # batch-norm running statistics after the same call sequence (one forward, one loss call)
    for (k, bh), (_, bo) in zip(hip.named_buffers(), ora.named_buffers()):
        if 'running_' in k or 'num_batches' in k:
            close(bh.float(), bo.float(), name='%s buffer %s' % (name, k))


def test_tie_fallback_was_taken_by_the_known_cases_only():
    """The escape hatch above (direct comparison fails -> compare on the device's LeakyReLU
    branches) is for the cases KNOWN to have a pre-activation at a tie on this library's tilings.
    A regression that turned other cases into "tie" cases must fail here, not print dots."""
    if not TIE_FALLBACK_SEEN_ALL:
        pytest.skip('needs the whole of test_forward_loss_grads_vs_oracle_and_golden in this run')
    print('tie fallback taken by: %s' % sorted(TIE_FALLBACK_TAKEN))
    extra = set(TIE_FALLBACK_TAKEN) - TIE_FALLBACK_KNOWN
    assert not extra, 'cases that newly need the tie fallback: %s' % sorted(extra)


@pytest.mark.parametrize('name', ['ae_cfg1', 'ae_cfg2', 'vae_cfg1', 'ae_cfg1_bn'])
def test_adam_trajectory_vs_oracle_and_golden(name):
    z, meta = load_case(name)
    hip, ora, hp = _pair(meta)
    variational = meta['model_class'] != 'ae'
    data_c = case_data(meta)
    data_g = {k: v.to(DEV) for k, v in data_c.items()}
    opt_o = ref_cpu.make_optimizer(ora, hp)
    opt_h = FlatAdamAMSGrad(hip.get_parameters(), lr=hp['learning_rate'],
                            weight_decay=hp.get('l2_reg', 0))
    if variational:
        hip.curr_epoch = ora.curr_epoch = meta['curr_epoch']
    losses_h = []
    try:
        for step in range(3):
            if variational:
                ora.eps_fn = EpsReplay(eps_list(z, 'adam/eps_step%d_' % step))
                hip_vaes.set_eps_provider(EpsReplay(eps_list(z, 'adam/eps_step%d_' % step), DEV))
            ref_cpu.train_step(ora, opt_o, data_c)
            hip.train()
            opt_h.zero_grad()
            losses_h.append(hip.loss(data_g, dataset=0, accumulate_grad=True)['loss'])
            opt_h.step()
    finally:
        hip_vaes.set_eps_provider(None)
    np.testing.assert_allclose(losses_h, z['adam/losses'], rtol=1e-4)
    # Adam's update m/sqrt(v) is scale-free, so fp32 summation-order noise in a near-zero
    # gradient moves a weight by a fraction of lr per step: allow 2 % of the maximal travel
    # (n_steps * lr) per element, on top of the 1e-4 relative tolerance.
    travel = 0.02 * 3 * hp['learning_rate']
    names = set(k for k, _ in ora.named_parameters())
    bn_case = bool(meta['extra_hp'].get('ae_batch_norm'))
    for i, ((k, ph), (_, po)) in enumerate(zip(hip.named_parameters(), ora.named_parameters())):
        if not po.requires_grad:
            assert torch.equal(ph.cpu(), po)
            continue
        got, want = ph.detach().cpu().numpy(), po.detach().numpy()
        if _bias_before_batchnorm(k, names):
            # zero-gradient parameter: Adam normalises rounding noise into +-lr steps, in the
            # reference as much as here; it can only be bounded by the full travel
            np.testing.assert_allclose(got, want, rtol=0, atol=3.03 * hp['learning_rate'],
                                       err_msg=k)
            continue
        if bn_case:
            # batch norm over 8 values per channel: gradient noise ~1e-4 of the tensor's scale in
            # either implementation, which Adam turns into sign flips for the elements below
            # it.  Bound every element by the full travel and the outliers to a small fraction.
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=3.03 * hp['learning_rate'],
                                       err_msg=k)
            frac = np.mean(np.abs(got - want) > travel + 1e-4 * np.abs(want))
            assert frac <= 0.05, (k, frac)
            continue
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=travel, err_msg=k)
        ref = z['adam/param/' + k + '/checksum']
        assert abs(checksum(got)[0] - ref[0]) <= 1e-4 * ref[1] + travel * got.size, k
    # optimizer state of the first and the largest tensor
    for i in (0, 8):
        m, v, vmax = opt_h.state_tensors(i)
        st = opt_o.state[list(ora.get_parameters())[i]]
        tol = 2e-3 if bn_case else 1e-4
        close(m, st['exp_avg'], name='exp_avg', norm_tol=tol)
        close(v, st['exp_avg_sq'], name='exp_avg_sq', norm_tol=tol)
        close(vmax, st['max_exp_avg_sq'], name='max_exp_avg_sq', norm_tol=tol)
    # running statistics: the conv biases under a batch norm random-walk by up to 3*lr (above)
    # and shift the channel means with them
    for (k, bh), (_, bo) in zip(hip.named_buffers(), ora.named_buffers()):
        if 'running_' in k or 'num_batches' in k:
            np.testing.assert_allclose(bh.float().cpu().numpy(), bo.float().numpy(), rtol=1e-3,
                                       atol=3.03 * hp['learning_rate'], err_msg=k)


def test_fit_on_gpu_reproduces_reference_rows(tmp_path):
    with open(os.path.join(GOLDEN, 'fit_cfg1.json')) as f:
        want = json.load(f)
    dim = [1, 32, 32]
    arch = load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    hp.update({'expt_dir': str(tmp_path), 'max_n_epochs': 2, 'min_n_epochs': 0,
               'val_check_interval': 1, 'enable_early_stop': False, 'early_stop_history': 10,
               'rng_seed_train': 0, 'export_latents': True, 'progress_bar': False,
               'device': 'cuda'})
    os.makedirs(os.path.join(str(tmp_path), 'version_0'))
    sess = SyntheticSession(10, 32, dim, seed=0, trial_splits='8;1;1;0')
    gen = SyntheticSessionsGenerator([sess], device=DEV, placement='device_u8')
    torch.manual_seed(0)
    model = AE(hp).to(DEV)
    model.version = 0

    class Exp(object):
        version = 0
        rows = []

        def log(self, row):
            self.rows.append(dict(row))

        def save(self):
            pass
    exp = Exp()
    best = fit(hp, model, gen, exp, method='ae')
    assert len(exp.rows) == len(want['rows'])
    for got, ref in zip(exp.rows, want['rows']):
        assert set(got.keys()) == set(ref.keys())
        for k, v in ref.items():
            if isinstance(v, float):
                assert got[k] == pytest.approx(v, rel=1e-4), k
            else:
                assert got[k] == v, k
    for k, v in model.state_dict().items():
        # 16 Adam steps: 2 % of the maximal travel per element (see the trajectory test above)
        ref = want['final_param_checksums'][k]
        got = checksum(v.cpu().numpy())
        assert abs(got[0] - ref[0]) <= 1e-4 * ref[1] + 0.02 * 16 * 1e-4 * v.numel(), k

    # checkpoint is key-compatible with the reference / the oracle
    sd = torch.load(os.path.join(str(tmp_path), 'version_0', 'best_val_model.pt'),
                    map_location='cpu')
    ora = ref_cpu.AE(case_hparams({'dim': dim, 'n_lat': 8, 'model_class': 'ae',
                                   'extra_hp': {}, 'n_labels': 0}))
    ora.load_state_dict(sd)

    # export_latents: pickle schema + values against the oracle encoder
    pkl = os.path.join(str(tmp_path), 'version_0', 'lab_expt_animal_sess_latents.pkl')
    with open(pkl, 'rb') as f:
        lat = pickle.load(f)
    assert set(lat.keys()) == {'latents', 'trials'}
    assert len(lat['latents']) == 10 and set(lat['trials'].keys()) == {'train', 'val', 'test'}
    ora.load_state_dict({k: v.cpu() for k, v in best.state_dict().items()})
    x0 = torch.from_numpy(sess.images_u8[0].astype(np.float32) / 255)
    with torch.no_grad():
        z0 = ora.encoding(x0)[0]
    assert lat['latents'][0].shape == (32, 8)
    close(torch.from_numpy(lat['latents'][0]), z0, name='exported latents')


def test_full_size_batch256_properties():
    """BASELINE config 2 at full size (1x128x128, 12 latents, batch 256 = chunks 200+56)."""
    arch = load_handcrafted_arch([1, 128, 128], 12, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    torch.manual_seed(0)
    model = AE(hp).to(DEV)
    x = torch.from_numpy(make_frames(256, [1, 128, 128], seed=11)).to(DEV)
    data = {'images': x[None]}

    # (1) determinism: two runs give bit-identical loss and gradients
    def run():
        model.zero_grad(set_to_none=True)
        out = model.loss(data, dataset=0, accumulate_grad=True)
        return out['loss'], [p.grad.clone() for p in model.parameters()]
    l1, g1 = run()
    l2, g2 = run()
    assert l1 == l2
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)

    # (2) chunk additivity: grad(batch) = grad(first 200) + grad(last 56), and the
    #     reported loss is the frame-weighted mean of the chunk losses (G3)
    def part(lo, hi):
        model.zero_grad(set_to_none=True)
        with record_branches(model) as rec:
            l = model.loss({'images': x[None, lo:hi]}, dataset=0, accumulate_grad=True)['loss']
        return l, [p.grad.clone() for p in model.parameters()], rec
    la, ga, ra = part(0, 200)
    lb, gb, rb = part(200, 256)
    assert l1 == pytest.approx((la * 200 + lb * 56) / 256, rel=1e-6)
    # The 256-frame pass and the 200- / 56-frame passes pick different tilings, so a frame's
    # activations differ in the last bit and, on noise frames, a few LeakyReLU pre-activations
    # within rounding distance of 0 take the other sign: isolated gradient elements of the two
    # sides then differ by ~1e-4 of the tensor's scale (with the shape-agnostic kernels, whose
    # arithmetic does not depend on the batch size, they agree to 1e-7: tools/diag_whole.py).
    # (Round 4: the 56-frame pass splits some reductions over workgroups, a few more flips: 1.04e-4.)
    # The property is therefore stated (a) in the L2 norm between the two device results and
    # (b) exactly: the float64 oracle's chunk loop, run on the branch pattern the two partial
    # passes took (tests/branches.py), must give ga + gb to 2e-5 -- the whole-batch pass is held
    # to the same oracle at the same 2e-5 in tests/test_gpu_bench_sizes.py.
    for g, a, b in zip(g1, ga, gb):
        ref = (a + b).double()
        err = (g.double() - ref)
        assert float(err.norm() / ref.norm().clamp_min(1e-30)) <= 2e-4, 'chunk additivity (L2)'
    pattern = {st: [None if u is None else torch.cat([u, v], 0) for u, v in zip(ra[st], rb[st])]
               for st in ra}
    torch.manual_seed(0)
    ora64 = ref_cpu.AE(case_hparams({'dim': [1, 128, 128], 'n_lat': 12, 'model_class': 'ae',
                                     'extra_hp': {}, 'n_labels': 0})).double()
    ora64.load_state_dict({k: v.cpu().double() for k, v in model.state_dict().items()})
    with BranchReplay(pattern) as br:
        l64 = ora64.loss({'images': x.cpu().double()[None]}, dataset=0, accumulate_grad=True)['loss']
    br.assert_only_ties()
    assert l1 == pytest.approx(l64, rel=1e-5)
    for (k, po), a, b in zip(ora64.named_parameters(), ga, gb):
        w = po.grad.numpy()
        err = np.abs((a + b).cpu().double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
        assert err <= 2e-5, 'chunk additivity vs float64 oracle, %s: %.3e' % (k, err)

    # (3) frame independence: reconstructing a sub-batch gives the same frames
    with torch.no_grad():
        full, _ = model(x[:64])
        part, _ = model(x[32:40])
    close(part, full[32:40], norm_tol=1e-6, name='frame independence')

    # (4) the loss value equals the oracle's on a 16-frame sample of the same batch
    ora = ref_cpu.AE(case_hparams({'dim': [1, 128, 128], 'n_lat': 12, 'model_class': 'ae',
                                   'extra_hp': {}, 'n_labels': 0}))
    ora.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    lo = ora.loss({'images': x[None, :16].cpu()}, accumulate_grad=False)['loss']
    lh = model.loss({'images': x[None, :16]}, accumulate_grad=False)['loss']
    assert lh == pytest.approx(lo, rel=1e-5)


@pytest.mark.parametrize('model_class', ['ae', 'vae'])
def test_stream_schedule_does_not_change_results(model_class):
    """The production schedule (gradients accumulated in place into the optimizer's flat arena,
    weight gradients on a side stream, odd chunks on an auxiliary stream) must give bit-identical
    losses and gradients to the plain one-stream schedule, run after run."""
    from behavenet_amd import hip_functions as hf
    arch = load_handcrafted_arch([1, 128, 128], 12, None, check_memory=False)
    extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10} \
        if model_class == 'vae' else None
    hp = base_hparams(arch, model_class, extra)
    torch.manual_seed(0)
    model = BUILDERS[model_class](hp).to(DEV)
    opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4, weight_decay=0)
    x = torch.from_numpy(make_frames(456, [1, 128, 128], seed=12)).to(DEV)   # chunks 200+200+56
    data = {'images': x[None]}
    eps = [torch.randn((n, 12), generator=torch.Generator().manual_seed(5 + i)).to(DEV)
           for i, n in enumerate((200, 200, 56))]

    def run(side, chunks):
        hf._use_side_stream, hf._use_chunk_streams = side, chunks
        try:
            if model_class == 'vae':
                it = iter(eps)
                hip_vaes.set_eps_provider(lambda like: next(it))
            opt.zero_grad()
            out = model.loss(data, dataset=0, accumulate_grad=True)
            torch.cuda.synchronize()
            return out, opt.flat_g.clone()
        finally:
            hf._use_side_stream, hf._use_chunk_streams = True, True
            hip_vaes.set_eps_provider(None)

    ref_out, ref_g = run(False, False)
    assert float(ref_g.abs().max()) > 0
    for side, chunks in ((True, False), (True, True), (True, True), (True, True)):
        out, g = run(side, chunks)
        assert out == ref_out, (side, chunks)
        assert torch.equal(g, ref_g), (side, chunks, float((g - ref_g).abs().max()))


@pytest.mark.parametrize('model_class', ['vae', 'beta-tcvae', 'ps-vae', 'cond-vae'])
def test_multichunk_variational_vs_oracle(model_class):
    """Two-chunk batches (200 + 10 frames) of the variational models: the single-pass schedule
    (latents sampled and every loss term normalised per chunk) against the oracle's
    chunk-by-chunk loop, same eps per chunk: loss dict and accumulated gradients."""
    from tests.golden_utils import make_labels
    dim, n_lat, n_frames = [1, 32, 32], 8, 210
    extra = {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10,
             'beta_tcvae.beta': 3.0, 'beta_tcvae.beta_anneal_epochs': 5,
             'ps_vae.alpha': 10, 'ps_vae.beta': 5, 'ps_vae.anneal_epochs': 5,
             'conditional_encoder': False}
    n_labels = 4 if model_class in ('ps-vae', 'cond-vae') else 0
    meta = {'dim': dim, 'n_lat': n_lat, 'model_class': model_class, 'extra_hp': extra,
            'n_labels': n_labels, 'n_frames': n_frames}
    hip, ora, hp = _pair(meta)
    data_c = case_data(meta)
    data_g = {k: v.to(DEV) for k, v in data_c.items()}
    g = torch.Generator().manual_seed(9)
    eps = [torch.randn((n, n_lat), generator=g).numpy() for n in (200, 10)]
    # float64 oracle: on 210 noise frames a LeakyReLU pre-activation within fp32 rounding of zero
    # is the rule, not the exception (for cond-vae with this seed: -1e-10 in float64, +9e-10 on
    # the device, layer maximum 4e-2, and the decoder gradients move by 1e-3), so the gradients
    # are compared on the branch pattern the device took, and that pattern is checked to differ
    # from the oracle's own only at such ties
    ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
    data64 = {k: v.double() for k, v in data_c.items()}
    for m in (hip, ora, ora64):
        m.train()
        m.curr_epoch = 3
    ora.eps_fn = EpsReplay(eps)
    ora64.eps_fn = EpsReplay([e.astype(np.float64) for e in eps])
    hip_vaes.set_eps_provider(EpsReplay(eps, DEV))
    try:
        hip.zero_grad()
        ora.zero_grad()
        ora64.zero_grad()
        loss_o = ora.loss(data_c, dataset=0, accumulate_grad=False)
        with record_branches(hip) as rec:
            loss_h = hip.loss(data_g, dataset=0, accumulate_grad=True)
        with BranchReplay(rec) as br:
            ora64.loss(data64, dataset=0, accumulate_grad=True)
    finally:
        hip_vaes.set_eps_provider(None)
    assert sorted(rec.keys()) == ['decoding', 'encoding']
    br.assert_only_ties()
    assert sorted(loss_h.keys()) == sorted(loss_o.keys())
    for k in loss_o:
        assert loss_h[k] == pytest.approx(loss_o[k], rel=1e-4, abs=1e-6), k
    grads_close_on_same_branches(hip, ora64, model_class)


@pytest.mark.parametrize('model_class', ['cond-ae-msp', 'cond-ae'])
def test_multichunk_label_models_vs_oracle(model_class):
    """Two-chunk batches (200 + 10 frames) of the label-conditioned deterministic models: the
    single-pass schedule against the oracle's chunk loop (loss dict, gradients on the device's
    LeakyReLU branch pattern, see tests/branches.py)."""
    extra = {'msp.alpha': 0.05, 'conditional_encoder': False}
    meta = {'dim': [1, 32, 32], 'n_lat': 8, 'model_class': model_class, 'extra_hp': extra,
            'n_labels': 4, 'n_frames': 210}
    hip, ora, hp = _pair(meta)
    data_c = case_data(meta)
    data_g = {k: v.to(DEV) for k, v in data_c.items()}
    ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
    for m in (hip, ora, ora64):
        m.train()
    hip.zero_grad()
    loss_o = ora.loss(data_c, dataset=0, accumulate_grad=False)
    with record_branches(hip) as rec:
        loss_h = hip.loss(data_g, dataset=0, accumulate_grad=True)
    with BranchReplay(rec) as br:
        ora64.loss({k: v.double() for k, v in data_c.items()}, dataset=0, accumulate_grad=True)
    br.assert_only_ties()
    assert sorted(loss_h.keys()) == sorted(loss_o.keys())
    for k in loss_o:
        assert loss_h[k] == pytest.approx(loss_o[k], rel=1e-4, abs=1e-6), k
    grads_close_on_same_branches(hip, ora64, model_class)


def test_mspsvae_vs_oracle_and_golden():
    """Multi-session PS-VAE (ref vaes.py:849-1098) on a two-session batch (18 + 15 frames):
    forward, the 13-key loss dict incl. the triplet term (same numpy permutations), gradients,
    the single-session (validation) loss and the Adam trajectory, against the oracle and the
    vectors recorded from the reference."""
    from tests.test_oracle_golden import _msps_case
    z, meta, datas_c = _msps_case()
    hip, ora, hp = _pair(meta)
    datas_g = [{k: v.to(DEV) for k, v in d.items()} for d in datas_c]
    sess = meta['sess']
    n_fwd = meta['n_fwd']
    hip.train()
    ora.train()
    try:
        ora.eps_fn = EpsReplay([z['fwd/eps']])
        hip_vaes.set_eps_provider(EpsReplay([z['fwd/eps']], DEV))
        with torch.no_grad():
            out_o = ora(datas_c[0]['images'][0][:n_fwd], dataset=None)
            out_h = hip(datas_g[0]['images'][0][:n_fwd], dataset=None)
        for nm, a, b in zip(['x_hat', 'z', 'mu', 'logvar', 'y_hat'], out_h, out_o):
            close(a, b, name='mspsvae fwd ' + nm)
            close(a, torch.from_numpy(z['fwd/' + nm]), name='mspsvae fwd golden ' + nm)

        hip.curr_epoch = ora.curr_epoch = meta['curr_epoch']
        ora.eps_fn = EpsReplay([z['loss/eps0']])
        hip_vaes.set_eps_provider(EpsReplay([z['loss/eps0']], DEV))
        hip.zero_grad()
        ora.zero_grad()
        np.random.seed(11)
        loss_o = ora.loss(datas_c, dataset=sess, accumulate_grad=True)
        np.random.seed(11)
        with record_branches(hip) as rec:
            loss_h = hip.loss(datas_g, dataset=sess, accumulate_grad=True)
        keys = [str(k) for k in z['loss/keys']]
        assert sorted(loss_h.keys()) == sorted(loss_o.keys()) == keys
        for k, want in zip(keys, z['loss/vals']):
            assert loss_h[k] == pytest.approx(loss_o[k], rel=1e-4, abs=1e-6), k
            assert loss_h[k] == pytest.approx(float(want), rel=1e-4, abs=1e-6), k
        assert loss_h['loss_triplet'] > 0
        for (k, ph), (_, po) in zip(hip.named_parameters(), ora.named_parameters()):
            if po.grad is None:
                assert ph.grad is None or not ph.requires_grad, k
                continue
            if k == 'encoding.C.bias':
                # a shift of all background latents leaves every triplet / pairwise distance
                # unchanged: the O(delta) summands of this gradient cancel analytically and what
                # is left is the (1e-5 times smaller) reconstruction part plus rounding noise of
                # the cancellation -- judged against the size of the summands
                tol = 1e-5 * hp['ps_vae.delta']
                assert float((ph.grad.cpu() - po.grad).abs().max()) <= tol, k
                assert float(po.grad.abs().max()) <= 100 * tol
                continue
            try:
                close(ph.grad, po.grad, name='mspsvae grad ' + k)
                assert checksum_close(checksum(ph.grad.cpu().numpy()),
                                      z['grad/' + k + '/checksum'], 1e-4), k
            except AssertionError:
                # a LeakyReLU tie on the other branch (tests/branches.py): accept only if all
                # gradients agree with the float64 oracle on the device's branch pattern and
                # that pattern differs from the oracle's own at ties only
                ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
                ora64.train()
                ora64.curr_epoch = meta['curr_epoch']
                ora64.eps_fn = EpsReplay([z['loss/eps0']])
                np.random.seed(11)
                with BranchReplay(rec) as br:
                    ora64.loss([{kk: v.double() for kk, v in d.items()} for d in datas_c],
                               dataset=sess, accumulate_grad=True)
                br.assert_only_ties()
                assert len(br.flips) > 0, 'gradient mismatch without a branch difference'
                for (k2, p2), (_, p64) in zip(hip.named_parameters(), ora64.named_parameters()):
                    if p64.grad is None or k2 == 'encoding.C.bias':
                        continue
                    w = p64.grad.numpy()
                    err = np.abs(p2.grad.cpu().double().numpy() - w).max() / max(np.abs(w).max(), 1e-30)
                    # (1e-4, the stated tolerance, not the 2e-5 of the other same-branch
                    # comparisons: the triplet term's O(delta) summands cancel in the encoder
                    # gradients, see encoding.C.bias above -- measured 2.9e-5 at conv0.weight)
                    assert err <= 1e-4, 'mspsvae grad %s: %.3e on the device branches' % (k2, err)
                break

        # one session (validation): no triplet term, key reported as 0, no gradient side effects
        hip_vaes.set_eps_provider(EpsReplay([z['loss1/eps0']], DEV))
        hip.zero_grad()
        loss1 = hip.loss(datas_g[0], dataset=1, accumulate_grad=False)
        keys1 = [str(k) for k in z['loss1/keys']]
        assert sorted(loss1.keys()) == keys1 and loss1['loss_triplet'] == 0
        for k, want in zip(keys1, z['loss1/vals']):
            assert loss1[k] == pytest.approx(float(want), rel=1e-4, abs=1e-6), k
        assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in hip.parameters())

        opt = FlatAdamAMSGrad(hip.get_parameters(), lr=hp['learning_rate'],
                              weight_decay=hp.get('l2_reg', 0))
        traj = []
        for step in range(3):
            hip_vaes.set_eps_provider(EpsReplay([z['adam/eps_step%d_0' % step]], DEV))
            np.random.seed(20 + step)
            opt.zero_grad()
            traj.append(hip.loss(datas_g, dataset=sess, accumulate_grad=True)['loss'])
            opt.step()
        np.testing.assert_allclose(traj, z['adam/losses'], rtol=1e-4)
    finally:
        hip_vaes.set_eps_provider(None)
    # latents handed to downstream tools: [z_s | z_b | z_u], supervised block through D
    with torch.no_grad():
        lat = hip.get_transformed_latents(datas_g[0]['images'][0][:4], as_numpy=True)
    assert lat.shape == (4, meta['n_lat'])
    back = hip.get_inverse_transformed_latents(lat, as_numpy=True)
    assert back.shape == lat.shape


def test_mspsvae_fit_and_export_on_multi_session_batches(tmp_path):
    """`fit_model` (the reference's ae_grid_search.main minus test-tube) for 'msps-vae' on a
    three-session generator serving two sessions per training batch: metric rows carry the
    triplet term, the best model is saved, latents [z_s | z_b | z_u] are exported per session."""
    from behavenet_amd.fitting.ae_grid_search import fit_model
    dim = [1, 32, 32]
    arch = load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, 'msps-vae', {
        'n_background': 2, 'n_sessions_per_batch': 2, 'ps_vae.alpha': 10, 'ps_vae.beta': 5,
        'ps_vae.delta': 50, 'ps_vae.anneal_epochs': 1, 'ps_vae.ms_loss': 'triplet'})
    hp.update({'expt_dir': str(tmp_path), 'max_n_epochs': 1, 'min_n_epochs': 0,
               'val_check_interval': 1, 'enable_early_stop': False, 'early_stop_history': 10,
               'rng_seed_train': 0, 'rng_seed_model': 0, 'export_latents': True,
               'progress_bar': False, 'device': 'cuda', 'n_parallel_gpus': 1})
    os.makedirs(os.path.join(str(tmp_path), 'version_0'))
    sessions = [SyntheticSession(10, 12, dim, seed=i, n_labels=2, trial_splits='8;1;1;0',
                                 name=('lab', 'expt', 'animal', 'sess-%d' % i)) for i in range(3)]
    gen = SyntheticSessionsGenerator(sessions, device=DEV, placement='device_u8',
                                     n_sessions_per_batch=2)

    class Exp(object):
        version = 0
        rows = []

        def log(self, row):
            self.rows.append(dict(row))

        def save(self):
            pass
    exp = Exp()
    model = fit_model(hp, gen, exp)
    assert type(model) is MSPSVAE and hp['n_labels'] == 2 and hp['training_completed']
    train_rows = [r for r in exp.rows if r.get('dataset') == -1 and 'tr_loss' in r]
    assert len(train_rows) == 2 and all(np.isfinite(r['tr_loss']) for r in train_rows)
    assert all(r['tr_loss_triplet'] > 0 for r in train_rows)
    val_rows = [r for r in exp.rows if r.get('dataset') == -1 and 'val_loss' in r]
    assert val_rows and all(r['val_loss_triplet'] == 0 for r in val_rows)
    assert os.path.exists(os.path.join(str(tmp_path), 'version_0', 'best_val_model.pt'))
    for i in range(3):
        pkl = os.path.join(str(tmp_path), 'version_0', 'lab_expt_animal_sess-%d_latents.pkl' % i)
        with open(pkl, 'rb') as f:
            lat = pickle.load(f)
        assert len(lat['latents']) == 10
        assert all(a.shape == (12, 8) and np.all(np.isfinite(a)) for a in lat['latents'])


def test_conv_decoder_vs_oracle_and_golden():
    """ConvDecoder (labels -> images, ref decoders.py:355-496), 210 frames = two chunks: forward,
    loss, accumulated gradients and the Adam(amsgrad) trajectory against the oracle and the
    vectors recorded from the reference."""
    z, meta = load_case('convdecoder_cfg1')
    hip, ora, hp = _pair(meta)
    data_c = case_data(meta)
    data_g = {k: v.to(DEV) for k, v in data_c.items()}
    n_fwd = meta['n_fwd']
    hip.train()
    ora.train()
    with torch.no_grad():
        out_h = hip(data_g['labels'][0][:n_fwd], dataset=0)
        out_o = ora(data_c['labels'][0][:n_fwd], dataset=0)
    assert isinstance(out_h, torch.Tensor)
    close(out_h, out_o, name='convdecoder fwd')
    close(out_h, torch.from_numpy(z['fwd/x_hat']), name='convdecoder fwd golden')
    hip.zero_grad()
    ora.zero_grad()
    loss_h = hip.loss(data_g, dataset=0, accumulate_grad=True)
    loss_o = ora.loss(data_c, dataset=0, accumulate_grad=True)
    assert sorted(loss_h.keys()) == ['loss']
    assert loss_h['loss'] == pytest.approx(loss_o['loss'], rel=1e-4)
    assert loss_h['loss'] == pytest.approx(float(z['loss/vals'][0]), rel=1e-4)
    # gradients: on the branch pattern the device took (see tests/branches.py); the golden
    # checksums of the reference's gradients are matched by the plain oracle on the CPU
    # (tests/test_oracle_golden.py)
    ora64 = seeded_build(ref_cpu.build_model, case_hparams(meta)).double()
    ora64.train()
    hip.zero_grad()
    with record_branches(hip) as rec:
        hip.loss(data_g, dataset=0, accumulate_grad=True)
    with BranchReplay(rec) as br:
        ora64.loss({k: v.double() for k, v in data_c.items()}, dataset=0, accumulate_grad=True)
    br.assert_only_ties()
    grads_close_on_same_branches(hip, ora64, 'convdecoder')
    for (k, ph), (_, po) in zip(hip.named_parameters(), ora.named_parameters()):
        close(ph.grad, po.grad, name='convdecoder grad ' + k, norm_tol=2e-3)
    # no gradient side effects without accumulate_grad
    hip.zero_grad()
    val = hip.loss(data_g, dataset=0, accumulate_grad=False)
    assert val['loss'] == pytest.approx(loss_h['loss'], rel=1e-6)
    assert all(p.grad is None or float(p.grad.abs().max()) == 0 for p in hip.parameters())
    opt = FlatAdamAMSGrad(hip.get_parameters(), lr=hp['learning_rate'],
                          weight_decay=hp.get('l2_reg', 0))
    traj = []
    for _ in range(3):
        opt.zero_grad()
        traj.append(hip.loss(data_g, dataset=0, accumulate_grad=True)['loss'])
        opt.step()
    np.testing.assert_allclose(traj, z['adam/losses'], rtol=1e-4)


def test_overlapped_bucketed_all_reduce_single_rank_rccl():
    """The data-parallel gradient exchange on the real RCCL path (a one-rank process group): every
    bucket but the last goes out during the backward pass and the trained parameters are
    bit-identical to the run without it.  (World size 2 is covered on gloo, CPU suite.)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dist_single_rank_check.py')
    env = dict(os.environ, BN_DIST_FORCE='1', BN_BUCKET_MB='4')
    res = subprocess.run([sys.executable, script, '210'], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['identical'] and out['losses_identical'], out
    assert out['n_buckets'] >= 3
    # single-pass schedule: all buckets are launched from inside the backward pass
    assert all(n == out['n_buckets'] for n in out['overlapped_per_step']), out


def test_host_u8_prefetch_feed_is_bit_identical():
    """The pinned-uint8 feed with one-trial look-ahead hands out exactly the frames the resident
    float32 feed does (two epochs over two sessions of ragged trials, same RNG consumption)."""
    def frames(placement):
        sessions = [SyntheticSession(10, [5, 7, 3, 9, 4, 6, 8, 5, 7, 4], [1, 32, 32], seed=3 + i,
                                     trial_splits='8;1;1;0') for i in range(2)]
        gen = SyntheticSessionsGenerator(sessions, device=DEV, placement=placement)
        out = []
        for epoch in range(2):
            torch.manual_seed(epoch)
            np.random.seed(epoch)
            gen.reset_iterators('train')
            while True:
                data, sess = gen.next_batch('train')
                if data is None:
                    break
                out.append((sess, int(data['batch_idx'][0]), data['images'][0].clone()))
        torch.cuda.synchronize()
        return out
    a, b = frames('device'), frames('host_u8')
    assert len(a) == len(b) == 32
    for (s1, t1, x1), (s2, t2, x2) in zip(a, b):
        assert (s1, t1) == (s2, t2)
        assert torch.equal(x1, x2)


def test_host_u8_feed_with_ragged_trials_runs_ahead_safely():
    """Pinned-uint8 feed with look-ahead copies on a second stream, trials of different lengths,
    and NO host synchronisation between steps (the host runs ahead of the device as in ``fit``):
    the staging buffers must never alias memory the compute stream still uses.  (Regression: they
    were allocated on the compute stream, and the first look-ahead copy into a fresh buffer
    overwrote live activations of the previous step with raw frame bytes -> NaN gradients.)"""
    dim = [1, 32, 32]
    arch = load_handcrafted_arch(list(dim), 4, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    sess = SyntheticSession(10, [5 + (t % 3) for t in range(10)], dim, seed=40,
                            trial_splits='8;1;1;0')
    gen = SyntheticSessionsGenerator([sess], device=DEV, placement='host_u8')
    refs = [torch.from_numpy(u.astype(np.float32) / 255).to(DEV) for u in sess.images_u8]
    torch.manual_seed(0)
    model = AE(hp).to(DEV)
    opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4)
    flags = []
    for epoch in range(3):
        torch.manual_seed(epoch)
        np.random.seed(epoch)
        gen.reset_iterators('train')
        for _ in range(gen.n_tot_batches['train']):
            model.train()
            opt.zero_grad()
            data, ds = gen.next_batch('train')
            x_ok = (data['images'][0] == refs[int(data['batch_idx'][0])]).all()
            model.loss(data, dataset=ds, accumulate_grad=True)
            flags.append((x_ok, torch.isfinite(opt.flat_g).all()))
            if epoch > 0:
                opt.step()
        gen.reset_iterators('val')
        data, ds = gen.next_batch('val')
        model.loss(data, dataset=ds, accumulate_grad=False)
    torch.cuda.synchronize()
    for i, (x_ok, g_ok) in enumerate(flags):
        assert bool(x_ok), 'step %d: wrong frames served' % i
        assert bool(g_ok), 'step %d: non-finite gradients' % i
    assert bool(torch.isfinite(opt.flat_p).all())


def test_file_backed_uint8_feed_on_device(tmp_path):
    """ConcatSessionsGenerator over data.npz session files, images uint8 from disk -> pinned host
    -> device one trial ahead -> bn_u8_to_unit_float: bit-identical batches to the resident
    float32 feed over two epochs, and a short fit() through it."""
    from behavenet_amd.data.data_generator import ConcatSessionsGenerator
    from tests.test_fit_host import _write_sessions, _epoch
    root = str(tmp_path)
    ids, sessions = _write_sessions(root, n_sessions=2, n_trials=10, dim=(1, 32, 32), n_labels=0)
    gen_f = ConcatSessionsGenerator(root, ids, device=DEV)            # placement='host_u8'
    gen_m = SyntheticSessionsGenerator(sessions, device=DEV, placement='device')
    for seed in (0, 1):
        a, b = _epoch(gen_f, 'train', seed), _epoch(gen_m, 'train', seed)
        torch.cuda.synchronize()
        assert len(a) == len(b) == 16
        for (s1, t1, d1), (s2, t2, d2) in zip(a, b):
            assert (s1, t1) == (s2, t2)
            assert d1['images'].is_cuda and torch.equal(d1['images'], d2['images'])
    arch = load_handcrafted_arch([1, 32, 32], 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    hp.update({'expt_dir': root, 'max_n_epochs': 1, 'min_n_epochs': 0, 'val_check_interval': 1,
               'enable_early_stop': False, 'early_stop_history': 10, 'rng_seed_train': 0,
               'export_latents': False, 'progress_bar': False, 'device': 'cuda'})
    os.makedirs(os.path.join(root, 'version_0'), exist_ok=True)

    def run(gen):
        torch.manual_seed(0)
        model = AE(dict(hp)).to(DEV)
        model.version = 0

        class Exp(object):
            version = 0

            def __init__(self):
                self.rows = []

            def log(self, row):
                self.rows.append(dict(row))

            def save(self):
                pass
        exp = Exp()
        fit(dict(hp), model, gen, exp, method='ae')
        return [r for r in exp.rows if 'tr_loss' in r or 'val_loss' in r]
    rows_f, rows_m = run(gen_f), run(gen_m)
    assert len(rows_f) == len(rows_m) > 0
    for rf, rm in zip(rows_f, rows_m):
        assert rf == rm


def test_losses_known_answers_on_device():
    """Closed-form answers of the reference's tests/test_fitting/test_losses.py:8-94."""
    LN2PI = np.log(2 * np.pi)
    x = torch.rand((5, 3), device=DEV)
    assert losses.mse(x, x).item() == 0
    a = torch.tensor([1, 2, 3, 4, 5, 6], dtype=torch.float, device=DEV)
    b = torch.tensor([2, 3, 4, 5, 6, 7], dtype=torch.float, device=DEV)
    m = torch.tensor([1, 0, 1, 0, 1, 0], dtype=torch.float, device=DEV)
    assert losses.mse(a, b, m).item() == 0.5
    n_batch, n_dims = 5, 3
    assert losses.gaussian_ll(x, x).item() == pytest.approx(-(0.5 * LN2PI) * n_dims, rel=1e-6)
    ones = torch.ones(n_batch, n_dims, device=DEV)
    zeros = torch.zeros(n_batch, n_dims, device=DEV)
    mask = torch.zeros(n_batch, n_dims, device=DEV)
    mask[:, 0] = 1
    assert losses.gaussian_ll(ones, zeros, masks=mask).item() == pytest.approx(
        -(0.5 * LN2PI) * n_dims - 0.5, rel=1e-6)
    ll = losses.gaussian_ll(ones, zeros)
    mse_ = 2 * (-ll.item() - 0.5 * LN2PI * n_dims) / n_dims
    assert np.allclose(losses.gaussian_ll_to_mse(ll.cpu().numpy(), n_dims), mse_)
    assert losses.kl_div_to_std_normal(torch.zeros(1, 1, device=DEV),
                                       torch.zeros(1, 1, device=DEV)).item() == 0
    zz, mu, lv = (torch.rand(5, 3, device=DEV) for _ in range(3))
    mi, tc, dw = losses.decomposed_kl(zz, mu, lv)
    assert losses.index_code_mi(zz, mu, lv).item() == mi.item()
    assert losses.total_correlation(zz, mu, lv).item() == tc.item()
    assert losses.dimension_wise_kl_to_std_normal(zz, mu, lv).item() == dw.item()


def test_deepcopy_after_arena_and_eval_no_grad_side_effects():
    z, meta = load_case('ae_cfg1')
    hip, ora, hp = _pair(meta)
    data = {k: v.to(DEV) for k, v in case_data(meta).items()}
    opt = FlatAdamAMSGrad(hip.get_parameters(), lr=1e-4)
    opt.zero_grad()
    hip.loss(data, accumulate_grad=True)
    g = [p.grad.clone() for p in hip.parameters()]
    hip.eval()
    hip.loss(data, accumulate_grad=False)   # validation pass: gradients untouched
    for a, p in zip(g, hip.parameters()):
        assert torch.equal(a, p.grad)
    hip.hparams = None
    snap = copy.deepcopy(hip)
    hip.hparams = snap.hparams = hp
    opt.step()
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(
        snap.state_dict().values(), hip.state_dict().values()))
    assert changed == len(hip.state_dict())   # the snapshot does not alias the live arena


def test_export_latents_from_a_trial_store_graph_replay_equals_eager(tmp_path):
    """BASELINE configs[4] through the product entry point over a FILE-BACKED session (data.npz trial store,
    pinned uint8 feed with reader threads, uint8 frames handed to the first conv layer as they are): the
    exporter's recorded trial encoder (fitting/eval.py, _GraphedTrialEncoder; trials of two lengths -> two
    graphs) writes bit for bit the latents of eager launches, and both match the oracle's encoder."""
    from behavenet_amd.data.data_generator import ConcatSessionsGenerator
    from behavenet_amd.data.trial_store import write_npz_session
    from behavenet_amd.fitting.eval import export_latents
    dim = [1, 64, 48]
    arch = load_handcrafted_arch(list(dim), 6, None, check_memory=False)
    hp = base_hparams(arch, 'ae', {'expt_dir': str(tmp_path), 'device': 'cuda'})
    torch.manual_seed(0)
    hip = AE(hp).to(DEV)
    hip.version = 0
    torch.manual_seed(0)
    ora = ref_cpu.AE(base_hparams(dict(arch), 'ae'))
    rng = np.random.default_rng(3)
    lens = [30, 30, 17, 30, 17, 30, 30, 17, 30, 30, 30, 17]
    trials = [rng.integers(0, 255, size=(t,) + tuple(dim), dtype=np.uint8) for t in lens]
    sess_dir = os.path.join(str(tmp_path), 'lab', 'expt', 'animal', 'sess')
    write_npz_session(os.path.join(sess_dir, 'data.npz'), {'images': trials})
    ids = {'lab': 'lab', 'expt': 'expt', 'animal': 'animal', 'session': 'sess'}

    def run(graph, name):
        gen = ConcatSessionsGenerator(str(tmp_path), [ids], signals_list=[['images']], transforms_list=[[None]],
                                      paths_list=[[os.path.join(sess_dir, 'data.npz')]], device='cuda',
                                      placement='host_u8', keep_in_memory=False)
        hip.hparams['hip_graph_encode'] = graph
        out = os.path.join(str(tmp_path), name)
        export_latents(gen, hip, filename=out)
        with open(out, 'rb') as f:
            return pickle.load(f), sorted(int(t) for k in ('train', 'val', 'test') for t in gen.datasets[0].batch_idxs[k])
    (a, used), (b, _) = run(True, 'graph.pkl'), run(False, 'eager.pkl')
    assert len(a['latents']) == len(lens) and len(used) == 10     # (12 trials in blocks of 8 + 1 + 1: two are in no split)
    for i, t in enumerate(lens):
        if i in used:
            assert a['latents'][i].shape == (t, 6)
            assert np.array_equal(a['latents'][i], b['latents'][i]), i
        else:
            assert a['latents'][i].size == 0 and b['latents'][i].size == 0
    i = used[2]
    with torch.no_grad():
        want = ora.encoding(torch.from_numpy(trials[i].astype(np.float32) / 255), dataset=0)[0].numpy()
    assert np.abs(a['latents'][i] - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('model_class', ['ae', 'vae', 'beta-tcvae', 'cond-ae', 'cond-vae', 'cond-ae-msp', 'ps-vae'])
def test_get_reconstruction_from_images_and_from_latents(model_class):
    """``fitting.eval.get_reconstruction`` (ref eval.py:286-374) for every class of the path: from images -- the
    reconstruction and the right element of ``forward`` as latents, equal to the CPU oracle's forward on the same
    parameters -- and from latents through the decoder (labels appended for the conditional classes, latents mapped
    back from the transformed space for AEMSP / PS-VAE); numpy out, no gradient state left behind."""
    from behavenet_amd.fitting.eval import get_reconstruction
    extra = {'msp.alpha': 0.05, 'conditional_encoder': False, 'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0,
             'beta_tcvae.beta': 1.0, 'beta_tcvae.beta_anneal_epochs': 0, 'ps_vae.alpha': 1, 'ps_vae.beta': 1,
             'ps_vae.anneal_epochs': 0, 'max_n_epochs': 4}
    n_labels = 4 if model_class in ('cond-ae', 'cond-vae', 'cond-ae-msp', 'ps-vae') else 0
    meta = {'dim': [1, 32, 32], 'n_lat': 8, 'model_class': model_class, 'extra_hp': extra, 'n_labels': n_labels,
            'n_frames': 6}
    hip, ora, hp = _pair(meta)
    if model_class == 'cond-ae-msp':
        hip.create_orthogonal_matrix()          # U = [P; null space of P]: what save() does before it writes
    data_c = case_data(meta)
    x = data_c['images'][0]
    labels = data_c['labels'][0] if n_labels else None
    kw = {'labels': labels.to(DEV)} if model_class in ('cond-ae', 'cond-vae') else {}
    recon, latents = get_reconstruction(hip, x.to(DEV), dataset=0, return_latents=True, **kw)
    assert isinstance(recon, np.ndarray) and recon.shape == tuple(x.shape) and latents.shape == (6, 8)
    assert not hip.training
    ora.eval()
    with torch.no_grad():
        okw = {'labels': labels} if model_class in ('cond-ae', 'cond-vae') else {}
        if model_class in ('vae', 'beta-tcvae', 'ps-vae'):
            okw['use_mean'] = True
        out = ora(x, dataset=0, **okw)
    slot = 2 if model_class == 'ps-vae' else 1
    if model_class != 'cond-vae':               # (the reference calls cond-VAE without use_mean: it samples)
        assert np.abs(recon - out[0].numpy()).max() <= 2e-6
        assert np.abs(latents - out[slot].numpy()).max() <= 2e-5 * max(1.0, np.abs(out[slot].numpy()).max())
    # plain numpy in, reconstruction only
    again = get_reconstruction(hip, x.numpy(), dataset=0, **kw)
    assert isinstance(again, np.ndarray) and again.shape == recon.shape
    if model_class != 'cond-vae':
        assert np.array_equal(again, recon)
    # from latents
    z = torch.from_numpy(latents).to(DEV)
    lkw = {'labels': labels.to(DEV)} if model_class in ('cond-ae', 'cond-vae') else {}
    rec_z, z_used = get_reconstruction(hip, z, return_latents=True, **lkw)
    with torch.no_grad():
        if model_class in ('cond-ae', 'cond-vae'):
            z_in = torch.cat((z, labels.to(DEV)), dim=1)
        elif model_class in ('cond-ae-msp', 'ps-vae'):
            z_in = hip.get_inverse_transformed_latents(z, as_numpy=False)
        else:
            z_in = z
        want = hip.decoding(z_in, None, None, dataset=None)
    assert np.array_equal(rec_z, want.cpu().numpy()) and np.array_equal(z_used, z_in.cpu().numpy())
    if model_class == 'ae':
        # the decoder applied to the encoder's latents is the forward pass
        assert np.abs(rec_z - recon).max() <= 1e-6
    for p_ in hip.parameters():
        assert p_.grad is None or float(p_.grad.abs().max()) == 0.0
    with pytest.raises(ValueError):
        hip.hparams['model_class'] = 'nope'
        get_reconstruction(hip, x.to(DEV))
    hip.hparams['model_class'] = model_class
