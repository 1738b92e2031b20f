"""Generate golden vectors by importing the REAL reference (themattinthehatt/behavenet).

Run in the build container only (the reference is mounted read-only at /root/reference and
never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs small fixtures next to this file:
  planner.json          layer-planner outputs for several input shapes / arch jsons
  ae_<case>.npz         AE forward/backward/Adam(amsgrad) trajectories
  vae_cfg1.npz, psvae_cfg4.npz, condvae_cfg1.npz, betatc_cfg1.npz   variational variants
  fit_cfg1.json         metric rows of the reference fit() driven by the repo's synthetic
                        generator stub (behavenet_amd.data.data_generator)

Large tensors (weights: 8.7 M floats) are NOT stored; they are pinned through float64
checksums and strided samples -- the same torch version + seed regenerates them bit-exactly
(checked by tests/test_oracle_golden.py).  Inputs are regenerated from numpy seeds.
"""

import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REF)
sys.path.insert(0, REPO)   # repo first: both trees have a `tests` package
sys.dont_write_bytecode = True

# --- commentjson shim (absent in this image; the reference only needs .load) ---------------
_shim = types.ModuleType('commentjson')


def _cj_load(f):
    txt = '\n'.join(line.split('#')[0] for line in f.read().splitlines())
    return json.loads(txt)


_shim.load = _cj_load
sys.modules['commentjson'] = _shim

from behavenet.models import ae_model_architecture_generator as ref_arch  # noqa: E402
from behavenet.models.aes import AE as RefAE  # noqa: E402
from behavenet.models.vaes import (  # noqa: E402
    VAE as RefVAE, PSVAE as RefPSVAE, ConditionalVAE as RefCondVAE, BetaTCVAE as RefBetaTC)
from behavenet.fitting.training import fit as ref_fit  # noqa: E402

from tests.golden_utils import (  # noqa: E402
    checksum, strided_sample, make_frames, make_labels, base_hparams)

torch.set_num_threads(8)


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    return o


def planner_fixture():
    out = {}
    cases = {
        'default_1x32x32': ([1, 32, 32], 8, None),
        'default_1x128x128': ([1, 128, 128], 12, None),
        'default_2x128x128': ([2, 128, 128], 16, None),
        'default_1x64x48': ([1, 64, 48], 8, None),
        'arch2_2x128x128': ([2, 128, 128], 12, 'ae_arch_2.json'),
        'archdefault_1x128x128': ([1, 128, 128], 12, 'ae_arch_default.json'),
    }
    for name, (dim, n_lat, js) in cases.items():
        path = None if js is None else os.path.join(REF, 'configs', 'ae_jsons', js)
        arch = ref_arch.load_handcrafted_arch(list(dim), n_lat, path, check_memory=False)
        out[name] = {'input_dim': dim, 'n_ae_latents': n_lat, 'arch_json': js,
                     'arch': jsonable(arch)}
    # calculate_output_dim grid
    grid = []
    for layer_type in ['conv', 'maxpool']:
        for pad in ['same', 'valid']:
            for inp in [15, 16, 17, 31, 32, 64, 128]:
                for k in ([2] if layer_type == 'maxpool' else [3, 4, 5, 7]):
                    for s in [1, 2, 3, 5]:
                        if pad == 'valid' and inp < k:
                            continue
                        r = ref_arch.calculate_output_dim(inp, k, s, pad, layer_type)
                        grid.append([inp, k, s, pad, layer_type] + [int(v) for v in r])
    out['calculate_output_dim'] = grid
    with open(os.path.join(HERE, 'planner.json'), 'w') as f:
        json.dump(out, f)
    print('planner.json written')


def tensor_record(store, prefix, t, full_limit=4096):
    a = t.detach().cpu().numpy()
    store[prefix + '/checksum'] = checksum(a)
    if a.size <= full_limit:
        store[prefix + '/full'] = a.astype(np.float32)
    else:
        store[prefix + '/sample'] = strided_sample(a)


def model_case(name, RefModel, dim, n_lat, n_frames, model_class, extra_hp=None, n_labels=0,
               chunked=False, store_xhat=True, dataset=0, masks=False, arch_json=None):
    arch = ref_arch.load_handcrafted_arch(
        list(dim), n_lat, os.path.join(HERE, arch_json) if arch_json else None,
        check_memory=False)
    hp = base_hparams(arch, model_class, extra_hp)
    if n_labels:
        hp['n_labels'] = n_labels
    np.random.seed(0)            # PS-VAE draws its orthogonal A/B from numpy's global RNG
    torch.manual_seed(0)
    model = RefModel(hp)
    model.train()
    store = {}
    for k, v in model.state_dict().items():
        tensor_record(store, 'param0/' + k, v, full_limit=1024)

    x = torch.from_numpy(make_frames(n_frames, dim, seed=1))
    data = {'images': x[None]}
    if n_labels:
        y = torch.from_numpy(make_labels(n_frames, n_labels, seed=2))
        data['labels'] = y[None]
    if masks:
        from tests.golden_utils import make_masks
        data['masks'] = torch.from_numpy(make_masks(n_frames, dim, seed=4))[None]
    if model_class == 'cond-ae' and (extra_hp or {}).get('conditional_encoder'):
        from tests.golden_utils import make_labels_sc
        data['labels_sc'] = torch.from_numpy(
            make_labels_sc(n_frames, n_labels // 2, dim, seed=3))[None]

    # eps: record what torch's CPU generator hands to reparameterize so that the oracle and the
    # device run can be fed the same noise
    variational = model_class in ('vae', 'ps-vae', 'cond-vae', 'beta-tcvae')
    eps_log = []
    if variational:
        import behavenet.models.vaes as ref_vaes
        orig = ref_vaes.reparameterize

        def recording(mu, logvar):
            std = torch.exp(logvar)
            eps = torch.randn_like(std)
            eps_log.append(eps.clone())
            return eps.mul(std).add_(mu)
        ref_vaes.reparameterize = recording

    # forward taps
    acts = []
    hooks = []
    for mod_name, mod in model.named_modules():
        if isinstance(mod, (torch.nn.LeakyReLU, torch.nn.Sigmoid)):
            hooks.append(mod.register_forward_hook(
                lambda m, i, o, nm=mod_name: acts.append((nm, o.detach().clone()))))
    torch.manual_seed(123)
    with torch.no_grad():
        kw = {}
        if model_class in ('cond-vae', 'cond-ae'):
            kw = {'labels': data['labels'][0], 'labels_2d': None}
        n_fwd = min(n_frames, 8)
        if model_class in ('cond-vae', 'cond-ae'):
            kw['labels'] = kw['labels'][:n_fwd]
            if 'labels_sc' in data:
                kw['labels_2d'] = data['labels_sc'][0][:n_fwd]
        out = model(x[:n_fwd], dataset=dataset, **kw)
    for h in hooks:
        h.remove()
    for nm, a in acts:
        store['act/' + nm + '/checksum'] = checksum(a.numpy())
    x_hat = out[0]
    if store_xhat:
        store['fwd/x_hat'] = x_hat.numpy().astype(np.float32)
    else:
        store['fwd/x_hat_first'] = x_hat[:1].numpy().astype(np.float32)
    store['fwd/x_hat/checksum'] = checksum(x_hat.numpy())
    names = {2: ['z'], 3: ['z', 'y_hat'], 4: ['z', 'mu', 'logvar'],
             5: ['z', 'mu', 'logvar', 'y_hat']}[len(out)]
    for nm, t in zip(names, out[1:]):
        store['fwd/' + nm] = t.numpy().astype(np.float32)
    if variational:
        store['fwd/eps'] = eps_log[0].numpy()
        eps_log.clear()

    # loss + grads (fresh eps stream for the loss call)
    torch.manual_seed(124)
    model.zero_grad()
    model.curr_epoch = 3 if variational else 0
    loss_dict = model.loss(data, dataset=dataset, accumulate_grad=True)
    store['loss/keys'] = np.array(sorted(loss_dict.keys()))
    store['loss/vals'] = np.array([float(loss_dict[k]) for k in sorted(loss_dict.keys())],
                                  dtype=np.float64)
    if variational:
        for i, e in enumerate(eps_log):
            store['loss/eps%d' % i] = e.numpy()
        eps_log.clear()
    for k, p in model.named_parameters():
        if p.grad is not None:
            tensor_record(store, 'grad/' + k, p.grad, full_limit=4096)

    # 3 Adam(amsgrad) steps, as training.py:284-286,336-352 (epoch > 0)
    opt = torch.optim.Adam(model.get_parameters(), lr=hp['learning_rate'],
                           weight_decay=hp.get('l2_reg', 0), amsgrad=True)
    traj = []
    for step in range(3):
        torch.manual_seed(200 + step)
        opt.zero_grad()
        ld = model.loss(data, dataset=dataset, accumulate_grad=True)
        opt.step()
        traj.append(float(ld['loss']))
        if variational:
            for i, e in enumerate(eps_log):
                store['adam/eps_step%d_%d' % (step, i)] = e.numpy()
            eps_log.clear()
    store['adam/losses'] = np.array(traj, dtype=np.float64)
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        tensor_record(store, 'adam/param/' + k, p, full_limit=1024)
        st = opt.state[p]
        if not st:            # a parameter that never receives a gradient (AEMSP.U)
            continue
        for sk in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
            store['adam/%s/%s/checksum' % (sk, k)] = checksum(st[sk].numpy())

    # batch-norm running statistics after: 1 no-grad train-mode forward, the loss call and the
    # three steps (one update per 200-frame chunk each time)
    for k, v in model.named_buffers():
        if 'running_' in k or 'num_batches_tracked' in k:
            store['adam/buffer/' + k] = v.detach().numpy().astype(np.float64)

    if variational:
        ref_vaes.reparameterize = orig

    meta = {'dim': list(dim), 'n_lat': n_lat, 'n_frames': n_frames, 'model_class': model_class,
            'n_labels': n_labels, 'extra_hp': extra_hp or {}, 'n_fwd': n_fwd,
            'curr_epoch': model.curr_epoch if variational else 0, 'dataset': dataset,
            'masks': bool(masks), 'arch_json': arch_json}
    store['meta'] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **store)
    print('%s written (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def decoder_case(name, dim, n_labels, n_frames, n_fwd=8):
    """ConvDecoder (labels -> images, ref decoders.py:355-496): parameters from the seed, forward,
    loss, gradients and a 3-step Adam(amsgrad) trajectory."""
    from behavenet.models.decoders import ConvDecoder as RefConvDecoder
    arch = ref_arch.load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, 'conv-decoder', None)
    hp['n_labels'] = n_labels
    np.random.seed(0)
    torch.manual_seed(0)
    model = RefConvDecoder(hp)
    model.train()
    store = {}
    for k, v in model.state_dict().items():
        tensor_record(store, 'param0/' + k, v, full_limit=1024)
    x = torch.from_numpy(make_frames(n_frames, dim, seed=1))
    y = torch.from_numpy(make_labels(n_frames, n_labels, seed=2))
    data = {'images': x[None], 'labels': y[None]}
    with torch.no_grad():
        out = model(y[:n_fwd], dataset=0)
    store['fwd/x_hat'] = out.numpy().astype(np.float32)
    store['fwd/x_hat/checksum'] = checksum(out.numpy())
    model.zero_grad()
    loss_dict = model.loss(data, dataset=0, accumulate_grad=True)
    store['loss/keys'] = np.array(sorted(loss_dict.keys()))
    store['loss/vals'] = np.array([float(loss_dict[k]) for k in sorted(loss_dict.keys())],
                                  dtype=np.float64)
    for k, p in model.named_parameters():
        tensor_record(store, 'grad/' + k, p.grad, full_limit=4096)
    opt = torch.optim.Adam(model.get_parameters(), lr=hp['learning_rate'],
                           weight_decay=hp.get('l2_reg', 0), amsgrad=True)
    traj = []
    for step in range(3):
        opt.zero_grad()
        traj.append(float(model.loss(data, dataset=0, accumulate_grad=True)['loss']))
        opt.step()
    store['adam/losses'] = np.array(traj, dtype=np.float64)
    for k, p in model.named_parameters():
        tensor_record(store, 'adam/param/' + k, p, full_limit=1024)
    meta = {'dim': list(dim), 'n_lat': 8, 'n_frames': n_frames, 'model_class': 'conv-decoder',
            'n_labels': n_labels, 'extra_hp': {}, 'n_fwd': n_fwd, 'curr_epoch': 0}
    store['meta'] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **store)
    print('%s written (%.1f KB)' % (path, os.path.getsize(path) / 1024))


def msps_case(name='mspsvae_cfg1'):
    """Multi-session PS-VAE (ref vaes.py:849-1098) on a two-session batch, plus known answers of
    losses.triplet_loss for 2, 3 and 4 sessions."""
    import behavenet.models.vaes as ref_vaes
    from behavenet.fitting import losses as ref_losses
    dim, n_lat, n_labels = [1, 32, 32], 8, 2
    extra = {'n_background': 2, 'n_sessions_per_batch': 2, 'ps_vae.alpha': 10, 'ps_vae.beta': 5,
             'ps_vae.delta': 50, 'ps_vae.anneal_epochs': 5, 'max_n_epochs': 10,
             'ps_vae.ms_loss': 'triplet'}
    arch = ref_arch.load_handcrafted_arch(list(dim), n_lat, None, check_memory=False)
    hp = base_hparams(arch, 'msps-vae', extra)
    hp['n_labels'] = n_labels
    np.random.seed(0)
    torch.manual_seed(0)
    model = ref_vaes.MSPSVAE(hp)
    model.train()
    store = {}
    for k, v in model.state_dict().items():
        tensor_record(store, 'param0/' + k, v, full_limit=1024)
    frames = [18, 15]
    datas = []
    for i, t in enumerate(frames):
        datas.append({'images': torch.from_numpy(make_frames(t, dim, seed=1 + i))[None],
                      'labels': torch.from_numpy(make_labels(t, n_labels, seed=5 + i))[None]})
    sess = [1, 0]            # dataset ids of the two batches, as the generator hands them over

    eps_log = []
    orig = ref_vaes.reparameterize

    def recording(mu, logvar):
        std = torch.exp(logvar)
        eps = torch.randn_like(std)
        eps_log.append(eps.clone())
        return eps.mul(std).add_(mu)
    ref_vaes.reparameterize = recording

    torch.manual_seed(123)
    with torch.no_grad():
        out = model(datas[0]['images'][0][:8], dataset=None)
    for nm, t in zip(['x_hat', 'z', 'mu', 'logvar', 'y_hat'], out):
        store['fwd/' + nm] = t.numpy().astype(np.float32)
    store['fwd/eps'] = eps_log[0].numpy()
    eps_log.clear()

    model.curr_epoch = 3
    torch.manual_seed(124)
    np.random.seed(11)
    model.zero_grad()
    ld = model.loss(datas, dataset=sess, accumulate_grad=True)
    store['loss/keys'] = np.array(sorted(ld.keys()))
    store['loss/vals'] = np.array([float(ld[k]) for k in sorted(ld.keys())], dtype=np.float64)
    store['loss/eps0'] = eps_log[0].numpy()
    eps_log.clear()
    for k, p in model.named_parameters():
        if p.grad is not None:
            tensor_record(store, 'grad/' + k, p.grad, full_limit=4096)
    # validation-style call: one session, no triplet term
    torch.manual_seed(125)
    ld1 = model.loss(datas[0], dataset=1, accumulate_grad=False)
    store['loss1/keys'] = np.array(sorted(ld1.keys()))
    store['loss1/vals'] = np.array([float(ld1[k]) for k in sorted(ld1.keys())], dtype=np.float64)
    store['loss1/eps0'] = eps_log[0].numpy()
    eps_log.clear()

    opt = torch.optim.Adam(model.get_parameters(), lr=hp['learning_rate'],
                           weight_decay=hp.get('l2_reg', 0), amsgrad=True)
    traj = []
    for step in range(3):
        torch.manual_seed(200 + step)
        np.random.seed(20 + step)
        opt.zero_grad()
        traj.append(float(model.loss(datas, dataset=sess, accumulate_grad=True)['loss']))
        opt.step()
        store['adam/eps_step%d_0' % step] = eps_log[0].numpy()
        eps_log.clear()
    store['adam/losses'] = np.array(traj, dtype=np.float64)
    for k, p in model.named_parameters():
        if p.requires_grad:
            tensor_record(store, 'adam/param/' + k, p, full_limit=1024)
    ref_vaes.reparameterize = orig

    # losses.triplet_loss known answers
    obj = torch.nn.TripletMarginLoss(margin=1.0, p=2)
    g = torch.Generator().manual_seed(5)
    for n_sess in (2, 3, 4):
        z = torch.randn((40 * n_sess, 3), generator=g)
        ids = np.repeat(np.arange(n_sess), 40)[np.random.RandomState(n_sess).permutation(40 * n_sess)]
        np.random.seed(30 + n_sess)
        store['triplet/%d/z' % n_sess] = z.numpy()
        store['triplet/%d/ids' % n_sess] = ids.astype(np.float64)
        store['triplet/%d/val' % n_sess] = np.float64(ref_losses.triplet_loss(obj, z, ids).item())

    meta = {'dim': dim, 'n_lat': n_lat, 'n_frames': frames, 'model_class': 'msps-vae',
            'n_labels': n_labels, 'extra_hp': extra, 'n_fwd': 8, 'curr_epoch': 3, 'sess': sess}
    store['meta'] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **store)
    print('%s written (%.1f KB)' % (path, os.path.getsize(path) / 1024))


class ListExp(object):
    """Stand-in for test_tube.Experiment: collects rows."""
    version = 0

    def __init__(self):
        self.rows = []

    def log(self, row):
        self.rows.append(dict(row))

    def save(self):
        pass


def fit_fixture():
    from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
    dim = [1, 32, 32]
    arch = ref_arch.load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, 'ae', None)
    with tempfile.TemporaryDirectory() as tmp:
        hp.update({'expt_dir': tmp, 'max_n_epochs': 2, 'min_n_epochs': 0,
                   'val_check_interval': 1, 'enable_early_stop': False,
                   'early_stop_history': 10, 'rng_seed_train': 0, 'export_latents': False})
        os.makedirs(os.path.join(tmp, 'version_0'))
        sess = SyntheticSession(10, 32, dim, seed=0, trial_splits='8;1;1;0')
        gen = SyntheticSessionsGenerator([sess], device='cpu', placement='host')
        torch.manual_seed(0)
        model = RefAE(hp)
        model.version = 0
        exp = ListExp()
        ref_fit(hp, model, gen, exp, method='ae')
        final = {k: checksum(v.numpy()).tolist() for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, 'fit_cfg1.json'), 'w') as f:
        json.dump({'rows': jsonable(exp.rows), 'final_param_checksums': final}, f)
    print('fit_cfg1.json written:', len(exp.rows), 'rows')


from tests.h5py_stub import install as _install_h5py_shim  # noqa: E402


def generator_fixture():
    """Batch order / trial splits of the reference's OWN ``ConcatSessionsGenerator``
    (behavenet/data/data_generator.py:432-633) over two sessions on disk, seeded per epoch the
    way ``fit`` seeds (training.py:329-330), two epochs of train + val + test draws, plus the
    tensors of a few batches.  ``behavenet_amd``'s generator is pinned to it
    (tests/test_fit_host.py)."""
    _install_h5py_shim()
    from behavenet.data.data_generator import ConcatSessionsGenerator as RefGen
    from tests.test_fit_host import _write_sessions
    tmp = tempfile.mkdtemp()
    ids, sessions = _write_sessions(tmp, n_sessions=2, n_trials=22, dim=(1, 8, 8), n_labels=2)
    signals = [['images', 'labels', 'masks']] * 2
    paths = [[os.path.join(tmp, i['lab'], i['expt'], i['animal'], i['session'], 'data.hdf5')] * 3
             for i in ids]
    out = {'n_sessions': 2, 'n_trials': 22, 'dim': [1, 8, 8], 'n_labels': 2, 'cases': []}
    for train_frac, splits in ((1.0, {'train_tr': 8, 'val_tr': 1, 'test_tr': 1, 'gap_tr': 0}),
                               (0.5, {'train_tr': 5, 'val_tr': 1, 'test_tr': 1, 'gap_tr': 1})):
        gen = RefGen(tmp, ids, signals_list=signals, transforms_list=[[None] * 3] * 2,
                     paths_list=paths, device='cpu', as_numpy=False, batch_load=True, rng_seed=0,
                     trial_splits=splits, train_frac=train_frac)
        case = {'train_frac': train_frac, 'trial_splits': splits,
                'n_tot_batches': {k: int(v) for k, v in gen.n_tot_batches.items()},
                'batch_idxs': [{k: [int(i) for i in v] for k, v in ds.batch_idxs.items()}
                               for ds in gen.datasets],
                'epochs': []}
        for epoch in range(2):
            torch.manual_seed(7 + epoch)
            np.random.seed(7 + epoch)
            rec = {}
            for dtype in ('train', 'val', 'test'):
                gen.reset_iterators(dtype)
                order = []
                for _ in range(gen.n_tot_batches[dtype]):
                    data, sess = gen.next_batch(dtype)
                    entry = [int(sess), int(data['batch_idx'][0])]
                    if len(order) < 2:
                        entry.append({k: [float(c) for c in checksum(v.numpy())] for k, v in data.items()
                                      if k != 'batch_idx'})
                        entry.append({k: list(v.shape) for k, v in data.items()})
                    order.append(entry)
                rec[dtype] = order
            case['epochs'].append(rec)
        out['cases'].append(case)
    with open(os.path.join(HERE, 'generator.json'), 'w') as f:
        json.dump(jsonable(out), f, indent=1)
    print('wrote generator.json')


def fitting_utils_fixture():
    """Paths and experiment bookkeeping as the REFERENCE's own behavenet.fitting.utils computes them
    (not restated known answers): the session tree of tests/test_fitting_utils.py is built with the
    reference's export_session_info_to_csv, then get_session_dir / get_expt_dir / get_model_params /
    find_session_dirs / experiment_exists (over meta_tags.pkl files the reference's test-tube
    logger would have written) are recorded, paths relative to the tree."""
    import pickle
    import behavenet.fitting.utils as ru
    from tests.test_fitting_utils import IDS, MULTI
    root = tempfile.mkdtemp()
    for ids in IDS:
        os.makedirs(os.path.join(root, ids['lab'], ids['expt'], ids['animal'], ids['session']))
    for rel_, idxs in MULTI.items():
        ru.export_session_info_to_csv(os.path.join(root, rel_), [IDS[i] for i in idxs])

    def rel(path):
        return os.path.relpath(path, root)

    def norm_ids(lst):
        return sorted('/'.join(str(d[k]) for k in ('lab', 'expt', 'animal', 'session')) for d in lst)
    out = {'session_dir': [], 'expt_dir': [], 'model_params': [], 'find_session_dirs': [],
           'experiment_exists': []}
    for lab, expt, animal, session, multi in [
            ('lab0', 'all', '', '', None), ('lab0', 'expt0', 'all', '', None),
            ('lab0', 'expt0', 'animal0', 'all', None), ('lab0', 'expt0', 'animal0', 'session-00', None),
            ('lab1', 'expt0', 'animal0', 'all', None), ('lab0', 'expt0', 'animal0', 'all', 1)]:
        hp = {'data_dir': root, 'save_dir': root, 'sessions_csv': '', 'lab': lab, 'expt': expt,
              'animal': animal, 'session': session}
        if multi is not None:
            hp['multisession'] = multi
        d, single = ru.get_session_dir(dict(hp), session_source='save')
        out['session_dir'].append({'hparams': {k: v for k, v in hp.items() if k not in ('data_dir', 'save_dir')},
                                   'dir': rel(d), 'sessions': norm_ids(single)})
    base = {'data_dir': root, 'save_dir': root, 'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal0',
            'session': 'session-00', 'experiment_name': 'grid', 'model_type': 'conv',
            'session_dir': os.path.join(root, 'lab0', 'expt0', 'animal0', 'session-00'),
            'rng_seed_data': 0, 'trial_splits': '8;1;1;0', 'train_frac': 1.0, 'rng_seed_model': 0,
            'fit_sess_io_layers': False, 'learning_rate': 1e-4, 'l2_reg': 0.0,
            'conditional_encoder': False, 'msp.alpha': 0.1, 'vae.beta': 2.0,
            'vae.beta_anneal_epochs': 5, 'beta_tcvae.beta': 3.0, 'beta_tcvae.beta_anneal_epochs': 4,
            'ps_vae.alpha': 1000, 'ps_vae.beta': 5, 'ps_vae.anneal_epochs': 100,
            'ps_vae.delta': 50, 'n_background': 3, 'n_sessions_per_batch': 2}
    for model_class, n_lat in (('ae', 8), ('vae', 10), ('beta-tcvae', 10), ('cond-vae', 8), ('cond-ae', 8),
                               ('cond-ae-msp', 8), ('ps-vae', 10), ('msps-vae', 11)):
        hp = dict(base, model_class=model_class, n_ae_latents=n_lat)
        out['expt_dir'].append({'model_class': model_class, 'n_ae_latents': n_lat,
                                'dir': rel(ru.get_expt_dir(dict(hp)))})
        out['model_params'].append({'model_class': model_class, 'n_ae_latents': n_lat,
                                    'params': jsonable(ru.get_model_params(dict(hp)))})
    dirs, ids = ru.find_session_dirs(dict(IDS[0], save_dir=root))
    out['find_session_dirs'] = {'dirs': sorted(rel(d) for d in dirs),
                                'n_single': sum(i['multisession'] is None for i in ids)}
    # experiment_exists: three finished grid points on disk, probed with matching / new hparams
    hp = dict(base, model_class='ae', n_ae_latents=8)
    expt_dir = ru.get_expt_dir(dict(hp))
    hp['expt_dir'] = expt_dir
    grid = [dict(hp, learning_rate=lr, l2_reg=l2) for lr, l2 in ((1e-4, 0.0), (1e-3, 0.0), (1e-4, 1e-5))]
    for v, g in enumerate(grid):
        vdir = os.path.join(expt_dir, 'version_%i' % v)
        os.makedirs(vdir)
        with open(os.path.join(vdir, 'meta_tags.pkl'), 'wb') as f:
            pickle.dump(dict(ru.get_model_params(dict(g)), training_completed=(v != 1)), f)
    for lr, l2 in ((1e-4, 0.0), (1e-3, 0.0), (1e-4, 1e-5), (5e-4, 0.0)):
        probe = dict(hp, learning_rate=lr, l2_reg=l2)
        exists, version = ru.experiment_exists(dict(probe), which_version=True)
        out['experiment_exists'].append({'learning_rate': lr, 'l2_reg': l2, 'exists': bool(exists),
                                         'version': None if version is None else int(version)})
    out['base_hparams'] = {k: v for k, v in base.items() if k not in ('data_dir', 'save_dir', 'session_dir')}
    with open(os.path.join(HERE, 'fitting_utils.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote fitting_utils.json')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'fitting_utils':
        fitting_utils_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'fit':
        fit_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'generator':
        generator_fixture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'bn':
        # batch-norm cases only (added later; the other fixtures are left untouched)
        model_case('ae_cfg1_bn', RefAE, [1, 32, 32], 8, 8, 'ae',
                   extra_hp={'ae_batch_norm': True})
        model_case('ae_cfg1_bn_b210', RefAE, [1, 32, 32], 8, 210, 'ae',
                   extra_hp={'ae_batch_norm': True, 'ae_batch_norm_momentum': None},
                   store_xhat=True)
        model_case('vae_1x64x48_bn', RefVAE, [1, 64, 48], 8, 6, 'vae',
                   extra_hp={'ae_batch_norm': True, 'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0,
                             'max_n_epochs': 10})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'options':
        # session-specific input/output layers (dataset 1 of 2) with pixel masks; linear AE
        model_case('ae_sessio_masks', RefAE, [1, 32, 32], 8, 210, 'ae',
                   extra_hp={'fit_sess_io_layers': True, 'n_datasets': 2}, dataset=1, masks=True)
        model_case('ae_linear', RefAE, [1, 32, 32], 8, 12, 'ae',
                   extra_hp={'model_type': 'linear'})
        model_case('ae_valid_1x30x26', RefAE, [1, 30, 26], 6, 12, 'ae', arch_json='arch_valid.json')
        model_case('ae_maxpool', RefAE, [1, 32, 32], 8, 12, 'ae', arch_json='arch_maxpool.json')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'arch2':
        # the second shipped architecture (configs/ae_jsons/ae_arch_2.json: five 64-channel layers of
        # 4x4 kernels, strides 2,2,2,2,1) on 1x128x128 frames
        model_case('ae_arch2_1x128x128', RefAE, [1, 128, 128], 12, 6, 'ae',
                   arch_json='../../behavenet_amd/configs/ae_jsons/ae_arch_2.json')
        # two-camera frames larger than the tiles of the specialised kernels (default architecture)
        model_case('ae_2x192x160', RefAE, [2, 192, 160], 12, 4, 'ae')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'maxpool_valid':
        # MaxPool2d(ceil_mode=False) / odd pre-pool sizes (aes.py:173-178): 30x26 -> 26x22 -> 13x11
        # -> 9x7 -> 4x3 (the pools drop the last row / column)
        model_case('ae_maxpool_valid', RefAE, [1, 30, 26], 6, 12, 'ae',
                   arch_json='arch_maxpool_valid.json')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'condae':
        from behavenet.models.aes import ConditionalAE as RefCondAE
        model_case('condae_cfg1', RefCondAE, [1, 32, 32], 8, 210, 'cond-ae', n_labels=4,
                   extra_hp={'conditional_encoder': False}, store_xhat=True)
        model_case('condae_enc_cfg1', RefCondAE, [1, 32, 32], 8, 12, 'cond-ae', n_labels=4,
                   extra_hp={'conditional_encoder': True}, store_xhat=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'msps':
        msps_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'convdec':
        decoder_case('convdecoder_cfg1', [1, 32, 32], 4, 210)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'msp':
        from behavenet.models.aes import AEMSP as RefAEMSP
        model_case('aemsp_cfg1', RefAEMSP, [1, 32, 32], 8, 8, 'cond-ae-msp', n_labels=4,
                   extra_hp={'msp.alpha': 0.01, 'device': 'cpu'})
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'lastff':
        # dense last decoder layer (aes.py:345-359): 1x32x32 -> Linear(1024, 1024) + Sigmoid
        model_case('ae_cfg1_lastff', RefAE, [1, 32, 32], 8, 8, 'ae',
                   extra_hp={'ae_decoding_last_FF_layer': 1})
        sys.exit(0)
    planner_fixture()
    model_case('ae_cfg1', RefAE, [1, 32, 32], 8, 8, 'ae')
    model_case('ae_cfg1_b210', RefAE, [1, 32, 32], 8, 210, 'ae', store_xhat=True)
    model_case('ae_cfg2', RefAE, [1, 128, 128], 12, 4, 'ae')
    model_case('ae_1x64x48', RefAE, [1, 64, 48], 8, 6, 'ae')
    model_case('vae_cfg1', RefVAE, [1, 32, 32], 8, 8, 'vae',
               extra_hp={'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10})
    model_case('betatc_cfg1', RefBetaTC, [1, 32, 32], 8, 8, 'beta-tcvae',
               extra_hp={'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0, 'beta_tcvae.beta': 3.0,
                         'beta_tcvae.beta_anneal_epochs': 5, 'max_n_epochs': 10})
    model_case('condvae_cfg1', RefCondVAE, [1, 32, 32], 8, 8, 'cond-vae', n_labels=4,
               extra_hp={'vae.beta': 1.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 10,
                         'conditional_encoder': False})
    model_case('psvae_cfg4', RefPSVAE, [2, 128, 128], 16, 6, 'ps-vae', n_labels=4,
               extra_hp={'ps_vae.alpha': 1000, 'ps_vae.beta': 5, 'ps_vae.anneal_epochs': 100,
                         'max_n_epochs': 10}, store_xhat=False)
    fit_fixture()
