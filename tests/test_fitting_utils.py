"""Paths and experiment bookkeeping (behavenet_amd/fitting/utils.py, experiment.py) against the
known answers of the reference's tests/test_fitting/test_utils_fitting.py (same directory tree,
same expected session / experiment directories) for the autoencoder classes, plus the version
bookkeeping (`experiment_exists`, `get_best_model_version`) over files written by the csv
experiment logger."""

import os
import pickle

import pytest

from behavenet_amd.fitting import utils
from behavenet_amd.fitting.experiment import Experiment

IDS = [
    {'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal0', 'session': 'session-00'},
    {'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal0', 'session': 'session-01'},
    {'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal0', 'session': 'session-02'},
    {'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal1', 'session': 'session-00'},
    {'lab': 'lab0', 'expt': 'expt1', 'animal': 'animal0', 'session': 'session-00'},
    {'lab': 'lab0', 'expt': 'expt1', 'animal': 'animal1', 'session': 'session-00'},
    {'lab': 'lab1', 'expt': 'expt0', 'animal': 'animal0', 'session': 'session-00'},
    {'lab': 'lab1', 'expt': 'expt0', 'animal': 'animal0', 'session': 'session-01'},
]
# multisession directories of the reference's fixture: relative path -> indices into IDS
MULTI = {
    'lab0/expt0/animal0/multisession-00': [0, 1, 2],
    'lab0/expt0/animal0/multisession-01': [1, 2],
    'lab0/expt0/animal1/multisession-03': [3],
    'lab0/expt0/animal1/multisession-04': [3],
    'lab0/expt0/multisession-00': [0, 1, 2, 3],
    'lab0/expt0/multisession-01': [0, 3],
    'lab0/multisession-00': [0, 1, 2, 3, 4, 5],
    'multisession-06': [0, 1, 2, 3, 4, 5, 6],
}


@pytest.fixture
def tree(tmp_path):
    root = str(tmp_path)
    for ids in IDS:
        os.makedirs(os.path.join(root, ids['lab'], ids['expt'], ids['animal'], ids['session']))
    for rel, idxs in MULTI.items():
        utils.export_session_info_to_csv(os.path.join(root, rel), [IDS[i] for i in idxs])
    return root


def _key(d):
    return ''.join(str(v) for v in d.values())


def test_get_subdirs(tree):
    with pytest.raises(StopIteration):
        utils.get_subdirs(os.path.join(tree, 'lab0', 'multisession-00'))
    assert sorted(utils.get_subdirs(os.path.join(tree, 'lab0'))) == \
        ['expt0', 'expt1', 'multisession-00']
    with pytest.raises(NotADirectoryError):
        utils.get_subdirs('/ZzZtestingZzZ')


def test_get_multisession_paths(tree):
    assert utils._get_multisession_paths(os.path.join(tree, 'lab1')) == []
    assert utils._get_multisession_paths(tree, lab='lab0') == \
        [os.path.join(tree, 'lab0', 'multisession-00')]
    e0 = os.path.join(tree, 'lab0', 'expt0')
    assert sorted(utils._get_multisession_paths(tree, lab='lab0', expt='expt0')) == \
        [os.path.join(e0, 'multisession-00'), os.path.join(e0, 'multisession-01')]
    a1 = os.path.join(e0, 'animal1')
    assert sorted(utils._get_multisession_paths(tree, lab='lab0', expt='expt0', animal='animal1')) \
        == [os.path.join(a1, 'multisession-03'), os.path.join(a1, 'multisession-04')]
    assert utils._get_multisession_paths(tree, lab='lab1', expt='expt0', animal='animal0') == []
    assert utils._get_multisession_paths('fakepath', lab='lab1', expt='expt0') == []


def test_get_single_sessions(tree):
    found = utils._get_single_sessions(tree, depth=4, curr_depth=0)
    for ids in IDS:
        assert ids in found


def test_get_session_dir_from_csv(tree):
    hp = {'data_dir': tree, 'save_dir': tree}
    cases = [   # (csv relative dir, expected session dir)
        ('lab0/expt0/animal1/multisession-03', 'lab0/expt0/animal1/session-00'),
        ('lab0/expt0/animal0/multisession-00', 'lab0/expt0/animal0/multisession-00'),
        ('lab0/expt0/multisession-00', 'lab0/expt0/multisession-00'),
        ('lab0/multisession-00', 'lab0/multisession-00'),
    ]
    for rel, want in cases:
        hp['sessions_csv'] = os.path.join(tree, rel, 'session_info.csv')
        sess_dir, single = utils.get_session_dir(hp, session_source='save')
        assert sess_dir == os.path.join(tree, want)
        assert single == [IDS[i] for i in MULTI[rel]]
    hp['sessions_csv'] = os.path.join(tree, 'multisession-06', 'session_info.csv')
    with pytest.raises(NotImplementedError):          # several labs
        utils.get_session_dir(hp, session_source='save')


def test_get_session_dir_from_all_keys(tree):
    hp = {'data_dir': tree, 'save_dir': tree, 'sessions_csv': '', 'lab': 'all', 'expt': 'all',
          'animal': 'all', 'session': 'all'}
    with pytest.raises(NotImplementedError):
        utils.get_session_dir(hp)
    cases = [   # (lab, expt, animal, session) -> (dir, indices)
        (('lab0', 'all', '', ''), 'lab0/multisession-00', [0, 1, 2, 3, 4, 5]),
        (('lab0', 'expt0', 'all', ''), 'lab0/expt0/multisession-00', [0, 1, 2, 3]),
        (('lab0', 'expt0', 'animal0', 'all'), 'lab0/expt0/animal0/multisession-00', [0, 1, 2]),
        (('lab0', 'expt0', 'animal0', 'session-00'), 'lab0/expt0/animal0/session-00', [0]),
        # no multisession with this session set yet: the next free index (0)
        (('lab1', 'expt0', 'animal0', 'all'), 'lab1/expt0/animal0/multisession-00', [6, 7]),
    ]
    for (lab, expt, animal, session), want, idxs in cases:
        hp.update({'lab': lab, 'expt': expt, 'animal': animal, 'session': session})
        sess_dir, single = utils.get_session_dir(hp, session_source='save')
        assert sess_dir == os.path.join(tree, want)
        assert sorted(_key(d) for d in single) == sorted(_key(IDS[i]) for i in idxs)
    # an explicitly requested existing multisession
    hp.update({'lab': 'lab0', 'expt': 'expt0', 'animal': 'animal0', 'session': 'all',
               'multisession': 1})
    sess_dir, single = utils.get_session_dir(hp)
    assert sess_dir == os.path.join(tree, 'lab0/expt0/animal0/multisession-01')
    assert sorted(_key(d) for d in single) == sorted(_key(IDS[i]) for i in (1, 2))
    with pytest.raises(ValueError):
        utils.get_session_dir(hp, session_source='test')


def test_get_expt_dir(tree):
    hp = {'data_dir': 'ddir', 'save_dir': 'sdir', 'lab': 'lab0', 'expt': 'expt0',
          'animal': 'animal0', 'session': 'session-00', 'experiment_name': 'tt_expt'}
    session_dir = os.path.join('ddir', 'lab0', 'expt0', 'animal0', 'session-00')
    hp['session_dir'] = session_dir
    for model_class, n_lat in (('ae', 8), ('vae', 10), ('beta-tcvae', 10), ('cond-vae', 8),
                               ('cond-ae', 8), ('cond-ae-msp', 8), ('ps-vae', 10),
                               ('msps-vae', 11)):
        hp.update({'model_class': model_class, 'model_type': 'conv', 'n_ae_latents': n_lat})
        want = os.path.join(session_dir, model_class, 'conv', '%02i_latents' % n_lat, 'tt_expt')
        assert utils.get_expt_dir(hp) == want
        assert utils.get_expt_dir(hp, model_class=model_class, model_type='conv',
                                  expt_name='tt_expt') == want
    # multisession autoencoder
    hp.update({'model_class': 'ae', 'n_ae_latents': 8, 'save_dir': tree, 'ae_multisession': 0})
    assert utils.get_expt_dir(hp) == os.path.join(
        tree, 'lab0', 'expt0', 'animal0', 'multisession-00', 'ae', 'conv', '08_latents', 'tt_expt')
    hp['ae_multisession'] = None
    for out_of_scope in ('arhmm', 'neural-ae', 'arhmm-labels', 'bayesian-decoding'):
        with pytest.raises(NotImplementedError):
            utils.get_expt_dir(hp, model_class=out_of_scope, model_type='mlp')
    with pytest.raises(ValueError):
        utils.get_expt_dir(hp, model_class='testing', model_type='mlp')


def test_contains_session_and_find_session_dirs(tree):
    multi = os.path.join(tree, 'lab0', 'expt0', 'animal0', 'multisession-01')
    assert utils.contains_session(multi, IDS[1])
    assert not utils.contains_session(multi, IDS[0])
    hp = dict(IDS[0], save_dir=tree)
    dirs, ids = utils.find_session_dirs(hp)
    want = ['lab0/multisession-00', 'lab0/expt0/multisession-00', 'lab0/expt0/multisession-01',
            'lab0/expt0/animal0/multisession-00', 'lab0/expt0/animal0/session-00']
    assert sorted(dirs) == sorted(os.path.join(tree, w) for w in want)
    assert sum(i['multisession'] is None for i in ids) == 1


def _hparams(tree, **kw):
    hp = {'save_dir': tree, 'data_dir': tree, 'sessions_csv': '', 'all_source': 'save',
          'experiment_name': 'grid', 'model_class': 'ae', 'model_type': 'conv',
          'n_ae_latents': 8, 'rng_seed_data': 0, 'trial_splits': '8;1;1;0', 'train_frac': 1.0,
          'rng_seed_model': 0, 'fit_sess_io_layers': False, 'learning_rate': 1e-4, 'l2_reg': 0.0}
    hp.update(IDS[0])
    hp.update(kw)
    return hp


def test_get_model_params():
    hp = _hparams('x', model_class='ps-vae', **{'ps_vae.alpha': 1000, 'ps_vae.beta': 5})
    less = utils.get_model_params(hp)
    assert less['ps_vae.alpha'] == 1000 and less['ps_vae.beta'] == 5 and 'vae.beta' not in less
    assert set(less) == {'rng_seed_data', 'trial_splits', 'train_frac', 'rng_seed_model',
                         'model_class', 'model_type', 'n_ae_latents', 'fit_sess_io_layers',
                         'learning_rate', 'l2_reg', 'ps_vae.alpha', 'ps_vae.beta'}
    assert 'vae.beta' in utils.get_model_params(_hparams('x', model_class='vae', **{'vae.beta': 2}))
    assert utils.get_model_params(_hparams('x', model_class='cond-ae'))['conditional_encoder'] \
        is False
    with pytest.raises(NotImplementedError):
        utils.get_model_params(_hparams('x', model_class='arhmm'))


def test_experiment_versions_and_bookkeeping(tree):
    """create_experiment -> version_0; a finished fit makes the same grid point 'exist'; another
    grid point gets version_1; get_best_model_version reads metrics.csv."""
    hp = _hparams(tree)
    with pytest.raises(NotADirectoryError):      # as the reference: the directory must exist
        utils.experiment_exists(dict(hp))
    hp0, sess_ids, exp = utils.create_experiment(hp)
    assert sess_ids == [IDS[0]] and exp.version == 0 and hp0['version'] == 0
    vdir = os.path.join(tree, 'lab0', 'expt0', 'animal0', 'session-00', 'ae', 'conv',
                        '08_latents', 'grid', 'version_0')
    assert os.path.isdir(vdir)
    hp0['training_completed'] = False
    utils.export_hparams(hp0, exp)
    assert utils.experiment_exists(_hparams(tree)) is False          # not finished yet
    for epoch, val in enumerate([0.5, 0.3, 0.4]):
        exp.log({'epoch': epoch, 'tr_loss': val + 0.1, 'dataset': -1})
        exp.log({'epoch': epoch, 'val_loss': val, 'best_val_epoch': 1, 'dataset': -1})
    exp.save()
    hp0['training_completed'] = True
    utils.export_hparams(hp0, exp)
    with open(os.path.join(vdir, 'meta_tags.pkl'), 'rb') as f:
        assert pickle.load(f)['training_completed'] is True
    assert os.path.exists(os.path.join(vdir, 'meta_tags.csv'))
    assert utils.experiment_exists(_hparams(tree), which_version=True) == (True, 0)
    assert utils.create_experiment(_hparams(tree)) == (None, None, None)   # fitted already

    hp1, _, exp1 = utils.create_experiment(_hparams(tree, learning_rate=1e-3))
    assert exp1.version == 1
    hp1['training_completed'] = True
    utils.export_hparams(hp1, exp1)
    exp1.log({'epoch': 0, 'val_loss': 0.2, 'dataset': -1})
    exp1.save()
    assert utils.experiment_exists(_hparams(tree, learning_rate=1e-3), which_version=True) == \
        (True, 1)
    assert utils.experiment_exists(_hparams(tree, learning_rate=3e-3), which_version=True) == \
        (False, None)
    expt_dir = os.path.dirname(vdir)
    assert utils.get_best_model_version(expt_dir) == [1]
    assert utils.get_best_model_version(expt_dir, n_best=2) == [1, 0]
    assert utils.get_best_model_version(expt_dir, best_def='max') == [0]


def test_experiment_logger_files(tmp_path):
    import pandas as pd
    exp = Experiment(name='e', save_dir=str(tmp_path))
    assert exp.version == 0
    exp.log({'epoch': 0, 'tr_loss': 1.0})
    exp.log({'epoch': 0, 'val_loss': 2.0, 'best_val_epoch': 0})
    exp.tag({'learning_rate': 1e-4, 'model_class': 'ae'})
    exp.save()
    df = pd.read_csv(os.path.join(str(tmp_path), 'e', 'version_0', 'metrics.csv'))
    assert list(df['epoch']) == [0, 0] and df['val_loss'].min() == 2.0
    assert pd.isna(df['val_loss'][0]) and df['tr_loss'][0] == 1.0
    tags = pd.read_csv(os.path.join(str(tmp_path), 'e', 'version_0', 'meta_tags.csv'))
    assert set(tags['key']) == {'learning_rate', 'model_class'}
    assert Experiment(name='e', save_dir=str(tmp_path)).version == 1
    assert Experiment(name='e', save_dir=str(tmp_path), version=7).version == 7


def _claim_version(args):
    save_dir, barrier_file = args
    import time
    while not os.path.exists(barrier_file):     # start all claimants at (nearly) the same time
        time.sleep(0.001)
    return Experiment(name='e', save_dir=save_dir).version


def test_experiment_versions_are_claimed_atomically(tmp_path):
    """The ranks of a grid search create their experiments at the same moment under one name
    (one grid point per rank): every process must get its own version_K."""
    import multiprocessing as mp
    flag = os.path.join(str(tmp_path), 'go')
    with mp.get_context('fork').Pool(8) as pool:
        res = pool.map_async(_claim_version, [(str(tmp_path), flag)] * 8, chunksize=1)
        open(flag, 'w').close()
        versions = res.get(timeout=60)
    assert sorted(versions) == list(range(8))


def _dp_create(rank, world, port, tree, out):
    import torch
    from behavenet_amd.fitting import distributed as bdist
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world)})
    torch.set_num_threads(1)
    bdist.init_from_env(backend='gloo', timeout_s=60)
    hp, _, exp = utils.create_experiment(_hparams(tree))
    hp['training_completed'] = True
    utils.export_hparams(hp, exp)
    exp.log({'epoch': 0, 'val_loss': 0.5 + rank, 'dataset': -1})
    exp.save()
    torch.distributed.barrier()
    again = utils.create_experiment(_hparams(tree))
    out.put((rank, exp.version, hp['version'], bool(exp.debug), again == (None, None, None)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_ranks_of_one_data_parallel_fit_share_one_version(tree):
    """Two ranks of ONE fit (process group up) call create_experiment: rank 0 claims version_0 and
    writes its files, rank 1 logs into a file-less experiment of the same version -- no second
    version_K with a finished meta_tags and no model in it, and both ranks agree that
    the grid point exists afterwards."""
    import pandas as pd
    import torch.multiprocessing as mp
    from tests.test_distributed_cpu import _free_port
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_create, args=(r, 2, port, tree, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 0, 0, False, True), (1, 0, 0, True, True)]
    expt_dir = os.path.join(tree, 'lab0', 'expt0', 'animal0', 'session-00', 'ae', 'conv',
                            '08_latents', 'grid')
    assert sorted(os.listdir(expt_dir)) == ['version_0']
    df = pd.read_csv(os.path.join(expt_dir, 'version_0', 'metrics.csv'))
    assert list(df['val_loss']) == [0.5]                    # rank 0's row only


def test_data_generator_inputs():
    from behavenet_amd.data.utils import get_data_generator_inputs
    from behavenet_amd.data.transforms import MakeOneHot2D
    hp = {'data_dir': 'd', 'model_class': 'ae'}
    _, sig, tr, paths = get_data_generator_inputs(hp, IDS[:2])
    assert sig == [['images'], ['images']] and tr == [[None], [None]]
    assert paths[1] == [os.path.join('d', 'lab0', 'expt0', 'animal0', 'session-01', 'data.hdf5')]
    hp = {'data_dir': 'd', 'model_class': 'ps-vae', 'use_output_mask': True,
          'use_label_mask': True}
    _, sig, _, _ = get_data_generator_inputs(hp, IDS[:1])
    assert sig == [['images', 'labels', 'masks', 'labels_masks']]
    hp = {'data_dir': 'd', 'model_class': 'cond-ae', 'conditional_encoder': True,
          'y_pixels': 4, 'x_pixels': 6, 'use_label_mask': True}
    _, sig, tr, _ = get_data_generator_inputs(hp, IDS[:1])
    assert sig == [['images', 'labels', 'labels_sc']] and isinstance(tr[0][2], MakeOneHot2D)
    with pytest.raises(NotImplementedError):
        get_data_generator_inputs({'data_dir': 'd', 'model_class': 'neural-ae'}, IDS[:1])


def test_make_one_hot_2d():
    """The docstring example of the reference transform (transforms.py:192-195)."""
    import numpy as np
    from behavenet_amd.data.transforms import MakeOneHot2D
    out = MakeOneHot2D(128, 128)(np.array([[64., 34., 56., 102.]]))
    assert out.shape == (1, 2, 128, 128) and out.sum() == 2
    assert out[0, 0, 56, 64] == 1 and out[0, 1, 102, 34] == 1
    out = MakeOneHot2D(8, 8)(np.array([[np.nan, 100., -3., 2.4]]))     # nan / clip / round
    assert out[0, 0, 0, 0] == 1 and out[0, 1, 2, 7] == 1


def test_paths_and_bookkeeping_match_the_reference_module(tree):
    """tests/golden/fitting_utils.json holds what the REFERENCE's behavenet.fitting.utils returned
    (tests/golden/make_golden.py fitting_utils, run against /root/reference) on this same session
    tree: get_session_dir, get_expt_dir, get_model_params, find_session_dirs and experiment_exists
    over meta_tags.pkl files -- the functions here must return the same."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                           'fitting_utils.json')) as f:
        want = json.load(f)

    def rel(path):
        return os.path.relpath(path, tree)

    def norm_ids(lst):
        return sorted('/'.join(str(d[k]) for k in ('lab', 'expt', 'animal', 'session')) for d in lst)
    for case in want['session_dir']:
        hp = dict(case['hparams'], data_dir=tree, save_dir=tree)
        d, single = utils.get_session_dir(hp, session_source='save')
        assert rel(d) == case['dir'] and norm_ids(single) == case['sessions'], case
    base = dict(want['base_hparams'], data_dir=tree, save_dir=tree,
                session_dir=os.path.join(tree, 'lab0', 'expt0', 'animal0', 'session-00'))
    for e, m in zip(want['expt_dir'], want['model_params']):
        hp = dict(base, model_class=e['model_class'], n_ae_latents=e['n_ae_latents'])
        assert rel(utils.get_expt_dir(dict(hp))) == e['dir']
        got = utils.get_model_params(dict(hp))
        assert set(got) == set(m['params']), (e['model_class'], set(got) ^ set(m['params']))
        for k, v in m['params'].items():
            assert got[k] == v, (e['model_class'], k)
    dirs, ids = utils.find_session_dirs(dict(IDS[0], save_dir=tree))
    assert sorted(rel(d) for d in dirs) == want['find_session_dirs']['dirs']
    assert sum(i['multisession'] is None for i in ids) == want['find_session_dirs']['n_single']
    # experiment_exists over the same three meta_tags.pkl files
    hp = dict(base, model_class='ae', n_ae_latents=8)
    expt_dir = utils.get_expt_dir(dict(hp))
    hp['expt_dir'] = expt_dir
    for v, (lr, l2) in enumerate(((1e-4, 0.0), (1e-3, 0.0), (1e-4, 1e-5))):
        vdir = os.path.join(expt_dir, 'version_%i' % v)
        os.makedirs(vdir)
        with open(os.path.join(vdir, 'meta_tags.pkl'), 'wb') as f:
            pickle.dump(dict(utils.get_model_params(dict(hp, learning_rate=lr, l2_reg=l2)),
                             training_completed=(v != 1)), f)
    for case in want['experiment_exists']:
        probe = dict(hp, learning_rate=case['learning_rate'], l2_reg=case['l2_reg'])
        exists, version = utils.experiment_exists(dict(probe), which_version=True)
        assert bool(exists) == case['exists'], case
        assert (None if version is None else int(version)) == case['version'], case


def test_package_level_directories_and_lab_example(tmp_path, monkeypatch):
    """behavenet/__init__.py:5-53 and fitting/utils.py:780-803: the user directory helpers a script written against the
    reference imports from the package, and the per-dataset parameter file."""
    import json
    import os
    import behavenet_amd
    from behavenet_amd.fitting.utils import get_lab_example
    monkeypatch.setenv('HOME', str(tmp_path))
    for k in ('BEHAVENET_DATA_DIR', 'BEHAVENET_SAVE_DIR', 'BEHAVENET_FIGS_DIR'):
        monkeypatch.delenv(k, raising=False)
    assert behavenet_amd.get_params_dir() == os.path.join(str(tmp_path), '.behavenet')
    assert behavenet_amd.get_user_dir('data') == os.path.join(str(tmp_path), '.behavenet', 'data')
    os.makedirs(behavenet_amd.get_params_dir())
    with open(os.path.join(behavenet_amd.get_params_dir(), 'directories.json'), 'w') as f:
        json.dump({'data_dir': '/d', 'save_dir': '/s', 'figs_dir': '/f'}, f)
    assert [behavenet_amd.get_user_dir(k) for k in ('data', 'save', 'figs')] == ['/d', '/s', '/f']
    with open(os.path.join(behavenet_amd.get_params_dir(), 'musall_vistrained_params.json'), 'w') as f:
        json.dump({'y_pixels': 128, 'x_pixels': 128, 'n_input_channels': 2}, f)
    hp = {'lab': 'musall', 'y_pixels': 1}
    get_lab_example(hp, 'musall', 'vistrained')
    assert hp == {'lab': 'musall', 'y_pixels': 128, 'x_pixels': 128, 'n_input_channels': 2}
    target = os.path.join(str(tmp_path), 'a', 'b', 'file.pkl')
    behavenet_amd.make_dir_if_not_exists(target)
    assert os.path.isdir(os.path.dirname(target)) and not os.path.exists(target)
    behavenet_amd.make_dir_if_not_exists(target)                    # twice: no error
