"""The json-config surface end to end on the MI355X: four config files -> grid -> one fit per
grid point (reference ae_grid_search.py:20-118 + hyperparam_utils.get_all_params), then the
bookkeeping the reference's analysis code relies on (experiment_exists, get_best_model_version,
get_best_model_and_data; reference fitting/utils.py:569-630,879-1063)."""

import json
import os
import pickle

import numpy as np
import pytest
import torch

from behavenet_amd.fitting import hyperparam_utils, utils
from behavenet_amd.fitting.ae_grid_search import run_grid

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(REPO, 'behavenet_amd', 'configs')


def _configs(tmp_path, data_dir, save_dir, **model_over):
    """The shipped json files with the data keys of the synthetic session and a short schedule."""
    out = {}
    for name, rel, over in (
            ('data', 'data_default.json',
             {'lab': 'lab', 'expt': 'expt', 'animal': 'animal', 'session': 'sess-0',
              'n_input_channels': 1, 'y_pixels': 32, 'x_pixels': 32}),
            ('model', os.path.join('ae_jsons', 'ae_model.json'),
             dict({'n_ae_latents': [4, 6], 'experiment_name': 'grid-test'}, **model_over)),
            ('training', os.path.join('ae_jsons', 'ae_training.json'),
             {'max_n_epochs': 2, 'min_n_epochs': 1, 'rng_seed_train': 0, 'export_latents': True,
              'export_train_plots': False}),
            ('compute', os.path.join('ae_jsons', 'ae_compute.json'), {'device': 'cuda'})):
        cfg = hyperparam_utils.load_config_json(os.path.join(CFG, rel))
        cfg.update(over)
        path = os.path.join(str(tmp_path), name + '.json')
        with open(path, 'w') as f:
            json.dump(cfg, f)
        out[name] = path
    os.environ['BEHAVENET_DATA_DIR'] = data_dir
    os.environ['BEHAVENET_SAVE_DIR'] = save_dir
    return ['--data_config', out['data'], '--model_config', out['model'],
            '--training_config', out['training'], '--compute_config', out['compute']]


def test_grid_search_from_json_configs(tmp_path, monkeypatch):
    from tests.test_fit_host import _write_sessions
    data_dir = os.path.join(str(tmp_path), 'data')
    save_dir = os.path.join(str(tmp_path), 'results')
    ids, sessions = _write_sessions(data_dir, n_sessions=1, n_trials=10, dim=(1, 32, 32),
                                    n_labels=0)
    args = _configs(tmp_path, data_dir, save_dir)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    try:
        ns = hyperparam_utils.get_all_params('grid_search', args=args)
        assert ns.data_dir == data_dir and ns.save_dir == save_dir
        results = run_grid(ns)
        assert len(results) == 2 and all(m is not None for _, m in results)
        sess_dir = os.path.join(save_dir, 'lab', 'expt', 'animal', 'sess-0')
        for (hp, model), n_lat in zip(results, (4, 6)):
            assert hp['n_ae_latents'] == n_lat and hp['training_completed'] is True
            expt_dir = os.path.join(sess_dir, 'ae', 'conv', '%02i_latents' % n_lat, 'grid-test')
            assert hp['expt_dir'] == expt_dir and hp['version'] == 0
            vdir = os.path.join(expt_dir, 'version_0')
            for name in ('best_val_model.pt', 'meta_tags.pkl', 'meta_tags.csv', 'metrics.csv',
                         'session_info.csv', 'lab_expt_animal_sess-0_latents.pkl'):
                assert os.path.exists(os.path.join(vdir, name)), name
            with open(os.path.join(vdir, 'lab_expt_animal_sess-0_latents.pkl'), 'rb') as f:
                lat = pickle.load(f)
            assert lat['latents'][0].shape[1] == n_lat
            assert utils.get_best_model_version(expt_dir) == [0]
        # the same grid again: every point is recognised as fitted
        again = run_grid(hyperparam_utils.get_all_params('grid_search', args=args))
        assert [m for _, m in again] == [None, None]

        # rebuild the fitted model + generator from the version directory
        hp_q = dict(results[1][0])
        for k in ('expt_dir', 'session_dir', 'version'):
            hp_q.pop(k)
        model, gen = utils.get_best_model_and_data(hp_q, version='best')
        assert model.version == 0 and gen.n_tot_batches == {'train': 8, 'val': 1, 'test': 1}
        fitted = results[1][1]
        for (k, a), (_, b) in zip(model.state_dict().items(), fitted.state_dict().items()):
            assert torch.equal(a, b), k
        import pandas as pd
        metrics = pd.read_csv(os.path.join(hp_q['expt_dir'] if 'expt_dir' in hp_q else
                                           results[1][0]['expt_dir'], 'version_0', 'metrics.csv'))
        gen.reset_iterators('val')
        data, sess = gen.next_batch('val')
        val = model.loss(data, dataset=sess, accumulate_grad=False)['loss']
        assert val == pytest.approx(metrics['val_loss'].min(), rel=1e-5)
        model2, none = utils.get_best_model_and_data(dict(hp_q), load_data=False, version=None)
        assert none is None and model2.version == 0
    finally:
        os.environ.pop('BEHAVENET_DATA_DIR', None)
        os.environ.pop('BEHAVENET_SAVE_DIR', None)
