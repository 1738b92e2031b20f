"""Model paths and experiment bookkeeping of the autoencoder path.

Host-side mirror of the parts of the reference ``behavenet/fitting/utils.py`` that
``ae_grid_search.py`` and analysis code touch: session / experiment directories
(``save_dir/lab/expt/animal/session[/multisession-xx]/<model_class>/<model_type>/NN_latents/
<experiment_name>/version_K``), the ``session_info.csv`` of multi-session fits, the
"was this grid point fitted already" check over ``meta_tags.pkl``, the choice of the best version
from ``metrics.csv`` and the re-construction of a fitted model (+ data generator) from a version
directory.  File names and formats are the reference's, so results written by either code base
are readable by the other.  Only the autoencoder model classes are served; the ARHMM / neural
decoder branches raise ``NotImplementedError`` (out of scope).

Reference lines: get_subdirs :16-38, _get_multisession_paths :41-76, _get_single_sessions :79-112,
get_session_dir :135-304, get_expt_dir :307-434, session csv helpers :437-505,
find_session_dirs :508-566, experiment_exists :569-630, get_model_params :633-753,
export_hparams :756-777, create_tt_experiment :838-876, get_best_model_version :879-941,
get_best_model_and_data :944-1063, _clean_tt_dir :1066-1073, _print_hparams :1076-1085.
"""

import copy
import csv
import os
import pickle
import shutil

import numpy as np

__all__ = [
    'get_subdirs', 'get_session_dir', 'get_expt_dir', 'read_session_info_from_csv',
    'export_session_info_to_csv', 'contains_session', 'find_session_dirs', 'experiment_exists',
    'get_model_params', 'export_hparams', 'create_tt_experiment', 'create_experiment',
    'get_best_model_version', 'get_best_model_and_data', 'get_lab_example']

AE_CLASSES = ('ae', 'vae', 'beta-tcvae', 'cond-vae', 'cond-ae', 'cond-ae-msp', 'ps-vae',
              'msps-vae')
_ID_KEYS = ('lab', 'expt', 'animal', 'session')


def _out_of_scope(model_class):
    raise NotImplementedError(
        'model class "%s" is outside the MI355X autoencoder path (SURVEY.md section 2)' %
        model_class)


# ------------------------------------------------------------------------------------------
# directory walking
# ------------------------------------------------------------------------------------------
def get_subdirs(path):
    """First-level subdirectory names of ``path``; ``NotADirectoryError`` if it does not exist,
    ``StopIteration`` if it has none (the reference's error contract, relied on by callers)."""
    if not os.path.exists(path):
        raise NotADirectoryError('%s is not a path' % path)
    subdirs = [e.name for e in os.scandir(path) if e.is_dir()]
    if len(subdirs) == 0:
        raise StopIteration('%s does not contain any subdirectories' % path)
    return subdirs


def _get_multisession_paths(base_dir, lab='', expt='', animal=''):
    """Absolute paths of the ``multisession-xx`` directories directly under
    ``base_dir/lab/expt/animal`` (empty strings skip a level); [] if there is nothing."""
    root = os.path.join(base_dir, lab, expt, animal)
    try:
        return [os.path.join(root, d) for d in get_subdirs(root) if d.startswith('multi')]
    except (ValueError, NotADirectoryError, StopIteration):
        print('warning: did not find any sessions')
        return []


def _get_single_sessions(base_dir, depth, curr_depth):
    """All single sessions ``depth`` levels below ``base_dir`` (multisession dirs skipped) as
    {'lab','expt','animal','session'} dicts read off the last four path components."""
    if curr_depth == depth:
        parts = str(base_dir).split(os.sep)
        return [dict(zip(_ID_KEYS, parts[-4:]))]
    found = []
    for sub in get_subdirs(base_dir):
        if not sub.startswith('multisession'):
            found += _get_single_sessions(os.path.join(base_dir, sub), depth, curr_depth + 1)
    return found


def read_session_info_from_csv(session_file):
    """Rows of a ``session_info.csv`` as dicts (columns lab, expt, animal, session)."""
    with open(session_file) as f:
        return [dict(row) for row in csv.DictReader(f)]


def export_session_info_to_csv(session_dir, ids_list):
    """Write ``session_dir/session_info.csv`` with one row per session dict."""
    os.makedirs(session_dir, exist_ok=True)
    with open(os.path.join(session_dir, 'session_info.csv'), mode='w') as f:
        writer = csv.DictWriter(f, fieldnames=list(ids_list[0].keys()))
        writer.writeheader()
        for ids in ids_list:
            writer.writerow(ids)


def _without_save_dir(rows):
    for row in rows:
        row.pop('save_dir', None)
    return rows


def contains_session(session_dir, session_id):
    """Is ``session_id`` one of the rows of ``session_dir/session_info.csv``?"""
    rows = _without_save_dir(
        read_session_info_from_csv(os.path.join(session_dir, 'session_info.csv')))
    return any(row == session_id for row in rows)


def get_session_dir(hparams, session_source='save'):
    """-> (session_dir, list of single-session id dicts).

    ``sessions_csv`` (non-empty) wins; else 'all' in ``expt`` / ``animal`` / ``session`` selects
    every session below that level (searched in ``save_dir`` or ``data_dir``), else the one
    named session.  More than one session => ``<level>/multisession-xx`` where xx is the index
    of an existing multisession with exactly this session set, or the next free index; a given
    ``hparams['multisession']`` index selects that existing one outright.
    """
    save_dir = hparams['save_dir']
    if session_source == 'save':
        search_dir = hparams['save_dir']
    elif session_source == 'data':
        search_dir = hparams['data_dir']
    else:
        raise ValueError('"%s" is an invalid session_source' % session_source)

    use_csv = len(hparams.get('sessions_csv', [])) > 0
    if use_csv:
        sessions_single = _without_save_dir(read_session_info_from_csv(hparams['sessions_csv']))
        cols = {k: np.array([s[k] for s in sessions_single]) for k in _ID_KEYS}
        first = sessions_single[0]
        # the deepest level at which all rows agree fixes the directory
        if len(np.unique(cols['session'])) == 1:
            level = [first['lab'], first['expt'], first['animal'], first['session']]
        elif len(np.unique(cols['animal'])) == 1:
            level = [first['lab'], first['expt'], first['animal']]
        elif len(np.unique(cols['expt'])) == 1:
            level = [first['lab'], first['expt']]
        elif len(np.unique(cols['lab'])) == 1:
            level = [first['lab']]
        else:
            raise NotImplementedError('multiple labs not currently supported')
        session_dir_base = os.path.join(save_dir, *level)
        multi_level = (level + ['', '', ''])[:3]
        multisession_paths = _get_multisession_paths(save_dir, *multi_level)
    else:
        lab = hparams['lab']
        if lab == 'all':
            raise NotImplementedError('multiple labs not currently supported')
        if hparams['expt'] == 'all':
            level, depth = [lab], 3
        elif hparams['animal'] == 'all':
            level, depth = [lab, hparams['expt']], 2
        elif hparams['session'] == 'all':
            level, depth = [lab, hparams['expt'], hparams['animal']], 1
        else:
            level, depth = [lab, hparams['expt'], hparams['animal'], hparams['session']], 0
        session_dir_base = os.path.join(save_dir, *level)
        if depth == 0:
            multisession_paths = []
            sessions_single = [{k: hparams[k] for k in _ID_KEYS}]
        else:
            multisession_paths = _get_multisession_paths(save_dir, *(level + ['', ''])[:3])
            sessions_single = _get_single_sessions(
                os.path.join(search_dir, *level), depth=depth, curr_depth=0)

    if hparams.get('multisession', None) is not None and not use_csv:
        session_dir = os.path.join(session_dir_base, 'multisession-%02i' % hparams['multisession'])
        sessions_single = _without_save_dir(
            read_session_info_from_csv(os.path.join(session_dir, 'session_info.csv')))
    elif len(sessions_single) > 1:
        want = set(tuple(sorted(d.items())) for d in sessions_single)
        multi_idx = None
        for path in multisession_paths:
            rows = _without_save_dir(
                read_session_info_from_csv(os.path.join(path, 'session_info.csv')))
            if set(tuple(sorted(d.items())) for d in rows) == want:
                multi_idx = int(path.split('-')[-1])
                break
        if multi_idx is None:
            taken = [int(path.split('-')[-1]) for path in multisession_paths]
            multi_idx = max(taken) + 1 if taken else 0
        session_dir = os.path.join(session_dir_base, 'multisession-%02i' % multi_idx)
    else:
        session_dir = session_dir_base
    return session_dir, sessions_single


def get_expt_dir(hparams, model_class=None, model_type=None, expt_name=None):
    """``session_dir/<model_class>/<model_type>/NN_latents/<experiment_name>`` for the
    autoencoder classes (with ``ae_multisession`` the session dir is that multisession's)."""
    model_class = hparams['model_class'] if model_class is None else model_class
    model_type = hparams['model_type'] if model_type is None else model_type
    expt_name = hparams['experiment_name'] if expt_name is None else expt_name
    if model_class in AE_CLASSES:
        model_path = os.path.join(model_class, model_type, '%02i_latents' % hparams['n_ae_latents'])
        if hparams.get('ae_multisession', None) is not None:
            hp = copy.deepcopy(hparams)
            hp['session'] = 'all'
            hp['multisession'] = hparams['ae_multisession']
            session_dir, _ = get_session_dir(hp)
        else:
            session_dir = hparams['session_dir']
    elif model_class == 'labels-images':
        model_path = os.path.join(model_class, model_type)
        session_dir = hparams['session_dir']
    elif model_class in ('neural-ae', 'neural-ae-me', 'ae-neural', 'neural-labels', 'labels-neural',
                         'neural-arhmm', 'arhmm-neural', 'arhmm', 'hmm', 'arhmm-labels',
                         'hmm-labels', 'bayesian-decoding'):
        _out_of_scope(model_class)
    else:
        raise ValueError('"%s" is an invalid model class' % model_class)
    return os.path.join(session_dir, model_path, expt_name)


def find_session_dirs(hparams):
    """Every session directory under ``save_dir/lab`` (single or multisession, at any level) that
    contains the session named in ``hparams`` -> (paths, id dicts with 'multisession')."""
    ids = {k: hparams[k] for k in _ID_KEYS}
    save_dir, lab = hparams['save_dir'], hparams['lab']
    dirs, found = [], []

    def multi(path, expt, animal, session):
        if contains_session(path, ids):
            dirs.append(path)
            found.append({'lab': lab, 'expt': expt, 'animal': animal, 'session': session,
                          'multisession': int(path[-2:])})

    for expt in get_subdirs(os.path.join(save_dir, lab)):
        expt_path = os.path.join(save_dir, lab, expt)
        if expt.startswith('multi'):
            multi(expt_path, 'all', '', '')
            continue
        for animal in get_subdirs(expt_path):
            animal_path = os.path.join(expt_path, animal)
            if animal.startswith('multi'):
                multi(animal_path, expt, 'all', '')
                continue
            for session in get_subdirs(animal_path):
                session_path = os.path.join(animal_path, session)
                if session.startswith('multi'):
                    multi(session_path, expt, animal, 'all')
                elif {'lab': lab, 'expt': expt, 'animal': animal, 'session': session} == ids:
                    dirs.append(session_path)
                    found.append({'lab': lab, 'expt': expt, 'animal': animal, 'session': session,
                                  'multisession': None})
    return dirs, found


# ------------------------------------------------------------------------------------------
# experiments
# ------------------------------------------------------------------------------------------
def get_model_params(hparams):
    """The hparams that identify a fit within its experiment directory (ref :633-753)."""
    model_class = hparams['model_class']
    less = {k: hparams[k] for k in ('rng_seed_data', 'trial_splits', 'train_frac',
                                    'rng_seed_model', 'model_class', 'model_type')}
    if model_class in AE_CLASSES:
        for k in ('n_ae_latents', 'fit_sess_io_layers', 'learning_rate', 'l2_reg'):
            less[k] = hparams[k]
        if model_class in ('cond-ae', 'cond-vae'):
            less['conditional_encoder'] = hparams.get('conditional_encoder', False)
        if model_class == 'cond-ae-msp':
            less['msp.alpha'] = hparams['msp.alpha']
        if model_class in ('vae', 'cond-vae'):
            less['vae.beta'] = hparams['vae.beta']
        if model_class == 'beta-tcvae':
            less['beta_tcvae.beta'] = hparams['beta_tcvae.beta']
        if model_class in ('ps-vae', 'msps-vae'):
            less['ps_vae.alpha'] = hparams['ps_vae.alpha']
            less['ps_vae.beta'] = hparams['ps_vae.beta']
        if model_class == 'msps-vae':
            for k in ('ps_vae.delta', 'n_background', 'n_sessions_per_batch'):
                less[k] = hparams[k]
    elif model_class == 'labels-images':
        for k in ('fit_sess_io_layers', 'learning_rate', 'l2_reg'):
            less[k] = hparams[k]
    elif model_class in ('arhmm', 'hmm', 'arhmm-labels', 'hmm-labels', 'neural-ae', 'neural-ae-me',
                         'ae-neural', 'neural-labels', 'labels-neural', 'neural-arhmm',
                         'arhmm-neural', 'bayesian-decoding'):
        _out_of_scope(model_class)
    else:
        raise NotImplementedError('"%s" is not a valid model class' % model_class)
    return less


def experiment_exists(hparams, which_version=False):
    """Has a version with the same identifying hparams finished training?  Fills in
    ``session_dir`` / ``expt_dir`` if absent.  -> bool, or (bool, version | None)."""
    if 'expt_dir' not in hparams:
        if 'session_dir' not in hparams:
            hparams['session_dir'], _ = get_session_dir(
                hparams, session_source=hparams.get('all_source', 'save'))
        hparams['expt_dir'] = get_expt_dir(hparams)
    try:
        versions = get_subdirs(hparams['expt_dir'])
    except StopIteration:
        return (False, None) if which_version else False
    want = get_model_params(hparams)
    match = None
    for version in versions:
        try:
            with open(os.path.join(hparams['expt_dir'], version, 'meta_tags.pkl'), 'rb') as f:
                have = pickle.load(f)
        except IOError:
            continue
        if all(have[k] == v for k, v in want.items()) and have['training_completed']:
            match = version
            break
    if which_version:
        return (True, int(match.split('_')[-1])) if match is not None else (False, None)
    return match is not None


def export_hparams(hparams, exp):
    """``meta_tags.pkl`` (pickled dict) + ``meta_tags.csv`` (through ``exp.tag``) of a version."""
    if getattr(exp, 'debug', False):
        return          # a non-main rank of a data-parallel fit: rank 0 writes the version's files
    meta_file = os.path.join(hparams['expt_dir'], 'version_%i' % exp.version, 'meta_tags.pkl')
    with open(meta_file, 'wb') as f:
        pickle.dump(hparams, f)
    exp.tag(hparams)
    exp.save()


def create_experiment(hparams):
    """Prepare the directories of a fit and open a fresh ``version_K`` in them.

    -> (hparams, sess_ids, exp), or (None, None, None) if this grid point was fitted already.
    The experiment object is :class:`behavenet_amd.fitting.experiment.Experiment` (csv files in
    test-tube's layout).
    """
    from behavenet_amd.fitting.experiment import Experiment
    from behavenet_amd.fitting import distributed as bdist
    # The ranks of ONE data-parallel fit (a process group is up) share one version: rank 0 decides
    # whether the grid point exists, claims version_K and tells the others, which log into a
    # file-less Experiment of the same K (rank 0 writes metrics and checkpoints, `fit`).  Ranks of
    # a grid search (one grid point per rank, no process group: `run_grid`) each claim their own.
    dp = bdist.is_active() and bdist.world_size() > 1
    main = (not dp) or bdist.rank() == 0
    hparams['session_dir'], sess_ids = get_session_dir(
        hparams, session_source=hparams.get('all_source', 'save'))
    hparams['expt_dir'] = get_expt_dir(hparams)
    version = None
    failure = None
    if main:
        try:
            if not os.path.isdir(hparams['session_dir']):
                os.makedirs(hparams['session_dir'])
                export_session_info_to_csv(hparams['session_dir'], sess_ids)
            os.makedirs(hparams['expt_dir'], exist_ok=True)
            if not experiment_exists(hparams):
                exp = Experiment(name=hparams['experiment_name'], debug=False,
                                 save_dir=os.path.dirname(hparams['expt_dir']))
                exp.save()
                version = exp.version
        except Exception as err:                     # noqa: BLE001 (re-raised below, on every rank)
            if not dp:
                raise
            failure = err
    if dp:
        # rank 0's failure travels with the version: the other ranks raise at once instead of
        # sitting in the broadcast until the collective timeout
        version, failed = bdist.broadcast_object(
            (version, None if failure is None else '%s: %s' % (type(failure).__name__, failure)),
            src=0)
        if failure is not None:
            raise failure
        if failed is not None:
            raise RuntimeError('create_experiment failed on rank 0 (%s)' % failed)
        if version is not None and not main:
            exp = Experiment(name=hparams['experiment_name'], debug=True, version=version,
                             save_dir=os.path.dirname(hparams['expt_dir']))
    if version is None:
        return None, None, None
    hparams['version'] = exp.version
    return hparams, sess_ids, exp


create_tt_experiment = create_experiment      # the reference's name (test-tube is not used here)


def get_best_model_version(expt_dir, measure='val_loss', best_def='min', n_best=1):
    """Version numbers of the ``n_best`` finished fits by the extreme of ``measure`` in their
    ``metrics.csv``, best first."""
    import pandas as pd
    rows = []
    for version in get_subdirs(expt_dir):
        meta_file = os.path.join(expt_dir, version, 'meta_tags.pkl')
        if not os.path.exists(meta_file):
            continue
        with open(meta_file, 'rb') as f:
            if not pickle.load(f)['training_completed']:
                continue
        metric = pd.read_csv(os.path.join(expt_dir, version, 'metrics.csv'))[measure]
        rows.append({'loss': metric.min() if best_def == 'min' else metric.max(),
                     'version': version})
    table = pd.DataFrame(rows)
    if n_best == 1:
        pick = table['loss'].idxmin() if best_def == 'min' else table['loss'].idxmax()
        best = [table['version'][pick]]
    else:
        if best_def != 'min':
            raise NotImplementedError
        best = list(table['version'][table['loss'].nsmallest(n_best, 'all').index])
        if len(best) != n_best:
            print('More versions than specified due to same validation loss')
    return [int(v.split('_')[-1]) for v in best]


def get_best_model_and_data(hparams, Model=None, load_data=True, version='best',
                            data_kwargs=None):
    """Rebuild a fitted model (and its data generator) from its version directory.

    ``version``: 'best' (lowest validation loss), None (the version matching ``hparams``), an
    int, or 'version_K'.  The stored ``meta_tags.pkl`` defines the model; paths and device come
    from the caller's ``hparams``.
    """
    import torch
    from behavenet_amd.data.utils import get_data_generator_inputs
    from behavenet_amd.data.data_generator import ConcatSessionsGenerator

    hparams['session_dir'], sess_ids = get_session_dir(
        hparams, session_source=hparams.get('all_source', 'save'))
    expt_dir = get_expt_dir(hparams)
    if version == 'best':
        version_name = 'version_%i' % get_best_model_version(expt_dir)[0]
    elif version is None:
        _, found = experiment_exists(hparams, which_version=True)
        version_name = 'version_{}'.format(found)
    elif isinstance(version, str) and version[0] == 'v':
        version_name = version
    else:
        version_name = 'version_{}'.format(version)
    version_dir = os.path.join(expt_dir, version_name)
    arch_file = os.path.join(version_dir, 'meta_tags.pkl')
    model_file = os.path.join(version_dir, 'best_val_model.pt')
    print('Loading model defined in %s' % arch_file)
    with open(arch_file, 'rb') as f:
        hp = pickle.load(f)
    hp['data_dir'] = hparams['data_dir']
    hp['session_dir'] = hparams['session_dir']
    hp['expt_dir'] = expt_dir
    hp['use_output_mask'] = hparams.get('use_output_mask', False)
    hp['use_label_mask'] = hparams.get('use_label_mask', False)
    hp['device'] = hparams.get('device', 'cuda')

    hp, signals, transforms, paths = get_data_generator_inputs(hp, sess_ids)
    data_generator = None
    if load_data:
        data_generator = ConcatSessionsGenerator(
            hp['data_dir'], sess_ids, signals_list=signals, transforms_list=transforms,
            paths_list=paths, device=hp['device'], as_numpy=hp['as_numpy'],
            batch_load=hp['batch_load'], rng_seed=hp['rng_seed_data'],
            train_frac=hp['train_frac'], **(data_kwargs or {}))

    if Model is None:
        import importlib
        from behavenet_amd.fitting.ae_grid_search import MODEL_CLASSES
        key = 'conv-decoder' if hparams['model_class'] == 'labels-images' else \
            hparams['model_class']
        if key not in MODEL_CLASSES:
            _out_of_scope(hparams['model_class'])
        module, name = MODEL_CLASSES[key]
        Model = getattr(importlib.import_module(module), name)
    model = Model(hp)
    model.version = int(version_name.split('_')[1])
    model.load_state_dict(torch.load(model_file, map_location=lambda storage, loc: storage))
    model.to(hp['device'])
    model.eval()
    return model, data_generator


def _clean_tt_dir(hparams):
    """Remove the sub-directories a logger may have left in the version directory."""
    version_dir = os.path.join(hparams['expt_dir'], 'version_%i' % hparams['version'])
    try:
        subdirs = get_subdirs(version_dir)
    except StopIteration:
        return
    for sub in subdirs:
        shutil.rmtree(os.path.join(version_dir, sub))


def _print_hparams(hparams):
    """Print the hparams grouped by the config file they came from."""
    from behavenet_amd.fitting.hyperparam_utils import load_config_json
    for name in ('data', 'compute', 'training', 'model'):
        path = hparams.get('%s_config' % name)
        if not path:
            continue
        print('\n%s CONFIG:' % name.upper())
        for key in load_config_json(path).keys():
            print('    {}: {}'.format(key, hparams.get(key)))
    print('')


def get_lab_example(hparams, lab, expt):
    """Update ``hparams`` in place with the dataset's parameters, ``~/.behavenet/<lab>_<expt>_params.json``
    (ref fitting/utils.py:780-803: frame size, channels, neural bin size ... of a dataset the user has registered)."""
    import json
    from behavenet_amd import get_params_dir
    with open(os.path.join(get_params_dir(), '%s_%s_params.json' % (lab, expt)), 'r') as f:
        hparams.update(json.load(f))
