"""hparams -> data generator for the autoencoder classes.

Mirror of ``get_data_generator_inputs`` / ``build_data_generator`` of the reference
``behavenet/data/utils.py:15-125,342-394`` restricted to the model classes of the conv-AE path:
which signals a class reads from each session's ``data.hdf5`` (or its ``data.npz`` mirror, see
``trial_store``), with which transform, and the generator built from them.
"""

import os

__all__ = ['get_data_generator_inputs', 'build_data_generator']

_IMAGES_ONLY = ('ae', 'vae', 'beta-tcvae')
_IMAGES_AND_LABELS = ('cond-ae', 'cond-ae-msp', 'cond-vae', 'ps-vae', 'msps-vae', 'labels-images')


def get_data_generator_inputs(hparams, sess_ids, check_splits=True):
    """-> (hparams, signals_list, transforms_list, paths_list), one entry per session.

    images [+ masks if ``use_output_mask``] for ae / vae / beta-tcvae; images + labels
    [+ masks] [+ labels_masks if ``use_label_mask`` and the class is cond-ae-msp or ps-vae]
    [+ labels_sc through ``MakeOneHot2D`` if ``conditional_encoder``] for the label-aware
    classes.  Every signal of a session lives in ``data_dir/lab/expt/animal/session/data.hdf5``.
    """
    model_class = hparams['model_class']
    if model_class not in _IMAGES_ONLY + _IMAGES_AND_LABELS:
        raise NotImplementedError(
            'model class "%s" is outside the MI355X autoencoder path (SURVEY.md section 2)' %
            model_class)
    signals_list, transforms_list, paths_list = [], [], []
    for sess_id in sess_ids:
        data_file = os.path.join(hparams['data_dir'], sess_id['lab'], sess_id['expt'],
                                 sess_id['animal'], sess_id['session'], 'data.hdf5')
        signals, transforms = ['images'], [None]
        if model_class in _IMAGES_AND_LABELS:
            signals.append('labels')
            transforms.append(None)
        if hparams.get('use_output_mask', False):
            signals.append('masks')
            transforms.append(None)
        if model_class in _IMAGES_AND_LABELS:
            if hparams.get('use_label_mask', False) and model_class in ('cond-ae-msp', 'ps-vae'):
                signals.append('labels_masks')
                transforms.append(None)
            if hparams.get('conditional_encoder', False):
                from behavenet_amd.data.transforms import MakeOneHot2D
                signals.append('labels_sc')
                transforms.append(MakeOneHot2D(hparams['y_pixels'], hparams['x_pixels']))
        signals_list.append(signals)
        transforms_list.append(transforms)
        paths_list.append([data_file] * len(signals))
    return hparams, signals_list, transforms_list, paths_list


def build_data_generator(hparams, sess_ids, export_csv=True):
    """The :class:`ConcatSessionsGenerator` a fit uses (``trial_splits`` as 'train;val;test;gap';
    ``n_sessions_per_batch`` > 1 serves multi-session training batches); writes the session list
    next to the model so that dataset indices can be mapped back to sessions."""
    from behavenet_amd.data.data_generator import ConcatSessionsGenerator
    from behavenet_amd.fitting.utils import export_session_info_to_csv
    print('using data from following sessions:')
    for ids in sess_ids:
        print('%s' % os.path.join(
            hparams['save_dir'], ids['lab'], ids['expt'], ids['animal'], ids['session']))
    hparams, signals, transforms, paths = get_data_generator_inputs(hparams, sess_ids)
    trial_splits = None
    if hparams.get('trial_splits', None) is not None:
        trs = [int(tr) for tr in hparams['trial_splits'].split(';')]
        trial_splits = dict(zip(('train_tr', 'val_tr', 'test_tr', 'gap_tr'), trs))
    print('constructing data generator...', end='')
    data_generator = ConcatSessionsGenerator(
        hparams['data_dir'], sess_ids, signals_list=signals, transforms_list=transforms,
        paths_list=paths, device=hparams['device'], as_numpy=hparams['as_numpy'],
        batch_load=hparams['batch_load'], rng_seed=hparams['rng_seed_data'],
        trial_splits=trial_splits, train_frac=hparams['train_frac'],
        n_sessions_per_batch=hparams.get('n_sessions_per_batch', 1))
    if export_csv:
        export_session_info_to_csv(os.path.join(
            hparams['expt_dir'], 'version_%i' % hparams['version']), sess_ids)
    print('done')
    print(data_generator)
    return data_generator
