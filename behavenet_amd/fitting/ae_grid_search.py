"""The reference's ``behavenet/fitting/ae_grid_search.py``: fit one autoencoder per grid point.

    python -m behavenet_amd.fitting.ae_grid_search --data_config D.json --model_config M.json \
        --training_config T.json --compute_config C.json

``main(hparams)`` is the reference's ``main`` (ref :20-118): merge the architecture dict, create
the experiment version (skipping grid points that were fitted already), build the data generator
and the model, export the hparams, ``fit``, mark the version complete.  ``build_model`` is the
``model_class`` string -> class dispatch with the reference's seeding, ``n_datasets`` /
``n_labels`` discovery, device move and pretrained-weight loading; ``fit_model`` chains it with
``fit`` for callers that bring their own generator / experiment object.

Grid execution (ref :150-198 hands the grid to test-tube's process pool / SLURM; not used here):
the grid points are run one after another in this process; under ``torchrun`` (WORLD_SIZE > 1)
rank r takes the points r, r + W, ... -- one model per GPU, no communication.
Data-parallel training of ONE model over several GPUs is a different mode
(``n_parallel_gpus`` > 1 with an initialised process group, see fitting/distributed.py).
"""

import os

import torch

__all__ = ['MODEL_CLASSES', 'NEEDS_LABELS', 'build_model', 'fit_model', 'main', 'run_grid']

MODEL_CLASSES = {
    'ae': ('behavenet_amd.models.aes', 'AE'),
    'vae': ('behavenet_amd.models.vaes', 'VAE'),
    'beta-tcvae': ('behavenet_amd.models.vaes', 'BetaTCVAE'),
    'ps-vae': ('behavenet_amd.models.vaes', 'PSVAE'),
    'msps-vae': ('behavenet_amd.models.vaes', 'MSPSVAE'),
    'cond-vae': ('behavenet_amd.models.vaes', 'ConditionalVAE'),
    'cond-ae': ('behavenet_amd.models.aes', 'ConditionalAE'),
    'cond-ae-msp': ('behavenet_amd.models.aes', 'AEMSP'),
    'conv-decoder': ('behavenet_amd.models.decoders', 'ConvDecoder'),   # decoder_grid_search.py
}
# classes whose constructor needs hparams['n_labels'] (ref ae_grid_search.py:52-55,68-84)
NEEDS_LABELS = ('ps-vae', 'msps-vae', 'cond-vae', 'cond-ae', 'cond-ae-msp', 'conv-decoder')


def _set_n_labels(data_generator, hparams):
    """Peek at one validation batch: labels are (1, n_frames, n_labels) (ref :52-55).  Like the
    reference this consumes one 'val' batch of the generator's iterator."""
    data, _ = data_generator.next_batch('val')
    hparams['n_labels'] = data['labels'].shape[2]


def build_model(hparams, data_generator=None, n_datasets=None):
    """Construct the model named by ``hparams['model_class']`` as the reference's main() does.

    Seeds torch with ``rng_seed_model`` (ref :59), records the RNG states the reference stores
    in hparams (``model_build_rng_seed``, ``training_rng_seed``), sets ``n_datasets``, discovers
    ``n_labels`` for the label-conditioned classes, moves the model to ``hparams['device']`` and
    loads ``pretrained_weights_path`` if given (ref :87-91).
    """
    import importlib
    from behavenet_amd.models.aes import load_pretrained_ae
    from behavenet_amd.models.base import CustomDataParallel

    model_class = hparams['model_class']
    if model_class not in MODEL_CLASSES:
        raise NotImplementedError(
            'The model class "%s" is not currently implemented' % model_class)
    torch.manual_seed(hparams['rng_seed_model'])
    hparams['model_build_rng_seed'] = torch.get_rng_state()
    if n_datasets is None:
        n_datasets = data_generator.n_datasets if data_generator is not None else 1
    hparams['n_datasets'] = n_datasets
    if model_class in NEEDS_LABELS:
        if data_generator is not None:
            _set_n_labels(data_generator, hparams)          # overwrites, as the reference does
        elif 'n_labels' not in hparams:
            raise ValueError('"%s" needs hparams["n_labels"] or a data generator' % model_class)
    module, name = MODEL_CLASSES[model_class]
    Model = getattr(importlib.import_module(module), name)
    model = Model(hparams)
    model.to(hparams['device'])
    if hasattr(model, 'encoding'):
        model = load_pretrained_ae(model, hparams)
    if hparams.get('n_parallel_gpus', 1) > 1:
        model = CustomDataParallel(model)
    hparams['training_rng_seed'] = torch.get_rng_state()
    return model


def fit_model(hparams, data_generator, exp):
    """build_model + fit, the body of the reference's main() (ref :57-118)."""
    from behavenet_amd.fitting.training import fit
    model = build_model(hparams, data_generator)
    model.version = exp.version
    hparams['training_completed'] = False
    fit(hparams, model, data_generator, exp, method='ae')
    hparams['training_completed'] = True
    return model


def main(hparams, *args):
    """Fit the model one grid point describes (ref ae_grid_search.py:20-118)."""
    from behavenet_amd.data.utils import build_data_generator
    from behavenet_amd.fitting.hyperparam_utils import trial_hparams
    from behavenet_amd.fitting.training import fit
    from behavenet_amd.fitting.utils import (
        _clean_tt_dir, _print_hparams, create_experiment, export_hparams)

    if not isinstance(hparams, dict):
        hparams = trial_hparams(hparams)
    elif hparams.get('model_type') == 'conv' and \
            isinstance(hparams.get('architecture_params'), dict):
        hparams = {**hparams['architecture_params'], **hparams}
    _print_hparams(hparams)
    if hparams['model_type'] == 'conv' and hparams['n_ae_latents'] > hparams['max_latents']:
        raise ValueError('Number of latents higher than max latents, architecture will not work')

    hparams, sess_ids, exp = create_experiment(hparams)
    if hparams is None:
        print('Experiment exists! Aborting fit')
        return None
    data_generator = build_data_generator(hparams, sess_ids)

    print('constructing model...', end='')
    model = build_model(hparams, data_generator, n_datasets=len(sess_ids))
    model.version = exp.version
    hparams['training_completed'] = False
    export_hparams(hparams, exp)
    print('done')
    print(model)

    fit(hparams, model, data_generator, exp, method='ae')

    hparams['training_completed'] = True
    export_hparams(hparams, exp)
    _clean_tt_dir(hparams)
    if hparams.get('export_train_plots', False):
        # plotting is outside the hot path; metrics.csv holds the curves
        print('training curves: %s' % os.path.join(
            hparams['expt_dir'], 'version_%i' % hparams['version'], 'metrics.csv'))
    return model


def run_grid(hyperparams, max_trials=None):
    """Run ``main`` over the grid of a parsed namespace; -> list of (hparams dict, model | None).
    With WORLD_SIZE > 1 in the environment each rank runs its own slice on its own GPU."""
    from behavenet_amd.fitting.hyperparam_utils import trial_hparams
    if getattr(hyperparams, 'device', None) == 'gpu':
        hyperparams.device = 'cuda'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    n = max_trials if max_trials is not None else getattr(hyperparams, 'tt_n_gpu_trials', None)
    results = []
    for i, trial in enumerate(hyperparams.trials(n)):
        if i % world != rank:
            continue
        hp = trial_hparams(trial)
        model = main(hp)
        # main() works on its own merged copy of the dict: report the one the model carries
        results.append((model.hparams if model is not None else hp, model))
    return results


if __name__ == '__main__':
    from behavenet_amd.fitting.hyperparam_utils import get_all_params
    run_grid(get_all_params('grid_search'))
