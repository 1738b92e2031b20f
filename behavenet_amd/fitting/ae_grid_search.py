"""Model construction and the fit entry point of the reference's ``ae_grid_search.main``
(ref behavenet/fitting/ae_grid_search.py:40-118), without the test-tube bookkeeping around it
(experiment folders, hyperparameter grid, SLURM: SURVEY.md section 2, out of scope).

``build_model`` is the ``model_class`` string -> class dispatch with the reference's seeding,
``n_datasets`` / ``n_labels`` discovery, device move and pretrained-weight loading;
``fit_model`` chains it with ``fit``.
"""

import torch

__all__ = ['MODEL_CLASSES', 'NEEDS_LABELS', 'build_model', 'fit_model']

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
