"""Config loading and hyperparameter grids for the autoencoder path.

Host-side mirror of the reference ``behavenet/fitting/hyperparam_utils.py:12-122`` (the part that
``ae_grid_search.py`` uses): the four json configs ``--data_config --model_config
--training_config --compute_config`` are merged into ONE flat namespace, every json value that
is a *list* becomes a grid axis, and ``architecture_params`` (the layer planner's dict for the
requested latent count, ``ae_model_architecture_generator.load_handcrafted_arches``) is added as
one more axis.  ``trials()`` expands the axes into one hparams namespace per grid point.

The reference builds this on the third-party ``test_tube.HyperOptArgumentParser`` (not
installed here, and tied to its SLURM launcher, which is out of scope: SURVEY.md section 2).
:class:`GridArgumentParser` is a small stand-in with the members the reference touches:
``add_argument``, ``opt_list(..., options=, tunable=)``, ``opt_args[name].opt_values/.tunable``,
``parse_known_args``, ``parse_args`` and ``parsed_args``; the namespace it returns has
``trials(n)``.  json files may carry ``#`` / ``//`` comments like the reference's
(``commentjson``).
"""

import argparse
import copy
import itertools
import json
import os
import sys

__all__ = [
    'load_config_json', 'strip_json_comments', 'GridArgumentParser', 'get_all_params',
    'add_to_parser', 'add_dependent_params', 'get_user_dir', 'AE_MODEL_CLASSES']

# model classes whose hparams get the conv architecture axis (ref hyperparam_utils.py:65-73)
AE_MODEL_CLASSES = ('ae', 'vae', 'beta-tcvae', 'cond-vae', 'cond-ae', 'cond-ae-msp', 'ps-vae',
                    'msps-vae', 'labels-images')


# ------------------------------------------------------------------------------------------
# comment-json
# ------------------------------------------------------------------------------------------
def strip_json_comments(text):
    """Remove ``# ...`` and ``// ...`` comments that are not inside a json string."""
    out = []
    for line in text.splitlines():
        in_str = False
        esc = False
        cut = len(line)
        for i, ch in enumerate(line):
            if in_str:
                if esc:
                    esc = False
                elif ch == '\\':
                    esc = True
                elif ch == '"':
                    in_str = False
                continue
            if ch == '"':
                in_str = True
            elif ch == '#' or (ch == '/' and line[i:i + 2] == '//'):
                cut = i
                break
        out.append(line[:cut])
    return '\n'.join(out)


def load_config_json(path):
    """Load one of the reference's commented json config files into a dict."""
    with open(path, 'r') as f:
        return json.loads(strip_json_comments(f.read()))


def get_user_dir(kind):
    """'data' | 'save' | 'figs' directory (ref behavenet/__init__.py:11-36): from
    ``~/.behavenet/directories.json`` if it exists, else ``~/.behavenet/<kind>``; the environment
    variables BEHAVENET_DATA_DIR / BEHAVENET_SAVE_DIR / BEHAVENET_FIGS_DIR take precedence."""
    env = os.environ.get('BEHAVENET_%s_DIR' % kind.upper())
    if env:
        return env
    base = os.path.join(os.path.expanduser('~'), '.behavenet')
    dirs_file = os.path.join(base, 'directories.json')
    if os.path.exists(dirs_file):
        with open(dirs_file, 'r') as f:
            return json.load(f)['%s_dir' % kind]
    return os.path.join(base, kind)


# ------------------------------------------------------------------------------------------
# grid parser
# ------------------------------------------------------------------------------------------
class _OptArg(object):
    """One list-valued option: a grid axis if ``tunable``."""

    def __init__(self, name, options, tunable):
        self.name = name
        self.opt_values = options
        self.tunable = tunable


class GridNamespace(argparse.Namespace):
    """Parsed hyperparameters; ``trials()`` enumerates the grid."""

    def trials(self, num=None):
        """One namespace per grid point (cartesian product of the tunable axes, first axis
        slowest), at most ``num`` of them."""
        axes = getattr(self, '_grid_axes', [])
        names = [a.name for a in axes]
        combos = itertools.product(*[list(a.opt_values) for a in axes]) if axes else [()]
        out = []
        for combo in combos:
            ns = GridNamespace(**{k: copy.deepcopy(v) for k, v in vars(self).items()
                                  if k != '_grid_axes'})
            for k, v in zip(names, combo):
                setattr(ns, k, copy.deepcopy(v))
            out.append(ns)
            if num is not None and len(out) >= num:
                break
        return out

    def as_dict(self):
        return {k: v for k, v in vars(self).items() if k != '_grid_axes'}


class GridArgumentParser(argparse.ArgumentParser):
    """argparse + list-valued options that span a grid (``strategy='grid_search'`` only)."""

    def __init__(self, strategy='grid_search', **kwargs):
        if strategy != 'grid_search':
            raise NotImplementedError('only strategy="grid_search" is implemented')
        kwargs.setdefault('add_help', False)
        super().__init__(**kwargs)
        self.strategy = strategy
        self.opt_args = {}
        self.parsed_args = None

    def opt_list(self, name, options=None, tunable=False, **kwargs):
        """Declare ``name`` with a list of candidate values.  On the command line the option
        still takes a single value; unless given there it parses to None and, if ``tunable``,
        becomes a grid axis."""
        kwargs.setdefault('default', None)
        self.add_argument(name, **kwargs)
        self.opt_args[name] = _OptArg(name.lstrip('-'), list(options) if options is not None
                                      else [], tunable)

    def _finish(self, ns):
        grid = GridNamespace(**vars(ns))
        axes = []
        for opt in self.opt_args.values():
            given = getattr(grid, opt.name, None)
            if given is None and opt.tunable and len(opt.opt_values) > 0:
                axes.append(opt)
        grid._grid_axes = axes
        return grid

    def parse_known_args(self, args=None, namespace=None):
        ns, extra = super().parse_known_args(args, namespace)
        return self._finish(ns), extra

    def parse_args(self, args=None, namespace=None):
        ns, extra = super().parse_known_args(args, namespace)
        if extra:
            self.error('unrecognized arguments: %s' % ' '.join(extra))
        grid = self._finish(ns)
        self.parsed_args = grid.as_dict()
        return grid


# ------------------------------------------------------------------------------------------
# the reference's entry points
# ------------------------------------------------------------------------------------------
def add_to_parser(parser, arg_name, value):
    """json key/value -> parser argument (ref :52-59): lists become grid axes; ``n_ae_latents``
    is parked under ``n_latents`` (as a string) until the architecture is chosen."""
    if arg_name == 'n_ae_latents':
        parser.add_argument('--n_latents', default=str(value))
    elif isinstance(value, list):
        parser.opt_list('--' + arg_name, options=value, tunable=True)
    else:
        parser.add_argument('--' + arg_name, default=value)


def add_dependent_params(parser, namespace):
    """Arguments derived from the json values (ref :62-122).  For the conv autoencoder classes:
    ``max_latents = 64`` and the ``architecture_params`` axis, one planned architecture per
    requested latent count (each carries its own ``n_ae_latents``)."""
    from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arches
    model_class = namespace.model_class
    if model_class in AE_MODEL_CLASSES:
        if namespace.model_type == 'conv':
            parser.add_argument('--max_latents', default=64)
            arch_dicts = load_handcrafted_arches(
                [namespace.n_input_channels, namespace.y_pixels, namespace.x_pixels],
                namespace.n_latents, namespace.ae_arch_json, check_memory=False,
                batch_size=namespace.approx_batch_size, mem_limit_gb=namespace.mem_limit_gb)
            parser.opt_list('--architecture_params', options=arch_dicts, tunable=True)
        elif namespace.model_type == 'linear':
            parser.add_argument('--n_ae_latents', default=namespace.n_latents, type=int)
        else:
            raise ValueError('%s is not a valid model type' % namespace.model_type)
    else:
        if getattr(namespace, 'n_latents', False):
            parser.add_argument('--n_ae_latents', default=namespace.n_latents, type=int)
    if model_class.find('neural') > -1 and getattr(namespace, 'subsample_method', 'none') != 'none':
        raise NotImplementedError(
            'neural decoders (subsample_idxs) are outside the MI355X hot path (SURVEY.md s2)')


def get_all_params(search_type='grid_search', args=None):
    """Parse ``--data_config A --model_config B --training_config C --compute_config D`` into a
    grid namespace (ref :12-49).  Exactly these eight command-line tokens are accepted."""
    argv = list(args) if args is not None else list(sys.argv[1:])
    if len(argv) != 8:
        raise ValueError('No command line arguments allowed other than config file names')
    parser = GridArgumentParser(strategy=search_type)
    for name in ('data_config', 'model_config', 'training_config', 'compute_config'):
        parser.add_argument('--' + name, type=str)
    namespace, _ = parser.parse_known_args(argv)
    for config in (namespace.data_config, namespace.model_config, namespace.training_config,
                   namespace.compute_config):
        for key, value in load_config_json(config).items():
            add_to_parser(parser, key, value)
    parser.add_argument('--save_dir', default=get_user_dir('save'), type=str)
    parser.add_argument('--data_dir', default=get_user_dir('data'), type=str)
    namespace, _ = parser.parse_known_args(argv)
    add_dependent_params(parser, namespace)
    return parser.parse_args(argv)


def trial_hparams(namespace):
    """Grid point -> the flat hparams dict ``ae_grid_search.main`` works with: the outer values
    win over the architecture dict's (ref ae_grid_search.py:25-27), and ``n_ae_latents`` comes
    from the architecture."""
    hp = namespace.as_dict() if hasattr(namespace, 'as_dict') else dict(vars(namespace))
    if hp.get('model_type') == 'conv' and isinstance(hp.get('architecture_params'), dict):
        hp = {**hp['architecture_params'], **hp}
    return hp
