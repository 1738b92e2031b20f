"""behavenet_amd: MI355X-native implementation of BehaveNet's conv-autoencoder hot path.

Python is glue (config, module containers, training loop); all arithmetic on the path runs in
the hand-written HIP kernels of ``libbehavenet_hip.so`` (see ``include/behavenet_hip.h``).
"""

import os as _os

# The step runs on up to six HIP streams (main, weight-gradient side stream, chunk stream, feed
# prefetch, collective launch + RCCL's own).  The HIP runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share a queue serialise, and a
# stream waiting on an event then blocks its queue-mate (measured: +0.4 ms per step once the
# collective streams exist).  Read by the runtime when it initialises, i.e. at the first HIP call.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

__version__ = '0.1.0'


# the reference's package-level helpers (behavenet/__init__.py:5-53): where the user's data / results / figures
# and the per-dataset parameter files live
def get_params_dir():
    """``~/.behavenet``: directories.json and the ``<lab>_<expt>_params.json`` files."""
    return _os.path.join(_os.path.expanduser('~'), '.behavenet')


def get_user_dir(type):                         # noqa: A002 (the reference's argument name)
    """'data' | 'save' | 'figs' directory (``fitting.hyperparam_utils.get_user_dir``)."""
    from behavenet_amd.fitting.hyperparam_utils import get_user_dir as _impl
    return _impl(type)


def make_dir_if_not_exists(save_file):
    """Create the directory a file is about to be written into."""
    folder = _os.path.dirname(save_file)
    if folder:
        _os.makedirs(folder, exist_ok=True)
