import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _poisoned_lds(request):
    """Every GPU test starts with NaNs in the CUs' LDS: a kernel that reads LDS it did not write
    (a padded tap with a zero weight, a halo it skipped) then fails instead of passing on whatever
    finite values the previous kernel left there."""
    if 'gpu' not in request.keywords or not _has_gpu():
        yield
        return
    from tests import debug_lib
    debug_lib.poison_lds('cuda')
    yield
