"""The C-ABI library loads and exports every symbol include/behavenet_hip.h declares
(no compute calls: this runs without a GPU)."""

import ctypes
import os
import re

from behavenet_amd import _hip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header='behavenet_hip.h'):
    with open(os.path.join(REPO, 'include', header)) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = re.findall(r'\b(bn_[a-zA-Z0-9_]+)\s*\(', text)
    return sorted(set(names))


def test_library_is_built():
    assert os.path.exists(_hip.lib_path()), \
        'libbehavenet_hip.so missing: run `python -c "import __graft_entry__ as g; g.build()"`'


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 26
    lib = ctypes.CDLL(_hip.lib_path())
    for n in names:
        assert hasattr(lib, n), 'header declares %s but the .so does not export it' % n
    # the ctypes table covers the header one to one
    assert sorted(_hip.SIGNATURES.keys()) == names


def test_info_calls():
    lib = _hip.load()
    assert lib.bn_version() == 1
    assert lib.bn_build_arch() == b'gfx950'
    assert b'BN_E_SHAPE' in lib.bn_error_string(-2)
    # argument errors are reported, not thrown
    assert lib.bn_conv2d_fwd(None, None, None, None, *([1] * 12), 0, 0.0, None, 0, None) == -1
    assert lib.bn_conv_ws_bytes(99, *([1] * 12)) == 0
    assert lib.bn_prof_select(99, 0, 0) == -1


def test_product_library_has_no_debug_symbols():
    """The probes / LDS poisoning live in the TEST-ONLY tests/native/libbn_debug.so."""
    lib = ctypes.CDLL(_hip.lib_path())
    for n in _declared_symbols('behavenet_hip_debug.h'):
        assert not hasattr(lib, n), 'product library exports the debug symbol %s' % n


def test_debug_library_matches_its_header():
    from tests import debug_lib
    names = _declared_symbols('behavenet_hip_debug.h')
    assert os.path.exists(debug_lib.lib_path()), 'run `make -C tests/native`'
    lib = ctypes.CDLL(debug_lib.lib_path())
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(debug_lib.SIGNATURES.keys()) == names
    debug_lib.load()
