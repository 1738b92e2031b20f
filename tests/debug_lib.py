"""ctypes binding of the TEST-ONLY library tests/native/libbn_debug.so
(include/behavenet_hip_debug.h): LDS poisoning for the GPU tests and the hardware probes of
tools/.  The product package never loads it."""

import ctypes
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'libbn_debug.so')
_c_int, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p

# name -> argtypes (all return int); mirrors include/behavenet_hip_debug.h one to one
SIGNATURES = {
    'bn_debug_poison_lds': [_c_void_p, _c_void_p],
    'bn_debug_probe_mfma': [_c_void_p, _c_int, _c_int, _c_void_p],
    'bn_debug_probe_mfma_lds': [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    'bn_debug_probe_fill': [_c_void_p, _c_size_t, _c_int, _c_void_p],
    'bn_debug_probe_fill2': [_c_void_p, _c_size_t, _c_int, _c_int, _c_void_p],
    'bn_debug_probe_fill3': [_c_void_p, _c_int, _c_void_p],
    'bn_debug_probe_fill4': [_c_void_p, _c_int, _c_int, _c_int, _c_void_p],
    'bn_debug_probe_lds_dma': [_c_void_p, _c_void_p, _c_int, _c_void_p],
}
_lib = None


def lib_path():
    return _PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError('%s missing: run `make -C tests/native` (or __graft_entry__.build())'
                               % _PATH)
        lib = ctypes.CDLL(_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = _c_int
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def poison_lds(device='cuda'):
    """Leave NaNs in every CU's LDS (see the header); synchronises."""
    import torch
    sink = torch.zeros(1, device=device)
    with torch.cuda.device(sink.device):
        rc = load().bn_debug_poison_lds(sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
