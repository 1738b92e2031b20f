"""ctypes binding of libbehavenet_hip.so (include/behavenet_hip.h).

This module is the only place where Python touches the C ABI.  It hands over raw device
pointers (``tensor.data_ptr()``) and the current HIP stream; PyTorch is used for device memory
and streams only.  There is NO fallback: if the library is missing, or a tensor is not a
contiguous fp32 device tensor, the call raises.
"""

import ctypes
import os

import torch

_LIB_NAME = 'libbehavenet_hip.so'
# BN_HIP_LIB: an alternative build of the same ABI (csrc `make tuning`: experiment hooks compiled in)
_LIB_PATH = os.environ.get('BN_HIP_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)
_lib = None

ACT_NONE, ACT_LRELU, ACT_SIGMOID = 0, 1, 2

PROF_NONE, PROF_CONV_FWD, PROF_CONV_BWD_D, PROF_CONV_BWD_W = 0, 1, 2, 3
PROF_CONVT_FWD, PROF_CONVT_BWD_D, PROF_CONVT_BWD_W, PROF_ADAM = 4, 5, 6, 7
PROF_LINEAR_FWD, PROF_LINEAR_BWD = 8, 9

_c_int, _c_float, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

_CONV_GEOM = [_c_int] * 12
_ACT_WS = [_c_int, _c_float, _c_void_p, _c_size_t, _c_void_p]   # act, slope, ws, ws_bytes, stream

OP_CONV_FWD, OP_CONV_BWD_D, OP_CONV_BWD_W = 1, 2, 3
OP_CONVT_FWD, OP_CONVT_BWD_D, OP_CONVT_BWD_W = 4, 5, 6

# name -> (restype, argtypes); mirrors include/behavenet_hip.h one to one
SIGNATURES = {
    'bn_version': (_c_int, []),
    'bn_build_arch': (ctypes.c_char_p, []),
    'bn_error_string': (ctypes.c_char_p, [_c_int]),
    'bn_set_force_generic': (_c_int, [_c_int]),
    'bn_set_bigk1_block_bytes': (_c_size_t, [_c_size_t]),
    'bn_conv_ws_bytes': (_c_size_t, [_c_int] + _CONV_GEOM),
    'bn_conv_taps_bytes': (_c_size_t, [_c_int] + _CONV_GEOM),
    'bn_conv_taps_pad': (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    'bn_conv_taps_hint': (_c_int, [_c_void_p, _c_void_p]),
    'bn_conv2d_fwd': (_c_int, [_c_void_p] * 4 + _CONV_GEOM + _ACT_WS),
    'bn_conv2d_fwd_u8_ws_bytes': (_c_size_t, _CONV_GEOM + [_c_int]),
    'bn_conv2d_fwd_u8': (_c_int, [_c_void_p] * 4 + _CONV_GEOM + _ACT_WS),
    'bn_conv2d_bwd_data': (_c_int, [_c_void_p] * 4 + _CONV_GEOM + _ACT_WS),
    'bn_conv2d_bwd_weight': (
        _c_int, [_c_void_p] * 4 + _CONV_GEOM + [_c_int, _c_void_p, _c_size_t, _c_void_p]),
    'bn_convT2d_fwd': (_c_int, [_c_void_p] * 4 + _CONV_GEOM + _ACT_WS),
    'bn_convT2d_bwd_data': (_c_int, [_c_void_p] * 4 + _CONV_GEOM + _ACT_WS),
    'bn_convT2d_bwd_weight': (
        _c_int, [_c_void_p] * 4 + _CONV_GEOM + [_c_int, _c_void_p, _c_size_t, _c_void_p]),
    'bn_batchnorm_ws_bytes': (_c_size_t, [_c_int, _c_int]),
    'bn_batchnorm_stats': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p, _c_size_t, _c_void_p]),
    'bn_batchnorm_finalize': (_c_int, [_c_void_p] * 5 + [_c_int] + [_c_float] * 3 + [_c_void_p]),
    'bn_batchnorm_act_fwd': (_c_int, [_c_void_p] * 6 + [_c_int] * 4 + [_c_float, _c_void_p]),
    'bn_batchnorm_train_fwd_chunks': (
        _c_int, [_c_void_p] * 11 + [_c_int] * 3 + [_c_float, _c_int, _c_float, _c_void_p, _c_size_t,
                                                  _c_void_p]),
    'bn_batchnorm_act_bwd_chunks': (
        _c_int, [_c_void_p] * 10 + [_c_int, _c_void_p] + [_c_int] * 4 + [_c_float, _c_void_p, _c_size_t,
                                                                        _c_void_p]),
    'bn_batchnorm_act_bwd': (
        _c_int, [_c_void_p] * 9 + [_c_int] * 6 + [_c_float, _c_void_p, _c_size_t, _c_void_p]),
    'bn_batchnorm_moment': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p, _c_size_t, _c_void_p]),
    'bn_batchnorm_bwd_reduce': (
        _c_int, [_c_void_p] * 7 + [_c_int] * 4 + [_c_float, _c_void_p, _c_size_t, _c_void_p]),
    'bn_batchnorm_bwd_apply': (
        _c_int, [_c_void_p] * 9 + [_c_int] * 3 + [_c_float, _c_int, _c_float, _c_void_p]),
    'bn_maxpool2d_fwd': (_c_int, [_c_void_p] * 3 + [_c_int] * 9 + [_c_void_p]),
    'bn_maxpool2d_bwd': (_c_int, [_c_void_p] * 3 + [_c_int] * 9 + [_c_void_p]),
    'bn_maxpool2d_act_fwd': (_c_int, [_c_void_p] * 3 + [_c_int] * 4 + [ctypes.c_float, _c_void_p]),
    'bn_conv2d_pool2_act_fwd': (_c_int, [_c_void_p] * 5 + _CONV_GEOM + [_c_int, _c_float, _c_void_p]),
    'bn_conv2d_pool2_act_ok': (_c_int, _CONV_GEOM),
    'bn_conv2d_pool2_bwd_weight_ws_bytes': (_c_size_t, _CONV_GEOM),
    'bn_conv2d_pool2_bwd_weight': (
        _c_int, [_c_void_p] * 6 + _CONV_GEOM + [_c_int, _c_float, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    'bn_maxpool2d_act_bwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 4 + [ctypes.c_float, _c_void_p]),
    'bn_maxunpool2d_fwd': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p]),
    'bn_maxunpool2d_fwd_k2': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p]),
    'bn_maxunpool2d_bwd': (_c_int, [_c_void_p] * 3 + [_c_int] * 3 + [_c_void_p]),
    'bn_act_fwd': (_c_int, [_c_void_p] * 2 + [_c_size_t, _c_int, _c_float, _c_void_p]),
    'bn_act_bwd': (_c_int, [_c_void_p] * 3 + [_c_size_t, _c_int, _c_float, _c_void_p]),
    'bn_linear_ws_bytes': (_c_size_t, [_c_int] * 3),
    'bn_linear_fwd': (_c_int, [_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p, _c_size_t, _c_void_p]),
    'bn_linear_bwd': (
        _c_int, [_c_void_p] * 5 + [_c_int, _c_float, _c_void_p, _c_void_p, _c_int] +
        [_c_int] * 3 + [_c_void_p, _c_size_t, _c_void_p]),
    'bn_sqerr_frame_sums': (_c_int, [_c_void_p] * 4 + [_c_int, _c_size_t, _c_void_p]),
    'bn_sqerr_bwd': (_c_int, [_c_void_p] * 4 + [_c_size_t, _c_float, _c_void_p, _c_void_p]),
    'bn_convT2d_fwd_sqerr_parts': (_c_int, _CONV_GEOM),
    'bn_convT2d_fwd_sqerr_ws_bytes': (_c_size_t, _CONV_GEOM + [_c_int]),
    'bn_convT2d_fwd_sqerr': (
        _c_int, [_c_void_p] * 8 + _CONV_GEOM + [_c_int, _c_float, _c_void_p, _c_size_t, _c_void_p]),
    'bn_scale_frames': (_c_int, [_c_void_p] * 4 + [_c_int, _c_size_t, _c_void_p]),
    'bn_reduce_sum': (_c_int, [_c_void_p] * 2 + [_c_size_t, _c_float, _c_void_p]),
    'bn_reparam_fwd': (_c_int, [_c_void_p] * 4 + [_c_size_t, _c_void_p]),
    'bn_kl_rows': (_c_int, [_c_void_p] * 3 + [_c_int, _c_int, _c_void_p]),
    'bn_reparam_bwd': (_c_int, [_c_void_p] * 4 + [_c_size_t, _c_void_p]),
    'bn_kl_bwd': (_c_int, [_c_void_p] * 4 + [_c_size_t, _c_float, _c_void_p, _c_void_p]),
    'bn_decomposed_kl_fwd': (_c_int, [_c_void_p] * 7 + [_c_int, _c_int, _c_void_p]),
    'bn_decomposed_kl_bwd': (_c_int, [_c_void_p] * 9 + [_c_int, _c_int, _c_void_p]),
    'bn_psvae_head_fwd': (_c_int, [_c_void_p] * 14 + [_c_int] * 3 + [_c_void_p]),
    'bn_psvae_head_combine': (
        _c_int, [_c_void_p] * 4 + [_c_int] + [_c_float] * 3 + [_c_int] + [_c_void_p] * 3),
    'bn_psvae_head_bwd': (
        _c_int, [_c_void_p] * 3 + [_c_int] + [_c_void_p] * 10 + [_c_float] + [_c_void_p] * 5 +
        [_c_int] * 4 + [_c_void_p]),
    'bn_adam_amsgrad_step': (
        _c_int, [_c_void_p] * 5 + [_c_size_t] + [_c_float] * 5 + [_c_int, _c_void_p]),
    'bn_u8_to_unit_float': (_c_int, [_c_void_p] * 2 + [_c_size_t, _c_void_p]),
    'bn_prof_select': (_c_int, [_c_int] * 3),
    'bn_prof_select_nth': (_c_int, [_c_int] * 4),
    'bn_prof_read': (_c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]),
    'bn_prof_read_main': (_c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]),
    'bn_prof_set_bracket': (_c_int, [_c_int]),
    'bn_prof_kernel_name': (ctypes.c_char_p, []),
    'bn_prof_dispatch_overhead_us': (ctypes.c_double, [_c_int, _c_void_p]),
}


class HipLibraryError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load(required=True):
    """Load the shared library (once).  Raises if it is missing and ``required``."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        if required:
            raise HipLibraryError(
                '%s not found next to the package (%s). Build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C behavenet_amd/csrc`. There is no CPU fallback.' % (_LIB_NAME, _LIB_PATH))
        return None
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load().bn_error_string(int(rc))
        raise HipLibraryError('%s failed: %s (code %d)' % (what, msg.decode() if msg else '?', rc))


def _ptr(t, name, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise HipLibraryError('%s: tensor is None' % name)
    if not t.is_cuda:
        raise HipLibraryError(
            '%s: expected a tensor on the GPU, got device %s (the HIP path has no CPU fallback)'
            % (name, t.device))
    if t.dtype != dtype:
        raise HipLibraryError('%s: expected dtype %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise HipLibraryError('%s: tensor must be contiguous' % name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------
# convolution roles.  geom = (N, C, H, W, K, R, S, stride, pad_t, pad_l, P, Q) for conv and
# (N, Ci, Hi, Wi, Co, R, S, stride, crop_t, crop_l, Ho, Wo) for convT -- the header's order.
# ------------------------------------------------------------------------------------------
_ws_cache = {}


def _arena(device, nbytes, tag=None):
    """Pointer to at least ``nbytes`` of scratch on ``device``: a grow-only arena per (device,
    stream, tag) -- kernels on different streams may run concurrently -- or, while a HIP graph is
    being recorded, a block of the GRAPH's memory pool for this call only (freed right away: the
    pool re-issues it within the recording in stream order and keeps it reserved for the replays).
    A recording's block must never enter the cache: an arena cached from one graph's pool outlives
    the pool, and the next recording (another pool) would be handed memory that went back to the
    driver with the first graph (round 5: "write access to a read-only page" in the third recording
    of a process)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(int(nbytes), dtype=torch.uint8, device=device).data_ptr()
    key = (device, torch.cuda.current_stream(device).cuda_stream) + ((tag,) if tag else ())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        with torch.cuda.stream(torch.cuda.current_stream(device)):
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf.data_ptr()


def _workspace(op, geom, device):
    """(pointer, nbytes) of the grow-only scratch arena of `device` sized for this call."""
    if torch.device(device).type != 'cuda':
        raise HipLibraryError(
            'expected tensors on the GPU, got device %s (the HIP path has no CPU fallback)' % device)
    nbytes = load().bn_conv_ws_bytes(op, *geom)
    if nbytes == 0:
        return None, 0
    return _arena(device, nbytes), nbytes


_taps_bytes = {}
_generic_epoch = 0      # set_force_generic() changes what the dispatch pads


def conv_taps_bytes(op, geom):
    """Bytes of the 5x5 copy of a small-kernel layer's taps if `op` pads them, else 0 (bn_conv_taps_bytes)."""
    key = (_generic_epoch, op) + tuple(geom)
    v = _taps_bytes.get(key)
    if v is None:
        v = _taps_bytes[key] = int(load().bn_conv_taps_bytes(op, *geom))
    return v


def conv_taps_pad(jobs, device):
    """jobs = [(op, geom, w)] -> [w5]: the 5x5 copies of the layers' taps, all in ONE launch (bn_conv_taps_pad).
    The copies are slices of one fresh buffer; pass them to the forward / data-gradient wrappers as ``w5=``."""
    n = len(jobs)
    if n == 0:
        return []
    sizes = [(conv_taps_bytes(op, geom) // 4 + 63) // 64 * 64 for op, geom, _ in jobs]
    if min(sizes) == 0:
        raise HipLibraryError('conv_taps_pad: a layer whose op does not pad its taps')
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    out, o = [], 0
    for sz in sizes:
        out.append(buf[o:o + sz])
        o += sz
    wp = (ctypes.c_void_p * n)(*[_ptr(w, 'w') for _, _, w in jobs])
    w5p = (ctypes.c_void_p * n)(*[t.data_ptr() for t in out])
    flat = []
    for op, geom, _ in jobs:
        flat.append(op)
        flat.extend(int(v) for v in geom)
    geoms = (ctypes.c_int * (13 * n))(*flat)
    _check(load().bn_conv_taps_pad(n, wp, w5p, geoms, _stream()), 'bn_conv_taps_pad')
    return out


def _taps_hint(w, w5):
    if w5 is not None:
        load().bn_conv_taps_hint(w.data_ptr(), w5.data_ptr())


def conv2d_fwd(x, w, b, geom, act, slope, w5=None):
    N, C, H, W, K, R, S, st, pt, pl, P, Q = geom
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
    ws, nb = _workspace(OP_CONV_FWD, geom, x.device)
    _taps_hint(w, w5)
    _check(load().bn_conv2d_fwd(
        _ptr(x, 'x'), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True), _ptr(y, 'y'), *geom,
        act, slope, ws, nb, _stream()), 'bn_conv2d_fwd')
    return y


def conv2d_fwd_u8(x_u8, w, b, geom, act, slope):
    """First encoder layer from uint8 frames (value / 255 fused into the patch load)."""
    N, C, H, W, K, R, S, st, pt, pl, P, Q = geom
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x_u8.device)
    lib = load()
    nbytes = lib.bn_conv2d_fwd_u8_ws_bytes(*geom, act)
    ws = None
    if nbytes:
        ws = _arena(x_u8.device, nbytes, 'fwd_u8')
    _check(lib.bn_conv2d_fwd_u8(
        _ptr(x_u8, 'x', dtype=torch.uint8), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True),
        _ptr(y, 'y'), *geom, act, slope, ws, nbytes, _stream()), 'bn_conv2d_fwd_u8')
    return y


def conv2d_bwd_data(dy, w, geom, dact_src, dact, slope, w5=None):
    N, C, H, W = geom[:4]
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    ws, nb = _workspace(OP_CONV_BWD_D, geom, dy.device)
    _taps_hint(w, w5)
    _check(load().bn_conv2d_bwd_data(
        _ptr(dy, 'dy'), _ptr(w, 'w'), _ptr(dx, 'dx'), _ptr(dact_src, 'dact_src', allow_none=True),
        *geom, dact, slope, ws, nb, _stream()), 'bn_conv2d_bwd_data')
    return dx


def conv2d_bwd_weight(x, dy, dw, db, geom, accumulate):
    ws, nb = _workspace(OP_CONV_BWD_W, geom, x.device)
    _check(load().bn_conv2d_bwd_weight(
        _ptr(x, 'x'), _ptr(dy, 'dy'), _ptr(dw, 'dw'), _ptr(db, 'db', allow_none=True), *geom,
        int(accumulate), ws, nb, _stream()), 'bn_conv2d_bwd_weight')


def convT2d_fwd(x, w, b, geom, act, slope, w5=None):
    N, Ci, Hi, Wi, Co, R, S, st, ct, cl, Ho, Wo = geom
    y = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device)
    ws, nb = _workspace(OP_CONVT_FWD, geom, x.device)
    _taps_hint(w, w5)
    _check(load().bn_convT2d_fwd(
        _ptr(x, 'x'), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True), _ptr(y, 'y'), *geom,
        act, slope, ws, nb, _stream()), 'bn_convT2d_fwd')
    return y


def convT2d_bwd_data(dy, w, geom, dact_src, dact, slope, w5=None):
    N, Ci, Hi, Wi = geom[:4]
    dx = torch.empty((N, Ci, Hi, Wi), dtype=torch.float32, device=dy.device)
    ws, nb = _workspace(OP_CONVT_BWD_D, geom, dy.device)
    _taps_hint(w, w5)
    _check(load().bn_convT2d_bwd_data(
        _ptr(dy, 'dy'), _ptr(w, 'w'), _ptr(dx, 'dx'), _ptr(dact_src, 'dact_src', allow_none=True),
        *geom, dact, slope, ws, nb, _stream()), 'bn_convT2d_bwd_data')
    return dx


def convT2d_bwd_weight(x, dy, dw, db, geom, accumulate):
    ws, nb = _workspace(OP_CONVT_BWD_W, geom, x.device)
    _check(lib_call('bn_convT2d_bwd_weight')(
        _ptr(x, 'x'), _ptr(dy, 'dy'), _ptr(dw, 'dw'), _ptr(db, 'db', allow_none=True), *geom,
        int(accumulate), ws, nb, _stream()), 'bn_convT2d_bwd_weight')


def set_force_generic(on):
    """Route convolutions through the shape-agnostic kernels (test hook); returns previous."""
    global _generic_epoch
    _generic_epoch += 1
    return bool(load().bn_set_force_generic(1 if on else 0))


def set_bigk1_block_bytes(nbytes):
    """Frame-block size of the shifted-copies path of stride-1 7x7 / 9x9 layers (test hook); returns previous."""
    return int(load().bn_set_bigk1_block_bytes(int(nbytes)))


def lib_call(name):
    return getattr(load(), name)


def _bn_ws(n, c, device):
    nbytes = load().bn_batchnorm_ws_bytes(n, c)
    return _arena(device, nbytes, 'bn'), nbytes


def batchnorm_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, act, slope, y=None):
    """-> (y, mean, invstd); updates the running statistics in place (train mode).  ``y``: where to
    write the result (a contiguous slice of a larger batch)."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    mean = torch.empty((c,), dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    invstd = torch.empty_like(mean)
    ws, nb = _bn_ws(n, c, x.device)
    _check(load().bn_batchnorm_stats(_ptr(x, 'x'), _ptr(mean, 'mean'), _ptr(var, 'var'), n, c, hw,
                                     ws, nb, _stream()), 'bn_batchnorm_stats')
    cnt = n * hw
    unbias = cnt / (cnt - 1.0) if cnt > 1 else 1.0
    _check(load().bn_batchnorm_finalize(
        _ptr(mean, 'mean'), _ptr(var, 'var'), _ptr(invstd, 'invstd'),
        _ptr(running_mean, 'running_mean', allow_none=True),
        _ptr(running_var, 'running_var', allow_none=True), c, eps, momentum, unbias, _stream()),
        'bn_batchnorm_finalize')
    if y is None:
        y = torch.empty_like(x)
    _check(load().bn_batchnorm_act_fwd(
        _ptr(x, 'x'), _ptr(mean, 'mean'), _ptr(invstd, 'invstd'),
        _ptr(gamma, 'gamma', allow_none=True), _ptr(beta, 'beta', allow_none=True), _ptr(y, 'y'),
        n, c, hw, act, slope, _stream()), 'bn_batchnorm_act_fwd')
    return y, mean, invstd


def batchnorm_train_fwd_chunks(x, gamma, beta, running_mean, running_var, factors, eps, act, slope,
                               bounds, num_batches_tracked=None):
    """Train-mode batch norm with statistics per chunk of frames (``bounds``: [(beg, end)] in order,
    ``factors``: the running-estimate factor of every chunk's update).  One library call.
    ``num_batches_tracked``: nn.BatchNorm2d's int64 counter, advanced by len(bounds) on the device.
    -> (y, mean (n_chunks, C), invstd (n_chunks, C))."""
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or
                                            not num_batches_tracked.is_cuda):
        raise TypeError('num_batches_tracked must be an int64 device tensor')
    import ctypes
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    k = len(bounds)
    mean = torch.empty((k, c), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    y = torch.empty_like(x)
    ws, nb = _bn_ws(max(e - b for b, e in bounds), c, x.device)
    flat = (ctypes.c_int * (2 * k))(*[v for be in bounds for v in be])
    fac = (ctypes.c_float * k)(*[float(f) for f in factors])
    _check(load().bn_batchnorm_train_fwd_chunks(
        _ptr(x, 'x'), _ptr(gamma, 'gamma', allow_none=True), _ptr(beta, 'beta', allow_none=True),
        _ptr(running_mean, 'running_mean', allow_none=True),
        _ptr(running_var, 'running_var', allow_none=True),
        num_batches_tracked.data_ptr() if num_batches_tracked is not None else None,
        _ptr(y, 'y'), _ptr(mean, 'mean'),
        _ptr(invstd, 'invstd'), ctypes.cast(flat, ctypes.c_void_p), ctypes.cast(fac, ctypes.c_void_p),
        k, c, hw, eps, act, slope, ws, nb, _stream()), 'bn_batchnorm_train_fwd_chunks')
    return y, mean, invstd


def batchnorm_bwd_chunks(x, y, dy, mean, invstd, gamma, dgamma, dbeta, accumulate, act, slope, bounds,
                         beta=None):
    """``y`` None (identity / LeakyReLU): the activation's sign is rebuilt from x through (gamma,
    ``beta``) -- the saved output is not read."""
    import ctypes
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    k = len(bounds)
    dx = torch.empty_like(x)
    ws, nb = _bn_ws(max(e - b for b, e in bounds), c, x.device)
    flat = (ctypes.c_int * (2 * k))(*[v for be in bounds for v in be])
    _check(load().bn_batchnorm_act_bwd_chunks(
        _ptr(x, 'x'), _ptr(y, 'y', allow_none=True), _ptr(dy, 'dy'), _ptr(mean, 'mean'),
        _ptr(invstd, 'invstd'), _ptr(gamma, 'gamma', allow_none=True),
        _ptr(beta, 'beta', allow_none=True), _ptr(dx, 'dx'),
        _ptr(dgamma, 'dgamma', allow_none=True), _ptr(dbeta, 'dbeta', allow_none=True),
        int(accumulate), ctypes.cast(flat, ctypes.c_void_p), k, c, hw, act, slope, ws, nb, _stream()),
        'bn_batchnorm_act_bwd_chunks')
    return dx


def batchnorm_eval_fwd(x, gamma, beta, running_mean, running_var, eps, act, slope):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    invstd = torch.empty((c,), dtype=torch.float32, device=x.device)
    _check(load().bn_batchnorm_finalize(
        _ptr(running_mean, 'running_mean'), _ptr(running_var, 'running_var'),
        _ptr(invstd, 'invstd'), None, None, c, eps, 0.0, 1.0, _stream()), 'bn_batchnorm_finalize')
    y = torch.empty_like(x)
    _check(load().bn_batchnorm_act_fwd(
        _ptr(x, 'x'), _ptr(running_mean, 'mean'), _ptr(invstd, 'invstd'),
        _ptr(gamma, 'gamma', allow_none=True), _ptr(beta, 'beta', allow_none=True), _ptr(y, 'y'),
        n, c, hw, act, slope, _stream()), 'bn_batchnorm_act_fwd')
    return y, invstd


def batchnorm_bwd(x, y, dy, mean, invstd, gamma, dgamma, dbeta, accumulate, batch_stats, act,
                  slope, dx=None):
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // (n * c)
    if dx is None:
        dx = torch.empty_like(x)
    ws, nb = _bn_ws(n, c, x.device)
    _check(load().bn_batchnorm_act_bwd(
        _ptr(x, 'x'), _ptr(y, 'y'), _ptr(dy, 'dy'), _ptr(mean, 'mean'), _ptr(invstd, 'invstd'),
        _ptr(gamma, 'gamma', allow_none=True), _ptr(dx, 'dx'),
        _ptr(dgamma, 'dgamma', allow_none=True), _ptr(dbeta, 'dbeta', allow_none=True),
        int(accumulate), int(batch_stats), n, c, hw, act, slope, ws, nb, _stream()),
        'bn_batchnorm_act_bwd')
    return dx


def batchnorm_sync_train_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, act, slope,
                             all_reduce):
    """Train-mode batch norm whose statistics are taken over the frames of ALL ranks:
    ``all_reduce(t)`` sums a small device tensor over ranks in place.  -> (y, mean, invstd,
    global count); x may hold zero frames on this rank."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // max(n * c, 1) if n else int(x.shape[2] * x.shape[3])
    dev = x.device
    s1 = torch.zeros((c + 1,), dtype=torch.float32, device=dev)     # channel sums + frame count
    if n:
        ws, nb = _bn_ws(n, c, dev)
        _check(load().bn_batchnorm_moment(_ptr(x, 'x'), None, _ptr(s1, 'sums'), n, c, hw, ws, nb,
                                          _stream()), 'bn_batchnorm_moment')
        s1[c] = float(n)
    all_reduce(s1)
    count = s1[c] * hw
    mean = (s1[:c] / count).contiguous()
    s2 = torch.zeros((c,), dtype=torch.float32, device=dev)
    if n:
        _check(load().bn_batchnorm_moment(_ptr(x, 'x'), _ptr(mean, 'mean'), _ptr(s2, 'sums'), n, c,
                                          hw, ws, nb, _stream()), 'bn_batchnorm_moment')
    all_reduce(s2)
    var = (s2 / count).contiguous()
    cnt = float(count.item())
    unbias = cnt / (cnt - 1.0) if cnt > 1 else 1.0
    invstd = torch.empty_like(mean)
    _check(load().bn_batchnorm_finalize(
        _ptr(mean, 'mean'), _ptr(var, 'var'), _ptr(invstd, 'invstd'),
        _ptr(running_mean, 'running_mean', allow_none=True),
        _ptr(running_var, 'running_var', allow_none=True), c, eps, momentum, unbias, _stream()),
        'bn_batchnorm_finalize')
    y = torch.empty_like(x)
    if n:
        _check(load().bn_batchnorm_act_fwd(
            _ptr(x, 'x'), _ptr(mean, 'mean'), _ptr(invstd, 'invstd'),
            _ptr(gamma, 'gamma', allow_none=True), _ptr(beta, 'beta', allow_none=True),
            _ptr(y, 'y'), n, c, hw, act, slope, _stream()), 'bn_batchnorm_act_fwd')
    return y, mean, invstd, cnt


def batchnorm_sync_bwd(x, y, dy, mean, invstd, gamma, count, act, slope, all_reduce):
    """-> (dx, this rank's sum_dz, sum_dzx): dx uses the sums of ALL ranks, the parameter
    gradients (dbeta = sum_dz, dgamma = sum_dzx) stay local -- they are summed with every other
    gradient by the all-reduce before the optimizer step."""
    n, c = x.shape[0], x.shape[1]
    hw = x.numel() // max(n * c, 1) if n else 1
    local = torch.zeros((2, c), dtype=torch.float32, device=x.device)
    if n:
        ws, nb = _bn_ws(n, c, x.device)
        _check(load().bn_batchnorm_bwd_reduce(
            _ptr(x, 'x'), _ptr(y, 'y'), _ptr(dy, 'dy'), _ptr(mean, 'mean'),
            _ptr(invstd, 'invstd'), _ptr(local[0], 'sum_dz'), _ptr(local[1], 'sum_dzx'), n, c, hw,
            act, slope, ws, nb, _stream()), 'bn_batchnorm_bwd_reduce')
    total = all_reduce(local.clone())
    dx = torch.empty_like(x)
    if n:
        _check(load().bn_batchnorm_bwd_apply(
            _ptr(x, 'x'), _ptr(y, 'y'), _ptr(dy, 'dy'), _ptr(mean, 'mean'),
            _ptr(invstd, 'invstd'), _ptr(gamma, 'gamma', allow_none=True),
            _ptr(total[0], 'sum_dz'), _ptr(total[1], 'sum_dzx'), _ptr(dx, 'dx'), n, c, hw,
            1.0 / float(count), act, slope, _stream()), 'bn_batchnorm_bwd_apply')
    return dx, local[0], local[1]


def act_fwd(x, act, slope):
    y = torch.empty_like(x)
    _check(load().bn_act_fwd(_ptr(x, 'x'), _ptr(y, 'y'), x.numel(), act, slope, _stream()),
           'bn_act_fwd')
    return y


def act_bwd(dy, y, act, slope, out=None):
    if out is None:
        out = torch.empty_like(dy)
    _check(load().bn_act_bwd(
        _ptr(dy, 'dy'), _ptr(y, 'y'), _ptr(out, 'dpre'), dy.numel(), act, slope, _stream()),
        'bn_act_bwd')
    return out


def _linear_ws(M, K, N, device):
    """(pointer, nbytes) of this stream's scratch arena, grown for the split reductions."""
    nbytes = load().bn_linear_ws_bytes(M, K, N)
    if nbytes == 0:
        return None, 0
    return _arena(device, nbytes, 'linear'), nbytes


def linear_fwd(x, w, b):
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws, nb = _linear_ws(M, K, N, x.device)
    _check(load().bn_linear_fwd(
        _ptr(x, 'x'), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True), _ptr(y, 'y'), M, K, N,
        ws, nb, _stream()), 'bn_linear_fwd')
    return y


def linear_bwd(x, w, dy, need_dx, dact_src, dact, slope, dw, db, accumulate):
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty((M, K), dtype=torch.float32, device=dy.device) if need_dx else None
    ws, nb = _linear_ws(M, K, N, dy.device) if need_dx else (None, 0)
    _check(load().bn_linear_bwd(
        _ptr(x, 'x', allow_none=True), _ptr(w, 'w'), _ptr(dy, 'dy'),
        _ptr(dx, 'dx', allow_none=True), _ptr(dact_src, 'dact_src', allow_none=True), dact, slope,
        _ptr(dw, 'dw', allow_none=True), _ptr(db, 'db', allow_none=True), int(accumulate),
        M, K, N, ws, nb, _stream()), 'bn_linear_bwd')
    return dx


def sqerr_frame_sums(pred, target, mask):
    N = pred.shape[0]
    D = pred.numel() // N
    out = torch.empty((N,), dtype=torch.float32, device=pred.device)
    _check(load().bn_sqerr_frame_sums(
        _ptr(pred, 'pred'), _ptr(target, 'target'), _ptr(mask, 'mask', allow_none=True),
        _ptr(out, 'frame_sums'), N, D, _stream()), 'bn_sqerr_frame_sums')
    return out


def sqerr_bwd(pred, target, mask, scale, gscale, out=None):
    dpred = torch.empty_like(pred) if out is None else out
    _check(load().bn_sqerr_bwd(
        _ptr(pred, 'pred'), _ptr(target, 'target'), _ptr(mask, 'mask', allow_none=True),
        _ptr(dpred, 'dpred'), pred.numel(), float(scale), _ptr(gscale, 'gscale', allow_none=True),
        _stream()), 'bn_sqerr_bwd')
    return dpred


def convT2d_fwd_sqerr(x, w, b, target, mask, geom, act, slope, want_xhat):
    """Last decoder layer + squared error in one pass -> (xhat | None, dpre, part (N, P)):
    ``part[n].sum()`` is frame n's masked squared error, ``dpre`` its gradient with respect to the
    layer's pre-activation (see include/behavenet_hip.h)."""
    N, Ci, Hi, Wi, Co, R, S, st, ct, cl, Ho, Wo = geom
    lib = load()
    P = lib.bn_convT2d_fwd_sqerr_parts(*geom)
    if P <= 0:
        raise HipLibraryError('bn_convT2d_fwd_sqerr_parts: bad geometry %s' % (geom,))
    xhat = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device) if want_xhat \
        else None
    dpre = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device)
    part = torch.empty((N, P), dtype=torch.float32, device=x.device)
    nbytes = lib.bn_convT2d_fwd_sqerr_ws_bytes(*geom, int(want_xhat))
    ws = None
    if nbytes:
        ws = _arena(x.device, nbytes, 'fwd_sqerr')
    _check(lib.bn_convT2d_fwd_sqerr(
        _ptr(x, 'x'), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True), _ptr(target, 'target'),
        _ptr(mask, 'mask', allow_none=True), _ptr(xhat, 'xhat', allow_none=True),
        _ptr(dpre, 'dpre'), _ptr(part, 'part'), *geom, act, slope, ws, nbytes, _stream()),
        'bn_convT2d_fwd_sqerr')
    return xhat, dpre, part


def scale_frames(t, frame_scale, group_scale=None, group_of_frame=None):
    """In place: t[n] *= frame_scale[n] * group_scale[group_of_frame[n]]."""
    N = t.shape[0]
    _check(load().bn_scale_frames(
        _ptr(t, 't'), _ptr(frame_scale, 'frame_scale'),
        _ptr(group_scale, 'group_scale', allow_none=True),
        _ptr(group_of_frame, 'group_of_frame', dtype=torch.int32, allow_none=True),
        N, t.numel() // N, _stream()), 'bn_scale_frames')
    return t


def reduce_sum(t, scale=1.0, out=None):
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=t.device)
    _check(load().bn_reduce_sum(_ptr(t, 'in'), _ptr(out, 'out'), t.numel(), float(scale),
                                _stream()), 'bn_reduce_sum')
    return out


def reparam_fwd(mu, logvar, eps):
    z = torch.empty_like(mu)
    _check(load().bn_reparam_fwd(
        _ptr(mu, 'mu'), _ptr(logvar, 'logvar'), _ptr(eps, 'eps'), _ptr(z, 'z'), mu.numel(),
        _stream()), 'bn_reparam_fwd')
    return z


def kl_rows(mu, logvar):
    N, D = mu.shape
    out = torch.empty((N,), dtype=torch.float32, device=mu.device)
    _check(load().bn_kl_rows(_ptr(mu, 'mu'), _ptr(logvar, 'logvar'), _ptr(out, 'kl_rows'), N, D,
                             _stream()), 'bn_kl_rows')
    return out


def decomposed_kl_fwd(z, mu, logvar, out3=None):
    """-> (out3, log_qz, lse): the three KL terms and what the backward pass needs."""
    N, D = z.shape
    if out3 is None:
        out3 = torch.empty((3,), dtype=torch.float32, device=z.device)
    log_qz = torch.empty((N,), dtype=torch.float32, device=z.device)
    lse = torch.empty((N, D), dtype=torch.float32, device=z.device)
    terms = torch.empty((3 * N,), dtype=torch.float32, device=z.device)
    _check(load().bn_decomposed_kl_fwd(
        _ptr(z, 'z'), _ptr(mu, 'mu'), _ptr(logvar, 'logvar'), _ptr(out3, 'out3'),
        _ptr(log_qz, 'log_qz'), _ptr(lse, 'lse'), _ptr(terms, 'terms'), N, D, _stream()),
        'bn_decomposed_kl_fwd')
    return out3, log_qz, lse


def decomposed_kl_bwd(z, mu, logvar, log_qz, lse, g3, out=None):
    N, D = z.shape
    dz, dmu, dlogvar = out if out is not None else (
        torch.empty_like(z), torch.empty_like(z), torch.empty_like(z))
    _check(load().bn_decomposed_kl_bwd(
        _ptr(z, 'z'), _ptr(mu, 'mu'), _ptr(logvar, 'logvar'), _ptr(log_qz, 'log_qz'),
        _ptr(lse, 'lse'), _ptr(g3, 'g3'), _ptr(dz, 'dz'), _ptr(dmu, 'dmu'),
        _ptr(dlogvar, 'dlogvar'), N, D, _stream()), 'bn_decomposed_kl_bwd')
    return dz, dmu, dlogvar


def psvae_head_fwd(y, w, logvar, eps, Dw, Db, labels, lmask):
    """-> (z, z_u, lv_u, yhat, row_sq, row_kl); see include/behavenet_hip.h."""
    N, L = y.shape
    U = logvar.shape[1] - L
    dev = y.device
    z = torch.empty((N, L + U), dtype=torch.float32, device=dev)
    z_u = torch.empty((N, U), dtype=torch.float32, device=dev)
    lv_u = torch.empty((N, U), dtype=torch.float32, device=dev)
    yhat = torch.empty((N, L), dtype=torch.float32, device=dev)
    rows = torch.empty((2, N), dtype=torch.float32, device=dev)
    _check(load().bn_psvae_head_fwd(
        _ptr(y, 'y'), _ptr(w, 'w', allow_none=True), _ptr(logvar, 'logvar'), _ptr(eps, 'eps'),
        _ptr(Dw, 'Dw'), _ptr(Db, 'Db', allow_none=True), _ptr(labels, 'labels'),
        _ptr(lmask, 'lmask', allow_none=True), _ptr(z, 'z'), _ptr(z_u, 'z_u'), _ptr(lv_u, 'lv_u'),
        _ptr(yhat, 'yhat'), _ptr(rows[0], 'row_sq'), _ptr(rows[1], 'row_kl'), N, L, U, _stream()),
        'bn_psvae_head_fwd')
    return z, z_u, lv_u, yhat, rows[0], rows[1]


def psvae_head_combine(row_sq, row_kl, dkl3, bounds_dev, n_chunks, alpha, kl, beta, L):
    """-> (T (n_chunks,), cols5 (n_chunks, 5))."""
    T = torch.empty((n_chunks,), dtype=torch.float32, device=row_sq.device)
    cols5 = torch.empty((n_chunks, 5), dtype=torch.float32, device=row_sq.device)
    _check(load().bn_psvae_head_combine(
        _ptr(row_sq, 'row_sq'), _ptr(row_kl, 'row_kl'), _ptr(dkl3, 'dkl3'),
        _ptr(bounds_dev, 'bounds', dtype=torch.int32), int(n_chunks), float(alpha), float(kl),
        float(beta), int(L), _ptr(T, 'T'), _ptr(cols5, 'cols5'), _stream()), 'bn_psvae_head_combine')
    return T, cols5


def psvae_head_bwd(dz, gT, bounds_dev, n_chunks, y, logvar, eps, yhat, labels, lmask, Dw, gz_u,
                   gmu_u, glv_u, alpha, dDw, dDb, accumulate):
    """-> (dy, dw, dlogvar); dDw / dDb are written (or added to) in place."""
    N, L = y.shape
    U = logvar.shape[1] - L
    dy = torch.empty_like(y)
    dw = torch.empty((N, U), dtype=torch.float32, device=y.device)
    dlogvar = torch.empty_like(logvar)
    _check(load().bn_psvae_head_bwd(
        _ptr(dz, 'dz'), _ptr(gT, 'gT'), _ptr(bounds_dev, 'bounds', dtype=torch.int32),
        int(n_chunks), _ptr(y, 'y'), _ptr(logvar, 'logvar'), _ptr(eps, 'eps'), _ptr(yhat, 'yhat'),
        _ptr(labels, 'labels'), _ptr(lmask, 'lmask', allow_none=True), _ptr(Dw, 'Dw'),
        _ptr(gz_u, 'gz_u'), _ptr(gmu_u, 'gmu_u'), _ptr(glv_u, 'glv_u'), float(alpha),
        _ptr(dy, 'dy'), _ptr(dw, 'dw'), _ptr(dlogvar, 'dlogvar'),
        _ptr(dDw, 'dDw', allow_none=True), _ptr(dDb, 'dDb', allow_none=True),
        int(bool(accumulate)), N, L, U, _stream()), 'bn_psvae_head_bwd')
    return dy, dw, dlogvar


def reparam_bwd(dz, z, mu):
    dlogvar = torch.empty_like(mu)
    _check(load().bn_reparam_bwd(
        _ptr(dz, 'dz'), _ptr(z, 'z'), _ptr(mu, 'mu'), _ptr(dlogvar, 'dlogvar'), mu.numel(),
        _stream()), 'bn_reparam_bwd')
    return dlogvar


def kl_bwd(mu, logvar, scale, gscale, out=None):
    dmu, dlogvar = out if out is not None else (torch.empty_like(mu), torch.empty_like(mu))
    _check(load().bn_kl_bwd(
        _ptr(mu, 'mu'), _ptr(logvar, 'logvar'), _ptr(dmu, 'dmu'), _ptr(dlogvar, 'dlogvar'),
        mu.numel(), float(scale), _ptr(gscale, 'gscale', allow_none=True), _stream()),
        'bn_kl_bwd')
    return dmu, dlogvar


def adam_amsgrad_step(p, g, m, v, vmax, lr, beta1, beta2, eps, weight_decay, step):
    _check(load().bn_adam_amsgrad_step(
        _ptr(p, 'p'), _ptr(g, 'g'), _ptr(m, 'm'), _ptr(v, 'v'), _ptr(vmax, 'vmax'), p.numel(),
        lr, beta1, beta2, eps, weight_decay, int(step), _stream()), 'bn_adam_amsgrad_step')


def u8_to_unit_float(u8):
    out = torch.empty(u8.shape, dtype=torch.float32, device=u8.device)
    _check(load().bn_u8_to_unit_float(
        _ptr(u8, 'in', dtype=torch.uint8), _ptr(out, 'out'), u8.numel(), _stream()),
        'bn_u8_to_unit_float')
    return out


def prof_select(family, C=0, K=0, nth=None):
    """Time the calls of one family (and channel pair); ``nth``: only the nth matching call."""
    if nth is None:
        _check(load().bn_prof_select(family, C, K), 'bn_prof_select')
    else:
        _check(load().bn_prof_select_nth(family, C, K, int(nth)), 'bn_prof_select_nth')


def prof_read():
    ms, n = ctypes.c_double(0.0), ctypes.c_long(0)
    _check(load().bn_prof_read(ctypes.byref(ms), ctypes.byref(n)), 'bn_prof_read')
    name = load().bn_prof_kernel_name()
    return ms.value, n.value, (name.decode() if name else '')


def prof_set_bracket(on):
    return load().bn_prof_set_bracket(1 if on else 0)


def prof_read_main():
    """(ms, launches) of the main kernels of the profiled calls (dispatch-attached events)."""
    ms, n = ctypes.c_double(0.0), ctypes.c_long(0)
    _check(load().bn_prof_read_main(ctypes.byref(ms), ctypes.byref(n)), 'bn_prof_read_main')
    return ms.value, n.value


# ------------------------------------------------------------------------------------------
# max pooling with indices / unpooling ('max_pooling' architectures)
# ------------------------------------------------------------------------------------------
def maxpool2d_fwd(x, k, stride, pad, out_hw):
    """x (N,C,H,W) -> (y (N,C,Ho,Wo), idx int32 (N,C,Ho,Wo)); pad = (top, left)."""
    N, C, H, W = x.shape
    Ho, Wo = out_hw
    y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, C, Ho, Wo), dtype=torch.int32, device=x.device)
    _check(load().bn_maxpool2d_fwd(_ptr(x, 'x'), _ptr(y, 'y'), _ptr(idx, 'idx', torch.int32),
                                   N * C, H, W, Ho, Wo, k, stride, pad[0], pad[1], _stream()),
           'bn_maxpool2d_fwd')
    return y, idx


def maxpool2d_act_fwd(x, act, slope):
    """2x2 / stride-2 pooling + activation in one pass -> (y, idx) or None where the kernel does not apply."""
    N, C, H, W = x.shape
    if H % 2 or W % 4:
        return None
    y = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, C, H // 2, W // 2), dtype=torch.int32, device=x.device)
    rc = load().bn_maxpool2d_act_fwd(_ptr(x, 'x'), _ptr(y, 'y'), _ptr(idx, 'idx', torch.int32), N * C, H, W,
                                     int(act), float(slope), _stream())
    return (y, idx) if rc == 0 else None


def conv2d_pool_act_ok(geom):
    """Whether bn_conv2d_pool2_act_fwd serves this layer (host-side query)."""
    return bool(load().bn_conv2d_pool2_act_ok(*geom))


def conv2d_pool_bwd_weight_ws_bytes(geom):
    """Scratch of bn_conv2d_pool2_bwd_weight; 0 where the pooled-side weight gradient is not served."""
    return int(load().bn_conv2d_pool2_bwd_weight_ws_bytes(*geom))


def conv2d_pool_bwd_weight(x, dy, y, idx, dw, db, geom, act, slope, accumulate):
    """dw (+)=, db (+)= of a layer run by conv2d_pool_act_fwd from the pooled gradient `dy`, its saved output `y` and the
    winners `idx` (bn_conv2d_pool2_bwd_weight)."""
    nbytes = conv2d_pool_bwd_weight_ws_bytes(geom)
    if nbytes == 0:
        raise HipLibraryError('conv2d_pool_bwd_weight: geometry not served')
    ws = _arena(x.device, nbytes)
    _check(load().bn_conv2d_pool2_bwd_weight(
        _ptr(x, 'x'), _ptr(dy, 'dy'), _ptr(y, 'y'), _ptr(idx, 'idx', torch.int32), _ptr(dw, 'dw'),
        _ptr(db, 'db', allow_none=True), *geom, int(act), float(slope), int(accumulate), ws, nbytes, _stream()),
        'bn_conv2d_pool2_bwd_weight')


def conv2d_pool_act_fwd(x, w, b, geom, act, slope):
    """Conv2d + 2x2 / stride-2 max pooling + activation in one kernel -> (y, idx) or None where it is not served."""
    N, C, H, W, K, R, S, st, pt, pl, P, Q = geom
    if P % 2 or Q % 4 or x.dtype != torch.float32:
        return None
    y = torch.empty((N, K, P // 2, Q // 2), dtype=torch.float32, device=x.device)
    idx = torch.empty((N, K, P // 2, Q // 2), dtype=torch.int32, device=x.device)
    rc = load().bn_conv2d_pool2_act_fwd(_ptr(x, 'x'), _ptr(w, 'w'), _ptr(b, 'b', allow_none=True), _ptr(y, 'y'),
                                        _ptr(idx, 'idx', torch.int32), *geom, int(act), float(slope), _stream())
    if rc == -2:                      # BN_E_SHAPE: not served
        return None
    _check(rc, 'bn_conv2d_pool2_act_fwd')
    return y, idx


def maxpool2d_act_bwd(dy, y, idx, in_hw, act, slope):
    N, C, Ho, Wo = dy.shape
    H, W = in_hw
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    _check(load().bn_maxpool2d_act_bwd(_ptr(dy, 'dy'), _ptr(y, 'y'), _ptr(idx, 'idx', torch.int32), _ptr(dx, 'dx'),
                                       N * C, H, W, int(act), float(slope), _stream()), 'bn_maxpool2d_act_bwd')
    return dx


def maxpool2d_bwd(dy, idx, in_hw, k, stride, pad):
    N, C, Ho, Wo = dy.shape
    H, W = in_hw
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    _check(load().bn_maxpool2d_bwd(_ptr(dy, 'dy'), _ptr(idx, 'idx', torch.int32), _ptr(dx, 'dx'),
                                   N * C, H, W, Ho, Wo, k, stride, pad[0], pad[1], _stream()),
           'bn_maxpool2d_bwd')
    return dx


def maxunpool2d_fwd(x, idx, out_hw, own_window=False):
    """``own_window``: idx are the indices of the 2x2 / stride-2 pooling this layer undoes (each inside its own
    window): the one-pass kernel where the maps qualify."""
    N, C, Hi, Wi = x.shape
    Ho, Wo = out_hw
    y = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    if own_window and (Ho, Wo) == (2 * Hi, 2 * Wi) and Wi % 2 == 0:
        rc = load().bn_maxunpool2d_fwd_k2(_ptr(x, 'x'), _ptr(idx, 'idx', torch.int32), _ptr(y, 'y'), N * C, Hi, Wi,
                                          _stream())
        if rc == 0:
            return y
    _check(load().bn_maxunpool2d_fwd(_ptr(x, 'x'), _ptr(idx, 'idx', torch.int32), _ptr(y, 'y'),
                                     N * C, Hi * Wi, Ho * Wo, _stream()), 'bn_maxunpool2d_fwd')
    return y


def maxunpool2d_bwd(dy, idx):
    N, C, Ho, Wo = dy.shape
    dx = torch.empty(idx.shape, dtype=torch.float32, device=dy.device)
    _check(load().bn_maxunpool2d_bwd(_ptr(dy, 'dy'), _ptr(idx, 'idx', torch.int32), _ptr(dx, 'dx'),
                                     N * C, idx.shape[2] * idx.shape[3], Ho * Wo, _stream()),
           'bn_maxunpool2d_bwd')
    return dx
