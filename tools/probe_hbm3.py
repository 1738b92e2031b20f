import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
lib = ctypes.CDLL(_hip.lib_path())
lib.bn_debug_probe_fill3.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
st = torch.cuda.current_stream().cuda_stream
for nf in (56, 200):
    n = nf * 32 * 4096
    out = torch.empty(n, device='cuda')
    ms = t(lambda: lib.bn_debug_probe_fill3(out.data_ptr(), nf, st))
    print('k_down_c1 store pattern, %d frames: %.1f us  %.2f TB/s' % (nf, ms * 1e3, n * 4 / ms / 1e9))
