import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
lib = ctypes.CDLL(_hip.lib_path())
lib.bn_debug_probe_fill2.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
st = torch.cuda.current_stream().cuda_stream
n = 105 * 1024 * 1024 // 4
out = torch.empty(n, device='cuda')
for mode in range(4):
    for blocks in (2048, 8192, 25600):
        ms = t(lambda: lib.bn_debug_probe_fill2(out.data_ptr(), n, blocks, mode, st))
        print('mode %d blocks %5d: %.1f us  %.2f TB/s' % (mode, blocks, ms * 1e3, n * 4 / ms / 1e9))
