import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
lib = ctypes.CDLL(_hip.lib_path())
lib.bn_debug_probe_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
st = torch.cuda.current_stream().cuda_stream
for mb in (29, 105, 420):
    n = mb * 1024 * 1024 // 4
    out = torch.empty(n, device='cuda'); src = torch.rand(n, device='cuda')
    for blocks in (1024, 2048, 8192):
        ms = t(lambda: lib.bn_debug_probe_fill(out.data_ptr(), n, blocks, st))
        print('fill %4d MB blocks %5d: %.1f us  %.2f TB/s' % (mb, blocks, ms * 1e3, n * 4 / ms / 1e9))
    ms = t(lambda: out.copy_(src))
    print('copy %4d MB (r+w): %.1f us  %.2f TB/s' % (mb, ms * 1e3, 2 * n * 4 / ms / 1e9))
    ms = t(lambda: out.zero_())
    print('zero %4d MB: %.1f us  %.2f TB/s' % (mb, ms * 1e3, n * 4 / ms / 1e9))
