"""Write ceiling with and without Infinity-Cache residency: one 105 MB buffer vs a ring of 5."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import debug_lib
lib = debug_lib.load()
st = torch.cuda.current_stream().cuda_stream
n = 200 * 32 * 64 * 64
def t(bufs, mode, blocks, it=40):
    i = [0]
    def fn():
        b = bufs[i[0] % len(bufs)]; i[0] += 1
        lib.bn_debug_probe_fill2(b.data_ptr(), n, blocks, mode, st)
    for _ in range(len(bufs)): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
one = [torch.empty(n, device='cuda')]
ring = [torch.empty(n, device='cuda') for _ in range(5)]
for mode in (0, 1, 2, 3):
    for blocks in (2048, 8192):
        a = t(one, mode, blocks); b = t(ring, mode, blocks)
        print('mode %d blocks %5d: same buffer %.1f us %.2f TB/s | ring of 5 %.1f us %.2f TB/s' % (
            mode, blocks, a * 1e3, n * 4 / a / 1e9, b * 1e3, n * 4 / b / 1e9))
