"""Store contiguity per instruction for an enc.conv0-shaped output stream (256 frames x 32 ch x
64 x 64 floats = 134 MB), ring of 3 buffers: 4 x 256 B vs 2 x 512 B vs 1 x 1 KB per wave store."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import debug_lib
lib = debug_lib.load()
st = torch.cuda.current_stream().cuda_stream
nf = 256
bufs = [torch.empty(nf * 32 * 64 * 64, device='cuda') for _ in range(3)]
def t(mode, grid, it=30):
    i = [0]
    def fn():
        b = bufs[i[0] % 3]; i[0] += 1
        lib.bn_debug_probe_fill4(b.data_ptr(), nf, mode, grid, st)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for grid in (2048, 3072, 4096):
    print('grid %d: ' % grid + '  '.join('mode %d %.1f us (%.2f TB/s)' % (
        m, t(m, grid) * 1e3, nf * 32 * 4096 * 4 / t(m, grid) / 1e9) for m in (0, 1, 2)))
