import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import debug_lib
lib = debug_lib.load()
out = torch.empty(4096 * 256, device='cuda')
for blocks in (256, 512, 1024, 2048):
    iters = 20000
    st = torch.cuda.current_stream().cuda_stream
    lib.bn_debug_probe_mfma(out.data_ptr(), blocks, 100, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.bn_debug_probe_mfma(out.data_ptr(), blocks, iters, st); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flop = blocks * 4 * iters * 4 * 4096.0
    print('blocks %d: %.3f ms  %.1f TFLOP/s' % (blocks, ms, flop / ms / 1e9))
