"""MFMA + LDS-operand ceilings in the conv kernels' shapes (see debug_probe.hip)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import debug_lib
lib = debug_lib.load()
out = torch.empty(4096 * 512, device='cuda')
st = torch.cuda.current_stream().cuda_stream
names = {0: '32x32x2 2x2 + LDS operands', 1: '16x16x4 x25 + LDS operands', 2: '32x32x2 2x2 register operands'}
for mode in (2, 0, 1):
    for blocks, threads in ((256, 256), (512, 256), (1024, 256), (256, 512), (512, 512)):
        iters = 4000
        lib.bn_debug_probe_mfma_lds(out.data_ptr(), blocks, threads, 50, mode, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.bn_debug_probe_mfma_lds(out.data_ptr(), blocks, threads, iters, mode, st); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        per_wave_iter = 4 * 4096.0 if mode != 1 else 25 * 2048.0
        flop = blocks * (threads // 64) * iters * per_wave_iter
        print('%-32s grid %4d x %3d thr (%.1f waves/SIMD): %.3f ms  %6.1f TFLOP/s' % (
            names[mode], blocks, threads, blocks * threads / 64 / 1024.0, ms, flop / ms / 1e9))
