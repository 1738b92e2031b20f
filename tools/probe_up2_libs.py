"""dec.convT1..3 forward (k_up2_mfma<3|4|5, 4>) with several builds of the library, through the C ABI alone:
python tools/probe_up2_libs.py lib1.so lib2.so ...   (us per launch, 256 frames, back-to-back launches)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import LAYERS

N = 256
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.bn_convT2d_fwd
    fn.restype = ctypes.c_int
    lib.bn_conv_ws_bytes.restype = ctypes.c_size_t
    out = []
    for rep in range(2):
        for name in ('D1', 'D2', 'D3'):
            kind, ci, hi, wi, co, ho, wo, st, off = LAYERS[name]
            g = torch.Generator().manual_seed(0)
            x = (torch.rand((N, ci, hi, wi), generator=g) - 0.3).cuda()
            w = ((torch.rand((ci, co, 5, 5), generator=g) - 0.5) / (ci * 6) ** 0.5).cuda()
            b = (torch.rand((co,), generator=g) - 0.5).cuda()
            y = torch.empty((N, co, ho, wo), device='cuda')
            wsb = lib.bn_conv_ws_bytes(4, N, ci, hi, wi, co, 5, 5, st, off, off, ho, wo)
            ws = torch.empty(max(int(wsb), 256), dtype=torch.uint8, device='cuda')
            args = (ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                    ctypes.c_void_p(y.data_ptr()), N, ci, hi, wi, co, 5, 5, st, off, off, ho, wo, 1, ctypes.c_float(0.05),
                    ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            for _ in range(10):
                assert fn(*args) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                fn(*args)
            e1.record(); torch.cuda.synchronize()
            out.append('%s %.1f' % (name, e0.elapsed_time(e1) / 40 * 1e3))
    print('%-44s %s' % (path, '  '.join(out)), flush=True)
