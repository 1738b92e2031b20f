"""k_up2_mfma<4, 4> (dec.convT2 forward: 128 -> 64 channels, 16x16 -> 32x32 maps, 256 frames) against the base
addresses of its input / output: does the 231 us of the training step (198 for its siblings) depend on where the
allocator put the tensors?  python tools/probe_up2_align.py [layer D1|D2|D3]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from behavenet_amd import _hip
from tools.kbench import LAYERS

name = sys.argv[1] if len(sys.argv) > 1 else 'D2'
kind, ci, hi, wi, co, ho, wo, st, off = LAYERS[name]
N = 256
lib = _hip.load()
fn = lib.bn_convT2d_fwd
g = torch.Generator().manual_seed(0)
nx, ny = N * ci * hi * wi, N * co * ho * wo
pad = 1 << 22
xbuf = torch.empty(nx + pad, device='cuda'); ybuf = torch.empty(ny + pad, device='cuda')
w = ((torch.rand((ci, co, 5, 5), generator=g) - 0.5) / (ci * 6) ** 0.5).cuda()
b = (torch.rand((co,), generator=g) - 0.5).cuda()
wsb = lib.bn_conv_ws_bytes(_hip.OP_CONVT_FWD if hasattr(_hip, 'OP_CONVT_FWD') else 3, N, ci, hi, wi, co, 5, 5, st, off, off, ho, wo)
ws = torch.empty(max(int(wsb), 256), dtype=torch.uint8, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
src = (torch.rand(nx, generator=g) - 0.3).cuda()


def run(xo, yo, iters=30):
    x = xbuf[xo:xo + nx]; x.copy_(src)
    y = ybuf[yo:yo + ny]
    args = (ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(b.data_ptr()),
            ctypes.c_void_p(y.data_ptr()), N, ci, hi, wi, co, 5, 5, st, off, off, ho, wo, 1, ctypes.c_float(0.05),
            ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), ctypes.c_void_p(stream))
    for _ in range(5):
        rc = fn(*args); assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(name, 'x base %#x y base %#x' % (xbuf.data_ptr(), ybuf.data_ptr()))
for xo in (0, 64, 1024, 16384, 1 << 18, (1 << 20) + 4096):
    print('x +%8d floats:' % xo, ' '.join('%6.1f' % run(xo, yo) for yo in (0, 64, 1024, 16384, 1 << 18, (1 << 20) + 4096)), flush=True)
