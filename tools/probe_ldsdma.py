import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import debug_lib
lib = debug_lib.load()
src = torch.arange(256, dtype=torch.float32, device='cuda') + 100
out = torch.zeros(256, device='cuda')
lib.bn_debug_probe_lds_dma(src.data_ptr(), out.data_ptr(), 256, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print(out[:12].tolist())
print('even lanes ok:', bool((out[0::2] == src[0::2]).all()), ' odd lanes:', set(out[1::2].tolist()))
