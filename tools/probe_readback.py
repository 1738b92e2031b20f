"""Is a non_blocking device->pinned copy really asynchronous on this runtime?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd.hip_functions import Readback
a = torch.randn(8192, 8192, device='cuda')
s = torch.zeros((), device='cuda')
for shape in ((1,), (2, 3), (4096,)):
    t = torch.zeros(shape, device='cuda')
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            b = a @ a            # ~7 ms each at fp32
        t1 = time.perf_counter()
        r = Readback(t)
        t2 = time.perf_counter()
        v = r.numpy()
        t3 = time.perf_counter()
        print('shape %-8s issue matmuls %.2f ms | Readback() call %.3f ms | wait %.2f ms' % (
            shape, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
