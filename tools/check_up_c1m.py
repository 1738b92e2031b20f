"""k_up_c1m (BN_UP_C1_M=1, tuning library) against k_up_c1v on dec.convT4: outputs, dL/dpre, frame sums,
and timing.  usage: BN_HIP_LIB=.../libbehavenet_hip_tuning.so python tools/check_up_c1m.py [Cb]"""
import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'worker':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from behavenet_amd import _hip
    cb, path = int(sys.argv[2]), sys.argv[3]
    N = 256
    g = torch.Generator().manual_seed(0)
    x = (torch.rand((N, 32, 64, 64), generator=g) - 0.3).cuda()
    w = ((torch.rand((32, cb, 5, 5), generator=g) - 0.5) * 0.2).cuda()
    b = (torch.rand((cb,), generator=g) - 0.5).cuda()
    tgt = torch.rand((N, cb, 128, 128), generator=g).cuda()
    msk = (torch.rand((N, cb, 128, 128), generator=g) < 0.7).float().cuda()
    geom = (N, 32, 64, 64, cb, 5, 5, 2, 1, 1, 128, 128)
    y = _hip.convT2d_fwd(x, w, b, geom, _hip.ACT_SIGMOID, 0.05)
    xh, dpre, part = _hip.convT2d_fwd_sqerr(x, w, b, tgt, msk, geom, _hip.ACT_SIGMOID, 0.05, True)
    xh2, dpre2, part2 = _hip.convT2d_fwd_sqerr(x, w, b, tgt, None, geom, _hip.ACT_SIGMOID, 0.05, False)
    ylr = _hip.convT2d_fwd(x, w, b, geom, _hip.ACT_LRELU, 0.05)
    torch.cuda.synchronize()
    ts = []
    for fn in (lambda: _hip.convT2d_fwd(x, w, b, geom, _hip.ACT_SIGMOID, 0.05),
               lambda: _hip.convT2d_fwd_sqerr(x, w, b, tgt, None, geom, _hip.ACT_SIGMOID, 0.05, False)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print('  BN_UP_C1_M=%s: fwd %.1f us, fwd + loss %.1f us (same buffers every launch)' % (os.environ.get('BN_UP_C1_M'), ts[0], ts[1]))
    np.savez(path, y=y.cpu().numpy(), xh=xh.cpu().numpy(), dpre=dpre.cpu().numpy(), s=part.sum(1).cpu().numpy(),
             dpre2=dpre2.cpu().numpy(), s2=part2.sum(1).cpu().numpy(), ylr=ylr.cpu().numpy())
    sys.exit(0)
cb = sys.argv[1] if len(sys.argv) > 1 else '1'
for m in ('0', '1'):
    subprocess.check_call([sys.executable, __file__, 'worker', cb, '/tmp/upc1m_%s.npz' % m], env=dict(os.environ, BN_UP_C1_M=m))
a, b = np.load('/tmp/upc1m_0.npz'), np.load('/tmp/upc1m_1.npz')
for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k]).max() / max(np.abs(a[k]).max(), 1e-30)
    print('  %-6s max |diff| / max = %.3e' % (k, d))
