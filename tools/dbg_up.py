import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd import _hip
torch.manual_seed(0)
N, Ci, Hi = 2, 32, 64
geom = (N, Ci, Hi, Hi, 1, 5, 5, 2, 1, 1, 2 * Hi, 2 * Hi)
x = torch.rand(N, Ci, Hi, Hi)
b = torch.zeros(1)
def ref(x, w):
    y = F.conv_transpose2d(x, w, None, stride=2)
    return y[:, :, 1:1 + 2 * Hi, 1:1 + 2 * Hi]
def run(w, tag):
    got = _hip.convT2d_fwd(x.cuda(), w.cuda(), b.cuda(), geom, _hip.ACT_NONE, 0.05).cpu()
    want = ref(x, w)
    err = (got - want).abs()
    if err.max() > 1e-4 * want.abs().max():
        idx = (err == err.max()).nonzero()[0].tolist()
        bad_rows = sorted(set((err.amax(dim=(0, 1, 3)) > 1e-4 * want.abs().max()).nonzero().flatten().tolist()))
        bad_cols = sorted(set((err.amax(dim=(0, 1, 2)) > 1e-4 * want.abs().max()).nonzero().flatten().tolist()))
        print(tag, 'ERR %.3e at %s rows %s cols %s' % (err.max(), idx, bad_rows[:12], bad_cols[:12]))
        return False
    return True
w = torch.rand(Ci, 1, 5, 5) - 0.5
print('full', run(w, 'full'))
for c in range(Ci):
    wc = torch.zeros_like(w); wc[c] = w[c]
    run(wc, 'chan %d' % c)
for t in range(25):
    wt = torch.zeros_like(w); wt.view(Ci, 25)[:, t] = w.view(Ci, 25)[:, t]
    run(wt, 'tap %d (r=%d s=%d)' % (t, t // 5, t % 5))
