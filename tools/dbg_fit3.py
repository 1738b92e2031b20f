import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.models import AE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from tests.golden_utils import base_hparams
dim = [1, 32, 32]
arch = load_handcrafted_arch(list(dim), 4, None, check_memory=False)
hp = base_hparams(arch, 'ae', None); hp['device'] = 'cuda'
sess = SyntheticSession(10, [5 + (t % 3) for t in range(10)], dim, seed=40, trial_splits='8;1;1;0')
gen = SyntheticSessionsGenerator([sess], device='cuda', placement=os.environ.get('PLACE', 'host_u8'))
mode = os.environ.get('MODE', '')
if mode == 'nopf':
    # no look-ahead: every fetch copies on the main stream
    orig = gen._fetch_host_u8
    def fetch(sess, trial, dtype):
        gen._pf = None
        img = orig(sess, trial, dtype)
        return img
    import types
    def no_lookahead(self, sess, trial, dtype):
        from behavenet_amd import _hip
        host = self._store[sess][0][trial]
        dev = host.to('cuda', non_blocking=True)
        return _hip.u8_to_unit_float(dev)
    gen._fetch_host_u8 = types.MethodType(no_lookahead, gen)
if mode == 'syncpf':
    orig = gen._fetch_host_u8
    def fetch(sess, trial, dtype):
        img = orig(sess, trial, dtype)
        gen._pf_stream.synchronize()
        return img
    gen._fetch_host_u8 = fetch
if mode == 'freshbuf':
    def staging(shape, slot):
        return torch.empty(shape, dtype=torch.uint8, device='cuda')
    gen._staging = staging
if mode == 'farbuf':
    pool = torch.zeros(64 << 20, dtype=torch.uint8, device='cuda')
    def staging(shape, slot):
        n = int(np.prod(shape))
        off = (slot * 3 + (shape[0] - 5)) * (4 << 20)
        return pool[off:off + n].view(shape)
    gen._staging = staging
bigs = {}
if mode == 'padbuf':
    G = 64 << 10
    def staging(shape, slot):
        key = (tuple(shape), slot)
        if key not in bigs:
            n = int(np.prod(shape))
            bigs[key] = (torch.zeros(n + 2 * G, dtype=torch.uint8, device='cuda'), n)
        big, n = bigs[key]
        return big[G:G + n].view(shape)
    gen._staging = staging
if mode in ('nocopy', 'noevents', 'd2d'):
    import types
    from behavenet_amd import _hip as H
    devsrc = [t.cuda() for t in gen._store[0][0]]
    def fetch(self, sess, trial, dtype):
        main = torch.cuda.current_stream()
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(); self._pf_slot = 0; self._pf_done = [None, None]
        host = self._store[sess][0][trial]
        # always copy the requested trial on main (correct data), like the fallback path
        slot = self._pf_slot
        dev_u8 = self._staging(host.shape, slot)
        if mode == 'noevents':
            self._pf_stream.synchronize()
        elif self._pf_done[slot] is not None:
            main.wait_event(self._pf_done[slot])
        dev_u8.copy_(host, non_blocking=True)
        img = H.u8_to_unit_float(dev_u8)
        done = torch.cuda.Event(); done.record(main)
        self._pf_done[slot] = done
        self._pf_slot = slot ^ 1
        queue = self._queues[sess][dtype]
        if queue:
            nxt = queue[0]
            h2 = self._store[sess][0][nxt]
            nslot = self._pf_slot
            buf = self._staging(h2.shape, nslot)
            with torch.cuda.stream(self._pf_stream):
                if mode != 'noevents' and self._pf_done[nslot] is not None:
                    self._pf_stream.wait_event(self._pf_done[nslot])
                if mode == 'd2d':
                    buf.copy_(devsrc[nxt], non_blocking=True)
                elif mode != 'nocopy':
                    buf.copy_(h2, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self._pf_stream)
                if mode == 'noevents':
                    self._pf_stream.synchronize()
        return img
    gen._fetch_host_u8 = types.MethodType(fetch, gen)
if mode == 'lazybig':
    lb = {}
    def staging(shape, slot):
        key = (tuple(shape), slot)
        if key not in lb:
            lb[key] = torch.zeros(2 << 20, dtype=torch.uint8, device='cuda')
        return lb[key][:int(np.prod(shape))].view(shape)
    gen._staging = staging
if mode == 'earlysmall':
    es = {}
    for nfr in (5, 6, 7):
        for slot in (0, 1):
            es[((nfr, 1, 32, 32), slot)] = torch.zeros((nfr, 1, 32, 32), dtype=torch.uint8, device='cuda')
    gen._staging = lambda shape, slot: es[(tuple(shape), slot)]
if mode == 'mainpf':
    gen._pf_stream = torch.cuda.current_stream()
    gen._pf_slot = 0
    gen._pf_done = [None, None]
if mode == 'keepev':
    keep = []
    orig = gen._fetch_host_u8
    def fetch(sess, trial, dtype):
        if gen._pf is not None: keep.append(gen._pf)
        img = orig(sess, trial, dtype)
        keep.append(gen._pf_done[:])
        return img
    gen._fetch_host_u8 = fetch
torch.manual_seed(0)
model = AE(hp).to('cuda')
opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-4)
flags = []
snaps = []
refs = [torch.from_numpy(u.astype(np.float32) / 255).cuda() for u in sess.images_u8]
for epoch in range(4):
    torch.manual_seed(epoch); np.random.seed(epoch)
    gen.reset_iterators('train')
    for i in range(gen.n_tot_batches['train']):
        model.train(); opt.zero_grad()
        data, ds = gen.next_batch('train')
        x = data['images'][0]
        t = int(data['batch_idx'][0])
        f_x = (x == refs[t]).all()
        out = model.loss(data, dataset=ds, accumulate_grad=True)
        f_g = torch.isfinite(opt.flat_g).all()
        snaps.append(opt.flat_g.clone())
        if epoch > 0: opt.step()
        f_p = torch.isfinite(opt.flat_p).all()
        flags.append(('ep%d it%d trial%d n%d loss %.5f' % (epoch, i, t, x.shape[0], out['loss']), f_x, f_g, f_p))
    gen.reset_iterators('val')
    data, ds = gen.next_batch('val')
    out = model.loss(data, dataset=ds, accumulate_grad=False)
    flags.append(('val ep%d loss %.5f' % (epoch, out['loss']), (data['images'][0] == refs[int(data['batch_idx'][0])]).all(), torch.tensor(True), torch.isfinite(opt.flat_p).all()))
torch.cuda.synchronize()
for tag, a, b, c in flags:
    print(tag, 'x_ok', bool(a), 'grad_finite', bool(b), 'param_finite', bool(c))

names = [(k, p.numel()) for k, p in model.named_parameters() if p.requires_grad]
for i, g in enumerate(snaps[:12]):
    if not torch.isfinite(g).all():
        off = 0
        for (k, n), o in zip(names, opt.offsets):
            seg = g[o:o + n]
            bad = (~torch.isfinite(seg)).sum().item()
            if bad:
                idx = (~torch.isfinite(seg)).nonzero().flatten()[:6].tolist()
                print('step', i, k, 'bad', bad, 'of', n, 'first idx', idx, 'vals', seg[idx[:3]].tolist())

for key, (big, n) in bigs.items():
    G = 64 << 10
    b = big.cpu().numpy()
    lo, hi = b[:G], b[G + n:]
    print('staging', key, 'n', n, 'nonzero before', int((lo != 0).sum()), 'after', int((hi != 0).sum()))
    if (hi != 0).any():
        idx = np.nonzero(hi)[0]
        print('   after: first', idx[:8], 'last', idx[-4:], 'as float32:', hi[idx[0] // 4 * 4: idx[0] // 4 * 4 + 32].view(np.float32))
    if (lo != 0).any():
        idx = np.nonzero(lo)[0]
        print('   before: first', idx[:8], 'last', idx[-4:])
