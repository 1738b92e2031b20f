"""Where export_latents' time goes on a file-backed session (BASELINE configs[4]): the feed alone (generator
iteration: file reads by the reader threads, pinned staging, H2D one trial ahead), the exporter on resident
uint8 trials, and the whole thing.   python tools/probe_export.py [trials]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from behavenet_amd.data.data_generator import ConcatSessionsGenerator, SyntheticSession, SyntheticSessionsGenerator
from behavenet_amd.data.synthetic import make_frames_u8
from behavenet_amd.data.trial_store import write_npz_session
from behavenet_amd.fitting.eval import export_latents
from behavenet_amd.models import AE

n_trials = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tmp = tempfile.mkdtemp(prefix='bn_export_', dir='/tmp')
ids = {'lab': 'lab', 'expt': 'expt', 'animal': 'animal', 'session': 'sess'}
sess_dir = os.path.join(tmp, 'lab', 'expt', 'animal', 'sess')
block = [make_frames_u8(bench.BATCH, bench.DIM, seed=1000 + i) for i in range(32)]
write_npz_session(os.path.join(sess_dir, 'data.npz'), {'images': [block[i % 32] for i in range(n_trials)]})


def gen_file():
    return ConcatSessionsGenerator(tmp, [ids], signals_list=[['images']], transforms_list=[[None]],
                                   paths_list=[[os.path.join(sess_dir, 'data.npz')]], device='cuda',
                                   placement='host_u8', keep_in_memory=False)


hp = bench.build_hparams(); hp.update({'expt_dir': tmp, 'device': 'cuda'})
torch.manual_seed(0)
ae = AE(hp).to('cuda'); ae.version = 0
for rep in range(2):
    g = gen_file(); g.serve_uint8 = True
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for dtype in ('train', 'val', 'test'):
        g.reset_iterators(dtype)
        for _ in range(g.n_tot_batches[dtype]):
            d, s = g.next_batch(dtype); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('feed alone (pass %d): %.3f ms per trial, %.0f frames/s' % (rep, dt / n * 1e3, n * 256 / dt))
for threads in (1, 3, 6):
    import behavenet_amd.data.data_generator as dg
    dg._LazyTrials._pool = None
    os.environ['BN_READ_THREADS'] = str(threads)
    g = gen_file(); g.serve_uint8 = True; g.read_ahead = max(4, threads + 2)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for dtype in ('train', 'val', 'test'):
        g.reset_iterators(dtype)
        for _ in range(g.n_tot_batches[dtype]):
            d, s = g.next_batch(dtype); n += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('feed alone, %d reader threads: %.3f ms per trial' % (threads, dt / n * 1e3))
sess = SyntheticSession(256, bench.BATCH, bench.DIM, seed=1, trial_splits='8;1;1;0')
gr = SyntheticSessionsGenerator([sess], device='cuda', placement='device_u8')
out = os.path.join(tmp, 'l.pkl')
for graph in (True, False):
    ae.hparams['hip_graph_encode'] = graph
    export_latents(gr, ae, filename=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    export_latents(gr, ae, filename=out)
    dt = time.perf_counter() - t0
    print('exporter on resident uint8 trials (graph %s): %.3f ms per trial' % (graph, dt / 250 * 1e3))
shutil.rmtree(tmp, ignore_errors=True)
