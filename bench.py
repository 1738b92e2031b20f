#!/usr/bin/env python
"""Headline benchmark: conv-AE training frames/s, 128x128x1 frames, batch (= trial) 256.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shard trial|frames]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the script launches its own
N ranks, one process per GPU, through torch.distributed.run on 127.0.0.1 -- the counterpart of the
reference's entry point, which forks one process per GPU itself
(behavenet/fitting/ae_grid_search.py:173-181, configs/ae_jsons/ae_compute.json:10).

One "step" is one pass of the reference's hot loop over one 256-frame trial
(training.py:336-352 with i_epoch > 0): zero_grad -> next_batch -> AE.loss(accumulate_grad=True)
(the reference's per-chunk loss normalisation, chunks of 200 + 56 frames; here ONE forward and
ONE backward pass over all 256 frames produce the same sum of per-chunk-mean gradients) ->
[RCCL all-reduce of the flat gradient] -> Adam(amsgrad) step.  Trials are synthetic uint8-noise frames (float32/255) already
resident in HBM.  With N > 1: ``--shard trial`` (default; weak scaling) every rank consumes its own
trial per step and the gradients are averaged over ranks before the identical optimizer step,
as ``fit()`` does in 'trial' mode; ``--shard frames`` (strong scaling, the parity-exact reading of
BASELINE configs[2]) all ranks take slices of the SAME 256-frame trial, every chunk term is
normalised by the global chunk size and the gradients are summed.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline:     enc.conv0 (the HBM-bound encoder conv BASELINE.json targets): algorithmic bytes
                (589,824 B/frame = 65,536 in + 524,288 out) / HIP-event time of its launches
                inside the timed region, vs 8 TB/s.
  cpu_baseline: the CPU oracle (oracle/ref_cpu.py, the pinned restatement of the reference)
                running the same step on this host's cores.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from behavenet_amd import _hip  # noqa: E402
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator  # noqa: E402
from behavenet_amd.fitting import distributed as bdist  # noqa: E402
from behavenet_amd.fitting.optim import FlatAdamAMSGrad  # noqa: E402
from behavenet_amd.models import AE  # noqa: E402
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch  # noqa: E402

METRIC = 'AE training frames/sec (128x128x1, batch 256)'
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
DIM = [1, 128, 128]
N_LATENTS = 12
BATCH = 256
CONV0_BYTES_PER_FRAME = 65536 + 524288            # enc.conv0 in + out
TRAIN_FLOP_PER_FRAME = 2.0843e9                   # fwd+bwd, 2*MAC
PRIME_STEPS = 24                                  # untimed runtime priming during setup (minimum)
PRIME_MAX = 120                                   # ... and its ceiling


def build_hparams():
    arch = load_handcrafted_arch(list(DIM), N_LATENTS, None, check_memory=False)
    hp = dict(arch)
    hp.update({'model_class': 'ae', 'device': 'cuda', 'learning_rate': 1e-4, 'l2_reg': 0.0,
               'fit_sess_io_layers': False, 'n_datasets': 1, 'rng_seed_model': 0})
    return hp


_PARTS = None    # BN_BENCH_TRACE=1: host time of the components of every step
_LOSS = None     # what a step calls for the loss: model.loss, or its HIP-graph replay (fitting/graph_step.py)
_AVERAGE = False  # N > 1, --shard trial: mean over the ranks' trials (fit()'s 'trial' mode)


def one_step(model, opt, gen):
    t = [time.perf_counter()] if _PARTS is not None else None
    model.train()
    opt.zero_grad()
    if t: t.append(time.perf_counter())
    data, dataset = gen.next_batch('train')
    if data is None:
        gen.reset_iterators('train')
        data, dataset = gen.next_batch('train')
    if t: t.append(time.perf_counter())
    loss = (_LOSS or model.loss)(data, dataset=dataset, accumulate_grad=True)
    if t: t.append(time.perf_counter())
    if getattr(opt, 'shard_over', 1) > 1:
        bdist.sharded_step(opt, average=_AVERAGE)      # --shard-optimizer
    else:
        bdist.reduce_gradients(opt, average=_AVERAGE)
        opt.step()
    if t:
        t.append(time.perf_counter())
        _PARTS.append([(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])])
    return loss


def cpu_baseline(hp_template, budget_s=20.0):
    """Time the CPU oracle on the same workload (bounded sample) on this host's cores."""
    from oracle import ref_cpu
    from behavenet_amd.data.synthetic import make_frames
    from behavenet_amd.hostinfo import limit_host_threads, usable_cpus
    # threads = the CPUs this container may use (cgroup quota), not the machine's core count:
    # torch's default of 128 threads on a 16-CPU quota ran this baseline ~4 x slower (round 3's
    # 54.7 frames/s "on 128 cores"); `cores` below is what is actually used
    threads = limit_host_threads()
    hp = dict(hp_template)
    hp['device'] = 'cpu'
    torch.manual_seed(0)
    model = ref_cpu.AE(hp)
    opt = ref_cpu.make_optimizer(model, hp)
    warm = {'images': torch.from_numpy(make_frames(16, DIM, seed=5))[None]}
    ref_cpu.train_step(model, opt, warm)
    data = {'images': torch.from_numpy(make_frames(BATCH, DIM, seed=6))[None]}
    steps, t0 = 0, time.perf_counter()
    while True:
        ref_cpu.train_step(model, opt, data)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 5 or (steps >= 1 and el + el / steps > 1.5 * budget_s):
            break
    return {'value': BATCH * steps / el, 'unit': 'frames/s', 'cores': threads,
            'kind': 'port',
            'sample_short': '%d full training steps (batch %d, fwd+bwd+Adam) in %.1f s, %d threads' % (
                steps, BATCH, el, threads),
            'sample': '%d full training step(s) of the same workload (batch %d, chunks 200+56, '
                      'fwd+bwd+Adam) in %.1f s, torch %s CPU, %d threads = the container\'s CPU '
                      'quota (the machine shows %d hardware threads)' % (
                          steps, BATCH, el, torch.__version__, usable_cpus(), os.cpu_count() or 0)}


def _timed(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _guarded(label, fn, *args, **kwargs):
    """A secondary measurement must never cost the headline line: what goes wrong in one of them (a full /tmp
    under the trial store, an out-of-memory in an odd geometry) is reported in ITS entry."""
    try:
        # every secondary starts from an empty allocator cache, like a fresh process: the blocks the previous
        # ones left behind (thousands of small tensors of the export / fit runs) otherwise end up under this one's
        # activations (round 6: ae_arch_2 / batch-norm steps read 1.7-2.5 % slower inside the bench process than
        # alone, the secondaries AFTER them did not)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        got = fn(*args, **kwargs)
        got['id'] = label
        return got
    except Exception as err:                                    # noqa: BLE001
        import traceback
        traceback.print_exc()
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        except Exception:                                       # noqa: BLE001
            pass
        return {'id': label, 'config': '%s (%s)' % (label, str(args[-1])[:120] if args and isinstance(args[-1], str) else ''),
                'value': None, 'error': '%s: %s' % (type(err).__name__, str(err)[:500])}


def secondary_configs(hp_ae, feed_rates=True):
    """The other BASELINE configs that fit one GPU, a few steps each, all driver-run:
    configs[3] (PS-VAE training), configs[4] (encode-only from resident uint8 trials, the
    per-GPU share of the 1M-frame job) and the PCIe-inclusive training rate of the headline
    workload (pinned uint8 trials prefetched per batch).  Same synthetic data recipe, same
    timing brackets (synchronize on both sides) as the headline."""
    from behavenet_amd.models import PSVAE
    from behavenet_amd.data.synthetic import base_hparams, make_frames, make_labels
    out = []
    # --- configs[3]: PS-VAE, 2x128x128, 16 latents, 4 labels, batch 256
    dim4 = [2, 128, 128]
    arch = load_handcrafted_arch(list(dim4), 16, None, check_memory=False)
    hp = base_hparams(arch, 'ps-vae', {'ps_vae.alpha': 1000, 'ps_vae.beta': 5,
                                       'ps_vae.anneal_epochs': 100, 'max_n_epochs': 200})
    hp['n_labels'] = 4
    hp['device'] = 'cuda'
    np.random.seed(0)
    torch.manual_seed(0)
    m = PSVAE(hp).to('cuda')
    m.curr_epoch = 3
    opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
    data = {'images': torch.from_numpy(make_frames(BATCH, dim4, seed=1)).cuda()[None],
            'labels': torch.from_numpy(make_labels(BATCH, 4, seed=2)).cuda()[None]}

    def step4():
        m.train()
        opt.zero_grad()
        m.loss(data, dataset=0, accumulate_grad=True)
        opt.step()
    t4 = _timed(step4, 30, 20)
    flop4 = 3 * 0.7079e9        # fwd 0.7079 GFLOP/frame, training = 3x
    out.append({'id': 'psvae_2x128x128',
                'config': 'configs[3]: PS-VAE training, 2x128x128, 16 latents, 4 labels, batch 256 '
                          '(all 11 loss keys, decomposed KL, Adam(amsgrad))',
                'value': round(BATCH / t4, 1), 'unit': 'frames/s', 'ms_per_step': round(t4 * 1e3, 3),
                'steps': 20, 'roofline': {
                    'bound': 'mfma', 'achieved': round(flop4 * BATCH / t4 / 1e12, 2),
                    'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(flop4 * BATCH / t4 / 1e12 / FP32_PEAK_TFLOPS, 4),
                    'note': 'whole step, algorithmic 2.124 GFLOP/frame'}})
    del m, opt, data
    # --- configs[4], one GPU's share: encode resident uint8 trials -> latents (no float copy)
    torch.manual_seed(0)
    ae = AE(dict(hp_ae)).to('cuda')
    ae.eval()
    xu = torch.randint(0, 255, (BATCH, 1, 128, 128), dtype=torch.uint8, device='cuda')

    def enc():
        with torch.no_grad():
            ae.encoding(xu, dataset=0)
    t5 = _timed(enc, 20, 50)
    out.append({'id': 'encode_u8',
                'config': 'configs[4] per GPU: encode-only, 1x128x128 uint8 trials of 256 frames '
                          '(resident in HBM) -> 12 latents; uint8 -> float fused into enc.conv0',
                'value': round(BATCH / t5, 1), 'unit': 'frames/s', 'ms_per_trial': round(t5 * 1e3, 3),
                'steps': 50, 'seconds_per_1M_frames': round(1e6 / (BATCH / t5), 2),
                'roofline': {'bound': 'mfma', 'achieved': round(0.3474e9 * BATCH / t5 / 1e12, 2),
                             'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': round(0.3474e9 * BATCH / t5 / 1e12 / FP32_PEAK_TFLOPS, 4),
                             'note': 'encoder forward, algorithmic 0.3474 GFLOP/frame'}})
    del ae
    # --- the product entry point itself: fit() on the headline workload
    out.append(_guarded('fit', fit_throughput, hp_ae))
    out.append(_guarded('export_latents', export_latents_throughput, hp_ae))
    # --- geometries OFF the benchmark's fast paths (their cost was never measured)
    cfg = os.path.join(REPO, 'behavenet_amd', 'configs', 'ae_jsons')
    out.append(_guarded('ae_arch_default', geometry_step, os.path.join(cfg, 'ae_arch_default.json'), [1, 128, 128],
                             'shipped configs/ae_jsons/ae_arch_default.json (4 layers 32-64-256-512, '
                             'k5 s2, last map 8x8) on 1x128x128'))
    out.append(_guarded('1x64x48', geometry_step, None, [1, 64, 48],
                             'default architecture on 1x64x48 frames (the reference\'s '
                             'tests/integration.py shape)'))
    out.append(_guarded('ae_arch_2', geometry_step, os.path.join(cfg, 'ae_arch_2.json'), [1, 128, 128],
                             'shipped configs/ae_jsons/ae_arch_2.json (5 layers of 64 channels, k4, '
                             'strides 2,2,2,2,1) on 1x128x128'))
    out.append(_guarded('2x192x160', geometry_step, None, [2, 192, 160],
                             'default architecture on 2x192x160 frames (48x40 / 24x20 / 12x10 maps '
                             'directly on the stride-2 families since round 4; edge layers on tiles)'))
    out.append(_guarded('1x192x192', geometry_step, None, [1, 192, 192],
                             'default architecture on 1x192x192 frames (the frame size of the reference\'s '
                             'examples/msps-vae/ibl_ephys_params.json; 48x48 maps: weight gradient in '
                             'column windows)'))
    out.append(_guarded('batch_norm', geometry_step, None, [1, 128, 128],
                             'default architecture with ae_batch_norm = 1 on 1x128x128 (per-chunk '
                             'statistics inside one pass; momentum None = cumulative average, the '
                             'reference\'s default)', names=False, extra={'ae_batch_norm': True}))
    out.append(_guarded('maxpool', geometry_step, os.path.join(REPO, 'tests', 'golden', 'arch_maxpool.json'), [1, 128, 128],
                             'max-pooling test architecture (tests/golden/arch_maxpool.json: 5x5 stride-1 '
                             'conv 1 -> 16 / pool / conv 16 -> 32 / pool, mirrored unpooling decoder) on '
                             '1x128x128', names=False))
    # two architectures as the reference's random search draws them (kernel sizes 3 / 5 / 7 / 9 with equal weight,
    # models/ae_model_architecture_generator.py:94 of the reference)
    out.append(_guarded('all_3x3', geometry_step, os.path.join(REPO, 'tools', 'arch_jsons', 'drawn_k3.json'), [1, 128, 128],
                             'drawn architecture, all 3x3 stride 2 (32-64-128-256-512; the tap window [1, 4) of the '
                             '5x5 stride-2 families) on 1x128x128', names=False))
    out.append(_guarded('k7_5_9_3', geometry_step, os.path.join(REPO, 'tools', 'arch_jsons', 'drawn_k7_k5_k9_k3.json'), [1, 128, 128],
                             'drawn architecture, kernels 7-5-9-3 stride 2 (32-64-128-256; 7x7 / 9x9 as stride-1 5x5 '
                             'layers on the four phases of the big map, no im2col) on 1x128x128', names=False))
    if feed_rates:
        # --- the headline step fed over PCIe: pinned uint8 trials, one-trial look-ahead
        torch.manual_seed(0)
        model = AE(dict(hp_ae)).to('cuda')
        opt = FlatAdamAMSGrad(model.get_parameters(), lr=hp_ae['learning_rate'])
        sess = SyntheticSession(20, BATCH, DIM, seed=100, trial_splits='8;1;1;0')
        gen = SyntheticSessionsGenerator([sess], device='cuda', placement='host_u8')
        torch.manual_seed(1)
        np.random.seed(1)
        gen.reset_iterators('train')
        tp = _timed(lambda: one_step(model, opt, gen), 20, 20)
        out.append({'id': 'pcie_fed', 'config': 'configs[1] fed over PCIe: same training step, trials in pinned host '
                              'memory as uint8 (4.2 MB each), copied one trial ahead on a copy stream',
                    'value': round(BATCH / tp, 1), 'unit': 'frames/s',
                    'ms_per_step': round(tp * 1e3, 3), 'steps': 20})
    return out


def fit_throughput(hp_ae, n_epochs=2):
    """`fit()` (reference training.py:244-461) over 20 resident trials of the headline workload:
    epoch 0 without optimizer steps, then `n_epochs` epochs of 16 training trials, a validation
    pass (2 trials) after every epoch, metric rows, best-model snapshot, the final test rows --
    frames/s over EVERY trial it pushes through the model (train + val + test)."""
    import contextlib
    import tempfile
    from behavenet_amd.fitting.training import fit
    tmp = tempfile.mkdtemp()
    hp = dict(hp_ae)
    hp.update({'max_n_epochs': n_epochs, 'min_n_epochs': n_epochs, 'enable_early_stop': False,
               'val_check_interval': 1, 'expt_dir': tmp, 'version': 0, 'device': 'cuda',
               'rng_seed_train': 0, 'export_latents': False, 'early_stop_history': 10,
               'progress_bar': False})
    os.makedirs(os.path.join(tmp, 'version_0'), exist_ok=True)

    class Exp(object):
        version = 0

        def __init__(self):
            self.rows = []

        def log(self, row):
            self.rows.append(dict(row))

        def save(self):
            pass

    def run():
        torch.manual_seed(0)
        model = AE(dict(hp)).to('cuda')
        model.version = 0
        sess = SyntheticSession(20, BATCH, DIM, seed=100, trial_splits='8;1;1;0')
        gen = SyntheticSessionsGenerator([sess], device='cuda', placement='device')
        exp = Exp()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):
            fit(hp, model, gen, exp, method='ae')
        torch.cuda.synchronize()
        n = gen.n_tot_batches
        trials = (n_epochs + 1) * (n['train'] + n['val']) + n['test']
        return time.perf_counter() - t0, trials, len(exp.rows), n
    run()                                  # (allocator pools, kernel attribute calls)
    # (the faster of two timed runs: fit() synchronises with the host at every validation check, where a
    # host hiccup costs it what it cannot cost the free-running bench loop)
    dt, trials, rows, n = min(run(), run())
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {'config': 'configs[1] through the product entry point fit(): epoch 0 + %d epochs over %d '
                      'train / %d val / %d test trials of 256 frames (resident float32), validation '
                      'after every epoch, metric rows, best-model snapshot (device-side refresh) and best_val_model.pt (background writer), test rows'
                      % (n_epochs, n['train'], n['val'], n['test']),
            'value': round(trials * BATCH / dt, 1), 'unit': 'frames/s (train + val + test trials)',
            'seconds': round(dt, 4), 'timed': 'the faster of two runs after one untimed run',
            'trials_through_the_model': trials, 'metric_rows': rows,
            'ms_per_trial': round(dt * 1e3 / trials, 3)}


def export_latents_throughput(hp_ae, n_trials=2048):
    """BASELINE configs[4] through the product entry point: ``export_latents()`` (reference
    fitting/eval.py:6-118) over a FILE-BACKED session -- ``n_trials`` trials of 256 uint8 frames in a
    ``data.npz`` trial store (the mirror of the reference's ``data.hdf5``, one member per trial) ->
    ``ConcatSessionsGenerator`` (pinned uint8, reader threads, one-trial-ahead device copy) -> encoder
    (uint8 -> float fused into enc.conv0) -> latents kept on the device -> one transfer -> the pickle the
    ARHMM stage reads.  Timed end to end: generator iteration, file reads, H2D copies, encode, D2H, pickle."""
    import pickle
    import shutil
    import tempfile
    from behavenet_amd.data.data_generator import ConcatSessionsGenerator
    from behavenet_amd.data.synthetic import make_frames_u8
    from behavenet_amd.data.trial_store import write_npz_session
    from behavenet_amd.fitting.eval import export_latents
    tmp = tempfile.mkdtemp(prefix='bn_export_', dir='/tmp')
    try:
        # (the store is 4.2 MB per trial: no more than a third of what /tmp has free)
        free = shutil.disk_usage(tmp).free
        n_trials = int(max(64, min(n_trials, free // 3 // (BATCH * int(np.prod(DIM))))))
        ids = {'lab': 'lab', 'expt': 'expt', 'animal': 'animal', 'session': 'sess'}
        sess_dir = os.path.join(tmp, 'lab', 'expt', 'animal', 'sess')
        os.makedirs(sess_dir)
        # (the frames' content does not matter for the rate: 64 distinct noise trials, written n_trials / 64 times)
        block = [make_frames_u8(BATCH, DIM, seed=1000 + i) for i in range(64)]
        t_w = time.perf_counter()
        write_npz_session(os.path.join(sess_dir, 'data.npz'),
                          {'images': [block[i % 64] for i in range(n_trials)]})
        t_w = time.perf_counter() - t_w
        gen = ConcatSessionsGenerator(tmp, [ids], signals_list=[['images']], transforms_list=[[None]],
                                      paths_list=[[os.path.join(sess_dir, 'data.npz')]], device='cuda',
                                      placement='host_u8', keep_in_memory=False)
        hp = dict(hp_ae)
        hp.update({'expt_dir': tmp, 'device': 'cuda'})
        torch.manual_seed(0)
        ae = AE(hp).to('cuda')
        ae.version = 0
        out_file = os.path.join(tmp, 'latents.pkl')

        def run():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                export_latents(gen, ae, filename=out_file)
            return time.perf_counter() - t0
        run()                                   # (page cache, allocator pools, reader threads)
        dt = run()
        with open(out_file, 'rb') as f:
            got = pickle.load(f)
        n_done = sum(1 for a in got['latents'] if a.shape == (BATCH, N_LATENTS))
        frames = n_done * BATCH
        return {'config': 'configs[4] through the product entry point export_latents(): %d trials of 256 uint8 '
                          'frames (1x128x128) from a data.npz trial store on local disk (page cache warm: second '
                          'pass) -> ConcatSessionsGenerator (pinned uint8, 2 reader threads, device copy one trial '
                          'ahead) -> encoder (a trial in one pass) -> latents on the device -> one D2H -> *_latents.pkl' % n_trials,
                'value': round(frames / dt, 1), 'unit': 'frames/s', 'seconds': round(dt, 3),
                'trials_encoded': n_done, 'ms_per_trial': round(dt * 1e3 / max(n_done, 1), 3),
                'seconds_per_1M_frames': round(1e6 * dt / max(frames, 1), 2),
                'store_bytes': os.path.getsize(os.path.join(sess_dir, 'data.npz')),
                'store_write_seconds': round(t_w, 2)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_hbm_traffic(kernel_substr='k_down_c1p', timeout_s=150):
    """HBM bytes per launch of the roofline kernel from the PMC counters, collected NOW: two separate
    `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: KB units,
    FETCH_SIZE doubled on gfx950) over tools/run_layer.py (enc.conv0 forward, 256 frames per launch),
    as child processes -- counters cannot be collected from inside this process.
    -> (bytes or None, {'FETCH_SIZE_KB', 'WRITE_SIZE_KB', 'launches'} or the reason it failed)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    got = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='bn_pmc_', dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'run',
               '--', sys.executable, os.path.join(REPO, 'tools', 'run_layer.py'), '--layer', 'E0',
               '--op', 'fwd', '--n', str(BATCH), '--iters', '6']
        try:
            subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            vals = []
            for f in files:
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get('Counter_Name') == counter and kernel_substr in r.get('Kernel_Name', ''):
                            vals.append(float(r['Counter_Value']))
        except Exception as err:                                    # noqa: BLE001
            return None, '%s pass failed: %s' % (counter, err)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if len(vals) < 3:
            return None, '%s pass: %d launches of %s found' % (counter, len(vals), kernel_substr)
        got[counter] = sum(vals[1:]) / len(vals[1:])            # (first launch: cold caches)
        got['launches'] = len(vals)
    info = {'FETCH_SIZE_KB': round(got['FETCH_SIZE'], 1), 'WRITE_SIZE_KB': round(got['WRITE_SIZE'], 1),
            'launches': got['launches']}
    return (2.0 * got['FETCH_SIZE'] + got['WRITE_SIZE']) * 1024.0, info


def geometry_step(arch_json, dim, label, batch=256, names=True, extra=None):
    """Training step of an architecture / frame size the specialised kernels were NOT tuned for:
    ms per step and, layer by layer and role by role, the kernel the dispatch chose."""
    from behavenet_amd.data.synthetic import base_hparams, make_frames
    arch = load_handcrafted_arch(list(dim), N_LATENTS, arch_json, check_memory=False)
    hp = base_hparams(arch, 'ae', dict(extra) if extra else None)
    hp['device'] = 'cuda'
    torch.manual_seed(0)
    m = AE(hp).to('cuda')
    opt = FlatAdamAMSGrad(m.get_parameters(), lr=1e-4)
    data = {'images': torch.from_numpy(make_frames(batch, dim, seed=1)).cuda()[None]}

    def step():
        m.train()
        opt.zero_grad()
        m.loss(data, dataset=0, accumulate_grad=True)
        opt.step()
    t = _timed(step, 12, 20)
    kernels = {}
    # layer by layer and role by role: one extra step each with the hook on the nth matching call
    # (layers may share a family and a channel pair: the forward pass visits them bottom-up, the
    # backward pass top-down)
    for stack, fams in () if not names else (('encoding', (('fwd', _hip.PROF_CONV_FWD, False), ('bwd_data', _hip.PROF_CONV_BWD_D, True),
                                      ('bwd_weight', _hip.PROF_CONV_BWD_W, False))),
                        ('decoding', (('fwd', _hip.PROF_CONVT_FWD, False), ('bwd_data', _hip.PROF_CONVT_BWD_D, True),
                                      ('bwd_weight', _hip.PROF_CONVT_BWD_W, True)))):
        plan = getattr(m, stack)._plan

        def ck(layer, swap):
            # (C, K) as the dispatch reports them: big-side, small-side channels for the
            # gather-down / weight-gradient launches, the other way round for gather-up
            big, small = (layer.cin, layer.cout) if layer.kind == 'conv' else (layer.cout, layer.cin)
            return (small, big) if (layer.kind == 'conv') == swap else (big, small)
        for i, layer in enumerate(plan):
            for role, fam, swap in fams:
                if stack == 'encoding' and i == 0 and role == 'bwd_data':
                    continue
                c, k = ck(layer, swap)
                same = [j for j, other in enumerate(plan) if ck(other, swap) == (c, k)
                        and not (stack == 'encoding' and j == 0 and role == 'bwd_data')]
                nth = same.index(i) if role == 'fwd' else same[::-1].index(i)
                _hip.prof_select(fam, c, k, nth=nth)
                step()
                torch.cuda.synchronize()
                ms, n, name = _hip.prof_read()
                kms, kn = _hip.prof_read_main()
                _hip.prof_select(_hip.PROF_NONE)
                key = '%s.%d %s' % ('enc' if stack == 'encoding' else 'dec', i, role)
                kernels[key] = ('%s %.0f us' % (name, (kms / kn if kn else ms / n) * 1e3)) if n else 'not matched'
    fwd_flop = sum(2.0 * l.cin * l.cout * l.R * l.S * (l.hout * l.wout if l.kind == 'conv' else l.hin * l.win)
                   for st in ('encoding', 'decoding') for l in getattr(m, st)._plan)
    tf = 3 * fwd_flop * batch / t / 1e12
    return {'config': label + ', batch %d, 12 latents: training step' % batch,
            'value': round(batch / t, 1), 'unit': 'frames/s', 'ms_per_step': round(t * 1e3, 3),
            'steps': 20, 'roofline': {'bound': 'mfma', 'achieved': round(tf, 2), 'peak': FP32_PEAK_TFLOPS,
                                      'unit': 'TFLOP/s', 'frac': round(tf / FP32_PEAK_TFLOPS, 4),
                                      'note': 'whole step, 3 x forward conv FLOPs (zero-padded taps counted)'},
            'dispatched_kernels': kernels}


LINE_LIMIT = 4096     # the driver keeps a bounded tail of stdout: the result line stays far below it


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + '...'


def compact_line(full):
    """The ONE line for stdout (< LINE_LIMIT characters) out of the full result dict: the contract's keys,
    `roofline` and `cpu_baseline` in numbers, `secondary` as {id, value, ms_per_step, frac}.  Everything
    else (kernel tables, prose, per-layer rooflines) goes to the detail file (`write_detail`)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
            'scaling', 'vs_baseline', 'dtype', 'data', 'hip_graph', 'final_loss', 'error', 'rank')
    line = {k: full[k] for k in keep if k in full}
    cfg = full.get('config') or {}
    line['config'] = {'workload': _short(cfg.get('workload_short') or cfg.get('workload', ''), 160)}
    for k in ('frames_per_step_per_gpu', 'global_frames_per_step'):
        if k in cfg:
            line['config'][k] = cfg[k]
    if full.get('n_gpus', 1) > 1 and cfg.get('sharding'):
        line['config']['sharding'] = _short(cfg['sharding'], 60)
    roof = full.get('roofline')
    if roof:
        line['roofline'] = {k: roof.get(k) for k in (
            'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'launches', 'avg_launch_us',
            'algorithmic_bytes_per_launch_avg', 'traffic')}
        line['roofline']['kernel'] = _short(line['roofline']['kernel'], 48)
    if 'whole_step_fp32_frac' in full:
        line['whole_step_fp32_frac'] = full['whole_step_fp32_frac']
    cpu = full.get('cpu_baseline')
    if cpu:
        line['cpu_baseline'] = {'value': None if cpu.get('value') is None else round(cpu['value'], 2),
                                'unit': cpu.get('unit'), 'cores': cpu.get('cores'), 'kind': cpu.get('kind'),
                                'sample': _short(cpu.get('sample_short') or cpu.get('sample', ''), 100)}
        if 'speedup_vs_cpu_baseline' in full:
            line['speedup_vs_cpu_baseline'] = full['speedup_vs_cpu_baseline']
    ar = full.get('allreduce')
    if ar:
        line['allreduce'] = {k: ar[k] for k in (
            'world_size', 'backend', 'op', 'gradient_bytes', 'allreduce_alone_ms', 'shard_optimizer',
            'ms_per_step_overlapped', 'ms_per_step_behind') if k in ar}
        line['allreduce']['chosen'] = _short(ar.get('chosen', ''), 48)
        ref = ar.get('single_gpu_reference')
        if ref:
            line['allreduce']['single_gpu_ms_per_step_max'] = ref.get('ms_per_step_max')
    sec = full.get('secondary')
    if sec:
        rows = []
        for s in sec:
            row = {'id': _short(s.get('id', '?'), 24), 'value': s.get('value')}
            ms = s.get('ms_per_step', s.get('ms_per_trial'))
            if ms is not None:
                row['ms_per_step'] = ms
            if isinstance(s.get('roofline'), dict):
                row['frac'] = s['roofline'].get('frac')
            if s.get('error'):
                row['error'] = _short(s['error'], 60)
            rows.append(row)
        line['secondary'] = rows
    if full.get('detail_file'):
        line['detail_file'] = full['detail_file']
    text = json.dumps(line, separators=(',', ':'))
    # (belt and braces: whatever grew, the headline keys go out)
    for drop in ('secondary', 'allreduce', 'detail_file'):
        if len(text) < LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(',', ':'))
    return text


def write_detail(full):
    """The full result (kernel tables, per-layer rooflines, prose) -> gpurun_out/bench_detail.json (N = 1) or
    bench_detail_n<N>.json; returns the path relative to the repository, or None when it cannot be written."""
    n = full.get('n_gpus', 1)
    rel = os.environ.get('BN_BENCH_DETAIL') or os.path.join(
        'gpurun_out', 'bench_detail.json' if n == 1 else 'bench_detail_n%d.json' % n)
    try:
        os.makedirs(os.path.dirname(os.path.join(REPO, rel)), exist_ok=True)
        with open(os.path.join(REPO, rel), 'w') as f:
            json.dump(full, f, indent=1)
        return rel
    except OSError:
        return None


def profile_kernel(model, opt, gen, family, C, K, steps=2):
    """HIP-event time of one kernel family/geometry over a few extra steps: per call (everything it
    launches, bracketed on the stream) and of its main kernel alone (events attached to the dispatch
    = the kernel's own begin / end timestamps, what rocprofv3 reports)."""
    _hip.prof_select(family, C, K)
    for _ in range(steps):
        one_step(model, opt, gen)
    torch.cuda.synchronize()
    ms, n, name = _hip.prof_read()
    main_ms, main_n = _hip.prof_read_main()
    _hip.prof_select(_hip.PROF_NONE)
    return ms, n, name, main_ms, main_n


def error_line(message, **extra):
    """The one JSON line of a run that failed: same metric name, no value, an ``error`` string."""
    out = {'metric': METRIC, 'value': None, 'unit': 'frames/s', 'error': str(message)[:1500]}
    out.update(extra)
    return json.dumps(out)


def first_to_report():
    """Several ranks may notice a failure: the first one to claim the job's marker file prints the line."""
    path = os.path.join('/tmp', 'bn_bench_error_%s_%s' % (os.environ.get('MASTER_PORT', '0'),
                                                          os.getppid()))
    try:
        os.close(os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY))
        return True
    except OSError:
        return False


def self_launch(n_gpus, limit_s):
    """`python bench.py --gpus N` without a launcher: run `torch.distributed.run` with N processes
    on this node (rendezvous on 127.0.0.1, a free port, same arguments) as a supervised child:
    its output is passed through, and if it ends without the JSON line (a rank died, a collective
    timed out and the launcher tore the job down, the whole thing exceeded ``limit_s``) ONE line
    with an ``error`` field is printed and the exit status is non-zero -- never a hang."""
    import socket
    import subprocess
    import threading
    if torch.cuda.device_count() < n_gpus and os.environ.get('BN_DIST_BACKEND') != 'gloo':
        print(error_line('--gpus %d but this node shows %d GPU(s)' % (n_gpus, torch.cuda.device_count()),
                         n_gpus=n_gpus))
        raise SystemExit(2)
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL between processes)
    from behavenet_amd.hostinfo import usable_cpus
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cpus() // n_gpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(n_gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    seen = {'line': False}

    def pump():
        for line in child.stdout:
            if line.startswith('{') and '"metric"' in line:
                seen['line'] = True
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = child.wait(timeout=limit_s)
        why = 'the launcher exited with status %d' % rc
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(child.pid, signal.SIGKILL)        # (its own session: the launcher and all ranks)
        child.wait()
        rc, why = 124, 'no result within %d s: job killed' % limit_s
    t.join(timeout=5)
    try:        # (the ranks' first-to-report marker: named after the port and the launcher's pid)
        os.remove(os.path.join('/tmp', 'bn_bench_error_%s_%s' % (port, child.pid)))
    except OSError:
        pass
    if not seen['line']:
        print(error_line('%d-rank run produced no result line (%s)' % (n_gpus, why), n_gpus=n_gpus))
        rc = rc or 1
    raise SystemExit(rc)


class Watchdog(object):
    """A rank that makes no progress for ``limit_s`` prints the error line and leaves (os._exit: a
    rank stuck inside a collective cannot raise).  ``pet()`` at every phase boundary / every few steps."""

    def __init__(self, limit_s, rank, world):
        import threading
        self.limit_s, self.rank, self.world = float(limit_s), rank, world
        self._last, self._where = time.monotonic(), 'start'
        self._stop = False
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def pet(self, where):
        self._last, self._where = time.monotonic(), where

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(1.0)
            idle = time.monotonic() - self._last
            if idle > self.limit_s and not self._stop:
                if first_to_report():
                    print(error_line('rank %d of %d made no progress for %.0f s in phase "%s" (a peer died or a '
                                     'collective hangs)' % (self.rank, self.world, idle, self._where),
                                     n_gpus=self.world, rank=self.rank), flush=True)
                os._exit(3)


def measure_allreduce(opt, iters=10):
    """The gradient exchange on its own: `iters` all-reduces of the flat gradient arena exactly as
    a step issues them (bucketed or flat), bracketed by barrier + synchronize; max over ranks."""
    import torch.distributed as dist
    keep = opt.flat_g.clone()
    reducer = getattr(opt, 'reducer', None)

    def once():
        if reducer is not None:
            reducer.begin()
        bdist.reduce_gradients(opt)
    once()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    dist.barrier()
    el = (time.perf_counter() - t0) / iters
    t = torch.tensor([el], dtype=torch.float64,
                     device='cpu' if dist.get_backend() == 'gloo' else 'cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    opt.flat_g.copy_(keep)
    return float(t.item()) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true',
                    help='skip the live rocprofv3 --pmc passes behind roofline.traffic')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary configs (PS-VAE, encode-only, PCIe-fed)')
    ap.add_argument('--feed', default='device', choices=['device', 'device_u8', 'host_u8', 'host'],
                    help="where the trials live (default 'device': resident float32, the headline "
                         "metric; 'host_u8' = pinned uint8 + prefetch, the PCIe-inclusive rate)")
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--full-line', action='store_true',
                    help='print the full result (kernel tables, per-layer rooflines: ~20 KB) on the line instead of '
                         'the compact one -- for tools/ab_*.sh; the driver-facing default stays below 4 KB')
    ap.add_argument('--shard-optimizer', dest='shard_optimizer', action='store_true', default=None,
                    help='N > 1: reduce-scatter -> Adam on 1/N of the arena per rank -> all-gather, instead '
                         'of the overlapped bucketed all-reduce + N identical steps (default: on for '
                         '--shard frames with N >= 4, fitting/distributed.py default_shard_optimizer)')
    ap.add_argument('--no-shard-optimizer', dest='shard_optimizer', action='store_false')
    ap.add_argument('--time-limit', type=float, default=float(os.environ.get('BN_BENCH_TIME_LIMIT_S', '1500')),
                    help='N > 1, self-launched: the whole job is killed (and an error line printed) after this many seconds')
    ap.add_argument('--shard', default='trial', choices=['trial', 'frames'],
                    help="N > 1: 'trial' = one 256-frame trial per rank per step (weak scaling), "
                         "'frames' = the ranks share ONE trial, each takes its slice of every "
                         "200-frame chunk (strong scaling, parity-exact)")
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False')
        self_launch(args.gpus, args.time_limit)          # does not return
    try:
        run(args)
    except SystemExit:
        raise
    except BaseException as err:                         # noqa: BLE001 (reported as the JSON line, then re-raised)
        import traceback
        traceback.print_exc()
        if int(os.environ.get('WORLD_SIZE', '1')) == 1 or first_to_report():
            print(error_line('%s: %s' % (type(err).__name__, err), n_gpus=args.gpus,
                             rank=int(os.environ.get('RANK', '0'))), flush=True)
        # (os._exit: a process group whose peer is gone can hang in its destructor)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)


def run(args):
    # the product's fit() hands loss dicts to its logger unresolved (hip_functions.set_lazy_losses): the host queues
    # step k + 1 without waiting for the forward pass of step k.  BN_BENCH_LAZY=0: the plain dict (one event wait per step)
    from behavenet_amd import hip_functions as _hf
    _hf.set_lazy_losses(os.environ.get('BN_BENCH_LAZY', '1') != '0')

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False')
    n_env = int(os.environ.get('WORLD_SIZE', '1'))
    gloo_test = os.environ.get('BN_DIST_BACKEND') == 'gloo'
    if n_env > 1:
        # fail in seconds, with a line, not in half an hour: short limits for the rendezvous and for
        # every collective of this short job (a user's BN_DIST_* settings win), errors of the
        # collective library raised instead of swallowed
        # (300 s for the rendezvous: on a cold node the ranks' first `import torch` takes minutes and need not end
        # together; collectives after that fail within 120 s)
        os.environ.setdefault('BN_DIST_RDZV_TIMEOUT_S', '300')
        os.environ.setdefault('BN_DIST_TIMEOUT_S', '120')
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if not gloo_test and torch.cuda.device_count() <= local:
            raise RuntimeError('LOCAL_RANK %d but this node shows %d GPU(s): --gpus %d needs one GPU per rank'
                               % (local, torch.cuda.device_count(), args.gpus))
    rank, world = bdist.init_from_env()
    if world != max(1, args.gpus):
        raise RuntimeError('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dog = Watchdog(float(os.environ.get('BN_BENCH_WATCHDOG_S', '150')), rank, world) if world > 1 else None
    if world > 1:
        # the launcher stops the surviving ranks with SIGTERM when one of them has died: the first of
        # them to get here says so in the line before it goes
        import signal

        def on_term(signum, frame):
            if first_to_report():
                print(error_line('rank %d of %d was stopped by the launcher (SIGTERM): another rank died or '
                                 'timed out' % (rank, world), n_gpus=world, rank=rank), flush=True)
            os._exit(1)
        signal.signal(signal.SIGTERM, on_term)

    def pet(where):
        if dog is not None:
            dog.pet(where)
    global _AVERAGE
    strong = world > 1 and args.shard == 'frames'
    if world > 1:
        bdist.set_shard_mode(args.shard)
        _AVERAGE = not strong
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if gloo_test:
        local = 0           # control-flow test of the N > 1 path: all ranks share the one GPU
    torch.cuda.set_device(local)
    _hip.load()
    devices = None
    if bdist.is_active():
        # one GPU per rank, and not the same one twice (the PCI bus id tells boards apart whatever
        # HIP_VISIBLE_DEVICES did to the ordinals)
        props = torch.cuda.get_device_properties(local)
        mine = {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device': local,
                'name': props.name, 'pci': getattr(props, 'pci_bus_id', None),
                'uuid': str(getattr(props, 'uuid', ''))}
        devices = [None] * world
        torch.distributed.all_gather_object(devices, mine)
        ids = [(d['device'], d['uuid'] or d['pci']) for d in devices]
        if not gloo_test and len(set(ids)) != world:
            raise RuntimeError('ranks share a GPU: %s' % devices)
    pet('model')

    hp = build_hparams()
    torch.manual_seed(hp['rng_seed_model'])
    model = AE(hp).to('cuda')
    shard_default = bdist.default_shard_optimizer(world, args.shard)
    shard_opt = (shard_default if args.shard_optimizer is None else args.shard_optimizer) and world > 1
    opt = FlatAdamAMSGrad(model.get_parameters(), lr=hp['learning_rate'],
                          weight_decay=hp['l2_reg'], shard_over=world if shard_opt else 1)
    bdist.broadcast_parameters_(opt.flat_p)
    if not shard_opt:
        bdist.attach_reducer(opt)
    # the step as fit() runs it: recorded into a HIP graph where graph_step.enabled_by_default() says so
    # (frame sharding over >= 4 ranks, or BN_GRAPH=1)
    global _LOSS
    from behavenet_amd.fitting import graph_step
    _LOSS = graph_step.GraphedLoss(model) if graph_step.enabled_by_default() else None

    # 20 trials x 256 frames per rank, trial_splits 8;1;1;0 -> 16 train trials
    # ('frames': every rank holds the same trials and walks them in the same order)
    data_rank = 0 if strong else rank
    sess = SyntheticSession(20, BATCH, DIM, seed=100 + data_rank, trial_splits='8;1;1;0')
    gen = SyntheticSessionsGenerator([sess], device='cuda', placement=args.feed)
    torch.manual_seed(1 + data_rank)
    np.random.seed(1 + data_rank)
    gen.reset_iterators('train')

    # Setup, not measurement: the HIP runtime grows internal pools (signals, kernarg chunks) once
    # after a few thousand dispatches -- a single 40-60 ms stall about 15 steps into a fresh
    # process (tools/spike_hunt.py, BN_BENCH_TRACE=1).  Prime it here so that it can fall neither
    # into the W warm-up steps' shadow nor into the K timed steps; the model state it touches is
    # the same training trajectory the warm-up continues.
    # Single process: in blocks of 8 steps until two blocks in a row run within 3 % of the best block
    # seen (at least PRIME_STEPS, at most PRIME_MAX steps -- half a second); N > 1: a fixed count, the
    # ranks' steps are collective.
    # BN_BENCH_PRIME=<count>: exactly that many (tests that compare the training trajectories of runs).
    fixed = os.environ.get('BN_BENCH_PRIME')
    primed = 0
    if fixed is not None or bdist.is_active():
        for _ in range(int(fixed) if fixed is not None else PRIME_STEPS + 16):
            one_step(model, opt, gen)
            primed += 1
            if primed % 8 == 0:
                pet('priming')
    else:
        best, streak, done = float('inf'), 0, 0
        while done < PRIME_MAX:
            torch.cuda.synchronize()
            t_p = time.perf_counter()
            for _ in range(8):
                one_step(model, opt, gen)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t_p) / 8
            done += 8
            streak = streak + 1 if dt <= best * 1.03 else 0
            best = min(best, dt)
            if done >= PRIME_STEPS and streak >= 2:
                break
        primed = done

    def barrier():
        if bdist.is_active():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Setup, N > 1 only: gradient all-reduce overlapped with the backward pass, or launched behind
    # it?  The overlapped form hides the transfer, but RCCL's blocks take CUs away from kernels
    # whose grids fill the chip exactly; which wins depends on RCCL's channel count on this node.
    # Measure both (all ranks agree on the max over ranks) and keep the faster one.
    allreduce_mode = None
    reducer = getattr(opt, 'reducer', None)
    pet('setup')
    extra_untimed = 0
    # N > 1: what ONE GPU of this job does on its own -- every rank runs the unsharded 256-frame step
    # without any exchange, all ranks at the same time (so the node's power / thermal state is the
    # job's), parameters and optimizer state put back afterwards.  This is the N = 1 figure the
    # scaling curve is measured against, taken on the same boxes in the same process.
    single_ref = None
    if bdist.is_active():
        keep = [t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.max_exp_avg_sq)]
        keep_count = opt.step_count
        prev_mode = bdist.set_shard_mode('trial')

        # (its own noise trial: the generator's position, hence the job's trajectory, is untouched)
        ref_data = {'images': torch.rand((1, BATCH) + tuple(DIM), device='cuda')}

        def local_step():
            opt.zero_grad()
            with bdist.emulate_rank(0, 1):          # collectives are identities in here
                model.loss(ref_data, dataset=0, accumulate_grad=True)
            opt.step()
        if reducer is not None:
            reducer.overlap = False
        for _ in range(5):
            local_step()
        barrier()
        t_l = time.perf_counter()
        for _ in range(10):
            local_step()
        torch.cuda.synchronize()
        mine_ms = (time.perf_counter() - t_l) / 10 * 1e3
        barrier()
        every = [None] * world
        torch.distributed.all_gather_object(every, round(mine_ms, 3))
        single_ref = {'ms_per_step_per_rank': every, 'ms_per_step_max': max(every),
                      'frames_per_s_one_gpu': round(BATCH / (max(every) * 1e-3), 1),
                      'what': 'the unsharded 256-frame step (no gradient exchange) run by every rank at the same '
                              'time, 10 steps after 5: the N = 1 reference of this node inside this job'}
        bdist.set_shard_mode(prev_mode)
        with torch.no_grad():
            for dst, src in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.max_exp_avg_sq), keep):
                dst.copy_(src)
        opt.step_count = keep_count
        del keep, ref_data
        if reducer is not None:
            reducer.overlap = True
        extra_untimed += 15
        pet('overlap probe')
    if bdist.is_active() and reducer is not None and os.environ.get('BN_OVERLAP_ALLREDUCE') is None:
        timing = {}
        for mode in (True, False):
            reducer.overlap = mode
            for _ in range(3):
                one_step(model, opt, gen)
            barrier()
            pet('overlap probe')
            t_a = time.perf_counter()
            for _ in range(8):
                one_step(model, opt, gen)
            barrier()
            extra_untimed += 11
            tt = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64,
                              device='cpu' if torch.distributed.get_backend() == 'gloo' else 'cuda')
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            timing[mode] = float(tt.item()) / 8
        reducer.overlap = timing[True] <= timing[False]
        allreduce_mode = {'chosen': 'overlapped with the backward pass' if reducer.overlap
                          else 'bucketed, launched behind the backward pass',
                          'ms_per_step_overlapped': round(timing[True] * 1e3, 3),
                          'ms_per_step_behind': round(timing[False] * 1e3, 3)}
    if bdist.is_active():
        # what the collective library saw, and the exchange timed on its own
        allreduce_mode = dict(allreduce_mode or {'chosen': (
            'sharded optimizer: reduce-scatter -> Adam on 1/N of the arena -> all-gather, behind the '
            'backward pass' if shard_opt else 'one flat all-reduce behind the backward pass')})
        allreduce_mode.update({
            'world_size': torch.distributed.get_world_size(),
            'backend': torch.distributed.get_backend() + (
                ' (RCCL)' if torch.distributed.get_backend() == 'nccl' else ''),
            'op': 'mean' if _AVERAGE else 'sum',
            'gradient_bytes': int(opt.flat_g.numel() * 4),
            'bucket_bytes': [int((hi - lo) * 4) for lo, hi, _ in reducer.buckets]
            if reducer is not None else [int(opt.flat_g.numel() * 4)],
            'allreduce_alone_ms': round(measure_allreduce(opt), 3),
            'shard_optimizer': bool(shard_opt),
            'shard_optimizer_decision': (
                'default for --shard %s at N = %d (fitting/distributed.py default_shard_optimizer): %s'
                % (args.shard, world, 'on' if shard_default else 'off')
                if args.shard_optimizer is None else 'command line: %s' % ('on' if shard_opt else 'off')),
            'devices': devices, 'single_gpu_reference': single_ref})
    pet('warm-up')
    for _ in range(args.warmup):
        one_step(model, opt, gen)
    # test hook (tests/test_gpu_sharding.py): BN_BENCH_FAULT='<rank>:<step>' -- that rank dies without
    # a word in front of that timed step; the job must end with an error line, not hang
    fault = os.environ.get('BN_BENCH_FAULT')
    fault = tuple(int(v) for v in fault.split(':')) if fault else None

    if os.environ.get('BN_BENCH_NOHOOK') != '1':
        _hip.prof_set_bracket(False)                    # dispatch-attached events only
        _hip.prof_select(_hip.PROF_CONV_FWD, 1, 32)     # enc.conv0 launches inside the timed region
    barrier()
    t0 = time.perf_counter()
    last = None
    trace = []
    if os.environ.get('BN_BENCH_TRACE') == '1':
        global _PARTS
        _PARTS = []
        ms0 = torch.cuda.memory_stats()
    for i_step in range(args.steps):
        if fault is not None and fault == (rank, i_step):
            os._exit(9)
        last = one_step(model, opt, gen)
        trace.append(time.perf_counter())
        if dog is not None and (i_step & 7) == 7:
            pet('timed steps')
    barrier()
    elapsed = time.perf_counter() - t0
    pet('report')
    if os.environ.get('BN_BENCH_TRACE') == '1' and rank == 0:
        print('host ms per step: ' + ' '.join(
            '%.2f' % ((b - a) * 1e3) for a, b in zip([t0] + trace[:-1], trace)) +
            ' | drain %.2f' % ((t0 + elapsed - trace[-1]) * 1e3), file=sys.stderr)
        ms1 = torch.cuda.memory_stats()
        for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_sync_all_streams',
                  'reserved_bytes.all.current'):
            print('  %s: %s -> %s' % (k, ms0.get(k), ms1.get(k)), file=sys.stderr)
        worst = max(range(len(_PARTS)), key=lambda i: sum(_PARTS[i]))
        print('slowest step %d: zero_grad %.2f next_batch %.2f loss %.2f allreduce+step %.2f' % (
            (worst,) + tuple(_PARTS[worst])), file=sys.stderr)
    _, _, conv0_name = _hip.prof_read()
    conv0_ms, conv0_n = _hip.prof_read_main()
    _hip.prof_select(_hip.PROF_NONE)
    _hip.prof_set_bracket(True)

    if bdist.is_active():
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cpu' if torch.distributed.get_backend() == 'gloo' else 'cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = BATCH * (1 if strong else world) * args.steps
    value = frames / elapsed

    # enc.conv0 roofline: one launch per step over the whole 256-frame batch
    conv0_frames = BATCH
    if strong:      # rank 0's slice of the two chunks
        conv0_frames = sum(e - b for b, e in bdist.shard_chunks(BATCH, 200)[1])
    # (per launch the hook SAW: under HIP-graph replay -- frame sharding over >= 4 ranks -- the dispatch events are
    # attached to the eager launches only, fewer than K; round 5 multiplied by K there and overstated `achieved`)
    conv0_bytes = CONV0_BYTES_PER_FRAME * conv0_frames * max(conv0_n, 0)
    achieved = conv0_bytes / (conv0_ms * 1e-3) / 1e9 if conv0_ms > 0 else 0.0
    traffic = None
    traffic_source = None
    # (not in the reduced runs of the tools / tests, which may themselves sit under rocprofv3)
    if rank == 0 and world == 1 and not args.no_pmc and not args.no_secondary:
        traffic, info = live_hbm_traffic()
        if traffic is not None:
            traffic_source = ('measured in this run: two child processes `rocprofv3 --kernel-trace --pmc '
                              'FETCH_SIZE` / `--pmc WRITE_SIZE` over tools/run_layer.py (enc.conv0 forward, '
                              '%d frames per launch, mean of %d launches after the first): 2 x %.1f KB + '
                              '%.1f KB' % (BATCH, info['launches'] - 1, info['FETCH_SIZE_KB'],
                                           info['WRITE_SIZE_KB']))
        else:
            traffic_source = 'live PMC pass unavailable (%s); ' % info
    if traffic is None:
        tr_path = os.path.join(REPO, 'profiles', 'conv0_hbm_traffic.json')
        if os.path.exists(tr_path):
            with open(tr_path) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch_avg')
        traffic_source = (traffic_source or '') + (
            'profiles/conv0_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over '
            'tools/run_layer.py (tools/pmc_hbm.sh), same kernel, 256 frames per launch')
    roofline = {
        'bound': 'hbm', 'kernel': conv0_name, 'layer': 'enc.conv0 fwd (1->32, k5 s2, +bias+lrelu)',
        'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 4),
        'launches': conv0_n, 'avg_launch_us': round(conv0_ms * 1e3 / max(conv0_n, 1), 2),
        'algorithmic_bytes_per_launch_avg': int(conv0_bytes // max(conv0_n, 1)),
        'traffic': traffic,
        'traffic_source': traffic_source,
        # what the same dispatch-attached events read around an EMPTY kernel: avg_launch_us is the
        # raw interval (not corrected); rocprofv3's kernel timestamps come out ~1.5-2 us lower
        'event_interval_of_empty_kernel_us': round(float(_hip.load().bn_prof_dispatch_overhead_us(
            50, torch.cuda.current_stream().cuda_stream)), 2)}

    out = {
        'metric': METRIC,
        'value': round(value, 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
        # untimed steps in front of the W warm-up steps (setup: runtime priming in blocks of 8 until the step
        # time is steady, and for N > 1 the single-GPU reference and the overlapped-vs-behind probe)
        'priming_steps': primed, 'other_untimed_steps': extra_untimed,
        'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload_short': 'configs[1]: conv AE default arch, 1x128x128 f32 frames resident in HBM, 12 latents, '
                                     'batch 256 (chunks 200+56), fwd+bwd+Adam(amsgrad)',
                   'workload': 'configs[1]: conv AE (default arch 32-64-128-256-512, k5, strides '
                               '2,2,2,2,5), 1x128x128 uint8-noise frames as float32/255, 12 '
                               'latents, one 256-frame trial per step per GPU (the reference\'s 200+56 '
                               'chunk loss normalisation; one forward/backward pass), '
                               'Adam(amsgrad) lr 1e-4',
                   'frames_per_step_per_gpu': BATCH // world if strong else BATCH,
                   'global_frames_per_step': BATCH if strong else BATCH * world,
                   'sharding': ('single GPU' if world == 1 else
                                'frames: the ranks share one 256-frame trial per step, rank r takes '
                                'frames [r n_c/R, (r+1) n_c/R) of each 200-frame chunk, chunk terms '
                                'normalised globally, all-reduce(sum) of the flat 35 MB gradient'
                                if strong else
                                'trial: one 256-frame trial per rank per step, all-reduce(mean) of '
                                'the flat 35 MB gradient'),
                   'inputs': {'device': 'resident in HBM (float32)',
                              'device_u8': 'resident in HBM (uint8, converted per batch)',
                              'host_u8': 'pinned host uint8, prefetched over PCIe per batch',
                              'host': 'pinned host float32, copied per batch'}[args.feed]},
        'allreduce': allreduce_mode,
        'hip_graph': bool(_LOSS is not None and _LOSS.n_replays > 0),
        'hip_graph_ranks_recorded': getattr(_LOSS, 'peers_recorded', None),
        'final_loss': last['loss'] if last else None,
        'whole_step_fp32_tflops_per_gpu': round(TRAIN_FLOP_PER_FRAME * value / world / 1e12, 2),
        'whole_step_fp32_frac': round(TRAIN_FLOP_PER_FRAME * value / world / 1e12 /
                                      FP32_PEAK_TFLOPS, 4),
        'roofline': roofline,
    }

    if rank == 0 and world == 1:
        # FLOP-bound middle layers, profiled over two extra (untimed) steps each
        extra = []
        # (C, K) as the dispatch reports them: gather-down / weight-gradient launches (big,
        # small) channels, gather-up launches (small, big) -- csrc/capi.hip run_down / run_up
        E4 = 26214400.0 * 16 / 25       # stride-5 layers: 16 of 25 taps ever meet data
        for label, fam, C, K, flop_per_frame in [
                ('enc.conv1 fwd', _hip.PROF_CONV_FWD, 32, 64, 104857600.0),
                ('enc.conv2 fwd', _hip.PROF_CONV_FWD, 64, 128, 104857600.0),
                ('enc.conv3 fwd', _hip.PROF_CONV_FWD, 128, 256, 104857600.0),
                ('enc.conv4 fwd (stride 5; executed FLOPs)', _hip.PROF_CONV_FWD, 256, 512, E4),
                ('enc.conv1 bwd-weight', _hip.PROF_CONV_BWD_W, 32, 64, 104857600.0),
                ('enc.conv3 bwd-weight', _hip.PROF_CONV_BWD_W, 128, 256, 104857600.0),
                ('enc.conv2 bwd-weight', _hip.PROF_CONV_BWD_W, 64, 128, 104857600.0),
                ('enc.conv1 bwd-data', _hip.PROF_CONV_BWD_D, 64, 32, 104857600.0),
                ('enc.conv2 bwd-data', _hip.PROF_CONV_BWD_D, 128, 64, 104857600.0),
                ('enc.conv3 bwd-data', _hip.PROF_CONV_BWD_D, 256, 128, 104857600.0),
                ('dec.convT1 fwd', _hip.PROF_CONVT_FWD, 256, 128, 104857600.0),
                ('dec.convT2 fwd', _hip.PROF_CONVT_FWD, 128, 64, 104857600.0),
                ('dec.convT3 fwd', _hip.PROF_CONVT_FWD, 64, 32, 104857600.0),
                ('dec.convT0 fwd (stride 5; executed FLOPs)', _hip.PROF_CONVT_FWD, 512, 256, E4),
                # the true dense GEMMs of the path (north_star: MFMA utilisation of the linear
                # latent projection): enc.FF 2048 -> 12 and dec.FF 12 -> 2048, 256 rows
                ('enc.FF fwd (Linear 2048->12)', _hip.PROF_LINEAR_FWD, 2048, N_LATENTS,
                 2.0 * 2048 * N_LATENTS),
                ('dec.FF fwd (Linear 12->2048)', _hip.PROF_LINEAR_FWD, N_LATENTS, 2048,
                 2.0 * 2048 * N_LATENTS),
                ('enc.FF bwd (dx + dW + db)', _hip.PROF_LINEAR_BWD, 2048, N_LATENTS,
                 4.0 * 2048 * N_LATENTS),
                ('dec.FF bwd (dx + dW + db)', _hip.PROF_LINEAR_BWD, N_LATENTS, 2048,
                 4.0 * 2048 * N_LATENTS)]:
            ms, n, name, kms, kn = profile_kernel(model, opt, gen, fam, C, K, steps=4)
            if n:
                # the main kernel alone when its launcher attaches the events, else the whole call
                t_ms, t_n = (kms, kn) if kn else (ms, n)
                tf = flop_per_frame * BATCH * t_n / (t_ms * 1e-3) / 1e12
                extra.append({'layer': label, 'kernel': name, 'bound': 'mfma',
                              'achieved': round(tf, 3), 'peak': FP32_PEAK_TFLOPS,
                              'unit': 'TFLOP/s', 'frac': round(tf / FP32_PEAK_TFLOPS, 5),
                              'launches': t_n, 'avg_launch_us': round(t_ms * 1e3 / t_n, 1),
                              'timed': 'main kernel (dispatch events)' if kn else 'whole call (stream events)',
                              'whole_call_us': round(ms * 1e3 / n, 1)})
            else:
                extra.append({'layer': label, 'error': 'no launch matched (%d, %d, %d)' % (fam, C, K)})
        # the two other HBM-bound edge kernels of the step, same event method as `roofline`
        D4_LOSS_BYTES = 524288 + 65536 + 65536          # read h (32x64x64), read target, write dpre
        for label, fam, C, K, bytes_per_frame in [
                ('dec.convT4 fwd + Sigmoid + pixel loss + dL/dpre', _hip.PROF_CONVT_FWD, 32, 1,
                 D4_LOSS_BYTES),
                ('enc.conv0 bwd-weight', _hip.PROF_CONV_BWD_W, 1, 32, CONV0_BYTES_PER_FRAME),
                ('dec.convT4 bwd-weight', _hip.PROF_CONVT_BWD_W, 1, 32, CONV0_BYTES_PER_FRAME),
                ('dec.convT4 bwd-data', _hip.PROF_CONVT_BWD_D, 1, 32, 65536 + 2 * 524288)]:
            ms, n, name, kms, kn = profile_kernel(model, opt, gen, fam, C, K, steps=4)
            if n:
                t_ms, t_n = (kms, kn) if kn else (ms, n)
                gbs = bytes_per_frame * BATCH * t_n / (t_ms * 1e-3) / 1e9
                extra.append({'layer': label, 'kernel': name, 'bound': 'hbm',
                              'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(gbs / HBM_PEAK_GBS, 4), 'launches': t_n,
                              'avg_launch_us': round(t_ms * 1e3 / t_n, 1),
                              'timed': 'main kernel (dispatch events)' if kn else 'whole call (stream events)',
                              'whole_call_us': round(ms * 1e3 / n, 1)})
            else:
                extra.append({'layer': label, 'error': 'no launch matched (%d, %d, %d)' % (fam, C, K)})
        out['roofline_other_kernels'] = extra
        if not args.no_secondary:
            del model, opt, gen
            torch.cuda.empty_cache()
            try:
                out['secondary'] = secondary_configs(hp)
            except Exception as err:                             # noqa: BLE001 (the headline line still goes out)
                import traceback
                traceback.print_exc()
                out['secondary'] = [{'id': 'secondary', 'config': 'secondary configurations', 'value': None,
                                     'error': '%s: %s' % (type(err).__name__, str(err)[:500])}]
        if not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(hp, args.cpu_budget)
                out['speedup_vs_cpu_baseline'] = round(value / out['cpu_baseline']['value'], 1)
            except Exception as err:                             # noqa: BLE001
                import traceback
                traceback.print_exc()
                out['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': 0, 'kind': 'port',
                                       'sample': 'failed: %s: %s' % (type(err).__name__, str(err)[:300])}

    if rank == 0:
        out['detail_file'] = write_detail(out)
        sys.stderr.flush()
        print(json.dumps(out) if args.full_line else compact_line(out), flush=True)
    if dog is not None:
        dog.stop()
    if bdist.is_active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
