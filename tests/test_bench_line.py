"""bench.py's result line: what the driver parses must stay a small, single JSON line (round 5 lost its
measurement to a 20 KB line: BENCH_r05.json `parsed: null`).  The fixture is round 5's full record."""
import copy
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


@pytest.fixture(scope='module')
def bench():
    import bench as mod
    return mod


@pytest.fixture()
def full():
    with open(os.path.join(REPO, 'tests', 'golden', 'bench_full_record.json')) as f:
        return json.load(f)


def test_compact_line_is_small_and_complete(bench, full):
    assert len(json.dumps(full)) > 15000                      # (the record that did not parse)
    line = bench.compact_line(full)
    assert '\n' not in line and len(line) < 4096
    d = json.loads(line)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['metric'] == bench.METRIC and d['value'] == full['value'] and d['ms_per_step'] == full['ms_per_step']
    assert isinstance(d['config']['workload'], str) and len(d['config']['workload']) <= 160
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], abs=1e-3)
    assert r['avg_launch_us'] > 0 and r['algorithmic_bytes_per_launch_avg'] == 256 * 589824
    assert 'traffic' in r
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == full['cpu_baseline']['cores'] and len(c['sample']) <= 100
    assert c['value'] == pytest.approx(full['cpu_baseline']['value'], rel=1e-3)
    sec = {s['id']: s for s in d['secondary']}
    assert len(sec) == len(full['secondary'])
    assert sec['batch_norm']['ms_per_step'] == 5.25 and sec['batch_norm']['frac'] == pytest.approx(0.6457, abs=1e-3)
    assert set(sec['maxpool']) <= {'id', 'value', 'ms_per_step', 'frac'}


def test_compact_line_survives_growth(bench, full):
    """Whatever a later round adds to the record, the line stays below the limit and keeps the headline keys."""
    big = copy.deepcopy(full)
    big['secondary'] = big['secondary'] * 12
    for s in big['secondary']:
        s['error'] = 'x' * 500
    big['config']['workload'] = 'w' * 5000
    big['cpu_baseline']['sample'] = 's' * 5000
    line = bench.compact_line(big)
    assert len(line) < 4096
    d = json.loads(line)
    assert d['roofline']['frac'] == full['roofline']['frac'] and d['cpu_baseline']['value'] > 0


def test_compact_line_multi_rank_and_error(bench, full):
    multi = copy.deepcopy(full)
    multi.update({'n_gpus': 8, 'scaling': 'weak'})
    multi.pop('secondary')
    multi.pop('cpu_baseline')
    multi['allreduce'] = {'chosen': 'overlapped with the backward pass', 'world_size': 8, 'backend': 'nccl (RCCL)',
                          'op': 'mean', 'gradient_bytes': 35033600, 'bucket_bytes': [1] * 40,
                          'allreduce_alone_ms': 0.5, 'shard_optimizer': False,
                          'devices': [{'rank': r, 'name': 'AMD Instinct MI355X', 'pci': '0000:%02x:00.0' % r,
                                       'uuid': 'GPU-%032x' % r} for r in range(8)],
                          'single_gpu_reference': {'ms_per_step_per_rank': [4.4] * 8, 'ms_per_step_max': 4.4,
                                                   'what': 'y' * 300}}
    line = bench.compact_line(multi)
    assert len(line) < 4096
    d = json.loads(line)
    assert d['n_gpus'] == 8 and d['allreduce']['world_size'] == 8 and d['allreduce']['allreduce_alone_ms'] == 0.5
    assert 'devices' not in d['allreduce'] and d['allreduce']['single_gpu_ms_per_step_max'] == 4.4
    err = json.loads(bench.error_line('boom ' * 2000, n_gpus=2))
    assert err['value'] is None and err['metric'] == bench.METRIC
    assert len(bench.error_line('boom ' * 2000, n_gpus=2)) < 4096


def test_write_detail_roundtrip(bench, full, tmp_path, monkeypatch):
    monkeypatch.setenv('BN_BENCH_DETAIL', str(tmp_path / 'detail.json'))
    rel = bench.write_detail(full)
    with open(os.path.join(REPO, rel)) as f:
        assert json.load(f) == full
