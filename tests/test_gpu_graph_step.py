"""The training step recorded into a HIP graph (behavenet_amd/fitting/graph_step.py) against the
same step on eager launches: same kernels, same order, same operands -- losses, gradients and the
parameters after several optimizer steps must be BIT-identical, for fresh data in every step."""
import numpy as np
import pytest
import torch

from behavenet_amd.fitting import distributed as bdist
from behavenet_amd.fitting.graph_step import GraphedLoss, LazyLoss
from behavenet_amd.fitting.optim import FlatAdamAMSGrad
from behavenet_amd.fitting.training import fit
from behavenet_amd.data.data_generator import SyntheticSession, SyntheticSessionsGenerator
from behavenet_amd.models import AE, ConditionalAE, VAE
from behavenet_amd.models.ae_model_architecture_generator import load_handcrafted_arch
from tests.golden_utils import base_hparams, make_frames, make_labels

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _hparams(dim, model_class='ae', extra=None):
    arch = load_handcrafted_arch(list(dim), 8, None, check_memory=False)
    hp = base_hparams(arch, model_class, extra or {})
    hp['device'] = DEV
    return hp


def _run(cls, hp, batches, graphed, shard=None, n_steps=None):
    np.random.seed(0)
    torch.manual_seed(0)
    model = cls(hp).to(DEV)
    opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-3)
    fn = GraphedLoss(model, warmup=1) if graphed else model.loss
    losses, grads = [], []
    for i, data in enumerate(batches):
        model.train()
        opt.zero_grad()
        if shard is not None:
            with bdist.emulate_rank(*shard):
                out = fn(data, dataset=0, accumulate_grad=True)
        else:
            out = fn(data, dataset=0, accumulate_grad=True)
        losses.append(dict(out))
        grads.append(opt.flat_g.clone())
        opt.step()
    # validation-style call: no gradients, eval mode
    model.eval()
    val = [dict(fn(batches[j], dataset=0, accumulate_grad=False)) for j in (0, 1, 2)]
    return losses, grads, opt.flat_p.clone(), val, fn


@pytest.mark.parametrize('dim, batch', [((1, 64, 48), 210), ((1, 32, 32), 24), ((1, 128, 128), 64)])
def test_graphed_step_is_bit_identical_to_eager(dim, batch):
    hp = _hparams(dim)
    batches = [{'images': [torch.from_numpy(make_frames(batch, list(dim), seed=10 + i)).to(DEV)]}
               for i in range(6)]
    l_e, g_e, p_e, v_e, _ = _run(AE, hp, batches, False)
    l_g, g_g, p_g, v_g, fn = _run(AE, hp, batches, True)
    assert fn.n_replays == 6 - 1 + 3 - 1 and fn.n_eager == 2      # one eager call per signature
    assert l_e == l_g
    assert v_e == v_g
    for a, b in zip(g_e, g_g):
        assert torch.equal(a, b)
    assert torch.equal(p_e, p_g)


def test_graphed_step_with_batch_norm_statistics_per_chunk():
    """Batch norm inside the recorded step: statistics per 200-frame chunk, the running estimates
    and the batch counter advance with every replay exactly as on eager launches."""
    dim = (1, 64, 48)
    hp = _hparams(dim, 'ae', {'ae_batch_norm': True, 'ae_batch_norm_momentum': 0.1})
    batches = [{'images': [torch.from_numpy(make_frames(210, list(dim), seed=50 + i)).to(DEV)]}
               for i in range(5)]

    def run(graphed):
        np.random.seed(0)
        torch.manual_seed(0)
        model = AE(hp).to(DEV)
        opt = FlatAdamAMSGrad(model.get_parameters(), lr=1e-3)
        fn = GraphedLoss(model, warmup=1) if graphed else model.loss
        losses = []
        for data in batches:
            model.train()
            opt.zero_grad()
            losses.append(dict(fn(data, dataset=0, accumulate_grad=True)))
            opt.step()
        bufs = {k: v.clone() for k, v in model.named_buffers()}
        return losses, opt.flat_p.clone(), bufs, fn
    l_e, p_e, b_e, _ = run(False)
    l_g, p_g, b_g, fn = run(True)
    assert fn.n_replays == 4 and fn.n_eager == 1
    assert l_e == l_g
    assert torch.equal(p_e, p_g)
    for k in b_e:
        assert torch.equal(b_e[k], b_g[k]), k


def test_graphed_step_conditional_ae_with_labels():
    dim = (1, 64, 48)
    hp = _hparams(dim, 'cond-ae', {'n_labels': 3, 'conditional_encoder': False})
    batches = [{'images': [torch.from_numpy(make_frames(40, list(dim), seed=20 + i)).to(DEV)],
                'labels': [torch.from_numpy(make_labels(40, 3, seed=30 + i)).to(DEV)]}
               for i in range(4)]
    l_e, g_e, p_e, v_e, _ = _run(ConditionalAE, hp, batches, False)
    l_g, g_g, p_g, v_g, fn = _run(ConditionalAE, hp, batches, True)
    assert fn.n_replays > 0
    assert l_e == l_g and v_e == v_g
    assert torch.equal(p_e, p_g)


def test_graphed_step_under_emulated_frame_sharding():
    dim = (1, 64, 48)
    hp = _hparams(dim)
    batches = [{'images': [torch.from_numpy(make_frames(210, list(dim), seed=40 + i)).to(DEV)]}
               for i in range(4)]
    prev = bdist.set_shard_mode('frames')
    try:
        l_e, g_e, p_e, _, _ = _run(AE, hp, batches, False, shard=(1, 4))
        l_g, g_g, p_g, _, fn = _run(AE, hp, batches, True, shard=(1, 4))
    finally:
        bdist.set_shard_mode(prev)
    assert fn.n_replays > 0
    assert l_e == l_g
    for a, b in zip(g_e, g_g):
        assert torch.equal(a, b)


def test_classes_without_a_deferred_tail_stay_eager():
    dim = (1, 32, 32)
    hp = _hparams(dim, 'vae', {'vae.beta': 2.0, 'vae.beta_anneal_epochs': 0, 'max_n_epochs': 5})
    model = VAE(hp).to(DEV)
    model.curr_epoch = 1
    FlatAdamAMSGrad(model.get_parameters(), lr=1e-3)
    fn = GraphedLoss(model, warmup=0)
    data = {'images': [torch.from_numpy(make_frames(16, list(dim), seed=1)).to(DEV)]}
    for _ in range(3):
        out = fn(data, dataset=0, accumulate_grad=True)
        assert isinstance(out, dict) and not isinstance(out, LazyLoss)
    assert fn.n_replays == 0 and fn.n_eager == 3


class _Exp(object):
    def __init__(self):
        self.rows, self.version = [], 0

    def log(self, row):
        self.rows.append(dict(row))

    def save(self):
        pass


def test_fit_rows_with_and_without_graphs(tmp_path):
    dim = (1, 64, 48)
    rows, params = [], []
    for use_graph in (False, True):
        hp = _hparams(dim)
        d = tmp_path / ('g%d' % use_graph)
        (d / 'version_0').mkdir(parents=True)
        hp.update({'max_n_epochs': 3, 'min_n_epochs': 3, 'enable_early_stop': False,
                   'val_check_interval': 1, 'expt_dir': str(d), 'version': 0,
                   'rng_seed_train': 0, 'export_latents': False, 'early_stop_history': 10,
                   'learning_rate': 1e-3, 'progress_bar': False, 'hip_graph': use_graph})
        torch.manual_seed(0)
        model = AE(hp).to(DEV)
        model.version = 0
        sess = SyntheticSession(10, 50, list(dim), seed=5, trial_splits='6;2;2;0')
        gen = SyntheticSessionsGenerator([sess], device=DEV, placement='device')
        exp = _Exp()
        fit(hp, model, gen, exp, method='ae')
        rows.append(exp.rows)
        params.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone())
    assert len(rows[0]) == len(rows[1]) > 0
    for a, b in zip(rows[0], rows[1]):
        assert a == b
    assert torch.equal(params[0], params[1])
