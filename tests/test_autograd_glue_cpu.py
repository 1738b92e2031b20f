"""The pure-torch autograd helpers of the VAE loss glue (no HIP kernels involved) against plain
torch expressions: values and gradients."""
import torch

from behavenet_amd import hip_functions as hf


def test_split_cols_matches_slicing():
    torch.manual_seed(0)
    t = torch.randn(7, 5, dtype=torch.float64, requires_grad=True)
    a, b = hf.split_cols(t, 2)
    assert a.is_contiguous() and b.is_contiguous()
    (a.pow(2).sum() + 3.0 * b.sum()).backward()
    g = t.grad.clone()
    t.grad = None
    (t[:, :2].pow(2).sum() + 3.0 * t[:, 2:].sum()).backward()
    assert torch.equal(g, t.grad)
    # one of the two parts unused: its gradient is zero
    t.grad = None
    a, _ = hf.split_cols(t, 2)
    a.sum().backward()
    assert torch.equal(t.grad[:, :2], torch.ones(7, 2, dtype=torch.float64))
    assert torch.count_nonzero(t.grad[:, 2:]) == 0


def test_combine_chunk_terms_matches_the_written_out_sum():
    torch.manual_seed(1)
    ll_x = torch.randn(2, requires_grad=True)
    ll_y = torch.randn(2, requires_grad=True)
    zs = torch.randn(2, requires_grad=True)
    dk = torch.randn(2, 3, requires_grad=True)
    alpha, kl, beta = 1000.0, 0.3, 5.0
    total, terms = hf.combine_chunk_terms([ll_x, ll_y, zs, dk],
                                          [[-1.0], [-alpha], [1.0], [kl, beta, kl]])
    want = -ll_x - alpha * ll_y + zs + kl * dk[:, 0] + beta * dk[:, 1] + kl * dk[:, 2]
    assert torch.allclose(total, want, rtol=1e-6, atol=1e-6)
    assert not terms.requires_grad
    assert torch.equal(terms, torch.cat([ll_x[:, None], ll_y[:, None], zs[:, None], dk], 1).detach())
    total.backward(torch.ones(2))
    got = [t.grad.clone() for t in (ll_x, ll_y, zs, dk)]
    for t in (ll_x, ll_y, zs, dk):
        t.grad = None
    want.backward(torch.ones(2))
    for g, t in zip(got, (ll_x, ll_y, zs, dk)):
        assert torch.allclose(g, t.grad, rtol=1e-6, atol=1e-6)


def test_frame_shards_refuse_on_every_rank_when_one_rank_has_no_frame():
    """a chunk with fewer frames than ranks leaves some rank's packed batch empty; the
    variational losses then must not run zero-row kernels / gathers on that rank alone -- ALL
    ranks refuse together (same exception everywhere, nobody is left waiting in a collective)."""
    import pytest
    from behavenet_amd.fitting import distributed as bdist
    from behavenet_amd.models.vaes import _FrameShards
    for r in range(8):
        with bdist.emulate_rank(r, 8):
            with pytest.raises(NotImplementedError, match='without a frame'):
                _FrameShards(4, 200)
            sh = _FrameShards(210, 200)        # 200 + 10 frames over 8 ranks: everyone has some
            assert sh.sharded and sh.n_local in (26, 27)
    assert not _FrameShards(4, 200).sharded    # no process group, no emulation: plain chunks
