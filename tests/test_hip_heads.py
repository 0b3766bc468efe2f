"""GPU: the row-work loss heads of csrc/ssac_markov.hip and the exploration / deterministic-policy helpers against plain
PyTorch fp32 autograd of the reference's expressions (learning.py:296-311, learning_utils.py:48-60, 272-285, 401-409,
distributions.py:74-114), at ragged sizes (one row, sizes that are not multiples of the wave or workgroup width)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ssa():
    import super_sac_amd
    return super_sac_amd


def _st(ssa):
    return ssa.engine.stream()


@pytest.mark.parametrize("n_pos,n", [(1, 2), (33, 66), (500, 1000), (7, 20)])
def test_bce_sigmoid_head(ssa, n_pos, n):
    g = torch.Generator().manual_seed(n)
    z = (torch.randn(n, generator=g) * 3).requires_grad_(True)
    labels = torch.cat((torch.ones(n_pos), torch.zeros(n - n_pos)))
    loss = F.binary_cross_entropy(torch.sigmoid(z), labels)
    loss.backward()
    zd, dz, out = z.detach().to(DEV), torch.empty(n, device=DEV), torch.zeros(1, device=DEV)
    ssa._lib.check(ssa._lib.lib.ssac_bce_sigmoid_bwd(zd.data_ptr(), n_pos, n, 0.7, dz.data_ptr(), out.data_ptr(), _st(ssa)))
    assert abs(float(out) - float(loss)) <= 2e-6 * max(1.0, abs(float(loss)))
    assert torch.allclose(dz.cpu(), 0.7 * z.grad, rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("B,D,max_dist", [(1, 5, 0.0), (17, 50, 0.01), (300, 64, 0.5), (64, 130, 0.0)])
def test_smoothness_head(ssa, B, D, max_dist):
    g = torch.Generator().manual_seed(B + D)
    s = torch.randn(B, D, generator=g).requires_grad_(True)
    s1 = (s.detach() + 0.3 * torch.randn(B, D, generator=g)).requires_grad_(True)
    if B > 2:
        with torch.no_grad():
            s1[1] = s[1]   # a zero distance row: torch.norm's backward gives 0 there
    dist = torch.norm(s1 - s, dim=-1, p=2) / math.sqrt(D)
    loss = F.relu(dist - max_dist).square().mean()
    loss.backward()
    sd, s1d = s.detach().to(DEV), s1.detach().to(DEV)
    ds, ds1, out = torch.full((B, D), 9.0, device=DEV), torch.full((B, D), 9.0, device=DEV), torch.zeros(1, device=DEV)
    lib = ssa._lib.lib
    ssa._lib.check(lib.ssac_markov_smoothness_bwd(sd.data_ptr(), D, s1d.data_ptr(), D, B, D, max_dist, 2.0, ds.data_ptr(), D,
                                                  ds1.data_ptr(), D, 0, out.data_ptr(), _st(ssa)))
    assert abs(float(out) - float(loss)) <= 5e-6 * max(1.0, abs(float(loss)))
    assert torch.allclose(ds1.cpu(), 2.0 * s1.grad, rtol=1e-4, atol=1e-7) and torch.allclose(ds.cpu(), 2.0 * s.grad, rtol=1e-4, atol=1e-7)
    # accumulate mode adds to what is there
    ssa._lib.check(lib.ssac_markov_smoothness_bwd(sd.data_ptr(), D, s1d.data_ptr(), D, B, D, max_dist, 2.0, 0, D,
                                                  ds1.data_ptr(), D, 1, out.data_ptr(), _st(ssa)))
    assert torch.allclose(ds1.cpu(), 4.0 * s1.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("B,D", [(1, 3), (8, 50), (513, 128)])
def test_frobenius_difference_head(ssa, B, D):
    g = torch.Generator().manual_seed(B)
    a = torch.randn(B, D, generator=g).requires_grad_(True)
    b = torch.randn(B, D, generator=g)
    loss = torch.norm(a - b)
    loss.backward()
    da, out, tot = torch.zeros(B, D, device=DEV), torch.zeros(1, device=DEV), torch.full((1,), 5.0, device=DEV)
    ad, bd = a.detach().to(DEV), b.to(DEV)   # (named: a temporary's data_ptr would dangle)
    ssa._lib.check(ssa._lib.lib.ssac_frobenius_diff_bwd(ad.data_ptr(), D, bd.data_ptr(), D, B, D, 0.25,
                                                        da.data_ptr(), D, 0, out.data_ptr(), tot.data_ptr(), _st(ssa)))
    assert abs(float(out) - float(loss)) <= 1e-5 * float(loss)
    assert abs(float(tot) - (5.0 + 0.25 * float(loss))) <= 1e-5 * (5.0 + float(loss))
    assert torch.allclose(da.cpu(), 0.25 * a.grad, rtol=1e-4, atol=1e-8)
    # identical inputs: norm 0, gradient 0 (torch.norm's backward), not NaN
    ssa._lib.check(ssa._lib.lib.ssac_frobenius_diff_bwd(bd.data_ptr(), D, bd.data_ptr(), D, B, D, 0.25,
                                                        da.data_ptr(), D, 0, out.data_ptr(), 0, _st(ssa)))
    assert float(out) == 0.0 and float(da.abs().sum()) == 0.0


def _tanh_normal_logp_data(out, lo, hi, a):
    mu, raw = out.chunk(2, dim=-1)
    log_std = lo + 0.5 * (hi - lo) * (torch.tanh(raw) + 1.0)
    std = log_std.exp()
    y = a.clamp(-0.99, 0.99)
    x = 0.5 * (torch.log1p(y) - torch.log1p(-y))
    base = -((x - mu) ** 2) / (2.0 * std * std) - log_std - 0.5 * math.log(2 * math.pi)
    return (base - 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))).sum(-1, keepdim=True)


@pytest.mark.parametrize("B,A", [(1, 1), (37, 6), (1030, 17)])
def test_action_invariance_head_continuous(ssa, B, A):
    g = torch.Generator().manual_seed(A)
    out_a = torch.randn(B, 2 * A, generator=g).requires_grad_(True)
    act = torch.tanh(torch.randn(B, A, generator=g) * 1.5)   # some beyond the +-0.99 clamp
    olp = torch.randn(B, 1, generator=g)
    loss = F.mse_loss(olp, _tanh_normal_logp_data(out_a, -5.0, 2.0, act))
    loss.backward()
    d, lo_, tot = torch.zeros(B, 2 * A, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    od, actd, olpd = out_a.detach().to(DEV), act.to(DEV), olp.reshape(-1).to(DEV)
    ssa._lib.check(ssa._lib.lib.ssac_action_invariance_bwd(
        od.data_ptr(), 2 * A, actd.data_ptr(), A, olpd.data_ptr(), B, A, -5.0,
        2.0, 0.5, d.data_ptr(), 2 * A, lo_.data_ptr(), tot.data_ptr(), _st(ssa)))
    assert abs(float(lo_) - float(loss)) <= 2e-5 * max(1.0, abs(float(loss)))
    assert abs(float(tot) - 0.5 * float(loss)) <= 2e-5 * max(1.0, abs(float(loss)))
    assert torch.allclose(d.cpu(), 0.5 * out_a.grad, rtol=5e-4, atol=1e-6 * float(out_a.grad.abs().max()))


@pytest.mark.parametrize("B,A", [(1, 2), (50, 5), (999, 18)])
def test_action_invariance_head_discrete(ssa, B, A):
    g = torch.Generator().manual_seed(B)
    lo = torch.randn(B, A, generator=g)
    la = torch.randn(B, A, generator=g).requires_grad_(True)
    act = torch.randint(0, A, (B,), generator=g)
    olp = torch.log_softmax(lo, -1).gather(-1, act[:, None]).squeeze(-1).sum(-1, keepdim=True)
    alp = torch.log_softmax(la, -1).gather(-1, act[:, None]).squeeze(-1).sum(-1, keepdim=True)
    loss = F.mse_loss(olp, alp)   # (the reference's summed log-probabilities: one number each)
    loss.backward()
    d, out = torch.zeros(B, A, device=DEV), torch.zeros(1, device=DEV)
    lod, lad, actd = lo.to(DEV), la.detach().to(DEV), act.float().to(DEV)
    ssa._lib.check(ssa._lib.lib.ssac_action_invariance_discrete_bwd(
        lod.data_ptr(), lad.data_ptr(), actd.data_ptr(), B, A, 0.1, d.data_ptr(),
        out.data_ptr(), 0, _st(ssa)))
    # (the two sums are ~ -3 B each and their difference is O(1): the fp32 summation ORDER shows in the 4th digit of
    #  the difference -- the kernel sums the per-row differences, which is the better-conditioned order)
    tol = 2e-6 * float(olp.abs()) + 1e-6
    S_ref = float(alp - olp)
    assert abs(math.sqrt(float(out)) - abs(S_ref)) <= tol
    assert torch.allclose(d.cpu(), 0.1 * la.grad, rtol=0.0, atol=0.1 * 2 * tol + 1e-6 * float(la.grad.abs().max()))


@pytest.mark.parametrize("B,A,clip", [(1, 1, 0.0), (100, 6, 0.3), (513, 17, 0.5)])
def test_exploration_noise_and_deterministic_logprob(ssa, B, A, clip):
    g = torch.Generator().manual_seed(B * 7 + A)
    S = 5
    x = torch.randn(B, S + A, generator=g).tanh()
    noise = torch.randn(B, A, generator=g)
    want = x.clone()
    nz = 0.4 * noise
    if clip > 0:
        nz = nz.clamp(-clip, clip)
    want[:, S:] = (x[:, S:] + nz).clamp(-1 + 1e-6, 1 - 1e-6)   # learning_utils.py:49-55
    xd, nd = x.to(DEV), noise.to(DEV)
    ssa._lib.check(ssa._lib.lib.ssac_exploration_noise(xd.data_ptr(), S + A, S, nd.data_ptr(), 0.4, clip, B, A, _st(ssa)))
    assert torch.allclose(xd.cpu(), want, atol=1e-7)
    eps = torch.randn(B, A, generator=g)
    loc = torch.randn(B, A, generator=g)
    dist = torch.distributions.Normal(loc, 1e-4)
    lp, ed = torch.zeros(B, device=DEV), eps.to(DEV)
    ssa._lib.check(ssa._lib.lib.ssac_det_logprob(ed.data_ptr(), B, A, lp.data_ptr(), _st(ssa)))
    assert torch.allclose(lp.cpu(), -(eps ** 2).sum(-1) / 2 + A * (-math.log(1e-4) - 0.5 * math.log(2 * math.pi)), rtol=1e-5)
    ssa._lib.check(ssa._lib.lib.ssac_det_logprob(0, B, A, lp.data_ptr(), _st(ssa)))
    assert torch.allclose(lp.cpu(), dist.log_prob(loc).sum(-1), rtol=1e-5)
