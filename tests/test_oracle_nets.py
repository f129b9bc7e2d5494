"""Sanity of the network oracle (parity UNPINNED at the TF boundary, see oracle/nets_oracle.py):
its analytic gradients agree with central finite differences of its own losses in fp64, and the
small closed-form pieces match hand-computed values."""
import numpy as np
import pytest
import torch

from oracle import nets_oracle as NO


def _setup(ln, B=12, O=5, A=3, H1=16, H2=12, seed=0):
    rs = np.random.RandomState(seed)
    da = NO.MLP(O, H1, H2, A, ln)
    dc = NO.MLP(O + A, H1, H2, 1, ln)
    pa, pc, pd = (NO.init_params(da, rs, 'actor').astype(np.float64),
                  rs.randn(NO.param_count(dc)) * 0.3, rs.randn(NO.param_count(dc)) * 0.3)
    pa = pa + rs.randn(pa.size) * 0.1
    data = dict(obs0=rs.randn(B, O) * 2, acs=rs.uniform(-1, 1, (B, A)), obs1=rs.randn(B, O),
                e_obs=rs.randn(B, O) + 0.5, e_acs=rs.uniform(-1, 1, (B, A)),
                rews=rs.uniform(0, 5, (B, 1)), dones=(rs.rand(B, 1) < 0.3).astype(np.float64),
                iws=rs.uniform(0.2, 1, (B, 1)), tc=rs.randn(B, 1) * 2,
                eps_ob=rs.rand(B, O), eps_ac=rs.rand(B, A), u=rs.rand(B))
    mean, std = rs.randn(O) * 0.1, 1 + rs.rand(O)
    return da, dc, pa, pc, pd, data, mean, std, rs


def _fd(fun, p, coords, h=1e-6):
    out = []
    for c in coords:
        pp, pm = p.copy(), p.copy()
        pp[c] += h; pm[c] -= h
        out.append((fun(pp) - fun(pm)) / (2 * h))
    return np.array(out)


@pytest.mark.parametrize('ln', [False, True])
def test_critic_grad_fd(ln):
    da, dc, pa, pc, pd, D, mean, std, rs = _setup(ln)

    def run(p):
        return NO.critic_loss_grad(p, dc, D['obs0'], D['acs'], D['iws'], D['tc'], mean, std, 1e-3)
    losses, g, td = run(pc)
    coords = rs.choice(pc.size, 25, replace=False)
    fd = _fd(lambda p: run(p)[0][2], pc, coords)
    np.testing.assert_allclose(g[coords], fd, rtol=2e-5, atol=1e-8)
    assert np.isclose(losses[2], losses[0] + losses[1])


@pytest.mark.parametrize('ln', [False, True])
def test_actor_grad_fd(ln):
    da, dc, pa, pc, pd, D, mean, std, rs = _setup(ln)

    def run(p):
        return NO.actor_loss_grad(p, da, pc, dc, pd, dc, D['obs0'], mean, std, 1.0, 1.0, 0.5)
    losses, g = run(pa)
    coords = rs.choice(pa.size, 25, replace=False)
    fd = _fd(lambda p: run(p)[0][2], pa, coords)
    np.testing.assert_allclose(g[coords], fd, rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('ent', [0.0, 0.01])
def test_disc_grad_fd(ln, ent):
    da, dc, pa, pc, pd, D, mean, std, rs = _setup(ln)
    real, fake = NO.smooth_labels(D['u'])

    def run(p):
        return NO.disc_loss_grad(p, dc, D['obs0'], D['acs'], D['e_obs'], D['e_acs'], D['eps_ob'],
                                 D['eps_ac'], real, fake, mean, std, ent_reg_scale=ent)
    losses, g = run(pd)
    coords = rs.choice(pd.size, 30, replace=False)
    fd = _fd(lambda p: run(p)[0][7], pd, coords)
    np.testing.assert_allclose(g[coords], fd, rtol=5e-5, atol=1e-7)
    assert losses[6] > 0  # the penalty is active in this setup


def test_closed_forms():
    # huber, delta 1 (math_util.py:22-29)
    x = torch.tensor([-2.0, -0.5, 0.0, 0.5, 1.0, 3.0], dtype=torch.float64)
    np.testing.assert_allclose(NO.huber(x).numpy(), [1.5, 0.125, 0, 0.125, 0.5, 2.5])
    # sigmoid CE against the textbook form
    z = torch.tensor([0.0, 1.0, 0.3], dtype=torch.float64)
    s = torch.tensor([0.7, -1.2, 2.0], dtype=torch.float64)
    ref = -(z * torch.log(torch.sigmoid(s)) + (1 - z) * torch.log(1 - torch.sigmoid(s)))
    np.testing.assert_allclose(NO.sigmoid_ce(s, z).numpy(), ref.numpy(), rtol=1e-12)
    # rms: init (0, eps, eps) then one batch (mpi_running_mean_std.py:15-34)
    sums = NO.rms_update(NO.rms_init(2), np.array([[1.0, 2.0], [3.0, 6.0]]), 2)
    mean, std = NO.rms_mean_std(sums, 2)
    np.testing.assert_allclose(mean, np.array([4, 8]) / 2.01, rtol=1e-6)
    np.testing.assert_allclose(std, np.sqrt(np.maximum(
        (np.array([10, 40]) + 1e-2) / 2.01 - (np.array([4, 8]) / 2.01) ** 2, 1e-2)), rtol=1e-5)
    # label smoothing range: real in (0.7, 1.2]
    real, fake = NO.smooth_labels(np.array([0.0, 0.5, 0.999]))
    assert real.max() == 1.2 and real.min() > 0.7 and (fake == 0).all()
    # clip: g * c / max(|g|, c)
    g, n = NO.clip_by_global_norm(np.array([3.0, 4.0]), 2.5)
    np.testing.assert_allclose(g, [1.5, 2.0]); assert n == 5.0
    g, n = NO.clip_by_global_norm(np.array([3.0, 4.0]), 50.0)
    np.testing.assert_allclose(g, [3.0, 4.0])


def test_adam_tf1_first_step():
    th, m, v = NO.adam_tf1(np.array([1.0]), np.zeros(1), np.zeros(1), np.array([0.5]), 1e-3, 1)
    # t=1: m=.05, v=2.5e-4, lr_t = lr*sqrt(1-.999)/(1-.9) -> step = lr_t*m/(sqrt(v)+eps) ~= lr
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(th, 1.0 - lr_t * 0.05 / (np.sqrt(2.5e-4) + 1e-8))
