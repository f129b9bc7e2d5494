"""Fused clip + TF1 Adam + Polyak vs the oracle (tolerance 1e-5 relative, stated per check)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,clip,world', [(5185, 5.0, 1), (166913, 0.5, 4), (1, 5.0, 1), (1000, None, 2)])
def test_clip_adam_polyak(n, clip, world):
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    rs = np.random.RandomState(n)
    theta = rs.randn(n).astype(np.float32)
    targ = rs.randn(n).astype(np.float32)
    m = np.zeros(n); v = np.zeros(n)
    th_d, tg_d = torch.as_tensor(theta).cuda(), torch.as_tensor(targ).cuda()
    m_d, v_d = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    part = ops.adam_partials()
    t_dev = torch.zeros(1, dtype=torch.int64, device='cuda')
    th, tg = theta.astype(np.float64), targ.astype(np.float64)
    for t in range(1, 6):
        gs = [rs.randn(n).astype(np.float32) * (3.0 if t % 2 else 0.01) for _ in range(world)]
        th, m, v = NO.mpi_adam_step(th, m, v, gs, 1e-3, t, clip)
        tg = NO.polyak(tg, th, 0.01)
        gsum = torch.as_tensor(np.sum(np.stack(gs).astype(np.float64), axis=0).astype(np.float32)).cuda()
        ops.clip_adam(th_d, gsum, m_d, v_d, 1e-3, t, clip, part, grad_scale=1.0 / world, targ=tg_d,
                      tau=0.01)
    # post-update weights within 1e-5 relative (north_star)
    np.testing.assert_allclose(th_d.cpu().numpy(), th, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(tg_d.cpu().numpy(), tg, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m_d.cpu().numpy(), m, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(v_d.cpu().numpy(), v, rtol=1e-5, atol=1e-9)


def test_clip_copy_polyak_mismatch():
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    rs = np.random.RandomState(0)
    g = rs.randn(4545).astype(np.float32)
    part = ops.adam_partials()
    for c in (5.0, 1000.0):
        out = ops.clip_copy(torch.as_tensor(g).cuda(), c, part)
        exp, _ = NO.clip_by_global_norm(g.astype(np.float64), c)
        np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=1e-5)
    a = torch.as_tensor(g).cuda(); b = a.clone()
    ops.polyak(b, a, 1.0)  # hard copy (sam_agent.py:33)
    assert int(ops.count_mismatch(a, b).item()) == 0
    b[17] += 1.0; b[4000] = 0.0
    assert int(ops.count_mismatch(a, b).item()) == 2
