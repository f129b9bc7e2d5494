"""The device pow that carries the host libm's bits (csrc/pow_bits.h) against libm itself, bit for bit
(tolerance: none).  Leaf values `priority ** alpha` (memory.py:341-342,376-377) are float32-born bases
|td| + 1e-6f widened to double: EVERY float32 from 1e-6 to 1e4 (2.78e8 bases, all binades and their boundary
neighbours included) for alpha 0.3 and 0.6, a 2^26 stretch with the demonstration bonus added; the importance
weights `(n * prob) ** -beta` (memory.py:301,308) are arbitrary doubles: 2^24 random ones per beta."""
import ctypes as C

import numpy as np
import pytest
import torch

from sam_tf_amd import _lib, ops

pytestmark = pytest.mark.gpu

CHUNK = 1 << 24


def _bits(f):
    return int(np.array([f], dtype=np.float32).view(np.uint32)[0])


def _range_mismatches(first, n, bonus, y):
    L = _lib.lib()
    dev = torch.empty(CHUNK, dtype=torch.float64, device='cuda')
    host = np.empty(CHUNK, dtype=np.float64)
    bad = 0
    for lo in range(0, n, CHUNK):
        m = min(CHUNK, n - lo)
        ops.check(L.samx_pow_device_f32_range(first + lo, m, bonus, y, C.c_void_p(dev.data_ptr()),
                                              ops.stream_ptr()))
        ops.check(L.samx_pow_host_f32_range(first + lo, m, bonus, y, C.c_void_p(host.ctypes.data)))
        got = dev[:m].cpu().numpy().view(np.uint64)
        bad += int((got != host[:m].view(np.uint64)).sum())
    return bad


def test_host_build_matches_this_machines_libm():
    assert ops.pow_matches_host(), "this machine's libm pow takes a code path pow_bits.h does not follow"


@pytest.mark.parametrize('alpha', [0.3, 0.6])
def test_every_float32_priority_between_1e_minus_6_and_1e4(alpha):
    first, last = _bits(1e-6), _bits(1e4)
    n = last - first + 1
    assert n >= 1 << 26
    assert _range_mismatches(first, n, 0.0, alpha) == 0


@pytest.mark.parametrize('alpha', [0.3, 0.6])
def test_priorities_with_the_demonstration_bonus(alpha):
    assert _range_mismatches(_bits(1e-6), 1 << 26, 0.1, alpha) == 0


@pytest.mark.parametrize('beta', [0.4, 0.5, 1.0, 0.123456789])
def test_importance_weight_pows(beta):
    L = _lib.lib()
    rs = np.random.RandomState(int(beta * 1000))
    n = 1 << 24
    x = np.ldexp(rs.uniform(1.0, 2.0, n), rs.randint(-40, 41, n))
    x[:6] = [1.0, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.0, 0.5, float.fromhex('0x1.6955500000000p-1')]
    want = np.empty(n)
    ops.check(L.samx_pow_host(C.c_void_p(x.ctypes.data), -beta, n, C.c_void_p(want.ctypes.data)))
    got = ops.pow_device(torch.as_tensor(x).cuda(), -beta).cpu().numpy()
    assert int((got.view(np.uint64) != want.view(np.uint64)).sum()) == 0


def test_leaves_from_td_on_the_device_match_the_host_refresh():
    """samx_per_leaves_from_td vs samx_per_leaves_from_td_host: leaves, max_priority, the skipped marker."""
    rs = np.random.RandomState(3)
    n, cap, n_ent, n_demos = 700, 1024, 900, 50
    td = (rs.standard_cauchy(n) * 10.0 ** rs.uniform(-6, 2, n)).astype(np.float32)
    idx = rs.randint(0, n_ent, n).astype(np.int64)
    want = np.empty(n)
    mp = ops.per_leaves_from_td_host(td, idx, n_ent, 0.6, n_demos, 0.1, 1.0, want)
    idx_d = idx.copy()
    idx_d[5] = -1            # "nothing pending" entries are skipped
    state = ops.make_per_state(1.0)
    leaf = torch.empty(n, dtype=torch.float64, device='cuda')
    ops.per_leaves_from_td(torch.as_tensor(td).cuda(), torch.as_tensor(idx_d).cuda(), cap, n_ent, 0.6, n_demos, 0.1,
                           leaf, state)
    got = leaf.cpu().numpy()
    keep = np.arange(n) != 5
    assert np.array_equal(got[keep], want[keep]) and got[5] == 0.0
    st = state.cpu().numpy()
    p5 = float(np.float32(abs(td[5])) + np.float32(1e-6)) + (0.1 if idx[5] < n_demos else 0.0)
    assert st[0] == mp or (p5 == mp and st[0] <= mp)
    assert st.view(np.uint64)[1] == 0
    # an index without an element behind it raises the reference's assert bit (memory.py:372)
    idx_d[7] = n_ent + 3
    ops.per_leaves_from_td(torch.as_tensor(td).cuda(), torch.as_tensor(idx_d).cuda(), cap, n_ent, 0.6, n_demos, 0.1,
                           leaf, state)
    assert int(state.cpu().numpy().view(np.uint64)[1]) == 2


@pytest.mark.parametrize('alpha,beta', [(0.3, 1.0), (0.6, 0.4), (1.0, 0.5)])
def test_driver_gate_checks_the_device_build_with_the_buffers_exponents(alpha, beta):
    """What UpdateStepDriver asks before it keeps the priority refresh on the device (round-2 advice: the host
    self-check alone never ran the device build, nor the configured alpha / -beta)."""
    assert ops.pow_device_matches_host(alpha, beta, 'cuda')
    assert (float(alpha), float(beta), torch.cuda.current_device()) in ops._POW_DEV_OK


def test_driver_falls_back_to_the_host_refresh_when_the_device_gate_fails(monkeypatch):
    """A device build that does not reproduce this machine's libm for the buffer's exponents: the driver keeps the
    host refresh (one row per launch, logical indices) instead of the device one."""
    import bench
    from sam_tf_amd.comm import TorchComm
    cfg = dict(bench.CONFIGS['C3'], mem=2048)
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 3, True)
    assert drv.per_dev and drv.K > 1
    rb = mu.replay_buffer
    monkeypatch.setitem(ops._POW_DEV_OK, (float(rb.alpha), float(rb.beta), torch.cuda.current_device()), False)
    from sam_tf_amd.driver import UpdateStepDriver
    drv2 = UpdateStepDriver(mu, d, pool, use_graph=True)
    assert not drv2.per_dev and drv2.K == 1 and not drv2.per_rows
    drv2.run(3)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t).all()) for t in (mu.params, d.params))
