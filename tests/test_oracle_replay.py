"""Pin the replay oracle: reference known-answer tests + golden vectors made by the reference."""
import os

import numpy as np
import pytest

from oracle.replay_oracle import (TreeOracle, ReplayOracle, PrioritizedReplayOracle,
                                  tree_capacity)
from tests import replay_cases as RC


# ---- the reference's own five known-answer tests, restated on the oracle
# (/root/reference/imitation/helpers/segment_tree_test.py:6-95)

def test_ref_tree_set():
    t = TreeOracle(4, 'sum')
    t.set(2, 1.0); t.set(3, 3.0)
    assert np.isclose(t.reduce(), 4.0)
    assert np.isclose(t.reduce(0, 2), 0.0)
    assert np.isclose(t.reduce(0, 3), 1.0)
    assert np.isclose(t.reduce(2, 3), 1.0)
    assert np.isclose(t.reduce(2, -1), 1.0)
    assert np.isclose(t.reduce(2, 4), 4.0)


def test_ref_tree_set_overlap():
    t = TreeOracle(4, 'sum')
    t.set(2, 1.0); t.set(2, 3.0)
    assert np.isclose(t.reduce(), 3.0)
    assert np.isclose(t.reduce(2, 3), 3.0)
    assert np.isclose(t.reduce(2, -1), 3.0)
    assert np.isclose(t.reduce(2, 4), 3.0)
    assert np.isclose(t.reduce(1, 2), 0.0)


def test_ref_prefixsum_idx():
    t = TreeOracle(4, 'sum')
    t.set(2, 1.0); t.set(3, 3.0)
    for s, e in ((0.0, 2), (0.5, 2), (0.99, 2), (1.01, 3), (3.0, 3), (4.0, 3)):
        assert t.find_prefixsum_idx(s) == e


def test_ref_prefixsum_idx2():
    t = TreeOracle(4, 'sum')
    for i, v in enumerate((0.5, 1.0, 1.0, 3.0)):
        t.set(i, v)
    for s, e in ((0.0, 0), (0.55, 1), (0.99, 1), (1.51, 2), (3.0, 3), (5.5, 3)):
        assert t.find_prefixsum_idx(s) == e


def test_ref_min_interval_tree():
    t = TreeOracle(4, 'min')
    t.set(0, 1.0); t.set(2, 0.5); t.set(3, 3.0)
    exp = {(0, None): 0.5, (0, 2): 1.0, (0, 3): 0.5, (0, -1): 0.5, (2, 4): 0.5, (3, 4): 3.0}
    for (a, b), e in exp.items():
        assert np.isclose(t.reduce(a, b), e)
    t.set(2, 0.7)
    exp = {(0, None): 0.7, (0, 2): 1.0, (0, 3): 0.7, (0, -1): 0.7, (2, 4): 0.7, (3, 4): 3.0}
    for (a, b), e in exp.items():
        assert np.isclose(t.reduce(a, b), e)
    t.set(2, 4.0)
    exp = {(0, None): 1.0, (0, 2): 1.0, (0, 3): 1.0, (0, -1): 1.0, (2, 4): 3.0, (2, 3): 4.0,
           (2, -1): 4.0, (3, 4): 3.0}
    for (a, b), e in exp.items():
        assert np.isclose(t.reduce(a, b), e)


def test_tree_capacity():
    assert [tree_capacity(x) for x in (1, 2, 3, 1000, 1024, 1025, 10 ** 6)] == \
        [1, 2, 4, 1024, 1024, 2048, 2 ** 20]


# ---- golden vectors recorded from the reference itself

def _oracle_trees(cap):
    st, mt = TreeOracle(cap, 'sum'), TreeOracle(cap, 'min')

    def t_set(i, v):
        st.set(i, v); mt.set(i, v)
    return (t_set, st.reduce, mt.reduce, st.find_prefixsum_idx,
            lambda kind: (st if kind == 'sum' else mt).v.copy())


@pytest.mark.parametrize('name', ['replay_tree_cap16', 'replay_tree_cap1024'])
def test_tree_golden(golden_dir, name):
    RC.run_tree_fixture(np.load(os.path.join(golden_dir, name + '.npz')), _oracle_trees)


@pytest.mark.parametrize('name', ['replay_per_a', 'replay_per_b', 'replay_per_demos'])
def test_per_golden(golden_dir, name):
    fx = np.load(os.path.join(golden_dir, name + '.npz'))
    rb = RC.run_per_fixture(
        fx, lambda limit, ob, ac, alpha, beta:
        PrioritizedReplayOracle(limit, (ob,), (ac,), alpha, beta))
    assert np.array_equal(rb.rings['obs0'].data, fx['final_obs0'])
    assert np.array_equal(rb.rings['rews'].data, fx['final_rews'])
    assert rb.rings['obs0'].start == int(fx['final_start'])
    assert rb.rings['obs0'].length == int(fx['final_length'])


@pytest.mark.parametrize('name', ['replay_rb', 'replay_rb_partial'])
def test_rb_golden(golden_dir, name):
    RC.run_rb_fixture(np.load(os.path.join(golden_dir, name + '.npz')),
                      lambda limit, ob, ac: ReplayOracle(limit, (ob,), (ac,)))


def test_c_oracle_matches_python():
    """oracle/per_tree.c == oracle/replay_oracle.py bit for bit on a PER workload."""
    from oracle.per_tree_c import PerTreesC
    import random
    rs = np.random.RandomState(5)
    limit = 3000
    py = PrioritizedReplayOracle(limit, (2,), (1,), 0.3, 1.0)
    ct = PerTreesC(py.cap)
    n = 2500
    for i in range(n):
        py.append(np.zeros(2), np.zeros(1), 0.0, np.zeros(2), 0.0)
    ct.fill_range(0, n, 1.0 ** 0.3)
    max_p = 1.0
    for r in range(6):
        u = rs.rand(128)
        idxs_c, iws_c = ct.per_sample(n, u, 1.0)
        idxs_p = py.sample_idxs(128, u)
        assert np.array_equal(idxs_c, idxs_p)
        assert np.array_equal(iws_c, py.importance_weights(idxs_p))
        pr = np.abs(rs.randn(128)) + 1e-6
        max_p = ct.per_update(idxs_c, pr, 0.3, max_priority=max_p)
        py.update_priorities(idxs_p, pr)
        assert max_p == py.max_priority
        for a, b in ((0, n), (0, n - 1), (17, 1999), (5, -1)):
            assert ct.sum(a, b) == py.sum_tree.reduce(a, b)
            assert ct.min(a, b) == py.min_tree.reduce(a, b)
    assert np.array_equal(ct.sumv, py.sum_tree.v)
    assert np.array_equal(ct.minv, py.min_tree.v)


@pytest.mark.parametrize('name', ['replay_nstep_a', 'replay_nstep_wrap'])
def test_nstep_lookahead_matches_reference(golden_dir, name):
    """RB.lookahead (memory.py:113-160) recorded from the reference: discounted n-step reward (float64
    Horner from the far end), episode trimming, the done-at-window-end quirk, ring wrap-around."""
    from oracle.replay_oracle import ReplayOracle
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    rb = ReplayOracle(int(g['limit']), (int(g['ob']),), (int(g['ac']),))
    for i in range(len(g['in_rews'])):
        rb.append(g['in_obs0'][i], g['in_acs'][i], g['in_rews'][i], g['in_obs1'][i], g['in_dones1'][i])
    assert rb.rings['obs0'].start == int(g['start']) and rb.num_entries == int(g['length'])
    la = rb.lookahead(g['idxs'], int(g['n']), float(g['gamma']))
    for k in ('obs0', 'obs1', 'acs', 'dones1', 'td_len'):
        assert np.array_equal(np.asarray(la[k], dtype=np.float64), np.asarray(g['la_' + k], dtype=np.float64)), k
    assert np.array_equal(la['rews'], g['la_rews'])   # float64, bit for bit
    assert (g['la_dones1'] == 1).any() and (g['la_td_len'] < int(g['n'])).any()
