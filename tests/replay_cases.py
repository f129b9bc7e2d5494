"""Scenario runners shared by the oracle tests (CPU) and the HIP parity tests (GPU).

Each runner replays the exact call sequence that produced a golden fixture
(tests/golden/make_golden_replay.py) against any object exposing the reference replay API
(memory.py:60-390: append / sample / sample_uniform / sample_recent / update_priorities /
add_demo_transitions_to_mem / num_entries) plus three read-back accessors (tree_sum, tree_min,
tree_values), and asserts bit-equality with the recorded outputs.
"""
import random

import numpy as np


class Demos(object):
    pass


def _eq(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(a, b), "%s differs: max|d|=%r" % (
        what, np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


def run_per_fixture(fx, make_rb, iws_exact=True):
    """fx: loaded npz; make_rb(limit, ob, ac, alpha, beta) -> PER object."""
    seed = int(fx['seed'])
    rs = np.random.RandomState(seed)  # same role as in the generator: priorities only
    random.seed(seed)
    np.random.seed(seed)
    limit, ob, ac = int(fx['limit']), int(fx['ob']), int(fx['ac'])
    B, rounds, n_more = int(fx['B']), int(fx['rounds']), int(fx['n_more'])
    n_demos, n_init = int(fx['n_demos']), int(fx['n_init'])
    rb = make_rb(limit, ob, ac, float(fx['alpha']), float(fx['beta']))
    T = {k: fx['T_' + k] for k in ('obs0', 'acs', 'rews', 'obs1', 'dones1')}
    # the generator consumed rs for the transition data first; burn the same draws
    total = n_demos + n_init + rounds * n_more
    rs.randn(total, ob); rs.uniform(-1, 1, (total, ac)); rs.uniform(0, 5, (total,))
    rs.randn(total, ob); rs.rand(total)
    cur = 0
    if n_demos:
        d = Demos()
        d.obs0, d.acs, d.env_rews = T['obs0'][:n_demos], T['acs'][:n_demos], T['rews'][:n_demos]
        d.obs1, d.dones1 = T['obs1'][:n_demos], T['dones1'][:n_demos]
        rb.add_demo_transitions_to_mem(d)
        cur = n_demos

    def push(k):
        nonlocal cur
        for _ in range(k):
            rb.append(T['obs0'][cur], T['acs'][cur], T['rews'][cur], T['obs1'][cur],
                      T['dones1'][cur])
            cur += 1

    push(n_init)
    for r in range(rounds):
        assert rb.num_entries == int(fx['n_entries'][r])
        _eq(rb.tree_sum(0, rb.num_entries - 1), fx['p_total'][r], 'p_total[%d]' % r)
        batch = rb.sample(B)
        _eq(np.asarray(batch['idxs'], dtype=np.int64), fx['idxs'][r], 'idxs[%d]' % r)
        iws = np.asarray(batch['iws'], dtype=np.float64).ravel()
        if iws_exact:
            _eq(iws, fx['iws'][r], 'iws[%d]' % r)
        else:
            # device pow (OCML) vs libm pow: value parity, not bit parity (DESIGN.md "IS weights")
            np.testing.assert_allclose(iws, fx['iws'][r], rtol=1e-13, atol=0)
        _eq(batch['obs0'], fx['b_obs0'][r], 'obs0[%d]' % r)
        _eq(batch['rews'], fx['b_rews'][r], 'rews[%d]' % r)
        ub = rb.sample_uniform(B)
        _eq(np.asarray(ub['idxs'], dtype=np.int64), fx['uni_idxs'][r], 'uni_idxs[%d]' % r)
        prios = np.abs(rs.randn(B)) + 1e-6
        _eq(prios, fx['prios'][r], 'prios[%d]' % r)
        rb.update_priorities(batch['idxs'], prios)
        _eq(rb.tree_sum(), fx['root_sum'][r], 'root_sum[%d]' % r)
        _eq(rb.tree_min(), fx['root_min'][r], 'root_min[%d]' % r)
        _eq(float(rb.max_priority), fx['max_prio'][r], 'max_prio[%d]' % r)
        push(n_more)
    _eq(rb.tree_values('sum'), fx['final_sum_values'], 'sum tree')
    _eq(rb.tree_values('min'), fx['final_min_values'], 'min tree')
    return rb


def run_rb_fixture(fx, make_rb):
    seed = int(fx['seed'])
    np.random.seed(seed)
    limit, ob, ac = int(fx['limit']), int(fx['ob']), int(fx['ac'])
    rb = make_rb(limit, ob, ac)
    T = {k: fx['T_' + k] for k in ('obs0', 'acs', 'rews', 'obs1', 'dones1')}
    for i in range(int(fx['n_app'])):
        rb.append(T['obs0'][i], T['acs'][i], T['rews'][i], T['obs1'][i], T['dones1'][i])
    assert rb.num_entries == int(fx['n_entries'])
    assert rb.latest_entry_idx == int(fx['latest'])
    B, window = int(fx['B']), int(fx['window'])
    s1 = rb.sample(B)
    _eq(np.asarray(s1['idxs'], dtype=np.int64), fx['s1_idxs'], 's1 idxs')
    for k in ('obs0', 'acs', 'rews', 'dones1', 'obs1'):
        _eq(s1[k], fx['s1_' + k], 's1 ' + k)
    s2 = rb.sample_recent(B, window)
    _eq(np.asarray(s2['idxs'], dtype=np.int64), fx['s2_idxs'], 's2 idxs')
    _eq(s2['obs0'], fx['s2_obs0'], 's2 obs0')
    s3 = rb.sample_recent(B, 10 * limit)
    _eq(np.asarray(s3['idxs'], dtype=np.int64), fx['s3_idxs'], 's3 idxs')
    _eq(s3['obs1'], fx['s3_obs1'], 's3 obs1')
    return rb


def run_tree_fixture(fx, make_trees):
    """make_trees(cap) -> (set(idx, val) applying to both trees, sum(a,b), min(a,b), prefix(s),
    values(kind))."""
    cap = int(fx['cap'])
    t_set, t_sum, t_min, t_prefix, t_values = make_trees(cap)
    for i, v in zip(fx['w_idx'], fx['w_val']):
        t_set(int(i), float(v))
    _eq(t_values('sum'), fx['sum_values'], 'sum values')
    _eq(t_values('min'), fx['min_values'], 'min values')
    for (a, b), es, em in zip(fx['q'], fx['q_sum'], fx['q_min']):
        _eq(t_sum(int(a), int(b)), es, 'sum(%d,%d)' % (a, b))
        _eq(t_min(int(a), int(b)), em, 'min(%d,%d)' % (a, b))
    for s, e in zip(fx['pref'], fx['pref_idx']):
        assert int(t_prefix(float(s))) == int(e), (s, e)
