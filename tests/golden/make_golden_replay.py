"""Generate the golden replay/segment-tree vectors by running the REFERENCE code.

Runs only in the build container (needs /root/reference); the GPU box never runs it.  It imports
/root/reference/imitation/helpers/segment_tree.py and .../imitation_algorithms/memory.py unchanged
(an empty module named ``tensorflow`` is placed in ``sys.modules`` because math_util/misc_util do
``import tensorflow`` at module top without using it on this path) and records inputs + outputs as
plain arrays in ``tests/golden/replay_*.npz``.  No reference source is stored, only data.

Priorities are handed to the reference as float64 so that its arithmetic is the float64 arithmetic
of the numpy==1.16.1 it pins (see oracle/replay_oracle.py header).

    python tests/golden/make_golden_replay.py
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference')
sys.modules.setdefault('tensorflow', types.ModuleType('tensorflow'))

from imitation.helpers.segment_tree import SumSegmentTree, MinSegmentTree  # noqa: E402
from imitation.imitation_algorithms import memory as M  # noqa: E402


def draw_uniforms(n):
    """Peek the next n random.random() draws without consuming them."""
    st = random.getstate()
    u = np.array([random.random() for _ in range(n)], dtype=np.float64)
    random.setstate(st)
    return u


def tree_values(t):
    return np.array(t._value, dtype=np.float64)


def gen_tree_ops(cap, seed, n_writes, n_queries):
    rs = np.random.RandomState(seed)
    st, mt = SumSegmentTree(cap), MinSegmentTree(cap)
    w_idx = rs.randint(0, cap, size=n_writes)
    w_val = np.abs(rs.randn(n_writes)) + 1e-3
    for i, v in zip(w_idx, w_val):
        st[int(i)] = float(v)
        mt[int(i)] = float(v)
    q = []
    for _ in range(n_queries):
        a = int(rs.randint(0, cap))
        b = int(rs.randint(a + 1, cap + 1))
        q.append((a, b))
    # negative `end` wraps by +cap (segment_tree.py:50-51)
    q += [(0, -1), (1, -1), (0, cap), (cap - 1, cap), (0, 1)]
    q = np.array(q, dtype=np.int64)
    qs = np.array([st.sum(int(a), int(b)) for a, b in q], dtype=np.float64)
    qm = np.array([mt.min(int(a), int(b)) for a, b in q], dtype=np.float64)
    pref = rs.rand(n_queries) * st.sum()
    pref = np.concatenate([pref, [0.0, st.sum()]])
    pidx = np.array([st.find_prefixsum_idx(float(s)) for s in pref], dtype=np.int64)
    return dict(cap=cap, w_idx=w_idx.astype(np.int64), w_val=w_val, q=q, q_sum=qs, q_min=qm,
                pref=pref, pref_idx=pidx, sum_values=tree_values(st), min_values=tree_values(mt))


class _Demos(object):
    pass


def gen_per(limit, ob, ac, alpha, beta, seed, n_init, rounds, B, n_more, n_demos=0):
    rs = np.random.RandomState(seed)
    random.seed(seed)
    np.random.seed(seed)
    rb = M.PrioritizedRB(limit, (ob,), (ac,), alpha=alpha, beta=beta)
    total = n_demos + n_init + rounds * n_more
    T = dict(obs0=rs.randn(total, ob).astype(np.float32),
             acs=rs.uniform(-1, 1, (total, ac)).astype(np.float32),
             rews=rs.uniform(0, 5, (total,)).astype(np.float32),
             obs1=rs.randn(total, ob).astype(np.float32),
             dones1=(rs.rand(total) < 0.05).astype(np.float32))
    cur = 0
    if n_demos:
        d = _Demos()
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
    rec = dict(u=[], idxs=[], iws=[], prios=[], root_sum=[], root_min=[], max_prio=[],
               n_entries=[], p_total=[], b_obs0=[], b_rews=[], uni_idxs=[])
    for _ in range(rounds):
        rec['n_entries'].append(rb.num_entries)
        rec['p_total'].append(rb.sum_st.sum(end=rb.num_entries - 1))
        u = draw_uniforms(B)
        batch = rb.sample(B)
        rec['u'].append(u)
        rec['idxs'].append(np.asarray(batch['idxs'], dtype=np.int64))
        rec['iws'].append(np.asarray(batch['iws'], dtype=np.float64).ravel())
        rec['b_obs0'].append(batch['obs0'])
        rec['b_rews'].append(batch['rews'])
        ub = rb.sample_uniform(B)
        rec['uni_idxs'].append(np.asarray(ub['idxs'], dtype=np.int64))
        prios = np.abs(rs.randn(B)) + 1e-6  # float64
        rb.update_priorities(batch['idxs'], prios)
        rec['prios'].append(prios)
        rec['root_sum'].append(rb.sum_st.sum())
        rec['root_min'].append(rb.min_st.min())
        rec['max_prio'].append(float(rb.max_priority))
        push(n_more)
    out = {k: np.array(v) for k, v in rec.items()}
    out.update({'T_' + k: v for k, v in T.items()})
    out.update(limit=limit, ob=ob, ac=ac, alpha=alpha, beta=beta, seed=seed, n_init=n_init,
               rounds=rounds, B=B, n_more=n_more, n_demos=n_demos,
               final_sum_values=tree_values(rb.sum_st), final_min_values=tree_values(rb.min_st),
               final_obs0=rb.ring_buffers['obs0'].data.copy(),
               final_rews=rb.ring_buffers['rews'].data.copy(),
               final_start=rb.ring_buffers['obs0'].start,
               final_length=rb.ring_buffers['obs0'].length)
    return out


def gen_rb(limit, ob, ac, seed, n_app, B, window):
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    rb = M.RB(limit, (ob,), (ac,))
    T = dict(obs0=rs.randn(n_app, ob).astype(np.float32),
             acs=rs.uniform(-1, 1, (n_app, ac)).astype(np.float32),
             rews=rs.uniform(0, 5, (n_app,)).astype(np.float32),
             obs1=rs.randn(n_app, ob).astype(np.float32),
             dones1=(rs.rand(n_app) < 0.05).astype(np.float32))
    for i in range(n_app):
        rb.append(T['obs0'][i], T['acs'][i], T['rews'][i], T['obs1'][i], T['dones1'][i])
    s1 = rb.sample(B)
    s2 = rb.sample_recent(B, window)
    s3 = rb.sample_recent(B, 10 * limit)  # window larger than the buffer
    out = {'T_' + k: v for k, v in T.items()}
    out.update(limit=limit, ob=ob, ac=ac, seed=seed, n_app=n_app, B=B, window=window,
               s1_idxs=s1['idxs'].astype(np.int64), s1_obs0=s1['obs0'], s1_acs=s1['acs'],
               s1_rews=s1['rews'], s1_dones1=s1['dones1'], s1_obs1=s1['obs1'],
               s2_idxs=s2['idxs'].astype(np.int64), s2_obs0=s2['obs0'],
               s3_idxs=s3['idxs'].astype(np.int64), s3_obs1=s3['obs1'],
               latest=rb.latest_entry_idx, n_entries=rb.num_entries)
    return out


def gen_nstep(limit, ob, ac, seed, n_app, n, gamma, p_done):
    """RB.lookahead (memory.py:113-160) on a ring with episode ends sprinkled in: the sampled logical
    indices (every index, so all edge cases occur: done inside / exactly at the end of the window,
    window cut by num_entries, ring wrapped when n_app > limit)."""
    rs = np.random.RandomState(seed)
    rb = M.RB(limit, (ob,), (ac,))
    tr = dict(obs0=rs.randn(n_app, ob).astype(np.float32), acs=rs.randn(n_app, ac).astype(np.float32),
              rews=rs.rand(n_app).astype(np.float32) * 5, obs1=rs.randn(n_app, ob).astype(np.float32),
              dones1=(rs.rand(n_app) < p_done).astype(np.float32))
    for i in range(n_app):
        rb.append(tr['obs0'][i], tr['acs'][i], tr['rews'][i], tr['obs1'][i], tr['dones1'][i])
    idxs = np.arange(rb.num_entries)
    la = rb.lookahead(transitions={'idxs': idxs}, n=n, gamma=gamma)
    out = {'in_' + k: v for k, v in tr.items()}
    out.update(limit=limit, ob=ob, ac=ac, n=n, gamma=gamma, idxs=idxs, start=rb.ring_buffers['obs0'].start,
               length=rb.ring_buffers['obs0'].length)
    for k in ('obs0', 'obs1', 'acs', 'rews', 'dones1', 'td_len'):
        out['la_' + k] = np.asarray(la[k])
    return out


def main():
    np.savez_compressed(os.path.join(HERE, 'replay_nstep_a.npz'), **gen_nstep(64, 3, 2, 31, 50, 5, 0.99, 0.15))
    np.savez_compressed(os.path.join(HERE, 'replay_nstep_wrap.npz'),
                        **gen_nstep(48, 3, 2, 32, 110, 7, 0.97, 0.1))
    np.savez_compressed(os.path.join(HERE, 'replay_tree_cap16.npz'), **gen_tree_ops(16, 1, 40, 24))
    np.savez_compressed(os.path.join(HERE, 'replay_tree_cap1024.npz'),
                        **gen_tree_ops(1024, 2, 1500, 64))
    np.savez_compressed(os.path.join(HERE, 'replay_per_a.npz'),
                        **gen_per(200, 3, 2, 0.3, 1.0, 11, 150, 8, 32, 20))
    np.savez_compressed(os.path.join(HERE, 'replay_per_b.npz'),
                        **gen_per(1000, 5, 3, 0.6, 0.4, 12, 300, 6, 64, 150))
    np.savez_compressed(os.path.join(HERE, 'replay_per_demos.npz'),
                        **gen_per(64, 3, 2, 0.3, 1.0, 13, 30, 6, 16, 15, n_demos=10))
    np.savez_compressed(os.path.join(HERE, 'replay_rb.npz'), **gen_rb(100, 4, 2, 21, 130, 16, 40))
    np.savez_compressed(os.path.join(HERE, 'replay_rb_partial.npz'),
                        **gen_rb(100, 4, 2, 22, 60, 16, 40))
    print('golden replay vectors written to', HERE)


if __name__ == '__main__':
    main()
