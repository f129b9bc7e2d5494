"""HIP replay / segment-tree kernels vs the golden vectors recorded from the reference and vs the
C oracle at full capacity.  Bit-exact for indices, tree nodes, sums AND the importance
weights (tolerance: none): their pows go through the device routine that carries the host pow's bits
(csrc/pow_bits.h, tests/test_gpu_pow.py)."""
import os
import random

import numpy as np
import pytest
import torch

from tests import replay_cases as RC

pytestmark = pytest.mark.gpu


def _dev_trees(cap):
    from sam_tf_amd import ops
    t = ops.DeviceTrees(cap)

    def t_set(i, v):
        t.set(torch.tensor([i], dtype=torch.int64, device='cuda'),
              torch.tensor([v], dtype=torch.float64, device='cuda'))

    def prefix(s):
        return int(t.find_prefixsum(torch.tensor([s], dtype=torch.float64, device='cuda')).item())
    return (t_set, lambda a, b: t.reduce('sum', a, b), lambda a, b: t.reduce('min', a, b), prefix,
            lambda kind: (t.sum if kind == 'sum' else t.min).cpu().numpy())


@pytest.mark.parametrize('name', ['replay_tree_cap16', 'replay_tree_cap1024'])
def test_tree_golden(golden_dir, name):
    RC.run_tree_fixture(np.load(os.path.join(golden_dir, name + '.npz')), _dev_trees)


def test_tree_known_answers():
    """The reference's own unit tests (segment_tree_test.py:6-95) through the C ABI."""
    from sam_tf_amd import ops
    t = ops.DeviceTrees(4)
    dev = 'cuda'

    def put(i, v):
        t.set(torch.tensor([i], device=dev), torch.tensor([v], dtype=torch.float64, device=dev))
    put(2, 1.0); put(3, 3.0)
    assert t.reduce('sum') == 4.0 and t.reduce('sum', 0, 2) == 0.0 and t.reduce('sum', 0, 3) == 1.0
    assert t.reduce('sum', 2, 3) == 1.0 and t.reduce('sum', 2, -1) == 1.0 and t.reduce('sum', 2, 4) == 4.0
    pre = torch.tensor([0.0, 0.5, 0.99, 1.01, 3.0, 4.0], dtype=torch.float64, device=dev)
    assert t.find_prefixsum(pre).tolist() == [2, 2, 2, 3, 3, 3]
    m = ops.DeviceTrees(4)

    def putm(i, v):
        m.set(torch.tensor([i], device=dev), torch.tensor([v], dtype=torch.float64, device=dev))
    putm(0, 1.0); putm(2, 0.5); putm(3, 3.0)
    assert m.reduce('min') == 0.5 and m.reduce('min', 0, 2) == 1.0 and m.reduce('min', 3, 4) == 3.0
    putm(2, 4.0)
    assert m.reduce('min') == 1.0 and m.reduce('min', 2, 4) == 3.0 and m.reduce('min', 2, -1) == 4.0


@pytest.mark.parametrize('name', ['replay_per_a', 'replay_per_b', 'replay_per_demos'])
def test_per_golden(golden_dir, name):
    from sam_tf_amd.memory import PrioritizedRB
    fx = np.load(os.path.join(golden_dir, name + '.npz'))
    rb = RC.run_per_fixture(
        fx, lambda limit, ob, ac, alpha, beta: PrioritizedRB(limit, (ob,), (ac,), alpha, beta),
        iws_exact=True)
    assert np.array_equal(rb.store['obs0'].cpu().numpy(), fx['final_obs0'])
    assert np.array_equal(rb.store['rews'].cpu().numpy().ravel(), fx['final_rews'].ravel())
    assert rb.start == int(fx['final_start']) and rb.length == int(fx['final_length'])


@pytest.mark.parametrize('name', ['replay_rb', 'replay_rb_partial'])
def test_rb_golden(golden_dir, name):
    """(numpy batches: the gather kernel stores the rows straight into pinned host memory, memory.RB.batchify)"""
    from sam_tf_amd.memory import RB
    RC.run_rb_fixture(np.load(os.path.join(golden_dir, name + '.npz')),
                      lambda limit, ob, ac: RB(limit, (ob,), (ac,)))


@pytest.mark.parametrize('n,dim', [(1, 11), (512, 11), (37, 5), (2048, 376)])
def test_obs_rms_update_numpy_form(n, dim):
    """MpiRunningMeanStd.update(x, comm) with numpy rows (mpi_running_mean_std.py:47-59): one row (store_transition),
    an even batch (one launch: sums + commit), an odd batch, a batch beyond the pinned-read size."""
    from sam_tf_amd.running_mean_std import DeviceRunningMeanStd
    from oracle import nets_oracle as NO
    rs = np.random.RandomState(n)
    r = DeviceRunningMeanStd(shape=(dim,))
    exp = NO.rms_init(dim)
    for _ in range(3):
        x = (rs.randn(n, dim) * 2 + 0.5).astype(np.float64 if n == 37 else np.float32)
        r.update(x, None)
        exp = NO.rms_update(exp, x.astype(np.float32), dim)
    np.testing.assert_allclose(r.sums.cpu().numpy(), exp, rtol=1e-12)
    em, es = NO.rms_mean_std(exp, dim)
    np.testing.assert_allclose(r.mean.cpu().numpy(), em, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r.std.cpu().numpy(), es, rtol=1e-6)


@pytest.mark.parametrize('cap_log2,B', [(20, 256), (22, 8192), (10, 1)])
def test_per_full_capacity_vs_c_oracle(cap_log2, B):
    """BASELINE configs C3 (1M, B=256) and C5 (4M, B=8192): sample -> update rounds, every index,
    every tree node bit-equal to oracle/per_tree.c.  Duplicates inside one update batch occur
    (stratified sampling of a skewed tree) and must resolve last-writer-wins."""
    from oracle.per_tree_c import PerTreesC
    from sam_tf_amd import ops
    cap = 1 << cap_log2
    n = cap - 7 if cap_log2 > 10 else 600
    rs = np.random.RandomState(cap_log2)
    ct = PerTreesC(cap)
    ct.fill_range(0, n, 1.0 ** 0.3)
    dt = ops.DeviceTrees(cap)
    dt.set(torch.arange(n, device='cuda'), torch.full((n,), 1.0 ** 0.3, dtype=torch.float64, device='cuda'))
    max_c = max_d = 1.0
    for r in range(4):
        u = rs.rand(B)
        ic, wc = ct.per_sample(n, u, 1.0)
        idd, wd = dt.per_sample(n, torch.as_tensor(u).cuda(), 1.0)
        assert np.array_equal(ic, idd.cpu().numpy()), 'round %d idxs' % r
        assert np.array_equal(wd.cpu().numpy(), wc), 'round %d importance weights' % r
        # heavy-tailed priorities make later rounds revisit the same leaves -> duplicates
        pr = np.abs(rs.standard_cauchy(B)) + 1e-6
        if B > 1:
            ic[B // 2] = ic[0]  # force one duplicate pair with different priorities
            idd = torch.as_tensor(ic).cuda()
        max_c = ct.per_update(ic, pr, 0.3, max_priority=max_c)
        leaf, max_d = ops.per_leaf_values_host(ic, pr, 0.3, 0, 0.1, max_d)
        dt.set(idd, torch.as_tensor(leaf).cuda())
        assert max_c == max_d
        assert dt.reduce('sum', 0, n) == ct.sum(0, n)
        assert dt.reduce('sum', 0, n - 1) == ct.sum(0, n - 1)
        assert dt.reduce('min', 0, n) == ct.min(0, n)
        assert dt.reduce('sum', 12345 % n, n - 3) == ct.sum(12345 % n, n - 3)
    assert np.array_equal(dt.sum.cpu().numpy(), ct.sumv)
    assert np.array_equal(dt.min.cpu().numpy(), ct.minv)
    assert int((dt.stamp != -1).sum().item()) == 0


@pytest.mark.parametrize('pattern', ['uniform', 'one_subtree', 'edges'])
def test_tree_set_large_batch_subtree_partition(pattern):
    """Batches of >= 1024 updates take the subtree-partitioned form of samx_tree_set (256 workgroups own
    disjoint subtrees, the top 8 levels are recomputed in a second launch): every node bit-equal to
    oracle/per_tree.c, for spread-out updates, updates that all fall into one subtree (one workgroup
    gets the whole batch) and updates on the first/last leaves, each with duplicates."""
    from oracle.per_tree_c import PerTreesC
    from sam_tf_amd import ops
    cap, B = 1 << 14, 2048
    rs = np.random.RandomState(5)
    ct = PerTreesC(cap)
    ct.fill_range(0, cap, 0.7)
    dt = ops.DeviceTrees(cap)
    dt.set(torch.arange(1000, device='cuda'), torch.full((1000,), 0.7, dtype=torch.float64, device='cuda'))
    dt.set(torch.arange(1000, cap, device='cuda'),
           torch.full((cap - 1000,), 0.7, dtype=torch.float64, device='cuda'))   # one-workgroup form
    mp_c = mp_d = 1.0
    for r in range(3):
        if pattern == 'uniform':
            idx = rs.randint(0, cap, size=B)
        elif pattern == 'one_subtree':
            idx = rs.randint(5, 40, size=B)          # 35 leaves, ~58 writers each
        else:
            idx = np.concatenate([np.zeros(B // 4, dtype=np.int64), np.full(B // 4, cap - 1),
                                  rs.randint(0, cap, size=B // 2)])
            rs.shuffle(idx)
        idx = idx.astype(np.int64)
        pr = np.abs(rs.standard_cauchy(B)) + 1e-6
        mp_c = ct.per_update(idx, pr, 0.3, max_priority=mp_c)
        leaf, mp_d = ops.per_leaf_values_host(idx, pr, 0.3, 0, 0.1, mp_d)
        dt.set(torch.as_tensor(idx).cuda(), torch.as_tensor(leaf).cuda())
        assert np.array_equal(dt.sum.cpu().numpy(), ct.sumv), (pattern, r)
        assert np.array_equal(dt.min.cpu().numpy(), ct.minv), (pattern, r)
    assert int((dt.stamp != -1).sum().item()) == 0


def test_per_rejects_degenerate_sizes():
    from sam_tf_amd import ops, _lib
    dt = ops.DeviceTrees(8)
    u = torch.rand(4, dtype=torch.float64, device='cuda')
    with pytest.raises(_lib.SamxError):
        dt.per_sample(1, u, 1.0)  # num_entries == 1: reference issues reduce(0, -1)
    with pytest.raises(_lib.SamxError):
        dt.reduce('sum', 3, 3)  # empty range


def test_gather_roundtrip_and_rms():
    """gather == indexing; running sums == float64 numpy column sums; mean/std formula of
    mpi_running_mean_std.py:31-34 (tolerance 1e-6 relative on the float32 outputs)."""
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    rs = np.random.RandomState(3)
    O = 376
    x1 = rs.randn(1024, O).astype(np.float32) * 3 + 1
    x2 = rs.randn(1000, O).astype(np.float32)
    add = torch.empty(2 * O + 1, dtype=torch.float64, device='cuda')
    ops.rms_batch_sums(torch.as_tensor(x1).cuda(), torch.as_tensor(x2).cuda(), add)
    sums0 = NO.rms_init(O)
    exp = NO.rms_update(sums0, np.concatenate([x1, x2]), O)
    np.testing.assert_allclose(add.cpu().numpy(), exp - sums0, rtol=1e-12)
    sums = torch.as_tensor(sums0).cuda()
    mean = torch.empty(O, device='cuda'); std = torch.empty(O, device='cuda')
    ops.rms_commit(sums, add, mean, std)
    np.testing.assert_allclose(sums.cpu().numpy(), exp, rtol=1e-12)
    em, es = NO.rms_mean_std(exp, O)
    np.testing.assert_allclose(mean.cpu().numpy(), em, rtol=1e-6)
    np.testing.assert_allclose(std.cpu().numpy(), es, rtol=1e-6)
    # second commit with nothing to add only re-derives mean/std
    ops.rms_commit(sums, None, mean, std)
    np.testing.assert_allclose(mean.cpu().numpy(), em, rtol=1e-6)


@pytest.mark.parametrize('name', ['replay_nstep_a', 'replay_nstep_wrap'])
def test_nstep_lookahead_kernel_matches_reference(name):
    """samx_nstep_lookahead vs the vectors recorded from the reference's RB.lookahead
    (memory.py:113-160): indices, lengths, done flags exact; the discounted reward bit-exact in float64."""
    import os
    import torch
    from sam_tf_amd import memory as XP, ops
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'))
    rb = XP.RB(int(g['limit']), (int(g['ob']),), (int(g['ac']),))
    for i in range(len(g['in_rews'])):
        rb.append(g['in_obs0'][i], g['in_acs'][i], g['in_rews'][i], g['in_obs1'][i], g['in_dones1'][i])
    assert rb.start == int(g['start']) and rb.num_entries == int(g['length'])
    idxs = torch.as_tensor(g['idxs'].astype(np.int64)).cuda()
    end, rews, dones, tl, r64 = ops.nstep_lookahead(rb.store, rb.start, rb.num_entries, idxs, int(g['n']),
                                                    float(g['gamma']), want64=True)
    assert np.array_equal(r64.cpu().numpy(), g['la_rews'].ravel())
    assert np.array_equal(rews.cpu().numpy(), g['la_rews'].ravel().astype(np.float32))
    assert np.array_equal(dones.cpu().numpy(), g['la_dones1'].ravel().astype(np.float32))
    assert np.array_equal(tl.cpu().numpy(), g['la_td_len'].ravel().astype(np.float32))
    got = rb.batchify_device(end, want=('obs1',))['obs1'].cpu().numpy()
    assert np.array_equal(got, g['la_obs1'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['replay_nstep_a', 'replay_nstep_wrap'])
def test_nstep_lookahead_many_matches_single(name):
    """samx_nstep_lookahead_many (K batches whose indices sit inside wider index words) gives, batch by
    batch, exactly what samx_nstep_lookahead gives — which the test above pins on the reference's vectors."""
    import os
    import torch
    from sam_tf_amd import memory as XP, ops
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = np.load(os.path.join(gdir, name + '.npz'))
    rb = XP.RB(int(g['limit']), (int(g['ob']),), (int(g['ac']),))
    for i in range(len(g['in_rews'])):
        rb.append(g['in_obs0'][i], g['in_acs'][i], g['in_rews'][i], g['in_obs1'][i], g['in_dones1'][i])
    K, B, off = 5, 37, 11
    W = off + B + 3
    rng = np.random.RandomState(5)
    words = rng.randint(0, rb.num_entries, size=(K, W)).astype(np.int64)
    words[0, off:off + min(B, len(g['idxs']))] = g['idxs'].ravel()[:B]
    wd = torch.as_tensor(words).cuda()
    end = torch.zeros(K, B, dtype=torch.int64, device='cuda')
    rews, dones, tl = (torch.zeros(K, B, dtype=torch.float32, device='cuda') for _ in range(3))
    ops.nstep_lookahead_many(rb.store, rb.start, rb.num_entries, wd, off, B, int(g['n']), float(g['gamma']), end,
                             rews, dones, tl)
    for k in range(K):
        e1, r1, d1, t1 = ops.nstep_lookahead(rb.store, rb.start, rb.num_entries, wd[k, off:off + B].contiguous(),
                                             int(g['n']), float(g['gamma']))
        assert torch.equal(end[k], e1) and torch.equal(rews[k], r1) and torch.equal(dones[k], d1) \
            and torch.equal(tl[k], t1)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [7, 96, 300])
@pytest.mark.parametrize('filled', [300, 823])
def test_nstep_lookahead_on_storage_rows(n, filled):
    """samx_nstep_lookahead_many_rows (storage rows in and out, the ring's geometry from device memory) against
    samx_nstep_lookahead on the logical indices: a ring that is not full yet (start 0, windows cut by num_entries)
    and one that wrapped (start 123)."""
    from sam_tf_amd import memory as XP, ops
    limit, ob, ac = 700, 3, 2
    rs = np.random.RandomState(n + filled)
    rb = XP.RB(limit, (ob,), (ac,))
    for i in range(filled):
        rb.append(rs.randn(ob).astype(np.float32), rs.uniform(-1, 1, ac).astype(np.float32), float(rs.uniform(0, 5)),
                  rs.randn(ob).astype(np.float32), float(rs.rand() < 0.025))
    assert rb.start == max(0, filled - limit)
    K, B, off = 3, 50, 4
    W = off + B + 2
    idx = rs.randint(0, rb.num_entries, size=(K, W)).astype(np.int64)
    idx[0, off:off + 5] = np.arange(rb.num_entries - 5, rb.num_entries)   # windows that run into num_entries
    rows = (idx + rb.start) % limit
    geom = torch.as_tensor(np.array([rb.start, rb.num_entries], dtype=np.int64)).cuda()
    end = torch.zeros(K, B, dtype=torch.int64, device='cuda')
    rews, dones, tl = (torch.zeros(K, B, dtype=torch.float32, device='cuda') for _ in range(3))
    ops.nstep_lookahead_many_rows(rb.store, geom, torch.as_tensor(rows).cuda(), off, B, n, 0.99, end, rews, dones, tl)
    for k in range(K):
        e1, r1, d1, t1 = ops.nstep_lookahead(rb.store, rb.start, rb.num_entries,
                                             torch.as_tensor(idx[k, off:off + B].copy()).cuda(), n, 0.99)
        assert torch.equal(end[k], (e1 + rb.start) % limit)
        assert torch.equal(rews[k], r1) and torch.equal(dones[k], d1) and torch.equal(tl[k], t1)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [2, 7, 96, 256, 300])
def test_nstep_lookahead_window_lengths_vs_oracle(n):
    """Both lookahead kernels — sixteen lanes per sampled index for windows of up to 256 rows, one thread per index
    beyond — against the oracle's lookahead (pinned on the reference's vectors by tests/test_oracle_golden.py): a
    wrapped ring, episode ends every ~40 rows, windows cut by num_entries; end index, n-step reward (float64 and
    float32), done flag and length bit for bit."""
    from oracle.replay_oracle import ReplayOracle
    from sam_tf_amd import memory as XP, ops
    limit, ob, ac = 700, 3, 2
    rs = np.random.RandomState(n)
    rb = XP.RB(limit, (ob,), (ac,))
    ref = ReplayOracle(limit, (ob,), (ac,))
    for i in range(limit + 123):   # wraps: start != 0
        row = (rs.randn(ob).astype(np.float32), rs.uniform(-1, 1, ac).astype(np.float32), float(rs.uniform(0, 5)),
               rs.randn(ob).astype(np.float32), float(rs.rand() < 0.025))
        rb.append(*row)
        ref.append(*row)
    idxs = np.concatenate([rs.randint(0, limit, size=200), np.arange(limit - 5, limit)]).astype(np.int64)
    want = ref.lookahead(idxs, n, 0.99)
    end, rews, dones, tl, r64 = ops.nstep_lookahead(rb.store, rb.start, rb.num_entries, torch.as_tensor(idxs).cuda(),
                                                    n, 0.99, want64=True)
    assert np.array_equal(end.cpu().numpy(), np.asarray(want['end_idx']).ravel().astype(np.int64))
    assert np.array_equal(r64.cpu().numpy(), np.asarray(want['rews'], dtype=np.float64).ravel())
    assert np.array_equal(rews.cpu().numpy(), np.asarray(want['rews'], dtype=np.float64).ravel().astype(np.float32))
    assert np.array_equal(dones.cpu().numpy(), np.asarray(want['dones1']).ravel().astype(np.float32))
    assert np.array_equal(tl.cpu().numpy(), np.asarray(want['td_len']).ravel().astype(np.float32))


@pytest.mark.parametrize('limit,n_demos,seed', [(50, 0, 1), (100, 7, 2), (257, 20, 3), (64, 0, 4), (1000, 0, 5)])
def test_random_op_sequences_vs_oracle(limit, n_demos, seed):
    """Random sequences of append / sample / sample_uniform / update_priorities (duplicate indices, demo
    rows, priorities over several orders of magnitude, ring wrap-around, non-power-of-two limits) on the HIP
    PrioritizedRB and on the oracle (pinned on the reference's vectors) side by side: indices, both trees,
    range sums / minima, max_priority and the importance weights bit for bit after every operation."""
    from oracle.replay_oracle import PrioritizedReplayOracle
    from sam_tf_amd.memory import PrioritizedRB
    ob, ac = 3, 2
    alpha, beta = 0.3, 1.0
    hip = PrioritizedRB(limit, (ob,), (ac,), alpha, beta)
    ref = PrioritizedReplayOracle(limit, (ob,), (ac,), alpha, beta)
    rs = np.random.RandomState(seed)

    def row():
        return (rs.randn(ob).astype(np.float32), rs.uniform(-1, 1, ac).astype(np.float32), float(rs.uniform(0, 5)),
                rs.randn(ob).astype(np.float32), float(rs.rand() < 0.1))
    if n_demos:
        d = RC.Demos()
        rows = [row() for _ in range(n_demos)]
        d.obs0 = np.stack([r[0] for r in rows]); d.acs = np.stack([r[1] for r in rows])
        d.env_rews = np.array([r[2] for r in rows], dtype=np.float32)
        d.obs1 = np.stack([r[3] for r in rows]); d.dones1 = np.array([r[4] for r in rows], dtype=np.float32)
        hip.add_demo_transitions_to_mem(d); ref.add_demo_transitions_to_mem(d)
    for _ in range(5):
        r = row()
        hip.append(*r); ref.append(*r)

    def same_trees(what):
        assert np.array_equal(hip.tree_values('sum'), ref.tree_values('sum')), what
        assert np.array_equal(hip.tree_values('min'), ref.tree_values('min')), what
        assert hip.num_entries == ref.num_entries and float(hip.max_priority) == float(ref.max_priority), what

    for step in range(60):
        op = rs.randint(4)
        if op == 0:
            for _ in range(int(rs.randint(1, limit // 2 + 2))):
                r = row()
                hip.append(*r); ref.append(*r)
        elif op == 1 and hip.num_entries >= 2:
            B = int(rs.choice([1, 4, 16, 33]))
            random.seed(1000 * seed + step); a = hip.sample(B)
            random.seed(1000 * seed + step); b = ref.sample(B)
            assert np.array_equal(np.asarray(a['idxs'], dtype=np.int64), np.asarray(b['idxs'], dtype=np.int64)), step
            assert np.array_equal(np.asarray(a['iws'], dtype=np.float64).ravel(), b['iws'].ravel()), step
            assert np.array_equal(np.asarray(a['obs0']), b['obs0']) and np.array_equal(np.asarray(a['rews']), b['rews'])
            # priorities over many orders of magnitude, with duplicate indices
            idxs = np.asarray(b['idxs'], dtype=np.int64)
            if B > 1:
                idxs[-1] = idxs[0]
            prios = np.abs(rs.randn(B)) * 10.0 ** rs.uniform(-6, 3, B) + 1e-6
            hip.update_priorities(idxs, prios); ref.update_priorities(idxs, prios)
        elif op == 2 and hip.num_entries >= 2:
            np.random.seed(2000 * seed + step); a = hip.sample_uniform(8)
            np.random.seed(2000 * seed + step); b = ref.sample_uniform(8)
            assert np.array_equal(np.asarray(a['idxs']), np.asarray(b['idxs']))
            assert np.array_equal(np.asarray(a['obs1']), b['obs1'])
        else:
            n = hip.num_entries
            lo = int(rs.randint(0, n)); hi = int(rs.randint(lo + 1, n + 1))   # [lo, hi): never empty
            assert hip.tree_sum(lo, hi) == ref.tree_sum(lo, hi) and hip.tree_min(lo, hi) == ref.tree_min(lo, hi)
        same_trees((step, op))


@pytest.mark.parametrize('pattern', ['spread', 'one_subtree', 'siblings', 'dups', 'single', 'edges'])
@pytest.mark.parametrize('cap_log2', [4, 6, 11, 12, 13, 17, 20])
def test_row_head_from_td_errors_vs_c_oracle(cap_log2, pattern):
    """samx_per_row_prologue_td (the prioritized row head of the device refresh: subtree workgroups walk the
    lower levels in LDS, the sampling workgroups rebuild the top) and samx_tree_set's one-workgroup form of the
    same walk, against oracle/per_tree.c: every node of both trees, the sampled indices, the importance weights,
    max_priority — bit for bit.  Capacities on both sides of the 12 LDS levels; batches that land in ONE subtree
    (one workgroup carries everything, several waves stay alive), on sibling leaves (both children touched at
    every level), on the first/last leaves, with heavy duplication (last position wins), and of one entry."""
    from oracle.per_tree_c import PerTreesC
    from sam_tf_amd import ops
    cap = 1 << cap_log2
    n_ent = cap - 3
    rs = np.random.RandomState(100 * cap_log2 + len(pattern))
    ct = PerTreesC(cap)
    dt = ops.DeviceTrees(cap)
    # a populated tree: priorities over several orders of magnitude (host leaves on both sides)
    max_c = max_d = 1.0
    for lo in range(0, n_ent, 800):
        ii = np.arange(lo, min(lo + 800, n_ent), dtype=np.int64)
        pr = np.abs(rs.randn(ii.size)) * 10.0 ** rs.uniform(-3, 2, ii.size) + 1e-6
        max_c = ct.per_update(ii, pr, 0.6, max_priority=max_c)
        leaf, max_d = ops.per_leaf_values_host(ii, pr, 0.6, 0, 0.1, max_d)
        dt.set(torch.as_tensor(ii).cuda(), torch.as_tensor(leaf).cuda())   # < 1024 entries: the one-workgroup walk
    assert np.array_equal(dt.sum.cpu().numpy(), ct.sumv) and np.array_equal(dt.min.cpu().numpy(), ct.minv)
    state = ops.make_per_state(max_d)
    B = 64
    idx_out = torch.empty(B, dtype=torch.int64, device='cuda')
    iws = torch.empty(B, dtype=torch.float64, device='cuda')
    iws32 = torch.empty(B, dtype=torch.float32, device='cuda')
    dummy = torch.zeros(4, dtype=torch.int64, device='cuda')
    for r in range(4):
        n = {'single': 1, 'dups': 300}.get(pattern, [5, 64, 257, 1000][r])
        n = min(n, 1023)
        if pattern == 'spread':
            idx = rs.randint(0, n_ent, n)
        elif pattern == 'one_subtree':
            w = max(2, cap // 16)
            base = (r % 16) * w
            idx = base + rs.randint(0, min(w, max(1, n_ent - base)), n)
        elif pattern == 'siblings':
            first = rs.randint(0, max(1, n_ent - n - 1))
            idx = (first + np.arange(n)) % n_ent
        elif pattern == 'dups':
            idx = rs.randint(0, n_ent, 7)[rs.randint(0, 7, n)]
        elif pattern == 'single':
            idx = rs.randint(0, n_ent, 1)
        else:
            idx = np.concatenate([np.arange(min(3, n_ent)), n_ent - 1 - np.arange(min(3, n_ent))])[:max(1, min(n, 6))]
        idx = idx.astype(np.int64)
        td = (rs.standard_cauchy(idx.size) * 10.0 ** rs.uniform(-5, 2, idx.size)).astype(np.float32)
        pr = (np.abs(td) + np.float32(1e-6)).astype(np.float64)      # sam_agent.py:702 in float32, then widened
        max_c = ct.per_update(idx, pr, 0.6, max_priority=max_c)
        u = rs.rand(B)
        ic, wc = ct.per_sample(n_ent, u, 0.7)
        ops.per_row_prologue(dt, torch.as_tensor(idx).cuda(), None, None, 0, dummy, B, None, None,
                             sample=(n_ent, torch.as_tensor(u).cuda(), 0.7, idx_out, iws, iws32),
                             td=(torch.as_tensor(td).cuda(), 0.6, 0, 0.0, state))
        what = '%s cap 2^%d round %d' % (pattern, cap_log2, r)
        assert np.array_equal(dt.sum.cpu().numpy(), ct.sumv), what
        assert np.array_equal(dt.min.cpu().numpy(), ct.minv), what
        assert np.array_equal(idx_out.cpu().numpy(), ic), what
        assert np.array_equal(iws.cpu().numpy(), wc), what
        st = state.cpu().numpy()
        assert st[0] == max_c and st.view(np.uint64)[1] == 0, what


@pytest.mark.gpu
@pytest.mark.parametrize('B,cap_log2', [(256, 12), (37, 9), (2048, 14)])
def test_per_geometry_words_equal_by_value_forms(B, cap_log2):
    """samx_per_sample_rows / samx_per_row_prologue_td_rows / samx_per_leaves_from_td_geom read {ring start,
    num_entries} from device words (captured graphs survive appends): same indices, weights, leaves, trees and
    state bits as the by-value forms, plus the storage rows (start + idx) % limit (memory.py:33-34)."""
    from sam_tf_amd import ops
    cap = 1 << cap_log2
    limit, n_ent, start = cap - 3, cap - 3, 41
    rs = np.random.RandomState(B)
    leaves = np.zeros(cap)
    leaves[:n_ent] = rs.rand(n_ent) ** 3 + 1e-3
    mk = lambda: ops.DeviceTrees(cap)
    ta, tb = mk(), mk()
    all_idx = torch.arange(cap, dtype=torch.int64, device='cuda')
    for t in (ta, tb):
        t.set(all_idx[:n_ent], torch.as_tensor(leaves[:n_ent]).cuda())
    u = torch.as_tensor(rs.rand(B)).cuda()
    geom = torch.tensor([start, n_ent], dtype=torch.int64, device='cuda')
    z = lambda dt: torch.zeros(B, dtype=dt, device='cuda')
    i0, w0, f0 = ta.per_sample(n_ent, u, 0.6, iws32_out=z(torch.float32))[0], None, None
    i0, w0 = ta.per_sample(n_ent, u, 0.6)
    i1, w1, f1, rows, cp = z(torch.int64), z(torch.float64), z(torch.float32), z(torch.int64), z(torch.int64)
    ops.per_sample_rows(tb, geom, limit, u, 0.6, i1, w1, f1, rows, idx_copy_out=cp)
    assert torch.equal(i0, i1) and torch.equal(w0, w1) and torch.equal(cp, i1)
    assert torch.equal(rows, (i1 + start) % limit) and torch.equal(f1, w1.float())
    # leaves from TD errors: an index at / beyond num_entries raises bit 1 in both forms
    td = torch.as_tensor((rs.randn(B) * 2).astype(np.float32)).cuda()
    idx = i0.clone()
    idx[0] = -1                                   # nothing pending
    sa, sb = ops.make_per_state(1.0), ops.make_per_state(1.0)
    la, lb = z(torch.float64), z(torch.float64)
    ops.per_leaves_from_td(td, idx, cap, n_ent, 0.3, 0, 0.1, la, sa)
    ops.per_leaves_from_td_geom(td, idx, cap, geom, 0.3, 0, 0.1, lb, sb)
    assert torch.equal(la, lb) and torch.equal(sa, sb)
    if B < 1024:   # the two-launch row head: refresh + walk + top rebuild + sample
        ops.per_row_prologue(ta, idx, None, None, 0, u, B, None, None, sample=(n_ent, u, 0.6, i0, w0, None),
                             td=(td, 0.3, 0, 0.1, sa))
        ops.per_row_prologue_rows(tb, idx, td, 0.3, 0, 0.1, sb, geom, limit, u, 0.6, i1, w1, f1, rows, idx_copy_out=cp)
        assert torch.equal(ta.sum, tb.sum) and torch.equal(ta.min, tb.min) and torch.equal(sa, sb)
        assert torch.equal(i0, i1) and torch.equal(w0, w1) and torch.equal(cp, i1)
        assert torch.equal(rows, (i1 + start) % limit)
