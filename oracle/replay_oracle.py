"""CPU oracle for the replay-memory rows of the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sam_tf_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as
the checker.  The product path runs on HIP kernels and fails loudly without them.

This is a restatement (own code, array based) of the algorithms in the reference:

* segment trees ............ /root/reference/imitation/helpers/segment_tree.py:4-109
* ring buffer / RB ......... /root/reference/imitation/imitation_algorithms/memory.py:16-213
* prioritized replay ....... /root/reference/imitation/imitation_algorithms/memory.py:216-390
* tree capacity helper ..... /root/reference/imitation/imitation_algorithms/memory.py:494-502

Parity is PINNED: ``tests/golden/replay_*.npz`` were produced by importing the reference modules in
the build container (``tests/golden/make_golden_replay.py``) and ``tests/test_oracle_replay.py``
replays them through this file bit for bit, together with the reference's own known-answer tests
(/root/reference/imitation/helpers/segment_tree_test.py:6-95).

Arithmetic contract (what "bit-exact" means here).  The reference pins numpy==1.16.1
(requirements.txt:31).  There, ``np.float32(p) ** 0.3`` and ``np.float32(p) + 0.1`` promote to
float64, so every tree node is an IEEE double and every leaf is libm ``pow((double)p, alpha)``.
Under numpy>=2 (NEP 50) the same expressions would stay float32; the oracle (and the golden
vectors, which feed float64 priorities to the reference) follow the numpy-1.16 behaviour the
reference was written against.
"""
import math
import random

import numpy as np


def tree_capacity(limit):
    """Smallest power of two >= limit (memory.py:494-502)."""
    cap = 1
    while cap < limit:
        cap *= 2
    return cap


class TreeOracle(object):
    """Array-backed binary tree over ``cap`` leaves; node i has children 2i, 2i+1; leaves at cap+i.

    kind 'sum' -> op add, neutral 0.0 ; kind 'min' -> op min, neutral +inf  (segment_tree.py:71-109).
    """

    def __init__(self, cap, kind):
        assert cap > 0 and (cap & (cap - 1)) == 0, "capacity must be a power of two"
        assert kind in ('sum', 'min')
        self.cap = cap
        self.kind = kind
        neutral = 0.0 if kind == 'sum' else float('inf')
        self.v = np.full(2 * cap, neutral, dtype=np.float64)

    def _op(self, a, b):
        if self.kind == 'sum':
            return a + b
        # Python's min(a, b) returns a unless b < a (segment_tree.py:104)
        return b if b < a else a

    # -- point write: leaf then every ancestor (segment_tree.py:55-64)
    def set(self, idx, val):
        node = idx + self.cap
        self.v[node] = val
        node //= 2
        while node >= 1:
            self.v[node] = self._op(self.v[2 * node], self.v[2 * node + 1])
            node //= 2

    def get(self, idx):
        assert 0 <= idx < self.cap
        return self.v[self.cap + idx]

    # -- range query, in the association order of the reference recursion (segment_tree.py:26-53)
    def _walk(self, lo, hi, node, nlo, nhi):
        if lo == nlo and hi == nhi:
            return self.v[node]
        mid = (nlo + nhi) // 2
        if hi <= mid:
            return self._walk(lo, hi, 2 * node, nlo, mid)
        if lo >= mid + 1:
            return self._walk(lo, hi, 2 * node + 1, mid + 1, nhi)
        left = self._walk(lo, mid, 2 * node, nlo, mid)
        right = self._walk(mid + 1, hi, 2 * node + 1, mid + 1, nhi)
        return self._op(left, right)

    def reduce(self, start=0, end=None):
        """op over leaves [start, end) ; end None -> cap ; end < 0 wraps by +cap."""
        if end is None:
            end = self.cap
        if end < 0:
            end += self.cap
        return self._walk(start, end - 1, 1, 0, self.cap - 1)

    # -- prefix-sum descent (segment_tree.py:82-97); ties (left == s) go RIGHT
    def find_prefixsum_idx(self, s):
        assert self.kind == 'sum'
        assert 0 <= s <= self.reduce() + 1e-5
        node = 1
        while node < self.cap:
            left = self.v[2 * node]
            if left > s:
                node = 2 * node
            else:
                s = s - left
                node = 2 * node + 1
        return node - self.cap


class RingOracle(object):
    """One float32 ring per transition atom (memory.py:16-57)."""

    def __init__(self, maxlen, shape):
        self.maxlen = maxlen
        self.start = 0
        self.length = 0
        self.data = np.zeros((maxlen,) + tuple(shape), dtype=np.float32)

    def rows(self, idxs):
        return self.data[(self.start + idxs) % self.maxlen]

    def write_slot(self, offset):
        """Advance the ring by one element and return the physical row to write.

        ``offset`` rows at the bottom are pinned demos that the FIFO never evicts."""
        span = self.maxlen - offset
        if self.length < self.maxlen:
            self.length += 1
            return offset + (self.start + self.length - 1) % span
        assert self.length == self.maxlen
        self.start = (self.start + 1) % span
        return offset + (self.start + self.length - offset - 1) % span

    def append(self, v, offset):
        self.data[self.write_slot(offset)] = v


ATOMS = ('obs0', 'acs', 'rews', 'dones1', 'obs1')


def _min2d(x):
    x = np.array(x)
    return x if x.ndim >= 2 else x.reshape(-1, 1)


class ReplayOracle(object):
    """Uniform replay (memory.py:60-213)."""

    def __init__(self, limit, ob_shape, ac_shape):
        self.limit = limit
        self.num_demos = 0
        shapes = (ob_shape, ac_shape, (1,), (1,), ob_shape)
        self.rings = {k: RingOracle(limit, s) for k, s in zip(ATOMS, shapes)}

    @property
    def num_entries(self):
        return self.rings['obs0'].length

    @property
    def latest_entry_idx(self):
        r = self.rings['obs0']
        return (r.start + r.length - 1) % r.maxlen

    def append(self, obs0, acs, rews, obs1, dones1, is_demo=False):
        off = 0 if is_demo else self.num_demos
        for k, val in zip(('obs0', 'acs', 'rews', 'obs1', 'dones1'),
                          (obs0, acs, rews, obs1, dones1)):
            self.rings[k].append(val, off)

    def add_demo_transitions_to_mem(self, dset):
        """Pin expert transitions at the bottom of every ring (memory.py:186-198)."""
        assert self.num_entries == 0 and self.num_demos == 0
        for tr in zip(dset.obs0, dset.acs, dset.env_rews, dset.obs1, dset.dones1):
            self.append(*tr, is_demo=True)
            self.num_demos += 1
        assert self.num_demos == self.num_entries

    def batchify(self, idxs):
        out = {k: _min2d(self.rings[k].rows(idxs)) for k in ATOMS}
        out['idxs'] = idxs
        return out

    def sample(self, batch_size, idxs=None):
        """memory.py:86-93 — ``idxs`` None draws np.random.randint exactly like the reference."""
        if idxs is None:
            idxs = np.random.randint(low=0, high=self.num_entries, size=batch_size)
        return self.batchify(idxs)

    def lookahead(self, idxs, n, gamma):
        """n-step TD lookahead from every sampled transition (memory.py:113-160).  Per index: the
        window [idx, min(idx+n, num_entries)-1] of LOGICAL indices, cut after the first done1 == 1
        inside it; rews = discounted sum over the kept rewards, accumulated in float64 from the far
        end (math_util.py:32-43: scipy lfilter on the reversed array = y[t] = x[t] + gamma*y[t+1]);
        obs1 of the last kept row; dones1 = 1 only when the cut happened BEFORE the window's end
        (a done exactly at the window's end yields 0 — the reference's quirk); td_len = rows kept."""
        idxs = np.asarray(idxs)
        out = {k: [] for k in ('obs0', 'acs', 'rews', 'dones1', 'obs1', 'td_len', 'end_idx')}
        for idx in idxs:
            idx = int(idx)
            end = min(idx + n, self.num_entries) - 1
            win = np.arange(idx, end + 1)
            dones = self.rings['dones1'].rows(win).ravel()
            hit = np.nonzero(dones == 1.0)[0]
            ep_end = idx + int(hit[0]) if hit.size else end
            td_len = ep_end - idx + 1
            rews = self.rings['rews'].rows(win).ravel()[:td_len]
            acc = 0.0
            for r in rews[::-1]:
                acc = float(r) + gamma * acc
            out['obs0'].append(self.rings['obs0'].rows(np.array([idx]))[0])
            out['acs'].append(self.rings['acs'].rows(np.array([idx]))[0])
            out['obs1'].append(self.rings['obs1'].rows(np.array([ep_end]))[0])
            out['rews'].append(acc)
            out['dones1'].append(0.0 if ep_end == end else 1.0)
            out['td_len'].append(td_len)
            out['end_idx'].append(ep_end)
        return {k: _min2d(v) for k, v in out.items()}

    def recent_window(self, window):
        """Logical indices of the ``window`` newest entries, oldest first (memory.py:99-106)."""
        width = min(window, self.num_entries)
        base = np.full(width, float(self.latest_entry_idx))
        base -= np.arange(width - 1, -1, -1)
        base += self.limit
        base %= self.limit
        return base.astype(int)

    def sample_recent(self, batch_size, window, picks=None):
        cand = self.recent_window(window)
        if picks is None:
            idxs = np.random.choice(cand, size=batch_size)
        else:
            idxs = cand[picks]
        return self.batchify(idxs)


class PrioritizedReplayOracle(ReplayOracle):
    """Proportional PER, non-ranked (memory.py:216-390)."""

    def __init__(self, limit, ob_shape, ac_shape, alpha, beta, demos_eps=0.1, max_priority=1.0):
        super().__init__(limit, ob_shape, ac_shape)
        assert 0. <= alpha <= 1. and beta > 0
        self.alpha = float(alpha)
        self.beta = float(beta)
        self.max_priority = float(max_priority)
        self.demos_eps = float(demos_eps)
        self.cap = tree_capacity(limit)
        self.sum_tree = TreeOracle(self.cap, 'sum')
        self.min_tree = TreeOracle(self.cap, 'min')

    def leaf_value(self, priority):
        # Python float ** float -> libm pow() on doubles (memory.py:341-342,376-377)
        return math.pow(float(priority), self.alpha)

    def append(self, *args, **kwargs):
        super().append(*args, **kwargs)
        idx = self.latest_entry_idx
        leaf = self.leaf_value(self.max_priority)
        self.sum_tree.set(idx, leaf)
        self.min_tree.set(idx, leaf)

    def stratified_prefixes(self, batch_size, u=None):
        """memory.py:266-280 : note p_total EXCLUDES the newest entry (end = num_entries - 1)."""
        assert self.num_entries
        p_total = self.sum_tree.reduce(0, self.num_entries - 1)
        piece = p_total / batch_size
        if u is None:
            u = [random.random() for _ in range(batch_size)]
        return np.array([(float(u[i]) * piece) + (i * piece) for i in range(batch_size)],
                        dtype=np.float64)

    def sample_idxs(self, batch_size, u=None):
        pref = self.stratified_prefixes(batch_size, u)
        return np.array([self.sum_tree.find_prefixsum_idx(s) for s in pref])

    def importance_weights(self, idxs):
        """memory.py:295-309 : here the sums INCLUDE the newest entry (end = num_entries)."""
        n = self.num_entries
        total = self.sum_tree.reduce(0, n)
        lowest = self.min_tree.reduce(0, n) / total
        max_w = (n * lowest) ** (-self.beta)
        iws = []
        for idx in idxs:
            prob = self.sum_tree.get(idx) / total
            iws.append(((n * prob) ** (-self.beta)) / max_w)
        return np.array(iws, dtype=np.float64)

    def sample(self, batch_size, u=None):
        idxs = self.sample_idxs(batch_size, u)
        iws = self.importance_weights(idxs)
        out = self.batchify(idxs)
        out['iws'] = _min2d(iws)
        out['idxs'] = np.array(idxs)
        return out

    def sample_uniform(self, batch_size, idxs=None):
        return super().sample(batch_size, idxs)

    # accessors shared with the HIP shim so one scenario runner drives both (tests/replay_cases.py)
    def tree_sum(self, start=0, end=None):
        return self.sum_tree.reduce(start, end)

    def tree_min(self, start=0, end=None):
        return self.min_tree.reduce(start, end)

    def tree_values(self, kind):
        return (self.sum_tree if kind == 'sum' else self.min_tree).v.copy()

    def update_priorities(self, idxs, priorities):
        """Sequential, last writer wins on duplicate indices (memory.py:369-379)."""
        assert len(idxs) == len(priorities)
        for idx, p in zip(idxs, priorities):
            p = float(p)
            assert p > 0
            assert 0 <= idx < self.num_entries
            if idx < self.num_demos:
                p += self.demos_eps
            leaf = self.leaf_value(p)
            self.sum_tree.set(int(idx), leaf)
            self.min_tree.set(int(idx), leaf)
            self.max_priority = max(p, self.max_priority)
