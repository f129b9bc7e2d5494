"""Device-resident replay memory with the reference's API
(/root/reference/imitation/imitation_algorithms/memory.py:60-390: RB, PrioritizedRB non-ranked).

Layout in HBM: five float32 arrays [limit, dim] (obs0, acs, rews, dones1, obs1) — one row per
transition, the same structure-of-arrays the reference keeps in numpy — plus, for PER, two float64
trees of 2*cap nodes and an int32[cap] stamp array (include/samx.h).  The ring arithmetic
(start/length/demo offset) is a few host integers, exactly the reference's; rows are gathered and
trees are walked by HIP kernels.  Host RNG draws (np.random.randint, random.random) are made with the
same generators in the same order as the reference, so seeded runs sample the same indices.

Not built (out of scope per SURVEY.md §2): ranked PER, UnrealRB.  n-step lookahead is a "next" row.
"""
import random

import numpy as np
import torch

from . import _lib
from . import ops

ATOMS = ('obs0', 'acs', 'rews', 'dones1', 'obs1')


def segment_tree_capacity(limit):
    """memory.py:494-502."""
    cap = 1
    while cap < limit:
        cap *= 2
    return cap


def array_min2d(x):
    x = np.array(x)
    return x if x.ndim >= 2 else x.reshape(-1, 1)


_MT_LOAN = np.random.RandomState(0)   # scratch generator for PrioritizedRB.draw_uniforms (state is overwritten)

class RB(object):
    """Uniform replay (memory.py:60-213) over device storage."""

    def __init__(self, limit, ob_shape, ac_shape, device='cuda'):
        assert len(ob_shape) == 1 and len(ac_shape) == 1, "flat observations/actions only"
        self.limit = int(limit)
        self.ob_shape, self.ac_shape = tuple(ob_shape), tuple(ac_shape)
        self.ob_dim, self.ac_dim = int(ob_shape[0]), int(ac_shape[0])
        self.device = torch.device(device)
        self.num_demos = 0
        self.atom_names = list(ATOMS)
        dims = dict(obs0=self.ob_dim, acs=self.ac_dim, rews=1, dones1=1, obs1=self.ob_dim)
        self.store = {k: torch.zeros(self.limit, dims[k], dtype=torch.float32, device=self.device)
                      for k in ATOMS}
        # ring state shared by the five atoms (memory.py:18-23)
        self.start = 0
        self.length = 0
        self._row_w = 2 * self.ob_dim + self.ac_dim + 2
        self._stage = torch.empty(1, self._row_w, dtype=torch.float32).pin_memory()
        self._stage_dev = torch.empty(1, self._row_w, dtype=torch.float32, device=self.device)
        self._row_host = torch.empty(1, dtype=torch.int64).pin_memory()
        self._row_dev = torch.empty(1, dtype=torch.int64, device=self.device)
        self._hb = {}

    # ---- ring arithmetic (memory.py:36-57)
    def _write_slot(self, offset):
        span = self.limit - offset
        if self.length < self.limit:
            self.length += 1
            return offset + (self.start + self.length - 1) % span
        if self.length == self.limit:
            self.start = (self.start + 1) % span
            return offset + (self.start + self.length - offset - 1) % span
        raise RuntimeError()

    @property
    def num_entries(self):
        return self.length

    @property
    def latest_entry_idx(self):
        return (self.start + self.length - 1) % self.limit

    def append(self, obs0, acs, rews, obs1, dones1, is_demo=False):
        """memory.py:177-184: one transition -> one packed H2D copy + one scatter kernel."""
        offset = 0 if is_demo else self.num_demos
        row = self._write_slot(offset)
        s = self._stage[0].numpy()
        o, a = self.ob_dim, self.ac_dim
        s[:o] = np.asarray(obs0, dtype=np.float32).ravel()
        s[o:o + a] = np.asarray(acs, dtype=np.float32).ravel()
        s[o + a] = np.asarray(rews, dtype=np.float32).reshape(-1)[0]
        s[o + a + 1:2 * o + a + 1] = np.asarray(obs1, dtype=np.float32).ravel()
        s[2 * o + a + 1] = np.asarray(dones1, dtype=np.float32).reshape(-1)[0]
        self._row_host[0] = row
        self._stage_dev.copy_(self._stage, non_blocking=True)
        self._row_dev.copy_(self._row_host, non_blocking=True)
        ops.replay_write(self.store, self._stage_dev, self._row_dev)
        # the pinned staging row is reused by the next append
        _lib.stream_sync()
        return row

    def extend_device(self, obs0, acs, rews, obs1, dones1):
        """Bulk append of n transitions already in HBM (synthetic prefill for benchmarks).  Same
        final state as n calls of append() while the buffer is not full and holds no demos."""
        n = obs0.shape[0]
        assert self.num_demos == 0 and self.length + n <= self.limit
        lo = self.length
        self.store['obs0'][lo:lo + n] = obs0
        self.store['acs'][lo:lo + n] = acs
        self.store['rews'][lo:lo + n] = rews.reshape(n, 1)
        self.store['obs1'][lo:lo + n] = obs1
        self.store['dones1'][lo:lo + n] = dones1.reshape(n, 1)
        self.length += n
        return lo

    def add_demo_transitions_to_mem(self, dset):
        """memory.py:186-198."""
        assert self.num_entries == 0 and self.num_demos == 0
        for tr in zip(dset.obs0, dset.acs, dset.env_rews, dset.obs1, dset.dones1):
            self.append(*tr, is_demo=True)
            self.num_demos += 1
        assert self.num_demos == self.num_entries

    # ---- batches
    def batchify_device(self, idxs_dev, want=ATOMS, out=None):
        return ops.replay_gather(self.store, self.start, idxs_dev, want=want, out=out)

    def _host_batch(self, B):
        """Staging of batchify for batches of B rows: pinned index words, and ONE pinned buffer the gather kernel
        stores the five atoms into directly (no device copy of the batch, no D2H call)."""
        f = self._hb.get(B)
        if f is None:
            import types
            f = types.SimpleNamespace()
            f.idx_pin = torch.zeros(B, dtype=torch.int64).pin_memory()
            f.idx_np = f.idx_pin.numpy()
            f.idx_dev = torch.zeros(B, dtype=torch.int64, device=self.device)
            dims = dict(obs0=self.ob_dim, acs=self.ac_dim, rews=1, dones1=1, obs1=self.ob_dim)
            f.pin = torch.zeros(B * sum(dims.values()), dtype=torch.float32).pin_memory()
            f.out, f.views, o = {}, {}, 0
            for k in ATOMS:
                t = f.pin[o:o + B * dims[k]].view(B, dims[k])
                f.out[k], f.views[k] = t, t.numpy()
                o += B * dims[k]
            self._hb[B] = f
        return f

    def batchify(self, idxs):
        """memory.py:74-84 — numpy in, numpy out (drop-in form): index words read from pinned staging, gather kernel
        storing into pinned host memory, one host wait.  (A host mirror of the storages gathered with numpy was
        measured slower at 1M rows: 57 us against 36 us per batch of 256 — random rows are DRAM-latency-bound on
        the host.)"""
        idxs = np.asarray(idxs)
        f = self._host_batch(len(idxs))
        f.idx_np[:] = idxs
        # (the gather reads its index words once: straight from the pinned staging, no upload in front)
        ops.replay_gather(self.store, self.start, f.idx_pin, out=f.out, host_out=True)
        _lib.stream_sync()
        out = {k: f.views[k].copy() for k in ATOMS}
        out['idxs'] = idxs
        return out

    def sample(self, batch_size):
        idxs = np.random.randint(low=0, high=self.num_entries, size=batch_size)  # memory.py:90
        return self.batchify(idxs)

    def sample_recent(self, batch_size, window):
        """memory.py:95-111."""
        width = window if window <= self.num_entries else self.num_entries
        idxs = np.ones(width) * self.latest_entry_idx
        idxs -= np.arange(start=width - 1, stop=-1, step=-1)
        idxs += self.limit
        idxs %= self.limit
        idxs = idxs.astype(int)
        assert len(idxs) == width
        idxs = np.random.choice(idxs, size=batch_size)
        return self.batchify(idxs)

    def __repr__(self):
        return "RB(limit={}, ob_shape={}, ac_shape={})".format(self.limit, self.ob_shape,
                                                               self.ac_shape)


class PrioritizedRB(RB):
    """Proportional prioritized replay (memory.py:216-390) with the two trees in HBM."""

    def __init__(self, limit, ob_shape, ac_shape, alpha, beta, demos_eps=0.1, ranked=False,
                 max_priority=1.0, device='cuda'):
        super().__init__(limit, ob_shape, ac_shape, device=device)
        assert 0. <= alpha <= 1.
        assert beta > 0, "beta must be positive"
        if ranked:
            raise NotImplementedError("ranked PER is out of scope (SURVEY.md §2)")
        self.alpha, self.beta = float(alpha), float(beta)
        self.max_priority = float(max_priority)
        self.demos_eps = float(demos_eps)
        self.ranked = False
        self.st_cap = segment_tree_capacity(limit)
        self.trees = ops.DeviceTrees(self.st_cap, device=self.device)
        self._i1 = torch.empty(1, dtype=torch.int64).pin_memory()
        self._d1 = torch.empty(1, dtype=torch.float64).pin_memory()
        self._i1_dev = torch.empty(1, dtype=torch.int64, device=self.device)
        self._d1_dev = torch.empty(1, dtype=torch.float64, device=self.device)

    def _leafs(self, idxs, prios):
        leaf, self.max_priority = ops.per_leaf_values_host(idxs, prios, self.alpha, self.num_demos,
                                                           self.demos_eps, self.max_priority)
        return leaf

    def append(self, *args, **kwargs):
        """memory.py:337-342: the new entry's leaf is max_priority ** alpha."""
        super().append(*args, **kwargs)
        idx = self.latest_entry_idx
        leaf, _ = ops.per_leaf_values_host(np.array([idx]), np.array([self.max_priority]), self.alpha,
                                           0, 0.0, self.max_priority)
        self._i1[0] = idx
        self._d1[0] = leaf[0]
        self._i1_dev.copy_(self._i1, non_blocking=True)
        self._d1_dev.copy_(self._d1, non_blocking=True)
        self.trees.set(self._i1_dev, self._d1_dev)
        torch.cuda.current_stream().synchronize()

    def extend_device(self, *args):
        lo = super().extend_device(*args)
        n = self.length - lo
        leaf = self.max_priority ** self.alpha
        self.trees.set(torch.arange(lo, lo + n, dtype=torch.int64, device=self.device),
                       torch.full((n,), leaf, dtype=torch.float64, device=self.device))
        return lo

    def draw_uniforms(self, batch_size):
        """The B draws of memory.py:278, same generator, same order."""
        if batch_size < 1024:
            return np.array([random.random() for _ in range(batch_size)], dtype=np.float64)
        # Large batches (C5: 8192 draws = 0.9 ms of Python): CPython's `random` and numpy's legacy
        # RandomState are the same MT19937 with the same 53-bit double recipe, so the state is lent to
        # numpy for one vectorised draw and handed back advanced (tests/test_replay_host.py pins this).
        ver, mt, gauss = random.getstate()
        rs = _MT_LOAN
        rs.set_state(('MT19937', np.array(mt[:-1], dtype=np.uint32), mt[-1]))
        u = rs.random_sample(batch_size)
        st = rs.get_state()
        random.setstate((ver, tuple(st[1].tolist()) + (int(st[2]),), gauss))
        return u

    def sample_device(self, batch_size, u=None):
        assert self.num_entries
        u = self.draw_uniforms(batch_size) if u is None else u
        u_dev = torch.as_tensor(u, dtype=torch.float64).to(self.device)
        iws32 = torch.empty(batch_size, dtype=torch.float32, device=self.device)
        idxs, iws = self.trees.per_sample(self.num_entries, u_dev, self.beta, iws32_out=iws32)
        batch = self.batchify_device(idxs)
        batch['idxs'], batch['iws'], batch['iws32'] = idxs, iws, iws32
        return batch

    def sample(self, batch_size):
        """memory.py:287-318 — numpy dict incl. 'iws' [B,1] float64 and 'idxs' int64."""
        d = self.sample_device(batch_size)
        out = {k: d[k].cpu().numpy() for k in ATOMS}
        out['iws'] = array_min2d(d['iws'].cpu().numpy())
        out['idxs'] = d['idxs'].cpu().numpy()
        return out

    def sample_uniform(self, batch_size):
        return super().sample(batch_size=batch_size)

    def leaves_for(self, idxs, priorities):
        """Host half of update_priorities (memory.py:344-379, non-ranked): the asserts, demos_eps,
        max_priority tracking and leaf = priority ** alpha with the host libm pow.  Returns
        (idxs int64, leaf float64) ready for the device trees."""
        idxs = np.asarray(idxs, dtype=np.int64).ravel()
        priorities = np.asarray(priorities, dtype=np.float64).ravel()
        assert len(idxs) == len(priorities), "the two arrays must be the same length"
        assert (priorities > 0).all(), "priorities must be positive"
        assert ((0 <= idxs) & (idxs < self.num_entries)).all(), \
            "no element in buffer associated w/ index"
        return idxs, self._leafs(idxs, priorities)

    def leaves_from_td(self, idxs, td, out):
        """leaves_for for the driver's refresh: priorities = |td| + 1e-6 (float32, sam_agent.py:702) ->
        asserts -> libm pow, one host call on staging arrays; `out` float64[n] receives the leaves."""
        self.max_priority = ops.per_leaves_from_td_host(td, idxs, self.num_entries, self.alpha, self.num_demos,
                                                        self.demos_eps, self.max_priority, out)
        return out

    def update_priorities(self, idxs, priorities):
        """memory.py:344-379 (non-ranked).  Host: asserts + libm pow; device: both trees."""
        idxs, leaf = self.leaves_for(idxs, priorities)
        self.trees.set(torch.as_tensor(idxs).to(self.device), torch.as_tensor(leaf).to(self.device))

    # read-back helpers (tests, logging)
    def tree_sum(self, start=0, end=None):
        return self.trees.reduce('sum', start, end)

    def tree_min(self, start=0, end=None):
        return self.trees.reduce('min', start, end)

    def tree_values(self, kind):
        return (self.trees.sum if kind == 'sum' else self.trees.min).cpu().numpy()

    def __repr__(self):
        fmt = "PrioritizedRB(limit={}, ob_shape={}, ac_shape={}, alpha={}, beta={}, "
        fmt += "demos_eps={}, ranked={}, max_priority={})"
        return fmt.format(self.limit, self.ob_shape, self.ac_shape, self.alpha, self.beta,
                          self.demos_eps, self.ranked, self.max_priority)
