"""Expert demonstrations: the reference's on-disk format and batching rule
(imitation_algorithms/demo_dataset.py:6-177 ; written by expert_algorithms/xpo_util.py:209-216).

Archive (.npz): per-trajectory sequences under the keys obs0, acs, env_rews, dones1, obs1 (one array
[T_i, dim] or [T_i] per trajectory) plus ep_lens and ep_env_rets.  Trajectories of different lengths
are stored by numpy as object arrays, which need `allow_pickle=True` to load — only pass that for
archives you trust (it is what the reference's numpy-1.16 `np.load` did implicitly); rectangular
archives load without it.

Host side (numpy) = the reference's semantics; `pool(device)` hands the shuffled (obs0, acs) pairs to
the device-resident pool the update step reads in place (sam_tf_amd/demo_pool.py).
"""
import numpy as np


def flatten(x):
    """demo_dataset.py:128-177: list of trajectories -> [total transitions, atom dim] (float64 like
    the reference's np.empty default), scalars per step become a column."""
    first = np.asarray(x[0])
    traj_lens = [len(t) for t in x]
    total = sum(traj_lens)
    if first.ndim == 1:
        shape = (total, 1)
    elif first.ndim == 4:  # raw pixels
        shape = (total,) + first.shape[1:]
    else:
        shape = (total, first.shape[1])
    out = np.full(shape, np.nan)
    o = 0
    for traj, n in zip(x, traj_lens):
        traj = np.asarray(traj)
        if first.ndim == 1:
            traj = traj[:, None]
        out[o:o + n] = traj
        o += n
    assert o == total and not np.any(np.isnan(out)), "array contains NaNs"
    return out


class PairDataset(object):
    """demo_dataset.py:6-41."""

    def __init__(self, obs0, acs, randomize):
        self.obs0, self.acs = obs0, acs
        self.num_entries = len(self.obs0)
        assert len(self.acs) == self.num_entries
        self.randomize = randomize
        self.init_pointer()

    def init_pointer(self):
        self.pointer = 0
        if self.randomize:
            indices = np.arange(self.num_entries)
            np.random.shuffle(indices)
            self.obs0 = self.obs0[indices, :]
            self.acs = self.acs[indices, :]

    def get_next_batch(self, batch_size):
        if batch_size < 0:
            return self.obs0, self.acs
        if self.pointer + batch_size >= self.num_entries:
            self.init_pointer()
        end = self.pointer + batch_size
        o, a = self.obs0[self.pointer:end, :], self.acs[self.pointer:end, :]
        self.pointer = end
        return o, a


class DemoDataset(object):
    """demo_dataset.py:44-116."""

    def __init__(self, expert_arxiv, size, train_fraction=None, randomize=True, full=False,
                 allow_pickle=False):
        traj = np.load(expert_arxiv, allow_pickle=allow_pickle)
        self.size = size
        assert 0 <= self.size <= len(traj['obs0']), "wrong demo dataset size"
        self.obs0 = np.array(flatten(traj['obs0'][:size]))
        self.acs = np.array(flatten(traj['acs'][:size]))
        if full:
            self.env_rews = np.array(flatten(traj['env_rews'][:size]))
            self.dones1 = np.array(flatten(traj['dones1'][:size]))
            self.obs1 = np.array(flatten(traj['obs1'][:size]))
        self.ep_rets = traj['ep_env_rets'][:size]
        self.ep_lens = traj['ep_lens'][:size]
        self.ret_mean, self.ret_std = np.mean(np.array(self.ep_rets)), np.std(np.array(self.ep_rets))
        self.len_mean, self.len_std = np.mean(np.array(self.ep_lens)), np.std(np.array(self.ep_lens))
        self.randomize = randomize
        self.pair_dset = PairDataset(self.obs0, self.acs, randomize)
        if train_fraction is not None:
            f = int(self.size * train_fraction)
            self.pair_train_set = PairDataset(self.obs0[:f, :], self.acs[:f, :], randomize)
            self.pair_val_set = PairDataset(self.obs0[f:, :], self.acs[f:, :], randomize)

    def get_next_pair_batch(self, batch_size, split=None):
        if split is None:
            return self.pair_dset.get_next_batch(batch_size)
        if split == 'train':
            return self.pair_train_set.get_next_batch(batch_size)
        if split == 'val':
            return self.pair_val_set.get_next_batch(batch_size)
        raise NotImplementedError

    def pool(self, device='cuda'):
        """The (obs0, acs) pairs in their CURRENT (already shuffled) order as a device-resident pool
        with the same pointer: its batches continue where this dataset's would."""
        from .demo_pool import DevicePairPool
        p = DevicePairPool(self.pair_dset.obs0, self.pair_dset.acs, randomize=False, device=device)
        p.randomize = self.randomize
        p.pointer = self.pair_dset.pointer
        return p
