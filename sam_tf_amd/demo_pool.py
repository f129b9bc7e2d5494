"""Expert (state, action) pairs resident in HBM with the batching rule of
imitation_algorithms/demo_dataset.py:6-41 (PairDataset): shuffle, hand out consecutive slices,
reshuffle when fewer than batch_size pairs remain.  Batches are views — no copy, no gather.
Loading the reference's .npz archives (demo_dataset.py:44-116) is a "next" row.
"""
import numpy as np
import torch


class DevicePairPool(object):

    def __init__(self, obs0, acs, randomize=True, device='cuda'):
        self.device = torch.device(device)
        self.obs0 = torch.as_tensor(np.asarray(obs0, dtype=np.float32)).to(self.device).contiguous()
        self.acs = torch.as_tensor(np.asarray(acs, dtype=np.float32)).to(self.device).contiguous()
        self.num_entries = self.obs0.shape[0]
        assert self.acs.shape[0] == self.num_entries
        self.randomize = randomize
        # the permutation travels through pinned staging (two buffers in turn, an event each) and is applied by
        # stream-ordered device work: a reshuffle never makes the host wait for the launches already queued
        self._stage = self._events = None
        if self.device.type == 'cuda':
            self._stage = [torch.empty(self.num_entries, dtype=torch.int64).pin_memory() for _ in range(2)]
            self._events = [torch.cuda.Event() for _ in range(2)]
            self._idx_dev = torch.empty(self.num_entries, dtype=torch.int64, device=self.device)
            self._tmp_obs, self._tmp_acs = torch.empty_like(self.obs0), torch.empty_like(self.acs)
        self._turn = 0
        self.init_pointer()

    def init_pointer(self):
        self.apply_permutation(self.shuffle_host())

    def shuffle_host(self):
        """The host half of demo_dataset.py:33-40's reshuffle: pointer back to 0 and the np.random.shuffle draw at
        its place in the generator's stream.  Returns the permutation (None without `randomize`) for
        apply_permutation — a caller that draws batches AHEAD of launching the rows that read them (the update-step
        driver) applies it only when the row that owns the wrapping batch comes up, so rows drawn before the wrap
        still read the old order."""
        self.pointer = 0
        if not self.randomize:
            return None
        indices = np.arange(self.num_entries)
        np.random.shuffle(indices)  # same generator as the reference
        return indices

    def apply_permutation(self, indices):
        """The device half: permute the resident pairs in place, stream-ordered."""
        if indices is not None:
            # permute in place: the storage address stays fixed (captured HIP graphs read it)
            if self._stage is None:
                idx = torch.as_tensor(indices).to(self.device)
                self.obs0.copy_(self.obs0.index_select(0, idx))
                self.acs.copy_(self.acs.index_select(0, idx))
                return
            t = self._turn % 2
            if self._turn >= 2:
                self._events[t].synchronize()   # the copy that read this staging buffer two reshuffles ago
            self._stage[t].numpy()[:] = indices
            # (by a kernel on the current stream, like the driver's index words: a hipMemcpyAsync between two graph
            # launches hands over to the copy engine and back)
            from . import ops
            ops.copy_words(self._stage[t], self._idx_dev, self.num_entries)
            self._events[t].record()
            self._turn += 1
            torch.index_select(self.obs0, 0, self._idx_dev, out=self._tmp_obs)
            torch.index_select(self.acs, 0, self._idx_dev, out=self._tmp_acs)
            self.obs0.copy_(self._tmp_obs)
            self.acs.copy_(self._tmp_acs)

    def get_next_pair_batch_device(self, batch_size):
        if self.pointer + batch_size >= self.num_entries:
            self.init_pointer()
        end = self.pointer + batch_size
        o, a = self.obs0[self.pointer:end], self.acs[self.pointer:end]
        self.pointer = end
        return o, a

    def get_next_pair_batch(self, batch_size):
        o, a = self.get_next_pair_batch_device(batch_size)
        return o.cpu().numpy(), a.cpu().numpy()
