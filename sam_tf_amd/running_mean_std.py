"""Device-resident MpiRunningMeanStd (helpers/mpi_running_mean_std.py:9-59).

State in HBM: float64 [sum(dim) | sumsq(dim) | count] (init 0, eps, eps) and the derived float32
mean / std the network kernels read.  update() = column-sum kernel, RCCL all-reduce of the
2*dim+1 doubles (line 55), commit kernel.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import ops


class DeviceRunningMeanStd(object):

    def __init__(self, epsilon=1e-2, shape=(), device='cuda'):
        self.shape = tuple(shape)
        self.dim = int(np.prod(self.shape)) if self.shape else 1
        d = self.dim
        init = np.zeros(2 * d + 1, dtype=np.float64)
        init[d:] = epsilon
        self.sums = torch.as_tensor(init).to(device)
        self.addvec = torch.zeros(2 * d + 1, dtype=torch.float64, device=device)
        self.mean = torch.empty(d, dtype=torch.float32, device=device)
        self.std = torch.empty(d, dtype=torch.float32, device=device)
        self.device = torch.device(device)
        self._stage = {}
        ops.rms_commit(self.sums, None, self.mean, self.std)

    def update_device(self, x1, x2, comm):
        """x1 (and optionally x2) float32 [n, dim] in HBM; rows of x2 follow those of x1."""
        ops.rms_batch_sums(x1, x2, self.addvec)
        if comm is not None:
            comm.allreduce_(self.addvec)
        ops.rms_commit(self.sums, self.addvec, self.mean, self.std)

    def update(self, x, comm):
        """Drop-in form: numpy [n, *shape] (mpi_running_mean_std.py:47-59).  The rows travel through pinned
        staging (per row count) and an asynchronous copy: the call enqueues and returns."""
        x = np.asarray(x)
        n = x.size // self.dim
        st = self._stage.get(n)
        if st is None:
            pin = torch.empty(n, self.dim, dtype=torch.float32).pin_memory()
            st = self._stage[n] = [pin, pin.numpy(), torch.empty(n, self.dim, dtype=torch.float32, device=self.device),
                                   torch.cuda.Event(), False]
        pin, view, dev, event, used = st
        if used:
            event.synchronize()   # the previous upload has read the staging rows
        np.copyto(view, x.reshape(n, self.dim), casting='same_kind')
        if pin.numel() * 4 <= (256 << 10):
            # small batches: the column-sum kernel reads the rows once, straight from the pinned staging
            one = comm is None or comm.Get_size() == 1
            if one and n % 2 == 0 and n * self.dim <= 16384:
                # one process, an even row count (the discriminator's policy + expert rows): sums AND commit in one
                # launch (samx_d_input_stats on dense rows: first half | second half)
                h = n // 2
                _lib.check(_lib.lib().samx_d_input_stats(
                    C.c_void_p(pin.data_ptr()), 0, h, None, C.c_void_p(pin.data_ptr() + 4 * h * self.dim), None, h,
                    self.dim, _lib.ptr(self.addvec), 1, _lib.ptr(self.sums), _lib.ptr(self.mean), _lib.ptr(self.std),
                    None, 0, _lib.stream_ptr()))
                event.record()
                st[4] = True
                return
            ops.rms_batch_sums(pin, None, self.addvec, host_ok=True)
            event.record()
            st[4] = True
            if comm is not None:
                comm.allreduce_(self.addvec)
            ops.rms_commit(self.sums, self.addvec, self.mean, self.std)
            return
        dev.copy_(pin, non_blocking=True)
        event.record()
        st[4] = True
        self.update_device(dev, None, comm)
