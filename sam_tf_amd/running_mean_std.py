"""Device-resident MpiRunningMeanStd (helpers/mpi_running_mean_std.py:9-59).

State in HBM: float64 [sum(dim) | sumsq(dim) | count] (init 0, eps, eps) and the derived float32
mean / std the network kernels read.  update() = column-sum kernel, RCCL all-reduce of the
2*dim+1 doubles (line 55), commit kernel.
"""
import numpy as np
import torch

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
        ops.rms_commit(self.sums, None, self.mean, self.std)

    def update_device(self, x1, x2, comm):
        """x1 (and optionally x2) float32 [n, dim] in HBM; rows of x2 follow those of x1."""
        ops.rms_batch_sums(x1, x2, self.addvec)
        if comm is not None:
            comm.allreduce_(self.addvec)
        ops.rms_commit(self.sums, self.addvec, self.mean, self.std)

    def update(self, x, comm):
        """Drop-in form: numpy [n, *shape] (mpi_running_mean_std.py:47-59)."""
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1, self.dim))
        self.update_device(torch.as_tensor(x).to(self.device), None, comm)
