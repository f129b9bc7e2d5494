"""MpiAdamOptimizer mirror (helpers/mpi_adam.py:8-72) over flat device vectors.

update(): RCCL all-reduce(SUM) of the flat gradient (line 27; the 1/size is folded into the kernel),
global-norm clip of the MEAN gradient (lines 31-33), TF1 Adam, optional fused Polyak.
"""
import torch

from . import ops


class MpiAdam(object):

    def __init__(self, comm, param, clip_norm, learning_rate, name='adam', m=None, v=None, t_dev=None):
        self.comm, self.param, self.name = comm, param, name
        self.clip_norm, self.lr = clip_norm, float(learning_rate)
        # m / v / t_dev may be views into a learner-wide flat state (ops.LearnerState)
        self.m = torch.zeros_like(param) if m is None else m
        self.v = torch.zeros_like(param) if v is None else v
        # the update count lives in HBM: the kernels of a captured graph read it, not a host argument
        self.t_dev = torch.zeros(1, dtype=torch.int64, device=param.device) if t_dev is None else t_dev
        self.partials = ops.adam_partials(param.device)
        self._mism = torch.zeros(1, dtype=torch.int32, device=param.device)

    @property
    def t(self):
        return int(self.t_dev.item())

    def update(self, grad, targ=None, tau=0.0, reduce=True):
        if reduce:
            self.comm.allreduce_(grad)
        self.enqueue(grad, targ, tau)

    def enqueue(self, grad, targ=None, tau=0.0):
        """Device work only (capturable): the update counter lives in HBM."""
        ops.clip_adam(self.param, grad, self.m, self.v, self.lr, 0, self.clip_norm, self.partials,
                      grad_scale=1.0 / self.comm.Get_size(), targ=targ, tau=tau, step_dev=self.t_dev)

    def sync_from_root(self, var_list=None):
        """lines 38-53: broadcast rank 0's parameters (one flat Bcast instead of one per variable)."""
        self.comm.Barrier()
        self.comm.bcast_(self.param, root=0)

    def check_synced(self, var_list=None):
        """lines 55-72: every rank must hold rank 0's parameters bit for bit."""
        self.comm.Barrier()
        root = self.param.clone()
        self.comm.bcast_(root, root=0)
        n = int(ops.count_mismatch(self.param, root, self._mism).item())
        assert n == 0, "mismatch: {} of {} parameters differ from rank 0 ({})".format(
            n, self.param.numel(), self.name)
