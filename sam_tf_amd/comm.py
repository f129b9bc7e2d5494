"""The `comm` object the reference threads through SAMAgent / Discriminator / MpiAdam is an mpi4py
communicator (run_sam.py:20-21).  This is the same surface — Get_rank / Get_size / Barrier /
Allreduce / Bcast — over torch.distributed: backend "nccl" (= RCCL over xGMI) for device tensors,
"gloo" on CPU for tests.  One process per GPU; with a single process every call is local.

Replaces helpers/mpi_moments.py:6-22 (mpi_mean_like) and the Allreduce/Bcast calls of
helpers/mpi_adam.py:27,49-52,65-70 and helpers/mpi_running_mean_std.py:55.
"""
import numpy as np
import torch
import torch.distributed as dist

SUM = 'sum'


class TorchComm(object):

    def __init__(self, device=None):
        self.active = dist.is_available() and dist.is_initialized()
        self.device = device
        self.direct = None   # sam_tf_amd.rccl.RcclComm once enable_direct() succeeded
        self.direct_error = None

    def enable_direct(self):
        """Device all-reduces go through librccl on the caller's stream (capturable into hipGraphs)
        instead of the process group's side stream.  Returns True on success on ALL ranks; otherwise every
        rank keeps using torch.distributed.  The all-ranks-or-none agreement is reached stage by stage
        inside RcclComm's constructor (library / id, communicator rendezvous, probe): a rank whose stage
        failed after that stage's collective completed takes its peers with it.  A rank that fails BEFORE
        it reaches a stage's rendezvous leaves its peers waiting there — bounded by the collectives' timeouts
        or the caller's watchdog only (see RcclComm); `direct_error` keeps the reason."""
        self.direct_error = None
        if not (self.active and self.Get_size() > 1 and dist.get_backend() == 'nccl'):
            self.direct_error = 'needs an initialised nccl process group with more than one rank'
            return False
        try:
            from .rccl import RcclComm
            self.direct = RcclComm(self.device if self.device is not None else torch.device('cuda'))
        except Exception as e:  # noqa: BLE001 - agreed by all ranks inside RcclComm: stay on torch.distributed
            print('sam_tf_amd.comm: direct RCCL unavailable (%s); using torch.distributed' % (e,), flush=True)
            self.direct_error = str(e)
            self.direct = None
        return self.direct is not None

    @property
    def capturable(self):
        return self.direct is not None

    def Get_rank(self):
        return dist.get_rank() if self.active else 0

    def Get_size(self):
        return dist.get_world_size() if self.active else 1

    def Barrier(self):
        if self.active:
            dist.barrier()

    # ---- device / tensor forms used on the hot path
    def allreduce_(self, t):
        """In-place SUM over ranks of a torch tensor (CUDA -> RCCL, CPU -> gloo)."""
        if self.active and self.Get_size() > 1:
            if self.direct is not None and t.is_cuda:
                return self.direct.all_reduce_(t)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def bcast_(self, t, root=0):
        if self.active and self.Get_size() > 1:
            dist.broadcast(t, src=root)
        return t

    # ---- mpi4py-shaped numpy forms (drop-in for host callers)
    def _staging(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.active and dist.get_backend() == 'nccl':
            t = t.to(self.device if self.device is not None else 'cuda')
        return t

    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        assert op == SUM, "only MPI.SUM is used by the reference on this path"
        t = self._staging(sendbuf).clone()
        self.allreduce_(t)
        recvbuf[...] = t.cpu().numpy().reshape(recvbuf.shape)

    def Bcast(self, buf, root=0):
        t = self._staging(buf).clone()
        self.bcast_(t, root)
        buf[...] = t.cpu().numpy().reshape(buf.shape)


def mpi_mean_like(x, comm):
    """helpers/mpi_moments.py:6-22: Allreduce(SUM) / size, any array shape."""
    x = np.asarray(x)
    out = np.zeros_like(x)
    comm.Allreduce(x, out, op=SUM)
    return out / comm.Get_size()
