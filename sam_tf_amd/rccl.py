"""In-stream RCCL collectives for the update step at N > 1.

torch.distributed's all_reduce runs on the process group's own stream (two cross-stream event
hand-offs per call) and costs the host ~10-25 us per call; the reference's three Allreduce calls per
step (mpi_adam.py:27 twice, mpi_running_mean_std.py:55) sit between kernels that take ~5-20 us each.
This communicator calls librccl (the one torch ships and has already loaded) directly on the CALLER's
stream: no side stream, no event, and the call is capturable into the same hipGraph as the kernels
around it, so a K-step launch carries its collectives inside the graph.

The unique id is created on rank 0 and handed to the other ranks through the existing
torch.distributed group (plumbing); one communicator per process, one process per GPU.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

_DTYPES = {torch.int64: 4, torch.float32: 7, torch.float64: 8}
_SUM = 0


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_byte * 128)]


_lib = None


def _rccl():
    global _lib
    if _lib is None:
        lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclCommAbort.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


class RcclError(RuntimeError):
    pass


def _check(rc, what, lib=None):
    if rc != 0:
        raise RcclError('%s failed: %s' % (what, (lib or _rccl()).ncclGetErrorString(rc).decode()))


class RcclComm(object):
    """One RCCL communicator over the ranks of the default torch.distributed group.

    Construction is COLLECTIVE and staged: after each stage every rank contributes "ok so far" to an
    all-reduce(MIN) over the torch.distributed group (`agree`), and all ranks go on or all give up
    (ncclCommAbort where a communicator already exists).  Stages: (1) library loaded, unique id obtained on
    rank 0; (2) id broadcast + ncclCommInitRank; (3) probe all-reduce with a known answer.
    What the agreement covers: a rank whose stage FAILED AFTER the stage's own collective completed (or that
    failed in purely local work, stage 1) — its peers learn of it in `agree` and nobody goes on alone.
    What it cannot cover: a rank that fails INSIDE a stage before reaching that stage's rendezvous
    (ncclCommInitRank returning an error before the bootstrap, ncclAllReduce refusing to enqueue) — its peers are
    then inside the rendezvous / waiting on the stream while it waits in `agree`, two different collectives.
    That hang is bounded only by the collectives' own timeouts (NCCL's bootstrap timeout, the process group's
    timeout) or by the caller's watchdog (bench.py prints a `degraded` line and exits 3);
    tests/test_comm_gloo.py::test_direct_rccl_failure_before_the_rendezvous_is_bounded_by_the_timeout_only shows it.
    N > 1 parity of this path is UNPINNED until it has run on two real GPUs (no multi-GPU box was
    available to tests/; tests/test_comm_gloo.py covers the agreement logic with a stand-in library)."""

    def __init__(self, device, agree=None, lib=None):
        assert dist.is_initialized(), 'RcclComm needs torch.distributed for the id exchange'
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device(device)
        self.comm = C.c_void_p()
        self._lib = None
        agree = agree or self._agree
        # ---- stage 1: local set-up only (nothing here can block on a peer)
        err, uid = None, _UniqueId()
        try:
            self._lib = (lib() if callable(lib) else lib) if lib is not None else _rccl()
            if self.rank == 0:
                _check(self._lib.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId', self._lib)
        except Exception as e:  # noqa: BLE001
            err = e
        if not agree(err is None):
            raise RcclError('direct RCCL set-up, stage 1 (library / unique id): %s' %
                            (err if err is not None else 'another rank failed'))
        # ---- stage 2: every rank takes part in the broadcast and in the communicator's rendezvous
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8)
        if dist.get_backend() == 'nccl':
            t = t.to(self.device)
        dist.broadcast(t, src=0)
        C.memmove(C.byref(uid), t.cpu().numpy().tobytes(), 128)
        try:
            with torch.cuda.device(self.device) if self.device.type == 'cuda' else _nullctx():
                _check(self._lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank),
                       'ncclCommInitRank', self._lib)
        except Exception as e:  # noqa: BLE001
            err = e
        if not agree(err is None):
            self.abort()
            raise RcclError('direct RCCL set-up, stage 2 (ncclCommInitRank): %s' %
                            (err if err is not None else 'another rank failed'))
        # ---- stage 3: first call outside any capture: lazy channel / proxy set-up happens here, and the
        # result proves the communicator before the hot path depends on it
        try:
            probe = torch.full((8,), float(self.rank + 1), dtype=torch.float32, device=self.device)
            self.all_reduce_(probe)
            if self.device.type == 'cuda':
                torch.cuda.current_stream(self.device).synchronize()
            want = self.world * (self.world + 1) / 2.0
            if not bool((probe == want).all()):
                raise RcclError('probe all-reduce returned %r, expected %r' % (probe.tolist(), want))
        except Exception as e:  # noqa: BLE001
            err = e
        if not agree(err is None):
            self.abort()
            raise RcclError('direct RCCL set-up, stage 3 (probe all-reduce): %s' %
                            (err if err is not None else 'another rank failed'))

    def _agree(self, ok):
        """all-reduce(MIN) of this rank's verdict over the torch.distributed group."""
        v = torch.tensor([1 if ok else 0], dtype=torch.int32,
                         device=self.device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(int(v.item()))

    def all_reduce_(self, t):
        """In-place SUM over ranks on torch's CURRENT stream (capturable)."""
        assert t.is_contiguous() and t.dtype in _DTYPES
        stream = torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else None
        _check(self._lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _DTYPES[t.dtype], _SUM, self.comm,
                                       stream), 'ncclAllReduce', self._lib)
        return t

    def abort(self):
        """Tear the communicator down without waiting for peers (after a disagreement)."""
        if self.comm and self._lib is not None:
            self._lib.ncclCommAbort(self.comm)
        self.comm = C.c_void_p()

    def destroy(self):
        if self.comm and self._lib is not None:
            self._lib.ncclCommDestroy(self.comm)
        self.comm = C.c_void_p()


class _nullctx(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
