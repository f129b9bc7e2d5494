"""ctypes loader for oracle/per_tree.c (TEST INFRASTRUCTURE ONLY, see that file's header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libper_oracle.so')


def build():
    src = os.path.join(_HERE, 'per_tree.c')
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])


def _lib():
    build()
    lib = C.CDLL(_SO)
    d, i64, p = C.c_double, C.c_int64, C.c_void_p
    lib.pt_init.argtypes = [p, i64, C.c_int]
    lib.pt_set.argtypes = [p, i64, C.c_int, i64, d]
    lib.pt_reduce.argtypes = [p, i64, C.c_int, i64, i64]
    lib.pt_reduce.restype = d
    lib.pt_find_prefixsum_idx.argtypes = [p, i64, d]
    lib.pt_find_prefixsum_idx.restype = i64
    lib.pt_per_sample.argtypes = [p, p, i64, i64, p, i64, d, p, p]
    lib.pt_per_update.argtypes = [p, p, i64, p, p, i64, d, i64, d, d]
    lib.pt_per_update.restype = d
    lib.pt_fill_range.argtypes = [p, p, i64, i64, i64, d]
    return lib


class PerTreesC(object):
    """Sum + min tree pair driven through the C oracle."""

    def __init__(self, cap):
        self.lib = _lib()
        self.cap = cap
        self.sumv = np.empty(2 * cap, dtype=np.float64)
        self.minv = np.empty(2 * cap, dtype=np.float64)
        self.lib.pt_init(self.sumv.ctypes.data, cap, 0)
        self.lib.pt_init(self.minv.ctypes.data, cap, 1)

    def set(self, idx, val):
        self.lib.pt_set(self.sumv.ctypes.data, self.cap, 0, idx, val)
        self.lib.pt_set(self.minv.ctypes.data, self.cap, 1, idx, val)

    def fill_range(self, first, n, leaf):
        self.lib.pt_fill_range(self.sumv.ctypes.data, self.minv.ctypes.data, self.cap, first, n,
                               leaf)

    def sum(self, start=0, end=None):
        return self.lib.pt_reduce(self.sumv.ctypes.data, self.cap, 0, start,
                                  self.cap if end is None else end)

    def min(self, start=0, end=None):
        return self.lib.pt_reduce(self.minv.ctypes.data, self.cap, 1, start,
                                  self.cap if end is None else end)

    def find_prefixsum_idx(self, s):
        return self.lib.pt_find_prefixsum_idx(self.sumv.ctypes.data, self.cap, s)

    def per_sample(self, num_entries, u, beta):
        u = np.ascontiguousarray(u, dtype=np.float64)
        B = len(u)
        idxs = np.empty(B, dtype=np.int64)
        iws = np.empty(B, dtype=np.float64)
        self.lib.pt_per_sample(self.sumv.ctypes.data, self.minv.ctypes.data, self.cap,
                               num_entries, u.ctypes.data, B, beta, idxs.ctypes.data,
                               iws.ctypes.data)
        return idxs, iws

    def per_update(self, idxs, prios, alpha, num_demos=0, demos_eps=0.1, max_priority=1.0):
        idxs = np.ascontiguousarray(idxs, dtype=np.int64)
        prios = np.ascontiguousarray(prios, dtype=np.float64)
        return self.lib.pt_per_update(self.sumv.ctypes.data, self.minv.ctypes.data, self.cap,
                                      idxs.ctypes.data, prios.ctypes.data, len(idxs), alpha,
                                      num_demos, demos_eps, max_priority)
