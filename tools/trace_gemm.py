"""Stage timestamps of one GEMM (workgroup 0, wave 0); needs a -DSAMX_TRACE build.
   python tools/trace_gemm.py M N R [layout NN|NT|TN]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_tf_amd import ops, _lib  # noqa: E402

M, N, R = (int(x) for x in sys.argv[1:4])
layout = sys.argv[4] if len(sys.argv) > 4 else 'NN'
nsplit = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = 'cuda'
g = torch.Generator(device=dev); g.manual_seed(0)
if layout == 'NN':
    A = torch.randn(M, R, generator=g, device=dev); B = torch.randn(R, N, generator=g, device=dev)
    call = lambda: ops.gemm_f32(A, (R, 1), B, (N, 1), M, N, R)
    ref = lambda: A @ B
elif layout == 'NT':
    A = torch.randn(M, R, generator=g, device=dev); B = torch.randn(N, R, generator=g, device=dev)
    call = lambda: ops.gemm_f32(A, (R, 1), B, (1, R), M, N, R)
    ref = lambda: A @ B.t()
else:
    A = torch.randn(R, M, generator=g, device=dev); B = torch.randn(R, N, generator=g, device=dev)
    call = lambda: ops.gemm_f32(A, (1, M), B, (N, 1), M, N, R, nsplit=nsplit)
    ref = lambda: A.t() @ B
out = call()
err = (out - ref()).abs().max().item() / ref().abs().max().item()
for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('%s nsplit %d %dx%dx%d: %.1f us, %.1f TFLOP/s, max rel err %.2e' % (layout, nsplit, M, N, R, us, 2.0 * M * N * R / us / 1e6, err))
lib = _lib.lib()
if hasattr(lib, 'samx_debug_gemm_trace'):
    buf = (C.c_ulonglong * 64)()
    lib.samx_debug_gemm_trace(buf)
    t = np.array(buf[:], dtype=np.float64)
    t0 = t[0]
    if t[62] > t[61] > 0:
        print('shader clock over the workgroup: %.0f MHz' % ((t[62] - t[61]) / ((t[63] - t[0]) / 100.0)))
        t[61] = t[62] = 0
    print('us since entry:', ' '.join('%d:%.2f' % (i, (x - t0) / 100.0) for i, x in enumerate(t) if x > 0))
