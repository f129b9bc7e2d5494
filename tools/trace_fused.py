"""Phase timestamps of the fused step kernel (needs a libsamx.so built with -DSAMX_TRACE:
   SAMX_EXTRA_FLAGS=-DSAMX_TRACE python -m sam_tf_amd.build --force).  Prints, per role, microseconds
   since the workgroup's first timestamp (wall_clock64, 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sam_tf_amd import ops, _lib  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'C2'])
mu, d, pool, drv = bench.build_learner(cfg, TorchComm(dev), dev, 0, True)
drv.prepare(1)
drv.run(64)
lib = _lib.lib()
buf = (C.c_ulonglong * 64)()
acc = np.zeros((20, 64))
for it in range(20):
    ops.step_grads_part(drv.L, drv.batch, drv.out, 1)
    torch.cuda.synchronize()
    assert lib.samx_debug_trace(buf) == 0
    acc[it] = np.array(buf[:], dtype=np.float64)
for role, name in ((0, 'actor'), (1, 'target+critic'), (2, 'disc'), (3, 'disc interpolate tile (or 2nd TC pass)')):
    t = acc[5:, role * 16:(role + 1) * 16]
    if t[:, 0].min() == 0:
        continue
    rel = (t - t[:, :1]) / 100.0  # us
    print(name, ' '.join('%d:%.2f' % (i, x) for i, x in enumerate(np.median(rel, axis=0)) if t[0, i] > 0))
