"""Phase timestamps of the 256-wide role-fused kernel (needs a libsamx.so built with -DSAMX_TRACE:
   SAMX_EXTRA_FLAGS=-DSAMX_TRACE python -m sam_tf_amd.build --force).  Prints microseconds since the
   target+critic workgroup 0's first timestamp (wall_clock64, 100 MHz).  SAMX_WIDE_ALONE=1: only the agent
   roles run (no discriminator beside them)."""
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
cfg = dict(bench.CONFIGS['C4'])
if os.environ.get('SAMX_TRACE_LN') == '1':   # with_layernorm nets (D interpolate phases: LTR stamps)
    cfg['ln'] = True
mu, d, pool, drv = bench.build_learner(cfg, TorchComm(dev), dev, 0, True)
drv.prepare(1)
drv.run(8)
lib = C.CDLL(_lib.SO_PATH)
buf = (C.c_ulonglong * 256)()
acc = np.zeros((20, 256))
part = 1 | (4 if os.environ.get('SAMX_WIDE_ALONE') == '1' else 0)
for it in range(20):
    ops.step_grads_part(drv.L, drv.batch, drv.out, part)
    torch.cuda.synchronize()
    assert lib.samx_debug_wide_trace(buf) == 0
    acc[it] = np.array(buf[:], dtype=np.float64)
t = acc[5:, :32]
rel = (t - t[:, :1]) / 100.0
print('target+critic', ' '.join('%d:%.2f' % (i, x) for i, x in enumerate(np.median(rel, axis=0)) if t[0, i] > 0))
base = t[:, :1]
ra, rd = acc[5:, 32:34], acc[5:, 64:70]
f = lambda x: ' '.join('%.2f' % v for v in np.median((x - base) / 100.0, axis=0))
print('actor role workgroup 0 (start end):', f(ra))
print('D role first tiles (interpolate start end | policy start end | expert start end):', f(rd))
print('D interpolate tile 0 phases (fwd1 start, fwd2 start, fwd2 end, head bwd done, gz1 done, g0 done, g done, '
      'sums done, V done, tangent 1 done):', f(acc[5:, 70:80]))
if os.environ.get('SAMX_TRACE_LN') == '1':
    print('D interpolate tile 0 with LayerNorm (fwd1 start, fwd2 start, fwd2 end, T_z2 done, T_n1 product done, T_z1 done, '
          'V done, zdot1 done, tan1 done, zdot2 done, tan2 done, P_n1 done, end):', f(acc[5:, 70:83]))
print('D interpolate tile 0 prologue (rows issued, weights drawn, statistics staged, barrier passed, half 0, half 1):', f(acc[5:, 80:86]))
st = acc[5:, 128:]
n = int((st[0] > 0).sum())
d = np.median(np.diff(st[:, :n], axis=1), axis=0) / 100.0
print('stage starts (us between consecutive weight stages of target+critic workgroup 0; product boundaries show as '
      'the larger gaps):')
print(' '.join('%d:%.2f' % (i, x) for i, x in enumerate(d)))
cyc = np.median(acc[5:, 121] - acc[5:, 120])
us = np.median(t[:, 13] - t[:, 0]) / 100.0
print('shader clock over the target+critic role: %.0f cycles in %.2f us = %.2f GHz' % (cyc, us, cyc / us / 1e3))
