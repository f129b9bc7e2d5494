"""Per-workgroup timestamps and placement of the register-direct weight-gradient kernel at C4 (needs a libsamx.so
   built with -DSAMX_TRACE: SAMX_EXTRA_FLAGS=-DSAMX_TRACE python -m sam_tf_amd.build --force).
   SAMX_TRACE_LN=1: the LayerNorm networks."""
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
if os.environ.get('SAMX_TRACE_LN') == '1':
    cfg['ln'] = True
mu, d, pool, drv = bench.build_learner(cfg, TorchComm(dev), dev, 0, True)
drv.prepare(1)
drv.run(8)
lib = C.CDLL(_lib.SO_PATH)
buf = (C.c_ulonglong * (10 * 1024))()
for it in range(6):
    ops.step_grads_part(drv.L, drv.batch, drv.out, 1)
    torch.cuda.synchronize()
assert lib.samx_debug_wd_trace(buf) == 0
t = np.array(buf[:], dtype=np.uint64).reshape(1024, 10)
live = t[:, 0] > 0
tt = t[live].astype(np.int64)
t0 = tt[:, 0].min()
hw, xcc = tt[:, 4] & 0xffffffff, (tt[:, 4] >> 32) & 0xf
cu, se, sh = (hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 12) & 1
where = xcc * 1000 + se * 100 + sh * 10 + cu   # one number per CU
blocks = np.nonzero(live)[0]
print('%d product workgroups; entry %.2f..%.2f us, end %.2f..%.2f us after the first entry' % (
    live.sum(), 0.0, (tt[:, 0].max() - t0) / 100.0, (tt[:, 3].min() - t0) / 100.0, (tt[:, 3].max() - t0) / 100.0))
dur = (tt[:, 3] - tt[:, 0]) / 100.0
print('workgroup time: median %.2f, min %.2f, max %.2f us; first batch pair done after %.2f us (median), loop %.2f us, '
      'reduction + stores %.2f us' % (np.median(dur), dur.min(), dur.max(), np.median(tt[:, 1] - tt[:, 0]) / 100.0,
                                      np.median(tt[:, 2] - tt[:, 0]) / 100.0, np.median(tt[:, 3] - tt[:, 2]) / 100.0))
full = dur > 0.8 * np.median(dur)
print('tail of wave 0, us after its loop: barrier passed %.2f, stores issued %.2f' % tuple(
    np.median((tt[full, k] - tt[full, 2]) / 100.0) for k in (5, 3)))
print('shader clock over wave 0\'s loop: %.0f MHz (median)' % np.median(
    (tt[full, 9] - tt[full, 8]) / ((tt[full, 2] - tt[full, 0]) / 100.0)))
u, c = np.unique(where, return_counts=True)
print('%d distinct CUs hold them; workgroups per CU: %s' % (len(u), dict(zip(*np.unique(c, return_counts=True)))))
print('XCD of block b (b %% 8 -> XCC_ID):', {int(b % 8): int(x) for b, x in zip(blocks[:64], xcc[:64])})
for b, row, w in list(zip(blocks, tt, where))[:: max(1, len(blocks) // 24)]:
    print('block %3d cu %4d: %s' % (b, w, ' '.join('%.2f' % ((x - t0) / 100.0) for x in row[:4])))
# which workgroups share a CU: slot s and slot s + k of one XCD?
by_cu = {}
for b, w in zip(blocks, where):
    by_cu.setdefault(int(w), []).append(int(b) >> 3)
d = [max(v) - min(v) for v in by_cu.values() if len(v) == 2]
if d:
    u, c = np.unique(d, return_counts=True)
    print('slot distance of the two workgroups of a CU (XCD-local slots): %s' % dict(zip(u.tolist(), c.tolist())))
