"""The reference's loop shape with a FULL replay ring (sam.py:305-433): per iteration a rollout appends transitions
(each moves the ring's start), then 25 training steps (d_steps 1, g_steps 3 -> 75 rows).  Wall time per iteration with
the captured plans kept (index words carry storage rows) and with a re-capture per iteration (what a moved start cost)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

base = 'C3' if 'C3' in sys.argv[1:] else 'C2'   # C3: prioritized replay (sampling kernels on device geometry words)
cfg = dict(bench.CONFIGS[base], mem=65536, d_steps=1, g_steps=3)
if 'launcher' in sys.argv[1:]:   # LayerNorm, return normalisation + PopArt, n-step returns n = 96
    cfg.update(ln=True, ret=True, nstep=96)
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
assert mu.replay_buffer.num_entries == mu.replay_buffer.limit
drv.prepare_all()
rs = np.random.RandomState(1)


def iteration(recapture):
    for t in range(4):
        mu.store_transition(rs.randn(mu.ob_dim).astype(np.float32), rs.uniform(-1, 1, mu.ac_dim).astype(np.float32),
                            float(rs.rand()), rs.randn(mu.ob_dim).astype(np.float32), 0.0)
    if recapture:
        drv.refresh()
    drv.start_iteration()
    drv.run(75)
    torch.cuda.synchronize()


for mode in (False, True, False):
    for _ in range(3):
        iteration(mode)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        iteration(mode)
    print(base + ' %s: %.2f ms per iteration (4 appends + 75 rows)' % ('re-capture per iteration' if mode else 'plans kept', (time.perf_counter() - t0) / n * 1e3))
