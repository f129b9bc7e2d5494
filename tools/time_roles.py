"""Device time of samx_step_grads with subsets of the fused roles (profiling aid).
    SAMX_FUSED_ROLES=<mask> python tools/time_roles.py [config] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sam_tf_amd import ops
from sam_tf_amd.comm import TorchComm

cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'C2'], mem=65536)
if len(sys.argv) > 2:
    cfg['B'] = int(sys.argv[2])
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, False)
drv.step()
st = torch.cuda.current_stream()
for name, fn in (('step_grads', lambda: ops.step_grads(drv.L, drv.batch, drv.out)),
                 ('step_apply', lambda: ops.step_apply(drv.L, drv.B, 3, 1.0))):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(200):
        fn()
    e1.record(st); e1.synchronize()
    print('roles=%s B=%d %s: %.2f us' % (os.environ.get('SAMX_FUSED_ROLES', '7'), cfg['B'], name,
                                         e0.elapsed_time(e1) / 200 * 1e3))
