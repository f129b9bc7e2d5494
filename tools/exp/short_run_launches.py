"""Where a short run's time goes: per launch its size, the host time of _launch and the wall clock when it returns
(C2 + LayerNorm, run(400) after run(50)).   python tools/exp/short_run_launches.py [steps] [warmup]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
w = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = dict(bench.CONFIGS['C2'], ln=True)
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
drv.prepare_all()
log = []
f = drv._launch


def g(K, more=False):
    t = time.perf_counter()
    parts.clear()
    f(K, more)
    log.append((K, time.perf_counter() - t, time.perf_counter(), dict(parts)))


drv._launch = g
parts = {}
def timed(obj, attr, key):
    fn = getattr(obj, attr)
    def h(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        parts[key] = parts.get(key, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, attr, h)
timed(drv, '_draw', 'draw'); timed(drv, '_upload', 'upload')
for key in list(drv.plans):
    fs = drv.plans[key]
    def mk(f0):
        def rep():
            t = time.perf_counter(); f0(); parts['replay'] = parts.get('replay', 0.0) + time.perf_counter() - t
        return rep
    drv.plans[key] = [mk(f0) for f0 in fs]
import gc
mode = os.environ.get('GCMODE', '')
if mode == 'freeze':
    gc.collect(); gc.freeze()
elif mode == 'off':
    gc.disable()
print('gc', mode, gc.get_count(), gc.get_threshold())
drv.run(w)
torch.cuda.synchronize()
del log[:]
t0 = time.perf_counter()
drv.run(n)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('run(%d): host %.2f ms, total %.2f ms -> %.1f us/step; plans %s' % (n, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / n * 1e6, sorted(drv.plans)))
print('gc counts after', gc.get_count(), gc.get_stats()[2])
for K, dt, at, pr in log:
    if dt < 1e-3: continue
    print('  K %2d  host %7.1f us  returned at %8.1f us  %s' % (K, dt * 1e6, (at - t0) * 1e6, {k: round(v * 1e6) for k, v in pr.items()}))
