"""Where a PER step's wall time goes on the host (C3): phases of UpdateStepDriver._launch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sam_tf_amd.comm import TorchComm
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'C3'])
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
drv.prepare(1)
drv.run(200)
T = {}
def tick(name, t0):
    t = time.perf_counter(); T[name] = T.get(name, 0.0) + (t - t0); return t
N = 1000
for _ in range(N):
    t = time.perf_counter()
    slot = drv.slots[drv.launch_no % len(drv.slots)]
    drv._draw(slot, 1, 0); t = tick('draw(+pending)', t)
    drv._upload(slot, 1); t = tick('upload H2D', t)
    for run in drv.plans[(1, 0)]:
        run()
    t = tick('graph launch', t)
    drv._td_host.copy_(drv.td_post, non_blocking=True)
    drv._idx_host.copy_(drv.idx_a, non_blocking=True); t = tick('D2H enqueue', t)
    nxt = drv.slots[(drv.launch_no + 1) % len(drv.slots)]
    if nxt.used:
        nxt.event.synchronize()
    drv._draw_rng(nxt, 1, 0); nxt.predrawn = True; t = tick('predraw next', t)
    torch.cuda.current_stream().synchronize(); t = tick('wait GPU', t)
    pr = np.abs(drv._td_host.numpy()) + np.float32(1e-6)
    idxs, leaf = mu.replay_buffer.leaves_for(drv._idx_host.numpy(), pr)
    drv._pending = (idxs.copy(), leaf); t = tick('host pow', t)
    drv.step_no += 1; drv.launch_no += 1
tot = sum(T.values())
for k, v in T.items():
    print('%-16s %6.1f us' % (k, v / N * 1e6))
print('%-16s %6.1f us' % ('total', tot / N * 1e6))
