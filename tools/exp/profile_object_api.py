"""Host-side profile of the object-API loop (bench.object_api_loop): cProfile + per-call wall times (GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
cfg = dict(bench.CONFIGS[name], ln=False, ret=False, nstep=0)
dev = torch.device('cuda', 0)
comm = TorchComm(dev)
mu, d, pool, drv = bench.build_learner(cfg, comm, dev, 0, False)
rs = np.random.RandomState(4321)
data = bench.HostPairs(rs.randn(32000, cfg['ob']) + 0.5, rs.uniform(-1, 1, (32000, cfg['ac'])))
bench.object_api_loop(mu, d, data, comm, 100)
torch.cuda.synchronize()
# per-call wall times
B = cfg['B']
T = {}
def tm(k, fn):
    t0 = time.perf_counter(); r = fn(); T[k] = T.get(k, 0.0) + time.perf_counter() - t0; return r
N = 300
for _ in range(N):
    xp = tm('sample', lambda: (mu.replay_buffer.sample_uniform if hasattr(mu.replay_buffer, 'sample_uniform') else mu.replay_buffer.sample)(batch_size=B))
    oe, ae = tm('expert', lambda: data.get_next_pair_batch(B))
    x = tm('concat', lambda: np.concatenate((xp['obs0'], oe), axis=0))
    tm('rms', lambda: d.obs_rms.update(x, comm))
    feeds = {d.p_obs: xp['obs0'], d.p_acs: xp['acs'], d.e_obs: oe, d.e_acs: ae}
    tm('get_grads', lambda: d.get_grads(feeds))
    tm('get_losses', lambda: d.get_losses(feeds))
    tm('optimize', lambda: d.optimize(feeds))
    tm('train', lambda: mu.train())
    tm('target', lambda: mu.update_target_net())
torch.cuda.synchronize()
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, 'us per call; total', round(sum(T.values()) / N * 1e6, 1))
pr = cProfile.Profile()
pr.enable()
bench.object_api_loop(mu, d, data, comm, 300)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
