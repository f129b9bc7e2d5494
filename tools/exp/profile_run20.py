"""cProfile of the host side of run(20) from idle (the driver's bench invocation), C2."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

n = 20
cfg = dict(bench.CONFIGS['C2'])
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
drv.prepare_all()
drv.prepare(n)
drv.run(5)
for _ in range(5):
    torch.cuda.synchronize(); drv.run(n)
pr = cProfile.Profile()
R = 50
for _ in range(R):
    torch.cuda.synchronize()
    pr.enable()
    drv.run(n)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative')
st.print_stats(28)
