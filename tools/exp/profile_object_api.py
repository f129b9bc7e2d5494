"""Host-side profile of the object-API loop (bench.object_api_loop) at one configuration: where a step's
wall-clock time goes.  python tools/exp/profile_object_api.py [C4] [steps]"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS[name], ln=False, ret=False, nstep=0)
comm = TorchComm(dev)
mu, d, pool, drv = bench.build_learner(cfg, comm, dev, 0, False)
rs = np.random.RandomState(4321)
data = bench.HostPairs(rs.randn(32000, cfg['ob']) + 0.5, rs.uniform(-1, 1, (32000, cfg['ac'])))
bench.object_api_loop(mu, d, data, comm, 30)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
bench.object_api_loop(mu, d, data, comm, n)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
