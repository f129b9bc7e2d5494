"""Does the two-branch PER row capture?  python tools/exp/two_branch_capture.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402

cfg = dict(ob=5, ac=2, h=(33, 70), B=1024, per=True, mem=8192, name='ragged-per-big')
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 3, True)
print('two branches', drv.per_two_branches, 'K', drv.K, flush=True)
drv.prepare(1)
print('captured 1', flush=True)
drv.step()
torch.cuda.synchronize()
print('stepped', flush=True)
drv.prepare(drv.K)
drv.run(2 * drv.K)
torch.cuda.synchronize()
print('ran', flush=True)
