"""Does any kernel of the step read workspace memory it did not write?  Fill the learner's step workspace with NaN
bytes before every call and look for NaN in the results (python tools/exp/nan_workspace.py [C4] [--launcher])."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from sam_tf_amd import ops
from sam_tf_amd.comm import TorchComm

name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
opt = sys.argv[2] if len(sys.argv) > 2 else ''
cfg = dict(bench.CONFIGS[name], mem=4096, ln=opt in ('--launcher', '--layernorm'), ret=opt == '--launcher',
           nstep=7 if opt == '--launcher' else 0, p_done=0.1)
dev = torch.device('cuda', 0)
mu, d, pool, drv = bench.build_learner(cfg, TorchComm(dev), dev, 3, False)
drv.prepare()
L = drv.L
for k in range(4):
    for B, ws in list(L._ws.items()) if hasattr(L, '_ws') else []:
        ws.view(torch.float32)[: ws.numel() // 4].fill_(float('nan'))
    drv.step()
    torch.cuda.synchronize()
    g = L.grads
    print('step', k, 'grads finite', bool(torch.isfinite(g).all()), 'params finite', bool(torch.isfinite(L.params).all()),
          'nan count', int(torch.isnan(g).sum()), flush=True)
