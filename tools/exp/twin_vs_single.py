"""C4 (or another config): a one-process learner beside a twin-rank (N = 2, identical twin) learner, step by step.
With twin ranks the all-reduce doubles and the optimiser pass halves: parameters must agree bit for bit."""
import sys, random
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from sam_tf_amd.comm import TorchComm
from tests.test_gpu_driver import _TwinComm

name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
cfg = dict(bench.CONFIGS[name], mem=2048, ln=False, ret=False, nstep=0, p_done=0.1)
runs = []
for comm in (TorchComm('cuda'), _TwinComm()):
    mu, d, pool, drv = bench.build_learner(cfg, comm, torch.device('cuda', 0), 3, True)
    np.random.seed(5); random.seed(5); pool.pointer = 0
    drv.prepare()
    snaps = []
    for k in range(3):
        drv.step()
        torch.cuda.synchronize()
        c = lambda t: t.double().cpu().numpy().copy()
        snaps.append(dict(idx_a=c(drv.idx_a), idx_p=c(drv.idx_p), grads=c(mu.L.grads), params=c(mu.L.params), m=c(mu.L.m),
                          v=c(mu.L.v), dsums=c(d.obs_rms.sums), dmean=c(d.obs_rms.mean),
                          seg=[c(mu.L.seg(mu.L.params, s)) for s in range(3)],
                          gseg=[c(mu.L.seg(mu.L.grads, s)) for s in range(3)]))
    runs.append(snaps)
for k in range(3):
    a, b = runs[0][k], runs[1][k]
    print('step', k, 'idx equal', np.array_equal(a['idx_a'], b['idx_a']), np.array_equal(a['idx_p'], b['idx_p']))
    for nm in ('grads', 'params', 'm', 'v', 'dsums', 'dmean'):
        x, y = a[nm], b[nm]
        if nm in ('grads', 'dsums'):
            y = y / 2
        print('   %-7s max|diff| %.3e  max|a| %.3e  n differing %d / %d' % (nm, np.max(np.abs(x - y)), np.max(np.abs(x)),
                                                                           int(np.sum(x != y)), x.size))
    for s, nm in enumerate(('actor', 'critic', 'reward')):
        print('   params/%s differing %d max %.3e ; grads differing %d max %.3e' % (
            nm, int(np.sum(a['seg'][s] != b['seg'][s])), np.max(np.abs(a['seg'][s] - b['seg'][s])),
            int(np.sum(a['gseg'][s] != b['gseg'][s] / 2)), np.max(np.abs(a['gseg'][s] - b['gseg'][s] / 2))))
