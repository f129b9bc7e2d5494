"""Where the 20-step timed region of the driver's bench invocation goes: host phases of the one launch, then the
wait for the GPU (from an idle, synchronised start, as bench.py times it)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cfg = dict(bench.CONFIGS['C2'])
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
    drv.prepare_all()
    drv.prepare(n)
    drv.run(5)
    T = {'draw': 0.0, 'upload': 0.0, 'replay': 0.0, 'total': 0.0, 'host': 0.0}

    def timed(obj, attr, key):
        f = getattr(obj, attr)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            T[key] += time.perf_counter() - t
            return r
        setattr(obj, attr, g)
    timed(drv, '_draw', 'draw'); timed(drv, '_upload', 'upload')
    f0 = drv.plans[(n, 0)][0]

    def rep():
        t = time.perf_counter()
        f0()
        T['replay'] += time.perf_counter() - t
    drv.plans[(n, 0)] = [rep]
    R = 30
    for _ in range(R):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.run(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        T['host'] += t1 - t0; T['total'] += t2 - t0
    print('run(%d) from idle, mean of %d: total %.0f us = host %.0f us (draw %.0f, upload %.0f, graph launch %.0f, rest '
          '%.0f) + wait for the GPU %.0f us  -> %.0f steps/s' %
          (n, R, 1e6 * T['total'] / R, 1e6 * T['host'] / R, 1e6 * T['draw'] / R, 1e6 * T['upload'] / R,
           1e6 * T['replay'] / R, 1e6 * (T['host'] - T['draw'] - T['upload'] - T['replay']) / R,
           1e6 * (T['total'] - T['host']) / R, n * R / T['total']))


if __name__ == '__main__':
    main()
