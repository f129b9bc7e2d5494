"""Host time of the phases of UpdateStepDriver._launch (draws / upload / graph replay) inside run(), against the
wall time of the whole run: is the host or the GPU the bottleneck of back-to-back K-row launches?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    cfg = dict(bench.CONFIGS[name])
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
    drv.prepare_all()
    drv.run(4 * drv.K)
    torch.cuda.synchronize()
    T = {'draw': 0.0, 'rng': 0.0, 'upload': 0.0, 'replay': 0.0, 'reshuffle': 0.0, 'launches': 0}

    def timed(obj, attr, key):
        f = getattr(obj, attr)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            T[key] += time.perf_counter() - t
            return r
        setattr(obj, attr, g)
    timed(drv, '_draw', 'draw'); timed(drv, '_draw_rng', 'rng'); timed(drv, '_upload', 'upload'); timed(pool, 'init_pointer', 'reshuffle')
    for key in list(drv.plans):
        def wrap(f):
            def g():
                t = time.perf_counter()
                f()
                T['replay'] += time.perf_counter() - t
                T['launches'] += 1
            return g
        drv.plans[key] = [wrap(f) for f in drv.plans[key]]
    t0 = time.perf_counter()
    drv.run(n)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    L = T['launches']
    print('%s K %d, %d rows in %d launches: host per launch _draw %.0f us (its RNG work %.0f us, of which pool reshuffles %.0f us; the rest waits for a free slot), upload '
          '%.0f us, graph replay %.0f us; host total %.1f ms, wall %.1f ms (%.1f us per row)'
          % (name, drv.K, n, L, 1e6 * T['draw'] / L, 1e6 * T['rng'] / L, 1e6 * T['reshuffle'] / L, 1e6 * T['upload'] / L,
             1e6 * T['replay'] / L, 1e3 * t_host, 1e3 * t_wall, 1e6 * t_wall / n))


if __name__ == '__main__':
    main()
