"""Host-visible latency of the per-environment-step calls (sam.py:50-71): mu.predict, d.get_reward,
mu.store_transition at B = 1 — SURVEY.md 8(f) N3's latency path."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
    cfg = dict(bench.CONFIGS[name], mem=1 << 16)
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
    rs = np.random.RandomState(0)
    ob = rs.randn(cfg['ob']).astype(np.float32)
    ac = rs.uniform(-1, 1, cfg['ac']).astype(np.float32)

    def t(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    print('%s: predict(noise, q) %.0f us, predict(no q) %.0f us, get_reward %.0f us, store_transition %.0f us' % (
        name, t(lambda: mu.predict(ob, True, True)), t(lambda: mu.predict(ob, False, False)),
        t(lambda: d.get_reward(ob, ac)), t(lambda: mu.store_transition(ob, ac, 1.0, ob, False))), flush=True)


if __name__ == '__main__':
    main()
