"""Host time of the parts of UpdateStepDriver._draw for a K-row launch at C2 (no GPU work in the timed parts)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sam_tf_amd import ops  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cfg = dict(bench.CONFIGS['C2'])
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
    slot = drv.slots[0]
    N = mu.replay_buffer.num_entries

    def t(f, n=300):
        f()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t0) / n * 1e6

    def rng():
        pool.pointer = 0
        drv._draw_rng(slot, K, 0)
    print('rows %d, host_rng %s, rb.start %d' % (K, drv.host_rng, mu.replay_buffer.start))
    print('_draw_rng            %6.1f us' % t(rng))
    print('_to_storage_rows     %6.1f us' % t(lambda: drv._to_storage_rows(slot, K)))
    print('_fill_pending        %6.1f us' % t(lambda: drv._fill_pending(slot, 0)))
    print('host_rng_ready       %6.1f us' % t(ops.host_rng_ready))
    if ops.host_rng_ready():
        print('randint lib          %6.1f us' % t(lambda: ops.host_randint_rows(slot.np_words, K, 2 * drv.B, N)))
    print('randint numpy        %6.1f us' % t(lambda: np.random.randint(low=0, high=N, size=(K, 2, drv.B))))
    tmp = np.random.randint(low=0, high=N, size=(K, 2, drv.B))

    def cp():
        slot.np_idx[:K] = tmp
    print('copy into the words  %6.1f us' % t(cp))
    print('np_e arange          %6.1f us' % t(lambda: slot.np_e.__setitem__(slice(0, K), 5 + drv.B * np.arange(K, dtype=np.int64))))
    print('_roles               %6.1f us' % t(lambda: drv._roles(K, 0)))


if __name__ == '__main__':
    main()
