"""Where a prioritized-replay row's wall time goes on the host (C3): the phases of
UpdateStepDriver._launch in the one-process graph mode (per_graph_io)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SAMX_PER_HOST_POW'] = '1'   # this tool times the HOST refresh (the device refresh has no host part)
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


def main():
    cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'C3'])
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
    assert drv.per_graph_io
    drv.prepare(1)
    drv.run(200)
    T = {}

    def tick(name, t0):
        t = time.perf_counter()
        T[name] = T.get(name, 0.0) + (t - t0)
        return t
    N = 2000
    rb = mu.replay_buffer
    stream = torch.cuda.current_stream()
    for _ in range(N):
        t = time.perf_counter()
        slot = drv.slots[drv.launch_no % len(drv.slots)]
        drv._draw(slot, 1, 0); t = tick('fill pending', t)
        np.copyto(drv._launch_np, slot.np_words[:1]); t = tick('memcpy launch words', t)
        for run in drv.plans[(1, 0)]:
            run()
        t = tick('graph launch (host)', t)
        nxt = drv.slots[(drv.launch_no + 1) % len(drv.slots)]
        if nxt.used:
            nxt.event.synchronize()
        drv._draw_rng(nxt, 1, 0); nxt.predrawn = True; t = tick('predraw next row (overlapped)', t)
        stream.synchronize(); t = tick('wait for the GPU', t)
        idxs = drv._idx_host_np.copy()
        leaf = rb.leaves_from_td(idxs, drv._td_host_np, drv._leaf_np).copy()
        drv._pending = (idxs, leaf); t = tick('leaves (libm pow)', t)
        drv.step_no += 1; drv.launch_no += 1
    tot = sum(T.values())
    for k, v in T.items():
        print('%-32s %6.1f us' % (k, v / N * 1e6))
    print('%-32s %6.1f us' % ('total', tot / N * 1e6))


if __name__ == '__main__':
    main()
