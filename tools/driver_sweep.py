"""Sweep of UpdateStepDriver over shapes / options / schedules: every case must run 2 launches' worth of
rows with finite parameters.  Looks for workspace / shape errors outside the pinned configurations."""
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402
from sam_tf_amd.driver import UpdateStepDriver  # noqa: E402


class TwinComm(object):
    """2 ranks whose second rank is an identical twin (SUM doubles): drives the N > 1 plans on one GPU."""

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 2

    def Barrier(self):
        pass

    def allreduce_(self, t):
        return t.mul_(2)

    def bcast_(self, t, root=0):
        return t

    def Allreduce(self, sendbuf, recvbuf, op='sum'):
        recvbuf[...] = 2 * np.asarray(sendbuf)

    def Bcast(self, buf, root=0):
        pass


def main(twin=False, quick=False):
    shapes = [(4, 1, (64, 64), 32), (11, 3, (64, 64), 256), (20, 4, (64, 64), 48), (376, 17, (256, 256), 64),
              (5, 2, (33, 70), 37), (50, 10, (64, 64), 32), (3, 1, (8, 8), 1), (111, 8, (400, 300), 16),
              (17, 6, (64, 64), 2048), (17, 6, (64, 64), 512)]
    bad = n = 0
    for ob, ac, hid, B in shapes:
        for per in (False, True):
            for ln, ret, nstep in ((False, False, 0), (True, True, 0 if per else 5), (False, True, 0 if per else 3)):
                for graph, g_steps in (((True, 3), (False, 1)) if quick else ((True, 1), (True, 3), (False, 1))):
                    cfg = dict(ob=ob, ac=ac, h=hid, B=B, per=per, mem=max(4 * B, 512), name='sweep', ln=ln, ret=ret,
                               nstep=nstep, p_done=0.1)
                    n += 1
                    try:
                        mu, d, pool, _ = bench.build_learner(cfg, TwinComm() if twin else TorchComm('cuda'), torch.device('cuda', 0), 1, graph)
                        drv = UpdateStepDriver(mu, d, pool, use_graph=graph, steps_per_launch=6, g_steps=g_steps)
                        drv.run(2 * drv.K + 1)
                        drv.flush()
                        torch.cuda.synchronize()
                        assert bool(torch.isfinite(mu.L.params).all())
                        print('ok  ', ob, ac, hid, B, per, ln, ret, nstep, graph, g_steps, 'fused' if drv.fused else 'lw',
                              flush=True)
                    except Exception as e:  # noqa: BLE001
                        bad += 1
                        print('FAIL', ob, ac, hid, B, per, ln, ret, nstep, graph, g_steps, repr(e)[:300], flush=True)
                        traceback.print_exc(limit=4)
    print('%d cases, %d failed' % (n, bad), flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(len(sys.argv) > 1 and sys.argv[1] == 'twin'))
