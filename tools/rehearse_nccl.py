"""Rehearsal of the N > 1 driver plan on ONE GPU: a real 1-rank RCCL process group carries every
collective of the plan (so graph replays and RCCL kernels interleave on the stream exactly as at N > 1),
and the communicator reports 2 ranks whose second rank is an identical twin (SUM doubles).  Times the
host + launch + RCCL-kernel overheads of the plan; the xGMI hop itself is not in it."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402


class TwinOverRccl(TorchComm):
    def __init__(self, device, direct):
        TorchComm.__init__(self, device)
        if direct:
            from sam_tf_amd.rccl import RcclComm
            self.direct = RcclComm(device)

    def Get_size(self):
        return 2

    def allreduce_(self, t):
        if self.direct is not None:
            self.direct.all_reduce_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.mul_(2)

    def bcast_(self, t, root=0):
        dist.broadcast(t, src=0)
        return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='C2')
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--port', type=int, default=29611)
    ap.add_argument('--direct', type=int, default=1)
    ap.add_argument('--in-graph', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % a.port, rank=0, world_size=1,
                            device_id=dev)
    comm = TwinOverRccl(dev, a.direct)
    cfg = dict(bench.CONFIGS[a.config])
    mu, d, pool, drv = bench.build_learner(cfg, comm, dev, 0, True)
    drv.set_collectives(a.in_graph)
    print('in_graph', drv.collectives_in_graph, 'world', drv.world, 'K', drv.K, 'inline_stats', drv.inline_stats, flush=True)
    drv.prepare_all()
    drv.run(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.run(a.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('%s: %.1f us/step, %.0f steps/s per rank; finite %s' %
          (a.config, el / a.steps * 1e6, a.steps / el, bool(torch.isfinite(mu.params).all())), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
