"""Host cost of the pieces of the N > 1 plan (1-rank RCCL group on one GPU)."""
import os, sys, time
import torch, torch.distributed as dist
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29613', rank=0, world_size=1, device_id=dev)
t = torch.zeros(13000, device=dev)
def T(name, f, n=3000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-28s host %.1f us  total %.1f us' % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
T('dist.all_reduce', lambda: dist.all_reduce(t))
T('mul_', lambda: t.mul_(1.0))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    t.mul_(1.0)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    t.mul_(1.0); t.mul_(1.0)
T('graph replay (2 kernels)', g.replay)
pg = dist.distributed_c10d._get_default_group()
T('pg.allreduce([t])', lambda: pg.allreduce([t]))
def both():
    g.replay(); pg.allreduce([t])
T('replay + allreduce', both)
T('dist.all_reduce (again)', lambda: dist.all_reduce(t))
T('mul_ (again)', lambda: t.mul_(1.0))
def ar_mul():
    dist.all_reduce(t); t.mul_(1.0)
T('all_reduce + mul_', ar_mul)
def rep_mul():
    g.replay(); t.mul_(1.0)
T('replay + mul_', rep_mul)
T('graph replay again', g.replay)
dist.destroy_process_group()
