"""Host cost of launching the captured C3 row graph: torch's CUDAGraph.replay() against hipGraphLaunch on the raw
executable graph through ctypes."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from sam_tf_amd.comm import TorchComm
cfg = dict(bench.CONFIGS['C3'])
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 0, True)
drv.prepare(1); drv.run(50)
hip = C.CDLL('libamdhip64.so')
graphs = [o for o in __import__('gc').get_objects() if isinstance(o, torch.cuda.CUDAGraph)]
print('graphs alive', len(graphs))
g = graphs[-1]
ex = C.c_void_p(g.raw_cuda_graph_exec())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, fn in (('torch replay', g.replay), ('hipGraphLaunch', lambda: hip.hipGraphLaunch(ex, st))):
    torch.cuda.synchronize()
    tl = 0.0
    N = 500
    for _ in range(N):
        t = time.perf_counter(); fn(); tl += time.perf_counter() - t
        torch.cuda.current_stream().synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn(); torch.cuda.current_stream().synchronize()
    print('%-16s host call %.1f us, call+sync %.1f us' % (name, tl / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
