"""Times samx_targ_q (5 layer GEMMs + 2 row kernels) at a given config with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sam_tf_amd import ops
from tests.nets_cases import CONFIGS
name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
ob, ac, h1, h2, B = CONFIGS[name]
da, dc = ops.mlp_desc(ob, h1, h2, ac), ops.mlp_desc(ob + ac, h1, h2, 1)
g = torch.Generator(device='cuda'); g.manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device='cuda')
pa, pc = r(ops.param_count(da)) * 0.05, r(ops.param_count(dc)) * 0.05
obs1, rews, dn = r(B, ob), r(B), torch.zeros(B, device='cuda')
ws = ops.nets_ws(da, dc, None, B)
out = torch.zeros(B, device='cuda')
f = lambda: ops.targ_q(da, pa, dc, pc, obs1, rews, dn, None, None, None, 0.99, 1.0, ws, out=out)
for _ in range(20): f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(10): f()
f = gr.replay
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); e1.synchronize()
print('%s dbg=%s targ_q: %.1f us' % (name, os.environ.get('SAMX_GEMM_DBG', '0'), e0.elapsed_time(e1) / 200 * 1e3))
