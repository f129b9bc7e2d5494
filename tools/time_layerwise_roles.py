"""Device time of each role of the layer-wise path alone (graph-captured, HIP events): target Q,
critic gradient, actor gradient, discriminator gradient.   python tools/time_layerwise_roles.py [C4]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_tf_amd import ops  # noqa: E402
from tests.nets_cases import CONFIGS  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C4'
ln = len(sys.argv) > 2 and sys.argv[2] == 'ln'
ob, ac, h1, h2, B = CONFIGS[name]
da, dc = ops.mlp_desc(ob, h1, h2, ac, ln), ops.mlp_desc(ob + ac, h1, h2, 1, ln)
g = torch.Generator(device='cuda'); g.manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g, device='cuda')
u = lambda *s: torch.rand(*s, generator=g, device='cuda')
pa, pc, pd = (r(ops.param_count(d)) * 0.05 for d in (da, dc, dc))
obs0, obs1, acs, rews, dn = r(B, ob), r(B, ob), u(B, ac) * 2 - 1, u(B), torch.zeros(B, device='cuda')
e_obs, e_acs = r(B, ob), u(B, ac) * 2 - 1
eps_ob, eps_ac, real = u(B, ob), u(B, ac), 1.2 - 0.5 * u(B)
mean, std = torch.zeros(ob, device='cuda'), torch.ones(ob, device='cuda')
ws, wsd = ops.nets_ws(da, dc, dc, B), ops.disc_ws(dc, B)
tq = torch.zeros(B, device='cuda')
roles = {
    'targ_q': lambda: ops.targ_q(da, pa, dc, pc, obs1, rews, dn, None, mean, std, 0.99, 1.0, ws, out=tq),
    'critic_grad': lambda: ops.critic_grad(dc, pc, obs0, acs, None, tq, mean, std, 1.0, 1e-3, ws),
    'actor_grad': lambda: ops.actor_grad(da, pa, dc, pc, dc, pd, obs0, mean, std, 1.0, 1.0, 1.0, ws),
    'disc_grad': lambda: ops.disc_grad(dc, pd, obs0, acs, e_obs, e_acs, eps_ob, eps_ac, real, None, mean, std,
                                       0.0, wsd),
}
for nm, f in roles.items():
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            f()
    for _ in range(3):
        gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gr.replay()
    e1.record(); e1.synchronize()
    print('%s%s %-12s %.1f us' % (name, ' ln' if ln else '', nm, e0.elapsed_time(e1) / 200 * 1e3), flush=True)
