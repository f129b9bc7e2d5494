"""Where does the device's free-running trajectory leave the fp64 oracle's?  (exploration aid, GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from oracle import nets_oracle as NO  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402
from tests.nets_cases import per_var_relerr  # noqa: E402
from tests.test_gpu_driver import OracleLearner  # noqa: E402

cfg = dict(ob=20, ac=3, h=(256, 256), B=48, per=False, name='wide-small', mem=2048, ln=False, ret=True, nstep=7, p_done=0.1)
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 3, False)
drv.prepare()
rs = np.random.RandomState(9)
for t in range(45):
    mu.store_transition(rs.randn(mu.ob_dim).astype(np.float32), rs.uniform(-1, 1, mu.ac_dim).astype(np.float32),
                        float(rs.rand()), rs.randn(mu.ob_dim).astype(np.float32), float(t % 13 == 12))
f64 = OracleLearner(mu, d, pool, False)
f32 = OracleLearner(mu, d, pool, False, dtype=torch.float32)
h = lambda t: t.cpu().numpy()
sl = NO.param_slices(f64.dc)
for k in range(6):
    drv.step()
    torch.cuda.synchronize()
    rb_ = mu.replay_buffer
    lg = (lambda t: (h(t) - rb_.start) % rb_.limit)
    e0 = int(drv.e_start.item())
    eps_ob, eps_ac, real, _ = drv.step_randoms(k)
    args = (lg(drv.idx_p), h(pool.obs0[e0:e0 + drv.B]), h(pool.acs[e0:e0 + drv.B]), h(eps_ob), h(eps_ac), h(real))
    kw = dict(idx_a=lg(drv.idx_a), u=h(drv.u), gpu_td=h(drv.td_post))
    if k in (3, 5):   # the pre-activations of unit 236 of the critic's first layer in the loss pass (fp64, fp32 CPU)
        b = f64.rows(kw['idx_a'])
        mean, std = NO.rms_mean_std(f64.rms_mu, mu.ob_dim)
        for dt in (torch.float64, torch.float32):
            P = NO.unflatten(torch.as_tensor(f64.pc, dtype=dt), f64.dc)
            x = torch.cat([NO.obz(torch.as_tensor(b['obs0'], dtype=dt), torch.as_tensor(mean, dtype=dt), torch.as_tensor(std, dtype=dt)),
                           torch.as_tensor(b['acs'], dtype=dt)], dim=1)
            z = (x @ P['W1'] + P['b1'])[:, 236]
            i = int(z.abs().argmin())
            print('  unit 236 %s: min |z| = %.3e at row %d (z = %.6e); typical |z| %.2e' % (dt, float(z.abs().min()), i, float(z[i]), float(z.abs().mean())))
    for fr in (f64, f32):
        fr.check_td = False
        fr.step(*args, **kw)
    gdev = h(drv.L.seg(drv.L.grads, 1)).astype(np.float64)
    pdev = mu.critic.get_flat().astype(np.float64)
    print('step', k)
    print('  grad  dev-f64', {n: '%.1e' % e for n, e in per_var_relerr(gdev, f64.last_g['c'], f64.dc).items()})
    print('  grad  f32-f64', {n: '%.1e' % e for n, e in per_var_relerr(f32.last_g['c'], f64.last_g['c'], f64.dc).items()})
    print('  param dev-f64', {n: '%.1e' % e for n, e in per_var_relerr(pdev, f64.pc, f64.dc).items()})
    print('  param f32-f64', {n: '%.1e' % e for n, e in per_var_relerr(f32.pc, f64.pc, f64.dc).items()})
    o, shp = sl['W1']
    n = int(np.prod(shp))
    diff = np.abs(pdev[o:o + n] - f64.pc[o:o + n])
    i = int(np.argmax(diff))
    g64, g32 = f64.last_g['c'][o:o + n], f32.last_g['c'][o:o + n]
    gmax = np.max(np.abs(g64))
    print('  W1 worst elem %d (row %d col %d): dparam %.3e  |W1|max %.3e   g64 %.3e gdev %.3e g32 %.3e  gmax %.3e  m %.3e v %.3e'
          % (i, i // shp[1], i % shp[1], diff[i], np.max(np.abs(f64.pc[o:o + n])), g64[i], gdev[o + i], g32[i], gmax,
             f64.opt['c'][0][o + i], f64.opt['c'][1][o + i]))
    # how many W1 elements have |g| below 1e-5 of the max
    print('  W1 elements with |g| < 1e-5 gmax: %d of %d; clipnorm factor: |g| = %.3e' % (
        int((np.abs(g64) < 1e-5 * gmax).sum()), n, np.linalg.norm(f64.last_g['c'])))
