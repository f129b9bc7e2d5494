"""Per-step, per-variable gradient error of the driver against the fp64 oracle beside the fp32-CPU floor, for the
return-normalisation cases of tests/test_gpu_driver.py::test_k_steps_match_oracle (exploration aid, GPU box).

    python tools/exp/ret_parity_probe.py C1-ret wsmall-wrappedretn ragged-wrappedlauncher
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

import bench  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402
from tests.nets_cases import per_var_relerr  # noqa: E402
from tests.test_gpu_driver import OracleLearner  # noqa: E402

EXTRA = {'ragged': dict(ob=5, ac=2, h=(33, 70), B=37, per=False, name='ragged'),
         'wsmall': dict(ob=20, ac=3, h=(256, 256), B=48, per=False, name='wide-small')}


def run(cfg_name, graph=False, feed=True, sync=False):
    base, _, opt = cfg_name.partition('-')
    wrapped = opt.startswith('wrapped')
    opt = opt[len('wrapped'):] if wrapped else opt
    cfg = dict(EXTRA.get(base) or bench.CONFIGS[base], mem=2048, ln=opt in ('ln', 'launcher'),
               ret=opt in ('launcher', 'ret', 'retn'), nstep=7 if opt in ('launcher', 'retn', 'nstep') else 0, p_done=0.1)
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 3, graph)
    drv.prepare()
    if wrapped:
        rs = np.random.RandomState(9)
        for t in range(45):
            mu.store_transition(rs.randn(mu.ob_dim).astype(np.float32), rs.uniform(-1, 1, mu.ac_dim).astype(np.float32),
                                float(rs.rand()), rs.randn(mu.ob_dim).astype(np.float32), float(t % 13 == 12))
    ref = OracleLearner(mu, d, pool, cfg['per'])
    ref.floor = True
    h = lambda t: t.cpu().numpy()
    print('==== %s sync=%s' % (cfg_name, sync))
    for k in range(6):
        if sync:   # teacher forcing: the oracle starts every update from the device's state
            ref.pa, ref.pc = (x.get_flat().astype(np.float64) for x in (mu.actor, mu.critic))
            ref.pa_t, ref.pc_t = (x.get_flat().astype(np.float64) for x in (mu.targ_actor, mu.targ_critic))
            ref.pd = d.reward_nn.get_flat().astype(np.float64)
            for key, o in (('a', mu.actor_ops['optimizer']), ('c', mu.critic_ops['optimizer']), ('d', d.optimizer)):
                ref.opt[key] = [h(o.m).astype(np.float64), h(o.v).astype(np.float64), int(o.t_dev.item())]
            ref.rms_d = h(d.obs_rms.sums).copy()
            if cfg['ret']:
                ref.ret_sums = h(mu.ret_rms.sums).copy()
        ret_old = (float(mu.ret_rms.mean.item()), float(mu.ret_rms.std.item())) if cfg['ret'] else None
        drv.step()
        torch.cuda.synchronize()
        if cfg['ret'] and feed:
            ref.gpu_ret = dict(old=ret_old, new=(float(mu.ret_rms.mean.item()), float(mu.ret_rms.std.item())))
        rb_ = mu.replay_buffer
        lg = (lambda t: (h(t) - rb_.start) % rb_.limit) if drv.storage_rows else h
        e0 = int(drv.e_start.item())
        eps_ob, eps_ac, real, _ = drv.step_randoms(k)
        ref.gpu_g = {key: h(drv.L.seg(drv.L.grads, seg)) for seg, key in enumerate('acd')}
        try:
            ref.step(lg(drv.idx_p), h(pool.obs0[e0:e0 + drv.B]), h(pool.acs[e0:e0 + drv.B]), h(eps_ob), h(eps_ac), h(real),
                     idx_a=lg(drv.idx_a), u=h(drv.u), gpu_td=h(drv.td_post))
        except AssertionError as e:
            print('  step %d: %s' % (k, str(e)[:300]))
        print('%s step %d  ret old %s -> new %s   oracle sums %s' % (
            cfg_name, k, ret_old, ref.gpu_ret and ref.gpu_ret['new'], ref.ret_sums))
        for seg, (key, desc) in enumerate((('a', ref.da), ('c', ref.dc), ('d', ref.dc))):
            errs = per_var_relerr(h(drv.L.seg(drv.L.grads, seg)), ref.last_g[key], desc)
            fl = ref.floor_g.get(key, {})
            print('   %s: ' % key + '  '.join('%s %.1e/%.1e%s' % (n, errs[n], fl.get(n, 0.0), ' <<<' if errs[n] > max(
                1e-5, 4 * fl.get(n, 0.0)) else '') for n in errs))
    print('   ret margins', ref.ret_margins, flush=True)


if __name__ == '__main__':
    for name in sys.argv[1:]:
        run(name, sync=os.environ.get('PROBE_SYNC') == '1')
