import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from sam_tf_amd.comm import TorchComm
from tests.test_gpu_driver import OracleLearner
from oracle import nets_oracle as NO
cfg = dict(bench.CONFIGS['C4'], mem=2048, ln=False, ret=False, nstep=0, p_done=0.1)
mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), torch.device('cuda', 0), 3, False)
drv.prepare()
ref = OracleLearner(mu, d, pool, cfg['per'])
h = lambda t: t.cpu().numpy()
pd0 = ref.pd.copy(); rms0 = ref.rms_d.copy()
drv.step(); torch.cuda.synchronize()
e0 = int(drv.e_start.item())
eps_ob, eps_ac, real, _ = drv.step_randoms(0)
p = ref.rows(h(drv.idx_p))
e_obs, e_acs = h(pool.obs0[e0:e0 + drv.B]), h(pool.acs[e0:e0 + drv.B])
ob = ref.da.in_dim
rms = NO.rms_update(rms0, np.concatenate((p['obs0'], e_obs)), ob)
mean, std = NO.rms_mean_std(rms, ob)
args = (pd0, ref.dc, p['obs0'], p['acs'], e_obs, e_acs, h(eps_ob), h(eps_ac), h(real), np.zeros(drv.B), mean, std)
_, g_full = NO.disc_loss_grad(*args)
NO.GP_COEFF = 0.0
_, g_base = NO.disc_loss_grad(*args)
g_gp = g_full - g_base
g_gpu = h(drv.L.seg(drv.L.grads, 2)).astype(np.float64)
diff = g_gpu - g_full
sl = NO.param_slices(ref.dc)
for name, (off, shape) in sl.items():
    n = int(np.prod(shape))
    a, b, dd_ = g_gp[off:off+n], g_base[off:off+n], diff[off:off+n]
    proj = float(np.dot(dd_, a) / max(np.dot(a, a), 1e-300))
    print(name, 'max|base| %.3e max|gp| %.3e max|diff| %.3e  diff projected on gp part: %.3e  residual %.3e' % (np.abs(b).max(), np.abs(a).max(), np.abs(dd_).max(), proj, np.abs(dd_ - proj * a).max()))
print('gpu mean/std vs oracle', np.abs(h(d.obs_rms.mean) - mean).max(), np.abs(h(d.obs_rms.std) - std).max())
# the same update through the dense API (materialised rows and randoms, statistics as plain mean/std)
from sam_tf_amd import ops
from tests.test_gpu_step import HP, _dev
mk = lambda dsc: ops.mlp_desc(dsc.in_dim, dsc.h1, dsc.h2, dsc.out_dim, dsc.layernorm)
L2 = ops.LearnerState(mk(ref.da), mk(ref.dc), mk(ref.dc), ob, ref.da.out_dim, dict(HP))
pa0 = mu.actor.get_flat(); pc0 = mu.critic.get_flat()
L2.params.copy_(_dev(np.concatenate([pa0, pc0, pd0.astype(np.float32)])))
L2.targets.copy_(_dev(np.concatenate([pa0, pc0])))
for kk in ('mu_mean', 'd_mean'): L2.set_ptr(kk, _dev(mean))
for kk in ('mu_std', 'd_std'): L2.set_ptr(kk, _dev(std))
B = drv.B
f32 = lambda a: _dev(np.asarray(a, dtype=np.float32))
batch = ops.make_batch(B, p_obs=f32(p['obs0']), p_acs=f32(p['acs']), e_obs=f32(e_obs), e_acs=f32(e_acs), eps_ob=eps_ob.contiguous(), eps_ac=eps_ac.contiguous(),
                       real_labels=real.contiguous(), fake_labels=torch.zeros(B, device='cuda'), obs0=f32(p['obs0']), acs=f32(p['acs']), rews=f32(p['rews']).view(-1), dones1=f32(p['dones1']).view(-1), obs1=f32(p['obs1']))
out2 = ops.make_step_out(B)
ops.step_grads(L2, batch, out2); torch.cuda.synchronize()
g2 = h(L2.seg(L2.grads, 2)).astype(np.float64)
for name, (off, shape) in sl.items():
    n = int(np.prod(shape))
    print('dense API', name, 'relerr vs oracle %.3e   vs driver-flow gpu %.3e' % (np.abs(g2[off:off+n] - g_full[off:off+n]).max() / np.abs(g_full[off:off+n]).max(), np.abs(g2[off:off+n] - g_gpu[off:off+n]).max() / np.abs(g_full[off:off+n]).max()))
# per-tile: the same update on 16-row sub-batches
NO.GP_COEFF = 10.0
worst = []
for t in range(0, 64):
    r = slice(16 * t, 16 * t + 16)
    bt = ops.make_batch(16, p_obs=f32(p['obs0'][r]), p_acs=f32(p['acs'][r]), e_obs=f32(e_obs[r]), e_acs=f32(e_acs[r]), eps_ob=eps_ob[r].contiguous(), eps_ac=eps_ac[r].contiguous(),
                        real_labels=real[r].contiguous(), fake_labels=torch.zeros(16, device='cuda'), obs0=f32(p['obs0'][r]), acs=f32(p['acs'][r]), rews=f32(p['rews'][r]).view(-1), dones1=f32(p['dones1'][r]).view(-1), obs1=f32(p['obs1'][r]))
    o16 = ops.make_step_out(16)
    ops.step_grads(L2, bt, o16); torch.cuda.synchronize()
    gg = h(L2.seg(L2.grads, 2)).astype(np.float64)
    _, ge = NO.disc_loss_grad(pd0, ref.dc, p['obs0'][r], p['acs'][r], e_obs[r], e_acs[r], h(eps_ob)[r], h(eps_ac)[r], h(real)[r], np.zeros(16), mean, std)
    off, shape = sl['W2']; n = int(np.prod(shape))
    worst.append((t, np.abs(gg[off:off+n] - ge[off:off+n]).max() / np.abs(ge[off:off+n]).max()))
print('per-tile W2 relerr (B = 16):', ' '.join('%d:%.1e' % w for w in worst), 'path', L2.path(16))

for Bs in (64, 256, 512, 1024):
    r = slice(0, Bs)
    bt = ops.make_batch(Bs, p_obs=f32(p['obs0'][r]), p_acs=f32(p['acs'][r]), e_obs=f32(e_obs[r]), e_acs=f32(e_acs[r]), eps_ob=eps_ob[r].contiguous(), eps_ac=eps_ac[r].contiguous(),
                        real_labels=real[r].contiguous(), fake_labels=torch.zeros(Bs, device='cuda'), obs0=f32(p['obs0'][r]), acs=f32(p['acs'][r]), rews=f32(p['rews'][r]).view(-1), dones1=f32(p['dones1'][r]).view(-1), obs1=f32(p['obs1'][r]))
    oB = ops.make_step_out(Bs)
    ops.step_grads(L2, bt, oB); torch.cuda.synchronize()
    gg = h(L2.seg(L2.grads, 2)).astype(np.float64)
    ls, ge = NO.disc_loss_grad(pd0, ref.dc, p['obs0'][r], p['acs'][r], e_obs[r], e_acs[r], h(eps_ob)[r], h(eps_ac)[r], h(real)[r], np.zeros(Bs), mean, std)
    off, shape = sl['W2']; n = int(np.prod(shape))
    print('B', Bs, 'W2 relerr %.2e' % (np.abs(gg[off:off+n] - ge[off:off+n]).max() / np.abs(ge[off:off+n]).max()), 'gp gpu %.8f ref %.8f' % (float(oB.t['d_losses'][6]), ls[6]))
t64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
r = slice(144, 160)
pz = NO.obz(t64(p['obs0'][r]), t64(mean), t64(std)).numpy(); ez = NO.obz(t64(e_obs[r]), t64(mean), t64(std)).numpy()
xp = np.concatenate([pz, p['acs'][r]], axis=1); xe = np.concatenate([ez, e_acs[r]], axis=1)
eps = np.concatenate([h(eps_ob)[r], h(eps_ac)[r]], axis=1).astype(np.float64)
xi = eps * xp + (1 - eps) * xe
for nm, x in (('policy', xp), ('expert', xe), ('interp', xi)):
    print('tile 9', nm, 'min |preact| per row:', ' '.join('%.1e' % v for v in NO.min_abs_preact(pd0, ref.dc, x)))
