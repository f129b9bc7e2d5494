"""CPU baseline for bench.py: the reference's update step restated on the host, in the reference's
OWN call structure (TEST/BENCH INFRASTRUCTURE ONLY — never imported by sam_tf_amd/).

The reference cannot run here (TensorFlow 1.12 is absent and unobtainable), so this is
`cpu_baseline.kind = "port"`: fp32 torch on ONE thread (the reference uses a single-threaded session,
run_sam.py:18 -> tf_util.py:97-99) executing, per update step, what sam.py:369-433 executes:

  discriminator  sample_uniform + expert batch, obs_rms.update,
                 get_grads (fwd+bwd), get_losses (fwd), optimize (fwd+bwd+Adam)   sam.py:372-402
  agent          sample, get_targ_q, actor losses (fwd), actor grads (fwd+bwd),
                 critic losses (fwd), critic grads (fwd+bwd), actor optimize, critic optimize,
                 [PER: td_errors + update_priorities]                     sam_agent.py:634-703
  targets        Polyak                                                           sam.py:433

with the replay memory in Python/numpy exactly as oracle/replay_oracle.py restates it.
"""
import random

import numpy as np
import torch

from . import nets_oracle as NO
from .replay_oracle import ReplayOracle, PrioritizedReplayOracle

F32 = torch.float32


class CpuReferenceStep(object):

    def __init__(self, ob, ac, h, B, per=False, mem=1 << 17, seed=0, ln=False):
        rs = np.random.RandomState(seed)
        random.seed(seed); np.random.seed(seed)
        self.ob, self.ac, self.B, self.per = ob, ac, B, per
        self.da = NO.MLP(ob, h[0], h[1], ac, ln)
        self.dc = NO.MLP(ob + ac, h[0], h[1], 1, ln)
        self.dd = NO.MLP(ob + ac, h[0], h[1], 1, ln)
        self.pa = NO.init_params(self.da, rs, 'actor'); self.pa_t = self.pa.copy()
        self.pc = NO.init_params(self.dc, rs, 'critic'); self.pc_t = self.pc.copy()
        self.pd = NO.init_params(self.dd, rs, 'reward')
        self.opt = {k: [np.zeros(p.size), np.zeros(p.size), 0] for k, p in
                    (('a', self.pa), ('c', self.pc), ('d', self.pd))}
        n = mem
        if per:
            self.rb = PrioritizedReplayOracle(n, (ob,), (ac,), 0.3, 1.0)
        else:
            self.rb = ReplayOracle(n, (ob,), (ac,))
        data = dict(obs0=rs.randn(n, ob), acs=rs.uniform(-1, 1, (n, ac)), rews=rs.uniform(0, 5, (n, 1)),
                    obs1=rs.randn(n, ob), dones1=(rs.rand(n, 1) < 1e-3))
        for k, v in data.items():  # bulk prefill (state identical to n appends on an empty buffer)
            self.rb.rings[k].data[:] = v.astype(np.float32)
            self.rb.rings[k].length = n
        if per:
            from .per_tree_c import PerTreesC
            ct = PerTreesC(self.rb.cap)
            ct.fill_range(0, n, 1.0)
            self.rb.sum_tree.v[:] = ct.sumv; self.rb.min_tree.v[:] = ct.minv
        self.e_obs = (rs.randn(32000, ob) + 0.5).astype(np.float32)
        self.e_acs = rs.uniform(-1, 1, (32000, ac)).astype(np.float32)
        self.ptr = 0
        self.rms_mu = NO.rms_update(NO.rms_init(ob), data['obs0'][:4096], ob)
        self.rms_d = NO.rms_init(ob)
        self.rs = rs

    def _adam(self, key, p, g, lr):
        m, v, t = self.opt[key]
        g, _ = NO.clip_by_global_norm(g.astype(np.float64), 5.0)
        th, m, v = NO.adam_tf1(p, m, v, g, lr, t + 1)
        self.opt[key] = [m, v, t + 1]
        return th.astype(np.float32)

    def step(self):
        B, rb = self.B, self.rb
        # ---- discriminator (sam.py:369-406)
        xp = rb.sample_uniform(B) if self.per else rb.sample(B)
        if self.ptr + B >= len(self.e_obs):
            self.ptr = 0
        eo, ea = self.e_obs[self.ptr:self.ptr + B], self.e_acs[self.ptr:self.ptr + B]
        self.ptr += B
        self.rms_d = NO.rms_update(self.rms_d, np.concatenate((xp['obs0'], eo)), self.ob)
        mean, std = NO.rms_mean_std(self.rms_d, self.ob)
        g = None
        for need_grad in (True, False, True):  # get_grads, get_losses, optimize: fresh randoms each
            eps_ob = self.rs.rand(B, self.ob).astype(np.float32)
            eps_ac = self.rs.rand(B, self.ac).astype(np.float32)
            real, fake = NO.smooth_labels(self.rs.rand(B))
            _, g = NO.disc_loss_grad(self.pd, self.dd, xp['obs0'], xp['acs'], eo, ea, eps_ob, eps_ac,
                                     real, fake, mean, std, dtype=F32, need_grad=need_grad)
        self.pd = self._adam('d', self.pd, g, 3e-4)
        # ---- agent (sam_agent.py:631-711)
        b = rb.sample(B)
        mean, std = NO.rms_mean_std(self.rms_mu, self.ob)
        tq = NO.targ_q(self.pa_t, self.pc_t, self.da, self.dc, b['obs1'], b['rews'], b['dones1'],
                       np.ones((B, 1)), mean, std, 0.99, 1.0, dtype=F32)
        iws = b['iws'] if self.per else None
        ga = gc = None
        for need_grad in (False, True, True):  # actor: losses, grads, optimize
            _, ga = NO.actor_loss_grad(self.pa, self.da, self.pc, self.dc, None, None, b['obs0'], mean,
                                       std, 1.0, 1.0, 0.0, dtype=F32, need_grad=need_grad)
        for need_grad in (False, True, True):  # critic: losses, grads, optimize
            _, gc, _ = NO.critic_loss_grad(self.pc, self.dc, b['obs0'], b['acs'], iws, tq, mean, std,
                                           1e-3, dtype=F32, need_grad=need_grad)
        self.pa = self._adam('a', self.pa, ga, 1e-4)
        self.pc = self._adam('c', self.pc, gc, 1e-3)
        if self.per:
            td = NO.td_errors(self.pc, self.dc, b['obs0'], b['acs'], tq, mean, std, dtype=F32)
            rb.update_priorities(b['idxs'], np.abs(td.ravel().astype(np.float64)) + 1e-6)
        # ---- targets (sam.py:433)
        self.pa_t = NO.polyak(self.pa_t, self.pa, 0.01).astype(np.float32)
        self.pc_t = NO.polyak(self.pc_t, self.pc, 0.01).astype(np.float32)
