"""SAMAgent with the reference's surface (imitation_algorithms/sam_agent.py:61-749) on libsamx.

State in HBM (float32 unless noted):
  params  = [actor | critic]            one flat vector -> ONE all-reduce bucket per update
  targets = [targ_actor | targ_critic]  same layout, Polyak-tracked in one launch
  grads, Adam m/v with the same layout; obs running sums (float64); replay storage (memory.py)
`train()` keeps the reference's return value (numpy, sam_agent.py:706-711).  It runs the agent roles of the
whole-step entry points (samx_step_grads_part without the discriminator role + samx_step_apply on the actor /
critic segments, no Polyak: the fused / role-fused / layer-wise kernels the update-step driver uses), on rows read
in place from the replay storage; the returned dict is ONE host synchronisation — the kernels store the clipped
gradients and the losses straight into pinned host memory.  `train_per_role()` is the same update through the
per-role entry points (samx_targ_q / samx_actor_grad / samx_critic_grad / samx_clip_adam).

Return normalisation + PopArt (sam_agent.py:161-201,554-590,662-672) and n-step returns
(sam_agent.py:483-498,634-660 ; memory.py:113-175) and adaptive parameter-space noise
(sam_agent.py:40-58,352-404,727-749) are built.
"""
import ctypes as C
import os
import types

import numpy as np
import torch

from . import _lib
from . import memory as XP
from . import ops
from .hps import check_supported
from .mpi_adam import MpiAdam
from .networks import FlatNet, init_flat
from .running_mean_std import DeviceRunningMeanStd


# SAMX_OBJECT_API=per_role: the object mirror's update calls go through the per-role entry points (A/B runs)
_PER_ROLE = os.environ.get('SAMX_OBJECT_API', '') == 'per_role'


class NormalAcNoise(object):
    """imitation_algorithms/ac_noise.py:15-28."""

    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma

    def generate(self):
        return np.random.normal(self.mu, self.sigma)

    def reset(self):
        pass


class OUAcNoise(object):
    """imitation_algorithms/ac_noise.py:30-59 (Ornstein-Uhlenbeck)."""

    def __init__(self, mu, sigma, theta=0.15, dt=1e-2, noise_0=None):
        self.mu, self.sigma, self.theta, self.dt, self.noise_0 = mu, sigma, theta, dt, noise_0
        self.reset()

    def generate(self):
        noise = (self.prev_noise + self.theta * (self.mu - self.prev_noise) * self.dt +
                 self.sigma * np.sqrt(self.dt) * np.random.normal(size=self.mu.shape))
        self.prev_noise = noise
        return noise

    def reset(self):
        self.prev_noise = self.noise_0 if self.noise_0 is not None else np.zeros_like(self.mu)


class AdaptiveParamNoise(object):
    """imitation_algorithms/param_noise.py:1-41."""

    def __init__(self, initial_std=0.1, delta=0.1):
        self.initial_std, self.delta, self.cur_std = initial_std, delta, initial_std

    def adapt_std(self, dist):
        if dist < self.delta:
            self.cur_std *= 1.01
        else:
            self.cur_std /= 1.01

    def adapt_delta(self, new_delta):
        self.delta = new_delta

    def __repr__(self):
        return "AdaptiveParamNoise(initial_std={}, delta={})".format(self.initial_std, self.delta)


class SAMAgent(object):

    def __init__(self, name, comm, env, hps, d, device='cuda'):
        check_supported(hps)
        self.name, self.scope, self.comm, self.env, self.hps = name, name, comm, env, hps
        self.device = torch.device(device)
        self.ob_shape = env.observation_space.shape
        self.ac_space = env.action_space
        self.ac_shape = self.ac_space.shape
        self.ob_dim, self.ac_dim = int(self.ob_shape[0]), int(self.ac_shape[0])
        assert all(self.ac_space.low == -self.ac_space.high), "non-admissible ac space bounds"
        self.ac_high = float(self.ac_space.high[0])
        assert all(c == self.ac_high for c in self.ac_space.high)
        assert hps.n > 1 or not hps.n_step_returns
        self.param_noise, self.ac_noise = self.parse_noise_type(hps.noise_type)

        h1, h2 = hps.hid_widths
        self.da = ops.mlp_desc(self.ob_dim, h1, h2, self.ac_dim, hps.with_layernorm)
        self.dc = ops.mlp_desc(self.ob_dim + self.ac_dim, h1, h2, 1, hps.with_layernorm)
        self.Pa, self.Pc = ops.param_count(self.da), ops.param_count(self.dc)
        rs = np.random.RandomState(hps.seed + 104729)
        init = np.concatenate([init_flat(self.da, rs, 'actor', hps.hid_w_init),
                               init_flat(self.dc, rs, 'critic', hps.hid_w_init)])
        # learner-wide flat state [actor | critic | reward-net]: the discriminator's vectors are
        # re-homed into it so that ONE all-reduce / ONE optimiser launch serve the whole step
        self.L = ops.LearnerState(self.da, self.dc, d.desc, self.ob_dim, self.ac_dim, dict(
            ac_high=self.ac_high, gamma=hps.gamma, polyak=hps.polyak,
            clip_norm=hps.clip_norm if hps.clip_norm else 0.0, actor_lr=hps.actor_lr,
            critic_lr=hps.critic_lr, d_lr=hps.d_lr, beta1=0.9, beta2=0.999, adam_eps=1e-8,
            td_loss_scale=hps.td_loss_1_scale, wd_scale=hps.wd_scale,
            q_actor_loss_scale=hps.q_actor_loss_scale, d_actor_loss_scale=hps.d_actor_loss_scale,
            ent_reg_scale=hps.ent_reg_scale,
            enable_popart=int(bool(hps.rmsify_rets and hps.enable_popart)),
            n_step=int(hps.n) if hps.n_step_returns else 0, td_loss_n_scale=hps.td_loss_n_scale),
            device=self.device)
        L = self.L
        Pac = self.Pa + self.Pc
        self.params, self.targets = L.params[:Pac], L.targets
        self.grads, self.clipped = L.grads[:Pac], L.clipped[:Pac]
        self.params.copy_(torch.as_tensor(init).to(self.device))
        d.rebind(L.seg(L.params, 2), L.seg(L.grads, 2), L.seg(L.m, 2), L.seg(L.v, 2), L.steps[2:3], learner=L)
        a, c = slice(0, self.Pa), slice(self.Pa, self.Pa + self.Pc)
        self.actor = FlatNet(self.scope, 'online_actor', self.da, self.params[a])
        self.critic = FlatNet(self.scope, 'online_critic', self.dc, self.params[c])
        self.targ_actor = FlatNet(self.scope, 'targ_actor', self.da, self.targets[a])
        self.targ_critic = FlatNet(self.scope, 'targ_critic', self.dc, self.targets[c])
        self.actor_grad, self.critic_grad = self.grads[a], self.grads[c]

        # the reward net the actor loss differentiates through (sam_agent.py:88,192): see
        # hps.tie_actor_reward_net
        self.d_desc = d.desc
        if hps.d_actor_loss_scale != 0:
            if hps.tie_actor_reward_net:
                self.d_params = d.params
            else:
                rs2 = np.random.RandomState(hps.seed + 15485863)
                self.d_params = torch.as_tensor(
                    init_flat(self.d_desc, rs2, 'reward', hps.hid_w_init)).to(self.device)
        else:
            self.d_params = None

        L.set_ptr('actor_reward_params', self.d_params)
        if hps.rmsify_obs:
            self.obs_rms = DeviceRunningMeanStd(shape=self.ob_shape, device=self.device)
            L.set_ptr('mu_mean', self.obs_rms.mean); L.set_ptr('mu_std', self.obs_rms.std)
            L.set_ptr('d_mean', d.obs_rms.mean); L.set_ptr('d_std', d.obs_rms.std)

        # return statistics (sam_agent.py:161-172): scalar running mean/std, updated only under PopArt
        self.ret = None
        if hps.rmsify_rets:
            self.ret_rms = DeviceRunningMeanStd(shape=(), device=self.device)
            self.ret = (self.ret_rms.mean, self.ret_rms.std)
            L.set_ptr('ret_mean', self.ret_rms.mean); L.set_ptr('ret_std', self.ret_rms.std)
            L.set_ptr('ret_sums', self.ret_rms.sums)
            self._ret_add2 = torch.zeros(3, dtype=torch.float64, device=self.device)

        if self.param_noise is not None:
            self.setup_param_noise()
        self.setup_replay_buffer()
        actor_opt = MpiAdam(comm, self.actor.flat, hps.clip_norm, hps.actor_lr, name='actor_adam',
                            m=L.seg(L.m, 0), v=L.seg(L.v, 0), t_dev=L.steps[0:1])
        critic_opt = MpiAdam(comm, self.critic.flat, hps.clip_norm, hps.critic_lr, name='critic_adam',
                             m=L.seg(L.m, 1), v=L.seg(L.v, 1), t_dev=L.steps[1:2])
        self.actor_ops = {'names': ['actor_q_loss', 'actor_d_loss', 'actor_total_loss'],
                          'optimizer': actor_opt}
        names = ['critic_td_loss_1'] + (['critic_td_loss_n'] if hps.n_step_returns else []) + \
            ['critic_wd', 'critic_total_loss']   # sam_agent.py:479-517
        self.critic_ops = {'names': names, 'optimizer': critic_opt}
        self.actor_losses = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.critic_losses = torch.zeros(4, dtype=torch.float32, device=self.device)
        B = hps.batch_size
        self.ws = ops.nets_ws(self.da, self.dc, self.d_desc, B, self.device)
        self.ws1 = ops.nets_ws(self.da, self.dc, self.d_desc, 1, self.device)
        self.targ_q = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.targ_q_n = torch.zeros(B, dtype=torch.float32, device=self.device)
        self.td = torch.zeros(B, dtype=torch.float32, device=self.device)
        self._td_host = torch.empty(B, dtype=torch.float32).pin_memory()
        self._idx_host = torch.empty(B, dtype=torch.int64).pin_memory()
        self._idx_dev = torch.empty(B, dtype=torch.int64, device=self.device)

    # ---- set-up pieces -------------------------------------------------------------------------
    def parse_noise_type(self, noise_type):
        """sam_agent.py:238-288 (Box action spaces)."""
        ac_noise = None
        param_noise = None
        for cur in noise_type.split(','):
            cur = cur.strip()
            if cur == 'none':
                pass
            elif 'adaptive-param' in cur:
                _, std = cur.split('_')
                param_noise = AdaptiveParamNoise(initial_std=float(std), delta=float(std))
            elif 'normal' in cur:
                _, std = cur.split('_')
                ac_noise = NormalAcNoise(mu=np.zeros(self.ac_dim), sigma=float(std) * np.ones(self.ac_dim))
            elif 'ou' in cur:
                _, std = cur.split('_')
                ac_noise = OUAcNoise(mu=np.zeros(self.ac_dim), sigma=float(std) * np.ones(self.ac_dim))
            else:
                raise RuntimeError("unknown specified noise type: '{}'".format(cur))
        return param_noise, ac_noise

    def setup_param_noise(self):
        """sam_agent.py:352-404: two perturbed copies of the actor — one acts in the environment
        (pnp), one only measures the action-space effect of the current std (apnp)."""
        self.pn_cur_std = self.param_noise.cur_std
        self.pn_dist = None
        self.pnp_actor = FlatNet(self.scope, 'pnp_actor', self.da, torch.zeros_like(self.actor.flat))
        self.apnp_actor = FlatNet(self.scope, 'apnp_actor', self.da, torch.zeros_like(self.actor.flat))
        self._pn_seed = int(self.hps.seed) * 7919 + 104729 * (self.comm.Get_rank() + 1)
        self._pn_count = 0
        self._pn_dist_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._perturb(self.pnp_actor)
        self._perturb(self.apnp_actor)

    def _perturb(self, net):
        """get_p_actor_updates (sam_agent.py:40-58) with the current std."""
        ops.perturb_params(self.da, self.actor.flat, net.flat, self.param_noise.cur_std, self._pn_seed,
                           self._pn_count)
        self._pn_count += 1

    def setup_replay_buffer(self):
        """sam_agent.py:310-335."""
        kw = dict(limit=self.hps.mem_size, ob_shape=self.ob_shape, ac_shape=self.ac_shape,
                  device=self.device)
        if self.hps.prioritized_replay:
            self.replay_buffer = XP.PrioritizedRB(alpha=self.hps.alpha, beta=self.hps.beta,
                                                  ranked=self.hps.ranked, **kw)
        else:
            self.replay_buffer = XP.RB(**kw)

    def _mean_std(self):
        return (self.obs_rms.mean, self.obs_rms.std) if self.hps.rmsify_obs else (None, None)

    # ---- acting --------------------------------------------------------------------------------
    def predict(self, obs, apply_noise, compute_q):
        """sam_agent.py:592-619."""
        o = torch.as_tensor(np.asarray(obs, dtype=np.float32)[None]).to(self.device)
        mean, std = self._mean_std()
        ac, q = ops.act_q(self.da, self.actor.flat, self.dc, self.critic.flat, o, mean, std,
                          self.ac_high, self.ws1, want_q=compute_q, ret=self.ret)
        if apply_noise and self.param_noise is not None:
            # p_act / p_act_q (sam_agent.py:373-378): the action of the parameter-noise-perturbed actor;
            # the Q value stays critic_pred_w_actor, i.e. Q(s, UNperturbed actor(s)) — as the reference
            ac, _ = ops.act_q(self.da, self.pnp_actor.flat, self.dc, self.critic.flat, o, mean, std,
                              self.ac_high, self.ws1, want_q=False, ret=self.ret)
        ac = ac.cpu().numpy().flatten()
        if apply_noise and self.ac_noise is not None:
            noise = self.ac_noise.generate()
            assert noise.shape == ac.shape
            ac += noise
        return ac, (float(q.item()) if compute_q else None)

    def store_transition(self, ob0, ac, rew, ob1, done1):
        """sam_agent.py:621-629."""
        rew *= self.hps.reward_scale
        self.replay_buffer.append(ob0, ac, rew, ob1, done1)
        if self.hps.rmsify_obs:
            self.obs_rms.update(np.array([ob0]), self.comm)

    # ---- the update ----------------------------------------------------------------------------
    def sample_device(self):
        """Replay batch in HBM.  Uniform: indices from np.random.randint (memory.py:90), one H2D
        copy, gather kernel.  PER: uniforms from random.random() (memory.py:278), tree kernels."""
        B, rb = self.hps.batch_size, self.replay_buffer
        if self.hps.prioritized_replay:
            return rb.sample_device(B)
        idxs = np.random.randint(low=0, high=rb.num_entries, size=B)
        self._idx_dev.copy_(torch.from_numpy(idxs))  # pageable, synchronous: safe to redraw at once
        batch = rb.batchify_device(self._idx_dev)
        batch['idxs'] = self._idx_dev
        return batch

    def lookahead_device(self, batch):
        """memory.py:113-160 on the device: the n-step companion of a sampled batch."""
        rb = self.replay_buffer
        end, rews, dones, td_len = ops.nstep_lookahead(rb.store, rb.start, rb.num_entries, batch['idxs'],
                                                       self.hps.n, self.hps.gamma)
        obs1 = rb.batchify_device(end, want=('obs1',))['obs1']
        return dict(obs1=obs1, rews=rews, dones1=dones, td_len=td_len)

    def grads_device(self, batch):
        """TD target(s) on the target nets, [ret_rms update + PopArt], then actor and critic losses +
        unclipped flat gradients on the pre-update parameters (sam_agent.py:651-687 without the four
        repeated sess.runs)."""
        mean, std = self._mean_std()
        hp = self.hps
        ops.targ_q(self.da, self.targ_actor.flat, self.dc, self.targ_critic.flat, batch['obs1'],
                   batch['rews'], batch['dones1'], None, mean, std, hp.gamma, self.ac_high, self.ws,
                   out=self.targ_q, ret=self.ret)
        tcns = None
        if hp.n_step_returns:  # sam_agent.py:656-660
            bn = self.lookahead_device(batch)
            ops.targ_q(self.da, self.targ_actor.flat, self.dc, self.targ_critic.flat, bn['obs1'], bn['rews'],
                       bn['dones1'], bn['td_len'], mean, std, hp.gamma, self.ac_high, self.ws,
                       out=self.targ_q_n, ret=self.ret)
            tcns = self.targ_q_n
        if hp.rmsify_rets and hp.enable_popart:  # sam_agent.py:662-672
            r = self.ret_rms
            ops.rms_batch_sums(self.targ_q.view(-1, 1), None, r.addvec)
            self.comm.allreduce_(r.addvec)          # mpi_running_mean_std.py:55
            if tcns is not None:
                ops.rms_batch_sums(tcns.view(-1, 1), None, self._ret_add2)
                self.comm.allreduce_(self._ret_add2)
            ops.ret_update_popart(r.sums, r.addvec, self._ret_add2 if tcns is not None else None, r.mean, r.std,
                                  True, self.dc, self.critic.flat, self.targ_critic.flat)
        ops.actor_grad(self.da, self.actor.flat, self.dc, self.critic.flat, self.d_desc, self.d_params,
                       batch['obs0'], mean, std, self.ac_high, hp.q_actor_loss_scale,
                       hp.d_actor_loss_scale, self.ws, losses=self.actor_losses, grad=self.actor_grad,
                       ret=self.ret)
        ops.critic_grad(self.dc, self.critic.flat, batch['obs0'], batch['acs'],
                        batch.get('iws32') if hp.prioritized_replay else None, self.targ_q, mean, std,
                        hp.td_loss_1_scale, hp.wd_scale, self.ws, losses=self.critic_losses,
                        grad=self.critic_grad, td=self.td, ret=self.ret, tcns=tcns,
                        tdn_scale=hp.td_loss_n_scale)

    def apply_device(self, fuse_polyak=False):
        """MpiAdam on both nets: ONE all-reduce over the [actor|critic] bucket (mpi_adam.py:27),
        then per-net clip + Adam (sam_agent.py:690-691)."""
        self.comm.allreduce_(self.grads)
        tau = self.hps.polyak
        self.actor_ops['optimizer'].update(self.actor_grad, reduce=False,
                                           targ=self.targ_actor.flat if fuse_polyak else None, tau=tau)
        self.critic_ops['optimizer'].update(self.critic_grad, reduce=False,
                                            targ=self.targ_critic.flat if fuse_polyak else None, tau=tau)

    def refresh_priorities(self, batch):
        """sam_agent.py:693-703: |td| + 1e-6 with the POST-update critic, then the tree update.
        The float32 sum matches numpy's (float32 array + python float stays float32)."""
        mean, std = self._mean_std()
        ops.td_errors(self.dc, self.critic.flat, batch['obs0'], batch['acs'], self.targ_q, mean, std,
                      self.ws, out=self.td, ret=self.ret)
        self._td_host.copy_(self.td, non_blocking=True)
        self._idx_host.copy_(batch['idxs'], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        new_priorities = np.abs(self._td_host.numpy()) + np.float32(1e-6)
        self.replay_buffer.update_priorities(self._idx_host.numpy(), new_priorities)

    def train_device(self, fuse_polyak=False, want_clipped=True):
        batch = self.sample_device()
        self.grads_device(batch)
        if want_clipped:  # flatgrad(loss, vars, clip_norm): the logged, locally clipped gradients
            a = slice(0, self.Pa); c = slice(self.Pa, self.Pa + self.Pc)
            ops.clip_copy(self.actor_grad, self.hps.clip_norm, self.actor_ops['optimizer'].partials,
                          out=self.clipped[a])
            ops.clip_copy(self.critic_grad, self.hps.clip_norm, self.critic_ops['optimizer'].partials,
                          out=self.clipped[c])
        self.apply_device(fuse_polyak)
        if self.hps.prioritized_replay:
            self.refresh_priorities(batch)
        return batch

    # ---- the update behind the reference's own call: the agent roles of the whole-step kernels --------------
    def _fast(self):
        """One-time set-up of train(): staging, the batch / output structs, a twin of the learner struct whose
        `clipped` vector lives in pinned host memory (the optimiser pass stores the logged gradients there)."""
        if getattr(self, '_f', None) is not None:
            return self._f
        L, hp, rb = self.L, self.hps, self.replay_buffer
        B, dev = hp.batch_size, self.device
        Pac = self.Pa + self.Pc
        f = types.SimpleNamespace()
        f.lib = _lib.lib()
        # pinned results: [actor_losses 3 | critic_losses 4 (td1, wd, total, tdn) | pad | clipped actor | clipped critic]
        f.pin = torch.zeros(8 + Pac, dtype=torch.float32).pin_memory()
        f.res = f.pin.numpy()
        f.c = _lib.Learner.from_buffer_copy(L.c)
        f.c.clipped = f.pin.data_ptr() + 32
        f.out = ops.make_step_out(B, dev)
        f.out.actor_losses = f.pin.data_ptr()
        f.out.critic_losses = f.pin.data_ptr() + 12
        f.idx_pin = torch.zeros(B, dtype=torch.int64).pin_memory()
        f.idx_np = f.idx_pin.numpy()
        f.idx_dev = torch.zeros(B, dtype=torch.int64, device=dev)
        st = rb.store
        kw = {}
        f.per = bool(hp.prioritized_replay)
        # uniform replay on the whole-network path: the kernels read the B index words straight from the pinned
        # staging (each once per tile) — no upload in front of the launch
        f.idx_zero_copy = bool(not f.per and L.path(B) == 1)
        idx_words = f.idx_pin if f.idx_zero_copy else f.idx_dev
        if f.per:
            f.u_pin = torch.zeros(B, dtype=torch.float64).pin_memory()
            f.u_np = f.u_pin.numpy()
            f.u_dev = torch.zeros(B, dtype=torch.float64, device=dev)
            f.iws = torch.zeros(B, dtype=torch.float64, device=dev)
            f.iws32 = torch.zeros(B, dtype=torch.float32, device=dev)
            f.td_post = torch.zeros(B, dtype=torch.float32, device=dev)
            # the refreshed TD errors and the sampled indices land in pinned host memory (same launch)
            f.td_pin = torch.zeros(B, dtype=torch.float32).pin_memory()
            f.pidx_pin = torch.zeros(B, dtype=torch.int64).pin_memory()
            f.td_np, f.pidx_np = f.td_pin.numpy(), f.pidx_pin.numpy()
            f.leaf_pin = torch.zeros(B, dtype=torch.float64).pin_memory()
            f.leaf_np = f.leaf_pin.numpy()
            f.leaf_dev = torch.zeros(B, dtype=torch.float64, device=dev)
            kw['iws'] = f.iws32
            vp = lambda t: C.c_void_p(t.data_ptr())
            f.tree_s, f.tree_m, f.u_p, f.idx_p = vp(rb.trees.sum), vp(rb.trees.min), vp(f.u_pin), vp(f.idx_dev)
            f.iws_p, f.iws32_p = vp(f.iws), vp(f.iws32)
        self.world = self.comm.Get_size()
        f.popart_split = bool(self.world > 1 and hp.rmsify_rets and hp.enable_popart)
        if f.popart_split:
            f.ret_addvec = torch.zeros(6, dtype=torch.float64, device=dev)
            kw['ret_addvec'] = f.ret_addvec
        # rows are read in place through the index words; the discriminator's inputs are never read by a call
        # that leaves its role out (they only have to be non-null)
        f.batch = ops.make_batch(B, p_obs=st['obs0'], p_acs=st['acs'], e_obs=st['obs0'], e_acs=st['acs'],
                                 obs0=st['obs0'], acs=st['acs'], rews=st['rews'].view(-1),
                                 dones1=st['dones1'].view(-1), obs1=st['obs1'], a_idx=idx_words, p_idx=idx_words,
                                 e_start=idx_words[:1], ring_start=rb.start, ring_len=rb.limit, rng_seed=0,
                                 label_smoothing=0, num_entries=rb.num_entries, host_ok=f.idx_zero_copy, **kw)
        f.ws = L.ws(B)
        f.ws_p, f.ws_n = C.c_void_p(f.ws.data_ptr()), f.ws.numel()
        self._f = f
        return f

    def train(self):
        """sam_agent.py:631-711."""
        if _PER_ROLE:
            return self.train_per_role()
        f = self._fast()
        lib, L, rb, hp = f.lib, self.L, self.replay_buffer, self.hps
        B = hp.batch_size
        b = f.batch
        b.ring_start, b.num_entries = rb.start, rb.num_entries   # the ring as this call finds it
        st = _lib.stream_ptr()
        chk = _lib.check
        if f.per:   # memory.py:248-309: stratified draws with `random`, descent and weights on the device trees
            assert rb.num_entries
            f.u_np[:] = rb.draw_uniforms(B)
            # (the sampling kernel reads its B uniforms once: straight from the pinned staging)
            chk(lib.samx_per_sample(f.tree_s, f.tree_m, rb.st_cap, rb.num_entries, f.u_p, B, rb.beta, f.idx_p,
                                    f.iws_p, f.iws32_p, st))
        else:       # memory.py:90
            f.idx_np[:] = np.random.randint(low=0, high=rb.num_entries, size=B)
            if not f.idx_zero_copy:
                f.idx_dev.copy_(f.idx_pin, non_blocking=True)
        cL, cb, co = C.byref(f.c), C.byref(b), C.byref(f.out)
        if f.popart_split:   # ret_rms.update all-reduces the targets' sums before PopArt (sam_agent.py:662-672)
            chk(lib.samx_step_grads_part(cL, cb, co, f.ws_p, f.ws_n, 1 | 4 | 16, st))
            self.comm.allreduce_(f.ret_addvec)
            chk(lib.samx_step_grads_part(cL, cb, co, f.ws_p, f.ws_n, 3 | 4 | 32, st))
        else:
            chk(lib.samx_step_grads_part(cL, cb, co, f.ws_p, f.ws_n, 3 | 4, st))
        # flatgrad(..., clip_norm) copies for the caller (tf_util.py:137-151), then MpiAdam: mean over the workers,
        # clip, Adam (mpi_adam.py:18-36) on the actor and critic segments; no Polyak (update_target_net is its own call)
        if self.world == 1:
            chk(lib.samx_step_apply(cL, B, 3 | 8 | 32, 1.0, f.ws_p, f.ws_n, st))
        else:
            chk(lib.samx_step_apply(cL, B, 1 | 8, 1.0, f.ws_p, f.ws_n, st))
            self.comm.allreduce_(self.grads)
            chk(lib.samx_step_apply(cL, B, 2 | 4 | 8 | 32, 1.0 / self.world, f.ws_p, f.ws_n, st))
        if f.per:   # TD errors with the POST-update critic (sam_agent.py:693-701), copies to pinned host memory
            mean, std = self._mean_std()
            ops.td_errors_rows(self.dc, self.critic.flat, rb.store['obs0'], rb.store['acs'], f.idx_dev, rb.start,
                               f.out.t['targ_q'], mean, std, f.td_post, ret=self.ret, td_host=f.td_pin,
                               idx_host=f.pidx_pin)
        chk(lib.samx_stream_synchronize(st))   # the one host wait of the call
        r, Pa = f.res, self.Pa
        cl = r[3:7]
        out = {'actor_grads': r[8:8 + Pa].copy(),
               'actor_losses': [float(r[0]), float(r[1]), float(r[2])],
               'critic_grads': r[8 + Pa:].copy(),
               'critic_losses': ([float(cl[0]), float(cl[3]), float(cl[1]), float(cl[2])] if hp.n_step_returns
                                 else [float(cl[0]), float(cl[1]), float(cl[2])])}
        if f.per:   # sam_agent.py:702-703 -> memory.py:344-379: asserts + libm pow on the host, trees on the device
            rb.leaves_from_td(f.pidx_np, f.td_np, f.leaf_np)
            f.leaf_dev.copy_(f.leaf_pin, non_blocking=True)
            rb.trees.set(f.idx_dev, f.leaf_dev)   # (idx_dev still holds this call's sample)
        return out

    def train_per_role(self):
        """train() through the per-role entry points (one launch chain per network pass)."""
        self.train_device()
        cl = self.clipped.cpu().numpy()
        return {'actor_grads': cl[:self.Pa].copy(),
                'actor_losses': [float(x) for x in self.actor_losses.cpu().numpy()],
                'critic_grads': cl[self.Pa:].copy(),
                'critic_losses': self._critic_losses_host()}

    def _critic_losses_host(self):
        """In the reference's order (sam_agent.py:479-517): td_loss_1, [td_loss_n], wd, total."""
        l = [float(x) for x in self.critic_losses.cpu().numpy()]
        return [l[0], l[3], l[1], l[2]] if self.hps.n_step_returns else l[:3]

    # ---- housekeeping --------------------------------------------------------------------------
    def sync_from_root(self):
        """sam_agent.py:713-717 (one broadcast of the [actor|critic] bucket)."""
        self.comm.Barrier()
        self.comm.bcast_(self.params, root=0)

    def initialize_target_nets(self):
        ops.polyak(self.targets, self.params, 1.0)

    def update_target_net(self):
        ops.polyak(self.targets, self.params, self.hps.polyak)

    def adapt_param_noise(self, comm):
        """sam_agent.py:727-740."""
        from .comm import mpi_mean_like
        batch = self.replay_buffer.sample(batch_size=self.hps.batch_size)
        obs0 = torch.as_tensor(np.ascontiguousarray(batch['obs0'], dtype=np.float32)).to(self.device)
        self._perturb(self.apnp_actor)
        mean, std = self._mean_std()
        B = obs0.shape[0]
        ws = self.ws if B == self.hps.batch_size else ops.nets_ws(self.da, self.dc, self.d_desc, B, self.device)
        a0, _ = ops.act_q(self.da, self.actor.flat, self.dc, self.critic.flat, obs0, mean, std, self.ac_high,
                          ws, want_q=False)
        a0 = a0.clone()
        a1, _ = ops.act_q(self.da, self.apnp_actor.flat, self.dc, self.critic.flat, obs0, mean, std,
                          self.ac_high, ws, want_q=False)
        dist = float(ops.rms_distance(a0, a1, self._pn_dist_dev).item())
        self.pn_dist = float(mpi_mean_like(np.array(dist), comm))
        self.param_noise.adapt_std(self.pn_dist)
        self.pn_cur_std = self.param_noise.cur_std

    def reset_noise(self):
        """sam_agent.py:741-749."""
        if self.param_noise is not None:
            self._perturb(self.pnp_actor)
        if self.ac_noise is not None:
            self.ac_noise.reset()
