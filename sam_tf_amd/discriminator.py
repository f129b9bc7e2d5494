"""Discriminator with the reference's surface (imitation_algorithms/discriminator.py:18-236) on
libsamx.  The TF-random tensors of the graph (GP epsilons, smoothed labels, lines 101-109,130-133)
are drawn on the device by a seeded torch generator, or injected by the caller (parity tests).
"""
import numpy as np
import torch

from . import ops
from .hps import check_supported
from .mpi_adam import MpiAdam
from .networks import FlatNet, init_flat
from .running_mean_std import DeviceRunningMeanStd

LOSS_NAMES = ['d_policy_loss', 'd_expert_loss', 'd_ent_mean', 'd_ent_loss', 'd_policy_acc',
              'd_expert_acc', 'd_grad_pen', 'd_total_loss']


class Discriminator(object):

    def __init__(self, name, env, hps, comm, device='cuda'):
        check_supported(hps)
        assert hps.ent_reg_scale >= 0, "'ent_reg_scale' must be non-negative"
        self.name, self.scope, self.env, self.hps, self.comm = name, name, env, hps, comm
        self.device = torch.device(device)
        self.ob_shape = env.observation_space.shape
        self.ac_shape = env.action_space.shape
        self.ob_dim, self.ac_dim = int(self.ob_shape[0]), int(self.ac_shape[0])
        # feed-dict keys (discriminator.py:52-59)
        self.p_obs, self.p_acs, self.e_obs, self.e_acs = 'p_obs', 'p_acs', 'e_obs', 'e_acs'
        self.desc = ops.mlp_desc(self.ob_dim + self.ac_dim, hps.hid_widths[0], hps.hid_widths[1], 1,
                                 hps.with_layernorm)
        rs = np.random.RandomState(hps.seed + 7919)
        flat = torch.as_tensor(init_flat(self.desc, rs, 'reward', hps.hid_w_init)).to(self.device)
        self.reward_nn = FlatNet(self.scope, 'sr', self.desc, flat)
        if hps.rmsify_obs:
            self.obs_rms = DeviceRunningMeanStd(shape=self.ob_shape, device=self.device)
        self.optimizer = MpiAdam(comm, flat, hps.clip_norm, hps.d_lr, name='d_adam')
        self.names = list(LOSS_NAMES)
        self.grad = torch.zeros_like(flat)
        self.losses = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(hps.seed) + 104729 * (comm.Get_rank() + 1))
        self._ws = {}

    def rebind(self, flat, grad, m, v, t_dev):
        """Move parameters / gradient / Adam state into views of a learner-wide flat state
        (ops.LearnerState, layout [actor | critic | reward-net]) so one all-reduce and one optimiser
        launch cover the whole step.  Values are carried over."""
        flat.copy_(self.reward_nn.flat); m.copy_(self.optimizer.m); v.copy_(self.optimizer.v)
        t_dev.copy_(self.optimizer.t_dev)
        self.reward_nn = FlatNet(self.scope, 'sr', self.desc, flat)
        self.grad = grad
        self.optimizer = MpiAdam(self.comm, flat, self.hps.clip_norm, self.hps.d_lr, name='d_adam', m=m,
                                 v=v, t_dev=t_dev)

    # ---- helpers
    @property
    def params(self):
        return self.reward_nn.flat

    @property
    def trainable_vars(self):
        return self.reward_nn.trainable_vars

    vars = trainable_vars

    def _mean_std(self):
        return (self.obs_rms.mean, self.obs_rms.std) if self.hps.rmsify_obs else (None, None)

    def ws(self, B):
        if B not in self._ws:
            self._ws[B] = (ops.disc_ws(self.desc, B, self.device),
                           ops.nets_ws(ops.mlp_desc(self.ob_dim, 1, 1, self.ac_dim), self.desc, None, B,
                                       self.device))
        return self._ws[B]

    def draw_randoms(self, B):
        """eps ~ U[0,1) per element; real labels 1.2 - 0.5u, fake 0 or 0.3 - 0.3u
        (tf.random_uniform with minval > maxval, discriminator.py:99-109)."""
        r = lambda *s: torch.rand(*s, generator=self.gen, device=self.device, dtype=torch.float32)
        eps_ob, eps_ac = r(B, self.ob_dim), r(B, self.ac_dim)
        real = fake = None
        if self.hps.label_smoothing:
            real = 1.2 - 0.5 * r(B)
            if not self.hps.one_sided_label_smoothing:
                fake = 0.3 - 0.3 * r(B)
        return eps_ob, eps_ac, real, fake

    # ---- device-resident hot path
    def grads_device(self, p_obs, p_acs, e_obs, e_acs, randoms=None):
        """One pass: the 8 losses and the unclipped flat gradient, both left in HBM."""
        B = p_obs.shape[0]
        eps_ob, eps_ac, real, fake = self.draw_randoms(B) if randoms is None else randoms
        mean, std = self._mean_std()
        ops.disc_grad(self.desc, self.params, p_obs, p_acs, e_obs, e_acs, eps_ob, eps_ac, real, fake,
                      mean, std, self.hps.ent_reg_scale, self.ws(B)[0], losses=self.losses,
                      grad=self.grad)
        return self.losses, self.grad

    def update_device(self, p_obs, p_acs, e_obs, e_acs, randoms=None, clipped_out=None):
        """get_grads + get_losses + optimize of sam.py:398-402 in ONE pass over one random draw."""
        self.grads_device(p_obs, p_acs, e_obs, e_acs, randoms)
        if clipped_out is not None:  # the locally clipped gradient the reference logs
            ops.clip_copy(self.grad, self.hps.clip_norm, self.optimizer.partials, out=clipped_out)
        self.optimizer.update(self.grad)
        return self.losses

    def get_reward_device(self, obs, acs, out=None):
        B = obs.shape[0]
        mean, std = self._mean_std()
        return ops.reward(self.desc, self.params, obs, acs, mean, std, self.hps.non_satur_grad,
                          self.ws(B)[1], out=out)

    # ---- drop-in (numpy / feed-dict) surface
    def _feeds(self, feeds):
        t = lambda k: torch.as_tensor(np.ascontiguousarray(feeds[k], dtype=np.float32)).to(self.device)
        return t(self.p_obs), t(self.p_acs), t(self.e_obs), t(self.e_acs)

    def get_grads(self, feeds):
        self.grads_device(*self._feeds(feeds))
        return ops.clip_copy(self.grad, self.hps.clip_norm, self.optimizer.partials).cpu().numpy()

    def get_losses(self, feeds):
        self.grads_device(*self._feeds(feeds))
        return [float(x) for x in self.losses.cpu().numpy()]

    def optimize(self, feeds):
        self.grads_device(*self._feeds(feeds))
        self.optimizer.update(self.grad)

    def get_reward(self, ob, ac):
        """discriminator.py:199-224: one (ob, ac) pair -> python float."""
        o = torch.as_tensor(np.asarray(ob, dtype=np.float32)[None]).to(self.device)
        a = torch.as_tensor(np.asarray(ac, dtype=np.float32)[None]).to(self.device)
        return float(self.get_reward_device(o, a).item())

    def sync_from_root(self):
        self.optimizer.sync_from_root(self.trainable_vars)
