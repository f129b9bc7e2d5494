"""Discriminator with the reference's surface (imitation_algorithms/discriminator.py:18-236) on
libsamx.

Once an agent has taken the discriminator's vectors into its learner-wide state (rebind), the feed-dict calls
of sam.py:392-402 — get_grads / get_losses / optimize — run the DISCRIMINATOR ROLE of the whole-step entry
points (samx_step_grads_part without the agent roles [+ samx_step_apply on the reward-net segment]): the fused /
role-fused / layer-wise kernels the update-step driver uses.  The four feed arrays travel in ONE packed upload;
what a call returns is ONE host synchronisation (the kernels store clipped gradient and losses in pinned host
memory).  The TF-random tensors of the graph (GP epsilons, smoothed labels, lines 101-109,130-133) are drawn in
the kernels from Philox keyed by (seed, call number, update count); every call draws afresh, as every sess.run of
the reference does; ops.disc_randoms(d.last_seed, <update count before the call>, ...) reproduces a call's tensors.
A discriminator without an agent keeps the per-role entry point (samx_disc_grad) with a seeded torch generator,
or randoms injected by the caller (parity tests).
"""
import ctypes as C
import os
import types

import numpy as np
import torch

from . import _lib
from . import ops
from .hps import check_supported
from .mpi_adam import MpiAdam
from .networks import FlatNet, init_flat
from .running_mean_std import DeviceRunningMeanStd

_PER_ROLE = os.environ.get('SAMX_OBJECT_API', '') == 'per_role'

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
        self._L = None
        self._fp = {}
        self._calls = 0
        self.last_seed = None
        self._seed0 = (int(hps.seed) * 2654435761 + 40503 * (comm.Get_rank() + 1)) & 0xFFFFFFFF

    def rebind(self, flat, grad, m, v, t_dev, learner=None):
        """Move parameters / gradient / Adam state into views of a learner-wide flat state
        (ops.LearnerState, layout [actor | critic | reward-net]) so one all-reduce and one optimiser
        launch cover the whole step.  Values are carried over."""
        flat.copy_(self.reward_nn.flat); m.copy_(self.optimizer.m); v.copy_(self.optimizer.v)
        t_dev.copy_(self.optimizer.t_dev)
        self.reward_nn = FlatNet(self.scope, 'sr', self.desc, flat)
        self.grad = grad
        self.optimizer = MpiAdam(self.comm, flat, self.hps.clip_norm, self.hps.d_lr, name='d_adam', m=m,
                                 v=v, t_dev=t_dev)
        self._L = learner
        self._fp = {}

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

    # ---- the discriminator role of the whole-step kernels behind the feed-dict calls
    def _fast(self, B):
        if B in self._fp:
            return self._fp[B]
        L, dev = self._L, self.device
        ob, ac = self.ob_dim, self.ac_dim
        Pd = self.params.numel()
        f = types.SimpleNamespace()
        f.lib = _lib.lib()
        # the four feed arrays, packed: one pinned staging buffer, one upload
        W = B * (2 * ob + 2 * ac)
        f.feed_pin = torch.zeros(W, dtype=torch.float32).pin_memory()
        f.feed_dev = torch.zeros(W, dtype=torch.float32, device=dev)
        n = f.feed_pin.numpy()
        o1, o2, o3 = B * ob, B * (ob + ac), B * (2 * ob + ac)
        f.views = {self.p_obs: n[:o1].reshape(B, ob), self.p_acs: n[o1:o2].reshape(B, ac),
                   self.e_obs: n[o2:o3].reshape(B, ob), self.e_acs: n[o3:].reshape(B, ac)}
        # whole-network kernels (64x64 nets) read every feed row once or twice: small batches are read by the
        # kernels straight from the pinned staging (no upload in front of the launch); every other path re-reads its
        # inputs tile by tile and gets a device copy
        f.zero_copy = bool(L.path(B) == 1 and 4 * W <= (256 << 10))
        dv = f.feed_pin if f.zero_copy else f.feed_dev
        p_obs, p_acs = dv[:o1].view(B, ob), dv[o1:o2].view(B, ac)
        e_obs, e_acs = dv[o2:o3].view(B, ob), dv[o3:].view(B, ac)
        f.event = torch.cuda.Event()
        f.used = False
        # pinned results: [d_losses 8 | clipped reward-net gradient]
        f.pin = torch.zeros(8 + Pd, dtype=torch.float32).pin_memory()
        f.res = f.pin.numpy()
        f.c = _lib.Learner.from_buffer_copy(L.c)
        # (the optimiser pass writes segment k at clipped + offset[k]: the reward-net segment then starts at res[8])
        f.c.clipped = f.pin.data_ptr() + 32 - 4 * (L.Pa + L.Pc)
        f.out = ops.make_step_out(B, dev)
        f.out.d_losses = f.pin.data_ptr()
        hp = self.hps
        label_mode = 0 if not hp.label_smoothing else (1 if hp.one_sided_label_smoothing else 2)
        # the agent's inputs are never read by a call that leaves its roles out (they only have to be non-null)
        f.batch = ops.make_batch(B, p_obs=p_obs, p_acs=p_acs, e_obs=e_obs, e_acs=e_acs, obs0=p_obs, acs=p_acs,
                                 rews=dv[:B], dones1=dv[:B], obs1=p_obs, rng_seed=0, label_smoothing=label_mode,
                                 host_ok=f.zero_copy)
        f.ws = L.ws(B)
        f.ws_p, f.ws_n = C.c_void_p(f.ws.data_ptr()), f.ws.numel()
        self.world = self.comm.Get_size()
        self._fp[B] = f
        return f

    def _role(self, feeds, tick):
        """Upload the feeds, run the discriminator role (losses + unclipped gradient into the learner's flat
        gradient vector); `tick`: an update follows (the reward net's update count advances)."""
        B = len(feeds[self.p_obs])
        f = self._fast(B)
        if f.used:
            f.event.synchronize()   # whatever read the staging buffer last (upload / kernels) has finished
            f.used = False
        for k, v in f.views.items():
            np.copyto(v, feeds[k], casting='same_kind')
        if not f.zero_copy:
            f.feed_dev.copy_(f.feed_pin, non_blocking=True)
            f.event.record()
            f.used = True
        # fresh draws per call (every sess.run of the reference re-draws its tf.random_uniform tensors)
        self._calls += 1
        f.batch.rng_seed = (self._seed0 + 7919 * self._calls) & 0xFFFFFFFFFFFF
        self.last_seed = int(f.batch.rng_seed)
        st = _lib.stream_ptr()
        _lib.check(f.lib.samx_step_grads_part(C.byref(f.c), C.byref(f.batch), C.byref(f.out), f.ws_p, f.ws_n,
                                              3 | 8 | (0 if tick else 64), st))
        return f, B, st

    def _use_fast(self):
        return self._L is not None and not _PER_ROLE

    def get_grads(self, feeds):
        """discriminator.py:161,173: flatgrad(d_loss, vars, clip_norm) -> numpy."""
        if not self._use_fast():
            self.grads_device(*self._feeds(feeds))
            return ops.clip_copy(self.grad, self.hps.clip_norm, self.optimizer.partials).cpu().numpy()
        f, B, st = self._role(feeds, tick=False)
        _lib.check(f.lib.samx_step_apply(C.byref(f.c), B, 1 | 16, 1.0, f.ws_p, f.ws_n, st))
        _lib.check(f.lib.samx_stream_synchronize(st))
        return f.res[8:].copy()

    def get_losses(self, feeds):
        """discriminator.py:145-158,172."""
        if not self._use_fast():
            self.grads_device(*self._feeds(feeds))
            return [float(x) for x in self.losses.cpu().numpy()]
        f, B, st = self._role(feeds, tick=False)
        _lib.check(f.lib.samx_stream_synchronize(st))
        return f.res[:8].tolist()

    def optimize(self, feeds):
        """discriminator.py:164-168,174: MpiAdam — mean over the workers, clip, Adam."""
        if not self._use_fast():
            self.grads_device(*self._feeds(feeds))
            self.optimizer.update(self.grad)
            return
        f, B, st = self._role(feeds, tick=True)
        if self.world == 1:
            _lib.check(f.lib.samx_step_apply(C.byref(f.c), B, 2 | 16, 1.0, f.ws_p, f.ws_n, st))
        else:
            self.comm.allreduce_(self.grad)
            _lib.check(f.lib.samx_step_apply(C.byref(f.c), B, 2 | 4 | 16, 1.0 / self.world, f.ws_p, f.ws_n, st))
        if f.zero_copy:   # the call returns without a host wait: its kernels may still be reading the staging
            f.event.record()
            f.used = True

    def get_reward(self, ob, ac):
        """discriminator.py:199-224: one (ob, ac) pair -> python float."""
        o = torch.as_tensor(np.asarray(ob, dtype=np.float32)[None]).to(self.device)
        a = torch.as_tensor(np.asarray(ac, dtype=np.float32)[None]).to(self.device)
        return float(self.get_reward_device(o, a).item())

    def sync_from_root(self):
        self.optimizer.sync_from_root(self.trainable_vars)
