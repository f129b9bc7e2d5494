"""Host driver for one "SAM update step" — the body of the reference's training loop
(imitation_algorithms/sam.py:323-433) with d_steps = g_steps = 1:

    discriminator update : uniform replay pairs + expert pairs -> d.obs_rms.update -> D loss/grad
                           (+GP) -> all-reduce -> clip -> Adam                 (sam.py:369-406)
    agent update         : replay sample -> TD target -> actor & critic loss/grad -> all-reduce ->
                           clip -> Adam x2 -> [PER priority refresh]           (sam.py:421 -> train())
    target update        : Polyak                                              (sam.py:433)

Everything stays in HBM.  The device work is `samx_step_grads` + `samx_step_apply` (include/samx.h)
around two row gathers and the running-mean kernels; the discriminator's and the agent's gradients
do not depend on each other's fresh parameters, so they are computed side by side and ONE flat
[actor | critic | reward-net] all-reduce serves the step (plus the 2*ob+1 doubles of the obs sums).
With `use_graph` each stretch of kernels between two collectives is captured once into a HIP graph
(torch.cuda.CUDAGraph on the stream the C ABI is handed) and replayed.  Host work per step: the
reference's RNG draws (np.random.randint / random.random, same generators, same order), two small
pinned H2D copies, the graph launches, and for N > 1 the RCCL calls.  Data-parallel ranks each own a
replay shard and a minibatch; only gradients and running-mean sums are exchanged.
"""
import random as _random

import numpy as np
import torch

from . import ops


class _Slot(object):
    """Pinned staging for one step's host-drawn numbers; an event guards its reuse."""

    def __init__(self, B):
        self.idx_p = torch.empty(B, dtype=torch.int64).pin_memory()
        self.idx_a = torch.empty(B, dtype=torch.int64).pin_memory()
        self.idx_e = torch.empty(B, dtype=torch.int64).pin_memory()
        self.u = torch.empty(B, dtype=torch.float64).pin_memory()
        self.event = None


class UpdateStepDriver(object):

    def __init__(self, mu, d, expert_pool, use_graph=True, n_slots=4):
        self.mu, self.d, self.pool = mu, d, expert_pool
        self.L = mu.L
        self.comm = mu.comm
        self.world = self.comm.Get_size()
        self.B = B = mu.hps.batch_size
        dev = mu.device
        self.dev = dev
        self.per = bool(mu.hps.prioritized_replay)
        self.use_graph = bool(use_graph)
        ob, ac = mu.ob_dim, mu.ac_dim
        f32 = dict(dtype=torch.float32, device=dev)
        # static device buffers (their addresses are baked into the captured graphs)
        self.idx_p = torch.zeros(B, dtype=torch.int64, device=dev)
        self.idx_a = torch.zeros(B, dtype=torch.int64, device=dev)
        self.idx_e = torch.zeros(B, dtype=torch.int64, device=dev)
        self.u = torch.zeros(B, dtype=torch.float64, device=dev)
        self.iws = torch.zeros(B, dtype=torch.float64, device=dev)
        self.iws32 = torch.zeros(B, **f32)
        self.pb = dict(obs0=torch.zeros(B, ob, **f32), acs=torch.zeros(B, ac, **f32))
        self.eb = dict(obs0=torch.zeros(B, ob, **f32), acs=torch.zeros(B, ac, **f32))
        self.ab = {k: torch.zeros(B, w, **f32) for k, w in
                   (('obs0', ob), ('acs', ac), ('rews', 1), ('dones1', 1), ('obs1', ob))}
        self.eps_ob = torch.zeros(B, ob, **f32)
        self.eps_ac = torch.zeros(B, ac, **f32)
        self.real = torch.zeros(B, **f32)
        self.fake = torch.zeros(B, **f32)
        hp = d.hps
        self.batch = ops.make_batch(
            B, p_obs=self.pb['obs0'], p_acs=self.pb['acs'], e_obs=self.eb['obs0'], e_acs=self.eb['acs'],
            eps_ob=self.eps_ob, eps_ac=self.eps_ac,
            real_labels=self.real if hp.label_smoothing else None,
            fake_labels=self.fake if (hp.label_smoothing and not hp.one_sided_label_smoothing) else None,
            obs0=self.ab['obs0'], acs=self.ab['acs'], rews=self.ab['rews'].view(B),
            dones1=self.ab['dones1'].view(B), obs1=self.ab['obs1'],
            iws=self.iws32 if self.per else None)
        self.out = ops.make_step_out(B, dev)
        self.td_post = torch.zeros(B, **f32)
        self._td_host = torch.empty(B, dtype=torch.float32).pin_memory()
        self._idx_host = torch.empty(B, dtype=torch.int64).pin_memory()
        self.slots = [_Slot(B) for _ in range(n_slots)]
        self.step_no = 0
        self.plan = None
        self.fused = self.L.fused(B)

    # ------------------------------------------------------------------ device segments
    def _seg_inputs(self):
        mu, d = self.mu, self.d
        mu.replay_buffer.batchify_device(self.idx_p, want=('obs0', 'acs'), out=self.pb)
        # expert pairs: consecutive rows of the shuffled pool (demo_dataset.py:28-41)
        ops.check(ops._L().samx_replay_gather(
            ops.ptr(self.pool.obs0), ops.ptr(self.pool.acs), None, None, None, self.pool.num_entries, 0,
            mu.ob_dim, mu.ac_dim, ops.ptr(self.idx_e), self.B, ops.ptr(self.eb['obs0']),
            ops.ptr(self.eb['acs']), None, None, None, ops.stream_ptr()))
        if d.hps.rmsify_obs:  # sam.py:389-390
            ops.rms_batch_sums(self.pb['obs0'], self.eb['obs0'], d.obs_rms.addvec)

    def _seg_grads(self):
        mu, d = self.mu, self.d
        if d.hps.rmsify_obs:
            ops.rms_commit(d.obs_rms.sums, d.obs_rms.addvec, d.obs_rms.mean, d.obs_rms.std)
        if self.per:  # memory.py:248-309 on the device trees
            mu.replay_buffer.trees.per_sample(mu.replay_buffer.num_entries, self.u, mu.hps.beta,
                                              idxs_out=self.idx_a, iws_out=self.iws,
                                              iws32_out=self.iws32)
        mu.replay_buffer.batchify_device(self.idx_a, out=self.ab)
        ops.step_grads(self.L, self.batch, self.out)
        # single GPU: the local norm IS the global one -> clipped copies + Adam + Polyak in one pass
        ops.step_apply(self.L, self.B, 3 if self.world == 1 else 1, 1.0)
        if self.world == 1:
            self._seg_refresh()

    def _seg_apply(self):
        ops.step_apply(self.L, self.B, 6, 1.0 / self.world)
        self._seg_refresh()

    def _seg_refresh(self):
        if self.per:  # TD errors with the POST-update critic (sam_agent.py:693-701)
            mu = self.mu
            mean, std = mu._mean_std()
            ops.td_errors(mu.dc, mu.critic.flat, self.ab['obs0'], self.ab['acs'], self.out.t['targ_q'],
                          mean, std, mu.ws, out=self.td_post)

    # ------------------------------------------------------------------ plan: graphs + collectives
    def _build_plan(self):
        d = self.d
        ar = self.comm.allreduce_
        if self.world == 1:
            segs, colls = [[self._seg_inputs, self._seg_grads]], [None]
        else:
            segs = [[self._seg_inputs], [self._seg_grads], [self._seg_apply]]
            colls = [(lambda: ar(d.obs_rms.addvec)) if d.hps.rmsify_obs else None,
                     lambda: ar(self.L.grads), None]
        plan = []
        for fns, coll in zip(segs, colls):
            run = (lambda fns=fns: [f() for f in fns])
            if self.use_graph:
                # warm-up on a side stream (module loads, LDS attributes), then capture
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    run()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                plan.append(g.replay)
            else:
                plan.append(run)
            if coll is not None:
                plan.append(coll)
        return plan

    # ------------------------------------------------------------------ host side of a step
    def _draw(self, slot):
        """The reference's host RNG draws for one step, in the reference's order."""
        rb = self.mu.replay_buffer
        B = self.B
        if slot.event is not None:
            slot.event.synchronize()
        # D batch: uniform pairs (sam.py:372-377 -> memory.py:90)
        slot.idx_p.copy_(torch.from_numpy(np.random.randint(low=0, high=rb.num_entries, size=B)))
        # expert batch pointer arithmetic (demo_dataset.py:33-40)
        if self.pool.pointer + B >= self.pool.num_entries:
            self.pool.init_pointer()
        slot.idx_e.copy_(torch.arange(self.pool.pointer, self.pool.pointer + B))
        self.pool.pointer += B
        if self.per:
            slot.u.copy_(torch.from_numpy(rb.draw_uniforms(B)))           # memory.py:278
        else:
            slot.idx_a.copy_(torch.from_numpy(np.random.randint(low=0, high=rb.num_entries, size=B)))

    def _upload(self, slot):
        self.idx_p.copy_(slot.idx_p, non_blocking=True)
        self.idx_e.copy_(slot.idx_e, non_blocking=True)
        if self.per:
            self.u.copy_(slot.u, non_blocking=True)
        else:
            self.idx_a.copy_(slot.idx_a, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        slot.event = ev
        # TF-random tensors of the D graph (discriminator.py:101-109,130-133), fresh every update
        d = self.d
        self.eps_ob.uniform_(0, 1, generator=d.gen)
        self.eps_ac.uniform_(0, 1, generator=d.gen)
        if d.hps.label_smoothing:
            self.real.uniform_(0, 1, generator=d.gen).mul_(-0.5).add_(1.2)
            if not d.hps.one_sided_label_smoothing:
                self.fake.uniform_(0, 1, generator=d.gen).mul_(-0.3).add_(0.3)

    def prepare(self):
        """Build (and, with use_graph, capture) the device plan.  Capturing executes the segments —
        learner state, running sums and host RNG positions are snapshotted and restored, so
        prepare() leaves the learner exactly as it found it."""
        if self.plan is not None:
            return
        mu, d, L = self.mu, self.d, self.L
        keep = [L.params, L.targets, L.m, L.v, L.steps]
        if d.hps.rmsify_obs:
            keep += [d.obs_rms.sums, d.obs_rms.mean, d.obs_rms.std]
        saved = [t.clone() for t in keep]
        rng_np, rng_py = np.random.get_state(), _random.getstate()
        ptr0, gen_state = self.pool.pointer, d.gen.get_state()
        slot = self.slots[0]
        self._draw(slot)
        self._upload(slot)
        torch.cuda.synchronize()
        self.plan = self._build_plan()
        torch.cuda.synchronize()
        for t, sv in zip(keep, saved):
            t.copy_(sv)
        np.random.set_state(rng_np)
        _random.setstate(rng_py)
        self.pool.pointer = ptr0
        d.gen.set_state(gen_state)
        torch.cuda.synchronize()

    def step(self):
        if self.plan is None:
            self.prepare()
        slot = self.slots[self.step_no % len(self.slots)]
        self._draw(slot)
        self._upload(slot)
        for run in self.plan:
            run()
        if self.per:
            # |td| + 1e-6 in float32 (numpy: float32 array + python float stays float32,
            # sam_agent.py:702), leaves = libm pow on the host, trees updated on the device
            self._td_host.copy_(self.td_post, non_blocking=True)
            self._idx_host.copy_(self.idx_a, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            new_priorities = np.abs(self._td_host.numpy()) + np.float32(1e-6)
            self.mu.replay_buffer.update_priorities(self._idx_host.numpy(), new_priorities)
        self.step_no += 1

    def outputs(self):
        """What the reference's loop logs per step (sam.py:398-431), still in HBM."""
        L, t = self.L, self.out.t
        return dict(d_losses=t['d_losses'], actor_losses=t['actor_losses'],
                    critic_losses=t['critic_losses'], actor_grads=L.seg(L.clipped, 0),
                    critic_grads=L.seg(L.clipped, 1), d_grads=L.seg(L.clipped, 2))
