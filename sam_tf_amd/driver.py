"""Host driver for one "SAM update step" — the body of the reference's training loop
(imitation_algorithms/sam.py:323-433) with d_steps = g_steps = 1:

    discriminator update : uniform replay pairs + expert pairs -> d.obs_rms.update -> D loss/grad
                           (+GP) -> all-reduce -> clip -> Adam                 (sam.py:369-406)
    agent update         : replay sample -> TD target -> actor & critic loss/grad -> all-reduce ->
                           clip -> Adam x2 -> [PER priority refresh]           (sam.py:421 -> train())
    target update        : Polyak                                              (sam.py:433)

Everything stays in HBM.  With `use_graph` the kernel sequence between two collectives is captured
once into a HIP graph (torch.cuda.CUDAGraph on the stream the C ABI is handed) and replayed, which
removes the per-launch host cost that dominates the 64-wide configurations.  Host work per step is
then: two np.random.randint draws, two small pinned H2D copies, the graph launches and (N>1) three
RCCL all-reduces.  Data-parallel ranks each own a replay shard and a minibatch; only gradients and
the running-mean sums are exchanged.
"""
import numpy as np
import torch

from . import ops


class _Slot(object):
    """Pinned staging for one step's host-drawn indices; an event guards its reuse."""

    def __init__(self, B):
        self.idx_p = torch.empty(B, dtype=torch.int64).pin_memory()
        self.idx_a = torch.empty(B, dtype=torch.int64).pin_memory()
        self.idx_e = torch.empty(B, dtype=torch.int64).pin_memory()
        self.u = torch.empty(B, dtype=torch.float64).pin_memory()
        self.event = None


class UpdateStepDriver(object):

    def __init__(self, mu, d, expert_pool, use_graph=True, n_slots=4):
        self.mu, self.d, self.pool = mu, d, expert_pool
        self.comm = mu.comm
        self.world = self.comm.Get_size()
        self.B = B = mu.hps.batch_size
        dev = mu.device
        self.dev = dev
        self.per = bool(mu.hps.prioritized_replay)
        self.use_graph = bool(use_graph)
        ob, ac = mu.ob_dim, mu.ac_dim
        f32 = dict(dtype=torch.float32, device=dev)
        # static device buffers (addresses are baked into the captured graphs)
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
        self.d_clipped = torch.zeros_like(d.grad)
        self.pool_store = dict(obs0=expert_pool.obs0, acs=expert_pool.acs)
        self.slots = [_Slot(B) for _ in range(n_slots)]
        self.step_no = 0
        self.graphs = None
        self.d.ws(B)

    # ------------------------------------------------------------------ segments (device work only)
    def _seg_d_inputs(self):
        mu, d = self.mu, self.d
        mu.replay_buffer.batchify_device(self.idx_p, want=('obs0', 'acs'), out=self.pb)
        # expert pairs: consecutive rows of the shuffled pool (demo_dataset.py:28-41)
        ops.check(ops._L().samx_replay_gather(
            ops.ptr(self.pool.obs0), ops.ptr(self.pool.acs), None, None, None, self.pool.num_entries, 0,
            mu.ob_dim, mu.ac_dim, ops.ptr(self.idx_e), self.B, ops.ptr(self.eb['obs0']),
            ops.ptr(self.eb['acs']), None, None, None, ops.stream_ptr()))
        if d.hps.rmsify_obs:
            ops.rms_batch_sums(self.pb['obs0'], self.eb['obs0'], d.obs_rms.addvec)

    def _seg_d_grad(self):
        d = self.d
        if d.hps.rmsify_obs:
            ops.rms_commit(d.obs_rms.sums, d.obs_rms.addvec, d.obs_rms.mean, d.obs_rms.std)
        real = self.real if d.hps.label_smoothing else None
        fake = self.fake if (d.hps.label_smoothing and not d.hps.one_sided_label_smoothing) else None
        d.grads_device(self.pb['obs0'], self.pb['acs'], self.eb['obs0'], self.eb['acs'],
                       randoms=(self.eps_ob, self.eps_ac, real, fake))
        ops.clip_copy(d.grad, d.hps.clip_norm, d.optimizer.partials, out=self.d_clipped)

    def _seg_d_apply(self):
        self.d.optimizer.enqueue(self.d.grad)

    def _seg_agent_grad(self):
        mu = self.mu
        if self.per:
            mu.replay_buffer.trees.per_sample(mu.replay_buffer.num_entries, self.u, mu.hps.beta,
                                              idxs_out=self.idx_a, iws_out=self.iws,
                                              iws32_out=self.iws32)
        mu.replay_buffer.batchify_device(self.idx_a, out=self.ab)
        batch = dict(self.ab)
        batch['iws32'] = self.iws32
        mu.grads_device(batch)
        a = slice(0, mu.Pa); c = slice(mu.Pa, mu.Pa + mu.Pc)
        ops.clip_copy(mu.actor_grad, mu.hps.clip_norm, mu.actor_ops['optimizer'].partials,
                      out=mu.clipped[a])
        ops.clip_copy(mu.critic_grad, mu.hps.clip_norm, mu.critic_ops['optimizer'].partials,
                      out=mu.clipped[c])

    def _seg_agent_apply(self):
        mu = self.mu
        tau = mu.hps.polyak
        mu.actor_ops['optimizer'].enqueue(mu.actor_grad, targ=mu.targ_actor.flat, tau=tau)
        mu.critic_ops['optimizer'].enqueue(mu.critic_grad, targ=mu.targ_critic.flat, tau=tau)
        if self.per:
            mean, std = mu._mean_std()
            ops.td_errors(mu.dc, mu.critic.flat, self.ab['obs0'], self.ab['acs'], mu.targ_q, mean, std,
                          mu.ws, out=mu.td)

    # ------------------------------------------------------------------ plan: graphs + collectives
    def _build_plan(self):
        """Device segments separated by the step's collectives.  N=1: one segment, one graph."""
        mu, d = self.mu, self.d
        ar = self.comm.allreduce_
        if self.world == 1:
            segs = [[self._seg_d_inputs, self._seg_d_grad, self._seg_d_apply, self._seg_agent_grad,
                     self._seg_agent_apply]]
            colls = [None]
        else:
            segs = [[self._seg_d_inputs], [self._seg_d_grad], [self._seg_d_apply, self._seg_agent_grad],
                    [self._seg_agent_apply]]
            colls = [(lambda: ar(d.obs_rms.addvec)) if d.hps.rmsify_obs else None,
                     lambda: ar(d.grad), lambda: ar(mu.grads), None]
        plan = []
        for fns, coll in zip(segs, colls):
            run = (lambda fns=fns: [f() for f in fns])
            if self.use_graph:
                # warm-up on a side stream (first-touch allocations, module loads), then capture
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
        mu = self.mu
        rb = mu.replay_buffer
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
            slot.u.copy_(torch.from_numpy(rb.draw_uniforms(B)))
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
        """Build (and, with use_graph, capture) the device plan.  Capturing executes the segments
        once for warm-up — optimiser state and running sums are snapshotted and restored, so
        prepare() leaves the learner exactly as it found it."""
        if self.graphs is not None:
            return
        mu, d = self.mu, self.d
        keep = [mu.params, mu.targets, d.params, d.optimizer.m, d.optimizer.v, d.optimizer.t_dev]
        for k in ('actor_ops', 'critic_ops'):
            o = getattr(mu, k)['optimizer']
            keep += [o.m, o.v, o.t_dev]
        if d.hps.rmsify_obs:
            keep += [d.obs_rms.sums, d.obs_rms.mean, d.obs_rms.std]
        saved = [t.clone() for t in keep]
        slot = self.slots[0]
        rng_np = np.random.get_state()
        import random as _random
        rng_py = _random.getstate()
        ptr0 = self.pool.pointer
        gen_state = d.gen.get_state()
        self._draw(slot)
        self._upload(slot)
        torch.cuda.synchronize()
        self.graphs = self._build_plan()
        torch.cuda.synchronize()
        for t, sv in zip(keep, saved):
            t.copy_(sv)
        np.random.set_state(rng_np)
        _random.setstate(rng_py)
        self.pool.pointer = ptr0
        d.gen.set_state(gen_state)
        torch.cuda.synchronize()

    def step(self):
        mu, d = self.mu, self.d
        if self.graphs is None:
            self.prepare()
        slot = self.slots[self.step_no % len(self.slots)]
        self._draw(slot)
        self._upload(slot)
        for run in self.graphs:
            run()
        d.optimizer.t += 1
        mu.actor_ops['optimizer'].t += 1
        mu.critic_ops['optimizer'].t += 1
        if self.per:
            mu._td_host.copy_(mu.td, non_blocking=True)
            mu._idx_host.copy_(self.idx_a, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            new_priorities = np.abs(mu._td_host.numpy()) + np.float32(1e-6)
            mu.replay_buffer.update_priorities(mu._idx_host.numpy(), new_priorities)
        self.step_no += 1

    def outputs(self):
        """What the reference's loop logs per step (sam.py:398-431), still in HBM."""
        mu = self.mu
        return dict(d_losses=self.d.losses, d_grads=self.d_clipped, actor_losses=mu.actor_losses,
                    critic_losses=mu.critic_losses, actor_grads=mu.clipped[:mu.Pa],
                    critic_grads=mu.clipped[mu.Pa:])
