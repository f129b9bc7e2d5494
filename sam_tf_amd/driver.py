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
(torch.cuda.CUDAGraph on the stream the C ABI is handed) and replayed.  Rows are read in place
from the replay storage / expert pool through index words (no gather pass) and the discriminator's
random tensors are drawn inside the kernels (Philox keyed by seed and update count).  Host work per
step: the reference's RNG draws (np.random.randint / random.random, same generators, same order), ONE
packed pinned H2D copy, the graph launches, and for N > 1 the RCCL calls.  Data-parallel ranks each own a
replay shard and a minibatch; only gradients and running-mean sums are exchanged.
"""
import random as _random

import numpy as np
import torch

from . import ops


class _Slot(object):
    """Pinned staging for one step's host-drawn numbers (one packed H2D copy); an event guards reuse.
    Layout (int64 words): [idx_p (B) | idx_a (B) | e_start (1)] then B float64 PER uniforms."""

    def __init__(self, B):
        self.words = torch.empty(2 * B + 1 + B, dtype=torch.int64).pin_memory()
        self.idx_p = self.words[:B]
        self.idx_a = self.words[B:2 * B]
        self.e_start = self.words[2 * B:2 * B + 1]
        self.u = self.words[2 * B + 1:].view(torch.float64)
        # numpy views of the pinned words: the host fills them without going through torch ops
        w = self.words.numpy()
        self.np_idx_p, self.np_idx_a, self.np_e = w[:B], w[B:2 * B], w[2 * B:2 * B + 1]
        self.np_u = w[2 * B + 1:].view(np.float64)
        self.event = torch.cuda.Event()
        self.used = False


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
        f32 = dict(dtype=torch.float32, device=dev)
        rb = mu.replay_buffer
        assert rb.num_demos == 0, "driver assumes a replay buffer without pinned demos"
        # static device buffers (their addresses are baked into the captured graphs)
        self.words = torch.zeros(3 * B + 1, dtype=torch.int64, device=dev)
        self.idx_p = self.words[:B]
        self.idx_a = self.words[B:2 * B]
        self.e_start = self.words[2 * B:2 * B + 1]
        self.u = self.words[2 * B + 1:].view(torch.float64)
        self.iws = torch.zeros(B, dtype=torch.float64, device=dev)
        self.iws32 = torch.zeros(B, **f32)
        hp = d.hps
        st = rb.store
        self.rng_seed = int(hp.seed) * 1000003 + 7919 * (self.comm.Get_rank() + 1)
        self.label_mode = 0 if not hp.label_smoothing else (1 if hp.one_sided_label_smoothing else 2)
        self._batch_start = None
        self.out = ops.make_step_out(B, dev)
        self.td_post = torch.zeros(B, **f32)
        self.gb = {k: torch.zeros(B, st[k].shape[1], **f32) for k in ('obs0', 'acs')}  # PER refresh rows
        self._td_host = torch.empty(B, dtype=torch.float32).pin_memory()
        self._idx_host = torch.empty(B, dtype=torch.int64).pin_memory()
        self.slots = [_Slot(B) for _ in range(n_slots)]
        self.step_no = 0
        self.plan = None
        self.fused = self.L.fused(B)
        self._make_batch()

    def _make_batch(self):
        """(Re)build the samx_batch: every row is read IN PLACE from the replay storage / expert pool
        through the index words (no gather pass); D randoms are drawn inside the kernels."""
        rb, st = self.mu.replay_buffer, self.mu.replay_buffer.store
        self._batch_start = rb.start
        self.batch = ops.make_batch(
            self.B, p_obs=st['obs0'], p_acs=st['acs'], e_obs=self.pool.obs0, e_acs=self.pool.acs,
            obs0=st['obs0'], acs=st['acs'], rews=st['rews'].view(-1), dones1=st['dones1'].view(-1),
            obs1=st['obs1'], iws=self.iws32 if self.per else None, p_idx=self.idx_p, a_idx=self.idx_a,
            e_start=self.e_start, ring_start=rb.start, ring_len=rb.limit, rng_seed=self.rng_seed,
            label_smoothing=self.label_mode)

    # ------------------------------------------------------------------ device segments
    def _seg_inputs(self):
        """d.obs_rms.update on the D batch (sam.py:389-390); single process: sums + commit at once."""
        d, rb = self.d, self.mu.replay_buffer
        if d.hps.rmsify_obs:
            r = d.obs_rms
            ops.d_input_stats(rb.store['obs0'], self._batch_start, self.idx_p, self.pool.obs0, self.e_start,
                              self.B, r.addvec, commit=self.world == 1, sums=r.sums, mean=r.mean, std=r.std)

    def _seg_grads(self):
        mu, d = self.mu, self.d
        if d.hps.rmsify_obs and self.world > 1:
            ops.rms_commit(d.obs_rms.sums, d.obs_rms.addvec, d.obs_rms.mean, d.obs_rms.std)
        if self.per:  # memory.py:248-309 on the device trees
            mu.replay_buffer.trees.per_sample(mu.replay_buffer.num_entries, self.u, mu.hps.beta,
                                              idxs_out=self.idx_a, iws_out=self.iws,
                                              iws32_out=self.iws32)
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
            mu.replay_buffer.batchify_device(self.idx_a, want=('obs0', 'acs'), out=self.gb)
            ops.td_errors(mu.dc, mu.critic.flat, self.gb['obs0'], self.gb['acs'], self.out.t['targ_q'],
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
        if slot.used:
            slot.event.synchronize()
        # D batch: uniform pairs (sam.py:372-377 -> memory.py:90)
        slot.np_idx_p[:] = np.random.randint(low=0, high=rb.num_entries, size=B)
        # expert batch pointer arithmetic (demo_dataset.py:33-40)
        if self.pool.pointer + B >= self.pool.num_entries:
            self.pool.init_pointer()
        slot.np_e[0] = self.pool.pointer
        self.pool.pointer += B
        if self.per:
            slot.np_u[:] = rb.draw_uniforms(B)                                     # memory.py:278
        else:
            slot.np_idx_a[:] = np.random.randint(low=0, high=rb.num_entries, size=B)

    def _upload(self, slot):
        self.words.copy_(slot.words, non_blocking=True)   # ONE packed H2D copy per step
        slot.event.record()
        slot.used = True

    def step_randoms(self, step):
        """The D randoms the kernels drew for reward-net update `step` (for checkers)."""
        return ops.disc_randoms(self.rng_seed, step, self.B, self.mu.ob_dim, self.mu.ac_dim,
                                want_real=self.label_mode >= 1, want_fake=self.label_mode == 2,
                                device=self.dev)

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
        ptr0 = self.pool.pointer
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
        torch.cuda.synchronize()

    def step(self):
        if self.plan is None:
            self.prepare()
        assert self.mu.replay_buffer.start == self._batch_start, \
            "replay ring moved: call driver.refresh() after appending transitions"
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

    def refresh(self):
        """Re-read the replay ring geometry (after rollouts appended transitions) and re-capture."""
        self.plan = None
        self._make_batch()
        self.prepare()

    def outputs(self):
        """What the reference's loop logs per step (sam.py:398-431), still in HBM."""
        L, t = self.L, self.out.t
        return dict(d_losses=t['d_losses'], actor_losses=t['actor_losses'],
                    critic_losses=t['critic_losses'], actor_grads=L.seg(L.clipped, 0),
                    critic_grads=L.seg(L.clipped, 1), d_grads=L.seg(L.clipped, 2))
