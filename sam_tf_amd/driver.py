"""Host driver for the body of the reference's training loop (imitation_algorithms/sam.py:323-433).
One "SAM update step" (d_steps = g_steps = 1, the bench metric) is:

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

d_steps / g_steps other than 1 (the launcher runs 1 discriminator and 3 agent updates per training step):
a training step becomes a sequence of ROWS, each with a role subset — D-only rows first, then rows that
carry a D and an agent update side by side, then agent-only rows.  The two updates do not read each other's
parameters (d_actor_loss_scale = 0, or an untied reward-net copy), so this is the reference's
"all D updates, then all agent updates" with every update sequence in its own order; the host RNG draws
keep the reference's order.  `step()` / `run(n)` count rows.

The other things the reference's training step does between updates (sam.py:330-367,408-415):
  * `window`: every d_step first trains D on pairs from the most recent `window` transitions
    (RB.sample_recent, np.random.choice), then on uniform pairs — two D rows per d_step, the first with its
    index words drawn the way np.random.choice draws them (randint over the window, mapped to ring offsets);
  * adaptive parameter noise: every `pn_adapt_frequency` training steps mu.adapt_param_noise(comm) runs after
    the D updates and before the agent updates.  It reads the actor as it is BEFORE that training step's
    agent updates and nothing the D updates write, so it is executed at the head of the launch that starts that
    training step, with its replay draw placed between the D draws and the agent draws of the step (host RNG
    order kept); such a training step is always the first of its launch;
  * a reward net TIED to the actor loss (d_actor_loss_scale != 0, tie_actor_reward_net): the agent update reads
    the D update of the same training step, so no row carries both: all D rows, then all agent rows.
"""
import random as _random

import gc
import os

import numpy as np
import torch

from . import ops


class _Slot(object):
    """Pinned staging for the host-drawn numbers of up to K consecutive steps (one packed H2D copy);
    an event guards reuse.  Row layout (int64 words): [idx_p (B) | idx_a (B) | e_start (1)], B float64
    PER uniforms, then the PREVIOUS step's priority update: B indices and B float64 leaf values."""

    def __init__(self, B, K):
        self.W = 5 * B + 1
        self.words = torch.empty(K, self.W, dtype=torch.int64).pin_memory()
        # numpy views of the pinned words: the host fills them without going through torch ops
        w = self.words.numpy()
        self.np_words = w
        self.np_idx = w[:, :2 * B].reshape(K, 2, B)   # [k][0] = idx_p, [k][1] = idx_a
        self.np_idx_p, self.np_idx_a, self.np_e = w[:, :B], w[:, B:2 * B], w[:, 2 * B]
        self.np_u = w[:, 2 * B + 1:3 * B + 1].view(np.float64)
        self.np_prev_idx = w[:, 3 * B + 1:4 * B + 1]
        self.np_prev_leaf = w[:, 4 * B + 1:].view(np.float64)
        self.event = torch.cuda.Event()
        self.used = False
        self.predrawn = False


class UpdateStepDriver(object):

    D, A = 1, 2   # role bits of a row

    def __init__(self, mu, d, expert_pool, use_graph=True, n_slots=4, steps_per_launch=16, d_steps=1, g_steps=1,
                 window=None, pn_adapt_frequency=None, per_host_pow=False):
        self.mu, self.d, self.pool = mu, d, expert_pool
        assert d_steps >= 1 and g_steps >= 1
        self.window = None if window is None else int(window)
        assert self.window is None or self.window > mu.hps.batch_size, "must have window > batch_size"  # sam.py:334
        # the D updates of a training step, in order: per d_step [recent-window pairs,] uniform pairs
        self.dseq = (['w', 'u'] if self.window is not None else ['u']) * int(d_steps)
        self.nD = len(self.dseq)
        tied = mu.hps.d_actor_loss_scale != 0 and mu.hps.tie_actor_reward_net
        both = 0 if tied else min(self.nD, g_steps)   # tied: the agent update reads this step's D updates
        self.d_steps, self.g_steps = int(d_steps), int(g_steps)
        self.sched = [self.D] * (self.nD - both) + [self.D | self.A] * both + [self.A] * (g_steps - both)
        self.S = len(self.sched)
        # adaptive parameter noise (sam.py:408-415): due when train_step_in_iter % frequency == 0
        self.pn_freq = int(pn_adapt_frequency) if (pn_adapt_frequency and mu.param_noise is not None) else 0
        self.train_step_in_iter = 0   # the reference's `training_step` (start_iteration() resets it)
        self.pn_log = []              # (pn_dist, pn_cur_std) per adaption, what sam.py:412-415 appends
        self._cache = None            # the host draws of the training step in progress (_draw_training_step)
        self.L = mu.L
        self.comm = mu.comm
        self.world = self.comm.Get_size()
        self.B = B = mu.hps.batch_size
        dev = mu.device
        self.dev = dev
        self.per = bool(mu.hps.prioritized_replay)
        self.use_graph = bool(use_graph)
        # PER, one process: the priority refresh (|td| + 1e-6 -> priority ** alpha, max_priority, the asserts;
        # sam_agent.py:693-703 -> memory.py:370-379) stays ON THE DEVICE — its pow carries the host libm's bits
        # (csrc/pow_bits.h), checked against this machine's libm when the driver is built.  Rows then need no
        # host between them and are packed K per graph like uniform replay.  per_host_pow=True (or
        # SAMX_PER_HOST_POW=1, or a libm the routine does not reproduce) keeps the host refresh of round 1.
        self.per_dev = bool(self.per and self.comm.Get_size() == 1 and not per_host_pow and
                            os.environ.get('SAMX_PER_HOST_POW', '0') != '1' and ops.pow_matches_host() and
                            # ... and the device build, with THIS buffer's exponents (alpha, -beta)
                            ops.pow_device_matches_host(mu.replay_buffer.alpha, mu.replay_buffer.beta, dev))
        # Several consecutive steps per graph launch: possible when nothing between two steps needs
        # the host — one process (no collective) and uniform replay (PER's priority refresh computes
        # p ** alpha with the host libm).  The host RNG draws of those steps are made up front, in the
        # reference's order, so the sampled indices are the same as with one launch per step.
        # N > 1 also packs K steps per host iteration: the D-input sums of all K steps are computed and
        # all-reduced ONCE (they do not depend on the learner), which leaves one collective per step.
        self.fused = self.L.fused(B)
        # 256x256 learners (samx_step_path 2): layer-wise in how the driver plans launches and collectives,
        # fused in how the D-input statistics ride inside the step (one process)
        self.wide = self.L.path(B) == 2
        # (SAMX_KEEP_IMAGES=0: every row packs its weight images itself, see _img_bits)
        self.keep_images = os.environ.get('SAMX_KEEP_IMAGES', '1') != '0'
        self.upload_kernel = os.environ.get('SAMX_UPLOAD_KERNEL', '1') != '0'
        # ... and only the words of a row this configuration reads: [idx_p | idx_a | e_start] + the two geometry words
        # under uniform replay, + the B uniforms with the device-side priority refresh (whole rows otherwise)
        B_ = int(mu.hps.batch_size)
        self.upload_len = 5 * B_ + 1
        # the K-row launches' replay index draws through the library's restatement of numpy's randint (checked against
        # this numpy when the driver is built; the generator is left exactly where it was)
        self.host_rng = ops.host_rng_ready()
        # whole-network path, one process: the optimiser pass rides in the reduction launch (samx_step_grads_part bit 9;
        # SAMX_APPLY_IN_REDUCTION=0 restores the two launches for A/B runs — same bits either way)
        self.apply_in_reduction = os.environ.get('SAMX_APPLY_IN_REDUCTION', '1') != '0'
        # PER on the layer-wise path, one process: the D chain beside the PER work + agent roles (see
        # _seg_grads_two_branches); SAMX_PER_TWO_BRANCHES=0 keeps every row on one chain
        self.per_two_branches = bool(self.per_dev and not self.fused and not self.wide and self.world == 1 and
                                     B >= 1024 and os.environ.get('SAMX_PER_TWO_BRANCHES', '1') != '0')
        self._side_stream = torch.cuda.Stream(device=dev) if self.per_two_branches else None
        m = 1
        while 2 * m * self.S <= int(steps_per_launch):
            m *= 2
        # rows per launch: whole training steps
        self.K = m * self.S if (self.use_graph and (not self.per or self.per_dev)) else 1
        if self.world > 1 and not (self.fused or self.wide):
            self.K = 1   # layer-wise path under N > 1: the three-segment plan, one step per launch
        # N > 1 with a communicator whose all-reduce runs on the caller's stream (sam_tf_amd.rccl): the
        # collectives are captured INSIDE the K-step graph, the host issues one launch per K steps
        self.collectives_in_graph = bool(self.world > 1 and (self.fused or self.wide) and self.use_graph and
                                         getattr(self.comm, 'capturable', False) and
                                         os.environ.get('SAMX_RCCL_GRAPH', '1') != '0')
        # N > 1 with return normalisation + PopArt: ret_rms.update all-reduces the targets' batch sums between
        # the target pass and the PopArt rescale (sam_agent.py:662-672, mpi_running_mean_std.py:47-59): the
        # row's gradient call splits there (samx_step_grads_part bits 4/5)
        self.popart_split = bool(self.world > 1 and mu.hps.rmsify_rets and mu.hps.enable_popart)
        self.ret_addvec = torch.zeros(6, dtype=torch.float64, device=dev) if self.popart_split else None
        f32 = dict(dtype=torch.float32, device=dev)
        rb = mu.replay_buffer
        # (demonstrations pinned in front of the ring — add_demos_to_mem, memory.py:186-198 — need nothing special
        # here: logical index -> storage row stays (start + idx) % limit, memory.py:25-28, and the priority refresh is
        # handed num_demos / demos_eps: test_k_steps_match_oracle[C1-demos | C3-demos])
        # static device buffers (their addresses are baked into the captured graphs)
        self.W = 5 * B + 1
        self.words = torch.zeros(self.K, self.W, dtype=torch.int64, device=dev)
        # step 0's views (what the single-step plan and the checkers read)
        self.idx_p = self.words[0, :B]
        self.idx_a = self.words[0, B:2 * B]
        self.e_start = self.words[0, 2 * B:2 * B + 1]
        self.u = self.words[0, 2 * B + 1:3 * B + 1].view(torch.float64)
        # {start, num_entries} of the ring as a launch finds it (see _write_geom): two spare words of row 0 — the PER
        # uniforms' place under uniform replay, the host-refresh leaves' place under prioritized replay
        self.geom_at = (3 * B + 1) if self.per else (2 * B + 1)
        if not self.per or self.per_dev:   # (the host-side priority refresh also sends the previous update: whole rows)
            self.upload_len = self.geom_at + 2
        self.geom = self.words[0, self.geom_at:self.geom_at + 2]
        # PER: the previous step's priority update rides in this launch's upload and is applied by the
        # first kernel of the launch (the host libm pow that produces it is the only host work between
        # two PER steps, memory.py:376-377)
        self.prev_idx = self.words[0, 3 * B + 1:4 * B + 1]
        self.prev_leaf = self.words[0, 4 * B + 1:].view(torch.float64)
        self._pending = None
        if self.per_dev:   # the pending update in HBM: the indices and POST-update TD errors of the last agent row
            self.pend_idx = torch.full((B,), -1, dtype=torch.int64, device=dev)   # -1: nothing pending
            self.pend_td = torch.zeros(B, dtype=torch.float32, device=dev)
            self.leaf_buf = torch.zeros(B, dtype=torch.float64, device=dev)
            self.per_state = ops.make_per_state(rb.max_priority, dev)
            self._dev_pending = False
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
        self._leaf_np = np.empty(B, dtype=np.float64)
        self._td_host_np, self._idx_host_np = self._td_host.numpy(), self._idx_host.numpy()
        # PER, one process: the row's H2D upload and the D2H of the refreshed TD errors are nodes of the
        # captured graph (fixed pinned buffers; the host waits for every row anyway), which leaves ONE
        # host call — the graph launch — between two rows' host arithmetic
        self.per_graph_io = bool(self.per and self.use_graph and self.world == 1 and not self.per_dev)
        self._launch_words = torch.zeros(1, self.W, dtype=torch.int64).pin_memory()
        self._launch_np = self._launch_words.numpy()
        # the captured graph reads _launch_words asynchronously: this event, recorded behind every launch,
        # is waited for before the host overwrites the buffer (rows without an agent update do not end in
        # a stream synchronisation of their own)
        self._launch_event = torch.cuda.Event()
        self._launch_busy = False
        self.slots = [_Slot(B, self.K) for _ in range(n_slots)]
        self._warm_slot = _Slot(B, self.K)   # prepare()'s own staging: never holds pre-drawn words
        self.step_no = 0     # rows done
        self.launch_no = 0
        self.plans = {}
        self._make_batch()

    def _make_batch(self):
        """(Re)build the samx_batch of each of the K steps of a launch: every row is read IN PLACE
        from the replay storage / expert pool through that step's index words (no gather pass); D
        randoms are drawn inside the kernels."""
        rb, st = self.mu.replay_buffer, self.mu.replay_buffer.store
        B = self.B
        # Uniform replay: the index words carry STORAGE rows — the host adds the ring's start to its draws
        # (memory.py:16-35: data[(start + idx) % maxlen]) — and the kernels see a ring that starts at 0, so the
        # captured graphs survive every append, also once the ring is full and each append moves its start (the
        # reference's loop appends between any two training phases; re-capturing there cost more than the phase).
        # The n-step lookahead needs the ring's geometry (window ends at num_entries): {start, num_entries} ride in
        # two spare words of row 0 of the same upload and the kernel reads them there
        # (samx_nstep_lookahead_many_rows).  Prioritized replay with the refresh on the device: the trees and the
        # priority update work on LOGICAL indices, so the sampling kernels read {start, num_entries} from the same two
        # device words and write the sampled logical indices (index words, pending-update buffer) NEXT TO their storage
        # rows (self.rows_a), which is what the network kernels gather from (samx_per_sample_rows /
        # samx_per_row_prologue_td_rows).  The host-refresh fallback keeps logical index words and re-captures.
        self.storage_rows = bool((not self.per or self.per_dev) and os.environ.get('SAMX_STORAGE_ROWS', '1') != '0')
        self.per_rows = bool(self.per and self.storage_rows)
        if self.per_rows:
            self.rows_a = torch.zeros(self.K, B, dtype=torch.int64, device=self.dev)
            self._junk_idx = torch.zeros(B, dtype=torch.int64, device=self.dev)
        self._batch_start = 0 if self.storage_rows else rb.start
        self._batch_entries = rb.num_entries
        # fused path: the d.obs_rms commit rides inside the step kernels; the batch sums come from one
        # kernel per launch (all-reduced once per launch when N > 1)
        self.inline_stats = bool((self.fused or self.wide) and self.d.hps.rmsify_obs)
        self.sums_small = 2 * B * self.mu.ob_dim <= 16384   # what the one-workgroup sums kernels hold (samx.h)
        # layer-wise path, one process: samx_step_grads computes and commits the row's D statistics itself,
        # on the discriminator's stream, while the agent roles already run (no separate stats segment)
        self.inline_stats_lw = bool(not self.fused and not self.wide and self.world == 1 and
                                    self.d.hps.rmsify_obs)
        r = self.d.obs_rms if (self.inline_stats or self.inline_stats_lw) else None
        if r is not None:  # the K batches' [sum, sumsq, count], filled by ONE kernel at the head of a launch
            self.addvecs = torch.zeros(self.K, 2 * self.mu.ob_dim + 1, dtype=torch.float64, device=self.dev)
        # n-step returns with uniform replay: every row's indices are known when the launch starts and the
        # lookahead (memory.py:113-160) reads only the replay storage -> ONE kernel for the K rows at the
        # head of the launch instead of one per row (PER samples row k + 1 after row k's priority update)
        self.nstep_ahead = bool(self.mu.hps.n_step_returns and not self.per)
        nk = {}
        if self.nstep_ahead:
            f32 = dict(dtype=torch.float32, device=self.dev)
            self.n_end = torch.zeros(self.K, B, dtype=torch.int64, device=self.dev)
            self.n_rews, self.n_dones, self.n_len = (torch.zeros(self.K, B, **f32) for _ in range(3))
        self.batches = []
        for k in range(self.K):
            w = self.words[k]
            if self.nstep_ahead:
                nk = dict(n_end=self.n_end[k], n_rews=self.n_rews[k], n_dones=self.n_dones[k], n_len=self.n_len[k])
            self.batches.append(ops.make_batch(
                B, p_obs=st['obs0'], p_acs=st['acs'], e_obs=self.pool.obs0, e_acs=self.pool.acs,
                obs0=st['obs0'], acs=st['acs'], rews=st['rews'].view(-1), dones1=st['dones1'].view(-1),
                obs1=st['obs1'], iws=self.iws32 if self.per else None, p_idx=w[:B],
                a_idx=self.rows_a[k] if self.per_rows else w[B:2 * B],
                e_start=w[2 * B:2 * B + 1], ring_start=self._batch_start, ring_len=rb.limit, rng_seed=self.rng_seed,
                label_smoothing=self.label_mode, num_entries=rb.num_entries,
                d_rms_sums=r.sums if r else None,
                d_rms_addvec=self.addvecs[k] if r else None, ret_addvec=self.ret_addvec, **nk))
        self.batch = self.batches[0]

    # ------------------------------------------------------------------ device segments
    def _seg_inputs(self, k=0, role=3):
        """d.obs_rms.update on the D batch (sam.py:389-390); single process: sums + commit at once."""
        d, rb = self.d, self.mu.replay_buffer
        if d.hps.rmsify_obs and not (self.inline_stats or self.inline_stats_lw) and role & self.D:
            r = d.obs_rms
            w = self.words[k]
            ops.d_input_stats(rb.store['obs0'], self._batch_start, w[:self.B], self.pool.obs0,
                              w[2 * self.B:2 * self.B + 1], self.B, r.addvec, commit=self.world == 1,
                              sums=r.sums, mean=r.mean, std=r.std)

    def _seg_sums(self, K):
        """The D-input sums of all K rows of the launch (they do not depend on the learner; rows
        without a D update carry index words 0 and their sums are not used)."""
        if self.inline_stats:
            rb, B = self.mu.replay_buffer, self.B
            # (large batches of wide observations: the two-stage sums of the K rows in two launches)
            ops.d_input_sums_many(rb.store['obs0'], self._batch_start, self.words[:K], B,
                                  self.pool.obs0, self.addvecs[:K])

    def _seg_lookahead(self, K):
        """The n-step lookahead of all K rows of the launch (rows without an agent update carry index
        words 0 and their results are not used)."""
        if self.nstep_ahead:
            rb, hp = self.mu.replay_buffer, self.mu.hps
            if self.storage_rows:
                ops.nstep_lookahead_many_rows(rb.store, self.geom, self.words[:K], self.B, self.B, hp.n, hp.gamma,
                                              self.n_end[:K], self.n_rews[:K], self.n_dones[:K], self.n_len[:K])
            else:
                ops.nstep_lookahead_many(rb.store, self._batch_start, self._batch_entries, self.words[:K], self.B,
                                         self.B, hp.n, hp.gamma, self.n_end[:K], self.n_rews[:K], self.n_dones[:K],
                                         self.n_len[:K])

    def _masks(self, role):
        """(part bits, apply-mode bits) that leave out the roles a row does not carry (samx.h)."""
        return ((0 if role & self.D else 4) | (0 if role & self.A else 8),
                (0 if role & self.D else 8) | (0 if role & self.A else 16))

    def _row_views(self, k):
        """(idx_a, u) of row k inside the launch's words."""
        B, w = self.B, self.words[k]
        return w[B:2 * B], w[2 * B + 1:3 * B + 1].view(torch.float64)

    def _gather_idx(self, k):
        """What the row gathers of row k's agent kernels read: storage rows, or the logical index words."""
        return self.rows_a[k] if self.per_rows else self._row_views(k)[0]

    def _per_td(self):
        rb = self.mu.replay_buffer
        return (self.pend_td, rb.alpha, rb.num_demos, rb.demos_eps, self.per_state)

    def _seg_trees(self, role=3, k=0):
        """PER: the pending priority update, then memory.py:248-309 on the device trees."""
        mu = self.mu
        if self.per and role & self.A:
            rb = mu.replay_buffer
            idx_a, u = self._row_views(k)
            if self.per_rows:   # ring geometry from the launch's device words: the graph survives appends
                ops.per_leaves_from_td_geom(self.pend_td, self.pend_idx, rb.st_cap, self.geom, rb.alpha, rb.num_demos,
                                            rb.demos_eps, self.leaf_buf, self.per_state)
                rb.trees.set(self.pend_idx, self.leaf_buf)
                ops.per_sample_rows(rb.trees, self.geom, rb.limit, u, mu.hps.beta, idx_a, self.iws, self.iws32,
                                    self.rows_a[k], idx_copy_out=self.pend_idx)
                return
            if self.per_dev:
                ops.per_leaves_from_td(self.pend_td, self.pend_idx, rb.st_cap, rb.num_entries, rb.alpha, rb.num_demos,
                                       rb.demos_eps, self.leaf_buf, self.per_state)
                rb.trees.set(self.pend_idx, self.leaf_buf)
            else:
                rb.trees.set(self.prev_idx, self.prev_leaf)
            rb.trees.per_sample(rb.num_entries, u, mu.hps.beta, idxs_out=idx_a, iws_out=self.iws,
                                iws32_out=self.iws32)

    def _img_bits(self, k, K):
        """256-wide path, K rows in one launch: the role kernels' forward products read k-interleaved images of
        W1 / W2 (fused_wide.hip), normally packed from the parameter vectors at the head of every row (one launch,
        ~5.5 us at C4).  Inside ONE launch nothing but the optimiser pass rewrites those matrices, so row 0 packs all
        five networks (part bit 8), every optimiser pass but the last keeps the images current (samx_step_apply mode
        bit 6) and rows 1 .. K-1 pack nothing (part bit 7).  Returns (part bits of row k's gradient calls, mode bit of
        its optimiser pass).  Between launches anything may touch the parameters (root sync, checkpoints, the object
        API), which is why every launch starts with a pack."""
        if not (self.wide and self.keep_images) or K == 1:
            return 0, 0
        return (256 if k == 0 else 128), (64 if k < K - 1 else 0)

    def _seg_targets(self, k=0, role=3, K=1):
        """N > 1 with PopArt: [PER sampling,] the target pass and the local return sums -> ret_addvec."""
        self._seg_trees(role, k)
        ops.step_grads_part(self.L, self.batches[k], self.out, 1 | 16 | self._masks(role)[0] | self._img_bits(k, K)[0])

    def _seg_grads(self, k=0, role=3, trees_done=False, K=1):
        mu, d = self.mu, self.d
        part, mode = self._masks(role)
        img_part, img_mode = self._img_bits(k, K)
        if d.hps.rmsify_obs and self.world > 1 and not self.inline_stats and role & self.D:
            ops.rms_commit(d.obs_rms.sums, d.obs_rms.addvec, d.obs_rms.mean, d.obs_rms.std)
        split = self.popart_split and role & self.A   # _seg_targets + all-reduce(ret_addvec) came before
        if not trees_done and not split:
            self._seg_trees(role, k)
        if split and img_part:
            img_part = 128   # (the target call of this row packed, or found, every image)
        # single GPU: the local norm IS the global one -> clipped copies + Adam + Polyak in one pass; on the
        # whole-network path that pass runs inside the reduction launch (part bit 9, same bits: one launch less per row)
        one = 512 if (self.world == 1 and self.fused and self.apply_in_reduction) else 0
        ops.step_grads_part(self.L, self.batches[k], self.out, 3 | part | (32 if split else 0) | img_part | one)
        if not one:
            ops.step_apply(self.L, self.B, (3 | img_mode if self.world == 1 else 1) | mode, 1.0)
        if self.world == 1:
            self._seg_refresh(role, k)

    def _seg_grads_two_branches(self, k, carry_td, last):
        """One D + agent row of the layer-wise path with prioritized replay, one process, as two branches: the
        discriminator's chain depends neither on the trees nor on the agent's rows, so it runs on a side stream
        while the launch stream carries [the previous row's post-update TD errors ->] priority refresh -> tree update
        -> stratified sample -> the agent roles (which fork inside samx_step_grads_part).  Same kernels on the
        same inputs as _seg_grads: the results are identical; only [TD pass, tree work, gather] — about 200 us of
        one-queue work per row at C5 — no longer run with the rest of the chip idle."""
        role = self.D | self.A
        cur = torch.cuda.current_stream()
        side = self._side_stream
        # (the D chain takes the side stream: the agent call forks its role streams itself, and a fork of a
        # forked stream made hipStreamEndCapture crash on ROCm 7.0 — every fork starts at the launch stream)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.step_grads_part(self.L, self.batches[k], self.out, 1 | 8)    # network passes, discriminator only
        if carry_td:
            self._seg_refresh(role, k - 1)
        self._seg_trees(role, k)
        ops.step_grads_part(self.L, self.batches[k], self.out, 1 | 4)        # network passes, agent roles only
        cur.wait_stream(side)
        ops.step_grads_part(self.L, self.batches[k], self.out, 2)            # the reduction part, every segment
        ops.step_apply(self.L, self.B, 3, 1.0)
        if last:
            self._seg_refresh(role, k)

    def _seg_apply(self, role=3, k=0, K=1):
        ops.step_apply(self.L, self.B, 6 | self._masks(role)[1] | self._img_bits(k, K)[1], 1.0 / self.world)
        self._seg_refresh(role)

    def _seg_refresh(self, role=3, k=0):
        if self.per and role & self.A:  # TD errors with the POST-update critic (sam_agent.py:693-701), rows read in place
            mu = self.mu
            mean, std = mu._mean_std()
            st = mu.replay_buffer.store
            # (one process: the TD errors and the sampled indices also land in pinned host memory, for the
            # host's leaf arithmetic of the next priority update)
            # (device refresh: they land in the pending-update buffers the next agent row's tree update reads)
            io = dict(td_host=self._td_host, idx_host=self._idx_host) if self.per_graph_io else {}
            if self.per_dev:
                io = dict(td_host=self.pend_td, idx_host=self.pend_idx)
            if self.per_rows:   # (the sampling kernel already left the logical indices in the pending-update buffer)
                io = dict(td_host=self.pend_td, idx_host=self._junk_idx)
            ops.td_errors_rows(mu.dc, mu.critic.flat, st['obs0'], st['acs'], self._gather_idx(k), self._batch_start,
                               self.out.t['targ_q'], mean, std, self.td_post, ret=mu.ret, **io)

    def _grads_of(self, role):
        """The slice of the flat gradient a row with these roles fills (what is all-reduced)."""
        L = self.L
        if role == self.D | self.A:
            return L.grads
        n_agent = L.seg(L.grads, 0).numel() + L.seg(L.grads, 1).numel()
        return L.grads[:n_agent] if role == self.A else L.grads[n_agent:]

    # ------------------------------------------------------------------ plan: graphs + collectives
    def _capture(self, run):
        if not self.use_graph:
            return run
        # warm-up on a side stream (module loads, LDS attributes, side streams), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while capturing: a collected CUDAGraph of an older driver releases
        # its memory pool (hipFree), which the capturing thread must not call (torch collects once before
        # the capture for the same reason, but our segments allocate Python objects and can trigger another)
        was = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            # thread_local: the process group's watchdog thread may touch events while we capture
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                run()
        finally:
            if was:
                gc.enable()
        return g.replay

    def _build_plan(self, K, phase=0):
        """The list of callables one launch of K rows executes in order (graph replays and, for N > 1,
        the collectives between them).  phase = position of the first row inside a training step."""
        d, L = self.d, self.L
        ar = self.comm.allreduce_
        roles = [self.sched[(phase + k) % self.S] for k in range(K)]
        any_d = any(r & self.D for r in roles)
        if self.world == 1:
            # (forking the D-input sums beside the PER tree work as a second graph branch was measured:
            # 8.1 k -> 7.1 k steps/s at C3 — a two-branch hipGraph costs more to launch than it overlaps)
            def run():
                rb = self.mu.replay_buffer
                in_prologue = (self.per_graph_io and K == 1 and roles[0] == self.D | self.A and self.inline_stats
                               and self.B < 1024 and self.sums_small)
                if self.per_graph_io and not in_prologue:
                    self.words[:1].copy_(self._launch_words, non_blocking=True)
                if in_prologue:
                    # pending priority update and this row's D-input sums: one launch, two workgroups
                    # ... and workgroup 0 goes on with the row's stratified sample on the updated trees
                    ops.per_row_prologue(rb.trees, self.prev_idx, self.prev_leaf, rb.store['obs0'],
                                         self._batch_start, self.words[0], self.B, self.pool.obs0, self.addvecs[0],
                                         sample=(rb.num_entries, self.u, self.mu.hps.beta, self.idx_a, self.iws,
                                                 self.iws32),
                                         # the row's host-drawn words come straight from pinned host memory
                                         # (all but idx_a, which the sampling writes): no upload node
                                         upload=(self._launch_words.view(-1), self.words[0], self.B, self.B))
                    self._seg_grads(0, roles[0], trees_done=True)
                elif self.per_dev:
                    # rows chained on the device: row k's post-update TD errors are row k + 1's pending update
                    # (the D-input sums of all K rows come from one launch at the head, as with uniform replay:
                    # they depend on host-drawn indices only)
                    if any_d:
                        self._seg_sums(K)
                    two = self.per_two_branches and all(r == self.D | self.A for r in roles)
                    for k in range(K):
                        if two:
                            self._seg_inputs(k, roles[k])
                            self._seg_grads_two_branches(k, carry_td=k > 0, last=k == K - 1)
                        elif self.B < 1024 and roles[k] & self.A:
                            # refresh + tree update + this row's stratified sample (samx_per_row_prologue_td)
                            idx_a, u = self._row_views(k)
                            if self.per_rows:
                                ops.per_row_prologue_rows(rb.trees, self.pend_idx, self.pend_td, rb.alpha, rb.num_demos,
                                                          rb.demos_eps, self.per_state, self.geom, rb.limit, u,
                                                          self.mu.hps.beta, idx_a, self.iws, self.iws32,
                                                          self.rows_a[k], idx_copy_out=self.pend_idx)
                            else:
                                ops.per_row_prologue(rb.trees, self.pend_idx, None, None, self._batch_start,
                                                     self.words[k], self.B, None, None,
                                                     sample=(rb.num_entries, u, self.mu.hps.beta, idx_a, self.iws,
                                                             self.iws32), td=self._per_td())
                            self._seg_inputs(k, roles[k])
                            self._seg_grads(k, roles[k], trees_done=True, K=K)
                        else:
                            self._seg_inputs(k, roles[k])
                            self._seg_grads(k, roles[k], K=K)
                else:
                    if any_d:
                        self._seg_sums(K)
                    self._seg_lookahead(K)
                    for k in range(K):
                        self._seg_inputs(k, roles[k])
                        self._seg_grads(k, roles[k], K=K)
            return [self._capture(run)]
        # N > 1: the launch as a list of device segments and collectives in execution order
        #   [D-input sums of the K rows -> all-reduce of the K float64 sum vectors]   (fused path)
        #   [n-step lookahead of the K rows]
        #   per row: [statistics -> all-reduce]                                      (layer-wise path)
        #            [targets + local return sums -> all-reduce]                     (PopArt)
        #            gradients (+ local clipped copies) -> all-reduce(grads) -> optimiser pass
        assert self.fused or self.wide or K == 1
        fused_stats = self.inline_stats or ((self.fused or self.wide) and not d.hps.rmsify_obs)
        items = []
        if self.inline_stats and any_d:
            items += [('dev', lambda: self._seg_sums(K)), ('ar', lambda: ar(self.addvecs[:K]))]
        items.append(('dev', lambda: self._seg_lookahead(K)))
        for k in range(K):
            role = roles[k]
            if not fused_stats:
                items.append(('dev', lambda k=k, role=role: self._seg_inputs(k, role)))
                if d.hps.rmsify_obs and role & self.D:
                    items.append(('ar', lambda: ar(d.obs_rms.addvec)))
            if self.popart_split and role & self.A:
                items += [('dev', lambda k=k, role=role: self._seg_targets(k, role, K=K)),
                          ('ar', lambda: ar(self.ret_addvec))]
            items += [('dev', lambda k=k, role=role: self._seg_grads(k, role, K=K)),
                      ('ar', lambda role=role: ar(self._grads_of(role))),
                      ('dev', lambda k=k, role=role: self._seg_apply(role, k, K))]
        if self.collectives_in_graph:   # collectives on the caller's stream: ONE graph for the launch
            return [self._capture(lambda: [f() for _, f in items])]
        # host-issued collectives: consecutive device segments share one graph (the optimiser pass of row
        # k and the gradient pass of row k + 1: per row the host issues ONE graph launch per collective)
        plan, run = [], []
        for kind, f in items + [('ar', None)]:
            if kind == 'dev':
                run.append(f)
                continue
            if run:
                plan.append(self._capture(lambda fs=tuple(run): [g() for g in fs]))
                run = []
            if f is not None:
                plan.append(f)
        return plan

    # ------------------------------------------------------------------ host side of a step
    def _roles(self, K, phase):
        return [self.sched[(phase + k) % self.S] for k in range(K)]

    def _draw(self, slot, K, phase):
        """The reference's host RNG draws for K consecutive rows, in the reference's order."""
        if slot.predrawn:   # the RNG part was done while the previous launch was running
            slot.predrawn = False
            self._fill_pending(slot, phase)
            return
        if slot.used:
            slot.event.synchronize()
        self._draw_rng(slot, K, phase)
        self._to_storage_rows(slot, K)
        self._fill_pending(slot, phase)

    def _to_storage_rows(self, slot, K):
        """Logical replay indices of K rows -> storage rows, in place (see _make_batch: storage_rows)."""
        rb = self.mu.replay_buffer
        if self.storage_rows:
            if rb.start:
                # (prioritized replay: only the discriminator's uniform draws are host-drawn indices)
                idx = slot.np_idx_p[:K] if self.per else slot.np_idx[:K]
                idx += rb.start
                idx[idx >= rb.limit] -= rb.limit
            self._write_geom(slot)

    def _write_geom(self, slot):
        """{start, num_entries} of the ring as this launch finds it, in the two words behind row 0's e_start (the PER
        uniforms' place, unused under uniform replay): device view self.geom."""
        rb = self.mu.replay_buffer
        assert not self.per or 2 <= rb.num_entries <= rb.st_cap, "prioritized sampling needs 2 <= num_entries <= capacity"
        slot.np_words[0, self.geom_at] = rb.start
        slot.np_words[0, self.geom_at + 1] = rb.num_entries

    def _fill_pending(self, slot, phase):
        if self.per and not self.per_dev and self.sched[phase] & self.A:
            if self._pending is None:
                self._pending = self._noop_update()
            slot.np_prev_idx[0], slot.np_prev_leaf[0] = self._pending
            self._pending = None

    def _expert_rows(self, slot, k):
        """demo_dataset.py:33-40: reshuffle when fewer than a batch is left, hand out the next slice."""
        if self.pool.pointer + self.B >= self.pool.num_entries:
            self.pool.init_pointer()
        slot.np_e[k] = self.pool.pointer
        self.pool.pointer += self.B

    def _draw_training_step(self, t_index):
        """Every host draw of ONE training step, in the reference's order: per D update the policy pairs
        (sam.py:372-377 -> memory.py:90; recent-window pairs: memory.py:95-111) and the expert batch
        (np.random.shuffle when the pool wraps, demo_dataset.py:33-40); the parameter-noise adaption with its own
        replay batch when due (sam.py:408-415); then the agent updates' batches (memory.py:90, or memory.py:278
        with `random` for PER).  Rows consume them as they are launched."""
        rb = self.mu.replay_buffer
        B, N = self.B, rb.num_entries
        c = {'d': [], 'a': []}
        for kind in self.dseq:
            idx = self._d_indices(kind, N)
            # (the reshuffle's np.random.shuffle draw is made here, at its place in the generator's stream; the
            # device-resident pool is permuted when the row that owns this batch is handed its words — the D rows
            # of this training step drawn before it are launched first and read the old order, as the reference's
            # loop does: demo_dataset.py:33-40 reshuffles after handing out batch j - 1)
            perm = False
            if self.pool.pointer + B >= self.pool.num_entries:
                perm = self.pool.shuffle_host()
            c['d'].append((idx, self.pool.pointer, perm))
            self.pool.pointer += B
        if self.pn_freq and t_index % self.pn_freq == 0:
            self._adapt_param_noise()
        for _ in range(self.g_steps):
            c['a'].append(rb.draw_uniforms(B) if self.per else np.random.randint(low=0, high=N, size=B))
        return c

    def _draw_rng(self, slot, K, phase, t_index=None):
        rb = self.mu.replay_buffer
        B, N = self.B, rb.num_entries
        roles = self._roles(K, phase)
        t_index = self.train_step_in_iter if t_index is None else t_index
        T = (phase + K + self.S - 1) // self.S
        plain = self.window is None and not (self.pn_freq and any((t_index + t) % self.pn_freq == 0
                                                                 for t in range(T)))
        if K == 1 or not plain or self.per:
            assert K == 1 or phase == 0
            ptr0 = self.pool.pointer
            for k in range(K):
                pos = (phase + k) % self.S
                if pos == 0 or self._cache is None:
                    self._cache = self._draw_training_step(t_index)
                    t_index += 1
                slot.np_e[k] = 0
                slot.np_idx_p[k] = 0
                if roles[k] & self.D:
                    slot.np_idx_p[k], slot.np_e[k], perm = self._cache['d'].pop(0)
                    if perm is not False:
                        assert k == 0, "a reshuffle of the expert pool cannot fall inside a multi-row launch"
                        self.pool.apply_permutation(perm)
                if roles[k] & self.A:
                    if self.per:
                        slot.np_u[k] = self._cache['a'].pop(0)
                    else:
                        slot.np_idx_a[k] = self._cache['a'].pop(0)
            assert K == 1 or self.pool.pointer >= ptr0, \
                "a reshuffle of the expert pool cannot fall inside a multi-row launch"
            return
        # whole training steps, no reshuffle inside (run() guarantees it) and uniform replay: per training
        # step the reference draws d_steps D batches, then g_steps agent batches, all from np.random —
        # one call of that total size consumes the generator exactly like the separate calls
        assert phase == 0 and K % self.S == 0 and not self.per
        T, d, g = K // self.S, self.nD, self.g_steps
        fast = self.host_rng and ops.host_rng_ready()   # the same draws at a quarter of numpy's time (include/samx.h)
        if self.S == 1 and fast:   # ... and straight into the pinned words
            ops.host_randint_rows(slot.np_words, K, 2 * B, N)
            draws = None
        elif fast:
            draws = np.empty((T, d + g, B), dtype=np.int64)
            ops.host_randint_rows(draws.reshape(T, (d + g) * B), T, (d + g) * B, N)
        else:
            draws = np.random.randint(low=0, high=N, size=(T, d + g, B))
        if self.S == 1:   # one D + agent row per training step: the draws ARE the rows' index words
            if draws is not None:
                slot.np_idx[:K] = draws
            if self.pool.pointer + K * B < self.pool.num_entries:   # no reshuffle inside (run() guarantees it for K > 1)
                slot.np_e[:K] = self.pool.pointer + B * np.arange(K, dtype=np.int64)
                self.pool.pointer += K * B
            else:
                for k in range(K):
                    self._expert_rows(slot, k)
            return
        k = 0
        for t in range(T):
            nd = na = 0
            for role in self.sched:
                slot.np_idx_p[k] = draws[t, nd] if role & self.D else 0
                slot.np_e[k] = 0
                if role & self.D:
                    assert self.pool.pointer + B < self.pool.num_entries, \
                        "a reshuffle of the expert pool cannot fall inside a multi-row launch"
                    self._expert_rows(slot, k)
                    nd += 1
                if role & self.A:
                    slot.np_idx_a[k] = draws[t, d + na]
                    na += 1
                k += 1

    # ---- the D updates' index source, the parameter-noise adaption
    def _d_indices(self, kind, N):
        if kind == 'u':
            return np.random.randint(low=0, high=N, size=self.B)
        # RB.sample_recent (memory.py:95-111): np.random.choice(idxs, size) = idxs[randint(0, len(idxs), size)]
        rb = self.mu.replay_buffer
        width = self.window if self.window <= N else N
        idxs = (rb.latest_entry_idx - np.arange(start=width - 1, stop=-1, step=-1) + rb.limit) % rb.limit
        return idxs.astype(np.int64)[np.random.randint(0, width, size=self.B)]

    def _pn_due(self, t_ahead):
        """Is the adaption due in the training step t_ahead steps after the one the next row belongs to?"""
        return bool(self.pn_freq) and (self.train_step_in_iter + t_ahead) % self.pn_freq == 0

    def _adapt_param_noise(self):
        """mu.adapt_param_noise at its place in the host RNG order (sam.py:408-415).  The device work is
        stream-ordered behind every earlier launch and ahead of the launch being drawn."""
        self.flush()   # the replay trees as the previous agent update left them (PER sampling reads them)
        self.mu.adapt_param_noise(self.comm)
        self.pn_log.append((self.mu.pn_dist, self.mu.pn_cur_std))

    def start_iteration(self):
        """The reference restarts `training_step` at 0 in every iteration (sam.py:323)."""
        assert self.step_no % self.S == 0, "an iteration starts at a training-step boundary"
        self.train_step_in_iter = 0

    def _noop_update(self):
        """A priority update that changes nothing (leaf 0 rewritten with its own value): what the
        first launch after a flush carries."""
        cap = self.mu.replay_buffer.st_cap
        leaf0 = float(self.mu.replay_buffer.trees.sum[cap].item())
        return np.zeros(self.B, dtype=np.int64), np.full(self.B, leaf0, dtype=np.float64)

    def flush(self):
        """Apply the priority update still waiting for the next launch (call before reading the trees
        or sampling from the replay buffer outside the driver)."""
        if self.per_dev:
            if self._dev_pending:
                rb = self.mu.replay_buffer
                ops.per_leaves_from_td(self.pend_td, self.pend_idx, rb.st_cap, rb.num_entries, rb.alpha, rb.num_demos,
                                       rb.demos_eps, self.leaf_buf, self.per_state)
                rb.trees.set(self.pend_idx, self.leaf_buf)
                self.pend_idx.fill_(-1)
                self._dev_pending = False
                # the reference's asserts and its max_priority, as the device tracked them (memory.py:371-379)
                mp, err = self.per_state.cpu().numpy().view(np.uint64)
                err = int(err)
                assert not err & 1, "priorities must be positive"
                assert not err & 2, "no element in buffer associated w/ index"
                rb.max_priority = max(rb.max_priority, float(np.uint64(mp).view(np.float64)))
            return
        if self._pending is not None:
            idxs, leaf = self._pending
            rb = self.mu.replay_buffer
            rb.trees.set(torch.as_tensor(idxs).to(self.dev), torch.as_tensor(leaf).to(self.dev))
            self._pending = None

    def _upload(self, slot, K):
        # ONE packed H2D copy per launch — by a kernel that reads the pinned words (the rows of a slot are contiguous),
        # so that nothing leaves the compute queue between two graph launches (SAMX_UPLOAD_KERNEL=0: hipMemcpyAsync)
        if self.upload_kernel:
            ops.copy_words(slot.words, self.words, K, self.upload_len, self.W)
        elif K == self.K:
            self.words.copy_(slot.words, non_blocking=True)
        else:
            self.words[:K].copy_(slot.words[:K], non_blocking=True)
        slot.event.record()
        slot.used = True

    def step_randoms(self, step):
        """The D randoms the kernels drew for reward-net update `step` (for checkers)."""
        return ops.disc_randoms(self.rng_seed, step, self.B, self.mu.ob_dim, self.mu.ac_dim,
                                want_real=self.label_mode >= 1, want_fake=self.label_mode == 2,
                                device=self.dev)

    def prepare(self, K=1, phase=0):
        """Build (and, with use_graph, capture) the device plan for launches of K rows starting at
        position `phase` of a training step.  Capturing executes the segments — learner state, running
        sums and host RNG positions are snapshotted and restored, so prepare() leaves the learner
        exactly as it found it."""
        if (K, phase) in self.plans:
            return
        mu, d, L = self.mu, self.d, self.L
        keep = [L.params, L.targets, L.m, L.v, L.steps]
        if d.hps.rmsify_obs:
            keep += [d.obs_rms.sums, d.obs_rms.mean, d.obs_rms.std]
        if mu.hps.rmsify_rets:
            keep += [mu.ret_rms.sums, mu.ret_rms.mean, mu.ret_rms.std]
        if self.per_dev:
            self.flush()   # the warm-up's first row then carries "nothing pending"
            keep += [self.per_state]
            if K > 1:      # ... its later rows apply the warm-up's own TD errors: the trees are put back
                keep += [mu.replay_buffer.trees.sum, mu.replay_buffer.trees.min]
        saved = [t.clone() for t in keep]
        rng_np, rng_py = np.random.get_state(), _random.getstate()
        ptr0 = self.pool.pointer
        slot = self._warm_slot
        # warm-up indices: drawn without touching the expert-pool order
        if slot.used:
            slot.event.synchronize()
        slot.np_idx[:K] = np.random.randint(low=0, high=mu.replay_buffer.num_entries, size=(K, 2, self.B))
        slot.np_e[:K] = 0
        if self.storage_rows:
            self._write_geom(slot)
        if self.per:
            for k in range(K):
                slot.np_u[k] = mu.replay_buffer.draw_uniforms(self.B)
            if not self.per_dev:
                self.flush()
                slot.np_prev_idx[0], slot.np_prev_leaf[0] = self._noop_update()
        self._upload(slot, K)
        torch.cuda.synchronize()   # nothing in flight reads _launch_words any more
        self._launch_busy = False
        np.copyto(self._launch_np, slot.np_words[:1])
        self.plans[(K, phase)] = self._build_plan(K, phase)
        torch.cuda.synchronize()
        for t, sv in zip(keep, saved):
            t.copy_(sv)
        if self.per_dev:
            self.pend_idx.fill_(-1)   # the warm-up rows' TD errors are not an update of anything
        np.random.set_state(rng_np)
        _random.setstate(rng_py)
        self.pool.pointer = ptr0
        torch.cuda.synchronize()

    def prepare_all(self):
        """Capture every launch run() can pick: K, K/2, ... whole training steps, and single rows at
        every position of a training step."""
        K = self.K
        while K > 1 and K % self.S == 0:
            self.prepare(K)
            K //= 2
        for phase in range(self.S):
            self.prepare(1, phase)
        for K in self._recurring_sizes():
            self.prepare(K)

    def _recurring_sizes(self):
        """Launch sizes below K that run() meets again and again: the training steps left of an expert-pool epoch
        after its full-size launches (an epoch starts with the training step that reshuffles), and of a
        parameter-noise adaption period."""
        S, Kt = self.S, self.K // self.S
        out = set()
        if Kt > 1:
            per_step = self.nD * self.B
            fit = (self.pool.num_entries - 1 - per_step) // per_step   # training steps after the reshuffling one
            if fit % Kt:
                out.add((fit % Kt) * S)
            if self.pn_freq and self.pn_freq % Kt:
                out.add((self.pn_freq % Kt) * S)
        return sorted(k for k in out if k > 1 and (k, 0) not in self.plans)

    @property
    def plan(self):
        return self.plans.get((1, 0))

    def _launch(self, K, more=False):
        phase = self.step_no % self.S
        role = self.sched[phase]
        rb = self.mu.replay_buffer
        if not self.storage_rows and (rb.start != self._batch_start or rb.num_entries != self._batch_entries):
            # transitions were appended since the plans were captured (the ring start and the entry
            # count are baked into them): re-read the geometry and capture again
            self.refresh()
        # (uniform replay without n-step returns: the entry count only bounds the HOST's index draws, which read
        # it live — the captured graphs survive appends until the ring wraps and its start moves)
        self._batch_entries = rb.num_entries
        self.prepare(K, phase)
        slot = self.slots[self.launch_no % len(self.slots)]
        self._draw(slot, K, phase)
        if self.per_graph_io:
            if self._launch_busy:   # the previous launch may still be reading the buffer (D-only rows)
                self._launch_event.synchronize()
            np.copyto(self._launch_np, slot.np_words[:1])   # host memcpy pinned -> pinned; the graph uploads it
        else:
            self._upload(slot, K)
        for run in self.plans[(K, phase)]:
            run()
        if self.per_graph_io:
            self._launch_event.record()
            self._launch_busy = True
        if self.per_dev:
            if any(r & self.A for r in self._roles(K, phase)):
                self._dev_pending = True
        elif self.per:
            refresh = bool(role & self.A)
            if refresh and not self.per_graph_io:
                self._td_host.copy_(self.td_post, non_blocking=True)
                self._idx_host.copy_(self.idx_a, non_blocking=True)
            # the next row's host RNG draws do not depend on this row: make them now, while the GPU works
            # (same generators, same order) — unless a parameter-noise adaption (device work that must see
            # this row's update) comes with them
            nphase = (phase + 1) % self.S
            t_next = self.train_step_in_iter + (phase + 1) // self.S
            adapt_next = bool(self.pn_freq) and nphase == 0 and t_next % self.pn_freq == 0
            if more and not adapt_next:
                nxt = self.slots[(self.launch_no + 1) % len(self.slots)]
                if nxt.used:
                    nxt.event.synchronize()
                self._draw_rng(nxt, 1, nphase, t_next)
                nxt.predrawn = True
            if refresh:
                torch.cuda.current_stream().synchronize()
                # |td| + 1e-6 in float32 (sam_agent.py:702), the reference's asserts, leaves = libm pow on
                # the host (one C call); the trees are updated on the device by the next agent row
                idxs = self._idx_host_np.copy()
                leaf = self.mu.replay_buffer.leaves_from_td(idxs, self._td_host_np, self._leaf_np).copy()
                self._pending = (idxs, leaf)
        self.step_no += K
        self.launch_no += 1
        self.train_step_in_iter += (phase + K) // self.S

    def step(self):
        """One row (with d_steps = g_steps = 1: one update step); the replay trees are up to date when
        it returns."""
        self._launch(1)
        self.flush()

    def run(self, n):
        """n consecutive rows, whole training steps packed K per graph launch where the configuration
        allows it (see __init__); same arithmetic and same sampled indices as n calls of step()."""
        B, pool, S = self.B, self.pool, self.S
        while n > 0:
            # the largest launch (K, K/2, ... whole training steps) that fits the rows left and ends before
            # the next reshuffle of the expert pool (host work between two rows, kept out of a launch)
            K = self.K if self.step_no % S == 0 else 1
            if K > 1:
                # whole training steps that fit: the rows left, the expert pool before its next reshuffle, the
                # training steps before the next due parameter-noise adaption (which heads its own launch)
                T = min(K // S, n // S, max(0, (pool.num_entries - 1 - pool.pointer) // (self.nD * B)))
                t_pn = next((t for t in range(1, T) if self._pn_due(t)), T)
                T = min(T, t_pn)
                exact = T * S
                # the sizes that recur — what is left of an expert-pool epoch, of a parameter-noise period — have
                # captured plans of their own (prepare_all): 32, 32, 32, 27 instead of 32, 32, 32, 16, 8, 2, 1.
                # Sizes without a plan fall back to K/2, K/4, ...
                if exact >= 1 and (exact == self.K or (exact, 0) in self.plans):
                    K = exact
                else:
                    while K > 1 and (K > n or K % S or
                                     pool.pointer + (K // S) * self.nD * B >= pool.num_entries or
                                     any(self._pn_due(t) for t in range(1, K // S))):
                        K = K // 2 if (K // 2) % S == 0 else 1
            self._launch(K, more=n > K)
            n -= K
        self.flush()

    def set_collectives(self, in_graph):
        """Choose between collectives captured inside the K-row graph (needs a capturable
        communicator) and collectives issued by the host between graph segments; re-plans."""
        in_graph = bool(in_graph and self.world > 1 and (self.fused or self.wide) and self.use_graph and
                        getattr(self.comm, 'capturable', False))
        if in_graph != self.collectives_in_graph:
            self.collectives_in_graph = in_graph
            self.plans = {}
        return in_graph

    def refresh(self):
        """Re-read the replay ring geometry (after rollouts appended transitions) and re-capture."""
        self.flush()
        self.plans = {}
        self._make_batch()

    def outputs(self):
        """What the reference's loop logs per step (sam.py:398-431), still in HBM."""
        L, t = self.L, self.out.t
        return dict(d_losses=t['d_losses'], actor_losses=t['actor_losses'],
                    critic_losses=t['critic_losses'] if self.mu.hps.n_step_returns else t['critic_losses'][:3],
                    actor_grads=L.seg(L.clipped, 0),
                    critic_grads=L.seg(L.clipped, 1), d_grads=L.seg(L.clipped, 2))
