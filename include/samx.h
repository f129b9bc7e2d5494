/* samx.h — C ABI of libsamx: the MI355X (gfx950) implementation of SAM's replayed-minibatch
 * update step (lionelblonde/sam-tf).
 *
 * The reference has no FFI: its boundary for this path is the Python object protocol that
 * imitation/imitation_algorithms/sam.py uses.  libsamx is what the Python mirror
 * of that protocol (sam_tf_amd/{agent,discriminator,memory}.py) binds through ctypes; each entry
 * point below cites the reference code it replaces (paths relative to /root/reference/imitation).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every pointer is DEVICE memory unless its name ends in
 *    `_host`.  `stream` is a hipStream_t passed as void*; calls only enqueue work on it.
 *  - Return 0 on success, <0 on error (SAMX_ERR_*); samx_last_error() returns the message of the
 *    calling thread's last failure.  Nothing throws across the ABI.
 *  - Matrices are row-major float32.  Flat parameter / gradient vectors use the reference's variable
 *    creation order (networks.py:50-70,359-404 ; tf_util.py:137-151):
 *       W1[in,h1] b1[h1] (beta1[h1] gamma1[h1]) W2[h1,h2] b2[h2] (beta2 gamma2) W3[h2,out] b3[out]
 *  - Calls that need scratch take `ws`/`ws_bytes`; ask samx_*_ws_bytes() for the size.  The library
 *    itself allocates nothing on these paths.
 */
#ifndef SAMX_H
#define SAMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAMX_OK 0
#define SAMX_ERR_ARG (-1)
#define SAMX_ERR_HIP (-2)
#define SAMX_ERR_WORKSPACE (-3)
#define SAMX_ERR_UNSUPPORTED (-4)

const char* samx_last_error(void);
int samx_version(void);
/* Wait until everything enqueued on `stream` has completed (hipStreamSynchronize).  The one host wait of an
 * object-mirror call that returns host values (SAMAgent.train, Discriminator.get_grads / get_losses,
 * RB.sample: the reference's sess.run blocks the same way, helpers/tf_util.py:11-28). */
int samx_stream_synchronize(void* stream);

/* ---- network description ------------------------------------------------------------------ */
/* Two hidden layers, leaky-ReLU(0.2), optional post-activation LayerNorm (eps 1e-8):
 * helpers/networks.py:45-72 (stack_fc_layers), helpers/network_util.py:5-72.
 * actor: in=ob, out=ac (networks.py:262-323); critic / reward net: in=ob+ac (action concatenated at
 * the first layer, ac_branch_in=1), out=1 (networks.py:222-259,337-404). */
typedef struct samx_mlp {
    int32_t in_dim, h1, h2, out_dim, layernorm;
} samx_mlp;

int64_t samx_mlp_param_count(const samx_mlp* d);

/* ---- segment trees / prioritized replay (bit-exact, float64) ------------------------------- */
/* Trees are device arrays of 2*cap doubles, root at [1], leaves at [cap..2cap) — the layout of
 * helpers/segment_tree.py:22-23.  `stamp` is an int32[cap] scratch that must hold -1 everywhere
 * between calls (samx_tree_init sets it). */

/* SegmentTree.__init__ (segment_tree.py:6-24): neutral elements 0.0 / +inf. */
int samx_tree_init(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap, void* stream);

/* A batch of SegmentTree.__setitem__ (segment_tree.py:55-64) applied to BOTH trees in index order:
 * duplicates resolve as in the reference's sequential loop (last writer wins,
 * imitation_algorithms/memory.py:370-379).  idxs int64[n], leaf float64[n]. */
int samx_tree_set(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                  const int64_t* idxs, const double* leaf, int64_t n, void* stream);

/* SegmentTree.reduce (segment_tree.py:26-53) in the reference's recursive association order.
 * kind 0 = sum, 1 = min.  `end` exclusive; end < 0 wraps by +cap; pass cap for None. out: double[1]. */
int samx_tree_reduce(const double* tree, int64_t cap, int kind, int64_t start, int64_t end,
                     double* out, void* stream);

/* SumSegmentTree.find_prefixsum_idx (segment_tree.py:82-97) for n prefix sums. */
int samx_tree_find_prefixsum(const double* sum_tree, int64_t cap, const double* prefix, int64_t n,
                             int64_t* idx_out, void* stream);

/* PrioritizedRB._sample_w_priorities + the weight loop of _sample (memory.py:248-309).
 * u: the B unit uniforms the reference draws with random.random() (host generates them with the
 * same generator and copies them in).  idxs_out int64[B]; iws_out float64[B] (and, if not NULL,
 * iws_f32_out float32[B] — what the critic loss is fed).  Requires num_entries >= 2.
 * Indices, tree sums AND the importance weights are bit-exact: the two pows of memory.py:301,308 go through
 * the device routine that follows the host pow bit for bit (samx_pow_device below). */
int samx_per_sample(const double* sum_tree, const double* min_tree, int64_t cap,
                    int64_t num_entries, const double* u, int64_t B, double beta,
                    int64_t* idxs_out, double* iws_out, float* iws_f32_out, void* stream);

/* HOST helper: leaf values `priority ** alpha` with the host libm pow — the same routine CPython's
 * float.__pow__ reaches in memory.py:341-342,376-377 — so leaves are bit-identical to the
 * reference's.  Adds demos_eps to entries whose index is < num_demos (memory.py:373-375), returns
 * through *max_priority_host the running max(priority) (memory.py:379).  All pointers HOST. */
int samx_per_leaf_values_host(const int64_t* idxs_host, const double* prios_host, int64_t n,
                              double alpha, int64_t num_demos, double demos_eps,
                              double* leaf_out_host, double* max_priority_host);

/* The PER refresh of SAMAgent.train() in one host call (sam_agent.py:693-703 -> memory.py:344-379):
 * priority = |td| + 1e-6 in float32 (as numpy computes it), the reference's asserts (positive priority,
 * index < num_entries), demos_eps, max_priority tracking and leaf = priority ** alpha with the host libm
 * pow.  Host pointers; the leaves then go to samx_tree_set. */
int samx_per_leaves_from_td_host(const float* td_host, const int64_t* idxs_host, int64_t n,
                                 int64_t num_entries, double alpha, int64_t num_demos, double demos_eps,
                                 double* leaf_out_host, double* max_priority_host);

/* The same refresh ON THE DEVICE (no host round trip between two prioritized update steps): td float32[n]
 * and idxs int64[n] in HBM -> leaf_out float64[n]; entries whose index lies outside [0, cap) mean "nothing
 * pending" and are skipped.  per_state: 16 bytes in HBM, {double max_priority; uint64 error bits}: the running
 * max(priority) (memory.py:379) and the reference's two asserts (bit 0: a priority that is not positive,
 * memory.py:371; bit 1: index >= num_entries, memory.py:372) which the host reads when it next needs them.
 * leaf = priority ** alpha with the host pow's bits (samx_pow_device). */
int samx_per_leaves_from_td(const float* td, const int64_t* idxs, int64_t n, int64_t cap, int64_t num_entries,
                            double alpha, int64_t num_demos, double demos_eps, double* leaf_out,
                            void* per_state, void* stream);

/* pow(x[i], y) with the bits of the HOST C library's pow (glibc's table-driven routine, FMA build; constants
 * taken from the image's libm by tools/extract_pow_tables.py), computed on the device: what lets
 * `priority ** alpha` (memory.py:341-342,376-377) and the importance weights (memory.py:301,308) stay in HBM.
 * Bit-for-bit domain: x positive and normal, 2^-65 <= |y| < 2^63, |y log x| < 512; outside it the platform pow
 * answers.  tests/test_gpu_pow.py compares >= 2^26 float32-born bases per exponent against libm.
 * samx_pow_device_f32_range / samx_pow_host_f32_range: the same for the n float32 values whose bit patterns
 * start at first_bits, widened to double, + bonus (device result in HBM / libm result in host memory).
 * samx_pow_selfcheck(n): HOST — the number of n pseudo-random cases in which the host build of the routine
 * differs from this machine's libm pow (0 expected; non-zero = another libm code path, keep the host refresh). */
int samx_pow_device(const double* x, double y, int64_t n, double* out, void* stream);
int samx_pow_device_f32_range(uint32_t first_bits, int64_t n, double bonus, double y, double* out, void* stream);
int samx_pow_host_f32_range(uint32_t first_bits, int64_t n, double bonus, double y, double* out_host);
int samx_pow_host(const double* x_host, double y, int64_t n, double* out_host);   /* libm pow, HOST pointers */
int64_t samx_pow_selfcheck(int64_t n);

/* HOST: np.random.randint(low=0, high=high, size=(rows, row_len)) of numpy's legacy global generator — the replay
 * index draws of RB.sample / sample_uniform (memory.py:90, called per batch from sam.py:372,424) for all rows of a
 * multi-row launch at once — written as int64 into out[r * row_stride + c] (e.g. straight into pinned staging).
 * mt_state = numpy's own MT19937 state (np.random.mtrand._rand._bit_generator.ctypes.state_address), advanced in
 * place exactly as numpy's call would advance it (masked rejection on 32-bit words; 1 <= high <= 2^32); the caller
 * holds no other reference to the generator meanwhile.  tests/test_host_rng.py compares it with numpy draw for draw,
 * interleaved with numpy's own calls; sam_tf_amd/ops.py checks it once per process before using it. */
int samx_host_mt19937_randint(void* mt_state, uint64_t high, int64_t rows, int64_t row_len, int64_t* out,
                              int64_t row_stride);

/* ---- replay storage ------------------------------------------------------------------------ */
/* The first row_len int64 words of each of `rows` rows (row_stride words apart, the same layout on both sides) from
 * (pinned, device-readable) host memory or HBM to HBM by a kernel on `stream`: the upload of a launch's host-drawn
 * index words (the reference passes them as feed_dict entries, sam.py:372-377,424) without a hand-over to the copy
 * engine between two graph launches, and without the words of a row the configuration does not use.  Both buffers
 * 16-byte aligned. */
int samx_copy_words(const int64_t* src, int64_t* dst, int64_t rows, int64_t row_len, int64_t row_stride, void* stream);

/* RingBuffer.get_batch for the five atoms at once (memory.py:33-34,74-84): row = (start+idx)%maxlen.
 * Any output may be NULL. */
int samx_replay_gather(const float* obs0, const float* acs, const float* rews, const float* dones1,
                       const float* obs1, int64_t maxlen, int64_t start, int ob_dim, int ac_dim,
                       const int64_t* idxs, int64_t B, float* o_obs0, float* o_acs, float* o_rews,
                       float* o_dones1, float* o_obs1, void* stream);

/* RB.lookahead (memory.py:113-160): n-step TD lookahead from B sampled LOGICAL indices.  Per index the
 * window [idx, min(idx+n, num_entries)-1] is cut after the first dones1 == 1; rews_out = discounted
 * sum of the kept rewards accumulated in float64 from the far end (helpers/math_util.py:32-43), then
 * rounded to float32 as the TF placeholder does (rews64_out, may be NULL, keeps the float64);
 * end_idx_out = logical index of the last kept row (whose obs1 is the n-step next observation);
 * dones_out = 1 only if the cut came before the window's end; td_len_out = rows kept. */
int samx_nstep_lookahead(const float* rews, const float* dones1, int64_t maxlen, int64_t start,
                         int64_t num_entries, const int64_t* idxs, int64_t B, int n, double gamma,
                         int64_t* end_idx_out, float* rews_out, double* rews64_out, float* dones_out,
                         float* td_len_out, void* stream);
/* The same for K batches in one launch: batch k's B indices start at idxs + k * idx_stride; outputs are
 * dense [K*B].  The lookahead reads only the replay storage, so the K batches of a multi-row launch
 * (uniform replay: all indices known up front) can be done ahead and handed to samx_step_grads through
 * samx_batch.n_end / n_rews / n_dones / n_len. */
int samx_nstep_lookahead_many(const float* rews, const float* dones1, int64_t maxlen, int64_t start,
                              int64_t num_entries, const int64_t* idxs, int64_t idx_stride, int64_t K,
                              int64_t B, int n, double gamma, int64_t* end_idx_out, float* rews_out,
                              float* dones_out, float* td_len_out, void* stream);

/* samx_nstep_lookahead_many on STORAGE rows: rows[k * row_stride + i] = (start + idx) % maxlen of the reference's
 * logical index (memory.py:16-35), geom = int64[2] {start, num_entries} IN DEVICE MEMORY, end_row_out = storage row of
 * the last kept transition.  The ring's geometry changes with every append (memory.py:36-57); a caller that replays a
 * captured graph uploads it with the rows instead of re-capturing (UpdateStepDriver under uniform replay).  Same
 * arithmetic as samx_nstep_lookahead (memory.py:113-160, math_util.py:32-43). */
int samx_nstep_lookahead_many_rows(const float* rews, const float* dones1, int64_t maxlen, const int64_t* geom,
                                   const int64_t* rows, int64_t row_stride, int64_t K, int64_t B, int n, double gamma,
                                   int64_t* end_row_out, float* rews_out, float* dones_out, float* td_len_out,
                                   void* stream);

/* RingBuffer.append of one staged transition (memory.py:36-57,177-184): `row` is the physical
 * slot the host ring arithmetic chose; packed = [obs0(ob) | acs(ac) | rew | obs1(ob) | done]. */
int samx_replay_write(float* obs0, float* acs, float* rews, float* dones1, float* obs1,
                      int ob_dim, int ac_dim, const float* packed, const int64_t* rows, int64_t n,
                      void* stream);

/* ---- running mean / std (helpers/mpi_running_mean_std.py) ---------------------------------- */
/* update() lines 47-53: addvec = [sum_x(dim), sum_x^2(dim), count] in float64 over the rows of x1
 * (n1 rows) followed by x2 (n2 rows, may be NULL/0).  The all-reduce of line 55 is the caller's
 * (RCCL) between this call and samx_rms_commit. */
int samx_rms_batch_sums(const float* x1, int64_t n1, const float* x2, int64_t n2, int dim,
                        double* addvec, void* stream);
/* lines 40-45 + 31-34: sums += addvec ; mean = f32(sum/count) ;
 * std = sqrt(max(f32(sumsq/count) - mean^2, 1e-2)). sums: double[2*dim+1]. */
int samx_rms_commit(double* sums, const double* addvec, int dim, float* mean_out, float* std_out,
                    void* stream);

/* ---- optimiser ----------------------------------------------------------------------------- */
/* MpiAdamOptimizer step (helpers/mpi_adam.py:18-36 + tf.train.AdamOptimizer): grad is the SUM over
 * workers of the local flat gradients (after the RCCL all-reduce; grad_scale = 1/world gives the
 * mean of line 27), then clip_by_global_norm (lines 31-33; clip_norm <= 0 disables), then TF1 Adam
 * with epsilon outside the square root; `step` is the 1-based update count.  If targ != NULL the
 * Polyak update of sam_agent.py:31-34 (targ = (1-tau) targ + tau param) is applied to the fresh
 * parameters in the same pass.  partials: float[samx_adam_partials()] scratch.
 * step_dev (may be NULL): an int64 update counter in device memory.  When given, the library
 * increments it and derives the bias correction from it on the device, and `step` is ignored — a
 * captured HIP graph can then be replayed without a per-step host argument. */
int samx_adam_partials(void);
int samx_clip_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                   float beta1, float beta2, float eps, int64_t step, float clip_norm,
                   float grad_scale, float* targ, float tau, float* partials, int64_t* step_dev,
                   void* stream);

/* get_target_updates (sam_agent.py:20-37): tau = 1 is the hard copy. */
int samx_polyak(float* targ, const float* param, int64_t n, float tau, void* stream);

/* flatgrad(..., clip_norm) (helpers/tf_util.py:137-151): out = grad * clip/max(|grad|, clip) — the
 * locally clipped gradient SAMAgent.train returns for logging (sam_agent.py:433,520,706-711). */
int samx_clip_copy(const float* grad, int64_t n, float clip_norm, float* out, float* partials,
                   void* stream);

/* check_synced (helpers/mpi_adam.py:55-72): *mismatch_out (int32) = number of differing words. */
int samx_count_mismatch(const float* a, const float* b, int64_t n, int32_t* mismatch_out,
                        void* stream);

/* Parameter-space noise (sam_agent.py:40-58,352-404): out = params + N(0, stddev^2) on the dense kernels
 * and biases, out = params on the LayerNorm variables (networks.py:325-334).  Draws are Philox /
 * Box-Muller keyed by (seed, counter, element index): the same (seed, counter) gives the same copy. */
int samx_perturb_params(const samx_mlp* d, const float* params, float* out, float stddev, uint64_t seed,
                        uint64_t counter, void* stream);
/* sqrt(mean((a - b)^2)) over n floats (the action-space distance of sam_agent.py:371); out float[1]. */
int samx_rms_distance(const float* a, const float* b, int64_t n, float* out, void* stream);

/* The fp32 MFMA GEMM every layer of the layer-wise path is made of, exposed for parity tests and
 * integration: C[i,j] = epi( sum_r A(i,r) B(r,j) ), A(i,r) = A[i*a_si + r*a_sr], B(r,j) = B[r*b_sr + j*b_sj],
 * one unit stride per operand.  Replaces tf.layers.dense / tf.matmul and their gradients
 * (networks.py:45-72,287,365,390; tf.gradients at tf_util.py:141).  The accumulation is the k-ordered fp32
 * fma chain of v_mfma_f32_32x32x2_f32.  epi: 0 none, 1 +bias[j], 2 lrelu(+bias), 3 alpha*tanh(+bias) clipped,
 * 4 * lrelu'(aux[i,j]), 5 (acc + add[i,j]) * tanh'(aux), 6 acc + alpha*aux[i,j].  a_ones_rows >= 0 appends a
 * virtual row M to A with A(M,r) = (r < a_ones_rows) (bias gradient); nsplit > 1 splits r over
 * workgroups through partials[nsplit][M(+1)][N] with a fixed-order reduce.  overread_ok != 0: the caller
 * guarantees 16 readable bytes behind the last element of A and of B, which lets operands whose contiguous
 * extent is not a multiple of 4 use the 16-byte staging path too (the values read there are never used). */
int samx_gemm_f32(const float* A, int64_t a_si, int64_t a_sr, const float* B, int64_t b_sr, int64_t b_sj,
                  float* C, int64_t ldc, int32_t M, int32_t N, int32_t R, int32_t a_ones_rows, int32_t epi,
                  const float* bias, const float* aux, int64_t ld_aux, const float* add, int64_t ld_add,
                  float alpha, int32_t nsplit, float* partials, int32_t overread_ok, void* stream);

/* The split weight-gradient product of the 256-wide path on its own, for parity tests: partials[z][i][j] =
 * sum over the rows r of chunk z (gemm_split_chunk(R, nsplit) = ceil(R / nsplit) rounded up to 32 rows each) of
 * A[r*lda + i] * B[r*ldb + j], i < M, j < N, plus row i == M = sum of B[r*ldb + j] over r < a_ones_rows when
 * a_ones_rows >= 0 (the bias gradient) — dW, db = X^T dZ of tf.gradients (tf_util.py:141) before the fixed-order sum
 * over the chunks.  kernel 1: the register-direct kernel (operands straight from L2 into MFMA registers; needs R even
 * and readable bytes behind odd extents as for samx_gemm_f32, else SAMX_ERR_UNSUPPORTED); kernel 0: the staged
 * (LDS ring) kernel it replaced.  partials: nsplit * (M + (a_ones_rows >= 0)) * N floats. */
int samx_wgrad_partials_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* partials, int32_t M,
                            int32_t N, int32_t R, int32_t a_ones_rows, int32_t nsplit, int32_t kernel,
                            int32_t overread_ok, void* stream);

/* ---- network passes ------------------------------------------------------------------------ */
/* ob_mean/ob_std: float[ob] from samx_rms_commit, or NULL when rmsify_obs is off
 * (sam_agent.py:150-159).  Observations are normalised and clipped to [-5, 5] inside.
 * Calls that take only a critic-shaped network get `ob_dim` (the action width is in_dim - ob_dim). */

/* Return normalisation (rmsify_rets, sam_agent.py:161-201 ; helpers/math_util.py:8-19): device scalars
 * ret_rms.mean / ret_rms.std (float[1] each).  Every call below takes `ret` (NULL = off): critic
 * outputs are DE-normalised (q * std + mean, lines 180-201), critic targets are normalised
 * ((t - mean) / std, lines 161-172) — including the reference's mixed-scale TD error. */
typedef struct samx_retnorm {
    const float* mean;
    const float* std;
} samx_retnorm;

/* Scratch bytes that cover every agent-side call below for batch B (reward_net may be NULL). */
size_t samx_nets_ws_bytes(const samx_mlp* actor, const samx_mlp* critic, const samx_mlp* reward_net,
                          int64_t B);

/* SAMAgent.act_q (sam_agent.py:177,186,209-212 ; predict at 592-619): ac = clip(high*tanh(actor)),
 * q = critic(obz, ac).  ac_out float[B,ac]; q_out float[B] or NULL. */
int samx_act_q(const samx_mlp* actor, const float* actor_params, const samx_mlp* critic,
               const float* critic_params, const float* obs, const float* ob_mean,
               const float* ob_std, int64_t B, float ac_high, float* ac_out, float* q_out,
               const samx_retnorm* ret, void* ws, size_t ws_bytes, void* stream);

/* get_targ_q (sam_agent.py:196-205,225-226): rews + gamma^td_len * (1-dones1) * Q'(obz1, mu'(obz1)).
 * td_len NULL means 1 for every row (sam_agent.py:648-654). out float[B]. */
int samx_targ_q(const samx_mlp* actor, const float* targ_actor_params, const samx_mlp* critic,
                const float* targ_critic_params, const float* obs1, const float* rews,
                const float* dones1, const float* td_len, const float* ob_mean,
                const float* ob_std, int64_t B, float gamma, float ac_high, float* out,
                const samx_retnorm* ret, void* ws, size_t ws_bytes, void* stream);

/* setup_critic (sam_agent.py:457-520): td = Q(obz0, acs) - tc1s ; loss = mean(huber(td) * iws) *
 * td_scale + wd_scale * (|W1|^2 + |W2|^2)/2.  iws NULL for uniform replay.
 * tcns != NULL adds the n-step term mean(huber(Q - tcns) * iws) * tdn_scale (lines 483-498).
 * losses_out float[4] = {critic_td_loss_1, critic_wd, critic_total_loss, critic_td_loss_n (0 without
 * tcns)}; grad_out float[P] the unclipped flat gradient; td_out / tdn_out float[B] or NULL. */
int samx_critic_grad(const samx_mlp* critic, const float* critic_params, const float* obs0,
                     const float* acs, const float* iws, const float* tc1s, const float* ob_mean,
                     const float* ob_std, int64_t B, int ob_dim, float td_scale, float wd_scale,
                     float* losses_out, float* grad_out, float* td_out, const samx_retnorm* ret,
                     const float* tcns, float tdn_scale, float* tdn_out, void* ws, size_t ws_bytes,
                     void* stream);

/* critic_ops['td_errors'] (sam_agent.py:534-538,693-701): Q(obz0, acs) - tc1s, float[B]. */
int samx_td_errors(const samx_mlp* critic, const float* critic_params, const float* obs0,
                   const float* acs, const float* tc1s, const float* ob_mean, const float* ob_std,
                   int64_t B, int ob_dim, float* td_out, const samx_retnorm* ret, void* ws,
                   size_t ws_bytes, void* stream);

/* The same TD errors for rows addressed IN PLACE in the replay storages: row r of the batch is storage row
 * (ring_start + idx[r]) % ring_len (idx NULL: obs0/acs are B dense rows).  64x64 critics without return
 * normalisation run as one whole-network launch; anything else gathers into ws and takes the layer-wise
 * pass.  This is the PER priority refresh with the post-update critic (sam_agent.py:693-701).
 * td_copy_out / idx_copy_out (both or neither; need idx): second copies of td_out and of idx, written by the
 * same launch — pointed at pinned host memory they hand the host what memory.py:344-379 needs for the next
 * priority update without a separate device-to-host step. */
size_t samx_td_errors_rows_ws_bytes(const samx_mlp* critic, int64_t B);
int samx_td_errors_rows(const samx_mlp* critic, const float* critic_params, const float* obs0,
                        const float* acs, const int64_t* idx, int64_t ring_start, int64_t ring_len,
                        const float* tc1s, const float* ob_mean, const float* ob_std, int64_t B, int ob_dim,
                        float* td_out, float* td_copy_out, int64_t* idx_copy_out, const samx_retnorm* ret,
                        void* ws, size_t ws_bytes, void* stream);

/* setup_actor (sam_agent.py:406-433): loss = -q_scale*mean Q(obz0, mu(obz0))
 *                                           -d_scale*mean D(obz0, mu(obz0)), gradient wrt the actor only.
 * reward_params may be NULL when d_scale == 0.  losses_out float[3] = {actor_q_loss, actor_d_loss,
 * actor_total_loss}; grad_out float[P_actor]. */
int samx_actor_grad(const samx_mlp* actor, const float* actor_params, const samx_mlp* critic,
                    const float* critic_params, const samx_mlp* reward_net,
                    const float* reward_params, const float* obs0, const float* ob_mean,
                    const float* ob_std, int64_t B, float ac_high, float q_scale, float d_scale,
                    float* losses_out, float* grad_out, const samx_retnorm* ret, void* ws,
                    size_t ws_bytes, void* stream);

/* ret_rms.update + PopArt (sam_agent.py:554-590,662-672 ; helpers/mpi_running_mean_std.py:40-45).
 * addvec1 (and addvec2, may be NULL: the n-step targets) are float64 [sum, sumsq, count] of a batch of
 * targets from samx_rms_batch_sums (dim 1), already all-reduced.  In one launch: remember the old
 * mean/std, sums += addvec*, refresh ret_mean / ret_std, and with enable_popart rescale the output
 * layer of the online and the target critic: w <- w*old_std/new_std,
 * b <- (b*old_std + old_mean - new_mean)/new_std. */
int samx_ret_update_popart(double* sums, const double* addvec1, const double* addvec2, float* ret_mean,
                           float* ret_std, int enable_popart, const samx_mlp* critic,
                           float* critic_params, float* targ_critic_params, void* stream);

/* Discriminator losses + gradient (imitation_algorithms/discriminator.py:77-161): BCE(policy,
 * fake) + BCE(expert, real) - ent_reg_scale*mean entropy + 10 * gradient penalty on the
 * interpolates eps*p + (1-eps)*e.  The tensors TF draws at run time are inputs here:
 * eps_ob float[B,ob], eps_ac float[B,ac] ~ U[0,1) (lines 129-134); real_labels / fake_labels float[B]
 * (lines 89-109; NULL means 1 / 0).  losses_out float[8] in the order of lines 145-158;
 * grad_out float[P_d] unclipped. */
size_t samx_disc_ws_bytes(const samx_mlp* reward_net, int64_t B);
int samx_disc_grad(const samx_mlp* reward_net, const float* params, const float* p_obs,
                   const float* p_acs, const float* e_obs, const float* e_acs,
                   const float* eps_ob, const float* eps_ac, const float* real_labels,
                   const float* fake_labels, const float* ob_mean, const float* ob_std, int64_t B,
                   int ob_dim, float ent_reg_scale, float* losses_out, float* grad_out, void* ws,
                   size_t ws_bytes, void* stream);

/* compute_reward (discriminator.py:180-194), batched: -log(1 - sigmoid(s) + 1e-8), or
 * log_sigmoid(s) when non_satur_grad.  out float[B]. */
int samx_reward(const samx_mlp* reward_net, const float* params, const float* obs, const float* acs,
                const float* ob_mean, const float* ob_std, int64_t B, int ob_dim, int non_satur_grad,
                float* out,
                void* ws, size_t ws_bytes, void* stream);

/* ---- one update step for the whole learner ------------------------------------------------ */
/* The body of the reference's training loop (imitation_algorithms/sam.py:369-433, d_steps = g_steps
 * = 1) as two calls: gradients, then the optimiser pass; the caller's gradient all-reduce
 * (helpers/mpi_adam.py:27) goes between them.  The learner is described by plain device pointers;
 * vectors of all three optimisers are laid out [actor | critic | reward-net] so that ONE all-reduce
 * covers an update.  For 64x64 networks and batches that are a multiple of 16 the whole step runs on
 * whole-network fused kernels (weights + activations in LDS), otherwise on the layer-wise MFMA path;
 * results are the same to rounding. */
typedef struct samx_learner {
    samx_mlp actor, critic, reward;   /* reward = the discriminator's RewardNN */
    int32_t ob_dim, ac_dim;
    float* params;                    /* [Pa + Pc + Pd] online parameters */
    float* grads;                     /* [Pa + Pc + Pd] out: unclipped flat gradients (all-reduce this) */
    float* clipped;                   /* [Pa + Pc + Pd] out: locally clipped gradients (logging), or NULL */
    float *adam_m, *adam_v;           /* [Pa + Pc + Pd] */
    float* targets;                   /* [Pa + Pc] target actor | target critic */
    int64_t* steps;                   /* device int64[3]: Adam update counts (actor, critic, reward) */
    const float* actor_reward_params; /* reward net the actor loss differentiates through
                                         (sam_agent.py:88,192); NULL when d_actor_loss_scale == 0 */
    const float *mu_mean, *mu_std;    /* agent obs statistics (sam_agent.py:150-156), NULL = off */
    const float *d_mean, *d_std;      /* discriminator obs statistics (discriminator.py:67-72) */
    float ac_high, gamma, polyak, clip_norm, actor_lr, critic_lr, d_lr, beta1, beta2, adam_eps;
    float td_loss_scale, wd_scale, q_actor_loss_scale, d_actor_loss_scale, ent_reg_scale;
    /* Return normalisation / PopArt / n-step returns (sam_agent.py:161-201,483-498,554-590,662-672).
     * All off when ret_mean == NULL and n_step == 0; when on, the step runs on the layer-wise path.
     * ret_mean / ret_std: device float[1] (ret_rms.mean / .std); ret_sums: device double[3] [sum, sumsq,
     * count].  With enable_popart the step updates the statistics with this batch's targets and
     * rescales the output layer of the online and the target critic before the losses (single
     * process; N > 1 needs an all-reduce of the sums in between: use the per-role entry points).
     * n_step > 1: the agent batch also gets its n-step lookahead targets (batch.a_idx required). */
    float *ret_mean, *ret_std;
    double* ret_sums;
    int32_t enable_popart, n_step;
    float td_loss_n_scale;
} samx_learner;

typedef struct samx_batch {  /* device inputs of one step; B rows each */
    int64_t B;
    /* discriminator update (sam.py:372-395): policy pairs, expert pairs, injected TF randoms */
    const float *p_obs, *p_acs, *e_obs, *e_acs, *eps_ob, *eps_ac, *real_labels, *fake_labels;
    /* agent update (sam_agent.py:634-660): replayed transitions; iws NULL = uniform, td_len NULL = 1 */
    const float *obs0, *acs, *rews, *dones1, *obs1, *iws, *td_len;
    /* Optional row indirection (all NULL/0 = the pointers above address B gathered rows).
     * p_idx: p_obs/p_acs address the replay storage and row r of the batch is storage row
     *        (ring_start + p_idx[r]) % ring_len — RingBuffer.get_batch, memory.py:33-34;
     * a_idx: the same for obs0/acs/rews/dones1/obs1;
     * e_start: device scalar; e_obs/e_acs address the expert pool and row r is pool row
     *        e_start[0] + r — PairDataset.get_next_batch, demo_dataset.py:37-40. */
    const int64_t *p_idx, *a_idx, *e_start;
    int64_t ring_start, ring_len;
    /* eps_ob == NULL: the library draws eps_ob/eps_ac and the smoothed labels itself (Philox keyed by
     * rng_seed and the reward-net update count).  label_smoothing: 0 = labels 1/0, 1 = one-sided
     * (real ~ 1.2 - 0.5u, discriminator.py:104-109), 2 = also fake ~ 0.3 - 0.3u (lines 97-103). */
    uint64_t rng_seed;
    int32_t label_smoothing;
    /* d_rms_sums != NULL: the commit of d.obs_rms.update for this batch (sam.py:389-390,
     * helpers/mpi_running_mean_std.py:40-45) happens INSIDE samx_step_grads.  d_rms_addvec
     * (double[2*ob+1]) must hold the batch's [sum, sum of squares, count] — from samx_d_input_stats
     * with commit = 0 or samx_d_input_sums_many, which can run any time before, e.g. for many steps
     * at once.  The discriminator normalises with the statistics of d_rms_sums + d_rms_addvec, and at
     * the end d_rms_sums (double[2*ob+1]) is advanced and learner.d_mean / d_std are rewritten (they
     * are outputs in this mode).  Single process only (N > 1 needs the all-reduce of line 55 in
     * between).  On the layer-wise path (samx_step_path == 0) samx_step_grads computes the batch's sums
     * itself (d_rms_addvec receives them; needs p_idx and e_start). */
    double* d_rms_sums;
    const double* d_rms_addvec;
    int64_t num_entries; /* entries in the replay ring: bounds the n-step lookahead window */
    /* learner.n_step > 0: the lookahead of a_idx done ahead (outputs of samx_nstep_lookahead[_many] for
     * exactly these B indices); all NULL = samx_step_grads runs the lookahead itself */
    const int64_t* n_end;
    const float *n_rews, *n_dones, *n_len;
    /* Return normalisation with PopArt under N processes (samx_step_grads_part bits 4/5): double[6] =
     * [sum, sum of squares, count] of the batch's 1-step targets, then of its n-step targets (zeros when
     * off) — what helpers/mpi_running_mean_std.py:47-59 all-reduces.  NULL otherwise. */
    double* ret_addvec;
} samx_batch;

typedef struct samx_step_out {
    float* d_losses;      /* [8]  discriminator.py:145-158 */
    float* actor_losses;  /* [3]  sam_agent.py:418-430 */
    float* critic_losses; /* [4]  sam_agent.py:479-517: td_loss_1, wd, total, td_loss_n (0 when off) */
    float* targ_q;        /* [B]  sam_agent.py:651-654 */
    float* td;            /* [B]  pre-update TD errors, or NULL */
    float* targ_q_n;      /* [B]  n-step targets (sam_agent.py:656-660), or NULL */
} samx_step_out;

int samx_fused_supported(const samx_learner* L, int64_t B);
/* Which kernels samx_step_grads runs for this learner and batch size: 1 = the whole-network fused kernels
 * (64x64 nets), 2 = the role-fused kernel for 256x256 leaky-ReLU nets without LayerNorm (one launch for the
 * layer chains of every role, one grouped launch for the weight-gradient products, one reduction), 0 = the
 * layer-wise GEMM path.  Paths 1 and 2 take batch.d_rms_addvec as an INPUT (the batch's sums, computed
 * ahead), path 0 computes it. */
int samx_step_path(const samx_learner* L, int64_t B);

/* The numbers samx_step_grads draws for update number `step` (0-based reward-net update count) when
 * batch.eps_ob is NULL, written out so a checker can replay the same update.  real/fake may be NULL. */
int samx_disc_randoms(uint64_t rng_seed, int64_t step, int64_t B, int ob_dim, int ac_dim,
                      float* eps_ob, float* eps_ac, float* real_labels, float* fake_labels, void* stream);

/* d.obs_rms.update on the discriminator's input rows WITHOUT gathering them first
 * (sam.py:389-390 + helpers/mpi_running_mean_std.py:47-59): float64 column sums over the policy rows
 * (replay storage via p_idx) and the expert rows (pool rows e_start[0]..+B).  With commit != 0 the sums
 * are also folded into `sums` and mean/std refreshed (single process); otherwise call samx_rms_commit
 * after the all-reduce of addvec. */
int samx_d_input_stats(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                       const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start, int64_t B,
                       int ob_dim, double* addvec, int commit, double* sums, float* mean_out,
                       float* std_out, void* ws, size_t ws_bytes, void* stream);
/* Scratch for samx_d_input_stats (large inputs are summed by many workgroups into per-group float64
 * partials that are then added in a fixed order); ws may be NULL when 2*B*ob_dim <= 16384. */
size_t samx_d_input_stats_ws_bytes(int64_t B, int ob_dim);
/* samx_d_input_stats WITHOUT a workspace argument and without any allocation.  No counterpart in the
 * reference: it exists for callers that cannot allocate device memory — code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The per-group float64 partials live in a
 * file-scope device array of the library (1 MiB + 256 bytes, held by every GPU the library runs on until
 * the process ends), which bounds the input: ob_dim <= samx_d_input_stats_static_max_ob() (512); wider
 * observations are rejected with SAMX_ERR_ARG.  Same arithmetic, same bits as samx_d_input_stats.  All calls
 * on one GPU share that array: calls that overlap on one GPU (two streams) race; calls on one stream are
 * ordered.  The array's address is looked up on the first call on each device, which must therefore
 * happen outside a stream capture. */
int samx_d_input_stats_static_max_ob(void);
int samx_d_input_stats_static(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                              const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start, int64_t B,
                              int ob_dim, double* addvec, int commit, double* sums, float* mean_out,
                              float* std_out, void* stream);
/* Head of a prioritized-replay row in ONE launch: workgroup 0 applies the pending priority update to both
 * trees (= samx_tree_set with n_upd < 1024: memory.py:373-379 of the PREVIOUS agent update), workgroup 1
 * computes the D-input sums of this row (= samx_d_input_stats with commit = 0: sam.py:389-390).  The two
 * do not depend on each other.  u != NULL: workgroup 0 goes on, on the updated trees, with this row's
 * stratified sample and importance weights (= samx_per_sample with the same arguments: memory.py:266-309)
 * while workgroup 1 still adds.  copy_src != NULL: before anything else both workgroups copy copy_n
 * int64 words from copy_src (pinned, device-accessible host memory: the row's host-drawn numbers) to copy_dst
 * (device) — every other input pointer may point into copy_dst; this replaces a separate upload.  Words
 * [copy_skip_lo, copy_skip_lo + copy_skip_n) are not copied: an output of this launch that lives inside
 * copy_dst (idxs_out) must lie there. */
int samx_per_row_prologue(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                          const int64_t* upd_idx, const double* upd_leaf, int64_t n_upd,
                          const float* replay_obs0, int64_t ring_start, int64_t ring_len, const int64_t* p_idx,
                          const float* pool_obs0, const int64_t* e_start, int64_t B, int ob_dim, double* addvec,
                          int64_t num_entries, const double* u, double beta, int64_t* idxs_out, double* iws_out,
                          float* iws_f32_out, const int64_t* copy_src, int64_t* copy_dst, int64_t copy_n,
                          int64_t copy_skip_lo, int64_t copy_skip_n, void* stream);

/* samx_per_row_prologue with the pending update given as TD ERRORS (upd_td float32[n_upd], HBM): workgroup 0
 * first runs the refresh of samx_per_leaves_from_td on them (leaves kept in LDS, per_state updated), then
 * goes on as above.  No upload words: every input already lives in HBM, so consecutive prioritized rows can
 * sit in one captured graph (sam_agent.py:693-703 of row k feeding memory.py:266-309 of row k + 1).
 * With u != NULL and cap <= 2^30 the work is spread over more of the chip in two launches: 16 subtree
 * workgroups refresh + rebuild the levels below the top 12 (one more adds the D-input sums), then B/32
 * workgroups each rebuild the top 12 levels in LDS and draw their share of the row's samples.
 * replay_obs0 == NULL: no D-input sums here (the caller computed them at the head of its launch). */
int samx_per_row_prologue_td(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                             const int64_t* upd_idx, const float* upd_td, int64_t n_upd, double alpha,
                             int64_t num_demos, double demos_eps, void* per_state,
                             const float* replay_obs0, int64_t ring_start, int64_t ring_len, const int64_t* p_idx,
                             const float* pool_obs0, const int64_t* e_start, int64_t B, int ob_dim, double* addvec,
                             int64_t num_entries, const double* u, double beta, int64_t* idxs_out, double* iws_out,
                             float* iws_f32_out, void* stream);

/* ---- the same sampling / refresh calls with the ring's geometry in DEVICE words -------------------------------
 * Every store_transition between two training phases changes num_entries and, once the ring is full, its start
 * (memory.py:36-57,337-342; the reference's loop appends between any two phases, sam.py:317-323).  Passed by value
 * they are baked into a captured graph, which then has to be captured again after every rollout.  These variants
 * read geom = int64[2] {ring start, num_entries} from device memory (the caller uploads the two words with the
 * launch's other host-drawn numbers and has checked 2 <= num_entries <= cap; the kernels clamp to that range for
 * memory safety), and hand out, next to the sampled LOGICAL indices the trees and the priority update work on
 * (idxs_out; idx_copy_out, if not NULL, receives a second copy: the pending-update buffer), the STORAGE rows
 * rows_out[i] = (start + idx) % ring_len (memory.py:33-34) the network kernels gather from with ring_start = 0.
 * Same arithmetic, same bits as the by-value forms (tests/test_gpu_replay.py). */
int samx_per_sample_rows(const double* sum_tree, const double* min_tree, int64_t cap, const int64_t* geom,
                         int64_t ring_len, const double* u, int64_t B, double beta, int64_t* idxs_out,
                         double* iws_out, float* iws_f32_out, int64_t* rows_out, int64_t* idx_copy_out,
                         void* stream);
/* samx_per_row_prologue_td (two-launch form, cap <= 2^30, no D-input sums) on device geometry words. */
int samx_per_row_prologue_td_rows(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                                  const int64_t* upd_idx, const float* upd_td, int64_t n_upd, double alpha,
                                  int64_t num_demos, double demos_eps, void* per_state, const int64_t* geom,
                                  int64_t ring_len, const double* u, int64_t B, double beta, int64_t* idxs_out,
                                  double* iws_out, float* iws_f32_out, int64_t* rows_out, int64_t* idx_copy_out,
                                  void* stream);
/* samx_per_leaves_from_td with num_entries (the bound of the reference's second assert) read from geom[1]. */
int samx_per_leaves_from_td_geom(const float* td, const int64_t* idxs, int64_t n, int64_t cap, const int64_t* geom,
                                 double alpha, int64_t num_demos, double demos_eps, double* leaf_out,
                                 void* per_state, void* stream);

/* a[n] (float) and b[n] (int64) to PINNED HOST memory (device-accessible) in one launch: the refreshed TD
 * errors and the sampled indices a prioritized-replay row hands back to the host (sam_agent.py:693-703). */
int samx_to_host_pair(const float* a, float* a_host, const int64_t* b, int64_t* b_host, int64_t n, void* stream);

/* The batch sums (no commit) of K consecutive steps in ONE launch: step k reads its index words at
 * p_idx + k*idx_stride and e_start + k*idx_stride (int64 words) and writes addvec + k*addvec_stride
 * (doubles).  The sums of a batch do not depend on the learner, so a multi-step launch computes them
 * up front.  2*B*ob_dim <= 16384 (the sizes of the fused path). */
int samx_d_input_sums_many(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                           const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start,
                           int64_t idx_stride, int64_t B, int ob_dim, double* addvec,
                           int64_t addvec_stride, int64_t K, void* stream);
/* The same for any batch size: 2*B*ob_dim > 16384 takes the two-stage sums of samx_d_input_stats for the K
 * steps in two launches; ws: K * samx_d_input_stats_ws_bytes(B, ob_dim) bytes (may be NULL for the small
 * sizes). */
int samx_d_input_sums_many_ws(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                              const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start,
                              int64_t idx_stride, int64_t B, int ob_dim, double* addvec, int64_t addvec_stride,
                              int64_t K, void* ws, size_t ws_bytes, void* stream);
size_t samx_step_ws_bytes(const samx_learner* L, int64_t B);

/* d.get_grads/get_losses + mu's get_targ_q / actor & critic losses and grads (sam.py:398-399,
 * sam_agent.py:651-687): fills learner.grads and the outputs, advances the three device step
 * counters, leaves the per-optimiser gradient norms in the workspace. */
int samx_step_grads(const samx_learner* L, const samx_batch* batch, const samx_step_out* out, void* ws,
                    size_t ws_bytes, void* stream);

/* Measurement aid: the same call split in two.  part 1 = the network passes only (fused path: the one
 * whole-network kernel; layer-wise path: the GEMM chains of the three roles), part 2 = the reduction
 * of the per-tile partials / norms / scalar losses / step tick, part 3 = both (= samx_step_grads).
 * bench.py times part 1 alone with HIP events for the roofline line.
 * Role subsets (the reference runs d_steps discriminator updates, then g_steps agent updates per training
 * step, sam.py:323-433; launcher: 1 and 3): part bit 2 (value 4) leaves the discriminator role out, bit 3
 * (value 8) the agent roles; the optimiser segments, losses, statistics commit and step counters of the
 * role left out are not touched.  Pass the same subset to samx_step_apply (mode bits 3 and 4).
 * Return normalisation with PopArt under N processes: ret_rms.update all-reduces the batch sums of the
 * targets BETWEEN the target pass and the PopArt rescale (sam_agent.py:662-672,
 * helpers/mpi_running_mean_std.py:47-59), so the call splits there: part bit 4 (value 16, with bit 0) runs
 * only [lookahead ->] targets -> batch sums into batch.ret_addvec and returns; after the caller's
 * all-reduce(SUM) of ret_addvec, part bit 5 (value 32) resumes: commit of ret_addvec, PopArt, then the
 * loss/gradient passes as usual against out.targ_q (/ targ_q_n) of the first call.  Same learner, batch,
 * out and workspace in both calls.
 * Part bit 6 (value 64): the update counters of the roles are NOT advanced — an evaluation that no optimiser
 * pass follows (d.get_grads / d.get_losses, sam.py:398-399: the reference runs them as sess.runs of their own
 * before d.optimize).
 * Part bits 7 and 8 (values 128, 256; samx_step_path 2 only, ignored elsewhere): the 256-wide role kernels read
 * k-interleaved images of every W1 / W2 for their forward products; by default each call writes the images it
 * needs from the parameter vectors first (one launch), so nothing can go stale behind the caller's back.  A caller
 * that runs several update steps back to back on ONE workspace may instead let the optimiser pass keep them
 * current (samx_step_apply mode bit 6): bit 8 = write the images of ALL five networks in this call, bit 7 = the
 * images in the workspace are current, write none.  Bit 7 is only valid if since the last call with bit 8 on this
 * workspace W1 / W2 of the learner's networks and targets were modified by nothing but samx_step_apply calls with
 * mode bit 6 (PopArt's rescale touches the heads only).
 * Part bit 9 (value 512; with bit 1, samx_step_path 1 only — refused elsewhere; one process): the reduction launch
 * also runs the optimiser pass, i.e. the call then includes samx_step_apply(mode 3 [| the same role subset],
 * grad_scale 1) and the caller does not make that call: gradients stay in registers between the two halves and the
 * blocks exchange nothing but their sums of squares (one tagged 64-bit word each, agent-scope atomics, bounded wait;
 * fused_small.hip finalize_apply_kernel).  Same bits as the two calls.  Needs learner.clipped, the counter tick (no
 * bit 6) and a workspace that was zero-filled before its first use; on a device that cannot make the launch's blocks
 * resident within 2 s the parameters are poisoned with NaN instead of being updated wrongly.
 * Under HIP stream capture: on the layer-wise path (samx_step_path 0) the call forks two role streams from
 * `stream`; `stream` must then be the capture's ORIGIN stream.  A fork of a forked stream crashes
 * hipStreamEndCapture in the HIP runtime this was built against (torch 2.10.0+rocm7.0) — with plain torch streams
 * too, libsamx not involved: profiles/r03_nested_fork_capture.txt, tools/exp/nested_fork_capture.py.  The runtime
 * offers no way to tell an origin stream from a forked one, so the call cannot reject it; SAMX_SERIAL_ROLES=1
 * makes the call fork nothing.  Paths 1 and 2 never fork. */
int samx_step_grads_part(const samx_learner* L, const samx_batch* batch, const samx_step_out* out,
                         void* ws, size_t ws_bytes, int part, void* stream);

/* d.optimize + actor/critic optimize + update_target_net (sam.py:402,433; sam_agent.py:690-691):
 * mode bit 0: write learner.clipped = grads clipped by their own norm (tf_util.py:137-151);
 * mode bit 1: grads * grad_scale -> clip_by_global_norm -> Adam -> Polyak on the agent's targets;
 * mode bit 2: recompute the norms first (set it when grads changed after samx_step_grads, i.e. after
 * the all-reduce); mode bit 3 (value 8): leave the reward-net segment out, bit 4 (value 16): leave the
 * actor and critic segments out (role subsets, see samx_step_grads_part); bit 5 (value 32): no Polyak step on
 * the targets (mu.train() and mu.update_target_net() are separate calls in the reference's loop, sam.py:421,433:
 * the object mirror runs the target update through samx_polyak); bit 6 (value 64, with bit 1 and without bit 5;
 * samx_step_path 2 only, ignored elsewhere): every updated W1 / W2 element, and its Polyak-averaged target, is
 * also written into the weight images of the workspace (samx_step_grads_part bits 7/8).  Single GPU: one call, mode 3.  N GPUs: mode 1, all-reduce(grads), mode 6 with
 * grad_scale = 1/N.  `ws` must be the workspace samx_step_grads used, B the same batch size. */
int samx_step_apply(const samx_learner* L, int64_t B, int mode, float grad_scale, void* ws,
                    size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMX_H */
