// Internal entry points of the layer-wise network path that the step (step.hip) calls with the batch
// still addressed IN PLACE (ring rows / pool slice / in-kernel randoms), see samx_batch in include/samx.h.
#pragma once
#include "samx_common.h"
#include "../../include/samx.h"

namespace samx {

// Where the discriminator's rows come from when they are not dense (all optional, zero = dense inputs):
struct DiscSrc {
    const int64_t* p_idx = nullptr;    // p_obs/p_acs address the replay storage: row r = (ring_start + p_idx[r]) % ring_len
    int64_t ring_start = 0, ring_len = 0;
    const int64_t* e_start = nullptr;  // e_obs/e_acs address the expert pool: row r = e_start[0] + r
    bool draw = false;                 // eps_ob / eps_ac / labels are drawn here (Philox, rng.h) instead of read
    uint64_t rng_seed = 0;
    const int64_t* step_dev = nullptr; // device: the reward-net update count that keys the draws
    float *real_out = nullptr, *fake_out = nullptr;   // [B] smoothed labels written by the prep kernel (or null)
};

// samx_disc_grad (include/samx.h) with the rows resolved inside its first kernel
int disc_grad_src(const samx_mlp* dd, const float* pd, const float* p_obs, const float* p_acs, const float* e_obs,
                  const float* e_acs, const float* eps_ob, const float* eps_ac, const float* real_labels,
                  const float* fake_labels, const float* mean, const float* stdv, int64_t B, int ob_dim,
                  float ent_reg_scale, float* losses_out, float* grad_out, void* ws, size_t ws_bytes,
                  const DiscSrc* src, void* stream);

// 256x256 leaky-ReLU learners (Humanoid-shaped): ONE launch of the role-fused kernel of fused_wide.hip for
// the chains of layer products of every role, ONE grouped launch for the weight-gradient products, ONE
// reduction.  ws_tc / ws_a: one samx_nets_ws_bytes workspace per agent role, ws_d: samx_disc_ws_bytes.
bool wide_step_supported(const samx_learner* L, int64_t B);
// Return normalisation / PopArt / n-step returns: the TD targets come from their own launch (the statistics and the
// critic heads change between the target pass and the loss pass, sam_agent.py:651-672).  targets_only: launch
// only wide::launch_targets (1-step targets into out.targ_q; with n_idx the n-step targets into tqn) and return;
// use_ext: the roles read out.targ_q / tqn instead of computing the targets.
struct WideExt {
    bool targets_only = false, use_ext = false;
    // packed: the k-interleaved weight images of every network this call's roles need were written by the
    // targets_only launch of the SAME samx_step_grads_part call (nothing but the critic heads changed since)
    bool packed = false;
    // images_current: every image in the workspace is current (samx_step_grads_part bit 7: the optimiser pass of the
    // previous row kept them, samx_step_apply mode bit 6) — nothing is packed; pack_all (bit 8): pack all five
    // networks' images whatever this call's roles need, so that later rows can rely on them
    bool images_current = false, pack_all = false;
    const int64_t* n_idx = nullptr;
    const float *n_rews = nullptr, *n_dones = nullptr, *n_len = nullptr;
    float* tqn = nullptr;
};
size_t wide_reduce_ws_bytes(const samx_learner* L, int64_t B);
// the weight images inside the reduction workspace: img[n][0 / 1] = W1 / W2 of n = actor, critic, reward net, target
// actor, target critic (for the optimiser pass that keeps them current)
// row[n][0 / 1]: the transposed images (wide::Net::W1t / W2t) of W1 / W2 of n = actor (W2 only), critic, reward net
int wide_images(const samx_learner* L, int64_t B, void* ws_red, size_t red_bytes, float* img[5][2], float* row[3][2]);
// Fills learner.grads, the per-segment sums of squares (sumsq, fused::kMaxSegBlocks per optimiser) and the
// losses, and ticks the step counters of the roles it ran.  part bit 0: the role launch and the grouped
// products (with the head-gradient slot sums); bit 1: the reduction that finishes the gradients, with the
// scalar losses, the statistics commit (batch.d_rms_sums: the batch's sums were computed ahead, as on the fused
// path) and the counter tick.  Four launches per step with samx_step_apply, all on the caller's stream.
int wide_step_roles(const samx_learner* L, const samx_batch* b, const samx_step_out* o, void* ws_tc, void* ws_a,
                    size_t nets_bytes, void* ws_d, size_t disc_bytes, void* ws_red, size_t red_bytes, float* sumsq,
                    bool do_a, bool do_d, int part, hipStream_t st, const WideExt* ext);

}  // namespace samx
