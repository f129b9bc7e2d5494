// Internal interface of the whole-network fused kernels (fused_small.hip) used by step.hip.
#pragma once
#include "ret_tail.h"
#include "samx_common.h"
#include "../../include/samx.h"

namespace samx {
namespace fused {

constexpr int kRows = 16;          // batch rows per workgroup
constexpr int kMaxSegBlocks = 256;  // blocks (and sum-of-squares partials) per optimiser segment

struct AgentArgs {
    // inputs (device): batch rows
    const float *obs0, *acs, *rews, *dones1, *obs1, *iws, *td_len, *mean, *stdv;
    // parameters (device, flat)
    const float *actor, *critic, *targ_actor, *targ_critic, *reward;
    int ob, ac;
    int64_t B;
    float high, gamma, td_scale, q_scale, d_scale;
    // outputs
    float *targ_q_out, *td_out;
    float* partials;  // [B/16][P] slots, P = Pa + Pc
    int64_t P, Pa;
    float* loss_part;  // [B/16][8]: sum huber*iw, sum Q, sum D, (block 0) sum W1^2 + W2^2 of the critic, sum huber_n*iw
    // optional row indirection into ring storage (memory.py:33-34); idx == nullptr: direct rows
    const int64_t* idx;
    int64_t ring_start, ring_len;
    // Return normalisation (sam_agent.py:161-201): device scalars, nullptr = off.  The critic outputs
    // normalised Q; predictions are de-normalised (s * std + mean), targets normalised.
    const float *ret_mean, *ret_std;
    // External targets: with return normalisation + PopArt or n-step returns the TD targets come from
    // launch_targets (the return statistics / critic heads change between target and loss passes).
    // ext_tq != nullptr: the target+critic role runs critic-only against ext_tq [B] (and ext_tqn [B]).
    const float *ext_tq, *ext_tqn;
    float tdn_scale;
    // launch_targets only: the n-step rows (memory.py:113-160): obs1 rows through n_idx (ring), dense
    // n_rews / n_dones / n_len [B]; targ_qn_out [B]
    const int64_t* n_idx;
    const float *n_rews, *n_dones, *n_len;
    float* targ_qn_out;
};

struct DiscArgs {
    const float *p_obs, *p_acs, *e_obs, *e_acs, *eps_ob, *eps_ac, *real, *fake, *mean, *stdv;
    const float* params;
    int ob, ac;
    int64_t B;
    float ent_scale, gp_coeff;
    float* partials;   // [3B/16][P] slots
    int64_t P;
    float* loss_part;  // [2B/16][4]: sum ce, sum entropy-CE, sum sigmoid
    double* gp_part;   // [B/16][2]: sum g_ob^2, sum g_ac^2
    float* gp_out;     // [1]: (N-1)^2
    const int64_t *p_idx, *e_start;  // optional indirection: replay ring rows / expert pool slice
    int64_t ring_start, ring_len;
    // in-kernel randoms (used when eps_ob == nullptr)
    uint64_t rng_seed;
    const int64_t* rng_step;  // device: reward-net update count
    int label_smoothing;
    // d.obs_rms.update folded into the step (sam.py:389-390).  rms_sums != NULL:
    // rms_addvec holds this batch's float64 column sums (computed ahead, all-reduced when N > 1); every workgroup normalises
    // with the statistics of sums + addvec and finalize commits them (FinalizeArgs::rms_*).
    // mean/stdv are then unused.
    const double* rms_sums;    // [2*ob+1]
    const double* rms_addvec;  // [2*ob+1]
};

struct Segment {
    const float* partials;  // first slot, already offset to this segment
    int64_t slot_stride;
    int nslots;
    int scaled_from;  // slots [scaled_from, nslots) are multiplied by the gradient-penalty scalar k
    int64_t n;
    float* grad;
    float* sumsq;  // [nblocks]
    int nblocks;
    // weight decay: grad += wd * param on two index ranges (critic hidden kernels)
    float wd;
    const float* param;
    int64_t wd0_lo, wd0_hi, wd1_lo, wd1_hi;
    int64_t* step_dev;  // ticked once per finalize
};

struct FinalizeArgs {
    Segment seg[3];
    int nseg;
    int64_t B;
    // agent losses
    const float* agent_loss_part;
    int agent_tiles;
    float q_scale, d_scale, td_scale, wd_scale;
    const float *ret_mean, *ret_std;   // return normalisation (device scalars), nullptr = off
    float tdn_scale;                   // n-step loss scale; agent_loss_part[.][4] holds its per-tile sums
    int has_tdn;
    float *actor_losses, *critic_losses;
    // discriminator losses
    const float* disc_loss_part;
    int disc_tiles_b;
    float ent_scale, gp_coeff;
    float* gp;               // [1] out: (N-1)^2
    const double* gp_part;   // [gp_tiles][2] sums of squares of the input gradient (NULL: no D segment)
    int gp_tiles, ob, ac;
    float* disc_losses;
    // commit of the inline running-statistics update (NULL: none)
    double* rms_sums;
    const double* rms_addvec;
    float *rms_mean, *rms_std;
};

struct ApplySeg {
    float *param, *m, *v, *targ, *clipped;
    const float* grad;
    float* sumsq;
    int nsumsq;
    int64_t n;
    int nblocks;
    float lr, clip;
    const int64_t* step_dev;
    // 256-wide path (samx_step_apply mode bit 6): the k-interleaved images of W1 / W2 that the role kernels' forward
    // products read (fused_wide.h: element (k, j) at ((k >> 2) * 256 + j) * 4 + (k & 3)) follow the update, for the
    // segment's network and, with Polyak, its target network.  W1 = [0, img_n1), W2 = [img_o2, img_o2 + 256 * 256).
    // tr1 / tr2: the images of the TRANSPOSE the backward products read, element (row, column) at
    // ((column >> 2) * J + row - r0) * 4 + (column & 3): tr2 = all of W2 (J = 256, r0 = 0); tr1 = rows >= tr1_r0 of W1
    // with tr1_J column slots (null: none)
    float *img1 = nullptr, *img2 = nullptr, *timg1 = nullptr, *timg2 = nullptr, *tr1 = nullptr, *tr2 = nullptr;
    int tr1_r0 = 0, tr1_J = 0;
    int64_t img_n1 = 0, img_o2 = 0;
};

struct ApplyArgs {
    ApplySeg seg[3];
    int nseg;
    int mode;  // bit 0: write clipped gradient, bit 1: Adam (+Polyak)
    float grad_scale, beta1, beta2, eps, tau;
};

bool supported(const samx_mlp* d, int64_t B);   // 64x64, in <= 64, out <= 16, B % 16 == 0 (with or without LayerNorm)
bool fits_lds(int ob, int ac);
// LayerNorm nets (fused_ln.hip): same contracts as the launchers below; the critic role always runs
// against stored targets (AgentArgs::ext_tq), which launch_targets_ln produces
bool fits_lds_ln(int ob, int ac, bool use_d);
int launch_step_ln(const AgentArgs* a, const DiscArgs* d, hipStream_t st);
int launch_targets_ln(const AgentArgs* a, hipStream_t st);
int launch_td_ln(const samx_mlp* dc, const float* critic, const float* obs0, const float* acs, const float* tc1s,
                 const float* mean, const float* stdv, int64_t B, int ob, const int64_t* idx, int64_t ring_start,
                 int64_t ring_len, float* td_out, const float* ret_mean, const float* ret_std, float* td_copy,
                 int64_t* idx_copy, hipStream_t st);
int seg_blocks(int64_t n);
struct StepArgs {
    AgentArgs a;
    DiscArgs d;
    int n_a, n_tc, n_disc;  // workgroups per role: actor tiles, target+critic tiles, D tiles (3B/16)
};
// agent roles + discriminator first kernel in ONE launch, then the discriminator's second kernel
int launch_step(const AgentArgs* a, const DiscArgs* d, hipStream_t st);
// TD targets of B rows (and the n-step targets when a->n_idx) with the target nets: one launch
int launch_targets(const AgentArgs* a, hipStream_t st);
// ret_rms.update(tq [, tqn]) + PopArt rescale of the critic heads (sam_agent.py:554-590,662-672), one launch
// (any hidden widths: also what the 256-wide path runs between its target launch and its role launch)
int launch_ret_popart(const float* tq, const float* tqn, int64_t B, double* sums, float* mean, float* stdv,
                      int popart, float* critic, float* targ_critic, const samx_mlp* dc, int phase,
                      double* addvec, hipStream_t st);   // phase 1: batch sums -> addvec[6] only; 2: commit addvec; 0: both
constexpr int kLossStride = 8;   // floats per tile in AgentArgs::loss_part
// TD errors of B rows (rows of the storages through idx when idx != nullptr) with one whole-network launch
int launch_td(const samx_mlp* dc, const float* critic, const float* obs0, const float* acs, const float* tc1s,
              const float* mean, const float* stdv, int64_t B, int ob, const int64_t* idx, int64_t ring_start,
              int64_t ring_len, float* td_out, const float* ret_mean, const float* ret_std, float* td_copy,
              int64_t* idx_copy, hipStream_t st);
int launch_finalize(const FinalizeArgs& f, hipStream_t st);
int launch_apply(const ApplyArgs& f, hipStream_t st);
// both in one launch (one process; fused_small.hip finalize_apply_kernel): slots [3 * kMaxSegBlocks] and epoch [2]
// live in the step workspace, zero before their first use
int launch_finalize_apply(const FinalizeArgs& f, const ApplyArgs& a, unsigned long long* slots, unsigned* epoch,
                          hipStream_t st);
// (ta / tb / tc: step counters advanced by the same launch, or null)
int launch_seg_sumsq(const ApplyArgs& f, hipStream_t st, int64_t* ta = nullptr, int64_t* tb = nullptr, int64_t* tc = nullptr);
int launch_tick(int64_t* a, int64_t* b, int64_t* c, hipStream_t st);

}  // namespace fused
}  // namespace samx
