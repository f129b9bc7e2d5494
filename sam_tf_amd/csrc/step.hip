// samx_step_grads / samx_step_apply: one SAM update step's gradients and optimiser pass for the whole
// learner (discriminator + actor + critic), choosing the whole-network fused kernels when the
// configuration allows (64x64 nets, batch a multiple of 16) and the layer-wise MFMA path otherwise.
//   reference sequence: sam.py:369-433 (D update, then mu.train(), then update_target_net())
#include "samx_common.h"
#include "fused_small.h"
#include "../../include/samx.h"

using namespace samx;

namespace {

struct StepWs {
    float *agent_part, *disc_part, *agent_loss, *disc_loss, *gp_out, *sumsq;
    double* gp_part;
    void* nets_ws;
    size_t nets_bytes;
    void* disc_ws;
    size_t disc_bytes;
};

bool use_fused(const samx_learner* L, int64_t B) {
    return fused::supported(&L->actor, B) && fused::supported(&L->critic, B) &&
           fused::supported(&L->reward, B) && L->critic.in_dim == L->ob_dim + L->ac_dim &&
           L->reward.in_dim == L->critic.in_dim;
}

StepWs carve(Arena& ar, const samx_learner* L, int64_t B) {
    StepWs w;
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const int64_t tiles = (B + fused::kRows - 1) / fused::kRows;
    const bool f = use_fused(L, B);
    w.agent_part = ar.take<float>(f ? tiles * (Pa + Pc) : 1);
    w.disc_part = ar.take<float>(f ? 3 * tiles * Pd : 1);
    w.agent_loss = ar.take<float>(tiles * 4);
    w.disc_loss = ar.take<float>(2 * tiles * 4);
    w.gp_out = ar.take<float>(4);
    w.sumsq = ar.take<float>(3 * fused::kMaxSegBlocks);
    w.gp_part = ar.take<double>(2 * tiles);
    w.nets_bytes = f ? 0 : samx_nets_ws_bytes(&L->actor, &L->critic, &L->reward, B);
    w.disc_bytes = f ? 0 : samx_disc_ws_bytes(&L->reward, B);
    w.nets_ws = ar.take<char>(w.nets_bytes);
    w.disc_ws = ar.take<char>(w.disc_bytes);
    return w;
}

int check_learner(const samx_learner* L, const char* who) {
    SAMX_REQUIRE(L, "%s: null learner", who);
    SAMX_REQUIRE(L->params && L->grads && L->adam_m && L->adam_v && L->targets && L->steps,
                 "%s: learner state pointers must be set", who);
    SAMX_REQUIRE(L->ob_dim > 0 && L->ac_dim > 0 && L->actor.in_dim == L->ob_dim &&
                     L->actor.out_dim == L->ac_dim && L->critic.in_dim == L->ob_dim + L->ac_dim &&
                     L->critic.out_dim == 1 && L->reward.in_dim == L->critic.in_dim &&
                     L->reward.out_dim == 1,
                 "%s: network shapes do not fit ob_dim/ac_dim", who);
    SAMX_REQUIRE((L->mu_mean == nullptr) == (L->mu_std == nullptr) &&
                     (L->d_mean == nullptr) == (L->d_std == nullptr),
                 "%s: mean/std must come together", who);
    SAMX_REQUIRE(L->d_actor_loss_scale == 0.f || L->actor_reward_params,
                 "%s: d_actor_loss_scale != 0 needs actor_reward_params", who);
    return SAMX_OK;
}

fused::ApplyArgs make_apply(const samx_learner* L, const StepWs& w, int mode, float grad_scale) {
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const int64_t off[3] = {0, Pa, Pa + Pc}, n[3] = {Pa, Pc, Pd};
    const float lr[3] = {L->actor_lr, L->critic_lr, L->d_lr};
    fused::ApplyArgs f;
    f.nseg = 3;
    f.mode = mode;
    f.grad_scale = grad_scale;
    f.beta1 = L->beta1; f.beta2 = L->beta2; f.eps = L->adam_eps; f.tau = L->polyak;
    for (int k = 0; k < 3; ++k) {
        fused::ApplySeg& s = f.seg[k];
        s.param = L->params + off[k];
        s.m = L->adam_m + off[k];
        s.v = L->adam_v + off[k];
        s.targ = k < 2 ? L->targets + off[k] : nullptr;
        s.clipped = L->clipped ? L->clipped + off[k] : nullptr;
        s.grad = L->grads + off[k];
        s.n = n[k];
        s.nblocks = fused::seg_blocks(n[k]);
        s.sumsq = w.sumsq + k * fused::kMaxSegBlocks;
        s.nsumsq = s.nblocks;
        s.lr = lr[k];
        s.clip = L->clip_norm;
        s.step_dev = L->steps + k;
    }
    return f;
}

}  // namespace

extern "C" {

int samx_fused_supported(const samx_learner* L, int64_t B) { return L && use_fused(L, B) ? 1 : 0; }

size_t samx_step_ws_bytes(const samx_learner* L, int64_t B) {
    if (!L || B <= 0) return 0;
    Arena ar(nullptr, 0);
    carve(ar, L, B);
    return align_up(ar.off, 256) + 256;
}

int samx_step_grads(const samx_learner* L, const samx_batch* b, const samx_step_out* o, void* ws,
                    size_t ws_bytes, void* stream) {
    int rc = check_learner(L, "samx_step_grads");
    if (rc != SAMX_OK) return rc;
    SAMX_REQUIRE(b && o && ws, "samx_step_grads: null argument");
    SAMX_REQUIRE(b->B > 0, "samx_step_grads: empty batch");
    SAMX_REQUIRE(b->p_obs && b->p_acs && b->e_obs && b->e_acs && b->eps_ob && b->eps_ac && b->obs0 &&
                     b->acs && b->rews && b->dones1 && b->obs1,
                 "samx_step_grads: missing batch input");
    SAMX_REQUIRE(o->d_losses && o->actor_losses && o->critic_losses && o->targ_q,
                 "samx_step_grads: missing output buffer");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    StepWs w = carve(ar, L, b->B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_step_grads: workspace %zu < %zu", ws_bytes, ar.off);
    const int64_t B = b->B;
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const float* pa = L->params;
    const float* pc = L->params + Pa;
    const float* pd = L->params + Pa + Pc;
    const float* ta = L->targets;
    const float* tc = L->targets + Pa;

    if (!use_fused(L, B)) {
        // layer-wise path, role by role (same order as the reference: D first)
        rc = samx_disc_grad(&L->reward, pd, b->p_obs, b->p_acs, b->e_obs, b->e_acs, b->eps_ob, b->eps_ac,
                            b->real_labels, b->fake_labels, L->d_mean, L->d_std, B, L->ob_dim,
                            L->ent_reg_scale, o->d_losses, L->grads + Pa + Pc, w.disc_ws, w.disc_bytes,
                            stream);
        if (rc != SAMX_OK) return rc;
        rc = samx_targ_q(&L->actor, ta, &L->critic, tc, b->obs1, b->rews, b->dones1, b->td_len, L->mu_mean,
                         L->mu_std, B, L->gamma, L->ac_high, o->targ_q, w.nets_ws, w.nets_bytes, stream);
        if (rc != SAMX_OK) return rc;
        rc = samx_actor_grad(&L->actor, pa, &L->critic, pc, &L->reward, L->actor_reward_params, b->obs0,
                             L->mu_mean, L->mu_std, B, L->ac_high, L->q_actor_loss_scale,
                             L->d_actor_loss_scale, o->actor_losses, L->grads, w.nets_ws, w.nets_bytes,
                             stream);
        if (rc != SAMX_OK) return rc;
        rc = samx_critic_grad(&L->critic, pc, b->obs0, b->acs, b->iws, o->targ_q, L->mu_mean, L->mu_std, B,
                              L->ob_dim, L->td_loss_scale, L->wd_scale, o->critic_losses, L->grads + Pa,
                              o->td, w.nets_ws, w.nets_bytes, stream);
        if (rc != SAMX_OK) return rc;
        fused::ApplyArgs f = make_apply(L, w, 0, 1.f);
        rc = fused::launch_seg_sumsq(f, st);
        if (rc != SAMX_OK) return rc;
        return fused::launch_tick(L->steps, L->steps + 1, L->steps + 2, st);
    }

    const int tiles = (int)(B / fused::kRows);
    fused::DiscArgs d;
    d.p_obs = b->p_obs; d.p_acs = b->p_acs; d.e_obs = b->e_obs; d.e_acs = b->e_acs;
    d.eps_ob = b->eps_ob; d.eps_ac = b->eps_ac; d.real = b->real_labels; d.fake = b->fake_labels;
    d.mean = L->d_mean; d.stdv = L->d_std; d.params = pd;
    d.ob = L->ob_dim; d.ac = L->ac_dim; d.B = B;
    d.ent_scale = L->ent_reg_scale; d.gp_coeff = 10.f;  // discriminator.py:154
    d.partials = w.disc_part; d.P = Pd; d.loss_part = w.disc_loss; d.gp_part = w.gp_part;
    d.gp_out = w.gp_out;

    fused::AgentArgs a;
    a.obs0 = b->obs0; a.acs = b->acs; a.rews = b->rews; a.dones1 = b->dones1; a.obs1 = b->obs1;
    a.iws = b->iws; a.td_len = b->td_len; a.mean = L->mu_mean; a.stdv = L->mu_std;
    a.actor = pa; a.critic = pc; a.targ_actor = ta; a.targ_critic = tc; a.reward = L->actor_reward_params;
    a.ob = L->ob_dim; a.ac = L->ac_dim; a.B = B;
    a.high = L->ac_high; a.gamma = L->gamma; a.td_scale = L->td_loss_scale;
    a.q_scale = L->q_actor_loss_scale; a.d_scale = L->d_actor_loss_scale;
    a.targ_q_out = o->targ_q; a.td_out = o->td;
    a.partials = w.agent_part; a.P = Pa + Pc; a.Pa = Pa; a.loss_part = w.agent_loss;
    rc = fused::launch_step(&a, &d, st);
    if (rc != SAMX_OK) return rc;

    fused::FinalizeArgs f;
    f.nseg = 3;
    f.B = B;
    const int64_t off[3] = {0, Pa, Pa + Pc}, n[3] = {Pa, Pc, Pd};
    for (int k = 0; k < 3; ++k) {
        fused::Segment& s = f.seg[k];
        s.partials = k < 2 ? w.agent_part + off[k] : w.disc_part;
        s.slot_stride = k < 2 ? Pa + Pc : Pd;
        s.nslots = k < 2 ? tiles : 3 * tiles;
        s.n = n[k];
        s.grad = L->grads + off[k];
        s.nblocks = fused::seg_blocks(n[k]);
        s.sumsq = w.sumsq + k * fused::kMaxSegBlocks;
        s.wd = 0.f; s.param = nullptr; s.wd0_lo = s.wd0_hi = s.wd1_lo = s.wd1_hi = 0;
        s.step_dev = L->steps + k;
    }
    {   // weight decay on the critic's two hidden kernels (networks.py:385-386,406-412)
        fused::Segment& s = f.seg[1];
        const int64_t inc = L->critic.in_dim, h = 64;
        s.wd = L->wd_scale; s.param = pc;
        s.wd0_lo = 0; s.wd0_hi = inc * h;
        s.wd1_lo = inc * h + h; s.wd1_hi = s.wd1_lo + h * h;
    }
    f.agent_loss_part = w.agent_loss; f.agent_tiles = tiles;
    f.q_scale = L->q_actor_loss_scale; f.d_scale = L->d_actor_loss_scale;
    f.td_scale = L->td_loss_scale; f.wd_scale = L->wd_scale;
    f.actor_losses = o->actor_losses; f.critic_losses = o->critic_losses;
    f.disc_loss_part = w.disc_loss; f.disc_tiles_b = tiles;
    f.ent_scale = L->ent_reg_scale; f.gp_coeff = 10.f; f.gp = w.gp_out; f.disc_losses = o->d_losses;
    return fused::launch_finalize(f, st);
}

int samx_step_apply(const samx_learner* L, int64_t B, int mode, float grad_scale, void* ws, size_t ws_bytes,
                    void* stream) {
    int rc = check_learner(L, "samx_step_apply");
    if (rc != SAMX_OK) return rc;
    SAMX_REQUIRE(ws && B > 0, "samx_step_apply: null workspace or bad batch");
    SAMX_REQUIRE((mode & ~7) == 0 && (mode & 3) != 0, "samx_step_apply: mode must use bits 0-2");
    SAMX_REQUIRE(!(mode & 1) || L->clipped, "samx_step_apply: mode bit 0 needs learner.clipped");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    StepWs w = carve(ar, L, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_step_apply: workspace %zu < %zu", ws_bytes, ar.off);
    fused::ApplyArgs f = make_apply(L, w, mode & 3, grad_scale);
    if (mode & 4) {  // gradients changed since samx_step_grads (all-reduce): recompute the norms
        rc = fused::launch_seg_sumsq(f, st);
        if (rc != SAMX_OK) return rc;
    }
    return fused::launch_apply(f, st);
}

}  // extern "C"
