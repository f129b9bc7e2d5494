// samx_step_grads / samx_step_apply: one SAM update step's gradients and optimiser pass for the whole
// learner (discriminator + actor + critic), choosing the whole-network fused kernels when the
// configuration allows (64x64 nets, batch a multiple of 16) and the layer-wise MFMA path otherwise.
//   reference sequence: sam.py:369-433 (D update, then mu.train(), then update_target_net())
#include "samx_common.h"
#include "fused_small.h"
#include "nets.h"
#include "fused_wide.h"
#include "tree_dev.h"
#include "rng.h"
#include <stdlib.h>
#include <mutex>
#include "../../include/samx.h"

using namespace samx;

namespace {

struct StepWs {
    float *agent_part, *disc_part, *agent_loss, *disc_loss, *gp_out, *sumsq;
    double* gp_part;
    void *nets_ws, *nets_ws2;  // layer-wise path: one per agent role (they run side by side)
    size_t nets_bytes;
    void* disc_ws;
    size_t disc_bytes;
    // layer-wise path only: the agent's gathered rows and the drawn labels (the discriminator's rows and
    // eps are resolved inside its first kernel)
    float *g_obs0, *g_acs, *g_rews, *g_dones, *g_obs1, *r_real, *r_fake;
    // n-step lookahead (memory.py:113-160) and return-statistics sums
    int64_t* n_end;
    float *n_rews, *n_dones, *n_len, *n_obs1, *n_tq;
    double *ret_add1, *ret_add2;
    void* stats_ws;  // layer-wise path: samx_d_input_stats scratch (inline d.obs_rms update)
    size_t stats_bytes;
    void* wide_red;  // 256x256 path: head slots + split partials of the grouped weight-gradient launch
    size_t wide_red_bytes;
    // whole-network path, reduction + optimiser pass in one launch (part bit 9): {tag, sum of squares} words of the
    // blocks and the launch counter behind them (fused_small.hip finalize_apply_kernel)
    unsigned long long* fa_slots;
};

// Two side streams per device for the layer-wise path: the discriminator, target+critic and actor
// roles of a step do not depend on each other, so they are enqueued on three streams that fork from
// and join back into the caller's stream (events only — this also works under stream capture, where
// the side streams become parallel branches of the captured graph).  Created on first use, which
// therefore must not happen while a capture is in progress (the host driver warms up first).
struct SideStreams {
    hipStream_t s[2];
    hipEvent_t fork, join[2], mid;
};
int side_streams(SideStreams** out) {
    static SideStreams* per_dev[64] = {nullptr};
    int dev = 0;
    SAMX_CHECK_HIP(hipGetDevice(&dev));
    SAMX_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    if (!per_dev[dev]) {
        SideStreams* ss = new SideStreams();
        for (int i = 0; i < 2; ++i) {
            SAMX_CHECK_HIP(hipStreamCreateWithFlags(&ss->s[i], hipStreamNonBlocking));
            SAMX_CHECK_HIP(hipEventCreateWithFlags(&ss->join[i], hipEventDisableTiming));
        }
        SAMX_CHECK_HIP(hipEventCreateWithFlags(&ss->fork, hipEventDisableTiming));
        SAMX_CHECK_HIP(hipEventCreateWithFlags(&ss->mid, hipEventDisableTiming));
        per_dev[dev] = ss;
    }
    *out = per_dev[dev];
    return SAMX_OK;
}

bool use_fused(const samx_learner* L, int64_t B);
bool use_wide(const samx_learner* L, int64_t B) {
    static int no_wide = -1;   // SAMX_NO_WIDE=1: 256x256 learners take the layer-wise path too (A/B comparisons)
    if (no_wide < 0) { const char* e = getenv("SAMX_NO_WIDE"); no_wide = e ? atoi(e) : 0; }
    return !no_wide && !use_fused(L, B) && samx::wide_step_supported(L, B);
}
bool use_fused(const samx_learner* L, int64_t B) {
    const bool ln = L->actor.layernorm != 0;
    return fused::supported(&L->actor, B) && fused::supported(&L->critic, B) &&
           fused::supported(&L->reward, B) && L->critic.in_dim == L->ob_dim + L->ac_dim &&
           L->reward.in_dim == L->critic.in_dim && (L->critic.layernorm != 0) == ln &&
           (L->reward.layernorm != 0) == ln &&
           (ln ? fused::fits_lds_ln(L->ob_dim, L->ac_dim, L->d_actor_loss_scale != 0.f)
               : fused::fits_lds(L->ob_dim, L->ac_dim));
}

StepWs carve(Arena& ar, const samx_learner* L, int64_t B) {
    StepWs w;
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const int64_t tiles = (B + fused::kRows - 1) / fused::kRows;
    const bool f = use_fused(L, B);
    w.agent_part = ar.take<float>(f ? tiles * (Pa + Pc) : 1);
    w.disc_part = ar.take<float>(f ? 3 * tiles * Pd : 1);
    w.agent_loss = ar.take<float>(tiles * fused::kLossStride);
    w.disc_loss = ar.take<float>(2 * tiles * 4);
    w.gp_out = ar.take<float>(4);
    w.sumsq = ar.take<float>(3 * fused::kMaxSegBlocks);
    w.gp_part = ar.take<double>(2 * tiles);
    w.nets_bytes = f ? 0 : samx_nets_ws_bytes(&L->actor, &L->critic, &L->reward, B);
    w.disc_bytes = f ? 0 : samx_disc_ws_bytes(&L->reward, B);
    w.nets_ws = ar.take<char>(w.nets_bytes);
    w.nets_ws2 = ar.take<char>(w.nets_bytes);
    w.disc_ws = ar.take<char>(w.disc_bytes);
    const int64_t ob = L->ob_dim, ac = L->ac_dim, n = f ? 0 : B;
    w.g_obs0 = ar.take<float>(n * ob); w.g_acs = ar.take<float>(n * ac);
    w.g_rews = ar.take<float>(n); w.g_dones = ar.take<float>(n); w.g_obs1 = ar.take<float>(n * ob);
    w.r_real = ar.take<float>(n); w.r_fake = ar.take<float>(n);
    const int64_t nn = L->n_step > 0 ? B : 0;   // both paths: lookahead outputs (the gathered n_obs1 rows
                                                // are only used by the layer-wise path)
    w.n_end = ar.take<int64_t>(nn);
    w.n_rews = ar.take<float>(nn); w.n_dones = ar.take<float>(nn); w.n_len = ar.take<float>(nn);
    w.n_obs1 = ar.take<float>(nn * ob); w.n_tq = ar.take<float>(nn);
    w.ret_add1 = ar.take<double>(4); w.ret_add2 = ar.take<double>(4);
    w.stats_bytes = f ? 0 : samx_d_input_stats_ws_bytes(B, L->ob_dim);
    w.stats_ws = ar.take<char>(w.stats_bytes);
    w.wide_red_bytes = (!f && samx::wide_step_supported(L, B)) ? samx::wide_reduce_ws_bytes(L, B) : 0;
    w.wide_red = ar.take<char>(w.wide_red_bytes);
    w.fa_slots = ar.take<unsigned long long>(f ? 3 * fused::kMaxSegBlocks + 1 : 0);
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
    SAMX_REQUIRE((L->ret_mean == nullptr) == (L->ret_std == nullptr), "%s: ret_mean/ret_std go together", who);
    SAMX_REQUIRE(!L->enable_popart || (L->ret_mean && L->ret_sums), "%s: PopArt needs ret_mean/ret_std/ret_sums",
                 who);
    SAMX_REQUIRE(L->n_step == 0 || L->n_step > 1, "%s: n_step must be 0 (off) or > 1", who);
    return SAMX_OK;
}

// segs: bit k set = optimiser segment k (0 actor, 1 critic, 2 reward net) takes part
fused::ApplyArgs make_apply(const samx_learner* L, const StepWs& w, int mode, float grad_scale, int segs = 7,
                            bool polyak = true, float* (*img)[2] = nullptr, float* (*row)[2] = nullptr) {
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const int64_t off[3] = {0, Pa, Pa + Pc}, n[3] = {Pa, Pc, Pd};
    const float lr[3] = {L->actor_lr, L->critic_lr, L->d_lr};
    fused::ApplyArgs f;
    f.nseg = 0;
    f.mode = mode;
    f.grad_scale = grad_scale;
    f.beta1 = L->beta1; f.beta2 = L->beta2; f.eps = L->adam_eps; f.tau = L->polyak;
    for (int k = 0; k < 3; ++k) {
        if (!(segs & (1 << k))) continue;
        fused::ApplySeg& s = f.seg[f.nseg++];
        s.param = L->params + off[k];
        s.m = L->adam_m + off[k];
        s.v = L->adam_v + off[k];
        s.targ = (k < 2 && polyak) ? L->targets + off[k] : nullptr;
        s.clipped = L->clipped ? L->clipped + off[k] : nullptr;
        s.grad = L->grads + off[k];
        s.n = n[k];
        s.nblocks = fused::seg_blocks(n[k]);
        s.sumsq = w.sumsq + k * fused::kMaxSegBlocks;
        s.nsumsq = s.nblocks;
        s.lr = lr[k];
        s.clip = L->clip_norm;
        s.step_dev = L->steps + k;
        if (img) {   // k-interleaved images of W1 / W2 of network k and (k < 2) of its target network
            const samx_mlp* dn = k == 0 ? &L->actor : (k == 1 ? &L->critic : &L->reward);
            s.img1 = img[k][0]; s.img2 = img[k][1];
            s.timg1 = k < 2 ? img[3 + k][0] : nullptr; s.timg2 = k < 2 ? img[3 + k][1] : nullptr;
            s.tr2 = row[k][1];
            s.tr1 = k == 0 ? nullptr : row[k][0];
            s.tr1_r0 = k == 1 ? L->ob_dim : 0; s.tr1_J = k == 2 ? 2 * samx::wide::kH : samx::wide::kH;
            s.img_n1 = (int64_t)dn->in_dim * dn->h1;
            s.img_o2 = s.img_n1 + dn->h1 + (dn->layernorm ? 2 * dn->h1 : 0);
        }
    }
    return f;
}

__global__ __launch_bounds__(256) void randoms_kernel(uint64_t seed, const int64_t* step_dev, int64_t step,
                                                      int64_t B, int ob, int ac, float* eps_ob, float* eps_ac,
                                                      float* real, float* fake) {
    const uint64_t st = step_dev ? (uint64_t)step_dev[0] : (uint64_t)step;
    const int64_t n_ob = B * ob, n_ac = B * ac;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < n_ob + n_ac + 2 * B;
         t += (int64_t)gridDim.x * 256) {
        if (t < n_ob) eps_ob[t] = philox_uniform(seed, st, RNG_EPS_OB, (uint64_t)t);
        else if (t < n_ob + n_ac) eps_ac[t - n_ob] = philox_uniform(seed, st, RNG_EPS_AC, (uint64_t)(t - n_ob));
        else if (t < n_ob + n_ac + B) {
            if (real) real[t - n_ob - n_ac] = real_label(philox_uniform(seed, st, RNG_REAL, (uint64_t)(t - n_ob - n_ac)));
        } else if (fake) {
            fake[t - n_ob - n_ac - B] = fake_label(philox_uniform(seed, st, RNG_FAKE, (uint64_t)(t - n_ob - n_ac - B)));
        }
    }
}

// float64 column sums of the discriminator's 2B input rows read in place, one workgroup.
// threads = rowlanes x dimpad; every row lane adds its rows in increasing order, lanes are then
// combined by a fixed binary tree (deterministic).
__device__ __forceinline__ void d_input_stats_body(int blk,
   
    const float* __restrict__ replay_obs0, int64_t ring_start, int64_t ring_len,
    const int64_t* __restrict__ p_idx, const float* __restrict__ pool_obs0,
    const int64_t* __restrict__ e_start, int64_t B, int dim, int dimpad, double* __restrict__ addvec,
    int commit, double* __restrict__ sums, float* __restrict__ mean_out, float* __restrict__ std_out,
    int64_t idx_stride, int64_t addvec_stride) {
    extern __shared__ double shd[];  // [2][1024]
    const int tid = threadIdx.x;
    // one workgroup per step of a multi-step launch (blockIdx.x = step; a single step has stride 0)
    if (p_idx) p_idx += blk * idx_stride;
    if (e_start) e_start += blk * idx_stride;
    addvec += blk * addvec_stride;
    const int col = tid % dimpad, lane = tid / dimpad, lanes = 1024 / dimpad;
    const int64_t e0 = e_start ? e_start[0] : 0;
    double a = 0.0, q = 0.0;
    if (col < dim) {
        // rows in batches of 16 (C3: every row lane's 16 rows in ONE round of index loads + one of data loads): all index loads, then all data loads, then the adds in row order
        for (int64_t rb = lane; rb < 2 * B; rb += (int64_t)lanes * 16) {
            const float* src[16];
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int64_t r = rb + (int64_t)u * lanes;
                src[u] = nullptr;
                if (r < B) {
                    int64_t pr = r;
                    if (p_idx) { pr = ring_start + p_idx[r]; if (pr >= ring_len) pr -= ring_len; }
                    src[u] = replay_obs0 + pr * dim + col;
                } else if (r < 2 * B) {
                    src[u] = pool_obs0 + (e0 + (r - B)) * dim + col;
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[u] ? *src[u] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (src[u]) {
                    const double x = (double)v[u];
                    a = __dadd_rn(a, x);
                    q = __dadd_rn(q, __dmul_rn(x, x));
                }
        }
    }
    shd[tid] = a;
    shd[1024 + tid] = q;
    __syncthreads();
    for (int s = lanes / 2; s > 0; s >>= 1) {
        if (lane < s) {
            shd[tid] = __dadd_rn(shd[tid], shd[tid + s * dimpad]);
            shd[1024 + tid] = __dadd_rn(shd[1024 + tid], shd[1024 + tid + s * dimpad]);
        }
        __syncthreads();
    }
    if (lane == 0 && col < dim) {
        const double ta = shd[tid], tq = shd[1024 + tid];
        addvec[col] = ta;
        addvec[dim + col] = tq;
        if (commit) {  // helpers/mpi_running_mean_std.py:31-45
            const double cnt = sums[2 * dim] + (double)(2 * B);
            const double s1 = __dadd_rn(sums[col], ta), s2 = __dadd_rn(sums[dim + col], tq);
            sums[col] = s1;
            sums[dim + col] = s2;
            const float mean = (float)(s1 / cnt), msq = (float)(s2 / cnt);
            mean_out[col] = mean;
            std_out[col] = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(mean, mean)), 1e-2f));
        }
    }
    __syncthreads();
    if (tid == 0) {
        addvec[2 * dim] = (double)(2 * B);
        if (commit) sums[2 * dim] = sums[2 * dim] + (double)(2 * B);
    }
}

__global__ __launch_bounds__(1024) void d_input_stats_kernel(
    const float* __restrict__ replay_obs0, int64_t ring_start, int64_t ring_len,
    const int64_t* __restrict__ p_idx, const float* __restrict__ pool_obs0,
    const int64_t* __restrict__ e_start, int64_t B, int dim, int dimpad, double* __restrict__ addvec,
    int commit, double* __restrict__ sums, float* __restrict__ mean_out, float* __restrict__ std_out,
    int64_t idx_stride, int64_t addvec_stride) {
    d_input_stats_body(blockIdx.x, replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, dim, dimpad,
                       addvec, commit, sums, mean_out, std_out, idx_stride, addvec_stride);
}

// The pending update given as TD errors instead of leaf values: the refresh itself (|td| + 1e-6 ->
// priority ** alpha with the host pow's bits, max_priority, the reference's asserts) runs at the head of
// workgroup 0, the leaves stay in LDS.
struct PerRefresh {
    const float* td;          // null: upd_leaf holds the leaf values
    double alpha, demos_eps;
    int64_t num_demos, num_entries;
    samx::PerState* state;
    const int64_t* geom;      // device {ring start, num_entries}: read instead of num_entries (tree_dev.h: SampleExt)
};

// PER row prologue: the pending priority update of the trees (workgroup 0) and the D-input sums of the
// row (workgroup 1) do not depend on each other — one launch instead of two dependent ones.
__global__ __launch_bounds__(1024) void per_row_prologue_kernel(
    double* __restrict__ ts, double* __restrict__ tm, int32_t* __restrict__ stamp, int64_t cap,
    const int64_t* __restrict__ upd_idx, const double* __restrict__ upd_leaf, int n_upd, int levels,
    const float* __restrict__ replay_obs0, int64_t ring_start, int64_t ring_len,
    const int64_t* __restrict__ p_idx, const float* __restrict__ pool_obs0,
    const int64_t* __restrict__ e_start, int64_t B, int dim, int dimpad, double* __restrict__ addvec,
    int64_t n_ent, const double* __restrict__ u, double beta, int64_t* __restrict__ idxs_out,
    double* __restrict__ iws_out, float* __restrict__ iws32_out, const int64_t* __restrict__ copy_src,
    int64_t* copy_dst, int64_t copy_n, int64_t skip_lo, int64_t skip_n, PerRefresh rf, int fast) {
    if (copy_src) {
        // the row's host-drawn words (pinned host memory) -> their device home, by BOTH workgroups (same
        // values to the same addresses; each then reads what it wrote itself): no separate upload node.
        // Words [skip_lo, skip_lo + skip_n) are left alone: an OUTPUT of this launch may live there (the
        // sampled indices), and the other workgroup's copy could land after it was written.
        for (int64_t t = threadIdx.x; t < copy_n; t += 1024)
            if (t < skip_lo || t >= skip_lo + skip_n) copy_dst[t] = copy_src[t];
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        // this workgroup's use of the launch's dynamic LDS (n_upd < 1024): the leaves, and with `fast` the
        // whole working set of the tree update (tree_dev.h: FastLds)
        extern __shared__ double leaf_sh[];
        samx::FastLds& F = *reinterpret_cast<samx::FastLds*>(leaf_sh);
        double* leaf_lds = fast ? F.cs : leaf_sh;
        const double* leaf = upd_leaf;
        if (rf.td) {
            samx::leaves_from_td_body(rf.td, upd_idx, n_upd, cap, rf.num_entries, rf.alpha, rf.num_demos,
                                      rf.demos_eps, leaf_lds, rf.state);
            leaf = leaf_lds;
        }
        if (fast) samx::tree_set_fast_body(ts, tm, cap, upd_idx, leaf, n_upd, levels, F);
        else samx::tree_set_body(ts, tm, stamp, cap, upd_idx, leaf, n_upd, levels);
        if (u) {   // ... and, on the updated trees, this row's stratified sample (memory.py:266-309)
            __syncthreads();
            if (fast)
                samx::per_sample_body(ts, tm, cap, n_ent, u, B, beta, idxs_out, iws_out, iws32_out, 0, B, F.top_s,
                                      2 * cap < samx::kFastTop ? 2 * cap : samx::kFastTop);
            else
                samx::per_sample_body(ts, tm, cap, n_ent, u, B, beta, idxs_out, iws_out, iws32_out, 0, B);
        }
    } else {
        d_input_stats_body(0, replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, dim, dimpad, addvec,
                           0, nullptr, nullptr, nullptr, 0, 0);
    }
}

// The same head of a prioritized row in TWO launches that fill more of the chip (tree_dev.h: "spread over
// several workgroups").  Launch 1: G subtree workgroups refresh + walk the lower levels, one more workgroup
// adds the row's D-input sums.  Launch 2: S workgroups each rebuild the top in LDS (workgroup 0 stores it)
// and draw B / S of the row's samples.
__global__ __launch_bounds__(1024) void per_refresh_walk_kernel(
    double* __restrict__ ts, double* __restrict__ tm, int64_t cap, const int64_t* __restrict__ upd_idx, int n_upd,
    int levels, PerRefresh rf, int G, const float* __restrict__ replay_obs0, int64_t ring_start, int64_t ring_len,
    const int64_t* __restrict__ p_idx, const float* __restrict__ pool_obs0, const int64_t* __restrict__ e_start,
    int64_t B, int dim, int dimpad, double* __restrict__ addvec) {
    extern __shared__ double walk_lds[];
    if ((int)blockIdx.x < G) {
        const int64_t sub = cap / G, lo = (int64_t)blockIdx.x * sub;
        if (rf.geom) rf.num_entries = rf.geom[1];
        samx::tree_walk_sub_body(ts, tm, cap, upd_idx, rf.td, n_upd, levels, lo, lo + sub, rf.num_entries, rf.alpha,
                                 rf.num_demos, rf.demos_eps, rf.state, *reinterpret_cast<samx::WalkLds*>(walk_lds));
    } else {
        d_input_stats_body(0, replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, dim, dimpad, addvec,
                           0, nullptr, nullptr, nullptr, 0, 0);
    }
}

__global__ __launch_bounds__(1024) void per_top_sample_kernel(
    double* __restrict__ ts, double* __restrict__ tm, int64_t cap, int levels, int64_t n_ent,
    const double* __restrict__ u, int64_t B, double beta, int64_t* __restrict__ idxs_out,
    double* __restrict__ iws_out, float* __restrict__ iws32_out, int64_t per, samx::SampleExt x) {
    extern __shared__ double top_lds[];   // [2][kFastTop]
    double* top_s = top_lds;
    double* top_m = top_lds + samx::kFastTop;
    samx::tree_top_body(ts, tm, cap, levels, top_s, top_m, blockIdx.x == 0);
    const int64_t topn = 2 * (cap >> samx::fast_lb(levels));
    samx::per_sample_body(ts, tm, cap, n_ent, u, B, beta, idxs_out, iws_out, iws32_out, blockIdx.x * per, per, top_s,
                          topn, top_m, x);
}

// two device arrays -> pinned host memory in one launch (the TD errors and sampled indices a PER row
// hands back; a graph memcpy node costs 4-5 us each)
__global__ void to_host_pair_kernel(const float* __restrict__ a, float* __restrict__ a_host,
                                    const int64_t* __restrict__ b, int64_t* __restrict__ b_host, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { a_host[i] = a[i]; b_host[i] = b[i]; }
}

// Large inputs (Humanoid-shaped: 2B x 376): the same sums in two launches.  Stage 1: workgroup g adds
// rows g, g+G, ... (columns across the threads, coalesced) in float64; stage 2: one wave per column
// adds the G partials (lane l takes l, l+64, ... in order, then a fixed xor tree) and commits.
constexpr int kStatGroupsMax = 128;

__global__ __launch_bounds__(256) void d_input_partials_kernel(
    const float* __restrict__ replay_obs0, int64_t ring_start, int64_t ring_len,
    const int64_t* __restrict__ p_idx, const float* __restrict__ pool_obs0,
    const int64_t* __restrict__ e_start, int64_t B, int dim, double* __restrict__ part,
    const double* __restrict__ sums, int64_t idx_stride, int64_t part_stride) {
    const int G = gridDim.x, g = blockIdx.x;
    // blockIdx.z: the step of a multi-step launch (a single step has strides 0)
    if (p_idx) p_idx += blockIdx.z * idx_stride;
    if (e_start) e_start += blockIdx.z * idx_stride;
    part += blockIdx.z * part_stride;
    const int64_t e0 = e_start ? e_start[0] : 0;
    // snapshot of the running count for the commit kernel (slot behind the partials)
    if (sums && g == 0 && blockIdx.y == 0 && threadIdx.x == 0) part[(int64_t)G * 2 * dim] = sums[2 * dim];
    {   // blockIdx.y: the chunk of 256 columns
        const int col = blockIdx.y * 256 + threadIdx.x;
        double a = 0.0, q = 0.0;
        // The rows of this group, sixteen per batch: their storage rows are resolved first by the first sixteen
        // threads (ONE dependent index -> row round trip per batch instead of one per row and thread), then every
        // thread has its sixteen row loads in flight; added in increasing row order.
        constexpr int U = 16;
        __shared__ int64_t srow[U];
        for (int64_t rb = g; rb < 2 * B; rb += (int64_t)G * U) {
            __syncthreads();   // the previous batch's rows have been read
            if (threadIdx.x < U) {
                const int64_t r = rb + (int64_t)threadIdx.x * G;
                int64_t sr = -1;
                if (r < 2 * B) {
                    if (r < B) {
                        sr = r;
                        if (p_idx) { sr = ring_start + p_idx[r]; if (sr >= ring_len) sr -= ring_len; }
                    } else {
                        sr = e0 + (r - B);
                    }
                }
                srow[threadIdx.x] = sr;
            }
            __syncthreads();
            float v[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = rb + (int64_t)u * G;
                ok[u] = r < 2 * B && col < dim;
                v[u] = 0.f;
                if (ok[u]) v[u] = (r < B ? replay_obs0 : pool_obs0)[srow[u] * dim + col];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
                    const double x = (double)v[u];
                    a = __dadd_rn(a, x);
                    q = __dadd_rn(q, __dmul_rn(x, x));
                }
        }
        if (col < dim) {
            part[((int64_t)g * 2) * dim + col] = a;
            part[((int64_t)g * 2 + 1) * dim + col] = q;
        }
    }
}

// One wave per column.  The running count every column divides by is the snapshot the partials kernel took
// (the launch before), so column 0 can advance the count in place without a third launch.
__global__ __launch_bounds__(64) void d_input_commit_kernel(const double* __restrict__ part, int G, int64_t B,
                                                            int dim, double* __restrict__ addvec, int commit,
                                                            double* __restrict__ sums,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ std_out, int64_t part_stride,
                                                            int64_t addvec_stride) {
    const int col = blockIdx.x, lane = threadIdx.x;
    part += blockIdx.y * part_stride;       // blockIdx.y: the step of a multi-step launch
    addvec += blockIdx.y * addvec_stride;
    double a = 0.0, q = 0.0;
    for (int g = lane; g < G; g += 64) {
        a = __dadd_rn(a, part[((int64_t)g * 2) * dim + col]);
        q = __dadd_rn(q, part[((int64_t)g * 2 + 1) * dim + col]);
    }
    for (int o = 32; o > 0; o >>= 1) {
        a = __dadd_rn(a, __shfl_xor(a, o, 64));
        q = __dadd_rn(q, __shfl_xor(q, o, 64));
    }
    if (lane == 0) {
        addvec[col] = a;
        addvec[dim + col] = q;
        if (col == 0) addvec[2 * dim] = (double)(2 * B);
        if (commit) {  // helpers/mpi_running_mean_std.py:31-45
            const double cnt = part[(int64_t)G * 2 * dim] + (double)(2 * B);   // old count: the snapshot
            const double s1 = __dadd_rn(sums[col], a), s2 = __dadd_rn(sums[dim + col], q);
            sums[col] = s1;
            sums[dim + col] = s2;
            const float mean = (float)(s1 / cnt), msq = (float)(s2 / cnt);
            mean_out[col] = mean;
            std_out[col] = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(mean, mean)), 1e-2f));
            if (col == 0) sums[2 * dim] = cnt;
        }
    }
}

int stat_groups(int64_t B) {
    int64_t g = (2 * B) / 16;
    return (int)(g < 1 ? 1 : (g > kStatGroupsMax ? kStatGroupsMax : g));
}

// ---- samx_d_input_stats_static: the per-group float64 partials in library-owned device memory --------
// For callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc).
// `scratch` is the HOST-SIDE NAME of a device array: it is never passed to a kernel or a memset.  Its
// device address is resolved once per device with hipGetSymbolAddress (each GPU holds its own copy at its
// own address) and only that pointer is used afterwards.
constexpr int kStaticObMax = 512;   // widest observation the static variant takes
constexpr size_t kStaticScratchBytes = (size_t)kStatGroupsMax * 2 * kStaticObMax * sizeof(double) + 256;  // 1 MiB + 256
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

int static_scratch(void** out, hipStream_t st) {
    static void* per_dev[64] = {nullptr};
    static std::mutex mu;
    int dev = 0;
    SAMX_CHECK_HIP(hipGetDevice(&dev));
    SAMX_REQUIRE(dev >= 0 && dev < 64, "device ordinal %d out of range", dev);
    if (st) {
        // the array used is the CURRENT device's copy: a stream of another GPU would be handed an address of this
        // one.  (A query, legal under capture; if the runtime cannot answer, the caller's contract stands.)
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(st, &sdev) == hipSuccess)
            SAMX_REQUIRE((int)sdev == dev, "samx_d_input_stats_static: the stream belongs to device %d, the current "
                         "device is %d (make the stream's device current first)", (int)sdev, dev);
        else
            (void)hipGetLastError();
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!per_dev[dev]) {
        void* ptr = nullptr;
        SAMX_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        SAMX_REQUIRE(ptr, "hipGetSymbolAddress returned a null address for the static scratch");
        per_dev[dev] = ptr;
    }
    *out = per_dev[dev];
    return SAMX_OK;
}

int d_input_stats_impl(const char* who, const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                       const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start, int64_t B,
                       int ob_dim, double* addvec, int commit, double* sums, float* mean_out, float* std_out,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    SAMX_REQUIRE(replay_obs0 && pool_obs0 && addvec && B > 0 && ob_dim > 0, "%s: bad arguments", who);
    SAMX_REQUIRE(ob_dim <= 1024, "%s: ob_dim %d > 1024", who, ob_dim);
    SAMX_REQUIRE(!p_idx || (ring_len > 0 && ring_start >= 0 && ring_start < ring_len), "%s: bad ring geometry", who);
    SAMX_REQUIRE(!commit || (sums && mean_out && std_out), "%s: commit needs sums/mean/std", who);
    if (2 * B * (int64_t)ob_dim > 16384) {  // two-stage form
        const int G = stat_groups(B);
        SAMX_REQUIRE(ws, "%s: this size needs a workspace (samx_d_input_stats_ws_bytes)", who);
        if (ws_bytes < samx_d_input_stats_ws_bytes(B, ob_dim))
            return fail(SAMX_ERR_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes,
                        samx_d_input_stats_ws_bytes(B, ob_dim));
        double* part = (double*)ws;
        d_input_partials_kernel<<<dim3(G, (ob_dim + 255) / 256), 256, 0, st>>>(
            replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, ob_dim, part, commit ? sums : nullptr,
            0, 0);
        SAMX_LAUNCH_CHECK();
        d_input_commit_kernel<<<ob_dim, 64, 0, st>>>(part, G, B, ob_dim, addvec, commit, sums, mean_out, std_out,
                                                     0, 0);
        SAMX_LAUNCH_CHECK();
        return SAMX_OK;
    }
    int dimpad = 1;
    while (dimpad < ob_dim) dimpad *= 2;
    d_input_stats_kernel<<<1, 1024, 2 * 1024 * sizeof(double), st>>>(
        replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, ob_dim, dimpad, addvec, commit, sums,
        mean_out, std_out, 0, 0);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // namespace

extern "C" {

int samx_disc_randoms(uint64_t rng_seed, int64_t step, int64_t B, int ob_dim, int ac_dim, float* eps_ob,
                      float* eps_ac, float* real_labels, float* fake_labels, void* stream) {
    SAMX_REQUIRE(eps_ob && eps_ac && B > 0 && ob_dim > 0 && ac_dim > 0 && step >= 0,
                 "samx_disc_randoms: bad arguments");
    const int64_t n = B * (ob_dim + ac_dim + 2);
    randoms_kernel<<<(unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256), 256, 0,
                     (hipStream_t)stream>>>(rng_seed, nullptr, step, B, ob_dim, ac_dim, eps_ob, eps_ac,
                                            real_labels, fake_labels);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

size_t samx_d_input_stats_ws_bytes(int64_t B, int ob_dim) {
    if (B <= 0 || ob_dim <= 0) return 0;
    return (size_t)stat_groups(B) * 2 * ob_dim * sizeof(double) + 256;
}

int samx_d_input_stats(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                       const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start, int64_t B,
                       int ob_dim, double* addvec, int commit, double* sums, float* mean_out,
                       float* std_out, void* ws, size_t ws_bytes, void* stream) {
    return d_input_stats_impl("samx_d_input_stats", replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start,
                              B, ob_dim, addvec, commit, sums, mean_out, std_out, ws, ws_bytes,
                              (hipStream_t)stream);
}

int samx_d_input_stats_static_max_ob(void) { return kStaticObMax; }

int samx_d_input_stats_static(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                              const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start, int64_t B,
                              int ob_dim, double* addvec, int commit, double* sums, float* mean_out,
                              float* std_out, void* stream) {
    SAMX_REQUIRE(ob_dim <= kStaticObMax,
                 "samx_d_input_stats_static: ob_dim %d exceeds the static scratch (at most %d columns; use "
                 "samx_d_input_stats with a workspace)", ob_dim, kStaticObMax);
    SAMX_REQUIRE(B > 0 && samx_d_input_stats_ws_bytes(B, ob_dim) <= kStaticScratchBytes,
                 "samx_d_input_stats_static: B %lld x ob_dim %d needs %zu scratch bytes, the static scratch holds %zu",
                 (long long)B, ob_dim, samx_d_input_stats_ws_bytes(B, ob_dim), kStaticScratchBytes);
    void* ptr = nullptr;   // this device's copy of the array (never the symbol itself)
    int rc = static_scratch(&ptr, (hipStream_t)stream);
    if (rc != SAMX_OK) return rc;
    // every word of the partials that the commit kernel reads is written by the partials kernel of the
    // same call first: nothing has to start at zero, the previous call's contents are simply overwritten
    return d_input_stats_impl("samx_d_input_stats_static", replay_obs0, ring_start, ring_len, p_idx, pool_obs0,
                              e_start, B, ob_dim, addvec, commit, sums, mean_out, std_out, ptr,
                              kStaticScratchBytes, (hipStream_t)stream);
}

static int per_row_prologue_impl(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                          const int64_t* upd_idx, const double* upd_leaf, const PerRefresh& rf, int64_t n_upd,
                          const float* replay_obs0, int64_t ring_start, int64_t ring_len, const int64_t* p_idx,
                          const float* pool_obs0, const int64_t* e_start, int64_t B, int ob_dim, double* addvec,
                          int64_t num_entries, const double* u, double beta, int64_t* idxs_out, double* iws_out,
                          float* iws_f32_out, const int64_t* copy_src, int64_t* copy_dst, int64_t copy_n,
                          int64_t copy_skip_lo, int64_t copy_skip_n, void* stream,
                          const samx::SampleExt& sx = samx::SampleExt()) {
    SAMX_REQUIRE(sum_tree && min_tree && stamp && upd_idx && (upd_leaf || rf.td), "samx_per_row_prologue: null tree argument");
    if (sx.geom) {   // geometry in device words: the by-value entry count is not used (and cannot be checked here)
        SAMX_REQUIRE(rf.td && u && cap <= (int64_t(1) << 30) && rf.geom == sx.geom && sx.limit > 0 && sx.limit <= cap,
                     "samx_per_row_prologue_td_rows: needs the TD form, a sample, cap <= 2^30 and ring_len in [1, cap]");
        num_entries = cap;
    }
    SAMX_REQUIRE(!rf.td || (rf.state && rf.num_entries >= 0 && rf.num_entries <= cap && samx_pow::in_domain(1.0, rf.alpha)),
                 "samx_per_row_prologue_td: needs the state words, num_entries in [0, cap] and alpha inside the pow domain");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_per_row_prologue: cap %lld is not a power of two",
                 (long long)cap);
    SAMX_REQUIRE(n_upd > 0 && n_upd < 1024, "samx_per_row_prologue: 1..1023 priority updates (got %lld)",
                 (long long)n_upd);
    const bool sums = replay_obs0 != nullptr;   // (the TD form may leave the D-input sums to the launch head)
    SAMX_REQUIRE(B > 0 && (sums ? (pool_obs0 && addvec && ob_dim > 0) : (rf.td != nullptr && u != nullptr)),
                 "samx_per_row_prologue: bad arguments");
    SAMX_REQUIRE(!sums || 2 * B * (int64_t)ob_dim <= 16384, "samx_per_row_prologue: batch too large (2*B*ob = %lld)",
                 (long long)(2 * B * ob_dim));
    SAMX_REQUIRE(!p_idx || (ring_len > 0 && ring_start >= 0 && ring_start < ring_len),
                 "samx_per_row_prologue: bad ring geometry");
    SAMX_REQUIRE(!copy_src || (copy_dst && copy_n > 0 && copy_n <= (1 << 20) && copy_skip_lo >= 0 &&
                               copy_skip_n >= 0),
                 "samx_per_row_prologue: copy_src needs copy_dst, 1 <= copy_n <= 2^20 and a non-negative skip range");
    SAMX_REQUIRE(!(copy_src && idxs_out) || idxs_out + B <= copy_dst || idxs_out >= copy_dst + copy_n ||
                     (idxs_out >= copy_dst + copy_skip_lo && idxs_out + B <= copy_dst + copy_skip_lo + copy_skip_n),
                 "samx_per_row_prologue: idxs_out inside copy_dst must lie in the skipped range");
    SAMX_REQUIRE(!u || (idxs_out && iws_out && num_entries >= 2 && num_entries <= cap && beta > 0),
                 "samx_per_row_prologue: sampling needs idxs_out, iws_out, num_entries in [2, cap] and beta > 0");
    int levels = 0;
    while ((int64_t(1) << levels) < cap) ++levels;
    int dimpad = 1;
    while (dimpad < ob_dim) dimpad *= 2;
    if (rf.td && u && cap <= (int64_t(1) << 30)) {
        int G = 16;
        while (G > 1 && (cap / G) < (int64_t(1) << samx::fast_lb(levels))) G /= 2;
        while (G > cap) G /= 2;
        static LdsGrant grant1;
        if (int rc = grant1.ensure((const void*)per_refresh_walk_kernel, sizeof(samx::WalkLds))) return rc;
        per_refresh_walk_kernel<<<G + (sums ? 1 : 0), 1024, sizeof(samx::WalkLds), (hipStream_t)stream>>>(
            sum_tree, min_tree, cap, upd_idx, (int)n_upd, levels, rf, G, replay_obs0, ring_start, ring_len, p_idx,
            pool_obs0, e_start, B, ob_dim, dimpad, addvec);
        SAMX_LAUNCH_CHECK();
        int S = (int)((B + 31) / 32);
        if (S > 16) S = 16;
        const int64_t per = (B + S - 1) / S;
        const size_t lds2 = 2 * samx::kFastTop * sizeof(double);
        static LdsGrant grant2;
        if (int rc = grant2.ensure((const void*)per_top_sample_kernel, lds2)) return rc;
        per_top_sample_kernel<<<S, 1024, lds2, (hipStream_t)stream>>>(sum_tree, min_tree, cap, levels, num_entries, u,
                                                                     B, beta, idxs_out, iws_out, iws_f32_out, per, sx);
        SAMX_LAUNCH_CHECK();
        return SAMX_OK;
    }
    const int fast = cap <= (int64_t(1) << 30) ? 1 : 0;   // n_upd < 1024 was required above (tree_set_fast_ok)
    const size_t lds = fast ? sizeof(samx::FastLds) : 2 * 1024 * sizeof(double);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)per_row_prologue_kernel, lds, 64 * 1024)) return rc;
    per_row_prologue_kernel<<<2, 1024, lds, (hipStream_t)stream>>>(
        sum_tree, min_tree, stamp, cap, upd_idx, upd_leaf, (int)n_upd, levels, replay_obs0, ring_start, ring_len,
        p_idx, pool_obs0, e_start, B, ob_dim, dimpad, addvec, num_entries, u, beta, idxs_out, iws_out, iws_f32_out,
        copy_src, copy_dst, copy_n, copy_skip_lo, copy_skip_n, rf, fast);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_per_row_prologue(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                          const int64_t* upd_idx, const double* upd_leaf, int64_t n_upd,
                          const float* replay_obs0, int64_t ring_start, int64_t ring_len, const int64_t* p_idx,
                          const float* pool_obs0, const int64_t* e_start, int64_t B, int ob_dim, double* addvec,
                          int64_t num_entries, const double* u, double beta, int64_t* idxs_out, double* iws_out,
                          float* iws_f32_out, const int64_t* copy_src, int64_t* copy_dst, int64_t copy_n,
                          int64_t copy_skip_lo, int64_t copy_skip_n, void* stream) {
    PerRefresh rf{};
    return per_row_prologue_impl(sum_tree, min_tree, stamp, cap, upd_idx, upd_leaf, rf, n_upd, replay_obs0, ring_start,
                                 ring_len, p_idx, pool_obs0, e_start, B, ob_dim, addvec, num_entries, u, beta,
                                 idxs_out, iws_out, iws_f32_out, copy_src, copy_dst, copy_n, copy_skip_lo,
                                 copy_skip_n, stream);
}

int samx_per_row_prologue_td(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                             const int64_t* upd_idx, const float* upd_td, int64_t n_upd, double alpha,
                             int64_t num_demos, double demos_eps, void* per_state,
                             const float* replay_obs0, int64_t ring_start, int64_t ring_len, const int64_t* p_idx,
                             const float* pool_obs0, const int64_t* e_start, int64_t B, int ob_dim, double* addvec,
                             int64_t num_entries, const double* u, double beta, int64_t* idxs_out, double* iws_out,
                             float* iws_f32_out, void* stream) {
    SAMX_REQUIRE(upd_td && per_state, "samx_per_row_prologue_td: null pointer");
    PerRefresh rf{upd_td, alpha, demos_eps, num_demos, num_entries, (samx::PerState*)per_state, nullptr};
    return per_row_prologue_impl(sum_tree, min_tree, stamp, cap, upd_idx, nullptr, rf, n_upd, replay_obs0, ring_start,
                                 ring_len, p_idx, pool_obs0, e_start, B, ob_dim, addvec, num_entries, u, beta,
                                 idxs_out, iws_out, iws_f32_out, nullptr, nullptr, 0, 0, 0, stream);
}

int samx_per_row_prologue_td_rows(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                                  const int64_t* upd_idx, const float* upd_td, int64_t n_upd, double alpha,
                                  int64_t num_demos, double demos_eps, void* per_state, const int64_t* geom,
                                  int64_t ring_len, const double* u, int64_t B, double beta, int64_t* idxs_out,
                                  double* iws_out, float* iws_f32_out, int64_t* rows_out, int64_t* idx_copy_out,
                                  void* stream) {
    SAMX_REQUIRE(upd_td && per_state && geom, "samx_per_row_prologue_td_rows: null pointer");
    PerRefresh rf{upd_td, alpha, demos_eps, num_demos, 0, (samx::PerState*)per_state, geom};
    samx::SampleExt sx;
    sx.geom = geom; sx.limit = ring_len; sx.rows_out = rows_out; sx.idx_copy = idx_copy_out;
    return per_row_prologue_impl(sum_tree, min_tree, stamp, cap, upd_idx, nullptr, rf, n_upd, nullptr, 0, 0, nullptr,
                                 nullptr, nullptr, B, 0, nullptr, 0, u, beta, idxs_out, iws_out, iws_f32_out,
                                 nullptr, nullptr, 0, 0, 0, stream, sx);
}

int samx_to_host_pair(const float* a, float* a_host, const int64_t* b, int64_t* b_host, int64_t n, void* stream) {
    SAMX_REQUIRE(a && a_host && b && b_host && n > 0, "samx_to_host_pair: bad arguments");
    to_host_pair_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, a_host, b, b_host, n);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_d_input_sums_many(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                           const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start,
                           int64_t idx_stride, int64_t B, int ob_dim, double* addvec,
                           int64_t addvec_stride, int64_t K, void* stream) {
    SAMX_REQUIRE(replay_obs0 && pool_obs0 && addvec && B > 0 && ob_dim > 0 && K > 0 && K <= 65535,
                 "samx_d_input_sums_many: bad arguments");
    SAMX_REQUIRE(2 * B * (int64_t)ob_dim <= 16384, "samx_d_input_sums_many: batch too large (2*B*ob = %lld)",
                 (long long)(2 * B * ob_dim));
    SAMX_REQUIRE(!p_idx || (ring_len > 0 && ring_start >= 0 && ring_start < ring_len),
                 "samx_d_input_sums_many: bad ring geometry");
    SAMX_REQUIRE(idx_stride >= 0 && addvec_stride >= 2 * ob_dim + 1, "samx_d_input_sums_many: bad strides");
    int dimpad = 1;
    while (dimpad < ob_dim) dimpad *= 2;
    d_input_stats_kernel<<<(unsigned)K, 1024, 2 * 1024 * sizeof(double), (hipStream_t)stream>>>(
        replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, ob_dim, dimpad, addvec, 0, nullptr,
        nullptr, nullptr, idx_stride, addvec_stride);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_d_input_sums_many_ws(const float* replay_obs0, int64_t ring_start, int64_t ring_len,
                              const int64_t* p_idx, const float* pool_obs0, const int64_t* e_start,
                              int64_t idx_stride, int64_t B, int ob_dim, double* addvec, int64_t addvec_stride,
                              int64_t K, void* ws, size_t ws_bytes, void* stream) {
    if (2 * B * (int64_t)ob_dim <= 16384)
        return samx_d_input_sums_many(replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, idx_stride, B,
                                      ob_dim, addvec, addvec_stride, K, stream);
    SAMX_REQUIRE(replay_obs0 && pool_obs0 && addvec && B > 0 && ob_dim > 0 && ob_dim <= 1024 && K > 0 && K <= 65535,
                 "samx_d_input_sums_many_ws: bad arguments");
    SAMX_REQUIRE(!p_idx || (ring_len > 0 && ring_start >= 0 && ring_start < ring_len),
                 "samx_d_input_sums_many_ws: bad ring geometry");
    SAMX_REQUIRE(idx_stride >= 0 && addvec_stride >= 2 * ob_dim + 1, "samx_d_input_sums_many_ws: bad strides");
    const size_t per = samx_d_input_stats_ws_bytes(B, ob_dim);
    SAMX_REQUIRE(ws, "samx_d_input_sums_many_ws: this size needs a workspace (K * samx_d_input_stats_ws_bytes)");
    if (ws_bytes < per * (size_t)K)
        return fail(SAMX_ERR_WORKSPACE, "samx_d_input_sums_many_ws: workspace %zu < %zu", ws_bytes, per * (size_t)K);
    const int G = stat_groups(B);
    const int64_t pstride = (int64_t)(per / sizeof(double));
    hipStream_t st = (hipStream_t)stream;
    d_input_partials_kernel<<<dim3(G, (ob_dim + 255) / 256, (unsigned)K), 256, 0, st>>>(
        replay_obs0, ring_start, ring_len, p_idx, pool_obs0, e_start, B, ob_dim, (double*)ws, nullptr, idx_stride,
        pstride);
    SAMX_LAUNCH_CHECK();
    d_input_commit_kernel<<<dim3(ob_dim, (unsigned)K), 64, 0, st>>>((const double*)ws, G, B, ob_dim, addvec, 0,
                                                                   nullptr, nullptr, nullptr, pstride,
                                                                   addvec_stride);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_td_errors_rows(const samx_mlp* dc, const float* pc, const float* obs0, const float* acs,
                        const int64_t* idx, int64_t ring_start, int64_t ring_len, const float* tc1s,
                        const float* mean, const float* stdv, int64_t B, int ob_dim, float* td_out,
                        float* td_copy_out, int64_t* idx_copy_out, const samx_retnorm* ret, void* ws,
                        size_t ws_bytes, void* stream) {
    SAMX_REQUIRE(dc && pc && obs0 && acs && tc1s && td_out && B > 0 && ob_dim > 0 && ob_dim < dc->in_dim,
                 "samx_td_errors_rows: bad arguments");
    SAMX_REQUIRE((td_copy_out == nullptr) == (idx_copy_out == nullptr) && (!idx_copy_out || idx),
                 "samx_td_errors_rows: td_copy_out and idx_copy_out come together, and only with idx");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_td_errors_rows: mean/std must come together");
    SAMX_REQUIRE(!idx || (ring_len > 0 && ring_start >= 0 && ring_start < ring_len),
                 "samx_td_errors_rows: bad ring geometry");
    const bool ret_on = ret && ret->mean;
    if (fused::supported(dc, B) && dc->out_dim == 1 && fused::fits_lds(ob_dim, dc->in_dim - ob_dim))
        return (dc->layernorm ? fused::launch_td_ln : fused::launch_td)(
            dc, pc, obs0, acs, tc1s, mean, stdv, B, ob_dim, idx, ring_start, ring_len, td_out,
            ret_on ? ret->mean : nullptr, ret_on ? ret->std : nullptr, td_copy_out, idx_copy_out,
            (hipStream_t)stream);
    // general path: gather the rows, then the layer-wise pass
    const int ac = dc->in_dim - ob_dim;
    SAMX_REQUIRE(ws, "samx_td_errors_rows: this configuration needs a workspace");
    Arena ar(ws, ws_bytes);
    float* g_obs = ar.take<float>(idx ? B * ob_dim : 0);
    float* g_acs = ar.take<float>(idx ? B * ac : 0);
    const size_t nets_bytes = samx_nets_ws_bytes(dc, dc, nullptr, B);  // an upper bound for a critic-only pass
    char* nws = ar.take<char>(nets_bytes);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_td_errors_rows: workspace %zu < %zu", ws_bytes, ar.off);
    if (idx) {
        int rc = samx_replay_gather(obs0, acs, nullptr, nullptr, nullptr, ring_len, ring_start, ob_dim, ac, idx, B,
                                    g_obs, g_acs, nullptr, nullptr, nullptr, stream);
        if (rc != SAMX_OK) return rc;
        obs0 = g_obs; acs = g_acs;
    }
    int rc = samx_td_errors(dc, pc, obs0, acs, tc1s, mean, stdv, B, ob_dim, td_out, ret, nws, nets_bytes, stream);
    if (rc != SAMX_OK || !td_copy_out) return rc;
    return samx_to_host_pair(td_out, td_copy_out, idx, idx_copy_out, B, stream);
}

size_t samx_td_errors_rows_ws_bytes(const samx_mlp* dc, int64_t B) {
    if (!dc || B <= 0) return 0;
    return (size_t)B * dc->in_dim * sizeof(float) + 1024 + samx_nets_ws_bytes(dc, dc, nullptr, B);
}

int samx_fused_supported(const samx_learner* L, int64_t B) { return L && use_fused(L, B) ? 1 : 0; }

int samx_step_path(const samx_learner* L, int64_t B) {
    if (!L || B <= 0) return 0;
    if (use_fused(L, B)) return 1;
    return use_wide(L, B) ? 2 : 0;
}

size_t samx_step_ws_bytes(const samx_learner* L, int64_t B) {
    if (!L || B <= 0) return 0;
    Arena ar(nullptr, 0);
    carve(ar, L, B);
    return align_up(ar.off, 256) + 256;
}

int samx_step_grads(const samx_learner* L, const samx_batch* b, const samx_step_out* o, void* ws,
                    size_t ws_bytes, void* stream) {
    return samx_step_grads_part(L, b, o, ws, ws_bytes, 3, stream);
}

int samx_step_grads_part(const samx_learner* L, const samx_batch* b, const samx_step_out* o, void* ws,
                         size_t ws_bytes, int part, void* stream) {
    int rc = check_learner(L, "samx_step_grads");
    SAMX_REQUIRE((part & ~1023) == 0 && (part & 3) != 0 && (part & 12) != 12 && (part & 384) != 384,
                 "samx_step_grads_part: part uses bits 0-9, needs bit 0 or 1, cannot skip both roles, and bits 7 and 8 "
                 "exclude each other");
    SAMX_REQUIRE(!(part & 512) || ((part & 2) && !(part & (16 | 64)) && L && L->clipped),
                 "samx_step_grads_part: bit 9 (optimiser pass inside the reduction launch) needs bit 1, the counter "
                 "tick (no bit 6), no bit 4 and learner.clipped");
    const bool tick = !(part & 64);   // bit 6: an evaluation that no update follows (d.get_grads / d.get_losses)
    const bool do_d = !(part & 4), do_a = !(part & 8);   // roles of this call
    const bool t_only = part & 16, t_done = part & 32;   // PopArt with N processes: the call split at the all-reduce
    if (t_only || t_done) {
        SAMX_REQUIRE(!(t_only && t_done) && (part & 1) && do_a,
                     "samx_step_grads_part: bits 4 and 5 exclude each other and need bit 0 and the agent roles");
        SAMX_REQUIRE(L->ret_mean && L->enable_popart && b && b->ret_addvec,
                     "samx_step_grads_part: bits 4/5 are for return normalisation with PopArt and need "
                     "batch.ret_addvec");
    }
    const int segs = (do_a ? 3 : 0) | (do_d ? 4 : 0);
    if (rc != SAMX_OK) return rc;
    SAMX_REQUIRE(b && o && ws, "samx_step_grads: null argument");
    SAMX_REQUIRE(b->B > 0, "samx_step_grads: empty batch");
    SAMX_REQUIRE(b->p_obs && b->p_acs && b->e_obs && b->e_acs && b->obs0 && b->acs && b->rews &&
                     b->dones1 && b->obs1,
                 "samx_step_grads: missing batch input");
    SAMX_REQUIRE((b->eps_ob == nullptr) == (b->eps_ac == nullptr), "samx_step_grads: eps_ob/eps_ac go together");
    SAMX_REQUIRE(b->label_smoothing >= 0 && b->label_smoothing <= 2, "samx_step_grads: label_smoothing");
    SAMX_REQUIRE(!b->n_end == !b->n_rews && !b->n_end == !b->n_dones && !b->n_end == !b->n_len,
                 "samx_step_grads: n_end / n_rews / n_dones / n_len come together or not at all");
    SAMX_REQUIRE((!b->p_idx && !b->a_idx) ||
                     (b->ring_len > 0 && b->ring_start >= 0 && b->ring_start < b->ring_len),
                 "samx_step_grads: bad ring geometry");
    SAMX_REQUIRE(o->d_losses && o->actor_losses && o->critic_losses && o->targ_q,
                 "samx_step_grads: missing output buffer");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    StepWs w = carve(ar, L, b->B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_step_grads: workspace %zu < %zu", ws_bytes, ar.off);
    SAMX_REQUIRE(!(part & 512) || use_fused(L, b->B),
                 "samx_step_grads_part: bit 9 is for the whole-network path (samx_step_path 1)");
    const int64_t B = b->B;
    const int64_t Pa = samx_mlp_param_count(&L->actor), Pc = samx_mlp_param_count(&L->critic),
                  Pd = samx_mlp_param_count(&L->reward);
    const float* pa = L->params;
    const float* pc = L->params + Pa;
    const float* pd = L->params + Pa + Pc;
    const float* ta = L->targets;
    const float* tc = L->targets + Pa;

    SAMX_REQUIRE((b->d_rms_sums == nullptr) == (b->d_rms_addvec == nullptr),
                 "samx_step_grads: d_rms_sums/d_rms_addvec go together");
    const bool wide = use_wide(L, B);   // 256x256 learners (fused_wide.hip); statistics as on the fused path
    if (b->d_rms_sums) {
        SAMX_REQUIRE(L->d_mean && L->d_std, "samx_step_grads: inline statistics need learner.d_mean/d_std");
        // fused and 256x256 paths: d_rms_addvec holds the row's sums (computed ahead).  Layer-wise path: the sums are
        // computed HERE, on the discriminator's stream, and d_rms_addvec receives them.
        SAMX_REQUIRE(use_fused(L, B) || wide || (b->p_idx && b->e_start),
                     "samx_step_grads: inline statistics on the layer-wise path need p_idx and e_start");
    }
    if (!use_fused(L, B) && !wide && !(part & 1)) {  // reduction part only
        fused::ApplyArgs f = make_apply(L, w, 0, 1.f, segs);
        // (the counters advance in the same launch: nothing in it reads them)
        return fused::launch_seg_sumsq(f, st, (tick && do_a) ? L->steps : nullptr, (tick && do_a) ? L->steps + 1 : nullptr,
                                       (tick && do_d) ? L->steps + 2 : nullptr);
    }
    if (wide) {
        // 256x256 nets: role-fused chains (fused_wide.hip) + batch reductions, all on the caller's stream
        samx::WideExt we;
        we.images_current = (part & 128) != 0;
        we.pack_all = (part & 256) != 0;
        auto roles = [&](const samx::WideExt* e) {
            return samx::wide_step_roles(L, b, o, w.nets_ws, w.nets_ws2, w.nets_bytes, w.disc_ws, w.disc_bytes,
                                         w.wide_red, w.wide_red_bytes, w.sumsq, do_a, do_d, (part & 3) | (tick ? 0 : 4), st, e);
        };
        if (do_a && (L->ret_mean || L->n_step > 0)) {
            // sam_agent.py:634-672: [n-step lookahead] -> targets with the OLD return statistics / heads ->
            // ret_rms.update + PopArt -> losses with the new ones
            we.tqn = L->n_step > 0 ? (o->targ_q_n ? o->targ_q_n : w.n_tq) : nullptr;
            if ((part & 1) && L->n_step > 0 && !t_done) {
                SAMX_REQUIRE(b->a_idx && b->num_entries > 0,
                             "samx_step_grads: n-step returns need a_idx (ring rows) and num_entries");
                if (b->n_end) {   // done ahead by the caller (samx_nstep_lookahead_many)
                    we.n_idx = b->n_end; we.n_rews = b->n_rews; we.n_dones = b->n_dones; we.n_len = b->n_len;
                } else {
                    rc = samx_nstep_lookahead(b->rews, b->dones1, b->ring_len, b->ring_start, b->num_entries, b->a_idx,
                                              B, L->n_step, (double)L->gamma, w.n_end, w.n_rews, nullptr, w.n_dones,
                                              w.n_len, stream);
                    if (rc != SAMX_OK) return rc;
                    we.n_idx = w.n_end; we.n_rews = w.n_rews; we.n_dones = w.n_dones; we.n_len = w.n_len;
                }
            }
            if (part & 1) {
                if (!t_done) {
                    we.targets_only = true;
                    rc = roles(&we);
                    if (rc != SAMX_OK) return rc;
                    we.targets_only = false;
                    we.packed = !t_only;   // (the weight images written there serve the role launch below)
                }
                if (L->ret_mean && L->enable_popart) {
                    // sam_agent.py:662-672: batch sums, commit and rescale in ONE launch; N processes (bits 4/5):
                    // the local sums -> batch.ret_addvec [caller: all-reduce] commit of the summed vector + rescale
                    rc = fused::launch_ret_popart(o->targ_q, we.tqn, B, L->ret_sums, L->ret_mean, L->ret_std, 1,
                                                  L->params + Pa, L->targets + Pa, &L->critic,
                                                  t_only ? 1 : (t_done ? 2 : 0), b->ret_addvec, st);
                    if (rc != SAMX_OK) return rc;
                }
                if (t_only) return SAMX_OK;
            }
            we.use_ext = true;
        }
        return roles(&we);
    }
    if (!use_fused(L, B)) {
        // layer-wise path, role by role (same order as the reference: D first).  It works on dense
        // rows and materialised randoms, so indirection / in-kernel draws are resolved first.
        samx_batch dense = *b;
        const samx_batch* orig = b;
        const int ob = L->ob_dim, ac = L->ac_dim;
        // agent rows first, then fork: the target+critic and actor roles start while the caller's
        // stream still prepares the discriminator's inputs (statistics, gathers, randoms)
        if (b->a_idx && do_a) {
            rc = samx_replay_gather(b->obs0, b->acs, b->rews, b->dones1, b->obs1, b->ring_len, b->ring_start,
                                    ob, ac, b->a_idx, B, w.g_obs0, w.g_acs, w.g_rews, w.g_dones, w.g_obs1,
                                    stream);
            if (rc != SAMX_OK) return rc;
            dense.obs0 = w.g_obs0; dense.acs = w.g_acs; dense.rews = w.g_rews; dense.dones1 = w.g_dones;
            dense.obs1 = w.g_obs1;
        }
        static int serial = -1;  // SAMX_SERIAL_ROLES=1: everything on the caller's stream (profiling aid)
        if (serial < 0) { const char* e = getenv("SAMX_SERIAL_ROLES"); serial = e ? atoi(e) : 0; }
        hipStream_t s_tc = st, s_a = st;
        SideStreams* ss = nullptr;
        if (!serial && do_a && !t_only) {
            rc = side_streams(&ss);
            if (rc != SAMX_OK) return rc;
            SAMX_CHECK_HIP(hipEventRecord(ss->fork, st));
            SAMX_CHECK_HIP(hipStreamWaitEvent(ss->s[0], ss->fork, 0));
            SAMX_CHECK_HIP(hipStreamWaitEvent(ss->s[1], ss->fork, 0));
            s_tc = ss->s[0];
            s_a = ss->s[1];
        }
        if (b->d_rms_sums && do_d && !t_only) {  // d.obs_rms.update on this row's D batch (sam.py:389-390), committed
            rc = samx_d_input_stats(b->p_obs, b->ring_start, b->ring_len, b->p_idx, b->e_obs, b->e_start, B, ob,
                                    const_cast<double*>(b->d_rms_addvec), 1, b->d_rms_sums,
                                    const_cast<float*>(L->d_mean), const_cast<float*>(L->d_std), w.stats_ws,
                                    w.stats_bytes, stream);
            if (rc != SAMX_OK) return rc;
        }
        // the discriminator's rows stay where they are: its first kernel reads the ring rows / the pool
        // slice in place and draws the randoms itself (no gather, slice or random-number launch in front)
        samx::DiscSrc dsrc;
        dsrc.p_idx = b->p_idx; dsrc.ring_start = b->ring_start; dsrc.ring_len = b->ring_len;
        dsrc.e_start = b->e_start;
        if (!b->eps_ob) {
            dsrc.draw = true; dsrc.rng_seed = b->rng_seed; dsrc.step_dev = L->steps + 2;
            dsrc.real_out = b->label_smoothing >= 1 ? w.r_real : nullptr;
            dsrc.fake_out = b->label_smoothing == 2 ? w.r_fake : nullptr;
        }
        b = &dense;
        samx_retnorm retv{L->ret_mean, L->ret_std};
        const samx_retnorm* ret = L->ret_mean ? &retv : nullptr;
        const bool popart_or_stats = L->ret_mean && L->enable_popart;  // sam_agent.py:662 (update only then)
        // ---- target + critic role (side 0)
        const float* tcns = nullptr;
        const int64_t* nl_end = w.n_end;
        const float *nl_rews = w.n_rews, *nl_dones = w.n_dones, *nl_len = w.n_len;
        if (do_a) {
        double* add1 = orig->ret_addvec ? orig->ret_addvec : w.ret_add1;
        double* add2 = orig->ret_addvec ? orig->ret_addvec + 3 : w.ret_add2;
        if (L->n_step > 0) tcns = o->targ_q_n ? o->targ_q_n : w.n_tq;
        if (!t_done) {
        if (L->n_step > 0) {  // n-step lookahead from the sampled indices (memory.py:113-160)
            SAMX_REQUIRE(orig->a_idx && orig->num_entries > 0,
                         "samx_step_grads: n-step returns need a_idx (ring rows) and num_entries");
            if (!orig->n_end) {
                rc = samx_nstep_lookahead(orig->rews, orig->dones1, orig->ring_len, orig->ring_start,
                                          orig->num_entries, orig->a_idx, B, L->n_step, (double)L->gamma, w.n_end,
                                          w.n_rews, nullptr, w.n_dones, w.n_len, (void*)s_tc);
                if (rc != SAMX_OK) return rc;
            } else {   // done ahead by the caller (samx_nstep_lookahead_many)
                nl_end = orig->n_end; nl_rews = orig->n_rews; nl_dones = orig->n_dones; nl_len = orig->n_len;
            }
            rc = samx_replay_gather(nullptr, nullptr, nullptr, nullptr, orig->obs1, orig->ring_len, orig->ring_start,
                                    ob, ac, nl_end, B, nullptr, nullptr, nullptr, nullptr, w.n_obs1, (void*)s_tc);
            if (rc != SAMX_OK) return rc;
        }
        rc = samx_targ_q(&L->actor, ta, &L->critic, tc, b->obs1, b->rews, b->dones1, b->td_len, L->mu_mean,
                         L->mu_std, B, L->gamma, L->ac_high, o->targ_q, ret, w.nets_ws, w.nets_bytes,
                         (void*)s_tc);
        if (rc != SAMX_OK) return rc;
        if (L->n_step > 0) {
            rc = samx_targ_q(&L->actor, ta, &L->critic, tc, w.n_obs1, nl_rews, nl_dones, nl_len, L->mu_mean,
                             L->mu_std, B, L->gamma, L->ac_high, const_cast<float*>(tcns), ret, w.nets_ws,
                             w.nets_bytes, (void*)s_tc);
            if (rc != SAMX_OK) return rc;
        }
        if (popart_or_stats) {  // batch sums of ret_rms.update(targ_q_1 [, targ_q_n]) (sam_agent.py:662-665)
            rc = samx_rms_batch_sums(o->targ_q, B, nullptr, 0, 1, add1, (void*)s_tc);
            if (rc != SAMX_OK) return rc;
            if (tcns) {
                rc = samx_rms_batch_sums(tcns, B, nullptr, 0, 1, add2, (void*)s_tc);
                if (rc != SAMX_OK) return rc;
            }
        }
        }  // !t_done
        if (t_only) return SAMX_OK;   // (bits 4/5 run on the caller's stream: s_tc == st)
        if (popart_or_stats) {  // commit + PopArt (sam_agent.py:662-672)
            rc = samx_ret_update_popart(L->ret_sums, add1, tcns ? add2 : nullptr, L->ret_mean,
                                        L->ret_std, 1, &L->critic, L->params + Pa, L->targets + Pa, (void*)s_tc);
            if (rc != SAMX_OK) return rc;
            if (ss) {  // the actor role reads the rescaled critic head and the new statistics
                SAMX_CHECK_HIP(hipEventRecord(ss->mid, s_tc));
                SAMX_CHECK_HIP(hipStreamWaitEvent(s_a, ss->mid, 0));
            }
        }
        rc = samx_critic_grad(&L->critic, pc, b->obs0, b->acs, b->iws, o->targ_q, L->mu_mean, L->mu_std, B,
                              L->ob_dim, L->td_loss_scale, L->wd_scale, o->critic_losses, L->grads + Pa,
                              o->td, ret, tcns, L->td_loss_n_scale, nullptr, w.nets_ws, w.nets_bytes,
                              (void*)s_tc);
        if (rc != SAMX_OK) return rc;
        // ---- actor role (side 1)
        rc = samx_actor_grad(&L->actor, pa, &L->critic, pc, &L->reward, L->actor_reward_params, b->obs0,
                             L->mu_mean, L->mu_std, B, L->ac_high, L->q_actor_loss_scale,
                             L->d_actor_loss_scale, o->actor_losses, L->grads, ret, w.nets_ws2, w.nets_bytes,
                             (void*)s_a);
        if (rc != SAMX_OK) return rc;
        }  // do_a
        if (do_d) {
        rc = samx::disc_grad_src(&L->reward, pd, b->p_obs, b->p_acs, b->e_obs, b->e_acs, b->eps_ob, b->eps_ac,
                           b->real_labels, b->fake_labels, L->d_mean, L->d_std, B, L->ob_dim,
                           L->ent_reg_scale, o->d_losses, L->grads + Pa + Pc, w.disc_ws, w.disc_bytes, &dsrc,
                           stream);
        if (rc != SAMX_OK) return rc;
        }  // do_d
        if (ss) {  // join
            for (int i = 0; i < 2; ++i) {
                SAMX_CHECK_HIP(hipEventRecord(ss->join[i], ss->s[i]));
                SAMX_CHECK_HIP(hipStreamWaitEvent(st, ss->join[i], 0));
            }
        }
        if (!(part & 2)) return SAMX_OK;
        fused::ApplyArgs f = make_apply(L, w, 0, 1.f, segs);
        // (the counters advance in the same launch: nothing in it reads them)
        return fused::launch_seg_sumsq(f, st, (tick && do_a) ? L->steps : nullptr, (tick && do_a) ? L->steps + 1 : nullptr,
                                       (tick && do_d) ? L->steps + 2 : nullptr);
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
    d.p_idx = b->p_idx; d.e_start = b->e_start; d.ring_start = b->ring_start; d.ring_len = b->ring_len;
    d.rng_seed = b->rng_seed; d.rng_step = L->steps + 2; d.label_smoothing = b->label_smoothing;
    d.rms_sums = b->d_rms_sums; d.rms_addvec = b->d_rms_addvec;

    fused::AgentArgs a;
    a.obs0 = b->obs0; a.acs = b->acs; a.rews = b->rews; a.dones1 = b->dones1; a.obs1 = b->obs1;
    a.iws = b->iws; a.td_len = b->td_len; a.mean = L->mu_mean; a.stdv = L->mu_std;
    a.actor = pa; a.critic = pc; a.targ_actor = ta; a.targ_critic = tc; a.reward = L->actor_reward_params;
    a.ob = L->ob_dim; a.ac = L->ac_dim; a.B = B;
    a.high = L->ac_high; a.gamma = L->gamma; a.td_scale = L->td_loss_scale;
    a.q_scale = L->q_actor_loss_scale; a.d_scale = L->d_actor_loss_scale;
    a.targ_q_out = o->targ_q; a.td_out = o->td;
    a.partials = w.agent_part; a.P = Pa + Pc; a.Pa = Pa; a.loss_part = w.agent_loss;
    a.idx = b->a_idx; a.ring_start = b->ring_start; a.ring_len = b->ring_len;
    a.ret_mean = L->ret_mean; a.ret_std = L->ret_std;
    a.ext_tq = a.ext_tqn = nullptr; a.tdn_scale = L->td_loss_n_scale;
    a.n_idx = nullptr; a.n_rews = a.n_dones = a.n_len = nullptr; a.targ_qn_out = nullptr;
    const bool ln = L->actor.layernorm != 0;
    // targets from their own launch: always with LayerNorm (the critic role then holds one network, not three)
    const bool ext = L->ret_mean || L->n_step > 0 || ln;
    if ((part & 1) && do_a && ext) {
        // sam_agent.py:634-672: [n-step lookahead] -> targets with the OLD return statistics / heads ->
        // ret_rms.update + PopArt -> losses with the new ones
        float* tqn = L->n_step > 0 ? (o->targ_q_n ? o->targ_q_n : w.n_tq) : nullptr;
        if (L->n_step > 0 && !t_done) {
            SAMX_REQUIRE(b->a_idx && b->num_entries > 0,
                         "samx_step_grads: n-step returns need a_idx (ring rows) and num_entries");
            if (b->n_end) {   // done ahead by the caller (samx_nstep_lookahead_many)
                a.n_idx = b->n_end; a.n_rews = b->n_rews; a.n_dones = b->n_dones; a.n_len = b->n_len;
            } else {
                rc = samx_nstep_lookahead(b->rews, b->dones1, b->ring_len, b->ring_start, b->num_entries, b->a_idx,
                                          B, L->n_step, (double)L->gamma, w.n_end, w.n_rews, nullptr, w.n_dones,
                                          w.n_len, stream);
                if (rc != SAMX_OK) return rc;
                a.n_idx = w.n_end; a.n_rews = w.n_rews; a.n_dones = w.n_dones; a.n_len = w.n_len;
            }
            a.targ_qn_out = tqn;
        }
        if (!t_done) {
            rc = (ln ? fused::launch_targets_ln : fused::launch_targets)(&a, st);
            if (rc != SAMX_OK) return rc;
        }
        if (L->ret_mean && L->enable_popart) {  // sam_agent.py:662-672
            rc = fused::launch_ret_popart(o->targ_q, tqn, B, L->ret_sums, L->ret_mean, L->ret_std, 1,
                                          L->params + Pa, L->targets + Pa, &L->critic,
                                          t_only ? 1 : (t_done ? 2 : 0), b->ret_addvec, st);
            if (rc != SAMX_OK) return rc;
        }
        if (t_only) return SAMX_OK;
        a.ext_tq = o->targ_q; a.ext_tqn = tqn;
    } else if (ext) {
        a.ext_tq = o->targ_q; a.ext_tqn = L->n_step > 0 ? (o->targ_q_n ? o->targ_q_n : w.n_tq) : nullptr;
    }
    if (part & 1) {
        rc = (ln ? fused::launch_step_ln : fused::launch_step)(do_a ? &a : nullptr, do_d ? &d : nullptr, st);
        if (rc != SAMX_OK) return rc;
    }
    if (!(part & 2)) return SAMX_OK;

    fused::FinalizeArgs f;
    f.nseg = 0;
    f.B = B;
    const int64_t off[3] = {0, Pa, Pa + Pc}, n[3] = {Pa, Pc, Pd};
    for (int k = 0; k < 3; ++k) {
        if (!(segs & (1 << k))) continue;
        fused::Segment& s = f.seg[f.nseg++];
        s.partials = k < 2 ? w.agent_part + off[k] : w.disc_part;
        s.slot_stride = k < 2 ? Pa + Pc : Pd;
        s.nslots = k < 2 ? tiles : 3 * tiles;
        s.scaled_from = k < 2 ? tiles : 2 * tiles;
        s.n = n[k];
        s.grad = L->grads + off[k];
        s.nblocks = fused::seg_blocks(n[k]);
        s.sumsq = w.sumsq + k * fused::kMaxSegBlocks;
        s.wd = 0.f; s.param = nullptr; s.wd0_lo = s.wd0_hi = s.wd1_lo = s.wd1_hi = 0;
        s.step_dev = tick ? L->steps + k : nullptr;
    }
    if (do_a) {   // weight decay on the critic's two hidden kernels (networks.py:385-386,406-412)
        fused::Segment& s = f.seg[1];
        const int64_t inc = L->critic.in_dim, h = 64;
        s.wd = L->wd_scale; s.param = pc;
        s.wd0_lo = 0; s.wd0_hi = inc * h;
        s.wd1_lo = inc * h + h + (ln ? 2 * h : 0); s.wd1_hi = s.wd1_lo + h * h;   // W2 (behind b1 [beta1 gamma1])
    }
    f.agent_loss_part = do_a ? w.agent_loss : nullptr; f.agent_tiles = tiles;
    f.q_scale = L->q_actor_loss_scale; f.d_scale = L->d_actor_loss_scale;
    f.td_scale = L->td_loss_scale; f.wd_scale = L->wd_scale;
    f.ret_mean = L->ret_mean; f.ret_std = L->ret_std;
    f.tdn_scale = L->td_loss_n_scale; f.has_tdn = L->n_step > 0 ? 1 : 0;
    f.actor_losses = o->actor_losses; f.critic_losses = o->critic_losses;
    f.disc_loss_part = do_d ? w.disc_loss : nullptr; f.disc_tiles_b = tiles;
    f.ent_scale = L->ent_reg_scale; f.gp_coeff = 10.f; f.gp = w.gp_out; f.disc_losses = o->d_losses;
    f.gp_part = do_d ? w.gp_part : nullptr; f.gp_tiles = tiles; f.ob = L->ob_dim; f.ac = L->ac_dim;
    f.rms_sums = do_d ? b->d_rms_sums : nullptr; f.rms_addvec = do_d ? b->d_rms_addvec : nullptr;
    f.rms_mean = const_cast<float*>(L->d_mean); f.rms_std = const_cast<float*>(L->d_std);
    if (part & 512) {   // = samx_step_apply(mode 3 [| 8 | 16 for the role subset], grad_scale 1) in the same launch
        fused::ApplyArgs ap = make_apply(L, w, 3, 1.f, segs);
        return fused::launch_finalize_apply(f, ap, w.fa_slots,
                                            reinterpret_cast<unsigned*>(w.fa_slots + 3 * fused::kMaxSegBlocks), st);
    }
    return fused::launch_finalize(f, st);
}

int samx_step_apply(const samx_learner* L, int64_t B, int mode, float grad_scale, void* ws, size_t ws_bytes,
                    void* stream) {
    int rc = check_learner(L, "samx_step_apply");
    if (rc != SAMX_OK) return rc;
    SAMX_REQUIRE(ws && B > 0, "samx_step_apply: null workspace or bad batch");
    SAMX_REQUIRE((mode & ~127) == 0 && (mode & 3) != 0 && (mode & 24) != 24,
                 "samx_step_apply: mode uses bits 0-6, needs bit 0 or 1, and cannot skip every segment");
    const int segs = ((mode & 16) ? 0 : 3) | ((mode & 8) ? 0 : 4);
    SAMX_REQUIRE(!(mode & 1) || L->clipped, "samx_step_apply: mode bit 0 needs learner.clipped");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    StepWs w = carve(ar, L, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_step_apply: workspace %zu < %zu", ws_bytes, ar.off);
    float* img[5][2];
    float* row[3][2];
    const bool keep = (mode & 64) && (mode & 2) && use_wide(L, B);   // (bit 6 means nothing on the other paths)
    if (keep) {
        SAMX_REQUIRE(!(mode & 32), "samx_step_apply: mode bit 6 needs the Polyak step in the same pass (no bit 5)");
        rc = samx::wide_images(L, B, w.wide_red, w.wide_red_bytes, img, row);
        if (rc != SAMX_OK) return rc;
    }
    fused::ApplyArgs f = make_apply(L, w, mode & 3, grad_scale, segs, !(mode & 32), keep ? img : nullptr, keep ? row : nullptr);
    if (mode & 4) {  // gradients changed since samx_step_grads (all-reduce): recompute the norms
        rc = fused::launch_seg_sumsq(f, st);
        if (rc != SAMX_OK) return rc;
    }
    return fused::launch_apply(f, st);
}

}  // extern "C"
