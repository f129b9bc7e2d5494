// Whole-network fused kernels for the small configurations (64x64 MLPs, inputs <= 64 wide).
//
// One workgroup (4 waves) owns R = 16 batch rows and runs the ENTIRE pass for them — observation
// normalisation, every layer forward, losses, backward to the inputs, and the per-row-block weight
// gradients — with all weights and activations staged in LDS and every matmul on
// v_mfma_f32_16x16x4_f32 (exact f32).  Each workgroup writes its weight-gradient contribution to its
// own slot of a partials buffer; `finalize` adds the slots in a fixed order (deterministic, no float
// atomics), `apply` clips and runs Adam/Polyak for up to three optimisers in one launch.
//
// A 64-wide SAM step is launch/latency bound (54 MFLOP, <1 MB touched): the point of this file is
// to do it in ~6 launches instead of ~57.
//
// LDS layout rule: every matrix (activations [16][*], weights [K][64]) uses a row pitch == 17 mod 32
// floats.  With that pitch all three MFMA operand access patterns used here (row-major A, row-major
// B, transposed B) touch 31-32 distinct banks per 32-lane half.
//
// Reference semantics: sam_agent.py:145-205,406-520 ; discriminator.py:61-161 ;
// networks.py:45-72,222-404 ; mpi_adam.py:18-36 ; tf_util.py:137-151.
#include "fused_dev.h"
#include <stdlib.h>

namespace samx {
namespace fused {
namespace {

// ------------------------------------------------------------------------------------------------
// Agent roles, 16 rows per workgroup.  Two independent roles run side by side in the step kernel:
//   TC : TD target on the target nets (obs1), then critic loss + gradient (obs0, batch actions)
//   A  : actor loss + gradient through the critic (and the reward net when d_scale != 0)
// ------------------------------------------------------------------------------------------------
struct AgentLds {
    NetL nA, nC, nD;
    float *XC, *XC0, *A1a, *A2a, *A1c, *A2c, *DZa, *DZb, *PRE, *s_, *tq, *dq, *red, *rw, *dn, *iw, *tl, *son;
};

__device__ inline AgentLds carve_agent_lds(float* p, int ob, int ac, bool use_d) {
    AgentLds L;
    L.nA = carve_net(p, ob, ac);
    L.nC = carve_net(p, ob + ac, 1);
    L.nD = L.nC;
    if (use_d) L.nD = carve_net(p, ob + ac, 1);
    L.XC = p; p += R * LD;
    L.XC0 = p; p += R * LD;
    L.A1a = p; p += R * LD;
    L.A2a = p; p += R * LD;
    L.A1c = p; p += R * LD;
    L.A2c = p; p += R * LD;
    L.DZa = p; p += R * LD;
    L.DZb = p; p += R * LD;
    L.PRE = p; p += R * LD3;
    L.s_ = p; L.tq = p + 16; L.dq = p + 32; L.red = p + 48;
    L.rw = p + 64; L.dn = p + 80; L.iw = p + 96; L.tl = p + 112; L.son = p + 128;
    return L;
}

// actor head: 16 x ac tile, tanh * high, clipped (sam_agent.py:177), into the action columns of XC
__device__ inline void actor_head(const AgentLds& L, int ob, int ac, float high) {
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) == 0) {
        f32x4 acc = tile_nn(L.A2a, LD, L.nA.W3, LD3, H, 0, lane);
        const int c = lane & 15, r0 = (lane >> 4) * 4;
        if (c < ac) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = high * tanhf(acc[j] + L.nA.b3[c]);
                L.XC[(r0 + j) * LD + ob + c] = fminf(fmaxf(v, -high), high);
            }
        }
    }
}

__device__ inline void agent_tc_body(const AgentArgs& a, int wg, float* lds, int trole = 1) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    AgentLds L = carve_agent_lds(lds, ob, ac, true);   // third slot: the online critic, staged early
    float* gC = a.partials + (int64_t)wg * a.P + a.Pa;  // critic part of this tile's gradient slot
    const int64_t cW1 = 0, cb1 = cW1 + (int64_t)in_c * H, cW2 = cb1 + H, cb2 = cW2 + H * H, cW3 = cb2 + H,
                  cb3 = cW3 + H;
    // ---- TD target: rews + gamma^len (1 - done) Q'(obz1, mu'(obz1))   (sam_agent.py:196-205)
    TRACE(trole, 0);
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));   // this thread's row
    const int64_t prs = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid & 15));  // row of scalar tid < R
    NetRegs wA, wC, wD;
    net_issue(wA, L.nA, a.targ_actor);
    net_issue(wC, L.nC, a.targ_critic);
    net_issue(wD, L.nD, a.critic);
    const HeadRegs hA = head_issue(L.nA, a.targ_actor), hC = head_issue(L.nC, a.targ_critic),
                   hD = head_issue(L.nD, a.critic);
    float rw = 0.f, dn = 0.f, iw = 1.f, tl = 1.f;
    if (tid < R) {
        if (a.iws) iw = a.iws[row0 + tid];          // importance weights are per BATCH row
        if (a.td_len) tl = a.td_len[row0 + tid];
    }
    TRACE(trole, 7);
    RowRegs x1, x0;
    rows_issue(x1, pr, a.obs1, nullptr, a.mean, a.stdv, ob, 0, ob);
    rows_issue(x0, pr, a.obs0, a.acs, a.mean, a.stdv, ob, ac, in_c);
    if (tid < R) {
        rw = a.rews[prs];
        dn = a.dones1[prs];
    }
    TRACE(trole, 8);
    net_commit(wA, L.nA);
    net_commit(wC, L.nC);
    net_commit(wD, L.nD);
    head_commit(L.nA, hA);
    head_commit(L.nC, hC);
    head_commit(L.nD, hD);
    TRACE(trole, 9);
    rows_commit(L.XC, x1, a.mean != nullptr, ob, ob);
    rows_commit(L.XC0, x0, a.mean != nullptr, ob, in_c);
    if (tid < R) { L.rw[tid] = rw; L.dn[tid] = dn; L.iw[tid] = iw; L.tl[tid] = tl; }
    TRACE(trole, 1);
    stage_wait();
    TRACE(trole, 2);
    // The online critic's forward (obs0, batch actions) does not depend on the target: it runs side by
    // side with the target actor's forward, two 16x64 tiles per wave and barrier.
    layer_fwd(L.XC, L.nA.k4, L.nA.W1, L.nA.b1, L.A1a);
    layer_fwd(L.XC0, L.nD.k4, L.nD.W1, L.nD.b1, L.A1c);
    __syncthreads();
    layer_fwd(L.A1a, H, L.nA.W2, L.nA.b2, L.A2a);
    layer_fwd(L.A1c, H, L.nD.W2, L.nD.b2, L.A2c);
    __syncthreads();
    actor_head(L, ob, ac, a.high);
    head_dot(L.A2c, L.nD, L.son);            // Q(obz0, acs), online critic
    __syncthreads();
    TRACE(trole, 3);
    // target critic on [obz1 | mu'(obz1)]; its activations reuse the target actor's tiles
    layer_fwd(L.XC, L.nC.k4, L.nC.W1, L.nC.b1, L.A1a);
    if (wg == 0) {
        // sum of squares of the decayed critic kernels for the weight-decay loss term
        // (network_util.py:43-56); padded LDS rows are zero
        double acc = 0.0;
        for (int t = tid; t < L.nD.k4 * H; t += NT) { const float w = L.nD.W1[(t >> 6) * LD + (t & 63)]; acc += (double)w * w; }
        for (int t = tid; t < H * H; t += NT) { const float w = L.nD.W2[(t >> 6) * LD + (t & 63)]; acc += (double)w * w; }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) L.red[wv] = (float)acc;
    }
    __syncthreads();
    if (wg == 0 && tid == 0) a.loss_part[3] = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    layer_fwd(L.A1a, H, L.nC.W2, L.nC.b2, L.A2a);
    __syncthreads();
    head_dot(L.A2a, L.nC, L.s_);
    __syncthreads();
    TRACE(trole, 4);
    // ---- TD target, critic loss + gradient (sam_agent.py:196-205,457-520)
    L.nC = L.nD;    // from here on: the online critic ...
    L.XC = L.XC0;   // ... and the [obz0 | acs] rows
    if (tid < R) {
        const int64_t g = row0 + tid;
        const float gam = a.td_len ? powf(a.gamma, L.tl[tid]) : a.gamma;
        const float t = L.rw[tid] + gam * ((1.f - L.dn[tid]) * L.s_[tid]);
        a.targ_q_out[g] = t;
        const float td = L.son[tid] - t;
        const float iw = L.iw[tid];
        const float ad = fabsf(td);
        const float hub = ad < 1.f ? 0.5f * td * td : ad - 0.5f;         // math_util.py:22-29
        const float dh = ad < 1.f ? td : (td > 0.f ? 1.f : -1.f);
        L.dq[tid] = (a.td_scale / (float)a.B) * dh * iw;
        L.red[tid] = hub * iw;
        if (a.td_out) a.td_out[g] = td;
    }
    __syncthreads();
    TRACE(trole, 5);
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += L.red[r];
        a.loss_part[wg * kLossStride + 0] = s;  // sum of huber*iw over this tile's rows
    }
    head_bwd(L.dq, L.nC, L.A2c, L.DZa);
    wgrad_head1(L.A2c, L.dq, gC + cW3, gC + cb3, R);
    __syncthreads();
    wgrad64(L.A1c, H, L.DZa, gC + cW2, gC + cb2, R);
    layer_bwd(L.DZa, L.nC.W2, L.A1c, L.DZb);
    __syncthreads();
    wgrad64(L.XC, in_c, L.DZb, gC + cW1, gC + cb1, R);
    TRACE(trole, 6);
}

// TD targets only (return normalisation + PopArt / n-step: the statistics and the critic heads change
// between the target pass and the loss pass, sam_agent.py:651-672).  pass 0: the 1-step target of the
// sampled rows; pass 1: the n-step target (rows n_idx, dense n_rews / n_dones / n_len).
__device__ inline void agent_t_body(const AgentArgs& a, int wg, float* lds, int pass) {
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac;
    AgentLds L = carve_agent_lds(lds, ob, ac, false);
    const int64_t* idx = pass ? a.n_idx : a.idx;
    const int64_t pr = ring_row(idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    const int64_t prs = ring_row(idx, a.ring_start, a.ring_len, row0 + (tid & 15));
    NetRegs wA, wC;
    net_issue(wA, L.nA, a.targ_actor);
    net_issue(wC, L.nC, a.targ_critic);
    const HeadRegs hA = head_issue(L.nA, a.targ_actor), hC = head_issue(L.nC, a.targ_critic);
    float rw = 0.f, dn = 0.f, tl = 1.f;
    if (tid < R) {
        if (pass) { rw = a.n_rews[row0 + tid]; dn = a.n_dones[row0 + tid]; tl = a.n_len[row0 + tid]; }
        else if (a.td_len) tl = a.td_len[row0 + tid];
    }
    RowRegs x1;
    rows_issue(x1, pr, a.obs1, nullptr, a.mean, a.stdv, ob, 0, ob);
    if (tid < R && !pass) { rw = a.rews[prs]; dn = a.dones1[prs]; }
    net_commit(wA, L.nA);
    net_commit(wC, L.nC);
    head_commit(L.nA, hA);
    head_commit(L.nC, hC);
    rows_commit(L.XC, x1, a.mean != nullptr, ob, ob);
    if (tid < R) { L.rw[tid] = rw; L.dn[tid] = dn; L.tl[tid] = tl; }
    stage_wait();
    layer_fwd(L.XC, L.nA.k4, L.nA.W1, L.nA.b1, L.A1a);
    __syncthreads();
    layer_fwd(L.A1a, H, L.nA.W2, L.nA.b2, L.A2a);
    __syncthreads();
    actor_head(L, ob, ac, a.high);
    __syncthreads();
    layer_fwd(L.XC, L.nC.k4, L.nC.W1, L.nC.b1, L.A1c);
    __syncthreads();
    layer_fwd(L.A1c, H, L.nC.W2, L.nC.b2, L.A2c);
    __syncthreads();
    head_dot(L.A2c, L.nC, L.s_);
    __syncthreads();
    if (tid < R) {
        const bool lens = pass || a.td_len;
        const float gam = lens ? powf(a.gamma, L.tl[tid]) : a.gamma;
        float qp = L.s_[tid];
        if (a.ret_mean) qp = qp * a.ret_std[0] + a.ret_mean[0];              // sam_agent.py:199-201
        const float t = L.rw[tid] + gam * ((1.f - L.dn[tid]) * qp);
        (pass ? a.targ_qn_out : a.targ_q_out)[row0 + tid] = t;
    }
}

// Critic loss + gradient against EXTERNAL targets (a.ext_tq, a.ext_tqn), with return normalisation:
// pred = s * std + mean, td = pred - (target - mean) / std   (sam_agent.py:161-182,457-520)
__device__ inline void agent_c_body(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    AgentLds L = carve_agent_lds(lds, ob, ac, false);
    float* gC = a.partials + (int64_t)wg * a.P + a.Pa;
    const int64_t cW1 = 0, cb1 = cW1 + (int64_t)in_c * H, cW2 = cb1 + H, cb2 = cW2 + H * H, cW3 = cb2 + H,
                  cb3 = cW3 + H;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wC;
    net_issue(wC, L.nC, a.critic);
    const HeadRegs hC = head_issue(L.nC, a.critic);
    float iw = 1.f, t1 = 0.f, tn = 0.f;
    if (tid < R) {
        if (a.iws) iw = a.iws[row0 + tid];
        t1 = a.ext_tq[row0 + tid];
        if (a.ext_tqn) tn = a.ext_tqn[row0 + tid];
    }
    const float rmean = a.ret_mean ? a.ret_mean[0] : 0.f, rstd = a.ret_std ? a.ret_std[0] : 1.f;
    RowRegs x0;
    rows_issue(x0, pr, a.obs0, a.acs, a.mean, a.stdv, ob, ac, in_c);
    net_commit(wC, L.nC);
    head_commit(L.nC, hC);
    rows_commit(L.XC, x0, a.mean != nullptr, ob, in_c);
    if (tid < R) { L.iw[tid] = iw; L.tq[tid] = t1; L.tl[tid] = tn; }
    stage_wait();
    layer_fwd(L.XC, L.nC.k4, L.nC.W1, L.nC.b1, L.A1c);
    if (wg == 0) {   // weight-decay loss term (network_util.py:43-56)
        double acc = 0.0;
        for (int t = tid; t < L.nC.k4 * H; t += NT) { const float w = L.nC.W1[(t >> 6) * LD + (t & 63)]; acc += (double)w * w; }
        for (int t = tid; t < H * H; t += NT) { const float w = L.nC.W2[(t >> 6) * LD + (t & 63)]; acc += (double)w * w; }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) L.red[wv] = (float)acc;
    }
    __syncthreads();
    if (wg == 0 && tid == 0) a.loss_part[3] = L.red[0] + L.red[1] + L.red[2] + L.red[3];
    layer_fwd(L.A1c, H, L.nC.W2, L.nC.b2, L.A2c);
    __syncthreads();
    head_dot(L.A2c, L.nC, L.s_);
    __syncthreads();
    if (tid < R) {
        const int64_t g = row0 + tid;
        const float pred = a.ret_mean ? L.s_[tid] * rstd + rmean : L.s_[tid];
        const float w = L.iw[tid];
        auto hub = [](float td, float& dh) {                              // math_util.py:22-29
            const float ad = fabsf(td);
            dh = ad < 1.f ? td : (td > 0.f ? 1.f : -1.f);
            return ad < 1.f ? 0.5f * td * td : ad - 0.5f;
        };
        const float td1 = pred - (a.ret_mean ? (L.tq[tid] - rmean) / rstd : L.tq[tid]);
        float dh1, dhn = 0.f, hn = 0.f;
        const float h1 = hub(td1, dh1);
        float dq = (a.td_scale / (float)a.B) * dh1;
        if (a.ext_tqn) {                                                  // sam_agent.py:483-498
            const float tdn = pred - (a.ret_mean ? (L.tl[tid] - rmean) / rstd : L.tl[tid]);
            hn = hub(tdn, dhn);
            dq += (a.tdn_scale / (float)a.B) * dhn;
        }
        L.dq[tid] = dq * w * rstd;
        L.red[tid] = h1 * w;
        L.son[tid] = hn * w;
        if (a.td_out) a.td_out[g] = td1;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, sn = 0.f;
        for (int r = 0; r < R; ++r) { s += L.red[r]; sn += L.son[r]; }
        a.loss_part[wg * kLossStride + 0] = s;
        a.loss_part[wg * kLossStride + 4] = sn;
    }
    head_bwd(L.dq, L.nC, L.A2c, L.DZa);
    wgrad_head1(L.A2c, L.dq, gC + cW3, gC + cb3, R);
    __syncthreads();
    wgrad64(L.A1c, H, L.DZa, gC + cW2, gC + cb2, R);
    layer_bwd(L.DZa, L.nC.W2, L.A1c, L.DZb);
    __syncthreads();
    wgrad64(L.XC, in_c, L.DZb, gC + cW1, gC + cb1, R);
}

__device__ inline void agent_a_body(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac;
    const bool use_d = a.d_scale != 0.f;
    TRACE(0, 0);
    AgentLds L = carve_agent_lds(lds, ob, ac, use_d);
    float* gA = a.partials + (int64_t)wg * a.P;  // actor part of this tile's gradient slot
    const int64_t aW1 = 0, ab1 = aW1 + (int64_t)ob * H, aW2 = ab1 + H, ab2 = aW2 + H * H, aW3 = ab2 + H,
                  ab3 = aW3 + (int64_t)H * ac;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wA, wC, wD;
    net_issue(wA, L.nA, a.actor);
    net_issue(wC, L.nC, a.critic);
    if (use_d) net_issue(wD, L.nD, a.reward);
    const HeadRegs hA = head_issue(L.nA, a.actor), hC = head_issue(L.nC, a.critic);
    HeadRegs hD = hC;
    if (use_d) hD = head_issue(L.nD, a.reward);
    RowRegs x0;
    rows_issue(x0, pr, a.obs0, nullptr, a.mean, a.stdv, ob, 0, ob);
    net_commit(wA, L.nA);
    net_commit(wC, L.nC);
    if (use_d) net_commit(wD, L.nD);
    head_commit(L.nA, hA);
    head_commit(L.nC, hC);
    if (use_d) head_commit(L.nD, hD);
    rows_commit(L.XC, x0, a.mean != nullptr, ob, ob);
    stage_wait();
    TRACE(0, 1);
    layer_fwd(L.XC, L.nA.k4, L.nA.W1, L.nA.b1, L.A1a);
    __syncthreads();
    layer_fwd(L.A1a, H, L.nA.W2, L.nA.b2, L.A2a);
    __syncthreads();
    actor_head(L, ob, ac, a.high);
    __syncthreads();
    TRACE(0, 2);
    // Q branch: -q_scale * mean Q(obz0, mu(obz0))   (sam_agent.py:412-413)
    layer_fwd(L.XC, L.nC.k4, L.nC.W1, L.nC.b1, L.A1c);
    __syncthreads();
    layer_fwd(L.A1c, H, L.nC.W2, L.nC.b2, L.A2c);
    __syncthreads();
    head_dot(L.A2c, L.nC, L.s_);
    const float rstd_ = a.ret_std ? a.ret_std[0] : 1.f;   // d(de-normalised Q)/d(critic output)
    if (tid < R) L.dq[tid] = -a.q_scale / (float)a.B * rstd_;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += L.s_[r];
        a.loss_part[wg * kLossStride + 1] = s;  // sum of Q(s, mu(s))
    }
    head_bwd(L.dq, L.nC, L.A2c, L.DZa);
    __syncthreads();
    layer_bwd(L.DZa, L.nC.W2, L.A1c, L.DZb);
    __syncthreads();
    // dQ/da = dZ1c * W1c[ob:ob+ac, :]^T   (wave 0, one tile)
    f32x4 dA = {0.f, 0.f, 0.f, 0.f};
    if (wv == 0) dA = tile_nt64(L.DZb, L.nC.W1, ob, lane);
    __syncthreads();
    TRACE(0, 3);
    if (use_d) {  // D branch: -d_scale * mean D(obz0, mu(obz0))   (sam_agent.py:421-422)
        layer_fwd(L.XC, L.nD.k4, L.nD.W1, L.nD.b1, L.A1c);
        __syncthreads();
        layer_fwd(L.A1c, H, L.nD.W2, L.nD.b2, L.A2c);
        __syncthreads();
        head_dot(L.A2c, L.nD, L.s_);
        if (tid < R) L.dq[tid] = -a.d_scale / (float)a.B * rstd_;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += L.s_[r];
            a.loss_part[wg * kLossStride + 2] = s;
        }
        head_bwd(L.dq, L.nD, L.A2c, L.DZa);
        __syncthreads();
        layer_bwd(L.DZa, L.nD.W2, L.A1c, L.DZb);
        __syncthreads();
        if (wv == 0) dA += tile_nt64(L.DZb, L.nD.W1, ob, lane);
        __syncthreads();
    } else if (tid == 0) {
        a.loss_part[wg * kLossStride + 2] = 0.f;
    }
    TRACE(0, 4);
    if (wv == 0) {  // through clip (identity inside the range) and tanh
        const int c = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = 0.f;
            if (c < ac) {
                const float t = L.XC[(r0 + j) * LD + ob + c] / a.high;
                v = dA[j] * (a.high * (1.f - t * t));
            }
            L.PRE[(r0 + j) * LD3 + c] = v;
        }
    }
    __syncthreads();
    {   // actor head gradient: dW3a[i][c] = sum_r A2a[r][i] * PRE[r][c]
        f32x4 acc = tile_tn(L.A2a, LD, 16 * wv, L.PRE, LD3, 0, lane);
        const int c = lane & 15, r0 = 16 * wv + (lane >> 4) * 4;
        if (c < ac) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gA[aW3 + (int64_t)(r0 + j) * ac + c] = acc[j];
        }
        if (tid < ac) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += L.PRE[r * LD3 + tid];
            gA[ab3 + tid] = s;
        }
    }
    {   // dZ2a = (PRE * W3a^T) .* lrelu'(A2a) : reduction over the (zero-padded) action index
        f32x4 acc = tile_nt(L.PRE, LD3, L.nA.W3, LD3, 16, 16 * wv, lane);
        const int c = 16 * wv + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) L.DZa[(r0 + j) * LD + c] = acc[j] * dlrelu(L.A2a[(r0 + j) * LD + c]);
    }
    __syncthreads();
    wgrad64(L.A1a, H, L.DZa, gA + aW2, gA + ab2, R);
    layer_bwd(L.DZa, L.nA.W2, L.A1a, L.DZb);
    __syncthreads();
    wgrad64(L.XC, ob, L.DZb, gA + aW1, gA + ab1, R);
    TRACE(0, 5);
}

// ------------------------------------------------------------------------------------------------
// Discriminator, first kernel.  Row tiles [0, 2B/16): policy / expert pairs — forward, BCE(+entropy)
// terms, full backward, weight-gradient slot.  Tiles [2B/16, 3B/16): interpolates — forward, the
// input gradient g = d score / d input, and this tile's share of sum g_ob^2, sum g_ac^2.
// ------------------------------------------------------------------------------------------------
// interpolate-row chain shared by both D kernels: forward, then g = ((w3.*m2) W2^T .* m1) W1^T
// leaves A1/A2 (activations), DZa (= gz2), DZb (= gz1), G[16][LD] (input gradient)
__device__ inline void disc_interp_chain(const NetL& n, float* XC, float* A1, float* A2, float* DZa,
                                         float* DZb, float* G, float* ones) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    layer_fwd(XC, n.k4, n.W1, n.b1, A1);
    __syncthreads();
    layer_fwd(A1, H, n.W2, n.b2, A2);
    if (threadIdx.x < R) ones[threadIdx.x] = 1.f;
    __syncthreads();
    head_bwd(ones, n, A2, DZa);
    __syncthreads();
    layer_bwd(DZa, n.W2, A1, DZb);
    __syncthreads();
    const int ti = (n.in + 15) / 16;
    for (int t = wv; t < ti; t += 4) {
        f32x4 acc = tile_nt(DZb, LD, n.W1, LD, H, 16 * t, lane);
        const int c = 16 * t + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) G[(r0 + j) * LD + c] = c < n.in ? acc[j] : 0.f;
    }
    __syncthreads();
}

__device__ inline void disc1_body(const DiscArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x;
    const int tiles_b = (int)(a.B / R);
    const int sect = wg / tiles_b;  // 0 policy, 1 expert, 2 interpolate
    const int64_t row0 = (int64_t)(wg % tiles_b) * R;
    const int in = a.ob + a.ac;
    float* p = lds;
    NetL n = carve_net(p, in, 1);
    float* XC = p; p += R * LD;
    float* A1 = p; p += R * LD;
    float* A2 = p; p += R * LD;
    float* DZa = p; p += R * LD;
    float* DZb = p; p += R * LD;
    float* sh = p; p += 96;
    float* ms = p; p += 128;  // mean[64] | std[64] when the statistics are updated inline
    float* s_ = sh; float* dq = sh + 16; float* r0_ = sh + 32; float* r1_ = sh + 48; float* r2_ = sh + 64;
    // traced tiles: the first policy tile (role 2) and the first interpolate tile (role 3)
    [[maybe_unused]] const int trole = sect == 2 ? 3 : 2, twg = sect == 2 ? wg - 2 * tiles_b : wg;
    TRACEW(twg, trole, 0);
    // ---- prologue: all global loads issued before the first wait (see "staged prologue")
    const int64_t pg = ring_row(a.p_idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    const int64_t e0 = a.e_start ? a.e_start[0] : 0;
    const uint64_t step = a.eps_ob ? 0 : (uint64_t)a.rng_step[0];
    NetRegs wn;
    net_issue(wn, n, a.params);
    const HeadRegs hd = head_issue(n, a.params);
    const bool norm = a.rms_sums != nullptr || a.mean != nullptr;
    // statistics the rows are normalised with.  d.obs_rms.update folded into the step
    // (helpers/mpi_running_mean_std.py:31-45): the batch's float64 column sums were computed ahead
    // (samx_d_input_sums_many); every workgroup derives the mean/std that INCLUDE this batch from
    // sums + addvec (same arithmetic as the commit in finalize, hence the same bits).
    double c0 = 0.0, c1 = 1.0, s1a = 0.0, s1b = 0.0, s2a = 0.0, s2b = 0.0;
    float gm = 0.f, gs = 1.f;
    if (tid < a.ob) {
        if (a.rms_sums) {
            c0 = a.rms_sums[2 * a.ob]; c1 = a.rms_addvec[2 * a.ob];
            s1a = a.rms_sums[tid]; s1b = a.rms_addvec[tid];
            s2a = a.rms_sums[a.ob + tid]; s2b = a.rms_addvec[a.ob + tid];
        } else if (a.mean) {
            gm = a.mean[tid]; gs = a.stdv[tid];
        }
    }
    DiscRowRegs xr;
    disc_rows_issue(xr, a, row0, sect, pg, e0);
    net_commit(wn, n);
    if (tid < a.ob) {
        if (a.rms_sums) {
            const double cnt = c0 + c1;
            const double s1 = __dadd_rn(s1a, s1b), s2 = __dadd_rn(s2a, s2b);
            const float mean = (float)(s1 / cnt), msq = (float)(s2 / cnt);
            gm = mean;
            gs = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(mean, mean)), 1e-2f));
        }
        ms[tid] = gm;
        ms[64 + tid] = gs;
    }
    head_commit(n, hd);
    __syncthreads();   // ms is read by other threads below
    disc_rows_commit(XC, xr, a, row0, sect, norm, ms, step);
    stage_wait();
    TRACEW(twg, trole, 1);
    if (sect == 2) {
        // Interpolate rows.  d(10*GP)/d(theta) = k * sum_rows <V, dg/dtheta> with V = g/(B*dim) and the
        // scalar k = 10*2(N-1)/N known only once every tile has contributed to N.  The slot gets the
        // sum WITHOUT k; finalize multiplies these slots by k (discriminator.py:136-140,154).
        float* G = p;  // sixth tile: the input gradient g, then V
        disc_interp_chain(n, XC, A1, A2, DZa, DZb, G, dq);
        if (tid < R) {  // this tile's sums of squares, observation / action columns (fixed order)
            float so = 0.f, sa = 0.f;
            for (int c = 0; c < a.ob; ++c) so = fmaf(G[tid * LD + c], G[tid * LD + c], so);
            for (int c = a.ob; c < in; ++c) sa = fmaf(G[tid * LD + c], G[tid * LD + c], sa);
            r0_[tid] = so; r1_[tid] = sa;
        }
        __syncthreads();
        if (tid == 0) {
            double o = 0.0, c = 0.0;
            for (int r = 0; r < R; ++r) { o += (double)r0_[r]; c += (double)r1_[r]; }
            a.gp_part[2 * (wg - 2 * tiles_b)] = o;
            a.gp_part[2 * (wg - 2 * tiles_b) + 1] = c;
        }
        const float co = 1.f / ((float)a.B * (float)a.ob), ca = 1.f / ((float)a.B * (float)a.ac);
        for (int t = tid; t < R * LD; t += NT) {
            const int c = t % LD;
            G[t] = c < in ? G[t] * (c < a.ob ? co : ca) : 0.f;
        }
        __syncthreads();
        float* part = a.partials + (int64_t)wg * a.P;
        const int64_t oW1 = 0, ob1 = (int64_t)in * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H,
                      ob3 = oW3 + H;
        wgrad64(G, in, DZb, part + oW1, nullptr, 0);      // V^T gz1
        layer_tan(G, n.k4, n.W1, A1, A1);                 // a1dot = (V W1) .* m1   (in place)
        if (tid < H) { part[ob1 + tid] = 0.f; part[ob2 + tid] = 0.f; }
        if (tid == 0) part[ob3] = 0.f;
        __syncthreads();
        wgrad64(A1, H, DZa, part + oW2, nullptr, 0);      // a1dot^T gz2
        layer_tan(A1, H, n.W2, A2, A2);                   // a2dot = (a1dot W2) .* m2
        __syncthreads();
        if (tid < H) {
            float s2 = 0.f;
            for (int r = 0; r < R; ++r) s2 += A2[r * LD + tid];
            part[oW3 + tid] = s2;                         // sum_r a2dot
        }
        TRACEW(twg, trole, 2);
        return;
    }
    // ---- policy / expert rows
    float* part = a.partials + (int64_t)wg * a.P;
    const int64_t oW1 = 0, ob1 = (int64_t)in * H, oW2 = ob1 + H, ob2 = oW2 + H * H, oW3 = ob2 + H, ob3 = oW3 + H;
    layer_fwd(XC, n.k4, n.W1, n.b1, A1);
    __syncthreads();
    layer_fwd(A1, H, n.W2, n.b2, A2);
    __syncthreads();
    head_dot(A2, n, s_);
    __syncthreads();
    if (tid < R) {
        const int64_t g = row0 + tid;
        const float s = s_[tid];
        float z;
        if (a.eps_ob) {  // injected tensors
            z = sect == 0 ? (a.fake ? a.fake[g] : 0.f) : (a.real ? a.real[g] : 1.f);
        } else {         // drawn here (discriminator.py:89-109)
            const uint64_t step = (uint64_t)a.rng_step[0];
            if (sect == 0)
                z = a.label_smoothing == 2 ? fake_label(philox_uniform(a.rng_seed, step, RNG_FAKE, (uint64_t)g))
                                           : 0.f;
            else
                z = a.label_smoothing >= 1 ? real_label(philox_uniform(a.rng_seed, step, RNG_REAL, (uint64_t)g))
                                           : 1.f;
        }
        const float sg = sigmoidf_(s);
        dq[tid] = (sg - z) / (float)a.B + a.ent_scale * s * sg * (1.f - sg) / (float)(2 * a.B);
        r0_[tid] = sig_ce(s, z);
        r1_[tid] = sig_ce(s, sg);
        r2_[tid] = sg;
    }
    __syncthreads();
    if (tid == 0) {
        float c = 0.f, e = 0.f, g = 0.f;
        for (int r = 0; r < R; ++r) { c += r0_[r]; e += r1_[r]; g += r2_[r]; }
        a.loss_part[wg * 4 + 0] = c;
        a.loss_part[wg * 4 + 1] = e;
        a.loss_part[wg * 4 + 2] = g;
    }
    head_bwd(dq, n, A2, DZa);
    wgrad_head1(A2, dq, part + oW3, part + ob3, R);
    __syncthreads();
    wgrad64(A1, H, DZa, part + oW2, part + ob2, R);
    layer_bwd(DZa, n.W2, A1, DZb);
    __syncthreads();
    wgrad64(XC, in, DZb, part + oW1, part + ob1, R);
    TRACEW(twg, trole, 2);
}

// critic_ops['td_errors'] (sam_agent.py:534-538,693-701) for 16 rows per workgroup: the PER refresh with
// the post-update critic, rows read in place through the index words.
struct TdArgs {
    const float *ret_mean, *ret_std;   // return normalisation (device scalars) or nullptr
    const float *obs0, *acs, *tc1s, *mean, *stdv, *critic;
    float* td_out;
    float* td_copy;       // optional second copy (pinned host memory) of td_out
    int64_t* idx_copy;    // optional copy of idx
    int ob, ac;
    const int64_t* idx;
    int64_t ring_start, ring_len;
};
__global__ __launch_bounds__(NT) void td_kernel(TdArgs a) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    float* p = lds;
    NetL n = carve_net(p, a.ob + a.ac, 1);
    float* XC = p; p += R * LD;
    float* A1 = p; p += R * LD;
    float* A2 = p; p += R * LD;
    float* s_ = p;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wn;
    net_issue(wn, n, a.critic);
    const HeadRegs h = head_issue(n, a.critic);
    RowRegs x;
    rows_issue(x, pr, a.obs0, a.acs, a.mean, a.stdv, a.ob, a.ac, a.ob + a.ac);
    net_commit(wn, n);
    head_commit(n, h);
    rows_commit(XC, x, a.mean != nullptr, a.ob, a.ob + a.ac);
    stage_wait();
    layer_fwd(XC, n.k4, n.W1, n.b1, A1);
    __syncthreads();
    layer_fwd(A1, H, n.W2, n.b2, A2);
    __syncthreads();
    head_dot(A2, n, s_);
    __syncthreads();
    if (tid < R) {
        float q = s_[tid], t = a.tc1s[row0 + tid];
        if (a.ret_mean) {   // sam_agent.py:161-182: de-normalised prediction minus normalised target
            const float rm = a.ret_mean[0], rs = a.ret_std[0];
            q = q * rs + rm;
            t = (t - rm) / rs;
        }
        a.td_out[row0 + tid] = q - t;
        if (a.td_copy) a.td_copy[row0 + tid] = q - t;
        if (a.idx_copy) a.idx_copy[row0 + tid] = a.idx[row0 + tid];
    }
}

// One launch for every gradient pass of the step that needs no grid-wide value: the two agent roles
// and the discriminator's first kernel are independent, so their workgroups share the chip.
// Longest role first.
__global__ __launch_bounds__(NT) void step_kernel(StepArgs s) {
    extern __shared__ float lds[];
    int b = blockIdx.x;
    if (b < s.n_a) { agent_a_body(s.a, b, lds); return; }
    b -= s.n_a;
#ifdef SAMX_TRACE_REPEAT
    if (b < s.n_tc) {   // second pass over the same code: warm instruction cache (results unchanged)
        for (int rep = 0; rep < 2; ++rep) { agent_tc_body(s.a, b, lds, rep ? 3 : 1); __syncthreads(); }
        return;
    }
#endif
    if (b < s.n_tc) {
        if (s.a.ext_tq) agent_c_body(s.a, b, lds); else agent_tc_body(s.a, b, lds);
        return;
    }
    b -= s.n_tc;
    disc1_body(s.d, b, lds);
}

// ret_rms.update(targ_q_1) [, update(targ_q_n)] and the PopArt rescale of the online and target critic
// heads in ONE launch (sam_agent.py:554-590,662-672; three launches on the layer-wise path).  One
// workgroup: float64 sums of x and x^2 per batch (thread-strided, then a fixed LDS tree), the running
// sums updated batch by batch, new mean/std, w3 <- w3 * old_std / new_std,
// b3 <- (b3 * old_std + old_mean - new_mean) / new_std.
__global__ __launch_bounds__(256) void ret_popart_kernel(const float* __restrict__ tq, const float* __restrict__ tqn,
                                                         int64_t B, double* __restrict__ sums, float* mean,
                                                         float* stdv, int popart, float* w3a, float* b3a,
                                                         float* w3b, float* b3b, int phase,
                                                         double* __restrict__ addvec, int h2) {
    ret_popart_body(tq, tqn, B, sums, mean, stdv, popart, w3a, b3a, w3b, b3b, phase, addvec, h2);   // ret_tail.h
}

__global__ __launch_bounds__(NT) void targets_kernel(AgentArgs a, int tiles) {
    extern __shared__ float lds[];
    const int b = blockIdx.x;
    agent_t_body(a, b % tiles, lds, b / tiles);
}

// ------------------------------------------------------------------------------------------------
// finalize: grad[i] = sum over slots (fixed order) [+ wd * W for decayed kernels], per-block sums of
// squares for the clip norm, and the scalar losses.  One launch covers up to three optimiser segments.
// ------------------------------------------------------------------------------------------------
// gradient-penalty scalars from the tiles' sums of squares: computed redundantly by every wave (one
// batch of loads + a fixed shuffle tree, identical everywhere) so that no barrier sits between the
// slot loads of the caller and their use
__device__ __forceinline__ void gp_scalars(const FinalizeArgs& f, int lane, float& k, double& gp_loss) {
    k = 0.f; gp_loss = 0.0;
    if (!f.gp_part) return;
    double so = 0.0, sa = 0.0;
    for (int i = lane; i < f.gp_tiles; i += 64) { so += f.gp_part[2 * i]; sa += f.gp_part[2 * i + 1]; }
    for (int o = 32; o > 0; o >>= 1) { so += __shfl_xor(so, o, 64); sa += __shfl_xor(sa, o, 64); }
    const double N = sqrt(so / ((double)f.B * f.ob) + sa / ((double)f.B * f.ac));
    k = (float)((double)f.gp_coeff * 2.0 * (N - 1.0) / N);
    gp_loss = (N - 1.0) * (N - 1.0);
}

// scalar losses, the commit of the inline d.obs_rms update and the step counters: one wave of the
// EXTRA last block, so this serial chain runs beside the gradient blocks instead of after one of them
__device__ inline void finalize_scalars(const FinalizeArgs& f, int tid, bool tick = true) {
    float kgp; double gp_loss;
    gp_scalars(f, tid, kgp, gp_loss);
    auto wsum = [](double x) {
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        return x;
    };
    if (f.agent_loss_part) {
        double h = 0.0, q = 0.0, dd = 0.0, hn = 0.0;
        for (int k = tid; k < f.agent_tiles; k += 64) {
            h += (double)f.agent_loss_part[kLossStride * k];
            q += (double)f.agent_loss_part[kLossStride * k + 1];
            dd += (double)f.agent_loss_part[kLossStride * k + 2];
            if (f.has_tdn) hn += (double)f.agent_loss_part[kLossStride * k + 4];
        }
        h = wsum(h); q = wsum(q); dd = wsum(dd); hn = wsum(hn);
        if (tid == 0) {
            const double B = (double)f.B;
            // with return normalisation the actor loss is on the DE-normalised prediction
            // (sam_agent.py:184-194): mean(s * std + mean) = mean(s) * std + mean
            const double rm = f.ret_mean ? (double)f.ret_mean[0] : 0.0, rs = f.ret_std ? (double)f.ret_std[0] : 1.0;
            const double ql = -(q / B * rs + rm) * f.q_scale;
            const double dl = f.d_scale != 0.f ? -(dd / B * rs + rm) * f.d_scale : 0.0;
            f.actor_losses[0] = (float)ql; f.actor_losses[1] = (float)dl;
            f.actor_losses[2] = (float)(ql + dl);
            const double tl = h / B * f.td_scale;
            const double tn = f.has_tdn ? hn / B * f.tdn_scale : 0.0;
            const double wl = 0.5 * (double)f.wd_scale * (double)f.agent_loss_part[3];
            f.critic_losses[0] = (float)tl; f.critic_losses[1] = (float)wl;
            f.critic_losses[2] = (float)(tl + wl + tn);
            f.critic_losses[3] = (float)tn;
        }
    }
    if (f.disc_loss_part) {
        double pc = 0, ec = 0, pe = 0, ee = 0, pg = 0, eg = 0;
        const int tb = f.disc_tiles_b;
        for (int k = tid; k < tb; k += 64) {
            pc += (double)f.disc_loss_part[4 * k]; pe += (double)f.disc_loss_part[4 * k + 1];
            pg += (double)f.disc_loss_part[4 * k + 2];
            ec += (double)f.disc_loss_part[4 * (tb + k)]; ee += (double)f.disc_loss_part[4 * (tb + k) + 1];
            eg += (double)f.disc_loss_part[4 * (tb + k) + 2];
        }
        pc = wsum(pc); ec = wsum(ec); pe = wsum(pe); ee = wsum(ee); pg = wsum(pg); eg = wsum(eg);
        if (tid == 0) {
            const double B = (double)f.B;
            const double pl = pc / B, el = ec / B, em = (pe + ee) / (2.0 * B),
                         eloss = -(double)f.ent_scale * em;
            f.disc_losses[0] = (float)pl; f.disc_losses[1] = (float)el; f.disc_losses[2] = (float)em;
            f.disc_losses[3] = (float)eloss; f.disc_losses[4] = (float)(pg / B);
            f.disc_losses[5] = (float)(eg / B);
            f.disc_losses[6] = (float)gp_loss;
            f.disc_losses[7] = (float)(pl + el + eloss + (double)f.gp_coeff * gp_loss);
        }
    }
    if (f.rms_sums) {  // commit of the inline d.obs_rms.update (mpi_running_mean_std.py:40-45,31-34)
        const int dim = f.ob;
        const double add_n = f.rms_addvec[2 * dim];
        const double cnt = f.rms_sums[2 * dim] + add_n;  // every lane reads the old count first
        for (int c = tid; c < dim; c += 64) {
            const double s1 = __dadd_rn(f.rms_sums[c], f.rms_addvec[c]),
                         s2 = __dadd_rn(f.rms_sums[dim + c], f.rms_addvec[dim + c]);
            f.rms_sums[c] = s1;
            f.rms_sums[dim + c] = s2;
            const float mean = (float)(s1 / cnt), msq = (float)(s2 / cnt);
            f.rms_mean[c] = mean;
            f.rms_std[c] = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(mean, mean)), 1e-2f));
        }
        if (tid == 0) f.rms_sums[2 * dim] = cnt;
    }
    if (tid == 0 && tick)
        for (int k = 0; k < f.nseg; ++k)
            if (f.seg[k].step_dev) f.seg[k].step_dev[0] += 1;
}

__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs f) {
    __shared__ double shd[4];
    const int tid = threadIdx.x;
    if (blockIdx.x == gridDim.x - 1) {  // the extra block
        if (tid < 64) finalize_scalars(f, tid);
        return;
    }
    // which segment does this block belong to?
    int seg = 0, b = blockIdx.x;
    while (seg < f.nseg - 1 && b >= f.seg[seg].nblocks) { b -= f.seg[seg].nblocks; ++seg; }
    const Segment& s = f.seg[seg];
    // 4 lanes per element: lane q adds slots q, q+4, ... in increasing order (8 loads in flight per
    // batch), then a fixed 2-level tree.  The GP scalars are computed while the first batch is in flight.
    const int q = tid & 3;
    const int64_t step = (int64_t)s.nblocks * 64;
    int64_t i = (int64_t)b * 64 + (tid >> 2);
    // (a segment of 33..48 slots — the discriminator's — requests its twelve per lane in ONE batch: the second batch
    // was a second memory round trip; every other segment keeps eight, whose four extra predicated loads measured
    // slower there.  Same additions in the same order either way, the zeros included: -0 + 0 = +0.)
    float t[12];
    const bool wide12 = s.nslots > 32 && s.nslots <= 48;
    auto issue = [&](int64_t e, int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 4 * j;
            t[j] = (e < s.n && k < s.nslots) ? s.partials[e + (int64_t)k * s.slot_stride] : 0.f;
        }
        if (wide12) {
#pragma unroll
            for (int j = 8; j < 12; ++j) {
                const int k = k0 + 4 * j;
                t[j] = (e < s.n && k < s.nslots) ? s.partials[e + (int64_t)k * s.slot_stride] : 0.f;
            }
        }
    };
    issue(i, q);
    float kgp; double gp_loss;
    gp_scalars(f, tid & 63, kgp, gp_loss);
    double acc = 0.0;
    for (; i < s.n; i += step) {
        float g0 = 0.f, g1 = 0.f;  // plain slots, gp-scaled slots
        if (wide12) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool plain = q + 4 * j < s.scaled_from;
                const float tj = j < 12 ? t[j] : 0.f;
                g0 += plain ? tj : 0.f;
                g1 += plain ? 0.f : tj;
            }
        } else {
            for (int k0 = q; k0 < s.nslots; k0 += 32) {
                if (k0 != q) issue(i, k0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool plain = k0 + 4 * j < s.scaled_from;
                    g0 += plain ? t[j] : 0.f;
                    g1 += plain ? 0.f : t[j];
                }
            }
        }
        g0 += __shfl_xor(g0, 1, 64); g0 += __shfl_xor(g0, 2, 64);
        g1 += __shfl_xor(g1, 1, 64); g1 += __shfl_xor(g1, 2, 64);
        float g = g0 + kgp * g1;
        // (requesting the decayed kernel's value with the slots, ahead of the sums, measured slower at C2:
        // 36.24 -> 35.98 k steps/s alternating on one box)
        if (s.wd > 0.f && ((i >= s.wd0_lo && i < s.wd0_hi) || (i >= s.wd1_lo && i < s.wd1_hi)))
            g += s.wd * s.param[i];
        if (q == 0) {
            s.grad[i] = g;
            acc += (double)g * (double)g;
        }
        if (i + step < s.n) issue(i + step, q);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) s.sumsq[b] = (float)(shd[0] + shd[1] + shd[2] + shd[3]);
}

// ------------------------------------------------------------------------------------------------
// apply: for each segment, norm from the per-block sums of squares, then (a) the locally clipped
// copy that the reference logs, (b) Adam (+Polyak).  mode bit 0: write clipped, bit 1: Adam.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void apply_kernel(ApplyArgs f) {
    __shared__ double shd[4], sh_pow[2];
    __shared__ float sh_scale, sh_lr;
    const int tid = threadIdx.x;
    int seg = 0, b = blockIdx.x;
    while (seg < f.nseg - 1 && b >= f.seg[seg].nblocks) { b -= f.seg[seg].nblocks; ++seg; }
    const ApplySeg& s = f.seg[seg];
    // this thread's first element is loaded BEFORE the norm / step-size chain below needs anything
    int64_t i = (int64_t)b * 256 + tid;
    const int64_t step = (int64_t)s.nblocks * 256;
    const bool adam = f.mode & 2;
    float g0 = 0.f, m0 = 0.f, v0 = 0.f, p0 = 0.f, t0 = 0.f;
    if (!s.img1 && i < s.n) {
        g0 = s.grad[i];
        if (adam) { m0 = s.m[i]; v0 = s.v[i]; p0 = s.param[i]; if (s.targ) t0 = s.targ[i]; }
    }
    // 256-wide path with weight images to keep (samx_step_apply mode bit 6): a work item is a QUAD of W1 or W2 —
    // rows 4 kg .. 4 kg + 3 of one column j, one 16-byte element of the k-interleaved image and of the target's,
    // four neighbours of the transposed image — or a single element of the rest (biases, LayerNorm, head).
    // Every element is updated by the same arithmetic as on the plain path; only which thread takes it differs.
    // As there, the first item's loads go out before the norm / step-size chain.
    struct Quad { int64_t e[4]; float g[4], m[4], v[4], p[4], t[4]; int n; int64_t kg, j; bool in1, quad; };
    const int64_t K1 = s.img_n1 >> 8, Q1 = ((K1 + 3) >> 2) << 8, Q2 = 64 << 8;
    const int64_t rest1 = s.img_o2 - s.img_n1, nw = s.img1 ? Q1 + Q2 + (s.n - s.img_n1 - 65536) : 0;
    auto fetch = [&](int64_t w, Quad& q) {
        q.quad = w < Q1 + Q2;
        q.in1 = w < Q1;
        if (q.quad) {
            const int64_t x = q.in1 ? w : w - Q1, base = q.in1 ? 0 : s.img_o2, rows = q.in1 ? K1 : 256;
            q.kg = x >> 8; q.j = x & 255;
            q.n = (int)(rows - 4 * q.kg < 4 ? rows - 4 * q.kg : 4);
#pragma unroll
            for (int m = 0; m < 4; ++m) q.e[m] = base + ((4 * q.kg + (m < q.n ? m : 0)) << 8) + q.j;
        } else {
            const int64_t r = w - Q1 - Q2;
            q.n = 1; q.kg = q.j = 0;
            q.e[0] = q.e[1] = q.e[2] = q.e[3] = r < rest1 ? s.img_n1 + r : s.img_o2 + 65536 + (r - rest1);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {   // (rows beyond the matrix re-read its row 4 kg: never stored)
            q.g[m] = s.grad[q.e[m]]; q.m[m] = s.m[q.e[m]]; q.v[m] = s.v[q.e[m]]; q.p[m] = s.param[q.e[m]];
            q.t[m] = s.targ ? s.targ[q.e[m]] : 0.f;
        }
    };
    Quad q0;
    int64_t w0 = (int64_t)b * 256 + tid;
    if (s.img1 && w0 < nw) fetch(w0, q0);
    // The step counter was advanced by finalize (or the host-enqueued tick kernel); here it is only read — blocks
    // of one launch may run in any order.  Its load goes out with the others, and the two float64 powers of the
    // step size run on two waves, ahead of the first barrier (they used to follow it, behind a bare load).
    double tstep = 0.0;
    if (tid == 64 || tid == 128) tstep = (double)s.step_dev[0];
    double acc = 0.0;
    for (int k = tid; k < s.nsumsq; k += 256) acc += (double)s.sumsq[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = acc;
    if (tid == 64) sh_pow[0] = pow((double)f.beta2, tstep);
    if (tid == 128) sh_pow[1] = pow((double)f.beta1, tstep);
    __syncthreads();
    if (tid == 0) {
        const float nrm = (float)sqrt(shd[0] + shd[1] + shd[2] + shd[3]) * f.grad_scale;
        sh_scale = s.clip > 0.f ? s.clip / fmaxf(nrm, s.clip) : 1.f;
    }
    if (tid == 64) sh_lr = (float)((double)s.lr * sqrt(1.0 - sh_pow[0]) / (1.0 - sh_pow[1]));
    __syncthreads();
    const float scale = sh_scale * f.grad_scale, lr_t = sh_lr;
    if (s.img1) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int64_t w = w0; w < nw;) {
            float pn[4] = {0.f, 0.f, 0.f, 0.f}, tn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m >= q0.n) continue;
                const int64_t e = q0.e[m];
                const float gi = q0.g[m] * scale;
                if (f.mode & 1) s.clipped[e] = gi;
                float mi = q0.m[m], vi = q0.v[m];
                mi = mi + (gi - mi) * (1.f - f.beta1);
                vi = vi + (gi * gi - vi) * (1.f - f.beta2);
                pn[m] = q0.p[m] - (mi * lr_t) / (sqrtf(vi) + f.eps);
                s.m[e] = mi; s.v[e] = vi; s.param[e] = pn[m];
                if (s.targ) { tn[m] = (1.f - f.tau) * q0.t[m] + f.tau * pn[m]; s.targ[e] = tn[m]; }
            }
            if (q0.quad) {
                const int64_t at = ((q0.kg << 8) + q0.j) * 4, j = q0.j;
                *(f4*)((q0.in1 ? s.img1 : s.img2) + at) = f4{pn[0], pn[1], pn[2], pn[3]};
                if (s.targ) *(f4*)((q0.in1 ? s.timg1 : s.timg2) + at) = f4{tn[0], tn[1], tn[2], tn[3]};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int64_t k = 4 * q0.kg + m;
                    if (m >= q0.n) continue;
                    if (!q0.in1) s.tr2[(((j >> 2) << 8) + k) * 4 + (j & 3)] = pn[m];
                    else if (s.tr1 && k >= s.tr1_r0)
                        s.tr1[((j >> 2) * (int64_t)s.tr1_J + k - s.tr1_r0) * 4 + (j & 3)] = pn[m];
                }
            }
            w += step;
            if (w < nw) fetch(w, q0);
        }
        return;
    }
    for (; i < s.n; i += step) {
        const float gi = g0 * scale;
        if (f.mode & 1) s.clipped[i] = gi;
        if (adam) {
            float mi = m0, vi = v0;
            mi = mi + (gi - mi) * (1.f - f.beta1);
            vi = vi + (gi * gi - vi) * (1.f - f.beta2);
            const float pi = p0 - (mi * lr_t) / (sqrtf(vi) + f.eps);
            s.m[i] = mi; s.v[i] = vi; s.param[i] = pi;
            if (s.targ) s.targ[i] = (1.f - f.tau) * t0 + f.tau * pi;
        }
        const int64_t nx = i + step;
        if (nx < s.n) {
            g0 = s.grad[nx];
            if (adam) { m0 = s.m[nx]; v0 = s.v[nx]; p0 = s.param[nx]; if (s.targ) t0 = s.targ[nx]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// finalize + apply in ONE launch (samx_step_grads_part bit 9; one process, whole-network path): every quad of lanes
// reduces the tile slots of ONE element as finalize_kernel does and keeps the gradient in a register; the only thing a
// block needs from the others is the clip norm, i.e. their per-block sums of squares — one float each.  A block
// publishes its float as ONE 64-bit word {launch tag, float bits} with an agent-scope atomic store and reads the
// words of its segment with agent-scope atomic loads until they carry this launch's tag: no fence, no second word
// (a device-scope release fence writes the XCD's L2 back — measured at ~15 us per launch when the optimiser pass
// hung behind tickets).  The arithmetic, and the order of every addition, is apply_kernel's: same bits.
// Tags: the workspace keeps a launch counter (epoch[0]); a launch uses epoch + 1 and the extra block stores it back
// once every slot carries it (so every block has read the counter, and the step counters, before they move).
// A word left by an earlier launch never carries the current tag: tags only grow, a fresh workspace is zero (tag 1
// != 0), and a stale word would have to be exactly 2^32 launches old to match (a slot that is not rewritten that
// long belongs to a role subset that did not run in 2^32 launches).
// Every block of the launch must be resident at once: the host refuses grids beyond 1024 blocks (an MI355X holds
// 2048 of these), and the wait is bounded by the wall clock — on expiry the block poisons its parameters with NaN and
// sets epoch[1] (loud, never silent; a healthy device does not get there).
// ------------------------------------------------------------------------------------------------
constexpr long long kSpinTicks = 200000000ll;   // wall_clock64() runs at 100 MHz: 2 s

__device__ __forceinline__ bool wait_slot(const unsigned long long* slot, unsigned tag, float& value) {
    long long t0 = 0;
    for (unsigned it = 0;; ++it) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == tag) { value = __uint_as_float((unsigned)v); return true; }
        if ((it & 63) == 63) {   // the clock is a scalar-memory round trip of its own: looked at every 64th miss
            const long long now = wall_clock64();
            if (it == 63) t0 = now;
            else if (now - t0 > kSpinTicks) { value = 0.f; return false; }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__global__ __launch_bounds__(256) void finalize_apply_kernel(FinalizeArgs f, ApplyArgs a, unsigned long long* slots,
                                                             unsigned* epoch) {
    __shared__ double shd[4], shn[4], sh_pow[2];
    __shared__ float sh_lr;
    const int tid = threadIdx.x;
    const unsigned tag = epoch[0] + 1u;
    if (blockIdx.x == gridDim.x - 1) {  // the extra block: scalars, then the counters once everybody has read them
        if (tid >= 64) return;
        finalize_scalars(f, tid, false);
        bool ok = true;
        for (int k = 0; k < f.nseg; ++k)
            for (int j = tid; j < f.seg[k].nblocks; j += 64) {
                float unused;
                ok = wait_slot(slots + k * kMaxSegBlocks + j, tag, unused) && ok;
            }
        const bool all_ok = __all(ok);
        if (tid == 0) {
            for (int k = 0; k < f.nseg; ++k)
                if (f.seg[k].step_dev) f.seg[k].step_dev[0] += 1;
            epoch[0] = tag;
            if (!all_ok) epoch[1] = 1u;
        }
        return;
    }
    int seg = 0, b = blockIdx.x;
    while (seg < f.nseg - 1 && b >= f.seg[seg].nblocks) { b -= f.seg[seg].nblocks; ++seg; }
    const Segment& s = f.seg[seg];
    const ApplySeg& p = a.seg[seg];
    const int q = tid & 3;
    const int64_t i = (int64_t)b * 64 + (tid >> 2);   // nblocks * 64 >= n (host-checked): one element per quad
    // a lane's slots q, q + 4, ...: eight loads per batch as in finalize_kernel; a segment of 33..48 slots (the
    // discriminator's at every batch size with 16 row tiles per stream) requests all twelve at once — its blocks
    // are the last to publish, and the second batch was a second memory round trip ahead of everybody's wait.
    // The additions run in the same order either way.
    float t[12];
    const bool wide12 = s.nslots > 32 && s.nslots <= 48;
    auto issue = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 4 * j;
            t[j] = (i < s.n && k < s.nslots) ? s.partials[i + (int64_t)k * s.slot_stride] : 0.f;
        }
    };
    issue(q);
    if (wide12) {
#pragma unroll
        for (int j = 8; j < 12; ++j) {
            const int k = q + 4 * j;
            t[j] = (i < s.n && k < s.nslots) ? s.partials[i + (int64_t)k * s.slot_stride] : 0.f;
        }
    }
    // the optimiser's operands of this element travel behind the slots
    const bool own = q == 0 && i < s.n;
    float m0 = 0.f, v0 = 0.f, p0 = 0.f, t0 = 0.f;
    if (own) { m0 = p.m[i]; v0 = p.v[i]; p0 = p.param[i]; if (p.targ) t0 = p.targ[i]; }
    double tstep = 0.0;   // the counter AFTER this update's tick (the extra block advances it behind everybody's read)
    if (tid == 64 || tid == 128) tstep = (double)p.step_dev[0] + 1.0;
    float kgp; double gp_loss;
    gp_scalars(f, tid & 63, kgp, gp_loss);
    float g = 0.f;
    double acc = 0.0;
    if (i < s.n) {
        float g0 = 0.f, g1 = 0.f;
        if (wide12) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {   // (two batches of eight there: the zeros are added too, -0 + 0 = +0)
                const bool plain = q + 4 * j < s.scaled_from;
                const float tj = j < 12 ? t[j] : 0.f;
                g0 += plain ? tj : 0.f;
                g1 += plain ? 0.f : tj;
            }
        } else {
            for (int k0 = q; k0 < s.nslots; k0 += 32) {
                if (k0 != q) issue(k0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool plain = k0 + 4 * j < s.scaled_from;
                    g0 += plain ? t[j] : 0.f;
                    g1 += plain ? 0.f : t[j];
                }
            }
        }
        g0 += __shfl_xor(g0, 1, 64); g0 += __shfl_xor(g0, 2, 64);
        g1 += __shfl_xor(g1, 1, 64); g1 += __shfl_xor(g1, 2, 64);
        g = g0 + kgp * g1;
        if (s.wd > 0.f && ((i >= s.wd0_lo && i < s.wd0_hi) || (i >= s.wd1_lo && i < s.wd1_hi)))
            g += s.wd * s.param[i];
        if (q == 0) {
            s.grad[i] = g;
            acc += (double)g * (double)g;
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = acc;
    if (tid == 64) sh_pow[0] = pow((double)a.beta2, tstep);
    if (tid == 128) sh_pow[1] = pow((double)a.beta1, tstep);
    __syncthreads();
    unsigned long long* my = slots + seg * kMaxSegBlocks;
    if (tid == 0) {
        const float ss = (float)(shd[0] + shd[1] + shd[2] + shd[3]);
        s.sumsq[b] = ss;
        __hip_atomic_store(my + b, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(ss),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the segment's norm: thread k fetches block k's sum (nblocks <= 256), added as apply_kernel adds them
    double part = 0.0;
    bool ok = true;
    if (tid < s.nblocks) {
        float x;
        ok = wait_slot(my + tid, tag, x);
        part += (double)x;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((tid & 63) == 0) shn[tid >> 6] = part;
    // (the step size needs nothing from the other blocks: it is ready behind the same barrier as the norm's sums,
    // and every lane that updates an element finishes norm and clip factor itself — apply_kernel's third barrier
    // is not needed)
    if (tid == 64) sh_lr = (float)((double)p.lr * sqrt(1.0 - sh_pow[0]) / (1.0 - sh_pow[1]));
    const int bad = __syncthreads_or(!ok);
    if (!own) return;
    const float nrm = (float)sqrt(shn[0] + shn[1] + shn[2] + shn[3]) * a.grad_scale;
    const float scale = (p.clip > 0.f ? p.clip / fmaxf(nrm, p.clip) : 1.f) * a.grad_scale, lr_t = sh_lr;
    const float gi = g * scale;
    if (a.mode & 1) p.clipped[i] = gi;
    float mi = m0, vi = v0;
    mi = mi + (gi - mi) * (1.f - a.beta1);
    vi = vi + (gi * gi - vi) * (1.f - a.beta2);
    float pi = p0 - (mi * lr_t) / (sqrtf(vi) + a.eps);
    if (bad) pi = __uint_as_float(0x7fc00000u);   // the wait expired: never a silent wrong update
    p.m[i] = mi; p.v[i] = vi; p.param[i] = pi;
    if (p.targ) p.targ[i] = (1.f - a.tau) * t0 + a.tau * pi;
}

__global__ void tick_kernel(int64_t* a, int64_t* b, int64_t* c) {
    if (threadIdx.x == 0) {
        if (a) a[0] += 1;
        if (b) b[0] += 1;
        if (c) c[0] += 1;
    }
}

// (ta / tb / tc: step counters to advance in the same launch — the layer-wise path's tick — or null)
__global__ __launch_bounds__(256) void seg_sumsq_kernel(ApplyArgs f, int64_t* ta, int64_t* tb, int64_t* tc) {
    __shared__ double shd[4];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) {
        if (ta) ta[0] += 1;
        if (tb) tb[0] += 1;
        if (tc) tc[0] += 1;
    }
    int seg = 0, b = blockIdx.x;
    while (seg < f.nseg - 1 && b >= f.seg[seg].nblocks) { b -= f.seg[seg].nblocks; ++seg; }
    const ApplySeg& s = f.seg[seg];
    double acc = 0.0;
    for (int64_t i = (int64_t)b * 256 + tid; i < s.n; i += (int64_t)s.nblocks * 256)
        acc += (double)s.grad[i] * (double)s.grad[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) s.sumsq[b] = (float)(shd[0] + shd[1] + shd[2] + shd[3]);
}

size_t agent_lds_bytes(int ob, int ac, bool use_d) {
    (void)use_d;  // the target+critic role always keeps three networks resident
    size_t f = net_lds_floats(ob, ac) + 2 * net_lds_floats(ob + ac, 1);
    f += 8 * R * LD + R * LD3 + 144;
    return f * sizeof(float);
}
size_t disc_lds_bytes(int in) {
    size_t f = net_lds_floats(in, 1) + 6 * R * LD + 96 + 128;
    return f * sizeof(float);
}

}  // namespace

bool supported(const samx_mlp* d, int64_t B) {
    return d && d->h1 == H && d->h2 == H && d->in_dim <= 64 && d->out_dim <= 16 && B > 0 &&
           B % R == 0;
}

// the agent roles keep three networks + their tiles in LDS: wide inputs do not fit (-> layer-wise path)
bool fits_lds(int ob, int ac) {
    return agent_lds_bytes(ob, ac, true) <= 150 * 1024 && disc_lds_bytes(ob + ac) <= 150 * 1024;
}

int seg_blocks(int64_t n) {
    int64_t b = (n + 63) / 64;
    return (int)(b < 1 ? 1 : (b > kMaxSegBlocks ? kMaxSegBlocks : b));
}

int launch_step(const AgentArgs* a, const DiscArgs* d, hipStream_t st) {
    StepArgs s;
    size_t lds = 0;
    s.n_a = s.n_tc = s.n_disc = 0;
    if (a) {
        s.a = *a;
        s.n_a = s.n_tc = (int)(a->B / R);
        lds = agent_lds_bytes(a->ob, a->ac, a->d_scale != 0.f);
    }
    if (d) {
        s.d = *d;
        s.n_disc = 3 * (int)(d->B / R);
        const size_t l2 = disc_lds_bytes(d->ob + d->ac);
        if (l2 > lds) lds = l2;
    }
    SAMX_REQUIRE(lds <= 150 * 1024, "fused step kernel needs %zu bytes of LDS", lds);
    // profiling aid: SAMX_FUSED_ROLES=<bitmask> (1 actor role, 2 target+critic role, 4 discriminator)
    // launches a subset of the roles; results are then incomplete — never set it in production
    static int roles = -1;
    if (roles < 0) {
        const char* e = getenv("SAMX_FUSED_ROLES");
        roles = e ? atoi(e) : 7;
    }
    if (!(roles & 1)) s.n_a = 0;
    if (!(roles & 2)) s.n_tc = 0;
    if (!(roles & 4)) s.n_disc = 0;
    if (s.n_a + s.n_tc + s.n_disc == 0) return SAMX_OK;
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)step_kernel, lds)) return rc;
    step_kernel<<<(unsigned)(s.n_a + s.n_tc + s.n_disc), NT, lds, st>>>(s);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_ret_popart(const float* tq, const float* tqn, int64_t B, double* sums, float* mean, float* stdv,
                      int popart, float* critic, float* targ_critic, const samx_mlp* dc, int phase,
                      double* addvec, hipStream_t st) {
    // 1 output; LayerNorm adds beta/gamma behind each hidden bias (the flat layout of samx_mlp)
    const int h1 = dc->h1, h2 = dc->h2;
    const int64_t oW3 = (int64_t)dc->in_dim * h1 + h1 + (int64_t)h1 * h2 + h2 + (dc->layernorm ? 2 * (h1 + h2) : 0),
                  ob3 = oW3 + h2;
    SAMX_REQUIRE(phase == 0 || addvec, "launch_ret_popart: phase %d needs addvec", phase);
    SAMX_REQUIRE(dc->out_dim == 1, "launch_ret_popart: the critic has one output");
    ret_popart_kernel<<<1, 256, 0, st>>>(tq, tqn, B, sums, mean, stdv, popart, critic + oW3, critic + ob3,
                                         targ_critic + oW3, targ_critic + ob3, phase, addvec, h2);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_targets(const AgentArgs* a, hipStream_t st) {
    const int tiles = (int)(a->B / R);
    const size_t lds = agent_lds_bytes(a->ob, a->ac, false);
    SAMX_REQUIRE(lds <= 150 * 1024, "fused targets kernel needs %zu bytes of LDS", lds);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)targets_kernel, lds)) return rc;
    targets_kernel<<<(unsigned)(tiles * (a->n_idx ? 2 : 1)), NT, lds, st>>>(*a, tiles);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_td(const samx_mlp* dc, const float* critic, const float* obs0, const float* acs, const float* tc1s,
              const float* mean, const float* stdv, int64_t B, int ob, const int64_t* idx, int64_t ring_start,
              int64_t ring_len, float* td_out, const float* ret_mean, const float* ret_std, float* td_copy,
              int64_t* idx_copy, hipStream_t st) {
    TdArgs a;
    a.td_copy = td_copy; a.idx_copy = idx ? idx_copy : nullptr;
    a.ret_mean = ret_mean; a.ret_std = ret_std;
    a.obs0 = obs0; a.acs = acs; a.tc1s = tc1s; a.mean = mean; a.stdv = stdv; a.critic = critic;
    a.td_out = td_out; a.ob = ob; a.ac = dc->in_dim - ob; a.idx = idx; a.ring_start = ring_start;
    a.ring_len = ring_len;
    const size_t lds = (net_lds_floats(dc->in_dim, 1) + 3 * R * LD + 16) * sizeof(float);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)td_kernel, lds)) return rc;
    td_kernel<<<(unsigned)(B / R), NT, lds, st>>>(a);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_finalize(const FinalizeArgs& f, hipStream_t st) {
    int blocks = 0;
    for (int i = 0; i < f.nseg; ++i) blocks += f.seg[i].nblocks;
    finalize_kernel<<<blocks + 1, 256, 0, st>>>(f);  // + the scalar block
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_finalize_apply(const FinalizeArgs& f, const ApplyArgs& a, unsigned long long* slots, unsigned* epoch,
                          hipStream_t st) {
    SAMX_REQUIRE(slots && epoch && f.nseg == a.nseg && (a.mode & 2), "finalize+apply: bad arguments");
    int blocks = 0;
    for (int i = 0; i < f.nseg; ++i) {
        SAMX_REQUIRE(f.seg[i].n == a.seg[i].n && f.seg[i].step_dev && !a.seg[i].img1 &&
                         (int64_t)f.seg[i].nblocks * 64 >= f.seg[i].n && f.seg[i].nblocks <= kMaxSegBlocks,
                     "finalize+apply: segment %d does not fit one element per lane quad", i);
        blocks += f.seg[i].nblocks;
    }
    SAMX_REQUIRE(blocks + 1 <= 1024, "finalize+apply: %d blocks must all be resident at once", blocks + 1);
    finalize_apply_kernel<<<blocks + 1, 256, 0, st>>>(f, a, slots, epoch);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_apply(const ApplyArgs& f, hipStream_t st) {
    int blocks = 0;
    for (int i = 0; i < f.nseg; ++i) blocks += f.seg[i].nblocks;
    apply_kernel<<<blocks, 256, 0, st>>>(f);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

#ifdef SAMX_TRACE
extern "C" int samx_debug_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 64);
}
#endif

int launch_tick(int64_t* a, int64_t* b, int64_t* c, hipStream_t st) {
    tick_kernel<<<1, 64, 0, st>>>(a, b, c);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_seg_sumsq(const ApplyArgs& f, hipStream_t st, int64_t* ta, int64_t* tb, int64_t* tc) {
    int blocks = 0;
    for (int i = 0; i < f.nseg; ++i) blocks += f.seg[i].nblocks;
    seg_sumsq_kernel<<<blocks, 256, 0, st>>>(f, ta, tb, tc);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // namespace fused
}  // namespace samx
