// Whole-network fused kernels for 64x64 nets WITH post-activation LayerNorm — the reference's canonical
// launcher configuration (launchers/imitation/mujoco_sam_imitate.sh:34 --with_layernorm;
// networks.py:53-70, network_util.py:32-39: a = lrelu(xW+b), xh = (a-mu)*rs, n = gamma*xh + beta,
// biased variance, eps 1e-8).  Same tile structure as fused_small.hip (16 batch rows per workgroup, all
// weights and activations in LDS, every matmul on v_mfma_f32_16x16x4_f32); LayerNorm adds one
// row-local "mini-phase" per layer and direction in which thread t owns row t >> 4 and the columns
// (t & 15) + 16 j — four values in registers, row statistics by xor-shuffles inside the 16-lane group.
//
// Gradient penalty with LayerNorm (discriminator.py:130-140): the second-order term is the
// reverse-mode gradient of sum_rows sdot, sdot = the forward-mode tangent of the score along
// xdot = V = dGP/dg.  The tangent of a LayerNorm layer depends on the primal activation, so the backward
// pass carries a tangent cotangent stream T (identical to the first backward that produced g: reused)
// and a primal cotangent stream P (derivation: DESIGN.md 3.5; formulas as in nets.hip ln_tan_kernel).
//
// With LayerNorm the TD targets always come from their own launch (targets_kernel_ln): the target+critic
// role then keeps one network resident instead of three.
#include "fused_dev.h"
#include <stdlib.h>

namespace samx {
namespace fused {
namespace {

constexpr float kLnEps = 1e-8f;
constexpr float kInvH = 1.f / 64.f;

__device__ __forceinline__ float sum16(float v) {   // all-reduce inside a 16-lane group (one batch row)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}
// sum over the 4 rows a wave holds (lanes l, l+16, l+32, l+48): valid in lanes 0..15
__device__ __forceinline__ float sum_rows4(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// flat offsets of a LayerNorm net's variables (include/samx.h: W1 b1 beta1 gamma1 W2 b2 beta2 gamma2 W3 b3)
struct Offs { int64_t W1, b1, be1, W2, b2, be2, W3, b3; };
__device__ __forceinline__ Offs offs_ln(int in, int out) {
    Offs o;
    o.W1 = 0; o.b1 = (int64_t)in * H; o.be1 = o.b1 + H; o.W2 = o.be1 + 2 * H; o.b2 = o.W2 + H * H;
    o.be2 = o.b2 + H; o.W3 = o.be2 + 2 * H; o.b3 = o.W3 + (int64_t)H * out;
    return o;
}

// ---- LayerNorm forward mini-phase: A (post-lrelu) -> N = gamma*xh + beta, stat = [mu(16) | rs(16)].
// N may alias A (forward-only passes) or be nullptr (statistics only).  head != nullptr: also the
// 1-output head s[r] = <n[r], w3> + b3 of the SAME rows (saves the head_dot phase).
__device__ inline void ln_fwd_phase(const float* A, float* N, float* stat, const float* be, const float* ga,
                                    const NetL* head = nullptr, float* s_out = nullptr) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = A[r * LD + c0 + 16 * j];
    const float mu = sum16((v[0] + v[1]) + (v[2] + v[3])) * kInvH;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; q = fmaf(d, d, q); }
    const float rs = 1.f / sqrtf(sum16(q) * kInvH + kLnEps);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        const float n = fmaf(ga[c], (v[j] - mu) * rs, be[c]);
        if (N) N[r * LD + c] = n;
        if (head) dot = fmaf(n, head->W3[c * LD3], dot);
    }
    if (head) {
        dot = sum16(dot);
        if (c0 == 0) s_out[r] = dot + head->b3[0];
    }
    if (c0 == 0) { stat[r] = mu; stat[16 + r] = rs; }
}

// ---- LayerNorm backward mini-phase: cotangent of n -> cotangent of the pre-activation z.
//   dn = DN[r][c]  (DN != nullptr)   or   dq[r] * w3[c]  (head form)
//   dxh = dn*gamma ; da = rs (dxh - mean dxh - xh mean(dxh xh)) [+ add] ; DZ = da * lrelu'(a)
// DZ may alias DN.  PS != nullptr: per-wave column sums of dn (-> dbeta) and dn*xh (-> dgamma) over the
// wave's 4 rows into PS[w][2][64]; ps_commit adds the four waves in a fixed order.
__device__ inline void ln_bwd_phase(const float* DN, const float* dq, const float* w3, const float* A,
                                    const float* stat, const float* ga, const float* add, float* DZ, float* PS) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float mu = stat[r], rs = stat[16 + r];
    float a[4], xh[4], dn[4], dx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        a[j] = A[r * LD + c];
        xh[j] = (a[j] - mu) * rs;
        dn[j] = DN ? DN[r * LD + c] : dq[r] * w3[c * LD3];
        dx[j] = dn[j] * ga[c];
        s1 += dx[j];
        s2 = fmaf(dx[j], xh[j], s2);
    }
    const float m1 = sum16(s1) * kInvH, m2 = sum16(s2) * kInvH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        float da = rs * (dx[j] - m1 - xh[j] * m2);
        if (add) da += add[r * LD + c];
        DZ[r * LD + c] = da * dlrelu(a[j]);
    }
    if (PS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = sum_rows4(dn[j]), g = sum_rows4(dn[j] * xh[j]);
            if (lane < 16) { PS[w * 128 + c0 + 16 * j] = b; PS[w * 128 + 64 + c0 + 16 * j] = g; }
        }
    }
}

// dst[0..63] = dbeta, dst[64..127] = dgamma (contiguous in the flat layout) from PS [+ extra gamma partials]
__device__ __forceinline__ void ps_commit(const float* PS, float* __restrict__ dst, const float* PSg = nullptr) {
    const int t = threadIdx.x;
    if (t < 128) {
        float v = (PS[t] + PS[128 + t]) + (PS[256 + t] + PS[384 + t]);
        if (PSg && t >= 64) v += (PSg[t - 64] + PSg[64 + t - 64]) + (PSg[128 + t - 64] + PSg[192 + t - 64]);
        dst[t] = v;
    }
}

// ---- tangent of a LayerNorm layer along adot (= zdot .* lrelu'(a)) and the primal cotangent it sends back:
//   d = adot - mean adot ; c = mean(xh adot) ; u = rs (d - c xh) ; ndot = gamma u
//   ub = gamma tn ; Aq = <ub, d> ; Bq = <ub, xh> ; ta = rs (ub - mean ub - xh Bq/H)       (= J(gamma tn))
//   pa = -(rs^2 (Aq - c Bq)/H) xh - (rs Bq/H) u - rs c ta          (times lrelu'(a) when mask_out)
// tn = TN[r][c] or w3[c] (head).  ND (may be nullptr, may alias AD) receives ndot; PA may alias AD.
// PSn (optional): per-wave column sums of ndot [w][64]; PSg (optional): of tn*u [w][64].
__device__ inline void ln_tan_phase(const float* AD, const float* A, const float* stat, const float* ga,
                                    const float* TN, const float* w3, float* ND, float* PA, bool mask_out,
                                    float* PSn, float* PSg) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float mu = stat[r], rs = stat[16 + r];
    float ad[4], a[4], xh[4], tn[4], ub[4], s1 = 0.f, s2 = 0.f, sa = 0.f, su = 0.f, sb = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        ad[j] = AD[r * LD + c];
        a[j] = A[r * LD + c];
        xh[j] = (a[j] - mu) * rs;
        tn[j] = TN ? TN[r * LD + c] : w3[c * LD3];
        ub[j] = ga[c] * tn[j];
        s1 += ad[j];
        s2 = fmaf(xh[j], ad[j], s2);
        sa = fmaf(ub[j], ad[j], sa);
        su += ub[j];
        sb = fmaf(ub[j], xh[j], sb);
    }
    const float mean_ad = sum16(s1) * kInvH, cc = sum16(s2) * kInvH;
    const float su_ = sum16(su), Aq = sum16(sa) - mean_ad * su_, Bq = sum16(sb);
    const float k1 = -(rs * rs * (Aq - cc * Bq) * kInvH), k2 = -(rs * Bq * kInvH), k3 = -(rs * cc);
    const float mub = su_ * kInvH, mbx = Bq * kInvH;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        const float u = rs * ((ad[j] - mean_ad) - cc * xh[j]);
        const float nd = ga[c] * u;
        const float ta = rs * (ub[j] - mub - xh[j] * mbx);
        const float pa = k1 * xh[j] + k2 * u + k3 * ta;
        if (ND) ND[r * LD + c] = nd;
        PA[r * LD + c] = mask_out ? pa * dlrelu(a[j]) : pa;
        if (PSn) { const float t = sum_rows4(nd); if (lane < 16) PSn[w * 64 + c] = t; }
        if (PSg) { const float t = sum_rows4(tn[j] * u); if (lane < 16) PSg[w * 64 + c] = t; }
    }
}

// dN1[16][64] = dZ2[16][64] * W2^T   (no mask: the LayerNorm backward comes next)
__device__ inline void layer_bwd_lin(const float* dZ2, const float* W2, float* dN1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc = tile_nt64(dZ2, W2, 16 * w, lane);
    const int c = 16 * w + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dN1[(r0 + j) * LD + c] = acc[j];
}

// partial weight gradient with two products: dst = X1^T D1 + X2^T D2 (X* [16 x nI], D* [16 x 64]);
// bias (optional) = column sums of D2
__device__ inline void wgrad64x2(const float* X1, const float* D1, const float* X2, const float* D2, int nI,
                                 float* __restrict__ dstW, float* __restrict__ dstb) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ti = (nI + 15) / 16;
    for (int t = w; t < ti * 4; t += 4) {
        const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int c_ = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int k = 0; k < R; k += 4) acc = mfma4(X1[(k + kq) * LD + i0 + c_], D1[(k + kq) * LD + j0 + c_], acc);
#pragma unroll
        for (int k = 0; k < R; k += 4) acc = mfma4(X2[(k + kq) * LD + i0 + c_], D2[(k + kq) * LD + j0 + c_], acc);
        const int c = j0 + (lane & 15), r0 = i0 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < nI) dstW[(r0 + j) * H + c] = acc[j];
    }
    if (dstb && threadIdx.x < H) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += D2[r * LD + threadIdx.x];
        dstb[threadIdx.x] = s;
    }
}

// ---- backward of a 1-output LayerNorm net with weight gradients (critic loss / D policy+expert rows):
// given dq[16] (cotangent of the score), the forward's tiles (XC, A1, N1, A2, N2) and statistics, fills
// every variable of this tile's gradient slot.  6 barriers.
__device__ inline void net1_bwd_ln(const NetL& n, const float* XC, int in, const float* A1, const float* N1,
                                   const float* A2, const float* N2, const float* st1, const float* st2,
                                   const float* dq, float* DZa, float* DZb, float* PS, float* __restrict__ slot) {
    const Offs o = offs_ln(in, 1);
    wgrad_head1(N2, dq, slot + o.W3, slot + o.b3, R);
    ln_bwd_phase(nullptr, dq, n.W3, A2, st2, n.ga2, nullptr, DZa, PS);
    __syncthreads();
    ps_commit(PS, slot + o.be2);
    wgrad64(N1, H, DZa, slot + o.W2, slot + o.b2, R);
    layer_bwd_lin(DZa, n.W2, DZb);
    __syncthreads();
    ln_bwd_phase(DZb, nullptr, nullptr, A1, st1, n.ga1, nullptr, DZb, PS);
    __syncthreads();
    ps_commit(PS, slot + o.be1);
    wgrad64(XC, in, DZb, slot + o.W1, slot + o.b1, R);
}

struct LnLds {   // tiles and strips of the agent roles
    NetL nA, nC, nD;
    float *XC, *A1a, *N1a, *A2a, *N2a, *A1c, *A2c, *DZa, *DZb, *PRE;
    float *s_, *tq, *dq, *red, *rw, *dn, *iw, *tl, *son, *sa1, *sa2, *sc1, *sc2, *PS;
};
constexpr int kLnStrip = 16 * 9 + 4 * 32 + 512;   // floats behind the tiles

// nets: bit 0 actor, bit 1 critic, bit 2 reward net; ntiles: how many of the 9 [16][LD] tiles are carved
__device__ inline LnLds carve_ln(float* p, int ob, int ac, int nets, int ntiles) {
    LnLds L;
    L.nA = L.nC = L.nD = NetL();
    if (nets & 1) L.nA = carve_net(p, ob, ac, true);
    if (nets & 2) L.nC = carve_net(p, ob + ac, 1, true);
    L.nD = L.nC;
    if (nets & 4) L.nD = carve_net(p, ob + ac, 1, true);
    float* t[9];
    for (int i = 0; i < 9; ++i) { t[i] = p; if (i < ntiles) p += R * LD; }
    L.XC = t[0]; L.A1c = t[1]; L.A2c = t[2]; L.DZa = t[3]; L.DZb = t[4];
    L.A1a = t[5]; L.N1a = t[6]; L.A2a = t[7]; L.N2a = t[8];
    L.PRE = p; p += R * LD3;
    L.s_ = p; L.tq = p + 16; L.dq = p + 32; L.red = p + 48; L.rw = p + 64; L.dn = p + 80; L.iw = p + 96;
    L.tl = p + 112; L.son = p + 128; p += 144;
    L.sa1 = p; L.sa2 = p + 32; L.sc1 = p + 64; L.sc2 = p + 96; p += 128;
    L.PS = p;
    return L;
}
__host__ __device__ inline size_t ln_lds_floats(int ob, int ac, int nets, int ntiles) {
    size_t f = 0;
    if (nets & 1) f += net_lds_floats(ob, ac, true);
    if (nets & 2) f += net_lds_floats(ob + ac, 1, true);
    if (nets & 4) f += net_lds_floats(ob + ac, 1, true);
    return f + (size_t)ntiles * R * LD + R * LD3 + kLnStrip;
}

// actor head on the LayerNorm output of the second layer (tanh * high, clipped) into XC's action columns
__device__ inline void actor_head_ln(const LnLds& L, const float* N2, int ob, int ac, float high) {
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) == 0) {
        f32x4 acc = tile_nn(N2, LD, L.nA.W3, LD3, H, 0, lane);
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

// forward of a 1-output LayerNorm net on the rows in X (forward only: n overwrites a), score into s_out
__device__ inline void net1_fwd_only_ln(const NetL& n, const float* X, float* T1, float* T2, float* stat,
                                        float* s_out) {
    layer_fwd(X, n.k4, n.W1, n.b1, T1);
    __syncthreads();
    ln_fwd_phase(T1, T1, stat, n.be1, n.ga1);
    __syncthreads();
    layer_fwd(T1, H, n.W2, n.b2, T2);
    __syncthreads();
    ln_fwd_phase(T2, nullptr, stat, n.be2, n.ga2, &n, s_out);
    __syncthreads();
}

// TD targets (1-step: pass 0, n-step: pass 1) with the target nets: same contract as agent_t_body
__device__ inline void agent_t_body_ln(const AgentArgs& a, int wg, float* lds, int pass) {
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac;
    LnLds L = carve_ln(lds, ob, ac, 3, 3);
    const int64_t* idx = pass ? a.n_idx : a.idx;
    const int64_t pr = ring_row(idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    const int64_t prs = ring_row(idx, a.ring_start, a.ring_len, row0 + (tid & 15));
    NetRegs wA, wC;
    net_issue<true>(wA, L.nA, a.targ_actor);
    net_issue<true>(wC, L.nC, a.targ_critic);
    const HeadRegs hA = head_issue<true>(L.nA, a.targ_actor), hC = head_issue<true>(L.nC, a.targ_critic);
    float rw = 0.f, dn = 0.f, tl = 1.f;
    if (tid < R) {
        if (pass) { rw = a.n_rews[row0 + tid]; dn = a.n_dones[row0 + tid]; tl = a.n_len[row0 + tid]; }
        else if (a.td_len) tl = a.td_len[row0 + tid];
    }
    RowRegs x1;
    rows_issue(x1, pr, a.obs1, nullptr, a.mean, a.stdv, ob, 0, ob);
    if (tid < R && !pass) { rw = a.rews[prs]; dn = a.dones1[prs]; }
    net_commit<true>(wA, L.nA);
    net_commit<true>(wC, L.nC);
    head_commit(L.nA, hA);
    head_commit(L.nC, hC);
    rows_commit(L.XC, x1, a.mean != nullptr, ob, ob);
    if (tid < R) { L.rw[tid] = rw; L.dn[tid] = dn; L.tl[tid] = tl; }
    stage_wait();
    layer_fwd(L.XC, L.nA.k4, L.nA.W1, L.nA.b1, L.A1c);
    __syncthreads();
    ln_fwd_phase(L.A1c, L.A1c, L.sa1, L.nA.be1, L.nA.ga1);
    __syncthreads();
    layer_fwd(L.A1c, H, L.nA.W2, L.nA.b2, L.A2c);
    __syncthreads();
    ln_fwd_phase(L.A2c, L.A2c, L.sa2, L.nA.be2, L.nA.ga2);
    __syncthreads();
    actor_head_ln(L, L.A2c, ob, ac, a.high);
    __syncthreads();
    net1_fwd_only_ln(L.nC, L.XC, L.A1c, L.A2c, L.sc1, L.s_);
    if (tid < R) {
        const bool lens = pass || a.td_len;
        const float gam = lens ? powf(a.gamma, L.tl[tid]) : a.gamma;
        float qp = L.s_[tid];
        if (a.ret_mean) qp = qp * a.ret_std[0] + a.ret_mean[0];              // sam_agent.py:199-201
        const float t = L.rw[tid] + gam * ((1.f - L.dn[tid]) * qp);
        (pass ? a.targ_qn_out : a.targ_q_out)[row0 + tid] = t;
    }
}

// Critic loss + gradient against the stored targets (a.ext_tq, a.ext_tqn): same contract as agent_c_body
__device__ inline void agent_c_body_ln(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    LnLds L = carve_ln(lds, ob, ac, 2, 7);
    float* N1 = L.A1a; float* N2 = L.N1a;
    float* gC = a.partials + (int64_t)wg * a.P + a.Pa;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wC;
    net_issue<true>(wC, L.nC, a.critic);
    const HeadRegs hC = head_issue<true>(L.nC, a.critic);
    float iw = 1.f, t1 = 0.f, tn = 0.f;
    if (tid < R) {
        if (a.iws) iw = a.iws[row0 + tid];
        t1 = a.ext_tq[row0 + tid];
        if (a.ext_tqn) tn = a.ext_tqn[row0 + tid];
    }
    const float rmean = a.ret_mean ? a.ret_mean[0] : 0.f, rstd = a.ret_std ? a.ret_std[0] : 1.f;
    RowRegs x0;
    rows_issue(x0, pr, a.obs0, a.acs, a.mean, a.stdv, ob, ac, in_c);
    net_commit<true>(wC, L.nC);
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
    ln_fwd_phase(L.A1c, N1, L.sc1, L.nC.be1, L.nC.ga1);
    __syncthreads();
    layer_fwd(N1, H, L.nC.W2, L.nC.b2, L.A2c);
    __syncthreads();
    ln_fwd_phase(L.A2c, N2, L.sc2, L.nC.be2, L.nC.ga2, &L.nC, L.s_);
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
    net1_bwd_ln(L.nC, L.XC, in_c, L.A1c, N1, L.A2c, N2, L.sc1, L.sc2, L.dq, L.DZa, L.DZb, L.PS, gC);
}

// dX-only pass through a 1-output LayerNorm net evaluated on [obz0 | mu(obz0)] (the actor loss's Q and D
// branches): forward (the LayerNorm outputs are transient, in T), sum of the scores into loss slot `li`,
// backward to the first layer's pre-activation cotangent in L.DZb.
__device__ inline void through_net_ln(const LnLds& L, const NetL& n, const AgentArgs& a, int wg, float scale,
                                      int li) {
    const int tid = threadIdx.x;
    float* T = L.DZa;   // transient LayerNorm output (free during the forward)
    layer_fwd(L.XC, n.k4, n.W1, n.b1, L.A1c);
    __syncthreads();
    ln_fwd_phase(L.A1c, T, L.sc1, n.be1, n.ga1);
    __syncthreads();
    layer_fwd(T, H, n.W2, n.b2, L.A2c);
    __syncthreads();
    ln_fwd_phase(L.A2c, nullptr, L.sc2, n.be2, n.ga2, &n, L.s_);
    if (tid < R) L.dq[tid] = scale;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += L.s_[r];
        a.loss_part[wg * kLossStride + li] = s;
    }
    ln_bwd_phase(nullptr, L.dq, n.W3, L.A2c, L.sc2, n.ga2, nullptr, L.DZa, nullptr);
    __syncthreads();
    layer_bwd_lin(L.DZa, n.W2, L.DZb);
    __syncthreads();
    ln_bwd_phase(L.DZb, nullptr, nullptr, L.A1c, L.sc1, n.ga1, nullptr, L.DZb, nullptr);
    __syncthreads();
}

// Actor loss + gradient through the critic (and the reward net): same contract as agent_a_body
__device__ inline void agent_a_body_ln(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac;
    const bool use_d = a.d_scale != 0.f;
    LnLds L = carve_ln(lds, ob, ac, use_d ? 7 : 3, 9);
    float* gA = a.partials + (int64_t)wg * a.P;
    const Offs o = offs_ln(ob, ac);
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wA, wC, wD;
    net_issue<true>(wA, L.nA, a.actor);
    net_issue<true>(wC, L.nC, a.critic);
    if (use_d) net_issue<true>(wD, L.nD, a.reward);
    const HeadRegs hA = head_issue<true>(L.nA, a.actor), hC = head_issue<true>(L.nC, a.critic);
    HeadRegs hD = hC;
    if (use_d) hD = head_issue<true>(L.nD, a.reward);
    RowRegs x0;
    rows_issue(x0, pr, a.obs0, nullptr, a.mean, a.stdv, ob, 0, ob);
    net_commit<true>(wA, L.nA);
    net_commit<true>(wC, L.nC);
    if (use_d) net_commit<true>(wD, L.nD);
    head_commit(L.nA, hA);
    head_commit(L.nC, hC);
    if (use_d) head_commit(L.nD, hD);
    rows_commit(L.XC, x0, a.mean != nullptr, ob, ob);
    stage_wait();
    layer_fwd(L.XC, L.nA.k4, L.nA.W1, L.nA.b1, L.A1a);
    __syncthreads();
    ln_fwd_phase(L.A1a, L.N1a, L.sa1, L.nA.be1, L.nA.ga1);
    __syncthreads();
    layer_fwd(L.N1a, H, L.nA.W2, L.nA.b2, L.A2a);
    __syncthreads();
    ln_fwd_phase(L.A2a, L.N2a, L.sa2, L.nA.be2, L.nA.ga2);
    __syncthreads();
    actor_head_ln(L, L.N2a, ob, ac, a.high);
    __syncthreads();
    const float rstd_ = a.ret_std ? a.ret_std[0] : 1.f;   // d(de-normalised Q)/d(critic output)
    // Q branch: -q_scale * mean Q(obz0, mu(obz0))   (sam_agent.py:412-413)
    through_net_ln(L, L.nC, a, wg, -a.q_scale / (float)a.B * rstd_, 1);
    f32x4 dA = {0.f, 0.f, 0.f, 0.f};
    if (wv == 0) dA = tile_nt64(L.DZb, L.nC.W1, ob, lane);   // dQ/da = dZ1c * W1c[ob:ob+ac, :]^T
    __syncthreads();
    if (use_d) {  // D branch: -d_scale * mean D(obz0, mu(obz0))   (sam_agent.py:421-422)
        through_net_ln(L, L.nD, a, wg, -a.d_scale / (float)a.B * rstd_, 2);
        if (wv == 0) dA += tile_nt64(L.DZb, L.nD.W1, ob, lane);
        __syncthreads();
    } else if (tid == 0) {
        a.loss_part[wg * kLossStride + 2] = 0.f;
    }
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
    {   // actor head gradient: dW3a[i][c] = sum_r N2a[r][i] * PRE[r][c]
        f32x4 acc = tile_tn(L.N2a, LD, 16 * wv, L.PRE, LD3, 0, lane);
        const int c = lane & 15, r0 = 16 * wv + (lane >> 4) * 4;
        if (c < ac) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gA[o.W3 + (int64_t)(r0 + j) * ac + c] = acc[j];
        }
        if (tid < ac) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += L.PRE[r * LD3 + tid];
            gA[o.b3 + tid] = s;
        }
    }
    {   // dN2a = PRE * W3a^T : reduction over the (zero-padded) action index
        f32x4 acc = tile_nt(L.PRE, LD3, L.nA.W3, LD3, 16, 16 * wv, lane);
        const int c = 16 * wv + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) L.DZa[(r0 + j) * LD + c] = acc[j];
    }
    __syncthreads();
    ln_bwd_phase(L.DZa, nullptr, nullptr, L.A2a, L.sa2, L.nA.ga2, nullptr, L.DZa, L.PS);
    __syncthreads();
    ps_commit(L.PS, gA + o.be2);
    wgrad64(L.N1a, H, L.DZa, gA + o.W2, gA + o.b2, R);
    layer_bwd_lin(L.DZa, L.nA.W2, L.DZb);
    __syncthreads();
    ln_bwd_phase(L.DZb, nullptr, nullptr, L.A1a, L.sa1, L.nA.ga1, nullptr, L.DZb, L.PS);
    __syncthreads();
    ps_commit(L.PS, gA + o.be1);
    wgrad64(L.XC, ob, L.DZb, gA + o.W1, gA + o.b1, R);
}

// ------------------------------------------------------------------------------------------------
// Discriminator tiles (same tile map as disc1_body: [0, 2B/16) policy / expert, [2B/16, 3B/16) interpolates)
// ------------------------------------------------------------------------------------------------
constexpr int kDiscLnTiles = 12;
constexpr int kDiscLnStrip = 96 + 128 + 64 + 512 + 3 * 256;

__device__ inline void disc1_body_ln(const DiscArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x;
    const int tiles_b = (int)(a.B / R);
    const int sect = wg / tiles_b;  // 0 policy, 1 expert, 2 interpolate
    const int64_t row0 = (int64_t)(wg % tiles_b) * R;
    const int in = a.ob + a.ac;
    float* p = lds;
    NetL n = carve_net(p, in, 1, true);
    float* T[kDiscLnTiles];
    for (int i = 0; i < kDiscLnTiles; ++i) { T[i] = p; p += R * LD; }
    float* XC = T[0]; float* A1 = T[1]; float* N1 = T[2]; float* A2 = T[3]; float* DZa = T[4]; float* DZb = T[5];
    float* sh = p; p += 96;
    float* ms = p; p += 128;  // mean[64] | std[64] when the statistics are updated inline
    float* st1 = p; float* st2 = p + 32; p += 64;
    float* PS = p; p += 512;
    float* PSg1 = p; float* PSg2 = p + 256; float* PS3 = p + 512; p += 768;
    float* s_ = sh; float* dq = sh + 16; float* r0_ = sh + 32; float* r1_ = sh + 48; float* r2_ = sh + 64;
    // ---- prologue: all global loads issued before the first wait (fused_dev.h "staged prologue")
    const int64_t pg = ring_row(a.p_idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    const int64_t e0 = a.e_start ? a.e_start[0] : 0;
    const uint64_t step = a.eps_ob ? 0 : (uint64_t)a.rng_step[0];
    NetRegs wn;
    net_issue<true>(wn, n, a.params);
    const HeadRegs hd = head_issue<true>(n, a.params);
    const bool norm = a.rms_sums != nullptr || a.mean != nullptr;
    // statistics the rows are normalised with (inline d.obs_rms.update: see disc1_body)
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
    net_commit<true>(wn, n);
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
    float* part = a.partials + (int64_t)wg * a.P;
    const Offs o = offs_ln(in, 1);
    // primal forward, first layer (all sections)
    layer_fwd(XC, n.k4, n.W1, n.b1, A1);
    __syncthreads();
    ln_fwd_phase(A1, N1, st1, n.be1, n.ga1);
    __syncthreads();
    layer_fwd(N1, H, n.W2, n.b2, A2);
    __syncthreads();
    if (sect == 2) {
        // ---- interpolate rows: g = d score / d input (T stream), this tile's share of sum g^2, then the
        // slot = sum_rows <V, dg/dtheta> WITHOUT the scalar k (finalize multiplies these slots by k)
        float* TN1 = T[6]; float* G = T[7]; float* ND1 = T[8]; float* PA1 = T[9]; float* PZ2 = T[10];
        float* PN1 = T[11];
        ln_fwd_phase(A2, nullptr, st2, n.be2, n.ga2);            // statistics only: n2 is never needed
        if (tid < R) dq[tid] = 1.f;
        __syncthreads();
        ln_bwd_phase(nullptr, dq, n.W3, A2, st2, n.ga2, nullptr, DZa, nullptr);   // T_z2
        __syncthreads();
        layer_bwd_lin(DZa, n.W2, TN1);                                             // T_n1
        __syncthreads();
        ln_bwd_phase(TN1, nullptr, nullptr, A1, st1, n.ga1, nullptr, DZb, nullptr);  // T_z1 (TN1 kept)
        __syncthreads();
        {
            const int lane = tid & 63, wv = tid >> 6;
            const int ti = (n.in + 15) / 16;
            for (int t = wv; t < ti; t += 4) {
                f32x4 acc = tile_nt(DZb, LD, n.W1, LD, H, 16 * t, lane);
                const int c = 16 * t + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) G[(r0 + j) * LD + c] = c < n.in ? acc[j] : 0.f;
            }
            // columns [16*ti, 64) of G are read by layer_tan's K loop only up to k4 <= 16*ti: leave them
        }
        __syncthreads();
        if (tid < R) {  // this tile's sums of squares, observation / action columns (fixed order)
            float so = 0.f, sa = 0.f;
            for (int c = 0; c < a.ob; ++c) so = fmaf(G[tid * LD + c], G[tid * LD + c], so);
            for (int c = a.ob; c < in; ++c) sa = fmaf(G[tid * LD + c], G[tid * LD + c], sa);
            r0_[tid] = so; r1_[tid] = sa;
        }
        __syncthreads();
        if (tid == 0) {
            double o_ = 0.0, c_ = 0.0;
            for (int r = 0; r < R; ++r) { o_ += (double)r0_[r]; c_ += (double)r1_[r]; }
            a.gp_part[2 * (wg - 2 * tiles_b)] = o_;
            a.gp_part[2 * (wg - 2 * tiles_b) + 1] = c_;
        }
        const float co = 1.f / ((float)a.B * (float)a.ob), ca = 1.f / ((float)a.B * (float)a.ac);
        const int gcols = (n.in + 15) / 16 * 16;
        for (int t = tid; t < R * 64; t += NT) {
            const int r = t >> 6, c = t & 63;
            if (c < gcols) G[r * LD + c] = c < in ? G[r * LD + c] * (c < a.ob ? co : ca) : 0.f;   // V
        }
        __syncthreads();
        // tangent forward along V, with the primal cotangents the two tangent LayerNorms send back
        layer_tan(G, n.k4, n.W1, A1, ND1);                       // adot1 = (V W1) .* m1
        __syncthreads();
        ln_tan_phase(ND1, A1, st1, n.ga1, TN1, nullptr, ND1, PA1, false, nullptr, PSg1);   // ndot1, PA1, sum tn1*u1
        __syncthreads();
        layer_tan(ND1, H, n.W2, A2, PZ2);                        // adot2 = (ndot1 W2) .* m2
        __syncthreads();
        ln_tan_phase(PZ2, A2, st2, n.ga2, nullptr, n.W3, nullptr, PZ2, true, PS3, PSg2);   // P_z2 in place
        __syncthreads();
        // P backward and the weight gradients (two products each)
        wgrad64x2(ND1, DZa, N1, PZ2, H, part + o.W2, part + o.b2);        // ndot1^T T_z2 + n1^T P_z2 ; sum P_z2
        layer_bwd_lin(PZ2, n.W2, PN1);                                     // P_n1
        if (tid < H) {
            part[o.W3 + tid] = (PS3[tid] + PS3[64 + tid]) + (PS3[128 + tid] + PS3[192 + tid]);       // sum ndot2
            part[o.be2 + tid] = 0.f;                                                                  // dbeta2
            part[o.be2 + H + tid] = (PSg2[tid] + PSg2[64 + tid]) + (PSg2[128 + tid] + PSg2[192 + tid]);  // dgamma2
        }
        if (tid == 0) part[o.b3] = 0.f;
        __syncthreads();
        ln_bwd_phase(PN1, nullptr, nullptr, A1, st1, n.ga1, PA1, PN1, PS);    // P_z1 in place; dbeta1, dgamma1 (P part)
        __syncthreads();
        ps_commit(PS, part + o.be1, PSg1);
        wgrad64x2(G, DZb, XC, PN1, in, part + o.W1, part + o.b1);           // V^T T_z1 + x^T P_z1 ; sum P_z1
        return;
    }
    // ---- policy / expert rows
    float* N2 = T[6];
    ln_fwd_phase(A2, N2, st2, n.be2, n.ga2, &n, s_);
    __syncthreads();
    if (tid < R) {
        const int64_t g = row0 + tid;
        const float s = s_[tid];
        float z;
        if (a.eps_ob) {  // injected tensors
            z = sect == 0 ? (a.fake ? a.fake[g] : 0.f) : (a.real ? a.real[g] : 1.f);
        } else {         // drawn here (discriminator.py:89-109)
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
    net1_bwd_ln(n, XC, in, A1, N1, A2, N2, st1, st2, dq, DZa, DZb, PS, part);
}

size_t disc_ln_lds_bytes(int in) {
    return ((size_t)net_lds_floats(in, 1, true) + (size_t)kDiscLnTiles * R * LD + kDiscLnStrip) * sizeof(float);
}

__global__ __launch_bounds__(NT) void step_kernel_ln(StepArgs s) {
    extern __shared__ float lds[];
    int b = blockIdx.x;
    if (b < s.n_a) { agent_a_body_ln(s.a, b, lds); return; }
    b -= s.n_a;
    if (b < s.n_tc) { agent_c_body_ln(s.a, b, lds); return; }
    b -= s.n_tc;
    disc1_body_ln(s.d, b, lds);
}

__global__ __launch_bounds__(NT) void targets_kernel_ln(AgentArgs a, int tiles) {
    extern __shared__ float lds[];
    const int b = blockIdx.x;
    agent_t_body_ln(a, b % tiles, lds, b / tiles);
}

struct TdLnArgs {
    const float *ret_mean, *ret_std;
    const float *obs0, *acs, *tc1s, *mean, *stdv, *critic;
    float* td_out;
    float* td_copy;
    int64_t* idx_copy;
    int ob, ac;
    const int64_t* idx;
    int64_t ring_start, ring_len;
};
// critic_ops['td_errors'] (sam_agent.py:534-538,693-701), LayerNorm critic: see td_kernel
__global__ __launch_bounds__(NT) void td_kernel_ln(TdLnArgs a) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * R;
    float* p = lds;
    NetL n = carve_net(p, a.ob + a.ac, 1, true);
    float* XC = p; p += R * LD;
    float* A1 = p; p += R * LD;
    float* A2 = p; p += R * LD;
    float* s_ = p; float* st = p + 16;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 4));
    NetRegs wn;
    net_issue<true>(wn, n, a.critic);
    const HeadRegs h = head_issue<true>(n, a.critic);
    RowRegs x;
    rows_issue(x, pr, a.obs0, a.acs, a.mean, a.stdv, a.ob, a.ac, a.ob + a.ac);
    net_commit<true>(wn, n);
    head_commit(n, h);
    rows_commit(XC, x, a.mean != nullptr, a.ob, a.ob + a.ac);
    stage_wait();
    net1_fwd_only_ln(n, XC, A1, A2, st, s_);
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

}  // namespace

constexpr size_t kLnLdsMax = 160 * 1024;   // a single workgroup may declare all of a CU's LDS

size_t ln_step_lds_bytes(int ob, int ac, bool agent, bool use_d, bool disc) {
    size_t b = 0;
    if (agent) {
        b = ln_lds_floats(ob, ac, use_d ? 7 : 3, 9) * sizeof(float);            // actor role
        const size_t c = ln_lds_floats(ob, ac, 2, 7) * sizeof(float);           // critic role
        if (c > b) b = c;
    }
    if (disc) { const size_t d = disc_ln_lds_bytes(ob + ac); if (d > b) b = d; }
    return b;
}

bool fits_lds_ln(int ob, int ac, bool use_d) {
    return ln_step_lds_bytes(ob, ac, true, use_d, true) <= kLnLdsMax &&
           ln_lds_floats(ob, ac, 3, 3) * sizeof(float) <= kLnLdsMax;
}

int launch_step_ln(const AgentArgs* a, const DiscArgs* d, hipStream_t st) {
    StepArgs s;
    s.n_a = s.n_tc = s.n_disc = 0;
    int ob = 0, ac = 0;
    if (a) {
        SAMX_REQUIRE(a->ext_tq, "launch_step_ln: the LayerNorm critic role needs stored targets (ext_tq)");
        s.a = *a;
        s.n_a = s.n_tc = (int)(a->B / R);
        ob = a->ob; ac = a->ac;
    }
    if (d) {
        s.d = *d;
        s.n_disc = 3 * (int)(d->B / R);
        ob = d->ob; ac = d->ac;
    }
    const size_t lds = ln_step_lds_bytes(ob, ac, a != nullptr, a && a->d_scale != 0.f, d != nullptr);
    SAMX_REQUIRE(lds <= kLnLdsMax, "fused LayerNorm step kernel needs %zu bytes of LDS", lds);
    static int roles = -1;   // profiling aid, see launch_step
    if (roles < 0) {
        const char* e = getenv("SAMX_FUSED_ROLES");
        roles = e ? atoi(e) : 7;
    }
    if (!(roles & 1)) s.n_a = 0;
    if (!(roles & 2)) s.n_tc = 0;
    if (!(roles & 4)) s.n_disc = 0;
    if (s.n_a + s.n_tc + s.n_disc == 0) return SAMX_OK;
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)step_kernel_ln, lds)) return rc;
    step_kernel_ln<<<(unsigned)(s.n_a + s.n_tc + s.n_disc), NT, lds, st>>>(s);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_targets_ln(const AgentArgs* a, hipStream_t st) {
    const int tiles = (int)(a->B / R);
    const size_t lds = ln_lds_floats(a->ob, a->ac, 3, 3) * sizeof(float);
    SAMX_REQUIRE(lds <= kLnLdsMax, "fused LayerNorm targets kernel needs %zu bytes of LDS", lds);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)targets_kernel_ln, lds)) return rc;
    targets_kernel_ln<<<(unsigned)(tiles * (a->n_idx ? 2 : 1)), NT, lds, st>>>(*a, tiles);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_td_ln(const samx_mlp* dc, const float* critic, const float* obs0, const float* acs, const float* tc1s,
                 const float* mean, const float* stdv, int64_t B, int ob, const int64_t* idx, int64_t ring_start,
                 int64_t ring_len, float* td_out, const float* ret_mean, const float* ret_std, float* td_copy,
                 int64_t* idx_copy, hipStream_t st) {
    TdLnArgs a;
    a.td_copy = td_copy; a.idx_copy = idx ? idx_copy : nullptr;
    a.ret_mean = ret_mean; a.ret_std = ret_std;
    a.obs0 = obs0; a.acs = acs; a.tc1s = tc1s; a.mean = mean; a.stdv = stdv; a.critic = critic;
    a.td_out = td_out; a.ob = ob; a.ac = dc->in_dim - ob; a.idx = idx; a.ring_start = ring_start;
    a.ring_len = ring_len;
    const size_t lds = ((size_t)net_lds_floats(dc->in_dim, 1, true) + 3 * R * LD + 64) * sizeof(float);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)td_kernel_ln, lds)) return rc;
    td_kernel_ln<<<(unsigned)(B / R), NT, lds, st>>>(a);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // namespace fused
}  // namespace samx
