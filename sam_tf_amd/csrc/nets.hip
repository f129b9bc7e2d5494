// Layer-wise network passes for the actor / critic / reward net: row-local kernels (input prep,
// heads, losses) around the MFMA GEMM of gemm.hip.  This is the general path (any width / batch);
// small networks additionally have whole-network fused kernels (fused_small.hip).
//   sam_agent.py:145-205,406-520 ; discriminator.py:61-194 ; networks.py:45-72,222-404
#include "samx_common.h"
#include "gemm.h"
#include "nets.h"
#include "fused_wide.h"
#include "fused_small.h"
#include "rng.h"
#include "../../include/samx.h"
#include <math.h>

namespace samx {
namespace {

struct Net {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    const float *beta1, *gamma1, *beta2, *gamma2;  // LayerNorm (networks.py:53-70), null without it
    int in, h1, h2, out, ln;
    int64_t oW1, ob1, oW2, ob2, oW3, ob3, obeta1, ogamma1, obeta2, ogamma2;  // offsets in the flat vector
};

Net split(const samx_mlp* d, const float* p) {
    Net n;
    n.in = d->in_dim; n.h1 = d->h1; n.h2 = d->h2; n.out = d->out_dim; n.ln = d->layernorm ? 1 : 0;
    int64_t o = 0;
    n.oW1 = o; o += (int64_t)n.in * n.h1;
    n.ob1 = o; o += n.h1;
    n.obeta1 = n.ogamma1 = n.obeta2 = n.ogamma2 = 0;
    n.beta1 = n.gamma1 = n.beta2 = n.gamma2 = nullptr;
    if (n.ln) { n.obeta1 = o; o += n.h1; n.ogamma1 = o; o += n.h1; }
    n.oW2 = o; o += (int64_t)n.h1 * n.h2;
    n.ob2 = o; o += n.h2;
    if (n.ln) { n.obeta2 = o; o += n.h2; n.ogamma2 = o; o += n.h2; }
    n.oW3 = o; o += (int64_t)n.h2 * n.out;
    n.ob3 = o;
    n.W1 = p + n.oW1; n.b1 = p + n.ob1; n.W2 = p + n.oW2; n.b2 = p + n.ob2;
    n.W3 = p + n.oW3; n.b3 = p + n.ob3;
    if (n.ln) {
        n.beta1 = p + n.obeta1; n.gamma1 = p + n.ogamma1;
        n.beta2 = p + n.obeta2; n.gamma2 = p + n.ogamma2;
    }
    return n;
}

int check_desc(const samx_mlp* d, const char* who) {
    SAMX_REQUIRE(d, "%s: null network description", who);
    SAMX_REQUIRE(d->in_dim > 0 && d->h1 > 0 && d->h2 > 0 && d->out_dim > 0,
                 "%s: non-positive layer size", who);
    SAMX_REQUIRE(d->h1 <= 4096 && d->h2 <= 4096 && d->in_dim <= 65536, "%s: layer too wide", who);
    return SAMX_OK;
}

// ---------------------------------------------------------------- row-local kernels

// f(row, column) for the elements t = row * w + column < tot of a row-major matrix, grid-stride over t.
// The split of t is a 32-bit division whenever tot fits (a 64-bit division per element cost these row-local
// kernels more than their memory traffic).
template <typename F>
__device__ __forceinline__ void for_rows_cols(int64_t tot, int w, F f) {
    if (tot <= 0x7fffffff) {
        const uint32_t n = (uint32_t)tot, uw = (uint32_t)w, stride = gridDim.x * blockDim.x;
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
            const uint32_t row = t / uw;
            f((int64_t)row, (int)(t - row * uw));   // (t + stride stays below 2^32: both are below 2^31)
        }
    } else {
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x)
            f(t / w, (int)(t % w));
    }
}

// out[row, 0:ob] = clip((obs - mean)/std, -5, 5)  (sam_agent.py:150-156, math_util.py:8-12) ;
// out[row, ob:ob+ac] = acs[row]  (critic / reward-net input concat, networks.py:238,363)
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ obs,
                                                   const float* __restrict__ acs,
                                                   const float* __restrict__ mean,
                                                   const float* __restrict__ stdv, int64_t B, int ob,
                                                   int ac, float* __restrict__ out, int ld) {
    const int w = ob + (acs ? ac : 0);
    const int64_t tot = B * w;
    auto body = [&](int64_t row, int c) {
        float v;
        if (c < ob) {
            v = obs[row * ob + c];
            if (mean) v = fminf(fmaxf((v - mean[c]) / stdv[c], -5.f), 5.f);
        } else {
            v = acs[row * ac + (c - ob)];
        }
        out[row * ld + c] = v;
    };
    for_rows_cols(tot, w, body);
}

// rows [0,B): policy pair, [B,2B): expert pair, [2B,3B): eps*policy + (1-eps)*expert on the
// NORMALISED observations and raw actions (discriminator.py:71-72,129-134)
// The rows may still live in the replay storage (src.p_idx) / the expert pool (src.e_start) and the randoms
// may be drawn here (src.draw; same Philox streams and indices as samx_disc_randoms): no gather, slice or
// random-number launch in front of the discriminator's chain.
__global__ __launch_bounds__(256) void disc_prep_kernel(
    const float* __restrict__ p_obs, const float* __restrict__ p_acs, const float* __restrict__ e_obs,
    const float* __restrict__ e_acs, const float* __restrict__ eps_ob,
    const float* __restrict__ eps_ac, const float* __restrict__ mean, const float* __restrict__ stdv,
    int64_t B, int ob, int ac, float* __restrict__ X, DiscSrc src) {
    const int w = ob + ac;
    const int64_t tot = B * w;
    const int64_t e0 = src.e_start ? src.e_start[0] : 0;
    auto body = [&](int64_t row, int c) {   // (injected weights; drawn ones: disc_prep_draw_kernel)
        int64_t pr = row;
        if (src.p_idx) {
            pr = src.ring_start + src.p_idx[row];
            if (pr >= src.ring_len) pr -= src.ring_len;
        }
        const int64_t er = e0 + row;
        float p, e, eps;
        if (c < ob) {
            p = p_obs[pr * ob + c];
            e = e_obs[er * ob + c];
            if (mean) {
                p = fminf(fmaxf((p - mean[c]) / stdv[c], -5.f), 5.f);
                e = fminf(fmaxf((e - mean[c]) / stdv[c], -5.f), 5.f);
            }
            eps = eps_ob[row * ob + c];
        } else {
            p = p_acs[pr * ac + (c - ob)];
            e = e_acs[er * ac + (c - ob)];
            eps = eps_ac[row * ac + (c - ob)];
        }
        X[row * w + c] = p;
        X[(B + row) * w + c] = e;
        X[(2 * B + row) * w + c] = eps * p + (1.f - eps) * e;
    };
    for_rows_cols(tot, w, body);
}

// The same with the interpolation weights and the labels drawn here (src.draw): a work item is FOUR consecutive
// elements of one of the two weight tensors — one Philox evaluation (rng.h: counter = element / 4) — or one row's
// labels.  (One Philox per element was 9.6 M evaluations per step at C5.)
__global__ __launch_bounds__(256) void disc_prep_draw_kernel(
    const float* __restrict__ p_obs, const float* __restrict__ p_acs, const float* __restrict__ e_obs,
    const float* __restrict__ e_acs, const float* __restrict__ mean, const float* __restrict__ stdv, int64_t B, int ob,
    int ac, float* __restrict__ X, DiscSrc src) {
    const int w = ob + ac;
    const uint64_t st = (uint64_t)src.step_dev[0];
    const int64_t e0 = src.e_start ? src.e_start[0] : 0;
    const int64_t n_ob = B * ob, n_ac = B * ac, g_ob = (n_ob + 3) >> 2, g_ac = (n_ac + 3) >> 2;
    const int64_t nw = g_ob + g_ac + B;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nw; t += (int64_t)gridDim.x * blockDim.x) {
        if (t >= g_ob + g_ac) {
            const int64_t row = t - g_ob - g_ac;
            if (src.real_out) src.real_out[row] = real_label(philox_uniform(src.rng_seed, st, RNG_REAL, (uint64_t)row));
            if (src.fake_out) src.fake_out[row] = fake_label(philox_uniform(src.rng_seed, st, RNG_FAKE, (uint64_t)row));
            continue;
        }
        const bool isob = t < g_ob;
        const int64_t grp = isob ? t : t - g_ob, n = isob ? n_ob : n_ac;
        const int cw = isob ? ob : ac;
        float u[4];
        philox_uniform4(src.rng_seed, st, isob ? RNG_EPS_OB : RNG_EPS_AC, (uint64_t)grp, u);
        // (row, column) of the group's first element by ONE division — 32-bit when the tensor allows it — and stepped
        // from there (four 64-bit divisions per work item used to be most of this kernel's instructions)
        const int64_t el0 = 4 * grp;
        int64_t row = n <= 0x7fffffff ? (int64_t)((uint32_t)el0 / (uint32_t)cw) : el0 / cw;
        int cc = (int)(el0 - row * cw);
        int64_t pr = -1, pr_row = -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t el = el0 + i;
            if (el >= n) break;
            if (i > 0 && ++cc == cw) { cc = 0; ++row; }
            if (pr_row != row) {
                pr_row = row;
                pr = row;
                if (src.p_idx) {
                    pr = src.ring_start + src.p_idx[row];
                    if (pr >= src.ring_len) pr -= src.ring_len;
                }
            }
            const int64_t er = e0 + row;
            float pv, ev;
            if (isob) {
                pv = p_obs[pr * ob + cc];
                ev = e_obs[er * ob + cc];
                if (mean) {
                    pv = fminf(fmaxf((pv - mean[cc]) / stdv[cc], -5.f), 5.f);
                    ev = fminf(fmaxf((ev - mean[cc]) / stdv[cc], -5.f), 5.f);
                }
            } else {
                pv = p_acs[pr * ac + cc];
                ev = e_acs[er * ac + cc];
            }
            const int c = isob ? cc : ob + cc;
            X[row * w + c] = pv;
            X[(B + row) * w + c] = ev;
            X[(2 * B + row) * w + c] = u[i] * pv + (1.f - u[i]) * ev;
        }
    }
}

enum HeadMode { HEAD_Q, HEAD_TARG, HEAD_TD, HEAD_CRITIC, HEAD_ACTOR_Q, HEAD_DISC, HEAD_REWARD };

struct HeadArgs {
    const float* H2;   // [M, h2] last hidden activation
    const float* w3;   // [h2] (out_dim == 1)
    const float* b3;   // [1]
    int64_t M;
    int h2;
    // mode-specific
    const float* r0 = nullptr;   // rews | tc1s | fake labels
    const float* r1 = nullptr;   // dones1 | iws | real labels
    const float* r2 = nullptr;   // td_len
    float c0 = 0.f, c1 = 0.f;    // gamma | coefficient | ent scale
    int64_t B = 0;               // rows per section (HEAD_DISC)
    int flag = 0;                // non_satur_grad
    float* out0 = nullptr;       // q | targ | td | rowloss | reward | ds
    float* out1 = nullptr;       // td | ce
    float* out2 = nullptr;       // dq | ent
    float* out3 = nullptr;       // sigmoid
    float* DZ2 = nullptr;        // [M, h2]: dq * w3 * lrelu'(H2), or (ln) the cotangent of the LayerNorm output dq * w3
    int ln = 0;
    // return normalisation (sam_agent.py:161-201): device scalars, null = off.  The critic output is
    // DE-normalised (s * std + mean), targets are normalised ((t - mean) / std).
    const float* ret_mean = nullptr;
    const float* ret_std = nullptr;
    // n-step term of the critic loss (sam_agent.py:483-498)
    const float* rn = nullptr;   // tcns
    float cn = 0.f;              // td_loss_n_scale / B
    float* out4 = nullptr;       // row loss of the n-step term
    float* out5 = nullptr;       // n-step td error
};

__device__ __forceinline__ float wave_sum(float x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x z + log1p(exp(-|x|))
__device__ __forceinline__ float sig_ce(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

// one wave per row: s = <H2[row], w3> + b3, then the mode's row-local tail
template <int MODE>
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const float* h = a.H2 + row * a.h2;
    float acc = 0.f;
    for (int c = lane; c < a.h2; c += 64) acc = fmaf(h[c], a.w3[c], acc);
    float s = wave_sum(acc) + a.b3[0];
    float dq = 0.f;
    const bool rn_on = a.ret_mean != nullptr;
    const float rmean = rn_on ? a.ret_mean[0] : 0.f, rstd = rn_on ? a.ret_std[0] : 1.f;
    if (MODE == HEAD_Q || MODE == HEAD_TARG || MODE == HEAD_TD || MODE == HEAD_CRITIC || MODE == HEAD_ACTOR_Q) {
        if (rn_on) s = s * rstd + rmean;  // dermsify (math_util.py:15-19)
    }
    if (MODE == HEAD_Q) {
        if (lane == 0) a.out0[row] = s;
        return;
    }
    if (MODE == HEAD_TARG) {
        if (lane == 0) {
            // rews + gamma^td_len * (1 - dones1) * q'   (sam_agent.py:202-205)
            const float g = a.r2 ? powf(a.c0, a.r2[row]) : a.c0;
            a.out0[row] = a.r0[row] + g * ((1.f - a.r1[row]) * s);
        }
        return;
    }
    if (MODE == HEAD_TD) {
        if (lane == 0) a.out0[row] = s - (rn_on ? (a.r0[row] - rmean) / rstd : a.r0[row]);
        return;
    }
    if (MODE == HEAD_REWARD) {
        if (lane == 0) {
            // discriminator.py:180-190
            a.out0[row] = a.flag ? (fminf(s, 0.f) - log1pf(expf(-fabsf(s))))
                                 : -logf(1.f - sigmoidf_(s) + 1e-8f);
        }
        return;
    }
    if (MODE == HEAD_CRITIC) {
        const float td = s - (rn_on ? (a.r0[row] - rmean) / rstd : a.r0[row]);
        const float iw = a.r1 ? a.r1[row] : 1.f;
        const float ad = fabsf(td);
        const float hub = ad < 1.f ? 0.5f * td * td : ad - 0.5f;      // math_util.py:22-29
        const float dh = ad < 1.f ? td : (td > 0.f ? 1.f : -1.f);
        dq = a.c0 * dh * iw;                                          // c0 = td_scale / B
        float hubn = 0.f, tdn = 0.f;
        if (a.rn) {  // n-step term on the same prediction
            tdn = s - (rn_on ? (a.rn[row] - rmean) / rstd : a.rn[row]);
            const float adn = fabsf(tdn);
            hubn = adn < 1.f ? 0.5f * tdn * tdn : adn - 0.5f;
            dq += a.cn * (adn < 1.f ? tdn : (tdn > 0.f ? 1.f : -1.f)) * iw;
        }
        dq *= rstd;  // d(dermsified prediction) / d(network output)
        if (lane == 0) {
            a.out0[row] = hub * iw;
            if (a.out1) a.out1[row] = td;
            a.out2[row] = dq;
            if (a.out4) a.out4[row] = hubn * iw;
            if (a.out5) a.out5[row] = tdn;
        }
    }
    if (MODE == HEAD_ACTOR_Q) {
        dq = a.c0 * rstd;  // -scale / B, through the dermsify
        if (lane == 0) a.out0[row] = s;
    }
    if (MODE == HEAD_DISC) {
        if (row < 2 * a.B) {
            const bool pol = row < a.B;
            const float z = pol ? (a.r0 ? a.r0[row] : 0.f) : (a.r1 ? a.r1[row - a.B] : 1.f);
            const float sg = sigmoidf_(s);
            // d/ds [CE(s,z)/B - ent_scale * CE(s, sigmoid(s)) / 2B]
            dq = (sg - z) / (float)a.B + a.c1 * s * sg * (1.f - sg) / (float)(2 * a.B);
            if (lane == 0) {
                a.out1[row] = sig_ce(s, z);
                a.out2[row] = sig_ce(s, sg);
                a.out3[row] = sg;
            }
        } else {
            // interpolate rows: seed of d(sum scores)/d(inputs) (discriminator.py:136).  flag: the
            // LayerNorm layout, where rows [2B,3B) are the primal copy and take no seed.
            dq = (a.flag && row < 3 * a.B) ? 0.f : 1.f;
        }
        if (lane == 0) a.out0[row] = dq;
    }
    // dZ2 = dq * w3 * lrelu'(A2)
    float* dz = a.DZ2 + row * a.h2;
    if (a.ln) {
        for (int c = lane; c < a.h2; c += 64) dz[c] = dq * a.w3[c];
    } else {
        for (int c = lane; c < a.h2; c += 64) dz[c] = dq * a.w3[c] * (h[c] >= 0.f ? 1.f : 0.2f);
    }
}

// ---------------------------------------------------------------- LayerNorm row kernels
// Post-activation LayerNorm over the last axis, biased variance, eps 1e-8 (network_util.py:32-39,
// networks.py:53-70).  One wave per row.
constexpr float kLnEps = 1e-8f;

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ A, const float* __restrict__ beta,
                                                     const float* __restrict__ gamma, int64_t M, int H,
                                                     float* __restrict__ XH, float* __restrict__ N,
                                                     float* __restrict__ R) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* a = A + row * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += a[c];
    const float mean = wave_sum(s) / (float)H;
    float v = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = a[c] - mean; v = fmaf(d, d, v); }
    const float rs = 1.f / sqrtf(wave_sum(v) / (float)H + kLnEps);
    for (int c = lane; c < H; c += 64) {
        const float xh = (a[c] - mean) * rs;
        XH[row * H + c] = xh;
        N[row * H + c] = fmaf(gamma[c], xh, beta[c]);
    }
    if (lane == 0) R[row] = rs;
}

// cotangent of the LayerNorm output -> cotangent of the pre-activation:
//   dxh = dn * gamma ; da = rs (dxh - mean(dxh) - xh mean(dxh xh)) [+ add] ; dz = da * lrelu'(a)
// TA (optional) receives da before `add` and before the mask.
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ DN, const float* __restrict__ A,
                                                     const float* __restrict__ XH, const float* __restrict__ R,
                                                     const float* __restrict__ gamma, int64_t M, int H,
                                                     const float* add, float* DZ, float* __restrict__ TA) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t o = row * H;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float d = DN[o + c] * gamma[c];
        s1 += d;
        s2 = fmaf(d, XH[o + c], s2);
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H, rs = R[row];
    for (int c = lane; c < H; c += 64) {
        const float xh = XH[o + c];
        float da = rs * (DN[o + c] * gamma[c] - m1 - xh * m2);
        if (TA) TA[o + c] = da;
        if (add) da += add[o + c];
        DZ[o + c] = da * (A[o + c] >= 0.f ? 1.f : 0.2f);
    }
}

// Tangent of a LayerNorm layer along a direction, and the cotangent that the tangent's dependence
// on the primal activation sends back (second-order term of the gradient penalty; derivation in
// DESIGN.md 3.5).  Row r of the tangent stream, primal quantities from the same row of the primal copy.
//   ad = ZD * lrelu'(a) ; d = ad - mean(ad) ; c = mean(xh ad) ; u = rs (d - c xh) ; nd = gamma u
//   ub = gamma TN ; Aq = <ub, d> ; Bq = <ub, xh>
//   PA = -(rs^2 (Aq - c Bq)/H) xh - (rs Bq/H) u - rs c TA      (times lrelu'(a) when mask_out)
__global__ __launch_bounds__(256) void ln_tan_kernel(const float* __restrict__ ZD, const float* __restrict__ A,
                                                     const float* __restrict__ XH, const float* __restrict__ R,
                                                     const float* __restrict__ gamma, const float* __restrict__ TN,
                                                     const float* __restrict__ TA, int64_t M, int H,
                                                     float* __restrict__ U, float* __restrict__ ND,
                                                     float* __restrict__ PA, int mask_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t o = row * H;
    float s1 = 0.f, s2 = 0.f, sa = 0.f, su = 0.f, sb = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float ad = ZD[o + c] * (A[o + c] >= 0.f ? 1.f : 0.2f);
        const float xh = XH[o + c], ub = gamma[c] * TN[o + c];
        s1 += ad;
        s2 = fmaf(xh, ad, s2);
        sa = fmaf(ub, ad, sa);
        su += ub;
        sb = fmaf(ub, xh, sb);
    }
    const float fH = (float)H;
    const float mean_ad = wave_sum(s1) / fH, cc = wave_sum(s2) / fH;
    const float Aq = wave_sum(sa) - mean_ad * wave_sum(su), Bq = wave_sum(sb), rs = R[row];
    const float k1 = -(rs * rs * (Aq - cc * Bq) / fH), k2 = -(rs * Bq / fH), k3 = -(rs * cc);
    for (int c = lane; c < H; c += 64) {
        const float ad = ZD[o + c] * (A[o + c] >= 0.f ? 1.f : 0.2f);
        const float xh = XH[o + c];
        const float u = rs * ((ad - mean_ad) - cc * xh);
        U[o + c] = u;
        ND[o + c] = gamma[c] * u;
        const float pa = k1 * xh + k2 * u + k3 * TA[o + c];
        PA[o + c] = mask_out ? pa * (A[o + c] >= 0.f ? 1.f : 0.2f) : pa;
    }
}

// LayerNorm parameter gradients: dbeta[j] = sum_{r < Mb} DN[r,j], dgamma[j] = sum_{r < Mg} DN[r,j] XH[r,j].
// Stage 1: block (x = 64 columns, y = row chunk) -> part[y][2][H]: four row-lanes per column, each adds its rows
// in increasing order with eight rows' loads in flight (one dependent load pair per row made this the longest
// small kernel of the LayerNorm configurations: 76 us at 8192 x 400).  Stage 2 adds the chunks: four chunk-lanes
// per column, each in increasing chunk order, then a fixed tree.
constexpr int kLnChunksMax = 128;
__global__ __launch_bounds__(256) void ln_pgrad1_kernel(const float* __restrict__ DN, const float* __restrict__ XH,
                                                        int64_t Mb, int64_t Mg, int H, int64_t chunk,
                                                        float* __restrict__ part) {
    __shared__ float sh[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, col = blockIdx.x * 64 + cl;
    const int64_t r0 = blockIdx.y * chunk, r1 = min(r0 + chunk, max(Mb, Mg));
    float b = 0.f, g = 0.f;
    if (col < H) {
        constexpr int U = 8;
        int64_t r = r0 + rl;
        for (; r + 4 * (U - 1) < r1; r += 4 * U) {
            float dn[U], xh[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                dn[u] = DN[(r + 4 * u) * H + col];
                xh[u] = XH[(r + 4 * u) * H + col];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + 4 * u < Mb) b += dn[u];
                if (r + 4 * u < Mg) g = fmaf(dn[u], xh[u], g);
            }
        }
        for (; r < r1; r += 4) {
            const float dn = DN[r * H + col];
            if (r < Mb) b += dn;
            if (r < Mg) g = fmaf(dn, XH[r * H + col], g);
        }
    }
    sh[0][rl][cl] = b;
    sh[1][rl][cl] = g;
    __syncthreads();
    if (rl == 0 && col < H) {
        part[((int64_t)blockIdx.y * 2) * H + col] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
        part[((int64_t)blockIdx.y * 2 + 1) * H + col] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
    }
}
__global__ __launch_bounds__(256) void ln_pgrad2_kernel(const float* __restrict__ part, int S, int H,
                                                        float* __restrict__ dbeta, float* __restrict__ dgamma) {
    __shared__ float sh[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6, col = blockIdx.x * 64 + cl;
    const int per = (S + 3) / 4, s0 = sl * per, s1 = min(S, s0 + per);
    float b = 0.f, g = 0.f;
    if (col < H) {
        constexpr int U = 8;
        int s = s0;
        for (; s + U <= s1; s += U) {
            float pb[U], pg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                pb[u] = part[((int64_t)(s + u) * 2) * H + col];
                pg[u] = part[((int64_t)(s + u) * 2 + 1) * H + col];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { b += pb[u]; g += pg[u]; }
        }
        for (; s < s1; ++s) {
            b += part[((int64_t)s * 2) * H + col];
            g += part[((int64_t)s * 2 + 1) * H + col];
        }
    }
    sh[0][sl][cl] = b;
    sh[1][sl][cl] = g;
    __syncthreads();
    if (sl == 0 && col < H) {
        dbeta[col] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
        dgamma[col] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
    }
}

template <int MODE>
int launch_head(const HeadArgs& a, hipStream_t st) {
    head_kernel<MODE><<<(unsigned)((a.M + 3) / 4), 256, 0, st>>>(a);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

__device__ __forceinline__ double block_sum_d(double x, double* sh) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    __syncthreads();  // protect sh reuse across consecutive calls
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;  // same value on every thread
}

__device__ double sum_f32(const float* x, int64_t n, double* sh) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += (double)x[i];
    return block_sum_d(a, sh);
}
__device__ double sumsq_f32(const float* x, int64_t n, double* sh) {
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += (double)x[i] * (double)x[i];
    return block_sum_d(a, sh);
}

// sum of squares of the two decayed kernels (network_util.py:43-56), spread over kWdBlocks workgroups;
// block b writes its float64 partial, critic_losses_kernel adds them in block order
constexpr int kWdBlocks = 64;
__global__ __launch_bounds__(256) void wd_sumsq_kernel(const float* __restrict__ W1, int64_t n1,
                                                       const float* __restrict__ W2, int64_t n2,
                                                       double* __restrict__ part) {
    __shared__ double sh[4];
    double a = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * 256) {
        const double w = (double)(i < n1 ? W1[i] : W2[i - n1]);
        a += w * w;
    }
    a = block_sum_d(a, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// {critic_td_loss_1, critic_wd, critic_total_loss}  (sam_agent.py:473-517)
__global__ __launch_bounds__(1024) void critic_losses_kernel(const float* rowloss, int64_t B,
                                                             float td_scale, float wd_scale,
                                                             const double* wd_part, int n_part,
                                                             float* losses, const float* rowloss_n,
                                                             float tdn_scale) {
    __shared__ double sh[16];
    const double l = sum_f32(rowloss, B, sh) / (double)B * td_scale;
    const double ln_ = rowloss_n ? sum_f32(rowloss_n, B, sh) / (double)B * tdn_scale : 0.0;
    if (threadIdx.x == 0) {
        double wd = 0.0;
        if (wd_scale > 0.f) {
            for (int i = 0; i < n_part; ++i) wd += wd_part[i];
            wd *= 0.5 * wd_scale;
        }
        losses[0] = (float)l;
        losses[1] = (float)wd;
        losses[2] = (float)(l + ln_ + wd);
        losses[3] = (float)ln_;
    }
}

// PopArt (sam_agent.py:554-590,662-672) with the return-statistics update that precedes it:
//   old = (mean, std) ; sums += addvec1 (+ addvec2) (mpi_running_mean_std.py:40-45) ; new = stats(sums) ;
//   for the online and the target critic:  w3 <- w3 * old_std / new_std ;
//                                          b3 <- (b3 * old_std + old_mean - new_mean) / new_std
// One workgroup.  addvec: float64 [sum, sumsq, count] of a batch of targets (after the all-reduce).
__global__ __launch_bounds__(256) void popart_kernel(double* sums, const double* addvec1, const double* addvec2,
                                                     float* mean, float* stdv, int popart, float* w3a,
                                                     float* b3a, float* w3b, float* b3b, int h2) {
    __shared__ float sh[4];
    if (threadIdx.x == 0) {
        sh[0] = mean[0];
        sh[1] = stdv[0];
        double s0 = sums[0], s1 = sums[1], s2 = sums[2];
        s0 = __dadd_rn(s0, addvec1[0]); s1 = __dadd_rn(s1, addvec1[1]); s2 = __dadd_rn(s2, addvec1[2]);
        if (addvec2) {
            s0 = __dadd_rn(s0, addvec2[0]); s1 = __dadd_rn(s1, addvec2[1]); s2 = __dadd_rn(s2, addvec2[2]);
        }
        sums[0] = s0; sums[1] = s1; sums[2] = s2;
        const float m = (float)(s0 / s2), msq = (float)(s1 / s2);
        const float sd = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(m, m)), 1e-2f));
        mean[0] = m;
        stdv[0] = sd;
        sh[2] = m;
        sh[3] = sd;
    }
    __syncthreads();
    if (!popart) return;
    const float om = sh[0], os = sh[1], nm = sh[2], ns = sh[3];
    for (int c = threadIdx.x; c < h2; c += 256) {
        w3a[c] = __fdiv_rn(__fmul_rn(w3a[c], os), ns);
        w3b[c] = __fdiv_rn(__fmul_rn(w3b[c], os), ns);
    }
    if (threadIdx.x == 0) {
        b3a[0] = __fdiv_rn(__fsub_rn(__fadd_rn(__fmul_rn(b3a[0], os), om), nm), ns);
        b3b[0] = __fdiv_rn(__fsub_rn(__fadd_rn(__fmul_rn(b3b[0], os), om), nm), ns);
    }
}

// {actor_q_loss, actor_d_loss, actor_total_loss}  (sam_agent.py:412-430)
__global__ __launch_bounds__(1024) void actor_losses_kernel(const float* rowq, const float* rowd,
                                                            int64_t B, float q_scale, float d_scale,
                                                            float* losses) {
    __shared__ double sh[16];
    const double q = -sum_f32(rowq, B, sh) / (double)B * q_scale;
    const double d = rowd ? -sum_f32(rowd, B, sh) / (double)B * d_scale : 0.0;
    if (threadIdx.x == 0) {
        losses[0] = (float)q;
        losses[1] = (float)d;
        losses[2] = (float)(q + d);
    }
}

// gradient-penalty reduction (discriminator.py:138-140): per-block partial sums of g^2 split into
// observation and action columns
constexpr int kGpBlocks = 128;
__global__ __launch_bounds__(256) void gp_sumsq_kernel(const float* __restrict__ G, int64_t B, int ob,
                                                       int ac, double* __restrict__ partials) {
    __shared__ double sh[4];
    const int w = ob + ac;
    const int64_t tot = B * w;
    double so = 0.0, sa = 0.0;
    // eight loads in flight per thread, added in the same increasing order of t; the column of an element
    // advances by (stride mod w) per step (no 64-bit division per element)
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int cstep = (int)(stride % w);
    int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    int c = (int)(t % w);
    for (; t + (U - 1) * stride < tot; t += U * stride) {
        float g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) g[u] = G[t + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double d = (double)g[u];
            if (c < ob) so += d * d; else sa += d * d;
            c += cstep;
            c = c >= w ? c - w : c;
        }
    }
    for (; t < tot; t += stride) {
        const double d = (double)G[t];
        if (c < ob) so += d * d; else sa += d * d;
        c += cstep;
        c = c >= w ? c - w : c;
    }
    so = block_sum_d(so, sh);
    sa = block_sum_d(sa, sh);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = so;
        partials[2 * blockIdx.x + 1] = sa;
    }
}

// V = dGP_total/dg = 10 * 2 (N-1)/N * g / (B*dim_part), N = sqrt(mean g_ob^2 + mean g_ac^2)
__global__ __launch_bounds__(256) void gp_scale_kernel(const float* __restrict__ G, int64_t B, int ob,
                                                       int ac, const double* __restrict__ partials,
                                                       int np, float coeff, float* __restrict__ V,
                                                       float* __restrict__ gp_out) {
    __shared__ double sh[4];
    double so = 0.0, sa = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) {
        so += partials[2 * i];
        sa += partials[2 * i + 1];
    }
    so = block_sum_d(so, sh);
    sa = block_sum_d(sa, sh);
    const double mo = so / ((double)B * ob), ma = sa / ((double)B * ac);
    const double N = sqrt(mo + ma);
    const double k = (double)coeff * 2.0 * (N - 1.0) / N;
    const float co = (float)(k / ((double)B * ob)), ca = (float)(k / ((double)B * ac));
    if (blockIdx.x == 0 && threadIdx.x == 0) gp_out[0] = (float)((N - 1.0) * (N - 1.0));
    const int w = ob + ac;
    const int64_t tot = B * w;
    for_rows_cols(tot, w, [&](int64_t row, int c) { V[row * w + c] = G[row * w + c] * (c < ob ? co : ca); });
}

// the 8 discriminator losses in the order of discriminator.py:145-158
__global__ __launch_bounds__(1024) void disc_losses_kernel(const float* ce, const float* ent,
                                                           const float* sg, const float* gp,
                                                           int64_t B, float ent_scale, float coeff,
                                                           float* losses) {
    __shared__ double sh[16];
    const double pl = sum_f32(ce, B, sh) / (double)B;
    const double el = sum_f32(ce + B, B, sh) / (double)B;
    const double em = sum_f32(ent, 2 * B, sh) / (double)(2 * B);
    const double pa = sum_f32(sg, B, sh) / (double)B;
    const double ea = sum_f32(sg + B, B, sh) / (double)B;
    if (threadIdx.x == 0) {
        const double eloss = -(double)ent_scale * em;
        losses[0] = (float)pl;
        losses[1] = (float)el;
        losses[2] = (float)em;
        losses[3] = (float)eloss;
        losses[4] = (float)pa;
        losses[5] = (float)ea;
        losses[6] = gp[0];
        losses[7] = (float)(pl + el + eloss + (double)coeff * (double)gp[0]);
    }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int ld, int64_t B, int n,
                                 float* __restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < B * n) dst[t] = src[(t / n) * ld + (t % n)];
}

// ---------------------------------------------------------------- GEMM wrappers

// Y[M,N] = epi(X[M,K] W[K,N] + b)
int fwd(const float* X, int64_t ldx, int64_t M, int K, const float* W, const float* b, int N, float* Y,
        int64_t ldy, Epilogue epi, float alpha, const float* aux, int64_t ld_aux, hipStream_t st) {
    GemmArgs g;
    g.A = X; g.a_si = ldx; g.a_sr = 1;
    g.B = W; g.b_sr = N; g.b_sj = 1;
    g.C = Y; g.ldc = ldy;
    g.M = (int)M; g.N = N; g.R = K;
    g.epi = epi; g.bias = b; g.alpha = alpha; g.aux = aux; g.ld_aux = ld_aux;
    g.overread_ok = true;   // X: padded workspace carve; W: followed by its bias in the flat vector
    return gemm(g, st);
}

// dX[M,K] = epi(dZ[M,N] W[K,N]^T)   (W row stride ldw: pass a row-offset W to take a column band)
int bwd_data(const float* dZ, int64_t ldz, int64_t M, int N, const float* W, int64_t ldw, int K,
             float* dX, int64_t ldx, Epilogue epi, const float* aux, int64_t ld_aux, const float* add,
             int64_t ld_add, float alpha, hipStream_t st) {
    GemmArgs g;
    g.A = dZ; g.a_si = ldz; g.a_sr = 1;
    g.B = W; g.b_sr = 1; g.b_sj = ldw;
    g.C = dX; g.ldc = ldx;
    g.M = (int)M; g.N = K; g.R = N;
    g.epi = epi; g.aux = aux; g.ld_aux = ld_aux; g.add = add; g.ld_add = ld_add; g.alpha = alpha;
    g.overread_ok = true;
    return gemm(g, st);
}

// [dW; db] ([K+1, N], contiguous in the flat gradient) = [X, 1]^T dZ (+ wd * W), reduction over rows;
// the ones column is 1 for rows < ones_rows
int wgrad(const float* X, int64_t ldx, int K, const float* dZ, int64_t ldz, int N, int64_t rows,
          int64_t ones_rows, float* dW, float wd, const float* W, float* partials,
          size_t partial_floats, hipStream_t st) {
    GemmArgs g;
    g.A = X; g.a_si = 1; g.a_sr = ldx;
    g.B = dZ; g.b_sr = ldz; g.b_sj = 1;
    g.C = dW; g.ldc = N;
    g.M = K; g.N = N; g.R = (int)rows;
    g.a_ones_rows = (int)ones_rows;
    if (wd > 0.f) { g.epi = EPI_AXPY; g.alpha = wd; g.aux = W; g.ld_aux = N; }
    g.nsplit = pick_nsplit(K + 1, N, (int)rows);
    if ((size_t)g.nsplit * (K + 1) * N > partial_floats) g.nsplit = 1;
    g.partials = partials;
    g.overread_ok = true;
    return gemm(g, st);
}

size_t wgrad_partial_floats(const samx_mlp* d, int64_t rows) {
    size_t m = 0;
    const int dims[3][2] = {{d->in_dim + 1, d->h1}, {d->h1 + 1, d->h2}, {d->h2 + 1, d->out_dim}};
    for (auto& q : dims) {
        size_t s = (size_t)pick_nsplit(q[0], q[1], (int)rows) * q[0] * q[1];
        if (s > m) m = s;
    }
    return m;
}

#define TRY(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != SAMX_OK) return rc__; \
    } while (0)

int grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ---------------------------------------------------------------- role: shared pieces

int rows4(int64_t M) { return (int)((M + 3) / 4); }

// Hidden-layer buffers of one network evaluation over M rows.  Without LayerNorm N1/N2 alias A1/A2.
struct Hid {
    float *A1, *A2;    // post-activation (the lrelu mask source)
    float *N1, *N2;    // what the next layer reads: LayerNorm output, or the activation itself
    float *XH1, *XH2;  // normalised activations (LayerNorm only)
    float *R1, *R2;    // 1/std per row (LayerNorm only)
};

Hid carve_hid(Arena& ar, const samx_mlp* d, int64_t M) {
    Hid h;
    const int h1 = d ? d->h1 : 1, h2 = d ? d->h2 : 1;
    const bool ln = d && d->layernorm;
    h.A1 = ar.take<float>(M * h1);
    h.A2 = ar.take<float>(M * h2);
    h.N1 = ln ? ar.take<float>(M * h1) : h.A1;
    h.N2 = ln ? ar.take<float>(M * h2) : h.A2;
    h.XH1 = ar.take<float>(ln ? M * h1 : 1);
    h.XH2 = ar.take<float>(ln ? M * h2 : 1);
    h.R1 = ar.take<float>(ln ? M : 1);
    h.R2 = ar.take<float>(ln ? M : 1);
    return h;
}

Hid rows_of(const Hid& h, const Net& c, int64_t r0) {  // the same buffers from row r0 on
    Hid o = h;
    o.A1 += r0 * c.h1; o.A2 += r0 * c.h2;
    o.N1 += r0 * c.h1; o.N2 += r0 * c.h2;
    if (c.ln) { o.XH1 += r0 * c.h1; o.XH2 += r0 * c.h2; o.R1 += r0; o.R2 += r0; }
    return o;
}

int ln_fwd(const float* A, const float* beta, const float* gamma, int64_t M, int H, float* XH, float* N,
           float* R, hipStream_t st) {
    ln_fwd_kernel<<<rows4(M), 256, 0, st>>>(A, beta, gamma, M, H, XH, N, R);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}
int ln_bwd(const float* DN, const float* A, const float* XH, const float* R, const float* gamma, int64_t M,
           int H, const float* add, float* DZ, float* TA, hipStream_t st) {
    ln_bwd_kernel<<<rows4(M), 256, 0, st>>>(DN, A, XH, R, gamma, M, H, add, DZ, TA);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}
int ln_chunks(int64_t M) {
    int64_t s = (M + 63) / 64;
    return (int)(s < 1 ? 1 : (s > kLnChunksMax ? kLnChunksMax : s));
}
// dbeta over rows < Mb, dgamma over rows < Mg; part: 2 * kLnChunksMax * H floats
int ln_pgrad(const float* DN, const float* XH, int64_t Mb, int64_t Mg, int H, float* dbeta, float* dgamma,
             float* part, hipStream_t st) {
    const int64_t M = Mb > Mg ? Mb : Mg;
    const int S = ln_chunks(M);
    const int64_t chunk = ((M + S - 1) / S + 3) / 4 * 4;
    ln_pgrad1_kernel<<<dim3((H + 63) / 64, S), 256, 0, st>>>(DN, XH, Mb, Mg, H, chunk, part);
    SAMX_LAUNCH_CHECK();
    ln_pgrad2_kernel<<<(H + 63) / 64, 256, 0, st>>>(part, S, H, dbeta, dgamma);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

// forward through the two hidden layers: h.N2 is the input of the head
int hidden(const Net& c, const float* X, int64_t ldx, int64_t M, const Hid& h, hipStream_t st) {
    TRY(fwd(X, ldx, M, c.in, c.W1, c.b1, c.h1, h.A1, c.h1, EPI_BIAS_LRELU, 0.f, nullptr, 0, st));
    if (c.ln) TRY(ln_fwd(h.A1, c.beta1, c.gamma1, M, c.h1, h.XH1, h.N1, h.R1, st));
    TRY(fwd(h.N1, c.h1, M, c.h1, c.W2, c.b2, c.h2, h.A2, c.h2, EPI_BIAS_LRELU, 0.f, nullptr, 0, st));
    if (c.ln) TRY(ln_fwd(h.A2, c.beta2, c.gamma2, M, c.h2, h.XH2, h.N2, h.R2, st));
    return SAMX_OK;
}

// Cotangent buffers of the two hidden layers: DN = of the layer output (LayerNorm only), DZ = of the
// pre-activation.
struct Cot {
    float *DN1, *DN2, *DZ1, *DZ2;
};
Cot carve_cot(Arena& ar, const samx_mlp* d, int64_t M) {
    Cot c;
    const int h1 = d ? d->h1 : 1, h2 = d ? d->h2 : 1;
    const bool ln = d && d->layernorm;
    c.DZ2 = ar.take<float>(M * h2);
    c.DZ1 = ar.take<float>(M * h1);
    c.DN2 = ln ? ar.take<float>(M * h2) : c.DZ2;
    c.DN1 = ln ? ar.take<float>(M * h1) : c.DZ1;
    return c;
}
// where a head (or the layer above) must leave the cotangent of the second hidden layer's output
float* top_cot(const Net& c, const Cot& k) { return c.ln ? k.DN2 : k.DZ2; }

// from the cotangent of the second hidden output (k.DN2 with LayerNorm; without it k.DZ2 already holds
// the masked pre-activation cotangent) down to k.DZ1
int back_hidden(const Net& c, int64_t M, const Hid& h, const Cot& k, hipStream_t st) {
    if (c.ln) {
        TRY(ln_bwd(k.DN2, h.A2, h.XH2, h.R2, c.gamma2, M, c.h2, nullptr, k.DZ2, nullptr, st));
        TRY(bwd_data(k.DZ2, c.h2, M, c.h2, c.W2, c.h2, c.h1, k.DN1, c.h1, EPI_NONE, nullptr, 0, nullptr, 0, 0.f,
                     st));
        TRY(ln_bwd(k.DN1, h.A1, h.XH1, h.R1, c.gamma1, M, c.h1, nullptr, k.DZ1, nullptr, st));
    } else {
        TRY(bwd_data(k.DZ2, c.h2, M, c.h2, c.W2, c.h2, c.h1, k.DZ1, c.h1, EPI_MASK, h.A1, c.h1, nullptr, 0, 0.f,
                     st));
    }
    return SAMX_OK;
}

// LayerNorm parameter gradients of both layers into the flat gradient
int ln_pgrads(const Net& c, int64_t Mb, int64_t Mg, const Hid& h, const Cot& k, float* grad, float* part,
              hipStream_t st) {
    if (!c.ln) return SAMX_OK;
    TRY(ln_pgrad(k.DN2, h.XH2, Mb, Mg, c.h2, grad + c.obeta2, grad + c.ogamma2, part, st));
    TRY(ln_pgrad(k.DN1, h.XH1, Mb, Mg, c.h1, grad + c.obeta1, grad + c.ogamma1, part, st));
    return SAMX_OK;
}

// XC[:, 0:ob] = obz ; XC[:, ob:] = actor(obz) ; hidden activations stay in ha
int actor_into(const Net& a, const float* obs, const float* mean, const float* stdv, int64_t B, int acdim,
               float high, float* XC, const Hid& ha, hipStream_t st) {
    const int ob = a.in, ld = ob + acdim;
    prep_kernel<<<grid1d(B * ob), 256, 0, st>>>(obs, nullptr, mean, stdv, B, ob, 0, XC, ld);
    SAMX_LAUNCH_CHECK();
    TRY(hidden(a, XC, ld, B, ha, st));
    TRY(fwd(ha.N2, a.h2, B, a.h2, a.W3, a.b3, a.out, XC + ob, ld, EPI_BIAS_TANH, high, nullptr, 0, st));
    return SAMX_OK;
}

struct AgentWs {
    float* XC;
    Hid ha, hc, hd;   // actor, critic, reward-net evaluations
    Cot ka, kc, kd;
    float *dAd, *dPre, *row0, *row1, *row2, *partials, *lnpart;
    double* wd_part;
    size_t partial_floats;
};

AgentWs carve_agent(Arena& ar, const samx_mlp* da, const samx_mlp* dc, const samx_mlp* dd, int64_t B) {
    AgentWs w;
    // action-gradient rows exist only with an actor: a critic-only carve (reward, critic gradient, TD errors)
    // must fit any workspace sized by samx_nets_ws_bytes for the same critic, whatever actor was named there
    const int acdim = da ? dc->in_dim - da->in_dim : 0;
    w.XC = ar.take<float>(B * dc->in_dim);
    w.ha = carve_hid(ar, da, B);
    w.hc = carve_hid(ar, dc, B);
    w.hd = carve_hid(ar, dd, B);
    w.ka = carve_cot(ar, da, B);
    w.kc = carve_cot(ar, dc, B);
    w.kd = carve_cot(ar, dd, B);
    w.dAd = ar.take<float>(B * (acdim > 0 ? acdim : 1));
    w.dPre = ar.take<float>(B * (acdim > 0 ? acdim : 1));
    w.row0 = ar.take<float>(B);
    w.row1 = ar.take<float>(B);
    w.row2 = ar.take<float>(B);
    w.wd_part = ar.take<double>(kWdBlocks);
    int hmax = dc->h1 > dc->h2 ? dc->h1 : dc->h2;
    if (da) { hmax = da->h1 > hmax ? da->h1 : hmax; hmax = da->h2 > hmax ? da->h2 : hmax; }
    w.lnpart = ar.take<float>((size_t)2 * kLnChunksMax * hmax);
    size_t pf = wgrad_partial_floats(dc, B);
    if (da) { size_t s = wgrad_partial_floats(da, B); if (s > pf) pf = s; }
    w.partial_floats = pf;
    w.partials = ar.take<float>(pf);
    return w;
}

void set_ret(HeadArgs& h, const samx_retnorm* ret) {
    if (ret && ret->mean) { h.ret_mean = ret->mean; h.ret_std = ret->std; }
}

HeadArgs head_of(const Net& c, const Hid& h, int64_t M) {
    HeadArgs a;
    a.H2 = h.N2; a.w3 = c.W3; a.b3 = c.b3; a.M = M; a.h2 = c.h2; a.ln = c.ln;
    return a;
}

}  // namespace
}  // namespace samx

using namespace samx;

extern "C" {

size_t samx_nets_ws_bytes(const samx_mlp* actor, const samx_mlp* critic, const samx_mlp* reward_net,
                          int64_t B) {
    if (!actor || !critic || B <= 0) return 0;
    Arena ar(nullptr, 0);
    carve_agent(ar, actor, critic, reward_net, B);
    return align_up(ar.off, 256) + 256;
}

int samx_act_q(const samx_mlp* da, const float* pa, const samx_mlp* dc, const float* pc,
               const float* obs, const float* mean, const float* stdv, int64_t B, float high,
               float* ac_out, float* q_out, const samx_retnorm* ret, void* ws, size_t ws_bytes,
               void* stream) {
    TRY(check_desc(da, "samx_act_q(actor)"));
    TRY(check_desc(dc, "samx_act_q(critic)"));
    SAMX_REQUIRE(pa && pc && obs && ac_out && ws, "samx_act_q: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_act_q: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dc->in_dim == da->in_dim + da->out_dim && dc->out_dim == 1,
                 "samx_act_q: critic input must be ob+ac and its output 1");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, da, dc, nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_act_q: workspace %zu < %zu", ws_bytes, ar.off);
    Net a = split(da, pa), c = split(dc, pc);
    TRY(actor_into(a, obs, mean, stdv, B, da->out_dim, high, w.XC, w.ha, st));
    copy_cols_kernel<<<grid1d(B * da->out_dim), 256, 0, st>>>(w.XC + da->in_dim, dc->in_dim, B,
                                                             da->out_dim, ac_out);
    SAMX_LAUNCH_CHECK();
    if (q_out) {
        TRY(hidden(c, w.XC, c.in, B, w.hc, st));
        HeadArgs h = head_of(c, w.hc, B);
        h.out0 = q_out;
        set_ret(h, ret);
        TRY(launch_head<HEAD_Q>(h, st));
    }
    return SAMX_OK;
}

int samx_targ_q(const samx_mlp* da, const float* pa, const samx_mlp* dc, const float* pc,
                const float* obs1, const float* rews, const float* dones1, const float* td_len,
                const float* mean, const float* stdv, int64_t B, float gamma, float high, float* out,
                const samx_retnorm* ret, void* ws, size_t ws_bytes, void* stream) {
    TRY(check_desc(da, "samx_targ_q(actor)"));
    TRY(check_desc(dc, "samx_targ_q(critic)"));
    SAMX_REQUIRE(pa && pc && obs1 && rews && dones1 && out && ws, "samx_targ_q: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_targ_q: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dc->in_dim == da->in_dim + da->out_dim && dc->out_dim == 1,
                 "samx_targ_q: critic input must be ob+ac and its output 1");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, da, dc, nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_targ_q: workspace %zu < %zu", ws_bytes, ar.off);
    Net a = split(da, pa), c = split(dc, pc);
    TRY(actor_into(a, obs1, mean, stdv, B, da->out_dim, high, w.XC, w.ha, st));
    TRY(hidden(c, w.XC, c.in, B, w.hc, st));
    HeadArgs h = head_of(c, w.hc, B);
    h.r0 = rews; h.r1 = dones1; h.r2 = td_len; h.c0 = gamma; h.out0 = out;
    set_ret(h, ret);
    return launch_head<HEAD_TARG>(h, st);
}

int samx_td_errors(const samx_mlp* dc, const float* pc, const float* obs0, const float* acs,
                   const float* tc1s, const float* mean, const float* stdv, int64_t B, int ob_dim,
                   float* td_out, const samx_retnorm* ret, void* ws, size_t ws_bytes, void* stream) {
    TRY(check_desc(dc, "samx_td_errors"));
    SAMX_REQUIRE(pc && obs0 && acs && tc1s && td_out && ws, "samx_td_errors: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_td_errors: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dc->out_dim == 1 && ob_dim > 0 && ob_dim < dc->in_dim,
                 "samx_td_errors: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, nullptr, dc, nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_td_errors: workspace %zu < %zu", ws_bytes, ar.off);
    Net c = split(dc, pc);
    prep_kernel<<<grid1d(B * c.in), 256, 0, st>>>(obs0, acs, mean, stdv, B, ob_dim, c.in - ob_dim, w.XC,
                                                 c.in);
    SAMX_LAUNCH_CHECK();
    TRY(hidden(c, w.XC, c.in, B, w.hc, st));
    HeadArgs h = head_of(c, w.hc, B);
    h.r0 = tc1s; h.out0 = td_out;
    set_ret(h, ret);
    return launch_head<HEAD_TD>(h, st);
}

int samx_reward(const samx_mlp* dd, const float* pd, const float* obs, const float* acs,
                const float* mean, const float* stdv, int64_t B, int ob_dim, int non_satur_grad,
                float* out, void* ws, size_t ws_bytes, void* stream) {
    TRY(check_desc(dd, "samx_reward"));
    SAMX_REQUIRE(pd && obs && acs && out && ws, "samx_reward: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_reward: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dd->out_dim == 1 && ob_dim > 0 && ob_dim < dd->in_dim, "samx_reward: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, nullptr, dd, nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_reward: workspace %zu < %zu", ws_bytes, ar.off);
    Net c = split(dd, pd);
    prep_kernel<<<grid1d(B * c.in), 256, 0, st>>>(obs, acs, mean, stdv, B, ob_dim, c.in - ob_dim, w.XC,
                                                 c.in);
    SAMX_LAUNCH_CHECK();
    TRY(hidden(c, w.XC, c.in, B, w.hc, st));
    HeadArgs h = head_of(c, w.hc, B);
    h.flag = non_satur_grad; h.out0 = out;
    return launch_head<HEAD_REWARD>(h, st);
}

int samx_critic_grad(const samx_mlp* dc, const float* pc, const float* obs0, const float* acs,
                     const float* iws, const float* tc1s, const float* mean, const float* stdv,
                     int64_t B, int ob_dim, float td_scale, float wd_scale, float* losses_out,
                     float* grad_out, float* td_out, const samx_retnorm* ret, const float* tcns,
                     float tdn_scale, float* tdn_out, void* ws, size_t ws_bytes, void* stream) {
    TRY(check_desc(dc, "samx_critic_grad"));
    SAMX_REQUIRE(pc && obs0 && acs && tc1s && losses_out && grad_out && ws,
                 "samx_critic_grad: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_critic_grad: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dc->out_dim == 1 && ob_dim > 0 && ob_dim < dc->in_dim,
                 "samx_critic_grad: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, nullptr, dc, nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_critic_grad: workspace %zu < %zu", ws_bytes, ar.off);
    Net c = split(dc, pc);
    prep_kernel<<<grid1d(B * c.in), 256, 0, st>>>(obs0, acs, mean, stdv, B, ob_dim, c.in - ob_dim, w.XC,
                                                 c.in);
    SAMX_LAUNCH_CHECK();
    TRY(hidden(c, w.XC, c.in, B, w.hc, st));
    HeadArgs h = head_of(c, w.hc, B);
    h.r0 = tc1s; h.r1 = iws; h.c0 = td_scale / (float)B;
    h.out0 = w.row0; h.out1 = td_out; h.out2 = w.row1; h.DZ2 = top_cot(c, w.kc);
    set_ret(h, ret);
    if (tcns) { h.rn = tcns; h.cn = tdn_scale / (float)B; h.out4 = w.row2; h.out5 = tdn_out; }
    TRY(launch_head<HEAD_CRITIC>(h, st));
    TRY(wgrad(w.hc.N2, c.h2, c.h2, w.row1, 1, 1, B, B, grad_out + c.oW3, 0.f, nullptr, w.partials,
              w.partial_floats, st));
    TRY(back_hidden(c, B, w.hc, w.kc, st));
    TRY(wgrad(w.hc.N1, c.h1, c.h1, w.kc.DZ2, c.h2, c.h2, B, B, grad_out + c.oW2, wd_scale, c.W2, w.partials,
              w.partial_floats, st));
    TRY(wgrad(w.XC, c.in, c.in, w.kc.DZ1, c.h1, c.h1, B, B, grad_out + c.oW1, wd_scale, c.W1, w.partials,
              w.partial_floats, st));
    TRY(ln_pgrads(c, B, B, w.hc, w.kc, grad_out, w.lnpart, st));
    if (wd_scale > 0.f) {
        wd_sumsq_kernel<<<kWdBlocks, 256, 0, st>>>(c.W1, (int64_t)c.in * c.h1, c.W2, (int64_t)c.h1 * c.h2,
                                                   w.wd_part);
        SAMX_LAUNCH_CHECK();
    }
    critic_losses_kernel<<<1, 1024, 0, st>>>(w.row0, B, td_scale, wd_scale, w.wd_part, kWdBlocks, losses_out,
                                             tcns ? w.row2 : nullptr, tdn_scale);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_actor_grad(const samx_mlp* da, const float* pa, const samx_mlp* dc, const float* pc,
                    const samx_mlp* dd, const float* pd, const float* obs0, const float* mean,
                    const float* stdv, int64_t B, float high, float q_scale, float d_scale,
                    float* losses_out, float* grad_out, const samx_retnorm* ret, void* ws, size_t ws_bytes,
                    void* stream) {
    TRY(check_desc(da, "samx_actor_grad(actor)"));
    TRY(check_desc(dc, "samx_actor_grad(critic)"));
    const bool use_d = d_scale != 0.f;
    if (use_d) {
        TRY(check_desc(dd, "samx_actor_grad(reward net)"));
        SAMX_REQUIRE(pd, "samx_actor_grad: d_scale != 0 needs reward-net parameters");
        SAMX_REQUIRE(dd->in_dim == dc->in_dim && dd->out_dim == 1, "samx_actor_grad: reward-net shape");
    }
    SAMX_REQUIRE(pa && pc && obs0 && losses_out && grad_out && ws, "samx_actor_grad: null pointer");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_actor_grad: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dc->in_dim == da->in_dim + da->out_dim && dc->out_dim == 1,
                 "samx_actor_grad: critic input must be ob+ac and its output 1");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    AgentWs w = carve_agent(ar, da, dc, use_d ? dd : nullptr, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_actor_grad: workspace %zu < %zu", ws_bytes, ar.off);
    Net a = split(da, pa), c = split(dc, pc);
    const int ob = a.in, A = a.out, ld = ob + A;
    TRY(actor_into(a, obs0, mean, stdv, B, A, high, w.XC, w.ha, st));
    // Q branch: forward, then back to the action input
    TRY(hidden(c, w.XC, c.in, B, w.hc, st));
    HeadArgs h = head_of(c, w.hc, B);
    h.c0 = -q_scale / (float)B; h.out0 = w.row0; h.DZ2 = top_cot(c, w.kc);
    set_ret(h, ret);
    TRY(launch_head<HEAD_ACTOR_Q>(h, st));
    TRY(back_hidden(c, B, w.hc, w.kc, st));
    const float* add = nullptr;
    if (use_d) {
        Net d = split(dd, pd);
        TRY(hidden(d, w.XC, d.in, B, w.hd, st));
        HeadArgs hd = head_of(d, w.hd, B);
        hd.c0 = -d_scale / (float)B; hd.out0 = w.row1; hd.DZ2 = top_cot(d, w.kd);
        set_ret(hd, ret);  // d_pred_w_actor is dermsified too (sam_agent.py:191-194)
        TRY(launch_head<HEAD_ACTOR_Q>(hd, st));
        TRY(back_hidden(d, B, w.hd, w.kd, st));
        TRY(bwd_data(w.kd.DZ1, d.h1, B, d.h1, d.W1 + (int64_t)ob * d.h1, d.h1, A, w.dAd, A, EPI_NONE,
                     nullptr, 0, nullptr, 0, 0.f, st));
        add = w.dAd;
    }
    // d loss / d pre-tanh = (dQ/da [+ dD/da]) * high * (1 - tanh^2)
    TRY(bwd_data(w.kc.DZ1, c.h1, B, c.h1, c.W1 + (int64_t)ob * c.h1, c.h1, A, w.dPre, A, EPI_ADD_TANHGRAD,
                 w.XC + ob, ld, add, A, high, st));
    TRY(wgrad(w.ha.N2, a.h2, a.h2, w.dPre, A, A, B, B, grad_out + a.oW3, 0.f, nullptr, w.partials,
              w.partial_floats, st));
    // actor backward: cotangent of its second hidden output = dPre W3^T
    if (a.ln) {
        TRY(bwd_data(w.dPre, A, B, A, a.W3, A, a.h2, w.ka.DN2, a.h2, EPI_NONE, nullptr, 0, nullptr, 0, 0.f, st));
    } else {
        TRY(bwd_data(w.dPre, A, B, A, a.W3, A, a.h2, w.ka.DZ2, a.h2, EPI_MASK, w.ha.A2, a.h2, nullptr, 0, 0.f,
                     st));
    }
    TRY(back_hidden(a, B, w.ha, w.ka, st));
    TRY(wgrad(w.ha.N1, a.h1, a.h1, w.ka.DZ2, a.h2, a.h2, B, B, grad_out + a.oW2, 0.f, nullptr, w.partials,
              w.partial_floats, st));
    TRY(wgrad(w.XC, ld, ob, w.ka.DZ1, a.h1, a.h1, B, B, grad_out + a.oW1, 0.f, nullptr, w.partials,
              w.partial_floats, st));
    TRY(ln_pgrads(a, B, B, w.ha, w.ka, grad_out, w.lnpart, st));
    actor_losses_kernel<<<1, 1024, 0, st>>>(w.row0, use_d ? w.row1 : nullptr, B, q_scale, d_scale,
                                            losses_out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

namespace {
// Row blocks of the discriminator pass.  Without LayerNorm: [policy | expert | interpolate] (3B).
// With LayerNorm: [policy | expert | interpolate copy (primal stream) | interpolate (tangent stream)]
// (4B) — see samx_disc_grad.
struct DiscWs {
    float* X;
    Hid h;
    Cot k;
    float *DS, *G, *ce, *ent, *sg, *gp, *partials, *ZD, *TA1, *TA2, *lnpart;
    double* gp_part;
    size_t partial_floats;
};
DiscWs carve_disc(Arena& ar, const samx_mlp* dd, int64_t B) {
    DiscWs w;
    const bool ln = dd->layernorm;
    const int64_t M = (ln ? 4 : 3) * B;
    const int hmax = dd->h1 > dd->h2 ? dd->h1 : dd->h2;
    w.X = ar.take<float>(M * dd->in_dim);
    w.h = carve_hid(ar, dd, M);
    w.k = carve_cot(ar, dd, M);
    w.DS = ar.take<float>(M);
    w.G = ar.take<float>(B * dd->in_dim);
    w.ce = ar.take<float>(2 * B);
    w.ent = ar.take<float>(2 * B);
    w.sg = ar.take<float>(2 * B);
    w.gp = ar.take<float>(4);
    w.gp_part = ar.take<double>(2 * kGpBlocks);
    w.ZD = ar.take<float>(ln ? B * hmax : 1);
    w.TA1 = ar.take<float>(ln ? B * dd->h1 : 1);
    w.TA2 = ar.take<float>(ln ? B * dd->h2 : 1);
    w.lnpart = ar.take<float>(ln ? (size_t)2 * kLnChunksMax * hmax : 1);
    w.partial_floats = wgrad_partial_floats(dd, M);
    w.partials = ar.take<float>(w.partial_floats);
    return w;
}

// interpolate rows also into a second block (LayerNorm path: the primal copy)
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int64_t n,
                                                        float* __restrict__ dst) {
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) dst[t] = src[t];
}
}  // namespace

int samx_ret_update_popart(double* sums, const double* addvec1, const double* addvec2, float* ret_mean,
                           float* ret_std, int enable_popart, const samx_mlp* dc, float* critic_params,
                           float* targ_critic_params, void* stream) {
    SAMX_REQUIRE(sums && addvec1 && ret_mean && ret_std, "samx_ret_update_popart: null pointer");
    float *w3a = nullptr, *b3a = nullptr, *w3b = nullptr, *b3b = nullptr;
    int h2 = 0;
    if (enable_popart) {
        TRY(check_desc(dc, "samx_ret_update_popart"));
        SAMX_REQUIRE(critic_params && targ_critic_params && dc->out_dim == 1,
                     "samx_ret_update_popart: PopArt needs both critics (scalar output)");
        Net a = split(dc, critic_params), b = split(dc, targ_critic_params);
        w3a = critic_params + a.oW3; b3a = critic_params + a.ob3;
        w3b = targ_critic_params + b.oW3; b3b = targ_critic_params + b.ob3;
        h2 = a.h2;
    }
    popart_kernel<<<1, 256, 0, (hipStream_t)stream>>>(sums, addvec1, addvec2, ret_mean, ret_std,
                                                      enable_popart ? 1 : 0, w3a, b3a, w3b, b3b, h2);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

size_t samx_disc_ws_bytes(const samx_mlp* dd, int64_t B) {
    if (!dd || B <= 0) return 0;
    Arena ar(nullptr, 0);
    carve_disc(ar, dd, B);
    return align_up(ar.off, 256) + 256;
}

int samx_disc_grad(const samx_mlp* dd, const float* pd, const float* p_obs, const float* p_acs,
                   const float* e_obs, const float* e_acs, const float* eps_ob, const float* eps_ac,
                   const float* real_labels, const float* fake_labels, const float* mean,
                   const float* stdv, int64_t B, int ob_dim, float ent_reg_scale, float* losses_out,
                   float* grad_out, void* ws, size_t ws_bytes, void* stream) {
    return samx::disc_grad_src(dd, pd, p_obs, p_acs, e_obs, e_acs, eps_ob, eps_ac, real_labels, fake_labels, mean,
                               stdv, B, ob_dim, ent_reg_scale, losses_out, grad_out, ws, ws_bytes, nullptr, stream);
}

}  // extern "C"

namespace samx {
int disc_grad_src(const samx_mlp* dd, const float* pd, const float* p_obs, const float* p_acs, const float* e_obs,
                  const float* e_acs, const float* eps_ob, const float* eps_ac, const float* real_labels,
                  const float* fake_labels, const float* mean, const float* stdv, int64_t B, int ob_dim,
                  float ent_reg_scale, float* losses_out, float* grad_out, void* ws, size_t ws_bytes,
                  const DiscSrc* srcp, void* stream) {
    const DiscSrc src = srcp ? *srcp : DiscSrc();
    TRY(check_desc(dd, "samx_disc_grad"));
    SAMX_REQUIRE(pd && p_obs && p_acs && e_obs && e_acs && (src.draw || (eps_ob && eps_ac)) && losses_out &&
                     grad_out && ws,
                 "samx_disc_grad: null pointer");
    SAMX_REQUIRE(!src.draw || src.step_dev, "samx_disc_grad: in-kernel draws need the device step counter");
    SAMX_REQUIRE(!src.p_idx || (src.ring_len > 0 && src.ring_start >= 0 && src.ring_start < src.ring_len),
                 "samx_disc_grad: bad ring geometry");
    SAMX_REQUIRE((mean == nullptr) == (stdv == nullptr), "samx_disc_grad: mean/std must come together");
    SAMX_REQUIRE(B > 0 && dd->out_dim == 1 && ob_dim > 0 && ob_dim < dd->in_dim, "samx_disc_grad: bad sizes");
    SAMX_REQUIRE(ent_reg_scale >= 0.f, "samx_disc_grad: ent_reg_scale must be non-negative");
    hipStream_t st = (hipStream_t)stream;
    Arena ar(ws, ws_bytes);
    DiscWs w = carve_disc(ar, dd, B);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "samx_disc_grad: workspace %zu < %zu", ws_bytes, ar.off);
    Net d = split(dd, pd);
    const int ob = ob_dim, ac = d.in - ob_dim;
    const float gp_coeff = 10.f;  // discriminator.py:154
    if (src.draw)
        disc_prep_draw_kernel<<<grid1d((B * d.in + 3) / 4 + B + 2), 256, 0, st>>>(p_obs, p_acs, e_obs, e_acs, mean, stdv, B,
                                                                              ob, ac, w.X, src);
    else
        disc_prep_kernel<<<grid1d(B * d.in), 256, 0, st>>>(p_obs, p_acs, e_obs, e_acs, eps_ob, eps_ac, mean,
                                                          stdv, B, ob, ac, w.X, src);
    SAMX_LAUNCH_CHECK();
    if (src.draw) { real_labels = src.real_out; fake_labels = src.fake_out; }

    if (!d.ln) {
        // ---- leaky-ReLU only: the tangent of the network is linear in its masks, the second-order
        // term has no primal stream.  Rows [0,B) policy, [B,2B) expert, [2B,3B) interpolates.
        const int64_t M = 3 * B;
        TRY(hidden(d, w.X, d.in, M, w.h, st));
        HeadArgs h = head_of(d, w.h, M);
        h.B = B; h.r0 = fake_labels; h.r1 = real_labels; h.c1 = ent_reg_scale;
        h.out0 = w.DS; h.out1 = w.ce; h.out2 = w.ent; h.out3 = w.sg; h.DZ2 = w.k.DZ2;
        TRY(launch_head<HEAD_DISC>(h, st));
        // back to the first pre-activation for all 3B rows; for the interpolate rows this is the
        // input-gradient chain of discriminator.py:136
        TRY(back_hidden(d, M, w.h, w.k, st));
        const float* DZ1i = w.k.DZ1 + 2 * B * d.h1;
        TRY(bwd_data(DZ1i, d.h1, B, d.h1, d.W1, d.h1, d.in, w.G, d.in, EPI_NONE, nullptr, 0, nullptr, 0, 0.f,
                     st));
        gp_sumsq_kernel<<<kGpBlocks, 256, 0, st>>>(w.G, B, ob, ac, w.gp_part);
        SAMX_LAUNCH_CHECK();
        // V (cotangent of the input gradient) replaces the interpolates in X: not needed again
        float* Xi = w.X + 2 * B * d.in;
        gp_scale_kernel<<<grid1d(B * d.in), 256, 0, st>>>(w.G, B, ob, ac, w.gp_part, kGpBlocks, gp_coeff, Xi,
                                                         w.gp);
        SAMX_LAUNCH_CHECK();
        // tangent pass along V through the interpolate rows, in place over their activations
        float* A1i = w.h.A1 + 2 * B * d.h1;
        float* A2i = w.h.A2 + 2 * B * d.h2;
        TRY(fwd(Xi, d.in, B, d.in, d.W1, nullptr, d.h1, A1i, d.h1, EPI_MASK, 0.f, A1i, d.h1, st));
        TRY(fwd(A1i, d.h1, B, d.h1, d.W2, nullptr, d.h2, A2i, d.h2, EPI_MASK, 0.f, A2i, d.h2, st));
        // parameter gradients: one reduction over the 3B stacked rows per layer
        TRY(wgrad(w.h.A2, d.h2, d.h2, w.DS, 1, 1, M, 2 * B, grad_out + d.oW3, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
        TRY(wgrad(w.h.A1, d.h1, d.h1, w.k.DZ2, d.h2, d.h2, M, 2 * B, grad_out + d.oW2, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
        TRY(wgrad(w.X, d.in, d.in, w.k.DZ1, d.h1, d.h1, M, 2 * B, grad_out + d.oW1, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
    } else {
        // ---- with LayerNorm the tangent of a layer depends on its primal activation, so the
        // gradient of <V, d score/d input> has two streams:
        //   T (tangent)  rows Ri = [3B,4B): the first backward leaves T_n / T_z there; afterwards their
        //                forward-side slots are overwritten by the tangent forward (V, u, n-dot);
        //   P (primal)   rows Rt = [2B,3B): an exact copy of the interpolates' forward; receives the
        //                cotangents the tangent layers send back to the primal activations.
        // One weight-gradient reduction per layer then runs over all 4B stacked rows, biases and
        // LayerNorm betas over the first 3B.
        const int64_t M = 4 * B, Rt = 2 * B, Ri = 3 * B;
        copy_rows_kernel<<<grid1d(B * d.in), 256, 0, st>>>(w.X + Rt * d.in, B * d.in, w.X + Ri * d.in);
        SAMX_LAUNCH_CHECK();
        TRY(hidden(d, w.X, d.in, M, w.h, st));
        HeadArgs h = head_of(d, w.h, M);
        h.B = B; h.r0 = fake_labels; h.r1 = real_labels; h.c1 = ent_reg_scale;
        h.out0 = w.DS; h.out1 = w.ce; h.out2 = w.ent; h.out3 = w.sg; h.DZ2 = w.k.DN2; h.flag = 1;
        TRY(launch_head<HEAD_DISC>(h, st));  // seeds: policy/expert dq, Rt 0, Ri 1
        // first backward over all rows, keeping T_a (pre-mask) of the Ri rows
        Hid hi = rows_of(w.h, d, Ri), ht = rows_of(w.h, d, Rt);
        TRY(ln_bwd(w.k.DN2, w.h.A2, w.h.XH2, w.h.R2, d.gamma2, Ri, d.h2, nullptr, w.k.DZ2, nullptr, st));
        TRY(ln_bwd(w.k.DN2 + Ri * d.h2, hi.A2, hi.XH2, hi.R2, d.gamma2, B, d.h2, nullptr, w.k.DZ2 + Ri * d.h2,
                   w.TA2, st));
        TRY(bwd_data(w.k.DZ2, d.h2, M, d.h2, d.W2, d.h2, d.h1, w.k.DN1, d.h1, EPI_NONE, nullptr, 0, nullptr, 0,
                     0.f, st));
        TRY(ln_bwd(w.k.DN1, w.h.A1, w.h.XH1, w.h.R1, d.gamma1, Ri, d.h1, nullptr, w.k.DZ1, nullptr, st));
        TRY(ln_bwd(w.k.DN1 + Ri * d.h1, hi.A1, hi.XH1, hi.R1, d.gamma1, B, d.h1, nullptr, w.k.DZ1 + Ri * d.h1,
                   w.TA1, st));
        TRY(bwd_data(w.k.DZ1 + Ri * d.h1, d.h1, B, d.h1, d.W1, d.h1, d.in, w.G, d.in, EPI_NONE, nullptr, 0,
                     nullptr, 0, 0.f, st));
        gp_sumsq_kernel<<<kGpBlocks, 256, 0, st>>>(w.G, B, ob, ac, w.gp_part);
        SAMX_LAUNCH_CHECK();
        float* Xi = w.X + Ri * d.in;  // V replaces the Ri copy of the interpolates
        gp_scale_kernel<<<grid1d(B * d.in), 256, 0, st>>>(w.G, B, ob, ac, w.gp_part, kGpBlocks, gp_coeff, Xi,
                                                         w.gp);
        SAMX_LAUNCH_CHECK();
        // tangent forward, layer 1: z-dot = V W1 ; LayerNorm tangent with the primal copy's statistics.
        // Its primal cotangent P_a1 is parked (unmasked) in DZ1[Rt] until P_n1 has come down.
        TRY(fwd(Xi, d.in, B, d.in, d.W1, nullptr, d.h1, w.ZD, d.h1, EPI_NONE, 0.f, nullptr, 0, st));
        ln_tan_kernel<<<rows4(B), 256, 0, st>>>(w.ZD, ht.A1, ht.XH1, ht.R1, d.gamma1, w.k.DN1 + Ri * d.h1, w.TA1,
                                                B, d.h1, hi.XH1, hi.N1, w.k.DZ1 + Rt * d.h1, 0);
        SAMX_LAUNCH_CHECK();
        // layer 2: z-dot = n-dot1 W2 ; P_z2 = P_a2 * mask straight into DZ2[Rt]
        TRY(fwd(hi.N1, d.h1, B, d.h1, d.W2, nullptr, d.h2, w.ZD, d.h2, EPI_NONE, 0.f, nullptr, 0, st));
        ln_tan_kernel<<<rows4(B), 256, 0, st>>>(w.ZD, ht.A2, ht.XH2, ht.R2, d.gamma2, w.k.DN2 + Ri * d.h2, w.TA2,
                                                B, d.h2, hi.XH2, hi.N2, w.k.DZ2 + Rt * d.h2, 1);
        SAMX_LAUNCH_CHECK();
        // primal stream: P_n1 = P_z2 W2^T ; P_z1 = (J1(gamma1 P_n1) + P_a1) * mask
        TRY(bwd_data(w.k.DZ2 + Rt * d.h2, d.h2, B, d.h2, d.W2, d.h2, d.h1, w.k.DN1 + Rt * d.h1, d.h1, EPI_NONE,
                     nullptr, 0, nullptr, 0, 0.f, st));
        TRY(ln_bwd(w.k.DN1 + Rt * d.h1, ht.A1, ht.XH1, ht.R1, d.gamma1, B, d.h1, w.k.DZ1 + Rt * d.h1,
                   w.k.DZ1 + Rt * d.h1, nullptr, st));
        // parameter gradients over the 4B stacked rows (ones row over the first 3B)
        TRY(wgrad(w.h.N2, d.h2, d.h2, w.DS, 1, 1, M, 3 * B, grad_out + d.oW3, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
        TRY(wgrad(w.h.N1, d.h1, d.h1, w.k.DZ2, d.h2, d.h2, M, 3 * B, grad_out + d.oW2, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
        TRY(wgrad(w.X, d.in, d.in, w.k.DZ1, d.h1, d.h1, M, 3 * B, grad_out + d.oW1, 0.f, nullptr, w.partials,
                  w.partial_floats, st));
        TRY(ln_pgrads(d, 3 * B, M, w.h, w.k, grad_out, w.lnpart, st));
    }
    disc_losses_kernel<<<1, 1024, 0, st>>>(w.ce, w.ent, w.sg, w.gp, B, ent_reg_scale, gp_coeff,
                                           losses_out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

namespace {
// ---- reduction that finishes the weight gradients of a 256x256 learner (after fused_wide.hip + the grouped
// split products): every element of the three flat gradients = fixed-order sum of its partial sums
// (split-GEMM partials [split][rows][cols], or per-tile head slots [tile][257]) [+ k * the interpolate rows'
// partial sums, k = the gradient-penalty scalar 10 * 2 (N - 1) / N, discriminator.py:138-140,154]
// [+ wd * W, network_util.py:43-56]; per-block sums of squares for the clip norm; the step counters tick.
struct WPiece {
    const float* p1; int n1; int64_t s1;                      // plain partial sums
    const float* p2; int n2; int64_t s2; int64_t n2_valid;    // partial sums that take the scalar k
    const float* wdW; float wd; int64_t wd_count;             // weight decay on the first wd_count elements
    int64_t count;                                            // elements of this piece ([W; b] contiguous)
};
constexpr int kWPieces = 5;   // [W1; b1] [beta1; gamma1] [W2; b2] [beta2; gamma2] [W3; b3] with LayerNorm, else 3
struct WFinishSeg {
    WPiece pc[kWPieces];
    int npc;
    int64_t n;
    float* grad;
    float* sumsq;
    int nblocks;
    int64_t* step_dev;
};
struct WLossArgs {
    int64_t B;
    const float* row_hub;             // critic (null = no agent roles)
    const float* row_hubn;            // n-step TD loss rows, or null
    float tdn_scale;
    const float *ret_mean, *ret_std;  // return normalisation (device scalars) or null
    const double* wd_part;
    int wd_blocks;
    float td_scale, wd_scale;
    float* critic_losses;
    const float* row_q;
    float q_scale;
    float* actor_losses;
    const float *ce, *ent, *sg;       // discriminator (null ce = no discriminator role)
    const double* gp_part;
    int tiles, ob, ac;
    float ent_scale, gp_coeff;
    float* d_losses;
};
struct WFinishArgs {
    WLossArgs loss;
    int main_blocks;                  // the three workgroups behind them take the scalar losses
    WFinishSeg seg[3];
    int nseg, gp_seg;
    const double* gp_part;
    int gp_tiles, ob, ac;
    int64_t B;
    float gp_coeff;
    // commit of the inline d.obs_rms.update (helpers/mpi_running_mean_std.py:31-45), or null
    double* rms_sums;
    const double* rms_addvec;
    float *rms_mean, *rms_std;
};
// The scalar losses, by three extra workgroups of the reduction launch: the critic's (sam_agent.py:473-517,
// weight decay network_util.py:43-56 from the role launch's partial sums), the actor's (sam_agent.py:412-430),
// the discriminator's 8 (discriminator.py:145-158, gradient penalty from the interpolate tiles' sums).
__device__ __forceinline__ double sum_f32_x4(const float* x, int64_t n, double* sh) {
    double a = 0.0;
    const int64_t nt = blockDim.x;
    int64_t i = threadIdx.x;
    for (; i + 3 * nt < n; i += 4 * nt) {   // four loads in flight
        const double v0 = x[i], v1 = x[i + nt], v2 = x[i + 2 * nt], v3 = x[i + 3 * nt];
        a += v0 + v1 + v2 + v3;
    }
    for (; i < n; i += nt) a += (double)x[i];
    return block_sum_d(a, sh);
}
__device__ void wide_losses(const WLossArgs& t, int which, double* sh) {
    const int64_t B = t.B;
    const int i = threadIdx.x;
    if (which == 0) {
        if (!t.row_hub) return;
        const double l = sum_f32_x4(t.row_hub, B, sh) / (double)B * t.td_scale;
        const double ln_ = t.row_hubn ? sum_f32_x4(t.row_hubn, B, sh) / (double)B * t.tdn_scale : 0.0;
        double wd = 0.0;
        if (t.wd_scale > 0.f) wd = 0.5 * t.wd_scale * block_sum_d(i < t.wd_blocks ? t.wd_part[i] : 0.0, sh);
        if (i == 0) {
            t.critic_losses[0] = (float)l;
            t.critic_losses[1] = (float)wd;
            t.critic_losses[2] = (float)(l + wd + ln_);
            t.critic_losses[3] = (float)ln_;
        }
    } else if (which == 1) {
        if (!t.row_q) return;
        // with return normalisation the actor loss is on the DE-normalised prediction (sam_agent.py:184-194):
        // mean(s * std + mean) = mean(s) * std + mean
        const double rm = t.ret_mean ? (double)t.ret_mean[0] : 0.0, rs = t.ret_std ? (double)t.ret_std[0] : 1.0;
        const double q = -(sum_f32_x4(t.row_q, B, sh) / (double)B * rs + rm) * t.q_scale;
        if (i == 0) {
            t.actor_losses[0] = (float)q;
            t.actor_losses[1] = 0.f;
            t.actor_losses[2] = (float)q;
        }
    } else {
        if (!t.ce) return;
        const double pl = sum_f32_x4(t.ce, B, sh) / (double)B;
        const double el = sum_f32_x4(t.ce + B, B, sh) / (double)B;
        const double em = sum_f32_x4(t.ent, 2 * B, sh) / (double)(2 * B);
        const double pa = sum_f32_x4(t.sg, B, sh) / (double)B;
        const double ea = sum_f32_x4(t.sg + B, B, sh) / (double)B;
        const double so = block_sum_d(i < t.tiles ? t.gp_part[2 * i] : 0.0, sh);
        const double sa = block_sum_d(i < t.tiles ? t.gp_part[2 * i + 1] : 0.0, sh);
        if (i == 0) {
            const double N = sqrt(so / ((double)B * t.ob) + sa / ((double)B * t.ac));
            const double gp = (N - 1.0) * (N - 1.0);
            const double eloss = -(double)t.ent_scale * em;
            float* losses = t.d_losses;
            losses[0] = (float)pl; losses[1] = (float)el; losses[2] = (float)em; losses[3] = (float)eloss;
            losses[4] = (float)pa; losses[5] = (float)ea; losses[6] = (float)gp;
            losses[7] = (float)(pl + el + eloss + (double)t.gp_coeff * gp);
        }
    }
}

__global__ __launch_bounds__(256) void wide_finish_kernel(WFinishArgs f) {
    __shared__ double shd[4];
    const int tid = threadIdx.x;
    int k = 0, b = blockIdx.x;
    if (b >= f.main_blocks) { wide_losses(f.loss, b - f.main_blocks, shd); return; }
    while (k < f.nseg - 1 && b >= f.seg[k].nblocks) { b -= f.seg[k].nblocks; ++k; }
    const WFinishSeg& s = f.seg[k];
    // The gradient penalty's scalar k: every wave derives it itself (same loads, same fixed tree).  Its loads go out
    // here; the shuffles and the float64 arithmetic wait until the first round's partial sums have been requested
    // (below) — they used to stand in front of those requests, on the segment with the most of them.
    float kgp = 0.f;
    const bool want_kgp = k == f.gp_seg && f.gp_part;
    double gso = 0.0, gsa = 0.0;
    if (want_kgp)
        for (int i = tid & 63; i < f.gp_tiles; i += 64) { gso += f.gp_part[2 * i]; gsa += f.gp_part[2 * i + 1]; }
    auto finish_kgp = [&]() {   // (called where every lane of the wave is active)
        double so = gso, sa = gsa;
        for (int o = 32; o > 0; o >>= 1) { so += __shfl_xor(so, o, 64); sa += __shfl_xor(sa, o, 64); }
        const double N = sqrt(so / ((double)f.B * f.ob) + sa / ((double)f.B * f.ac));
        kgp = (float)((double)f.gp_coeff * 2.0 * (N - 1.0) / N);
    };
    // every thread of the block runs the first round (uniform): then k can be finished inside it
    const bool kgp_late = want_kgp && (int64_t)(b + 1) * 256 <= s.n;
    if (want_kgp && !kgp_late) finish_kgp();
    bool kgp_pending = kgp_late;
    if (b == 0 && tid == 0 && s.step_dev) s.step_dev[0] += 1;
    double acc = 0.0;
    const int64_t stride = (int64_t)s.nblocks * 256;
    constexpr int E = 4;   // elements per thread and round: their partial-sum loads are in flight together
    for (int64_t i0 = (int64_t)b * 256 + tid; i0 < s.n; i0 += E * stride) {
        const float* p1[E]; const float* p2[E]; const float* pw[E];
        int n1[E], n2[E];
        int64_t s1[E], s2[E];
        float wd[E], v[E], u[E];
        int nmax1 = 0, nmax2 = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t i = i0 + e * stride;
            p1[e] = p2[e] = pw[e] = nullptr; n1[e] = n2[e] = 0; s1[e] = s2[e] = 0; wd[e] = 0.f; v[e] = u[e] = 0.f;
            if (i < s.n) {
                int q = 0;
                int64_t li = i;
                while (q < s.npc - 1 && li >= s.pc[q].count) { li -= s.pc[q].count; ++q; }
                const WPiece& pc = s.pc[q];
                p1[e] = pc.p1 + li; n1[e] = pc.n1; s1[e] = pc.s1;
                if (pc.p2 && li < pc.n2_valid) { p2[e] = pc.p2 + li; n2[e] = pc.n2; s2[e] = pc.s2; }
                if (pc.wdW && li < pc.wd_count) { pw[e] = pc.wdW + li; wd[e] = pc.wd; }
                nmax1 = n1[e] > nmax1 ? n1[e] : nmax1;
                nmax2 = n2[e] > nmax2 ? n2[e] : nmax2;
            }
        }
        for (int z = 0; z < nmax1; z += 2) {   // splits in increasing order; 2 x E loads in flight
            float t0[E], t1[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                t0[e] = z < n1[e] ? p1[e][z * s1[e]] : 0.f;
                t1[e] = z + 1 < n1[e] ? p1[e][(z + 1) * s1[e]] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) { if (z < n1[e]) v[e] += t0[e]; if (z + 1 < n1[e]) v[e] += t1[e]; }
        }
        for (int z = 0; z < nmax2; z += 2) {
            float t0[E], t1[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                t0[e] = z < n2[e] ? p2[e][z * s2[e]] : 0.f;
                t1[e] = z + 1 < n2[e] ? p2[e][(z + 1) * s2[e]] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) { if (z < n2[e]) u[e] += t0[e]; if (z + 1 < n2[e]) u[e] += t1[e]; }
        }
        if (kgp_pending) { finish_kgp(); kgp_pending = false; }   // (first round: all lanes are here)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t i = i0 + e * stride;
            if (i < s.n) {
                float g = v[e];
                if (p2[e]) g += kgp * u[e];
                if (pw[e]) g += wd[e] * pw[e][0];
                s.grad[i] = g;
                acc += (double)g * (double)g;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) s.sumsq[b] = (float)(shd[0] + shd[1] + shd[2] + shd[3]);
    // the commit of d.obs_rms (one block; nothing in this launch reads what it writes, so it comes last: its float64
    // divisions used to delay that block's gradient elements)
    if (k == f.gp_seg && b == 0 && f.rms_sums) {   // same arithmetic as the role kernel's loader
        const int dim = f.ob;
        const double cnt = f.rms_sums[2 * dim] + f.rms_addvec[2 * dim];   // every thread reads the old count first
        __syncthreads();
        for (int c = tid; c < dim; c += 256) {
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
}

int wide_nsplit(int64_t R) {   // reduction chunks of 512 rows: 16 stages per workgroup
    static int chunk = 0;   // SAMX_WIDE_CHUNK: rows per chunk (measurement aid)
    if (!chunk) { const char* e = getenv("SAMX_WIDE_CHUNK"); chunk = e ? atoi(e) : 512; if (chunk < 32) chunk = 512; }
    int64_t n = (R + chunk - 1) / chunk;
    return (int)(n < 2 ? 2 : n);
}
// float layout of the reduction workspace: head slots (critic, D), then one partials buffer per product
struct WideRed {
    // w3_sum: [3][sw] critic, D policy+expert, D interpolates (sw = slot width: 257, + the four LayerNorm vectors)
    float *w3c, *w3d, *w3_sum, *P[9], *w3a_tmp;
    float *lna, *lna_sum;   // LayerNorm: the actor's per-tile [beta1 | gamma1 | beta2 | gamma2] slots and their sum
    // k-interleaved images of W1 / W2 (fused_wide.h: Net::W1p / W2p) of actor, critic, reward net, target actor,
    // target critic — scratch, rewritten from the parameter vectors at the head of every launch sequence
    float* pk[5][2];
    // transposed images (Net::W1t / W2t) of actor (W2 only), critic, reward net for the backward products
    float* pt[3][2];
    int ns[9];
    int64_t cnt[9];
    int sw;
};
WideRed carve_wide_red(Arena& ar, int ob, int ac, int64_t B, bool ln) {
    WideRed w;
    const int64_t tiles = B / wide::kRows, in = ob + ac, H = wide::kH;
    w.sw = wide::slot_width(ln);
    w.w3c = ar.take<float>(tiles * w.sw);
    w.w3d = ar.take<float>(3 * tiles * w.sw);
    w.w3_sum = ar.take<float>(3 * w.sw);
    w.lna = ar.take<float>(ln ? tiles * wide::kLnSlot : 1);
    w.lna_sum = ar.take<float>(ln ? wide::kLnSlot : 1);
    // LayerNorm: the interpolate rows' products run over 2B rows (primal stream | tangent stream) and carry a bias
    // row over the primal half
    const int nsA = wide_nsplit(B), nsD1 = wide_nsplit(2 * B), nsD2 = ln ? nsD1 : nsA;
    const int ns[9] = {nsA, nsA, nsA, nsA, nsA, nsD1, nsD2, nsD1, nsD2};
    const int64_t cnt[9] = {(H + 1) * H, (in + 1) * H, (H + 1) * ac, (H + 1) * H, (ob + 1) * H,
                            (H + 1) * H, (ln ? H + 1 : H) * H, (in + 1) * H, (ln ? in + 1 : in) * H};
    for (int p = 0; p < 9; ++p) {
        w.ns[p] = ns[p];
        w.cnt[p] = cnt[p];
        w.P[p] = ar.take<float>((size_t)ns[p] * cnt[p]);
    }
    w.w3a_tmp = ar.take<float>((H + 1) * ac);
    const int kin[5] = {ob, (int)in, (int)in, ob, (int)in};
    for (int n = 0; n < 5; ++n) {
        w.pk[n][0] = ar.take<float>((size_t)wide::packed_rows(kin[n]) * H);
        w.pk[n][1] = ar.take<float>((size_t)H * H);
    }
    for (int n = 0; n < 3; ++n) {
        w.pt[n][0] = ar.take<float>(n == 0 ? 1 : wide::timage_floats(n == 2 ? 2 * (int)H : (int)H));
        w.pt[n][1] = ar.take<float>(wide::timage_floats((int)H));
    }
    return w;
}
}  // namespace

size_t wide_reduce_ws_bytes(const samx_learner* L, int64_t B) {
    Arena ar(nullptr, 0);
    carve_wide_red(ar, L->ob_dim, L->ac_dim, B, L->critic.layernorm != 0);
    return align_up(ar.off, 256) + 256;
}

int wide_images(const samx_learner* L, int64_t B, void* ws_red, size_t red_bytes, float* img[5][2], float* row[3][2]) {
    Arena ar(ws_red, red_bytes);
    WideRed wr = carve_wide_red(ar, L->ob_dim, L->ac_dim, B, L->critic.layernorm != 0);
    if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "wide_images: reduction workspace too small");
    for (int n = 0; n < 5; ++n) { img[n][0] = wr.pk[n][0]; img[n][1] = wr.pk[n][1]; }
    for (int n = 0; n < 3; ++n) { row[n][0] = wr.pt[n][0]; row[n][1] = wr.pt[n][1]; }
    return SAMX_OK;
}

bool wide_step_supported(const samx_learner* L, int64_t B) {
    return wide::supported(&L->actor, &L->critic, &L->reward, L->ob_dim, L->ac_dim, B) &&
           L->d_actor_loss_scale == 0.f && B / wide::kRows <= kGpBlocks;
}

int wide_step_roles(const samx_learner* L, const samx_batch* b, const samx_step_out* o, void* ws_tc, void* ws_a,
                    size_t nets_bytes, void* ws_d, size_t disc_bytes, void* ws_red, size_t red_bytes, float* sumsq,
                    bool do_a, bool do_d, int part, hipStream_t st, const WideExt* ext) {
    const bool run = (part & 1) != 0 && !(ext && ext->targets_only);
    static_assert(kWdBlocks <= 256 && kGpBlocks <= 256, "one 256-thread workgroup adds the partial sums");
    const int64_t B = b->B;
    const samx_mlp *da = &L->actor, *dc = &L->critic, *dd = &L->reward;
    const int64_t Pa = samx_mlp_param_count(da), Pc = samx_mlp_param_count(dc), Pd = samx_mlp_param_count(dd);
    const int ob = L->ob_dim, ac = L->ac_dim, in = ob + ac, H = wide::kH;
    const int tiles = (int)(B / wide::kRows);
    const float gp_coeff = 10.f;  // discriminator.py:154
    const bool ln = dc->layernorm != 0;
    const int sw = wide::slot_width(ln);
    auto net = [](const Net& n) {
        wide::Net w{n.W1, n.b1, n.W2, n.b2, n.W3, n.b3, n.in, n.out};
        w.be1 = n.beta1; w.ga1 = n.gamma1; w.be2 = n.beta2; w.ga2 = n.gamma2;
        return w;
    };
    Arena ared(ws_red, red_bytes);
    WideRed wr = carve_wide_red(ared, ob, ac, B, ln);
    if (!ared.ok()) return fail(SAMX_ERR_WORKSPACE, "wide_step_roles: reduction workspace too small");
    GemmArgs gl[9];
    int ng = 0;
    auto prob = [&](int p, const float* A, int64_t lda, int M, const float* Bm, int64_t ldb, int N, int64_t R,
                    int64_t ones) {
        GemmArgs g;
        g.A = A; g.a_si = 1; g.a_sr = lda;
        g.B = Bm; g.b_sr = ldb; g.b_sj = 1;
        g.C = wr.P[p]; g.ldc = N;   // unused: only partials are written
        g.M = M; g.N = N; g.R = (int)R;
        g.a_ones_rows = (int)ones;
        g.nsplit = wr.ns[p];
        g.partials = wr.P[p];
        g.overread_ok = true;
        gl[ng++] = g;
    };
    WFinishArgs f;
    f.nseg = 0; f.gp_seg = -1; f.gp_part = nullptr; f.gp_tiles = tiles; f.ob = ob; f.ac = ac; f.B = B;
    f.gp_coeff = gp_coeff;
    f.rms_sums = nullptr; f.rms_addvec = nullptr; f.rms_mean = f.rms_std = nullptr;
    const float* pk_src[5][2] = {};
    {   // (all five, whatever the roles of this call: WideExt::pack_all)
        const Net s0 = split(da, L->params), s1 = split(dc, L->params + Pa), s2 = split(dd, L->params + Pa + Pc);
        const Net s3 = split(da, L->targets), s4 = split(dc, L->targets + Pa);
        const Net* sn[5] = {&s0, &s1, &s2, &s3, &s4};
        for (int q = 0; q < 5; ++q) { pk_src[q][0] = sn[q]->W1; pk_src[q][1] = sn[q]->W2; }
    }
    auto pack = [&](int mask) -> int {   // bit n: network n of [actor, critic, reward, target actor, target critic]
        if (ext && ext->images_current) return SAMX_OK;
        if (ext && ext->pack_all) mask = 0x1F;
        const float* W[16];
        float* dst[16];
        int K[16], J[16], n = 0;
        const int kin[5] = {ob, in, in, ob, in};
        for (int q = 0; q < 5; ++q)
            if (mask & (1 << q)) {
                W[n] = pk_src[q][0]; dst[n] = wr.pk[q][0]; J[n] = 0; K[n++] = kin[q];
                W[n] = pk_src[q][1]; dst[n] = wr.pk[q][1]; J[n] = 0; K[n++] = H;
                if (q < 3) {   // transposed images of the networks that run backward (the targets only run forward)
                    W[n] = pk_src[q][1]; dst[n] = wr.pt[q][1]; J[n] = H; K[n++] = H;
                    if (q == 1) {   // critic: rows ob .. ob + ac - 1 of W1 (dQ/da)
                        W[n] = pk_src[q][0] + (int64_t)ob * H; dst[n] = wr.pt[q][0]; J[n] = H; K[n++] = ac;
                    } else if (q == 2) {   // reward net: all of W1 (the gradient penalty's input gradient)
                        W[n] = pk_src[q][0]; dst[n] = wr.pt[q][0]; J[n] = 2 * H; K[n++] = in;
                    }
                }
            }
        return wide::launch_pack(W, dst, K, J, n, st);
    };
    wide::AgentArgs a;
    a.wd_part = nullptr; a.wd_blocks = 0;
    wide::DiscArgs d;
    AgentWs w1{}, w2{};
    DiscWs wdsc{};
    Net nc;
    const bool w3a_group = ac >= 4;
    auto piece = [](const float* p1, int n1, int64_t s1, int64_t count) {
        WPiece pc;
        pc.p1 = p1; pc.n1 = n1; pc.s1 = s1; pc.p2 = nullptr; pc.n2 = 0; pc.s2 = 0; pc.n2_valid = 0;
        pc.wdW = nullptr; pc.wd = 0.f; pc.wd_count = 0; pc.count = count;
        return pc;
    };
    auto seg = [&](int k, int64_t n, float* grad) -> WFinishSeg& {
        WFinishSeg& s = f.seg[f.nseg++];
        s.npc = 3;
        s.n = n; s.grad = grad; s.nblocks = fused::seg_blocks(n);
        s.sumsq = sumsq + k * fused::kMaxSegBlocks;
        s.step_dev = (part & 4) ? nullptr : L->steps + k;   // part bit 2 here: no counter tick
        return s;
    };
    if (do_a) {
        Arena a1(ws_tc, nets_bytes), a2(ws_a, nets_bytes);
        w1 = carve_agent(a1, da, dc, nullptr, B);
        w2 = carve_agent(a2, da, dc, nullptr, B);
        if (!a1.ok() || !a2.ok()) return fail(SAMX_ERR_WORKSPACE, "wide_step_roles: agent workspace too small");
        Net na = split(da, L->params);
        nc = split(dc, L->params + Pa);
        Net ta = split(da, L->targets), tc = split(dc, L->targets + Pa);
        a.obs0 = b->obs0; a.acs = b->acs; a.rews = b->rews; a.dones1 = b->dones1; a.obs1 = b->obs1;
        a.iws = b->iws; a.td_len = b->td_len; a.mean = L->mu_mean; a.stdv = L->mu_std;
        a.idx = b->a_idx; a.ring_start = b->ring_start; a.ring_len = b->ring_len;
        a.actor = net(na); a.critic = net(nc); a.targ_actor = net(ta); a.targ_critic = net(tc);
        a.actor.W1p = wr.pk[0][0]; a.actor.W2p = wr.pk[0][1];
        a.critic.W1p = wr.pk[1][0]; a.critic.W2p = wr.pk[1][1];
        a.targ_actor.W1p = wr.pk[3][0]; a.targ_actor.W2p = wr.pk[3][1];
        a.targ_critic.W1p = wr.pk[4][0]; a.targ_critic.W2p = wr.pk[4][1];
        a.actor.W1t = wr.pt[0][0]; a.actor.W2t = wr.pt[0][1];
        a.critic.W1t = wr.pt[1][0]; a.critic.W2t = wr.pt[1][1];
        a.ob = ob; a.ac = ac; a.B = B;
        a.high = L->ac_high; a.gamma = L->gamma; a.td_scale = L->td_loss_scale; a.q_scale = L->q_actor_loss_scale;
        a.targ_q = o->targ_q; a.td_out = o->td;
        // (with LayerNorm the layer inputs of the weight-gradient products are the LayerNorm outputs N1 / N2 — the
        // same buffers as A1 / A2 without it; the normalised activations travel through XH*, see fused_wide.h)
        a.Xc = w1.XC; a.A1c = w1.hc.N1; a.A2c = w1.hc.N2; a.DZ1c = w1.kc.DZ1; a.DZ2c = w1.kc.DZ2;
        a.row_hub = w1.row0; a.row_dq = w1.row1; a.w3c_part = wr.w3c;
        a.Xa = w2.XC; a.A1a = w2.ha.N1; a.A2a = w2.ha.N2; a.DZ1a = w2.ka.DZ1; a.DZ2a = w2.ka.DZ2;
        if (ln) {
            a.XH1c = w1.hc.XH1; a.XH2c = w1.hc.XH2;
            a.XH1a = w2.ha.XH1; a.XH2a = w2.ha.XH2; a.XH1q = w2.hc.XH1; a.XH2q = w2.hc.XH2;
            a.lna_part = wr.lna;
        }
        a.dPre = w2.dPre; a.row_q = w2.row0;
        a.ret_mean = L->ret_mean; a.ret_std = L->ret_std; a.tdn_scale = L->td_loss_n_scale;
        a.ext_tq = a.ext_tqn = nullptr; a.n_idx = nullptr; a.n_rews = a.n_dones = a.n_len = nullptr;
        a.targ_qn = nullptr; a.row_hubn = nullptr;
        if (ext) {
            a.n_idx = ext->n_idx; a.n_rews = ext->n_rews; a.n_dones = ext->n_dones; a.n_len = ext->n_len;
            a.targ_qn = ext->tqn;
            if (ext->targets_only) {
                TRY(pack(0x1B | (do_d ? 0x04 : 0)));   // the target networks, and what the role launch will need
                return wide::launch_targets(&a, st);
            }
            if (ext->use_ext) {
                a.ext_tq = o->targ_q; a.ext_tqn = ext->tqn;
                if (ext->tqn) a.row_hubn = w1.row2;
            }
        }
        // weight-gradient products: critic W2, W1 ; actor W3, W2, W1
        prob(0, w1.hc.N1, H, H, w1.kc.DZ2, H, H, B, B);
        prob(1, w1.XC, in, in, w1.kc.DZ1, H, H, B, B);
        if (w3a_group) prob(2, w2.ha.N2, H, H, w2.dPre, ac, ac, B, B);
        prob(3, w2.ha.N1, H, H, w2.ka.DZ2, H, H, B, B);
        prob(4, w2.XC, in, ob, w2.ka.DZ1, H, H, B, B);
        // flat layout per network: [W1; b1] ([beta1; gamma1]) [W2; b2] ([beta2; gamma2]) [W3; b3]
        WFinishSeg& sa = seg(0, Pa, L->grads);
        WFinishSeg& sc = seg(1, Pc, L->grads + Pa);
        const WPiece aW1 = piece(wr.P[4], wr.ns[4], wr.cnt[4], wr.cnt[4]);
        const WPiece aW2 = piece(wr.P[3], wr.ns[3], wr.cnt[3], wr.cnt[3]);
        const WPiece aW3 = w3a_group ? piece(wr.P[2], wr.ns[2], wr.cnt[2], wr.cnt[2])
                                     : piece(wr.w3a_tmp, 1, wr.cnt[2], wr.cnt[2]);
        WPiece cW1 = piece(wr.P[1], wr.ns[1], wr.cnt[1], wr.cnt[1]);
        cW1.wdW = nc.W1; cW1.wd = L->wd_scale; cW1.wd_count = (int64_t)in * H;
        WPiece cW2 = piece(wr.P[0], wr.ns[0], wr.cnt[0], wr.cnt[0]);
        cW2.wdW = nc.W2; cW2.wd = L->wd_scale; cW2.wd_count = (int64_t)H * H;
        if (L->wd_scale <= 0.f) { cW1.wdW = nullptr; cW2.wdW = nullptr; }
        const WPiece cW3 = piece(wr.w3_sum, 1, H + 1, H + 1);
        if (ln) {
            sa.npc = sc.npc = 5;
            sa.pc[0] = aW1; sa.pc[1] = piece(wr.lna_sum, 1, 2 * H, 2 * H); sa.pc[2] = aW2;
            sa.pc[3] = piece(wr.lna_sum + 2 * H, 1, 2 * H, 2 * H); sa.pc[4] = aW3;
            sc.pc[0] = cW1; sc.pc[1] = piece(wr.w3_sum + (H + 1), 1, 2 * H, 2 * H); sc.pc[2] = cW2;
            sc.pc[3] = piece(wr.w3_sum + (H + 1) + 2 * H, 1, 2 * H, 2 * H); sc.pc[4] = cW3;
        } else {
            sa.pc[0] = aW1; sa.pc[1] = aW2; sa.pc[2] = aW3;
            sc.pc[0] = cW1; sc.pc[1] = cW2; sc.pc[2] = cW3;
        }
    }
    if (do_d) {
        Arena ar(ws_d, disc_bytes);
        wdsc = carve_disc(ar, dd, B);
        DiscWs& w = wdsc;
        if (!ar.ok()) return fail(SAMX_ERR_WORKSPACE, "wide_step_roles: discriminator workspace too small");
        Net nd = split(dd, L->params + Pa + Pc);
        d.p_obs = b->p_obs; d.p_acs = b->p_acs; d.e_obs = b->e_obs; d.e_acs = b->e_acs;
        d.eps_ob = b->eps_ob; d.eps_ac = b->eps_ac; d.real = b->real_labels; d.fake = b->fake_labels;
        d.mean = L->d_mean; d.stdv = L->d_std;
        d.rms_sums = b->d_rms_sums; d.rms_addvec = b->d_rms_addvec;
        f.rms_sums = b->d_rms_sums; f.rms_addvec = b->d_rms_addvec;
        f.rms_mean = const_cast<float*>(L->d_mean); f.rms_std = const_cast<float*>(L->d_std);
        d.p_idx = b->p_idx; d.e_start = b->e_start; d.ring_start = b->ring_start; d.ring_len = b->ring_len;
        d.rng_seed = b->rng_seed; d.rng_step = L->steps + 2; d.label_smoothing = b->label_smoothing;
        d.net = net(nd); d.ob = ob; d.ac = ac; d.B = B; d.ent_scale = L->ent_reg_scale;
        d.net.W1p = wr.pk[2][0]; d.net.W2p = wr.pk[2][1];
        d.net.W1t = wr.pt[2][0]; d.net.W2t = wr.pt[2][1];
        d.X = w.X; d.A1 = w.h.N1; d.A2 = w.h.N2; d.DZ1 = w.k.DZ1; d.DZ2 = w.k.DZ2;
        d.DS = w.DS; d.ce = w.ce; d.ent = w.ent; d.sg = w.sg; d.gp_part = w.gp_part; d.w3_part = wr.w3d;
        if (ln) { d.XH1 = w.h.XH1; d.XH2 = w.h.XH2; d.TN1 = w.TA1; d.PA1 = w.TA2; }
        // D W2 / W1: policy + expert rows (with the bias row), then the interpolate rows (take the scalar k).
        // LayerNorm: the interpolate rows are 2B rows [primal stream | tangent stream] — n1^T P_z2 + ndot1^T T_z2 and
        // x^T P_z1 + V^T T_z1 — with the bias row over the primal half (db = sum P_z)
        const int64_t RI = ln ? 2 * B : B, onesI = ln ? B : -1;
        prob(5, w.h.N1, H, H, w.k.DZ2, H, H, 2 * B, 2 * B);
        prob(6, w.h.N1 + 2 * B * H, H, H, w.k.DZ2 + 2 * B * H, H, H, RI, onesI);
        prob(7, w.X, in, in, w.k.DZ1, H, H, 2 * B, 2 * B);
        prob(8, w.X + 2 * B * in, in, in, w.k.DZ1 + 2 * B * H, H, H, RI, onesI);
        WFinishSeg& sd = seg(2, Pd, L->grads + Pa + Pc);
        WPiece dW1 = piece(wr.P[7], wr.ns[7], wr.cnt[7], wr.cnt[7]);
        dW1.p2 = wr.P[8]; dW1.n2 = wr.ns[8]; dW1.s2 = wr.cnt[8]; dW1.n2_valid = wr.cnt[8];
        WPiece dW2 = piece(wr.P[5], wr.ns[5], wr.cnt[5], wr.cnt[5]);
        dW2.p2 = wr.P[6]; dW2.n2 = wr.ns[6]; dW2.s2 = wr.cnt[6]; dW2.n2_valid = wr.cnt[6];
        const float* sumPE = wr.w3_sum + sw;        // policy + expert tiles
        const float* sumI = wr.w3_sum + 2 * sw;     // interpolate tiles (take k)
        WPiece dW3 = piece(sumPE, 1, H + 1, H + 1);
        dW3.p2 = sumI; dW3.n2 = 1; dW3.s2 = H + 1; dW3.n2_valid = H;
        if (ln) {
            sd.npc = 5;
            WPiece l1 = piece(sumPE + (H + 1), 1, 2 * H, 2 * H), l2 = piece(sumPE + (H + 1) + 2 * H, 1, 2 * H, 2 * H);
            l1.p2 = sumI + (H + 1); l1.n2 = 1; l1.s2 = 2 * H; l1.n2_valid = 2 * H;
            l2.p2 = sumI + (H + 1) + 2 * H; l2.n2 = 1; l2.s2 = 2 * H; l2.n2_valid = 2 * H;
            sd.pc[0] = dW1; sd.pc[1] = l1; sd.pc[2] = dW2; sd.pc[3] = l2; sd.pc[4] = dW3;
        } else {
            sd.pc[0] = dW1; sd.pc[1] = dW2; sd.pc[2] = dW3;
        }
        f.gp_seg = f.nseg - 1; f.gp_part = w.gp_part;
    }
    if (run) {
        // the forward products read k-interleaved images of W1 / W2: written here, from the parameter vectors as
        // they are NOW (scratch of this call; with stored targets the target networks' images are not needed)
        if (!(ext && ext->packed))
            TRY(pack((do_a ? (0x03 | ((ext && ext->use_ext) ? 0 : 0x18)) : 0) | (do_d ? 0x04 : 0)));
        // every role in ONE launch (the decayed kernels' sums of squares ride along), then ONE for all
        // weight-gradient products and the head-gradient slot sums
        if (do_a && L->wd_scale > 0.f) { a.wd_part = w1.wd_part; a.wd_blocks = kWdBlocks; }
        TRY(wide::launch_roles(do_a ? &a : nullptr, do_d ? &d : nullptr, (do_a ? 3 : 0) | (do_d ? 4 : 0), st));
        if (do_a && !w3a_group)   // an action head narrower than 16 bytes: the plain split product
            TRY(wgrad(w2.ha.N2, H, H, w2.dPre, ac, ac, B, B, wr.w3a_tmp, 0.f, nullptr, w2.partials,
                      w2.partial_floats, st));
        SlotSum sl[kSlotSumMax];
        int nsl = 0;
        if (do_a) sl[nsl++] = SlotSum{wr.w3c, tiles, sw, wr.w3_sum};
        if (do_a && ln) sl[nsl++] = SlotSum{wr.lna, tiles, wide::kLnSlot, wr.lna_sum};
        if (do_d) {
            sl[nsl++] = SlotSum{wr.w3d + (int64_t)tiles * sw, 2 * tiles, sw, wr.w3_sum + sw};
            sl[nsl++] = SlotSum{wr.w3d, tiles, sw, wr.w3_sum + 2 * sw};
        }
        TRY(gemm_group_partials(gl, ng, st, sl, nsl));
    }
    if (part & 2) {   // the reduction, the statistics commit, the counters, the scalar losses
        WLossArgs& t = f.loss;
        t.B = B;
        t.row_hub = do_a ? w1.row0 : nullptr;
        t.row_hubn = (do_a && ext && ext->use_ext && ext->tqn) ? w1.row2 : nullptr;
        t.tdn_scale = L->td_loss_n_scale; t.ret_mean = L->ret_mean; t.ret_std = L->ret_std;
        t.wd_part = w1.wd_part; t.wd_blocks = kWdBlocks;
        t.td_scale = L->td_loss_scale; t.wd_scale = L->wd_scale; t.critic_losses = o->critic_losses;
        t.row_q = do_a ? w2.row0 : nullptr; t.q_scale = L->q_actor_loss_scale; t.actor_losses = o->actor_losses;
        t.ce = do_d ? wdsc.ce : nullptr; t.ent = wdsc.ent; t.sg = wdsc.sg; t.gp_part = wdsc.gp_part;
        t.tiles = tiles; t.ob = ob; t.ac = ac; t.ent_scale = L->ent_reg_scale; t.gp_coeff = gp_coeff;
        t.d_losses = o->d_losses;
        int blocks = 0;
        for (int k = 0; k < f.nseg; ++k) blocks += f.seg[k].nblocks;
        f.main_blocks = blocks;
        wide_finish_kernel<<<blocks + 3, 256, 0, st>>>(f);
        SAMX_LAUNCH_CHECK();
    }
    return SAMX_OK;
}

}  // namespace samx
