// Role-fused forward + input-gradient kernel for 256x256 networks (fused_wide.hip): one workgroup runs a
// role's whole chain of layer products for 16 batch rows with the activations in LDS and the weights
// loaded by each wave straight from L2 into MFMA registers (k-interleaved images); what the weight-gradient
// products need (layer inputs and pre-activation cotangents) is written to HBM, where ONE grouped launch
// reduces it over the batch.
#pragma once
#include "samx_common.h"
#include "../../include/samx.h"

namespace samx {
namespace wide {

constexpr int kH = 256;     // hidden width (both layers)
constexpr int kRows = 16;   // batch rows per workgroup
constexpr int kMaxIn = 416; // widest network input (a multiple of 32)
// floats of a transposed image with J column slots per k-group (Net::W1t / W2t)
inline size_t timage_floats(int J) { return (size_t)(kH / 4) * J * 4; }

struct Net {   // leaky-ReLU net, h1 = h2 = 256 (device pointers into a flat parameter vector)
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int in, out;
    // k-interleaved images of W1 / W2 for the forward products (launch_pack writes them from W1 / W2 at the head of
    // every launch sequence: element (k, j) at ((k >> 2) * 256 + j) * 4 + (k & 3), packed_rows(in) / 256 rows)
    const float *W1p = nullptr, *W2p = nullptr;
    // images for the backward products (B[k][j] = W[j][k]): the k-interleaved image of the TRANSPOSE, element
    // (row j, column k) of W at ((k >> 2) * J + j) * 4 + (k & 3) — the same loader as the forward products, 16 lanes
    // reading 256 contiguous bytes.  (Straight from the parameter vector the 16 lanes of a quarter wave read 16
    // different rows: one cache line per lane, 64 tag lookups per wave-wide load instead of 16 — 1.0 us per stage
    // against 0.65.)  W2t: J = 256, all of W2.  W1t: what the role needs — reward net: J = 512, all rows of W1
    // (the input gradient of the gradient penalty); critic: J = 256, rows ob .. ob + ac - 1 (dQ/da); actor: none.
    const float *W1t = nullptr, *W2t = nullptr;
    // LayerNorm behind each hidden activation (networks.py:53-70, network_util.py:32-39), null without it; all
    // networks of a learner carry it or none does (hps.with_layernorm)
    const float *be1 = nullptr, *ga1 = nullptr, *be2 = nullptr, *ga2 = nullptr;
};

// Per-tile gradient slot of a network with LayerNorm: [W3 (256) | b3 | beta1 | gamma1 | beta2 | gamma2]; without
// it only the first kH + 1 floats exist.  (The actor's slot has no head part: kLnSlot floats, beta1 first.)
constexpr int kLnSlot = 4 * 256;
inline int slot_width(bool ln) { return 256 + 1 + (ln ? kLnSlot : 0); }

struct AgentArgs {
    const float *obs0, *acs, *rews, *dones1, *obs1, *iws, *td_len, *mean, *stdv;
    const int64_t* idx;   // optional ring indirection (memory.py:33-34)
    int64_t ring_start, ring_len;
    Net actor, critic, targ_actor, targ_critic;
    int ob, ac;
    int64_t B;
    float high, gamma, td_scale, q_scale;
    // target + critic role: TD targets / errors, and the critic's weight-gradient operands
    float *targ_q, *td_out;
    float *Xc, *A1c, *A2c, *DZ1c, *DZ2c;   // [B][ob+ac], [B][256] x 4
    float *row_hub, *row_dq;               // [B] huber * iw ; d loss / d score
    float* w3c_part;                       // [B/16][slot_width]: per-tile A2c^T dq (256) and sum dq: the head's
                                           // gradient [+ the critic's LayerNorm gradients of the tile's rows]
    // LayerNorm only: normalised activations x^ [B][256] per hidden layer, written by the forward and read back BY
    // THE SAME THREAD in the backward (critic of the target+critic role; actor and critic of the actor role), and
    // the actor's per-tile LayerNorm gradient slots [B/16][kLnSlot].  With LayerNorm A1* / A2* hold the LayerNorm
    // OUTPUTS (what the next layer reads: the weight-gradient products' operands).
    float *XH1c = nullptr, *XH2c = nullptr, *XH1a = nullptr, *XH2a = nullptr, *XH1q = nullptr, *XH2q = nullptr;
    float* lna_part = nullptr;
    // actor role
    float *Xa, *A1a, *A2a, *DZ1a, *DZ2a;   // [B][ob+ac] (obz0 | mu(obz0)), [B][256] x 4
    float *dPre, *row_q;                   // [B][ac] ; [B] Q(obz0, mu(obz0))
    // Return normalisation / PopArt / n-step returns (sam_agent.py:161-201,483-498,651-672): the statistics and
    // the critic heads change between the target pass and the loss pass, so the targets then come from their
    // own launch (launch_targets) and the roles read them back:
    //   ret_mean / ret_std: device scalars or null;  ext_tq / ext_tqn: targets computed ahead (null = the
    //   target+critic role computes the 1-step targets itself);  tdn_scale: weight of the n-step TD loss;
    //   n_idx / n_rews / n_dones / n_len: the n-step lookahead of the sampled rows (launch_targets only);
    //   targ_qn: [B] out of launch_targets;  row_hubn: [B] huber(n-step TD) * iw
    const float *ret_mean, *ret_std, *ext_tq, *ext_tqn;
    float tdn_scale;
    const int64_t* n_idx;
    const float *n_rews, *n_dones, *n_len;
    float *targ_qn, *row_hubn;
    // weight decay (network_util.py:43-56): wd_blocks extra workgroups leave float64 partial sums of squares of
    // the critic's two hidden kernels in wd_part[wd_blocks] (0 = off)
    double* wd_part;
    int wd_blocks;
};

struct DiscArgs {
    const float *p_obs, *p_acs, *e_obs, *e_acs, *eps_ob, *eps_ac, *real, *fake, *mean, *stdv;
    // rms_sums != null: the rows are normalised with the statistics of rms_sums + rms_addvec (double[2 ob + 1]
    // each: the running sums and this batch's, helpers/mpi_running_mean_std.py:31-45) instead of mean / stdv
    const double *rms_sums, *rms_addvec;
    const int64_t *p_idx, *e_start;
    int64_t ring_start, ring_len;
    uint64_t rng_seed;
    const int64_t* rng_step;
    int label_smoothing;
    Net net;
    int ob, ac;
    int64_t B;
    float ent_scale;
    // rows [0,B) policy, [B,2B) expert, [2B,3B) interpolates (for those: X = V, A1/A2 = the tangents)
    // With LayerNorm the interpolates take two row blocks: [2B,3B) the primal stream P (X = the
    // interpolated inputs, A1 = n1, DZ* = P_z*), [3B,4B) the tangent stream T (X = V, A1 = ndot1, DZ* = T_z*); A2 is
    // not used for them.  XH1 / XH2 [3B][256]: normalised activations (same-thread read-back); TN1, PA1 [B][256]:
    // the tangent stream's cotangent of n1 and the primal cotangent the first tangent LayerNorm sends back.
    float *X, *A1, *A2, *DZ1, *DZ2;   // [3B or 4B][ob+ac], [3B or 4B][256] x 4
    float *XH1 = nullptr, *XH2 = nullptr, *TN1 = nullptr, *PA1 = nullptr;
    float *DS, *ce, *ent, *sg;        // [3B] ; [2B] x 3
    double* gp_part;                  // [B/16][2]
    // per-tile head gradient [3B/16][slot_width], tiles in launch order: interpolate (sum a2dot, 0), policy, expert
    float* w3_part;
};

bool supported(const samx_mlp* actor, const samx_mlp* critic, const samx_mlp* reward, int ob, int ac, int64_t B);
// roles: bit 0 actor, bit 1 target+critic, bit 2 discriminator
int launch_roles(const AgentArgs* a, const DiscArgs* d, int roles, hipStream_t st);
// rows of the k-interleaved image of a [K][256] matrix (whole 32-deep stages; the rows beyond K are zero)
inline int packed_rows(int K) { return (K + 31) / 32 * 32; }
// images of up to 16 matrices in ONE launch, from W[i] ([K[i]][256]): J[i] = 0: the k-interleaved image
// (dst[i]: packed_rows(K[i]) * 256 floats); J[i] > 0: the transposed image with J[i] column slots (dst[i]:
// timage_floats(J[i]) floats, slots 0 .. K[i] - 1 written)
int launch_pack(const float* const* W, float* const* dst, const int* K, const int* J, int n, hipStream_t st);
// TD targets only: the 1-step targets of the sampled rows into a->targ_q and, with a->n_idx, the n-step targets
// into a->targ_qn (de-normalised with the CURRENT return statistics, sam_agent.py:196-205)
int launch_targets(const AgentArgs* a, hipStream_t st);

}  // namespace wide
}  // namespace samx
