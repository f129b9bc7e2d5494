// Device helpers shared by the whole-network fused kernels (fused_small.hip: plain leaky-ReLU nets;
// fused_ln.hip: nets with post-activation LayerNorm): LDS layout constants, MFMA tile products, the staged
// prologue (weights / heads / batch rows), layer phases and per-tile weight gradients.
// Included inside an anonymous namespace by each translation unit.
#pragma once
#include "samx_common.h"
#include "fused_small.h"
#include "rng.h"
#include <math.h>
#include <stdint.h>

namespace samx {
namespace fused {
namespace {

constexpr int H = 64;     // hidden width (both layers)
constexpr int LD = 81;    // LDS pitch of 64-wide matrices (== 17 mod 32)
constexpr int LD3 = 17;   // pitch of the head matrix W3[64][out<=16]
constexpr int R = 16;     // batch rows per workgroup
constexpr int NT = 256;   // threads per workgroup
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Phase timestamps of workgroup 0 of each role (profiling builds only: -DSAMX_TRACE; never shipped)
#ifdef SAMX_TRACE
__device__ unsigned long long g_trace[64];
#define TRACEW(w_, role, i) do { if ((w_) == 0 && threadIdx.x == 0) g_trace[(role) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define TRACEW(w_, role, i) do { } while (0)
#endif
#define TRACE(role, i) TRACEW(wg, role, i)

__device__ __forceinline__ float lrelu(float z) { return fmaxf(z, 0.2f * z); }
__device__ __forceinline__ float dlrelu(float a) { return a >= 0.f ? 1.f : 0.2f; }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// accumulator element j of lane l  <->  C[row = (l>>4)*4 + j][col = l & 15]

// C = A[16 x K4] * W[K4 x 16 cols from col0]   (A row-major pitch lda, W row-major pitch ldw)
__device__ __forceinline__ f32x4 tile_nn(const float* A, int lda, const float* W, int ldw, int K4,
                                         int col0, int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    int k = 0;
    for (; k + 8 <= K4; k += 8) {
        a0 = mfma4(A[r * lda + k + kq], W[(k + kq) * ldw + col0 + r], a0);
        a1 = mfma4(A[r * lda + k + 4 + kq], W[(k + 4 + kq) * ldw + col0 + r], a1);
    }
    for (; k < K4; k += 4) a0 = mfma4(A[r * lda + k + kq], W[(k + kq) * ldw + col0 + r], a0);
    return a0 + a1;
}

// C = A[16 x K4] * W^T, W rows row0..row0+15 (pitch ldw): C[r][c] = sum_k A[r][k] W[row0 + c][k]
__device__ __forceinline__ f32x4 tile_nt(const float* A, int lda, const float* W, int ldw, int K4,
                                         int row0, int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    int k = 0;
    for (; k + 8 <= K4; k += 8) {
        a0 = mfma4(A[r * lda + k + kq], W[(row0 + r) * ldw + k + kq], a0);
        a1 = mfma4(A[r * lda + k + 4 + kq], W[(row0 + r) * ldw + k + 4 + kq], a1);
    }
    for (; k < K4; k += 4) a0 = mfma4(A[r * lda + k + kq], W[(row0 + r) * ldw + k + kq], a0);
    return a0 + a1;
}

// K = 64 forms, fully unrolled: all 32 LDS reads are issued ahead of the 16 MFMAs
__device__ __forceinline__ f32x4 tile_nn64(const float* A, const float* W, int col0, int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const float* pa = A + (lane & 15) * LD + (lane >> 4);
    const float* pw = W + (lane >> 4) * LD + col0 + (lane & 15);
#pragma unroll
    for (int k = 0; k < H; k += 8) {
        a0 = mfma4(pa[k], pw[k * LD], a0);
        a1 = mfma4(pa[k + 4], pw[(k + 4) * LD], a1);
    }
    return a0 + a1;
}
__device__ __forceinline__ f32x4 tile_nt64(const float* A, const float* W, int row0, int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    const float* pa = A + (lane & 15) * LD + (lane >> 4);
    const float* pw = W + (row0 + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
    for (int k = 0; k < H; k += 8) {
        a0 = mfma4(pa[k], pw[k], a0);
        a1 = mfma4(pa[k + 4], pw[k + 4], a1);
    }
    return a0 + a1;
}

// C[i][j] = sum_{r<16} X[r][i0 + i] * D[r][j0 + j]   (weight gradient of one 16x16 block)
__device__ __forceinline__ f32x4 tile_tn(const float* X, int ldx, int i0, const float* D, int ldd, int j0,
                                         int lane) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int k = 0; k < R; k += 4) a0 = mfma4(X[(k + kq) * ldx + i0 + c], D[(k + kq) * ldd + j0 + c], a0);
    return a0;
}

struct NetL {  // one network's parameters in LDS
    float *W1, *b1, *W2, *b2, *W3, *b3;
    float *be1, *ga1, *be2, *ga2;   // LayerNorm beta / gamma of the two hidden layers (ln nets only)
    int in, k4, out;
};

__host__ __device__ inline int net_lds_floats(int in, int out, bool ln = false) {
    const int k4 = (in + 3) / 4 * 4;
    return k4 * LD + H + H * LD + H + H * LD3 + 16 + (out > 0 ? 0 : 0) + (ln ? 4 * H : 0);
}

__device__ inline NetL carve_net(float*& p, int in, int out, bool ln = false) {
    NetL n;
    n.in = in; n.k4 = (in + 3) / 4 * 4; n.out = out;
    n.W1 = p; p += n.k4 * LD;
    n.b1 = p; p += H;
    n.W2 = p; p += H * LD;
    n.b2 = p; p += H;
    n.W3 = p; p += H * LD3;
    n.b3 = p; p += 16;
    n.be1 = n.ga1 = n.be2 = n.ga2 = nullptr;
    if (ln) {
        n.be1 = p; p += H;
        n.ga1 = p; p += H;
        n.be2 = p; p += H;
        n.ga2 = p; p += H;
    }
    return n;
}

// ---- staged prologue -----------------------------------------------------------------------------
// Every global load of a role is ISSUED before anything waits, as straight-line code so that the
// compiler's counted s_waitcnt lets later loads stay in flight: ring indices first, then the weights
// (16-byte loads into registers), the head matrices and the per-row scalars; the rows themselves go
// out as soon as the indices are back; registers are written to their padded LDS images (…_commit)
// in arrival order.  The only dependent pair is  ring index -> row.
// (An earlier version staged the weights with LDS-DMA, one 256-byte row per instruction: ~63
// instructions per wave whose issue alone took 3.4 us of a 15 us role.)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// [W1 (in x 64) | b1 | W2 (64 x 64) | b2] is one contiguous run of (in + 66) rows of 64 floats;
// with LayerNorm [W1 | b1 | beta1 | gamma1 | W2 | b2 | beta2 | gamma2]: (in + 70) rows
constexpr int kNetQ = 9;  // 16-byte loads per thread: (64 + 70) * 16 / 256, rounded up
struct NetRegs { f32x4 q[kNetQ]; };

template <bool LN = false>
__device__ __forceinline__ void net_issue(NetRegs& x, const NetL& n, const float* __restrict__ g) {
    const int total = (n.in + (LN ? 70 : 66)) * 16;
#pragma unroll
    for (int j = 0; j < kNetQ; ++j) {
        int q = threadIdx.x + NT * j;
        q = q < total ? q : total - 1;   // clamped, not predicated: the loads stay straight-line
        x.q[j] = *(const f32x4u*)(g + 4 * q);
    }
}

template <bool LN = false>
__device__ __forceinline__ void net_commit(const NetRegs& x, const NetL& n) {
    const int total = (n.in + (LN ? 70 : 66)) * 16;
#pragma unroll
    for (int j = 0; j < kNetQ; ++j) {
        const int q = threadIdx.x + NT * j;
        if (q < total) {
            const int rho = q >> 4, c = (q & 15) * 4;
            float* dst;
            if (LN) {
                const int t = rho - n.in;   // rows behind W1: b1 beta1 gamma1 | W2 (64) | b2 beta2 gamma2
                dst = t < 0 ? n.W1 + rho * LD
                    : t == 0 ? n.b1 : t == 1 ? n.be1 : t == 2 ? n.ga1
                    : t < 3 + H ? n.W2 + (t - 3) * LD
                    : t == 3 + H ? n.b2 : t == 4 + H ? n.be2 : n.ga2;
            } else {
                dst = rho < n.in ? n.W1 + rho * LD
                    : rho == n.in ? n.b1
                    : rho <= n.in + H ? n.W2 + (rho - n.in - 1) * LD : n.b2;
            }
            dst[c] = x.q[j][0]; dst[c + 1] = x.q[j][1]; dst[c + 2] = x.q[j][2]; dst[c + 3] = x.q[j][3];
        }
    }
    for (int t = threadIdx.x; t < (n.k4 - n.in) * H; t += NT) n.W1[(n.in + (t >> 6)) * LD + (t & 63)] = 0.f;
}

// head matrix W3[64][out <= 16] -> padded [64][LD3] image.  Thread t owns row t >> 2 and the columns
// (t & 3) + 4 j: no integer division, and the zero padding up to 16 columns comes with it.
struct HeadRegs { float w[4]; float b; };

template <bool LN = false>
__device__ __forceinline__ HeadRegs head_issue(const NetL& n, const float* __restrict__ g) {
    const float* gW3 = g + (int64_t)n.in * H + H + H * H + H + (LN ? 4 * H : 0);
    const float* gb3 = gW3 + H * n.out;
    const int row = threadIdx.x >> 2, c0 = threadIdx.x & 3;
    HeadRegs h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 4 * j;
        h.w[j] = gW3[row * n.out + (c < n.out ? c : 0)];   // clamped: loads stay straight-line
    }
    h.b = gb3[(int)threadIdx.x < n.out ? threadIdx.x : 0];
    return h;
}

__device__ __forceinline__ void head_commit(const NetL& n, const HeadRegs& h) {
    const int row = threadIdx.x >> 2, c0 = threadIdx.x & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 4 * j;
        n.W3[row * LD3 + c] = c < n.out ? h.w[j] : 0.f;
    }
    if (threadIdx.x < 16) n.b3[threadIdx.x] = (int)threadIdx.x < n.out ? h.b : 0.f;
}

__device__ __forceinline__ void stage_wait() { __syncthreads(); }

// out[16][64] = lrelu(in[16][K4] * W + b)          (all 4 waves, one 16-col tile each)
__device__ inline void layer_fwd(const float* in, int K4, const float* W, const float* b, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc = K4 == H ? tile_nn64(in, W, 16 * w, lane) : tile_nn(in, LD, W, LD, K4, 16 * w, lane);
    const int c = 16 * w + (lane & 15), r0 = (lane >> 4) * 4;
    const float bb = b[c];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[(r0 + j) * LD + c] = lrelu(acc[j] + bb);
}

// tangent / masked linear map: out = (in * W) .* lrelu'(mask_src)   (no bias); out may alias mask_src
__device__ inline void layer_tan(const float* in, int K4, const float* W, const float* mask_src, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc = K4 == H ? tile_nn64(in, W, 16 * w, lane) : tile_nn(in, LD, W, LD, K4, 16 * w, lane);
    const int c = 16 * w + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[(r0 + j) * LD + c] = acc[j] * dlrelu(mask_src[(r0 + j) * LD + c]);
}

// dZ1[16][64] = (dZ2[16][64] * W2^T) .* lrelu'(A1)
__device__ inline void layer_bwd(const float* dZ2, const float* W2, const float* A1, float* dZ1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 acc = tile_nt64(dZ2, W2, 16 * w, lane);
    const int c = 16 * w + (lane & 15), r0 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) dZ1[(r0 + j) * LD + c] = acc[j] * dlrelu(A1[(r0 + j) * LD + c]);
}

// s[r] = <A2[r], W3[:,0]> + b3[0] for the 16 rows; every thread gets nothing back, result in sh_s[16]
__device__ inline void head_dot(const float* A2, const NetL& n, float* sh_s) {
    const int tid = threadIdx.x;
    const int r = tid >> 4, q = tid & 15;  // 16 threads per row
    float acc = 0.f;
#pragma unroll
    for (int c = q; c < H; c += 16) acc = fmaf(A2[r * LD + c], n.W3[c * LD3], acc);
    acc += __shfl_xor(acc, 8, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 1, 64);
    if (q == 0) sh_s[r] = acc + n.b3[0];
}

// dZ2[r][c] = dq[r] * w3[c] * lrelu'(A2[r][c])
__device__ inline void head_bwd(const float* dq, const NetL& n, const float* A2, float* dZ2) {
    for (int t = threadIdx.x; t < R * H; t += NT) {
        const int r = t >> 6, c = t & 63;
        dZ2[r * LD + c] = dq[r] * n.W3[c * LD3] * dlrelu(A2[r * LD + c]);
    }
}

// partial weight gradient of a [nI x 64] matrix + its bias: dst[i*64 + j], dst_b[j]
__device__ inline void wgrad64(const float* X, int nI, const float* dZ, float* __restrict__ dstW,
                               float* __restrict__ dstb, int ones_rows) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ti = (nI + 15) / 16;
    for (int t = w; t < ti * 4; t += 4) {
        const int i0 = (t >> 2) * 16, j0 = (t & 3) * 16;
        f32x4 acc = tile_tn(X, LD, i0, dZ, LD, j0, lane);
        const int c = j0 + (lane & 15), r0 = i0 + (lane >> 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (r0 + j < nI) dstW[(r0 + j) * H + c] = acc[j];
    }
    if (dstb && threadIdx.x < H) {
        float s = 0.f;
        for (int r = 0; r < ones_rows; ++r) s += dZ[r * LD + threadIdx.x];
        dstb[threadIdx.x] = s;
    }
}

// head of a 1-output net: dW3[c] = sum_r A2[r][c] * dq[r], db3 = sum_{r < ones_rows} dq[r]
__device__ inline void wgrad_head1(const float* A2, const float* dq, float* __restrict__ dstW,
                                   float* __restrict__ dstb, int ones_rows) {
    const int tid = threadIdx.x;
    if (tid < H) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) s = fmaf(A2[r * LD + tid], dq[r], s);
        dstW[tid] = s;
    }
    if (tid == H) {
        float s = 0.f;
        for (int r = 0; r < ones_rows; ++r) s += dq[r];
        dstb[0] = s;
    }
}

__device__ __forceinline__ int64_t ring_row(const int64_t* idx, int64_t start, int64_t len, int64_t g) {
    if (!idx) return g;
    int64_t s = start + idx[g];  // 0 <= idx < len and 0 <= start < len (host-checked)
    return s >= len ? s - len : s;
}

// Rows of the batch (w <= 64 columns): thread t owns row t >> 4 and the columns (t & 15) + 16 j —
// one ring-index load per thread and no integer division.
struct RowRegs { float v[4], m[4], s[4]; };

// raw values (+ the column statistics they will be normalised with): loads only.  pr = physical row.
__device__ __forceinline__ void rows_issue(RowRegs& x, int64_t pr, const float* __restrict__ obs,
                                           const float* __restrict__ acs, const float* __restrict__ mean,
                                           const float* __restrict__ stdv, int ob, int ac, int w) {
    const int c0 = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        x.v[j] = 0.f; x.m[j] = 0.f; x.s[j] = 1.f;
        if (c < ob) {
            x.v[j] = obs[pr * ob + c];
            if (mean) { x.m[j] = mean[c]; x.s[j] = stdv[c]; }
        } else if (c < w) {
            x.v[j] = acs[pr * ac + (c - ob)];
        }
    }
}

// normalised observation columns (+ action columns as they are) into XC[16][LD]; columns [w, 64) zero
// (the pitch columns 64..80 are never read)
__device__ __forceinline__ void rows_commit(float* XC, const RowRegs& x, bool norm, int ob, int w) {
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        float v = x.v[j];
        if (norm && c < ob) v = fminf(fmaxf((v - x.m[j]) / x.s[j], -5.f), 5.f);
        XC[r * LD + c] = c < w ? v : 0.f;
    }
}


// D input rows, staged like the agent's (see "staged prologue").  sect 0: policy rows, 1: expert rows,
// 2: eps*policy + (1-eps)*expert  (discriminator.py:129-134)
struct DiscRowRegs { float pv[4], ev[4], e[4]; };

// thread t owns row t >> 4 and the columns (t & 15) + 16 j; pg = physical policy row of that row
__device__ __forceinline__ void disc_rows_issue(DiscRowRegs& x, const DiscArgs& a, int64_t row0, int sect,
                                                int64_t pg, int64_t e0) {
    const int ob = a.ob, ac = a.ac, w = ob + ac;
    const int c0 = threadIdx.x & 15;
    const int64_t g = row0 + (threadIdx.x >> 4), eg = e0 + g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        x.pv[j] = 0.f; x.ev[j] = 0.f; x.e[j] = 0.f;
        if (c < ob) {
            if (sect != 1) x.pv[j] = a.p_obs[pg * ob + c];
            if (sect != 0) x.ev[j] = a.e_obs[eg * ob + c];
            if (sect == 2 && a.eps_ob) x.e[j] = a.eps_ob[g * ob + c];
        } else if (c < w) {
            if (sect != 1) x.pv[j] = a.p_acs[pg * ac + (c - ob)];
            if (sect != 0) x.ev[j] = a.e_acs[eg * ac + (c - ob)];
            if (sect == 2 && a.eps_ob) x.e[j] = a.eps_ac[g * ac + (c - ob)];
        }
    }
}

// ms = mean[64] | std[64] in LDS (valid when norm)
// (The interpolation weights that are drawn here are drawn in THIS half.  Drawing them in disc_rows_issue, while the
// row loads fly, measured slower at C2 — 35.97 -> 35.66 k steps/s alternating on one box: the Philox evaluation then
// stands in front of the weights' LDS commit and the statistics.)
__device__ __forceinline__ void disc_rows_commit(float* XC, const DiscRowRegs& x, const DiscArgs& a, int64_t row0,
                                                 int sect, bool norm, const float* ms, uint64_t step) {
    const int ob = a.ob, ac = a.ac, w = ob + ac;
    const int r = threadIdx.x >> 4, c0 = threadIdx.x & 15;
    const int64_t g = row0 + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + 16 * j;
        float pv = x.pv[j], ev = x.ev[j], e = x.e[j];
        if (c < ob) {
            if (sect == 2 && !a.eps_ob)
                e = philox_uniform(a.rng_seed, step, RNG_EPS_OB, (uint64_t)(g * ob + c));
            if (norm) {
                const float m = ms[c], sd = ms[64 + c];
                pv = fminf(fmaxf((pv - m) / sd, -5.f), 5.f);
                ev = fminf(fmaxf((ev - m) / sd, -5.f), 5.f);
            }
        } else if (c < w && sect == 2 && !a.eps_ob) {
            e = philox_uniform(a.rng_seed, step, RNG_EPS_AC, (uint64_t)(g * ac + (c - ob)));
        }
        XC[r * LD + c] = c < w ? (sect == 0 ? pv : (sect == 1 ? ev : e * pv + (1.f - e) * ev)) : 0.f;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sig_ce(float x, float z) {
    return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

}  // namespace
}  // namespace fused
}  // namespace samx
