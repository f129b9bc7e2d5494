// Role-fused kernel for 256x256 leaky-ReLU networks (Humanoid-shaped configuration: ob 376, ac 17,
// batch 1024).  One workgroup owns 16 batch rows and runs the ENTIRE chain of layer products of its role
// for them.  It has 8 waves, two per SIMD, and every wave computes: wave w owns the output columns
// [32 w, 32 w + 32) of every product (two 16x16 accumulator blocks), loads ITS weight fragments straight from L2
// into registers in MFMA layout (the weights never touch LDS: each element is used by one wave, once per tile),
// and runs its share of every epilogue.  The row code (input gather, normalisation) runs on all 512 threads, the
// LayerNorm mini-phases and the 1-wide heads on the 256 "row threads" of waves 0-3.
// Roles:
//   TC  target actor + target critic on obs1 -> TD target, then the online critic on (obs0, acs): forward,
//       huber loss, backward to the first pre-activation            (sam_agent.py:196-205,457-520)
//   A   actor forward, critic forward on [obz0 | mu(obz0)], backward through the critic to the action,
//       tanh, backward through the actor                            (sam_agent.py:184-194,406-433)
//   D   policy / expert rows: forward, BCE terms, backward; interpolate rows: forward, input gradient g,
//       its squared sums, V = g / (B dim) and the tangent forward along V (gradient penalty)
//                                                                   (discriminator.py:77-161)
// Activations stay in LDS ([16][260] tiles).  All matmuls are v_mfma_f32_16x16x4_f32 (exact f32).  A layer's k
// index is visited in the order k = 32 t + 16 g + 4 (lane >> 4) + m so that the activation fragment is ONE
// ds_read_b128 per four MFMAs and the weight fragment ONE 16-byte global load per four MFMAs — from a
// k-interleaved image of the matrix (forward products) or of its transpose (backward products), element (k, j) at
// ((k >> 2) * J + j) * 4 + (k & 3): pack_kernel writes the images from the parameter vectors, the optimiser pass
// keeps them current inside a multi-row launch (fused_small.hip apply_kernel, samx_step_apply mode bit 6).
// What the weight-gradient products need — layer inputs and pre-activation cotangents — is written to
// HBM; they are batch reductions and run as ONE grouped MFMA launch afterwards (nets.hip: gemm_group).
// (History: round 2 streamed the weights through a ring of three 32-deep LDS stages filled by four loader waves;
// what is left of it here is the name `ring` of the LDS area that now holds the staged action-head matrix, the
// LayerNorm column-sum scratch and the interpolation weights — free_slot.)
#include "fused_wide.h"
#include "rng.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace samx {
namespace wide {
namespace {

constexpr int NT = 512;                // threads per workgroup: eight waves, two per SIMD; every wave owns 32 output
                                       // columns of every product (round 2: waves 4-7 streamed the weights through LDS)
constexpr int NC = 256;                // "row threads" (waves 0-3): everything that is not a product or its epilogue
constexpr int NB = 2;                  // 16-column accumulator blocks per wave
constexpr int R = kRows;
constexpr int H = kH;
constexpr int LDT = 260;               // pitch of a [16][256] tile
constexpr int WP = 260;                // pitch of a staged weight row (NN image)
constexpr int STAGE_F = 32 * WP;       // floats per ring stage (NT image: 256 * 32 fits)
constexpr int NSTG = 2;                // aux slots (the weight ring of round 2; now: head matrices, LayerNorm scratch, E)
constexpr int PP = 36;                 // pitch of the [16][ac <= 32] tile of the actor's pre-tanh cotangent
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Phase timestamps of workgroup 0 of each role (profiling builds only: -DSAMX_TRACE; never shipped)
#ifdef SAMX_TRACE
__device__ unsigned long long g_wtrace[256];
#define WTRACE(w_, role, i) do { if ((w_) == 0 && threadIdx.x == 0) g_wtrace[(role) * 32 + (i)] = wall_clock64(); } while (0)
// per weight stage of block 0 (the first target+critic workgroup): entries 128 + stream index
#define WTRACE_STAGE(s_) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (s_) < 128) g_wtrace[128 + (s_)] = wall_clock64(); } while (0)
#else
#define WTRACE(w_, role, i) do { } while (0)
#define WTRACE_STAGE(s_) do { } while (0)
#endif
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;

__device__ __forceinline__ float lrelu(float z) { return fmaxf(z, 0.2f * z); }
__device__ __forceinline__ float dlrelu(float a) { return a >= 0.f ? 1.f : 0.2f; }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Barrier that does not drain the global loads in flight (a __syncthreads() would).  The wait is the BUILTIN
// (vmcnt 63, expcnt 7, lgkmcnt 0) so that the compiler's wait-count bookkeeping sees it: behind an inline-asm
// wait it assumed the fragment reads of the previous stage were still pending and made the first MFMAs of a
// stage wait for reads issued for the NEXT one (which queue behind the loader waves' LDS writes).
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("s_barrier" ::: "memory");
}
// wave-uniform (scalar) role of this wave
__device__ __forceinline__ bool is_compute() { return __builtin_amdgcn_readfirstlane((int)threadIdx.x) < NC; }
__device__ __forceinline__ bool is_row() { return is_compute(); }   // wave-uniform: one of the 256 row threads
__device__ __forceinline__ float sum16(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ __forceinline__ float sum32(float v) {   // over the 32 lanes of a row in the rows mapping
    v += __shfl_xor(v, 16, 64);
    return sum16(v);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sig_ce(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ int64_t ring_row(const int64_t* idx, int64_t start, int64_t len, int64_t g) {
    if (!idx) return g;
    int64_t s = start + idx[g];
    return s >= len ? s - len : s;
}

// acc[nb] (+)= A[16][K] * B[K][64 w + 16 nb .. +16)   for this wave's four 16-column blocks.
//   A: LDS, pitch lda (a multiple of 4), zero in the columns [K, 32 ceil(K/32)).
//   NTR = false: B[k][j] = W[k * ldw + jg0 + j], rows k < K (clamped: the padded A columns are zero)
//   NTR = true : B[k][j] = W[(jg0 + j) * ldw + k], rows jg0 + j < N (clamped)
// nvalid: number of valid output columns of this call (blocks that start beyond it are skipped).
// Ends with a barrier: afterwards every wave is done with A and with the ring.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte-aligned address

// one layer product's weight operand:
//   B[k][j] = the element at ((k >> 2) * ldw + jg0 + j) * 4 + (k & 3) of a K-INTERLEAVED IMAGE (pack_kernel):
//   forward products (w_nn): the image of a [K][256] matrix, ldw = 256, rows up to 32 ceil(K / 32) present (zero);
//   backward products (w_t): the image of the TRANSPOSE, ldw = its column slots (Net::W1t / W2t), K = 256; slots
//   beyond the matrix hold anything — what they produce is never stored
// nvalid: number of valid output columns (16-column blocks that start beyond it are skipped)
struct WDesc {
    const float* W;
    int ldw, K, N, jg0, nvalid;
    bool nt;
};
__device__ __forceinline__ int wstages(const WDesc& d) { return (d.K + 31) >> 5; }

// a wave-uniform pointer in SGPRs: loads through it take the scalar-base form (one VGPR offset per lane
// instead of a 64-bit VGPR address per load)
typedef const __attribute__((address_space(1))) f32x4u glb_f32x4u;
__device__ __forceinline__ glb_float* uniform_ptr(const float* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (glb_float*)(((uint64_t)hi << 32) | lo);   // stays a GLOBAL pointer (a generic one loads through flat)
}
__device__ __forceinline__ f32x4 load16(glb_float* base_uniform, unsigned off_floats) {
    return *(glb_f32x4u*)(base_uniform + off_floats);
}

// (The lane index goes through an opaque asm so that the per-lane offsets are recomputed where they are
// used instead of being hoisted out of the stage loop.)
__device__ __forceinline__ int opaque_lane() {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    return lane;
}

// The weight operand never touches LDS (round 3).  Every weight element of a product is used by exactly ONE wave
// (wave w owns the output columns [64 w, 64 w + 64)) and exactly once per 16-row tile, so staging it through a
// shared ring bought nothing and cost a 32 KB LDS write burst, 32 KB of fragment reads and a workgroup barrier per
// 32-deep stage.  Each wave loads ITS fragments straight from L2 into registers, in MFMA layout, with 16-byte loads:
// lane (r = lane & 15, kq = lane >> 4) needs B[32 t + 16 g + 4 kq + m][64 w + 16 nb + r] for m = 0..3 —
//   NT products (backward): the same, from the k-interleaved image of the transpose (Net::W1t / W2t);
//   NN products (forward): four consecutive floats of the k-interleaved image of W that pack_kernel writes at the
//       head of the launch sequence (a transposition inside blocks of 4 rows: 3.3 MB for the five networks).
// Eight loads per wave and stage, NSET - 1 stages in flight; the only barriers left are the two around a product
// (activation tile complete / everyone done with it).  No row clamps, as before: an NT stage may read rows beyond
// the matrix but inside the parameter vector (what it fetches lands in output columns >= nvalid that are never
// stored); the NN image is padded with zero rows.  The k order of every accumulator is what it was with the ring.
// (Measured and dropped: the forward products straight from the row-major matrices, no packed image — number
// the wave's columns 32 w + 2 r + nb and a lane's two blocks are one 8-byte load per k.  Bit-identical, no pack
// launch, and slower: C4 7.68 -> 7.55 k steps/s, launcher options 4.72 -> 4.50 k.  Eight 8-byte loads per wave and
// stage instead of four 16-byte ones: the address path takes 16 cycles per wave-wide load whatever its width.)
struct BFrag { f32x4 b[2][NB]; };   // one stage of this wave: [g][nb], element m = k offset
// Stages of the NEXT product requested before the current one's epilogue; the remaining NSET - 1 - PRE are
// requested when the product starts.  Measured on one box, C4: PRE 2 -> 7.69 / 7.63 k steps/s, PRE 3 (everything a
// product has in flight, so that no first-stage load queues behind the epilogue's stores) -> 7.60 / 7.57 k: the
// extra 16 registers cost more than the shorter first stage gains.
#ifndef SAMX_WIDE_PRE
#define SAMX_WIDE_PRE 2
#endif
constexpr int PRE = SAMX_WIDE_PRE;
struct WPipe {
    BFrag pre[PRE];   // stages 0 .. PRE-1 of the next product, in flight across the epilogue
    int s;            // stages consumed so far by this chain (trace stamps)
    bool primed;      // pre[] was requested for the product that comes next
};
__device__ __forceinline__ void wpipe_init(WPipe& p) { p.s = 0; p.primed = false; }

__device__ __forceinline__ void bstage_load(BFrag& f, const WDesc& d, int t) {
    const int lane = opaque_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), r = lane & 15, kq = lane >> 4;
    const unsigned off = (unsigned)((kq * d.ldw + 16 * NB * wv + r) * 4);   // (ldw: a constant at every call site)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        glb_float* base = uniform_ptr(d.W + ((int64_t)(8 * t + 4 * g) * d.ldw + d.jg0) * 4);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) f.b[g][nb] = load16(base + nb * 64, off);
    }
}

// Stage t's MFMAs run from register set t % NSET while the loads of stages t+1 .. t+NSET-1 are in flight (the loop
// is unrolled by NSET so that no set is indexed at run time) and the activation fragment of stage t+1 is read from
// LDS.  (Six sets were measured: the same 0.64 us per stage and longer epilogues — more registers to park.)
struct AFrag { f32x4 a4[2]; };
#ifndef SAMX_WIDE_NSET
#define SAMX_WIDE_NSET 4
#endif
constexpr int NSET = SAMX_WIDE_NSET;   // even (the activation fragments alternate between two sets)
static_assert(NSET >= 4 && NSET % 2 == 0, "NSET");
__device__ __forceinline__ void wcompute_run(WPipe& pipe, const WDesc cur, const WDesc next, const bool has_next,
                                             const float* A, int lda, f32x4 (&acc)[NB]) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const int T = wstages(cur);
    auto rda = [&](int t, AFrag& f) {
#pragma unroll
        for (int g = 0; g < 2; ++g) f.a4[g] = *(const f32x4*)(A + r * lda + (t << 5) + 16 * g + 4 * kq);
    };
    // consecutive MFMAs go to different accumulators (a dependent 16x16x4 pair costs 40 cycles, not 32);
    // blocks beyond nvalid compute finite values that are never stored
    auto mm = [&](const AFrag& a, const BFrag& f) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma4(a.a4[g][m], f.b[g][nb][m], acc[nb]);
    };
    static_assert(PRE <= NSET - 1, "a product starts with NSET - 1 stages in flight");
    BFrag f[NSET];
    if (pipe.primed) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) f[u] = pipe.pre[u];
#pragma unroll
        for (int u = PRE; u < NSET - 1; ++u)
            if (u < T) bstage_load(f[u], cur, u);
    } else {
#pragma unroll
        for (int u = 0; u < NSET - 1; ++u)
            if (u < T) bstage_load(f[u], cur, u);
    }
    // (No barrier here: every caller has one between the last write of the activation tile and this call — the
    // epilogue of the product in front, a row / LayerNorm mini-phase, the action head — and nothing but requests
    // and register work in between.)
    WTRACE_STAGE(pipe.s);
    AFrag a[2];
    rda(0, a[0]);
#pragma nounroll
    for (int t = 0; t < T; t += NSET) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            if (t + i < T) {
                if (t + i + NSET - 1 < T) bstage_load(f[(i + NSET - 1) % NSET], cur, t + i + NSET - 1);
                if (t + i + 1 < T) rda(t + i + 1, a[(i + 1) & 1]);
                if (t + i > 0) WTRACE_STAGE(pipe.s + t + i);
                // (a sched_barrier here, to keep the stages-ahead requests in front of this stage's MFMAs — the compiler
                // sinks the activation-fragment reads behind them — measured 8.26 -> 8.22 k steps/s at C4: not kept)
                mm(a[i & 1], f[i]);
            }
        }
    }
    if (has_next) {   // the next product's first stages fly during this product's epilogue
        const int Tn = wstages(next);
#pragma unroll
        for (int u = 0; u < PRE; ++u)
            if (u < Tn) bstage_load(pipe.pre[u], next, u);
    }
    // every wave is done with the activation tile.  (Needed only where the epilogue overwrites that tile; leaving it
    // out everywhere else measured neutral — C4 8.46 -> 8.43 k, launcher options 5.14 -> 5.16 k — and is not done.)
    bar();
}

// acc[nb] (+)= A[16][K] * B[K][64 w + 16 nb .. +16) for wave w's four 16-column blocks; A in LDS with pitch lda (a
// multiple of 4) and zeros in the columns [K, 32 ceil(K/32)).  Called by ALL waves.
// Starts WITHOUT a barrier (the caller's last barrier must follow the last write of A) and ends with one: afterwards
// every wave is done with A.
template <bool CW>
__device__ __forceinline__ void wprod(WPipe& pipe, const WDesc cur, const WDesc next, bool has_next, const float* A,
                                      int lda, float* ring, f32x4 (&acc)[NB]) {
    (void)ring;
    static_assert(CW, "every wave computes (round 2's loader waves are gone)");
    wcompute_run(pipe, cur, next, has_next, A, lda, acc);
    pipe.s += wstages(cur);
    pipe.primed = has_next;
}
template <bool CW>
__device__ __forceinline__ void wprod(WPipe& pipe, const WDesc cur, const float* A, int lda, float* ring,
                                      f32x4 (&acc)[NB]) {   // last product of a chain
    wprod<CW>(pipe, cur, cur, false, A, lda, ring, acc);
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NB]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// f(row, col, value, bit index) over this (compute) wave's accumulator elements
template <typename F>
__device__ __forceinline__ void for_acc(const f32x4 (&acc)[NB], F f) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            f((lane >> 4) * 4 + j, 16 * NB * wv + 16 * nb + (lane & 15), acc[nb][j], nb * 4 + j);
}

__device__ __forceinline__ WDesc w_nn(const float* W, int K) { return WDesc{W, H, K, H, 0, H, false}; }
// backward product through an image of the transpose with J column slots: output columns jg0 .. jg0 + 255
__device__ __forceinline__ WDesc w_t(const float* Wt, int J, int jg0) { return WDesc{Wt, J, H, H, jg0, H, false}; }

// this lane's bias values in the for_acc mapping (column 16 NB wv + 16 nb + (lane & 15) -> b[nb])
struct BiasRegs { float b[NB]; };
__device__ __forceinline__ void bias_issue(BiasRegs& bb, const float* __restrict__ b) {
    const int c0 = 16 * NB * ((int)threadIdx.x >> 6) + ((int)threadIdx.x & 15);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bb.b[nb] = b[c0 + 16 * nb];
}

// hidden layer: out = lrelu(A W + b) into the LDS tile T and, if hbm, to hbm[(row0 + r)][256].
// Called by all waves; returns this (compute) lane's 16 mask bits (value >= 0).
template <bool CW>
__device__ __forceinline__ unsigned hidden_fwd(WPipe& pipe, const WDesc cur, const WDesc next, const bool has_next,
                                               const float* A, int lda, const float* __restrict__ b, float* ring,
                                               float* T, float* __restrict__ hbm, int64_t row0) {
    f32x4 acc[NB];
    zero_acc(acc);
    BiasRegs bb;
    bias_issue(bb, b);   // ahead of the product: behind its last barrier the load would be a bare L2 round trip
    wprod<CW>(pipe, cur, next, has_next, A, lda, ring, acc);
    unsigned bits = 0;
    for_acc(acc, [&](int r, int c, float v, int bi) {
            const float a = lrelu(v + bb.b[bi >> 2]);
            T[r * LDT + c] = a;
            if (hbm) hbm[(row0 + r) * H + c] = a;
            bits |= (a >= 0.f ? 1u : 0u) << bi;
        });
    bar();   // T is complete for every reader that follows
    return bits;
}

// The helpers below (head1 .. head_bwd) are for the row threads only: threadIdx.x < NC.
// s[r] = <T[r], w3> + b3  (1-output head; w3 / b3 from global memory)
// The head's weights travel ahead (head_issue before the product that fills T, head1 after it): their loads
// would otherwise be a bare L2 round trip in the middle of the chain.
struct HeadRegs { float w[H / 16], b; };
__device__ __forceinline__ void head_issue(HeadRegs& h, const float* __restrict__ w3, const float* __restrict__ b3) {
    const int q = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < H / 16; ++i) h.w[i] = w3[q + 16 * i];
    h.b = b3[0];
}
__device__ __forceinline__ void head1(const float* T, const HeadRegs& h, float* s_out) {
    const int r = threadIdx.x >> 4, q = threadIdx.x & 15;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < H / 16; ++i) acc = fmaf(T[r * LDT + q + 16 * i], h.w[i], acc);
    acc = sum16(acc);
    if (q == 0) s_out[r] = acc + h.b;
}

// this tile's share of the head's gradient: slot[c] = sum_r T[r][c] dq[r] (c < 256), slot[256] = sum_r dq[r].
// Thread t reads only column t of T: head_bwd (same thread, same column) may follow without a barrier.
__device__ __forceinline__ void head_grad_slot(const float* T, const float* dq, float* __restrict__ slot) {
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) s = fmaf(T[r * LDT + c], dq[r], s);
    slot[c] = s;
    if (c == 0) {
        float b = 0.f;
        for (int r = 0; r < R; ++r) b += dq[r];
        slot[H] = b;
    }
}

// T[r][c] = dq[r] * w3[c] * lrelu'(T[r][c]) in place, and to hbm
// (w3c = w3[threadIdx.x & 255], loaded ahead by the caller: thread t only meets column t & 255); called by the
// first NTH threads of the workgroup (the 256 row threads where head_grad_slot reads the same column first, else all)
template <int NTH>
__device__ __forceinline__ void head_bwd(float* T, const float* dq, const float w3c,
                                         float* __restrict__ hbm, int64_t row0) {
    static_assert(NTH % H == 0, "one column per thread");
    for (int t = threadIdx.x; t < R * H; t += NTH) {
        const int r = t >> 8, c = t & 255;
        const float v = dq[r] * w3c * dlrelu(T[r * LDT + c]);
        T[r * LDT + c] = v;
        hbm[(row0 + r) * H + c] = v;
    }
}

// the whole head matrix W3[256][out <= 32] (contiguous) into one ring slot as it lies: S[k * out + j].
// With a primed weight stream (stages 0 and 1 of the next product in LDS) the free slot is
// free_slot(ring, pipe): stage 2 lands there only after the next product's first barrier.
__device__ __forceinline__ float* free_slot(float* ring, const WPipe& pipe) {
    return ring + ((pipe.s + 2) % NSTG) * STAGE_F;
}
// Two halves (every thread of the workgroup, all waves): the loads go out before the product in front,
// the LDS writes and the barrier come after it.
struct HeadStage { f32x4 q[2048 / NT]; };   // 64 * out <= 2048 chunks of 16 bytes over the workgroup's threads
__device__ __forceinline__ void stage_head_issue(HeadStage& hs, const float* __restrict__ W3, int out) {
    const int n4 = 64 * out;
#pragma unroll
    for (int i = 0; i < 2048 / NT; ++i) {
        const int t = min((int)threadIdx.x + NT * i, n4 - 1);
        hs.q[i] = *(const f32x4u*)(W3 + 4 * t);
    }
}
__device__ __forceinline__ void stage_head_commit(const HeadStage& hs, int out, float* slot) {
    const int n4 = 64 * out;
#pragma unroll
    for (int i = 0; i < 2048 / NT; ++i) {
        const int t = threadIdx.x + NT * i;
        if (t < n4) *(f32x4*)(slot + 4 * t) = hs.q[i];
    }
    // the backward through the head reads k up to 4 ceil(out / 4) - 1 of every row: behind the LAST row that is up to
    // 3 floats past the image, times a zero of the padded cotangent — which must not meet a NaN pattern left in LDS
    // (with the weight ring of round 2 whatever lay there was a finite weight)
    if (threadIdx.x < 4) slot[256 * out + threadIdx.x] = 0.f;
    bar();
}

struct Lds {
    float *ring, *X, *T1, *T2, *PRE, *s_, *tq, *dq, *red, *rw, *dn, *iw, *tl;
    float* st;   // [kMaxIn][2]: (mean, 1 / std) of the input columns the rows are normalised with
    int px;
};
// (the full width whatever the input: rows_commit writes all kMaxIn columns)
__host__ __device__ inline int pitch_x(int) { return kMaxIn + 4; }
__host__ __device__ inline size_t lds_floats(int in) {
    return (size_t)NSTG * STAGE_F + 2 * kMaxIn + (size_t)R * pitch_x(in) + 2 * R * LDT + R * PP + 16 * 8 + 64;
}
__device__ inline Lds carve(float* p, int in) {
    Lds L;
    L.px = pitch_x(in);
    L.ring = p; p += NSTG * STAGE_F;
    L.st = p; p += 2 * kMaxIn;
    L.X = p; p += R * L.px;
    L.T1 = p; p += R * LDT;
    L.T2 = p; p += R * LDT;
    L.PRE = p; p += R * PP;
    L.s_ = p; L.tq = p + 16; L.dq = p + 32; L.rw = p + 48; L.dn = p + 64; L.iw = p + 80; L.tl = p + 96;
    L.red = p + 112;   // 64 floats
    return L;
}

// rows [obz | acs] of the batch into X[16][px], zero up to k32 columns; optional dense copy to HBM.
// ALL 512 threads: thread t owns row t >> 5 and the columns (t & 31) + 32 u (a wave-wide load covers two rows x 128
// contiguous bytes).  Two halves: rows_issue sends ALL the row loads (raw values, one register per column) and
// returns; rows_commit normalises and writes.  A role issues the next input's rows before a product that does not
// need X and commits them after it: the HBM gather's latency (2-3 us) hides behind the product.
// The statistics come from an LDS table of (mean, 1 / std) pairs that the role stages once (stats_stage): with the
// 16-column mapping on four waves, a division per element and the statistics fetched from global memory inside the
// commit, a commit cost 2.5 us in the middle of the target+critic chain and the discriminator's interpolate tiles
// spent 6.5 us combining their two rows.
constexpr int kRowRegs = kMaxIn / 32;   // 13
static_assert(kMaxIn % 32 == 0 && NT == 32 * R, "rows mapping: 32 threads per row");
struct RowRegs { float v[kRowRegs]; };
// (thread index through an opaque asm, as in the loaders: the per-column predicates must be recomputed where
// they are used — kept live from the issue to the commit across a product they spill by the hundred)
__device__ __forceinline__ int opaque_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ void rows_issue(RowRegs& x, int k32, const float* __restrict__ obs,
                                           const float* __restrict__ acs, int ob, int ac, int acw, int64_t prow) {
    // branch-free: every column loads from a valid address (columns beyond the row re-read its last element;
    // rows_commit zeroes them) — a branch per column costs more than the loads
    const int c0 = opaque_tid() & 31;
    const float* ro = obs + prow * ob;
    const float* ra = acw > 0 ? acs + prow * ac : ro;
    const int amax = acw > 0 ? acw - 1 : 0;
#pragma unroll
    for (int u = 0; u < kRowRegs; ++u) {
        const int c = c0 + 32 * u;
        const bool isob = c < ob;
        const int ci = isob ? c : min(c - ob, amax);
        x.v[u] = (isob ? ro : ra)[ci];
    }
}
// (mean, 1 / std) of the ob normalised columns into the LDS table; all threads; the caller's next barrier publishes it
__device__ __forceinline__ void stats_stage(const Lds& L, const float* __restrict__ mean,
                                            const float* __restrict__ stdv, int ob) {
    if (!mean) return;
    for (int c = threadIdx.x; c < ob; c += NT) {
        L.st[2 * c] = mean[c];
        L.st[2 * c + 1] = 1.f / stdv[c];
    }
}
// one element: clip((v - mean) / std, -5, 5) for an observation column (as a product with the staged reciprocal),
// the raw value for an action column, zero beyond the row
__device__ __forceinline__ float row_norm(const Lds& L, float v, int c, int ob, int acw, bool norm) {
    if (norm) {
        const float2 ms = *(const float2*)(L.st + 2 * min(c, ob - 1));
        const float vn = fminf(fmaxf((v - ms.x) * ms.y, -5.f), 5.f);   // computed for every lane
        v = c < ob ? vn : v;
    }
    return c < ob + acw ? v : 0.f;
}
// Straight-line on purpose (selects, no branch per column).  All kMaxIn columns are written — X rows always have the
// full pitch (pitch_x) and columns >= ob + acw get zeros.  norm: the table was staged (stats_stage + barrier).
__device__ __forceinline__ void rows_commit(const Lds& L, RowRegs& x, int k32, bool norm, int ob, int acw,
                                            float* __restrict__ hbm, int64_t hrow, int ldh) {
    const int t = opaque_tid();
    const int r = t >> 5, c0 = t & 31;
    float* xr = L.X + r * L.px + c0;
#pragma unroll
    for (int u = 0; u < kRowRegs; ++u) {
        const float v = row_norm(L, x.v[u], c0 + 32 * u, ob, acw, norm);
        x.v[u] = v;
        xr[32 * u] = v;
    }
    // the HBM copy last: stores count in vmcnt like loads
    if (hbm) {
        float* hr = hbm + hrow * ldh + c0;
#pragma unroll
        for (int u = 0; u < kRowRegs; ++u)
            if (c0 + 32 * u < ob + acw) hr[32 * u] = x.v[u];
    }
}
// the first rows of a role: loads out, statistics staged while they fly, barrier, commit
__device__ __forceinline__ void load_rows_first(const Lds& L, int k32, const float* __restrict__ obs,
                                                const float* __restrict__ acs, const float* __restrict__ mean,
                                                const float* __restrict__ stdv, int ob, int ac, int acw, int64_t prow,
                                                float* __restrict__ hbm, int64_t hrow, int ldh) {
    RowRegs x;
    rows_issue(x, k32, obs, acs, ob, ac, acw, prow);
    stats_stage(L, mean, stdv, ob);
    bar();
    rows_commit(L, x, k32, mean != nullptr, ob, acw, hbm, hrow, ldh);
}

// ================================================================================================
// LayerNorm behind every hidden activation (networks.py:53-70, network_util.py:32-39: a = lrelu(xW + b),
// xh = (a - mean a) rs, rs = (var a + 1e-8)^-1/2 with the biased variance, n = gamma xh + beta).
// The 256 columns of a row are spread over the four compute waves in the accumulator layout, so every LayerNorm
// step is a MINI-PHASE between two products in the "rows mapping": the product's epilogue leaves its raw tile in
// LDS, then thread t of the 256 compute threads owns row t >> 4 and the 16 columns (t & 15) + 16 u and reduces with
// shuffles inside its 16-lane group (mean, then the centred sum of squares: two passes like tf.nn.moments).
// What a backward mini-phase needs from the forward one of the same layer travels PER THREAD: the lrelu mask as 16
// bits in a register, rs per row in a 16-float LDS strip, and xh through HBM — written and read back by the same
// thread (same mapping), so no cross-thread visibility of global stores is ever needed.  The tiles of the forward
// (a, n) are not kept: the LDS budget is the ring + X + two tiles, as without LayerNorm.
// Per-tile gradients of beta / gamma (column sums over the tile's 16 rows) go through a scratch [4 waves][512] in
// the ring slot the weight stream leaves free between two products (free_slot) and one more barrier.
constexpr float kLnEps = 1e-8f;
constexpr float kInvH = 1.f / (float)H;
constexpr int LNU = H / 32;   // columns per thread: 32 threads per row on all 512 threads (first form: 16 per row on 256)
__device__ __forceinline__ float sum_rows2(float v) {   // over the 2 rows a wave holds in the rows mapping
    v += __shfl_xor(v, 32, 64);
    return v;
}

// T holds a = lrelu(z) [16][256]; on return T holds n.  xh_hbm / n_hbm (either may be null): [rows][256] arrays in
// HBM, this tile's rows start at grow0.  Returns the mask bits (a >= 0) of this thread's 16 elements.
__device__ __forceinline__ unsigned ln_fwd_rows(float* T, const float* __restrict__ be, const float* __restrict__ ga,
                                                float* __restrict__ xh_hbm, float* __restrict__ n_hbm, int64_t grow0,
                                                float* rs_strip) {
    const int t = opaque_tid(), r = t >> 5, c0 = t & 31;
    float a[LNU], g[LNU], b[LNU];
    float s = 0.f;
    unsigned bits = 0;
#pragma unroll
    for (int u = 0; u < LNU; ++u) { g[u] = ga[c0 + 32 * u]; b[u] = be[c0 + 32 * u]; }
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        a[u] = T[r * LDT + c0 + 32 * u];
        s += a[u];
        bits |= (a[u] >= 0.f ? 1u : 0u) << u;
    }
    const float mu = sum32(s) * kInvH;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < LNU; ++u) { const float d = a[u] - mu; q = fmaf(d, d, q); }
    const float rs = 1.f / sqrtf(sum32(q) * kInvH + kLnEps);
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        const float xh = (a[u] - mu) * rs;
        a[u] = xh;
        T[r * LDT + c0 + 32 * u] = fmaf(g[u], xh, b[u]);
    }
    if (rs_strip && c0 == 0) rs_strip[r] = rs;
    // the HBM copies last (stores count in vmcnt like loads)
    if (xh_hbm) {
        float* o = xh_hbm + (grow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = a[u];
    }
    if (n_hbm) {
        float* o = n_hbm + (grow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = fmaf(g[u], a[u], b[u]);
    }
    return bits;
}

// Backward through lrelu -> LayerNorm of one layer: cotangent of n -> cotangent of the pre-activation z.
//   dn = T[r][c], or dq[r] w3[c] (HEAD: the layer under a 1-output head)
//   dxh = dn gamma ; da = rs (dxh - mean dxh - xh mean(dxh xh)) [+ add] ; dz = da lrelu'(a)
// dz -> T (in place) and dz_hbm rows drow0.. (null: LDS only); tn_hbm (optional): keeps dn (rows trow0..);
// add_hbm (optional, rows trow0..): a primal cotangent of a added before the mask (gradient penalty, P stream).
// SLOTS: per-wave column sums of dn (-> dbeta) and dn xh (-> dgamma) over the wave's 4 rows -> scratch[w][512].
template <bool HEAD, bool SLOTS>
__device__ __forceinline__ void ln_bwd_rows(float* T, const float* dq, const float* __restrict__ w3,
                                            const float* __restrict__ ga, const float* __restrict__ xh_hbm,
                                            int64_t grow0, const float* rs_strip, unsigned bits,
                                            const float* __restrict__ add_hbm, float* __restrict__ dz_hbm,
                                            int64_t drow0, float* __restrict__ tn_hbm, int64_t trow0,
                                            float* scratch) {
    const int t = opaque_tid(), r = t >> 5, c0 = t & 31, lane = t & 63, wv = t >> 6;
    float xh[LNU], dn[LNU], dx[LNU];
    {
        const float* xi = xh_hbm + (grow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) xh[u] = xi[32 * u];
    }
    const float rs = rs_strip[r];
    float s1 = 0.f, s2 = 0.f;
    const float dqr = HEAD ? dq[r] : 0.f;
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        const int c = c0 + 32 * u;
        dn[u] = HEAD ? dqr * w3[c] : T[r * LDT + c];
        dx[u] = dn[u] * ga[c];
        s1 += dx[u];
        s2 = fmaf(dx[u], xh[u], s2);
    }
    const float m1 = sum32(s1) * kInvH, m2 = sum32(s2) * kInvH;
    float ad[LNU];
    if (add_hbm) {
        const float* ai = add_hbm + (trow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) ad[u] = ai[32 * u];
    }
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        float da = rs * (dx[u] - m1 - xh[u] * m2);
        if (add_hbm) da += ad[u];
        dx[u] = da * ((bits >> u) & 1u ? 1.f : 0.2f);
        T[r * LDT + c0 + 32 * u] = dx[u];
    }
    if (SLOTS) {
#pragma unroll
        for (int u = 0; u < LNU; ++u) {
            const float b = sum_rows2(dn[u]), g = sum_rows2(dn[u] * xh[u]);
            if (lane < 32) { scratch[wv * 512 + c0 + 32 * u] = b; scratch[wv * 512 + 256 + c0 + 32 * u] = g; }
        }
    }
    if (dz_hbm) {
        float* o = dz_hbm + (drow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = dx[u];
    }
    if (tn_hbm) {
        float* o = tn_hbm + (trow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = dn[u];
    }
}
// the eight waves' column sums in a fixed order
__device__ __forceinline__ float ln_col8(const float* scratch, int o) {
    return ((scratch[o] + scratch[512 + o]) + (scratch[1024 + o] + scratch[1536 + o])) +
           ((scratch[2048 + o] + scratch[2560 + o]) + (scratch[3072 + o] + scratch[3584 + o]));
}
// dst[0..256) = dbeta, dst[256..512) = dgamma (contiguous in the flat layout)
__device__ __forceinline__ void ln_cols_commit(const float* scratch, float* __restrict__ dst) {   // all threads
    const int t = threadIdx.x;
    dst[t] = ln_col8(scratch, t);
}

// Tangent of lrelu -> LayerNorm along zdot (T holds the raw tangent pre-activation) and the primal cotangent the
// tangent LayerNorm sends back (gradient penalty's second-order term; nets.hip ln_tan_kernel):
//   adot = zdot lrelu'(a) ; d = adot - mean adot ; c = mean(xh adot) ; u = rs (d - c xh) ; ndot = gamma u
//   ub = gamma tn ; Aq = <ub, d> ; Bq = <ub, xh> ; ta = rs (ub - mean ub - xh Bq/H)
//   pa = -(rs^2 (Aq - c Bq)/H) xh - (rs Bq/H) u - rs c ta
// tn = the tangent stream's cotangent of n: tn_hbm rows trow0.. (same-thread read-back) or w3[c] (layer under the head).
// LAYER2 = false: T <- ndot (next product's operand), nd_hbm rows drow0.. <- ndot, pa_hbm rows trow0.. <- pa (unmasked);
//                 scratch[w][256..512) <- column sums of tn u.
// LAYER2 = true : T <- pa lrelu'(a) (= P_z2), pz_hbm rows drow0.. <- the same; scratch[w][0..256) <- column sums of ndot,
//                 [256..512) <- of tn u.
template <bool LAYER2>
__device__ __forceinline__ void ln_tan_rows(float* T, const float* __restrict__ ga, const float* __restrict__ xh_hbm,
                                            int64_t grow0, const float* rs_strip, unsigned bits,
                                            const float* __restrict__ tn_hbm, const float* __restrict__ w3,
                                            int64_t trow0, float* __restrict__ out_hbm, int64_t drow0,
                                            float* __restrict__ pa_hbm, float* scratch) {
    const int t = opaque_tid(), r = t >> 5, c0 = t & 31, lane = t & 63, wv = t >> 6;
    float xh[LNU], ad[LNU], ub[LNU], tn[LNU];
    {
        const float* xi = xh_hbm + (grow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) xh[u] = xi[32 * u];
        if (!LAYER2) {
            const float* ti = tn_hbm + (trow0 + r) * H + c0;
#pragma unroll
            for (int u = 0; u < LNU; ++u) tn[u] = ti[32 * u];
        } else {
#pragma unroll
            for (int u = 0; u < LNU; ++u) tn[u] = w3[c0 + 32 * u];
        }
    }
    const float rs = rs_strip[r];
    float s1 = 0.f, s2 = 0.f, sa = 0.f, su = 0.f, sb = 0.f;
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        const int c = c0 + 32 * u;
        ad[u] = T[r * LDT + c] * ((bits >> u) & 1u ? 1.f : 0.2f);
        ub[u] = ga[c] * tn[u];
        s1 += ad[u];
        s2 = fmaf(xh[u], ad[u], s2);
        sa = fmaf(ub[u], ad[u], sa);
        su += ub[u];
        sb = fmaf(ub[u], xh[u], sb);
    }
    const float mean_ad = sum32(s1) * kInvH, cc = sum32(s2) * kInvH;
    const float su_ = sum32(su), Aq = sum32(sa) - mean_ad * su_, Bq = sum32(sb);
    const float k1 = -(rs * rs * (Aq - cc * Bq) * kInvH), k2 = -(rs * Bq * kInvH), k3 = -(rs * cc);
    const float mub = su_ * kInvH, mbx = Bq * kInvH;
    float o1[LNU], o2[LNU];
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        const int c = c0 + 32 * u;
        const float uu = rs * ((ad[u] - mean_ad) - cc * xh[u]);
        const float nd = ga[c] * uu;
        const float ta = rs * (ub[u] - mub - xh[u] * mbx);
        const float pa = k1 * xh[u] + k2 * uu + k3 * ta;
        const float pz = pa * ((bits >> u) & 1u ? 1.f : 0.2f);
        T[r * LDT + c] = LAYER2 ? pz : nd;
        o1[u] = LAYER2 ? pz : nd;
        o2[u] = pa;
        const float sg = sum_rows2(tn[u] * uu);
        if (lane < 32) scratch[wv * 512 + 256 + c] = sg;
        if (LAYER2) {
            const float sn = sum_rows2(nd);
            if (lane < 32) scratch[wv * 512 + c] = sn;
        }
    }
    {
        float* o = out_hbm + (drow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = o1[u];
    }
    if (!LAYER2) {
        float* o = pa_hbm + (trow0 + r) * H + c0;
#pragma unroll
        for (int u = 0; u < LNU; ++u) o[32 * u] = o2[u];
    }
}

// (Measured and dropped: the LayerNorm parameters and 1-wide head weights of a role's networks staged in an LDS
// table at the start of the role, so that no mini-phase reads them from global memory behind a barrier — launcher
// options at C4 5242 against 5242 steps/s on one box, alternating: those reads are not what a mini-phase waits for.)
// hidden layer with LayerNorm: T <- n = LN(lrelu(A W + b)); two barriers more than hidden_fwd.  Called by all
// waves; returns the mask bits of this (compute) thread's elements IN THE ROWS MAPPING.
template <bool CW>
__device__ __forceinline__ unsigned hidden_fwd_ln(WPipe& pipe, const WDesc cur, const WDesc next, const bool has_next,
                                                  const float* A, int lda, const float* __restrict__ b,
                                                  const float* __restrict__ be, const float* __restrict__ ga,
                                                  float* ring, float* T, float* __restrict__ xh_hbm,
                                                  float* __restrict__ n_hbm, int64_t grow0, float* rs_strip) {
    f32x4 acc[NB];
    zero_acc(acc);
    BiasRegs bb;
    bias_issue(bb, b);
    wprod<CW>(pipe, cur, next, has_next, A, lda, ring, acc);
    for_acc(acc, [&](int r, int c, float v, int bi) { T[r * LDT + c] = lrelu(v + bb.b[bi >> 2]); });
    bar();
    unsigned bits = 0;
    bits = ln_fwd_rows(T, be, ga, xh_hbm, n_hbm, grow0, rs_strip);
    bar();
    return bits;
}
// either form of a hidden layer behind one call (the forward-only chains of the target networks):
// layer 1 or 2 of net n; hbm: where the next layer's input goes for the weight-gradient products (or null)
template <bool CW, bool LN>
__device__ __forceinline__ unsigned hfwd(WPipe& pipe, const WDesc cur, const WDesc next, const bool has_next,
                                         const float* A, int lda, const Net& n, int layer, float* ring, float* T,
                                         float* __restrict__ hbm, float* __restrict__ xh_hbm, int64_t grow0,
                                         float* rs_strip) {
    if constexpr (LN)
        return hidden_fwd_ln<CW>(pipe, cur, next, has_next, A, lda, layer == 1 ? n.b1 : n.b2,
                                 layer == 1 ? n.be1 : n.be2, layer == 1 ? n.ga1 : n.ga2, ring, T, xh_hbm, hbm, grow0,
                                 rs_strip);
    else
        return hidden_fwd<CW>(pipe, cur, next, has_next, A, lda, layer == 1 ? n.b1 : n.b2, ring, T, hbm, grow0);
}
// raw product output into a tile (the cotangent / tangent a LayerNorm mini-phase takes next); all waves
template <bool CW>
__device__ __forceinline__ void prod_to_tile(WPipe& pipe, const WDesc cur, const WDesc next, const bool has_next,
                                             const float* A, int lda, float* ring, float* T) {
    f32x4 acc[NB];
    zero_acc(acc);
    wprod<CW>(pipe, cur, next, has_next, A, lda, ring, acc);
    for_acc(acc, [&](int r, int c, float v, int) { T[r * LDT + c] = v; });
    bar();
}

// actor head on T2: acs = clip(high * tanh(T2 W3 + b3)) into X[:, ob:ob+ac] (and to hbm rows, ld = ob + ac);
// called by all waves.  The 256-deep reduction is split over the eight waves (one or two 16-column blocks x eight or
// four k ranges), each wave reads its operands in one batch, the partial tiles meet in `part` (2048 floats of LDS
// that nobody else uses meanwhile: the callers pass T1) and are added in a fixed order.  (Two waves running 64
// dependent [two LDS reads -> MFMA] steps took 2.6 us.)
template <int KQ>   // MFMAs per wave: the wave's k range is 4 KQ deep, lane group kq owns KQ consecutive k
__device__ __forceinline__ f32x4 actor_head_part(const float* T2, const float* slot, int ac, int k0, int jc) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const int kb = k0 + kq * KQ;
    float av[KQ], bv[KQ];
#pragma unroll
    for (int i = 0; i < KQ; i += 4) {
        const f32x4 q = *(const f32x4*)(T2 + r * LDT + kb + i);
#pragma unroll
        for (int m = 0; m < 4; ++m) av[i + m] = q[m];
    }
#pragma unroll
    for (int i = 0; i < KQ; ++i) bv[i] = slot[(kb + i) * ac + jc];
    f32x4 a4[4];   // four independent chains (a dependent 16x16x4 pair costs 40 cycles), summed at the end
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KQ; ++i) a4[i & 3] = mfma4(av[i], bv[i], a4[i & 3]);
    return (a4[0] + a4[1]) + (a4[2] + a4[3]);
}
template <bool CW>
__device__ __forceinline__ void actor_head(const Lds& L, const Net& n, int ob, int ac, float high,
                                           float* __restrict__ hbm, int64_t row0, float* slot,
                                           const HeadStage& hs, float* part) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15;
    // epilogue mapping: thread t < 256 nblk owns (block t >> 8, row (t >> 4) & 15, column t & 15)
    const int ecb = tid >> 8, erow = (tid >> 4) & 15, ec = 16 * ecb + (tid & 15);
    const float b3v = n.b3[min(ec, ac - 1)];   // (ahead of the barrier: its round trip hides behind the head)
    stage_head_commit(hs, ac, slot);
    const int nblk = ac > 16 ? 2 : 1;
    {
        const int cb = wv & (nblk - 1), kp = nblk == 2 ? wv >> 1 : wv;
        const int jc = min(16 * cb + r, ac - 1);
        const f32x4 acc = nblk == 2 ? actor_head_part<16>(L.T2, slot, ac, 64 * kp, jc)
                                    : actor_head_part<8>(L.T2, slot, ac, 32 * kp, jc);
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wv * 256 + ((lane >> 4) * 4 + j) * 16 + r] = acc[j];
    }
    bar();
    if (ecb < nblk && ec < ac) {
        float sum = 0.f;
        const int e = erow * 16 + (tid & 15);
        if (nblk == 2) {
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) sum += part[(2 * kp + ecb) * 256 + e];
        } else {
#pragma unroll
            for (int kp = 0; kp < 8; ++kp) sum += part[kp * 256 + e];
        }
        float v = high * tanhf(sum + b3v);
        v = fminf(fmaxf(v, -high), high);
        L.X[erow * L.px + ob + ec] = v;
        if (hbm) hbm[(row0 + erow) * (ob + ac) + ob + ec] = v;
    }
    bar();
}

// ------------------------------------------------------------------------------------------------
// TD targets only (return normalisation + PopArt / n-step).  pass 0: the 1-step target of the sampled rows;
// pass 1: the n-step target (rows n_idx, dense n_rews / n_dones / n_len).
template <bool CW, bool LN>
__device__ inline void role_t(const AgentArgs& a, int wg, float* lds, int pass) {
    const int tid = threadIdx.x;
    const bool cw = is_row();   // the 256 row threads; products and their epilogues run on all eight waves
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    Lds L = carve(lds, in_c);
    const int k32c = ((in_c + 31) / 32) * 32;
    const int64_t* idx = pass ? a.n_idx : a.idx;
    const int64_t pr = ring_row(idx, a.ring_start, a.ring_len, row0 + (tid >> 5));
    WPipe pipe;
    wpipe_init(pipe);
    const WDesc ta1 = w_nn(a.targ_actor.W1p, ob), ta2 = w_nn(a.targ_actor.W2p, H);
    const WDesc tc1 = w_nn(a.targ_critic.W1p, in_c), tc2 = w_nn(a.targ_critic.W2p, H);
    if (cw && tid < R) {
        if (pass) {
            L.rw[tid] = a.n_rews[row0 + tid]; L.dn[tid] = a.n_dones[row0 + tid]; L.tl[tid] = a.n_len[row0 + tid];
        } else {
            const int64_t prs = ring_row(idx, a.ring_start, a.ring_len, row0 + tid);
            L.rw[tid] = a.rews[prs];
            L.dn[tid] = a.dones1[prs];
            L.tl[tid] = a.td_len ? a.td_len[row0 + tid] : 1.f;
        }
    }
    load_rows_first(L, k32c, a.obs1, nullptr, a.mean, a.stdv, ob, ac, 0, pr, nullptr, 0, 0);
    bar();
    hfwd<CW, LN>(pipe, ta1, ta2, true, L.X, L.px, a.targ_actor, 1, L.ring, L.T1, nullptr, nullptr, 0, nullptr);
    HeadStage hs;
    stage_head_issue(hs, a.targ_actor.W3, ac);
    hfwd<CW, LN>(pipe, ta2, tc1, true, L.T1, LDT, a.targ_actor, 2, L.ring, L.T2, nullptr, nullptr, 0, nullptr);
    actor_head<CW>(L, a.targ_actor, ob, ac, a.high, nullptr, 0, free_slot(L.ring, pipe), hs, L.T1);
    hfwd<CW, LN>(pipe, tc1, tc2, true, L.X, L.px, a.targ_critic, 1, L.ring, L.T1, nullptr, nullptr, 0, nullptr);
    HeadRegs hd;
    if (cw) head_issue(hd, a.targ_critic.W3, a.targ_critic.b3);
    hfwd<CW, LN>(pipe, tc2, tc2, false, L.T1, LDT, a.targ_critic, 2, L.ring, L.T2, nullptr, nullptr, 0, nullptr);
    if (cw) head1(L.T2, hd, L.s_);
    bar();
    if (cw && tid < R) {
        const bool lens = pass || a.td_len;
        const float gam = lens ? powf(a.gamma, L.tl[tid]) : a.gamma;
        float qp = L.s_[tid];
        if (a.ret_mean) qp = qp * a.ret_std[0] + a.ret_mean[0];              // sam_agent.py:199-201
        (pass ? a.targ_qn : a.targ_q)[row0 + tid] = L.rw[tid] + gam * ((1.f - L.dn[tid]) * qp);
    }
}

template <bool CW, bool LN>
__device__ inline void role_tc(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x;
    const bool cw = is_row();   // the 256 row threads; products and their epilogues run on all eight waves
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    Lds L = carve(lds, in_c);
    const int k32c = ((in_c + 31) / 32) * 32;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 5));
    WPipe pipe;
    wpipe_init(pipe);
    const WDesc ta1 = w_nn(a.targ_actor.W1p, ob), ta2 = w_nn(a.targ_actor.W2p, H);
    const WDesc tc1 = w_nn(a.targ_critic.W1p, in_c), tc2 = w_nn(a.targ_critic.W2p, H);
    const WDesc c1 = w_nn(a.critic.W1p, in_c), c2 = w_nn(a.critic.W2p, H), c3 = w_t(a.critic.W2t, H, 0);
    const bool ext = a.ext_tq != nullptr;   // the targets were computed ahead (launch_targets)
    if (cw && tid < R) {
        L.iw[tid] = a.iws ? a.iws[row0 + tid] : 1.f;
        if (ext) {
            L.tq[tid] = a.ext_tq[row0 + tid];
            L.tl[tid] = a.ext_tqn ? a.ext_tqn[row0 + tid] : 0.f;   // (the n-step target sits in the tl strip)
        } else {
            const int64_t prs = ring_row(a.idx, a.ring_start, a.ring_len, row0 + tid);
            L.rw[tid] = a.rews[prs];
            L.dn[tid] = a.dones1[prs];
            L.tl[tid] = a.td_len ? a.td_len[row0 + tid] : 1.f;
        }
    }
    RowRegs x0;
    HeadRegs hd;
    if (ext) {
        rows_issue(x0, k32c, a.obs0, a.acs, ob, ac, ac, pr);
        stats_stage(L, a.mean, a.stdv, ob);
        bar();
    } else {
    // ---- TD target with the target nets on obs1
    WTRACE(wg, 0, 0);
#ifdef SAMX_TRACE
    if (cw && wg == 0 && tid == 0) g_wtrace[120] = clock64();   // shader clock beside the 100 MHz wall clock
#endif
    load_rows_first(L, k32c, a.obs1, nullptr, a.mean, a.stdv, ob, ac, 0, pr, nullptr, 0, 0);
    bar();
    WTRACE(wg, 0, 1);
    hfwd<CW, LN>(pipe, ta1, ta2, true, L.X, L.px, a.targ_actor, 1, L.ring, L.T1, nullptr, nullptr, 0, nullptr);
    WTRACE(wg, 0, 2);
    HeadStage hs;
    stage_head_issue(hs, a.targ_actor.W3, ac);
    hfwd<CW, LN>(pipe, ta2, tc1, true, L.T1, LDT, a.targ_actor, 2, L.ring, L.T2, nullptr, nullptr, 0, nullptr);
    WTRACE(wg, 0, 3);
    actor_head<CW>(L, a.targ_actor, ob, ac, a.high, nullptr, 0, free_slot(L.ring, pipe), hs, L.T1);
    WTRACE(wg, 0, 4);
    hfwd<CW, LN>(pipe, tc1, tc2, true, L.X, L.px, a.targ_critic, 1, L.ring, L.T1, nullptr, nullptr, 0, nullptr);
    WTRACE(wg, 0, 5);
    // the online critic's input rows travel while the target critic's second layer runs
    rows_issue(x0, k32c, a.obs0, a.acs, ob, ac, ac, pr);
    if (cw) head_issue(hd, a.targ_critic.W3, a.targ_critic.b3);
    hfwd<CW, LN>(pipe, tc2, c1, true, L.T1, LDT, a.targ_critic, 2, L.ring, L.T2, nullptr, nullptr, 0, nullptr);
    WTRACE(wg, 0, 6);
    if (cw) head1(L.T2, hd, L.s_);
    bar();
    WTRACE(wg, 0, 7);
    if (cw && tid < R) {
        const float gam = a.td_len ? powf(a.gamma, L.tl[tid]) : a.gamma;
        const float t = L.rw[tid] + gam * ((1.f - L.dn[tid]) * L.s_[tid]);   // sam_agent.py:202-205
        a.targ_q[row0 + tid] = t;
        L.tq[tid] = t;
    }
    }   // !ext
    // ---- online critic on (obz0, acs): forward, loss, backward
    WTRACE(wg, 0, 14);
    rows_commit(L, x0, k32c, a.mean != nullptr, ob, ac, a.Xc, row0 + (tid >> 5), in_c);
    WTRACE(wg, 0, 15);
    bar();
    WTRACE(wg, 0, 8);
    // (LayerNorm: mask bits per layer in the rows mapping, rs strips in L.red, xh through HBM — see ln_fwd_rows)
    float* rs1 = L.red;
    float* rs2 = L.red + 16;
    const unsigned m1 = hfwd<CW, LN>(pipe, c1, c2, true, L.X, L.px, a.critic, 1, L.ring, L.T1, a.A1c, a.XH1c, row0, rs1);
    WTRACE(wg, 0, 9);
    if (cw) head_issue(hd, a.critic.W3, a.critic.b3);
    const float w3c = a.critic.W3[tid & (H - 1)];   // for head_bwd, ahead of the product
    const unsigned m2 = hfwd<CW, LN>(pipe, c2, c3, true, L.T1, LDT, a.critic, 2, L.ring, L.T2, a.A2c, a.XH2c, row0, rs2);
    WTRACE(wg, 0, 10);
    if (cw) head1(L.T2, hd, L.s_);
    bar();
    if (cw && tid < R) {
        // with return normalisation: de-normalised prediction minus normalised target (sam_agent.py:161-182)
        const float rmean = a.ret_mean ? a.ret_mean[0] : 0.f, rstd = a.ret_std ? a.ret_std[0] : 1.f;
        const float pred = a.ret_mean ? L.s_[tid] * rstd + rmean : L.s_[tid];
        auto hub = [](float td, float& dh) {                              // math_util.py:22-29
            const float ad = fabsf(td);
            dh = ad < 1.f ? td : (td > 0.f ? 1.f : -1.f);
            return ad < 1.f ? 0.5f * td * td : ad - 0.5f;
        };
        const float td = pred - (a.ret_mean ? (L.tq[tid] - rmean) / rstd : L.tq[tid]);
        const float iw = L.iw[tid];
        float dh, dhn = 0.f, hn = 0.f;
        const float h1 = hub(td, dh);
        float dq = (a.td_scale / (float)a.B) * dh;
        if (a.ext_tqn) {                                                  // sam_agent.py:483-498
            const float tdn = pred - (a.ret_mean ? (L.tl[tid] - rmean) / rstd : L.tl[tid]);
            hn = hub(tdn, dhn);
            dq += (a.tdn_scale / (float)a.B) * dhn;
        }
        dq = a.ret_mean ? dq * iw * rstd : dq * iw;
        L.dq[tid] = dq;
        a.row_dq[row0 + tid] = dq;
        a.row_hub[row0 + tid] = h1 * iw;
        if (a.row_hubn) a.row_hubn[row0 + tid] = hn * iw;
        if (a.td_out) a.td_out[row0 + tid] = td;
    }
    bar();
    if constexpr (LN) {
        float* slot = a.w3c_part + (int64_t)wg * (H + 1 + kLnSlot);
        if (cw) head_grad_slot(L.T2, L.dq, slot);   // T2 = n2: the head's input
        bar();
        float* scratch = free_slot(L.ring, pipe);
            ln_bwd_rows<true, true>(L.T2, L.dq, a.critic.W3, a.critic.ga2, a.XH2c, row0, rs2, m2, nullptr, a.DZ2c,
                                    row0, nullptr, 0, scratch);
        bar();
        ln_cols_commit(scratch, slot + (H + 1) + 2 * H);
        prod_to_tile<CW>(pipe, c3, c3, false, L.T2, LDT, L.ring, L.T1);   // cotangent of n1
        scratch = free_slot(L.ring, pipe);
            ln_bwd_rows<false, true>(L.T1, nullptr, nullptr, a.critic.ga1, a.XH1c, row0, rs1, m1, nullptr, a.DZ1c,
                                     row0, nullptr, 0, scratch);
        bar();
        ln_cols_commit(scratch, slot + (H + 1));
        return;
    }
    (void)m1; (void)m2; (void)w3c;
    if (cw) {   // (the same thread reads column t for the head's gradient, then overwrites it: no barrier between)
        head_grad_slot(L.T2, L.dq, a.w3c_part + (int64_t)wg * (H + 1));
        head_bwd<NC>(L.T2, L.dq, w3c, a.DZ2c, row0);
    }
    bar();
    WTRACE(wg, 0, 11);
    {
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, c3, L.T2, LDT, L.ring, acc);
        WTRACE(wg, 0, 12);
        
            for_acc(acc, [&](int r, int c, float v, int) {
                a.DZ1c[(row0 + r) * H + c] = v * dlrelu(L.T1[r * LDT + c]);
            });
    }
    WTRACE(wg, 0, 13);
#ifdef SAMX_TRACE
    if (cw && wg == 0 && tid == 0) g_wtrace[121] = clock64();
#endif
}

template <bool CW, bool LN>
__device__ inline void role_a(const AgentArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool cw = is_row();   // the 256 row threads; products and their epilogues run on all eight waves
    const int64_t row0 = (int64_t)wg * R;
    const int ob = a.ob, ac = a.ac, in_c = ob + ac;
    Lds L = carve(lds, in_c);
    const int k32c = ((in_c + 31) / 32) * 32;
    const int64_t pr = ring_row(a.idx, a.ring_start, a.ring_len, row0 + (tid >> 5));
    WPipe pipe;
    wpipe_init(pipe);
    const WDesc a1 = w_nn(a.actor.W1p, ob), a2 = w_nn(a.actor.W2p, H), a3 = w_t(a.actor.W2t, H, 0);
    const WDesc c1 = w_nn(a.critic.W1p, in_c), c2 = w_nn(a.critic.W2p, H), c3 = w_t(a.critic.W2t, H, 0);
    const WDesc c4 = w_t(a.critic.W1t, H, 0);   // rows ob .. ob+ac of W1c: dQ/da
    if (cw) WTRACE(wg, 1, 0);
    load_rows_first(L, k32c, a.obs0, nullptr, a.mean, a.stdv, ob, ac, 0, pr, a.Xa, row0 + (tid >> 5), in_c);
    for (int t = tid; t < R * PP; t += NT) L.PRE[t] = 0.f;
    bar();
    // (LayerNorm: rs strips of actor 1 / 2 and critic 1 / 2 in L.red; mask bits in the rows mapping)
    float* rs = L.red;
    const unsigned m1a = hfwd<CW, LN>(pipe, a1, a2, true, L.X, L.px, a.actor, 1, L.ring, L.T1, a.A1a, a.XH1a, row0, rs);
    HeadStage hs;
    stage_head_issue(hs, a.actor.W3, ac);
    const unsigned m2a = hfwd<CW, LN>(pipe, a2, c1, true, L.T1, LDT, a.actor, 2, L.ring, L.T2, a.A2a, a.XH2a, row0, rs + 16);
    actor_head<CW>(L, a.actor, ob, ac, a.high, a.Xa, row0, free_slot(L.ring, pipe), hs, L.T1);
    // Q branch: -q_scale * mean Q(obz0, mu(obz0))   (sam_agent.py:412-413)
    const unsigned m1c = hfwd<CW, LN>(pipe, c1, c2, true, L.X, L.px, a.critic, 1, L.ring, L.T1, nullptr, a.XH1q, row0, rs + 32);
    HeadRegs hd;
    if (cw) head_issue(hd, a.critic.W3, a.critic.b3);
    const float w3c = a.critic.W3[tid & (H - 1)];   // for dZ2c below, ahead of the product
    (void)w3c;
    const unsigned m2c = hfwd<CW, LN>(pipe, c2, c3, true, L.T1, LDT, a.critic, 2, L.ring, L.T2, nullptr, a.XH2q, row0, rs + 48);
    (void)m1c; (void)m2c;
    if (cw) {
        head1(L.T2, hd, L.s_);
        // (d(de-normalised Q)/d(critic output) = ret_std, sam_agent.py:184-187)
        if (tid < R) L.dq[tid] = a.ret_std ? -a.q_scale / (float)a.B * a.ret_std[0] : -a.q_scale / (float)a.B;
    }
    bar();
    if constexpr (LN) {
        if (cw && tid < R) a.row_q[row0 + tid] = L.s_[tid];
        ln_bwd_rows<true, false>(L.T2, L.dq, a.critic.W3, a.critic.ga2, a.XH2q, row0, rs + 48, m2c, nullptr, nullptr,
                                 0, nullptr, 0, nullptr);   // dZ2c in place
        bar();
        prod_to_tile<CW>(pipe, c3, c4, true, L.T2, LDT, L.ring, L.T1);   // cotangent of the critic's n1
            ln_bwd_rows<false, false>(L.T1, nullptr, nullptr, a.critic.ga1, a.XH1q, row0, rs + 32, m1c, nullptr, nullptr,
                                      0, nullptr, 0, nullptr);   // dZ1c in place
        bar();
    } else {
    if (cw) {
        if (tid < R) a.row_q[row0 + tid] = L.s_[tid];
        for (int t = tid; t < R * H; t += NC) {   // dZ2c in place
            const int r = t >> 8, c = t & 255;
            L.T2[r * LDT + c] = L.dq[r] * w3c * dlrelu(L.T2[r * LDT + c]);
        }
    }
    bar();
    {   // dZ1c = (dZ2c W2c^T) .* lrelu'(A1c), in place over A1c (every wave is past the product's last barrier)
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, c3, c4, true, L.T2, LDT, L.ring, acc);
        for_acc(acc, [&](int r, int c, float v, int) { L.T1[r * LDT + c] = v * dlrelu(L.T1[r * LDT + c]); });
    }
    bar();
    }
    {   // dQ/da = dZ1c W1c[ob:ob+ac, :]^T, through clip (identity inside the range) and tanh
        f32x4 acc[NB];
        zero_acc(acc);
        stage_head_issue(hs, a.actor.W3, ac);   // for the backward through the action head below
        wprod<CW>(pipe, c4, a3, true, L.T1, LDT, L.ring, acc);
        if (cw && wv == 0) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int c = 16 * nb + (lane & 15);   // (for_acc's column numbering, wave 0)
                if (c < ac) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = (lane >> 4) * 4 + j;
                        const float t = L.X[r * L.px + ob + c] / a.high;
                        const float v = acc[nb][j] * (a.high * (1.f - t * t));
                        L.PRE[r * PP + c] = v;
                        a.dPre[(row0 + r) * ac + c] = v;
                    }
                }
            }
        }
    }
    bar();
    {   // dZ2a = (PRE W3a^T) .* lrelu'(A2a): reduction over the (zero-padded) action index
        float* slot = free_slot(L.ring, pipe);
        stage_head_commit(hs, ac, slot);
        const int r = lane & 15, kq = lane >> 4;
        f32x4 acc[NB];
        zero_acc(acc);
        const int ks = (ac + 3) / 4;
        for (int kk = 0; kk < ks; ++kk) {   // (every wave: its 32 columns)
            const float av = L.PRE[r * PP + 4 * kk + kq];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = mfma4(av, slot[(16 * NB * wv + 16 * nb + r) * ac + 4 * kk + kq], acc[nb]);
        }
        bar();
        if constexpr (LN) {
            // the cotangent of the actor's n2 (raw), then back through LayerNorm 2, W2a^T, LayerNorm 1
            for_acc(acc, [&](int r_, int c, float v, int) { L.T2[r_ * LDT + c] = v; });
            bar();
            float* lslot = a.lna_part + (int64_t)wg * kLnSlot;
                ln_bwd_rows<false, true>(L.T2, nullptr, nullptr, a.actor.ga2, a.XH2a, row0, rs + 16, m2a, nullptr,
                                         a.DZ2a, row0, nullptr, 0, slot);   // (the head matrix in `slot` is dead)
            bar();
            ln_cols_commit(slot, lslot + 2 * H);
            prod_to_tile<CW>(pipe, a3, a3, false, L.T2, LDT, L.ring, L.T1);
            float* scratch = free_slot(L.ring, pipe);
                ln_bwd_rows<false, true>(L.T1, nullptr, nullptr, a.actor.ga1, a.XH1a, row0, rs, m1a, nullptr, a.DZ1a,
                                         row0, nullptr, 0, scratch);
            bar();
            ln_cols_commit(scratch, lslot);
            return;
        }
        
            for_acc(acc, [&](int r_, int c, float v, int bi) {
                const float z = v * ((m2a >> bi) & 1u ? 1.f : 0.2f);
                L.T2[r_ * LDT + c] = z;
                a.DZ2a[(row0 + r_) * H + c] = z;
            });
    }
    bar();
    {   // dZ1a = (dZ2a W2a^T) .* lrelu'(A1a)
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, a3, L.T2, LDT, L.ring, acc);
        
            for_acc(acc, [&](int r, int c, float v, int bi) {
                a.DZ1a[(row0 + r) * H + c] = v * ((m1a >> bi) & 1u ? 1.f : 0.2f);
            });
    }
    if (cw) WTRACE(wg, 1, 1);
}

template <bool CW, bool LN>
__device__ inline void role_d(const DiscArgs& a, int wg, float* lds) {
    const int tid = threadIdx.x;
    const bool cw = is_row();   // the 256 row threads; products and their epilogues run on all eight waves
    const int tiles_b = (int)(a.B / R);
    // interpolate tiles first (the longest chain), then policy, then expert
    const int sect = wg < tiles_b ? 2 : (wg < 2 * tiles_b ? 0 : 1);
    const int64_t row0 = (int64_t)(wg % tiles_b) * R;   // row inside the section
    const int64_t hrow0 = (int64_t)sect * a.B + row0;   // row of the stacked [policy | expert | interpolate] buffers
    const int ob = a.ob, ac = a.ac, in = ob + ac;
    const Net& n = a.net;
    Lds L = carve(lds, in);
    const int k32 = ((in + 31) / 32) * 32;
    WPipe pipe;
    wpipe_init(pipe);
    const WDesc d1 = w_nn(n.W1p, in), d2 = w_nn(n.W2p, H), d3 = w_t(n.W2t, H, 0);
    const int tsec = sect == 2 ? 0 : (sect == 0 ? 2 : 4);
    (void)tsec;   // only the SAMX_TRACE build reads it
    if (cw) WTRACE(wg % tiles_b, 2, tsec);
    // statistics the rows are normalised with, staged in LDS (L.st: mean, 1 / std).
    // d.obs_rms.update folded into the step: the batch's float64 column sums were computed ahead, every
    // workgroup derives the mean/std that INCLUDE this batch from sums + addvec — the same arithmetic as the
    // commit behind this launch (nets.hip wide_finish_kernel), hence the same bits.
    const bool norm = a.rms_sums != nullptr || a.mean != nullptr;
    // Every row load goes out FIRST (rows_issue: index word -> 13 + 13 row elements, two dependent HBM round
    // trips); the statistics and the interpolation weights below are computed while they fly.
    RowRegs xp, xe;
    const int64_t g = row0 + (tid >> 5);   // this thread's row inside the section
    {   // rows: policy (ring rows), expert (pool slice)
        const int64_t pg = ring_row(a.p_idx, a.ring_start, a.ring_len, g);
        const int64_t eg = (a.e_start ? a.e_start[0] : 0) + g;
        if (sect != 1) rows_issue(xp, k32, a.p_obs, a.p_acs, ob, ac, ac, pg);
        if (sect != 0) rows_issue(xe, k32, a.e_obs, a.e_acs, ob, ac, ac, eg);
    }
    // The interpolation weights of this tile into LDS, E[r * ob + c] then E[16 ob + r * ac + c'].  The tile's elements
    // of either tensor are one contiguous range — rows row0 .. row0 + 15 of a [B][ob] / [B][ac] tensor, as
    // samx_disc_randoms numbers them — so when they are drawn here all eight waves draw them four per Philox.
    float* E = L.ring;
#define PTR(i) do { if (cw && sect == 2) WTRACE(wg, 2, 16 + (i)); } while (0)   // interpolate tile 0: prologue stamps
    PTR(0);
    if (sect == 2) {
        const uint64_t step = a.eps_ob ? 0 : (uint64_t)a.rng_step[0];
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const int w = sidx ? ac : ob;
            float* Es = sidx ? E + R * ob : E;
            const int64_t e0 = row0 * w;
            if (a.eps_ob) {   // injected tensors
                const float* src = (sidx ? a.eps_ac : a.eps_ob) + e0;
                for (int e = tid; e < R * w; e += NT) Es[e] = src[e];
            } else {
                const int64_t lo = e0 >> 2, hi = (e0 + (int64_t)R * w - 1) >> 2;
                for (int64_t grp = lo + tid; grp <= hi; grp += NT) {
                    float u[4];
                    philox_uniform4(a.rng_seed, step, sidx ? RNG_EPS_AC : RNG_EPS_OB, (uint64_t)grp, u);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int64_t e = 4 * grp + i - e0;
                        if (e >= 0 && e < (int64_t)R * w) Es[e] = u[i];
                    }
                }
            }
        }
    }
    PTR(1);
    if (norm) {
        for (int c = tid; c < ob; c += NT) {   // every thread of the workgroup
            float gm, gs;
            if (a.rms_sums) {
                const double cnt = a.rms_sums[2 * ob] + a.rms_addvec[2 * ob];
                const double s1 = __dadd_rn(a.rms_sums[c], a.rms_addvec[c]),
                             s2 = __dadd_rn(a.rms_sums[ob + c], a.rms_addvec[ob + c]);
                const float mean = (float)(s1 / cnt), msq = (float)(s2 / cnt);
                gm = mean;
                gs = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(mean, mean)), 1e-2f));
            } else {
                gm = a.mean[c]; gs = a.stdv[c];
            }
            L.st[2 * c] = gm;
            L.st[2 * c + 1] = 1.f / gs;
        }
    }
    PTR(2);
    bar();   // statistics and interpolation weights are in LDS
    PTR(3);
    {   // eps p + (1 - eps) e  (discriminator.py:129-134): the rows are normalised and written (branch-free,
        // the HBM copy last — see rows_commit)
        const int r = tid >> 5;
        if (sect == 0) rows_commit(L, xp, k32, norm, ob, ac, a.X, hrow0 + r, in);
        else if (sect == 1) rows_commit(L, xe, k32, norm, ob, ac, a.X, hrow0 + r, in);
        else {
            // interpolates of the NORMALISED rows; X receives V later, nothing goes to HBM here
            const int t = opaque_tid();
            const int c0 = t & 31;
            float* xr = L.X + (t >> 5) * L.px + c0;
#pragma unroll
            for (int u = 0; u < kRowRegs; ++u) {
                const int c = c0 + 32 * u;
                const float pv = row_norm(L, xp.v[u], c, ob, ac, norm), ev = row_norm(L, xe.v[u], c, ob, ac, norm);
                // (columns >= in read a valid word; pv = ev = 0 there)
                const float eps = c < ob ? E[r * ob + c] : E[R * ob + r * ac + min(c - ob, ac - 1)];
                const float v = eps * pv + (1.f - eps) * ev;
                xr[32 * u] = v;
                xp.v[u] = v;
            }
            PTR(4);
            if constexpr (LN) {   // the primal stream's layer input (x^T P_z1 of the weight gradient)
                float* hr = a.X + (hrow0 + (t >> 5)) * in + c0;
#pragma unroll
                for (int u = 0; u < kRowRegs; ++u)
                    if (c0 + 32 * u < in) hr[32 * u] = xp.v[u];
            }
        }
    }
    bar();
    if constexpr (LN) {
        // ================= LayerNorm networks (mini-phases: ln_fwd_rows / ln_bwd_rows / ln_tan_rows) =================
        float* rs1 = L.red + 32;   // (L.red[0..32) takes the interpolate tiles' sums of squares)
        float* rs2 = L.red + 48;
        float* slot = a.w3_part + (int64_t)wg * (H + 1 + kLnSlot);
        if (sect != 2) {
            // ---- policy / expert rows: forward, BCE terms, backward with the LayerNorm gradients of the tile
            const unsigned m1 = hidden_fwd_ln<CW>(pipe, d1, d2, true, L.X, L.px, n.b1, n.be1, n.ga1, L.ring, L.T1, a.XH1,
                                                  a.A1, hrow0, rs1);
            HeadRegs hd;
            if (cw) head_issue(hd, n.W3, n.b3);
            const unsigned m2 = hidden_fwd_ln<CW>(pipe, d2, d3, true, L.T1, LDT, n.b2, n.be2, n.ga2, L.ring, L.T2, a.XH2,
                                                  a.A2, hrow0, rs2);
            if (cw) head1(L.T2, hd, L.s_);
            bar();
            if (cw && tid < R) {
                const int64_t g = row0 + tid;
                const float s = L.s_[tid];
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
                const float dq = (sg - z) / (float)a.B + a.ent_scale * s * sg * (1.f - sg) / (float)(2 * a.B);
                L.dq[tid] = dq;
                a.DS[hrow0 + tid] = dq;
                a.ce[hrow0 + tid] = sig_ce(s, z);
                a.ent[hrow0 + tid] = sig_ce(s, sg);
                a.sg[hrow0 + tid] = sg;
            }
            bar();
            if (cw) head_grad_slot(L.T2, L.dq, slot);   // T2 = n2: the head's input
            bar();
            float* scratch = free_slot(L.ring, pipe);
                ln_bwd_rows<true, true>(L.T2, L.dq, n.W3, n.ga2, a.XH2, hrow0, rs2, m2, nullptr, a.DZ2, hrow0, nullptr, 0,
                                        scratch);
            bar();
            ln_cols_commit(scratch, slot + (H + 1) + 2 * H);
            prod_to_tile<CW>(pipe, d3, d3, false, L.T2, LDT, L.ring, L.T1);   // cotangent of n1
            scratch = free_slot(L.ring, pipe);
                ln_bwd_rows<false, true>(L.T1, nullptr, nullptr, n.ga1, a.XH1, hrow0, rs1, m1, nullptr, a.DZ1, hrow0,
                                         nullptr, 0, scratch);
            bar();
            ln_cols_commit(scratch, slot + (H + 1));
            return;
        }
        // ---- interpolate rows.  The gradient penalty is no longer piecewise linear: next to the tangent stream T
        // (the first backward that yields g, then the forward along V) a primal cotangent stream P runs back from the
        // two tangent LayerNorms.  HBM rows: hP = [2B, 3B) the primal stream (X = x, A1 = n1,
        // DZ* = P_z*), hT = [3B, 4B) the tangent stream (X = V, A1 = ndot1, DZ* = T_z*).
        const int64_t hP = hrow0, hT = hrow0 + a.B;
        const WDesc g0 = w_t(n.W1t, 2 * H, 0), g1 = w_t(n.W1t, 2 * H, H);
        const WDesc t1 = w_nn(n.W1p, in), t2 = w_nn(n.W2p, H);
#define LTR(i) do { if (cw) WTRACE(wg, 2, 6 + (i)); } while (0)   // interpolate tile 0 with LayerNorm: phase stamps 6 ..
        LTR(0);
        const unsigned m1 = hidden_fwd_ln<CW>(pipe, d1, d2, true, L.X, L.px, n.b1, n.be1, n.ga1, L.ring, L.T1, a.XH1,
                                              a.A1, hP, rs1);
        LTR(1);
        const unsigned m2 = hidden_fwd_ln<CW>(pipe, d2, d3, true, L.T1, LDT, n.b2, n.be2, n.ga2, L.ring, L.T2, a.XH2,
                                              nullptr, hP, rs2);
        LTR(2);
        if (cw && tid < R) { L.dq[tid] = 1.f; a.DS[hrow0 + tid] = 1.f; }
        bar();
        // T_z2: the head's cotangent is w3 (d score = 1)
            ln_bwd_rows<true, false>(L.T2, L.dq, n.W3, n.ga2, a.XH2, hP, rs2, m2, nullptr, a.DZ2, hT, nullptr, 0, nullptr);
        bar();
        LTR(3);
        prod_to_tile<CW>(pipe, d3, g0, true, L.T2, LDT, L.ring, L.T1);   // T_n1
        LTR(4);
        // T_z1 in place; T_n1 kept for the tangent LayerNorm of layer 1 (same-thread read-back)
            ln_bwd_rows<false, false>(L.T1, nullptr, nullptr, n.ga1, a.XH1, hP, rs1, m1, nullptr, a.DZ1, hT, a.TN1, row0,
                                      nullptr);
        bar();
        LTR(5);
        {   // g = T_z1 W1^T, 256 input columns at a time, over X (x is in HBM by now)
            auto put = [&](const f32x4 (&acc)[NB], int j0) {
                
                    for_acc(acc, [&](int r, int c, float v, int) {
                        if (j0 + c < k32) L.X[r * L.px + j0 + c] = j0 + c < in ? v : 0.f;
                    });
            };
            f32x4 acc[NB];
            zero_acc(acc);
            if (in > H) {
                wprod<CW>(pipe, g0, g1, true, L.T1, LDT, L.ring, acc);
                put(acc, 0);
                zero_acc(acc);
                wprod<CW>(pipe, g1, t1, true, L.T1, LDT, L.ring, acc);
                put(acc, H);
            } else {
                wprod<CW>(pipe, g0, t1, true, L.T1, LDT, L.ring, acc);
                put(acc, 0);
            }
        }
        bar();
        {   // (all threads: 32 per row)
            const int r = tid >> 5, q = tid & 31;
            float so = 0.f, sa = 0.f;
            for (int c = q; c < in; c += 32) {
                const float g = L.X[r * L.px + c];
                if (c < ob) so = fmaf(g, g, so); else sa = fmaf(g, g, sa);
            }
            so += __shfl_xor(so, 16, 64); sa += __shfl_xor(sa, 16, 64);
            so = sum16(so); sa = sum16(sa);
            if (q == 0) { L.red[r] = so; L.red[16 + r] = sa; }
        }
        bar();
        if (cw && tid == 0) {
            double o = 0.0, c = 0.0;
            for (int r = 0; r < R; ++r) { o += (double)L.red[r]; c += (double)L.red[16 + r]; }
            a.gp_part[2 * (wg % tiles_b)] = o;
            a.gp_part[2 * (wg % tiles_b) + 1] = c;
        }
        {   // (all threads: 32 per row)
            const float co = 1.f / ((float)a.B * (float)ob), ca = 1.f / ((float)a.B * (float)ac);
            const int r = tid >> 5, c0 = tid & 31;
            for (int c = c0; c < in; c += 32) {
                const float v = L.X[r * L.px + c] * (c < ob ? co : ca);
                L.X[r * L.px + c] = v;
                a.X[(hT + r) * in + c] = v;   // V: the tangent stream's layer input
            }
        }
        bar();
        LTR(6);
        prod_to_tile<CW>(pipe, t1, t2, true, L.X, L.px, L.ring, L.T1);   // zdot1 = V W1
        LTR(7);
        float* scratch = free_slot(L.ring, pipe);
        // ndot1 -> T1 and A1[hT]; PA1 (unmasked) -> HBM; column sums of tn1 u1 (part of dgamma1)
            ln_tan_rows<false>(L.T1, n.ga1, a.XH1, hP, rs1, m1, a.TN1, nullptr, row0, a.A1, hT, a.PA1, scratch);
        bar();
        LTR(8);
        if (cw) slot[(H + 1) + H + tid] = ln_col8(scratch, 256 + tid);
        prod_to_tile<CW>(pipe, t2, d3, true, L.T1, LDT, L.ring, L.T2);   // zdot2 = ndot1 W2
        LTR(9);
        scratch = free_slot(L.ring, pipe);
        // P_z2 -> T2 and DZ2[hP]; column sums of ndot2 (-> dW3) and of tn2 u2 (-> dgamma2)
            ln_tan_rows<true>(L.T2, n.ga2, a.XH2, hP, rs2, m2, nullptr, n.W3, row0, a.DZ2, hP, nullptr, scratch);
        bar();
        if (cw) {
            slot[tid] = ln_col8(scratch, tid);                              // sum ndot2 (without k)
            slot[(H + 1) + 2 * H + tid] = 0.f;                               // dbeta2
            slot[(H + 1) + 3 * H + tid] = ln_col8(scratch, 256 + tid);     // dgamma2
            if (tid == 0) slot[H] = 0.f;
        }
        LTR(10);
        prod_to_tile<CW>(pipe, d3, d3, false, L.T2, LDT, L.ring, L.T1);   // P_n1 = P_z2 W2^T
        LTR(11);
        scratch = free_slot(L.ring, pipe);
        // P_z1 = J1(gamma1 P_n1) + PA1, masked -> DZ1[hP]; dbeta1, dgamma1 (P part)
            ln_bwd_rows<false, true>(L.T1, nullptr, nullptr, n.ga1, a.XH1, hP, rs1, m1, a.PA1, a.DZ1, hP, nullptr, row0,
                                     scratch);
        bar();
        if (cw) {
            slot[(H + 1) + tid] = ln_col8(scratch, tid);
            slot[(H + 1) + H + tid] += ln_col8(scratch, 256 + tid);   // (this thread stored the tangent part above)
        }
        LTR(12);
        if (cw) WTRACE(wg, 2, 1);
        return;
    }
    if (sect != 2) {
        // ---- policy / expert rows
        hidden_fwd<CW>(pipe, d1, d2, true, L.X, L.px, n.b1, L.ring, L.T1, a.A1, hrow0);
        HeadRegs hd;
        if (cw) head_issue(hd, n.W3, n.b3);
        const float w3c = n.W3[tid & (H - 1)];   // for head_bwd, ahead of the product
        hidden_fwd<CW>(pipe, d2, d3, true, L.T1, LDT, n.b2, L.ring, L.T2, a.A2, hrow0);
        if (cw) head1(L.T2, hd, L.s_);
        bar();
        if (cw && tid < R) {
            const int64_t g = row0 + tid;
            const float s = L.s_[tid];
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
            const float dq = (sg - z) / (float)a.B + a.ent_scale * s * sg * (1.f - sg) / (float)(2 * a.B);
            L.dq[tid] = dq;
            a.DS[hrow0 + tid] = dq;
            a.ce[hrow0 + tid] = sig_ce(s, z);
            a.ent[hrow0 + tid] = sig_ce(s, sg);
            a.sg[hrow0 + tid] = sg;
        }
        bar();
        if (cw) {
            head_grad_slot(L.T2, L.dq, a.w3_part + (int64_t)wg * (H + 1));
            head_bwd<NC>(L.T2, L.dq, w3c, a.DZ2, hrow0);
        }
        bar();
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, d3, L.T2, LDT, L.ring, acc);
        
            for_acc(acc, [&](int r, int c, float v, int) {
                a.DZ1[(hrow0 + r) * H + c] = v * dlrelu(L.T1[r * LDT + c]);
            });
        if (cw) WTRACE(wg % tiles_b, 2, tsec + 1);
        return;
    }
    // ---- interpolate rows: g = d score / d input, sum g^2, V = g / (B dim), tangent forward along V.
    // The scalar k = 10 * 2 (N - 1) / N needs every tile's sums: it is applied to these rows' share of the
    // weight gradients after this launch, not here.
    const WDesc g0 = w_t(n.W1t, 2 * H, 0), g1 = w_t(n.W1t, 2 * H, H);
    const WDesc t1 = w_nn(n.W1p, in), t2 = w_nn(n.W2p, H);
#define DTR(i) do { if (cw) WTRACE(wg, 2, 6 + (i)); } while (0)   // interpolate tile 0: phase stamps 6 ..
    DTR(0);
    hidden_fwd<CW>(pipe, d1, d2, true, L.X, L.px, n.b1, L.ring, L.T1, nullptr, 0);                       // A1 (kept: mask)
    DTR(1);
    const float w3c = n.W3[tid & (H - 1)];   // for head_bwd, ahead of the product
    const unsigned m2 = hidden_fwd<CW>(pipe, d2, d3, true, L.T1, LDT, n.b2, L.ring, L.T2, nullptr, 0);    // A2 -> mask bits
    DTR(2);
    if (cw && tid < R) { L.dq[tid] = 1.f; a.DS[hrow0 + tid] = 1.f; }
    bar();
    head_bwd<NT>(L.T2, L.dq, w3c, a.DZ2, hrow0);                                                 // gz2 (all threads)
    bar();
    DTR(3);
    {   // gz1 = (gz2 W2^T) .* m1, in place over gz2
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, d3, g0, true, L.T2, LDT, L.ring, acc);
        
            for_acc(acc, [&](int r, int c, float v, int) {
                const float z = v * dlrelu(L.T1[r * LDT + c]);
                L.T2[r * LDT + c] = z;
                a.DZ1[(hrow0 + r) * H + c] = z;
            });
    }
    bar();
    DTR(4);
    {   // g = gz1 W1^T, 256 input columns at a time, over X (dead by now)
        auto put = [&](const f32x4 (&acc)[NB], int j0) {
            
                for_acc(acc, [&](int r, int c, float v, int) {
                    if (j0 + c < k32) L.X[r * L.px + j0 + c] = j0 + c < in ? v : 0.f;
                });
        };
        f32x4 acc[NB];
        zero_acc(acc);
        if (in > H) {
            wprod<CW>(pipe, g0, g1, true, L.T2, LDT, L.ring, acc);
            put(acc, 0);
            DTR(5);
            zero_acc(acc);
            wprod<CW>(pipe, g1, t1, true, L.T2, LDT, L.ring, acc);
            put(acc, H);
        } else {
            wprod<CW>(pipe, g0, t1, true, L.T2, LDT, L.ring, acc);
            put(acc, 0);
        }
    }
    bar();
    DTR(6);
    {   // (all threads: 32 per row)
        const int r = tid >> 5, q = tid & 31;
        float so = 0.f, sa = 0.f;
        for (int c = q; c < in; c += 32) {
            const float g = L.X[r * L.px + c];
            if (c < ob) so = fmaf(g, g, so); else sa = fmaf(g, g, sa);
        }
        so += __shfl_xor(so, 16, 64); sa += __shfl_xor(sa, 16, 64);
        so = sum16(so); sa = sum16(sa);
        if (q == 0) { L.red[r] = so; L.red[16 + r] = sa; }
    }
    bar();
    DTR(7);
    if (cw && tid == 0) {
        double o = 0.0, c = 0.0;
        for (int r = 0; r < R; ++r) { o += (double)L.red[r]; c += (double)L.red[16 + r]; }
        a.gp_part[2 * (wg % tiles_b)] = o;
        a.gp_part[2 * (wg % tiles_b) + 1] = c;
    }
    {   // (all threads: 32 per row)
        const float co = 1.f / ((float)a.B * (float)ob), ca = 1.f / ((float)a.B * (float)ac);
        const int r = tid >> 5, c0 = tid & 31;
        for (int c = c0; c < in; c += 32) {
            const float v = L.X[r * L.px + c] * (c < ob ? co : ca);
            L.X[r * L.px + c] = v;
            a.X[(hrow0 + r) * in + c] = v;   // V takes the interpolates' place among the layer inputs
        }
    }
    bar();
    DTR(8);
    {   // a1dot = (V W1) .* m1, in place over A1
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, t1, t2, true, L.X, L.px, L.ring, acc);
        
            for_acc(acc, [&](int r, int c, float v, int) {
                const float z = v * dlrelu(L.T1[r * LDT + c]);
                L.T1[r * LDT + c] = z;
                a.A1[(hrow0 + r) * H + c] = z;
            });
    }
    bar();
    DTR(9);
    {   // a2dot = (a1dot W2) .* m2
        f32x4 acc[NB];
        zero_acc(acc);
        wprod<CW>(pipe, t2, L.T1, LDT, L.ring, acc);
        
            for_acc(acc, [&](int r, int c, float v, int bi) {
                a.A2[(hrow0 + r) * H + c] = v * ((m2 >> bi) & 1u ? 1.f : 0.2f);
            });
        // head gradient of these rows: sum_r a2dot[r][c] (without k), rows in a fixed order
        float* slot = a.w3_part + (int64_t)wg * (H + 1);
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {   // (every wave: its 32 columns)
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sm += acc[nb][j] * ((m2 >> (nb * 4 + j)) & 1u ? 1.f : 0.2f);
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            if (lane < 16) slot[16 * NB * wv + 16 * nb + lane] = sm;
        }
        if (tid == 0) slot[H] = 0.f;
    }
    if (cw) WTRACE(wg % tiles_b, 2, 1);
}

// Placement: blocks b and b + 8 share an XCD (and its 4 MiB L2), so the 8 residue classes of blockIdx.x are
// dealt to the roles — classes [0, g_tc) run target+critic tiles, [g_tc, g_tc + g_a) actor tiles, the rest
// discriminator tiles — and an XCD's L2 only has to hold the weights of ONE role (with every role on every
// XCD the five networks plus the activation stores overflow it and the weight stream comes from the Infinity
// Cache at less than half the rate).  Affinity only: nothing depends on where a block runs.
struct RoleArgs {
    AgentArgs a;
    DiscArgs d;
    int n_tc, n_a, n_d;
    int g_tc, g_a, per;   // residue classes per role; blocks per class
};

// partial sums of squares of the decayed kernels: workgroup b of n takes elements b * NT + tid, + n * NT, ...
__device__ inline void role_wd(const AgentArgs& a, int b) {
    __shared__ double sh[NT / 64];
    const int64_t n1 = (int64_t)(a.ob + a.ac) * H, n2 = (int64_t)H * H;
    double s = 0.0;
    for (int64_t i = (int64_t)b * NT + threadIdx.x; i < n1 + n2; i += (int64_t)a.wd_blocks * NT) {
        const double w = (double)(i < n1 ? a.critic.W1[i] : a.critic.W2[i - n1]);
        s += w * w;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < NT / 64; ++i) t += sh[i];
        a.wd_part[b] = t;
    }
}

template <bool LN>
__global__ __launch_bounds__(NT, 1) void roles_kernel(RoleArgs s) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x;
    if (b >= 8 * s.per) { role_wd(s.a, b - 8 * s.per); return; }
    const int g = b & 7, j = b >> 3;
    if (g < s.g_tc) {
        const int wg = j * s.g_tc + g;
        if (wg >= s.n_tc) return;
        role_tc<true, LN>(s.a, wg, lds);
    } else if (g < s.g_tc + s.g_a) {
        const int wg = j * s.g_a + (g - s.g_tc);
        if (wg >= s.n_a) return;
        role_a<true, LN>(s.a, wg, lds);
    } else {
        const int wg = j * (8 - s.g_tc - s.g_a) + (g - s.g_tc - s.g_a);   // interpolate tiles come first
        if (wg >= s.n_d) return;
        role_d<true, LN>(s.d, wg, lds);
    }
}

template <bool LN>
__global__ __launch_bounds__(NT, 1) void targets_kernel(AgentArgs a, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pass = blockIdx.x / tiles, wg = blockIdx.x % tiles;
    role_t<true, LN>(a, wg, lds, pass);
}

// k-interleaved image of a [K][256] matrix (see WDesc); rows beyond K are zero.
constexpr int kPackMax = 16;
struct PackArgs {
    const float* W[kPackMax];
    float* dst[kPackMax];
    int K[kPackMax];
    int J[kPackMax];      // 0: k-interleaved image; > 0: transposed image with J column slots
    int first[kPackMax + 1];
    int n;
};
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    // thread = (4-row group kg, 4 columns j0 .. j0+3): four 16-byte loads (one per row), a 4 x 4 transposition in
    // registers, 64 contiguous bytes out; a block covers 4 row groups
    int p = 0;
    const int b = blockIdx.x;
    while (p + 1 < a.n && b >= a.first[p + 1]) ++p;
    const int kg = 4 * (b - a.first[p]) + (threadIdx.x >> 6), j0 = 4 * (threadIdx.x & 63), K = a.K[p];
    const float* W = a.W[p];
    f32x4 r[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k = 4 * kg + m;
        r[m] = k < K ? *(const f32x4u*)(W + (int64_t)k * H + j0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.J[p]) {   // transposed image: (row, columns j0 .. j0 + 3) is one 16-byte element, the four rows contiguous
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (4 * kg + m < K) *(f32x4*)(a.dst[p] + ((int64_t)(j0 >> 2) * a.J[p] + 4 * kg + m) * 4) = r[m];
        return;
    }
    float* o = a.dst[p] + ((int64_t)kg * H + j0) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) *(f32x4*)(o + 4 * c) = f32x4{r[0][c], r[1][c], r[2][c], r[3][c]};
}

bool net_ok(const samx_mlp* d) { return d && d->h1 == kH && d->h2 == kH; }

}  // namespace

bool supported(const samx_mlp* actor, const samx_mlp* critic, const samx_mlp* reward, int ob, int ac, int64_t B) {
    // (LayerNorm: all of a learner's networks or none, hps.with_layernorm)
    return net_ok(actor) && net_ok(critic) && net_ok(reward) && !actor->layernorm == !critic->layernorm &&
           !actor->layernorm == !reward->layernorm && ob > 0 && ac > 0 && ac <= 32 &&
           ob + ac <= kMaxIn && B > 0 && B % kRows == 0 && actor->in_dim == ob && actor->out_dim == ac &&
           critic->in_dim == ob + ac && critic->out_dim == 1 && reward->in_dim == ob + ac && reward->out_dim == 1 &&
           lds_floats(ob + ac) * sizeof(float) <= 160 * 1024;
}

int launch_roles(const AgentArgs* a, const DiscArgs* d, int roles, hipStream_t st) {
    RoleArgs s;
    s.n_tc = s.n_a = s.n_d = 0;
    int in = 0;
    if (a && (roles & 3)) {
        s.a = *a;
        if (roles & 2) s.n_tc = (int)(a->B / kRows);
        if (roles & 1) s.n_a = (int)(a->B / kRows);
        in = a->ob + a->ac;
    }
    if (d && (roles & 4)) {
        s.d = *d;
        s.n_d = 3 * (int)(d->B / kRows);
        in = d->ob + d->ac;
    }
    const int n_wd = (a && (roles & 2) && a->wd_part) ? a->wd_blocks : 0;
    if (s.n_tc + s.n_a + s.n_d == 0) return SAMX_OK;
    {   // residue classes in proportion to the roles' tiles (each present role gets at least one)
        const int n[3] = {s.n_tc, s.n_a, s.n_d};
        int g[3] = {0, 0, 0}, left = 8;
        for (int i = 0; i < 3; ++i)
            if (n[i] > 0) { g[i] = 1; --left; }
        while (left > 0) {   // one more class to the role with the most tiles per class
            int best = 0;
            double load = -1.0;
            for (int i = 0; i < 3; ++i)
                if (n[i] > 0 && (double)n[i] / g[i] > load) { load = (double)n[i] / g[i]; best = i; }
            ++g[best];
            --left;
        }
        s.g_tc = g[0]; s.g_a = g[1];
        s.per = 0;
        for (int i = 0; i < 3; ++i)
            if (n[i] > 0 && (n[i] + g[i] - 1) / g[i] > s.per) s.per = (n[i] + g[i] - 1) / g[i];
    }
    const size_t lds = lds_floats(in) * sizeof(float);
    SAMX_REQUIRE(lds <= 160 * 1024, "wide role kernel needs %zu bytes of LDS", lds);
    const bool ln = (a && (roles & 3)) ? a->critic.ga1 != nullptr : (d && d->net.ga1 != nullptr);
    if (ln) {
        static LdsGrant grant_ln;
        if (int rc = grant_ln.ensure((const void*)roles_kernel<true>, lds)) return rc;
        roles_kernel<true><<<(unsigned)(8 * s.per + n_wd), NT, lds, st>>>(s);
    } else {
        static LdsGrant grant;
        if (int rc = grant.ensure((const void*)roles_kernel<false>, lds)) return rc;
        roles_kernel<false><<<(unsigned)(8 * s.per + n_wd), NT, lds, st>>>(s);
    }
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_pack(const float* const* W, float* const* dst, const int* K, const int* J, int n, hipStream_t st) {
    SAMX_REQUIRE(n >= 0 && n <= kPackMax, "wide::launch_pack: 0..%d matrices", kPackMax);
    if (n == 0) return SAMX_OK;
    PackArgs a;
    a.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        SAMX_REQUIRE(W[i] && dst[i] && K[i] > 0, "wide::launch_pack: matrix %d", i);
        a.W[i] = W[i]; a.dst[i] = dst[i]; a.K[i] = K[i]; a.J[i] = J[i];
        a.first[i] = blocks;
        blocks += packed_rows(K[i]) / 16;   // one 256-thread block per 16 rows
    }
    a.first[n] = blocks;
    pack_kernel<<<blocks, 256, 0, st>>>(a);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int launch_targets(const AgentArgs* a, hipStream_t st) {
    SAMX_REQUIRE(a && a->targ_q && (!a->n_idx || (a->n_rews && a->n_dones && a->n_len && a->targ_qn)),
                 "wide::launch_targets: missing buffers");
    const int tiles = (int)(a->B / kRows);
    const size_t lds = lds_floats(a->ob + a->ac) * sizeof(float);
    SAMX_REQUIRE(lds <= 160 * 1024, "wide target kernel needs %zu bytes of LDS", lds);
    if (a->targ_critic.ga1) {
        static LdsGrant grant_ln;
        if (int rc = grant_ln.ensure((const void*)targets_kernel<true>, lds)) return rc;
        targets_kernel<true><<<(unsigned)(tiles * (a->n_idx ? 2 : 1)), NT, lds, st>>>(*a, tiles);
    } else {
        static LdsGrant grant;
        if (int rc = grant.ensure((const void*)targets_kernel<false>, lds)) return rc;
        targets_kernel<false><<<(unsigned)(tiles * (a->n_idx ? 2 : 1)), NT, lds, st>>>(*a, tiles);
    }
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // namespace wide
}  // namespace samx

#ifdef SAMX_TRACE
extern "C" int samx_debug_wide_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(samx::wide::g_wtrace), sizeof(unsigned long long) * 256);
}
#endif
