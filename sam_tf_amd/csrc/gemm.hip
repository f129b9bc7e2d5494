// fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: exact f32 fma chain, so results are those of a k-ordered
// fmaf loop) with fused bias / activation / mask epilogues.  64x64 output tile per workgroup
// (2x2 quadrants of 32x32, one accumulator each).
//
// Staging runs on the LDS-DMA path (global_load_lds: no VGPR round trip) through a ring of three
// 32-deep stages (48 KB: three workgroups per CU; a ring of four measured 6 % slower at C5 and 11 % slower on
// an 8192x8192x2048 product, 86.7 vs 97.7 TFLOP/s): two stages are in flight while one is on the matrix cores, retired by
// COUNTED s_waitcnt vmcnt(N) and raw s_barrier (a __syncthreads() would drain the DMAs).  What bounds
// these GEMMs is how many DMA instructions a CU can retire: a dword DMA moves 256 B per instruction and a
// CU retires one every ~32 cycles (measured: 63 per wave took 3.4 us), i.e. 8 B/clk against the 16 B per
// 256 flop a 64x64x32 stage needs.  So every operand that can is staged with the 16-byte form
// (global_load_lds_dwordx4, 1 KiB per instruction, 4-byte-aligned sources are fine): 16 instead of 64
// instructions per stage.
//
// Operands are addressed by (free, reduction) strides so the same kernel serves
//   forward      Y = X W        (NN: A reduction-contiguous, B free-contiguous),
//   backward     dX = dZ W^T    (NT: both reduction-contiguous),
//   weight grad  dW = X^T dZ    (TN: both free-contiguous; reduction over the batch, optional split
//                                + a virtual ones row for db).
// LDS image of one operand stage (8 KB, lane-linear per DMA instruction as the hardware requires):
//   free-contiguous      : S[r][f], 32 rows of 64 floats.  Wide: one DMA per 4 rows (lane -> row l/16,
//                          4 floats at (l%16)*4), needs F % 4 == 0; narrow: one dword DMA per row.
//                          Fragment = ds_read_b32 of 32 consecutive f: conflict-free.
//   reduction-contiguous : S[f][32], 64 rows of 32 floats.
//                          Wide (R % 4 == 0): one DMA per 8 rows; the eight 16-byte chunks of row f are
//                          XOR-permuted by (f >> 1) & 7 ON THE SOURCE ADDRESS and again on the read; a
//                          fragment read is ONE ds_read_b128 per two MFMAs (lanes < 32 use elements 0/2,
//                          lanes >= 32 elements 1/3) and its four 16-lane groups each cover 16 different
//                          16-byte slots of the 256-byte bank row.
//                          Narrow: two rows per dword DMA, the 32 positions XOR-swizzled by f & 31,
//                          ds_read_b32 fragments.
// Out-of-range free indices read a clamped (valid) address: they only feed output rows/columns that
// are never stored.  Out-of-range reduction indices (last stage) and the ones row are patched in LDS
// by the wave that issued the DMA, after its own vmcnt wait and before the barrier.
//
// The split weight-gradient products of the 256-wide path do NOT take this kernel any more (round 3): both of their
// operands are free-contiguous, every MFMA fragment is a coalesced global load, and wgrad_direct_kernel further down
// runs them from registers alone ("Register-direct weight-gradient products").  gemm_group_kernel, the staged form
// of that launch, stays for products that do not fit it (odd row counts, offsets beyond 31 bits) and as the second
// implementation the parity tests compare against (samx_wgrad_partials_f32).
#include "gemm.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace samx {
namespace {

#ifndef SAMX_GEMM_RING
#define SAMX_GEMM_RING 3
#endif
#ifndef SAMX_GEMM_RING_FEW
#define SAMX_GEMM_RING_FEW 6
#endif
constexpr int BM = 64, BN = 64, BK = 32;
// stage slots of the ring: KS = 1 takes NSU one-stage steps (3, or 2 for the four-workgroups-per-CU variant)
template <int KS, int NSU> constexpr int ring() { return KS == 1 ? NSU : SAMX_GEMM_RING_FEW; }
constexpr int TILE = BK * 64;            // floats per operand stage
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;

// Stage timestamps of workgroup (0,0,0), wave 0 (profiling builds only: -DSAMX_TRACE)
#ifdef SAMX_TRACE
__device__ unsigned long long g_gemm_trace[64];
#define GTRACE(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (i) < 64) \
                         g_gemm_trace[(i)] = wall_clock64(); } while (0)
// register-direct weight-gradient kernel: per workgroup (wave 0) [entry, first batch landed, loop done, stored, where]
__device__ unsigned long long g_wd_trace[10 * 1024];
#define WDTRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_wd_trace[10 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define GTRACE(i) do { } while (0)
#define WDTRACE(i) do { } while (0)
#endif

struct EpiCtx {
    const float* bias;
    const float* aux;
    int64_t ld_aux;
    const float* add;
    int64_t ld_add;
    float alpha;
    int M;  // real rows (the ones row, if any, is row M)
};

__device__ __forceinline__ float apply_epi(int epi, float v, int i, int j, const EpiCtx& e) {
    switch (epi) {
        case EPI_BIAS: return v + e.bias[j];
        case EPI_BIAS_LRELU: {
            float x = v + e.bias[j];
            return fmaxf(x, 0.2f * x);
        }
        case EPI_BIAS_TANH: {
            float a = e.alpha * tanhf(v + e.bias[j]);
            return fminf(fmaxf(a, -e.alpha), e.alpha);  // clip_acs (sam_agent.py:97,177)
        }
        case EPI_MASK: return v * (e.aux[i * e.ld_aux + j] >= 0.f ? 1.f : 0.2f);
        case EPI_ADD_TANHGRAD: {
            float a = e.aux[i * e.ld_aux + j];
            float t = a / e.alpha;
            float s = e.add ? v + e.add[i * e.ld_add + j] : v;
            return s * (e.alpha * (1.f - t * t));
        }
        case EPI_AXPY: return (i < e.M) ? v + e.alpha * e.aux[i * e.ld_aux + j] : v;
        default: return v;
    }
}

struct Operand {
    const float* p;
    int64_t sf, sr;  // strides of the free / reduction index (one of them is 1)
    int F;           // free extent (rows of A incl. nothing virtual, columns of B)
    int f0;          // first free index of this workgroup's tile
};

// number of DMA instructions per operand stage
template <bool W> constexpr int ndma() { return W ? 8 : 32; }

// DMA number d of an operand stage; r0 = first reduction index of the stage
template <bool RC, bool W>
__device__ __forceinline__ void dma_one(const Operand& o, int d, int r0, int r_end, float* S, int lane) {
    if (RC && W) {
        const int f = d * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((f >> 1) & 7);   // source chunk that lands at LDS chunk lane & 7
        int gr = r0 + 4 * c;
        gr = gr < r_end ? gr : ((r_end - 1) & ~3);   // a chunk may straddle r_end (GemmArgs::overread_ok)
        int gf = o.f0 + f;
        gf = gf < o.F ? gf : o.F - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gf * o.sf + gr), (lds_float*)(S + d * 256),
                                         16, 0, 0);
    } else if (RC) {
        const int fp = d * 2;  // row pair
        const int f = fp + (lane >> 5);
        int gr = r0 + ((lane & 31) ^ (f & 31));
        gr = gr < r_end ? gr : r_end - 1;
        int gf = o.f0 + f;
        gf = gf < o.F ? gf : o.F - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gf * o.sf + gr), (lds_float*)(S + fp * 32),
                                         4, 0, 0);
    } else if (W) {
        int gf = o.f0 + 4 * (lane & 15);
        gf = gf < o.F ? gf : ((o.F - 1) & ~3);       // a chunk may straddle F (GemmArgs::overread_ok)
        int gr = r0 + 4 * d + (lane >> 4);
        gr = gr < r_end ? gr : r_end - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gr * o.sr + gf), (lds_float*)(S + d * 256),
                                         16, 0, 0);
    } else {
        int gf = o.f0 + lane;
        gf = gf < o.F ? gf : o.F - 1;
        int gr = r0 + d;
        gr = gr < r_end ? gr : r_end - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gr * o.sr + gf), (lds_float*)(S + d * 64),
                                         4, 0, 0);
    }
}

// zero the reduction tail (r >= rv within the stage) of what DMA d filled (the issuing lane's own words)
template <bool RC, bool W>
__device__ __forceinline__ void zero_tail(int rv, float* S, int d, int lane) {
    if (RC && W) {
        const int f = d * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((f >> 1) & 7);
        if (4 * c + 3 >= rv) {
            f32x4* p = (f32x4*)(S + d * 256 + lane * 4);
            f32x4 x = *p;   // own words, landed (this wave's vmcnt wait precedes)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * c + e >= rv) x[e] = 0.f;
            *p = x;
        }
    } else if (RC) {
        if (((lane & 31) ^ ((2 * d + (lane >> 5)) & 31)) >= rv) S[d * 64 + lane] = 0.f;
    } else if (W) {
        if (4 * d + (lane >> 4) >= rv) *(f32x4*)(S + d * 256 + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        if (d >= rv) S[d * 64 + lane] = 0.f;
    }
}

// this wave's MPW operand fragments of a stage: element (f = w32 + (lane & 31), r = 2 kk + (lane >> 5))
// for kk = k0 .. k0 + MPW - 1
template <bool RC, bool W, int MPW>
__device__ __forceinline__ void load_frags(const float* S, int w32, int k0, int lane, float (&fr)[MPW]) {
    if (RC && W) {
        const int f = w32 + (lane & 31);
        const float* row = S + f * 32;
        const int g = (f >> 1) & 7;
        const bool hi = lane >= 32;
#pragma unroll
        for (int i = 0; i < MPW / 2; ++i) {   // chunk c holds r = 4c .. 4c+3 = k-steps 2c, 2c+1
            const f32x4 x = *(const f32x4*)(row + (((k0 / 2 + i) ^ g) << 2));
            fr[2 * i] = hi ? x[1] : x[0];
            fr[2 * i + 1] = hi ? x[3] : x[2];
        }
    } else if (RC) {
        const float* row = S + (w32 + (lane & 31)) * 32;
#pragma unroll
        for (int k = 0; k < MPW; ++k) fr[k] = row[(2 * (k0 + k) + (lane >> 5)) ^ (lane & 31)];
    } else {
        const float* col = S + (lane >> 5) * 64 + w32 + (lane & 31);
#pragma unroll
        for (int k = 0; k < MPW; ++k) fr[k] = col[(k0 + k) * 128];
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KS = 1: 4 waves, wave w owns the 32x32 quadrant w of the tile and runs the whole reduction.
// KS = 4: 16 waves, wave w owns quadrant w & 3 and the quarter w >> 2 of every stage's reduction; the
//         four partial accumulators of a quadrant are added through LDS in slice order at the end.
//         Used when a launch has too few workgroups to give every SIMD a wave: the serial chain of
//         dependent MFMAs per wave is what such a launch waits for, and this cuts it by four.
// A_W / B_W: the operand is staged with 16-byte DMAs (see the header).
// NSU = 3: ring of three steps (one on the matrix cores, two in flight; 48 KB, three workgroups per CU).
// NSU = 2: ring of two (32 KB, four workgroups per CU, 128 VGPRs): the refill of a slot is issued right after the
//          barrier that follows its last fragment read, so a stage is still one step in flight before it is needed.
template <bool A_RC, bool B_RC, int KS, bool A_W, bool B_W, int NSU = 3>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz) {
    // A "step" covers SPB consecutive 32-deep stages behind ONE barrier: the 16-wave variant has only
    // four MFMAs per wave and stage, so it takes two stages per barrier (a barrier + wait + issue round
    // costs ~0.4 us there against 0.44 us of matrix-core time per stage).
#ifndef SAMX_GEMM_SPB
#define SAMX_GEMM_SPB 2
#endif
    constexpr int SPB = KS == 1 ? 1 : SAMX_GEMM_SPB;   // stages per step
    static_assert(NSU == 3 || (NSU == 2 && KS == 1), "ring of three, or of two for the 4-wave kernel");
    constexpr int NS = NSU * SPB;                   // stage slots
    static_assert(NS == ring<KS, NSU>(), "launch_inst sizes the LDS from ring<KS, NSU>()");
    constexpr int NW = 4 * KS;                      // waves
    constexpr int DA = ndma<A_W>(), DT = DA + ndma<B_W>();  // DMAs per stage: A's, then B's
    static_assert(DT % NW == 0, "every wave issues the same number of DMAs per stage");
    constexpr int CNT = DT / NW;                    // DMAs per wave per stage
    constexpr int MPW = (BK / 2) / KS;              // MFMAs per wave per stage
    constexpr int FPS = SPB * MPW;                  // fragments per operand per step
    extern __shared__ __attribute__((aligned(16))) float lds[];  // ONE array: [NS stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = wave & 3, slice = wave >> 2;
    const int wm = quad >> 1, wn = quad & 1;
    const int i0 = by * BM, j0 = bx * BN;
    const bool ones = g.a_ones_rows >= 0;
    const int Mo = g.M + (ones ? 1 : 0);
    int r_begin = 0, r_end = g.R;
    if (g.nsplit > 1) {
        int chunk = ((g.R + g.nsplit - 1) / g.nsplit + BK - 1) / BK * BK;
        r_begin = bz * chunk;
        r_end = min(g.R, r_begin + chunk);
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

    const int T = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;  // stages (block-uniform)
    const int TU = (T + SPB - 1) / SPB;                                   // steps
    const Operand oa{g.A, g.a_si, g.a_sr, g.M, i0}, ob{g.B, g.b_sj, g.b_sr, g.N, j0};
    const int ones_col = g.M - i0;  // column of the virtual ones row in this tile (A free-contiguous)
    const bool patch_ones = ones && ones_col >= 0 && ones_col < BM;

    // this wave's DMAs of stage t: numbers wave + NW q (q < CNT) of the list [A_0 .. A_DA-1, B_0 ..].
    // A stage past the end of the reduction (the second half of the last step when T is odd) is
    // issued with clamped addresses and zeroed whole by zero_tail (rv <= 0).
    auto issue_step = [&](int u) {
#pragma unroll
        for (int h = 0; h < SPB; ++h) {
            const int t = u * SPB + h;
            float* S = lds + (t % NS) * 2 * TILE;
            const int r0 = r_begin + t * BK;
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                const int j = wave + NW * q;   // wave-uniform
                if (j < DA) dma_one<A_RC, A_W>(oa, j, r0, r_end, S, lane);
                else dma_one<B_RC, B_W>(ob, j - DA, r0, r_end, S + TILE, lane);
            }
        }
    };

    // retire step u (counted wait on this wave's own DMAs, tail / ones-row patches, barrier)
    auto retire = [&](int u, const bool head = false) {
        // steps issued after u and still allowed in flight: u+1 .. min(u+NSU-2, TU-1); the ring of two has step 1
        // in flight only while step 0 is retired (head)
        const int ahead = NSU == 2 ? ((head && TU > 1) ? 1 : 0) : min(NSU - 2, TU - 1 - u);
        if (ahead <= 0) wait_vm<0>();
        else wait_vm<SPB * CNT>();
#pragma unroll
        for (int h = 0; h < SPB; ++h) {
            const int t = u * SPB + h;
            float* Sa = lds + (t % NS) * 2 * TILE;
            float* Sb = Sa + TILE;
            const int r0 = r_begin + t * BK, rv = r_end - r0;
            if (rv < BK) {  // partial (or absent) stage: this wave zeroes the tail of what it loaded
#pragma unroll
                for (int q = 0; q < CNT; ++q) {
                    const int j = wave + NW * q;
                    if (j < DA) zero_tail<A_RC, A_W>(rv, Sa, j, lane);
                    else zero_tail<B_RC, B_W>(rv, Sb, j - DA, lane);
                }
            }
            if (patch_ones) {  // A(M, r) = (r < a_ones_rows); A is free-contiguous: S[r][64], column ones_col
#pragma unroll
                for (int q = 0; q < CNT; ++q) {
                    const int j = wave + NW * q;
                    if (j < DA) {
                        const int row = A_W ? 4 * j + lane : j;   // rows this DMA filled: 4 (wide) or 1
                        if (lane < (A_W ? 4 : 1)) {
                            const int gr = r0 + row;
                            Sa[row * 64 + ones_col] = (gr < r_end && gr < g.a_ones_rows) ? 1.f : 0.f;
                        }
                    }
                }
            }
        }
        // lgkmcnt(0): this wave's own fragment reads of the PREVIOUS step are complete too, so after
        // the barrier that step's slots may be refilled by anyone
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    auto frags = [&](int u, float (&fa)[FPS], float (&fb)[FPS]) {
#pragma unroll
        for (int h = 0; h < SPB; ++h) {
            const float* Sa = lds + ((u * SPB + h) % NS) * 2 * TILE;
            float ta[MPW], tb[MPW];
            load_frags<A_RC, A_W, MPW>(Sa, wm * 32, slice * MPW, lane, ta);
            load_frags<B_RC, B_W, MPW>(Sa + TILE, wn * 32, slice * MPW, lane, tb);
#pragma unroll
            for (int k = 0; k < MPW; ++k) { fa[h * MPW + k] = ta[k]; fb[h * MPW + k] = tb[k]; }
        }
    };
    // One step: the MFMAs of step u run from registers while the fragment reads of step u+1 (into the
    // other register set) and the refill DMAs are in flight.
    auto step = [&](int u, float (&ca)[FPS], float (&cb)[FPS], float (&na)[FPS], float (&nb)[FPS]) {
        if (NSU == 2) {
            if (u + 1 < TU) {
                retire(u + 1);   // behind its barrier every wave has read the fragments of step u: that slot is free
                if (u + 2 < TU) issue_step(u + 2);
                frags(u + 1, na, nb);
            }
        } else {
            // slots of step u-1: its fragments were consumed before the barrier of step u was passed
            if (u + NSU - 1 < TU) issue_step(u + NSU - 1);
            if (u + 1 < TU) {
                retire(u + 1);
                frags(u + 1, na, nb);
            }
        }
#pragma unroll
        for (int k = 0; k < FPS; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[k], cb[k], acc, 0, 0, 0);
    };

    GTRACE(0);
#ifdef SAMX_TRACE
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_gemm_trace[61] = clock64();
#endif
    for (int u = 0; u < 2 && u < TU; ++u) issue_step(u);   // both rings start with two steps in flight
    GTRACE(1);
    float fa0[FPS], fb0[FPS], fa1[FPS], fb1[FPS];
    if (TU > 0) {
        retire(0, true);
        GTRACE(2);
        frags(0, fa0, fb0);
    }
    for (int u = 0; u < TU; u += 2) {
        step(u, fa0, fb0, fa1, fb1);
        GTRACE(3 + u);
        if (u + 1 < TU) { step(u + 1, fa1, fb1, fa0, fb0); GTRACE(4 + u); }
    }

    GTRACE(59);
    // Epilogue.  KS = 1: a wave finishes its own 16 accumulator rows.  KS = 4: the four slices of a
    // quadrant put their accumulators in LDS and EACH finishes a quarter of the rows (sum over the
    // slices in slice order: deterministic), so the tail is spread over all 16 waves.
    constexpr int NQ = 16 / KS;      // accumulator registers this wave finishes
    const int q0 = KS > 1 ? slice * NQ : 0;
    float val[NQ];
    if (KS > 1) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every fragment read is done
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[((slice * 4 + quad) * 16 + q) * 64 + lane] = acc[q];
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            float v = lds[((0 * 4 + quad) * 16 + q0 + n) * 64 + lane];
#pragma unroll
            for (int sl = 1; sl < KS; ++sl) v += lds[((sl * 4 + quad) * 16 + q0 + n) * 64 + lane];
            val[n] = v;
        }
    } else {
#pragma unroll
        for (int n = 0; n < NQ; ++n) val[n] = acc[n];
    }
    GTRACE(60);
    // Two passes so that no load sits between two stores (stores count in vmcnt on this ISA: a
    // load -> use between them would wait for every earlier store to complete): first every operand
    // the epilogue needs goes into registers, then the values are finished and stored.
    const int epi = (int)g.epi;
    const int j = j0 + wn * 32 + (lane & 31);
    const bool jv = j < g.N;
    auto row_of = [&](int n) {
        const int q = q0 + n;
        return i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    };
    if (g.nsplit > 1) {
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int i = row_of(n);
            if (i < Mo && jv) g.partials[((int64_t)bz * Mo + i) * g.N + j] = val[n];
        }
    } else {
        float bj = 0.f, ax[NQ], ad[NQ];
        if (epi >= EPI_BIAS && epi <= EPI_BIAS_TANH && jv) bj = g.bias[j];
#pragma unroll
        for (int n = 0; n < NQ; ++n) { ax[n] = 0.f; ad[n] = 0.f; }
        if (epi >= EPI_MASK) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const int i = row_of(n);
                if (i < g.M && jv) ax[n] = g.aux[(int64_t)i * g.ld_aux + j];   // the ones row has no aux row
            }
            if (epi == EPI_ADD_TANHGRAD && g.add) {
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    const int i = row_of(n);
                    if (i < g.M && jv) ad[n] = g.add[(int64_t)i * g.ld_add + j];
                }
            }
        }
        const float alpha = g.alpha;
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int i = row_of(n);
            float v = val[n];
            switch (epi) {   // registers only
                case EPI_BIAS: v += bj; break;
                case EPI_BIAS_LRELU: { const float x = v + bj; v = fmaxf(x, 0.2f * x); break; }
                case EPI_BIAS_TANH: {
                    const float a = alpha * tanhf(v + bj);
                    v = fminf(fmaxf(a, -alpha), alpha);  // clip_acs (sam_agent.py:97,177)
                    break;
                }
                case EPI_MASK: v *= (ax[n] >= 0.f ? 1.f : 0.2f); break;
                case EPI_ADD_TANHGRAD: {
                    const float t = ax[n] / alpha;
                    v = (v + ad[n]) * (alpha * (1.f - t * t));
                    break;
                }
                case EPI_AXPY: if (i < g.M) v += alpha * ax[n]; break;
                default: break;
            }
            if (i < Mo && jv) g.C[(int64_t)i * g.ldc + j] = v;
        }
    }
    GTRACE(63);
#ifdef SAMX_TRACE
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_gemm_trace[62] = clock64();
#endif
}

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD
// has its own L2: with blockIdx taken as it comes the gx column tiles of one 64-row panel of A land on gx different
// XCDs and every one of them fetches the panel from HBM / Infinity Cache (measured at 8192x400x376: 65 MB fetched
// per launch against 13 MB of operands, profiles/pmc_C5.json of round 3's first pass).  Here the linear block
// number L is read as (xcd = L % 8, slot = L / 8) and consecutive slots of one XCD walk the tiles of ONE panel:
// panel = (by, bz) with gx tiles for unsplit products; for split products (bz = a chunk of the reduction: its
// tiles share both operand chunks) panel = bz with gx * gy tiles.  A bijection on the full groups of 8 panels;
// the remainder keeps its own number.  Affinity only — nothing depends on where a block runs.
__device__ __forceinline__ void xcd_tile(int gx, int gy, int gz, int& bx, int& by, int& bz) {
    const int inner = gz >= 8 ? gx * gy : gx;          // tiles per panel
    const int panels = gz >= 8 ? gz : gy * gz;
    const int L = bx + gx * (by + gy * bz);
    const int full = (panels >> 3) << 3;
    if (L >= inner * full) return;
    const int xcd = L & 7, slot = L >> 3;
    const int panel = (slot / inner) * 8 + xcd, t = slot % inner;
    if (gz >= 8) { bz = panel; by = t / gx; bx = t % gx; }
    else { bx = t; by = panel % gy; bz = panel / gy; }
}

template <bool A_RC, bool B_RC, int KS, bool A_W, bool B_W, int NSU>
__global__ __launch_bounds__(256 * KS, KS == 1 ? (NSU == 2 ? 4 : 3) : 1) void gemm_kernel(GemmArgs g) {
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.tile_order == 1) xcd_tile(gridDim.x, gridDim.y, gridDim.z, bx, by, bz);
    gemm_tile<A_RC, B_RC, KS, A_W, B_W, NSU>(g, bx, by, bz);
}

// Several weight-gradient products (TN: both operands free-contiguous, 16-byte staging, split over the
// reduction) in ONE launch: the partial sums of every (problem, split, tile) go to that problem's partials
// buffer; the caller's reduction finishes them.  blockIdx.x -> (problem, tile x, tile y, split).
struct GemmGroup {
    GemmArgs g[kGemmGroupMax];
    int first[kGemmGroupMax + 1];   // first block of each problem
    int gx[kGemmGroupMax], gy[kGemmGroupMax];
    int n;
    SlotSum sums[kSlotSumMax];
    int sum_first[kSlotSumMax + 1];   // first 16-lane group of each slot sum
    int nsums;
};
__global__ __launch_bounds__(256) void gemm_group_kernel(GemmGroup G) {
    int p = 0;
    const int b = blockIdx.x;
    if (b >= G.first[G.n]) {   // slot sums
        const int g = ((b - G.first[G.n]) * 256 + (int)threadIdx.x) >> 4, l = threadIdx.x & 15;
        int q = 0;
        while (q < G.nsums && g >= G.sum_first[q + 1]) ++q;
        if (q >= G.nsums) return;   // whole 16-lane groups leave together
        const SlotSum& s = G.sums[q];
        const int c = g - G.sum_first[q];
        float sm = 0.f;
        for (int z = l; z < s.n; z += 16) sm += s.slots[(int64_t)z * s.width + c];
        for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
        if (l == 0) s.out[c] = sm;
        return;
    }
    while (p + 1 < G.n && b >= G.first[p + 1]) ++p;
    const int local = b - G.first[p];
    const int bx = local % G.gx[p], by = (local / G.gx[p]) % G.gy[p], bz = local / (G.gx[p] * G.gy[p]);
    gemm_tile<false, false, 1, true, true>(G.g[p], bx, by, bz);
}

// ---- Register-direct weight-gradient products ---------------------------------------------------------------
// dW = X^T dZ has BOTH operands free-contiguous, so the MFMA fragment of a k-step (two batch rows) is 32 consecutive
// floats of a row: a coalesced global load, nothing has to pass through LDS.  Lane (m, hi) loads NE consecutive
// columns of X row r + hi (ONE 16-byte load for NE = 4) and two of dZ; register e of the first is the A operand of
// the 32-row block {i0 + NE ii + e}, register f of the second the B operand of the columns {j0 + 2 jj + f}: 2 NE
// MFMAs per pair of loads, a 32 NE x 64 output tile per WAVE in 2 NE accumulators (NE = 4: 128 x 64, 21 flop per
// byte from L2).  No barrier and no LDS traffic in the loop; the four waves of a workgroup take a quarter of the
// reduction chunk each and add their accumulators through LDS at the end, (q0 + q2) + (q1 + q3) — a fixed order.
// The bias gradient (the virtual ones row of the staged kernel, a whole extra row of tiles when M % 64 == 0) is
// the column sum of dZ: two adds per k-step on registers the first M tile holds anyway.
// Workgroups that share operands — the tiles of one (product, chunk) — are placed on ONE XCD (blockIdx % 8), the
// groups dealt to the XCDs longest first by the host.
struct WProb {
    const float *A, *B;
    float* P;          // [nsplit][Mo][N]
    int lda, ldb, M, N, R, ones, chunk, Mo;
    int tmn, tnn;      // M tiles (128 wide; the last one narrower: NE of 1, 2 or 4), N tiles (64 wide)
};
constexpr int kWGroupMax = 384;
#ifndef SAMX_WGRAD_TILE_NE
#define SAMX_WGRAD_TILE_NE 2
#endif
constexpr int kWTileNE = SAMX_WGRAD_TILE_NE;   // full M tiles are 32 kWTileNE wide
static_assert(kWTileNE == 1 || kWTileNE == 2 || kWTileNE == 4, "M tiles of 32, 64 or 128");
constexpr int kWOcc = kWTileNE == 1 ? 4 : kWTileNE == 2 ? 2 : 1;   // workgroups per CU the register budget is set for
struct WDirect {
    WProb g[kGemmGroupMax];
    SlotSum sums[kSlotSumMax];
    int sum_first[kSlotSumMax + 1];
    int nsums;
    int slots;                    // workgroup slots per XCD: blocks [0, 8 slots) are product tiles
    int xfirst[9];                // groups of XCD x: grp[xfirst[x] .. xfirst[x + 1])
    uint32_t grp[kWGroupMax];     // product | chunk << 8 | first slot << 16
};
int wgrad_direct(const GemmArgs* list, int n, hipStream_t stream, const SlotSum* sums, int nsums);
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NE>
__device__ __forceinline__ void load_rows(const __amdgpu_buffer_rsrc_t r, const int lane_off, const int row_off,
                                          float (&o)[NE]) {
    if constexpr (NE == 4) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, row_off, 0));
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    } else if constexpr (NE == 2) {
        const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, row_off, 0));
        o[0] = v[0]; o[1] = v[1];
    } else {
        o[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, row_off, 0));
    }
}

template <int NE>
__device__ __forceinline__ void wdirect_tile(const WProb& g, const int i0, const int j0, const int z, float* red) {
    constexpr int KD = kWTileNE == 4 ? 16 : 8;   // k-steps (row pairs) in flight
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the row addresses below stay on the SALU)
    const int r_begin = z * g.chunk, r_end = min(g.R, r_begin + g.chunk);
    const int q = g.chunk >> 2;    // rows per wave (the chunk is a multiple of 32, R is even)
    const int w_begin = r_begin + wave * q, w_end = min(r_end, w_begin + q);
    const bool colsum = g.ones >= 0 && i0 == 0;
    const int ones_lim = colsum ? g.ones : 0;   // rows below it count for the bias gradient
    // lanes past the edge read a clamped (valid) address: they feed rows / columns that are never stored
    const int ia = min(i0 + NE * m, (g.M - 1) & ~(NE - 1)), jb = min(j0 + 2 * m, (g.N - 1) & ~1);
    // Buffer loads: resource (scalar) + a fixed 32-bit lane offset + the row's byte offset as the SCALAR offset
    // operand.  With one wave per SIMD every VALU instruction between two MFMAs is time the matrix pipe idles
    // (tools/exp/mfma_mix.hip: one 64-bit multiply-add per load costs 10 of 64 clocks per MFMA), so the loop keeps
    // its address arithmetic on the SALU.  (The host checks that the byte offsets fit 31 bits.)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0x7fffffff, 0x00020000);
    const int sa = g.lda * 4, sb = g.ldb * 4;
    const int oa = (hi * g.lda + ia) * 4, ob = (hi * g.ldb + jb) * 4;
    auto row_a = [&](const int r, float (&o)[NE]) { load_rows<NE>(ra, oa, r * sa, o); };
    auto row_b = [&](const int r, float (&o)[2]) { load_rows<2>(rb, ob, r * sb, o); };
    f32x16 acc[NE][2];
#pragma unroll
    for (int e = 0; e < NE; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[e][f][v] = 0.f;
    float cs[2] = {0.f, 0.f};

    // Main loop over groups of KD k-steps (2 KD rows), branch-free: the registers of a k-step are refilled with the
    // rows of the NEXT group right after its MFMAs have been issued (KD - 1 k-steps = 2 x KD - 2 loads stay in
    // flight: the counted waits are the compiler's); the refills of the last group re-read it (discarded).  Then
    // the rows that do not fill a group, one k-step at a time.
    const int rows = w_end > w_begin ? w_end - w_begin : 0;
    const int nd = rows / (2 * KD);
    WDTRACE(0);
#ifdef SAMX_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wd_trace[10 * blockIdx.x + 4] = ((unsigned long long)xcc << 32) | hw;
        g_wd_trace[10 * blockIdx.x + 8] = clock64();
    }
#endif
    if (nd > 0) {
        float a[KD][NE], b[KD][2];
#pragma unroll
        for (int s = 0; s < KD; ++s) {
            // (in the loop's own order: the counted waits come out of the merge of this state with the back edge's —
            // with the loads reordered here the loop head gets a vmcnt(0))
            __builtin_amdgcn_sched_barrier(0);
            row_a(w_begin + 2 * s, a[s]);
            row_b(w_begin + 2 * s, b[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int last = w_begin + (nd - 1) * 2 * KD;
        // one group; MODE (compile time) is how the bias row counts its rows: 0 not at all (no bias row, or not
        // this tile's job), 1 all of them, 2 those below ones_lim
        auto group = [&](const int r0, const int rn, auto mode) {
            constexpr int MODE = decltype(mode)::value;
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink every load to just before its use)
#pragma unroll
                for (int e = 0; e < NE; ++e)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][e], b[s][f], acc[e][f], 0, 0, 0);
                if (MODE == 1) {
                    cs[0] += b[s][0];
                    cs[1] += b[s][1];
                } else if (MODE == 2) {
                    const bool in = r0 + 2 * s + hi < ones_lim;
                    cs[0] += in ? b[s][0] : 0.f;
                    cs[1] += in ? b[s][1] : 0.f;
                }
                row_a(rn + 2 * s, a[s]);
                row_b(rn + 2 * s, b[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int n = 0; n < nd; ++n) {
            const int r0 = w_begin + n * 2 * KD, rn = min(r0 + 2 * KD, last);
            if (r0 >= ones_lim) group(r0, rn, std::integral_constant<int, 0>());
            else if (r0 + 2 * KD <= ones_lim) group(r0, rn, std::integral_constant<int, 1>());
            else group(r0, rn, std::integral_constant<int, 2>());
            if (n == 0) WDTRACE(1);
        }
    }
    for (int r = w_begin + nd * 2 * KD; r < w_end; r += 2) {
        float a[NE], b[2];
        row_a(r, a);
        row_b(r, b);
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[f], acc[e][f], 0, 0, 0);
        const bool in = r + hi < ones_lim;
        cs[0] += in ? b[0] : 0.f;
        cs[1] += in ? b[1] : 0.f;
    }

    WDTRACE(2);
#ifdef SAMX_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wd_trace[10 * blockIdx.x + 9] = clock64();
#endif
    // The four partial tiles are added through LDS, ((q0 + q1) + q2) + q3: wave u finishes accumulator registers
    // 4u .. 4u + 3 of every accumulator (output rows NE (k + 8u + 4 hi) + e) and takes the other waves' from LDS
    // [wave][accumulator][register][lane].
    constexpr int NA = 2 * NE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u == wave) continue;   // (wave-uniform)
#pragma unroll
        for (int c = 0; c < NA; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                red[((wave * NA + c) * 16 + 4 * u + k) * 64 + lane] = acc[c >> 1][c & 1][4 * u + k];
    }
    float* red_cs = red + 4 * NA * 16 * 64;
    if (colsum && wave > 0) { red_cs[(wave * 2) * 64 + lane] = cs[0]; red_cs[(wave * 2 + 1) * 64 + lane] = cs[1]; }
    __syncthreads();
    WDTRACE(5);
    float* P = g.P + (int64_t)z * g.Mo * g.N;
    const int j = j0 + 2 * m;
    const bool fast = i0 + 32 * NE <= g.M && (g.N & 1) == 0;   // every row of the tile exists, pairs are aligned
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u != wave) continue;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            float x[2][4];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = e * 2 + f, v = 4 * u + k;
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float term = t == u ? acc[e][f][v] : red[((t * NA + c) * 16 + v) * 64 + lane];
                        sum = t == 0 ? term : sum + term;
                    }
                    x[f][k] = sum;
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + NE * (k + 8 * u + 4 * hi) + e;
                float* dst = P + (int64_t)i * g.N + j;
                if (fast) {
                    if (j < g.N) *(f32x2u*)dst = f32x2u{x[0][k], x[1][k]};
                } else if (i < g.M) {
                    if (j < g.N) dst[0] = x[0][k];
                    if (j + 1 < g.N) dst[1] = x[1][k];
                }
            }
        }
    }
    WDTRACE(3);
    if (colsum && wave == 0) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float x = cs[f];
#pragma unroll
            for (int t = 1; t < 4; ++t) x += red_cs[(t * 2 + f) * 64 + lane];
            x += __shfl_down(x, 32);
            if (hi == 0 && j + f < g.N) P[(int64_t)g.M * g.N + j + f] = x;
        }
    }
}

__global__ __launch_bounds__(256, kWOcc) void wgrad_direct_kernel(WDirect G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x;
    if (b >= 8 * G.slots) {   // slot sums (as in gemm_group_kernel)
        const int g = ((b - 8 * G.slots) * 256 + (int)threadIdx.x) >> 4, l = threadIdx.x & 15;
        int q = 0;
        while (q < G.nsums && g >= G.sum_first[q + 1]) ++q;
        if (q >= G.nsums) return;
        const SlotSum& s = G.sums[q];
        const int c = g - G.sum_first[q];
        float sm = 0.f;
        for (int z = l; z < s.n; z += 16) sm += s.slots[(int64_t)z * s.width + c];
        for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 16);
        if (l == 0) s.out[c] = sm;
        return;
    }
    const int xcd = b & 7, slot = b >> 3;
    for (int k = G.xfirst[xcd]; k < G.xfirst[xcd + 1]; ++k) {
        const uint32_t e = G.grp[k];
        const int p = e & 255, z = (e >> 8) & 255, first = (int)(e >> 16);
        const WProb& g = G.g[p];
        const int t = slot - first;
        if (t < 0 || t >= g.tmn * g.tnn) continue;
        const int tm = t / g.tnn, tn = t - tm * g.tnn;
        const int i0 = tm * 32 * kWTileNE, rem = g.M - i0;
        if constexpr (kWTileNE == 4) {
            if (rem > 64) { wdirect_tile<4>(g, i0, tn * 64, z, lds); return; }
        }
        if constexpr (kWTileNE >= 2) {
            if (rem > 32) { wdirect_tile<2>(g, i0, tn * 64, z, lds); return; }
        }
        wdirect_tile<1>(g, i0, tn * 64, z, lds);
        return;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(GemmArgs g, int Mo) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Mo * g.N) return;
    const int i = (int)(idx / g.N), j = (int)(idx % g.N);
    const int64_t zs = (int64_t)Mo * g.N;
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= g.nsplit; z += 4) {  // four loads in flight, added in split order
        const float t0 = g.partials[z * zs + idx], t1 = g.partials[(z + 1) * zs + idx],
                    t2 = g.partials[(z + 2) * zs + idx], t3 = g.partials[(z + 3) * zs + idx];
        s += t0; s += t1; s += t2; s += t3;
    }
    for (; z < g.nsplit; ++z) s += g.partials[z * zs + idx];
    EpiCtx e{g.bias, g.aux, g.ld_aux, g.add, g.ld_add, g.alpha, g.M};
    g.C[(int64_t)i * g.ldc + j] = apply_epi((int)g.epi, s, i, j, e);
}

template <bool A_RC, bool B_RC, int KS, bool A_W, bool B_W, int NSU = 3>
int launch_inst(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    constexpr size_t lds = (size_t)ring<KS, NSU>() * 2 * TILE * sizeof(float);
    static LdsGrant grant;  // per instantiation and device; > 64 KB of LDS needs the opt-in
    if (int rc = grant.ensure((const void*)gemm_kernel<A_RC, B_RC, KS, A_W, B_W, NSU>, lds, 64 * 1024)) return rc;
    gemm_kernel<A_RC, B_RC, KS, A_W, B_W, NSU><<<grid, 256 * KS, lds, stream>>>(g);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

template <bool A_RC, bool B_RC>
int launch_layout(const GemmArgs& g, dim3 grid, bool few, hipStream_t stream) {
    // 16-byte staging needs whole 4-element chunks along the operand's contiguous index
    static int narrow = -1;  // SAMX_GEMM_NARROW=1: dword staging everywhere (A/B experiments)
    if (narrow < 0) { const char* e = getenv("SAMX_GEMM_NARROW"); narrow = e ? atoi(e) : 0; }
    const int ea = A_RC ? g.R : g.M, eb = B_RC ? g.R : g.N;   // extent along the contiguous index
    const bool aw = !narrow && ea >= 4 && (ea % 4 == 0 || g.overread_ok);
    const bool bw = !narrow && eb >= 4 && (eb % 4 == 0 || g.overread_ok);
    if (few) {  // 16 waves: both wide or both narrow (every wave issues the same number of DMAs)
        if (aw && bw) return launch_inst<A_RC, B_RC, 4, true, true>(g, grid, stream);
        return launch_inst<A_RC, B_RC, 4, false, false>(g, grid, stream);
    }
    if (aw && bw) {
        // Unsplit launches of more than three tiles per CU take the ring of two (four workgroups per CU): 769..1024
        // tiles then run in ONE round (8192x400x376: 50.1 -> 53.2 TFLOP/s), longer launches gain from the fourth
        // resident workgroup (8192^2 x 2048: 100.4 -> 103.7; C5, whose D chain has 1920..2688 tiles per product,
        // 684 -> 701 steps/s).  Split weight-gradient products lose 3-5 % with it and keep the ring of three.
        static int four = -1;  // SAMX_GEMM_FOUR: 0 never, 1 as above (default), 2 always (A/B experiments)
        if (four < 0) { const char* e = getenv("SAMX_GEMM_FOUR"); four = e ? atoi(e) : 1; }
        const int64_t wgs = (int64_t)grid.x * grid.y * grid.z;
        if (four == 2 || (four == 1 && wgs > 768 && g.nsplit == 1))
            return launch_inst<A_RC, B_RC, 1, true, true, 2>(g, grid, stream);
        return launch_inst<A_RC, B_RC, 1, true, true>(g, grid, stream);
    }
    if (aw) return launch_inst<A_RC, B_RC, 1, true, false>(g, grid, stream);
    if (bw) return launch_inst<A_RC, B_RC, 1, false, true>(g, grid, stream);
    return launch_inst<A_RC, B_RC, 1, false, false>(g, grid, stream);
}

int launch(const GemmArgs& g_in, hipStream_t stream) {
    GemmArgs g = g_in;
    static int xcd = -1;   // SAMX_GEMM_XCD=0: blockIdx as it comes (A/B experiments)
    if (xcd < 0) { const char* e = getenv("SAMX_GEMM_XCD"); xcd = e ? atoi(e) : 1; }
    g.tile_order = xcd ? 1 : 0;
    const int Mo = g.M + (g.a_ones_rows >= 0 ? 1 : 0);
    dim3 grid(cdiv(g.N, BN), cdiv(Mo, BM), g.nsplit);
    // fewer than two 4-wave workgroups per CU: spread each tile's reduction over 16 waves instead
    const bool few = (int64_t)grid.x * grid.y * grid.z < 512;
    const bool a_rc = g.a_sr == 1 && !(g.a_ones_rows >= 0), b_rc = g.b_sr == 1 && g.b_sj != 1;
    int rc = SAMX_ERR_UNSUPPORTED;
    if (!a_rc && !b_rc && g.nsplit > 1) {
        // SAMX_WGRAD_DIRECT_SPLIT=1: split weight-gradient products of the layer-wise path on the register-direct
        // kernel too.  Off: at C5, where three role chains run side by side, it measured 705 against 709 steps/s
        // (686..707 over 384..1536 split slots) — three staged workgroups per CU mix better with the other chains'
        // launches than two 67 KB ones.
        static int direct = -1;
        if (direct < 0) { const char* e = getenv("SAMX_WGRAD_DIRECT_SPLIT"); direct = e ? atoi(e) : 0; }
        if (direct) rc = wgrad_direct(&g, 1, stream, nullptr, 0);
    }
    if (rc != SAMX_ERR_UNSUPPORTED) { if (rc != SAMX_OK) return rc; }
    else if (a_rc && !b_rc) rc = launch_layout<true, false>(g, grid, few, stream);
    else if (a_rc && b_rc) rc = launch_layout<true, true>(g, grid, few, stream);
    else if (!a_rc && !b_rc) rc = launch_layout<false, false>(g, grid, few, stream);
    else return fail(SAMX_ERR_UNSUPPORTED, "gemm: operand layout (A free-contiguous, B reduction-contiguous)");
    if (rc != SAMX_OK) return rc;
    if (g.nsplit > 1) {
        int64_t tot = (int64_t)Mo * g.N;
        reduce_partials_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(g, Mo);
        SAMX_LAUNCH_CHECK();
    }
    return SAMX_OK;
}

}  // namespace

namespace {
// SAMX_ERR_UNSUPPORTED (silently: the caller falls back to the staged kernel) when a product does not fit
int wgrad_direct(const GemmArgs* list, int n, hipStream_t stream, const SlotSum* sums, int nsums) {
    WDirect G;
    struct Grp { int tiles, p, z; };
    Grp grp[kWGroupMax];
    int ngrp = 0;
    for (int p = 0; p < n; ++p) {
        const GemmArgs& g = list[p];
        if (!(g.a_si == 1 && g.b_sj == 1 && (g.overread_ok || (g.M % kWTileNE == 0 && g.N % 2 == 0)) &&
              (g.R & 1) == 0 && g.M >= 4 && g.N >= 2 &&
              g.nsplit <= 255 && ngrp + g.nsplit <= kWGroupMax &&
              ((int64_t)g.R + 1) * g.a_sr * 4 < (1ll << 31) && ((int64_t)g.R + 1) * g.b_sr * 4 < (1ll << 31)))
            return SAMX_ERR_UNSUPPORTED;
        WProb& w = G.g[p];
        w.A = g.A; w.B = g.B; w.P = g.partials;
        w.lda = (int)g.a_sr; w.ldb = (int)g.b_sr; w.M = g.M; w.N = g.N; w.R = g.R; w.ones = g.a_ones_rows;
        w.chunk = gemm_split_chunk(g.R, g.nsplit);
        w.Mo = g.M + (g.a_ones_rows >= 0 ? 1 : 0);
        w.tmn = cdiv(g.M, 32 * kWTileNE); w.tnn = cdiv(g.N, 64);
        for (int z = 0; z < g.nsplit; ++z) grp[ngrp++] = Grp{w.tmn * w.tnn, p, z};
    }
    // longest group first onto the XCD with the fewest workgroups so far
    std::stable_sort(grp, grp + ngrp, [](const Grp& a, const Grp& b) { return a.tiles > b.tiles; });
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xcd_of[kWGroupMax], first[kWGroupMax];
    for (int k = 0; k < ngrp; ++k) {
        int x = 0;
        for (int y = 1; y < 8; ++y) if (load[y] < load[x]) x = y;
        xcd_of[k] = x; first[k] = load[x]; load[x] += grp[k].tiles;
    }
    G.slots = 0;
    for (int x = 0; x < 8; ++x) G.slots = load[x] > G.slots ? load[x] : G.slots;
    if (G.slots > 65535) return SAMX_ERR_UNSUPPORTED;
    int e = 0;
    for (int x = 0; x < 8; ++x) {
        G.xfirst[x] = e;
        for (int k = 0; k < ngrp; ++k)
            if (xcd_of[k] == x) G.grp[e++] = (uint32_t)grp[k].p | ((uint32_t)grp[k].z << 8) | ((uint32_t)first[k] << 16);
    }
    G.xfirst[8] = e;
    G.nsums = nsums;
    int groups = 0;
    for (int q = 0; q < nsums; ++q) {
        G.sums[q] = sums[q];
        G.sum_first[q] = groups;
        groups += sums[q].width;
    }
    G.sum_first[nsums] = groups;
    const int blocks = 8 * G.slots + (groups * 16 + 255) / 256;
    // [wave][accumulator][register][lane] of the widest tile + the column sums
    constexpr size_t lds = (size_t)(4 * 2 * kWTileNE * 16 + 8) * 64 * sizeof(float);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)wgrad_direct_kernel, lds, 64 * 1024)) return rc;
    wgrad_direct_kernel<<<blocks, 256, lds, stream>>>(G);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}
}  // namespace

int gemm_group_partials(const GemmArgs* list, int n, hipStream_t stream, const SlotSum* sums, int nsums, int kernel) {
    SAMX_REQUIRE(list && n > 0 && n <= kGemmGroupMax, "gemm_group_partials: 1..%d problems", kGemmGroupMax);
    SAMX_REQUIRE(nsums >= 0 && nsums <= kSlotSumMax && (nsums == 0 || sums), "gemm_group_partials: 0..%d slot sums",
                 kSlotSumMax);
    for (int p = 0; p < n; ++p) {
        const GemmArgs& g = list[p];
        SAMX_REQUIRE(g.A && g.B && g.partials && g.M > 0 && g.N > 0 && g.R > 0 && g.nsplit >= 2,
                     "gemm_group_partials: problem %d needs operands, a partials buffer and nsplit >= 2", p);
        SAMX_REQUIRE(g.a_si == 1 && g.b_sj == 1, "gemm_group_partials: problem %d is not a free-contiguous product", p);
    }
    for (int q = 0; q < nsums; ++q)
        SAMX_REQUIRE(sums[q].slots && sums[q].out && sums[q].n > 0 && sums[q].width > 0,
                     "gemm_group_partials: slot sum %d", q);
    {   // the register-direct kernel takes the call when every product fits its form (SAMX_WGRAD_DIRECT=0: never)
        static int direct = -1;
        if (direct < 0) { const char* e = getenv("SAMX_WGRAD_DIRECT"); direct = e ? atoi(e) : 1; }
        if (kernel == 1 || (kernel < 0 && direct)) {
            const int rc = wgrad_direct(list, n, stream, sums, nsums);
            if (rc != SAMX_ERR_UNSUPPORTED) return rc;
            if (kernel == 1)
                return fail(SAMX_ERR_UNSUPPORTED, "gemm_group_partials: a product does not fit the register-direct kernel");
        }
    }
    GemmGroup G;
    G.n = n;
    int blocks = 0;
    for (int p = 0; p < n; ++p) {
        const GemmArgs& g = list[p];
        const int ea = g.M, eb = g.N;
        SAMX_REQUIRE(ea >= 4 && eb >= 4 && (ea % 4 == 0 || g.overread_ok) && (eb % 4 == 0 || g.overread_ok),
                     "gemm_group_partials: problem %d does not fit the 16-byte staging of the staged kernel", p);
        const int Mo = g.M + (g.a_ones_rows >= 0 ? 1 : 0);
        G.g[p] = g;
        G.gx[p] = cdiv(g.N, BN);
        G.gy[p] = cdiv(Mo, BM);
        G.first[p] = blocks;
        blocks += G.gx[p] * G.gy[p] * g.nsplit;
    }
    G.first[n] = blocks;
    G.nsums = nsums;
    int groups = 0;
    for (int q = 0; q < nsums; ++q) {
        G.sums[q] = sums[q];
        G.sum_first[q] = groups;
        groups += sums[q].width;
    }
    G.sum_first[nsums] = groups;
    blocks += (groups * 16 + 255) / 256;
    constexpr size_t lds = (size_t)ring<1, 3>() * 2 * TILE * sizeof(float);
    static LdsGrant grant;
    if (int rc = grant.ensure((const void*)gemm_group_kernel, lds, 64 * 1024)) return rc;
    gemm_group_kernel<<<blocks, 256, lds, stream>>>(G);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int pick_nsplit(int Mo, int N, int R) {
    int tiles = cdiv(Mo, BM) * cdiv(N, BN);
    // Workgroups to aim for: three 4-wave workgroups per CU now fit (launch bounds, 168 VGPRs).  Measured inside the
    // C5 step (three role streams running side by side, one box, alternating): 512 -> 623 steps/s, 768 -> 636,
    // 1024 -> 622, 256 (one round of the 16-wave kernel; best for a product running ALONE, 46.6 -> 41.3 us at
    // 376x400x8192) -> 619.
    static int slots = -1;   // SAMX_GEMM_SPLIT_SLOTS: A/B experiments
    if (slots < 0) { const char* e = getenv("SAMX_GEMM_SPLIT_SLOTS"); slots = e ? atoi(e) : 768; if (slots < 1) slots = 768; }
    int want = slots / tiles;
    int most = R / 128;      // keep at least 4 stages per split
    int s = want < most ? want : most;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

int gemm(const GemmArgs& g, hipStream_t stream) {
    SAMX_REQUIRE(g.A && g.B && g.C, "gemm: null operand");
    SAMX_REQUIRE(g.M > 0 && g.N > 0 && g.R > 0, "gemm: empty problem %dx%dx%d", g.M, g.N, g.R);
    SAMX_REQUIRE((g.a_si == 1 || g.a_sr == 1) && (g.b_sr == 1 || g.b_sj == 1),
                 "gemm: each operand needs one unit stride");
    SAMX_REQUIRE(g.nsplit == 1 || g.partials, "gemm: split needs a partials buffer");
    SAMX_REQUIRE(g.nsplit >= 1 && g.nsplit <= 65535, "gemm: bad nsplit");
    SAMX_REQUIRE(g.a_ones_rows < 0 || g.a_si == 1, "gemm: the ones row needs a free-contiguous A");
    SAMX_REQUIRE((int)g.epi >= EPI_NONE && (int)g.epi <= EPI_AXPY, "gemm: unknown epilogue %d", (int)g.epi);
    return launch(g, stream);
}

}  // namespace samx

#ifdef SAMX_TRACE
extern "C" int samx_debug_wd_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(samx::g_wd_trace), sizeof(unsigned long long) * 10 * 1024);
}
extern "C" int samx_debug_gemm_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(samx::g_gemm_trace), sizeof(unsigned long long) * 64);
}
#endif

extern "C" int samx_wgrad_partials_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* partials,
                                       int32_t M, int32_t N, int32_t R, int32_t a_ones_rows, int32_t nsplit,
                                       int32_t kernel, int32_t overread_ok, void* stream) {
    SAMX_REQUIRE(A && B && partials && M > 0 && N > 0 && R > 0 && nsplit >= 2, "samx_wgrad_partials_f32: arguments");
    SAMX_REQUIRE(kernel == 0 || kernel == 1, "samx_wgrad_partials_f32: kernel 0 (staged) or 1 (register-direct)");
    samx::GemmArgs g;
    g.A = A; g.a_si = 1; g.a_sr = lda;
    g.B = B; g.b_sr = ldb; g.b_sj = 1;
    g.C = partials; g.ldc = N;
    g.M = M; g.N = N; g.R = R;
    g.a_ones_rows = a_ones_rows;
    g.nsplit = nsplit; g.partials = partials;
    g.overread_ok = overread_ok != 0;
    return samx::gemm_group_partials(&g, 1, (hipStream_t)stream, nullptr, 0, kernel);
}

extern "C" int samx_gemm_f32(const float* A, int64_t a_si, int64_t a_sr, const float* B, int64_t b_sr, int64_t b_sj,
                             float* C, int64_t ldc, int32_t M, int32_t N, int32_t R, int32_t a_ones_rows,
                             int32_t epi, const float* bias, const float* aux, int64_t ld_aux, const float* add,
                             int64_t ld_add, float alpha, int32_t nsplit, float* partials, int32_t overread_ok,
                             void* stream) {
    samx::GemmArgs g;
    g.A = A; g.a_si = a_si; g.a_sr = a_sr;
    g.B = B; g.b_sr = b_sr; g.b_sj = b_sj;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.R = R;
    g.a_ones_rows = a_ones_rows;
    g.epi = (samx::Epilogue)epi;
    g.bias = bias; g.aux = aux; g.ld_aux = ld_aux; g.add = add; g.ld_add = ld_add; g.alpha = alpha;
    g.nsplit = nsplit; g.partials = partials;
    g.overread_ok = overread_ok != 0;
    SAMX_REQUIRE(epi < samx::EPI_BIAS || epi > samx::EPI_BIAS_TANH || bias, "samx_gemm_f32: epilogue %d needs bias", epi);
    SAMX_REQUIRE(epi < samx::EPI_MASK || aux, "samx_gemm_f32: epilogue %d needs aux", epi);
    return samx::gemm(g, (hipStream_t)stream);
}
