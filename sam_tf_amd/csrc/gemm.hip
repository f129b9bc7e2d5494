// fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: exact f32 fma chain, so results are those of a k-ordered
// fmaf loop) with fused bias / activation / mask epilogues.  64x64 output tile per 256-thread
// workgroup (2x2 waves, one 32x32 accumulator each).
//
// These GEMMs are short (R = 256..400 for the layers, a few hundred rows per split for the weight
// gradients) and often launch fewer workgroups than the chip has CUs, so what bounds them is the
// global -> LDS -> MFMA latency chain of each reduction stage, not bandwidth.  Staging therefore runs
// on the LDS-DMA path (global_load_lds_dword: no VGPR round trip, no compiler-scheduled waits) through
// a ring of four 32-deep stages: up to three stages are in flight while one is on the matrix cores,
// retired by COUNTED s_waitcnt vmcnt(N) and raw s_barrier (a __syncthreads() would drain the DMAs).
//
// Operands are addressed by (free, reduction) strides so the same kernel serves
//   forward      Y = X W        (NN: A reduction-contiguous, B free-contiguous),
//   backward     dX = dZ W^T    (NT: both reduction-contiguous),
//   weight grad  dW = X^T dZ    (TN: both free-contiguous; reduction over the batch, optional split
//                                + a virtual ones row for db).
// LDS image of one operand stage (8 KB, lane-linear per DMA instruction as the hardware requires):
//   free-contiguous      : S[r][f], 32 rows of 64 floats, one DMA per row;
//   reduction-contiguous : S[f][32], 64 rows of 32 floats, two rows per DMA, with the 32 reduction
//                          positions of row f XOR-swizzled by (f & 31) ON THE SOURCE ADDRESS and again
//                          on the read, so the MFMA operand read (32 lanes = 32 different f, same r)
//                          hits 32 different banks.
// Out-of-range free indices read a clamped (valid) address: they only feed output rows/columns that
// are never stored.  Out-of-range reduction indices (last stage) and the ones row are patched in LDS
// by the wave that issued the DMA, after its own vmcnt wait and before the barrier.
#include "gemm.h"
#include <stdlib.h>

namespace samx {
namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int TILE = BK * 64;            // floats per operand stage
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;

struct EpiCtx {
    const float* bias;
    const float* aux;
    int64_t ld_aux;
    const float* add;
    int64_t ld_add;
    float alpha;
    int M;  // real rows (the ones row, if any, is row M)
};

template <int EPI>
__device__ __forceinline__ float apply_epi(float v, int i, int j, const EpiCtx& e) {
    if (EPI == EPI_BIAS) return v + e.bias[j];
    if (EPI == EPI_BIAS_LRELU) {
        float x = v + e.bias[j];
        return fmaxf(x, 0.2f * x);
    }
    if (EPI == EPI_BIAS_TANH) {
        float a = e.alpha * tanhf(v + e.bias[j]);
        return fminf(fmaxf(a, -e.alpha), e.alpha);  // clip_acs (sam_agent.py:97,177)
    }
    if (EPI == EPI_MASK) return v * (e.aux[i * e.ld_aux + j] >= 0.f ? 1.f : 0.2f);
    if (EPI == EPI_ADD_TANHGRAD) {
        float a = e.aux[i * e.ld_aux + j];
        float t = a / e.alpha;
        float s = e.add ? v + e.add[i * e.ld_add + j] : v;
        return s * (e.alpha * (1.f - t * t));
    }
    if (EPI == EPI_AXPY) return (i < e.M) ? v + e.alpha * e.aux[i * e.ld_aux + j] : v;
    return v;
}


struct Operand {
    const float* p;
    int64_t sf, sr;  // strides of the free / reduction index (one of them is 1)
    int F;           // free extent (rows of A incl. nothing virtual, columns of B)
    int f0;          // first free index of this workgroup's tile
};

// DMA number d (0..31) of an operand stage; r0 = first reduction index of the stage
template <bool RC>
__device__ __forceinline__ void dma_one(const Operand& o, int d, int r0, int r_end, float* S, int lane) {
    if (RC) {
        const int fp = d * 2;  // row pair
        const int f = fp + (lane >> 5);
        int gr = r0 + ((lane & 31) ^ (f & 31));
        gr = gr < r_end ? gr : r_end - 1;
        int gf = o.f0 + f;
        gf = gf < o.F ? gf : o.F - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gf * o.sf + gr), (lds_float*)(S + fp * 32),
                                         4, 0, 0);
    } else {
        int gf = o.f0 + lane;
        gf = gf < o.F ? gf : o.F - 1;
        int gr = r0 + d;
        gr = gr < r_end ? gr : r_end - 1;
        __builtin_amdgcn_global_load_lds((glb_float*)(o.p + (int64_t)gr * o.sr + gf), (lds_float*)(S + d * 64),
                                         4, 0, 0);
    }
}

// zero the reduction tail (r0 + r >= r_end) of what DMA d filled
template <bool RC>
__device__ __forceinline__ void zero_tail(int rv, float* S, int d, int lane) {
    if (RC) {
        if (((lane & 31) ^ ((2 * d + (lane >> 5)) & 31)) >= rv) S[d * 64 + lane] = 0.f;
    } else {
        if (d >= rv) S[d * 64 + lane] = 0.f;
    }
}

template <bool RC>
__device__ __forceinline__ int frag_base(int lane, int w32) {
    // operand element (f = w32 + (lane & 31), r = 2 kk + (lane >> 5))
    if (RC) return (w32 + (lane & 31)) * 32;
    return (lane >> 5) * 64 + w32 + (lane & 31);
}
template <bool RC>
__device__ __forceinline__ float frag(const float* S, int base, int kk, int lane) {
    if (RC) return S[base + ((2 * kk + (lane >> 5)) ^ (lane & 31))];  // (w32 + l) & 31 == l & 31
    return S[base + kk * 128];
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    if (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    if (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
}

// KS = 1: 4 waves, wave w owns the 32x32 quadrant w of the tile and runs the whole reduction.
// KS = 4: 16 waves, wave w owns quadrant w & 3 and the quarter w >> 2 of every stage's reduction; the
//         four partial accumulators of a quadrant are added through LDS in slice order at the end.
//         Used when a launch has too few workgroups to give every SIMD a wave: the serial chain of
//         dependent MFMAs per wave is what such a launch waits for, and this cuts it by four.
template <int EPI, bool A_RC, bool B_RC, int KS, int NS>
__global__ __launch_bounds__(256 * KS) void gemm_kernel(GemmArgs g) {
    constexpr int NW = 4 * KS;        // waves
    constexpr int DPW = 32 / NW;      // DMAs per wave per operand stage
    constexpr int MPW = (BK / 2) / KS;  // MFMAs per wave per stage
    extern __shared__ float lds[];  // ONE array: [NS stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int quad = wave & 3, slice = wave >> 2;
    const int wm = quad >> 1, wn = quad & 1;
    const int i0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    const bool ones = g.a_ones_rows >= 0;
    const int Mo = g.M + (ones ? 1 : 0);
    int r_begin = 0, r_end = g.R;
    if (g.nsplit > 1) {
        int chunk = ((g.R + g.nsplit - 1) / g.nsplit + BK - 1) / BK * BK;
        r_begin = blockIdx.z * chunk;
        r_end = min(g.R, r_begin + chunk);
    }
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;

    const int T = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;  // block-uniform
    const Operand oa{g.A, g.a_si, g.a_sr, g.M, i0}, ob{g.B, g.b_sj, g.b_sr, g.N, j0};
    const int ones_lane = g.M - i0;  // lane of the virtual ones row in this tile (A free-contiguous)
    const bool patch_ones = ones && ones_lane >= 0 && ones_lane < BM;
    const int abase = frag_base<A_RC>(lane, wm * 32), bbase = frag_base<B_RC>(lane, wn * 32);
    const int d0 = wave * DPW;  // this wave's DMAs of a stage: d0 .. d0 + DPW - 1, for A and for B

    for (int s = 0; s < NS - 1 && s < T; ++s) {
#pragma unroll
        for (int q = 0; q < DPW; ++q) {
            dma_one<A_RC>(oa, d0 + q, r_begin + s * BK, r_end, lds + s * 2 * TILE, lane);
            dma_one<B_RC>(ob, d0 + q, r_begin + s * BK, r_end, lds + s * 2 * TILE + TILE, lane);
        }
    }
    for (int t = 0; t < T; ++t) {
        float* Sa = lds + (t % NS) * 2 * TILE;
        float* Sb = Sa + TILE;
        // stages issued after t and still allowed in flight: t+1 .. min(t+NS-2, T-1)
        const int ahead = min(NS - 2, T - 1 - t);
        // 2 * DPW DMAs per stage may stay in flight for each of the `ahead` later stages
#define SAMX_VM(k) wait_vm<((k) <= NS - 2 ? 2 * (k) * DPW : 0)>()
        switch (ahead) {
            case 0: SAMX_VM(0); break;
            case 1: SAMX_VM(1); break;
            case 2: SAMX_VM(2); break;
            case 3: SAMX_VM(3); break;
            case 4: SAMX_VM(4); break;
            case 5: SAMX_VM(5); break;
            default: SAMX_VM(6); break;
        }
#undef SAMX_VM
        const int r0 = r_begin + t * BK, rv = r_end - r0;
        if (rv < BK) {  // last, partial stage: this wave zeroes the tail of what it loaded
#pragma unroll
            for (int q = 0; q < DPW; ++q) {
                zero_tail<A_RC>(rv, Sa, d0 + q, lane);
                zero_tail<B_RC>(rv, Sb, d0 + q, lane);
            }
        }
        if (patch_ones && lane == ones_lane) {  // A(M, r) = (r < a_ones_rows); A is free-contiguous
#pragma unroll
            for (int q = 0; q < DPW; ++q) {
                const int gr = r0 + d0 + q;
                Sa[(d0 + q) * 64 + lane] = (gr < r_end && gr < g.a_ones_rows) ? 1.f : 0.f;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // every wave is past the MFMAs of stage t-1, so that slot is refilled now, the DMAs issued
        // between the MFMAs of this stage
        const bool refill = t + NS - 1 < T;
        float* Nx = lds + ((t + NS - 1) % NS) * 2 * TILE;
        const int rn = r0 + (NS - 1) * BK;
        float fa[MPW], fb[MPW];  // this wave's operand fragments of the stage, read ahead of the MFMAs
#pragma unroll
        for (int k = 0; k < MPW; ++k) {
            fa[k] = frag<A_RC>(Sa, abase, slice * MPW + k, lane);
            fb[k] = frag<B_RC>(Sb, bbase, slice * MPW + k, lane);
        }
        if (refill) {
#pragma unroll
            for (int k = 0; k < MPW; ++k) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k], fb[k], acc, 0, 0, 0);
                if (k < DPW) dma_one<A_RC>(oa, d0 + k, rn, r_end, Nx, lane);
                else if (k < 2 * DPW) dma_one<B_RC>(ob, d0 + k - DPW, rn, r_end, Nx + TILE, lane);
            }
        } else {
#pragma unroll
            for (int k = 0; k < MPW; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k], fb[k], acc, 0, 0, 0);
        }
    }

    if (KS > 1) {  // add the slices' accumulators (slice order: deterministic)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (slice > 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[(((slice - 1) * 4 + quad) * 16 + q) * 64 + lane] = acc[q];
        }
        __syncthreads();
        if (slice > 0) return;
#pragma unroll
        for (int sl = 0; sl < KS - 1; ++sl)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += lds[((sl * 4 + quad) * 16 + q) * 64 + lane];
    }

    EpiCtx e{g.bias, g.aux, g.ld_aux, g.add, g.ld_add, g.alpha, g.M};
    const int j = j0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (i < Mo && j < g.N) {
            if (g.nsplit > 1)
                g.partials[((int64_t)blockIdx.z * Mo + i) * g.N + j] = acc[q];
            else
                g.C[(int64_t)i * g.ldc + j] = apply_epi<EPI>(acc[q], i, j, e);
        }
    }
}

template <int EPI>
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
    g.C[(int64_t)i * g.ldc + j] = apply_epi<EPI>(s, i, j, e);
}

template <int EPI, bool A_RC, bool B_RC, int KS, int NS>
int launch_inst(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    constexpr size_t lds = (size_t)NS * 2 * TILE * sizeof(float);
    static bool attr_done = false;  // per instantiation; > 64 KB of LDS needs the opt-in
    if (!attr_done && lds > 64 * 1024) {
        SAMX_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<EPI, A_RC, B_RC, KS, NS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    gemm_kernel<EPI, A_RC, B_RC, KS, NS><<<grid, 256 * KS, lds, stream>>>(g);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

template <int EPI, int KS, int NS>
int launch_ks(const GemmArgs& g, dim3 grid, hipStream_t stream) {
    const bool a_rc = g.a_sr == 1 && !(g.a_ones_rows >= 0), b_rc = g.b_sr == 1 && g.b_sj != 1;
    if (a_rc && !b_rc) return launch_inst<EPI, true, false, KS, NS>(g, grid, stream);
    if (a_rc && b_rc) return launch_inst<EPI, true, true, KS, NS>(g, grid, stream);
    if (!a_rc && !b_rc) return launch_inst<EPI, false, false, KS, NS>(g, grid, stream);
    return fail(SAMX_ERR_UNSUPPORTED, "gemm: operand layout (A free-contiguous, B reduction-contiguous)");
}

template <int EPI>
int launch(const GemmArgs& g, hipStream_t stream) {
    const int Mo = g.M + (g.a_ones_rows >= 0 ? 1 : 0);
    dim3 grid(cdiv(g.N, BN), cdiv(Mo, BM), g.nsplit);
    // fewer than two 4-wave workgroups per CU: spread each tile's reduction over 16 waves instead
    const bool few = (int64_t)grid.x * grid.y * grid.z < 512;
    static int ns_small = -1;  // SAMX_GEMM_NS: ring depth of the 16-wave variant (experiment knob)
    if (ns_small < 0) { const char* e = getenv("SAMX_GEMM_NS"); ns_small = e ? atoi(e) : 4; }
    int rc;
    if (few) rc = ns_small == 8 ? launch_ks<EPI, 4, 8>(g, grid, stream)
                 : ns_small == 6 ? launch_ks<EPI, 4, 6>(g, grid, stream) : launch_ks<EPI, 4, 4>(g, grid, stream);
    else rc = launch_ks<EPI, 1, 4>(g, grid, stream);
    if (rc != SAMX_OK) return rc;
    if (g.nsplit > 1) {
        int64_t tot = (int64_t)Mo * g.N;
        reduce_partials_kernel<EPI><<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(g, Mo);
        SAMX_LAUNCH_CHECK();
    }
    return SAMX_OK;
}

}  // namespace

int pick_nsplit(int Mo, int N, int R) {
    int tiles = cdiv(Mo, BM) * cdiv(N, BN);
    int want = 512 / tiles;  // two workgroups per CU
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
    switch (g.epi) {
        case EPI_NONE: return launch<EPI_NONE>(g, stream);
        case EPI_BIAS: return launch<EPI_BIAS>(g, stream);
        case EPI_BIAS_LRELU: return launch<EPI_BIAS_LRELU>(g, stream);
        case EPI_BIAS_TANH: return launch<EPI_BIAS_TANH>(g, stream);
        case EPI_MASK: return launch<EPI_MASK>(g, stream);
        case EPI_ADD_TANHGRAD: return launch<EPI_ADD_TANHGRAD>(g, stream);
        case EPI_AXPY: return launch<EPI_AXPY>(g, stream);
    }
    return fail(SAMX_ERR_ARG, "gemm: unknown epilogue %d", (int)g.epi);
}

}  // namespace samx
