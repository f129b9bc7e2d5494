// fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32: exact f32 fma chain, so results are those of a k-ordered
// fmaf loop) with fused bias / activation / mask epilogues.  64x64 output tile per 256-thread
// workgroup (2x2 waves, one 32x32 accumulator each), K staged through LDS 16 deep.
// Operands are addressed by (free, reduction) strides so the same kernel serves
//   forward      Y = X W        (NN),
//   backward     dX = dZ W^T    (NT),
//   weight grad  dW = X^T dZ    (TN, reduction over the batch, optional split + ones row for db).
#include "gemm.h"

namespace samx {
namespace {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

template <int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
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

    for (int r0 = r_begin; r0 < r_end; r0 += BK) {
        // ---- stage A tile: As[r][i]
        if (g.a_sr == 1) {  // reduction index contiguous in memory
            const int i = tid >> 2, rq = (tid & 3) * 4, gi = i0 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + rq + q;
                float v = 0.f;
                if (r < r_end) {
                    if (gi < g.M) v = g.A[(int64_t)gi * g.a_si + r];
                    else if (ones && gi == g.M) v = (r < g.a_ones_rows) ? 1.f : 0.f;
                }
                As[rq + q][i] = v;
            }
        } else {  // free index contiguous
            const int rr = tid >> 4, iq = (tid & 15) * 4, r = r0 + rr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gi = i0 + iq + q;
                float v = 0.f;
                if (r < r_end) {
                    if (gi < g.M) v = g.A[(int64_t)r * g.a_sr + gi];
                    else if (ones && gi == g.M) v = (r < g.a_ones_rows) ? 1.f : 0.f;
                }
                As[rr][iq + q] = v;
            }
        }
        // ---- stage B tile: Bs[r][j]
        if (g.b_sj == 1) {
            const int rr = tid >> 4, jq = (tid & 15) * 4, r = r0 + rr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gj = j0 + jq + q;
                Bs[rr][jq + q] = (r < r_end && gj < g.N) ? g.B[(int64_t)r * g.b_sr + gj] : 0.f;
            }
        } else {
            const int j = tid >> 2, rq = (tid & 3) * 4, gj = j0 + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + rq + q;
                Bs[rq + q][j] = (r < r_end && gj < g.N) ? g.B[(int64_t)gj * g.b_sj + r] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[2 * kk + (lane >> 5)][wm * 32 + (lane & 31)];
            const float b = Bs[2 * kk + (lane >> 5)][wn * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
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
    float s = 0.f;
    for (int z = 0; z < g.nsplit; ++z) s += g.partials[(int64_t)z * Mo * g.N + idx];
    EpiCtx e{g.bias, g.aux, g.ld_aux, g.add, g.ld_add, g.alpha, g.M};
    g.C[(int64_t)i * g.ldc + j] = apply_epi<EPI>(s, i, j, e);
}

template <int EPI>
int launch(const GemmArgs& g, hipStream_t stream) {
    const int Mo = g.M + (g.a_ones_rows >= 0 ? 1 : 0);
    dim3 grid(cdiv(g.N, BN), cdiv(Mo, BM), g.nsplit);
    gemm_kernel<EPI><<<grid, 256, 0, stream>>>(g);
    SAMX_LAUNCH_CHECK();
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
    int want = 256 / tiles;
    int most = R / 128;  // keep at least 8 K-steps per split
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
