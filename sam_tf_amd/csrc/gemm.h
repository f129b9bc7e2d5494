// Internal fp32 MFMA GEMM used by the layer-wise network path (nets.hip).
#pragma once
#include "samx_common.h"

namespace samx {

enum Epilogue {
    EPI_NONE = 0,
    EPI_BIAS,          // acc + bias[j]
    EPI_BIAS_LRELU,    // lrelu(acc + bias[j])            (network_util.py:59-72, leak 0.2)
    EPI_BIAS_TANH,     // high * tanh(acc + bias[j])       (networks.py:287-302)
    EPI_MASK,          // acc * lrelu'(aux[i,j])           aux = saved post-activation
    EPI_ADD_TANHGRAD,  // (acc + add[i,j]) * high*(1 - (aux[i,j]/high)^2)   aux = saved action
    EPI_AXPY,          // acc + alpha * aux[i,j]           (weight decay term, network_util.py:43-56)
};

// C[i,j] = epilogue( sum_r A(i,r) * B(r,j) ),  i < M (+1 when a_ones_rows >= 0), j < N, r < R.
// A(i,r) = A[i*a_si + r*a_sr], B(r,j) = B[r*b_sr + j*b_sj]; one stride of each pair must be 1.
// a_ones_rows >= 0 appends a virtual row i == M to A with A(M,r) = (r < a_ones_rows) — the bias
// gradient falls out of the weight-gradient GEMM as its last output row.
struct GemmArgs {
    const float* A = nullptr;
    int64_t a_si = 0, a_sr = 0;
    const float* B = nullptr;
    int64_t b_sr = 0, b_sj = 0;
    float* C = nullptr;
    int64_t ldc = 0;
    int M = 0, N = 0, R = 0;
    int a_ones_rows = -1;
    Epilogue epi = EPI_NONE;
    const float* bias = nullptr;
    const float* aux = nullptr;
    int64_t ld_aux = 0;
    const float* add = nullptr;
    int64_t ld_add = 0;
    float alpha = 0.f;  // EPI_AXPY scale, EPI_*TANH* action bound
    // split over r: when nsplit > 1 partial sums go to `partials` ([nsplit][Mo][N]) and a second
    // kernel applies the epilogue; partials must hold nsplit*Mo*N floats (Mo = M or M+1)
    int nsplit = 1;
    float* partials = nullptr;
    // 16-byte staging of an operand whose contiguous extent is not a multiple of 4 reads up to 12 bytes
    // past the end of a row (into the next row, or past the matrix after its last row).  Callers that
    // guarantee 16 readable bytes behind both operands set this (nets.hip: workspace carves are padded,
    // weight matrices are followed by their bias in the flat vector); otherwise such operands are
    // staged with dword DMAs.
    bool overread_ok = false;
    // tile order of the launch: 0 = blockIdx as it comes, 1 = XCD-aware (set by gemm(): see xcd_tile in gemm.hip)
    int tile_order = 0;
};

int gemm(const GemmArgs& g, hipStream_t stream);

// Up to kGemmGroupMax split weight-gradient products (A and B free-contiguous, nsplit >= 2) in ONE launch;
// only the partial sums are written ([nsplit][Mo][N] per problem, split z covers the reduction indices
// [z * chunk, (z + 1) * chunk), chunk = gemm_split_chunk(R, nsplit)): the caller reduces them.
// The same launch can also add up small per-tile slot arrays (extra workgroups at the end of the grid):
// out[c] = sum over z < n of slots[z * width + c], 16 lanes per element (lane l takes z = l, l + 16, ...,
// then a fixed xor tree).
constexpr int kGemmGroupMax = 10, kSlotSumMax = 4;
struct SlotSum { const float* slots; int n, width; float* out; };
// kernel: -1 the register-direct kernel when every product fits it (SAMX_WGRAD_DIRECT=0: never), 0 the staged
// (LDS ring) kernel, 1 the register-direct kernel or SAMX_ERR_UNSUPPORTED.
int gemm_group_partials(const GemmArgs* list, int n, hipStream_t stream, const SlotSum* sums = nullptr,
                        int nsums = 0, int kernel = -1);
inline int gemm_split_chunk(int R, int nsplit) { return ((R + nsplit - 1) / nsplit + 31) / 32 * 32; }

// number of r-splits that fills the chip for a weight-gradient GEMM
int pick_nsplit(int Mo, int N, int R);

}  // namespace samx
