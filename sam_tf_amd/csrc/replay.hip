// Replay row gather / write and running mean-std sums.
//   imitation_algorithms/memory.py:33-57,74-84 ; helpers/mpi_running_mean_std.py:31-59
// All HBM-bound byte moving; the gather reads B random rows of (2*ob+ac+2) floats.
#include "samx_common.h"
#include "../../include/samx.h"

namespace {

// one wave per sampled row; lanes stride over the row's floats
__global__ __launch_bounds__(256) void gather_kernel(
    const float* __restrict__ obs0, const float* __restrict__ acs, const float* __restrict__ rews,
    const float* __restrict__ dones1, const float* __restrict__ obs1, int64_t maxlen, int64_t start,
    int ob, int ac, const int64_t* __restrict__ idxs, int64_t B, float* __restrict__ o_obs0,
    float* __restrict__ o_acs, float* __restrict__ o_rews, float* __restrict__ o_dones1,
    float* __restrict__ o_obs1) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B) return;
    int64_t idx = idxs[row];
    int64_t src = (start + idx) % maxlen;
    if (src < 0) src += maxlen;
    if (o_obs0)
        for (int c = lane; c < ob; c += 64) o_obs0[row * ob + c] = obs0[src * ob + c];
    if (o_obs1)
        for (int c = lane; c < ob; c += 64) o_obs1[row * ob + c] = obs1[src * ob + c];
    if (o_acs)
        for (int c = lane; c < ac; c += 64) o_acs[row * ac + c] = acs[src * ac + c];
    if (lane == 0) {
        if (o_rews) o_rews[row] = rews[src];
        if (o_dones1) o_dones1[row] = dones1[src];
    }
}

__global__ __launch_bounds__(256) void write_kernel(float* __restrict__ obs0, float* __restrict__ acs,
                                                    float* __restrict__ rews,
                                                    float* __restrict__ dones1,
                                                    float* __restrict__ obs1, int ob, int ac,
                                                    const float* __restrict__ packed,
                                                    const int64_t* __restrict__ rows, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t i = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const int w = 2 * ob + ac + 2;
    const float* p = packed + i * w;
    const int64_t r = rows[i];
    for (int c = lane; c < ob; c += 64) obs0[r * ob + c] = p[c];
    for (int c = lane; c < ac; c += 64) acs[r * ac + c] = p[ob + c];
    for (int c = lane; c < ob; c += 64) obs1[r * ob + c] = p[ob + ac + 1 + c];
    if (lane == 0) {
        rews[r] = p[ob + ac];
        dones1[r] = p[2 * ob + ac + 1];
    }
}

// Column sums of x and x^2 in float64.  Block = 16 row-lanes x 16 columns; every row-lane adds its
// rows in increasing order, the 16 partials are then added in lane order: a fixed association, so
// results are reproducible run to run (they differ from numpy's strictly sequential order only in
// the last bits of a double).
constexpr int kRmsRows = 16, kRmsCols = 16;
__global__ __launch_bounds__(kRmsRows* kRmsCols) void rms_sums_kernel(
    const float* __restrict__ x1, int64_t n1, const float* __restrict__ x2, int64_t n2, int dim,
    double* __restrict__ addvec) {
    __shared__ double s1[kRmsRows][kRmsCols], s2[kRmsRows][kRmsCols];
    const int cx = threadIdx.x % kRmsCols, ry = threadIdx.x / kRmsCols;
    const int col = blockIdx.x * kRmsCols + cx;
    double a = 0.0, q = 0.0;
    if (col < dim) {
        const int64_t n = n1 + n2;
        // eight loads in flight per row-lane, added in the same increasing row order (one load per dependent
        // add cost a memory round trip per row: 17 us for 1024 rows of one column)
        constexpr int U = 8;
        int64_t r = ry;
        for (; r + (U - 1) * kRmsRows < n; r += U * kRmsRows) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t ru = r + u * kRmsRows;
                v[u] = (ru < n1) ? x1[ru * dim + col] : x2[(ru - n1) * dim + col];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double d = (double)v[u];
                a = __dadd_rn(a, d);
                q = __dadd_rn(q, __dmul_rn(d, d));
            }
        }
        for (; r < n; r += kRmsRows) {
            double v = (r < n1) ? (double)x1[r * dim + col] : (double)x2[(r - n1) * dim + col];
            a = __dadd_rn(a, v);
            q = __dadd_rn(q, __dmul_rn(v, v));
        }
    }
    s1[ry][cx] = a;
    s2[ry][cx] = q;
    __syncthreads();
    if (ry == 0 && col < dim) {
        double ta = 0.0, tq = 0.0;
        for (int r = 0; r < kRmsRows; ++r) {
            ta = __dadd_rn(ta, s1[r][cx]);
            tq = __dadd_rn(tq, s2[r][cx]);
        }
        addvec[col] = ta;
        addvec[dim + col] = tq;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) addvec[2 * dim] = (double)(n1 + n2);
}

__global__ void rms_commit_kernel(double* __restrict__ sums, const double* __restrict__ addvec,
                                  int dim, float* __restrict__ mean_out,
                                  float* __restrict__ std_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const double cnt = sums[2 * dim] + (addvec ? addvec[2 * dim] : 0.0);
    if (c < dim) {
        double s = sums[c], q = sums[dim + c];
        if (addvec) {
            s = __dadd_rn(s, addvec[c]);
            q = __dadd_rn(q, addvec[dim + c]);
            sums[c] = s;
            sums[dim + c] = q;
        }
        // mpi_running_mean_std.py:31-34: the casts to float32 come BEFORE the subtraction
        const float mean = (float)(s / cnt);
        const float msq = (float)(q / cnt);
        const float var = __fsub_rn(msq, __fmul_rn(mean, mean));
        mean_out[c] = mean;
        std_out[c] = sqrtf(fmaxf(var, 1e-2f));
    }
    // the count is written by a thread that cannot race with readers of the old value: the grid is
    // one block whenever dim <= blockDim, otherwise readers may see either count — so do it last
    // from a separate tiny launch (see host wrapper)
}

__global__ void rms_count_kernel(double* sums, const double* addvec, int dim) {
    if (threadIdx.x == 0 && blockIdx.x == 0) sums[2 * dim] = sums[2 * dim] + addvec[2 * dim];
}

}  // namespace

namespace {
// One thread per sampled index (memory.py:125-148).  Rows are read in batches of 32 so that the loads
// of a batch are in flight together; the discounted sum then runs from the far end in float64 with
// separate multiply and add (the order and rounding of scipy's lfilter on the reversed rewards).
__global__ __launch_bounds__(128) void nstep_kernel(const float* __restrict__ rews,
                                                    const float* __restrict__ dones1, int64_t maxlen,
                                                    int64_t start, int64_t num_entries,
                                                    const int64_t* __restrict__ idxs, int64_t idx_stride,
                                                    int64_t rows_per_batch, int64_t B, int n,
                                                    double gamma, int64_t* __restrict__ end_idx_out,
                                                    float* __restrict__ rews_out,
                                                    double* __restrict__ rews64_out,
                                                    float* __restrict__ dones_out,
                                                    float* __restrict__ td_len_out,
                                                    const int64_t* __restrict__ geom) {
    const int64_t t = blockIdx.x * (int64_t)128 + threadIdx.x;
    if (t >= B) return;
    // batch k's indices start at idxs + k*idx_stride (the driver's per-row index words); outputs are dense
    int64_t idx = idxs[(t / rows_per_batch) * idx_stride + t % rows_per_batch];
    int64_t phys0;
    if (geom) {   // storage rows in, storage rows out; the ring's geometry from device memory (see nstep_group_kernel)
        start = geom[0]; num_entries = geom[1];
        phys0 = idx;
        idx -= start;
        if (idx < 0) idx += maxlen;
    } else {
        phys0 = start + idx;
        if (phys0 >= maxlen) phys0 -= maxlen;
    }
    const int64_t last = (idx + n < num_entries ? idx + n : num_entries) - 1;  // lookahead_end_idx
    const int win = (int)(last - idx + 1);
    // first done inside the window (NB loads in flight per batch: a window of n = 96 rows is three memory round
    // trips per pass, not twelve — the kernel is nothing but these round trips)
    constexpr int NB = 32;
    int hit = -1;
    for (int k0 = 0; k0 < win && hit < 0; k0 += NB) {
        float d[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            int64_t p = phys0 + k0 + u;
            if (p >= maxlen) p -= maxlen;
            d[u] = (k0 + u < win) ? dones1[p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (hit < 0 && d[u] == 1.0f) hit = k0 + u;
    }
    const int td_len = hit >= 0 ? hit + 1 : win;
    double acc = 0.0;
    for (int k0 = td_len - 1; k0 >= 0; k0 -= NB) {
        float r[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            int64_t p = phys0 + (k0 - u);
            if (p >= maxlen) p -= maxlen;
            r[u] = (k0 - u >= 0) ? rews[p] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (k0 - u >= 0) acc = __dadd_rn((double)r[u], __dmul_rn(gamma, acc));
    }
    int64_t end = idx + td_len - 1;
    if (geom) { end = phys0 + td_len - 1; if (end >= maxlen) end -= maxlen; }
    end_idx_out[t] = end;
    rews_out[t] = (float)acc;
    if (rews64_out) rews64_out[t] = acc;
    dones_out[t] = (hit >= 0 && hit != win - 1) ? 1.f : 0.f;
    td_len_out[t] = (float)td_len;
}

// The same lookahead with SIXTEEN lanes per sampled index (windows of n <= kNstepGroupMax rows): the lanes read the
// window's dones / rewards side by side (consecutive ring rows: whole cache lines per group instead of one line per
// lane and load — with one thread per index the kernel was 60 us of uncoalesced gathers on 32 CUs for 16 x 256
// indices at n = 96), park them in LDS, and lane 0 of the group runs the reference's sequential arithmetic on them —
// the same operations in the same order as nstep_kernel, so the same bits.
// geom != nullptr (both kernels): the indices are STORAGE rows ((start + idx) % maxlen of the reference's logical
// index, what the driver's index words carry under uniform replay), {start, num_entries} are read from device memory
// — they change with every append, and a captured graph must not hold them — and end_idx_out is a storage row too.
constexpr int kNstepGroupMax = 256;
__global__ __launch_bounds__(256) void nstep_group_kernel(const float* __restrict__ rews,
                                                          const float* __restrict__ dones1, int64_t maxlen,
                                                          int64_t start, int64_t num_entries,
                                                          const int64_t* __restrict__ idxs, int64_t idx_stride,
                                                          int64_t rows_per_batch, int64_t B, int n, double gamma,
                                                          int64_t* __restrict__ end_idx_out,
                                                          float* __restrict__ rews_out,
                                                          double* __restrict__ rews64_out,
                                                          float* __restrict__ dones_out,
                                                          float* __restrict__ td_len_out,
                                                          const int64_t* __restrict__ geom) {
    __shared__ float sd[16][kNstepGroupMax], sr[16][kNstepGroupMax];
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int64_t t = blockIdx.x * (int64_t)16 + g;
    const bool valid = t < B;
    int64_t idx = 0, phys0 = 0;
    int win = 0;
    if (valid) {
        idx = idxs[(t / rows_per_batch) * idx_stride + t % rows_per_batch];
        if (geom) {
            start = geom[0]; num_entries = geom[1];
            phys0 = idx;
            idx -= start;
            if (idx < 0) idx += maxlen;
        } else {
            phys0 = start + idx;
            if (phys0 >= maxlen) phys0 -= maxlen;
        }
        const int64_t last = (idx + n < num_entries ? idx + n : num_entries) - 1;  // lookahead_end_idx
        win = (int)(last - idx + 1);
    }
    for (int k = l; k < win; k += 16) {
        int64_t p = phys0 + k;
        if (p >= maxlen) p -= maxlen;
        sd[g][k] = dones1[p];
        sr[g][k] = rews[p];
    }
    __syncthreads();
    if (!valid || l != 0) return;
    int hit = -1;
    for (int k = 0; k < win && hit < 0; ++k)
        if (sd[g][k] == 1.0f) hit = k;
    const int td_len = hit >= 0 ? hit + 1 : win;
    double acc = 0.0;
    for (int k = td_len - 1; k >= 0; --k) acc = __dadd_rn((double)sr[g][k], __dmul_rn(gamma, acc));
    int64_t end = idx + td_len - 1;
    if (geom) { end = phys0 + td_len - 1; if (end >= maxlen) end -= maxlen; }
    end_idx_out[t] = end;
    rews_out[t] = (float)acc;
    if (rews64_out) rews64_out[t] = acc;
    dones_out[t] = (hit >= 0 && hit != win - 1) ? 1.f : 0.f;
    td_len_out[t] = (float)td_len;
}
}  // namespace

namespace {
// the host-drawn index words of a launch: pinned host memory -> HBM by a KERNEL (16 bytes per thread), so that the
// upload stays on the compute queue between two graph launches (a stream-ordered hipMemcpyAsync hands over to the
// copy engine and back: the GPU idles on both edges)
__global__ __launch_bounds__(256) void copy_words_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst,
                                                          int64_t rows, int64_t row_len, int64_t row_stride) {
    // pairs of words: row_len and row_stride are even or the row is copied word by word (host-chosen)
    const int64_t per_row = (row_len + 1) / 2, t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= rows * per_row) return;
    const int64_t r = t / per_row, i = 2 * (t - r * per_row), at = r * row_stride + i;
    if (i + 1 < row_len && (at & 1) == 0) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        *(ll2*)(dst + at) = *(const ll2*)(src + at);
    } else {
        dst[at] = src[at];
        if (i + 1 < row_len) dst[at + 1] = src[at + 1];
    }
}
}  // namespace

extern "C" {

int samx_copy_words(const int64_t* src, int64_t* dst, int64_t rows, int64_t row_len, int64_t row_stride, void* stream) {
    SAMX_REQUIRE(src && dst && rows >= 0 && row_len >= 0 && row_stride >= row_len,
                 "samx_copy_words: null pointer, negative count or row_stride < row_len");
    SAMX_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "samx_copy_words: 16-byte aligned buffers");
    const int64_t n = rows * ((row_len + 1) / 2);
    if (n == 0) return SAMX_OK;
    copy_words_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, dst, rows, row_len, row_stride);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_replay_gather(const float* obs0, const float* acs, const float* rews, const float* dones1,
                       const float* obs1, int64_t maxlen, int64_t start, int ob_dim, int ac_dim,
                       const int64_t* idxs, int64_t B, float* o_obs0, float* o_acs, float* o_rews,
                       float* o_dones1, float* o_obs1, void* stream) {
    SAMX_REQUIRE(idxs, "samx_replay_gather: null idxs");
    SAMX_REQUIRE(maxlen > 0 && ob_dim > 0 && ac_dim > 0 && B >= 0, "samx_replay_gather: bad sizes");
    SAMX_REQUIRE((!o_obs0 || obs0) && (!o_acs || acs) && (!o_rews || rews) &&
                     (!o_dones1 || dones1) && (!o_obs1 || obs1),
                 "samx_replay_gather: an output was requested without its source array");
    if (B == 0) return SAMX_OK;
    gather_kernel<<<(unsigned)((B + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        obs0, acs, rews, dones1, obs1, maxlen, start, ob_dim, ac_dim, idxs, B, o_obs0, o_acs, o_rews,
        o_dones1, o_obs1);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_replay_write(float* obs0, float* acs, float* rews, float* dones1, float* obs1, int ob_dim,
                      int ac_dim, const float* packed, const int64_t* rows, int64_t n, void* stream) {
    SAMX_REQUIRE(obs0 && acs && rews && dones1 && obs1 && packed && rows,
                 "samx_replay_write: null pointer");
    SAMX_REQUIRE(ob_dim > 0 && ac_dim > 0 && n >= 0, "samx_replay_write: bad sizes");
    if (n == 0) return SAMX_OK;
    write_kernel<<<(unsigned)((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        obs0, acs, rews, dones1, obs1, ob_dim, ac_dim, packed, rows, n);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_rms_batch_sums(const float* x1, int64_t n1, const float* x2, int64_t n2, int dim,
                        double* addvec, void* stream) {
    SAMX_REQUIRE(addvec && dim > 0, "samx_rms_batch_sums: bad arguments");
    SAMX_REQUIRE(n1 >= 0 && n2 >= 0 && (n1 == 0 || x1) && (n2 == 0 || x2),
                 "samx_rms_batch_sums: row count without data");
    rms_sums_kernel<<<samx::cdiv(dim, kRmsCols), kRmsRows * kRmsCols, 0, (hipStream_t)stream>>>(
        x1, n1, x2, n2, dim, addvec);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_rms_commit(double* sums, const double* addvec, int dim, float* mean_out, float* std_out,
                    void* stream) {
    SAMX_REQUIRE(sums && mean_out && std_out && dim > 0, "samx_rms_commit: bad arguments");
    rms_commit_kernel<<<samx::cdiv(dim, 256), 256, 0, (hipStream_t)stream>>>(sums, addvec, dim,
                                                                            mean_out, std_out);
    SAMX_LAUNCH_CHECK();
    if (addvec) {
        rms_count_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sums, addvec, dim);
        SAMX_LAUNCH_CHECK();
    }
    return SAMX_OK;
}


int samx_nstep_lookahead(const float* rews, const float* dones1, int64_t maxlen, int64_t start,
                         int64_t num_entries, const int64_t* idxs, int64_t B, int n, double gamma,
                         int64_t* end_idx_out, float* rews_out, double* rews64_out, float* dones_out,
                         float* td_len_out, void* stream) {
    SAMX_REQUIRE(rews && dones1 && idxs && end_idx_out && rews_out && dones_out && td_len_out,
                 "samx_nstep_lookahead: null pointer");
    SAMX_REQUIRE(B > 0 && n > 1 && n <= 4096, "samx_nstep_lookahead: need B > 0 and 1 < n <= 4096");
    SAMX_REQUIRE(maxlen > 0 && start >= 0 && start < maxlen && num_entries > 0 && num_entries <= maxlen,
                 "samx_nstep_lookahead: bad ring geometry");
    if (n <= kNstepGroupMax)
        nstep_group_kernel<<<(unsigned)((B + 15) / 16), 256, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, start, num_entries, idxs, B, B, B, n, gamma, end_idx_out, rews_out, rews64_out,
            dones_out, td_len_out, nullptr);
    else
        nstep_kernel<<<(unsigned)((B + 127) / 128), 128, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, start, num_entries, idxs, B, B, B, n, gamma, end_idx_out, rews_out, rews64_out,
            dones_out, td_len_out, nullptr);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_nstep_lookahead_many(const float* rews, const float* dones1, int64_t maxlen, int64_t start,
                              int64_t num_entries, const int64_t* idxs, int64_t idx_stride, int64_t K,
                              int64_t B, int n, double gamma, int64_t* end_idx_out, float* rews_out,
                              float* dones_out, float* td_len_out, void* stream) {
    SAMX_REQUIRE(rews && dones1 && idxs && end_idx_out && rews_out && dones_out && td_len_out,
                 "samx_nstep_lookahead_many: null pointer");
    SAMX_REQUIRE(B > 0 && K > 0 && idx_stride >= B && n > 1 && n <= 4096,
                 "samx_nstep_lookahead_many: need B > 0, K > 0, idx_stride >= B and 1 < n <= 4096");
    SAMX_REQUIRE(maxlen > 0 && start >= 0 && start < maxlen && num_entries > 0 && num_entries <= maxlen,
                 "samx_nstep_lookahead_many: bad ring geometry");
    if (n <= kNstepGroupMax)
        nstep_group_kernel<<<(unsigned)((K * B + 15) / 16), 256, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, start, num_entries, idxs, idx_stride, B, K * B, n, gamma, end_idx_out, rews_out,
            nullptr, dones_out, td_len_out, nullptr);
    else
        nstep_kernel<<<(unsigned)((K * B + 127) / 128), 128, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, start, num_entries, idxs, idx_stride, B, K * B, n, gamma, end_idx_out, rews_out,
            nullptr, dones_out, td_len_out, nullptr);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_nstep_lookahead_many_rows(const float* rews, const float* dones1, int64_t maxlen, const int64_t* geom,
                                   const int64_t* rows, int64_t row_stride, int64_t K, int64_t B, int n, double gamma,
                                   int64_t* end_row_out, float* rews_out, float* dones_out, float* td_len_out,
                                   void* stream) {
    SAMX_REQUIRE(rews && dones1 && geom && rows && end_row_out && rews_out && dones_out && td_len_out,
                 "samx_nstep_lookahead_many_rows: null pointer");
    SAMX_REQUIRE(B > 0 && K > 0 && row_stride >= B && n > 1 && n <= 4096 && maxlen > 0,
                 "samx_nstep_lookahead_many_rows: need B > 0, K > 0, row_stride >= B, 1 < n <= 4096, maxlen > 0");
    if (n <= kNstepGroupMax)
        nstep_group_kernel<<<(unsigned)((K * B + 15) / 16), 256, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, 0, 0, rows, row_stride, B, K * B, n, gamma, end_row_out, rews_out, nullptr, dones_out,
            td_len_out, geom);
    else
        nstep_kernel<<<(unsigned)((K * B + 127) / 128), 128, 0, (hipStream_t)stream>>>(
            rews, dones1, maxlen, 0, 0, rows, row_stride, B, K * B, n, gamma, end_row_out, rews_out, nullptr, dones_out,
            td_len_out, geom);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // extern "C"
