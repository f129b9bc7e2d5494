// Segment-tree and prioritized-replay kernels (float64, bit-exact with the reference's Python).
//   helpers/segment_tree.py:26-97 ; imitation_algorithms/memory.py:248-309,337-379
// HBM/latency-bound pointer chasing over a 2*cap double array (16 MiB at cap 2^20); no MFMA here.
#include "samx_common.h"
#include "tree_dev.h"
#include "../../include/samx.h"
#include <math.h>
#include <string.h>

namespace {

using samx::kTreeThreads;
using samx::kMaxLevels;
using samx::tree_op;
using samx::range_reduce_wave;
using samx::prefix_descent;

__global__ void tree_init_kernel(double* s, double* m, int32_t* stamp, int64_t cap) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < 2 * cap) {
        s[i] = 0.0;
        m[i] = INFINITY;
    }
    if (i < cap) stamp[i] = -1;
}

__global__ __launch_bounds__(kTreeThreads) void tree_set_kernel(double* __restrict__ s,
                                                                double* __restrict__ m,
                                                                int32_t* __restrict__ stamp,
                                                                int64_t cap,
                                                                const int64_t* __restrict__ idxs,
                                                                const double* __restrict__ leaf,
                                                                int n, int levels) {
    samx::tree_set_body(s, m, stamp, cap, idxs, leaf, n, levels);   // tree_dev.h
}

// n < 1024: the update whose dependent chain runs in LDS (tree_dev.h: tree_set_fast_body)
__global__ __launch_bounds__(kTreeThreads) void tree_set_fast_kernel(double* __restrict__ s, double* __restrict__ m,
                                                                     int64_t cap, const int64_t* __restrict__ idxs,
                                                                     const double* __restrict__ leaf, int n,
                                                                     int levels) {
    extern __shared__ double fast_lds[];
    samx::tree_set_fast_body(s, m, cap, idxs, leaf, n, levels, *reinterpret_cast<samx::FastLds*>(fast_lds));
}

// Large batches (C5: 8192 updates into a 2^22 tree): one workgroup is bound by the number of random
// 8-byte accesses a single CU can keep in flight (8192 x 4 loads per level, 22 levels: ~460 us measured).
// Split by SUBTREE instead: workgroup g owns the leaves [g * cap/G, (g+1) * cap/G) and every internal
// node below that subtree's root, so the workgroups never touch the same node and need no grid-wide
// synchronisation for the lower `sub_levels` levels.  Each workgroup scans the whole batch, keeps the
// positions that fall into its leaf range (LDS list), runs the same election / leaf write / bottom-up
// recompute as the one-workgroup kernel on them.  The top log2(G) levels are then recomputed IN FULL by
// tree_set_top_kernel (a second launch: G - 1 nodes, every one from its two children, so untouched nodes
// are rewritten with their own value).  Same arithmetic per node as the reference, hence the same bits.
constexpr int kSetGroups = 256;        // subtrees (power of two)
constexpr int kSetMaxBatch = 8192;     // LDS list capacity (positions)

__global__ __launch_bounds__(kTreeThreads) void tree_set_sub_kernel(double* __restrict__ s,
                                                                    double* __restrict__ m,
                                                                    int32_t* __restrict__ stamp,
                                                                    int64_t cap,
                                                                    const int64_t* __restrict__ idxs,
                                                                    const double* __restrict__ leaf,
                                                                    int n, int sub_levels) {
    __shared__ int list[kSetMaxBatch];
    __shared__ int cnt;
    const int tid = threadIdx.x;
    const int64_t sub = cap / gridDim.x, lo = (int64_t)blockIdx.x * sub, hi = lo + sub;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kTreeThreads) {
        const int64_t idx = idxs[i];
        if (idx >= lo && idx < hi) list[atomicAdd(&cnt, 1)] = i;
    }
    __syncthreads();
    const int c = cnt;
    if (c == 0) return;
    for (int k = tid; k < c; k += kTreeThreads) {
        const int i = list[k];
        atomicMax(&stamp[idxs[i]], i);
    }
    __syncthreads();
    for (int k = tid; k < c; k += kTreeThreads) {
        const int i = list[k];
        const int64_t idx = idxs[i];
        const int32_t w = __hip_atomic_load(&stamp[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == i) {
            const double v = leaf[i];
            s[cap + idx] = v;
            m[cap + idx] = v;
        }
    }
    __syncthreads();
    for (int k = tid; k < c; k += kTreeThreads) stamp[idxs[list[k]]] = -1;
    for (int l = 1; l <= sub_levels; ++l) {
        __syncthreads();
        for (int k = tid; k < c; k += kTreeThreads) {
            const int64_t node = (cap + idxs[list[k]]) >> l;
            s[node] = __dadd_rn(s[2 * node], s[2 * node + 1]);
            const double a = m[2 * node], b = m[2 * node + 1];
            m[node] = (b < a) ? b : a;
        }
    }
}

// levels sub_levels+1 .. levels (counted from the leaves), every node of each level
__global__ __launch_bounds__(kTreeThreads) void tree_set_top_kernel(double* __restrict__ s,
                                                                    double* __restrict__ m, int64_t cap,
                                                                    int sub_levels, int levels) {
    for (int l = sub_levels + 1; l <= levels; ++l) {
        const int64_t count = cap >> l;   // nodes [count, 2 * count) form this level
        for (int64_t j = threadIdx.x; j < count; j += kTreeThreads) {
            const int64_t node = count + j;
            s[node] = __dadd_rn(s[2 * node], s[2 * node + 1]);
            const double a = m[2 * node], b = m[2 * node + 1];
            m[node] = (b < a) ? b : a;
        }
        __syncthreads();
    }
}

__global__ void tree_reduce_kernel(const double* v, int64_t cap, int kind, int64_t lo, int64_t hi,
                                   double* out) {
    __shared__ int64_t nodes[2 * kMaxLevels + 2];
    __shared__ double vals[2 * kMaxLevels + 2];
    __shared__ int counts[2];
    const double r = range_reduce_wave(v, cap, kind, lo, hi, threadIdx.x, nodes, vals, counts);
    if (threadIdx.x == 0) out[0] = r;
}

__global__ void tree_find_kernel(const double* v, int64_t cap, const double* prefix, int64_t n,
                                 int64_t* out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = prefix_descent(v, cap, prefix[i]);
}

__global__ __launch_bounds__(256) void per_sample_kernel(
    const double* __restrict__ sumv, const double* __restrict__ minv, int64_t cap, int64_t n_ent,
    const double* __restrict__ u, int64_t B, double beta, int64_t* __restrict__ idxs,
    double* __restrict__ iws, float* __restrict__ iws32, samx::SampleExt x) {
    samx::per_sample_body(sumv, minv, cap, n_ent, u, B, beta, idxs, iws, iws32,
                          blockIdx.x * (int64_t)256, 256, nullptr, 0, nullptr, x);   // tree_dev.h
}

// the priority refresh on the device (memory.py:370-379 for the entries of one update)
__global__ __launch_bounds__(kTreeThreads) void per_leaves_from_td_kernel(
    const float* __restrict__ td, const int64_t* __restrict__ idxs, int n, int64_t cap, int64_t num_entries,
    double alpha, int64_t num_demos, double demos_eps, double* __restrict__ leaf_out, samx::PerState* state,
    const int64_t* __restrict__ geom) {
    if (geom) num_entries = geom[1];   // the ring's entry count as the launch's upload left it (device words)
    samx::leaves_from_td_body(td, idxs, n, cap, num_entries, alpha, num_demos, demos_eps, leaf_out, state);
}

__global__ void pow_bits_kernel(const double* __restrict__ x, double y, int64_t n, double* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = samx_pow::pow_bits(x[i], y);
}

// every float32 base from bits lo on, widened (+ bonus): count of results that differ from `expect` is
// not formed here - the device writes its results, the host compares (tests/test_gpu_pow.py)
__global__ void pow_bits_f32_range_kernel(uint32_t lo, int64_t n, double bonus, double y, double* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        const float f = __uint_as_float(lo + (uint32_t)i);
        out[i] = samx_pow::pow_bits(__dadd_rn((double)f, bonus), y);
    }
}

int levels_of(int64_t cap) {
    int l = 0;
    while ((int64_t(1) << l) < cap) ++l;
    return l;
}

}  // namespace

extern "C" {

int samx_tree_init(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap, void* stream) {
    SAMX_REQUIRE(sum_tree && min_tree && stamp, "samx_tree_init: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_tree_init: cap %lld is not a power of two",
                 (long long)cap);
    int64_t n = 2 * cap;
    tree_init_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(sum_tree, min_tree,
                                                                                  stamp, cap);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_tree_set(double* sum_tree, double* min_tree, int32_t* stamp, int64_t cap,
                  const int64_t* idxs, const double* leaf, int64_t n, void* stream) {
    SAMX_REQUIRE(sum_tree && min_tree && stamp && idxs && leaf, "samx_tree_set: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_tree_set: cap %lld is not a power of two",
                 (long long)cap);
    SAMX_REQUIRE(n >= 0 && n <= (1 << 24), "samx_tree_set: batch %lld out of range", (long long)n);
    if (n == 0) return SAMX_OK;
    const int levels = levels_of(cap);
    if (n >= 1024 && n <= kSetMaxBatch && cap >= 16 * (int64_t)kSetGroups) {
        // out-of-range indices are ignored by both forms (no subtree claims them)
        int top = 0;
        while ((1 << top) < kSetGroups) ++top;
        tree_set_sub_kernel<<<kSetGroups, kTreeThreads, 0, (hipStream_t)stream>>>(
            sum_tree, min_tree, stamp, cap, idxs, leaf, (int)n, levels - top);
        SAMX_LAUNCH_CHECK();
        tree_set_top_kernel<<<1, kTreeThreads, 0, (hipStream_t)stream>>>(sum_tree, min_tree, cap, levels - top,
                                                                        levels);
        SAMX_LAUNCH_CHECK();
        return SAMX_OK;
    }
    if (n < kTreeThreads && cap <= (int64_t(1) << 30)) {
        static samx::LdsGrant grant;
        if (int rc = grant.ensure((const void*)tree_set_fast_kernel, sizeof(samx::FastLds), 64 * 1024)) return rc;
        tree_set_fast_kernel<<<1, kTreeThreads, sizeof(samx::FastLds), (hipStream_t)stream>>>(
            sum_tree, min_tree, cap, idxs, leaf, (int)n, levels);
        SAMX_LAUNCH_CHECK();
        return SAMX_OK;
    }
    tree_set_kernel<<<1, kTreeThreads, 0, (hipStream_t)stream>>>(sum_tree, min_tree, stamp, cap, idxs,
                                                                leaf, (int)n, levels);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_tree_reduce(const double* tree, int64_t cap, int kind, int64_t start, int64_t end,
                     double* out, void* stream) {
    SAMX_REQUIRE(tree && out, "samx_tree_reduce: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_tree_reduce: cap not a power of two");
    SAMX_REQUIRE(kind == 0 || kind == 1, "samx_tree_reduce: kind must be 0 (sum) or 1 (min)");
    if (end < 0) end += cap;  // segment_tree.py:50-51
    int64_t hi = end - 1;     // segment_tree.py:52
    SAMX_REQUIRE(start >= 0 && hi >= start && hi < cap,
                 "samx_tree_reduce: empty or out-of-range query [%lld, %lld) on cap %lld",
                 (long long)start, (long long)end, (long long)cap);
    tree_reduce_kernel<<<1, 64, 0, (hipStream_t)stream>>>(tree, cap, kind, start, hi, out);  // one wave
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_tree_find_prefixsum(const double* sum_tree, int64_t cap, const double* prefix, int64_t n,
                             int64_t* idx_out, void* stream) {
    SAMX_REQUIRE(sum_tree && prefix && idx_out, "samx_tree_find_prefixsum: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_tree_find_prefixsum: bad cap");
    if (n <= 0) return SAMX_OK;
    tree_find_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(sum_tree, cap,
                                                                                  prefix, n, idx_out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_per_sample(const double* sum_tree, const double* min_tree, int64_t cap,
                    int64_t num_entries, const double* u, int64_t B, double beta,
                    int64_t* idxs_out, double* iws_out, float* iws_f32_out, void* stream) {
    SAMX_REQUIRE(sum_tree && min_tree && u && idxs_out && iws_out, "samx_per_sample: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_per_sample: bad cap");
    // num_entries == 1 makes the reference issue the degenerate query reduce(0, -1)
    // (memory.py:270 with segment_tree.py:52); refuse instead of emulating it.
    SAMX_REQUIRE(num_entries >= 2 && num_entries <= cap,
                 "samx_per_sample: num_entries %lld must be in [2, cap]", (long long)num_entries);
    SAMX_REQUIRE(B > 0, "samx_per_sample: batch must be positive");
    SAMX_REQUIRE(beta > 0, "samx_per_sample: beta must be positive");
    per_sample_kernel<<<(unsigned)((B + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        sum_tree, min_tree, cap, num_entries, u, B, beta, idxs_out, iws_out, iws_f32_out, samx::SampleExt());
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_per_sample_rows(const double* sum_tree, const double* min_tree, int64_t cap, const int64_t* geom,
                         int64_t ring_len, const double* u, int64_t B, double beta, int64_t* idxs_out,
                         double* iws_out, float* iws_f32_out, int64_t* rows_out, int64_t* idx_copy_out,
                         void* stream) {
    SAMX_REQUIRE(sum_tree && min_tree && u && idxs_out && iws_out && geom, "samx_per_sample_rows: null pointer");
    SAMX_REQUIRE(cap > 0 && (cap & (cap - 1)) == 0, "samx_per_sample_rows: bad cap");
    SAMX_REQUIRE(ring_len > 0 && ring_len <= cap, "samx_per_sample_rows: ring_len %lld must be in [1, cap]",
                 (long long)ring_len);
    SAMX_REQUIRE(B > 0 && beta > 0, "samx_per_sample_rows: batch and beta must be positive");
    samx::SampleExt x;
    x.geom = geom; x.limit = ring_len; x.rows_out = rows_out; x.idx_copy = idx_copy_out;
    per_sample_kernel<<<(unsigned)((B + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        sum_tree, min_tree, cap, 0, u, B, beta, idxs_out, iws_out, iws_f32_out, x);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_per_leaf_values_host(const int64_t* idxs_host, const double* prios_host, int64_t n,
                              double alpha, int64_t num_demos, double demos_eps,
                              double* leaf_out_host, double* max_priority_host) {
    SAMX_REQUIRE(prios_host && leaf_out_host && max_priority_host,
                 "samx_per_leaf_values_host: null pointer");
    SAMX_REQUIRE(num_demos == 0 || idxs_host, "samx_per_leaf_values_host: idxs needed with demos");
    double mp = *max_priority_host;
    for (int64_t i = 0; i < n; ++i) {
        double p = prios_host[i];
        SAMX_REQUIRE(p > 0, "samx_per_leaf_values_host: priority %lld is not positive",
                     (long long)i);  // memory.py:371
        if (num_demos > 0 && idxs_host[i] < num_demos) p += demos_eps;
        leaf_out_host[i] = pow(p, alpha);
        if (p > mp) mp = p;
    }
    *max_priority_host = mp;
    return SAMX_OK;
}

int samx_per_leaves_from_td(const float* td, const int64_t* idxs, int64_t n, int64_t cap, int64_t num_entries,
                            double alpha, int64_t num_demos, double demos_eps, double* leaf_out,
                            void* per_state, void* stream) {
    SAMX_REQUIRE(td && idxs && leaf_out && per_state, "samx_per_leaves_from_td: null pointer");
    SAMX_REQUIRE(n > 0 && n <= (1 << 24), "samx_per_leaves_from_td: batch %lld out of range", (long long)n);
    SAMX_REQUIRE(cap > 0 && num_entries >= 0 && num_entries <= cap, "samx_per_leaves_from_td: bad cap / num_entries");
    SAMX_REQUIRE(samx_pow::in_domain(1.0, alpha), "samx_per_leaves_from_td: alpha %g outside the pow domain", alpha);
    per_leaves_from_td_kernel<<<1, kTreeThreads, 0, (hipStream_t)stream>>>(
        td, idxs, (int)n, cap, num_entries, alpha, num_demos, demos_eps, leaf_out, (samx::PerState*)per_state,
        nullptr);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_per_leaves_from_td_geom(const float* td, const int64_t* idxs, int64_t n, int64_t cap, const int64_t* geom,
                                 double alpha, int64_t num_demos, double demos_eps, double* leaf_out,
                                 void* per_state, void* stream) {
    SAMX_REQUIRE(td && idxs && leaf_out && per_state && geom, "samx_per_leaves_from_td_geom: null pointer");
    SAMX_REQUIRE(n > 0 && n <= (1 << 24), "samx_per_leaves_from_td_geom: batch %lld out of range", (long long)n);
    SAMX_REQUIRE(cap > 0, "samx_per_leaves_from_td_geom: bad cap");
    SAMX_REQUIRE(samx_pow::in_domain(1.0, alpha), "samx_per_leaves_from_td_geom: alpha %g outside the pow domain",
                 alpha);
    per_leaves_from_td_kernel<<<1, kTreeThreads, 0, (hipStream_t)stream>>>(
        td, idxs, (int)n, cap, 0, alpha, num_demos, demos_eps, leaf_out, (samx::PerState*)per_state, geom);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_pow_device(const double* x, double y, int64_t n, double* out, void* stream) {
    SAMX_REQUIRE(x && out && n > 0, "samx_pow_device: bad arguments");
    pow_bits_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_pow_device_f32_range(uint32_t first_bits, int64_t n, double bonus, double y, double* out, void* stream) {
    SAMX_REQUIRE(out && n > 0 && n <= (int64_t(1) << 31), "samx_pow_device_f32_range: bad arguments");
    pow_bits_f32_range_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(first_bits, n, bonus, y, out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_pow_host_f32_range(uint32_t first_bits, int64_t n, double bonus, double y, double* out_host) {
    SAMX_REQUIRE(out_host && n > 0, "samx_pow_host_f32_range: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t u = first_bits + (uint32_t)i;
        float f;
        memcpy(&f, &u, 4);
        out_host[i] = pow((double)f + bonus, y);   // the platform libm: what the reference's Python calls
    }
    return SAMX_OK;
}

int samx_pow_host(const double* x_host, double y, int64_t n, double* out_host) {
    SAMX_REQUIRE(x_host && out_host && n > 0, "samx_pow_host: bad arguments");
    for (int64_t i = 0; i < n; ++i) out_host[i] = pow(x_host[i], y);
    return SAMX_OK;
}

int64_t samx_pow_selfcheck(int64_t n) {
    // the HOST build of pow_bits against this machine's libm pow on n pseudo-random bases and the two usual
    // exponent signs; a machine whose libm takes another code path (no FMA/AVX2, another glibc) shows up
    // here as a non-zero count and the callers keep the host round trip
    uint64_t s = 0x9E3779B97F4A7C15ull;
    int64_t bad = 0;
    const double ys[4] = {0.3, 0.6, -0.4, -1.0};
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint64_t ex = 1023 - 30 + (s >> 52) % 61;
        const uint64_t u = (ex << 52) | ((s * 0x2545F4914F6CDD1Dull) & ((1ull << 52) - 1));
        double x;
        memcpy(&x, &u, 8);
        const double y = ys[i & 3];
        const double a = samx_pow::pow_bits(x, y), b = pow(x, y);
        if (memcmp(&a, &b, 8) != 0) ++bad;
    }
    return bad;
}

int samx_per_leaves_from_td_host(const float* td_host, const int64_t* idxs_host, int64_t n,
                                 int64_t num_entries, double alpha, int64_t num_demos, double demos_eps,
                                 double* leaf_out_host, double* max_priority_host) {
    SAMX_REQUIRE(td_host && idxs_host && leaf_out_host && max_priority_host,
                 "samx_per_leaves_from_td_host: null pointer");
    double mp = *max_priority_host;
    for (int64_t i = 0; i < n; ++i) {
        // new priority = |td| + 1e-6 in FLOAT32 (numpy: float32 array + python float stays float32,
        // sam_agent.py:702), then float64 for the tree (memory.py:344-379)
        const float p32 = fabsf(td_host[i]) + 1e-6f;
        double p = (double)p32;
        SAMX_REQUIRE(p > 0, "samx_per_leaves_from_td_host: priority %lld is not positive", (long long)i);
        SAMX_REQUIRE(idxs_host[i] >= 0 && idxs_host[i] < num_entries,
                     "samx_per_leaves_from_td_host: no element in buffer associated w/ index %lld",
                     (long long)idxs_host[i]);   // memory.py:372
        if (num_demos > 0 && idxs_host[i] < num_demos) p += demos_eps;
        leaf_out_host[i] = pow(p, alpha);
        if (p > mp) mp = p;
    }
    *max_priority_host = mp;
    return SAMX_OK;
}

}  // extern "C"
