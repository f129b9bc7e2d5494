// Device-side bodies of the segment-tree kernels (shared by tree.hip and the PER row prologue of step.hip):
// the one-workgroup batch update, the range reduction in the reference's association order, the prefix-sum
// descent and the stratified sampling with importance weights.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "pow_bits.h"

namespace samx {

constexpr int kTreeThreads = 1024;
constexpr int kMaxLevels = 48;

__device__ __forceinline__ double tree_op(int kind, double a, double b) {
    // kind 0: operator.add ; kind 1: Python min(a, b) = b if b < a else a
    if (kind == 0) return __dadd_rn(a, b);
    return (b < a) ? b : a;
}


// State words of a prioritized buffer kept on the device when the priority refresh runs there:
// [0] max_priority (memory.py:379), [1] error bits as a double-sized word (bit 0: a priority that is not
// positive, memory.py:371; bit 1: an index without an element behind it, memory.py:372).
struct PerState {
    double max_priority;
    unsigned long long err;
};

// One entry of the priority refresh (sam_agent.py:702 -> memory.py:370-379): priority = |td| + 1e-6 in
// float32, widened; the demonstration bonus; leaf = priority ** alpha with the bits of the host pow
// (pow_bits.h).  Same arithmetic as samx_per_leaves_from_td_host.
__device__ __forceinline__ double leaf_from_td(float td, int64_t idx, int64_t num_demos, double demos_eps,
                                               double alpha, double* prio) {
    const float p32 = __fadd_rn(fabsf(td), 1e-6f);
    double p = (double)p32;
    if (num_demos > 0 && idx < num_demos) p = __dadd_rn(p, demos_eps);
    *prio = p;
    return samx_pow::pow_bits(p, alpha);
}

// The refresh of n entries by one workgroup (every thread calls this; barrier inside): leaves -> leaf_out
// (LDS or global), max_priority and the reference's two asserts -> state.  Entries whose index is outside
// [0, cap) are the "nothing pending" marker and are skipped.
__device__ __forceinline__ void leaves_from_td_body(const float* __restrict__ td, const int64_t* __restrict__ idxs,
                                                    int n, int64_t cap, int64_t num_entries, double alpha,
                                                    int64_t num_demos, double demos_eps, double* leaf_out,
                                                    PerState* state) {
    __shared__ double wmax[kTreeThreads / 64];
    __shared__ unsigned werr[kTreeThreads / 64];
    const int tid = threadIdx.x;
    double mx = 0.0;
    unsigned err = 0;
    for (int i = tid; i < n; i += kTreeThreads) {
        const int64_t idx = idxs[i];
        if (idx < 0 || idx >= cap) { leaf_out[i] = 0.0; continue; }
        double p;
        leaf_out[i] = leaf_from_td(td[i], idx, num_demos, demos_eps, alpha, &p);
        if (!(p > 0.0)) err |= 1u;
        if (idx >= num_entries) err |= 2u;
        if (p > mx) mx = p;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_xor(mx, o, 64);
        if (other > mx) mx = other;
        err |= (unsigned)__shfl_xor((int)err, o, 64);
    }
    if ((tid & 63) == 0) { wmax[tid >> 6] = mx; werr[tid >> 6] = err; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kTreeThreads / 64; ++w) {
            if (wmax[w] > mx) mx = wmax[w];
            err |= werr[w];
        }
        if (mx > state->max_priority) state->max_priority = mx;
        if (err) state->err |= err;
    }
    __syncthreads();
}

// One workgroup of kTreeThreads. Phase 1 elects, per distinct leaf, the LAST position in the batch (the
// reference loop is sequential, so the last write survives); phase 2 writes the elected leaves; then every
// level is recomputed bottom-up as value[2i] + value[2i+1] (or min), one barrier per level.
// Threads that share an ancestor compute and store the same value, which is harmless.
// (helpers/segment_tree.py:55-64 applied to a batch; memory.py:373-379)
__device__ __forceinline__ void tree_set_body(double* __restrict__ s, double* __restrict__ m,
                                              int32_t* __restrict__ stamp, int64_t cap,
                                              const int64_t* __restrict__ idxs,
                                              const double* leaf, int n, int levels) {
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += kTreeThreads) {
        int64_t idx = idxs[i];
        if (idx >= 0 && idx < cap) atomicMax(&stamp[idx], i);
    }
    __syncthreads();
    for (int i = tid; i < n; i += kTreeThreads) {
        int64_t idx = idxs[i];
        if (idx >= 0 && idx < cap) {
            int32_t w = __hip_atomic_load(&stamp[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == i) {
                double v = leaf[i];
                s[cap + idx] = v;
                m[cap + idx] = v;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += kTreeThreads) {
        int64_t idx = idxs[i];
        if (idx >= 0 && idx < cap) stamp[idx] = -1;
    }
    for (int l = 1; l <= levels; ++l) {
        __syncthreads();
        for (int i = tid; i < n; i += kTreeThreads) {
            int64_t idx = idxs[i];
            if (idx < 0 || idx >= cap) continue;
            int64_t node = (cap + idx) >> l;
            s[node] = __dadd_rn(s[2 * node], s[2 * node + 1]);
            double a = m[2 * node], b = m[2 * node + 1];
            m[node] = (b < a) ? b : a;
        }
    }
}

// reduce over leaves [lo, hi] (inclusive) with the association order of the reference recursion
// (segment_tree.py:26-39): descend to the split node, then the left part is a suffix query folded
// left-nested and the right part a prefix query folded right-nested.
// Done by one WAVE with ONE round of loads: lane 0 walks the tree with integer logic only
// and writes the node numbers the reference recursion touches into `nodes` (LDS, kMaxLevels * 2 + 2
// entries): [left base | left terms ... | right base | right terms ...]; every lane then loads one of
// them (all loads in flight together instead of a chain of dependent L2 round trips), and lane 0 folds
// them in the reference's association order.  Result valid in lane 0.
__device__ double range_reduce_wave(const double* __restrict__ v, int64_t cap, int kind, int64_t lo,
                                    int64_t hi, int lane, int64_t* nodes, double* vals, int* counts) {
    if (lane == 0) {
        int64_t node = 1, nlo = 0, nhi = cap - 1;
        bool single = false;
        while (true) {
            if (lo == nlo && hi == nhi) { single = true; break; }
            const int64_t mid = (nlo + nhi) / 2;
            if (hi <= mid) { node = 2 * node; nhi = mid; }
            else if (lo >= mid + 1) { node = 2 * node + 1; nlo = mid + 1; }
            else break;
        }
        if (single) {
            nodes[0] = node;
            counts[0] = -1; counts[1] = 0;
        } else {
            const int64_t mid0 = (nlo + nhi) / 2;
            int n = 1, nl = 0, nr = 0;
            {   // left child: suffix [lo, mid0]
                int64_t nd = 2 * node, a = nlo, b = mid0;
                while (lo != a) {
                    const int64_t mid = (a + b) / 2;
                    if (lo >= mid + 1) { nd = 2 * nd + 1; a = mid + 1; }
                    else { nodes[n++] = 2 * nd + 1; ++nl; nd = 2 * nd; b = mid; }
                }
                nodes[0] = nd;
            }
            const int rbase = n++;
            {   // right child: prefix [mid0+1, hi]
                int64_t nd = 2 * node + 1, a = mid0 + 1, b = nhi;
                while (hi != b) {
                    const int64_t mid = (a + b) / 2;
                    if (hi <= mid) { nd = 2 * nd; b = mid; }
                    else { nodes[n++] = 2 * nd; ++nr; nd = 2 * nd + 1; a = mid + 1; }
                }
                nodes[rbase] = nd;
            }
            counts[0] = nl; counts[1] = nr;
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nl = counts[0], nr = counts[1];
    const int total = nl < 0 ? 1 : nl + nr + 2;
    for (int t = lane; t < total; t += 64) vals[t] = v[nodes[t]];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    double out = 0.0;
    if (lane == 0) {
        if (nl < 0) {
            out = vals[0];
        } else {
            double left = vals[0];
            for (int t = nl - 1; t >= 0; --t) left = tree_op(kind, left, vals[1 + t]);
            const int rb = nl + 1;
            double right = vals[rb];
            for (int t = nr - 1; t >= 0; --t) right = tree_op(kind, vals[rb + 1 + t], right);
            out = tree_op(kind, left, right);
        }
    }
    return out;
}


constexpr int kTopLevels = 11;               // levels of the sum tree a sampling block keeps in LDS
constexpr int kTopNodes = 1 << kTopLevels;   // nodes 1 .. kTopNodes-1

// find_prefixsum_idx (segment_tree.py:82-97): the top levels from LDS (`top`, valid when top_n > 0),
// the rest from memory
__device__ __forceinline__ int64_t prefix_descent(const double* __restrict__ v, int64_t cap,
                                                  double s, const double* top = nullptr, int64_t top_n = 0) {
    int64_t node = 1;
    while (2 * node < top_n) {
        const double left = top[2 * node];
        if (left > s) {
            node = 2 * node;
        } else {
            s = __dadd_rn(s, -left);
            node = 2 * node + 1;
        }
    }
    while (node < cap) {
        double left = v[2 * node];
        if (left > s) {
            node = 2 * node;
        } else {
            s = __dadd_rn(s, -left);
            node = 2 * node + 1;
        }
    }
    return node - cap;
}


// PrioritizedRB._sample_w_priorities + the importance weights (memory.py:266-309) for samples
// first .. first + span of a batch of B, by ONE workgroup in which every thread calls this (barrier
// inside); the threads with threadIdx.x < 256 work: three waves take the three scalar range reductions,
// each with one round of loads (range_reduce_wave), meanwhile the top levels of the sum tree are copied
// into LDS for the descents.
__device__ __forceinline__ void per_sample_body(const double* __restrict__ sumv, const double* __restrict__ minv,
                                                int64_t cap, int64_t n_ent, const double* __restrict__ u,
                                                int64_t B, double beta, int64_t* __restrict__ idxs,
                                                double* __restrict__ iws, float* __restrict__ iws32,
                                                int64_t first, int64_t span) {
    __shared__ double sh[3];
    __shared__ int64_t nodes[3][2 * kMaxLevels + 2];
    __shared__ double vals[3][2 * kMaxLevels + 2];
    __shared__ int counts[3][2];
    __shared__ double top[kTopNodes];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t top_n = cap < kTopNodes ? cap : kTopNodes;   // nodes [1, top_n) are internal or leaves < 2 cap
    if (tid < 256) {
        for (int64_t t = tid; t < top_n; t += 256) top[t] = t ? sumv[t] : 0.0;
        if (wv < 3) {
            const double r = wv == 0 ? range_reduce_wave(sumv, cap, 0, 0, n_ent - 2, lane, nodes[0], vals[0], counts[0])  // memory.py:270
                           : wv == 1 ? range_reduce_wave(sumv, cap, 0, 0, n_ent - 1, lane, nodes[1], vals[1], counts[1])  // memory.py:299,306
                                     : range_reduce_wave(minv, cap, 1, 0, n_ent - 1, lane, nodes[2], vals[2], counts[2]);
            if (lane == 0) sh[wv] = r;
        }
    }
    __syncthreads();
    if (tid >= 256) return;
    const double p_total = sh[0], total = sh[1], lowest_p = sh[2];
    const double piece = p_total / (double)B;
    const double n = (double)n_ent;
    const double lowest = lowest_p / total;
    const double max_w = samx_pow::pow_bits(__dmul_rn(n, lowest), -beta);
    for (int64_t i = first + tid; i < B && i < first + span; i += 256) {
        const double s = __dadd_rn(__dmul_rn(u[i], piece), __dmul_rn((double)i, piece));
        const int64_t idx = prefix_descent(sumv, cap, s, top, top_n);
        idxs[i] = idx;
        const double prob = sumv[cap + idx] / total;
        const double w = samx_pow::pow_bits(__dmul_rn(n, prob), -beta) / max_w;
        iws[i] = w;
        if (iws32) iws32[i] = (float)w;
    }
}

}  // namespace samx
