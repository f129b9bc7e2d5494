// Device-side body of the one-workgroup segment-tree batch update (shared by tree.hip and the PER row
// prologue of step.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace samx {

constexpr int kTreeThreads = 1024;

// One workgroup of kTreeThreads. Phase 1 elects, per distinct leaf, the LAST position in the batch (the
// reference loop is sequential, so the last write survives); phase 2 writes the elected leaves; then every
// level is recomputed bottom-up as value[2i] + value[2i+1] (or min), one barrier per level.
// Threads that share an ancestor compute and store the same value, which is harmless.
// (helpers/segment_tree.py:55-64 applied to a batch; memory.py:373-379)
__device__ __forceinline__ void tree_set_body(double* __restrict__ s, double* __restrict__ m,
                                              int32_t* __restrict__ stamp, int64_t cap,
                                              const int64_t* __restrict__ idxs,
                                              const double* __restrict__ leaf, int n, int levels) {
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

}  // namespace samx
