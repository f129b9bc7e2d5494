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

// ---- the same batch update with the dependent chain kept out of HBM ---------------------------------
// tree_set_body pays one L2 round trip + barrier per level (20 levels at cap 2^20: ~20 us).  Here, for
// n < 1024 entries (one per thread), ONE workgroup:
//   * loads, all at once, the top 12 levels of both trees into LDS and, per entry, the OLD value of the
//     sibling at every level below them (independent loads: one round trip);
//   * resolves duplicates (last position wins, as the reference's sequential loop) through an LDS hash;
//   * walks the lower levels in LDS: every touched node has exactly one carrier thread; a carrier finds its
//     sibling's carrier through the LDS hash (both children touched: the even child's carrier goes on) or
//     takes the preloaded old value; parents are stored to HBM as they are formed (nobody waits for them);
//   * recomputes the top levels densely in LDS (every node from its two children — untouched nodes get the
//     value they already had, as in tree_set_top_kernel) and stores them.
// Every node is still left + right (or min) of its two children in float64: the same bits as the reference.
// The LDS copy of the sum tree's top stays valid for the sampling that follows (per_sample_body).
#ifndef SAMX_TREE_STAMP
#define SAMX_TREE_STAMP(i)            // phase stamps of tools/exp/tree_fast_trace.hip
#endif
constexpr int kFastTop = 4096;        // nodes [1, 4096) of each tree
constexpr int kFastTopLevels = 12;
constexpr int kFastHash = 2048;       // slots (>= 2 x entries)
constexpr int kFastMaxLb = 19;        // hash-walked levels: cap <= 2^30
struct FastLds {
    double top_s[kFastTop];
    double top_m[kFastTop];
    double cs[kTreeThreads];
    double cm[kTreeThreads];
    int hkey[2][kFastHash];
    int hown[2][kFastHash];
};

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the global
// stores in flight (a full write round trip per level — what made the levelled update slow in the first place);
// inside the update only LDS is shared between the waves, the stores to HBM are drained once, at the end.
__device__ __forceinline__ void lds_bar() {
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    asm volatile("s_barrier" ::: "memory");
}

__device__ __forceinline__ int fast_hash(int key) { return (int)(((unsigned)key * 2654435761u) >> 21); }

__device__ __forceinline__ bool tree_set_fast_ok(int64_t cap, int64_t n) {
    return n > 0 && n < kTreeThreads && cap <= (int64_t(1) << 30);
}

// every thread of a workgroup of kTreeThreads calls this; `leaf` may point into L.cs / L.cm
__device__ __forceinline__ void tree_set_fast_body(double* __restrict__ s, double* __restrict__ m, int64_t cap,
                                                   const int64_t* __restrict__ idxs, const double* leaf, int n,
                                                   int levels, FastLds& L) {
    const int tid = threadIdx.x;
    const int topn = (int)(2 * cap < kFastTop ? 2 * cap : kFastTop);
    const int lb = levels + 1 > kFastTopLevels ? levels + 1 - kFastTopLevels : 0;   // level lb is the lowest one inside top
    for (int j = tid; j < 2 * kFastHash; j += kTreeThreads) {
        (&L.hkey[0][0])[j] = -1;
        (&L.hown[0][0])[j] = -1;
    }
    bool active = false;
    int node = 0;
    double vs = 0.0, vm = 0.0;
    if (tid < n) {
        const int64_t idx = idxs[tid];
        if (idx >= 0 && idx < cap) {
            active = true;
            node = (int)(cap + idx);
            vs = vm = leaf[tid];
        }
    }
    double sib_s[kFastMaxLb], sib_m[kFastMaxLb];
#pragma unroll
    for (int j = 0; j < kFastMaxLb; ++j) {
        sib_s[j] = 0.0;
        sib_m[j] = 0.0;
        if (j < lb && active) {
            const int sn = (node >> j) ^ 1;
            sib_s[j] = s[sn];
            sib_m[j] = m[sn];
        }
    }
    {   // top fill: all of a thread's loads in flight together
        double ts_[kFastTop / kTreeThreads], tm_[kFastTop / kTreeThreads];
#pragma unroll
        for (int q = 0; q < kFastTop / kTreeThreads; ++q) {
            const int j = tid + q * kTreeThreads;
            ts_[q] = (j && j < topn) ? s[j] : 0.0;
            tm_[q] = (j && j < topn) ? m[j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kFastTop / kTreeThreads; ++q) {
            const int j = tid + q * kTreeThreads;
            if (j < topn) { L.top_s[j] = ts_[q]; L.top_m[j] = tm_[q]; }
        }
    }
    SAMX_TREE_STAMP(1)
    lds_bar();   // tables empty, top filled, leaves read out of `leaf`
    SAMX_TREE_STAMP(2)
    int slot = 0;
    if (active) {      // duplicates: the last position in the batch wins
        slot = fast_hash(node);
        while (true) {
            const int prev = atomicCAS(&L.hkey[1][slot], -1, node);
            if (prev == -1 || prev == node) break;
            slot = (slot + 1) & (kFastHash - 1);
        }
        atomicMax(&L.hown[1][slot], tid);
    }
    lds_bar();
    SAMX_TREE_STAMP(3)
    if (active && L.hown[1][slot] != tid) active = false;
    if (active) {
        s[node] = vs;
        m[node] = vm;
        if (lb == 0) { L.top_s[node] = vs; L.top_m[node] = vm; }
    }
    // one hash-walked level (children at level l - 1 -> parents at level l); old_s / old_m: the preloaded
    // old value of this thread's sibling at level l - 1.  Spelled out per level below so that the preloaded
    // values stay in registers (constant indices).
    auto level = [&](const int l, const double old_s, const double old_m) {
        const int b = (l - 1) & 1;
        if (active) {
            L.cs[tid] = vs;
            L.cm[tid] = vm;
            slot = fast_hash(node);
            while (atomicCAS(&L.hkey[b][slot], -1, node) != -1) slot = (slot + 1) & (kFastHash - 1);
            L.hown[b][slot] = tid;
        }
        lds_bar();
        for (int j = tid; j < kFastHash; j += kTreeThreads) {   // the table of the level before (nobody reads it any more)
            L.hkey[b ^ 1][j] = -1;
            L.hown[b ^ 1][j] = -1;
        }
        if (active) {
            const int sn = node ^ 1;
            int sl = fast_hash(sn), o = -1;
            while (true) {
                const int k = L.hkey[b][sl];
                if (k == sn) { o = L.hown[b][sl]; break; }
                if (k == -1) break;
                sl = (sl + 1) & (kFastHash - 1);
            }
            const bool odd = node & 1;
            if (o >= 0 && odd) {
                active = false;   // both children touched: the even child's carrier forms the parent
            } else {
                const double ss = o >= 0 ? L.cs[o] : old_s, sm = o >= 0 ? L.cm[o] : old_m;
                const double ls = odd ? ss : vs, rs = odd ? vs : ss, lm = odd ? sm : vm, rm = odd ? vm : sm;
                vs = __dadd_rn(ls, rs);
                vm = (rm < lm) ? rm : lm;
                node >>= 1;
                s[node] = vs;
                m[node] = vm;
                if (l == lb) { L.top_s[node] = vs; L.top_m[node] = vm; }
            }
        }
        lds_bar();
    };
#define SAMX_TREE_LEVEL(LV) if (lb >= LV) level(LV, sib_s[LV - 1], sib_m[LV - 1]);
    SAMX_TREE_LEVEL(1) SAMX_TREE_LEVEL(2) SAMX_TREE_LEVEL(3) SAMX_TREE_LEVEL(4) SAMX_TREE_LEVEL(5)
    SAMX_TREE_LEVEL(6) SAMX_TREE_LEVEL(7) SAMX_TREE_LEVEL(8) SAMX_TREE_LEVEL(9) SAMX_TREE_LEVEL(10)
    SAMX_TREE_LEVEL(11) SAMX_TREE_LEVEL(12) SAMX_TREE_LEVEL(13) SAMX_TREE_LEVEL(14) SAMX_TREE_LEVEL(15)
    SAMX_TREE_LEVEL(16) SAMX_TREE_LEVEL(17) SAMX_TREE_LEVEL(18) SAMX_TREE_LEVEL(19)
#undef SAMX_TREE_LEVEL
    SAMX_TREE_STAMP(4)
    static_assert(kFastMaxLb == 19, "one SAMX_TREE_LEVEL per hash-walked level");
    lds_bar();
    for (int l = lb + 1; l <= levels; ++l) {
        const int count = (int)(cap >> l);
        for (int j = tid; j < count; j += kTreeThreads) {
            const int nd = count + j;
            L.top_s[nd] = __dadd_rn(L.top_s[2 * nd], L.top_s[2 * nd + 1]);
            const double a = L.top_m[2 * nd], b = L.top_m[2 * nd + 1];
            L.top_m[nd] = (b < a) ? b : a;
        }
        lds_bar();
    }
    SAMX_TREE_STAMP(5)
    const int wn = (int)(cap >> lb);   // nodes [1, wn): the densely recomputed levels
    for (int j = tid + 1; j < wn; j += kTreeThreads) {
        s[j] = L.top_s[j];
        m[j] = L.top_m[j];
    }
    __syncthreads();   // the stores above are what the sampling (and the next kernel) read
}

// ---- the same update spread over several workgroups --------------------------------------------------
// One workgroup's random 8-byte loads (every entry's old sibling at every level below the top: ~18 per
// entry) are bound by what ONE compute unit can keep in flight, not by latency.  So the lower levels are
// split by SUBTREE (as tree_set_sub_kernel does for large batches): workgroup g of G takes the entries whose
// leaf lies in [g cap/G, (g+1) cap/G), refreshes them (leaf_from_td) and walks them up to the lowest level
// that fits the top (tree_walk_sub_body); the workgroups of the NEXT launch each load that level (coalesced),
// recompute the top densely in LDS (tree_top_body; workgroup 0 stores it) and sample on it.
struct WalkLds {
    double hs[2][kFastHash];   // the carried node's sum / min, stored IN the hash slot: a sibling lookup reads key and
    double hm[2][kFastHash];   // values of the home slot in one round of LDS reads
    int hkey[2][kFastHash];
    int hown[2][kFastHash];    // (duplicates' table and the compaction list only)
};

__host__ __device__ __forceinline__ int fast_lb(int levels) {
    return levels + 1 > kFastTopLevels ? levels + 1 - kFastTopLevels : 0;
}

// entries [0, n) with leaf in [lo, hi): refresh from TD errors, duplicates, leaves and `lb` levels of parents.
// The workgroup first compacts its entries onto its first threads; waves left without an entry end at once
// (s_barrier only waits for the waves still alive), so a subtree with <= 64 entries - the usual case - is
// walked by ONE wave whose barriers cost nothing.
__device__ __forceinline__ void tree_walk_sub_body(double* __restrict__ s, double* __restrict__ m, int64_t cap,
                                                   const int64_t* __restrict__ idxs, const float* __restrict__ td,
                                                   int n, int levels, int64_t lo, int64_t hi, int64_t num_entries,
                                                   double alpha, int64_t num_demos, double demos_eps,
                                                   PerState* state, WalkLds& L) {
    __shared__ int mine_cnt;
    const int tid = threadIdx.x;
    const int lb = fast_lb(levels);
    if (tid == 0) mine_cnt = 0;
    for (int j = tid; j < 2 * kFastHash; j += kTreeThreads) {
        (&L.hkey[0][0])[j] = -1;
        (&L.hown[0][0])[j] = -1;
    }
    __syncthreads();
    if (tid < n) {
        const int64_t idx = idxs[tid];
        if (idx >= lo && idx < hi && idx >= 0 && idx < cap) {
            const int pos = atomicAdd(&mine_cnt, 1);
            L.hown[0][pos] = tid;                       // list of this subtree's batch positions (n < 1024 <= slots)
            L.hkey[0][pos] = (int)(cap + idx);
        }
    }
    __syncthreads();
    const int cnt = mine_cnt;
    if ((tid & ~63) >= cnt) return;                     // this wave has nothing to carry
    bool active = tid < cnt;
    int node = 0, entry = -1;
    double vs = 0.0, vm = 0.0;
    if (active) { entry = L.hown[0][tid]; node = L.hkey[0][tid]; }
    lds_bar();                                          // (the surviving waves) list read ...
    for (int j = tid; j < cnt; j += kTreeThreads) {     // ... before its slots are emptied again (tid < cnt holds)
        L.hown[0][j] = -1;
        L.hkey[0][j] = -1;
    }
    SAMX_TREE_STAMP(8)
    double sib_s[kFastMaxLb], sib_m[kFastMaxLb];
#pragma unroll
    for (int j = 0; j < kFastMaxLb; ++j) {
        sib_s[j] = 0.0;
        sib_m[j] = 0.0;
        if (j < lb && active) {
            const int sn = (node >> j) ^ 1;
            sib_s[j] = s[sn];
            sib_m[j] = m[sn];
        }
    }
    SAMX_TREE_STAMP(9)
    if (active) {   // memory.py:370-379 for this entry; the state words take max / or from every workgroup
        double p;
        vs = vm = leaf_from_td(td[entry], node - cap, num_demos, demos_eps, alpha, &p);
        unsigned long long err = 0;
        if (!(p > 0.0)) err |= 1ull;
        if (node - cap >= num_entries) err |= 2ull;
        if (err) atomicOr(&state->err, err);
        // positive doubles order like their bit patterns
        if (p > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(&state->max_priority),
                               (unsigned long long)__double_as_longlong(p));
    }
    SAMX_TREE_STAMP(10)
    lds_bar();
    SAMX_TREE_STAMP(11)
    int slot = 0, dslot = -1;
    if (active) {      // duplicates: the last position in the batch wins
        slot = fast_hash(node);
        while (true) {
            const int prev = atomicCAS(&L.hkey[1][slot], -1, node);
            if (prev == -1 || prev == node) break;
            slot = (slot + 1) & (kFastHash - 1);
        }
        atomicMax(&L.hown[1][slot], entry);
        dslot = slot;
    }
    lds_bar();
    if (active && L.hown[1][slot] != entry) active = false;
    if (active) {
        s[node] = vs;
        m[node] = vm;
    }
    SAMX_TREE_STAMP(12)
    auto level = [&](const int l, const double old_s, const double old_m) {
        // (the tables are emptied slot by slot by the threads that filled them, a barrier after their last
        // reader: with one or two waves left a sweep over all slots would cost more than the level itself)
        const int b = (l - 1) & 1;
        int myslot = -1;
        if (l == 2) { SAMX_TREE_STAMP(19) }
        if (active) {
            slot = fast_hash(node);
            while (atomicCAS(&L.hkey[b][slot], -1, node) != -1) slot = (slot + 1) & (kFastHash - 1);
            L.hs[b][slot] = vs;
            L.hm[b][slot] = vm;
            myslot = slot;
        }
        if (l == 2) { SAMX_TREE_STAMP(15) }
        lds_bar();
        if (l == 2) { SAMX_TREE_STAMP(16) }
        if (l == 1 && dslot >= 0) {   // the duplicates' table: every reader passed the barrier above
            L.hkey[1][dslot] = -1;
            L.hown[1][dslot] = -1;
        }
        if (active) {
            const int sn = node ^ 1;
            int sl = fast_hash(sn);
            int k = L.hkey[b][sl];
            double fs = L.hs[b][sl], fm = L.hm[b][sl];   // read with the key: the home slot is almost always the one
            bool found = false;
            while (true) {
                if (k == sn) { found = true; break; }
                if (k == -1) break;
                sl = (sl + 1) & (kFastHash - 1);
                k = L.hkey[b][sl];
                fs = L.hs[b][sl];
                fm = L.hm[b][sl];
            }
            const bool odd = node & 1;
            if (found && odd) {
                active = false;
            } else {
                const double ss = found ? fs : old_s, sm = found ? fm : old_m;
                const double ls = odd ? ss : vs, rs = odd ? vs : ss, lm = odd ? sm : vm, rm = odd ? vm : sm;
                vs = __dadd_rn(ls, rs);
                vm = (rm < lm) ? rm : lm;
                node >>= 1;
                s[node] = vs;
                m[node] = vm;
            }
        }
        if (l == 2) { SAMX_TREE_STAMP(17) }
        lds_bar();
        if (l == 2) { SAMX_TREE_STAMP(18) }
        if (myslot >= 0) L.hkey[b][myslot] = -1;
    };
#define SAMX_TREE_LEVEL(LV) if (lb >= LV) level(LV, sib_s[LV - 1], sib_m[LV - 1]);
    SAMX_TREE_LEVEL(1) SAMX_TREE_LEVEL(2) SAMX_TREE_LEVEL(3) SAMX_TREE_LEVEL(4) SAMX_TREE_LEVEL(5)
    SAMX_TREE_LEVEL(6) SAMX_TREE_LEVEL(7) SAMX_TREE_LEVEL(8) SAMX_TREE_LEVEL(9) SAMX_TREE_LEVEL(10)
    SAMX_TREE_LEVEL(11) SAMX_TREE_LEVEL(12) SAMX_TREE_LEVEL(13) SAMX_TREE_LEVEL(14) SAMX_TREE_LEVEL(15)
    SAMX_TREE_LEVEL(16) SAMX_TREE_LEVEL(17) SAMX_TREE_LEVEL(18) SAMX_TREE_LEVEL(19)
#undef SAMX_TREE_LEVEL
    SAMX_TREE_STAMP(13)
}

// level lb of both trees (memory) -> LDS, every level above it recomputed densely; `store`: write the
// recomputed levels back (one workgroup of a launch does).  top_s / top_m: kFastTop doubles each.
__device__ __forceinline__ void tree_top_body(double* __restrict__ s, double* __restrict__ m, int64_t cap, int levels,
                                              double* top_s, double* top_m, bool store) {
    const int tid = threadIdx.x;
    const int lb = fast_lb(levels);
    const int wn = (int)(cap >> lb), topn = 2 * wn;   // level lb = nodes [wn, topn)
    {
        double ts_[kFastTop / 2 / kTreeThreads], tm_[kFastTop / 2 / kTreeThreads];
#pragma unroll
        for (int q = 0; q < kFastTop / 2 / kTreeThreads; ++q) {
            const int j = wn + tid + q * kTreeThreads;
            ts_[q] = j < topn ? s[j] : 0.0;
            tm_[q] = j < topn ? m[j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kFastTop / 2 / kTreeThreads; ++q) {
            const int j = wn + tid + q * kTreeThreads;
            if (j < topn) { top_s[j] = ts_[q]; top_m[j] = tm_[q]; }
        }
    }
    if (tid == 0) { top_s[0] = 0.0; top_m[0] = 0.0; }
    lds_bar();
    // Dense recompute with two workgroup barriers instead of one per level: wave w first folds its own
    // contiguous slice of level lb up to the slice's root (the waves' slices are disjoint subtrees, lanes of
    // one wave run in lockstep), then one wave folds the 16 slice roots up to node 1.
    auto fold = [&](const int nd) {
        top_s[nd] = __dadd_rn(top_s[2 * nd], top_s[2 * nd + 1]);
        const double a = top_m[2 * nd], b = top_m[2 * nd + 1];
        top_m[nd] = (b < a) ? b : a;
    };
    constexpr int kWaves = kTreeThreads / 64;
    const int wave = tid >> 6, lane = tid & 63;
    const int roots = wn >= kWaves ? kWaves : 1;         // slices (power of two); wn is a power of two
    if (wave < roots)
        for (int count = wn / 2; count >= roots; count >>= 1) {   // level of `count` nodes: [count, 2 count)
            const int per = count / roots;                       // ... `per` of them inside each slice
            for (int j = lane; j < per; j += 64) fold(count + wave * per + j);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    lds_bar();
    if (wave == 0)
        for (int count = roots / 2; count >= 1; count >>= 1) {
            if (lane < count) fold(count + lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    lds_bar();
    if (store)
        for (int j = tid + 1; j < wn; j += kTreeThreads) {
            s[j] = top_s[j];
            m[j] = top_m[j];
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
// `top` (optional): an LDS copy of nodes [1, topn) that is NEWER than memory (the caller recomputed it).
__device__ double range_reduce_wave(const double* __restrict__ v, int64_t cap, int kind, int64_t lo,
                                    int64_t hi, int lane, int64_t* nodes, double* vals, int* counts,
                                    const double* top = nullptr, int64_t topn = 0) {
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
    for (int t = lane; t < total; t += 64) vals[t] = nodes[t] < topn ? top[nodes[t]] : v[nodes[t]];
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
// leaf_val (optional): receives the value of the leaf found when the last memory round reached the leaves (its
// odd siblings are loaded in that round too), *have_leaf says whether it did.
__device__ __forceinline__ int64_t prefix_descent(const double* __restrict__ v, int64_t cap,
                                                  double s, const double* top = nullptr, int64_t top_n = 0,
                                                  double* leaf_val = nullptr, bool* have_leaf = nullptr) {
    if (have_leaf) *have_leaf = false;
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
    // Below the LDS levels every step would be one dependent round trip to memory.  Three levels per round
    // instead: the left sums of all 1 + 2 + 4 nodes the next three decisions can reach are loaded together
    // (7 independent loads), then the three decisions are taken in order with the same comparisons and
    // subtractions as the one-level loop.
    while (node < cap) {
        const int64_t n2 = 2 * node, n4 = 4 * node, n8 = 8 * node;
        const bool l2 = n2 < cap, l3 = n4 < cap;
        const double a = v[n2];
        double b[2] = {0.0, 0.0}, c[4] = {0.0, 0.0, 0.0, 0.0};
        if (l2) { b[0] = v[n4]; b[1] = v[n4 + 2]; }
        if (l3) { c[0] = v[n8]; c[1] = v[n8 + 2]; c[2] = v[n8 + 4]; c[3] = v[n8 + 6]; }
        const bool last = leaf_val && (l3 ? n8 : l2 ? n4 : n2) >= cap;   // this round ends on the leaves
        double o[4] = {0.0, 0.0, 0.0, 0.0};                               // ... then their odd members too
        if (last) {
            if (l3) { o[0] = v[n8 + 1]; o[1] = v[n8 + 3]; o[2] = v[n8 + 5]; o[3] = v[n8 + 7]; }
            else if (l2) { o[0] = v[n4 + 1]; o[1] = v[n4 + 3]; }
            else o[0] = v[n2 + 1];
        }
        if (a > s) { node = n2; } else { s = __dadd_rn(s, -a); node = n2 + 1; }
        if (l2) {
            const double left = (node & 1) ? b[1] : b[0];
            if (left > s) { node = 2 * node; } else { s = __dadd_rn(s, -left); node = 2 * node + 1; }
        }
        if (l3) {
            const int k = (int)(node & 3);
            const double left = k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3];
            if (left > s) { node = 2 * node; } else { s = __dadd_rn(s, -left); node = 2 * node + 1; }
        }
        if (last) {
            const bool odd = node & 1;
            const int k = (int)((node >> 1) & 3);
            double val;
            if (l3) val = odd ? (k == 0 ? o[0] : k == 1 ? o[1] : k == 2 ? o[2] : o[3])
                              : (k == 0 ? c[0] : k == 1 ? c[1] : k == 2 ? c[2] : c[3]);
            else if (l2) val = odd ? ((k & 1) ? o[1] : o[0]) : ((k & 1) ? b[1] : b[0]);
            else val = odd ? o[0] : a;
            *leaf_val = val;
            *have_leaf = true;
        }
    }
    return node - cap;
}


// Ring geometry read from DEVICE memory and the extra outputs that go with it (every member optional).  A sampling
// kernel inside a captured graph must not carry num_entries / the ring's start as launch arguments: every
// store_transition between two training phases changes them (memory.py:36-57,337-342) and the graph would have to be
// captured again.  geom = int64[2] {ring start, num_entries} as the upload of the launch left them.
struct SampleExt {
    const int64_t* geom = nullptr;   // device words; num_entries is clamped to [2, cap] (memory safety; the host
                                     // checked the real value when it wrote the words)
    int64_t limit = 0;               // ring length
    int64_t* rows_out = nullptr;     // storage rows (start + idx) % limit of the sampled entries: what the row
                                     // gathers of the network kernels read (their ring then "starts at 0")
    int64_t* idx_copy = nullptr;     // second copy of the sampled LOGICAL indices (the pending-update buffer)
};

// PrioritizedRB._sample_w_priorities + the importance weights (memory.py:266-309) for samples
// first .. first + span of a batch of B, by ONE workgroup in which every thread calls this (barrier
// inside); the threads with threadIdx.x < 256 work: three waves take the three scalar range reductions,
// each with one round of loads (range_reduce_wave), meanwhile the top levels of the sum tree are copied
// into LDS for the descents.
__device__ __forceinline__ void per_sample_body(const double* __restrict__ sumv, const double* __restrict__ minv,
                                                int64_t cap, int64_t n_ent, const double* __restrict__ u,
                                                int64_t B, double beta, int64_t* __restrict__ idxs,
                                                double* __restrict__ iws, float* __restrict__ iws32,
                                                int64_t first, int64_t span, const double* ext_top = nullptr,
                                                int64_t ext_top_n = 0, const double* ext_top_min = nullptr,
                                                const SampleExt& x = SampleExt()) {
    int64_t ring_start = 0;
    if (x.geom) {
        ring_start = x.geom[0];
        n_ent = x.geom[1];
        n_ent = n_ent < 2 ? 2 : (n_ent > cap ? cap : n_ent);
    }
    __shared__ double sh[3];
    __shared__ int64_t nodes[3][2 * kMaxLevels + 2];
    __shared__ double vals[3][2 * kMaxLevels + 2];
    __shared__ int counts[3][2];
    __shared__ double own_top[kTopNodes];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // nodes [1, top_n) are internal or leaves < 2 cap; ext_top: the caller's LDS copy of the sum tree's top
    const int64_t top_n = ext_top ? ext_top_n : (cap < kTopNodes ? cap : kTopNodes);
    const double* top = ext_top ? ext_top : own_top;
    if (tid < 256) {
        if (!ext_top)
            for (int64_t t = tid; t < top_n; t += 256) own_top[t] = t ? sumv[t] : 0.0;
        if (wv < 3) {
            // (ext_top_min given: the caller's LDS tops are newer than memory and serve the reductions too)
            const double* rs = ext_top_min ? ext_top : nullptr;
            const int64_t rn = ext_top_min ? ext_top_n : 0;
            const double r = wv == 0 ? range_reduce_wave(sumv, cap, 0, 0, n_ent - 2, lane, nodes[0], vals[0], counts[0], rs, rn)  // memory.py:270
                           : wv == 1 ? range_reduce_wave(sumv, cap, 0, 0, n_ent - 1, lane, nodes[1], vals[1], counts[1], rs, rn)  // memory.py:299,306
                                     : range_reduce_wave(minv, cap, 1, 0, n_ent - 1, lane, nodes[2], vals[2], counts[2], ext_top_min, rn);
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
        double lv = 0.0;
        bool have = false;
        const int64_t idx = prefix_descent(sumv, cap, s, top, top_n, &lv, &have);
        idxs[i] = idx;
        if (x.idx_copy) x.idx_copy[i] = idx;
        if (x.rows_out) {
            int64_t r = ring_start + idx;
            if (r >= x.limit) r -= x.limit;
            x.rows_out[i] = r;
        }
        const double prob = (have ? lv : sumv[cap + idx]) / total;
        const double w = samx_pow::pow_bits(__dmul_rn(n, prob), -beta) / max_w;
        iws[i] = w;
        if (iws32) iws32[i] = (float)w;
    }
}

}  // namespace samx
