/* CPU oracle (plain C) for the prioritized-replay segment trees, sized for full-capacity cases
 * (cap 2^20 / 2^22) that the Python oracle is too slow to fill.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg through ctypes; the product never links or calls it.
 *
 * Restates /root/reference/imitation/helpers/segment_tree.py:26-97 and the tree-touching parts of
 * /root/reference/imitation/imitation_algorithms/memory.py:248-309,337-379.  Pinned indirectly:
 * tests/test_oracle_replay.py::test_c_oracle_matches_python checks it bit for bit against
 * oracle/replay_oracle.py, which is itself pinned on vectors recorded from the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC oracle/per_tree.c -o oracle/libper_oracle.so -lm
 * (-ffp-contract=off: the reference never fuses a multiply into an add).
 */
#include <math.h>
#include <stdint.h>

/* kind: 0 = sum tree, 1 = min tree.  v has 2*cap doubles, root at 1, leaves at cap.. */
static double op(int kind, double a, double b) {
    if (kind == 0) return a + b;
    return (b < a) ? b : a; /* Python min(a, b) */
}

void pt_init(double *v, int64_t cap, int kind) {
    for (int64_t i = 0; i < 2 * cap; ++i) v[i] = kind == 0 ? 0.0 : INFINITY;
}

/* segment_tree.py:55-64 */
void pt_set(double *v, int64_t cap, int kind, int64_t idx, double val) {
    int64_t node = idx + cap;
    v[node] = val;
    for (node >>= 1; node >= 1; node >>= 1) v[node] = op(kind, v[2 * node], v[2 * node + 1]);
}

/* segment_tree.py:26-39 — same recursion, hence same association order */
static double walk(const double *v, int kind, int64_t lo, int64_t hi, int64_t node, int64_t nlo,
                   int64_t nhi) {
    if (lo == nlo && hi == nhi) return v[node];
    int64_t mid = (nlo + nhi) / 2;
    if (hi <= mid) return walk(v, kind, lo, hi, 2 * node, nlo, mid);
    if (lo >= mid + 1) return walk(v, kind, lo, hi, 2 * node + 1, mid + 1, nhi);
    double l = walk(v, kind, lo, mid, 2 * node, nlo, mid);
    double r = walk(v, kind, mid + 1, hi, 2 * node + 1, mid + 1, nhi);
    return op(kind, l, r);
}

/* segment_tree.py:41-53 ; `end` is exclusive, negative wraps, caller passes cap for None */
double pt_reduce(const double *v, int64_t cap, int kind, int64_t start, int64_t end) {
    if (end < 0) end += cap;
    return walk(v, kind, start, end - 1, 1, 0, cap - 1);
}

/* segment_tree.py:82-97 */
int64_t pt_find_prefixsum_idx(const double *v, int64_t cap, double s) {
    int64_t node = 1;
    while (node < cap) {
        double left = v[2 * node];
        if (left > s) {
            node = 2 * node;
        } else {
            s -= left;
            node = 2 * node + 1;
        }
    }
    return node - cap;
}

/* memory.py:266-285 + 295-309: stratified sampling and IS weights for one batch.
 * u[B] are the unit uniforms the reference draws with random.random(). */
void pt_per_sample(const double *sumv, const double *minv, int64_t cap, int64_t num_entries,
                   const double *u, int64_t B, double beta, int64_t *idxs, double *iws) {
    double p_total = pt_reduce(sumv, cap, 0, 0, num_entries - 1);
    double piece = p_total / (double)B;
    for (int64_t i = 0; i < B; ++i) {
        double s = (u[i] * piece) + ((double)i * piece);
        idxs[i] = pt_find_prefixsum_idx(sumv, cap, s);
    }
    double total = pt_reduce(sumv, cap, 0, 0, num_entries);
    double lowest = pt_reduce(minv, cap, 1, 0, num_entries) / total;
    double max_w = pow((double)num_entries * lowest, -beta);
    for (int64_t i = 0; i < B; ++i) {
        double prob = sumv[cap + idxs[i]] / total;
        iws[i] = pow((double)num_entries * prob, -beta) / max_w;
    }
}

/* memory.py:369-379: sequential, last writer wins; returns the updated max_priority. */
double pt_per_update(double *sumv, double *minv, int64_t cap, const int64_t *idxs,
                     const double *prios, int64_t B, double alpha, int64_t num_demos,
                     double demos_eps, double max_priority) {
    for (int64_t i = 0; i < B; ++i) {
        double p = prios[i];
        if (idxs[i] < num_demos) p += demos_eps;
        double leaf = pow(p, alpha);
        pt_set(sumv, cap, 0, idxs[i], leaf);
        pt_set(minv, cap, 1, idxs[i], leaf);
        if (p > max_priority) max_priority = p; /* max(priority, self.max_priority) */
    }
    return max_priority;
}

/* memory.py:337-342 for a run of n consecutive fresh indices (bulk prefill of a large buffer) */
void pt_fill_range(double *sumv, double *minv, int64_t cap, int64_t first, int64_t n, double leaf) {
    for (int64_t i = 0; i < n; ++i) {
        pt_set(sumv, cap, 0, first + i, leaf);
        pt_set(minv, cap, 1, first + i, leaf);
    }
}
