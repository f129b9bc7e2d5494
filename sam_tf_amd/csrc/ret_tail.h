// Return statistics + PopArt (sam_agent.py:554-590,662-672) as a device routine: the body of ret_popart_kernel.
#pragma once
#include <stdint.h>

namespace samx {

// phase 0: everything (one process).  N processes (mpi_running_mean_std.py:47-59 all-reduces the batch sums): phase 1
// = the batch sums only -> addvec[6] = [sum, sumsq, n | the same for the n-step targets]; phase 2 = commit the
// (all-reduced) addvec and rescale.  Called by every thread of a workgroup of 256 or more threads; the first 256 do
// the work (so that the order of the additions — the bits — does not depend on the caller's size).
__device__ inline void ret_popart_body(const float* __restrict__ tq, const float* __restrict__ tqn, int64_t B,
                                       double* __restrict__ sums, float* mean, float* stdv, int popart, float* w3a,
                                       float* b3a, float* w3b, float* b3b, int phase, double* __restrict__ addvec,
                                       int h2) {
    __shared__ double sh[4][4];
    __shared__ float shf[4];
    const int tid = threadIdx.x;
    const bool worker = tid < 256;
    // everything the tail needs that does not depend on the batch is requested NOW (old statistics, running sums,
    // the heads to rescale): behind the reduction each of these was a bare round trip in a one-workgroup kernel
    float old_m = 0.f, old_s = 1.f, wa = 0.f, wb = 0.f, ba = 0.f, bb = 0.f;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (phase != 1 && worker) {
        if (tid == 0) { old_m = mean[0]; old_s = stdv[0]; s0 = sums[0]; s1 = sums[1]; s2 = sums[2]; }
        if (popart) {
            if (tid < h2) { wa = w3a[tid]; wb = w3b[tid]; }
            if (tid == 0) { ba = b3a[0]; bb = b3b[0]; }
        }
    }
    double a1 = 0.0, q1 = 0.0, a2 = 0.0, q2 = 0.0;
    if (phase != 2 && worker)
        for (int64_t i = tid; i < B; i += 256) {
            const double x = (double)tq[i];
            a1 = __dadd_rn(a1, x); q1 = __dadd_rn(q1, __dmul_rn(x, x));
            if (tqn) {
                const double y = (double)tqn[i];
                a2 = __dadd_rn(a2, y); q2 = __dadd_rn(q2, __dmul_rn(y, y));
            }
        }
    // fixed order: butterflies inside each wave, then the four waves in turn (one barrier; the shared-memory tree
    // this replaces took eight)
    for (int o = 32; o > 0; o >>= 1) {
        a1 = __dadd_rn(a1, __shfl_xor(a1, o, 64)); q1 = __dadd_rn(q1, __shfl_xor(q1, o, 64));
        a2 = __dadd_rn(a2, __shfl_xor(a2, o, 64)); q2 = __dadd_rn(q2, __shfl_xor(q2, o, 64));
    }
    if (worker && (tid & 63) == 0) { sh[0][tid >> 6] = a1; sh[1][tid >> 6] = q1; sh[2][tid >> 6] = a2; sh[3][tid >> 6] = q2; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            sh[k][0] = __dadd_rn(__dadd_rn(sh[k][0], sh[k][1]), __dadd_rn(sh[k][2], sh[k][3]));
    }
    if (phase == 1) {
        if (tid == 0) {
            addvec[0] = sh[0][0]; addvec[1] = sh[1][0]; addvec[2] = (double)B;
            addvec[3] = sh[2][0]; addvec[4] = sh[3][0]; addvec[5] = tqn ? (double)B : 0.0;
        }
        return;
    }
    if (tid == 0) {
        shf[0] = old_m;
        shf[1] = old_s;
        double t0 = sh[0][0], t1 = sh[1][0], t2 = (double)B, u0 = sh[2][0], u1 = sh[3][0], u2 = (double)B;
        if (phase == 2) { t0 = addvec[0]; t1 = addvec[1]; t2 = addvec[2]; u0 = addvec[3]; u1 = addvec[4]; u2 = addvec[5]; }
        s0 = __dadd_rn(s0, t0); s1 = __dadd_rn(s1, t1); s2 = __dadd_rn(s2, t2);
        if (tqn) { s0 = __dadd_rn(s0, u0); s1 = __dadd_rn(s1, u1); s2 = __dadd_rn(s2, u2); }
        sums[0] = s0; sums[1] = s1; sums[2] = s2;
        const float m = (float)(s0 / s2), msq = (float)(s1 / s2);
        const float sd = sqrtf(fmaxf(__fsub_rn(msq, __fmul_rn(m, m)), 1e-2f));
        mean[0] = m;
        stdv[0] = sd;
        shf[2] = m;
        shf[3] = sd;
    }
    __syncthreads();
    if (!popart || !worker) return;
    const float om = shf[0], os = shf[1], nm = shf[2], ns = shf[3];
    for (int c = tid; c < h2; c += 256) {   // (the first 256 were requested at the top)
        const float va = c < 256 ? wa : w3a[c], vb = c < 256 ? wb : w3b[c];
        w3a[c] = __fdiv_rn(__fmul_rn(va, os), ns);
        w3b[c] = __fdiv_rn(__fmul_rn(vb, os), ns);
    }
    if (tid == 0) {
        b3a[0] = __fdiv_rn(__fsub_rn(__fadd_rn(__fmul_rn(ba, os), om), nm), ns);
        b3b[0] = __fdiv_rn(__fsub_rn(__fadd_rn(__fmul_rn(bb, os), om), nm), ns);
    }
}

// (Measured and dropped: this routine as the TAIL of the target launch — every workgroup of that launch ends with a
// device-scope release fence and draws a ticket, the one that draws the last ticket acquires and runs the statistics
// and the rescale, so that the one-workgroup launch is not needed.  Correct (301 parity tests) and slower: launcher
// options at C2 21.50 -> 20.88 k steps/s, at C4 5.41 -> 5.02 k on one box, alternating.  A device-scope fence writes
// back the whole L2 of its XCD while the other workgroups of the launch are still filling it; 16-128 of them cost
// more than the 4-5 us launch they replace.  The same lesson as the optimiser pass inside the reduction launch,
// DESIGN.md 7.)
}  // namespace samx
