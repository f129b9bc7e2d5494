// pow(x, y) with the bits of the host C library's double-precision pow, callable on the device.
//
// Why: prioritized replay computes leaf values `priority ** alpha` and importance weights
// `(n * prob) ** -beta` with the host pow (memory.py:301,308,341-342,376-377); the sampled indices of every
// later step depend on those bits.  The device math library's pow differs in the last place, which forced
// a host round trip into every prioritized update row.  glibc's pow (2.28 and later; this image: 2.35) is
// the table-driven log/exp2 routine; its constants are read out of the image's libm by
// tools/extract_pow_tables.py (pow_tables.h), and the operation order below - including which
// products the compiler of that libm fused - follows the FMA/AVX2 build of the routine (the variant the
// resolver picks on every CPU with both, read from the disassembly of this image's libm).  Here every fma
// is explicit and nothing else may contract.
//
// Proof is by test, not by reading: tools/exp/pow_check.cpp (CPU build of this header against libm's pow) and
// tests/test_gpu_pow.py (device against libm) compare >= 2^26 bases per exponent bit for bit.
// Domain of the bit-for-bit claim: x positive and normal, 2^-65 <= |y| < 2^63, no overflow/underflow
// (|y log x| < 512).  Outside it the routine defers to the platform pow (and the claim does not apply).
#pragma once
#include <stdint.h>
#include <math.h>
#include "pow_tables.h"

#if defined(__HIPCC__)
#define SAMX_POW_HD __host__ __device__ __forceinline__
#else
#define SAMX_POW_HD static inline
#endif

namespace samx_pow {

#define SAMX_POW_HOST_D(name, n, ...) static const double name##_h[n] = __VA_ARGS__;
#define SAMX_POW_HOST_U(name, n, ...) static const uint64_t name##_h[n] = __VA_ARGS__;
SAMX_POW_TABLES(SAMX_POW_HOST_D, SAMX_POW_HOST_U)
#if defined(__HIPCC__)
#define SAMX_POW_DEV_D(name, n, ...) static __device__ const double name##_d[n] = __VA_ARGS__;
#define SAMX_POW_DEV_U(name, n, ...) static __device__ const uint64_t name##_d[n] = __VA_ARGS__;
SAMX_POW_TABLES(SAMX_POW_DEV_D, SAMX_POW_DEV_U)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SAMX_POW_T(name) name##_d
#else
#define SAMX_POW_T(name) name##_h
#endif

SAMX_POW_HD uint64_t bits_of(double x) { union { double d; uint64_t u; } c; c.d = x; return c.u; }
SAMX_POW_HD double dbl_of(uint64_t u) { union { double d; uint64_t u; } c; c.u = u; return c.d; }
SAMX_POW_HD double mul(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dmul_rn(a, b);
#else
    volatile double r = a * b; return r;
#endif
}
SAMX_POW_HD double add(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __dadd_rn(a, b);
#else
    volatile double r = a + b; return r;
#endif
}
SAMX_POW_HD double fma_(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fma_rn(a, b, c);
#else
    return __builtin_fma(a, b, c);
#endif
}

// log(x) as hi + *tail, x given by its bits (positive, normal)
SAMX_POW_HD double log_parts(uint64_t ix, double* tail) {
    const double* T = SAMX_POW_T(kPowLogTab);
    const double* A = SAMX_POW_T(kPowLogPoly);
    const double ln2hi = SAMX_POW_T(kPowLn2)[0], ln2lo = SAMX_POW_T(kPowLn2)[1];
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> (52 - 7)) % 128);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double z = dbl_of(iz), kd = (double)k;
    const double invc = T[3 * i], logc = T[3 * i + 1], logctail = T[3 * i + 2];
    const double r = fma_(z, invc, -1.0);                       // exact
    const double t1 = fma_(kd, ln2hi, logc);
    const double t2 = add(t1, r);
    const double lo1 = fma_(kd, ln2lo, logctail);
    const double lo2 = add(add(t1, -t2), r);
    const double ar = mul(A[0], r);
    const double ar2 = mul(r, ar);
    const double ar3 = mul(r, ar2);
    const double hi = add(t2, ar2);
    const double lo3 = fma_(ar, r, -ar2);
    const double lo4 = add(add(t2, -hi), ar2);
    const double q1 = fma_(r, A[2], A[1]), q3 = fma_(r, A[4], A[3]), q5 = fma_(r, A[6], A[5]);
    const double pp = fma_(ar2, fma_(q5, ar2, q3), q1);
    const double lo = fma_(ar3, pp, add(add(add(lo1, lo2), lo3), lo4));
    const double y = add(hi, lo);
    *tail = add(add(hi, -y), lo);
    return y;
}

// exp(x + xtail) for 2^-54 <= |x| < 512 (the caller checks)
SAMX_POW_HD double exp_parts(double x, double xtail) {
    const double* E = SAMX_POW_T(kPowExpC);
    const uint64_t* T = SAMX_POW_T(kPowExpTab);
    double kd = fma_(E[0], x, E[1]);
    const uint64_t ki = bits_of(kd);
    kd = add(kd, -E[1]);
    double r = fma_(kd, E[3], fma_(kd, E[2], x));
    r = add(r, xtail);
    const uint64_t idx = 2 * (ki % 128);
    const uint64_t top = ki << (52 - 7);
    const double tail = dbl_of(T[idx]);
    const uint64_t sbits = T[idx + 1] + top;
    const double r2 = mul(r, r);
    const double tmp = fma_(fma_(r, E[7], E[6]), mul(r2, r2), fma_(fma_(r, E[5], E[4]), r2, add(r, tail)));
    const double scale = dbl_of(sbits);
    return fma_(tmp, scale, scale);
}

SAMX_POW_HD bool in_domain(double x, double y) {
    const uint32_t topx = (uint32_t)(bits_of(x) >> 52), topy = (uint32_t)(bits_of(y) >> 52);
    return !(topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu);
}

SAMX_POW_HD double pow_bits(double x, double y) {
    if (!in_domain(x, y)) return pow(x, y);
    double lo;
    const double hi = log_parts(bits_of(x), &lo);
    const double ehi = mul(y, hi);
    const double elo = fma_(y, lo, fma_(y, hi, -ehi));
    const uint32_t abstop = (uint32_t)(bits_of(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
        if (abstop - 0x3c9u >= 0x80000000u) return add(1.0, ehi);   // |y log x| < 2^-54
        return pow(x, y);                                          // overflow / underflow range
    }
    return exp_parts(ehi, elo);
}

}  // namespace samx_pow
