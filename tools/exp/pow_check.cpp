// CPU check of sam_tf_amd/csrc/pow_bits.h against the image's libm pow, bit for bit.
//   g++ -O2 -ffp-contract=off -fopenmp -I sam_tf_amd/csrc tools/exp/pow_check.cpp -o /tmp/pow_check -lm && /tmp/pow_check
// (1) every float32 base in [1e-6, 1e4] widened to double (the priorities |td| + 1e-6f), alone and with the
//     demonstration bonus added, for alpha in {0.3, 0.6, 0.7, 1.0-2^-53};  (2) 2^27 random double bases in
//     [2^-40, 2^40] with exponents -beta (importance weights), beta in {0.4, 0.5, 1.0, 0.123456789}.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <initializer_list>
#include "pow_bits.h"
static inline uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
    float lo = 1e-6f, hi = 1e4f;
    uint32_t blo, bhi; memcpy(&blo, &lo, 4); memcpy(&bhi, &hi, 4);
    const double alphas[] = {0.3, 0.6, 0.7, 0.9999999999999999};
    const double bonus[] = {0.0, 0.1};
    long long total = 0, bad = 0;
    for (double a : alphas) for (double e : bonus) {
        long long nb = 0;
#pragma omp parallel for reduction(+:nb) schedule(static)
        for (int64_t b = blo; b <= (int64_t)bhi; ++b) {
            uint32_t u = (uint32_t)b; float f; memcpy(&f, &u, 4);
            double x = (double)f; if (e != 0.0) x += e;
            if (bits(samx_pow::pow_bits(x, a)) != bits(pow(x, a))) ++nb;
        }
        printf("alpha %.17g bonus %.1f: %lld bases, %lld mismatches\n", a, e, (long long)(bhi - blo + 1), nb);
        total += bhi - blo + 1; bad += nb;
    }
    const double betas[] = {0.4, 0.5, 1.0, 0.123456789};
    for (double be : betas) {
        long long nb = 0;
        const int64_t N = 1ll << 27;
#pragma omp parallel for reduction(+:nb) schedule(static)
        for (int64_t c = 0; c < 64; ++c) {
            uint64_t s = 0x9E3779B97F4A7C15ull * (c + 1);
            for (int64_t i = 0; i < N / 64; ++i) {
                uint64_t r = rng(s);
                uint64_t ex = 1023 - 40 + (r >> 52) % 81;
                uint64_t u = (ex << 52) | (rng(s) & ((1ull << 52) - 1));
                double x; memcpy(&x, &u, 8);
                if (bits(samx_pow::pow_bits(x, -be)) != bits(pow(x, -be))) ++nb;
            }
        }
        printf("beta %.17g: %lld random bases, %lld mismatches\n", be, (long long)N, nb);
        total += N; bad += nb;
    }
    // exact one, neighbours of one and of the table switch points
    const double edge[] = {1.0, 0x1.fffffffffffffp-1, 0x1.0000000000001p+0, 2.0, 0.5, 0x1.6955500000000p-1};
    for (double x : edge) for (double y : {0.3, 0.6, -0.4, -1.0}) {
        ++total; if (bits(samx_pow::pow_bits(x, y)) != bits(pow(x, y))) { ++bad; printf("edge mismatch %a ^ %a\n", x, y); }
    }
    printf("total %lld, mismatches %lld\n", total, bad);
    return bad != 0;
}
