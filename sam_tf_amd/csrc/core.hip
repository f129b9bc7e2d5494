// Error reporting and version for libsamx.
#include "samx_common.h"
#include "../../include/samx.h"
#include <stdarg.h>

namespace samx {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace samx

extern "C" {

const char* samx_last_error(void) { return samx::err_buf(); }
int samx_version(void) { return 100; }

int samx_stream_synchronize(void* stream) {
    SAMX_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SAMX_OK;
}

int64_t samx_mlp_param_count(const samx_mlp* d) {
    if (!d) return -1;
    int64_t ln = d->layernorm ? 2 * (int64_t)(d->h1 + d->h2) : 0;
    return (int64_t)d->in_dim * d->h1 + d->h1 + (int64_t)d->h1 * d->h2 + d->h2 +
           (int64_t)d->h2 * d->out_dim + d->out_dim + ln;
}

// ---- the replay index draws of a multi-row launch, on the HOST (memory.py:90: np.random.randint(low=0, high=N, size=B))
// numpy's legacy generator, restated: MT19937 (numpy/random/src/mt19937/mt19937.h: mt19937_gen + the four tempering
// steps of mt19937_next) feeding random_bounded_uint64_fill's 32-bit masked-rejection branch
// (numpy/random/src/distributions/distributions.c: `while ((val = next_uint32 & mask) > rng);`, mask = gen_mask(rng),
// no buffering for 32-bit values; rng == 0xFFFFFFFF takes the raw word, rng == 0 draws nothing), which is what
// RandomState.randint runs for an int64 request whose range fits 32 bits (_bounded_integers._rand_int64, masked).
// It works IN PLACE on the generator's own state (bit_generator.ctypes.state_address: uint32 key[624]; int pos), so
// numpy's stream continues behind it exactly as behind its own call — at ~1.5 ns per draw instead of ~6.
namespace {
struct MtState { uint32_t key[624]; int pos; };
inline void mt_gen(uint32_t* k) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int i = 0;
    for (; i < N - M; ++i) { const uint32_t y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + M] ^ (y >> 1) ^ (-(y & 1) & A); }
    for (; i < N - 1; ++i) { const uint32_t y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + (M - N)] ^ (y >> 1) ^ (-(y & 1) & A); }
    const uint32_t y = (k[N - 1] & UP) | (k[0] & LO);
    k[N - 1] = k[M - 1] ^ (y >> 1) ^ (-(y & 1) & A);
}
// n accepted draws into o[0, n): the words a block of 624 still holds are tempered and masked in one branch-free
// pass (a rejected value is written and then overwritten by the next one: never beyond o[n - 1])
inline void mt_fill(MtState* st, uint32_t rng, uint32_t mask, int64_t n, int64_t* __restrict o) {
    int pos = st->pos;
    uint32_t* key = st->key;
    for (int64_t c = 0; c < n;) {
        if (pos == 624) { mt_gen(key); pos = 0; }
        int64_t take = 624 - pos;
        if (take > n - c) take = n - c;
        int64_t got = 0;
        for (int64_t j = 0; j < take; ++j) {
            uint32_t y = key[pos + j];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            y &= mask;
            o[c + got] = (int64_t)y;
            got += (y <= rng);
        }
        pos += (int)take;
        c += got;
    }
    st->pos = pos;
}
}  // namespace

int samx_host_mt19937_randint(void* mt_state, uint64_t high, int64_t rows, int64_t row_len, int64_t* out,
                              int64_t row_stride) {
    SAMX_REQUIRE(mt_state && out && high >= 1 && high <= 0x100000000ull && rows >= 0 && row_len >= 0 &&
                     row_stride >= row_len,
                 "samx_host_mt19937_randint: bad arguments (1 <= high <= 2^32, row_stride >= row_len)");
    MtState* st = (MtState*)mt_state;
    SAMX_REQUIRE(st->pos >= 0 && st->pos <= 624, "samx_host_mt19937_randint: not an MT19937 state (pos %d)", st->pos);
    const uint64_t rng = high - 1;
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t r = 0; r < rows; ++r) {
        int64_t* o = out + r * row_stride;
        if (rng == 0)   // (numpy draws nothing for a one-value range)
            for (int64_t c = 0; c < row_len; ++c) o[c] = 0;
        else            // (rng == 2^32 - 1: mask == rng, every word is accepted — numpy's raw-word branch)
            mt_fill(st, (uint32_t)rng, (uint32_t)mask, row_len, o);
    }
    return SAMX_OK;
}

}  // extern "C"
