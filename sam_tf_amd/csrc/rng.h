// Counter-based uniforms for the tensors TensorFlow draws inside the discriminator graph
// (tf.random_uniform in discriminator.py:101-109,130-133).  Philox4x32-10 keyed by the run seed; the
// counter is (element index / 4, stream id, update count) and element e takes output word e % 4, so every
// element of every update has its own draw, identical no matter which kernel (fused or layer-wise) or which
// thread produces it — and a kernel that needs a contiguous range of elements pays one Philox per four (the
// 256-wide role kernel spent 11 of its 73 us drawing one element per Philox, on half of its waves).
// TF's own stream cannot be reproduced (and parity tests inject these tensors instead).
#pragma once
#include <stdint.h>

namespace samx {

__host__ __device__ inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

enum RngStream { RNG_EPS_OB = 0, RNG_EPS_AC = 1, RNG_REAL = 2, RNG_FAKE = 3 };

// the uniforms in [0, 1) (24 random bits each) of elements 4 group .. 4 group + 3
__host__ __device__ inline void philox_uniform4(uint64_t seed, uint64_t step, uint32_t stream, uint64_t group,
                                                float u[4]) {
    uint32_t c[4] = {(uint32_t)group, (uint32_t)(group >> 32) ^ (stream << 28), (uint32_t)step,
                     (uint32_t)(step >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}
// one element
__host__ __device__ inline float philox_uniform(uint64_t seed, uint64_t step, uint32_t stream,
                                                uint64_t elem) {
    float u[4];
    philox_uniform4(seed, step, stream, elem >> 2, u);
    const int w = (int)(elem & 3);
    return w == 0 ? u[0] : (w == 1 ? u[1] : (w == 2 ? u[2] : u[3]));
}

// smoothed labels: tf.random_uniform(minval=1.2, maxval=0.7) = 1.2 - 0.5u ; (0.3, 0.0) = 0.3 - 0.3u
__host__ __device__ inline float real_label(float u) { return 1.2f - 0.5f * u; }
__host__ __device__ inline float fake_label(float u) { return 0.3f - 0.3f * u; }

}  // namespace samx
