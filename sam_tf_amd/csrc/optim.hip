// Global-norm clip + TF1 Adam + Polyak over flat parameter vectors.
//   helpers/mpi_adam.py:18-36 (+ tf.train.AdamOptimizer dense update), helpers/tf_util.py:137-151,
//   imitation_algorithms/sam_agent.py:20-37.
// HBM-bound streaming: 28 bytes/parameter for Adam (+12 with the fused Polyak update).
#include "samx_common.h"
#include "rng.h"
#include <math.h>
#include "../../include/samx.h"
#include <math.h>

namespace {

constexpr int kPartials = 256;  // max blocks of the sum-of-squares pass
constexpr int kThreads = 256;

__device__ __forceinline__ double block_sum(double x, double* sh) {
    // fixed-order tree reduction -> deterministic
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = x;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;  // valid on thread 0
}

__global__ __launch_bounds__(kThreads) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                         float* __restrict__ partials,
                                                         int64_t* __restrict__ step_dev) {
    __shared__ double sh[kThreads / 64];
    // the device-side update counter ticks here, one launch ahead of the kernel that reads it
    if (step_dev && blockIdx.x == 0 && threadIdx.x == 0) step_dev[0] += 1;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        float x = g[i];
        acc += (double)x * (double)x;
    }
    double t = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = (float)t;
}

__device__ __forceinline__ float norm_from_partials(const float* __restrict__ partials, int np,
                                                    double* sh, float* shn) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += kThreads) acc += (double)partials[i];
    double t = block_sum(acc, sh);
    if (threadIdx.x == 0) *shn = (float)sqrt(t);
    __syncthreads();
    return *shn;
}

__global__ __launch_bounds__(kThreads) void adam_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    int64_t n, float lr_t, float beta1, float beta2, float eps, float clip_norm, float grad_scale,
    float* __restrict__ targ, float tau, const float* __restrict__ partials, int np,
    const int64_t* __restrict__ step_dev, float lr) {
    __shared__ double sh[kThreads / 64];
    __shared__ float shn;
    __shared__ float sh_lr;
    if (step_dev) {
        // lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), t read from HBM so that a captured graph
        // needs no per-step host argument
        if (threadIdx.x == 0) {
            const double t = (double)step_dev[0];
            sh_lr = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
        }
        __syncthreads();
        lr_t = sh_lr;
    }
    float scale = grad_scale;
    if (clip_norm > 0.f) {
        // partials hold |sum-over-workers grad|^2 ; the mean gradient's norm is grad_scale * that
        const float nrm = norm_from_partials(partials, np, sh, &shn) * grad_scale;
        scale = grad_scale * (clip_norm / fmaxf(nrm, clip_norm));
    }
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads) {
        const float gi = g[i] * scale;
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * (1.f - beta1);
        vi = vi + (gi * gi - vi) * (1.f - beta2);
        const float pi = p[i] - (mi * lr_t) / (sqrtf(vi) + eps);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
        if (targ) targ[i] = (1.f - tau) * targ[i] + tau * pi;
    }
}

__global__ __launch_bounds__(kThreads) void polyak_kernel(float* __restrict__ targ,
                                                          const float* __restrict__ p, int64_t n,
                                                          float tau) {
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads)
        targ[i] = (1.f - tau) * targ[i] + tau * p[i];
}

__global__ __launch_bounds__(kThreads) void clip_copy_kernel(const float* __restrict__ g, int64_t n,
                                                             float clip_norm, float* __restrict__ out,
                                                             const float* __restrict__ partials,
                                                             int np) {
    __shared__ double sh[kThreads / 64];
    __shared__ float shn;
    float scale = 1.f;
    if (clip_norm > 0.f) {
        const float nrm = norm_from_partials(partials, np, sh, &shn);
        scale = clip_norm / fmaxf(nrm, clip_norm);
    }
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads)
        out[i] = g[i] * scale;
}

__global__ __launch_bounds__(kThreads) void mismatch_kernel(const uint32_t* __restrict__ a,
                                                            const uint32_t* __restrict__ b, int64_t n,
                                                            int32_t* __restrict__ out) {
    int c = 0;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * kThreads)
        c += (a[i] != b[i]);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int grid_for(int64_t n) {
    int64_t b = (n + (int64_t)kThreads * 4 - 1) / ((int64_t)kThreads * 4);
    if (b < 1) b = 1;
    if (b > kPartials) b = kPartials;
    return (int)b;
}

}  // namespace

namespace {
// Parameter-space noise (sam_agent.py:40-58): out = param + N(0, std^2) on the perturbable variables
// (dense kernels and biases), out = param on the LayerNorm variables (networks.py:325-334).
// Normal draws: Box-Muller over two Philox uniforms keyed by (seed, counter, element); TF's own
// tf.random_normal stream cannot be reproduced.
struct NoiseRanges {
    int64_t lo[2], hi[2];  // index ranges of the non-perturbable (LayerNorm) variables
};
__global__ __launch_bounds__(256) void perturb_kernel(const float* __restrict__ p, float* __restrict__ out,
                                                      int64_t n, float stdv, uint64_t seed, uint64_t counter,
                                                      NoiseRanges r) {
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const bool frozen = (i >= r.lo[0] && i < r.hi[0]) || (i >= r.lo[1] && i < r.hi[1]);
        float v = p[i];
        if (!frozen) {
            const float u1 = samx::philox_uniform(seed, counter, 4, (uint64_t)i);
            const float u2 = samx::philox_uniform(seed, counter, 5, (uint64_t)i);
            const float z = sqrtf(-2.f * logf(1.f - u1)) * cosf(6.28318530717958647692f * u2);  // 1-u1 in (0,1]
            v += stdv * z;
        }
        out[i] = v;
    }
}

// sqrt(mean((a - b)^2)) (sam_agent.py:371), one workgroup, float64 accumulation in fixed order
__global__ __launch_bounds__(1024) void rms_distance_kernel(const float* __restrict__ a,
                                                            const float* __restrict__ b, int64_t n,
                                                            float* __restrict__ out) {
    __shared__ double sh[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double d = (double)a[i] - (double)b[i];
        acc += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += sh[i];
        out[0] = (float)sqrt(t / (double)n);
    }
}
}  // namespace

extern "C" {

int samx_adam_partials(void) { return kPartials; }

int samx_clip_adam(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                   float beta1, float beta2, float eps, int64_t step, float clip_norm,
                   float grad_scale, float* targ, float tau, float* partials, int64_t* step_dev,
                   void* stream) {
    SAMX_REQUIRE(param && grad && m && v, "samx_clip_adam: null pointer");
    SAMX_REQUIRE(n > 0 && (step_dev || step >= 1), "samx_clip_adam: n and step must be positive");
    SAMX_REQUIRE((clip_norm <= 0.f && !step_dev) || partials,
                 "samx_clip_adam: clipping / device step counter need the partials scratch");
    const int g = grid_for(n);
    if (clip_norm > 0.f || step_dev) {
        sumsq_kernel<<<g, kThreads, 0, (hipStream_t)stream>>>(grad, n, partials, step_dev);
        SAMX_LAUNCH_CHECK();
    }
    const double st = step_dev ? 1.0 : (double)step;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, st)) / (1.0 - pow((double)beta1, st));
    adam_kernel<<<g, kThreads, 0, (hipStream_t)stream>>>(param, grad, m, v, n, (float)lr_t, beta1,
                                                        beta2, eps, clip_norm, grad_scale, targ, tau,
                                                        partials, g, step_dev, lr);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_polyak(float* targ, const float* param, int64_t n, float tau, void* stream) {
    SAMX_REQUIRE(targ && param && n > 0, "samx_polyak: bad arguments");
    polyak_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(targ, param, n, tau);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_clip_copy(const float* grad, int64_t n, float clip_norm, float* out, float* partials,
                   void* stream) {
    SAMX_REQUIRE(grad && out && n > 0, "samx_clip_copy: bad arguments");
    SAMX_REQUIRE(clip_norm <= 0.f || partials, "samx_clip_copy: clipping needs the partials scratch");
    const int g = grid_for(n);
    if (clip_norm > 0.f) {
        sumsq_kernel<<<g, kThreads, 0, (hipStream_t)stream>>>(grad, n, partials, nullptr);
        SAMX_LAUNCH_CHECK();
    }
    clip_copy_kernel<<<g, kThreads, 0, (hipStream_t)stream>>>(grad, n, clip_norm, out, partials, g);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_count_mismatch(const float* a, const float* b, int64_t n, int32_t* mismatch_out,
                        void* stream) {
    SAMX_REQUIRE(a && b && mismatch_out && n > 0, "samx_count_mismatch: bad arguments");
    SAMX_CHECK_HIP(hipMemsetAsync(mismatch_out, 0, sizeof(int32_t), (hipStream_t)stream));
    mismatch_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>((const uint32_t*)a,
                                                                      (const uint32_t*)b, n,
                                                                      mismatch_out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}


int samx_perturb_params(const samx_mlp* d, const float* params, float* out, float stddev, uint64_t seed,
                        uint64_t counter, void* stream) {
    SAMX_REQUIRE(d && params && out && stddev >= 0.f, "samx_perturb_params: bad arguments");
    const int64_t n = samx_mlp_param_count(d);
    NoiseRanges r;
    r.lo[0] = r.hi[0] = r.lo[1] = r.hi[1] = 0;
    if (d->layernorm) {  // layout: W1 b1 [beta1 gamma1] W2 b2 [beta2 gamma2] W3 b3
        r.lo[0] = (int64_t)d->in_dim * d->h1 + d->h1;
        r.hi[0] = r.lo[0] + 2 * d->h1;
        r.lo[1] = r.hi[0] + (int64_t)d->h1 * d->h2 + d->h2;
        r.hi[1] = r.lo[1] + 2 * d->h2;
    }
    perturb_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(params, out, n, stddev, seed, counter, r);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

int samx_rms_distance(const float* a, const float* b, int64_t n, float* out, void* stream) {
    SAMX_REQUIRE(a && b && out && n > 0, "samx_rms_distance: bad arguments");
    rms_distance_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a, b, n, out);
    SAMX_LAUNCH_CHECK();
    return SAMX_OK;
}

}  // extern "C"
