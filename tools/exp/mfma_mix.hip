// What sits between the MFMAs of one wave per SIMD and what it costs: 8 independent v_mfma_f32_32x32x2_f32 per
// round plus (1) nothing, (2) 12 fp32 VALU adds, (3) four 64-bit multiply-adds, (4) two global loads that hit in L2
// with a counted wait 16 rounds later, (5) all of it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_mix tools/exp/mfma_mix.hip && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, const float* src, int lda, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    float va[12];
    for (int i = 0; i < 12; ++i) va[i] = a * i;
    long long ia = threadIdx.x, ib = lda;
    f32x4u ring[16];
    const float* p = src + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096 + (blockIdx.x & 31) * 65536;
    for (int i = 0; i < 16; ++i) ring[i] = *(const f32x4u*)(p + i * lda);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            __builtin_amdgcn_sched_barrier(0);
            float fa = a, fb = b;
            if (KIND == 4 || KIND == 5) { fa = ring[u][0] + ring[u][3]; fb = ring[u][1] + ring[u][2]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
            if (KIND == 2 || KIND == 5) {
#pragma unroll
                for (int i = 0; i < 12; ++i) va[i] += b;
            }
            if (KIND == 3 || KIND == 5) {
                ia = ia * lda + ib; ib = ib * lda + ia; ia = ia * lda + ib; ib = ib * lda + ia;
            }
            if (KIND == 4 || KIND == 5) ring[u] = *(const f32x4u*)(p + ((it * 16 + u) & 255) * lda);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
    for (int i = 0; i < 12; ++i) s += va[i];
    for (int i = 0; i < 16; ++i) s += ring[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(ia + ib);
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, const float* src) {
    const int blocks = 256, threads = 256, iters = 200;
    float* out; long long* clk;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads); (void)hipMalloc(&clk, sizeof(long long) * blocks);
    k<KIND><<<blocks, threads>>>(out, clk, src, 393, iters);
    (void)hipDeviceSynchronize();
    k<KIND><<<blocks, threads>>>(out, clk, src, 393, iters);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-60s %.1f shader clocks per MFMA\n", name, c / ((double)iters * 128));
    (void)hipFree(out); (void)hipFree(clk);
}
int main() {
    float* src; (void)hipMalloc(&src, 64u << 20); (void)hipMemset(src, 0, 64u << 20);
    run<1>("8 MFMAs per round, nothing else", src);
    run<2>("+ 12 fp32 adds", src);
    run<3>("+ four 64-bit multiply-adds", src);
    run<4>("+ one 16-byte global load (row stride 393 floats), used 16 rounds later", src);
    run<5>("+ all of it", src);
    return 0;
}
