// Issue rate of the fp32 MFMAs: one wave per SIMD (or two), 8 independent accumulators, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/exp/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x16 acc[8];
    f32x4 acc4[8];
    for (int i = 0; i < 8; ++i) { for (int v = 0; v < 16; ++v) acc[i][v] = 0.f; for (int v = 0; v < 4; ++v) acc4[i][v] = 0.f; }
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) { for (int v = 0; v < 16; ++v) s += acc[i][v]; for (int v = 0; v < 4; ++v) s += acc4[i][v]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int blocks, int threads, double flops_per) {
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&clk, sizeof(long long) * blocks);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, threads>>>(out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)iters * 32;   // MFMAs per wave
    printf("%s, %d workgroups x %d waves: %.1f us, %.1f shader clocks per MFMA (clock64), %.1f ns per MFMA, %.1f TFLOP/s\n", name, blocks,
           threads / 64, ms * 1e3, c / n, ms * 1e6 / n, n * flops_per * blocks * (threads / 64) / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}
int main() {
    run<0>("v_mfma_f32_32x32x2_f32", 256, 256, 4096.0);
    run<0>("v_mfma_f32_32x32x2_f32", 256, 512, 4096.0);
    run<0>("v_mfma_f32_32x32x2_f32", 1, 256, 4096.0);
    run<1>("v_mfma_f32_16x16x4_f32", 256, 256, 2048.0);
    run<1>("v_mfma_f32_16x16x4_f32", 1, 256, 2048.0);
    return 0;
}
