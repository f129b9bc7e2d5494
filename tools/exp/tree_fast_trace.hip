// Phase times of the prioritized row head's tree work (tree_dev.h) at cap 2^20, 256 entries: shader-clock
// stamps by thread 0 of workgroup 0.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   -I sam_tf_amd/csrc tools/exp/tree_fast_trace.hip -o tools/exp/tree_fast_trace ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long g_stamp[32];
#define SAMX_TREE_STAMP(i) if (threadIdx.x == 0 && blockIdx.x == 0) g_stamp[i] = (long long)__builtin_readcyclecounter();
#include "tree_dev.h"

__global__ __launch_bounds__(1024) void k_walk(double* s, double* m, int64_t cap, const int64_t* idx, const float* td,
                                              int n, int levels, int G, samx::PerState* st) {
    extern __shared__ double lds[];
    SAMX_TREE_STAMP(7)
    const int64_t sub = cap / G, lo = blockIdx.x * sub;
    samx::tree_walk_sub_body(s, m, cap, idx, td, n, levels, lo, lo + sub, cap, 0.6, 0, 0.0, st,
                             *reinterpret_cast<samx::WalkLds*>(lds));
    SAMX_TREE_STAMP(14)
}

__global__ __launch_bounds__(1024) void k_top(double* s, double* m, int64_t cap, int levels) {
    extern __shared__ double lds[];
    SAMX_TREE_STAMP(20)
    samx::tree_top_body(s, m, cap, levels, lds, lds + samx::kFastTop, blockIdx.x == 0);
    SAMX_TREE_STAMP(21)
}

int main() {
    const int64_t cap = 1 << 20;
    const int n = 256, levels = 20, G = 16;
    double *s, *m; float* td; int64_t* idx; samx::PerState* st;
    hipMalloc(&s, 2 * cap * 8); hipMalloc(&m, 2 * cap * 8); hipMalloc(&td, n * 4); hipMalloc(&idx, n * 8);
    hipMalloc(&st, 16); hipMemset(st, 0, 16);
    std::vector<double> z(2 * cap, 0.0), inf(2 * cap, 1.0 / 0.0);
    hipMemcpy(s, z.data(), 2 * cap * 8, hipMemcpyHostToDevice);
    hipMemcpy(m, inf.data(), 2 * cap * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_top, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * samx::kFastTop * 8);
    std::vector<int64_t> hi(n); std::vector<float> hl(n);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int r = 0; r < 6; ++r) {
        for (int i = 0; i < n; ++i) { hi[i] = rand() % (cap - 7); hl[i] = 0.1f + (rand() % 1000) * 1e-3f; }
        hipMemcpy(idx, hi.data(), n * 8, hipMemcpyHostToDevice);
        hipMemcpy(td, hl.data(), n * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        k_walk<<<G, 1024, sizeof(samx::WalkLds)>>>(s, m, cap, idx, td, n, levels, G, st);
        hipEventRecord(e1);
        k_top<<<8, 1024, 2 * samx::kFastTop * 8>>>(s, m, cap, levels);
        hipEventRecord(e2);
        hipDeviceSynchronize();
        float ms1, ms2; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms2, e1, e2);
        long long t[32];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_stamp), sizeof(t));
        printf("run %d: walk %.1f us top %.1f us; walk cycles:", r, ms1 * 1e3, ms2 * 1e3);
        for (int i = 8; i <= 14; ++i) printf(" [%d] %lld", i, t[i] - t[i - 1]);
        printf(" ; top %lld ; level 2: insert %lld barA %lld lookup %lld barB %lld\n", t[21] - t[20], t[15] - t[19], t[16] - t[15],
               t[17] - t[16], t[18] - t[17]);
    }
    return 0;
}
