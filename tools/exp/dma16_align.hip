// Does global_load_lds_dwordx4 accept a 4-byte-aligned (not 16-byte-aligned) global source on gfx950?
// Each lane copies 16 bytes from src + off + lane*4 floats ... prints mismatches per offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) float lds_float;
typedef const __attribute__((address_space(1))) float glb_float;
__global__ void k(const float* src, int off, float* out) {
    __shared__ float S[256];
    const int lane = threadIdx.x;
    __builtin_amdgcn_global_load_lds((glb_float*)(src + off + lane * 4), (lds_float*)S, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = S[i];
}
int main() {
    float h[1024], *d, *o, r[256];
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 1024);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int off = 0; off < 8; ++off) {
        hipMemset(o, 0, 1024);
        k<<<1, 64>>>(d, off, o);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != (float)(off + i);
        printf("off %d floats: err=%d mismatches=%d (r[0..4]= %g %g %g %g %g)\n", off, (int)e, bad, r[0], r[1], r[2], r[3], r[4]);
    }
    return 0;
}
