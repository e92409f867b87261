// What v_permlane32_swap / v_permlane16_swap (gfx950) do to a pair of registers holding the same value -- the cross-row step of a
// 64-lane max over lanes {j, j+16, j+32, j+48} without LDS.  (The clang builtin mis-assigns its second result when both operands
// are copies of one value -- ROCm 7.2 -- hence inline asm.)   hipcc --offload-arch=gfx950 -O3 permlane_swap_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float max_xor32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ inline float max_xor16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__global__ void k(float *o, const float *x) {
    const float v = x[threadIdx.x];
    o[threadIdx.x] = max_xor32(v);
    o[64 + threadIdx.x] = max_xor16(v);
    o[128 + threadIdx.x] = max_xor16(max_xor32(v));
}
int main() {
    float h[64], r[192], *dx, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, sizeof(r));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dout, dx);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    int bad32 = 0, bad16 = 0, bad = 0;
    for (int i = 0; i < 64; ++i) {
        bad32 += r[i] != fmaxf(h[i], h[i ^ 32]);
        bad16 += r[64 + i] != fmaxf(h[i], h[i ^ 16]);
        const int j = i & 15;
        bad += r[128 + i] != fmaxf(fmaxf(h[j], h[j + 16]), fmaxf(h[j + 32], h[j + 48]));
    }
    printf("permlane32_swap as xor-32 max: %d wrong; permlane16_swap as xor-16 max: %d wrong; both = max over the 4 rows: %d wrong\n", bad32, bad16, bad);
    return bad32 + bad16 + bad ? 1 : 0;
}
