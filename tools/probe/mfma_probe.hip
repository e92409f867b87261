// Micro-probe: how do f32 MFMAs overlap with VALU / LDS / VMEM work issued by the SAME wave?
// One wave per SIMD (grid = 256 CUs x 1 WG of 256 threads); cycles per MFMA from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV, int NACC>
__global__ __launch_bounds__(256) void probe16(float *out, const float *in, long long *cyc, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = in[tid], b = in[tid + 256];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in[tid + 512 + i];
    const float *lp = lds + (tid & 63) * 4;
    const float *gp = in + tid;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float av = a, bv = b;
            if (MODE == 1) av = lp[u * 64];            // ds_read feeding A directly
            if (MODE == 2) bv = gp[(u + it * 16) & 1023];  // global load feeding B directly (L1/L2 hit)
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k % 8] = v[k % 8] * 1.0001f + 0.5f;  // independent VALU (fp-contract off: 2 ops)
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV>
__global__ __launch_bounds__(256) void probe32(float *out, const float *in, long long *cyc, int iters) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    const int tid = threadIdx.x;
    float a = in[tid], b = in[tid + 256];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in[tid + 512 + i];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k % 8] = v[k % 8] * 1.0001f + 0.5f;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char *name, K kern, int nmfma_per_iter) {
    float *in, *out; long long *cyc;
    hipMalloc(&in, 1 << 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<float> h(1 << 14, 0.001f);
    hipMemcpy(in, h.data(), 1 << 16, hipMemcpyHostToDevice);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 256>>>(out, in, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 256>>>(out, in, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
    printf("%-34s memtime-ticks/MFMA %7.2f   wall ns/MFMA %6.2f\n", name, avg / (iters * nmfma_per_iter), ms * 1e6 / (iters * nmfma_per_iter));
    hipFree(in); hipFree(out); hipFree(cyc);
}

#define RUN16(MODE, NV, NACC) run("16x16x4 mode" #MODE " valu" #NV " acc" #NACC, probe16<MODE, NV, NACC>, 16)
int main() {
    RUN16(0, 0, 1); RUN16(0, 0, 2); RUN16(0, 0, 4);
    RUN16(0, 1, 2); RUN16(0, 2, 2); RUN16(0, 3, 2); RUN16(0, 4, 2); RUN16(0, 6, 2); RUN16(0, 8, 2);
    RUN16(1, 0, 2); RUN16(1, 2, 2); RUN16(2, 0, 2); RUN16(2, 2, 2);
    run("32x32x2 valu0", probe32<0>, 16); run("32x32x2 valu2", probe32<2>, 16); run("32x32x2 valu4", probe32<4>, 16);
    run("32x32x2 valu8", probe32<8>, 16); run("32x32x2 valu12", probe32<12>, 16);
    return 0;
}
