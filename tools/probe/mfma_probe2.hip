// Micro-probe 2: cost of memory / LDS / scalar instructions issued between f32 MFMAs by the same wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 3: one coalesced dword global load per MFMA (consumed 16 MFMAs later)
// MODE 4: one gather-pattern dword load per MFMA (64 lanes -> ~12 separate rows of 24 B)
// MODE 5: one ds_write_b32 per MFMA
// MODE 6: scalar work per MFMA (uniform integer chain)
// MODE 7: one coalesced dwordx4 load per 4 MFMAs
// MODE 8: MODE 4 with NL loads per MFMA group of 4 (NLOAD template)
template <int MODE, int PER16>
__global__ __launch_bounds__(256) void probe(float *out, const float *in, long long *cyc, int iters, int stride) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = in[tid], b = in[tid + 256];
    float hold[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) hold[i] = 0.f;
    float sum = 0.f;
    int sacc = blockIdx.x;
    // gather pattern: lane -> (channel c = lane/36, row r = (lane%36)/6, col = lane%6) in a [C][256][256] tensor
    const int c = lane / 36, p = lane % 36;
    const size_t goff = (size_t)c * 65536 + (size_t)(p / 6 + (blockIdx.x % 60) * 4) * 256 + (p % 6) + (blockIdx.x / 60) * 8;
    const float *gp = in + (MODE == 4 ? goff : (size_t)blockIdx.x * 4096 + tid);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 1], 0, 0, 0);
            if (u < PER16) {
                if (MODE == 3 || MODE == 4) { sum += hold[u]; hold[u] = gp[(size_t)((it * 16 + u) & 127) * stride]; }
                if (MODE == 5) lds[tid + 256 * (u & 7)] = a;
                if (MODE == 6) sacc = (sacc * 5 + it) ^ (sacc >> 3);
                if (MODE == 7 && (u & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4 *>(in + (size_t)blockIdx.x * 4096 + ((it * 4 + u / 4) & 3) * 1024 + tid * 4);
                    sum += hold[u]; hold[u] = q.x + q.w;
                }
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += hold[i];
    sum += acc[0][0] + acc[1][0] + lds[tid] + (float)sacc;
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char *name, K kern, int stride) {
    float *in, *out; long long *cyc;
    const size_t n = (size_t)64 << 20;  // 256 MiB of input
    hipMalloc(&in, n * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, n * 4);
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, 256>>>(out, in, cyc, iters, stride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, 256>>>(out, in, cyc, iters, stride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
    printf("%-44s ticks/MFMA %7.2f   wall ns/MFMA %6.2f\n", name, avg / (iters * 16), ms * 1e6 / (iters * 16));
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
    run("baseline (no extra)", probe<0, 0>, 0);
    run("coalesced dword load, 1 per MFMA", probe<3, 16>, 256 * 256);
    run("coalesced dword load, 1 per 4 MFMA", probe<3, 4>, 256 * 256);
    run("gather dword load (L2 resident), 1 per MFMA", probe<4, 16>, 0);
    run("gather dword load (walks channels), 1/MFMA", probe<4, 16>, 2 * 65536);
    run("gather dword load (walks channels), 1/4 MFMA", probe<4, 4>, 2 * 65536);
    run("gather dword load (walks channels), 1/2 MFMA", probe<4, 8>, 2 * 65536);
    run("ds_write_b32, 1 per MFMA", probe<5, 16>, 0);
    run("scalar chain per MFMA", probe<6, 16>, 0);
    run("coalesced dwordx4, 1 per 4 MFMA", probe<7, 16>, 0);
    return 0;
}
