// Probe (not part of the product; round 6): what does it cost a launch to WRITE B bytes, by store flavour?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_probe tools/probe/store_probe.hip && ./store_probe [out.json]
// conv_in (33.5 MB written) takes 16.6 us whatever its structure (LDS-transposed 16-byte stores, 8 waves per CU: unchanged), and
// tools/probe/boundary_split.hip's plain coalesced 16.8 MB write took 7.2 us = 2.3 TB/s.  The L2 is a write-back cache: plain stores
// leave dirty lines that the end-of-kernel release writes back; MI355X_MICROARCH.md's publish-large row has write-through (sc1)
// stores 2.7x faster than plain stores + release for 64 KB per workgroup.  Chains of 60 dependent launches in a hipGraph, each
// writing B bytes as 16-byte stores, consecutive lanes consecutive addresses; us per launch minus the empty launch's 1.62.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR>
__device__ __forceinline__ void st(float4 *p, f32x4 v) {
    if constexpr (FLAVOUR == 0) *reinterpret_cast<f32x4 *>(p) = v;
    else if constexpr (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// every workgroup writes `per_wg` float4 per lane; lane-consecutive 16 bytes (1 KB per wave instruction); PATTERN 0: a workgroup's
// region is contiguous; 1: grid-strided (instruction i of all workgroups together covers a contiguous span)
template <int FLAVOUR, int PATTERN>
__global__ __launch_bounds__(256) void writer(float4 *dst, int per_lane, float seed) {
    const f32x4 v = {seed, seed + 1.f, seed + 2.f, (float)threadIdx.x};
    if (PATTERN == 0) {
        float4 *q = dst + ((long)blockIdx.x * per_lane) * 256 + threadIdx.x;
        for (int i = 0; i < per_lane; ++i) st<FLAVOUR>(q + (long)i * 256, v);
    } else {
        float4 *q = dst + (long)blockIdx.x * 256 + threadIdx.x;
        for (int i = 0; i < per_lane; ++i) st<FLAVOUR>(q + (long)i * 256 * gridDim.x, v);
    }
}

__global__ void empty(float *p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[1] == 12345.f) p[0] = 1.f; }

static hipStream_t st_;
struct Row { std::string name; double us; };
static std::vector<Row> rows;

template <typename F>
static double chain(F launch, int n = 60) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) launch(i);
    CK(hipStreamEndCapture(st_, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(ex, st_));
    CK(hipStreamSynchronize(st_));
    std::vector<double> v;
    for (int r = 0; r < 9; ++r) {
        CK(hipEventRecord(e0, st_)); CK(hipGraphLaunch(ex, st_)); CK(hipEventRecord(e1, st_)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3 / n);
    }
    std::sort(v.begin(), v.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    CK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    const long BYTES = 1L << 30;
    float4 *dst; CK(hipMalloc(&dst, BYTES)); CK(hipMemset(dst, 0, BYTES));
    float *small; CK(hipMalloc(&small, 64)); CK(hipMemset(small, 0, 64));
    const double base = chain([&](int) { empty<<<256, 256, 0, st_>>>(small); });
    printf("empty launch: %.2f us\n", base);
    rows.push_back({"empty launch", base});
    const char *fl[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"};
    const char *pt[] = {"contiguous per workgroup", "grid-strided"};
    for (int mb : {1, 4, 16, 32}) {
        for (int grid : {256, 1024, 4096}) {
            const long f4 = (long)mb * (1 << 20) / 16;
            const int per_lane = (int)(f4 / ((long)grid * 256));
            if (per_lane < 1) continue;
            for (int pattern = 0; pattern < 2; ++pattern) {
                for (int f = 0; f < 5; ++f) {
                    // (each launch of the chain writes its own window of the 1 GiB buffer: no line is rewritten while still dirty)
                    auto L = [&](int i) {
                        float4 *d = dst + ((long)i * f4) % (BYTES / 16 - f4);
#define W(F, P) writer<F, P><<<grid, 256, 0, st_>>>(d, per_lane, (float)i)
                        if (pattern == 0) { if (f == 0) W(0, 0); else if (f == 1) W(1, 0); else if (f == 2) W(2, 0); else if (f == 3) W(3, 0); else W(4, 0); }
                        else { if (f == 0) W(0, 1); else if (f == 1) W(1, 1); else if (f == 2) W(2, 1); else if (f == 3) W(3, 1); else W(4, 1); }
#undef W
                    };
                    const double us = chain(L);
                    char name[256];
                    snprintf(name, sizeof name, "%2d MB, grid %4d, %s, %s stores", mb, grid, pt[pattern], fl[f]);
                    rows.push_back({name, us});
                    printf("%-78s %7.2f us  -> %6.2f TB/s net of the empty launch\n", name, us, mb * 1.048576 / (us - base));
                    fflush(stdout);
                }
            }
        }
    }
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"unit\": \"us per dependent launch (hipGraph of 60 launches, median of 9 replays)\", \"rows\": [\n");
        for (size_t i = 0; i < rows.size(); ++i) fprintf(f, "  {\"case\": \"%s\", \"us\": %.3f}%s\n", rows[i].name.c_str(), rows[i].us, i + 1 < rows.size() ? "," : "");
        fprintf(f, "]}\n"); fclose(f);
    }
    return 0;
}
