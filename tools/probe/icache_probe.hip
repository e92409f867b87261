// Probe (not part of the product; VERDICT r5 next #3, second step): what does straight-line code cost per launch?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o icache_probe tools/probe/icache_probe.hip && ./icache_probe [out.json]
// tools/probe/boundary_split.hip found the dependent boundary itself at 1.62 us whatever the grid, LDS, registers, arguments or
// the number of distinct kernels -- and 15 us for a kernel with ~30 KB of straight-line code.  The tile kernels are 6 - 34 KB of
// code executed front to back (prologue, unrolled chunk body, epilogue).  Here: KB kilobytes of v_fma (8 bytes each, four
// independent accumulators) executed ONCE per wave, against a LOOP executing the same instruction count from 512 bytes of code:
// the difference is instruction fetch.  Chains of 120 dependent launches in a hipGraph, us per launch; the same kernel repeated or
// 8 distinct kernels of the same size cycled.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#define F4 asm volatile("v_fma_f32 %0, %0, %4, 0.5\n\tv_fma_f32 %1, %1, %4, 0.5\n\tv_fma_f32 %2, %2, %4, 0.5\n\tv_fma_f32 %3, %3, %4, 0.5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16   /* 64 instructions = 512 bytes */

template <int N> struct Fat {
    __device__ static __forceinline__ void run(float &a0, float &a1, float &a2, float &a3, float m) {
        F64
        Fat<N - 1>::run(a0, a1, a2, a3, m);
    }
};
template <> struct Fat<0> { __device__ static __forceinline__ void run(float &, float &, float &, float &, float) {} };

// HALFKB = code size in units of 512 bytes; LOOP: the same instruction count from one 512-byte body
template <int ID, int HALFKB, bool LOOP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    const float m = 1.0f + 1e-7f * (float)(ID + 1);
    if constexpr (LOOP) {
        for (int i = 0; i < HALFKB; ++i) { F64 }
    } else {
        Fat<HALFKB>::run(a0, a1, a2, a3, m);
    }
    const float r = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0 && (blockIdx.x == 0 || r == 12345.678f)) out[ID] = r;
}

typedef void (*launch_fn)(float *, int, hipStream_t);
template <int ID, int HALFKB, bool LOOP> void launch(float *out, int grid, hipStream_t s) { k<ID, HALFKB, LOOP><<<grid, 256, 0, s>>>(out, 0.5f); }
template <int HALFKB, bool LOOP, int... I> std::vector<launch_fn> table(std::integer_sequence<int, I...>) { return {launch<I, HALFKB, LOOP>...}; }

static hipStream_t st;
struct Row { std::string name; double us, us_min; };
static std::vector<Row> rows;

static void run(const std::string &name, const std::vector<launch_fn> &fns, int grid, float *out, int n = 120) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) fns[i % fns.size()](out, grid, st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    std::vector<double> v;
    for (int r = 0; r < 15; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3 / n);
    }
    std::sort(v.begin(), v.end());
    rows.push_back({name, v[v.size() / 2], v[0]});
    printf("%-72s %7.2f us per launch (min %7.2f)\n", name.c_str(), v[v.size() / 2], v[0]); fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
}

template <int HALFKB> void size_rows(float *out) {
    const std::string kb = std::to_string(HALFKB / 2) + " KB";
    using seq = std::make_integer_sequence<int, 8>;
    run("loop, " + kb + " worth of instructions, same kernel, grid 256", {launch<0, HALFKB, true>}, 256, out);
    run("straight-line " + kb + ", same kernel, grid 256", {launch<0, HALFKB, false>}, 256, out);
    run("straight-line " + kb + ", 8 distinct kernels, grid 256", table<HALFKB, false>(seq{}), 256, out);
    run("straight-line " + kb + ", same kernel, grid 1024 (4 workgroups per CU)", {launch<0, HALFKB, false>}, 1024, out);
    run("straight-line " + kb + ", same kernel, grid 32", {launch<0, HALFKB, false>}, 32, out);
}

int main(int argc, char **argv) {
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *out; CK(hipMalloc(&out, 4096 * 4)); CK(hipMemset(out, 0, 4096 * 4));
    run("empty kernel", {launch<0, 0, false>}, 256, out);
    size_rows<4>(out); size_rows<8>(out); size_rows<16>(out); size_rows<32>(out); size_rows<48>(out); size_rows<64>(out);
    size_rows<96>(out); size_rows<128>(out);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"unit\": \"us per dependent launch (hipGraph of 120 launches; median of 15 replays, min)\", \"rows\": [\n");
        for (size_t i = 0; i < rows.size(); ++i)
            fprintf(f, "  {\"case\": \"%s\", \"us\": %.3f, \"us_min\": %.3f}%s\n", rows[i].name.c_str(), rows[i].us, rows[i].us_min, i + 1 < rows.size() ? "," : "");
        fprintf(f, "]}\n"); fclose(f);
    }
    return 0;
}
