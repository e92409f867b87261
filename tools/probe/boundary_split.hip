// Probe (not part of the product): what is a dependent kernel boundary made of on this stack?
//   hipcc --offload-arch=gfx950 -O3 -o boundary_split tools/probe/boundary_split.hip && ./boundary_split [out.json]
// The sparse forward is 102 dependent launches whose smallest kernels (gn_finish: 8 workgroups of trivial work) take 4.7 - 4.9 us in
// the kernel trace, against 1.45 us for a chain of trivial kernels in MI355X_MICROARCH.md's price list.  Each row below is a
// hipGraph of N dependent launches replayed R times between HIP events on the capture stream; value = us per launch (median).
// One ingredient varies per row: the same kernel repeated vs N DISTINCT kernels (instruction cache / code TLB), static LDS bytes,
// register budget, kernel-argument bytes, code bytes executed per wave (straight-line "fat" bodies), bytes left dirty, a
// prologue-like pull of KB per workgroup, cold data pages per launch, and an L2 / Infinity-Cache thrash between launches (the
// forward streams 455 MB of weights: nothing a launch needs is still cached when it starts).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Args {
    float *out;         // [>= 4096] one float per launch id
    const float4 *src;  // pull source (>= 256 MB)
    float4 *dirty;      // dirty destination (>= 64 MB)
    int pull_f4;        // float4 loads per lane (prologue-like pull)
    int dirty_f4;       // float4 stores per lane
    long page_off;      // float4 offset of this launch's data window
};
struct BigArgs { Args a; int pad[96]; };  // 64 + 384 bytes of kernel arguments

// straight-line code of ~CODE * 14.5 bytes (measured: CODE 1024 -> 8.8 KB, 2048 -> 29.6 KB of ISA) that every wave executes once: v_fmac with a literal each (no loop: the tile kernels' prologue,
// unrolled K-loop body and epilogue are 10 - 30 KB executed front to back)
template <int ID, int CODE>
__device__ __forceinline__ float fat(float x) {
    if constexpr (CODE > 0) {
#pragma unroll
        for (int i = 0; i < CODE; ++i) x = __builtin_fmaf(x, 1.0f + 1e-7f * (float)(i * 131 + ID * 7 + 1), 0.25f);
    }
    return x;
}

template <int ID, int LDSB, int CODE>
__device__ __forceinline__ void body(const Args &a) {
    __shared__ float lds[LDSB > 0 ? LDSB / 4 : 1];
    const int tid = threadIdx.x;
    float acc = (float)ID;
    if (LDSB > 0) { lds[tid] = acc; __syncthreads(); acc += lds[(tid + 1) & 255]; }
    if (a.pull_f4 > 0) {
        const float4 *p = a.src + a.page_off + ((long)blockIdx.x * a.pull_f4) * 256 + tid;
        for (int i = 0; i < a.pull_f4; ++i) { const float4 v = p[(long)i * 256]; acc += v.x + v.w; }
    }
    acc = fat<ID, CODE>(acc);
    if (a.dirty_f4 > 0) {
        float4 *q = a.dirty + ((long)blockIdx.x * a.dirty_f4) * 256 + tid;
        for (int i = 0; i < a.dirty_f4; ++i) q[(long)i * 256] = make_float4(acc, acc, acc, acc);
    }
    if (tid == 0 && (blockIdx.x == 0 || acc == 12345.678f)) a.out[ID & 4095] = acc;
}

template <int ID, int LDSB, int CODE>
__global__ __launch_bounds__(256) void nullk(const Args a) { body<ID, LDSB, CODE>(a); }
template <int ID, int LDSB, int CODE>
__global__ __launch_bounds__(256) void nullk_bigargs(const BigArgs a) { body<ID, LDSB, CODE>(a.a); }
// register budget: the kernel descriptor asks for >= VG registers per lane (wave launch initialises nothing, but the allocation is
// part of admitting a wave)
template <int ID>
__global__ __launch_bounds__(256) void nullk_v128(const Args a) { asm volatile("" ::: "v127"); body<ID, 0, 0>(a); }
template <int ID>
__global__ __launch_bounds__(256) void nullk_v250(const Args a) { asm volatile("" ::: "v249"); body<ID, 0, 0>(a); }

__global__ __launch_bounds__(256) void thrash(const float4 *src, float *out, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}

typedef void (*launch_fn)(const Args &, int grid, hipStream_t);
template <int ID, int LDSB, int CODE> void launch_plain(const Args &a, int grid, hipStream_t s) { nullk<ID, LDSB, CODE><<<grid, 256, 0, s>>>(a); }
template <int ID, int LDSB, int CODE> void launch_big(const Args &a, int grid, hipStream_t s) { BigArgs b{}; b.a = a; nullk_bigargs<ID, LDSB, CODE><<<grid, 256, 0, s>>>(b); }
template <int ID> void launch_v128(const Args &a, int grid, hipStream_t s) { nullk_v128<ID><<<grid, 256, 0, s>>>(a); }
template <int ID> void launch_v250(const Args &a, int grid, hipStream_t s) { nullk_v250<ID><<<grid, 256, 0, s>>>(a); }

// tables of DISTINCT kernels: 64 lean ones, 32 fat ones of ~16 KB, 32 of ~32 KB
template <int LDSB, int CODE, int... I> std::vector<launch_fn> table(std::integer_sequence<int, I...>) { return {launch_plain<I, LDSB, CODE>...}; }

struct Row { std::string name; double us; double us_min; };
static std::vector<Row> rows;

struct Chain {
    std::vector<launch_fn> fns;  // cycled
    int n = 120, grid = 256;
    Args a{};
    long page_stride = 0;        // float4s between consecutive launches' data windows
    int thrash_every = 0;        // a thrash kernel after every k-th launch (its time is measured separately and subtracted)
};

static hipStream_t st;
static const float4 *g_src; static float *g_out; static long g_src_f4;

static std::pair<double, double> time_graph(hipGraphExec_t ex, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> v;
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        v.push_back(ms * 1e3);
    }
    std::sort(v.begin(), v.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return {v[v.size() / 2], v[0]};  // (median, min) of the replays, us
}

static void run(const std::string &name, const Chain &c) {
    auto build = [&](bool with_kernels, bool with_thrash) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < c.n; ++k) {
            Args a = c.a;
            a.page_off = (c.page_stride * k) % (g_src_f4 / 2);
            if (with_kernels) c.fns[k % c.fns.size()](a, c.grid, st);
            if (with_thrash && c.thrash_every && (k + 1) % c.thrash_every == 0) thrash<<<2048, 256, 0, st>>>(g_src + g_src_f4 / 2, g_out, g_src_f4 / 2);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        const auto t = time_graph(ex, c.thrash_every ? 7 : 21);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
        return t;
    };
    auto p = build(true, true);
    double med = p.first, mn = p.second;
    if (c.thrash_every) {  // subtract the thrash kernels alone
        const auto q = build(false, true);
        med -= q.first; mn -= q.second;
    }
    rows.push_back({name, med / c.n, mn / c.n});
    printf("%-78s %7.2f us per launch (min %7.2f)\n", name.c_str(), med / c.n, mn / c.n);
    fflush(stdout);
}

int main(int argc, char **argv) {
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const long SRC_BYTES = 1L << 30;  // 1 GiB: first half = data windows, second half = thrash source (512 MB > L2 + Infinity Cache)
    float4 *src, *dirty; float *out;
    CK(hipMalloc(&src, SRC_BYTES)); CK(hipMalloc(&dirty, 256L << 20)); CK(hipMalloc(&out, 4096 * 4));
    CK(hipMemset(src, 0, SRC_BYTES)); CK(hipMemset(dirty, 0, 256L << 20)); CK(hipMemset(out, 0, 4096 * 4));
    g_src = src; g_out = out; g_src_f4 = SRC_BYTES / 16;
    Args base{}; base.out = out; base.src = src; base.dirty = dirty;

    const auto lean = table<0, 0>(std::make_integer_sequence<int, 64>{});
    const auto fat16 = table<0, 1024>(std::make_integer_sequence<int, 32>{});
    const auto fat32 = table<0, 2048>(std::make_integer_sequence<int, 32>{});
    auto one = [](launch_fn f) { return std::vector<launch_fn>{f}; };

    Chain c; c.a = base;
    // 1. grid size, same trivial kernel
    for (int g : {8, 128, 256, 512, 1024}) { c.fns = one(lean[0]); c.grid = g; run("same trivial kernel, grid " + std::to_string(g), c); }
    c.grid = 256;
    // 2. distinct trivial kernels (64 instantiations cycled)
    c.fns = lean; run("64 distinct trivial kernels, grid 256", c);
    c.grid = 8; run("64 distinct trivial kernels, grid 8", c); c.grid = 256;
    // 3. static LDS
    c.fns = one(launch_plain<0, 16384, 0>); run("same kernel, LDS 16 KB", c);
    c.fns = one(launch_plain<0, 47104, 0>); run("same kernel, LDS 46 KB", c);
    c.fns = one(launch_plain<0, 65536, 0>); run("same kernel, LDS 64 KB", c);
    // 4. register budget
    c.fns = one(launch_v128<0>); run("same kernel, 128 VGPRs", c);
    c.fns = one(launch_v250<0>); run("same kernel, 250 VGPRs", c);
    // 5. kernel-argument bytes
    c.fns = one(launch_big<0, 0, 0>); run("same kernel, 448 B of kernel arguments", c);
    // 6. code bytes
    c.fns = one(fat16[0]); run("same kernel, ~9 KB straight-line code", c);
    c.fns = one(fat32[0]); run("same kernel, ~30 KB straight-line code", c);
    c.fns = fat16; run("32 distinct kernels of ~9 KB code", c);
    c.fns = fat32; run("32 distinct kernels of ~30 KB code", c);
    // 7. dirty bytes left by the predecessor (per launch: grid * 256 lanes * 16 B * dirty_f4)
    c.fns = one(lean[0]);
    for (int d : {1, 4, 16}) { c.a = base; c.a.dirty_f4 = d; run("same kernel, " + std::to_string(d * 256 * 256 * 16 / 1024) + " KB written per launch", c); }
    c.a = base;
    // 8. prologue-like pull (per workgroup: 256 lanes * 16 B * pull_f4), same window every launch (warm) vs a new window (cold)
    for (int p : {4, 16, 32}) {
        c.a = base; c.a.pull_f4 = p; c.page_stride = 0;
        run("same kernel, pull " + std::to_string(p * 4) + " KB per workgroup, warm window", c);
        c.page_stride = (long)256 * p * 256 + 4096;
        run("same kernel, pull " + std::to_string(p * 4) + " KB per workgroup, new window per launch", c);
    }
    c.a = base; c.page_stride = 0;
    // 9. everything a launch of the forward meets: distinct fat kernels, LDS, pull from a new window, a little dirty data ...
    c.fns = table<47104, 1024>(std::make_integer_sequence<int, 32>{});
    c.a.pull_f4 = 16; c.a.dirty_f4 = 1; c.page_stride = (long)256 * 16 * 256 + 4096;
    run("32 distinct 9 KB-code kernels, LDS 46 KB, pull 64 KB cold, 1 MB written", c);
    // ... and with the caches thrashed every 4 launches (512 MB streamed: code, kernel arguments and data all come from HBM)
    c.thrash_every = 4; run("  + L2 / Infinity Cache thrashed every 4 launches", c);
    c.fns = one(launch_plain<0, 47104, 1024>); run("  same, but ONE kernel repeated (code stays hot only in the instruction cache)", c);
    c.fns = lean; c.a = base; c.page_stride = 0; run("64 distinct trivial kernels, caches thrashed every 4 launches", c);
    c.fns = one(lean[0]); run("same trivial kernel, caches thrashed every 4 launches", c);
    c.thrash_every = 0;

    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"unit\": \"us per dependent launch (hipGraph of 120 launches; median of the replays, min)\", \"rows\": [\n");
        for (size_t i = 0; i < rows.size(); ++i)
            fprintf(f, "  {\"case\": \"%s\", \"us\": %.3f, \"us_min\": %.3f}%s\n", rows[i].name.c_str(), rows[i].us, rows[i].us_min, i + 1 < rows.size() ? "," : "");
        fprintf(f, "]}\n");
        fclose(f);
    }
    return 0;
}
