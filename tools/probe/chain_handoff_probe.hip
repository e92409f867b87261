// Probe (not part of the product; VERDICT r5 next #2, measured instead of argued): conv1 -> conv2 of a sparse residual block as ONE
// launch with per-tile tickets, against the two launches the library issues today.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o chain_handoff_probe tools/probe/chain_handoff_probe.hip && ./chain_handoff_probe [out.json]
// The stand-in for a tile conv launch at a 1.2 % edit (124 tiles, C = 128: 62 tile pairs x 4 output-channel blocks = 248 workgroups
// of 256 lanes, one per CU): every workgroup
//   (a) pulls its WEIGHT slice (72 KB per chunk pair, shared by the 62 workgroups of its channel block -> L2 hits after the first),
//   (b) pulls its ACTIVATION window (18 KB) -- in phase 2 the conv1 outputs of its own tile pair and its two neighbours (all four
//       channel blocks of each: 12 producer workgroups), read with sc1 loads,
//   (c) runs a dependent-FMA stand-in for the K loop (COMPUTE_ITERS, ~5 us),
//   (d) writes its 4 KB of output with sc1 (write-through) stores, drains them (s_waitcnt vmcnt(0)) and -- fused form -- adds 1 to the
//       ticket of its tile pair (four arrivals per tile pair).
// Two launches: phase 1 kernel, phase 2 kernel (plain loads of the outputs: the kernel boundary made them visible).
// One launch:   phase 1; the phase-2 weight pull ISSUED (its 72 KB in flight); a bounded spin until the tickets of the three tile pairs
//               it needs read 4 (one lane polls with sc1 loads and s_sleep; never a hang: a poll budget, then a flag the host prints);
//               phase 2 with sc1 loads of the outputs.  The tickets are re-zeroed by a memset node in front of every pair (both forms
//               carry that node, so that the comparison is fair).
// Chains of 40 pairs in a hipGraph, us per PAIR; weights from the same region every pair (L2-warm) or a new region per pair.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPairs = 62, kNg = 4, kWG = kPairs * kNg;  // 248 workgroups
constexpr int kWeightF4 = 72 * 1024 / 16 / 256;            // float4 loads per lane for the weight slice (18)
constexpr int kOutF4PerWG = 256;                           // 4 KB of output per workgroup: one float4 per lane
constexpr int kComputeIters = 2600;                        // ~5 us of dependent FMAs

// (a buffer load with the sc1 cache-policy bit: the compiler tracks its vmcnt -- an inline-asm load would come back into registers the
//  compiler has already given to something else)
__device__ __forceinline__ f32x4 ld_sc1(const float4 *base, unsigned f4_index, unsigned total_f4) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(base), 0, (int)(total_f4 * 16u), 0x00020000);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(f4_index * 16u), 0, 2 /* sc1 */));
}
__device__ __forceinline__ void st_sc1(float4 *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

struct Args {
    const float4 *w1, *w2;   // weight regions of the two phases: [kNg][kWeightF4 * 256]
    const float4 *x;         // phase-1 activations: [kPairs][1152 float4] (18 KB per tile pair)
    float4 *y1, *y2;         // outputs: [kWG][256 float4]
    int *tickets;            // [kPairs]
    int *gaveup;
};

__device__ __forceinline__ float pull_weights(const float4 *w, int ng, int tid) {
    const float4 *p = w + (size_t)ng * kWeightF4 * 256 + tid;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < kWeightF4; ++i) { const float4 v = p[i * 256]; acc += v.x + v.w; }
    return acc;
}
__device__ __forceinline__ float compute(float a) {
    for (int k = 0; k < kComputeIters; ++k) a = __builtin_fmaf(a, 1.0000001f, 0.25f);
    return a;
}

// phase 1: weights + own activations -> y1 (sc1 stores)
__device__ __forceinline__ void phase1(const Args &a, int pair, int ng, int tid, bool ticket) {
    float acc = pull_weights(a.w1, ng, tid);
    const float4 *xp = a.x + (size_t)pair * 1152 + tid;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = xp[i * 256]; acc += v.x; }
    { const float4 v = xp[(tid < 128 ? 4 * 256 : 0)]; acc += v.y; }
    acc = compute(acc);
    const f32x4 o = {acc, acc + 1.f, acc + 2.f, (float)tid};
    st_sc1(a.y1 + (size_t)(pair * kNg + ng) * kOutF4PerWG + tid, o);
    if (ticket) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.tickets + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// phase 2: weights + the conv1 outputs of tile pairs pair-1, pair, pair+1 (12 producer workgroups) -> y2
template <bool SC1>
__device__ __forceinline__ void phase2(const Args &a, int pair, int ng, int tid, float wacc) {
    float acc = wacc;
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
        const int q = min(max(pair + d, 0), kPairs - 1);
#pragma unroll
        for (int g = 0; g < kNg; ++g) {
            const unsigned e = (unsigned)((q * kNg + g) * kOutF4PerWG + tid);
            if (SC1) { const f32x4 v = ld_sc1(a.y1, e, (unsigned)(kWG * kOutF4PerWG)); acc += v.x + v.w; }
            else { const float4 v = a.y1[e]; acc += v.x + v.w; }
        }
    }
    acc = compute(acc);
    const f32x4 o = {acc, acc + 1.f, acc + 2.f, (float)tid};
    st_sc1(a.y2 + (size_t)(pair * kNg + ng) * kOutF4PerWG + tid, o);
}

__global__ __launch_bounds__(256) void k_phase1(const Args a) { phase1(a, blockIdx.x / kNg, blockIdx.x % kNg, threadIdx.x, false); }
__global__ __launch_bounds__(256) void k_phase2(const Args a) {
    const float w = pull_weights(a.w2, blockIdx.x % kNg, threadIdx.x);
    phase2<false>(a, blockIdx.x / kNg, blockIdx.x % kNg, threadIdx.x, w);
}
// PREFETCH: the phase-2 weight pull is issued BEFORE the wait (its latency sits under the producers' tails)
template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_fused(const Args a) {
    const int pair = blockIdx.x / kNg, ng = blockIdx.x % kNg, tid = threadIdx.x;
    phase1(a, pair, ng, tid, true);
    float w = 0.f;
    if (PREFETCH) w = pull_weights(a.w2, ng, tid);
    __shared__ int ok;
    if (tid == 0) {
        int good = 0;
        for (int spin = 0; spin < (1 << 18) && !good; ++spin) {
            good = 1;
            for (int d = -1; d <= 1; ++d) {
                const int q = min(max(pair + d, 0), kPairs - 1);
                if (__hip_atomic_load(a.tickets + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < kNg) good = 0;
            }
            if (!good) __builtin_amdgcn_s_sleep(2);
        }
        ok = good;
        if (!good) atomicAdd(a.gaveup, 1);
    }
    __syncthreads();
    if (!ok) return;
    if (!PREFETCH) w = pull_weights(a.w2, ng, tid);
    phase2<true>(a, pair, ng, tid, w);
}

static hipStream_t st;
struct Row { std::string name; double us; };
static std::vector<Row> rows;

template <typename F>
static double chain(F pair_launch, int n = 40) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) pair_launch(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ex, st));
    CK(hipStreamSynchronize(st));
    std::vector<double> v;
    for (int r = 0; r < 11; ++r) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ex, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3 / n);
    }
    std::sort(v.begin(), v.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g));
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int NREG = 40;
    const size_t wregion = (size_t)kNg * kWeightF4 * 256;  // float4 per weight region (288 KB)
    float4 *w, *x, *y1, *y2; int *tickets, *gaveup;
    CK(hipMalloc(&w, 2 * NREG * wregion * 16)); CK(hipMemset(w, 0, 2 * NREG * wregion * 16));
    CK(hipMalloc(&x, (size_t)kPairs * 1152 * 16)); CK(hipMemset(x, 0, (size_t)kPairs * 1152 * 16));
    CK(hipMalloc(&y1, (size_t)kWG * kOutF4PerWG * 16)); CK(hipMalloc(&y2, (size_t)kWG * kOutF4PerWG * 16));
    CK(hipMalloc(&tickets, kPairs * 4)); CK(hipMalloc(&gaveup, 4)); CK(hipMemset(gaveup, 0, 4)); CK(hipMemset(tickets, 0, kPairs * 4));
    auto args = [&](int i, bool cold) {
        Args a{};
        const int r = cold ? i % NREG : 0;
        a.w1 = w + (size_t)(2 * r) * wregion; a.w2 = w + (size_t)(2 * r + 1) * wregion;
        a.x = x; a.y1 = y1; a.y2 = y2; a.tickets = tickets; a.gaveup = gaveup;
        return a;
    };
    for (int cold = 0; cold < 2; ++cold) {
        const std::string tag = cold ? "a new weight region per pair" : "the same weight region every pair (L2-warm)";
        const double two = chain([&](int i) {
            CK(hipMemsetAsync(tickets, 0, kPairs * 4, st));
            const Args a = args(i, cold);
            k_phase1<<<kWG, 256, 0, st>>>(a);
            k_phase2<<<kWG, 256, 0, st>>>(a);
        });
        const double fused = chain([&](int i) {
            CK(hipMemsetAsync(tickets, 0, kPairs * 4, st));
            k_fused<false><<<kWG, 256, 0, st>>>(args(i, cold));
        });
        const double fused_pf = chain([&](int i) {
            CK(hipMemsetAsync(tickets, 0, kPairs * 4, st));
            k_fused<true><<<kWG, 256, 0, st>>>(args(i, cold));
        });
        const double one = chain([&](int i) {
            CK(hipMemsetAsync(tickets, 0, kPairs * 4, st));
            k_phase1<<<kWG, 256, 0, st>>>(args(i, cold));
        });
        printf("%s\n  two launches (phase 1, phase 2)                      %7.2f us per pair\n  one launch, per-tile tickets                         %7.2f\n"
               "  one launch, tickets, phase-2 weights issued before the wait %7.2f\n  (phase 1 alone + the memset node                      %7.2f)\n",
               tag.c_str(), two, fused, fused_pf, one);
        rows.push_back({tag + ": two launches", two});
        rows.push_back({tag + ": one launch, per-tile tickets", fused});
        rows.push_back({tag + ": one launch, tickets, phase-2 weights issued before the wait", fused_pf});
        rows.push_back({tag + ": phase 1 alone (+ memset node)", one});
    }
    int g = 0; CK(hipMemcpy(&g, gaveup, 4, hipMemcpyDeviceToHost));
    printf("workgroups that gave up waiting: %d\n", g);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        fprintf(f, "{\"unit\": \"us per conv1 -> conv2 pair (hipGraph of 40 pairs, median of 11 replays)\", \"gave_up\": %d, \"rows\": [\n", g);
        for (size_t i = 0; i < rows.size(); ++i) fprintf(f, "  {\"case\": \"%s\", \"us\": %.3f}%s\n", rows[i].name.c_str(), rows[i].us, i + 1 < rows.size() ? "," : "");
        fprintf(f, "]}\n"); fclose(f);
    }
    return 0;
}
