// Probe (not part of the product): can a dependent kernel START before its predecessor has finished, and what does a kernel
// boundary cost when the dependency is a flag in memory instead of the queue's barrier?
//   hipcc --offload-arch=gfx950 -O3 -o chain_probe tools/probe/chain_probe.hip && ./chain_probe
// Part 1 (concurrency): kernel A raises flagA and waits (bounded) for flagB; kernel B raises flagB.  A sees flagB only if B ran while A
//   was still running: (a) same stream, plain launches (control: never), (b) same stream, B launched with hipExtAnyOrderLaunch,
//   (c) two streams.
// Part 2 (chain): N kernels of G workgroups, kernel k needs kernel k-1's result.  (a) plain launches on one stream (the queue's
//   barrier), (b) any-order launches on one stream + in-kernel flags (every workgroup of k waits until all G workgroups of k-1
//   have signalled), (c) two streams alternating + flags, (d) = (a) inside a hipGraph, (e) = (c) inside a hipGraph.
//   Each workgroup does `work` dependent FMAs per lane and one 16-byte load of its predecessor's output first.
// Every wait is bounded (wall_clock64, 100 MHz): a kernel that gives up raises *gaveup and the host prints it -- never a hang.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int ld_acq(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_rel(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void kernel_a(int *flagA, const int *flagB, int *saw, long long limit) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st_rel(flagA, 1);
        const long long t0 = wall_clock64();
        int s = 0;
        while (!(s = ld_acq(flagB)) && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(4);
        *saw = s;
    }
}
__global__ void kernel_b(int *flagB) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st_rel(flagB, 1);
}

// one link of the chain: wait for `need` signals on `wait` (nullptr: no wait), read the predecessor's data, work, write, signal
__global__ __launch_bounds__(256) void link(const int *wait, int need, const float4 *in, float4 *out, int *signal, int work, int *gaveup,
                                            long long limit) {
    __shared__ int ok;
    if (wait) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            int s;
            while ((s = ld_acq(wait)) < need && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(1);
            ok = s >= need;
            if (!ok) atomicAdd(gaveup, 1);
        }
        __syncthreads();
        if (!ok) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    // (reads a DIFFERENT workgroup's output of the previous link: a stale cache line would show)
    float4 v = in[(i + 256 * 7) % (gridDim.x * 256)];
    float a = v.x;
    for (int k = 0; k < work; ++k) a = a * 1.0000001f + 0.5f;
    v.x = a; v.y += 1.0f;
    out[i] = v;
    if (signal) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(signal, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static float ms_between(hipEvent_t a, hipEvent_t b) { float m; CK(hipEventElapsedTime(&m, a, b)); return m; }

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 100, G = argc > 2 ? atoi(argv[2]) : 256, WORK = argc > 3 ? atoi(argv[3]) : 2000;
    const long long LIMIT = 2000000;  // 20 ms of 100 MHz ticks
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int *flags, *saw_h;
    CK(hipMalloc(&flags, 4096 * sizeof(int)));
    CK(hipHostMalloc(&saw_h, 64));
    // ---- part 1
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemsetAsync(flags, 0, 64, s1));
        CK(hipStreamSynchronize(s1));
        int *saw_d = flags + 8;
        void *a_args[] = {(void *)&flags, (void *)&flags /*dummy*/, (void *)&saw_d, (void *)&LIMIT};
        int *fa = flags, *fb = flags + 1;
        a_args[0] = &fa; a_args[1] = &fb;
        CK(hipExtLaunchKernel((const void *)kernel_a, dim3(1), dim3(64), a_args, 0, s1, nullptr, nullptr, 0));
        void *b_args[] = {(void *)&fb};
        if (mode == 0) CK(hipExtLaunchKernel((const void *)kernel_b, dim3(1), dim3(64), b_args, 0, s1, nullptr, nullptr, 0));
        if (mode == 1) CK(hipExtLaunchKernel((const void *)kernel_b, dim3(1), dim3(64), b_args, 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch));
        if (mode == 2) CK(hipExtLaunchKernel((const void *)kernel_b, dim3(1), dim3(64), b_args, 0, s2, nullptr, nullptr, 0));
        CK(hipStreamSynchronize(s1));
        CK(hipStreamSynchronize(s2));
        int saw;
        CK(hipMemcpy(&saw, saw_d, 4, hipMemcpyDeviceToHost));
        printf("{\"part\": 1, \"mode\": \"%s\", \"B_ran_while_A_was_running\": %d}\n",
               mode == 0 ? "same stream, plain" : mode == 1 ? "same stream, B any-order" : "two streams", saw);
    }
    // ---- part 2
    float4 *buf[2];
    CK(hipMalloc(&buf[0], (size_t)G * 256 * 16));
    CK(hipMalloc(&buf[1], (size_t)G * 256 * 16));
    int *gaveup = flags + 16, *sig = flags + 64;  // sig[k]: signals of link k
    hipEvent_t e0, e1, ev[2];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto issue = [&](int mode, hipStream_t sa, hipStream_t sb) {
        // mode 0: plain; 1: any-order + flags on sa; 2: alternate sa / sb + flags
        for (int k = 0; k < N; ++k) {
            const int *wait = (mode && k) ? sig + (k - 1) : nullptr;
            int *signal = mode ? sig + k : nullptr;
            const float4 *in = buf[k & 1];
            float4 *out = buf[(k + 1) & 1];
            int need = G, work = WORK;
            void *args[] = {(void *)&wait, (void *)&need, (void *)&in, (void *)&out, (void *)&signal, (void *)&work, (void *)&gaveup, (void *)&LIMIT};
            hipStream_t st = (mode == 2 && (k & 1)) ? sb : sa;
            CK(hipExtLaunchKernel((const void *)link, dim3(G), dim3(256), args, 0, st, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0));
        }
    };
    const char *names[] = {"plain launches, one stream", "any-order launches + flags, one stream", "two streams alternating + flags"};
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<float> t;
        int gave = 0;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(flags + 16, 0, (4096 - 16) * sizeof(int), s1));
            CK(hipMemsetAsync(buf[0], 0, (size_t)G * 256 * 16, s1));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            CK(hipEventRecord(e0, s1));
            if (mode == 2) { CK(hipEventRecord(ev[0], s1)); CK(hipStreamWaitEvent(s2, ev[0], 0)); }
            issue(mode, s1, s2);
            if (mode == 2) { CK(hipEventRecord(ev[1], s2)); CK(hipStreamWaitEvent(s1, ev[1], 0)); }
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            t.push_back(ms_between(e0, e1));
            int g; CK(hipMemcpy(&g, gaveup, 4, hipMemcpyDeviceToHost)); gave += g;
        }
        // the result: every element's .y counts the links it went through
        std::vector<float4> h((size_t)G * 256);
        CK(hipMemcpy(h.data(), buf[N & 1], h.size() * 16, hipMemcpyDeviceToHost));
        int bad = 0;
        for (auto &v : h) bad += v.y != (float)N;
        std::sort(t.begin(), t.end());
        printf("{\"part\": 2, \"mode\": \"%s\", \"links\": %d, \"workgroups\": %d, \"work\": %d, \"median_ms\": %.4f, \"us_per_link\": %.3f, \"min_ms\": %.4f, \"gave_up\": %d, \"wrong_elements\": %d}\n",
               names[mode], N, G, WORK, t[t.size() / 2], 1e3 * t[t.size() / 2] / N, t[0], gave, bad);
    }
    // ---- the same inside a hipGraph (stream capture)
    for (int mode = 0; mode < 3; mode += 2) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(flags + 16, 0, (4096 - 16) * sizeof(int), s1));
        if (mode == 2) { CK(hipEventRecord(ev[0], s1)); CK(hipStreamWaitEvent(s2, ev[0], 0)); }
        issue(mode, s1, s2);
        if (mode == 2) { CK(hipEventRecord(ev[1], s2)); CK(hipStreamWaitEvent(s1, ev[1], 0)); }
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<float> t;
        int gave = 0;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipMemsetAsync(buf[0], 0, (size_t)G * 256 * 16, s1));
            CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s1));
            CK(hipGraphLaunch(ge, s1));
            CK(hipEventRecord(e1, s1));
            CK(hipStreamSynchronize(s1));
            t.push_back(ms_between(e0, e1));
            int gg; CK(hipMemcpy(&gg, gaveup, 4, hipMemcpyDeviceToHost)); gave += gg;
        }
        std::vector<float4> h((size_t)G * 256);
        CK(hipMemcpy(h.data(), buf[N & 1], h.size() * 16, hipMemcpyDeviceToHost));
        int bad = 0;
        for (auto &v : h) bad += v.y != (float)N;
        std::sort(t.begin(), t.end());
        printf("{\"part\": 2, \"mode\": \"hipGraph: %s\", \"links\": %d, \"workgroups\": %d, \"work\": %d, \"median_ms\": %.4f, \"us_per_link\": %.3f, \"min_ms\": %.4f, \"gave_up\": %d, \"wrong_elements\": %d}\n",
               names[mode], N, G, WORK, t[t.size() / 2], 1e3 * t[t.size() / 2] / N, t[0], gave, bad);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
