// Launch plans: the library calls of one sparse forward (and of the mask -> index pipeline in front of it) recorded ONCE
// and replayed from C with the active-tile counts of the CURRENT mask -- no Python, no ctypes, no re-capture per edit.
//
// Why: the reference sizes every launch from `activeIndices.size(0)` (sige/cuda/gather_kernel.cu:78-84,111;
// sige/utils.py:30; sige/nn/gather.py:101-107) and two of its three applications run ONE sparse forward per mask
// (gaugan/runner.py:150-195, diffusion_demo/runner.py:134-164).  A hipGraph bakes grids and counts in, so a new edit meant a
// new capture (5.3 ms of host time for 102 launches, round 3) or an eager forward through ~18 000 Python calls (4.25 ms).
// A plan keeps the ARGUMENTS of every entry-point call; an argument that is an active-tile count is looked up, at replay,
// under the pointer of the index list (or tile table) it belongs to.  Everything a launch decides from the count -- output
// block, grid, K split, tickets -- is decided inside the entry point, i.e. again at replay.  Buffers sized by a count are
// allocated for the largest possible count by the host side (sige_amd/plan.py), so every pointer stays valid.
//
// Recording is per host thread (like the conv-pair state); a call made while a plan records is executed AND stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace sige {

struct Plan;

struct PlanCall {
    // what THIS call contributed to Plan::unbound / Plan::shape_bound when it was recorded: a call truncated out of the plan (a probe
    // that returned an error, sige_hip_plan_truncate) takes its contribution with it
    int unbound = 0;
    bool fixed = false;
    virtual ~PlanCall() {}
    virtual int run(Plan &p, hipStream_t st) = 0;
};

enum { PLAN_MASKS = 0, PLAN_FORWARD = 1, PLAN_SECTIONS = 2 };

struct Plan {
    std::vector<std::unique_ptr<PlanCall>> calls[PLAN_SECTIONS];
    std::vector<int> slots;                              // active-tile counts, one per index list of the network
    std::unordered_map<const void *, int> slot_of;       // index list / tile table pointer -> slot
    int32_t *host_counts = nullptr;                      // pinned staging of the count read-back
    int host_cap = 0;
    int device = -1;
    long runs[PLAN_SECTIONS] = {0, 0};
    std::atomic<bool> recording{false};                  // between plan_begin and plan_end (on some thread)
    // a recorded call sizes its work from a count that has no index-list argument to look it up under (the NCHW / two-kernel
    // forms: tiles = B * N handed over as one number): such a plan only replays under the counts it was recorded with
    bool shape_bound = false;
    // pointers of index lists whose count never changes with the mask (the all-tiles list of a dense layer): a count argument
    // next to one of these keeps its recorded value by design
    std::unordered_set<const void *> const_ptrs;
    // count arguments recorded next to a pointer that is neither bound to a slot nor registered as constant (a copy of an index
    // list made outside the recorded mask pipeline): such a count cannot follow a new mask, so the plan is marked shape_bound
    int unbound = 0;
    ~Plan() {
        if (host_counts) (void)hipHostFree(host_counts);
    }
    // the count to use for an argument recorded next to `key` (the recorded value if the pointer was never bound)
    int lookup(const void *key, int recorded) const {
        const auto it = slot_of.find(key);
        return it == slot_of.end() ? recorded : slots[(size_t)it->second];
    }
};

// the plan (and section) the calling thread records into, nullptr otherwise; cleared while a plan replays
extern thread_local Plan *g_plan_rec;
extern thread_local int g_plan_section;

// "argument NP is the active-tile count of the index list / tile table in argument IP"
template <int IP, int NP>
struct CountOf {
    static constexpr int ip = IP, np = NP;
};

template <typename P, typename T>
inline void plan_patch(const Plan &p, T &t) {
    std::get<P::np>(t) = p.lookup(static_cast<const void *>(std::get<P::ip>(t)), std::get<P::np>(t));
}

// at RECORD time: is the pointer a count argument hangs on known to the plan?  (a miss used to fall back to the
// recorded count silently -- stale counts under every later mask.)  A null pointer is an absent operand (its count is 0).
template <typename P, typename T>
inline void plan_check_key(Plan &p, PlanCall &call, const T &t) {
    const void *key = static_cast<const void *>(std::get<P::ip>(t));
    if (!key || p.slot_of.count(key) || p.const_ptrs.count(key)) return;
    ++call.unbound;
    ++p.unbound;
    p.shape_bound = true;
}

// one recorded entry-point call: the function, its arguments by value, which of them are counts.  STREAM: the last argument
// is the stream (replaced by the replay's).
template <bool STREAM, typename Patches, typename... A>
struct TypedCall;
template <bool STREAM, typename... Ps, typename... A>
struct TypedCall<STREAM, std::tuple<Ps...>, A...> : PlanCall {
    int (*fn)(A...);
    std::tuple<std::decay_t<A>...> args;
    TypedCall(int (*f)(A...), A... a) : fn(f), args(a...) {}
    int run(Plan &p, hipStream_t st) override {
        auto t = args;
        (plan_patch<Ps>(p, t), ...);
        if constexpr (STREAM) std::get<sizeof...(A) - 1>(t) = static_cast<void *>(st);
        return std::apply(fn, t);
    }
};

template <bool STREAM, typename... Ps, typename... A, typename... B>
inline void plan_record(int (*fn)(A...), B... b) {
    Plan *p = g_plan_rec;
    if (!p) return;
    auto *call = new TypedCall<STREAM, std::tuple<Ps...>, A...>(fn, static_cast<A>(b)...);
    (plan_check_key<Ps>(*p, *call, call->args), ...);
    p->calls[g_plan_section].emplace_back(call);
}

}  // namespace sige

// First statement of a recordable entry point.  `...` = the entry point's parameters, in order.
#define SIGE_PLAN_HOOK(fn, ...)                                                        \
    do {                                                                               \
        if (sige::g_plan_rec) sige::plan_record<true>(&fn, __VA_ARGS__);              \
    } while (0)
// ... for an entry point whose sizes cannot follow a new mask (see Plan::shape_bound)
#define SIGE_PLAN_HOOK_FIXED(fn, ...)                                                  \
    do {                                                                               \
        if (sige::g_plan_rec) {                                                        \
            sige::g_plan_rec->shape_bound = true;                                      \
            sige::plan_record<true>(&fn, __VA_ARGS__);                                 \
            sige::g_plan_rec->calls[sige::g_plan_section].back()->fixed = true;        \
        }                                                                              \
    } while (0)
// ... with the (index list, count) argument positions: SIGE_PLAN_HOOK_N(fn, (sige::CountOf<9, 10>), args...)
#define SIGE_PLAN_UNPAREN(...) __VA_ARGS__
#define SIGE_PLAN_HOOK_N(fn, patches, ...)                                                              \
    do {                                                                                                \
        if (sige::g_plan_rec) sige::plan_record<true, SIGE_PLAN_UNPAREN patches>(&fn, __VA_ARGS__);   \
    } while (0)
// entry points without arguments (conv_pair_begin / _end)
#define SIGE_PLAN_HOOK0(fn)                                        \
    do {                                                           \
        if (sige::g_plan_rec) sige::plan_record<false>(&fn);       \
    } while (0)
