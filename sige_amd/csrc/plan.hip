// Launch plans (plan.hpp): C ABI.  Host-side mirror: sige_amd/plan.py.
#include "common.hpp"

#include <new>

namespace sige {

thread_local Plan *g_plan_rec = nullptr;
thread_local int g_plan_section = 0;

// The one synchronisation of the mask -> index pipeline (the reference's torch.nonzero synchronises the same way,
// sige/utils.py:30): `n` counts written by the compaction kernels recorded before it are copied to the host and become the
// values of slots [first, first + n); every call recorded AFTER it sees the new counts.
struct ReadbackCall : PlanCall {
    const int32_t *dev;
    int first, n;
    ReadbackCall(const int32_t *d, int f, int k) : dev(d), first(f), n(k) {}
    int run(Plan &p, hipStream_t st) override {
        if (n <= 0) return SIGE_HIP_OK;
        if (p.host_cap < n) {
            if (p.host_counts) (void)hipHostFree(p.host_counts);
            p.host_counts = nullptr;
            if (hipHostMalloc(reinterpret_cast<void **>(&p.host_counts), sizeof(int32_t) * (size_t)n, hipHostMallocDefault) != hipSuccess) {
                p.host_cap = 0;
                (void)hipGetLastError();
                return SIGE_HIP_ELAUNCH;
            }
            p.host_cap = n;
        }
        if (hipMemcpyAsync(p.host_counts, dev, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            (void)hipGetLastError();
            return SIGE_HIP_ELAUNCH;
        }
        for (int i = 0; i < n; ++i) p.slots[(size_t)(first + i)] = p.host_counts[i];
        return SIGE_HIP_OK;
    }
};

}  // namespace sige

using namespace sige;

extern "C" void *sige_hip_plan_create(void) {
    Plan *p = new (std::nothrow) Plan();
    if (p && hipGetDevice(&p->device) != hipSuccess) p->device = -1;
    return p;
}

extern "C" int sige_hip_plan_destroy(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p) return SIGE_HIP_EINVAL;
    if (g_plan_rec == p) g_plan_rec = nullptr;
    else if (p->recording.load(std::memory_order_acquire)) return SIGE_HIP_EINVAL;  // (another thread records into it: its thread-local pointer cannot be cleared from here)
    delete p;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_begin(void *plan, int section, int append) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || section < 0 || section >= PLAN_SECTIONS || g_plan_rec) return SIGE_HIP_EINVAL;
    if (!append) p->calls[section].clear();
    p->recording.store(true, std::memory_order_release);
    g_plan_rec = p;
    g_plan_section = section;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_end(void *plan) {
    if (!plan || g_plan_rec != static_cast<Plan *>(plan)) return SIGE_HIP_EINVAL;
    g_plan_rec = nullptr;
    static_cast<Plan *>(plan)->recording.store(false, std::memory_order_release);
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_recording(void) { return g_plan_rec ? 1 : 0; }

extern "C" int sige_hip_plan_shape_bound(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    return p ? (p->shape_bound ? 1 : 0) : -1;
}

extern "C" int sige_hip_plan_calls(void *plan, int section) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || section < 0 || section >= PLAN_SECTIONS) return -1;
    return (int)p->calls[section].size();
}

extern "C" int sige_hip_plan_new_slots(void *plan, int n) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || n < 0) return -1;
    const int first = (int)p->slots.size();
    p->slots.resize((size_t)(first + n), 0);
    return first;
}

extern "C" int sige_hip_plan_bind_ptr(void *plan, const void *ptr, int slot) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || !ptr || slot < 0 || slot >= (int)p->slots.size()) return SIGE_HIP_EINVAL;
    p->slot_of[ptr] = slot;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_bind_const(void *plan, const void *ptr) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || !ptr) return SIGE_HIP_EINVAL;
    p->const_ptrs.insert(ptr);
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_unbound(void *plan) {
    Plan *p = static_cast<Plan *>(plan);
    return p ? p->unbound : -1;
}

extern "C" int sige_hip_plan_truncate(void *plan, int section, int calls) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || section < 0 || section >= PLAN_SECTIONS || calls < 0 || calls > (int)p->calls[section].size()) return SIGE_HIP_EINVAL;
    p->calls[section].resize((size_t)calls);
    // what the removed calls had marked goes with them: the plan is shape bound iff a REMAINING call says so
    p->unbound = 0;
    p->shape_bound = false;
    for (int s = 0; s < PLAN_SECTIONS; ++s)
        for (const auto &c : p->calls[s]) {
            p->unbound += c->unbound;
            p->shape_bound = p->shape_bound || c->fixed || c->unbound > 0;
        }
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_set_slot(void *plan, int slot, int count) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || slot < 0 || slot >= (int)p->slots.size() || count < 0) return SIGE_HIP_EINVAL;
    p->slots[(size_t)slot] = count;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_get_slots(void *plan, int32_t *out, int n) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || (n > 0 && !out)) return -1;
    const int m = (int)p->slots.size() < n ? (int)p->slots.size() : n;
    for (int i = 0; i < m; ++i) out[i] = p->slots[(size_t)i];
    return (int)p->slots.size();
}

extern "C" int sige_hip_plan_record_readback(void *plan, const int32_t *device_counts, int first_slot, int n) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || g_plan_rec != p || !device_counts || n < 0 || first_slot < 0 || first_slot + n > (int)p->slots.size())
        return SIGE_HIP_EINVAL;
    p->calls[g_plan_section].emplace_back(new ReadbackCall(device_counts, first_slot, n));
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_plan_run(void *plan, int section, void *stream) {
    Plan *p = static_cast<Plan *>(plan);
    if (!p || section < 0 || section >= PLAN_SECTIONS || g_plan_rec) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    ++p->runs[section];
    for (auto &c : p->calls[section]) {
        const int rc = c->run(*p, st);
        // (SIGE_HIP_EUNSUPPORTED from a call that succeeded when it was recorded means the new counts left the shapes its
        //  kernels cover: the caller records again under the new mask)
        if (rc != SIGE_HIP_OK) {
            (void)sige_hip_conv_pair_end();  // (a replay that stops between pair_begin and pair_end must not leave a held conv behind)
            return rc;
        }
    }
    return SIGE_HIP_OK;
}
