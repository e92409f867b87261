// Version / error / device queries of libsige_hip.so.
#include "common.hpp"

#include <atomic>
#include <climits>
#include <mutex>
#include <vector>

// ---- measurement builds: the dispatch knobs (csrc/tuning.hpp) ----
#ifdef SIGE_HIP_TUNING
namespace sige {
std::atomic<int> g_tuning[SIGE_HIP_TUNE_COUNT] = {{kTuningDefaults[0]}, {kTuningDefaults[1]}, {kTuningDefaults[2]}, {kTuningDefaults[3]},
                                                  {kTuningDefaults[4]}, {kTuningDefaults[5]}, {kTuningDefaults[6]}, {kTuningDefaults[7]},
                                                  {kTuningDefaults[8]}, {kTuningDefaults[9]}, {kTuningDefaults[10]}, {kTuningDefaults[11]}, {kTuningDefaults[12]}, {kTuningDefaults[13]}};
}
static_assert(SIGE_HIP_TUNE_COUNT == 14, "g_tuning's initialiser lists every key");

extern "C" int sige_hip_tuning_set(int key, int value) {
    bool ok = false;
    switch (key) {
        case SIGE_HIP_TUNE_CONV_TILE_MT: ok = value == 0 || value == 16 || value == 32; break;
        case SIGE_HIP_TUNE_CONV_TILE_NB: ok = value >= 0 && value <= 2; break;
        case SIGE_HIP_TUNE_CONV_WAVES: ok = value == 0 || value == 4 || value == 8; break;
        case SIGE_HIP_TUNE_CONV_LARGE_GRID_NB1: ok = value >= -1; break;
        case SIGE_HIP_TUNE_CONV_KSPLIT: ok = value >= 0 && value <= 8; break;
        case SIGE_HIP_TUNE_CONV_KSPLIT_SECOND_PASS: ok = value == 0 || value == 1; break;
        case SIGE_HIP_TUNE_TILE3_F16_TPW4_MIN: ok = value >= -1; break;
        case SIGE_HIP_TUNE_TILE3_F16_PAIR_MIN: ok = value >= -1; break;
        case SIGE_HIP_TUNE_TILE3_F16_SPARSE_MIN: ok = value >= -1; break;
        case SIGE_HIP_TUNE_GATHER_ONE_TILE_ROWS: ok = value == 0 || value == 1; break;
        case SIGE_HIP_TUNE_SCATTER_GATHER_FORM: ok = value >= 0 && value <= 3; break;
        case SIGE_HIP_TUNE_SMALL_COUT_SCALAR: ok = value == 0 || value == 1; break;
        case SIGE_HIP_TUNE_WIDE_KSPLIT: ok = value >= 0 && value <= 16; break;  // (conv_wide.hpp: kWideMaxSplit)
        case SIGE_HIP_TUNE_ATTENTION_FORM: ok = value >= 0 && value <= 2; break;
        default: return SIGE_HIP_EINVAL;
    }
    if (!ok) return SIGE_HIP_EINVAL;
    sige::g_tuning[key].store(value, std::memory_order_relaxed);
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_tuning_get(int key) {
    if (key < 0 || key >= SIGE_HIP_TUNE_COUNT) return INT_MIN;
    return sige::g_tuning[key].load(std::memory_order_relaxed);
}
#endif

// ---- code-object preload (include/sige_hip.h: sige_hip_preload) ----
namespace {
struct PreloadState {
    std::mutex mu;
    std::vector<const void *> anchors;  // one kernel per translation unit
    bool loaded[64] = {};
};
PreloadState &preload_state() {
    static PreloadState s;  // (constructed on first use: the registrars below run from other units' static initialisers)
    return s;
}
}  // namespace

void sige::preload_register(const void *host_kernel) {
    PreloadState &s = preload_state();
    std::lock_guard<std::mutex> lock(s.mu);
    s.anchors.push_back(host_kernel);
}

extern "C" int sige_hip_preload(void) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return SIGE_HIP_ENODEVICE;
    PreloadState &s = preload_state();
    std::lock_guard<std::mutex> lock(s.mu);
    if (dev < 64 && s.loaded[dev]) return 0;
    int n = 0;
    for (const void *k : s.anchors) {
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, k) != hipSuccess) {
            (void)hipGetLastError();
            return SIGE_HIP_ELAUNCH;
        }
        ++n;
    }
    if (dev < 64) s.loaded[dev] = true;
    return n;
}

static std::atomic<long> g_launches{0};
static std::atomic<int> g_last_device{-1};
void sige::note_launches(int kernels) {
    g_launches.fetch_add(kernels, std::memory_order_relaxed);
    // which device HIP had current when the entry point launched (the C ABI launches on the CURRENT device: the caller guards);
    // a debug aid next to the launch counter, so that the host-side device guard can be tested on a one-GPU box
    int dev = -1;
    if (kernels > 0 && hipGetDevice(&dev) == hipSuccess) g_last_device.store(dev, std::memory_order_relaxed);
}

// ---- stacked edits ("throughput mode") ----
// E edited versions of ONE original image, each with its own mask, run as ONE tall image: every activation [E,C,H,W]
// (channels-last) is handed over as [1,C,E*H,W] -- the same bytes --, masks and cached tensors are stacked the same way, and
// every launch then sees the active tiles of all E edits at once (the sum of their tile counts: what lifts a 1 % edit out of the
// launch-bound regime).  The only thing a kernel has to know is where one image ends: a halo row on the other side of a seam is
// zero padding, not the neighbour's pixels.  Per host thread, like the conv-pair state.
static thread_local int g_edit_batch = 1;

int sige::stacked_shift(int H) {
    const int E = g_edit_batch;
    if (E <= 1) return 0;
    if (H <= 0 || H % E) return -1;
    const int hp = H / E;
    if (hp < 4 || (hp & (hp - 1))) return -1;  // (one image's height must be a power of two: the seam test is a shift)
    int s = 0;
    while ((1 << s) < hp) ++s;
    return s;
}

extern "C" int sige_hip_set_edit_batch(int E) {
    if (sige::g_plan_rec) sige::plan_record<false>(&sige_hip_set_edit_batch, E);
    if (E < 1 || E > 4096) return SIGE_HIP_EINVAL;
    g_edit_batch = E;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_get_edit_batch(void) { return g_edit_batch; }

extern "C" int sige_hip_version(void) { return SIGE_HIP_VERSION; }

extern "C" int64_t sige_hip_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

extern "C" int sige_hip_last_launch_device(void) { return g_last_device.load(std::memory_order_relaxed); }

extern "C" const char *sige_hip_error_string(int status) {
    switch (status) {
        case SIGE_HIP_OK: return "ok";
        case SIGE_HIP_EINVAL: return "invalid argument (null pointer, negative or inconsistent size, non-broadcastable operand)";
        case SIGE_HIP_EUNSUPPORTED: return "unsupported (unknown activation or shape outside the kernel's limits)";
        case SIGE_HIP_ELAUNCH: return "HIP launch failed (hipGetLastError != hipSuccess)";
        case SIGE_HIP_ENODEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

extern "C" const char *sige_hip_device_arch(void) {
    static char arch[256];
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return nullptr;
    snprintf(arch, sizeof(arch), "%s", prop.gcnArchName);
    return arch;
}
