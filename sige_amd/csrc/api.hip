// Version / error / device queries of libsige_hip.so.
#include "common.hpp"

#include <atomic>

static std::atomic<long> g_launches{0};
static std::atomic<int> g_last_device{-1};
void sige::note_launches(int kernels) {
    g_launches.fetch_add(kernels, std::memory_order_relaxed);
    // which device HIP had current when the entry point launched (the C ABI launches on the CURRENT device: the caller guards);
    // a debug aid next to the launch counter, so that the host-side device guard can be tested on a one-GPU box
    int dev = -1;
    if (kernels > 0 && hipGetDevice(&dev) == hipSuccess) g_last_device.store(dev, std::memory_order_relaxed);
}

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
