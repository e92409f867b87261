// Dispatch knobs of the library, for MEASUREMENT builds only.
//
// The product library has no process-global mutable dispatch state: sige::tuning(key) is a compile-time constant there (the
// defaults below), the kernels a call runs depend on its arguments alone, and no `sige_hip_tuning_*` symbol is exported.
// Built with -DSIGE_HIP_TUNING (lib/libsige_hip_tuning.so: `python -m sige_amd.build --tuning`; what tools/, the bench
// sections that compare kernel forms and the tests that force a form load), the same call sites read an atomic table
// that `sige_hip_tuning_set(key, value)` writes.  One pair of entry points instead of the ten `*_force_*` setters of round 4.
#pragma once
#include "sige_hip.h"

#ifdef SIGE_HIP_TUNING
#include <atomic>
#endif

namespace sige {

constexpr int kTuningDefaults[SIGE_HIP_TUNE_COUNT] = {
    /* CONV_TILE_MT */ 0, /* CONV_TILE_NB */ 0, /* CONV_WAVES */ 0, /* CONV_LARGE_GRID_NB1 */ -1, /* CONV_KSPLIT */ 0,
    /* CONV_KSPLIT_SECOND_PASS */ 0, /* GATHER_ONE_TILE_ROWS */ 0, /* SCATTER_GATHER_FORM */ 0, /* SMALL_COUT_SCALAR */ 0,
    /* WIDE_KSPLIT */ 0, /* ATTENTION_FORM */ 0, /* TILE3_F16_TPW4_MIN */ -1, /* TILE3_F16_PAIR_MIN */ -1, /* TILE3_F16_SPARSE_MIN */ -1};

#ifdef SIGE_HIP_TUNING
extern std::atomic<int> g_tuning[SIGE_HIP_TUNE_COUNT];
inline int tuning(int key) { return g_tuning[key].load(std::memory_order_relaxed); }
#else
constexpr int tuning(int key) { return kTuningDefaults[key]; }
#endif

}  // namespace sige
