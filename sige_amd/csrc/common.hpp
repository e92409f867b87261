// Shared device/host helpers for libsige_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sige_hip.h"
#include "plan.hpp"
#include "tuning.hpp"

// One empty kernel per translation unit (build.py passes -DSIGE_TU_ID=<n>): sige_hip_preload() asks HIP for its attributes,
// which loads the code object of this translation unit -- and with it every kernel the unit holds -- on the current device.
namespace sige {
void preload_register(const void *host_kernel);
#ifdef SIGE_TU_ID
template <int ID> __global__ void tu_anchor_kernel() {}
template __global__ void tu_anchor_kernel<SIGE_TU_ID>();
static const int g_tu_anchor_registered = (preload_register((const void *)&tu_anchor_kernel<SIGE_TU_ID>), 0);
#endif
}  // namespace sige

namespace sige {

constexpr int kWave = 64;  // CDNA wavefront

// A broadcastable 4-D operand as element strides (0 on broadcast dims).
// data == nullptr  <=>  operand absent.
struct Bcast4 {
    const float *data;
    int sb, sc, sh, sw;
};

inline bool bcast_ok(const float *p, int b, int c, int h, int w, int B, int C, int H, int W) {
    if (!p) return true;
    auto ok = [](int d, int full) { return d == 1 || d == full; };
    return ok(b, B) && ok(c, C) && ok(h, H) && ok(w, W);
}

inline Bcast4 make_bcast(const float *p, int b, int c, int h, int w) {
    Bcast4 r{p, 0, 0, 0, 0};
    if (!p) return r;
    r.sw = (w > 1) ? 1 : 0;
    r.sh = (h > 1) ? w : 0;
    r.sc = (c > 1) ? h * w : 0;
    r.sb = (b > 1) ? c * h * w : 0;
    return r;
}

__device__ __forceinline__ float bcast_load(const Bcast4 &t, int b, int c, int h, int w) {
    return t.data[(size_t)b * t.sb + (size_t)c * t.sc + (size_t)h * t.sh + (size_t)w * t.sw];
}

// SiLU.  The reference evaluates z / (1.0 + exp(-z)) with the sum and quotient in double (sige/cpu/common_cpu.cpp:29-35);
// this is within ~3e-7 relative of it (the tests allow 1e-6) at a third of the instructions of expf() + an IEEE division --
// the standalone gather / scatter_gather kernels with a fused SiLU were bound by exactly those instructions.
//   e^-z = 2^t, t = -z * log2(e) as a (hi, lo) pair -- the rounding of the product alone would cost |z| * 6e-8 relative;
//   v_exp_f32 (1 ulp) on hi, first-order correction for lo;  1 / (1 + e): v_rcp_f32 (1 ulp) + one Newton step.
//   |z| is clamped to 88 for the exponent only: e^88 is finite, z * r then underflows / saturates as the exact form does.
__device__ __forceinline__ float swish(float z) {
    const float nz = fminf(fmaxf(-z, -88.0f), 88.0f);
    const float t = nz * 1.44269504088896341f;
    const float tl = __builtin_fmaf(nz, 1.44269504088896341f, -t) + nz * 1.92596299112661746e-8f;
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147180559945309f, e);
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return z * r;
}

// expf() + IEEE division (~1e-7 relative).  For an activation that is FOLLOWED by the affine (activation_first): a shift
// that cancels the activated value turns one ulp of it into an arbitrarily large relative error of the result, so that
// order keeps the form whose roundings the golden vectors were accepted with.
__device__ __forceinline__ float swish_exact(float z) { return z / (1.0f + expf(-z)); }

template <int ACT, bool EXACT = false>
__device__ __forceinline__ float activate(float z) {
    if (ACT == SIGE_HIP_ACT_SWISH) return EXACT ? swish_exact(z) : swish(z);
    return z;
}

// scale/shift + activation in the reference's order (gather.cpp:33-53):
// two separately rounded ops (the file is built with -ffp-contract=off).
template <int ACT, bool ACT_FIRST>
__device__ __forceinline__ float affine_act(float z, const Bcast4 &scale, const Bcast4 &shift,
                                            int b, int c, int h, int w) {
    if (!ACT_FIRST) {
        if (scale.data) z = bcast_load(scale, b, c, h, w) * z;
        if (shift.data) z = bcast_load(shift, b, c, h, w) + z;
    }
    z = activate<ACT, ACT_FIRST>(z);
    if (ACT_FIRST) {
        if (scale.data) z = bcast_load(scale, b, c, h, w) * z;
        if (shift.data) z = bcast_load(shift, b, c, h, w) + z;
    }
    return z;
}

// (api.hip) stacked edits: E edited images stacked along H in every tensor a launch sees (sige_hip_set_edit_batch).  0 = off, else
// log2 of the height of ONE image at a launch whose tensors are H rows tall; -1 = the launch cannot honour the mode.
int stacked_shift(int H);

// (api.hip) every entry point reports how many kernels it launched: sige_hip_launch_count()
void note_launches(int kernels);

inline int launch_status(int kernels = 1) {
    note_launches(kernels);
    return hipGetLastError() == hipSuccess ? SIGE_HIP_OK : SIGE_HIP_ELAUNCH;
}

// (block_conv.hip) tickets of the in-launch K-split finish: `blocks` zeroed ints that stay valid while the launch runs
// (or, during a hipGraph capture, for the life of the graph); nullptr = unavailable, the caller must not split
int32_t *split_tickets(hipStream_t st, long blocks);

// All-reduce over the 16 lanes of a DPP row (lanes 16r .. 16r+15 of a wave) on the VALU's data-parallel-primitive path: quad
// swaps (xor 1, xor 2), then the mirror of a half row and of the row -- four VALU operations, every lane ends with the same
// bits (each step combines two disjoint, complete groups symmetrically).  __shfl_xor compiles to ds_bpermute_b32: a trip
// through the LDS pipeline per step, ~8 dependent trips per softmax row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));  // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

// One dword of every 64-byte line of the kernel-argument segment, read (and waited for) at the top of a kernel with a large argument
// struct.  The compiler fetches kernel arguments where it first needs them: a conv kernel reads its ~400-byte struct in three to
// five dependent phases (grid decomposition -> pointers -> epilogue options ...), each one a scalar-cache miss of its own; after this
// call every later fetch hits the scalar cache, and the one miss it pays is the one the first phase would have paid anyway.
// `-DSIGE_NO_KERNARG_TOUCH` removes it (the A/B build).  BYTES must not exceed the kernel's argument segment (the last dword read is at
// 64 * ((BYTES - 1) / 64)): a kernel that uses no hidden argument has none, its segment ends with its last explicit argument.
template <int BYTES>
__device__ __forceinline__ void kernarg_touch() {
#ifndef SIGE_NO_KERNARG_TOUCH
    typedef const __attribute__((address_space(4))) unsigned *kptr_t;
    kptr_t k = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) acc ^= k[o / 4];
    asm volatile("" ::"s"(acc));
#endif
}

// Grid-stride loop over `units` work items of a T-thread workgroup, with the item index a 32-bit unsigned whenever the count allows:
// the kernels that use it split the index by run-time divisors (channel quads, pixels of a tile, tiles of an image), and a 64-bit
// division by a run-time value is a branchy ~130-instruction routine -- four of them were most of the instructions of a scatter
// launch (round 6).  `body(u)`: `return` where a loop body would `continue`.
template <int T, typename F>
__device__ __forceinline__ void for_units(long units, F &&body) {
    if (units <= 0x7fffffffL) {
        const unsigned n = (unsigned)units, step = gridDim.x * T;
        for (unsigned u = blockIdx.x * T + threadIdx.x; u < n; u += step) body(u);
    } else {
        for (long u = (long)blockIdx.x * T + threadIdx.x; u < units; u += (long)gridDim.x * T) body(u);
    }
}

// 16-byte store of a kernel's OUTPUT (an epilogue's result that only later launches read).  Write-through (`sc1`): the bytes go to
// the memory side while the kernel still runs instead of staying dirty in the XCD's write-back L2 until the end-of-kernel release
// writes them back.  Measured (tools/probe/store_probe.hip, profiles/r6_store_probe.json: dependent launches each writing B bytes,
// us per launch incl. the 1.7 us boundary): 1 MB 1.73 vs 1.93 plain, 4 MB 1.89 vs 2.29, 16 MB 3.47 vs 4.62, 32 MB 5.75 vs 7.15.
// Not for data the SAME launch reads back (the line leaves this XCD's L2), nor for 4-byte stores (one fabric write each).
// -DSIGE_PLAIN_STORES: plain stores everywhere (the A/B build).
__device__ __forceinline__ void store_out4(float *p, float4 v) {
#ifdef SIGE_PLAIN_STORES
    *reinterpret_cast<float4 *>(p) = v;
#else
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
#endif
}

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace sige
