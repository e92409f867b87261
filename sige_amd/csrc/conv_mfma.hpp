// Stacked-block convolution on the fp32 matrix cores of gfx950 (kernel template).
//
//   x tiles [T,Cin,R,S] (*) w [Cout,Cin,k,k] -> out tiles [T,Cout,Ro,So], padding 0
//
// as ONE LDS-tiled implicit GEMM, M = T*Ro*So output pixels, N = Cout,
// K = Cin*k*k, exact fp32 products and fp32 accumulation
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32) -- optionally with the
// producer of the tiles (Gather: sige/cpu/gather.cpp:4-58, ScatterGather:
// sige/cpu/scatter_gather.cpp:5-56) fused into the staging path, so that the
// [T,Cin,R,S] tensor the reference materialises between gather_cuda and
// F.conv2d (sige/nn/gather.py:80-89 -> sige/nn/base.py:88-89) never exists.
//
// Work decomposition (MI355X: 256 CUs x 4 SIMDs, f32 MFMA = 64 FLOP/clk/SIMD,
// saturated by ONE wave per SIMD):
//   workgroup = 256 lanes = 4 waves -> one MT x (NB*MT) output block, full K.
//   The four waves split K (each owns a quarter of every channel chunk) and
//   reduce through LDS at the end, so even a conv with 18 active tiles spreads
//   over >= 256 workgroups.  MT in {16, 32} and NB in {1, 2} are picked per
//   launch so that the grid still covers the chip.
//   A (im2col of the input tiles) is never materialised: raw input tiles of a
//   channel chunk live in LDS as [tile][channel][R][S]; an MFMA lane reads its
//   A element with one ds_read_b32 at a compile-time offset from a per-lane base.
//   B (weights) is pre-packed once per weight tensor in exactly the order the
//   lanes consume it: every B access is a coalesced 16-byte-per-lane load
//   straight into registers (no reuse across the K-split waves -> no LDS hop).
//
// Software pipeline (all index math hoisted out of the K loop):
//   every lane owns NS fixed staging slots (tile, channel-in-chunk, pixel).  The
//   raw values of chunk c+1 sit in registers while the MFMAs of chunk c run;
//   between MFMA groups each slot is finished (cached-GroupNorm affine + SiLU,
//   rounded like the reference: scale, then shift, then activation), written to
//   the other LDS buffer, and its register immediately re-issued as the load of
//   chunk c+2.  B registers are re-issued the same way right after their last
//   MFMA.  So every global load has a full chunk of MFMA time to land and the
//   VALU work of the staging is spread between matrix instructions.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace sige {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

enum { SRC_TILES = 0, SRC_GATHER = 1, SRC_SCATTER_GATHER = 2 };
enum { DST_TILES = 0, DST_NCHW = 1 };  // DST_NCHW = "into a full tensor" (in the launch's layout)

template <int MT_> struct Mfma;
template <> struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int REGS = 16;
    __device__ static __forceinline__ acc_t op(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int REGS = 4;
    __device__ static __forceinline__ acc_t op(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// fp16 operands (8 k-values per lane and instruction), fp32 accumulation: v_mfma_f32_32x32x16_f16 / 16x16x32_f16,
// 16x the rate of the f32-input forms above
template <int MT_> struct MfmaH;
template <> struct MfmaH<32> {
    using acc_t = f32x16;
    static constexpr int REGS = 16;
    __device__ static __forceinline__ acc_t op(f16x8 a, f16x8 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct MfmaH<16> {
    using acc_t = f32x4;
    static constexpr int REGS = 4;
    __device__ static __forceinline__ acc_t op(f16x8 a, f16x8 b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <int KH, int STR, int R_, int MT_>
struct ConvGeo {
    static constexpr bool F16 = false, X3 = false;
    static constexpr int K = KH, S = STR, R = R_, MT = MT_;
    static constexpr int KK = KH * KH;
    static constexpr int RS = R_ * R_;
    static constexpr int RO = (R_ - KH) / STR + 1;
    static constexpr int PX = RO * RO;                 // output pixels per tile: 16 or 4
    static constexpr int NL = 64 / MT_;                // k values per MFMA = lane groups (2 or 4)
    static constexpr int TPB = MT_ / PX;               // tiles per M block
    static constexpr int CW = (KK == 1 ? 16 : 4) * NL; // channels per wave per chunk
    static constexpr int CC = 4 * CW;                  // channels per LDS chunk
    static constexpr int L = (CW / NL) * KK;           // MFMA k-steps per wave per chunk (36 or 16)
    static constexpr int F = L / 4;                    // 16-byte weight loads per lane per chunk and N sub-block
    static constexpr int TILE_FLOATS = CC * RS;        // floats of one staged tile
    static constexpr int BUF = TPB * TILE_FLOATS;      // floats per LDS stage, laid out [tile][channel][R][S]
    static constexpr int RED = MT_ + 4;                // padded pixel stride of the reduction buffer
    static_assert(L % 4 == 0, "wave slice must be a whole number of float4 weight loads");
    static_assert(MT_ % PX == 0 && TPB >= 1, "tile pixels must divide the M block");
};

// The same tile geometry computed on the fp16 matrix cores ("f16 compute"): tensors stay fp32 in HBM; the staging path
// finishes a value in fp32 (cached GroupNorm affine + SiLU), rounds it to fp16 (RNE) into LDS, the weights are packed
// as fp16, products are exact in fp32 and accumulate in fp32.  One MFMA contracts SLAB = 8 * NL channels at ONE tap:
// an A operand is 8 consecutive channels of a pixel = one ds_read_b128 of the channels-last stage.
template <int KH, int STR, int R_, int MT_>
struct ConvGeoH {
    static constexpr bool F16 = true, X3 = false;
    static constexpr int K = KH, S = STR, R = R_, MT = MT_;
    static constexpr int KK = KH * KH;
    static constexpr int RS = R_ * R_;
    static constexpr int RO = (R_ - KH) / STR + 1;
    static constexpr int PX = RO * RO;
    static constexpr int NL = 64 / MT_;                // lane groups of an MFMA (2 or 4)
    static constexpr int KP = 8;                       // k values (channels) per lane and MFMA
    static constexpr int SLAB = NL * KP;               // channels one MFMA contracts (16 or 32)
    static constexpr int NSLAB = KK == 1 ? 2 : 1;      // slabs per wave per chunk
    static constexpr int TPB = MT_ / PX;
    static constexpr int CW = NSLAB * SLAB;            // channels per wave per chunk
    static constexpr int CC = 4 * CW;                  // channels per LDS chunk (4 waves)
    static constexpr int L = NSLAB * KK;               // MFMAs per wave per chunk (9 or 2)
    static constexpr int F = L;                        // 16-byte weight loads per lane per chunk and N sub-block
    static constexpr int TILE_FLOATS = CC * RS;
    static constexpr int BUF = TPB * TILE_FLOATS;
    static constexpr int RED = MT_ + 4;
    static_assert(MT_ % PX == 0 && TPB >= 1, "tile pixels must divide the M block");
};

// The same on SPLIT fp16 operands ("f16x3", round 3): every fp32 value -- staged activation or weight -- is carried as an fp16
// pair hi = fp16(v), lo = fp16(v - hi); the three products hi*hi + lo*hi + hi*lo accumulate in fp32 (22-bit operands, the
// dropped lo*lo term is 2^-22 relative): fp32-level results at a third of the fp16 matrix rate = 5.3x the f32-input rate.
// The LDS stage holds two planes per pixel row ([hi CC halves][lo CC halves]), the packed weights two 16-byte registers per
// k-step (hi, lo), pre-scaled by a power of two so that their lo parts are normal fp16 numbers (ConvArgs::wscale_ptr).
template <int KH, int STR, int R_, int MT_>
struct ConvGeoX : ConvGeoH<KH, STR, R_, MT_> {
    static constexpr bool X3 = true;
    static constexpr int F = 2 * ConvGeoH<KH, STR, R_, MT_>::L;  // 16-byte weight loads per lane per chunk: (hi, lo) per k-step
};

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) -- a guaranteed full unroll
// (the MFMA body is too large for `#pragma unroll` to accept)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct ConvArgs {
    const float *x;       // TILES: [T,Cin,R,S] | GATHER: [B,Csplit,H,W] | SCATTER_GATHER: conv-1 tiles [B*N,Cin,Rx,Sx]
    const float *y;       // SCATTER_GATHER: cached full tensor [B,Cin,H,W]
    const int32_t *idx;   // gather modes / DST_NCHW: [N,2]
    const int32_t *map;   // SCATTER_GATHER: [H,W,3]
    const float *packed;
    const float *bias;
    float *out;
    int T, Cin, Cout, nchunks;  // nchunks: channel chunks of THIS launch (W * CW channels each)
    int nblk;                   // (chunk, wave) blocks per output-channel group in `packed`
    int B, N, H, W;
    int RxSx, Sx;
    const float *scale, *shift;  // per-(batch, channel) affine of the gather modes (same broadcast shape)
    int aff_sb, aff_sc;          // element strides of scale / shift over (batch, channel)
    // GATHER: channels [0,Csplit) come from x, [Csplit,Cin) from x2 (a fused torch.cat; B == 1, Csplit % CC == 0)
    const float *x2;
    int Csplit;
    // DST_NCHW: write the output tiles straight into a full tensor [B,Cout,Ho,Wo] at
    // ((off+idx)/stride), clipped, + residual[B,Cout,Ho,Wo] (dense layers: all tiles active)
    const float *residual;
    int Ho, Wo, offH, offW, strH, strW;
    // grid decomposition: blockIdx.x -> (mb, ng); ng_fast = 2: XCD-grouped by M block (see the kernel); ng_fast = 1: consecutive workgroups (= consecutive
    // XCDs) take different output-channel blocks, so each XCD's L2 streams 1/8 of the weights
    int mbk, ngk, ng_fast;
    // cross-workgroup K split: grid.y = ksplit workgroups share one output block, `out` is a workspace of
    // ksplit copies of the output (split_stride floats apart)
    int ksplit, chunks_per_split;
    size_t split_stride;
    // optional epilogue: out = act(oscale[co] * (conv + bias + residual) + oshift[co]) -- the NEXT layer's cached
    // GroupNorm affine + SiLU applied once per output element by the producer, so that the consumer stages raw
    // values (no VALU work between its MFMAs, and none repeated per output-channel block)
    const float *oscale, *oshift;
    int oact;
    // DST full, block residual (ScatterWithBlockResidual fused into the epilogue, scatter.cpp:41-68): `residual` is the
    // cached shortcut tensor y1; where a shortcut tile covers the pixel, out += x1 - y1 (x1 tiles [B*N1,R1,S1,Cout])
    const float *x1;
    const int32_t *table1;
    int gW1, N1, R1, S1;
    int up;          // GATHER: 1 = gather from the half-resolution tensor as if it were nearest-upsampled x2
    float *ws;       // host side: workspace for the partial outputs (nullptr / ksplit_max <= 1: no K split)
    int ksplit_max;  // host side: how many output copies `ws` holds
    // K split finished inside the launch: one ticket per output block (zero between launches), and the real destination
    int32_t *counters;
    float *fout;
    // DST full: up to two "twin" outputs of the same launch, twin_k[addr] = SiLU(tscale_k[co] * v + tshift_k[co]) with v the value
    // the primary output receives before its own out-affine -- the activated input of a CONSUMER's conv1 (its cached GroupNorm
    // affine + SiLU applied once by the producer instead of once per output-channel block by the consumer's staging path)
    float *twin0, *twin1;
    const float *tscale0, *tshift0, *tscale1, *tshift1;
    // split fp16 operands (ConvGeoX): the packed weights are w * 2^S; *wscale_ptr = 2^-S (written by the pack kernels)
    const float *wscale_ptr;
    // fp16-stored caches (the "_c16" entry points): `y` (SCATTER_GATHER: the cached tensor, or its
    // activated copy) holds halves -- a compile-time form of the kernel (Y16), this field picks it on the host; `residual`
    // holds halves (the cached shortcut tensor of a fused ScatterWithBlockResidual) -- a run-time branch of the epilogue
    int y_f16, res_f16;
    // stacked edits (sige_hip_set_edit_batch): log2 of one image's height when E images are stacked along H (0 = off): a
    // tile's halo rows beyond ITS image are zero padding (channels-last gather / scatter_gather staging)
    int hp_shift;
#ifdef SIGE_CONV_PROBE
    unsigned long long *probe;  // tools/conv_phase_probe.py build only: 8 timestamps per workgroup
#endif
};

// Phase timestamps (s_memtime) of workgroup 0..4095, lane 0 -- compiled in only for the measurement build of
// tools/conv_phase_probe.py (python -m sige_amd.build --probe); the product library contains none of this.
#ifdef SIGE_CONV_PROBE
#define SIGE_PROBE(k)                                                                                       \
    do {                                                                                                    \
        if (a.probe && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096)                            \
            a.probe[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();                                   \
    } while (0)
#else
#define SIGE_PROBE(k)
#endif
// Ablations for the probe build (timing experiments only, results are WRONG): a bit mask of what the K loop leaves out.
//   1 weight loads | 2 activation (slot) loads | 4 slot finish + LDS store | 8 the per-chunk barrier | 16 the A operand LDS reads
#if defined(SIGE_CONV_PROBE) && defined(SIGE_ABL)
#define SIGE_ABL_HAS(bit) (((SIGE_ABL) & (bit)) != 0)
#else
#define SIGE_ABL_HAS(bit) false
#endif

// SiLU for the fused staging path: v_exp_f32 + v_rcp_f32 (each <= 1 ulp) instead of
// expf + IEEE division; |relative error| ~1e-6 (the standalone gather keeps the
// exact form).  z -> -inf: e = +inf, r = 0, result -0 (true value -0).
__device__ __forceinline__ float swish_fast(float z) {
    const float e = __builtin_amdgcn_exp2f(z * -1.44269504088896341f);
    return z * __builtin_amdgcn_rcpf(1.0f + e);
}

// MODE: what the staging path applies to a gathered value
//   0 raw copy | 1 scale*x + shift, then SiLU | 2 scale*x + shift   (both scale and shift present)
enum { MODE_RAW = 0, MODE_AFFINE_SWISH = 1, MODE_AFFINE = 2 };

// ---- buffer addressing -----------------------------------------------------------
// Measured on MI355X (tools/probe/mfma_probe.hip): the f32-input MFMA shares its issue
// slots with f32 VALU work of the SAME wave -- every VALU instruction between two
// MFMAs adds ~4 cycles to the 32 (16x16x4) / 64 (32x32x2) cycle matrix instruction,
// while LDS and memory instructions overlap for free.  So the K loop below is built
// to contain (almost) no VALU instructions: all addressing is a per-slot constant
// 32-bit byte offset into a buffer descriptor that is advanced per channel chunk with
// SCALAR arithmetic, and every "is this element real data?" test is the hardware's
// buffer range check (out-of-range lanes read 0) instead of compare + select.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOOB = 0x80000000u;  // a byte offset no descriptor contains (tensors are < 2 GiB: host side)

__device__ __forceinline__ rsrc_t make_rsrc(const float *base, long elem_off, long total_elems) {
    // window [elem_off, total_elems) of the tensor at `base`; empty past the end.
    // Every caller passes wave-uniform values, but the arithmetic behind them contains 64-bit multiplies / compares and
    // divisions by runtime values, which LLVM evaluates on the VECTOR unit; a descriptor assembled from VGPRs then costs a
    // waterfall loop (4 v_readfirstlane + 2 v_cmp + exec juggling) in front of EVERY buffer load that uses it -- measured in
    // the ISA of the K loop: 18 loops per 72 MFMAs.  Both words are 32-bit here (tensors are < 2 GiB: host side) and are
    // pinned to scalar registers once per descriptor.
    const int off = __builtin_amdgcn_readfirstlane((int)elem_off);
    int bytes = __builtin_amdgcn_readfirstlane((int)((total_elems - elem_off) * 4));
    bytes = bytes < 0 ? 0 : bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base) + off, 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float4 buf_f32x4(rsrc_t r, unsigned byte_off, int soff) {
    // (bit_cast the whole vector: element-wise bit_casts of the builtin's result are narrowed
    // by hipcc 7.2 to a single dword load of element 0)
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soff, 0));
    return make_float4(v[0], v[1], v[2], v[3]);
}

// fp16 storage: a window over a tensor of halves, and 4 consecutive halves widened to fp32 (exact)
__device__ __forceinline__ rsrc_t make_rsrc_h(const float *base_as_halves, long elem_off, long total_elems) {
    const int off = __builtin_amdgcn_readfirstlane((int)elem_off);
    int bytes = __builtin_amdgcn_readfirstlane((int)((total_elems - elem_off) * 2));
    bytes = bytes < 0 ? 0 : bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<_Float16 *>(const_cast<float *>(base_as_halves)) + off, 0, bytes, 0x00020000);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 buf_h4(rsrc_t r, unsigned byte_off, int soff) {
    const f16x4 h = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, soff, 0));
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ float4 ld_f32x4_or_h4(const float *base, size_t elem, bool halves) {
    if (halves) {
        const f16x4 h = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(base) + elem);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    return *reinterpret_cast<const float4 *>(base + elem);
}

// The same load with the conversion LEFT OUT (halves: the 8 bytes as they lie, in .x / .y) and done where the value is used: a
// residual requested with the prologue's loads must not be converted there -- the conversion is its first use, and the wait in
// front of it drains every load of the prologue (s_waitcnt vmcnt(0): the weights of two chunks, the index entries, the tables).
__device__ __forceinline__ float4 ld_raw_f32x4_or_h4(const float *base, size_t elem, bool halves) {
    if (halves) {
        const float2 h = *reinterpret_cast<const float2 *>(reinterpret_cast<const _Float16 *>(base) + elem);
        return make_float4(h.x, h.y, 0.f, 0.f);
    }
    return *reinterpret_cast<const float4 *>(base + elem);
}
__device__ __forceinline__ float4 cvt_raw_f32x4_or_h4(float4 raw, bool halves) {
    if (halves) {
        const float2 r2 = make_float2(raw.x, raw.y);
        const f16x4 h = __builtin_bit_cast(f16x4, r2);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    return raw;
}

// Memory layout of every activation / tile tensor a launch touches:
//   NCHW  [B,C,H,W] / tiles [T,C,R,S]            -- the reference's layout (torch contiguous)
//   NHWC  [B,H,W,C] / tiles [T,R,S,C]            -- torch channels_last
// The arithmetic is identical; only the staging addresses and the epilogue differ.  In NCHW
// a gathered 6x6 window is 6 segments of 24 bytes per channel (one cache line each: the
// texture-addresser cost of such a wave-load is ~100 cycles, tools/probe/mfma_probe2.hip);
// in NHWC a pixel's channels are contiguous, so the same chunk is staged with 16-byte
// fully coalesced loads -- 25x fewer address-processing cycles per byte.
enum { LAYOUT_NCHW = 0, LAYOUT_NHWC = 1 };

// W = waves per workgroup (4 or 8).  All W waves split K: a channel chunk is W * CW channels.
// W = 8 (512 lanes) puts TWO waves on every SIMD of the CU running the workgroup: measured
// (tools/conv_floor.py) the K loop sustains ~88 % of the f32 MFMA rate with two waves per SIMD
// against ~50-65 % with one, because one wave's waits are covered by the other's MFMAs -- the
// form for grids that cannot give every CU two workgroups.  The packed weight order
// [ng][chunk][wave][f][lane] read with chunk' = chunk / 2, wave' = 4 * (chunk % 2) + wave is
// exactly the 8-wave order, so both forms share one packed tensor (chunk count padded to even).
// 16 bytes to / from the device's coherence point (two relaxed agent-scope 8-byte atomics: visible to every XCD without
// cache maintenance) -- the partial sums of the in-kernel K-split finish
__device__ __forceinline__ void coherent_store(float *p, float4 v) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    const unsigned long long lo = (unsigned long long)__builtin_bit_cast(unsigned, v.x) | ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32);
    const unsigned long long hi = (unsigned long long)__builtin_bit_cast(unsigned, v.z) | ((unsigned long long)__builtin_bit_cast(unsigned, v.w) << 32);
    __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 coherent_load(const float *p) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<float *>(p));
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                       __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32)));
}

// LDS floats of one workgroup (the kernels below declare the array; the body only receives the pointer, so that two
// bodies sharing a launch -- conv_pair_kernel -- share one allocation)
template <typename G, int NB, int MODE, int LAYOUT, int W>
__host__ __device__ constexpr int conv_lds_floats() {
    constexpr int CCk = W * G::CW;
    constexpr bool NHWC = LAYOUT == LAYOUT_NHWC;
    constexpr int LDC = G::F16 ? (G::X3 ? 2 * CCk + 8 : CCk + 8) : CCk + 4;
    constexpr int STAGE = NHWC ? (G::F16 ? G::TPB * G::RS * LDC / 2 : G::TPB * G::RS * LDC) : G::TPB * CCk * G::RS;
    constexpr int TABF = MODE != MODE_RAW ? 2 * (CCk + 4) : 0;
    return cmax(2 * STAGE + 2 * TABF, W * NB * G::MT * (G::MT + 4));
}

// The whole launch of one workgroup (bx, by) = what blockIdx would be in a launch of its own.
template <typename G, int NB, int SRC, int MODE, int DST, int LAYOUT, int W, bool Y16 = false>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs &a, const int bx, const int by, float *const smem) {
    static_assert(!Y16 || (SRC == SRC_SCATTER_GATHER && LAYOUT == LAYOUT_NHWC), "fp16-stored cache: the channels-last scatter_gather source");
    constexpr unsigned YB = Y16 ? 2u : 4u;       // bytes per element of the cached tensor `y`
    constexpr bool F16 = G::F16;                 // fp16 operands in LDS / registers (ConvGeoH)
    using M = std::conditional_t<F16, MfmaH<G::MT>, Mfma<G::MT>>;
    static_assert(!F16 || LAYOUT == LAYOUT_NHWC, "the f16-compute kernels are channels-last");
    constexpr int NT = 64 * W;                   // lanes per workgroup
    constexpr int CCk = W * G::CW;               // channels per LDS chunk
    constexpr int TILEF = CCk * G::RS;           // floats of one staged tile (NCHW stage)
    constexpr int BUFk = G::TPB * TILEF;
    constexpr bool NHWC = LAYOUT == LAYOUT_NHWC;
    constexpr bool VEC = NHWC || SRC == SRC_TILES;            // staging slots are float4 units
#ifndef SIGE_CONV_NACC32
#define SIGE_CONV_NACC32 1
#endif
    // accumulators per N sub-block that consecutive k-steps alternate between: a chain of MFMAs on ONE accumulator pays for
    // every instruction issued between two of them (16x16x4: 40-cycle dependent latency vs 32-cycle issue)
    constexpr int NACC = NB == 1 ? ((G::MT == 16 || F16) ? 2 : SIGE_CONV_NACC32) : 1;
    constexpr bool AFF = MODE != MODE_RAW;
    // LDS stage of one channel chunk: NCHW [tile][channel][R][S]; NHWC [tile][R][S][LDC] (LDC = CC + 4 pad)
    //   (f16 compute: the stage holds HALVES, row = CC + 8 halves; STAGE stays in floats)
    constexpr bool X3 = G::X3;                   // split fp16 operands (ConvGeoX): two planes in LDS, (hi, lo) weight registers
    constexpr int LDC = F16 ? (X3 ? 2 * CCk + 8 : CCk + 8) : CCk + 4;
    constexpr int STAGE = NHWC ? (F16 ? G::TPB * G::RS * LDC / 2 : G::TPB * G::RS * LDC) : BUFk;
    constexpr int TROW = CCk + 4;                            // table row: CC channels + 4 zeros
    constexpr int TABF = AFF ? 2 * TROW : 0;                   // scale row | shift row
    constexpr int RP = G::MT + 4;                              // padded row of the reduction buffer
    constexpr int LDS_FLOATS = cmax(2 * STAGE + 2 * TABF, W * NB * G::MT * RP);
    static_assert(LDS_FLOATS == conv_lds_floats<G, NB, MODE, LAYOUT, W>(), "conv_lds_floats() out of step with the body");
    float *const tab = smem + 2 * STAGE;

    SIGE_PROBE(0);  // entry
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int kq = lane / G::MT, j = lane % G::MT;
    int mb, ng;
    if (a.ng_fast == 1) { ng = bx % a.ngk; mb = bx / a.ngk; }
    else if (a.ng_fast == 2) {
        // activation-dominant launches: all output-channel blocks of one M block on ONE XCD (workgroup b runs on XCD
        // b % 8), so the M block's input tiles are fetched into one L2 instead of into up to ngk of them; the grid is
        // padded to 8 * ceil(mbk / 8) * ngk and the surplus workgroups leave here (before any barrier)
        const int jj = bx >> 3;
        mb = (bx & 7) + 8 * (jj / a.ngk);
        ng = jj % a.ngk;
        if (mb >= a.mbk) return;
    } else { mb = bx % a.mbk; ng = bx / a.mbk; }
    SIGE_PROBE(6);  // first kernel arguments are in registers
    const int Cin = a.Cin;
    const int HW = (SRC == SRC_GATHER) ? (a.H >> a.up) * (a.W >> a.up) : a.H * a.W;  // pixels of the SOURCE tensor
    // cross-workgroup K split (deep-K, small-M layers): by owns chunks [first, last]
    const int split = by;
    const int first = split * a.chunks_per_split;
    const int last = min(a.nchunks, first + a.chunks_per_split) - 1;

    // ---- B: F float4 per lane per chunk and N sub-block, contiguous per (ng, chunk, wave) ----
    const int ngtot = (a.Cout + G::MT - 1) / G::MT;
    const long packed_floats_total = (long)ngtot * a.nblk * G::F * 64 * 4;  // nblk = (4-wave chunks, padded to even) * 4
    int gsel[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) gsel[nb] = min(ng * NB + nb, ngtot - 1);  // past-the-end sub-blocks re-read the last one; stores are masked
    rsrc_t r_b[NB];
    auto set_b_chunk = [&](int chunk) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            r_b[nb] = make_rsrc(a.packed, ((long)gsel[nb] * a.nblk + chunk * W + wave) * G::F * 64 * 4, packed_floats_total);
    };
    float4 bset[2][NB][G::F];
    auto b_load = [&](float4 &dst, int nb, int f) { dst = buf_f32x4(r_b[nb], lane * 16, f * 1024); };

    // The weight loads of the first two chunks do not depend on the index list; they are issued right behind the
    // (tiny) index / table loads below, so that everything the prologue needs is in flight in one round trip.
    auto b_prologue = [&]() {
        set_b_chunk(first);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int f = 0; f < G::F; ++f) b_load(bset[0][nb][f], nb, f);
        set_b_chunk(min(first + 1, last));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int f = 0; f < G::F; ++f) b_load(bset[1][nb][f], nb, f);
    };
    if (!NHWC) b_prologue();

    // ---- staging slots ------------------------------------------------------
    // One slot = one LDS unit (a float, or a float4 when VEC) this lane fills for every chunk.
    // s_off / s_off2: byte offsets at channel chunk 0 into the source window (kOOB = zero fill):
    //   TILES           s_off  into the tile slab
    //   GATHER          s_off  into x, s_off2 into x2 (NHWC only: the two have different pitches)
    //   SCATTER_GATHER  s_off  into the conv-1 tiles, s_off2 into the cached tensor y (one of them kOOB)
    constexpr int UNITS = VEC ? (NHWC ? G::TPB * G::RS * CCk / 4 : BUFk / 4) : BUFk;
    constexpr int NS = (UNITS + NT - 1) / NT;
    constexpr bool TWO = SRC == SRC_SCATTER_GATHER || (NHWC && SRC == SRC_GATHER);
    float st_z[2][VEC ? 1 : NS], st_z2[2][(!VEC && TWO) ? NS : 1];
    float4 st_q[2][VEC ? NS : 1], st_q2[2][(VEC && SRC == SRC_SCATTER_GATHER) ? NS : 1];
    unsigned s_off[NS], s_off2[TWO ? NS : 1];
    int s_tab[AFF ? NS : 1];   // float index of the slot's entry in a table row (the zero entries for zero fills)
    int s_dst[NS];             // LDS float index of the unit inside a stage
    int s_cl[VEC ? NS : 1];    // first channel of the unit inside the chunk (partial last chunk test)

    // (scale, shift) of channel chunk `chunk` for table entry (tid mod CC); entries past Cin are 0
    const int trow = tid % CCk;
    const int tab_b0 = AFF ? ((mb * G::TPB) / a.N) * a.aff_sb : 0;  // (a per-batch affine needs one batch per M block: host side)
    auto tab_fetch = [&](int chunk, float &sc, float &sh) {
        const int c = chunk * CCk + trow;
        const int cc = c < Cin ? c : 0;
        const float vs = a.scale[tab_b0 + cc * a.aff_sc], vh = a.shift[tab_b0 + cc * a.aff_sc];
        sc = c < Cin ? vs : 0.f;
        sh = c < Cin ? vh : 0.f;
    };
    float t_sc = 0.f, t_sh = 0.f, t_sc1 = 0.f, t_sh1 = 0.f;

    // Output units of the epilogue (one float4 = 4 consecutive output channels of one pixel per lane and step).
    // Channels-last, full-tensor destination: where each of this lane's (at most EU) units goes, its bias and its
    // residual are fetched HERE, with the prologue's loads -- left to the epilogue, tile origin -> address -> residual
    // -> store are two more dependent memory round trips at the end of every launch.
    constexpr int UNITS_NB = G::MT * G::MT / 4;         // float4 units per N sub-block
    constexpr int OUT_UNITS = NB * UNITS_NB;
    constexpr int EU = (OUT_UNITS + NT - 1) / NT;
    // (4-wave workgroups only: with 8 waves the register budget per lane is 256 and the loop needs it)
    constexpr bool EPRE = NHWC && DST != DST_TILES && (W == 4 || NB == 1);
    int e_h[EPRE ? EU : 1], e_w[EPRE ? EU : 1];          // tile origin, then the unit's output pixel
    size_t e_q[EPRE ? EU : 1];                            // element offset of the unit in the output tensor
    bool e_in[EPRE ? EU : 1];                             // the unit exists and its pixel is inside the output
    float4 e_res[EPRE ? EU : 1];
    constexpr bool EPV = NHWC && (W == 4 || NB == 1);                 // per-channel epilogue vectors (bias, out_affine) fetched up front
    float4 e_bias[EPV ? EU : 1], e_os[EPV ? EU : 1], e_oh[EPV ? EU : 1];
    // (K split with a second launch: partial sums only, bias / residual in the second pass; with in-kernel finish any
    //  workgroup may turn out to be the one that runs the epilogue)
    const bool e_first_pass = a.ksplit <= 1 || a.counters != nullptr;
    float wsc = 1.0f;  // split fp16 operands: the weights were packed as w * 2^S, the sums come out times 2^S
    if constexpr (X3) wsc = *a.wscale_ptr;

    // the epilogue's output units (pixel, address, residual) from the tile origins requested above.  Called BEHIND the prologue's
    // activation loads: ahead of them (where it was until round 6) it put 370 instructions between the index values arriving and
    // those loads going out (forward 1.345 -> 1.341 ms at 1.2 %, 1.697 -> 1.684 at 5 %, f16 -0.9 %: profiles/r6an_*)
    auto epre_setup = [&]() {
        if constexpr (EPRE)
            static_for<0, EU>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                const int o = tid + k * NT;
                const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
                const int n4 = o1 % (G::MT / 4), m = o1 / (G::MT / 4);
                const int pxo = m % G::PX;
                const int t = mb * G::TPB + m / G::PX, co = (ng * NB + nb) * G::MT + 4 * n4;
                const int b = min(t, a.T - 1) / a.N;
                const int h = (a.offH + e_h[k]) / a.strH + pxo / G::RO, w = (a.offW + e_w[k]) / a.strW + pxo % G::RO;
                const bool in = o < OUT_UNITS && t < a.T && co < a.Cout && h >= 0 && h < a.Ho && w >= 0 && w < a.Wo;
                e_h[k] = h; e_w[k] = w; e_in[k] = in;
                e_q[k] = in ? (((size_t)b * a.Ho + h) * a.Wo + w) * a.Cout + co : 0;  // (dead units: a valid address, never stored)
                e_res[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e_first_pass && a.residual) e_res[k] = ld_raw_f32x4_or_h4(a.residual, e_q[k], a.res_f16 != 0);  // (converted at its use)
            });
    };
    if constexpr (NHWC) {
        // Channels-last: the slot set-up in load-batched, branch-free form.  Written slot by slot (index load ->
        // bounds test -> offset, as in the NCHW branch below) every slot costs a dependent memory round trip behind the
        // weight loads -- 3 to 6 of them before the first activation load can be issued (measured in the ISA:
        // global_load / s_waitcnt vmcnt(0) per slot).  Here: (1) pure arithmetic, (2) ALL tile origins, then the
        // affine table entries, then the weights of two chunks in flight together, (3) offsets with selects;
        // SCATTER_GATHER adds one more batched round trip for the scatter-map entries.
        constexpr int UPT = G::RS * CCk / 4, UPP = CCk / 4;  // units per tile / per pixel
        int z_t[NS], z_p[NS], z_c[NS], z_h[NS], z_w[NS], z_b[NS];
        bool z_live[NS];
        static_for<0, NS>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            const int v = tid + NT * i;
            const int t_l = v / UPT;
            const int p = (v - t_l * UPT) / UPP;
            const int c_l = ((v - t_l * UPT) - p * UPP) * 4;
            s_dst[i] = (t_l * G::RS + p) * LDC + c_l;
            s_cl[i] = c_l;
            z_c[i] = c_l; z_p[i] = p;
            z_t[i] = mb * G::TPB + t_l;
            z_live[i] = v < UNITS && z_t[i] < a.T;
        });
        if (SRC != SRC_TILES) {
            static_for<0, NS>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                const int tt = min(z_t[i], a.T - 1);  // (clamped: always a valid address; dead slots are masked below)
                const int b = tt / a.N, n = tt - b * a.N;
                const int2 o = *reinterpret_cast<const int2 *>(a.idx + 2 * n);
                z_b[i] = b; z_h[i] = o.x + z_p[i] / G::R; z_w[i] = o.y + z_p[i] % G::R;
            });
        }
        if constexpr (EPRE) {
            static_for<0, EU>([&](auto k_tag) {
                constexpr int k = decltype(k_tag)::value;
                const int o = tid + k * NT;
                const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
                const int m = o1 / (G::MT / 4);
                const int tt = min(mb * G::TPB + m / G::PX, a.T - 1);
                const int n = tt - (tt / a.N) * a.N;
                const int2 og = *reinterpret_cast<const int2 *>(a.idx + 2 * n);
                e_h[k] = og.x; e_w[k] = og.y;
            });
        }
        if constexpr (EPV) static_for<0, EU>([&](auto k_tag) {  // bias / out_affine entries of this lane's output channels
            constexpr int k = decltype(k_tag)::value;
            const int o = tid + k * NT;
            const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
            const int co = (ng * NB + nb) * G::MT + 4 * (o1 % (G::MT / 4));
            const int cs = (o < OUT_UNITS && co < a.Cout) ? co : 0;  // (dead units: a valid address, never used)
            e_bias[k] = e_os[k] = e_oh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e_first_pass) {
                if (a.bias) e_bias[k] = *reinterpret_cast<const float4 *>(a.bias + cs);
                if (a.oscale) {
                    e_os[k] = *reinterpret_cast<const float4 *>(a.oscale + cs);
                    e_oh[k] = *reinterpret_cast<const float4 *>(a.oshift + cs);
                }
            }
        });
        if (AFF) {
            tab_fetch(first, t_sc, t_sh);
            tab_fetch(min(first + 1, last), t_sc1, t_sh1);
        }
        SIGE_PROBE(7);  // slot arithmetic done, tile origins / bias / table entries requested
        b_prologue();
        int z_hw[NS], z_m0[NS], z_m1[NS], z_m2[NS];
        static_for<0, NS>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            unsigned off = kOOB, off2 = kOOB;
            const int t = z_t[i], c_l = z_c[i];
            if (SRC == SRC_TILES) {
                off = z_live[i] ? (unsigned)((t * G::RS + z_p[i]) * Cin + c_l) * 4u : kOOB;
            } else {
                const int h = z_h[i], w = z_w[i], b = z_b[i];
                int hlo = 0, hhi = a.H;
                if (a.hp_shift) {  // (stacked edits: the image this tile belongs to -- its window's third row is always inside it)
                    hlo = ((h - z_p[i] / G::R + 2) >> a.hp_shift) << a.hp_shift;
                    hhi = hlo + (1 << a.hp_shift);
                }
                const bool in = z_live[i] && h >= hlo && h < hhi && w >= 0 && w < a.W;
                z_live[i] = in;
                const int hw = (SRC == SRC_GATHER) ? (h >> a.up) * (a.W >> a.up) + (w >> a.up) : h * a.W + w;
                if (SRC == SRC_GATHER) {
                    off = in ? (unsigned)((b * HW + hw) * a.Csplit + c_l) * 4u : kOOB;
                    off2 = in ? (unsigned)(hw * (Cin - a.Csplit) + c_l) * 4u : kOOB;  // (x2: B == 1, host side)
                } else {
                    const int hwc = in ? hw : 0;
                    z_hw[i] = hwc;
                    const int32_t *m = a.map + 3 * (size_t)hwc;
                    z_m0[i] = m[0]; z_m1[i] = m[1]; z_m2[i] = m[2];
                }
            }
            if (SRC != SRC_SCATTER_GATHER) {
                s_off[i] = off;
                if (TWO) s_off2[i] = (off == kOOB ? 0u : off2 - off);
                if (AFF) s_tab[i] = (off != kOOB || off2 != kOOB) ? c_l : CCk;
            }
        });
        if constexpr (SRC == SRC_SCATTER_GATHER) {
            static_for<0, NS>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                const int blk = z_m0[i], c_l = z_c[i], b = z_b[i];
                const bool in = z_live[i];
                const unsigned off = (in && blk >= 0)
                    ? (unsigned)(((b * a.N + blk) * a.RxSx + z_m1[i] * a.Sx + z_m2[i]) * Cin + c_l) * 4u : kOOB;
                const unsigned off2 = (in && blk < 0) ? (unsigned)((b * HW + z_hw[i]) * Cin + c_l) * YB : kOOB;
                s_off[i] = off;
                s_off2[i] = off2;
                if (AFF) s_tab[i] = in ? c_l : CCk;
            });
        }
    } else {
    // (static_for, not `#pragma unroll`: a loop the unroller gives up on would index the slot arrays
    //  dynamically and push them to scratch memory -- every scratch load then drains vmcnt to 0)
    static_for<0, NS>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const int v = tid + NT * i;
        int t_l, c_l, p;  // tile in block, channel in chunk, pixel in tile
        if (NHWC) {
            constexpr int UPT = G::RS * CCk / 4, UPP = CCk / 4;  // units per tile / per pixel
            t_l = v / UPT;
            p = (v - t_l * UPT) / UPP;
            c_l = ((v - t_l * UPT) - p * UPP) * 4;
            s_dst[i] = (t_l * G::RS + p) * LDC + c_l;
        } else if (VEC) {
            constexpr int U4 = TILEF / 4;
            t_l = v / U4;
            const int e = (v - t_l * U4) * 4;
            c_l = e / G::RS; p = e - c_l * G::RS;  // (a float4 may straddle channels: c_l = its first)
            s_dst[i] = 4 * v;
        } else {
            t_l = v / TILEF;
            c_l = (v - t_l * TILEF) / G::RS;
            p = (v - t_l * TILEF) - c_l * G::RS;
            s_dst[i] = v;
        }
        if (VEC) s_cl[i] = c_l;
        const int t = mb * G::TPB + t_l;
        unsigned off = kOOB, off2 = kOOB;
        if (v < UNITS && t < a.T) {
            if (SRC == SRC_TILES) {
                off = NHWC ? (unsigned)((t * G::RS + p) * Cin + c_l) * 4u : (unsigned)(t * Cin * G::RS + (v % (TILEF / 4)) * 4) * 4u;
            } else {
                const int b = t / a.N;
                const int n = t - b * a.N;
                const int h = a.idx[2 * n] + p / G::R, w = a.idx[2 * n + 1] + p % G::R;
                if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
                    // GATHER with a.up = 1: the source tensor is the HALF-resolution one and (H, W) its x2
                    // nearest-neighbour upsampling, which therefore never has to exist (F.interpolate fused)
                    const int hw = (SRC == SRC_GATHER) ? (h >> a.up) * (a.W >> a.up) + (w >> a.up) : h * a.W + w;
                    if (SRC == SRC_GATHER) {
                        if (NHWC) {
                            off = (unsigned)((b * HW + hw) * a.Csplit + c_l) * 4u;
                            off2 = (unsigned)(hw * (Cin - a.Csplit) + c_l) * 4u;  // (x2: B == 1, host side)
                        } else {
                            off = (unsigned)((b * Cin + c_l) * HW + hw) * 4u;     // (with x2: B == 1)
                        }
                    } else {
                        const int32_t *m = a.map + 3 * (size_t)hw;
                        const int blk = m[0];
                        if (NHWC) {
                            if (blk >= 0) off = (unsigned)(((b * a.N + blk) * a.RxSx + m[1] * a.Sx + m[2]) * Cin + c_l) * 4u;
                            else off2 = (unsigned)((b * HW + hw) * Cin + c_l) * 4u;
                        } else {
                            if (blk >= 0) off = (unsigned)(((b * a.N + blk) * Cin + c_l) * a.RxSx + m[1] * a.Sx + m[2]) * 4u;
                            else off2 = (unsigned)((b * Cin + c_l) * HW + hw) * 4u;
                        }
                    }
                }
            }
        }
        s_off[i] = off;
        // GATHER / NHWC keeps the x2-side offset as a DELTA to the x side (added when a chunk lives in x2):
        // a select between two array elements would make the compiler keep both arrays in scratch memory
        if (TWO) s_off2[i] = (SRC == SRC_GATHER) ? (off == kOOB ? 0u : off2 - off) : off2;
        if (AFF) s_tab[i] = (off != kOOB || off2 != kOOB) ? c_l : CCk;
    });
    }

    // descriptors of the staging sources, advanced to channel chunk `chunk` with scalar arithmetic
    rsrc_t r_a, r_a2;
    bool use2 = false;  // NHWC GATHER: this chunk's channels live in x2
    auto set_chunk = [&](int chunk) {
        const int c0 = chunk * CCk;
        const long cstep = NHWC ? 1 : HW;  // elements between consecutive channels of a full tensor
        if (SRC == SRC_TILES) {
            r_a = make_rsrc(a.x, (long)c0 * (NHWC ? 1 : G::RS), (long)a.T * Cin * G::RS);
        } else if (SRC == SRC_GATHER) {
            // channels [0, Csplit) live in x [B,Csplit,H,W], [Csplit, Cin) in x2 [1,Cin-Csplit,H,W]
            use2 = c0 >= a.Csplit;
            if (!use2) r_a = make_rsrc(a.x, c0 * cstep, (long)a.B * a.Csplit * HW);
            else r_a = make_rsrc(a.x2, (c0 - a.Csplit) * cstep, (long)(Cin - a.Csplit) * HW);
        } else {
            r_a = make_rsrc(a.x, (long)c0 * (NHWC ? 1 : a.RxSx), (long)a.B * a.N * Cin * a.RxSx);
            if constexpr (Y16) r_a2 = make_rsrc_h(a.y, c0 * cstep, (long)a.B * Cin * HW);
            else r_a2 = make_rsrc(a.y, c0 * cstep, (long)a.B * Cin * HW);
        }
    };
    // float4 units: a partial last chunk must not read past the source's channels
    // (NCHW tile slab: the next tile; NHWC: the next pixel)
    auto vec_off = [&](unsigned off, int i, int chunk, int csrc, int cbase) -> unsigned {
        const int left = csrc - (chunk * CCk - cbase);  // channels of this source from the chunk start on
        if (NHWC) return s_cl[i] < left ? off : kOOB;
        return (int)((tid + NT * i) % (TILEF / 4)) * 4 < left * G::RS ? off : kOOB;
    };
    auto slot_load = [&](int set, int i, int chunk) {
        if (!VEC) {
            st_z[set][i] = buf_f32(r_a, s_off[i]);
            if (SRC == SRC_SCATTER_GATHER) st_z2[set][i] = buf_f32(r_a2, s_off2[i]);
        } else if (SRC == SRC_TILES) {
            st_q[set][i] = buf_f32x4(r_a, vec_off(s_off[i], i, chunk, Cin, 0), 0);
        } else if (SRC == SRC_GATHER) {
            const unsigned o = s_off[i] + (use2 ? s_off2[i] : 0u);
            st_q[set][i] = buf_f32x4(r_a, vec_off(o, i, chunk, use2 ? Cin - a.Csplit : a.Csplit, use2 ? a.Csplit : 0), 0);
        } else {
            st_q[set][i] = buf_f32x4(r_a, vec_off(s_off[i], i, chunk, Cin, 0), 0);
            if constexpr (Y16) st_q2[set][i] = buf_h4(r_a2, vec_off(s_off2[i], i, chunk, Cin, 0), 0);
            else st_q2[set][i] = buf_f32x4(r_a2, vec_off(s_off2[i], i, chunk, Cin, 0), 0);
        }
    };
    // finish slot i (affine + activation) and write it to LDS stage `buf`; `tb` = this chunk's table.
    //   scale, then shift, then activation: two separately rounded ops as in the reference
    //   (gather.cpp:33-53; built with -ffp-contract=off).  Zero fills read the table's zero
    //   entries: 0*0 + 0 = 0 and act(0) = 0, i.e. NOT affine-transformed (gather.cpp:27-30).
    auto finish = [&](float z, float sc, float sh) -> float {
        if (AFF) { z = sc * z; z = sh + z; }
        if (MODE == MODE_AFFINE_SWISH) z = swish_fast(z);
        return z;
    };
    auto slot_store = [&](int set, int i, float *buf, const float *tb) {
        if (NT * (i + 1) > UNITS && tid >= UNITS - NT * i) return;  // (only the last slot of a ragged split)
        if (!VEC) {
            float z = st_z[set][i];
            if (SRC == SRC_SCATTER_GATHER) z += st_z2[set][i];  // exactly one of the two is data, the other an exact 0
            float sc = 0.f, sh = 0.f;
            if (AFF) { sc = tb[s_tab[i]]; sh = tb[TROW + s_tab[i]]; }
            buf[s_dst[i]] = finish(z, sc, sh);
        } else {
            float4 q = st_q[set][i];
            if (SRC == SRC_SCATTER_GATHER) {
                const float4 q2 = st_q2[set][i];
                q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
            }
            if (AFF) {  // (NHWC only: the 4 values are 4 consecutive channels)
                const float4 sc = *reinterpret_cast<const float4 *>(tb + s_tab[i]);
                const float4 sh = *reinterpret_cast<const float4 *>(tb + TROW + s_tab[i]);
                q.x = finish(q.x, sc.x, sh.x); q.y = finish(q.y, sc.y, sh.y);
                q.z = finish(q.z, sc.z, sh.z); q.w = finish(q.w, sc.w, sh.w);
            }
            if constexpr (F16) {
                const f16x4 h = {(_Float16)q.x, (_Float16)q.y, (_Float16)q.z, (_Float16)q.w};  // RNE
                *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(buf) + s_dst[i]) = h;
                if constexpr (X3) {  // the lo plane of the same pixel row: what fp16 rounding dropped (exact in fp32, then RNE)
                    const f16x4 l = {(_Float16)(q.x - (float)h[0]), (_Float16)(q.y - (float)h[1]),
                                     (_Float16)(q.z - (float)h[2]), (_Float16)(q.w - (float)h[3])};
                    *reinterpret_cast<f16x4 *>(reinterpret_cast<_Float16 *>(buf) + s_dst[i] + CCk) = l;
                }
            } else {
                *reinterpret_cast<float4 *>(buf + s_dst[i]) = q;
            }
        }
    };
    auto tab_load = [&](int chunk) {
        if (AFF) tab_fetch(chunk, t_sc, t_sh);
    };
    auto tab_put = [&](float *tb, float sc, float sh) {
        tb[trow] = sc;
        tb[TROW + trow] = sh;
        if (tid < 4) { tb[CCk + tid] = 0.f; tb[TROW + CCk + tid] = 0.f; }
    };
    auto tab_store = [&](float *tb) {
        if (AFF) tab_put(tb, t_sc, t_sh);
    };

    typename M::acc_t acc[NB][NACC];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int i = 0; i < M::REGS; ++i) acc[nb][q][i] = 0.0f;

    // A: this lane's output pixel = row j of the M block; k-step u = (channel group q, tap)
    const int tl = j / G::PX, px = j % G::PX;
    const int oy = px / G::RO, ox = px % G::RO;
    //   (f16 compute: k-step u = (slab, tap); the lane's operand = 8 consecutive channels, offsets in halves)
    const int a_base = F16 ? (tl * G::RS + oy * G::S * G::R + ox * G::S) * LDC + wave * G::CW + kq * 8
                     : NHWC ? (tl * G::RS + oy * G::S * G::R + ox * G::S) * LDC + wave * G::CW + kq
                            : tl * TILEF + (wave * G::CW + kq) * G::RS + oy * G::S * G::R + ox * G::S;
    auto a_off = [](int u) {
        const int q = u / G::KK, tap = u % G::KK, pix = (tap / G::K) * G::R + (tap % G::K);
        if (F16) return pix * LDC + q * (G::NL * 8);
        return NHWC ? pix * LDC + q * G::NL : q * G::NL * G::RS + pix;
    };

    // ---- prologue: chunks 0 / 1 -> register sets 0 / 1, B sets 0 / 1, tables 0 / 1;
    //      chunk 0 -> LDS[0]; register set 0 re-issued as chunk 2 ----
    set_chunk(first);
    static_for<0, NS>([&](auto i_tag) { slot_load(0, decltype(i_tag)::value, first); });
    set_chunk(min(first + 1, last));
    static_for<0, NS>([&](auto i_tag) { slot_load(1, decltype(i_tag)::value, min(first + 1, last)); });
    if constexpr (NHWC && EPRE) epre_setup();
    if (AFF) {
        if (!NHWC) {  // (channels-last: both entries are already in flight, see the slot set-up)
            tab_fetch(first, t_sc, t_sh);
            tab_fetch(min(first + 1, last), t_sc1, t_sh1);
        }
        tab_put(tab, t_sc, t_sh);
        tab_put(tab + TABF, t_sc1, t_sh1);
        __syncthreads();
    }
    set_chunk(min(first + 2, last));
    SIGE_PROBE(1);  // all prologue loads issued (index / map round trips done)
    static_for<0, NS>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        slot_store(0, i, smem, tab);
        slot_load(0, i, min(first + 2, last));
    });
    SIGE_PROBE(2);  // first chunk's data arrived and is in LDS
    __syncthreads();
    SIGE_PROBE(3);  // every wave's share is in LDS

    // one chunk: MFMAs on LDS[PAR] with B set PAR; register set PAR^1 (chunk+1) -> LDS[PAR^1],
    // re-issued as chunk+3; B set PAR re-issued as chunk+2.  Loads past the last chunk re-read it.
    auto body = [&](auto par_tag, int chunk) {
        constexpr int PAR = decltype(par_tag)::value;
        const float *as = smem + PAR * STAGE + a_base;
        float *nxt = smem + (PAR ^ 1) * STAGE;
        const int c2 = min(chunk + 2, last), c3 = min(chunk + 3, last);
        set_chunk(c3);
        set_b_chunk(c2);
        tab_load(c2);
        if constexpr (F16) {
            // one 16-byte A operand (8 channels of the lane's pixel at one tap) and one 16-byte B register per k-step
            const _Float16 *ah = reinterpret_cast<const _Float16 *>(smem + PAR * STAGE) + a_base;
            f16x8 avh[G::L], avl[X3 ? G::L : 1];
#pragma unroll
            for (int u = 0; u < G::L; ++u) {
                avh[u] = *reinterpret_cast<const f16x8 *>(ah + a_off(u));
                if constexpr (X3) avl[u] = *reinterpret_cast<const f16x8 *>(ah + a_off(u) + CCk);
            }
            static_for<0, G::L>([&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                static_for<0, NB>([&](auto nb_tag) {
                    constexpr int nb = decltype(nb_tag)::value;
                    if constexpr (X3) {
                        // hi*hi and hi*lo on one accumulator, lo*hi on the other: consecutive MFMAs alternate accumulators
                        const float4 bq = bset[PAR][nb][2 * u], bl = bset[PAR][nb][2 * u + 1];
                        const f32x4 bhr = {bq.x, bq.y, bq.z, bq.w}, blr = {bl.x, bl.y, bl.z, bl.w};
                        acc[nb][u % NACC] = M::op(avh[u], __builtin_bit_cast(f16x8, bhr), acc[nb][u % NACC]);
                        acc[nb][(u + 1) % NACC] = M::op(avl[u], __builtin_bit_cast(f16x8, bhr), acc[nb][(u + 1) % NACC]);
                        acc[nb][u % NACC] = M::op(avh[u], __builtin_bit_cast(f16x8, blr), acc[nb][u % NACC]);
                    } else {
                        const float4 bq = bset[PAR][nb][u];
                        const f32x4 braw = {bq.x, bq.y, bq.z, bq.w};
                        acc[nb][u % NACC] = M::op(avh[u], __builtin_bit_cast(f16x8, braw), acc[nb][u % NACC]);
                    }
                });
                static_for<(u * NS) / G::L, ((u + 1) * NS) / G::L>([&](auto i_tag) {
                    constexpr int i = decltype(i_tag)::value;
                    slot_store(PAR ^ 1, i, nxt, tab + (PAR ^ 1) * TABF);
                    slot_load(PAR ^ 1, i, c3);
                });
                static_for<0, NB>([&](auto nb_tag) {
                    constexpr int nb = decltype(nb_tag)::value;
                    if constexpr (X3) {
                        b_load(bset[PAR][nb][2 * u], nb, 2 * u);
                        b_load(bset[PAR][nb][2 * u + 1], nb, 2 * u + 1);
                    } else {
                        b_load(bset[PAR][nb][u], nb, u);
                    }
                });
            });
            tab_store(tab + PAR * TABF);
            __syncthreads();
            return;
        } else {
        // all A values of the chunk up front: LDS reads overlap with the matrix pipe for free
        float av[G::L];
#pragma unroll
        for (int u = 0; u < G::L; ++u) av[u] = SIGE_ABL_HAS(16) ? __builtin_bit_cast(float, 0x3f800000 + u + lane) : as[a_off(u)];
        static_for<0, G::L / 4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            static_for<0, 4>([&](auto e_tag) {
                constexpr int e = decltype(e_tag)::value;
                constexpr int u = 4 * g + e;
                static_for<0, NB>([&](auto nb_tag) {
                    constexpr int nb = decltype(nb_tag)::value;
                    const float4 bq = bset[PAR][nb][g];
                    const float bv = (e == 0) ? bq.x : (e == 1) ? bq.y : (e == 2) ? bq.z : bq.w;
                    acc[nb][u % NACC] = M::op(av[u], bv, acc[nb][u % NACC]);
                });
                // staging slots spread evenly over the k-steps
                static_for<(u * NS) / G::L, ((u + 1) * NS) / G::L>([&](auto i_tag) {
                    constexpr int i = decltype(i_tag)::value;
                    if (!SIGE_ABL_HAS(4)) slot_store(PAR ^ 1, i, nxt, tab + (PAR ^ 1) * TABF);
                    if (!SIGE_ABL_HAS(2)) slot_load(PAR ^ 1, i, c3);
                });
            });
            static_for<0, NB>([&](auto nb_tag) {
                constexpr int nb = decltype(nb_tag)::value;
                if (!SIGE_ABL_HAS(1)) b_load(bset[PAR][nb][g], nb, g);
            });
        });
        tab_store(tab + PAR * TABF);  // table of chunk+2 replaces the one this chunk's predecessor used
        if (!SIGE_ABL_HAS(8)) __syncthreads();
        }
    };

    for (int chunk = first; chunk <= last; chunk += 2) {
        body(std::integral_constant<int, 0>{}, chunk);
        if (chunk + 1 <= last) body(std::integral_constant<int, 1>{}, chunk + 1);
    }

    SIGE_PROBE(4);  // K loop done
    // ---- K-split reduction across the 4 waves, bias, store -----------------
    // MT=32: reg r of lane (kq, j): pixel row m = (r&3) + 8*(r>>2) + 4*kq ; MT=16: m = 4*kq + r ; column (cout) = j
    float *red = smem;  // safe: the loop ended with a barrier
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        typename M::acc_t s = acc[nb][0];
        if (NACC == 2) {
#pragma unroll
            for (int i = 0; i < M::REGS; ++i) s[i] += acc[nb][NACC - 1][i];
        }
        if (NHWC) {
            // red[wave][nb][pixel m][cout n]
            float *r = red + ((wave * NB + nb) * G::MT) * RP + j;
#pragma unroll
            for (int q = 0; q < M::REGS; ++q) {
                const int m = G::MT == 32 ? (q & 3) + 8 * (q >> 2) + 4 * kq : 4 * kq + q;
                r[m * RP] = s[q];
            }
        } else {
            // red[wave][nb][cout n][pixel m]
            float *r = red + ((wave * NB + nb) * G::MT + j) * RP;
            if (G::MT == 32) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(r + 8 * g + 4 * kq) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
            } else {
                *reinterpret_cast<float4 *>(r + 4 * kq) = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
    }
    __syncthreads();

    // K split: every split writes its partial sums (no bias / residual) to its own copy of the output in the workspace.
    //   a.counters == nullptr: splitk_reduce_nhwc_kernel (a second launch) adds them up with the epilogue;
    //   a.counters != nullptr: the LAST workgroup to finish an output block (a ticket per block, device-scope release /
    //   acquire around it) adds the copies of ITS block up in split order -- the same fixed order, so the result is the
    //   same bits as the second pass gives -- and runs the epilogue into a.fout.  No second launch.
    const bool split_k = a.ksplit > 1;
    float *const outp = a.out + (size_t)split * a.split_stride;
    if (NHWC) {
        // one float4 = 4 consecutive output channels of one pixel per lane and step
        struct Unit { bool ok; size_t addr; int co, b, h, w; float4 rr; };
        auto locate = [&](auto k_tag, const bool want_res) -> Unit {
            constexpr int k = decltype(k_tag)::value;
            Unit u;
            u.ok = false; u.addr = 0; u.co = 0; u.b = 0; u.h = 0; u.w = 0; u.rr = make_float4(0.f, 0.f, 0.f, 0.f);
            const int o = tid + k * NT;
            if (EU * NT > OUT_UNITS && o >= OUT_UNITS) return u;
            const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
            const int n4 = o1 % (G::MT / 4), m = o1 / (G::MT / 4);
            const int t_l = m / G::PX, pxo = m % G::PX;
            const int t = mb * G::TPB + t_l;
            u.co = (ng * NB + nb) * G::MT + 4 * n4;
            if (!(t < a.T && u.co < a.Cout)) return u;  // (Cout % 4 == 0: host side)
            u.b = t / a.N;
            if constexpr (DST == DST_TILES) {
                u.ok = true;
                u.addr = ((size_t)t * G::PX + pxo) * a.Cout + u.co;
            } else if constexpr (EPRE) {
                // (4 waves: pixel, address and residual were fetched with the prologue's loads)
                u.ok = e_in[k]; u.h = e_h[k]; u.w = e_w[k]; u.addr = e_q[k]; u.rr = cvt_raw_f32x4_or_h4(e_res[k], a.res_f16 != 0);
            } else {
                const int n = t - u.b * a.N;
                u.h = (a.offH + a.idx[2 * n]) / a.strH + pxo / G::RO;
                u.w = (a.offW + a.idx[2 * n + 1]) / a.strW + pxo % G::RO;
                u.ok = u.h >= 0 && u.h < a.Ho && u.w >= 0 && u.w < a.Wo;
                u.addr = u.ok ? (((size_t)u.b * a.Ho + u.h) * a.Wo + u.w) * a.Cout + u.co : 0;
                if (u.ok && a.residual && want_res) u.rr = ld_f32x4_or_h4(a.residual, u.addr, a.res_f16 != 0);
            }
            return u;
        };
        // bias, residual (block residual), the consumer's affine + activation, store: the epilogue proper
        auto emit = [&](auto k_tag, const Unit &u, float4 s, float *dst) {
            constexpr int k = decltype(k_tag)::value;
            if (a.bias) {
                float4 bb;
                if constexpr (EPV) bb = e_bias[k];
                else bb = *reinterpret_cast<const float4 *>(a.bias + u.co);
                s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
            }
            if constexpr (DST != DST_TILES) {
                if (a.residual) {
                    const float4 rr = u.rr;
                    s.x += rr.x; s.y += rr.y; s.z += rr.z; s.w += rr.w;
                    if (a.x1) {
                        const int t1 = a.table1[(u.h / a.R1) * a.gW1 + u.w / a.S1];
                        if (t1 >= 0) {
                            const float4 xv = *reinterpret_cast<const float4 *>(
                                a.x1 + ((((size_t)u.b * a.N1 + t1) * a.R1 + u.h % a.R1) * a.S1 + u.w % a.S1) * a.Cout + u.co);
                            s.x += xv.x - rr.x; s.y += xv.y - rr.y; s.z += xv.z - rr.z; s.w += xv.w - rr.w;
                        }
                    }
                }
            }
            if constexpr (DST != DST_TILES) {
                auto twin = [&](float *dst2, const float *ts, const float *tt) {
                    const float4 sc = *reinterpret_cast<const float4 *>(ts + u.co), sh = *reinterpret_cast<const float4 *>(tt + u.co);
                    float4 t;
                    t.x = sc.x * s.x; t.y = sc.y * s.y; t.z = sc.z * s.z; t.w = sc.w * s.w;
                    t.x = sh.x + t.x; t.y = sh.y + t.y; t.z = sh.z + t.z; t.w = sh.w + t.w;
                    t.x = swish(t.x); t.y = swish(t.y); t.z = swish(t.z); t.w = swish(t.w);
                    store_out4(dst2 + u.addr, t);
                };
                if (a.twin0) twin(a.twin0, a.tscale0, a.tshift0);
                if (a.twin1) twin(a.twin1, a.tscale1, a.tshift1);
            }
            if (a.oscale) {
                float4 os, oh;
                if constexpr (EPV) { os = e_os[k]; oh = e_oh[k]; }
                else { os = *reinterpret_cast<const float4 *>(a.oscale + u.co); oh = *reinterpret_cast<const float4 *>(a.oshift + u.co); }
                s.x = os.x * s.x; s.y = os.y * s.y; s.z = os.z * s.z; s.w = os.w * s.w;
                s.x = oh.x + s.x; s.y = oh.y + s.y; s.z = oh.z + s.z; s.w = oh.w + s.w;
                if (a.oact == SIGE_HIP_ACT_SWISH) { s.x = swish(s.x); s.y = swish(s.y); s.z = swish(s.z); s.w = swish(s.w); }
            }
            store_out4(dst + u.addr, s);  // (write-through: common.hpp)
        };
        static_for<0, EU>([&](auto k_tag) {
            constexpr int k = decltype(k_tag)::value;
            const Unit u = locate(k_tag, !split_k);
            if (!u.ok) return;
            const int o = tid + k * NT;
            const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
            const int n4 = o1 % (G::MT / 4), m = o1 / (G::MT / 4);
            const float *r0 = red + (nb * G::MT + m) * RP + 4 * n4;
            float4 s = *reinterpret_cast<const float4 *>(r0);
#pragma unroll
            for (int w = 1; w < W; ++w) {
                const float4 v = *reinterpret_cast<const float4 *>(r0 + w * NB * G::MT * RP);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if constexpr (X3) { s.x *= wsc; s.y *= wsc; s.z *= wsc; s.w *= wsc; }  // (a power of two: exact; K-split partials are stored scaled back)
            if (!split_k) emit(k_tag, u, s, a.out);
            else if (!a.counters) *reinterpret_cast<float4 *>(outp + u.addr) = s;
            else coherent_store(outp + u.addr, s);
        });
        if (split_k && a.counters) {
            // The partial sums went out as device-coherent (relaxed, agent-scope atomic) stores: they are at the device's
            // coherence point once they have completed (vmcnt 0) -- no L2 write-back / invalidate of the whole XCD, which a
            // release / acquire fence would cost every workgroup (measured: +11 us per launch).  Then the block's ticket.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int32_t *const cnt = a.counters + mb * a.ngk + ng;
            if (tid == 0) red[0] = __builtin_bit_cast(float, __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __syncthreads();
            const int ticket = __builtin_bit_cast(int, red[0]);
            if (ticket == a.ksplit - 1) {
                if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                static_for<0, EU>([&](auto k_tag) {
                    const Unit u = locate(k_tag, true);
                    if (!u.ok) return;
                    // all (<= 8) copies in flight at once, added in split order
                    float4 pv[8];
#pragma unroll
                    for (int sidx = 0; sidx < 8; ++sidx)
                        pv[sidx] = coherent_load(a.out + (size_t)(sidx < a.ksplit ? sidx : a.ksplit - 1) * a.split_stride + u.addr);
                    float4 s = pv[0];
#pragma unroll
                    for (int sidx = 1; sidx < 8; ++sidx)
                        if (sidx < a.ksplit) { s.x += pv[sidx].x; s.y += pv[sidx].y; s.z += pv[sidx].z; s.w += pv[sidx].w; }
                    emit(k_tag, u, s, a.fout);
                });
            }
        }
        SIGE_PROBE(5);  // stores issued
        return;
    }
    const float *const biasp = split_k ? nullptr : a.bias;
    const float *const resp = split_k ? nullptr : a.residual;
    // NCHW: one float4 (4 consecutive pixels of one tile and one output channel) per lane and step
    constexpr int P4 = G::PX / 4;                       // float4 per (tile, channel)
#pragma unroll
    for (int o = tid; o < OUT_UNITS; o += NT) {
        const int nb = o / UNITS_NB, o1 = o - nb * UNITS_NB;
        const int p4 = o1 % P4;
        const int co_l = (o1 / P4) % G::MT;
        const int t_l = o1 / (P4 * G::MT);
        const int rrow = t_l * G::PX + p4 * 4;
        const float *r0 = red + (nb * G::MT + co_l) * RP + rrow;
        float4 s = *reinterpret_cast<const float4 *>(r0);
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(r0 + w * NB * G::MT * RP);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int t = mb * G::TPB + t_l, co = (ng * NB + nb) * G::MT + co_l;
        if (t < a.T && co < a.Cout) {
            const float bb = biasp ? biasp[co] : 0.0f;
            s.x += bb; s.y += bb; s.z += bb; s.w += bb;
            if (DST == DST_TILES) {
                *reinterpret_cast<float4 *>(outp + ((size_t)t * a.Cout + co) * G::PX + p4 * 4) = s;
            } else {
                const int b = t / a.N, n = t - b * a.N;
                const int h0 = (a.offH + a.idx[2 * n]) / a.strH, w0 = (a.offW + a.idx[2 * n + 1]) / a.strW;
                const size_t plane = ((size_t)b * a.Cout + co) * a.Ho * a.Wo;
                const float sv[4] = {s.x, s.y, s.z, s.w};
                if (G::RO == 4) {
                    // one 4-pixel output row of the tile
                    const int h = h0 + p4;
                    if (h >= 0 && h < a.Ho) {
                        const size_t q = plane + (size_t)h * a.Wo + w0;
                        if (w0 >= 0 && w0 + 3 < a.Wo && ((q & 3) == 0)) {
                            float4 ov = s;
                            if (resp) {
                                const float4 rr = *reinterpret_cast<const float4 *>(resp + q);
                                ov.x += rr.x; ov.y += rr.y; ov.z += rr.z; ov.w += rr.w;
                            }
                            *reinterpret_cast<float4 *>(outp + q) = ov;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (w0 + i >= 0 && w0 + i < a.Wo)
                                    outp[q + i] = sv[i] + (resp ? resp[q + i] : 0.0f);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pp = p4 * 4 + i;
                        const int h = h0 + pp / G::RO, w = w0 + pp % G::RO;
                        if (h >= 0 && h < a.Ho && w >= 0 && w < a.Wo) {
                            const size_t q = plane + (size_t)h * a.Wo + w;
                            outp[q] = sv[i] + (resp ? resp[q] : 0.0f);
                        }
                    }
                }
            }
        }
    }
}

// ---- launch ------------------------------------------------------------------
inline int conv_grid_x(const ConvArgs &a) { return a.ng_fast == 2 ? 8 * ((a.mbk + 7) / 8) * a.ngk : a.mbk * a.ngk; }

template <typename G, int NB, int SRC, int MODE, int DST, int LAYOUT = LAYOUT_NCHW, int W = 4, bool Y16 = false>
__global__ __launch_bounds__(64 * W) void conv_mfma_kernel(const ConvArgs a) {
    kernarg_touch<sizeof(ConvArgs)>();
    __shared__ __attribute__((aligned(16))) float smem[conv_lds_floats<G, NB, MODE, LAYOUT, W>()];
    conv_mfma_body<G, NB, SRC, MODE, DST, LAYOUT, W, Y16>(a, blockIdx.x, blockIdx.y, smem);
}

// Two independent convs of a residual block in ONE launch (horizontal fusion): workgroups [0, na) run conv A
// (3x3, gather + cached affine + SiLU: the block's conv1), the rest conv B (the 1x1 shortcut on the same input).  Both are
// a few microseconds of work dominated by their start-up, and B's ~5 us launch disappears behind A.  A may be K-split over
// gridDim.y; B never is (its workgroups with blockIdx.y > 0 leave at once).
// MODEA: staging of conv A -- MODE_AFFINE_SWISH (the consumer activates its input), or MODE_RAW (its producers wrote an
// activated twin: ConvArgs::twin0/1)
template <typename GA, int NBA, typename GB, int DST, int W, int MODEA>
__global__ __launch_bounds__(64 * W) void conv_pair_kernel(const ConvArgs a, const ConvArgs b, const int na) {
    kernarg_touch<2 * sizeof(ConvArgs)>();
    constexpr int LA = conv_lds_floats<GA, NBA, MODEA, LAYOUT_NHWC, W>();
    constexpr int LB = conv_lds_floats<GB, 1, MODE_RAW, LAYOUT_NHWC, W>();
    __shared__ __attribute__((aligned(16))) float smem[cmax(LA, LB)];
    if ((int)blockIdx.x < na)
        conv_mfma_body<GA, NBA, SRC_GATHER, MODEA, DST, LAYOUT_NHWC, W>(a, blockIdx.x, blockIdx.y, smem);
    else if (blockIdx.y == 0)
        conv_mfma_body<GB, 1, SRC_GATHER, MODE_RAW, DST, LAYOUT_NHWC, W>(b, blockIdx.x - na, 0, smem);
}

template <typename GA, int NBA, typename GB, int DST, int W>
void launch_conv_pair(ConvArgs a, ConvArgs b, int mode_a, hipStream_t st);

#define SIGE_CONV_PAIR_INSTANTIATE(GA, NBA, GB, DST, W)                                                   \
    template <> void launch_conv_pair<GA, NBA, GB, DST, W>(ConvArgs a, ConvArgs b, int mode_a, hipStream_t st) { \
        const int na = conv_grid_x(a);                                                                    \
        const dim3 grid(na + conv_grid_x(b), a.ksplit);                                                   \
        if (mode_a == MODE_RAW) conv_pair_kernel<GA, NBA, GB, DST, W, MODE_RAW><<<grid, 64 * W, 0, st>>>(a, b, na); \
        else conv_pair_kernel<GA, NBA, GB, DST, W, MODE_AFFINE_SWISH><<<grid, 64 * W, 0, st>>>(a, b, na);  \
    }


template <typename G, int NB, int SRC, int DST, int LAYOUT, int W>
void launch_conv_geo(ConvArgs a, int mode, hipStream_t st);

// mode: MODE_* (host side maps (scale, shift, activation) onto it)
#define SIGE_CONV_LAUNCH3(G, NB, SRC, DST, LAY, W)                                                        \
    template <> void launch_conv_geo<G, NB, SRC, DST, LAY, W>(ConvArgs a, int mode, hipStream_t st) {     \
        const dim3 grid(conv_grid_x(a), a.ksplit);                                                        \
        if (mode == MODE_AFFINE_SWISH) conv_mfma_kernel<G, NB, SRC, MODE_AFFINE_SWISH, DST, LAY, W><<<grid, 64 * W, 0, st>>>(a); \
        else if (mode == MODE_AFFINE) conv_mfma_kernel<G, NB, SRC, MODE_AFFINE, DST, LAY, W><<<grid, 64 * W, 0, st>>>(a);        \
        else conv_mfma_kernel<G, NB, SRC, MODE_RAW, DST, LAY, W><<<grid, 64 * W, 0, st>>>(a);             \
    }

// explicit-instantiation helper used by the per-geometry translation units
#define SIGE_CONV_INSTANTIATE(G, NB, LAY, W)                                                              \
    template <> void launch_conv_geo<G, NB, SRC_TILES, DST_TILES, LAY, W>(ConvArgs a, int, hipStream_t st) { \
        conv_mfma_kernel<G, NB, SRC_TILES, MODE_RAW, DST_TILES, LAY, W><<<dim3(conv_grid_x(a), a.ksplit), 64 * W, 0, st>>>(a); \
    }                                                                                                     \
    SIGE_CONV_LAUNCH3(G, NB, SRC_GATHER, DST_TILES, LAY, W)                                               \
    SIGE_CONV_LAUNCH3(G, NB, SRC_GATHER, DST_NCHW, LAY, W)                                                \
    SIGE_CONV_LAUNCH3(G, NB, SRC_SCATTER_GATHER, DST_TILES, LAY, W)

// scatter_gather source whose cached tensor `y` is stored as fp16 (channels-last, 4 waves): tiles or full-tensor destination
template <typename G, int NB, int DST>
void launch_conv_c16(ConvArgs a, int mode, hipStream_t st);
#define SIGE_CONV_INSTANTIATE_C16_DST(G, NB, DST)                                                         \
    template <> void launch_conv_c16<G, NB, DST>(ConvArgs a, int mode, hipStream_t st) {                  \
        const dim3 grid(conv_grid_x(a), a.ksplit);                                                        \
        if (mode == MODE_AFFINE_SWISH) conv_mfma_kernel<G, NB, SRC_SCATTER_GATHER, MODE_AFFINE_SWISH, DST, LAYOUT_NHWC, 4, true><<<grid, 256, 0, st>>>(a); \
        else if (mode == MODE_AFFINE) conv_mfma_kernel<G, NB, SRC_SCATTER_GATHER, MODE_AFFINE, DST, LAYOUT_NHWC, 4, true><<<grid, 256, 0, st>>>(a);        \
        else conv_mfma_kernel<G, NB, SRC_SCATTER_GATHER, MODE_RAW, DST, LAYOUT_NHWC, 4, true><<<grid, 256, 0, st>>>(a);             \
    }
#define SIGE_CONV_INSTANTIATE_C16(G, NB) SIGE_CONV_INSTANTIATE_C16_DST(G, NB, DST_TILES) SIGE_CONV_INSTANTIATE_C16_DST(G, NB, DST_NCHW)

// scatter_gather source written straight into a full tensor (conv2 -> Scatter fused): channels-last 3x3 only
#define SIGE_CONV_INSTANTIATE_SG_FULL(G, NB, LAY, W) SIGE_CONV_LAUNCH3(G, NB, SRC_SCATTER_GATHER, DST_NCHW, LAY, W)

}  // namespace sige
