// Dense-layer convolution on the fp16 matrix cores of gfx950: 64 pixels x 64 output channels per workgroup.
//
//   out[b,h,w,co] = epilogue( sum_{ci,tap} act(scale[ci] * cat(x, x2)[b, h+dy, w+dx, ci] + shift[ci]) * w[co,ci,tap] )
//
// for the layers of a SIGE network that run DENSE: the resolutions below the sparse threshold in the sparse pass
// (sige_fused_unet.py:112-123: 41.9 of the 60 GFLOP of a 1.2 % forward) and every conv of the full pass that produces the
// caches (sige/nn/base.py:85-86, diffusion/samplers/ddim_ddpm_sampler.py:60-66).  Channels-last fp32 tensors in HBM, zero
// padding = the staging path's zero fill AFTER affine / activation (exactly like padding the activated tensor).
//
// Why a second conv kernel next to conv_mfma.hpp: that one is built for a handful of 4x4 tiles (16 or 32 pixels x 32 or 64
// channels per workgroup, every workgroup pulling its own copy of a weight slice); a dense layer has 64 ... 65536 pixels
// in a regular grid, and on the f32-input MFMA (64 FLOP/clk/SIMD) it is matrix-bound at best.  Here:
//   * the product runs on v_mfma_f32_32x32x16_f16 (1024 FLOP/clk/SIMD) in one of two operand forms
//       F16   : operands rounded to fp16 (RNE), fp32 accumulation                      (BASELINE.json configs[4])
//       F16X3 : every fp32 operand is split x = hi + lo, hi = fp16(x), lo = fp16(x - hi); three products
//               hi*hi + lo*hi + hi*lo accumulate in fp32 -- 22-bit operands, the dropped lo*lo term is 2^-22 relative:
//               fp32-level results (max |d| vs an fp64 conv ~1e-6 relative) at 5.3x the f32-MFMA rate.  Weights are
//               pre-scaled by a power of two at pack time (exact) so that their lo parts are normal fp16 numbers.
//   * workgroup = 8x8 output pixels x 64 output channels = 2x2 MFMA tiles per wave; the four waves split K (each owns a
//     quarter of every channel chunk) and meet in LDS at the end -- so inside a workgroup no operand byte is fetched or
//     staged twice: A (the 10x10 halo patch of the wave's 16 channels, affine + SiLU applied ONCE per element, split
//     into fp16 planes) goes through a wave-PRIVATE double-buffered LDS stage (no workgroup barrier in the K loop);
//     B (pre-packed fp16 planes in MFMA lane order) streams straight into a register ring, prefetched RB steps ahead.
//   * small layers (8x8 ... 32x32 pixels, K up to 9216) split K across workgroups and finish inside the launch
//     (ticket per output block, conv_mfma.hpp's scheme), so the weights of a layer are read from HBM exactly once and
//     shared through L2 by the (few) pixel blocks.
#pragma once
#include "conv_mfma.hpp"

namespace sige {

// PWO_: width of the output patch of a workgroup -- 8 (8 x 8 pixels, two waves per SIMD: small maps, many workgroups) or
// 16 (8 x 16 pixels = four M tiles per wave, ONE wave per SIMD with the 512-register budget: every weight byte pulled from
// L2 feeds twice the matrix work, and the weight ring holds a whole chunk -- a workgroup of the 8 x 8 form needs 4 KB of
// weights per 384 cycles of MFMA and wave, 85 B/clk/CU with eight waves, about twice what a CU can pull from L2).
enum { WIDE_F16 = 0, WIDE_X3 = 1, WIDE_F32 = 2 };  // operand form (the `prec` argument of the C ABI)

template <int KH_, int PREC_, int PWO_ = 8>
struct WideGeo {
    static constexpr int KH = KH_, KK = KH_ * KH_;
    static constexpr int PREC = PREC_;
    static constexpr bool X3 = PREC_ == WIDE_X3, F32 = PREC_ == WIDE_F32;
    static constexpr int PWO = PWO_;                   // output patch: 8 rows x PWO columns
    static constexpr int MTN = PWO_ / 4;               // 32-pixel M tiles per wave (2 | 4)
    static constexpr int RPT = 32 / PWO_;              // patch rows per M tile (4 | 2)
    static constexpr int BM = 8 * PWO_;                // output pixels per workgroup
    static constexpr int OCC = PWO_ == 16 ? 1 : 2;     // waves per SIMD the register budget is set for
    // 16-byte operand pieces per lane and k-step: F16 one (8 halves); X3 two planes (hi, lo); F32 two (8 floats = k-steps 0..3 | 4..7
    // of the eight v_mfma_f32_32x32x2_f32 that contract the 16 channels: lane group kq holds channels 2 s + kq)
    static constexpr int NP = PREC_ == WIDE_F16 ? 1 : 2;
    static constexpr int KS = KH_ == 1 ? 2 : 1;        // 16-channel k-steps per tap and chunk (per wave)
    static constexpr int CW = 16 * KS;                 // channels per wave per chunk
    static constexpr int CC = 4 * CW;                  // channels per chunk (4 waves split K)
    static constexpr int STEPS = KS * KK;              // k-steps per chunk; one step = 2x2 tiles x (1 | 3) MFMAs
    static constexpr int PW = PWO_ + (KH_ == 3 ? 2 : 0);  // width of the staged patch (outputs + halo)
    static constexpr int PH = 8 + (KH_ == 3 ? 2 : 0);  // its height
    static constexpr int NPX = PW * PH;                // staged pixels
    static constexpr int QP = CW / 4;                  // float4 units per staged pixel (per wave)
    static constexpr int UNITS = NPX * QP;
    static constexpr int NS = (UNITS + 63) / 64;       // staging slots per lane
    static constexpr int KSB = 32 * NP;                // bytes of one k-step group of a pixel row: 16 hi halves (+ 16 lo) | 16 floats
    static constexpr int KQB = F32 ? 32 : 16;          // byte offset of lane group kq = 1 inside a k-step group
    static constexpr int PLB = F32 ? 16 : 32;          // byte offset of the second operand piece
    static constexpr int ROWB = KS * KSB + 16;         // LDS row of one staged pixel (padded against bank conflicts)
    static constexpr int ABUF = NPX * ROWB;            // bytes of one stage of one wave
    static constexpr int STEPB = 2 * NP * 1024;        // packed weight bytes per k-step of one wave: [nt][plane][lane][16 B]
    // weight register ring, in k-steps (= prefetch distance): a whole chunk where the register budget allows
    static constexpr int RB = KH_ == 3 ? (PREC_ == WIDE_F16 ? 9 : (PWO_ == 8 ? 3 : 6)) : 4;
    static_assert((2 * STEPS) % RB == 0, "the ring position of a step must not depend on the chunk");
    static_assert(RB <= 9, "kWidePadSteps");
    static constexpr int LDS_BYTES = cmax(4 * 2 * ABUF, 4 * BM * 68 * 4);
};

constexpr int kWideMaxSplit = 16;
constexpr int kWidePadSteps = 9;   // k-steps of padding behind the packed weights (>= every geometry's RB: the prefetch runs past the last step)

struct WideArgs {
    const float *x, *x2;        // [B,Hs,Ws,C1], [B,Hs,Ws,C2] channels-last (Hs = H >> up); channels of x2 follow those of x
    const void *packed;         // packed weights of this launch's form (sige_hip_wide_conv_pack)
    const float *bias, *scale, *shift, *residual, *oscale, *oshift;
    float *out;                 // [B,H,W,Cout]; K split: the workspace (ksplit copies, split_stride floats apart)
    float *fout;                // K split: the real destination
    float *twin0, *twin1;       // optional activated twins (conv_mfma.hpp: ConvArgs::twin0/1)
    const float *tscale0, *tshift0, *tscale1, *tshift1;
    int32_t *counters;          // K split: one ticket per output block
    // optional per-channel statistics of what `out` receives (before its out-affine): stats[(pixel block) * Cout + co] =
    // (sum, sum of squares) over the block's 8 x PWO pixels -- what a GroupNorm of the output needs, without re-reading it
    float2 *stats;
    size_t split_stride;
    float wscale;               // 2^-S: the weights were packed as w * 2^S
    int B, H, W, C1, C2, Cout, up, act, oact, aff_sb;
    int th, tw;                 // output patches (8 x PWO) per image: rows, columns
    int ntn;                    // 64-channel output blocks
    int nchunks, nchunks1;      // channel chunks in total / in x
    int ksplit, chunks_per_split;
    int hp_shift;               // stacked edits (sige_hip_set_edit_batch): log2 of one image's height, 0 = off
#ifdef SIGE_WIDE_PROBE
    unsigned long long *probe;  // tools/probe/wide_phase_probe.py build only: 8 timestamps per workgroup
#endif
};

// Phase timestamps (s_memtime) of workgroups 0..4095, lane 0 -- compiled in only for the measurement build
// (SIGE_PROBE_WIDE=1 python -m sige_amd.build --probe); the product library contains none of this.
#ifdef SIGE_WIDE_PROBE
#define SIGE_WPROBE(k)                                                                                      \
    do {                                                                                                    \
        if (a.probe && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096)                            \
            a.probe[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime();                                   \
    } while (0)
#else
#define SIGE_WPROBE(k)
#endif

__device__ __forceinline__ f16x8 buf_h8(rsrc_t r, unsigned byte_off, int soff) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soff, 0));
    return __builtin_bit_cast(f16x8, v);
}

// One workgroup of the launch: (bx, by) = what blockIdx would be in a launch of its own (conv_wide_pair_kernel shares a launch
// between this body and a tile-kernel body).  AFF: the staging path applies scale * x + shift (and SiLU if a.act); CAT: channels
// from two tensors.
template <typename G, bool AFF, bool CAT>
__device__ __forceinline__ void conv_wide_body(const WideArgs &a, const int bx, const int by, unsigned char *const smem) {
    constexpr bool X3 = G::X3, F32 = G::F32;
    constexpr int NS = G::NS, STEPS = G::STEPS, RB = G::RB, NP = G::NP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SIGE_WPROBE(0);  // entry
    // consecutive workgroups (= consecutive XCDs) take different output-channel blocks; all pixel blocks of one channel
    // block land on XCD (ntile mod 8) when ntn is a multiple of 8: a layer's weights are fetched into one L2 each
    const int ntile = bx % a.ntn, mtile = bx / a.ntn;
    const int split = by;
    const int first = split * a.chunks_per_split;
    const int last = min(a.nchunks, first + a.chunks_per_split) - 1;
    const int tpi = a.th * a.tw;
    const int b = mtile / tpi, tr = mtile - b * tpi;
    const int ph0 = (tr / a.tw) * 8, pw0 = (tr % a.tw) * G::PWO;
    const int Hs = a.H >> a.up, Ws = a.W >> a.up;

    // ---- staging slots: slot i of this lane = float4 unit v = lane + 64 i of the wave's patch [pixel][QP] ----
    unsigned voff[NS], voff2[CAT ? NS : 1];
    int ldsw[NS];
    unsigned livemask = 0;
    // (stacked edits: rows of the halo patch beyond THIS patch's image are zero padding; patches never straddle a seam)
    const int hlo = a.hp_shift ? ((ph0 >> a.hp_shift) << a.hp_shift) : 0;
    const int hhi = a.hp_shift ? hlo + (1 << a.hp_shift) : a.H;
    static_for<0, NS>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const int v = lane + 64 * i;
        const int p = v / G::QP, c4 = v % G::QP;
        const int hy = p / G::PW, hx = p - hy * G::PW;
        const int h = ph0 + hy - (G::KH == 3 ? 1 : 0), w = pw0 + hx - (G::KH == 3 ? 1 : 0);
        const bool in = v < G::UNITS && h >= hlo && h < hhi && w >= 0 && w < a.W;
        const int spx = (b * Hs + (h >> a.up)) * Ws + (w >> a.up);
        const int cb = wave * G::CW + c4 * 4;
        voff[i] = in ? (unsigned)(spx * a.C1 + cb) * 4u : kOOB;
        if constexpr (CAT) voff2[i] = in ? (unsigned)(spx * a.C2 + cb) * 4u : kOOB;
        livemask |= in ? (1u << i) : 0u;
        ldsw[i] = p * G::ROWB + (c4 >> 2) * G::KSB + (c4 & 3) * 8;
    });
    unsigned char *const mybuf = smem + wave * 2 * G::ABUF;

    auto a_rsrc = [&](int chunk) -> rsrc_t {
        if (!CAT || chunk < a.nchunks1) return make_rsrc(a.x, (long)chunk * G::CC, (long)a.B * Hs * Ws * a.C1);
        return make_rsrc(a.x2, (long)(chunk - a.nchunks1) * G::CC, (long)a.B * Hs * Ws * a.C2);
    };
    float4 st[NS];
    auto a_load = [&](int chunk) {
        const rsrc_t r = a_rsrc(chunk);
        const bool use2 = CAT && chunk >= a.nchunks1;
        static_for<0, NS>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            unsigned o = voff[i];
            if constexpr (CAT) o = use2 ? voff2[i] : o;
            st[i] = buf_f32x4(r, o, 0);
        });
    };
    // affine entries of this lane's 4 channels (the same 4 in every slot: 64 % QP == 0)
    const int cbl = wave * G::CW + (lane % G::QP) * 4;
    auto aff_load = [&](int chunk, float4 &sc, float4 &sh) {
        if constexpr (AFF) {
            const int c = b * a.aff_sb + chunk * G::CC + cbl;
            sc = *reinterpret_cast<const float4 *>(a.scale + c);
            sh = *reinterpret_cast<const float4 *>(a.shift + c);
        }
    };
    // finish one slot -- scale, then shift, then SiLU, separately rounded like the reference (gather.cpp:33-53); padding
    // stays an exact 0 (gather.cpp:27-30) -- split it into fp16 planes and store it to the stage
    const bool do_act = a.act == SIGE_HIP_ACT_SWISH;
    auto fin = [&](float z, float sc, float sh, bool live) -> float {
        if constexpr (AFF) {
            z = sc * z;
            z = sh + z;
            if (do_act) z = swish_fast(z);
            z = live ? z : 0.0f;
        }
        if constexpr (F32) return z;
        return __builtin_fminf(__builtin_fmaxf(z, -65504.0f), 65504.0f);  // (fp16 range: saturate instead of +-inf)
    };
    auto a_store = [&](auto i_tag, unsigned char *buf, const float4 sc, const float4 sh) {
        constexpr int i = decltype(i_tag)::value;
        if (64 * (i + 1) > G::UNITS && lane >= G::UNITS - 64 * i) return;  // (ragged last slot)
        const bool live = (livemask >> i) & 1u;
        const float4 q = st[i];
        const float z0 = fin(q.x, sc.x, sh.x, live), z1 = fin(q.y, sc.y, sh.y, live);
        const float z2 = fin(q.z, sc.z, sh.z, live), z3 = fin(q.w, sc.w, sh.w, live);
        if constexpr (F32) {
            // exact fp32: the even channels of the unit go to lane group 0's half of the row, the odd ones to group 1's
            *reinterpret_cast<float2 *>(buf + ldsw[i]) = make_float2(z0, z2);
            *reinterpret_cast<float2 *>(buf + ldsw[i] + 32) = make_float2(z1, z3);
        } else {
            const f16x4 hi = {(_Float16)z0, (_Float16)z1, (_Float16)z2, (_Float16)z3};  // RNE
            *reinterpret_cast<f16x4 *>(buf + ldsw[i]) = hi;
            if constexpr (X3) {
                const f16x4 lo = {(_Float16)(z0 - (float)hi[0]), (_Float16)(z1 - (float)hi[1]),
                                  (_Float16)(z2 - (float)hi[2]), (_Float16)(z3 - (float)hi[3])};
                *reinterpret_cast<f16x4 *>(buf + ldsw[i] + 32) = lo;
            }
        }
    };

    // ---- B: this wave's stream of packed weights, contiguous over (chunk, k-step) ----
    const long stream_bytes = (long)a.nchunks * STEPS * G::STEPB;
    const unsigned char *const bstream = reinterpret_cast<const unsigned char *>(a.packed) +
                                         ((long)(ntile * 4 + wave) * a.nchunks + first) * STEPS * G::STEPB;
    // (range: the rest of the packed tensor from here on; the allocation carries RB steps of padding for the prefetch past `last`)
    const long left = (long)a.ntn * 4 * stream_bytes - ((long)(ntile * 4 + wave) * a.nchunks + first) * STEPS * G::STEPB + (long)kWidePadSteps * G::STEPB;
    const rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(bstream), 0,
                                                         __builtin_amdgcn_readfirstlane((int)(left > 0x7fffffffL ? 0x7fffffffL : left)), 0x00020000);
    f16x8 bring[RB][2][NP];
    auto b_issue = [&](auto slot_tag, int g) {  // k-step g of this workgroup's slice -> ring position `slot`
        constexpr int slot = decltype(slot_tag)::value;
        const int soff = __builtin_amdgcn_readfirstlane(g * G::STEPB);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bring[slot][nt][pl] = buf_h8(r_b, lane * 16 + (nt * NP + pl) * 1024, soff);
    };

    // ---- A operand of this lane: pixel i of M tile mt (rows 4 mt .. 4 mt + 3 of the patch), k-group kq ----
    const int i32 = lane & 31, kq = lane >> 5;
    const int abase = ((i32 / G::PWO) * G::PW + (i32 % G::PWO)) * G::ROWB + kq * G::KQB;  // + mt * RPT * PW * ROWB
    constexpr int MTN = G::MTN;
    struct AHalf { f16x8 v[G::MTN]; };
    // plane 0 = hi, 1 = lo (32 bytes further in the pixel's row)
    auto a_read = [&](auto s_tag, const unsigned char *buf, int plane) -> AHalf {
        constexpr int s = decltype(s_tag)::value;
        constexpr int ks = s / G::KK, tap = s % G::KK;
        constexpr int off = ((tap / G::KH) * G::PW + tap % G::KH) * G::ROWB + ks * G::KSB;
        AHalf r;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
            r.v[mt] = *reinterpret_cast<const f16x8 *>(buf + abase + mt * G::RPT * G::PW * G::ROWB + off + plane * G::PLB);
        return r;
    };

    f32x16 acc[MTN][2];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // ---- prologue: weights of the first RB steps, chunk `first` -> stage 0, chunk first+1 -> registers ----
    float4 sc_c = make_float4(0.f, 0.f, 0.f, 0.f), sh_c = sc_c, sc_n = sc_c, sh_n = sc_c;
    a_load(first);
    aff_load(first, sc_c, sh_c);
    static_for<0, RB>([&](auto g_tag) { b_issue(g_tag, decltype(g_tag)::value); });
    aff_load(min(first + 1, last), sc_n, sh_n);
    static_for<0, NS>([&](auto i_tag) { a_store(i_tag, mybuf, sc_c, sh_c); });
    a_load(min(first + 1, last));
    __builtin_amdgcn_wave_barrier();
    AHalf a_hi = a_read(std::integral_constant<int, 0>{}, mybuf, 0), a_lo = a_hi;
    if constexpr (NP == 2) a_lo = a_read(std::integral_constant<int, 0>{}, mybuf, 1);

    // one chunk: MFMAs on stage PAR; the registers holding chunk+1 are finished into stage PAR^1 and re-issued as chunk+2;
    // every ring position is re-issued RB steps ahead right after its MFMAs
    auto body = [&](auto par_tag, int chunk) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char *cur = mybuf + PAR * G::ABUF;
        unsigned char *nxt = mybuf + (PAR ^ 1) * G::ABUF;
        const int c2 = min(chunk + 2, last);
        const int g0 = (chunk - first) * STEPS;
        float4 sc_t = sc_n, sh_t = sh_n;
        const rsrc_t r_a2 = a_rsrc(c2);
        const bool use2 = CAT && c2 >= a.nchunks1;
        static_for<0, STEPS>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            constexpr int slot = (PAR * STEPS + s) % RB;
            AHalf n_hi = a_hi, n_lo = a_lo;
            if constexpr (s + 1 < STEPS) n_hi = a_read(std::integral_constant<int, s + 1>{}, cur, 0);
            if constexpr (F32) {
                // exact fp32: eight v_mfma_f32_32x32x2_f32 per tile contract the 16 channels (k-step ss: channels 2 ss + kq);
                // the four tiles alternate, so consecutive MFMAs never share an accumulator
                if constexpr (s + 1 < STEPS) n_lo = a_read(std::integral_constant<int, s + 1>{}, cur, 1);
                static_for<0, 8>([&](auto ss_tag) {
                    constexpr int ss = decltype(ss_tag)::value;
#pragma unroll
                    for (int mt = 0; mt < MTN; ++mt) {
                        const f32x4 av = __builtin_bit_cast(f32x4, ss < 4 ? a_hi.v[mt] : a_lo.v[mt]);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const f32x4 bv = __builtin_bit_cast(f32x4, bring[slot][nt][ss / 4]);
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss & 3], bv[ss & 3], acc[mt][nt], 0, 0, 0);
                        }
                    }
                });
            } else {
            // lo*hi first: the lo operands die after the first group and next step's lo takes their registers; then hi*hi and
            // hi*lo.  Consecutive MFMAs never share an accumulator.
            if constexpr (X3) {
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo.v[mt], bring[slot][nt][0], acc[mt][nt], 0, 0, 0);
                if constexpr (s + 1 < STEPS) n_lo = a_read(std::integral_constant<int, s + 1>{}, cur, 1);
            }
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi.v[mt], bring[slot][nt][0], acc[mt][nt], 0, 0, 0);
            if constexpr (X3) {
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi.v[mt], bring[slot][nt][NP - 1], acc[mt][nt], 0, 0, 0);
            }
            }
            // staging slots of chunk+1 spread over the steps before the last one
            if constexpr (s < STEPS - 1 || STEPS == 1) {
                constexpr int SD = STEPS > 1 ? STEPS - 1 : 1;
                static_for<(s * NS) / SD, ((s + 1) * NS) / SD>([&](auto i_tag) {
                    constexpr int i = decltype(i_tag)::value;
                    a_store(i_tag, nxt, sc_t, sh_t);
                    unsigned o = voff[i];
                    if constexpr (CAT) o = use2 ? voff2[i] : o;
                    st[i] = buf_f32x4(r_a2, o, 0);
                });
                if constexpr (s == 0) aff_load(c2, sc_n, sh_n);
            }
            b_issue(std::integral_constant<int, slot>{}, g0 + s + RB);
            // (the machine scheduler otherwise sinks every prefetch down to its first use -- measured in the ISA: load,
            //  s_waitcnt vmcnt(0), MFMA -- to shorten live ranges; nothing may move across a step boundary)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s + 1 == STEPS) {
                __builtin_amdgcn_wave_barrier();  // (the stage was written by other lanes of this wave: LDS is in order per wave)
                n_hi = a_read(std::integral_constant<int, 0>{}, nxt, 0);
                if constexpr (NP == 2) n_lo = a_read(std::integral_constant<int, 0>{}, nxt, 1);
            }
            a_hi = n_hi;
            a_lo = n_lo;
        });
    };
    SIGE_WPROBE(1);  // prologue done: first stage in LDS, weight ring filled
    for (int chunk = first; chunk <= last; chunk += 2) {
        body(std::integral_constant<int, 0>{}, chunk);
        if (chunk + 1 <= last) body(std::integral_constant<int, 1>{}, chunk + 1);
    }

    SIGE_WPROBE(2);  // K loop done (this wave)
    // ---- reduction of the four waves' K shares through LDS ----
    __syncthreads();  // (the reduction buffer overlaps the other waves' stages)
    SIGE_WPROBE(3);  // every wave's K loop done
    constexpr int RP = 68;
    float *const red = reinterpret_cast<float *>(smem);
    {
        float *r = red + wave * G::BM * RP + i32;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * kq;
                    r[m * RP + nt * 32] = acc[mt][nt][q];
                }
    }
    __syncthreads();

    SIGE_WPROBE(4);  // accumulators in LDS
    // ---- epilogue: one float4 = 4 consecutive output channels of one pixel per lane and step ----
    const bool split_k = a.ksplit > 1;
    float *const outp = a.out + (size_t)split * a.split_stride;
    struct Unit { bool ok; size_t addr; int co; };
    auto locate = [&](int k) -> Unit {
        const int o = tid + 256 * k;
        const int n4 = o & 15, m = o >> 4;
        const int h = ph0 + m / G::PWO, w = pw0 + m % G::PWO;
        Unit u;
        u.co = ntile * 64 + 4 * n4;
        u.ok = h < a.H && w < a.W && u.co < a.Cout;
        u.addr = u.ok ? (((size_t)b * a.H + h) * a.W + w) * a.Cout + u.co : 0;
        return u;
    };
    // (this lane's units all have the same four output channels: n4 = tid & 15)
    float4 st_s = make_float4(0.f, 0.f, 0.f, 0.f), st_q = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](const Unit &u, float4 s) {
        s.x *= a.wscale; s.y *= a.wscale; s.z *= a.wscale; s.w *= a.wscale;  // (a power of two: exact)
        if (a.bias) {
            const float4 bb = *reinterpret_cast<const float4 *>(a.bias + u.co);
            s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
        }
        if (a.residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(a.residual + u.addr);
            s.x += rr.x; s.y += rr.y; s.z += rr.z; s.w += rr.w;
        }
        if (a.stats) {
            st_s.x += s.x; st_s.y += s.y; st_s.z += s.z; st_s.w += s.w;
            st_q.x += s.x * s.x; st_q.y += s.y * s.y; st_q.z += s.z * s.z; st_q.w += s.w * s.w;
        }
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
        if (a.oscale) {
            const float4 os = *reinterpret_cast<const float4 *>(a.oscale + u.co), oh = *reinterpret_cast<const float4 *>(a.oshift + u.co);
            s.x = os.x * s.x; s.y = os.y * s.y; s.z = os.z * s.z; s.w = os.w * s.w;
            s.x = oh.x + s.x; s.y = oh.y + s.y; s.z = oh.z + s.z; s.w = oh.w + s.w;
            if (a.oact == SIGE_HIP_ACT_SWISH) { s.x = swish(s.x); s.y = swish(s.y); s.z = swish(s.z); s.w = swish(s.w); }
        }
        store_out4((split_k ? a.fout : a.out) + u.addr, s);  // (write-through: common.hpp)
    };
    // the workgroup's (sum, sum of squares) per output channel: 16 lanes per channel quad (4 in each wave) -> LDS -> one row
    // of `stats`; fixed order, no atomics: the same launch gives the same bits every time
    auto flush_stats = [&]() {
        if (!a.stats) return;  // (uniform)
        __syncthreads();       // every lane is done with the reduction buffer
        float *sb = red;       // [wave][16 quads][8]
#pragma unroll
        for (int d = 16; d < 64; d <<= 1) {
            st_s.x += __shfl_xor(st_s.x, d); st_s.y += __shfl_xor(st_s.y, d); st_s.z += __shfl_xor(st_s.z, d); st_s.w += __shfl_xor(st_s.w, d);
            st_q.x += __shfl_xor(st_q.x, d); st_q.y += __shfl_xor(st_q.y, d); st_q.z += __shfl_xor(st_q.z, d); st_q.w += __shfl_xor(st_q.w, d);
        }
        if (lane < 16) {
            float *d = sb + (wave * 16 + lane) * 8;
            *reinterpret_cast<float4 *>(d) = st_s;
            *reinterpret_cast<float4 *>(d + 4) = st_q;
        }
        __syncthreads();
        if (tid < 64) {
            const int q = tid >> 2, e = tid & 3;  // channel quad, channel inside it
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += sb[(w * 16 + q) * 8 + e]; s2 += sb[(w * 16 + q) * 8 + 4 + e]; }
            const int co = ntile * 64 + tid;
            if (co < a.Cout) a.stats[(size_t)mtile * a.Cout + co] = make_float2(s1, s2);
        }
    };
    constexpr int EU = G::BM * 16 / 256;  // float4 units per lane
#pragma unroll
    for (int k = 0; k < EU; ++k) {
        const Unit u = locate(k);
        const int o = tid + 256 * k;
        const float *r0 = red + (o >> 4) * RP + 4 * (o & 15);
        float4 s = *reinterpret_cast<const float4 *>(r0);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(r0 + w * G::BM * RP);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (!u.ok) continue;
        if (!split_k) emit(u, s);
        else coherent_store(outp + u.addr, s);
    }
    SIGE_WPROBE(5);  // output stores issued
    if (!split_k) flush_stats();
    SIGE_WPROBE(6);  // statistics written
    if (split_k) {
        // partial sums went out as device-coherent stores (complete = visible to every XCD); then the block's ticket; the
        // workgroup that draws the last one adds the copies in split order (deterministic) and runs the epilogue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int32_t *const cnt = a.counters + bx;
        if (tid == 0) red[0] = __builtin_bit_cast(float, __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        const int ticket = __builtin_bit_cast(int, red[0]);
        if (ticket == a.ksplit - 1) {
            if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the copies of this lane's units (four units at a time), four splits in flight at a time, added in split order
#pragma unroll
            for (int k0 = 0; k0 < EU; k0 += 4) {
                Unit us[4];
                float4 sum[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { us[k] = locate(k0 + k); sum[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
                for (int s0 = 0; s0 < a.ksplit; s0 += 4) {
                    float4 pv[4][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            pv[k][d] = coherent_load(a.out + (size_t)(s0 + d < a.ksplit ? s0 + d : a.ksplit - 1) * a.split_stride + us[k].addr);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            if (s0 + d < a.ksplit) { sum[k].x += pv[k][d].x; sum[k].y += pv[k][d].y; sum[k].z += pv[k][d].z; sum[k].w += pv[k][d].w; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (us[k].ok) emit(us[k], sum[k]);
            }
            flush_stats();
        }
    }
}

template <typename G, bool AFF, bool CAT>
__global__ __launch_bounds__(256, G::OCC) void conv_wide_kernel(const WideArgs a) {
    kernarg_touch<sizeof(WideArgs)>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
    conv_wide_body<G, AFF, CAT>(a, blockIdx.x, blockIdx.y, smem);
}

// A dense residual block's conv1 (this kernel, 3x3) and its 1x1 shortcut (the tile kernel's body, conv_mfma.hpp) in ONE launch:
// workgroups [0, na) run the dense-layer conv (possibly K-split over gridDim.y), the rest the shortcut (blockIdx.y == 0 only) --
// the horizontal fusion of conv_pair_kernel for layers routed to this kernel (round 4).  Both bodies are 256
// lanes; LDS = the larger of the two; the register budget is this kernel's (two workgroups per CU).
template <typename G, bool AFF, bool CAT, typename GB>
__global__ __launch_bounds__(256, G::OCC) void conv_wide_pair_kernel(const WideArgs a, const ConvArgs b, const int na) {
    kernarg_touch<sizeof(WideArgs) + sizeof(ConvArgs)>();
    constexpr int LB = conv_lds_floats<GB, 1, MODE_RAW, LAYOUT_NHWC, 4>() * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[cmax(G::LDS_BYTES, LB)];
    if ((int)blockIdx.x < na) conv_wide_body<G, AFF, CAT>(a, blockIdx.x, blockIdx.y, smem);
    else if (blockIdx.y == 0)
        conv_mfma_body<GB, 1, SRC_GATHER, MODE_RAW, DST_NCHW, LAYOUT_NHWC, 4>(b, blockIdx.x - na, 0, reinterpret_cast<float *>(smem));
}

template <int PREC, typename GB>
void launch_conv_wide_pair(const WideArgs &a, bool aff, bool cat, ConvArgs b, hipStream_t st);

// (block_conv.hip) the conv-pair state of the calling thread, as far as a dense-layer launch needs it
int held_shortcut_prec(hipStream_t st);
bool take_held_shortcut(ConvArgs *b, int *mt);
int flush_held_conv();

#define SIGE_WIDE_PAIR_INSTANTIATE(PREC, GB)                                                               \
    template <> void launch_conv_wide_pair<PREC, GB>(const WideArgs &a, bool aff, bool cat, ConvArgs b, hipStream_t st) { \
        using G = WideGeo<3, PREC, 8>;                                                                     \
        const int na = a.B * a.th * a.tw * a.ntn;                                                          \
        const dim3 grid(na + conv_grid_x(b), a.ksplit);                                                    \
        if (aff && cat) conv_wide_pair_kernel<G, true, true, GB><<<grid, 256, 0, st>>>(a, b, na);          \
        else if (aff) conv_wide_pair_kernel<G, true, false, GB><<<grid, 256, 0, st>>>(a, b, na);           \
        else if (cat) conv_wide_pair_kernel<G, false, true, GB><<<grid, 256, 0, st>>>(a, b, na);           \
        else conv_wide_pair_kernel<G, false, false, GB><<<grid, 256, 0, st>>>(a, b, na);                   \
    }

template <int KH, int PREC, int PWO>
void launch_conv_wide(const WideArgs &a, bool aff, bool cat, hipStream_t st);

#define SIGE_WIDE_INSTANTIATE(KH, PREC, PWO)                                                               \
    template <> void launch_conv_wide<KH, PREC, PWO>(const WideArgs &a, bool aff, bool cat, hipStream_t st) { \
        using G = WideGeo<KH, PREC, PWO>;                                                                  \
        const dim3 grid(a.B * a.th * a.tw * a.ntn, a.ksplit);                                              \
        if (aff && cat) conv_wide_kernel<G, true, true><<<grid, 256, 0, st>>>(a);                          \
        else if (aff) conv_wide_kernel<G, true, false><<<grid, 256, 0, st>>>(a);                           \
        else if (cat) conv_wide_kernel<G, false, true><<<grid, 256, 0, st>>>(a);                           \
        else conv_wide_kernel<G, false, false><<<grid, 256, 0, st>>>(a);                                   \
    }

}  // namespace sige
