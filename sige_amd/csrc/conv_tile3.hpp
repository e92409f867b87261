// Tile conv v3: the 3x3 / stride-1 stacked-block conv with conv_wide.hpp's K loop -- wave-PRIVATE stages (no workgroup barrier
// per channel chunk), a weight register ring streamed RB steps ahead, a 64-channel output block per workgroup -- over the
// 6x6 windows of a SIGE tile list instead of an 8x8 patch of a dense image.
//
//   out = epilogue( sum_{ci,tap} stage(ci, window pixel + tap) * w[co, ci, tap] )     exact fp32: v_mfma_f32_32x32x2_f32
//
// Why: conv_mfma.hpp shares ONE stage between the four waves of a workgroup, so every
// 32-channel chunk costs a workgroup barrier, and a 32-pixel x 32-channel block streams a weight byte per 32 pixels.  On full
// grids (>= 2 workgroups per CU: 15 % edits, stacked edits) that structure holds the tile kernel at 0.44-0.55 of the fp32 MFMA
// peak where conv_wide's exact-fp32 form reaches 0.81-0.86 on dense layers of the same size.  Here a workgroup is TPW tiles
// (2 -> 32 pixels, 46 KB of LDS: two or three workgroups per CU) x 64 output channels; the four waves split K -- each stages
// and contracts ITS 16 channels of every 64-channel chunk -- and meet once, in LDS, before the epilogue.
// Launch-bound grids (one workgroup per CU: a 1.2 % edit) stay on conv_mfma.hpp, whose 16- and 32-channel blocks and in-launch
// K split fill the chip from 18 tiles on; the host routes (sige_amd/hip.py: TILE3_MIN_BLOCKS).
//
// Staging sources (the reference's Gather / ScatterGather fused in front of its F.conv2d, sige/nn/base.py:87-89):
//   T3_GATHER          tiles of a full tensor (two tensors = a fused torch.cat, x2 nearest upsampling in the addressing),
//                      zero padding outside the image, optional cached affine + SiLU (sige/cpu/gather.cpp:4-58)
//   T3_SCATTER_GATHER  conv-1 tiles where the scatter map has one, the cached tensor elsewhere (scatter_gather.cpp:5-56), raw
// Destinations: tiles [T,4,4,Cout], or straight into a full tensor (Scatter / ScatterWithBlockResidual fused behind the conv,
// sige/cpu/scatter.cpp:4-68) with bias, residual, block residual, activated twins and the consumer's out-affine.
#pragma once
#include "conv_wide.hpp"

namespace sige {

enum { T3_GATHER = 1, T3_SCATTER_GATHER = 2 };

// PREC_: WIDE_F32 -- exact fp32 (v_mfma_f32_32x32x2_f32); WIDE_F16 (round 6, BASELINE.json configs[4]) -- operands rounded to fp16
// (RNE) in the staging path, fp32 accumulation on v_mfma_f32_32x32x16_f16: ONE matrix instruction per tap and tile pair contracts
// the wave's 16 channels (eight in the fp32 form), the stage of a pixel is 32 bytes instead of 64 and a k-step of packed weights 2 KB
// instead of 4; a three-step weight ring as in the fp32 form (108-128 registers: three workgroups per CU).
template <int TPW_, int PREC_ = WIDE_F32>
struct Tile3Geo {
    static constexpr int TPW = TPW_;                   // tiles per workgroup
    static constexpr int PREC = PREC_;
    static constexpr bool F32 = PREC_ == WIDE_F32;
    static_assert(PREC_ == WIDE_F32 || PREC_ == WIDE_F16, "tile conv v3: exact fp32 or fp16 operands");
    static constexpr int KK = 9;
    static constexpr int MTN = TPW_ / 2;               // 32-pixel M tiles per wave (two 4x4 tiles each)
    static constexpr int BM = 16 * TPW_;               // output pixels per workgroup
    static constexpr int NP = F32 ? 2 : 1, CW = 16, CC = 64, STEPS = 9;
    static constexpr int NPX = 36 * TPW_;              // staged pixels: TPW windows of 6x6
    static constexpr int QP = CW / 4;                  // float4 units per staged pixel (per wave)
    static constexpr int UNITS = NPX * QP;
    static constexpr int NS = (UNITS + 63) / 64;       // staging slots per lane
    // exact fp32 row: 8 even channels | 8 odd channels (conv_wide.hpp WIDE_F32); fp16 row: 16 halves
    static constexpr int KSB = 32 * NP, KQB = F32 ? 32 : 16, PLB = F32 ? 16 : 32;
    static constexpr int ROWB = KSB + 16;              // LDS row of one staged pixel (padded against bank conflicts)
    static constexpr int ABUF = NPX * ROWB;            // bytes of one stage of one wave
    static constexpr int STEPB = 2 * NP * 1024;        // packed weight bytes per k-step of one wave (sige_hip_wide_conv_pack)
    // fp16: (workgroups per CU, ring) measured at (2, 9) / (3, 9: spills) / (3, 6) / (3, 3) / (4, 3: spills) -- profiles/r6l_*, r6m_*:
    // three workgroups per CU with the fp32 form's three-step ring is the fastest launch by launch (15 % edit, 256^2: gather
    // 17.8 -> 16.2 us, scatter_gather 25.4 -> 22.1 us against (2, 9)); -D overrides for A/B builds (tools/build_variant.py)
#ifndef SIGE_T3H_RB
#define SIGE_T3H_RB 3
#endif
#ifndef SIGE_T3H_OCC
#define SIGE_T3H_OCC 3
#endif
    static constexpr int RB = F32 ? 3 : SIGE_T3H_RB;   // weight ring, in k-steps
    // (fp32: 46 KB of LDS and <= 168 registers: three workgroups per CU)
    static constexpr int OCC = F32 ? (TPW_ == 2 ? 3 : 1) : (TPW_ == 2 ? SIGE_T3H_OCC : 2);  // (fp16, 4 tiles: 70 KB of LDS -- two workgroups per CU)
    static constexpr int LDS_BYTES = cmax(4 * 2 * ABUF, 4 * BM * 68 * 4);
    static_assert((2 * STEPS) % RB == 0, "the ring position of a step must not depend on the chunk");
};

struct Tile3Args {
    const float *x, *x2;        // GATHER: [B,Hs,Ws,C1] (+ [B,Hs,Ws,C2]: channels of x2 follow x) | SCATTER_GATHER: conv tiles [B*N,Rx,Sx,Cin], cached y [B,H,W,Cin]
    const int32_t *idx;         // [N,2] tile origins
    const int32_t *map;         // SCATTER_GATHER: [H,W,3]
    const void *packed;         // sige_hip_wide_conv_pack(prec = WIDE_F32)
    const float *bias, *scale, *shift, *residual, *oscale, *oshift;
    float *out;
    float *twin0, *twin1;
    const float *tscale0, *tshift0, *tscale1, *tshift1;
    const float *x1;            // block residual: shortcut tiles [B*N1,R1,S1,Cout] (then `residual` is the cached shortcut tensor)
    const int32_t *table1;
    int gW1, N1, R1, S1;
    int B, N, T, H, W, C1, C2, Cout, up, act, oact, aff_sb;
    int Rx, Sx;
    int Ho, Wo, offH, offW;     // full destination
    int ntn, nchunks, nchunks1;
    int hp_shift;
    int res_f16;                // `residual` (a fused ScatterWithBlockResidual's cached shortcut tensor) holds halves (fp16-stored caches)
};

// One workgroup: tiles [mtile * TPW, (mtile + 1) * TPW) x output channels [64 ntile, 64 ntile + 64).
// Y16: SCATTER_GATHER only -- the cached tensor (x2) holds halves (SIGEModel.set_cache_dtype("f16"))
template <typename G, int SRC, bool AFF, bool CAT, bool FULL, bool Y16 = false>
__device__ __forceinline__ void conv_tile3_body(const Tile3Args &a, const int bx, unsigned char *const smem) {
    constexpr int NS = G::NS, STEPS = G::STEPS, RB = G::RB, NP = G::NP, MTN = G::MTN;
    constexpr bool SG = SRC == T3_SCATTER_GATHER;
    constexpr bool F32 = G::F32;
    static_assert(!Y16 || SG, "fp16-stored cache: the scatter_gather source");
    constexpr bool TWO = CAT || SG;  // a slot may come from the second tensor
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = bx % a.ntn, mtile = bx / a.ntn;
    const int first = 0, last = a.nchunks - 1;
    const int Hs = a.H >> a.up, Ws = a.W >> a.up;
    const int Cin = a.C1 + a.C2;

    // ---- staging slots: slot i of this lane = float4 unit v = lane + 64 i of the wave's stage [staged pixel][QP] ----
    unsigned voff[NS], voff2[TWO ? NS : 1];
    int ldsw[NS];
    unsigned livemask = 0;
    int z_h[NS], z_w[NS], z_b[NS];
    bool z_ok[NS];
    static_for<0, NS>([&](auto i_tag) {  // (1) tile origins: all loads issued before the first one is used
        constexpr int i = decltype(i_tag)::value;
        const int v = lane + 64 * i;
        const int p = v / G::QP;
        const int tl = p / 36, q = p - tl * 36;
        const int t = mtile * G::TPW + tl;
        z_ok[i] = v < G::UNITS && t < a.T;
        const int tt = min(t, a.T - 1);
        const int b = tt / a.N, n = tt - b * a.N;
        const int2 o = *reinterpret_cast<const int2 *>(a.idx + 2 * n);
        z_b[i] = b;
        z_h[i] = o.x + q / 6;
        z_w[i] = o.y + q % 6;
        int hlo = 0, hhi = a.H;
        if (a.hp_shift) {  // stacked edits: rows beyond the tile's own image are zero padding (its window's third row is inside it)
            hlo = ((o.x + 2) >> a.hp_shift) << a.hp_shift;
            hhi = hlo + (1 << a.hp_shift);
        }
        z_ok[i] = z_ok[i] && z_h[i] >= hlo && z_h[i] < hhi && z_w[i] >= 0 && z_w[i] < a.W;
        ldsw[i] = p * G::ROWB + (v % G::QP) * 8;
    });
    int z_m0[SG ? NS : 1], z_m1[SG ? NS : 1], z_m2[SG ? NS : 1];
    if constexpr (SG) {
        static_for<0, NS>([&](auto i_tag) {  // (2) scatter-map entries, batched
            constexpr int i = decltype(i_tag)::value;
            const int hw = z_ok[i] ? z_h[i] * a.W + z_w[i] : 0;
            const int32_t *m = a.map + 3 * (size_t)hw;
            z_m0[i] = m[0]; z_m1[i] = m[1]; z_m2[i] = m[2];
        });
    }
    static_for<0, NS>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        const int v = lane + 64 * i;
        const int cb = wave * G::CW + (v % G::QP) * 4;
        const bool in = z_ok[i];
        if constexpr (SG) {
            const int blk = z_m0[i];
            voff[i] = (in && blk >= 0) ? (unsigned)((((z_b[i] * a.N + blk) * a.Rx + z_m1[i]) * a.Sx + z_m2[i]) * Cin + cb) * 4u : kOOB;
            voff2[i] = (in && blk < 0) ? (unsigned)(((z_b[i] * a.H + z_h[i]) * a.W + z_w[i]) * Cin + cb) * (Y16 ? 2u : 4u) : kOOB;
        } else {
            const int spx = (z_b[i] * Hs + (z_h[i] >> a.up)) * Ws + (z_w[i] >> a.up);
            voff[i] = in ? (unsigned)(spx * a.C1 + cb) * 4u : kOOB;
            if constexpr (CAT) voff2[i] = in ? (unsigned)(spx * a.C2 + cb) * 4u : kOOB;
        }
        livemask |= in ? (1u << i) : 0u;
    });
    unsigned char *const mybuf = smem + wave * 2 * G::ABUF;

    float4 st[NS];
    // descriptors of one chunk: GATHER: the tensor the chunk's channels come from (x, or x2 behind a fused cat);
    // SCATTER_GATHER: the conv tiles AND the cached tensor (a slot reads both; the one it does not come from is out of range: 0)
    auto rsrc_a = [&](int chunk) -> rsrc_t {
        if constexpr (SG) return make_rsrc(a.x, (long)chunk * G::CC, (long)a.T * a.Rx * a.Sx * Cin);
        if (CAT && chunk >= a.nchunks1) return make_rsrc(a.x2, (long)(chunk - a.nchunks1) * G::CC, (long)a.B * Hs * Ws * a.C2);
        return make_rsrc(a.x, (long)chunk * G::CC, (long)a.B * Hs * Ws * a.C1);
    };
    auto rsrc_y = [&](int chunk) -> rsrc_t {
        if constexpr (Y16) return make_rsrc_h(a.x2, (long)chunk * G::CC, (long)a.B * a.H * a.W * Cin);
        return make_rsrc(a.x2, (long)chunk * G::CC, (long)a.B * a.H * a.W * Cin);
    };
    auto slot_load = [&](auto i_tag, const rsrc_t r, const rsrc_t ry, const bool use2) {
        constexpr int i = decltype(i_tag)::value;
        if constexpr (SG) {
            const float4 p = buf_f32x4(r, voff[i], 0);
            float4 q;
            if constexpr (Y16) q = buf_h4(ry, voff2[i], 0);
            else q = buf_f32x4(ry, voff2[i], 0);
            st[i] = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        } else {
            unsigned o = voff[i];
            if constexpr (CAT) o = use2 ? voff2[i] : o;
            st[i] = buf_f32x4(r, o, 0);
        }
    };
    auto a_load = [&](int chunk) {
        const rsrc_t r = rsrc_a(chunk);
        const rsrc_t ry = SG ? rsrc_y(chunk) : r;
        const bool use2 = CAT && chunk >= a.nchunks1;
        static_for<0, NS>([&](auto i_tag) { slot_load(i_tag, r, ry, use2); });
    };
    // affine entries of this lane's 4 channels (the same 4 in every slot: 64 % QP == 0); per batch: a workgroup's tiles may belong
    // to different images only when aff_sb == 0 (host side)
    const int cbl = wave * G::CW + (lane % G::QP) * 4;
    const int b0 = min(mtile * G::TPW, a.T - 1) / a.N;
    auto aff_load = [&](int chunk, float4 &sc, float4 &sh) {
        if constexpr (AFF) {
            const int c = b0 * a.aff_sb + chunk * G::CC + cbl;
            sc = *reinterpret_cast<const float4 *>(a.scale + c);
            sh = *reinterpret_cast<const float4 *>(a.shift + c);
        }
    };
    const bool do_act = a.act == SIGE_HIP_ACT_SWISH;
    auto fin = [&](float z, float sc, float sh, bool live) -> float {
        if constexpr (AFF) {  // scale, then shift, then SiLU, separately rounded (gather.cpp:33-53); padding stays an exact 0
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
            // the even channels of the unit go to lane group 0's half of the row, the odd ones to group 1's
            *reinterpret_cast<float2 *>(buf + ldsw[i]) = make_float2(z0, z2);
            *reinterpret_cast<float2 *>(buf + ldsw[i] + 32) = make_float2(z1, z3);
        } else {
            const f16x4 hv = {(_Float16)z0, (_Float16)z1, (_Float16)z2, (_Float16)z3};  // RNE
            *reinterpret_cast<f16x4 *>(buf + ldsw[i]) = hv;
        }
    };

    // ---- B: this wave's stream of packed weights, contiguous over (chunk, k-step) ----
    const long stream_bytes = (long)a.nchunks * STEPS * G::STEPB;
    const unsigned char *const bstream = reinterpret_cast<const unsigned char *>(a.packed) + ((long)(ntile * 4 + wave) * a.nchunks + first) * STEPS * G::STEPB;
    const long left = (long)a.ntn * 4 * stream_bytes - ((long)(ntile * 4 + wave) * a.nchunks + first) * STEPS * G::STEPB + (long)kWidePadSteps * G::STEPB;
    const rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(bstream), 0,
                                                         __builtin_amdgcn_readfirstlane((int)(left > 0x7fffffffL ? 0x7fffffffL : left)), 0x00020000);
    f16x8 bring[RB][2][NP];
    auto b_issue = [&](auto slot_tag, int g) {
        constexpr int slot = decltype(slot_tag)::value;
        const int soff = __builtin_amdgcn_readfirstlane(g * G::STEPB);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bring[slot][nt][pl] = buf_h8(r_b, lane * 16 + (nt * NP + pl) * 1024, soff);
    };

    // ---- A operand of this lane: pixel i32 of M tile mt = tiles 2 mt, 2 mt + 1 of the workgroup; k-group kq ----
    const int i32 = lane & 31, kq = lane >> 5;
    const int abase = ((i32 >> 4) * 36 + ((i32 & 15) >> 2) * 6 + (i32 & 3)) * G::ROWB + kq * G::KQB;  // + mt * 72 * ROWB
    struct AHalf { f16x8 v[MTN]; };
    auto a_read = [&](auto s_tag, const unsigned char *buf, int plane) -> AHalf {
        constexpr int tap = decltype(s_tag)::value;
        constexpr int off = ((tap / 3) * 6 + tap % 3) * G::ROWB;
        AHalf r;
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) r.v[mt] = *reinterpret_cast<const f16x8 *>(buf + abase + mt * 72 * G::ROWB + off + plane * G::PLB);
        return r;
    };

    f32x16 acc[MTN][2];
#pragma unroll
    for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // ---- prologue: weights of the first RB steps, chunk 0 -> stage 0, chunk 1 -> registers ----
    float4 sc_c = make_float4(0.f, 0.f, 0.f, 0.f), sh_c = sc_c, sc_n = sc_c, sh_n = sc_c;
    a_load(first);
    aff_load(first, sc_c, sh_c);
    static_for<0, RB>([&](auto g_tag) { b_issue(g_tag, decltype(g_tag)::value); });
    aff_load(min(first + 1, last), sc_n, sh_n);
    static_for<0, NS>([&](auto i_tag) { a_store(i_tag, mybuf, sc_c, sh_c); });
    a_load(min(first + 1, last));
    __builtin_amdgcn_wave_barrier();
    AHalf a_hi = a_read(std::integral_constant<int, 0>{}, mybuf, 0), a_lo = a_hi;
    if constexpr (F32) a_lo = a_read(std::integral_constant<int, 0>{}, mybuf, 1);

    auto body = [&](auto par_tag, int chunk) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char *cur = mybuf + PAR * G::ABUF;
        unsigned char *nxt = mybuf + (PAR ^ 1) * G::ABUF;
        const int c2 = min(chunk + 2, last);
        const int g0 = (chunk - first) * STEPS;
        float4 sc_t = sc_n, sh_t = sh_n;
        const rsrc_t r2 = rsrc_a(c2);
        const rsrc_t ry2 = SG ? rsrc_y(c2) : r2;
        const bool use2 = CAT && c2 >= a.nchunks1;
        static_for<0, STEPS>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            constexpr int slot = (PAR * STEPS + s) % RB;
            AHalf n_hi = a_hi, n_lo = a_lo;
            if constexpr (s + 1 < STEPS) {
                n_hi = a_read(std::integral_constant<int, s + 1>{}, cur, 0);
                if constexpr (F32) n_lo = a_read(std::integral_constant<int, s + 1>{}, cur, 1);
            }
            if constexpr (F32) {
                // eight v_mfma_f32_32x32x2_f32 per tile contract the wave's 16 channels at this tap (k-step ss: channels 2 ss + kq)
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
                // fp16 operands: one v_mfma_f32_32x32x16_f16 per tile pair and 32 output channels contracts all 16 channels
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi.v[mt], bring[slot][nt][0], acc[mt][nt], 0, 0, 0);
            }
            // staging slots of chunk+1 spread over the steps before the last one: finish a slot into the other stage, re-issue it as chunk+2
            if constexpr (s < STEPS - 1) {
                constexpr int SD = STEPS - 1;
                static_for<(s * NS) / SD, ((s + 1) * NS) / SD>([&](auto i_tag) {
                    a_store(i_tag, nxt, sc_t, sh_t);       // chunk+1: finished into the other stage ...
                    slot_load(i_tag, r2, ry2, use2);       // ... and the slot's registers re-issued as chunk+2
                });
                if constexpr (s == 0) aff_load(c2, sc_n, sh_n);
            }
            b_issue(std::integral_constant<int, slot>{}, g0 + s + RB);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s + 1 == STEPS) {
                __builtin_amdgcn_wave_barrier();  // (the stage was written by other lanes of this wave: LDS is in order per wave)
                n_hi = a_read(std::integral_constant<int, 0>{}, nxt, 0);
                if constexpr (F32) n_lo = a_read(std::integral_constant<int, 0>{}, nxt, 1);
            }
            a_hi = n_hi;
            a_lo = n_lo;
        });
    };
    for (int chunk = first; chunk <= last; chunk += 2) {
        body(std::integral_constant<int, 0>{}, chunk);
        if (chunk + 1 <= last) body(std::integral_constant<int, 1>{}, chunk + 1);
    }

    // ---- reduction of the four waves' K shares through LDS ----
    __syncthreads();
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

    // ---- epilogue: one float4 = 4 consecutive output channels of one pixel per lane and step ----
    constexpr int EU = G::BM * 16 / 256;
#pragma unroll
    for (int k = 0; k < EU; ++k) {
        const int o = tid + 256 * k;
        const int n4 = o & 15, m = o >> 4;
        const int tl = m >> 4, pix = m & 15;
        const int t = mtile * G::TPW + tl;
        const int co = ntile * 64 + 4 * n4;
        const float *r0 = red + m * RP + 4 * n4;
        float4 s = *reinterpret_cast<const float4 *>(r0);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(r0 + w * G::BM * RP);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (t >= a.T || co >= a.Cout) continue;
        if (a.bias) {
            const float4 bb = *reinterpret_cast<const float4 *>(a.bias + co);
            s.x += bb.x; s.y += bb.y; s.z += bb.z; s.w += bb.w;
        }
        size_t addr;
        if constexpr (FULL) {
            const int b = t / a.N, n = t - b * a.N;
            const int2 og = *reinterpret_cast<const int2 *>(a.idx + 2 * n);
            const int h = a.offH + og.x + (pix >> 2), w = a.offW + og.y + (pix & 3);
            if (h < 0 || h >= a.Ho || w < 0 || w >= a.Wo) continue;
            addr = (((size_t)b * a.Ho + h) * a.Wo + w) * a.Cout + co;
            if (a.residual) {  // out = conv + residual; with a block residual: + (x1 - residual) where a shortcut tile covers the pixel
                const float4 rr = ld_f32x4_or_h4(a.residual, addr, a.res_f16 != 0);
                s.x += rr.x; s.y += rr.y; s.z += rr.z; s.w += rr.w;
                if (a.x1) {
                    const int t1 = a.table1[(h / a.R1) * a.gW1 + w / a.S1];
                    if (t1 >= 0) {
                        const float4 xv = *reinterpret_cast<const float4 *>(a.x1 + ((((size_t)b * a.N1 + t1) * a.R1 + h % a.R1) * a.S1 + w % a.S1) * a.Cout + co);
                        s.x += xv.x - rr.x; s.y += xv.y - rr.y; s.z += xv.z - rr.z; s.w += xv.w - rr.w;
                    }
                }
            }
            auto twin = [&](float *dst2, const float *ts, const float *tt) {
                const float4 sc = *reinterpret_cast<const float4 *>(ts + co), sh = *reinterpret_cast<const float4 *>(tt + co);
                float4 tv;
                tv.x = sc.x * s.x; tv.y = sc.y * s.y; tv.z = sc.z * s.z; tv.w = sc.w * s.w;
                tv.x = sh.x + tv.x; tv.y = sh.y + tv.y; tv.z = sh.z + tv.z; tv.w = sh.w + tv.w;
                tv.x = swish(tv.x); tv.y = swish(tv.y); tv.z = swish(tv.z); tv.w = swish(tv.w);
                store_out4(dst2 + addr, tv);
            };
            if (a.twin0) twin(a.twin0, a.tscale0, a.tshift0);
            if (a.twin1) twin(a.twin1, a.tscale1, a.tshift1);
        } else {
            addr = ((size_t)t * 16 + pix) * a.Cout + co;
        }
        if (a.oscale) {
            const float4 os = *reinterpret_cast<const float4 *>(a.oscale + co), oh = *reinterpret_cast<const float4 *>(a.oshift + co);
            s.x = os.x * s.x; s.y = os.y * s.y; s.z = os.z * s.z; s.w = os.w * s.w;
            s.x = oh.x + s.x; s.y = oh.y + s.y; s.z = oh.z + s.z; s.w = oh.w + s.w;
            if (a.oact == SIGE_HIP_ACT_SWISH) { s.x = swish(s.x); s.y = swish(s.y); s.z = swish(s.z); s.w = swish(s.w); }
        }
        store_out4(a.out + addr, s);
    }
}

template <typename G, int SRC, bool AFF, bool CAT, bool FULL, bool Y16 = false>
__global__ __launch_bounds__(256, G::OCC) void conv_tile3_kernel(const Tile3Args a) {
    kernarg_touch<sizeof(Tile3Args)>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
    conv_tile3_body<G, SRC, AFF, CAT, FULL, Y16>(a, blockIdx.x, smem);
}

// the host side of a v3 launch (conv_tile3.hip): argument checks, Tile3Args, launch.  SIGE_HIP_EUNSUPPORTED: the caller falls back
int tile_conv3_launch(int source, const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
                      const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
                      const float *scale, const float *shift, int affineB, int activation,
                      const float *packed, const float *bias, int Cout,
                      int to_full, int offsetH, int offsetW, int Ho, int Wo, const float *residual,
                      const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
                      const float *out_scale, const float *out_shift, int out_activation,
                      float *twin0, const float *twin_scale0, const float *twin_shift0,
                      float *twin1, const float *twin_scale1, const float *twin_shift1,
                      float *out, void *stream, int prec = WIDE_F32, int y_f16 = 0, int residual_f16 = 0);

// launchers (instantiated in conv_tile3_*.hip)
template <int TPW, int PREC = WIDE_F32>
void launch_conv_tile3_gather(const Tile3Args &a, bool aff, bool cat, bool full, hipStream_t st);
template <int TPW, int PREC = WIDE_F32>
void launch_conv_tile3_sg(const Tile3Args &a, bool full, bool y16, hipStream_t st);

}  // namespace sige
