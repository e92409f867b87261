// Channels-last (NHWC) forms of the data-movement ops of the SIGE path for gfx950.
//
// The reference keeps every activation NCHW (sige/cpu/gather.cpp:4-58,
// scatter.cpp:4-68, scatter_gather.cpp:5-56).  A SIGE tile is a 6x6 / 4x4 spatial
// window over ALL channels; in NCHW that is C x 6 separate 24-byte segments, in
// NHWC it is 36 runs of C contiguous floats -- every lane below moves 16 bytes
// (4 channels of one pixel) and a wave covers whole 256-byte+ runs.
//   full tensors [B,H,W,C] (torch.channels_last), tiles [T,R,S,C], C % 4 == 0.
// Arithmetic and its order are those of the NCHW kernels (gather.hip, scatter.hip):
// bit-identical results.
#include "common.hpp"

namespace sige {

constexpr int kT = 256;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// a kernel's OUTPUT in global memory: write-through (common.hpp store_out4; round 6) -- never for LDS pointers (st4 above)
__device__ __forceinline__ void st4_out(float *p, float4 v) { store_out4(p, v); }
// fp16 storage (the resident cache of the "_f16" entry points: SURVEY.md 8b / 8f row 4): 4 consecutive halves = one 8-byte
// load, widened exactly; stores round to nearest even
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const _Float16 *p) {
    const h16x4 h = *reinterpret_cast<const h16x4 *>(p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void st4(_Float16 *p, float4 v) {
    const h16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<h16x4 *>(p) = h;
}

// per-(batch, channel) affine then activation on 4 consecutive channels; scale, then shift:
// two separately rounded ops (gather.cpp:33-53; -ffp-contract=off); SiLU as in common.hpp (~3e-7 relative)
template <int ACT>
__device__ __forceinline__ float4 affine_act4(float4 z, const float *scale, const float *shift, int so, int c) {
    if (scale) { const float4 s = ld4(scale + so + c); z.x = s.x * z.x; z.y = s.y * z.y; z.z = s.z * z.z; z.w = s.w * z.w; }
    if (shift) { const float4 s = ld4(shift + so + c); z.x = s.x + z.x; z.y = s.y + z.y; z.z = s.z + z.z; z.w = s.w + z.w; }
    z.x = activate<ACT>(z.x); z.y = activate<ACT>(z.y); z.z = activate<ACT>(z.z); z.w = activate<ACT>(z.w);
    return z;
}

// Stacked edits (sige_hip_set_edit_batch: E images stacked along H, hp_shift = log2 of one image's height): a tile belongs to the
// image its window's THIRD row lies in (the rule of reduce_mask.hip and of the conv kernels' staging) and rows of its window
// beyond that image are zero padding, not the neighbour's pixels.  First row of the tile's image (0 when the mode is off).
__device__ __forceinline__ int seam_lo(int h0, int hp_shift) { return hp_shift ? (((h0 + 2) >> hp_shift) << hp_shift) : 0; }

// ------------------------------------------------------------------ gather ----
// x [B,H,W,C] -> out [B*N, bH, bW, C]; scale/shift [1|B, C]
template <int ACT, typename XT = float>
__global__ __launch_bounds__(kT) void gather_nhwc_kernel(const XT *__restrict__ x, int B, int C, int H, int W, int bH, int bW,
                                                        const int32_t *__restrict__ idx, int N,
                                                        const float *scale, const float *shift, int aff_sb,
                                                        float *__restrict__ out, long units, int hp_shift) {
    kernarg_touch<128>();
    const int C4 = C / 4, RS = bH * bW;
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const decltype(u) tp = u / C4;
        const int p = (int)(tp % RS);
        const int t = (int)(tp / RS);
        const int b = t / N, n = t - b * N;
        const int h0 = idx[2 * n];
        const int h = h0 + p / bW, w = idx[2 * n + 1] + p % bW;
        const int hlo = seam_lo(h0, hp_shift), hhi = hp_shift ? hlo + (1 << hp_shift) : H;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= hlo && h < hhi && w >= 0 && w < W)
            z = affine_act4<ACT>(ld4(x + (((size_t)b * H + h) * W + w) * C + c), scale, shift, b * aff_sb, c);
        st4_out(out + (size_t)u * 4, z);
    });
}

// ---------------------------------------------------------- scatter_gather ----
// x [B*N, Rx, Sx, C] conv-1 tiles, y [B,H,W,C] cached, map [H,W,3] -> out [B*N, bH, bW, C]
template <int ACT, typename CT = float>
__global__ __launch_bounds__(kT) void scatter_gather_nhwc_kernel(const float *__restrict__ x, const CT *__restrict__ y,
                                                                int B, int C, int H, int W, int Rx, int Sx, int bH, int bW,
                                                                const int32_t *__restrict__ idx, int N,
                                                                const int32_t *__restrict__ map,
                                                                const float *scale, const float *shift, int aff_sb,
                                                                float *__restrict__ out, long units, int hp_shift) {
    kernarg_touch<192>();
    const int C4 = C / 4, RS = bH * bW;
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const decltype(u) tp = u / C4;
        const int p = (int)(tp % RS);
        const int t = (int)(tp / RS);
        const int b = t / N, n = t - b * N;
        const int h0 = idx[2 * n];
        const int h = h0 + p / bW, w = idx[2 * n + 1] + p % bW;
        const int hlo = seam_lo(h0, hp_shift), hhi = hp_shift ? hlo + (1 << hp_shift) : H;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= hlo && h < hhi && w >= 0 && w < W) {
            const int32_t *m = map + 3 * ((size_t)h * W + w);
            const int blk = m[0];
            const float4 v = blk >= 0 ? ld4(x + ((((size_t)b * N + blk) * Rx + m[1]) * Sx + m[2]) * C + c)
                                      : ld4(y + (((size_t)b * H + h) * W + w) * C + c);
            z = affine_act4<ACT>(v, scale, shift, b * aff_sb, c);
        }
        st4_out(out + (size_t)u * 4, z);
    });
}

// ---------------------------------------------------------- SPADE modulation ----
// GauGAN's SPADE layer on tiles (gaugan/models/sige_normalization.py:62-88): out = normalized * (1 + gamma) + beta, where
//   normalized = the conv's input tiles: Gather(x, scale, shift) or ScatterGather(conv tiles, cache, scale, shift) -- the
//                param-free norm folded into the cached affine (sige_fused_spade_generator.py:151-167);
//   gamma|beta = ScatterGather(mlp_gamma_beta's output tiles, its cache) re-tiled to the conv's input tiles (for the
//                shortcut branch Scatter then Gather, which is the same lookup through the scatter map),
// followed by the block's leaky ReLU (sige_fused_spade_generator.py:139,161,166).  The reference runs this as two
// gather-type kernels, a split and four elementwise kernels over [N,2C,b,b]; here it is ONE pass that writes the conv's
// input tile slab.  Same fp32 operations in the same order as that chain (scale, shift; 1 + gamma; product; + beta; leaky).
struct SpadeArgs {
    const float *x_full, *x_tiles;      // normalized source: full tensor (gather) or cache + conv tiles (scatter_gather)
    const int32_t *map_x;
    int Nx, Rx, Sx;
    const float *scale, *shift;
    int aff_sb;
    const float *gb_tiles, *gb_full;    // gamma | beta on the channel axis (2C channels)
    const int32_t *map_g;
    int Ng, Rg, Sg;
    const int32_t *idx;
    int N, bH, bW, B, C, H, W;
    float slope;
    int leaky;
    float *out;
    int hp_shift;  // stacked edits: log2 of one image's height (0 = off)
};

__global__ __launch_bounds__(kT) void spade_modulate_nhwc_kernel(SpadeArgs a, long units) {
    kernarg_touch<sizeof(SpadeArgs) + 8>();
    const int C4 = a.C / 4, RS = a.bH * a.bW, C2 = 2 * a.C;
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const decltype(u) tp = u / C4;
        const int p = (int)(tp % RS);
        const int t = (int)(tp / RS);
        const int b = t / a.N, n = t - b * a.N;
        const int h0 = a.idx[2 * n];
        const int h = h0 + p / a.bW, w = a.idx[2 * n + 1] + p % a.bW;
        const int hlo = seam_lo(h0, a.hp_shift), hhi = a.hp_shift ? hlo + (1 << a.hp_shift) : a.H;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= hlo && h < hhi && w >= 0 && w < a.W) {
            const size_t pix = ((size_t)b * a.H + h) * a.W + w;
            // everything this unit reads is addressed before the first value is used: the three loads go out together
            const float *xs = a.x_full + pix * a.C + c;
            if (a.x_tiles) {
                const int32_t *m = a.map_x + 3 * ((size_t)h * a.W + w);
                const int blk = m[0];
                if (blk >= 0) xs = a.x_tiles + ((((size_t)b * a.Nx + blk) * a.Rx + m[1]) * a.Sx + m[2]) * a.C + c;
            }
            const int32_t *mg = a.map_g + 3 * ((size_t)h * a.W + w);
            const int gblk = mg[0];
            const float *gs = gblk >= 0 ? a.gb_tiles + ((((size_t)b * a.Ng + gblk) * a.Rg + mg[1]) * a.Sg + mg[2]) * C2 + c
                                        : a.gb_full + pix * C2 + c;
            const float4 xv = ld4(xs), gamma = ld4(gs), beta = ld4(gs + a.C);
            const float4 nv = affine_act4<SIGE_HIP_ACT_IDENTITY>(xv, a.scale, a.shift, b * a.aff_sb, c);
            float4 g1 = make_float4(1.0f + gamma.x, 1.0f + gamma.y, 1.0f + gamma.z, 1.0f + gamma.w);
            z = make_float4(nv.x * g1.x, nv.y * g1.y, nv.z * g1.z, nv.w * g1.w);
            z = make_float4(z.x + beta.x, z.y + beta.y, z.z + beta.z, z.w + beta.w);
            if (a.leaky) {
                z.x = z.x > 0.f ? z.x : z.x * a.slope; z.y = z.y > 0.f ? z.y : z.y * a.slope;
                z.z = z.z > 0.f ? z.z : z.z * a.slope; z.w = z.w > 0.f ? z.w : z.w * a.slope;
            }
        }
        st4_out(a.out + (size_t)u * 4, z);
    });
}

// ------------------------------------------------------------------ scatter ----
template <typename CT>
struct ScatterNhwcArgsT {
    const float *x0, *x1, *res;  // main tiles, shortcut tiles, residual
    const CT *y0, *y1;           // cached tensor, cached shortcut tensor (fp32, or fp16 storage: the "_f16" entry points)
    float *out;
    const int32_t *table0, *table1, *idx0, *idx1;
    int B, C, H, W;
    int R0, S0, N0, gW0, offH, offW, strH, strW;
    int R1, S1, N1, gW1;
};
using ScatterNhwcArgs = ScatterNhwcArgsT<float>;

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// value of output pixel (b,h,w), channels c..c+3, given the tiles covering it (t0 / t1, -1 = none):
//   out = y0;  main tile: out = x0 + residual (BLOCK_RES: residual = y1);  shortcut tile: out += x1 - y1
// (scatter.cpp:4-39 then 41-68, same operation order)
template <bool BLOCK_RES, typename CT>
__device__ __forceinline__ float4 scatter_value(const ScatterNhwcArgsT<CT> &a, int b, int h, int w, int c, int t0, int t1, size_t q) {
    float4 v;
    float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BLOCK_RES && (t0 >= 0 || t1 >= 0)) r1 = ld4(a.y1 + q);
    if (t0 >= 0) {
        v = ld4(a.x0 + ((((size_t)b * a.N0 + t0) * a.R0 + h % a.R0) * a.S0 + w % a.S0) * a.C + c);
        if (BLOCK_RES) v = add4(r1, v);
        else if (a.res) v = add4(ld4(a.res + q), v);
    } else {
        v = ld4(a.y0 + q);
    }
    if (BLOCK_RES && t1 >= 0)
        v = add4(v, sub4(ld4(a.x1 + ((((size_t)b * a.N1 + t1) * a.R1 + h % a.R1) * a.S1 + w % a.S1) * a.C + c), r1));
    return v;
}

// reference semantics: a fresh full tensor, ONE streaming pass (no clone + overwrite)
template <bool BLOCK_RES, typename CT = float>
__global__ __launch_bounds__(kT) void scatter_full_nhwc_kernel(ScatterNhwcArgsT<CT> a, long units) {
    kernarg_touch<sizeof(ScatterNhwcArgsT<CT>) + 8>();
    const int C4 = a.C / 4;
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const decltype(u) pix = u / C4;
        const int w = (int)(pix % a.W);
        const decltype(u) bh = pix / a.W;
        const int h = (int)(bh % a.H), b = (int)(bh / a.H);
        const int t0 = a.table0[(h / a.R0) * a.gW0 + w / a.S0];
        const int t1 = BLOCK_RES ? a.table1[(h / a.R1) * a.gW1 + w / a.S1] : -1;
        st4_out(a.out + (size_t)u * 4, scatter_value<BLOCK_RES, CT>(a, b, h, w, c, t0, t1, (size_t)u * 4));
    });
}

// in-place form: `out` already holds y0 outside the covered pixels (a persistent buffer of the
// Scatter module); only the pixels under a main tile -- and, BLOCK_RES, under a shortcut tile
// that no main tile covers -- are written.  Traffic ~ active tiles instead of the full tensor.
template <bool BLOCK_RES, typename CT = float>
__global__ __launch_bounds__(kT) void scatter_tiles_nhwc_kernel(ScatterNhwcArgsT<CT> a, long units0, long units) {
    kernarg_touch<sizeof(ScatterNhwcArgsT<CT>) + 16>();
    const int C4 = a.C / 4;
    for_units<kT>(units, [&](auto u) {
        const bool main = u < units0;
        const decltype(u) v = main ? u : u - units0;
        const int R = main ? a.R0 : a.R1, S = main ? a.S0 : a.S1, N = main ? a.N0 : a.N1;
        const int c = (int)(v % C4) * 4;
        const decltype(u) tp = v / C4;
        const int p = (int)(tp % (R * S));
        const int t = (int)(tp / (R * S));
        const int b = t / N, n = t - b * N;
        int h, w;
        if (main) { h = (a.offH + a.idx0[2 * n]) / a.strH + p / S; w = (a.offW + a.idx0[2 * n + 1]) / a.strW + p % S; }
        else { h = a.idx1[2 * n] + p / S; w = a.idx1[2 * n + 1] + p % S; }
        if (h < 0 || h >= a.H || w < 0 || w >= a.W) return;
        int t0, t1;
        if (main) { t0 = n; t1 = BLOCK_RES ? a.table1[(h / a.R1) * a.gW1 + w / a.S1] : -1; }
        else { t1 = n; t0 = a.table0[(h / a.R0) * a.gW0 + w / a.S0]; if (t0 >= 0) return; }  // a main tile writes this pixel
        const size_t q = ((((size_t)b * a.H + h) * a.W + w) * a.C) + c;
        st4_out(a.out + q, scatter_value<BLOCK_RES, CT>(a, b, h, w, c, t0, t1, q));
    });
}

static int grid_for(long units) {
    long g = (units + kT - 1) / kT;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// ---------------------------------------------------------------- attention ----
typedef float f32x4 __attribute__((ext_vector_type(4)));

// qkv [B,HW,3C] (q | k | v on the channel axis).  S[i][j] = scale * sum_c q[i][c] k[j][c]
__global__ __launch_bounds__(kT) void attn_scores_nhwc_kernel(const float *__restrict__ qkv, int C, int HW, float scale,
                                                             float *__restrict__ S) {
    __shared__ float red[4][16][20];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, n = lane & 15;
    const int C3 = 3 * C, cw = C / 4;  // wave w: channels [w*cw, (w+1)*cw), lane group kq: 4 consecutive channels per step of 16
    const float *qa = qkv + ((size_t)b * HW + i0 + n) * C3 + wave * cw + kq * 4;       // A[m = query][k]
    const float *kb = qkv + ((size_t)b * HW + j0 + n) * C3 + C + wave * cw + kq * 4;   // B[k][n = key]
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // each 16-byte load feeds 4 MFMA k-steps; the k order (kq*4 + e within a block of 16 channels)
    // is the same for A and B, which is all the contraction needs
    // (8 steps = 16 loads in flight per lane: a loop that loads as it goes pays the memory latency every step)
    for (int s0 = 0; s0 < cw; s0 += 128) {
        float4 a4[8], b4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + 16 * u;
            const bool ok = s < cw;  // (steps past the end contribute exact zeros)
            a4[u] = ok ? ld4(qa + s) : make_float4(0.f, 0.f, 0.f, 0.f);
            b4[u] = ok ? ld4(kb + s) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].x, b4[u].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].y, b4[u].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].z, b4[u].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u].w, b4[u].w, acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n] = acc0[r] + acc1[r];
    __syncthreads();
    const int qi = tid >> 4, kj = tid & 15;
    const float v = (red[0][qi][kj] + red[1][qi][kj]) + (red[2][qi][kj] + red[3][qi][kj]);
    S[((size_t)b * HW + i0 + qi) * HW + j0 + kj] = v * scale;
}

// out[i][c] = sum_j softmax_j(S[i][.])[j] v[j][c]:  A[m = query][k = key] = P (LDS), B[k = key][n = channel] = v (global,
// 64-byte coalesced per lane group).  Workgroup = 16 queries x 64 channels (one 16x16 tile per wave).
__global__ __launch_bounds__(kT) void attn_apply_nhwc_kernel(const float *__restrict__ qkv, const float *__restrict__ S,
                                                            int C, int HW, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float P[];  // [16][HW + 4]
    const int PS = HW + 4;
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 16, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, n = lane & 15;
    const int c = c0 + wave * 16 + n;
    const bool cok = c < C;
    const size_t rs = (size_t)3 * C;
    // 256 keys per block: all 64 value loads of a block are issued before the first MFMA needs one (a loop that
    // loads one step ahead pays the memory latency every step: 16 x ~0.6 us for 256 tokens); the first block's
    // are issued before the softmax, which does not depend on them
    // (one buffer descriptor over this batch's qkv rows from the value columns on, ONE 32-bit offset register per lane, the key step
    //  as a scalar offset: the 64 loads of a block are 64 instructions, not 64 x ten of 64-bit address arithmetic -- 1 us of issue
    //  in front of everything else, round 6)
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(qkv + (size_t)b * HW * 3 * C + 2 * C), 0,
        (unsigned)min((size_t)0x7fffffff, ((size_t)HW * 3 * C - 2 * C) * sizeof(float)), 0x00020000);
    const int v_lane = ((cok ? c : 0) + kq * (int)rs) * (int)sizeof(float);
    const int v_step = 4 * (int)rs * (int)sizeof(float);  // bytes between key steps
    float bv[64];
    auto load_values = [&](int j0) {
        const int nk = min(256, HW - j0) / 4;  // k-steps in this block (HW % 16 == 0)
#pragma unroll
        for (int u = 0; u < 64; ++u)
            bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_v, v_lane, (j0 / 4 + min(u, nk - 1)) * v_step, 0));
    };
    {
        const int row = tid >> 4, l16 = tid & 15;
        const float *srow = S + ((size_t)b * HW + i0 + row) * HW;
        if (HW <= 256) {
            // the whole score row of this lane group in registers: 4 loads in flight, one pass -- requested AHEAD of the values (the
            // softmax is the dependent chain; the values are not needed before the first MFMA)
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = l16 * 4 + 64 * u;
                const float4 q4 = ld4(srow + min(j, HW - 4));  // (branch-free: a group past the row reads its last four, masked below)
                t[u] = j < HW ? q4 : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
            load_values(0);
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(t[u].x, t[u].y)), fmaxf(t[u].z, t[u].w));
            m = row16_max(m);  // (lane groups of 16 = DPP rows: no LDS crossbar round trips)
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (l16 * 4 + 64 * u < HW) {
                    t[u].x = expf(t[u].x - m); t[u].y = expf(t[u].y - m); t[u].z = expf(t[u].z - m); t[u].w = expf(t[u].w - m);
                    sum += (t[u].x + t[u].y) + (t[u].z + t[u].w);
                }
            }
            sum = row16_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = l16 * 4 + 64 * u;
                if (j < HW) {
                    t[u].x *= inv; t[u].y *= inv; t[u].z *= inv; t[u].w *= inv;
                    st4(P + row * PS + j, t[u]);
                }
            }
        } else {
            load_values(0);
            float m = -INFINITY;
            for (int j = l16 * 4; j < HW; j += 64) {
                const float4 t = ld4(srow + j);
                st4(P + row * PS + j, t);
                m = fmaxf(fmaxf(m, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
            float sum = 0.f;
            for (int j = l16 * 4; j < HW; j += 64) {
                float4 t = ld4(P + row * PS + j);
                t.x = expf(t.x - m); t.y = expf(t.y - m); t.z = expf(t.z - m); t.w = expf(t.w - m);
                sum += (t.x + t.y) + (t.z + t.w);
                st4(P + row * PS + j, t);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
            const float inv = 1.0f / sum;
            for (int j = l16 * 4; j < HW; j += 64) {
                float4 t = ld4(P + row * PS + j);
                t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
                st4(P + row * PS + j, t);
            }
        }
    }
    __syncthreads();
    const float *pa = P + n * PS + kq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < HW; j0 += 256) {
        const int nk = min(256, HW - j0) / 4;
        if (j0 > 0) load_values(j0);
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const float av = pa[j0 + 4 * min(u, nk - 1)];
            const float bb = u < nk ? bv[u] : 0.f;  // (steps past the block contribute exact zeros)
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bb, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bb, acc0, 0, 0, 0);
        }
    }
    // D[row = query 4*kq + r][col = channel n]
    if (cok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((size_t)b * HW + i0 + 4 * kq + r) * C + c] = acc0[r] + acc1[r];
    }
}

}  // namespace sige

using namespace sige;

static bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static bool affine_shape(const float *scale, int sB, int sC, const float *shift, int tB, int tC, int B, int C, int *aff_sb) {
    *aff_sb = 0;
    for (int k = 0; k < 2; ++k) {
        const float *p = k ? shift : scale;
        const int pb = k ? tB : sB, pc = k ? tC : sC;
        if (!p) continue;
        if (!((pb == 1 || pb == B) && pc == C) || !al16(p)) return false;
        if (pb > 1) *aff_sb = C;
    }
    if (scale && shift && sB != tB) return false;
    return true;
}

template <typename XT>
static int gather_nhwc_impl(const XT *x, int B, int C, int H, int W, int bH, int bW,
                            const int32_t *active_indices, int N,
                            const float *scale, int scaleB, int scaleC,
                            const float *shift, int shiftB, int shiftC,
                            int activation, float *out, void *stream) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || bH <= 0 || bW <= 0 || N < 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    const int hp_shift = stacked_shift(H);  // (stacked edits: halo rows beyond a tile's own image are zero padding)
    if (hp_shift < 0 || (hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !out || !active_indices) return SIGE_HIP_EINVAL;
    int aff_sb;
    if (C % 4 || !al16(x) || !al16(out) || !affine_shape(scale, scaleB, scaleC, shift, shiftB, shiftC, B, C, &aff_sb))
        return SIGE_HIP_EUNSUPPORTED;
    const long units = (long)B * N * bH * bW * (C / 4);
    hipStream_t st = as_stream(stream);
    if (activation == SIGE_HIP_ACT_SWISH)
        gather_nhwc_kernel<SIGE_HIP_ACT_SWISH, XT><<<grid_for(units), kT, 0, st>>>(x, B, C, H, W, bH, bW, active_indices, N, scale, shift, aff_sb, out, units, hp_shift);
    else
        gather_nhwc_kernel<SIGE_HIP_ACT_IDENTITY, XT><<<grid_for(units), kT, 0, st>>>(x, B, C, H, W, bH, bW, active_indices, N, scale, shift, aff_sb, out, units, hp_shift);
    return launch_status();
}

extern "C" int sige_hip_gather_nhwc_f32(const float *x, int B, int C, int H, int W, int bH, int bW,
                                        const int32_t *active_indices, int N,
                                        const float *scale, int scaleB, int scaleC,
                                        const float *shift, int shiftB, int shiftC,
                                        int activation, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_nhwc_f32, (sige::CountOf<7, 8>), x, B, C, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
    return gather_nhwc_impl<float>(x, B, C, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
}

extern "C" int sige_hip_gather_nhwc_f16(const void *x, int B, int C, int H, int W, int bH, int bW,
                                        const int32_t *active_indices, int N,
                                        const float *scale, int scaleB, int scaleC,
                                        const float *shift, int shiftB, int shiftC,
                                        int activation, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_nhwc_f16, (sige::CountOf<7, 8>), x, B, C, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
    return gather_nhwc_impl<_Float16>(static_cast<const _Float16 *>(x), B, C, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
}

template <typename CT>
static int scatter_gather_nhwc_impl(const float *x, const CT *y, int B, int C, int H, int W,
                                    int Rx, int Sx, int bH, int bW,
                                    const int32_t *active_indices, int N, const int32_t *scatter_map,
                                    const float *scale, int scaleB, int scaleC,
                                    const float *shift, int shiftB, int shiftC,
                                    int activation, float *out, void *stream) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || Rx <= 0 || Sx <= 0 || bH <= 0 || bW <= 0 || N < 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    const int hp_shift = stacked_shift(H);
    if (hp_shift < 0 || (hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    int aff_sb;
    if (C % 4 || !al16(x) || !al16(y) || !al16(out) || !affine_shape(scale, scaleB, scaleC, shift, shiftB, shiftC, B, C, &aff_sb))
        return SIGE_HIP_EUNSUPPORTED;
    const long units = (long)B * N * bH * bW * (C / 4);
    hipStream_t st = as_stream(stream);
    if (activation == SIGE_HIP_ACT_SWISH)
        scatter_gather_nhwc_kernel<SIGE_HIP_ACT_SWISH, CT><<<grid_for(units), kT, 0, st>>>(x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, shift, aff_sb, out, units, hp_shift);
    else
        scatter_gather_nhwc_kernel<SIGE_HIP_ACT_IDENTITY, CT><<<grid_for(units), kT, 0, st>>>(x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, shift, aff_sb, out, units, hp_shift);
    return launch_status();
}

extern "C" int sige_hip_scatter_gather_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W,
                                                int Rx, int Sx, int bH, int bW,
                                                const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                const float *scale, int scaleB, int scaleC,
                                                const float *shift, int shiftB, int shiftC,
                                                int activation, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_nhwc_f32, (sige::CountOf<10, 11>), x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
    return scatter_gather_nhwc_impl<float>(x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
}

extern "C" int sige_hip_scatter_gather_nhwc_f16(const float *x, const void *y, int B, int C, int H, int W,
                                                int Rx, int Sx, int bH, int bW,
                                                const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                const float *scale, int scaleB, int scaleC,
                                                const float *shift, int shiftB, int shiftC,
                                                int activation, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_nhwc_f16, (sige::CountOf<10, 11>), x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
    return scatter_gather_nhwc_impl<_Float16>(x, static_cast<const _Float16 *>(y), B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, out, stream);
}

extern "C" int sige_hip_spade_modulate_nhwc_f32(
        const float *x_full, const float *x_tiles, const int32_t *map_x, int Nx, int Rx, int Sx,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC,
        const float *gb_tiles, const float *gb_full, const int32_t *map_g, int Ng, int Rg, int Sg,
        int B, int C, int H, int W, int bH, int bW, const int32_t *active_indices, int N,
        int leaky, float slope, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_spade_modulate_nhwc_f32, (sige::CountOf<24, 25>, sige::CountOf<2, 3>, sige::CountOf<14, 15>), x_full, x_tiles, map_x, Nx, Rx, Sx, scale, scaleB, scaleC, shift, shiftB, shiftC, gb_tiles, gb_full, map_g, Ng, Rg, Sg, B, C, H, W, bH, bW, active_indices, N, leaky, slope, out, stream);
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || bH <= 0 || bW <= 0 || N < 0 || Ng < 0 || Nx < 0) return SIGE_HIP_EINVAL;
    const int hp_shift = stacked_shift(H);
    if (hp_shift < 0 || (hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x_full || !gb_full || !map_g || !active_indices || !out) return SIGE_HIP_EINVAL;
    if ((Ng > 0 && (!gb_tiles || Rg <= 0 || Sg <= 0)) || (x_tiles && (!map_x || Rx <= 0 || Sx <= 0))) return SIGE_HIP_EINVAL;
    int aff_sb = 0;
    if (C % 4 || !al16(x_full) || !al16(x_tiles) || !al16(gb_tiles) || !al16(gb_full) || !al16(out) ||
        !affine_shape(scale, scaleB, scaleC, shift, shiftB, shiftC, B, C, &aff_sb))
        return SIGE_HIP_EUNSUPPORTED;
    SpadeArgs a{};
    a.x_full = x_full; a.x_tiles = x_tiles; a.map_x = map_x; a.Nx = Nx; a.Rx = Rx; a.Sx = Sx;
    a.scale = scale; a.shift = shift; a.aff_sb = aff_sb;
    a.gb_tiles = gb_tiles; a.gb_full = gb_full; a.map_g = map_g; a.Ng = Ng; a.Rg = Rg; a.Sg = Sg;
    a.idx = active_indices; a.N = N; a.bH = bH; a.bW = bW; a.B = B; a.C = C; a.H = H; a.W = W;
    a.slope = slope; a.leaky = leaky; a.out = out; a.hp_shift = hp_shift;
    const long units = (long)B * N * bH * bW * (C / 4);
    spade_modulate_nhwc_kernel<<<grid_for(units), kT, 0, as_stream(stream)>>>(a, units);
    return launch_status();
}

template <typename CT>
static int scatter_nhwc_impl(const float *x, const CT *y, int B, int C, int H, int W, int R, int S,
                             int offsetH, int offsetW, int strideH, int strideW,
                             const int32_t *active_indices, const int32_t *table, int gH, int gW, int N,
                             const float *residual, int in_place, float *out, void *stream) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || R <= 0 || S <= 0 || N < 0 || strideH <= 0 || strideW <= 0) return SIGE_HIP_EINVAL;
    if ((long)B * H * W == 0) return SIGE_HIP_OK;
    if (!y || !out || (N && (!x || !table || !active_indices))) return SIGE_HIP_EINVAL;
    if (C % 4 || !al16(x) || !al16(y) || !al16(out) || !al16(residual)) return SIGE_HIP_EUNSUPPORTED;
    if (gH < (H + R - 1) / R || gW < (W + S - 1) / S) return SIGE_HIP_EINVAL;
    ScatterNhwcArgsT<CT> a{};
    a.x0 = x; a.y0 = y; a.res = residual; a.out = out; a.table0 = table; a.idx0 = active_indices;
    a.B = B; a.C = C; a.H = H; a.W = W; a.R0 = R; a.S0 = S; a.N0 = N; a.gW0 = gW;
    a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    a.R1 = a.S1 = 1;
    hipStream_t st = as_stream(stream);
    if (in_place) {
        const long units = (long)B * N * R * S * (C / 4);
        if (units) scatter_tiles_nhwc_kernel<false, CT><<<grid_for(units), kT, 0, st>>>(a, units, units);
    } else {
        const long units = (long)B * H * W * (C / 4);
        scatter_full_nhwc_kernel<false, CT><<<grid_for(units), kT, 0, st>>>(a, units);
    }
    return launch_status();
}

extern "C" int sige_hip_scatter_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                                         int offsetH, int offsetW, int strideH, int strideW,
                                         const int32_t *active_indices, const int32_t *table, int gH, int gW, int N,
                                         const float *residual, int in_place, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_nhwc_f32, (sige::CountOf<12, 16>), x, y, B, C, H, W, R, S, offsetH, offsetW, strideH, strideW, active_indices, table, gH, gW, N, residual, in_place, out, stream);
    return scatter_nhwc_impl<float>(x, y, B, C, H, W, R, S, offsetH, offsetW, strideH, strideW, active_indices, table, gH, gW, N, residual, in_place, out, stream);
}

extern "C" int sige_hip_scatter_nhwc_f16(const float *x, const void *y, int B, int C, int H, int W, int R, int S,
                                         int offsetH, int offsetW, int strideH, int strideW,
                                         const int32_t *active_indices, const int32_t *table, int gH, int gW, int N,
                                         const float *residual, int in_place, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_nhwc_f16, (sige::CountOf<12, 16>), x, y, B, C, H, W, R, S, offsetH, offsetW, strideH, strideW, active_indices, table, gH, gW, N, residual, in_place, out, stream);
    return scatter_nhwc_impl<_Float16>(x, static_cast<const _Float16 *>(y), B, C, H, W, R, S, offsetH, offsetW, strideH, strideW, active_indices, table, gH, gW, N, residual, in_place, out, stream);
}

template <typename CT>
static int scatter_with_block_residual_nhwc_impl(
        const float *x0, const CT *y0, const float *x1, const CT *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *active_indices1, const int32_t *table1, int gH1, int gW1, int N1,
        int in_place, float *out, void *stream) {
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || R0 <= 0 || S0 <= 0 || R1 <= 0 || S1 <= 0 || N0 < 0 || N1 < 0) return SIGE_HIP_EINVAL;
    if (strideH <= 0 || strideW <= 0) return SIGE_HIP_EINVAL;
    if ((long)B * H * W == 0) return SIGE_HIP_OK;
    if (!y0 || !y1 || !out || !table0 || !table1 || (N0 && (!x0 || !active_indices0)) || (N1 && (!x1 || !active_indices1)))
        return SIGE_HIP_EINVAL;
    if (C % 4 || !al16(x0) || !al16(y0) || !al16(x1) || !al16(y1) || !al16(out)) return SIGE_HIP_EUNSUPPORTED;
    if (gH0 < (H + R0 - 1) / R0 || gW0 < (W + S0 - 1) / S0 || gH1 < (H + R1 - 1) / R1 || gW1 < (W + S1 - 1) / S1) return SIGE_HIP_EINVAL;
    ScatterNhwcArgsT<CT> a{};
    a.x0 = x0; a.y0 = y0; a.x1 = x1; a.y1 = y1; a.out = out;
    a.table0 = table0; a.table1 = table1; a.idx0 = active_indices0; a.idx1 = active_indices1;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.R0 = R0; a.S0 = S0; a.N0 = N0; a.gW0 = gW0; a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    a.R1 = R1; a.S1 = S1; a.N1 = N1; a.gW1 = gW1;
    hipStream_t st = as_stream(stream);
    if (in_place) {
        const long units0 = (long)B * N0 * R0 * S0 * (C / 4), units = units0 + (long)B * N1 * R1 * S1 * (C / 4);
        if (units) scatter_tiles_nhwc_kernel<true, CT><<<grid_for(units), kT, 0, st>>>(a, units0, units);
    } else {
        const long units = (long)B * H * W * (C / 4);
        scatter_full_nhwc_kernel<true, CT><<<grid_for(units), kT, 0, st>>>(a, units);
    }
    return launch_status();
}

extern "C" int sige_hip_scatter_with_block_residual_nhwc_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *active_indices1, const int32_t *table1, int gH1, int gW1, int N1,
        int in_place, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_with_block_residual_nhwc_f32, (sige::CountOf<16, 20>, sige::CountOf<21, 25>), x0, y0, x1, y1, B, C, H, W, R0, S0, R1, S1, offsetH, offsetW, strideH, strideW, active_indices0, table0, gH0, gW0, N0, active_indices1, table1, gH1, gW1, N1, in_place, out, stream);
    return scatter_with_block_residual_nhwc_impl<float>(x0, y0, x1, y1, B, C, H, W, R0, S0, R1, S1, offsetH, offsetW, strideH, strideW,
                                                        active_indices0, table0, gH0, gW0, N0, active_indices1, table1, gH1, gW1, N1, in_place, out, stream);
}

extern "C" int sige_hip_scatter_with_block_residual_nhwc_f16(
        const float *x0, const void *y0, const float *x1, const void *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *active_indices1, const int32_t *table1, int gH1, int gW1, int N1,
        int in_place, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_with_block_residual_nhwc_f16, (sige::CountOf<16, 20>, sige::CountOf<21, 25>), x0, y0, x1, y1, B, C, H, W, R0, S0, R1, S1, offsetH, offsetW, strideH, strideW, active_indices0, table0, gH0, gW0, N0, active_indices1, table1, gH1, gW1, N1, in_place, out, stream);
    return scatter_with_block_residual_nhwc_impl<_Float16>(x0, static_cast<const _Float16 *>(y0), x1, static_cast<const _Float16 *>(y1), B, C, H, W, R0, S0, R1, S1,
                                                           offsetH, offsetW, strideH, strideW, active_indices0, table0, gH0, gW0, N0,
                                                           active_indices1, table1, gH1, gW1, N1, in_place, out, stream);
}

// out = act(scale[b, c] * x + shift[b, c]) over a whole channels-last tensor: the activated copy of a ScatterGather cache
// (sige_amd.nn.ScatterGather.cache_activated) in one streaming pass -- the full pass otherwise spends a multiply, an add, a
// SiLU and a copy kernel on it.  Same two separately rounded ops and the same fp32 SiLU as the standalone gather.
template <typename XT, typename OT>
__global__ __launch_bounds__(kT) void affine_act_nhwc_kernel(const XT *__restrict__ x, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, int aff_sb, int C, size_t hwc4,
                                                            size_t total4, int act, OT *__restrict__ out) {
    for (size_t u = (size_t)blockIdx.x * kT + threadIdx.x; u < total4; u += (size_t)gridDim.x * kT) {
        const size_t b = u / hwc4;
        const int c = (int)((u * 4) % C);
        const float4 v = ld4(x + u * 4), sc = ld4(scale + b * aff_sb + c), sh = ld4(shift + b * aff_sb + c);
        float4 z;
        z.x = sc.x * v.x; z.y = sc.y * v.y; z.z = sc.z * v.z; z.w = sc.w * v.w;
        z.x = sh.x + z.x; z.y = sh.y + z.y; z.z = sh.z + z.z; z.w = sh.w + z.w;
        if (act == SIGE_HIP_ACT_SWISH) { z.x = swish(z.x); z.y = swish(z.y); z.z = swish(z.z); z.w = swish(z.w); }
        st4(out + u * 4, z);
    }
}

template <typename XT, typename OT>
static int affine_act_nhwc_impl(const XT *x, int B, int C, int H, int W, const float *scale, const float *shift,
                                int affineB, int activation, OT *out, void *stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !scale || !shift || !out) return SIGE_HIP_EINVAL;
    if (affineB != 1 && affineB != B) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (C % 4 || !al16(x) || !al16(out) || !al16(scale) || !al16(shift)) return SIGE_HIP_EUNSUPPORTED;
    const size_t hwc4 = (size_t)H * W * C / 4, total4 = hwc4 * B;
    affine_act_nhwc_kernel<XT, OT><<<grid_for((long)total4), kT, 0, as_stream(stream)>>>(x, scale, shift, affineB > 1 ? C : 0, C, hwc4, total4,
                                                                                        activation, out);
    return launch_status();
}

extern "C" int sige_hip_affine_act_nhwc_f32(const float *x, int B, int C, int H, int W, const float *scale, const float *shift,
                                            int affineB, int activation, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_affine_act_nhwc_f32, x, B, C, H, W, scale, shift, affineB, activation, out, stream);
    return affine_act_nhwc_impl<float, float>(x, B, C, H, W, scale, shift, affineB, activation, out, stream);
}

// ... of an fp16-stored cache: `x` fp16; `out` fp16 (out_f16 != 0: the activated copy kept next to an fp16 cache) or fp32 (a
// persistent activated twin)
extern "C" int sige_hip_affine_act_nhwc_f16(const void *x, int B, int C, int H, int W, const float *scale, const float *shift,
                                            int affineB, int activation, void *out, int out_f16, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_affine_act_nhwc_f16, x, B, C, H, W, scale, shift, affineB, activation, out, out_f16, stream);
    if (out_f16)
        return affine_act_nhwc_impl<_Float16, _Float16>(static_cast<const _Float16 *>(x), B, C, H, W, scale, shift, affineB, activation,
                                                        static_cast<_Float16 *>(out), stream);
    return affine_act_nhwc_impl<_Float16, float>(static_cast<const _Float16 *>(x), B, C, H, W, scale, shift, affineB, activation,
                                                 static_cast<float *>(out), stream);
}

// dst (fp32) <- src (fp16), n elements (n % 4 == 0, 8 / 16-byte aligned): the refresh of a persistent Scatter output from an
// fp16-stored cache; dst (fp16) <- src (fp32): storing a full-pass output into the fp16 cache
template <typename ST, typename DT>
__global__ __launch_bounds__(kT) void convert4_kernel(const ST *__restrict__ src, DT *__restrict__ dst, size_t n4) {
    for (size_t u = (size_t)blockIdx.x * kT + threadIdx.x; u < n4; u += (size_t)gridDim.x * kT) st4(dst + u * 4, ld4(src + u * 4));
}

extern "C" int sige_hip_convert_f16_f32(const void *src, float *dst, size_t n, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_convert_f16_f32, src, dst, n, stream);
    if (!src || !dst) return SIGE_HIP_EINVAL;
    if (n == 0) return SIGE_HIP_OK;
    if (n % 4 || (reinterpret_cast<uintptr_t>(src) & 7) || !al16(dst)) return SIGE_HIP_EUNSUPPORTED;
    convert4_kernel<_Float16, float><<<grid_for((long)(n / 4)), kT, 0, as_stream(stream)>>>(static_cast<const _Float16 *>(src), dst, n / 4);
    return launch_status();
}

extern "C" int sige_hip_convert_f32_f16(const float *src, void *dst, size_t n, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_convert_f32_f16, src, dst, n, stream);
    if (!src || !dst) return SIGE_HIP_EINVAL;
    if (n == 0) return SIGE_HIP_OK;
    if (n % 4 || (reinterpret_cast<uintptr_t>(dst) & 7) || !al16(src)) return SIGE_HIP_EUNSUPPORTED;
    convert4_kernel<float, _Float16><<<grid_for((long)(n / 4)), kT, 0, as_stream(stream)>>>(src, static_cast<_Float16 *>(dst), n / 4);
    return launch_status();
}

extern "C" int sige_hip_attention_nhwc_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                                           float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_attention_nhwc_f32, qkv, B, C, HW, scale, workspace, out, stream);
    if (B <= 0 || C <= 0 || HW <= 0) return SIGE_HIP_EINVAL;
    if (!qkv || !workspace || !out) return SIGE_HIP_EINVAL;
    // 16x16 tiles; 4 waves x 16-channel steps of 16-byte loads; P rows in LDS
    if (HW % 16 || C % 64 || B > 65535 || !al16(qkv) || !al16(workspace)) return SIGE_HIP_EUNSUPPORTED;
    const size_t lds = (size_t)16 * (HW + 4) * sizeof(float);
    if (lds > 64 * 1024 || (size_t)HW * 3 * C * sizeof(float) >= 0x7fffffffu) return SIGE_HIP_EUNSUPPORTED;  // (32-bit value offsets)
    hipStream_t st = as_stream(stream);
    attn_scores_nhwc_kernel<<<dim3(HW / 16, HW / 16, B), kT, 0, st>>>(qkv, C, HW, scale, workspace);
    attn_apply_nhwc_kernel<<<dim3(ceil_div(C, 64), HW / 16, B), kT, lds, st>>>(qkv, workspace, C, HW, out);
    return launch_status(2);
}
