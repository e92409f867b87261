// Channels-last helpers that take the LAST torch kernels out of the GauGAN SPADE generator's sparse forward, so that the whole
// forward reaches the GPU through this library and a launch plan (plan.hpp) can record it:
//
//   resize_nearest     F.interpolate(mode="nearest") by an integer factor, up or down: the label map at every block's
//                      resolution (sige_fused_spade_generator.py:143-146) and the x2 up-sampling between blocks (:243-257)
//   act_split          act(x) written as `parts` dense channel groups: ReLU + torch.split of a block's label features
//                      (gaugan/models/sige_normalization.py: mlp_shared -> one slice per SPADE layer)
//   scatter_gather_split  ScatterGather of the label branch (sige/cpu/scatter_gather.cpp:5-56 arithmetic) followed by the
//                      same ReLU + split, in one pass.  relu(relu(v)) == relu(v), so applying it to the cached values
//                      (stored AFTER the ReLU by the full pass) as well is exact
//   spade_modulate_dense  out = leaky((scale * x + shift) * (1 + gamma) + beta) on a full tensor (the blocks below
//                      `num_sparse_layers`, which the reference recomputes densely in sparse mode: :133-173)
//
// Same fp32 operations in the same order as the torch chains they replace; every lane moves 16 bytes (4 channels of a pixel).
#include "common.hpp"

namespace sige {

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__device__ __forceinline__ float act1(float z, int act, float slope) {
    if (act == SIGE_HIP_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == SIGE_HIP_ACT_LEAKY) return z > 0.f ? z : z * slope;
    return z;
}
__device__ __forceinline__ float4 act4(float4 z, int act, float slope) {
    return make_float4(act1(z.x, act, slope), act1(z.y, act, slope), act1(z.z, act, slope), act1(z.w, act, slope));
}

inline int grid_of(long units) {
    const long b = (units + kT - 1) / kT;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// x [B,H,W,C] -> out [B,Ho,Wo,C], source pixel (ho * H / Ho, wo * W / Wo): exact for integer factors either way
__global__ __launch_bounds__(kT) void resize_nearest_nhwc_kernel(const float *__restrict__ x, int C4, int H, int W, int Ho, int Wo,
                                                                float *__restrict__ out, long units) {
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4);
        const decltype(u) p = u / C4;
        const int wo = (int)(p % Wo);
        const decltype(u) bh = p / Wo;
        const int ho = (int)(bh % Ho);
        const long b = (long)(bh / Ho);
        const int h = (int)((long)ho * H / Ho), w = (int)((long)wo * W / Wo);
        st4(out + (size_t)u * 4, ld4(x + (((b * H + h) * W + w) * C4 + c) * 4));
    });
}

// x [P, C] -> out [parts][P, C / parts]
__global__ __launch_bounds__(kT) void act_split_nhwc_kernel(const float *__restrict__ x, long part_stride, int C4, int Cp4, int act, float slope,
                                                           float *__restrict__ out, long units) {
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4);
        const decltype(u) p = u / C4;
        const int part = c / Cp4;
        st4(out + (long)part * part_stride + ((long)p * Cp4 + (c - part * Cp4)) * 4, act4(ld4(x + (size_t)u * 4), act, slope));
    });
}

// x [B*N, Rx, Sx, C] conv tiles, y [B,H,W,C] cached, map [H,W,3] -> out [parts][B*N, bH, bW, C / parts]
__global__ __launch_bounds__(kT) void scatter_gather_split_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                                      int B, int C, int H, int W, int Rx, int Sx, int bH, int bW,
                                                                      const int32_t *__restrict__ idx, int N,
                                                                      const int32_t *__restrict__ map, int act, float slope, int Cp4,
                                                                      long part_stride, float *__restrict__ out, long units, int hp_shift) {
    kernarg_touch<192>();
    const int C4 = C / 4, RS = bH * bW;
    for_units<kT>(units, [&](auto u) {
        const int c4 = (int)(u % C4);
        const decltype(u) tp = u / C4;
        const int p = (int)(tp % RS);
        const int t = (int)(tp / RS);
        const int b = t / N, n = t - b * N;
        const int h0 = idx[2 * n];
        const int h = h0 + p / bW, w = idx[2 * n + 1] + p % bW;
        // (stacked edits: the tile belongs to the image its third row lies in -- csrc/nhwc_ops.hip seam_lo -- rows beyond are padding)
        const int hlo = hp_shift ? (((h0 + 2) >> hp_shift) << hp_shift) : 0, hhi = hp_shift ? hlo + (1 << hp_shift) : H;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h >= hlo && h < hhi && w >= 0 && w < W) {
            const int32_t *m = map + 3 * ((size_t)h * W + w);
            const int blk = m[0];
            const float4 v = blk >= 0 ? ld4(x + ((((size_t)b * N + blk) * Rx + m[1]) * Sx + m[2]) * C + 4 * c4)
                                      : ld4(y + (((size_t)b * H + h) * W + w) * C + 4 * c4);
            z = act4(v, act, slope);
        }
        const int part = c4 / Cp4;
        st4(out + (long)part * part_stride + ((long)tp * Cp4 + (c4 - part * Cp4)) * 4, z);
    });
}

// x [B,H,W,C], gb [B,H,W,2C] (gamma | beta), scale / shift [1|B, C]
__global__ __launch_bounds__(kT) void spade_modulate_dense_nhwc_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, int aff_sb,
                                                                      const float *__restrict__ gb, int C4, long hw, int leaky,
                                                                      float slope, float *__restrict__ out, long units) {
    kernarg_touch<128>();
    for_units<kT>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const decltype(u) p = u / C4;
        const long b = (long)(p / (decltype(u))hw);  // (hw <= units: it fits whenever u's type does)
        const float4 v = ld4(x + (size_t)u * 4), gamma = ld4(gb + (size_t)p * 8 * C4 + c), beta = ld4(gb + (size_t)p * 8 * C4 + 4 * C4 + c);
        const float4 sc = ld4(scale + b * aff_sb + c), sh = ld4(shift + b * aff_sb + c);
        // scale, then shift (two separately rounded ops: -ffp-contract=off), 1 + gamma, product, + beta, leaky: the order of
        // sige_normalization.py:74-88 with the param-free norm folded into the cached affine
        float4 n = make_float4(sc.x * v.x, sc.y * v.y, sc.z * v.z, sc.w * v.w);
        n = make_float4(sh.x + n.x, sh.y + n.y, sh.z + n.z, sh.w + n.w);
        const float4 g1 = make_float4(1.0f + gamma.x, 1.0f + gamma.y, 1.0f + gamma.z, 1.0f + gamma.w);
        float4 z = make_float4(n.x * g1.x, n.y * g1.y, n.z * g1.z, n.w * g1.w);
        z = make_float4(z.x + beta.x, z.y + beta.y, z.z + beta.z, z.w + beta.w);
        if (leaky) z = act4(z, SIGE_HIP_ACT_LEAKY, slope);
        st4(out + (size_t)u * 4, z);
    });
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool act_ok(int a) { return a == SIGE_HIP_ACT_IDENTITY || a == SIGE_HIP_ACT_RELU || a == SIGE_HIP_ACT_LEAKY; }

}  // namespace
}  // namespace sige

using namespace sige;

extern "C" int sige_hip_resize_nearest_nhwc_f32(const float *x, int B, int C, int H, int W, int Ho, int Wo, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_resize_nearest_nhwc_f32, x, B, C, H, W, Ho, Wo, out, stream);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || !x || !out) return SIGE_HIP_EINVAL;
    // integer factors only: torch computes the source index in floating point, which agrees with the integer form exactly there
    if (!((Ho % H == 0 || H % Ho == 0) && (Wo % W == 0 || W % Wo == 0))) return SIGE_HIP_EUNSUPPORTED;
    if (C % 4 || !al16(x) || !al16(out)) return SIGE_HIP_EUNSUPPORTED;
    // (stacked edits: an integer factor maps image e's rows onto image e's rows of the tall result -- nothing to do)
    const long units = (long)B * Ho * Wo * (C / 4);
    resize_nearest_nhwc_kernel<<<grid_of(units), kT, 0, as_stream(stream)>>>(x, C / 4, H, W, Ho, Wo, out, units);
    return launch_status();
}

extern "C" int sige_hip_act_split_nhwc_f32(const float *x, int64_t pixels, int C, int parts, int64_t part_stride, int activation,
                                           float slope, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_act_split_nhwc_f32, x, pixels, C, parts, part_stride, activation, slope, out, stream);
    if (pixels < 0 || C <= 0 || parts <= 0) return SIGE_HIP_EINVAL;
    if (part_stride < pixels * (C / parts) || part_stride % 4) return SIGE_HIP_EINVAL;
    if (!act_ok(activation)) return SIGE_HIP_EUNSUPPORTED;
    if (pixels == 0) return SIGE_HIP_OK;
    if (!x || !out) return SIGE_HIP_EINVAL;
    if (C % (4 * parts) || !al16(x) || !al16(out)) return SIGE_HIP_EUNSUPPORTED;
    const long units = (long)pixels * (C / 4);
    act_split_nhwc_kernel<<<grid_of(units), kT, 0, as_stream(stream)>>>(x, (long)part_stride, C / 4, C / 4 / parts, activation, slope, out, units);
    return launch_status();
}

extern "C" int sige_hip_scatter_gather_split_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W, int Rx, int Sx, int bH,
                                                      int bW, const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                      int activation, float slope, int parts, int64_t part_stride, float *out,
                                                      void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_split_nhwc_f32, (sige::CountOf<10, 11>), x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, activation, slope, parts, part_stride, out, stream);
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || Rx <= 0 || Sx <= 0 || bH <= 0 || bW <= 0 || N < 0 || parts <= 0) return SIGE_HIP_EINVAL;
    if (!act_ok(activation)) return SIGE_HIP_EUNSUPPORTED;
    const int hp_shift = stacked_shift(H);  // (stacked edits: halo rows beyond a tile's own image are zero padding)
    if (hp_shift < 0 || (hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    if (C % (4 * parts) || !al16(x) || !al16(y) || !al16(out)) return SIGE_HIP_EUNSUPPORTED;
    if (part_stride < (int64_t)B * N * bH * bW * (C / parts) || part_stride % 4) return SIGE_HIP_EINVAL;  // (a part holds every tile of THIS mask)
    const long units = (long)B * N * bH * bW * (C / 4);
    scatter_gather_split_nhwc_kernel<<<grid_of(units), kT, 0, as_stream(stream)>>>(x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map,
                                                                                  activation, slope, C / 4 / parts, (long)part_stride, out, units, hp_shift);
    return launch_status();
}

extern "C" int sige_hip_spade_modulate_dense_nhwc_f32(const float *x, const float *scale, const float *shift, int affineB, const float *gb,
                                                      int B, int C, int H, int W, int leaky, float slope, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_spade_modulate_dense_nhwc_f32, x, scale, shift, affineB, gb, B, C, H, W, leaky, slope, out, stream);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !x || !scale || !shift || !gb || !out) return SIGE_HIP_EINVAL;
    if (affineB != 1 && affineB != B) return SIGE_HIP_EINVAL;
    if (C % 4 || !al16(x) || !al16(gb) || !al16(out) || !al16(scale) || !al16(shift)) return SIGE_HIP_EUNSUPPORTED;  // (per pixel: stacked edits change nothing)
    const long units = (long)B * H * W * (C / 4);
    spade_modulate_dense_nhwc_kernel<<<grid_of(units), kT, 0, as_stream(stream)>>>(x, scale, shift, affineB > 1 ? C : 0, gb, C / 4, (long)H * W,
                                                                                  leaky, slope, out, units);
    return launch_status();
}
