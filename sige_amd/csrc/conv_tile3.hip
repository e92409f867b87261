// Tile conv v3 (conv_tile3.hpp): C ABI.  One entry point for both staging sources and both destinations; weights in the
// exact-fp32 layout of the dense-layer kernel (sige_hip_wide_conv_pack(prec = 2)).
#include "conv_tile3.hpp"

namespace sige {
template <> void launch_conv_tile3_gather<2>(const Tile3Args &, bool, bool, bool, hipStream_t);
template <> void launch_conv_tile3_sg<2>(const Tile3Args &, bool, bool, hipStream_t);
template <> void launch_conv_tile3_gather<2, WIDE_F16>(const Tile3Args &, bool, bool, bool, hipStream_t);
template <> void launch_conv_tile3_sg<2, WIDE_F16>(const Tile3Args &, bool, bool, hipStream_t);
template <> void launch_conv_tile3_gather<4, WIDE_F16>(const Tile3Args &, bool, bool, bool, hipStream_t);
template <> void launch_conv_tile3_sg<4, WIDE_F16>(const Tile3Args &, bool, bool, hipStream_t);
// fp16 operands: 4 tiles per workgroup from this many 2-tile workgroups on (SIGE_HIP_TUNE_TILE3_F16_TPW4_MIN = -1: this value)
constexpr int kTile3F16Tpw4Min = 0;  // (measured, profiles/r6i_tile3_f16_tpw4_bench.json: wins at 834 workgroups only, -0.2 % on the forward: never)
int flush_held_conv();  // (block_conv.hip: a shortcut conv held by sige_hip_conv_pair_begin is launched on its own first)
}  // namespace sige

using namespace sige;

extern "C" int sige_hip_tile_conv3_supported(int C1, int C2, int Cout) {
    return (C1 > 0 && C2 >= 0 && C1 % 64 == 0 && C2 % 64 == 0 && Cout > 0 && Cout % 64 == 0) ? 1 : 0;
}

// (no plan hook: the extern "C" wrapper below and the routing entry points of block_conv.hip record themselves)
int sige::tile_conv3_launch(
        int source, const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
        const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
        const float *scale, const float *shift, int affineB, int activation,
        const float *packed, const float *bias, int Cout,
        int to_full, int offsetH, int offsetW, int Ho, int Wo, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        const float *out_scale, const float *out_shift, int out_activation,
        float *twin0, const float *twin_scale0, const float *twin_shift0,
        float *twin1, const float *twin_scale1, const float *twin_shift1,
        float *out, void *stream, int prec, int y_f16, int residual_f16) {
    if (source != T3_GATHER && source != T3_SCATTER_GATHER) return SIGE_HIP_EINVAL;
    if (prec != WIDE_F32 && prec != WIDE_F16) return SIGE_HIP_EUNSUPPORTED;
    if (y_f16 && source != T3_SCATTER_GATHER) return SIGE_HIP_EINVAL;
    if ((y_f16 || residual_f16) && prec != WIDE_F16) return SIGE_HIP_EUNSUPPORTED;  // (fp16-stored caches: the f16 form only)
    if (B < 0 || N < 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || H <= 0 || W <= 0) return SIGE_HIP_EINVAL;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    const bool sg = source == T3_SCATTER_GATHER;
    if (!x || !packed || !out || !active_indices || ((C2 || sg) && !x2) || (sg && (!scatter_map || Rx <= 0 || Sx <= 0))) return SIGE_HIP_EINVAL;
    if (sg && (C2 || upsample2x)) return SIGE_HIP_EUNSUPPORTED;  // (one channel range; round 6: the cached affine + SiLU in the staging path too -- SD's conv2)
    if (!sige_hip_tile_conv3_supported(C1, C2, Cout)) return SIGE_HIP_EUNSUPPORTED;
    if ((scale == nullptr) != (shift == nullptr)) return SIGE_HIP_EINVAL;
    if (scale && affineB != 1 && affineB != B) return SIGE_HIP_EINVAL;
    if (!scale && activation != SIGE_HIP_ACT_IDENTITY) return SIGE_HIP_EUNSUPPORTED;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((out_scale == nullptr) != (out_shift == nullptr)) return SIGE_HIP_EINVAL;
    if (out_scale && out_activation != SIGE_HIP_ACT_IDENTITY && out_activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (upsample2x && ((H | W) & 1)) return SIGE_HIP_EINVAL;
    if (to_full && (Ho <= 0 || Wo <= 0)) return SIGE_HIP_EINVAL;
    if (!to_full && (residual || x1 || twin0 || twin1)) return SIGE_HIP_EINVAL;
    if (x1 && (!residual || !table1 || N1 < 0 || R1 <= 0 || S1 <= 0 || gW1 <= 0)) return SIGE_HIP_EINVAL;
    if ((twin0 && !(twin_scale0 && twin_shift0)) || (twin1 && !(twin_scale1 && twin_shift1))) return SIGE_HIP_EINVAL;
    const int Cin = C1 + C2;
    const long src_px = sg ? (long)B * H * W : (long)B * (H >> (upsample2x ? 1 : 0)) * (W >> (upsample2x ? 1 : 0));
    if (src_px * Cin >= (1L << 29) || (long)B * N * 36 * Cin >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;  // 32-bit byte offsets in the kernel
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al(x) || !al(x2) || !al(packed) || !al(out) || !al(bias) || !al(scale) || !al(shift) || !al(residual) || !al(x1) ||
        !al(out_scale) || !al(out_shift) || !al(twin0) || !al(twin1) || !al(twin_scale0) || !al(twin_shift0) || !al(twin_scale1) || !al(twin_shift1))
        return SIGE_HIP_EUNSUPPORTED;
    Tile3Args a{};
    a.x = x; a.x2 = x2 ? x2 : x; a.idx = active_indices; a.map = scatter_map; a.packed = packed; a.bias = bias;
    a.scale = scale; a.shift = shift; a.residual = residual; a.oscale = out_scale; a.oshift = out_shift; a.out = out;
    a.twin0 = twin0; a.tscale0 = twin_scale0; a.tshift0 = twin_shift0; a.twin1 = twin1; a.tscale1 = twin_scale1; a.tshift1 = twin_shift1;
    a.x1 = x1; a.table1 = table1; a.gW1 = gW1; a.N1 = N1; a.R1 = R1; a.S1 = S1;
    a.B = B; a.N = N; a.T = B * N; a.H = H; a.W = W; a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.up = upsample2x ? 1 : 0;
    a.act = activation; a.oact = out_activation; a.aff_sb = (scale && affineB > 1) ? Cin : 0;
    if (a.aff_sb && N % Tile3Geo<2>::TPW) return SIGE_HIP_EUNSUPPORTED;  // (a workgroup's tiles must share one image's affine)
    a.Rx = Rx; a.Sx = Sx; a.Ho = Ho; a.Wo = Wo; a.offH = offsetH; a.offW = offsetW;
    a.ntn = Cout / 64; a.nchunks = Cin / 64; a.nchunks1 = C1 / 64;
    a.hp_shift = stacked_shift(H);
    if (a.hp_shift < 0 || (a.hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    a.res_f16 = residual_f16 ? 1 : 0;
    const int rc = flush_held_conv();
    if (rc != SIGE_HIP_OK) return rc;
    hipStream_t st = as_stream(stream);
    if (prec == WIDE_F16) {
        int t4min = tuning(SIGE_HIP_TUNE_TILE3_F16_TPW4_MIN);
        if (t4min < 0) t4min = kTile3F16Tpw4Min;
        const bool four = t4min > 0 && (long)((a.T + 1) / 2) * a.ntn >= t4min && !(a.aff_sb && N % 4);
        if (four) {
            if (sg) launch_conv_tile3_sg<4, WIDE_F16>(a, to_full != 0, y_f16 != 0, st);
            else launch_conv_tile3_gather<4, WIDE_F16>(a, scale != nullptr, C2 > 0, to_full != 0, st);
        } else if (sg) launch_conv_tile3_sg<2, WIDE_F16>(a, to_full != 0, y_f16 != 0, st);
        else launch_conv_tile3_gather<2, WIDE_F16>(a, scale != nullptr, C2 > 0, to_full != 0, st);
    } else {
        if (sg) launch_conv_tile3_sg<2>(a, to_full != 0, false, st);
        else launch_conv_tile3_gather<2>(a, scale != nullptr, C2 > 0, to_full != 0, st);
    }
    return launch_status(1);
}

extern "C" int sige_hip_tile_conv3_nhwc_f32(
        int source, const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
        const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
        const float *scale, const float *shift, int affineB, int activation,
        const float *packed, const float *bias, int Cout,
        int to_full, int offsetH, int offsetW, int Ho, int Wo, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        const float *out_scale, const float *out_shift, int out_activation,
        float *twin0, const float *twin_scale0, const float *twin_shift0,
        float *twin1, const float *twin_scale1, const float *twin_shift1,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_tile_conv3_nhwc_f32, (sige::CountOf<9, 10>, sige::CountOf<28, 31>), source, x, x2, B, C1, C2, H, W, upsample2x, active_indices, N, scatter_map, Rx, Sx, scale, shift, affineB, activation, packed, bias, Cout, to_full, offsetH, offsetW, Ho, Wo, residual, x1, table1, gH1, gW1, N1, R1, S1, out_scale, out_shift, out_activation, twin0, twin_scale0, twin_shift0, twin1, twin_scale1, twin_shift1, out, stream);
    return tile_conv3_launch(source, x, x2, B, C1, C2, H, W, upsample2x, active_indices, N, scatter_map, Rx, Sx, scale, shift, affineB, activation,
                             packed, bias, Cout, to_full, offsetH, offsetW, Ho, Wo, residual, x1, table1, gH1, gW1, N1, R1, S1,
                             out_scale, out_shift, out_activation, twin0, twin_scale0, twin_shift0, twin1, twin_scale1, twin_shift1, out, stream);
}

// fp16 operands (BASELINE.json configs[4]): `packed` = sige_hip_wide_conv_pack(prec = 0) of the same weight; activations are fp32 in
// HBM and rounded to fp16 (RNE) in the staging path, products exact, accumulation fp32.  y_f16: source 2's cached tensor x2 holds
// halves; residual_f16: `residual` holds halves (the fp16-stored caches of SIGEModel.set_cache_dtype("f16")).
extern "C" int sige_hip_tile_conv3_nhwc_f16c(
        int source, const float *x, const void *x2, int y_f16, int B, int C1, int C2, int H, int W, int upsample2x,
        const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
        const float *scale, const float *shift, int affineB, int activation,
        const float *packed, const float *bias, int Cout,
        int to_full, int offsetH, int offsetW, int Ho, int Wo, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        const float *out_scale, const float *out_shift, int out_activation,
        float *twin0, const float *twin_scale0, const float *twin_shift0,
        float *twin1, const float *twin_scale1, const float *twin_shift1,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_tile_conv3_nhwc_f16c, (sige::CountOf<10, 11>, sige::CountOf<30, 33>), source, x, x2, y_f16, B, C1, C2, H, W, upsample2x, active_indices, N, scatter_map, Rx, Sx, scale, shift, affineB, activation, packed, bias, Cout, to_full, offsetH, offsetW, Ho, Wo, residual, residual_f16, x1, table1, gH1, gW1, N1, R1, S1, out_scale, out_shift, out_activation, twin0, twin_scale0, twin_shift0, twin1, twin_scale1, twin_shift1, out, stream);
    return tile_conv3_launch(source, x, static_cast<const float *>(x2), B, C1, C2, H, W, upsample2x, active_indices, N, scatter_map, Rx, Sx, scale, shift,
                             affineB, activation, packed, bias, Cout, to_full, offsetH, offsetW, Ho, Wo, static_cast<const float *>(residual),
                             x1, table1, gH1, gW1, N1, R1, S1, out_scale, out_shift, out_activation, twin0, twin_scale0, twin_shift0,
                             twin1, twin_scale1, twin_shift1, out, stream, WIDE_F16, y_f16, residual_f16);
}
