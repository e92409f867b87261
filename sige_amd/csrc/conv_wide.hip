// Dense-layer convolution on the fp16 matrix cores (conv_wide.hpp): weight packing, launch plan, C ABI.
//
// Reference: the layers a SIGE network runs densely -- below `sparse_resolution_threshold` in the sparse pass
// (diffusion/models/ddpm_arch/sige_fused_unet.py:112-123) and every conv of the full pass (sige/nn/base.py:85-86) --
// go through F.conv2d (cuDNN / MIOpen) after separate elementwise kernels for the cached affine, SiLU, cat and skip add.
#include "conv_wide.hpp"

namespace sige {

#define SIGE_WIDE_DECLARE(KH, PREC) template <> void launch_conv_wide<KH, PREC, 8>(const WideArgs &, bool, bool, hipStream_t);
SIGE_WIDE_DECLARE(3, WIDE_F16) SIGE_WIDE_DECLARE(3, WIDE_X3) SIGE_WIDE_DECLARE(3, WIDE_F32)
SIGE_WIDE_DECLARE(1, WIDE_F16) SIGE_WIDE_DECLARE(1, WIDE_X3) SIGE_WIDE_DECLARE(1, WIDE_F32)

// a 3x3 dense-layer conv with the block's 1x1 shortcut (tile kernel body) in the same launch: conv_wide_pair_*.hip
using PK11_16 = ConvGeo<1, 1, 4, 16>;
using PK11_32 = ConvGeo<1, 1, 4, 32>;
using PH11_16 = ConvGeoH<1, 1, 4, 16>;
using PH11_32 = ConvGeoH<1, 1, 4, 32>;
#define SIGE_WIDE_PAIR_DECLARE(PREC, GB) template <> void launch_conv_wide_pair<PREC, GB>(const WideArgs &, bool, bool, ConvArgs, hipStream_t);
SIGE_WIDE_PAIR_DECLARE(WIDE_F32, PK11_16) SIGE_WIDE_PAIR_DECLARE(WIDE_F32, PK11_32)
SIGE_WIDE_PAIR_DECLARE(WIDE_X3, PK11_16) SIGE_WIDE_PAIR_DECLARE(WIDE_X3, PK11_32)
SIGE_WIDE_PAIR_DECLARE(WIDE_F16, PH11_16) SIGE_WIDE_PAIR_DECLARE(WIDE_F16, PH11_32)

// packed[ntile][wave][chunk][ks][tap][nt][plane][lane = (kq, j)][e] =
//     plane(w[co = 64 ntile + 32 nt + j][ci = chunk*CC + wave*CW + 16 ks + 8 kq + e][tap] * 2^wshift)
// plane 0 = fp16(v) (RNE), plane 1 = fp16(v - plane 0); 0 beyond Cout / Cin.  One wave's share of a launch is contiguous.
template <typename G>
__global__ void pack_wide_kernel(const float *__restrict__ w, int Cout, int Cin, float wmul, _Float16 *__restrict__ packed, long total) {
    const int nchunks = (Cin + G::CC - 1) / G::CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 8; r /= 8;
        const int j = r % 32; r /= 32;
        const int kq = r % 2; r /= 2;
        const int pl = r % G::NP; r /= G::NP;
        const int nt = r % 2; r /= 2;
        const int tap = r % G::KK; r /= G::KK;
        const int ks = r % G::KS; r /= G::KS;
        const int chunk = r % nchunks; r /= nchunks;
        const int wave = r % 4; r /= 4;
        const int ntile = (int)r;
        const int co = 64 * ntile + 32 * nt + j;
        const int ci = chunk * G::CC + wave * G::CW + 16 * ks + 8 * kq + e;
        float v = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * G::KK + tap] * wmul : 0.0f;
        v = fminf(fmaxf(v, -65504.0f), 65504.0f);
        const _Float16 hi = (_Float16)v;
        packed[i] = pl == 0 ? hi : (_Float16)(v - (float)hi);
    }
}

// exact fp32 form: packed[ntile][wave][chunk][ks][tap][nt][piece][lane = (kq, j)][e] = w[co][ci = ... + 16 ks + 2 (4 piece + e) + kq][tap]
// (k-step 4 piece + e of the eight v_mfma_f32_32x32x2_f32 per tap; lane group kq holds channel 2 s + kq)
template <typename G>
__global__ void pack_wide_f32_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed, long total) {
    const int nchunks = (Cin + G::CC - 1) / G::CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 4; r /= 4;
        const int j = r % 32; r /= 32;
        const int kq = r % 2; r /= 2;
        const int pl = r % 2; r /= 2;
        const int nt = r % 2; r /= 2;
        const int tap = r % G::KK; r /= G::KK;
        const int ks = r % G::KS; r /= G::KS;
        const int chunk = r % nchunks; r /= nchunks;
        const int wave = r % 4; r /= 4;
        const int ntile = (int)r;
        const int co = 64 * ntile + 32 * nt + j;
        const int ci = chunk * G::CC + wave * G::CW + 16 * ks + 2 * (4 * pl + e) + kq;
        packed[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * G::KK + tap] : 0.0f;
    }
}

template <typename G>
static size_t wide_packed_bytes(int Cout, int Cin) {
    const size_t nchunks = (Cin + G::CC - 1) / G::CC, ntn = (Cout + 63) / 64;
    return ntn * 4 * nchunks * G::STEPS * G::STEPB + (size_t)kWidePadSteps * G::STEPB;  // + padding for the prefetch past the last step
}

static bool wide_shape_ok(int C1, int C2, int Cout, int kH, int kW) {
    if (kH != kW || (kH != 1 && kH != 3)) return false;
    const int cc = kH == 3 ? WideGeo<3, WIDE_F16>::CC : WideGeo<1, WIDE_F16>::CC;
    return C1 > 0 && C2 >= 0 && C1 % cc == 0 && C2 % cc == 0 && Cout > 0 && Cout % 64 == 0;
}


// width of a workgroup's output patch: 8 x 8 pixels / two waves per SIMD.  (An 8 x 16 form -- one wave per SIMD with the
// 512-register budget, twice the matrix work per weight byte, a whole chunk of weights in the ring -- was built and measured
// SLOWER on every layer of the DDPM-256 U-Net, 256^2 128->128: 92 vs 73 us, 32^2 768->256: 34 vs 24 us
// (profiles/r3b_wide_bench.jsonl): one in-order wave per SIMD hides nothing.  Removed; WideGeo keeps the parameter.)
static int wide_patch(int W) { (void)W; return 8; }

// K split of a launch with `blocks` output blocks and `nchunks` channel chunks: enough workgroups for two per CU
static int wide_ksplit(long blocks, int nchunks, size_t out_floats, size_t ws_floats) {
    const int force = tuning(SIGE_HIP_TUNE_WIDE_KSPLIT);
    int s = force ? force : (blocks >= 384 ? 1 : (int)((512 + blocks - 1) / blocks));
    s = s < nchunks ? s : nchunks;
    s = s < kWideMaxSplit ? s : kWideMaxSplit;
    while (s > 1 && (size_t)s * out_floats > ws_floats) --s;
    return s < 1 ? 1 : s;
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_wide_conv_supported(int C1, int C2, int Cout, int kH, int kW) {
    return wide_shape_ok(C1, C2, Cout, kH, kW) ? 1 : 0;
}

extern "C" size_t sige_hip_wide_conv_packed_size(int Cout, int Cin, int kH, int kW, int prec) {
    if (!wide_shape_ok(Cin, 0, Cout, kH, kW) || prec < WIDE_F16 || prec > WIDE_F32) return 0;
    size_t bytes;  // (the split-fp16 and the exact-fp32 forms have the same footprint: 4 bytes per weight)
    if (kH == 3) bytes = prec != WIDE_F16 ? wide_packed_bytes<WideGeo<3, WIDE_X3>>(Cout, Cin) : wide_packed_bytes<WideGeo<3, WIDE_F16>>(Cout, Cin);
    else bytes = prec != WIDE_F16 ? wide_packed_bytes<WideGeo<1, WIDE_X3>>(Cout, Cin) : wide_packed_bytes<WideGeo<1, WIDE_F16>>(Cout, Cin);
    return bytes / 4;
}

extern "C" int sige_hip_wide_conv_pack(const float *w, int Cout, int Cin, int kH, int kW, int prec, int wshift,
                                       float *packed, void *stream) {
    if (!w || !packed || wshift < -60 || wshift > 60 || prec < WIDE_F16 || prec > WIDE_F32) return SIGE_HIP_EINVAL;
    if (!wide_shape_ok(Cin, 0, Cout, kH, kW)) return SIGE_HIP_EUNSUPPORTED;
    if (prec == WIDE_F32 && wshift != 0) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    const size_t units = sige_hip_wide_conv_packed_size(Cout, Cin, kH, kW, prec);
    if (hipMemsetAsync(packed, 0, units * 4, st) != hipSuccess) return SIGE_HIP_ELAUNCH;  // (incl. the padding)
    if (prec == WIDE_F32) {
        auto gof = [&](auto g_tag) {
            using G = decltype(g_tag);
            const long total = (long)((wide_packed_bytes<G>(Cout, Cin) - (size_t)kWidePadSteps * G::STEPB) / 4);
            const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
            pack_wide_f32_kernel<G><<<grid, 256, 0, st>>>(w, Cout, Cin, packed, total);
        };
        if (kH == 3) gof(WideGeo<3, WIDE_F32>{}); else gof(WideGeo<1, WIDE_F32>{});
        return launch_status(1);
    }
    const bool x3 = prec == WIDE_X3;
    const float wmul = ldexpf(1.0f, wshift);
    _Float16 *ph = reinterpret_cast<_Float16 *>(packed);
    auto go = [&](auto g_tag) {
        using G = decltype(g_tag);
        const long total = (long)((wide_packed_bytes<G>(Cout, Cin) - (size_t)kWidePadSteps * G::STEPB) / 2);
        const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
        pack_wide_kernel<G><<<grid, 256, 0, st>>>(w, Cout, Cin, wmul, ph, total);
    };
    if (kH == 3) { if (x3) go(WideGeo<3, WIDE_X3>{}); else go(WideGeo<3, WIDE_F16>{}); }
    else { if (x3) go(WideGeo<1, WIDE_X3>{}); else go(WideGeo<1, WIDE_F16>{}); }
    return launch_status(1);
}

extern "C" size_t sige_hip_wide_conv_workspace(int B, int H, int W, int C1, int C2, int Cout, int kH, int kW) {
    if (!wide_shape_ok(C1, C2, Cout, kH, kW) || B <= 0 || H <= 0 || W <= 0) return 0;
    const long blocks = (long)B * ceil_div(H, 8) * ceil_div(W, wide_patch(W)) * (Cout / 64);
    const int cc = kH == 3 ? WideGeo<3, WIDE_F16>::CC : WideGeo<1, WIDE_F16>::CC;
    const size_t out_floats = (size_t)B * H * W * Cout;
    const int s = wide_ksplit(blocks * (wide_patch(W) / 8), (C1 + C2) / cc, out_floats, (size_t)-1);
    return s > 1 ? (size_t)s * out_floats : 0;
}

#ifdef SIGE_WIDE_PROBE
static unsigned long long *g_wprobe_buf = nullptr;
// (measurement build only, not declared in include/sige_hip.h)
extern "C" int sige_hip_wide_probe_read(unsigned long long *host, int workgroups) {
    if (!g_wprobe_buf || workgroups > 4096) return SIGE_HIP_EINVAL;
    return hipMemcpy(host, g_wprobe_buf, (size_t)workgroups * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess
               ? SIGE_HIP_OK : SIGE_HIP_ELAUNCH;
}
extern "C" int sige_hip_wide_probe_clear(void) {
    if (!g_wprobe_buf && hipMalloc(&g_wprobe_buf, 8 * 4096 * sizeof(unsigned long long)) != hipSuccess) { g_wprobe_buf = nullptr; return SIGE_HIP_ELAUNCH; }
    return hipMemset(g_wprobe_buf, 0, 8 * 4096 * sizeof(unsigned long long)) == hipSuccess ? SIGE_HIP_OK : SIGE_HIP_ELAUNCH;
}
#endif

extern "C" int sige_hip_wide_conv_nhwc(const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
                                       const float *scale, const float *shift, int affineB, int activation,
                                       const float *packed, int prec, int wshift, const float *bias, int Cout, int kH, int kW,
                                       const float *residual, const float *out_scale, const float *out_shift, int out_activation,
                                       float *twin0, const float *twin_scale0, const float *twin_shift0,
                                       float *twin1, const float *twin_scale1, const float *twin_shift1,
                                       float *workspace, size_t workspace_floats, float *out, float *stats, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_wide_conv_nhwc, x, x2, B, C1, C2, H, W, upsample2x, scale, shift, affineB, activation, packed, prec, wshift, bias, Cout, kH, kW, residual, out_scale, out_shift, out_activation, twin0, twin_scale0, twin_shift0, twin1, twin_scale1, twin_shift1, workspace, workspace_floats, out, stats, stream);
    if (B <= 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || prec < WIDE_F16 || prec > WIDE_F32) return SIGE_HIP_EINVAL;
    if (!x || (C2 && !x2) || !packed || !out) return SIGE_HIP_EINVAL;
    if (!wide_shape_ok(C1, C2, Cout, kH, kW)) return SIGE_HIP_EUNSUPPORTED;
    if ((scale == nullptr) != (shift == nullptr)) return SIGE_HIP_EINVAL;
    if (scale && affineB != 1 && affineB != B) return SIGE_HIP_EINVAL;
    if (!scale && activation != SIGE_HIP_ACT_IDENTITY) return SIGE_HIP_EUNSUPPORTED;  // (an activation comes with the cached affine)
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (out_scale && out_activation != SIGE_HIP_ACT_IDENTITY && out_activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((out_scale == nullptr) != (out_shift == nullptr)) return SIGE_HIP_EINVAL;
    if (upsample2x && ((H | W) & 1)) return SIGE_HIP_EINVAL;
    const long src_px = (long)B * (H >> (upsample2x ? 1 : 0)) * (W >> (upsample2x ? 1 : 0));
    if (src_px * (C1 > C2 ? C1 : C2) >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;  // 32-bit byte offsets in the kernel
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al(x) || !al(x2) || !al(packed) || !al(out) || !al(bias) || !al(scale) || !al(shift) || !al(residual) || !al(out_scale) ||
        !al(out_shift) || !al(twin0) || !al(twin1) || !al(workspace) || (reinterpret_cast<uintptr_t>(stats) & 7))
        return SIGE_HIP_EUNSUPPORTED;
    WideArgs a{};
    a.x = x; a.x2 = x2 ? x2 : x; a.packed = packed; a.bias = bias; a.scale = scale; a.shift = shift; a.residual = residual;
    a.oscale = out_scale; a.oshift = out_shift; a.oact = out_activation; a.out = out; a.fout = out;
    a.twin0 = twin0; a.tscale0 = twin_scale0; a.tshift0 = twin_shift0;
    a.twin1 = twin1; a.tscale1 = twin_scale1; a.tshift1 = twin_shift1;
    if ((twin0 && !(twin_scale0 && twin_shift0)) || (twin1 && !(twin_scale1 && twin_shift1))) return SIGE_HIP_EINVAL;
    a.wscale = ldexpf(1.0f, -wshift);
    a.stats = reinterpret_cast<float2 *>(stats);
#ifdef SIGE_WIDE_PROBE
    a.probe = g_wprobe_buf;
#endif
    a.B = B; a.H = H; a.W = W; a.C1 = C1; a.C2 = C2; a.Cout = Cout; a.up = upsample2x ? 1 : 0; a.act = activation;
    a.hp_shift = stacked_shift(H);
    if (a.hp_shift < 0 || (a.hp_shift && (B != 1 || (1 << a.hp_shift) % 8 || stats))) return SIGE_HIP_EUNSUPPORTED;  // (8-row patches must not straddle a seam)
    a.aff_sb = (scale && affineB > 1) ? C1 + C2 : 0;
    const int pwo = wide_patch(W);
    a.th = ceil_div(H, 8); a.tw = ceil_div(W, pwo); a.ntn = Cout / 64;
    const int cc = kH == 3 ? WideGeo<3, WIDE_F16>::CC : WideGeo<1, WIDE_F16>::CC;
    a.nchunks = (C1 + C2) / cc; a.nchunks1 = C1 / cc;
    hipStream_t st = as_stream(stream);
    const long blocks = (long)B * a.th * a.tw * a.ntn;
    const size_t out_floats = (size_t)B * H * W * Cout;
    // (an 8 x 16 workgroup is one per CU: the same fill target in units of 8 x 8 blocks)
    a.ksplit = workspace ? wide_ksplit(blocks * (pwo / 8), a.nchunks, out_floats, workspace_floats) : 1;
    if (a.ksplit > 1) {
        a.counters = split_tickets(st, blocks);
        if (!a.counters) a.ksplit = 1;
    }
    a.chunks_per_split = ceil_div(a.nchunks, a.ksplit);
    a.ksplit = ceil_div(a.nchunks, a.chunks_per_split);
    if (a.ksplit > 1) { a.out = workspace; a.split_stride = out_floats; }
    const bool aff = scale != nullptr, cat = C2 > 0;
    // a residual block's 1x1 shortcut held by sige_hip_conv_pair_begin() rides along a 3x3 launch (exact-fp32 or split-operand
    // conv1 with an exact-fp32 shortcut, fp16 conv1 with an fp16 shortcut); anything else held is launched on its own first, so
    // that results and order never depend on pairing
    {
        const int hp = held_shortcut_prec(st);
        const bool fits = kH == 3 && !stats && ((hp == 0 && (prec == WIDE_F32 || prec == WIDE_X3)) || (hp == 1 && prec == WIDE_F16));
        ConvArgs b;
        int bmt = 0;
        if (fits && take_held_shortcut(&b, &bmt)) {
            if (prec == WIDE_F32) { if (bmt == 32) launch_conv_wide_pair<WIDE_F32, PK11_32>(a, aff, cat, b, st); else launch_conv_wide_pair<WIDE_F32, PK11_16>(a, aff, cat, b, st); }
            else if (prec == WIDE_X3) { if (bmt == 32) launch_conv_wide_pair<WIDE_X3, PK11_32>(a, aff, cat, b, st); else launch_conv_wide_pair<WIDE_X3, PK11_16>(a, aff, cat, b, st); }
            else { if (bmt == 32) launch_conv_wide_pair<WIDE_F16, PH11_32>(a, aff, cat, b, st); else launch_conv_wide_pair<WIDE_F16, PH11_16>(a, aff, cat, b, st); }
            return launch_status(1);
        }
        const int rc = flush_held_conv();
        if (rc != SIGE_HIP_OK) return rc;
    }
#define SIGE_WIDE_GO(KH)                                                                            \
    do {                                                                                            \
        if (prec == WIDE_F32) launch_conv_wide<KH, WIDE_F32, 8>(a, aff, cat, st);                   \
        else if (prec == WIDE_X3) launch_conv_wide<KH, WIDE_X3, 8>(a, aff, cat, st);                \
        else launch_conv_wide<KH, WIDE_F16, 8>(a, aff, cat, st);                                    \
    } while (0)
    if (kH == 3) SIGE_WIDE_GO(3); else SIGE_WIDE_GO(1);
#undef SIGE_WIDE_GO
    return launch_status(1);
}
