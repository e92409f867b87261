// Stacked-block convolution for gfx950: C ABI, weight packing, tile-shape choice.
//
// The MFMA kernel itself is the template in conv_mfma.hpp, instantiated per tile
// geometry in conv_k3s1.hip / conv_k1.hip / conv_k3s2.hip (parallel compilation).
// The reference hands this conv to F.conv2d (sige/nn/base.py:88-89; cuDNN/MIOpen
// see a batch of T tiny 6x6 images) after a separate gather kernel
// (sige/cuda/gather_kernel.cu:7-67) or scatter_gather kernel
// (scatter_gather_kernel.cu:8-67).
#include <mutex>

#include "conv_mfma.hpp"
#include "conv_tile3.hpp"  // (tile_conv3_launch: the routing entry points at the end of this file)

namespace sige {

// ---- weight packing -------------------------------------------------------
// packed[ng][chunk][wave][f][kq][j][e] = w[co = MT*ng + j][ci][tap]   (0 beyond Cin/Cout)
//   u = 4f + e,  q = u / KK,  tap = u % KK,  ci = chunk*CC + wave*CW + NL*q + kq
template <int KK, int MT>
__global__ void pack_weights_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                    long total) {
    constexpr int NL = 64 / MT, CW = (KK == 1 ? 16 : 4) * NL, CC = 4 * CW, L = (CW / NL) * KK, F = L / 4;
    const int nchunks = ((Cin + CC - 1) / CC + 1) & ~1;  // padded to even: the 8-wave kernels read chunk PAIRS
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 4; r /= 4;
        const int j = r % MT; r /= MT;
        const int kq = r % NL; r /= NL;
        const int f = r % F; r /= F;
        const int wave = r % 4; r /= 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int ng = (int)r;
        const int u = 4 * f + e, q = u / KK, tap = u % KK;
        const int ci = chunk * CC + wave * CW + NL * q + kq;
        const int co = MT * ng + j;
        packed[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f;
    }
}

// fp16 packing for the f16-compute kernels (ConvGeoH): 8 halves per lane and k-step,
//   packed[ng][chunk][wave][u][kq][j][e] = half(w[co = MT*ng + j][ci][tap]),  u = slab * KK + tap,
//   ci = chunk*CC + wave*CW + slab*SLAB + 8*kq + e
template <int KK, int MT>
__global__ void pack_weights_h_kernel(const float *__restrict__ w, int Cout, int Cin, _Float16 *__restrict__ packed, long total) {
    using G = ConvGeoH<(KK == 1 ? 1 : 3), 1, (KK == 1 ? 4 : 6), MT>;  // (stride / tile edge do not enter the layout)
    const int nchunks = ((Cin + G::CC - 1) / G::CC + 1) & ~1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 8; r /= 8;
        const int j = r % MT; r /= MT;
        const int kq = r % G::NL; r /= G::NL;
        const int u = r % G::L; r /= G::L;
        const int wave = r % 4; r /= 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int ng = (int)r;
        const int slab = u / KK, tap = u % KK;
        const int ci = chunk * G::CC + wave * G::CW + slab * G::SLAB + 8 * kq + e;
        const int co = MT * ng + j;
        packed[i] = (_Float16)((ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f);
    }
}

// split fp16 packing for ConvGeoX: two planes per k-step,
//   packed[ng][chunk][wave][u][plane][kq][j][e],  plane 0 = hi = fp16(v), plane 1 = lo = fp16(v - hi),  v = w * 2^S
// S is chosen ON THE DEVICE so that max |w| * 2^S lies in [2^13, 2^14) (the lo parts of all but the tiniest weights are then
// normal fp16 numbers): wmax_bits_kernel reduces max |w| into header[0], every pack thread derives S from it, thread 0
// writes header[1] = 2^-S, which the conv kernel multiplies back in (ConvArgs::wscale_ptr).  No host synchronisation.
__global__ void wmax_bits_kernel(const float *__restrict__ w, long n, unsigned *__restrict__ header) {
    float m = 0.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = fabsf(w[i]);
        m = (v == v && v <= 3.0e38f) ? fmaxf(m, v) : m;  // (NaN / inf do not steer the scale)
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(header, __float_as_uint(m));  // (non-negative floats order like their bit patterns)
}

__device__ __forceinline__ int x3_shift(unsigned wmax_bits) {
    const int e = (int)((wmax_bits >> 23) & 0xff);  // biased exponent of max |w|: max |w| in [2^(e-127), 2^(e-126))
    if (e == 0) return 0;                             // all-zero (or subnormal) weights
    const int s = 13 - (e - 127);
    return s < -40 ? -40 : (s > 40 ? 40 : s);
}

template <int KK, int MT>
__global__ void pack_weights_x_kernel(const float *__restrict__ w, int Cout, int Cin, _Float16 *__restrict__ packed, long total,
                                      float *__restrict__ header) {
    using G = ConvGeoH<(KK == 1 ? 1 : 3), 1, (KK == 1 ? 4 : 6), MT>;
    const int nchunks = ((Cin + G::CC - 1) / G::CC + 1) & ~1;
    const int S = x3_shift(__float_as_uint(header[0]));
    const float wmul = ldexpf(1.0f, S);
    if (blockIdx.x == 0 && threadIdx.x == 0) header[1] = ldexpf(1.0f, -S);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 8; r /= 8;
        const int j = r % MT; r /= MT;
        const int kq = r % G::NL; r /= G::NL;
        const int pl = r % 2; r /= 2;
        const int u = r % G::L; r /= G::L;
        const int wave = r % 4; r /= 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int ng = (int)r;
        const int slab = u / KK, tap = u % KK;
        const int ci = chunk * G::CC + wave * G::CW + slab * G::SLAB + 8 * kq + e;
        const int co = MT * ng + j;
        float v = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * KK + tap] * wmul : 0.0f;
        v = fminf(fmaxf(v, -65504.0f), 65504.0f);
        const _Float16 hi = (_Float16)v;
        packed[i] = pl == 0 ? hi : (_Float16)(v - (float)hi);
    }
}

// size of the packed f16 weights of one tile shape, in 4-byte units (the Python side allocates fp32 storage)
static size_t packed_units_h(int Cout, int Cin, int KK, int MT) {
    const int NL = 64 / MT, CW = (KK == 1 ? 2 : 1) * NL * 8, CC = 4 * CW, L = (KK == 1 ? 2 : 1) * KK;
    return (size_t)ceil_div(Cout, MT) * ((ceil_div(Cin, CC) + 1) & ~1) * 4 * L * 64 * 4;
}

// split fp16: two planes; the two tile shapes are followed by a 16-unit header (max |w| bits, 2^-S)
static size_t packed_units_x(int Cout, int Cin, int KK, int MT) { return 2 * packed_units_h(Cout, Cin, KK, MT); }
constexpr size_t kX3Header = 16;

static size_t packed_floats(int Cout, int Cin, int KK, int MT) {
    const int NL = 64 / MT, CW = (KK == 1 ? 16 : 4) * NL, CC = 4 * CW, F = (CW / NL) * KK / 4;
    return (size_t)ceil_div(Cout, MT) * ((ceil_div(Cin, CC) + 1) & ~1) * 4 * F * 64 * 4;  // chunk count padded to even
}

// ---- any-shape direct kernel (groups, odd tiles): one lane per output -------
__global__ void block_conv_direct_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                         const float *__restrict__ bias, float *__restrict__ out,
                                         int T, int Cin, int R, int S, int Cout, int kH, int kW,
                                         int strH, int strW, int dilH, int dilW, int groups, int Ro, int So, long total) {
    const int cig = Cin / groups, cog = Cout / groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int ox = r % So; r /= So;
        const int oy = r % Ro; r /= Ro;
        const int co = r % Cout; r /= Cout;
        const int t = (int)r;
        const int g = co / cog;
        float acc = bias ? bias[co] : 0.0f;
        for (int ci = 0; ci < cig; ++ci) {
            const float *xp = x + (((size_t)t * Cin + g * cig + ci) * R + oy * strH) * S + ox * strW;
            const float *wq = w + ((size_t)co * cig + ci) * kH * kW;
            for (int ky = 0; ky < kH; ++ky)
                for (int kx = 0; kx < kW; ++kx) acc = fmaf(xp[ky * dilH * S + kx * dilW], wq[ky * kW + kx], acc);
        }
        out[i] = acc;
    }
}

static int mfma_kind(int kH, int kW, int R, int S, int strH, int strW, int groups) {
    if (groups != 1 || kH != kW || R != S || strH != strW) return 0;
    if (kH == 3 && strH == 1 && R == 6) return 1;
    if (kH == 1 && strH == 1 && R == 4) return 2;
    if (kH == 3 && strH == 2 && R == 5) return 3;
    return 0;
}

// specialisations defined in the per-geometry translation units
#define SIGE_CONV_DECLARE(G, NB, LAY, W)                                                                 \
    template <> void launch_conv_geo<G, NB, SRC_TILES, DST_TILES, LAY, W>(ConvArgs, int, hipStream_t);          \
    template <> void launch_conv_geo<G, NB, SRC_GATHER, DST_TILES, LAY, W>(ConvArgs, int, hipStream_t);         \
    template <> void launch_conv_geo<G, NB, SRC_GATHER, DST_NCHW, LAY, W>(ConvArgs, int, hipStream_t);          \
    template <> void launch_conv_geo<G, NB, SRC_SCATTER_GATHER, DST_TILES, LAY, W>(ConvArgs, int, hipStream_t);
#define SIGE_CONV_DECLARE_LAYOUTS(G, NB) SIGE_CONV_DECLARE(G, NB, LAYOUT_NCHW, 4) SIGE_CONV_DECLARE(G, NB, LAYOUT_NHWC, 4)
using K31_16 = ConvGeo<3, 1, 6, 16>;
using K31_32 = ConvGeo<3, 1, 6, 32>;
using K11_16 = ConvGeo<1, 1, 4, 16>;
using K11_32 = ConvGeo<1, 1, 4, 32>;
using K32_16 = ConvGeo<3, 2, 5, 16>;
using K32_32 = ConvGeo<3, 2, 5, 32>;
SIGE_CONV_DECLARE_LAYOUTS(K31_16, 1)
SIGE_CONV_DECLARE_LAYOUTS(K31_16, 2)
SIGE_CONV_DECLARE_LAYOUTS(K31_32, 1)
SIGE_CONV_DECLARE_LAYOUTS(K31_32, 2)
SIGE_CONV_DECLARE_LAYOUTS(K11_16, 1)
SIGE_CONV_DECLARE_LAYOUTS(K11_16, 2)
SIGE_CONV_DECLARE_LAYOUTS(K11_32, 1)
SIGE_CONV_DECLARE_LAYOUTS(K11_32, 2)
SIGE_CONV_DECLARE_LAYOUTS(K32_16, 1)  // stride 2: NB = 1 only (conv_k3s2*.hip)
SIGE_CONV_DECLARE_LAYOUTS(K32_32, 1)
// scatter_gather -> conv -> scatter in one launch (channels-last 3x3): conv_k3s1_nhwc*.hip
#define SIGE_CONV_DECLARE_SG_FULL(G, NB, W) \
    template <> void launch_conv_geo<G, NB, SRC_SCATTER_GATHER, DST_NCHW, LAYOUT_NHWC, W>(ConvArgs, int, hipStream_t);
SIGE_CONV_DECLARE_SG_FULL(K31_16, 1, 4) SIGE_CONV_DECLARE_SG_FULL(K31_16, 2, 4)
SIGE_CONV_DECLARE_SG_FULL(K31_32, 1, 4) SIGE_CONV_DECLARE_SG_FULL(K31_32, 2, 4)
SIGE_CONV_DECLARE_SG_FULL(K31_16, 1, 8) SIGE_CONV_DECLARE_SG_FULL(K31_16, 2, 8)
SIGE_CONV_DECLARE_SG_FULL(K31_32, 1, 8) SIGE_CONV_DECLARE_SG_FULL(K31_32, 2, 8)
// 8-wave workgroups: channels-last, stride 1 (conv_k3s1_nhwc_w8.hip, conv_k1_nhwc_w8.hip)
SIGE_CONV_DECLARE(K31_16, 1, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K31_16, 2, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K31_32, 1, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K31_32, 2, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K11_16, 1, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K11_16, 2, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K11_32, 1, LAYOUT_NHWC, 8)
SIGE_CONV_DECLARE(K11_32, 2, LAYOUT_NHWC, 8)

// f16-compute forms (ConvGeoH; channels-last, 4 waves): conv_k*_nhwc_h.hip
using H31_16 = ConvGeoH<3, 1, 6, 16>;
using H31_32 = ConvGeoH<3, 1, 6, 32>;
using H11_16 = ConvGeoH<1, 1, 4, 16>;
using H11_32 = ConvGeoH<1, 1, 4, 32>;
SIGE_CONV_DECLARE(H31_16, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(H31_16, 2, LAYOUT_NHWC, 4)
SIGE_CONV_DECLARE(H31_32, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(H31_32, 2, LAYOUT_NHWC, 4)
SIGE_CONV_DECLARE(H11_16, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(H11_16, 2, LAYOUT_NHWC, 4)
SIGE_CONV_DECLARE(H11_32, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(H11_32, 2, LAYOUT_NHWC, 4)
// (no f16-compute form of the stride-2 geometry: 4 tiles x 25 pixels x 128 channels per chunk = 13 staging slots per
//  lane do not fit the register file; those few convs -- the U-Net's downsamplers -- stay on the fp32 matrix path)
SIGE_CONV_DECLARE_SG_FULL(H31_16, 1, 4) SIGE_CONV_DECLARE_SG_FULL(H31_16, 2, 4)
SIGE_CONV_DECLARE_SG_FULL(H31_32, 1, 4) SIGE_CONV_DECLARE_SG_FULL(H31_32, 2, 4)

// split-fp16-operand forms (ConvGeoX; channels-last, 4 waves, NB = 1): conv_k*_nhwc_x.hip
using X31_16 = ConvGeoX<3, 1, 6, 16>;
using X31_32 = ConvGeoX<3, 1, 6, 32>;
using X11_16 = ConvGeoX<1, 1, 4, 16>;
using X11_32 = ConvGeoX<1, 1, 4, 32>;
SIGE_CONV_DECLARE(X31_16, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(X31_32, 1, LAYOUT_NHWC, 4)
SIGE_CONV_DECLARE(X11_16, 1, LAYOUT_NHWC, 4) SIGE_CONV_DECLARE(X11_32, 1, LAYOUT_NHWC, 4)
SIGE_CONV_DECLARE_SG_FULL(X31_16, 1, 4) SIGE_CONV_DECLARE_SG_FULL(X31_32, 1, 4)

// scatter_gather source with an fp16-stored cache (conv_k3s1_nhwc*_c16.hip)
#define SIGE_CONV_DECLARE_C16(G, NB)                                                        \
    template <> void launch_conv_c16<G, NB, DST_TILES>(ConvArgs, int, hipStream_t);         \
    template <> void launch_conv_c16<G, NB, DST_NCHW>(ConvArgs, int, hipStream_t);
SIGE_CONV_DECLARE_C16(K31_16, 1) SIGE_CONV_DECLARE_C16(K31_16, 2) SIGE_CONV_DECLARE_C16(K31_32, 1) SIGE_CONV_DECLARE_C16(K31_32, 2)
SIGE_CONV_DECLARE_C16(H31_16, 1) SIGE_CONV_DECLARE_C16(H31_16, 2) SIGE_CONV_DECLARE_C16(H31_32, 1) SIGE_CONV_DECLARE_C16(H31_32, 2)
SIGE_CONV_DECLARE_C16(X31_16, 1) SIGE_CONV_DECLARE_C16(X31_32, 1)

// ---- cross-workgroup K split: deterministic second pass ------------------------
// out[i] = sum_s ws[s][i] + bias[channel(i)] + residual[i]   (channels-last: channel = i mod C)
__global__ __launch_bounds__(256) void splitk_reduce_nhwc_kernel(const float *__restrict__ ws, int S, size_t stride, size_t n4, int C,
                                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                                const float *__restrict__ oscale, const float *__restrict__ oshift, int oact,
                                                                float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        // everything this element needs is requested before the first value is used: bias, residual and the
        // out-affine entries do not depend on the partial sums (loaded after them, each was one more memory round trip)
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = (int)((4 * i) % C);
        float4 b = zero, r = zero, os = zero, oh = zero;
        if (bias) b = *reinterpret_cast<const float4 *>(bias + c);
        if (residual) r = *reinterpret_cast<const float4 *>(residual + 4 * i);
        if (oscale) {
            os = *reinterpret_cast<const float4 *>(oscale + c);
            oh = *reinterpret_cast<const float4 *>(oshift + c);
        }
        // all (<= 8) partials in flight at once, added in split order
        float4 p[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) p[s] = *reinterpret_cast<const float4 *>(ws + (size_t)(s < S ? s : S - 1) * stride + 4 * i);
        float4 v = p[0];
#pragma unroll
        for (int s = 1; s < 8; ++s)
            if (s < S) { v.x += p[s].x; v.y += p[s].y; v.z += p[s].z; v.w += p[s].w; }
        if (bias) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (residual) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        if (oscale) {
            v.x = os.x * v.x; v.y = os.y * v.y; v.z = os.z * v.z; v.w = os.w * v.w;
            v.x = oh.x + v.x; v.y = oh.y + v.y; v.z = oh.z + v.z; v.w = oh.w + v.w;
            if (oact == SIGE_HIP_ACT_SWISH) { v.x = swish(v.x); v.y = swish(v.y); v.z = swish(v.z); v.w = swish(v.w); }
        }
        *reinterpret_cast<float4 *>(out + 4 * i) = v;
    }
}

// K split factor for a conv whose smallest-tile grid (16 pixels x 16 channels per workgroup) cannot
// cover the chip: every wave then runs K/4 MFMA steps back to back however few tiles there are
// (the 8x8 layers of the U-Net: 64 pixels, K = 4608..9216), so the chunks are shared out.
static int ksplit_for(long blocks, int nchunks, int cap) {
    // (the second pass costs ~4.7 us per launch; measured on the DDPM-256 dense remainder the split still
    //  wins 140 us per forward: 923 vs 1064 us over its 58 conv launches)
    const int force = tuning(SIGE_HIP_TUNE_CONV_KSPLIT);  // (a measurement build's knob; 0 in the product)
    if (cap <= 1 || (blocks >= 224 && !force)) return 1;
    int s = force ? force : (int)((224 + blocks - 1) / blocks);
    s = s < nchunks / 2 ? s : nchunks / 2;  // >= 2 chunks per split (the software pipeline's depth)
    s = s < 8 ? s : 8;
    s = s < cap ? s : cap;
    return s < 1 ? 1 : s;
}

// Output block of a workgroup: the largest of (MT x NB*MT) in
//   32x64, 32x32, 16x32, 16x16   (pixels x output channels)
// whose grid still covers the 256 CUs (every workgroup runs the full K, the f32
// matrix pipe is saturated by one wave per SIMD, so a grid below ~1 workgroup per
// CU leaves matrix cores idle while a larger block only saves operand traffic).
// (SIGE_HIP_TUNE_CONV_TILE_MT / _NB override the choice in a measurement build.)
// Unsplit launches that 32 pixel x 64 channel output blocks (NB = 2) would fill the chip with: from this many such blocks on they
// take 32 x 32 blocks (NB = 1) instead.  The NB = 2 kernels of the exact-fp32 3x3 geometry need 290-320 registers = ONE workgroup
// per CU, the NB = 1 kernels 150-190 = two or three: on a grid of several blocks per CU the start-up of one workgroup runs under
// the K loop of another.  Measured (tools/plan_policy_bench.py, profiles/r4k_plan_policy*.json; same bits either way): exact fp32
// one image at 1.2 / 5 / 15 % edit 1.406 / 1.874 / 2.509 -> 1.406 / 1.765 / 2.306 ms, 8 stacked edits 6.26 -> 5.61 ms, for every
// threshold from 1 to 192 blocks; fp16 operands: no gain.  -1 = the library's choice (exact fp32: always; other operand forms:
// never), 0 = never, n > 0 = from n blocks on in every operand form (benchmarking).
__device__ int32_t g_zero_idx[2] = {0, 0};

#ifdef SIGE_CONV_PROBE
static unsigned long long *g_probe_buf = nullptr;
static unsigned long long *conv_probe_buffer() {
    if (!g_probe_buf && hipMalloc(&g_probe_buf, 8 * 4096 * sizeof(unsigned long long)) != hipSuccess) g_probe_buf = nullptr;
    return g_probe_buf;
}
// (measurement build only, not declared in include/sige_hip.h) copy the timestamps of the last conv launches to the host
extern "C" int sige_hip_conv_probe_read(unsigned long long *host, int workgroups) {
    if (!g_probe_buf || workgroups > 4096) return SIGE_HIP_EINVAL;
    return hipMemcpy(host, g_probe_buf, (size_t)workgroups * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess
               ? SIGE_HIP_OK : SIGE_HIP_ELAUNCH;
}
extern "C" int sige_hip_conv_probe_clear(void) {
    if (!conv_probe_buffer()) return SIGE_HIP_ELAUNCH;
    return hipMemset(g_probe_buf, 0, 8 * 4096 * sizeof(unsigned long long)) == hipSuccess ? SIGE_HIP_OK : SIGE_HIP_ELAUNCH;
}
#endif

// ---- horizontal fusion: a residual block's 1x1 shortcut held back and launched WITH the block's conv1 ----
// sige_hip_conv_pair_begin(): the next eligible 1x1 launch of this thread (channels-last gather -> conv, raw staging,
// fp32) is recorded instead of launched; the next eligible 3x3 launch (gather + affine + SiLU -> conv, same stream and
// destination kind) then runs both in one conv_pair_kernel launch.  Anything else -- another conv kind, or
// sige_hip_conv_pair_end() -- launches the held conv on its own first, so results never depend on pairing.
struct HeldConv {
    bool active = false;
    ConvArgs a;
    int mode = 0, dst = 0, prec = 0;
    hipStream_t st = nullptr;
    int (*launch)(ConvArgs, int, hipStream_t) = nullptr;
};
static thread_local HeldConv g_held;
static thread_local bool g_pairing = false;
static std::atomic<long> g_pairs_fused{0};

static int flush_held() {
    if (!g_held.active) return SIGE_HIP_OK;
    const HeldConv h = g_held;
    g_held.active = false;
    const bool p = g_pairing;
    g_pairing = false;
    const int rc = h.launch(h.a, h.mode, h.st);
    note_launches(1);
    g_pairing = p;
    return rc;
}

template <typename GA, int NBA, typename GB, int DST, int W>
void launch_conv_pair(ConvArgs a, ConvArgs b, int mode_a, hipStream_t st);
#define SIGE_PAIR_DECLARE(DST, W)                                                                        \
    template <> void launch_conv_pair<K31_16, 1, K11_16, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_16, 1, K11_32, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_16, 2, K11_16, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_16, 2, K11_32, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_32, 1, K11_16, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_32, 1, K11_32, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_32, 2, K11_16, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<K31_32, 2, K11_32, DST, W>(ConvArgs, ConvArgs, int, hipStream_t);
SIGE_PAIR_DECLARE(DST_TILES, 4) SIGE_PAIR_DECLARE(DST_TILES, 8) SIGE_PAIR_DECLARE(DST_NCHW, 4) SIGE_PAIR_DECLARE(DST_NCHW, 8)
#define SIGE_PAIR_DECLARE_H(DST)                                                                         \
    template <> void launch_conv_pair<H31_16, 1, H11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_16, 1, H11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_16, 2, H11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_16, 2, H11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_32, 1, H11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_32, 1, H11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_32, 2, H11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<H31_32, 2, H11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);
SIGE_PAIR_DECLARE_H(DST_TILES) SIGE_PAIR_DECLARE_H(DST_NCHW)
#define SIGE_PAIR_DECLARE_X(DST)                                                                         \
    template <> void launch_conv_pair<X31_16, 1, X11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<X31_16, 1, X11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<X31_32, 1, X11_16, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);       \
    template <> void launch_conv_pair<X31_32, 1, X11_32, DST, 4>(ConvArgs, ConvArgs, int, hipStream_t);
SIGE_PAIR_DECLARE_X(DST_TILES) SIGE_PAIR_DECLARE_X(DST_NCHW)

// Tickets of the in-kernel K-split finish (conv_mfma.hpp): one int per output block of a split launch, zero whenever no
// such launch is running.  One buffer per device: a ring for eager launches (a slice is only live while its launch runs;
// 2^20 tickets = hundreds of launches in flight before a wrap could meet a running one) and a bump-allocated region for
// launches recorded into a hipGraph, whose slice is baked into the graph and must never be handed out again.  nullptr
// (first use during a capture, region exhausted, allocation failure): the launch falls back to the second pass.
constexpr size_t kTicketRing = size_t(1) << 20, kTicketGraph = size_t(3) << 20;
struct TicketPool { int32_t *buf = nullptr; size_t ring_pos = 0, graph_pos = 0; bool failed = false; };
static TicketPool g_tickets[32];
static std::mutex g_tickets_mu;

int32_t *split_tickets(hipStream_t st, long blocks) {
    int dev = -1;
    if (tuning(SIGE_HIP_TUNE_CONV_KSPLIT_SECOND_PASS) || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32 || blocks > (long)kTicketRing) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;
    const bool capturing = cs != hipStreamCaptureStatusNone;
    std::lock_guard<std::mutex> lock(g_tickets_mu);
    TicketPool &t = g_tickets[dev];
    if (!t.buf) {
        if (capturing || t.failed) return nullptr;
        const size_t bytes = (kTicketRing + kTicketGraph) * sizeof(int32_t);
        if (hipMalloc(&t.buf, bytes) != hipSuccess || hipMemset(t.buf, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            t.buf = nullptr; t.failed = true;
            (void)hipGetLastError();
            return nullptr;
        }
    }
    if (capturing) {
        if (t.graph_pos + blocks > kTicketGraph) return nullptr;
        int32_t *p = t.buf + kTicketRing + t.graph_pos;
        t.graph_pos += blocks;
        return p;
    }
    if (t.ring_pos + blocks > kTicketRing) t.ring_pos = 0;
    int32_t *p = t.buf + t.ring_pos;
    t.ring_pos += blocks;
    return p;
}

int release_graph_tickets() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return SIGE_HIP_EINVAL;
    std::lock_guard<std::mutex> lock(g_tickets_mu);
    g_tickets[dev].graph_pos = 0;
    return SIGE_HIP_OK;
}

struct ConvPlan { int mt, nb, waves; };

// PREC: 0 exact fp32 (ConvGeo) | 1 fp16 operands (ConvGeoH) | 2 split fp16 operands (ConvGeoX)
template <int PREC, int KH, int STR, int R, int MT>
using GeoOf = std::conditional_t<PREC == 2, ConvGeoX<KH, STR, R, MT>, std::conditional_t<PREC == 1, ConvGeoH<KH, STR, R, MT>, ConvGeo<KH, STR, R, MT>>>;

// Everything a launch decides on the host: output block, waves, grid order, K split.  `want_waves` != 0 / `nb1`: the
// constraints of the second conv of a pair (same workgroup size as the first, NB = 1, no K split: cap = 1).
template <int KH, int STR, int R, int SRC, int LAY, int PREC>
static int plan_conv(ConvArgs &a, int cap, int want_waves, bool nb1, ConvPlan &p) {
    using G32 = GeoOf<PREC, KH, STR, R, 32>;
    using G16 = GeoOf<PREC, KH, STR, R, 16>;
    const bool kHasNB2 = STR == 1 && !nb1 && PREC != 2;  // (split operands: the (hi, lo) weight registers of two sub-blocks do not fit)
    auto blocks = [&](int tpb, int mt, int nb) { return (long)ceil_div(a.T, tpb) * ceil_div(a.Cout, mt * nb); };
    // constraints of the staging path (conv_mfma.hpp): a fused torch.cat must split on a chunk
    // boundary; a per-batch affine needs every M block inside one batch
    auto usable = [&](int mt) {
        const int cc = (mt == 32 ? G32::CC : G16::CC) * (want_waves == 8 ? 2 : 1), tpb = mt == 32 ? G32::TPB : G16::TPB;
        if (SRC == SRC_GATHER && a.Csplit != a.Cin && a.Csplit % cc) return false;
        if (SRC != SRC_TILES && a.aff_sb != 0 && a.B > 1 && a.N % tpb) return false;
        return true;
    };
    const long kFill = want_waves ? 64 : 224;  // (a pair's second conv shares the chip with the first)
    int mt = 0, nb = 1;
    const int force_mt = tuning(SIGE_HIP_TUNE_CONV_TILE_MT), force_nb = tuning(SIGE_HIP_TUNE_CONV_TILE_NB), force_waves = tuning(SIGE_HIP_TUNE_CONV_WAVES);
    const int large_grid_nb1 = tuning(SIGE_HIP_TUNE_CONV_LARGE_GRID_NB1), force_ksplit = tuning(SIGE_HIP_TUNE_CONV_KSPLIT);
    if (force_mt && !want_waves) { mt = force_mt == 32 ? 32 : 16; nb = (force_nb == 2 && kHasNB2) ? 2 : 1; if (!usable(mt)) mt = 0; }
    // Exact fp32, channels-last, stride 1: 128 unsplit 16-channel blocks beat 32 blocks x 4 K splits -- the 8x8 layers of the
    // DDPM U-Net, 13.4 vs 14.4 us per launch, 1.437 -> 1.419 ms per forward (tools/probe/ksplit_forward_probe.py, round 3): the
    // split's second phase (partial sums out, ticket, the last workgroup's sum over the copies) costs more than the 4x shorter
    // K loop saves.  Below half a chip of blocks the split still wins.
    constexpr bool kW8Geo = LAY == LAYOUT_NHWC && STR == 1 && PREC == 0;
    const long kSplitBelow = (kW8Geo && !want_waves) ? 112 : kFill;
    const bool stay_unsplit = cap > 1 && usable(16) && blocks(G16::TPB, 16, 1) >= kSplitBelow && blocks(G16::TPB, 16, 1) < kFill && !force_ksplit;
    if (!mt && cap > 1 && usable(16) && blocks(G16::TPB, 16, 1) < kSplitBelow) {
        // too few tiles for any block shape: split K across workgroups, largest block that then fills the chip
        const int nc32 = ceil_div(a.Cin, G32::CC), nc16 = ceil_div(a.Cin, G16::CC);
        auto filled = [&](int tpb, int m, int n, int nc) { long b = blocks(tpb, m, n); return b * ksplit_for(b, nc, cap); };
        if (usable(32) && kHasNB2 && filled(G32::TPB, 32, 2, nc32) >= kFill) { mt = 32; nb = 2; }
        else if (usable(32) && filled(G32::TPB, 32, 1, nc32) >= kFill) { mt = 32; nb = 1; }
        else if (kHasNB2 && filled(G16::TPB, 16, 2, nc16) >= kFill) { mt = 16; nb = 2; }
        else { mt = 16; nb = 1; }
    }
    if (!mt) {
        const bool many = large_grid_nb1 < 0 ? PREC == 0 : (large_grid_nb1 > 0 && blocks(G32::TPB, 32, 2) >= large_grid_nb1);
        if (usable(32) && kHasNB2 && !many && blocks(G32::TPB, 32, 2) >= kFill) { mt = 32; nb = 2; }
        else if (usable(32) && blocks(G32::TPB, 32, 1) >= kFill) { mt = 32; nb = 1; }
        else if (usable(16) && kHasNB2 && blocks(G16::TPB, 16, 2) >= kFill) { mt = 16; nb = 2; }
        else if (usable(16)) { mt = 16; nb = 1; }
        else if (usable(32)) { mt = 32; nb = 1; }
        else return SIGE_HIP_EUNSUPPORTED;
    }
    const int tpb = mt == 32 ? G32::TPB : G16::TPB;
    a.mbk = ceil_div(a.T, tpb);
    a.ngk = ceil_div(a.Cout, mt * nb);
    const int cc4 = mt == 32 ? G32::CC : G16::CC;
    const int nchunks4 = (ceil_div(a.Cin, cc4) + 1) & ~1;  // as packed (padded to even)
    a.nblk = nchunks4 * 4;
    // 8-wave workgroups (two waves per SIMD) when the grid cannot give every CU two 4-wave workgroups
    constexpr bool kHasW8 = LAY == LAYOUT_NHWC && STR == 1 && PREC == 0;
    int waves = 4;
    const bool w8_ok = kHasW8 && !(SRC == SRC_GATHER && a.Csplit != a.Cin && a.Csplit % (2 * cc4)) && a.Cin > cc4;
    if (want_waves) {
        if (want_waves == 8 && !w8_ok) return SIGE_HIP_EUNSUPPORTED;
        waves = want_waves;
    } else if (force_waves == 8 && w8_ok) {
        // (round 2 picked 8 waves for grids below 160 blocks; with today's kernels and the 8x8 layers unsplit the 4-wave form wins
        //  in the forward -- 1.398 vs 1.417 ms, tools/probe/plan_forward_probe.py -- so 8 waves are the benchmarking knob's only)
        waves = 8;
    }
    a.nchunks = waves == 8 ? nchunks4 / 2 : ceil_div(a.Cin, cc4);
    // which operand should stay XCD-local: weights (dense layers) or input tiles (many active tiles)
    const double wbytes = (double)a.Cout * a.Cin * KH * KH, abytes = (double)a.T * a.Cin * R * R;
    // (0: M blocks fastest -- small launches; 1: weights dominate; 2: activations dominate and there are enough M blocks
    //  to give every XCD whole ones)
    a.ng_fast = wbytes > abytes ? 1 : ((a.mbk >= 16 && a.ngk > 1) ? 2 : 0);
    if (mt == 16)  // the MT=16 layout follows the MT=32 one
        a.packed += PREC == 2 ? packed_units_x(a.Cout, a.Cin, KH * KH, 32)
                              : (PREC == 1 ? packed_units_h(a.Cout, a.Cin, KH * KH, 32) : packed_floats(a.Cout, a.Cin, KH * KH, 32));
    // K split (channels-last launches that came with a workspace)
    a.ksplit = stay_unsplit ? 1 : ksplit_for((long)a.mbk * a.ngk, a.nchunks, cap);
    a.chunks_per_split = ceil_div(a.nchunks, a.ksplit);
    a.ksplit = ceil_div(a.nchunks, a.chunks_per_split);
#ifdef SIGE_CONV_PROBE
    a.probe = conv_probe_buffer();
#endif
    p.mt = mt; p.nb = nb; p.waves = waves;
    return SIGE_HIP_OK;
}

// conv A (planned: pa) + the held 1x1 conv in one launch; false: no pair kernel for this combination
template <int DST, int PREC>
static bool launch_pair(const ConvArgs &a, const ConvPlan &pa, int mode_a, ConvArgs b, hipStream_t st) {
    using A16 = GeoOf<PREC, 3, 1, 6, 16>;
    using A32 = GeoOf<PREC, 3, 1, 6, 32>;
    using B16 = GeoOf<PREC, 1, 1, 4, 16>;
    using B32 = GeoOf<PREC, 1, 1, 4, 32>;
    ConvPlan pb;
    if (plan_conv<1, 1, 4, SRC_GATHER, LAYOUT_NHWC, PREC>(b, 1, pa.waves, true, pb) != SIGE_HIP_OK) return false;
#define SIGE_PAIR_GO(GA, NBA, W)                                                                         \
    do {                                                                                                 \
        if (pb.mt == 32) launch_conv_pair<GA, NBA, B32, DST, W>(a, b, mode_a, st);                       \
        else launch_conv_pair<GA, NBA, B16, DST, W>(a, b, mode_a, st);                                   \
    } while (0)
#define SIGE_PAIR_W(W)                                                                                   \
    do {                                                                                                 \
        if (pa.mt == 32 && pa.nb == 2) SIGE_PAIR_GO(A32, 2, W);                                          \
        else if (pa.mt == 32) SIGE_PAIR_GO(A32, 1, W);                                                   \
        else if (pa.nb == 2) SIGE_PAIR_GO(A16, 2, W);                                                    \
        else SIGE_PAIR_GO(A16, 1, W);                                                                    \
    } while (0)
    if constexpr (PREC == 0) {
        if (pa.waves == 8) SIGE_PAIR_W(8); else SIGE_PAIR_W(4);
    } else if constexpr (PREC == 1) {
        if (pa.waves != 4) return false;
        SIGE_PAIR_W(4);
    } else {
        if (pa.waves != 4 || pa.nb != 1) return false;
        if (pa.mt == 32) SIGE_PAIR_GO(A32, 1, 4); else SIGE_PAIR_GO(A16, 1, 4);
    }
#undef SIGE_PAIR_W
#undef SIGE_PAIR_GO
    g_pairs_fused.fetch_add(1, std::memory_order_relaxed);
    return true;
}

// ---- the same pairing for a conv1 routed to the dense-layer kernel (conv_wide.hip asks here) ----
// arithmetic of the 1x1 conv held on `st` for a full-tensor destination (0 exact fp32 | 1 fp16 operands | 2 split operands), or
// -1 if nothing is held that a dense-layer launch on `st` could take along
int held_shortcut_prec(hipStream_t st) {
    if (!g_held.active || g_held.st != st || g_held.dst != DST_NCHW || g_held.mode != MODE_RAW) return -1;
    return g_held.prec;
}

// plan the held 1x1 for a 4-wave partner (NB = 1, no K split) and hand it over: *b = its launch arguments, *mt = its output
// block (16 | 32).  false: no plan for this shape -- the caller flushes it instead.
bool take_held_shortcut(ConvArgs *b, int *mt) {
    if (!g_held.active) return false;
    ConvArgs a = g_held.a;
    ConvPlan p;
    int rc;
    if (g_held.prec == 0) rc = plan_conv<1, 1, 4, SRC_GATHER, LAYOUT_NHWC, 0>(a, 1, 4, true, p);
    else if (g_held.prec == 1) rc = plan_conv<1, 1, 4, SRC_GATHER, LAYOUT_NHWC, 1>(a, 1, 4, true, p);
    else rc = plan_conv<1, 1, 4, SRC_GATHER, LAYOUT_NHWC, 2>(a, 1, 4, true, p);
    if (rc != SIGE_HIP_OK || p.waves != 4 || p.nb != 1) return false;
    *b = a;
    *mt = p.mt;
    g_held.active = false;
    g_pairs_fused.fetch_add(1, std::memory_order_relaxed);
    return true;
}

int flush_held_conv() { return flush_held(); }

template <int KH, int STR, int R, int SRC, int DST, int LAY, int PREC = 0>
static int launch_kind(ConvArgs a, int mode, hipStream_t st) {
    using G32 = GeoOf<PREC, KH, STR, R, 32>;
    using G16 = GeoOf<PREC, KH, STR, R, 16>;
    constexpr bool kHasNB2 = STR == 1 && PREC != 2;
    if constexpr (PREC == 2)  // 2^-S sits in the header behind the two packed tile shapes (pack_weights_x_kernel)
        a.wscale_ptr = a.packed + packed_units_x(a.Cout, a.Cin, KH * KH, 32) + packed_units_x(a.Cout, a.Cin, KH * KH, 16) + 1;
    constexpr bool kPairLayout = SRC == SRC_GATHER && LAY == LAYOUT_NHWC && STR == 1;
    constexpr bool kPairFirst = kPairLayout && KH == 3, kPairSecond = kPairLayout && KH == 1;
    const bool may_pair = kPairFirst && g_held.active && (mode == MODE_AFFINE_SWISH || mode == MODE_RAW) && g_held.st == st && g_held.dst == DST && g_held.prec == PREC;
    if (g_held.active && !may_pair) {
        const int rc = flush_held();
        if (rc != SIGE_HIP_OK) return rc;
    }
    if (kPairSecond && g_pairing && !g_held.active && mode == MODE_RAW && !tuning(SIGE_HIP_TUNE_CONV_TILE_MT)) {
        g_held.active = true; g_held.a = a; g_held.mode = mode; g_held.dst = DST; g_held.st = st; g_held.prec = PREC;
        g_held.launch = &launch_kind<KH, STR, R, SRC, DST, LAY, PREC>;
        note_launches(-1);  // (the caller counts one launch per call: this one happens later, or inside its partner's)
        return SIGE_HIP_OK;
    }
    const int cap = (LAY == LAYOUT_NHWC && a.ws) ? a.ksplit_max : 1;
    ConvPlan p;
    const ConvArgs a0 = a;  // (as handed in: plan_conv rewrites the packed-weight pointer and the grid fields)
    const int prc = plan_conv<KH, STR, R, SRC, LAY, PREC>(a, cap, 0, false, p);
    if (prc != SIGE_HIP_OK) {
        const int rc = flush_held();
        return rc != SIGE_HIP_OK ? rc : prc;
    }
    float *final_out = a.out;
    if (a.ksplit > 1) {
        a.counters = split_tickets(st, (long)a.mbk * a.ngk);
        if (!a.counters && (a.twin0 || a.twin1)) {
            // (the second-pass kernel does not write twins: without tickets such a launch runs unsplit)
            a = a0;
            const int rc1 = plan_conv<KH, STR, R, SRC, LAY, PREC>(a, 1, 0, false, p);
            if (rc1 != SIGE_HIP_OK) {
                // no unsplit plan either: a split launch finished by the second pass would leave the twins unwritten while the
                // caller reports them as written -- refuse, the caller falls back and its consumers activate for themselves
                const int rc = flush_held();
                return rc != SIGE_HIP_OK ? rc : SIGE_HIP_EUNSUPPORTED;
            }
        }
    }
    if (a.ksplit > 1) {
        a.out = a.ws;
        a.fout = final_out;
    }
    const int mt = p.mt, nb = p.nb, waves = p.waves;
    bool done = false;
    if constexpr (kPairFirst) {
        if (may_pair) {
            done = launch_pair<DST, PREC>(a, p, mode, g_held.a, st);
            if (done) g_held.active = false;
            else {
                const int rc = flush_held();
                if (rc != SIGE_HIP_OK) return rc;
            }
        }
    }
    if constexpr (SRC == SRC_SCATTER_GATHER && LAY == LAYOUT_NHWC && KH == 3 && STR == 1) {
        if (!done && a.y_f16) {  // the cached tensor is stored as fp16: the Y16 form of the kernel (4 waves)
            if (waves != 4) return SIGE_HIP_EUNSUPPORTED;
            if constexpr (kHasNB2) {
                if (mt == 32 && nb == 2) { launch_conv_c16<G32, 2, DST>(a, mode, st); done = true; }
                else if (nb == 2) { launch_conv_c16<G16, 2, DST>(a, mode, st); done = true; }
            }
            if (!done) {
                if (mt == 32) launch_conv_c16<G32, 1, DST>(a, mode, st);
                else launch_conv_c16<G16, 1, DST>(a, mode, st);
                done = true;
            }
        }
    } else {
        if (a.y_f16) return SIGE_HIP_EUNSUPPORTED;
    }
    constexpr bool kHasW8 = LAY == LAYOUT_NHWC && STR == 1 && PREC == 0;
    if constexpr (kHasW8) {
        if (!done && waves == 8) {
            if (mt == 32 && nb == 2) launch_conv_geo<G32, 2, SRC, DST, LAY, 8>(a, mode, st);
            else if (mt == 32) launch_conv_geo<G32, 1, SRC, DST, LAY, 8>(a, mode, st);
            else if (nb == 2) launch_conv_geo<G16, 2, SRC, DST, LAY, 8>(a, mode, st);
            else launch_conv_geo<G16, 1, SRC, DST, LAY, 8>(a, mode, st);
            done = true;
        }
    }
    if constexpr (kHasNB2) {
        if (!done && mt == 32 && nb == 2) { launch_conv_geo<G32, 2, SRC, DST, LAY, 4>(a, mode, st); done = true; }
        else if (!done && nb == 2) { launch_conv_geo<G16, 2, SRC, DST, LAY, 4>(a, mode, st); done = true; }
    }
    if (!done) {
        if (mt == 32) launch_conv_geo<G32, 1, SRC, DST, LAY, 4>(a, mode, st);
        else launch_conv_geo<G16, 1, SRC, DST, LAY, 4>(a, mode, st);
    }
    if (a.ksplit > 1 && !a.counters) {
        const size_t n4 = a.split_stride / 4;
        const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        splitk_reduce_nhwc_kernel<<<grid, 256, 0, st>>>(a.ws, a.ksplit, a.split_stride, n4, a.Cout, a.bias, a.residual,
                                                       a.oscale, a.oshift, a.oact, final_out);
        note_launches(1);
    }
    return SIGE_HIP_OK;
}

template <int SRC, int DST = DST_TILES, int LAY = LAYOUT_NCHW, int PREC = 0>
static int launch_conv(const ConvArgs &a, int mode, int kH, int kW, int R, int S, int strH, int strW, hipStream_t st) {
    int rc;
    switch (mfma_kind(kH, kW, R, S, strH, strW, 1)) {
        case 1: rc = launch_kind<3, 1, 6, SRC, DST, LAY, PREC>(a, mode, st); break;
        case 2: rc = launch_kind<1, 1, 4, SRC, DST, LAY, PREC>(a, mode, st); break;
        case 3:
            if constexpr (PREC != 0) return SIGE_HIP_EUNSUPPORTED;  // (see the f16-compute declarations above)
            else rc = launch_kind<3, 2, 5, SRC, DST, LAY, PREC>(a, mode, st);
            break;
        default: return SIGE_HIP_EUNSUPPORTED;
    }
    return rc != SIGE_HIP_OK ? rc : launch_status();
}

// (scale, shift, activation) -> staging mode of the fused kernels; -1: not expressible
// (the caller then uses gather / scatter_gather + block_conv).  scale and shift must be
// both present with one broadcast shape [1|B, 1|C, 1, 1], or both absent.
static int staging_mode(const float *scale, int sB, int sC, const float *shift, int tB, int tC, int activation,
                        int B, int C, int *aff_sb, int *aff_sc) {
    *aff_sb = *aff_sc = 0;
    if (!scale && !shift) return activation == SIGE_HIP_ACT_IDENTITY ? MODE_RAW : -1;
    if (!scale || !shift || sB != tB || sC != tC) return -1;
    if (!((sB == 1 || sB == B) && (sC == 1 || sC == C))) return -1;
    *aff_sb = sB > 1 ? sC : 0;
    *aff_sc = sC > 1 ? 1 : 0;
    return activation == SIGE_HIP_ACT_SWISH ? MODE_AFFINE_SWISH : MODE_AFFINE;
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_conv_pair_begin(void) {
    SIGE_PLAN_HOOK0(sige_hip_conv_pair_begin);
    const int rc = flush_held();
    g_pairing = true;
    return rc;
}

extern "C" int sige_hip_conv_pair_end(void) {
    SIGE_PLAN_HOOK0(sige_hip_conv_pair_end);
    g_pairing = false;
    const int rc = flush_held();
    return rc != SIGE_HIP_OK ? rc : launch_status(0);
}

extern "C" int64_t sige_hip_conv_pairs_fused(void) { return (int64_t)g_pairs_fused.load(std::memory_order_relaxed); }

extern "C" size_t sige_hip_block_conv_packed_size(int Cout, int Cin, int kH, int kW, int R, int S,
                                                  int strideH, int strideW, int groups) {
    if (Cout <= 0 || Cin <= 0) return 0;
    if (!mfma_kind(kH, kW, R, S, strideH, strideW, groups)) return 0;
    return packed_floats(Cout, Cin, kH * kW, 32) + packed_floats(Cout, Cin, kH * kW, 16);
}

extern "C" int sige_hip_block_conv_pack_f32(const float *w, int Cout, int Cin, int kH, int kW,
                                            float *packed, void *stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return SIGE_HIP_EINVAL;
    if (kH != kW || (kH != 1 && kH != 3)) return SIGE_HIP_EUNSUPPORTED;
    const int KK = kH * kW;
    hipStream_t st = as_stream(stream);
    const long n32 = (long)packed_floats(Cout, Cin, KK, 32), n16 = (long)packed_floats(Cout, Cin, KK, 16);
    auto blocks = [](long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); };
    if (KK == 9) {
        pack_weights_kernel<9, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, packed, n32);
        pack_weights_kernel<9, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, packed + n32, n16);
    } else {
        pack_weights_kernel<1, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, packed, n32);
        pack_weights_kernel<1, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, packed + n32, n16);
    }
    return launch_status(2);
}

extern "C" size_t sige_hip_block_conv_packed_size_f16c(int Cout, int Cin, int kH, int kW, int R, int S,
                                                       int strideH, int strideW, int groups) {
    if (Cout <= 0 || Cin <= 0) return 0;
    const int kind = mfma_kind(kH, kW, R, S, strideH, strideW, groups);
    if (kind != 1 && kind != 2) return 0;  // stride-1 3x3 on 6x6 and 1x1 on 4x4 (the stride-2 geometry stays fp32)
    return packed_units_h(Cout, Cin, kH * kW, 32) + packed_units_h(Cout, Cin, kH * kW, 16);
}

extern "C" int sige_hip_block_conv_pack_f16c(const float *w, int Cout, int Cin, int kH, int kW,
                                             float *packed, void *stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return SIGE_HIP_EINVAL;
    if (kH != kW || (kH != 1 && kH != 3)) return SIGE_HIP_EUNSUPPORTED;
    const int KK = kH * kW;
    hipStream_t st = as_stream(stream);
    const long n32 = 2 * (long)packed_units_h(Cout, Cin, KK, 32), n16 = 2 * (long)packed_units_h(Cout, Cin, KK, 16);  // halves
    auto blocks = [](long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); };
    _Float16 *ph = reinterpret_cast<_Float16 *>(packed);
    if (KK == 9) {
        pack_weights_h_kernel<9, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, ph, n32);
        pack_weights_h_kernel<9, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, ph + n32, n16);
    } else {
        pack_weights_h_kernel<1, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, ph, n32);
        pack_weights_h_kernel<1, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, ph + n32, n16);
    }
    return launch_status(2);
}

extern "C" size_t sige_hip_block_conv_packed_size_f16x3(int Cout, int Cin, int kH, int kW, int R, int S,
                                                        int strideH, int strideW, int groups) {
    if (Cout <= 0 || Cin <= 0) return 0;
    const int kind = mfma_kind(kH, kW, R, S, strideH, strideW, groups);
    if (kind != 1 && kind != 2) return 0;  // stride-1 3x3 on 6x6 and 1x1 on 4x4 (the stride-2 geometry stays fp32)
    return packed_units_x(Cout, Cin, kH * kW, 32) + packed_units_x(Cout, Cin, kH * kW, 16) + kX3Header;
}

extern "C" int sige_hip_block_conv_pack_f16x3(const float *w, int Cout, int Cin, int kH, int kW,
                                              float *packed, void *stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return SIGE_HIP_EINVAL;
    if (kH != kW || (kH != 1 && kH != 3)) return SIGE_HIP_EUNSUPPORTED;
    const int KK = kH * kW;
    hipStream_t st = as_stream(stream);
    const size_t u32 = packed_units_x(Cout, Cin, KK, 32), u16 = packed_units_x(Cout, Cin, KK, 16);
    const long n32 = 2 * (long)u32, n16 = 2 * (long)u16;  // halves
    float *header = packed + u32 + u16;
    if (hipMemsetAsync(header, 0, kX3Header * sizeof(float), st) != hipSuccess) return SIGE_HIP_ELAUNCH;
    const long nw = (long)Cout * Cin * KK;
    wmax_bits_kernel<<<(int)((nw + 255) / 256 < 1024 ? (nw + 255) / 256 : 1024), 256, 0, st>>>(w, nw, reinterpret_cast<unsigned *>(header));
    auto blocks = [](long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); };
    _Float16 *ph = reinterpret_cast<_Float16 *>(packed);
    if (KK == 9) {
        pack_weights_x_kernel<9, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, ph, n32, header);
        pack_weights_x_kernel<9, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, ph + n32, n16, header);
    } else {
        pack_weights_x_kernel<1, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, ph, n32, header);
        pack_weights_x_kernel<1, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, ph + n32, n16, header);
    }
    return launch_status(3);
}

extern "C" int sige_hip_block_conv_f32(const float *x, int T, int Cin, int R, int S,
                                       const float *packed, const float *bias, int Cout, int kH, int kW,
                                       int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_block_conv_f32, x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (T < 0 || Cin <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    if ((long)T * Cin * R * S >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;  // 32-bit byte offsets in the kernel
    ConvArgs a{};
    a.x = x; a.packed = packed; a.bias = bias; a.out = out;
    a.T = T; a.Cin = Cin; a.Cout = Cout;
    if (reinterpret_cast<uintptr_t>(x) & 15 || ((long)Cin * R * S) % 4) {
        // The tile slab cannot be staged with 16-byte loads: run it as a gather of ONE tile at
        // (0, 0) from each of T "images" [Cin, R, S] (element-wise staging, same MFMA order).
        void *zero_idx = nullptr;
        if (hipGetSymbolAddress(&zero_idx, HIP_SYMBOL(g_zero_idx)) != hipSuccess) return SIGE_HIP_ELAUNCH;
        a.x2 = x; a.Csplit = Cin; a.idx = static_cast<const int32_t *>(zero_idx);
        a.B = T; a.N = 1; a.H = R; a.W = S;
        return launch_conv<SRC_GATHER>(a, MODE_RAW, kH, kW, R, S, strideH, strideW, as_stream(stream));
    }
    return launch_conv<SRC_TILES>(a, 0, kH, kW, R, S, strideH, strideW, as_stream(stream));
}

static bool channel_affine(const float *p, int b, int c, int h, int w, int B, int C) {
    return !p || ((b == 1 || b == B) && (c == 1 || c == C) && h == 1 && w == 1);
}

extern "C" int sige_hip_gather_conv_f32(const float *x, int B, int Cin, int H, int W, int bH, int bW,
                                        const int32_t *active_indices, int N,
                                        const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                        const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                        int activation,
                                        const float *packed, const float *bias, int Cout, int kH, int kW,
                                        int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_gather_conv_f32, x, B, Cin, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, scaleH, scaleW, shift, shiftB, shiftC, shiftH, shiftW, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (!channel_affine(scale, scaleB, scaleC, scaleH, scaleW, B, Cin) ||
        !channel_affine(shift, shiftB, shiftC, shiftH, shiftW, B, Cin))
        return SIGE_HIP_EUNSUPPORTED;  // spatially varying affine: use gather + block_conv
    if ((long)B * Cin * H * W >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;  // 32-bit element offsets in the kernel
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out || !active_indices) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.idx = active_indices; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    if (stacked_shift(H) != 0) return SIGE_HIP_EUNSUPPORTED;  // (stacked edits: channels-last fused kernels only)
    a.Csplit = Cin;
    a.x2 = x;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    return launch_conv<SRC_GATHER>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_gather_conv_nchw_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW, int offsetH, int offsetW,
                                             const float *residual, int Ho, int Wo, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_gather_conv_nchw_f32, x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, offsetH, offsetW, residual, Ho, Wo, out, stream);
    const int Cin = C1 + C2;
    if (B < 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Ho <= 0 || Wo <= 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((scale && !((scaleB == 1 || scaleB == B) && (scaleC == 1 || scaleC == Cin))) ||
        (shift && !((shiftB == 1 || shiftB == B) && (shiftC == 1 || shiftC == Cin))))
        return SIGE_HIP_EINVAL;
    if ((long)B * Cin * H * W >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;  // 32-bit element offsets in the kernel
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || (C2 && !x2) || !packed || !out || !active_indices) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.x2 = x2; a.Csplit = C1; a.idx = active_indices; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    if (stacked_shift(H) != 0) return SIGE_HIP_EUNSUPPORTED;  // (stacked edits: channels-last fused kernels only)
    if (C2 && B != 1) return SIGE_HIP_EUNSUPPORTED;  // the two-tensor input is per image (conv_mfma.hpp)
    if (!x2) a.x2 = x;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    a.residual = residual; a.Ho = Ho; a.Wo = Wo; a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    return launch_conv<SRC_GATHER, DST_NCHW>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_scatter_gather_conv_f32(const float *x, const float *y, int B, int Cin, int H, int W,
                                                int Rx, int Sx, int bH, int bW,
                                                const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                                const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                                int activation,
                                                const float *packed, const float *bias, int Cout, int kH, int kW,
                                                int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_gather_conv_f32, x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, scaleH, scaleW, shift, shiftB, shiftC, shiftH, shiftW, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Rx <= 0 || Sx <= 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (!channel_affine(scale, scaleB, scaleC, scaleH, scaleW, B, Cin) ||
        !channel_affine(shift, shiftB, shiftC, shiftH, shiftW, B, Cin))
        return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * Cin * H * W >= (1L << 29) || (long)B * N * Cin * Rx * Sx >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !packed || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.y = y; a.idx = active_indices; a.map = scatter_map; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    if (stacked_shift(H) != 0) return SIGE_HIP_EUNSUPPORTED;  // (stacked edits: channels-last fused kernels only)
    a.RxSx = Rx * Sx; a.Sx = Sx;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    return launch_conv<SRC_SCATTER_GATHER>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

// ---- channels-last (NHWC) forms: the same kernels, tensors laid out [B,H,W,C] / [T,R,S,C] ----
static bool nhwc_ok(int Cin, int C1, int Cout, const void *p0, const void *p1, const void *p2, const void *p3) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return Cin % 4 == 0 && C1 % 4 == 0 && Cout % 4 == 0 && al(p0) && al(p1) && al(p2) && al(p3);
}

extern "C" int sige_hip_conv_ksplit_hint(int T, int Cin, int Cout, int kH, int kW, int strideH, int strideW) {
    if (T <= 0 || Cin <= 0 || Cout <= 0) return 1;
    const int px = (kH == 3 && strideH == 2) ? 4 : 16;  // output pixels per tile
    const long blocks16 = (long)ceil_div(T, 16 / px) * ceil_div(Cout, 16);
    if (blocks16 >= 224 && !tuning(SIGE_HIP_TUNE_CONV_KSPLIT)) return 1;
    return 8;  // (the launch decides the actual factor, at most 8)
}

template <int PREC>
static int block_conv_nhwc_impl(const float *x, int T, int Cin, int R, int S,
                                const float *packed, const float *bias, int Cout, int kH, int kW,
                                int strideH, int strideW, float *out, void *stream) {
    if (T < 0 || Cin <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out) return SIGE_HIP_EINVAL;
    if (!nhwc_ok(Cin, Cin, Cout, x, packed, out, bias)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)T * Cin * R * S >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;
    ConvArgs a{};
    a.x = x; a.packed = packed; a.bias = bias; a.out = out;
    a.T = T; a.Cin = Cin; a.Cout = Cout;
    return launch_conv<SRC_TILES, DST_TILES, LAYOUT_NHWC, PREC>(a, 0, kH, kW, R, S, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_block_conv_nhwc_f32(const float *x, int T, int Cin, int R, int S,
                                            const float *packed, const float *bias, int Cout, int kH, int kW,
                                            int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_block_conv_nhwc_f32, x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return block_conv_nhwc_impl<0>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
}

extern "C" int sige_hip_block_conv_nhwc_f16c(const float *x, int T, int Cin, int R, int S,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_block_conv_nhwc_f16c, x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return block_conv_nhwc_impl<1>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
}

extern "C" int sige_hip_block_conv_nhwc_f16x3(const float *x, int T, int Cin, int R, int S,
                                              const float *packed, const float *bias, int Cout, int kH, int kW,
                                              int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_block_conv_nhwc_f16x3, x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return block_conv_nhwc_impl<2>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
}

// ... over the tiles of an index list: T = B * N.  `count_key` is the index list the N tiles belong to; it is not read -- a launch
// plan looks N up under it (plan.hpp: CountOf), like the tile count of every gather-type entry point, so that a conv over a
// tile SLAB (GauGAN: the convs behind the SPADE modulation) follows a new mask too.  compute: 0 fp32 | 1 f16 operands | 2 split fp16.
extern "C" int sige_hip_block_conv_nhwc_keyed(int compute, const float *x, const int32_t *count_key, int B, int N, int Cin, int R, int S,
                                              const float *packed, const float *bias, int Cout, int kH, int kW,
                                              int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_block_conv_nhwc_keyed, (sige::CountOf<2, 4>), compute, x, count_key, B, N, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (B < 0 || N < 0 || !count_key || (long)B * N > 0x7fffffffL) return SIGE_HIP_EINVAL;
    const int T = B * N;
    if (compute == 0) return block_conv_nhwc_impl<0>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (compute == 1) return block_conv_nhwc_impl<1>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    if (compute == 2) return block_conv_nhwc_impl<2>(x, T, Cin, R, S, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return SIGE_HIP_EINVAL;
}

template <int PREC>
static int gather_conv_nhwc_impl(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW,
                                             int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                             float *workspace, size_t workspace_floats,
                                             const float *out_scale, const float *out_shift, int out_activation,
                                             int upsample2x,
                                             float *twin0, const float *twin0_scale, const float *twin0_shift,
                                             float *twin1, const float *twin1_scale, const float *twin1_shift,
                                             float *out, void *stream) {
    const int Cin = C1 + C2;
    if (upsample2x && ((H | W) & 1)) return SIGE_HIP_EINVAL;  // (H, W) = the upsampled size
    if (B < 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0) return SIGE_HIP_EINVAL;
    if (to_full && (Ho <= 0 || Wo <= 0)) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * Cin * H * W >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || (C2 && !x2) || !packed || !out || !active_indices) return SIGE_HIP_EINVAL;
    if (C2 && B != 1) return SIGE_HIP_EUNSUPPORTED;
    if (!nhwc_ok(Cin, C1, Cout, x, C2 ? x2 : x, out, bias) || (reinterpret_cast<uintptr_t>(packed) & 15) ||
        (reinterpret_cast<uintptr_t>(residual) & 15))
        return SIGE_HIP_EUNSUPPORTED;
    ConvArgs a{};
    a.x = x; a.x2 = C2 ? x2 : x; a.Csplit = C1; a.idx = active_indices; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.hp_shift = stacked_shift(H);  // (stacked edits: sige_hip_set_edit_batch)
    if (a.hp_shift < 0 || (a.hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    if ((out_scale == nullptr) != (out_shift == nullptr)) return SIGE_HIP_EINVAL;
    if (out_activation != SIGE_HIP_ACT_IDENTITY && out_activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(out_scale) | reinterpret_cast<uintptr_t>(out_shift)) & 15) return SIGE_HIP_EUNSUPPORTED;
    a.oscale = out_scale; a.oshift = out_shift; a.oact = out_activation;
    a.up = upsample2x ? 1 : 0;
    {   // optional K split: `workspace` holds whole copies of the output
        const int Ro = (bH - kH) / strideH + 1, So = (bW - kW) / strideW + 1;
        a.split_stride = to_full ? (size_t)B * Ho * Wo * Cout : (size_t)B * N * Ro * So * Cout;
        a.ws = workspace;
        a.ksplit_max = (workspace && a.split_stride && !(reinterpret_cast<uintptr_t>(workspace) & 15))
                           ? (int)(workspace_floats / a.split_stride < 8 ? workspace_floats / a.split_stride : 8) : 1;
        // (the second pass reads whole output copies: every pixel must be written by some tile)
        if (to_full && (long)N * Ro * So < (long)Ho * Wo) a.ksplit_max = 1;
    }
    if ((twin0 || twin1) && !to_full) return SIGE_HIP_EINVAL;  // twins share the addressing of a full-tensor destination
    if ((twin0 && (!twin0_scale || !twin0_shift)) || (twin1 && (!twin1_scale || !twin1_shift))) return SIGE_HIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(twin0) | reinterpret_cast<uintptr_t>(twin1) | reinterpret_cast<uintptr_t>(twin0_scale) |
         reinterpret_cast<uintptr_t>(twin0_shift) | reinterpret_cast<uintptr_t>(twin1_scale) | reinterpret_cast<uintptr_t>(twin1_shift)) & 15)
        return SIGE_HIP_EUNSUPPORTED;
    a.twin0 = twin0; a.tscale0 = twin0_scale; a.tshift0 = twin0_shift;
    a.twin1 = twin1; a.tscale1 = twin1_scale; a.tshift1 = twin1_shift;
    if (to_full) {
        a.residual = residual; a.Ho = Ho; a.Wo = Wo; a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
        return launch_conv<SRC_GATHER, DST_NCHW, LAYOUT_NHWC, PREC>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
    }
    return launch_conv<SRC_GATHER, DST_TILES, LAYOUT_NHWC, PREC>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

#define SIGE_GATHER_CONV_ARGS x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, \
    activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace,          \
    workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, \
    twin1_shift, out, stream
extern "C" int sige_hip_gather_conv_nhwc_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW,
                                             int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                             float *workspace, size_t workspace_floats,
                                             const float *out_scale, const float *out_shift, int out_activation,
                                             int upsample2x,
                                             float *twin0, const float *twin0_scale, const float *twin0_shift,
                                             float *twin1, const float *twin1_scale, const float *twin1_shift,
                                             float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_conv_nhwc_f32, (sige::CountOf<9, 10>), x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace, workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return gather_conv_nhwc_impl<0>(SIGE_GATHER_CONV_ARGS);
}
extern "C" int sige_hip_gather_conv_nhwc_f16c(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW,
                                             int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                             float *workspace, size_t workspace_floats,
                                             const float *out_scale, const float *out_shift, int out_activation,
                                             int upsample2x,
                                             float *twin0, const float *twin0_scale, const float *twin0_shift,
                                             float *twin1, const float *twin1_scale, const float *twin1_shift,
                                             float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_conv_nhwc_f16c, (sige::CountOf<9, 10>), x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace, workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return gather_conv_nhwc_impl<1>(SIGE_GATHER_CONV_ARGS);
}
extern "C" int sige_hip_gather_conv_nhwc_f16x3(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW,
                                             int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                             float *workspace, size_t workspace_floats,
                                             const float *out_scale, const float *out_shift, int out_activation,
                                             int upsample2x,
                                             float *twin0, const float *twin0_scale, const float *twin0_shift,
                                             float *twin1, const float *twin1_scale, const float *twin1_shift,
                                             float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_conv_nhwc_f16x3, (sige::CountOf<9, 10>), x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace, workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return gather_conv_nhwc_impl<2>(SIGE_GATHER_CONV_ARGS);
}

template <int PREC>
static int scatter_gather_conv_nhwc_impl(const float *x, const float *y, int B, int Cin, int H, int W,
                                                     int Rx, int Sx, int bH, int bW,
                                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                     const float *scale, int scaleB, int scaleC,
                                                     const float *shift, int shiftB, int shiftC,
                                                     int activation,
                                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                                     int strideH, int strideW, float *out, void *stream, int y_f16 = 0) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Rx <= 0 || Sx <= 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * Cin * H * W >= (1L << 29) || (long)B * N * Cin * Rx * Sx >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !packed || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    if (!nhwc_ok(Cin, Cin, Cout, x, y, out, bias) || (reinterpret_cast<uintptr_t>(packed) & 15)) return SIGE_HIP_EUNSUPPORTED;
    if (y_f16 && (kH != 3 || kW != 3 || strideH != 1 || strideW != 1)) return SIGE_HIP_EUNSUPPORTED;
    ConvArgs a{};
    a.y_f16 = y_f16 ? 1 : 0;
    a.x = x; a.y = y; a.idx = active_indices; a.map = scatter_map; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.hp_shift = stacked_shift(H);  // (stacked edits: sige_hip_set_edit_batch)
    if (a.hp_shift < 0 || (a.hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    a.RxSx = Rx * Sx; a.Sx = Sx;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    return launch_conv<SRC_SCATTER_GATHER, DST_TILES, LAYOUT_NHWC, PREC>(a, mode, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

#define SIGE_SG_CONV_ARGS x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, \
    shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream
extern "C" int sige_hip_scatter_gather_conv_nhwc_f32(const float *x, const float *y, int B, int Cin, int H, int W,
                                                     int Rx, int Sx, int bH, int bW,
                                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                     const float *scale, int scaleB, int scaleC,
                                                     const float *shift, int shiftB, int shiftC,
                                                     int activation,
                                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                                     int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_nhwc_f32, (sige::CountOf<10, 11>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return scatter_gather_conv_nhwc_impl<0>(SIGE_SG_CONV_ARGS);
}
extern "C" int sige_hip_scatter_gather_conv_nhwc_f16c(const float *x, const float *y, int B, int Cin, int H, int W,
                                                     int Rx, int Sx, int bH, int bW,
                                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                     const float *scale, int scaleB, int scaleC,
                                                     const float *shift, int shiftB, int shiftC,
                                                     int activation,
                                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                                     int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_nhwc_f16c, (sige::CountOf<10, 11>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return scatter_gather_conv_nhwc_impl<1>(SIGE_SG_CONV_ARGS);
}
extern "C" int sige_hip_scatter_gather_conv_nhwc_f16x3(const float *x, const float *y, int B, int Cin, int H, int W,
                                                     int Rx, int Sx, int bH, int bW,
                                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                     const float *scale, int scaleB, int scaleC,
                                                     const float *shift, int shiftB, int shiftC,
                                                     int activation,
                                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                                     int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_nhwc_f16x3, (sige::CountOf<10, 11>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    return scatter_gather_conv_nhwc_impl<2>(SIGE_SG_CONV_ARGS);
}

// scatter_gather -> conv -> Scatter / ScatterWithBlockResidual in ONE launch: the conv's output tiles go straight
// into `out` [B,H,W,Cout], a buffer that already holds the cached tensor outside this mask's tiles (in-place scatter).
template <int PREC>
static int scatter_gather_conv_scatter_nhwc_impl(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream, int y_f16 = 0, int res_f16 = 0) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Rx <= 0 || Sx <= 0) return SIGE_HIP_EINVAL;
    if (kH != 3 || kW != 3 || bH != 6 || bW != 6) return SIGE_HIP_EUNSUPPORTED;  // the stride-1 3x3 geometry of a ResBlock's conv2
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * Cin * H * W >= (1L << 29) || (long)B * N * Cin * Rx * Sx >= (1L << 29)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !packed || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    if (x1 && (!residual || !table1 || R1 <= 0 || S1 <= 0 || gH1 < (H + R1 - 1) / R1 || gW1 < (W + S1 - 1) / S1)) return SIGE_HIP_EINVAL;
    if (!nhwc_ok(Cin, Cin, Cout, x, y, out, bias) || (reinterpret_cast<uintptr_t>(packed) & 15) ||
        (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(x1) & 15))
        return SIGE_HIP_EUNSUPPORTED;
    if (res_f16 && !residual) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.y_f16 = y_f16 ? 1 : 0; a.res_f16 = res_f16 ? 1 : 0;
    a.x = x; a.y = y; a.idx = active_indices; a.map = scatter_map; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.hp_shift = stacked_shift(H);  // (stacked edits: sige_hip_set_edit_batch)
    if (a.hp_shift < 0 || (a.hp_shift && B != 1)) return SIGE_HIP_EUNSUPPORTED;
    a.RxSx = Rx * Sx; a.Sx = Sx;
    a.scale = scale; a.shift = shift;
    const int mode = staging_mode(scale, scaleB, scaleC, shift, shiftB, shiftC, activation, B, Cin, &a.aff_sb, &a.aff_sc);
    if (mode < 0) return SIGE_HIP_EUNSUPPORTED;
    a.residual = residual; a.Ho = H; a.Wo = W; a.offH = offsetH; a.offW = offsetW; a.strH = 1; a.strW = 1;
    a.x1 = x1; a.table1 = table1; a.gW1 = gW1; a.N1 = N1; a.R1 = R1 > 0 ? R1 : 1; a.S1 = S1 > 0 ? S1 : 1;
    if ((twin0 && (!twin0_scale || !twin0_shift)) || (twin1 && (!twin1_scale || !twin1_shift))) return SIGE_HIP_EINVAL;
    if ((reinterpret_cast<uintptr_t>(twin0) | reinterpret_cast<uintptr_t>(twin1) | reinterpret_cast<uintptr_t>(twin0_scale) |
         reinterpret_cast<uintptr_t>(twin0_shift) | reinterpret_cast<uintptr_t>(twin1_scale) | reinterpret_cast<uintptr_t>(twin1_shift)) & 15)
        return SIGE_HIP_EUNSUPPORTED;
    a.twin0 = twin0; a.tscale0 = twin0_scale; a.tshift0 = twin0_shift;
    a.twin1 = twin1; a.tscale1 = twin1_scale; a.tshift1 = twin1_shift;
    return launch_kind<3, 1, 6, SRC_SCATTER_GATHER, DST_NCHW, LAYOUT_NHWC, PREC>(a, mode, as_stream(stream)) != SIGE_HIP_OK
               ? SIGE_HIP_EUNSUPPORTED : launch_status();
}

#define SIGE_SGS_CONV_ARGS x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, \
    shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, \
    twin1_scale, twin1_shift, out, stream
extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_f32(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_f32, (sige::CountOf<10, 11>, sige::CountOf<29, 32>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return scatter_gather_conv_scatter_nhwc_impl<0>(SIGE_SGS_CONV_ARGS);
}
extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_f16c(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_f16c, (sige::CountOf<10, 11>, sige::CountOf<29, 32>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return scatter_gather_conv_scatter_nhwc_impl<1>(SIGE_SGS_CONV_ARGS);
}
extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_f16x3(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_f16x3, (sige::CountOf<10, 11>, sige::CountOf<29, 32>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    return scatter_gather_conv_scatter_nhwc_impl<2>(SIGE_SGS_CONV_ARGS);
}

// ---- the same two fused launches over fp16-STORED caches (SURVEY.md 8b export list "_f16", 8f row 4) ----
// `y` (the cached tensor of the ScatterGather, or its activated copy) and -- for a fused ScatterWithBlockResidual (x1 != NULL) --
// `residual` (the cached shortcut tensor) hold halves; everything else as in the fp32-storage entry points.  `compute`: 0 exact
// fp32 products | 1 fp16 operands | 2 split fp16 operands (which packing `packed` has).
extern "C" int sige_hip_scatter_gather_conv_nhwc_c16(int compute, const float *x, const void *y, int B, int Cin, int H, int W,
                                                     int Rx, int Sx, int bH, int bW,
                                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                     const float *scale, int scaleB, int scaleC,
                                                     const float *shift, int shiftB, int shiftC,
                                                     int activation,
                                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                                     int strideH, int strideW, float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_nhwc_c16, (sige::CountOf<11, 12>), compute, x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream);
    const float *yh = static_cast<const float *>(y);  // (halves behind a float pointer: ConvArgs::y_f16)
    if (compute == 0) return scatter_gather_conv_nhwc_impl<0>(x, yh, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream, 1);
    if (compute == 1) return scatter_gather_conv_nhwc_impl<1>(x, yh, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream, 1);
    if (compute == 2) return scatter_gather_conv_nhwc_impl<2>(x, yh, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, out, stream, 1);
    return SIGE_HIP_EINVAL;
}

extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_c16(
        int compute, const float *x, const void *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_c16, (sige::CountOf<11, 12>, sige::CountOf<31, 34>), compute, x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, residual_f16, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
    const float *yh = static_cast<const float *>(y), *rh = static_cast<const float *>(residual);
#define SIGE_SGS_C16_ARGS x, yh, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, \
    shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, rh, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, \
    twin1_scale, twin1_shift, out, stream, 1, residual_f16
    if (compute == 0) return scatter_gather_conv_scatter_nhwc_impl<0>(SIGE_SGS_C16_ARGS);
    if (compute == 1) return scatter_gather_conv_scatter_nhwc_impl<1>(SIGE_SGS_C16_ARGS);
    if (compute == 2) return scatter_gather_conv_scatter_nhwc_impl<2>(SIGE_SGS_C16_ARGS);
#undef SIGE_SGS_C16_ARGS
    return SIGE_HIP_EINVAL;
}

extern "C" int sige_hip_block_conv_direct_f32(const float *x, int T, int Cin, int R, int S,
                                              const float *w, const float *bias, int Cout, int kH, int kW,
                                              int strideH, int strideW, int dilationH, int dilationW, int groups,
                                              float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_block_conv_direct_f32, x, T, Cin, R, S, w, bias, Cout, kH, kW, strideH, strideW, dilationH, dilationW, groups, out, stream);
    if (T < 0 || Cin <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || strideH <= 0 || strideW <= 0 || groups <= 0 ||
        dilationH <= 0 || dilationW <= 0)
        return SIGE_HIP_EINVAL;
    const int eH = (kH - 1) * dilationH + 1, eW = (kW - 1) * dilationW + 1;  // extent of the dilated kernel
    if (Cin % groups || Cout % groups || R < eH || S < eW) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !w || !out) return SIGE_HIP_EINVAL;
    const int Ro = (R - eH) / strideH + 1, So = (S - eW) / strideW + 1;
    const long total = (long)T * Cout * Ro * So;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    block_conv_direct_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, w, bias, out, T, Cin, R, S, Cout, kH, kW,
                                                                   strideH, strideW, dilationH, dilationW, groups, Ro, So, total);
    return launch_status();
}

extern "C" int sige_hip_release_graph_tickets(void) { return sige::release_graph_tickets(); }

// ---- routing entry points: conv_mfma.hpp or the tile conv v3 (conv_tile3.hpp), decided HERE from the tile count ----
// The same calls as sige_hip_gather_conv_nhwc_f32 / sige_hip_scatter_gather_conv_scatter_nhwc_f32 plus the weights in the v3
// layout and a threshold: a launch whose v3 grid -- tile pairs x 64-channel output blocks -- has at least `min_blocks` workgroups
// runs on the v3 kernel, anything else exactly as before.  The decision is taken in C so that a launch plan, which replays the
// recorded ENTRY POINT under a new mask's tile count, routes like the module-level forward under that mask does: plan and
// module path stay bit-identical (round 5: the first router lived in Python and test_launch_plan_follows_mask_changes caught
// the two running different kernels).
// fp16 operands: a conv1 whose shortcut is held goes to the v3 kernel anyway from this many workgroups on (the shortcut is then
// launched on its own: tile_conv3_launch -> flush_held_conv); SIGE_HIP_TUNE_TILE3_F16_PAIR_MIN = -1: this value, 0 = never
constexpr int kTile3F16PairMin = 256;  // (profiles/r6j_tile3_f16_pairs_bench.json: -2.6 % at a 15 % edit, -3.3 % at 20 %, nothing lost below)

// fp16 operands: launches over a SPARSE tile list (fewer tiles than the tensor has 4 x 4 cells) go to v3 from this many workgroups
// on; dense layers (every tile active: deep K over a few pixels, where conv_mfma.hpp's K split wins) keep `min_blocks`.
// SIGE_HIP_TUNE_TILE3_F16_SPARSE_MIN = -1: this value, 0 = no separate rule
constexpr int kTile3F16SparseMin = 128;  // (profiles/r6o_tile3_f16_sparse_min.json: -1 % at 10 - 20 % edits, nothing lost at 1.2 / 5 %; 32 - 96 lose at 1.2 %)

static bool tile3_takes(const float *packed_tile3, int min_blocks, int B, int N, int C1, int C2, int Cout, int kH, int kW, int bH, int bW,
                        int strideH, int strideW, hipStream_t st, bool f16 = false, int H = 0, int W = 0) {
    if (!packed_tile3 || min_blocks <= 0) return false;
    if (kH != 3 || kW != 3 || bH != 6 || bW != 6 || strideH != 1 || strideW != 1) return false;
    if (!sige_hip_tile_conv3_supported(C1, C2, Cout)) return false;
    const long blocks = (long)((B * (long)N + 1) / 2) * (Cout / 64);
    int need = min_blocks;
    if (f16 && H > 0 && (long)N * 16 < (long)H * W) {  // a sparse tile list
        int sm = tuning(SIGE_HIP_TUNE_TILE3_F16_SPARSE_MIN);
        if (sm < 0) sm = kTile3F16SparseMin;
        if (sm > 0 && sm < need) need = sm;
    }
    if (blocks < need) return false;
    // a 1x1 shortcut held by conv_pair_begin() shares the conv_mfma.hpp launch of this conv1: keeping the pair beats the v3 kernel
    // plus a launch of its own for the shortcut (46.7 vs 39.5 + 8.7 us at a 15 % edit: profiles/r5j_sequence_15pct_*.csv)
    if (g_held.active && g_held.st == st) {
        if (!f16) return false;
        int pm = tuning(SIGE_HIP_TUNE_TILE3_F16_PAIR_MIN);
        if (pm < 0) pm = kTile3F16PairMin;
        if (pm == 0 || blocks < pm) return false;
    }
    return true;
}

extern "C" int sige_hip_gather_conv_nhwc_v3_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                                int bH, int bW, const int32_t *active_indices, int N,
                                                const float *scale, int scaleB, int scaleC,
                                                const float *shift, int shiftB, int shiftC,
                                                int activation,
                                                const float *packed, const float *bias, int Cout, int kH, int kW,
                                                int strideH, int strideW,
                                                int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                                float *workspace, size_t workspace_floats,
                                                const float *out_scale, const float *out_shift, int out_activation,
                                                int upsample2x,
                                                float *twin0, const float *twin0_scale, const float *twin0_shift,
                                                float *twin1, const float *twin1_scale, const float *twin1_shift,
                                                const float *packed_tile3, int min_blocks,
                                                float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_conv_nhwc_v3_f32, (sige::CountOf<9, 10>), x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace, workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, packed_tile3, min_blocks, out, stream);
    const int Cin = C1 + C2;
    const bool aff_ok = (!scale && !shift) || (scale && shift && scaleC == Cin && shiftC == Cin && scaleB == shiftB && (scaleB == 1 || scaleB == B));
    if (B > 0 && N > 0 && aff_ok && tile3_takes(packed_tile3, min_blocks, B, N, C1, C2, Cout, kH, kW, bH, bW, strideH, strideW, as_stream(stream))) {
        const int rc = tile_conv3_launch(T3_GATHER, x, x2, B, C1, C2, H, W, upsample2x, active_indices, N, nullptr, 0, 0, scale, shift,
                                         scale ? scaleB : 0, activation, packed_tile3, bias, Cout, to_full, offsetH, offsetW, Ho, Wo,
                                         to_full ? residual : nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, out_scale, out_shift, out_activation,
                                         twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
        if (rc != SIGE_HIP_EUNSUPPORTED) return rc;
    }
    return gather_conv_nhwc_impl<0>(SIGE_GATHER_CONV_ARGS);
}

extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_v3_f32(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        const float *packed_tile3, int min_blocks,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_v3_f32, (sige::CountOf<10, 11>, sige::CountOf<29, 32>), x, y, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, packed_tile3, min_blocks, out, stream);
    const bool aff_ok = (!scale && !shift && activation == SIGE_HIP_ACT_IDENTITY) ||
                        (scale && shift && scaleC == Cin && shiftC == Cin && scaleB == shiftB && (scaleB == 1 || scaleB == B));
    if (B > 0 && N > 0 && aff_ok &&
        tile3_takes(packed_tile3, min_blocks, B, N, Cin, 0, Cout, kH, kW, bH, bW, 1, 1, as_stream(stream))) {
        const int rc = tile_conv3_launch(T3_SCATTER_GATHER, x, y, B, Cin, 0, H, W, 0, active_indices, N, scatter_map, Rx, Sx, scale, shift, scale ? scaleB : 0,
                                         activation, packed_tile3, bias, Cout, 1, offsetH, offsetW, H, W, residual,
                                         x1, table1, gH1, gW1, N1, R1, S1, nullptr, nullptr, 0,
                                         twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream);
        if (rc != SIGE_HIP_EUNSUPPORTED) return rc;
    }
    return scatter_gather_conv_scatter_nhwc_impl<0>(SIGE_SGS_CONV_ARGS);
}


// ... and for fp16 operands (round 6, BASELINE.json configs[4]): sige_hip_gather_conv_nhwc_f16c /
// sige_hip_scatter_gather_conv_scatter_nhwc_f16c | _c16(compute = 1) with the weights in the v3 fp16 layout
// (`packed_tile3` = sige_hip_wide_conv_pack(prec = 0)) and a threshold beside them; routed exactly like the fp32 pair above.
extern "C" int sige_hip_gather_conv_nhwc_v3_f16c(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                                 int bH, int bW, const int32_t *active_indices, int N,
                                                 const float *scale, int scaleB, int scaleC,
                                                 const float *shift, int shiftB, int shiftC,
                                                 int activation,
                                                 const float *packed, const float *bias, int Cout, int kH, int kW,
                                                 int strideH, int strideW,
                                                 int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                                 float *workspace, size_t workspace_floats,
                                                 const float *out_scale, const float *out_shift, int out_activation,
                                                 int upsample2x,
                                                 float *twin0, const float *twin0_scale, const float *twin0_shift,
                                                 float *twin1, const float *twin1_scale, const float *twin1_shift,
                                                 const float *packed_tile3, int min_blocks,
                                                 float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_gather_conv_nhwc_v3_f16c, (sige::CountOf<9, 10>), x, x2, B, C1, C2, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, strideH, strideW, to_full, offsetH, offsetW, residual, Ho, Wo, workspace, workspace_floats, out_scale, out_shift, out_activation, upsample2x, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, packed_tile3, min_blocks, out, stream);
    const int Cin = C1 + C2;
    const bool aff_ok = (!scale && !shift) || (scale && shift && scaleC == Cin && shiftC == Cin && scaleB == shiftB && (scaleB == 1 || scaleB == B));
    if (B > 0 && N > 0 && aff_ok && tile3_takes(packed_tile3, min_blocks, B, N, C1, C2, Cout, kH, kW, bH, bW, strideH, strideW, as_stream(stream), true, H, W)) {
        const int rc = tile_conv3_launch(T3_GATHER, x, x2, B, C1, C2, H, W, upsample2x, active_indices, N, nullptr, 0, 0, scale, shift,
                                         scale ? scaleB : 0, activation, packed_tile3, bias, Cout, to_full, offsetH, offsetW, Ho, Wo,
                                         to_full ? residual : nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, out_scale, out_shift, out_activation,
                                         twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream, WIDE_F16, 0, 0);
        if (rc != SIGE_HIP_EUNSUPPORTED) return rc;
    }
    return gather_conv_nhwc_impl<1>(SIGE_GATHER_CONV_ARGS);
}

// y_f16 / residual_f16: the cached tensor / the cached shortcut tensor hold halves (the _c16 form); 0 / 0 = the _f16c form
extern "C" int sige_hip_scatter_gather_conv_scatter_nhwc_v3_f16c(
        const float *x, const void *y, int y_f16, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        const float *packed_tile3, int min_blocks,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_gather_conv_scatter_nhwc_v3_f16c, (sige::CountOf<11, 12>, sige::CountOf<31, 34>), x, y, y_f16, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, residual, residual_f16, x1, table1, gH1, gW1, N1, R1, S1, twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, packed_tile3, min_blocks, out, stream);
    const float *yh = static_cast<const float *>(y), *rh = static_cast<const float *>(residual);
    const bool aff_ok = (!scale && !shift && activation == SIGE_HIP_ACT_IDENTITY) ||
                        (scale && shift && scaleC == Cin && shiftC == Cin && scaleB == shiftB && (scaleB == 1 || scaleB == B));
    if (B > 0 && N > 0 && aff_ok &&
        tile3_takes(packed_tile3, min_blocks, B, N, Cin, 0, Cout, kH, kW, bH, bW, 1, 1, as_stream(stream), true, H, W)) {
        const int rc = tile_conv3_launch(T3_SCATTER_GATHER, x, yh, B, Cin, 0, H, W, 0, active_indices, N, scatter_map, Rx, Sx, scale, shift, scale ? scaleB : 0,
                                         activation, packed_tile3, bias, Cout, 1, offsetH, offsetW, H, W, rh,
                                         x1, table1, gH1, gW1, N1, R1, S1, nullptr, nullptr, 0,
                                         twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream, WIDE_F16, y_f16, residual_f16);
        if (rc != SIGE_HIP_EUNSUPPORTED) return rc;
    }
    return scatter_gather_conv_scatter_nhwc_impl<1>(x, yh, B, Cin, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, shift, shiftB,
                                                    shiftC, activation, packed, bias, Cout, kH, kW, offsetH, offsetW, rh, x1, table1, gH1, gW1, N1, R1, S1,
                                                    twin0, twin0_scale, twin0_shift, twin1, twin1_scale, twin1_shift, out, stream, y_f16, residual_f16);
}
