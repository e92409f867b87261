// Input convolution of the U-Net: a 3x3 / padding-1 conv from a handful of input channels
// (3 for DDPM's conv_in) to Cout = 32..128 over the FULL-resolution image, which the
// reference runs densely in sparse mode too (sige_fused_unet.py:395, a plain nn.Conv2d).
//
// With K = 9*Cin = 27 the layer is one thin GEMM per pixel block: M = 32 pixels, N = Cout,
// K = 27 (padded to 28 = 14 steps of the 32x32x2 f32 MFMA).  It is bound by the 33.5 MB of
// channels-last output (0.8 MB in), so the kernel is arranged around the stores.
// Lane (kq, j) owns pixel j of the block and the K indices 2s + kq.
// The input is addressed through element strides (NCHW or channels-last alike).
#include "common.hpp"

namespace sige {

typedef float floatx16_in __attribute__((ext_vector_type(16)));

// Round 6 (16.6 us = 2.0 TB/s of stores, one wave per SIMD running load -> MFMA -> store in sequence):
//   * ONE 32-pixel block per wave and 8 waves per CU (512 workgroups at 256 x 256): the A loads of one wave, the matrix
//     instructions of another and the stores of a third overlap;
//   * the weights reach the lanes through LDS (one coalesced read of the 13.8 KB tensor per workgroup; before: 56 dword loads
//     per lane at a 108-byte lane stride);
//   * the accumulators go through a wave-private LDS tile [32 pixels][Cout] and leave as 16-byte stores whose 64 lanes cover
//     1 KB of consecutive addresses (before: 64 dword stores per lane).
template <int CIN, int NBK>
__global__ __launch_bounds__(256) void conv_in_gemm_kernel(const float *__restrict__ x, long sb, long sc, long sh, long sw,
                                                          int B, int H, int W, const float *__restrict__ w,  // [Cout, CIN, 3, 3]
                                                          const float *__restrict__ bias, float *__restrict__ out, long npix,
                                                          const int32_t *__restrict__ idx, int N, int bH, int bW) {
    kernarg_touch<128>();
    constexpr int K = 9 * CIN, KS = (K + 1) / 2, COUT = 32 * NBK;
    constexpr int KP = 2 * KS + 1;  // odd row pitch of the weight stage: lane j reads row j -> 32 different banks
    __shared__ __attribute__((aligned(16))) float wl[COUT * KP];
    __shared__ __attribute__((aligned(16))) float tile[4][32 * COUT];
    __shared__ long opix[4][32];  // tile-list form: the output pixel of each row of a wave's block (-1: none)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kq = lane >> 5;
    // (32-bit index arithmetic throughout: the host refuses a launch whose pixel count or largest input offset does not fit -- a
    //  64-bit division by a run-time value is a branchy ~130-instruction routine, and there were three of them in front of the
    //  first load)
    const int blk = (int)blockIdx.x * 4 + wave;
    const int p = blk * 32 + j;
    bool live = p < (int)npix;

    // the weights: every lane's share requested NOW, written to LDS after the A operand's loads are out as well (the rolled
    // `for (i = tid; ...) wl[..] = w[i]` was 14 dependent load -> wait -> ds_write round trips in front of the barrier: 12 of the
    // kernel's 14 us inside the forward, where each trip goes to a cold L2)
    constexpr int WN = COUT * K, WIT = (WN + 255) / 256;
    float wv[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) wv[it] = w[min(tid + 256 * it, WN - 1)];

    // A operand: lane (kq, j) owns pixel j and the K indices 2s + kq = ci*9 + tap
    float a[KS];
    {
        unsigned okm = 0;
        int b, h, ww;
        if (idx) {
            // tile-list form (round 6): pixel p of the launch = pixel (p % (bH*bW)) of window (p / (bH*bW)) % N of image p / (N*bH*bW);
            // windows are clipped to the image, overlapping windows recompute the same values
            const unsigned win = (unsigned)(bH * bW);
            const unsigned t = (unsigned)p / win;
            const int q = (int)((unsigned)p - t * win);
            b = (int)(t / (unsigned)N);
            const int n = (int)(t - (unsigned)b * (unsigned)N);
            const int2 o = *reinterpret_cast<const int2 *>(idx + 2 * (live ? n : 0));
            const int qh = (int)((unsigned)q / (unsigned)bW);
            h = o.x + qh; ww = o.y + (q - qh * bW);
            live = live && h >= 0 && h < H && ww >= 0 && ww < W;
            if (kq == 0) opix[wave][j] = live ? ((long)b * H + h) * W + ww : -1;
        } else {
            const unsigned hw = (unsigned)(H * W);
            b = (int)((unsigned)p / hw);
            const int rem = (int)((unsigned)p - (unsigned)b * hw);
            h = (int)((unsigned)rem / (unsigned)W); ww = rem - h * W;
        }
        const int isb = (int)sb, isc = (int)sc, ish = (int)sh, isw = (int)sw;
        const int base = b * isb + h * ish + ww * isw;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (the two candidates of a lane -- K index 2s or 2s + 1 -- are wave-uniform offsets: scalar arithmetic, one select)
            const int k0 = 2 * s, k1 = min(2 * s + 1, K - 1);
            const int off0 = (k0 / 9) * isc + ((k0 % 9) / 3 - 1) * ish + ((k0 % 9) % 3 - 1) * isw;
            const int off1 = (k1 / 9) * isc + ((k1 % 9) / 3 - 1) * ish + ((k1 % 9) % 3 - 1) * isw;
            const int tap = kq ? (2 * s + 1) % 9 : (2 * s) % 9;
            const int ih = h + tap / 3 - 1, iw = ww + tap % 3 - 1;
            const bool ok = live && 2 * s + kq < K && ih >= 0 && ih < H && iw >= 0 && iw < W;
            okm |= ok ? (1u << s) : 0u;
            // (branch-free: a conditional load compiles to an exec-masked branch per element with s_waitcnt vmcnt(0) between groups.
            //  Here every lane loads a valid address -- element 0 when the tap is padding -- and the select follows, all loads in flight)
            a[s] = x[ok ? base + (kq ? off1 : off0) : 0];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = (okm >> s) & 1u ? a[s] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int i = min(tid + 256 * it, WN - 1);  // (past the end: the last element again, the same value to the same place)
        wl[(i / K) * KP + i % K] = wv[it];
    }
    if (K < 2 * KS)
        for (int n = tid; n < COUT; n += 256) wl[n * KP + K] = 0.f;  // (K odd: the padded k index)
    float biasr[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) biasr[nb] = bias ? bias[nb * 32 + j] : 0.f;
    __syncthreads();
    if ((long)blk * 32 >= npix) return;  // (wave-uniform; after the only barrier)

    floatx16_in acc[NBK];
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = biasr[nb];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wl[(nb * 32 + j) * KP + 2 * s + kq], acc[nb], 0, 0, 0);
    // reg r of lane (kq, j): pixel row m = (r & 3) + 8 * (r >> 2) + 4 * kq, output channel nb*32 + j  ->  tile[m][channel]
    float *const tl = tile[wave];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) tl[((r & 3) + 8 * (r >> 2) + 4 * kq) * COUT + nb * 32 + j] = acc[nb][r];
    __builtin_amdgcn_wave_barrier();  // (LDS is in order per wave: the tile is this wave's own)
    if (idx) {
        // tile-list form: row m of the tile goes to output pixel opix[m] (COUT consecutive floats each)
#pragma unroll
        for (int i = 0; i < 32 * COUT / 256; ++i) {
            const int o = (i * 64 + lane) * 4;
            const long px = opix[wave][o / COUT];
            if (px >= 0) store_out4(out + px * COUT + o % COUT, *reinterpret_cast<const float4 *>(tl + o));
        }
        return;
    }
    // the tile is 32 * COUT consecutive floats of the output: 64 lanes x 16 bytes = 1 KB per store instruction
    float *const ob = out + (long)blk * 32 * COUT;
    const long left = (npix - (long)blk * 32) * COUT;  // floats of the output from this block on
#pragma unroll
    for (int i = 0; i < 32 * COUT / 256; ++i) {
        const int o = (i * 64 + lane) * 4;
        if (o < left) store_out4(ob + o, *reinterpret_cast<const float4 *>(tl + o));
    }
}

}  // namespace sige

using namespace sige;

static int conv3x3_small_cin_impl(const float *x, int64_t strideB, int64_t strideC, int64_t strideH, int64_t strideW,
                                  int B, int Cin, int H, int W, const float *weight, const float *bias, int Cout,
                                  const int32_t *active_indices, int N, int bH, int bW, float *out, void *stream) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (!x || !weight || !out) return SIGE_HIP_EINVAL;
    if (Cin > 3 || !(Cout == 32 || Cout == 64 || Cout == 128)) return SIGE_HIP_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(out) & 15) return SIGE_HIP_EUNSUPPORTED;
    const long npix = active_indices ? (long)B * N * bH * bW : (long)B * H * W;
    if (npix == 0) return SIGE_HIP_OK;
    if ((long)B * H * W * Cout >= (1L << 40)) return SIGE_HIP_EUNSUPPORTED;
    const long nblk = (npix + 31) / 32, grid = (nblk + 3) / 4;  // one 32-pixel block per wave
    // (the kernel's index arithmetic is 32-bit: pixel count -- padded to whole workgroups -- and the largest input element offset)
    auto mag = [](int64_t v) { return v < 0 ? -v : v; };
    const int64_t reach = (int64_t)(B - 1) * mag(strideB) + (int64_t)(Cin - 1) * mag(strideC) + (int64_t)H * mag(strideH) + (int64_t)W * mag(strideW);
    if (grid * 128 > 0x7fffffffL || reach > 0x7fffffffL || (long)H * W > 0x7fffffffL || (active_indices && (long)N * bH * bW > 0x7fffffffL))
        return SIGE_HIP_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
#define SIGE_CI(CI, NBK) \
    conv_in_gemm_kernel<CI, NBK><<<(int)grid, 256, 0, st>>>(x, strideB, strideC, strideH, strideW, B, H, W, weight, bias, out, npix, active_indices, N, bH, bW);
#define SIGE_CI_N(CI)                                                                         \
    if (Cout == 32) { SIGE_CI(CI, 1) } else if (Cout == 64) { SIGE_CI(CI, 2) } else { SIGE_CI(CI, 4) }
    if (Cin == 1) { SIGE_CI_N(1) } else if (Cin == 2) { SIGE_CI_N(2) } else { SIGE_CI_N(3) }
#undef SIGE_CI_N
#undef SIGE_CI
    return launch_status();
}

extern "C" int sige_hip_conv3x3_small_cin_nhwc_f32(const float *x, int64_t strideB, int64_t strideC, int64_t strideH, int64_t strideW,
                                                   int B, int Cin, int H, int W,
                                                   const float *weight, const float *bias, int Cout,
                                                   float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_conv3x3_small_cin_nhwc_f32, x, strideB, strideC, strideH, strideW, B, Cin, H, W, weight, bias, Cout, out, stream);
    return conv3x3_small_cin_impl(x, strideB, strideC, strideH, strideW, B, Cin, H, W, weight, bias, Cout, nullptr, 0, 0, 0, out, stream);
}

// The same conv evaluated ONLY on the bH x bW windows at active_indices (clipped to the image), written into `out` [B,H,W,Cout] in
// place; every other pixel of `out` is left as it is.  In sparse mode the first conv's output is only ever read through Gather
// windows (sige_fused_unet.py:395-400: hs[0] feeds down[0].block[0] and, as a skip, the last up block -- both tiled at this
// resolution with the same index list), so the 94 % of it outside the active windows of a 1.2 % edit is never looked at.
extern "C" int sige_hip_conv3x3_small_cin_tiles_nhwc_f32(const float *x, int64_t strideB, int64_t strideC, int64_t strideH, int64_t strideW,
                                                         int B, int Cin, int H, int W,
                                                         const float *weight, const float *bias, int Cout,
                                                         const int32_t *active_indices, int N, int bH, int bW,
                                                         float *out, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_conv3x3_small_cin_tiles_nhwc_f32, (sige::CountOf<12, 13>), x, strideB, strideC, strideH, strideW, B, Cin, H, W, weight, bias, Cout, active_indices, N, bH, bW, out, stream);
    if (N < 0 || bH <= 0 || bW <= 0 || (N > 0 && !active_indices)) return SIGE_HIP_EINVAL;
    if (N == 0) return SIGE_HIP_OK;
    return conv3x3_small_cin_impl(x, strideB, strideC, strideH, strideW, B, Cin, H, W, weight, bias, Cout, active_indices, N, bH, bW, out, stream);
}
