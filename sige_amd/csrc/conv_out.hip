// Output convolution of the U-Net: a 3x3 / padding-1 conv with a handful of output
// channels (3 for DDPM's conv_out) over the FULL-resolution activation, with the
// preceding GroupNorm affine + SiLU fused into the input path.
//
// In SIGE's sparse mode the reference still runs norm_out / swish / conv_out densely
// on the whole 256x256x128 tensor (sige_fused_unet.py:430-434).  A 128 -> 3 conv is a
// poor fit for the matrix cores (3 of 16 output columns used); it is bandwidth-bound
// (33.5 MB in, 0.8 MB out).  Here: channels-last input, one workgroup per 8x32 pixel
// tile, the activated (halo-extended) tile of a 32-channel chunk staged once in LDS,
// every lane accumulates its pixel's COUT outputs with 16-byte LDS reads; the weights
// are read with scalar loads (wave-uniform addresses) and enter the FMAs as SGPR operands.
#include "common.hpp"

namespace sige {

__device__ __forceinline__ float silu_fast(float z) {
    const float e = __builtin_amdgcn_exp2f(z * -1.44269504088896341f);
    return z * __builtin_amdgcn_rcpf(1.0f + e);
}

constexpr int kTH = 8, kTW = 32, kCC = 32;  // tile rows / cols, channels per chunk
constexpr int kPH = kTH + 2, kPW = kTW + 2;
constexpr int kLDC = kCC + 4;               // padded pixel pitch in LDS (floats)

template <int COUT, int ACT>
__global__ __launch_bounds__(256) void conv_out_nhwc_kernel(const float *__restrict__ x, int B, int C, int H, int W,
                                                           const float *__restrict__ scale, const float *__restrict__ shift, int aff_sb,
                                                           const float *__restrict__ w,  // [COUT, C, 3, 3]
                                                           const float *__restrict__ bias, float *__restrict__ out, float slope, int out_act) {
    __shared__ __attribute__((aligned(16))) float tile[kPH * kPW * kLDC];
    const int tid = threadIdx.x;
    const int tx = tid % kTW, ty = tid / kTW;
    const int tilesW = (W + kTW - 1) / kTW, tilesH = (H + kTH - 1) / kTH;
    const int b = blockIdx.x / (tilesW * tilesH);
    const int tr = blockIdx.x % (tilesW * tilesH);
    const int h0 = (tr / tilesW) * kTH, w0 = (tr % tilesW) * kTW;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
    for (int c0 = 0; c0 < C; c0 += kCC) {
        __syncthreads();
        // stage act(scale*x + shift) of the (8+2) x (32+2) pixel window, channels c0..c0+31 (zeros outside the
        // image).  All of a lane's loads are issued before the first one is used (a rolled loop would pay the
        // memory latency once per iteration: 11 x ~1.5 us per chunk -- measured 52 us for the kernel).
        constexpr int kUnits = kPH * kPW * (kCC / 4), kIter = (kUnits + 255) / 256;
        float4 raw[kIter];
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            const int u = tid + 256 * it;
            const int c4 = (u % (kCC / 4)) * 4, p = u / (kCC / 4);
            const int h = h0 + p / kPW - 1, ww = w0 + p % kPW - 1;
            const bool ok = u < kUnits && h >= 0 && h < H && ww >= 0 && ww < W && c0 + c4 < C;
            raw[it] = ok ? *reinterpret_cast<const float4 *>(x + (((size_t)b * H + h) * W + ww) * C + c0 + c4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < kIter; ++it) {
            const int u = tid + 256 * it;
            if (u >= kUnits) break;
            const int c4 = (u % (kCC / 4)) * 4, p = u / (kCC / 4);
            const int h = h0 + p / kPW - 1, ww = w0 + p % kPW - 1;
            float4 v = raw[it];
            if (h >= 0 && h < H && ww >= 0 && ww < W && c0 + c4 < C) {
                if (scale) {
                    const float4 s4 = *reinterpret_cast<const float4 *>(scale + b * aff_sb + c0 + c4);
                    const float4 t4 = *reinterpret_cast<const float4 *>(shift + b * aff_sb + c0 + c4);
                    v.x = s4.x * v.x; v.y = s4.y * v.y; v.z = s4.z * v.z; v.w = s4.w * v.w;
                    v.x = t4.x + v.x; v.y = t4.y + v.y; v.z = t4.z + v.z; v.w = t4.w + v.w;
                }
                if (ACT == SIGE_HIP_ACT_SWISH) {  // v_exp_f32 / v_rcp_f32 form (<= 1e-6 relative), as in the fused conv staging
                    v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
                }
                if (ACT == SIGE_HIP_ACT_LEAKY) {
                    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
                    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
                }
            }
            *reinterpret_cast<float4 *>(tile + p * kLDC + c4) = v;
        }
        __syncthreads();
        // weights straight from global memory with wave-uniform addresses: scalar loads into SGPRs, so
        // the only LDS traffic is the activations (LDS-bandwidth was the first version's limit: 61 us)
#pragma unroll
        for (int c4 = 0; c4 < kCC; c4 += 4) {
            if (c0 + c4 >= C) break;  // uniform
            float4 a[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                a[tap] = *reinterpret_cast<const float4 *>(tile + ((ty + tap / 3) * kPW + tx + tap % 3) * kLDC + c4);
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float *wp = w + ((size_t)co * C + c0 + c4) * 9;  // [4 channels][9 taps], contiguous
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    acc[co] = fmaf(a[tap].x, wp[tap], acc[co]);
                    acc[co] = fmaf(a[tap].y, wp[9 + tap], acc[co]);
                    acc[co] = fmaf(a[tap].z, wp[18 + tap], acc[co]);
                    acc[co] = fmaf(a[tap].w, wp[27 + tap], acc[co]);
                }
            }
        }
    }
    const int h = h0 + ty, ww = w0 + tx;
    if (h < H && ww < W) {
        float *o = out + (((size_t)b * H + h) * W + ww) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) o[co] = out_act == SIGE_HIP_ACT_TANH ? tanhf(acc[co]) : acc[co];
    }
}


// ---- tap-GEMM form (C = 64 / 128, COUT <= 3): the matrix cores after all ------------------------------
// Taking the 9 taps as extra output COLUMNS instead of extra K makes the layer a plain GEMM,
//     P[p][tap*COUT + co] = sum_ch act(x[p][ch]) * w[co][ch][tap]      M = pixels, N = 9*COUT <= 32, K = C
// followed by a 9-term shifted sum   out[y][x][co] = bias[co] + sum_tap P[(y+dy, x+dx)][tap*COUT + co].
// 27 of the 32 columns of a 32x32x2 f32 MFMA do useful work (3 of 16 in the implicit-GEMM form), every
// activation is computed once per workgroup, and the only VALU work left is the affine + SiLU itself.
// One workgroup = a 16x16 output tile: its (16+2)^2 = 324 input pixels are 11 M-blocks of 32 shared out over
// 4 waves; lane (kq, j) owns pixel j of the block and channels [kq*C/2, (kq+1)*C/2) -- the weights of those
// channels stay in C/2 registers for the whole kernel, the pixel's channels arrive as 16-byte loads that are
// issued one block ahead.  P lives in LDS (38 KB); out-of-image taps are skipped by the summing lane, which is
// the zero padding of the activated tensor.
constexpr int kGT = 16, kGP = kGT + 2, kGPix = kGP * kGP, kGBlocks = (kGPix + 31) / 32;
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int CIN, int COUT, int ACT>
__global__ __launch_bounds__(256) void conv_out_gemm_kernel(const float *__restrict__ x, int B, int H, int W,
                                                           const float *__restrict__ scale, const float *__restrict__ shift, int aff_sb,
                                                           const float *__restrict__ w,  // [COUT, CIN, 3, 3]
                                                           const float *__restrict__ bias, float *__restrict__ out, float slope, int out_act) {
    kernarg_touch<128>();
    constexpr int KS = CIN / 2;  // MFMA steps (K = 2 each: one channel of either half)
    constexpr int NV = KS / 4;   // 16-byte loads per lane and block
    constexpr int NP = 9 * COUT;
    __shared__ float P[kGBlocks * 32 * NP];
    __shared__ __attribute__((aligned(16))) float s_sc[CIN], s_sh[CIN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kq = lane >> 5;
    const int tilesW = (W + kGT - 1) / kGT, tilesH = (H + kGT - 1) / kGT;
    const int b = blockIdx.x / (tilesW * tilesH);
    const int tr = blockIdx.x % (tilesW * tilesH);
    const int h0 = (tr / tilesW) * kGT, w0 = (tr % tilesW) * kGT;

    auto pixel = [&](int blk) -> const float4 * {
        // (clamped: slots past the tile and pixels outside the image read a valid address; nobody sums them)
        const int p = min(blk * 32 + j, kGPix - 1);
        const int h = min(max(h0 + p / kGP - 1, 0), H - 1), ww = min(max(w0 + p % kGP - 1, 0), W - 1);
        return reinterpret_cast<const float4 *>(x + (((size_t)b * H + h) * W + ww) * CIN + kq * KS);
    };
    float4 raw[NV];
    {
        const float4 *src = pixel(wave);
#pragma unroll
        for (int i = 0; i < NV; ++i) raw[i] = src[i];
    }
    // B operand: column j = tap*COUT + co of the channels of this lane's half
    float breg[KS];
    {
        // (columns j >= NP are never written to P: they read column NP - 1's weights instead of taking an exec-masked branch
        //  around each of the KS loads)
        const int jc = min(j, NP - 1);
        const int co = jc % COUT, tap = jc / COUT;
        const float *wp = w + ((size_t)co * CIN + kq * KS) * 9 + tap;
#pragma unroll
        for (int s = 0; s < KS; ++s) breg[s] = wp[s * 9];
    }
    for (int c = tid; c < CIN; c += 256) {
        s_sc[c] = scale ? scale[b * aff_sb + c] : 1.f;
        s_sh[c] = shift ? shift[b * aff_sb + c] : 0.f;
    }
    __syncthreads();

    for (int blk = wave; blk < kGBlocks; blk += 4) {
        float a[KS];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float4 s4 = *reinterpret_cast<const float4 *>(s_sc + kq * KS + 4 * i);
            const float4 t4 = *reinterpret_cast<const float4 *>(s_sh + kq * KS + 4 * i);
            float4 v = raw[i];
            v.x = s4.x * v.x; v.y = s4.y * v.y; v.z = s4.z * v.z; v.w = s4.w * v.w;
            v.x = t4.x + v.x; v.y = t4.y + v.y; v.z = t4.z + v.z; v.w = t4.w + v.w;
            if (ACT == SIGE_HIP_ACT_SWISH) { v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w); }
            if (ACT == SIGE_HIP_ACT_LEAKY) {
                v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
                v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
            }
            a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
        }
        if (blk + 4 < kGBlocks) {  // (wave-uniform) next block's pixel while this one is in the matrix pipe
            const float4 *src = pixel(blk + 4);
#pragma unroll
            for (int i = 0; i < NV; ++i) raw[i] = src[i];
        }
        floatx16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], breg[s], acc, 0, 0, 0);
        // reg r of lane (kq, j): pixel row m = (r & 3) + 8 * (r >> 2) + 4 * kq, column j
        if (j < NP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * kq) * NP + j] = acc[r];
        }
    }
    __syncthreads();

    const int oy = tid / kGT, ox = tid % kGT;
    const int h = h0 + oy, ww = w0 + ox;
    if (h < H && ww < W) {
        float o[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) o[co] = bias ? bias[co] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ih = h + tap / 3 - 1, iw = ww + tap % 3 - 1;
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                const float *pp = P + ((oy + tap / 3) * kGP + ox + tap % 3) * NP + tap * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) o[co] += pp[co];
            }
        }
        float *op = out + (((size_t)b * H + h) * W + ww) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) op[co] = out_act == SIGE_HIP_ACT_TANH ? tanhf(o[co]) : o[co];
    }
}

}  // namespace sige

using namespace sige;


static int conv3x3_small_cout_impl(const float *x, int B, int C, int H, int W,
                                   const float *scale, int scaleB, int scaleC,
                                   const float *shift, int shiftB, int shiftC, int activation, float slope,
                                   const float *weight, const float *bias, int Cout, int out_act,
                                   float *out, void *stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (!x || !weight || !out) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH && activation != SIGE_HIP_ACT_LEAKY) return SIGE_HIP_EUNSUPPORTED;
    if (out_act != SIGE_HIP_ACT_IDENTITY && out_act != SIGE_HIP_ACT_TANH) return SIGE_HIP_EUNSUPPORTED;
    if (Cout > 4 || C % 4 || (reinterpret_cast<uintptr_t>(x) & 15)) return SIGE_HIP_EUNSUPPORTED;
    if ((scale == nullptr) != (shift == nullptr)) return SIGE_HIP_EUNSUPPORTED;
    int aff_sb = 0;
    if (scale) {
        if (scaleC != C || shiftC != C || scaleB != shiftB || !(scaleB == 1 || scaleB == B)) return SIGE_HIP_EUNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15) return SIGE_HIP_EUNSUPPORTED;
        aff_sb = scaleB > 1 ? C : 0;
    }
    const long blocks = (long)B * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
    if (blocks > 0x7fffffffL) return SIGE_HIP_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    if (Cout <= 3 && (C == 128 || C == 64) && !tuning(SIGE_HIP_TUNE_SMALL_COUT_SCALAR)) {
        const long tiles = (long)B * ((H + kGT - 1) / kGT) * ((W + kGT - 1) / kGT);
        if (tiles > 0x7fffffffL) return SIGE_HIP_EUNSUPPORTED;
#define SIGE_CG(CI, N)                                                                                                 \
    if (activation == SIGE_HIP_ACT_SWISH)                                                                             \
        conv_out_gemm_kernel<CI, N, SIGE_HIP_ACT_SWISH><<<(int)tiles, 256, 0, st>>>(x, B, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act); \
    else if (activation == SIGE_HIP_ACT_LEAKY)                                                                        \
        conv_out_gemm_kernel<CI, N, SIGE_HIP_ACT_LEAKY><<<(int)tiles, 256, 0, st>>>(x, B, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act); \
    else                                                                                                              \
        conv_out_gemm_kernel<CI, N, SIGE_HIP_ACT_IDENTITY><<<(int)tiles, 256, 0, st>>>(x, B, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act);
        if (C == 128) { if (Cout == 1) { SIGE_CG(128, 1) } else if (Cout == 2) { SIGE_CG(128, 2) } else { SIGE_CG(128, 3) } }
        else { if (Cout == 1) { SIGE_CG(64, 1) } else if (Cout == 2) { SIGE_CG(64, 2) } else { SIGE_CG(64, 3) } }
#undef SIGE_CG
        return launch_status();
    }
#define SIGE_CO(N)                                                                                                    \
    if (activation == SIGE_HIP_ACT_SWISH)                                                                             \
        conv_out_nhwc_kernel<N, SIGE_HIP_ACT_SWISH><<<(int)blocks, 256, 0, st>>>(x, B, C, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act); \
    else if (activation == SIGE_HIP_ACT_LEAKY)                                                                        \
        conv_out_nhwc_kernel<N, SIGE_HIP_ACT_LEAKY><<<(int)blocks, 256, 0, st>>>(x, B, C, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act); \
    else                                                                                                              \
        conv_out_nhwc_kernel<N, SIGE_HIP_ACT_IDENTITY><<<(int)blocks, 256, 0, st>>>(x, B, C, H, W, scale, shift, aff_sb, weight, bias, out, slope, out_act);
    switch (Cout) {
        case 1: SIGE_CO(1) break;
        case 2: SIGE_CO(2) break;
        case 3: SIGE_CO(3) break;
        default: SIGE_CO(4) break;
    }
#undef SIGE_CO
    return launch_status();
}

extern "C" int sige_hip_conv3x3_small_cout_nhwc_f32(const float *x, int B, int C, int H, int W,
                                                    const float *scale, int scaleB, int scaleC,
                                                    const float *shift, int shiftB, int shiftC, int activation,
                                                    const float *weight, const float *bias, int Cout,
                                                    float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_conv3x3_small_cout_nhwc_f32, x, B, C, H, W, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, weight, bias, Cout, out, stream);
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    return conv3x3_small_cout_impl(x, B, C, H, W, scale, scaleB, scaleC, shift, shiftB, shiftC, activation, 0.f, weight, bias, Cout,
                                   SIGE_HIP_ACT_IDENTITY, out, stream);
}

// ... with a leaky-ReLU in front and a tanh behind: GauGAN's `tanh(conv_img(leaky_relu(x, 0.2)))` (sige_fused_spade_generator.py:259-260)
extern "C" int sige_hip_conv3x3_small_cout_act_nhwc_f32(const float *x, int B, int C, int H, int W, int activation, float slope,
                                                        const float *weight, const float *bias, int Cout, int out_activation,
                                                        float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_conv3x3_small_cout_act_nhwc_f32, x, B, C, H, W, activation, slope, weight, bias, Cout, out_activation, out, stream);
    return conv3x3_small_cout_impl(x, B, C, H, W, nullptr, 0, 0, nullptr, 0, 0, activation, slope, weight, bias, Cout, out_activation, out,
                                   stream);
}
