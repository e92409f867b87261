// Token-matrix helpers of Stable Diffusion's spatial transformer (stable-diffusion/ldm/modules/sige_attention.py:86-185,
// ldm/modules/attention.py: BasicTransformerBlock): what sits BETWEEN the GEMMs of a block -- residual add + bias + LayerNorm,
// GEGLU's a * gelu(gate), the last residual add -- as one launch each instead of three / two / one torch kernels.
//
//   add_layer_norm   y = x (+ delta + bias[c]);  sum_out = y (optional);  out = (y - mean(y)) * rstd(y) * gamma[c] + beta[c]
//   geglu            out[t, d] = x[t, d] * gelu(x[t, D + d])          (erf form, as F.gelu's default)
//   add_bias         out = x + delta + bias[c]
//
// Tokens [T, C] row-major fp32 (a channels-last tile slab or feature map IS that matrix).  One wavefront per token row for the
// LayerNorm: the row lives in registers (C <= 64 * 32), mean and variance are two wave reductions (two-pass: exact mean first).
#include "common.hpp"

namespace sige {
namespace {

constexpr int kMaxPerLane = 32;  // C <= 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

template <int PER>
__global__ __launch_bounds__(256) void add_layer_norm_kernel(const float *__restrict__ x, const float *__restrict__ delta,
                                                            const float *__restrict__ bias, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, long T, int C, float eps,
                                                            float *__restrict__ sum_out, float *__restrict__ out) {
    kernarg_touch<128>();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float *xr = x + row * C;
    // (branch-free: `if (c < C) { load; add; store }` per element compiled to an exec-masked block with its own s_waitcnt vmcnt(0) --
    //  PER dependent round trips per row.  Every lane loads a valid column -- the last one past the end --, all loads go out
    //  together, the selects follow)
    float v[PER], dv[PER], bv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = min(lane + 64 * i, C - 1);
        v[i] = xr[c];
        dv[i] = delta ? delta[row * C + c] : 0.f;
        bv[i] = bias ? bias[c] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + 64 * i;
        float z = v[i];
        if (delta) z = z + dv[i];
        if (bias) z = z + bv[i];
        if (c < C && sum_out) sum_out[row * C + c] = z;
        v[i] = c < C ? z : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + 64 * i;
        const float d = c < C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float gv[PER], ev[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = min(lane + 64 * i, C - 1);
        gv[i] = gamma[c]; ev[i] = beta[c];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + 64 * i;
        if (c < C) out[row * C + c] = (v[i] - mean) * rstd * gv[i] + ev[i];
    }
}

__global__ __launch_bounds__(256) void geglu_kernel(const float *__restrict__ x, long T, int D4, float *__restrict__ out, long units) {
    for_units<256>(units, [&](auto u) {
        const decltype(u) t = u / D4;
        const int d = (int)(u - t * D4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(x + (size_t)t * 8 * D4 + d);
        const float4 g = *reinterpret_cast<const float4 *>(x + (size_t)t * 8 * D4 + 4 * D4 + d);
        auto gelu = [](float z) { return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f)); };
        *reinterpret_cast<float4 *>(out + (size_t)t * 4 * D4 + d) = make_float4(a.x * gelu(g.x), a.y * gelu(g.y), a.z * gelu(g.z), a.w * gelu(g.w));
    });
}

__global__ __launch_bounds__(256) void add_bias_kernel(const float *__restrict__ x, const float *__restrict__ delta,
                                                      const float *__restrict__ bias, int C4, float *__restrict__ out, long units) {
    for_units<256>(units, [&](auto u) {
        const int c = (int)(u % C4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(x + (size_t)u * 4), b = *reinterpret_cast<const float4 *>(delta + (size_t)u * 4);
        float4 r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4 *>(bias + c);
            r = make_float4(r.x + bb.x, r.y + bb.y, r.z + bb.z, r.w + bb.w);
        }
        *reinterpret_cast<float4 *>(out + (size_t)u * 4) = r;
    });
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int grid_of(long units) {
    const long b = (units + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace
}  // namespace sige

using namespace sige;

extern "C" int sige_hip_add_layer_norm_tokens_f32(const float *x, const float *delta, const float *bias, const float *gamma,
                                                  const float *beta, int64_t T, int C, float eps, float *sum_out, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_add_layer_norm_tokens_f32, x, delta, bias, gamma, beta, T, C, eps, sum_out, out, stream);
    if (T < 0 || C <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !gamma || !beta || !out) return SIGE_HIP_EINVAL;
    if (C > 64 * kMaxPerLane || T > 0x7fffffffL * 4) return SIGE_HIP_EUNSUPPORTED;
    const int per = (C + 63) / 64;
    const dim3 grid((unsigned)((T + 3) / 4));
    hipStream_t st = as_stream(stream);
#define SIGE_LN(P) add_layer_norm_kernel<P><<<grid, 256, 0, st>>>(x, delta, bias, gamma, beta, (long)T, C, eps, sum_out, out)
    if (per <= 5) SIGE_LN(5);
    else if (per <= 10) SIGE_LN(10);
    else if (per <= 20) SIGE_LN(20);
    else SIGE_LN(32);
#undef SIGE_LN
    return launch_status();
}

extern "C" int sige_hip_geglu_tokens_f32(const float *x, int64_t T, int D, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_geglu_tokens_f32, x, T, D, out, stream);
    if (T < 0 || D <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !out) return SIGE_HIP_EINVAL;
    if (D % 4 || !al16(x) || !al16(out)) return SIGE_HIP_EUNSUPPORTED;
    const long units = (long)T * (D / 4);
    geglu_kernel<<<grid_of(units), 256, 0, as_stream(stream)>>>(x, (long)T, D / 4, out, units);
    return launch_status();
}

extern "C" int sige_hip_add_bias_tokens_f32(const float *x, const float *delta, const float *bias, int64_t T, int C, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_add_bias_tokens_f32, x, delta, bias, T, C, out, stream);
    if (T < 0 || C <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !delta || !out) return SIGE_HIP_EINVAL;
    if (C % 4 || !al16(x) || !al16(delta) || !al16(out) || !al16(bias)) return SIGE_HIP_EUNSUPPORTED;
    const long units = (long)T * (C / 4);
    add_bias_kernel<<<grid_of(units), 256, 0, as_stream(stream)>>>(x, delta, bias, C / 4, out, units);
    return launch_status();
}
