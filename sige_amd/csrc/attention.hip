// Single-head spatial self-attention of the U-Net's AttnBlock for gfx950.
//
// The reference's attention block (diffusion/models/ddpm_arch/sige_fused_unet.py:
// 186-199 -> unet.py AttnBlock) runs at the dense low resolutions (16x16 = 256
// tokens, C = 512 for DDPM-256) as  bmm(q^T, k) -> softmax -> bmm(v, attn^T)  on
// NCHW tensors.  On MI355X the generic GEMM library picks a 256x256 macro tile
// for the 256x256x512 score product -- ONE workgroup, 119 us per block, 13 % of a
// whole sparse forward.  Here it is two launches that cover the chip:
//
//   attn_scores_kernel   S[i][j] = scale * sum_c q[c][i] k[c][j]
//       one 16x16 score tile per workgroup, the 4 waves split the channels,
//       operands straight from global memory (64-byte coalesced segments per
//       lane group) into v_mfma_f32_16x16x4_f32, LDS reduction.
//   attn_apply_kernel    out[c][i] = sum_j v[c][j] softmax_j(S[i][.])[j]
//       workgroup = 16 queries x 64 channels: the 16 score rows are softmax-ed
//       once into LDS (exact expf, fp32), v tiles are staged through LDS with
//       16-byte loads (the contraction runs along v's contiguous axis, so the
//       MFMA operand order needs the transpose), one 16x16 output tile per wave.
//
// Layout: qkv [B, 3C, HW] (q, k, v stacked on the channel axis, each [C][HW]),
// scores workspace [B, HW, HW], out [B, C, HW]; fp32 throughout.
#include "common.hpp"

namespace sige {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void attn_scores_kernel(const float *__restrict__ qkv, int C, int HW, float scale,
                                                          float *__restrict__ S) {
    __shared__ float red[4][16][20];
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, n = lane & 15;
    const float *q = qkv + (size_t)b * 3 * C * HW;
    const float *k = q + (size_t)C * HW;
    // wave w takes channels [w*C/4, (w+1)*C/4)
    const int cw = C / 4;
    const float *qa = q + (size_t)(wave * cw + kq) * HW + i0 + n;  // A[m = query][k = channel]
    const float *kb = k + (size_t)(wave * cw + kq) * HW + j0 + n;  // B[k = channel][n = key]
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= cw; s += 8) {
        const float a0 = qa[(size_t)s * HW], b0 = kb[(size_t)s * HW];
        const float a1 = qa[(size_t)(s + 4) * HW], b1 = kb[(size_t)(s + 4) * HW];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    for (; s < cw; s += 4) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[(size_t)s * HW], kb[(size_t)s * HW], acc0, 0, 0, 0);
    // D[row = query 4*kq + r][col = key n]
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][n] = acc0[r] + acc1[r];
    __syncthreads();
    const int qi = tid >> 4, kj = tid & 15;
    const float v = (red[0][qi][kj] + red[1][qi][kj]) + (red[2][qi][kj] + red[3][qi][kj]);
    S[((size_t)b * HW + i0 + qi) * HW + j0 + kj] = v * scale;
}

constexpr int kAttnKeys = 256;  // keys per LDS chunk of v
constexpr int kAttnCh = 64;     // channels per workgroup (4 waves x 16)

// dynamic LDS: P [16][HW + 4] | V chunk [64][kAttnKeys + 4]
__global__ __launch_bounds__(256) void attn_apply_kernel(const float *__restrict__ qkv, const float *__restrict__ S,
                                                         int C, int HW, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int PS = HW + 4, VS = kAttnKeys + 4;
    float *P = lds;
    float *V = lds + 16 * PS;
    const int b = blockIdx.z;
    const int i0 = blockIdx.y * 16, c0 = blockIdx.x * kAttnCh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float *v = qkv + ((size_t)b * 3 + 2) * C * HW;

    // ---- softmax of the 16 score rows: 16 lanes per row ----
    {
        const int row = tid >> 4, l16 = tid & 15;
        const float *srow = S + ((size_t)b * HW + i0 + row) * HW;
        float m = -INFINITY;
        for (int j = l16 * 4; j < HW; j += 64) {
            const float4 t = *reinterpret_cast<const float4 *>(srow + j);
            *reinterpret_cast<float4 *>(P + row * PS + j) = t;
            m = fmaxf(fmaxf(m, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
        float sum = 0.f;
        for (int j = l16 * 4; j < HW; j += 64) {
            float4 t = *reinterpret_cast<float4 *>(P + row * PS + j);
            t.x = expf(t.x - m); t.y = expf(t.y - m); t.z = expf(t.z - m); t.w = expf(t.w - m);
            sum += (t.x + t.y) + (t.z + t.w);
            *reinterpret_cast<float4 *>(P + row * PS + j) = t;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = 1.0f / sum;
        for (int j = l16 * 4; j < HW; j += 64) {
            float4 t = *reinterpret_cast<float4 *>(P + row * PS + j);
            t.x *= inv; t.y *= inv; t.z *= inv; t.w *= inv;
            *reinterpret_cast<float4 *>(P + row * PS + j) = t;
        }
    }

    // ---- out[c][i] = sum_j v[c][j] P[i][j]:  A[m = channel][k = key] from V, B[k = key][n = query] from P ----
    const int kq = lane >> 4, n = lane & 15;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int jc = 0; jc < HW; jc += kAttnKeys) {
        const int nk = min(kAttnKeys, HW - jc);  // multiple of 16
        __syncthreads();                         // P complete / previous V chunk consumed
        const int n4 = nk / 4;
        for (int u = tid; u < kAttnCh * n4; u += 256) {
            const int c = u / n4, j4 = (u - c * n4) * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + c < C) t = *reinterpret_cast<const float4 *>(v + (size_t)(c0 + c) * HW + jc + j4);
            *reinterpret_cast<float4 *>(V + c * VS + j4) = t;
        }
        __syncthreads();
        const float *va = V + (wave * 16 + n) * VS + kq;
        const float *pb = P + n * PS + jc + kq;
        int s = 0;
        for (; s + 8 <= nk; s += 8) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], pb[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s + 4], pb[s + 4], acc1, 0, 0, 0);
        }
        for (; s < nk; s += 4) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s], pb[s], acc0, 0, 0, 0);
    }
    // D[row = channel 4*kq + r][col = query n]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + wave * 16 + 4 * kq + r;
        if (c < C) out[((size_t)b * C + c) * HW + i0 + n] = acc0[r] + acc1[r];
    }
}

}  // namespace sige

using namespace sige;

extern "C" size_t sige_hip_attention_workspace(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0) return 0;
    return (size_t)B * HW * HW;
}

extern "C" int sige_hip_attention_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                                      float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_attention_f32, qkv, B, C, HW, scale, workspace, out, stream);
    if (B <= 0 || C <= 0 || HW <= 0) return SIGE_HIP_EINVAL;
    if (!qkv || !workspace || !out) return SIGE_HIP_EINVAL;
    // 16x16 tiles, 4-way channel split in 4-channel MFMA steps, 16-byte loads; P row + v chunk in 160 KiB of LDS
    if (HW % 16 || C % 16 || HW > 4096 || B > 65535) return SIGE_HIP_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(workspace)) & 15) return SIGE_HIP_EINVAL;
    const size_t lds = ((size_t)16 * (HW + 4) + (size_t)kAttnCh * (kAttnKeys + 4)) * sizeof(float);
    if (lds > 160 * 1024) return SIGE_HIP_EUNSUPPORTED;
    hipStream_t st = as_stream(stream);
    attn_scores_kernel<<<dim3(HW / 16, HW / 16, B), 256, 0, st>>>(qkv, C, HW, scale, workspace);
    if (lds > 64 * 1024) {
        static bool raised = false;
        if (!raised) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(attn_apply_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return SIGE_HIP_ELAUNCH;
            raised = true;
        }
    }
    attn_apply_kernel<<<dim3(ceil_div(C, kAttnCh), HW / 16, B), 256, lds, st>>>(qkv, workspace, C, HW, out);
    return launch_status(2);
}
