// The U-Net's single-head spatial self-attention in ONE launch (channels-last): out = softmax(scale * q k^T) v for
// qkv [B, HW, 3C] (q | k | v on the channel axis) -> out [B, HW, C].
//
// diffusion/models/ddpm_arch/sige_fused_unet.py:186-199 (-> unet.py AttnBlock): 16 x 16 = 256 tokens (and 8 x 8 in the middle
// block), C = 512, six blocks per forward.  Rounds 1-3 ran it as two launches (csrc/nhwc_ops.hip: a score kernel writing
// S [HW, HW] to a workspace, an apply kernel reading it back): 12 of the 102 launches of a sparse forward, 75 us.  A single
// workgroup per 16 queries cannot do it in one launch at speed -- 16 x 256 x 512 multiply-adds are 6.8 us of ONE CU's exact-fp32
// matrix rate --, and the apply kernel's decomposition (16 queries x 64 channels per workgroup) would recompute every score
// row 8 times.  Here the KEYS are split across workgroups (flash-decoding): workgroup = 16 queries x 64 keys,
//
//   scores   the 4 waves split the C channels of the contraction (each: its C / 4 channels for all 4 key tiles; one 16-byte load
//            feeds 4 k-steps of v_mfma_f32_16x16x4_f32), partial tiles summed through LDS in wave order;
//   softmax  of the 64-key slice: m = row max, p = exp2((s - m) log2 e), l = sum p (16 lanes per row) -> P tile in LDS;
//   values   O_slice[16, C] = P V: the waves split the C output channels, V[key][c] straight from global memory (16 lanes = 64
//            contiguous bytes), issued before the score MFMAs so that they arrive under them;
//   finish   one slice (HW <= 64): out = O / l.  Several: (O, m, l) go to the workspace with device-coherent stores, the block's
//            ticket is taken, and the LAST slice to finish combines all of them in slice order -- out = sum_s O_s 2^(m_s - m) /
//            sum_s l_s 2^(m_s - m) -- the mechanism of the K-split finish of the conv kernels (csrc/conv_mfma.hpp), same
//            tickets (split_tickets), no second launch, the summation order fixed (bit-reproducible).
//
// Exact fp32 products, no score recompute, HW / 16 x HW / 64 workgroups (64 for the 16 x 16 level).
#include "common.hpp"

namespace sige {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void st_coherent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coherent(const float *p) { return __hip_atomic_load(const_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 ld_coherent4(const float *p) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<float *>(p));
    const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                       __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32)));
}

constexpr int kSliceKeys = 64;   // keys per workgroup (4 tiles of 16)
constexpr int kMaxSlices = 16;   // HW <= 1024

}  // namespace

// floats of workspace per (batch, query block, slice): O [16][C], then m [16], l [16]
__host__ __device__ inline size_t attn_fused_slot(int C) { return (size_t)16 * C + 32; }

template <int NT>  // C = 64 * NT: every wave owns 16 * NT channels
__global__ __launch_bounds__(256) void attn_fused_nhwc_kernel(const float *__restrict__ qkv, int HW, float scale_log2e,
                                                             float *__restrict__ ws, int32_t *__restrict__ counters,
                                                             float *__restrict__ out) {
    constexpr int C = 64 * NT, C3 = 3 * C, CW = 16 * NT;
    constexpr int RS = kSliceKeys + 4;
    __shared__ __attribute__((aligned(16))) float red[4][16][RS];  // per-wave partial scores [query][key]
    __shared__ __attribute__((aligned(16))) float P[16][RS];
    __shared__ float m_s[16], l_s[16];
    __shared__ float wt[kMaxSlices][16];
    __shared__ float inv_s[16];
    __shared__ int ticket_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const int slice = blockIdx.x, KS = gridDim.x, qb = blockIdx.y, b = blockIdx.z;
    const int i0 = qb * 16, j0 = slice * kSliceKeys;
    const float *base = qkv + (size_t)b * HW * C3;

    // ---- operands of the scores: this wave's CW channels of Q (16 queries) and of K (4 key tiles) ----
    float4 q4[NT], k4[4][NT];
    {
        const float *qa = base + (size_t)(i0 + n) * C3 + wave * CW + kq * 4;
#pragma unroll
        for (int u = 0; u < NT; ++u) q4[u] = *reinterpret_cast<const float4 *>(qa + 16 * u);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = min(j0 + 16 * t + n, HW - 1);  // (keys past the end: a valid address, masked in the softmax)
            const float *kb = base + (size_t)key * C3 + C + wave * CW + kq * 4;
#pragma unroll
            for (int u = 0; u < NT; ++u) k4[t][u] = *reinterpret_cast<const float4 *>(kb + 16 * u);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- partial scores; the values this wave will need afterwards (its CW output channels, the slice's 64 keys) are issued a
    // quarter per key tile, BEHIND that tile's MFMAs: they arrive under the score phase, and the wait for the next tile's K
    // operands (issued before them) does not wait for them.  (The scheduling barriers keep this order: left to itself the
    // compiler sinks every load to its use, and the memory latency is paid 32 times.) ----
    float bv[16][NT];
    const float *vb = base + 2 * C + wave * CW + n;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[u].x, k4[t][u].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[u].y, k4[t][u].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[u].z, k4[t][u].z, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4[u].w, k4[t][u].w, a1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 4 * t; tt < 4 * t + 4; ++tt) {
            const int key = min(j0 + 4 * tt + kq, HW - 1);
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) bv[tt][nn] = vb[(size_t)key * C3 + 16 * nn];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][4 * kq + r][16 * t + n] = a0[r] + a1[r];
    }
    __syncthreads();
    // ---- softmax of the slice: 16 lanes per query row, 4 keys per lane ----
    {
        const int row = tid >> 4, l16 = tid & 15;
        const float4 r0 = *reinterpret_cast<const float4 *>(&red[0][row][4 * l16]);
        const float4 r1 = *reinterpret_cast<const float4 *>(&red[1][row][4 * l16]);
        const float4 r2 = *reinterpret_cast<const float4 *>(&red[2][row][4 * l16]);
        const float4 r3 = *reinterpret_cast<const float4 *>(&red[3][row][4 * l16]);
        float s[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                      (r0.w + r1.w) + (r2.w + r3.w)};
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] = (j0 + 4 * l16 + e < HW) ? s[e] * scale_log2e : -INFINITY;
            m = fmaxf(m, s[e]);
        }
        m = row16_max(m);  // (finite: a slice has at least one live key)
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s[e] = __builtin_amdgcn_exp2f(s[e] - m);
            sum += s[e];
        }
        sum = row16_sum(sum);
        *reinterpret_cast<float4 *>(&P[row][4 * l16]) = make_float4(s[0], s[1], s[2], s[3]);
        if (l16 == 0) { m_s[row] = m; l_s[row] = sum; }
    }
    __syncthreads();
    // ---- O_slice = P V: A[m = query n][k = key 4t + kq] from LDS, B[k][column] = the values loaded above ----
    f32x4 o[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) o[nn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float a = P[n][4 * t + kq];
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) o[nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[t][nn], o[nn], 0, 0, 0);
    }
    // o[nn][r] = O[query 4kq + r][channel wave * CW + 16 nn + n]
    if (KS == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            const float inv = 1.0f / l_s[row];
            float *orow = out + ((size_t)b * HW + i0 + row) * C + wave * CW + n;
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) orow[16 * nn] = o[nn][r] * inv;
        }
        return;
    }
    const size_t slot = attn_fused_slot(C);
    float *const blk = ws + ((size_t)(b * gridDim.y + qb) * KS) * slot;  // the KS slots of this (batch, query block)
    float *const mine = blk + (size_t)slice * slot;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *prow = mine + (size_t)(4 * kq + r) * C + wave * CW + n;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) st_coherent(prow + 16 * nn, o[nn][r]);
    }
    if (tid < 16) st_coherent(mine + 16 * C + tid, m_s[tid]);
    else if (tid < 32) st_coherent(mine + 16 * C + tid, l_s[tid - 16]);
    // device-coherent stores are at the coherence point once complete (vmcnt 0); then the block's ticket (csrc/conv_mfma.hpp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int32_t *const cnt = counters + b * gridDim.y + qb;
    if (tid == 0) ticket_s = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket_s != KS - 1) return;
    if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    // ---- the last slice to arrive: weights of the slices per row, then the combination in slice order ----
    if (tid < 16) {
        float m = -INFINITY;
        for (int s = 0; s < KS; ++s) m = fmaxf(m, ld_coherent(blk + (size_t)s * slot + 16 * C + tid));
        float L = 0.f;
        for (int s = 0; s < KS; ++s) {
            const float w = __builtin_amdgcn_exp2f(ld_coherent(blk + (size_t)s * slot + 16 * C + tid) - m);
            wt[s][tid] = w;
            L += w * ld_coherent(blk + (size_t)s * slot + 16 * C + 16 + tid);
        }
        inv_s[tid] = 1.0f / L;
    }
    __syncthreads();
    constexpr int UPR = C / 4;                 // float4 units per row
    constexpr int UNITS = 16 * UPR / 256;      // per thread (= NT)
#pragma unroll
    for (int k = 0; k < UNITS; ++k) {
        const int e = tid + 256 * k;
        const int row = e / UPR, c = (e - row * UPR) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < KS; ++s) {
            const float4 pv = ld_coherent4(blk + (size_t)s * slot + (size_t)row * C + c);
            const float w = wt[s][row];
            acc.x += w * pv.x; acc.y += w * pv.y; acc.z += w * pv.z; acc.w += w * pv.w;
        }
        const float inv = inv_s[row];
        *reinterpret_cast<float4 *>(out + ((size_t)b * HW + i0 + row) * C + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

}  // namespace sige

using namespace sige;

static int attn_fused_nt(int C) { return (C == 64 || C == 128 || C == 256 || C == 512) ? C / 64 : 0; }

extern "C" size_t sige_hip_attention_fused_workspace(int B, int C, int HW) {
    if (B <= 0 || HW <= 0 || HW % 16 || !attn_fused_nt(C) || HW > kSliceKeys * kMaxSlices) return 0;
    const int KS = (HW + kSliceKeys - 1) / kSliceKeys;
    return KS == 1 ? 4 : (size_t)B * (HW / 16) * KS * attn_fused_slot(C);  // (never 0 for a supported shape: 0 says "unsupported")
}

extern "C" int sige_hip_attention_fused_nhwc_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                                                 float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_attention_fused_nhwc_f32, qkv, B, C, HW, scale, workspace, out, stream);
    if (B <= 0 || C <= 0 || HW <= 0) return SIGE_HIP_EINVAL;
    if (!qkv || !workspace || !out) return SIGE_HIP_EINVAL;
    const int nt = attn_fused_nt(C);
    if (!nt || HW % 16 || HW > kSliceKeys * kMaxSlices || B > 65535) return SIGE_HIP_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(out)) & 15) return SIGE_HIP_EUNSUPPORTED;
    const int KS = (HW + kSliceKeys - 1) / kSliceKeys;
    hipStream_t st = as_stream(stream);
    int32_t *counters = nullptr;
    if (KS > 1) {
        counters = split_tickets(st, (long)B * (HW / 16));
        if (!counters) return SIGE_HIP_EUNSUPPORTED;  // (no tickets: the caller runs the two-launch form)
    }
    const dim3 grid(KS, HW / 16, B);
    const float sl = scale * 1.44269504088896341f;
    switch (nt) {
        case 1: attn_fused_nhwc_kernel<1><<<grid, 256, 0, st>>>(qkv, HW, sl, workspace, counters, out); break;
        case 2: attn_fused_nhwc_kernel<2><<<grid, 256, 0, st>>>(qkv, HW, sl, workspace, counters, out); break;
        case 4: attn_fused_nhwc_kernel<4><<<grid, 256, 0, st>>>(qkv, HW, sl, workspace, counters, out); break;
        default: attn_fused_nhwc_kernel<8><<<grid, 256, 0, st>>>(qkv, HW, sl, workspace, counters, out); break;
    }
    return launch_status();
}
