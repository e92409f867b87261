// GroupNorm statistics -> per-channel affine, for gfx950.
//
// The cached-affine contract of SIGE (diffusion/models/common.py:37-57): the
// full pass turns GroupNorm into per-channel (scale, shift) so that the sparse
// pass can fuse "normalise + SiLU" into a gather.  PyTorch's generic
// RowwiseMoments kernel uses ONE workgroup per (batch, group) -- 32 workgroups on
// a 256-CU chip, 171 us for the U-Net's output norm [1,128,256,256].  Here each
// (batch, group) is split over many workgroups (16-byte streaming loads, wave
// shuffle + LDS reduction, partial sums in a workspace), and a second tiny
// launch combines the partials in fp64 and emits scale/shift.
#include "common.hpp"

namespace sige {

constexpr int kGNThreads = 256;

__global__ __launch_bounds__(kGNThreads) void gn_partial_kernel(const float *__restrict__ x, size_t group_elems,
                                                                int splits, float *__restrict__ ws) {
    // grid: (splits, B*groups); each block reduces a contiguous slice of one group
    const int bg = blockIdx.y, sp = blockIdx.x;
    const size_t per = (((group_elems + 3) / 4 + splits - 1) / splits) * 4;
    const size_t lo = (size_t)sp * per, hi = min(group_elems, lo + per);
    const float *g = x + (size_t)bg * group_elems;
    float s = 0.f, ss = 0.f;
    if (((reinterpret_cast<uintptr_t>(g) | (lo * 4)) & 15) == 0) {
        const size_t n4 = hi > lo ? (hi - lo) / 4 : 0;
        const float4 *g4 = reinterpret_cast<const float4 *>(g + lo);
        for (size_t i = threadIdx.x; i < n4; i += kGNThreads) {
            const float4 v = g4[i];
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        for (size_t i = lo + n4 * 4 + threadIdx.x; i < hi; i += kGNThreads) { const float v = g[i]; s += v; ss += v * v; }
    } else {
        for (size_t i = lo + threadIdx.x; i < hi; i += kGNThreads) { const float v = g[i]; s += v; ss += v * v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); ss += __shfl_down(ss, o); }
    __shared__ float sh[2][kGNThreads / kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wave] = s; sh[1][wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < kGNThreads / kWave; ++w) { a += sh[0][w]; b += sh[1][w]; }
        ws[((size_t)bg * splits + sp) * 2] = a;
        ws[((size_t)bg * splits + sp) * 2 + 1] = b;
    }
}

// one 64-lane wave per (batch, group): lanes add the split partials in fp64, a shuffle tree
// combines them, then lane k writes channel k of the group
__global__ __launch_bounds__(64) void gn_finish_kernel(const float *__restrict__ ws, int splits, int B, int C, int groups,
                                                      double group_elems, float eps, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ scale,
                                                      float *__restrict__ shift, const float *__restrict__ cbias = nullptr) {
    kernarg_touch<128>();
    const int bg = blockIdx.x;  // b*groups + g
    const int b = bg / groups, g = bg - b * groups;
    const int lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    const float *p = ws + ((size_t)bg * splits) * 2;
    for (int k0 = lane; k0 < splits; k0 += 64 * 8) {  // 8 partials in flight per lane, added in split order
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)  // (clamped address, not a conditional load: a branch per load would serialise them)
            v[u] = *reinterpret_cast<const float2 *>(p + 2 * min(k0 + 64 * u, splits - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 64 * u < splits) { s += v[u].x; ss += v[u].y; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    const double mean = s / group_elems;
    double var = ss / group_elems - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int cg = C / groups;
    for (int k = lane; k < cg; k += 64) {
        const int c = g * cg + k;
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = ga * rstd;
        scale[b * C + c] = sc;
        // (cbias: the statistics are those of x + cbias[c]; the affine is returned for x itself -- GN(x + cbias) == x * sc + shift)
        shift[b * C + c] = be - (float)mean * sc + (cbias ? cbias[c] * sc : 0.0f);
    }
}

static int gn_splits(size_t group_elems) {
    // ~64 KiB of input per workgroup, at most 64 splits per group
    size_t s = (group_elems + 16383) / 16384;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}

// ------------------------------------------------------- GroupNorm affine ----
// x [B,H,W,C]: per (batch, group) sum / sum of squares over H*W pixels x (C/groups) channels
__global__ __launch_bounds__(kGNThreads) void gn_partial_nhwc_kernel(const float *__restrict__ x, int C, int HW, int groups, int splits,
                                                            float *__restrict__ ws, const float *__restrict__ cbias = nullptr) {
    // grid: (splits, B); block: each lane owns channel quad (tid % C4) of pixels (tid / C4) + k * (kGNThreads / C4) in its slice.
    // Requires C4 <= kGNThreads and kGNThreads % C4 == 0 (host side) so that a lane's channel quad is fixed.
    const int b = blockIdx.y, sp = blockIdx.x;
    const int C4 = C / 4, ppb = kGNThreads / C4;
    const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int per = (HW + splits - 1) / splits;
    const int lo = sp * per, hi = min(HW, lo + per);
    float s = 0.f, ss = 0.f;
    const float *xb = x + (size_t)b * HW * C;
    // optional per-channel bias added before the statistics (the timestep embedding of a residual block: sige_fused_unet.py:116-117)
    const float4 cb = cbias ? *reinterpret_cast<const float4 *>(cbias + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    // 16 loads in flight per lane (a rolled loop pays the memory latency once per iteration: the [1,128,256,256]
    // output norm took 19 us as 32 dependent round trips per lane; the host sizes a slice to one round)
    constexpr int U = 16;
    for (int p0 = lo + pl; p0 < hi; p0 += ppb * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (branch-free: `p < hi ? load : 0` compiled to an exec-masked branch around every load with s_waitcnt vmcnt(0) inside it --
            //  16 DEPENDENT round trips per lane, not 16 loads in flight.  A slot past the slice reads the slice's last pixel and
            //  is zeroed by the select below)
            const int p = min(p0 + u * ppb, hi - 1);
            v[u] = *reinterpret_cast<const float4 *>(xb + (size_t)p * C + cq * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = p0 + u * ppb < hi;
            v[u].x = ok ? v[u].x + cb.x : 0.f; v[u].y = ok ? v[u].y + cb.y : 0.f;
            v[u].z = ok ? v[u].z + cb.z : 0.f; v[u].w = ok ? v[u].w + cb.w : 0.f;
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
            ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
        }
    }
    // combine the lanes of one group: channels per group cg = C / groups; quads per group = cg / 4 (cg % 4 == 0: host side)
    __shared__ float sh[2][kGNThreads];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = ss;
    __syncthreads();
    const int qpg = C4 / groups;
    if (threadIdx.x < groups) {
        float a0 = 0.f, a1 = 0.f;
        for (int pp = 0; pp < ppb; ++pp)
            for (int k = 0; k < qpg; ++k) { a0 += sh[0][pp * C4 + threadIdx.x * qpg + k]; a1 += sh[1][pp * C4 + threadIdx.x * qpg + k]; }
        float *o = ws + (((size_t)b * groups + threadIdx.x) * splits + sp) * 2;
        o[0] = a0; o[1] = a1;
    }
}


// ---- GroupNorm affine from per-channel statistics (round 3) -------------------------------------------------------------
// The dense-layer conv (conv_wide.hpp) can leave, per 8x8 pixel block and output channel, the (sum, sum of squares) of what it
// wrote: the GroupNorm of that tensor then needs no pass over the tensor at all -- one launch over ~1 MB of partial sums
// instead of a 33 MB read + a finish launch.  Two parts = the statistics of a torch.cat([a, b], 1) that never has to exist.  Per-channel sums also carry a per-channel bias exactly:
//   sum(x + t) = S1 + n t,  sum((x + t)^2) = S2 + 2 t S1 + n t^2   (the timestep embedding added before norm2).
// A nearest-neighbour upsampling of the producer's output has the same mean / variance: `count` is the producer's pixel count.
struct StatsPart { const float2 *st; int tiles, C; double count; };

__global__ __launch_bounds__(256) void gn_from_stats_kernel(StatsPart p1, StatsPart p2, int groups, float eps,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ cbias, float *__restrict__ scale,
                                                            float *__restrict__ shift) {
    kernarg_touch<128>();
    const int C = p1.C + p2.C, cg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int c0 = g * cg;
    // lane = (channel k of the group, tile slice): consecutive lanes read consecutive float2.  A group may straddle the two
    // parts (768 = 512 + 256 channels in groups of 24): every lane picks its channel's part.
    const int tstep = 256 / cg;                                // tile slices (lanes past tstep * cg idle: cg need not divide 256)
    const bool live = (int)threadIdx.x < tstep * cg;
    const int k = threadIdx.x % cg;
    const bool second = c0 + k >= p1.C;
    const StatsPart p = second ? p2 : p1;
    const int t0 = live ? threadIdx.x / cg : p.tiles;
    const float2 *base = p.st + ((size_t)b * p.tiles) * p.C + (second ? c0 + k - p1.C : c0 + k);
    double s1 = 0.0, s2 = 0.0;
    for (int t = t0; t < p.tiles; t += tstep * 8) {  // 8 partials in flight per lane, added in tile order
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)min(t + u * tstep, p.tiles - 1) * p.C];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t + u * tstep < p.tiles) { s1 += v[u].x; s2 += v[u].y; }
    }
    __shared__ double sh1[256], sh2[256];
    __shared__ double sh_mean, sh_rstd;
    sh1[threadIdx.x] = s1; sh2[threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < cg) {
        // per-channel MEANS of x and x^2 (fixed order), with the channel's bias folded in; means, not sums: the two parts may
        // have been summed at different resolutions (a producer in front of a fused nearest upsampling)
        double a1 = 0.0, a2 = 0.0;
        for (int j = 0; j < tstep; ++j) { a1 += sh1[j * cg + threadIdx.x]; a2 += sh2[j * cg + threadIdx.x]; }
        a1 /= p.count; a2 /= p.count;
        if (cbias) {
            const double t = cbias[c0 + threadIdx.x];
            a2 += 2.0 * t * a1 + t * t;
            a1 += t;
        }
        sh1[threadIdx.x] = a1; sh2[threadIdx.x] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a1 = 0.0, a2 = 0.0;
        for (int j = 0; j < cg; ++j) { a1 += sh1[j]; a2 += sh2[j]; }
        const double mean = a1 / cg;
        double var = a2 / cg - mean * mean;
        if (var < 0.0) var = 0.0;
        sh_mean = mean; sh_rstd = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    if ((int)threadIdx.x < cg) {
        const int c = c0 + threadIdx.x;
        const float rstd = (float)sh_rstd;
        const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float sc = ga * rstd;
        scale[b * C + c] = sc;
        // (cbias: the statistics are those of x + cbias[c]; the affine is returned for x itself -- GN(x + cbias) == x * sc + shift)
        shift[b * C + c] = be - (float)sh_mean * sc + (cbias ? cbias[c] * sc : 0.0f);
    }
}

// Per-channel statistics of a channels-last tensor in the layout the dense-layer conv leaves them (above): for a tensor whose
// producer left none (the first conv, the tile kernels).  One pass; "pixel block" = kStatPix consecutive pixels.
constexpr int kStatPix = 128;

__global__ __launch_bounds__(kGNThreads) void channel_stats_nhwc_kernel(const float *__restrict__ x, int C, int HW, int tiles,
                                                                      float2 *__restrict__ st) {
    // grid (tiles, B); lane = (pixel lane pl, channel quad cq); C4 <= 256 lanes per pixel, ppb pixel lanes
    const int b = blockIdx.y, tile = blockIdx.x;
    const int C4 = C / 4, ppb = kGNThreads / C4;
    const int cq = threadIdx.x % C4, pl = threadIdx.x / C4;
    const bool live = pl < ppb;
    const int lo = tile * kStatPix, hi = min(HW, lo + kStatPix);
    const float *xb = x + (size_t)b * HW * C;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 8;
    if (live)
        for (int p0 = lo + pl; p0 < hi; p0 += ppb * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * ppb;
                v[u] = p < hi ? *reinterpret_cast<const float4 *>(xb + (size_t)p * C + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
                q.x += v[u].x * v[u].x; q.y += v[u].y * v[u].y; q.z += v[u].z * v[u].z; q.w += v[u].w * v[u].w;
            }
        }
    __shared__ float4 sh[2][kGNThreads];
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
    __syncthreads();
    if ((int)threadIdx.x < C4) {  // (fixed order over the pixel lanes)
        float4 a = sh[0][threadIdx.x], c = sh[1][threadIdx.x];
        for (int pp = 1; pp < ppb; ++pp) {
            const float4 a2 = sh[0][pp * C4 + threadIdx.x], c2 = sh[1][pp * C4 + threadIdx.x];
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
            c.x += c2.x; c.y += c2.y; c.z += c2.z; c.w += c2.w;
        }
        float2 *o = st + ((size_t)b * tiles + tile) * C + threadIdx.x * 4;
        o[0] = make_float2(a.x, c.x); o[1] = make_float2(a.y, c.y); o[2] = make_float2(a.z, c.z); o[3] = make_float2(a.w, c.w);
    }
}


}  // namespace sige

using namespace sige;

extern "C" size_t sige_hip_group_norm_affine_workspace(int B, int C, int H, int W, int groups) {
    if (B <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
    const size_t ge = (size_t)(C / groups) * H * W;
    return (size_t)B * groups * gn_splits(ge) * 2;
}

extern "C" int sige_hip_group_norm_affine_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                              const float *gamma, const float *beta, float *workspace,
                                              float *scale, float *shift, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_group_norm_affine_f32, x, B, C, H, W, groups, eps, gamma, beta, workspace, scale, shift, stream);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups) return SIGE_HIP_EINVAL;
    if (!x || !workspace || !scale || !shift) return SIGE_HIP_EINVAL;
    if ((long)B * groups > 65535) return SIGE_HIP_EUNSUPPORTED;
    const size_t ge = (size_t)(C / groups) * H * W;
    const int splits = gn_splits(ge);
    hipStream_t st = as_stream(stream);
    gn_partial_kernel<<<dim3(splits, B * groups), kGNThreads, 0, st>>>(x, ge, splits, workspace);
    gn_finish_kernel<<<B * groups, 64, 0, st>>>(workspace, splits, B, C, groups, (double)ge, eps, gamma, beta,
                                                          scale, shift);
    return launch_status(2);
}

// ---- channels-last form: x [B,H,W,C] ----
static int gn_nhwc_splits(int HW) {
    int s = (HW + 127) / 128;  // 128 pixels per workgroup = one round of 16 loads per lane at C = 128
    return s < 1 ? 1 : (s > 512 ? 512 : s);
}

extern "C" size_t sige_hip_group_norm_affine_nhwc_workspace(int B, int C, int H, int W, int groups) {
    if (B <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
    return (size_t)B * groups * gn_nhwc_splits(H * W) * 2;
}

static int group_norm_affine_nhwc(const float *x, int B, int C, int H, int W, int groups, float eps, const float *gamma,
                                  const float *beta, const float *cbias, float *workspace, float *scale, float *shift, void *stream);

extern "C" int sige_hip_group_norm_affine_nhwc_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                                   const float *gamma, const float *beta, float *workspace,
                                                   float *scale, float *shift, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_group_norm_affine_nhwc_f32, x, B, C, H, W, groups, eps, gamma, beta, workspace, scale, shift, stream);
    return group_norm_affine_nhwc(x, B, C, H, W, groups, eps, gamma, beta, nullptr, workspace, scale, shift, stream);
}

extern "C" int sige_hip_group_norm_affine_nhwc_bias_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                                        const float *gamma, const float *beta, const float *channel_bias,
                                                        float *workspace, float *scale, float *shift, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_group_norm_affine_nhwc_bias_f32, x, B, C, H, W, groups, eps, gamma, beta, channel_bias, workspace, scale, shift, stream);
    if (!channel_bias || (reinterpret_cast<uintptr_t>(channel_bias) & 15)) return SIGE_HIP_EINVAL;
    return group_norm_affine_nhwc(x, B, C, H, W, groups, eps, gamma, beta, channel_bias, workspace, scale, shift, stream);
}

static int group_norm_affine_nhwc(const float *x, int B, int C, int H, int W, int groups, float eps, const float *gamma,
                                  const float *beta, const float *cbias, float *workspace, float *scale, float *shift, void *stream) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || groups <= 0 || C % groups) return SIGE_HIP_EINVAL;
    if (!x || !workspace || !scale || !shift) return SIGE_HIP_EINVAL;
    const int C4 = C / 4;
    // one lane per channel quad, whole quads per group, whole pixels per 256-lane block
    if (C % 4 || (C / groups) % 4 || C4 > kGNThreads || kGNThreads % C4 || groups > kGNThreads || B > 65535 || (reinterpret_cast<uintptr_t>(x) & 15)) return SIGE_HIP_EUNSUPPORTED;
    const int HW = H * W, splits = gn_nhwc_splits(HW);
    hipStream_t st = as_stream(stream);
    gn_partial_nhwc_kernel<<<dim3(splits, B), kGNThreads, 0, st>>>(x, C, HW, groups, splits, workspace, cbias);
    gn_finish_kernel<<<B * groups, 64, 0, st>>>(workspace, splits, B, C, groups, (double)(C / groups) * HW, eps,
                                                          gamma, beta, scale, shift, cbias);
    return launch_status(2);
}

extern "C" int sige_hip_group_norm_affine_from_stats_f32(const float *stats1, int tiles1, int C1, int count1,
                                                         const float *stats2, int tiles2, int C2, int count2,
                                                         int B, int groups, float eps, const float *gamma, const float *beta,
                                                         const float *channel_bias, float *scale, float *shift, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_group_norm_affine_from_stats_f32, stats1, tiles1, C1, count1, stats2, tiles2, C2, count2, B, groups, eps, gamma, beta, channel_bias, scale, shift, stream);
    if (B <= 0 || C1 <= 0 || C2 < 0 || groups <= 0 || tiles1 <= 0 || count1 <= 0 || (C2 && (tiles2 <= 0 || count2 <= 0))) return SIGE_HIP_EINVAL;
    if (!stats1 || (C2 && !stats2) || !scale || !shift) return SIGE_HIP_EINVAL;
    const int C = C1 + C2;
    if (C % groups) return SIGE_HIP_EINVAL;
    const int cg = C / groups;
    // a group's channels = the fast index of the 256 lanes
    if (cg > 256 || (long)B * groups > 65535) return SIGE_HIP_EUNSUPPORTED;
    StatsPart p1{reinterpret_cast<const float2 *>(stats1), tiles1, C1, (double)count1};
    StatsPart p2{reinterpret_cast<const float2 *>(stats2), tiles2, C2, (double)count2};
    gn_from_stats_kernel<<<B * groups, 256, 0, as_stream(stream)>>>(p1, p2, groups, eps, gamma, beta, channel_bias, scale, shift);
    return launch_status(1);
}

extern "C" int sige_hip_channel_stats_tiles(int H, int W) { return (H * W + kStatPix - 1) / kStatPix; }

extern "C" int sige_hip_channel_stats_nhwc_f32(const float *x, int B, int C, int H, int W, float *stats, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_channel_stats_nhwc_f32, x, B, C, H, W, stats, stream);
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIGE_HIP_EINVAL;
    if (!x || !stats) return SIGE_HIP_EINVAL;
    const int C4 = C / 4;
    if (C % 4 || C4 > kGNThreads || B > 65535 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(stats) & 7)) return SIGE_HIP_EUNSUPPORTED;
    const int tiles = sige_hip_channel_stats_tiles(H, W);
    channel_stats_nhwc_kernel<<<dim3(tiles, B), kGNThreads, 0, as_stream(stream)>>>(x, C, H * W, tiles, reinterpret_cast<float2 *>(stats));
    return launch_status(1);
}
