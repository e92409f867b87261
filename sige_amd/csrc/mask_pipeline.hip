// Edit -> difference mask -> dilation -> mask pyramid, on the device and without a host round trip.
//
// Replaces the torch-op chains of sige/utils.py: compute_difference_mask (:74-85), dilate_mask (:40-71) and
// downsample_mask (:88-118).  The reference's downsample_mask evaluates `min(threshold, level.max() - eps)` in
// Python once per pyramid level -- a device -> host synchronisation per level when the mask lives on the GPU
// (sige/utils.py:107).  Here the whole pyramid is ONE launch: the running float mask is halved level by level by a
// single workgroup (the masks are 512 x 512 at most: 1 MB), the per-level maximum is a workgroup reduction, and the
// thresholded + dilated bits of every level land in one packed byte buffer.  Together with the index compaction
// (reduce_mask.hip, which writes its counts to device memory) the host reads ONE small array at the end.
//
// Arithmetic follows the reference's fp32 ops exactly:
//   difference   |a - b| > eps  (difference and comparison in fp32), OR over channels
//   dilation     OR of the ORIGINAL mask shifted by 1..dH along H and by 1..dW along W (a plus, not a box)
//   threshold    t = level.max() - eps;  if (threshold < t) t = threshold;  bit = level > t
//   halving      F.interpolate(level, (h/2, w/2), mode="bilinear", align_corners=False): source index
//                max(0, (dst + 0.5) * in/out - 0.5), lerp between floor and floor + 1 (clamped), evaluated as
//                hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11).  For the even sizes every SIGE model
//                produces this is the 2 x 2 mean of 0/1-derived dyadic values: exact in fp32 in any order.
#include "common.hpp"

namespace sige {

constexpr int kMPThreads = 1024;

__global__ void difference_mask_kernel(const float *__restrict__ a, const float *__restrict__ b, int C, long HW,
                                       long strideC, long stridePix, float eps, uint8_t *__restrict__ out) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
        bool any = false;
        for (int c = 0; c < C; ++c) {
            const long o = c * strideC + p * stridePix;
            any |= fabsf(a[o] - b[o]) > eps;
        }
        out[p] = any ? 1 : 0;
    }
}

// (lo, hi): the rows of the image pixel (h, w) belongs to -- [0, H) normally; with stacked edits (sige_hip_set_edit_batch) the
// mask is E masks stacked along H and a dilation must not reach into the neighbour's rows
__device__ __forceinline__ uint8_t dilated(const uint8_t *__restrict__ m, int W, int h, int w, int dH, int dW, int lo, int hi) {
    if (m[(size_t)h * W + w]) return 1;
    for (int i = 1; i <= dH; ++i) {
        if (h - i >= lo && m[(size_t)(h - i) * W + w]) return 1;
        if (h + i < hi && m[(size_t)(h + i) * W + w]) return 1;
    }
    for (int i = 1; i <= dW; ++i) {
        if (w - i >= 0 && m[(size_t)h * W + (w - i)]) return 1;
        if (w + i < W && m[(size_t)h * W + (w + i)]) return 1;
    }
    return 0;
}

__global__ void dilate_mask_kernel(const uint8_t *__restrict__ mask, int H, int W, int dH, int dW, uint8_t *__restrict__ out, int hp) {
    const long n = (long)H * W;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int h = (int)(p / W);
        const int lo = (h / hp) * hp;
        out[p] = dilated(mask, W, h, (int)(p % W), dH, dW, lo, lo + hp);
    }
}

// One workgroup, all levels.  level_a / level_b: float ping-pong (H*W and (H/2)*(W/2) floats), bits: H*W bytes of scratch
// for the thresholded (not yet dilated) bits of the current level, out: the packed pyramid (level k after level k-1).
__global__ __launch_bounds__(kMPThreads) void mask_pyramid_kernel(
        const uint8_t *__restrict__ mask, int H, int W, int min_h, int min_w, int dH, int dW, float threshold, float eps,
        float *__restrict__ level_a, float *__restrict__ level_b, uint8_t *__restrict__ bits, uint8_t *__restrict__ out,
        size_t scratch_stride_floats) {
    __shared__ float s_max[kMPThreads / kWave];
    __shared__ float s_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // stacked edits: workgroup e builds the pyramid of image e (H = ONE image's height; its own maxima and thresholds, as the
    // edit's own downsample_mask call would) and writes its rows of every level of the TALL pyramid
    const int E = gridDim.x, e = blockIdx.x;
    mask += (size_t)e * H * W;
    level_a += (size_t)e * scratch_stride_floats;
    level_b += (size_t)e * scratch_stride_floats;
    bits += (size_t)e * scratch_stride_floats * 4;
    int h = H, w = W;
    const float *cur = nullptr;  // nullptr: level 0 = the byte mask itself
    float *nxt = level_a;
    size_t out_off = 0;
    while (true) {
        const long n = (long)h * w;
        // (1) level.max()
        float m = -INFINITY;
        for (long p = tid; p < n; p += kMPThreads) m = fmaxf(m, cur ? cur[p] : (mask[p] ? 1.0f : 0.0f));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) s_max[wave] = m;
        __syncthreads();
        if (tid == 0) {
            float mm = s_max[0];
            for (int k = 1; k < kMPThreads / kWave; ++k) mm = fmaxf(mm, s_max[k]);
            float t = mm - eps;
            if (threshold < t) t = threshold;
            s_t = t;
        }
        __syncthreads();
        const float t = s_t;
        // (2) thresholded bits of this level
        for (long p = tid; p < n; p += kMPThreads) bits[p] = (cur ? cur[p] : (mask[p] ? 1.0f : 0.0f)) > t ? 1 : 0;
        __syncthreads();
        // (3) dilated bits -> the packed output
        for (long p = tid; p < n; p += kMPThreads) out[out_off + (size_t)e * n + p] = dilated(bits, w, (int)(p / w), (int)(p % w), dH, dW, 0, h);
        out_off += (size_t)E * n;
        const int h2 = h / 2, w2 = w / 2;
        if (h2 < min_h && w2 < min_w) break;
        if (h2 <= 0 || w2 <= 0) break;  // (host side never asks for this: guards the division below)
        // (4) bilinear halving of the FLOAT level (not of the bits)
        const float sy = (float)h / (float)h2, sx = (float)w / (float)w2;
        const long n2 = (long)h2 * w2;
        for (long p = tid; p < n2; p += kMPThreads) {
            const int i = (int)(p / w2), j = (int)(p % w2);
            float fy = sy * ((float)i + 0.5f) - 0.5f;
            if (fy < 0.f) fy = 0.f;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = fy - (float)y0, hy = 1.f - ly;
            float fx = sx * ((float)j + 0.5f) - 0.5f;
            if (fx < 0.f) fx = 0.f;
            const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            auto at = [&](int y, int x) -> float {
                const size_t q = (size_t)y * w + x;
                return cur ? cur[q] : (mask[q] ? 1.0f : 0.0f);
            };
            nxt[p] = hy * (hx * at(y0, x0) + lx * at(y0, x1)) + ly * (hx * at(y1, x0) + lx * at(y1, x1));
        }
        __syncthreads();
        cur = nxt;
        nxt = (nxt == level_a) ? level_b : level_a;
        h = h2; w = w2;
    }
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_difference_mask_u8(const float *a, const float *b, int C, int H, int W,
                                           int64_t strideC, int64_t strideH, int64_t strideW, float eps,
                                           uint8_t *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_difference_mask_u8, a, b, C, H, W, strideC, strideH, strideW, eps, out, stream);
    if (C <= 0 || H <= 0 || W <= 0) return SIGE_HIP_EINVAL;
    if (!a || !b || !out) return SIGE_HIP_EINVAL;
    if (strideH != (int64_t)W * strideW) return SIGE_HIP_EUNSUPPORTED;  // pixels must be evenly strided (NCHW or NHWC)
    const long HW = (long)H * W;
    const int blocks = (int)((HW + 255) / 256 < 2048 ? (HW + 255) / 256 : 2048);
    difference_mask_kernel<<<blocks, 256, 0, as_stream(stream)>>>(a, b, C, HW, strideC, strideW, eps, out);
    return launch_status();
}

extern "C" int sige_hip_dilate_mask_u8(const uint8_t *mask, int H, int W, int dilationH, int dilationW,
                                       uint8_t *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_dilate_mask_u8, mask, H, W, dilationH, dilationW, out, stream);
    if (H <= 0 || W <= 0 || dilationH < 0 || dilationW < 0) return SIGE_HIP_EINVAL;
    if (!mask || !out || mask == out) return SIGE_HIP_EINVAL;
    const long n = (long)H * W;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    const int sh = stacked_shift(H);  // (stacked edits: a dilation stays inside its own image)
    if (sh < 0) return SIGE_HIP_EUNSUPPORTED;
    dilate_mask_kernel<<<blocks, 256, 0, as_stream(stream)>>>(mask, H, W, dilationH, dilationW, out, sh ? (1 << sh) : H);
    return launch_status();
}

// number of pyramid levels and their sizes (the loop of sige/utils.py:105-117); returns the level count,
// writes min(count, capacity) entries of hs / ws (either may be NULL)
extern "C" int sige_hip_mask_pyramid_levels(int H, int W, int min_h, int min_w, int *hs, int *ws, int capacity) {
    if (H <= 0 || W <= 0) return SIGE_HIP_EINVAL;
    int n = 0, h = H, w = W;
    while (true) {
        if (n < capacity) {
            if (hs) hs[n] = h;
            if (ws) ws[n] = w;
        }
        ++n;
        h /= 2; w /= 2;
        if ((h < min_h && w < min_w) || h <= 0 || w <= 0) break;
    }
    return n;
}

extern "C" int sige_hip_mask_pyramid_u8(const uint8_t *mask, int H, int W, int min_h, int min_w,
                                        int dilationH, int dilationW, float threshold, float eps,
                                        float *scratch, size_t scratch_floats, uint8_t *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_mask_pyramid_u8, mask, H, W, min_h, min_w, dilationH, dilationW, threshold, eps, scratch, scratch_floats, out, stream);
    if (H <= 0 || W <= 0 || dilationH < 0 || dilationW < 0) return SIGE_HIP_EINVAL;
    if (!mask || !scratch || !out) return SIGE_HIP_EINVAL;
    // stacked edits (sige_hip_set_edit_batch): `mask` is E masks stacked along H -- one workgroup per image, each with its own
    // slice of the scratch; `out` is the packed pyramid of the TALL image (level k: [E * h_k, w_k]); min_h applies to ONE image
    const int sh = stacked_shift(H);
    if (sh < 0) return SIGE_HIP_EUNSUPPORTED;
    const int Hp = sh ? (1 << sh) : H, E = H / Hp;
    // scratch per image: level_a [Hp/2 * W/2] | level_b [Hp/4 * W/4] floats | bits [Hp * W] bytes (rounded up to floats)
    const size_t na = (size_t)(Hp / 2) * (W / 2), nb = (size_t)(Hp / 4) * (W / 4);
    const size_t need = na + nb + ((size_t)Hp * W + 3) / 4 + 8;
    if (scratch_floats < need * (size_t)E) return SIGE_HIP_EINVAL;
    float *level_a = scratch, *level_b = scratch + na + 4;
    uint8_t *bits = reinterpret_cast<uint8_t *>(scratch + na + nb + 8);
    mask_pyramid_kernel<<<E, kMPThreads, 0, as_stream(stream)>>>(mask, Hp, W, min_h, min_w, dilationH, dilationW, threshold, eps,
                                                                level_a, level_b, bits, out, need);
    return launch_status();
}
