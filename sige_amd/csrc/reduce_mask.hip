// reduce_mask on the device: difference mask [H,W] -> sorted int32 [N,2] list
// of active tile origins.
//
// Replaces the torch-op chain of sige/utils.py:8-37 (F.pad -> F.max_pool2d ->
// "> 0.5" -> torch.nonzero -> stride*i - pad): one workgroup walks the
// candidate grid in row-major order, 1024 candidates per step; each lane tests
// its candidate's bH x bW window (clipped to the image: the pad region is all
// zeros), a wave ballot + a 16-entry LDS scan give every active candidate its
// rank, so the list comes out in exactly nonzero()'s order.
#include "common.hpp"

namespace sige {

constexpr int kRMThreads = 1024;

__global__ __launch_bounds__(kRMThreads) void reduce_mask_kernel(
        const uint8_t *__restrict__ mask, int H, int W, int bH, int bW, int strH, int strW, int padH, int padW,
        int gh, int gw, int32_t *__restrict__ indices, int capacity, int32_t *__restrict__ count, int hp_shift) {
    __shared__ int s_wave[kRMThreads / kWave];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int total = gh * gw;
    for (int start = 0; start < total; start += kRMThreads) {
        const int cand = start + tid;
        bool active = false;
        int h0 = 0, w0 = 0;
        if (cand < total) {
            const int i = cand / gw, j = cand - i * gw;
            h0 = i * strH - padH;
            w0 = j * strW - padW;
            int ha = max(h0, 0), hb = min(h0 + bH, H);
            if (hp_shift) {
                // stacked edits (sige_hip_set_edit_batch): the mask is E masks stacked along H; a candidate belongs to the image its
                // window's third row lies in (the rule of the conv kernels' seam test) and only looks at THAT image's rows -- so the
                // list holds the per-edit lists, one after the other (a window reaching into the neighbour's mask would activate a
                // tile the single-edit forward leaves cached).  The trailing candidate of an image (h0 = hp - pad) is assigned to
                // the NEXT image: harmless, its output rows are out of that image's range)
                const int lo = ((h0 + 2) >> hp_shift) << hp_shift;
                ha = max(ha, lo);
                hb = min(hb, lo + (1 << hp_shift));
            }
            const int wa = max(w0, 0), wb = min(w0 + bW, W);
            for (int h = ha; h < hb && !active; ++h)
                for (int w = wa; w < wb; ++w)
                    if (mask[(size_t)h * W + w]) { active = true; break; }
        }
        const unsigned long long ballot = __ballot(active);
        const int rank = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(ballot);
        __syncthreads();
        int before = s_base, all = 0;
#pragma unroll
        for (int k = 0; k < kRMThreads / kWave; ++k) {
            const int c = s_wave[k];
            if (k < wave) before += c;
            all += c;
        }
        if (active) {
            const int pos = before + rank;
            if (pos < capacity) { indices[2 * pos] = h0; indices[2 * pos + 1] = w0; }
        }
        __syncthreads();
        if (tid == 0) s_base += all;
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_reduce_mask_capacity(int H, int W, int strideH, int strideW, int padH, int padW) {
    if (H < 0 || W < 0 || strideH <= 0 || strideW <= 0 || padH < 0 || padW < 0) return SIGE_HIP_EINVAL;
    return ((H + padH) / strideH + 1) * ((W + padW) / strideW + 1);
}

extern "C" int sige_hip_reduce_mask_i32(const uint8_t *mask, int H, int W, int bH, int bW,
                                        int strideH, int strideW, int padH, int padW,
                                        int32_t *indices, int capacity, int32_t *count, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_reduce_mask_i32, mask, H, W, bH, bW, strideH, strideW, padH, padW, indices, capacity, count, stream);
    if (H <= 0 || W <= 0 || bH <= 0 || bW <= 0 || strideH <= 0 || strideW <= 0 || padH < 0 || padW < 0 ||
        capacity < 0)
        return SIGE_HIP_EINVAL;
    if (!mask || !count || (capacity && !indices)) return SIGE_HIP_EINVAL;
    const int gh = (H + padH) / strideH + 1, gw = (W + padW) / strideW + 1;
    const int hp_shift = stacked_shift(H);
    if (hp_shift < 0) return SIGE_HIP_EUNSUPPORTED;
    // (the seam rule below places a window by its third row: with a padding above 2 the first candidate's third row is above
    //  row 0 and would be assigned to "image -1"; the tile convs have padding <= 1)
    if (hp_shift > 0 && padH > 2) return SIGE_HIP_EUNSUPPORTED;
    reduce_mask_kernel<<<1, kRMThreads, 0, as_stream(stream)>>>(mask, H, W, bH, bW, strideH, strideW, padH, padW,
                                                               gh, gw, indices, capacity, count, hp_shift);
    return launch_status();
}
