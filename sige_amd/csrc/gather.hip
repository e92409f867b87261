// gather / scatter_gather for gfx950.
//
// One workgroup = one active tile x one chunk of channels.  The tile's pixel
// table (source offset + validity for each of the bH*bW pixels, including the
// 2-px halo) is computed ONCE per workgroup into LDS; the 256 lanes then stream
// the [channels][bH][bW] output slab, which is contiguous in HBM, with 16-byte
// stores (4 consecutive output elements per lane, each with its own source
// load).  Tile origins are wave-uniform scalar loads.
//
// Replaces: gather_cpu_kernel / gather_cuda_kernel (sige/cpu/gather.cpp:4-58,
// sige/cuda/gather_kernel.cu:7-67: one thread per element, four integer
// div/mod and an index re-read per element) and scatter_gather_*_kernel
// (sige/cpu/scatter_gather.cpp:5-56, scatter_gather_kernel.cu:8-67: a 12-byte
// map read per element per channel).
#include "common.hpp"

namespace sige {

constexpr int kGatherThreads = 256;
constexpr int kMaxTilePixels = 1024;  // bH*bW upper bound for the LDS pixel table

struct GatherArgs {
    const float *x;        // gather: full input [B,C,H,W]; scatter_gather: conv-1 tiles [B*N,C,Rx,Sx]
    const float *y;        // scatter_gather only: cached full tensor [B,C,H,W]
    float *out;            // [B*N,C,bH,bW]
    const int32_t *idx;    // [N,2]
    const int32_t *map;    // scatter_gather only: [H,W,3]
    int B, C, H, W, N;
    int bH, bW;            // used when the template block dims are 0
    int RxSx, Sx;          // scatter_gather: x tile pixels / row length
    int cchunk;            // channels per workgroup
    Bcast4 scale, shift;
};

// src code per pixel:  -2 out of image (output 0) | -1 read y (mapped only) |
// >=0  gather: h*W+w ; scatter_gather: blk*C*RxSx + hx*Sx + wx
template <int TR, int TS, int ACT, bool ACT_FIRST, bool MAPPED, int VEC>
__global__ __launch_bounds__(kGatherThreads) void gather_kernel(GatherArgs a) {
    const int R = TR ? TR : a.bH, S = TS ? TS : a.bW;
    const int RS = R * S;
    __shared__ int s_src[TR ? TR * TS : kMaxTilePixels];
    __shared__ int s_hw[TR ? TR * TS : kMaxTilePixels];

    const int tile = blockIdx.x;  // b*N + n
    const int b = tile / a.N, n = tile - b * a.N;
    const int c0 = blockIdx.y * a.cchunk;
    const int cc = min(a.cchunk, a.C - c0);
    const int h0 = a.idx[2 * n], w0 = a.idx[2 * n + 1];

    for (int p = threadIdx.x; p < RS; p += kGatherThreads) {
        const int r = p / S, s = p - r * S;
        const int h = h0 + r, w = w0 + s;
        int src = -2;
        if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
            if (MAPPED) {
                const int32_t *m = a.map + 3 * ((size_t)h * a.W + w);
                const int blk = m[0];
                src = blk >= 0 ? blk * a.C * a.RxSx + m[1] * a.Sx + m[2] : -1;
            } else {
                src = h * a.W + w;
            }
        }
        s_src[p] = src;
        s_hw[p] = (h << 16) | (w & 0xffff);
    }
    __syncthreads();

    const size_t HW = (size_t)a.H * a.W;
    const float *xb = MAPPED ? a.x + (size_t)b * a.N * a.C * a.RxSx : a.x + (size_t)b * a.C * HW;
    const float *yb = a.y + (size_t)b * a.C * HW;
    float *ob = a.out + ((size_t)tile * a.C + c0) * RS;
    const int total = cc * RS;

    for (int e0 = threadIdx.x * VEC; e0 < total; e0 += kGatherThreads * VEC) {
        float v[VEC];
        int cl = e0 / RS;
        int p = e0 - cl * RS;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = c0 + cl;
            const int src = s_src[p];
            float z = 0.0f;
            if (src != -2) {
                if (MAPPED)
                    z = (src >= 0) ? xb[(size_t)c * a.RxSx + src] : yb[(size_t)c * HW + (s_hw[p] >> 16) * a.W + (s_hw[p] & 0xffff)];
                else
                    z = xb[(size_t)c * HW + src];
                if (ACT != SIGE_HIP_ACT_IDENTITY || a.scale.data || a.shift.data) {
                    const int hw = s_hw[p];
                    z = affine_act<ACT, ACT_FIRST>(z, a.scale, a.shift, b, c, hw >> 16, hw & 0xffff);
                }
            }
            v[i] = z;
            if (++p == RS) { p = 0; ++cl; }
        }
        if (VEC == 4) {
            *reinterpret_cast<float4 *>(ob + e0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) ob[e0 + i] = v[i];
        }
    }
}

// ---- plain gather, row form (the reference layout's fast path) ----------------------------------------------------
// In NCHW a tile is C x TR rows of TS contiguous floats.  One lane = one (channel, row): the row comes in with ONE
// 16-byte + one 8-byte (4-byte) load at its natural 4-byte alignment (gfx950 global loads need no more) and leaves with
// two stores -- 3x fewer memory instructions than an element per load, which is what bounds a strided gather
// (address processing, not HBM bytes: the rows of neighbouring tiles share their cache lines).  Consecutive lanes
// write consecutive 4*TS-byte pieces of the [C][TR][TS] output slab: fully coalesced stores.
// Rows that leave the image horizontally take the guarded per-element path; rows above / below are zero fill.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

template <int TR, int TS, int ACT, bool ACT_FIRST>
__global__ __launch_bounds__(kGatherThreads) void gather_rows_kernel(GatherArgs a) {
    static_assert(TS >= 4 && TS <= 6, "row forms for 4-, 5- and 6-wide tiles");
    const int tile = blockIdx.x;  // b*N + n
    const int b = tile / a.N, n = tile - b * a.N;
    const int c0 = blockIdx.y * a.cchunk;
    const int cc = min(a.cchunk, a.C - c0);
    const int h0 = a.idx[2 * n], w0 = a.idx[2 * n + 1];
    const size_t HW = (size_t)a.H * a.W;
    const float *xb = a.x + (size_t)b * a.C * HW;
    float *ob = a.out + ((size_t)tile * a.C + c0) * (TR * TS);
    const bool inside_w = w0 >= 0 && w0 + TS <= a.W;
    const bool plain = ACT == SIGE_HIP_ACT_IDENTITY && !a.scale.data && !a.shift.data;
    const bool row_uniform = (a.scale.sh | a.scale.sw | a.shift.sh | a.shift.sw) == 0;  // (absent operands have zero strides)
    for (int u = threadIdx.x; u < cc * TR; u += kGatherThreads) {
        const int cl = u / TR, r = u - cl * TR;
        const int c = c0 + cl, h = h0 + r;
        float v[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) v[i] = 0.0f;
        if (h >= 0 && h < a.H) {
            const float *src = xb + (size_t)c * HW + (size_t)h * a.W + w0;
            if (inside_w) {
                const f4u q = *reinterpret_cast<const f4u *>(src);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
                if (TS == 6) { const f2u t = *reinterpret_cast<const f2u *>(src + 4); v[4] = t[0]; v[5] = t[1]; }
                if (TS == 5) v[4] = src[4];
                if (!plain && row_uniform) {
                    // per-(batch, channel) affine: one scale / shift for the whole row (same two separately rounded ops)
                    const float sv = a.scale.data ? bcast_load(a.scale, b, c, 0, 0) : 1.0f;
                    const float tv = a.shift.data ? bcast_load(a.shift, b, c, 0, 0) : 0.0f;
#pragma unroll
                    for (int i = 0; i < TS; ++i) {
                        float z = v[i];
                        if (!ACT_FIRST) { if (a.scale.data) z = sv * z; if (a.shift.data) z = tv + z; }
                        z = activate<ACT>(z);
                        if (ACT_FIRST) { if (a.scale.data) z = sv * z; if (a.shift.data) z = tv + z; }
                        v[i] = z;
                    }
                } else if (!plain) {
#pragma unroll
                    for (int i = 0; i < TS; ++i) v[i] = affine_act<ACT, ACT_FIRST>(v[i], a.scale, a.shift, b, c, h, w0 + i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TS; ++i) {
                    const int w = w0 + i;
                    if (w >= 0 && w < a.W) {
                        const float z = src[i];
                        v[i] = plain ? z : affine_act<ACT, ACT_FIRST>(z, a.scale, a.shift, b, c, h, w);
                    }
                }
            }
        }
        float *dst = ob + (size_t)u * TS;
        *reinterpret_cast<f4u *>(dst) = f4u{v[0], v[1], v[2], v[3]};
        if (TS == 6) *reinterpret_cast<f2u *>(dst + 4) = f2u{v[4], v[5]};
        if (TS == 5) dst[4] = v[4];
    }
}

// ---- plain gather, grouped row form (round 2) ----------------------------------------------------------------------
// The row form above gives every lane its own cache line to read (rows of one tile are W floats apart, channels H*W):
// 64 line requests per wave-wide load for 24 useful bytes each -- 0.30 of the HBM peak, bound by address processing.
// Index lists come row-major sorted from reduce_mask, so CONSECUTIVE tiles are usually horizontal neighbours whose
// windows overlap (stride 4, width 6).  Here one workgroup takes kGroup consecutive tiles x a channel chunk and maps its
// lanes tile-fastest: the 8 lanes of one (channel, row) read 8 overlapping 24-byte windows out of the same two or three
// cache lines, which the address unit merges -- 3x fewer line requests.  The rows go to LDS ([tile][channel][row], one
// padded slab per tile) and leave as each tile's contiguous [C-chunk][TR][TS] slab with 16-byte stores.  Values and
// rounding are those of the row form (the same affine / activation code).
constexpr int kGroup = 8;

template <int TR, int TS, int ACT, bool ACT_FIRST>
__global__ __launch_bounds__(kGatherThreads) void gather_rows_grouped_kernel(GatherArgs a) {
    static_assert(TS >= 4 && TS <= 6, "row forms for 4-, 5- and 6-wide tiles");
    extern __shared__ __attribute__((aligned(16))) float g_lds[];
    __shared__ int s_org[kGroup][2];
    const int tiles = a.B * a.N;
    const int tile0 = blockIdx.x * kGroup;
    const int c0 = blockIdx.y * a.cchunk;
    const int cc = min(a.cchunk, a.C - c0);
    const int slab = cc * TR * TS;            // floats of one tile's output slab (contiguous in HBM)
    const int slab_pad = slab + 4;            // LDS pitch per tile: the tile-fastest writers hit different banks
    if (threadIdx.x < kGroup) {
        const int t = min(tile0 + (int)threadIdx.x, tiles - 1);
        const int n = t % a.N;
        s_org[threadIdx.x][0] = a.idx[2 * n];
        s_org[threadIdx.x][1] = a.idx[2 * n + 1];
    }
    __syncthreads();
    const size_t HW = (size_t)a.H * a.W;
    const bool plain = ACT == SIGE_HIP_ACT_IDENTITY && !a.scale.data && !a.shift.data;
    const bool row_uniform = (a.scale.sh | a.scale.sw | a.shift.sh | a.shift.sw) == 0;  // (absent operands have zero strides)
    for (int u = threadIdx.x; u < cc * TR * kGroup; u += kGatherThreads) {
        const int g = u % kGroup, r = (u / kGroup) % TR, cl = u / (kGroup * TR);
        const int tile = tile0 + g;
        if (tile >= tiles) continue;
        const int b = tile / a.N;
        const int c = c0 + cl, h0 = s_org[g][0], w0 = s_org[g][1], h = h0 + r;
        float v[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) v[i] = 0.0f;
        if (h >= 0 && h < a.H) {
            const float *src = a.x + ((size_t)b * a.C + c) * HW + (size_t)h * a.W + w0;
            if (w0 >= 0 && w0 + TS <= a.W) {
                const f4u q = *reinterpret_cast<const f4u *>(src);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
                if (TS == 6) { const f2u t = *reinterpret_cast<const f2u *>(src + 4); v[4] = t[0]; v[5] = t[1]; }
                if (TS == 5) v[4] = src[4];
                if (!plain && row_uniform) {
                    const float sv = a.scale.data ? bcast_load(a.scale, b, c, 0, 0) : 1.0f;
                    const float tv = a.shift.data ? bcast_load(a.shift, b, c, 0, 0) : 0.0f;
#pragma unroll
                    for (int i = 0; i < TS; ++i) {
                        float z = v[i];
                        if (!ACT_FIRST) { if (a.scale.data) z = sv * z; if (a.shift.data) z = tv + z; }
                        z = activate<ACT>(z);
                        if (ACT_FIRST) { if (a.scale.data) z = sv * z; if (a.shift.data) z = tv + z; }
                        v[i] = z;
                    }
                } else if (!plain) {
#pragma unroll
                    for (int i = 0; i < TS; ++i) v[i] = affine_act<ACT, ACT_FIRST>(v[i], a.scale, a.shift, b, c, h, w0 + i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TS; ++i) {
                    const int w = w0 + i;
                    if (w >= 0 && w < a.W) {
                        const float z = src[i];
                        v[i] = plain ? z : affine_act<ACT, ACT_FIRST>(z, a.scale, a.shift, b, c, h, w);
                    }
                }
            }
        }
        float *d = g_lds + g * slab_pad + (cl * TR + r) * TS;
#pragma unroll
        for (int i = 0; i < TS; ++i) d[i] = v[i];
    }
    __syncthreads();
    // every tile's slab: contiguous in HBM, 4-byte aligned in general (TS = 5), 16-byte pieces
    const int q4 = slab / 4, rem = slab - 4 * q4;
    for (int g = 0; g < kGroup; ++g) {
        const int tile = tile0 + g;
        if (tile >= tiles) break;  // uniform
        float *ob = a.out + ((size_t)tile * a.C + c0) * (TR * TS);
        const float *sl = g_lds + g * slab_pad;
        for (int i = threadIdx.x; i < q4; i += kGatherThreads)
            *reinterpret_cast<f4u *>(ob + 4 * i) = *reinterpret_cast<const f4u *>(sl + 4 * i);
        if ((int)threadIdx.x < rem) ob[4 * q4 + threadIdx.x] = sl[4 * q4 + threadIdx.x];
    }
}


template <int TR, int TS>
static void launch_rows(const GatherArgs &a, int act, bool first, dim3 grid, hipStream_t st) {
    dim3 blk(kGatherThreads);
    if (act == SIGE_HIP_ACT_SWISH) {
        if (first) gather_rows_kernel<TR, TS, SIGE_HIP_ACT_SWISH, true><<<grid, blk, 0, st>>>(a);
        else gather_rows_kernel<TR, TS, SIGE_HIP_ACT_SWISH, false><<<grid, blk, 0, st>>>(a);
    } else {
        gather_rows_kernel<TR, TS, SIGE_HIP_ACT_IDENTITY, false><<<grid, blk, 0, st>>>(a);
    }
}

template <int TR, int TS>
static void launch_rows_grouped(const GatherArgs &a, int act, bool first, hipStream_t st) {
    dim3 blk(kGatherThreads), grid(ceil_div(a.B * a.N, kGroup), ceil_div(a.C, a.cchunk));
    const size_t lds = (size_t)kGroup * (a.cchunk * TR * TS + 4) * sizeof(float);
    if (act == SIGE_HIP_ACT_SWISH) {
        if (first) gather_rows_grouped_kernel<TR, TS, SIGE_HIP_ACT_SWISH, true><<<grid, blk, lds, st>>>(a);
        else gather_rows_grouped_kernel<TR, TS, SIGE_HIP_ACT_SWISH, false><<<grid, blk, lds, st>>>(a);
    } else {
        gather_rows_grouped_kernel<TR, TS, SIGE_HIP_ACT_IDENTITY, false><<<grid, blk, lds, st>>>(a);
    }
}

template <int TR, int TS, bool MAPPED, int VEC>
static void launch_act(const GatherArgs &a, int act, bool first, dim3 grid, hipStream_t st) {
    dim3 blk(kGatherThreads);
    if (act == SIGE_HIP_ACT_SWISH) {
        if (first) gather_kernel<TR, TS, SIGE_HIP_ACT_SWISH, true, MAPPED, VEC><<<grid, blk, 0, st>>>(a);
        else gather_kernel<TR, TS, SIGE_HIP_ACT_SWISH, false, MAPPED, VEC><<<grid, blk, 0, st>>>(a);
    } else {
        // identity: activation_first does not change the result
        gather_kernel<TR, TS, SIGE_HIP_ACT_IDENTITY, false, MAPPED, VEC><<<grid, blk, 0, st>>>(a);
    }
}

static bool g_gather_grouped = true;  // sige_hip_gather_force_rows (benchmarking): false = always the one-tile row form

template <bool MAPPED>
static int launch(GatherArgs a, int act, bool first, hipStream_t st) {
    const int RS = a.bH * a.bW;
    if (RS > kMaxTilePixels) return SIGE_HIP_EUNSUPPORTED;
    const int tiles = a.B * a.N;
    if (tiles == 0 || a.C == 0 || RS == 0) return SIGE_HIP_OK;  // N = 0: empty output
    // channels per workgroup: ~2K-4K elements, but keep >= ~512 workgroups when the problem allows
    int cchunk = max(4, (4096 / RS) & ~3);
    while (cchunk > 8 && (long)tiles * ceil_div(a.C, cchunk) < 512) cchunk = (cchunk / 2) & ~3;
    if (cchunk < 4) cchunk = 4;
    a.cchunk = cchunk;
    dim3 grid(tiles, ceil_div(a.C, cchunk));
    if (!MAPPED && a.bH == a.bW && a.bH >= 4 && a.bH <= 6) {
        // enough tiles to fill the chip in groups of kGroup: the grouped row form (32 channels per workgroup = 37 KB of LDS)
        if (g_gather_grouped && (long)ceil_div(tiles, kGroup) * ceil_div(a.C, 32) >= 512) {
            a.cchunk = 32;
            if (a.bH == 6) launch_rows_grouped<6, 6>(a, act, first, st);
            else if (a.bH == 5) launch_rows_grouped<5, 5>(a, act, first, st);
            else launch_rows_grouped<4, 4>(a, act, first, st);
            return launch_status();
        }
        // row form: one lane per (channel, tile row); ~64 channels per workgroup, but >= ~512 workgroups when possible
        int cch = 64;
        while (cch > 8 && (long)tiles * ceil_div(a.C, cch) < 512) cch /= 2;
        a.cchunk = cch;
        dim3 rgrid(tiles, ceil_div(a.C, cch));
        if (a.bH == 6) launch_rows<6, 6>(a, act, first, rgrid, st);
        else if (a.bH == 5) launch_rows<5, 5>(a, act, first, rgrid, st);
        else launch_rows<4, 4>(a, act, first, rgrid, st);
        return launch_status();
    }
    const bool vec4 = ((long)a.C * RS) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
#define SIGE_DISPATCH(TR, TS)                                                    \
    do {                                                                         \
        if (vec4) launch_act<TR, TS, MAPPED, 4>(a, act, first, grid, st);        \
        else launch_act<TR, TS, MAPPED, 1>(a, act, first, grid, st);             \
    } while (0)
    if (a.bH == 6 && a.bW == 6) SIGE_DISPATCH(6, 6);
    else if (a.bH == 4 && a.bW == 4) SIGE_DISPATCH(4, 4);
    else if (a.bH == 5 && a.bW == 5) SIGE_DISPATCH(5, 5);
    else SIGE_DISPATCH(0, 0);
#undef SIGE_DISPATCH
    return launch_status();
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_gather_force_rows(int one_tile_rows) {
    g_gather_grouped = one_tile_rows == 0;
    return SIGE_HIP_OK;
}

extern "C" int sige_hip_gather_f32(const float *x, int B, int C, int H, int W, int bH, int bW,
                                   const int32_t *active_indices, int N,
                                   const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                   const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                   int activation, int activation_first, float *out, void *stream) {
    if (B < 0 || C < 0 || H < 0 || W < 0 || N < 0 || bH <= 0 || bW <= 0) return SIGE_HIP_EINVAL;
    if (H >= 32768 || W >= 32768) return SIGE_HIP_EUNSUPPORTED;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N * C > 0 && (!x || !out || !active_indices)) return SIGE_HIP_EINVAL;
    if (!bcast_ok(scale, scaleB, scaleC, scaleH, scaleW, B, C, H, W) ||
        !bcast_ok(shift, shiftB, shiftC, shiftH, shiftW, B, C, H, W))
        return SIGE_HIP_EINVAL;
    GatherArgs a{};
    a.x = x; a.y = nullptr; a.out = out; a.idx = active_indices; a.map = nullptr;
    a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.bH = bH; a.bW = bW;
    a.scale = make_bcast(scale, scaleB, scaleC, scaleH, scaleW);
    a.shift = make_bcast(shift, shiftB, shiftC, shiftH, shiftW);
    return launch<false>(a, activation, activation_first != 0, as_stream(stream));
}

extern "C" int sige_hip_scatter_gather_f32(const float *x, const float *y, int B, int C, int H, int W,
                                           int Rx, int Sx, int bH, int bW,
                                           const int32_t *active_indices, int N, const int32_t *scatter_map,
                                           const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                           const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                           int activation, int activation_first, float *out, void *stream) {
    if (B < 0 || C < 0 || H < 0 || W < 0 || N < 0 || bH <= 0 || bW <= 0 || Rx <= 0 || Sx <= 0) return SIGE_HIP_EINVAL;
    if (H >= 32768 || W >= 32768) return SIGE_HIP_EUNSUPPORTED;
    if ((long)N * C * Rx * Sx >= (1L << 31)) return SIGE_HIP_EUNSUPPORTED;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N * C > 0 && (!x || !y || !out || !active_indices || !scatter_map)) return SIGE_HIP_EINVAL;
    if (!bcast_ok(scale, scaleB, scaleC, scaleH, scaleW, B, C, H, W) ||
        !bcast_ok(shift, shiftB, shiftC, shiftH, shiftW, B, C, H, W))
        return SIGE_HIP_EINVAL;
    GatherArgs a{};
    a.x = x; a.y = y; a.out = out; a.idx = active_indices; a.map = scatter_map;
    a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.bH = bH; a.bW = bW;
    a.RxSx = Rx * Sx; a.Sx = Sx;
    a.scale = make_bcast(scale, scaleB, scaleC, scaleH, scaleW);
    a.shift = make_bcast(shift, shiftB, shiftC, shiftH, shiftW);
    return launch<true>(a, activation, activation_first != 0, as_stream(stream));
}
