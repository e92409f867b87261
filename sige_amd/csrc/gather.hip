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
                        z = activate<ACT, ACT_FIRST>(z);
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
    // workgroup b runs on XCD b % 8: every XCD takes a contiguous eighth of the groups, so that the rows two vertically
    // adjacent groups share are fetched into ONE L2 (the grid is padded to 8 * per_xcd)
    const int ngroups = (tiles + kGroup - 1) / kGroup, per_xcd = (ngroups + 7) / 8;
    const int group = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (group >= ngroups) return;
    const int tile0 = group * kGroup;
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
                        z = activate<ACT, ACT_FIRST>(z);
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


// ---- scatter_gather, row form (round 3) -----------------------------------------------------------------------------
// The element form above (gather_kernel<MAPPED>) issues one 4-byte load per output element plus, with an affine, two more
// for its scale / shift: 3 loads and ~10 integer instructions per element -- 0.27 of the HBM peak, bound by instruction
// issue, not by bytes.  The sources are far more regular than that: conv-1's tiles are compact [blk][C][Rx][Sx] slabs on
// the stride grid, so the middle four pixels of a window row are FOUR CONSECUTIVE floats of one tile (the window's own
// tile for the inner rows, the neighbour above / below for the halo rows) or, where no tile covers them, of one row of
// the cached tensor.  One lane = one (channel, window row): the pixel table row comes from LDS, the middle four pixels
// with ONE 16-byte load when the table says they are consecutive, the remaining (halo column) pixels one by one, the
// per-(batch, channel) affine once per row; consecutive lanes store consecutive 4*TS-byte pieces of the output slab.
// Same values, same rounding (the same affine / activation code as the element form).
constexpr int kSgRowsChannels = 128;  // channels per workgroup, at most (3 full passes of the 256 lanes over a 6-row window)

template <int TR, int TS, int ACT, bool ACT_FIRST>
__global__ __launch_bounds__(kGatherThreads) void scatter_gather_rows_kernel(GatherArgs a) {
    static_assert(TS >= 4 && TS <= 6, "row forms for 4-, 5- and 6-wide windows");
    constexpr int RS = TR * TS, V0 = (TS - 4) / 2;  // the vector part: window columns [V0, V0 + 4)
    __shared__ int s_src[RS];
    const int tile = blockIdx.x;  // b*N + n
    const int b = tile / a.N, n = tile - b * a.N;
    const int c0 = blockIdx.y * a.cchunk;
    const int cc = min(a.cchunk, a.C - c0);
    const int h0 = a.idx[2 * n], w0 = a.idx[2 * n + 1];
    for (int p = threadIdx.x; p < RS; p += kGatherThreads) {
        const int r = p / TS, s = p - r * TS;
        const int h = h0 + r, w = w0 + s;
        int src = -2;
        if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
            const int32_t *m = a.map + 3 * ((size_t)h * a.W + w);
            const int blk = m[0];
            src = blk >= 0 ? blk * a.C * a.RxSx + m[1] * a.Sx + m[2] : -1;
        }
        s_src[p] = src;
    }
    __syncthreads();
    const size_t HW = (size_t)a.H * a.W;
    const float *xb = a.x + (size_t)b * a.N * a.C * a.RxSx;
    const float *yb = a.y + (size_t)b * a.C * HW;
    float *ob = a.out + ((size_t)tile * a.C + c0) * RS;
    const bool plain = ACT == SIGE_HIP_ACT_IDENTITY && !a.scale.data && !a.shift.data;
    const bool row_uniform = (a.scale.sh | a.scale.sw | a.shift.sh | a.shift.sw) == 0;  // (absent operands have zero strides)
    // All loads of the lane's (up to IT) rows are issued before the first store: written as a plain loop the stores of one
    // row order the loads of the next behind them (the compiler cannot rule out aliasing), and every pass then pays a full
    // memory round trip of its own -- the kernel is bound by that latency, not by bytes.
    constexpr int IT = (kSgRowsChannels * TR + kGatherThreads - 1) / kGatherThreads;
    int src[IT][TS];
    float v[IT][TS], sv[IT], tv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int u = threadIdx.x + it * kGatherThreads;
        const bool live = u < cc * TR;
        const int cl = live ? u / TR : 0, r = live ? u - cl * TR : 0;
        const int c = c0 + cl, h = h0 + r;
#pragma unroll
        for (int i = 0; i < TS; ++i) { src[it][i] = live ? s_src[r * TS + i] : -2; v[it][i] = 0.0f; }
        const float *xc = xb + (size_t)c * a.RxSx;
        const float *yr = yb + (long)c * (long)HW + (long)h * a.W + w0;  // (dereferenced only where the table says "inside, no tile")
        const int s0 = src[it][V0];
        const bool vx = s0 >= 0 && src[it][V0 + 1] == s0 + 1 && src[it][V0 + 2] == s0 + 2 && src[it][V0 + 3] == s0 + 3;
        const bool vy = s0 == -1 && src[it][V0 + 1] == -1 && src[it][V0 + 2] == -1 && src[it][V0 + 3] == -1;
        if (vx || vy) {
            const f4u q = *reinterpret_cast<const f4u *>(vx ? xc + s0 : yr + V0);
            v[it][V0] = q[0]; v[it][V0 + 1] = q[1]; v[it][V0 + 2] = q[2]; v[it][V0 + 3] = q[3];
        }
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            if ((vx || vy) && i >= V0 && i < V0 + 4) continue;
            if (src[it][i] >= 0) v[it][i] = xc[src[it][i]];
            else if (src[it][i] == -1) v[it][i] = yr[i];
        }
        sv[it] = 1.0f; tv[it] = 0.0f;
        if (!plain && row_uniform && live) {
            if (a.scale.data) sv[it] = bcast_load(a.scale, b, c, 0, 0);
            if (a.shift.data) tv[it] = bcast_load(a.shift, b, c, 0, 0);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int u = threadIdx.x + it * kGatherThreads;
        if (u >= cc * TR) break;
        const int cl = u / TR, r = u - cl * TR;
        const int c = c0 + cl, h = h0 + r;
        if (!plain && row_uniform) {
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                float z = v[it][i];
                if (!ACT_FIRST) { if (a.scale.data) z = sv[it] * z; if (a.shift.data) z = tv[it] + z; }
                z = activate<ACT, ACT_FIRST>(z);
                if (ACT_FIRST) { if (a.scale.data) z = sv[it] * z; if (a.shift.data) z = tv[it] + z; }
                v[it][i] = src[it][i] != -2 ? z : 0.0f;  // (outside the image: 0, not affine-transformed -- scatter_gather.cpp:27-30)
            }
        } else if (!plain) {
#pragma unroll
            for (int i = 0; i < TS; ++i)
                if (src[it][i] != -2) v[it][i] = affine_act<ACT, ACT_FIRST>(v[it][i], a.scale, a.shift, b, c, h, w0 + i);
        }
        float *dst = ob + (size_t)u * TS;
        *reinterpret_cast<f4u *>(dst) = f4u{v[it][0], v[it][1], v[it][2], v[it][3]};
        if (TS == 6) *reinterpret_cast<f2u *>(dst + 4) = f2u{v[it][4], v[it][5]};
        if (TS == 5) dst[4] = v[it][4];
    }
}

// ---- scatter_gather, grouped row form (round 3): 6x6 windows over 4x4 conv-1 tiles ------------------------------------------
// What bounds the row form is the number of cache lines its loads touch, not bytes: a window draws on NINE conv-1 tiles
// (its own, four edge neighbours for a row / column of four pixels each, four corner neighbours for ONE pixel each), every
// one of them a separate 64-byte slab per channel -- nine line requests per channel pair for 36 output pixels (measured: the
// two halo-column loads cost as much as two 16-byte loads of the middle pixels, 7 us each of a 38 us launch).  But the
// halo column of a window row IS the first / last of the four middle pixels of the same row of the horizontally adjacent
// window (windows overlap by two pixels), and index lists come row-major sorted from reduce_mask, so that window is usually
// the next tile of the list.  One workgroup takes kSgGroup consecutive tiles x 16 channels; every lane loads the middle
// four pixels of its (tile, channel, row) as before and leaves the outer two in a small LDS exchange array, from which the
// neighbouring tile's lane takes its halo pixel -- three line requests per channel pair and tile instead of nine.  Nothing
// is assumed about adjacency: a halo pixel is taken from the neighbour only where the scatter map says both read the same
// element; everything else (group ends, holes in the grid, the cached tensor) is loaded as in the row form.
// Same values, same rounding as the element form.
constexpr int kSgGroup = 8, kSgGroupCh = 16;

template <int ACT, bool ACT_FIRST, int G = kSgGroup, int CCH = kSgGroupCh>
__global__ __launch_bounds__(kGatherThreads) void scatter_gather_rows_grouped_kernel(GatherArgs a) {
    constexpr int TR = 6, TS = 6, RS = 36, V0 = 1, ROWS = CCH * TR;
    __shared__ int s_src[G][RS];    // -2 outside the image (0) | -1 the cached tensor | >= 0 offset in the batch's conv-1 tiles
    __shared__ int s_org[G][2], s_tb[G];
    // per window row: 1 the left halo pixel = the last middle pixel of the previous tile's row | 2 the right one = the first
    // middle pixel of the next tile's | 4 the middle four are consecutive conv-1 floats | 8 ... of the cached tensor
    __shared__ int s_nb[G][TR];
    __shared__ float2 s_ex[G][ROWS];  // (first, last) middle pixel of every (tile, channel, row)
    const int tiles = a.B * a.N;
    // workgroup b runs on XCD b % 8: every XCD takes a contiguous eighth of the groups, so that the conv-1 tiles two groups
    // share (the rows above / below, the slab past a group end) are fetched into ONE L2 (the grid is padded to 8 * per_xcd)
    const int ngroups = (tiles + G - 1) / G, per_xcd = (ngroups + 7) / 8;
    const int group = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (group >= ngroups) return;
    const int tile0 = group * G;
    const int glast = min(G, tiles - tile0) - 1;  // last live tile of the group
    const int c0 = blockIdx.y * CCH;
    const int cc = min(CCH, a.C - c0);
    const int tid = threadIdx.x;
    if (tid < G) {
        const int t = min(tile0 + tid, tiles - 1);
        const int b = t / a.N, n = t - b * a.N;
        s_tb[tid] = b;
        s_org[tid][0] = a.idx[2 * n];
        s_org[tid][1] = a.idx[2 * n + 1];
    }
    __syncthreads();
    {
        constexpr int NQ = (G * RS + kGatherThreads - 1) / kGatherThreads;
        int32_t m0[NQ], m1[NQ], m2[NQ];
        bool in[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {  // (all map entries of a lane in flight together)
            const int q = min(tid + k * kGatherThreads, G * RS - 1);
            const int g = q / RS, p = q - g * RS;
            const int r = p / TS, s = p - r * TS;
            const int h = s_org[g][0] + r, w = s_org[g][1] + s;
            in[k] = g <= glast && h >= 0 && h < a.H && w >= 0 && w < a.W;
            const int32_t *m = a.map + 3 * (in[k] ? (size_t)h * a.W + w : 0);
            m0[k] = m[0]; m1[k] = m[1]; m2[k] = m[2];
        }
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int q = tid + k * kGatherThreads;
            if (q < G * RS) (&s_src[0][0])[q] = !in[k] ? -2 : (m0[k] >= 0 ? m0[k] * a.C * 16 + m1[k] * 4 + m2[k] : -1);
        }
    }
    __syncthreads();
    if (tid < G * TR) {
        const int g = tid / TR, r = tid - g * TR;
        auto mid = [&](int gg) {  // 4: consecutive conv-1 floats | 8: the cached tensor | 0: mixed
            const int *sp = &s_src[gg][r * TS + V0];
            if (sp[0] >= 0 && sp[1] == sp[0] + 1 && sp[2] == sp[0] + 2 && sp[3] == sp[0] + 3) return 4;
            if (sp[0] == -1 && sp[1] == -1 && sp[2] == -1 && sp[3] == -1) return 8;
            return 0;
        };
        int f = mid(g);
        const int *me = &s_src[g][r * TS];
        // (same batch, both conv-1 sourced, the SAME element: codes are offsets inside one batch's tiles)
        if (g > 0 && s_tb[g - 1] == s_tb[g] && me[0] >= 0 && mid(g - 1) == 4 && me[0] == s_src[g - 1][r * TS + V0 + 3]) f |= 1;
        if (g < glast && s_tb[g + 1] == s_tb[g] && me[TS - 1] >= 0 && mid(g + 1) == 4 && me[TS - 1] == s_src[g + 1][r * TS + V0]) f |= 2;
        s_nb[g][r] = f;
    }
    __syncthreads();
    const size_t HW = (size_t)a.H * a.W;
    const bool plain = ACT == SIGE_HIP_ACT_IDENTITY && !a.scale.data && !a.shift.data;
    const bool row_uniform = (a.scale.sh | a.scale.sw | a.shift.sh | a.shift.sw) == 0;  // (absent operands have zero strides)
    constexpr int IT = G * ROWS / kGatherThreads;
    float v[IT][TS], sv[IT], tv[IT];
    int fl[IT];       // the row's s_nb flags | 16 the lane has a row in this pass
    int oob[IT];      // bit i: pixel i lies outside the image (stays 0, not affine-transformed -- scatter_gather.cpp:27-30)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int u = tid + it * kGatherThreads;
        const int g = u / ROWS, rem = u - g * ROWS;
        const int cl = rem / TR, r = rem - cl * TR;
        fl[it] = 0; oob[it] = 0; sv[it] = 1.0f; tv[it] = 0.0f;
#pragma unroll
        for (int i = 0; i < TS; ++i) v[it][i] = 0.0f;
        if (g > glast || cl >= cc) continue;
        const int bg = s_tb[g];
        const int c = c0 + cl, h = s_org[g][0] + r, w0 = s_org[g][1];
        const int f = s_nb[g][r];
        fl[it] = f | 16;
        const float *xc = a.x + ((size_t)bg * a.N * a.C + c) * 16;
        const float *yr = a.y + ((long)bg * a.C + c) * (long)HW + (long)h * a.W + w0;  // (dereferenced only where the table says "inside, no tile")
        int src[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) { src[i] = s_src[g][r * TS + i]; if (src[i] == -2) oob[it] |= 1 << i; }
        if (f & 12) {
            const f4u q = *reinterpret_cast<const f4u *>((f & 4) ? xc + src[V0] : yr + V0);
            v[it][V0] = q[0]; v[it][V0 + 1] = q[1]; v[it][V0 + 2] = q[2]; v[it][V0 + 3] = q[3];
            if (f & 4) s_ex[g][rem] = make_float2(q[0], q[3]);
        }
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            if ((f & 12) && i >= V0 && i < V0 + 4) continue;
            if ((i == 0 && (f & 1)) || (i == TS - 1 && (f & 2))) continue;  // (from the neighbour's lane, below)
            if (src[i] >= 0) v[it][i] = xc[src[i]];
            else if (src[i] == -1) v[it][i] = yr[i];
        }
        if (!plain && row_uniform) {
            if (a.scale.data) sv[it] = bcast_load(a.scale, bg, c, 0, 0);
            if (a.shift.data) tv[it] = bcast_load(a.shift, bg, c, 0, 0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (!(fl[it] & 16)) continue;
        const int u = tid + it * kGatherThreads;
        const int g = u / ROWS, rem = u - g * ROWS;
        const int cl = rem / TR, r = rem - cl * TR;
        if (fl[it] & 1) v[it][0] = s_ex[g - 1][rem].y;
        if (fl[it] & 2) v[it][TS - 1] = s_ex[g + 1][rem].x;
        const int c = c0 + cl;
        if (!plain && row_uniform) {
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                float z = v[it][i];
                if (!ACT_FIRST) { if (a.scale.data) z = sv[it] * z; if (a.shift.data) z = tv[it] + z; }
                z = activate<ACT, ACT_FIRST>(z);
                if (ACT_FIRST) { if (a.scale.data) z = sv[it] * z; if (a.shift.data) z = tv[it] + z; }
                v[it][i] = (oob[it] >> i) & 1 ? 0.0f : z;
            }
        } else if (!plain) {
            const int h = s_org[g][0] + r, w0 = s_org[g][1];
#pragma unroll
            for (int i = 0; i < TS; ++i)
                if (!((oob[it] >> i) & 1)) v[it][i] = affine_act<ACT, ACT_FIRST>(v[it][i], a.scale, a.shift, s_tb[g], c, h, w0 + i);
        }
        float *dst = a.out + ((size_t)(tile0 + g) * a.C + c) * RS + r * TS;
        *reinterpret_cast<f4u *>(dst) = f4u{v[it][0], v[it][1], v[it][2], v[it][3]};
        *reinterpret_cast<f2u *>(dst + 4) = f2u{v[it][4], v[it][5]};
    }
}

template <int G, int CCH>
static void launch_sg_rows_grouped_as(const GatherArgs &a, int act, bool first, hipStream_t st) {
    dim3 blk(kGatherThreads), grid(8 * ceil_div(ceil_div(a.B * a.N, G), 8), ceil_div(a.C, CCH));
    if (act == SIGE_HIP_ACT_SWISH) {
        if (first) scatter_gather_rows_grouped_kernel<SIGE_HIP_ACT_SWISH, true, G, CCH><<<grid, blk, 0, st>>>(a);
        else scatter_gather_rows_grouped_kernel<SIGE_HIP_ACT_SWISH, false, G, CCH><<<grid, blk, 0, st>>>(a);
    } else {
        scatter_gather_rows_grouped_kernel<SIGE_HIP_ACT_IDENTITY, false, G, CCH><<<grid, blk, 0, st>>>(a);
    }
}

static void launch_sg_rows_grouped(const GatherArgs &a, int act, bool first, hipStream_t st) {
    launch_sg_rows_grouped_as<kSgGroup, kSgGroupCh>(a, act, first, st);
}

template <int TR, int TS>
static void launch_sg_rows(const GatherArgs &a, int act, bool first, dim3 grid, hipStream_t st) {
    dim3 blk(kGatherThreads);
    if (act == SIGE_HIP_ACT_SWISH) {
        if (first) scatter_gather_rows_kernel<TR, TS, SIGE_HIP_ACT_SWISH, true><<<grid, blk, 0, st>>>(a);
        else scatter_gather_rows_kernel<TR, TS, SIGE_HIP_ACT_SWISH, false><<<grid, blk, 0, st>>>(a);
    } else {
        scatter_gather_rows_kernel<TR, TS, SIGE_HIP_ACT_IDENTITY, false><<<grid, blk, 0, st>>>(a);
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
    dim3 blk(kGatherThreads), grid(8 * ceil_div(ceil_div(a.B * a.N, kGroup), 8), ceil_div(a.C, a.cchunk));
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
    if (MAPPED && !(tuning(SIGE_HIP_TUNE_SCATTER_GATHER_FORM) & 1) && a.bH == a.bW && a.bH >= 4 && a.bH <= 6) {
        // 6x6 windows over 4x4 tiles with enough tiles to fill the chip in groups: the grouped form (halo pixels through LDS)
        if (!(tuning(SIGE_HIP_TUNE_SCATTER_GATHER_FORM) & 2) && a.bH == 6 && a.RxSx == 16 && a.Sx == 4 && (long)ceil_div(tiles, kSgGroup) * ceil_div(a.C, kSgGroupCh) >= 512) {
            launch_sg_rows_grouped(a, act, first, st);
            return launch_status();
        }
        // row form: one lane per (channel, window row); 128 channels per workgroup (3 full passes of the 256 lanes for a 6-row
        // window), but >= ~512 workgroups when the problem allows
        int cch = kSgRowsChannels;
        while (cch > 8 && (long)tiles * ceil_div(a.C, cch) < 512) cch /= 2;
        a.cchunk = cch;
        dim3 rgrid(tiles, ceil_div(a.C, cch));
        if (a.bH == 6) launch_sg_rows<6, 6>(a, act, first, rgrid, st);
        else if (a.bH == 5) launch_sg_rows<5, 5>(a, act, first, rgrid, st);
        else launch_sg_rows<4, 4>(a, act, first, rgrid, st);
        return launch_status();
    }
    if (!MAPPED && a.bH == a.bW && a.bH >= 4 && a.bH <= 6) {
        // enough tiles to fill the chip in groups of kGroup: the grouped row form (32 channels per workgroup = 37 KB of LDS)
        if (!tuning(SIGE_HIP_TUNE_GATHER_ONE_TILE_ROWS) && (long)ceil_div(tiles, kGroup) * ceil_div(a.C, 32) >= 512) {
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

extern "C" int sige_hip_gather_f32(const float *x, int B, int C, int H, int W, int bH, int bW,
                                   const int32_t *active_indices, int N,
                                   const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                   const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                   int activation, int activation_first, float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_gather_f32, x, B, C, H, W, bH, bW, active_indices, N, scale, scaleB, scaleC, scaleH, scaleW, shift, shiftB, shiftC, shiftH, shiftW, activation, activation_first, out, stream);
    if (stacked_shift(H) != 0) return SIGE_HIP_EUNSUPPORTED;  // (stacked edits: channels-last fused kernels only)
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
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_gather_f32, x, y, B, C, H, W, Rx, Sx, bH, bW, active_indices, N, scatter_map, scale, scaleB, scaleC, scaleH, scaleW, shift, shiftB, shiftC, shiftH, shiftW, activation, activation_first, out, stream);
    if (stacked_shift(H) != 0) return SIGE_HIP_EUNSUPPORTED;  // (stacked edits: channels-last fused kernels only)
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
