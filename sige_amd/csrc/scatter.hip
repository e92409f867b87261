// scatter / scatter_with_block_residual / get_scatter_map for gfx950.
//
// Two forms of each scatter:
//   * two-pass  (any index list): 16-byte streaming copy y -> out, then a tile
//     pass (one workgroup = one tile x a channel chunk, coalesced 16-byte reads
//     of the contiguous tile slab) that overwrites the covered pixels;
//   * fused     (index lists from reduce_mask: tiles on a regular grid): ONE
//     streaming pass over the output.  Each lane owns 4 consecutive pixels of
//     a row; a [gH,gW] int32 tile table (L2-resident, built once per mask)
//     says whether they come from a conv-output tile (+residual) or from the
//     cached tensor.  No clone + overwrite, no second launch.
//
// Replaces: scatter_kernel / calibrate_residual_kernel (sige/cpu/scatter.cpp:4-68),
// scatter_cuda_kernel / calibrate_residual_cuda_kernel
// (sige/cuda/scatter_kernel.cu:8-74) and the y.clone() in their host wrappers
// (scatter.cpp:83, scatter_kernel.cu:89); get_scatter_map_*_kernel
// (sige/cpu/scatter_gather.cpp:58-84, scatter_gather_kernel.cu:69-98).
#include "common.hpp"

namespace sige {

constexpr int kThreads = 256;

// ---------------------------------------------------------------- copy ----
template <int VEC>
__global__ __launch_bounds__(kThreads) void copy_kernel(const float *__restrict__ src,
                                                        float *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (VEC == 4) {
        const size_t n4 = n / 4;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (; i + 3 * stride < n4; i += 4 * stride) {  // 4 independent 16-B loads in flight
            float4 a = s4[i], b = s4[i + stride], c = s4[i + 2 * stride], d = s4[i + 3 * stride];
            d4[i] = a; d4[i + stride] = b; d4[i + 2 * stride] = c; d4[i + 3 * stride] = d;
        }
        for (; i < n4; i += stride) d4[i] = s4[i];
        const size_t tail = n4 * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x;
        if (tail < n) dst[tail] = src[tail];
    } else {
        for (; i < n; i += stride) dst[i] = src[i];
    }
}

static void launch_copy(const float *src, float *dst, size_t n, hipStream_t st) {
    if (n == 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const size_t units = vec ? (n + 3) / 4 : n;
    const int blocks = (int)((units + kThreads - 1) / kThreads < 2048 ? (units + kThreads - 1) / kThreads : 2048);
    if (vec) copy_kernel<4><<<blocks, kThreads, 0, st>>>(src, dst, n);
    else copy_kernel<1><<<blocks, kThreads, 0, st>>>(src, dst, n);
}

// ----------------------------------------------------------- tile pass ----
struct TileArgs {
    const float *x;       // tiles [B*N,C,R,S]
    const float *y1;      // MODE 1: the cached shortcut tensor [B,C,H,W]
    float *out;           // [B,C,H,W]
    const int32_t *idx;   // [N,2]
    int B, C, H, W, N, R, S;
    int offH, offW, strH, strW;
    int cchunk;
    Bcast4 res;
};

// MODE 0: out[p] = x + residual[p]          (scatter_kernel, scatter.cpp:4-39)
// MODE 1: out[p] += x - y1[p]               (calibrate_residual_kernel, scatter.cpp:41-68)
template <int MODE, int VEC>
__global__ __launch_bounds__(kThreads) void tile_scatter_kernel(TileArgs a) {
    const int tile = blockIdx.x;
    const int b = tile / a.N, n = tile - b * a.N;
    const int c0 = blockIdx.y * a.cchunk;
    const int cc = min(a.cchunk, a.C - c0);
    const int RS = a.R * a.S;
    int h0 = a.idx[2 * n], w0 = a.idx[2 * n + 1];
    if (MODE == 0) { h0 = (a.offH + h0) / a.strH; w0 = (a.offW + w0) / a.strW; }
    const float *xb = a.x + ((size_t)tile * a.C + c0) * RS;
    const size_t HW = (size_t)a.H * a.W;
    const int total = cc * RS;
    for (int e0 = threadIdx.x * VEC; e0 < total; e0 += kThreads * VEC) {
        float v[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(xb + e0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = xb[e0];
        }
        int cl = e0 / RS;
        int p = e0 - cl * RS;
        int r = p / a.S, s = p - r * a.S;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int h = h0 + r, w = w0 + s, c = c0 + cl;
            if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
                const size_t q = ((size_t)b * a.C + c) * HW + (size_t)h * a.W + w;
                if (MODE == 0) {
                    float z = v[i];
                    if (a.res.data) z = bcast_load(a.res, b, c, h, w) + z;
                    a.out[q] = z;
                } else {
                    a.out[q] += v[i] - a.y1[q];
                }
            }
            if (++s == a.S) { s = 0; if (++r == a.R) { r = 0; ++cl; } }
        }
    }
}

template <int MODE>
static void launch_tiles(TileArgs a, hipStream_t st) {
    const int RS = a.R * a.S;
    const int tiles = a.B * a.N;
    if (tiles == 0 || a.C == 0 || RS == 0) return;
    int cchunk = max(4, (4096 / RS) & ~3);
    while (cchunk > 8 && (long)tiles * ceil_div(a.C, cchunk) < 512) cchunk = (cchunk / 2) & ~3;
    if (cchunk < 4) cchunk = 4;
    a.cchunk = cchunk;
    dim3 grid(tiles, ceil_div(a.C, cchunk));
    const bool vec4 = ((long)a.C * RS) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
    if (vec4) tile_scatter_kernel<MODE, 4><<<grid, kThreads, 0, st>>>(a);
    else tile_scatter_kernel<MODE, 1><<<grid, kThreads, 0, st>>>(a);
}

// ----------------------------------------------------------- tile table ----
__global__ void tile_table_kernel(const int32_t *idx, int N, int offH, int offW, int strH, int strW,
                                  int R, int S, int gH, int gW, int32_t *table) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int h0 = (offH + idx[2 * n]) / strH, w0 = (offW + idx[2 * n + 1]) / strW;
    if (h0 < 0 || w0 < 0) return;
    const int gh = h0 / R, gw = w0 / S;
    if (gh < gH && gw < gW) table[gh * gW + gw] = n;
}

// --------------------------------------------------------- fused scatter ----
struct FusedArgs {
    const float *x0, *y0, *x1, *y1;
    float *out;
    const int32_t *table0, *table1;
    int B, C, H, W;
    int R0, S0, N0, gW0;
    int R1, S1, N1, gW1;
    Bcast4 res;
};

// Lanes own VEC consecutive pixels of one row of one (b,c) plane.
// TS = compile-time tile width for the main tiles (4 => a 16-byte group never
// straddles two tiles, so tile data is read with one 16-byte load); 0 = generic.
// BLOCK_RES: scatter_with_block_residual (residual = y1 tensor + shortcut tiles).
template <int TS, bool BLOCK_RES, int VEC>
__global__ __launch_bounds__(kThreads) void fused_scatter_kernel(FusedArgs a) {
    const int plane = blockIdx.y;  // b*C + c
    const int b = plane / a.C, c = plane - b * a.C;
    const int Wv = a.W / VEC;  // VEC==4 only when W % 4 == 0
    const int units = a.H * Wv;
    const size_t pbase = (size_t)plane * a.H * a.W;
    const int RS0 = a.R0 * a.S0, RS1 = a.R1 * a.S1;
    const float *x0b = a.x0 + ((size_t)b * a.N0 * a.C + c) * RS0;
    const float *x1b = BLOCK_RES ? a.x1 + ((size_t)b * a.N1 * a.C + c) * RS1 : nullptr;
    for (int u = blockIdx.x * kThreads + threadIdx.x; u < units; u += gridDim.x * kThreads) {
        const int h = u / Wv, w = (u - h * Wv) * VEC;
        const size_t q = pbase + (size_t)h * a.W + w;
        float v[VEC], r1[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(a.y0 + q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = a.y0[q];
        }
        // tile lookups first: the cached shortcut tensor y1 is only needed where a
        // main or a shortcut tile covers the pixels (a few % of the plane)
        int t0v[VEC], t1v[VEC];
        bool any = false;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int ww = w + i;
            if (TS == 4 && VEC == 4) t0v[i] = (i == 0) ? a.table0[(h / a.R0) * a.gW0 + (w >> 2)] : t0v[0];
            else t0v[i] = a.table0[(h / a.R0) * a.gW0 + ww / a.S0];
            t1v[i] = BLOCK_RES ? a.table1[(h / a.R1) * a.gW1 + ww / a.S1] : -1;
            any |= (t0v[i] >= 0) | (t1v[i] >= 0);
        }
        if (BLOCK_RES && any) {
            if (VEC == 4) {
                const float4 s = *reinterpret_cast<const float4 *>(a.y1 + q);
                r1[0] = s.x; r1[1] = s.y; r1[2] = s.z; r1[3] = s.w;
            } else {
                r1[0] = a.y1[q];
            }
        }
        if (TS == 4 && VEC == 4) {
            // main tiles: 4 wide, w % 4 == 0  ->  one tile, one 16-byte row of it
            const int t0 = t0v[0];
            if (t0 >= 0) {
                const float4 t = *reinterpret_cast<const float4 *>(x0b + (size_t)t0 * a.C * RS0 + (h % a.R0) * 4);
                const float xv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float z = xv[i];
                    if (BLOCK_RES) z = r1[i] + z;
                    else if (a.res.data) z = bcast_load(a.res, b, c, h, w + i) + z;
                    v[i] = z;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int ww = w + i;
                const int t0 = t0v[i];
                if (t0 >= 0) {
                    float z = x0b[(size_t)t0 * a.C * RS0 + (h % a.R0) * a.S0 + ww % a.S0];
                    if (BLOCK_RES) z = r1[i] + z;
                    else if (a.res.data) z = bcast_load(a.res, b, c, h, ww) + z;
                    v[i] = z;
                }
            }
        }
        if (BLOCK_RES) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const int ww = w + i;
                const int t1 = t1v[i];
                if (t1 >= 0) v[i] += x1b[(size_t)t1 * a.C * RS1 + (h % a.R1) * a.S1 + ww % a.S1] - r1[i];
            }
        }
        if (VEC == 4) *reinterpret_cast<float4 *>(a.out + q) = make_float4(v[0], v[1], v[2], v[3]);
        else a.out[q] = v[0];
    }
}

template <bool BLOCK_RES>
static int launch_fused(const FusedArgs &a, hipStream_t st) {
    const long planes = (long)a.B * a.C;
    if (planes == 0 || a.H == 0 || a.W == 0) return SIGE_HIP_OK;
    if (planes > 65535) return SIGE_HIP_EUNSUPPORTED;
    uintptr_t al = reinterpret_cast<uintptr_t>(a.y0) | reinterpret_cast<uintptr_t>(a.out) |
                   reinterpret_cast<uintptr_t>(a.x0);
    if (BLOCK_RES) al |= reinterpret_cast<uintptr_t>(a.y1);
    const bool vec4 = (a.W % 4 == 0) && (al & 15) == 0;
    const int units = a.H * (vec4 ? a.W / 4 : a.W);
    int bx = ceil_div(units, kThreads * 4);  // ~4 units per lane
    if (bx < 1) bx = 1;
    dim3 grid(bx, (unsigned)planes);
    if (vec4 && a.S0 == 4) fused_scatter_kernel<4, BLOCK_RES, 4><<<grid, kThreads, 0, st>>>(a);
    else if (vec4) fused_scatter_kernel<0, BLOCK_RES, 4><<<grid, kThreads, 0, st>>>(a);
    else fused_scatter_kernel<0, BLOCK_RES, 1><<<grid, kThreads, 0, st>>>(a);
    return launch_status();
}

// ------------------------------------------------------------ scatter map ----
__global__ void scatter_map_kernel(int H, int W, int R, int S, int offH, int offW, int strH, int strW,
                                   const int32_t *idx, int N, int32_t *map) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int RS = R * S;
    if (t >= N * RS) return;
    const int n = t / RS, p = t - n * RS;
    const int r = p / S, s = p - r * S;
    const int h = (offH + idx[2 * n]) / strH + r, w = (offW + idx[2 * n + 1]) / strW + s;
    if (h < 0 || h >= H || w < 0 || w >= W) return;
    int32_t *m = map + 3 * ((size_t)h * W + w);
    m[0] = n; m[1] = r; m[2] = s;
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_copy_f32(const float *src, float *dst, size_t n, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_copy_f32, src, dst, n, stream);
    if (n && (!src || !dst)) return SIGE_HIP_EINVAL;
    launch_copy(src, dst, n, as_stream(stream));
    return launch_status();
}

extern "C" int sige_hip_scatter_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                                    int offsetH, int offsetW, int strideH, int strideW,
                                    const int32_t *active_indices, int N,
                                    const float *residual, int resB, int resC, int resH, int resW,
                                    float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_f32, x, y, B, C, H, W, R, S, offsetH, offsetW, strideH, strideW, active_indices, N, residual, resB, resC, resH, resW, out, stream);
    if (B < 0 || C < 0 || H < 0 || W < 0 || N < 0 || R <= 0 || S <= 0 || strideH <= 0 || strideW <= 0)
        return SIGE_HIP_EINVAL;
    const size_t n = (size_t)B * C * H * W;
    if (n && (!y || !out || y == out)) return SIGE_HIP_EINVAL;
    if ((long)B * N * C > 0 && (!x || !active_indices)) return SIGE_HIP_EINVAL;
    if (!bcast_ok(residual, resB, resC, resH, resW, B, C, H, W)) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    launch_copy(y, out, n, st);
    TileArgs a{};
    a.x = x; a.out = out; a.idx = active_indices;
    a.B = B; a.C = C; a.H = H; a.W = W; a.N = N; a.R = R; a.S = S;
    a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    a.res = make_bcast(residual, resB, resC, resH, resW);
    launch_tiles<0>(a, st);
    return launch_status(2);
}

extern "C" int sige_hip_scatter_with_block_residual_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, int N0, const int32_t *active_indices1, int N1,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_with_block_residual_f32, x0, y0, x1, y1, B, C, H, W, R0, S0, R1, S1, offsetH, offsetW, strideH, strideW, active_indices0, N0, active_indices1, N1, out, stream);
    if (B < 0 || C < 0 || H < 0 || W < 0 || N0 < 0 || N1 < 0 || R0 <= 0 || S0 <= 0 || R1 <= 0 || S1 <= 0 ||
        strideH <= 0 || strideW <= 0)
        return SIGE_HIP_EINVAL;
    const size_t n = (size_t)B * C * H * W;
    if (n && (!y0 || !y1 || !out || y0 == out || y1 == out)) return SIGE_HIP_EINVAL;
    if ((long)B * N0 * C > 0 && (!x0 || !active_indices0)) return SIGE_HIP_EINVAL;
    if ((long)B * N1 * C > 0 && (!x1 || !active_indices1)) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    launch_copy(y0, out, n, st);
    TileArgs a{};
    a.x = x0; a.out = out; a.idx = active_indices0;
    a.B = B; a.C = C; a.H = H; a.W = W; a.N = N0; a.R = R0; a.S = S0;
    a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    a.res = make_bcast(y1, B, C, H, W);
    launch_tiles<0>(a, st);
    TileArgs c{};
    c.x = x1; c.y1 = y1; c.out = out; c.idx = active_indices1;
    c.B = B; c.C = C; c.H = H; c.W = W; c.N = N1; c.R = R1; c.S = S1;
    c.offH = 0; c.offW = 0; c.strH = 1; c.strW = 1;
    launch_tiles<1>(c, st);
    return launch_status(3);
}

extern "C" int sige_hip_tile_table_i32(const int32_t *active_indices, int N, int offsetH, int offsetW,
                                       int strideH, int strideW, int R, int S, int gH, int gW,
                                       int32_t *table, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_tile_table_i32, (sige::CountOf<0, 1>), active_indices, N, offsetH, offsetW, strideH, strideW, R, S, gH, gW, table, stream);
    if (N < 0 || R <= 0 || S <= 0 || gH < 0 || gW < 0 || strideH <= 0 || strideW <= 0) return SIGE_HIP_EINVAL;
    if ((long)gH * gW == 0) return SIGE_HIP_OK;
    if (!table || (N && !active_indices)) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(table, 0xFF, sizeof(int32_t) * (size_t)gH * gW, st) != hipSuccess) return SIGE_HIP_ELAUNCH;
    if (N) tile_table_kernel<<<ceil_div(N, 256), 256, 0, st>>>(active_indices, N, offsetH, offsetW, strideH, strideW,
                                                              R, S, gH, gW, table);
    return launch_status();
}

extern "C" int sige_hip_scatter_fused_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                                          const int32_t *table, int gH, int gW, int N,
                                          const float *residual, int resB, int resC, int resH, int resW,
                                          float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_fused_f32, x, y, B, C, H, W, R, S, table, gH, gW, N, residual, resB, resC, resH, resW, out, stream);
    if (B < 0 || C < 0 || H < 0 || W < 0 || N < 0 || R <= 0 || S <= 0) return SIGE_HIP_EINVAL;
    if (gH != ceil_div(H, R) || gW != ceil_div(W, S)) return SIGE_HIP_EINVAL;
    const size_t n = (size_t)B * C * H * W;
    if (n && (!y || !out || !table || y == out)) return SIGE_HIP_EINVAL;
    if ((long)B * N * C > 0 && !x) return SIGE_HIP_EINVAL;
    if (!bcast_ok(residual, resB, resC, resH, resW, B, C, H, W)) return SIGE_HIP_EINVAL;
    FusedArgs a{};
    a.x0 = x; a.y0 = y; a.out = out; a.table0 = table;
    a.B = B; a.C = C; a.H = H; a.W = W; a.R0 = R; a.S0 = S; a.N0 = N; a.gW0 = gW;
    a.R1 = 1; a.S1 = 1;
    a.res = make_bcast(residual, resB, resC, resH, resW);
    return launch_fused<false>(a, as_stream(stream));
}

extern "C" int sige_hip_scatter_with_block_residual_fused_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *table1, int gH1, int gW1, int N1,
        float *out, void *stream) {
    SIGE_PLAN_HOOK_FIXED(sige_hip_scatter_with_block_residual_fused_f32, x0, y0, x1, y1, B, C, H, W, R0, S0, R1, S1, table0, gH0, gW0, N0, table1, gH1, gW1, N1, out, stream);
    if (B < 0 || C < 0 || H < 0 || W < 0 || N0 < 0 || N1 < 0 || R0 <= 0 || S0 <= 0 || R1 <= 0 || S1 <= 0)
        return SIGE_HIP_EINVAL;
    if (gH0 != ceil_div(H, R0) || gW0 != ceil_div(W, S0) || gH1 != ceil_div(H, R1) || gW1 != ceil_div(W, S1))
        return SIGE_HIP_EINVAL;
    const size_t n = (size_t)B * C * H * W;
    if (n && (!y0 || !y1 || !out || !table0 || !table1 || y0 == out || y1 == out)) return SIGE_HIP_EINVAL;
    if ((long)B * N0 * C > 0 && !x0) return SIGE_HIP_EINVAL;
    if ((long)B * N1 * C > 0 && !x1) return SIGE_HIP_EINVAL;
    FusedArgs a{};
    a.x0 = x0; a.y0 = y0; a.x1 = x1; a.y1 = y1; a.out = out; a.table0 = table0; a.table1 = table1;
    a.B = B; a.C = C; a.H = H; a.W = W;
    a.R0 = R0; a.S0 = S0; a.N0 = N0; a.gW0 = gW0;
    a.R1 = R1; a.S1 = S1; a.N1 = N1; a.gW1 = gW1;
    return launch_fused<true>(a, as_stream(stream));
}

extern "C" int sige_hip_scatter_map_i32(int H, int W, int bH, int bW, int kH, int kW,
                                        int offsetH, int offsetW, int strideH, int strideW,
                                        const int32_t *active_indices, int N, int32_t *map, void *stream) {
    SIGE_PLAN_HOOK_N(sige_hip_scatter_map_i32, (sige::CountOf<10, 11>), H, W, bH, bW, kH, kW, offsetH, offsetW, strideH, strideW, active_indices, N, map, stream);
    if (H < 0 || W < 0 || N < 0 || strideH <= 0 || strideW <= 0 || kH <= 0 || kW <= 0 || bH < kH || bW < kW)
        return SIGE_HIP_EINVAL;
    if ((long)H * W == 0) return SIGE_HIP_OK;
    if (!map || (N && !active_indices)) return SIGE_HIP_EINVAL;
    const int R = (bH - kH) / strideH + 1, S = (bW - kW) / strideW + 1;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(map, 0xFF, sizeof(int32_t) * 3 * (size_t)H * W, st) != hipSuccess) return SIGE_HIP_ELAUNCH;
    const long total = (long)N * R * S;
    if (total) scatter_map_kernel<<<(int)((total + 255) / 256), 256, 0, st>>>(H, W, R, S, offsetH, offsetW, strideH,
                                                                               strideW, active_indices, N, map);
    return launch_status();
}
