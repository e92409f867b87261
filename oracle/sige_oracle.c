/*
 * sige_oracle.c -- CPU restatement of the reference's tiling-sparse-conv hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sige_amd/ may import, link or call
 * this file; it exists so tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the HIP path against the reference's
 * algorithm.  Parity is PINNED: tests/test_oracle_golden.py checks every entry
 * point below against vectors produced by the reference's own sige/cpu
 * extension + sige.nn / sige.utils Python (tests/golden/make_golden.py), and
 * tests/test_oracle_golden.py::test_oracle_vs_compiled_reference checks it live
 * against oracle/_ref (the reference's sige/cpu compiled here) when that is built.
 *
 * Each function names the reference file:line it restates (paths relative to
 * /root/reference).  Plain C99, scalar, single-threaded unless built with
 * -fopenmp (the reference CPU path is OpenMP collapse(3) over (B, N, C):
 * sige/cpu/gather.cpp:17, scatter.cpp:14,51, scatter_gather.cpp:18).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_ACT_IDENTITY 0
#define ORACLE_ACT_SWISH 1

/* A broadcastable 4-D operand (scale / shift / residual): dims are either 1 or
 * the full extent; data == NULL means "operand absent".
 * Restates binary_op_array, sige/cpu/common_cpu.cpp:13-27. */
typedef struct {
    const float *data;
    int B, C, H, W;
} oracle_bcast4;

static inline size_t bcast_offset(const oracle_bcast4 *t, int b, int c, int h, int w) {
    size_t p = 0;
    if (t->W > 1) p = (size_t)w;
    if (t->H > 1) p += (size_t)h * t->W;
    if (t->C > 1) p += (size_t)c * t->H * t->W;
    if (t->B > 1) p += (size_t)b * t->C * t->H * t->W;
    return p;
}

/* sige/cpu/common_cpu.cpp:29-35: SWISH is `z / (1.0 + exp(-z))` -- exp on a
 * float argument (expf), the sum and quotient in double, rounded to float on
 * return. */
static inline float act_apply(int act, float z) {
    if (act == ORACLE_ACT_SWISH) return (float)((double)z / (1.0 + (double)expf(-z)));
    return z;
}

/* scale/shift + activation in the order gather.cpp:33-53 applies them. */
static inline float affine_act(float z, const oracle_bcast4 *scale, const oracle_bcast4 *shift,
                               int act, int act_first, int b, int c, int h, int w) {
    if (!act_first) {
        if (scale->data) z = scale->data[bcast_offset(scale, b, c, h, w)] * z;
        if (shift->data) z = shift->data[bcast_offset(shift, b, c, h, w)] + z;
    }
    z = act_apply(act, z);
    if (act_first) {
        if (scale->data) z = scale->data[bcast_offset(scale, b, c, h, w)] * z;
        if (shift->data) z = shift->data[bcast_offset(shift, b, c, h, w)] + z;
    }
    return z;
}

/* ---- gather: sige/cpu/gather.cpp:4-58 --------------------------------- */
/* x [B,C,H,W] -> out [B*N,C,R,S]; idx [N,2] (h,w) tile origins in input
 * coordinates; out-of-image elements are exactly 0 (no affine, no act). */
void oracle_gather_f32(const float *x, int B, int C, int H, int W, int R, int S,
                       const int32_t *idx, int N,
                       const float *scale, int sB, int sC, int sH, int sW,
                       const float *shift, int tB, int tC, int tH, int tW,
                       int act, int act_first, float *out) {
    const oracle_bcast4 sc = {scale, sB, sC, sH, sW}, sh = {shift, tB, tC, tH, tW};
    const long total = (long)B * N * C;
#pragma omp parallel for
    for (long job = 0; job < total; ++job) {
        const int c = (int)(job % C);
        const int n = (int)((job / C) % N);
        const int b = (int)(job / ((long)C * N));
        const int h0 = idx[2 * n], w0 = idx[2 * n + 1];
        float *o = out + (((size_t)b * N + n) * C + c) * R * S;
        const float *plane = x + ((size_t)b * C + c) * H * W;
        for (int r = 0; r < R; ++r)
            for (int s = 0; s < S; ++s) {
                const int h = h0 + r, w = w0 + s;
                if (h < 0 || h >= H || w < 0 || w >= W) { o[r * S + s] = 0.0f; continue; }
                o[r * S + s] = affine_act(plane[(size_t)h * W + w], &sc, &sh, act, act_first, b, c, h, w);
            }
    }
}

/* ---- scatter: sige/cpu/scatter.cpp:4-39 (kernel) + 70-109 (clone of y) -- */
static void scatter_tiles(const float *x, int B, int C, int H, int W, int R, int S,
                          int offH, int offW, int strH, int strW,
                          const int32_t *idx, int N, const oracle_bcast4 *res, float *out) {
    const long total = (long)B * N * C;
#pragma omp parallel for
    for (long job = 0; job < total; ++job) {
        const int c = (int)(job % C);
        const int n = (int)((job / C) % N);
        const int b = (int)(job / ((long)C * N));
        const int h0 = (offH + idx[2 * n]) / strH, w0 = (offW + idx[2 * n + 1]) / strW;
        const float *t = x + (((size_t)b * N + n) * C + c) * R * S;
        float *plane = out + ((size_t)b * C + c) * H * W;
        for (int r = 0; r < R && h0 + r < H; ++r)
            for (int s = 0; s < S && w0 + s < W; ++s) {
                const int h = h0 + r, w = w0 + s;
                float z = t[r * S + s];
                if (res->data) z = res->data[bcast_offset(res, b, c, h, w)] + z;
                plane[(size_t)h * W + w] = z;
            }
    }
}

/* x [B*N,C,R,S] tiles, y [B,C,H,W] cached output; out = copy(y) with tiles
 * written at ((off+idx)/stride), clipped bottom/right, + residual. */
void oracle_scatter_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                        int offH, int offW, int strH, int strW, const int32_t *idx, int N,
                        const float *residual, int rB, int rC, int rH, int rW, float *out) {
    const oracle_bcast4 res = {residual, rB, rC, rH, rW};
    memcpy(out, y, sizeof(float) * (size_t)B * C * H * W);
    scatter_tiles(x, B, C, H, W, R, S, offH, offW, strH, strW, idx, N, &res, out);
}

/* ---- scatter_with_block_residual: scatter.cpp:41-68 + 111-135 ---------- */
/* out = scatter(x0, y0, residual=y1) ; then out[p] += x1[tile] - y1[p] over
 * the shortcut tiles (coordinates used directly: no offset, no stride). */
void oracle_scatter_with_block_residual_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offH, int offW, int strH, int strW,
        const int32_t *idx0, int N0, const int32_t *idx1, int N1, float *out) {
    const oracle_bcast4 res = {y1, B, C, H, W};
    memcpy(out, y0, sizeof(float) * (size_t)B * C * H * W);
    scatter_tiles(x0, B, C, H, W, R0, S0, offH, offW, strH, strW, idx0, N0, &res, out);
    const long total = (long)B * N1 * C;
#pragma omp parallel for
    for (long job = 0; job < total; ++job) {
        const int c = (int)(job % C);
        const int n = (int)((job / C) % N1);
        const int b = (int)(job / ((long)C * N1));
        const int h0 = idx1[2 * n], w0 = idx1[2 * n + 1];
        const float *t = x1 + (((size_t)b * N1 + n) * C + c) * R1 * S1;
        const size_t plane = ((size_t)b * C + c) * H * W;
        for (int r = 0; r < R1 && h0 + r < H; ++r)
            for (int s = 0; s < S1 && w0 + s < W; ++s) {
                const size_t p = plane + (size_t)(h0 + r) * W + (w0 + s);
                out[p] += t[r * S1 + s] - y1[p];
            }
    }
}

/* ---- get_scatter_map: sige/cpu/scatter_gather.cpp:58-84 + 150-170 ------- */
/* int32 [H,W,3] = -1 everywhere, then (tile, r, s) on every pixel an OUTPUT
 * tile of the paired conv covers; R=(bH-kH)/strH+1. */
void oracle_get_scatter_map_i32(int H, int W, int bH, int bW, int kH, int kW,
                                int offH, int offW, int strH, int strW,
                                const int32_t *idx, int N, int32_t *map) {
    const int R = (bH - kH) / strH + 1, S = (bW - kW) / strW + 1;
    for (size_t i = 0; i < (size_t)H * W * 3; ++i) map[i] = -1;
    for (int n = 0; n < N; ++n) {
        const int h0 = (offH + idx[2 * n]) / strH, w0 = (offW + idx[2 * n + 1]) / strW;
        for (int r = 0; r < R && h0 + r < H; ++r)
            for (int s = 0; s < S && w0 + s < W; ++s) {
                int32_t *m = map + 3 * ((size_t)(h0 + r) * W + (w0 + s));
                m[0] = n; m[1] = r; m[2] = s;
            }
    }
}

/* ---- scatter_gather: sige/cpu/scatter_gather.cpp:5-56 ------------------- */
/* For every element of the NEXT conv's input tile: source = conv-1 output
 * tile x[(b*N+blk), c, r, s] where map says a tile covers the pixel, else the
 * cached y[b,c,h,w]; then affine+act exactly as gather; 0 outside the image. */
void oracle_scatter_gather_f32(const float *x, const float *y, int B, int C, int H, int W,
                               int Rx, int Sx, int Ro, int So,
                               const int32_t *idx, int N, const int32_t *map,
                               const float *scale, int sB, int sC, int sH, int sW,
                               const float *shift, int tB, int tC, int tH, int tW,
                               int act, int act_first, float *out) {
    const oracle_bcast4 sc = {scale, sB, sC, sH, sW}, sh = {shift, tB, tC, tH, tW};
    const long total = (long)B * N * C;
#pragma omp parallel for
    for (long job = 0; job < total; ++job) {
        const int c = (int)(job % C);
        const int n = (int)((job / C) % N);
        const int b = (int)(job / ((long)C * N));
        const int h0 = idx[2 * n], w0 = idx[2 * n + 1];
        float *o = out + (((size_t)b * N + n) * C + c) * Ro * So;
        for (int r = 0; r < Ro; ++r)
            for (int s = 0; s < So; ++s) {
                const int h = h0 + r, w = w0 + s;
                if (h < 0 || h >= H || w < 0 || w >= W) { o[r * So + s] = 0.0f; continue; }
                const int32_t *m = map + 3 * ((size_t)h * W + w);
                float z;
                if (m[0] >= 0)
                    z = x[((((size_t)b * N + m[0]) * C + c) * Rx + m[1]) * Sx + m[2]];
                else
                    z = y[(((size_t)b * C + c) * H + h) * W + w];
                o[r * So + s] = affine_act(z, &sc, &sh, act, act_first, b, c, h, w);
            }
    }
}

/* ---- reduce_mask: sige/utils.py:8-37 ------------------------------------ */
/* mask [H,W] (bytes, non-zero = edited) -> active tile origins.  The reference
 * pads the mask by `pad` on top/left and by a whole block on bottom/right
 * (utils.py:27), max-pools with kernel=block, stride=stride (floor mode), and
 * lists pooled>0.5 cells in row-major order as stride*i - pad (utils.py:28-32).
 * Returns N; writes at most `cap` pairs to idx (may be NULL to count only). */
int oracle_reduce_mask_i32(const uint8_t *mask, int H, int W, int bH, int bW,
                           int strH, int strW, int padH, int padW, int32_t *idx, int cap) {
    const int gh = (H + padH) / strH + 1; /* floor((H+pad+b - b)/s) + 1 */
    const int gw = (W + padW) / strW + 1;
    int n = 0;
    for (int i = 0; i < gh; ++i)
        for (int j = 0; j < gw; ++j) {
            const int h0 = i * strH - padH, w0 = j * strW - padW;
            int any = 0;
            for (int r = 0; r < bH && !any; ++r) {
                const int h = h0 + r;
                if (h < 0 || h >= H) continue;
                for (int s = 0; s < bW; ++s) {
                    const int w = w0 + s;
                    if (w >= 0 && w < W && mask[(size_t)h * W + w]) { any = 1; break; }
                }
            }
            if (any) {
                if (idx && n < cap) { idx[2 * n] = h0; idx[2 * n + 1] = w0; }
                ++n;
            }
        }
    return n;
}

/* ---- dilate_mask (2-D branch): sige/utils.py:40-61 ---------------------- */
/* OR of the mask shifted by 1..dH rows (both directions) then 1..dW columns of
 * the ORIGINAL mask (not of the running result: the reference ORs `mask`
 * slices into `ret`), i.e. a plus-shaped, not square, structuring element. */
void oracle_dilate_mask_u8(const uint8_t *mask, int H, int W, int dH, int dW, uint8_t *out) {
    for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = mask[i] ? 1 : 0;
    if (dH <= 0 && dW <= 0) return;
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            if (!mask[(size_t)h * W + w]) continue;
            for (int i = 1; i <= dH; ++i) {
                if (h - i >= 0) out[(size_t)(h - i) * W + w] = 1;
                if (h + i < H) out[(size_t)(h + i) * W + w] = 1;
            }
            for (int i = 1; i <= dW; ++i) {
                if (w - i >= 0) out[(size_t)h * W + (w - i)] = 1;
                if (w + i < W) out[(size_t)h * W + (w + i)] = 1;
            }
        }
}

/* ---- one pyramid level of downsample_mask: sige/utils.py:88-118 --------- */
/* Bilinear resize (align_corners=False, the F.interpolate call at utils.py:117)
 * of the running FLOAT mask from [H,W] to [h,w]; source index
 * = max(0, (dst+0.5)*in/out - 0.5), lerp between floor and floor+1 (clamped). */
void oracle_bilinear_resize_f32(const float *src, int H, int W, float *dst, int h, int w) {
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (int i = 0; i < h; ++i) {
        float fy = sy * ((float)i + 0.5f) - 0.5f;
        if (fy < 0.f) fy = 0.f;
        const int y0 = (int)fy, y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float ly = fy - (float)y0, hy = 1.f - ly;
        for (int j = 0; j < w; ++j) {
            float fx = sx * ((float)j + 0.5f) - 0.5f;
            if (fx < 0.f) fx = 0.f;
            const int x0 = (int)fx, x1 = x0 + (x0 < W - 1 ? 1 : 0);
            const float lx = fx - (float)x0, hx = 1.f - lx;
            dst[(size_t)i * w + j] =
                hy * (hx * src[(size_t)y0 * W + x0] + lx * src[(size_t)y0 * W + x1]) +
                ly * (hx * src[(size_t)y1 * W + x0] + lx * src[(size_t)y1 * W + x1]);
        }
    }
}

/* threshold step of downsample_mask (utils.py:107-109):
 * t = min(threshold, max(level) - eps); bit = level > t. */
void oracle_threshold_mask_f32(const float *level, int H, int W, float threshold, float eps, uint8_t *out) {
    float mx = level[0];
    for (size_t i = 1; i < (size_t)H * W; ++i) if (level[i] > mx) mx = level[i];
    float t = mx - eps;
    if (threshold < t) t = threshold;
    for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = level[i] > t ? 1 : 0;
}

/* ---- stacked-block conv: sige/nn/base.py:85-92 (F.conv2d, padding 0) ----- */
/* x [T,Cin,R,S] (*) w [Cout,Cin/groups,kH,kW] + bias -> out [T,Cout,Ro,So],
 * Ro=(R-kH)/strH+1.  Direct fp32 accumulation in (ci,ky,kx) order; PyTorch's
 * own summation order is unspecified, hence the 1e-3 abs tolerance of
 * SURVEY.md section 8(c) for conv-containing paths. */
void oracle_block_conv_f32(const float *x, int T, int Cin, int R, int S,
                           const float *w, const float *bias, int Cout, int kH, int kW,
                           int strH, int strW, int groups, float *out) {
    const int Ro = (R - kH) / strH + 1, So = (S - kW) / strW + 1;
    const int cig = Cin / groups, cog = Cout / groups;
    const long total = (long)T * Cout;
#pragma omp parallel for
    for (long job = 0; job < total; ++job) {
        const int co = (int)(job % Cout);
        const int t = (int)(job / Cout);
        const int g = co / cog;
        for (int oy = 0; oy < Ro; ++oy)
            for (int ox = 0; ox < So; ++ox) {
                float acc = bias ? bias[co] : 0.0f;
                for (int ci = 0; ci < cig; ++ci) {
                    const float *xp = x + (((size_t)t * Cin + g * cig + ci) * R + oy * strH) * S + ox * strW;
                    const float *wp = w + ((size_t)co * cig + ci) * kH * kW;
                    for (int ky = 0; ky < kH; ++ky)
                        for (int kx = 0; kx < kW; ++kx) acc += xp[ky * S + kx] * wp[ky * kW + kx];
                }
                out[(((size_t)t * Cout + co) * Ro + oy) * So + ox] = acc;
            }
    }
}

int oracle_version(void) { return 1; }
