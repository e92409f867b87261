/*
 * sige_hip.h -- C ABI of libsige_hip.so: the MI355X (gfx950) native backend of
 * SIGE's tiling-based sparse convolution path.
 *
 * This is the drop-in boundary for the path: every entry point replaces one
 * function of the reference's native backend surface (the five functions each of
 * sige/cpu/pybind_cpu.cpp:5-12, sige/cuda/pybind_cuda.cpp:5-12 export), plus the
 * index reduction of sige/utils.py:8-37 and the stacked-block convolution the
 * reference delegates to F.conv2d (sige/nn/base.py:85-92).
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types.  All tensors are dense,
 *     contiguous, fp32 (the reference is fp32-only: sige/nn/base.py:15), NCHW like the
 *     reference's -- except in the entry points named *_nhwc_*, which take the same
 *     tensors channels-last ([B,H,W,C], tiles [T,R,S,C]; same arithmetic);
 *     index tensors are int32 [N,2] = (h, w) tile origins in the INPUT
 *     coordinates of the paired conv (sige/utils.py:30-37).
 *   - the CALLER allocates outputs (the reference's wrappers call torch::empty /
 *     y.clone() themselves: gather.cpp:75, scatter.cpp:83); nothing here
 *     allocates, frees or synchronises, so every call is hipGraph-capturable.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the
 *     call returns immediately (the reference launches on the legacy default
 *     stream with no error check: sige/cuda/gather_kernel.cu:113).
 *   - broadcastable operands (scale / shift / residual) are passed as a pointer
 *     (NULL = absent) plus their four dims, each either 1 or the full extent
 *     (binary_op_array, sige/cpu/common_cpu.cpp:13-27).
 *   - return value: SIGE_HIP_OK or a negative SIGE_HIP_E* code; the reference's
 *     asserts / __builtin_unreachable (sige/common.cpp:17-23) become
 *     SIGE_HIP_EINVAL / SIGE_HIP_EUNSUPPORTED.
 */
#ifndef SIGE_HIP_H
#define SIGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIGE_HIP_VERSION 309 /* 0.3.9: round 6 -- tile conv v3 on fp16 operands (configs[4]), write-through epilogue stores */

enum {
    SIGE_HIP_OK = 0,
    SIGE_HIP_EINVAL = -1,       /* bad pointer / negative or inconsistent size */
    SIGE_HIP_EUNSUPPORTED = -2, /* unknown activation, unsupported shape */
    SIGE_HIP_ELAUNCH = -3,      /* hipGetLastError() after the launch was not hipSuccess */
    SIGE_HIP_ENODEVICE = -4     /* no gfx950 device visible to the process */
};

/* activation ids: the two names the reference's native code knows
 * (sige/common.cpp:4,11-23). */
enum { SIGE_HIP_ACT_IDENTITY = 0, SIGE_HIP_ACT_SWISH = 1,
       /* library extensions, accepted only by the entry points that say so (the GauGAN helpers below): */
       SIGE_HIP_ACT_RELU = 2, SIGE_HIP_ACT_LEAKY = 3 /* leaky ReLU with the call's `slope` */, SIGE_HIP_ACT_TANH = 4 };

/* ---- thread-safety ("re-entrant, no global state") ---------------------------------------------------------
 * Every compute entry point (gather / scatter / scatter_gather / scatter_map / reduce_mask / mask pipeline / block_conv /
 * wide_conv / group_norm / attention / spade / conv_in / conv_out ...) is RE-ENTRANT: it reads its arguments, enqueues on
 * the stream it is handed and returns; the kernels a call runs depend on its arguments alone.  The product library keeps
 * no process-global mutable dispatch state.  What state there is:
 *   per HOST THREAD  : sige_hip_set_edit_batch (the stacked-edit factor E of the tensors this thread hands over; a launch
 *                      plan records the call, so a replayed plan restores it), sige_hip_conv_pair_begin / _end (the held
 *                      shortcut conv of a pair), sige_hip_plan_begin / _end (the plan this thread is recording into);
 *   per DEVICE, mutex: the K-split ticket buffer (sige_hip_split_tickets_reset) and sige_hip_preload's loaded-device set;
 *   atomic counters  : sige_hip_launch_count, sige_hip_last_launch_device, sige_hip_conv_pairs_fused (measurement aids);
 *   a plan OBJECT    : one thread at a time per plan (create / record / bind / run / destroy are not internally locked).
 *
 * ---- measurement builds only: dispatch knobs ----------------------------------------------------------------------------
 * Round 4 shipped ten process-global `sige_hip_*_force_*` setters.  They are gone from the product library; a library built
 * with -DSIGE_HIP_TUNING (lib/libsige_hip_tuning.so; tools/, bench sections that compare kernel forms, tests that force a form)
 * exports ONE setter / getter over the keys below (atomic; process-wide; 0 / -1 defaults = the product's behaviour). */
enum {
    SIGE_HIP_TUNE_CONV_TILE_MT = 0,          /* pin the tile conv's output block: 16 | 32 pixels (0 = per launch) */
    SIGE_HIP_TUNE_CONV_TILE_NB = 1,          /* ... x nb * mt output channels, nb 1 | 2 (with CONV_TILE_MT) */
    SIGE_HIP_TUNE_CONV_WAVES = 2,            /* waves per workgroup of the channels-last stride-1 kernels: 4 | 8 (0 = per launch) */
    SIGE_HIP_TUNE_CONV_LARGE_GRID_NB1 = 3,   /* unsplit full grids take 32 x 32 blocks from this many 32 x 64 blocks on (-1 = library: exact fp32 always; 0 never) */
    SIGE_HIP_TUNE_CONV_KSPLIT = 4,           /* cross-workgroup K split: 1..8 = at most this many (0 = per launch) */
    SIGE_HIP_TUNE_CONV_KSPLIT_SECOND_PASS = 5, /* 1 = finish a K split by a second launch instead of inside the launch */
    SIGE_HIP_TUNE_GATHER_ONE_TILE_ROWS = 6,  /* 1 = the NCHW gather always in its one-tile-per-workgroup row form */
    SIGE_HIP_TUNE_SCATTER_GATHER_FORM = 7,   /* bit 0: always the element form; bit 1: never the grouped row form */
    SIGE_HIP_TUNE_SMALL_COUT_SCALAR = 8,     /* 1 = conv3x3_small_cout always on its scalar-weight kernel */
    SIGE_HIP_TUNE_WIDE_KSPLIT = 9,           /* dense-layer conv: pin the K split (0 = automatic) */
    SIGE_HIP_TUNE_ATTENTION_FORM = 10,       /* attention_tokens: 1 = 16 queries per workgroup | 2 = 32 (0 = automatic) */
    SIGE_HIP_TUNE_TILE3_F16_TPW4_MIN = 11,  /* tile conv v3, fp16 operands: 4 tiles per workgroup from this many 2-tile workgroups on (-1 = library; 0 never) */
    SIGE_HIP_TUNE_TILE3_F16_PAIR_MIN = 12,  /* tile conv v3, fp16 operands: a conv1 whose 1x1 shortcut is held for the pair kernel goes to v3 anyway (the shortcut launched on its own) from this many workgroups on (-1 = library; 0 never) */
    SIGE_HIP_TUNE_TILE3_F16_SPARSE_MIN = 13, /* tile conv v3, fp16 operands: launches over a SPARSE tile list (fewer tiles than 4x4 cells) go to v3 from this many workgroups on, below the general threshold (-1 = library; 0 = no separate rule) */
    SIGE_HIP_TUNE_COUNT = 14
};
#ifdef SIGE_HIP_TUNING
int sige_hip_tuning_set(int key, int value); /* SIGE_HIP_EINVAL for an unknown key or a value outside the key's range */
int sige_hip_tuning_get(int key);            /* the current value (the default if never set); INT_MIN for an unknown key */
#endif

/* Load the code objects of every translation unit of this library on the CURRENT device now (HIP loads a code object on the
 * first launch of one of its kernels: 44 of them, 0.3 - 40 ms each, which otherwise lands on the first forward that happens to
 * select a not-yet-used kernel variant -- e.g. a new mask whose tile count picks another block shape).  Idempotent per
 * device; not capturable (call it before capturing).  Returns the number of code objects touched, or a negative status. */
int sige_hip_preload(void);

int sige_hip_version(void);
const char *sige_hip_error_string(int status);
/* name of device 0's gcnArchName ("gfx950...") or NULL if no device. */
const char *sige_hip_device_arch(void);
/* number of GPU kernels this library has launched in the process so far (all entry
 * points, all streams): measurement aid -- launches per forward in bench.py. */
int64_t sige_hip_launch_count(void);
/* HIP's current device at the most recent kernel launch of this library (-1: none yet).  Every entry point launches on the
 * CURRENT device with the stream it is handed (like the reference's sige/cuda wrappers, which have no CUDAGuard either); the
 * host side guards (sige_amd/hip.py: _Guarded).  Debug aid for testing that guard. */
int sige_hip_last_launch_device(void);

/* ---- gather : replaces gather_cpu / gather_cuda -------------------------
 * (sige/cpu/gather.cpp:60-114, sige/cuda/gather_kernel.cu:69-124, gather.h:5-12)
 * x [B,C,H,W] -> out [B*N,C,bH,bW];  out-of-image elements are exactly 0.   */
int sige_hip_gather_f32(const float *x, int B, int C, int H, int W, int bH, int bW,
                        const int32_t *active_indices, int N,
                        const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                        const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                        int activation, int activation_first, float *out, void *stream);

/* ---- scatter : replaces scatter_cpu / scatter_cuda ----------------------
 * (sige/cpu/scatter.cpp:70-109, sige/cuda/scatter_kernel.cu:76-117, scatter.h:5-11)
 * out [B,C,H,W] = y, with the tiles x [B*N,C,R,S] written at (off+idx)/stride
 * (clipped bottom/right) + residual.  `out` must not alias `y`.
 * Works for ARBITRARY index lists (copy pass + tile pass).                   */
int sige_hip_scatter_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                         int offsetH, int offsetW, int strideH, int strideW,
                         const int32_t *active_indices, int N,
                         const float *residual, int resB, int resC, int resH, int resW,
                         float *out, void *stream);

/* ---- scatter_with_block_residual : replaces scatter_with_block_residual_*
 * (sige/cpu/scatter.cpp:111-135, sige/cuda/scatter_kernel.cu:119-146, scatter.h:13-19)
 * out = scatter(x0, y0, residual = y1); out += x1 - y1 on the shortcut tiles
 * idx1 (coordinates used directly).                                          */
int sige_hip_scatter_with_block_residual_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, int N0, const int32_t *active_indices1, int N1,
        float *out, void *stream);

/* ---- tile table + fused single-pass scatter (MI355X-first fast path) ------
 * For index lists produced by reduce_mask, output tiles lie on a regular grid
 * of pitch (R,S) (block_stride = out_tile * conv_stride, sige/nn/gather.py:37),
 * so "which tile covers pixel (h,w)" is a [gH,gW] int32 lookup, gH=ceil(H/R).
 * sige_hip_tile_table_i32 builds it (cells with no tile = -1); the *_fused
 * variants then produce `out` in ONE streaming pass (no clone + overwrite).
 * Results are bit-identical to the two-pass entry points above.             */
int sige_hip_tile_table_i32(const int32_t *active_indices, int N, int offsetH, int offsetW,
                            int strideH, int strideW, int R, int S, int gH, int gW,
                            int32_t *table, void *stream);
int sige_hip_scatter_fused_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                               const int32_t *table, int gH, int gW, int N,
                               const float *residual, int resB, int resC, int resH, int resW,
                               float *out, void *stream);
int sige_hip_scatter_with_block_residual_fused_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *table1, int gH1, int gW1, int N1,
        float *out, void *stream);

/* ---- get_scatter_map : replaces get_scatter_map_cpu / _cuda ---------------
 * (sige/cpu/scatter_gather.cpp:150-170, scatter_gather_kernel.cu:160-188,
 *  scatter_gather.h:16-22).  map int32 [H,W,3] = (tile, r, s) or -1 triples.  */
int sige_hip_scatter_map_i32(int H, int W, int bH, int bW, int kH, int kW,
                             int offsetH, int offsetW, int strideH, int strideW,
                             const int32_t *active_indices, int N, int32_t *map, void *stream);

/* ---- scatter_gather : replaces scatter_gather_cpu / _cuda -----------------
 * (sige/cpu/scatter_gather.cpp:86-148, scatter_gather_kernel.cu:100-158,
 *  scatter_gather.h:5-14).  x [B*N,C,Rx,Sx] conv-1 output tiles, y [B,C,H,W]
 * cached full tensor -> out [B*N,C,bH,bW] (input tiles of the next conv).    */
int sige_hip_scatter_gather_f32(const float *x, const float *y, int B, int C, int H, int W,
                                int Rx, int Sx, int bH, int bW,
                                const int32_t *active_indices, int N, const int32_t *scatter_map,
                                const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                int activation, int activation_first, float *out, void *stream);

/* ---- mask pipeline : replaces sige.utils.compute_difference_mask / dilate_mask /
 * downsample_mask (sige/utils.py:74-85, 40-71, 88-118) for masks that live on the GPU,
 * without the reference's device -> host synchronisation per pyramid level
 * (`min(threshold, level.max() - eps)`, utils.py:107).  All masks are bytes (0 / 1),
 * row-major [H,W].
 *   difference_mask  out[h,w] = any_c |a[c,h,w] - b[c,h,w]| > eps; a / b addressed through
 *                    element strides (NCHW or channels-last; strideH must equal W*strideW)
 *   dilate_mask      OR of the ORIGINAL mask shifted by 1..dilationH rows and 1..dilationW
 *                    columns (a plus-shaped element, utils.py:57-61); out != mask
 *   mask_pyramid     every level of downsample_mask in ONE launch: level 0 = mask, level k+1 =
 *                    bilinear (align_corners=False) halving of the FLOAT level k; each level
 *                    is thresholded at min(threshold, max(level) - eps) and dilated; levels
 *                    are written back to back into `out` (sizes: sige_hip_mask_pyramid_levels,
 *                    which mirrors the loop `h//=2; w//=2; stop when h<min_h and w<min_w`).
 *                    scratch_floats >= (H/2)*(W/2) + (H/4)*(W/4) + (H*W+3)/4 + 8.          */
int sige_hip_difference_mask_u8(const float *a, const float *b, int C, int H, int W,
                                int64_t strideC, int64_t strideH, int64_t strideW, float eps,
                                uint8_t *out, void *stream);
int sige_hip_dilate_mask_u8(const uint8_t *mask, int H, int W, int dilationH, int dilationW,
                            uint8_t *out, void *stream);
int sige_hip_mask_pyramid_levels(int H, int W, int min_h, int min_w, int *hs, int *ws, int capacity);
int sige_hip_mask_pyramid_u8(const uint8_t *mask, int H, int W, int min_h, int min_w,
                             int dilationH, int dilationW, float threshold, float eps,
                             float *scratch, size_t scratch_floats, uint8_t *out, void *stream);

/* ---- reduce_mask : replaces sige.utils.reduce_mask ------------------------
 * (sige/utils.py:8-37: F.pad -> F.max_pool2d -> nonzero -> stride*i - pad).
 * mask: H*W bytes (non-zero = edited).  Writes up to `capacity` (h,w) pairs in
 * row-major order of the candidate grid to `indices` and the TOTAL number of
 * active tiles to *count (device int32).  The candidate grid is
 * ((H+padH)/strH+1) x ((W+padW)/strW+1); sige_hip_reduce_mask_capacity returns
 * its size.                                                                  */
int sige_hip_reduce_mask_capacity(int H, int W, int strideH, int strideW, int padH, int padW);
int sige_hip_reduce_mask_i32(const uint8_t *mask, int H, int W, int bH, int bW,
                             int strideH, int strideW, int padH, int padW,
                             int32_t *indices, int capacity, int32_t *count, void *stream);

/* ---- stacked-block convolution : replaces the F.conv2d call of
 * SIGEConv2d.forward in sparse mode (sige/nn/base.py:88-89) ----------------
 * x [T,Cin,R,S] (*) w [Cout,Cin/groups,kH,kW] + bias -> out [T,Cout,Ro,So],
 * padding 0, dilation 1, Ro=(R-kH)/strH+1.
 * The MFMA path (fp32-in/fp32-acc v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact
 * fp32 products; the library picks the tile by grid size) needs the weights
 * re-laid once per weight tensor (both tile layouts, back to back):
 *   n = sige_hip_block_conv_packed_size(...)   floats to allocate (0 = this shape
 *                                              has no MFMA path; use _direct)
 *   sige_hip_block_conv_pack_f32(w, ..., packed)
 *   sige_hip_block_conv_f32(x, ..., packed, bias, ..., out)
 * sige_hip_block_conv_direct_f32 is the any-shape (groups, odd tiles, dilation:
 * Ro=(R-(kH-1)*dilationH-1)/strH+1) vector FMA kernel reading the original
 * weight layout.                                                              */
size_t sige_hip_block_conv_packed_size(int Cout, int Cin, int kH, int kW, int R, int S,
                                       int strideH, int strideW, int groups);
int sige_hip_block_conv_pack_f32(const float *w, int Cout, int Cin, int kH, int kW,
                                 float *packed, void *stream);
int sige_hip_block_conv_f32(const float *x, int T, int Cin, int R, int S,
                            const float *packed, const float *bias, int Cout, int kH, int kW,
                            int strideH, int strideW, float *out, void *stream);
int sige_hip_block_conv_direct_f32(const float *x, int T, int Cin, int R, int S,
                                   const float *w, const float *bias, int Cout, int kH, int kW,
                                   int strideH, int strideW, int dilationH, int dilationW, int groups,
                                   float *out, void *stream);
/* ---- SPADE modulation of tiles (GauGAN; gaugan/models/sige_normalization.py:62-88) --------
 * out[B*N,bH,bW,C] = leaky(normalized * (1 + gamma) + beta), channels-last, one pass:
 *   normalized = scale*X + shift with X = x_full [B,H,W,C] at the tile's pixels (x_tiles == NULL:
 *                Gather) or, through map_x, the conv tiles x_tiles [B*Nx,Rx,Sx,C] where one covers
 *                the pixel and the cache x_full elsewhere (ScatterGather);
 *   gamma|beta = gb_tiles [B*Ng,Rg,Sg,2C] / gb_full [B,H,W,2C] through map_g the same way.
 * Pixels outside the image give 0 (as Gather / ScatterGather zero-fill both operands).  leaky != 0
 * applies x > 0 ? x : slope*x.  Same fp32 operations, in the same order, as the module chain
 * Gather | ScatterGather, ScatterGather, split, 1 + gamma, *, + beta, leaky_relu.            */
int sige_hip_spade_modulate_nhwc_f32(
        const float *x_full, const float *x_tiles, const int32_t *map_x, int Nx, int Rx, int Sx,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC,
        const float *gb_tiles, const float *gb_full, const int32_t *map_g, int Ng, int Rg, int Sg,
        int B, int C, int H, int W, int bH, int bW, const int32_t *active_indices, int N,
        int leaky, float slope, float *out, void *stream);

/* ---- f16 compute ("_f16c"): the same stacked-block convs on the fp16 matrix cores ---------
 * (v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16, 16x the rate of the f32-input forms).
 * Tensors stay fp32 in HBM (x, y, residual, out, bias, scale / shift): the staging path
 * finishes a value in fp32 (cached GroupNorm affine + SiLU), rounds it to fp16 (RNE) into LDS;
 * weights are packed as fp16 by sige_hip_block_conv_pack_f16c (same fp32 weight tensor in);
 * products are exact and accumulate in fp32.  Not in the reference, which is fp32-only
 * (sige/nn/base.py:15,55-63): BASELINE.json configs[4]; parity vs the fp32 oracle is quoted at
 * 2e-2 abs.  Channels-last entry points only; geometries: 3x3/s1 on 6x6 and
 * 1x1 on 4x4 (packed_size_f16c returns 0 for the stride-2 geometry: pack that conv for fp32).
 * `packed` sizes are in 4-byte units as for the fp32 forms; arguments as the _f32 functions
 * of the same name (declared further down).                                              */
size_t sige_hip_block_conv_packed_size_f16c(int Cout, int Cin, int kH, int kW, int R, int S,
                                            int strideH, int strideW, int groups);
int sige_hip_block_conv_pack_f16c(const float *w, int Cout, int Cin, int kH, int kW,
                                  float *packed, void *stream);
int sige_hip_block_conv_nhwc_f16c(const float *x, int T, int Cin, int R, int S,
                                  const float *packed, const float *bias, int Cout, int kH, int kW,
                                  int strideH, int strideW, float *out, void *stream);
int sige_hip_gather_conv_nhwc_f16c(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                   int bH, int bW, const int32_t *active_indices, int N,
                                   const float *scale, int scaleB, int scaleC,
                                   const float *shift, int shiftB, int shiftC,
                                   int activation,
                                   const float *packed, const float *bias, int Cout, int kH, int kW,
                                   int strideH, int strideW,
                                   int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                   float *workspace, size_t workspace_floats,
                                   const float *out_scale, const float *out_shift, int out_activation,
                                   int upsample2x,
                                   float *twin0, const float *twin0_scale, const float *twin0_shift,
                                   float *twin1, const float *twin1_scale, const float *twin1_shift,
                                   float *out, void *stream);
int sige_hip_scatter_gather_conv_nhwc_f16c(const float *x, const float *y, int B, int Cin, int H, int W,
                                           int Rx, int Sx, int bH, int bW,
                                           const int32_t *active_indices, int N, const int32_t *scatter_map,
                                           const float *scale, int scaleB, int scaleC,
                                           const float *shift, int shiftB, int shiftC,
                                           int activation,
                                           const float *packed, const float *bias, int Cout, int kH, int kW,
                                           int strideH, int strideW, float *out, void *stream);
int sige_hip_scatter_gather_conv_scatter_nhwc_f16c(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream);

/* Horizontal fusion of the two independent convs at the head of a residual block.  After pair_begin() the next
 * channels-last fp32 1x1 gather -> conv launch with raw staging (the block's shortcut) is HELD: the call returns
 * SIGE_HIP_OK without launching.  The next channels-last fp32 3x3/s1 gather -> conv launch with affine + SiLU staging on
 * the same stream and with the same destination kind (the block's conv1) then runs both in ONE kernel (workgroups of
 * both convs side by side; the 1x1's own ~5 us launch disappears).  Any other conv launch, and pair_end(), launch the held
 * conv on its own first, so results never depend on whether a pair was formed.  Per host thread; the pointers of the
 * held call must stay valid until it has been launched.  pairs_fused(): how many pairs this process has formed (atomic). */
int sige_hip_conv_pair_begin(void);
int sige_hip_conv_pair_end(void);
int64_t sige_hip_conv_pairs_fused(void);
/* A K-split launch is finished inside the launch: the last workgroup to finish an output block adds the partial copies of
 * that block up in split order and runs the epilogue (tickets in a library-owned, per-device buffer); when no tickets are to
 * be had, by a second launch over the whole output.  Both add in the same order: identical results. */
/* Tickets of launches made while a stream is being CAPTURED into a hipGraph are bump-allocated (a replayed graph owns its
 * tickets for its lifetime) out of 3 M per device; when they run out, captured K-split launches fall back to the second-pass
 * finish (still correct, one more launch).  A long-running host that re-captures per mask calls this once every hipGraph
 * captured so far on the CURRENT device has been destroyed: the region is handed out again from the start. */
int sige_hip_release_graph_tickets(void);

/* ---- fused gather -> conv and scatter_gather -> conv ------------------------
 * The same MFMA conv with the producer of its input tiles fused into the
 * prologue: the [B*N,Cin,bH,bW] tile tensor of gather (sige/cpu/gather.cpp:4-58)
 * resp. scatter_gather (sige/cpu/scatter_gather.cpp:5-56) is staged straight
 * into LDS and never written to HBM.  Equivalent to
 *     sige_hip_gather_f32(...)          ; sige_hip_block_conv_f32(...)
 *     sige_hip_scatter_gather_f32(...)  ; sige_hip_block_conv_f32(...)
 * for activation_first = 0 and a per-(batch,channel) affine (scale/shift dims
 * [1|B, 1|C, 1, 1], which is what every caller in the reference passes:
 * sige_fused_unet.py:111,118-120); anything else -> SIGE_HIP_EUNSUPPORTED and
 * the caller uses the two-call form.  out [B*N,Cout,Ro,So].                   */
int sige_hip_gather_conv_f32(const float *x, int B, int Cin, int H, int W, int bH, int bW,
                             const int32_t *active_indices, int N,
                             const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                             const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                             int activation,
                             const float *packed, const float *bias, int Cout, int kH, int kW,
                             int strideH, int strideW, float *out, void *stream);
int sige_hip_scatter_gather_conv_f32(const float *x, const float *y, int B, int Cin, int H, int W,
                                     int Rx, int Sx, int bH, int bW,
                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                     const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                     const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                     int activation,
                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                     int strideH, int strideW, float *out, void *stream);

/* ---- dense layers through the same kernel ("all tiles active") ---------------
 * the U-Net's low-resolution dense blocks in sparse mode
 * (sige_fused_unet.py:112-114,121-123: h*scale+shift -> swish -> nn.Conv2d ->
 * + skip) as ONE launch: gather-conv whose input channels may come from two
 * tensors (x [B,C1,H,W] and x2 [B,C2,H,W]: a fused torch.cat), whose output
 * tiles are written straight into out [B,Cout,Ho,Wo] at (offset+idx)/stride
 * (clipped) plus an optional residual [B,Cout,Ho,Wo].  With the index list of
 * ALL tiles this is a dense conv with padding = offset.  scale/shift: [1|B, 1|Cin]. */
int sige_hip_gather_conv_nchw_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                  int bH, int bW, const int32_t *active_indices, int N,
                                  const float *scale, int scaleB, int scaleC,
                                  const float *shift, int shiftB, int shiftC,
                                  int activation,
                                  const float *packed, const float *bias, int Cout, int kH, int kW,
                                  int strideH, int strideW, int offsetH, int offsetW,
                                  const float *residual, int Ho, int Wo, float *out, void *stream);

/* ---- channels-last (NHWC) forms of the fused convolutions ----------------------
 * Same arithmetic, tensors laid out [B,H,W,C] (torch.channels_last) and tiles
 * [T,R,S,C].  In the reference's NCHW layout a gathered 6x6 window is 6 separate
 * 24-byte segments per channel; with the channels contiguous every staging load is a
 * coalesced 16 bytes per lane.  Requirements: channel counts
 * multiples of 4, 16-byte aligned pointers; scale/shift [1|B, 1|C] as above.
 *   sige_hip_block_conv_nhwc_f32            x [T,R,S,Cin]              -> out [T,Ro,So,Cout]
 *   sige_hip_gather_conv_nhwc_f32           x [B,H,W,C1] (+ x2 [1,H,W,C2]: a fused cat)
 *        to_full = 0 -> out [B*N,Ro,So,Cout];  to_full = 1 -> out [B,Ho,Wo,Cout] written at
 *        (offset+idx)/stride, clipped, + residual [B,Ho,Wo,Cout] (dense layers)
 *   sige_hip_scatter_gather_conv_nhwc_f32   x [B*N,Rx,Sx,Cin] tiles, y [B,H,W,Cin] -> out [B*N,Ro,So,Cout] */
int sige_hip_block_conv_nhwc_f32(const float *x, int T, int Cin, int R, int S,
                                 const float *packed, const float *bias, int Cout, int kH, int kW,
                                 int strideH, int strideW, float *out, void *stream);
int sige_hip_gather_conv_nhwc_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                  int bH, int bW, const int32_t *active_indices, int N,
                                  const float *scale, int scaleB, int scaleC,
                                  const float *shift, int shiftB, int shiftC,
                                  int activation,
                                  const float *packed, const float *bias, int Cout, int kH, int kW,
                                  int strideH, int strideW,
                                  int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                  float *workspace, size_t workspace_floats,
                                  const float *out_scale, const float *out_shift, int out_activation,
                                  int upsample2x,
                                  float *twin0, const float *twin0_scale, const float *twin0_shift,
                                  float *twin1, const float *twin1_scale, const float *twin1_shift,
                                  float *out, void *stream);
/* `out_scale` / `out_shift` ([Cout], optional): epilogue out = act(out_scale * (conv + bias + residual) + out_shift)
 * -- the CONSUMER's cached GroupNorm affine + SiLU applied by the producer, once per element; the consumer then
 * gathers with no affine (e.g. conv1 -> conv2 of a ResBlock, sige_fused_unet.py:112-125).
 * `upsample2x` = 1: x / x2 are [B,H/2,W/2,C] and the gather reads pixel (h/2, w/2): the x2 nearest-neighbour
 * upsampling in front of the U-Net's Upsample conv (sige_fused_unet.py: F.interpolate) fused into the gather. */
/* `workspace` (optional, NULL = none): room for up to 8 copies of the output.  When the conv has
 * too few tiles to cover the chip with ANY block shape (the 8x8 layers: 64 pixels, K up to 9216),
 * the channel chunks are split across workgroups that write partial sums there, and a second
 * launch adds them in a fixed order with bias / residual (deterministic; no atomics).
 * sige_hip_conv_ksplit_hint: how many copies such a call would use (1 = no split). */
int sige_hip_conv_ksplit_hint(int T, int Cin, int Cout, int kH, int kW, int strideH, int strideW);
int sige_hip_scatter_gather_conv_nhwc_f32(const float *x, const float *y, int B, int Cin, int H, int W,
                                          int Rx, int Sx, int bH, int bW,
                                          const int32_t *active_indices, int N, const int32_t *scatter_map,
                                          const float *scale, int scaleB, int scaleC,
                                          const float *shift, int shiftB, int shiftC,
                                          int activation,
                                          const float *packed, const float *bias, int Cout, int kH, int kW,
                                          int strideH, int strideW, float *out, void *stream);

/* conv2 -> Scatter / ScatterWithBlockResidual fused (in-place scatter mode): the scatter_gather-fed 3x3 conv writes
 * out[b, (offset+idx)/1 + r, ..., :] = conv + bias + residual  straight into `out` [B,H,W,Cout], a buffer that already
 * equals the cached tensor outside this mask's tiles.  x1 != NULL: block residual -- `residual` is the cached shortcut
 * tensor y1 and out += x1 - y1 wherever a shortcut tile (table1 over R1 x S1 cells, tiles x1 [B*N1,R1,S1,Cout]) covers
 * the pixel (scatter.cpp:41-68).  The shortcut tiles must lie inside the main tiles (true for index lists that
 * reduce_mask derives from one mask: a 4x4 block hit by the mask is inside an active 6x6 window). */
int sige_hip_scatter_gather_conv_scatter_nhwc_f32(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream);

/* ---- channels-last forms of gather / scatter_gather / scatter ------------------
 * (materialising forms: the tiles are written to HBM; the fused convs above do not
 * need them).  scale / shift: [1|B, C].  Results are bit-identical to the NCHW
 * entry points on the same values.
 *   sige_hip_scatter_nhwc_f32 / sige_hip_scatter_with_block_residual_nhwc_f32:
 *     in_place = 0  reference semantics: `out` [B,H,W,C] is a fresh tensor, written in
 *                   ONE pass (cached tensor outside the tiles, tiles + residual inside);
 *     in_place = 1  `out` is a buffer that already holds the cached tensor y outside
 *                   the tiles of this mask (kept by the caller across calls): only the
 *                   covered pixels are written -- traffic ~ active tiles, not B*H*W*C.
 *   `table` / gH / gW: the tile table of sige_hip_tile_table_i32.                 */
int sige_hip_gather_nhwc_f32(const float *x, int B, int C, int H, int W, int bH, int bW,
                             const int32_t *active_indices, int N,
                             const float *scale, int scaleB, int scaleC,
                             const float *shift, int shiftB, int shiftC,
                             int activation, float *out, void *stream);
int sige_hip_scatter_gather_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W,
                                     int Rx, int Sx, int bH, int bW,
                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                     const float *scale, int scaleB, int scaleC,
                                     const float *shift, int shiftB, int shiftC,
                                     int activation, float *out, void *stream);
int sige_hip_scatter_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W, int R, int S,
                              int offsetH, int offsetW, int strideH, int strideW,
                              const int32_t *active_indices, const int32_t *table, int gH, int gW, int N,
                              const float *residual, int in_place, float *out, void *stream);
int sige_hip_scatter_with_block_residual_nhwc_f32(
        const float *x0, const float *y0, const float *x1, const float *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *active_indices1, const int32_t *table1, int gH1, int gW1, int N1,
        int in_place, float *out, void *stream);

/* ---- tile conv v3 (csrc/conv_tile3.hpp): the 3x3 / stride-1 stacked-block conv over 6x6 tiles with the dense-layer kernel's
 * K loop -- wave-private stages (no workgroup barrier per channel chunk), 64 output channels per workgroup -- for grids that
 * fill the chip (large edits, stacked edits); exact fp32 (v_mfma_f32_32x32x2_f32).  Replaces, on those grids, what
 * sige_hip_gather_conv_nhwc_f32 (source 1) and sige_hip_scatter_gather_conv[_scatter]_nhwc_f32 (source 2) do:
 *   source 1: tiles of x [B,H>>up,W>>up,C1] (+ x2 [.., C2]: a fused torch.cat) at active_indices, zero padded, optional cached
 *             affine [affineB in {1,B}, C1+C2] + SiLU;   source 2: x = conv tiles [B*N,Rx,Sx,C1], x2 = the cached tensor
 *             [B,H,W,C1] through scatter_map (raw);
 *   to_full 0: out = tiles [B*N,4,4,Cout];  1: straight into out [B,Ho,Wo,Cout] at offset + origin, clipped, + residual, with x1 /
 *             table1: + (x1 - residual) where a shortcut tile covers the pixel (ScatterWithBlockResidual), twins, out-affine.
 * `packed` = sige_hip_wide_conv_pack(prec = 2) of the [Cout, C1+C2, 3, 3] weight; C1, C2, Cout multiples of 64. */
int sige_hip_tile_conv3_supported(int C1, int C2, int Cout);
int sige_hip_tile_conv3_nhwc_f32(
        int source, const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
        const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
        const float *scale, const float *shift, int affineB, int activation,
        const float *packed, const float *bias, int Cout,
        int to_full, int offsetH, int offsetW, int Ho, int Wo, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        const float *out_scale, const float *out_shift, int out_activation,
        float *twin0, const float *twin_scale0, const float *twin_shift0,
        float *twin1, const float *twin_scale1, const float *twin_shift1,
        float *out, void *stream);

/* ... and the two entry points the sparse forward actually calls: sige_hip_gather_conv_nhwc_f32 / sige_hip_scatter_gather_conv_scatter_nhwc_f32
 * with the weights in the v3 layout (`packed_tile3`, may be NULL) and a threshold beside them.  A launch whose v3 grid (tile pairs x
 * 64-channel output blocks) has >= min_blocks workgroups -- and that is not about to share its launch with a held 1x1 shortcut
 * (sige_hip_conv_pair_begin) -- runs on the v3 kernel; every other launch exactly as the plain entry point.  The decision is
 * taken HERE, from N, so that a launch plan (which replays the recorded entry point with the new mask's count) routes like the
 * module-level forward under that mask: the two stay bit-identical.  min_blocks <= 0 or packed_tile3 == NULL: never. */
int sige_hip_gather_conv_nhwc_v3_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                     int bH, int bW, const int32_t *active_indices, int N,
                                     const float *scale, int scaleB, int scaleC,
                                     const float *shift, int shiftB, int shiftC,
                                     int activation,
                                     const float *packed, const float *bias, int Cout, int kH, int kW,
                                     int strideH, int strideW,
                                     int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                     float *workspace, size_t workspace_floats,
                                     const float *out_scale, const float *out_shift, int out_activation,
                                     int upsample2x,
                                     float *twin0, const float *twin0_scale, const float *twin0_shift,
                                     float *twin1, const float *twin1_scale, const float *twin1_shift,
                                     const float *packed_tile3, int min_blocks,
                                     float *out, void *stream);
int sige_hip_scatter_gather_conv_scatter_nhwc_v3_f32(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        const float *packed_tile3, int min_blocks,
        float *out, void *stream);

/* ... and the fp16-operand form of the three (round 6; BASELINE.json configs[4]): `packed` / `packed_tile3` =
 * sige_hip_wide_conv_pack(prec = 0) of the weight; activations fp32 in HBM, rounded to fp16 (RNE) in the staging path, products exact
 * in fp32, accumulation fp32 (v_mfma_f32_32x32x16_f16).  y_f16: the cached tensor of source 2 (x2 / y) holds halves; residual_f16:
 * `residual` holds halves -- the fp16-stored caches of the _c16 entry points.  The two routing entry points replace
 * sige_hip_gather_conv_nhwc_f16c and sige_hip_scatter_gather_conv_scatter_nhwc_f16c / _c16(compute = 1). */
int sige_hip_tile_conv3_nhwc_f16c(
        int source, const float *x, const void *x2, int y_f16, int B, int C1, int C2, int H, int W, int upsample2x,
        const int32_t *active_indices, int N, const int32_t *scatter_map, int Rx, int Sx,
        const float *scale, const float *shift, int affineB, int activation,
        const float *packed, const float *bias, int Cout,
        int to_full, int offsetH, int offsetW, int Ho, int Wo, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        const float *out_scale, const float *out_shift, int out_activation,
        float *twin0, const float *twin_scale0, const float *twin_shift0,
        float *twin1, const float *twin_scale1, const float *twin_shift1,
        float *out, void *stream);
int sige_hip_gather_conv_nhwc_v3_f16c(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                      int bH, int bW, const int32_t *active_indices, int N,
                                      const float *scale, int scaleB, int scaleC,
                                      const float *shift, int shiftB, int shiftC,
                                      int activation,
                                      const float *packed, const float *bias, int Cout, int kH, int kW,
                                      int strideH, int strideW,
                                      int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                      float *workspace, size_t workspace_floats,
                                      const float *out_scale, const float *out_shift, int out_activation,
                                      int upsample2x,
                                      float *twin0, const float *twin0_scale, const float *twin0_shift,
                                      float *twin1, const float *twin1_scale, const float *twin1_shift,
                                      const float *packed_tile3, int min_blocks,
                                      float *out, void *stream);
int sige_hip_scatter_gather_conv_scatter_nhwc_v3_f16c(
        const float *x, const void *y, int y_f16, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        const float *packed_tile3, int min_blocks,
        float *out, void *stream);

/* ---- token-matrix helpers of Stable Diffusion's spatial transformer (csrc/token_ops.hip; sige_attention.py:86-185): tokens [T,C]
 * row-major fp32.  add_layer_norm: y = x (+ delta + bias[c] when delta != NULL; bias may be NULL); sum_out = y when not NULL;
 * out = LayerNorm(y) * gamma + beta (eps as nn.LayerNorm's).  geglu: out[t,d] = x[t,d] * gelu(x[t,D+d]), x [T,2D] (F.gelu's erf
 * form).  add_bias: out = x + delta + bias[c] (bias may be NULL). */
int sige_hip_add_layer_norm_tokens_f32(const float *x, const float *delta, const float *bias, const float *gamma,
                                       const float *beta, int64_t T, int C, float eps, float *sum_out, float *out, void *stream);
int sige_hip_geglu_tokens_f32(const float *x, int64_t T, int D, float *out, void *stream);
int sige_hip_add_bias_tokens_f32(const float *x, const float *delta, const float *bias, int64_t T, int C, float *out, void *stream);

/* ---- 3x3 / padding-1 conv with <= 4 output channels over a full channels-last tensor
 * (the U-Net's conv_out after norm_out + SiLU, sige_fused_unet.py:430-434, which the
 * reference runs densely in sparse mode too): out [B,H,W,Cout] = conv(act(scale*x + shift)),
 * x [B,H,W,C], weight [Cout,C,3,3] (the nn.Conv2d layout), scale / shift [1|B, C] or NULL. */
int sige_hip_conv3x3_small_cout_nhwc_f32(const float *x, int B, int C, int H, int W,
                                         const float *scale, int scaleB, int scaleC,
                                         const float *shift, int shiftB, int shiftC, int activation,
                                         const float *weight, const float *bias, int Cout,
                                         float *out, void *stream);

/* ... with an activation in front (SIGE_HIP_ACT_IDENTITY | _SWISH | _LEAKY with `slope`) and one behind (_IDENTITY | _TANH): GauGAN's
 * tanh(conv_img(leaky_relu(x, 0.2))) (gaugan/models/spade_generators/sige_fused_spade_generator.py:259-260) in one launch. */
int sige_hip_conv3x3_small_cout_act_nhwc_f32(const float *x, int B, int C, int H, int W, int activation, float slope,
                                             const float *weight, const float *bias, int Cout, int out_activation,
                                             float *out, void *stream);

/* ---- channels-last helpers of the GauGAN SPADE generator's sparse forward (csrc/spade_ops.hip): what was left to torch kernels
 * in round 4, so that the whole forward goes through this library and a launch plan can record it.
 * resize_nearest: F.interpolate(mode="nearest") by an INTEGER factor up or down, x [B,H,W,C] -> out [B,Ho,Wo,C]
 *   (sige_fused_spade_generator.py:143-146 the label map per block, :243-257 the x2 up-sampling between blocks).
 * act_split: out[part] = act(x[..., part*C/parts : (part+1)*C/parts]) as `parts` dense [pixels, C/parts] tensors one after
 *   the other, `part_stride` floats apart (>= pixels * C/parts: a launch plan sizes it for every candidate tile, so the parts'
 *   addresses do not move with the mask) -- ReLU + torch.split of a block's label features; act in IDENTITY | RELU | LEAKY.
 * scatter_gather_split: scatter_gather (sige/cpu/scatter_gather.cpp:5-56; no affine) + the same act + split in one pass;
 *   out = `parts` dense [B*N,bH,bW,C/parts] tile slabs.
 * spade_modulate_dense: out = leaky?((scale*x + shift) * (1 + gamma) + beta) on a FULL tensor, gb [B,H,W,2C] = gamma | beta,
 *   scale / shift [affineB in {1, B}, C] (sige_normalization.py:74-88 for the blocks below num_sparse_layers). */
int sige_hip_resize_nearest_nhwc_f32(const float *x, int B, int C, int H, int W, int Ho, int Wo, float *out, void *stream);
int sige_hip_act_split_nhwc_f32(const float *x, int64_t pixels, int C, int parts, int64_t part_stride, int activation, float slope,
                                float *out, void *stream);
int sige_hip_scatter_gather_split_nhwc_f32(const float *x, const float *y, int B, int C, int H, int W, int Rx, int Sx, int bH,
                                           int bW, const int32_t *active_indices, int N, const int32_t *scatter_map,
                                           int activation, float slope, int parts, int64_t part_stride, float *out,
                                           void *stream);
int sige_hip_spade_modulate_dense_nhwc_f32(const float *x, const float *scale, const float *shift, int affineB, const float *gb,
                                           int B, int C, int H, int W, int leaky, float slope, float *out, void *stream);

/* ---- 3x3 / padding-1 conv with <= 3 input channels and Cout = 32 / 64 / 128 over a full image
 * (the U-Net's conv_in, sige_fused_unet.py:395: a plain nn.Conv2d in every mode):
 * out [B,H,W,Cout] (channels-last) = conv(x) + bias; x is addressed through its element strides
 * (NCHW or channels-last), weight [Cout,Cin,3,3]. */
int sige_hip_conv3x3_small_cin_nhwc_f32(const float *x, int64_t strideB, int64_t strideC, int64_t strideH, int64_t strideW,
                                        int B, int Cin, int H, int W,
                                        const float *weight, const float *bias, int Cout,
                                        float *out, void *stream);
/* ... evaluated only on the bH x bW windows at active_indices (clipped), written in place into `out` [B,H,W,Cout]; other pixels of
 * `out` keep their contents (round 6: in sparse mode the first conv's output is only read through Gather windows). */
int sige_hip_conv3x3_small_cin_tiles_nhwc_f32(const float *x, int64_t strideB, int64_t strideC, int64_t strideH, int64_t strideW,
                                              int B, int Cin, int H, int W,
                                              const float *weight, const float *bias, int Cout,
                                              const int32_t *active_indices, int N, int bH, int bW,
                                              float *out, void *stream);

/* per-group mean / rstd of a [B,C,H,W] tensor -> per-channel (scale, shift) with
 * GroupNorm(x) == x*scale + shift  (scale = gamma*rstd, shift = beta - mean*scale):
 * the producer of the cached affine (diffusion/models/common.py:37-57) as two
 * streaming launches.  `workspace`: 2*B*groups*ceil(...) floats, see
 * sige_hip_group_norm_affine_workspace. */
size_t sige_hip_group_norm_affine_workspace(int B, int C, int H, int W, int groups);
int sige_hip_group_norm_affine_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                   const float *gamma, const float *beta, float *workspace,
                                   float *scale, float *shift, void *stream);

/* channels-last form: x [B,H,W,C] */
size_t sige_hip_group_norm_affine_nhwc_workspace(int B, int C, int H, int W, int groups);
int sige_hip_group_norm_affine_nhwc_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                        const float *gamma, const float *beta, float *workspace,
                                        float *scale, float *shift, void *stream);

/* the same with a per-channel bias [C] added BEFORE the statistics (a residual block's GroupNorm of h + temb,
 * diffusion/models/ddpm_arch/sige_fused_unet.py:116-117); the affine is returned for x itself:
 * GroupNorm(x + channel_bias) == x * scale + shift.  One batch-independent bias vector. */
int sige_hip_group_norm_affine_nhwc_bias_f32(const float *x, int B, int C, int H, int W, int groups, float eps,
                                             const float *gamma, const float *beta, const float *channel_bias,
                                             float *workspace, float *scale, float *shift, void *stream);

/* the same affine from per-channel statistics instead of the tensor: statsK [B * tilesK, CK, 2] = per pixel block and channel
 * (sum, sum of squares) as sige_hip_wide_conv_nhwc leaves them (tilesK blocks per batch element, countK = pixels per batch
 * element the sums run over).  Two parts = the GroupNorm of torch.cat([a, b], 1) over C1 + C2 channels without the cat
 * (stats2 NULL / C2 0: one tensor; a group may straddle the parts).  channel_bias [C1 + C2] or NULL as above.  One launch. */
int sige_hip_group_norm_affine_from_stats_f32(const float *stats1, int tiles1, int C1, int count1,
                                              const float *stats2, int tiles2, int C2, int count2,
                                              int B, int groups, float eps, const float *gamma, const float *beta,
                                              const float *channel_bias, float *scale, float *shift, void *stream);

/* per-channel statistics of a channels-last tensor x [B,H,W,C] in that layout, for a tensor whose producer left none:
 * stats [B * sige_hip_channel_stats_tiles(H, W), C, 2], count = H * W.  One pass over x. */
int sige_hip_channel_stats_tiles(int H, int W);
int sige_hip_channel_stats_nhwc_f32(const float *x, int B, int C, int H, int W, float *stats, void *stream);

/* ---- single-head spatial self-attention of the U-Net's dense AttnBlock ------
 * (diffusion/models/ddpm_arch/unet.py AttnBlock.forward, reached from
 * sige_fused_unet.py:186-199): qkv [B,3C,HW] = q, k, v stacked on the channel axis,
 * out[b,c,i] = sum_j v[b,c,j] * softmax_j(scale * sum_c' q[b,c',i] k[b,c',j]).
 * Two launches (score tiles, then softmax + value product); `workspace` holds the
 * [B,HW,HW] scores (sige_hip_attention_workspace floats).  HW and C multiples of 16. */
size_t sige_hip_attention_workspace(int B, int C, int HW);
int sige_hip_attention_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                           float *out, void *stream);
/* channels-last form: qkv [B,HW,3C] -> out [B,HW,C]; C % 64 == 0 */
int sige_hip_attention_nhwc_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                                float *out, void *stream);
/* the same in ONE launch (round 4): workgroup = 16 queries x 64 keys, exact fp32 MFMA scores, the 64-key slices of a query
 * block combined by the last one to finish (flash-decoding split; the tickets of the conv kernels' K-split finish) -- no score
 * tensor in HBM, 6 launches fewer per DDPM forward.  C in {64, 128, 256, 512}, HW % 16 == 0, HW <= 1024; `workspace` holds the
 * per-slice partial outputs (sige_hip_attention_fused_workspace floats; 0 = shape unsupported).  SIGE_HIP_EUNSUPPORTED: use
 * sige_hip_attention_nhwc_f32. */
size_t sige_hip_attention_fused_workspace(int B, int C, int HW);
int sige_hip_attention_fused_nhwc_f32(const float *qkv, int B, int C, int HW, float scale, float *workspace,
                                      float *out, void *stream);

/* ---- split fp16 operands ("_f16x3"): fp32-level results from the fp16 matrix cores -------------------
 * The same entry points once more: every fp32 operand (staged activation after the cached affine + SiLU, weight) is
 * carried as an fp16 pair hi = fp16(v), lo = fp16(v - hi); hi*hi + lo*hi + hi*lo accumulate in fp32 (22-bit operands;
 * the dropped lo*lo term is 2^-22 relative).  Results are inside the fp32 path's 1e-3 (measured ~1e-6 relative against an
 * fp64 conv) at a third of the fp16 matrix rate = 5.3x the f32-input rate.  sige_hip_block_conv_pack_f16x3 scales the
 * weights by a power of two chosen on the device (max |w| * 2^S in [2^13, 2^14): lo parts stay normal fp16 numbers) and
 * stores 2^-S behind the packed data; no host synchronisation.  Geometries and arguments as the _f16c functions.  */
size_t sige_hip_block_conv_packed_size_f16x3(int Cout, int Cin, int kH, int kW, int R, int S,
                                            int strideH, int strideW, int groups);
int sige_hip_block_conv_pack_f16x3(const float *w, int Cout, int Cin, int kH, int kW,
                                  float *packed, void *stream);
int sige_hip_block_conv_nhwc_f16x3(const float *x, int T, int Cin, int R, int S,
                                  const float *packed, const float *bias, int Cout, int kH, int kW,
                                  int strideH, int strideW, float *out, void *stream);
/* ... over the tiles of an index list, T = B * N: `count_key` is that list's pointer (not read); a launch plan looks N up under
 * it, so a conv over a tile SLAB follows a new mask like the gather-type entry points do.  compute: 0 fp32 | 1 f16c | 2 f16x3. */
int sige_hip_block_conv_nhwc_keyed(int compute, const float *x, const int32_t *count_key, int B, int N, int Cin, int R, int S,
                                   const float *packed, const float *bias, int Cout, int kH, int kW,
                                   int strideH, int strideW, float *out, void *stream);
int sige_hip_gather_conv_nhwc_f16x3(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                   int bH, int bW, const int32_t *active_indices, int N,
                                   const float *scale, int scaleB, int scaleC,
                                   const float *shift, int shiftB, int shiftC,
                                   int activation,
                                   const float *packed, const float *bias, int Cout, int kH, int kW,
                                   int strideH, int strideW,
                                   int to_full, int offsetH, int offsetW, const float *residual, int Ho, int Wo,
                                   float *workspace, size_t workspace_floats,
                                   const float *out_scale, const float *out_shift, int out_activation,
                                   int upsample2x,
                                   float *twin0, const float *twin0_scale, const float *twin0_shift,
                                   float *twin1, const float *twin1_scale, const float *twin1_shift,
                                   float *out, void *stream);
int sige_hip_scatter_gather_conv_nhwc_f16x3(const float *x, const float *y, int B, int Cin, int H, int W,
                                           int Rx, int Sx, int bH, int bW,
                                           const int32_t *active_indices, int N, const int32_t *scatter_map,
                                           const float *scale, int scaleB, int scaleC,
                                           const float *shift, int shiftB, int shiftC,
                                           int activation,
                                           const float *packed, const float *bias, int Cout, int kH, int kW,
                                           int strideH, int strideW, float *out, void *stream);
int sige_hip_scatter_gather_conv_scatter_nhwc_f16x3(
        const float *x, const float *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const float *residual,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream);

/* ---- dense layers on the fp16 matrix cores ("wide" conv: 8x8 pixels x 64 output channels per workgroup) -------
 * The layers a SIGE network runs DENSELY -- below `sparse_resolution_threshold` in the sparse pass
 * (diffusion/models/ddpm_arch/sige_fused_unet.py:112-123) and every conv of the cache-producing full pass
 * (sige/nn/base.py:85-86; diffusion/samplers/ddim_ddpm_sampler.py:60-66) -- as ONE launch each:
 *
 *   out = out_act(out_scale * (conv(act(scale * cat(x, x2) + shift)) + bias + residual) + out_shift)
 *
 * 3x3 / padding 1 or 1x1, stride 1, channels-last fp32 tensors (x [B,H,W,C1], x2 [B,H,W,C2] or NULL, residual / out /
 * twins [B,H,W,Cout]); zero padding is applied AFTER affine + activation.  `upsample2x`: x (and x2) are the
 * half-resolution tensors [B,H/2,W/2,C] read as their nearest x2 upsampling (F.interpolate fused, sige_fused_unet.py:222-227).
 * scale / shift: [affineB, C1+C2] (affineB 1 or B) or NULL / NULL (then activation must be identity).
 * Arithmetic (`prec`), fp32 accumulation in every form:
 *   0: v_mfma_f32_32x32x16_f16, operands rounded to fp16 (BASELINE.json configs[4]; tolerance 2e-2 + 1e-2 |ref|);
 *   1: the same instruction, every operand split into fp16 hi + lo, products hi*hi + lo*hi + hi*lo (22-bit operands):
 *      fp32-level results, inside the fp32 path's 1e-3.  `wshift`: the weights were packed as w * 2^wshift (a power of two
 *      chosen by the caller so that max |w| * 2^wshift is in [2^13, 2^14): keeps the lo parts normal fp16 numbers);
 *   2: v_mfma_f32_32x32x2_f32, exact fp32 products (wshift must be 0).
 * Shapes: C1, C2 multiples of 64 (3x3) / 128 (1x1), Cout a multiple of 64 (sige_hip_wide_conv_supported).
 * twinK (optional): twinK = SiLU(twin_scaleK * v + twin_shiftK), v = the value `out` receives before its out-affine.
 * workspace (optional, sige_hip_wide_conv_workspace floats): small layers split K across workgroups and finish inside
 * the launch (deterministic summation order); without it they run unsplit.
 * stats (optional, [B * ceil(H/8) * ceil(W/8), Cout, 2] floats): per 8x8 pixel block and output channel the (sum, sum of
 * squares) of what `out` receives (before its out-affine) -- the input of sige_hip_group_norm_affine_from_stats_f32, so that
 * the GroupNorm of the output needs no pass over it.                                                              */
int sige_hip_wide_conv_supported(int C1, int C2, int Cout, int kH, int kW);
/* packed size in 4-byte units (0: unsupported shape) */
size_t sige_hip_wide_conv_packed_size(int Cout, int Cin, int kH, int kW, int prec);
int sige_hip_wide_conv_pack(const float *w, int Cout, int Cin, int kH, int kW, int prec, int wshift,
                            float *packed, void *stream);
size_t sige_hip_wide_conv_workspace(int B, int H, int W, int C1, int C2, int Cout, int kH, int kW);
int sige_hip_wide_conv_nhwc(const float *x, const float *x2, int B, int C1, int C2, int H, int W, int upsample2x,
                            const float *scale, const float *shift, int affineB, int activation,
                            const float *packed, int prec, int wshift, const float *bias, int Cout, int kH, int kW,
                            const float *residual, const float *out_scale, const float *out_shift, int out_activation,
                            float *twin0, const float *twin_scale0, const float *twin_shift0,
                            float *twin1, const float *twin_scale1, const float *twin_shift1,
                            float *workspace, size_t workspace_floats, float *out, float *stats, void *stream);

/* out = act(scale[b,c] * x + shift[b,c]) over a channels-last tensor [B,H,W,C] (scale / shift [affineB, C], affineB 1 or B):
 * the activated copy of a ScatterGather cache that the full pass keeps next to the cache (one pass instead of the
 * multiply / add / SiLU / copy kernels of the torch expression; sige_amd.nn.ScatterGather.cache_activated).          */
int sige_hip_affine_act_nhwc_f32(const float *x, int B, int C, int H, int W, const float *scale, const float *shift,
                                 int affineB, int activation, float *out, void *stream);

/* ---- multi-head attention over token matrices (Stable Diffusion's spatial transformer) ---------------------------------
 * out[b, i, h*d .. (h+1)*d) = softmax_j(scale * q[b,i,h] . k[b,j,h]) v[b,j,h]  for q [B,Nq,C], k / v [B,Nk,C], C = heads * d,
 * row-major fp32 -- a channels-last [B,C,H,W] tensor IS its token matrix [B,HW,C] and channels-last tiles [T,C,4,4] ARE
 * [T*16, C], so the sparse-query attention of stable-diffusion/ldm/modules/sige_attention.py:151-176 (queries = the tokens of
 * the active tiles, keys / values = every token of the scattered feature map, or the text context) needs no rearrange copy
 * and no score tensor in HBM: one launch, exact fp32 products, online softmax (attention.py's CrossAttention.forward does
 * rearrange x 3, einsum, softmax, einsum, rearrange).  Nq % 16 == 0, d % 4 == 0, d <= 160; Nk arbitrary.                */
int sige_hip_attention_tokens_supported(int Nq, int Nk, int C, int heads);
int sige_hip_attention_tokens_f32(const float *q, const float *k, const float *v, int B, int Nq, int Nk, int C,
                                  int heads, float scale, float *out, void *stream);

/* ---- fp16-STORED caches: the "_f16" forms of SURVEY.md 8b's export list (8f row 4: fp16 cache) ---------------------
 * The reference is fp32-only (sige/nn/base.py:15,55-63).  Here the CACHED tensors of a SIGE model -- Scatter /
 * ScatterGather `original_outputs`, ScatterWithBlockResidual `original_outputs` / `original_residuals`, and the activated
 * copy of a ScatterGather cache -- may be stored as fp16 (SIGEModel.set_cache_dtype("f16"): half the resident bytes,
 * half the bytes of the cache broadcast, no conversion pass on either side of the wire).  Activations, tiles, affines and
 * outputs stay fp32; a cached value is widened exactly when it is read.  Channels-last only.  Arguments as in the
 * "_f32" entry points of the same name, with the cached tensor(s) `const void *` = halves:
 *   gather_nhwc_f16                       x is the fp16 tensor (tiles of a cache)
 *   scatter_gather_nhwc_f16               y
 *   scatter_nhwc_f16                      y          (residual fp32)
 *   scatter_with_block_residual_nhwc_f16  y0 and y1
 *   affine_act_nhwc_f16                   x; out fp16 (out_f16 != 0: the activated copy) or fp32 (a persistent twin)
 *   convert_f16_f32 / convert_f32_f16     n elements, n % 4 == 0 (persistent-output refresh / storing a full-pass output)
 *   scatter_gather_conv_nhwc_c16, scatter_gather_conv_scatter_nhwc_c16
 *                                         the fused scatter_gather -> conv (-> scatter) launches with `y` fp16 and, in the
 *                                         second, `residual` fp16 when residual_f16 != 0 (a fused ScatterWithBlockResidual's
 *                                         cached shortcut); `compute`: 0 exact fp32 | 1 fp16 operands | 2 split fp16
 *                                         operands = the packing of `packed`.  3x3 / stride 1 only.                   */
int sige_hip_gather_nhwc_f16(const void *x, int B, int C, int H, int W, int bH, int bW,
                             const int32_t *active_indices, int N,
                             const float *scale, int scaleB, int scaleC,
                             const float *shift, int shiftB, int shiftC,
                             int activation, float *out, void *stream);
int sige_hip_scatter_gather_nhwc_f16(const float *x, const void *y, int B, int C, int H, int W,
                                     int Rx, int Sx, int bH, int bW,
                                     const int32_t *active_indices, int N, const int32_t *scatter_map,
                                     const float *scale, int scaleB, int scaleC,
                                     const float *shift, int shiftB, int shiftC,
                                     int activation, float *out, void *stream);
int sige_hip_scatter_nhwc_f16(const float *x, const void *y, int B, int C, int H, int W, int R, int S,
                              int offsetH, int offsetW, int strideH, int strideW,
                              const int32_t *active_indices, const int32_t *table, int gH, int gW, int N,
                              const float *residual, int in_place, float *out, void *stream);
int sige_hip_scatter_with_block_residual_nhwc_f16(
        const float *x0, const void *y0, const float *x1, const void *y1,
        int B, int C, int H, int W, int R0, int S0, int R1, int S1,
        int offsetH, int offsetW, int strideH, int strideW,
        const int32_t *active_indices0, const int32_t *table0, int gH0, int gW0, int N0,
        const int32_t *active_indices1, const int32_t *table1, int gH1, int gW1, int N1,
        int in_place, float *out, void *stream);
int sige_hip_affine_act_nhwc_f16(const void *x, int B, int C, int H, int W, const float *scale, const float *shift,
                                 int affineB, int activation, void *out, int out_f16, void *stream);
int sige_hip_convert_f16_f32(const void *src, float *dst, size_t n, void *stream);
int sige_hip_convert_f32_f16(const float *src, void *dst, size_t n, void *stream);
int sige_hip_scatter_gather_conv_nhwc_c16(int compute, const float *x, const void *y, int B, int Cin, int H, int W,
                                          int Rx, int Sx, int bH, int bW,
                                          const int32_t *active_indices, int N, const int32_t *scatter_map,
                                          const float *scale, int scaleB, int scaleC,
                                          const float *shift, int shiftB, int shiftC,
                                          int activation,
                                          const float *packed, const float *bias, int Cout, int kH, int kW,
                                          int strideH, int strideW, float *out, void *stream);
int sige_hip_scatter_gather_conv_scatter_nhwc_c16(
        int compute, const float *x, const void *y, int B, int Cin, int H, int W, int Rx, int Sx, int bH, int bW,
        const int32_t *active_indices, int N, const int32_t *scatter_map,
        const float *scale, int scaleB, int scaleC, const float *shift, int shiftB, int shiftC, int activation,
        const float *packed, const float *bias, int Cout, int kH, int kW,
        int offsetH, int offsetW, const void *residual, int residual_f16,
        const float *x1, const int32_t *table1, int gH1, int gW1, int N1, int R1, int S1,
        float *twin0, const float *twin0_scale, const float *twin0_shift,
        float *twin1, const float *twin1_scale, const float *twin1_shift,
        float *out, void *stream);

/* ---- stacked edits ("throughput mode"; not in the reference, whose batch shares ONE mask: sige/cpu/gather.cpp:17-21) ----
 * E edited versions of one original image, each with ITS OWN mask, processed by one set of launches: every activation
 * [E,C,H,W] (channels-last) is handed to this library as the tall image [1,C,E*H,W] -- the same bytes --, and masks, index
 * lists, scatter maps, tile tables and the cached tensors are those of the tall image (the cache of the original repeated E
 * times).  A launch then sees the active tiles of all E edits at once -- at a 1 % edit the sum of eight edits' tiles is what a
 * 10 % edit has, and a launch leaves the launch-bound regime.  sige_hip_set_edit_batch(E) tells the library where the seams are:
 * a halo row beyond a tile's own image is zero padding (not the neighbour image's pixels) in the channels-last fused gather /
 * scatter_gather -> conv kernels, the dense-layer conv, the standalone channels-last gather / scatter_gather, the SPADE
 * modulation and sige_hip_scatter_gather_split_nhwc_f32; entry points whose kernels have no seam test (the NCHW forms) return
 * SIGE_HIP_EUNSUPPORTED while E > 1; per-pixel helpers (nearest resize by an integer factor, act_split, the dense SPADE
 * modulation) need none.  One image's height must be a power of two at every resolution.  Per host thread; 1 = off (default).  Whole-image ops (conv_in / conv_out, attention, GroupNorm) are simply
 * called with B = E on the same memory (sige_amd/stacked.py).  The mask pipeline follows: sige_hip_reduce_mask_i32 lets a
 * candidate tile see only its own image's mask rows, sige_hip_dilate_mask_u8 does not dilate across a seam, and
 * sige_hip_mask_pyramid_u8 builds the pyramid of every image with that image's own maxima and thresholds (one workgroup per
 * image; min_h applies to one image) -- so the index lists of a stacked mask are exactly the per-edit lists.            */
int sige_hip_set_edit_batch(int E);
int sige_hip_get_edit_batch(void);

/* ---- launch plans: a sparse forward that survives a mask change -------------------------------------------------
 * The reference sizes every launch from `activeIndices.size(0)` at call time (sige/cuda/gather_kernel.cu:78-84,111 via
 * sige/utils.py:30 and sige/nn/gather.py:101-107), and two of its three applications run ONE sparse forward per mask
 * (gaugan/runner.py:150-195, diffusion_demo/runner.py:134-164).  A plan records the calls of this library made between
 * sige_hip_plan_begin and _end on the calling thread (each call is executed as usual AND stored with its arguments) into
 * one of two sections -- 0: the mask -> index pipeline (dilation / pyramid / compaction / tile tables / scatter maps / the
 * refresh of the persistent outputs), 1: the sparse forward -- and sige_hip_plan_run replays a section on a stream without
 * any host work per launch.  Arguments that are active-tile counts are not replayed as recorded: the host binds the
 * pointer of every index list / tile table to a SLOT (sige_hip_plan_bind_ptr); at replay a count argument takes the
 * current value of the slot its index-list argument is bound to.  Slots are set by the host (_set_slot) or, inside
 * section 0, by a recorded read-back of the compaction kernels' device-side counts (_record_readback: one
 * hipMemcpyAsync + stream synchronisation, the same single synchronisation torch.nonzero costs the reference).  Output
 * block, grid, K split and tickets are chosen inside each entry point, i.e. again at every replay.  The host keeps every
 * buffer a recorded call points at alive and sized for the largest count (sige_amd/plan.py).
 * Entry points of the NCHW / two-kernel forms receive tile counts without an index list to look them up under; a plan
 * that recorded one is "shape bound" (sige_hip_plan_shape_bound) and only valid under the counts it was recorded with.
 * Replay stops at the first call that fails and returns its status.  A section can be replayed under hipGraph stream
 * capture (section 1; section 0 synchronises).                                                                       */
void *sige_hip_plan_create(void);
int sige_hip_plan_destroy(void *plan);
int sige_hip_plan_begin(void *plan, int section, int append);
int sige_hip_plan_end(void *plan);
int sige_hip_plan_recording(void);
int sige_hip_plan_shape_bound(void *plan);
int sige_hip_plan_calls(void *plan, int section);
/* n new slots (initial count 0); returns the index of the first, -1 on error */
int sige_hip_plan_new_slots(void *plan, int n);
int sige_hip_plan_bind_ptr(void *plan, const void *ptr, int slot);
int sige_hip_plan_set_slot(void *plan, int slot, int count);
/* `ptr` is an index list whose tile count does not depend on the mask (the all-tiles list of a dense layer): a count
 * recorded next to it keeps its value.  A count recorded next to a pointer that is neither bound nor constant cannot follow
 * a new mask: it is counted (plan_unbound) and marks the plan shape bound. */
int sige_hip_plan_bind_const(void *plan, const void *ptr);
int sige_hip_plan_unbound(void *plan);
/* Drop the calls of `section` recorded after the first `calls` (the host side removes an entry-point call that returned an
 * error status while recording: the hook stores a call before the entry point validates it). */
int sige_hip_plan_truncate(void *plan, int section, int calls);
/* copies min(n, slots) counts to `out`; returns the number of slots */
int sige_hip_plan_get_slots(void *plan, int32_t *out, int n);
int sige_hip_plan_record_readback(void *plan, const int32_t *device_counts, int first_slot, int n);
int sige_hip_plan_run(void *plan, int section, void *stream);

/* ---- plain device copy used by the cache broadcast path (packs the cached
 * activations of Scatter / ScatterGather modules into one buffer) ---------- */
int sige_hip_copy_f32(const float *src, float *dst, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGE_HIP_H */
