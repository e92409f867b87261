// Stacked-block convolution for gfx950: the conv that SIGE runs on the gathered
// tiles, x [T,Cin,R,S] (*) w [Cout,Cin,k,k] -> out [T,Cout,Ro,So], padding 0 --
// optionally with the Gather / ScatterGather that produces those tiles fused
// into its prologue, so the [T,Cin,R,S] tensor never exists in HBM.
//
// The reference hands the conv to F.conv2d (sige/nn/base.py:88-89; cuDNN/MIOpen
// see a batch of T tiny 6x6 images) after a separate gather kernel
// (sige/cuda/gather_kernel.cu:7-67) or scatter_gather kernel
// (scatter_gather_kernel.cu:8-67).  Here it is ONE LDS-tiled implicit GEMM on
// the fp32-input matrix cores:  M = T*Ro*So output pixels, N = Cout,
// K = Cin*k*k, exact fp32 products, fp32 accumulate.
//
//   workgroup = 256 lanes = 4 waves, one MT x MT output tile, full K.
//     MT = 32: v_mfma_f32_32x32x2_f32   (large grids: less operand traffic)
//     MT = 16: v_mfma_f32_16x16x4_f32   (small grids: 4x the workgroups, so a
//              conv with a few dozen active tiles still covers the 256 CUs)
//   The four waves split K (each takes a quarter of every channel chunk) and
//   reduce through LDS at the end.
//   A (im2col of the input tiles) is never materialised: whole input tiles of
//   a channel chunk are staged in LDS as [tile][channel][R][S], register-
//   prefetched one chunk ahead, and each lane reads its A element with
//   ds_read_b32 at a compile-time offset from a per-lane base.  The staging
//   source is
//     SRC_TILES           the contiguous tile slab (16-byte copies),
//     SRC_GATHER          the full activation [B,C,H,W] through a per-workgroup
//                         pixel table (halo, zero fill outside the image) with
//                         the cached-GroupNorm affine + SiLU applied on the way,
//     SRC_SCATTER_GATHER  conv-1's output tiles / the cached tensor through the
//                         scatter map, same affine + SiLU.
//   B (weights) is pre-packed once per weight tensor into the exact order the
//   lanes consume it: every B load is a fully coalesced 16-byte-per-lane read.
//
//   K order inside a wave's slice of a chunk: lane group kq = lane / MT takes the
//   channels congruent to kq; u = q*k*k + tap enumerates (channel group q, tap);
//   MFMA #u multiplies A[pixel][ch NL*q+kq, tap] by B[ch NL*q+kq, tap][co].
#include "common.hpp"

namespace sige {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SRC_TILES = 0, SRC_GATHER = 1, SRC_SCATTER_GATHER = 2 };
enum { DST_TILES = 0, DST_NCHW = 1 };

template <int MT_> struct Mfma;
template <> struct Mfma<32> {
    using acc_t = f32x16;
    static constexpr int REGS = 16;
    __device__ static __forceinline__ acc_t op(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma<16> {
    using acc_t = f32x4;
    static constexpr int REGS = 4;
    __device__ static __forceinline__ acc_t op(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

template <int KH, int STR, int R_, int MT_>
struct ConvGeo {
    static constexpr int K = KH, S = STR, R = R_, MT = MT_;
    static constexpr int KK = KH * KH;
    static constexpr int RS = R_ * R_;
    static constexpr int RO = (R_ - KH) / STR + 1;
    static constexpr int PX = RO * RO;                 // output pixels per tile: 16 or 4
    static constexpr int NL = 64 / MT_;                // k values per MFMA = lane groups (2 or 4)
    static constexpr int TPB = MT_ / PX;               // tiles per M block
    static constexpr int CW = (KK == 1 ? 16 : 4) * NL; // channels per wave per chunk
    static constexpr int CC = 4 * CW;                  // channels per LDS chunk
    static constexpr int L = (CW / NL) * KK;           // MFMAs per wave per chunk (36 or 16)
    static constexpr int F = L / 4;                    // 16-byte weight loads per lane per chunk
    static constexpr int PAD = (PX == 4) ? 4 : 16;
    static constexpr int TSTRIDE = CC * RS + PAD;      // floats between staged tiles
    static constexpr int BUF = TPB * TSTRIDE;          // floats per LDS stage
    static constexpr int RED = MT_ + 4;                // padded column stride of the reduction buffer
    static_assert(L % 4 == 0, "wave slice must be a whole number of float4 weight loads");
    static_assert(MT_ % PX == 0 && TPB >= 1, "tile pixels must divide the M block");
};

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

struct ConvArgs {
    const float *x;       // TILES: [T,Cin,R,S] | GATHER: [B,Cin,H,W] | SCATTER_GATHER: conv-1 tiles [B*N,Cin,Rx,Sx]
    const float *y;       // SCATTER_GATHER: cached full tensor [B,Cin,H,W]
    const int32_t *idx;   // gather modes: [N,2]
    const int32_t *map;   // SCATTER_GATHER: [H,W,3]
    const float *packed;
    const float *bias;
    float *out;
    int T, Cin, Cout, nchunks;
    int B, N, H, W;
    int RxSx, Sx;
    const float *scale, *shift;  // per-(batch, channel) affine of the gather modes
    int scale_sb, scale_sc, shift_sb, shift_sc;
    // GATHER: channels [0,Csplit) come from x, [Csplit,Cin) from x2 (a fused torch.cat)
    const float *x2;
    int Csplit;
    // DST_NCHW: write the output tiles straight into a full tensor [B,Cout,Ho,Wo] at
    // ((off+idx)/stride), clipped, + residual[B,Cout,Ho,Wo] (dense layers: all tiles active)
    const float *residual;
    int Ho, Wo, offH, offW, strH, strW;
};

// ---- weight packing -------------------------------------------------------
// packed[ng][chunk][wave][f][kq][j][e] = w[co = MT*ng + j][ci][tap]   (0 beyond Cin/Cout)
//   u = 4f + e,  q = u / KK,  tap = u % KK,  ci = chunk*CC + wave*CW + NL*q + kq
template <int KK, int MT>
__global__ void pack_weights_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                    long total) {
    constexpr int NL = 64 / MT, CW = (KK == 1 ? 16 : 4) * NL, CC = 4 * CW, L = (CW / NL) * KK, F = L / 4;
    const int nchunks = (Cin + CC - 1) / CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 4; r /= 4;
        const int j = r % MT; r /= MT;
        const int kq = r % NL; r /= NL;
        const int f = r % F; r /= F;
        const int wave = r % 4; r /= 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int ng = (int)r;
        const int u = 4 * f + e, q = u / KK, tap = u % KK;
        const int ci = chunk * CC + wave * CW + NL * q + kq;
        const int co = MT * ng + j;
        packed[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f;
    }
}

static size_t packed_floats(int Cout, int Cin, int KK, int MT) {
    const int NL = 64 / MT, CW = (KK == 1 ? 16 : 4) * NL, CC = 4 * CW, F = (CW / NL) * KK / 4;
    return (size_t)ceil_div(Cout, MT) * ceil_div(Cin, CC) * 4 * F * 64 * 4;
}

// ---- the MFMA kernel ---------------------------------------------------------
template <typename G, int SRC, int ACT, int VEC, int DST>
__global__ __launch_bounds__(256) void block_conv_mfma_kernel(ConvArgs a) {
    using M = Mfma<G::MT>;
    constexpr int LDS_FLOATS = cmax(2 * G::BUF, 4 * G::MT * G::RED);
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
    // pixel tables of the gather modes (one entry per staged pixel)
    __shared__ int s_src[SRC == SRC_TILES ? 1 : G::TPB * G::RS];
    __shared__ int s_hw[SRC == SRC_SCATTER_GATHER ? G::TPB * G::RS : 1];
    __shared__ int s_b[SRC == SRC_TILES ? 1 : G::TPB];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int kq = lane / G::MT, j = lane % G::MT;
    const int mb = blockIdx.x, ng = blockIdx.y;
    const int Cin = a.Cin;

    if (SRC != SRC_TILES) {
        // src: -2 = outside the image / past the last tile (value 0, no affine, no activation)
        //      -1 = SCATTER_GATHER: take the cached tensor at s_hw
        //     >=0 = GATHER: h*W+w ; SCATTER_GATHER: blk*Cin*RxSx + r*Sx + s in conv-1's output tiles
        for (int p = tid; p < G::TPB * G::RS; p += 256) {
            const int t_l = p / G::RS, rs = p - t_l * G::RS;
            const int t = mb * G::TPB + t_l;
            int src = -2, hw = 0;
            if (t < a.T) {
                const int n = t % a.N;
                const int h = a.idx[2 * n] + rs / G::R, w = a.idx[2 * n + 1] + rs % G::R;
                if (h >= 0 && h < a.H && w >= 0 && w < a.W) {
                    hw = h * a.W + w;
                    if (SRC == SRC_GATHER) {
                        src = hw;
                    } else {
                        const int32_t *m = a.map + 3 * (size_t)hw;
                        const int blk = m[0];
                        src = blk >= 0 ? blk * Cin * a.RxSx + m[1] * a.Sx + m[2] : -1;
                    }
                }
            }
            s_src[p] = src;
            if (SRC == SRC_SCATTER_GATHER) s_hw[p] = hw;
        }
        if (tid < G::TPB) {
            const int t = mb * G::TPB + tid;
            s_b[tid] = t < a.T ? t / a.N : 0;
        }
        __syncthreads();
    }

    // A: this lane's output pixel = row j of the M block
    const int tl = j / G::PX, px = j % G::PX;
    const int oy = px / G::RO, ox = px % G::RO;
    const int a_base = tl * G::TSTRIDE + (wave * G::CW + kq) * G::RS + oy * G::S * G::R + ox * G::S;

    // staging: the block's TPB tiles x CC channels x RS floats, in units of VEC floats
    constexpr int UNITS_PER_TILE = G::CC * G::RS / VEC;
    constexpr int UNITS = G::TPB * UNITS_PER_TILE;
    constexpr int NLD = (UNITS + 255) / 256;
    float stage[NLD][VEC];
    float st_scale[SRC == SRC_TILES ? 1 : NLD], st_shift[SRC == SRC_TILES ? 1 : NLD];
    unsigned st_ok = 0;

    auto stage_load = [&](int chunk) {
        const int c0 = chunk * G::CC;
        if (SRC == SRC_TILES) {
            const int valid = min(G::CC, Cin - c0) * G::RS;  // floats of real data per tile
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int v = tid + i * 256;
                const int t_l = v / UNITS_PER_TILE;
                const int e = (v - t_l * UNITS_PER_TILE) * VEC;
                const int t = mb * G::TPB + t_l;
                const bool ok = (v < UNITS) && (t < a.T) && (e < valid);
                const float *src = a.x + ((size_t)t * Cin + c0) * G::RS + e;
                if (VEC == 4) {
                    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) q4 = *reinterpret_cast<const float4 *>(src);
                    stage[i][0] = q4.x; stage[i][1] = q4.y; stage[i][2] = q4.z; stage[i][3] = q4.w;
                } else {
                    stage[i][0] = ok ? *src : 0.0f;
                }
            }
        } else {
            const size_t HW = (size_t)a.H * a.W;
            st_ok = 0;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int v = tid + i * 256;
                const int t_l = v / UNITS_PER_TILE;
                const int rem = v - t_l * UNITS_PER_TILE;
                const int c_l = rem / G::RS, p = rem - c_l * G::RS;
                const int c = c0 + c_l;
                float z = 0.0f, sc = 1.0f, sh = 0.0f;
                if (v < UNITS && c < Cin) {
                    const int src = s_src[t_l * G::RS + p];
                    const int b = s_b[t_l];
                    if (src != -2) {
                        st_ok |= 1u << i;
                        if (SRC == SRC_GATHER) {
                            z = (c < a.Csplit) ? a.x[((size_t)b * a.Csplit + c) * HW + src]
                                               : a.x2[((size_t)b * (Cin - a.Csplit) + (c - a.Csplit)) * HW + src];
                        } else if (src >= 0) {
                            z = a.x[((size_t)b * a.N * Cin + c) * a.RxSx + src];
                        } else {
                            z = a.y[((size_t)b * Cin + c) * HW + s_hw[t_l * G::RS + p]];
                        }
                        if (a.scale) sc = a.scale[(size_t)b * a.scale_sb + (size_t)c * a.scale_sc];
                        if (a.shift) sh = a.shift[(size_t)b * a.shift_sb + (size_t)c * a.shift_sc];
                    }
                }
                stage[i][0] = z; st_scale[i] = sc; st_shift[i] = sh;
            }
        }
    };
    auto stage_store = [&](int buf) {
        float *dst = smem + buf * G::BUF;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * 256;
            if (v < UNITS) {
                const int t_l = v / UNITS_PER_TILE;
                const int e = (v - t_l * UNITS_PER_TILE) * VEC;
                float *d = dst + t_l * G::TSTRIDE + e;
                if (SRC == SRC_TILES) {
                    if (VEC == 4) *reinterpret_cast<float4 *>(d) = make_float4(stage[i][0], stage[i][1], stage[i][2], stage[i][3]);
                    else *d = stage[i][0];
                } else {
                    // scale, then shift, then activation: two separately rounded ops as in the
                    // reference (gather.cpp:33-53; the file is built with -ffp-contract=off)
                    float z = stage[i][0];
                    if ((st_ok >> i) & 1u) {
                        z = st_scale[i] * z;
                        z = st_shift[i] + z;
                        z = activate<ACT>(z);
                    }
                    *d = z;
                }
            }
        }
    };

    // B: F float4 per lane per chunk, contiguous per (ng, chunk, wave)
    const float4 *wp = reinterpret_cast<const float4 *>(a.packed) +
                       ((size_t)ng * a.nchunks * 4 + wave) * G::F * 64 + lane;
    float4 bcur[G::F], bnext[G::F];
    auto b_load = [&](float4 (&dst)[G::F], int chunk) {
#pragma unroll
        for (int f = 0; f < G::F; ++f) dst[f] = wp[((size_t)chunk * 4 * G::F + f) * 64];
    };

    typename M::acc_t acc;
#pragma unroll
    for (int i = 0; i < M::REGS; ++i) acc[i] = 0.0f;

    stage_load(0);
    b_load(bcur, 0);
    stage_store(0);
    __syncthreads();

    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < a.nchunks;
        if (more) {
            stage_load(chunk + 1);
            b_load(bnext, chunk + 1);
        }
        const float *as = smem + buf * G::BUF + a_base;
#pragma unroll
        for (int f = 0; f < G::F; ++f) {
            const float bv[4] = {bcur[f].x, bcur[f].y, bcur[f].z, bcur[f].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int u = 4 * f + e;
                const int q = u / G::KK, tap = u % G::KK;
                const int off = q * G::NL * G::RS + (tap / G::K) * G::R + (tap % G::K);
                acc = M::op(as[off], bv[e], acc);
            }
        }
        if (more) {
            stage_store(buf ^ 1);
#pragma unroll
            for (int f = 0; f < G::F; ++f) bcur[f] = bnext[f];
        }
        __syncthreads();
    }

    // ---- K-split reduction across the 4 waves, bias, store -----------------
    // MT=32: reg r of lane (kq, j): row = (r&3) + 8*(r>>2) + 4*kq ; MT=16: row = 4*kq + r ; col = j
    float *red = smem;  // safe: the loop ended with a barrier
    if (G::MT == 32) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(red + (wave * 32 + j) * G::RED + 8 * g + 4 * kq) =
                make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    } else {
        *reinterpret_cast<float4 *>(red + (wave * 16 + j) * G::RED + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();

    // one float4 (4 consecutive pixels of one tile and one output channel) per lane
    constexpr int P4 = G::PX / 4;                  // float4 per (tile, channel)
    constexpr int OUT_UNITS = G::MT * G::MT / 4;   // 256 or 64
    if (tid < OUT_UNITS) {
        const int p4 = tid % P4;
        const int co_l = (tid / P4) % G::MT;
        const int t_l = tid / (P4 * G::MT);
        const int rrow = t_l * G::PX + p4 * 4;
        float4 s = *reinterpret_cast<const float4 *>(red + co_l * G::RED + rrow);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(red + (w * G::MT + co_l) * G::RED + rrow);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int t = mb * G::TPB + t_l, co = ng * G::MT + co_l;
        if (t < a.T && co < a.Cout) {
            const float bb = a.bias ? a.bias[co] : 0.0f;
            s.x += bb; s.y += bb; s.z += bb; s.w += bb;
            if (DST == DST_TILES) {
                *reinterpret_cast<float4 *>(a.out + ((size_t)t * a.Cout + co) * G::PX + p4 * 4) = s;
            } else {
                const int b = t / a.N, n = t - b * a.N;
                const int h0 = (a.offH + a.idx[2 * n]) / a.strH, w0 = (a.offW + a.idx[2 * n + 1]) / a.strW;
                const size_t plane = ((size_t)b * a.Cout + co) * a.Ho * a.Wo;
                const float sv[4] = {s.x, s.y, s.z, s.w};
                if (G::RO == 4) {
                    // one 4-pixel output row of the tile
                    const int h = h0 + p4;
                    if (h >= 0 && h < a.Ho) {
                        const size_t q = plane + (size_t)h * a.Wo + w0;
                        if (w0 >= 0 && w0 + 3 < a.Wo && ((q & 3) == 0)) {
                            float4 o = s;
                            if (a.residual) {
                                const float4 r = *reinterpret_cast<const float4 *>(a.residual + q);
                                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                            }
                            *reinterpret_cast<float4 *>(a.out + q) = o;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (w0 + i >= 0 && w0 + i < a.Wo)
                                    a.out[q + i] = sv[i] + (a.residual ? a.residual[q + i] : 0.0f);
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pp = p4 * 4 + i;
                        const int h = h0 + pp / G::RO, w = w0 + pp % G::RO;
                        if (h >= 0 && h < a.Ho && w >= 0 && w < a.Wo) {
                            const size_t q = plane + (size_t)h * a.Wo + w;
                            a.out[q] = sv[i] + (a.residual ? a.residual[q] : 0.0f);
                        }
                    }
                }
            }
        }
    }
}

// ---- any-shape direct kernel (groups, odd tiles): one lane per output -------
__global__ void block_conv_direct_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                         const float *__restrict__ bias, float *__restrict__ out,
                                         int T, int Cin, int R, int S, int Cout, int kH, int kW,
                                         int strH, int strW, int groups, int Ro, int So, long total) {
    const int cig = Cin / groups, cog = Cout / groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int ox = r % So; r /= So;
        const int oy = r % Ro; r /= Ro;
        const int co = r % Cout; r /= Cout;
        const int t = (int)r;
        const int g = co / cog;
        float acc = bias ? bias[co] : 0.0f;
        for (int ci = 0; ci < cig; ++ci) {
            const float *xp = x + (((size_t)t * Cin + g * cig + ci) * R + oy * strH) * S + ox * strW;
            const float *wq = w + ((size_t)co * cig + ci) * kH * kW;
            for (int ky = 0; ky < kH; ++ky)
                for (int kx = 0; kx < kW; ++kx) acc = fmaf(xp[ky * S + kx], wq[ky * kW + kx], acc);
        }
        out[i] = acc;
    }
}

static int mfma_kind(int kH, int kW, int R, int S, int strH, int strW, int groups) {
    if (groups != 1 || kH != kW || R != S || strH != strW) return 0;
    if (kH == 3 && strH == 1 && R == 6) return 1;
    if (kH == 1 && strH == 1 && R == 4) return 2;
    if (kH == 3 && strH == 2 && R == 5) return 3;
    return 0;
}

template <typename G, int SRC, int DST>
static void launch_geo(ConvArgs a, int act, hipStream_t st) {
    a.nchunks = ceil_div(a.Cin, G::CC);
    dim3 grid(ceil_div(a.T, G::TPB), ceil_div(a.Cout, G::MT));
    if (SRC == SRC_TILES) {
        const bool vec = ((long)a.Cin * G::RS) % 4 == 0 && (G::RS % 4 == 0 || a.Cin % 4 == 0) &&
                         (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
        if (vec) block_conv_mfma_kernel<G, SRC_TILES, 0, 4, DST_TILES><<<grid, 256, 0, st>>>(a);
        else block_conv_mfma_kernel<G, SRC_TILES, 0, 1, DST_TILES><<<grid, 256, 0, st>>>(a);
    } else if (act == SIGE_HIP_ACT_SWISH) {
        block_conv_mfma_kernel<G, SRC, SIGE_HIP_ACT_SWISH, 1, DST><<<grid, 256, 0, st>>>(a);
    } else {
        block_conv_mfma_kernel<G, SRC, SIGE_HIP_ACT_IDENTITY, 1, DST><<<grid, 256, 0, st>>>(a);
    }
}

// Tile choice: 32x32 tiles unless that leaves most of the 256 CUs idle.
template <int KH, int STR, int R, int SRC, int DST>
static void launch_kind(ConvArgs a, int act, hipStream_t st) {
    using G32 = ConvGeo<KH, STR, R, 32>;
    using G16 = ConvGeo<KH, STR, R, 16>;
    const long blocks32 = (long)ceil_div(a.T, G32::TPB) * ceil_div(a.Cout, 32);
    if (blocks32 >= 192) {
        launch_geo<G32, SRC, DST>(a, act, st);
    } else {
        a.packed += packed_floats(a.Cout, a.Cin, KH * KH, 32);  // the MT=16 layout follows the MT=32 one
        launch_geo<G16, SRC, DST>(a, act, st);
    }
}

template <int SRC, int DST = DST_TILES>
static int launch_conv(const ConvArgs &a, int act, int kH, int kW, int R, int S, int strH, int strW, hipStream_t st) {
    switch (mfma_kind(kH, kW, R, S, strH, strW, 1)) {
        case 1: launch_kind<3, 1, 6, SRC, DST>(a, act, st); break;
        case 2: launch_kind<1, 1, 4, SRC, DST>(a, act, st); break;
        case 3: launch_kind<3, 2, 5, SRC, DST>(a, act, st); break;
        default: return SIGE_HIP_EUNSUPPORTED;
    }
    return launch_status();
}

}  // namespace sige

using namespace sige;

extern "C" size_t sige_hip_block_conv_packed_size(int Cout, int Cin, int kH, int kW, int R, int S,
                                                  int strideH, int strideW, int groups) {
    if (Cout <= 0 || Cin <= 0) return 0;
    if (!mfma_kind(kH, kW, R, S, strideH, strideW, groups)) return 0;
    return packed_floats(Cout, Cin, kH * kW, 32) + packed_floats(Cout, Cin, kH * kW, 16);
}

extern "C" int sige_hip_block_conv_pack_f32(const float *w, int Cout, int Cin, int kH, int kW,
                                            float *packed, void *stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return SIGE_HIP_EINVAL;
    if (kH != kW || (kH != 1 && kH != 3)) return SIGE_HIP_EUNSUPPORTED;
    const int KK = kH * kW;
    hipStream_t st = as_stream(stream);
    const long n32 = (long)packed_floats(Cout, Cin, KK, 32), n16 = (long)packed_floats(Cout, Cin, KK, 16);
    auto blocks = [](long n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); };
    if (KK == 9) {
        pack_weights_kernel<9, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, packed, n32);
        pack_weights_kernel<9, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, packed + n32, n16);
    } else {
        pack_weights_kernel<1, 32><<<blocks(n32), 256, 0, st>>>(w, Cout, Cin, packed, n32);
        pack_weights_kernel<1, 16><<<blocks(n16), 256, 0, st>>>(w, Cout, Cin, packed + n32, n16);
    }
    return launch_status();
}

extern "C" int sige_hip_block_conv_f32(const float *x, int T, int Cin, int R, int S,
                                       const float *packed, const float *bias, int Cout, int kH, int kW,
                                       int strideH, int strideW, float *out, void *stream) {
    if (T < 0 || Cin <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.packed = packed; a.bias = bias; a.out = out;
    a.T = T; a.Cin = Cin; a.Cout = Cout;
    return launch_conv<SRC_TILES>(a, 0, kH, kW, R, S, strideH, strideW, as_stream(stream));
}

static bool channel_affine(const float *p, int b, int c, int h, int w, int B, int C) {
    return !p || ((b == 1 || b == B) && (c == 1 || c == C) && h == 1 && w == 1);
}

extern "C" int sige_hip_gather_conv_f32(const float *x, int B, int Cin, int H, int W, int bH, int bW,
                                        const int32_t *active_indices, int N,
                                        const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                        const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                        int activation,
                                        const float *packed, const float *bias, int Cout, int kH, int kW,
                                        int strideH, int strideW, float *out, void *stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (!channel_affine(scale, scaleB, scaleC, scaleH, scaleW, B, Cin) ||
        !channel_affine(shift, shiftB, shiftC, shiftH, shiftW, B, Cin))
        return SIGE_HIP_EUNSUPPORTED;  // spatially varying affine: use gather + block_conv
    if ((long)H * W >= (1L << 31)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out || !active_indices) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.idx = active_indices; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.Csplit = Cin;
    a.scale = scale; a.scale_sb = scaleB > 1 ? scaleC : 0; a.scale_sc = scaleC > 1 ? 1 : 0;
    a.shift = shift; a.shift_sb = shiftB > 1 ? shiftC : 0; a.shift_sc = shiftC > 1 ? 1 : 0;
    return launch_conv<SRC_GATHER>(a, activation, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_gather_conv_nchw_f32(const float *x, const float *x2, int B, int C1, int C2, int H, int W,
                                             int bH, int bW, const int32_t *active_indices, int N,
                                             const float *scale, int scaleB, int scaleC,
                                             const float *shift, int shiftB, int shiftC,
                                             int activation,
                                             const float *packed, const float *bias, int Cout, int kH, int kW,
                                             int strideH, int strideW, int offsetH, int offsetW,
                                             const float *residual, int Ho, int Wo, float *out, void *stream) {
    const int Cin = C1 + C2;
    if (B < 0 || C1 <= 0 || C2 < 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Ho <= 0 || Wo <= 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if ((scale && !((scaleB == 1 || scaleB == B) && (scaleC == 1 || scaleC == Cin))) ||
        (shift && !((shiftB == 1 || shiftB == B) && (shiftC == 1 || shiftC == Cin))))
        return SIGE_HIP_EINVAL;
    if ((long)H * W >= (1L << 31)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || (C2 && !x2) || !packed || !out || !active_indices) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.x2 = x2; a.Csplit = C1; a.idx = active_indices; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.scale = scale; a.scale_sb = scaleB > 1 ? scaleC : 0; a.scale_sc = scaleC > 1 ? 1 : 0;
    a.shift = shift; a.shift_sb = shiftB > 1 ? shiftC : 0; a.shift_sc = shiftC > 1 ? 1 : 0;
    a.residual = residual; a.Ho = Ho; a.Wo = Wo; a.offH = offsetH; a.offW = offsetW; a.strH = strideH; a.strW = strideW;
    return launch_conv<SRC_GATHER, DST_NCHW>(a, activation, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_scatter_gather_conv_f32(const float *x, const float *y, int B, int Cin, int H, int W,
                                                int Rx, int Sx, int bH, int bW,
                                                const int32_t *active_indices, int N, const int32_t *scatter_map,
                                                const float *scale, int scaleB, int scaleC, int scaleH, int scaleW,
                                                const float *shift, int shiftB, int shiftC, int shiftH, int shiftW,
                                                int activation,
                                                const float *packed, const float *bias, int Cout, int kH, int kW,
                                                int strideH, int strideW, float *out, void *stream) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || N < 0 || Rx <= 0 || Sx <= 0) return SIGE_HIP_EINVAL;
    if (activation != SIGE_HIP_ACT_IDENTITY && activation != SIGE_HIP_ACT_SWISH) return SIGE_HIP_EUNSUPPORTED;
    if (!channel_affine(scale, scaleB, scaleC, scaleH, scaleW, B, Cin) ||
        !channel_affine(shift, shiftB, shiftC, shiftH, shiftW, B, Cin))
        return SIGE_HIP_EUNSUPPORTED;
    if ((long)H * W >= (1L << 31) || (long)N * Cin * Rx * Sx >= (1L << 31)) return SIGE_HIP_EUNSUPPORTED;
    if ((long)B * N == 0) return SIGE_HIP_OK;
    if (!x || !y || !packed || !out || !active_indices || !scatter_map) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    ConvArgs a{};
    a.x = x; a.y = y; a.idx = active_indices; a.map = scatter_map; a.packed = packed; a.bias = bias; a.out = out;
    a.T = B * N; a.Cin = Cin; a.Cout = Cout; a.B = B; a.N = N; a.H = H; a.W = W;
    a.RxSx = Rx * Sx; a.Sx = Sx;
    a.scale = scale; a.scale_sb = scaleB > 1 ? scaleC : 0; a.scale_sc = scaleC > 1 ? 1 : 0;
    a.shift = shift; a.shift_sb = shiftB > 1 ? shiftC : 0; a.shift_sc = shiftC > 1 ? 1 : 0;
    return launch_conv<SRC_SCATTER_GATHER>(a, activation, kH, kW, bH, bW, strideH, strideW, as_stream(stream));
}

extern "C" int sige_hip_block_conv_direct_f32(const float *x, int T, int Cin, int R, int S,
                                              const float *w, const float *bias, int Cout, int kH, int kW,
                                              int strideH, int strideW, int groups, float *out, void *stream) {
    if (T < 0 || Cin <= 0 || Cout <= 0 || kH <= 0 || kW <= 0 || strideH <= 0 || strideW <= 0 || groups <= 0)
        return SIGE_HIP_EINVAL;
    if (Cin % groups || Cout % groups || R < kH || S < kW) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !w || !out) return SIGE_HIP_EINVAL;
    const int Ro = (R - kH) / strideH + 1, So = (S - kW) / strideW + 1;
    const long total = (long)T * Cout * Ro * So;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    block_conv_direct_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, w, bias, out, T, Cin, R, S, Cout, kH, kW,
                                                                   strideH, strideW, groups, Ro, So, total);
    return launch_status();
}
