// Stacked-block convolution for gfx950: the conv that SIGE runs on the gathered
// tiles, x [T,Cin,R,S] (*) w [Cout,Cin,k,k] -> out [T,Cout,Ro,So], padding 0.
//
// The reference hands this to F.conv2d (sige/nn/base.py:88-89; cuDNN/MIOpen see
// a batch of T tiny 6x6 images).  Here it is an LDS-tiled implicit GEMM on the
// fp32-input matrix cores:  M = T*Ro*So output pixels, N = Cout, K = Cin*k*k,
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//
//   workgroup = 256 lanes = 4 waves, one 32(M) x 32(N) output tile, full K.
//   The four waves split K (each takes a quarter of every channel chunk) and
//   reduce through LDS at the end -- with only a few hundred output tiles per
//   conv at 1-15 % edit ratio this is what keeps all 256 CUs busy.
//   A (im2col of the input tiles) is never materialised: whole input tiles of
//   a channel chunk are staged in LDS ([tile][channel][R][S], straight 16-byte
//   copies of the contiguous HBM slab, register-prefetched one chunk ahead),
//   and each lane reads its A element with ds_read_b32 at a compile-time
//   offset  q*2*R*S + ky*S + kx  from a per-lane base.
//   B (weights) is pre-packed once per weight tensor into the exact order the
//   lanes consume it, so every B load is a fully coalesced 16-byte-per-lane
//   read that stays L2-resident across workgroups.
//
//   K order inside a wave's slice: lane half h = lane>>5 takes channels of
//   parity h; u = q*k*k + tap enumerates (channel pair q, tap); MFMA #u
//   multiplies A[pixel][ch 2q+h, tap] by B[ch 2q+h, tap][co].
#include "common.hpp"

namespace sige {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KH, int STR, int R_>
struct ConvGeo {
    static constexpr int K = KH, S = STR, R = R_;
    static constexpr int KK = KH * KH;
    static constexpr int RS = R_ * R_;
    static constexpr int RO = (R_ - KH) / STR + 1;
    static constexpr int PX = RO * RO;            // output pixels per tile: 16 or 4
    static constexpr int TPB = 32 / PX;           // tiles per 32-row M block
    static constexpr int CW = (KK == 1) ? 32 : 8; // channels per wave per chunk
    static constexpr int CC = 4 * CW;             // channels per LDS chunk
    static constexpr int L = (CW / 2) * KK;       // k values per lane half per wave-chunk
    static constexpr int F = L / 4;               // 16-byte weight loads per lane per chunk
    static constexpr int PAD = (PX == 4) ? 4 : 16;
    static constexpr int TSTRIDE = CC * RS + PAD; // floats between staged tiles
    static constexpr int BUF = TPB * TSTRIDE;     // floats per LDS stage
    static_assert(L % 4 == 0, "wave slice must be a whole number of float4 weight loads");
    static_assert(32 % PX == 0, "tile pixels must divide the 32-row MFMA block");
};

constexpr int kRedStride = 36;                   // padded column stride of the K-split reduction buffer
constexpr int kRedFloats = 4 * 32 * kRedStride;  // 4 waves x 32 cols x 36

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- weight packing -------------------------------------------------------
// packed[ng][chunk][wave][f][h][j][e] = w[co = 32*ng + j][ci][tap]   (0 beyond Cin/Cout)
//   u = 4f + e,  q = u / KK,  tap = u % KK,  ci = chunk*CC + wave*CW + 2q + h
template <int KK>
__global__ void pack_weights_kernel(const float *__restrict__ w, int Cout, int Cin, float *__restrict__ packed,
                                    long total) {
    constexpr int CW = (KK == 1) ? 32 : 8, CC = 4 * CW, L = (CW / 2) * KK, F = L / 4;
    const int nchunks = (Cin + CC - 1) / CC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e = r % 4; r /= 4;
        const int j = r % 32; r /= 32;
        const int h = r % 2; r /= 2;
        const int f = r % F; r /= F;
        const int wave = r % 4; r /= 4;
        const int chunk = r % nchunks; r /= nchunks;
        const int ng = (int)r;
        const int u = 4 * f + e, q = u / KK, tap = u % KK;
        const int ci = chunk * CC + wave * CW + 2 * q + h;
        const int co = 32 * ng + j;
        packed[i] = (ci < Cin && co < Cout) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f;
    }
}

// ---- the MFMA kernel ---------------------------------------------------------
template <typename G, int VEC>
__global__ __launch_bounds__(256) void block_conv_mfma_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ packed,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ out,
                                                              int T, int Cin, int Cout, int nchunks) {
    constexpr int LDS_FLOATS = cmax(2 * G::BUF, kRedFloats);
    __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5, j = lane & 31;
    const int mb = blockIdx.x, ng = blockIdx.y;

    // A: this lane's output pixel = row j of the 32-row block
    const int tl = j / G::PX, p = j % G::PX;
    const int oy = p / G::RO, ox = p % G::RO;
    const int a_base = tl * G::TSTRIDE + (wave * G::CW + h) * G::RS + oy * G::S * G::R + ox * G::S;

    // staging: the block's TPB tiles x CC channels x RS floats, in units of VEC floats
    constexpr int UNITS_PER_TILE = G::CC * G::RS / VEC;
    constexpr int UNITS = G::TPB * UNITS_PER_TILE;
    constexpr int NLD = (UNITS + 255) / 256;
    float stage[NLD][VEC];

    auto stage_load = [&](int chunk) {
        const int c0 = chunk * G::CC;
        const int valid = min(G::CC, Cin - c0) * G::RS;  // floats of real data per tile
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * 256;
            const int t_l = v / UNITS_PER_TILE;
            const int e = (v - t_l * UNITS_PER_TILE) * VEC;
            const int t = mb * G::TPB + t_l;
            const bool ok = (v < UNITS) && (t < T) && (e < valid);
            const float *src = x + ((size_t)t * Cin + c0) * G::RS + e;
            if (VEC == 4) {
                float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) q4 = *reinterpret_cast<const float4 *>(src);
                stage[i][0] = q4.x; stage[i][1] = q4.y; stage[i][2] = q4.z; stage[i][3] = q4.w;
            } else {
                stage[i][0] = ok ? *src : 0.0f;
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
                if (VEC == 4) *reinterpret_cast<float4 *>(d) = make_float4(stage[i][0], stage[i][1], stage[i][2], stage[i][3]);
                else *d = stage[i][0];
            }
        }
    };

    // B: F float4 per lane per chunk, contiguous per (ng, chunk, wave)
    const float4 *wp = reinterpret_cast<const float4 *>(packed) +
                       ((size_t)ng * nchunks * 4 + wave) * G::F * 64 + lane;
    float4 bcur[G::F], bnext[G::F];
    auto b_load = [&](float4 (&dst)[G::F], int chunk) {
#pragma unroll
        for (int f = 0; f < G::F; ++f) dst[f] = wp[((size_t)chunk * 4 * G::F + f) * 64];
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;

    stage_load(0);
    b_load(bcur, 0);
    stage_store(0);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        if (more) {
            stage_load(chunk + 1);
            b_load(bnext, chunk + 1);
        }
        const float *a = smem + buf * G::BUF + a_base;
#pragma unroll
        for (int f = 0; f < G::F; ++f) {
            const float bv[4] = {bcur[f].x, bcur[f].y, bcur[f].z, bcur[f].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int u = 4 * f + e;
                const int q = u / G::KK, tap = u % G::KK;
                const int off = q * 2 * G::RS + (tap / G::K) * G::R + (tap % G::K);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[off], bv[e], acc, 0, 0, 0);
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
    // acc reg r of lane (h, j): row = (r&3) + 8*(r>>2) + 4*h, col = j
    float *red = smem;  // safe: the loop ended with a barrier
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(red + (wave * 32 + j) * kRedStride + 8 * g + 4 * h) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    __syncthreads();

    int co_l, t_l, row0;
    if (G::PX == 16) { row0 = (tid & 3) * 4; co_l = (tid >> 2) & 31; t_l = tid >> 7; }
    else             { row0 = 0;             co_l = tid & 31;        t_l = tid >> 5; }
    const int rrow = t_l * G::PX + row0;
    float4 s = *reinterpret_cast<const float4 *>(red + co_l * kRedStride + rrow);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 v = *reinterpret_cast<const float4 *>(red + (w * 32 + co_l) * kRedStride + rrow);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int t = mb * G::TPB + t_l, co = ng * 32 + co_l;
    if (t < T && co < Cout) {
        const float bb = bias ? bias[co] : 0.0f;
        s.x += bb; s.y += bb; s.z += bb; s.w += bb;
        *reinterpret_cast<float4 *>(out + ((size_t)t * Cout + co) * G::PX + row0) = s;
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

template <typename G>
static void launch_mfma(const float *x, int T, int Cin, const float *packed, const float *bias, int Cout,
                        float *out, hipStream_t st) {
    const int nchunks = ceil_div(Cin, G::CC);
    dim3 grid(ceil_div(T, G::TPB), ceil_div(Cout, 32));
    const bool vec = ((long)Cin * G::RS) % 4 == 0 && (G::RS % 4 == 0 || Cin % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) block_conv_mfma_kernel<G, 4><<<grid, 256, 0, st>>>(x, packed, bias, out, T, Cin, Cout, nchunks);
    else block_conv_mfma_kernel<G, 1><<<grid, 256, 0, st>>>(x, packed, bias, out, T, Cin, Cout, nchunks);
}

}  // namespace sige

using namespace sige;

extern "C" size_t sige_hip_block_conv_packed_size(int Cout, int Cin, int kH, int kW, int R, int S,
                                                  int strideH, int strideW, int groups) {
    if (Cout <= 0 || Cin <= 0) return 0;
    if (!mfma_kind(kH, kW, R, S, strideH, strideW, groups)) return 0;
    const int KK = kH * kW;
    const int CW = (KK == 1) ? 32 : 8, CC = 4 * CW, F = (CW / 2) * KK / 4;
    return (size_t)ceil_div(Cout, 32) * ceil_div(Cin, CC) * 4 * F * 2 * 32 * 4;
}

extern "C" int sige_hip_block_conv_pack_f32(const float *w, int Cout, int Cin, int kH, int kW,
                                            float *packed, void *stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return SIGE_HIP_EINVAL;
    if (kH != kW || (kH != 1 && kH != 3)) return SIGE_HIP_EUNSUPPORTED;
    const int KK = kH * kW;
    const int CW = (KK == 1) ? 32 : 8, CC = 4 * CW, F = (CW / 2) * KK / 4;
    const long total = (long)ceil_div(Cout, 32) * ceil_div(Cin, CC) * 4 * F * 2 * 32 * 4;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (KK == 9) pack_weights_kernel<9><<<blocks, 256, 0, as_stream(stream)>>>(w, Cout, Cin, packed, total);
    else pack_weights_kernel<1><<<blocks, 256, 0, as_stream(stream)>>>(w, Cout, Cin, packed, total);
    return launch_status();
}

extern "C" int sige_hip_block_conv_f32(const float *x, int T, int Cin, int R, int S,
                                       const float *packed, const float *bias, int Cout, int kH, int kW,
                                       int strideH, int strideW, float *out, void *stream) {
    if (T < 0 || Cin <= 0 || Cout <= 0) return SIGE_HIP_EINVAL;
    if (T == 0) return SIGE_HIP_OK;
    if (!x || !packed || !out) return SIGE_HIP_EINVAL;
    if (reinterpret_cast<uintptr_t>(out) & 15 || reinterpret_cast<uintptr_t>(packed) & 15) return SIGE_HIP_EINVAL;
    hipStream_t st = as_stream(stream);
    switch (mfma_kind(kH, kW, R, S, strideH, strideW, 1)) {
        case 1: launch_mfma<ConvGeo<3, 1, 6>>(x, T, Cin, packed, bias, Cout, out, st); break;
        case 2: launch_mfma<ConvGeo<1, 1, 4>>(x, T, Cin, packed, bias, Cout, out, st); break;
        case 3: launch_mfma<ConvGeo<3, 2, 5>>(x, T, Cin, packed, bias, Cout, out, st); break;
        default: return SIGE_HIP_EUNSUPPORTED;
    }
    return launch_status();
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
