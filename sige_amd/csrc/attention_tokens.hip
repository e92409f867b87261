// Multi-head attention over token matrices, one launch: out = softmax(scale * q k^T) v per (batch, head).
//
// Stable Diffusion's spatial transformer (stable-diffusion/ldm/modules/sige_attention.py:151-176 -> attention.py CrossAttention):
// in sparse mode the QUERIES are the tokens of the active 4x4 tiles (a few hundred to a few thousand), the keys / values of
// the self-attention all HW tokens of the (scattered) feature map, those of the cross-attention the 77 text tokens.  The
// reference -- and rounds 1-3 here -- runs it as rearrange('b n (h d) -> (b h) n d') x 3 (copies), bmm, softmax, bmm,
// rearrange back (copy): per attention 2 library GEMMs, a softmax over a [B*h, Nq, Nk] score tensor written to and read
// from HBM, and 8 copy kernels (profiles/r2m_kerneltrace_sd_unet_sparse_15pct.csv: 138 copy kernels, 5 cunn_SoftMaxForward
// and ~25 Cijk_* launches of this kind per forward).  Here: heads are strides, the score tile never leaves the workgroup.
//
//   workgroup = 16 queries of one (batch, head); its 4 waves take the key blocks (16 keys each) round-robin, each with its own
//   running (max, sum, O) -- flash-attention's online softmax --, and meet in LDS at the end.
//   S = Q K^T on v_mfma_f32_16x16x4_f32 (exact fp32 products): lane (kq, j) holds, per 16-channel unit, Q[j][16u + 4kq ..+3] and
//   K[key j][same 4 channels] -- one 16-byte load each, four k-steps; the head dimension d (40 / 80 / 160 for SD v1) is padded
//   to whole units with zeros.
//   P = exp(S - m) goes through a wave-private 16 x 16 LDS tile (C layout -> A layout); O += P V with V[key 4t + kq][16n + j]
//   straight from global memory (16 lanes = 64 contiguous bytes of a key's row).  Row maxima / sums over the 16 lanes of a
//   score row on the DPP path (common.hpp row16_max / row16_sum), not through LDS shuffles.
//   Measured at SD's shapes (1008 queries x 4096 keys x 8 heads x 40, batch 2: 10.6 GFLOP): rounds 4-5 217 us = 49 TFLOP/s; round 6
//   189 us = 56 TFLOP/s (0.36 of the fp32 MFMA peak; d = 40 fills 40 / 48 of the padded tiles), 160 x 1024 x 8 x 80: 38.3 -> 25.6 us
//   (tools/attention_tokens_bench.py, profiles/r6x_attention_tokens.jsonl): the K / V operands as branch-free buffer loads issued
//   one key block ahead (they were 15 loads behind exec-masked branches, waited for at once: -9 % / -12 % without the look-ahead,
//   another -5 % / -24 % with it), the (batch, head) pairs dealt to the XCDs (-1 %).  Variants measured and not kept (profiles/r4h_bench_sd.json,
//   r4i_bench_sd.json): 64 queries per workgroup with the K / V blocks staged through LDS (4x fewer L2 reads, but one wave per
//   SIMD: 1.5x slower); two query tiles per wave sharing each K / V fragment (template parameter QT: equal); LDS shuffles
//   instead of DPP (equal) -- the kernel is bound by the dependent chain scores -> softmax -> P through LDS -> values of a
//   wave, not by loads.
//
// q [B,Nq,C], k / v [B,Nk,C], out [B,Nq,C], C = heads * d, all row-major fp32 (a channels-last [B,C,H,W] tensor IS [B,HW,C];
// channels-last tiles [T,C,4,4] ARE [T*16, C]: no copy on either side).  Nq % 16 == 0; Nk arbitrary (tail keys masked).
#include "common.hpp"

namespace sige {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNITS, int QT>  // 16-channel units covering the head dimension (d <= 16 * UNITS); 16-query tiles per workgroup
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UNITS * QT <= 3 ? 4 : 1)))  // (SD's d = 40: 1 008 workgroups on 1 024 slots)
void attention_tokens_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                               const float *__restrict__ v, float *__restrict__ out,
                                                               int Nq, int Nk, int C, int heads, int d, float scale_log2e,
                                                               int q_tiles, int xcd_pairs) {
    constexpr int DT = UNITS;            // 16-column tiles of O
    constexpr int OS = UNITS * 16 + 4;   // padded row of the merge buffer
    constexpr int QR = 16 * QT;          // query rows of the workgroup
    __shared__ __attribute__((aligned(16))) float p_lds[4][QT][16][20];  // wave-private P tiles
    __shared__ float m_lds[4][QR], l_lds[4][QR];
    extern __shared__ __attribute__((aligned(16))) float o_lds[];        // [4 waves][QR queries][OS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, j = lane & 15;
    // workgroup -> (batch * head pair, query tile).  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with
    // its own 4 MB L2: with `xcd_pairs` > 0 (the host sets it when B * heads is a multiple of 8) the pairs are dealt to the XCDs --
    // pair = xcd * xcd_pairs + ... -- so that one L2 holds the K / V of ITS pairs (SD, 64 x 64 latent: 2 x 1.3 MB) instead of every
    // L2 streaming all of them (16 x 1.3 MB: each key block re-read from the Infinity Cache by every XCD)
    int pair, qtile;
    if (xcd_pairs > 0) {
        const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
        pair = xcd * xcd_pairs + local / q_tiles;
        qtile = local - (local / q_tiles) * q_tiles;
    } else {
        pair = blockIdx.x / q_tiles;
        qtile = blockIdx.x - pair * q_tiles;
    }
    const int head = pair % heads, b = pair / heads;
    const int q0 = qtile * QR;
    const size_t hoff = (size_t)head * d;
    const float *kb = k + (size_t)b * Nk * C + hoff;
    const float *vb = v + (size_t)b * Nk * C + hoff;

    // Q of this lane: row j of every query tile, channels 16u + 4kq .. +3 of every unit (zeros beyond d; a tile past Nq -- the
    // last workgroup of an odd tile count -- reads the last tile's rows and is not stored)
    float4 qr[QT][UNITS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qrow = min(q0 + 16 * t, Nq - 16) + j;
        const float *qb = q + ((size_t)b * Nq + qrow) * C + hoff;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int c = 16 * u + 4 * kq;
            qr[t][u] = c < d ? *reinterpret_cast<const float4 *>(qb + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // running softmax state of rows 4kq .. 4kq+3 of every tile (replicated over the 16 lanes j) and the O accumulators (C
    // layout: o[t][n][r] = O[row 16t + 4kq + r][column 16n + j])
    float m_run[QT][4], l_run[QT][4];
    f32x4 o[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { m_run[t][r] = -INFINITY; l_run[t][r] = 0.f; }
#pragma unroll
        for (int n = 0; n < DT; ++n) o[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nkb = (Nk + 15) / 16;
    // K and V of a key block: loaded ONCE, used by every query tile of the workgroup -- and one block AHEAD (the operands of block
    // kblk + 4 are requested before block kblk is computed on: a block's 15 loads come from another XCD's half of the Infinity Cache
    // as often as not, 1 - 2 us away, and nothing else of this wave can run under them: its next step needs exactly these values)
    // Branch-free buffer loads, one 32-bit offset register per key row: a lane whose channels lie past d (the padding of the last
    // unit) reads the next head's channels -- or, past the end of this batch's [Nk, C] matrix, the zeros a buffer load returns out of
    // range --, a lane whose key lies past Nk (the tail block) the last key's; what it reads meets a ZERO on the other side of the
    // product (Q is zero on padded channels, P is zero on masked keys) or lands in a padded column of O that is never stored
    const unsigned kv_bytes = (unsigned)(((size_t)Nk * C - hoff) * sizeof(float));
    const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(kb), 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vb), 0, kv_bytes, 0x00020000);
    const int row_bytes = C * (int)sizeof(float);
    float4 kn[UNITS];
    float vn[4][DT];
    auto fetch = [&](int kblk) {
        const int key0 = kblk * 16;
        const int ko = min(key0 + j, Nk - 1) * row_bytes + 16 * kq;
#pragma unroll
        for (int u = 0; u < UNITS; ++u)
            kn[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_k, ko + 64 * u, 0, 0));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int vo = min(key0 + 4 * t + kq, Nk - 1) * row_bytes + 4 * j;
#pragma unroll
            for (int n = 0; n < DT; ++n) vn[t][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_v, vo + 64 * n, 0, 0));
        }
    };
    if (wave < nkb) fetch(wave);
    for (int kblk = wave; kblk < nkb; kblk += 4) {
        const int key0 = kblk * 16;
        float4 kr[UNITS];
        float vr[4][DT];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) kr[u] = kn[u];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int n = 0; n < DT; ++n) vr[t][n] = vn[t][n];
        }
#ifndef SIGE_ATTENTION_NO_PREFETCH
        fetch(min(kblk + 4, nkb - 1));  // (past the end: the last block again -- a valid address, never used)
#else
        if (kblk + 4 < nkb) { __builtin_amdgcn_s_waitcnt(0); fetch(kblk + 4); __builtin_amdgcn_s_waitcnt(0); }
#endif
        const bool live = key0 + j < Nk;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            // ---- S = Q K^T ----
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[qt][u].x, kr[u].x, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[qt][u].y, kr[u].y, s1, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[qt][u].z, kr[u].z, s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qr[qt][u].w, kr[u].w, s1, 0, 0, 0);
            }
            // ---- online softmax; s = S[row 4kq + r][key0 + j] in units of log2: exp(x) = exp2(x * log2 e) ----
            float alpha[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv = live ? (s0[r] + s1[r]) * scale_log2e : -INFINITY;
                const float m_new = fmaxf(m_run[qt][r], row16_max(sv));  // (finite: every block has at least one live key)
                alpha[r] = __builtin_amdgcn_exp2f(m_run[qt][r] - m_new);  // (exp2(-inf) = 0 on the first block)
                const float p = __builtin_amdgcn_exp2f(sv - m_new);
                l_run[qt][r] = l_run[qt][r] * alpha[r] + row16_sum(p);
                m_run[qt][r] = m_new;
                p_lds[wave][qt][4 * kq + r][j] = p;
            }
#pragma unroll
            for (int n = 0; n < DT; ++n) {
                o[qt][n][0] *= alpha[0]; o[qt][n][1] *= alpha[1]; o[qt][n][2] *= alpha[2]; o[qt][n][3] *= alpha[3];
            }
        }
        __builtin_amdgcn_wave_barrier();  // (LDS is in order per wave: the tiles written above are complete for this wave's reads)
        // ---- O += P V: A[row j][k = key 4t + kq] from the LDS tile ----
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float a = p_lds[wave][qt][j][4 * t + kq];
#pragma unroll
                for (int n = 0; n < DT; ++n) o[qt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vr[t][n], o[qt][n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the four waves' (m, l, O) ----
    float *ow = o_lds + (size_t)wave * QR * OS;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * qt + 4 * kq + r;
            if (j == 0) { m_lds[wave][row] = m_run[qt][r]; l_lds[wave][row] = l_run[qt][r]; }
#pragma unroll
            for (int n = 0; n < DT; ++n) ow[row * OS + 16 * n + j] = o[qt][n][r];
        }
    }
    __syncthreads();
    // thread -> (query row, 4 consecutive channels)
    const int units_per_row = d / 4;
    const int rows = min(QR, Nq - q0);
    for (int e = tid; e < rows * units_per_row; e += 256) {
        const int row = e / units_per_row, c = (e - row * units_per_row) * 4;
        float M = fmaxf(fmaxf(m_lds[0][row], m_lds[1][row]), fmaxf(m_lds[2][row], m_lds[3][row]));
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = m_lds[w][row];
            const float f = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - M);  // (a wave without a key block)
            L += f * l_lds[w][row];
            const float4 ov = *reinterpret_cast<const float4 *>(o_lds + ((size_t)w * QR + row) * OS + c);
            acc.x += f * ov.x; acc.y += f * ov.y; acc.z += f * ov.z; acc.w += f * ov.w;
        }
        const float inv = 1.0f / L;
        *reinterpret_cast<float4 *>(out + ((size_t)b * Nq + q0 + row) * C + hoff + c) =
            make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}


// ---- round 6: the score tile TRANSPOSED -----------------------------------------------------------------------------------------
// S^T = K Q^T instead of S = Q K^T -- the same two register fragments, the MFMA operands swapped.  The accumulator layout then gives
// lane (kq, j) the scores of ONE query (j) against keys key0 + 4kq + r, r = 0..3, and that changes everything after the scores:
//   * the row maximum of a query is a maximum over a lane's own 4 values + over the 4 lane rows (two v_permlane*_swap, no DPP
//     ladder per score row, 4 rows per lane);  the row SUM is not reduced per block at all: each lane keeps the sum of its own keys,
//     the four rows meet once, in the merge;
//   * P^T in the accumulator layout IS the B operand of O^T = V^T P^T (k index = 4kq + r: the k-step r takes p[r] as it lies) -- the
//     probabilities never go through LDS;
//   * O^T's accumulator holds 4 CONSECUTIVE channels of query j per lane: the merge writes 16-byte rows.
// Per 16-key block and wave: ~60 VALU + 4 exp2 + 24 MFMA instead of ~190 VALU + 8 exp2 + 8 LDS ops + 24 MFMA.
// Measured (tools/attention_tokens_bench.py, profiles/r6z_attention_tokens.jsonl): SD's self-attention at 64 x 64 (1 008 queries x
// 4 096 keys x 8 heads x 40, batch 2) 186 -> 142 us = 74 TFLOP/s of useful work, 89 with the 40 -> 48 channel padding = 0.57 of the
// nominal fp32 MFMA peak; 4 096 queries: 686 -> 480 us = 90 / 108 TFLOP/s = 0.69.  Counters at the first shape
// (profiles/r6aa_attention_pmc.txt): L2 hit rate 98.5 % (the XCD dealing), SQ_VALU_MFMA_BUSY = 57 % of the SIMD cycles, 75 % of the
// wave cycles are issue stalls behind the matrix pipe -- what is left is the pipe itself, the padding, and 1 008 workgroups on
// 1 024 slots.
// (v_permlane32_swap a, b: a's lanes 32..63 <-> b's lanes 0..31; with a == b == v both end up holding a half of v twice, max(a, b)
//  is max(v[l], v[l ^ 32]) in every lane; v_permlane16_swap likewise for the odd / even rows of 16.  Inline asm: the clang builtin of
//  ROCm 7.2 returns its first result twice when both operands are copies of one value -- tools/probe/permlane_swap_probe.hip.)
__device__ __forceinline__ float max_over_rows(float v) {
#ifdef SIGE_ATTENTION_SHFL
    v = fmaxf(v, __shfl_xor(v, 32));
    return fmaxf(v, __shfl_xor(v, 16));
#else
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b); b = a;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
#endif
}

template <int UNITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(UNITS <= 3 ? 4 : 1)))
void attention_tokens_t_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                               float *__restrict__ out, int Nq, int Nk, int C, int heads, int d, float scale_log2e,
                               int q_tiles, int xcd_pairs) {
    constexpr int DT = UNITS;           // 16-channel tiles of O
    constexpr int OS = UNITS * 16 + 4;  // padded row of the merge buffer (a multiple of 4 floats: 16-byte rows)
    __shared__ float m_lds[4][16], l_lds[4][4][16];
    extern __shared__ __attribute__((aligned(16))) float o_lds[];  // [4 waves][16 queries][OS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, j = lane & 15;
    int pair, qtile;  // (workgroup -> XCD: as in attention_tokens_kernel)
    if (xcd_pairs > 0) {
        const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
        pair = xcd * xcd_pairs + local / q_tiles;
        qtile = local - (local / q_tiles) * q_tiles;
    } else {
        pair = blockIdx.x / q_tiles;
        qtile = blockIdx.x - pair * q_tiles;
    }
    const int head = pair % heads, b = pair / heads;
    const int q0 = qtile * 16;
    const size_t hoff = (size_t)head * d;
    const float *kb = k + (size_t)b * Nk * C + hoff;
    const float *vb = v + (size_t)b * Nk * C + hoff;

    // Q of this lane = the B operand: query j, channels 16u + 4kq .. +3 of every unit (zeros beyond d)
    float4 qr[UNITS];
    {
        const float *qb = q + ((size_t)b * Nq + q0 + j) * C + hoff;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int c = 16 * u + 4 * kq;
            const float4 t = *reinterpret_cast<const float4 *>(qb + min(c, d - 4));
            qr[u] = c < d ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float m_run = -INFINITY, l_run = 0.f;  // of query j: the maximum over every key so far, the sum over THIS lane's keys
    f32x4 o[DT];                           // o[n][r] = O[query j][channel 16n + 4kq + r]
#pragma unroll
    for (int n = 0; n < DT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = (Nk + 15) / 16;
    const unsigned kv_bytes = (unsigned)(((size_t)Nk * C - hoff) * sizeof(float));
    const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(kb), 0, kv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vb), 0, kv_bytes, 0x00020000);
    const int row_bytes = C * (int)sizeof(float);
    float4 kn[UNITS];  // the A operand of the scores: key j, the same channels as qr
    float vn[4][DT];   // the A operand of O^T, k-step r: V[key0 + 4kq + r][16n + j]
    auto fetch = [&](int kblk) {  // (branch-free, one block ahead: see attention_tokens_kernel)
        const int key0 = kblk * 16;
        const int ko = min(key0 + j, Nk - 1) * row_bytes + 16 * kq;
#pragma unroll
        for (int u = 0; u < UNITS; ++u)
            kn[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_k, ko + 64 * u, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int vo = min(key0 + 4 * kq + r, Nk - 1) * row_bytes + 4 * j;
#pragma unroll
            for (int n = 0; n < DT; ++n) vn[r][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_v, vo + 64 * n, 0, 0));
        }
    };
    if (wave < nkb) fetch(wave);
    for (int kblk = wave; kblk < nkb; kblk += 4) {
        const int key0 = kblk * 16;
        float4 kr[UNITS];
        float vr[4][DT];
#pragma unroll
        for (int u = 0; u < UNITS; ++u) kr[u] = kn[u];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int n = 0; n < DT; ++n) vr[r][n] = vn[r][n];
        }
        fetch(min(kblk + 4, nkb - 1));  // (past the end: the last block again -- a valid address, never used)
        // ---- S^T = K Q^T: s[r] = S[query j][key0 + 4kq + r] ----
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[u].x, qr[u].x, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[u].y, qr[u].y, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[u].z, qr[u].z, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kr[u].w, qr[u].w, s1, 0, 0, 0);
        }
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = (s0[r] + s1[r]) * scale_log2e;  // (units of log2: exp(x) = exp2(x * log2 e))
        if (key0 + 16 > Nk) {  // (wave-uniform: the tail block only)
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = key0 + 4 * kq + r < Nk ? sv[r] : -INFINITY;
        }
        // ---- online softmax of query j ----
        const float mb = max_over_rows(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));  // (finite: a block has a live key)
        if (__builtin_amdgcn_ballot_w64(mb > m_run) != 0) {  // (wave-uniform; once the maxima have settled no block enters)
            const float m_new = fmaxf(m_run, mb);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // (exp2(-inf) = 0 on the first block; 1 for an unchanged row)
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int n = 0; n < DT; ++n) { o[n][0] *= alpha; o[n][1] *= alpha; o[n][2] *= alpha; o[n][3] *= alpha; }
        }
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(sv[r] - m_run);
        l_run += (pr[0] + pr[1]) + (pr[2] + pr[3]);
        // ---- O^T += V^T P^T: k-step r contracts keys key0 + 4kq + r ----
        // (issue order -- one score chain or two, key-step-major or channel-tile-major here -- measured: no difference,
        //  profiles/r6ab_attention_mfma_order.jsonl)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int n = 0; n < DT; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[r][n], pr[r], o[n], 0, 0, 0);
        }
    }

    // ---- merge the four waves' (m, l, O) ----
    float *ow = o_lds + (size_t)wave * 16 * OS;
    if (kq == 0) m_lds[wave][j] = m_run;
    l_lds[wave][kq][j] = l_run;
#pragma unroll
    for (int n = 0; n < DT; ++n)
        *reinterpret_cast<float4 *>(ow + j * OS + 16 * n + 4 * kq) = make_float4(o[n][0], o[n][1], o[n][2], o[n][3]);
    __syncthreads();
    // thread -> (query row, 4 consecutive channels)
    const int units_per_row = d / 4;
    for (int e = tid; e < 16 * units_per_row; e += 256) {
        const int row = e / units_per_row, c = (e - row * units_per_row) * 4;
        const float M = fmaxf(fmaxf(m_lds[0][row], m_lds[1][row]), fmaxf(m_lds[2][row], m_lds[3][row]));
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = m_lds[w][row];
            const float f = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - M);  // (a wave without a key block)
            L += f * ((l_lds[w][0][row] + l_lds[w][1][row]) + (l_lds[w][2][row] + l_lds[w][3][row]));
            const float4 ov = *reinterpret_cast<const float4 *>(o_lds + ((size_t)w * 16 + row) * OS + c);
            acc.x += f * ov.x; acc.y += f * ov.y; acc.z += f * ov.z; acc.w += f * ov.w;
        }
        const float inv = 1.0f / L;
        *reinterpret_cast<float4 *>(out + ((size_t)b * Nq + q0 + row) * C + hoff + c) =
            make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

}  // namespace sige

using namespace sige;

extern "C" int sige_hip_attention_tokens_supported(int Nq, int Nk, int C, int heads) {
    if (Nq <= 0 || Nk <= 0 || C <= 0 || heads <= 0 || C % heads) return 0;
    const int d = C / heads;
    return (Nq % 16 == 0 && d % 4 == 0 && d <= 160 && C % 4 == 0) ? 1 : 0;
}

extern "C" int sige_hip_attention_tokens_f32(const float *q, const float *k, const float *v, int B, int Nq, int Nk, int C,
                                             int heads, float scale, float *out, void *stream) {
    SIGE_PLAN_HOOK(sige_hip_attention_tokens_f32, q, k, v, B, Nq, Nk, C, heads, scale, out, stream);
    if (B < 0 || Nq < 0 || Nk <= 0 || C <= 0 || heads <= 0) return SIGE_HIP_EINVAL;
    if ((long)B * Nq == 0) return SIGE_HIP_OK;
    if (!q || !k || !v || !out) return SIGE_HIP_EINVAL;
    if (!sige_hip_attention_tokens_supported(Nq, Nk, C, heads)) return SIGE_HIP_EUNSUPPORTED;
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al(q) || !al(k) || !al(v) || !al(out) || (size_t)Nk * C * sizeof(float) >= 0x7fffffffu) return SIGE_HIP_EUNSUPPORTED;
    const int d = C / heads;
    const int units = (d + 15) / 16;
    const float sl = scale * 1.44269504088896341f;
    hipStream_t st = as_stream(stream);
    // forms: 0 (automatic) = the transposed-score kernel (round 6); 1 = round 4's 16 queries per workgroup with P through LDS;
    // 2 = 32 queries per workgroup (two query tiles share every K / V fragment a wave loads; d <= 96) -- measured equal to form 1
    // at SD's shapes (profiles/r4i_bench_sd.json).  1 and 2 are kept for the A/B in the measurement build.
    const int knob = tuning(SIGE_HIP_TUNE_ATTENTION_FORM);
    const int form = (knob == 2 && units <= 6) ? 2 : (knob == 1 ? 1 : 0);
    const int pairs = B * heads, t16 = Nq / 16, t32 = (Nq + 31) / 32;
    if ((long)pairs * t16 > 0x7fffffffL) return SIGE_HIP_EUNSUPPORTED;
#ifdef SIGE_ATTENTION_PLAIN_ORDER
    const int xp = 0;
#else
    const int xp = pairs % 8 == 0 ? pairs / 8 : 0;
#endif
#define SIGE_ATT_GO(U)                                                                                             \
    do {                                                                                                           \
        if (form == 0) attention_tokens_t_kernel<U><<<dim3(t16 * pairs), 256, (size_t)4 * 16 * (U * 16 + 4) * sizeof(float), st>>>(q, k, v, out, Nq, Nk, C, heads, d, sl, t16, xp); \
        else if (form == 2) attention_tokens_kernel<(U <= 6 ? U : 1), 2><<<dim3(t32 * pairs), 256, (size_t)4 * 32 * (U * 16 + 4) * sizeof(float), st>>>(q, k, v, out, Nq, Nk, C, heads, d, sl, t32, xp); \
        else attention_tokens_kernel<U, 1><<<dim3(t16 * pairs), 256, (size_t)4 * 16 * (U * 16 + 4) * sizeof(float), st>>>(q, k, v, out, Nq, Nk, C, heads, d, sl, t16, xp); \
    } while (0)
    switch (units) {
        case 1: SIGE_ATT_GO(1); break;
        case 2: SIGE_ATT_GO(2); break;
        case 3: SIGE_ATT_GO(3); break;
        case 4: SIGE_ATT_GO(4); break;
        case 5: SIGE_ATT_GO(5); break;
        case 6: SIGE_ATT_GO(6); break;
        case 8: SIGE_ATT_GO(8); break;
        case 10: SIGE_ATT_GO(10); break;
        default: return SIGE_HIP_EUNSUPPORTED;
    }
#undef SIGE_ATT_GO
    return launch_status();
}
