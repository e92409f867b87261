// conv_tile3.hpp instantiated: fp16 operands (v_mfma_f32_32x32x16_f16, fp32 accumulation), two tiles (32 pixels) x 64 output
// channels per workgroup; scatter_gather source with the cached tensor in fp32 or in halves (fp16-stored caches).
#include "conv_tile3.hpp"
namespace sige {
using H2 = Tile3Geo<2, WIDE_F16>;
template <> void launch_conv_tile3_gather<2, WIDE_F16>(const Tile3Args &a, bool aff, bool cat, bool full, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, H2::TPW) * a.ntn);
#define SIGE_T3(AFF, CAT)                                                                                      \
    do {                                                                                                       \
        if (full) conv_tile3_kernel<H2, T3_GATHER, AFF, CAT, true><<<grid, 256, 0, st>>>(a);                   \
        else conv_tile3_kernel<H2, T3_GATHER, AFF, CAT, false><<<grid, 256, 0, st>>>(a);                       \
    } while (0)
    if (aff && cat) SIGE_T3(true, true);
    else if (aff) SIGE_T3(true, false);
    else if (cat) SIGE_T3(false, true);
    else SIGE_T3(false, false);
#undef SIGE_T3
}
template <> void launch_conv_tile3_sg<2, WIDE_F16>(const Tile3Args &a, bool full, bool y16, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, H2::TPW) * a.ntn);
#define SIGE_T3S(AFF)                                                                                                   \
    do {                                                                                                                \
        if (y16) {                                                                                                      \
            if (full) conv_tile3_kernel<H2, T3_SCATTER_GATHER, AFF, false, true, true><<<grid, 256, 0, st>>>(a);        \
            else conv_tile3_kernel<H2, T3_SCATTER_GATHER, AFF, false, false, true><<<grid, 256, 0, st>>>(a);            \
        } else {                                                                                                        \
            if (full) conv_tile3_kernel<H2, T3_SCATTER_GATHER, AFF, false, true><<<grid, 256, 0, st>>>(a);              \
            else conv_tile3_kernel<H2, T3_SCATTER_GATHER, AFF, false, false><<<grid, 256, 0, st>>>(a);                  \
        }                                                                                                               \
    } while (0)
    if (a.scale) SIGE_T3S(true);
    else SIGE_T3S(false);
#undef SIGE_T3S
}
}  // namespace sige

namespace sige {

// four tiles (64 pixels) x 64 output channels per workgroup: a weight byte pulled from L2 feeds twice the matrix work -- for launches
// whose 2-tile grid covers the chip several times over (tile_conv3_launch: kTile3F16Tpw4Min)
using H4 = Tile3Geo<4, WIDE_F16>;
template <> void launch_conv_tile3_gather<4, WIDE_F16>(const Tile3Args &a, bool aff, bool cat, bool full, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, H4::TPW) * a.ntn);
#define SIGE_T3(AFF, CAT)                                                                                      \
    do {                                                                                                       \
        if (full) conv_tile3_kernel<H4, T3_GATHER, AFF, CAT, true><<<grid, 256, 0, st>>>(a);                   \
        else conv_tile3_kernel<H4, T3_GATHER, AFF, CAT, false><<<grid, 256, 0, st>>>(a);                       \
    } while (0)
    if (aff && cat) SIGE_T3(true, true);
    else if (aff) SIGE_T3(true, false);
    else if (cat) SIGE_T3(false, true);
    else SIGE_T3(false, false);
#undef SIGE_T3
}
template <> void launch_conv_tile3_sg<4, WIDE_F16>(const Tile3Args &a, bool full, bool y16, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, H4::TPW) * a.ntn);
    if (y16) {
        if (full) conv_tile3_kernel<H4, T3_SCATTER_GATHER, false, false, true, true><<<grid, 256, 0, st>>>(a);
        else conv_tile3_kernel<H4, T3_SCATTER_GATHER, false, false, false, true><<<grid, 256, 0, st>>>(a);
    } else {
        if (full) conv_tile3_kernel<H4, T3_SCATTER_GATHER, false, false, true><<<grid, 256, 0, st>>>(a);
        else conv_tile3_kernel<H4, T3_SCATTER_GATHER, false, false, false><<<grid, 256, 0, st>>>(a);
    }
}
}  // namespace sige
