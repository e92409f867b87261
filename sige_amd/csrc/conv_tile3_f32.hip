// conv_tile3.hpp instantiated: exact fp32, two tiles (32 pixels) x 64 output channels per workgroup.
#include "conv_tile3.hpp"
namespace sige {
using G2 = Tile3Geo<2>;
template <> void launch_conv_tile3_gather<2>(const Tile3Args &a, bool aff, bool cat, bool full, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, G2::TPW) * a.ntn);
#define SIGE_T3(AFF, CAT)                                                                                      \
    do {                                                                                                       \
        if (full) conv_tile3_kernel<G2, T3_GATHER, AFF, CAT, true><<<grid, 256, 0, st>>>(a);                   \
        else conv_tile3_kernel<G2, T3_GATHER, AFF, CAT, false><<<grid, 256, 0, st>>>(a);                       \
    } while (0)
    if (aff && cat) SIGE_T3(true, true);
    else if (aff) SIGE_T3(true, false);
    else if (cat) SIGE_T3(false, true);
    else SIGE_T3(false, false);
#undef SIGE_T3
}
template <> void launch_conv_tile3_sg<2>(const Tile3Args &a, bool full, bool, hipStream_t st) {
    const dim3 grid(ceil_div(a.T, G2::TPW) * a.ntn);
    if (a.scale) {  // (the cached affine (+ SiLU) applied to the scatter_gathered window: scatter_gather.cpp:58-84)
        if (full) conv_tile3_kernel<G2, T3_SCATTER_GATHER, true, false, true><<<grid, 256, 0, st>>>(a);
        else conv_tile3_kernel<G2, T3_SCATTER_GATHER, true, false, false><<<grid, 256, 0, st>>>(a);
    } else {
        if (full) conv_tile3_kernel<G2, T3_SCATTER_GATHER, false, false, true><<<grid, 256, 0, st>>>(a);
        else conv_tile3_kernel<G2, T3_SCATTER_GATHER, false, false, false><<<grid, 256, 0, st>>>(a);
    }
}
}  // namespace sige
