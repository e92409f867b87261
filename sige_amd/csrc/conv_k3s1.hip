// 3x3 stride-1 tiles (6x6 in), NCHW: explicit instantiations of the MFMA stacked-block conv.
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeo<3, 1, 6, 16>;
using G32 = ConvGeo<3, 1, 6, 32>;
SIGE_CONV_INSTANTIATE(G16, 1, LAYOUT_NCHW, 4)
SIGE_CONV_INSTANTIATE(G32, 1, LAYOUT_NCHW, 4)
SIGE_CONV_INSTANTIATE(G16, 2, LAYOUT_NCHW, 4)
SIGE_CONV_INSTANTIATE(G32, 2, LAYOUT_NCHW, 4)
}  // namespace sige
