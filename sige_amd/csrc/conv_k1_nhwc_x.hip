// 1x1 stride-1 tiles (4x4 in), channels-last, split fp16 operands (ConvGeoX): explicit instantiations.
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeoX<1, 1, 4, 16>;
using G32 = ConvGeoX<1, 1, 4, 32>;
SIGE_CONV_INSTANTIATE(G16, 1, LAYOUT_NHWC, 4)
SIGE_CONV_INSTANTIATE(G32, 1, LAYOUT_NHWC, 4)
}  // namespace sige
