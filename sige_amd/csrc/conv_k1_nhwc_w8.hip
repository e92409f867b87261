// 1x1 stride-1 tiles, channels-last, 8-wave workgroups (two waves per SIMD): explicit instantiations.
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeo<1, 1, 4, 16>;
using G32 = ConvGeo<1, 1, 4, 32>;
SIGE_CONV_INSTANTIATE(G16, 1, LAYOUT_NHWC, 8)
SIGE_CONV_INSTANTIATE(G16, 2, LAYOUT_NHWC, 8)
SIGE_CONV_INSTANTIATE(G32, 1, LAYOUT_NHWC, 8)
SIGE_CONV_INSTANTIATE(G32, 2, LAYOUT_NHWC, 8)
}  // namespace sige
