// 3x3 stride-1 tiles (6x6 in), channels-last, split fp16 operands (ConvGeoX: fp32-level results on the fp16 matrix cores):
// explicit instantiations (one output-channel sub-block per workgroup: the (hi, lo) weight registers of two do not fit).
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeoX<3, 1, 6, 16>;
using G32 = ConvGeoX<3, 1, 6, 32>;
SIGE_CONV_INSTANTIATE(G16, 1, LAYOUT_NHWC, 4)
SIGE_CONV_INSTANTIATE(G32, 1, LAYOUT_NHWC, 4)
SIGE_CONV_INSTANTIATE_SG_FULL(G16, 1, LAYOUT_NHWC, 4)
SIGE_CONV_INSTANTIATE_SG_FULL(G32, 1, LAYOUT_NHWC, 4)
}  // namespace sige
