// 3x3 stride-1 tiles (6x6 in), channels-last, exact fp32, scatter_gather source whose cached tensor is stored as fp16
// (the "_c16" entry points: SURVEY.md 8f row 4 -- fp16 resident cache): explicit instantiations.
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeo<3, 1, 6, 16>;
using G32 = ConvGeo<3, 1, 6, 32>;
SIGE_CONV_INSTANTIATE_C16(G16, 1)
SIGE_CONV_INSTANTIATE_C16(G32, 1)
SIGE_CONV_INSTANTIATE_C16(G16, 2)
SIGE_CONV_INSTANTIATE_C16(G32, 2)
}  // namespace sige
