// ... fp16 operands for conv1 and the shortcut.
#include "conv_wide.hpp"
namespace sige {
using PH11_16 = ConvGeoH<1, 1, 4, 16>;
using PH11_32 = ConvGeoH<1, 1, 4, 32>;
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_F16, PH11_16)
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_F16, PH11_32)
}  // namespace sige
