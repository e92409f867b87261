// conv_wide.hpp instantiated for the 1x1 dense convs, 8 x 8 output pixels per workgroup: fp16, split fp16 and exact fp32 operands
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(1, WIDE_F16, 8)
SIGE_WIDE_INSTANTIATE(1, WIDE_X3, 8)
SIGE_WIDE_INSTANTIATE(1, WIDE_F32, 8)
}  // namespace sige
