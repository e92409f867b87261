// conv_wide.hpp instantiated for the 3x3 dense convs, 8 x 8 output pixels per workgroup: fp16 operands, split fp16 operands
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(3, WIDE_F16, 8)
SIGE_WIDE_INSTANTIATE(3, WIDE_X3, 8)
}  // namespace sige
