// conv_wide.hpp instantiated for the 1x1 dense convs, 8 x 8 output pixels per workgroup (fp16 operands; split fp16 operands)
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(1, false, 8)
SIGE_WIDE_INSTANTIATE(1, true, 8)
}  // namespace sige
