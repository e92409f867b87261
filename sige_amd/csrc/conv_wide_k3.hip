// conv_wide.hpp instantiated for the 3x3 dense convs (fp16 operands; split fp16 operands)
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(3, false)
SIGE_WIDE_INSTANTIATE(3, true)
}  // namespace sige
