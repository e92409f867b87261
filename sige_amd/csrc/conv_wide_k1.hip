// conv_wide.hpp instantiated for the 1x1 dense convs (fp16 operands; split fp16 operands)
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(1, false)
SIGE_WIDE_INSTANTIATE(1, true)
}  // namespace sige
