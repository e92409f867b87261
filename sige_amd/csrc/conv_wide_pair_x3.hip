// ... split fp16 operands for conv1, exact-fp32 shortcut (a dense layer's 1x1 is below the split-operand threshold).
#include "conv_wide.hpp"
namespace sige {
using PK11_16 = ConvGeo<1, 1, 4, 16>;
using PK11_32 = ConvGeo<1, 1, 4, 32>;
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_X3, PK11_16)
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_X3, PK11_32)
}  // namespace sige
