// A dense 3x3 conv1 on the dense-layer kernel (exact fp32 / split fp16 operands) with the block's exact-fp32 1x1 shortcut (tile
// kernel body) in the same launch: explicit instantiations (conv_wide.hpp: conv_wide_pair_kernel).
#include "conv_wide.hpp"
namespace sige {
using PK11_16 = ConvGeo<1, 1, 4, 16>;
using PK11_32 = ConvGeo<1, 1, 4, 32>;
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_F32, PK11_16)
SIGE_WIDE_PAIR_INSTANTIATE(WIDE_F32, PK11_32)
}  // namespace sige
