// conv1 (3x3, gather; 64 pixel x 32 channel blocks) + 1x1 shortcut of a residual block in one launch.
#include "conv_mfma.hpp"
namespace sige {
using B16 = ConvGeo<1, 1, 4, 16>;
using B32 = ConvGeo<1, 1, 4, 32>;
SIGE_CONV_PAIR_INSTANTIATE_MB2(B16, DST_TILES)
SIGE_CONV_PAIR_INSTANTIATE_MB2(B32, DST_TILES)
SIGE_CONV_PAIR_INSTANTIATE_MB2(B16, DST_NCHW)
SIGE_CONV_PAIR_INSTANTIATE_MB2(B32, DST_NCHW)
}  // namespace sige
