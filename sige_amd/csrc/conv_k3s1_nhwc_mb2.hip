// 3x3 stride-1 tiles (6x6 in), NHWC, 64 pixel x 32 channel output blocks (two M tiles per workgroup sharing the weight registers):
// explicit instantiations of the MFMA stacked-block conv and of its pair kernel.
#include "conv_mfma.hpp"
namespace sige {
SIGE_CONV_INSTANTIATE_MB2(SRC_GATHER, DST_TILES)
SIGE_CONV_INSTANTIATE_MB2(SRC_GATHER, DST_NCHW)
SIGE_CONV_INSTANTIATE_MB2(SRC_SCATTER_GATHER, DST_TILES)
SIGE_CONV_INSTANTIATE_MB2(SRC_SCATTER_GATHER, DST_NCHW)
}  // namespace sige
