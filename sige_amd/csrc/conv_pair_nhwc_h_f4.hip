// conv1 + 1x1 shortcut of a residual block in one launch, f16 compute (ConvGeoH): explicit instantiations of
// conv_pair_kernel, destination = full tensor, 4 waves per workgroup.
#include "conv_mfma.hpp"
namespace sige {
using A16 = ConvGeoH<3, 1, 6, 16>;
using A32 = ConvGeoH<3, 1, 6, 32>;
using B16 = ConvGeoH<1, 1, 4, 16>;
using B32 = ConvGeoH<1, 1, 4, 32>;
SIGE_CONV_PAIR_INSTANTIATE(A16, 1, B16, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A16, 1, B32, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A16, 2, B16, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A16, 2, B32, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A32, 1, B16, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A32, 1, B32, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A32, 2, B16, DST_NCHW, 4)
SIGE_CONV_PAIR_INSTANTIATE(A32, 2, B32, DST_NCHW, 4)
}  // namespace sige
