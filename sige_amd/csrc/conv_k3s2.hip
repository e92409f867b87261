// 3x3 stride-2 tiles (5x5 in), NCHW: explicit instantiations of the MFMA stacked-block conv.
#include "conv_mfma.hpp"
namespace sige {
using G16 = ConvGeo<3, 2, 5, 16>;
using G32 = ConvGeo<3, 2, 5, 32>;
SIGE_CONV_INSTANTIATE(G16, 1, LAYOUT_NCHW, 4)
SIGE_CONV_INSTANTIATE(G32, 1, LAYOUT_NCHW, 4)
// (25 staging slots per lane in NCHW: only the single-accumulator forms)
}  // namespace sige
