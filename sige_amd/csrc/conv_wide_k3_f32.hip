// conv_wide.hpp instantiated for the 3x3 dense convs, 8 x 8 output pixels per workgroup: exact fp32 (v_mfma_f32_32x32x2_f32)
#include "conv_wide.hpp"
namespace sige {
SIGE_WIDE_INSTANTIATE(3, WIDE_F32, 8)
}  // namespace sige
