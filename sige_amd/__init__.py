"""sige_amd -- MI355X-native implementation of SIGE's tiling-based sparse
convolution path (Gather / Scatter / ScatterGather / ScatterWithBlockResidual /
SIGEConv2d / SIGEModel) behind the reference's own `sige.nn` module API.

    from sige_amd.nn import Gather, Scatter, ScatterGather, ScatterWithBlockResidual
    from sige_amd.nn import SIGEConv2d, SIGEModel, SIGEModule
    from sige_amd.utils import reduce_mask, dilate_mask, downsample_mask, compute_difference_mask

`sige_amd.compat.install()` additionally registers the package under the name
`sige`, so model files written against the reference (`from sige.nn import ...`)
load unchanged.
"""
__version__ = "0.2.0"

from . import nn, utils  # noqa: E402,F401
