from .base import SIGEConv2d, SIGEModel, SIGEModule, SIGEModuleWrapper, paired_convs  # noqa: F401
from .gather import Gather  # noqa: F401
from .scatter import Scatter, ScatterWithBlockResidual  # noqa: F401
from .scatter_gather import ScatterGather  # noqa: F401
