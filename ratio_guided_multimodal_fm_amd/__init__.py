"""MI355X-native ratio-guided flow-matching sampler.

Drop-in for the sampler path of foubari/ratio_guided_Multimodal_FM
(``src/utils/flow_utils.py`` + ``src/sample*.py``): the module layout below
mirrors the reference's ``src.models.*`` / ``src.utils.*`` import paths, the
classes keep the reference constructor arguments and ``state_dict`` keys, and
every ``forward`` / ``sample*`` call goes through the C-ABI library
``csrc/librgfm_hip.so`` (hand-written HIP for gfx950, declared in
``include/rgfm.h``).  There is no CPU or eager-PyTorch fallback: if the
library is missing, or a tensor is not on a HIP device, the call raises.
"""
from . import models, utils  # noqa: F401
from .utils.flow_utils import CFMSchedule, sample_bimodal_guided  # noqa: F401
from .sample_mnist_svhn import sample_bimodal_guided_mnist_svhn  # noqa: F401

__all__ = [
    "models",
    "utils",
    "CFMSchedule",
    "sample_bimodal_guided",
    "sample_bimodal_guided_mnist_svhn",
]
