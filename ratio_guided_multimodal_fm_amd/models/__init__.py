"""Model API of the sampler path (mirrors the reference ``src/models``)."""
from .flow_matching import FlowMatchingModel  # noqa: F401
from .unet import FlowMatchingUNet, UNetMNIST  # noqa: F401
from .unet_flexible import (FlexibleUNet, FlowMatchingUNetMNIST, FlowMatchingUNetSVHN,  # noqa: F401
                            timestep_embedding)
from .ratio_estimator import RatioEstimator  # noqa: F401
from .ratio_flexible import RatioEstimatorMNISTSVHN  # noqa: F401
from .svhn_classifier import MNISTClassifier32, SVHNClassifier  # noqa: F401
from .classifier import MNISTClassifier  # noqa: F401
