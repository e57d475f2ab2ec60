"""28x28 U-Net entry points (reference ``src/models/unet.py``).

``UNetMNIST`` (``:122-278``) and ``FlowMatchingUNet`` (``:282-305``) have the
same parameters as ``FlexibleUNet`` at ``img_size=28`` (GroupNorm(8, C) equals
GroupNorm(min(8, C), C) for every C >= 8 used), so both map onto the same HIP
network description.
"""
from .unet_flexible import FlexibleUNet, timestep_embedding  # noqa: F401


class UNetMNIST(FlexibleUNet):
    def __init__(self, in_channels=1, model_channels=32, channel_mult=(1, 2, 2), num_res_blocks=2,
                 dropout=0.0):
        super().__init__(in_channels=in_channels, img_size=28, model_channels=model_channels,
                         channel_mult=channel_mult, num_res_blocks=num_res_blocks, dropout=dropout)


class FlowMatchingUNet(UNetMNIST):
    """Default velocity net of ``src/sample.py`` (``--model unet``)."""

    def __init__(self, img_channels=1, model_channels=32, channel_mult=(1, 2), num_res_blocks=2,
                 dropout=0.1):
        super().__init__(in_channels=img_channels, model_channels=model_channels,
                         channel_mult=channel_mult, num_res_blocks=num_res_blocks, dropout=dropout)
