"""Velocity U-Net, host side.

API mirror of the reference ``src/models/unet_flexible.py`` (FlexibleUNet
``:111-261``, presets ``:266-291``, ``timestep_embedding`` ``:16-36``): same
constructor arguments, same ``state_dict`` keys and shapes, same
``forward(x_t[B,C,H,W], t[B]) -> v[B,C,H,W]`` contract.  The modules below are
parameter containers only; ``forward`` packs the parameters into the C-ABI
library (``include/rgfm.h``) and runs the hand-written HIP path.  There is no
PyTorch implementation of the network in this package.
"""
import math

import torch
import torch.nn as nn

from .._engine import UNetEngine, engine_property


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding, cos half first (reference unet_flexible.py:16-36).

    Host helper kept for API parity; the sampler computes the same table on
    the device (csrc/unet_kernels.hip (time_embed_kernel)) from the frequency table built here.
    """
    half = dim // 2
    freqs = embedding_freqs(dim, max_period).to(timesteps.device)
    args = timesteps[:, None] * freqs[None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def embedding_freqs(dim, max_period=10000):
    """fp32 frequency table exp(-ln(max_period) * i / half), i < dim // 2."""
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(half) / half)


class _ResBlockParams(nn.Module):
    """Parameters of one residual block (reference ResBlock, :39-85)."""

    def __init__(self, cin, cout, temb_dim, dropout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = nn.GroupNorm(min(8, cin), cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(temb_dim, cout))
        self.norm2 = nn.GroupNorm(min(8, cout), cout)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.skip = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()


class _ConvHolder(nn.Module):
    """Downsample (:88-96) / Upsample (:99-108): a single 3x3 conv named `conv`."""

    def __init__(self, ch, stride):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=1)


class FlexibleUNet(nn.Module):
    """U-Net velocity field v(x_t, t) evaluated by the HIP library."""

    _engine = engine_property(lambda m: UNetEngine(m))

    def __init__(self, in_channels=1, img_size=28, model_channels=32, channel_mult=(1, 2),
                 num_res_blocks=2, dropout=0.1):
        super().__init__()
        self.in_channels = in_channels
        self.img_size = img_size
        self.model_channels = model_channels
        self.channel_mult = tuple(channel_mult)
        self.num_res_blocks = num_res_blocks

        temb = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, temb), nn.SiLU(), nn.Linear(temb, temb))
        self.input_conv = nn.Conv2d(in_channels, model_channels, 3, padding=1)

        self.encoder_blocks = nn.ModuleList()
        self.downsamplers = nn.ModuleList()
        ch = model_channels
        skips = [ch]
        last = len(self.channel_mult) - 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                self.encoder_blocks.append(_ResBlockParams(ch, model_channels * mult, temb, dropout))
                ch = model_channels * mult
                skips.append(ch)
            if level < last:
                self.downsamplers.append(_ConvHolder(ch, 2))
                skips.append(ch)

        self.middle_block1 = _ResBlockParams(ch, ch, temb, dropout)
        self.middle_block2 = _ResBlockParams(ch, ch, temb, dropout)

        self.decoder_blocks = nn.ModuleList()
        self.upsamplers = nn.ModuleList()
        for level in range(last, -1, -1):
            out_ch = model_channels * self.channel_mult[level]
            for _ in range(num_res_blocks + 1):
                self.decoder_blocks.append(_ResBlockParams(ch + skips.pop(), out_ch, temb, dropout))
                ch = out_ch
            if level > 0:
                self.upsamplers.append(_ConvHolder(ch, 1))

        self.out_norm = nn.GroupNorm(min(8, ch), ch)
        self.out_conv = nn.Conv2d(ch, in_channels, 3, padding=1)
        # the reference zero-initialises the output conv (:200-201)
        nn.init.zeros_(self.out_conv.weight)
        nn.init.zeros_(self.out_conv.bias)


    def forward(self, x, t):
        """x: [B,C,H,W] fp32 on a HIP device, t: [B] (or [1]) -> velocity [B,C,H,W]."""
        return self._engine.forward(x, t)


class FlowMatchingUNetMNIST(FlexibleUNet):
    """MNIST preset, 1x28x28 or 1x32x32 (reference :266-277)."""

    def __init__(self, img_size=28):
        super().__init__(in_channels=1, img_size=img_size, model_channels=32, channel_mult=(1, 2),
                         num_res_blocks=2, dropout=0.1)


class FlowMatchingUNetSVHN(FlexibleUNet):
    """SVHN preset, 3x32x32 (reference :280-291)."""

    def __init__(self):
        super().__init__(in_channels=3, img_size=32, model_channels=64, channel_mult=(1, 2, 2),
                         num_res_blocks=2, dropout=0.1)
