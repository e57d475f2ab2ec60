"""``FlowMatchingModel`` -- the reference's "original" encoder-decoder velocity net
(``src/models/flow_matching.py:127-173``, selected by ``src/sample.py --model original``).

Parameter container with the reference's constructor arguments, ``state_dict`` keys and shapes (so
reference checkpoints load); ``forward`` and the samplers run in librgfm_hip.so (``rgfm_fmnet_*``):
the stride-2 convs and both ConvTranspose2d(k4,s2,p1) on the MFMA implicit-GEMM conv kernel (the
transposed convs as four 2x2-tap output-parity classes), the 12544<->256/384 Linears on the MFMA
linear kernels with their weights re-indexed to NHWC once at handle creation.  No PyTorch compute path.
"""
import torch.nn as nn

from .._engine import FmNetEngine, engine_property


class SinusoidalPositionEmbeddings(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim


class ImageEncoder(nn.Module):
    def __init__(self, in_channels=1, feature_dim=256):
        super().__init__()
        plan = [(in_channels, 32, 1), (32, 64, 2), (64, 128, 2), (128, 256, 1)]
        for i, (ci, co, st) in enumerate(plan, 1):
            setattr(self, f"conv{i}", nn.Conv2d(ci, co, 3, stride=st, padding=1))
            setattr(self, f"gn{i}", nn.GroupNorm(8, co))
        self.fc = nn.Linear(256 * 7 * 7, feature_dim)


class VelocityDecoder(nn.Module):
    def __init__(self, feature_dim=256, time_emb_dim=128, out_channels=1):
        super().__init__()
        self.fc1 = nn.Linear(feature_dim + time_emb_dim, 256 * 7 * 7)
        self.deconv1 = nn.ConvTranspose2d(256, 128, 4, stride=2, padding=1)
        self.gn1 = nn.GroupNorm(8, 128)
        self.deconv2 = nn.ConvTranspose2d(128, 64, 4, stride=2, padding=1)
        self.gn2 = nn.GroupNorm(8, 64)
        self.conv3 = nn.Conv2d(64, 32, 3, padding=1)
        self.gn3 = nn.GroupNorm(8, 32)
        self.conv_out = nn.Conv2d(32, out_channels, 3, padding=1)


class FlowMatchingModel(nn.Module):
    _engine = engine_property(lambda m: FmNetEngine(m))

    def __init__(self, img_channels=1, feature_dim=256, time_emb_dim=128):
        super().__init__()
        self.img_channels = img_channels
        self.feature_dim = feature_dim
        self.time_emb_dim = time_emb_dim
        self.time_embed = SinusoidalPositionEmbeddings(time_emb_dim)
        self.encoder = ImageEncoder(img_channels, feature_dim)
        self.decoder = VelocityDecoder(feature_dim, time_emb_dim, img_channels)

    def forward(self, x_t, t):
        """v_t [B,1,28,28] = model(x_t [B,1,28,28], t [B] or [1])  (reference :153-173)."""
        return self._engine.forward(x_t, t)
