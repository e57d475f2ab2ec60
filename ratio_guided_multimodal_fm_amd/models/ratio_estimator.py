"""28x28 same-modality density-ratio estimator, host side.

API mirror of ``RatioEstimator`` (reference ``src/models/ratio_estimator.py:96-191``,
GroupNorm encoder ``:34-93``).  Parameter containers only; evaluation runs in
the HIP library.
"""
import torch.nn as nn

from .._engine import RatioEngine, engine_property


class ImageEncoder(nn.Module):
    def __init__(self, in_channels=1, feature_dim=256):
        super().__init__()
        chans = [in_channels, 32, 64, 128, 128]
        for i in range(4):
            setattr(self, f"conv{i + 1}", nn.Conv2d(chans[i], chans[i + 1], 3, padding=1))
            setattr(self, f"gn{i + 1}", nn.GroupNorm(8, chans[i + 1]))
            if i < 3:
                setattr(self, f"pool{i + 1}", nn.MaxPool2d(2))
        self.pool_final = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(128, feature_dim)


class RatioEstimator(nn.Module):
    _engine = engine_property(lambda m: RatioEngine(m, kind="mnist28"))

    def __init__(self, feature_dim=256, hidden_dim=512, loss_type='disc'):
        super().__init__()
        self.feature_dim = feature_dim
        self.hidden_dim = hidden_dim
        self.loss_type = loss_type
        self.encoder_x = ImageEncoder(1, feature_dim)
        self.encoder_y = ImageEncoder(1, feature_dim)
        h = hidden_dim
        self.score_net = nn.Sequential(
            nn.Linear(feature_dim * 2, h), nn.LayerNorm(h), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(h, h // 2), nn.LayerNorm(h // 2), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(h // 2, 1))

    def forward(self, x, y):
        return self._engine.eval(x, y, "score")

    def log_ratio(self, x, y):
        if self.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {self.loss_type}")
        return self._engine.eval(x, y, "log_ratio")

    def grad_log_ratio(self, x, y):
        """(d log_ratio/dx, d log_ratio/dy): what ``torch.autograd.grad(self.log_ratio(x, y).sum(), (x, y))`` returns for
        the reference module in eval mode (reference ``ratio_estimator.py:137-191``; the quantity of the README's
        "Gradient Log-Ratio" guidance, ``README.md:159-164``).  Hand-written reverse pass on the device through the
        GroupNorm encoders (``csrc/ratio_grad.hip``: ``gn_bwd_kernel``); the parameters themselves get no gradient."""
        if self.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {self.loss_type}")
        gx, gy, _ = self._engine.grad_log_ratio(x, y)
        return gx, gy
