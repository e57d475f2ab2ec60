"""MNIST-SVHN density-ratio estimator, host side.

API mirror of ``RatioEstimatorMNISTSVHN`` (reference
``src/models/ratio_flexible.py:305-385``; encoders ``:185-302``): parameter
containers with the reference's ``state_dict`` keys; ``forward`` /
``log_ratio`` run in the HIP library (eval-mode semantics: BatchNorm uses
running statistics, Dropout is the identity).
"""
import torch.nn as nn

from .._engine import RatioEngine, engine_property


class _BNEncoderParams(nn.Module):
    def __init__(self, plan, feature_dim):
        super().__init__()
        for name, cin, cout in plan:
            setattr(self, "conv" + name, nn.Conv2d(cin, cout, 3, padding=1))
            setattr(self, "bn" + name, nn.BatchNorm2d(cout))
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(plan[-1][2], feature_dim)


class MNISTEncoder(_BNEncoderParams):
    """1x32x32 -> feature_dim (reference :185-232)."""

    def __init__(self, feature_dim=256):
        super().__init__([("1", 1, 32), ("2", 32, 64), ("3", 64, 128), ("4", 128, 128)], feature_dim)


class SVHNEncoder(_BNEncoderParams):
    """3x32x32 -> feature_dim (reference :235-302)."""

    def __init__(self, feature_dim=256):
        super().__init__([("1a", 3, 64), ("1b", 64, 64), ("2a", 64, 128), ("2b", 128, 128),
                          ("3a", 128, 256), ("3b", 256, 256), ("4a", 256, 256), ("4b", 256, 256)],
                         feature_dim)


class RatioEstimatorMNISTSVHN(nn.Module):
    _engine = engine_property(lambda m: RatioEngine(m, kind="mnist_svhn"))

    def __init__(self, feature_dim=256, hidden_dim=512, loss_type='disc'):
        super().__init__()
        self.feature_dim = feature_dim
        self.hidden_dim = hidden_dim
        self.loss_type = loss_type
        self.encoder_mnist = MNISTEncoder(feature_dim)
        self.encoder_svhn = SVHNEncoder(feature_dim)
        h = hidden_dim
        self.score_net = nn.Sequential(
            nn.Linear(feature_dim * 2, h), nn.LayerNorm(h), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(h, h), nn.LayerNorm(h), nn.SiLU(), nn.Dropout(0.1),
            nn.Linear(h, h // 2), nn.LayerNorm(h // 2), nn.SiLU(),
            nn.Linear(h // 2, 1))

    def forward(self, x, y):
        """Scores T(x, y): x [B,1,32,32], y [B,3,32,32] -> [B]."""
        return self._engine.eval(x, y, "score")

    def log_ratio(self, x, y):
        """log r(x, y); raises ValueError for an unknown loss_type (reference :384-385)."""
        if self.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {self.loss_type}")
        return self._engine.eval(x, y, "log_ratio")

    def grad_log_ratio(self, x, y):
        """(d log_ratio/dx, d log_ratio/dy): what ``torch.autograd.grad(self.log_ratio(x, y).sum(), (x, y))``
        returns for the reference module in eval mode -- the quantity of the reference README's "Gradient
        Log-Ratio" guidance (``README.md:159-164``), which the reference itself never computes.  Hand-written
        reverse pass on the device (``csrc/ratio_grad.hip``); the parameters themselves get no gradient."""
        if self.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {self.loss_type}")
        gx, gy, _ = self._engine.grad_log_ratio(x, y)
        return gx, gy
