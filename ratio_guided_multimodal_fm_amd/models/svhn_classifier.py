"""Evaluation classifiers of the MNIST-SVHN experiment (reference
``src/models/svhn_classifier.py``: ``SVHNClassifier`` ``:11-71``, ``MNISTClassifier32`` ``:74-116``).

They run ONCE on the final samples (< 0.01 % of a sampling call's work), so they are
ordinary PyTorch modules executed by PyTorch-ROCm: not part of the accelerated hot path.  Same
``state_dict`` keys and shapes as the reference, so its trained classifier checkpoints load.
"""
import torch.nn as nn
import torch.nn.functional as F


class SVHNClassifier(nn.Module):
    """3x32x32 -> 10 logits: 4 x (conv3x3 + BatchNorm + ReLU), max-pool after the first two."""

    def __init__(self):
        super().__init__()
        chans = (3, 32, 64, 128, 128)
        for i in range(4):
            setattr(self, f"conv{i + 1}", nn.Conv2d(chans[i], chans[i + 1], 3, padding=1))
            setattr(self, f"bn{i + 1}", nn.BatchNorm2d(chans[i + 1]))
        self.fc1 = nn.Linear(128 * 8 * 8, 256)
        self.fc2 = nn.Linear(256, 10)
        self.dropout = nn.Dropout(0.3)

    def forward(self, x):
        for i in (1, 2, 3, 4):
            x = F.relu(getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(x)))
            if i <= 2:
                x = F.max_pool2d(x, 2)
        x = F.relu(self.fc1(x.flatten(1)))
        return self.fc2(self.dropout(x))


class MNISTClassifier32(nn.Module):
    """1x32x32 -> 10 logits: 3 x (conv3x3 + ReLU), max-pool after the first two."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 32, 3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1)
        self.conv3 = nn.Conv2d(64, 64, 3, padding=1)
        self.fc1 = nn.Linear(64 * 8 * 8, 128)
        self.fc2 = nn.Linear(128, 10)
        self.dropout = nn.Dropout(0.25)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 2)
        x = F.relu(self.conv3(x))
        x = F.relu(self.fc1(x.flatten(1)))
        return self.fc2(self.dropout(x))
