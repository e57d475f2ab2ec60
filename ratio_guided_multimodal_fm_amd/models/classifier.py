"""``MNISTClassifier`` of the 28x28 evaluation harness (reference ``src/models/classifier.py:9-52``).

Runs once on the final samples, so it is an ordinary PyTorch module executed by PyTorch-ROCm (same
policy as ``svhn_classifier.py``); ``state_dict`` keys/shapes match the reference's checkpoint
``checkpoints/mnist_classifier.pth``.
"""
import torch.nn as nn
import torch.nn.functional as F


class MNISTClassifier(nn.Module):
    """1x28x28 -> 10 logits: conv-ReLU-pool, conv-ReLU-pool, Linear(3136,128)-ReLU-Dropout-Linear(128,10)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 32, 3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, 3, padding=1)
        self.fc1 = nn.Linear(64 * 7 * 7, 128)
        self.fc2 = nn.Linear(128, 10)
        self.dropout = nn.Dropout(0.25)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.conv1(x)), 2)
        x = F.max_pool2d(F.relu(self.conv2(x)), 2)
        x = F.relu(self.fc1(x.flatten(1)))
        return self.fc2(self.dropout(x))
