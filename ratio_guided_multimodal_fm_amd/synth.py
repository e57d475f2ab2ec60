"""Deterministic synthetic parameters for parity tests and benchmarks.

No trained checkpoint ships with the reference (its ``checkpoints/`` holds a
``.gitkeep`` only), so every parity and bench run uses parameters produced by
this recipe.  It depends only on ``(seed, position in state_dict, shape)`` and
the torch CPU generator, so the reference modules (in
``tests/golden/make_golden.py``) and this package's modules receive bitwise
identical values without any weight file travelling.
"""
import math

import torch


def synth_state_dict(module, seed=0):
    """New state_dict for `module` (any nn.Module) filled by the recipe.

    conv / linear weights ~ N(0, 1/fan_in); biases ~ 0.05 N(0,1); norm scales
    ~ 1 + 0.1 N(0,1); BatchNorm running_mean ~ 0.1 N(0,1), running_var ~
    U(0.5, 1.5).  The U-Net output conv is NOT zeroed (the reference's zero
    init, unet_flexible.py:200-201, would make every velocity 0).
    """
    out = {}
    for i, (k, v) in enumerate(module.state_dict().items()):
        g = torch.Generator().manual_seed(int(seed) * 1000003 + i)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[k] = torch.zeros_like(v, device='cpu')
        elif leaf == 'running_mean':
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif leaf == 'running_var':
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif v.dim() == 1 and leaf == 'weight':
            out[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif leaf == 'bias':
            out[k] = 0.05 * torch.randn(v.shape, generator=g)
        else:
            fan_in = max(1, v[0].numel())
            out[k] = torch.randn(v.shape, generator=g) / math.sqrt(fan_in)
    return out


def load_synth(module, seed=0):
    module.load_state_dict(synth_state_dict(module, seed))
    return module


def paired_noise(seed, batch, n_mc, shape_x, shape_y):
    """(x0, y0, mc_x0, mc_y0) from the torch CPU generator, in the reference's
    draw order (sample_mnist_svhn.py:74,75,89,98)."""
    g = torch.Generator().manual_seed(int(seed))
    x0 = torch.randn(batch, *shape_x, generator=g)
    y0 = torch.randn(batch, *shape_y, generator=g)
    mx = torch.randn(n_mc, *shape_x, generator=g) if n_mc else None
    my = torch.randn(n_mc, *shape_y, generator=g) if n_mc else None
    return x0, y0, mx, my
