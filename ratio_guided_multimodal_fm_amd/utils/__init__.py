"""Seeding and checkpoint ingestion (reference ``src/utils/__init__.py``)."""
import random

import numpy as np
import torch


def set_seed(seed: int = 42):
    """Seed python / numpy / torch generators (reference :7-22).

    The samplers draw their noise from the global torch generator of the
    target device in the reference order x0, y0, mc_x0, mc_y0.
    """
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


def load_checkpoint(model, path, device='cpu'):
    """Load either checkpoint format the reference writes (reference :25-51).

    ``{'model_state_dict': ..., 'epoch': ..., 'best_loss': ...}`` dicts
    (train_flow_mnist32.py / train_flow_svhn.py) or a raw ``state_dict``
    (train_flow.py, ratio and classifier trainers).  Returns
    ``{'epoch', 'best_loss'}`` for the dict format, ``{}`` otherwise.
    """
    ckpt = torch.load(path, map_location=device)
    if isinstance(ckpt, dict) and 'model_state_dict' in ckpt:
        model.load_state_dict(ckpt['model_state_dict'])
        return {'epoch': ckpt.get('epoch', 0), 'best_loss': ckpt.get('best_loss', float('inf'))}
    model.load_state_dict(ckpt)
    return {}
