"""Checkpoint naming (reference ``src/utils/path_utils.py:7-32``)."""
import os


def get_checkpoint_path(model_type, *args):
    """``get_checkpoint_path('flow', 'x', None, 'best') -> 'checkpoints/flow_x_best.pth'``.

    ``None`` identifiers are dropped; the ``checkpoints`` directory is created,
    as the reference does.
    """
    base_dir = 'checkpoints'
    os.makedirs(base_dir, exist_ok=True)
    parts = [str(a) for a in args if a is not None]
    return os.path.join(base_dir, f"{model_type}_{'_'.join(parts)}.pth")
