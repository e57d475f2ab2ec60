"""28x28 paired sampler CLI (reference ``src/sample.py``).

Flags and defaults of the reference ``main`` (``:117-137``), its checkpoint naming
(``get_checkpoint_path('flow','x',None,'best')`` etc., ``:156-157,:182``) and both ``--model``
choices: ``unet`` (``FlowMatchingUNet``) and ``original`` (``FlowMatchingModel``).  Sampling is
``utils.flow_utils.sample_bimodal_guided`` on the HIP path.  The matplotlib grid of
``visualize_pairs`` (``:33-114``) is out of scope: samples are saved as a ``.pt`` file.
"""
import argparse
import os

import torch

from .models.flow_matching import FlowMatchingModel
from .models.ratio_estimator import RatioEstimator
from .models.unet import FlowMatchingUNet
from .utils import load_checkpoint, set_seed
from .utils.flow_utils import sample_bimodal_guided
from .utils.path_utils import get_checkpoint_path


def build_flow_models(model, device):
    """The reference's ``--model`` switch (:149-154)."""
    if model == 'unet':
        return FlowMatchingUNet().to(device), FlowMatchingUNet().to(device)
    if model == 'original':
        return FlowMatchingModel().to(device), FlowMatchingModel().to(device)
    raise ValueError(f"unknown --model {model!r} (choices: unet, original)")


def add_common_args(p, mc_default):
    p.add_argument('--transform_type', type=str, default='rotate90')
    p.add_argument('--mc_batch_size', type=int, default=mc_default)
    p.add_argument('--loss_type', type=str, default='disc')
    p.add_argument('--num_steps', type=int, default=100)
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--model', type=str, default='unet', choices=['unet', 'original'])
    p.add_argument('--seed', type=int, default=42)


def main(argv=None):
    p = argparse.ArgumentParser(description='Sample bimodal pairs (MI355X)')
    p.add_argument('--guidance_method', type=str, default='none', choices=['none', 'mc_feng'])
    p.add_argument('--guidance_strength', type=float, default=0.5)
    p.add_argument('--num_samples', type=int, default=64)
    add_common_args(p, mc_default=128)
    args = p.parse_args(argv)

    set_seed(args.seed)
    print(f"Random seed: {args.seed}")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; this sampler has no CPU path")
    device = torch.device(args.device)
    print(f"Using device: {device}")

    fm_x, fm_y = build_flow_models(args.model, device)
    path_x = get_checkpoint_path('flow', 'x', None, 'best')
    path_y = get_checkpoint_path('flow', 'y', args.transform_type, 'best')
    for path in (path_x, path_y):
        if not os.path.exists(path):
            print(f"ERROR: FM checkpoint not found: {path} (train it with the reference: python src/train_flow.py)")
            return 1
    load_checkpoint(fm_x, path_x, device)
    load_checkpoint(fm_y, path_y, device)
    print(f"  Loaded FM_x from: {path_x}\n  Loaded FM_y from: {path_y}")

    ratio = None
    if args.guidance_method != 'none':
        ratio = RatioEstimator(loss_type=args.loss_type).to(device)
        path_ratio = get_checkpoint_path('ratio', args.loss_type, args.transform_type, 'best')
        if not os.path.exists(path_ratio):
            print(f"ERROR: Ratio estimator checkpoint not found: {path_ratio}")
            return 1
        load_checkpoint(ratio, path_ratio, device)
        print(f"  Loaded ratio estimator from: {path_ratio}")

    print(f"\nSampling {args.num_samples} pairs...")
    xs, ys = sample_bimodal_guided(fm_x, fm_y, ratio, args.guidance_method, args.guidance_strength,
                                   args.num_samples, args.num_steps, device, args.mc_batch_size)
    os.makedirs('outputs', exist_ok=True)
    out = f"outputs/samples_{args.guidance_method}_gamma{args.guidance_strength}_{args.transform_type}.pt"
    torch.save({'x': xs.cpu(), 'y': ys.cpu()}, out)
    print(f"Saved samples: {out}")
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
