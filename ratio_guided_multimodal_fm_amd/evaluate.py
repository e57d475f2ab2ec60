"""Coherence evaluation of generated 28x28 pairs (reference ``src/evaluate.py``).

``get_inverse_transform`` (``:31-54``), ``evaluate_coherence`` (``:57-96``) and the sweep of ``main``
(``:99-240``) with its flags, skip rule and JSON schema
(``method, guidance_strength, transform_type, coherence_acc, num_samples``).

The reference un-rotates y with ``torchvision.transforms.functional.rotate`` (nearest, about the
image centre); for the multiples of 90 degrees it uses on a square image that is the exact pixel
permutation ``torch.rot90`` (counter-clockwise for positive angles), which is what is used here --
torchvision is not a dependency of this package.  Sampling runs on the HIP path, the classifier once
per configuration in PyTorch-ROCm.
"""
import argparse
import json
import os

import torch

from .models.classifier import MNISTClassifier
from .models.ratio_estimator import RatioEstimator
from .sample import add_common_args, build_flow_models
from .utils import load_checkpoint, set_seed
from .utils.flow_utils import sample_bimodal_guided
from .utils.path_utils import get_checkpoint_path


def get_inverse_transform(transform_type):
    """Function undoing the y-modality transform; unknown names are the identity (reference :53-54)."""
    table = {
        'rotate90': lambda img: torch.rot90(img, 1, (-2, -1)),    # TF.rotate(img, 90)
        'rotate180': lambda img: torch.rot90(img, 2, (-2, -1)),   # TF.rotate(img, 180)
        'rotate270': lambda img: torch.rot90(img, -1, (-2, -1)),  # TF.rotate(img, -90)
        'invert': lambda img: -img,
        'flip_h': lambda img: torch.flip(img, (-1,)),
        'flip_v': lambda img: torch.flip(img, (-2,)),
    }
    return table.get(transform_type, lambda img: img)


def evaluate_coherence(samples_x, samples_y, classifier, device, transform_type='rotate90'):
    """P(argmax clf(x) == argmax clf(inverse_transform(y))); returns the reference's dict."""
    classifier.eval()
    y_inv = get_inverse_transform(transform_type)(samples_y)
    with torch.no_grad():
        pred_x = classifier(samples_x.to(device)).argmax(dim=1)
        pred_y = classifier(y_inv.to(device)).argmax(dim=1)
    acc = (pred_x == pred_y).float().mean().item() if len(samples_x) else float('nan')
    return {'coherence_acc': float(acc), 'num_samples': len(samples_x)}


def run_sweep(fm_x, fm_y, make_ratio, classifier, methods, strengths, num_samples, num_steps, device,
              mc_batch_size, transform_type, sampler=sample_bimodal_guided):
    """The reference's nested loop (:165-222). `make_ratio()` returns a fresh ratio estimator or None."""
    results = []
    for method in methods:
        for strength in strengths:
            if method == 'none' and strength > 0:
                continue
            ratio = make_ratio() if method != 'none' else None
            if method != 'none' and ratio is None:
                continue
            xs, ys = sampler(fm_x, fm_y, ratio, method, strength, num_samples, num_steps, device, mc_batch_size)
            metrics = evaluate_coherence(xs, ys, classifier, device, transform_type)
            results.append({'method': method, 'guidance_strength': strength, 'transform_type': transform_type,
                            **metrics})
            print(f"  method={method} gamma={strength} -> coherence accuracy {metrics['coherence_acc']:.3f}")
    return results


def main(argv=None):
    p = argparse.ArgumentParser(description='Evaluate guided sampling (MI355X)')
    p.add_argument('--guidance_methods', nargs='+', default=['none', 'mc_feng'])
    p.add_argument('--guidance_strengths', nargs='+', type=float, default=[0.0, 0.5, 1.0])
    p.add_argument('--num_samples', type=int, default=500)
    add_common_args(p, mc_default=256)
    args = p.parse_args(argv)

    set_seed(args.seed)
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; the sampler has no CPU path")
    device = torch.device(args.device)

    clf_path = 'checkpoints/mnist_classifier.pth'
    path_x = get_checkpoint_path('flow', 'x', None, 'best')
    path_y = get_checkpoint_path('flow', 'y', args.transform_type, 'best')
    for path in (clf_path, path_x, path_y):
        if not os.path.exists(path):
            print(f"ERROR: checkpoint not found: {path} (train it with the reference scripts)")
            return 1
    classifier = MNISTClassifier().to(device)
    classifier.load_state_dict(torch.load(clf_path, map_location=device))
    fm_x, fm_y = build_flow_models(args.model, device)
    load_checkpoint(fm_x, path_x, device)
    load_checkpoint(fm_y, path_y, device)

    def make_ratio():
        path = get_checkpoint_path('ratio', args.loss_type, args.transform_type, 'best')
        if not os.path.exists(path):
            print(f"ERROR: Ratio estimator not found: {path}")
            return None
        r = RatioEstimator(loss_type=args.loss_type).to(device)
        load_checkpoint(r, path, device)
        return r

    results = run_sweep(fm_x, fm_y, make_ratio, classifier, args.guidance_methods, args.guidance_strengths,
                        args.num_samples, args.num_steps, device, args.mc_batch_size, args.transform_type)
    os.makedirs('outputs', exist_ok=True)
    out = 'outputs/evaluation_results.json'
    with open(out, 'w') as f:
        json.dump(results, f, indent=2)
    print(f"Results saved to: {out}")
    for r in results:
        print(f"  {r['method']:20s} gamma={r['guidance_strength']:.1f} -> coherence={r['coherence_acc']:.3f}")
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
