"""Coherence evaluation of generated MNIST-SVHN pairs (reference ``src/evaluate_mnist_svhn.py``).

``evaluate_coherence`` (``:28-57``) and the method x strength sweep of ``main`` (``:60-199``) with the
reference's flags, skip rule (``none`` with gamma > 0), single ``set_seed`` before the sweep (so each
configuration's noise depends on sweep order, as in the reference) and JSON schema
(``method, guidance_strength, experiment, coherence_acc, num_samples``).  Sampling runs on the HIP
path; the two classifiers run once per configuration in PyTorch-ROCm.

Multi-GPU (new; the reference is single-device): launched under ``torch.distributed.run`` with ``--sharded``,
e.g. BASELINE configs[4]

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m ratio_guided_multimodal_fm_amd.evaluate_mnist_svhn --sharded --num_samples 8192 --num_steps 200 \
        --guidance_strengths 0 0.5 1 2 5

every rank loads the checkpoints, integrates its rows of each configuration (distributed.py: sharded MC
pre-phase, one RCCL all_gather of the MC set, one all_gather of the outputs) and rank 0 classifies and writes
the JSON.
"""
import argparse
import json
import os

import torch

from .models.ratio_flexible import RatioEstimatorMNISTSVHN
from .models.svhn_classifier import MNISTClassifier32, SVHNClassifier
from .models.unet_flexible import FlowMatchingUNetMNIST, FlowMatchingUNetSVHN
from .sample_mnist_svhn import sample_bimodal_guided_mnist_svhn
from .utils import load_checkpoint, set_seed


def evaluate_coherence(samples_mnist, samples_svhn, mnist_classifier, svhn_classifier, device):
    """P(argmax clf_mnist(x) == argmax clf_svhn(y)) over the pairs; returns the reference's dict."""
    mnist_classifier.eval()
    svhn_classifier.eval()
    with torch.no_grad():
        pred_m = mnist_classifier(samples_mnist.to(device)).argmax(dim=1)
        pred_s = svhn_classifier(samples_svhn.to(device)).argmax(dim=1)
    acc = (pred_m == pred_s).float().mean().item() if len(samples_mnist) else float('nan')
    return {'coherence_acc': float(acc), 'num_samples': len(samples_mnist)}


def run_sweep(fm_mnist, fm_svhn, make_ratio, mnist_classifier, svhn_classifier, methods, strengths,
              num_samples, num_steps, device, mc_batch_size, sampler=sample_bimodal_guided_mnist_svhn):
    """The reference's nested loop (:130-183). `make_ratio()` returns a fresh ratio estimator or None."""
    results = []
    for method in methods:
        for strength in strengths:
            if method == 'none' and strength > 0:
                continue
            ratio = make_ratio() if method != 'none' else None
            if method != 'none' and ratio is None:
                continue
            xs, ys = sampler(fm_mnist, fm_svhn, ratio, method, strength, num_samples, num_steps, device,
                             mc_batch_size)
            metrics = evaluate_coherence(xs, ys, mnist_classifier, svhn_classifier, device)
            results.append({'method': method, 'guidance_strength': strength, 'experiment': 'mnist_svhn',
                            **metrics})
            print(f"  method={method} gamma={strength} -> coherence accuracy {metrics['coherence_acc']:.3f}")
    return results


def main(argv=None):
    from . import distributed  # noqa: F401  (first: it puts HSA_ENABLE_IPC_MODE_LEGACY=0 in place before any HIP call of this process)
    p = argparse.ArgumentParser(description='Evaluate MNIST-SVHN guided sampling (MI355X)')
    p.add_argument('--guidance_methods', nargs='+', default=['none', 'mc_feng'])
    p.add_argument('--guidance_strengths', nargs='+', type=float, default=[0.0, 0.5, 1.0])
    p.add_argument('--mc_batch_size', type=int, default=256)
    p.add_argument('--loss_type', type=str, default='disc')
    p.add_argument('--num_samples', type=int, default=500)
    p.add_argument('--num_steps', type=int, default=100)
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--sharded', action='store_true',
                   help='rows sharded over the ranks of a torch.distributed.run launch (one process per GPU, RCCL)')
    args = p.parse_args(argv)

    set_seed(args.seed)  # (every rank alike: the sharded sampler slices ONE noise set)
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; the sampler has no CPU path")
    rank, sampler = 0, sample_bimodal_guided_mnist_svhn
    if args.sharded:
        from .distributed import init_from_env, make_sharded_sampler
        rank, world, device = init_from_env("nccl")
        sampler = make_sharded_sampler((1, 32, 32), (3, 32, 32), gather="all")
        print(f"rank {rank} of {world} on {device}")
    else:
        device = torch.device(args.device)

    paths = {'mnist_clf': 'checkpoints/mnist32_classifier.pth', 'svhn_clf': 'checkpoints/svhn_classifier.pth',
             'fm_mnist': 'checkpoints/flow_mnist32_best.pth', 'fm_svhn': 'checkpoints/flow_svhn_best.pth'}
    for k, v in paths.items():
        if not os.path.exists(v):
            print(f"ERROR: checkpoint not found: {v} (train it with the reference scripts)")
            return 1
    mnist_clf = MNISTClassifier32().to(device)
    svhn_clf = SVHNClassifier().to(device)
    mnist_clf.load_state_dict(torch.load(paths['mnist_clf'], map_location=device))
    svhn_clf.load_state_dict(torch.load(paths['svhn_clf'], map_location=device))
    fm_mnist = FlowMatchingUNetMNIST(img_size=32).to(device)
    fm_svhn = FlowMatchingUNetSVHN().to(device)
    load_checkpoint(fm_mnist, paths['fm_mnist'], device)
    load_checkpoint(fm_svhn, paths['fm_svhn'], device)

    def make_ratio():
        path = f'checkpoints/ratio_{args.loss_type}_mnist_svhn_best.pth'
        if not os.path.exists(path):
            print(f"ERROR: Ratio estimator not found: {path}")
            return None
        r = RatioEstimatorMNISTSVHN(loss_type=args.loss_type).to(device)
        r.load_state_dict(torch.load(path, map_location=device))
        return r

    results = run_sweep(fm_mnist, fm_svhn, make_ratio, mnist_clf, svhn_clf, args.guidance_methods,
                        args.guidance_strengths, args.num_samples, args.num_steps, device, args.mc_batch_size,
                        sampler=sampler)
    if args.sharded:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        if rank != 0:
            return 0
    os.makedirs('outputs/mnist_svhn', exist_ok=True)
    out = 'outputs/mnist_svhn/evaluation_results.json'
    with open(out, 'w') as f:
        json.dump(results, f, indent=2)
    print(f"Results saved to: {out}")
    for r in results:
        print(f"  {r['method']:20s} gamma={r['guidance_strength']:.1f} -> coherence={r['coherence_acc']:.3f}")
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
