"""MNIST32 + SVHN paired sampler (reference ``src/sample_mnist_svhn.py``).

``sample_bimodal_guided_mnist_svhn`` keeps the reference signature
(``:39-49``) and semantics (``:68-177``); the CLI mirrors ``main`` (``:247-337``)
minus the matplotlib grid, which is out of scope: samples are saved as a
``.pt`` file instead.
"""
import argparse
import os

import torch

from .models.ratio_flexible import RatioEstimatorMNISTSVHN
from .models.unet_flexible import FlowMatchingUNetMNIST, FlowMatchingUNetSVHN
from .utils import load_checkpoint, set_seed
from .utils.flow_utils import paired_sampler


def sample_bimodal_guided_mnist_svhn(fm_mnist, fm_svhn, ratio_estimator=None, guidance_method='none',
                                     guidance_strength=0.0, num_samples=16, num_steps=100,
                                     device='cuda', mc_batch_size=64):
    """Returns ``(samples_mnist [n,1,32,32], samples_svhn [n,3,32,32])`` on `device`."""
    return paired_sampler(fm_mnist, fm_svhn, ratio_estimator, guidance_method, guidance_strength,
                          num_samples, num_steps, device, mc_batch_size, (1, 32, 32), (3, 32, 32))


def main(argv=None):
    from . import distributed  # noqa: F401  (first: it puts HSA_ENABLE_IPC_MODE_LEGACY=0 in place before any HIP call of this process)
    p = argparse.ArgumentParser(description='Sample MNIST-SVHN pairs (MI355X)')
    p.add_argument('--guidance_method', type=str, default='none', choices=['none', 'mc_feng', 'grad_log_ratio'],
                   help="'grad_log_ratio' (v + gamma * grad log r, reference README.md:159-164) is this build's extension: "
                        "the reference CLI accepts 'none' and 'mc_feng' only")
    p.add_argument('--guidance_strength', type=float, default=0.5)
    p.add_argument('--mc_batch_size', type=int, default=256)
    p.add_argument('--loss_type', type=str, default='disc')
    p.add_argument('--num_samples', type=int, default=32)
    p.add_argument('--num_steps', type=int, default=100)
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--sharded', action='store_true',
                   help='rows sharded over the ranks of a torch.distributed.run launch (one process per GPU, RCCL); '
                        'e.g. BASELINE configs[3]: --nproc-per-node 8 ... --sharded --num_samples 4096 '
                        '--guidance_method mc_feng --guidance_strength 1.0')
    args = p.parse_args(argv)

    set_seed(args.seed)  # (every rank alike: the sharded sampler slices ONE noise set)
    print(f"Random seed: {args.seed}")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible; this sampler has no CPU path")
    rank, sampler = 0, sample_bimodal_guided_mnist_svhn
    if args.sharded:
        from .distributed import init_from_env, make_sharded_sampler
        rank, world, device = init_from_env("nccl")
        sampler = make_sharded_sampler((1, 32, 32), (3, 32, 32), gather="rank0")
        print(f"rank {rank} of {world}")
    else:
        device = torch.device(args.device)
    print(f"Using device: {device}")

    fm_mnist = FlowMatchingUNetMNIST(img_size=32).to(device)
    fm_svhn = FlowMatchingUNetSVHN().to(device)
    for model, path, hint in ((fm_mnist, 'checkpoints/flow_mnist32_best.pth', 'train_flow_mnist32.py'),
                              (fm_svhn, 'checkpoints/flow_svhn_best.pth', 'train_flow_svhn.py')):
        if not os.path.exists(path):
            print(f"ERROR: checkpoint not found: {path}")
            print(f"Please train first with the reference: python src/{hint}")
            return 1
        load_checkpoint(model, path, device)
        print(f"  Loaded {path}")

    ratio = None
    if args.guidance_method != 'none':
        ratio = RatioEstimatorMNISTSVHN(loss_type=args.loss_type).to(device)
        path = f'checkpoints/ratio_{args.loss_type}_mnist_svhn_best.pth'
        if not os.path.exists(path):
            print(f"ERROR: Ratio estimator not found: {path}")
            return 1
        ratio.load_state_dict(torch.load(path, map_location=device))
        print(f"  Loaded ratio estimator from: {path}")

    print(f"\nSampling {args.num_samples} pairs...")
    xs, ys = sampler(fm_mnist, fm_svhn, ratio, args.guidance_method, args.guidance_strength, args.num_samples,
                     args.num_steps, device, args.mc_batch_size)
    if args.sharded:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        if rank != 0:
            return 0
    os.makedirs('outputs/mnist_svhn', exist_ok=True)
    out = f"outputs/mnist_svhn/samples_{args.guidance_method}_gamma{args.guidance_strength}.pt"
    torch.save({'mnist': xs.cpu(), 'svhn': ys.cpu()}, out)
    print(f"Saved samples: {out}")
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
