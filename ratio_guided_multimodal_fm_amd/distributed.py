"""Multi-GPU sampler: one process per GPU, rows sharded, RCCL over xGMI.

New design (the reference has no multi-device code).  Rows of (x_t, y_t) never
interact -- per-sample GroupNorm, eval-mode BatchNorm, and guidance weights that
depend only on the row and the SHARED Monte-Carlo set
(reference src/sample_mnist_svhn.py:130-167) -- so:

  1. the MC pre-phase (sample_mnist_svhn.py:85-104) is sharded too: rank r
     integrates rows [r*N/W, (r+1)*N/W) of mc_x0 / mc_y0 and evaluates the
     ratio estimator on them;
  2. ONE all_gather of (mc_x1, mc_y1, mc_ratios) makes the MC set identical
     on every rank (N*(Dx+Dy+1)*4 bytes, 4.2 MB at N=256: latency-bound);
  3. each rank runs the guided Euler loop on its rows of (x0, y0);
  4. ONE gather (or all_gather) of the outputs.

No collective sits inside the Euler loop.  `backend` abstracts the three compute
calls so that the sharding logic can be exercised with gloo on CPU ranks (the
tests inject the CPU oracle there); the default backend is the HIP library.
"""
import os

# ROCr reads its flags at hsa_init, i.e. at the first HIP call of the process (torch.cuda.is_available() already is
# one): the dmabuf-only IPC setting these hosts need must be in the environment BEFORE that, so it is set when this
# module is imported -- the launchers import it first thing in main(), ahead of set_seed / any torch.cuda call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n rows; the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class HipBackend:
    """Compute calls routed to librgfm_hip.so (the product path)."""

    def sample_single(self, model, x, num_steps):
        from . import _engine
        return _engine.sample_single(model, x, num_steps)

    def sample_two(self, fm_x, x, fm_y, y, num_steps):
        from . import _engine
        return _engine.sample_two_streams(fm_x, x, fm_y, y, num_steps)

    def ratios(self, ratio_estimator, mc_x1, mc_y1):
        return ratio_estimator._engine.eval(mc_x1, mc_y1, "ratio")

    def sample_pair(self, fm_x, fm_y, x, y, mc_x1, mc_y1, mc_ratios, num_steps, gamma):
        from . import _engine
        return _engine.sample_pair(fm_x, fm_y, x, y, mc_x1, mc_y1, mc_ratios, num_steps, gamma)

    def sample_pair_grad(self, fm_x, fm_y, ratio_estimator, x, y, num_steps, gamma):
        from . import _engine
        return _engine.sample_pair_grad(fm_x, fm_y, ratio_estimator, x, y, num_steps, gamma)


def _via_host(t, group):
    """gloo group + device tensor (the two-ranks-on-one-GPU test, or a CPU-only fabric): the collective runs on a
    host copy.  RCCL groups never take this path."""
    return t.is_cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"


def _all_gather_rows(t, counts, group):
    """all_gather of row-sharded tensors with possibly unequal row counts."""
    world = len(counts)
    if not dist.is_initialized():
        return t
    if _via_host(t, group):
        return _all_gather_rows(t.cpu(), counts, group).to(t.device)
    mx = max(counts)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], 0)
    t = t.contiguous()
    if all(c == mx for c in counts):
        out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def sharded_paired_sampler(fm_x, fm_y, ratio_estimator, guidance_method, guidance_strength, num_steps,
                           noise, device, backend=None, group=None, gather="rank0", keep=None):
    """Sharded equivalent of ``paired_sampler`` on explicit host noise.

    noise = (x0, y0, mc_x0, mc_y0): the FULL tensors, identical on every rank
    (every rank regenerates them from the seed; each slices its own rows).
    Returns (x, y) with all rows: on every rank for gather="all", on rank 0
    only (None, None elsewhere) for gather="rank0".
    keep: optional dict that receives references to the gathered MC set the guided loop ran on
    (``mc_x1``, ``mc_y1``, ``mc_ratios``) -- bench.py's parity gate follows rows of the timed call from it.
    """
    backend = backend or HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    x0, y0, mc_x0, mc_y0 = noise
    B = x0.shape[0]
    if guidance_method not in ('none', 'mc_feng', 'grad_log_ratio'):
        raise ValueError(f"Unknown guidance_method: {guidance_method}")
    guided = guidance_method == 'mc_feng' and ratio_estimator is not None
    # gradient guidance needs no MC set: every rank integrates its rows with the estimator's gradient
    grad_guided = guidance_method == 'grad_log_ratio' and ratio_estimator is not None
    if grad_guided and not hasattr(backend, "sample_pair_grad"):
        raise ValueError("this backend has no gradient log-ratio guidance (sample_pair_grad)")
    for m in (fm_x, fm_y, ratio_estimator):
        if m is not None:
            m.eval()

    mc_x1 = mc_y1 = mc_r = None
    if guided:
        N = mc_x0.shape[0]
        lo, hi = shard_bounds(N, world, rank)
        sx = mc_x0[lo:hi].to(device, copy=True).contiguous()  # never clobber the caller's noise
        sy = mc_y0[lo:hi].to(device, copy=True).contiguous()
        if hi > lo:
            if hasattr(backend, "sample_two"):
                backend.sample_two(fm_x, sx, fm_y, sy, num_steps)
            else:
                backend.sample_single(fm_x, sx, num_steps)
                backend.sample_single(fm_y, sy, num_steps)
            sr = backend.ratios(ratio_estimator, sx, sy)
        else:
            sr = torch.empty(0, device=device)
        counts = [shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0] for r in range(world)]
        mc_x1 = _all_gather_rows(sx, counts, group)
        mc_y1 = _all_gather_rows(sy, counts, group)
        mc_r = _all_gather_rows(sr, counts, group)
        if keep is not None:
            keep.update(mc_x1=mc_x1, mc_y1=mc_y1, mc_ratios=mc_r)

    lo, hi = shard_bounds(B, world, rank)
    x = x0[lo:hi].to(device, copy=True).contiguous()
    y = y0[lo:hi].to(device, copy=True).contiguous()
    if hi > lo:
        if grad_guided:
            backend.sample_pair_grad(fm_x, fm_y, ratio_estimator, x, y, num_steps, guidance_strength)
        else:
            backend.sample_pair(fm_x, fm_y, x, y, mc_x1, mc_y1, mc_r, num_steps, guidance_strength)
    if not dist.is_initialized():
        return x, y
    counts = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    if gather == "all":
        return _all_gather_rows(x, counts, group), _all_gather_rows(y, counts, group)
    return _gather_rows(x, counts, group, rank), _gather_rows(y, counts, group, rank)


def _gather_rows(t, counts, group, rank, dst=0):
    if _via_host(t, group):
        out = _gather_rows(t.cpu(), counts, group, rank, dst)
        return None if out is None else out.to(t.device)
    world = len(counts)
    mx = max(counts)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], 0)
    t = t.contiguous()
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def init_from_env(backend="nccl"):
    """Process-group setup of a `torch.distributed.run` launch (one process per GPU):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, RCCL over xGMI.  Returns (rank, world, device).
    A plain `python` launch (no WORLD_SIZE) is rank 0 of 1 with no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on these hosts
    if backend == "nccl":
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def make_sharded_sampler(shape_x, shape_y, backend=None, group=None, gather="all"):
    """A drop-in for ``sample_bimodal_guided*`` (same positional signature) that shards the rows over the ranks.

    The reference draws x0, y0[, mc_x0, mc_y0] from the target device's global generator
    (sample_mnist_svhn.py:74,75,89,98).  Ranks must integrate rows of ONE noise set, so here every rank draws
    the FULL tensors from its torch CPU generator in that order -- seed all ranks alike (``set_seed``), as the
    launchers do -- and ``sharded_paired_sampler`` slices its rows: consecutive calls consume one generator
    stream exactly like the reference's sweep (evaluate_mnist_svhn.py:80: one seeding, no re-seed)."""
    def sampler(fm_x, fm_y, ratio_estimator=None, guidance_method='none', guidance_strength=0.0, num_samples=16,
                num_steps=100, device='cuda', mc_batch_size=64):
        guided = guidance_method == 'mc_feng' and ratio_estimator is not None
        x0 = torch.randn(num_samples, *shape_x)
        y0 = torch.randn(num_samples, *shape_y)
        mx = torch.randn(mc_batch_size, *shape_x) if guided else None
        my = torch.randn(mc_batch_size, *shape_y) if guided else None
        return sharded_paired_sampler(fm_x, fm_y, ratio_estimator, guidance_method, guidance_strength, num_steps,
                                      (x0, y0, mx, my), torch.device(device), backend=backend, group=group, gather=gather)
    return sampler
