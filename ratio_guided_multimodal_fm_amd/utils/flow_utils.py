"""Samplers of the 28x28 experiments (reference ``src/utils/flow_utils.py``).

``CFMSchedule.sample`` (``:69-100``) and ``sample_bimodal_guided``
(``:178-375``) keep the reference signatures and return values; the Euler
loops, the U-Net evaluations, the ratio estimator and the MC guidance all run
inside librgfm_hip.so (one C-ABI call per phase).
"""
import torch

from .. import _engine


def _device(device):
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError(
            f"device '{device}' requested: the MI355X sampler has no CPU path; pass a HIP device "
            "('cuda' / 'cuda:N').")
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


class CFMSchedule:
    """Rectified-flow schedule; only the sampler is on the accelerated path."""

    def __init__(self, sigma=0.0):
        self.sigma = sigma

    def compute_mu_t(self, x_0, x_1, t):
        t = t.view(-1, 1, 1, 1)
        return (1 - t) * x_0 + t * x_1

    def compute_sigma_t(self, t):
        return self.sigma

    def sample(self, model, num_samples, num_steps=100, device='cuda'):
        """x0 ~ N(0, I) [n,1,28,28]; num_steps explicit Euler steps (reference :69-100)."""
        model.eval()
        dev = _device(device)
        x_t = torch.randn(num_samples, 1, 28, 28, device=dev)
        return _engine.sample_single(model, x_t, num_steps)


def paired_sampler(fm_x, fm_y, ratio_estimator, guidance_method, guidance_strength, num_samples,
                   num_steps, device, mc_batch_size, shape_x, shape_y, noise=None, verbose=True):
    """Shared body of both paired samplers.

    `noise` = (x0, y0, mc_x0, mc_y0) overrides the generator draws (parity
    tests upload CPU-generated noise); tensors are copied, never modified.
    """
    fm_x.eval()
    fm_y.eval()
    if ratio_estimator is not None:
        ratio_estimator.eval()
    dev = _device(device)
    guided = guidance_method == 'mc_feng' and ratio_estimator is not None
    grad_guided = guidance_method == 'grad_log_ratio' and ratio_estimator is not None

    if noise is None:
        x_t = torch.randn(num_samples, *shape_x, device=dev)
        y_t = torch.randn(num_samples, *shape_y, device=dev)
    else:
        x_t, y_t = noise[0].to(dev, copy=True).contiguous(), noise[1].to(dev, copy=True).contiguous()

    mc_x1 = mc_y1 = mc_ratios = None
    if guided:
        if verbose:
            print(f"  Generating {mc_batch_size} independent MC samples from flows...")
        # noise in the reference's draw order (x0, y0, mc_x0, mc_y0), then the two independent
        # pre-phase integrations run concurrently on two HIP streams (the N_mc-row launches are too
        # small to fill 256 CUs one net at a time)
        if noise is None:
            mc_x1 = torch.randn(mc_batch_size, *shape_x, device=dev)
            mc_y1 = torch.randn(mc_batch_size, *shape_y, device=dev)
        else:
            mc_x1 = noise[2].to(dev, copy=True).contiguous()
            mc_y1 = noise[3].to(dev, copy=True).contiguous()
        _engine.sample_two_streams(fm_x, mc_x1, fm_y, mc_y1, num_steps)
        if verbose:
            print(f"  Generated MC samples: x shape={mc_x1.shape}, y shape={mc_y1.shape}")
        if ratio_estimator.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {ratio_estimator.loss_type}")
        mc_ratios = ratio_estimator._engine.eval(mc_x1, mc_y1, "ratio")
        if verbose:
            print(f"  MC ratios: min={mc_ratios.min():.4f}, max={mc_ratios.max():.4f}, "
                  f"mean={mc_ratios.mean():.4f}")

    if grad_guided:
        # "Gradient Log-Ratio" of the reference README (:159-164): v + gamma * grad log r(x_t, y_t) every step.  The
        # reference accepts only 'none' / 'mc_feng' and has no code for this mode; see rgfm_sample_pair_grad.
        if ratio_estimator.loss_type not in ("disc", "rulsif"):
            raise ValueError(f"Unknown loss_type: {ratio_estimator.loss_type}")
        _engine.sample_pair_grad(fm_x, fm_y, ratio_estimator, x_t, y_t, num_steps, guidance_strength)
        return x_t, y_t
    _engine.sample_pair(fm_x, fm_y, x_t, y_t, mc_x1, mc_y1, mc_ratios, num_steps, guidance_strength)
    return x_t, y_t


def sample_bimodal_guided(fm_x, fm_y, ratio_estimator=None, guidance_method='none',
                          guidance_strength=0.0, num_samples=16, num_steps=100, device='cuda',
                          mc_batch_size=64):
    """Pairs of 1x28x28 images, optional mc_feng guidance (reference :178-375).

    Returns ``(samples_x [n,1,28,28], samples_y [n,1,28,28])`` on `device`.
    Guidance is silently off when `ratio_estimator` is None, skipped on step 0
    (t > 1e-3 test), and `guidance_strength` is not clamped -- all as in the
    reference.  The reference's one-shot diagnostics print (:349-363) is not
    reproduced.
    """
    return paired_sampler(fm_x, fm_y, ratio_estimator, guidance_method, guidance_strength,
                          num_samples, num_steps, device, mc_batch_size, (1, 28, 28), (1, 28, 28))
