"""Timing of the BASELINE.json parity-case configurations that fit one GPU (synthetic weights, device-resident noise):
  cfg0: MNIST 28x28 FM_x only, 50 Euler steps, batch 64      (CFMSchedule.sample)
  cfg1: MNIST 28x28 pair, mc_feng 0.5, 100 steps, batch 256, N_mc 128 (sample_bimodal_guided)
  cfg2: MNIST32 + SVHN pair, mc_feng 0.5, 100 steps, batch 512, N_mc 256 (bench.py's workload)
Usage: python tools/bench_configs.py   (prints images/s per configuration, median of 3 after 1 warm-up)"""
import statistics
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from ratio_guided_multimodal_fm_amd import _engine, models as M  # noqa: E402
from ratio_guided_multimodal_fm_amd.synth import load_synth, paired_noise  # noqa: E402
from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


u28 = load_synth(M.FlowMatchingUNet(), 1).eval().to(dev)
u28y = load_synth(M.FlowMatchingUNet(), 2).eval().to(dev)
r28 = load_synth(M.RatioEstimator(), 3).eval().to(dev)
x0 = torch.randn(64, 1, 28, 28, device=dev)
t = timed(lambda: _engine.sample_single(u28, x0.clone(), 50))
print(f"cfg0  28x28 single net, B=64, 50 steps:            {1e3 * t:8.1f} ms/call  {64 / t:9.1f} images/s")

n1 = tuple(v.to(dev) for v in paired_noise(42, 256, 128, (1, 28, 28), (1, 28, 28)))
t = timed(lambda: paired_sampler(u28, u28y, r28, "mc_feng", 0.5, 256, 100, dev, 128, (1, 28, 28), (1, 28, 28), noise=n1,
                                 verbose=False))
print(f"cfg1  28x28 pair mc_feng, B=256, N=128, 100 steps:  {1e3 * t:8.1f} ms/call  {256 / t:9.1f} paired images/s")

fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
rr = load_synth(M.RatioEstimatorMNISTSVHN(), 2).eval().to(dev)
n2 = tuple(v.to(dev) for v in paired_noise(42, 512, 256, (1, 32, 32), (3, 32, 32)))
t = timed(lambda: paired_sampler(fm, fs, rr, "mc_feng", 0.5, 512, 100, dev, 256, (1, 32, 32), (3, 32, 32), noise=n2,
                                 verbose=False))
print(f"cfg2  MNIST32+SVHN mc_feng, B=512, N=256, 100 steps: {1e3 * t:8.1f} ms/call  {512 / t:9.1f} paired images/s")
