"""Timing of the FlowMatchingModel ('--model original') path: one Euler step = one net evaluation.
Usage: python tools/bench_fmnet.py [B] [steps]   (prints ms/step, images/s and the conv-class MFMA rate)"""
import sys
import time

import torch

from ratio_guided_multimodal_fm_amd import _engine, models as M
from ratio_guided_multimodal_fm_amd.synth import load_synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
m = load_synth(M.FlowMatchingModel(), 19).eval().to(dev)
x = torch.randn(B, 1, 28, 28, device=dev)
_engine.sample_single(m, x.clone(), 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
_engine.sample_single(m, x.clone(), steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} steps={steps}: {1e3 * dt / steps:.3f} ms/step, {B / dt:.1f} samples/s ({steps}-step integration)")
_engine.profile(enable=True, reset=True)
_engine.sample_single(m, x.clone(), steps)
torch.cuda.synchronize()
busy, tot, n, fl = _engine.profile_read(0)
print(f"conv class: {n} launches, {busy / steps:.3f} ms/step busy, {fl / busy / 1e9:.1f} TFLOP/s")
busy, tot, n, fl = _engine.profile_read(1)
print(f"other class: {n} launches, {busy / steps:.3f} ms/step busy")
_engine.profile(enable=False)
