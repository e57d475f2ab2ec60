"""Eager vs hipGraph-replayed Euler loop on the benchmark shape (development tool, one GPU): time per call of the
guided main loop with RGFM_GRAPH=0 and =1, interleaved, same process."""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from ratio_guided_multimodal_fm_amd import _engine, models as M  # noqa: E402
from ratio_guided_multimodal_fm_amd.synth import load_synth, paired_noise  # noqa: E402

dev = torch.device("cuda:0")
fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
rr = load_synth(M.RatioEstimatorMNISTSVHN(), 2).eval().to(dev)
for B in (512, 64):
    x0, y0, mx0, my0 = (v.to(dev) for v in paired_noise(42, B, 256, (1, 32, 32), (3, 32, 32)))
    mx1, my1 = mx0.clone(), my0.clone()
    _engine.sample_two_streams(fm, mx1, fs, my1, 100)
    r = rr._engine.eval(mx1, my1, "ratio")
    ts = {"0": [], "1": []}
    for rep in range(4):
        for g in ("0", "1"):
            os.environ["RGFM_GRAPH"] = g
            xa, ya = x0.clone(), y0.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _engine.sample_pair(fm, fs, xa, ya, mx1, my1, r, 100, 0.5)
            torch.cuda.synchronize()
            if rep:
                ts[g].append(time.perf_counter() - t0)
    e, gr = sorted(ts["0"])[1], sorted(ts["1"])[1]
    print(f"B={B}: guided main loop, 100 steps: eager {1e3 * e:.1f} ms, graph replay {1e3 * gr:.1f} ms ({e / gr:.3f}x)")
