"""Is the small-row pre-phase (what one rank of an 8-GPU run integrates: 32 rows per net) bound by the host's launch rate
or by the kernels?  Wall time per step of each net alone and of both on two streams, against the sum of the kernel
durations of the same loop (run under `rocprofv3 --kernel-trace --stats` for the latter).  Development tool, one GPU.

  python tools/small_rows.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from ratio_guided_multimodal_fm_amd import _engine, models as M  # noqa: E402
from ratio_guided_multimodal_fm_amd.synth import load_synth  # noqa: E402

dev = torch.device("cuda:0")
fm = load_synth(M.FlowMatchingUNetMNIST(32), 0).eval().to(dev)
fs = load_synth(M.FlowMatchingUNetSVHN(), 1).eval().to(dev)
S = 50


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


for nb in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    xx, yy = torch.randn(nb, 1, 32, 32, device=dev), torch.randn(nb, 3, 32, 32, device=dev)
    tx = t(lambda: _engine._sample_single(fm, xx.clone(), S))
    ty = t(lambda: _engine._sample_single(fs, yy.clone(), S))
    tb = t(lambda: _engine.sample_two_streams(fm, xx.clone(), fs, yy.clone(), S))
    # host side only: enqueue time of one loop (no synchronise inside)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _engine._sample_single(fs, yy.clone(), S)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{nb:4d} rows: MNIST net {1e3 * tx / S:.3f} ms/step, SVHN net {1e3 * ty / S:.3f}, both on two streams {1e3 * tb / S:.3f}; "
          f"host enqueue of the SVHN loop {1e3 * th / S:.3f} ms/step", flush=True)
