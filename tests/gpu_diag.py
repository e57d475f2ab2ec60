#!/usr/bin/env python3
"""Prints per-layer / per-function GPU-vs-oracle differences without asserting
(debug aid for the GPU box: python tests/gpu_diag.py > gpurun_out/diag.txt)."""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from oracle import oracle as O
from helpers import golden, make_module, maxdiff, oracle_net, paired_noise
from ratio_guided_multimodal_fm_amd import _engine

dev = torch.device("cuda:0")


def section(name):
    print(f"\n=== {name}", flush=True)


def unet_layers(tag, shape, B=3):
    section(f"unet layers {tag} B={B}")
    m = make_module(tag, dev)
    desc, blob = oracle_net(tag)
    x = torch.randn(B, *shape, generator=torch.Generator().manual_seed(77))
    for tval in (0.0, 0.37):
        t = torch.full((B,), tval)
        out, acts = m._engine.forward_trace(x.to(dev), t.to(dev))
        torch.cuda.synchronize()
        ro, racts = O.unet_forward(desc, blob, x.numpy(), t.numpy(), trace=True)
        print(f" t={tval}: n_acts gpu={len(acts)} oracle={len(racts)}")
        for i, (a, r) in enumerate(zip(acts, racts)):
            a = a.cpu().numpy()
            print(f"  act {i:2d} shape {tuple(a.shape)} absmax {np.abs(r).max():8.3f} maxdiff {maxdiff(a, r):.3e}"
                  f" nan={np.isnan(a).any()}")
        print(f"  OUT maxdiff {maxdiff(out.cpu().numpy(), ro):.3e}")
    # single shared t
    out = m(x.to(dev), torch.tensor([0.5], device=dev))
    ro = O.unet_forward(desc, blob, x.numpy(), np.array([0.5], np.float32))
    print(f"  shared-t OUT maxdiff {maxdiff(out.cpu().numpy(), ro):.3e}")


def ratio(tag, sx, sy):
    section(f"ratio {tag}")
    m = make_module(tag, dev)
    kind, blob = oracle_net(tag)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(7, *sx, generator=g), torch.randn(7, *sy, generator=g)
    for loss in ("disc", "rulsif"):
        m.loss_type = loss
        s = m(x.to(dev), y.to(dev)).cpu().numpy()
        lr = m.log_ratio(x.to(dev), y.to(dev)).cpu().numpy()
        rs = O.ratio_eval(kind, blob, x.numpy(), y.numpy(), "score", loss)
        rl = O.ratio_eval(kind, blob, x.numpy(), y.numpy(), "log_ratio", loss)
        print(f" {loss}: score maxdiff {maxdiff(s, rs):.3e} log_ratio maxdiff {maxdiff(lr, rl):.3e}  (|s| max {np.abs(rs).max():.3f})")


def guidance():
    section("guidance")
    for (B, N, sx, sy) in ((6, 12, (1, 32, 32), (3, 32, 32)), (33, 70, (1, 28, 28), (1, 28, 28))):
        g = torch.Generator().manual_seed(3)
        x, y = torch.randn(B, *sx, generator=g), torch.randn(B, *sy, generator=g)
        vx, vy = torch.randn(B, *sx, generator=g), torch.randn(B, *sy, generator=g)
        mx, my = torch.randn(N, *sx, generator=g), torch.randn(N, *sy, generator=g)
        r = torch.exp(0.5 * torch.randn(N, generator=g))
        for t, gamma in ((0.05, 0.5), (0.5, 1.0), (0.9, 2.0), (0.99, 5.0)):
            gvx, gvy = vx.clone().to(dev), vy.clone().to(dev)
            w = _engine.guidance_apply(x.to(dev), y.to(dev), gvx, gvy, mx.to(dev), my.to(dev), r.to(dev), t, gamma, True)
            ovx, ovy, ow = O.guidance_apply(x.numpy(), y.numpy(), vx.numpy(), vy.numpy(), mx.numpy(), my.numpy(),
                                            r.numpy(), t, gamma, True)
            print(f" B={B} N={N} t={t} gamma={gamma}: vx {maxdiff(gvx.cpu().numpy(), ovx):.3e} vy "
                  f"{maxdiff(gvy.cpu().numpy(), ovy):.3e} w {maxdiff(w.cpu().numpy(), ow):.3e} |v|max {np.abs(ovx).max():.2f}")


def samplers():
    section("samplers vs golden")
    from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler
    g = golden("sampler_pair_ms")
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    for ci in range(8):
        guided, gamma, B, N, S, seed = g[f"c{ci}_cfg"]
        noise = paired_noise(int(seed), int(B), int(N) if guided else 0, (1, 32, 32), (3, 32, 32))
        t0 = time.time()
        xs, ys = paired_sampler(fm, fs, rr if guided else None, "mc_feng" if guided else "none", gamma, int(B),
                                int(S), dev, int(N), (1, 32, 32), (3, 32, 32), noise=noise, verbose=False)
        torch.cuda.synchronize()
        print(f" ms c{ci} guided={bool(guided)} gamma={gamma} B={int(B)} N={int(N)} S={int(S)}: x {maxdiff(xs.cpu().numpy(), g[f'c{ci}_x']):.3e}"
              f" y {maxdiff(ys.cpu().numpy(), g[f'c{ci}_y']):.3e}  ({time.time() - t0:.2f}s)")
    g = golden("sampler_pair28")
    fx, fy, rr = make_module("unet28", dev), make_module("unet28_y", dev), make_module("ratio28", dev)
    for ci in range(3):
        guided, gamma, B, N, S, seed = g[f"c{ci}_cfg"]
        noise = paired_noise(int(seed), int(B), int(N) if guided else 0, (1, 28, 28), (1, 28, 28))
        xs, ys = paired_sampler(fx, fy, rr if guided else None, "mc_feng" if guided else "none", gamma, int(B),
                                int(S), dev, int(N), (1, 28, 28), (1, 28, 28), noise=noise, verbose=False)
        print(f" 28 c{ci}: x {maxdiff(xs.cpu().numpy(), g[f'c{ci}_x']):.3e} y {maxdiff(ys.cpu().numpy(), g[f'c{ci}_y']):.3e}")
    g = golden("sampler_cfm28")
    n, steps, seed = (int(v) for v in g["cfg"])
    x0 = torch.randn(n, 1, 28, 28, generator=torch.Generator().manual_seed(seed)).to(dev)
    xs = _engine.sample_single(fx, x0, steps)
    print(f" cfm28: x {maxdiff(xs.cpu().numpy(), g['x']):.3e}")


def main():
    print(torch.cuda.get_device_name(0))
    steps = [lambda: unet_layers("mnist32", (1, 32, 32)), lambda: unet_layers("svhn", (3, 32, 32), B=5),
             lambda: unet_layers("unet28", (1, 28, 28)), lambda: ratio("ratio_ms", (1, 32, 32), (3, 32, 32)),
             lambda: ratio("ratio28", (1, 28, 28), (1, 28, 28)), guidance, samplers]
    for s in steps:
        try:
            s()
        except Exception:
            traceback.print_exc(file=sys.stdout)
            sys.stdout.flush()


if __name__ == "__main__":
    main()
