"""GPU tests of the BASELINE configurations the one-GPU box can cover beyond tests/test_gpu_parity.py:

* configs[4] in miniature -- the reference's guidance-strength sweep at 200 Euler steps driven through
  ``evaluate_mnist_svhn.run_sweep`` on the real HIP sampler, against a reference-generated fixture
  (tests/golden/sweep200.npz: samples, classifier logits and coherence per configuration);
* configs[3]'s per-rank shape -- gamma 1.0, 512 rows, N_mc 256, all 100 steps -- through
  ``sharded_paired_sampler`` under a one-rank RCCL group, rows compared with the CPU oracle over the whole
  trajectory including the stiff last steps (t >= 0.9);
* the checkpoint boundary: both on-disk formats through ``load_checkpoint`` and the CLIs
  ``sample_mnist_svhn.main`` / ``evaluate_mnist_svhn.main`` / ``sample.main`` end to end from checkpoint files;
* the time-embedding table against the reference's ``timestep_embedding`` fixture (a7).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from helpers import golden, make_module, maxdiff, oracle_net, paired_noise
from ratio_guided_multimodal_fm_amd import _engine, _lib

pytestmark = pytest.mark.gpu

TOL_SAMPLER = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _lib.lib()
    return torch.device("cuda:0")


def test_sweep_200_steps_on_the_hip_sampler(dev):
    """evaluate_mnist_svhn.run_sweep (methods x strengths, skip rule, one seeding, no re-seed) with the real HIP
    sampler at 200 steps: samples 1e-4, the classifiers' logits on them, coherence per configuration."""
    from ratio_guided_multimodal_fm_amd.evaluate_mnist_svhn import run_sweep
    from ratio_guided_multimodal_fm_amd.sample_mnist_svhn import sample_bimodal_guided_mnist_svhn  # noqa: F401
    from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler
    g = golden("sweep200")
    B, N, S, seed = (int(v) for v in g["cfg"])
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    cm, cs = make_module("clf_mnist", dev), make_module("clf_svhn", dev)
    produced = []

    def sampler(fm_m, fm_s, ratio, method, strength, n, steps, device, mc):
        # the reference draws x0, y0[, mc_x0, mc_y0] from ONE generator stream seeded once before the sweep
        # (evaluate_mnist_svhn.py:80); the fixture was produced on the torch CPU generator, so the draws are
        # made there, in the reference's order, and uploaded
        guided = method == "mc_feng" and ratio is not None
        x0 = torch.randn(n, 1, 32, 32)
        y0 = torch.randn(n, 3, 32, 32)
        mx = torch.randn(mc, 1, 32, 32) if guided else None
        my = torch.randn(mc, 3, 32, 32) if guided else None
        xs, ys = paired_sampler(fm_m, fm_s, ratio, method, strength, n, steps, device, mc, (1, 32, 32), (3, 32, 32),
                                noise=(x0, y0, mx, my), verbose=False)
        produced.append((xs, ys))
        return xs, ys

    torch.manual_seed(seed)
    res = run_sweep(fm, fs, lambda: rr, cm, cs, ["none", "mc_feng"], [float(v) for v in g["strengths"]], B, S, dev, N,
                    sampler=sampler)
    assert len(res) == int(g["n_cfg"]) == len(produced)
    for ci, (r, (xs, ys)) in enumerate(zip(res, produced)):
        assert (r["method"] == "mc_feng") == bool(g[f"c{ci}_guided"]) and r["guidance_strength"] == float(g[f"c{ci}_gamma"])
        assert maxdiff(xs.cpu().numpy(), g[f"c{ci}_x"]) < TOL_SAMPLER, ci
        assert maxdiff(ys.cpu().numpy(), g[f"c{ci}_y"]) < TOL_SAMPLER, ci
        with torch.no_grad():
            lm, ls = cm(xs).cpu().numpy(), cs(ys).cpu().numpy()
            # classifiers themselves: 1e-5 on the reference's own samples
            assert maxdiff(cm(torch.from_numpy(g[f"c{ci}_x"]).to(dev)).cpu().numpy(), g[f"c{ci}_logits_mnist"]) < 1e-5
            assert maxdiff(cs(torch.from_numpy(g[f"c{ci}_y"]).to(dev)).cpu().numpy(), g[f"c{ci}_logits_svhn"]) < 1e-5
        # logits on the HIP samples: the samples differ by <= 1e-4, the classifiers amplify that a little
        assert maxdiff(lm, g[f"c{ci}_logits_mnist"]) < 2e-3 and maxdiff(ls, g[f"c{ci}_logits_svhn"]) < 2e-3
        assert np.array_equal(lm.argmax(1), g[f"c{ci}_logits_mnist"].argmax(1))
        assert np.array_equal(ls.argmax(1), g[f"c{ci}_logits_svhn"].argmax(1))
        assert abs(r["coherence_acc"] - float(g[f"c{ci}_coherence_acc"])) < 1e-7 and r["num_samples"] == B


def test_config3_rank_shape_through_sharded_sampler(dev):
    """BASELINE configs[3] as one rank sees it at 8 GPUs: gamma 1.0, 512 rows, N_mc 256, ALL 100 steps, through
    sharded_paired_sampler with the RCCL collectives (one-rank group).  Eight rows (first / middle / last) are
    followed by the CPU oracle over the whole trajectory -- including t >= 0.9, where sigma_t^2 ~ 1e-4..1e-2
    makes the importance weights near one-hot -- from the same MC set."""
    import torch.distributed as dist
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    B, N, S, gamma = 512, 256, 100, 1.0
    noise = paired_noise(12, B, N, (1, 32, 32), (3, 32, 32))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29573")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        xs, ys = sharded_paired_sampler(fm, fs, rr, "mc_feng", gamma, S, noise, dev, gather="rank0")
    finally:
        dist.destroy_process_group()
    assert torch.isfinite(xs).all() and torch.isfinite(ys).all()
    # the MC set the sampler used (deterministic: same calls on the same noise)
    mx1, my1 = noise[2].to(dev, copy=True), noise[3].to(dev, copy=True)
    _engine.sample_two_streams(fm, mx1, fs, my1, S)
    r = rr._engine.eval(mx1, my1, "ratio")
    dx, bx = oracle_net("mnist32")
    dy, by = oracle_net("svhn")
    rows = [0, 1, 127, 255, 256, 383, 510, 511]
    x0, y0 = noise[0][rows].numpy(), noise[1][rows].numpy()
    mxn, myn, rn = mx1.cpu().numpy(), my1.cpu().numpy(), r.cpu().numpy()
    # whole trajectory
    ox, oy = O.sample_pair(dx, bx, dy, by, x0, y0, mxn, myn, rn, S, gamma, 0, S)
    assert maxdiff(xs[rows].cpu().numpy(), ox) < TOL_SAMPLER
    assert maxdiff(ys[rows].cpu().numpy(), oy) < TOL_SAMPLER
    # the last ten steps on their own, started from the oracle's state at t = 0.9: the HIP guidance at
    # near-one-hot weights, at the benchmark N
    ox90, oy90 = O.sample_pair(dx, bx, dy, by, x0, y0, mxn, myn, rn, S, gamma, 0, 90)
    xg, yg = torch.from_numpy(ox90).to(dev), torch.from_numpy(oy90).to(dev)
    _engine.sample_pair(fm, fs, xg, yg, mx1, my1, r, S, gamma, 90, S)
    ox2, oy2 = O.sample_pair(dx, bx, dy, by, ox90, oy90, mxn, myn, rn, S, gamma, 90, S)
    assert maxdiff(xg.cpu().numpy(), ox2) < TOL_SAMPLER and maxdiff(yg.cpu().numpy(), oy2) < TOL_SAMPLER


def test_config0_full_size_cfm_sampler(dev, monkeypatch):
    """BASELINE configs[0] at FULL size: CFMSchedule.sample semantics -- MNIST 28x28 net alone, no
    guidance, batch 64, all 50 Euler steps (reference src/utils/flow_utils.py:69-100) -- first / last rows followed by
    the CPU oracle over the whole trajectory, and every row compared between the default arithmetic and the exact-fp32
    MFMA path."""
    fm = make_module("unet28", dev)
    B, S = 64, 50
    x0 = torch.randn(B, 1, 28, 28, generator=torch.Generator().manual_seed(50))
    xa = x0.to(dev, copy=True)
    _engine.sample_single(fm, xa, S)
    monkeypatch.setenv("RGFM_CONV", "f32")
    xb = x0.to(dev, copy=True)
    _engine.sample_single(fm, xb, S)
    monkeypatch.delenv("RGFM_CONV")
    assert torch.isfinite(xa).all()
    assert maxdiff(xa.cpu().numpy(), xb.cpu().numpy()) < TOL_SAMPLER
    d, b = oracle_net("unet28")
    rows = [0, 1, 31, 62, 63]
    ox = O.sample_single(d, b, x0[rows].numpy(), S)
    assert maxdiff(xa[rows].cpu().numpy(), ox) < TOL_SAMPLER


def test_config2_own_gamma_all_steps(dev):
    """BASELINE configs[2] with ITS OWN guidance strength (gamma 0.5; the rank-shape test above runs gamma 1.0): batch
    512, N_mc 256, ALL 100 steps on the HIP path, eight rows followed by the CPU oracle from the same MC set."""
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    B, N, S, gamma = 512, 256, 100, 0.5
    noise = paired_noise(13, B, N, (1, 32, 32), (3, 32, 32))
    mx1, my1 = noise[2].to(dev, copy=True), noise[3].to(dev, copy=True)
    _engine.sample_two_streams(fm, mx1, fs, my1, S)
    r = rr._engine.eval(mx1, my1, "ratio")
    xa, ya = noise[0].to(dev, copy=True), noise[1].to(dev, copy=True)
    before = _engine.range_fallbacks
    _engine.sample_pair(fm, fs, xa, ya, mx1, my1, r, S, gamma)
    assert _engine.range_fallbacks == before  # the benchmark workload runs on the default arithmetic, no fallback
    assert torch.isfinite(xa).all() and torch.isfinite(ya).all()
    dx, bx = oracle_net("mnist32")
    dy, by = oracle_net("svhn")
    rows = [0, 1, 127, 255, 256, 383, 510, 511]
    ox, oy = O.sample_pair(dx, bx, dy, by, noise[0][rows].numpy(), noise[1][rows].numpy(), mx1.cpu().numpy(),
                           my1.cpu().numpy(), r.cpu().numpy(), S, gamma, 0, S)
    assert maxdiff(xa[rows].cpu().numpy(), ox) < TOL_SAMPLER
    assert maxdiff(ya[rows].cpu().numpy(), oy) < TOL_SAMPLER


def test_config4_rank_shape_200_steps_strong_guidance(dev):
    """BASELINE configs[4] as one rank sees it at 8 GPUs: 1024 rows, N_mc 256, ALL 200 Euler steps
    (reference sweep: evaluate_mnist_svhn.py:130-165), for the two extrapolating guidance strengths gamma = 2 and 5
    (gamma is not clamped, sample_mnist_svhn.py:170); rows from both ends and the middle are followed by the CPU
    oracle over the whole trajectory from the same MC set."""
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    B, N, S = 1024, 256, 200
    noise = paired_noise(15, B, N, (1, 32, 32), (3, 32, 32))
    mx1, my1 = noise[2].to(dev, copy=True), noise[3].to(dev, copy=True)
    _engine.sample_two_streams(fm, mx1, fs, my1, S)
    r = rr._engine.eval(mx1, my1, "ratio")
    dx, bx = oracle_net("mnist32")
    dy, by = oracle_net("svhn")
    mxn, myn, rn = mx1.cpu().numpy(), my1.cpu().numpy(), r.cpu().numpy()
    before = _engine.range_fallbacks
    for gamma, rows in ((2.0, [0, 1, 511, 512, 1022, 1023]), (5.0, [2, 300, 513, 700, 1000, 1021])):
        xa, ya = noise[0].to(dev, copy=True), noise[1].to(dev, copy=True)
        _engine.sample_pair(fm, fs, xa, ya, mx1, my1, r, S, gamma)
        assert torch.isfinite(xa).all() and torch.isfinite(ya).all()
        ox, oy = O.sample_pair(dx, bx, dy, by, noise[0][rows].numpy(), noise[1][rows].numpy(), mxn, myn, rn, S, gamma, 0, S)
        dxm, dym = maxdiff(xa[rows].cpu().numpy(), ox), maxdiff(ya[rows].cpu().numpy(), oy)
        print(f"gamma={gamma}: max|dx| {dxm:.2e} max|dy| {dym:.2e} (|x| {float(np.abs(ox).max()):.1f}, |y| {float(np.abs(oy).max()):.1f})")
        assert dxm < TOL_SAMPLER and dym < TOL_SAMPLER, (gamma, dxm, dym)
    assert _engine.range_fallbacks == before


def test_time_embedding_table_against_reference_fixture(dev):
    """a7: the sinusoidal embedding (cos half first, unscaled t; unet_flexible.py:16-36) as the DEVICE evaluates it
    in front of the time MLPs (rgfm_unet_time_embedding), against the reference's timestep_embedding fixture, for
    both model widths (32: MNIST net, 64: SVHN net); plus the host-side mirror of the module API."""
    from ratio_guided_multimodal_fm_amd.models.unet_flexible import timestep_embedding
    g = golden("timestep_embedding")
    t = torch.tensor(g["t"], dtype=torch.float32)
    for tag, dim in (("mnist32", 32), ("svhn", 64)):
        ref = g[f"emb{dim}"]
        assert maxdiff(timestep_embedding(t, dim).numpy(), ref) < 1e-6
        m = make_module(tag, dev)
        got = m._engine.time_embedding(t.to(dev)).cpu().numpy()
        assert got.shape == ref.shape and maxdiff(got, ref) < 2e-6, (tag, maxdiff(got, ref))


def test_graph_replay_is_bitwise_the_eager_path(dev, monkeypatch):
    """The guided Euler loop as ONE captured hipGraph replayed per step (RGFM_GRAPH=1) against the kernel-by-kernel path
    (default): same kernels, same arguments (the time-table row and the step's guidance scalars are read on the
    device through a step counter), so the results must be bit-identical -- also for a step range that does not
    start at 0 and with the two nets on one stream."""
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    B, N, S = 9, 6, 12
    x0, y0, mx0, my0 = (t.to(dev) for t in paired_noise(14, B, N, (1, 32, 32), (3, 32, 32)))
    mx1, my1 = mx0.clone(), my0.clone()
    _engine.sample_two_streams(fm, mx1, fs, my1, S)
    r = rr._engine.eval(mx1, my1, "ratio")
    for env, (sb, se) in (({}, (0, S)), ({}, (3, S)), ({"RGFM_OVERLAP": "0"}, (0, S))):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        res = {}
        for graph in ("1", "0"):
            monkeypatch.setenv("RGFM_GRAPH", graph)
            xa, ya = x0.clone(), y0.clone()
            _engine.sample_pair(fm, fs, xa, ya, mx1, my1, r, S, 0.7, sb, se)
            res[graph] = (xa.clone(), ya.clone())
        assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1]), (env, sb)
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.delenv("RGFM_GRAPH")
    # and against the oracle (12 steps from 0)
    dx, bx = oracle_net("mnist32")
    dy, by = oracle_net("svhn")
    xa, ya = x0.clone(), y0.clone()
    _engine.sample_pair(fm, fs, xa, ya, mx1, my1, r, S, 0.7)
    ox, oy = O.sample_pair(dx, bx, dy, by, x0.cpu().numpy(), y0.cpu().numpy(), mx1.cpu().numpy(), my1.cpu().numpy(),
                           r.cpu().numpy(), S, 0.7)
    assert maxdiff(xa.cpu().numpy(), ox) < TOL_SAMPLER and maxdiff(ya.cpu().numpy(), oy) < TOL_SAMPLER


# ------------------------------------------------------------------ gradient log-ratio guidance
@pytest.mark.parametrize("loss", ["disc", "rulsif"])
def test_ratio_gradient_vs_autograd_fixture_and_oracle(dev, loss):
    """d log r / d(x, y) of RatioEstimatorMNISTSVHN on the HIP path (hand-written reverse pass, csrc/ratio_grad.hip)
    against torch.autograd.grad on the reference module (tests/golden/ratio_grad.npz) and against the oracle on a
    ragged batch.  Tolerance: relative 1e-4 of the largest gradient entry (fp32 forward + reverse pass)."""
    from ratio_guided_multimodal_fm_amd import models as M
    from ratio_guided_multimodal_fm_amd.synth import load_synth
    g = golden("ratio_grad")
    rr = load_synth(M.RatioEstimatorMNISTSVHN(loss_type=loss), 16).eval().to(dev)
    gen = torch.Generator().manual_seed(85)
    x = torch.randn(3, 1, 32, 32, generator=gen)
    y = torch.randn(3, 3, 32, 32, generator=gen)
    for tag, sc in (("n", 1.0), ("s", 0.3)):
        gx, gy, lr = rr._engine.grad_log_ratio((x * sc).to(dev), (y * sc).to(dev))
        rx, ry = g[f"{loss}_{tag}_gx"], g[f"{loss}_{tag}_gy"]
        assert maxdiff(lr.cpu().numpy(), g[f"{loss}_{tag}_lr"]) < 1e-5
        assert maxdiff(gx.cpu().numpy(), rx) < 1e-4 * np.abs(rx).max(), (tag, maxdiff(gx.cpu().numpy(), rx))
        assert maxdiff(gy.cpu().numpy(), ry) < 1e-4 * np.abs(ry).max(), (tag, maxdiff(gy.cpu().numpy(), ry))
        a, b = rr.grad_log_ratio((x * sc).to(dev), (y * sc).to(dev))  # the module-level API
        assert torch.equal(a, gx) and torch.equal(b, gy)
    # ragged batch (partial tiles at the 2x2 / 4x4 levels) against the oracle
    _, br = oracle_net("ratio_ms")
    xb = torch.randn(7, 1, 32, 32, generator=gen)
    yb = torch.randn(7, 3, 32, 32, generator=gen)
    gx, gy, lr = rr._engine.grad_log_ratio(xb.to(dev), yb.to(dev))
    ox, oy, olr = O.ratio_grad(br, xb.numpy(), yb.numpy(), loss)
    assert maxdiff(lr.cpu().numpy(), olr) < 1e-5
    assert maxdiff(gx.cpu().numpy(), ox) < 1e-4 * np.abs(ox).max() and maxdiff(gy.cpu().numpy(), oy) < 1e-4 * np.abs(oy).max()
    with pytest.raises(_lib.RgfmError):  # the estimator's own image shapes only
        rr._engine.grad_log_ratio(torch.zeros(1, 1, 28, 28, device=dev), torch.zeros(1, 1, 28, 28, device=dev))


@pytest.mark.parametrize("loss", ["disc", "rulsif"])
def test_ratio28_gradient_vs_autograd_fixture_and_oracle(dev, loss):
    """The same for the 28x28 RatioEstimator (GroupNorm encoders, ratio_estimator.py:34-191; 28 -> 14 -> 7 -> 3 with a
    floor pooling that leaves the 7x7 map's last row and column without gradient): HIP reverse pass
    (grad_act_kernel<true> + gn_bwd_kernel + the transposed-weight convs) against torch.autograd.grad on the reference
    module (tests/golden/ratio_grad28.npz) and against the oracle on a ragged batch."""
    from ratio_guided_multimodal_fm_amd import models as M
    from ratio_guided_multimodal_fm_amd.synth import load_synth
    g = golden("ratio_grad28")
    rr = load_synth(M.RatioEstimator(loss_type=loss), 15).eval().to(dev)
    gen = torch.Generator().manual_seed(86)
    x = torch.randn(3, 1, 28, 28, generator=gen)
    y = torch.randn(3, 1, 28, 28, generator=gen)
    for tag, sc in (("n", 1.0), ("s", 0.3)):
        gx, gy, lr = rr._engine.grad_log_ratio((x * sc).to(dev), (y * sc).to(dev))
        rx, ry = g[f"{loss}_{tag}_gx"], g[f"{loss}_{tag}_gy"]
        assert maxdiff(lr.cpu().numpy(), g[f"{loss}_{tag}_lr"]) < 1e-5
        assert maxdiff(gx.cpu().numpy(), rx) < 1e-4 * np.abs(rx).max(), (tag, maxdiff(gx.cpu().numpy(), rx))
        assert maxdiff(gy.cpu().numpy(), ry) < 1e-4 * np.abs(ry).max(), (tag, maxdiff(gy.cpu().numpy(), ry))
        a, b = rr.grad_log_ratio((x * sc).to(dev), (y * sc).to(dev))  # the module-level API
        assert torch.equal(a, gx) and torch.equal(b, gy)
    _, br = oracle_net("ratio28")
    xb = torch.randn(37, 1, 28, 28, generator=gen)
    yb = torch.randn(37, 1, 28, 28, generator=gen)
    gx, gy, lr = rr._engine.grad_log_ratio(xb.to(dev), yb.to(dev))
    ox, oy, olr = O.ratio_grad(br, xb.numpy(), yb.numpy(), loss, kind="mnist28")
    assert maxdiff(lr.cpu().numpy(), olr) < 1e-5
    assert maxdiff(gx.cpu().numpy(), ox) < 1e-4 * np.abs(ox).max() and maxdiff(gy.cpu().numpy(), oy) < 1e-4 * np.abs(oy).max()
    with pytest.raises(_lib.RgfmError):
        rr._engine.grad_log_ratio(torch.zeros(1, 1, 32, 32, device=dev), torch.zeros(1, 3, 32, 32, device=dev))


def test_grad_log_ratio_sampler28_vs_oracle(dev):
    """guidance_method='grad_log_ratio' on the 28x28 pair (two FlowMatchingUNet nets + RatioEstimator) against the
    oracle's composition (parity unpinned for the loop, pinned for the gradient, as for the 32x32 pair)."""
    from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler
    fx, fy, rr = make_module("unet28", dev), make_module("unet28_y", dev), make_module("ratio28", dev)
    noise = paired_noise(34, 5, 0, (1, 28, 28), (1, 28, 28))
    S, gamma = 6, 2.0
    xs, ys = paired_sampler(fx, fy, rr, "grad_log_ratio", gamma, 5, S, dev, 4, (1, 28, 28), (1, 28, 28), noise=noise, verbose=False)
    dx, bx = oracle_net("unet28")
    dy, by = oracle_net("unet28_y")
    _, br = oracle_net("ratio28")
    ox, oy = O.sample_pair_grad(dx, bx, dy, by, br, noise[0].numpy(), noise[1].numpy(), S, gamma, kind="mnist28")
    assert maxdiff(xs.cpu().numpy(), ox) < TOL_SAMPLER and maxdiff(ys.cpu().numpy(), oy) < TOL_SAMPLER
    xn, yn = paired_sampler(fx, fy, None, "none", 0.0, 5, S, dev, 4, (1, 28, 28), (1, 28, 28), noise=noise, verbose=False)
    assert maxdiff(xs.cpu().numpy(), xn.cpu().numpy()) > 1e-5  # (the guidance does something)


def test_grad_log_ratio_sampler_vs_oracle(dev):
    """guidance_method='grad_log_ratio' (reference README.md:159-164; no reference code exists, so the composition
    x <- x + (v + gamma grad log r) dt is checked against this build's own oracle: "parity unpinned" for the loop,
    pinned for the gradient).  gamma 0 must equal the unguided sampler."""
    from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    noise = paired_noise(33, 5, 0, (1, 32, 32), (3, 32, 32))
    S, gamma = 6, 2.0
    xs, ys = paired_sampler(fm, fs, rr, "grad_log_ratio", gamma, 5, S, dev, 4, (1, 32, 32), (3, 32, 32), noise=noise, verbose=False)
    dx, bx = oracle_net("mnist32")
    dy, by = oracle_net("svhn")
    _, br = oracle_net("ratio_ms")
    ox, oy = O.sample_pair_grad(dx, bx, dy, by, br, noise[0].numpy(), noise[1].numpy(), S, gamma)
    assert maxdiff(xs.cpu().numpy(), ox) < TOL_SAMPLER and maxdiff(ys.cpu().numpy(), oy) < TOL_SAMPLER
    x0, y0 = paired_sampler(fm, fs, rr, "grad_log_ratio", 0.0, 5, S, dev, 4, (1, 32, 32), (3, 32, 32), noise=noise, verbose=False)
    xn, yn = paired_sampler(fm, fs, None, "none", 0.0, 5, S, dev, 4, (1, 32, 32), (3, 32, 32), noise=noise, verbose=False)
    assert maxdiff(x0.cpu().numpy(), xn.cpu().numpy()) < 1e-6 and maxdiff(y0.cpu().numpy(), yn.cpu().numpy()) < 1e-6
    assert maxdiff(xs.cpu().numpy(), xn.cpu().numpy()) > 1e-5  # (the guidance does something)


def test_grad_sampler_reverse_pass_on_the_two_plane_arithmetic(dev, monkeypatch):
    """Inside the gradient-guided sampler the estimator's REVERSE convs run on the U-Nets' default arithmetic (two scaled
    fp16 planes) with a measured power-of-two pre-scale of every gradient tensor (ConvArgs::in_amax; round 4).  One Euler
    step with a huge gamma makes the update x1 - x0 ~ dt gamma grad log r, so the difference between the default run and
    the run with the reverse convs on the exact fp32 MFMA (RGFM_REV_HX2=0; the same forward pass, hence the same max-pool
    routing), divided by dt gamma, IS the error the reverse arithmetic puts into the gradient: it must be far inside the
    gradient's stated tolerance (1e-4 of its largest entry).  The stand-alone gradient (always exact fp32) gives the
    scale; against RGFM_CONV=f32 -- whose FORWARD pass differs at 1e-6, enough to flip a max-pool argmax here and there --
    most samples must still agree."""
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    B, S, gamma = 48, 100, 1.0e4
    g = torch.Generator().manual_seed(41)
    x0 = torch.randn(B, 1, 32, 32, generator=g).to(dev)
    y0 = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    gx, gy, _ = rr._engine.grad_log_ratio(x0, y0)
    res = {}
    before = _engine.range_fallbacks
    for tag, env in (("hx2", {}), ("rev32", {"RGFM_REV_HX2": "0"}), ("f32", {"RGFM_CONV": "f32"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x, y = x0.clone(), y0.clone()
        _engine.sample_pair_grad(fm, fs, rr, x, y, S, gamma, 10, 11)
        res[tag] = (x.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64))
        for k in env:
            monkeypatch.delenv(k)
    assert _engine.range_fallbacks == before
    dtg = gamma / S
    mx, my = float(gx.abs().max()), float(gy.abs().max())
    # (the update really is the gradient's: the exact run moved by ~ dt gamma g)
    moved = np.abs(res["f32"][0] - x0.cpu().numpy()).max() / dtg
    assert 0.5 * mx < moved < 2.0 * mx + 1e-3, (moved, mx)
    ex = np.abs(res["hx2"][0] - res["rev32"][0]).max() / dtg
    ey = np.abs(res["hx2"][1] - res["rev32"][1]).max() / dtg
    print(f"gradient error of the two-plane reverse convs: x {ex:.2e} (max|g| {mx:.2e}), y {ey:.2e} (max|g| {my:.2e})")
    assert 0.0 < ex + ey and ex < 1e-5 * mx and ey < 1e-5 * my, (ex, mx, ey, my)
    # against the all-fp32 run: a flipped argmax moves a whole receptive field of one sample, so the statement is per
    # sample -- most samples inside 1e-4 of the maximum
    for k, m in ((0, mx), (1, my)):
        d = (np.abs(res["hx2"][k] - res["f32"][k]) / dtg).reshape(B, -1).max(axis=1)
        print(f"vs the all-fp32 run, plane {k}: {int((d < 1e-4 * m).sum())}/{B} samples inside 1e-4 max|g|, median {np.median(d):.2e}, max {d.max():.2e}")
        assert np.median(d) < 1e-4 * m, (k, float(np.median(d)), m)


# ------------------------------------------------------------------ checkpoints and CLIs
def _write_checkpoints(tmp_path):
    """The files the reference's trainers write, from synthetic weights: dict format for the two flow nets
    (train_flow_mnist32.py:137-143, train_flow_svhn.py:164-170), raw state_dict for the ratio estimator and the
    classifiers (train_ratio_mnist_svhn.py:142-143, train_classifiers_mnist_svhn.py:152-153,172-173)."""
    ck = tmp_path / "checkpoints"
    ck.mkdir()
    fm, fs, rr = make_module("mnist32"), make_module("svhn"), make_module("ratio_ms")
    cm, cs = make_module("clf_mnist"), make_module("clf_svhn")
    torch.save({"epoch": 7, "model_state_dict": fm.state_dict(), "optimizer_state_dict": {}, "best_loss": 0.25},
               ck / "flow_mnist32_best.pth")
    torch.save({"epoch": 3, "model_state_dict": fs.state_dict(), "optimizer_state_dict": {}, "best_loss": 0.5},
               ck / "flow_svhn_best.pth")
    torch.save(rr.state_dict(), ck / "ratio_disc_mnist_svhn_best.pth")
    torch.save(cm.state_dict(), ck / "mnist32_classifier.pth")
    torch.save(cs.state_dict(), ck / "svhn_classifier.pth")
    return fm, fs, rr, cm, cs


def test_checkpoint_formats_give_bitwise_equal_forward(dev, tmp_path):
    from ratio_guided_multimodal_fm_amd import models as M
    from ratio_guided_multimodal_fm_amd.utils import load_checkpoint
    fm, fs, rr, _, _ = _write_checkpoints(tmp_path)
    ck = tmp_path / "checkpoints"
    raw = tmp_path / "raw.pth"
    torch.save(fm.state_dict(), raw)  # train_flow.py:101 format
    x = torch.randn(3, 1, 32, 32, generator=torch.Generator().manual_seed(1)).to(dev)
    t = torch.tensor([0.2, 0.5, 0.7], device=dev)
    ref = fm.to(dev)(x, t)
    a = M.FlowMatchingUNetMNIST(32).to(dev).eval()
    info = load_checkpoint(a, str(ck / "flow_mnist32_best.pth"), dev)
    assert info == {"epoch": 7, "best_loss": 0.25}
    b = M.FlowMatchingUNetMNIST(32).to(dev).eval()
    assert load_checkpoint(b, str(raw), dev) == {}
    assert torch.equal(a(x, t), ref) and torch.equal(b(x, t), ref)
    with pytest.raises(RuntimeError):  # a wrong architecture fails like torch's load_state_dict does
        load_checkpoint(M.FlowMatchingUNetSVHN().to(dev), str(raw), dev)


def test_cli_mains_from_checkpoint_files(dev, tmp_path, monkeypatch):
    """sample_mnist_svhn.main and evaluate_mnist_svhn.main end to end from checkpoint files (tiny B / steps),
    outputs compared with the direct API call under the same seed."""
    import ratio_guided_multimodal_fm_amd as R
    from ratio_guided_multimodal_fm_amd import evaluate_mnist_svhn, sample_mnist_svhn
    fm, fs, rr, cm, cs = _write_checkpoints(tmp_path)
    monkeypatch.chdir(tmp_path)
    assert sample_mnist_svhn.main(["--guidance_method", "mc_feng", "--guidance_strength", "0.5", "--num_samples", "3",
                                   "--num_steps", "4", "--mc_batch_size", "5", "--seed", "9"]) == 0
    saved = torch.load(tmp_path / "outputs" / "mnist_svhn" / "samples_mc_feng_gamma0.5.pt")
    R.utils.set_seed(9)
    xs, ys = R.sample_bimodal_guided_mnist_svhn(fm.to(dev), fs.to(dev), rr.to(dev), "mc_feng", 0.5, num_samples=3,
                                                num_steps=4, device=dev, mc_batch_size=5)
    assert torch.equal(saved["mnist"], xs.cpu()) and torch.equal(saved["svhn"], ys.cpu())
    # missing checkpoint: the reference prints an error and returns (sample_mnist_svhn.py:283-291)
    os.rename(tmp_path / "checkpoints" / "flow_svhn_best.pth", tmp_path / "checkpoints" / "moved.pth")
    assert sample_mnist_svhn.main(["--num_samples", "2", "--num_steps", "2"]) == 1
    os.rename(tmp_path / "checkpoints" / "moved.pth", tmp_path / "checkpoints" / "flow_svhn_best.pth")

    assert evaluate_mnist_svhn.main(["--guidance_methods", "none", "mc_feng", "--guidance_strengths", "0.0", "1.0",
                                     "--num_samples", "4", "--num_steps", "3", "--mc_batch_size", "5", "--seed", "5"]) == 0
    res = json.load(open(tmp_path / "outputs" / "mnist_svhn" / "evaluation_results.json"))
    assert [(r["method"], r["guidance_strength"]) for r in res] == [("none", 0.0), ("mc_feng", 0.0), ("mc_feng", 1.0)]
    R.utils.set_seed(5)
    want = evaluate_mnist_svhn.run_sweep(fm.to(dev), fs.to(dev), lambda: rr.to(dev), cm.to(dev), cs.to(dev),
                                         ["none", "mc_feng"], [0.0, 1.0], 4, 3, dev, 5)
    assert res == want


def test_cli_mains_sharded_at_world_one(dev, tmp_path, monkeypatch):
    """The `--sharded` launch path of both CLIs (torch.distributed.run environment, init_from_env("nccl"),
    make_sharded_sampler) on the GPU at world size 1: the sharded sampler draws the noise from the CPU generator in the
    reference's order, so the files must equal the explicit-noise API under the same seed bit for bit."""
    import ratio_guided_multimodal_fm_amd as R
    from ratio_guided_multimodal_fm_amd import evaluate_mnist_svhn, sample_mnist_svhn
    from ratio_guided_multimodal_fm_amd.distributed import make_sharded_sampler
    from ratio_guided_multimodal_fm_amd.utils.flow_utils import paired_sampler
    fm, fs, rr, cm, cs = _write_checkpoints(tmp_path)
    monkeypatch.chdir(tmp_path)
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29576")):
        monkeypatch.setenv(k, v)
    assert sample_mnist_svhn.main(["--guidance_method", "mc_feng", "--guidance_strength", "1.0", "--num_samples", "5",
                                   "--num_steps", "4", "--mc_batch_size", "6", "--seed", "11", "--sharded"]) == 0
    saved = torch.load(tmp_path / "outputs" / "mnist_svhn" / "samples_mc_feng_gamma1.0.pt")
    from ratio_guided_multimodal_fm_amd import models as M
    # (the CLI builds its modules between the seeding and the sampling, as the reference's does: their parameter
    # initialisation consumes the CPU generator the sharded sampler then draws the noise from)
    R.utils.set_seed(11)
    M.FlowMatchingUNetMNIST(img_size=32), M.FlowMatchingUNetSVHN(), M.RatioEstimatorMNISTSVHN(loss_type="disc")
    noise = (torch.randn(5, 1, 32, 32), torch.randn(5, 3, 32, 32), torch.randn(6, 1, 32, 32), torch.randn(6, 3, 32, 32))
    xs, ys = paired_sampler(fm.to(dev), fs.to(dev), rr.to(dev), "mc_feng", 1.0, 5, 4, dev, 6, (1, 32, 32), (3, 32, 32),
                            noise=noise, verbose=False)
    assert torch.equal(saved["mnist"], xs.cpu()) and torch.equal(saved["svhn"], ys.cpu())

    assert evaluate_mnist_svhn.main(["--guidance_methods", "none", "mc_feng", "--guidance_strengths", "0.0", "2.0",
                                     "--num_samples", "6", "--num_steps", "3", "--mc_batch_size", "5", "--seed", "6",
                                     "--sharded"]) == 0
    res = json.load(open(tmp_path / "outputs" / "mnist_svhn" / "evaluation_results.json"))
    R.utils.set_seed(6)
    M.MNISTClassifier32(), M.SVHNClassifier(), M.FlowMatchingUNetMNIST(img_size=32), M.FlowMatchingUNetSVHN()

    def make_ratio():  # (as the CLI: a fresh module per request, its initialisation draws from the generator too)
        r = M.RatioEstimatorMNISTSVHN(loss_type="disc").to(dev)
        r.load_state_dict(rr.state_dict())
        return r
    want = evaluate_mnist_svhn.run_sweep(fm.to(dev), fs.to(dev), make_ratio, cm.to(dev), cs.to(dev),
                                         ["none", "mc_feng"], [0.0, 2.0], 6, 3, dev, 5,
                                         sampler=make_sharded_sampler((1, 32, 32), (3, 32, 32), gather="all"))
    assert res == want and len(res) == 3


def test_bench_under_the_distributed_launcher(dev):
    """`bench.py --gpus 1` launched the way the driver launches N > 1 (python -m torch.distributed.run, one rank): the
    RANK / LOCAL_RANK / WORLD_SIZE plumbing, a ONE-rank RCCL group (barrier, the per-rank all_gather, the MAX
    all_reduce), the kernel timers and the parity gate of the timed call, at a size that takes seconds."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29578", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1",
           "--warmup", "1", "--batch-per-gpu", "24", "--mc", "16", "--euler-steps", "6", "--no-cpu-baseline",
           "--no-alt-mode", "--no-arith-check"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["steps"] == 1
    assert line["ranks"] == [{"rank": 0, "world_size_seen": 1, "device": 0, "seconds": line["ranks"][0]["seconds"]}]
    assert line["parity_check"]["rows"] == 8 and line["parity_check"]["max_abs"] <= line["parity_check"]["tolerance"] == 1e-4


def test_cli_sample28_from_checkpoint_files(dev, tmp_path, monkeypatch):
    """src/sample.py twin: checkpoint names from get_checkpoint_path, raw state_dict files (train_flow.py:101)."""
    import ratio_guided_multimodal_fm_amd as R
    from ratio_guided_multimodal_fm_amd import sample as sample28
    from ratio_guided_multimodal_fm_amd.utils.path_utils import get_checkpoint_path
    monkeypatch.chdir(tmp_path)
    fx, fy, rr = make_module("unet28"), make_module("unet28_y"), make_module("ratio28")
    torch.save(fx.state_dict(), get_checkpoint_path("flow", "x", None, "best"))
    torch.save(fy.state_dict(), get_checkpoint_path("flow", "y", "rotate90", "best"))
    torch.save(rr.state_dict(), get_checkpoint_path("ratio", "disc", "rotate90", "best"))
    rc = sample28.main(["--guidance_method", "mc_feng", "--guidance_strength", "0.5", "--num_samples", "3", "--num_steps", "3",
                        "--mc_batch_size", "4", "--seed", "3"])
    assert rc == 0
    outs = list((tmp_path / "outputs").rglob("*.pt"))
    assert len(outs) == 1
    saved = torch.load(outs[0])
    R.utils.set_seed(3)
    xs, ys = R.sample_bimodal_guided(fx.to(dev), fy.to(dev), rr.to(dev), "mc_feng", 0.5, num_samples=3, num_steps=3,
                                     device=dev, mc_batch_size=4)
    vals = list(saved.values())
    assert torch.equal(vals[0], xs.cpu()) and torch.equal(vals[1], ys.cpu())


def _two_rank_worker(rank, world, port, out_dir):
    """One of two processes sharing cuda:0: HIP compute per rank, gloo collectives (RCCL refuses two ranks on one
    device, and the box has one GPU)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler
    dev = torch.device("cuda:0")
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    noise = paired_noise(77, 37, 24, (1, 32, 32), (3, 32, 32))  # uneven shards: 19 + 18 rows, 12 + 12 MC rows
    x, y = sharded_paired_sampler(fm, fs, rr, "mc_feng", 0.5, 12, noise, dev, gather="all")
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=x.cpu().numpy(), y=y.cpu().numpy())
    dist.destroy_process_group()


def test_two_ranks_hip_compute_equal_single_process(dev, tmp_path):
    """world_size 2 with the HIP library as compute backend (two processes on the one GPU, gloo for the two
    collectives): sharded MC pre-phase, all_gather of the MC set, sharded guided loop, all_gather of the outputs --
    bit-identical to the single-process HIP run (rows are independent given the MC set, and a row's arithmetic does
    not depend on how many rows share its launch)."""
    import torch.multiprocessing as mp
    from ratio_guided_multimodal_fm_amd.distributed import sharded_paired_sampler
    mp.spawn(_two_rank_worker, args=(2, 29581, str(tmp_path)), nprocs=2, join=True)
    fm, fs, rr = make_module("mnist32", dev), make_module("svhn", dev), make_module("ratio_ms", dev)
    noise = paired_noise(77, 37, 24, (1, 32, 32), (3, 32, 32))
    x1, y1 = sharded_paired_sampler(fm, fs, rr, "mc_feng", 0.5, 12, noise, dev)
    for r in range(2):
        g = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(g["x"], x1.cpu().numpy()) and np.array_equal(g["y"], y1.cpu().numpy())


@pytest.mark.parametrize("tag", ["mnist32", "svhn"])
def test_rows_do_not_depend_on_the_launch_shape(dev, tag):
    """A row's result is bitwise the same whether its launch is full or under-filled: at 512 rows the convs run as
    two-tile / 128-channel workgroups, at 8 rows (and for the 8x8 level at 512) as four-wave / 64-channel ones
    (conv_mfma_hx2p.hip: hx2p_cfg) -- same arithmetic, same summation order per output element."""
    m = make_module(tag, dev)
    cin = 1 if tag == "mnist32" else 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, cin, 32, 32, generator=g).to(dev)
    t = torch.rand(512, generator=g).to(dev)
    full = m(x, t)
    for lo, hi in ((0, 8), (250, 263), (500, 512)):
        part = m(x[lo:hi].contiguous(), t[lo:hi].contiguous())
        assert torch.equal(part, full[lo:hi]), (tag, lo, hi, float((part - full[lo:hi]).abs().max()))
