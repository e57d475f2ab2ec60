#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference.

Run in the build container only (the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

What it does: imports the reference from /root/reference, fills its modules
with this repo's deterministic synthetic parameters (ratio_guided_multimodal_fm_amd/synth.py),
runs the reference's own functions on seeded CPU noise and stores inputs'
fingerprints and expected outputs as small .npz files.  The fixtures are data
only (no reference source).  Reference entry points exercised:

  * src.models.unet_flexible.timestep_embedding            (unet_flexible.py:16-36)
  * FlowMatchingUNet / FlowMatchingUNetMNIST(32) / FlowMatchingUNetSVHN forward
                                                            (unet.py:216-278, unet_flexible.py:203-261)
  * RatioEstimator / RatioEstimatorMNISTSVHN forward, log_ratio
                                                            (ratio_estimator.py:137-191, ratio_flexible.py:347-385)
  * CFMSchedule.sample                                      (flow_utils.py:69-100)
  * sample_bimodal_guided                                   (flow_utils.py:178-375)
  * sample_bimodal_guided_mnist_svhn                        (sample_mnist_svhn.py:39-177)
  * evaluate_coherence, MNISTClassifier32, SVHNClassifier   (evaluate_mnist_svhn.py:28-57, svhn_classifier.py)
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from ratio_guided_multimodal_fm_amd import models as ours  # noqa: E402
from ratio_guided_multimodal_fm_amd.synth import paired_noise, synth_state_dict  # noqa: E402

from src.models.flow_matching import FlowMatchingModel as RefFMOriginal  # noqa: E402
from src.models.ratio_estimator import RatioEstimator as RefRatio28  # noqa: E402
from src.models.ratio_flexible import RatioEstimatorMNISTSVHN as RefRatioMS  # noqa: E402
from src.models.unet import FlowMatchingUNet as RefUNet28  # noqa: E402
from src.models.unet_flexible import (FlowMatchingUNetMNIST as RefUNetMNIST,  # noqa: E402
                                      FlowMatchingUNetSVHN as RefUNetSVHN,
                                      timestep_embedding as ref_timestep_embedding)
from src.evaluate_mnist_svhn import evaluate_coherence as ref_evaluate_coherence  # noqa: E402
from src.models.svhn_classifier import MNISTClassifier32 as RefMNISTClf, SVHNClassifier as RefSVHNClf  # noqa: E402
from src.sample_mnist_svhn import sample_bimodal_guided_mnist_svhn as ref_sample_ms  # noqa: E402
from src.utils.flow_utils import CFMSchedule as RefCFM, sample_bimodal_guided as ref_sample_28  # noqa: E402

torch.set_num_threads(8)

SEED_W = {"unet28": 11, "unet28_y": 12, "mnist32": 13, "svhn": 14, "ratio28": 15, "ratio_ms": 16,
          "clf_mnist": 17, "clf_svhn": 18, "fm_original": 19, "fm_original_y": 20, "clf_mnist28": 21}
N_PROBE = 256


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
        return fn(*a, **k)


def build(ref_cls, our_cls, seed, *args):
    """Reference module + ours with identical synthetic parameters."""
    ref = ref_cls(*args)
    mine = our_cls(*args)
    assert list(ref.state_dict().keys()) == list(mine.state_dict().keys()), ref_cls.__name__
    for (k, a), (_, b) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert a.shape == b.shape, k
    sd = synth_state_dict(mine, seed)
    ref.load_state_dict(sd)
    ref.eval()
    return ref


def probe_idx(numel, salt):
    g = torch.Generator().manual_seed(900 + salt)
    return torch.randint(0, numel, (N_PROBE,), generator=g)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def n(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- embedding
def gen_embedding():
    ts = torch.tensor([0.0, 0.01, 0.5, 0.99], dtype=torch.float32)
    out = {"t": n(ts)}
    for dim in (32, 64):
        out[f"emb{dim}"] = n(ref_timestep_embedding(ts, dim))
    save("timestep_embedding", **out)


# ---------------------------------------------------------------- U-Net per-layer
def unet_trace(ref, x, t):
    """Outputs in production order: input_conv, then every ResBlock /
    Downsample / Upsample module output, then the network output."""
    order = []
    hooks = []

    def hook(name):
        def f(mod, inp, out):
            order.append((name, out.detach().clone()))
        return f

    hooks.append(ref.input_conv.register_forward_hook(hook("input_conv")))
    for name, mod in ref.named_modules():
        cls = type(mod).__name__
        if cls in ("ResBlock", "Downsample", "Upsample"):
            hooks.append(mod.register_forward_hook(hook(name)))
    with torch.no_grad():
        out = ref(x, t)
    for h in hooks:
        h.remove()
    order.append(("output", out))
    return order


def gen_unet_layers():
    cases = [("unet28", RefUNet28, ours.FlowMatchingUNet, (), (1, 28, 28)),
             ("mnist32", RefUNetMNIST, ours.FlowMatchingUNetMNIST, (32,), (1, 32, 32)),
             ("svhn", RefUNetSVHN, ours.FlowMatchingUNetSVHN, (), (3, 32, 32))]
    for tag, rc, oc, args, shape in cases:
        ref = build(rc, oc, SEED_W[tag], *args)
        g = torch.Generator().manual_seed(77)
        x = torch.randn(2, *shape, generator=g)
        out = {}
        for ti, tval in enumerate((0.0, 0.37)):
            t = torch.full((2,), tval)
            tr = unet_trace(ref, x, t)
            names = []
            for li, (name, act) in enumerate(tr):
                names.append(name)
                flat = act.reshape(-1)
                idx = probe_idx(flat.numel(), li)
                out[f"t{ti}_probe_{li}"] = n(flat[idx])
                out[f"t{ti}_stat_{li}"] = np.array([flat.double().mean().item(),
                                                    flat.double().abs().mean().item()])
                out[f"t{ti}_shape_{li}"] = np.array(act.shape)
            out[f"t{ti}_output"] = n(tr[-1][1])
            out["names"] = np.array(names)
        # per-row timesteps (module API allows a different t per row)
        t = torch.tensor([0.1, 0.8])
        with torch.no_grad():
            out["tvec_output"] = n(ref(x, t))
        out["x_fingerprint"] = n(x.reshape(-1)[:8])
        save(f"unet_layers_{tag}", **out)


GENERIC_UNETS = {  # FlexibleUNet shapes outside the three presets: odd tile counts, 4 levels, tiny maps, 3 blocks
    "g24": dict(in_channels=3, img_size=24, model_channels=32, channel_mult=(1, 2, 4), num_res_blocks=1),
    "g16": dict(in_channels=1, img_size=16, model_channels=64, channel_mult=(1, 1, 2, 2), num_res_blocks=3),
    "g40": dict(in_channels=3, img_size=40, model_channels=32, channel_mult=(2, 2), num_res_blocks=2),
}


def gen_unet_generic():
    from src.models.unet_flexible import FlexibleUNet as RefFlex
    out = {}
    for i, (tag, kw) in enumerate(GENERIC_UNETS.items()):
        ref = RefFlex(**kw)
        mine = ours.FlexibleUNet(**kw)
        assert list(ref.state_dict().keys()) == list(mine.state_dict().keys())
        ref.load_state_dict(synth_state_dict(mine, 70 + i))
        ref.eval()
        x = torch.randn(5, kw["in_channels"], kw["img_size"], kw["img_size"], generator=torch.Generator().manual_seed(300 + i))
        t = torch.tensor([0.0, 0.2, 0.5, 0.8, 0.99])
        with torch.no_grad():
            out[f"{tag}_out"] = n(ref(x, t))
        out[f"{tag}_x_fp"] = n(x.reshape(-1)[:8])
    save("unet_generic", **out)


def gen_fp64():
    """The reference nets evaluated in float64 (same weights and inputs, cast up): the yardstick for
    the arithmetic error of the fp32-MFMA and the split-bf16 (bx3) conv paths."""
    cases = [("unet28", RefUNet28, ours.FlowMatchingUNet, (), (1, 28, 28)),
             ("mnist32", RefUNetMNIST, ours.FlowMatchingUNetMNIST, (32,), (1, 32, 32)),
             ("svhn", RefUNetSVHN, ours.FlowMatchingUNetSVHN, (), (3, 32, 32)),
             ("fm_original", RefFMOriginal, ours.FlowMatchingModel, (), (1, 28, 28))]
    out = {}
    for tag, rc, oc, args, shape in cases:
        ref = build(rc, oc, SEED_W[tag], *args)
        x = torch.randn(4, *shape, generator=torch.Generator().manual_seed(91))
        t = torch.tensor([0.05, 0.37, 0.71, 0.99])
        with torch.no_grad():
            o32 = ref(x, t)
            o64 = ref.double()(x.double(), t.double())
        out[f"{tag}_f64"] = o64.numpy()
        out[f"{tag}_ref32_err"] = np.array(float((o32.double() - o64).abs().max()))
        out[f"{tag}_x_fp"] = n(x.reshape(-1)[:8])
    save("fp64_eval", **out)


# ---------------------------------------------------------------- ratio nets
def gen_ratio():
    g = torch.Generator().manual_seed(78)
    x = torch.randn(6, 1, 32, 32, generator=g)
    y = torch.randn(6, 3, 32, 32, generator=g)
    out = {}
    for loss in ("disc", "rulsif"):
        ref = build(RefRatioMS, ours.RatioEstimatorMNISTSVHN, SEED_W["ratio_ms"])
        ref.loss_type = loss
        with torch.no_grad():
            out[f"score_{loss}"] = n(ref(x, y))
            out[f"log_ratio_{loss}"] = n(ref.log_ratio(x, y))
            fm = ref.encoder_mnist(x)
            fs = ref.encoder_svhn(y)
        out["feat_mnist"] = n(fm)
        out["feat_svhn"] = n(fs)
    save("ratio_mnist_svhn", **out)

    x = torch.randn(6, 1, 28, 28, generator=g)
    y = torch.randn(6, 1, 28, 28, generator=g)
    out = {}
    for loss in ("disc", "rulsif"):
        ref = build(RefRatio28, ours.RatioEstimator, SEED_W["ratio28"])
        ref.loss_type = loss
        with torch.no_grad():
            out[f"score_{loss}"] = n(ref(x, y))
            out[f"log_ratio_{loss}"] = n(ref.log_ratio(x, y))
            out["feat_x"] = n(ref.encoder_x(x))
            out["feat_y"] = n(ref.encoder_y(y))
    save("ratio_mnist28", **out)


# ---------------------------------------------------------------- guidance block alone
class ConstVelocity(nn.Module):
    """Stand-in velocity net v(x, t) = a * x, so the reference sampler's output
    isolates its guidance block + Euler update."""

    def __init__(self, a):
        super().__init__()
        self.a = a

    def forward(self, x, t):
        return self.a * x


class FixedRatio(nn.Module):
    def __init__(self, log_r):
        super().__init__()
        self.log_r = log_r

    def log_ratio(self, x, y):
        return self.log_r.clone()


def gen_guidance():
    for tag, fn, sx, sy in (("ms", ref_sample_ms, (1, 32, 32), (3, 32, 32)),
                            ("28", ref_sample_28, (1, 28, 28), (1, 28, 28))):
        B, N = 6, 12
        g = torch.Generator().manual_seed(5)
        log_r = 0.8 * torch.randn(N, generator=g)
        out = {"log_r": n(log_r), "B": B, "N": N}
        for ci, (gamma, steps, a) in enumerate([(0.5, 20, 0.0), (1.0, 20, -0.3), (2.0, 50, 0.2),
                                                (5.0, 20, 0.0), (0.5, 100, -0.5)]):
            seed = 100 + ci
            # the reference draws x0, y0, then mc_x0, mc_y0 from the global generator; with
            # a*x velocity the "MC pre-phase" turns mc_x0 into mc_x0 * prod(1 + a*dt).
            torch.manual_seed(seed)
            xs, ys = quiet(fn, ConstVelocity(a), ConstVelocity(a), FixedRatio(log_r), 'mc_feng',
                           gamma, B, steps, 'cpu', N)
            out[f"c{ci}_cfg"] = np.array([gamma, steps, a, seed], dtype=np.float64)
            out[f"c{ci}_x"] = n(xs)
            out[f"c{ci}_y"] = n(ys)
            x0, y0, mx, my = paired_noise(seed, B, N, sx, sy)
            out[f"c{ci}_x0_fp"] = n(x0.reshape(-1)[:4])
            out[f"c{ci}_mx_fp"] = n(mx.reshape(-1)[:4])
        save(f"guidance_{tag}", **out)


# ---------------------------------------------------------------- full samplers
def mc_prephase(model, noise, steps):
    """Terminal MC samples: the reference pre-phase loop (sample_mnist_svhn.py:89-104),
    driven from here to expose its intermediate result."""
    x = noise.clone()
    dt = 1.0 / steps
    for s in range(steps):
        t = torch.full((x.shape[0],), s * dt)
        with torch.no_grad():
            v = model(x, t)
        x = x + v * dt
    return x


def gen_samplers():
    # config-1 shape: CFMSchedule.sample on the 28x28 U-Net
    ref28 = build(RefUNet28, ours.FlowMatchingUNet, SEED_W["unet28"])
    torch.manual_seed(21)
    xs = quiet(RefCFM().sample, ref28, 4, 20, 'cpu')
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(4, 1, 28, 28, generator=g)
    save("sampler_cfm28", x=n(xs), x0_fp=n(x0.reshape(-1)[:4]), cfg=np.array([4, 20, 21]))

    # 28x28 pair (flow_utils.sample_bimodal_guided)
    ref28y = build(RefUNet28, ours.FlowMatchingUNet, SEED_W["unet28_y"])
    rr28 = build(RefRatio28, ours.RatioEstimator, SEED_W["ratio28"])
    out = {}
    B, N, S = 4, 8, 20
    for ci, (method, gamma) in enumerate([("none", 0.0), ("mc_feng", 0.5), ("mc_feng", 2.0)]):
        seed = 30 + ci
        torch.manual_seed(seed)
        xs, ys = quiet(ref_sample_28, ref28, ref28y, rr28, method, gamma, B, S, 'cpu', N)
        out[f"c{ci}_cfg"] = np.array([method == "mc_feng", gamma, B, N, S, seed], dtype=np.float64)
        out[f"c{ci}_x"], out[f"c{ci}_y"] = n(xs), n(ys)
    save("sampler_pair28", **out)

    # MNIST32 + SVHN pair (sample_mnist_svhn.sample_bimodal_guided_mnist_svhn)
    rm = build(RefUNetMNIST, ours.FlowMatchingUNetMNIST, SEED_W["mnist32"], 32)
    rs = build(RefUNetSVHN, ours.FlowMatchingUNetSVHN, SEED_W["svhn"])
    rr = build(RefRatioMS, ours.RatioEstimatorMNISTSVHN, SEED_W["ratio_ms"])
    out = {}
    cfgs = [("none", 0.0, 4, 8, 20), ("mc_feng", 0.0, 4, 8, 20), ("mc_feng", 0.5, 4, 8, 20),
            ("mc_feng", 1.0, 4, 8, 20), ("mc_feng", 2.0, 4, 8, 20), ("mc_feng", 5.0, 4, 8, 20),
            ("mc_feng", 0.5, 2, 4, 10), ("mc_feng", 0.5, 4, 8, 100)]
    for ci, (method, gamma, B, N, S) in enumerate(cfgs):
        seed = 40 + ci
        torch.manual_seed(seed)
        xs, ys = quiet(ref_sample_ms, rm, rs, rr, method, gamma, B, S, 'cpu', N)
        out[f"c{ci}_cfg"] = np.array([method == "mc_feng", gamma, B, N, S, seed], dtype=np.float64)
        out[f"c{ci}_x"], out[f"c{ci}_y"] = n(xs), n(ys)
        if ci in (2, 6):
            x0, y0, mx0, my0 = paired_noise(seed, B, N, (1, 32, 32), (3, 32, 32))
            mx1 = mc_prephase(rm, mx0, S)
            my1 = mc_prephase(rs, my0, S)
            with torch.no_grad():
                out[f"c{ci}_mc_ratios"] = n(rr.log_ratio(mx1, my1).exp())
            out[f"c{ci}_mc_x1"], out[f"c{ci}_mc_y1"] = n(mx1), n(my1)
        if method == "none":
            # noise-order check: externally drawn noise reproduces the function bitwise
            x0, y0, _, _ = paired_noise(seed, B, 0, (1, 32, 32), (3, 32, 32))
            xr, yr = x0.clone(), y0.clone()
            dt = 1.0 / S
            for s in range(S):
                t = torch.full((B,), s * dt)
                with torch.no_grad():
                    xr = xr + rm(xr, t) * dt
                    yr = yr + rs(yr, t) * dt
            assert torch.equal(xr, xs) and torch.equal(yr, ys), "noise draw order mismatch"
    save("sampler_pair_ms", **out)


def gen_coherence():
    """Classifier logits + coherence accuracy of the reference harness on golden sampler outputs."""
    from ratio_guided_multimodal_fm_amd.models import svhn_classifier as our_clf
    cm = build(RefMNISTClf, our_clf.MNISTClassifier32, SEED_W["clf_mnist"])
    cs = build(RefSVHNClf, our_clf.SVHNClassifier, SEED_W["clf_svhn"])
    g = np.load(os.path.join(HERE, "sampler_pair_ms.npz"))
    xs = torch.from_numpy(np.concatenate([g[f"c{i}_x"] for i in range(6)]))
    ys = torch.from_numpy(np.concatenate([g[f"c{i}_y"] for i in range(6)]))
    with torch.no_grad():
        lm, ls = cm(xs), cs(ys)
    res = ref_evaluate_coherence(xs, ys, cm, cs, 'cpu')
    save("coherence", logits_mnist=n(lm), logits_svhn=n(ls), coherence_acc=res["coherence_acc"],
         num_samples=res["num_samples"])


def gen_sweep200():
    """BASELINE configs[4] in miniature: the reference's guidance-strength sweep (evaluate_mnist_svhn.py:130-183) at
    its 200 Euler steps -- methods x strengths in the reference's loop order, `none` with gamma > 0 skipped, ONE
    seeding before the sweep (:80) and no re-seed between configurations, so configuration i integrates the i-th
    set of draws (x0, y0[, mc_x0, mc_y0]) of one generator stream -- followed by the two classifiers and
    evaluate_coherence on each configuration's samples.  The reference builds a fresh ratio estimator per
    configuration (:141-151); on its own default device ('cuda') that does not touch the generator the noise is
    drawn from, so here the estimator is built before the stream is seeded."""
    from ratio_guided_multimodal_fm_amd.models import svhn_classifier as our_clf
    rm = build(RefUNetMNIST, ours.FlowMatchingUNetMNIST, SEED_W["mnist32"], 32)
    rs = build(RefUNetSVHN, ours.FlowMatchingUNetSVHN, SEED_W["svhn"])
    rr = build(RefRatioMS, ours.RatioEstimatorMNISTSVHN, SEED_W["ratio_ms"])
    cm = build(RefMNISTClf, our_clf.MNISTClassifier32, SEED_W["clf_mnist"])
    cs = build(RefSVHNClf, our_clf.SVHNClassifier, SEED_W["clf_svhn"])
    B, N, S, seed = 3, 6, 200, 50
    methods, strengths = ["none", "mc_feng"], [0.0, 0.5, 1.0, 2.0, 5.0]
    out = {"cfg": np.array([B, N, S, seed]), "strengths": np.array(strengths)}
    torch.manual_seed(seed)
    ci = 0
    for method in methods:
        for gamma in strengths:
            if method == "none" and gamma > 0:
                continue
            xs, ys = quiet(ref_sample_ms, rm, rs, rr if method != "none" else None, method, gamma, B, S, 'cpu', N)
            with torch.no_grad():
                lm, ls = cm(xs), cs(ys)
            res = ref_evaluate_coherence(xs, ys, cm, cs, 'cpu')
            out[f"c{ci}_guided"] = np.array(method == "mc_feng")
            out[f"c{ci}_gamma"] = np.array(gamma)
            out[f"c{ci}_x"], out[f"c{ci}_y"] = n(xs), n(ys)
            out[f"c{ci}_logits_mnist"], out[f"c{ci}_logits_svhn"] = n(lm), n(ls)
            out[f"c{ci}_coherence_acc"] = np.array(res["coherence_acc"])
            print(f"  sweep200 c{ci}: {method} gamma={gamma} coherence={res['coherence_acc']:.3f}", flush=True)
            ci += 1
    out["n_cfg"] = np.array(ci)
    save("sweep200", **out)


def gen_ratio_grad():
    """d log r(x, y) / d(x, y) by torch.autograd on the reference RatioEstimatorMNISTSVHN
    (ratio_flexible.py:347-385) in eval mode, both loss types, at noise-like and at image-like inputs."""
    out = {}
    g = torch.Generator().manual_seed(85)
    x = torch.randn(3, 1, 32, 32, generator=g)
    y = torch.randn(3, 3, 32, 32, generator=g)
    out["x_fp"], out["y_fp"] = n(x.reshape(-1)[:8]), n(y.reshape(-1)[:8])
    for loss in ("disc", "rulsif"):
        ref = build(RefRatioMS, ours.RatioEstimatorMNISTSVHN, SEED_W["ratio_ms"])
        ref.loss_type = loss
        for tag, scale in (("n", 1.0), ("s", 0.3)):
            xx = (x * scale).clone().requires_grad_(True)
            yy = (y * scale).clone().requires_grad_(True)
            lr = ref.log_ratio(xx, yy)
            gx, gy = torch.autograd.grad(lr.sum(), (xx, yy))
            out[f"{loss}_{tag}_lr"], out[f"{loss}_{tag}_gx"], out[f"{loss}_{tag}_gy"] = n(lr.detach()), n(gx), n(gy)
    save("ratio_grad", **out)


def gen_ratio_grad28():
    """The same for the 28x28 RatioEstimator (GroupNorm encoders, ratio_estimator.py:34-191): torch.autograd on the
    reference module in eval mode, both loss types, noise-like and image-like inputs."""
    out = {}
    g = torch.Generator().manual_seed(86)
    x = torch.randn(3, 1, 28, 28, generator=g)
    y = torch.randn(3, 1, 28, 28, generator=g)
    out["x_fp"], out["y_fp"] = n(x.reshape(-1)[:8]), n(y.reshape(-1)[:8])
    for loss in ("disc", "rulsif"):
        ref = build(RefRatio28, ours.RatioEstimator, SEED_W["ratio28"])
        ref.loss_type = loss
        for tag, scale in (("n", 1.0), ("s", 0.3)):
            xx = (x * scale).clone().requires_grad_(True)
            yy = (y * scale).clone().requires_grad_(True)
            lr = ref.log_ratio(xx, yy)
            gx, gy = torch.autograd.grad(lr.sum(), (xx, yy))
            out[f"{loss}_{tag}_lr"], out[f"{loss}_{tag}_gx"], out[f"{loss}_{tag}_gy"] = n(lr.detach()), n(gx), n(gy)
    save("ratio_grad28", **out)


def gen_coherence28():
    """MNISTClassifier (src/models/classifier.py) logits on the golden 28x28 pairs.  src/evaluate.py itself
    cannot be imported here (it needs torchvision), so the coherence value is its :84-91 restated on the
    reference classifier's own logits for the identity transform."""
    from src.models.classifier import MNISTClassifier as RefMNISTClf28
    from ratio_guided_multimodal_fm_amd.models.classifier import MNISTClassifier as OurClf28
    clf = build(RefMNISTClf28, OurClf28, SEED_W["clf_mnist28"])
    g = np.load(os.path.join(HERE, "sampler_pair28.npz"))
    xs = torch.from_numpy(np.concatenate([g[f"c{i}_x"] for i in range(3)]))
    ys = torch.from_numpy(np.concatenate([g[f"c{i}_y"] for i in range(3)]))
    with torch.no_grad():
        lx, ly = clf(xs), clf(ys)
    acc = float((lx.argmax(1).numpy() == ly.argmax(1).numpy()).mean())
    save("coherence28", logits_x=n(lx), logits_y=n(ly), coherence_acc_identity=acc, num_samples=len(xs))


def gen_fm_original():
    """FlowMatchingModel (--model original): forward at shared and per-row t, its time embedding, and
    CFMSchedule.sample driven with it (flow_utils.py:69-100)."""
    ref = build(RefFMOriginal, ours.FlowMatchingModel, SEED_W["fm_original"])
    g = torch.Generator().manual_seed(79)
    x = torch.randn(3, 1, 28, 28, generator=g)
    out = {"x_fingerprint": n(x.reshape(-1)[:8])}
    with torch.no_grad():
        out["v_t037"] = n(ref(x, torch.full((3,), 0.37)))
        out["v_tvec"] = n(ref(x, torch.tensor([0.0, 0.5, 0.99])))
        out["temb"] = n(ref.time_embed(torch.tensor([0.0, 0.01, 0.5, 0.99])))
    torch.manual_seed(22)
    out["sample"] = n(quiet(RefCFM().sample, ref, 3, 10, 'cpu'))
    # src/sample.py --model original: two FlowMatchingModel nets + the 28x28 ratio estimator
    refy = build(RefFMOriginal, ours.FlowMatchingModel, SEED_W["fm_original_y"])
    rr28 = build(RefRatio28, ours.RatioEstimator, SEED_W["ratio28"])
    B, N, S = 5, 8, 10
    for ci, (method, gamma) in enumerate([("none", 0.0), ("mc_feng", 0.5), ("mc_feng", 2.0)]):
        seed = 60 + ci
        torch.manual_seed(seed)
        xs, ys = quiet(ref_sample_28, ref, refy, rr28, method, gamma, B, S, 'cpu', N)
        out[f"c{ci}_cfg"] = np.array([method == "mc_feng", gamma, B, N, S, seed], dtype=np.float64)
        out[f"c{ci}_x"], out[f"c{ci}_y"] = n(xs), n(ys)
    save("fm_original", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["embedding", "unet_layers", "ratio", "guidance", "samplers", "coherence", "fm_original", "coherence28", "fp64", "unet_generic", "sweep200", "ratio_grad", "ratio_grad28"]
    for w in which:
        {"embedding": gen_embedding, "unet_layers": gen_unet_layers, "ratio": gen_ratio,
         "guidance": gen_guidance, "samplers": gen_samplers, "coherence": gen_coherence,
         "fm_original": gen_fm_original, "coherence28": gen_coherence28, "fp64": gen_fp64, "unet_generic": gen_unet_generic,
         "sweep200": gen_sweep200, "ratio_grad": gen_ratio_grad, "ratio_grad28": gen_ratio_grad28}[w]()
