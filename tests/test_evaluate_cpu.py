"""the config-5 harness.  The classifiers are ordinary PyTorch modules (they
run once on final samples), so their parity with the reference and the harness logic are checked on CPU."""
import json

import numpy as np
import torch

from helpers import golden, make_module, maxdiff
from ratio_guided_multimodal_fm_amd.evaluate_mnist_svhn import evaluate_coherence, run_sweep


def _golden_samples():
    g = golden("sampler_pair_ms")
    xs = torch.from_numpy(np.concatenate([g[f"c{i}_x"] for i in range(6)]))
    ys = torch.from_numpy(np.concatenate([g[f"c{i}_y"] for i in range(6)]))
    return xs, ys


def test_classifiers_and_coherence_match_reference():
    g = golden("coherence")
    cm, cs = make_module("clf_mnist"), make_module("clf_svhn")
    xs, ys = _golden_samples()
    with torch.no_grad():
        assert maxdiff(cm(xs).numpy(), g["logits_mnist"]) < 1e-5
        assert maxdiff(cs(ys).numpy(), g["logits_svhn"]) < 1e-5
    res = evaluate_coherence(xs, ys, cm, cs, "cpu")
    assert res["num_samples"] == int(g["num_samples"])
    assert abs(res["coherence_acc"] - float(g["coherence_acc"])) < 1e-7


def test_sweep_schema_and_skip_rule():
    """Sweep order, the `none` with gamma>0 skip, and the JSON schema (evaluate_mnist_svhn.py:130-189)."""
    cm, cs = make_module("clf_mnist"), make_module("clf_svhn")
    xs, ys = _golden_samples()
    calls = []

    def fake_sampler(fm_m, fm_s, ratio, method, strength, n, steps, device, mc):
        calls.append((method, strength, ratio is not None, n, steps, mc))
        return xs[:n], ys[:n]

    res = run_sweep(None, None, lambda: object(), cm, cs, ["none", "mc_feng"], [0.0, 0.5, 2.0], 8, 20, "cpu", 16,
                    sampler=fake_sampler)
    assert [(c[0], c[1]) for c in calls] == [("none", 0.0), ("mc_feng", 0.0), ("mc_feng", 0.5), ("mc_feng", 2.0)]
    assert [c[2] for c in calls] == [False, True, True, True]
    assert all(set(r) == {"method", "guidance_strength", "experiment", "coherence_acc", "num_samples"} for r in res)
    assert all(r["experiment"] == "mnist_svhn" and r["num_samples"] == 8 for r in res)
    json.dumps(res)


# ---- 28x28 harness (reference src/evaluate.py, src/sample.py) -----------------------------------
def _golden_samples28():
    g = golden("sampler_pair28")
    xs = torch.from_numpy(np.concatenate([g[f"c{i}_x"] for i in range(3)]))
    ys = torch.from_numpy(np.concatenate([g[f"c{i}_y"] for i in range(3)]))
    return xs, ys


def test_classifier28_and_coherence_match_reference():
    from ratio_guided_multimodal_fm_amd.evaluate import evaluate_coherence as ec28
    g = golden("coherence28")
    clf = make_module("clf_mnist28")
    xs, ys = _golden_samples28()
    with torch.no_grad():
        assert maxdiff(clf(xs).numpy(), g["logits_x"]) < 1e-5
        assert maxdiff(clf(ys).numpy(), g["logits_y"]) < 1e-5
    res = ec28(xs, ys, clf, "cpu", transform_type="none")  # unknown name = identity (evaluate.py:53-54)
    assert res["num_samples"] == int(g["num_samples"])
    assert abs(res["coherence_acc"] - float(g["coherence_acc_identity"])) < 1e-7


def test_inverse_transforms():
    """evaluate.py:31-54: positive angles rotate counter-clockwise; every inverse undoes its transform."""
    from ratio_guided_multimodal_fm_amd.evaluate import get_inverse_transform as inv
    img = torch.arange(2 * 28 * 28, dtype=torch.float32).reshape(2, 1, 28, 28)
    dot = torch.zeros(1, 1, 28, 28)
    dot[0, 0, 0, 27] = 1.0  # top-right
    assert inv("rotate90")(dot)[0, 0, 0, 0] == 1.0        # counter-clockwise: top-right -> top-left
    assert inv("rotate270")(dot)[0, 0, 27, 27] == 1.0     # clockwise: top-right -> bottom-right
    assert inv("rotate180")(dot)[0, 0, 27, 0] == 1.0
    assert torch.equal(inv("rotate270")(inv("rotate90")(img)), img)
    assert torch.equal(inv("rotate180")(inv("rotate180")(img)), img)
    assert torch.equal(inv("flip_h")(img), img.flip(-1)) and torch.equal(inv("flip_v")(img), img.flip(-2))
    assert torch.equal(inv("invert")(img), -img)
    assert torch.equal(inv("something_else")(img), img)


def test_sweep28_schema_and_cli_defaults():
    from ratio_guided_multimodal_fm_amd import evaluate as E, sample as S
    clf = make_module("clf_mnist28")
    xs, ys = _golden_samples28()
    calls = []

    def fake_sampler(fx, fy, ratio, method, strength, n, steps, device, mc):
        calls.append((method, strength, ratio is not None))
        return xs[:n], ys[:n]

    res = E.run_sweep(None, None, lambda: object(), clf, ["none", "mc_feng"], [0.0, 1.0], 6, 10, "cpu", 8, "flip_h",
                      sampler=fake_sampler)
    assert calls == [("none", 0.0, False), ("mc_feng", 0.0, True), ("mc_feng", 1.0, True)]
    assert all(set(r) == {"method", "guidance_strength", "transform_type", "coherence_acc", "num_samples"} for r in res)
    assert all(r["transform_type"] == "flip_h" for r in res)
    json.dumps(res)
    # --model switch of src/sample.py:149-154
    from ratio_guided_multimodal_fm_amd import models as M
    fx, fy = S.build_flow_models("original", "cpu")
    assert isinstance(fx, M.FlowMatchingModel) and isinstance(fy, M.FlowMatchingModel) and fx is not fy
    fx, _ = S.build_flow_models("unet", "cpu")
    assert isinstance(fx, M.FlowMatchingUNet)
    import pytest
    with pytest.raises(ValueError):
        S.build_flow_models("resnet", "cpu")
    if not torch.cuda.is_available():  # no HIP device: the CLIs refuse loudly instead of falling back
        with pytest.raises(RuntimeError):
            S.main([])
        with pytest.raises(RuntimeError):
            E.main([])
