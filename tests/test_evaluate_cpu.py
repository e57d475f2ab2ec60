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
