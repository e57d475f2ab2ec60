"""SURVEY 8(f) row 2 on CPU: checkpoint ingestion and naming are host logic (no device needed).

Reference: ``load_checkpoint`` src/utils/__init__.py:25-51 (dict with 'model_state_dict' or a raw state_dict),
``get_checkpoint_path`` src/utils/path_utils.py:7-32, the files the trainers write
(train_flow_mnist32.py:137-143, train_flow_svhn.py:164-170 dict form; train_flow.py:101 raw form)."""
import copy
import os

import pytest
import torch

from helpers import make_module
from ratio_guided_multimodal_fm_amd import models as M
from ratio_guided_multimodal_fm_amd.utils import load_checkpoint
from ratio_guided_multimodal_fm_amd.utils.path_utils import get_checkpoint_path


def _same(a, b):
    sa, sb = a.state_dict(), b.state_dict()
    return list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)


@pytest.mark.parametrize("tag,ctor", [("mnist32", lambda: M.FlowMatchingUNetMNIST(32)), ("svhn", M.FlowMatchingUNetSVHN),
                                      ("ratio_ms", M.RatioEstimatorMNISTSVHN), ("fm_original", M.FlowMatchingModel)])
def test_both_checkpoint_formats(tmp_path, tag, ctor):
    src = make_module(tag)
    dict_path, raw_path = tmp_path / "dict.pth", tmp_path / "raw.pth"
    torch.save({"epoch": 12, "model_state_dict": src.state_dict(), "optimizer_state_dict": {"state": {}}, "best_loss": 0.125},
               dict_path)
    torch.save(src.state_dict(), raw_path)
    a, b = ctor(), ctor()
    assert load_checkpoint(a, str(dict_path), "cpu") == {"epoch": 12, "best_loss": 0.125}
    assert load_checkpoint(b, str(raw_path), "cpu") == {}
    assert _same(a, src) and _same(b, src)
    # a dict without the optional fields gets the reference's defaults (:44-47)
    torch.save({"model_state_dict": src.state_dict()}, dict_path)
    info = load_checkpoint(ctor(), str(dict_path), "cpu")
    assert info["epoch"] == 0 and info["best_loss"] == float("inf")


def test_wrong_architecture_raises(tmp_path):
    p = tmp_path / "m.pth"
    torch.save(make_module("mnist32").state_dict(), p)
    with pytest.raises(RuntimeError):
        load_checkpoint(M.FlowMatchingUNetSVHN(), str(p), "cpu")


def test_checkpoint_path_naming(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    assert get_checkpoint_path("flow", "x", None, "best") == os.path.join("checkpoints", "flow_x_best.pth")
    assert os.path.isdir(tmp_path / "checkpoints")  # created, as the reference does (:24)
    assert get_checkpoint_path("flow", "y", "rotate90", "best") == os.path.join("checkpoints", "flow_y_rotate90_best.pth")
    assert get_checkpoint_path("ratio", "disc", "rotate90", "best") == os.path.join("checkpoints", "ratio_disc_rotate90_best.pth")
    assert get_checkpoint_path("classifier", None, "best") == os.path.join("checkpoints", "classifier_best.pth")


def test_modules_deepcopy_and_pickle_roundtrip(tmp_path):
    """The per-module engine is a cache, not state: copies and whole-module pickles get their own, lazily."""
    m = make_module("mnist32")
    e = m._engine
    c = copy.deepcopy(m)
    assert c._engine is not e and c._engine._module() is c and _same(c, m)
    torch.save(m, tmp_path / "whole.pt")
    r = torch.load(tmp_path / "whole.pt", weights_only=False)
    assert r._engine is not e and r._engine._module() is r and _same(r, m)
