"""CPU-only checks of the boundary: the C-ABI library loads and exports exactly
the symbols include/rgfm.h declares; descriptor-only entry points (no GPU work)
agree with the oracle's parameter counts; the product path fails loudly without
a HIP device."""
import ctypes
import os
import re

import pytest
import torch

from oracle import oracle as O
from helpers import make_module
from ratio_guided_multimodal_fm_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "rgfm.h")).read()
    declared = set(re.findall(r"\b(rgfm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    L = _lib.lib()
    for name in _lib.SIGNATURES:
        assert hasattr(L, name), name
    assert L.rgfm_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("tag", ["unet28", "mnist32", "svhn"])
def test_unet_param_count(tag):
    m = make_module(tag)
    d = m._engine.desc()
    n = ctypes.c_size_t()
    assert _lib.lib().rgfm_unet_param_floats(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == sum(v.numel() for v in m.state_dict().values())
    assert n.value == O.unet_param_floats(O.desc_of(m))


@pytest.mark.parametrize("tag", ["ratio_ms", "ratio28"])
def test_ratio_param_count(tag):
    m = make_module(tag)
    d = m._engine.desc()
    n = ctypes.c_size_t()
    assert _lib.lib().rgfm_ratio_param_floats(ctypes.byref(d), ctypes.byref(n)) == 0
    assert n.value == sum(v.numel() for v in m.state_dict().values())


def test_bad_descriptor_is_an_error_code():
    d = _lib.UNetDesc()
    d.in_channels, d.img_size, d.model_channels, d.num_levels, d.num_res_blocks = 2, 32, 32, 2, 2
    n = ctypes.c_size_t()
    rc = _lib.lib().rgfm_unet_param_floats(ctypes.byref(d), ctypes.byref(n))
    assert rc == -1 and b"in_channels" in _lib.lib().rgfm_last_error()


def test_no_cpu_fallback():
    m = make_module("mnist32")
    with pytest.raises(_lib.RgfmError, match="no CPU path|HIP device"):
        m(torch.zeros(1, 1, 32, 32), torch.zeros(1))
    from ratio_guided_multimodal_fm_amd import CFMSchedule
    with pytest.raises(RuntimeError, match="no CPU path"):
        CFMSchedule().sample(m, 2, 2, device="cpu")


def test_state_dict_keys_match_reference_layout():
    """Key order/shape ABI: counts pinned by the reference import in make_golden.py."""
    assert len(make_module("mnist32").state_dict()) == 148
    assert len(make_module("svhn").state_dict()) == 208
    assert len(make_module("ratio_ms").state_dict()) == 102
    sd = make_module("svhn").state_dict()
    assert list(sd)[:6] == ["time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight",
                            "time_embed.2.bias", "input_conv.weight", "input_conv.bias"]
    assert "decoder_blocks.8.skip.weight" in sd and "upsamplers.1.conv.bias" in sd
