"""ctypes binding of csrc/librgfm_hip.so (C ABI: include/rgfm.h).

Loading is lazy and LOUD: there is no fallback implementation, so a missing
or stale library raises with the build command instead of degrading.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RGFM_LIB: another build of the same ABI (A/B measurements of two library versions inside one run)
LIB_PATH = os.environ.get("RGFM_LIB") or os.path.join(_HERE, "csrc", "librgfm_hip.so")
ABI_VERSION = 3

_lib = None

c_void_p, c_int, c_size_t, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_double
c_int32, c_int64 = ctypes.c_int32, ctypes.c_int64
P = ctypes.POINTER


class UNetDesc(ctypes.Structure):
    _fields_ = [("in_channels", c_int32), ("img_size", c_int32), ("model_channels", c_int32),
                ("num_levels", c_int32), ("channel_mult", c_int32 * 4), ("num_res_blocks", c_int32)]


class RatioDesc(ctypes.Structure):
    _fields_ = [("kind", c_int32), ("feature_dim", c_int32), ("hidden_dim", c_int32),
                ("loss_type", c_int32)]


class FmNetDesc(ctypes.Structure):
    _fields_ = [("img_channels", c_int32), ("feature_dim", c_int32), ("time_emb_dim", c_int32)]


# name -> (restype, argtypes); every symbol include/rgfm.h declares.
SIGNATURES = {
    "rgfm_unet_param_floats": (c_int, [P(UNetDesc), P(c_size_t)]),
    "rgfm_unet_create": (c_int, [P(UNetDesc), c_void_p, c_size_t, c_void_p, P(c_void_p)]),
    "rgfm_unet_destroy": (None, [c_void_p]),
    "rgfm_unet_workspace_bytes": (c_int, [c_void_p, c_int, P(c_size_t)]),
    "rgfm_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                  c_size_t, c_void_p]),
    "rgfm_unet_set_trace": (c_int, [c_void_p, c_int]),
    "rgfm_unet_p_handovers": (c_int, [c_void_p, P(c_int)]),
    "rgfm_unet_wino_convs": (c_int, [c_void_p, P(c_int)]),
    "rgfm_unet_time_embedding": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rgfm_unet_num_activations": (c_int, [c_void_p, P(c_int)]),
    "rgfm_unet_activation_shape": (c_int, [c_void_p, c_int, P(c_int), P(c_int), P(c_int)]),
    "rgfm_unet_read_activation": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rgfm_ratio_param_floats": (c_int, [P(RatioDesc), P(c_size_t)]),
    "rgfm_ratio_create": (c_int, [P(RatioDesc), c_void_p, c_size_t, c_void_p, P(c_void_p)]),
    "rgfm_ratio_destroy": (None, [c_void_p]),
    "rgfm_ratio_workspace_bytes": (c_int, [c_void_p, c_int, P(c_size_t)]),
    "rgfm_ratio_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                c_size_t, c_void_p]),
    "rgfm_ratio_grad_workspace_bytes": (c_int, [c_void_p, c_int, P(c_size_t)]),
    "rgfm_ratio_grad_log_ratio": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_size_t, c_void_p]),
    "rgfm_sample_pair_grad_workspace_bytes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, P(c_size_t)]),
    "rgfm_sample_pair_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double,
                                      c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rgfm_sample_single_workspace_bytes": (c_int, [c_void_p, c_int, P(c_size_t)]),
    "rgfm_sample_single": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                   c_size_t, c_void_p]),
    "rgfm_sample_pair_workspace_bytes": (c_int, [c_void_p, c_void_p, c_int, c_int, P(c_size_t)]),
    "rgfm_sample_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                 c_size_t, c_void_p]),
    "rgfm_fmnet_param_floats": (c_int, [P(FmNetDesc), P(c_size_t)]),
    "rgfm_fmnet_create": (c_int, [P(FmNetDesc), c_void_p, c_size_t, c_void_p, P(c_void_p)]),
    "rgfm_fmnet_destroy": (None, [c_void_p]),
    "rgfm_fmnet_workspace_bytes": (c_int, [c_void_p, c_int, P(c_size_t)]),
    "rgfm_fmnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                   c_size_t, c_void_p]),
    "rgfm_fmnet_sample_single": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                         c_size_t, c_void_p]),
    "rgfm_fmnet_sample_pair_workspace_bytes": (c_int, [c_void_p, c_void_p, c_int, c_int, P(c_size_t)]),
    "rgfm_fmnet_sample_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_int, c_double, c_int, c_int, c_void_p,
                                       c_size_t, c_void_p]),
    "rgfm_guidance_workspace_bytes": (c_int, [c_int, c_int, P(c_size_t)]),
    "rgfm_guidance_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_int, c_double, c_double,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "rgfm_profile_enable": (c_int, [c_int]),
    "rgfm_profile_reset": (c_int, []),
    "rgfm_profile_read": (c_int, [c_int, P(c_double), P(c_double), P(c_int64), P(c_double)]),
    "rgfm_profile_reserve": (c_int, [c_int64]),
    "rgfm_ubench_mfma_f16": (c_int, [P(c_double)]),
    "rgfm_ubench_hbm_copy": (c_int, [c_size_t, P(c_double)]),
    "rgfm_unet_set_conv_mode": (c_int, [c_void_p, c_int]),
    "rgfm_fmnet_set_conv_mode": (c_int, [c_void_p, c_int]),
    "rgfm_unet_range_flag": (c_int, [c_void_p, P(c_int), c_int, c_void_p]),
    "rgfm_fmnet_range_flag": (c_int, [c_void_p, P(c_int), c_int, c_void_p]),
    "rgfm_abi_version": (c_int, []),
    "rgfm_last_error": (ctypes.c_char_p, []),
}


class RgfmError(RuntimeError):
    pass


def lib():
    """Return the loaded library, loading it on first use."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RgfmError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or "
            "`make -C ratio_guided_multimodal_fm_amd/csrc`). There is no CPU fallback.")
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library is stale
        fn.restype, fn.argtypes = res, args
    got = handle.rgfm_abi_version()
    if got != ABI_VERSION:
        raise RgfmError(f"librgfm_hip.so ABI {got} != expected {ABI_VERSION}; rebuild it")
    _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().rgfm_last_error()
        raise RgfmError(f"librgfm_hip error {rc}: {msg.decode() if msg else '?'}")
