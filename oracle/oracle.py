"""ctypes wrapper of oracle/librgfm_oracle.so -- the CPU restatement of the
reference sampler path (see oracle/rgfm_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Parity status: pinned against reference-generated golden vectors
(tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librgfm_oracle.so")
_lib = None

F32P = ctypes.POINTER(ctypes.c_float)


class UNetDesc(ctypes.Structure):
    _fields_ = [("in_channels", ctypes.c_int32), ("img_size", ctypes.c_int32),
                ("model_channels", ctypes.c_int32), ("num_levels", ctypes.c_int32),
                ("channel_mult", ctypes.c_int32 * 4), ("num_res_blocks", ctypes.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "rgfm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "librgfm_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.ro_unet_param_floats.restype = ctypes.c_size_t
        L.ro_fm_param_floats.restype = ctypes.c_size_t
        L.ro_fm_forward.restype = None
        L.ro_fm_time_embedding.restype = None
        L.ro_ratio_param_floats.restype = ctypes.c_size_t
        L.ro_unet_num_activations.restype = ctypes.c_int
        L.ro_num_threads.restype = ctypes.c_int
        for f in (L.ro_unet_forward, L.ro_ratio_eval, L.ro_guidance_apply, L.ro_sample_single,
                  L.ro_sample_pair, L.ro_timestep_embedding, L.ro_unet_activation_shape, L.ro_ratio_grad,
                  L.ro_sample_pair_grad):
            f.restype = None
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(F32P)


def unet_desc(in_channels, img_size, model_channels, channel_mult, num_res_blocks=2):
    d = UNetDesc()
    d.in_channels, d.img_size, d.model_channels = in_channels, img_size, model_channels
    d.num_levels = len(channel_mult)
    for i, c in enumerate(channel_mult):
        d.channel_mult[i] = c
    d.num_res_blocks = num_res_blocks
    return d


def desc_of(module):
    """Descriptor from a module exposing the reference attributes."""
    return unet_desc(module.in_channels, module.img_size, module.model_channels,
                     module.channel_mult, module.num_res_blocks)


def blob_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    return np.concatenate([v.detach().cpu().reshape(-1).float().numpy() for v in sd.values()])


def unet_param_floats(desc):
    return lib().ro_unet_param_floats(ctypes.byref(desc))


def num_threads():
    return lib().ro_num_threads()


def timestep_embedding(t, dim):
    t, tp = _f(t)
    out = np.empty((t.shape[0], dim), np.float32)
    lib().ro_timestep_embedding(tp, ctypes.c_int(t.shape[0]), ctypes.c_int(dim),
                                out.ctypes.data_as(F32P))
    return out


def unet_forward(desc, params, x, t, trace=False):
    L = lib()
    params, pp = _f(params)
    assert params.size == L.ro_unet_param_floats(ctypes.byref(desc)), "parameter blob size"
    x, xp = _f(x)
    t, tp = _f(np.atleast_1d(t))
    B = x.shape[0]
    out = np.empty_like(x)
    acts, acts_arg = None, None
    if trace:
        n = L.ro_unet_num_activations(ctypes.byref(desc))
        acts = []
        arr = (F32P * n)()
        for i in range(n):
            c, h, w = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.ro_unet_activation_shape(ctypes.byref(desc), i, ctypes.byref(c), ctypes.byref(h),
                                       ctypes.byref(w))
            a = np.empty((B, c.value, h.value, w.value), np.float32)
            acts.append(a)
            arr[i] = a.ctypes.data_as(F32P)
        acts_arg = arr
    L.ro_unet_forward(ctypes.byref(desc), pp, xp, tp, ctypes.c_int(t.size), out.ctypes.data_as(F32P),
                      ctypes.c_int(B), acts_arg)
    return (out, acts) if trace else out


_KIND = {"mnist_svhn": 0, "mnist28": 1}
_LOSS = {"disc": 0, "rulsif": 1}
_WHAT = {"score": 0, "log_ratio": 1, "ratio": 2}


def ratio_eval(kind, params, x, y, what="log_ratio", loss="disc", feature_dim=256, hidden_dim=512,
               want_feat=False):
    L = lib()
    params, pp = _f(params)
    assert params.size == L.ro_ratio_param_floats(_KIND[kind], feature_dim, hidden_dim)
    x, xp = _f(x)
    y, yp = _f(y)
    n = x.shape[0]
    out = np.empty(n, np.float32)
    feat = np.empty((n, 2 * feature_dim), np.float32) if want_feat else None
    L.ro_ratio_eval(_KIND[kind], feature_dim, hidden_dim, _LOSS[loss], pp, xp, yp,
                    out.ctypes.data_as(F32P), n, _WHAT[what],
                    feat.ctypes.data_as(F32P) if want_feat else None)
    return (out, feat) if want_feat else out


def ratio_grad(params, x, y, loss="disc", feature_dim=256, hidden_dim=512, kind="mnist_svhn"):
    """(d log_ratio / dx, d log_ratio / dy, log_ratio) of RatioEstimatorMNISTSVHN / RatioEstimator ("mnist28") in eval mode."""
    L = lib()
    params, pp = _f(params)
    x, xp = _f(x)
    y, yp = _f(y)
    n = x.shape[0]
    gx, gy, lr = np.empty_like(x), np.empty_like(y), np.empty(n, np.float32)
    L.ro_ratio_grad(_KIND[kind], feature_dim, hidden_dim, _LOSS[loss], pp, xp, yp, gx.ctypes.data_as(F32P),
                    gy.ctypes.data_as(F32P), lr.ctypes.data_as(F32P), n)
    return gx, gy, lr


def sample_pair_grad(desc_x, params_x, desc_y, params_y, ratio_params, x0, y0, num_steps, gamma, loss="disc",
                     step_begin=0, step_end=None, feature_dim=256, hidden_dim=512, kind="mnist_svhn"):
    px, pxp = _f(params_x)
    py, pyp = _f(params_y)
    pr, prp = _f(ratio_params)
    x = np.array(x0, dtype=np.float32, order="C")
    y = np.array(y0, dtype=np.float32, order="C")
    lib().ro_sample_pair_grad(ctypes.byref(desc_x), pxp, ctypes.byref(desc_y), pyp, _KIND[kind], feature_dim, hidden_dim,
                              _LOSS[loss], prp, x.ctypes.data_as(F32P), y.ctypes.data_as(F32P), x.shape[0],
                              num_steps, ctypes.c_double(gamma), step_begin,
                              num_steps if step_end is None else step_end)
    return x, y


def guidance_apply(x, y, vx, vy, mc_x1, mc_y1, mc_ratios, t, gamma, want_weights=False):
    """Returns (vx', vy'[, weights]); inputs are not modified."""
    x, xp = _f(x)
    y, yp = _f(y)
    vx = np.array(vx, dtype=np.float32, order="C")
    vy = np.array(vy, dtype=np.float32, order="C")
    mx, mxp = _f(mc_x1)
    my, myp = _f(mc_y1)
    r, rp = _f(mc_ratios)
    B, N = x.shape[0], mx.shape[0]
    dx, dy = x[0].size, y[0].size
    w = np.empty((B, N), np.float32) if want_weights else None
    lib().ro_guidance_apply(xp, yp, vx.ctypes.data_as(F32P), vy.ctypes.data_as(F32P), mxp, myp, rp,
                            B, N, dx, dy, ctypes.c_double(t), ctypes.c_double(gamma),
                            w.ctypes.data_as(F32P) if want_weights else None)
    return (vx, vy, w) if want_weights else (vx, vy)


def sample_single(desc, params, x0, num_steps, step_begin=0, step_end=None):
    params, pp = _f(params)
    x = np.array(x0, dtype=np.float32, order="C")
    lib().ro_sample_single(ctypes.byref(desc), pp, x.ctypes.data_as(F32P), x.shape[0], num_steps,
                           step_begin, num_steps if step_end is None else step_end)
    return x


def sample_pair(desc_x, params_x, desc_y, params_y, x0, y0, mc_x1, mc_y1, mc_ratios, num_steps,
                gamma, step_begin=0, step_end=None):
    px, pxp = _f(params_x)
    py, pyp = _f(params_y)
    x = np.array(x0, dtype=np.float32, order="C")
    y = np.array(y0, dtype=np.float32, order="C")
    n_mc = 0 if mc_x1 is None else len(mc_x1)
    if n_mc:
        mx, mxp = _f(mc_x1)
        my, myp = _f(mc_y1)
        r, rp = _f(mc_ratios)
    else:
        mxp = myp = rp = None
    lib().ro_sample_pair(ctypes.byref(desc_x), pxp, ctypes.byref(desc_y), pyp,
                         x.ctypes.data_as(F32P), y.ctypes.data_as(F32P), mxp, myp, rp, n_mc,
                         x.shape[0], num_steps, ctypes.c_double(gamma), step_begin,
                         num_steps if step_end is None else step_end)
    return x, y


def paired_sampler(desc_x, params_x, desc_y, params_y, ratio_kind, ratio_params, loss, noise,
                   guided, gamma, num_steps):
    """Whole reference call (pre-phase + ratio + main loop) from explicit noise."""
    x0, y0, mx0, my0 = noise
    mx1 = my1 = r = None
    if guided:
        mx1 = sample_single(desc_x, params_x, mx0, num_steps)
        my1 = sample_single(desc_y, params_y, my0, num_steps)
        r = ratio_eval(ratio_kind, ratio_params, mx1, my1, "ratio", loss)
    x, y = sample_pair(desc_x, params_x, desc_y, params_y, x0, y0, mx1, my1, r, num_steps, gamma)
    return x, y, (mx1, my1, r)


def fm_forward(params, x, t):
    """FlowMatchingModel ('original' net) forward: x [B,1,28,28], t [1] or [B]."""
    L = lib()
    params, pp = _f(params)
    assert params.size == L.ro_fm_param_floats(), "parameter blob size"
    x, xp = _f(x)
    t, tp = _f(np.atleast_1d(t))
    out = np.empty_like(x)
    L.ro_fm_forward(pp, xp, tp, ctypes.c_int(t.size), out.ctypes.data_as(F32P), ctypes.c_int(x.shape[0]))
    return out


def fm_time_embedding(t, dim=128):
    t, tp = _f(t)
    out = np.empty((t.shape[0], dim), np.float32)
    lib().ro_fm_time_embedding(tp, ctypes.c_int(t.shape[0]), ctypes.c_int(dim), out.ctypes.data_as(F32P))
    return out
