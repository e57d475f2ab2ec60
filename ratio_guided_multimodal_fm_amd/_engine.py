"""Host plumbing between the nn.Module parameter containers and the C ABI.

PyTorch is used for what it is good at here: owning device memory (parameter
blob, workspace, I/O tensors) and naming the stream.  All arithmetic happens
inside librgfm_hip.so.
"""
import ctypes
import os
import weakref

import torch

from . import _lib

_LOSS = {"disc": 0, "rulsif": 1}
_RATIO_KIND = {"mnist_svhn": 0, "mnist28": 1}
_RATIO_OUT = {"score": 0, "log_ratio": 1, "ratio": 2}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.RgfmError(
                "this package runs only on a HIP device (MI355X): got a tensor on "
                f"'{t.device}'. There is no CPU path; move the model and inputs to 'cuda'.")
        if t.dtype != torch.float32:
            raise _lib.RgfmError(f"fp32 tensors expected, got {t.dtype}")


class _Workspace:
    """Grow-only byte buffer per device."""

    def __init__(self):
        self.buf = None
        self.streams = set()

    def get(self, nbytes, device):
        cur = torch.cuda.current_stream(device)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            old = self.buf
            self.buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
            if old is not None:
                # work already enqueued on other streams may still use the old buffer: keep the
                # caching allocator from handing its memory out before those streams get there
                for st in self.streams:
                    old.record_stream(st)
            self.streams = set()
        self.streams.add(cur)
        return self.buf


def engine_property(factory):
    """Class attribute `_engine = engine_property(factory)`: the module's engine, built on first use.

    The engine is a per-module cache (device handle, packed weights, workspace) kept out of
    state_dict, deepcopy and pickling: a copied / unpickled module simply builds its own.
    """
    def get(module):
        e = module.__dict__.get("_engine_obj")
        if e is None or e._module() is not module:
            e = factory(module)
            module.__dict__["_engine_obj"] = e
        return e
    return property(get)


# ---- range guard of the default fp16 conv path -------------------------------------------------
# The fp16 two-plane convs (conv_mfma_hx2*.hip) emulate fp32 products inside a window: activations
# below 2048, and residual-stream tensors (consumed without a GroupNorm in front) not smaller than
# 2^-8 per 64-pixel block.  Outside it a launch raises the range flag of ITS HANDLE instead of
# returning degraded numbers.  Every public compute entry point below runs under _range_guarded:
# if a flag of one of the engines involved is up after the call, the in-place state is restored
# and the call is repeated with those handles switched to the split-bf16 convs (too large: fp32 exponent
# range) or the exact fp32 matrix-core convs (too small) -- rgfm_*_set_conv_mode, a handle setting,
# nothing process-wide is touched.  RGFM_RANGE_CHECK=0
# disables the check (and its end-of-call synchronisation), e.g. for stream capture.
CONV_DEFAULT, CONV_HX2, CONV_BX3, CONV_F32 = -1, 0, 1, 2


def _hx2_active():
    return os.environ.get("RGFM_CONV", "hx2") == "hx2" and os.environ.get("RGFM_RANGE_CHECK", "1") != "0"


range_fallbacks = 0  # number of calls repeated on the bf16 path (tests read it)
last_range_flags = 0  # OR of the flag bits that caused the latest fallback (1: too large, 2: too small)


def _range_guarded(device, state, fn, engines):
    """fn() with the fp16-range check; `state` = tensors fn updates in place, `engines` = the
    velocity-net engines whose handles fn launches on."""
    global range_fallbacks, last_range_flags
    engines = [e for i, e in enumerate(engines) if e is not None and all(e is not o for o in engines[:i])]
    if not _hx2_active() or not engines:
        return fn()
    saved = [t.clone() for t in state]
    out = fn()
    bits = 0
    with torch.cuda.device(device):
        for e in engines:
            bits |= e.read_range_flag(device)
    if bits:
        range_fallbacks += 1
        last_range_flags = bits
        for t, t0 in zip(state, saved):
            t.copy_(t0)
        # too large (bit 0): the split-bf16 convs have fp32's exponent range at the top; too small (bit 1): the repeat
        # runs on the exact fp32 matrix-core convs, which ARE the reference's arithmetic at any magnitude
        fallback = CONV_F32 if bits & 2 else CONV_BX3
        with torch.cuda.device(device):
            for e in engines:
                e.set_conv_mode(device, fallback)
            try:
                out = fn()
            finally:
                for e in engines:
                    e.set_conv_mode(device, CONV_DEFAULT)
    return out


class _EngineBase:
    def __init__(self, module):
        self._module = weakref.ref(module)
        self._handle = None
        self._key = None
        self._blob = None
        self._ws = _Workspace()

    # engines are per-module caches: never copied or pickled with the module (engine_property rebuilds them)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    def _state_key(self, sd):
        return tuple((k, v.data_ptr(), v._version, str(v.device)) for k, v in sd.items())

    def _destroy(self):
        raise NotImplementedError

    def __del__(self):
        try:
            if self._handle is not None:
                self._destroy()
        except Exception:
            pass

    def _blob_from(self, sd, device):
        parts = [v.detach().reshape(-1).to(device=device, dtype=torch.float32) for v in sd.values()]
        return torch.cat(parts).contiguous()

    def _check_eval(self, module):
        if module.training:
            raise _lib.RgfmError(
                f"{type(module).__name__} is in training mode; the HIP path implements eval-mode "
                "semantics only (Dropout = identity, BatchNorm = running statistics). Call .eval().")


class _VelocityEngine(_EngineBase):
    """Shared handle cache / forward of the velocity nets; subclasses name the ABI family."""
    PREFIX = None          # rgfm_<family>_{param_floats,create,destroy,workspace_bytes,forward}
    SINGLE = SINGLE_WS = PAIR = PAIR_WS = None

    def desc(self):
        raise NotImplementedError

    def _check_input(self, m, x):
        raise NotImplementedError

    def _fn(self, name):
        return getattr(_lib.lib(), f"{self.PREFIX}_{name}")

    def handle(self, device):
        m = self._module()
        sd = m.state_dict()
        key = self._state_key(sd)
        if self._handle is not None and key == self._key:
            return self._handle
        if self._handle is not None:
            self._destroy()
        d = self.desc()
        n = ctypes.c_size_t()
        _lib.check(self._fn("param_floats")(ctypes.byref(d), ctypes.byref(n)))
        blob = self._blob_from(sd, device)
        if blob.numel() != n.value:
            raise _lib.RgfmError(f"parameter blob has {blob.numel()} floats, library expects {n.value}")
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self._fn("create")(ctypes.byref(d), _ptr(blob), blob.numel(), _stream(device),
                                          ctypes.byref(h)))
        self._handle, self._key, self._blob = h, key, blob
        return h

    def _destroy(self):
        self._fn("destroy")(self._handle)
        self._handle = None

    def workspace(self, fn_name, batch, device):
        L = _lib.lib()
        n = ctypes.c_size_t()
        _lib.check(getattr(L, fn_name)(self.handle(device), int(batch), ctypes.byref(n)))
        return self._ws.get(n.value, device), n.value

    def read_range_flag(self, device, reset=True):
        """Flag bits raised by this engine's launches since the last reset (waits for the current stream)."""
        flag = ctypes.c_int()
        _lib.check(self._fn("range_flag")(self.handle(device), ctypes.byref(flag), 1 if reset else 0, _stream(device)))
        return flag.value

    def set_conv_mode(self, device, mode):
        """Conv arithmetic of this engine's handle: CONV_DEFAULT (RGFM_CONV), CONV_HX2, CONV_BX3, CONV_F32."""
        _lib.check(self._fn("set_conv_mode")(self.handle(device), int(mode)))

    def forward(self, x, t):
        m = self._module()
        self._check_eval(m)
        _require_hip(x, t)
        self._check_input(m, x)
        B = x.shape[0]
        t = t.reshape(-1)
        if t.numel() not in (1, B):
            raise _lib.RgfmError(f"t must have 1 or {B} elements, got {t.numel()}")
        x = x.contiguous()
        t = t.contiguous()
        out = torch.empty_like(x)
        if B == 0:
            return out
        dev = x.device

        def run():
            with torch.cuda.device(dev):
                h = self.handle(dev)
                ws, nb = self.workspace(f"{self.PREFIX}_workspace_bytes", B, dev)
                _lib.check(self._fn("forward")(h, _ptr(x), _ptr(t), t.numel(), _ptr(out), B, _ptr(ws), nb,
                                               _stream(dev)))
            return out
        return _range_guarded(dev, [], run, [self])


class FmNetEngine(_VelocityEngine):
    """FlowMatchingModel ('--model original', reference src/models/flow_matching.py:127-173)."""
    PREFIX = "rgfm_fmnet"
    SINGLE, SINGLE_WS = "rgfm_fmnet_sample_single", "rgfm_fmnet_workspace_bytes"
    PAIR, PAIR_WS = "rgfm_fmnet_sample_pair", "rgfm_fmnet_sample_pair_workspace_bytes"

    def desc(self):
        m = self._module()
        d = _lib.FmNetDesc()
        d.img_channels, d.feature_dim, d.time_emb_dim = m.img_channels, m.feature_dim, m.time_emb_dim
        return d

    def _check_input(self, m, x):
        if x.dim() != 4 or tuple(x.shape[1:]) != (m.img_channels, 28, 28):
            raise _lib.RgfmError(f"expected x of shape [B,{m.img_channels},28,28], got {tuple(x.shape)}")


class UNetEngine(_VelocityEngine):
    PREFIX = "rgfm_unet"
    SINGLE, SINGLE_WS = "rgfm_sample_single", "rgfm_sample_single_workspace_bytes"
    PAIR, PAIR_WS = "rgfm_sample_pair", "rgfm_sample_pair_workspace_bytes"

    def _check_input(self, m, x):
        if x.dim() != 4 or x.shape[1] != m.in_channels or x.shape[2] != m.img_size or x.shape[3] != m.img_size:
            raise _lib.RgfmError(f"expected x of shape [B,{m.in_channels},{m.img_size},{m.img_size}], got {tuple(x.shape)}")

    def desc(self):
        m = self._module()
        d = _lib.UNetDesc()
        d.in_channels, d.img_size, d.model_channels = m.in_channels, m.img_size, m.model_channels
        d.num_levels = len(m.channel_mult)
        if d.num_levels > 4:
            raise _lib.RgfmError("at most 4 resolution levels are supported")
        for i, c in enumerate(m.channel_mult):
            d.channel_mult[i] = c
        d.num_res_blocks = m.num_res_blocks
        return d

    # ---- parity hooks -------------------------------------------------
    def time_embedding(self, t):
        """timestep_embedding(t, model_channels) as the device evaluates it (rgfm_unet_time_embedding)."""
        _require_hip(t)
        t = t.reshape(-1).contiguous()
        dev = t.device
        out = torch.empty(t.numel(), self._module().model_channels, device=dev)
        with torch.cuda.device(dev):
            h = self.handle(dev)
            ws, nb = self.workspace("rgfm_unet_workspace_bytes", t.numel(), dev)
            _lib.check(_lib.lib().rgfm_unet_time_embedding(h, _ptr(t), t.numel(), _ptr(out), _ptr(ws), nb, _stream(dev)))
        return out

    def forward_trace(self, x, t):
        """Forward in trace mode; returns (out, [activation tensors, NCHW])."""
        dev = x.device
        L = _lib.lib()
        with torch.cuda.device(dev):
            h = self.handle(dev)
            _lib.check(L.rgfm_unet_set_trace(h, 1))
            try:
                out = self.forward(x, t)
                ws, _ = self.workspace("rgfm_unet_workspace_bytes", x.shape[0], dev)
                n = ctypes.c_int()
                _lib.check(L.rgfm_unet_num_activations(h, ctypes.byref(n)))
                acts = []
                for i in range(n.value):
                    c, hh, ww = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                    _lib.check(L.rgfm_unet_activation_shape(h, i, ctypes.byref(c), ctypes.byref(hh),
                                                            ctypes.byref(ww)))
                    a = torch.empty(x.shape[0], c.value, hh.value, ww.value, device=dev)
                    _lib.check(L.rgfm_unet_read_activation(h, i, x.shape[0], _ptr(ws), _ptr(a),
                                                           _stream(dev)))
                    acts.append(a)
            finally:
                _lib.check(L.rgfm_unet_set_trace(h, 0))
        return out, acts


class RatioEngine(_EngineBase):
    def __init__(self, module, kind):
        super().__init__(module)
        self.kind = kind

    def desc(self):
        m = self._module()
        d = _lib.RatioDesc()
        d.kind = _RATIO_KIND[self.kind]
        d.feature_dim, d.hidden_dim = m.feature_dim, m.hidden_dim
        d.loss_type = _LOSS.get(m.loss_type, 0)
        return d

    def handle(self, device):
        m = self._module()
        sd = m.state_dict()
        key = self._state_key(sd) + (m.loss_type,)
        if self._handle is not None and key == self._key:
            return self._handle
        L = _lib.lib()
        if self._handle is not None:
            self._destroy()
        d = self.desc()
        n = ctypes.c_size_t()
        _lib.check(L.rgfm_ratio_param_floats(ctypes.byref(d), ctypes.byref(n)))
        blob = self._blob_from(sd, device)
        if blob.numel() != n.value:
            raise _lib.RgfmError(f"parameter blob has {blob.numel()} floats, library expects {n.value}")
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.rgfm_ratio_create(ctypes.byref(d), _ptr(blob), blob.numel(), _stream(device),
                                           ctypes.byref(h)))
        self._handle, self._key, self._blob = h, key, blob
        return h

    def _destroy(self):
        _lib.lib().rgfm_ratio_destroy(self._handle)
        self._handle = None

    def eval(self, x, y, what):
        m = self._module()
        self._check_eval(m)
        _require_hip(x, y)
        if x.shape[0] != y.shape[0]:
            raise _lib.RgfmError("x and y must have the same batch size")
        n = x.shape[0]
        x, y = x.contiguous(), y.contiguous()
        out = torch.empty(n, device=x.device, dtype=torch.float32)
        if n == 0:
            return out
        dev = x.device
        L = _lib.lib()
        with torch.cuda.device(dev):
            h = self.handle(dev)
            nb = ctypes.c_size_t()
            _lib.check(L.rgfm_ratio_workspace_bytes(h, n, ctypes.byref(nb)))
            ws = self._ws.get(nb.value, dev)
            _lib.check(L.rgfm_ratio_eval(h, _ptr(x), _ptr(y), _ptr(out), n, _RATIO_OUT[what],
                                         _ptr(ws), nb.value, _stream(dev)))
        return out


    def grad_log_ratio(self, x, y):
        """(d log_ratio/dx, d log_ratio/dy, log_ratio): rgfm_ratio_grad_log_ratio (either estimator)."""
        m = self._module()
        self._check_eval(m)
        _require_hip(x, y)
        if x.shape[0] != y.shape[0]:
            raise _lib.RgfmError("x and y must have the same batch size")
        sx, sy = ((1, 32, 32), (3, 32, 32)) if self.kind == "mnist_svhn" else ((1, 28, 28), (1, 28, 28))
        if x.dim() != 4 or y.dim() != 4 or tuple(x.shape[1:]) != sx or tuple(y.shape[1:]) != sy:
            raise _lib.RgfmError(f"expected x of shape [B,{sx[0]},{sx[1]},{sx[2]}] and y of shape [B,{sy[0]},{sy[1]},{sy[2]}], got "
                                 f"{tuple(x.shape)} and {tuple(y.shape)}")
        n = x.shape[0]
        x, y = x.contiguous(), y.contiguous()
        gx, gy = torch.empty_like(x), torch.empty_like(y)
        lr = torch.empty(n, device=x.device, dtype=torch.float32)
        if n == 0:
            return gx, gy, lr
        dev = x.device
        L = _lib.lib()
        with torch.cuda.device(dev):
            h = self.handle(dev)
            nb = ctypes.c_size_t()
            _lib.check(L.rgfm_ratio_grad_workspace_bytes(h, n, ctypes.byref(nb)))
            ws = self._ws.get(nb.value, dev)
            _lib.check(L.rgfm_ratio_grad_log_ratio(h, _ptr(x), _ptr(y), _ptr(gx), _ptr(gy), _ptr(lr), n, _ptr(ws),
                                                   nb.value, _stream(dev)))
        return gx, gy, lr


# ---- sampler entry points ------------------------------------------------

_sampler_ws = _Workspace()


def sample_single(model, x, num_steps, step_begin=0, step_end=None):
    """In-place unguided Euler integration of `x` (rgfm_sample_single)."""
    if x.is_cuda and x.shape[0]:
        return _range_guarded(x.device, [x], lambda: _sample_single(model, x, num_steps, step_begin, step_end),
                              [model._engine])
    return _sample_single(model, x, num_steps, step_begin, step_end)


def _sample_single(model, x, num_steps, step_begin=0, step_end=None):
    eng = model._engine
    eng._check_eval(model)
    _require_hip(x)
    if not x.is_contiguous():
        raise _lib.RgfmError("x must be contiguous (it is updated in place)")
    if step_end is None:
        step_end = num_steps
    B, dev = x.shape[0], x.device
    if B == 0:
        return x
    with torch.cuda.device(dev):
        h = eng.handle(dev)
        ws, nb = eng.workspace(eng.SINGLE_WS, B, dev)
        _lib.check(getattr(_lib.lib(), eng.SINGLE)(h, _ptr(x), B, int(num_steps), int(step_begin),
                                                   int(step_end), _ptr(ws), nb, _stream(dev)))
    return x


_side_streams = {}


def sample_two_streams(fm_x, x, fm_y, y, num_steps):
    """Two independent unguided integrations (the MC pre-phase) on two HIP streams.

    Same arithmetic as two sample_single calls; the second net runs on a side stream that
    forks from / joins back into the current stream, so callers keep stream-ordered semantics.
    """
    dev = x.device

    def run():
        # one module passed for both modalities (legal in the reference) has ONE engine workspace:
        # its two integrations must not run concurrently
        if os.environ.get("RGFM_OVERLAP", "1") == "0" or fm_x._engine is fm_y._engine:
            _sample_single(fm_x, x, num_steps)
            _sample_single(fm_y, y, num_steps)
            return x, y
        cur = torch.cuda.current_stream(dev)
        side = _side_streams.get(dev)
        if side is None:
            side = _side_streams[dev] = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            _sample_single(fm_y, y, num_steps)
        _sample_single(fm_x, x, num_steps)
        cur.wait_stream(side)
        return x, y
    if x.is_cuda and y.is_cuda:
        return _range_guarded(dev, [x, y], run, [fm_x._engine, fm_y._engine])
    return run()


def sample_pair(fm_x, fm_y, x, y, mc_x1, mc_y1, mc_ratios, num_steps, gamma, step_begin=0,
                step_end=None):
    """In-place paired Euler loop with optional MC guidance (rgfm_sample_pair)."""
    for m in (fm_x, fm_y):
        m._engine._check_eval(m)
    ex = fm_x._engine
    if type(ex) is not type(fm_y._engine):
        raise _lib.RgfmError("paired sampling needs two velocity nets of the same family "
                             f"(got {type(fm_x).__name__} and {type(fm_y).__name__})")
    _require_hip(x, y, mc_x1, mc_y1, mc_ratios)
    if not (x.is_contiguous() and y.is_contiguous()):
        raise _lib.RgfmError("x and y must be contiguous (they are updated in place)")
    if step_end is None:
        step_end = num_steps
    B, dev = x.shape[0], x.device
    if B == 0:
        return x, y
    n_mc = 0 if mc_x1 is None else mc_x1.shape[0]
    if n_mc:
        mc_x1, mc_y1, mc_ratios = mc_x1.contiguous(), mc_y1.contiguous(), mc_ratios.contiguous()
    L = _lib.lib()

    def run():
        with torch.cuda.device(dev):
            hx, hy = fm_x._engine.handle(dev), fm_y._engine.handle(dev)
            nb = ctypes.c_size_t()
            _lib.check(getattr(L, ex.PAIR_WS)(hx, hy, B, n_mc, ctypes.byref(nb)))
            ws = _sampler_ws.get(nb.value, dev)
            _lib.check(getattr(L, ex.PAIR)(hx, hy, _ptr(x), _ptr(y), _ptr(mc_x1 if n_mc else None),
                                          _ptr(mc_y1 if n_mc else None),
                                          _ptr(mc_ratios if n_mc else None), n_mc, B, int(num_steps),
                                          float(gamma), int(step_begin), int(step_end), _ptr(ws),
                                          nb.value, _stream(dev)))
        return x, y
    return _range_guarded(dev, [x, y], run, [fm_x._engine, fm_y._engine])


def sample_pair_grad(fm_x, fm_y, ratio_estimator, x, y, num_steps, gamma, step_begin=0, step_end=None):
    """In-place paired Euler loop with gradient log-ratio guidance (rgfm_sample_pair_grad)."""
    for m in (fm_x, fm_y, ratio_estimator):
        m._engine._check_eval(m)
    if not (isinstance(fm_x._engine, UNetEngine) and isinstance(fm_y._engine, UNetEngine)):
        raise _lib.RgfmError("gradient log-ratio guidance needs two U-Net velocity nets")
    _require_hip(x, y)
    if not (x.is_contiguous() and y.is_contiguous()):
        raise _lib.RgfmError("x and y must be contiguous (they are updated in place)")
    if step_end is None:
        step_end = num_steps
    B, dev = x.shape[0], x.device
    if B == 0:
        return x, y
    L = _lib.lib()

    def run():
        with torch.cuda.device(dev):
            hx, hy, hr = fm_x._engine.handle(dev), fm_y._engine.handle(dev), ratio_estimator._engine.handle(dev)
            nb = ctypes.c_size_t()
            _lib.check(L.rgfm_sample_pair_grad_workspace_bytes(hx, hy, hr, B, ctypes.byref(nb)))
            ws = _sampler_ws.get(nb.value, dev)
            _lib.check(L.rgfm_sample_pair_grad(hx, hy, hr, _ptr(x), _ptr(y), B, int(num_steps), float(gamma),
                                               int(step_begin), int(step_end), _ptr(ws), nb.value, _stream(dev)))
        return x, y
    return _range_guarded(dev, [x, y], run, [fm_x._engine, fm_y._engine])


def guidance_apply(x, y, vx, vy, mc_x1, mc_y1, mc_ratios, t, gamma, want_weights=False):
    """One guidance evaluation (parity hook): overwrites vx, vy; returns weights or None."""
    _require_hip(x, y, vx, vy, mc_x1, mc_y1, mc_ratios)
    B, N, dev = x.shape[0], mc_x1.shape[0], x.device
    dx, dy = x[0].numel(), y[0].numel()
    w = torch.empty(B, N, device=dev) if want_weights else None
    L = _lib.lib()
    with torch.cuda.device(dev):
        nb = ctypes.c_size_t()
        _lib.check(L.rgfm_guidance_workspace_bytes(B, N, ctypes.byref(nb)))
        ws = _sampler_ws.get(nb.value, dev)
        _lib.check(L.rgfm_guidance_apply(_ptr(x.contiguous()), _ptr(y.contiguous()), _ptr(vx), _ptr(vy),
                                         _ptr(mc_x1.contiguous()), _ptr(mc_y1.contiguous()),
                                         _ptr(mc_ratios.contiguous()), B, N, dx, dy, float(t),
                                         float(gamma), _ptr(w), _ptr(ws), nb.value, _stream(dev)))
    return w


def profile(enable=None, reset=False, reserve=None):
    L = _lib.lib()
    if reserve is not None:
        _lib.check(L.rgfm_profile_reserve(int(reserve)))
    if enable is not None:
        _lib.check(L.rgfm_profile_enable(1 if enable else 0))
    if reset:
        _lib.check(L.rgfm_profile_reset())


def profile_read(kclass):
    """(busy_ms [union of launch intervals], sum_ms, launches, flops) of a kernel class."""
    busy, tot, n, fl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    _lib.check(_lib.lib().rgfm_profile_read(kclass, ctypes.byref(busy), ctypes.byref(tot), ctypes.byref(n),
                                            ctypes.byref(fl)))
    return busy.value, tot.value, n.value, fl.value
