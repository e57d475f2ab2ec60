// api_ratio.cpp -- ratio estimators: evaluation, gradient of log r, and the gradient-guided paired sampler (C ABI: include/rgfm.h).
#include "rgfm_host.h"

// ================================================================== ratio estimators
struct rgfm_ratio {
  rgfm_ratio_desc d;
  float* params = nullptr;
  float* packed = nullptr;
  float* bn = nullptr;  // folded BatchNorm scale/shift pairs
  // gradient path (kind RGFM_RATIO_MNIST_SVHN): transposed weights, built at create time
  float* gradw = nullptr;  // [packed W^T of every conv after the first | fc W^T | dense W^T | zeros]
  size_t n_gradw = 0, g_zeros = 0;
  // the encoders' 3x3 convs once more as two scaled fp16 planes (conv_mfma_hx2*.hip): the forward half of the
  // gradient-guided sampler's per-step ratio pass runs on the default arithmetic of the U-Nets it guides
  unsigned short* packedh = nullptr;
  float* hq = nullptr;
  size_t n_packedh = 0;
  int n_hq = 0;
  size_t n_params = 0, n_packed = 0, n_bn = 0;
  struct Conv {
    size_t wt_pk = 0;  // packed transposed weights (offset into gradw), convs after the first
    ConvW wt_h;        // the same transposed weights as two scaled fp16 planes (offsets into packedh / hq): the reverse pass
                       // inside the gradient-guided sampler runs on the U-Nets' default arithmetic (round 4)
    ConvW w;
    size_t nw = 0, nb = 0;                 // GroupNorm weight/bias (mnist28) or BatchNorm w/b
    size_t rm = 0, rv = 0;                 // BatchNorm running stats
    size_t bn_scale = 0, bn_shift = 0;     // offsets into `bn`
    bool pool_after = false;
  };
  struct Encoder {
    int in_ch = 1, size = 32;
    std::vector<Conv> convs;
    size_t fcw = 0, fcb = 0;
    size_t fcw_t = 0;  // fc weight transposed [fc_in][F] (offset into gradw)
    int fc_in = 0;
  };
  Encoder ex, ey;
  struct Dense {
    size_t w, b, lw, lb;
    size_t w_t = 0;  // weight transposed [in][out] (offset into gradw)
    int in, out;
  };
  std::vector<Dense> hidden;
  size_t headw = 0, headb = 0;
  int head_in = 0;
};

namespace {

// Parameter order of RatioEstimatorMNISTSVHN (src/models/ratio_flexible.py:191-208,
// :241-269, :327-345) and RatioEstimator (src/models/ratio_estimator.py:43-65, :121-135).
size_t plan_ratio(const rgfm_ratio_desc& d, rgfm_ratio* h) {
  Cursor c, pk, bn, gw;
  const int F = d.feature_dim, Hd = d.hidden_dim;
  auto encoder = [&](int in_ch, int size, const std::vector<int>& chans, const std::vector<int>& pools, bool batchnorm) {
    rgfm_ratio::Encoder e;
    e.in_ch = in_ch, e.size = size;
    int ci = in_ch;
    for (size_t i = 0; i < chans.size(); ++i) {
      rgfm_ratio::Conv cv;
      cv.w.cin = ci, cv.w.cout = chans[i], cv.w.taps = 9;
      cv.w.w_raw = c.take((size_t)chans[i] * ci * 9);
      cv.w.b = c.take(chans[i]);
      if (i > 0) cv.w.w_pk = pk.take((size_t)chans[i] * ci * 9);
      if (i > 0) cv.wt_pk = gw.take((size_t)chans[i] * ci * 9);
      if (i > 0 && h) cv.w.w_hx2 = h->n_packedh, h->n_packedh += (size_t)chans[i] * ci * 9 * 2, cv.w.hq = h->n_hq++;
      if (i > 0 && h) {
        cv.wt_h.cin = chans[i], cv.wt_h.cout = ci, cv.wt_h.taps = 9;
        cv.wt_h.w_hx2 = h->n_packedh, h->n_packedh += (size_t)chans[i] * ci * 9 * 2, cv.wt_h.hq = h->n_hq++;
      }
      cv.nw = c.take(chans[i]);
      cv.nb = c.take(chans[i]);
      if (batchnorm) {
        cv.rm = c.take(chans[i]);
        cv.rv = c.take(chans[i]);
        c.take(1);  // num_batches_tracked
        cv.bn_scale = bn.take(chans[i]);
        cv.bn_shift = bn.take(chans[i]);
      }
      cv.pool_after = pools[i] != 0;
      e.convs.push_back(cv);
      ci = chans[i];
    }
    e.fc_in = ci;
    e.fcw = c.take((size_t)F * ci);
    e.fcb = c.take(F);
    e.fcw_t = gw.take((size_t)F * ci);
    return e;
  };
  rgfm_ratio::Encoder ex, ey;
  std::vector<int> dims;
  if (d.kind == RGFM_RATIO_MNIST_SVHN) {
    ex = encoder(1, 32, {32, 64, 128, 128}, {1, 1, 1, 0}, true);
    ey = encoder(3, 32, {64, 64, 128, 128, 256, 256, 256, 256}, {0, 1, 0, 1, 0, 1, 0, 1}, true);
    dims = {2 * F, Hd, Hd, Hd / 2};
  } else {
    ex = encoder(1, 28, {32, 64, 128, 128}, {1, 1, 1, 0}, false);
    ey = encoder(1, 28, {32, 64, 128, 128}, {1, 1, 1, 0}, false);
    dims = {2 * F, Hd, Hd / 2};
  }
  std::vector<rgfm_ratio::Dense> hidden;
  for (size_t l = 0; l + 1 < dims.size(); ++l) {
    rgfm_ratio::Dense dn;
    dn.in = dims[l], dn.out = dims[l + 1];
    dn.w = c.take((size_t)dn.in * dn.out), dn.b = c.take(dn.out);
    dn.w_t = gw.take((size_t)dn.in * dn.out);
    dn.lw = c.take(dn.out), dn.lb = c.take(dn.out);
    hidden.push_back(dn);
  }
  const size_t headw = c.take(dims.back()), headb = c.take(1);
  if (h) {
    h->ex = ex, h->ey = ey, h->hidden = hidden, h->headw = headw, h->headb = headb, h->head_in = dims.back();
    h->n_packed = pk.off, h->n_bn = bn.off;
    h->g_zeros = gw.take(1024);
    h->n_gradw = gw.off;
  }
  return c.off;
}

int check_ratio_desc(const rgfm_ratio_desc* d) {
  if (!d) return fail(RGFM_EINVAL, "null descriptor");
  if (d->kind != RGFM_RATIO_MNIST_SVHN && d->kind != RGFM_RATIO_MNIST28) return fail(RGFM_EINVAL, "unknown ratio kind");
  if (d->feature_dim % 64 || d->hidden_dim % 128 || d->feature_dim > 512 || d->hidden_dim > 1024)
    return fail(RGFM_EINVAL, "feature_dim must be a multiple of 64 (<=512), hidden_dim of 128 (<=1024)");
  if (d->loss_type != RGFM_LOSS_DISC && d->loss_type != RGFM_LOSS_RULSIF) return fail(RGFM_EINVAL, "unknown loss_type");
  return RGFM_OK;
}

struct RatioRun {
  rgfm_ratio* h;
  int n;
  Bump* ws;
  hipStream_t s;
  bool dry;

  // one encoder: image NCHW -> features written at feat[:, col0 : col0+F] (row stride 2F)
  void encode(const rgfm_ratio::Encoder& e, const float* img, float* feat, int col0) {
    const bool gn = h->d.kind == RGFM_RATIO_MNIST28;
    const int F = h->d.feature_dim;
    int S = e.size;
    Tensor cur;
    float* ab = nullptr;  // pending GroupNorm scale/shift of `cur` (mnist28)
    for (size_t i = 0; i < e.convs.size(); ++i) {
      const rgfm_ratio::Conv& cv = e.convs[i];
      const TileGeom g = make_geom(S, S);
      Tensor o;
      o.C = cv.w.cout, o.S = S;
      o.data = ws->f((size_t)n * S * S * o.C);
      o.stats = gn ? ws->f((size_t)n * g.nparts * o.C * 2) : nullptr;
      if (!dry) {
        const float* es = gn ? nullptr : h->bn + cv.bn_scale;
        const float* eh = gn ? nullptr : h->bn + cv.bn_shift;
        if (i == 0) {
          ConvInArgs ci{};
          ci.x = img, ci.w = h->params + cv.w.w_raw, ci.bias = h->params + cv.w.b;
          ci.ep_scale = es, ci.ep_shift = eh;
          ci.out = o.data, ci.stats_out = o.stats, ci.B = n, ci.C0 = o.C, ci.g = g;
          ProfScope p(RGFM_KCLASS_OTHER, 0, s);
          launch_conv_in(ci, e.in_ch, s);
        } else {
          ConvArgs c{};
          c.in0 = cur.data, c.C0 = cur.C, c.Hin = c.Win = S;
          c.wpk = h->packed + cv.w.w_pk, c.bias = h->params + cv.w.b;
          c.ep_scale = es, c.ep_shift = eh;
          c.out = o.data, c.stats_out = o.stats, c.B = n, c.Cout = o.C, c.g = g;
          c.halo_px = g.spt * (g.th + 2) * (g.W + 2);
          ProfScope p(RGFM_KCLASS_CONV_MFMA, conv_flops(n, S * S, o.C, 9 * cur.C), s);
          launch_conv(c, CONV_S1, s);
        }
      }
      cur = o;
      if (gn) {
        ab = ws->f((size_t)n * o.C * 2);
        if (!dry) {
          GnFinalizeArgs f{};
          f.stats0 = o.stats, f.C0 = o.C, f.groups = 8;
          f.gamma = h->params + cv.nw, f.beta = h->params + cv.nb, f.ab = ab, f.B = n, f.g = g;
          ProfScope p(RGFM_KCLASS_OTHER, 0, s);
          launch_gn_finalize(f, s);
        }
      }
      if (cv.pool_after) {
        Tensor pl;
        pl.C = cur.C, pl.S = S / 2;
        pl.data = ws->f((size_t)n * pl.S * pl.S * pl.C);
        if (!dry) {
          ProfScope p(RGFM_KCLASS_OTHER, 0, s);
          launch_pool2(cur.data, gn ? ab : nullptr, pl.data, n, S, S, cur.C, s);
        }
        cur = pl;
        S /= 2;
        ab = nullptr;
      }
    }
    float* pooled = ws->f((size_t)n * cur.C);
    if (!dry) {
      ProfScope p(RGFM_KCLASS_OTHER, 0, s);
      launch_avgpool(cur.data, ab, pooled, n, S * S, cur.C, s);
      launch_linear_mfma(pooled, h->params + e.fcw, h->params + e.fcb, feat + col0, n, cur.C, F, cur.C, 2 * F, s);
    }
  }

  void run(const float* x, const float* y, float* out, int what) {
    const int F = h->d.feature_dim;
    float* feat = ws->f((size_t)n * 2 * F);
    encode(h->ex, x, feat, 0);
    encode(h->ey, y, feat, F);
    float* cur = feat;
    for (const auto& dn : h->hidden) {
      float* nxt = ws->f((size_t)n * dn.out);
      if (!dry) {
        ProfScope p(RGFM_KCLASS_OTHER, 0, s);
        launch_linear_mfma(cur, h->params + dn.w, h->params + dn.b, nxt, n, dn.in, dn.out, dn.in, dn.out, s);
        launch_layernorm_silu(nxt, h->params + dn.lw, h->params + dn.lb, n, dn.out, s);
      }
      cur = nxt;
    }
    if (!dry) {
      ProfScope p(RGFM_KCLASS_OTHER, 0, s);
      launch_ratio_head(cur, h->params + h->headw, h->params + h->headb, out, n, h->head_in, h->d.loss_type, what, s);
    }
  }
};

}  // namespace

extern "C" int rgfm_ratio_param_floats(const rgfm_ratio_desc* desc, size_t* n_floats) {
  int rc = check_ratio_desc(desc);
  if (rc) return rc;
  if (!n_floats) return fail(RGFM_EINVAL, "null output");
  *n_floats = plan_ratio(*desc, nullptr);
  return RGFM_OK;
}

extern "C" int rgfm_ratio_create(const rgfm_ratio_desc* desc, const float* params_dev, size_t n_floats,
                                 rgfm_stream_t stream, rgfm_ratio** out) {
  int rc = check_ratio_desc(desc);
  if (rc) return rc;
  if (!params_dev || !out) return fail(RGFM_EINVAL, "null argument");
  if ((rc = ensure_init())) return rc;
  hipStream_t s = (hipStream_t)stream;
  rgfm_ratio* h = new rgfm_ratio();
  h->d = *desc;
  h->n_params = plan_ratio(*desc, h);
  if (h->n_params != n_floats) {
    const size_t want = h->n_params;
    delete h;
    return fail(RGFM_EINVAL, "parameter blob has %zu floats, architecture needs %zu", n_floats, want);
  }
  auto bail = [&](int code, const char* what) {
    rgfm_ratio_destroy(h);
    return fail(code, "%s", what);
  };
  if (hipMalloc(&h->params, n_floats * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(params)");
  if (hipMalloc(&h->packed, (h->n_packed + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packed)");
  if (hipMalloc(&h->bn, (h->n_bn + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(bn)");
  if (hipMemcpyAsync(h->params, params_dev, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return bail(RGFM_EHIP, "hipMemcpyAsync(params)");
  for (const auto* e : {&h->ex, &h->ey})
    for (size_t i = 0; i < e->convs.size(); ++i) {
      const auto& cv = e->convs[i];
      if (i > 0) launch_pack_conv(h->params + cv.w.w_raw, h->packed + cv.w.w_pk, cv.w.cout, cv.w.cin, 9, nt32_of(cv.w.cout), s);
      if (desc->kind == RGFM_RATIO_MNIST_SVHN)
        launch_bn_fold(h->params + cv.nw, h->params + cv.nb, h->params + cv.rm, h->params + cv.rv,
                       h->bn + cv.bn_scale, h->bn + cv.bn_shift, cv.w.cout, s);
    }
  {
    // gradient path (rgfm_ratio_grad_log_ratio): dL/d(in) of a 3x3 conv is the conv of dL/d(out) with the weights
    // transposed and the taps flipped; of a Linear, the Linear with W^T
    if (hipMalloc(&h->gradw, (h->n_gradw + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(gradw)");
    if (hipMalloc(&h->packedh, (h->n_packedh + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packedh)");
    if (hipMalloc(&h->hq, ((size_t)h->n_hq * 4 + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(hq)");
    float* tmp = nullptr;
    if (hipMalloc(&tmp, (size_t)256 * 256 * 9 * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(tmp)");
    for (const auto* e : {&h->ex, &h->ey}) {
      for (size_t i = 1; i < e->convs.size(); ++i) {
        const auto& cv = e->convs[i];
        launch_conv_weight_transpose(h->params + cv.w.w_raw, tmp, cv.w.cout, cv.w.cin, s);
        launch_pack_conv(tmp, h->gradw + cv.wt_pk, cv.w.cin, cv.w.cout, 9, nt32_of(cv.w.cin), s);
        launch_pack_conv_hx2(tmp, h->packedh + cv.wt_h.w_hx2, h->hq + 4 * cv.wt_h.hq, cv.w.cin, cv.w.cout, 9, CONV_S1, s);
      }
      launch_transpose2d(h->params + e->fcw, h->gradw + e->fcw_t, desc->feature_dim, e->fc_in, s);
    }
    for (const auto& dn : h->hidden) launch_transpose2d(h->params + dn.w, h->gradw + dn.w_t, dn.out, dn.in, s);
    launch_fill(h->gradw + h->g_zeros, 0.f, 1024, s);
    if (hipStreamSynchronize(s) != hipSuccess) {
      (void)hipFree(tmp);
      return bail(RGFM_EHIP, "building the transposed weights failed");
    }
    (void)hipFree(tmp);
    std::vector<ConvW*> all;
    for (auto* e : {&h->ex, &h->ey})
      for (size_t i = 1; i < e->convs.size(); ++i) {
        ConvW& w = e->convs[i].w;
        launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + w.w_hx2, h->hq + 4 * w.hq, w.cout, w.cin, 9, CONV_S1, s);
        all.push_back(&w);
        all.push_back(&e->convs[i].wt_h);
      }
    if (read_hx_flags(h->hq, h->n_hq, all, s) != RGFM_OK) return bail(RGFM_EHIP, "reading the fp16 scale records failed");
  }
  *out = h;
  return RGFM_OK;
}

extern "C" void rgfm_ratio_destroy(rgfm_ratio* h) {
  if (!h) return;
  if (h->params) (void)hipFree(h->params);
  if (h->packed) (void)hipFree(h->packed);
  if (h->bn) (void)hipFree(h->bn);
  if (h->packedh) (void)hipFree(h->packedh);
  if (h->hq) (void)hipFree(h->hq);
  if (h->gradw) (void)hipFree(h->gradw);
  delete h;
}

extern "C" int rgfm_ratio_workspace_bytes(const rgfm_ratio* h, int n, size_t* bytes) {
  if (!h || !bytes || n < 1) return fail(RGFM_EINVAL, "bad argument");
  Bump b;
  RatioRun r{const_cast<rgfm_ratio*>(h), n, &b, nullptr, true};
  r.run(nullptr, nullptr, nullptr, 0);
  *bytes = b.off;
  return RGFM_OK;
}

extern "C" int rgfm_ratio_eval(rgfm_ratio* h, const float* x, const float* y, float* out, int n, int what,
                               void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!h || !x || !y || !out || !ws) return fail(RGFM_EINVAL, "null argument");
  if (what < 0 || what > 2) return fail(RGFM_EINVAL, "bad output selector");
  size_t need = 0;
  int rc = rgfm_ratio_workspace_bytes(h, n, &need);
  if (rc) return rc;
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  RatioRun r{h, n, &b, (hipStream_t)stream, false};
  r.run(x, y, out, what);
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

// ------------------------------------------------------------------ gradient of log r
namespace {

// Forward of RatioEstimatorMNISTSVHN with every pre-activation kept (BatchNorm output z of each conv, Linear output
// u of each score_net layer), then the reverse pass down to the two images.  Same kernels as RatioRun for the
// forward (the conv epilogue stores z instead of silu(z); SiLU is applied by the consumer through an identity
// scale/shift array), conv_mfma with transposed weights / linear_mfma with W^T / ratio_grad.hip for the reverse.
struct RatioGradRun {
  rgfm_ratio* h;
  int n;
  Bump* ws;
  hipStream_t s;
  bool dry;
  // the range-flag word of the U-Net handle whose sampler loop this pass belongs to, or null (stand-alone gradient:
  // exact fp32 convs).  With it the encoders' forward convs follow that handle's conv arithmetic (g_modes, set by the
  // caller's ModeScope) and raise ITS flag, so that the sampler's range guard and fallback cover them.
  unsigned* flag = nullptr;
  float* ab1 = nullptr;  // [n][256][2] identity scale/shift: "SiLU on load"
  unsigned* amax = nullptr;  // one word per reverse conv: bits of max |gradient| of its input (ConvArgs::in_amax); zeroed per call
  int amax_used = 0;

  struct Kept {
    float* z;
    int C, S;
    bool pooled;
    float* ab = nullptr;  // GroupNorm encoders: the samples' scale/shift pairs [n][C][2] ...
    float* mr = nullptr;  // ... and (mean, rstd) of every group [n][8][2]
  };

  // RatioEstimator's ImageEncoder (ratio_estimator.py:67-93) with the conv outputs and their norms' statistics kept
  void encode_gn(const rgfm_ratio::Encoder& e, const float* img, float* feat, int col0, std::vector<Kept>& kept) {
    const int F = h->d.feature_dim;
    int S = e.size;
    const float* cur = nullptr;
    int curC = e.in_ch;
    for (size_t i = 0; i < e.convs.size(); ++i) {
      const rgfm_ratio::Conv& cv = e.convs[i];
      const TileGeom g = make_geom(S, S);
      const int C = cv.w.cout;
      float* z = ws->f((size_t)n * S * S * C);
      float* stats = ws->f((size_t)n * g.nparts * C * 2);
      float* ab = ws->f((size_t)n * C * 2);
      float* mr = ws->f((size_t)n * 8 * 2);
      if (!dry) {
        if (i == 0) {
          ConvInArgs ci{};
          ci.x = img, ci.w = h->params + cv.w.w_raw, ci.bias = h->params + cv.w.b;
          ci.out = z, ci.stats_out = stats, ci.B = n, ci.C0 = C, ci.g = g;
          launch_conv_in(ci, e.in_ch, s);
        } else {
          ConvArgs c{};
          c.in0 = cur, c.C0 = curC, c.Hin = c.Win = S;
          c.wpk = h->packed + cv.w.w_pk, c.bias = h->params + cv.w.b;
          c.out = z, c.stats_out = stats, c.B = n, c.Cout = C, c.g = g;
          c.halo_px = g.spt * (g.th + 2) * (g.W + 2);
          launch_conv_mfma(c, CONV_S1, s);
        }
        GnFinalizeArgs f{};
        f.stats0 = stats, f.C0 = C, f.groups = 8;
        f.gamma = h->params + cv.nw, f.beta = h->params + cv.nb, f.ab = ab, f.mr = mr, f.B = n, f.g = g;
        launch_gn_finalize(f, s);
      }
      Kept k{z, C, S, cv.pool_after};
      k.ab = ab, k.mr = mr;
      kept.push_back(k);
      curC = C;
      if (cv.pool_after) {
        float* pl = ws->f((size_t)n * (S / 2) * (S / 2) * C);
        if (!dry) launch_pool2(z, ab, pl, n, S, S, C, s);
        cur = pl;
        S /= 2;
      } else {
        cur = z;  // (only the last conv: the average pool applies its norm and SiLU)
      }
    }
    const Kept& last = kept.back();
    float* pooled = ws->f((size_t)n * curC);
    if (!dry) {
      launch_avgpool(cur, last.pooled ? nullptr : last.ab, pooled, n, S * S, curC, s);
      launch_linear_mfma(pooled, h->params + e.fcw, h->params + e.fcb, feat + col0, n, curC, F, curC, 2 * F, s);
    }
  }

  float* encode(const rgfm_ratio::Encoder& e, const float* img, float* feat, int col0, std::vector<Kept>& kept) {
    const int F = h->d.feature_dim;
    int S = e.size;
    const float* cur = nullptr;  // input of the next conv
    bool cur_is_z = false;       // ... is a kept pre-activation (SiLU on load) rather than a pooled map
    int curC = e.in_ch;
    for (size_t i = 0; i < e.convs.size(); ++i) {
      const rgfm_ratio::Conv& cv = e.convs[i];
      const TileGeom g = make_geom(S, S);
      float* z = ws->f((size_t)n * S * S * cv.w.cout);
      if (!dry) {
        if (i == 0) {
          ConvInArgs ci{};
          ci.x = img, ci.w = h->params + cv.w.w_raw, ci.bias = h->params + cv.w.b;
          ci.ep_scale = h->bn + cv.bn_scale, ci.ep_shift = h->bn + cv.bn_shift, ci.ep_nosilu = 1;
          ci.out = z, ci.stats_out = nullptr, ci.B = n, ci.C0 = cv.w.cout, ci.g = g;
          launch_conv_in(ci, e.in_ch, s);
        } else {
          ConvArgs c{};
          c.in0 = cur, c.C0 = curC, c.Hin = c.Win = S;
          c.ab = cur_is_z ? ab1 : nullptr;
          c.wpk = h->packed + cv.w.w_pk, c.bias = h->params + cv.w.b;
          c.ep_scale = h->bn + cv.bn_scale, c.ep_shift = h->bn + cv.bn_shift, c.ep_nosilu = 1;
          c.out = z, c.stats_out = nullptr, c.B = n, c.Cout = cv.w.cout, c.g = g;
          c.halo_px = g.spt * (g.th + 2) * (g.W + 2);
          if (flag && g_modes.conv == CONV_ARITH_HX2 && cv.w.hx_ok) {
            c.wpkh = h->packedh + cv.w.w_hx2, c.hq = h->hq + 4 * cv.w.hq, c.range_flag = flag;
            launch_conv(c, CONV_S1, s);  // (fp16 two-plane conv with the BatchNorm epilogue; fp32 MFMA when unsupported)
          } else {
            launch_conv_mfma(c, CONV_S1, s);
          }
        }
      }
      kept.push_back({z, cv.w.cout, S, cv.pool_after});
      curC = cv.w.cout;
      if (cv.pool_after) {
        float* pl = ws->f((size_t)n * (S / 2) * (S / 2) * curC);
        if (!dry) launch_pool2(z, ab1, pl, n, S, S, curC, s);
        cur = pl, cur_is_z = false;
        S /= 2;
      } else {
        cur = z, cur_is_z = true;
      }
    }
    float* pooled = ws->f((size_t)n * curC);
    if (!dry) {
      launch_avgpool(cur, cur_is_z ? ab1 : nullptr, pooled, n, S * S, curC, s);
      launch_linear_mfma(pooled, h->params + e.fcw, h->params + e.fcb, feat + col0, n, curC, F, curC, 2 * F, s);
    }
    return pooled;
  }

  // reverse pass of one encoder: gfeat [n][2F] (columns col0 .. col0+F) -> gimg NCHW
  void encode_bwd(const rgfm_ratio::Encoder& e, const std::vector<Kept>& kept, const float* gfeat, int col0, float* gimg) {
    const int F = h->d.feature_dim;
    const float* zeros = h->gradw + h->g_zeros;
    const Kept& last = kept.back();
    float* g = ws->f((size_t)n * last.C);  // gradient of the average-pooled vector
    if (!dry) launch_linear_mfma(gfeat + col0, h->gradw + e.fcw_t, zeros, g, n, F, last.C, 2 * F, last.C, s);
    int mode = 2;  // first step: g is [n][C] behind the global average pool
    for (int i = (int)kept.size() - 1; i >= 0; --i) {
      const Kept& k = kept[i];
      const rgfm_ratio::Conv& cv = e.convs[i];
      if (i != (int)kept.size() - 1) mode = k.pooled ? 1 : 0;
      else mode = k.pooled ? 3 : 2;  // (SVHN encoder: a max-pool sits between the last conv and the average pool)
      float* gz = ws->f((size_t)n * k.S * k.S * k.C);
      unsigned* am = nullptr;
      if (!dry) {
        if (k.ab) {  // GroupNorm encoder: SiLU' (and the max-pool routing) at u = a z + b, then the norm's backward in place
          launch_grad_act_gn(g, k.z, k.ab, gz, n, k.S, k.C, mode, s);
          launch_gn_bwd(gz, k.z, h->params + cv.nw, k.mr, n, k.S * k.S, k.C, 8, s);
        } else {
          // (inside the sampler the next conv runs on the two-plane arithmetic: it needs the tensor's magnitude)
          am = (flag && i > 0 && amax && g_modes.conv == CONV_ARITH_HX2 && g_modes.rev_hx2 && cv.wt_h.hx_ok) ? amax + amax_used++ : nullptr;
          launch_grad_act(g, k.z, h->bn + cv.bn_scale, gz, n, k.S, k.C, mode, s, am);
        }
      }
      if (i == 0) {
        if (!dry) launch_conv_bwd_img(gz, h->params + cv.w.w_raw, gimg, n, k.S, k.C, e.in_ch, s);
      } else {
        float* gin = ws->f((size_t)n * k.S * k.S * cv.w.cin);
        if (!dry) {
          ConvArgs c{};
          c.in0 = gz, c.C0 = k.C, c.Hin = c.Win = k.S;
          c.wpk = h->gradw + cv.wt_pk, c.bias = zeros;
          c.out = gin, c.stats_out = nullptr, c.B = n, c.Cout = cv.w.cin;
          c.g = make_geom(k.S, k.S);
          c.halo_px = c.g.spt * (c.g.th + 2) * (c.g.W + 2);
          // Inside the gradient-guided sampler (flag set): the data gradient on the two-plane fp16 arithmetic.  Gradients
          // are 1e-3 ... 1e-6 in magnitude -- below the raw staging window -- so the conv stages them x 2^-e for the
          // tensor's measured maximum f 2^e (grad_act wrote its bits) and scales its outputs back; the staged values are
          // then <= 16 and the x net's range flag covers the rest.  Otherwise (stand-alone gradient, fallback modes,
          // weights outside the fp16 window): the exact fp32 matrix-core conv.
          bool hx = false;
          if (am) {
            c.wpkh = h->packedh + cv.wt_h.w_hx2, c.hq = h->hq + 4 * cv.wt_h.hq, c.range_flag = flag, c.in_amax = am;
            hx = conv_hx2_supported(c, CONV_S1);
          }
          if (hx) launch_conv_hx2(c, CONV_S1, s);
          else launch_conv_mfma(c, CONV_S1, s);
        }
        g = gin;
      }
    }
  }

  void run(const float* x, const float* y, float* gx, float* gy, float* log_ratio) {
    const int F = h->d.feature_dim;
    ab1 = ws->f((size_t)n * 256 * 2);
    amax = reinterpret_cast<unsigned*>(ws->f(64));
    amax_used = 0;
    if (!dry) {
      launch_fill_ab_identity(ab1, (size_t)n * 256, s);
      (void)hipMemsetAsync(amax, 0, 64 * sizeof(unsigned), s);
    }
    float* feat = ws->f((size_t)n * 2 * F);
    std::vector<Kept> kx, ky;
    if (h->d.kind == RGFM_RATIO_MNIST28) {
      encode_gn(h->ex, x, feat, 0, kx);
      encode_gn(h->ey, y, feat, F, ky);
    } else {
      encode(h->ex, x, feat, 0, kx);
      encode(h->ey, y, feat, F, ky);
    }
    std::vector<float*> us, ins{feat};
    float* cur = feat;
    for (const auto& dn : h->hidden) {
      float* u = ws->f((size_t)n * dn.out);
      float* a = ws->f((size_t)n * dn.out);
      if (!dry) {
        launch_linear_mfma(cur, h->params + dn.w, h->params + dn.b, u, n, dn.in, dn.out, dn.in, dn.out, s);
        (void)hipMemcpyAsync(a, u, (size_t)n * dn.out * sizeof(float), hipMemcpyDeviceToDevice, s);
        launch_layernorm_silu(a, h->params + dn.lw, h->params + dn.lb, n, dn.out, s);
      }
      us.push_back(u);
      cur = a;
    }
    float* score = ws->f(n);
    float* g = ws->f((size_t)n * h->head_in);
    if (!dry) {
      launch_ratio_head(cur, h->params + h->headw, h->params + h->headb, score, n, h->head_in, h->d.loss_type, 0, s);
      launch_ratio_head_bwd(score, h->params + h->headw, g, log_ratio, n, h->head_in, h->d.loss_type, s);
    }
    const float* zeros = h->gradw + h->g_zeros;
    for (int l = (int)h->hidden.size() - 1; l >= 0; --l) {
      const auto& dn = h->hidden[l];
      float* gu = ws->f((size_t)n * dn.out);
      float* gi = ws->f((size_t)n * dn.in);
      if (!dry) {
        launch_layernorm_silu_bwd(us[l], g, h->params + dn.lw, h->params + dn.lb, gu, n, dn.out, s);
        launch_linear_mfma(gu, h->gradw + dn.w_t, zeros, gi, n, dn.out, dn.in, dn.out, dn.in, s);
      }
      g = gi;
    }
    encode_bwd(h->ex, kx, g, 0, gx);
    encode_bwd(h->ey, ky, g, F, gy);
  }
};

size_t ratio_grad_bytes(rgfm_ratio* h, int n) {
  Bump b;
  RatioGradRun r{h, n, &b, nullptr, true};
  r.run(nullptr, nullptr, nullptr, nullptr, nullptr);
  return b.off;
}

}  // namespace

extern "C" int rgfm_ratio_grad_workspace_bytes(const rgfm_ratio* h, int n, size_t* bytes) {
  if (!h || !bytes || n < 1) return fail(RGFM_EINVAL, "bad argument");
  *bytes = ratio_grad_bytes(const_cast<rgfm_ratio*>(h), n);
  return RGFM_OK;
}

extern "C" int rgfm_ratio_grad_log_ratio(rgfm_ratio* h, const float* x, const float* y, float* gx, float* gy,
                                         float* log_ratio_out, int n, void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  if (!h || !x || !y || !gx || !gy || !ws) return fail(RGFM_EINVAL, "null argument");
  size_t need = 0;
  int rc = rgfm_ratio_grad_workspace_bytes(h, n, &need);
  if (rc) return rc;
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  refresh_modes();
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  RatioGradRun r{h, n, &b, (hipStream_t)stream, false};
  r.run(x, y, gx, gy, log_ratio_out);
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

// Paired Euler loop with gradient log-ratio guidance (reference README.md:159-164: v_guided = v_ind + gamma *
// grad log r(x_t, y_t); the reference ships no code for it): x <- x + (v_x + gamma dlogr/dx) dt, every step.
extern "C" int rgfm_sample_pair_grad_workspace_bytes(const rgfm_unet* hx, const rgfm_unet* hy, const rgfm_ratio* hr, int batch,
                                                     size_t* bytes) {
  if (!hx || !hy || !hr || !bytes || batch < 1) return fail(RGFM_EINVAL, "bad argument");
  size_t base = 0, rg = 0;
  int rc = rgfm_sample_pair_workspace_bytes(hx, hy, batch, 0, &base);
  if (rc) return rc;
  if ((rc = rgfm_ratio_grad_workspace_bytes(hr, batch, &rg))) return rc;
  const size_t dx = (size_t)hx->d.in_channels * hx->d.img_size * hx->d.img_size;
  const size_t dy = (size_t)hy->d.in_channels * hy->d.img_size * hy->d.img_size;
  *bytes = base + rg + ((batch * dx * 4 + 255) & ~(size_t)255) + ((batch * dy * 4 + 255) & ~(size_t)255);
  return RGFM_OK;
}

extern "C" int rgfm_sample_pair_grad(rgfm_unet* hx, rgfm_unet* hy, rgfm_ratio* hr, float* x_inout, float* y_inout, int batch,
                                     int num_steps, double gamma, int step_begin, int step_end, void* ws, size_t ws_bytes,
                                     rgfm_stream_t stream) {
  refresh_modes();
  if (!hx || !hy || !hr || !x_inout || !y_inout || !ws) return fail(RGFM_EINVAL, "null argument");
  if (hr->d.kind == RGFM_RATIO_MNIST_SVHN) {
    if (hx->d.in_channels != 1 || hx->d.img_size != 32 || hy->d.in_channels != 3 || hy->d.img_size != 32)
      return fail(RGFM_EINVAL, "gradient guidance with RatioEstimatorMNISTSVHN needs the 1x32x32 + 3x32x32 pair");
  } else if (hx->d.in_channels != 1 || hx->d.img_size != 28 || hy->d.in_channels != 1 || hy->d.img_size != 28) {
    return fail(RGFM_EINVAL, "gradient guidance with RatioEstimator needs the 1x28x28 + 1x28x28 pair");
  }
  if (batch < 1 || num_steps < 1 || step_begin < 0 || step_end > num_steps || step_begin > step_end)
    return fail(RGFM_EINVAL, "bad step range [%d,%d) of %d", step_begin, step_end, num_steps);
  const int ns = step_end - step_begin;
  if (ns > 4096) return fail(RGFM_EINVAL, "at most 4096 steps per call");
  size_t need = 0;
  int rc = rgfm_sample_pair_grad_workspace_bytes(hx, hy, hr, batch, &need);
  if (rc) return rc;
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  if (ns == 0) return RGFM_OK;
  DevState* ds = cur_dev();
  if (!ds) return fail(RGFM_EINVAL, "no handle has been created on the current device");
  hipStream_t s = (hipStream_t)stream;
  const int dx = hx->d.in_channels * hx->d.img_size * hx->d.img_size, dy = hy->d.in_channels * hy->d.img_size * hy->d.img_size;
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  float* tx = b.f((size_t)4096 * hx->temb_total);
  float* ty = b.f((size_t)4096 * hy->temb_total);
  float* vx = b.f((size_t)batch * dx);
  float* vy = b.f((size_t)batch * dy);
  float* gx = b.f((size_t)batch * dx);
  float* gy = b.f((size_t)batch * dy);
  unsigned* cnt_x = reinterpret_cast<unsigned*>(b.f(batch));
  unsigned* cnt_y = reinterpret_cast<unsigned*>(b.f(batch));
  HIP_TRY(hipMemsetAsync(cnt_x, 0, (size_t)batch * sizeof(unsigned), s));
  HIP_TRY(hipMemsetAsync(cnt_y, 0, (size_t)batch * sizeof(unsigned), s));
  launch_time_table(hx, nullptr, num_steps, step_begin, ns, tx, s);
  launch_time_table(hy, nullptr, num_steps, step_begin, ns, ty, s);
  const size_t mark_x = b.off;
  const size_t mark_y = mark_x + unet_eval_bytes(hx, batch);
  const size_t mark_r = mark_y + unet_eval_bytes(hy, batch);
  const float dt = (float)(1.0 / (double)num_steps), gf = (float)gamma;
  const bool overlap = g_modes.overlap;
  for (int i = 0; i < ns; ++i) {
    hipStream_t sy = overlap ? ds->side : s;
    if (overlap) {
      HIP_TRY(hipEventRecord(ds->fork, s));
      HIP_TRY(hipStreamWaitEvent(ds->side, ds->fork, 0));
    }
    {
      b.off = mark_y;
      UNetRun r{hy, batch, &b, sy, ty + (size_t)i * hy->temb_total, 0, false};
      r.fin_counter = cnt_y;
      if ((rc = r.run(y_inout, vy, nullptr, 0.f))) return rc;
    }
    if (overlap) HIP_TRY(hipEventRecord(ds->join, ds->side));
    {
      b.off = mark_x;
      UNetRun r{hx, batch, &b, s, tx + (size_t)i * hx->temb_total, 0, false};
      r.fin_counter = cnt_x;
      if ((rc = r.run(x_inout, vx, nullptr, 0.f))) return rc;
    }
    {
      b.off = mark_r;
      RatioGradRun r{hr, batch, &b, s, false};
      r.flag = hx->range_flag;
      ModeScope ratio_mode(hx->conv_mode);  // (the estimator's forward convs follow the x net's handle)
      r.run(x_inout, y_inout, gx, gy, nullptr);
    }
    if (overlap) HIP_TRY(hipStreamWaitEvent(s, ds->join, 0));
    launch_euler_grad(x_inout, vx, gx, (size_t)batch * dx, gf, dt, s);
    launch_euler_grad(y_inout, vy, gy, (size_t)batch * dy, gf, dt, s);
  }
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}
