// api_unet.cpp -- the U-Net handle: parameter ingestion / packing, forward, trace hooks (the walk itself: rgfm_host.h, UNetRun) (C ABI: include/rgfm.h).
#include "rgfm_host.h"

extern "C" int rgfm_unet_param_floats(const rgfm_unet_desc* desc, size_t* n_floats) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!n_floats) return fail(RGFM_EINVAL, "null output");
  *n_floats = plan_unet(*desc, nullptr);
  return RGFM_OK;
}

extern "C" int rgfm_unet_create(const rgfm_unet_desc* desc, const float* params_dev, size_t n_floats,
                                rgfm_stream_t stream, rgfm_unet** out) {
  int rc = check_desc(desc);
  if (rc) return rc;
  if (!params_dev || !out) return fail(RGFM_EINVAL, "null argument");
  if ((rc = ensure_init())) return rc;
  hipStream_t s = (hipStream_t)stream;
  rgfm_unet* h = new rgfm_unet();
  h->d = *desc;
  h->n_params = plan_unet(*desc, h);
  if (h->n_params != n_floats) {
    const size_t want = h->n_params;
    delete h;
    return fail(RGFM_EINVAL, "parameter blob has %zu floats, architecture needs %zu", n_floats, want);
  }
  for (const auto* v : {&h->enc, &h->mid, &h->dec})
    for (const ResW& r : *v) {
      if (r.cin % KC || r.cout % 32) {
        delete h;
        return fail(RGFM_EINVAL, "channel counts must be multiples of 16 (in) / 32 (out)");
      }
    }
  auto bail = [&](int code, const char* what) {
    rgfm_unet_destroy(h);
    return fail(code, "%s", what);
  };
  if (hipMalloc(&h->params, n_floats * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(params)");
  if (hipMalloc(&h->packed, (h->n_packed + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packed)");
  if (hipMalloc(&h->packedh, (h->n_packedh + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packedh)");
  if (hipMalloc(&h->hq, ((size_t)h->n_hq * 4 + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(hq)");
  if (alloc_flag_word(&h->range_flag) != RGFM_OK) return bail(RGFM_ENOMEM, "hipMalloc(range flag)");
  if (hipMalloc(&h->packed3, (h->n_packed3 + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packed3)");
  if (hipMemcpyAsync(h->params, params_dev, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return bail(RGFM_EHIP, "hipMemcpyAsync(params)");
  std::vector<ConvW*> all;
  for (auto* v : {&h->enc, &h->mid, &h->dec})
    for (ResW& r : *v) {
      all.push_back(&r.c1), all.push_back(&r.c2);
      if (r.has_skip) all.push_back(&r.sk);
    }
  for (ConvW* w : all) pack_one(h, *w, CONV_S1, s);
  for (ConvW& w : h->down) pack_one(h, w, CONV_S2, s), all.push_back(&w);  // stride-2 convs: phase-ordered weights
  for (ConvW& w : h->down)  // ... and once more in plain tap order (same scale record: same weights)
    launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + (w.w_hx9 - 1), h->hq + 4 * w.hq, w.cout, w.cin, 9, CONV_S1, s);
  for (ConvW& w : h->up) pack_one(h, w, CONV_S1, s), all.push_back(&w);
  // the Upsample convs once more as ConvTranspose2d(4, 2, 1) weights (summed taps), packed in parity-class order; the
  // ResBlock convs once more as Winograd images (one scratch array serves both)
  std::vector<ConvW> t2rec(h->up.size());
  std::vector<ConvW*> wino;
  for (auto* v : {&h->enc, &h->mid, &h->dec})
    for (ResW& r : *v) {
      if (r.c1.w_w) wino.push_back(&r.c1);
      if (r.c2.w_w) wino.push_back(&r.c2);
    }
  std::vector<ConvW> wrec(wino.size());
  float* t2tmp = nullptr;
  {
    size_t mx = 0;
    for (const ConvW& w : h->up) mx = std::max(mx, (size_t)w.cin * w.cout * 16);
    for (const ConvW* w : wino) mx = std::max(mx, (size_t)w->cin * w->cout * 16);
    if (mx && hipMalloc(&t2tmp, mx * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(upsample weights)");
    for (size_t i = 0; i < h->up.size(); ++i) {
      const ConvW& w = h->up[i];
      launch_up2_as_deconv(h->params + w.w_raw, t2tmp, w.cout, w.cin, s);
      launch_pack_conv_hx2(t2tmp, h->packedh + (w.w_t2 - 1), h->hq + 4 * w.hq_t2, w.cout, w.cin, 16, CONV_T2, s);
      t2rec[i].hq = w.hq_t2;
      all.push_back(&t2rec[i]);
    }
    for (size_t i = 0; i < wino.size(); ++i) {
      const ConvW& w = *wino[i];
      launch_pack_conv_hx2w(h->params + w.w_raw, h->packedh + (w.w_w - 1), h->hq + 4 * w.hq_w, t2tmp, w.cout, w.cin, s);
      wrec[i].hq = w.hq_w;
      all.push_back(&wrec[i]);
    }
  }
  const int rc_flags = read_hx_flags(h->hq, h->n_hq, all, s);  // (synchronises: the temporary is free)
  if (t2tmp) (void)hipFree(t2tmp);
  if (rc_flags != RGFM_OK) return bail(RGFM_EHIP, "reading the fp16 scale records failed");
  for (size_t i = 0; i < h->up.size(); ++i) h->up[i].t2_ok = t2rec[i].hx_ok;
  for (size_t i = 0; i < wino.size(); ++i) wino[i]->w_ok = wrec[i].hx_ok;
  {
    // (norm_params_ok: convs behind a GroupNorm with out-of-window parameters leave the fp16 path here)
    std::vector<float> host(n_floats);
    if (hipMemcpyAsync(host.data(), h->params, n_floats * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return bail(RGFM_EHIP, "reading the parameters back failed");
    auto norm_ok = [&](size_t gw, size_t gb, int C) { return norm_params_ok(host, gw, gb, C); };
    for (auto* v : {&h->enc, &h->mid, &h->dec})
      for (ResW& r : *v) {
        if (!norm_ok(r.n1w, r.n1b, r.cin)) r.c1.hx_ok = false;
        if (!norm_ok(r.n2w, r.n2b, r.cout)) r.c2.hx_ok = false;
      }
  }
  launch_pack_conv_out(h->params + h->ocw, h->packed + h->ocw_pk, desc->in_channels, h->final_ch, s);
  // frequency table exp(-ln(1e4) * i / half) in fp32, as torch evaluates it (unet_flexible.py:28-31)
  const int half = h->mc / 2;
  std::vector<float> fr(half);
  const float neg_log = (float)(-std::log(10000.0));
  for (int i = 0; i < half; ++i) fr[i] = std::exp(((float)i * neg_log) / (float)half);
  std::vector<TimeLinear> lin;
  for (const auto* v : {&h->enc, &h->mid, &h->dec})
    for (const ResW& r : *v) lin.push_back({(int)r.tw, (int)r.tb, r.cout, r.temb_off});
  h->nlin = (int)lin.size();
  if (hipMalloc(&h->freqs, half * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(freqs)");
  if (hipMalloc(&h->lin_dev, lin.size() * sizeof(TimeLinear)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(lin)");
  if (hipMemcpy(h->freqs, fr.data(), half * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->lin_dev, lin.data(), lin.size() * sizeof(TimeLinear), hipMemcpyHostToDevice) != hipSuccess)
    return bail(RGFM_EHIP, "hipMemcpy(tables)");
  *out = h;
  return RGFM_OK;
}

extern "C" void rgfm_unet_destroy(rgfm_unet* h) {
  if (!h) return;
  if (h->params) (void)hipFree(h->params);
  if (h->packed) (void)hipFree(h->packed);
  if (h->packedh) (void)hipFree(h->packedh);
  if (h->hq) (void)hipFree(h->hq);
  if (h->packed3) (void)hipFree(h->packed3);
  if (h->freqs) (void)hipFree(h->freqs);
  if (h->lin_dev) (void)hipFree(h->lin_dev);
  if (h->range_flag) (void)hipFree(h->range_flag);
  delete h;
}

extern "C" int rgfm_unet_set_conv_mode(rgfm_unet* h, int mode) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  if (int rc = check_conv_mode(mode)) return rc;
  h->conv_mode = mode;
  return RGFM_OK;
}
extern "C" int rgfm_unet_range_flag(rgfm_unet* h, int* flagged, int reset, rgfm_stream_t stream) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  return read_flag_word(h->range_flag, flagged, reset, (hipStream_t)stream);
}

extern "C" int rgfm_unet_workspace_bytes(const rgfm_unet* h, int batch, size_t* bytes) {
  if (!h || !bytes || batch < 1) return fail(RGFM_EINVAL, "bad argument");
  *bytes = unet_eval_bytes(const_cast<rgfm_unet*>(h), batch) + table_bytes(h, batch) + counter_bytes(batch);
  return RGFM_OK;
}

extern "C" int rgfm_unet_forward(rgfm_unet* h, const float* x, const float* t_dev, int t_count, float* v_out,
                                 int batch, void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!h || !x || !t_dev || !v_out || !ws) return fail(RGFM_EINVAL, "null argument");
  if (batch < 1 || (t_count != 1 && t_count != batch)) return fail(RGFM_EINVAL, "t_count must be 1 or batch");
  hipStream_t s = (hipStream_t)stream;
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  size_t need = 0;
  rgfm_unet_workspace_bytes(h, batch, &need);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  float* table = b.f((size_t)t_count * h->temb_total);
  unsigned* cnt = reinterpret_cast<unsigned*>(b.f(batch));
  HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)batch * sizeof(unsigned), s));
  launch_time_table(h, t_dev, 1, 0, t_count, table, s);
  UNetRun r{h, batch, &b, s, table, t_count == batch ? 1 : 0, false};
  r.fin_counter = cnt;
  int rc = r.run(x, v_out, nullptr, 0.f);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_unet_time_embedding(rgfm_unet* h, const float* t_dev, int t_count, float* emb_out, void* ws,
                                        size_t ws_bytes, rgfm_stream_t stream) {
  if (!h || !t_dev || !emb_out || !ws || t_count < 1) return fail(RGFM_EINVAL, "bad argument");
  if (table_bytes(h, t_count) > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small");
  TimeEmbedArgs a{};
  a.params = h->params, a.freqs = h->freqs, a.mc = h->mc, a.temb = h->temb;
  a.te0w = (int)h->te0w, a.te0b = (int)h->te0b, a.te2w = (int)h->te2w, a.te2b = (int)h->te2b;
  a.lin = h->lin_dev, a.nlin = h->nlin, a.total = h->temb_total;
  a.t_dev = t_dev, a.num_steps = 1, a.step_begin = 0, a.table = reinterpret_cast<float*>(ws), a.emb_out = emb_out;
  launch_time_embed(a, t_count, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_unet_set_trace(rgfm_unet* h, int enable) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  h->trace = enable != 0;
  h->acts.clear();
  return RGFM_OK;
}

namespace {
void act_table(const rgfm_unet* h, std::vector<std::pair<int, int>>& t) {
  const rgfm_unet_desc& d = h->d;
  int S = d.img_size, ch = h->mc;
  t.push_back({ch, S});
  for (int l = 0; l < d.num_levels; ++l) {
    const int oc = h->mc * d.channel_mult[l];
    for (int r = 0; r < d.num_res_blocks; ++r) {
      t.push_back({oc, S});
      t.push_back({oc, S});
      ch = oc;
    }
    if (l < d.num_levels - 1) {
      S /= 2;
      t.push_back({ch, S});
    }
  }
  for (int i = 0; i < 4; ++i) t.push_back({ch, S});
  for (int l = d.num_levels - 1; l >= 0; --l) {
    const int oc = h->mc * d.channel_mult[l];
    for (int i = 0; i < d.num_res_blocks + 1; ++i) {
      t.push_back({oc, S});
      t.push_back({oc, S});
      ch = oc;
    }
    if (l > 0) {
      S *= 2;
      t.push_back({ch, S});
    }
  }
  t.push_back({d.in_channels, S});
}
}  // namespace

extern "C" int rgfm_unet_p_handovers(const rgfm_unet* h, int* blocks) {
  if (!h || !blocks) return fail(RGFM_EINVAL, "null argument");
  *blocks = h->p_handovers;
  return RGFM_OK;
}
extern "C" int rgfm_unet_wino_convs(const rgfm_unet* h, int* convs) {
  if (!h || !convs) return fail(RGFM_EINVAL, "null argument");
  *convs = h->wino_convs;
  return RGFM_OK;
}

extern "C" int rgfm_unet_num_activations(const rgfm_unet* h, int* n) {
  if (!h || !n) return fail(RGFM_EINVAL, "null argument");
  std::vector<std::pair<int, int>> t;
  act_table(h, t);
  *n = (int)t.size();
  return RGFM_OK;
}

extern "C" int rgfm_unet_activation_shape(const rgfm_unet* h, int index, int* channels, int* height, int* width) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  std::vector<std::pair<int, int>> t;
  act_table(h, t);
  if (index < 0 || index >= (int)t.size()) return fail(RGFM_EINVAL, "activation index out of range");
  *channels = t[index].first;
  *height = *width = t[index].second;
  return RGFM_OK;
}

extern "C" int rgfm_unet_read_activation(rgfm_unet* h, int index, int batch, const void* ws, float* out_dev,
                                         rgfm_stream_t stream) {
  (void)ws;
  if (!h || !out_dev) return fail(RGFM_EINVAL, "null argument");
  if (index < 0 || index >= (int)h->acts.size()) return fail(RGFM_EINVAL, "no traced activation %d (run a forward in trace mode first)", index);
  const auto& a = h->acts[index];
  hipStream_t s = (hipStream_t)stream;
  if (a.nchw) HIP_TRY(hipMemcpyAsync(out_dev, a.data, (size_t)batch * a.C * a.S * a.S * sizeof(float), hipMemcpyDeviceToDevice, s));
  else launch_nhwc_to_nchw(a.data, out_dev, batch, a.C, a.S * a.S, s);
  return RGFM_OK;
}
