// api_fmnet.cpp -- FlowMatchingModel ("--model original"): handle, forward, samplers (C ABI: include/rgfm.h).
#include "rgfm_host.h"

// ================================================================== FlowMatchingModel ("--model original")
// Encoder-decoder velocity net of src/models/flow_matching.py:34-173, 1x28x28 images.
struct rgfm_fmnet {
  rgfm_fmnet_desc d;
  float* params = nullptr;
  float* packed = nullptr;  // packed conv / deconv weights + re-indexed Linear weights
  unsigned short* packed3 = nullptr;  // 3-plane bf16 conv / deconv weights (conv_mfma_bx3.hip)
  unsigned short* packedh = nullptr;  // 2-plane scaled fp16 conv / deconv weights + scale records (conv_mfma_hx2.hip)
  float* hq = nullptr;
  unsigned* range_flag = nullptr;  // this handle's range-flag word
  int conv_mode = -1;              // rgfm_fmnet_set_conv_mode: -1 = RGFM_CONV from the environment
  float* freqs = nullptr;
  size_t n_params = 0, n_packed = 0, n_packed3 = 0, n_packedh = 0;
  int n_hq = 0;
  size_t c1w = 0, c1b = 0;           // encoder.conv1 (reference layout, conv_in kernel)
  size_t egw[4], egb[4];             // encoder.gn1..4
  ConvW ec[3];                       // encoder.conv2..4
  size_t fcw = 0, fcb = 0, fc_pk = 0;
  size_t f1w = 0, f1b = 0, f1w_pk = 0, f1b_pk = 0;
  ConvW d1, d2;                      // decoder.deconv1/2 (taps = 16 raw, packed per parity)
  size_t dgw[3], dgb[3];             // decoder.gn1..3
  ConvW c3;                          // decoder.conv3
  size_t cow = 0, cob = 0, cow_pk = 0;  // decoder.conv_out (raw; re-laid out for conv_out_kernel in `packed`)
};

namespace {

constexpr int FM_S = 28, FM_P = 49, FM_CF = 256;  // image size; 7x7 bottleneck pixels x 256 channels
constexpr int FM_FC_SPLITS = 14;                  // 12544/16 = 784 K-chunks = 14 x 56

// state_dict order of FlowMatchingModel (flow_matching.py:43-54, :88-98, :147-151)
size_t plan_fmnet(const rgfm_fmnet_desc& d, rgfm_fmnet* h) {
  Cursor c, pk, p3, ph;
  int nhq = 0;
  rgfm_fmnet t;
  const int F = d.feature_dim, T = d.time_emb_dim;
  auto conv = [&](int cin, int cout, int taps) {
    ConvW w;
    w.cin = cin, w.cout = cout, w.taps = taps;
    w.w_raw = c.take((size_t)cout * cin * taps);
    w.b = c.take(cout);
    w.w_pk = pk.take((size_t)cout * cin * taps);
    w.w_bx3 = p3.take((size_t)cout * cin * taps * 3);
    w.w_hx2 = ph.take((size_t)cout * cin * taps * 2);
    w.hq = nhq++;
    return w;
  };
  t.c1w = c.take((size_t)32 * d.img_channels * 9), t.c1b = c.take(32);
  t.egw[0] = c.take(32), t.egb[0] = c.take(32);
  const int ech[4] = {32, 64, 128, 256};
  for (int i = 1; i < 4; ++i) {
    t.ec[i - 1] = conv(ech[i - 1], ech[i], 9);
    t.egw[i] = c.take(ech[i]), t.egb[i] = c.take(ech[i]);
  }
  t.fcw = c.take((size_t)F * FM_CF * FM_P), t.fcb = c.take(F);
  t.fc_pk = pk.take((size_t)F * FM_CF * FM_P);
  t.f1w = c.take((size_t)FM_CF * FM_P * (F + T)), t.f1b = c.take((size_t)FM_CF * FM_P);
  t.f1w_pk = pk.take((size_t)FM_CF * FM_P * (F + T)), t.f1b_pk = pk.take((size_t)FM_CF * FM_P);
  t.d1 = conv(256, 128, 16);
  t.dgw[0] = c.take(128), t.dgb[0] = c.take(128);
  t.d2 = conv(128, 64, 16);
  t.dgw[1] = c.take(64), t.dgb[1] = c.take(64);
  t.c3 = conv(64, 32, 9);
  t.dgw[2] = c.take(32), t.dgb[2] = c.take(32);
  t.cow = c.take((size_t)d.img_channels * 32 * 9), t.cob = c.take(d.img_channels);
  t.cow_pk = pk.take((size_t)d.img_channels * 32 * 9);
  if (h) {
    float *pa = h->params, *pp = h->packed, *fr = h->freqs, *hqp = h->hq;
    unsigned short *p3p = h->packed3, *php = h->packedh;
    unsigned* rf = h->range_flag;
    const int cm = h->conv_mode;
    *h = t;
    h->d = d, h->params = pa, h->packed = pp, h->freqs = fr, h->packed3 = p3p, h->packedh = php, h->hq = hqp, h->range_flag = rf;
    h->conv_mode = cm;
    h->n_packed = pk.off, h->n_packed3 = p3.off, h->n_packedh = ph.off, h->n_hq = nhq;
  }
  return c.off;
}

int check_fm_desc(const rgfm_fmnet_desc* d) {
  if (!d) return fail(RGFM_EINVAL, "null descriptor");
  if (d->img_channels != 1) return fail(RGFM_EINVAL, "FlowMatchingModel: img_channels must be 1");
  if (d->feature_dim < 64 || d->feature_dim % 64 || d->feature_dim > 1024)
    return fail(RGFM_EINVAL, "feature_dim must be a multiple of 64 in 64..1024");
  if (d->time_emb_dim < 16 || d->time_emb_dim % 16 || d->time_emb_dim > 1024)
    return fail(RGFM_EINVAL, "time_emb_dim must be a multiple of 16 in 16..1024");
  return RGFM_OK;
}

struct FmRun {
  rgfm_fmnet* h;
  int B;
  Bump* ws;
  hipStream_t s;
  bool dry;
  const float* t_dev;  // explicit times (t_count 1 or B) or null: t of sampler step `step`
  int t_count, num_steps, step;
  unsigned* fin_counter = nullptr;  // see UNetRun
  PendingConv pend{};

  struct Map {  // NHWC activation + GroupNorm partials; rep 4 = written by a CONV_T2 launch over an (S/2)^2 raster
    float* data = nullptr;
    float* stats = nullptr;
    int C = 0, S = 0, rep = 1;
  };
  Map new_map(int C, int S, int rep) {
    Map m;
    m.C = C, m.S = S, m.rep = rep;
    const TileGeom g = make_geom(rep == 4 ? S / 2 : S, rep == 4 ? S / 2 : S);
    m.data = ws->f((size_t)B * S * S * C);
    m.stats = ws->f((size_t)B * g.nparts * rep * C * 2);
    return m;
  }
  float* finalize(const Map& a, size_t gamma, size_t beta, float* ab = nullptr) {
    if (!ab) ab = ws->f((size_t)B * a.C * 2);
    if (dry) return ab;
    const bool fused = try_fuse_finalize(pend, a.data, nullptr, 0, h->params + gamma, h->params + beta, ab, fin_counter);
    flush_conv(pend, s);
    if (fused) return ab;
    GnFinalizeArgs f{};
    f.stats0 = a.stats, f.C0 = a.C, f.groups = 8;
    f.gamma = h->params + gamma, f.beta = h->params + beta;
    f.ab = ab, f.B = B, f.rep = a.rep;
    const int sg = a.rep == 4 ? a.S / 2 : a.S;
    f.g = make_geom(sg, sg);
    ProfScope p(RGFM_KCLASS_OTHER, 0, s);
    launch_gn_finalize(f, s);
    return ab;
  }
  Map conv(const Map& a, const NormRef* norm, const ConvW& w, int mode) {
    const int So = mode == CONV_S2 ? (a.S + 1) / 2 : (mode == CONV_T2 ? a.S * 2 : a.S);
    Map o = new_map(w.cout, So, mode == CONV_T2 ? 4 : 1);
    float* ab_buf = norm ? ws->f((size_t)B * a.C * 2) : nullptr;  // used by the table path only
    if (dry) return o;
    ConvArgs c{};
    c.in0 = a.data, c.C0 = a.C, c.Hin = c.Win = a.S, c.ab = nullptr;
    c.wpk = h->packed + w.w_pk, c.wpk3 = h->packed3 + w.w_bx3, c.bias = h->params + w.b;
    fill_hx2(c, h->packedh, h->hq, h->range_flag, w, nullptr);
    c.out = o.data, c.stats_out = o.stats, c.B = B, c.Cout = w.cout;
    const int sg = mode == CONV_T2 ? a.S : So;  // raster the tiles walk
    c.g = make_geom(sg, sg);
    c.halo_px = mode == CONV_S2 ? c.g.spt * (2 * c.g.th + 1) * (2 * c.g.W + 1) : c.g.spt * (c.g.th + 2) * (c.g.W + 2);
    const double fl = mode == CONV_T2 ? conv_flops(B, 4 * sg * sg, w.cout, 4 * w.cin) : conv_flops(B, So * So, w.cout, 9 * w.cin);
    if (norm) {
      const int gs = a.rep == 4 ? a.S / 2 : a.S;  // raster the statistics parts of `a` refer to
      const TileGeom gg = make_geom(gs, gs);
      if (!try_consumer_gn(c, mode, a.stats, nullptr, gg.nparts * a.rep, gg, h->params + norm->gamma,
                           h->params + norm->beta))
        c.ab = finalize(a, norm->gamma, norm->beta, ab_buf);
    }
    flush_conv(pend, s);
    pend.valid = true, pend.c = c, pend.mode = mode, pend.flops = fl;
    return o;
  }

  // FlowMatchingModel.forward (flow_matching.py:153-173)
  int run(const float* x, float* v_out, float* x_state, float dt) {
    ModeScope mode_scope(h->conv_mode);
    const int F = h->d.feature_dim, T = h->d.time_emb_dim;
    // ImageEncoder.forward (:56-72)
    Map cur = new_map(32, FM_S, 1);
    if (!dry) {
      ConvInArgs ci{};
      ci.x = x, ci.w = h->params + h->c1w, ci.bias = h->params + h->c1b;
      ci.out = cur.data, ci.stats_out = cur.stats, ci.B = B, ci.C0 = 32, ci.g = make_geom(FM_S, FM_S);
      ProfScope p(RGFM_KCLASS_OTHER, 0, s);
      launch_conv_in(ci, 1, s);
    }
    const int modes[3] = {CONV_S2, CONV_S2, CONV_S1};
    for (int i = 0; i < 3; ++i) {
      const NormRef nr{h->egw[i], h->egb[i]};
      cur = conv(cur, &nr, h->ec[i], modes[i]);
    }
    float* ab4 = finalize(cur, h->egw[3], h->egb[3]);
    float* comb = ws->f((size_t)B * (F + T));  // torch.cat([features, t_emb], dim=1) (:111)
    float* part = ws->f((size_t)FM_FC_SPLITS * B * F);
    float* d0 = ws->f((size_t)B * FM_P * FM_CF);
    if (!dry) {
      flush_conv(pend, s);
      ProfScope p(RGFM_KCLASS_OTHER, 0, s);
      launch_linear_mfma_splitk(cur.data, ab4, FM_CF, h->packed + h->fc_pk, h->params + h->fcb, comb, part,
                                FM_FC_SPLITS, B, FM_P * FM_CF, F, F + T, s);
      launch_fm_time_embed(t_dev, t_count, num_steps, step, h->freqs, comb, B, T, F + T, F, s);
      // VelocityDecoder.forward (:100-124); fc1 rows re-indexed so the result is the NHWC 7x7x256 map
      launch_linear_mfma(comb, h->packed + h->f1w_pk, h->packed + h->f1b_pk, d0, B, F + T, FM_P * FM_CF, F + T,
                         FM_P * FM_CF, s);
      // deconv1 stages this map RAW (no norm in front, flow_matching.py:113-116): the low side of the two-plane
      // representation is checked here, as a producing conv's epilogue would (ConvArgs::small_check)
      if (g_modes.conv == CONV_ARITH_HX2 && h->d1.hx_ok) launch_range_low_check(d0, B, FM_P, FM_CF, h->range_flag, s);
    }
    Map m0;
    m0.data = d0, m0.C = FM_CF, m0.S = 7;
    const NormRef g1{h->dgw[0], h->dgb[0]}, g2{h->dgw[1], h->dgb[1]};
    Map u1 = conv(m0, nullptr, h->d1, CONV_T2);
    Map u2 = conv(u1, &g1, h->d2, CONV_T2);
    Map u3 = conv(u2, &g2, h->c3, CONV_S1);
    float* ab3 = finalize(u3, h->dgw[2], h->dgb[2]);
    if (!dry) {
      flush_conv(pend, s);
      ConvOutArgs co{};
      co.in = u3.data, co.ab = ab3, co.w = h->packed + h->cow_pk, co.bias = h->params + h->cob;
      co.v_out = v_out, co.x_state = x_state, co.dt = dt, co.B = B, co.Cin = 32;
      co.g = make_geom(FM_S, FM_S);
      co.halo_px = co.g.spt * (co.g.th + 2) * (co.g.W + 2);
      ProfScope p(RGFM_KCLASS_OTHER, 0, s);
      launch_conv_out(co, 1, s);
    }
    return RGFM_OK;
  }
};

size_t fm_eval_bytes(rgfm_fmnet* h, int B) {
  Bump b;
  FmRun r{h, B, &b, nullptr, true, nullptr, 1, 1, 0};
  r.run(nullptr, nullptr, nullptr, 0.f);
  return b.off;
}

}  // namespace

extern "C" int rgfm_fmnet_param_floats(const rgfm_fmnet_desc* desc, size_t* n_floats) {
  int rc = check_fm_desc(desc);
  if (rc) return rc;
  if (!n_floats) return fail(RGFM_EINVAL, "null output");
  *n_floats = plan_fmnet(*desc, nullptr);
  return RGFM_OK;
}

extern "C" int rgfm_fmnet_create(const rgfm_fmnet_desc* desc, const float* params_dev, size_t n_floats,
                                 rgfm_stream_t stream, rgfm_fmnet** out) {
  int rc = check_fm_desc(desc);
  if (rc) return rc;
  if (!params_dev || !out) return fail(RGFM_EINVAL, "null argument");
  if ((rc = ensure_init())) return rc;
  hipStream_t s = (hipStream_t)stream;
  rgfm_fmnet* h = new rgfm_fmnet();
  h->n_params = plan_fmnet(*desc, h);
  if (h->n_params != n_floats) {
    const size_t want = h->n_params;
    delete h;
    return fail(RGFM_EINVAL, "parameter blob has %zu floats, architecture needs %zu", n_floats, want);
  }
  auto bail = [&](int code, const char* what) {
    rgfm_fmnet_destroy(h);
    return fail(code, "%s", what);
  };
  const int F = desc->feature_dim, T = desc->time_emb_dim, half = T / 2;
  if (hipMalloc(&h->params, n_floats * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(params)");
  if (hipMalloc(&h->packed, (h->n_packed + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packed)");
  if (hipMalloc(&h->freqs, half * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(freqs)");
  if (hipMalloc(&h->packed3, (h->n_packed3 + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packed3)");
  if (hipMalloc(&h->packedh, (h->n_packedh + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packedh)");
  if (hipMalloc(&h->hq, ((size_t)h->n_hq * 4 + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(hq)");
  if (alloc_flag_word(&h->range_flag) != RGFM_OK) return bail(RGFM_ENOMEM, "hipMalloc(range flag)");
  if (hipMemcpyAsync(h->params, params_dev, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return bail(RGFM_EHIP, "hipMemcpyAsync(params)");
  {
    auto packh = [&](const ConvW& w, int mode) {
      launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + w.w_hx2, h->hq + 4 * w.hq, w.cout, w.cin, w.taps, mode, s);
    };
    packh(h->ec[0], CONV_S2), packh(h->ec[1], CONV_S2), packh(h->ec[2], CONV_S1), packh(h->c3, CONV_S1);
    packh(h->d1, CONV_T2), packh(h->d2, CONV_T2);
    std::vector<ConvW*> all{&h->ec[0], &h->ec[1], &h->ec[2], &h->c3, &h->d1, &h->d2};
    if (read_hx_flags(h->hq, h->n_hq, all, s) != RGFM_OK) return bail(RGFM_EHIP, "reading the fp16 scale records failed");
    // (as rgfm_unet_create: a conv behind a GroupNorm with out-of-window parameters leaves the fp16 path)
    std::vector<float> host(n_floats);
    if (hipMemcpyAsync(host.data(), h->params, n_floats * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return bail(RGFM_EHIP, "reading the parameters back failed");
    const int ech[3] = {32, 64, 128};
    for (int i = 0; i < 3; ++i)
      if (!norm_params_ok(host, h->egw[i], h->egb[i], ech[i])) h->ec[i].hx_ok = false;
    if (!norm_params_ok(host, h->dgw[0], h->dgb[0], 128)) h->d2.hx_ok = false;
    if (!norm_params_ok(host, h->dgw[1], h->dgb[1], 64)) h->c3.hx_ok = false;
  }
  for (int i = 0; i < 3; ++i) {  // encoder conv2 / conv3 are stride 2 (phase-ordered weights), conv4 stride 1
    const ConvW& w = h->ec[i];
    if (i < 2) launch_pack_conv_bx3_s2(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, s);
    else launch_pack_conv_bx3(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, 9, s);
  }
  launch_pack_conv_bx3(h->params + h->c3.w_raw, h->packed3 + h->c3.w_bx3, 32, 64, 9, s);
  launch_pack_deconv_bx3(h->params + h->d1.w_raw, h->packed3 + h->d1.w_bx3, 256, 128, s);
  launch_pack_deconv_bx3(h->params + h->d2.w_raw, h->packed3 + h->d2.w_bx3, 128, 64, s);
  for (const ConvW& w : h->ec) launch_pack_conv(h->params + w.w_raw, h->packed + w.w_pk, w.cout, w.cin, 9, nt32_of(w.cout), s);
  launch_pack_conv(h->params + h->c3.w_raw, h->packed + h->c3.w_pk, 32, 64, 9, 1, s);
  launch_pack_deconv(h->params + h->d1.w_raw, h->packed + h->d1.w_pk, 256, 128, nt32_of(128), s);
  launch_pack_deconv(h->params + h->d2.w_raw, h->packed + h->d2.w_pk, 128, 64, nt32_of(64), s);
  launch_pack_conv_out(h->params + h->cow, h->packed + h->cow_pk, desc->img_channels, 32, s);
  launch_permute_cols(h->params + h->fcw, h->packed + h->fc_pk, F, FM_CF, FM_P, s);
  launch_permute_rows(h->params + h->f1w, h->params + h->f1b, h->packed + h->f1w_pk, h->packed + h->f1b_pk, FM_CF, FM_P,
                      F + T, s);
  // exp(arange(half) * -(ln(1e4) / (half - 1))) in fp32, as torch evaluates it (flow_matching.py:25-27)
  std::vector<float> fr(half);
  const float neg = (float)(-(std::log(10000.0) / (double)(half - 1)));
  for (int i = 0; i < half; ++i) fr[i] = std::exp((float)i * neg);
  if (hipMemcpy(h->freqs, fr.data(), half * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return bail(RGFM_EHIP, "hipMemcpy(freqs)");
  *out = h;
  return RGFM_OK;
}

extern "C" void rgfm_fmnet_destroy(rgfm_fmnet* h) {
  if (!h) return;
  if (h->params) (void)hipFree(h->params);
  if (h->packed) (void)hipFree(h->packed);
  if (h->packed3) (void)hipFree(h->packed3);
  if (h->packedh) (void)hipFree(h->packedh);
  if (h->hq) (void)hipFree(h->hq);
  if (h->freqs) (void)hipFree(h->freqs);
  if (h->range_flag) (void)hipFree(h->range_flag);
  delete h;
}

extern "C" int rgfm_fmnet_set_conv_mode(rgfm_fmnet* h, int mode) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  if (int rc = check_conv_mode(mode)) return rc;
  h->conv_mode = mode;
  return RGFM_OK;
}
extern "C" int rgfm_fmnet_range_flag(rgfm_fmnet* h, int* flagged, int reset, rgfm_stream_t stream) {
  if (!h) return fail(RGFM_EINVAL, "null handle");
  return read_flag_word(h->range_flag, flagged, reset, (hipStream_t)stream);
}

extern "C" int rgfm_fmnet_workspace_bytes(const rgfm_fmnet* h, int batch, size_t* bytes) {
  if (!h || !bytes || batch < 1) return fail(RGFM_EINVAL, "bad argument");
  *bytes = fm_eval_bytes(const_cast<rgfm_fmnet*>(h), batch) + counter_bytes(batch);
  return RGFM_OK;
}

extern "C" int rgfm_fmnet_forward(rgfm_fmnet* h, const float* x, const float* t_dev, int t_count, float* v_out,
                                  int batch, void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!h || !x || !t_dev || !v_out || !ws) return fail(RGFM_EINVAL, "null argument");
  if (batch < 1 || (t_count != 1 && t_count != batch)) return fail(RGFM_EINVAL, "t_count must be 1 or batch");
  const size_t need = fm_eval_bytes(h, batch) + counter_bytes(batch);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  unsigned* cnt = reinterpret_cast<unsigned*>(b.f(batch));
  HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)batch * sizeof(unsigned), (hipStream_t)stream));
  FmRun r{h, batch, &b, (hipStream_t)stream, false, t_dev, t_count, 1, 0};
  r.fin_counter = cnt;
  int rc = r.run(x, v_out, nullptr, 0.f);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_fmnet_sample_single(rgfm_fmnet* h, float* x_inout, int batch, int num_steps, int step_begin,
                                        int step_end, void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!h || !x_inout || !ws) return fail(RGFM_EINVAL, "null argument");
  if (batch < 1 || num_steps < 1 || step_begin < 0 || step_end > num_steps || step_begin > step_end)
    return fail(RGFM_EINVAL, "bad step range [%d,%d) of %d", step_begin, step_end, num_steps);
  const size_t need = fm_eval_bytes(h, batch) + counter_bytes(batch);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  const float dt = (float)(1.0 / (double)num_steps);
  unsigned* cnt = nullptr;
  for (int st = step_begin; st < step_end; ++st) {
    Bump b;
    b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
    unsigned* c0 = reinterpret_cast<unsigned*>(b.f(batch));
    if (!cnt) {
      cnt = c0;
      HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)batch * sizeof(unsigned), (hipStream_t)stream));
    }
    FmRun r{h, batch, &b, (hipStream_t)stream, false, nullptr, 1, num_steps, st};
    r.fin_counter = cnt;
    int rc = r.run(x_inout, nullptr, x_inout, dt);
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_fmnet_sample_pair_workspace_bytes(const rgfm_fmnet* hx, const rgfm_fmnet* hy, int batch, int n_mc,
                                                      size_t* bytes) {
  if (!hx || !hy || !bytes || batch < 1 || n_mc < 0) return fail(RGFM_EINVAL, "bad argument");
  const size_t d = (size_t)FM_S * FM_S;
  size_t total = fm_eval_bytes(const_cast<rgfm_fmnet*>(hx), batch) + fm_eval_bytes(const_cast<rgfm_fmnet*>(hy), batch);
  total += 2 * ((batch * d * 4 + 255) & ~(size_t)255) + 2 * counter_bytes(batch);
  total += guid_scratch_bytes(batch, n_mc);
  *bytes = total;
  return RGFM_OK;
}

extern "C" int rgfm_fmnet_sample_pair(rgfm_fmnet* hx, rgfm_fmnet* hy, float* x_inout, float* y_inout,
                                      const float* mc_x1, const float* mc_y1, const float* mc_ratios, int n_mc,
                                      int batch, int num_steps, double gamma, int step_begin, int step_end, void* ws,
                                      size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!hx || !hy || !x_inout || !y_inout || !ws) return fail(RGFM_EINVAL, "null argument");
  if (n_mc < 0 || (n_mc > 0 && (!mc_x1 || !mc_y1 || !mc_ratios))) return fail(RGFM_EINVAL, "MC set missing");
  if (batch < 1 || num_steps < 1 || step_begin < 0 || step_end > num_steps || step_begin > step_end)
    return fail(RGFM_EINVAL, "bad step range [%d,%d) of %d", step_begin, step_end, num_steps);
  size_t need = 0;
  rgfm_fmnet_sample_pair_workspace_bytes(hx, hy, batch, n_mc, &need);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  const int ns = step_end - step_begin;
  if (ns == 0) return RGFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d = FM_S * FM_S;
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  float* vx = b.f((size_t)batch * d);
  float* vy = b.f((size_t)batch * d);
  float* logp = b.f(guid_scratch_bytes(batch, n_mc) / sizeof(float));
  unsigned* cnt_x = reinterpret_cast<unsigned*>(b.f(batch));
  unsigned* cnt_y = reinterpret_cast<unsigned*>(b.f(batch));
  HIP_TRY(hipMemsetAsync(cnt_x, 0, (size_t)batch * sizeof(unsigned), s));
  HIP_TRY(hipMemsetAsync(cnt_y, 0, (size_t)batch * sizeof(unsigned), s));
  const size_t mark_x = b.off;
  const size_t mark_y = mark_x + fm_eval_bytes(hx, batch);
  auto eval_x = [&](int i, hipStream_t st, float* v_out, float* x_state, float dt, const int*) {
    b.off = mark_x;
    FmRun r{hx, batch, &b, st, false, nullptr, 1, num_steps, step_begin + i};
    r.fin_counter = cnt_x;
    return r.run(x_inout, v_out, x_state, dt);
  };
  auto eval_y = [&](int i, hipStream_t st, float* v_out, float* y_state, float dt, const int*) {
    b.off = mark_y;
    FmRun r{hy, batch, &b, st, false, nullptr, 1, num_steps, step_begin + i};
    r.fin_counter = cnt_y;
    return r.run(y_inout, v_out, y_state, dt);
  };
  return pair_loop(eval_x, eval_y, x_inout, y_inout, mc_x1, mc_y1, mc_ratios, n_mc, batch, num_steps, gamma, step_begin,
                   ns, d, d, vx, vy, logp, s);
}
