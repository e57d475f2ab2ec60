// api_sampler.cpp -- the Euler samplers over U-Net handles and the stand-alone guidance block (C ABI: include/rgfm.h).
#include "rgfm_host.h"

// ================================================================== samplers
extern "C" int rgfm_sample_single_workspace_bytes(const rgfm_unet* h, int batch, size_t* bytes) {
  if (!h || !bytes || batch < 1) return fail(RGFM_EINVAL, "bad argument");
  // the time table is sized for up to 4096 steps per call
  *bytes = unet_eval_bytes(const_cast<rgfm_unet*>(h), batch) + table_bytes(h, 4096) + counter_bytes(batch);
  return RGFM_OK;
}

extern "C" int rgfm_sample_single(rgfm_unet* h, float* x_inout, int batch, int num_steps, int step_begin,
                                  int step_end, void* ws, size_t ws_bytes, rgfm_stream_t stream) {
  refresh_modes();
  if (!h || !x_inout || !ws) return fail(RGFM_EINVAL, "null argument");
  if (batch < 1 || num_steps < 1 || step_begin < 0 || step_end > num_steps || step_begin > step_end)
    return fail(RGFM_EINVAL, "bad step range [%d,%d) of %d", step_begin, step_end, num_steps);
  const int ns = step_end - step_begin;
  if (ns > 4096) return fail(RGFM_EINVAL, "at most 4096 steps per call");
  size_t need = 0;
  rgfm_sample_single_workspace_bytes(h, batch, &need);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  if (ns == 0) return RGFM_OK;
  hipStream_t s = (hipStream_t)stream;
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  float* table = b.f((size_t)4096 * h->temb_total);
  unsigned* cnt = reinterpret_cast<unsigned*>(b.f(batch));
  HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)batch * sizeof(unsigned), s));
  launch_time_table(h, nullptr, num_steps, step_begin, ns, table, s);
  const size_t mark = b.off;
  const float dt = (float)(1.0 / (double)num_steps);
  for (int i = 0; i < ns; ++i) {
    b.off = mark;
    UNetRun r{h, batch, &b, s, table + (size_t)i * h->temb_total, 0, false};
    r.fin_counter = cnt;
    int rc = r.run(x_inout, nullptr, x_inout, dt);
    if (rc) return rc;
  }
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_guidance_workspace_bytes(int batch, int n_mc, size_t* bytes) {
  if (!bytes || batch < 1 || n_mc < 1) return fail(RGFM_EINVAL, "bad argument");
  *bytes = guid_scratch_bytes(batch, n_mc);
  return RGFM_OK;
}

extern "C" int rgfm_guidance_apply(const float* x, const float* y, float* vx, float* vy, const float* mc_x1,
                                   const float* mc_y1, const float* mc_ratios, int batch, int n_mc, int dim_x,
                                   int dim_y, double t, double gamma, float* weights_out, void* ws,
                                   size_t ws_bytes, rgfm_stream_t stream) {
  if (!x || !y || !vx || !vy || !mc_x1 || !mc_y1 || !mc_ratios || !ws) return fail(RGFM_EINVAL, "null argument");
  size_t need = 0;
  int rc = rgfm_guidance_workspace_bytes(batch, n_mc, &need);
  if (rc) return rc;
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  if ((rc = ensure_init())) return rc;
  rc = guidance_launch(x, y, vx, vy, mc_x1, mc_y1, mc_ratios, batch, n_mc, dim_x, dim_y, t, gamma, (float*)ws,
                       weights_out, nullptr, nullptr, 0.f, (hipStream_t)stream);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}

extern "C" int rgfm_sample_pair_workspace_bytes(const rgfm_unet* hx, const rgfm_unet* hy, int batch, int n_mc,
                                                size_t* bytes) {
  if (!hx || !hy || !bytes || batch < 1 || n_mc < 0) return fail(RGFM_EINVAL, "bad argument");
  const size_t ex = unet_eval_bytes(const_cast<rgfm_unet*>(hx), batch);
  const size_t ey = unet_eval_bytes(const_cast<rgfm_unet*>(hy), batch);
  const size_t dx = (size_t)hx->d.in_channels * hx->d.img_size * hx->d.img_size;
  const size_t dy = (size_t)hy->d.in_channels * hy->d.img_size * hy->d.img_size;
  size_t total = table_bytes(hx, 4096) + table_bytes(hy, 4096) + ex + ey + 2 * counter_bytes(batch);  // the two nets run concurrently
  total += ((batch * dx * 4 + 255) & ~(size_t)255) + ((batch * dy * 4 + 255) & ~(size_t)255);
  total += guid_scratch_bytes(batch, n_mc);
  total += 256 + (size_t)4096 * 4 * sizeof(float);  // step counter + per-step guidance scalars (graph replay)
  *bytes = total;
  return RGFM_OK;
}

extern "C" int rgfm_sample_pair(rgfm_unet* hx, rgfm_unet* hy, float* x_inout, float* y_inout, const float* mc_x1,
                                const float* mc_y1, const float* mc_ratios, int n_mc, int batch, int num_steps,
                                double gamma, int step_begin, int step_end, void* ws, size_t ws_bytes,
                                rgfm_stream_t stream) {
  refresh_modes();
  if (!hx || !hy || !x_inout || !y_inout || !ws) return fail(RGFM_EINVAL, "null argument");
  if (n_mc < 0 || (n_mc > 0 && (!mc_x1 || !mc_y1 || !mc_ratios))) return fail(RGFM_EINVAL, "MC set missing");
  if (batch < 1 || num_steps < 1 || step_begin < 0 || step_end > num_steps || step_begin > step_end)
    return fail(RGFM_EINVAL, "bad step range [%d,%d) of %d", step_begin, step_end, num_steps);
  const int ns = step_end - step_begin;
  if (ns > 4096) return fail(RGFM_EINVAL, "at most 4096 steps per call");
  size_t need = 0;
  rgfm_sample_pair_workspace_bytes(hx, hy, batch, n_mc, &need);
  if (need > ws_bytes) return fail(RGFM_ENOMEM, "workspace too small: %zu < %zu", ws_bytes, need);
  if (ns == 0) return RGFM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int dx = hx->d.in_channels * hx->d.img_size * hx->d.img_size;
  const int dy = hy->d.in_channels * hy->d.img_size * hy->d.img_size;
  Bump b;
  b.base = (char*)ws, b.cap = ws_bytes, b.dry = false;
  float* tx = b.f((size_t)4096 * hx->temb_total);
  float* ty = b.f((size_t)4096 * hy->temb_total);
  float* vx = b.f((size_t)batch * dx);
  float* vy = b.f((size_t)batch * dy);
  float* logp = b.f(guid_scratch_bytes(batch, n_mc) / sizeof(float));
  unsigned* cnt_x = reinterpret_cast<unsigned*>(b.f(batch));
  unsigned* cnt_y = reinterpret_cast<unsigned*>(b.f(batch));
  float* gstate = b.f(64 + (size_t)4096 * 4);
  HIP_TRY(hipMemsetAsync(cnt_x, 0, (size_t)batch * sizeof(unsigned), s));
  HIP_TRY(hipMemsetAsync(cnt_y, 0, (size_t)batch * sizeof(unsigned), s));
  launch_time_table(hx, nullptr, num_steps, step_begin, ns, tx, s);
  launch_time_table(hy, nullptr, num_steps, step_begin, ns, ty, s);
  const size_t mark_x = b.off;
  const size_t mark_y = mark_x + unet_eval_bytes(hx, batch);
  // (step: the device-side step counter of the graph-replay path -- the time-table row is then chosen on the device)
  auto eval_x = [&](int i, hipStream_t st, float* v_out, float* x_state, float dt, const int* step) {
    b.off = mark_x;
    UNetRun r{hx, batch, &b, st, step ? tx : tx + (size_t)i * hx->temb_total, 0, false};
    r.fin_counter = cnt_x, r.step_ptr = step;
    return r.run(x_inout, v_out, x_state, dt);
  };
  auto eval_y = [&](int i, hipStream_t st, float* v_out, float* y_state, float dt, const int* step) {
    b.off = mark_y;
    UNetRun r{hy, batch, &b, st, step ? ty : ty + (size_t)i * hy->temb_total, 0, false};
    r.fin_counter = cnt_y, r.step_ptr = step;
    return r.run(y_inout, v_out, y_state, dt);
  };
  return pair_loop(eval_x, eval_y, x_inout, y_inout, mc_x1, mc_y1, mc_ratios, n_mc, batch, num_steps, gamma,
                   step_begin, ns, dx, dy, vx, vy, logp, s, gstate);
}
