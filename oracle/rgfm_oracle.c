/*
 * rgfm_oracle.c -- CPU restatement of the reference sampler path (see rgfm_oracle.h).
 * TEST INFRASTRUCTURE ONLY; parity pinned by tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference repo).  Layout NCHW fp32 like the reference; one OpenMP task per
 * batch row (rows never interact: per-sample GroupNorm, eval-mode BatchNorm).
 */
#include "rgfm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NOFMA __attribute__((optimize("fp-contract=off")))

int ro_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static float* fmalloc(size_t n) { return (float*)malloc((n ? n : 1) * sizeof(float)); }

/* ---------------------------------------------------------------- primitives */

/* F.silu */
static inline float silu(float v) { return v / (1.0f + expf(-v)); }

/* nn.Linear: y = W x + b, W [out,in] */
static void linear(const float* W, const float* b, const float* x, int in, int out, float* y) {
  for (int o = 0; o < out; ++o) {
    float acc = b ? b[o] : 0.0f;
    const float* w = W + (size_t)o * in;
    for (int i = 0; i < in; ++i) acc += w[i] * x[i];
    y[o] = acc;
  }
}

/* nn.Conv2d(k=3, padding=1, stride s) on one sample. in [Ci,H,W] -> out [Co,Ho,Wo]. */
static void conv3x3(const float* in, int Ci, int H, int W, const float* w, const float* b, int Co,
                    int stride, float* out) {
  const int Hp = H + 2, Wp = W + 2;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  float* pad = (float*)calloc((size_t)Ci * Hp * Wp, sizeof(float));
  for (int c = 0; c < Ci; ++c)
    for (int y = 0; y < H; ++y)
      memcpy(pad + ((size_t)c * Hp + y + 1) * Wp + 1, in + ((size_t)c * H + y) * W, W * sizeof(float));
  for (int co = 0; co < Co; ++co) {
    float* o = out + (size_t)co * Ho * Wo;
    const float bv = b ? b[co] : 0.0f;
    for (int i = 0; i < Ho * Wo; ++i) o[i] = bv;
    for (int ci = 0; ci < Ci; ++ci) {
      const float* wk = w + ((size_t)co * Ci + ci) * 9;
      const float* p = pad + (size_t)ci * Hp * Wp;
      if (stride == 1) {
        for (int y = 0; y < Ho; ++y) {
          const float* r0 = p + (size_t)y * Wp;
          const float* r1 = r0 + Wp;
          const float* r2 = r1 + Wp;
          float* orow = o + (size_t)y * Wo;
          for (int x = 0; x < Wo; ++x) {
            float a = orow[x];
            a += wk[0] * r0[x] + wk[1] * r0[x + 1] + wk[2] * r0[x + 2];
            a += wk[3] * r1[x] + wk[4] * r1[x + 1] + wk[5] * r1[x + 2];
            a += wk[6] * r2[x] + wk[7] * r2[x + 1] + wk[8] * r2[x + 2];
            orow[x] = a;
          }
        }
      } else {
        for (int y = 0; y < Ho; ++y)
          for (int x = 0; x < Wo; ++x) {
            float a = o[(size_t)y * Wo + x];
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx)
                a += wk[ky * 3 + kx] * p[(size_t)(y * stride + ky) * Wp + x * stride + kx];
            o[(size_t)y * Wo + x] = a;
          }
      }
    }
  }
  free(pad);
}

/* nn.Conv2d(k=1) */
static void conv1x1(const float* in, int Ci, int HW, const float* w, const float* b, int Co,
                    float* out) {
  for (int co = 0; co < Co; ++co) {
    float* o = out + (size_t)co * HW;
    for (int i = 0; i < HW; ++i) o[i] = b[co];
    for (int ci = 0; ci < Ci; ++ci) {
      const float wv = w[(size_t)co * Ci + ci];
      const float* p = in + (size_t)ci * HW;
      for (int i = 0; i < HW; ++i) o[i] += wv * p[i];
    }
  }
}

/* nn.GroupNorm(G, C), eps 1e-5, affine, optionally followed by SiLU; one sample. */
static void groupnorm(const float* in, int C, int HW, int G, const float* gamma, const float* beta,
                      int do_silu, float* out) {
  const int cpg = C / G;
  for (int g = 0; g < G; ++g) {
    const float* p = in + (size_t)g * cpg * HW;
    const size_t n = (size_t)cpg * HW;
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += p[i];
    const double mean = s / (double)n;
    double m2 = 0.0;
    for (size_t i = 0; i < n; ++i) {
      const double d = p[i] - mean;
      m2 += d * d;
    }
    const float rstd = (float)(1.0 / sqrt(m2 / (double)n + 1e-5));
    const float meanf = (float)mean;
    for (int c = 0; c < cpg; ++c) {
      const int ch = g * cpg + c;
      const float a = rstd * gamma[ch];
      const float bb = beta[ch] - meanf * a;
      const float* src = in + (size_t)ch * HW;
      float* dst = out + (size_t)ch * HW;
      if (do_silu)
        for (int i = 0; i < HW; ++i) dst[i] = silu(src[i] * a + bb);
      else
        for (int i = 0; i < HW; ++i) dst[i] = src[i] * a + bb;
    }
  }
}

/* F.interpolate(scale_factor=2, mode='nearest') */
static void upsample2(const float* in, int C, int H, int W, float* out) {
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < 2 * H; ++y)
      for (int x = 0; x < 2 * W; ++x)
        out[((size_t)c * 2 * H + y) * 2 * W + x] = in[((size_t)c * H + (y >> 1)) * W + (x >> 1)];
}

/* ---------------------------------------------------------------- U-Net */

typedef struct {
  const float* p;
} cursor;
static const float* take(cursor* c, size_t n) {
  const float* r = c->p;
  c->p += n;
  return r;
}

typedef struct {
  int cin, cout;
  const float *n1w, *n1b, *c1w, *c1b, *tw, *tb, *n2w, *n2b, *c2w, *c2b, *sw, *sb;
} resblock;

static void rb_take(cursor* c, resblock* r, int cin, int cout, int temb) {
  r->cin = cin;
  r->cout = cout;
  r->n1w = take(c, cin);
  r->n1b = take(c, cin);
  r->c1w = take(c, (size_t)cout * cin * 9);
  r->c1b = take(c, cout);
  r->tw = take(c, (size_t)cout * temb);
  r->tb = take(c, cout);
  r->n2w = take(c, cout);
  r->n2b = take(c, cout);
  r->c2w = take(c, (size_t)cout * cout * 9);
  r->c2b = take(c, cout);
  if (cin != cout) {
    r->sw = take(c, (size_t)cout * cin);
    r->sb = take(c, cout);
  } else {
    r->sw = r->sb = NULL;
  }
}

#define MAXB 32
typedef struct {
  int mc, temb, nenc, ndec, ndown, nup;
  const float *te0w, *te0b, *te2w, *te2b, *icw, *icb, *onw, *onb, *ocw, *ocb;
  resblock enc[MAXB], mid[2], dec[MAXB];
  const float *dw[4], *db[4], *uw[4], *ub[4];
  int dch[4], uch[4];
  size_t total;
} unet_plan;

/* Parameter registration order of FlexibleUNet.__init__ (src/models/unet_flexible.py:146-201);
 * UNetMNIST (src/models/unet.py:155-214) is identical. */
static void plan_unet(const ro_unet_desc* d, const float* params, unet_plan* P) {
  cursor c = {params};
  const int mc = d->model_channels, temb = 4 * mc;
  memset(P, 0, sizeof(*P));
  P->mc = mc;
  P->temb = temb;
  P->te0w = take(&c, (size_t)temb * mc);
  P->te0b = take(&c, temb);
  P->te2w = take(&c, (size_t)temb * temb);
  P->te2b = take(&c, temb);
  P->icw = take(&c, (size_t)mc * d->in_channels * 9);
  P->icb = take(&c, mc);
  int ch = mc, skips[64], ns = 0;
  skips[ns++] = ch;
  int down_ch[4], nd = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    const int oc = mc * d->channel_mult[l];
    for (int r = 0; r < d->num_res_blocks; ++r) {
      rb_take(&c, &P->enc[P->nenc++], ch, oc, temb);
      ch = oc;
      skips[ns++] = ch;
    }
    if (l < d->num_levels - 1) {
      down_ch[nd++] = ch;
      skips[ns++] = ch;
    }
  }
  for (int i = 0; i < nd; ++i) {
    P->dch[i] = down_ch[i];
    P->dw[i] = take(&c, (size_t)down_ch[i] * down_ch[i] * 9);
    P->db[i] = take(&c, down_ch[i]);
  }
  P->ndown = nd;
  rb_take(&c, &P->mid[0], ch, ch, temb);
  rb_take(&c, &P->mid[1], ch, ch, temb);
  int up_ch[4], nu = 0;
  for (int l = d->num_levels - 1; l >= 0; --l) {
    const int oc = mc * d->channel_mult[l];
    for (int i = 0; i < d->num_res_blocks + 1; ++i) {
      rb_take(&c, &P->dec[P->ndec++], ch + skips[--ns], oc, temb);
      ch = oc;
    }
    if (l > 0) up_ch[nu++] = ch;
  }
  for (int i = 0; i < nu; ++i) {
    P->uch[i] = up_ch[i];
    P->uw[i] = take(&c, (size_t)up_ch[i] * up_ch[i] * 9);
    P->ub[i] = take(&c, up_ch[i]);
  }
  P->nup = nu;
  P->onw = take(&c, ch);
  P->onb = take(&c, ch);
  P->ocw = take(&c, (size_t)d->in_channels * ch * 9);
  P->ocb = take(&c, d->in_channels);
  P->total = (size_t)(c.p - params);
}

size_t ro_unet_param_floats(const ro_unet_desc* d) {
  unet_plan P;
  plan_unet(d, NULL, &P);
  return P.total;
}

/* activation table: (channels, size) per index, in production order */
static int act_table(const ro_unet_desc* d, int* ch_out, int* sz_out) {
  const int mc = d->model_channels;
  int n = 0, ch = mc, sz = d->img_size;
#define PUSH(C, S)        \
  do {                    \
    if (ch_out) {         \
      ch_out[n] = (C);    \
      sz_out[n] = (S);    \
    }                     \
    ++n;                  \
  } while (0)
  PUSH(mc, sz);
  for (int l = 0; l < d->num_levels; ++l) {
    const int oc = mc * d->channel_mult[l];
    for (int r = 0; r < d->num_res_blocks; ++r) {
      PUSH(oc, sz);
      PUSH(oc, sz);
      ch = oc;
    }
    if (l < d->num_levels - 1) {
      sz = (sz - 1) / 2 + 1;
      PUSH(ch, sz);
    }
  }
  for (int i = 0; i < 2; ++i) {
    PUSH(ch, sz);
    PUSH(ch, sz);
  }
  for (int l = d->num_levels - 1; l >= 0; --l) {
    const int oc = mc * d->channel_mult[l];
    for (int i = 0; i < d->num_res_blocks + 1; ++i) {
      PUSH(oc, sz);
      PUSH(oc, sz);
      ch = oc;
    }
    if (l > 0) {
      sz *= 2;
      PUSH(ch, sz);
    }
  }
  PUSH(d->in_channels, sz);
#undef PUSH
  return n;
}

int ro_unet_num_activations(const ro_unet_desc* d) { return act_table(d, NULL, NULL); }

void ro_unet_activation_shape(const ro_unet_desc* d, int idx, int* c, int* h, int* w) {
  int ch[256], sz[256];
  act_table(d, ch, sz);
  *c = ch[idx];
  *h = *w = sz[idx];
}

/* timestep_embedding (src/models/unet_flexible.py:16-36): cos half first. */
void ro_timestep_embedding(const float* t, int n, int dim, float* out) {
  const int half = dim / 2;
  const float neg_log = (float)(-log(10000.0));
  for (int b = 0; b < n; ++b)
    for (int i = 0; i < half; ++i) {
      const float f = expf(((float)i * neg_log) / (float)half);
      const float a = t[b] * f;
      out[(size_t)b * dim + i] = cosf(a);
      out[(size_t)b * dim + half + i] = sinf(a);
    }
}

static void store_act(float** acts, int idx, int b, const float* src, size_t n) {
  if (acts && acts[idx]) memcpy(acts[idx] + (size_t)b * n, src, n * sizeof(float));
}

/* ResBlock.forward (src/models/unet_flexible.py:71-85), one sample.
 * x [cin,S,S] -> returns malloc'd [cout,S,S]. st = silu(t_emb). */
static float* resblock_fwd(const resblock* r, const float* x, int S, const float* st, int temb,
                           float** acts, int* ai, int b) {
  const int HW = S * S;
  const int G1 = r->cin < 8 ? r->cin : 8, G2 = r->cout < 8 ? r->cout : 8;
  float* a = fmalloc((size_t)r->cin * HW);
  groupnorm(x, r->cin, HW, G1, r->n1w, r->n1b, 1, a);
  float* h = fmalloc((size_t)r->cout * HW);
  conv3x3(a, r->cin, S, S, r->c1w, r->c1b, r->cout, 1, h);
  free(a);
  float* tv = fmalloc(r->cout);
  linear(r->tw, r->tb, st, temb, r->cout, tv);
  for (int c = 0; c < r->cout; ++c)
    for (int i = 0; i < HW; ++i) h[(size_t)c * HW + i] += tv[c];
  free(tv);
  store_act(acts, (*ai)++, b, h, (size_t)r->cout * HW);
  float* a2 = fmalloc((size_t)r->cout * HW);
  groupnorm(h, r->cout, HW, G2, r->n2w, r->n2b, 1, a2);
  conv3x3(a2, r->cout, S, S, r->c2w, r->c2b, r->cout, 1, h);
  free(a2);
  if (r->sw) {
    float* sk = fmalloc((size_t)r->cout * HW);
    conv1x1(x, r->cin, HW, r->sw, r->sb, r->cout, sk);
    for (size_t i = 0; i < (size_t)r->cout * HW; ++i) h[i] += sk[i];
    free(sk);
  } else {
    for (size_t i = 0; i < (size_t)r->cout * HW; ++i) h[i] += x[i];
  }
  store_act(acts, (*ai)++, b, h, (size_t)r->cout * HW);
  return h;
}

/* FlexibleUNet.forward (src/models/unet_flexible.py:203-261), one sample. */
static void unet_one(const ro_unet_desc* d, const unet_plan* P, const float* x, float t, float* out,
                     float** acts, int b) {
  const int mc = P->mc, temb = P->temb;
  int S = d->img_size, ai = 0;
  float emb[512], h1[2048], te[2048], st[2048];
  ro_timestep_embedding(&t, 1, mc, emb);
  linear(P->te0w, P->te0b, emb, mc, temb, h1);
  for (int i = 0; i < temb; ++i) h1[i] = silu(h1[i]);
  linear(P->te2w, P->te2b, h1, temb, temb, te);
  for (int i = 0; i < temb; ++i) st[i] = silu(te[i]); /* time_mlp's leading SiLU */

  float* hs[64];
  int hs_c[64], hs_s[64], ns = 0;
  int ch = mc;
  float* h = fmalloc((size_t)mc * S * S);
  conv3x3(x, d->in_channels, S, S, P->icw, P->icb, mc, 1, h);
  store_act(acts, ai++, b, h, (size_t)mc * S * S);
  hs[ns] = h, hs_c[ns] = ch, hs_s[ns] = S, ++ns;
  int e = 0;
  for (int l = 0; l < d->num_levels; ++l) {
    for (int r = 0; r < d->num_res_blocks; ++r) {
      h = resblock_fwd(&P->enc[e], h, S, st, temb, acts, &ai, b);
      ch = P->enc[e].cout;
      ++e;
      hs[ns] = h, hs_c[ns] = ch, hs_s[ns] = S, ++ns;
    }
    if (l < d->num_levels - 1) {
      const int So = (S - 1) / 2 + 1;
      float* dn = fmalloc((size_t)ch * So * So);
      conv3x3(h, ch, S, S, P->dw[l], P->db[l], ch, 2, dn);
      S = So;
      h = dn;
      store_act(acts, ai++, b, h, (size_t)ch * S * S);
      hs[ns] = h, hs_c[ns] = ch, hs_s[ns] = S, ++ns;
    }
  }
  /* every encoder tensor stays alive in hs (they are the decoder's skips) */
  float* m1 = resblock_fwd(&P->mid[0], h, S, st, temb, acts, &ai, b);
  float* m2 = resblock_fwd(&P->mid[1], m1, S, st, temb, acts, &ai, b);
  free(m1);
  h = m2;
  int di = 0, ui = 0;
  for (int l = d->num_levels - 1; l >= 0; --l) {
    for (int i = 0; i < d->num_res_blocks + 1; ++i) {
      --ns;
      const int sc = hs_c[ns];
      const size_t HW = (size_t)S * S;
      float* cat = fmalloc((size_t)(ch + sc) * HW);
      memcpy(cat, h, (size_t)ch * HW * sizeof(float));
      memcpy(cat + (size_t)ch * HW, hs[ns], (size_t)sc * HW * sizeof(float));
      free(h);
      free(hs[ns]);
      h = resblock_fwd(&P->dec[di], cat, S, st, temb, acts, &ai, b);
      free(cat);
      ch = P->dec[di].cout;
      ++di;
    }
    if (l > 0) {
      float* up = fmalloc((size_t)ch * 4 * S * S);
      upsample2(h, ch, S, S, up);
      free(h);
      S *= 2;
      h = fmalloc((size_t)ch * S * S);
      conv3x3(up, ch, S, S, P->uw[ui], P->ub[ui], ch, 1, h);
      free(up);
      ++ui;
      store_act(acts, ai++, b, h, (size_t)ch * S * S);
    }
  }
  float* a = fmalloc((size_t)ch * S * S);
  groupnorm(h, ch, S * S, ch < 8 ? ch : 8, P->onw, P->onb, 1, a);
  conv3x3(a, ch, S, S, P->ocw, P->ocb, d->in_channels, 1, out);
  free(a);
  free(h);
  store_act(acts, ai++, b, out, (size_t)d->in_channels * S * S);
}

void ro_unet_forward(const ro_unet_desc* d, const float* params, const float* x, const float* t,
                     int t_count, float* out, int B, float** acts) {
  unet_plan P;
  plan_unet(d, params, &P);
  const size_t n = (size_t)d->in_channels * d->img_size * d->img_size;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b)
    unet_one(d, &P, x + b * n, t[t_count == 1 ? 0 : b], out + b * n, acts, b);
}

/* ---------------------------------------------------------------- ratio estimators */

static void bn_silu(float* h, int C, int HW, const float* w, const float* b, const float* rm,
                    const float* rv) {
  /* nn.BatchNorm2d in eval mode (running stats, eps 1e-5) then F.silu */
  for (int c = 0; c < C; ++c) {
    const float inv = 1.0f / sqrtf(rv[c] + 1e-5f);
    for (int i = 0; i < HW; ++i) {
      const float v = (h[(size_t)c * HW + i] - rm[c]) * inv * w[c] + b[c];
      h[(size_t)c * HW + i] = silu(v);
    }
  }
}

/* F.max_pool2d(h, 2): floor mode */
static float* maxpool2(float* in, int C, int* S) {
  const int Si = *S, So = Si / 2;
  float* out = fmalloc((size_t)C * So * So);
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < So; ++y)
      for (int x = 0; x < So; ++x) {
        const float* p = in + ((size_t)c * Si + 2 * y) * Si + 2 * x;
        out[((size_t)c * So + y) * So + x] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[Si], p[Si + 1]));
      }
  free(in);
  *S = So;
  return out;
}

static void avgpool_fc(const float* h, int C, int HW, const float* fw, const float* fb, int F,
                       float* out) {
  float pooled[512];
  for (int c = 0; c < C; ++c) {
    float s = 0.0f;
    for (int i = 0; i < HW; ++i) s += h[(size_t)c * HW + i];
    pooled[c] = s / (float)HW;
  }
  linear(fw, fb, pooled, C, F, out);
}

/* MNISTEncoder / SVHNEncoder.forward (src/models/ratio_flexible.py:210-232, :271-302).
 * chans: nconv+1 channel counts; pool_after[i] = max-pool after conv i. */
static void bn_encoder(cursor* c, const float* img, int S, const int* chans, int nconv,
                       const int* pool_after, int F, float* feat) {
  float* h = fmalloc((size_t)chans[0] * S * S);
  memcpy(h, img, (size_t)chans[0] * S * S * sizeof(float));
  for (int i = 0; i < nconv; ++i) {
    const int ci = chans[i], co = chans[i + 1];
    const float* cw = take(c, (size_t)co * ci * 9);
    const float* cb = take(c, co);
    const float* bw = take(c, co);
    const float* bb = take(c, co);
    const float* rm = take(c, co);
    const float* rv = take(c, co);
    take(c, 1); /* num_batches_tracked */
    float* o = fmalloc((size_t)co * S * S);
    conv3x3(h, ci, S, S, cw, cb, co, 1, o);
    free(h);
    h = o;
    bn_silu(h, co, S * S, bw, bb, rm, rv);
    if (pool_after[i]) h = maxpool2(h, co, &S);
  }
  const float* fw = take(c, (size_t)F * chans[nconv]);
  const float* fb = take(c, F);
  avgpool_fc(h, chans[nconv], S * S, fw, fb, F, feat);
  free(h);
}

/* ImageEncoder.forward (src/models/ratio_estimator.py:67-93): conv-GN-SiLU(-maxpool) x4 */
static void gn_encoder(cursor* c, const float* img, int S, int F, float* feat) {
  static const int chans[5] = {1, 32, 64, 128, 128};
  float* h = fmalloc((size_t)S * S);
  memcpy(h, img, (size_t)S * S * sizeof(float));
  for (int i = 0; i < 4; ++i) {
    const int ci = chans[i], co = chans[i + 1];
    const float* cw = take(c, (size_t)co * ci * 9);
    const float* cb = take(c, co);
    const float* gw = take(c, co);
    const float* gb = take(c, co);
    float* o = fmalloc((size_t)co * S * S);
    conv3x3(h, ci, S, S, cw, cb, co, 1, o);
    free(h);
    h = fmalloc((size_t)co * S * S);
    groupnorm(o, co, S * S, 8, gw, gb, 1, h);
    free(o);
    if (i < 3) h = maxpool2(h, co, &S);
  }
  const float* fw = take(c, (size_t)F * 128);
  const float* fb = take(c, F);
  avgpool_fc(h, 128, S * S, fw, fb, F, feat);
  free(h);
}

/* nn.LayerNorm(n), eps 1e-5, then SiLU */
static void layernorm_silu(float* v, int n, const float* w, const float* b) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i];
  const double mean = s / n;
  double m2 = 0.0;
  for (int i = 0; i < n; ++i) m2 += (v[i] - mean) * (v[i] - mean);
  const float rstd = (float)(1.0 / sqrt(m2 / n + 1e-5)), mf = (float)mean;
  for (int i = 0; i < n; ++i) v[i] = silu((v[i] - mf) * rstd * w[i] + b[i]);
}

static inline float logsigmoidf(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

/* log_ratio (src/models/ratio_flexible.py:366-385, ratio_estimator.py:160-191) */
static float finish_ratio(float s, int loss, int what) {
  if (what == 0) return s;
  float lr;
  if (loss == 0) {
    lr = logsigmoidf(s) - logsigmoidf(-s);
  } else {
    const float w = s > 20.0f ? s : log1pf(expf(s)); /* F.softplus, threshold 20 */
    lr = logf(w + 1e-8f);
  }
  return what == 1 ? lr : expf(lr);
}

static size_t ratio_walk(int kind, int F, int Hd, const float* params, const float* x,
                         const float* y, int loss, int what, float* out, float* feat) {
  cursor c = {params};
  float fx[1024];
  float* f = fx;
  if (kind == RO_RATIO_MNIST_SVHN) {
    static const int cm[5] = {1, 32, 64, 128, 128}, pm[4] = {1, 1, 1, 0};
    static const int cs[9] = {3, 64, 64, 128, 128, 256, 256, 256, 256};
    static const int ps[8] = {0, 1, 0, 1, 0, 1, 0, 1};
    if (x) {
      bn_encoder(&c, x, 32, cm, 4, pm, F, f);
      bn_encoder(&c, y, 32, cs, 8, ps, F, f + F);
    } else {
      for (int i = 0; i < 4; ++i) take(&c, (size_t)cm[i + 1] * cm[i] * 9 + 5 * cm[i + 1] + 1);
      take(&c, (size_t)F * 128 + F);
      for (int i = 0; i < 8; ++i) take(&c, (size_t)cs[i + 1] * cs[i] * 9 + 5 * cs[i + 1] + 1);
      take(&c, (size_t)F * 256 + F);
    }
  } else {
    if (x) {
      gn_encoder(&c, x, 28, F, f);
      gn_encoder(&c, y, 28, F, f + F);
    } else {
      static const int ch[5] = {1, 32, 64, 128, 128};
      for (int e = 0; e < 2; ++e) {
        for (int i = 0; i < 4; ++i) take(&c, (size_t)ch[i + 1] * ch[i] * 9 + 3 * ch[i + 1]);
        take(&c, (size_t)F * 128 + F);
      }
    }
  }
  if (x && feat) memcpy(feat, f, 2 * F * sizeof(float));
  /* score_net (ratio_flexible.py:332-345: 3 hidden layers; ratio_estimator.py:125-135: 2) */
  int dims[5], nl;
  if (kind == RO_RATIO_MNIST_SVHN) {
    dims[0] = 2 * F, dims[1] = Hd, dims[2] = Hd, dims[3] = Hd / 2, nl = 3;
  } else {
    dims[0] = 2 * F, dims[1] = Hd, dims[2] = Hd / 2, nl = 2;
  }
  float bufa[2048], bufb[2048];
  float* cur = f;
  float* nxt = bufa;
  for (int l = 0; l < nl; ++l) {
    const float* w = take(&c, (size_t)dims[l + 1] * dims[l]);
    const float* b = take(&c, dims[l + 1]);
    const float* lw = take(&c, dims[l + 1]);
    const float* lb = take(&c, dims[l + 1]);
    if (x) {
      linear(w, b, cur, dims[l], dims[l + 1], nxt);
      layernorm_silu(nxt, dims[l + 1], lw, lb);
      cur = nxt;
      nxt = (nxt == bufa) ? bufb : bufa;
    }
  }
  const float* w = take(&c, dims[nl]);
  const float* b = take(&c, 1);
  if (x) {
    float s;
    linear(w, b, cur, dims[nl], 1, &s);
    *out = finish_ratio(s, loss, what);
  }
  return (size_t)(c.p - params);
}

size_t ro_ratio_param_floats(int kind, int feature_dim, int hidden_dim) {
  return ratio_walk(kind, feature_dim, hidden_dim, NULL, NULL, NULL, 0, 0, NULL, NULL);
}

void ro_ratio_eval(int kind, int feature_dim, int hidden_dim, int loss, const float* params,
                   const float* x, const float* y, float* out, int n, int what, float* feat) {
  const size_t nx = kind == RO_RATIO_MNIST_SVHN ? 1024 : 784;
  const size_t ny = kind == RO_RATIO_MNIST_SVHN ? 3072 : 784;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n; ++i)
    ratio_walk(kind, feature_dim, hidden_dim, params, x + i * nx, y + i * ny, loss, what, out + i,
               feat ? feat + (size_t)i * 2 * feature_dim : NULL);
}

/* ---------------------------------------------------------------- gradient of log r
 * d log_ratio(x, y) / d(x, y) of RatioEstimatorMNISTSVHN (src/models/ratio_flexible.py:347-385): what
 * torch.autograd.grad(model.log_ratio(x, y).sum(), (x, y)) returns for the reference module in eval mode
 * (BatchNorm on running statistics, Dropout off).  Hand-written reverse pass, one sample at a time:
 * forward with every pre-activation kept, then head -> score_net (Linear / LayerNorm / SiLU) -> the two
 * encoders (Linear, AdaptiveAvgPool, [SiLU, BatchNorm(eval), Conv3x3, MaxPool2] blocks). */

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float dsilu(float v) {
  const float s = sigmoidf_(v);
  return s * (1.0f + v * (1.0f - s));
}

/* gradient of conv3x3 (pad 1, stride 1) wrt its input: gin[ci,y,x] = sum_co,ky,kx w[co,ci,ky,kx] g[co,y-ky+1,x-kx+1] */
static void conv3x3_bwd_data(const float* g, int Co, int S, const float* w, int Ci, float* gin) {
  memset(gin, 0, (size_t)Ci * S * S * sizeof(float));
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci) {
      const float* wk = w + ((size_t)co * Ci + ci) * 9;
      const float* gp = g + (size_t)co * S * S;
      float* o = gin + (size_t)ci * S * S;
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
          const float wv = wk[ky * 3 + kx];
          /* output pixel (yo, xo) read input (yo + ky - 1, xo + kx - 1) */
          for (int yo = 0; yo < S; ++yo) {
            const int yi = yo + ky - 1;
            if (yi < 0 || yi >= S) continue;
            for (int xo = 0; xo < S; ++xo) {
              const int xi = xo + kx - 1;
              if (xi < 0 || xi >= S) continue;
              o[(size_t)yi * S + xi] += wv * gp[(size_t)yo * S + xo];
            }
          }
        }
    }
}

typedef struct {
  int ci, co, S, pooled;
  const float *cw, *bw, *bb, *rm, *rv; /* (GroupNorm encoder: bw / bb are the norm's weight / bias, rm / rv unused) */
  float* z; /* BatchNorm output (pre-SiLU) [co,S,S]; GroupNorm encoder: the conv output (the norm's input) */
  float mean[8], rstd[8]; /* GroupNorm encoder: per-group statistics of z */
} enc_layer;

/* forward of one BatchNorm encoder with the pre-activations kept; returns the number of layers */
static void bn_encoder_fwd_keep(cursor* c, const float* img, int S, const int* chans, int nconv, const int* pool_after,
                                int F, enc_layer* L, const float** fw_out, float* feat) {
  float* h = fmalloc((size_t)chans[0] * S * S);
  memcpy(h, img, (size_t)chans[0] * S * S * sizeof(float));
  for (int i = 0; i < nconv; ++i) {
    const int ci = chans[i], co = chans[i + 1];
    L[i].ci = ci, L[i].co = co, L[i].S = S, L[i].pooled = pool_after[i];
    L[i].cw = take(c, (size_t)co * ci * 9);
    const float* cb = take(c, co);
    L[i].bw = take(c, co), L[i].bb = take(c, co), L[i].rm = take(c, co), L[i].rv = take(c, co);
    take(c, 1);
    float* z = fmalloc((size_t)co * S * S);
    conv3x3(h, ci, S, S, L[i].cw, cb, co, 1, z);
    free(h);
    for (int ch = 0; ch < co; ++ch) {
      const float inv = 1.0f / sqrtf(L[i].rv[ch] + 1e-5f);
      for (int p = 0; p < S * S; ++p) z[(size_t)ch * S * S + p] = (z[(size_t)ch * S * S + p] - L[i].rm[ch]) * inv * L[i].bw[ch] + L[i].bb[ch];
    }
    L[i].z = z;
    h = fmalloc((size_t)co * S * S);
    for (size_t p = 0; p < (size_t)co * S * S; ++p) h[p] = silu(z[p]);
    if (pool_after[i]) h = maxpool2(h, co, &S);
  }
  const float* fw = take(c, (size_t)F * chans[nconv]);
  const float* fb = take(c, F);
  *fw_out = fw;
  avgpool_fc(h, chans[nconv], S * S, fw, fb, F, feat);
  free(h);
}

/* reverse pass of one encoder: gfeat [F] -> gimg [chans[0], 32, 32]; frees the kept pre-activations */
static void bn_encoder_bwd(enc_layer* L, int nconv, const float* fw, int F, const float* gfeat, float* gimg) {
  const int Cl = L[nconv - 1].co;
  int S = L[nconv - 1].pooled ? L[nconv - 1].S / 2 : L[nconv - 1].S; /* map size in front of the average pool */
  /* fc + AdaptiveAvgPool2d(1) */
  float* g = fmalloc((size_t)Cl * S * S);
  for (int ch = 0; ch < Cl; ++ch) {
    float a = 0.0f;
    for (int f = 0; f < F; ++f) a += fw[(size_t)f * Cl + ch] * gfeat[f];
    a /= (float)(S * S);
    for (int p = 0; p < S * S; ++p) g[(size_t)ch * S * S + p] = a;
  }
  for (int i = nconv - 1; i >= 0; --i) {
    const int co = L[i].co, ci = L[i].ci, Sz = L[i].S;
    float* gz = fmalloc((size_t)co * Sz * Sz);
    if (L[i].pooled) {
      /* F.max_pool2d(h, 2) backward: the gradient goes to the first maximum of the window (scan order ky, kx,
       * strict '>' as ATen's CPU kernel), then SiLU' */
      const int So = Sz / 2;
      memset(gz, 0, (size_t)co * Sz * Sz * sizeof(float));
      for (int ch = 0; ch < co; ++ch)
        for (int y = 0; y < So; ++y)
          for (int x = 0; x < So; ++x) {
            const float* zp = L[i].z + ((size_t)ch * Sz + 2 * y) * Sz + 2 * x;
            int best = 0;
            float bv = silu(zp[0]);
            const int offs[4] = {0, 1, Sz, Sz + 1};
            for (int k = 1; k < 4; ++k) {
              const float v = silu(zp[offs[k]]);
              if (v > bv) bv = v, best = k;
            }
            const size_t o = ((size_t)ch * Sz + 2 * y) * Sz + 2 * x + offs[best];
            gz[o] = g[((size_t)ch * So + y) * So + x] * dsilu(L[i].z[o]);
          }
    } else {
      for (size_t p = 0; p < (size_t)co * Sz * Sz; ++p) gz[p] = g[p] * dsilu(L[i].z[p]);
    }
    free(g);
    /* BatchNorm (eval): z = (u - rm) * inv * w + b -> du = dz * inv * w */
    for (int ch = 0; ch < co; ++ch) {
      const float sc = L[i].bw[ch] / sqrtf(L[i].rv[ch] + 1e-5f);
      for (int p = 0; p < Sz * Sz; ++p) gz[(size_t)ch * Sz * Sz + p] *= sc;
    }
    float* gin = i == 0 ? gimg : fmalloc((size_t)ci * Sz * Sz);
    conv3x3_bwd_data(gz, co, Sz, L[i].cw, ci, gin);
    free(gz);
    free(L[i].z);
    g = gin;
  }
}

/* ImageEncoder (src/models/ratio_estimator.py:34-93) with what the reverse pass needs kept: conv outputs and the
 * per-group statistics of their GroupNorm */
static void gn_encoder_fwd_keep(cursor* c, const float* img, int S, int F, enc_layer* L, const float** fw_out, float* feat) {
  static const int chans[5] = {1, 32, 64, 128, 128};
  float* h = fmalloc((size_t)S * S);
  memcpy(h, img, (size_t)S * S * sizeof(float));
  for (int i = 0; i < 4; ++i) {
    const int ci = chans[i], co = chans[i + 1], HW = S * S, cpg = co / 8;
    L[i].ci = ci, L[i].co = co, L[i].S = S, L[i].pooled = i < 3;
    L[i].cw = take(c, (size_t)co * ci * 9);
    const float* cb = take(c, co);
    L[i].bw = take(c, co), L[i].bb = take(c, co), L[i].rm = L[i].rv = NULL;
    float* z = fmalloc((size_t)co * HW);
    conv3x3(h, ci, S, S, L[i].cw, cb, co, 1, z);
    free(h);
    L[i].z = z;
    for (int g = 0; g < 8; ++g) { /* the statistics exactly as groupnorm() takes them */
      const float* p = z + (size_t)g * cpg * HW;
      const size_t n = (size_t)cpg * HW;
      double sm = 0.0, m2 = 0.0;
      for (size_t k = 0; k < n; ++k) sm += p[k];
      const double mean = sm / (double)n;
      for (size_t k = 0; k < n; ++k) m2 += (p[k] - mean) * (p[k] - mean);
      L[i].mean[g] = (float)mean, L[i].rstd[g] = (float)(1.0 / sqrt(m2 / (double)n + 1e-5));
    }
    h = fmalloc((size_t)co * HW);
    groupnorm(z, co, HW, 8, L[i].bw, L[i].bb, 1, h);
    if (i < 3) h = maxpool2(h, co, &S);
  }
  const float* fw = take(c, (size_t)F * 128);
  const float* fb = take(c, F);
  *fw_out = fw;
  avgpool_fc(h, 128, S * S, fw, fb, F, feat);
  free(h);
}

/* reverse pass of the GroupNorm encoder: gfeat [F] -> gimg [1, 28, 28]; frees the kept conv outputs.
 * y = silu(u), u = gamma xhat + beta, xhat = (z - mean_g) rstd_g over the group's cpg x S x S values:
 *   d z = rstd_g (d xhat - mean_g(d xhat) - xhat mean_g(d xhat xhat)),  d xhat = d u gamma   (nn.GroupNorm backward) */
static void gn_encoder_bwd(enc_layer* L, const float* fw, int F, const float* gfeat, float* gimg) {
  const int Cl = L[3].co;
  int S = L[3].S; /* no pool between conv4 and the average pool */
  float* g = fmalloc((size_t)Cl * S * S);
  for (int ch = 0; ch < Cl; ++ch) {
    float a = 0.0f;
    for (int f = 0; f < F; ++f) a += fw[(size_t)f * Cl + ch] * gfeat[f];
    a /= (float)(S * S);
    for (int p = 0; p < S * S; ++p) g[(size_t)ch * S * S + p] = a;
  }
  for (int i = 3; i >= 0; --i) {
    const int co = L[i].co, ci = L[i].ci, Sz = L[i].S, HW = Sz * Sz, cpg = co / 8;
    /* u = GroupNorm output (pre-SiLU), recomputed */
    float* u = fmalloc((size_t)co * HW);
    groupnorm(L[i].z, co, HW, 8, L[i].bw, L[i].bb, 0, u);
    float* gu = fmalloc((size_t)co * HW);
    if (L[i].pooled) { /* F.max_pool2d(., 2) (floor: the last row / column of an odd map gets no gradient), then SiLU' */
      const int So = Sz / 2;
      memset(gu, 0, (size_t)co * HW * sizeof(float));
      for (int ch = 0; ch < co; ++ch)
        for (int y = 0; y < So; ++y)
          for (int x = 0; x < So; ++x) {
            const float* up = u + ((size_t)ch * Sz + 2 * y) * Sz + 2 * x;
            int best = 0;
            float bv = silu(up[0]);
            const int offs[4] = {0, 1, Sz, Sz + 1};
            for (int k = 1; k < 4; ++k) {
              const float v = silu(up[offs[k]]);
              if (v > bv) bv = v, best = k;
            }
            const size_t o = ((size_t)ch * Sz + 2 * y) * Sz + 2 * x + offs[best];
            gu[o] = g[((size_t)ch * So + y) * So + x] * dsilu(u[o]);
          }
    } else {
      for (size_t p = 0; p < (size_t)co * HW; ++p) gu[p] = g[p] * dsilu(u[p]);
    }
    free(g);
    free(u);
    float* gz = fmalloc((size_t)co * HW);
    for (int gr = 0; gr < 8; ++gr) {
      const size_t n = (size_t)cpg * HW;
      double s1 = 0.0, s2 = 0.0;
      for (int c = 0; c < cpg; ++c) {
        const int ch = gr * cpg + c;
        for (int p = 0; p < HW; ++p) {
          const float xh = (L[i].z[(size_t)ch * HW + p] - L[i].mean[gr]) * L[i].rstd[gr];
          const float gx = gu[(size_t)ch * HW + p] * L[i].bw[ch];
          s1 += gx, s2 += (double)gx * xh;
        }
      }
      const float m1 = (float)(s1 / (double)n), m2 = (float)(s2 / (double)n);
      for (int c = 0; c < cpg; ++c) {
        const int ch = gr * cpg + c;
        for (int p = 0; p < HW; ++p) {
          const float xh = (L[i].z[(size_t)ch * HW + p] - L[i].mean[gr]) * L[i].rstd[gr];
          gz[(size_t)ch * HW + p] = L[i].rstd[gr] * (gu[(size_t)ch * HW + p] * L[i].bw[ch] - m1 - xh * m2);
        }
      }
    }
    free(gu);
    float* gin = i == 0 ? gimg : fmalloc((size_t)ci * HW);
    conv3x3_bwd_data(gz, co, Sz, L[i].cw, ci, gin);
    free(gz);
    free(L[i].z);
    g = gin;
  }
}

/* loss: 0 disc, 1 rulsif.  RO_RATIO_MNIST_SVHN: gx [1,32,32], gy [3,32,32]; RO_RATIO_MNIST28 (RatioEstimator,
 * src/models/ratio_estimator.py:96-191): gx, gy [1,28,28].  Returns log_ratio */
static float ratio_grad_one(int kind, int F, int Hd, const float* params, const float* x, const float* y, int loss, float* gx,
                            float* gy) {
  static const int cm[5] = {1, 32, 64, 128, 128}, pm[4] = {1, 1, 1, 0};
  static const int cs[9] = {3, 64, 64, 128, 128, 256, 256, 256, 256};
  static const int ps[8] = {0, 1, 0, 1, 0, 1, 0, 1};
  const int ms = kind == RO_RATIO_MNIST_SVHN;
  cursor c = {params};
  enc_layer Lm[4], Ls[8];
  const float *fwm, *fws;
  float feat[1024];
  if (ms) {
    bn_encoder_fwd_keep(&c, x, 32, cm, 4, pm, F, Lm, &fwm, feat);
    bn_encoder_fwd_keep(&c, y, 32, cs, 8, ps, F, Ls, &fws, feat + F);
  } else {
    gn_encoder_fwd_keep(&c, x, 28, F, Lm, &fwm, feat);
    gn_encoder_fwd_keep(&c, y, 28, F, Ls, &fws, feat + F);
  }
  /* score_net forward (ratio_flexible.py:332-345: 3 hidden layers; ratio_estimator.py:125-135: 2), inputs and
   * pre-LayerNorm values kept */
  int dims[4], nl;
  if (ms) dims[0] = 2 * F, dims[1] = Hd, dims[2] = Hd, dims[3] = Hd / 2, nl = 3;
  else dims[0] = 2 * F, dims[1] = Hd, dims[2] = Hd / 2, dims[3] = 0, nl = 2;
  const float *W[3], *lw[3], *lbs[3];
  float *in[4], *u[3];
  float mean[3], rstd[3];
  in[0] = fmalloc(2 * F);
  memcpy(in[0], feat, 2 * F * sizeof(float));
  for (int l = 0; l < nl; ++l) {
    W[l] = take(&c, (size_t)dims[l + 1] * dims[l]);
    const float* b = take(&c, dims[l + 1]);
    lw[l] = take(&c, dims[l + 1]);
    lbs[l] = take(&c, dims[l + 1]);
    const int n = dims[l + 1];
    u[l] = fmalloc(n);
    linear(W[l], b, in[l], dims[l], n, u[l]);
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += u[l][i];
    const double mu = s / n;
    double m2 = 0.0;
    for (int i = 0; i < n; ++i) m2 += (u[l][i] - mu) * (u[l][i] - mu);
    mean[l] = (float)mu, rstd[l] = (float)(1.0 / sqrt(m2 / n + 1e-5));
    in[l + 1] = fmalloc(n);
    for (int i = 0; i < n; ++i) in[l + 1][i] = silu((u[l][i] - mean[l]) * rstd[l] * lw[l][i] + lbs[l][i]);
  }
  const float* hw = take(&c, dims[nl]);
  const float* hb = take(&c, 1);
  float s;
  linear(hw, hb, in[nl], dims[nl], 1, &s);
  float lr, ds;
  if (loss == 0) {
    lr = logsigmoidf(s) - logsigmoidf(-s);
    ds = sigmoidf_(-s) + sigmoidf_(s); /* d logsigmoid(s) - d logsigmoid(-s) */
  } else {
    const float w = s > 20.0f ? s : log1pf(expf(s));
    lr = logf(w + 1e-8f);
    ds = (s > 20.0f ? 1.0f : sigmoidf_(s)) / (w + 1e-8f);
  }
  /* reverse: head */
  float* g = fmalloc(dims[nl]);
  for (int i = 0; i < dims[nl]; ++i) g[i] = ds * hw[i];
  for (int l = nl - 1; l >= 0; --l) {
    const int n = dims[l + 1];
    /* y = silu(v), v = gamma * uhat + beta, uhat = (u - mean) * rstd */
    float* guh = fmalloc(n);
    double m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < n; ++i) {
      const float uh = (u[l][i] - mean[l]) * rstd[l];
      const float v = uh * lw[l][i] + lbs[l][i];
      guh[i] = g[i] * dsilu(v) * lw[l][i];
      m1 += guh[i];
      m2 += (double)guh[i] * uh;
    }
    m1 /= n, m2 /= n;
    float* gu = fmalloc(n);
    for (int i = 0; i < n; ++i) {
      const float uh = (u[l][i] - mean[l]) * rstd[l];
      gu[i] = rstd[l] * (guh[i] - (float)m1 - uh * (float)m2);
    }
    free(guh);
    free(g);
    g = fmalloc(dims[l]);
    for (int k = 0; k < dims[l]; ++k) {
      float a = 0.0f;
      for (int o = 0; o < n; ++o) a += W[l][(size_t)o * dims[l] + k] * gu[o];
      g[k] = a;
    }
    free(gu);
  }
  if (ms) {
    bn_encoder_bwd(Lm, 4, fwm, F, g, gx);
    bn_encoder_bwd(Ls, 8, fws, F, g + F, gy);
  } else {
    gn_encoder_bwd(Lm, fwm, F, g, gx);
    gn_encoder_bwd(Ls, fws, F, g + F, gy);
  }
  free(g);
  for (int l = 0; l < nl; ++l) free(u[l]);
  for (int l = 0; l <= nl; ++l) free(in[l]);
  return lr;
}

void ro_ratio_grad(int kind, int feature_dim, int hidden_dim, int loss, const float* params, const float* x, const float* y,
                   float* gx, float* gy, float* log_ratio, int n) {
  const size_t nx = kind == RO_RATIO_MNIST_SVHN ? 1024 : 784;
  const size_t ny = kind == RO_RATIO_MNIST_SVHN ? 3072 : 784;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n; ++i) {
    const float lr = ratio_grad_one(kind, feature_dim, hidden_dim, params, x + (size_t)i * nx, y + (size_t)i * ny, loss,
                                    gx + (size_t)i * nx, gy + (size_t)i * ny);
    if (log_ratio) log_ratio[i] = lr;
  }
}

/* Gradient log-ratio guidance (reference README.md:159-164: v_guided = v_ind + gamma * grad log r(x_t, y_t)), explicit
 * Euler as the other samplers: x <- x + (v_x + gamma g_x) dt.  The reference ships no code for this mode: the
 * composition is this build's reading of the README line ("parity unpinned"); the gradient itself is pinned. */
void ro_sample_pair_grad(const ro_unet_desc* ddx, const float* px, const ro_unet_desc* ddy, const float* py, int kind,
                         int feature_dim, int hidden_dim, int loss, const float* pr, float* x, float* y, int B,
                         int num_steps, double gamma, int step_begin, int step_end) {
  const size_t dx = (size_t)ddx->in_channels * ddx->img_size * ddx->img_size;
  const size_t dy = (size_t)ddy->in_channels * ddy->img_size * ddy->img_size;
  float* vx = fmalloc(B * dx);
  float* vy = fmalloc(B * dy);
  float* gx = fmalloc(B * dx);
  float* gy = fmalloc(B * dy);
  const double dtd = 1.0 / (double)num_steps;
  const float dt = (float)dtd, gf = (float)gamma;
  for (int s = step_begin; s < step_end; ++s) {
    const float t = (float)((double)s * dtd);
    ro_unet_forward(ddx, px, x, &t, 1, vx, B, NULL);
    ro_unet_forward(ddy, py, y, &t, 1, vy, B, NULL);
    ro_ratio_grad(kind, feature_dim, hidden_dim, loss, pr, x, y, gx, gy, NULL, B);
    for (size_t i = 0; i < B * dx; ++i) x[i] = x[i] + (vx[i] + gf * gx[i]) * dt;
    for (size_t i = 0; i < B * dy; ++i) y[i] = y[i] + (vy[i] + gf * gy[i]) * dt;
  }
  free(vx), free(vy), free(gx), free(gy);
}

/* ---------------------------------------------------------------- guidance + Euler */

/* MC importance-weighted guidance block (src/sample_mnist_svhn.py:124-171,
 * src/utils/flow_utils.py:273-369).  Scalars follow the reference's Python-double
 * arithmetic, rounded to fp32 where a tensor op consumes them. */
NOFMA void ro_guidance_apply(const float* x, const float* y, float* vx, float* vy,
                             const float* mc_x1, const float* mc_y1, const float* mc_ratios, int B,
                             int N, int dx, int dy, double t, double gamma, float* weights) {
  const double eps = 1e-3;
  const double sigma_t = 1.0 - t + eps;
  const float s2 = (float)(sigma_t * sigma_t);
  const float tf = (float)t;
  const float cden = (float)(1.0 - t + eps);
  const float g1 = (float)(1.0 - gamma), g2 = (float)gamma;
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    float* lp = fmalloc(N);
    float* w = fmalloc(N);
    const float* xb = x + (size_t)b * dx;
    const float* yb = y + (size_t)b * dy;
    for (int i = 0; i < N; ++i) {
      double sx = 0.0, sy = 0.0;
      const float* m = mc_x1 + (size_t)i * dx;
      for (int k = 0; k < dx; ++k) {
        const float mu = tf * m[k];
        const float df = xb[k] - mu;
        sx += (double)(df * df);
      }
      m = mc_y1 + (size_t)i * dy;
      for (int k = 0; k < dy; ++k) {
        const float mu = tf * m[k];
        const float df = yb[k] - mu;
        sy += (double)(df * df);
      }
      const float lx = (-0.5f * (float)sx) / s2;
      const float ly = (-0.5f * (float)sy) / s2;
      lp[i] = lx + ly;
    }
    float mx = lp[0];
    for (int i = 1; i < N; ++i) mx = fmaxf(mx, lp[i]);
    float ps = 0.0f, zs = 0.0f;
    for (int i = 0; i < N; ++i) {
      w[i] = expf(lp[i] - mx); /* p_joint */
      ps += w[i];
      zs += mc_ratios[i] * w[i];
    }
    const float pbar = ps / (float)N + 1e-10f;
    const float zbar = zs / (float)N + 1e-10f;
    float ws = 0.0f;
    for (int i = 0; i < N; ++i) {
      w[i] = (mc_ratios[i] / zbar) * (w[i] / pbar);
      ws += w[i];
    }
    ws += 1e-10f;
    for (int i = 0; i < N; ++i) w[i] = w[i] / ws;
    if (weights) memcpy(weights + (size_t)b * N, w, N * sizeof(float));
    for (int k = 0; k < dx; ++k) {
      float g = 0.0f;
      for (int i = 0; i < N; ++i) g += w[i] * ((mc_x1[(size_t)i * dx + k] - xb[k]) / cden);
      vx[(size_t)b * dx + k] = g1 * vx[(size_t)b * dx + k] + g2 * g;
    }
    for (int k = 0; k < dy; ++k) {
      float g = 0.0f;
      for (int i = 0; i < N; ++i) g += w[i] * ((mc_y1[(size_t)i * dy + k] - yb[k]) / cden);
      vy[(size_t)b * dy + k] = g1 * vy[(size_t)b * dy + k] + g2 * g;
    }
    free(lp);
    free(w);
  }
}

/* x <- x + v * dt: mul then add, two roundings (sample_mnist_svhn.py:174-175) */
NOFMA static void euler(float* x, const float* v, size_t n, float dt) {
  for (size_t i = 0; i < n; ++i) {
    const float s = v[i] * dt;
    x[i] = x[i] + s;
  }
}

/* CFMSchedule.sample loop (src/utils/flow_utils.py:87-98) / MC pre-phase loops
 * (src/sample_mnist_svhn.py:90-95, :99-104). */
void ro_sample_single(const ro_unet_desc* d, const float* params, float* x, int B, int num_steps,
                      int step_begin, int step_end) {
  const size_t n = (size_t)B * d->in_channels * d->img_size * d->img_size;
  const double dt = 1.0 / num_steps;
  float* v = fmalloc(n);
  for (int s = step_begin; s < step_end; ++s) {
    const float t = (float)(s * dt);
    ro_unet_forward(d, params, x, &t, 1, v, B, NULL);
    euler(x, v, n, (float)dt);
  }
  free(v);
}

/* main loop of sample_bimodal_guided_mnist_svhn (src/sample_mnist_svhn.py:114-175) and
 * sample_bimodal_guided (src/utils/flow_utils.py:263-373). */
void ro_sample_pair(const ro_unet_desc* ddx, const float* px, const ro_unet_desc* ddy,
                    const float* py, float* x, float* y, const float* mc_x1, const float* mc_y1,
                    const float* mc_ratios, int n_mc, int B, int num_steps, double gamma,
                    int step_begin, int step_end) {
  const int dx = ddx->in_channels * ddx->img_size * ddx->img_size;
  const int dy = ddy->in_channels * ddy->img_size * ddy->img_size;
  const double dt = 1.0 / num_steps, eps = 1e-3;
  float* vx = fmalloc((size_t)B * dx);
  float* vy = fmalloc((size_t)B * dy);
  for (int s = step_begin; s < step_end; ++s) {
    const double t = s * dt;
    const float tf = (float)t;
    ro_unet_forward(ddx, px, x, &tf, 1, vx, B, NULL);
    ro_unet_forward(ddy, py, y, &tf, 1, vy, B, NULL);
    if (n_mc > 0 && t > eps)
      ro_guidance_apply(x, y, vx, vy, mc_x1, mc_y1, mc_ratios, B, n_mc, dx, dy, t, gamma, NULL);
    euler(x, vx, (size_t)B * dx, (float)dt);
    euler(y, vy, (size_t)B * dy, (float)dt);
  }
  free(vx);
  free(vy);
}

/* ---------------------------------------------------------------- FlowMatchingModel ("original" net) */

/* nn.ConvTranspose2d(k=4, stride=2, padding=1): in [Ci,H,W] -> out [Co,2H,2W]; w [Ci][Co][4][4] */
static void deconv4x4s2(const float* in, int Ci, int H, int W, const float* w, const float* b, int Co,
                        float* out) {
  const int Ho = 2 * H, Wo = 2 * W;
  for (int co = 0; co < Co; ++co)
    for (int i = 0; i < Ho * Wo; ++i) out[(size_t)co * Ho * Wo + i] = b[co];
  for (int ci = 0; ci < Ci; ++ci)
    for (int co = 0; co < Co; ++co) {
      const float* wk = w + ((size_t)ci * Co + co) * 16;
      float* o = out + (size_t)co * Ho * Wo;
      for (int iy = 0; iy < H; ++iy)
        for (int ix = 0; ix < W; ++ix) {
          const float v = in[((size_t)ci * H + iy) * W + ix];
          for (int ky = 0; ky < 4; ++ky) {
            const int oy = 2 * iy - 1 + ky;
            if (oy < 0 || oy >= Ho) continue;
            for (int kx = 0; kx < 4; ++kx) {
              const int ox = 2 * ix - 1 + kx;
              if (ox < 0 || ox >= Wo) continue;
              o[(size_t)oy * Wo + ox] += v * wk[ky * 4 + kx];
            }
          }
        }
    }
}

size_t ro_fm_param_floats(void) {
  /* encoder: conv1..4 + gn1..4 + fc ; decoder: fc1, deconv1, gn1, deconv2, gn2, conv3, gn3, conv_out */
  size_t n = 0;
  const int ec[5] = {1, 32, 64, 128, 256};
  for (int i = 0; i < 4; ++i) n += (size_t)ec[i + 1] * ec[i] * 9 + 3 * ec[i + 1];
  n += (size_t)256 * 12544 + 256;
  n += (size_t)12544 * 384 + 12544;
  n += (size_t)256 * 128 * 16 + 128 + 2 * 128;
  n += (size_t)128 * 64 * 16 + 64 + 2 * 64;
  n += (size_t)32 * 64 * 9 + 32 + 2 * 32;
  n += (size_t)32 * 9 + 1;
  return n;
}

/* SinusoidalPositionEmbeddings (src/models/flow_matching.py:17-31): sin half first, denominator half-1 */
void ro_fm_time_embedding(const float* t, int n, int dim, float* out) {
  const int half = dim / 2;
  const float neg = (float)(-(log(10000.0) / (double)(half - 1)));
  for (int b = 0; b < n; ++b)
    for (int i = 0; i < half; ++i) {
      const float a = t[b] * expf((float)i * neg);
      out[(size_t)b * dim + i] = sinf(a);
      out[(size_t)b * dim + half + i] = cosf(a);
    }
}

/* FlowMatchingModel.forward (src/models/flow_matching.py:153-173; encoder :56-72, decoder :100-124) */
void ro_fm_forward(const float* params, const float* x, const float* t, int t_count, float* out, int B) {
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    cursor c = {params};
    const int ec[5] = {1, 32, 64, 128, 256}, es[4] = {1, 2, 2, 1};
    int S = 28;
    float* h = fmalloc(784);
    memcpy(h, x + (size_t)b * 784, 784 * sizeof(float));
    for (int i = 0; i < 4; ++i) {
      const float* cw = take(&c, (size_t)ec[i + 1] * ec[i] * 9);
      const float* cb = take(&c, ec[i + 1]);
      const float* gw = take(&c, ec[i + 1]);
      const float* gb = take(&c, ec[i + 1]);
      const int So = (S + 2 - 3) / es[i] + 1;
      float* o = fmalloc((size_t)ec[i + 1] * So * So);
      conv3x3(h, ec[i], S, S, cw, cb, ec[i + 1], es[i], o);
      free(h);
      h = fmalloc((size_t)ec[i + 1] * So * So);
      groupnorm(o, ec[i + 1], So * So, 8, gw, gb, 1, h);
      free(o);
      S = So;
    }
    const float* fw = take(&c, (size_t)256 * 12544);
    const float* fb = take(&c, 256);
    float comb[384];
    linear(fw, fb, h, 12544, 256, comb);
    free(h);
    const float tb = t[t_count == 1 ? 0 : b];
    ro_fm_time_embedding(&tb, 1, 128, comb + 256);
    const float* f1w = take(&c, (size_t)12544 * 384);
    const float* f1b = take(&c, 12544);
    float* d0 = fmalloc(12544);
    linear(f1w, f1b, comb, 384, 12544, d0);
    const float* d1w = take(&c, (size_t)256 * 128 * 16);
    const float* d1b = take(&c, 128);
    const float* g1w = take(&c, 128);
    const float* g1b = take(&c, 128);
    float* u1 = fmalloc((size_t)128 * 14 * 14);
    deconv4x4s2(d0, 256, 7, 7, d1w, d1b, 128, u1);
    free(d0);
    float* a1 = fmalloc((size_t)128 * 196);
    groupnorm(u1, 128, 196, 8, g1w, g1b, 1, a1);
    free(u1);
    const float* d2w = take(&c, (size_t)128 * 64 * 16);
    const float* d2b = take(&c, 64);
    const float* g2w = take(&c, 64);
    const float* g2b = take(&c, 64);
    float* u2 = fmalloc((size_t)64 * 784);
    deconv4x4s2(a1, 128, 14, 14, d2w, d2b, 64, u2);
    free(a1);
    float* a2 = fmalloc((size_t)64 * 784);
    groupnorm(u2, 64, 784, 8, g2w, g2b, 1, a2);
    free(u2);
    const float* c3w = take(&c, (size_t)32 * 64 * 9);
    const float* c3b = take(&c, 32);
    const float* g3w = take(&c, 32);
    const float* g3b = take(&c, 32);
    float* u3 = fmalloc((size_t)32 * 784);
    conv3x3(a2, 64, 28, 28, c3w, c3b, 32, 1, u3);
    free(a2);
    float* a3 = fmalloc((size_t)32 * 784);
    groupnorm(u3, 32, 784, 8, g3w, g3b, 1, a3);
    free(u3);
    const float* cow = take(&c, 32 * 9);
    const float* cob = take(&c, 1);
    conv3x3(a3, 32, 28, 28, cow, cob, 1, 1, out + (size_t)b * 784);
    free(a3);
  }
}
