// rgfm_api.cpp -- host side of the C ABI declared in include/rgfm.h: parameter
// ingestion/packing, network walks that enqueue the gfx950 kernels, the Euler
// sampler loops, and the hipEvent-based kernel-class timers used by bench.py.
// No PyTorch types, no allocation and no synchronisation inside forward/sample
// calls (everything is carved from the caller's workspace, stream-ordered).
#include "../../include/rgfm.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rgfm_kernels.h"

using namespace rgfm;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return fail(RGFM_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* rgfm_last_error(void) { return g_err.c_str(); }
extern "C" int rgfm_abi_version(void) { return RGFM_ABI_VERSION; }

// ------------------------------------------------------------------ profiling (bench support)
namespace {
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int> cls;
  size_t used = 0;
  double flops[RGFM_KCLASS_COUNT] = {};  // algorithmic FLOPs (conv class) or algorithmic HBM bytes (the others)
  double sum_ms[RGFM_KCLASS_COUNT] = {};
  int64_t launches[RGFM_KCLASS_COUNT] = {};
  std::vector<std::pair<double, double>> iv[RGFM_KCLASS_COUNT];  // [start, stop] ms since the first event
  hipEvent_t base = nullptr;
  bool have_base = false;
} g_prof;

struct ProfScope {
  bool active = false;
  size_t idx = 0;
  hipStream_t s;
  ProfScope(int kclass, double flops, hipStream_t stream) : s(stream) {
    if (!g_prof.on) return;
    if (g_prof.used + 2 > g_prof.ev.size()) {
      for (int i = 0; i < 4096; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_prof.ev.push_back(e);
      }
      g_prof.cls.resize(g_prof.ev.size() / 2);
    }
    if (!g_prof.have_base) {
      if (!g_prof.base && hipEventCreate(&g_prof.base) != hipSuccess) return;
      (void)hipEventRecord(g_prof.base, s);
      g_prof.have_base = true;
    }
    idx = g_prof.used;
    g_prof.used += 2;
    g_prof.cls[idx / 2] = kclass;
    g_prof.flops[kclass] += flops;
    g_prof.launches[kclass] += 1;
    (void)hipEventRecord(g_prof.ev[idx], s);
    active = true;
  }
  ~ProfScope() {
    if (active) (void)hipEventRecord(g_prof.ev[idx + 1], s);
  }
};

int prof_collect() {
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    HIP_TRY(hipEventSynchronize(g_prof.ev[i + 1]));
    float t0 = 0.f, t1 = 0.f;
    HIP_TRY(hipEventElapsedTime(&t0, g_prof.base, g_prof.ev[i]));
    HIP_TRY(hipEventElapsedTime(&t1, g_prof.base, g_prof.ev[i + 1]));
    const int k = g_prof.cls[i / 2];
    g_prof.sum_ms[k] += (double)t1 - (double)t0;
    g_prof.iv[k].push_back({(double)t0, (double)t1});
  }
  g_prof.used = 0;
  return RGFM_OK;
}

double prof_union_ms(int k) {
  auto v = g_prof.iv[k];
  std::sort(v.begin(), v.end());
  double total = 0.0, lo = 0.0, hi = -1.0;
  for (const auto& p : v) {
    if (p.first > hi) {
      if (hi >= lo) total += hi - lo;
      lo = p.first, hi = p.second;
    } else if (p.second > hi) {
      hi = p.second;
    }
  }
  if (hi >= lo) total += hi - lo;
  return total;
}
}  // namespace

extern "C" int rgfm_profile_enable(int enable) {
  g_prof.on = enable != 0;
  return RGFM_OK;
}
extern "C" int rgfm_profile_reset(void) {
  int rc = prof_collect();
  for (int k = 0; k < RGFM_KCLASS_COUNT; ++k) {
    g_prof.flops[k] = g_prof.sum_ms[k] = 0, g_prof.launches[k] = 0;
    g_prof.iv[k].clear();
  }
  g_prof.have_base = false;
  return rc;
}
extern "C" int rgfm_profile_read(int kclass, double* busy_ms, double* sum_ms, int64_t* launches, double* flops) {
  if (kclass < 0 || kclass >= RGFM_KCLASS_COUNT) return fail(RGFM_EINVAL, "bad kernel class %d", kclass);
  int rc = prof_collect();
  if (rc) return rc;
  if (busy_ms) *busy_ms = prof_union_ms(kclass);
  if (sum_ms) *sum_ms = g_prof.sum_ms[kclass];
  if (launches) *launches = g_prof.launches[kclass];
  if (flops) *flops = g_prof.flops[kclass];
  return RGFM_OK;
}

extern "C" int rgfm_profile_reserve(int64_t launches) {
  if (launches < 0) return fail(RGFM_EINVAL, "negative launch count");
  const size_t want = (size_t)launches * 2;
  while (g_prof.ev.size() < want) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    g_prof.ev.push_back(e);
  }
  g_prof.cls.resize(g_prof.ev.size() / 2);
  if (!g_prof.base) HIP_TRY(hipEventCreate(&g_prof.base));
  return RGFM_OK;
}

// ------------------------------------------------------------------ small helpers
namespace {

struct Bump {  // workspace carving; dry = size-only pass
  char* base = nullptr;
  size_t off = 0;
  size_t cap = 0;
  bool dry = true;
  bool overflow = false;
  float* f(size_t nfloats) {
    const size_t bytes = (nfloats * sizeof(float) + 255) & ~(size_t)255;
    const size_t o = off;
    off += bytes;
    if (dry) return nullptr;
    if (off > cap) {
      overflow = true;
      return reinterpret_cast<float*>(base);  // never dereferenced: caller checks overflow first
    }
    return reinterpret_cast<float*>(base + o);
  }
};

struct Tensor {  // NHWC activation + its GroupNorm partial statistics
  float* data = nullptr;
  float* stats = nullptr;
  int C = 0, S = 0;
};

struct Cursor {
  size_t off = 0;
  size_t take(size_t n) {
    const size_t r = off;
    off += n;
    return r;
  }
};

struct ConvW {  // one packed conv
  size_t w_raw = 0, b = 0;  // offsets into the params blob
  size_t w_pk = 0;          // offset into the packed buffer
  size_t w_bx3 = 0;         // offset (bf16 elements) into the 3-plane bf16 buffer of conv_mfma_bx3.hip
  size_t w_hx2 = 0;         // offset (fp16 elements) into the 2-plane fp16 buffer of conv_mfma_hx2.hip
  size_t w_hx9 = 0;         // stride-2 convs: offset of the plain nine-tap fp16 image (conv_mfma_hx2s.hip), + 1 (0: none)
  int hq = 0;               // index of the conv's scale record {q, 1/q, s_w, eligible} in the handle's hq array
  bool hx_ok = false;       // weights inside the fp16 path's range (set after packing)
  int cin = 0, cout = 0, taps = 9;
};

struct ResW {
  int cin = 0, cout = 0;
  size_t n1w, n1b, n2w, n2b, tw, tb;
  ConvW c1, c2, sk;
  bool has_skip = false;
  int temb_off = 0;
};

int nt32_of(int cout) { return (cout % 64 == 0) ? 2 : 1; }

bool on_gfx950() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

// Run-time switches, read from the environment ONCE per API call (refresh_modes), never on the launch path:
//   RGFM_CONV = hx2 (default: conv_mfma_hx2.hip, fp32 operands as two scaled fp16 planes, three f16-MFMA products
//               per fp32 product) | bx3 (conv_mfma_bx3.hip: three exact bf16 planes, six products; fp32 range)
//               | f32 (conv_mfma.hip: v_mfma_f32_32x32x2_f32 everywhere);
//   RGFM_OVERLAP=0   both velocity nets of a step on the caller's stream;
//   RGFM_FUSE_FIN=0  separate gn_finalize launches instead of the producer-side finalize;
//   RGFM_GN=table    every GroupNorm finalized into a scale/shift array instead of the consumer-side prologue;
//   RGFM_HX2P=0, RGFM_HX2Q=0, RGFM_HX2S=0, RGFM_HX2C=0, RGFM_GRAPH=1: A/B switches of the pipelined fp16 kernel, its
//               four-waves-per-SIMD version, the stride-2 kernel, the 8x8-level kernel and the hipGraph replay.
enum { CONV_ARITH_HX2 = 0, CONV_ARITH_BX3 = 1, CONV_ARITH_F32 = 2 };
struct Modes {
  int conv = CONV_ARITH_HX2;
  bool overlap = true, fuse_fin = true, gn_consumer = true;
  bool pipelined = true;  // RGFM_HX2P=0: the fp16 convs on conv_mfma_hx2_kernel only (A/B switch)
  bool quad = true;       // RGFM_HX2Q=0: no four-waves-per-SIMD workgroups (conv_mfma_hx2q.hip; A/B switch, bit-identical)
  bool c8 = true;         // RGFM_HX2C=0: the 8x8 level on conv_mfma_hx2p_kernel (A/B switch)
  bool s2 = true;         // RGFM_HX2S=0: the Downsample convs on conv_mfma_hx2_kernel<*, CONV_S2, *> (A/B switch; same to 1e-6)
  bool graph = false;     // RGFM_GRAPH=1: the guided steps of the paired U-Net loop replayed from one captured hipGraph
                          // (bit-identical; measured 0.995-1.002x of the kernel-by-kernel path: the host is not the bottleneck)
};
thread_local Modes g_modes;  // per host thread: a handle's own conv arithmetic (rgfm_*_set_conv_mode) overrides it per network walk
void refresh_modes() {
  Modes m;
  const char* e = getenv("RGFM_CONV");
  if (e && strcmp(e, "bx3") == 0) m.conv = CONV_ARITH_BX3;
  else if (e && strcmp(e, "f32") == 0) m.conv = CONV_ARITH_F32;
  e = getenv("RGFM_OVERLAP");
  m.overlap = !(e && e[0] == '0');
  e = getenv("RGFM_FUSE_FIN");
  m.fuse_fin = !(e && e[0] == '0');
  e = getenv("RGFM_GN");
  m.gn_consumer = !(e && strcmp(e, "table") == 0);
  e = getenv("RGFM_HX2P");
  m.pipelined = !(e && e[0] == '0');
  e = getenv("RGFM_HX2Q");
  m.quad = !(e && e[0] == '0');
  e = getenv("RGFM_HX2S");
  m.s2 = !(e && e[0] == '0');
  e = getenv("RGFM_HX2C");
  m.c8 = !(e && e[0] == '0');
  e = getenv("RGFM_GRAPH");
  m.graph = e && e[0] == '1';
  g_modes = m;
}

// A handle's own conv arithmetic (rgfm_unet_set_conv_mode / rgfm_fmnet_set_conv_mode; -1: the environment's) for the
// duration of one network walk.
struct ModeScope {
  int saved;
  explicit ModeScope(int handle_mode) : saved(g_modes.conv) {
    if (handle_mode >= 0) g_modes.conv = handle_mode;
  }
  ~ModeScope() { g_modes.conv = saved; }
};

// Per-device state, created by the first rgfm_*_create on that device (never inside forward / sample calls):
// raised dynamic-LDS limits (a per-device function attribute), the side stream + fork/join events of the paired
// sampler, and the range-flag word of conv_mfma_hx2.hip.
constexpr int MAX_DEVICES = 16;
struct DevState {
  bool init = false;
  int num_cus = 256;
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  // the legacy default stream cannot be captured: a caller on it has its graph-replayed loop run on `main`,
  // forked from / joined back into the default stream with these events
  hipStream_t main = nullptr;
  hipEvent_t main_fork = nullptr, main_join = nullptr;
  // hipGraphs of earlier sampler calls that may still be executing: destroyed once `graph_done` (recorded behind the
  // latest replay) has completed
  hipEvent_t graph_done = nullptr;
  std::vector<std::pair<hipGraphExec_t, hipGraph_t>> graphs;
};
DevState g_dev[MAX_DEVICES];

DevState* cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  return g_dev[dev].init ? &g_dev[dev] : nullptr;
}

int ensure_init() {
  if (!on_gfx950()) return fail(RGFM_ENODEVICE, "librgfm_hip needs a gfx950 (MI355X) device; none is current");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(RGFM_EINVAL, "device ordinal %d out of range (max %d)", dev, MAX_DEVICES - 1);
  DevState& d = g_dev[dev];
  if (!d.init) {
    if (conv_mfma_init() != 0 || conv_bx3_init() != 0 || conv_hx2_init() != 0 || conv_hx2p_init() != 0 || conv_hx2q_init() != 0 ||
        conv_hx2s_init() != 0 || conv_hx2c_init() != 0 || guid_apply_init() != 0)
      return fail(RGFM_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) d.num_cus = p.multiProcessorCount;
    conv_hx2p_set_half(d.num_cus);  // launches with fewer 128-channel workgroups than CUs take 64-channel workgroups
    HIP_TRY(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d.fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.graph_done, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&d.main, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&d.main_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.main_join, hipEventDisableTiming));
    d.init = true;
  }
  return RGFM_OK;
}

void launch_conv(const ConvArgs& c, int mode, hipStream_t s) {
  if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.s2 && conv_hx2s_supported(c, mode)) launch_conv_hx2s(c, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.c8 && conv_hx2c_supported(c, mode)) launch_conv_hx2c(c, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.quad && conv_hx2q_supported(c, mode)) launch_conv_hx2q(c, mode, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && conv_hx2p_supported(c, mode)) launch_conv_hx2p(c, mode, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && conv_hx2_supported(c, mode)) launch_conv_hx2(c, mode, s);
  else if (g_modes.conv != CONV_ARITH_F32 && conv_bx3_supported(c, mode)) launch_conv_bx3(c, mode, s);
  else launch_conv_mfma(c, mode, s);
}

}  // namespace

// ------------------------------------------------------------------ fp16-path range flag
// Every U-Net / FlowMatchingModel handle owns one device word.  conv_mfma_hx2*.hip OR into it: bit 0 when a staged
// activation reaches |S_A a| >= 32768 (fp16 would overflow), bit 1 when an output that a later conv stages raw is too
// small for the two-plane representation (ConvArgs::small_check).  Either way the results of the handle's calls since
// the last reset are not fp32-class and the caller repeats them with the handle switched to RGFM_CONV_BX3 (bit 0: fp32's
// exponent range at the top) or to RGFM_CONV_F32 (bit 1: the exact fp32 matrix-core convs, the reference's arithmetic
// at any magnitude) -- what _engine._range_guarded and INTEGRATION.md section B do.  Per handle, so that two threads /
// two engines on one device cannot consume each other's flag.
static int read_flag_word(unsigned* word, int* flagged, int reset, hipStream_t s) {
  if (!flagged) return fail(RGFM_EINVAL, "null output");
  *flagged = 0;
  if (!word) return RGFM_OK;
  unsigned v = 0;
  HIP_TRY(hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *flagged = (int)v;
  if (reset && v) HIP_TRY(hipMemsetAsync(word, 0, sizeof(v), s));
  return RGFM_OK;
}
static int alloc_flag_word(unsigned** word) {
  HIP_TRY(hipMalloc(word, 256));
  HIP_TRY(hipMemset(*word, 0, 256));
  return RGFM_OK;
}
static int check_conv_mode(int mode) {
  if (mode < -1 || mode > CONV_ARITH_F32) return fail(RGFM_EINVAL, "conv mode must be RGFM_CONV_DEFAULT, _HX2, _BX3 or _F32");
  return RGFM_OK;
}

// ================================================================== U-Net
struct rgfm_unet {
  rgfm_unet_desc d;
  float* params = nullptr;  // device copy of the state_dict-order blob
  float* packed = nullptr;  // packed conv weights
  unsigned short* packedh = nullptr;  // 2-plane scaled fp16 weights (conv_mfma_hx2.hip)
  size_t n_packedh = 0;
  float* hq = nullptr;      // [n_hq][4] scale records of packedh
  int n_hq = 0;
  unsigned* range_flag = nullptr;  // this handle's range-flag word
  int conv_mode = -1;              // rgfm_unet_set_conv_mode: -1 = RGFM_CONV from the environment
  unsigned short* packed3 = nullptr;  // 3-plane bf16 weights (conv_mfma_bx3.hip)
  size_t n_packed3 = 0;
  float* freqs = nullptr;
  TimeLinear* lin_dev = nullptr;
  size_t n_params = 0, n_packed = 0;
  int mc = 0, temb = 0, nlin = 0, temb_total = 0;
  size_t te0w, te0b, te2w, te2b, icw, icb, onw, onb, ocw, ocb;
  size_t ocw_pk = 0;  // out_conv weights re-laid out for conv_out_kernel (offset into `packed`)
  std::vector<ResW> enc, mid, dec;
  std::vector<ConvW> down, up;
  int final_ch = 0;
  bool trace = false;
  struct Act {
    float* data;
    int C, S;
    bool nchw;
  };
  std::vector<Act> acts;  // filled by the last (non-dry) run in trace mode
};

namespace {

// Walks the parameter registration order of FlexibleUNet.__init__
// (reference src/models/unet_flexible.py:146-201; UNetMNIST, src/models/unet.py:155-214,
// is identical) and records blob offsets.  Returns the total float count.
size_t plan_unet(const rgfm_unet_desc& d, rgfm_unet* h) {
  Cursor c;
  Cursor pk, p3, ph;
  int nhq = 0;
  const int mc = d.model_channels, temb = 4 * mc;
  std::vector<ResW> enc, mid, dec;
  std::vector<ConvW> down, up;
  int temb_off = 0;
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
  auto res = [&](int cin, int cout) {
    ResW r;
    r.cin = cin, r.cout = cout;
    r.n1w = c.take(cin), r.n1b = c.take(cin);
    r.c1 = conv(cin, cout, 9);
    r.tw = c.take((size_t)cout * temb), r.tb = c.take(cout);
    r.n2w = c.take(cout), r.n2b = c.take(cout);
    r.c2 = conv(cout, cout, 9);
    r.has_skip = cin != cout;
    if (r.has_skip) r.sk = conv(cin, cout, 1);
    r.temb_off = temb_off;
    temb_off += cout;
    return r;
  };
  const size_t te0w = c.take((size_t)temb * mc), te0b = c.take(temb);
  const size_t te2w = c.take((size_t)temb * temb), te2b = c.take(temb);
  const size_t icw = c.take((size_t)mc * d.in_channels * 9), icb = c.take(mc);
  int ch = mc;
  std::vector<int> skips{ch}, down_ch, up_ch;
  for (int l = 0; l < d.num_levels; ++l) {
    const int oc = mc * d.channel_mult[l];
    for (int r = 0; r < d.num_res_blocks; ++r) {
      enc.push_back(res(ch, oc));
      ch = oc;
      skips.push_back(ch);
    }
    if (l < d.num_levels - 1) {
      down_ch.push_back(ch);
      skips.push_back(ch);
    }
  }
  for (int dc : down_ch) {
    ConvW w = conv(dc, dc, 9);
    w.w_hx9 = ph.take((size_t)dc * dc * 9 * 2) + 1;  // (the Downsample convs twice: phase-major and plain)
    down.push_back(w);
  }
  mid.push_back(res(ch, ch));
  mid.push_back(res(ch, ch));
  for (int l = d.num_levels - 1; l >= 0; --l) {
    const int oc = mc * d.channel_mult[l];
    for (int i = 0; i < d.num_res_blocks + 1; ++i) {
      dec.push_back(res(ch + skips.back(), oc));
      skips.pop_back();
      ch = oc;
    }
    if (l > 0) up_ch.push_back(ch);
  }
  for (int uc : up_ch) up.push_back(conv(uc, uc, 9));
  const size_t onw = c.take(ch), onb = c.take(ch);
  const size_t ocw = c.take((size_t)d.in_channels * ch * 9), ocb = c.take(d.in_channels);
  const size_t ocw_pk = pk.take((size_t)d.in_channels * ch * 9);
  if (h) {
    h->mc = mc, h->temb = temb;
    h->te0w = te0w, h->te0b = te0b, h->te2w = te2w, h->te2b = te2b;
    h->icw = icw, h->icb = icb, h->onw = onw, h->onb = onb, h->ocw = ocw, h->ocb = ocb, h->ocw_pk = ocw_pk;
    h->enc = enc, h->mid = mid, h->dec = dec, h->down = down, h->up = up;
    h->final_ch = ch;
    h->temb_total = temb_off;
    h->n_packed = pk.off;
    h->n_packedh = ph.off, h->n_hq = nhq;
    h->n_packed3 = p3.off;
  }
  return c.off;
}

int check_desc(const rgfm_unet_desc* d) {
  if (!d) return fail(RGFM_EINVAL, "null descriptor");
  if (d->in_channels != 1 && d->in_channels != 3) return fail(RGFM_EINVAL, "in_channels must be 1 or 3");
  if (d->num_levels < 1 || d->num_levels > RGFM_MAX_LEVELS) return fail(RGFM_EINVAL, "1..4 levels supported");
  if (d->model_channels % 32 != 0 || d->model_channels > 256) return fail(RGFM_EINVAL, "model_channels must be a multiple of 32, <= 256");
  if (d->num_res_blocks < 1 || d->num_res_blocks > 8) return fail(RGFM_EINVAL, "num_res_blocks out of range");
  if (d->img_size < 4 || d->img_size > 64) return fail(RGFM_EINVAL, "img_size must be in 4..64");
  int s = d->img_size;
  for (int l = 0; l < d->num_levels; ++l) {
    if (d->channel_mult[l] < 1) return fail(RGFM_EINVAL, "channel_mult must be >= 1");
    if (d->model_channels * d->channel_mult[l] > 256) return fail(RGFM_EINVAL, "at most 256 channels per level");
    if (l < d->num_levels - 1) {
      if (s % 2) return fail(RGFM_EINVAL, "odd resolution before a downsample is not supported");
      s /= 2;
    }
  }
  return RGFM_OK;
}

void pack_one(const rgfm_unet* h, const ConvW& w, int mode, hipStream_t s) {
  launch_pack_conv(h->params + w.w_raw, h->packed + w.w_pk, w.cout, w.cin, w.taps, nt32_of(w.cout), s);
  if (mode == CONV_S2) launch_pack_conv_bx3_s2(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, s);
  else launch_pack_conv_bx3(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, w.taps, s);
  launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + w.w_hx2, h->hq + 4 * w.hq, w.cout, w.cin, w.taps, mode, s);
}

// after the pack launches: which convs may run on the fp16 path (one synchronising copy at create time)
int read_hx_flags(const float* hq_dev, int n, std::vector<ConvW*>& convs, hipStream_t s) {
  std::vector<float> host((size_t)n * 4);
  HIP_TRY(hipMemcpyAsync(host.data(), hq_dev, host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (ConvW* w : convs) w->hx_ok = host[(size_t)w->hq * 4 + 3] != 0.f;
  return RGFM_OK;
}

// fills the fp16-path fields of a conv launch (main conv `w`, optional fused 1x1 skip `sk`)
void fill_hx2(ConvArgs& c, const unsigned short* packedh, const float* hq, unsigned* flag, const ConvW& w, const ConvW* sk) {
  if (!w.hx_ok || (sk && !sk->hx_ok)) return;
  c.wpkh = packedh + w.w_hx2, c.hq = hq + 4 * w.hq, c.range_flag = flag;
  if (w.w_hx9) c.wpkh9 = packedh + (w.w_hx9 - 1);
  if (sk) c.wskiph = packedh + sk->w_hx2, c.hq_skip = hq + 4 * sk->hq;
}

// The normalised inputs of the fp16 path are S_A silu(gamma xhat + beta): in range for every trained net the reference
// can produce, but a conv whose norm parameters are tiny (the activation would sit in the fp16 subnormals) or huge is
// routed to the split-bf16 kernel once, at create, like a conv with out-of-window weights.  `host` = the parameter blob.
bool norm_params_ok(const std::vector<float>& host, size_t gw, size_t gb, int C) {
  float mg = 0.f, mb = 0.f;
  for (int i = 0; i < C; ++i) mg = std::max(mg, std::fabs(host[gw + i])), mb = std::max(mb, std::fabs(host[gb + i]));
  if (!(mg <= 3.0e38f) || !(mb <= 3.0e38f)) return false;
  const float hi = 8.f * mg + mb, lo = std::max(mg, mb);  // |gamma xhat + beta| for |xhat| <= 8; the activation's scale
  return hi < 1024.f && lo >= 0.015625f;
}

double conv_flops(int B, int HW, int cout, int kprod) { return 2.0 * B * HW * (double)cout * kprod; }

// A conv that has been described but not launched yet: if the next thing the walk asks for is the
// GroupNorm finalize of its output, the finalize is attached to it (ConvArgs::fin_*: the last wave per
// sample computes the scale/shift) and no gn_finalize launch happens -- each such launch is a full
// drain-and-refill bubble between two convs (measured: 18 % of the sampling call).
struct PendingConv {
  bool valid = false;
  ConvArgs c{};
  int mode = 0;
  double flops = 0.0;
};

void flush_conv(PendingConv& p, hipStream_t s) {
  if (!p.valid) return;
  p.valid = false;
  ProfScope ps(RGFM_KCLASS_CONV_MFMA, p.flops, s);
  launch_conv(p.c, p.mode, s);
}

// true when `p` can take the finalize of cat(its output, partner) itself
bool try_fuse_finalize(PendingConv& p, const float* first_data, const float* stats1, int C1, const float* gamma,
                       const float* beta, float* ab, unsigned* counter) {
  if (!g_modes.fuse_fin || !p.valid || !counter || p.c.out != first_data || !p.c.stats_out) return false;
  if ((p.c.Cout + C1) % 8 != 0 || p.c.Cout + C1 < 32 || p.c.Cout + C1 > 256) return false;  // <= 4 channels per lane
  if (p.c.g.nparts * (p.mode == CONV_T2 ? 4 : 1) > 16) return false;  // the finalizing wave holds <= 16 partials per channel
  p.c.fin_ab = ab, p.c.fin_counter = counter, p.c.fin_expected = conv_fin_expected(p.c, p.mode);
  p.c.fin_stats1 = stats1, p.c.fin_C1 = C1, p.c.fin_gamma = gamma, p.c.fin_beta = beta;
  return true;
}

// Consumer-side GroupNorm (default): the split-operand conv derives the scale/shift in its own prologue from the
// partial statistics (ConvArgs::gn_*); the table path (RGFM_GN=table) remains for the kernels that cannot
// (conv_out, RGFM_CONV=f32).  On failure the gn_* fields are cleared and the caller supplies `ab`.
bool try_consumer_gn(ConvArgs& c, int mode, const float* stats0, const float* stats1, int nparts0, const TileGeom& gg,
                     const float* gamma, const float* beta) {
  c.gn_stats0 = stats0, c.gn_stats1 = stats1, c.gn_gamma = gamma, c.gn_beta = beta, c.gn_nparts0 = nparts0, c.gn_g = gg;
  bool ok = false;
  if (g_modes.gn_consumer) {
    if (g_modes.conv == CONV_ARITH_HX2 && conv_hx2_supported(c, mode)) ok = conv_hx2_gn_supported(c, mode);
    else if (g_modes.conv != CONV_ARITH_F32) ok = conv_bx3_gn_supported(c, mode);
  }
  if (!ok) c.gn_stats0 = c.gn_stats1 = c.gn_gamma = c.gn_beta = nullptr;
  return ok;
}

struct NormRef {  // a GroupNorm in front of a conv: parameters (offsets into the blob)
  size_t gamma, beta;
};

struct UNetRun {
  rgfm_unet* h;
  int B;
  Bump* ws;
  hipStream_t s;
  const float* temb_row;  // table row(s) for this evaluation
  int temb_per_row;
  bool dry;
  unsigned* fin_counter = nullptr;  // [B] arrival counters (zero between launches) or null: separate gn_finalize
  const int* step_ptr = nullptr;    // device-side step counter: temb_row is then the table's first row (ConvArgs::step_ptr)
  PendingConv pend{};

  Tensor new_tensor(int C, int S) {
    Tensor t;
    t.C = C, t.S = S;
    const TileGeom g = make_geom(S, S);
    t.data = ws->f((size_t)B * S * S * C);
    t.stats = ws->f((size_t)B * g.nparts * C * 2);
    return t;
  }
  void record(const Tensor& t) {
    if (!dry && h->trace) h->acts.push_back({t.data, t.C, t.S, false});
  }
  float* finalize(const Tensor& a, const Tensor* b, size_t gamma, size_t beta, float* ab = nullptr) {
    const int C = a.C + (b ? b->C : 0);
    if (!ab) ab = ws->f((size_t)B * C * 2);
    if (dry) return ab;
    const bool fused = try_fuse_finalize(pend, a.data, b ? b->stats : nullptr, b ? b->C : 0, h->params + gamma,
                                         h->params + beta, ab, fin_counter);
    flush_conv(pend, s);
    if (fused) return ab;
    GnFinalizeArgs f{};
    f.stats0 = a.stats, f.stats1 = b ? b->stats : nullptr;
    f.C0 = a.C, f.C1 = b ? b->C : 0;
    f.groups = C < 8 ? C : 8;
    f.gamma = h->params + gamma, f.beta = h->params + beta;
    f.ab = ab, f.B = B, f.g = make_geom(a.S, a.S);
    ProfScope p(RGFM_KCLASS_OTHER, 0, s);
    launch_gn_finalize(f, s);
    return ab;
  }
  // generic 3x3 conv launch
  // raw_consumed: a later conv stages this output without a GroupNorm in front (ConvArgs::small_check)
  Tensor conv(const Tensor& a, const Tensor* b, const NormRef* norm, const ConvW& w, int mode, const float* temb,
              int res_mode, const Tensor* r0, const Tensor* r1, const ConvW* sk, bool raw_consumed) {
    const int So = mode == CONV_S2 ? a.S / 2 : (mode == CONV_UP2 ? a.S * 2 : a.S);
    Tensor o = new_tensor(w.cout, So);
    float* ab_buf = norm ? ws->f((size_t)B * (a.C + (b ? b->C : 0)) * 2) : nullptr;  // used by the table path only
    if (dry) return o;
    const float* ab = nullptr;
    ConvArgs c{};
    c.in0 = a.data, c.in1 = b ? b->data : nullptr;
    c.C0 = a.C, c.C1 = b ? b->C : 0;
    c.Hin = c.Win = a.S;
    c.ab = ab;
    c.wpk = h->packed + w.w_pk;
    c.wpk3 = h->packed3 + w.w_bx3;
    c.bias = h->params + w.b;
    c.temb = temb, c.temb_stride = h->temb_total, c.temb_per_row = temb_per_row;
    c.step_ptr = temb ? step_ptr : nullptr;
    c.res_mode = res_mode;
    if (res_mode) {
      c.res0 = r0->data, c.res1 = r1 ? r1->data : nullptr;
      c.R0 = r0->C, c.R1 = r1 ? r1->C : 0;
    }
    if (res_mode == 2) c.wskip = h->packed + sk->w_pk, c.wskip3 = h->packed3 + sk->w_bx3, c.skip_bias = h->params + sk->b;
    c.out = o.data, c.stats_out = o.stats;
    c.B = B, c.Cout = w.cout;
    c.g = make_geom(So, So);
    c.halo_px = mode == CONV_S2 ? c.g.spt * (2 * c.g.th + 1) * (2 * c.g.W + 1) : c.g.spt * (c.g.th + 2) * (c.g.W + 2);
    const int kprod = 9 * w.cin + (res_mode == 2 ? sk->cin : 0);
    fill_hx2(c, h->packedh, h->hq, h->range_flag, w, res_mode == 2 ? sk : nullptr);
    if (g_modes.conv == CONV_ARITH_HX2) c.range_flag = h->range_flag, c.small_check = raw_consumed ? 1 : 0;
    if (norm) {
      const TileGeom gg = make_geom(a.S, a.S);
      if (!try_consumer_gn(c, mode, a.stats, b ? b->stats : nullptr, gg.nparts, gg, h->params + norm->gamma,
                           h->params + norm->beta))
        c.ab = finalize(a, b, norm->gamma, norm->beta, ab_buf);  // (may attach itself to the pending producer)
    }
    flush_conv(pend, s);
    pend.valid = true, pend.c = c, pend.mode = mode;
    pend.flops = conv_flops(B, So * So, w.cout, kprod);
    return o;
  }
  // ResBlock.forward (unet_flexible.py:71-85)
  Tensor resblock(const ResW& r, const Tensor& a, const Tensor* b) {
    const NormRef n1{r.n1w, r.n1b}, n2{r.n2w, r.n2b};
    Tensor h1 = conv(a, b, &n1, r.c1, CONV_S1, dry ? nullptr : temb_row + r.temb_off, 0, nullptr, nullptr, nullptr, false);
    record(h1);
    // (a ResBlock's output is the residual stream: the next block's 1x1 skip, a Downsample or an Upsample reads it raw)
    Tensor o = conv(h1, nullptr, &n2, r.c2, CONV_S1, nullptr, r.has_skip ? 2 : 1, &a, b, r.has_skip ? &r.sk : nullptr, true);
    record(o);
    return o;
  }

  // FlexibleUNet.forward (unet_flexible.py:203-261).  Exactly one of v_out / x_state
  // may be non-null... both allowed: v_out receives the velocity, x_state the Euler update.
  int run(const float* x, float* v_out, float* x_state, float dt) {
    const rgfm_unet_desc& d = h->d;
    ModeScope mode_scope(h->conv_mode);
    if (!dry && h->trace) h->acts.clear();
    int S = d.img_size;
    Tensor cur = new_tensor(h->mc, S);
    if (!dry) {
      ConvInArgs ci{};
      ci.x = x, ci.w = h->params + h->icw, ci.bias = h->params + h->icb;
      ci.out = cur.data, ci.stats_out = cur.stats, ci.B = B, ci.C0 = h->mc, ci.g = make_geom(S, S);
      ci.range_flag = h->range_flag, ci.small_check = g_modes.conv == CONV_ARITH_HX2 ? 1 : 0;  // (the last decoder block's skip reads it raw)
      // algorithmic bytes: the NCHW image in, the NHWC map (+ its statistics) out
      const TileGeom g0 = make_geom(S, S);
      ProfScope p(d.in_channels == 1 ? RGFM_KCLASS_CONV_IN1 : RGFM_KCLASS_CONV_IN3,
                  4.0 * B * ((double)S * S * (d.in_channels + h->mc) + 2.0 * g0.nparts * h->mc), s);
      launch_conv_in(ci, d.in_channels, s);
    }
    record(cur);
    std::vector<Tensor> skips{cur};
    size_t e = 0;
    for (int l = 0; l < d.num_levels; ++l) {
      for (int r = 0; r < d.num_res_blocks; ++r) {
        cur = resblock(h->enc[e++], cur, nullptr);
        skips.push_back(cur);
      }
      if (l < d.num_levels - 1) {
        cur = conv(cur, nullptr, nullptr, h->down[l], CONV_S2, nullptr, 0, nullptr, nullptr, nullptr, true);
        record(cur);
        skips.push_back(cur);
      }
    }
    cur = resblock(h->mid[0], cur, nullptr);
    cur = resblock(h->mid[1], cur, nullptr);
    size_t di = 0, ui = 0;
    for (int l = d.num_levels - 1; l >= 0; --l) {
      for (int i = 0; i < d.num_res_blocks + 1; ++i) {
        Tensor sk = skips.back();
        skips.pop_back();
        cur = resblock(h->dec[di++], cur, &sk);
      }
      if (l > 0) {
        cur = conv(cur, nullptr, nullptr, h->up[ui++], CONV_UP2, nullptr, 0, nullptr, nullptr, nullptr, true);
        record(cur);
      }
    }
    float* ab = finalize(cur, nullptr, h->onw, h->onb);
    if (!dry) {
      flush_conv(pend, s);
      ConvOutArgs co{};
      co.in = cur.data, co.ab = ab, co.w = h->packed + h->ocw_pk, co.bias = h->params + h->ocb;
      co.v_out = v_out, co.x_state = x_state, co.dt = dt, co.B = B, co.Cin = cur.C;
      co.g = make_geom(cur.S, cur.S);
      co.halo_px = co.g.spt * (co.g.th + 2) * (co.g.W + 2);
      // algorithmic bytes: the NHWC map + its scale/shift in, the NCHW velocity out (fused Euler: state in and out)
      const double px = (double)B * cur.S * cur.S;
      ProfScope p(d.in_channels == 1 ? RGFM_KCLASS_CONV_OUT1 : RGFM_KCLASS_CONV_OUT3,
                  4.0 * (px * cur.C + 2.0 * B * cur.C + px * d.in_channels * ((v_out ? 1 : 0) + (x_state ? 2 : 0))), s);
      launch_conv_out(co, d.in_channels, s);
      if (h->trace && v_out) h->acts.push_back({v_out, d.in_channels, cur.S, true});
    }
    return RGFM_OK;
  }
};

size_t counter_bytes(int B) { return (((size_t)B * sizeof(unsigned)) + 255) & ~(size_t)255; }

size_t unet_eval_bytes(rgfm_unet* h, int B) {
  Bump b;
  UNetRun r{h, B, &b, nullptr, nullptr, 0, true};
  r.run(nullptr, nullptr, nullptr, 0.f);
  return b.off;
}

int launch_time_table(rgfm_unet* h, const float* t_dev, int num_steps, int step_begin, int nt, float* table,
                      hipStream_t s) {
  TimeEmbedArgs a{};
  a.params = h->params, a.freqs = h->freqs, a.mc = h->mc, a.temb = h->temb;
  a.te0w = (int)h->te0w, a.te0b = (int)h->te0b, a.te2w = (int)h->te2w, a.te2b = (int)h->te2b;
  a.lin = h->lin_dev, a.nlin = h->nlin, a.total = h->temb_total;
  a.t_dev = t_dev, a.num_steps = num_steps, a.step_begin = step_begin, a.table = table;
  ProfScope p(RGFM_KCLASS_OTHER, 0, s);
  launch_time_embed(a, nt, s);
  return RGFM_OK;
}

}  // namespace

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
  if (read_hx_flags(h->hq, h->n_hq, all, s) != RGFM_OK) return bail(RGFM_EHIP, "reading the fp16 scale records failed");
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

static size_t table_bytes(const rgfm_unet* h, int rows) {
  return (((size_t)rows * h->temb_total * sizeof(float)) + 255) & ~(size_t)255;
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

static size_t guid_dist_bytes(int batch, int n_mc) {  // sliced fp64 distances, RGFM_GUID_SLICES slices at most
  return (((size_t)RGFM_GUID_SLICES * batch * (n_mc > 0 ? n_mc : 1) * sizeof(double)) + 255) & ~(size_t)255;
}
static size_t guid_wbuf_bytes(int batch, int n_mc) {  // the step's importance weights [B][N]
  return (((size_t)batch * (n_mc > 0 ? n_mc : 1) * sizeof(float)) + 255) & ~(size_t)255;
}
static size_t guid_scratch_bytes(int batch, int n_mc) {  // + the weights and their row sums [B]
  return guid_dist_bytes(batch, n_mc) + guid_wbuf_bytes(batch, n_mc) + (((size_t)batch * sizeof(float) + 255) & ~(size_t)255);
}

extern "C" int rgfm_guidance_workspace_bytes(int batch, int n_mc, size_t* bytes) {
  if (!bytes || batch < 1 || n_mc < 1) return fail(RGFM_EINVAL, "bad argument");
  *bytes = guid_scratch_bytes(batch, n_mc);
  return RGFM_OK;
}

namespace {
// The step's scalars of the guidance block: Python-double arithmetic of the reference
// (sample_mnist_svhn.py:115,127,135,159,170), rounded to fp32 where a tensor op consumes it.
void guidance_scalars(double t, float* tf, float* s2, float* cden) {
  const double eps = 1e-3;
  const double sigma_t = 1.0 - t + eps;
  *tf = (float)t, *s2 = (float)(sigma_t * sigma_t), *cden = (float)(1.0 - t + eps);
}

int guidance_launch(const float* x, const float* y, float* vx, float* vy, const float* mx, const float* my,
                    const float* r, int B, int N, int dx, int dy, double t, double gamma, float* logp,
                    float* weights_out, float* xs, float* ys, float dt, hipStream_t s, const float* sched = nullptr,
                    const int* step_ptr = nullptr) {
  if (dx % 4 || dy % 4) return fail(RGFM_EINVAL, "flattened image sizes must be multiples of 4");
  if ((size_t)4 * N * sizeof(float) > 64 * 1024) return fail(RGFM_EINVAL, "n_mc too large (max 4096)");
  // Python-double scalar arithmetic of the reference (sample_mnist_svhn.py:115,127,135,159,170),
  // rounded to fp32 where a tensor op consumes it.
  GuidanceArgs a{};
  a.x = x, a.y = y, a.vx = vx, a.vy = vy, a.mc_x1 = mx, a.mc_y1 = my, a.mc_ratios = r;
  a.B = B, a.N = N, a.dx = dx, a.dy = dy;
  guidance_scalars(t, &a.tf, &a.s2, &a.cden);
  a.sched = sched, a.step_ptr = sched ? step_ptr : nullptr;
  a.g1 = (float)(1.0 - gamma), a.g2 = (float)gamma;
  a.dist = reinterpret_cast<double*>(logp), a.weights_out = weights_out, a.x_state = xs, a.y_state = ys, a.dt = dt;
  a.wbuf = reinterpret_cast<float*>(reinterpret_cast<char*>(logp) + guid_dist_bytes(B, N));
  a.wsum = reinterpret_cast<float*>(reinterpret_cast<char*>(a.wbuf) + guid_wbuf_bytes(B, N));
  a.slice_len = 512;  // 512-element slices (8 x 16 x 4 = 512 workgroups at the benchmark shape) unless that needs more than RGFM_GUID_SLICES of them
  while ((dx + a.slice_len - 1) / a.slice_len + (dy + a.slice_len - 1) / a.slice_len > RGFM_GUID_SLICES) a.slice_len *= 2;
  a.nsx = (dx + a.slice_len - 1) / a.slice_len, a.nsy = (dy + a.slice_len - 1) / a.slice_len;
  // algorithmic bytes.  logp: rows of x, y and the MC set in, the sliced fp64 distances out.  apply: the
  // distances, x, y, v and the MC set in, the new state (or velocity) out.
  const double D = (double)dx + dy, dist_b = 8.0 * (a.nsx + a.nsy) * (double)B * N;
  {
    ProfScope p(RGFM_KCLASS_GUID_LOGP, 4.0 * (B + N) * D + dist_b, s);
    launch_guid_logp(a, s);
  }
  {
    ProfScope p(RGFM_KCLASS_GUID_APPLY, dist_b + 4.0 * N * D + 4.0 * B * D * 3.0, s);
    launch_guid_apply(a, s);
  }
  return RGFM_OK;
}
}  // namespace

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

namespace {
// Shared Euler loop of paired_sampler (src/utils/flow_utils.py:186-278 with the guidance of
// src/sample_mnist_svhn.py:117-175): eval_x / eval_y enqueue one velocity-net evaluation of step i
// on the given stream, writing the raw velocity (guided steps) or the fused Euler update.
// Graph replay (U-Net pairs, RGFM_GRAPH=1; never with active kernel timers): every guided step enqueues the
// same ~135 launches with the same arguments except the time-table row and three guidance scalars.  Those are read
// on the device through a step counter (`gstate`: [0] the counter, [64..] the per-step scalars), so the first guided
// step is captured once -- both streams, fork and join included -- into a hipGraph and every guided step is one
// hipGraphLaunch.  Results are bit-identical to the kernel-by-kernel path (same kernels, same arguments).
template <class EvalX, class EvalY>
int pair_loop(EvalX&& eval_x, EvalY&& eval_y, float* x_inout, float* y_inout, const float* mc_x1,
              const float* mc_y1, const float* mc_ratios, int n_mc, int batch, int num_steps, double gamma,
              int step_begin, int ns, int dx, int dy, float* vx, float* vy, float* logp, hipStream_t caller,
              float* gstate = nullptr) {
  const double dtd = 1.0 / (double)num_steps;
  const float dt = (float)dtd;
  // The two velocity nets of a step are independent (reference :119-121): the second one runs on a
  // side stream forked from / joined back into the caller's stream every step, which fills the CUs
  // that one net's small-grid launches (8x8 level, kernel tails) leave idle.
  // (the stream and its fork/join events belong to the device's DevState, created with the first handle)
  DevState* ds = cur_dev();
  if (!ds) return fail(RGFM_EINVAL, "no handle has been created on the current device");
  const bool overlap = g_modes.overlap;
  hipStream_t side = ds->side;
  hipEvent_t ev_fork = ds->fork, ev_join = ds->join;
  const bool use_graph = gstate && g_modes.graph && !g_prof.on && n_mc > 0 && ns >= 4;
  hipStream_t s = caller;
  if (use_graph && caller == nullptr) {  // (see DevState::main)
    s = ds->main;
    HIP_TRY(hipEventRecord(ds->main_fork, caller));
    HIP_TRY(hipStreamWaitEvent(s, ds->main_fork, 0));
  }
  int* step_dev = nullptr;
  float* sched_dev = nullptr;
  if (use_graph) {
    // graphs of earlier calls: release them once the device is past their last replay
    if (!ds->graphs.empty() && hipEventQuery(ds->graph_done) == hipSuccess) {
      for (auto& g : ds->graphs) (void)hipGraphExecDestroy(g.first), (void)hipGraphDestroy(g.second);
      ds->graphs.clear();
    }
    step_dev = reinterpret_cast<int*>(gstate);
    sched_dev = gstate + 64;
    HIP_TRY(hipMemsetAsync(step_dev, 0, 256, s));
    launch_guid_schedule(sched_dev, step_begin, ns, num_steps, s);
  }
  auto one_step = [&](int i, bool guided) -> int {
    hipStream_t sy = overlap ? side : s;
    if (overlap) {
      HIP_TRY(hipEventRecord(ev_fork, s));
      HIP_TRY(hipStreamWaitEvent(side, ev_fork, 0));
    }
    int rc = eval_y(i, sy, guided ? vy : nullptr, guided ? nullptr : y_inout, dt, step_dev);
    if (rc) return rc;
    if (overlap) HIP_TRY(hipEventRecord(ev_join, side));
    rc = eval_x(i, s, guided ? vx : nullptr, guided ? nullptr : x_inout, dt, step_dev);
    if (rc) return rc;
    if (overlap) HIP_TRY(hipStreamWaitEvent(s, ev_join, 0));
    if (guided) {
      const double t = (double)(step_begin + i) * dtd;
      rc = guidance_launch(x_inout, y_inout, vx, vy, mc_x1, mc_y1, mc_ratios, batch, n_mc, dx, dy, t, gamma, logp,
                           nullptr, x_inout, y_inout, dt, s, sched_dev, step_dev);
      if (rc) return rc;
    }
    if (step_dev) launch_step_inc(step_dev, s);
    return RGFM_OK;
  };
  hipGraphExec_t exec = nullptr;
  // On EVERY exit path -- also the error returns inside the loop -- the work already enqueued must stay ordered: the
  // event that guards the destruction of this call's graph is recorded behind its last launch, and the caller's
  // stream is joined behind whatever ran on the device's `main` stream (a stale graph_done could let the
  // next call destroy a graph that is still executing).
  struct ExitGuard {
    DevState* ds;
    hipStream_t s, caller;
    hipGraphExec_t* exec;
    ~ExitGuard() {
      if (*exec) (void)hipEventRecord(ds->graph_done, s);
      if (s != caller) {
        (void)hipEventRecord(ds->main_join, s);
        (void)hipStreamWaitEvent(caller, ds->main_join, 0);
      }
    }
  } exit_guard{ds, s, caller, &exec};
  for (int i = 0; i < ns; ++i) {
    const double t = (double)(step_begin + i) * dtd;
    const bool guided = n_mc > 0 && t > 1e-3;  // `t > eps` test of the reference (:124)
    if (use_graph && guided) {
      if (!exec) {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        const int rc = one_step(i, true);
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc) {  // (a captured graph that will never run: nothing refers to it)
          if (graph) (void)hipGraphDestroy(graph);
          return rc;
        }
        if (ce != hipSuccess || !graph) {
          if (graph) (void)hipGraphDestroy(graph);
          return fail(RGFM_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        }
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
          exec = nullptr;
          (void)hipGraphDestroy(graph);
          return fail(RGFM_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(ie));
        }
        ds->graphs.push_back({exec, graph});
      }
      HIP_TRY(hipGraphLaunch(exec, s));
      continue;
    }
    const int rc = one_step(i, guided);
    if (rc) return rc;
  }
  // (the guard's destructor records graph_done and joins `main` into the caller's stream -- on this path too, so that a
  // failing record cannot skip the join)
  HIP_TRY(hipGetLastError());
  return RGFM_OK;
}
}  // namespace

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
    float* tmp = nullptr;
    if (hipMalloc(&tmp, (size_t)256 * 256 * 9 * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(tmp)");
    for (const auto* e : {&h->ex, &h->ey}) {
      for (size_t i = 1; i < e->convs.size(); ++i) {
        const auto& cv = e->convs[i];
        launch_conv_weight_transpose(h->params + cv.w.w_raw, tmp, cv.w.cout, cv.w.cin, s);
        launch_pack_conv(tmp, h->gradw + cv.wt_pk, cv.w.cin, cv.w.cout, 9, nt32_of(cv.w.cin), s);
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
    if (hipMalloc(&h->packedh, (h->n_packedh + 8) * sizeof(unsigned short)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(packedh)");
    if (hipMalloc(&h->hq, ((size_t)h->n_hq * 4 + 4) * sizeof(float)) != hipSuccess) return bail(RGFM_ENOMEM, "hipMalloc(hq)");
    std::vector<ConvW*> all;
    for (auto* e : {&h->ex, &h->ey})
      for (size_t i = 1; i < e->convs.size(); ++i) {
        ConvW& w = e->convs[i].w;
        launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + w.w_hx2, h->hq + 4 * w.hq, w.cout, w.cin, 9, CONV_S1, s);
        all.push_back(&w);
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
      if (!dry) {
        if (k.ab) {  // GroupNorm encoder: SiLU' (and the max-pool routing) at u = a z + b, then the norm's backward in place
          launch_grad_act_gn(g, k.z, k.ab, gz, n, k.S, k.C, mode, s);
          launch_gn_bwd(gz, k.z, h->params + cv.nw, k.mr, n, k.S * k.S, k.C, 8, s);
        } else {
          launch_grad_act(g, k.z, h->bn + cv.bn_scale, gz, n, k.S, k.C, mode, s);
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
          launch_conv_mfma(c, CONV_S1, s);
        }
        g = gin;
      }
    }
  }

  void run(const float* x, const float* y, float* gx, float* gy, float* log_ratio) {
    const int F = h->d.feature_dim;
    ab1 = ws->f((size_t)n * 256 * 2);
    if (!dry) launch_fill_ab_identity(ab1, (size_t)n * 256, s);
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
