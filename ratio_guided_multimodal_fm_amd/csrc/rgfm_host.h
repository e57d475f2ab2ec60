// rgfm_host.h -- host-side internals shared by the translation units of the C ABI (api_*.cpp): error text, the
// hipEvent kernel-class timers, workspace carving, the per-thread run-time switches, per-device state, conv dispatch,
// the U-Net handle and its network walk (used by the samplers and the gradient-guided loop too), the guidance launch and
// the shared paired Euler loop.  Not part of the public ABI (that is include/rgfm.h); everything here is `inline` /
// C++17 inline variables, so every unit sees ONE definition.
//
//   api_core.cpp     rgfm_last_error, rgfm_abi_version, rgfm_profile_*
//   api_unet.cpp     rgfm_unet_*            (create / forward / trace hooks)
//   api_sampler.cpp  rgfm_sample_single, rgfm_guidance_*, rgfm_sample_pair
//   api_ratio.cpp    rgfm_ratio_*, gradient of log r, rgfm_sample_pair_grad
//   api_fmnet.cpp    rgfm_fmnet_*           (FlowMatchingModel)
//
// No PyTorch types, no allocation and no synchronisation inside forward / sample calls (everything is carved from the
// caller's workspace, stream-ordered).
#pragma once
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
inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
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


// ------------------------------------------------------------------ profiling (bench support)
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
};
inline Prof g_prof;

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

inline int prof_collect() {
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

inline double prof_union_ms(int k) {
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


// ------------------------------------------------------------------ small helpers

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
  void* p = nullptr;     // the same map in P format (ConvArgs::pout / pin0), when its producer wrote it
  bool p_valid = false;
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
  // Upsample convs (nearest x 2, then 3x3: unet_flexible.py:107-108) a second time as the EQUIVALENT ConvTranspose2d(4, 2, 1)
  // -- four 2x2-tap parity classes over the INPUT raster, 16 instead of 36 tap products per input pixel (launch_up2_as_deconv)
  // ResBlock convs with Cout % 64 == 0 a second time as Winograd F(2x2, 3x3) images (conv_mfma_hx2w.hip)
  size_t w_w = 0;           // offset (fp16 elements) of the transformed two-plane image, + 1 (0: none)
  int hq_w = 0;             // its scale record
  bool w_ok = false;        // ... inside the fp16 path's range
  size_t w_t2 = 0;          // offset of its packed two-plane image, + 1 (0: none)
  int hq_t2 = 0;            // its scale record
  bool t2_ok = false;       // ... inside the fp16 path's range
};

struct ResW {
  int cin = 0, cout = 0;
  size_t n1w, n1b, n2w, n2b, tw, tb;
  ConvW c1, c2, sk;
  bool has_skip = false;
  int temb_off = 0;
};

inline int nt32_of(int cout) { return (cout % 64 == 0) ? 2 : 1; }

inline bool on_gfx950() {
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
//   RGFM_HX2P=0, RGFM_HX2Q=0, RGFM_HX2S=0, RGFM_HX2C=0, RGFM_HX2D=0, RGFM_GRAPH=1: A/B switches of the pipelined fp16
//               kernel, its four-waves-per-SIMD version, the stride-2 kernel, the 8x8-level kernel, the P-format hand-over
//               between a ResBlock's two convs (conv_mfma_hx2d.hip) and the hipGraph replay.
enum { CONV_ARITH_HX2 = 0, CONV_ARITH_BX3 = 1, CONV_ARITH_F32 = 2 };
struct Modes {
  int conv = CONV_ARITH_HX2;
  bool overlap = true, fuse_fin = true, gn_consumer = true;
  bool pipelined = true;  // RGFM_HX2P=0: the fp16 convs on conv_mfma_hx2_kernel only (A/B switch)
  bool quad = true;       // RGFM_HX2Q=0: no four-waves-per-SIMD workgroups (conv_mfma_hx2q.hip; A/B switch, bit-identical)
  bool c8 = true;         // RGFM_HX2C=0: the 8x8 level on conv_mfma_hx2p_kernel (A/B switch)
  bool s2 = true;         // RGFM_HX2S=0: the Downsample convs on conv_mfma_hx2_kernel<*, CONV_S2, *> (A/B switch; same to 1e-6)
  bool pfmt = true;       // RGFM_HX2D=0: no P-format hand-over conv1 -> conv2 at the 16x16 / 8x8 levels (A/B switch)
  bool wino = false;      // RGFM_WINO=1: the Winograd form of the long-K stride-1 convs (conv_mfma_hx2w.hip) -- opt-in: faster per
                          // layer in isolation at 32x32, slower on the whole bench
  bool up_t2 = true;      // RGFM_UP_T2=0: the Upsample convs as nine taps over the upsampled raster instead of four parity classes (A/B switch)
  bool rev_hx2 = true;    // RGFM_REV_HX2=0: the reverse convs of the gradient-guided sampler on the exact fp32 MFMA (A/B switch)
  bool graph = false;     // RGFM_GRAPH=1: the guided steps of the paired U-Net loop replayed from one captured hipGraph
                          // (bit-identical; measured 0.995-1.002x of the kernel-by-kernel path: the host is not the bottleneck)
};
inline thread_local Modes g_modes;  // per host thread: a handle's own conv arithmetic (rgfm_*_set_conv_mode) overrides it per network walk
inline void refresh_modes() {
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
  e = getenv("RGFM_HX2D");
  m.pfmt = !(e && e[0] == '0');
  conv_hx2d_set(e && e[0] == '1' ? 1 : (e && e[0] == '2' ? 2 : 3));  // (1 / 2: one cut of conv_mfma_hx2d.hip everywhere -- A/B; process-wide, tools only)
  e = getenv("RGFM_WINO");
  m.wino = e && e[0] == '1';
  e = getenv("RGFM_UP_T2");
  m.up_t2 = !(e && e[0] == '0');
  e = getenv("RGFM_REV_HX2");
  m.rev_hx2 = !(e && e[0] == '0');
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
  void* zeros = nullptr;  // 256 zero bytes: the source of the padding records of conv_mfma_hx2d_kernel's halo DMA
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
inline DevState g_dev[MAX_DEVICES];

inline DevState* cur_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
  return g_dev[dev].init ? &g_dev[dev] : nullptr;
}

inline int ensure_init() {
  if (!on_gfx950()) return fail(RGFM_ENODEVICE, "librgfm_hip needs a gfx950 (MI355X) device; none is current");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) return fail(RGFM_EINVAL, "device ordinal %d out of range (max %d)", dev, MAX_DEVICES - 1);
  DevState& d = g_dev[dev];
  if (!d.init) {
    if (conv_mfma_init() != 0 || conv_bx3_init() != 0 || conv_hx2_init() != 0 || conv_hx2p_init() != 0 || conv_hx2q_init() != 0 ||
        conv_hx2s_init() != 0 || conv_hx2c_init() != 0 || conv_hx2d_init() != 0 || conv_hx2w_init() != 0 || guid_apply_init() != 0)
      return fail(RGFM_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) d.num_cus = p.multiProcessorCount;
    conv_hx2p_set_half(d.num_cus);  // launches with fewer 128-channel workgroups than CUs take 64-channel workgroups
    if (const char* e = getenv("RGFM_HX2P_HALF")) conv_hx2p_set_half(atoi(e));  // (A/B of that threshold; every cut gives the same bits)
    HIP_TRY(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&d.zeros, 256));
    HIP_TRY(hipMemset(d.zeros, 0, 256));
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

// Would launch_conv send this launch to a kernel that can write ConvArgs::pout (the P-format hand-over)?
inline bool p_producer_ok(const ConvArgs& c, int mode) {
  if (g_modes.conv != CONV_ARITH_HX2 || !g_modes.pipelined || !g_modes.pfmt) return false;
  if (g_modes.s2 && conv_hx2s_supported(c, mode)) return false;
  if (g_modes.c8 && conv_hx2c_supported(c, mode)) return true;
  if (g_modes.quad && conv_hx2q_supported(c, mode)) return false;
  // conv_mfma_hx2p_kernel: 16x16 rasters (a tile = one whole sample), whole power-of-two groups per 32-channel wave block
  if (mode == CONV_S1 && c.g.W == 16 && c.g.H == 16 && c.g.spt == 1 && c.g.tps == 1 && (c.Cout == 64 || c.Cout == 128 || c.Cout == 256) &&
      !c.ep_scale && !c.fin_ab && conv_hx2p_supported(c, mode))
    return true;
  return false;
}

// Where the Winograd form is the faster kernel (tools/kbench, B = 512, profiles/r04_kbench/hx2w_vs_direct.txt): its K loop
// costs less per chunk than the direct kernels', its epilogue (the output transform through LDS) more -- so the long-K
// layers: 128 or more input channels.  By layer shape only, never by batch.
inline bool hx2w_pays(const ConvArgs& c) {
  static const int w32_only = getenv("RGFM_WINO_W32") ? atoi(getenv("RGFM_WINO_W32")) : 0;  // (A/B: the 32x32 layers only)
  return c.wpkw != nullptr && c.C0 + c.C1 >= 128 && (!w32_only || c.g.W == 32);
}

inline void launch_conv(const ConvArgs& c, int mode, hipStream_t s) {
  if (c.pin0) {  // (P-format input: only conv_mfma_hx2d_kernel reads it; the walk has checked conv_hx2d_supported)
    launch_conv_hx2d(c, s);
    return;
  }
  if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.wino && hx2w_pays(c) && conv_hx2w_supported(c, mode)) {
    launch_conv_hx2w(c, s);
    return;
  }
  if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.s2 && conv_hx2s_supported(c, mode)) launch_conv_hx2s(c, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.c8 && conv_hx2c_supported(c, mode)) launch_conv_hx2c(c, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.quad && conv_hx2q_supported(c, mode)) launch_conv_hx2q(c, mode, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && conv_hx2p_supported(c, mode)) launch_conv_hx2p(c, mode, s);
  else if (g_modes.conv == CONV_ARITH_HX2 && conv_hx2_supported(c, mode)) launch_conv_hx2(c, mode, s);
  else if (g_modes.conv != CONV_ARITH_F32 && conv_bx3_supported(c, mode)) launch_conv_bx3(c, mode, s);
  else launch_conv_mfma(c, mode, s);
}



// ------------------------------------------------------------------ fp16-path range flag
// Every U-Net / FlowMatchingModel handle owns one device word.  conv_mfma_hx2*.hip OR into it: bit 0 when a staged
// activation reaches |S_A a| >= 32768 (fp16 would overflow), bit 1 when an output that a later conv stages raw is too
// small for the two-plane representation (ConvArgs::small_check).  Either way the results of the handle's calls since
// the last reset are not fp32-class and the caller repeats them with the handle switched to RGFM_CONV_BX3 (bit 0: fp32's
// exponent range at the top) or to RGFM_CONV_F32 (bit 1: the exact fp32 matrix-core convs, the reference's arithmetic
// at any magnitude) -- what _engine._range_guarded and INTEGRATION.md section B do.  Per handle, so that two threads /
// two engines on one device cannot consume each other's flag.
inline int read_flag_word(unsigned* word, int* flagged, int reset, hipStream_t s) {
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
inline int alloc_flag_word(unsigned** word) {
  HIP_TRY(hipMalloc(word, 256));
  HIP_TRY(hipMemset(*word, 0, 256));
  return RGFM_OK;
}
inline int check_conv_mode(int mode) {
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
  int wino_convs = 0;   // convs of the latest walk described for the Winograd kernel (rgfm_unet_wino_convs)
  int p_handovers = 0;  // ResBlocks of the latest walk whose conv1 -> conv2 hand-over took the P format (rgfm_unet_p_handovers)
  struct Act {
    float* data;
    int C, S;
    bool nchw;
  };
  std::vector<Act> acts;  // filled by the last (non-dry) run in trace mode
};


// Walks the parameter registration order of FlexibleUNet.__init__
// (reference src/models/unet_flexible.py:146-201; UNetMNIST, src/models/unet.py:155-214,
// is identical) and records blob offsets.  Returns the total float count.
inline size_t plan_unet(const rgfm_unet_desc& d, rgfm_unet* h) {
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
    if (cout % 64 == 0 && cin % KC == 0) {  // the Winograd images (16 positions instead of 9 taps)
      r.c1.w_w = ph.take((size_t)cout * cin * 16 * 2) + 1, r.c1.hq_w = nhq++;
      r.c2.w_w = ph.take((size_t)cout * cout * 16 * 2) + 1, r.c2.hq_w = nhq++;
    }
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
  for (int uc : up_ch) {
    ConvW w = conv(uc, uc, 9);
    w.w_t2 = ph.take((size_t)uc * uc * 16 * 2) + 1;  // (the Upsample convs twice: nine taps, and the parity-class form)
    w.hq_t2 = nhq++;
    up.push_back(w);
  }
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

inline int check_desc(const rgfm_unet_desc* d) {
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

inline void pack_one(const rgfm_unet* h, const ConvW& w, int mode, hipStream_t s) {
  launch_pack_conv(h->params + w.w_raw, h->packed + w.w_pk, w.cout, w.cin, w.taps, nt32_of(w.cout), s);
  if (mode == CONV_S2) launch_pack_conv_bx3_s2(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, s);
  else launch_pack_conv_bx3(h->params + w.w_raw, h->packed3 + w.w_bx3, w.cout, w.cin, w.taps, s);
  launch_pack_conv_hx2(h->params + w.w_raw, h->packedh + w.w_hx2, h->hq + 4 * w.hq, w.cout, w.cin, w.taps, mode, s);
}

// after the pack launches: which convs may run on the fp16 path (one synchronising copy at create time)
inline int read_hx_flags(const float* hq_dev, int n, std::vector<ConvW*>& convs, hipStream_t s) {
  std::vector<float> host((size_t)n * 4);
  HIP_TRY(hipMemcpyAsync(host.data(), hq_dev, host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (ConvW* w : convs) w->hx_ok = host[(size_t)w->hq * 4 + 3] != 0.f;
  return RGFM_OK;
}

// fills the fp16-path fields of a conv launch (main conv `w`, optional fused 1x1 skip `sk`)
inline void fill_hx2(ConvArgs& c, const unsigned short* packedh, const float* hq, unsigned* flag, const ConvW& w, const ConvW* sk) {
  if (!w.hx_ok || (sk && !sk->hx_ok)) return;
  c.wpkh = packedh + w.w_hx2, c.hq = hq + 4 * w.hq, c.range_flag = flag;
  if (w.w_hx9) c.wpkh9 = packedh + (w.w_hx9 - 1);
  if (w.w_w && w.w_ok && !sk) c.wpkw = packedh + (w.w_w - 1), c.hqw = hq + 4 * w.hq_w;
  if (sk) c.wskiph = packedh + sk->w_hx2, c.hq_skip = hq + 4 * sk->hq;
}

// The normalised inputs of the fp16 path are S_A silu(gamma xhat + beta): in range for every trained net the reference
// can produce, but a conv whose norm parameters are tiny (the activation would sit in the fp16 subnormals) or huge is
// routed to the split-bf16 kernel once, at create, like a conv with out-of-window weights.  `host` = the parameter blob.
inline bool norm_params_ok(const std::vector<float>& host, size_t gw, size_t gb, int C) {
  float mg = 0.f, mb = 0.f;
  for (int i = 0; i < C; ++i) mg = std::max(mg, std::fabs(host[gw + i])), mb = std::max(mb, std::fabs(host[gb + i]));
  if (!(mg <= 3.0e38f) || !(mb <= 3.0e38f)) return false;
  const float hi = 8.f * mg + mb, lo = std::max(mg, mb);  // |gamma xhat + beta| for |xhat| <= 8; the activation's scale
  return hi < 1024.f && lo >= 0.015625f;
}

inline double conv_flops(int B, int HW, int cout, int kprod) { return 2.0 * B * HW * (double)cout * kprod; }

// True when the statistics of a map written by a CONV_T2 launch over an S x S raster (4 parity classes x that raster's
// parts) are indistinguishable, part for part, from those of a plain 2S x 2S map: same number of parts, same pixel
// counts.  (32 <- 16 and 16 <- 8: 64-pixel parts on both sides.)
inline bool up_parts_match(int S) {
  const TileGeom gl = make_geom(S, S), gf = make_geom(2 * S, 2 * S);
  if (gf.nparts != 4 * gl.nparts || gf.nparts > 16) return false;
  for (int p = 0; p < gf.nparts; ++p)
    if (geom_part_count(gf, p) != geom_part_count(gl, p % gl.nparts)) return false;
  return true;
}

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

inline void flush_conv(PendingConv& p, hipStream_t s) {
  if (!p.valid) return;
  p.valid = false;
  ProfScope ps(RGFM_KCLASS_CONV_MFMA, p.flops, s);
  launch_conv(p.c, p.mode, s);
}

// true when `p` can take the finalize of cat(its output, partner) itself
inline bool try_fuse_finalize(PendingConv& p, const float* first_data, const float* stats1, int C1, const float* gamma,
                       const float* beta, float* ab, unsigned* counter) {
  if (!g_modes.fuse_fin || !p.valid || !counter || p.c.out != first_data || !p.c.stats_out) return false;
  if (p.c.pin0) return false;  // (conv_mfma_hx2d_kernel has no producer-side finalize: a gn_finalize launch follows)
  if ((p.c.Cout + C1) % 8 != 0 || p.c.Cout + C1 < 32 || p.c.Cout + C1 > 256) return false;  // <= 4 channels per lane
  if (p.c.g.nparts * (p.mode == CONV_T2 ? 4 : 1) > 16) return false;  // the finalizing wave holds <= 16 partials per channel
  p.c.fin_ab = ab, p.c.fin_counter = counter, p.c.fin_expected = conv_fin_expected(p.c, p.mode);
  p.c.fin_stats1 = stats1, p.c.fin_C1 = C1, p.c.fin_gamma = gamma, p.c.fin_beta = beta;
  return true;
}

// Consumer-side GroupNorm (default): the split-operand conv derives the scale/shift in its own prologue from the
// partial statistics (ConvArgs::gn_*); the table path (RGFM_GN=table) remains for the kernels that cannot
// (conv_out, RGFM_CONV=f32).  On failure the gn_* fields are cleared and the caller supplies `ab`.
inline bool try_consumer_gn(ConvArgs& c, int mode, const float* stats0, const float* stats1, int nparts0, const TileGeom& gg,
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
    if (mode == CONV_UP2 && g_modes.conv == CONV_ARITH_HX2 && g_modes.up_t2 && w.w_t2 && w.t2_ok && up_parts_match(a.S)) {
      // nearest x 2 + 3x3 == ConvTranspose2d(4, 2, 1) with summed taps: the parity-class form over the INPUT raster
      // (4 / 9 of the matrix work, a quarter of the halo per output).  Its statistics parts -- four classes x the input
      // raster's parts -- have the sizes of the output raster's parts (up_parts_match), so readers see a plain map.
      ConvArgs ct = c;
      ct.g = make_geom(a.S, a.S);
      ct.halo_px = ct.g.spt * (ct.g.th + 2) * (ct.g.W + 2);
      ct.wpkh = h->packedh + (w.w_t2 - 1), ct.hq = h->hq + 4 * w.hq_t2;
      if (conv_hx2_supported(ct, CONV_T2)) c = ct, mode = CONV_T2;
    }
    bool p_in = false;
    if (norm && a.p_valid && !b) {
      // the producer (still pending) can hand this input over in P format: take it if conv_mfma_hx2d_kernel can run this
      // conv, otherwise the producer goes back to the fp32 map + statistics
      ConvArgs cp = c;
      DevState* ds = cur_dev();
      cp.pin0 = a.p, cp.zeros = ds ? ds->zeros : nullptr;
      if (pend.valid && pend.c.pout == a.p && conv_hx2d_supported(cp, mode)) {
        c = cp, p_in = true;
        h->p_handovers += 1;
        if (!h->trace) pend.c.out = nullptr, pend.c.stats_out = nullptr;  // (nothing else reads the fp32 map)
      } else if (pend.valid && pend.c.pout == a.p) {
        pend.c.pout = nullptr;
      }
    }
    if (norm && !p_in) {
      const TileGeom gg = make_geom(a.S, a.S);
      if (!try_consumer_gn(c, mode, a.stats, b ? b->stats : nullptr, gg.nparts, gg, h->params + norm->gamma,
                           h->params + norm->beta))
        c.ab = finalize(a, b, norm->gamma, norm->beta, ab_buf);  // (may attach itself to the pending producer)
    }
    flush_conv(pend, s);
    if (g_modes.conv == CONV_ARITH_HX2 && g_modes.pipelined && g_modes.wino && hx2w_pays(c) && conv_hx2w_supported(c, mode)) h->wino_convs += 1;
    pend.valid = true, pend.c = c, pend.mode = mode;
    pend.flops = conv_flops(B, So * So, w.cout, kprod);
    return o;
  }
  // ResBlock.forward (unet_flexible.py:71-85)
  Tensor resblock(const ResW& r, const Tensor& a, const Tensor* b) {
    const NormRef n1{r.n1w, r.n1b}, n2{r.n2w, r.n2b};
    Tensor h1 = conv(a, b, &n1, r.c1, CONV_S1, dry ? nullptr : temb_row + r.temb_off, 0, nullptr, nullptr, nullptr, false);
    // P-format hand-over (conv_mfma_hx2d.hip): at the 16x16 / 8x8 levels conv1's workgroups own whole (sample, group) sets
    // of h1, whose only reader is conv2 through norm2 + SiLU (unet_flexible.py:79-81) -- conv1 writes silu(norm2(h1))
    // already split, conv2 stages it by LDS-DMA.  The buffer is carved whenever the SHAPE allows it (dry and real walks
    // alike); whether the two launches take the hand-over is decided from the kernels they are dispatched to.
    // Where it pays (tools/kbench, profiles/r04_kbench/p_producer_overhead.txt + hx2d_cuts.txt, B = 512): writing the P
    // format costs the producer 2.2 us at 8x8 (of 36), 3.5 us at 16x16 / 64 channels, 6.5 us at 16x16 / 128 channels (its
    // epilogue is vector-ALU-bound: SiLU + split of every output); the consumer gains 6.7 / 5.5 - 6.5 / 3.6 - 4.4 us.  So:
    // the 8x8 level and 64-channel 16x16 layers; 128-channel 16x16 layers keep the fp32 hand-over.
    if ((a.S == 8 || (a.S == 16 && r.cout == 64)) && r.cout % 64 == 0) {
      h1.p = ws->f((size_t)B * a.S * a.S * r.cout);
      if (!dry && pend.valid && pend.c.out == h1.data && r.c1.hx_ok && r.c2.hx_ok && (!r.has_skip || r.sk.hx_ok)) {
        pend.c.pout = h1.p, pend.c.pn_gamma = h->params + r.n2w, pend.c.pn_beta = h->params + r.n2b;
        if (p_producer_ok(pend.c, pend.mode)) h1.p_valid = true;
        else pend.c.pout = nullptr;
      }
    }
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
    if (!dry) h->p_handovers = 0, h->wino_convs = 0;
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

inline size_t counter_bytes(int B) { return (((size_t)B * sizeof(unsigned)) + 255) & ~(size_t)255; }

inline size_t unet_eval_bytes(rgfm_unet* h, int B) {
  Bump b;
  UNetRun r{h, B, &b, nullptr, nullptr, 0, true};
  r.run(nullptr, nullptr, nullptr, 0.f);
  return b.off;
}

inline int launch_time_table(rgfm_unet* h, const float* t_dev, int num_steps, int step_begin, int nt, float* table,
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



inline size_t table_bytes(const rgfm_unet* h, int rows) {
  return (((size_t)rows * h->temb_total * sizeof(float)) + 255) & ~(size_t)255;
}


inline size_t guid_dist_bytes(int batch, int n_mc) {  // sliced fp64 distances, RGFM_GUID_SLICES slices at most
  return (((size_t)RGFM_GUID_SLICES * batch * (n_mc > 0 ? n_mc : 1) * sizeof(double)) + 255) & ~(size_t)255;
}
inline size_t guid_wbuf_bytes(int batch, int n_mc) {  // the step's importance weights [B][N]
  return (((size_t)batch * (n_mc > 0 ? n_mc : 1) * sizeof(float)) + 255) & ~(size_t)255;
}
inline size_t guid_scratch_bytes(int batch, int n_mc) {  // + the weights and their row sums [B]
  return guid_dist_bytes(batch, n_mc) + guid_wbuf_bytes(batch, n_mc) + (((size_t)batch * sizeof(float) + 255) & ~(size_t)255);
}


// The step's scalars of the guidance block: Python-double arithmetic of the reference
// (sample_mnist_svhn.py:115,127,135,159,170), rounded to fp32 where a tensor op consumes it.
inline void guidance_scalars(double t, float* tf, float* s2, float* cden) {
  const double eps = 1e-3;
  const double sigma_t = 1.0 - t + eps;
  *tf = (float)t, *s2 = (float)(sigma_t * sigma_t), *cden = (float)(1.0 - t + eps);
}

inline int guidance_launch(const float* x, const float* y, float* vx, float* vy, const float* mx, const float* my,
                    const float* r, int B, int N, int dx, int dy, double t, double gamma, float* logp,
                    float* weights_out, float* xs, float* ys, float dt, hipStream_t s, const float* sched = nullptr,
                    const int* step_ptr = nullptr, int phase = 0) {
  // phase 0: the whole block; 1: distances + importance weights only -- they need the step's (x_t, y_t) and the MC set,
  // not the velocities; 2: the rest (needs v)
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
  if (phase < 2) {  // (timer class "guid_logp": the distances AND the importance weights made of them)
    ProfScope p(RGFM_KCLASS_GUID_LOGP, 4.0 * (B + N) * D + dist_b, s);
    launch_guid_logp(a, s);
    launch_guid_weights(a, s);
  }
  if (phase != 1) {
    ProfScope p(RGFM_KCLASS_GUID_APPLY, 4.0 * B * N + 4.0 * N * D + 4.0 * B * D * 3.0, s);
    launch_guid_apply(a, s);
  }
  return RGFM_OK;
}


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
    const double t = (double)(step_begin + i) * dtd;
    // The distances to the MC set and the importance weights depend on (x_t, y_t) only (sample_mnist_svhn.py:130-156; a
    // guided step's nets write velocities, the state moves in guid_apply): they go on this stream BEFORE it waits for the
    // other net -- the x net of a pair is the quicker one, so they run in its shadow instead of on the step's critical path
    if (guided && overlap) {
      rc = guidance_launch(x_inout, y_inout, vx, vy, mc_x1, mc_y1, mc_ratios, batch, n_mc, dx, dy, t, gamma, logp,
                           nullptr, x_inout, y_inout, dt, s, sched_dev, step_dev, 1);
      if (rc) return rc;
    }
    if (overlap) HIP_TRY(hipStreamWaitEvent(s, ev_join, 0));
    if (guided) {
      rc = guidance_launch(x_inout, y_inout, vx, vy, mc_x1, mc_y1, mc_ratios, batch, n_mc, dx, dy, t, gamma, logp,
                           nullptr, x_inout, y_inout, dt, s, sched_dev, step_dev, overlap ? 2 : 0);
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
  // (Measured and rejected in round 4, profiles/r04_ab_decoupled_loop.txt: the loop WITHOUT the per-step join -- each
  // modality a chain net -> its half of the guided update on its own stream, coupled only through two events per step
  // around the importance weights, the x net up to one evaluation ahead -- is bit-identical and 0.5 - 0.9 % SLOWER
  // (469 - 471 vs 473 paired images/s, same box, alternating): the join costs nothing the chains could use.)
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

