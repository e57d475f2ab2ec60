// Single-layer benchmark of the conv kernels (development tool; not part of the library).
//   conv_bench <S> <Cin> <Cout> [mode 0|1|2] [res 0|1|2] [B] [which: bx3|f32|hx2|hx2p|hx2q|hx2s|hx2c]
//   (mode 1: the stride-2 conv of a Downsample -- S is the OUTPUT size, the input is raw: hx2 or hx2s)
// Every run also checks the selected kernel against the exact-fp32 MFMA kernel on the same data.
// Builds one ConvArgs with random NHWC input / packed weights, launches it 20x, prints us and
// fp32-equivalent TFLOP/s; with -DRGFM_BX3_PROF also the per-phase cycle counts of the bx3w kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_bx3.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2.hip"
#ifdef RGFM_KB_SCAFFOLD
// round 3's timing ablations, phase stamps and rejected cuts (-DRGFM_HX2P_ABL=n, -DRGFM_HX2P_PROF, -DRGFM_HX2P_CHUNK_EXP,
// -DRGFM_HX2Q_ABL=n, -DRGFM_HX2Q_PROF, ...): frozen copies of the kernels as they were before round 4 took the scaffolding out
#include "variants/conv_mfma_hx2p_r03_scaffold.hip"
#include "variants/conv_mfma_hx2q_r03_scaffold.hip"
#else
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2p.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2q.hip"
#endif
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2s.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2c.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2d.hip"
#include "../../ratio_guided_multimodal_fm_amd/csrc/conv_mfma_hx2w.hip"
#ifdef RGFM_KB_HX2W4
#include "../experiments/conv_mfma_hx2w4.hip"  // the rejected four-wave cut (RGFM_HX2W_CUT=2)
#endif
#include "../../ratio_guided_multimodal_fm_amd/csrc/unet_kernels.hip"

using namespace rgfm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = scale * ((float)(s >> 8) / 8388608.0f - 1.0f);
  }
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

__global__ void pack_plain(const float* w, float* out, int Cout, int Cin, int taps, int nb) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = i % 16;
    size_t r = i / 16;
    const int n = r % nb;
    r /= nb;
    const int tap = r % taps;
    r /= taps;
    const int nch = Cin / 16;
    const int ch = r % nch;
    const int blk = r / nch;
    out[i] = w[((size_t)(blk * nb + n) * Cin + ch * 16 + kk) * taps + tap];
  }
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 32, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 64;
  const int mode = argc > 4 ? atoi(argv[4]) : 0, res = argc > 5 ? atoi(argv[5]) : 0, B = argc > 6 ? atoi(argv[6]) : 512;
  const bool f32 = argc > 7 && strcmp(argv[7], "f32") == 0;
  const bool hx2 = argc > 7 && strcmp(argv[7], "hx2") == 0;
  const bool hx2p = argc > 7 && strcmp(argv[7], "hx2p") == 0;
  const bool hx2q = argc > 7 && strcmp(argv[7], "hx2q") == 0;
  const bool hx2s = argc > 7 && strcmp(argv[7], "hx2s") == 0;
  const bool hx2c = argc > 7 && strcmp(argv[7], "hx2c") == 0;
  const bool hx2d = argc > 7 && strcmp(argv[7], "hx2d") == 0;
  const bool hx2w = argc > 7 && strcmp(argv[7], "hx2w") == 0;  // Winograd F(2x2, 3x3) on the two-plane arithmetic
  const int Sin = mode == CONV_UP2 ? S / 2 : (mode == CONV_S2 ? 2 * S : S);
  const int nt = Cout % 64 == 0 ? 2 : 1;
  CK(hipSetDevice(0));
  conv_mfma_init();
  conv_bx3_init();
  conv_hx2_init();
  conv_hx2p_init();
  conv_hx2q_init();
  conv_hx2s_init();
  conv_hx2c_init();
  conv_hx2d_init();
  if (getenv("RGFM_HX2D")) conv_hx2d_set(atoi(getenv("RGFM_HX2D")));  // 1: eight waves, 2: four waves x two workgroups per CU
  conv_hx2c_set_all(1);
  if (getenv("RGFM_HX2Q_MIN")) conv_hx2q_set_min(atoi(getenv("RGFM_HX2Q_MIN")));
  if (getenv("RGFM_HX2Q_TPW")) conv_hx2q_set_tpw(atoi(getenv("RGFM_HX2Q_TPW")));
  conv_hx2q_set_all(1);
  if (getenv("RGFM_HX2Q_CUT")) conv_hx2q_set_cut(atoi(getenv("RGFM_HX2Q_CUT")));
  if (getenv("RGFM_HX2P_W4")) conv_hx2p_set_w4(atoi(getenv("RGFM_HX2P_W4")));
  if (getenv("RGFM_HX2P_HALF")) conv_hx2p_set_half(atoi(getenv("RGFM_HX2P_HALF")));
#ifdef RGFM_KB_SCAFFOLD
  if (getenv("RGFM_HX2P_CHUNK")) conv_hx2p_set_chunk(atoi(getenv("RGFM_HX2P_CHUNK")));
#if RGFM_HX2P_QEXP
  if (getenv("RGFM_HX2P_Q")) conv_hx2p_set_q(atoi(getenv("RGFM_HX2P_Q")));
#endif
#endif

  ConvArgs a{};
  a.in0 = dev_rand((size_t)B * Sin * Sin * Cin, 1.f, 1);
  a.C0 = Cin, a.Hin = a.Win = Sin;
  // input GroupNorm: random partial statistics (mean, M2) of the input map, finalized into the scale/shift array
  // the table-path kernels read; the pipelined kernel derives the same numbers from the statistics itself
  const TileGeom gin = make_geom(Sin, Sin);
  float* gstats;
  {
    const size_t ns = (size_t)B * gin.nparts * Cin * 2;
    std::vector<float> hs(ns);
    unsigned sd = 777u;
    for (size_t i = 0; i < ns; i += 2) {
      sd = sd * 1664525u + 1013904223u;
      hs[i] = 0.3f * ((float)(sd >> 8) / 8388608.0f - 1.0f);
      sd = sd * 1664525u + 1013904223u;
      hs[i + 1] = 64.f * (0.2f + 0.2f * (float)(sd >> 8) / 16777216.0f);
    }
    hipMalloc(&gstats, ns * 4);
    hipMemcpy(gstats, hs.data(), ns * 4, hipMemcpyHostToDevice);
  }
  float* ggamma = dev_rand(Cin, 1.f, 21);
  float* gbeta = dev_rand(Cin, 0.3f, 22);
  float* abbuf;
  hipMalloc(&abbuf, (size_t)B * Cin * 2 * 4);
  {
    GnFinalizeArgs f{};
    f.stats0 = gstats, f.C0 = Cin, f.groups = 8, f.gamma = ggamma, f.beta = gbeta, f.ab = abbuf, f.B = B, f.g = gin;
    launch_gn_finalize(f, 0);
  }
  a.ab = abbuf;
  float* w = dev_rand((size_t)Cout * Cin * 9, 0.05f, 3);
  float* wp;
  hipMalloc(&wp, (size_t)Cout * Cin * 9 * 4);
  pack_plain<<<256, 256>>>(w, wp, Cout, Cin, 9, 32 * nt);
  void* w3;
  hipMalloc(&w3, (size_t)Cout * Cin * 9 * 6);
  launch_pack_conv_bx3(w, w3, Cout, Cin, 9, 0);
  a.wpk = wp, a.wpk3 = w3;
  void* wh;
  hipMalloc(&wh, (size_t)Cout * Cin * 9 * 4);
  float* hq;
  hipMalloc(&hq, 8 * sizeof(float));
  launch_pack_conv_hx2(w, wh, hq, Cout, Cin, 9, mode == CONV_S2 ? CONV_S2 : CONV_S1, 0);
  a.wpkh = wh, a.hq = hq;
  if (mode == CONV_S2) {  // the plain nine-tap image once more (conv_mfma_hx2s.hip)
    void* wh9;
    hipMalloc(&wh9, (size_t)Cout * Cin * 9 * 4);
    launch_pack_conv_hx2(w, wh9, hq, Cout, Cin, 9, CONV_S1, 0);
    a.wpkh9 = wh9;
    a.ab = nullptr;  // (raw input)
  }
  unsigned* flag;
  hipMalloc(&flag, 4);
  hipMemset(flag, 0, 4);
  a.range_flag = flag;
  a.bias = dev_rand(Cout, 0.1f, 4);
  a.temb = dev_rand(Cout, 0.1f, 5), a.temb_stride = Cout;
  if (mode == CONV_S2) a.temb = nullptr;  // (a Downsample has no time term)
  a.res_mode = res;
  if (getenv("RGFM_KB_SC")) a.small_check = atoi(getenv("RGFM_KB_SC"));  // the output's low-range check (ConvArgs::small_check)
  int skipk = 0;
  if (res == 1) a.res0 = dev_rand((size_t)B * S * S * Cout, 1.f, 6), a.R0 = Cout;
  if (res == 2) {
    const int R = getenv("RGFM_KB_R") ? atoi(getenv("RGFM_KB_R")) : Cin;  // 1x1 skip from an R-channel source at output resolution
    a.res0 = dev_rand((size_t)B * S * S * R, 1.f, 6), a.R0 = R;
    float* ws = dev_rand((size_t)Cout * R, 0.05f, 7);
    float* wsp;
    hipMalloc(&wsp, (size_t)Cout * R * 4);
    pack_plain<<<64, 256>>>(ws, wsp, Cout, R, 1, 32 * nt);
    void* ws3;
    hipMalloc(&ws3, (size_t)Cout * R * 6);
    launch_pack_conv_bx3(ws, ws3, Cout, R, 1, 0);
    a.wskip = wsp, a.wskip3 = ws3, a.skip_bias = dev_rand(Cout, 0.1f, 8);
    void* wsh;
    hipMalloc(&wsh, (size_t)Cout * R * 4);
    launch_pack_conv_hx2(ws, wsh, hq + 4, Cout, R, 1, CONV_S1, 0);
    a.wskiph = wsh, a.hq_skip = hq + 4;
    skipk = R;
  }
  float* out;
  hipMalloc(&out, (size_t)B * S * S * Cout * 4);
  a.out = out;
  a.g = make_geom(S, S);
  float* st;
  hipMalloc(&st, (size_t)B * a.g.nparts * Cout * 2 * 4);
  a.stats_out = st;
  a.B = B, a.Cout = Cout;
  a.halo_px = a.g.spt * (a.g.th + 2) * (a.g.W + 2);
  const double flops = 2.0 * B * S * S * (double)Cout * (9 * Cin + skipk);
  ConvArgs ap = a;  // the pipelined kernel takes the norm itself
  ap.ab = nullptr, ap.gn_stats0 = gstats, ap.gn_gamma = ggamma, ap.gn_beta = gbeta, ap.gn_nparts0 = gin.nparts, ap.gn_g = gin;
  ConvArgs ad = a;  // conv_mfma_hx2d_kernel: the same input, normalised + activated + split beforehand (P format)
  if (hx2d) {
    void* pbuf;
    hipMalloc(&pbuf, (size_t)B * Sin * Sin * Cin * 4);
    launch_hx_presplit(a.in0, abbuf, pbuf, B, Sin * Sin, Cin, 0);
    void* zer;
    hipMalloc(&zer, 256);
    hipMemset(zer, 0, 256);
    ad.ab = nullptr, ad.pin0 = pbuf, ad.zeros = zer;
    if (!conv_hx2d_supported(ad, mode)) { printf("hx2d: unsupported shape\n"); return 1; }
  }
  if (getenv("RGFM_KB_POUT")) {  // the producing side of the P-format hand-over: 1 = beside the fp32 map, 2 = instead of it
    void* pob;
    hipMalloc(&pob, (size_t)B * S * S * Cout * 4);
    ap.pout = pob, ap.pn_gamma = dev_rand(Cout, 1.f, 31), ap.pn_beta = dev_rand(Cout, 0.3f, 32);
  }
  ConvArgs aw = a;  // conv_mfma_hx2w_kernel: its own packed (transformed) weights and scale record
  if (hx2w) {
    void* whw;
    float *hqw, *tmpw;
    hipMalloc(&whw, (size_t)Cout * Cin * 16 * 4);
    hipMalloc(&hqw, 8 * sizeof(float));
    hipMalloc(&tmpw, (size_t)Cout * Cin * 16 * 4);
    launch_pack_conv_hx2w(w, whw, hqw, tmpw, Cout, Cin, 0);
    aw.wpkw = whw, aw.hqw = hqw;
    if (getenv("RGFM_KB_GN")) aw = ap, aw.wpkw = whw, aw.hqw = hqw;  // the consumer-side norm (statistics) instead of the array
    CK(hipDeviceSynchronize());
    if (conv_hx2w_init() != 0 || !conv_hx2w_supported(aw, mode)) { printf("hx2w: unsupported shape\n"); return 1; }
#ifdef RGFM_KB_HX2W4
    conv_hx2w4_init();
#endif
  }
  if (hx2p && !conv_hx2p_supported(ap, mode)) { printf("hx2p: unsupported shape\n"); return 1; }
  if (hx2q && !conv_hx2q_supported(ap, mode)) { printf("hx2q: unsupported shape\n"); return 1; }
  if (hx2s && !conv_hx2s_supported(a, mode)) { printf("hx2s: unsupported shape\n"); return 1; }
  if (hx2c && !conv_hx2c_supported(ap, mode)) { printf("hx2c: unsupported shape\n"); return 1; }
  auto launch = [&]() {
    if (f32) launch_conv_mfma(a, mode, 0);
    else if (hx2) launch_conv_hx2(a, mode, 0);
    else if (hx2p) launch_conv_hx2p(ap, mode, 0);
    else if (hx2q) launch_conv_hx2q(ap, mode, 0);
    else if (hx2s) launch_conv_hx2s(a, 0);
    else if (hx2c) launch_conv_hx2c(ap, 0);
    else if (hx2d) launch_conv_hx2d(ad, 0);
#ifdef RGFM_KB_HX2W4
    else if (hx2w && getenv("RGFM_HX2W_CUT") && atoi(getenv("RGFM_HX2W_CUT")) == 2) launch_conv_hx2w4(aw, 0);
#endif
    else if (hx2w) launch_conv_hx2w(aw, 0);
    else launch_conv_bx3(a, mode, 0);
  };
  {  // reference: the exact-fp32 MFMA kernel on the same data
    const size_t no = (size_t)B * S * S * Cout, ns = (size_t)B * a.g.nparts * Cout * 2;
    std::vector<float> ref(no), got(no), sref(ns), sgot(ns);
    if (mode == CONV_S2) launch_conv_hx2(a, mode, 0);  // (stride 2: the reference is conv_mfma_hx2_kernel, itself pinned by the library's parity tests)
    else launch_conv_mfma(a, mode, 0);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    hipMemcpy(ref.data(), out, no * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sref.data(), st, ns * 4, hipMemcpyDeviceToHost);
    hipMemset(out, 0, no * 4);
    launch();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    hipMemcpy(got.data(), out, no * 4, hipMemcpyDeviceToHost);
    hipMemcpy(sgot.data(), st, ns * 4, hipMemcpyDeviceToHost);
    double emax = 0, vmax = 0, smax = 0;
    for (size_t i = 0; i < no; ++i) {
      const double d = fabs((double)got[i] - ref[i]);
      if (!(d <= emax)) emax = d;
      if (fabs(ref[i]) > vmax) vmax = fabs(ref[i]);
    }
    for (size_t i = 0; i < ns; ++i) {
      const double d = fabs((double)sgot[i] - sref[i]) / (1.0 + fabs(sref[i]));
      if (!(d <= smax)) smax = d;
    }
    unsigned fl = 0;
    hipMemcpy(&fl, flag, 4, hipMemcpyDeviceToHost);
    printf("check vs f32 kernel: max|diff| %.3e (max|ref| %.3f), stats rel diff %.3e, range flag %u\n", emax, vmax, smax, fl);
  }
#ifdef RGFM_HX2Q_PROF
  {
    unsigned long long zq[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_hx2q_prof), zq, sizeof(zq));
  }
#endif
#ifdef RGFM_HX2P_PROF
  unsigned long long zerop[34] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_hx2p_prof), zerop, sizeof(zerop));
#endif
#if defined(RGFM_BX3_PROF) || defined(RGFM_HX2_PROF)
  unsigned long long zero[10] = {0};
#ifdef RGFM_HX2_PROF
  hipMemcpyToSymbol(HIP_SYMBOL(g_hx2_prof), zero, sizeof(zero));
#else
  hipMemcpyToSymbol(HIP_SYMBOL(g_bx3_prof), zero, sizeof(zero));
#endif
#endif
  if (getenv("RGFM_KB_POUT") && atoi(getenv("RGFM_KB_POUT")) == 2) ap.out = nullptr, ap.stats_out = nullptr;  // (timing only)
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  CK(hipEventSynchronize(e1));
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%s S=%d Cin=%d Cout=%d mode=%d res=%d B=%d: %8.1f us  %6.1f TFLOP/s (fp32-equivalent)\n", f32 ? "f32" : (hx2 ? "hx2" : (hx2p ? "hx2p" : (hx2q ? "hx2q" : (hx2s ? "hx2s" : (hx2c ? "hx2c" : (hx2d ? "hx2d" : (hx2w ? "hx2w" : "bx3"))))))), S, Cin,
         Cout, mode, res, B, us, flops / us / 1e6);
#ifdef RGFM_HX2Q_PROF
  if (hx2q) {
    unsigned long long pq[16];
    hipMemcpyFromSymbol(pq, HIP_SYMBOL(g_hx2q_prof), sizeof(pq));
    const char* nm[7] = {"gn table", "decode+barrier", "fill", "acc init", "K loops", "epilogues", "whole workgroup"};
    for (int g = 0; g < 2; ++g) {
      printf("  wave %d (s_memtime ticks per workgroup, all its tiles):", g * 4);
      for (int i = 0; i < 7; ++i) printf(" %s %.0f |", nm[i], (double)pq[g * 8 + i] / (double)pq[g * 8 + 7]);
      printf("\n");
    }
  }
#endif
#ifdef RGFM_HX2P_PROF
  {
    unsigned long long pp[34];
    hipMemcpyFromSymbol(pp, HIP_SYMBOL(g_hx2p_prof), sizeof(pp));
    const double nbk = (double)pp[33];
    printf("  wave   pro+fill    stage     mfma  barrier  (clk/block)\n");
    for (int w = 0; w < 8; ++w)
      printf("  w%d   %9.0f %8.0f %8.0f %8.0f\n", w, pp[w * 4] / nbk, pp[w * 4 + 1] / nbk, pp[w * 4 + 2] / nbk, pp[w * 4 + 3] / nbk);
    printf("  w0 epilogue %.0f\n", pp[32] / nbk);
    static long long tr[2][64][9];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_hx2p_trace), sizeof(tr));
    const long long t0 = tr[1][4][2];
    printf("  trace of block 3 (cycles since unit 4's start): unit | w0: stage[b,e] mfma[b,e] barrier[b,e] | w4: mfma[b,e] stage[b,e] barrier[b,e]\n");
    for (int u = 4; u < 14; ++u)
      printf("  u%02d | %6lld %6lld  %6lld %6lld  %6lld %6lld | %6lld %6lld  %6lld %6lld  %6lld %6lld\n", u, tr[0][u][0] - t0, tr[0][u][1] - t0,
             tr[0][u][2] - t0, tr[0][u][3] - t0, tr[0][u][4] - t0, tr[0][u][5] - t0, tr[1][u][2] - t0, tr[1][u][3] - t0, tr[1][u][0] - t0,
             tr[1][u][1] - t0, tr[1][u][4] - t0, tr[1][u][5] - t0);
    printf("  prologue of block 3 (index + acc init | item decode | GN table | barrier | fill): w0 %lld %lld %lld %lld %lld | w4 %lld %lld %lld %lld %lld\n",
           tr[0][63][0], tr[0][63][1], tr[0][63][2], tr[0][63][3], tr[0][63][4], tr[1][63][0], tr[1][63][1], tr[1][63][2], tr[1][63][3], tr[1][63][4]);
    printf("  acc-init section: index math w0 %lld w4 %lld | bias/temb w0 %lld w4 %lld\n", tr[0][63][5], tr[1][63][5], tr[0][63][6], tr[1][63][6]);
    printf("  stage split (commit_b | commit_a | issue_b | issue_a):\n");
    for (int u = 4; u < 14; ++u)
      printf("  u%02d | w0 %5lld %5lld %5lld %5lld | w4 %5lld %5lld %5lld %5lld\n", u, tr[0][u][6] - tr[0][u][0], tr[0][u][7] - tr[0][u][6],
             tr[0][u][8] - tr[0][u][7], tr[0][u][1] - tr[0][u][8], tr[1][u][6] - tr[1][u][0], tr[1][u][7] - tr[1][u][6],
             tr[1][u][8] - tr[1][u][7], tr[1][u][1] - tr[1][u][8]);
  }
#endif
#if defined(RGFM_BX3_PROF) || defined(RGFM_HX2_PROF)
  unsigned long long p[10];
#ifdef RGFM_HX2_PROF
  hipMemcpyFromSymbol(p, HIP_SYMBOL(g_hx2_prof), sizeof(p));
#else
  hipMemcpyFromSymbol(p, HIP_SYMBOL(g_bx3_prof), sizeof(p));
#endif
  const double nb = (double)p[7];
  const char* names[7] = {"prologue", "issue", "mfma", "commit-wait", "commit-A", "commit-B", "epilogue"};
  double tot = 0;
  for (int i = 0; i < 7; ++i) tot += (double)p[i];
  for (int i = 0; i < 7; ++i) printf("  %-12s %10.0f clk/block  %5.1f%%\n", names[i], p[i] / nb, 100.0 * p[i] / tot);
  printf("  total        %10.0f clk/block over %.0f block-launches; shader clock during the blocks: %.0f MHz\n", tot / nb, nb,
         100.0 * (double)p[8] / (double)p[9]);
#endif
  return 0;
}
