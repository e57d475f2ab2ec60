// rgfm_kernels.h -- internal interface between the C-ABI host code (rgfm_host.h, api_*.cpp)
// and the gfx950 kernels.  Not part of the public ABI (that is include/rgfm.h).
//
// HBM data layout (DESIGN.md "Data layout"):
//   * boundary tensors (x_t, v, MC set): NCHW fp32, as the reference passes them;
//   * every internal activation: NHWC fp32  [B][H][W][C]  (channels contiguous:
//     the implicit-GEMM K axis), produced once and consumed in place -- channel
//     concat, nearest-upsample, GroupNorm-apply and SiLU are folded into the
//     consumer's load path, never materialised;
//   * GroupNorm statistics: per (sample, 64-pixel wave segment, channel) partial
//     (mean, M2) pairs written by the producer's epilogue:
//     stats[B][nparts][C][2]; combined (Chan) by gn_finalize into per
//     (sample, channel) scale/shift  ab[B][C][2].
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rgfm {

constexpr int KC = 16;   // input channels per K-chunk of the MFMA conv
constexpr int LDP = 20;  // LDS row length (floats) of a 16-channel record: 16 + 4 pad,
                         // conflict-free for ds_read_b128 (stride 80 B)

// Output tiling of a [B, H, W] raster into 256-pixel workgroup tiles made of
// four 64-pixel wave segments.
struct TileGeom {
  int H, W, HW;
  int spt;     // samples per tile: 4 when HW <= 64 (one sample per wave segment), else 1
  int th;      // output rows per tile (spt == 1)
  int tps;     // tiles per sample (spt == 1)
  int nparts;  // statistics parts per sample
};

__host__ __device__ inline TileGeom make_geom(int H, int W) {
  TileGeom g;
  g.H = H;
  g.W = W;
  g.HW = H * W;
  if (g.HW <= 64) {
    g.spt = 4;
    g.th = H;
    g.tps = 1;
    g.nparts = 1;
  } else {
    g.spt = 1;
    const int thmax = 256 / W;
    const int nt = (H + thmax - 1) / thmax;
    g.th = (H + nt - 1) / nt;
    g.tps = nt;
    g.nparts = nt * 4;
  }
  return g;
}

__host__ __device__ inline int geom_num_tiles(const TileGeom& g, int B) {
  return g.spt == 1 ? B * g.tps : (B + g.spt - 1) / g.spt;
}

// valid pixels of statistics part `part` of a sample
__host__ __device__ inline int geom_part_count(const TileGeom& g, int part) {
  if (g.spt != 1) return g.HW;
  const int tile = part >> 2, w = part & 3;
  int rows = g.H - tile * g.th;
  if (rows > g.th) rows = g.th;
  int n = rows * g.W - 64 * w;
  return n < 0 ? 0 : (n > 64 ? 64 : n);
}

// CONV_T2: ConvTranspose2d(k=4, s=2, p=1) as four 2x2-tap parity-class convs over the INPUT raster
// (ConvArgs::g describes that raster; the output is [B][2H][2W][Cout]); weights from launch_pack_deconv.
enum ConvMode { CONV_S1 = 0, CONV_S2 = 1, CONV_UP2 = 2, CONV_T2 = 3 };

struct ConvArgs {
  // input: up to two NHWC sources concatenated along C (torch.cat([h, skip], 1))
  const float* in0;
  const float* in1;
  int C0, C1;        // channels of in0 / in1 (C1 = 0: single source)
  int Hin, Win;      // spatial size of the sources
  const float* ab;   // [B][C0+C1][2] GroupNorm scale/shift (then SiLU) applied on load; null = raw
  // Consumer-side GroupNorm (conv_mfma_bx3.hip): instead of `ab`, the partial statistics of the input map(s)
  // themselves -- every workgroup derives the scale/shift of its own sample(s) in its prologue, so no finalize
  // launch (or producer-side finalize) sits between two convs.
  const float* gn_stats0;  // [B][gn_nparts0][C0][2] statistics of in0 (null: use `ab`)
  const float* gn_stats1;  // [B][gn_g.nparts][C1][2] statistics of in1 or null
  const float* gn_gamma;   // [C0 + C1]
  const float* gn_beta;
  int gn_nparts0;          // parts of in0: gn_g.nparts, or 4 x that when in0 was written by a CONV_T2 launch
  TileGeom gn_g;           // tiling of the raster the parts refer to (part sizes)
  // conv_mfma_hx2d.hip: the input ALREADY normalised, activated and split ("P format": [pixel][C0/16][h ch 0-7 | h ch
  // 8-15 | l ch 0-7 | l ch 8-15] fp16, 64 B per (pixel, 16-channel chunk), written by the producing conv's epilogue or
  // by launch_hx_presplit) -- staged by LDS-DMA only; `zeros` = 64 zero bytes on the device (the padding records' source)
  // raw inputs far from unit scale (the gradients of the reverse pass, 1e-3 ... 1e-6): when set, *in_amax holds the bits
  // of max |input| over the tensor (its producer's launch_grad_act wrote them) and the kernel stages 2^-e x the values
  // (max = f 2^e, f in [0.5, 1)) and multiplies its outputs by 2^e -- the two-plane representation then keeps its 22 bits
  // below the tensor's maximum.  conv_mfma_hx2_kernel (launch_conv_hx2), raw inputs without residual / skip only.
  const unsigned* in_amax;
  const void* pin0;
  const void* zeros;
  // ... and the producing side: when set, the epilogue ALSO (out != null) or ONLY (out == null) writes the output in P
  // format for the ONE norm that consumes it (8 groups over Cout channels, parameters pn_gamma / pn_beta [Cout]):
  // kernels whose workgroups own whole (sample, group) sets (rgfm_host.h: p_producer_ok)
  void* pout;
  const float* pn_gamma;
  const float* pn_beta;
  const float* wpk;  // packed 3x3 weights  [Cout/(32NT)][Cin/16][9][32NT][16]
  const void* wpk3;  // the same weights as three bf16 planes [..][9][32NT][3][16] (conv_mfma_bx3.hip) or null
  const void* wskip3;
  // conv_mfma_hx2.hip: the same weights as two scaled fp16 planes [..][9][32NT][2][16] + their scale record
  // hq = {q = S_A s_w, 1 / q, s_w, eligible} (launch_pack_conv_hx2), and the device word the kernel ORs 1 into
  // when a staged activation leaves the fp16 range (the host then repeats the call on the split-bf16 kernel)
  const void* wpkh;
  const void* wskiph;
  const void* wpkh9;  // stride-2 convs: the weights once more as the plain nine-tap image (conv_mfma_hx2s.hip), or null
  const float* hq;
  const float* hq_skip;
  unsigned* range_flag;
  // 1: a later conv stages this output RAW (no GroupNorm: the fused 1x1 skip, Downsample, Upsample -- reference
  // unet_flexible.py:85,96,107-108), i.e. as S_A x value in two fp16 planes.  The epilogue then checks the low side of
  // that representation: a wave block (64 pixels x its channels) whose largest |value| is below HX_SMALL (and not 0)
  // ORs 2 into *range_flag, and the caller repeats the call on the split-bf16 kernels (fp32 exponent range).
  int small_check;
  const float* bias; // [Cout]
  const float* temb; // time-embedding add: temb[(per_row ? b : 0) * temb_stride + c]; null = none
  int temb_stride;
  int temb_per_row;
  const int* step_ptr;  // optional device word: the row of the time table is *step_ptr (+ b when temb_per_row): lets
                        // one captured launch sequence (hipGraph) serve every Euler step
  // residual: res_mode 0 none; 1 identity (res0 NHWC [.., Cout]); 2 fused 1x1 conv over cat(res0, res1)
  int res_mode;
  const float* res0;
  const float* res1;
  int R0, R1;
  const float* wskip;      // packed 1x1 weights [Cout/(32NT)][(R0+R1)/16][1][32NT][16]
  const float* skip_bias;  // [Cout]
  // epilogue activation (ratio-net BatchNorm folded to scale/shift, then SiLU); null = none
  const float* ep_scale;
  const float* ep_shift;
  int ep_nosilu;     // 1: store ep_scale * acc + ep_shift itself (the gradient path keeps the pre-activation)
  float* out;        // NHWC [B][H][W][Cout]
  float* stats_out;  // [B][nparts][Cout][2] or null
  // Producer-side GroupNorm finalize (conv_mfma_bx3.hip): when fin_ab is set, the LAST wave to deliver
  // statistics of a sample (arrival counter) computes the scale/shift of the norm that consumes this
  // output -- cat(this output, fin_stats1's tensor) -- instead of a separate gn_finalize launch.
  float* fin_ab;             // [B][Cout + fin_C1][2] or null
  unsigned* fin_counter;     // [B], zero between launches (the finishing wave resets its entry)
  int fin_expected;          // arrivals per sample
  const float* fin_stats1;   // partner statistics [B][nparts][fin_C1][2] (same spatial size as the output) or null
  int fin_C1;
  const float* fin_gamma;    // [Cout + fin_C1]
  const float* fin_beta;
  // Winograd F(2x2, 3x3) image of the same 3x3 weights (conv_mfma_hx2w.hip: launch_pack_conv_hx2w) and its scale record, or null
  const void* wpkw;
  const float* hqw;
  int B, Cout;
  TileGeom g;        // OUTPUT raster tiling
  int halo_px;       // pixels of the staged input tile
};

struct ConvInArgs {  // first conv of a net: NCHW image -> NHWC features
  const float* x;    // [B][CIN][H][W]
  const float* w;    // [C0][CIN][3][3] (reference layout)
  const float* bias;
  const float* ep_scale;  // optional BatchNorm scale/shift + SiLU epilogue (ratio nets)
  const float* ep_shift;
  int ep_nosilu;          // 1: no SiLU (see ConvArgs::ep_nosilu)
  float* out;        // NHWC [B][H][W][C0]
  float* stats_out;  // or null
  int B, C0;
  TileGeom g;
  unsigned* range_flag;  // with small_check (see ConvArgs::small_check): the first conv's output feeds raw consumers too
  int small_check;
};

struct ConvOutArgs {  // out_conv(silu(out_norm(h))) -> NCHW velocity, optional fused Euler
  const float* in;    // NHWC [B][H][W][Cin]
  const float* ab;    // [B][Cin][2]
  const float* w;     // [Cin/16][9][CIMG][16] (launch_pack_conv_out)
  const float* bias;
  float* v_out;       // NCHW velocity or null
  float* x_state;     // NCHW state for the fused Euler update or null
  float dt;
  int B, Cin;
  TileGeom g;
  int halo_px;
};

struct GnFinalizeArgs {
  const float* stats0;  // [B][nparts][C0][2]
  const float* stats1;  // [B][nparts][C1][2] or null
  int C0, C1, groups;
  const float* gamma;   // [C0+C1]
  const float* beta;
  float* ab;            // [B][C0+C1][2]
  int B;
  TileGeom g;
  int rep;              // 0/1: stats0 has g.nparts parts; 4: output of a CONV_T2 launch (4 x g.nparts parts of a 4*HW-pixel map)
  float* mr;            // optional [B][groups][2]: (mean, rstd) of every group (the norm's backward: launch_gn_bwd)
};

struct TimeLinear {  // one ResBlock time_mlp Linear: rows [out_off, out_off+cout) of the table
  int w_off, b_off, cout, out_off;
};

struct TimeEmbedArgs {
  const float* params;     // state_dict-order blob (device copy)
  const float* freqs;      // [mc/2]
  int mc, temb;
  int te0w, te0b, te2w, te2b;  // offsets into params
  const TimeLinear* lin;   // device array
  int nlin;
  int total;               // table row length
  // time values: either explicit t_dev[nt] or step index based: t = (float)((step_begin + i) * (1.0/num_steps))
  const float* t_dev;
  int num_steps, step_begin;
  float* table;            // [nt][total]
  float* emb_out;          // optional [nt][mc]: the sinusoidal embedding itself (parity hook)
};

void launch_conv_mfma(const ConvArgs& a, int mode, hipStream_t s);
size_t conv_mfma_lds_bytes(const ConvArgs& a);
int conv_mfma_init();  // raises the dynamic-LDS limit of every instantiation

// fp32 conv with operands split into three bf16 planes, on the bf16 matrix cores (conv_mfma_bx3.hip)
bool conv_bx3_supported(const ConvArgs& a, int mode);
bool conv_bx3_gn_supported(const ConvArgs& a, int mode);  // a.gn_* filled: can the kernel take the norm itself?
int conv_fin_expected(const ConvArgs& a, int mode);  // arrivals per sample for ConvArgs::fin_expected (all conv_mfma* kernels)
int conv_bx3_init();
void launch_conv_bx3(const ConvArgs& a, int mode, hipStream_t s);
void launch_pack_conv_bx3(const float* w, void* out, int Cout, int Cin, int taps, hipStream_t s);
void launch_pack_deconv_bx3(const float* w, void* out, int Cin, int Cout, hipStream_t s);
void launch_pack_conv_bx3_s2(const float* w, void* out, int Cout, int Cin, hipStream_t s);  // weights of a stride-2 3x3 conv

// fp32 conv with operands split into two scaled fp16 planes, on the f16 matrix cores (conv_mfma_hx2.hip)
bool conv_hx2_supported(const ConvArgs& a, int mode);
bool conv_hx2_gn_supported(const ConvArgs& a, int mode);
int conv_hx2_init();
void launch_conv_hx2(const ConvArgs& a, int mode, hipStream_t s);
// producer / consumer pipelined version for CONV_S1 / CONV_UP2 (conv_mfma_hx2p.hip); same arguments
bool conv_hx2p_supported(const ConvArgs& a, int mode);
int conv_hx2p_init();
void conv_hx2p_set_half(int v);  // launches with fewer workgroups than this are cut finer (0: never; the CU count)
void conv_hx2p_set_w4(int v);  // tools/kbench A/B: 1 / 2 = four-wave workgroups forced, see conv_mfma_hx2p.hip
void launch_conv_hx2p(const ConvArgs& a, int mode, hipStream_t s);
// four-waves-per-SIMD version (conv_mfma_hx2q.hip: one tile x 64 channels at a time, two workgroups per CU, several
// tiles per workgroup behind one continuous staging stream) for full stride-1 launches over 16- / 32-pixel-wide
// rasters with Cout % 64 == 0 and a consumer-side input norm; bit-identical results
bool conv_hx2q_supported(const ConvArgs& a, int mode);
int conv_hx2q_init();
// conv_mfma_hx2s.hip: the Downsample convs (stride 2, raw input) with a chunk's four parity planes staged together
bool conv_hx2s_supported(const ConvArgs& a, int mode);
int conv_hx2s_init();
void conv_hx2s_set(int on);
void launch_conv_hx2s(const ConvArgs& a, hipStream_t s);
// conv_mfma_hx2c.hip: the stride-1 convs of the 8x8 level, one barrier per chunk (cut for the workgroup's serial chain)
bool conv_hx2c_supported(const ConvArgs& a, int mode);
int conv_hx2c_init();
void conv_hx2c_set(int on);
void conv_hx2c_set_all(int on);
void launch_conv_hx2c(const ConvArgs& a, hipStream_t s);
// conv_mfma_hx2d.hip: stride-1 convs of the 16x16 / 8x8 levels over a P-format input (ConvArgs::pin0), staged by LDS-DMA only
bool conv_hx2d_supported(const ConvArgs& a, int mode);
int conv_hx2d_init();
void conv_hx2d_set(int on);
void launch_conv_hx2d(const ConvArgs& a, hipStream_t s);
// P format of silu(scale x + shift) from an fp32 NHWC map and per-(sample, channel) pairs ab[B][C][2]
void launch_hx_presplit(const float* in, const float* ab, void* pout, int B, int HW, int C, hipStream_t s);
void conv_hx2q_set_min(int v);  // launches with fewer workgroups than this stay on conv_mfma_hx2p_kernel (0: never used)
void conv_hx2q_set_target(int v);  // workgroups a launch is cut into when it has the tiles (two per CU)
void conv_hx2q_set_tpw(int v);     // tools/kbench: force the tiles per workgroup
void conv_hx2q_set_cut(int v);     // tools/kbench: force the workgroup cut (10 NG + NT: 11, 21, 12, 22)
void conv_hx2q_set_all(int v);     // tools/kbench: 1 = every supported shape, not only those where it is the faster kernel
void launch_conv_hx2q(const ConvArgs& a, int mode, hipStream_t s);
// w [Cout][Cin][3][3] of "nearest-upsample x 2, then 3x3 conv" -> the ConvTranspose2d(4, 2, 1) weight K [Cin][Cout][4][4] of
// the same linear map: K[ky][kx] = sum of w[i][j] over i in A(ky), j in A(kx), A = {2}, {1, 2}, {0, 1}, {0} (unet_kernels.hip)
void launch_up2_as_deconv(const float* w, float* k, int Cout, int Cin, hipStream_t s);
// packs w (mode CONV_S1: [Cout][Cin][taps]; CONV_S2: the phase-major stride-2 order; CONV_T2: a ConvTranspose2d
// weight [Cin][Cout][4][4], taps ignored) and writes the scale record hq[4] (device)
void launch_pack_conv_hx2(const float* w, void* out, float* hq, int Cout, int Cin, int taps, int mode, hipStream_t s);

// Winograd F(2x2, 3x3) form of the stride-1 3x3 conv on the two-plane arithmetic (conv_mfma_hx2w.hip); its own packed weights
void hx2_scale_launch(const float* w, size_t n, float* hq, hipStream_t s);
void launch_pack_conv_hx2w(const float* w, void* out, float* hq, float* tmp, int Cout, int Cin, hipStream_t s);
bool conv_hx2w_supported(const ConvArgs& a, int mode);
int conv_hx2w_init();
void launch_conv_hx2w(const ConvArgs& a, hipStream_t s);

void launch_conv_in(const ConvInArgs& a, int cin, hipStream_t s);
void launch_conv_out(const ConvOutArgs& a, int cimg, hipStream_t s);
void launch_pack_conv_out(const float* w, float* out, int cimg, int Cin, hipStream_t s);
void launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s);
void launch_time_embed(const TimeEmbedArgs& a, int nt, hipStream_t s);
void launch_pack_conv(const float* w, float* out, int Cout, int Cin, int taps, int nt32, hipStream_t s);
// ConvTranspose2d weight [Cin][Cout][4][4] -> [4 parity][Cout/(32 nt32)][Cin/16][4 taps][32 nt32][16]
void launch_pack_deconv(const float* w, float* out, int Cin, int Cout, int nt32, hipStream_t s);
// out[o][p*C + c] = w[o][c*P + p]  (Linear consuming an NCHW-flattened map, re-indexed for NHWC)
void launch_permute_cols(const float* w, float* out, int rows, int C, int P, hipStream_t s);
// out[(p*C + c)][k] = w[(c*P + p)][k], bias likewise  (Linear producing an NCHW-flattened map)
void launch_permute_rows(const float* w, const float* b, float* wout, float* bout, int C, int P, int K, hipStream_t s);
// SinusoidalPositionEmbeddings (flow_matching.py:17-31) into out[b*stride + col0 + 0..dim)
void launch_fm_time_embed(const float* t_dev, int t_count, int num_steps, int step, const float* freqs, float* out,
                          int B, int dim, int stride, int col0, hipStream_t s);
void launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t s);
// ORs 2 into *flag when a (sample, 32-channel) block of the NHWC map is non-zero but below HX_SMALL (see ConvArgs::small_check)
void launch_range_low_check(const float* in, int B, int HW, int C, unsigned* flag, hipStream_t s);

// ---- ratio-estimator helpers
void launch_pool2(const float* in, const float* ab, float* out, int B, int H, int W, int C, hipStream_t s);
void launch_avgpool(const float* in, const float* ab, float* out, int B, int HW, int C, hipStream_t s);
void launch_linear_mfma(const float* x, const float* w, const float* b, float* y, int rows, int in,
                        int out, int x_stride, int y_stride, hipStream_t s);
// split-K variant with GroupNorm-apply + SiLU on the x load (x is an NHWC map flattened per row,
// ab[row][k % C][2]); `part` holds splits x rows x out floats
void launch_linear_mfma_splitk(const float* x, const float* ab, int C, const float* w, const float* b, float* y,
                               float* part, int splits, int rows, int in, int out, int y_stride, hipStream_t s);
void launch_layernorm_silu(float* x, const float* w, const float* b, int rows, int n, hipStream_t s);
void launch_ratio_head(const float* x, const float* w, const float* b, float* out, int rows, int n,
                       int loss, int what, hipStream_t s);
void launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                    float* shift, int C, hipStream_t s);

// ---- gradient of log r (ratio_grad.hip)
// [Co][Ci][9] -> [Ci][Co][9] with the taps flipped: the weights of the conv that maps dL/d(out) to dL/d(in)
void launch_conv_weight_transpose(const float* w, float* wt, int Co, int Ci, hipStream_t s);
void launch_transpose2d(const float* w, float* wt, int rows, int cols, hipStream_t s);  // [rows][cols] -> [cols][rows]
void launch_fill(float* p, float v, size_t n, hipStream_t s);
void launch_fill_ab_identity(float* ab, size_t n_pairs, hipStream_t s);  // (scale, shift) = (1, 0)
// gz[b,y,x,c] = route * silu'(z) * scale[c].  mode 0: g has z's shape; 1: g is the 2x2-max-pooled map's gradient
// (routed to the first maximum of silu(z) in each window); 2: g is [B][C], the gradient of the global average of
// silu(z); 3: g is [B][C], the gradient of the global average of the 2x2-max-pooled map (modes 2 then 1 in one)
// amax (optional): device word that receives (atomic max) the bits of the largest |gz| -- zero it before the launch
void launch_grad_act(const float* g, const float* z, const float* scale, float* gz, int B, int S, int C, int mode, hipStream_t s,
                     unsigned* amax = nullptr);
// the GroupNorm encoders (RatioEstimator, 28x28): activation backward with per-sample scale/shift pairs, then the norm's own
void launch_grad_act_gn(const float* g, const float* z, const float* ab, float* gu, int B, int S, int C, int mode, hipStream_t s);
void launch_gn_bwd(float* gu, const float* z, const float* gamma, const float* mr, int B, int HW, int C, int groups, hipStream_t s);
// first conv of an encoder, input gradient as an NCHW image: gimg[b,c,y,x] = sum_co,k w[co,c,k] gz[b, y-ky+1, x-kx+1, co]
void launch_conv_bwd_img(const float* gz, const float* w, float* gimg, int B, int S, int Co, int cimg, hipStream_t s);
// y = silu(LayerNorm(u)): gu from gy (wave per row); u is the Linear output kept by the forward
void launch_layernorm_silu_bwd(const float* u, const float* gy, const float* w, const float* b, float* gu, int rows, int n, hipStream_t s);
// gh[row][k] = d log_ratio / d score (score[row]) * w[k]; log_ratio (optional) [rows]
void launch_ratio_head_bwd(const float* score, const float* w, float* gh, float* log_ratio, int rows, int n, int loss, hipStream_t s);
// x <- x + (v + gamma g) dt
void launch_euler_grad(float* x, const float* v, const float* g, size_t n, float gamma, float dt, hipStream_t s);

// ---- guidance / Euler
struct GuidanceArgs {
  const float* x;  // [B][dx]
  const float* y;  // [B][dy]
  float* vx;       // in: model velocity, out: blended velocity (when x_state null)
  float* vy;
  const float* mc_x1;
  const float* mc_y1;
  const float* mc_ratios;
  int B, N, dx, dy;
  float tf, s2, cden, g1, g2;
  double* dist;        // [nsx + nsy][B][N] scratch: sliced squared distances (guid_logp -> guid_apply)
  int slice_len, nsx, nsy;  // D-slices per modality (<= RGFM_GUID_SLICES in all)
  // optional (hipGraph replay): the step's scalars {tf, s2, cden, -} come from sched[4 * *step_ptr] instead of the fields
  const float* sched;
  const int* step_ptr;
  float* weights_out;  // optional [B][N]
  float* wbuf;         // [B][N] importance weights of the step (scratch: behind the distance slices)
  float* wsum;         // [B] their row sums (scratch: behind wbuf)
  float* x_state;      // optional fused Euler: x_state += dt * blended
  float* y_state;
  float dt;
};
constexpr int RGFM_GUID_SLICES = 8;
void launch_guid_logp(const GuidanceArgs& a, hipStream_t s);   // distances -> GuidanceArgs::dist
void launch_guid_weights(const GuidanceArgs& a, hipStream_t s);  // distances + ratios -> importance weights (GuidanceArgs::wbuf, wsum)
void launch_guid_apply(const GuidanceArgs& a, hipStream_t s);  // guided velocity (the [B,N] x [N,D] GEMM), blend (+ Euler)
int guid_apply_init();                                         // (once per process: the GEMM kernel's LDS size)
void launch_euler(float* x, const float* v, size_t n, float dt, hipStream_t s);
void launch_step_inc(int* step, hipStream_t s);  // *step += 1
void launch_guid_schedule(float* sched, int step_begin, int ns, int num_steps, hipStream_t s);  // [ns][4] = {tf, s2, cden, 0}

}  // namespace rgfm
