// conv_mfma_hx2s.hip -- the stride-2 3x3 conv of the U-Nets' Downsample (reference src/models/unet_flexible.py:88-97:
// Conv2d(ch, ch, 3, stride=2, padding=1) on the un-normalised residual stream) on the two-plane fp16 arithmetic
// (conv_hx2_common.h), one K iteration per 16-channel chunk.
//
// conv_mfma_hx2_kernel<*, CONV_S2, *> runs this conv as (phase, chunk) iterations -- the input read as its four
// pixel-parity planes, one plane staged and 1 / 2 / 2 / 4 taps multiplied per iteration: 4 x Cin / 16 iterations of two
// barriers and one exposed round trip each around 2.25 taps of matrix work (MFMA-busy 0.06 - 0.13, 0.20 of the layers'
// roofline).  Here a chunk's FOUR planes are staged together (74 - 83 KB of LDS) and all nine taps run between two
// barriers; the next chunk's raw values are fetched into registers and its weights arrive by LDS-DMA in the other weight
// buffer while the taps run.  Output (r, x) takes input (2r + ky - 1, 2x + kx - 1): kernel row ky = 1 is plane
// a = 0 at plane row r, ky = 0 / 2 plane a = 1 at plane row r - 1 / r -- likewise the columns.  A plane tile is
// (th + 1) x (W + 1) records per sample (one border row / column at the top / left, zero where it leaves the image).
//
// Scope (conv_hx2s_supported): raw single-source input, no residual / time term / epilogue activation, square outputs
// of 16x16 (one sample per tile) or 8x8 (four samples per tile), Cout a multiple of 32.  Weights: the plain nine-tap
// fp16 image (launch_pack_conv_hx2 with CONV_S1: ConvArgs::wpkh9), a workgroup reads one 32- / 64-channel block or its
// half of a 128-channel block.  Summation order: chunk by chunk, taps (ky, kx) in raster order inside a chunk, the three
// plane products of a tap as in the other hx2 kernels; a row's result depends on its own sample only.
#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

template <int WL2, int NG>
__global__ __launch_bounds__(256 * NG, 1) void conv_mfma_hx2s_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int W = 1 << WL2, TH = W, SPT = (W == 16) ? 1 : 4;  // output raster W x W; samples per 256-pixel tile
  static_assert(W == 16 || W == 8, "16x16 or 8x8 outputs");
  constexpr int PW = W + 1, PR = TH + 1, PREC = PR * PW;  // plane tile of one sample
  constexpr int NREC = 4 * SPT * PREC;                    // records of the four planes
  constexpr int ABYTES = (NREC + 1) * HRW;                // + a pad record (the store target of lanes past the end)
  constexpr int NTHR = 256 * NG, NW = 4 * NG;
  constexpr int CB = 32 * NG;         // output channels per workgroup
  constexpr int TAPB = CB * HRW;      // one tap's weight slab
  constexpr int CHB = 9 * TAPB;       // one chunk's weights
  constexpr int PPT = TAPB / 1024;    // 1-KB DMA pieces per tap
  constexpr int NPC = 9 * PPT;        // pieces per chunk
  constexpr int NPW = (NPC + NW - 1) / NW;
  constexpr int NIT = (NREC * 4 + NTHR - 1) / NTHR;  // (pixel, 4 channels) items per thread and chunk
  constexpr int MT_OFF = (32 / W) * PW * HRW;        // pixel p + 32 of a segment: 2 (W = 16) or 4 (W = 8) plane rows down
  extern __shared__ __attribute__((aligned(16))) char sms[];
  char* const sA = sms;
  char* const sB = sms + ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31 = lane & 31, hp = lane >> 5;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int b0 = tile * SPT;  // first sample of the tile
  const int Hin = 2 * TH, Win = 2 * W, cin = a.C0, nch = cin / KC;

  // ---- per-item decode, once: source offset (floats) of the item's four channels in chunk 0, LDS destination, validity
  const int q4 = tid & 3;
  unsigned poff[NIT];
  int adst[NIT];
  unsigned okm = 0u;
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int it = tid + NTHR * j;
    poff[j] = 0u, adst[j] = NREC * HRW + (q4 >> 1) * 16 + (q4 & 1) * 8;
    if (it < NREC * 4) {
      const int rec = it >> 2;
      const int p = rec / (SPT * PREC), rem = rec - p * (SPT * PREC);
      const int s = rem / PREC, rr = rem - s * PREC;
      const int hr = rr / PW, hc = rr - hr * PW;
      const int iy = 2 * (hr - 1) + (p >> 1), ix = 2 * (hc - 1) + (p & 1);  // (<= 2 TH - 1 / 2 W - 1: inside by construction)
      adst[j] = rec * HRW + ((((q4 >> 1) ^ (hc >> 2)) & 3) * 16) + (q4 & 1) * 8;  // plane l: ^ 32
      if (iy >= 0 && ix >= 0 && b0 + s < a.B) {
        okm |= 1u << j;
        poff[j] = (unsigned)(((b0 + s) * Hin + iy) * Win + ix) * (unsigned)cin + (unsigned)(q4 * 4);
      }
    }
  }

  // ---- fragment offsets: this lane's pixel 64 seg + l31 (+ 32: MT_OFF) in a plane tile, at plane column x (+ 1)
  int aofs[2];
  {
    const int p = 64 * seg + l31;
    const int s = SPT == 1 ? 0 : seg, pp = SPT == 1 ? p : l31;  // (8x8: a segment is a sample)
    const int r = pp >> WL2, x = pp & (W - 1);
    const int arec = s * PREC + r * PW + x;
#pragma unroll
    for (int dc = 0; dc < 2; ++dc) aofs[dc] = (arec + dc) * HRW + ((hp ^ (((x + dc) >> 2) & 3)) & 3) * 16;
  }
  int bofs;
  {
    const int rec = grp * 32 + l31;
    bofs = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
  }

  // ---- weights: [channel block][chunk][tap] slabs; a workgroup of a 128-channel block takes its 64-channel half
  const bool nb128 = CB == 64 && (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh9) + (size_t)wblk * nch * 9 * TAPS + whalf;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned sB_lds = (unsigned)(size_t)sB;
  auto wdma = [&](int ch) {  // chunk ch -> weight buffer ch & 1 (NPW pieces per wave; surplus ones repeat a piece)
    const char* src = wpk + (size_t)ch * 9 * TAPS;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pq = wave_s + NW * j;
      const int pc = pq < NPC ? pq : pq - NPC;
      const char* gsrc = src + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16;
      const unsigned dst = sB_lds + (unsigned)((ch & 1) * CHB + pc * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };

  f32x4 ra[NIT];
  float hmax = 0.f;
  auto issue = [&](int ch) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) ra[j] = *(const hx_gf32x4*)(a.in0 + (size_t)(poff[j] + (unsigned)(ch * KC)));
  };
  auto commit = [&]() {  // raw values -> S_A x value in two fp16 planes (zero where the plane leaves the image)
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const float sa = ((okm >> j) & 1u) ? HX_SA : 0.f;
      const f32x4 v = ra[j] * sa;
      unsigned h0, l0, h1, l1;
      hsplit2(v.x, v.y, h0, l0);
      hsplit2(v.z, v.w, h1, l1);
      hmax = hx_absmax3(v.x, v.y, hmax);
      hmax = hx_absmax3(v.z, v.w, hmax);
      const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
      *reinterpret_cast<hx_u32x2*>(sA + adst[j]) = ph;
      *reinterpret_cast<hx_u32x2*>(sA + (adst[j] ^ 32)) = pl;
    }
  };

  // ---- accumulators: bias, scaled by q (they hold q x the true sums)
  const float qmain = a.hq[0];
  const int ch0 = cb * CB + grp * 32 + l31;  // this lane's output channel
  f32x16 acc[2];
  {
    const float add0 = a.bias[ch0] * qmain;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = add0;
  }

  // one tap: kernel row KY, column KX -> plane (KY != 1, KX != 1) at plane row r + (KY != 0), column x + (KX != 0)
  auto tap = [&](auto ky_tag, auto kx_tag, const char* sBc) {
    constexpr int KY = decltype(ky_tag)::value, KX = decltype(kx_tag)::value;
    constexpr int P = (KY != 1 ? 2 : 0) + (KX != 1 ? 1 : 0), DR = KY != 0 ? 1 : 0, DC = KX != 0 ? 1 : 0;
    const char* sAp = sA + (P * SPT * PREC + DR * PW) * HRW;
    const char* sBt = sBc + (KY * 3 + KX) * TAPB;
    const int o0 = aofs[DC], o1 = o0 ^ 32;
    f16x8 af[2][2], bf[2];
    af[0][0] = *reinterpret_cast<const f16x8*>(sAp + o0);
    af[0][1] = *reinterpret_cast<const f16x8*>(sAp + o1);
    af[1][0] = *reinterpret_cast<const f16x8*>(sAp + o0 + MT_OFF);
    af[1][1] = *reinterpret_cast<const f16x8*>(sAp + o1 + MT_OFF);
    bf[0] = *reinterpret_cast<const f16x8*>(sBt + bofs);
    bf[1] = *reinterpret_cast<const f16x8*>(sBt + (bofs ^ 32));
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[PB[q]], acc[mt], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- K loop
  issue(0);
  wdma(0);
#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    if (ch) __syncthreads();  // every wave has read the planes of chunk ch - 1
    commit();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's weight pieces of chunk ch have landed
    __syncthreads();
    if (ch + 1 < nch) {
      issue(ch + 1);
      wdma(ch + 1);
    }
    const char* sBc = sB + (ch & 1) * CHB;
    tap(I0{}, I0{}, sBc), tap(I0{}, I1{}, sBc), tap(I0{}, I2{}, sBc);
    tap(I1{}, I0{}, sBc), tap(I1{}, I1{}, sBc), tap(I1{}, I2{}, sBc);
    tap(I2{}, I0{}, sBc), tap(I2{}, I1{}, sBc), tap(I2{}, I2{}, sBc);
  }
  if (!(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);  // (rare) plane h would be >= 32768 (or inf)

  // ---- epilogue: every pixel of a sample that exists is valid (whole samples per segment / tile)
  {
    const float qinv = a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = acc[mt] * qinv;
  }
  const int sample = SPT == 1 ? tile : b0 + seg;
  if (sample >= a.B) return;  // (wave-uniform)
  if (a.small_check && a.range_flag) {  // (ConvArgs::small_check: the output's low range)
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][r], acc[mt][r + 1], m);
    hx_small_flag(a.range_flag, m);
  }
  const size_t pix0 = (size_t)tile * 256 + 64 * seg;  // (whole samples: the tile's pixels are consecutive in the output)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
      a.out[(pix0 + pl) * a.Cout + ch0] = acc[mt][r];
    }
  if (a.stats_out) {
    const int part = SPT == 1 ? seg : 0;
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[mt][r];
    s += __shfl_xor(s, 32);
    const float mean = s / 64.f;
    float m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[mt][r] - mean;
        m2 += d * d;
      }
    m2 += __shfl_xor(m2, 32);
    if (hp == 0) store_stats(a, a.stats_out + (((size_t)sample * a.g.nparts + part) * a.Cout + ch0) * 2, mean, m2);
  }
}

// ---------------------------------------------------------------- host side
static int g_hx2s_on = 1;
void conv_hx2s_set(int v) { g_hx2s_on = v; }

static int hx2s_ng(const ConvArgs& a) { return a.Cout % 64 == 0 ? 2 : 1; }
static size_t hx2s_lds_bytes(const ConvArgs& a) {
  const int W = a.g.W, spt = W == 16 ? 1 : 4;
  return (size_t)(4 * spt * (W + 1) * (W + 1) + 1) * HRW + (size_t)2 * 9 * 32 * hx2s_ng(a) * HRW;
}

bool conv_hx2s_supported(const ConvArgs& a, int mode) {
  if (!g_hx2s_on || mode != CONV_S2) return false;
  if (!a.wpkh9 || !a.hq || !a.range_flag) return false;
  if (a.C1 != 0 || a.ab || a.gn_stats0 || a.res_mode != 0 || a.temb || a.ep_scale || a.fin_ab) return false;
  const TileGeom& g = a.g;
  if (!((g.W == 16 && g.H == 16 && g.spt == 1 && g.th == 16 && g.tps == 1) || (g.W == 8 && g.H == 8 && g.spt == 4))) return false;
  if (a.Hin != 2 * g.H || a.Win != 2 * g.W) return false;
  if (a.C0 % KC != 0 || a.Cout % 32 != 0) return false;
  // 32-bit element offsets inside the kernel
  if ((size_t)a.B * a.Hin * a.Win * a.C0 >= (1ull << 32) || (size_t)a.B * g.HW * a.Cout >= (1ull << 32)) return false;
  return hx2s_lds_bytes(a) <= 160 * 1024;
}

int conv_hx2s_init() {
  int rc = 0;
#define RAISES(WL, G) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2s_kernel<WL, G>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISES(4, 1); RAISES(4, 2); RAISES(3, 1); RAISES(3, 2);
#undef RAISES
  return rc;
}

void launch_conv_hx2s(const ConvArgs& a, hipStream_t s) {
  const int ng = hx2s_ng(a), tiles = geom_num_tiles(a.g, a.B);
  const dim3 grid(tiles, a.Cout / (32 * ng));
  const size_t lds = hx2s_lds_bytes(a);
#define LAUNCHS(WL, G) hipLaunchKernelGGL((conv_mfma_hx2s_kernel<WL, G>), grid, dim3(256 * (G)), lds, s, a, tiles)
  if (a.g.W == 16) {
    if (ng == 2) LAUNCHS(4, 2);
    else LAUNCHS(4, 1);
  } else {
    if (ng == 2) LAUNCHS(3, 2);
    else LAUNCHS(3, 1);
  }
#undef LAUNCHS
}

}  // namespace rgfm
