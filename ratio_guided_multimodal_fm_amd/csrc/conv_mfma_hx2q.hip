// conv_mfma_hx2q.hip -- the two-plane fp16 conv (conv_mfma_hx2p.hip's arithmetic, LDS images and summation order, so
// a row's result is bit for bit the pipelined kernel's) cut for FOUR waves per SIMD: a workgroup is one 256-pixel tile x
// 64 output channels (8 waves = 4 pixel segments x 2 groups of 32 channels, one 32x32 accumulator pair per wave), at
// most 128 VGPRs and 80 KB of LDS, so TWO workgroups share a CU.
//
// Why: conv_mfma_hx2p_kernel needs 195-255 VGPRs, i.e. one workgroup per CU, and every workgroup of a launch starts
// at the same time: all CUs run their prologues together (residual tile, GroupNorm partials, pipeline fill: memory
// round trips at ~11 B/clk/CU, matrix pipe idle), then their K loops together (memory idle), then their store tails.
// On the short-K layers (Cin <= 128 at 32x32: a third of a U-Net step) the prologue is as long as the K loop.  With
// two independent workgroups per CU one's memory phases run under the other's K loop.
//
// Scope: stride-1 / upsampling 3x3 convs over 16- or 32-pixel-wide rasters whose tiles are whole (256 / W rows each),
// Cout % 64 == 0; everything else stays on conv_mfma_hx2p_kernel / conv_mfma_hx2_kernel (conv_hx2q_supported).
// Same ConvArgs, same packed weights (a workgroup reads one 64-channel block, or its half of a 128-channel block).
#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

#ifdef RGFM_HX2Q_PROF  // (tools/kbench: phase stamps of wave 0 / wave 4 of every workgroup, summed)
__device__ unsigned long long g_hx2q_prof[16];
#define QPROF_T(var) const long long var = __builtin_amdgcn_s_memtime()
#else
#define QPROF_T(var)
#endif

#ifndef RGFM_HX2Q_FAST
#define RGFM_HX2Q_FAST 1  // (0: every unit through the general staging path -- kbench A/B)
#endif
#ifdef RGFM_HX2Q_SKEW  // (tools/kbench experiment: the second workgroup of every CU starts late, so that the two stay out of phase)
__device__ int g_hx2q_skew[2];  // {sleep iterations of 4096 cycles, workgroups of the first dispatch round per slot}
#endif

template <int MODE, int WL2>
__global__ __launch_bounds__(512, 4) void conv_mfma_hx2q_kernel(const ConvArgs a, const int num_tiles) {
  static_assert(MODE == CONV_S1 || MODE == CONV_UP2, "stride-2 / transposed convs run on conv_mfma_hx2_kernel");
  constexpr int W = 1 << WL2, TH = 256 / W, WR = W + 2, HR = TH + 2, HALO = HR * WR;
  constexpr int ABYTES = (HALO + 1) * HRW;  // one halo buffer + a pad record (the store target of lanes past the halo)
  constexpr int TAPB = 64 * HRW;            // one tap's weight slab: 64 channels
  constexpr int UB = 3 * TAPB;              // one unit: a kernel row of a 16-channel chunk
  constexpr int MT_OFF = (32 / W) * WR * HRW;  // pixel p + 32 of a segment: one (W = 32) or two (W = 16) halo rows down, same column
  constexpr int NIT = 3;                    // halo items (pixel, 4 channels) per thread and chunk: ceil(HALO * 4 / 512)
  static_assert(HALO * 4 <= NIT * 512 && HALO * 4 > (NIT - 1) * 512, "three items per thread");
  extern __shared__ __attribute__((aligned(16))) char smq[];
#ifdef RGFM_HX2Q_SKEW
  if ((int)blockIdx.x >= g_hx2q_skew[1] && (int)blockIdx.x < 2 * g_hx2q_skew[1])
    for (int i = 0; i < g_hx2q_skew[0]; ++i) __builtin_amdgcn_s_sleep(64);
#endif
  QPROF_T(tq0);
  char* const sB = smq + 2 * ABYTES;
  float* const sTab = reinterpret_cast<float*>(sB + 2 * UB);  // [2][cin][2]: S_A x (scale, shift) of the sample, and a zero row
  const int cin = a.C0 + a.C1;
  const bool gn_on = a.gn_stats0 != nullptr;
  char* const sDesc = reinterpret_cast<char*>(sTab) + (gn_on ? 2 * cin * 8 : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31 = lane & 31, hp = lane >> 5;
  const int H = a.g.H, HW = a.g.HW;

  // workgroup -> (tile, 64-channel block): the blocks of one tile are dealt to the same XCD (ids 8 apart) next to each
  // other in time, so the second one finds the tile's input in that L2
  const int ncb = a.Cout >> 6;
  int tile, cb;
  {
    const int id = (int)blockIdx.x;
    if ((num_tiles & 7) == 0) {
      const int per = 8 * ncb, gq = id / per, r = id - gq * per;
      cb = r >> 3, tile = gq * 8 + (r & 7);
    } else {
      tile = id / ncb, cb = id - tile * ncb;
    }
  }
  const int b0 = tile / a.g.tps, row0 = (tile - b0 * a.g.tps) * TH;
  const int n0 = cb * 64 + grp * 32;  // first output channel of this wave
  const size_t pix0 = (size_t)b0 * HW + (size_t)row0 * W;

  // ---- consumer-side GroupNorm: scale/shift of this tile's sample from the producers' partial statistics (as
  // conv_mfma_hx2p_kernel: one table row; as many waves as it takes to give every lane one channel)
  if (gn_on) {
    const int gn_cpg = cin >> 3;
    const int gn_wsh = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : 2);  // log2 of the waves that take part (cin <= 256)
    if (wave < (1 << gn_wsh)) {
      const int gn_lpg = 8 << gn_wsh;
      const int gn_gl = lane >> (3 + gn_wsh), gn_sub = lane & (gn_lpg - 1);
      const int gn_gi = wave * (8 >> gn_wsh) + gn_gl;
      const int gn_kmax = (gn_cpg + gn_lpg - 1) / gn_lpg;
      float gam[4], bet[4];
      double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 1
      for (int k = 0; k < gn_kmax; ++k) {
        const int c = gn_gi * gn_cpg + gn_sub + gn_lpg * k;
        const bool have = gn_sub + gn_lpg * k < gn_cpg;
        const bool first = !have || c < a.C0;
        const float* st = first ? a.gn_stats0 : a.gn_stats1;
        const int cs = first ? a.C0 : a.C1, cc = have ? (first ? c : c - a.C0) : 0;
        const int npt = first ? a.gn_nparts0 : a.gn_g.nparts;
        float2 gn_v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
          gn_v[p] = *reinterpret_cast<const float2*>(st + (((size_t)b0 * npt + (p < npt ? p : 0)) * cs + cc) * 2);
        const float gv = a.gn_gamma[have ? c : 0], bv = a.gn_beta[have ? c : 0];
        if (k == 0) gam[0] = gv, bet[0] = bv;
        else if (k == 1) gam[1] = gv, bet[1] = bv;
        else if (k == 2) gam[2] = gv, bet[2] = bv;
        else gam[3] = gv, bet[3] = bv;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const double np = (have && p < npt) ? (double)geom_part_count(a.gn_g, p % a.gn_g.nparts) : 0.0;
          const double mp = (double)gn_v[p].x;
          n += np;
          s1 += np * mp;
          s2 += np > 0.0 ? (double)gn_v[p].y + np * mp * mp : 0.0;
        }
      }
      for (int o = 1; o < gn_lpg; o <<= 1) n += __shfl_xor(n, o), s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
      const double mean = n > 0.0 ? s1 / n : 0.0;
      const double var = n > 0.0 ? s2 / n - mean * mean : 0.0;
      const float gm = (float)mean;
      const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (gn_sub + gn_lpg * k < gn_cpg) {
          const float sc = rstd * gam[k];
          float2 o;
          o.x = HX_SA * sc;
          o.y = HX_SA * (bet[k] - gm * sc);
          *reinterpret_cast<float2*>(sTab + ((size_t)gn_gi * gn_cpg + gn_sub + gn_lpg * k) * 2) = o;
        }
      }
    }
    for (int i = tid; i < 2 * cin; i += 512) sTab[cin * 2 + i] = 0.f;  // the all-zero row of the padding items
  }

  QPROF_T(tq1);
  const int nmain = cin / KC;
  const int nskip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nmain + nskip;
  const int G = 3 * nmain + nskip;
  if (tid < ntot) {  // chunk descriptors: which tensor a chunk comes from (input / concat partner / 1x1-skip sources)
    const bool skip = tid >= nmain;
    const int c = (skip ? tid - nmain : tid) * KC;
    const float* src;
    int cs, cc;
    if (!skip) {
      if (c < a.C0) src = a.in0, cs = a.C0, cc = c;
      else src = a.in1, cs = a.C1, cc = c - a.C0;
    } else {
      if (c < a.R0) src = a.res0, cs = a.R0, cc = c;
      else src = a.res1, cs = a.R1, cc = c - a.R0;
    }
    const unsigned long long pv = reinterpret_cast<unsigned long long>(src + cc);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 d = {(unsigned)pv, (unsigned)(pv >> 32), (unsigned)cs, 0u};
    *reinterpret_cast<u32x4*>(sDesc + tid * 16) = d;
  }

  // ---- fragment offsets.  A: this lane's pixel 64 seg + l31 (+ 32: MT_OFF) at tap column kx, planes h / l; the four
  // 16-byte slots of a halo record are swizzled with its halo column (conv_mfma_hx2p.hip)
  int aofs[3][2];
  {
    const int p = 64 * seg + l31, r = p >> WL2, x = p & (W - 1);
    const int arec = r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int sw = ((x + kx) >> 2) & 3;
      aofs[kx][0] = (arec + kx) * HRW + ((hp ^ sw) & 3) * 16;
      aofs[kx][1] = (arec + kx) * HRW + (((2 + hp) ^ sw) & 3) * 16;
    }
  }
  int bofs[2];
  {
    const int rec = grp * 32 + l31;
    bofs[0] = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
    bofs[1] = rec * HRW + (((2 + hp) ^ (rec >> 2)) & 3) * 16;
  }

  // ---- per-item decode, once: source pixel offset, LDS destination, scale/shift row
  const int q4 = tid & 3;
  int poff[NIT], adst[NIT], trow[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int it = tid + 512 * j;
    poff[j] = 0, adst[j] = HALO * HRW + (q4 >> 1) * 16 + (q4 & 1) * 8, trow[j] = (cin + 4 * q4) * 8;
    if (it < HALO * 4) {
      const int hpx = it >> 2;
      const int hy = hpx / WR, hx = hpx - hy * WR;
      int y, x;
      bool ok;
      if (MODE == CONV_S1) {
        y = row0 + hy - 1, x = hx - 1;
        ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
      } else {
        const int yu = row0 + hy - 1, xu = hx - 1;
        ok = (yu >= 0) && (yu < H) && (xu >= 0) && (xu < W);
        y = yu >> 1, x = xu >> 1;
      }
      adst[j] = hpx * HRW + ((((q4 >> 1) ^ (hx >> 2)) & 3) * 16) + (q4 & 1) * 8;  // plane l: ^ 32
      if (ok) {
        poff[j] = (b0 * a.Hin + y) * a.Win + x;
        trow[j] = (4 * q4) * 8;
      }
    }
  }

  // packed weights: [channel block][chunk][tap] slabs; a workgroup of a 128-channel block takes its 64-channel half
  const bool nb128 = (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  const char* wsk = reinterpret_cast<const char*>(a.wskiph) + (size_t)wblk * nskip * TAPS + whalf;
  // this thread's two 16-byte weight items of a unit (768 items): the second wraps (same bytes to the same address)
  int boff[2], soff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = tid + 512 * j;
    boff[j] = (it < UB / 16 ? it : it - UB / 16) * 16;
    soff[j] = (boff[j] >> 12) * TAPS + (boff[j] & (TAPB - 1));
  }

  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  u16x2 hmax = {0, 0};
  f32x4 ra[NIT], rb[2];
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto chunk_desc = [&](int ch) { return *reinterpret_cast<const u32x4*>(sDesc + ch * 16); };
  auto issue_a = [&](const u32x4& d, int j) {
    const float* src = reinterpret_cast<const float*>(((unsigned long long)d.y << 32) | (unsigned long long)d.x);
    ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j], d.z) + (unsigned)(q4 * 4)));
  };
  auto commit_a = [&](int ch, int j, bool xform) {
    f32x4 v = ra[j];
    if (xform) {
      const char* ep = reinterpret_cast<const char*>(sTab) + ch * (KC * 8) + trow[j];
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 16);
      v.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
      v.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
      v.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
      v.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
    } else {
      const float sa = trow[j] < cin * 8 ? HX_SA : 0.f;  // (the zero row: an out-of-image item)
      v.x *= sa, v.y *= sa, v.z *= sa, v.w *= sa;
    }
    unsigned h0, l0, h1, l1;
    hsplit2(v.x, v.y, h0, l0);
    hsplit2(v.z, v.w, h1, l1);
    const unsigned m = 0x7fff7fffu;
    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(u16x2, h0 & m));
    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(u16x2, h1 & m));
    const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
    char* base = smq + (ch & 1) * ABYTES;
    *reinterpret_cast<hx_u32x2*>(base + adst[j]) = ph;
    *reinterpret_cast<hx_u32x2*>(base + (adst[j] ^ 32)) = pl;
  };
  auto issue_b = [&](int gg) {
    const bool main = gg < 3 * nmain;
    const char* src = main ? wpk + (size_t)gg * 3 * TAPS : wsk + (size_t)(gg - 3 * nmain) * TAPS;
    rb[0] = *(const hx_gf32x4*)(src + (main ? soff[0] : (tid < TAPB / 16 ? tid : 0) * 16));
    if (main) rb[1] = *(const hx_gf32x4*)(src + soff[1]);
  };
  auto commit_b = [&](int gg) {
    char* dst = sB + (gg & 1) * UB;
    if (gg < 3 * nmain) {
      *reinterpret_cast<f32x4*>(dst + boff[0]) = rb[0];
      *reinterpret_cast<f32x4*>(dst + boff[1]) = rb[1];
    } else if (tid < TAPB / 16) {
      *reinterpret_cast<f32x4*>(dst + tid * 16) = rb[0];
    }
  };

  QPROF_T(tq2);
  __syncthreads();  // the scale/shift table and the chunk descriptors are complete
  QPROF_T(tq3);
  // ---- pipeline fill: halo of chunk 0 and weights of unit 0 in LDS, raw halo of chunk 1 and weights of unit 1 in registers
  {
    const u32x4 d0 = chunk_desc(0);
#pragma unroll
    for (int j = 0; j < NIT; ++j) issue_a(d0, j);
    issue_b(0);
  }
  // bias (+ skip bias + time embedding), scaled by q: the accumulators hold q x the true sums; an identity residual
  // is fetched raw into the accumulators here and scaled once the fill is through (its loads fly meanwhile)
  const float qmain = a.hq[0];
  f32x16 acc[2];
  float add0;
  {
    const int c = n0 + l31;
    float v = a.bias[c];
    if (a.res_mode == 2) v += a.skip_bias[c];
    if (a.temb) v += a.temb[((size_t)(a.temb_per_row ? b0 : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c];
    add0 = v * qmain;
  }
  if (a.res_mode == 1) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = 64 * seg + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
        acc[mt][r] = a.res0[(size_t)(__umul24((unsigned)pix0 + (unsigned)p, (unsigned)a.Cout) + (unsigned)(n0 + l31))];
      }
  }
  {
    const bool xf = gn_on && 0 < nmain;
#pragma unroll
    for (int j = 0; j < NIT; ++j) commit_a(0, j, xf);
    commit_b(0);
    if (ntot > 1) {
      const u32x4 d1 = chunk_desc(1);
#pragma unroll
      for (int j = 0; j < NIT; ++j) issue_a(d1, j);
    }
    issue_b(1);
  }
  if (a.res_mode == 1) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(acc[mt][r], qmain, add0);
  } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = add0;
  }
  QPROF_T(tq4);
  __syncthreads();
  QPROF_T(tq5);

  // one tap (kernel column KX of the halo row at sArow): 6 fragment reads, 6 MFMAs (a_l w_h, a_h w_l, a_h w_h per tile)
  auto tap = [&](const char* sArow, const char* sBt, int o0, int o1) {
    f16x8 af[2][2], bf[2];
    af[0][0] = *reinterpret_cast<const f16x8*>(sArow + o0);
    af[0][1] = *reinterpret_cast<const f16x8*>(sArow + o1);
    af[1][0] = *reinterpret_cast<const f16x8*>(sArow + o0 + MT_OFF);
    af[1][1] = *reinterpret_cast<const f16x8*>(sArow + o1 + MT_OFF);
    bf[0] = *reinterpret_cast<const f16x8*>(sBt + bofs[0]);
    bf[1] = *reinterpret_cast<const f16x8*>(sBt + bofs[1]);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[PB[q]], acc[mt], 0, 0, 0);
  };
  // staging work of unit g = (chunk c, item U; U < 0: a one-tap unit, every item): weights of unit g + 1 -> LDS, of unit
  // g + 2 -> registers; halo item U of chunk c + 1 -> LDS, of chunk c + 2 -> registers
  auto stage = [&](int c, int U, int gidx) {
    const bool have1 = c + 1 < ntot, have2 = c + 2 < ntot;
    const bool xf = gn_on && c + 1 < nmain;
    u32x4 dn = {0u, 0u, 0u, 0u};
    if (have2) dn = chunk_desc(c + 2);
    if (gidx + 1 < G) commit_b(gidx + 1);
    if (have1) {
      if (U >= 0) commit_a(c + 1, U, xf);
      else {
#pragma unroll
        for (int j = 0; j < NIT; ++j) commit_a(c + 1, j, xf);
      }
    }
    if (have2) {
      if (U >= 0) issue_a(dn, U);
      else {
#pragma unroll
        for (int j = 0; j < NIT; ++j) issue_a(dn, j);
      }
    }
    if (gidx + 2 < G) issue_b(gidx + 2);
  };

  int gidx = 0;
  // ---- main body: chunks 0 .. nmain - 2 (the next chunk is a main chunk too, unit g + 2 exists).  Straight-line, no
  // conditionals, so that hipcc counts its vmcnt waits: every unit first stores what was fetched a unit (weights) or
  // two units (halo) earlier, then fetches -- the weights FIRST, so that the wait for them at the top of the next unit
  // (vmcnt is in order) leaves the slower halo fetch behind them in flight -- then multiplies.  Halo items of chunk
  // c + 1 / c + 2: item 0 in unit 0, items 1 and 2 in unit 1, none in unit 2: at the loop's back edge (where hipcc
  // waits for vmcnt(0)) only the weights of the next unit are outstanding.
  auto wfetch = [&](int gg) {  // main unit gg -> rb
    const char* src = wpk + (size_t)gg * 3 * TAPS;
    rb[0] = *(const hx_gf32x4*)(src + soff[0]);
    rb[1] = *(const hx_gf32x4*)(src + soff[1]);
  };
  auto wstore = [&](int gg) {
    char* dst = sB + (gg & 1) * UB;
    *reinterpret_cast<f32x4*>(dst + boff[0]) = rb[0];
    *reinterpret_cast<f32x4*>(dst + boff[1]) = rb[1];
  };
  auto taps3 = [&](int c, int U) {
    const char* sArow = smq + (c & 1) * ABYTES + U * WR * HRW;
    const char* sBu = sB + (gidx & 1) * UB;
    tap(sArow, sBu, aofs[0][0], aofs[0][1]);
    tap(sArow, sBu + TAPB, aofs[1][0], aofs[1][1]);
    tap(sArow, sBu + 2 * TAPB, aofs[2][0], aofs[2][1]);
  };
  int c0 = 0;
#if RGFM_HX2Q_FAST
  auto main_loop = [&](auto xf_tag) {
    constexpr bool XF = decltype(xf_tag)::value;
#pragma unroll 1
    for (; c0 < nmain - 1; ++c0) {
      const int c2 = c0 + 2 < ntot ? c0 + 2 : ntot - 1;  // (no such chunk: a harmless re-fetch, never stored)
      const u32x4 dn = chunk_desc(c2);
      // unit 0
      wstore(gidx + 1);
      commit_a(c0 + 1, 0, XF);
      wfetch(gidx + 2);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(dn, 0);
      __builtin_amdgcn_sched_barrier(0);
      taps3(c0, 0);
      __syncthreads();
      ++gidx;
      // unit 1
      wstore(gidx + 1);
      commit_a(c0 + 1, 1, XF);
      commit_a(c0 + 1, 2, XF);
      wfetch(gidx + 2);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(dn, 1);
      issue_a(dn, 2);
      __builtin_amdgcn_sched_barrier(0);
      taps3(c0, 1);
      __syncthreads();
      ++gidx;
      // unit 2
      wstore(gidx + 1);
      wfetch(gidx + 2);
      __builtin_amdgcn_sched_barrier(0);
      taps3(c0, 2);
      __syncthreads();
      ++gidx;
    }
  };
  main_loop(std::true_type{});  // (conv_hx2q_supported: the input takes the consumer-side norm)
#endif
  // the remaining main chunks (the last one; all of them without RGFM_HX2Q_FAST): general staging, the two waves of
  // this workgroup on a SIMD (w, w + 4) stage at opposite ends of a unit
#define HX2Q_UNIT(c, U)                                                \
  do {                                                                 \
    const char* sArow = smq + ((c) & 1) * ABYTES + (U) * WR * HRW;    \
    const char* sBu = sB + (gidx & 1) * UB;                            \
    if (grp == 0) stage((c), (U), gidx);                               \
    tap(sArow, sBu, aofs[0][0], aofs[0][1]);                           \
    tap(sArow, sBu + TAPB, aofs[1][0], aofs[1][1]);                    \
    tap(sArow, sBu + 2 * TAPB, aofs[2][0], aofs[2][1]);                \
    if (grp != 0) stage((c), (U), gidx);                               \
    if (gidx != G - 1) __syncthreads();                                \
    ++gidx;                                                            \
  } while (0)
#pragma unroll 1
  for (int c = c0; c < nmain; ++c) {
    HX2Q_UNIT(c, 0);
    HX2Q_UNIT(c, 1);
    HX2Q_UNIT(c, 2);
  }
#undef HX2Q_UNIT
  if (nskip) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
    const float rs = a.hq_skip[0] * a.hq[1];
    acc[0] = acc[0] * rs, acc[1] = acc[1] * rs;
#pragma unroll 1
    for (int c = nmain; c < ntot; ++c, ++gidx) {
      if (grp == 0) stage(c, -1, gidx);
      tap(smq + (c & 1) * ABYTES + WR * HRW, sB + (gidx & 1) * UB, aofs[1][0], aofs[1][1]);  // (centre tap: row 1, column 1)
      if (grp != 0) stage(c, -1, gidx);
      if (gidx != G - 1) __syncthreads();
    }
  }
  QPROF_T(tq6);
  {
    const float qinv = nskip ? a.hq_skip[1] : a.hq[1];
    acc[0] = acc[0] * qinv, acc[1] = acc[1] * qinv;
    if (hmax[0] >= 0x7800 || hmax[1] >= 0x7800) atomicOr(a.range_flag, 1u);  // (rare) |a'| >= 32768 (or inf / nan)
  }

  // ---------------------------------------------------------------- epilogue: every pixel of the tile is valid
  {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = 64 * seg + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
        a.out[(size_t)(__umul24((unsigned)pix0 + (unsigned)p, (unsigned)a.Cout) + (unsigned)(n0 + l31))] = acc[mt][r];
      }
    if (a.stats_out) {
      const int nparts = a.g.nparts;
      const int part = (tile - b0 * a.g.tps) * 4 + seg;
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
      if (hp == 0) store_stats(a, a.stats_out + (((size_t)b0 * nparts + part) * a.Cout + n0 + l31) * 2, mean, m2);
      if (a.fin_ab) fin_arrive(a, b0, lane, nparts, false);
    }
  }
#ifdef RGFM_HX2Q_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QPROF_T(tq7);
  if (lane == 0 && seg == 0) {
    unsigned long long* pp = g_hx2q_prof + grp * 8;
    atomicAdd(pp + 0, (unsigned long long)(tq1 - tq0));  // GroupNorm table
    atomicAdd(pp + 1, (unsigned long long)(tq2 - tq1));  // descriptors, offsets, item decode
    atomicAdd(pp + 2, (unsigned long long)(tq3 - tq2));  // barrier
    atomicAdd(pp + 3, (unsigned long long)(tq4 - tq3));  // fill + accumulator init
    atomicAdd(pp + 4, (unsigned long long)(tq5 - tq4));  // barrier
    atomicAdd(pp + 5, (unsigned long long)(tq6 - tq5));  // K loop
    atomicAdd(pp + 6, (unsigned long long)(tq7 - tq6));  // epilogue incl. store drain
    atomicAdd(pp + 7, 1ull);
  }
#endif
}

// ---------------------------------------------------------------- host side
static size_t hx2q_lds_bytes(const ConvArgs& a) {
  const int W = a.g.W, halo = (256 / W + 2) * (W + 2);
  size_t bytes = (size_t)2 * (halo + 1) * HRW + (size_t)2 * 3 * 64 * HRW;
  if (a.gn_stats0) bytes += (size_t)2 * (a.C0 + a.C1) * 2 * sizeof(float);
  bytes += (size_t)(a.C0 + a.C1 + (a.res_mode == 2 ? a.R0 + a.R1 : 0));  // 16 bytes per 16-channel chunk: descriptors
  return bytes;
}

// launches with fewer workgroups than this stay on conv_mfma_hx2p_kernel (0: this kernel never runs)
static int g_hx2q_min = 256;
void conv_hx2q_set_min(int v) { g_hx2q_min = v; }

bool conv_hx2q_supported(const ConvArgs& a, int mode) {
  if (!g_hx2q_min) return false;
  if (mode != CONV_S1 && mode != CONV_UP2) return false;
  if (!a.gn_stats0) return false;  // convs of raw inputs (the upsamplers) stay on conv_mfma_hx2p_kernel
  if (!conv_hx2_supported(a, mode) || !conv_hx2_gn_supported(a, mode)) return false;
  const TileGeom& g = a.g;
  if (g.spt != 1 || (g.W != 16 && g.W != 32) || g.th * g.W != 256 || g.H % g.th != 0) return false;
  if (a.Cout % 64 != 0 || (a.C0 + a.C1) % KC != 0) return false;
  if (a.res_mode == 2 && (a.R0 + a.R1) % KC != 0) return false;
  if (hx2q_lds_bytes(a) > 80 * 1024) return false;
  return geom_num_tiles(g, a.B) * (a.Cout / 64) >= g_hx2q_min;
}

int conv_hx2q_init() {
  int rc = 0;
#define RAISEQ(M, WL) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2q_kernel<M, WL>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
  RAISEQ(CONV_S1, 4); RAISEQ(CONV_S1, 5); RAISEQ(CONV_UP2, 4); RAISEQ(CONV_UP2, 5);
#undef RAISEQ
  return rc;
}

int conv_hx2q_fin_expected(const ConvArgs& a) { return a.g.tps * 4 * (a.Cout / 32); }

void launch_conv_hx2q(const ConvArgs& a_in, int mode, hipStream_t s) {
  ConvArgs a = a_in;
  const int tiles = geom_num_tiles(a.g, a.B);
  if (a.fin_ab) a.fin_expected = conv_hx2q_fin_expected(a);
  const dim3 grid(tiles * (a.Cout / 64));
  const size_t lds = hx2q_lds_bytes(a);
#define LAUNCHQ(M, WL) hipLaunchKernelGGL((conv_mfma_hx2q_kernel<M, WL>), grid, dim3(512), lds, s, a, tiles)
  if (mode == CONV_S1) {
    if (a.g.W == 32) LAUNCHQ(CONV_S1, 5);
    else LAUNCHQ(CONV_S1, 4);
  } else {
    if (a.g.W == 32) LAUNCHQ(CONV_UP2, 5);
    else LAUNCHQ(CONV_UP2, 4);
  }
#undef LAUNCHQ
}

}  // namespace rgfm
