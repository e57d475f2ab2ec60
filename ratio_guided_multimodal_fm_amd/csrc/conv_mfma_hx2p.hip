// conv_mfma_hx2p.hip -- pipelined, staggered version of conv_mfma_hx2_kernel for the stride-1 and upsampling convs
// (CONV_S1, CONV_UP2: 96 % of the conv time of a U-Net evaluation).  Same arithmetic (two scaled fp16 planes, three
// f16-MFMA products per fp32 product), same tiling, prologue and epilogue; the K loop is re-cut so that the matrix
// pipe of a SIMD always has one of its two waves to feed it:
//
//   * The K loop runs in UNITS of three taps (one kernel row of a 16-channel chunk: 36 MFMAs per wave; a 1x1-skip
//     chunk is a one-tap unit) with ONE barrier per unit.  Weights are double-buffered per unit, the activation halo
//     per chunk: while unit g is multiplied, every wave stores the weights of unit g+1 (fetched into registers one
//     unit earlier), fetches those of unit g+2, transforms + splits + stores its slice of the halo of chunk c+1
//     (fetched three units earlier) and fetches the same slice of chunk c+2.  Nothing a unit reads is written
//     during that unit, so a unit needs no barrier inside.
//   * STAGGER: waves 0-3 do that staging BEFORE their MFMAs of the unit, waves 4-7 AFTER.  The two waves of a SIMD
//     (w, w+4) are therefore always in opposite phases: one wave's GroupNorm/SiLU/split VALU work and memory
//     instructions issue while the other wave's MFMAs occupy the matrix pipe, instead of both staging and then both
//     multiplying as in conv_mfma_hx2_kernel (where the pipe idles 55 % of the time).
//
// Tried before this and rejected (tools/experiments/conv_mfma_hx2_prodcons.hip, numbers in DESIGN.md): four dedicated
// producer waves (one per SIMD) beside eight MFMA waves, weights and raw activations by LDS-DMA.  A lone wave is
// latency-bound on everything it does there -- ~280 cycles per global_load_lds issue, ~800 per LDS round trip, 8.5
// cycles per VALU instruction beside two MFMA waves -- so the eight consumers waited for it at every barrier.
//
// The external scale/shift array path (ConvArgs::ab without gn_stats0) and the stride-2 / transposed modes stay on
// conv_mfma_hx2_kernel.
#include <stdlib.h>

#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

#ifdef RGFM_HX2P_PROF
// wave 0 / wave 4: [0]/[4] prologue + fill, [1]/[5] staging, [2]/[6] MFMA units, [3]/[7] barrier waits; [8] epilogue (wave 0); [9] blocks
__device__ unsigned long long g_hx2p_prof[10];
#define PPROF_T(var) const long long var = clock64()
#define PPROF_ADD(slot, t0, t1) pacc[slot] += (t1) - (t0)
#else
#define PPROF_T(var)
#define PPROF_ADD(slot, t0, t1)
#endif

template <int NT, int MODE, bool PAIRN>
__global__ __launch_bounds__(512, 2) void conv_mfma_hx2p_kernel(const ConvArgs a, const int num_tiles) {
  static_assert(MODE == CONV_S1 || MODE == CONV_UP2, "stride-2 / transposed convs run on conv_mfma_hx2_kernel");
  constexpr int NG = PAIRN ? 2 : 1;            // channel groups per block
  constexpr int NA = PAIRN ? 1 : 2;            // pixel tiles per block
  constexpr int NBLK = 32 * NT;                // channels per group
  constexpr int NBT = NBLK * NG;               // channels per block
  constexpr int TAPB = NBT * HRW;              // bytes of one tap's weight slab
  constexpr int UB = 3 * TAPB;                 // weights of one unit (3 taps)
  constexpr int NB = (UB / 16 + 511) / 512;    // 16-byte weight items per thread and unit
  constexpr int MAXIT = (NA * 448 * 4 + 511) / 512;  // halo items (pixel, 4 channels) per thread and chunk
  extern __shared__ __attribute__((aligned(16))) char smemp[];
#ifdef RGFM_HX2P_PROF
  long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PPROF_T(tk0);
  const int abytes = (NA * a.halo_px + 1) * HRW;  // one halo buffer (+ a pad record: the store target of lanes past the halo)
  char* const sB = smemp + 2 * abytes;         // two unit-sized weight buffers
  float* const sTab = reinterpret_cast<float*>(sB + 2 * UB);  // [NA * spt][cin][2] S_A x (scale, shift)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31p = lane & 31, hp_ = lane >> 5;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;

  // tile origins of the block's one or two pixel tiles (block-uniform: scalar registers)
  auto tile_origin = [&](int tile, int& b0, int& row0) {
    if (tile >= num_tiles) {
      b0 = a.B, row0 = 0;  // idle half of the last block: every sample index is out of range
    } else if (g.spt == 1) {
      b0 = tile / g.tps;
      row0 = (tile - b0 * g.tps) * g.th;
    } else {
      b0 = tile * g.spt;
      row0 = 0;
    }
  };
  int tb0_[2], trow0_[2];
  tile_origin(PAIRN ? (int)blockIdx.x : (int)blockIdx.x * 2, tb0_[0], trow0_[0]);
  tile_origin(PAIRN ? (int)blockIdx.x : (int)blockIdx.x * 2 + 1, tb0_[1], trow0_[1]);
  const int ga_w = PAIRN ? 0 : grp;
  const int my_tile = PAIRN ? (int)blockIdx.x : (int)blockIdx.x * 2 + grp;
  const int my_cb = PAIRN ? (int)blockIdx.y * 2 + grp : (int)blockIdx.y;
  const int b0 = ga_w ? tb0_[1] : tb0_[0], row0 = ga_w ? trow0_[1] : trow0_[0];
  // exact n / d for 0 <= n < 2048 as (n * m) >> 16 with m = ceil(65536 / d): full-rate 24-bit multiplies
  // instead of the emulated 32-bit division (n (d - 1) < 65536 holds: n <= 1791, d <= 34)
  const unsigned mW = (65536u + (unsigned)g.W - 1u) / (unsigned)g.W;
  const unsigned mWR = (65536u + (unsigned)g.W + 1u) / (unsigned)(g.W + 2);
  const unsigned mPER = (65536u + (unsigned)((g.th + 2) * (g.W + 2)) - 1u) / (unsigned)((g.th + 2) * (g.W + 2));
  const int n0 = my_cb * NBLK;
  const int pc = (MODE == CONV_T2) ? (int)blockIdx.z : 0, py = pc >> 1, px = pc & 1;
  const int HR = g.th + 2, WR = W + 2;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;

  int arec[2], aofs[2][3][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * seg + 32 * mt + l31p;
    int s, q;
    if (g.spt == 1) {
      s = 0;
      q = p < nvalid ? p : nvalid - 1;
    } else {
      s = seg;
      q = (p & 63) < HW ? (p & 63) : HW - 1;
    }
    const int r = (int)(__umul24((unsigned)q, mW) >> 16), x = q - r * W;
    arec[mt] = (PAIRN ? 0 : grp) * a.halo_px + (s * HR + r) * WR + x;
    // byte offset of this lane's fragment of tap (ky = 0, kx) inside a halo buffer, plane h / l: in THIS kernel the
    // four 16-byte slots of a halo record are swizzled with its halo column, (x >> 2) & 3 -- 16 consecutive
    // columns still cover all 16 slot columns of the bank row, and the term no longer depends on the kernel row,
    // so the six offsets are computed once per block instead of per tap
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int sw = ((x + kx) >> 2) & 3;
      aofs[mt][kx][0] = (arec[mt] + kx) * HRW + ((hp_ ^ sw) & 3) * 16;
      aofs[mt][kx][1] = (arec[mt] + kx) * HRW + (((2 + hp_) ^ sw) & 3) * 16;
    }
  }
  int bbase[NT], bsw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int rec = (PAIRN ? grp : 0) * NBLK + nt * 32 + l31p;
    bbase[nt] = rec * HRW + ((hp_ ^ (rec >> 2)) & 3) * 16;  // plane h (NBT is a multiple of 16: the swizzle does not depend on the tap)
    bsw[nt] = rec * HRW + (((2 + hp_) ^ (rec >> 2)) & 3) * 16;  // plane l
  }

  const int bw = (g.spt == 1) ? b0 : b0 + seg;
  const bool sample_ok = bw < a.B;
  const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W : (size_t)bw * HW;
  const float qmain = a.hq[0];
  f32x16 acc[2][NT];
  {
    float add0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = n0 + nt * 32 + l31p;
      float v = a.bias[c];
      if (a.res_mode == 2) v += a.skip_bias[c];
      if (a.temb && sample_ok) v += a.temb[((size_t)(a.temb_per_row ? bw : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c];
      add0[nt] = v * qmain;  // the accumulators hold q x the true sums
    }
    if (a.res_mode == 1) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp_;
          const int p = 64 * seg + pl;
          const bool valid = (g.spt == 1) ? (sample_ok && p < nvalid) : (sample_ok && pl < HW);
          const unsigned pix = valid ? (unsigned)pix0 + (unsigned)((g.spt == 1) ? p : pl) : 0u;
          const float* rp = a.res0 + (size_t)(__umul24(pix, (unsigned)a.Cout) + (unsigned)(n0 + l31p));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = rp[nt * 32];
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], qmain, add0[nt]);
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = add0[nt];
    }
  }

  // ---- per-item decode, once: source pixel offset, LDS destination, validity bit, scale/shift slot
  const int q4 = tid & 3;
  const int nA = a.halo_px * 4;
  // per item: source pixel offset; LDS destination (plane h; plane l: ^ 32; lanes past the halo: a trash slot, so the
  // staging code has no per-lane branches); byte offset of its scale/shift pairs in the table -- out-of-image /
  // out-of-batch items point at an all-zero row behind the table, which makes silu(0 x + 0) = 0 the zero padding
  int poff[MAXIT], adst[MAXIT], trow[MAXIT];
  unsigned okmask = 0u;
  const int nrows_tab = NA * g.spt;  // table rows (+ the zero row)
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int it = tid + 512 * j;
      poff[j] = 0, adst[j] = NA * a.halo_px * HRW + (q4 >> 1) * 16 + (q4 & 1) * 8, trow[j] = (nrows_tab * (a.C0 + a.C1) + 4 * q4) * 8;
      if (it < NA * nA) {
        const int ga = (NA == 2 && it >= nA) ? 1 : 0;
        const int ita = it - ga * nA;
        const int tb0 = ga ? tb0_[1] : tb0_[0], trow0 = ga ? trow0_[1] : trow0_[0];
        const int hp = ita >> 2;
        // hp < 448; per = HR * WR >= 81 when spt == 4 (hp (per - 1) < 65536 needs hp <= 448: per <= 146 there)
        const int s = (g.spt == 1) ? 0 : (int)(__umul24((unsigned)hp, mPER) >> 16);
        const int rem = hp - s * per;
        const int hy = (int)(__umul24((unsigned)rem, mWR) >> 16), hx = rem - hy * WR;
        const int b = tb0 + s;
        int y, x;
        bool ok;
        if (MODE == CONV_S1 || MODE == CONV_T2) {
          y = trow0 + hy - 1, x = hx - 1;
          ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        } else if (MODE == CONV_S2) {
          const int pi = trow0 + hy - 1, pj = hx - 1;  // phase-plane coordinates; phase (0,0) pixel = (2 pi, 2 pj)
          ok = (pi >= 0) && (pi < H) && (pj >= 0) && (pj < W);
          y = 2 * pi, x = 2 * pj;
        } else {
          const int yu = trow0 + hy - 1, xu = hx - 1;
          ok = (yu >= 0) && (yu < H) && (xu >= 0) && (xu < W);
          y = yu >> 1, x = xu >> 1;
        }
        ok = ok && (b < a.B);
        const int rec = ga * a.halo_px + hp;
        adst[j] = (int)__umul24((unsigned)rec, HRW) + ((((q4 >> 1) ^ (hx >> 2)) & 3) * 16) + (q4 & 1) * 8;  // plane l: ^ 32
        if (ok) {
          poff[j] = (int)__umul24(__umul24((unsigned)b, (unsigned)a.Hin) + (unsigned)y, (unsigned)a.Win) + x;  // < 2^24 pixels
          okmask |= 1u << j;
          trow[j] = ((ga * g.spt + s) * (a.C0 + a.C1) + 4 * q4) * 8;
        }
      }
    }
  }


  const int cin = a.C0 + a.C1;
  const int nmain = cin / KC;                                   // 16-channel chunks of the input
  const int nskip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;  // 1x1-skip chunks (one tap each)
  const int ntot = nmain + nskip;
  const int G = 3 * nmain + nskip;                              // units
  const char* wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)blockIdx.y * nmain * 9 * TAPB;
  const char* wsk = reinterpret_cast<const char*>(a.wskiph) + (size_t)blockIdx.y * nskip * TAPB;

  f32x4 ra[MAXIT], rb[NB];
  // range flag: the largest |fp16| (as a bit pattern, per 16-bit half) this thread has stored; >= 0x7800 is |a'| >= 32768
  typedef unsigned short hx_u16x2 __attribute__((ext_vector_type(2)));
  hx_u16x2 hmax = {0, 0};
  const int nitems = (NA * nA + 511) >> 9;  // items that exist for at least one thread (block-uniform)

  // raw fp32 fetch of item j of chunk ch
  auto issue_a = [&](int ch, int j) {
    const bool skip = ch >= nmain;
    const float* src;
    int cs, cc;
    const int c = (skip ? ch - nmain : ch) * KC;
    if (!skip) {
      if (c < a.C0) src = a.in0, cs = a.C0, cc = c;
      else src = a.in1, cs = a.C1, cc = c - a.C0;
    } else {
      if (c < a.R0) src = a.res0, cs = a.R0, cc = c;
      else src = a.res1, cs = a.R1, cc = c - a.R0;
    }
    ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j], (unsigned)cs) + (unsigned)(cc + q4 * 4)));
  };
  // GroupNorm + SiLU + split + store of item j of chunk ch into halo buffer ch & 1 (branch-free per lane)
  auto commit_a = [&](int ch, int j) {
    const bool xform = ch < nmain && a.gn_stats0 != nullptr;
    f32x4 v = ra[j];
    if (xform) {
      const char* ep = reinterpret_cast<const char*>(sTab) + ch * (KC * 8) + trow[j];
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep);
      const f32x4 e1 = *reinterpret_cast<const f32x4*>(ep + 16);
      v.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
      v.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
      v.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
      v.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
    } else {
      const float sa = ((okmask >> j) & 1u) ? HX_SA : 0.f;
      v.x *= sa, v.y *= sa, v.z *= sa, v.w *= sa;
    }
    unsigned h0, l0, h1, l1;
    hsplit2(v.x, v.y, h0, l0);
    hsplit2(v.z, v.w, h1, l1);
    const unsigned m = 0x7fff7fffu;
    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(hx_u16x2, h0 & m));
    hmax = __builtin_elementwise_max(hmax, __builtin_bit_cast(hx_u16x2, h1 & m));
    const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
    char* base = smemp + (ch & 1) * abytes;
    *reinterpret_cast<hx_u32x2*>(base + adst[j]) = ph;
    *reinterpret_cast<hx_u32x2*>(base + (adst[j] ^ 32)) = pl;
  };
  // weights of unit gg: the packed image is the LDS byte image, a linear 16-byte copy (a skip unit is one tap)
  auto issue_b = [&](int gg) {
    const bool main = gg < 3 * nmain;
    const char* src = main ? wpk + (size_t)gg * UB : wsk + (size_t)(gg - 3 * nmain) * TAPB;
    const int nit = main ? UB / 16 : TAPB / 16;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 512 * j;
      rb[j] = *(const hx_gf32x4*)(src + (size_t)(it < nit ? it : 0) * 16);
    }
  };
  auto commit_b = [&](int gg) {
    const int nit = gg < 3 * nmain ? UB / 16 : TAPB / 16;
    char* dst = sB + (gg & 1) * UB;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 512 * j;
      if (it < nit) *reinterpret_cast<f32x4*>(dst + it * 16) = rb[j];
    }
  };

  if (a.gn_stats0) {
    // ---- consumer-side GroupNorm: scale/shift of this block's sample(s) from the producers' partial statistics,
    // before chunk 0's prefetch registers come alive (with them the partials would spill).  Wave w owns table row
    // w (= ga * spt + s, the slot of smask); lane = group * 8 + sub, sub strides over the group's channels; all of a
    // lane's <= 4 x 16 partials are fetched in one round trip, reduced in fp64 without divisions in the loop:
    //   N = sum n_p, S1 = sum n_p mean_p, S2 = sum [M2_p + n_p mean_p^2]  ->  mean = S1 / N, var = S2 / N - mean^2
    if (wave < NA * g.spt) {
      const int ga = (g.spt == 1) ? wave : (wave >> 2), sl = (g.spt == 1) ? 0 : (wave & 3);
      const int b = (ga ? tb0_[1] : tb0_[0]) + sl;
      const TileGeom gg = a.gn_g;
      const int cpg = cin >> 3, gi = lane >> 3, sub = lane & 7;
      float gam[4], bet[4];
      const bool bok = b < a.B;
      double n = 0.0, s1 = 0.0, s2 = 0.0;
      const int kmax = (cpg + 7) >> 3;  // channels per lane (wave-uniform)
#pragma unroll 1
      for (int k = 0; k < kmax; ++k) {  // one channel (16 partials) per round trip: more at once spills
        const int c = gi * cpg + sub + 8 * k;
        const bool have = bok && sub + 8 * k < cpg;
        const bool first = !have || c < a.C0;  // (no k-th channel: entry 0 of the first source, never used)
        const float* st = first ? a.gn_stats0 : a.gn_stats1;
        const int cs = first ? a.C0 : a.C1, cc = have ? (first ? c : c - a.C0) : 0;
        const int npt = first ? a.gn_nparts0 : gg.nparts;
        const size_t bb = bok ? (size_t)b : 0;
        float2 v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p)
          v[p] = *reinterpret_cast<const float2*>(st + ((bb * npt + (p < npt ? p : 0)) * cs + cc) * 2);
        const float gv = a.gn_gamma[have ? c : 0], bv = a.gn_beta[have ? c : 0];
        if (k == 0) gam[0] = gv, bet[0] = bv;
        else if (k == 1) gam[1] = gv, bet[1] = bv;
        else if (k == 2) gam[2] = gv, bet[2] = bv;
        else gam[3] = gv, bet[3] = bv;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const double np = (have && p < npt) ? (double)geom_part_count(gg, p % gg.nparts) : 0.0;
          const double mp = (double)v[p].x;
          n += np;
          s1 += np * mp;
          s2 += np > 0.0 ? (double)v[p].y + np * mp * mp : 0.0;
        }
      }
      n = sub_sum(n), s1 = sub_sum(s1), s2 = sub_sum(s2);
      const double mean = n > 0.0 ? s1 / n : 0.0;
      const double var = n > 0.0 ? s2 / n - mean * mean : 0.0;
      const float gm = (float)mean;
      const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (sub + 8 * k < cpg) {
          const float sc = rstd * gam[k];
          float2 o;
          o.x = HX_SA * sc;
          o.y = HX_SA * (bet[k] - gm * sc);
          *reinterpret_cast<float2*>(sTab + ((size_t)wave * cin + gi * cpg + sub + 8 * k) * 2) = o;
        }
      }
    }
    // (visible to every wave after the barrier that opens commit(0))
  }

  if (a.gn_stats0)  // the all-zero row of the padding items
    for (int i = tid; i < 2 * (a.C0 + a.C1); i += 512) sTab[nrows_tab * (a.C0 + a.C1) * 2 + i] = 0.f;
  __syncthreads();  // the scale/shift table is complete
  // ---- pipeline fill: halo of chunk 0 and weights of unit 0 in LDS, raw halo of chunk 1 and weights of unit 1 in registers
#pragma unroll
  for (int j = 0; j < MAXIT; ++j)
    if (j < nitems) issue_a(0, j);
  issue_b(0);
#pragma unroll
  for (int j = 0; j < MAXIT; ++j)
    if (j < nitems) commit_a(0, j);
  commit_b(0);
#pragma unroll
  for (int j = 0; j < MAXIT; ++j)
    if (j < nitems) issue_a(1, j);
  issue_b(1);
  __syncthreads();
  PPROF_T(tk1);
  PPROF_ADD(grp * 4 + 0, tk0, tk1);

  // one tap (kernel column KX of the row at byte offset `rowoff`): fragments of this wave's 2 pixel tiles x NT channel
  // tiles, 3 MFMAs per tile pair
  auto tap = [&](const char* sArow, const char* sBt, auto kx_tag) {
    constexpr int KX = decltype(kx_tag)::value;
    f16x8 af[2][2], bf[NT][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      af[mt][0] = *reinterpret_cast<const f16x8*>(sArow + aofs[mt][KX][0]);
      af[mt][1] = *reinterpret_cast<const f16x8*>(sArow + aofs[mt][KX][1]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf[nt][0] = *reinterpret_cast<const f16x8*>(sBt + bbase[nt]);
      bf[nt][1] = *reinterpret_cast<const f16x8*>(sBt + bsw[nt]);
    }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[nt][PB[q]], acc[mt][nt], 0, 0, 0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;

  // staging work of unit g = (chunk c, slice U; U < 0: a one-tap unit, every item): weights of unit g + 1 -> LDS,
  // weights of unit g + 2 -> registers; halo items j % 3 == U of chunk c + 1 -> LDS, of chunk c + 2 -> registers.
  // U is a compile-time constant: the item list of a unit is static and its stores / transforms / fetches sit in
  // straight-line code (only block-uniform scalar branches around whole items).
  auto stage = [&](int c, auto u_tag, int gidx) {
    constexpr int U = decltype(u_tag)::value;
    PPROF_T(ts0);
    if (gidx + 1 < G) commit_b(gidx + 1);
    if (gidx + 2 < G) issue_b(gidx + 2);
    if (c + 1 < ntot) {
#pragma unroll
      for (int j = 0; j < MAXIT; ++j)
        if ((U < 0 || (j % 3) == U) && j < nitems) commit_a(c + 1, j);
    }
    if (c + 2 < ntot) {
#pragma unroll
      for (int j = 0; j < MAXIT; ++j)
        if ((U < 0 || (j % 3) == U) && j < nitems) issue_a(c + 2, j);
    }
    PPROF_T(ts1);
    PPROF_ADD(grp * 4 + 1, ts0, ts1);
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using UA = std::integral_constant<int, -1>;

  int gidx = 0;
  auto unit = [&](int c, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    const char* sAc = smemp + (c & 1) * abytes;
    const char* sBu = sB + (gidx & 1) * UB;
    if (grp == 0) stage(c, u_tag, gidx);
    PPROF_T(tm0);
    const char* sArow = sAc + U * WR * HRW;
    tap(sArow, sBu, K0{});
    tap(sArow, sBu + TAPB, K1{});
    tap(sArow, sBu + 2 * TAPB, K2{});
#ifdef RGFM_HX2P_PROF
    asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][NT - 1][15]));
#endif
    PPROF_T(tm1);
    if (grp != 0) stage(c, u_tag, gidx);
    PPROF_T(tm2);
    if (gidx != G - 1) __syncthreads();
    PPROF_T(tm3);
    PPROF_ADD(grp * 4 + 2, tm0, tm1);
    PPROF_ADD(grp * 4 + 3, tm2, tm3);
    ++gidx;
  };
#pragma unroll 1
  for (int c = 0; c < nmain; ++c) {
    unit(c, U0{});
    unit(c, U1{});
    unit(c, U2{});
  }
  if (nskip) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
    const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
#pragma unroll 1
    for (int c = nmain; c < ntot; ++c, ++gidx) {
      if (grp == 0) stage(c, UA{}, gidx);
      tap(smemp + (c & 1) * abytes + WR * HRW, sB + (gidx & 1) * UB, K1{});  // (centre tap: row 1, column 1)
      if (grp != 0) stage(c, UA{}, gidx);
      if (gidx != G - 1) __syncthreads();
    }
  }
  PPROF_T(te0);
  {
    const float qinv = nskip ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
    if (hmax[0] >= 0x7800 || hmax[1] >= 0x7800) atomicOr(a.range_flag, 1u);  // (rare) |a'| >= 32768 (or inf / nan): the host re-runs on bx3
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_pf_kernel)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31 = lane_e & 31, h = lane_e >> 5;
  // Two instantiations of the same epilogue: FULL (every pixel of this wave's 64-pixel segment is valid -- all
  // waves of all interior tiles) has no per-element predicates, which are a third of its instructions.
  auto epilogue = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    unsigned vmask[2] = {0u, 0u};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int p = 64 * seg + pl;
        const bool valid = FULL || ((g.spt == 1) ? (sample_ok && p < nvalid) : (sample_ok && pl < HW));
        if (!FULL && valid) vmask[mt] |= 1u << r;
        unsigned pix = (unsigned)pix0 + (unsigned)((g.spt == 1) ? p : pl);
        if (MODE == CONV_T2) {
          const int pp = (g.spt == 1) ? row0 * W + p : pl;
          const int rr = (int)(__umul24((unsigned)pp, mW) >> 16), xx = pp - rr * W;
          pix = (unsigned)((bw * (2 * H) + 2 * rr + py) * (2 * W) + 2 * xx + px);
        }
        float* op = a.out + (size_t)(__umul24(pix, (unsigned)a.Cout) + (unsigned)(n0 + l31));
        if (valid) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) op[nt * 32] = acc[mt][nt][r];
        }
      }
    if (a.stats_out) {
      int nw;
      if (FULL) {
        nw = 64;
      } else if (g.spt == 1) {
        nw = nvalid - 64 * seg;
        nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
        if (!sample_ok) nw = 0;
      } else {
        nw = sample_ok ? HW : 0;
      }
      const int nparts = (MODE == CONV_T2) ? 4 * g.nparts : g.nparts;
      const int part = ((g.spt == 1) ? (my_tile - b0 * g.tps) * 4 + seg : 0) + pc * g.nparts;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (FULL || (vmask[mt] & (1u << r))) s += acc[mt][nt][r];
        s += __shfl_xor(s, 32);
        const float mean = nw > 0 ? s / (float)nw : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (FULL || (vmask[mt] & (1u << r))) {
              const float d = acc[mt][nt][r] - mean;
              m2 += d * d;
            }
        m2 += __shfl_xor(m2, 32);
        if (h == 0 && sample_ok) {
          const int c = n0 + nt * 32 + l31;
          store_stats(a, a.stats_out + (((size_t)bw * nparts + part) * a.Cout + c) * 2, mean, m2);
        }
      }
      if (a.fin_ab && sample_ok) fin_arrive(a, bw, lane_e, nparts, MODE == CONV_T2);
    }
  };
  const bool full_seg = sample_ok && ((g.spt == 1) ? (nvalid - 64 * seg >= 64) : (HW == 64));  // wave-uniform
  if (full_seg) epilogue(std::true_type{});
  else epilogue(std::false_type{});

#ifdef RGFM_HX2P_PROF
  PPROF_T(te1);
  PPROF_ADD(8, te0, te1);
  if ((tid & 255) == 0) {
    for (int i = grp * 4; i < grp * 4 + 4; ++i) atomicAdd(&g_hx2p_prof[i], (unsigned long long)pacc[i]);
    if (tid == 0) atomicAdd(&g_hx2p_prof[8], (unsigned long long)pacc[8]), atomicAdd(&g_hx2p_prof[9], 1ull);
  }
#endif
}

// ---------------------------------------------------------------- host side
static int hx2p_halo(const ConvArgs& a) { return a.g.spt * (a.g.th + 2) * (a.g.W + 2); }
static bool hx2p_pairn(const ConvArgs& a) { return a.Cout % 128 == 0; }
static size_t hx2p_lds_bytes(const ConvArgs& a) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const bool pn = hx2p_pairn(a);
  const int na = pn ? 1 : 2, nbt = 32 * nt * (pn ? 2 : 1);
  size_t bytes = (size_t)2 * (na * hx2p_halo(a) + 1) * HRW + (size_t)2 * 3 * nbt * HRW;  // two halo buffers (+ pad record) + two weight units
  if (a.gn_stats0) bytes += (size_t)(na * a.g.spt + 1) * (a.C0 + a.C1) * 2 * sizeof(float);  // scale/shift table + the zero row
  return bytes;
}

// the pipelined kernel takes: stride-1 / upsampling convs whose input norm (if any) is the consumer-side one
bool conv_hx2p_supported(const ConvArgs& a, int mode) {
  if (mode != CONV_S1 && mode != CONV_UP2) return false;
  if (a.ab && !a.gn_stats0) return false;
  if (!conv_hx2_supported(a, mode)) return false;
  if (a.gn_stats0 && !conv_hx2_gn_supported(a, mode)) return false;
  return hx2p_halo(a) <= 448 && hx2p_lds_bytes(a) <= 160 * 1024;
}

int conv_hx2p_init() {
  int rc = 0;
#define RAISEP(NTV, M, P) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2p_kernel<NTV, M, P>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISEP(1, CONV_S1, false); RAISEP(1, CONV_UP2, false);
  RAISEP(2, CONV_S1, false); RAISEP(2, CONV_UP2, false);
  RAISEP(2, CONV_S1, true); RAISEP(2, CONV_UP2, true);
#undef RAISEP
  return rc;
}

void launch_conv_hx2p(const ConvArgs& a_in, int mode, hipStream_t s) {
  ConvArgs a = a_in;
  a.halo_px = hx2p_halo(a_in);
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int tiles = geom_num_tiles(a.g, a.B);
  const bool pn = hx2p_pairn(a);
  dim3 grid(pn ? tiles : (tiles + 1) / 2, pn ? a.Cout / 128 : a.Cout / (32 * nt), 1);
  const size_t lds = hx2p_lds_bytes(a);
#define LAUNCHP(NTV, M, P) hipLaunchKernelGGL((conv_mfma_hx2p_kernel<NTV, M, P>), grid, dim3(512), lds, s, a, tiles)
  if (pn) {
    if (mode == CONV_S1) LAUNCHP(2, CONV_S1, true);
    else LAUNCHP(2, CONV_UP2, true);
  } else if (nt == 2) {
    if (mode == CONV_S1) LAUNCHP(2, CONV_S1, false);
    else LAUNCHP(2, CONV_UP2, false);
  } else {
    if (mode == CONV_S1) LAUNCHP(1, CONV_S1, false);
    else LAUNCHP(1, CONV_UP2, false);
  }
#undef LAUNCHP
}

}  // namespace rgfm
