// conv_mfma_hx2p.hip -- pipelined version of conv_mfma_hx2_kernel for the stride-1 and upsampling convs
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
//   * FAST units (chunks 0 .. nmain - 2 of every conv with the consumer-side GroupNorm, i.e. nearly all of them;
//     unit_fast below): the staging instructions sit BETWEEN the three taps' MFMAs in one straight-line stream, so a
//     wave's unit time is max(MFMA, everything else) instead of their sum; TAIL units (the last chunk of a conv
//     without 1x1-skip chunks) are MFMA streams.  The staggered form remains for the last chunk with skip chunks
//     behind it, the skip chunks themselves and convs without an input norm.
//   * Under-filled launches are cut into twice the workgroups (HX2P_PAIRN_HALF, HX2P_FOUR_WAVES: see CFG below).
//
// Tried before this and rejected (tools/experiments/conv_mfma_hx2_prodcons.hip, numbers in DESIGN.md): four dedicated
// producer waves (one per SIMD) beside eight MFMA waves, weights and raw activations by LDS-DMA.  A lone wave is
// latency-bound on everything it does there -- ~280 cycles per global_load_lds issue, ~800 per LDS round trip, 8.5
// cycles per VALU instruction beside two MFMA waves -- so the eight consumers waited for it at every barrier.
//
// The stride-2 / transposed modes stay on conv_mfma_hx2_kernel.  An external scale/shift array (ConvArgs::ab without
// gn_stats0) is copied into the same LDS table the consumer-side norm fills (round 3).
#include <stdlib.h>

#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {


// CFG: how a workgroup is cut.  HX2P_TWO_TILES: 8 waves = two 256-pixel tiles x one channel group (32 NT channels), the
// weights are staged once for both tiles; HX2P_PAIRN: 8 waves = one tile x two channel groups (Cout % 128 == 0), the
// halo is staged once for both groups; HX2P_FOUR_WAVES: 4 waves = one tile x one group, 256 threads and at most 80 KB
// of LDS, so that TWO workgroups share a CU and one's prologue / epilogue (memory round trips, every CU at once)
// could run under the other's K loop -- measured: it does not, so it is used for what it is good at:
// under-filled launches, where one tile per workgroup means twice the workgroups.
// HX2P_PAIRN_HALF: as HX2P_PAIRN with 32-channel groups (NT = 1): one tile x 64 of a 128-channel weight block, twice the
// workgroups -- for launches that would otherwise leave CUs without a workgroup (the 8x8 level, small batches).
// (Chunk-sized units for that cut -- nine taps between two barriers -- were measured 2 ... 9 % slower on every under-filled
// launch: a unit's time is proportional to the work in it, not to the number of barriers.  tools/kbench/variants/ has them.)
enum { HX2P_TWO_TILES = 0, HX2P_PAIRN = 1, HX2P_FOUR_WAVES = 2, HX2P_PAIRN_HALF = 3 };

// POUT: the instantiations that can ALSO write the output in P format (ConvArgs::pout) -- separate ones, so that the block
// costs the others nothing (in the common template it pushed the 255-register two-tile kernel into scratch)
template <int NT, int MODE, int CFG, bool POUT = false>
__global__ __launch_bounds__(CFG == HX2P_FOUR_WAVES ? 256 : 512, 2) void conv_mfma_hx2p_kernel(const ConvArgs a, const int num_tiles) {
  constexpr bool PAIRN = CFG == HX2P_PAIRN || CFG == HX2P_PAIRN_HALF;
  constexpr bool HALF = CFG == HX2P_PAIRN_HALF;  // the packed weight blocks hold 128 channels, this workgroup takes 64 of them
  // CONV_T2 (ConvTranspose2d(4, 2, 1) -- also the form the U-Net's Upsample convs take, rgfm_host.h): blockIdx.z is the
  // output's parity class (py, px); its four taps read halo rows py, py + 1 and columns px, px + 1 of the INPUT raster's
  // halo: units of two taps (one kernel row), two units per chunk
  constexpr bool T2 = MODE == CONV_T2;
  constexpr int TPU = T2 ? 2 : 3;              // taps per unit
  constexpr int UPC = T2 ? 2 : 3;              // units per main chunk
  static_assert(!HALF || NT == 1, "HX2P_PAIRN_HALF: 2 groups x 32 channels");
  constexpr bool W4 = CFG == HX2P_FOUR_WAVES;
  constexpr int NTHR = W4 ? 256 : 512;

  static_assert(MODE == CONV_S1 || MODE == CONV_UP2 || MODE == CONV_T2, "stride-2 convs run on conv_mfma_hx2s_kernel / conv_mfma_hx2_kernel");
  constexpr int NG = PAIRN ? 2 : 1;            // channel groups per block
  constexpr int NA = (PAIRN || W4) ? 1 : 2;    // pixel tiles per block
  constexpr int NBLK = 32 * NT;                // channels per group
  constexpr int NBT = NBLK * NG;               // channels per block
  constexpr int TAPB = NBT * HRW;              // bytes of one tap's weight slab
  constexpr int UB = TPU * TAPB;               // weights of one unit
  constexpr int NB = (UB / 16 + NTHR - 1) / NTHR;    // 16-byte weight items per thread and unit
  constexpr int MAXIT = (NA * 448 * 4 + NTHR - 1) / NTHR;  // halo items (pixel, 4 channels) per thread and chunk
  extern __shared__ __attribute__((aligned(16))) char smemp[];
  const int abytes = (NA * a.halo_px + 1) * HRW;  // one halo buffer (+ a pad record: the store target of lanes past the halo)
  char* const sB = smemp + 2 * abytes;         // two unit-sized weight buffers
  float* const sTab = reinterpret_cast<float*>(sB + 2 * UB);  // [NA * spt][cin][2] S_A x (scale, shift) + a zero row
  // per 16-channel chunk: {source pointer of the chunk's first channel (lo, hi), channel stride of that source, -}
  const bool has_tab = a.gn_stats0 != nullptr || a.ab != nullptr;  // a scale/shift table: from the statistics, or from an external array
  char* const sDesc = reinterpret_cast<char*>(sTab) + (has_tab ? (size_t)(NA * a.g.spt + 1) * (a.C0 + a.C1) * 8 : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = W4 ? 0 : wave >> 2, seg = wave & 3;
  // which half of the staggered pair a wave is (0: stages before its MFMAs, 1: after): its wave group, or -- four
  // waves: the partner on the SIMD belongs to another workgroup -- the parity of the workgroup
  const int role = W4 ? (int)(blockIdx.x & 1) : grp;
  const int l31p = lane & 31, hp_ = lane >> 5;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;
  // rotation key of a halo record's four 16-byte slots: its halo column >> swz.  >> 2 keeps the fragment reads of 32- and
  // 64-wide rasters free of bank conflicts but leaves those of 16- / 8- / 7-wide ones two- to three-way conflicted (a
  // ds_read_b128 phase of 16 lanes then covers two or three raster rows); >> 1 is conflict-free there (tools/lds_swizzle.py)
  const int swz = (W == 16 || W <= 8) ? 1 : 2;

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
  tile_origin(NA == 1 ? (int)blockIdx.x : (int)blockIdx.x * 2, tb0_[0], trow0_[0]);
  tile_origin(NA == 1 ? (int)blockIdx.x : (int)blockIdx.x * 2 + 1, tb0_[1], trow0_[1]);
  const int ga_w = NA == 1 ? 0 : grp;
  const int my_tile = NA == 1 ? (int)blockIdx.x : (int)blockIdx.x * 2 + grp;
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
    arec[mt] = ga_w * a.halo_px + (s * HR + r) * WR + x;
    // byte offset of this lane's fragment of tap (ky = 0, kx) inside a halo buffer, plane h / l: in THIS kernel the
    // four 16-byte slots of a halo record are swizzled with its halo column, (x >> swz) & 3 -- 16 consecutive
    // columns still cover all 16 slot columns of the bank row, and the term no longer depends on the kernel row,
    // so the six offsets are computed once per block instead of per tap
#pragma unroll
    for (int kxi = 0; kxi < 3; ++kxi) {
      const int kx = T2 ? px + kxi : kxi;  // (T2: entry tx is halo column px + tx; entry 2 is not used)
      const int sw = ((x + kx) >> swz) & 3;
      aofs[mt][kxi][0] = (arec[mt] + kx) * HRW + ((hp_ ^ sw) & 3) * 16;
      aofs[mt][kxi][1] = (arec[mt] + kx) * HRW + (((2 + hp_) ^ sw) & 3) * 16;
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
  // ---- GroupNorm prologue, part 1: which (row, group, channel) this lane reduces, and the request for the first
  // channel's partials -- they fly during the accumulator set-up and the item decode (part 2 is further down)
  constexpr int NW = NTHR / 64;
  const int gn_cin = a.C0 + a.C1;
  const int gn_rows = NA * g.spt;                       // 1, 2, 4 or 8 (<= NW)
  const int gn_cpg = gn_cin >> 3;
  // log2 of the waves per row (all powers of two): as many as it takes to give every lane ONE channel (a round trip
  // per channel is what the time goes into), never more -- measured: with 8-channel groups the one-wave-per-row
  // form is 3-5 % faster per layer (fewer waves fetching and reducing in fp64), at 32-channel groups four waves
  // per row are 3 % faster
  int gn_wsh = (31 - __builtin_clz(NW)) - (31 - __builtin_clz(gn_rows));
  {
    const int need = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : (gn_cpg <= 32 ? 2 : 3));
    gn_wsh = gn_wsh < need ? gn_wsh : need;
  }
  const int gn_row = wave >> gn_wsh, gn_part = wave & ((1 << gn_wsh) - 1);
  const bool gn_active = gn_row < gn_rows;  // (wave-uniform: the waves past the last row sit this out)
  const int gn_lpg = 8 << gn_wsh;                       // lanes per group = 64 / (8 / WPR)
  const int gn_gl = lane >> (3 + gn_wsh), gn_sub = lane & (gn_lpg - 1);
  const int gn_gi = gn_part * (8 >> gn_wsh) + gn_gl;    // group of this lane
  const int gn_kmax = (gn_cpg + gn_lpg - 1) / gn_lpg;   // channels per lane (wave-uniform, <= 4)
  const int gn_b = ((g.spt == 1 ? gn_row : (gn_row >> 2)) ? tb0_[1] : tb0_[0]) + (g.spt == 1 ? 0 : (gn_row & 3));
  const bool gn_bok = gn_active && gn_b < a.B;
  float2 gn_v[16];
  float gn_gv = 0.f, gn_bv = 0.f;
  int gn_npt = 0;
  auto gn_fetch = [&](int k) {
    const int c = gn_gi * gn_cpg + gn_sub + gn_lpg * k;
    const bool have = gn_bok && gn_sub + gn_lpg * k < gn_cpg;
    const bool first = !have || c < a.C0;  // (no k-th channel: entry 0 of the first source, never used)
    const float* st = first ? a.gn_stats0 : a.gn_stats1;
    const int cs = first ? a.C0 : a.C1, cc = have ? (first ? c : c - a.C0) : 0;
    const int npt = first ? a.gn_nparts0 : a.gn_g.nparts;
    const size_t bb = gn_bok ? (size_t)gn_b : 0;
#pragma unroll
    for (int p = 0; p < 16; ++p)
      gn_v[p] = *reinterpret_cast<const float2*>(st + ((bb * npt + (p < npt ? p : 0)) * cs + cc) * 2);
    gn_gv = a.gn_gamma[have ? c : 0], gn_bv = a.gn_beta[have ? c : 0];
    gn_npt = npt;
  };
  f32x16 acc[2][NT];
  // bias (+ skip bias + time embedding) and, for an identity residual, the RAW residual values go into the
  // accumulators here; they are scaled by q only after the pipeline fill (finish_acc below), so that the 64 residual
  // loads of a lane are in flight during the item decode, the GroupNorm table and the fill instead of being waited
  // for right here
  float add0[NT];
  {
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
    }
  }
  auto finish_acc = [&]() {
    if (a.res_mode == 1) {
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
  };

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
      const int it = tid + NTHR * j;
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
        adst[j] = (int)__umul24((unsigned)rec, HRW) + ((((q4 >> 1) ^ (hx >> swz)) & 3) * 16) + (q4 & 1) * 8;  // plane l: ^ 32
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
  const int G = UPC * nmain + nskip;                            // units
  // packed weights: [channel block][chunk][tap] slabs of TAPS bytes (TAPS = TAPB, or 2 TAPB when this workgroup takes
  // half of a 128-channel block: then a tap's slab holds this half at offset whalf)
  constexpr int TAPS = HALF ? 2 * TAPB : TAPB;
  constexpr int UBS = TPU * TAPS;
  const int wblk = HALF ? (int)blockIdx.y >> 1 : (int)blockIdx.y;
  const int whalf = HALF ? ((int)blockIdx.y & 1) * TAPB : 0;
  // (T2: [parity class][channel block][chunk][4 taps])
  const int wblocks = HALF ? (int)gridDim.y >> 1 : (int)gridDim.y;
  const char* wpk = reinterpret_cast<const char*>(a.wpkh) + ((size_t)(T2 ? pc * wblocks : 0) + wblk) * nmain * (UPC * TPU) * TAPS + whalf;
  const char* wsk = reinterpret_cast<const char*>(a.wskiph) + (size_t)wblk * nskip * TAPS + whalf;

  f32x4 ra[MAXIT], rb[NB];
  // byte offset of this thread's j-th 16-byte weight item of a 3-tap unit; threads past the unit's last item repeat an
  // earlier one (same bytes to the same address), so the fast path below copies without per-lane predicates
  int boff[NB], soff[NB];  // (LDS image offset, source offset inside the unit's three slabs)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int it = tid + NTHR * j;
    boff[j] = (it < UB / 16 ? it : it - UB / 16) * 16;
    soff[j] = HALF ? (boff[j] / TAPB) * TAPS + (boff[j] % TAPB) : boff[j];
  }
  float hmax = 0.f;  // range flag: the largest |S_A a| this thread has split (two v_max3_f32 per item)
  const int nitems = (NA * nA + NTHR - 1) / NTHR;  // items that exist for at least one thread (block-uniform)

  // raw fp32 fetch of item j of a chunk whose descriptor is d (chunk_desc)
  typedef unsigned hx_u32x4 __attribute__((ext_vector_type(4)));
  auto chunk_desc = [&](int ch) { return *reinterpret_cast<const hx_u32x4*>(sDesc + ch * 16); };
  auto issue_a = [&](const hx_u32x4& d, int j) {
    const float* src = reinterpret_cast<const float*>(((unsigned long long)d.y << 32) | (unsigned long long)d.x);
    ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j], d.z) + (unsigned)(q4 * 4)));
  };
  // scale/shift pairs of item j's four channels in chunk ch (read ahead of the stores of a staging phase)
  auto table_a = [&](int ch, int j, f32x4& e0, f32x4& e1) {
    const char* ep = reinterpret_cast<const char*>(sTab) + ch * (KC * 8) + trow[j];
    e0 = *reinterpret_cast<const f32x4*>(ep);
    e1 = *reinterpret_cast<const f32x4*>(ep + 16);
  };
  // GroupNorm + SiLU + split + store of item j of chunk ch into halo buffer ch & 1 (branch-free per lane)
  auto commit_a = [&](int ch, int j, bool xform, const f32x4& e0, const f32x4& e1) {
    f32x4 v = ra[j];
    if (xform) {
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
    hmax = hx_absmax3(v.x, v.y, hmax);
    hmax = hx_absmax3(v.z, v.w, hmax);
    const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
    char* base = smemp + (ch & 1) * abytes;
    *reinterpret_cast<hx_u32x2*>(base + adst[j]) = ph;
    *reinterpret_cast<hx_u32x2*>(base + (adst[j] ^ 32)) = pl;
  };
  // weights of unit gg: the packed image is the LDS byte image, a linear 16-byte copy (a skip unit is one tap)
  auto issue_b = [&](int gg) {
    const bool main = gg < UPC * nmain;
    const char* src = main ? wpk + (size_t)gg * UBS : wsk + (size_t)(gg - UPC * nmain) * TAPS;
    const int nit = main ? UB / 16 : TAPB / 16;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + NTHR * j;
      rb[j] = *(const hx_gf32x4*)(src + (main ? soff[j] : (it < nit ? it : 0) * 16));  // (main: as the fast unit fetches)
    }
  };
  auto commit_b = [&](int gg) {
    const int nit = gg < UPC * nmain ? UB / 16 : TAPB / 16;
    char* dst = sB + (gg & 1) * UB;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + NTHR * j;
      if (it < nit) *reinterpret_cast<f32x4*>(dst + it * 16) = rb[j];
    }
  };

  // ---- pipeline fill, part 1: the raw halo of chunk 0 (always the first channels of in0) and the weights of unit 0 are
  // requested HERE, ahead of the GroupNorm table: their round trip runs under the table's (statistics fetch, fp64
  // reduction, barrier) instead of behind it
  // (the cuts whose 64-channel wave tiles leave no registers for requests in flight across the table -- two tiles or four
  // waves with NT = 2: the compiler spills them -- request behind the table's barrier as before)
  constexpr bool EARLY = PAIRN || NT == 1;
  const unsigned long long in0_bits = reinterpret_cast<unsigned long long>(a.in0);
  const hx_u32x4 d0 = {(unsigned)in0_bits, (unsigned)(in0_bits >> 32), (unsigned)a.C0, 0u};
  if constexpr (EARLY) {
#pragma unroll
    for (int j = 0; j < MAXIT; ++j)
      if (j < nitems) issue_a(d0, j);
    issue_b(0);
  }

  if (a.gn_stats0 && gn_active) {
    // ---- consumer-side GroupNorm: scale/shift of this block's sample(s) from the producers' partial statistics.
    // Up to all waves take part: the block's R = NA * spt table rows (one per sample slot) are split over the NW waves,
    // WPR = NW / R waves per row, each wave GPW = 8 / WPR of the row's 8 groups, LPG = 64 / GPW lanes per group; a lane
    // strides over its group's channels (sub, sub + LPG, ...), fetches the <= 16 partials of a channel in one
    // round trip (the first channel's were requested at the top of the kernel: gn_fetch(0)) and reduces
    //   N = sum n_p, S1 = sum n_p mean_p, S2 = sum [M2_p + n_p mean_p^2]  ->  mean = S1 / N, var = S2 / N - mean^2
    // in fp64 without divisions in the loop; an LPG-lane butterfly gives the group's sums.
    float gam[4], bet[4];
    double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 1
    for (int k = 0; k < gn_kmax; ++k) {
      gn_fetch(k);
      if (k == 0) gam[0] = gn_gv, bet[0] = gn_bv;
      else if (k == 1) gam[1] = gn_gv, bet[1] = gn_bv;
      else if (k == 2) gam[2] = gn_gv, bet[2] = gn_bv;
      else gam[3] = gn_gv, bet[3] = gn_bv;
      const bool have = gn_bok && gn_sub + gn_lpg * k < gn_cpg;
      const int npt = gn_npt;
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
        *reinterpret_cast<float2*>(sTab + ((size_t)gn_row * cin + gn_gi * gn_cpg + gn_sub + gn_lpg * k) * 2) = o;
      }
    }
    // (visible to every wave after the barrier that opens commit(0))
  }

  if (!a.gn_stats0 && a.ab) {
    // external scale/shift array ab[B][cin][2] (RGFM_GN=table; the ratio encoders' "SiLU on load" identity pairs):
    // the same table, S_A x (scale, shift) of every sample slot of this block, copied instead of derived
    const int cin_t = a.C0 + a.C1, nrow_t = NA * g.spt;
    for (int i = tid; i < nrow_t * cin_t; i += NTHR) {
      const int row = i / cin_t, c = i - row * cin_t;
      const int b = ((g.spt == 1 ? row : (row >> 2)) ? tb0_[1] : tb0_[0]) + (g.spt == 1 ? 0 : (row & 3));
      float2 o = {0.f, 0.f};
      if (b < a.B) {
        const float2 e = *reinterpret_cast<const float2*>(a.ab + ((size_t)b * cin_t + c) * 2);
        o.x = HX_SA * e.x, o.y = HX_SA * e.y;
      }
      *reinterpret_cast<float2*>(sTab + (size_t)i * 2) = o;
    }
  }
  if (has_tab)  // the all-zero row of the padding items
    for (int i = tid; i < 2 * (a.C0 + a.C1); i += NTHR) sTab[nrows_tab * (a.C0 + a.C1) * 2 + i] = 0.f;
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
    const hx_u32x4 d = {(unsigned)pv, (unsigned)(pv >> 32), (unsigned)cs, 0u};
    *reinterpret_cast<hx_u32x4*>(sDesc + tid * 16) = d;
  }
  __syncthreads();  // the scale/shift table and the chunk descriptors are complete
  const bool gn_on = has_tab;
  // ---- pipeline fill, part 2: halo of chunk 0 and weights of unit 0 into LDS, raw halo of chunk 1 and weights of unit 1 in registers
  {
    if constexpr (!EARLY) {
#pragma unroll
      for (int j = 0; j < MAXIT; ++j)
        if (j < nitems) issue_a(d0, j);
      issue_b(0);
    }
#pragma unroll
    for (int j = 0; j < MAXIT; ++j)
      if (j < nitems) {
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
        const bool xf = gn_on && 0 < nmain;
        if (xf) table_a(0, j, e0, e1);
        commit_a(0, j, xf, e0, e1);
      }
    commit_b(0);
    if (ntot > 1) {
      const hx_u32x4 d1 = chunk_desc(1);
#pragma unroll
      for (int j = 0; j < MAXIT; ++j)
        if (j < nitems) issue_a(d1, j);
    }
    issue_b(1);
  }
  finish_acc();
  __syncthreads();

  // one tap (kernel column KX of the row at byte offset `rowoff`): fragments of this wave's 2 pixel tiles x NT channel
  // tiles, 3 MFMAs per tile pair
  auto tap = [&](const char* sArow, const char* sBt, auto kx_tag) {
    constexpr int KX = decltype(kx_tag)::value;
    f16x8 af[2][2], bf[NT][2];
    {
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
    // Order: (1) LDS reads the phase needs (descriptor of chunk c + 2, scale/shift pairs of this unit's items) ahead of
    // every store; (2) every consumer of a fetched register; (3) every new fetch, the halo's first.  hipcc cannot
    // count loads across the loop's back edge and waits for vmcnt(0) before the first use (or re-use) of a register
    // fetched in an earlier unit: with a fresh fetch already in flight that would be a full memory round trip
    // inside every staging phase.
    constexpr int NU = (U < 0) ? MAXIT : (MAXIT + UPC - 1 - U) / UPC;  // items of this unit: j = U, U + UPC, ... (all of them in a one-tap unit)
    const bool have1 = c + 1 < ntot, have2 = c + 2 < ntot;
    const bool xf = gn_on && c + 1 < nmain;
    hx_u32x4 dn = {0u, 0u, 0u, 0u};
    if (have2) dn = chunk_desc(c + 2);
    f32x4 e0[NU], e1[NU];
    if (have1 && xf) {
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int j = (U < 0) ? k : U + UPC * k;
        if (j < nitems) table_a(c + 1, j, e0[k], e1[k]);
      }
    }
    if (gidx + 1 < G) commit_b(gidx + 1);
    if (have1) {
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int j = (U < 0) ? k : U + UPC * k;
        if (j < nitems) commit_a(c + 1, j, xf, e0[k], e1[k]);
      }
    }
    if (have2) {
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int j = (U < 0) ? k : U + UPC * k;
        if (j < nitems) issue_a(dn, j);
      }
    }
    if (gidx + 2 < G) issue_b(gidx + 2);
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using UA = std::integral_constant<int, -1>;

  int gidx = 0;
  // ---- FAST unit (chunks 0 .. nmain - 2 of a conv whose input takes the consumer-side GroupNorm): one straight-line
  // instruction stream per unit in which the staging work sits BETWEEN the three taps' MFMAs instead of in a phase of
  // its own.  A wave's unit time is then max(MFMA time, issue time of everything else) rather than their sum: the
  // vmcnt wait, the table read and the exp / rcp chains of the transform complete under MFMAs that are already in
  // the pipe.  LDS operations keep the order written here (hipcc cannot tell the buffers apart), vector ALU and
  // global loads may move across the segment marks (sched_barrier mask), MFMAs and LDS operations may not.
  //   consumers of fetched registers (weights -> LDS, halo item -> transform -> LDS) come first and the new fetches
  //   right behind them, at ONE point of the unit: every fetch is a full unit old when hipcc's vmcnt(0) meets it.
  auto frag = [&](const char* sArow, const char* sBt, auto kx_tag, f16x8 (&af)[2][2], f16x8 (&bf)[NT][2]) {
    constexpr int KX = decltype(kx_tag)::value;
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
  };
  auto mfma_tap = [&](const f16x8 (&af)[2][2], const f16x8 (&bf)[NT][2]) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[nt][PB[q]], acc[mt][nt], 0, 0, 0);
  };
#define HX2P_SEG() __builtin_amdgcn_sched_barrier(0x0002 | 0x0004 | 0x0010 | 0x0400)  // VALU, SALU, VMEM, transcendentals may cross
  auto unit_fast = [&](int c, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    constexpr int NU = (MAXIT + 2 - U) / 3;
    const char* sArow = smemp + (c & 1) * abytes + U * WR * HRW;
    const char* sBu = sB + (gidx & 1) * UB;
    char* sBn = sB + ((gidx + 1) & 1) * UB;
    const char* wsrc = wpk + (size_t)(gidx + 2) * UBS;
    f16x8 af0[2][2], bf0[NT][2], af1[2][2], bf1[NT][2];
    frag(sArow, sBu, K0{}, af0, bf0);
    const int c2 = c + 2 < ntot ? c + 2 : ntot - 1;  // (no such chunk: a harmless re-fetch, never committed)
    const hx_u32x4 dn = chunk_desc(c2);
    f32x4 e0[NU], e1[NU];
    table_a(c + 1, U, e0[0], e1[0]);
    HX2P_SEG();
    frag(sArow, sBu + TAPB, K1{}, af1, bf1);
    mfma_tap(af0, bf0);
    HX2P_SEG();
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(sBn + boff[j]) = rb[j];
    commit_a(c + 1, U, true, e0[0], e1[0]);
#pragma unroll
    for (int k = 1; k < NU; ++k)
      if (U + 3 * k < nitems) {
        table_a(c + 1, U + 3 * k, e0[k], e1[k]);
        commit_a(c + 1, U + 3 * k, true, e0[k], e1[k]);
      }
#pragma unroll
    for (int k = 0; k < NU; ++k)
      if (k == 0 || U + 3 * k < nitems) issue_a(dn, U + 3 * k);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb[j] = *(const hx_gf32x4*)(wsrc + soff[j]);
    frag(sArow, sBu + 2 * TAPB, K2{}, af0, bf0);
    mfma_tap(af1, bf1);
    HX2P_SEG();
    mfma_tap(af0, bf0);
    __syncthreads();
    ++gidx;
  };
  // the last chunk of a conv without 1x1-skip chunks: nothing is left to stage but the weights of its own units
  // (unit g + 1 while U < 2, fetch of unit g + 2 while U < 1), so the three units are MFMA streams
  auto unit_tail = [&](int c, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    const char* sArow = smemp + (c & 1) * abytes + U * WR * HRW;
    const char* sBu = sB + (gidx & 1) * UB;
    f16x8 af0[2][2], bf0[NT][2], af1[2][2], bf1[NT][2];
    frag(sArow, sBu, K0{}, af0, bf0);
    HX2P_SEG();
    frag(sArow, sBu + TAPB, K1{}, af1, bf1);
    mfma_tap(af0, bf0);
    HX2P_SEG();
    if (U < 2) {
      char* sBn = sB + ((gidx + 1) & 1) * UB;
#pragma unroll
      for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(sBn + boff[j]) = rb[j];
    }
    if (U < 1) {
      const char* wsrc = wpk + (size_t)(gidx + 2) * UBS;
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = *(const hx_gf32x4*)(wsrc + soff[j]);
    }
    frag(sArow, sBu + 2 * TAPB, K2{}, af0, bf0);
    mfma_tap(af1, bf1);
    HX2P_SEG();
    mfma_tap(af0, bf0);
    if (U < 2) __syncthreads();
    ++gidx;
  };
  auto unit = [&](int c, auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    const char* sAc = smemp + (c & 1) * abytes;
    const char* sBu = sB + (gidx & 1) * UB;
    if (role == 0) stage(c, u_tag, gidx);
    const char* sArow = sAc + (T2 ? py + U : U) * WR * HRW;
    tap(sArow, sBu, K0{});
    tap(sArow, sBu + TAPB, K1{});
    if constexpr (!T2) tap(sArow, sBu + 2 * TAPB, K2{});
    if (role != 0) stage(c, u_tag, gidx);
    if (gidx != G - 1) __syncthreads();
    ++gidx;
  };
  int c0 = 0;
  if constexpr (!T2) {
    if (gn_on && nitems >= 3) {
#pragma unroll 1
      for (; c0 < nmain - 1; ++c0) {
        unit_fast(c0, U0{});
        unit_fast(c0, U1{});
        unit_fast(c0, U2{});
      }
      if (nskip == 0) {
        unit_tail(c0, U0{});
        unit_tail(c0, U1{});
        unit_tail(c0, U2{});
        ++c0;
      }
    }
  }
#pragma unroll 1
  for (int c = c0; c < nmain; ++c) {
    unit(c, U0{});
    unit(c, U1{});
    if constexpr (!T2) unit(c, U2{});
  }
  if (nskip) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
    const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
#pragma unroll 1
    for (int c = nmain; c < ntot; ++c, ++gidx) {
      if (role == 0) stage(c, UA{}, gidx);
      tap(smemp + (c & 1) * abytes + WR * HRW, sB + (gidx & 1) * UB, K1{});  // (centre tap: row 1, column 1)
      if (role != 0) stage(c, UA{}, gidx);
      if (gidx != G - 1) __syncthreads();
    }
  }
  {
    const float qinv = nskip ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
    if (!(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);  // (rare) plane h would be >= 32768 (or inf): the host re-runs on bx3
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_pf_kernel)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31 = lane_e & 31, h = lane_e >> 5;
  // folded eval-mode BatchNorm (+ SiLU) of the ratio estimators' encoders, as conv_mfma.hip.  ONCE, in front of the two
  // instantiations of the epilogue: inside them (two copies) it cost the 128-channel instantiations 16-20 bytes of scratch
  if (a.ep_scale) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float es = a.ep_scale[n0 + nt * 32 + l31], eh = a.ep_shift[n0 + nt * 32 + l31];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mt][nt][r] * es + eh;
          acc[mt][nt][r] = a.ep_nosilu ? v : silu_f(v);
        }
    }
  }
  // Two instantiations of the same epilogue: FULL (every pixel of this wave's 64-pixel segment is valid -- all
  // waves of all interior tiles) has no per-element predicates, which are a third of its instructions.
  auto epilogue = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    unsigned vmask[2] = {0u, 0u};
    // ConvArgs::small_check: the output's low range.  For full segments HERE, while nothing but the accumulators is live
    // (behind the stores and the statistics it costs registers: conv_mfma_hx2q.hip)
    if (FULL && a.small_check && a.range_flag) {
      float m = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][nt][r], acc[mt][nt][r + 1], m);
      hx_small_flag(a.range_flag, m);
    }
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
        if (valid && (!POUT || a.out)) {  // (out == null: a P-format-only producer, ConvArgs::pout)
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
    if (!FULL && a.small_check && a.range_flag) {  // (edge tiles: only the valid pixels count)
      float m = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (vmask[mt] & (1u << r)) m = fmaxf(m, fabsf(acc[mt][nt][r]));
      hx_small_flag(a.range_flag, m);
    }
  };
  const bool full_seg = sample_ok && ((g.spt == 1) ? (nvalid - 64 * seg >= 64) : (HW == 64));  // wave-uniform
  if (full_seg) epilogue(std::true_type{});
  else epilogue(std::false_type{});

  // ---------------------------------------------------------------- P format for the consumer's norm (ConvArgs::pout)
  // 16x16 rasters only (the host checks: a tile is one whole sample, 256 pixels = the four segments of the four waves that
  // share this wave's channels): every wave publishes its channels' (mean, M2) over its 64 pixels in LDS, combines the
  // four segments in segment order (fp64, the consumer-side prologue's formula), finishes the group statistics with a
  // butterfly over the group's lanes and writes S_A silu(norm(h)) split into the two fp16 planes (conv_hx2_common.h).
  // A result does not depend on how the launch was cut: the four partials of a channel are the same numbers in every CFG.
  if constexpr (POUT && MODE == CONV_S1) {
    if (a.pout) {  // (kernel-uniform: every wave of every workgroup meets the two barriers)
      __syncthreads();  // all waves are past their last fragment reads: the staging buffers are free
      float* const sx = reinterpret_cast<float*>(smemp);  // [wave][NT][32][2]
      float gam[NT], bet[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        gam[nt] = a.pn_gamma[c], bet[nt] = a.pn_beta[c];
        float sm = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) sm += acc[mt][nt][r];
        sm += __shfl_xor(sm, 32);
        const float mean = sm / 64.f;
        float m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[mt][nt][r] - mean;
            m2 += d * d;
          }
        m2 += __shfl_xor(m2, 32);
        if (h == 0) {
          float2 v;
          v.x = mean, v.y = m2;
          *reinterpret_cast<float2*>(sx + ((wave * NT + nt) * 32 + l31) * 2) = v;
        }
      }
      __syncthreads();
      if (full_seg) {
        const unsigned pstride = (unsigned)a.Cout * 4u;
        float pm = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int sg = 0; sg < 4; ++sg) {
            const float2 v = *reinterpret_cast<const float2*>(sx + (((wave - seg + sg) * NT + nt) * 32 + l31) * 2);
            s1 += 64.0 * (double)v.x;
            s2 += (double)v.y + 64.0 * (double)v.x * (double)v.x;
          }
          float sc, sh;
          hx_group_affine_s(s1, s2, 256.0, a.Cout >> 3, gam[nt], bet[nt], sc, sh);
          const int c = n0 + nt * 32 + l31;
          char* const rec0 = reinterpret_cast<char*>(a.pout) + (pix0 + (size_t)(64 * seg)) * pstride + (size_t)(c >> 4) * 64;
          pm = fmaxf(pm, hx_p_emit(acc[0][nt], acc[1][nt], sc, sh, rec0, pstride, l31, h));
        }
        if (!(pm < HX_BIG)) atomicOr(a.range_flag, 1u);
      }
    }
  }

}

// ---------------------------------------------------------------- host side
static int hx2p_halo(const ConvArgs& a) { return a.g.spt * (a.g.th + 2) * (a.g.W + 2); }
static size_t hx2p_lds_bytes(const ConvArgs& a, int cfg, int mode) {
  const bool halfc = cfg == HX2P_PAIRN_HALF;
  const int nt = halfc ? 1 : ((a.Cout % 64 == 0) ? 2 : 1);
  const int na = cfg == HX2P_TWO_TILES ? 2 : 1, nbt = 32 * nt * ((cfg == HX2P_PAIRN || halfc) ? 2 : 1);
  const int tpu = mode == CONV_T2 ? 2 : 3;
  size_t bytes = (size_t)2 * (na * hx2p_halo(a) + 1) * HRW + (size_t)2 * tpu * nbt * HRW;  // two halo buffers (+ pad record) + two weight units
  if (a.gn_stats0 || a.ab) bytes += (size_t)(na * a.g.spt + 1) * (a.C0 + a.C1) * 2 * sizeof(float);  // scale/shift table + the zero row
  bytes += (size_t)(a.C0 + a.C1 + (a.res_mode == 2 ? a.R0 + a.R1 : 0));                    // 16 bytes per 16-channel chunk: descriptors
  return bytes;
}
// Under-filled launches (fewer workgroups than CUs: the 8x8 level, the MC pre-phase, the per-rank shapes of a
// multi-GPU run) are cut finer: 128-channel workgroups into two 64-channel ones (HX2P_PAIRN_HALF), two-tile workgroups
// into two four-wave ones (HX2P_FOUR_WAVES) -- a workgroup's time is set by its serial chain of round trips, not by
// its MFMA count, so twice the workgroups on idle CUs is up to twice the rate (measured -15..-34 % per launch).
// On FULL launches both are slower (four-wave workgroups: a four-wave workgroup takes as long as an eight-wave one
// co-resident or not, whole bench 438 vs 450 img/s; 64-channel workgroups: +20 % time), so only below g_hx2p_half.
// tools/kbench: RGFM_HX2P_W4 = 1 / 2 forces four-wave workgroups (layers with Cout % 128 != 0 / every layer whose
// LDS need allows two workgroups per CU), RGFM_HX2P_HALF sets the threshold.
static int g_hx2p_w4 = 0;
void conv_hx2p_set_w4(int v) { g_hx2p_w4 = v; }
// launches with fewer workgroups than this are cut finer (0: never); the CU count
static int g_hx2p_half = 256;
void conv_hx2p_set_half(int v) { g_hx2p_half = v; }
static int hx2p_cfg(const ConvArgs& a, int mode) {
  const bool fits = hx2p_lds_bytes(a, HX2P_FOUR_WAVES, mode) <= 80 * 1024;  // two workgroups per CU
  if (g_hx2p_w4 == 2 && fits) return HX2P_FOUR_WAVES;
  if (g_hx2p_w4 == 1 && fits && a.Cout % 128 != 0) return HX2P_FOUR_WAVES;
  const int tiles = geom_num_tiles(a.g, a.B) * (mode == CONV_T2 ? 4 : 1);  // (workgroups: a parity class each)
  if (a.Cout % 128 != 0) {
    const int wgs = ((tiles + 1) / 2) * (a.Cout / (32 * ((a.Cout % 64 == 0) ? 2 : 1)));
    return (g_hx2p_half && wgs < g_hx2p_half) ? HX2P_FOUR_WAVES : HX2P_TWO_TILES;
  }
  const int wgs = tiles * (a.Cout / 128);
  if (!(g_hx2p_half && wgs < g_hx2p_half)) return HX2P_PAIRN;
  return HX2P_PAIRN_HALF;
}

// the pipelined kernel takes: stride-1 / upsampling convs whose input norm (if any) is the consumer-side one
bool conv_hx2p_supported(const ConvArgs& a, int mode) {
  if (mode != CONV_S1 && mode != CONV_UP2 && mode != CONV_T2) return false;
  if (mode == CONV_T2 && (a.res_mode != 0 || a.pout)) return false;
  if (!conv_hx2_supported(a, mode)) return false;
  if (a.gn_stats0 && !conv_hx2_gn_supported(a, mode)) return false;
  return hx2p_halo(a) <= 448 && hx2p_lds_bytes(a, hx2p_cfg(a, mode), mode) <= 160 * 1024;
}

int conv_hx2p_init() {
  int rc = 0;
#define RAISEP(NTV, M, P) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2p_kernel<NTV, M, P>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISEP(1, CONV_S1, HX2P_TWO_TILES); RAISEP(1, CONV_UP2, HX2P_TWO_TILES); RAISEP(1, CONV_T2, HX2P_TWO_TILES);
  RAISEP(2, CONV_S1, HX2P_TWO_TILES); RAISEP(2, CONV_UP2, HX2P_TWO_TILES); RAISEP(2, CONV_T2, HX2P_TWO_TILES);
  RAISEP(2, CONV_S1, HX2P_PAIRN); RAISEP(2, CONV_UP2, HX2P_PAIRN); RAISEP(2, CONV_T2, HX2P_PAIRN);
  RAISEP(1, CONV_S1, HX2P_PAIRN_HALF); RAISEP(1, CONV_UP2, HX2P_PAIRN_HALF); RAISEP(1, CONV_T2, HX2P_PAIRN_HALF);
  RAISEP(1, CONV_S1, HX2P_FOUR_WAVES); RAISEP(1, CONV_UP2, HX2P_FOUR_WAVES); RAISEP(1, CONV_T2, HX2P_FOUR_WAVES);
  RAISEP(2, CONV_S1, HX2P_FOUR_WAVES); RAISEP(2, CONV_UP2, HX2P_FOUR_WAVES); RAISEP(2, CONV_T2, HX2P_FOUR_WAVES);
#undef RAISEP
#define RAISEPP(NTV, P) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2p_kernel<NTV, CONV_S1, P, true>), \
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISEPP(2, HX2P_PAIRN); RAISEPP(1, HX2P_PAIRN_HALF); RAISEPP(2, HX2P_TWO_TILES); RAISEPP(2, HX2P_FOUR_WAVES);
#undef RAISEPP
  return rc;
}

void launch_conv_hx2p(const ConvArgs& a_in, int mode, hipStream_t s) {
  ConvArgs a = a_in;
  a.halo_px = hx2p_halo(a_in);
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int tiles = geom_num_tiles(a.g, a.B);
  const int cfg = hx2p_cfg(a, mode);
  if (cfg == HX2P_PAIRN_HALF) a.fin_expected *= 2;  // (producer-side finalize: twice the 4-wave groups along the channels arrive)
  dim3 grid(cfg == HX2P_TWO_TILES ? (tiles + 1) / 2 : tiles,
            cfg == HX2P_PAIRN ? a.Cout / 128 : (cfg == HX2P_PAIRN_HALF ? a.Cout / 64 : a.Cout / (32 * nt)), mode == CONV_T2 ? 4 : 1);
  const size_t lds = hx2p_lds_bytes(a, cfg, mode);
#define LAUNCHP(NTV, M, P) hipLaunchKernelGGL((conv_mfma_hx2p_kernel<NTV, M, P>), grid, dim3(P == HX2P_FOUR_WAVES ? 256 : 512), lds, s, a, tiles)
  if (a.pout) {  // (the walk has checked hx2p_pout_supported: 16x16 stride-1, 64 / 128 / 256 output channels)
#define LAUNCHPP(NTV, P) hipLaunchKernelGGL((conv_mfma_hx2p_kernel<NTV, CONV_S1, P, true>), grid, dim3(P == HX2P_FOUR_WAVES ? 256 : 512), lds, s, a, tiles)
    if (cfg == HX2P_PAIRN) LAUNCHPP(2, HX2P_PAIRN);
    else if (cfg == HX2P_PAIRN_HALF) LAUNCHPP(1, HX2P_PAIRN_HALF);
    else if (cfg == HX2P_FOUR_WAVES) LAUNCHPP(2, HX2P_FOUR_WAVES);
    else LAUNCHPP(2, HX2P_TWO_TILES);
#undef LAUNCHPP
    return;
  }
#define LAUNCHM(NTV, P)                           \
  do {                                            \
    if (mode == CONV_S1) LAUNCHP(NTV, CONV_S1, P); \
    else if (mode == CONV_T2) LAUNCHP(NTV, CONV_T2, P); \
    else LAUNCHP(NTV, CONV_UP2, P);               \
  } while (0)
  if (cfg == HX2P_PAIRN) LAUNCHM(2, HX2P_PAIRN);
  else if (cfg == HX2P_PAIRN_HALF) LAUNCHM(1, HX2P_PAIRN_HALF);
  else if (cfg == HX2P_FOUR_WAVES) {
    if (nt == 2) LAUNCHM(2, HX2P_FOUR_WAVES);
    else LAUNCHM(1, HX2P_FOUR_WAVES);
  }
  else {
    if (nt == 2) LAUNCHM(2, HX2P_TWO_TILES);
    else LAUNCHM(1, HX2P_TWO_TILES);
  }
#undef LAUNCHM
#undef LAUNCHP
}

}  // namespace rgfm
