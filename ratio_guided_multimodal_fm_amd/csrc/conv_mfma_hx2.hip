// conv_mfma_hx2.hip -- the implicit-GEMM convolution of conv_mfma_bx3.hip with fp32 operands carried as
// TWO fp16 planes and the product formed on the f16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate):
//     a' = S_A a,  a' ~= a_h + a_l,  a_h = f16(a'), a_l = f16(a' - a_h)      (round-to-nearest-even both times)
//     w' = s_w w,  w' ~= w_h + w_l   (s_w: a per-conv power of two that puts max|w| in [2^13, 2^14))
//     a'w' ~= a_l*w_h + a_h*w_l + a_h*w_h                                     (3 MFMAs, small terms first)
// Each plane carries 11 significand bits plus the sign of the residual, so a_h + a_l represents a' to
// 2^-24 |a'| (half an fp32 ulp) as long as a_l is a normal fp16, i.e. |a'| >= 2^-2; below that the
// absolute error is <= 2^-25 (|a| error <= 2^-29 at S_A = 16).  Every f16 x f16 product is exact in fp32
// and the dropped term a_l*w_l is < 2^-24 |a'w'|: per product the error is <= 3 * 2^-24 relative -- the
// class of conv_mfma_bx3.hip's 2^-23 -- at HALF its MFMA count.  The accumulators hold q = S_A s_w times
// the true sums (bias / time embedding / residual enter multiplied by q; the epilogue multiplies by 1/q;
// powers of two, exact).  Range: fp16 overflows at 65504, so |activation| must stay below 2048 (S_A = 16;
// GroupNorm+SiLU outputs are O(10)); the staging path raises ConvArgs::range_flag when it sees
// |a'| >= 32768 and the host re-runs the call on the split-bf16 kernel (fp32 range); convs whose
// weights fall outside [2^-40, 2^40] are never routed here (launch_pack_conv_hx2 reports it).
//
// Same fusion set, tiling, prologue (consumer-side GroupNorm) and epilogue as conv_mfma_bx3w_kernel.
// LDS records are 2 planes x 16 fp16 = 64 B; the four 16-byte slots of a record are XOR-swizzled with
// (record >> 2) & 3, so 16 consecutive records cover all 16 slot columns of the 256-byte bank row
// (conflict-free ds_read_b128).  DESIGN.md section 4 has the measurements.
#include <stdlib.h>

#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

// ------------------------------------------------------------------------------------
// Workgroup = 512 threads = 2 waves per SIMD (one wave alone cannot keep the 16-bit MFMA pipe issuing
// back to back), built as TWO of conv_mfma_pf_kernel's 4-wave tiles sharing one operand in LDS:
//   PAIRN  (Cout % (64 NT) == 0): one 256-pixel tile x two adjacent 32NT-channel groups -- the
//          activation halo (the expensive GroupNorm+SiLU+split staging) is staged once for both;
//   !PAIRN: two consecutive 256-pixel tiles x one channel group -- the weights are staged once.
// LDS records are unpadded (64 B: two planes x 16 fp16) with their four 16-byte slots XOR-swizzled
// (conv_hx2_common.h: hswz), which keeps ds_read_b128 conflict-free for 16 consecutive records.
// The stride-1 / upsampling convs run on conv_mfma_hx2p.hip's pipelined version of this kernel; this one
// keeps the stride-2 and transposed modes and the external scale/shift array path.
// ------------------------------------------------------------------------------------
#ifdef RGFM_HX2_PROF
__device__ unsigned long long g_hx2_prof[10];  // prologue, issue, mfma, commit-wait, commit-A, commit-B, epilogue, blocks, [8] shader clk, [9] 100 MHz ticks
#define PROF_T(var) const long long var = clock64()
#define PROF_ADD(slot, t0, t1) prof_acc[slot] += (t1) - (t0)
#else
#define PROF_T(var)
#define PROF_ADD(slot, t0, t1)
#endif

template <int NT, int MODE, bool PAIRN>
__global__ __launch_bounds__(512, 2) void conv_mfma_hx2_kernel(const ConvArgs a, const int num_tiles) {
  // CONV_S2 (3x3, stride 2, pad 1): the input is read as its four pixel-parity phases (a, b) = (row & 1, col & 1),
  // each a plane of the OUTPUT's size; output (r, x) takes phase (a, b) at plane offsets dr in {-1 (a = 1 only), 0},
  // dx likewise, i.e. 1 / 2 / 2 / 4 taps for phases (0,0) / (0,1) / (1,0) / (1,1) -- 9 in all, nothing multiplied by
  // zero.  The K loop runs over (phase, 16-channel chunk); a halo tile is one phase plane, staged like CONV_S1.
  constexpr int NTAPS = (MODE == CONV_T2 || MODE == CONV_S2) ? 4 : 9;  // weight taps resident in LDS
  constexpr int NG = PAIRN ? 2 : 1;            // channel groups per block
  constexpr int NA = PAIRN ? 1 : 2;            // pixel tiles per block
  constexpr int NBLK = 32 * NT;                // channels per group
  constexpr int NBIG = NTAPS * NBLK * 4;       // 16-byte weight items per chunk and group
  constexpr int NB = (NBIG * NG + 511) / 512;
  constexpr int MAXIT = (NA * 448 * 4 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) char smemh[];
#ifdef RGFM_HX2_PROF
  long long prof_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const long long wall0 = wall_clock64();
#endif
  PROF_T(tp0);
  char* sA = smemh;
  char* sB = smemh + NA * a.halo_px * HRW;
  float* sAB = reinterpret_cast<float*>(sB + NTAPS * NBLK * NG * HRW);  // [NA * spt][16][2]   (external `ab` path)
  float* sG = sAB + 256;                                                // [NA * spt][8][2] group (mean, rstd)
  float* sTab = sG + 128;                                               // [NA * spt][cin][2] (consumer-side GroupNorm)

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

  int arec[2];
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
  }
  int bbase[NT], bsw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int rec = (PAIRN ? grp : 0) * NBLK + nt * 32 + l31p;
    bbase[nt] = rec * HRW;  // + hswz(rec, plane, hp_): NBLK * NG is a multiple of 16, so the swizzle does not depend on the tap
    bsw[nt] = (rec >> 2) & 3;
  }

  const int bw = (g.spt == 1) ? b0 : b0 + seg;
  const bool sample_ok = bw < a.B;
  const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W : (size_t)bw * HW;
  // ConvArgs::in_amax: power-of-two pre-scale of a raw input tensor whose magnitude is far from 1 (the gradients of the
  // ratio estimator's reverse pass): max = f 2^e, f in [0.5, 1) -> the values are staged x 2^-e, the outputs x 2^e
  float in_sc = 1.f, out_sc = 1.f;
  if (a.in_amax) {
    const float m = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)*a.in_amax));
    if (m > 0.f && m < 3.0e38f) {
      int e;
      (void)frexpf(m, &e);
      in_sc = ldexpf(1.f, -e), out_sc = ldexpf(1.f, e);
    }
    in_sc = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(in_sc)));
    out_sc = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(out_sc)));
  }
  const float sa_raw = HX_SA * in_sc;
  const float qmain = a.hq[0] * in_sc;
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
  int poff[MAXIT], adst[MAXIT];
  unsigned okmask = 0u, inmask = 0u, smask = 0u;
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int it = tid + 512 * j;
      poff[j] = 0, adst[j] = 0;
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
        adst[j] = (int)__umul24((unsigned)rec, HRW) + hswz(rec, 0, q4 >> 1) + (q4 & 1) * 8;  // plane l: ^ 32
        inmask |= 1u << j;
        if (ok) {
          poff[j] = (int)__umul24(__umul24((unsigned)b, (unsigned)a.Hin) + (unsigned)y, (unsigned)a.Win) + x;  // < 2^24 pixels
          okmask |= 1u << j;
          smask |= (unsigned)(ga * 4 + s) << (3 * j);
        }
      }
    }
  }
  // scale/shift table slot of this thread (tid < NA * spt * 8): sample and channel pair
  int ab_b = -1;
  if (tid < NA * g.spt * 8) {
    const int slot = tid >> 3;
    const int ga = (g.spt == 1) ? slot : (slot >> 2), s = (g.spt == 1) ? 0 : (slot & 3);
    const int tb0 = ga ? tb0_[1] : tb0_[0];
    if (tb0 + s < a.B) ab_b = tb0 + s;
  }
  const int ab_slot = (NA == 2 && g.spt == 1) ? (tid >> 3) * 4 : (tid >> 3);  // table index ga * 4 + s

  const int cin = a.C0 + a.C1;
  const int nch_in = cin / KC;                                   // 16-channel chunks of the input
  const int nch_main = (MODE == CONV_S2) ? 4 * nch_in : nch_in;  // K chunks (x 4 phases for stride 2)
  const int nch_skip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nch_main + nch_skip;
  const char* wpk3 = reinterpret_cast<const char*>(a.wpkh);
  const char* wskip3 = reinterpret_cast<const char*>(a.wskiph);

  f32x4 ra[MAXIT], rb[NB], rab;
  float amax = 0.f;  // max |a'| this thread has staged (range flag)
  rab = f32x4{1.f, 0.f, 1.f, 0.f};

  auto issue = [&](int ch) {
    const bool skip = ch >= nch_main;
    const float* src;
    int cs, cc, c;
    int ph = 0;  // CONV_S2: phase of the chunk
    if (!skip) {
      if (MODE == CONV_S2) ph = ch / nch_in;
      c = (ch - ph * nch_in) * KC;
      if (c < a.C0) src = a.in0, cs = a.C0, cc = c;
      else src = a.in1, cs = a.C1, cc = c - a.C0;
    } else {
      c = (ch - nch_main) * KC;
      if (c < a.R0) src = a.res0, cs = a.R0, cc = c;
      else src = a.res1, cs = a.R1, cc = c - a.R0;
    }
    {
      const unsigned pshift = (MODE == CONV_S2) ? (unsigned)((ph >> 1) * a.Win + (ph & 1)) : 0u;  // phase pixel offset
#pragma unroll
      for (int j = 0; j < MAXIT; ++j)
        ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j] + pshift, (unsigned)cs) + (unsigned)(cc + q4 * 4)));
    }
    // packed weights: the chunk's LDS image ([tap][NG groups][NBLK channels] records of 96 B, halves
    // pre-swapped) is contiguous in global memory (launch_pack_conv_bx3 with nb = NBLK * NG)
    int nbit = skip ? NBLK * NG * 4 : NBIG * NG;
    const char* w0 = skip ? wskip3 + ((size_t)(blockIdx.y * nch_skip + (ch - nch_main))) * (NBLK * NG * HRW)
                          : wpk3 + ((size_t)((pc * gridDim.y + blockIdx.y) * nch_main + ch) * NTAPS) * (NBLK * NG * HRW);
    if (MODE == CONV_S2) {
      // packed [channel block][phase][chunk][taps of the phase][nb]: 0 / 1 / 3 / 5 taps precede phases 0..3
      const int ntp = ((ph >> 1) + 1) * ((ph & 1) + 1), tbefore = (ph == 0) ? 0 : (ph == 1 ? 1 : (ph == 2 ? 3 : 5));
      nbit = ntp * NBLK * NG * 4;
      w0 = wpk3 + ((size_t)blockIdx.y * 9 * nch_in + (size_t)tbefore * nch_in + (size_t)(ch - ph * nch_in) * ntp) * (NBLK * NG * HRW);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 512 * j;
      rb[j] = *(const hx_gf32x4*)(w0 + (size_t)(it < nbit ? it : 0) * 16);
    }
    if (!skip && a.ab && !a.gn_stats0) {
      const size_t o = ab_b >= 0 ? ((size_t)ab_b * cin + c + 2 * (tid & 7)) * 2 : 0;
      rab = *(const hx_gf32x4*)(a.ab + o);
    }
  };

  auto commit = [&](int ch) {
    const bool skip = ch >= nch_main;
    const int phc = (MODE == CONV_S2 && !skip) ? ch / nch_in : 0;
    const int chc = (ch - phc * nch_in) * KC;  // first channel of the chunk in the concatenated input
    const bool gnk = a.gn_stats0 != nullptr;
    const bool xform = !skip && (a.ab != nullptr || gnk);
    PROF_T(tc0);
    if (xform && !gnk && tid < NA * g.spt * 8) *reinterpret_cast<f32x4*>(sAB + (ab_slot * 8 + (tid & 7)) * 4) = rab;
    __syncthreads();
    PROF_T(tc1);
    PROF_ADD(3, tc0, tc1);
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      if ((inmask >> j) & 1u) {
        f32x4 v = ra[j];
        const bool okj = (okmask >> j) & 1u;
        if (!okj) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xform && okj) {
          const int s = (smask >> (3 * j)) & 7u;
          // scale/shift of the item's 4 channels: this chunk's slice of the block-lifetime table (consumer-side
          // GroupNorm; slot = (s >> 2) * spt + (s & 3)) or the per-chunk sAB copy of the external `ab`
          const float* ep = gnk ? sTab + ((((s >> 2) * g.spt + (s & 3)) * cin + chc + q4 * 4) * 2) : sAB + (s * 16 + q4 * 4) * 2;
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(ep + 4);
          // the table path holds S_A x (scale, shift): z' = S_A z, silu' = z' / (1 + exp(-z' / S_A)) = S_A silu(z)
          const float ks = gnk ? 1.f : HX_SA;
          v.x = silu_scaled(ks * (e0.x * v.x + e0.y));
          v.y = silu_scaled(ks * (e0.z * v.y + e0.w));
          v.z = silu_scaled(ks * (e1.x * v.z + e1.y));
          v.w = silu_scaled(ks * (e1.z * v.w + e1.w));
        } else {
          v = v * sa_raw;
        }
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        unsigned h0, l0, h1, l1;
        hsplit2(v.x, v.y, h0, l0);
        hsplit2(v.z, v.w, h1, l1);
        const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
        char* dst = sA + adst[j];
        *reinterpret_cast<hx_u32x2*>(dst) = ph;
        *reinterpret_cast<hx_u32x2*>(sA + (adst[j] ^ 32)) = pl;
      }
    }
    PROF_T(tc2);
    PROF_ADD(4, tc1, tc2);
    int nbit = skip ? NBLK * NG * 4 : NBIG * NG;
    if (MODE == CONV_S2) nbit = ((phc >> 1) + 1) * ((phc & 1) + 1) * NBLK * NG * 4;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 512 * j;
      if (it < nbit) *reinterpret_cast<f32x4*>(sB + it * 16) = rb[j];
    }
    __syncthreads();
    PROF_T(tc3);
    PROF_ADD(5, tc2, tc3);
  };

  PROF_T(tp1);
  PROF_ADD(0, tp0, tp1);
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
  issue(0);
  commit(0);
  for (int ch = 0; ch < ntot; ++ch) {
    const bool skip = ch >= nch_main;
    if (nch_skip && ch == nch_main) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
      const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
    }
    PROF_T(ti0);
    if (ch + 1 < ntot) issue(ch + 1);
    PROF_T(ti1);
    PROF_ADD(1, ti0, ti1);
    const int phm = (MODE == CONV_S2) ? ch / nch_in : 0, pa = phm >> 1, pb = phm & 1;
    const int tap_lo = skip ? 4 : 0, tap_hi = skip ? 5 : (MODE == CONV_S2 ? (pa + 1) * (pb + 1) : NTAPS);
#pragma unroll 1
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
      int ky, kx;
      if (MODE == CONV_T2) {
        ky = py + (tap >> 1), kx = px + (tap & 1);
      } else if (MODE == CONV_S2) {
        // halo rows are plane rows r - 1 (ky = 0) and r (ky = 1): phase a = 0 uses r only; a = 1 uses r - 1 then r
        const int ty = pb ? (tap >> 1) : tap, tx = pb ? (tap & 1) : 0;
        ky = pa ? ty : 1, kx = pb ? tx : 1;
      } else {
        ky = tap / 3, kx = tap - 3 * ky;
      }
      const int toff = ky * WR + kx;
      const int boff = (skip ? 0 : tap) * (NBLK * NG * HRW);
      f16x8 af[2][2], bf[NT][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int rec = arec[mt] + toff;
        const char* pa = sA + rec * HRW;
        const int o0 = ((hp_ ^ (rec >> 2)) & 3) * 16;  // slot of (plane h, half hp_); plane l: ^ 32
        af[mt][0] = *reinterpret_cast<const f16x8*>(pa + o0);
        af[mt][1] = *reinterpret_cast<const f16x8*>(pa + (o0 ^ 32));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int o0 = ((hp_ ^ bsw[nt]) & 3) * 16;
        bf[nt][0] = *reinterpret_cast<const f16x8*>(sB + bbase[nt] + boff + o0);
        bf[nt][1] = *reinterpret_cast<const f16x8*>(sB + bbase[nt] + boff + (o0 ^ 32));
      }
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[nt][PB[q]], acc[mt][nt], 0, 0, 0);
    }
    PROF_T(tm1);
    PROF_ADD(2, ti1, tm1);
    if (ch + 1 < ntot) commit(ch + 1);
  }
  PROF_T(te0);
  {
    const float qinv = (nch_skip ? a.hq_skip[1] : a.hq[1]) * out_sc;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
    if (!(amax < HX_LIMIT)) atomicOr(a.range_flag, 1u);  // (rare) an activation left the fp16 range: the host re-runs on bx3
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_pf_kernel)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31 = lane_e & 31, h = lane_e >> 5;
  // folded eval-mode BatchNorm (+ SiLU) of the ratio estimators' encoders, as conv_mfma.hip (once, in front of the two
  // instantiations of the epilogue: conv_mfma_hx2p.hip)
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
#ifdef RGFM_HX2_PROF
  PROF_T(te1);
  PROF_ADD(6, te0, te1);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_hx2_prof[i], (unsigned long long)prof_acc[i]);
    atomicAdd(&g_hx2_prof[7], 1ull);
    atomicAdd(&g_hx2_prof[8], (unsigned long long)(te1 - tp0));
    atomicAdd(&g_hx2_prof[9], (unsigned long long)(wall_clock64() - wall0));
  }
#endif
}

// ---------------------------------------------------------------- weight packing (scaled 2-way fp16 split)
// Per-conv scale: hq[0] = q = S_A s_w, hq[1] = 1 / q, hq[2] = s_w, hq[3] = 1 when the conv may run on this
// kernel (finite weights with max|w| in [2^-40, 2^40]; an all-zero tensor counts as in range, s_w = 1).
__global__ void hx2_scale_kernel(const float* w, size_t n, float* hq) {
  __shared__ float red[256];
  float m = 0.f;
  bool bad = false;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = fabsf(w[i]);
    if (!(v <= 3.0e38f)) bad = true;  // inf / nan
    m = fmaxf(m, v);
  }
  red[threadIdx.x] = bad ? __builtin_inff() : m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mx = red[0];
    float sw = 1.f, ok = 1.f;
    if (mx > 0.f) {
      if (!(mx <= 3.0e38f)) {
        ok = 0.f;
      } else {
        const int e = ilogbf(mx);  // 2^e <= mx < 2^(e+1)
        if (e < -40 || e > 40) ok = 0.f;
        else sw = ldexpf(1.f, 13 - e);  // mx s_w in [2^13, 2^14)
      }
    }
    hq[0] = HX_SA * sw, hq[1] = 1.f / (HX_SA * sw), hq[2] = sw, hq[3] = ok;
  }
}

void hx2_scale_launch(const float* w, size_t n, float* hq, hipStream_t s) {  // (for conv_mfma_hx2w.hip's transformed weights)
  hipLaunchKernelGGL(hx2_scale_kernel, dim3(1), dim3(256), 0, s, w, n, hq);
}

__device__ __forceinline__ void hsplit1(float v, unsigned short& h, unsigned short& l) {
  unsigned ph, pl;
  hsplit2(v, 0.f, ph, pl);
  h = (unsigned short)(ph & 0xffffu), l = (unsigned short)(pl & 0xffffu);
}

// position (in fp16 elements) of channel kk of plane p inside the 32-element record `rec` of a weight tile
__device__ __forceinline__ int hrec_pos(int rec, int p, int kk) { return (((2 * p + (kk >> 3)) ^ (rec >> 2)) & 3) * 8 + (kk & 7); }

// [Cout][Cin][taps] fp32 -> [Cout/nb][Cin/16][taps][nb][2][16] fp16, nb = channels of one workgroup
// (hx2_block_channels): each [taps][nb] slab is the byte image of the kernel's LDS weight tile
__global__ void pack_conv_hx2_kernel(const float* w, unsigned short* out, const float* hq, int Cout, int Cin, int taps, int nb) {
  const float sw = hq[2];
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
    const int blk = (int)(r / nch);
    const int co = blk * nb + n, ci = ch * 16 + kk;
    unsigned short h, l;
    hsplit1(sw * w[((size_t)co * Cin + ci) * taps + tap], h, l);
    const int rec = tap * nb + n;  // record index inside the chunk's LDS image
    unsigned short* rp = out + (i / 16) * 32;
    rp[hrec_pos(rec, 0, kk)] = h, rp[hrec_pos(rec, 1, kk)] = l;
  }
}

// ConvTranspose2d [Cin][Cout][4][4] -> [4 parity][Cout/nb][Cin/16][4 taps][nb][2][16] fp16 (see pack_deconv_kernel)
__global__ void pack_deconv_hx2_kernel(const float* w, unsigned short* out, const float* hq, int Cin, int Cout, int nb) {
  const float sw = hq[2];
  const size_t per = (size_t)Cout * Cin * 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 4 * per; i += (size_t)gridDim.x * blockDim.x) {
    const int pc = (int)(i / per);
    size_t r = i - (size_t)pc * per;
    const int kk = r % 16;
    r /= 16;
    const int n = r % nb;
    r /= nb;
    const int tap = r % 4;
    r /= 4;
    const int nch = Cin / 16;
    const int ch = r % nch;
    const int blk = (int)(r / nch);
    const int co = blk * nb + n, ci = ch * 16 + kk;
    const int ky = 3 - (pc >> 1) - 2 * (tap >> 1), kx = 3 - (pc & 1) - 2 * (tap & 1);
    unsigned short h, l;
    hsplit1(sw * w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx], h, l);
    const int rec = tap * nb + n;
    unsigned short* rp = out + (i / 16) * 32;
    rp[hrec_pos(rec, 0, kk)] = h, rp[hrec_pos(rec, 1, kk)] = l;
  }
}

// stride-2 3x3 conv: [Cout][Cin][3][3] -> [Cout/nb][4 phases][Cin/16][taps of the phase][nb][2][16] fp16.
// Phase (a, b), tap (ty, tx): kernel row = a ? 2 * ty : 1 (plane row r - 1 is input row 2r - 1 = kernel row 0,
// plane row r of phase a = 1 is input row 2r + 1 = kernel row 2; phase a = 0 is input row 2r = kernel row 1).
__global__ void pack_conv_hx2_s2_kernel(const float* w, unsigned short* out, const float* hq, int Cout, int Cin, int nb) {
  const float sw = hq[2];
  const int nch = Cin / 16;
  const size_t total = (size_t)Cout * Cin * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = i % 16;
    size_t r = i / 16;                      // record index
    const size_t per_blk = (size_t)9 * nch * nb;
    const int blk = (int)(r / per_blk);
    size_t q = r - (size_t)blk * per_blk;   // record within the channel block
    int ph = 0;
    for (int p = 0; p < 4; ++p) {
      const int ntp = ((p >> 1) + 1) * ((p & 1) + 1);
      if (q < (size_t)ntp * nch * nb) { ph = p; break; }
      q -= (size_t)ntp * nch * nb;
    }
    const int pa = ph >> 1, pb = ph & 1, ntp = (pa + 1) * (pb + 1);
    const int ch = (int)(q / ((size_t)ntp * nb));
    const int rem = (int)(q - (size_t)ch * ntp * nb);
    const int tap = rem / nb, n = rem - tap * nb;
    const int ty = pb ? (tap >> 1) : tap, tx = pb ? (tap & 1) : 0;
    const int kyo = pa ? 2 * ty : 1, kxo = pb ? 2 * tx : 1;
    const int co = blk * nb + n, ci = ch * 16 + kk;
    unsigned short h, l;
    hsplit1(sw * w[((size_t)co * Cin + ci) * 9 + kyo * 3 + kxo], h, l);
    const int rec = tap * nb + n;
    unsigned short* rp = out + r * 32;
    rp[hrec_pos(rec, 0, kk)] = h, rp[hrec_pos(rec, 1, kk)] = l;
  }
}

// channels one workgroup covers: 128 when Cout % 128 == 0 (two 64-channel groups), else 64 or 32
static int hx2_block_channels(int Cout) { return Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32); }

// mode: CONV_S1 (also CONV_UP2 and 1x1: plain [taps] order), CONV_S2 (phase-major), CONV_T2 (w is a ConvTranspose2d weight)
void launch_pack_conv_hx2(const float* w, void* out, float* hq, int Cout, int Cin, int taps, int mode, hipStream_t s) {
  const size_t n = (size_t)Cout * Cin * (mode == CONV_T2 ? 16 : taps);
  hipLaunchKernelGGL(hx2_scale_kernel, dim3(1), dim3(256), 0, s, w, n, hq);
  const int nb = hx2_block_channels(Cout);
  if (mode == CONV_S2) hipLaunchKernelGGL(pack_conv_hx2_s2_kernel, dim3(256), dim3(256), 0, s, w, (unsigned short*)out, hq, Cout, Cin, nb);
  else if (mode == CONV_T2) hipLaunchKernelGGL(pack_deconv_hx2_kernel, dim3(256), dim3(256), 0, s, w, (unsigned short*)out, hq, Cin, Cout, nb);
  else hipLaunchKernelGGL(pack_conv_hx2_kernel, dim3(256), dim3(256), 0, s, w, (unsigned short*)out, hq, Cout, Cin, taps, nb);
}

// halo of one tile in this kernel: (th + 2) x (W + 2) plane pixels per sample in every mode (ConvArgs::halo_px is
// the fp32 kernel's, which differs for stride 2)
static int hx2_halo(const ConvArgs& a) { return a.g.spt * (a.g.th + 2) * (a.g.W + 2); }

static bool hx2_pairn(const ConvArgs& a) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  return a.Cout % (64 * nt) == 0;
}
static size_t hx2_lds_bytes(const ConvArgs& a, int mode) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int ntaps = (mode == CONV_T2 || mode == CONV_S2) ? 4 : 9;
  const bool pn = hx2_pairn(a);
  size_t bytes = (size_t)((pn ? 1 : 2) * hx2_halo(a) + ntaps * 32 * nt * (pn ? 2 : 1)) * HRW + (256 + 128) * sizeof(float);
  if (a.gn_stats0) bytes += (size_t)(pn ? 1 : 2) * a.g.spt * (a.C0 + a.C1) * 2 * sizeof(float);  // scale/shift table
  return bytes;
}
bool conv_hx2_supported(const ConvArgs& a, int mode) {
  if (!a.wpkh || !a.hq || !a.range_flag) return false;
  if (a.res_mode == 2 && (!a.wskiph || !a.hq_skip)) return false;
  // 24-bit pixel indices and 32-bit element offsets inside the kernel (B = 8192 rows of 32x32x256 still fit)
  const size_t px_in = (size_t)a.B * a.Hin * a.Win, px_out = (size_t)a.B * a.g.HW * (mode == CONV_T2 ? 4 : 1);
  const size_t cmax = (size_t)(a.C0 > a.C1 ? a.C0 : a.C1) > (size_t)a.Cout ? (size_t)(a.C0 > a.C1 ? a.C0 : a.C1) : (size_t)a.Cout;
  if (px_in >= (1u << 24) || px_out >= (1u << 24) || (px_in > px_out ? px_in : px_out) * cmax >= (1ull << 32)) return false;
  if (mode == CONV_S2 && (a.Hin != 2 * a.g.H || a.Win != 2 * a.g.W || a.C1 != 0 || a.res_mode != 0)) return false;
  return hx2_halo(a) <= 448 && hx2_lds_bytes(a, mode) <= 160 * 1024;
}

bool conv_hx2_gn_supported(const ConvArgs& a, int mode) {
  const int cin = a.C0 + a.C1;
  if (!a.gn_stats0 || cin < 32 || cin % 8 != 0 || cin > 256) return false;  // <= 4 channels per lane in the prologue
  if (a.gn_nparts0 > 16 || a.gn_g.nparts > 16) return false;  // the prologue holds <= 16 partials per channel
  return conv_hx2_supported(a, mode);                         // (its LDS check includes the table)
}

int conv_hx2_init() {
  int rc = 0;
#define RAISEW(NTV, M, P) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2_kernel<NTV, M, P>), \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISEW(1, CONV_S1, false); RAISEW(1, CONV_UP2, false); RAISEW(1, CONV_T2, false);
  RAISEW(2, CONV_S1, false); RAISEW(2, CONV_UP2, false); RAISEW(2, CONV_T2, false);
  RAISEW(2, CONV_S1, true); RAISEW(2, CONV_UP2, true); RAISEW(2, CONV_T2, true);
  RAISEW(1, CONV_S2, false); RAISEW(2, CONV_S2, false); RAISEW(2, CONV_S2, true);
#undef RAISEW
  return rc;
}

void launch_conv_hx2(const ConvArgs& a_in, int mode, hipStream_t s) {
  ConvArgs a = a_in;
  a.halo_px = hx2_halo(a_in);
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  {
    const int tiles = geom_num_tiles(a.g, a.B);
    const bool pn = hx2_pairn(a);  // nt == 2 and Cout % 128 == 0
    dim3 grid(pn ? tiles : (tiles + 1) / 2, pn ? a.Cout / 128 : a.Cout / (32 * nt), mode == CONV_T2 ? 4 : 1);
    const size_t lds = hx2_lds_bytes(a, mode);
#define LAUNCHW(NTV, M, P) hipLaunchKernelGGL((conv_mfma_hx2_kernel<NTV, M, P>), grid, dim3(512), lds, s, a, tiles)
    if (pn) {
      if (mode == CONV_S1) LAUNCHW(2, CONV_S1, true);
      else if (mode == CONV_UP2) LAUNCHW(2, CONV_UP2, true);
      else if (mode == CONV_S2) LAUNCHW(2, CONV_S2, true);
      else LAUNCHW(2, CONV_T2, true);
    } else if (nt == 2) {
      if (mode == CONV_S1) LAUNCHW(2, CONV_S1, false);
      else if (mode == CONV_UP2) LAUNCHW(2, CONV_UP2, false);
      else if (mode == CONV_S2) LAUNCHW(2, CONV_S2, false);
      else LAUNCHW(2, CONV_T2, false);
    } else {
      if (mode == CONV_S1) LAUNCHW(1, CONV_S1, false);
      else if (mode == CONV_UP2) LAUNCHW(1, CONV_UP2, false);
      else if (mode == CONV_S2) LAUNCHW(1, CONV_S2, false);
      else LAUNCHW(1, CONV_T2, false);
    }
#undef LAUNCHW
  }
}

}  // namespace rgfm
