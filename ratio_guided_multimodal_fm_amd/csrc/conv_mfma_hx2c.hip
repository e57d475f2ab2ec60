// conv_mfma_hx2c.hip -- the stride-1 3x3 convs of the 8x8 level on the two-plane fp16 arithmetic, one K iteration and ONE
// barrier per 16-channel chunk.
//
// At 8x8 a 256-pixel tile is four whole samples and a launch has 128 tiles at B = 512: fewer workgroups than CUs, so a
// launch lasts as long as ONE workgroup's serial chain, whatever the batch (conv_mfma_hx2p_kernel: 128 -> 128 takes 41 us at
// 512 rows and 33 us at 32 rows -- 4 us per chunk: three units of barrier + weight wait + three taps each, where the
// chunk's matrix work is 1.45 us).  This kernel is cut for that chain: a workgroup is one tile x 64 channels (two per
// tile at Cout = 128: 256 workgroups), the halo and the weights of a WHOLE chunk are double-buffered in LDS (25 + 37 KB
// each; one workgroup per CU anyway), all nine taps run between two barriers, and while they run the next chunk is
// transformed + stored into the other halo buffer, its weights arrive by LDS-DMA and the chunk after that is fetched
// into registers.  Same arithmetic, LDS record layout, tap and product order as the other hx2 kernels (a fused 1x1 skip
// runs as one-tap chunks behind the main chunks, under its own weight scale, as there).
//
// Scope (conv_hx2c_supported): stride 1, 8x8 rasters (TileGeom::spt == 4), consumer-side GroupNorm on the input
// (ConvArgs::gn_stats0), Cout a multiple of 64, identity residual or fused 1x1 skip or none, optional time term; no
// producer-side finalize, no epilogue activation.  Weights: the packed fp16 images of conv_mfma_hx2p_kernel (a workgroup
// reads one 64-channel block or its half of a 128-channel block).
#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

template <bool SKIP>
__global__ __launch_bounds__(512, 1) void conv_mfma_hx2c_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int W = 8, SPT = 4, WR = W + 2, PREC = WR * WR, HALO = SPT * PREC;
  constexpr int ABYTES = (HALO + 1) * HRW;  // + a pad record (the store target of lanes past the halo)
  constexpr int NTHR = 512, NW = 8, CB = 64;
  constexpr int TAPB = CB * HRW, CHB = 9 * TAPB, PPT = TAPB / 1024;
  constexpr int NIT = (HALO * 4 + NTHR - 1) / NTHR;  // 4
  constexpr int MT_OFF = 4 * WR * HRW;               // pixel p + 32 of a sample: four halo rows down
  // The four 16-byte slots of a halo record are rotated with its halo column so that the 16 lanes of a ds_read_b128 phase
  // (the guide's lane groups) hit 64 distinct banks.  The rotation key (hx >> 2) of the 32-wide rasters leaves the
  // fragment reads of an 8-wide raster two-way conflicted (8 consecutive pixels, then a row stride of 10 records: only
  // two of the four slots in use per bank group); hx >> 1 is conflict-free here (tools/lds_swizzle.py).
  constexpr int SWZ = 1;
  extern __shared__ __attribute__((aligned(16))) char smc[];
  char* const sA = smc;
  char* const sB = smc + 2 * ABYTES;
  float* const sTab = reinterpret_cast<float*>(sB + 2 * CHB);  // [SPT + 1][cin][2]: S_A x (scale, shift) per sample, and a zero row
  const int cin = a.C0 + a.C1;
  char* const sDesc = reinterpret_cast<char*>(sTab) + (SPT + 1) * cin * 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31 = lane & 31, hp = lane >> 5;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int b0 = tile * SPT;

  // ---- consumer-side GroupNorm: scale/shift of the tile's four samples from the producers' partial statistics
  // (as conv_mfma_hx2q_kernel: rows side by side on the waves, two waves per row)
  {
    const int gn_cpg = cin >> 3;
    int gn_wsh = 1;  // log2 (waves per row): 8 waves / 4 rows
    {
      const int need = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : 2);
      gn_wsh = gn_wsh < need ? gn_wsh : need;
    }
    const int gn_row = wave >> gn_wsh, gn_part = wave & ((1 << gn_wsh) - 1);
    const int gn_b = b0 + gn_row;
    if (gn_row < SPT && gn_b < a.B) {
      const int gn_lpg = 8 << gn_wsh;
      const int gn_gl = lane >> (3 + gn_wsh), gn_sub = lane & (gn_lpg - 1);
      const int gn_gi = gn_part * (8 >> gn_wsh) + gn_gl;
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
          gn_v[p] = *reinterpret_cast<const float2*>(st + (((size_t)gn_b * npt + (p < npt ? p : 0)) * cs + cc) * 2);
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
          *reinterpret_cast<float2*>(sTab + ((size_t)gn_row * cin + gn_gi * gn_cpg + gn_sub + gn_lpg * k) * 2) = o;
        }
      }
    }
    for (int i = tid; i < 2 * cin; i += NTHR) sTab[SPT * cin * 2 + i] = 0.f;  // the all-zero row of the padding items
    // (rows of samples past the batch are never read: their items are invalid and take the zero row)
  }

  const int nmain = cin / KC;
  const int nskip = SKIP ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nmain + nskip;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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
    const u32x4 d = {(unsigned)pv, (unsigned)(pv >> 32), (unsigned)cs, 0u};
    *reinterpret_cast<u32x4*>(sDesc + tid * 16) = d;
  }

  // ---- per-item decode, once: source PIXEL index, LDS destination, table row (the sample inside the tile; SPT: invalid)
  const int q4 = tid & 3;
  unsigned ppix[NIT];
  int adst[NIT], trow[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int it = tid + NTHR * j;
    ppix[j] = 0u, trow[j] = SPT, adst[j] = HALO * HRW + (q4 >> 1) * 16 + (q4 & 1) * 8;
    if (it < HALO * 4) {
      const int rec = it >> 2;
      const int s = rec / PREC, rr = rec - s * PREC;
      const int hy = rr / WR, hx = rr - hy * WR;
      const int y = hy - 1, x = hx - 1;
      adst[j] = rec * HRW + ((((q4 >> 1) ^ (hx >> SWZ)) & 3) * 16) + (q4 & 1) * 8;  // plane l: ^ 32
      if (y >= 0 && y < W && x >= 0 && x < W && b0 + s < a.B) {
        trow[j] = s;
        ppix[j] = (unsigned)(((b0 + s) * W + y) * W + x);
      }
    }
  }

  // ---- fragment offsets: this lane's pixel l31 (+ 32: MT_OFF) of sample seg at tap column kx
  int aofs[3];
  {
    const int r = l31 >> 3, x = l31 & 7;
    const int arec = seg * PREC + r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aofs[kx] = (arec + kx) * HRW + ((hp ^ (((x + kx) >> SWZ) & 3)) & 3) * 16;
  }
  int bofs;
  {
    const int rec = grp * 32 + l31;
    bofs = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
  }

  // ---- weights: [channel block][chunk][tap] slabs; a workgroup of a 128-channel block takes its 64-channel half
  const bool nb128 = (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  const char* const wsk = reinterpret_cast<const char*>(SKIP ? a.wskiph : a.wpkh) + (size_t)wblk * nskip * TAPS + whalf;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned sB_lds = (unsigned)(size_t)sB;
  // position c (main chunk c, or skip chunk c - nmain: one tap) -> weight buffer c & 1.  Main: 36 one-KB pieces, five per
  // wave (the surplus ones repeat a piece); skip: four pieces, one per wave of the first four... every wave issues the
  // same count, so the hand-counted wait below holds for all of them.
  constexpr int NPW = 5;
  auto wdma = [&](int c) {
    const bool main = !SKIP || c < nmain;
    const char* src = wpk + (size_t)c * 9 * TAPS;
    if (SKIP && !main) src = wsk + (size_t)(c - nmain) * TAPS;
    const int npc = main ? 9 * PPT : PPT;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pq = wave_s + NW * j;
      const int pc = main ? (pq < npc ? pq : pq - npc) : (pq & (PPT - 1));
      const char* gsrc = src + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16;
      const unsigned dst = sB_lds + (unsigned)((c & 1) * CHB + pc * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };

  // raw values of the positions in flight: TWO sets -- a position is fetched two iterations before it is stored (one
  // iteration, 2.7 - 3.9 us, does not cover the fetches' round trip: with one set the loop ran at the memory latency,
  // whatever its barrier / unit structure)
  f32x4 ra0[NIT], ra1[NIT];
  float hmax = 0.f;
  auto issue = [&](f32x4 (&ra)[NIT], int c) {
    const u32x4 d = *reinterpret_cast<const u32x4*>(sDesc + c * 16);
    const float* src = reinterpret_cast<const float*>(((unsigned long long)d.y << 32) | (unsigned long long)d.x);
#pragma unroll
    for (int j = 0; j < NIT; ++j) ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24(ppix[j], d.z) + (unsigned)(q4 * 4)));
  };
  // GroupNorm + SiLU (main chunks) or the plain scale (skip chunks) + split + store of position c into halo buffer c & 1
  auto commit = [&](const f32x4 (&ra)[NIT], int c) {
    char* base = sA + (c & 1) * ABYTES;
    const bool xf = !SKIP || c < nmain;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const f32x4 v = ra[j];
      f32x4 o;
      if (xf) {
        const char* ep = reinterpret_cast<const char*>(sTab) + (trow[j] * cin + c * KC + 4 * q4) * 8;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 16);
        o.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
        o.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
        o.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
        o.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
      } else {
        const float sa = trow[j] < SPT ? HX_SA : 0.f;
        o.x = v.x * sa, o.y = v.y * sa, o.z = v.z * sa, o.w = v.w * sa;
      }
      unsigned h0, l0, h1, l1;
      hsplit2(o.x, o.y, h0, l0);
      hsplit2(o.z, o.w, h1, l1);
      hmax = hx_absmax3(o.x, o.y, hmax);
      hmax = hx_absmax3(o.z, o.w, hmax);
      const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
      *reinterpret_cast<hx_u32x2*>(base + adst[j]) = ph;
      *reinterpret_cast<hx_u32x2*>(base + (adst[j] ^ 32)) = pl;
    }
  };
#define HX2C_VM_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

  __syncthreads();  // the scale/shift table and the chunk descriptors are complete

  // ---- pipeline fill: chunk 0 in halo buffer 0 / weight buffer 0, chunk 1 in registers
  issue(ra0, 0);
  wdma(0);
  commit(ra0, 0);
  if (ntot > 1) issue(ra1, 1);
  if (ntot > 2) issue(ra0, 2);

  // ---- accumulators: bias (+ skip bias + time term), scaled by q (they hold q x the true sums); an identity residual
  // enters as fma(res, q, .)
  const float qmain = a.hq[0];
  const int sample = b0 + seg;  // this wave's sample
  const int ch0 = cb * CB + grp * 32 + l31;
  f32x16 acc[2];
  {
    float v = a.bias[ch0];
    if (SKIP) v += a.skip_bias[ch0];
    if (a.temb) v += a.temb[((size_t)(a.temb_per_row ? (sample < a.B ? sample : 0) : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + ch0];
    const float add0 = v * qmain;
    if (!SKIP && a.res_mode == 1) {
      const size_t pixr = (size_t)(sample < a.B ? sample : 0) * (W * W);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
          acc[mt][r] = a.res0[(pixr + p) * a.Cout + ch0];
        }
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
  }

  // one tap: 6 fragment reads, 6 MFMAs (a_l w_h, a_h w_l, a_h w_h per tile).  The reads of tap t + 1 are issued BEFORE the
  // MFMAs of tap t (two fragment sets, pinned with sched_barrier): left to itself hipcc emits read x 6, wait, MFMA x 6
  // per tap, and every tap of a wave then waits out an LDS round trip in front of its MFMAs (with two waves per SIMD
  // that was half of a chunk's time: 3.0 us per chunk against 1.45 us of matrix work)
  struct Frag {
    f16x8 a[2][2], b[2];
  };
  auto ldf = [&](Frag& f, const char* sArow, const char* sBt, int o0) {
    const int o1 = o0 ^ 32;
    f.a[0][0] = *reinterpret_cast<const f16x8*>(sArow + o0);
    f.a[0][1] = *reinterpret_cast<const f16x8*>(sArow + o1);
    f.a[1][0] = *reinterpret_cast<const f16x8*>(sArow + o0 + MT_OFF);
    f.a[1][1] = *reinterpret_cast<const f16x8*>(sArow + o1 + MT_OFF);
    f.b[0] = *reinterpret_cast<const f16x8*>(sBt + bofs);
    f.b[1] = *reinterpret_cast<const f16x8*>(sBt + (bofs ^ 32));
  };
  auto mma = [&](const Frag& f) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mt][PA[q]], f.b[PB[q]], acc[mt], 0, 0, 0);
  };

  // ---- K loop: one barrier per position.  Behind it the next position is transformed + stored (the other halo buffer),
  // its weights requested (the other weight buffer), the position after that fetched, and this position multiplied
  // one iteration; `ra` is the register set of position c + 1 (stored here; refilled with position c + 3)
  auto iteration = [&](f32x4 (&ra)[NIT], int c) {
    // this wave's weight pieces of position c have landed (the NIT fetches of position c + 2 issued behind them stay in
    // flight; those of position c + 1, older, are due now anyway)
    if (c + 2 < ntot) HX2C_VM_WAIT(4);
    else HX2C_VM_WAIT(0);
    static_assert(NIT == 4, "the hand-counted wait above");
    __syncthreads();
    // The two waves of a SIMD (a workgroup's waves w and w + 4: the two channel groups) take the two halves of the
    // iteration in OPPOSITE order -- group 0 stages the next position and then multiplies, group 1 multiplies and then
    // stages -- so that one wave's MFMAs run beside the other's transform + store.  In the same order both staged at once
    // and both multiplied at once: 3.0 us per chunk = 1.45 us of MFMAs + 1.0 us of staging + 0.5 us of DMA issue, summed.
    auto stage_next = [&]() {
      if (c + 1 < ntot) {
        commit(ra, c + 1);
        if (grp == 0) wdma(c + 1);
        if (c + 3 < ntot) issue(ra, c + 3);
      }
    };
    auto multiply = [&]() {
      const char* sAc = sA + (c & 1) * ABYTES;
      const char* sBc = sB + (c & 1) * CHB;
      if (!SKIP || c < nmain) {
        Frag f0, f1;
        // tap t = 3 ky + kx: halo row ky, column offset aofs[kx], weight slab t
#define HX2C_LDF(F, T) ldf(F, sAc + ((T) / 3) * WR * HRW, sBc + (T) * TAPB, aofs[(T) % 3])
#define HX2C_STEP(FN, FC, T)               \
  do {                                     \
      HX2C_LDF(FN, (T) + 1);                 \
      __builtin_amdgcn_sched_barrier(0);     \
      mma(FC);                               \
      __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
        HX2C_LDF(f0, 0);
        HX2C_STEP(f1, f0, 0);
        HX2C_STEP(f0, f1, 1);
        HX2C_STEP(f1, f0, 2);
        HX2C_STEP(f0, f1, 3);
        HX2C_STEP(f1, f0, 4);
        HX2C_STEP(f0, f1, 5);
        HX2C_STEP(f1, f0, 6);
        HX2C_STEP(f0, f1, 7);
        mma(f0);
#undef HX2C_STEP
#undef HX2C_LDF
        if (SKIP && c == nmain - 1) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
          const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt] = acc[mt] * rs;
        }
      } else {
        Frag f;
        ldf(f, sAc + WR * HRW, sBc, aofs[1]);  // (the centre tap)
        mma(f);
      }
    };
    if (grp == 0) {
      stage_next();
      multiply();
    } else {
      if (c + 1 < ntot) wdma(c + 1);  // (requested first: a chunk of MFMAs for it to arrive)
      multiply();
      stage_next();
    }
  };
#pragma unroll 1
  for (int c = 0; c < ntot; c += 2) {
    iteration(ra1, c);
    if (c + 1 < ntot) iteration(ra0, c + 1);
  }
#undef HX2C_VM_WAIT
  if (!(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);  // (rare) plane h would be >= 32768 (or inf)

  // ---- epilogue: a wave's 64 pixels are one whole sample
  {
    const float qinv = SKIP ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = acc[mt] * qinv;
  }
  if (sample >= a.B) return;  // (wave-uniform; no barrier follows)
  if (a.small_check && a.range_flag) {  // (ConvArgs::small_check: the output's low range)
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][r], acc[mt][r + 1], m);
    hx_small_flag(a.range_flag, m);
  }
  const size_t pix0 = (size_t)sample * (W * W);
  if (a.out) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
        a.out[(pix0 + p) * a.Cout + ch0] = acc[mt][r];
      }
  }
  if (a.stats_out || a.pout) {
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
    if (a.stats_out && hp == 0) store_stats(a, a.stats_out + ((size_t)sample * a.g.nparts * a.Cout + ch0) * 2, mean, m2);
    if (a.pout) {
      // P format for the norm that consumes this output (ConvArgs::pout): a wave holds a whole sample x 32 channels, i.e.
      // whole GroupNorm groups (Cout / 8 = 8, 16 or 32 channels), so the norm's statistics are finished right here
      float sc, sh;
      hx_group_affine(mean, m2, 64.f, a.Cout >> 3, a.pn_gamma[ch0], a.pn_beta[ch0], sc, sh);
      const unsigned pstride = (unsigned)a.Cout * 4u;
      char* const rec0 = reinterpret_cast<char*>(a.pout) + pix0 * pstride + (size_t)(ch0 >> 4) * 64;
      const float pm = hx_p_emit(acc[0], acc[1], sc, sh, rec0, pstride, l31, hp);
      if (!(pm < HX_BIG)) atomicOr(a.range_flag, 1u);
    }
  }
}

// ---------------------------------------------------------------- host side
static int g_hx2c_on = 1;
void conv_hx2c_set(int v) { g_hx2c_on = v; }
static int g_hx2c_all = 0;  // tools/kbench: every supported shape
void conv_hx2c_set_all(int v) { g_hx2c_all = v; }

static size_t hx2c_lds_bytes(const ConvArgs& a) {
  const int cin = a.C0 + a.C1, nch = cin / KC + (a.res_mode == 2 ? (a.R0 + a.R1) / KC : 0);
  return (size_t)2 * (4 * 100 + 1) * HRW + (size_t)2 * 9 * 64 * HRW + (size_t)5 * cin * 8 + (size_t)nch * 16;
}

bool conv_hx2c_supported(const ConvArgs& a, int mode) {
  if (!g_hx2c_on || mode != CONV_S1) return false;
  if (!a.gn_stats0 || a.ab) return false;  // convs of raw inputs stay on conv_mfma_hx2p_kernel
  if (!conv_hx2_supported(a, mode) || !conv_hx2_gn_supported(a, mode)) return false;
  const TileGeom& g = a.g;
  if (g.W != 8 || g.H != 8 || g.spt != 4 || a.Hin != 8 || a.Win != 8) return false;
  if (a.Cout % 64 != 0 || (a.C0 + a.C1) % KC != 0) return false;
  if (a.res_mode == 2 && (a.R0 + a.R1) % KC != 0) return false;
  if (a.res_mode == 1 && a.R0 != a.Cout) return false;
  if (a.ep_scale || a.fin_ab) return false;
  if (a.pout && (a.Cout != 64 && a.Cout != 128 && a.Cout != 256)) return false;  // (whole power-of-two groups per wave)
  if (hx2c_lds_bytes(a) > 160 * 1024) return false;
  // Where it pays (tools/kbench/scripts/q23.sh, q24.sh): 128 -> 128 and 256 -> 128 without a fused skip -3 ... -8 % against
  // conv_mfma_hx2p_kernel at 32 ... 512 rows; with the 1x1 skip (16 one-tap chunks, each behind a whole chunk of staging)
  // +7 %: those stay on hx2p.  g_hx2c_all (kbench) lifts the restriction.
  return g_hx2c_all || a.res_mode != 2;
}

int conv_hx2c_init() {
  int rc = 0;
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2c_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2c_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

void launch_conv_hx2c(const ConvArgs& a, hipStream_t s) {
  const int tiles = geom_num_tiles(a.g, a.B);
  const dim3 grid(tiles, a.Cout / 64);
  const size_t lds = hx2c_lds_bytes(a);
  if (a.res_mode == 2) hipLaunchKernelGGL(conv_mfma_hx2c_kernel<true>, grid, dim3(512), lds, s, a, tiles);
  else hipLaunchKernelGGL(conv_mfma_hx2c_kernel<false>, grid, dim3(512), lds, s, a, tiles);
}

}  // namespace rgfm
