// conv_mfma_hx2d.hip -- stride-1 3x3 convs of the 16x16 and 8x8 levels whose input arrives PRE-NORMALISED and PRE-SPLIT
// ("P format", written by the producing conv's epilogue: conv_hx2_common.h, hx_p_*), so that the K loop stages by
// LDS-DMA only: no GroupNorm table, no fetch -> transform -> store chain through registers, no vector-ALU work between
// the MFMAs but the fragment reads.
//
// Why: in a ResBlock conv1's output h is consumed by exactly one reader, conv2,
// through GroupNorm + SiLU (reference unet_flexible.py:79-81).  At the 16x16 and 8x8 levels a workgroup of the producing
// conv owns whole (sample, group) sets, so it can finish norm2's statistics itself and write S_A silu(norm2(h)) as the two
// fp16 planes the consumer's MFMAs read -- once per element instead of once per staged halo element per consuming
// workgroup, and outside anybody's K loop.  The consumer's staging then is global_load_lds_dwordx4 with per-lane source
// addresses (the swizzle of the LDS record is a permutation of the SOURCE slots; padding records read a zero page).
//
// Tiling, arithmetic, LDS record layout, tap and product order: conv_mfma_hx2c_kernel's (one tile x 64 channels per
// workgroup, whole-chunk double buffers, one barrier per chunk, fragment reads a tap ahead), for W = 8 (a tile = four
// samples) and W = 16 (a tile = one sample, a wave segment = four image rows).  A fused 1x1 skip (raw sources) runs as
// one-tap chunks behind the main chunks, staged through registers as in conv_mfma_hx2c_kernel.
#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

template <int W, bool SKIP>
__global__ __launch_bounds__(512, 1) void conv_mfma_hx2d_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int SPT = W == 8 ? 4 : 1, H = W, WR = W + 2, HR = H + 2;
  constexpr int PREC = WR * HR, HALO = SPT * PREC;  // 400 (W = 8) / 324 (W = 16) records of 64 B
  constexpr int NPC = (HALO + 15) / 16;             // 1-KB DMA pieces per chunk: 25 / 21 (the last one of W = 16 is partial)
  constexpr int ABYTES = NPC * 1024;
  constexpr int NTHR = 512, NW = 8, CB = 64;
  constexpr int TAPB = CB * HRW, CHB = 9 * TAPB, PPT = TAPB / 1024;
  constexpr int NPH = (NPC + NW - 1) / NW;          // halo pieces per wave and chunk: 4 / 3
  constexpr int RPS = 64 / W;                       // image rows per 64-pixel wave segment: 8 / 4
  constexpr int MT_OFF = (RPS / 2) * WR * HRW;      // pixel p + 32 of a segment: half a segment's rows down
  constexpr int SWZ = 1;                            // slot rotation key hx >> 1 (tools/lds_swizzle.py: conflict-free for 8- / 16-wide rasters)
  extern __shared__ __attribute__((aligned(16))) char smd[];
  char* const sA = smd;
  char* const sB = smd + 2 * ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31 = lane & 31, hp = lane >> 5;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int b0 = tile * SPT;
  const int cin = a.C0;
  const int nmain = cin / KC;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  // ---- halo DMA: this lane's source of each of its pieces, once.  Lane L of piece pc fills LDS bytes [16 L, 16 L + 16) of
  // the piece = physical slot L & 3 of record 16 pc + L / 4; the record's logical slot there is (L & 3) ^ key(hx).
  // P format (conv_hx2_common.h): [pixel][chunk][h ch 0-7 | h ch 8-15 | l ch 0-7 | l ch 8-15], 64 B per (pixel, chunk).
  unsigned soff[NPH];
  bool sval[NPH];
#pragma unroll
  for (int j = 0; j < NPH; ++j) {
    const int pq = wave + NW * j;
    const int pc = pq < NPC ? pq : pq - NPC;  // (surplus slots repeat an earlier piece: the same bytes to the same place)
    const int rec = pc * 16 + (lane >> 2);
    const int s = rec / PREC, rr = rec - s * PREC;
    const int hy = rr / WR, hx = rr - hy * WR;
    const int y = hy - 1, x = hx - 1;
    const int logical = ((lane & 3) ^ (hx >> SWZ)) & 3;
    sval[j] = rec < HALO && y >= 0 && y < H && x >= 0 && x < W && b0 + s < a.B;
    soff[j] = sval[j] ? (unsigned)((((b0 + s) * H + y) * W + x) * nmain) * 64u + (unsigned)logical * 16u : 0u;
  }
  const char* const pin = reinterpret_cast<const char*>(a.pin0);
  const char* const zeros = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const unsigned sA_lds = (unsigned)(size_t)sA, sB_lds = (unsigned)(size_t)sB;
  auto hdma1 = [&](int c, int j) {  // halo piece j (0 .. NPH - 1) of this wave, chunk c
    const int pq = wave_s + NW * j;
    const int pc = pq < NPC ? pq : pq - NPC;
    const char* gsrc = sval[j] ? pin + (size_t)soff[j] + (size_t)c * 64 : zeros;
    const unsigned dst = sA_lds + (unsigned)((c & 1) * ABYTES + pc * 1024);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  auto hdma = [&](int c) {
#pragma unroll
    for (int j = 0; j < NPH; ++j) hdma1(c, j);
  };

  // ---- weights: as conv_mfma_hx2c_kernel ([channel block][chunk][tap] slabs; a workgroup of a 128-channel block takes its half)
  const int nskip = SKIP ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nmain + nskip;
  const bool nb128 = (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  const char* const wsk = reinterpret_cast<const char*>(SKIP ? a.wskiph : a.wpkh) + (size_t)wblk * nskip * TAPS + whalf;
  constexpr int NPW = 5;  // 36 one-KB pieces of a main chunk over 8 waves (the surplus ones repeat a piece)
  auto wdma1 = [&](int c, int j) {  // weight piece j (0 .. NPW - 1) of this wave, position c
    const bool main = !SKIP || c < nmain;
    const char* src = wpk + (size_t)c * 9 * TAPS;
    if (SKIP && !main) src = wsk + (size_t)(c - nmain) * TAPS;
    const int npc = main ? 9 * PPT : PPT;
    const int pq = wave_s + NW * j;
    const int pc = main ? (pq < npc ? pq : pq - npc) : (pq & (PPT - 1));
    const char* gsrc = src + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16;
    const unsigned dst = sB_lds + (unsigned)((c & 1) * CHB + pc * 1024);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  auto wdma = [&](int c) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) wdma1(c, j);
  };

  // ---- the 1x1 skip's raw sources (SKIP): centre records only, through registers (scale, split, store) as the raw chunks of
  // conv_mfma_hx2c_kernel; a skip chunk overwrites the interior records of a halo buffer, its border records are not read
  constexpr int NIS = SKIP ? (SPT * H * W * 4) / NTHR : 1;  // 16-byte items of the 256 centre pixels per thread: 2
  unsigned spix[NIS];
  int sdst[NIS];
  bool sok[NIS];
  const int q4 = tid & 3;
  if (SKIP) {
#pragma unroll
    for (int j = 0; j < NIS; ++j) {
      const int it = tid + NTHR * j, px = it >> 2;
      const int s = px / (H * W), pr = px - s * (H * W);
      const int y = pr / W, x = pr - y * W;
      const int rec = s * PREC + (y + 1) * WR + (x + 1);
      sdst[j] = rec * HRW + ((((q4 >> 1) ^ ((x + 1) >> SWZ)) & 3) * 16) + (q4 & 1) * 8;
      sok[j] = b0 + s < a.B;
      spix[j] = sok[j] ? (unsigned)((b0 + s) * (H * W) + pr) : 0u;
    }
  }
  f32x4 rs0[NIS], rs1[NIS];
  float hmax = 0.f;
  auto sissue = [&](f32x4 (&ra)[NIS], int k) {  // skip chunk k (16 channels of cat(res0, res1))
    const int c = k * KC;
    const bool first = c < a.R0;
    const float* src = first ? a.res0 + c : a.res1 + (c - a.R0);
    const unsigned cs = (unsigned)(first ? a.R0 : a.R1);
#pragma unroll
    for (int j = 0; j < NIS; ++j) ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24(spix[j], cs) + (unsigned)(q4 * 4)));
  };
  auto scommit = [&](const f32x4 (&ra)[NIS], int pos) {
    char* base = sA + (pos & 1) * ABYTES;
#pragma unroll
    for (int j = 0; j < NIS; ++j) {
      const float sa = sok[j] ? HX_SA : 0.f;
      const f32x4 v = ra[j];
      f32x4 o;
      o.x = v.x * sa, o.y = v.y * sa, o.z = v.z * sa, o.w = v.w * sa;
      unsigned h0, l0, h1, l1;
      hsplit2(o.x, o.y, h0, l0);
      hsplit2(o.z, o.w, h1, l1);
      hmax = hx_absmax3(o.x, o.y, hmax);
      hmax = hx_absmax3(o.z, o.w, hmax);
      const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
      *reinterpret_cast<hx_u32x2*>(base + sdst[j]) = ph;
      *reinterpret_cast<hx_u32x2*>(base + (sdst[j] ^ 32)) = pl;
    }
  };

  // ---- fragment offsets: this lane's pixel l31 (+ 32: MT_OFF) of segment seg at tap column kx
  int aofs[3];
  {
    const int r = l31 / W, x = l31 % W;
    const int arec = (SPT == 4 ? seg * PREC : seg * RPS * WR) + r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aofs[kx] = (arec + kx) * HRW + ((hp ^ (((x + kx) >> SWZ) & 3)) & 3) * 16;
  }
  int bofs;
  {
    const int rec = grp * 32 + l31;
    bofs = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
  }

  // ---- pipeline fill: chunk 0 on its way into buffer 0
  hdma(0);
  wdma(0);
  if (SKIP) {
    sissue(rs0, 0);
    if (nskip > 1) sissue(rs1, 1);
  }

  // ---- accumulators: bias (+ skip bias + time term), scaled by q; an identity residual enters as fma(res, q, .)
  const float qmain = a.hq[0];
  const int sample = SPT == 4 ? b0 + seg : b0;     // this wave's sample
  const int part = SPT == 4 ? 0 : seg;             // ... and its statistics part / 64-pixel segment of the sample
  const int ch0 = cb * CB + grp * 32 + l31;
  const size_t pix0 = (size_t)sample * (H * W) + (SPT == 4 ? 0 : seg * 64);
  f32x16 acc[2];
  {
    float v = a.bias[ch0];
    if (SKIP) v += a.skip_bias[ch0];
    if (a.temb) v += a.temb[((size_t)(a.temb_per_row ? (sample < a.B ? sample : 0) : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + ch0];
    const float add0 = v * qmain;
    if (!SKIP && a.res_mode == 1) {
      const size_t pixr = sample < a.B ? pix0 : 0;
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

  // ---- K loop: one barrier per position.  Every DMA this wave has in flight at the top of iteration c belongs to
  // position c (issued one iteration earlier), so the wait is vmcnt(0) (SKIP: the register fetches of later skip chunks,
  // issued behind them, are waited for too -- they are two iterations old by then).
  auto iteration = [&](f32x4 (&ra)[NIS], int c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* sAc = sA + (c & 1) * ABYTES;
    const char* sBc = sB + (c & 1) * CHB;
    // The next position's NPW + NPH one-KB pieces are requested BETWEEN this position's taps, two behind each of the first
    // taps: an LDS-DMA costs its wave 60 - 180 issue cycles (MI355X_MICROARCH.md), 9 of them in a row in front of the MFMAs
    // were a quarter of a chunk's time (2.0 us per chunk where the matrix work is 1.45); between the taps they issue in
    // the MFMAs' shadow, and the last four taps cover the landing of the last pieces.
    const bool nxt = c + 1 < ntot;
    const bool nxt_main = !SKIP || c + 1 < nmain;
    auto dma_slot = [&](int t) {  // pieces 2 t and 2 t + 1 of the next position: weights first, then the halo
#pragma unroll
      for (int k = 2 * t; k < 2 * t + 2; ++k) {
        if (k < NPW) {
          if (nxt) wdma1(c + 1, k);
        } else if (k < NPW + NPH) {
          if (nxt && nxt_main) hdma1(c + 1, k - NPW);
        }
      }
    };
    if (!SKIP || c < nmain) {
      Frag f0, f1;
#define HX2D_LDF(F, T) ldf(F, sAc + ((T) / 3) * WR * HRW, sBc + (T) * TAPB, aofs[(T) % 3])
#define HX2D_STEP(FN, FC, T)               \
  do {                                     \
      HX2D_LDF(FN, (T) + 1);                 \
      __builtin_amdgcn_sched_barrier(0);     \
      dma_slot(T);                           \
      mma(FC);                               \
      __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
      HX2D_LDF(f0, 0);
      HX2D_STEP(f1, f0, 0);
      HX2D_STEP(f0, f1, 1);
      HX2D_STEP(f1, f0, 2);
      HX2D_STEP(f0, f1, 3);
      HX2D_STEP(f1, f0, 4);
      HX2D_STEP(f0, f1, 5);
      HX2D_STEP(f1, f0, 6);
      HX2D_STEP(f0, f1, 7);
      mma(f0);
#undef HX2D_STEP
#undef HX2D_LDF
      if (SKIP && c == nmain - 1) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
        const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt] = acc[mt] * rs;
      }
    } else {
      if (nxt) wdma(c + 1);
      Frag f;
      ldf(f, sAc + WR * HRW, sBc, aofs[1]);  // (the centre tap)
      mma(f);
    }
    // SKIP: the raw chunk of position c + 1 goes into the other halo buffer behind this position's MFMAs (every wave is
    // past its reads of that buffer: they ended before this iteration's barrier), the one after next is fetched
    if (SKIP && c + 1 >= nmain && c + 1 < ntot) {
      scommit(ra, c + 1);
      if (c + 3 < ntot) sissue(ra, c + 3 - nmain);
    }
  };
  if (SKIP) {
    // register sets alternate with the position's parity from the first skip chunk on: set (k & 1) holds skip chunk k
#pragma unroll 1
    for (int c = 0; c < ntot; ++c) {
      const int k1 = c + 1 - nmain;  // the skip chunk committed in this iteration
      if (k1 >= 0 && (k1 & 1)) iteration(rs1, c);
      else iteration(rs0, c);
    }
  } else {
#pragma unroll 1
    for (int c = 0; c < ntot; ++c) iteration(rs0, c);
  }
  if (SKIP && !(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);  // (rare) plane h of a raw skip source would be >= 32768 (or inf)

  // ---- epilogue (conv_mfma_hx2c_kernel's): a wave's 64 pixels are one whole sample (W = 8) or four rows of one (W = 16)
  {
    const float qinv = SKIP ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = acc[mt] * qinv;
  }
  if (sample >= a.B) return;  // (wave-uniform; no barrier follows)
  if (a.small_check && a.range_flag) {
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][r], acc[mt][r + 1], m);
    hx_small_flag(a.range_flag, m);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
      a.out[(pix0 + p) * a.Cout + ch0] = acc[mt][r];
    }
  if (a.stats_out) {
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

// (Measured and rejected, round 4 -- tools/experiments/conv_mfma_hx2d16.hip, profiles/r04_kbench/hx2d_mfma_16x16x32.txt: the
// eight-wave kernel re-cut for v_mfma_f32_16x16x32_f16.  A register-only loop of that shape sustains 1.17 x the FLOP/s of the
// 32x32x16 loop on this part (mfma_shapes.txt: 1982 vs 1694 TFLOP/s, a higher clock at equal cycles), but the conv is not a
// bare MFMA loop: with K = 32 = the two planes of a chunk as one operand and a_h w_h of two taps paired by
// v_permlane32_swap (no extra LDS reads, same results to rounding) the chunk needs 112 instructions + 24 swaps instead of
// 54, and the launch takes 31.9 vs 28.9 us at 512 rows, 25.8 vs 19.5 at 32.)

// ---------------------------------------------------------------- four waves, two workgroups per CU
// conv_mfma_hx2d4_kernel: the same conv cut for the LDS.  The eight-wave kernel above reads 1 KB of fragments per MFMA (a
// wave tile of 64 pixels x 32 channels: four A and two B reads per six MFMAs) -- with the DMA writes the LDS is busy ~70 % of a
// chunk's MFMA time, and its chunk takes 1.9 - 2.9 us where the matrix work is 1.45 (tools/kbench, round 4).  Here a wave
// owns 64 pixels x 64 channels (NT = 2: eight reads per twelve MFMAs, 0.67 KB per MFMA), a workgroup is FOUR waves = one
// tile x 64 channels, and the weights are double-buffered per UNIT of three taps (12 KB) instead of per chunk, so that a
// workgroup needs <= 80 KB and TWO share a CU: the two waves of a SIMD belong to different workgroups with their own
// barriers, and one's barrier / DMA-issue gaps fall under the other's MFMAs.
// (Tried on this cut and dropped, profiles/r04_kbench/hx2d4_triple_weight_buffers.txt: THREE weight buffers at 16x16 with the
// weights requested two units ahead and a counted vmcnt that leaves the previous unit's requests in flight -- 105.7 vs
// 103.6 us at 128 channels, 33.1 vs 32.3 at 64, slower at 32 rows: the wait for the DMA is not what the unit's time is made of.
// Nor is the LDS round trip behind the barrier: a unit's LAST tap multiplied behind the NEXT unit's barrier, from fragments
// kept in registers (same summation order), made it 110.6 vs 103.6 / 34.7 vs 32.3 us -- hx2d4_deferred_last_tap.txt.)
template <int W, bool SKIP>
__global__ __launch_bounds__(256, 2) void conv_mfma_hx2d4_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int SPT = W == 8 ? 4 : 1, H = W, WR = W + 2, HR = H + 2;
  constexpr int PREC = WR * HR, HALO = SPT * PREC;
  constexpr int NPC = (HALO + 15) / 16;             // 25 / 21 one-KB pieces per chunk
  constexpr int ABYTES = NPC * 1024;
  constexpr int NTHR = 256, NW = 4, CB = 64;
  constexpr int TAPB = CB * HRW, UB = 3 * TAPB, PPT = TAPB / 1024;  // a tap's slab 4 KB, a unit 12 KB
  constexpr int NPH = (NPC + NW - 1) / NW;          // halo pieces per wave and chunk: 7 / 6
  constexpr int NPU = 3 * PPT / NW;                 // weight pieces per wave and unit: 3
  constexpr int RPS = 64 / W;
  constexpr int MT_OFF = (RPS / 2) * WR * HRW;
  constexpr int SWZ = 1;
  extern __shared__ __attribute__((aligned(16))) char smd4[];
  char* const sA = smd4;
  char* const sB = smd4 + 2 * ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int l31 = lane & 31, hp = lane >> 5;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int b0 = tile * SPT;
  const int nmain = a.C0 / KC;
  const int wave_s = __builtin_amdgcn_readfirstlane(seg);

  unsigned soff[NPH];
  bool sval[NPH];
#pragma unroll
  for (int j = 0; j < NPH; ++j) {
    const int pq = seg + NW * j;
    const int pc = pq < NPC ? pq : pq - NPC;
    const int rec = pc * 16 + (lane >> 2);
    const int s = rec / PREC, rr = rec - s * PREC;
    const int hy = rr / WR, hx = rr - hy * WR;
    const int y = hy - 1, x = hx - 1;
    const int logical = ((lane & 3) ^ (hx >> SWZ)) & 3;
    sval[j] = rec < HALO && y >= 0 && y < H && x >= 0 && x < W && b0 + s < a.B;
    soff[j] = sval[j] ? (unsigned)((((b0 + s) * H + y) * W + x) * nmain) * 64u + (unsigned)logical * 16u : 0u;
  }
  const char* const pin = reinterpret_cast<const char*>(a.pin0);
  const char* const zeros = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const unsigned sA_lds = (unsigned)(size_t)sA, sB_lds = (unsigned)(size_t)sB;
  auto dma = [&](const char* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  auto hdma1 = [&](int c, int j) {  // halo piece j (0 .. NPH - 1) of this wave, chunk c
    const int pq = wave_s + NW * j;
    const int pc = pq < NPC ? pq : pq - NPC;
    dma(sval[j] ? pin + (size_t)soff[j] + (size_t)c * 64 : zeros, sA_lds + (unsigned)((c & 1) * ABYTES + pc * 1024));
  };

  // ---- weights.  Unit g < 3 nmain: kernel row g % 3 of main chunk g / 3 (three consecutive taps of the packed image);
  // g >= 3 nmain: the one tap of skip chunk g - 3 nmain.  Unit g lives in weight buffer g & 1.
  const int nskip = SKIP ? (a.R0 + a.R1) / KC : 0;
  const int G = 3 * nmain + nskip;
  const bool nb128 = (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  const char* const wsk = reinterpret_cast<const char*>(SKIP ? a.wskiph : a.wpkh) + (size_t)wblk * nskip * TAPS + whalf;
  auto wdma1 = [&](int g, int j) {  // weight piece j (0 .. NPU - 1) of this wave, unit g
    const bool main = !SKIP || g < 3 * nmain;
    const int pq = wave_s + NW * j;                  // 0 .. 11: tap pq / 4, KB pq % 4 of the unit
    const int pc = main ? pq : (pq & (PPT - 1));     // (a skip unit is one tap: every wave repeats its KB)
    const char* src = main ? wpk + (size_t)g * 3 * TAPS : wsk + (size_t)(g - 3 * nmain) * TAPS;
    dma(src + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16, sB_lds + (unsigned)((g & 1) * UB + pc * 1024));
  };

  // ---- the 1x1 skip's raw sources (SKIP): centre records through registers, as in the eight-wave kernel
  constexpr int NIS = SKIP ? (SPT * H * W * 4) / NTHR : 1;  // 4
  unsigned spix[NIS];
  int sdst[NIS];
  bool sok[NIS];
  const int q4 = tid & 3;
  if (SKIP) {
#pragma unroll
    for (int j = 0; j < NIS; ++j) {
      const int it = tid + NTHR * j, px = it >> 2;
      const int s = px / (H * W), pr = px - s * (H * W);
      const int y = pr / W, x = pr - y * W;
      const int rec = s * PREC + (y + 1) * WR + (x + 1);
      sdst[j] = rec * HRW + ((((q4 >> 1) ^ ((x + 1) >> SWZ)) & 3) * 16) + (q4 & 1) * 8;
      sok[j] = b0 + s < a.B;
      spix[j] = sok[j] ? (unsigned)((b0 + s) * (H * W) + pr) : 0u;
    }
  }
  f32x4 rs0[NIS], rs1[NIS];
  float hmax = 0.f;
  auto sissue = [&](f32x4 (&ra)[NIS], int k) {
    const int c = k * KC;
    const bool first = c < a.R0;
    const float* src = first ? a.res0 + c : a.res1 + (c - a.R0);
    const unsigned cs = (unsigned)(first ? a.R0 : a.R1);
#pragma unroll
    for (int j = 0; j < NIS; ++j) ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24(spix[j], cs) + (unsigned)(q4 * 4)));
  };
  auto scommit = [&](const f32x4 (&ra)[NIS], int pos) {  // into halo buffer pos & 1
    char* base = sA + (pos & 1) * ABYTES;
#pragma unroll
    for (int j = 0; j < NIS; ++j) {
      const float sa = sok[j] ? HX_SA : 0.f;
      const f32x4 v = ra[j];
      f32x4 o;
      o.x = v.x * sa, o.y = v.y * sa, o.z = v.z * sa, o.w = v.w * sa;
      unsigned h0, l0, h1, l1;
      hsplit2(o.x, o.y, h0, l0);
      hsplit2(o.z, o.w, h1, l1);
      hmax = hx_absmax3(o.x, o.y, hmax);
      hmax = hx_absmax3(o.z, o.w, hmax);
      const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
      *reinterpret_cast<hx_u32x2*>(base + sdst[j]) = ph;
      *reinterpret_cast<hx_u32x2*>(base + (sdst[j] ^ 32)) = pl;
    }
  };

  // ---- fragment offsets
  int aofs[3];
  {
    const int r = l31 / W, x = l31 % W;
    const int arec = (SPT == 4 ? seg * PREC : seg * RPS * WR) + r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aofs[kx] = (arec + kx) * HRW + ((hp ^ (((x + kx) >> SWZ) & 3)) & 3) * 16;
  }
  int bofs[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int rec = nt * 32 + l31;
    bofs[nt] = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
  }

  // ---- pipeline fill: chunk 0's halo and unit 0's weights on their way
#pragma unroll
  for (int j = 0; j < NPH; ++j) hdma1(0, j);
#pragma unroll
  for (int j = 0; j < NPU; ++j) wdma1(0, j);
  if (SKIP) {
    sissue(rs0, 0);
    if (nskip > 1) sissue(rs1, 1);
  }

  // ---- accumulators
  const float qmain = a.hq[0];
  const int sample = SPT == 4 ? b0 + seg : b0;
  const int part = SPT == 4 ? 0 : seg;
  const int ch0 = cb * CB + l31;  // (+ 32 nt)
  const size_t pix0 = (size_t)sample * (H * W) + (SPT == 4 ? 0 : seg * 64);
  f32x16 acc[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int ch = ch0 + 32 * nt;
    float v = a.bias[ch];
    if (SKIP) v += a.skip_bias[ch];
    if (a.temb) v += a.temb[((size_t)(a.temb_per_row ? (sample < a.B ? sample : 0) : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + ch];
    const float add0 = v * qmain;
    if (!SKIP && a.res_mode == 1) {
      const size_t pixr = sample < a.B ? pix0 : 0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
          acc[mt][nt][r] = a.res0[(pixr + p) * a.Cout + ch];
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(acc[mt][nt][r], qmain, add0);
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = add0;
    }
  }

  struct Frag {
    f16x8 a[2][2], b[2][2];
  };
  auto ldf = [&](Frag& f, const char* sArow, const char* sBt, int o0) {
    const int o1 = o0 ^ 32;
    f.a[0][0] = *reinterpret_cast<const f16x8*>(sArow + o0);
    f.a[0][1] = *reinterpret_cast<const f16x8*>(sArow + o1);
    f.a[1][0] = *reinterpret_cast<const f16x8*>(sArow + o0 + MT_OFF);
    f.a[1][1] = *reinterpret_cast<const f16x8*>(sArow + o1 + MT_OFF);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f.b[nt][0] = *reinterpret_cast<const f16x8*>(sBt + bofs[nt]);
      f.b[nt][1] = *reinterpret_cast<const f16x8*>(sBt + (bofs[nt] ^ 32));
    }
  };
  auto mma = [&](const Frag& f) {  // (a_l w_h, a_h w_l, a_h w_h per accumulator: the other kernels' product order)
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[mt][PA[q]], f.b[nt][PB[q]], acc[mt][nt], 0, 0, 0);
  };

  // ---- K loop: one barrier per unit.  At the top of unit g every DMA this wave has in flight was issued during unit
  // g - 1 (unit g's weights; a share of the next chunk's halo), so the wait is vmcnt(0).  Behind the barrier: the weights
  // of unit g + 1 into the other weight buffer (last read in unit g - 1), this unit's share of the NEXT chunk's halo into
  // the other halo buffer (last read in the previous chunk), between the taps' MFMAs.
  constexpr int HS0 = (NPH + 2) / 3, HS1 = (NPH + 1) / 3;  // halo pieces issued in a chunk's units 0 / 1 / 2: 3 2 2 or 2 2 2
  auto unit_main = [&](int c, auto ky_tag) {
    constexpr int KY = decltype(ky_tag)::value;
    const int g = 3 * c + KY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* sArow = sA + (c & 1) * ABYTES + KY * WR * HRW;
    const char* sBu = sB + (g & 1) * UB;
    const bool nxt = g + 1 < G;
    const bool halo_nxt = c + 1 < nmain;
    constexpr int J0 = KY == 0 ? 0 : (KY == 1 ? HS0 : HS0 + HS1);
    constexpr int J1 = KY == 0 ? HS0 : (KY == 1 ? HS0 + HS1 : NPH);
    Frag f0, f1;
    ldf(f0, sArow, sBu, aofs[0]);
    ldf(f1, sArow, sBu + TAPB, aofs[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (nxt) wdma1(g + 1, 0);
    if (halo_nxt && J0 < J1) hdma1(c + 1, J0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    ldf(f0, sArow, sBu + 2 * TAPB, aofs[2]);
    __builtin_amdgcn_sched_barrier(0);
    if (nxt) wdma1(g + 1, 1);
    if (halo_nxt && J0 + 1 < J1) hdma1(c + 1, J0 + 1);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    if (nxt) wdma1(g + 1, 2);
    if (halo_nxt && J0 + 2 < J1) hdma1(c + 1, J0 + 2);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int c = 0; c < nmain; ++c) {
    unit_main(c, K0{});
    unit_main(c, K1{});
    unit_main(c, K2{});
    // SKIP: the first raw chunk goes into the other halo buffer behind the last main chunk's MFMAs (nobody reads that
    // buffer any more: the last main chunk's predecessor is done)
    if (SKIP && c == nmain - 1) {
      const float rs = a.hq_skip[0] * a.hq[1];  // the 1x1 skip weights carry their own scale: q_main -> q_skip
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
      scommit(rs0, nmain);
      if (nskip > 2) sissue(rs0, 2);
    }
  }
  if (SKIP) {
    auto unit_skip = [&](f32x4 (&ra)[NIS], int k) {  // skip chunk k: halo buffer (nmain + k) & 1, `ra` holds chunk k + 1
      const int g = 3 * nmain + k;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (g + 1 < G) wdma1(g + 1, 0);
      Frag f;
      ldf(f, sA + ((nmain + k) & 1) * ABYTES + WR * HRW, sB + (g & 1) * UB, aofs[1]);  // (the centre tap)
      mma(f);
      if (k + 1 < nskip) {
        scommit(ra, nmain + k + 1);
        if (k + 3 < nskip) sissue(ra, k + 3);
      }
    };
#pragma unroll 1
    for (int k = 0; k < nskip; k += 2) {
      unit_skip(rs1, k);
      if (k + 1 < nskip) unit_skip(rs0, k + 1);
    }
    if (!(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);
  }

  // ---- epilogue
  {
    const float qinv = SKIP ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
  }
  if (sample >= a.B) return;  // (wave-uniform; no barrier follows)
  if (a.small_check && a.range_flag) {
    float m = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][nt][r], acc[mt][nt][r + 1], m);
    hx_small_flag(a.range_flag, m);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp;
      float* op = a.out + (pix0 + p) * a.Cout + ch0;
      op[0] = acc[mt][0][r];
      op[32] = acc[mt][1][r];
    }
  if (a.stats_out) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[mt][nt][r];
      s += __shfl_xor(s, 32);
      const float mean = s / 64.f;
      float m2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[mt][nt][r] - mean;
          m2 += d * d;
        }
      m2 += __shfl_xor(m2, 32);
      if (hp == 0) store_stats(a, a.stats_out + (((size_t)sample * a.g.nparts + part) * a.Cout + ch0 + 32 * nt) * 2, mean, m2);
    }
  }
}

// ---------------------------------------------------------------- host side
// 0: off; 1: the eight-wave kernel everywhere; 2: the four-wave kernel everywhere (tools/kbench A/B); 3 (default): by
// layer shape -- never by batch, so a row's result cannot depend on its launch; the two cuts add the same products in the
// same order anyway.  Measured at B = 512 / 32 (tools/kbench, profiles/r04_kbench/hx2d_cuts.txt): with a fused 1x1 skip the
// four-wave cut wins everywhere (16x16, 128 channels, 256-channel skip: 149 vs 204 us); without one the eight-wave cut
// wins at 8x8 (28.9 vs 30.8 us; 19.4 vs 22.6 at 32 rows) and the four-wave cut at 16x16 (103.6 vs 107.2).
static int g_hx2d_on = 3;
void conv_hx2d_set(int v) { g_hx2d_on = v; }

static size_t hx2d4_lds_bytes(const ConvArgs& a) {
  const int npc = a.g.W == 8 ? 25 : 21;
  return (size_t)2 * npc * 1024 + (size_t)2 * 3 * 64 * HRW;
}

static size_t hx2d_lds_bytes(const ConvArgs& a) {
  const int npc = a.g.W == 8 ? 25 : 21;
  return (size_t)2 * npc * 1024 + (size_t)2 * 9 * 64 * HRW;
}

bool conv_hx2d_supported(const ConvArgs& a, int mode) {
  if (!g_hx2d_on || mode != CONV_S1) return false;
  if (!a.pin0 || !a.zeros || a.C1 != 0) return false;
  if (!a.wpkh || !a.hq || !a.range_flag) return false;
  const TileGeom& g = a.g;
  if (!((g.W == 8 && g.H == 8 && g.spt == 4) || (g.W == 16 && g.H == 16 && g.spt == 1 && g.tps == 1))) return false;
  if (a.Hin != g.H || a.Win != g.W) return false;
  if (a.Cout % 64 != 0 || a.C0 % KC != 0) return false;
  if (a.res_mode == 2 && ((a.R0 + a.R1) % KC != 0 || !a.wskiph || !a.hq_skip || (a.R0 % KC) != 0)) return false;
  if (a.res_mode == 1 && a.R0 != a.Cout) return false;
  if (a.ep_scale || a.fin_ab) return false;
  return true;
}

int conv_hx2d_init() {
  int rc = 0;
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d4_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d4_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d4_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2d4_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  return rc;
}

void launch_conv_hx2d(const ConvArgs& a, hipStream_t s) {
  const int tiles = geom_num_tiles(a.g, a.B);
  const dim3 grid(tiles, a.Cout / 64);
  const size_t lds = hx2d_lds_bytes(a);
  const bool skip = a.res_mode == 2;
  if (g_hx2d_on == 2 || (g_hx2d_on == 3 && (skip || a.g.W == 16))) {
    const size_t lds4 = hx2d4_lds_bytes(a);
    if (a.g.W == 8) {
      if (skip) hipLaunchKernelGGL((conv_mfma_hx2d4_kernel<8, true>), grid, dim3(256), lds4, s, a, tiles);
      else hipLaunchKernelGGL((conv_mfma_hx2d4_kernel<8, false>), grid, dim3(256), lds4, s, a, tiles);
    } else {
      if (skip) hipLaunchKernelGGL((conv_mfma_hx2d4_kernel<16, true>), grid, dim3(256), lds4, s, a, tiles);
      else hipLaunchKernelGGL((conv_mfma_hx2d4_kernel<16, false>), grid, dim3(256), lds4, s, a, tiles);
    }
    return;
  }
  if (a.g.W == 8) {
    if (skip) hipLaunchKernelGGL((conv_mfma_hx2d_kernel<8, true>), grid, dim3(512), lds, s, a, tiles);
    else hipLaunchKernelGGL((conv_mfma_hx2d_kernel<8, false>), grid, dim3(512), lds, s, a, tiles);
  } else {
    if (skip) hipLaunchKernelGGL((conv_mfma_hx2d_kernel<16, true>), grid, dim3(512), lds, s, a, tiles);
    else hipLaunchKernelGGL((conv_mfma_hx2d_kernel<16, false>), grid, dim3(512), lds, s, a, tiles);
  }
}

// ---------------------------------------------------------------- P format from an fp32 map (tools/kbench, tests, and the
// producers that cannot emit it themselves): out[pixel][chunk] = split(S_A silu(scale x + shift)) with the per-(sample,
// channel) scale/shift pairs of gn_finalize (ab[B][C][2]); one thread per (pixel, 4 channels)
__global__ __launch_bounds__(256) void hx_presplit_kernel(const float* in, const float* ab, void* pout, int B, int HW, int C) {
  const size_t n = (size_t)B * HW * (C / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % (C / 4));
    const size_t px = i / (C / 4);
    const int b = (int)(px / HW);
    const f32x4 v = *reinterpret_cast<const f32x4*>(in + px * C + q * 4);
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(ab + ((size_t)b * C + q * 4) * 2);
    const f32x4 e1 = *reinterpret_cast<const f32x4*>(ab + ((size_t)b * C + q * 4) * 2 + 4);
    f32x4 o;
    o.x = silu_scaled(fmaf(HX_SA * e0.x, v.x, HX_SA * e0.y));
    o.y = silu_scaled(fmaf(HX_SA * e0.z, v.y, HX_SA * e0.w));
    o.z = silu_scaled(fmaf(HX_SA * e1.x, v.z, HX_SA * e1.y));
    o.w = silu_scaled(fmaf(HX_SA * e1.z, v.w, HX_SA * e1.w));
    unsigned h0, l0, h1, l1;
    hsplit2(o.x, o.y, h0, l0);
    hsplit2(o.z, o.w, h1, l1);
    const int chunk = q >> 2, q4 = q & 3;
    char* rec = reinterpret_cast<char*>(pout) + (px * (C / KC) + chunk) * 64;
    const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
    *reinterpret_cast<hx_u32x2*>(rec + q4 * 8) = ph;        // h plane: channels 4 q4 .. 4 q4 + 3 of the chunk
    *reinterpret_cast<hx_u32x2*>(rec + 32 + q4 * 8) = pl;   // l plane
  }
}

void launch_hx_presplit(const float* in, const float* ab, void* pout, int B, int HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(hx_presplit_kernel, dim3(1024), dim3(256), 0, s, in, ab, pout, B, HW, C);
}

}  // namespace rgfm
