// FROZEN COPY (round 3, commit ef17e0f) of the kernel with its experiment scaffolding -- timing ablations (results wrong by
// construction), phase stamps, rejected cuts.  tools/kbench only (-DRGFM_KB_SCAFFOLD); the shipped kernel is csrc/'s.
// conv_mfma_hx2q.hip -- the two-plane fp16 conv (conv_mfma_hx2p.hip's arithmetic, LDS images and summation order, so
// a row's result is bit for bit the pipelined kernel's) cut for FOUR waves per SIMD and for workgroups that walk
// SEVERAL tiles with one continuous staging stream.
//
//   * A workgroup is one 256-pixel tile x 64 output channels at a time (8 waves = 4 pixel segments x 2 groups of 32
//     channels, one 32x32 accumulator pair per wave), at most 128 VGPRs and 80 KB of LDS: TWO workgroups share a CU.
//   * It processes `tpw` consecutive tiles (at 32x32: the four tiles of one sample).  What conv_mfma_hx2p_kernel pays
//     per tile is paid once: the GroupNorm scale/shift table (phase stamps: ~12 k cycles of dependent round trips for
//     ONE wave, seven waiting), the item decode, the chunk descriptors.
//   * The K loop is a STREAM of (tile, chunk) pairs: while chunk c of tile t is multiplied, the halo of the stream's
//     next chunk is transformed and stored and the one after that is fetched -- across tile boundaries, so a tile's
//     pipeline fill (~15 k cycles with the matrix pipe idle) runs under the previous tile's last chunks, and the store
//     tail of tile t drains under the first chunks of tile t + 1.  Every unit is straight-line code: what the stream
//     points at (source, tile origin, validity of the halo rows, table row, raw or normalised, which weight unit)
//     is DATA -- scalar selects, no branches -- so hipcc counts its vmcnt waits and the fetches stay in flight.
//     Staging past the end of the stream re-fetches the last chunk into the buffer nobody reads (harmless).
//
// Scope: stride-1 3x3 convs whose input takes the consumer-side GroupNorm, over 16- or 32-pixel-wide rasters whose
// tiles are whole (256 / W rows each), Cout % 64 == 0; everything else stays on conv_mfma_hx2p_kernel /
// conv_mfma_hx2_kernel (conv_hx2q_supported).  Same ConvArgs, same packed weights (a workgroup reads one 64-channel
// block, or its half of a 128-channel block).
#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

#ifndef RGFM_HX2Q_STAGGER
#define RGFM_HX2Q_STAGGER 1  // 1: waves 4-7 transform + store their halo items AFTER the unit's MFMAs (waves 0-3: before)
#endif
#ifndef RGFM_HX2Q_ABL
#define RGFM_HX2Q_ABL 0      // kbench timing ablations (results wrong): 1 no halo staging in the K loop, 2 nor weight DMA,
#endif                       // 3 nor fragment reads (MFMAs + barriers only), 4: everything but the MFMAs
#ifndef RGFM_HX2Q_PRIO
#define RGFM_HX2Q_PRIO 0     // 1: raised issue priority during a unit's MFMAs
#endif
#ifdef RGFM_HX2Q_PROF  // (tools/kbench: phase stamps of wave 0 / wave 4 of every 16th workgroup, summed)
__device__ unsigned long long g_hx2q_prof[16];
#define QPROF_T(var) const long long var = __builtin_amdgcn_s_memtime()
#define QPROF_ACC(i, t0, t1) qacc[i] += (t1) - (t0)
#else
#define QPROF_T(var)
#define QPROF_ACC(i, t0, t1)
#endif

// SKIP: res_mode == 2 (fused 1x1 skip conv: raw one-tap chunks behind the main chunks).  NG: 32-channel groups per
// workgroup -- 2: 8 waves x <= 128 VGPRs, two workgroups per CU = four waves per SIMD; 1 (Cout = 32: the MNIST net's
// 32x32 level): 4 waves, two workgroups per CU by LDS = two waves per SIMD, <= 256 VGPRs, twice the halo items per thread.
// NT: 32-channel accumulator columns per wave (1: a 64-pixel x 32-channel wave tile, 2: 64 x 64 as
// conv_mfma_hx2p_kernel -- two thirds of the fragment reads per MFMA, 256 VGPRs: one (NG = 2) or two (NG = 1)
// workgroups per CU).
template <int WL2, bool SKIP, int NG, int NT>
__global__ __launch_bounds__(256 * NG, (NG == 2 && NT == 1) ? 4 : 2) void conv_mfma_hx2q_kernel(const ConvArgs a, const int num_tiles, const int tpw, const int nrows) {
  constexpr int W = 1 << WL2, TH = 256 / W, WR = W + 2, HR = TH + 2, HALO = HR * WR;
  constexpr int ABYTES = (HALO + 1) * HRW;  // one halo buffer + a pad record (the store target of lanes past the halo)
  constexpr int NTHR = 256 * NG, NW = 4 * NG;
  constexpr int CB = 32 * NT * NG;          // output channels per workgroup
  constexpr int TAPB = CB * HRW;            // one tap's weight slab
  constexpr int UB = 3 * TAPB;              // one unit: a kernel row of a 16-channel chunk
  constexpr int PPT = TAPB / 1024;          // 1-KB DMA pieces per tap
  constexpr int MT_OFF = (32 / W) * WR * HRW;  // pixel p + 32 of a segment: one (W = 32) or two (W = 16) halo rows down, same column
  constexpr int NIT = 6 / NG;               // halo items (pixel, 4 channels) per thread and chunk: ceil(HALO * 4 / NTHR)
  static_assert(HALO * 4 <= NIT * NTHR && HALO * 4 > (NIT - 1) * NTHR - NTHR, "items per thread");
  extern __shared__ __attribute__((aligned(16))) char smq[];
#ifdef RGFM_HX2Q_PROF
  long long qacc[3] = {0, 0, 0};
#endif
  QPROF_T(tq0);
  char* const sB = smq + 2 * ABYTES;
  float* const sTab = reinterpret_cast<float*>(sB + 2 * UB);  // [nrows + 1][cin][2]: S_A x (scale, shift) per sample, and a zero row
  const int cin = a.C0 + a.C1;
  char* const sDesc = reinterpret_cast<char*>(sTab) + (nrows + 1) * cin * 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int l31 = lane & 31, hp = lane >> 5;
  const int H = a.g.H, tps = a.g.tps;

  // workgroup -> (tile group, 64-channel block)
  const int ncb = a.Cout / CB;
  const int tg = (int)blockIdx.x / ncb, cb = (int)blockIdx.x - tg * ncb;
  const int tile0 = tg * tpw;
  const int ntw = num_tiles - tile0 < tpw ? num_tiles - tile0 : tpw;  // tiles of this workgroup
  const int bfirst = tile0 / tps, rfirst = tile0 - bfirst * tps;

  // ---- consumer-side GroupNorm: scale/shift of this workgroup's sample(s) from the producers' partial statistics
  // (as conv_mfma_hx2p_kernel: one table row per sample; as many waves per row as it takes to give every lane one
  // channel, rows side by side on the waves)
  {
    const int gn_cpg = cin >> 3;
    int gn_wsh = (NG + 1) - (31 - __builtin_clz(nrows));  // log2 (waves / rows): nrows is 1, 2 or 4
    {
      const int need = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : 2);  // (cin <= 256)
      gn_wsh = gn_wsh < need ? gn_wsh : need;
    }
    const int gn_row = wave >> gn_wsh, gn_part = wave & ((1 << gn_wsh) - 1);
    const int gn_b = bfirst + gn_row;
    if (gn_row < nrows && gn_b < a.B) {
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
    for (int i = tid; i < 2 * cin; i += NTHR) sTab[nrows * cin * 2 + i] = 0.f;  // the all-zero row of the padding items
  }
  QPROF_T(tq1);

  const int nmain = cin / KC;
  const int nskip = SKIP ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nmain + nskip;
  const int G = 3 * nmain + nskip;
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

  // ---- fragment offsets.  A: this lane's pixel 64 seg + l31 (+ 32: MT_OFF) at tap column kx, planes h / l; the four
  // 16-byte slots of a halo record are swizzled with its halo column (conv_mfma_hx2p.hip)
  // (plane l of a record is plane h's slot ^ 2, i.e. the byte offset ^ 32: one register per column)
  int aofs[3];
  {
    const int p = 64 * seg + l31, r = p >> WL2, x = p & (W - 1);
    const int arec = r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int sw = ((x + kx) >> 2) & 3;
      aofs[kx] = (arec + kx) * HRW + ((hp ^ sw) & 3) * 16;
    }
  }
  int bofs;
  {
    const int rec = grp * 32 * NT + l31;  // (column nt: 32 records = 2048 bytes further, same swizzle)
    bofs = rec * HRW + ((hp ^ (rec >> 2)) & 3) * 16;
  }

  // ---- per-item decode, once per workgroup (tile-independent): pixel offset relative to the tile's first row, LDS
  // destination, and three flag bits above it -- 16: top halo row (outside the image in a sample's first tile),
  // 17: bottom halo row (outside in its last tile), 18: never valid (outside columns, lanes past the halo)
  const int q4 = tid & 3;
  unsigned meta[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int it = tid + NTHR * j;
    meta[j] = (unsigned)(HALO * HRW + (q4 >> 1) * 16 + (q4 & 1) * 8) | (1u << 18);
    if (it < HALO * 4) {
      const int hpx = it >> 2;
      const int hy = hpx / WR, hx = hpx - hy * WR;
      const int x = hx - 1;
      const bool xok = (x >= 0) && (x < W);
      meta[j] = (unsigned)(hpx * HRW + ((((q4 >> 1) ^ (hx >> 2)) & 3) * 16) + (q4 & 1) * 8)  // plane l: ^ 32
                | (hy == 0 ? 1u << 16 : 0u) | (hy == HR - 1 ? 1u << 17 : 0u) | (xok ? 0u : 1u << 18);
    }
  }

  // packed weights: [channel block][chunk][tap] slabs; a workgroup of a 128-channel block takes its 64-channel half
  // (the weights are packed in blocks of 128 / 64 / 32 channels -- hx2_block_channels; conv_hx2q_supported lets a
  // workgroup cover a whole block, or half of a 128-channel one)
  const bool nb128 = CB == 64 && (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  const char* const wsk = reinterpret_cast<const char*>(a.wskiph) + (size_t)wblk * nskip * TAPS + whalf;
  // this thread's two 16-byte weight items of a unit: 768 items in a main unit (three taps), 256 in a skip unit (one
  // tap); threads past the unit's end repeat an earlier item (same bytes to the same address)
  float hmax = 0.f;  // range flag: the largest |S_A a| this thread has split (two v_max3_f32 per item)
  f32x4 ra[NIT];

  // ---- the stream.  A position is (tile index inside the workgroup, chunk, sample, tile inside the sample); the
  // staging of a unit works on position + 1 (store) and position + 2 (fetch)
  struct Pos {
    int t, c, b, r;
  };
  auto advance = [&](Pos& p) {
    if (++p.c == ntot) {
      p.c = 0, ++p.t;
      if (++p.r == tps) p.r = 0, ++p.b;
    }
  };
  // is item j outside the image at position p?  (its flag bits against the tile's place in the sample; a position
  // past the workgroup's last tile invalidates every item: the stream's harmless overrun)
  auto item_bad = [&](const Pos& p, int j) -> bool {
    const unsigned mask = 0x40000u | (p.r == 0 ? 0x10000u : 0u) | (p.r == tps - 1 ? 0x20000u : 0u);
    return p.t >= ntw || (meta[j] & mask) != 0u;
  };
  auto issue_a = [&](const Pos& p, int j) {
    const u32x4 d = *reinterpret_cast<const u32x4*>(sDesc + p.c * 16);
    const float* src = reinterpret_cast<const float*>(((unsigned long long)d.y << 32) | (unsigned long long)d.x);
    const bool bad = item_bad(p, j);
    const int hpx = (int)(meta[j] & 0xffffu) >> 6;  // the halo record: (hy, hx) -> pixel (hy - 1, hx - 1) of the tile
    const int hy = (int)(__umul24((unsigned)hpx, (65536u + WR - 1u) / WR) >> 16);
    const int pix = (p.b * H + p.r * TH - 1) * W - 1 + hy * (W - WR) + hpx;
    const unsigned po = bad ? 0u : (unsigned)pix;
    ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24(po, d.z) + (unsigned)(q4 * 4)));
  };
  // GroupNorm + SiLU (main chunks) or the plain scale (1x1-skip chunks) + split + store of item j of position p
  // (XF: position p is a main chunk -- known where the call stands, so there is no per-element select or branch)
  auto commit_a = [&](auto xf_tag, const Pos& p, int gcn, int j) {
    constexpr bool XF = decltype(xf_tag)::value;
    const bool bad = item_bad(p, j);
    const f32x4 v = ra[j];
    f32x4 o;
    if (XF) {
      const int row = bad ? nrows : p.b - bfirst;
      const char* ep = reinterpret_cast<const char*>(sTab) + (row * cin + p.c * KC + 4 * q4) * 8;
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 16);
      o.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
      o.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
      o.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
      o.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
    } else {
      const float sa = bad ? 0.f : HX_SA;
      o.x = v.x * sa, o.y = v.y * sa, o.z = v.z * sa, o.w = v.w * sa;
    }
    unsigned h0, l0, h1, l1;
    hsplit2(o.x, o.y, h0, l0);
    hsplit2(o.z, o.w, h1, l1);
    hmax = hx_absmax3(o.x, o.y, hmax);
    hmax = hx_absmax3(o.z, o.w, hmax);
    const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
    char* base = smq + (gcn & 1) * ABYTES;
    const int ad = (int)(meta[j] & 0xffffu);
    *reinterpret_cast<hx_u32x2*>(base + ad) = ph;
    *reinterpret_cast<hx_u32x2*>(base + (ad ^ 32)) = pl;
  };
  // weights of unit u (0 .. G - 1 inside a tile; the stream wraps) -> weight buffer gun & 1, by LDS-DMA: the packed image
  // IS the LDS byte image, so a unit is twelve (a skip unit's one tap: four) linear 1-KB pieces, one
  // global_load_lds_dwordx4 wave-instruction each.  Every wave issues two (the surplus ones repeat a piece: same bytes to
  // the same address).  No registers, no ds_write; hipcc does not see these loads, so their completion is waited for by
  // hand (wdma_wait) before the barrier that precedes the unit -- hipcc's own counted waits for the halo fetches then
  // only ever over-wait.
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned sB_lds = (unsigned)(size_t)sB;
  auto wdma = [&](int u, int gun) {
    const bool main = u < 3 * nmain;
    const char* src = main ? wpk + (size_t)u * 3 * TAPS : wsk + (size_t)(u - 3 * nmain) * TAPS;
    constexpr int NPW = (3 * PPT + NW - 1) / NW;  // pieces per wave
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pq = wave_s + NW * j;
      const int pc = main ? (pq < 3 * PPT ? pq : pq - 3 * PPT) : (pq & (PPT - 1));
      const char* gsrc = src + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16;
      const unsigned dst = sB_lds + (unsigned)((gun & 1) * UB + pc * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };
#define HX2Q_WDMA_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")  // all but the N youngest vector-memory operations
  auto next_unit = [&](int u) { return u + 1 == G ? 0 : u + 1; };

  __syncthreads();  // the scale/shift table and the chunk descriptors are complete
  QPROF_T(tq2);

  // ---- pipeline fill: halo of the first chunk and weights of unit 0 in LDS, raw halo of the second chunk and weights
  // of unit 1 in registers
  Pos cur = {0, 0, bfirst, rfirst};
  auto succ = [&](Pos p) {
    advance(p);
    return p;
  };
  int gc = 0, gu = 0;           // stream counters (buffer parities)
  int u1 = next_unit(0);        // the unit after the current one (inside a tile: 0 .. G - 1, wrapping)
  {
#pragma unroll
    for (int j = 0; j < NIT; ++j) issue_a(cur, j);
    wdma(0, 0);
#pragma unroll
    for (int j = 0; j < NIT; ++j) commit_a(std::true_type{}, cur, 0, j);
    const Pos p1 = succ(cur);
#pragma unroll
    for (int j = 0; j < NIT; ++j) issue_a(p1, j);
    if (NIT == 3) HX2Q_WDMA_WAIT(3);  // (the fetches just issued stay in flight)
    else HX2Q_WDMA_WAIT(6);
  }

  // the scales and (when every row shares the time value: the samplers) the per-channel additive term are the same for
  // every tile of the workgroup: loaded ONCE (scalar registers / one VGPR per column) -- re-derived per tile they were two
  // to three dependent memory round trips in front of every tile's first MFMA and behind its last
  // (NG == 1 only -- the cut with registers to spare: +1 ... 2 % on its layers; in the 128-register cut the extra live
  // values spill and the layers lose 2 ... 3 %: tools/kbench/scripts/q19.sh)
  constexpr bool HOIST = NG == 1;
  const float qmain = a.hq[0];
  const float qinv_all = HOIST ? (SKIP ? a.hq_skip[1] : a.hq[1]) : 0.f;
  const bool add_const = HOIST && !(a.temb && a.temb_per_row);
  float add_all[NT];
#pragma unroll
  for (int nt = 0; HOIST && nt < NT; ++nt) {
    const int c = cb * CB + grp * 32 * NT + l31 + nt * 32;
    float v = a.bias[c];
    if (SKIP) v += a.skip_bias[c];
    if (a.temb && !a.temb_per_row) v += a.temb[(a.step_ptr ? (size_t)*a.step_ptr : 0) * a.temb_stride + c];
    add_all[nt] = v * qmain;
  }
  f32x16 acc[2][NT];
  // one tap (kernel column KX of the halo row at sArow): 6 fragment reads, 6 MFMAs (a_l w_h, a_h w_l, a_h w_h per tile)
  auto tap = [&](const char* sArow, const char* sBt, int o0) {
    f16x8 af[2][2], bf[NT][2];
    const int o1 = o0 ^ 32;
#if RGFM_HX2Q_ABL == 3
    af[0][0] = af[0][1] = af[1][0] = af[1][1] = __builtin_bit_cast(f16x8, ra[0]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[nt][0] = bf[nt][1] = af[0][0];
    (void)sArow, (void)sBt, (void)o1;
    if (false)
#endif
    {
    af[0][0] = *reinterpret_cast<const f16x8*>(sArow + o0);
    af[0][1] = *reinterpret_cast<const f16x8*>(sArow + o1);
    af[1][0] = *reinterpret_cast<const f16x8*>(sArow + o0 + MT_OFF);
    af[1][1] = *reinterpret_cast<const f16x8*>(sArow + o1 + MT_OFF);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf[nt][0] = *reinterpret_cast<const f16x8*>(sBt + bofs + nt * 32 * HRW);
      bf[nt][1] = *reinterpret_cast<const f16x8*>(sBt + (bofs ^ 32) + nt * 32 * HRW);
    }
    }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#if RGFM_HX2Q_ABL == 4
    asm volatile("" :: "v"(af[0][0]), "v"(af[0][1]), "v"(af[1][0]), "v"(af[1][1]), "v"(bf[0][0]), "v"(bf[0][1]));
#else
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bf[nt][PB[q]], acc[mt][nt], 0, 0, 0);
#endif
  };
  auto taps3 = [&](int U) {
    const char* sArow = smq + (gc & 1) * ABYTES + U * WR * HRW;
    const char* sBu = sB + (gu & 1) * UB;
#if RGFM_HX2Q_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    tap(sArow, sBu, aofs[0]);
    tap(sArow, sBu + TAPB, aofs[1]);
    tap(sArow, sBu + 2 * TAPB, aofs[2]);
#if RGFM_HX2Q_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // the weight half of a unit's staging: the next unit's weights -> the other weight buffer (free since the last barrier)
#define HX2Q_W_STEP()                       \
  do {                                      \
    wdma(u1, gu + 1);                       \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
  // end of a unit: this wave's DMA pieces have landed (N: the halo fetches issued behind them stay in flight), barrier
#define HX2Q_U_NEXT(N)                      \
  do {                                      \
    HX2Q_WDMA_WAIT(N);                      \
    __syncthreads();                        \
    ++gu, u1 = next_unit(u1);               \
  } while (0)

  QPROF_T(tq3);
#pragma unroll 1
  for (int t = 0; t < ntw; ++t) {
    QPROF_T(tt0);
    // ---- accumulator init: bias (+ skip bias + time embedding), scaled by q (the accumulators hold q x the true
    // sums); an identity residual enters as fma(res, q, .)
    const int tb = cur.b, tr = cur.r;  // this tile: sample, tile inside the sample
    const size_t pix0 = ((size_t)tb * H + (size_t)tr * TH) * W;
    // What only the tile boundary needs is re-derived at every boundary instead of living in registers through the K
    // loop: the arguments are re-read from the kernel-argument segment, the per-lane offsets recomputed from the thread
    // index (pointer and index are opaque to hipcc per tile, so nothing here is loop-invariant to it).
    const ConvArgs* kp = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    int tid_b = threadIdx.x;
    asm volatile("" : "+v"(tid_b));
    const int seg_b = (tid_b >> 6) & 3, l31_b = tid_b & 31, hp_b = (tid_b >> 5) & 1;
    const int ch_b = cb * CB + (tid_b >> 8) * 32 * NT + l31_b;  // this lane's first output channel (column nt: + 32 nt)
    const ConvArgs& ka = *kp;
    {
      float add0[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (add_const) {
          add0[nt] = add_all[nt];
        } else {  // (a time value per row -- rgfm_unet_forward --, or the 128-register cut)
          const int c = ch_b + nt * 32;
          float v = ka.bias[c];
          if (SKIP) v += ka.skip_bias[c];
          if (ka.temb) v += ka.temb[((size_t)(ka.temb_per_row ? tb : 0) + (ka.step_ptr ? (size_t)*ka.step_ptr : 0)) * ka.temb_stride + c];
          add0[nt] = v * qmain;
        }
      }
      if (!SKIP && ka.res_mode == 1) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int p = 64 * seg_b + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp_b;
            const float* rp = ka.res0 + (size_t)(__umul24((unsigned)pix0 + (unsigned)p, (unsigned)ka.Cout) + (unsigned)ch_b);
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
    if (t == 0) __syncthreads();  // (the fill's stores)
    QPROF_T(tt1);
    // ---- main chunks: three units each.  Item 0 of the stream's next chunk is stored (and of the one after fetched)
    // in unit 0, items 1 and 2 in unit 1, none in unit 2; the weights FIRST, so that the wait for them at the top of the
    // next unit (vmcnt is in order) leaves the slower halo fetches behind them in flight
    // (the chunk the stream stores next is a main chunk, except behind the last main chunk of a conv with skip chunks)
#if RGFM_HX2Q_ABL >= 1 && RGFM_HX2Q_ABL <= 3
#define commit_a(...) ((void)0)
#define issue_a(...) ((void)0)
#endif
#if RGFM_HX2Q_ABL >= 2 && RGFM_HX2Q_ABL <= 3
#define wdma(...) ((void)0)
#endif
    auto main_chunk = [&](auto xf_tag) {
      const Pos p1 = succ(cur), p2 = succ(p1);
      (void)p2;
#if RGFM_HX2Q_STAGGER
      const bool early = NG == 2 ? grp == 0 : ((int)blockIdx.x & 1) == 0;
#else
      constexpr bool early = true;
#endif
      // the halo items of a unit.  NG == 2 (three items): item 0 in unit 0, items 1 and 2 in unit 1, none in unit 2
      // (fewer staging sites: the 128-VGPR budget); NG == 1 (six items): j % 3 == U
      auto commit_items = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        if constexpr (NG == 2) {
          if (U == 0) commit_a(xf_tag, p1, gc + 1, 0);
          if (U == 1) commit_a(xf_tag, p1, gc + 1, 1), commit_a(xf_tag, p1, gc + 1, 2);
        } else {
          commit_a(xf_tag, p1, gc + 1, U), commit_a(xf_tag, p1, gc + 1, U + 3);
        }
      };
      auto issue_items = [&](auto u_tag) {
        constexpr int U = decltype(u_tag)::value;
        if constexpr (NG == 2) {
          if (U == 0) issue_a(p2, 0);
          if (U == 1) issue_a(p2, 1), issue_a(p2, 2);
        } else {
          issue_a(p2, U), issue_a(p2, U + 3);
        }
      };
#define HX2Q_UNIT(U)                                             \
  do {                                                           \
    using UT = std::integral_constant<int, (U)>;                 \
    if (NG == 2 && (U) == 2) { /* no halo items in this unit */  \
      HX2Q_W_STEP();                                             \
      taps3(U);                                                  \
    } else if (early) {                                          \
      commit_items(UT{});                                        \
      HX2Q_W_STEP();                                             \
      issue_items(UT{});                                         \
      __builtin_amdgcn_sched_barrier(0);                         \
      taps3(U);                                                  \
    } else {                                                     \
      HX2Q_W_STEP();                                             \
      taps3(U);                                                  \
      __builtin_amdgcn_sched_barrier(0);                         \
      commit_items(UT{});                                        \
      issue_items(UT{});                                         \
    }                                                            \
    if (NG == 1 || (U) == 1) HX2Q_U_NEXT(2);                     \
    else if ((U) == 0) HX2Q_U_NEXT(1);                           \
    else HX2Q_U_NEXT(0);                                         \
  } while (0)
      HX2Q_UNIT(0);
      HX2Q_UNIT(1);
      HX2Q_UNIT(2);
#undef HX2Q_UNIT
      ++gc, cur = p1;
    };
#pragma unroll 1
    for (int c = 0; c < nmain - 1; ++c) main_chunk(std::true_type{});
    if constexpr (!SKIP) {
      main_chunk(std::true_type{});
    } else {  // the 1x1 skip weights carry their own scale: q_main -> q_skip; one unit per chunk (the centre tap)
      main_chunk(std::false_type{});
      const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
      auto skip_chunk = [&](auto xf_tag) {
        const Pos p1 = succ(cur), p2 = succ(p1);
#pragma unroll
        for (int j = 0; j < NIT; ++j) commit_a(xf_tag, p1, gc + 1, j);
        HX2Q_W_STEP();
#pragma unroll
        for (int j = 0; j < NIT; ++j) issue_a(p2, j);
        __builtin_amdgcn_sched_barrier(0);
        tap(smq + (gc & 1) * ABYTES + WR * HRW, sB + (gu & 1) * UB, aofs[1]);
        if (NIT == 3) HX2Q_U_NEXT(3);
        else HX2Q_U_NEXT(6);
        ++gc, cur = p1;
      };
#pragma unroll 1
      for (int c = 0; c < nskip - 1; ++c) skip_chunk(std::false_type{});
      skip_chunk(std::true_type{});  // (behind the last skip chunk: the next tile's first chunk)
    }
    QPROF_T(tt2);
    asm volatile("" : "+s"(kp));
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int seg_e = (tid_e >> 6) & 3, l31_e = tid_e & 31, hp_e = (tid_e >> 5) & 1, lane_e = tid_e & 63;
    const int ch_e = cb * CB + (tid_e >> 8) * 32 * NT + l31_e;
    const ConvArgs& ke = *kp;
    {
      const float qinv = HOIST ? qinv_all : (SKIP ? ke.hq_skip[1] : ke.hq[1]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
    }
    // (ConvArgs::small_check: the output's low range.  HERE, while nothing but the accumulators is live: behind the stores
    // and the statistics it cost the 128-register cut 33 more spilled registers)
    if (ke.small_check && ke.range_flag) {
      float m = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; r += 2) m = hx_absmax3(acc[mt][nt][r], acc[mt][nt][r + 1], m);
      hx_small_flag(ke.range_flag, m);
    }
    // ---------------------------------------------------------------- epilogue: every pixel of the tile is valid
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = 64 * seg_e + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp_e;
        float* op = ke.out + (size_t)(__umul24((unsigned)pix0 + (unsigned)p, (unsigned)ke.Cout) + (unsigned)ch_e);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) op[nt * 32] = acc[mt][nt][r];
      }
    if (ke.stats_out) {
      const int nparts = ke.g.nparts;
      const int part = tr * 4 + seg_e;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
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
        if (hp_e == 0) store_stats(ke, ke.stats_out + (((size_t)tb * nparts + part) * ke.Cout + ch_e + nt * 32) * 2, mean, m2);
      }
      if (ke.fin_ab) fin_arrive(ke, tb, lane_e, nparts, false);
    }
    QPROF_T(tt3);
    QPROF_ACC(0, tt0, tt1);
    QPROF_ACC(1, tt1, tt2);
    QPROF_ACC(2, tt2, tt3);
  }
#undef HX2Q_W_STEP
#undef HX2Q_U_NEXT
#if RGFM_HX2Q_ABL >= 1 && RGFM_HX2Q_ABL <= 3
#undef commit_a
#undef issue_a
#endif
#if RGFM_HX2Q_ABL >= 2 && RGFM_HX2Q_ABL <= 3
#undef wdma
#endif
#undef HX2Q_WDMA_WAIT
  if (!(hmax < HX_BIG)) atomicOr(a.range_flag, 1u);  // (rare) plane h would be >= 32768 (or inf)
#ifdef RGFM_HX2Q_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  QPROF_T(tq7);
  if (lane == 0 && seg == 0 && (blockIdx.x & 15) == 0) {
    unsigned long long* pp = g_hx2q_prof + grp * 8;
    atomicAdd(pp + 0, (unsigned long long)(tq1 - tq0));  // GroupNorm table
    atomicAdd(pp + 1, (unsigned long long)(tq2 - tq1));  // descriptors, offsets, item decode, barrier
    atomicAdd(pp + 2, (unsigned long long)(tq3 - tq2));  // fill
    atomicAdd(pp + 3, (unsigned long long)qacc[0]);      // accumulator init (all tiles)
    atomicAdd(pp + 4, (unsigned long long)qacc[1]);      // K loops
    atomicAdd(pp + 5, (unsigned long long)qacc[2]);      // epilogues
    atomicAdd(pp + 6, (unsigned long long)(tq7 - tq0));  // whole workgroup
    atomicAdd(pp + 7, 1ull);
  }
#endif
}

// ---------------------------------------------------------------- host side
// launches with fewer (tile, channel block) pairs than this stay on conv_mfma_hx2p_kernel (0: this kernel never runs):
// at two workgroups per CU the stream needs >= 2 tiles per workgroup to pay (tools/kbench: B = 128 rows lose)
static int g_hx2q_min = 1024;
void conv_hx2q_set_min(int v) { g_hx2q_min = v; }
static int g_hx2q_target = 512;  // workgroups a launch is cut into when it has the tiles: two per CU
void conv_hx2q_set_target(int v) { g_hx2q_target = v > 0 ? v : 1; }
static int g_hx2q_all = 0;       // tools/kbench: every supported shape, not only those where this kernel is the faster one
void conv_hx2q_set_all(int v) { g_hx2q_all = v; }
static int g_hx2q_tpw = 0;       // tools/kbench: force the tiles per workgroup (0: hx2q_tiles_per_wg)
void conv_hx2q_set_tpw(int v) { g_hx2q_tpw = v; }
static int g_hx2q_cut = 0;       // tools/kbench: force the workgroup cut, 10 NG + NT (0: hx2q_cut)
void conv_hx2q_set_cut(int v) { g_hx2q_cut = v; }

// The workgroup cut {NG 32 NT-channel groups, NT accumulator columns per wave} -> channels per workgroup 32 NG NT:
//   {1,1} 32 ch, 4 waves, two workgroups per CU (LDS)      {2,1} 64 ch, 8 waves x 128 VGPRs, two workgroups per CU
//   {1,2} 64 ch, 4 waves x 256 VGPRs, two workgroups per CU  {2,2} 128 ch, 8 waves x 256 VGPRs, one workgroup per CU
struct Hx2qCut {
  int ng, nt;
  int cb() const { return 32 * ng * nt; }
};
static Hx2qCut hx2q_cut(const ConvArgs& a) {
  if (g_hx2q_cut) return {g_hx2q_cut / 10, g_hx2q_cut % 10};
  if (a.Cout % 64 != 0) return {1, 1};
  return {2, 1};
}
// tiles per workgroup: 1, 2 or 4 -- whole samples or whole fractions of one -- as many as leave `target` workgroups
// (half of it for the one-workgroup-per-CU cut)
static int hx2q_tiles_per_wg(const ConvArgs& a) {
  const Hx2qCut c = hx2q_cut(a);
  const int tiles = geom_num_tiles(a.g, a.B), ncb = a.Cout / c.cb(), tps = a.g.tps;
  if (g_hx2q_tpw) return g_hx2q_tpw;
  const int target = (c.ng == 2 && c.nt == 2) ? g_hx2q_target / 2 : g_hx2q_target;
  int tpw = 1;
  while (tpw * 2 <= 4 && (tiles / (tpw * 2)) * ncb >= target && ((tpw * 2) % tps == 0 || tps % (tpw * 2) == 0)) tpw *= 2;
  return tpw;
}
static int hx2q_rows(const ConvArgs& a, int tpw) { return tpw > a.g.tps ? tpw / a.g.tps : 1; }

static size_t hx2q_lds_bytes(const ConvArgs& a, int tpw) {
  const int W = a.g.W, halo = (256 / W + 2) * (W + 2);
  size_t bytes = (size_t)2 * (halo + 1) * HRW + (size_t)2 * 3 * hx2q_cut(a).cb() * HRW;
  bytes += (size_t)(hx2q_rows(a, tpw) + 1) * (a.C0 + a.C1) * 2 * sizeof(float);
  bytes += (size_t)(a.C0 + a.C1 + (a.res_mode == 2 ? a.R0 + a.R1 : 0));  // 16 bytes per 16-channel chunk: descriptors
  return bytes;
}

bool conv_hx2q_supported(const ConvArgs& a, int mode) {
  if (!g_hx2q_min) return false;
  if (mode != CONV_S1) return false;
  if (!a.gn_stats0) return false;  // convs of raw inputs (the upsamplers) stay on conv_mfma_hx2p_kernel
  if (!conv_hx2_supported(a, mode) || !conv_hx2_gn_supported(a, mode)) return false;
  const TileGeom& g = a.g;
  if (g.spt != 1 || (g.W != 16 && g.W != 32) || g.th * g.W != 256 || g.H % g.th != 0) return false;
  if (a.Hin != g.H || a.Win != g.W) return false;
  if ((a.C0 + a.C1) % KC != 0) return false;
  if (a.res_mode == 2 && (a.R0 + a.R1) % KC != 0) return false;
  const Hx2qCut c = hx2q_cut(a);
#ifndef RGFM_HX2Q_ALL_CUTS
  if (c.nt != 1) return false;
#endif
  // a workgroup covers one packed weight block (128 / 64 / 32 channels: hx2_block_channels) or half of a 128-channel one
  const int nb = a.Cout % 128 == 0 ? 128 : (a.Cout % 64 == 0 ? 64 : 32);
  if (a.Cout % c.cb() != 0 || !(c.cb() == nb || (c.cb() == 64 && nb == 128))) return false;
  const int tpw = hx2q_tiles_per_wg(a);
  if (hx2q_lds_bytes(a, tpw) > ((c.ng == 2 && c.nt == 2) ? 160 : 80) * 1024) return false;
  if (geom_num_tiles(g, a.B) * (a.Cout / c.cb()) < g_hx2q_min) return false;
  {  // the GroupNorm prologue must cut a row over the same number of waves as conv_mfma_hx2p_kernel (bit-identical tables)
    const int rows = hx2q_rows(a, tpw), cpg = (a.C0 + a.C1) / 8;
    const int need = cpg <= 8 ? 1 : (cpg <= 16 ? 2 : 4);
    if (4 * c.ng / rows < need) return false;
  }
  // Where it pays (tools/kbench, same box, B = 512): 64 -> 64 at 32x32 +9..10 %, 128 -> 64 +1 %; with a fused 1x1 skip
  // (192 -> 64: -1.5 %), at 16x16 (-3 %) and with Cout = 128 (two workgroups per tile transform the halo twice:
  // -3..-8 %) conv_mfma_hx2p_kernel is faster.  g_hx2q_all (kbench) lifts the restriction.
  if (g_hx2q_all) return true;
  // Cout = 32 (the MNIST net's 32x32 level, where a tile has 6-18 units and the per-tile fixed costs dominate):
  // 32 -> 32 +25..30 %, 64 -> 32 +16 %, 96 -> 32 +2.5 % at B = 512; +10 % / 0 at B = 256
  if (a.Cout % 64 != 0) return g.W == 32;
  return g.W == 32 && a.Cout == 64 && a.C0 + a.C1 <= 128 && a.res_mode != 2;
}

// The cuts with 64-channel wave tiles are built for tools/kbench only (-DRGFM_HX2Q_ALL_CUTS): measured against
// conv_mfma_hx2p_kernel over every layer shape of the two U-Nets at B = 512 (tools/kbench/scripts/q11.sh), {1,2} is
// 2..25 % slower everywhere and {2,2} -- conv_mfma_hx2p's own tile behind the tile stream -- within +-2 %: the stream
// pays where a tile's K loop is short (Cout <= 64 at 32x32), not where hx2p's interleaved units already hide the
// per-tile costs.
#ifdef RGFM_HX2Q_ALL_CUTS
#define HX2Q_FOR_ALL(X) \
  X(4, false, 1, 1) X(4, true, 1, 1) X(5, false, 1, 1) X(5, true, 1, 1) \
  X(4, false, 2, 1) X(4, true, 2, 1) X(5, false, 2, 1) X(5, true, 2, 1) \
  X(4, false, 1, 2) X(4, true, 1, 2) X(5, false, 1, 2) X(5, true, 1, 2) \
  X(4, false, 2, 2) X(4, true, 2, 2) X(5, false, 2, 2) X(5, true, 2, 2)
#else
#define HX2Q_FOR_ALL(X) \
  X(4, false, 1, 1) X(4, true, 1, 1) X(5, false, 1, 1) X(5, true, 1, 1) \
  X(4, false, 2, 1) X(4, true, 2, 1) X(5, false, 2, 1) X(5, true, 2, 1)
#endif

int conv_hx2q_init() {
  int rc = 0;
#define RAISEQ(WL, SK, G, T)                                                                               \
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2q_kernel<WL, SK, G, T>),     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, ((G) == 2 && (T) == 2 ? 160 : 80) * 1024);
  HX2Q_FOR_ALL(RAISEQ)
#undef RAISEQ
  return rc;
}

void launch_conv_hx2q(const ConvArgs& a_in, int mode, hipStream_t s) {
  (void)mode;
  ConvArgs a = a_in;
  const int tiles = geom_num_tiles(a.g, a.B);
  const Hx2qCut c = hx2q_cut(a);
  if (a.fin_ab) a.fin_expected = a.g.tps * 4 * (a.Cout / (32 * c.nt));  // every wave of every tile of a sample arrives
  const int tpw = hx2q_tiles_per_wg(a), nrows = hx2q_rows(a, tpw);
  const dim3 grid(((tiles + tpw - 1) / tpw) * (a.Cout / c.cb()));
  const size_t lds = hx2q_lds_bytes(a, tpw);
  const int wl = a.g.W == 32 ? 5 : 4;
  const bool sk = a.res_mode == 2;
#define LAUNCHQ(WL, SK, G, T)                                                                                         \
  if (wl == (WL) && sk == (SK) && c.ng == (G) && c.nt == (T))                                                         \
    hipLaunchKernelGGL((conv_mfma_hx2q_kernel<WL, SK, G, T>), grid, dim3(256 * (G)), lds, s, a, tiles, tpw, nrows);
  HX2Q_FOR_ALL(LAUNCHQ)
#undef LAUNCHQ
}
#undef HX2Q_FOR_ALL

}  // namespace rgfm
