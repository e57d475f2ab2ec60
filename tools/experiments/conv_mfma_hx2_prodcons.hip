// conv_mfma_hx2p.hip -- producer / consumer version of conv_mfma_hx2.hip for the stride-1 and upsampling convs
// (CONV_S1, CONV_UP2: 96 % of the conv time of a U-Net evaluation).  Same arithmetic (two scaled fp16 planes,
// three f16-MFMA products per fp32 product), same tiling, prologue arithmetic and epilogue; what changes is who
// does what, so that the matrix pipe never waits for staging:
//
//   * 768 threads = 12 waves = 3 per SIMD.  Waves 0-7 are the CONSUMERS: the eight 64-pixel x 32NT-channel MFMA
//     tiles of conv_mfma_hx2_kernel, two per SIMD, whose K loop is nothing but fragment reads + MFMAs.  Waves 8-11
//     are the PRODUCERS, one per SIMD: they fetch, GroupNorm+SiLU-transform, split and store the activation halo
//     and stream the weights, in the VALU / memory issue slots the MFMAs leave free (an MFMA holds the SIMD's
//     issue port for 8 of its 32 cycles).
//   * The activation halo of a 16-channel chunk is double-buffered in LDS: chunk c+1 is staged while chunk c is
//     multiplied.  Its raw fp32 loads are issued a whole chunk earlier still and wait in registers.
//   * Weights never touch registers: the packed image (launch_pack_conv_hx2) IS the LDS byte image, so it is
//     streamed by LDS-DMA (global_load_lds_dwordx4) into a ring of four 3-tap slots, two slots ahead of use.
//   * One s_barrier per UNIT (3 taps of one chunk = 36 MFMAs per consumer wave; a 1x1-skip chunk is a one-tap
//     unit).  Before the barrier that ends unit g the producers have waited for the weights of unit g+2 and for
//     their own LDS stores of the slice of chunk c+1 they staged during the unit; after it the consumers may read
//     both.  The slot of unit g-1 and the halo buffer of chunk c-1 are then free for the producers.
//
// The external scale/shift array path (ConvArgs::ab without gn_stats0) and the stride-2 / transposed modes stay on
// conv_mfma_hx2_kernel.  DESIGN.md section 4 has the measurements.
#include <stdlib.h>

#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

typedef __attribute__((address_space(3))) void hx_lds_void;
typedef __attribute__((address_space(1))) const void hx_gbl_void;

#ifdef RGFM_HX2P_PROF
// consumer wave 0: [0] prologue, [1] tap loops, [2] barrier waits, [3] epilogue; producer wave 8: [4] table + fill,
// [5] stage, [6] issue (LDS-DMA + raw loads), [7] vmcnt wait, [8] barrier wait; [9] blocks
__device__ unsigned long long g_hx2p_prof[10];
#define PPROF_T(var) const long long var = clock64()
#define PPROF_ADD(slot, t0, t1) pacc[slot] += (t1) - (t0)
#else
#define PPROF_T(var)
#define PPROF_ADD(slot, t0, t1)
#endif


template <int NT, int MODE, bool PAIRN, int RS>
__global__ __launch_bounds__(768, 3) void conv_mfma_hx2p_kernel(const ConvArgs a, const int num_tiles) {
  static_assert(MODE == CONV_S1 || MODE == CONV_UP2, "stride-2 / transposed convs run on conv_mfma_hx2_kernel");
  constexpr int NG = PAIRN ? 2 : 1;            // channel groups per block
  constexpr int NA = PAIRN ? 1 : 2;            // pixel tiles per block
  constexpr int NBLK = 32 * NT;                // channels per group
  constexpr int NBT = NBLK * NG;               // channels per block
  constexpr int TAPB = NBT * HRW;              // bytes of one tap's weight slab
  constexpr int UB = 3 * TAPB;                 // one ring slot = one unit = 3 taps
  constexpr int NPIECE = UB / 1024;            // 1 KB LDS-DMA pieces per slot (24 / 12 / 6)
  constexpr int GB = (NPIECE + 3) / 4;         // LDS-DMA instructions per producer wave and unit
  constexpr int NSLOT = PAIRN ? 3 : 2;         // weight ring slots (what fits beside three halo buffers)
  constexpr int NIT = 3 * RS;                  // halo items (pixel, 4 channels) per producer thread: RS per unit of a chunk
  extern __shared__ __attribute__((aligned(16))) char smemp[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;
  const int HR = g.th + 2, WR = W + 2;
  const int abytes = ((NA * a.halo_px + 15) >> 4) * 1024;  // one halo buffer: whole 1 KB (16-record) DMA pieces
  char* const sB = smemp + 3 * abytes;                     // weight ring
  float* const sTab = reinterpret_cast<float*>(sB + NSLOT * UB);  // [NA * spt][cin][2] S_A x (scale, shift)

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

  const int cin = a.C0 + a.C1;
  const int nmain = cin / KC;                                          // 16-channel chunks of the input
  const int nskip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;         // 1x1-skip chunks (one tap each)
  const int ntot = nmain + nskip;
  const int G = 3 * nmain + nskip;                                     // units
  const unsigned mW = (65536u + (unsigned)g.W - 1u) / (unsigned)g.W;

#ifdef RGFM_HX2P_PROF
  long long pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PPROF_T(tk0);
  if (wave >= 8) {
    // =============================================================================== PRODUCERS
    const int pw = wave - 8, pt = tid - 512, q4 = pt & 3;
    // The producer is the youngest wave of its SIMD and would lose every issue arbitration to the two MFMA waves
    // (measured: ~25 cycles per VALU instruction, the consumers then wait for it at every barrier); its stream is a
    // fifth of the SIMD's issue capacity, so it goes first whenever it has something to issue.
    __builtin_amdgcn_s_setprio(3);
    const unsigned mWR = (65536u + (unsigned)g.W + 1u) / (unsigned)(g.W + 2);
    const unsigned mPER = (65536u + (unsigned)((g.th + 2) * (g.W + 2)) - 1u) / (unsigned)((g.th + 2) * (g.W + 2));

    if (a.gn_stats0) {
      // ---- consumer-side GroupNorm (see conv_mfma_bx3.hip): table row = ga * spt + s, one wave per row, lane = group * 8 + sub
      for (int row = pw; row < NA * g.spt; row += 4) {
        const int ga = (g.spt == 1) ? row : (row >> 2), sl = (g.spt == 1) ? 0 : (row & 3);
        const int b = (ga ? tb0_[1] : tb0_[0]) + sl;
        const TileGeom gg = a.gn_g;
        const int cpg = cin >> 3, gi = lane >> 3, sub = lane & 7;
        float gam[4], bet[4];
        const bool bok = b < a.B;
        double n = 0.0, s1 = 0.0, s2 = 0.0;
        const int kmax = (cpg + 7) >> 3;  // channels per lane (wave-uniform)
#pragma unroll 1
        for (int k = 0; k < kmax; ++k) {
          const int c = gi * cpg + sub + 8 * k;
          const bool have = bok && sub + 8 * k < cpg;
          const bool first = !have || c < a.C0;
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
            *reinterpret_cast<float2*>(sTab + ((size_t)row * cin + gi * cpg + sub + 8 * k) * 2) = o;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B0: the table is complete

    // ---- per-item decode, once.  Item j of this thread = 16 bytes (4 channels, quarter q4 = lane & 3) of halo record
    // 16 (pw + 4 j) + (lane >> 2): exactly the bytes lane `lane` of this wave's LDS-DMA piece pw + 4 j lands, so a
    // thread only ever transforms what it fetched itself (its own vmcnt wait is all the ordering it needs).  Every
    // wave handles the same NIT = 3 RS items (three slices of RS, one slice per unit of a chunk); a piece past the
    // halo is fetched into / transformed inside a 1 KB trash piece, so that the instruction stream has no per-item
    // branches and every wave issues the same number of LDS-DMA operations per unit (static vmcnt immediates).
    const int nrec = NA * a.halo_px;
    const int npieces = (nrec + 15) >> 4;
    int poff[NIT];    // source pixel offset (0: outside the image / batch -> the item becomes zeros)
    int trow[NIT];    // byte offset of the item's scale/shift pairs inside its table row: (row * cin + 4 q4) * 8
    unsigned okmask = 0u, inmask = 0u;
    {
      const int per = HR * WR;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int rec = 16 * (pw + 4 * j) + (lane >> 2);
        poff[j] = 0, trow[j] = 0;
        if (rec < nrec) {
          const int ga = (NA == 2 && rec >= a.halo_px) ? 1 : 0;
          const int hp = rec - ga * a.halo_px;
          const int tb0 = ga ? tb0_[1] : tb0_[0], trow0 = ga ? trow0_[1] : trow0_[0];
          const int s = (g.spt == 1) ? 0 : (int)(__umul24((unsigned)hp, mPER) >> 16);
          const int rem = hp - s * per;
          const int hy = (int)(__umul24((unsigned)rem, mWR) >> 16), hx = rem - hy * WR;
          const int b = tb0 + s;
          int y, x;
          bool ok;
          if (MODE == CONV_S1) {
            y = trow0 + hy - 1, x = hx - 1;
            ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
          } else {
            const int yu = trow0 + hy - 1, xu = hx - 1;
            ok = (yu >= 0) && (yu < H) && (xu >= 0) && (xu < W);
            y = yu >> 1, x = xu >> 1;
          }
          ok = ok && (b < a.B);
          inmask |= 1u << j;
          if (ok) {
            poff[j] = (int)__umul24(__umul24((unsigned)b, (unsigned)a.Hin) + (unsigned)y, (unsigned)a.Win) + x;
            okmask |= 1u << j;
            trow[j] = ((ga * g.spt + s) * cin + 4 * q4) * 8;
          }
        }
      }
    }
    // LDS offsets of this lane inside a piece: its raw quarter, and where its four fp16 (plane h; plane l: ^ 32) go
    // (16 (pw + 4 j) is a multiple of 16 records, so the swizzle term is the same for every item)
    const int lrec = lane >> 2;
    const int raw_off = lrec * HRW + q4 * 16;
    const int dst_off = lrec * HRW + hswz(lrec, 0, q4 >> 1) + (q4 & 1) * 8;
    const int trash = 3 * abytes + NSLOT * UB + (a.gn_stats0 ? NA * g.spt * cin * 8 : 0);  // 1 KB piece no one reads
    // LDS byte offset of piece pw + 4 j of halo buffer bsel (wave-uniform)
    auto piece_off = [&](int bsel, int j) { return (pw + 4 * j < npieces) ? bsel * abytes + (pw + 4 * j) * 1024 : trash; };

    float amax = 0.f;   // max |a'| this thread has staged (range flag)

    // chunk source: tensor, its channel count, first channel inside it (chunks past the end: the last one, harmlessly)
    auto chunk_src = [&](int ch, const float*& src, int& cs, int& cc) {
      if (ch > ntot - 1) ch = ntot - 1;
      const bool skip = ch >= nmain;
      const int c = (skip ? ch - nmain : ch) * KC;
      if (!skip) {
        if (c < a.C0) src = a.in0, cs = a.C0, cc = c;
        else src = a.in1, cs = a.C1, cc = c - a.C0;
      } else {
        if (c < a.R0) src = a.res0, cs = a.R0, cc = c;
        else src = a.res1, cs = a.R1, cc = c - a.R0;
      }
    };
    // raw fp32 fetch of slice SL (RS items) of chunk ch, straight into halo buffer bsel (quarter q4 of a record =
    // channels 4 q4 .. 4 q4 + 3, unswizzled); out-of-image / out-of-batch lanes fetch element 0 and are zeroed by
    // stage(); a chunk past the last one is fetched into the trash piece (keeps the operation count static)
    auto dma_raw = [&](int ch, int bsel, auto sl_tag) {
      constexpr int SL = decltype(sl_tag)::value;
      const float* src;
      int cs, cc;
      chunk_src(ch, src, cs, cc);
      const float* lane_src = src + cc + q4 * 4;
      const bool real = ch < ntot;
#pragma unroll
      for (int j = SL * RS; j < SL * RS + RS; ++j)
        __builtin_amdgcn_global_load_lds((hx_gbl_void*)(lane_src + __umul24((unsigned)poff[j], (unsigned)cs)),
                                         (hx_lds_void*)(smemp + (real ? piece_off(bsel, j) : trash)), 16, 0, 0);
    };
    // In-place transform of slice SL of chunk ch in halo buffer bsel: raw fp32 quarter -> GroupNorm + SiLU -> two fp16
    // planes.  Branch-free, the RS items of the slice in one basic block (a lone wave hides neither LDS nor VALU
    // latency unless it has several independent chains in flight).  The four lanes of a record read their quarters
    // in ONE ds_read instruction before any of them stores (a wave's LDS instructions execute in order), so
    // overwriting a neighbour's source quarter is safe.  Lanes past the halo store into the trash piece.
    auto stage = [&](int ch, int bsel, auto sl_tag) {
      constexpr int SL = decltype(sl_tag)::value;
      const bool xform = ch < nmain && a.gn_stats0 != nullptr;
      const char* tab = reinterpret_cast<const char*>(sTab) + (xform ? ch * KC * 8 : 0);
      f32x4 v[RS], e0[RS], e1[RS];
      char* pb[RS];
#if defined(RGFM_ABL) && RGFM_ABL == 1
      return;  // ablation: no transform at all
#endif
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int j = SL * RS + i;
        pb[i] = smemp + piece_off(bsel, j);
#if defined(RGFM_ABL) && RGFM_ABL == 2
        v[i] = f32x4{(float)j, 1.f, 2.f, (float)trow[j]};  // ablation: no LDS reads
        e0[i] = v[i], e1[i] = v[i];
#else
        v[i] = *reinterpret_cast<const f32x4*>(pb[i] + raw_off);
        if (xform) {
          e0[i] = *reinterpret_cast<const f32x4*>(tab + trow[j]);
          e1[i] = *reinterpret_cast<const f32x4*>(tab + trow[j] + 16);
        }
#endif
      }
#pragma unroll
      for (int i = 0; i < RS; ++i) {
        const int j = SL * RS + i;
        f32x4 w = v[i];
        if (xform) {
          w.x = silu_scaled(e0[i].x * w.x + e0[i].y);
          w.y = silu_scaled(e0[i].z * w.y + e0[i].w);
          w.z = silu_scaled(e1[i].x * w.z + e1[i].y);
          w.w = silu_scaled(e1[i].z * w.w + e1[i].w);
        } else {
          w = w * HX_SA;
        }
        const bool okj = (okmask >> j) & 1u;
        w.x = okj ? w.x : 0.f, w.y = okj ? w.y : 0.f, w.z = okj ? w.z : 0.f, w.w = okj ? w.w : 0.f;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
        unsigned h0, l0, h1, l1;
        hsplit2(w.x, w.y, h0, l0);
        hsplit2(w.z, w.w, h1, l1);
        const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
        const bool inj = (inmask >> j) & 1u;
        char* d = inj ? pb[i] + dst_off : smemp + trash + raw_off;  // (a lane's own 16 trash bytes)
        *reinterpret_cast<hx_u32x2*>(d) = ph;
        *reinterpret_cast<hx_u32x2*>(inj ? pb[i] + (dst_off ^ 32) : smemp + trash + raw_off + 8) = pl;
      }
    };
    // LDS-DMA of the weights of unit gg into ring slot `slot`: GB pieces per wave, always (a piece past the unit's
    // bytes -- the two idle taps of a 1x1-skip unit, or a unit past the last one -- goes to the trash piece)
    const char* wpk_lane = reinterpret_cast<const char*>(a.wpkh) + (size_t)blockIdx.y * nmain * 9 * TAPB + lane * 16;
    const char* wsk_lane = reinterpret_cast<const char*>(a.wskiph) + (size_t)blockIdx.y * nskip * TAPB + lane * 16;
    auto dma_b = [&](int gg, int slot) {
      const bool real = gg < G, main = gg < 3 * nmain;
      const char* src = (main || !real) ? wpk_lane + (size_t)(real ? gg : 0) * UB : wsk_lane + (size_t)(gg - 3 * nmain) * TAPB;
      const int valid = !real ? 0 : (main ? NPIECE : NPIECE / 3);
      const int dst = 3 * abytes + slot * UB;
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        const int p = pw + 4 * i;
        const bool ex = p < valid;
        __builtin_amdgcn_global_load_lds((hx_gbl_void*)(src + (ex ? p * 1024 : 0)), (hx_lds_void*)(smemp + (ex ? dst + p * 1024 : trash)), 16, 0, 0);
      }
    };
    using SL0 = std::integral_constant<int, 0>;
    using SL1 = std::integral_constant<int, 1>;
    using SL2 = std::integral_constant<int, 2>;

    // ---- pipeline fill: raw halo of chunk 0, weights of units 0 .. NSLOT-2, halo of chunk 0 transformed, raw of chunk 1
    dma_raw(0, 0, SL0{}), dma_raw(0, 0, SL1{}), dma_raw(0, 0, SL2{});
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) dma_b(i, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage(0, 0, SL0{}), stage(0, 0, SL1{}), stage(0, 0, SL2{});
    dma_raw(1, 1, SL0{}), dma_raw(1, 1, SL1{}), dma_raw(1, 1, SL2{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B1
    PPROF_T(tk1);
    PPROF_ADD(4, tk0, tk1);

    // One unit of a 3x3 chunk c, slice SL:  (a) slice SL of chunk c + 1, in place in its buffer;  (b) weights of unit
    // g + NSLOT - 1 into the slot the consumers left at the end of unit g - 1;  (c) raw fetch of slice SL of chunk c + 2
    // into the buffer they left at the end of chunk c - 1;  (d) the weights of unit g + 1 have landed and the own LDS
    // stores are done -> barrier.  Every wave issues GB + RS LDS-DMA operations per unit, (b) before (c), so in steady
    // state the waits are immediates: (a)'s raw data was fetched three units ago = all but the last 2 (GB + RS)
    // operations; (d): NSLOT 3: the weights were (b) of the previous unit = all but the last GB + 2 RS; NSLOT 2: they
    // are this unit's (b) = all but the last RS.
    int gidx = 0, b1 = 1, b2 = 2, sl_next = NSLOT - 1;  // buffers of chunks c + 1 / c + 2; ring slot of unit g + NSLOT - 1
    auto unit = [&](int c, auto sl_tag, bool steady) {
      PPROF_T(tu0);
      if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (GB + RS)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PPROF_T(tw1);
      if (c + 1 < ntot) stage(c + 1, b1, sl_tag);
      PPROF_T(tu1);
      dma_b(gidx + NSLOT - 1, sl_next);
      dma_raw(c + 2, b2, sl_tag);
      PPROF_T(tu2);
      if (NSLOT == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RS) : "memory");
      else if (steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GB + 2 * RS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PPROF_T(tu3);
      if (gidx != G - 1) asm volatile("s_barrier" ::: "memory");
      PPROF_T(tu4);
      PPROF_ADD(9, tu0, tw1);
      PPROF_ADD(5, tw1, tu1);
      PPROF_ADD(6, tu1, tu2);
      PPROF_ADD(7, tu2, tu3);
      PPROF_ADD(8, tu3, tu4);
      sl_next = sl_next + 1 == NSLOT ? 0 : sl_next + 1;
      ++gidx;
    };
#pragma unroll 1
    for (int c = 0; c < nmain; ++c) {
      const bool steady = c > 0;
      unit(c, SL0{}, steady);
      unit(c, SL1{}, steady);
      unit(c, SL2{}, steady);
      b1 = b1 == 2 ? 0 : b1 + 1;
      b2 = b2 == 2 ? 0 : b2 + 1;
    }
    // 1x1-skip chunks: one-tap units; the whole next chunk is staged / fetched per unit, with conservative waits
#pragma unroll 1
    for (int c = nmain; c < ntot; ++c) {
      PPROF_T(tu0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (c + 1 < ntot) stage(c + 1, b1, SL0{}), stage(c + 1, b1, SL1{}), stage(c + 1, b1, SL2{});
      dma_b(gidx + NSLOT - 1, sl_next);
      if (c + 2 < ntot) dma_raw(c + 2, b2, SL0{}), dma_raw(c + 2, b2, SL1{}), dma_raw(c + 2, b2, SL2{});
      if (NSLOT == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (NSLOT 3: the next unit's weights came with the previous unit)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (gidx != G - 1) asm volatile("s_barrier" ::: "memory");
      PPROF_T(tu4);
      PPROF_ADD(5, tu0, tu4);
      sl_next = sl_next + 1 == NSLOT ? 0 : sl_next + 1;
      ++gidx;
      b1 = b1 == 2 ? 0 : b1 + 1;
      b2 = b2 == 2 ? 0 : b2 + 1;
    }
#ifdef RGFM_HX2P_PROF
    if (tid == 512) for (int i = 4; i < 9; ++i) atomicAdd(&g_hx2p_prof[i], (unsigned long long)pacc[i]);
    if (tid == 512) atomicAdd(&g_hx2p_prof[3], (unsigned long long)pacc[9]);  // (reuses the epilogue slot: raw-DMA wait)
#endif
    if (!(amax < HX_LIMIT)) atomicOr(a.range_flag, 1u);  // (rare) an activation left the fp16 range: the host re-runs on bx3
    return;
  }

  // ================================================================================= CONSUMERS
  const int grp = wave >> 2, seg = wave & 3;
  const int l31p = lane & 31, hp_ = lane >> 5;
  const int ga_w = PAIRN ? 0 : grp;
  const int my_tile = PAIRN ? (int)blockIdx.x : (int)blockIdx.x * 2 + grp;
  const int my_cb = PAIRN ? (int)blockIdx.y * 2 + grp : (int)blockIdx.y;
  const int b0 = ga_w ? tb0_[1] : tb0_[0], row0 = ga_w ? trow0_[1] : trow0_[0];
  const int n0 = my_cb * NBLK;
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
    bbase[nt] = rec * HRW;  // NBT is a multiple of 16: the swizzle term does not depend on the tap
    bsw[nt] = (rec >> 2) & 3;
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
      if (a.temb && sample_ok) v += a.temb[(size_t)(a.temb_per_row ? bw : 0) * a.temb_stride + c];
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
  PPROF_T(tc1);
  asm volatile("s_barrier" ::: "memory");  // B0
  asm volatile("s_barrier" ::: "memory");  // B1: halo of chunk 0 and the weights of units 0-1 are in LDS
  PPROF_T(tc2);
  PPROF_ADD(0, tk0, tc1);
  PPROF_ADD(2, tc1, tc2);

  // one tap: fragments of this wave's 2 pixel tiles x NT channel tiles, 3 MFMAs per tile pair
  auto tap = [&](const char* sAc, const char* sBu, int toff, int boff) {
    f16x8 af[2][2], bf[NT][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int rec = arec[mt] + toff;
      const char* pa = sAc + rec * HRW;
      const int o0 = ((hp_ ^ (rec >> 2)) & 3) * 16;  // slot of (plane h, half hp_); plane l: ^ 32
      af[mt][0] = *reinterpret_cast<const f16x8*>(pa + o0);
      af[mt][1] = *reinterpret_cast<const f16x8*>(pa + (o0 ^ 32));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int o0 = ((hp_ ^ bsw[nt]) & 3) * 16;
      bf[nt][0] = *reinterpret_cast<const f16x8*>(sBu + bbase[nt] + boff + o0);
      bf[nt][1] = *reinterpret_cast<const f16x8*>(sBu + bbase[nt] + boff + (o0 ^ 32));
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

  int gidx = 0, bcur = 0, scur = 0;  // halo buffer c % 3, ring slot g % NSLOT
#pragma unroll 1
  for (int c = 0; c < nmain; ++c) {
    const char* sAc = smemp + bcur * abytes;
    bcur = bcur == 2 ? 0 : bcur + 1;
#pragma unroll 1
    for (int u = 0; u < 3; ++u, ++gidx) {
      const char* sBu = sB + scur * UB;
      scur = scur + 1 == NSLOT ? 0 : scur + 1;
      PPROF_T(tm0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) tap(sAc, sBu, u * WR + kx, kx * TAPB);
#ifdef RGFM_HX2P_PROF
      asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][NT - 1][15]));
#endif
      PPROF_T(tm1);
      if (gidx != G - 1) asm volatile("s_barrier" ::: "memory");
      PPROF_T(tm2);
      PPROF_ADD(1, tm0, tm1);
      PPROF_ADD(2, tm1, tm2);
    }
  }
  if (nskip) {  // the 1x1 skip weights carry their own scale: q_main -> q_skip
    const float rs = a.hq_skip[0] * a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * rs;
#pragma unroll 1
    for (int c = nmain; c < ntot; ++c, ++gidx) {
      tap(smemp + bcur * abytes, sB + scur * UB, WR + 1, 0);
      bcur = bcur == 2 ? 0 : bcur + 1;
      scur = scur + 1 == NSLOT ? 0 : scur + 1;
      if (gidx != G - 1) asm volatile("s_barrier" ::: "memory");
    }
  }
  {
    const float qinv = nskip ? a.hq_skip[1] : a.hq[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * qinv;
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_hx2_kernel)
  PPROF_T(te0);
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int l31 = lane_e & 31, h = lane_e >> 5;
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
        const unsigned pix = (unsigned)pix0 + (unsigned)((g.spt == 1) ? p : pl);
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
      const int nparts = g.nparts;
      const int part = (g.spt == 1) ? (my_tile - b0 * g.tps) * 4 + seg : 0;
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
      if (a.fin_ab && sample_ok) fin_arrive(a, bw, lane_e, nparts, false);
    }
  };
  const bool full_seg = sample_ok && ((g.spt == 1) ? (nvalid - 64 * seg >= 64) : (HW == 64));  // wave-uniform
  if (full_seg) epilogue(std::true_type{});
  else epilogue(std::false_type{});
#ifdef RGFM_HX2P_PROF
  PPROF_T(te1);
  PPROF_ADD(3, te0, te1);
  if (tid == 0) {
    for (int i = 0; i < 3; ++i) atomicAdd(&g_hx2p_prof[i], (unsigned long long)pacc[i]);
    atomicAdd(&g_hx2p_prof[9], 1ull);
  }
#endif
}

// ---------------------------------------------------------------- host side
static int hx2p_halo(const ConvArgs& a) { return a.g.spt * (a.g.th + 2) * (a.g.W + 2); }
static bool hx2p_pairn(const ConvArgs& a) { return a.Cout % 128 == 0; }
static int hx2p_pieces(const ConvArgs& a) { return ((hx2p_pairn(a) ? 1 : 2) * hx2p_halo(a) + 15) / 16; }
// items per producer thread and unit: a chunk's pieces over 4 waves x 3 units
static int hx2p_rs(const ConvArgs& a) { return ((hx2p_pieces(a) + 3) / 4 + 2) / 3; }
static size_t hx2p_lds_bytes(const ConvArgs& a) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const bool pn = hx2p_pairn(a);
  const int na = pn ? 1 : 2, nbt = 32 * nt * (pn ? 2 : 1);
  const size_t abytes = (size_t)hx2p_pieces(a) * 1024;  // halo buffer in whole 1 KB DMA pieces
  size_t bytes = 3 * abytes + (size_t)(pn ? 3 : 2) * 3 * nbt * HRW;  // three halo buffers + the weight ring
  if (a.gn_stats0) bytes += (size_t)na * a.g.spt * (a.C0 + a.C1) * 2 * sizeof(float);  // scale/shift table
  return bytes + 1024;  // + the trash piece
}

// the pipelined kernel takes: stride-1 / upsampling convs whose input norm (if any) is the consumer-side one
bool conv_hx2p_supported(const ConvArgs& a, int mode) {
  if (mode != CONV_S1 && mode != CONV_UP2) return false;
  if (a.ab && !a.gn_stats0) return false;
  if (!conv_hx2_supported(a, mode)) return false;
  if (a.gn_stats0 && !conv_hx2_gn_supported(a, mode)) return false;
  const int rs = hx2p_rs(a);
  if (hx2p_pairn(a) ? (rs < 2 || rs > 3) : (rs < 3 || rs > 4)) return false;  // the instantiated slice sizes
  return hx2p_lds_bytes(a) <= 160 * 1024;
}

#define HX2P_FOR_ALL(X)                                                                              \
  X(1, CONV_S1, false, 3) X(1, CONV_S1, false, 4) X(1, CONV_UP2, false, 3) X(1, CONV_UP2, false, 4)  \
  X(2, CONV_S1, false, 3) X(2, CONV_S1, false, 4) X(2, CONV_UP2, false, 3) X(2, CONV_UP2, false, 4)  \
  X(2, CONV_S1, true, 2) X(2, CONV_S1, true, 3) X(2, CONV_UP2, true, 2) X(2, CONV_UP2, true, 3)

int conv_hx2p_init() {
  int rc = 0;
#define RAISEP(NTV, M, P, R) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2p_kernel<NTV, M, P, R>), \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  HX2P_FOR_ALL(RAISEP)
#undef RAISEP
  return rc;
}

void launch_conv_hx2p(const ConvArgs& a_in, int mode, hipStream_t s) {
  ConvArgs a = a_in;
  a.halo_px = hx2p_halo(a_in);
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int tiles = geom_num_tiles(a.g, a.B);
  const bool pn = hx2p_pairn(a);
  const int rs = hx2p_rs(a);
  dim3 grid(pn ? tiles : (tiles + 1) / 2, pn ? a.Cout / 128 : a.Cout / (32 * nt), 1);
  const size_t lds = hx2p_lds_bytes(a);
#define LAUNCHP(NTV, M, P, R) \
  if (nt == NTV && mode == M && pn == P && rs == R) hipLaunchKernelGGL((conv_mfma_hx2p_kernel<NTV, M, P, R>), grid, dim3(768), lds, s, a, tiles);
  HX2P_FOR_ALL(LAUNCHP)
#undef LAUNCHP
}

}  // namespace rgfm
