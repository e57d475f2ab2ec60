// conv_mfma_v3.hip -- persistent, double-buffered, software-pipelined version of the
// fp32-MFMA implicit-GEMM convolution (same arithmetic and tiling as conv_mfma.hip;
// see the header there for what the kernel fuses and the reference lines it replaces).
//
// Why a third structure: with two co-resident single-buffer workgroups per CU the
// workgroups run in lockstep, so every per-chunk LDS-write phase, every barrier and
// every per-block prologue/epilogue idles the matrix pipe on all four SIMDs at once
// (measured: 78 % MFMA-busy on the best layers, 65 % overall).  Here:
//   * ONE workgroup per CU (4 waves = one per SIMD, up to 512 VGPRs each), persistent:
//     it walks tiles  t = blockIdx.x, += gridDim.x  (tile = 256 pixels x 32*NT channels);
//   * LDS holds TWO copies of the staged input halo tile and of the 9-tap weight chunk;
//     the global loads of K-chunk k+1 are issued at tap 0 of chunk k and their
//     GroupNorm+SiLU transform and ds_writes are spread over taps 2..8, i.e. they issue
//     in the 64-cycle shadows of the wave's own MFMAs -- one barrier per chunk, nothing
//     else between two chunks;
//   * the A/B fragments of tap t+1 are read from LDS while the MFMAs of tap t run;
//   * the next tile's first chunk is prefetched during the current tile's last chunk,
//     the identity-residual tile is fetched at tile start and only added in the epilogue.
#include <stdlib.h>

#include "rgfm_device.h"

namespace rgfm {

// explicitly global (address space 1) float4: guarantees global_load, never flat_load
typedef __attribute__((address_space(1))) f32x4 gf32x4;

constexpr int V3_MAXIT = 7;  // ceil(max halo_px * 4 / 256), halo_px <= 448

template <int NT>
struct V3Stage {  // everything needed to stage chunks of one tile
  int poff[V3_MAXIT];
  unsigned okmask, smask;
  int b0, nblk;
};

template <int NT, int MODE, bool SPT4, bool XFORM>
__global__ __launch_bounds__(256, 1) void conv_mfma_v3_kernel(const ConvArgs a, int n_nblk, int total_tiles) {
  constexpr int NB = (9 * 32 * NT * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int strideA = a.halo_px * LDP;
  constexpr int strideB = 9 * 32 * NT * LDP;
  float* const sA0 = smem;
  float* const sB0 = smem + 2 * strideA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int q4 = tid & 3;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;
  const int HR = g.th + 2, WR = W + 2;
  const int nA = a.halo_px * 4;
  const int cin = a.C0 + a.C1;
  const int nch_main = cin / KC;
  const int nch_skip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nch_main + nch_skip;

  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 32 + l31) * LDP + h * 8;

  // ---------------- staging machinery
  // GroupNorm scale/shift registers: one pair per item only when a tile spans 4 samples
  constexpr int NE = SPT4 ? V3_MAXIT : 1;
  f32x4 ra[V3_MAXIT], re0[NE], re1[NE], rb[NB];

  auto make_stage = [&](int tile, V3Stage<NT>& st) {
    const int ptile = tile / n_nblk;
    st.nblk = tile - ptile * n_nblk;
    int row0;
    if (g.spt == 1) {
      st.b0 = ptile / g.tps;
      row0 = (ptile - st.b0 * g.tps) * g.th;
    } else {
      st.b0 = ptile * g.spt;
      row0 = 0;
    }
    st.okmask = 0u;
    st.smask = 0u;
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < V3_MAXIT; ++j) {
      const int it = tid + 256 * j;
      st.poff[j] = 0;
      if (it < nA) {
        const int hp = it >> 2;
        const int s = hp / per;
        const int rem = hp - s * per;
        const int hy = rem / WR, hx = rem - hy * WR;
        const int b = st.b0 + s;
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
        ok = ok && (b < a.B);
        if (ok) {
          st.poff[j] = (b * a.Hin + y) * a.Win + x;
          st.okmask |= 1u << j;
          st.smask |= (unsigned)s << (2 * j);
        }
      }
    }
  };

  // Uniform description of the chunk being staged.
  struct ChunkDesc {
    const float* src;
    const float* wsrc;
    int cs, cc, c, nbit;
  };
  auto describe = [&](const V3Stage<NT>& st, int ch) {
    ChunkDesc d;
    const bool skip = ch >= nch_main;
    // scalar selects only (an if/else ladder over the four source pointers became a scratch-resident
    // pointer table, and loads through it were emitted as flat_load, which drains vmcnt AND lgkmcnt)
    const float* p0 = skip ? a.res0 : a.in0;
    const float* p1 = skip ? a.res1 : a.in1;
    const int n0c = skip ? a.R0 : a.C0, n1c = skip ? a.R1 : a.C1;
    d.c = (skip ? ch - nch_main : ch) * KC;
    const bool second = d.c >= n0c;
    d.src = second ? p1 : p0;
    d.cs = second ? n1c : n0c;
    d.cc = second ? d.c - n0c : d.c;
    const float* w0 = a.wpk + ((size_t)(st.nblk * nch_main + ch) * 9) * (32 * NT * KC);
    const float* w1 = a.wskip + ((size_t)(st.nblk * nch_skip + (ch - nch_main))) * (32 * NT * KC);
    d.wsrc = skip ? w1 : w0;
    d.nbit = skip ? 32 * NT * 4 : 9 * 32 * NT * 4;
    return d;
  };
  // global loads of ONE staging item into registers (clamped, unconditional: invalid items read
  // pixel 0 and are zeroed at commit).  Issued one item per tap so that at most ~3 items are live.
  auto issue_a = [&](const V3Stage<NT>& st, const ChunkDesc& d, int j) {
    ra[j] = *(const gf32x4*)(d.src + (size_t)st.poff[j] * d.cs + d.cc + q4 * 4);
    if (XFORM && (SPT4 || j == 0)) {
      int bb = st.b0 + (SPT4 ? (int)((st.smask >> (2 * j)) & 3u) : 0);
      bb = bb < a.B ? bb : 0;
      const gf32x4* p = (const gf32x4*)(a.ab + ((size_t)bb * cin + d.c + q4 * 4) * 2);
      re0[SPT4 ? j : 0] = p[0];
      re1[SPT4 ? j : 0] = p[1];
    }
  };
  auto issue_b = [&](const ChunkDesc& d, int j) {
    const int it = tid + 256 * j;
    rb[j] = *(const gf32x4*)(d.wsrc + (size_t)(it < d.nbit ? it : 0) * 4);
  };
  auto issue = [&](const V3Stage<NT>& st, int ch) {
    const ChunkDesc d = describe(st, ch);
#pragma unroll
    for (int j = 0; j < V3_MAXIT; ++j) issue_a(st, d, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) issue_b(d, j);
  };

  // Branch-free commits (they must live in the same basic block as the MFMAs so that the
  // scheduler can put them in the MFMA shadows): out-of-range items go to a dummy LDS slot,
  // invalid (padding) pixels and untransformed chunks are handled by selects.
  float* const sDummy = smem + 2 * strideA + 2 * strideB;  // 16 floats
  int tid_c = tid;  // re-laundered at every chunk: keeps per-item LDS addresses from being hoisted
  auto commit_a = [&](const V3Stage<NT>& st, int ch, int j, float* sA) {
    const int it = tid_c + 256 * j;
    f32x4 v = ra[j];
    if (XFORM) {
      const bool xf = ch < nch_main;  // wave-uniform select, no branch
      const f32x4 e0 = re0[SPT4 ? j : 0], e1 = re1[SPT4 ? j : 0];
      const float tx = silu_fast(e0.x * v.x + e0.y), ty = silu_fast(e0.z * v.y + e0.w);
      const float tz = silu_fast(e1.x * v.z + e1.y), tw = silu_fast(e1.z * v.w + e1.w);
      v.x = xf ? tx : v.x, v.y = xf ? ty : v.y, v.z = xf ? tz : v.z, v.w = xf ? tw : v.w;
    }
    const bool ok = (st.okmask >> j) & 1u;
    v.x = ok ? v.x : 0.f, v.y = ok ? v.y : 0.f, v.z = ok ? v.z : 0.f, v.w = ok ? v.w : 0.f;
    float* dst = (it < nA) ? sA + (it >> 2) * LDP + (tid_c & 3) * 4 : sDummy + (tid_c & 3) * 4;
    *reinterpret_cast<f32x4*>(dst) = v;
  };
  auto commit_b = [&](int ch, int j, float* sB) {
    const int nbit = (ch >= nch_main) ? 32 * NT * 4 : 9 * 32 * NT * 4;
    const int it = tid_c + 256 * j;
    float* dst = (it < nbit) ? sB + (it >> 2) * LDP + (tid_c & 3) * 4 : sDummy + (tid_c & 3) * 4;
    *reinterpret_cast<f32x4*>(dst) = rb[j];
  };

  // ---------------- prologue: stage chunk 0 of this block's first tile
  V3Stage<NT> st_cur, st_nxt;
  int tile = blockIdx.x;
  if (tile >= total_tiles) return;
  make_stage(tile, st_cur);
  issue(st_cur, 0);
#pragma unroll
  for (int j = 0; j < V3_MAXIT; ++j) commit_a(st_cur, 0, j, sA0);
#pragma unroll
  for (int j = 0; j < NB; ++j) commit_b(0, j, sB0);
  __syncthreads();
  int gbuf = 0;  // LDS buffer holding the chunk about to be computed

  for (; tile < total_tiles; tile += gridDim.x) {
    // ---------------- per-tile compute context
    const int ptile = tile / n_nblk;
    const int nblk = tile - ptile * n_nblk;
    int b0, row0;
    if (g.spt == 1) {
      b0 = ptile / g.tps;
      row0 = (ptile - b0 * g.tps) * g.th;
    } else {
      b0 = ptile * g.spt;
      row0 = 0;
    }
    const int n0 = nblk * (32 * NT);
    int rows_valid = H - row0;
    if (rows_valid > g.th) rows_valid = g.th;
    const int nvalid = rows_valid * W;
    int abase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int p = 64 * wave + 32 * mt + l31;
      int s, q;
      if (g.spt == 1) {
        s = 0;
        q = p < nvalid ? p : nvalid - 1;
      } else {
        s = wave;
        q = (p & 63) < HW ? (p & 63) : HW - 1;
      }
      const int r = q / W, x = q - r * W;
      abase[mt] = ((s * HR + r) * WR + x) * LDP + h * 8;
    }
    const int bw = (g.spt == 1) ? b0 : b0 + wave;
    const bool sample_ok = bw < a.B;
    const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W : (size_t)bw * HW;

    f32x16 acc[2][NT];
    {
      float add0[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        float v = a.bias[c];
        if (a.res_mode == 2) v += a.skip_bias[c];
        if (a.temb && sample_ok) v += a.temb[(size_t)(a.temb_per_row ? bw : 0) * a.temb_stride + c];
        add0[nt] = v;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = add0[nt];
    }
    // identity residual: fetched (branch-free, clamped) behind the last tap's staging work of the
    // tile's last chunk, when the staging registers are free again; consumed in the epilogue
    f32x16 res[2][NT];
    auto load_residual = [&]() {
      int lane_r = lane;
      asm volatile("" : "+v"(lane_r));
      const int l31 = lane_r & 31, h = lane_r >> 5;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int p = 64 * wave + pl;
          const bool valid = (g.spt == 1) ? (p < nvalid) : (sample_ok && pl < HW);
          const size_t pix = valid ? pix0 + ((g.spt == 1) ? p : pl) : (sample_ok ? pix0 : 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) res[mt][nt][r] = a.res0[pix * a.Cout + n0 + nt * 32 + l31];
        }
    };

    const bool have_next_tile = tile + (int)gridDim.x < total_tiles;

    for (int ch = 0; ch < ntot; ++ch) {
      const bool skip = ch >= nch_main;
      // opaque copies: stop LICM from hoisting 60+ loop-invariant LDS addresses out of the chunk loop
      asm volatile("" : "+v"(abase[0]), "+v"(abase[1]), "+v"(tid_c));
      const float* sA = sA0 + gbuf * strideA;
      const float* sB = sB0 + gbuf * strideB;
      float* sAn = sA0 + (gbuf ^ 1) * strideA;
      float* sBn = sB0 + (gbuf ^ 1) * strideB;
      // what gets staged while this chunk computes
      const bool last = ch + 1 == ntot;
      // staging target: next chunk of this tile, else chunk 0 of the next tile, else (nothing left)
      // chunk 0 of this tile again -- harmless in-bounds loads whose LDS image is never read, so the
      // staging code needs no branch
      const int nch = last ? 0 : ch + 1;
      if (last && have_next_tile) make_stage(tile + gridDim.x, st_nxt);
      // (selected field by field: a reference chosen at run time would force both structs to scratch)
      const bool use_nxt = last && have_next_tile;
      V3Stage<NT> stn;
#pragma unroll
      for (int j = 0; j < V3_MAXIT; ++j) stn.poff[j] = use_nxt ? st_nxt.poff[j] : st_cur.poff[j];
      stn.okmask = use_nxt ? st_nxt.okmask : st_cur.okmask;
      stn.smask = use_nxt ? st_nxt.smask : st_cur.smask;
      stn.b0 = use_nxt ? st_nxt.b0 : st_cur.b0;
      stn.nblk = use_nxt ? st_nxt.nblk : st_cur.nblk;
      const ChunkDesc dn = describe(stn, nch);

      if (!skip) {
        float af[2][2][8], bf[2][NT][8];
        auto load_frags = [&](int tap, int set) {
          const int ky = tap / 3, kx = tap - 3 * ky;
          const int aoff = (ky * WR + kx) * LDP;
          const int boff = tap * (32 * NT * LDP);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff + 4);
            af[set][mt][0] = v0.x, af[set][mt][1] = v0.y, af[set][mt][2] = v0.z, af[set][mt][3] = v0.w;
            af[set][mt][4] = v1.x, af[set][mt][5] = v1.y, af[set][mt][6] = v1.z, af[set][mt][7] = v1.w;
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff + 4);
            bf[set][nt][0] = v0.x, bf[set][nt][1] = v0.y, bf[set][nt][2] = v0.z, bf[set][nt][3] = v0.w;
            bf[set][nt][4] = v1.x, bf[set][nt][5] = v1.y, bf[set][nt][6] = v1.z, bf[set][nt][7] = v1.w;
          }
        };
        load_frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int set = tap & 1;
          // One scheduling region per k-step: hipcc otherwise leaves the staging VALU as one
          // clump behind the tap's 32 MFMAs (matrix pipe idle ~20 % of the tap).  Each k-step's
          // 2*NT MFMAs are followed by ONE stage of the commit pipeline of staging item (tap-2):
          // 4 independent VALU/TRANS ops or one ds_write, i.e. ~30 issue cycles in a 128..256-cycle
          // matrix shadow.
          __builtin_amdgcn_sched_barrier(0);
          if (tap < V3_MAXIT) issue_a(stn, dn, tap);
          if (tap < 7 && tap < NB) issue_b(dn, tap);
          if (tap == 5 && NB > 7) issue_b(dn, 7);
          if (tap == 6 && NB > 8) issue_b(dn, 8);
          if (tap + 1 < 9) load_frags(tap + 1, set ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          const int cj = tap >= 2 ? tap - 2 : 0;
          const bool cxf = XFORM && nch < nch_main;  // wave-uniform
          f32x4 cv = ra[cj], ct = cv, cu = cv;
#pragma unroll
          for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][mt][s], bf[set][nt][s], acc[mt][nt], 0, 0, 0);
            if (tap >= 2) {
              if (XFORM) {
                const f32x4 e0 = re0[SPT4 ? cj : 0], e1 = re1[SPT4 ? cj : 0];
                if (s == 0) ct.x = e0.x * cv.x + e0.y, ct.y = e0.z * cv.y + e0.w, ct.z = e1.x * cv.z + e1.y, ct.w = e1.z * cv.w + e1.w;
                if (s == 1) cu.x = ct.x * -1.44269504088896341f, cu.y = ct.y * -1.44269504088896341f, cu.z = ct.z * -1.44269504088896341f, cu.w = ct.w * -1.44269504088896341f;
                if (s == 2) cu.x = __builtin_amdgcn_exp2f(cu.x), cu.y = __builtin_amdgcn_exp2f(cu.y), cu.z = __builtin_amdgcn_exp2f(cu.z), cu.w = __builtin_amdgcn_exp2f(cu.w);
                if (s == 3) cu.x = 1.0f + cu.x, cu.y = 1.0f + cu.y, cu.z = 1.0f + cu.z, cu.w = 1.0f + cu.w;
                if (s == 4) cu.x = __builtin_amdgcn_rcpf(cu.x), cu.y = __builtin_amdgcn_rcpf(cu.y), cu.z = __builtin_amdgcn_rcpf(cu.z), cu.w = __builtin_amdgcn_rcpf(cu.w);
                if (s == 5) {
                  ct.x *= cu.x, ct.y *= cu.y, ct.z *= cu.z, ct.w *= cu.w;
                  cv.x = cxf ? ct.x : cv.x, cv.y = cxf ? ct.y : cv.y, cv.z = cxf ? ct.z : cv.z, cv.w = cxf ? ct.w : cv.w;
                }
              }
              if (s == 6) {
                const int it = tid_c + 256 * cj;
                const bool ok = (stn.okmask >> cj) & 1u;
                cv.x = ok ? cv.x : 0.f, cv.y = ok ? cv.y : 0.f, cv.z = ok ? cv.z : 0.f, cv.w = ok ? cv.w : 0.f;
                float* dst = (it < nA) ? sAn + (it >> 2) * LDP + (tid_c & 3) * 4 : sDummy + (tid_c & 3) * 4;
                *reinterpret_cast<f32x4*>(dst) = cv;
              }
              if (s == 7) {
                if (cj < NB) commit_b(nch, cj, sBn);
                if (tap == 8) {
#pragma unroll
                  for (int j = 7; j < NB; ++j) commit_b(nch, j, sBn);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (tap == 8 && last && a.res_mode == 1) load_residual();
        }
      } else {
        // fused 1x1 skip conv: centre tap only
        issue(stn, nch);
        float af[2][8], bf[NT][8];
        const int aoff = (WR + 1) * LDP;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff + 4);
          af[mt][0] = v0.x, af[mt][1] = v0.y, af[mt][2] = v0.z, af[mt][3] = v0.w;
          af[mt][4] = v1.x, af[mt][5] = v1.y, af[mt][6] = v1.z, af[mt][7] = v1.w;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(sB + bbase[nt]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + 4);
          bf[nt][0] = v0.x, bf[nt][1] = v0.y, bf[nt][2] = v0.z, bf[nt][3] = v0.w;
          bf[nt][4] = v1.x, bf[nt][5] = v1.y, bf[nt][6] = v1.z, bf[nt][7] = v1.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][s], bf[nt][s], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < V3_MAXIT; ++j) commit_a(stn, nch, j, sAn);
#pragma unroll
        for (int j = 0; j < NB; ++j) commit_b(nch, j, sBn);
      }
      __syncthreads();  // buffer gbuf fully consumed, buffer gbuf^1 fully written
      gbuf ^= 1;
    }

    // ---------------- epilogue
    // (lane id laundered so that the tile-start addresses are recomputed here instead of being
    // kept alive across the K loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l31 = lane_e & 31, h = lane_e >> 5;
    float eps_[NT], eph_[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = n0 + nt * 32 + l31;
      eps_[nt] = a.ep_scale ? a.ep_scale[c] : 1.f;
      eph_[nt] = a.ep_scale ? a.ep_shift[c] : 0.f;
    }
    unsigned vmask[2] = {0u, 0u};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int p = 64 * wave + pl;
        const bool valid = (g.spt == 1) ? (p < nvalid) : (sample_ok && pl < HW);
        if (valid) vmask[mt] |= 1u << r;
        const size_t pix = pix0 + ((g.spt == 1) ? p : pl);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int c = n0 + nt * 32 + l31;
          float v = acc[mt][nt][r];
          if (a.res_mode == 1) v += res[mt][nt][r];
          if (a.ep_scale) v = silu_f(v * eps_[nt] + eph_[nt]);
          acc[mt][nt][r] = v;
          if (valid) a.out[pix * a.Cout + c] = v;
        }
      }
    if (a.stats_out) {
      int nw;
      if (g.spt == 1) {
        nw = nvalid - 64 * wave;
        nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
      } else {
        nw = sample_ok ? HW : 0;
      }
      const int part = (g.spt == 1) ? (ptile - b0 * g.tps) * 4 + wave : 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float s = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (vmask[mt] & (1u << r)) s += acc[mt][nt][r];
        s += __shfl_xor(s, 32);
        const float mean = nw > 0 ? s / (float)nw : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (vmask[mt] & (1u << r)) {
              const float d = acc[mt][nt][r] - mean;
              m2 += d * d;
            }
        m2 += __shfl_xor(m2, 32);
        if (h == 0 && sample_ok) {
          const int c = n0 + nt * 32 + l31;
          float2 st;
          st.x = mean;
          st.y = m2;
          *reinterpret_cast<float2*>(a.stats_out + (((size_t)bw * g.nparts + part) * a.Cout + c) * 2) = st;
        }
      }
    }
    st_cur = st_nxt;
  }
}

size_t conv_v3_lds_bytes(const ConvArgs& a) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  return (size_t)2 * (a.halo_px + 9 * 32 * nt) * LDP * sizeof(float) + 16 * sizeof(float);
}

bool conv_v3_supported(const ConvArgs& a, int mode) {
  return mode != CONV_S2 && a.halo_px * 4 <= V3_MAXIT * 256 && conv_v3_lds_bytes(a) <= 160 * 1024;
}

template <int NT, int MODE>
static int raise_lds_v3() {
  int rc = 0;
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<NT, MODE, false, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<NT, MODE, false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<NT, MODE, true, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_v3_kernel<NT, MODE, true, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

int conv_v3_init() {
  return raise_lds_v3<1, CONV_S1>() | raise_lds_v3<1, CONV_UP2>() | raise_lds_v3<2, CONV_S1>() |
         raise_lds_v3<2, CONV_UP2>();
}

void launch_conv_v3(const ConvArgs& a, int mode, int num_cus, hipStream_t s) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int n_nblk = a.Cout / (32 * nt);
  const int total = geom_num_tiles(a.g, a.B) * n_nblk;
  const int grid = total < num_cus ? total : num_cus;
  const size_t lds = conv_v3_lds_bytes(a);
#define LAUNCH4(NTV, M, S4, XF) \
  hipLaunchKernelGGL((conv_mfma_v3_kernel<NTV, M, S4, XF>), dim3(grid), dim3(256), lds, s, a, n_nblk, total)
#define LAUNCH(NTV, M)                                      \
  do {                                                      \
    const bool s4 = a.g.spt != 1, xf = a.ab != nullptr;     \
    if (!s4 && !xf) LAUNCH4(NTV, M, false, false);          \
    else if (!s4 && xf) LAUNCH4(NTV, M, false, true);       \
    else if (s4 && !xf) LAUNCH4(NTV, M, true, false);       \
    else LAUNCH4(NTV, M, true, true);                       \
  } while (0)
  if (nt == 2) {
    if (mode == CONV_S1) LAUNCH(2, CONV_S1);
    else LAUNCH(2, CONV_UP2);
  } else {
    if (mode == CONV_S1) LAUNCH(1, CONV_S1);
    else LAUNCH(1, CONV_UP2);
  }
#undef LAUNCH
#undef LAUNCH4
}

}  // namespace rgfm
