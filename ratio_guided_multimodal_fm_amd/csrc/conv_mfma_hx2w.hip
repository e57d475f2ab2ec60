// conv_mfma_hx2w.hip -- Winograd F(2x2, 3x3) form of the stride-1 3x3 conv on the two-plane fp16 arithmetic
// (conv_mfma_hx2.hip): 16 products per 2x2 output tile and input channel instead of 36, i.e. 4 / 9 of the f16-MFMAs of
// the direct form.  Same reference ops as conv_mfma_hx2p_kernel for the layers it takes (Conv2d 3x3 stride 1 behind
// GroupNorm + SiLU, concat on the load path, bias / time embedding / identity residual, GroupNorm partial statistics of
// the output; src/models/unet_flexible.py:71-85); only the summation order differs from the direct form.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        d: 4x4 patch of S_A silu(norm(x)),  g: 3x3 filter
//
// * The input transform B^T d B is taken in fp32 on the normalised, activated values and split into the two fp16 planes
//   AFTER it; the filter transform G g G^T in fp32 at create, scaled (its own scale record) and split (launch_pack_conv_hx2w).
//   Each of the 16 Winograd positions is then an ordinary two-plane GEMM over the input channels: a_l w_h + a_h w_l + a_h w_h.
// * Workgroup = 512 threads = 64 Winograd tiles (256 output pixels: the pixel tiles and statistics parts of the other conv
//   kernels) x 64 output channels.  Wave w owns positions 2w, 2w + 1 for ALL 64 tiles x 64 channels: 2 x (2 x 2) 32x32
//   accumulators = 128 registers, 8 fragment reads per 12 MFMAs as in the direct kernels.
// * Per 16-channel chunk: (1) raw halo (fetched one chunk ahead) -> GroupNorm + SiLU -> LDS image R (fp32, de-interleaved by
//   pixel parity so that the stride-2 patch reads are unit-stride), the chunk's weight fragments requested straight from
//   L2 into registers (a wave needs only its own two positions: 8 x 16 B per lane) | barrier | (2) thread = (tile, 4
//   channels, upper / lower half of the 4x4): 12 ds_read_b128 of R -> B^T d B -> split -> 16 ds_write_b64 into
//   V[position][tile] (the 64-byte two-plane records of the direct kernels) | barrier | (3) 24 MFMAs per wave.
// * Epilogue: the 16 positions of a tile live in 8 waves -- four passes (32 tiles x 32 channels each) through LDS:
//   accumulators -> E[position][tile][channel], then thread = (tile, channel): A^T m A, x 1/q, + bias / time term /
//   residual, 2x2 pixels stored, (mean, M2) of the four values -> S[tile][channel]; the parts' statistics (16 tiles = 64
//   pixels, the direct kernels' parts) are combined from S in tile order (Chan, fp64).
#include <stdlib.h>

#include <type_traits>

#include "conv_hx2_common.h"

namespace rgfm {

// ---------------------------------------------------------------- weights
// u[(co * Cin + ci) * 16 + pos] = (G g G^T)[pos],  G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]
__global__ void hx2w_filter_transform_kernel(const float* w, float* u, int Cout, int Cin) {
  const size_t total = (size_t)Cout * Cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float* g = w + i * 9;
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      t[0][j] = g[j];
      t[1][j] = 0.5f * (g[j] + g[3 + j] + g[6 + j]);
      t[2][j] = 0.5f * (g[j] - g[3 + j] + g[6 + j]);
      t[3][j] = g[6 + j];
    }
    for (int r = 0; r < 4; ++r) {
      u[i * 16 + r * 4 + 0] = t[r][0];
      u[i * 16 + r * 4 + 1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      u[i * 16 + r * 4 + 2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      u[i * 16 + r * 4 + 3] = t[r][2];
    }
  }
}

__device__ __forceinline__ void hx2w_split1(float v, unsigned short& h, unsigned short& l) {
  unsigned ph, pl;
  hsplit2(v, 0.f, ph, pl);
  h = (unsigned short)(ph & 0xffffu), l = (unsigned short)(pl & 0xffffu);
}

// packed image: [Cout / 64][Cin / 16][position 16][nt 2][plane 2][lane 64][8] fp16 -- lane L of a wave reads its B fragment of
// (position, 32-channel column nt, plane) as ONE 16-byte load: output channel 32 nt + (L & 31), input channels 8 (L >> 5) .. + 7
__global__ void hx2w_pack_kernel(const float* u, unsigned short* out, const float* hq, int Cout, int Cin) {
  const float sw = hq[2];
  const int nch = Cin / 16;
  const size_t total = (size_t)Cout * Cin * 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 7), lane = (int)((i >> 3) & 63), nt = (int)((i >> 9) & 1), pos = (int)((i >> 10) & 15);
    const size_t r = i >> 14;
    const int ch = (int)(r % nch), cb = (int)(r / nch);
    const int co = cb * 64 + nt * 32 + (lane & 31), ci = ch * 16 + (lane >> 5) * 8 + kk;
    unsigned short h, l;
    hx2w_split1(sw * u[((size_t)co * Cin + ci) * 16 + pos], h, l);
    // element i of the plane-h image; plane l: 512 elements (one [lane][8] block) further
    const size_t o = ((((((size_t)cb * nch + ch) * 16 + pos) * 2 + nt) * 2) * 64 + lane) * 8 + kk;
    out[o] = h, out[o + 512] = l;
  }
}

// out: Cout * Cin * 16 * 2 fp16; hq: the scale record {q, 1/q, s_w, eligible}; tmp: Cout * Cin * 16 floats of scratch
void hx2_scale_launch(const float* w, size_t n, float* hq, hipStream_t s);  // (conv_mfma_hx2.hip)
void launch_pack_conv_hx2w(const float* w, void* out, float* hq, float* tmp, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(hx2w_filter_transform_kernel, dim3(256), dim3(256), 0, s, w, tmp, Cout, Cin);
  hx2_scale_launch(tmp, (size_t)Cout * Cin * 16, hq, s);
  hipLaunchKernelGGL(hx2w_pack_kernel, dim3(256), dim3(256), 0, s, tmp, (unsigned short*)out, hq, Cout, Cin);
}

// ---------------------------------------------------------------- the conv
constexpr int HX2W_RREC = 80;  // bytes of an R record: 16 fp32 + 16 bytes of padding (consecutive records: conflict-free b128)

template <int WL2>
__global__ __launch_bounds__(512, 2) void conv_mfma_hx2w_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int W = 1 << WL2, TH = 256 / W, TX = W / 2, WR = W + 2, HR = TH + 2, HALO = HR * WR;
  constexpr int PW = WR / 2, PH = HR / 2, PSZ = PW * PH;  // a parity plane of the halo
  constexpr int RBYTES = ((4 * PSZ * HX2W_RREC + 1023) / 1024) * 1024;
  constexpr int MAXIT = (HALO * 4 + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) char smw[];
  // K loop: [R][V][table]; epilogue: [E: 16 x 64 x 32 fp32 = 128 KB][S: 64 x 32 x 2 fp32][table]
  char* const sV = smw + RBYTES;  // [16][64] records of 64 B
  float* const sTab = reinterpret_cast<float*>(smw + 128 * 1024 + 64 * 32 * 8);  // [cin][2]: S_A x (scale, shift) of this tile's sample

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // (scalar: the branches on it are)
  const int l31 = lane & 31, hp = lane >> 5;
  const TileGeom g = a.g;
  const int H = g.H;
  const int tile = (int)blockIdx.x;
  const int b0 = tile / g.tps, row0 = (tile - b0 * g.tps) * TH;
  const int cb = (int)blockIdx.y;
  const int cin = a.C0 + a.C1, nch = cin / KC;
  (void)num_tiles;

  // ---- staging items (halo pixel, 4 channels)
  const int q4 = tid & 3;
  int poff[MAXIT], rdst[MAXIT];
  unsigned okmask = 0u;
#pragma unroll
  for (int j = 0; j < MAXIT; ++j) {
    const int it = tid + 512 * j;
    poff[j] = 0, rdst[j] = 4 * PSZ * HX2W_RREC + q4 * 16;  // (lanes past the halo: a trash record behind the planes -- no per-lane branch)
    if (it < HALO * 4) {
      const int hpx = it >> 2;
      const int hy = hpx / WR, hx = hpx - hy * WR;
      const int y = row0 + hy - 1, x = hx - 1;
      const int rec = ((hy & 1) * 2 + (hx & 1)) * PSZ + (hy >> 1) * PW + (hx >> 1);
      rdst[j] = rec * HX2W_RREC + q4 * 16;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        okmask |= 1u << j;
        poff[j] = (b0 * H + y) * W + x;
      }
    }
  }
  static_assert(RBYTES >= 4 * PSZ * HX2W_RREC + 64, "room for the trash record");
  f32x4 ra[MAXIT];
  float hmax = 0.f;  // range flag: the largest |S_A silu(.)| this thread has staged
  auto issue_a = [&](int c) {
    const int ch0 = c * KC;
    const bool first = ch0 < a.C0;
    const float* src = first ? a.in0 + ch0 : a.in1 + (ch0 - a.C0);
    const unsigned cs = (unsigned)(first ? a.C0 : a.C1);
#pragma unroll
    for (int j = 0; j < MAXIT; ++j)
      ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j], cs) + (unsigned)(q4 * 4)));
  };
  issue_a(0);  // (ahead of the table: its round trip runs under the table's)

  // ---- scale / shift table of this tile's sample: from the producers' partial statistics (consumer-side GroupNorm, as
  // conv_mfma_hx2p_kernel's prologue: as many waves as it takes to give every lane ONE channel, fp64 sums, a butterfly
  // over the group's lanes), or copied from an external array (ConvArgs::ab)
  if (a.gn_stats0) {
    const int gn_cpg = cin >> 3;
    const int gn_wsh = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : (gn_cpg <= 32 ? 2 : 3));  // log2 of the waves that take part
    if (wave < (1 << gn_wsh)) {
      const int gn_lpg = 8 << gn_wsh;  // lanes per group
      const int gn_gi = wave * (8 >> gn_wsh) + (lane >> (3 + gn_wsh)), gn_sub = lane & (gn_lpg - 1);
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
        float2 gv[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) gv[p] = *reinterpret_cast<const float2*>(st + (((size_t)b0 * npt + (p < npt ? p : 0)) * cs + cc) * 2);
        const float g_ = a.gn_gamma[have ? c : 0], b_ = a.gn_beta[have ? c : 0];
        if (k == 0) gam[0] = g_, bet[0] = b_;
        else if (k == 1) gam[1] = g_, bet[1] = b_;
        else if (k == 2) gam[2] = g_, bet[2] = b_;
        else gam[3] = g_, bet[3] = b_;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const double np = (have && p < npt) ? (double)geom_part_count(a.gn_g, p % a.gn_g.nparts) : 0.0;
          const double mp = (double)gv[p].x;
          n += np;
          s1 += np * mp;
          s2 += np > 0.0 ? (double)gv[p].y + np * mp * mp : 0.0;
        }
      }
      for (int o = 1; o < gn_lpg; o <<= 1) n += __shfl_xor(n, o), s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
      const double mean = n > 0.0 ? s1 / n : 0.0;
      const double var = n > 0.0 ? s2 / n - mean * mean : 0.0;
      const float gm = (float)mean;
      const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (gn_sub + gn_lpg * k < gn_cpg) {
          const float sc = rstd * gam[k];
          float2 o;
          o.x = HX_SA * sc, o.y = HX_SA * (bet[k] - gm * sc);
          *reinterpret_cast<float2*>(sTab + (gn_gi * gn_cpg + gn_sub + gn_lpg * k) * 2) = o;
        }
    }
  } else {
    for (int c = tid; c < cin; c += 512) {
      const float2 e = *reinterpret_cast<const float2*>(a.ab + ((size_t)b0 * cin + c) * 2);
      float2 o;
      o.x = HX_SA * e.x, o.y = HX_SA * e.y;
      *reinterpret_cast<float2*>(sTab + 2 * c) = o;
    }
  }

  auto commit_a = [&](int c) {  // -> R
    const char* ep = reinterpret_cast<const char*>(sTab) + (c * KC + 4 * q4) * 8;
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 16);
    char* const rb = smw;
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      f32x4 v = ra[j];
      v.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
      v.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
      v.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
      v.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
      const float keep = ((okmask >> j) & 1u) ? 1.f : 0.f;  // zero padding
      v = v * keep;
      // range flag: a transformed value is a signed sum of four of these, so 4 max|.| bounds every plane-h operand
      hmax = hx_absmax3(v.x, v.y, hmax);
      hmax = hx_absmax3(v.z, v.w, hmax);
      *reinterpret_cast<f32x4*>(rb + rdst[j]) = v;
    }
  };

  // ---- transform item: (tile tt, channel quad tq) of this wave's HALF of the position grid: waves 0-3 take rows 0, 1
  // (positions 0 .. 7) -- which are also the positions waves 0-3 multiply --, waves 4-7 rows 2, 3: the two groups of four
  // waves share R and nothing else, so they run half a chunk apart (below)
  const int grp = wave >> 2;  // (wave-uniform)
  const int tt = tid & 63, tq = wave & 3;
  const int tty = tt / TX, ttx = tt - tty * TX;
  const int rbase = (tty * PW + ttx) * HX2W_RREC + tq * 16;  // record (plane 0, tty, ttx); element (i, j): + a constant
  const int vkey = (tt >> 2) & 3;
  const int vdst_h = grp * (8 * 64 * HRW) + tt * HRW + (((tq >> 1) ^ vkey) & 3) * 16 + (tq & 1) * 8;        // plane h, first position of the half
  const int vdst_l = grp * (8 * 64 * HRW) + tt * HRW + (((2 + (tq >> 1)) ^ vkey) & 3) * 16 + (tq & 1) * 8;  // plane l

  // ---- fragments
  int aofs[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) aofs[mt][pl] = (32 * mt + l31) * HRW + (((2 * pl + hp) ^ ((l31 >> 2) & 3)) & 3) * 16;
  const char* const wbase = reinterpret_cast<const char*>(a.wpkw) + (size_t)cb * nch * (16 * 4096) + (size_t)(2 * wave) * 4096 + lane * 16;
  f16x8 bfr[2][2][2];  // [position][nt][plane]
  auto issue_b = [&](int c) {
    const char* p = wbase + (size_t)c * (16 * 4096);
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          bfr[pi][nt][pl] = __builtin_bit_cast(f16x8, *(const hx_gf32x4*)(p + pi * 4096 + nt * 2048 + pl * 1024));
  };

  f32x16 acc[2][2][2];
#pragma unroll
  for (int pi = 0; pi < 2; ++pi)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pi][mt][nt][r] = 0.f;

  // B^T d B of (tile, 4 channels), this group's two rows of the position grid, from R[c & 1] into V
  auto transform = [&](int c) {
    const char* const rb = smw + rbase;
    // element (i, j) of the patch: parity plane (i & 1, j & 1), record (tty + (i >> 1), ttx + (j >> 1))
    auto rd = [&](int i, int j) -> f32x4 {
      const int off = (((i & 1) * 2 + (j & 1)) * PSZ + (i >> 1) * PW + (j >> 1)) * HX2W_RREC;
      return *reinterpret_cast<const f32x4*>(rb + off);
    };
    // (on float2 halves: v_pk_add_f32, two channels per instruction -- this phase has no MFMAs beside it)
    struct P2 {
      hx_f32x2 lo, hi;
    };
    auto ld = [&](int i, int j) -> P2 {
      const f32x4 v = rd(i, j);
      return P2{hx_f32x2{v.x, v.y}, hx_f32x2{v.z, v.w}};
    };
    auto sub = [](const P2& x, const P2& y) { return P2{x.lo - y.lo, x.hi - y.hi}; };
    auto add = [](const P2& x, const P2& y) { return P2{x.lo + y.lo, x.hi + y.hi}; };
    P2 t0[4], t1[4];
    if (grp == 0) {  // rows 0, 1 of B^T d: d0 - d2, d1 + d2
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const P2 d0 = ld(0, j), d1 = ld(1, j), d2 = ld(2, j);
        t0[j] = sub(d0, d2), t1[j] = add(d1, d2);
      }
    } else {  // rows 2, 3: d2 - d1, d1 - d3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const P2 d1 = ld(1, j), d2 = ld(2, j), d3 = ld(3, j);
        t0[j] = sub(d2, d1), t1[j] = sub(d1, d3);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const P2* t = r ? t1 : t0;
      const P2 v[4] = {sub(t[0], t[2]), add(t[1], t[2]), sub(t[2], t[1]), sub(t[1], t[3])};
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        unsigned h0, l0, h1, l1;
        hsplit2(v[cc].lo.x, v[cc].lo.y, h0, l0);
        hsplit2(v[cc].hi.x, v[cc].hi.y, h1, l1);
        const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
        *reinterpret_cast<hx_u32x2*>(sV + vdst_h + (r * 4 + cc) * (64 * HRW)) = ph;
        *reinterpret_cast<hx_u32x2*>(sV + vdst_l + (r * 4 + cc) * (64 * HRW)) = pl;
      }
    }
  };
  auto multiply = [&]() {  // this wave's two positions
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const char* vp = sV + (2 * wave + pi) * (64 * HRW);
      f16x8 af[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[mt][pl] = *reinterpret_cast<const f16x8*>(vp + aofs[mt][pl]);
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            acc[pi][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bfr[pi][nt][PB[q]], acc[pi][mt][nt], 0, 0, 0);
    }
  };

  // ---- the K loop: per chunk  (1) staged halo -> GroupNorm + SiLU -> R, weight fragments requested | barrier |  (2) next
  // chunk's halo requested, R -> V (the input transform) | barrier |  (3) multiply.
  // Two re-orderings were measured against this on 512-row launches (profiles/r04_kbench/hx2w_vs_direct.txt): the two groups
  // of four waves half a chunk apart (one group's transform under the other's MFMAs, R double-buffered): 11.3 instead of
  // 9.1 us per chunk -- a transform wave alone on its SIMD is latency-bound; the next chunk's staging under this chunk's
  // MFMAs: 128.9 instead of 123.9 us for 16x16 128 -> 128 -- exp / rcp beside MFMAs slow both.  A third: FOUR waves (one per
  // SIMD, the accumulators of four positions in the accumulation registers) with the transform of chunk c + 1 between the
  // MFMAs of chunk c in one instruction stream, V double-buffered (tools/experiments/conv_mfma_hx2w4.hip; bit-identical):
  // 150.6 against 123.6 us -- one wave per SIMD has nobody to hide its LDS round trips and its exp / rcp phase behind.
  // And a tile loop per workgroup (four tiles; the samples' tables made once, side by side on the waves; a tile's first halo
  // request issued before the previous tile's epilogue): 123.6 -> 120.6 us at 16x16, +-0 at 32x32, for 108 bytes of scratch
  // (profiles/r04_kbench/hx2w_tile_loop.txt) -- not kept: the fixed 10 us per tile are the epilogue's two passes, not the prologue.
  __syncthreads();  // the table
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    commit_a(c);
    issue_b(c);
    __syncthreads();
    issue_a(c + 1 < nch ? c + 1 : c);  // (past the end: a harmless re-fetch)
    transform(c);
    __syncthreads();
    multiply();
    // (the next chunk's phase (1) writes R only -- read in phase (2), which every wave has left; its barrier then orders
    // these fragment reads of V before the next phase (2) overwrites it)
  }
  if (!(4.f * hmax < HX_BIG)) atomicOr(a.range_flag, 1u);

  // ---------------------------------------------------------------- epilogue: two passes of 32 output channels
  const float qinv = a.hqw[1];
  float* const sE = reinterpret_cast<float*>(smw);                      // [16][64][32]
  float* const sS = reinterpret_cast<float*>(smw + 128 * 1024);         // [64][32][2]
  const bool sample_ok = b0 < a.B;
  const int et = tid >> 3, ecq = tid & 7;                               // this thread's tile and channel quad
  const int ety = et / TX, etx = et - ety * TX;
  const size_t pix = ((size_t)b0 * H + row0 + 2 * ety) * W + 2 * etx;
  auto pass = [&](auto nt_tag) {
    constexpr int nt = decltype(nt_tag)::value;
    __syncthreads();  // E and S of the previous pass (or the K loop's buffers) are done with
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float* e = sE + ((size_t)(2 * wave + pi) * 64 + 32 * mt) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2) + 4 * hp) * 32] = acc[pi][mt][nt][r];
      }
    __syncthreads();
    const int c = cb * 64 + nt * 32 + 4 * ecq;
    f32x4 add = *reinterpret_cast<const f32x4*>(a.bias + c);
    if (a.temb && sample_ok)
      add += *reinterpret_cast<const f32x4*>(a.temb + ((size_t)(a.temb_per_row ? b0 : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c);
    f32x4 mm[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) mm[p] = *reinterpret_cast<const f32x4*>(sE + ((size_t)(p * 64 + et) * 32 + 4 * ecq));
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      s0[cc] = mm[cc] + mm[4 + cc] + mm[8 + cc];
      s1[cc] = mm[4 + cc] - mm[8 + cc] - mm[12 + cc];
    }
    f32x4 y[4] = {s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      y[i] = y[i] * qinv + add;
      const size_t o = (pix + (i >> 1) * W + (i & 1)) * a.Cout + c;
      if (a.res_mode == 1 && sample_ok) y[i] += *reinterpret_cast<const f32x4*>(a.res0 + o);
      if (sample_ok) *reinterpret_cast<f32x4*>(a.out + o) = y[i];
    }
    if (a.small_check && a.range_flag && sample_ok) {
      // ConvArgs::small_check, on this wave's block of the pass (8 tiles x 32 channels = 32 pixels x 32 channels: a finer
      // block than the direct kernels' 64 x 32, i.e. the stricter test)
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) m = hx_absmax3(y[i].x, y[i].y, hx_absmax3(y[i].z, y[i].w, m));
      hx_small_flag(a.range_flag, m);
    }
    {
      const f32x4 mean = ((y[0] + y[1]) + (y[2] + y[3])) * 0.25f;
      f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) m2 += (y[i] - mean) * (y[i] - mean);
      const f32x4 lo = {mean.x, m2.x, mean.y, m2.y}, hi = {mean.z, m2.z, mean.w, m2.w};
      float* sp = sS + ((size_t)et * 32 + 4 * ecq) * 2;
      *reinterpret_cast<f32x4*>(sp) = lo;
      *reinterpret_cast<f32x4*>(sp + 4) = hi;
    }
    __syncthreads();
    if (tid < 128 && a.stats_out && sample_ok) {  // (part, channel): 16 tiles = 64 pixels, combined in tile order
      const int part = tid >> 5, co = tid & 31;
      // (unrolled: the sixteen reads go out together and the weights 4 / n, 4 (n - 4) / n are constants -- as a loop this was
      // sixteen dependent LDS round trips and thirty-two fp64 divisions on two waves while six waited at the next barrier)
      float2 sv[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) sv[k] = *reinterpret_cast<const float2*>(sS + ((size_t)(16 * part + k) * 32 + co) * 2);
      double mean = 0.0, m2 = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const double d = (double)sv[k].x - mean;
        mean += d * (1.0 / (double)(k + 1));
        m2 += (double)sv[k].y + d * d * (4.0 * (double)k / (double)(k + 1));
      }
      const int gpart = (tile - b0 * g.tps) * 4 + part;
      float2 o;
      o.x = (float)mean, o.y = (float)m2;
      *reinterpret_cast<float2*>(a.stats_out + (((size_t)b0 * g.nparts + gpart) * a.Cout + cb * 64 + nt * 32 + co) * 2) = o;
    }
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------- host side
static size_t hx2w_lds_bytes(const ConvArgs& a) {
  // the epilogue's image (E 128 KB + S 16 KB) is the larger one; the K loop's 2 R + V fit inside E
  return (size_t)128 * 1024 + 64 * 32 * 8 + (size_t)(a.C0 + a.C1) * 8;
}

bool conv_hx2w_supported(const ConvArgs& a, int mode) {
  if (mode != CONV_S1 || !a.wpkw || !a.hqw || !a.range_flag) return false;
  const TileGeom& g = a.g;
  if (g.spt != 1 || (g.W != 16 && g.W != 32) || g.th * g.W != 256 || g.H % g.th != 0) return false;
  if (a.Hin != g.H || a.Win != g.W) return false;
  if (a.Cout % 64 != 0 || (a.C0 + a.C1) % KC != 0 || a.C0 % KC != 0) return false;
  if (a.res_mode == 2 || a.fin_ab || a.pout || a.pin0 || a.ep_scale || !a.out) return false;
  if (a.res_mode == 1 && a.R0 != a.Cout) return false;
  if (a.gn_stats0) {
    const int cin = a.C0 + a.C1;
    if (cin < 32 || cin % 8 != 0 || cin > 256 || a.gn_nparts0 > 16 || a.gn_g.nparts > 16) return false;
  } else if (!a.ab) {
    return false;  // (convs of raw inputs stay on the direct kernels)
  }
  const size_t px = (size_t)a.B * a.g.HW;
  if (px >= (1u << 24) || px * (size_t)(a.C0 > a.Cout ? a.C0 : a.Cout) >= (1ull << 32)) return false;
  return hx2w_lds_bytes(a) <= 160 * 1024;
}

int conv_hx2w_init() {
  int rc = 0;

  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2w_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2w_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

void launch_conv_hx2w(const ConvArgs& a, hipStream_t s) {
  const int tiles = geom_num_tiles(a.g, a.B);
  const dim3 grid(tiles, a.Cout / 64);
  const size_t lds = hx2w_lds_bytes(a);
  if (a.g.W == 16) hipLaunchKernelGGL((conv_mfma_hx2w_kernel<4>), grid, dim3(512), lds, s, a, tiles);
  else hipLaunchKernelGGL((conv_mfma_hx2w_kernel<5>), grid, dim3(512), lds, s, a, tiles);
}

}  // namespace rgfm
