// conv_mfma.hip -- 3x3 (+ fused 1x1 skip) convolution as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, bitwise an fmaf chain).
//
// Replaces, fused into ONE kernel per conv, the reference ops of a ResBlock half
// (src/models/unet_flexible.py:71-85):
//     GroupNorm-apply + SiLU  (on the load path, from per-(sample,channel) scale/shift)
//     torch.cat([h, skip],1)  (two source pointers)
//     F.interpolate(nearest, x2) (folded into the staging addresses, :107)
//     Conv2d 3x3 stride 1 / stride 2 (:93)  -> MFMA main loop
//     + bias + time-embedding add (:77-78)
//     + identity residual or 1x1 skip conv (:85) (extra K-chunks into the same accumulators)
//     + per-channel GroupNorm partial statistics of the OUTPUT for the next norm.
//
// Tiling (DESIGN.md "conv_mfma"): workgroup = 256 threads = 4 waves; block tile =
// 256 output pixels (4 wave segments of 64) x 32*NT output channels; each wave
// owns 2 x NT accumulator tiles of 32x32.  K loop: 16 input channels at a time;
// the (GroupNorm+SiLU-transformed, zero-padded) input halo tile and the 9 taps of
// weights for those channels are staged in LDS once and reused by all 9 taps.
// MFMA operand k-permutation: lane half h of k-step s holds channel 8h+s of the
// chunk, so every lane reads its 8 channels as two ds_read_b128.
#include <stdlib.h>

#include "rgfm_device.h"

namespace rgfm {

template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + a.halo_px * LDP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;

  int b0, row0;
  if (g.spt == 1) {
    b0 = blockIdx.x / g.tps;
    row0 = (blockIdx.x - b0 * g.tps) * g.th;
  } else {
    b0 = blockIdx.x * g.spt;
    row0 = 0;
  }
  const int n0 = blockIdx.y * (32 * NT);
  const int HR = (MODE == CONV_S2) ? 2 * g.th + 1 : g.th + 2;
  const int WR = (MODE == CONV_S2) ? 2 * W + 1 : W + 2;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;  // valid pixels of this tile (spt == 1)

  // LDS float offsets of this lane's A rows (tap (0,0)) and B rows (tap 0)
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
    const int hp = (MODE == CONV_S2) ? (s * HR + 2 * r) * WR + 2 * x : (s * HR + r) * WR + x;
    abase[mt] = hp * LDP + h * 8;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 32 + l31) * LDP + h * 8;

  // Accumulators start from bias (+ skip bias) + time embedding (+ identity residual):
  // the residual tile is fetched while the first K-chunk is being staged instead of in a
  // serial epilogue tail.  D layout: column (channel) = lane & 31,
  // row (pixel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
  const int bw = (g.spt == 1) ? b0 : b0 + wave;  // sample of this wave's segment
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
      if (a.temb && sample_ok) v += a.temb[((size_t)(a.temb_per_row ? bw : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c];
      add0[nt] = v;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int p = 64 * wave + pl;
        const bool valid = (g.spt == 1) ? (p < nvalid) : (sample_ok && pl < HW);
        const size_t pix = pix0 + ((g.spt == 1) ? p : pl);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v = add0[nt];
          if (a.res_mode == 1 && valid) v += a.res0[pix * a.Cout + n0 + nt * 32 + l31];
          acc[mt][nt][r] = v;
        }
      }
  }

  const int cin = a.C0 + a.C1;
  const int nch_main = cin / KC;
  const int nch_skip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;

  for (int ch = 0; ch < nch_main + nch_skip; ++ch) {
    const bool skip = ch >= nch_main;
    __syncthreads();  // previous chunk's LDS reads are done
    if (!skip) {
      stage_input<MODE>(sA, a.in0, a.in1, a.C0, a.C1, a.Hin, a.Win, a.ab, ch * KC, a.B, b0, row0, H,
                        W, HR, WR, a.halo_px, tid, 256);
      const float* wsrc = a.wpk + ((size_t)(blockIdx.y * nch_main + ch) * 9) * (32 * NT * KC);
      for (int it = tid; it < 9 * 32 * NT * 4; it += 256) {
        const int row = it >> 2, q = it & 3;
        *reinterpret_cast<f32x4*>(sB + row * LDP + q * 4) =
            *reinterpret_cast<const f32x4*>(wsrc + (size_t)it * 4);
      }
    } else {
      const int cs = ch - nch_main;
      stage_input<CONV_S1>(sA, a.res0, a.res1, a.R0, a.R1, a.Hin, a.Win, nullptr, cs * KC, a.B, b0,
                           row0, H, W, HR, WR, a.halo_px, tid, 256);
      const float* wsrc = a.wskip + ((size_t)(blockIdx.y * nch_skip + cs)) * (32 * NT * KC);
      for (int it = tid; it < 32 * NT * 4; it += 256) {
        const int row = it >> 2, q = it & 3;
        *reinterpret_cast<f32x4*>(sB + row * LDP + q * 4) =
            *reinterpret_cast<const f32x4*>(wsrc + (size_t)it * 4);
      }
    }
    __syncthreads();

    const int tap_lo = skip ? 4 : 0, tap_hi = skip ? 5 : 9;
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int aoff = (ky * WR + kx) * LDP;
      const int boff = (skip ? 0 : tap) * (32 * NT * LDP);
      float af[2][8], bf[NT][8];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff + 4);
        af[mt][0] = v0.x, af[mt][1] = v0.y, af[mt][2] = v0.z, af[mt][3] = v0.w;
        af[mt][4] = v1.x, af[mt][5] = v1.y, af[mt][6] = v1.z, af[mt][7] = v1.w;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff + 4);
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
    }
  }

  // ---------------------------------------------------------------- epilogue
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
      const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;  // pixel within the wave segment
      const int p = 64 * wave + pl;
      const bool valid = (g.spt == 1) ? (p < nvalid) : (sample_ok && pl < HW);
      if (valid) vmask[mt] |= 1u << r;
      const size_t pix = pix0 + ((g.spt == 1) ? p : pl);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        float v = acc[mt][nt][r];
        if (a.ep_scale) v = a.ep_nosilu ? v * eps_[nt] + eph_[nt] : silu_f(v * eps_[nt] + eph_[nt]);
        acc[mt][nt][r] = v;
        if (valid) a.out[pix * a.Cout + c] = v;
      }
    }
  if (a.small_check && a.range_flag) {  // (ConvArgs::small_check: see conv_mfma_bx3.hip; wave-uniform)
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

  if (a.stats_out) {
    int nw;
    if (g.spt == 1) {
      nw = nvalid - 64 * wave;
      nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
    } else {
      nw = sample_ok ? HW : 0;
    }
    const int part = (g.spt == 1) ? (blockIdx.x - b0 * g.tps) * 4 + wave : 0;
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
        store_stats(a, a.stats_out + (((size_t)bw * g.nparts + part) * a.Cout + c) * 2, mean, m2);
      }
    }
    if (a.fin_ab && sample_ok) fin_arrive(a, bw, lane, g.nparts, false);
  }
}


// ------------------------------------------------------------------------------------
// Software-prefetched variant (stride-1 and upsample modes; the production path).
// Same tiling and arithmetic as conv_mfma_kernel; differences:
//   * the halo-item decode (3 integer divisions per item) is done once per block;
//   * the global loads of K-chunk k+1 (activations, weights, GroupNorm scale/shift)
//     are issued into registers BEFORE the 288 MFMAs of chunk k and only transformed
//     and written to LDS after them, so their latency hides under the matrix work
//     instead of stalling both co-resident blocks (which run in lockstep).
// ------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) f32x4 pf_gf32x4;  // explicit global loads (never flat)
constexpr int MAXIT = 7;  // ceil(max halo_px * 4 / 256): halo_px <= 448 in these modes

template <int NT, int MODE>
__global__ __launch_bounds__(256, 2) void conv_mfma_pf_kernel(const ConvArgs a) {
  // CONV_T2: one output-parity class (blockIdx.z) of a ConvTranspose2d(k=4, s=2, p=1): a 2x2-tap conv on
  // the input raster whose result lands on every other pixel of the 2H x 2W output.
  constexpr int NTAPS = (MODE == CONV_T2) ? 4 : 9;
  constexpr int NB = (NTAPS * 32 * NT * 4 + 255) / 256;  // weight float4 items per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;
  float* sB = smem + a.halo_px * LDP;
  float* sAB = sB + NTAPS * 32 * NT * LDP;  // [spt][16][2] scale/shift of the chunk being committed

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31p = lane & 31, hp_ = lane >> 5;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;

  int b0, row0;
  if (g.spt == 1) {
    b0 = blockIdx.x / g.tps;
    row0 = (blockIdx.x - b0 * g.tps) * g.th;
  } else {
    b0 = blockIdx.x * g.spt;
    row0 = 0;
  }
  const int n0 = blockIdx.y * (32 * NT);
  const int pc = (MODE == CONV_T2) ? (int)blockIdx.z : 0, py = pc >> 1, px = pc & 1;
  const int HR = g.th + 2, WR = W + 2;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;

  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + l31p;
    int s, q;
    if (g.spt == 1) {
      s = 0;
      q = p < nvalid ? p : nvalid - 1;
    } else {
      s = wave;
      q = (p & 63) < HW ? (p & 63) : HW - 1;
    }
    const int r = q / W, x = q - r * W;
    abase[mt] = ((s * HR + r) * WR + x) * LDP + hp_ * 8;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 32 + l31p) * LDP + hp_ * 8;

  const int bw = (g.spt == 1) ? b0 : b0 + wave;
  const bool sample_ok = bw < a.B;
  const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W : (size_t)bw * HW;
  f32x16 acc[2][NT];
  {
    float add0[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = n0 + nt * 32 + l31p;
      float v = a.bias[c];
      if (a.res_mode == 2) v += a.skip_bias[c];
      if (a.temb && sample_ok) v += a.temb[((size_t)(a.temb_per_row ? bw : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c];
      add0[nt] = v;
    }
    if (a.res_mode == 1) {
      // identity residual: unconditional loads from a clamped pixel -- a per-element branch
      // around each load makes hipcc wait for every load before issuing the next one
      // (64 serial L2 round trips per lane).
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pl = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hp_;
          const int p = 64 * wave + pl;
          const bool valid = (g.spt == 1) ? (p < nvalid) : (sample_ok && pl < HW);
          const size_t pix = valid ? pix0 + ((g.spt == 1) ? p : pl) : (sample_ok ? pix0 : 0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = a.res0[pix * a.Cout + n0 + nt * 32 + l31p];
        }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += add0[nt];
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = add0[nt];
    }
  }

  // ---- per-item decode, once: source pixel offset, validity bit, sample index
  const int q4 = tid & 3;
  const int nA = a.halo_px * 4;
  int poff[MAXIT];
  unsigned okmask = 0u, smask = 0u;
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int it = tid + 256 * j;
      poff[j] = 0;
      if (it < nA) {
        const int hp = it >> 2;
        const int s = hp / per;
        const int rem = hp - s * per;
        const int hy = rem / WR, hx = rem - hy * WR;
        const int b = b0 + s;
        int y, x;
        bool ok;
        if (MODE == CONV_S1 || MODE == CONV_T2) {
          y = row0 + hy - 1, x = hx - 1;
          ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        } else {
          const int yu = row0 + hy - 1, xu = hx - 1;
          ok = (yu >= 0) && (yu < H) && (xu >= 0) && (xu < W);
          y = yu >> 1, x = xu >> 1;
        }
        ok = ok && (b < a.B);
        if (ok) {
          poff[j] = (b * a.Hin + y) * a.Win + x;
          okmask |= 1u << j;
          smask |= (unsigned)s << (2 * j);
        }
      }
    }
  }

  const int cin = a.C0 + a.C1;
  const int nch_main = cin / KC;
  const int nch_skip = (a.res_mode == 2) ? (a.R0 + a.R1) / KC : 0;
  const int ntot = nch_main + nch_skip;

  f32x4 ra[MAXIT], rb[NB], rab;
  rab = f32x4{1.f, 0.f, 1.f, 0.f};

  auto issue = [&](int ch) {
    const bool skip = ch >= nch_main;
    const float* src;
    int cs, cc, c;
    if (!skip) {
      c = ch * KC;
      if (c < a.C0) src = a.in0, cs = a.C0, cc = c;
      else src = a.in1, cs = a.C1, cc = c - a.C0;
    } else {
      c = (ch - nch_main) * KC;
      if (c < a.R0) src = a.res0, cs = a.R0, cc = c;
      else src = a.res1, cs = a.R1, cc = c - a.R0;
    }
    // Unconditional, clamped loads (padding / out-of-range items read pixel 0 and are zeroed at commit).
    // A load under a per-item branch merges with the zero-initialised register at the join, which makes
    // hipcc wait vmcnt(0) right there -- i.e. BEFORE the MFMAs the prefetch is supposed to hide under.
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) ra[j] = *(const pf_gf32x4*)(src + (size_t)poff[j] * cs + cc + q4 * 4);
    const float* wsrc = skip ? a.wskip + ((size_t)(blockIdx.y * nch_skip + (ch - nch_main))) * (32 * NT * KC)
                             : a.wpk + ((size_t)((pc * gridDim.y + blockIdx.y) * nch_main + ch) * NTAPS) * (32 * NT * KC);
    const int nbit = skip ? 32 * NT * 4 : NTAPS * 32 * NT * 4;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 256 * j;
      rb[j] = *(const pf_gf32x4*)(wsrc + (size_t)(it < nbit ? it : 0) * 4);
    }
    if (!skip && a.ab) {  // wave-uniform; lanes beyond the table read entry 0 (never used)
      const bool use = tid < g.spt * 8 && b0 + (tid >> 3) < a.B;
      const size_t o = use ? ((size_t)(b0 + (tid >> 3)) * cin + c + 2 * (tid & 7)) * 2 : 0;
      rab = *(const pf_gf32x4*)(a.ab + o);
    }
  };

  auto commit = [&](int ch) {
    const bool skip = ch >= nch_main;
    const bool xform = !skip && (a.ab != nullptr);
    if (xform && tid < g.spt * 8) *reinterpret_cast<f32x4*>(sAB + tid * 4) = rab;
    __syncthreads();  // every wave is done reading the previous chunk; sAB is visible after it
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      const int it = tid + 256 * j;
      if (it < nA) {
        f32x4 v = ra[j];
        const bool okj = (okmask >> j) & 1u;
        if (!okj) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xform && okj) {
          const int s = (smask >> (2 * j)) & 3u;
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(sAB + (s * 16 + q4 * 4) * 2);
          const f32x4 e1 = *reinterpret_cast<const f32x4*>(sAB + (s * 16 + q4 * 4) * 2 + 4);
          v.x = silu_fast(e0.x * v.x + e0.y);
          v.y = silu_fast(e0.z * v.y + e0.w);
          v.z = silu_fast(e1.x * v.z + e1.y);
          v.w = silu_fast(e1.z * v.w + e1.w);
        }
        *reinterpret_cast<f32x4*>(sA + (it >> 2) * LDP + q4 * 4) = v;
      }
    }
    const int nbit = skip ? 32 * NT * 4 : NTAPS * 32 * NT * 4;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 256 * j;
      if (it < nbit) *reinterpret_cast<f32x4*>(sB + (it >> 2) * LDP + q4 * 4) = rb[j];
    }
    __syncthreads();
  };

  issue(0);
  commit(0);
  for (int ch = 0; ch < ntot; ++ch) {
    const bool skip = ch >= nch_main;
    if (ch + 1 < ntot) issue(ch + 1);
    const int tap_lo = skip ? 4 : 0, tap_hi = skip ? 5 : NTAPS;
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
      int ky, kx;
      if (MODE == CONV_T2) ky = py + (tap >> 1), kx = px + (tap & 1);
      else ky = tap / 3, kx = tap - 3 * ky;
      const int aoff = (ky * WR + kx) * LDP;
      const int boff = (skip ? 0 : tap) * (32 * NT * LDP);
      float af[2][8], bf[NT][8];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sA + abase[mt] + aoff + 4);
        af[mt][0] = v0.x, af[mt][1] = v0.y, af[mt][2] = v0.z, af[mt][3] = v0.w;
        af[mt][4] = v1.x, af[mt][5] = v1.y, af[mt][6] = v1.z, af[mt][7] = v1.w;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sB + bbase[nt] + boff + 4);
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
    }
    if (ch + 1 < ntot) commit(ch + 1);
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_kernel)
  // Launder the lane id: otherwise the compiler keeps the accumulator-init addresses
  // (32 x 64-bit per lane) alive across the whole K loop to reuse them here.
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
      size_t pix = pix0 + ((g.spt == 1) ? p : pl);
      if (MODE == CONV_T2) {  // scatter to (2r + py, 2x + px) of the 2H x 2W raster
        const int pp = (g.spt == 1) ? row0 * W + p : pl;
        const int rr = pp / W, xx = pp - rr * W;
        pix = ((size_t)bw * (2 * H) + 2 * rr + py) * (2 * W) + 2 * xx + px;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        float v = acc[mt][nt][r];
        if (a.ep_scale) v = a.ep_nosilu ? v * eps_[nt] + eph_[nt] : silu_f(v * eps_[nt] + eph_[nt]);
        acc[mt][nt][r] = v;
        if (valid) a.out[pix * a.Cout + c] = v;
      }
    }
  if (a.small_check && a.range_flag) {  // (ConvArgs::small_check: see conv_mfma_bx3.hip; wave-uniform)
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
  if (a.stats_out) {
    int nw;
    if (g.spt == 1) {
      nw = nvalid - 64 * wave;
      nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
    } else {
      nw = sample_ok ? HW : 0;
    }
    // CONV_T2: the four parity classes are separate statistics parts (gn_finalize rep = 4)
    const int nparts = (MODE == CONV_T2) ? 4 * g.nparts : g.nparts;
    const int part = ((g.spt == 1) ? (blockIdx.x - b0 * g.tps) * 4 + wave : 0) + pc * g.nparts;
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
        store_stats(a, a.stats_out + (((size_t)bw * nparts + part) * a.Cout + c) * 2, mean, m2);
      }
    }
    if (a.fin_ab && sample_ok) fin_arrive(a, bw, lane_e, nparts, MODE == CONV_T2);
  }
}

// debug switch (RGFM_CONV_SIMPLE=1): route every conv through the non-prefetching kernel
static const bool g_conv_force_simple = getenv("RGFM_CONV_SIMPLE") != nullptr;

size_t conv_mfma_lds_bytes(const ConvArgs& a) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;  // (sized for 9 taps; CONV_T2 uses the first 4)
  return (size_t)(a.halo_px + 9 * 32 * nt) * LDP * sizeof(float) + 128 * sizeof(float);
}

template <int NT, int MODE>
static int raise_lds() {
  int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<NT, MODE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (MODE != CONV_S2)
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_pf_kernel<NT, MODE == CONV_S2 ? CONV_S1 : MODE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

int conv_mfma_init() {
  int rc = 0;
  rc |= raise_lds<1, CONV_S1>();
  rc |= raise_lds<1, CONV_S2>();
  rc |= raise_lds<1, CONV_UP2>();
  rc |= raise_lds<2, CONV_S1>();
  rc |= raise_lds<2, CONV_S2>();
  rc |= raise_lds<2, CONV_UP2>();
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_pf_kernel<1, CONV_T2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_pf_kernel<2, CONV_T2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

void launch_conv_mfma(const ConvArgs& a, int mode, hipStream_t s) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  dim3 grid(geom_num_tiles(a.g, a.B), a.Cout / (32 * nt), mode == CONV_T2 ? 4 : 1);
  const size_t lds = conv_mfma_lds_bytes(a);
#define LAUNCH(NTV, M) hipLaunchKernelGGL((conv_mfma_kernel<NTV, M>), grid, dim3(256), lds, s, a)
#define LAUNCH_PF(NTV, M) hipLaunchKernelGGL((conv_mfma_pf_kernel<NTV, M>), grid, dim3(256), lds, s, a)
  const bool pf = mode != CONV_S2 && a.halo_px * 4 <= MAXIT * 256 && !g_conv_force_simple;
  if (mode == CONV_T2) {  // prefetching kernel only (halo_px * 4 <= MAXIT * 256 is checked by the caller)
    if (nt == 2) LAUNCH_PF(2, CONV_T2);
    else LAUNCH_PF(1, CONV_T2);
  } else if (nt == 2) {
    if (mode == CONV_S2) LAUNCH(2, CONV_S2);
    else if (mode == CONV_S1) { if (pf) LAUNCH_PF(2, CONV_S1); else LAUNCH(2, CONV_S1); }
    else { if (pf) LAUNCH_PF(2, CONV_UP2); else LAUNCH(2, CONV_UP2); }
  } else {
    if (mode == CONV_S2) LAUNCH(1, CONV_S2);
    else if (mode == CONV_S1) { if (pf) LAUNCH_PF(1, CONV_S1); else LAUNCH(1, CONV_S1); }
    else { if (pf) LAUNCH_PF(1, CONV_UP2); else LAUNCH(1, CONV_UP2); }
  }
#undef LAUNCH
#undef LAUNCH_PF
}

}  // namespace rgfm
