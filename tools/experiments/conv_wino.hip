// conv_wino.hip -- Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution on the fp32 matrix
// cores: 16 multiplies per 2x2 output tile and input channel instead of 36 (2.25x fewer MFMAs).
//
// Replaces the same reference ops as conv_mfma.hip (Conv2d 3x3 stride 1 incl. the Upsample conv,
// GroupNorm+SiLU apply and concat on the load path, bias / time-embedding / identity residual,
// GroupNorm partial statistics of the output; src/models/unet_flexible.py:71-108) for the layers
// where it applies: H = W in {8, 16, 32}, Cout % 64 == 0, no fused 1x1 skip conv.  Everything is
// exact fp32 (v_mfma_f32_32x32x2_f32 + fp32 adds); only the summation order differs from the direct
// form, so it stays inside the stated parity tolerance.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter
//
// Why it pays here: on gfx950 the f32 MFMA shares the SIMD issue budget with VALU/LDS work
// (tools/ubench), so the transforms are paid at full price -- B^T d B costs 32 adds per patch and
// channel, A^T m A 24 adds per tile and output channel -- but that is ~3.5 k issue cycles per
// 16-channel chunk against 10 k MFMA cycles saved.
//
// Structure: one 256-thread workgroup per CU; block tile = 64 Winograd tiles (256 output pixels, the
// same pixel tiles and statistics segments as conv_mfma.hip) x 64 output channels; wave (wm, wn) owns
// 32 tiles x 32 channels x ALL 16 Winograd positions = sixteen 32x32 accumulators (256 AGPRs), so the
// output transform is register-local.  Per 16-channel chunk:
//   (a) prefetched raw halo tile -> GroupNorm+SiLU -> LDS image R            | barrier
//   (b) thread = (tile, 4-channel group): 16 ds_read_b128 of R -> B^T d B -> 16 ds_write_b128 into
//       V[pos][tile][16 ch]                                                  | barrier
//   (c) per position: A fragment from V (2 ds_read_b128), B fragment = transformed weights straight
//       from L2 (2 global_load_dwordx4, prefetched three positions ahead), 8 MFMAs.
#include <stdlib.h>

#include "rgfm_device.h"

namespace rgfm {

typedef __attribute__((address_space(1))) f32x4 wgf32x4;  // explicit global loads (never flat)

constexpr int WINO_MAXIT = 7;

// U[nblk][chunk][pos][64][16] = (G g G^T)[pos] for (cout = nblk*64 + n, cin = chunk*16 + k)
__global__ void wino_pack_kernel(const float* w, float* out, int Cout, int Cin) {
  const size_t total = (size_t)Cout * Cin;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = i % Cin;
    const int co = i / Cin;
    const float* g = w + ((size_t)co * Cin + ci) * 9;
    // t = G g  (4x3), u = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
      t[0][j] = g[j];
      t[1][j] = 0.5f * (g[j] + g[3 + j] + g[6 + j]);
      t[2][j] = 0.5f * (g[j] - g[3 + j] + g[6 + j]);
      t[3][j] = g[6 + j];
    }
    const int nblk = co / 64, n = co % 64, ch = ci / 16, k = ci % 16, nch = Cin / 16;
    for (int r = 0; r < 4; ++r) {
      const float u0 = t[r][0];
      const float u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
      const float u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
      const float u3 = t[r][2];
      const float u[4] = {u0, u1, u2, u3};
      for (int c = 0; c < 4; ++c) {
        const int pos = r * 4 + c;
        out[((((size_t)nblk * nch + ch) * 16 + pos) * 64 + n) * 16 + k] = u[c];
      }
    }
  }
}

void launch_wino_pack(const float* w, float* out, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(wino_pack_kernel, dim3(256), dim3(256), 0, s, w, out, Cout, Cin);
}

template <int MODE, bool SPT4>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a, const float* __restrict__ upk, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const sR = smem;                       // [halo_px][LDP]
  float* const sV = smem + a.halo_px * LDP;     // [16][64][LDP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int q4 = tid & 3;
  const TileGeom g = a.g;
  const int W = g.W, H = g.H, HW = g.HW;
  const int HR = g.th + 2, WR = W + 2;
  const int TW = W >> 1;  // Winograd tiles per tile row

  int b0, row0;
  if (g.spt == 1) {
    b0 = blockIdx.x / g.tps;
    row0 = (blockIdx.x - b0 * g.tps) * g.th;
  } else {
    b0 = blockIdx.x * g.spt;
    row0 = 0;
  }
  const int nblk = blockIdx.y;
  const int n0 = nblk * 64 + 32 * wn;  // first output channel of this wave
  const int cin = a.C0 + a.C1;
  const int nch = cin / KC;
  const int nA = a.halo_px * 4;

  // ---- staging items (raw halo tile), decoded once
  int poff[WINO_MAXIT];
  unsigned okmask = 0u, smask = 0u;
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < WINO_MAXIT; ++j) {
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
          poff[j] = (b * a.Hin + y) * a.Win + x;
          okmask |= 1u << j;
          smask |= (unsigned)s << (2 * j);
        }
      }
    }
  }
  constexpr int NE = SPT4 ? WINO_MAXIT : 1;
  f32x4 ra[WINO_MAXIT], re0[NE], re1[NE];
  const bool xform = a.ab != nullptr;

  auto issue = [&](int ch) {
    const int c = ch * KC;
    const bool second = c >= a.C0;
    const float* src = second ? a.in1 : a.in0;
    const int cs = second ? a.C1 : a.C0;
    const int cc = second ? c - a.C0 : c;
#pragma unroll
    for (int j = 0; j < WINO_MAXIT; ++j) {
      ra[j] = *(const wgf32x4*)(src + (size_t)poff[j] * cs + cc + q4 * 4);
      if (xform && (SPT4 || j == 0)) {
        int bb = b0 + (SPT4 ? (int)((smask >> (2 * j)) & 3u) : 0);
        bb = bb < a.B ? bb : 0;
        const wgf32x4* p = (const wgf32x4*)(a.ab + ((size_t)bb * cin + c + q4 * 4) * 2);
        re0[SPT4 ? j : 0] = p[0];
        re1[SPT4 ? j : 0] = p[1];
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < WINO_MAXIT; ++j) {
      const int it = tid + 256 * j;
      if (it < nA) {
        f32x4 v = ra[j];
        if ((okmask >> j) & 1u) {
          if (xform) {
            const f32x4 e0 = re0[SPT4 ? j : 0], e1 = re1[SPT4 ? j : 0];
            v.x = silu_fast(e0.x * v.x + e0.y);
            v.y = silu_fast(e0.z * v.y + e0.w);
            v.z = silu_fast(e1.x * v.z + e1.y);
            v.w = silu_fast(e1.z * v.w + e1.w);
          }
        } else {
          v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<f32x4*>(sR + (it >> 2) * LDP + q4 * 4) = v;
      }
    }
  };

  // ---- input-transform item of this thread: tile tt (0..63), channel group q4
  const int tt = tid >> 2;
  int pbase;  // halo pixel index of the patch's top-left corner
  if (g.spt == 1) {
    const int ty = tt / TW, tx = tt - ty * TW;
    pbase = (2 * ty) * WR + 2 * tx;
  } else {
    const int s = tt >> 4, t16 = tt & 15;  // 16 tiles per 8x8 sample
    const int ty = t16 / TW, tx = t16 - ty * TW;
    pbase = (s * HR + 2 * ty) * WR + 2 * tx;
  }

  f32x16 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

  const int arow = (32 * wm + l31) * LDP + h * 8;  // A-fragment row inside a position's V slab
  const float* ub = upk + ((size_t)nblk * nch * 16 * 64 + 32 * wn + l31) * 16 + h * 8;

  issue(0);
  f32x4 bq[4][2];  // B (transformed-weight) fragments of the next four positions
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    bq[p][0] = *(const wgf32x4*)(ub + (size_t)p * 64 * 16);
    bq[p][1] = *(const wgf32x4*)(ub + (size_t)p * 64 * 16 + 4);
  }
  for (int ch = 0; ch < nch; ++ch) {
    // (a) raw halo tile -> R
    if (!(dbg & 4)) commit();
    __syncthreads();
    if (ch + 1 < nch) issue(ch + 1);  // in flight during (b) and (c)
    // (b) B^T d B for one (tile, 4 channels) per thread
    if (!(dbg & 2)) {
      f32x4 d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          d[i][j] = *reinterpret_cast<const f32x4*>(sR + (pbase + i * WR + j) * LDP + q4 * 4);
      f32x4 t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 v0 = t[i][0] - t[i][2];
        const f32x4 v1 = t[i][1] + t[i][2];
        const f32x4 v2 = t[i][2] - t[i][1];
        const f32x4 v3 = t[i][1] - t[i][3];
        float* dst = sV + ((i * 4) * 64 + tt) * LDP + q4 * 4;
        *reinterpret_cast<f32x4*>(dst) = v0;
        *reinterpret_cast<f32x4*>(dst + 64 * LDP) = v1;
        *reinterpret_cast<f32x4*>(dst + 2 * 64 * LDP) = v2;
        *reinterpret_cast<f32x4*>(dst + 3 * 64 * LDP) = v3;
      }
    }
    __syncthreads();
    if (!(dbg & 1))
    // (c) 16 position GEMMs: acc[pos] += V[pos] (32 tiles x 16 ch) * U[pos] (16 ch x 32 couts).
    // B fragments come straight from L2, four positions ahead; the first four of a chunk were
    // issued before the previous chunk's GEMMs finished (or before the loop), so their latency
    // hides under phases (a) and (b) instead of stalling the first MFMA of every chunk.
    {
      const float* uc = ub + (size_t)ch * 16 * 64 * 16;
      f32x4 aq[2][2];
      aq[0][0] = *reinterpret_cast<const f32x4*>(sV + arow);
      aq[0][1] = *reinterpret_cast<const f32x4*>(sV + arow + 4);
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int as = p & 1, bs = p & 3;
        if (p + 1 < 16) {
          aq[as ^ 1][0] = *reinterpret_cast<const f32x4*>(sV + (p + 1) * 64 * LDP + arow);
          aq[as ^ 1][1] = *reinterpret_cast<const f32x4*>(sV + (p + 1) * 64 * LDP + arow + 4);
        }
        const f32x4 a0 = aq[as][0], a1 = aq[as][1], b0v = bq[bs][0], b1v = bq[bs][1];
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0v.x, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0v.y, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0v.z, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0v.w, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1v.x, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1v.y, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1v.z, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1v.w, acc[p], 0, 0, 0);
        // refill this slot: position p+4 of this chunk, or position p-12 of the next chunk
        const bool wrap = p + 4 >= 16;
        const float* nu = wrap ? uc + (size_t)16 * 64 * 16 + (size_t)(p + 4 - 16) * 64 * 16 : uc + (size_t)(p + 4) * 64 * 16;
        if (!wrap || ch + 1 < nch) {
          bq[bs][0] = *(const wgf32x4*)(nu);
          bq[bs][1] = *(const wgf32x4*)(nu + 4);
        }
        // pin the software pipeline: without this hipcc sinks the refill next to its use (one
        // position ahead), which exposes the L2 latency at every position
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // no barrier here: the next (a) only rewrites R, which (c) does not read; the barrier after it
    // orders every wave's (c) before anyone's next (b) overwrites V
  }

  if (dbg & 8) return;
  // ---------------------------------------------------------------- epilogue: Y = A^T M A
  // lane = output channel n0 + l31; accumulator row r <-> tile T = 32 wm + (r&3) + 8 (r>>2) + 4 h.
  const int c = n0 + l31;
  const int segbase = 2 * wm;  // the wave's two 64-pixel statistics segments
  float addv = a.bias[c];
  const float eps_ = a.ep_scale ? a.ep_scale[c] : 1.f, eph_ = a.ep_scale ? a.ep_shift[c] : 0.f;
  float sum[2] = {0.f, 0.f};
  float yv[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int T = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
    int bw, pl;  // sample and pixel (within the sample raster) of the tile's top-left output pixel
    if (g.spt == 1) {
      const int ty = T / TW, tx = T - ty * TW;
      bw = b0;
      pl = (row0 + 2 * ty) * W + 2 * tx;
    } else {
      const int t16 = T & 15;
      const int ty = t16 / TW, tx = t16 - ty * TW;
      bw = b0 + (T >> 4);
      pl = (2 * ty) * W + 2 * tx;
    }
    const bool ok = bw < a.B;
    float m[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p] = acc[p][r];
    // rows: R_i = [m_i0 + m_i1 + m_i2, m_i1 - m_i2 - m_i3]
    float r0[4], r1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r0[i] = m[4 * i] + m[4 * i + 1] + m[4 * i + 2];
      r1[i] = m[4 * i + 1] - m[4 * i + 2] - m[4 * i + 3];
    }
    float y[4];
    y[0] = r0[0] + r0[1] + r0[2];
    y[1] = r1[0] + r1[1] + r1[2];
    y[2] = r0[1] - r0[2] - r0[3];
    y[3] = r1[1] - r1[2] - r1[3];
    float tadd = addv;
    if (a.temb && ok) tadd += a.temb[(size_t)(a.temb_per_row ? bw : 0) * a.temb_stride + c];
    const size_t pix = (size_t)(ok ? bw : 0) * HW + pl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t px = pix + (k >> 1) * W + (k & 1);
      float v = y[k] + tadd;
      if (a.res_mode == 1) v += a.res0[px * a.Cout + c];
      if (a.ep_scale) v = silu_f(v * eps_ + eph_);
      if (ok) a.out[px * a.Cout + c] = v;
      v = ok ? v : 0.f;
      yv[r][k] = v;
      sum[r >> 3] += v;
    }
  }
  if (a.stats_out) {
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
      const int seg = segbase + sgi;  // segment inside the block tile
      const int bw = (g.spt == 1) ? b0 : b0 + seg;
      const bool ok = bw < a.B;
      float s = sum[sgi];
      s += __shfl_xor(s, 32);
      const float mean = s * (1.0f / 64.0f);
      float m2 = 0.f;
#pragma unroll
      for (int r = 8 * sgi; r < 8 * sgi + 8; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dd = yv[r][k] - mean;
          m2 += dd * dd;
        }
      m2 += __shfl_xor(m2, 32);
      if (h == 0 && ok) {
        const int part = (g.spt == 1) ? (blockIdx.x - b0 * g.tps) * 4 + seg : 0;
        float2 st;
        st.x = mean, st.y = m2;
        *reinterpret_cast<float2*>(a.stats_out + (((size_t)bw * g.nparts + part) * a.Cout + c) * 2) = st;
      }
    }
  }
}

size_t conv_wino_lds_bytes(const ConvArgs& a) { return (size_t)(a.halo_px + 16 * 64) * LDP * sizeof(float); }

bool conv_wino_supported(const ConvArgs& a, int mode) {
  if (mode == CONV_S2 || a.res_mode == 2) return false;
  const int S = a.g.H;
  if (a.g.W != S || (S != 8 && S != 16 && S != 32)) return false;
  if (a.Cout % 64 || (a.C0 % KC) || (a.C1 % KC)) return false;
  if (a.halo_px * 4 > WINO_MAXIT * 256) return false;
  return conv_wino_lds_bytes(a) <= 160 * 1024;
}

int conv_wino_init() {
  int rc = 0;
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<CONV_S1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<CONV_S1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<CONV_UP2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<CONV_UP2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}

void launch_conv_wino(const ConvArgs& a, int mode, const float* upk, hipStream_t s) {
  dim3 grid(geom_num_tiles(a.g, a.B), a.Cout / 64);
  const size_t lds = conv_wino_lds_bytes(a);
  const bool s4 = a.g.spt != 1;
  const char* de = getenv("RGFM_WINO_DBG");  // timing ablations only (outputs become wrong)
  const int dbg = de ? atoi(de) : 0;
  if (mode == CONV_S1) {
    if (s4) hipLaunchKernelGGL((conv_wino_kernel<CONV_S1, true>), grid, dim3(256), lds, s, a, upk, dbg);
    else hipLaunchKernelGGL((conv_wino_kernel<CONV_S1, false>), grid, dim3(256), lds, s, a, upk, dbg);
  } else {
    if (s4) hipLaunchKernelGGL((conv_wino_kernel<CONV_UP2, true>), grid, dim3(256), lds, s, a, upk, dbg);
    else hipLaunchKernelGGL((conv_wino_kernel<CONV_UP2, false>), grid, dim3(256), lds, s, a, upk, dbg);
  }
}

}  // namespace rgfm
