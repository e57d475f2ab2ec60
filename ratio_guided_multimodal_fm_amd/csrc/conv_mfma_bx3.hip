// conv_mfma_bx3.hip -- the implicit-GEMM convolution of conv_mfma.hip with fp32 operands carried as
// three bf16 planes (a = a_h + a_m + a_l, an EXACT decomposition of the 24-bit significand) and the
// product formed on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) as
//     a*b ~= a_l*b_h + a_h*b_l + a_m*b_m + a_m*b_h + a_h*b_m + a_h*b_h        (6 MFMAs, small terms first)
// Every bf16 x bf16 product is exact in fp32; the dropped terms (a_m*b_l, a_l*b_m, a_l*b_l) are
// < 2^-23 |a*b|, i.e. below the rounding of a single fp32 multiply.  The bf16 pipe runs 16x the
// fp32-MFMA rate, so six passes still leave ~2.7x of MFMA headroom over v_mfma_f32_32x32x2_f32.
//
// Same fusion set, tiling, prefetch structure and epilogue as conv_mfma_pf_kernel (see conv_mfma.hip);
// differences: LDS records are 3 planes x 16 bf16 (+16 B pad = 112 B, conflict-free for ds_read_b128),
// the staging path splits the (GroupNorm+SiLU-transformed) activations, weights arrive pre-split from
// launch_pack_conv_bx3 / launch_pack_deconv_bx3, and one workgroup per CU (115 KB of LDS).
#include <stdlib.h>

#include "rgfm_device.h"

namespace rgfm {

typedef __attribute__((address_space(1))) f32x4 bx_gf32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RB = 112;         // bytes per LDS record
constexpr int BX_MAXIT = 7;     // halo items per thread (halo_px <= 448)

// exact 3-way bf16 split of two floats: planes h, m, l as packed bf16 pairs
__device__ __forceinline__ void split2(float a, float b, unsigned& ph, unsigned& pm, unsigned& pl) {
  f32x2 v = {a, b};
  ph = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 hf = {__builtin_bit_cast(float, ph << 16), __builtin_bit_cast(float, ph & 0xffff0000u)};
  v = v - hf;
  pm = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 mf = {__builtin_bit_cast(float, pm << 16), __builtin_bit_cast(float, pm & 0xffff0000u)};
  v = v - mf;
  pl = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NT, int MODE>
__global__ __launch_bounds__(256, 1) void conv_mfma_bx3_kernel(const ConvArgs a) {
  constexpr int NTAPS = (MODE == CONV_T2) ? 4 : 9;
  constexpr int NBI = NTAPS * 32 * NT * 6;     // 16-byte weight items per chunk (96 B per (tap, channel))
  constexpr int NB = (NBI + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem3[];
  char* sA = smem3;
  char* sB = smem3 + a.halo_px * RB;
  float* sAB = reinterpret_cast<float*>(sB + NTAPS * 32 * NT * RB);

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
    abase[mt] = ((s * HR + r) * WR + x) * RB + hp_ * 16;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (nt * 32 + l31p) * RB + hp_ * 16;

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
      if (a.temb && sample_ok) v += a.temb[(size_t)(a.temb_per_row ? bw : 0) * a.temb_stride + c];
      add0[nt] = v;
    }
    if (a.res_mode == 1) {
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
  int poff[BX_MAXIT];
  unsigned okmask = 0u, smask = 0u;
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < BX_MAXIT; ++j) {
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
  const char* wpk3 = reinterpret_cast<const char*>(a.wpk3);
  const char* wskip3 = reinterpret_cast<const char*>(a.wskip3);

  f32x4 ra[BX_MAXIT], rb[NB], rab;
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
#pragma unroll
    for (int j = 0; j < BX_MAXIT; ++j) ra[j] = *(const bx_gf32x4*)(src + (size_t)poff[j] * cs + cc + q4 * 4);
    // packed weights: 96 B per (tap, channel): [3 planes][16 k] bf16
    const char* wsrc = skip ? wskip3 + ((size_t)(blockIdx.y * nch_skip + (ch - nch_main))) * (32 * NT * 96)
                            : wpk3 + ((size_t)((pc * gridDim.y + blockIdx.y) * nch_main + ch) * NTAPS) * (32 * NT * 96);
    const int nbit = skip ? 32 * NT * 6 : NBI;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 256 * j;
      rb[j] = *(const bx_gf32x4*)(wsrc + (size_t)(it < nbit ? it : 0) * 16);
    }
    if (!skip && a.ab) {
      const bool use = tid < g.spt * 8 && b0 + (tid >> 3) < a.B;
      const size_t o = use ? ((size_t)(b0 + (tid >> 3)) * cin + c + 2 * (tid & 7)) * 2 : 0;
      rab = *(const bx_gf32x4*)(a.ab + o);
    }
  };

  auto commit = [&](int ch) {
    const bool skip = ch >= nch_main;
    const bool xform = !skip && (a.ab != nullptr);
    if (xform && tid < g.spt * 8) *reinterpret_cast<f32x4*>(sAB + tid * 4) = rab;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BX_MAXIT; ++j) {
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
        unsigned h0, m0, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0, l0);
        split2(v.z, v.w, h1, m1, l1);
        const u32x2 ph = {h0, h1}, pm = {m0, m1}, pl = {l0, l1};
        char* dst = sA + (it >> 2) * RB + q4 * 8;
        *reinterpret_cast<u32x2*>(dst) = ph;
        *reinterpret_cast<u32x2*>(dst + 32) = pm;
        *reinterpret_cast<u32x2*>(dst + 64) = pl;
      }
    }
    const int nbit = skip ? 32 * NT * 6 : NBI;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int it = tid + 256 * j;
      // item it = record (it / 6), 16-byte slot (it % 6) of its 96 payload bytes
      if (it < nbit) *reinterpret_cast<f32x4*>(sB + (it / 6) * RB + (it % 6) * 16) = rb[j];
    }
    __syncthreads();
  };

  issue(0);
  commit(0);
  for (int ch = 0; ch < ntot; ++ch) {
    const bool skip = ch >= nch_main;
    if (ch + 1 < ntot) issue(ch + 1);
    const int tap_lo = skip ? 4 : 0, tap_hi = skip ? 5 : NTAPS;
#pragma unroll 1
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
      int ky, kx;
      if (MODE == CONV_T2) ky = py + (tap >> 1), kx = px + (tap & 1);
      else ky = tap / 3, kx = tap - 3 * ky;
      const int aoff = (ky * WR + kx) * RB;
      const int boff = (skip ? 0 : tap) * (32 * NT * RB);
      bf16x8 af[2][3], bf[NT][3];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[mt][p] = *reinterpret_cast<const bf16x8*>(sA + abase[mt] + aoff + p * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[nt][p] = *reinterpret_cast<const bf16x8*>(sB + bbase[nt] + boff + p * 32);
      // plane pairs (activation, weight), smallest products first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][PA[q]], bf[nt][PB[q]], acc[mt][nt], 0, 0, 0);
    }
    if (ch + 1 < ntot) commit(ch + 1);
  }

  // ---------------------------------------------------------------- epilogue (as conv_mfma_pf_kernel)
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
      if (MODE == CONV_T2) {
        const int pp = (g.spt == 1) ? row0 * W + p : pl;
        const int rr = pp / W, xx = pp - rr * W;
        pix = ((size_t)bw * (2 * H) + 2 * rr + py) * (2 * W) + 2 * xx + px;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        float v = acc[mt][nt][r];
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
        float2 st;
        st.x = mean;
        st.y = m2;
        *reinterpret_cast<float2*>(a.stats_out + (((size_t)bw * nparts + part) * a.Cout + c) * 2) = st;
      }
    }
  }
}

// ---------------------------------------------------------------- weight packing (exact 3-way split)
__device__ __forceinline__ void split1(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
  unsigned ph, pm, pl;
  split2(v, 0.f, ph, pm, pl);
  h = (unsigned short)(ph & 0xffffu), m = (unsigned short)(pm & 0xffffu), l = (unsigned short)(pl & 0xffffu);
}

// [Cout][Cin][taps] fp32 -> [Cout/nb][Cin/16][taps][nb][3][16] bf16
__global__ void pack_conv_bx3_kernel(const float* w, unsigned short* out, int Cout, int Cin, int taps, int nb) {
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
    unsigned short h, m, l;
    split1(w[((size_t)co * Cin + ci) * taps + tap], h, m, l);
    unsigned short* rec = out + (i / 16) * 48;
    rec[kk] = h, rec[16 + kk] = m, rec[32 + kk] = l;
  }
}

// ConvTranspose2d [Cin][Cout][4][4] -> [4 parity][Cout/nb][Cin/16][4 taps][nb][3][16] bf16 (see pack_deconv_kernel)
__global__ void pack_deconv_bx3_kernel(const float* w, unsigned short* out, int Cin, int Cout, int nb) {
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
    unsigned short h, m, l;
    split1(w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx], h, m, l);
    unsigned short* rec = out + (i / 16) * 48;
    rec[kk] = h, rec[16 + kk] = m, rec[32 + kk] = l;
  }
}

void launch_pack_conv_bx3(const float* w, void* out, int Cout, int Cin, int taps, int nt32, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_bx3_kernel, dim3(256), dim3(256), 0, s, w, (unsigned short*)out, Cout, Cin, taps, 32 * nt32);
}
void launch_pack_deconv_bx3(const float* w, void* out, int Cin, int Cout, int nt32, hipStream_t s) {
  hipLaunchKernelGGL(pack_deconv_bx3_kernel, dim3(256), dim3(256), 0, s, w, (unsigned short*)out, Cin, Cout, 32 * nt32);
}

static size_t bx3_lds_bytes(const ConvArgs& a, int mode) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  const int ntaps = mode == CONV_T2 ? 4 : 9;
  return (size_t)(a.halo_px + ntaps * 32 * nt) * RB + 128 * sizeof(float);
}

bool conv_bx3_supported(const ConvArgs& a, int mode) {
  if (mode == CONV_S2 || !a.wpk3) return false;
  if (a.res_mode == 2 && !a.wskip3) return false;
  return a.halo_px * 4 <= BX_MAXIT * 256 && bx3_lds_bytes(a, mode) <= 160 * 1024;
}

int conv_bx3_init() {
  int rc = 0;
#define RAISE(NTV, M) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_bx3_kernel<NTV, M>), \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  RAISE(1, CONV_S1); RAISE(1, CONV_UP2); RAISE(1, CONV_T2);
  RAISE(2, CONV_S1); RAISE(2, CONV_UP2); RAISE(2, CONV_T2);
#undef RAISE
  return rc;
}

void launch_conv_bx3(const ConvArgs& a, int mode, hipStream_t s) {
  const int nt = (a.Cout % 64 == 0) ? 2 : 1;
  dim3 grid(geom_num_tiles(a.g, a.B), a.Cout / (32 * nt), mode == CONV_T2 ? 4 : 1);
  const size_t lds = bx3_lds_bytes(a, mode);
#define LAUNCH3(NTV, M) hipLaunchKernelGGL((conv_mfma_bx3_kernel<NTV, M>), grid, dim3(256), lds, s, a)
  if (nt == 2) {
    if (mode == CONV_S1) LAUNCH3(2, CONV_S1);
    else if (mode == CONV_UP2) LAUNCH3(2, CONV_UP2);
    else LAUNCH3(2, CONV_T2);
  } else {
    if (mode == CONV_S1) LAUNCH3(1, CONV_S1);
    else if (mode == CONV_UP2) LAUNCH3(1, CONV_UP2);
    else LAUNCH3(1, CONV_T2);
  }
#undef LAUNCH3
}

}  // namespace rgfm
