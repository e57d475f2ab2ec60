// unet_kernels.hip -- the non-GEMM kernels of a U-Net evaluation: first conv
// (NCHW image -> NHWC features), output head (+ fused Euler update), GroupNorm
// statistics finalisation, time-embedding MLP, weight packing, layout helpers.
// All are HBM/latency-bound; none is reshaped into a GEMM.
#include "rgfm_device.h"

namespace rgfm {

// ------------------------------------------------------------------ conv_in
// input_conv (src/models/unet_flexible.py:155, :219) and the first conv of the ratio encoders
// (src/models/ratio_flexible.py:195/:245, ratio_estimator.py:46): NCHW image -> NHWC features
// (+ bias, + optional folded-BatchNorm/SiLU epilogue, + GroupNorm partial statistics).
// Workgroup = one 256-pixel tile, wave = one 64-pixel segment.  lane = output channel; with 64
// (or more) channels the whole wave works on ONE pixel at a time, with 32 channels the two wave
// halves take alternate pixels.  The zero-padded halo is staged as one float4 (<= 4 input channels)
// per pixel, so a tap is a single aligned broadcast ds_read_b128: 9 LDS reads + 9*CIN FMAs per
// pixel (f32 VALU and LDS issue share the SIMD: the instruction count is the bound).
// conv_in_pk_kernel (C0 % 64 == 0, no epilogue activation: the U-Nets' input convs): TWO output channels per lane (c, c + 32) as
// one packed accumulator -- v_pk_fma_f32 with the tap value in both halves and the lane's weight pair -- and the two wave halves
// on alternate pixels, so a wave instruction covers two pixels x 64 channels.  The kernel is bound by its instruction count
// (9 LDS reads + 9 CIN multiply-adds per pixel and lane in the one-channel form): this halves both.
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_pk_kernel(const ConvInArgs a) {
  typedef float ci_f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int HR = g.th + 2, WR = W + 2, per = HR * WR;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;
  for (int it = tid; it < g.spt * per; it += 256) {
    const int s = it / per, rem = it - s * per;
    const int hy = rem / WR, hx = rem - hy * WR;
    const int y = row0 + hy - 1, x = hx - 1, b = b0 + s;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && x >= 0 && x < W && b < a.B) {
      const float* p = a.x + ((size_t)b * CIN * H + y) * W + x;
      v.x = p[0];
      if (CIN > 1) v.y = p[(size_t)HW];
      if (CIN > 2) v.z = p[(size_t)2 * HW];
    }
    *reinterpret_cast<f32x4*>(smem + 4 * it) = v;
  }
  __syncthreads();

  const int bw = (g.spt == 1) ? b0 : b0 + wave;
  const bool sample_ok = bw < a.B;
  const int sidx = (g.spt == 1) ? 0 : wave;
  int nw;
  if (g.spt == 1) {
    nw = nvalid - 64 * wave;
    nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
  } else {
    nw = sample_ok ? HW : 0;
  }
  const int q0 = (g.spt == 1) ? 64 * wave : 0;
  const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W + q0 : (size_t)bw * HW;
  const int part = (g.spt == 1) ? (blockIdx.x - b0 * g.tps) * 4 + wave : 0;
  const int hh = lane >> 5;
  for (int cg = 0; cg < a.C0 / 64; ++cg) {
    const int c = cg * 64 + (lane & 31);  // this lane's channels: c and c + 32
    ci_f32x2 wr[CIN * 9];
#pragma unroll
    for (int i = 0; i < CIN * 9; ++i) wr[i] = ci_f32x2{a.w[(size_t)c * CIN * 9 + i], a.w[(size_t)(c + 32) * CIN * 9 + i]};
    const ci_f32x2 bias = {a.bias[c], a.bias[c + 32]};
    ci_f32x2 sum = {0.f, 0.f}, sq = {0.f, 0.f}, pivot = {0.f, 0.f};
    float vmx = 0.f;
    int r = (q0 + hh) / W, x = (q0 + hh) - r * W;
#pragma unroll 2
    for (int k = hh; k < nw; k += 2) {
      const float* base = smem + 4 * (sidx * per + r * WR + x);
      ci_f32x2 acc = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(base + 4 * (ky * WR + kx));
          acc = __builtin_elementwise_fma(ci_f32x2{t.x, t.x}, wr[ky * 3 + kx], acc);
          if (CIN > 1) acc = __builtin_elementwise_fma(ci_f32x2{t.y, t.y}, wr[9 + ky * 3 + kx], acc);
          if (CIN > 2) acc = __builtin_elementwise_fma(ci_f32x2{t.z, t.z}, wr[18 + ky * 3 + kx], acc);
        }
      float* op = a.out + (pix0 + k) * a.C0 + c;
      op[0] = acc.x, op[32] = acc.y;
      vmx = fmaxf(vmx, fmaxf(fabsf(acc.x), fabsf(acc.y)));
      if (k == hh) pivot = acc;
      const ci_f32x2 d = acc - pivot;
      sum += d;
      sq = __builtin_elementwise_fma(d, d, sq);
      x += 2;
      if (x >= W) x -= W, ++r;
    }
    if (a.small_check && a.range_flag) {  // (ConvArgs::small_check; wave-uniform)
      float m = vmx;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (m > 0.f && m < HX_SMALL && lane == 0) atomicOr(a.range_flag, 2u);
    }
    if (a.stats_out && sample_ok && nw > 0) {
      const float nme = (float)((nw - hh + 1) / 2);  // pixels this lane saw
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float dm = nme > 0.f ? sum[e] / nme : 0.f;
        float mean = pivot[e] + dm;
        float m2 = sq[e] - nme * dm * dm;
        m2 = m2 < 0.f ? 0.f : m2;
        // Chan-combine the two halves (disjoint pixel sets of the same channel)
        const float no = __shfl_xor(nme, 32), mo = __shfl_xor(mean, 32), m2o = __shfl_xor(m2, 32);
        const float nt = nme + no;
        const float dl = mo - mean;
        m2 = m2 + m2o + dl * dl * nme * no / nt;
        mean = mean + dl * no / nt;
        if (hh == 0) {
          float2 st;
          st.x = mean, st.y = m2;
          *reinterpret_cast<float2*>(a.stats_out + (((size_t)bw * g.nparts + part) * a.C0 + c + 32 * e) * 2) = st;
        }
      }
    }
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_in_kernel(const ConvInArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int HR = g.th + 2, WR = W + 2, per = HR * WR;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;
  for (int it = tid; it < g.spt * per; it += 256) {
    const int s = it / per, rem = it - s * per;
    const int hy = rem / WR, hx = rem - hy * WR;
    const int y = row0 + hy - 1, x = hx - 1, b = b0 + s;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && x >= 0 && x < W && b < a.B) {
      const float* p = a.x + ((size_t)b * CIN * H + y) * W + x;
      v.x = p[0];
      if (CIN > 1) v.y = p[(size_t)HW];
      if (CIN > 2) v.z = p[(size_t)2 * HW];
    }
    *reinterpret_cast<f32x4*>(smem + 4 * it) = v;
  }
  __syncthreads();

  const int bw = (g.spt == 1) ? b0 : b0 + wave;
  const bool sample_ok = bw < a.B;
  const int sidx = (g.spt == 1) ? 0 : wave;
  int nw;
  if (g.spt == 1) {
    nw = nvalid - 64 * wave;
    nw = nw < 0 ? 0 : (nw > 64 ? 64 : nw);
  } else {
    nw = sample_ok ? HW : 0;
  }
  const int q0 = (g.spt == 1) ? 64 * wave : 0;  // first pixel of the segment within the tile
  const size_t pix0 = (g.spt == 1) ? (size_t)b0 * HW + (size_t)row0 * W + q0 : (size_t)bw * HW;
  const int part = (g.spt == 1) ? (blockIdx.x - b0 * g.tps) * 4 + wave : 0;

  const int nh = (a.C0 % 64 == 0) ? 1 : 2;  // pixel-interleave factor of the two wave halves
  const int cw = 64 / nh;                   // channels per pass
  const int hh = (nh == 2) ? lane >> 5 : 0;
  for (int cg = 0; cg < a.C0 / cw; ++cg) {
    const int c = cg * cw + ((nh == 2) ? (lane & 31) : lane);
    float wr[CIN * 9];
#pragma unroll
    for (int i = 0; i < CIN * 9; ++i) wr[i] = a.w[(size_t)c * CIN * 9 + i];
    const float bias = a.bias[c];
    const float es = a.ep_scale ? a.ep_scale[c] : 1.f, eh = a.ep_scale ? a.ep_shift[c] : 0.f;
    // statistics in one pass around a pivot (the segment's first value): M2 = sum (v-p)^2 - n (mean-p)^2.
    // Keeping all 64 values for a two-pass form needs a fully unrolled loop, which hipcc turns into
    // 256 VGPRs + scratch (it hoists every LDS read), i.e. one wave per SIMD.
    float sum = 0.f, sq = 0.f, pivot = 0.f, vmx = 0.f;
    int r = (q0 + hh) / W, x = (q0 + hh) - r * W;  // incremental raster walk (no per-pixel division)
#pragma unroll 2
    for (int k = hh; k < nw; k += nh) {
      const float* base = smem + 4 * (sidx * per + r * WR + x);
      float acc = bias;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(base + 4 * (ky * WR + kx));
          acc += wr[ky * 3 + kx] * t.x;
          if (CIN > 1) acc += wr[9 + ky * 3 + kx] * t.y;
          if (CIN > 2) acc += wr[18 + ky * 3 + kx] * t.z;
        }
      if (a.ep_scale) acc = a.ep_nosilu ? acc * es + eh : silu_f(acc * es + eh);
      a.out[(pix0 + k) * a.C0 + c] = acc;
      vmx = fmaxf(vmx, fabsf(acc));
      if (k == hh) pivot = acc;
      const float d = acc - pivot;
      sum += d;
      sq += d * d;
      x += nh;
      if (x >= W) x -= W, ++r;
    }
    if (a.small_check && a.range_flag) {  // (ConvArgs::small_check; wave-uniform)
      float m = vmx;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (m > 0.f && m < 0.00390625f && lane == 0) atomicOr(a.range_flag, 2u);
    }
    if (a.stats_out && sample_ok && nw > 0) {
      float nme = (float)((nw - hh + nh - 1) / nh);      // pixels this lane saw
      const float dm = nme > 0.f ? sum / nme : 0.f;      // mean - pivot
      float mean = pivot + dm;
      float m2 = sq - nme * dm * dm;
      m2 = m2 < 0.f ? 0.f : m2;
      if (nh == 2) {  // Chan-combine the two halves (disjoint pixel sets of the same channel)
        const float no = __shfl_xor(nme, 32), mo = __shfl_xor(mean, 32), m2o = __shfl_xor(m2, 32);
        const float nt = nme + no;
        const float dl = mo - mean;
        m2 = m2 + m2o + dl * dl * nme * no / nt;
        mean = mean + dl * no / nt;
        if (hh) continue;  // lanes 0..31 store
      }
      float2 st;
      st.x = mean, st.y = m2;
      *reinterpret_cast<float2*>(a.stats_out + (((size_t)bw * g.nparts + part) * a.C0 + c) * 2) = st;
    }
  }
}

// Upsample.forward (unet_flexible.py:107-108) is conv3x3(nearest_x2(x)).  Output pixel 2i + a reads upsampled rows
// 2i + a - 1 .. 2i + a + 1, i.e. source rows {i - 1, i, i} (a = 0) or {i, i, i + 1} (a = 1): two source rows per parity class
// with the kernel rows that fall on the same source row ADDED -- a ConvTranspose2d(k = 4, s = 2, p = 1),
// out[o] = sum_i x[i] K[o + 1 - 2 i], with K[3] = w[0], K[1] = w[1] + w[2], K[2] = w[0] + w[1], K[0] = w[2] per axis.  The zero
// padding agrees (upsampled row -1 / 2H <-> source row -1 / H).  Sums in fp32, once, at create.
__global__ void up2_as_deconv_kernel(const float* w, float* k, int Cout, int Cin) {
  const size_t total = (size_t)Cout * Cin * 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kx = (int)(i & 3), ky = (int)((i >> 2) & 3);
    const size_t r = i >> 4;
    const int co = (int)(r % Cout), ci = (int)(r / Cout);
    const int lo_y = ky == 0 ? 2 : (ky == 1 ? 1 : 0), hi_y = ky == 0 ? 2 : (ky == 1 ? 2 : (ky == 2 ? 1 : 0));
    const int lo_x = kx == 0 ? 2 : (kx == 1 ? 1 : 0), hi_x = kx == 0 ? 2 : (kx == 1 ? 2 : (kx == 2 ? 1 : 0));
    const float* wp = w + ((size_t)co * Cin + ci) * 9;
    float acc = 0.f;
    for (int y = lo_y; y <= hi_y; ++y)
      for (int x = lo_x; x <= hi_x; ++x) acc += wp[y * 3 + x];
    k[i] = acc;
  }
}
void launch_up2_as_deconv(const float* w, float* k, int Cout, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(up2_as_deconv_kernel, dim3(256), dim3(256), 0, s, w, k, Cout, Cin);
}

void launch_conv_in(const ConvInArgs& a, int cin, hipStream_t s) {
  const int per = (a.g.th + 2) * (a.g.W + 2);
  dim3 grid(geom_num_tiles(a.g, a.B));
  const size_t lds4 = (size_t)a.g.spt * per * 4 * sizeof(float);
  if (a.C0 % 64 == 0 && !a.ep_scale) {  // (the U-Nets' 64-channel input convs: two channels per lane)
    if (cin == 1) hipLaunchKernelGGL(conv_in_pk_kernel<1>, grid, dim3(256), lds4, s, a);
    else hipLaunchKernelGGL(conv_in_pk_kernel<3>, grid, dim3(256), lds4, s, a);
    return;
  }
  if (cin == 1) hipLaunchKernelGGL(conv_in_kernel<1>, grid, dim3(256), lds4, s, a);
  else hipLaunchKernelGGL(conv_in_kernel<3>, grid, dim3(256), lds4, s, a);
}

// ------------------------------------------------------------------ conv_out
// out_conv(silu(out_norm(h))) (src/models/unet_flexible.py:257-259) with the Euler
// update x <- x + v*dt (src/sample_mnist_svhn.py:174-175) optionally fused.
// thread = output pixel; the transformed input chunk is staged in LDS exactly as
// in conv_mfma; the weights are wave-uniform (scalar loads).
// HBM-bound (reads [B,H,W,Cin] once): the halo items of a thread are decoded once per block, every chunk's fetches
// are unconditional (clamped) loads issued together, and the fetches of chunk c + 1 are in flight while chunk c is
// multiplied -- the first version fetched each item under a per-item branch inside stage_input's loop, which
// serialises the round trips (hipcc waits for a load under a branch before leaving the iteration).
template <int CIMG>
__global__ __launch_bounds__(256) void conv_out_kernel(const ConvOutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MAXI = 7;  // halo items (pixel, 4 channels) per thread: halo_px <= 448
  const int tid = threadIdx.x;
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
  const int HR = g.th + 2, WR = W + 2;
  int rows_valid = H - row0;
  if (rows_valid > g.th) rows_valid = g.th;
  const int nvalid = rows_valid * W;

  const int p = tid;
  int s, q;
  bool valid;
  if (g.spt == 1) {
    s = 0;
    valid = p < nvalid;
    q = valid ? p : 0;
  } else {
    s = p >> 6;
    valid = ((p & 63) < HW) && (b0 + s < a.B);
    q = (p & 63) < HW ? (p & 63) : 0;
  }
  const int r = q / W, x = q - r * W;
  const int abase = ((s * HR + r) * WR + x) * LDP;

  // per item: source offset (floats, clamped to 0 when outside), scale/shift offset, LDS offset, flags
  const int nitem = a.halo_px * 4;
  const int nit = (nitem + 255) >> 8;  // block-uniform
  unsigned poff[MAXI];  // (float index into the activation: B H W Cin < 2^32)
  int aoff[MAXI], ldst[MAXI];
  unsigned okm = 0u;
  {
    const int per = HR * WR;
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
      const int it = tid + 256 * j;
      poff[j] = 0, aoff[j] = 0, ldst[j] = -1;
      if (it < nitem) {
        const int hp = it >> 2, qq = it & 3;
        const int ss = hp / per;
        const int rem = hp - ss * per;
        const int hy = rem / WR, hx = rem - hy * WR;
        const int b = b0 + ss, y = row0 + hy - 1, xx = hx - 1;
        ldst[j] = hp * LDP + qq * 4;
        if ((y >= 0) && (y < H) && (xx >= 0) && (xx < W) && (b < a.B)) {
          okm |= 1u << j;
          poff[j] = (unsigned)((b * H + y) * W + xx) * (unsigned)a.Cin + (unsigned)(qq * 4);
          aoff[j] = (b * a.Cin + qq * 4) * 2;
        }
      }
    }
  }
  // scale/shift pairs: with one sample per tile (spt == 1: every 32x32 / 28x28 map) they depend on the thread's channel
  // quad only (it & 3 == tid & 3 for all its items) -- one fetch per chunk; several samples per tile: per item, at commit
  f32x4 ra[MAXI], re0, re1;
  const bool one_b = g.spt == 1;
  const int aoff1 = ((b0 < a.B ? b0 : 0) * a.Cin + (tid & 3) * 4) * 2;
  auto issue = [&](int ch) {
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
      if (j < nit) ra[j] = *reinterpret_cast<const f32x4*>(a.in + (size_t)poff[j] + ch * KC);
    if (a.ab && one_b) {
      const f32x4* pp = reinterpret_cast<const f32x4*>(a.ab + (size_t)aoff1 + ch * KC * 2);
      re0 = pp[0], re1 = pp[1];
    }
  };
  auto commit = [&](int ch) {
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
      if (j < nit) {
        f32x4 v = ra[j];
        if (a.ab) {
          f32x4 e0 = re0, e1 = re1;
          if (!one_b) {
            const f32x4* pp = reinterpret_cast<const f32x4*>(a.ab + (size_t)aoff[j] + ch * KC * 2);
            e0 = pp[0], e1 = pp[1];
          }
          v.x = silu_fast(e0.x * v.x + e0.y);
          v.y = silu_fast(e0.z * v.y + e0.w);
          v.z = silu_fast(e1.x * v.z + e1.y);
          v.w = silu_fast(e1.z * v.w + e1.w);
        }
        if (!((okm >> j) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ldst[j] >= 0) *reinterpret_cast<f32x4*>(smem + ldst[j]) = v;
      }
  };

  // Two partial sums per output channel (even / odd input channels of every chunk): the multiply-adds of a tap are
  // v_pk_fma_f32 -- two per instruction, the scalar weight pair as one SGPR-pair operand.  This kernel is VALU-bound
  // (CIMG = 3: 432 multiply-adds per pixel and chunk against 36 LDS reads), and unlike the conv kernels it has no MFMAs
  // of its own for the packed form to get in the way of (its workgroups do not share a SIMD with a conv's: those take
  // all of a CU's registers).
  typedef float co_f32x2 __attribute__((ext_vector_type(2)));
  co_f32x2 acc[CIMG];
#pragma unroll
  for (int co = 0; co < CIMG; ++co) acc[co] = co_f32x2{0.f, 0.f};

  const int nch = a.Cin / KC;
  issue(0);
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();
    commit(ch);
    __syncthreads();
    if (ch + 1 < nch) issue(ch + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const float* ap = smem + abase + (ky * WR + kx) * LDP;
      co_f32x2 v[KC / 2];
#pragma unroll
      for (int j = 0; j < KC / 4; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(ap + 4 * j);
        v[2 * j] = co_f32x2{t.x, t.y}, v[2 * j + 1] = co_f32x2{t.z, t.w};
      }
      // weights re-laid out as [chunk][tap][co][16] (launch_pack_conv_out): the 16 * CIMG scalars of a tap are
      // contiguous, so they arrive as wide scalar loads instead of 16 * CIMG strided s_load_dword
      const float* wt = a.w + ((size_t)(ch * 9 + tap) * CIMG) * KC;
#pragma unroll
      for (int co = 0; co < CIMG; ++co)
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk)
          acc[co] = __builtin_elementwise_fma(v[kk], *reinterpret_cast<const co_f32x2*>(wt + co * KC + 2 * kk), acc[co]);
    }
  }
  if (!valid) return;
  const int b = b0 + s;
  const int pixl = (g.spt == 1) ? row0 * W + p : (p & 63);
#pragma unroll
  for (int co = 0; co < CIMG; ++co) {
    const float v = (acc[co].x + acc[co].y) + a.bias[co];
    const size_t idx = ((size_t)b * CIMG + co) * HW + pixl;
    if (a.v_out) a.v_out[idx] = v;
    if (a.x_state) a.x_state[idx] = __fadd_rn(a.x_state[idx], __fmul_rn(v, a.dt));
  }
}

// out_conv weight [CIMG][Cin][3][3] -> [Cin/16][9][CIMG][16]
__global__ void pack_conv_out_kernel(const float* w, float* out, int cimg, int Cin) {
  const int total = cimg * Cin * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int kk = i % 16;
    int r = i / 16;
    const int co = r % cimg;
    r /= cimg;
    const int tap = r % 9, ch = r / 9;
    out[i] = w[((size_t)co * Cin + ch * 16 + kk) * 9 + tap];
  }
}

void launch_pack_conv_out(const float* w, float* out, int cimg, int Cin, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_out_kernel, dim3(16), dim3(256), 0, s, w, out, cimg, Cin);
}

void launch_conv_out(const ConvOutArgs& a, int cimg, hipStream_t s) {
  const size_t lds = (size_t)a.halo_px * LDP * sizeof(float);
  dim3 grid(geom_num_tiles(a.g, a.B));
  if (cimg == 1) hipLaunchKernelGGL(conv_out_kernel<1>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(conv_out_kernel<3>, grid, dim3(256), lds, s, a);
}

// ------------------------------------------------------------------ gn_finalize
// nn.GroupNorm statistics (src/models/unet_flexible.py:51,61,196): combine the
// producers' per-(segment, channel) (mean, M2) partials with Chan's parallel
// formula (fp64, fixed order => deterministic), then emit per-(sample, channel)
//   a = rstd * gamma,  b = beta - mean * a      so that  GN(x) = a*x + b.
// One workgroup per sample; thread = channel of the (possibly concatenated) input.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnFinalizeArgs a) {
  __shared__ double s_mean[512], s_m2[512];
  __shared__ float s_gm[32], s_rstd[32];
  const int b = blockIdx.x;
  const int C = a.C0 + a.C1;
  const TileGeom g = a.g;
  const int rep = a.rep > 1 ? a.rep : 1;
  const int nparts = g.nparts * rep;
  const double npix = (double)g.HW * (double)rep;  // pixels per channel of the normalised map
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* st;
    int cs, cc;
    if (c < a.C0) {
      st = a.stats0, cs = a.C0, cc = c;
    } else {
      st = a.stats1, cs = a.C1, cc = c - a.C0;
    }
    // division-free in the loops (an fp64 divide is ~40 instructions and the kernel sits between two convs):
    // mean = sum n_p mean_p / N,  M2 = sum [M2_p + n_p (mean_p - mean)^2]
    double s1 = 0.0;
    for (int p = 0; p < nparts; ++p) {
      const int np = geom_part_count(g, p % g.nparts);
      if (np == 0) continue;
      s1 += (double)np * (double)st[(((size_t)b * nparts + p) * cs + cc) * 2];
    }
    const double mean = s1 / npix;
    double m2 = 0.0;
    for (int p = 0; p < nparts; ++p) {
      const int np = geom_part_count(g, p % g.nparts);
      if (np == 0) continue;
      const float2 v = *reinterpret_cast<const float2*>(st + (((size_t)b * nparts + p) * cs + cc) * 2);
      const double d = (double)v.x - mean;
      m2 += (double)v.y + (double)np * d * d;
    }
    s_mean[c] = mean;
    s_m2[c] = m2;
  }
  __syncthreads();
  const int cpg = C / a.groups;
  if ((int)threadIdx.x < a.groups) {
    const int gi = threadIdx.x;
    double gm = 0.0;
    for (int j = 0; j < cpg; ++j) gm += s_mean[gi * cpg + j];
    gm /= (double)cpg;
    double m2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
      const double d = s_mean[gi * cpg + j] - gm;
      m2 += s_m2[gi * cpg + j] + d * d * npix;
    }
    const double var = m2 / ((double)cpg * npix);
    s_gm[gi] = (float)gm;
    s_rstd[gi] = (float)(1.0 / sqrt(var + 1e-5));
    if (a.mr) a.mr[((size_t)b * a.groups + gi) * 2] = s_gm[gi], a.mr[((size_t)b * a.groups + gi) * 2 + 1] = s_rstd[gi];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int gi = c / cpg;
    const float sc = s_rstd[gi] * a.gamma[c];
    float2 o;
    o.x = sc;
    o.y = a.beta[c] - s_gm[gi] * sc;
    *reinterpret_cast<float2*>(a.ab + ((size_t)b * C + c) * 2) = o;
  }
}

void launch_gn_finalize(const GnFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ time embedding
// timestep_embedding + time_embed MLP (src/models/unet_flexible.py:16-36, :148-152,
// :215-216) + every ResBlock's time_mlp = Linear(SiLU(t_emb)) (:55-58, :77).
// One workgroup per time value; each wave computes dot products with lane-strided
// (coalesced) weight reads and a shuffle reduction.  Inside the samplers t depends
// only on the step index, so the whole [steps][total] table is built once per call.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void time_embed_kernel(const TimeEmbedArgs a) {
  __shared__ float s_emb[256], s_h[1024], s_st[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x;
  float t;
  if (a.t_dev) t = a.t_dev[i];
  else t = (float)((double)(a.step_begin + i) * (1.0 / (double)a.num_steps));
  const int half = a.mc / 2;
  if (tid < half) {
    const float arg = t * a.freqs[tid];
    s_emb[tid] = cosf(arg);
    s_emb[half + tid] = sinf(arg);
    if (a.emb_out) a.emb_out[(size_t)i * a.mc + tid] = s_emb[tid], a.emb_out[(size_t)i * a.mc + half + tid] = s_emb[half + tid];
  }
  __syncthreads();
  const float* P = a.params;
  for (int o = wave; o < a.temb; o += 4) {
    float acc = 0.f;
    for (int k = lane; k < a.mc; k += 64) acc += P[a.te0w + (size_t)o * a.mc + k] * s_emb[k];
    acc = wave_sum(acc);
    if (lane == 0) s_h[o] = silu_f(acc + P[a.te0b + o]);
  }
  __syncthreads();
  for (int o = wave; o < a.temb; o += 4) {
    float acc = 0.f;
    for (int k = lane; k < a.temb; k += 64) acc += P[a.te2w + (size_t)o * a.temb + k] * s_h[k];
    acc = wave_sum(acc);
    if (lane == 0) s_st[o] = silu_f(acc + P[a.te2b + o]);  // SiLU of time_mlp
  }
  __syncthreads();
  for (int li = 0; li < a.nlin; ++li) {
    const TimeLinear L = a.lin[li];
    for (int o = wave; o < L.cout; o += 4) {
      float acc = 0.f;
      for (int k = lane; k < a.temb; k += 64) acc += P[L.w_off + (size_t)o * a.temb + k] * s_st[k];
      acc = wave_sum(acc);
      if (lane == 0) a.table[(size_t)i * a.total + L.out_off + o] = acc + P[L.b_off + o];
    }
  }
}

void launch_time_embed(const TimeEmbedArgs& a, int nt, hipStream_t s) {
  hipLaunchKernelGGL(time_embed_kernel, dim3(nt), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ weight packing
// [Cout][Cin][taps] (reference Conv2d layout) -> [Cout/nb][Cin/16][taps][nb][16], nb = 32*nt32
__global__ void pack_conv_kernel(const float* w, float* out, int Cout, int Cin, int taps, int nb) {
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
    const int blk = r / nch;
    const int co = blk * nb + n, ci = ch * 16 + kk;
    out[i] = w[((size_t)co * Cin + ci) * taps + tap];
  }
}

void launch_pack_conv(const float* w, float* out, int Cout, int Cin, int taps, int nt32, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(256), dim3(256), 0, s, w, out, Cout, Cin, taps, 32 * nt32);
}

// ConvTranspose2d(k=4, s=2, p=1) weight [Cin][Cout][4][4] (flow_matching.py:92,96) regrouped by output
// parity (py, px): output (2i+py, 2j+px) sums input (i+py+a-1, j+px+b-1) * w[.., 3-py-2a, 3-px-2b],
// a, b in {0,1}.  Layout [pc][Cout/nb][Cin/16][a*2+b][nb][16].
__global__ void pack_deconv_kernel(const float* w, float* out, int Cin, int Cout, int nb) {
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
    out[i] = w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
  }
}

void launch_pack_deconv(const float* w, float* out, int Cin, int Cout, int nt32, hipStream_t s) {
  hipLaunchKernelGGL(pack_deconv_kernel, dim3(256), dim3(256), 0, s, w, out, Cin, Cout, 32 * nt32);
}

__global__ void permute_cols_kernel(const float* w, float* out, int rows, int C, int P) {
  const size_t total = (size_t)rows * C * P;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const size_t r = i / C;
    const int p = r % P;
    const size_t o = r / P;
    out[i] = w[(o * C + c) * P + p];
  }
}

void launch_permute_cols(const float* w, float* out, int rows, int C, int P, hipStream_t s) {
  hipLaunchKernelGGL(permute_cols_kernel, dim3(512), dim3(256), 0, s, w, out, rows, C, P);
}

__global__ void permute_rows_kernel(const float* w, const float* b, float* wout, float* bout, int C, int P, int K) {
  const size_t total = (size_t)C * P * K;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = i % K;
    const size_t r = i / K;  // destination row p*C + c
    const int c = r % C;
    const int p = (int)(r / C);
    const size_t src = (size_t)c * P + p;
    wout[i] = w[src * K + k];
    if (k == 0) bout[r] = b[src];
  }
}

void launch_permute_rows(const float* w, const float* b, float* wout, float* bout, int C, int P, int K, hipStream_t s) {
  hipLaunchKernelGGL(permute_rows_kernel, dim3(512), dim3(256), 0, s, w, b, wout, bout, C, P, K);
}

// SinusoidalPositionEmbeddings.forward (flow_matching.py:22-31): out = [sin(t f_i), cos(t f_i)], f_i = exp(-i ln(1e4)/(half-1)).
// t comes from t_dev (t_count 1 or B) or, inside a sampler, from the step index exactly as flow_utils.py:92 builds it.
__global__ void fm_time_embed_kernel(const float* t_dev, int t_count, int num_steps, int step, const float* freqs,
                                     float* out, int B, int dim, int stride, int col0) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  float t;
  if (t_dev) t = t_dev[t_count == 1 ? 0 : b];
  else t = (float)((double)step * (1.0 / (double)num_steps));
  const float arg = t * freqs[k];
  out[(size_t)b * stride + col0 + k] = sinf(arg);
  out[(size_t)b * stride + col0 + half + k] = cosf(arg);
}

void launch_fm_time_embed(const float* t_dev, int t_count, int num_steps, int step, const float* freqs, float* out,
                          int B, int dim, int stride, int col0, hipStream_t s) {
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(fm_time_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t_dev, t_count, num_steps, step,
                     freqs, out, B, dim, stride, col0);
}

// ------------------------------------------------------------------ low side of a raw-staged map (ConvArgs::small_check)
// For a tensor that a two-plane fp16 conv stages RAW but that no conv produced (FlowMatchingModel: the fc1 output that
// deconv1 consumes, flow_matching.py:113-116): one wave per (sample, 32-channel block) takes the largest |value| over the
// sample's pixels and raises flag bit 1 exactly as a producing conv's epilogue would (hx_small_flag).
__global__ __launch_bounds__(256) void range_low_check_kernel(const float* in, int B, int HW, int C, unsigned* flag) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nblk = C >> 5;
  if (wave >= B * nblk) return;  // (wave-uniform)
  const int b = wave / nblk, cb = wave - b * nblk;
  const float* p = in + (size_t)b * HW * C + cb * 32 + (lane & 31);
  float m = 0.f;
  for (int px = lane >> 5; px < HW; px += 2) m = fmaxf(m, fabsf(p[(size_t)px * C]));
  hx_small_flag(flag, m);
}

void launch_range_low_check(const float* in, int B, int HW, int C, unsigned* flag, hipStream_t s) {
  const int waves = B * (C / 32);
  hipLaunchKernelGGL(range_low_check_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, in, B, HW, C, flag);
}

// ------------------------------------------------------------------ layout helper (parity hook)
__global__ void nhwc_to_nchw_kernel(const float* in, float* out, int B, int C, int HW) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = i % HW;
    const size_t r = i / HW;
    const int c = r % C;
    const size_t b = r / C;
    out[i] = in[(b * HW + p) * C + c];
  }
}

void launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(512), dim3(256), 0, s, in, out, B, C, HW);
}

}  // namespace rgfm
