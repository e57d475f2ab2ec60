// ratio_grad.hip -- reverse pass of RatioEstimatorMNISTSVHN (src/models/ratio_flexible.py:185-385) for the
// gradient log-ratio guidance v + gamma * grad log r(x_t, y_t) (reference README.md:159-164).
// The 3x3 data gradients run on the fp32 MFMA conv (conv_mfma.hip) with the weights transposed and the taps
// flipped, the dense ones on linear_mfma; this file holds what is left: SiLU' / BatchNorm-scale / max-pool and
// average-pool routing, the image-side conv gradient, LayerNorm+SiLU backward, the head, and the Euler update.
#include "rgfm_device.h"

namespace rgfm {

__device__ __forceinline__ float sigmoid_g(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float dsilu_g(float v) {
  const float s = sigmoid_g(v);
  return s * (1.0f + v * (1.0f - s));
}

__global__ void conv_weight_transpose_kernel(const float* w, float* wt, int Co, int Ci) {
  const size_t total = (size_t)Co * Ci * 9;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = i % 9;
    const size_t r = i / 9;
    const int co = r % Co, ci = r / Co;  // i indexes wt[ci][co][k]
    wt[i] = w[((size_t)co * Ci + ci) * 9 + (8 - k)];
  }
}
void launch_conv_weight_transpose(const float* w, float* wt, int Co, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(conv_weight_transpose_kernel, dim3(256), dim3(256), 0, s, w, wt, Co, Ci);
}

__global__ void transpose2d_kernel(const float* w, float* wt, int rows, int cols) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = i % rows;
    const size_t c = i / rows;  // i indexes wt[c][r]
    wt[i] = w[(size_t)r * cols + c];
  }
}
void launch_transpose2d(const float* w, float* wt, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose2d_kernel, dim3(256), dim3(256), 0, s, w, wt, rows, cols);
}

__global__ void fill_kernel(float* p, float v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill(float* p, float v, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(fill_kernel, dim3(n < 65536 ? (unsigned)((n + 255) / 256) : 256u), dim3(256), 0, s, p, v, n);
}
__global__ void fill_ab_kernel(float* ab, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ab[2 * i] = 1.f, ab[2 * i + 1] = 0.f;
}
void launch_fill_ab_identity(float* ab, size_t n_pairs, hipStream_t s) {
  hipLaunchKernelGGL(fill_ab_kernel, dim3(256), dim3(256), 0, s, ab, n_pairs);
}

// sigmoid by v_exp_f32 + v_rcp_f32 (~1 ulp each; the IEEE expf + division of the first version made this kernel
// VALU-bound at 1.2 TB/s: ~50 instructions per sigmoid, 32 sigmoids per thread in the pooled modes); silu and its
// derivative share it: silu(z) = z s, silu'(z) = s (1 + z (1 - s))
__device__ __forceinline__ float sigmoid_fast(float v) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f));
}

// one thread per (sample, output-side pixel group, 4 channels).  GN (the GroupNorm encoders of RatioEstimator,
// ratio_estimator.py:75-88): z is the conv output, the activation's argument is u = a z + b with the sample's
// scale/shift pairs ab [B][C][2] (gn_finalize), and the result is d/du -- the norm's own backward follows
// (gn_bwd_kernel); otherwise z is the folded-BatchNorm output and `scale` its per-channel factor.
template <bool GN>
__global__ __launch_bounds__(256) void grad_act_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                       const float* __restrict__ scale, float* __restrict__ gz, int B, int S,
                                                       int C, int mode, unsigned* __restrict__ amax) {
  const int C4 = C / 4;
  float tmax = 0.f;  // largest |output| this thread wrote (ConvArgs::in_amax of the conv that reads gz)
  // u = ea z + eb per element (GN), scale factor of the result
  auto norm_of = [&](int b, int c4, f32x4& ea, f32x4& eb, f32x4& sc) {
    if (GN) {
      const f32x4* p = reinterpret_cast<const f32x4*>(scale + ((size_t)b * C + c4 * 4) * 2);
      const f32x4 e0 = p[0], e1 = p[1];
      ea = f32x4{e0.x, e0.z, e1.x, e1.z}, eb = f32x4{e0.y, e0.w, e1.y, e1.w}, sc = f32x4{1.f, 1.f, 1.f, 1.f};
    } else {
      ea = f32x4{1.f, 1.f, 1.f, 1.f}, eb = f32x4{0.f, 0.f, 0.f, 0.f};
      sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
    }
  };
  if (mode == 1 || mode == 3) {
    const int So = S / 2;
    const float inv_o = 1.0f / (float)(So * So);
    const size_t total = (size_t)B * So * So * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int c4 = i % C4;
      size_t r = i / C4;
      const int ox = r % So;
      r /= So;
      const int oy = r % So;
      const int b = r / So;
      f32x4 gv;
      if (mode == 3) gv = *reinterpret_cast<const f32x4*>(g + (size_t)b * C + c4 * 4) * inv_o;
      else gv = *reinterpret_cast<const f32x4*>(g + ((size_t)(b * So + oy) * So + ox) * C + c4 * 4);
      f32x4 ea, eb, sc;
      norm_of(b, c4, ea, eb, sc);
      f32x4 zv[4];
      size_t off[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        off[k] = ((size_t)(b * S + 2 * oy + (k >> 1)) * S + 2 * ox + (k & 1)) * C + c4 * 4;
        zv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + off[k]));
        if (GN) zv[k] = zv[k] * ea + eb;
      }
      f32x4 o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // F.max_pool2d backward: the first maximum of silu(z) in scan order (ky, kx), strict '>'
        float sg[4], sv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[k] = sigmoid_fast(zv[k][e]), sv[k] = zv[k][e] * sg[k];
        int best = 0;
        float bv = sv[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (sv[k] > bv) bv = sv[k], best = k;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == best) o[k][e] = gv[e] * (sg[k] * (1.0f + zv[k][e] * (1.0f - sg[k]))) * sc[e];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __builtin_nontemporal_store(o[k], reinterpret_cast<f32x4*>(gz + off[k]));
        tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(o[k].x), fabsf(o[k].y))), fmaxf(fabsf(o[k].z), fabsf(o[k].w)));
      }
    }
  } else {
    const size_t total = (size_t)B * S * S * C4;
    const float inv = 1.0f / (float)(S * S);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int c4 = i % C4;
      const size_t px = i / C4;
      const size_t b = px / ((size_t)S * S);
      f32x4 zv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + px * C + c4 * 4));
      f32x4 ea, eb, sc;
      norm_of((int)b, c4, ea, eb, sc);
      if (GN) zv = zv * ea + eb;
      f32x4 gv;
      if (mode == 2) {
        gv = *reinterpret_cast<const f32x4*>(g + b * C + c4 * 4);
        gv = gv * inv;
      } else {
        gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g + px * C + c4 * 4));
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = sigmoid_fast(zv[e]);
        o[e] = gv[e] * (sg * (1.0f + zv[e] * (1.0f - sg))) * sc[e];
      }
      *reinterpret_cast<f32x4*>(gz + px * C + c4 * 4) = o;
      tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
  }
  if (amax) {  // (kernel-uniform) non-negative floats order like their bit patterns
    // One atomic per WORKGROUP, and only when it would raise the word as this workgroup last saw it (a plain, possibly
    // stale read: staleness costs an atomic, never a wrong maximum) -- one atomic per wave on one address serialised
    // 131 k of them at the memory side and made this kernel 5 x slower.
    __shared__ float smax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = tmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
      if (m > __uint_as_float(__hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) atomicMax(amax, __float_as_uint(m));
    }
  }
}
void launch_grad_act(const float* g, const float* z, const float* scale, float* gz, int B, int S, int C, int mode, hipStream_t s,
                     unsigned* amax) {
  const size_t total = (size_t)B * ((mode == 1 || mode == 3) ? (S / 2) * (S / 2) : S * S) * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(grad_act_kernel<false>, dim3(blocks ? blocks : 1), dim3(256), 0, s, g, z, scale, gz, B, S, C, mode, amax);
}
// GroupNorm encoders: ab = the samples' scale/shift pairs; an odd map's last row and column are outside every 2x2
// window (F.max_pool2d floors) and get a zero gradient
void launch_grad_act_gn(const float* g, const float* z, const float* ab, float* gu, int B, int S, int C, int mode, hipStream_t s) {
  const bool pooled = mode == 1 || mode == 3;
  if (pooled && (S & 1)) (void)hipMemsetAsync(gu, 0, (size_t)B * S * S * C * sizeof(float), s);
  const size_t total = (size_t)B * (pooled ? (S / 2) * (S / 2) : S * S) * (C / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(grad_act_kernel<true>, dim3(blocks ? blocks : 1), dim3(256), 0, s, g, z, ab, gu, B, S, C, mode, (unsigned*)nullptr);
}

// nn.GroupNorm backward, in place on gu (d/du, u = gamma xhat + beta) -> d/dz, one workgroup per (sample, group):
//   d xhat = d u gamma,  d z = rstd (d xhat - mean(d xhat) - xhat mean(d xhat xhat)),  xhat = (z - mean) rstd
// with (mean, rstd) of the group from gn_finalize (GnFinalizeArgs::mr).  A group is cpg channels x S x S values
// (3136 at most in the 28x28 encoders): two passes over data that stays in L2.
__global__ __launch_bounds__(256) void gn_bwd_kernel(float* __restrict__ gu, const float* __restrict__ z,
                                                     const float* __restrict__ gamma, const float* __restrict__ mr, int HW, int C,
                                                     int groups) {
  __shared__ double red[2][256];
  const int b = blockIdx.x / groups, gi = blockIdx.x - b * groups;
  const int cpg = C / groups, q4 = cpg / 4;  // (cpg is a multiple of 4)
  const float mean = mr[((size_t)b * groups + gi) * 2], rstd = mr[((size_t)b * groups + gi) * 2 + 1];
  const int items = HW * q4;
  double s1 = 0.0, s2 = 0.0;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int p = it / q4, q = it - p * q4;
    const size_t o = ((size_t)b * HW + p) * C + gi * cpg + q * 4;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gu + o), zv = *reinterpret_cast<const f32x4*>(z + o);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + gi * cpg + q * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gx = gv[e] * gm[e], xh = (zv[e] - mean) * rstd;
      s1 += (double)gx, s2 += (double)gx * (double)xh;
    }
  }
  red[0][threadIdx.x] = s1, red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[0][threadIdx.x] += red[0][threadIdx.x + o], red[1][threadIdx.x] += red[1][threadIdx.x + o];
    __syncthreads();
  }
  const double n = (double)HW * (double)cpg;
  const float m1 = (float)(red[0][0] / n), m2 = (float)(red[1][0] / n);
  for (int it = threadIdx.x; it < items; it += 256) {
    const int p = it / q4, q = it - p * q4;
    const size_t o = ((size_t)b * HW + p) * C + gi * cpg + q * 4;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gu + o), zv = *reinterpret_cast<const f32x4*>(z + o);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + gi * cpg + q * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (zv[e] - mean) * rstd;
      r[e] = rstd * (gv[e] * gm[e] - m1 - xh * m2);
    }
    *reinterpret_cast<f32x4*>(gu + o) = r;
  }
}
void launch_gn_bwd(float* gu, const float* z, const float* gamma, const float* mr, int B, int HW, int C, int groups, hipStream_t s) {
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * groups), dim3(256), 0, s, gu, z, gamma, mr, HW, C, groups);
}

// gimg[b][c][y][x] = sum_{co, ky, kx} w[co][c][ky][kx] gz[b][y - ky + 1][x - kx + 1][co].  One thread per output
// pixel (all CIMG image channels); the weights sit in LDS as [tap][co][CIMG] so that a tap's four output channels are
// wave-uniform broadcast reads; gz is read as 16-byte pieces of a neighbour pixel's channel vector (the nine
// neighbours of adjacent pixels overlap: L1 / L2 serve them).  The first version (one WAVE per pixel, lanes over co,
// a wave reduction per image channel) ran at 2 TFLOP/s: 0.89 ms for the SVHN encoder at B = 512.
template <int CIMG>
__global__ __launch_bounds__(256) void conv_bwd_img_kernel(const float* __restrict__ gz, const float* __restrict__ w,
                                                           float* __restrict__ gimg, int B, int S, int Co) {
  extern __shared__ __attribute__((aligned(16))) float swt[];  // [9][Co][CIMG padded to 4]
  for (int i = threadIdx.x; i < 9 * Co * 4; i += 256) {
    const int c = i & 3, co = (i >> 2) % Co, k = (i >> 2) / Co;
    swt[i] = c < CIMG ? w[((size_t)co * CIMG + c) * 9 + k] : 0.f;
  }
  __syncthreads();
  const size_t px = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (px >= (size_t)B * S * S) return;
  const int x = px % S;
  const int y = (px / S) % S;
  const size_t b = px / ((size_t)S * S);
  float acc[CIMG];
#pragma unroll
  for (int c = 0; c < CIMG; ++c) acc[c] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yo = y - ky + 1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xo = x - kx + 1;
      if (yo < 0 || yo >= S || xo < 0 || xo >= S) continue;
      const f32x4* gp = reinterpret_cast<const f32x4*>(gz + ((b * S + yo) * S + xo) * Co);
      const f32x4* wp = reinterpret_cast<const f32x4*>(swt + (size_t)(ky * 3 + kx) * Co * 4);
#pragma unroll 4
      for (int c4 = 0; c4 < Co / 4; ++c4) {
        const f32x4 gv = gp[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 wv = wp[c4 * 4 + e];
#pragma unroll
          for (int c = 0; c < CIMG; ++c) acc[c] = fmaf(wv[c], gv[e], acc[c]);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CIMG; ++c) gimg[((b * CIMG + c) * S + y) * S + x] = acc[c];
}
void launch_conv_bwd_img(const float* gz, const float* w, float* gimg, int B, int S, int Co, int cimg, hipStream_t s) {
  const unsigned blocks = (unsigned)(((size_t)B * S * S + 255) / 256);
  const size_t lds = (size_t)9 * Co * 4 * sizeof(float);
  if (cimg == 1) hipLaunchKernelGGL(conv_bwd_img_kernel<1>, dim3(blocks), dim3(256), lds, s, gz, w, gimg, B, S, Co);
  else hipLaunchKernelGGL(conv_bwd_img_kernel<3>, dim3(blocks), dim3(256), lds, s, gz, w, gimg, B, S, Co);
}

__device__ __forceinline__ float wsum_g(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// y = silu(v), v = w * uhat + b, uhat = (u - mean) * rstd:  gu = rstd * (guh - mean(guh) - uhat * mean(guh * uhat)),
// guh = gy * silu'(v) * w.  Wave per row (statistics recomputed exactly as the forward kernel does).
__global__ __launch_bounds__(256) void layernorm_silu_bwd_kernel(const float* u, const float* gy, const float* w, const float* b,
                                                                 float* gu, int rows, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = u + (size_t)row * n;
  const float* q = gy + (size_t)row * n;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += p[i];
  const float mean = wsum_g(s) / (float)n;
  float m2 = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float d = p[i] - mean;
    m2 += d * d;
  }
  const float rstd = 1.0f / sqrtf(wsum_g(m2) / (float)n + 1e-5f);
  float a1 = 0.f, a2 = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float uh = (p[i] - mean) * rstd;
    const float guh = q[i] * dsilu_g(uh * w[i] + b[i]) * w[i];
    a1 += guh, a2 += guh * uh;
  }
  a1 = wsum_g(a1) / (float)n, a2 = wsum_g(a2) / (float)n;
  for (int i = lane; i < n; i += 64) {
    const float uh = (p[i] - mean) * rstd;
    const float guh = q[i] * dsilu_g(uh * w[i] + b[i]) * w[i];
    gu[(size_t)row * n + i] = rstd * (guh - a1 - uh * a2);
  }
}
void launch_layernorm_silu_bwd(const float* u, const float* gy, const float* w, const float* b, float* gu, int rows, int n,
                               hipStream_t s) {
  hipLaunchKernelGGL(layernorm_silu_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, u, gy, w, b, gu, rows, n);
}

__device__ __forceinline__ float logsigmoid_g(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// disc: log_ratio = logsigmoid(s) - logsigmoid(-s), d/ds = sigmoid(-s) + sigmoid(s); rulsif: log(softplus(s) + 1e-8),
// d/ds = sigmoid(s) / (softplus(s) + 1e-8) (softplus threshold 20, ratio_flexible.py:379-383)
__global__ void ratio_head_bwd_kernel(const float* score, const float* w, float* gh, float* log_ratio, int rows, int n, int loss) {
  const size_t total = (size_t)rows * n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = i % n;
    const size_t row = i / n;
    const float s = score[row];
    float ds, lr;
    if (loss == 0) {
      ds = sigmoid_g(-s) + sigmoid_g(s);
      lr = logsigmoid_g(s) - logsigmoid_g(-s);
    } else {
      const float sp = s > 20.f ? s : log1pf(expf(s));
      ds = (s > 20.f ? 1.f : sigmoid_g(s)) / (sp + 1e-8f);
      lr = logf(sp + 1e-8f);
    }
    gh[i] = ds * w[k];
    if (log_ratio && k == 0) log_ratio[row] = lr;
  }
}
void launch_ratio_head_bwd(const float* score, const float* w, float* gh, float* log_ratio, int rows, int n, int loss,
                           hipStream_t s) {
  const size_t total = (size_t)rows * n;
  hipLaunchKernelGGL(ratio_head_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, score, w, gh, log_ratio,
                     rows, n, loss);
}

__global__ void euler_grad_kernel(float* x, const float* v, const float* g, size_t n, float gamma, float dt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = __fadd_rn(x[i], __fmul_rn(fmaf(gamma, g[i], v[i]), dt));
}
void launch_euler_grad(float* x, const float* v, const float* g, size_t n, float gamma, float dt, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(euler_grad_kernel, dim3(blocks), dim3(256), 0, s, x, v, g, n, gamma, dt);
}

}  // namespace rgfm
