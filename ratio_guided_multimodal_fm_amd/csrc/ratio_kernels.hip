// ratio_kernels.hip -- the pieces of the density-ratio estimators that are not
// convolutions (src/models/ratio_flexible.py:185-385, src/models/ratio_estimator.py:34-191):
// 2x2 max-pool, global average pool, the dense projection heads (Linear layers on
// the fp32 MFMA), LayerNorm+SiLU and the log-ratio read-out.  Evaluated once per
// sampling call on the N_mc terminal Monte-Carlo samples.
#include "rgfm_device.h"

namespace rgfm {

// F.max_pool2d(., 2) (floor) on NHWC, with an optional GroupNorm-apply + SiLU
// (a*x+b from gn_finalize) in front for the GroupNorm encoder; SiLU by v_exp_f32 + v_rcp_f32 (~1 ulp each, as on the
// convs' load path and in grad_act_kernel's routing: the IEEE form made these kernels VALU-bound)
// (ratio_estimator.py:75-82: silu(gn(conv)) then pool).
__global__ void pool2_kernel(const float* in, const float* ab, float* out, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const size_t total = (size_t)B * Ho * Wo * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = i % C4;
    size_t r = i / C4;
    const int ox = r % Wo;
    r /= Wo;
    const int oy = r % Ho;
    const int b = r / Ho;
    f32x4 e0 = {1.f, 0.f, 1.f, 0.f}, e1 = e0;
    if (ab) {
      const f32x4* p = reinterpret_cast<const f32x4*>(ab + ((size_t)b * C + c4 * 4) * 2);
      e0 = p[0], e1 = p[1];
    }
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx) {
        f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)(b * H + 2 * oy + dy) * W + 2 * ox + dx) * C + c4 * 4);
        if (ab) {
          v.x = silu_fast(e0.x * v.x + e0.y);
          v.y = silu_fast(e0.z * v.y + e0.w);
          v.z = silu_fast(e1.x * v.z + e1.y);
          v.w = silu_fast(e1.z * v.w + e1.w);
        }
        m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
      }
    *reinterpret_cast<f32x4*>(out + ((size_t)(b * Ho + oy) * Wo + ox) * C + c4 * 4) = m;
  }
}

void launch_pool2(const float* in, const float* ab, float* out, int B, int H, int W, int C, hipStream_t s) {
  const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pool2_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, in, ab, out, B, H, W, C);
}

// nn.AdaptiveAvgPool2d(1) on NHWC -> [B][C], optional GroupNorm-apply + SiLU in front.
__global__ void avgpool_kernel(const float* in, const float* ab, float* out, int B, int HW, int C) {
  const size_t total = (size_t)B * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % C;
    const size_t b = i / C;
    float sa = 1.f, sb = 0.f;
    if (ab) {
      sa = ab[(b * C + c) * 2];
      sb = ab[(b * C + c) * 2 + 1];
    }
    float s = 0.f;
    for (int p = 0; p < HW; ++p) {
      float v = in[(b * HW + p) * C + c];
      if (ab) v = silu_fast(sa * v + sb);
      s += v;
    }
    out[i] = s / (float)HW;
  }
}

void launch_avgpool(const float* in, const float* ab, float* out, int B, int HW, int C, hipStream_t s) {
  const size_t total = (size_t)B * C;
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, ab, out, B, HW, C);
}

// nn.Linear on the fp32 matrix cores: y[rows][out] = x[rows][in] W[out][in]^T + b.
// Block tile 64 rows x 64 outputs, 4 waves of one 32x32 accumulator each; both
// operands are K-contiguous, staged 16 K-values at a time into [64][LDP] LDS rows
// and read as two b128 per lane (same k-permutation as conv_mfma).
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* x, const float* w, const float* bias,
                                                          float* y, int rows, int in, int out,
                                                          int x_stride, int y_stride) {
  __shared__ __attribute__((aligned(16))) float sA[64 * LDP];
  __shared__ __attribute__((aligned(16))) float sB[64 * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < in; k0 += KC) {
    __syncthreads();
    {
      const int row = tid >> 2, q = tid & 3;
      f32x4 va = {0.f, 0.f, 0.f, 0.f};
      if (r0 + row < rows) va = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + row) * x_stride + k0 + q * 4);
      *reinterpret_cast<f32x4*>(sA + row * LDP + q * 4) = va;
      const f32x4 vb = *reinterpret_cast<const f32x4*>(w + (size_t)(n0 + row) * in + k0 + q * 4);
      *reinterpret_cast<f32x4*>(sB + row * LDP + q * 4) = vb;
    }
    __syncthreads();
    const float* ap = sA + (wm * 32 + l31) * LDP + h * 8;
    const float* bp = sB + (wn * 32 + l31) * LDP + h * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
  }
  const int col = n0 + wn * 32 + l31;
  const float bv = bias[col];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < rows) y[(size_t)row * y_stride + col] = acc[r] + bv;
  }
}

void launch_linear_mfma(const float* x, const float* w, const float* b, float* y, int rows, int in,
                        int out, int x_stride, int y_stride, hipStream_t s) {
  hipLaunchKernelGGL(linear_mfma_kernel, dim3((rows + 63) / 64, out / 64), dim3(256), 0, s, x, w, b, y,
                     rows, in, out, x_stride, y_stride);
}

// Split-K Linear whose input rows are NHWC maps flattened per sample, with the pending
// GroupNorm scale/shift + SiLU applied on the load (ImageEncoder: fc(flatten(silu(gn4(.)))),
// flow_matching.py:69-71, with fc.weight re-indexed to NHWC order at create time).
// Grid (rows/64, out/64, splits); partial sums go to part[z][rows][out].
__global__ __launch_bounds__(256) void linear_splitk_kernel(const float* x, const float* ab, int C, const float* w,
                                                            float* part, int rows, int in, int out, int kper) {
  __shared__ __attribute__((aligned(16))) float sA[64 * LDP];
  __shared__ __attribute__((aligned(16))) float sB[64 * LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int kbeg = blockIdx.z * kper;
  int kend = kbeg + kper;
  if (kend > in) kend = in;
  const int row = tid >> 2, q = tid & 3;
  const bool rok = r0 + row < rows;
  const float* xr = x + (size_t)(rok ? r0 + row : 0) * in + q * 4;
  const float* abr = ab + (size_t)(rok ? r0 + row : 0) * C * 2;
  const float* wr = w + (size_t)(n0 + row) * in + q * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 va = *reinterpret_cast<const f32x4*>(xr + kbeg);
  f32x4 vb = *reinterpret_cast<const f32x4*>(wr + kbeg);
  int c0 = (kbeg + q * 4) % C;
  f32x4 e0 = *reinterpret_cast<const f32x4*>(abr + c0 * 2), e1 = *reinterpret_cast<const f32x4*>(abr + c0 * 2 + 4);
  for (int k0 = kbeg; k0 < kend; k0 += KC) {
    __syncthreads();
    {
      f32x4 v;
      v.x = silu_f(e0.x * va.x + e0.y);
      v.y = silu_f(e0.z * va.y + e0.w);
      v.z = silu_f(e1.x * va.z + e1.y);
      v.w = silu_f(e1.z * va.w + e1.w);
      if (!rok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(sA + row * LDP + q * 4) = v;
      *reinterpret_cast<f32x4*>(sB + row * LDP + q * 4) = vb;
    }
    __syncthreads();
    const int kn = k0 + KC < kend ? k0 + KC : kbeg;  // prefetch the next chunk under the MFMAs
    va = *reinterpret_cast<const f32x4*>(xr + kn);
    vb = *reinterpret_cast<const f32x4*>(wr + kn);
    c0 = (kn + q * 4) % C;
    e0 = *reinterpret_cast<const f32x4*>(abr + c0 * 2), e1 = *reinterpret_cast<const f32x4*>(abr + c0 * 2 + 4);
    const float* ap = sA + (wm * 32 + l31) * LDP + h * 8;
    const float* bp = sB + (wn * 32 + l31) * LDP + h * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
  }
  const int col = n0 + wn * 32 + l31;
  float* dst = part + (size_t)blockIdx.z * rows * out;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = r0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (orow < rows) dst[(size_t)orow * out + col] = acc[r];
  }
}

__global__ void splitk_reduce_kernel(const float* part, const float* bias, float* y, int splits, int rows, int out,
                                     int y_stride) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * out) return;
  const int col = i % out;
  const size_t row = i / out;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += part[(size_t)z * rows * out + i];
  y[row * y_stride + col] = s + bias[col];
}

void launch_linear_mfma_splitk(const float* x, const float* ab, int C, const float* w, const float* b, float* y,
                               float* part, int splits, int rows, int in, int out, int y_stride, hipStream_t s) {
  const int kper = ((in / KC + splits - 1) / splits) * KC;
  hipLaunchKernelGGL(linear_splitk_kernel, dim3((rows + 63) / 64, out / 64, splits), dim3(256), 0, s, x, ab, C, w, part,
                     rows, in, out, kper);
  const size_t n = (size_t)rows * out;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, b, y, splits, rows,
                     out, y_stride);
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// nn.LayerNorm(n) (eps 1e-5, biased variance) followed by SiLU, in place; wave per row.
__global__ __launch_bounds__(256) void layernorm_silu_kernel(float* x, const float* w, const float* b, int rows, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = x + (size_t)row * n;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += p[i];
  const float mean = wsum(s) / (float)n;
  float m2 = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float d = p[i] - mean;
    m2 += d * d;
  }
  const float rstd = 1.0f / sqrtf(wsum(m2) / (float)n + 1e-5f);
  for (int i = lane; i < n; i += 64) p[i] = silu_f((p[i] - mean) * rstd * w[i] + b[i]);
}

void launch_layernorm_silu(float* x, const float* w, const float* b, int rows, int n, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_silu_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, rows, n);
}

__device__ __forceinline__ float logsigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// final Linear(n -> 1) + log_ratio (ratio_flexible.py:366-385) / exp (sample_mnist_svhn.py:111)
__global__ __launch_bounds__(256) void ratio_head_kernel(const float* x, const float* w, const float* b, float* out,
                                                        int rows, int n, int loss, int what) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += x[(size_t)row * n + i] * w[i];
  s = wsum(s) + b[0];
  if (lane != 0) return;
  float r = s;
  if (what != 0) {
    if (loss == 0) {
      r = logsigmoid_f(s) - logsigmoid_f(-s);
    } else {
      const float sp = s > 20.f ? s : log1pf(expf(s));
      r = logf(sp + 1e-8f);
    }
    if (what == 2) r = expf(r);
  }
  out[row] = r;
}

void launch_ratio_head(const float* x, const float* w, const float* b, float* out, int rows, int n,
                       int loss, int what, hipStream_t s) {
  hipLaunchKernelGGL(ratio_head_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, out, rows, n, loss, what);
}

// eval-mode nn.BatchNorm2d as per-channel scale/shift: y = x*scale + shift
__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                               float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = w[c] / sqrtf(rv[c] + 1e-5f);
  scale[c] = sc;
  shift[c] = b[c] - rm[c] * sc;
}

void launch_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                    float* shift, int C, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, s, w, b, rm, rv, scale, shift, C);
}

}  // namespace rgfm
