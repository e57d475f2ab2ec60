// ubench.hip -- measured ceilings for bench.py's roofline line (bench support, not on the product path):
// the sustained f16 MFMA rate of the device under DVFS and its HBM copy bandwidth.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/rgfm.h"

namespace {

typedef _Float16 ub_f16x8 __attribute__((ext_vector_type(8)));
typedef float ub_f32x16 __attribute__((ext_vector_type(16)));
typedef float ub_f32x4 __attribute__((ext_vector_type(4)));

// 512 threads = 2 waves per SIMD; each wave keeps four independent 32x32 accumulators and issues 12 MFMAs per
// iteration (the conv kernel's tile: 2 x 2 tiles x 3 products) on pseudo-random fp16 operands held in registers
__global__ __launch_bounds__(512, 2) void ub_mfma_kernel(float* sink, int iters) {
  unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  ub_f16x8 a[2][2], b[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s = s * 1664525u + 1013904223u;
        a[i][p][k] = (_Float16)(((float)(s >> 8) / 8388608.0f - 1.0f) * (p ? 0.001f : 1.0f));
        s = s * 1664525u + 1013904223u;
        b[i][p][k] = (_Float16)(((float)(s >> 8) / 8388608.0f - 1.0f) * (p ? 0.001f : 1.0f));
      }
  ub_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][q == 0], b[j][q == 1], acc[i][j], 0, 0, 0);
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[i][j][r];
  if (t == 123.456f) sink[0] = t;  // (keeps the accumulators live)
}

// float4 copy with eight 16-byte loads in flight per thread before the first store (the one-load-per-trip loop of
// round 2 read 4.9 TB/s where the guide measures 6.29 TB/s for a float4 copy on this part: too few bytes in flight);
// NT: non-temporal loads and stores (streamed once)
template <bool NT>
__global__ __launch_bounds__(256) void ub_copy_kernel(const ub_f32x4* __restrict__ in, ub_f32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; base < n; base += stride) {
    ub_f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = base + (size_t)k * 256;
      if (i < n) v[k] = NT ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = base + (size_t)k * 256;
      if (i < n) {
        if (NT) __builtin_nontemporal_store(v[k], out + i);
        else out[i] = v[k];
      }
    }
  }
}

}  // namespace

#define UB_TRY(expr)                      \
  do {                                    \
    if ((expr) != hipSuccess) return RGFM_EHIP; \
  } while (0)

extern "C" int rgfm_ubench_mfma_f16(double* tflops) {
  if (!tflops) return RGFM_EINVAL;
  int dev = 0;
  hipDeviceProp_t p;
  UB_TRY(hipGetDevice(&dev));
  UB_TRY(hipGetDeviceProperties(&p, dev));
  const int blocks = p.multiProcessorCount;
  float* sink = nullptr;
  UB_TRY(hipMalloc(&sink, 256));
  hipEvent_t e0, e1;
  UB_TRY(hipEventCreate(&e0));
  UB_TRY(hipEventCreate(&e1));
  const int iters = 20000;  // 12 MFMAs x 2 waves per SIMD x 32 cycles = 15 M cycles ~ 8 ms per launch
  hipLaunchKernelGGL(ub_mfma_kernel, dim3(blocks), dim3(512), 0, 0, sink, iters);  // warm-up
  UB_TRY(hipDeviceSynchronize());
  const int reps = 30;  // ~0.25 s back to back: the clock settles where it sits under the conv kernels
  UB_TRY(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(ub_mfma_kernel, dim3(blocks), dim3(512), 0, 0, sink, iters);
  UB_TRY(hipEventRecord(e1, 0));
  UB_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  UB_TRY(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)reps * blocks * 8.0 * iters * 12.0 * (2.0 * 32 * 32 * 16);
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return RGFM_OK;
}

extern "C" int rgfm_ubench_hbm_copy(size_t bytes, double* gbps) {
  if (!gbps || bytes < (1u << 20)) return RGFM_EINVAL;
  const size_t n = bytes / 16;
  ub_f32x4 *in = nullptr, *out = nullptr;
  if (hipMalloc(&in, n * 16) != hipSuccess) return RGFM_ENOMEM;
  if (hipMalloc(&out, n * 16) != hipSuccess) {
    (void)hipFree(in);
    return RGFM_ENOMEM;
  }
  UB_TRY(hipMemset(in, 1, n * 16));
  hipEvent_t e0, e1;
  UB_TRY(hipEventCreate(&e0));
  UB_TRY(hipEventCreate(&e1));
  int dev = 0;
  hipDeviceProp_t p;
  UB_TRY(hipGetDevice(&dev));
  UB_TRY(hipGetDeviceProperties(&p, dev));
  const dim3 grid(p.multiProcessorCount * 8);  // eight 256-thread workgroups per CU, grid-stride
  double best = 0.0;
  for (int nt = 0; nt < 2; ++nt) {  // default cache policy and non-temporal: the better of the two is the ceiling
    auto launch = [&]() {
      if (nt) hipLaunchKernelGGL(ub_copy_kernel<true>, grid, dim3(256), 0, 0, in, out, n);
      else hipLaunchKernelGGL(ub_copy_kernel<false>, grid, dim3(256), 0, 0, in, out, n);
    };
    launch();
    UB_TRY(hipDeviceSynchronize());
    const int reps = 10;
    UB_TRY(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    UB_TRY(hipEventRecord(e1, 0));
    UB_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    UB_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double g = 2.0 * (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
    best = g > best ? g : best;
  }
  *gbps = best;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(in);
  (void)hipFree(out);
  return RGFM_OK;
}
