// Does a VALU-only wave overlap with a bf16-MFMA-only wave (v_mfma_f32_32x32x16_bf16) on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run VALU fma chains (8 independent)
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, const float* in, int mode) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      bf16x8 av, bv; for (int i = 0; i < 8; ++i) { av[i] = (__bf16)in[threadIdx.x + i]; bv[i] = (__bf16)in[threadIdx.x + 8 + i]; }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m & 3], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
  } else {
    if (mode & 2) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16 * 16; ++m) v[m & 7] = v[m & 7] * 1.0001f + 0.5f;  // 256 fma = 1024 issue cycles
      }
      for (int i = 0; i < 8; ++i) s += v[i];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *out, *in;
  hipMalloc(&out, (1 << 20) * 4);
  hipMalloc(&in, 8192 * 4);
  hipMemset(in, 0, 8192 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    k<<<256, 512>>>(out, 10, in, mode);
    hipEventRecord(e0);
    k<<<256, 512>>>(out, iters, in, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s%s): %.3f ms  -> %.1f cycles@2.4GHz per iteration (16 bf16 MFMA = 512 cyc; 256 FMA = 1024 cyc)\n", mode,
           (mode & 1) ? "MFMA waves " : "", (mode & 2) ? "VALU waves" : "", ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
