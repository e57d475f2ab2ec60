// Microbenchmark: how many VALU / TRANS / DS-read instructions fit for free between
// v_mfma_f32_32x32x2_f32 issues of ONE wave (and of two waves) on a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KV, int KT, int KD>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, const float* in) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = in[threadIdx.x], b = in[threadIdx.x + 1];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  const float* lp = lds + (threadIdx.x & 63) * 4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < KV; ++j) v[j & 7] = v[j & 7] * 1.0001f + 0.5f;
#pragma unroll
      for (int j = 0; j < KT; ++j) v[j & 7] = __builtin_amdgcn_exp2f(v[j & 7]);
#pragma unroll
      for (int j = 0; j < KD; ++j) {
        float4 t = *reinterpret_cast<const float4*>(lp + ((m * 7 + j) & 15) * 256);
        v[j & 7] += t.x;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (iters * 16.0f);
}

template <int KV, int KT, int KD>
void run(int threads, const char* tag) {
  float *out, *in;
  hipMalloc(&out, ((1 << 20) + 16) * 4);
  hipMalloc(&in, 8192 * 4);
  hipMemset(in, 0, 8192 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  k<KV, KT, KD><<<256, threads>>>(out, 10, in);
  hipEventRecord(e0);
  k<KV, KT, KD><<<256, threads>>>(out, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float cyc;
  hipMemcpy(&cyc, out + (1 << 20), 4, hipMemcpyDeviceToHost);
  const double waves = threads / 64.0;
  const double tf = 256.0 * waves * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-10s valu=%2d trans=%2d dsread=%2d threads=%3d : %7.1f clk/MFMA(wave)  %6.1f TFLOP/s\n", tag, KV, KT, KD,
         threads, cyc, tf);
  hipFree(out);
  hipFree(in);
}

int main() {
  run<0, 0, 0>(256, "1w/SIMD");
  run<2, 0, 0>(256, "1w/SIMD");
  run<4, 0, 0>(256, "1w/SIMD");
  run<8, 0, 0>(256, "1w/SIMD");
  run<12, 0, 0>(256, "1w/SIMD");
  run<16, 0, 0>(256, "1w/SIMD");
  run<0, 2, 0>(256, "1w/SIMD");
  run<0, 4, 0>(256, "1w/SIMD");
  run<0, 8, 0>(256, "1w/SIMD");
  run<0, 0, 1>(256, "1w/SIMD");
  run<0, 0, 2>(256, "1w/SIMD");
  run<4, 2, 1>(256, "1w/SIMD");
  run<0, 0, 0>(512, "2w/SIMD");
  run<4, 0, 0>(512, "2w/SIMD");
  run<8, 0, 0>(512, "2w/SIMD");
  run<16, 0, 0>(512, "2w/SIMD");
  run<0, 4, 0>(512, "2w/SIMD");
  run<4, 2, 1>(512, "2w/SIMD");
  return 0;
}
