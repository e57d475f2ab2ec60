// Does VALU work overlap with f16 MFMAs (v_mfma_f32_32x32x16_f16) on one SIMD of gfx950 -- in the SAME wave's stream
// (F independent v_fma_f32 between consecutive MFMAs) and from OTHER waves of the SIMD (wave roles)?
//   mfma_valu_overlap           prints cycles per MFMA for every arrangement
// Arrangements: W waves per SIMD (1, 2, 4); each wave runs either a mixed stream (1 MFMA + F fma) or is a pure MFMA /
// pure VALU wave.  Cycles from s_memtime over the loop, median over workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F>
__device__ __forceinline__ void mixed(f32x16 (&acc)[2], const f16x8& a, const f16x8& b, float (&v)[8], int iters) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 1], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) v[(m * F + f) & 7] = __builtin_fmaf(v[(m * F + f) & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// role of a wave: 0 mixed stream, 1 MFMA only, 2 VALU only (the same number of fma as a mixed wave would issue), 3 idle
template <int F>
__global__ __launch_bounds__(1024, 1) void k(float* out, long long* cyc, int iters, const float* in, int nwaves_simd, int roles) {
  const int wave = threadIdx.x >> 6;           // waves are dealt to SIMDs round-robin: wave w sits on SIMD w % 4
  const int slot = wave >> 2;                  // 0 .. nwaves_simd-1: which of the SIMD's waves
  const int role = (roles >> (2 * slot)) & 3;
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)in[threadIdx.x + i], b[i] = (_Float16)in[threadIdx.x + 8 + i];
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + i];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 0) mixed<F>(acc, a, b, v, iters);
  else if (role == 1) mixed<0>(acc, a, b, v, iters);
  else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8 * (F ? F : 1); ++m) v[m & 7] = __builtin_fmaf(v[m & 7], 1.0001f, 0.5f);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int F>
static void run(const char* what, int nws, int roles, float* out, long long* cyc, const float* in) {
  const int iters = 2000, blocks = 256;
  hipMemset(cyc, 0, blocks * 16 * 8);
  hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256 * nws), 0, 0, out, cyc, 50, in, nws, roles);
  hipLaunchKernelGGL(k<F>, dim3(blocks), dim3(256 * nws), 0, 0, out, cyc, iters, in, nws, roles);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 16);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  printf("%-58s F=%d:", what, F);
  for (int slot = 0; slot < nws; ++slot) {
    std::vector<long long> v;
    for (int b = 0; b < blocks; ++b) v.push_back(h[b * 16 + slot * 4]);
    std::sort(v.begin(), v.end());
    printf("  wave%d %7.1f", slot, (double)v[v.size() / 2] / (iters * 8.0));
  }
  printf("   (s_memtime ticks per MFMA slot)\n");
}

int main() {
  float *out, *in;
  long long* cyc;
  hipMalloc(&out, (1 << 20) * 4);
  hipMalloc(&in, 8192 * 4);
  hipMalloc(&cyc, 256 * 16 * 8);
  std::vector<float> hin(8192);
  for (int i = 0; i < 8192; ++i) hin[i] = (float)((i * 2654435761u >> 8) & 1023) / 512.f - 1.f;
  hipMemcpy(in, hin.data(), 8192 * 4, hipMemcpyHostToDevice);
  // same-wave fillers, one wave per SIMD
  run<0>("1 wave/SIMD, MFMA only", 1, 0x1, out, cyc, in);
  run<2>("1 wave/SIMD, 1 MFMA + F fma", 1, 0x0, out, cyc, in);
  run<4>("1 wave/SIMD, 1 MFMA + F fma", 1, 0x0, out, cyc, in);
  run<6>("1 wave/SIMD, 1 MFMA + F fma", 1, 0x0, out, cyc, in);
  run<8>("1 wave/SIMD, 1 MFMA + F fma", 1, 0x0, out, cyc, in);
  run<12>("1 wave/SIMD, 1 MFMA + F fma", 1, 0x0, out, cyc, in);
  // two waves per SIMD
  run<0>("2 waves/SIMD, both MFMA only", 2, 0x5, out, cyc, in);
  run<4>("2 waves/SIMD, both mixed", 2, 0x0, out, cyc, in);
  run<8>("2 waves/SIMD, both mixed", 2, 0x0, out, cyc, in);
  run<4>("2 waves/SIMD, wave0 MFMA only + wave1 VALU only (4/slot)", 2, 0x9, out, cyc, in);
  run<8>("2 waves/SIMD, wave0 MFMA only + wave1 VALU only (8/slot)", 2, 0x9, out, cyc, in);
  run<8>("2 waves/SIMD, wave0 idle + wave1 VALU only (8/slot)", 2, 0xB, out, cyc, in);
  // four waves per SIMD
  run<0>("4 waves/SIMD, all MFMA only", 4, 0x55, out, cyc, in);
  run<4>("4 waves/SIMD, all mixed", 4, 0x00, out, cyc, in);
  run<4>("4 waves/SIMD, 2 MFMA only + 2 VALU only (4/slot)", 4, 0xA5, out, cyc, in);
  run<8>("4 waves/SIMD, 2 MFMA only + 2 VALU only (8/slot)", 4, 0xA5, out, cyc, in);
  return 0;
}
