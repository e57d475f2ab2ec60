// Microbenchmark: inner loop of an fp32 conv emulated with 3-way bf16 split operands
// (a = h + m + l; products hh, hm, mh, mm, hl, lh on v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Per tap and wave: MT x NT accumulator tiles, (MT + NT) x 3 ds_read_b128 fragment loads, MT*NT*6 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 112;  // bytes per (pixel | channel) record: 3 planes x 16 bf16 + 16 pad

template <int MT, int NT, int WPB, int NPROD>
__global__ __launch_bounds__(WPB * 64, (WPB == 4 ? 2 : 1)) void k(float* out, int iters, const float* in) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                       // [WPB*MT*32 + 64 px][ROWB]
  char* sB = smem + (WPB * MT * 32 + 72) * ROWB;  // [9][NT*32][ROWB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid / 64, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < ((WPB * MT * 32 + 72) * ROWB + 9 * NT * 32 * ROWB) / 4; i += WPB * 64) ((float*)smem)[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[MT][NT];
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int abase[MT], bbase[NT];
  for (int i = 0; i < MT; ++i) abase[i] = (wave * MT * 32 + i * 32 + l31) * ROWB + h * 16;
  for (int j = 0; j < NT; ++j) bbase[j] = (j * 32 + l31) * ROWB + h * 16;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    for (int tap = 0; tap < 9; ++tap) {
      asm volatile("" ::: "memory");
      const int aoff = ((tap / 3) * 34 + tap % 3) * ROWB, boff = tap * NT * 32 * ROWB;
      bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(sA + abase[i] + (aoff % (64 * ROWB)) + p * 32);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(sB + bbase[j] + boff + p * 32);
      // product order: small terms first
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int q = 6 - NPROD; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[q]], bf[j][PB[q]], acc[i][j], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < MT; ++i) for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (tid == 0 && blockIdx.x == 0) out[1 << 22] = (float)(t1 - t0) / (iters * 9.0f);
}

template <int MT, int NT, int WPB, int NPROD>
void run(const char* tag) {
  float *out, *in;
  hipMalloc(&out, ((1 << 22) + 16) * 4);
  hipMalloc(&in, 8192 * 4);
  hipMemset(in, 0, 8192 * 4);
  const size_t lds = (WPB * MT * 32 + 72) * ROWB + 9 * NT * 32 * ROWB;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MT, NT, WPB, NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 400, blocks = 256 * (WPB == 4 ? 2 : 1);
  k<MT, NT, WPB, NPROD><<<blocks, WPB * 64, lds>>>(out, 5, in);
  hipEventRecord(e0);
  k<MT, NT, WPB, NPROD><<<blocks, WPB * 64, lds>>>(out, iters, in);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("%s: launch failed (lds %zu)\n", tag, lds); return; }
  float ms, cyc;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&cyc, out + (1 << 22), 4, hipMemcpyDeviceToHost);
  // fp32-equivalent flops: per tap per wave 2 * (MT*32) * (NT*32) * 16
  const double eq = (double)blocks * WPB * iters * 9 * 2.0 * (MT * 32) * (NT * 32) * 16 / (ms * 1e-3) / 1e12;
  const double ideal = MT * NT * NPROD * 32.0;
  printf("%-28s MT=%d NT=%d waves/blk=%d prods=%d lds=%3zuKB: %7.1f clk/tap (MFMA-only ideal %5.0f, %4.1f%%)  fp32-equivalent %6.1f TFLOP/s\n",
         tag, MT, NT, WPB, NPROD, lds / 1024, cyc, ideal, 100.0 * ideal / cyc, eq);
  hipFree(out);
  hipFree(in);
}

int main() {
  run<2, 2, 4, 6>("2x2 tiles, 2 blk/CU");
  run<2, 2, 8, 6>("2x2 tiles, 8 waves 1 blk/CU");
  run<4, 2, 4, 6>("4x2 tiles, 2 blk/CU");
  run<2, 4, 4, 6>("2x4 tiles, 2 blk/CU");
  run<4, 2, 8, 6>("4x2 tiles, 8 waves");
  run<2, 2, 4, 3>("2x2, 3 products (bf16x2)");
  run<2, 2, 4, 1>("2x2, 1 product (plain bf16)");
  return 0;
}
