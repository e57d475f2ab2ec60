// guidance.hip -- MC importance-weighted guidance (mc_feng) and the Euler update.
//
// Replaces the per-step guidance block of the reference samplers
// (src/sample_mnist_svhn.py:124-171, src/utils/flow_utils.py:273-369), which
// materialises six [B, N, D] broadcast temporaries per step, by two passes that
// never leave registers/LDS:
//   guid_logp : l[b,i] = -0.5 (|x_b - t m^x_i|^2 / s2) + -0.5 (|y_b - t m^y_i|^2 / s2)
//               (direct differences, no GEMM expansion: late-time weights are
//               scaled by 1/sigma_t^2 ~ 8e3 and cancellation would flip near-ties)
//   guid_apply: row-normalised importance weights (exact reference epsilon
//               sequence), g = sum_i w_i (m_i - x)/(1 - t + eps) as an fp32-MFMA
//               GEMM [B,N] x [N,D], blend (1-gamma) v + gamma g, and optionally the
//               Euler update.
// guid_logp is a VALU-bound elementwise reduction, guid_apply a small fp32 GEMM; the MC set
// (N x D) stays L2-resident.
#include "rgfm_device.h"

namespace rgfm {

constexpr int GD = 64;        // D-chunk
constexpr int GLD = GD + 4;   // LDS row stride (floats): conflict-free b128 rows

// Squared distances, sliced along D: block (bx, by, z) adds up slice z of one modality for a 32 x 64 tile of
// (row, MC sample) pairs and writes the fp64 partial sum; guid_apply adds the slices (fp64 sums of fp32 chunk
// partials are exact, so the slicing does not change a bit) -- 4x the workgroups of an unsliced launch at the
// benchmark shape, where 16 x 8 tiles would leave half of the 256 CUs idle on a kernel that nothing overlaps.
// the step's scalars: launch arguments, or (hipGraph replay) the schedule row of the device-side step counter
__device__ __forceinline__ GuidanceArgs with_schedule(const GuidanceArgs& in) {
  GuidanceArgs a = in;
  if (in.sched) {
    const float* q = in.sched + 4 * (size_t)*in.step_ptr;
    a.tf = q[0], a.s2 = q[1], a.cden = q[2];
  }
  return a;
}

// Thread = 2 rows x 4 MC samples of a 32 x 64 tile; the differences and squares are packed-fp32 instructions (two
// elements each: v_pk_add_f32, v_pk_fma_f32 -- full rate where no MFMA runs beside them), accumulated per element parity and
// added at the end of a 64-element chunk.  The next chunk's rows are fetched into registers while this one is multiplied.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// a - b on both halves in one instruction (hipcc scalarises a v2f32 subtraction, also when written as a + (-b) or fma(b, -1, a))
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__global__ __launch_bounds__(256) void guid_logp_kernel(const GuidanceArgs a_in) {
  const GuidanceArgs a = with_schedule(a_in);
  __shared__ __attribute__((aligned(16))) float sx[32 * GLD];
  __shared__ __attribute__((aligned(16))) float sm[64 * GLD];
  const int tid = threadIdx.x;
  const int tb = tid >> 4, ti = tid & 15;
  const int bb = blockIdx.x * 32, ib = blockIdx.y * 64;
  const int z = blockIdx.z;
  const int part = z >= a.nsx ? 1 : 0;
  const float* X = part ? a.y : a.x;
  const float* M = part ? a.mc_y1 : a.mc_x1;
  const int D = part ? a.dy : a.dx;
  const int dbeg = (part ? z - a.nsx : z) * a.slice_len;
  const int dend = dbeg + a.slice_len < D ? dbeg + a.slice_len : D;
  // staging items: item it < 512 is (row it / 16, float4 it % 16) of x, the 1024 behind them of the MC set; six per thread
  const float* src[6];
  bool ok[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int it = tid + 256 * j;
    const bool isx = it < 512;
    const int row = isx ? it >> 4 : (it - 512) >> 4;
    const int grow = isx ? bb + row : ib + row;
    ok[j] = isx ? grow < a.B : grow < a.N;
    src[j] = (isx ? X : M) + (size_t)(ok[j] ? grow : 0) * D;  // (the row; the float4 inside it: fetch)
  }
  f32x4 rv[6];
  auto fetch = [&](int d0) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int d = d0 + (tid & 15) * 4;
      const bool in = ok[j] && d < dend;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src[j] + (in ? d : 0));  // (clamped to the row's first float4: always valid)
      rv[j] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // chunk partials in fp32 (64 terms), running total in fp64: the sum is then exact
  // to fp32 rounding, which matters because l is later scaled by 1/sigma_t^2 (up to ~8e3)
  double S[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) S[p][q] = 0.0;
  fetch(dbeg);
  for (int d0 = dbeg; d0 < dend; d0 += GD) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int it = tid + 256 * j;
      f32x4 v = rv[j];
      if (it >= 512) {  // mu = t * x_1 (rounded, as the reference)
        v.x = __fmul_rn(a.tf, v.x), v.y = __fmul_rn(a.tf, v.y), v.z = __fmul_rn(a.tf, v.z), v.w = __fmul_rn(a.tf, v.w);
        *reinterpret_cast<f32x4*>(sm + ((it - 512) >> 4) * GLD + (it & 15) * 4) = v;
      } else {
        *reinterpret_cast<f32x4*>(sx + (it >> 4) * GLD + (it & 15) * 4) = v;
      }
    }
    __syncthreads();
    if (d0 + GD < dend) fetch(d0 + GD);
    f32x2 c[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[p][q] = f32x2{0.f, 0.f};
#pragma unroll
    for (int d = 0; d < GD; d += 4) {
      f32x4 xv[2], mv[4];
#pragma unroll
      for (int p = 0; p < 2; ++p) xv[p] = *reinterpret_cast<const f32x4*>(sx + (tb + 16 * p) * GLD + d);
#pragma unroll
      for (int q = 0; q < 4; ++q) mv[q] = *reinterpret_cast<const f32x4*>(sm + (ti + 16 * q) * GLD + d);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 d0v = pk_sub(f32x2{xv[p].x, xv[p].y}, f32x2{mv[q].x, mv[q].y});
          const f32x2 d1v = pk_sub(f32x2{xv[p].z, xv[p].w}, f32x2{mv[q].z, mv[q].w});
          c[p][q] = __builtin_elementwise_fma(d0v, d0v, c[p][q]);
          c[p][q] = __builtin_elementwise_fma(d1v, d1v, c[p][q]);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) S[p][q] += (double)(c[p][q].x + c[p][q].y);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = bb + tb + 16 * p, i = ib + ti + 16 * q;
      if (b < a.B && i < a.N) a.dist[((size_t)z * a.B + b) * a.N + i] = S[p][q];
    }
}

// l[b,i] = -0.5 |x_b - t m^x_i|^2 / s2 + -0.5 |y_b - t m^y_i|^2 / s2 from the sliced sums (each modality's
// term rounded to fp32 on its own, then added: the reference's two-statement form, :141-143)
__device__ __forceinline__ float logp_of(const GuidanceArgs& a, int b, int i) {
  double sx = 0.0, sy = 0.0;
  for (int z = 0; z < a.nsx; ++z) sx += a.dist[((size_t)z * a.B + b) * a.N + i];
  for (int z = a.nsx; z < a.nsx + a.nsy; ++z) sy += a.dist[((size_t)z * a.B + b) * a.N + i];
  const float lx = __fadd_rn(0.f, __fmul_rn(-0.5f, (float)sx) / a.s2);
  return __fadd_rn(lx, __fmul_rn(-0.5f, (float)sy) / a.s2);
}

__device__ __forceinline__ float wave_sum_g(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_g(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Importance weights of one row per wave (sample_mnist_svhn.py:146-156), written once per step to a.wbuf [B][N]
// (and to weights_out when the caller asks for them): guid_apply's workgroups -- sixteen column blocks per row
// block -- read them back instead of each recomputing the softmax from the distance slices.
__global__ __launch_bounds__(256) void guid_weights_kernel(const GuidanceArgs a_in) {
  const GuidanceArgs a = with_schedule(a_in);
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [4][N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  float* swr = sw + (size_t)wave * N;
  float mx = -INFINITY;
  for (int i = lane; i < N; i += 64) {
    const float l = logp_of(a, b, i);
    swr[i] = l;
    mx = fmaxf(mx, l);
  }
  mx = wave_max_g(mx);
  float ps = 0.f, zs = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float p = expf(swr[i] - mx);
    swr[i] = p;
    ps += p;
    zs += __fmul_rn(a.mc_ratios[i], p);
  }
  ps = wave_sum_g(ps);
  zs = wave_sum_g(zs);
  const float pbar = __fadd_rn(ps / (float)N, 1e-10f);
  const float zbar = __fadd_rn(zs / (float)N, 1e-10f);
  float ws = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float w = __fmul_rn(a.mc_ratios[i] / zbar, swr[i] / pbar);
    swr[i] = w;
    ws += w;
  }
  ws = __fadd_rn(wave_sum_g(ws), 1e-10f);
  float rsum = 0.f;
  for (int i = lane; i < N; i += 64) {
    const float w = swr[i] / ws;
    a.wbuf[(size_t)b * N + i] = w;
    if (a.weights_out) a.weights_out[(size_t)b * N + i] = w;
    rsum += w;
  }
  rsum = wave_sum_g(rsum);  // the row's weight sum as guid_apply_mfma_kernel needs it (1 up to rounding and the epsilon)
  if (lane == 0) a.wsum[b] = rsum;
}

// The guided velocity g_b = sum_i w_bi (m_i - x_b) / c (sample_mnist_svhn.py:157-160) as the fp32 GEMM it is:
//   g_b = (sum_i w_bi m_i  -  x_b sum_i w_bi) / c,   [B, N] x [N, D] on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32
// accumulation: the arithmetic of an fmaf chain), then the blend (1 - gamma) v + gamma g and the optional Euler update in
// the epilogue.  The first versions formed every (m_i - x_b) / c w_bi on the vector ALU in the reference's operation
// order: 16 VALU instructions per (row, sample, float4), 110 us at B = 512, N = 256, D = 3072 + 1024 against 7 us of
// matrix work.  Both forms carry the same error: terms of size |m - x| / c summed to a result that may be far smaller.
//
// grid (ceil(B / 32), ceil(dx / 128) + ceil(dy / 128)): both modalities in one launch; workgroup = a 32-row x 128-column output tile, its four
// waves each take a quarter of the MC samples (the two lane halves of a wave: the two halves of that quarter, one
// sample each per MFMA) and add their partial tiles through LDS in wave order.  Lane l supplies W[b0 + l % 32][k] as
// the A element and M[k][d0 + 4 (l % 32) + j] as the B element of column tile j: one float4 of the MC set feeds four
// MFMAs and the lane ends up with four consecutive columns of a row (float4 loads and stores in the epilogue).  A
// row's result depends on its own weights and the MC set only, in an order fixed by N: independent of the batch it is in.
template <bool W4>  // N % 32 == 0: a lane's run of samples starts at a multiple of four and its weights arrive as float4
__global__ __launch_bounds__(256, 2) void guid_apply_mfma_kernel(const GuidanceArgs a_in) {
  const GuidanceArgs a = with_schedule(a_in);
  const int nbx = (a.dx + 127) >> 7;  // column blocks of modality x; those of y behind them
  const int part = (int)blockIdx.y >= nbx ? 1 : 0;
  extern __shared__ __attribute__((aligned(16))) float sred[];  // [4 waves][64 accumulators][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int N = a.N, B = a.B;
  const float* X = part ? a.y : a.x;
  const float* M = part ? a.mc_y1 : a.mc_x1;
  float* V = part ? a.vy : a.vx;
  float* XS = part ? a.y_state : a.x_state;
  const int D = part ? a.dy : a.dx;
  const int b0 = blockIdx.x * 32;
  const int d = ((int)blockIdx.y - (part ? nbx : 0)) * 128 + 4 * l31;
  const bool dok = d < D;
  const int KH = (N + 7) >> 3;              // samples per (wave, lane half)
  const int k0 = (wave * 2 + hf) * KH;
  const bool rowok = b0 + l31 < B;
  const float* wrow = a.wbuf + (size_t)(rowok ? b0 + l31 : 0) * N;
  const float* mcol = M + (dok ? d : 0);
  constexpr int U = 16;  // samples per group: two groups of registers, one being multiplied (64 MFMAs, 1.7 us) while the other loads
  struct Grp {
    float w[U];
    f32x4 m[U];
  };
  auto load = [&](Grp& g, int s0) {  // (branch-free: clamped addresses, selects)
    if constexpr (W4) {
#pragma unroll
      for (int u = 0; u < U; u += 4) {
        const int s = s0 + u;
        const bool ok = s < KH && rowok;  // (whole quads: KH % 4 == 0)
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + (s < KH ? k0 + s : 0));
        g.w[u] = ok ? wv.x : 0.f, g.w[u + 1] = ok ? wv.y : 0.f, g.w[u + 2] = ok ? wv.z : 0.f, g.w[u + 3] = ok ? wv.w : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u, k = k0 + s;
      const bool kok = s < KH && k < N;
      const int kc = kok ? k : 0;
      if constexpr (!W4) {
        const float wv = wrow[kc];
        g.w[u] = (kok && rowok) ? wv : 0.f;  // (a sample past the end enters as 0 x finite)
      }
      g.m[u] = *reinterpret_cast<const f32x4*>(mcol + (size_t)kc * D);
    }
  };
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  auto mult = [&](const Grp& g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[u], g.m[u].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[u], g.m[u].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[u], g.m[u].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w[u], g.m[u].w, acc[3], 0, 0, 0);
    }
  };
  Grp g0, g1;  // (no register copies between iterations: a copy would wait for the loads it copies)
  load(g0, 0);
  for (int s0 = 0; s0 < KH; s0 += 2 * U) {
    load(g1, s0 + U);  // (past the end: clamped addresses, zero weights)
    mult(g0);
    if (s0 + 2 * U < KH) load(g0, s0 + 2 * U);
    if (s0 + U < KH) mult(g1);
  }
  // the epilogue's rows of x and v (this wave finishes rows b0 + 8 wave + 4 hf + 0 .. 3), fetched under the exchange of the partial tiles
  f32x4 ex[4], ev[4];
  float esw[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int b = b0 + 8 * wave + 4 * hf + rr;
    const size_t o = (size_t)(b < B ? b : 0) * D + (dok ? d : 0);
    ex[rr] = *reinterpret_cast<const f32x4*>(X + o);
    ev[rr] = *reinterpret_cast<const f32x4*>(V + o);
    esw[rr] = a.wsum[b < B ? b : 0];
  }
  // partial tiles -> LDS; wave w then owns accumulator rows r = 4w .. 4w + 3: output rows b0 + 8w + 4 (lane / 32) + (r & 3)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) sred[((wave * 64) + j * 16 + r) * 64 + lane] = acc[j][r];
  __syncthreads();
  const float rcden = 1.0f / a.cden;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * wave + rr;
    const int b = b0 + 8 * wave + 4 * hf + rr;
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = sred[(j * 16 + r) * 64 + lane];
#pragma unroll
      for (int p = 1; p < 4; ++p) t = __fadd_rn(t, sred[((p * 64) + j * 16 + r) * 64 + lane]);
      c[j] = t;
    }
    if (b >= B || !dok) continue;
    const size_t o = (size_t)b * D + d;
    const f32x4 xv = ex[rr], v = ev[rr];
    const float sw = esw[rr];
    f32x4 g, nv;
    g.x = __fmul_rn(__fsub_rn(c[0], __fmul_rn(xv.x, sw)), rcden);
    g.y = __fmul_rn(__fsub_rn(c[1], __fmul_rn(xv.y, sw)), rcden);
    g.z = __fmul_rn(__fsub_rn(c[2], __fmul_rn(xv.z, sw)), rcden);
    g.w = __fmul_rn(__fsub_rn(c[3], __fmul_rn(xv.w, sw)), rcden);
    nv.x = __fadd_rn(__fmul_rn(a.g1, v.x), __fmul_rn(a.g2, g.x));
    nv.y = __fadd_rn(__fmul_rn(a.g1, v.y), __fmul_rn(a.g2, g.y));
    nv.z = __fadd_rn(__fmul_rn(a.g1, v.z), __fmul_rn(a.g2, g.z));
    nv.w = __fadd_rn(__fmul_rn(a.g1, v.w), __fmul_rn(a.g2, g.w));
    if (XS) {
      f32x4 xs = xv;
      xs.x = __fadd_rn(xs.x, __fmul_rn(nv.x, a.dt));
      xs.y = __fadd_rn(xs.y, __fmul_rn(nv.y, a.dt));
      xs.z = __fadd_rn(xs.z, __fmul_rn(nv.z, a.dt));
      xs.w = __fadd_rn(xs.w, __fmul_rn(nv.w, a.dt));
      *reinterpret_cast<f32x4*>(XS + o) = xs;
    } else {
      *reinterpret_cast<f32x4*>(V + o) = nv;
    }
  }
}

void launch_guid_logp(const GuidanceArgs& a, hipStream_t s) {
  dim3 g1((a.B + 31) / 32, (a.N + 63) / 64, a.nsx + a.nsy);
  hipLaunchKernelGGL(guid_logp_kernel, g1, dim3(256), 0, s, a);
}

int guid_apply_init() {  // (64 KB of dynamic LDS)
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&guid_apply_mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) |
         (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&guid_apply_mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
}

void launch_guid_weights(const GuidanceArgs& a, hipStream_t s) {  // (needs the distances only: no velocity)
  hipLaunchKernelGGL(guid_weights_kernel, dim3((a.B + 3) / 4), dim3(256), (size_t)4 * a.N * sizeof(float), s, a);
}

void launch_guid_apply(const GuidanceArgs& a, hipStream_t s) {
  const size_t lds = (size_t)4 * 64 * 64 * sizeof(float);
  const dim3 grid((a.B + 31) / 32, (a.dx + 127) / 128 + (a.dy + 127) / 128);
  if (a.N % 32 == 0) hipLaunchKernelGGL(guid_apply_mfma_kernel<true>, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(guid_apply_mfma_kernel<false>, grid, dim3(256), lds, s, a);
}

// x <- x + v * dt (mul, then add: two roundings like the reference's x_t + v * dt)
__global__ void euler_kernel(float* x, const float* v, size_t n, float dt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = __fadd_rn(x[i], __fmul_rn(v[i], dt));
}

// Per-step scalars of the guidance block for steps step_begin .. step_begin + ns - 1: the reference's Python-double
// arithmetic (sample_mnist_svhn.py:115,127,135,159: t = step * dt, sigma_t = 1 - t + eps), the same IEEE double
// operations the host path performs, rounded to fp32 where a tensor op consumes them.
__global__ void guid_schedule_kernel(float* sched, int step_begin, int ns, int num_steps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const double dtd = 1.0 / (double)num_steps;
  const double t = (double)(step_begin + i) * dtd;
  const double eps = 1e-3;
  const double sigma_t = 1.0 - t + eps;
  sched[4 * i] = (float)t, sched[4 * i + 1] = (float)(sigma_t * sigma_t), sched[4 * i + 2] = (float)(1.0 - t + eps), sched[4 * i + 3] = 0.f;
}
void launch_guid_schedule(float* sched, int step_begin, int ns, int num_steps, hipStream_t s) {
  hipLaunchKernelGGL(guid_schedule_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, sched, step_begin, ns, num_steps);
}

__global__ void step_inc_kernel(int* step) { *step += 1; }
void launch_step_inc(int* step, hipStream_t s) { hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step); }

void launch_euler(float* x, const float* v, size_t n, float dt, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(euler_kernel, dim3(blocks), dim3(256), 0, s, x, v, n, dt);
}

}  // namespace rgfm
