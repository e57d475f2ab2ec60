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
//               sequence), g = sum_i w_i (m_i - x)/(1 - t + eps), blend
//               (1-gamma) v + gamma g, and optionally the Euler update.
// Both are HBM/L2-bound elementwise-reduction kernels; the MC set (N x D) stays
// L2-resident.
#include "rgfm_device.h"

namespace rgfm {

constexpr int GD = 64;        // D-chunk
constexpr int GLD = GD + 4;   // LDS row stride (floats): conflict-free b128 rows

// Squared distances, sliced along D: block (bx, by, z) adds up slice z of one modality for a 32 x 32 tile of
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

__global__ __launch_bounds__(256) void guid_logp_kernel(const GuidanceArgs a_in) {
  const GuidanceArgs a = with_schedule(a_in);
  __shared__ __attribute__((aligned(16))) float sx[32 * GLD];
  __shared__ __attribute__((aligned(16))) float sm[32 * GLD];
  const int tid = threadIdx.x;
  const int tb = tid >> 4, ti = tid & 15;
  const int bb = blockIdx.x * 32, ib = blockIdx.y * 32;
  const int z = blockIdx.z;
  const int part = z >= a.nsx ? 1 : 0;
  const float* X = part ? a.y : a.x;
  const float* M = part ? a.mc_y1 : a.mc_x1;
  const int D = part ? a.dy : a.dx;
  const int dbeg = (part ? z - a.nsx : z) * a.slice_len;
  const int dend = dbeg + a.slice_len < D ? dbeg + a.slice_len : D;
  // chunk partials in fp32 (64 terms), running total in fp64: the sum is then exact
  // to fp32 rounding, which matters because l is later scaled by 1/sigma_t^2 (up to ~8e3)
  double S[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int d0 = dbeg; d0 < dend; d0 += GD) {
    __syncthreads();
    // 32 rows x 16 float4 for each operand: 512 items each, 2 per thread
    for (int it = tid; it < 512; it += 256) {
      const int row = it >> 4, q = it & 15;
      const int d = d0 + q * 4;
      f32x4 vx = {0.f, 0.f, 0.f, 0.f}, vm = {0.f, 0.f, 0.f, 0.f};
      if (d < dend) {
        if (bb + row < a.B) vx = *reinterpret_cast<const f32x4*>(X + (size_t)(bb + row) * D + d);
        if (ib + row < a.N) {
          vm = *reinterpret_cast<const f32x4*>(M + (size_t)(ib + row) * D + d);
          vm.x = __fmul_rn(a.tf, vm.x);  // mu = t * x_1 (rounded, as the reference)
          vm.y = __fmul_rn(a.tf, vm.y);
          vm.z = __fmul_rn(a.tf, vm.z);
          vm.w = __fmul_rn(a.tf, vm.w);
        }
      }
      *reinterpret_cast<f32x4*>(sx + row * GLD + q * 4) = vx;
      *reinterpret_cast<f32x4*>(sm + row * GLD + q * 4) = vm;
    }
    __syncthreads();
    float c[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int d = 0; d < GD; d += 4) {
      f32x4 xv[2], mv[2];
      xv[0] = *reinterpret_cast<const f32x4*>(sx + tb * GLD + d);
      xv[1] = *reinterpret_cast<const f32x4*>(sx + (tb + 16) * GLD + d);
      mv[0] = *reinterpret_cast<const f32x4*>(sm + ti * GLD + d);
      mv[1] = *reinterpret_cast<const f32x4*>(sm + (ti + 16) * GLD + d);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float df;
          df = xv[p].x - mv[q].x, c[p][q] += df * df;
          df = xv[p].y - mv[q].y, c[p][q] += df * df;
          df = xv[p].z - mv[q].z, c[p][q] += df * df;
          df = xv[p].w - mv[q].w, c[p][q] += df * df;
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) S[p][q] += (double)c[p][q];
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
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
  for (int i = lane; i < N; i += 64) {
    const float w = swr[i] / ws;
    a.wbuf[(size_t)b * N + i] = w;
    if (a.weights_out) a.weights_out[(size_t)b * N + i] = w;
  }
}

// grid (ceil(B/16), ceil(D/256)); one launch per modality (part).  Wave w owns rows b0 + 4w .. + 3 (their weights sit
// in a wave-private piece of LDS: no workgroup barrier), lane l the four components d0 + 4l .. + 3.  The four waves
// of a workgroup stream the same 256-wide column block of the MC set, so it leaves L2 once per 16 rows, and the grid
// has 512 workgroups at the benchmark shape (the first version: 4 rows x 1024 columns per workgroup, 128
// workgroups, weights recomputed by every one of them).  Arithmetic and summation order are unchanged.
template <int RW>  // rows per wave: 4 (16 rows per workgroup; N <= 1024), or 1 for larger MC sets (LDS: [4 RW][N] floats)
__global__ __launch_bounds__(256) void guid_apply_kernel(const GuidanceArgs a_in, int part) {
  const GuidanceArgs a = with_schedule(a_in);
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [4 RW][N]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = a.N;
  const int b0 = blockIdx.x * (4 * RW) + wave * RW;
  float* swv = sw + (size_t)wave * RW * N;
#pragma unroll
  for (int r = 0; r < RW; ++r)
    for (int i = lane; i < N; i += 64) swv[r * N + i] = (b0 + r < a.B) ? a.wbuf[(size_t)(b0 + r) * N + i] : 0.f;
  // (wave-private LDS: the lanes of this wave wrote what they now read; LDS operations of a wave execute in order)

  const float* X = part ? a.y : a.x;
  const float* M = part ? a.mc_y1 : a.mc_x1;
  float* V = part ? a.vy : a.vx;
  float* XS = part ? a.y_state : a.x_state;
  const int D = part ? a.dy : a.dx;
  const int d = blockIdx.y * 256 + lane * 4;
  if (d >= D) return;
  f32x4 xv[RW], g[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    g[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    xv[r] = (b0 + r < a.B) ? *reinterpret_cast<const f32x4*>(X + (size_t)(b0 + r) * D + d) : g[r];
  }
  // (m - x) / (1 - t + eps) of the reference (sample_mnist_svhn.py:159) as a multiplication by the correctly rounded
  // reciprocal: one more rounding (<= 1 ulp per term) for a loop that was 70 % division instructions
  const float rcden = 1.0f / a.cden;
  for (int i = 0; i < N; ++i) {
    float w[RW];
    bool any = false;
#pragma unroll
    for (int r = 0; r < RW; ++r) w[r] = swv[r * N + i], any = any || (w[r] != 0.f);
    // exact skip: a zero weight contributes +0 to every sum
    if (!any) continue;
    const f32x4 m = *reinterpret_cast<const f32x4*>(M + (size_t)i * D + d);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      if (w[r] != 0.f) {
        g[r].x = __fadd_rn(g[r].x, __fmul_rn(w[r], __fmul_rn(m.x - xv[r].x, rcden)));
        g[r].y = __fadd_rn(g[r].y, __fmul_rn(w[r], __fmul_rn(m.y - xv[r].y, rcden)));
        g[r].z = __fadd_rn(g[r].z, __fmul_rn(w[r], __fmul_rn(m.z - xv[r].z, rcden)));
        g[r].w = __fadd_rn(g[r].w, __fmul_rn(w[r], __fmul_rn(m.w - xv[r].w, rcden)));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if (b0 + r >= a.B) continue;
    const size_t o = (size_t)(b0 + r) * D + d;
    const f32x4 v = *reinterpret_cast<const f32x4*>(V + o);
    f32x4 nv;
    nv.x = __fadd_rn(__fmul_rn(a.g1, v.x), __fmul_rn(a.g2, g[r].x));
    nv.y = __fadd_rn(__fmul_rn(a.g1, v.y), __fmul_rn(a.g2, g[r].y));
    nv.z = __fadd_rn(__fmul_rn(a.g1, v.z), __fmul_rn(a.g2, g[r].z));
    nv.w = __fadd_rn(__fmul_rn(a.g1, v.w), __fmul_rn(a.g2, g[r].w));
    if (XS) {
      f32x4 xs = xv[r];
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
  dim3 g1((a.B + 31) / 32, (a.N + 31) / 32, a.nsx + a.nsy);
  hipLaunchKernelGGL(guid_logp_kernel, g1, dim3(256), 0, s, a);
}

void launch_guid_apply(const GuidanceArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(guid_weights_kernel, dim3((a.B + 3) / 4), dim3(256), (size_t)4 * a.N * sizeof(float), s, a);
  if ((size_t)16 * a.N * sizeof(float) <= 64 * 1024) {
    const size_t lds = (size_t)16 * a.N * sizeof(float);
    hipLaunchKernelGGL(guid_apply_kernel<4>, dim3((a.B + 15) / 16, (a.dx + 255) / 256), dim3(256), lds, s, a, 0);
    hipLaunchKernelGGL(guid_apply_kernel<4>, dim3((a.B + 15) / 16, (a.dy + 255) / 256), dim3(256), lds, s, a, 1);
  } else {  // (N <= 4096: rgfm_guidance_apply / the samplers check)
    const size_t lds = (size_t)4 * a.N * sizeof(float);
    hipLaunchKernelGGL(guid_apply_kernel<1>, dim3((a.B + 3) / 4, (a.dx + 255) / 256), dim3(256), lds, s, a, 0);
    hipLaunchKernelGGL(guid_apply_kernel<1>, dim3((a.B + 3) / 4, (a.dy + 255) / 256), dim3(256), lds, s, a, 1);
  }
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
