// rgfm_device.h -- device helpers shared by the conv kernels.
#pragma once
#include "rgfm_kernels.h"

namespace rgfm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

// SiLU on the conv load path: v_exp_f32 + v_rcp_f32 (each ~1 ulp) instead of the
// IEEE expf / division sequences (~25 VALU instructions per element, which the
// staging of every K-chunk pays 28 times per thread).  |rel err| < 3e-7.
__device__ __forceinline__ float silu_fast(float v) {
  const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
  return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// Stage the input halo tile of channel chunk `c` (16 channels starting at concat
// channel c) into sA[halo_px][LDP].  Shared with conv_out.
template <int MODE>
__device__ __forceinline__ void stage_input(float* sA, const float* in0, const float* in1, int C0,
                                            int C1, int Hin, int Win, const float* ab, int c,
                                            int B, int b0, int row0, int H, int W, int HR, int WR,
                                            int halo_px, int tid, int nthreads) {
  const float* src;
  int cs, cc;
  if (c < C0) {
    src = in0, cs = C0, cc = c;
  } else {
    src = in1, cs = C1, cc = c - C0;
  }
  const int ctot = C0 + C1;
  const int per = HR * WR;
  for (int it = tid; it < halo_px * 4; it += nthreads) {
    const int hp = it >> 2, q = it & 3;
    const int s = hp / per;
    const int rem = hp - s * per;
    const int hy = rem / WR;
    const int hx = rem - hy * WR;
    const int b = b0 + s;
    int y, x;
    bool ok;
    if (MODE == CONV_S1) {
      y = row0 + hy - 1;
      x = hx - 1;
      ok = (y >= 0) && (y < H) && (x >= 0) && (x < W);
    } else if (MODE == CONV_S2) {
      y = 2 * row0 + hy - 1;
      x = hx - 1;
      ok = (y >= 0) && (y < Hin) && (x >= 0) && (x < Win);
    } else {
      const int yu = row0 + hy - 1, xu = hx - 1;
      ok = (yu >= 0) && (yu < H) && (xu >= 0) && (xu < W);
      y = yu >> 1;
      x = xu >> 1;
    }
    ok = ok && (b < B);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      v = *reinterpret_cast<const f32x4*>(src + ((size_t)(b * Hin + y) * Win + x) * cs + cc + q * 4);
      if (ab) {
        const f32x4* p = reinterpret_cast<const f32x4*>(ab + ((size_t)b * ctot + c + q * 4) * 2);
        const f32x4 e0 = p[0], e1 = p[1];
        v.x = silu_fast(e0.x * v.x + e0.y);
        v.y = silu_fast(e0.z * v.y + e0.w);
        v.z = silu_fast(e1.x * v.z + e1.y);
        v.w = silu_fast(e1.z * v.w + e1.w);
      }
    }
    *reinterpret_cast<f32x4*>(sA + hp * LDP + q * 4) = v;
  }
}

}  // namespace rgfm
