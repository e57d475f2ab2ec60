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

// GroupNorm finalize of ONE sample by ONE wave (see ConvArgs::fin_ab).  lane = group * 8 + sub; sub strides over
// the group's channels.  Group statistics from the (count, mean, M2) partials of every (channel, part), one pass in
// fp64 without divisions in the loop (an fp64 divide is ~40 instructions and this runs on the conv's tail):
//   N = sum n_p,  S1 = sum n_p mean_p,  S2 = sum [M2_p + n_p mean_p^2]  ->  mean = S1 / N,  M2 = S2 - N mean^2
// (inputs are fp32; the cancellation in fp64 costs ~1e-13), then a = rstd * gamma, b = beta - mean * a.
__device__ __forceinline__ double sub_sum(double v) {  // sum over the 8 sub-lanes of a group
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void fin_sample(const ConvArgs& a, int b, int lane, int nparts0, bool t2) {
  const int C0 = a.Cout, C = a.Cout + a.fin_C1;
  const int cpg = C >> 3;  // 8 groups (C >= 32 on this path)
  const int gi = lane >> 3, sub = lane & 7;
  const TileGeom g = a.g;
  double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int c = gi * cpg + sub; c < (gi + 1) * cpg; c += 8) {
    const bool first = c < C0;
    const float* st = first ? a.stats_out : a.fin_stats1;
    const int cs = first ? C0 : a.fin_C1, cc = first ? c : c - C0;
    // the partner is a plain map of the OUTPUT's size (<= 16 parts); behind a CONV_T2 launch g is the input raster and the
    // host has checked that the output raster's parts are four times g's, size for size (rgfm_host.h: up_parts_match)
    const int np_total = first ? nparts0 : (t2 ? 4 * g.nparts : g.nparts);
    // The channel's partials are fetched before any is used (one memory round trip).  The first source was
    // written during THIS launch, possibly by a CU of another XCD (whose L2 is not coherent with ours for plain
    // accesses): agent-scope atomic loads go to the coherence point.
    unsigned long long bits[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const float* sp = st + (((size_t)b * np_total + (p < np_total ? p : 0)) * cs + cc) * 2;
      bits[p] = first ? __hip_atomic_load(reinterpret_cast<const unsigned long long*>(sp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : *reinterpret_cast<const unsigned long long*>(sp);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      // part sizes: a CONV_T2 output is four parity classes of the input raster g; a partner map of a stride-1 /
      // upsampling conv has the output raster g itself
      const double np = p < np_total ? (double)geom_part_count(g, t2 ? p % g.nparts : p) : 0.0;
      const double mp = (double)__uint_as_float((unsigned)(bits[p] & 0xffffffffull));
      const double qp = (double)__uint_as_float((unsigned)(bits[p] >> 32));
      n += np;
      s1 += np * mp;
      s2 += np > 0.0 ? qp + np * mp * mp : 0.0;
    }
  }
  n = sub_sum(n), s1 = sub_sum(s1), s2 = sub_sum(s2);
  const double mean = s1 / n;
  const double var = (s2 - n * mean * mean) / n;
  const float gm = (float)mean;
  const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
#pragma unroll 1
  for (int c = gi * cpg + sub; c < (gi + 1) * cpg; c += 8) {
    const float sc = rstd * a.fin_gamma[c];
    float2 o;
    o.x = sc;
    o.y = a.fin_beta[c] - gm * sc;
    *reinterpret_cast<float2*>(a.fin_ab + ((size_t)b * C + c) * 2) = o;
  }
}

// Statistics store of a conv epilogue: plain, or -- when the finalizing wave of this same launch will read it
// back (possibly from a CU of another XCD, whose L2 is not coherent with ours for plain accesses) -- an
// agent-scope atomic store that goes to the coherence point.
__device__ __forceinline__ void store_stats(const ConvArgs& a, float* sp, float mean, float m2) {
  if (a.fin_ab) {
    static_assert(sizeof(unsigned long long) == 2 * sizeof(float), "(mean, M2) pair is published as one 8-byte store");
    // slots are float2-indexed from a 256-byte-aligned workspace region (host: Bump::f), so sp is 8-byte aligned
    const unsigned long long bits = (unsigned long long)__float_as_uint(mean) | ((unsigned long long)__float_as_uint(m2) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(sp), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    float2 st;
    st.x = mean, st.y = m2;
    *reinterpret_cast<float2*>(sp) = st;
  }
}

// Called by every wave of a conv after its statistics stores (wave-uniform arguments): counts the wave in and,
// if it is the last one of sample b, finalizes.  No agent-scope fence: on this multi-XCD part it writes back /
// invalidates the whole L2 (measured: the conv class dropped from 152 to 92 TFLOP/s); statistics and counter
// use agent-scope atomic accesses instead, ordered by waiting for the stores' acknowledgement.
__device__ __forceinline__ void fin_arrive(const ConvArgs& a, int b, int lane, int nparts0, bool t2) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned last = 0u;
  if (lane == 0)
    last = (__hip_atomic_fetch_add(a.fin_counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
            (unsigned)(a.fin_expected - 1)) ? 1u : 0u;
  last = __shfl(last, 0);
  // the statistics loads of fin_sample must not be hoisted above the counter's result by the compiler
  asm volatile("" ::: "memory");
  if (last) {
    fin_sample(a, b, lane, nparts0, t2);
    if (lane == 0) __hip_atomic_store(a.fin_counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Low side of the raw staging path: S_A a is held as fp16(S_A a) + fp16(remainder); the remainder plane is an fp16
// subnormal below |S_A a| = 2^-3, so the pair keeps 22 significant bits only above that and has an ABSOLUTE error floor
// of 2^-25 / S_A below.  A producer's wave block whose largest |value| is under HX_SMALL = 2^-8 (S_A a < 2^-4: every
// element already loses bits) raises flag bit 1; anything larger keeps the error of every element below
// 2^-25 / (S_A HX_SMALL) = 2^-21 of the block's maximum.
constexpr float HX_SMALL = 0.00390625f;
__device__ __forceinline__ void hx_small_flag(unsigned* flag, float m) {  // m: this lane's max |output|; wave-uniform call
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (m > 0.f && m < HX_SMALL && (threadIdx.x & 63) == 0) atomicOr(flag, 2u);
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
