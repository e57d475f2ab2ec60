// conv_hx2_common.h -- device helpers shared by conv_mfma_hx2.hip and conv_mfma_hx2p.hip (two-plane fp16 split,
// LDS record layout).  See the header comment of conv_mfma_hx2.hip for the arithmetic.
#pragma once
#include "rgfm_device.h"

namespace rgfm {

typedef __attribute__((address_space(1))) f32x4 hx_gf32x4;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float hx_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned hx_u32x2 __attribute__((ext_vector_type(2)));

constexpr float HX_SA = 16.f;         // activation scale S_A
constexpr float HX_LIMIT = 32768.f;   // |a'| at or above this raises the range flag

// 2-way fp16 split of two (already scaled) floats: planes h, l as packed fp16 pairs.  h = v_cvt_pk_f16_f32 (RNE); the
// remainder a - float(h) in ONE v_fma_mix_f32 per value (the fp16 half is an operand: no v_cvt_f32_f16, and no packed
// v_pk_add_f32, which hipcc makes of the vector form and which is an anti-lever beside MFMAs); exact either way, so the
// planes are bit for bit those of the conversion + subtraction form.  The staging path is VALU-issue-bound in the MFMA
// shadow (tools/ubench/mfma_valu_overlap.hip: ~5.5 vector instructions per MFMA slot from the other waves of a SIMD):
// every instruction per item counts.
__device__ __forceinline__ void hsplit2(float a, float b, unsigned& ph, unsigned& pl) {
  hx_f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  ph = __builtin_bit_cast(unsigned, h);
  float la, lb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(ph), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(ph), "v"(b));
  const hx_f32x2 lv = {la, lb};
  pl = __builtin_bit_cast(unsigned, __builtin_convertvector(lv, f16x2));
}

// High side of the staging path: plane h = fp16(S_A a) reaches 32768 (fp16's top binade, where the remainder plane stops
// being exact -- and 65520 rounds to inf) from |S_A a| >= 32760 (round to nearest even).  The running max is taken on
// the fp32 values before the split: v_max3_f32 with |.| source modifiers, one instruction per two values.  A NaN is not
// caught here and need not be: it propagates into every output it touches, as in the reference.
constexpr float HX_BIG = 32760.f;
__device__ __forceinline__ float hx_absmax3(float a, float b, float m) {
  return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), m);
}

// S_A silu(z) from z' = S_A z (v_exp_f32 + v_rcp_f32 as silu_fast)
__device__ __forceinline__ float silu_scaled(float zs) {
  const float e = __builtin_amdgcn_exp2f(zs * (-1.44269504088896341f / HX_SA));
  return zs * __builtin_amdgcn_rcpf(1.0f + e);
}

// (HX_SMALL / hx_small_flag, the low side of the raw staging path, live in rgfm_device.h: the bf16x3 and fp32 kernels
// check it too when they produce a tensor that a two-plane conv stages raw)

// ---- "P format": an activation ALREADY normalised, activated and split, as the consumer's MFMAs read it
// (conv_mfma_hx2d.hip stages it by LDS-DMA only).  One (pixel, 16-channel chunk) = 64 B:
//   [plane h: channels 0 .. 15 | plane l: channels 0 .. 15]  fp16 of S_A silu(scale x + shift),   [B][H][W][C / 16][64 B]
// -- the bytes of the fp32 map it replaces.  Written by the producing conv's epilogue where a workgroup owns whole
// (sample, GroupNorm group) sets (ConvArgs::pout), or by launch_hx_presplit.

// exchange between the two lanes of a channel pair (2 k, 2 k + 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float hx_swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// GroupNorm scale / shift (x S_A) of this lane's channel from its (mean, M2) over n pixels, when all `cpg` channels of
// its group sit in consecutive lanes of ONE wave (cpg a power of two <= 32; both lane halves hold the same values).
// The consumer-side prologue's formula (fp64: N, sum n mean, sum M2 + n mean^2 -> mean, var), here with one channel per
// lane and a butterfly over the group's lanes.
// (hx_group_affine_s: from the channel's sums s1 = sum n_p mean_p, s2 = sum M2_p + n_p mean_p^2 over its parts, n pixels in all)
__device__ __forceinline__ void hx_group_affine_s(double s1, double s2, double n, int cpg, float gamma, float beta, float& sc, float& sh) {
  for (int o = 1; o < cpg; o <<= 1) s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
  const double N = n * (double)cpg, mean = s1 / N, var = s2 / N - mean * mean;
  const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
  const float a = rstd * gamma;
  sc = HX_SA * a;
  sh = HX_SA * (beta - (float)mean * a);
}
__device__ __forceinline__ void hx_group_affine(float mean_c, float m2_c, float n, int cpg, float gamma, float beta, float& sc, float& sh) {
  double s1 = (double)n * (double)mean_c, s2 = (double)m2_c + (double)n * (double)mean_c * (double)mean_c;
  for (int o = 1; o < cpg; o <<= 1) s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
  const double N = (double)n * (double)cpg, mean = s1 / N, var = s2 / N - mean * mean;
  const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
  const float a = rstd * gamma;
  sc = HX_SA * a;
  sh = HX_SA * (beta - (float)mean * a);
}

// One wave's 64 pixels x 32 channels (accumulator layout: channel = lane & 31, acc<mt>[r] = pixel 32 mt + (r & 3) +
// 8 (r >> 2) + 4 (lane >> 5) of the block) as P records: S_A silu(sc x + sh), split, two channels per lane -- the lanes of a
// channel pair trade one value of every pixel pair, so each holds (2 k, 2 k + 1) of ONE pixel -- and one 4-byte store per
// plane.  rec0: the record of (pixel 0 of the block, this lane's chunk); pstride: bytes per pixel (C / 16 x 64).
// Returns the largest |S_A silu(.)| seen (the caller raises range-flag bit 0 on >= HX_BIG, as the staging paths do).
__device__ __forceinline__ float hx_p_emit(const f32x16& acc0, const f32x16& acc1, float sc, float sh, char* rec0, unsigned pstride, int l31, int hp) {
  const bool odd = (l31 & 1) != 0;
  char* const base = rec0 + ((l31 & 15) >> 1) * 4 + (size_t)(4 * hp + (odd ? 1 : 0)) * pstride;
  float m = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x16& am = mt ? acc1 : acc0;
      const float o0 = silu_scaled(fmaf(sc, am[r], sh)), o1 = silu_scaled(fmaf(sc, am[r + 1], sh));
      const float recv = hx_swap1(odd ? o0 : o1);
      const float lo = odd ? recv : o0, hi = odd ? o1 : recv;
      m = hx_absmax3(lo, hi, m);
      unsigned ph, pl;
      hsplit2(lo, hi, ph, pl);
      char* const q = base + (size_t)(32 * mt + (r & 3) + 8 * (r >> 2)) * pstride;
      *reinterpret_cast<unsigned*>(q) = ph;
      *reinterpret_cast<unsigned*>(q + 32) = pl;
    }
  return m;
}

constexpr int HRW = 64;  // bytes per LDS record: [plane h | plane l] x 16 fp16
// byte offset of 16-byte slot (plane, half) inside record `rec`
__device__ __forceinline__ int hswz(int rec, int plane, int half) { return (((2 * plane + half) ^ (rec >> 2)) & 3) * 16; }

}  // namespace rgfm
