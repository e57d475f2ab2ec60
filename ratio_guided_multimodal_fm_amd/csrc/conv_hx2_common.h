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

constexpr int HRW = 64;  // bytes per LDS record: [plane h | plane l] x 16 fp16
// byte offset of 16-byte slot (plane, half) inside record `rec`
__device__ __forceinline__ int hswz(int rec, int plane, int half) { return (((2 * plane + half) ^ (rec >> 2)) & 3) * 16; }

}  // namespace rgfm
