// conv_mfma_hx2d16.hip -- EXPERIMENT (round 4, rejected; not built into the library).  Paste behind conv_mfma_hx2d_kernel in
// csrc/conv_mfma_hx2d.hip, raise its dynamic-LDS limit in conv_hx2d_init and launch it from launch_conv_hx2d to reproduce
// profiles/r04_kbench/hx2d_mfma_16x16x32.txt (tools/kbench: conv_bench ... hx2d with RGFM_HX2D=4 in that build).
// Results are correct (same error against the fp32 kernel as the 32x32x16 cuts); it is 7 - 10 % slower at 512 rows and
// 20 - 35 % slower at 32 rows than conv_mfma_hx2d_kernel.
// ---------------------------------------------------------------- the same conv on v_mfma_f32_16x16x32_f16
// conv_mfma_hx2d16_kernel: the eight-wave kernel above with its matrix work re-cut for the 16x16x32 MFMA.  On this part a
// register-only loop of v_mfma_f32_16x16x32_f16 sustains 1.17 x the FLOP/s of the 32x32x16 loop at equal cycles per FLOP
// (tools/ubench/mfma_shapes.hip: 1982 vs 1694 TFLOP/s -- the chip holds a higher clock on that shape; the guide's "DVFS
// give-back" item 7), and the clock is what separates a full launch's chunk (2.9 us) from an under-filled one's (1.9 us).
//   * K = 32 per instruction = the TWO planes of a 16-channel chunk as ONE operand.  A fragment = the pixel's LDS record in
//     its natural order [a_h | a_l] (lane (g = l >> 4, c = l & 15): pixel c of the 16-pixel block, logical slot g); B
//     fragment = the weight record read as [w_l | w_h].  One instruction then gives a_h w_l + a_l w_h of a tap.
//   * a_h w_h of TWO taps shares one instruction: v_permlane32_swap on the two taps' A fragments leaves [a_h(t) | a_h(t')] in
//     one of them, the same swap on their B fragments [w_h(t) | w_h(t')] -- no further LDS reads.  The chunk's ninth tap
//     multiplies [a_h | a_l] by [w_h | 0].  Per chunk and 16x16 block: 9 + 4 + 1 instructions of 16 cycles (13.5 would be
//     the FLOPs' worth: +3.7 %); fragment reads per MFMA cycle as before.
//   * A wave's 64 pixels x 32 channels = 4 x 2 blocks of 16 x 16, 32 accumulator registers as before, but a lane now holds
//     2 channels x 16 pixels (channel 16 sn + c, pixel 16 sm + 4 g + r): prologue and epilogue are written for that layout.
// Products are summed in another order than in the 32x32x16 kernels (fp32-class all the same); every launch of a layer
// takes the same kernel whatever its batch, so a row's bits still do not depend on it.  No fused skip yet.
__device__ __forceinline__ void hx_swap32(f16x8& x, f16x8& y) {  // x.hi <-> y.lo (halves of 32 lanes)
  typedef unsigned sw_u32x4 __attribute__((ext_vector_type(4)));
  sw_u32x4 a = __builtin_bit_cast(sw_u32x4, x), b = __builtin_bit_cast(sw_u32x4, y);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto r = __builtin_amdgcn_permlane32_swap(a[i], b[i], false, false);
    a[i] = r[0], b[i] = r[1];
  }
  x = __builtin_bit_cast(f16x8, a), y = __builtin_bit_cast(f16x8, b);
}

template <int W>
__global__ __launch_bounds__(512, 1) void conv_mfma_hx2d16_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int SPT = W == 8 ? 4 : 1, H = W, WR = W + 2, HR = H + 2;
  constexpr int PREC = WR * HR, HALO = SPT * PREC;
  constexpr int NPC = (HALO + 15) / 16;
  constexpr int ABYTES = NPC * 1024;
  constexpr int NW = 8, CB = 64;
  constexpr int TAPB = CB * HRW, CHB = 9 * TAPB, PPT = TAPB / 1024;
  constexpr int NPH = (NPC + NW - 1) / NW;
  constexpr int RPS = 64 / W;
  constexpr int SM_OFF = (16 / W) * WR * HRW;  // the next 16 pixels of a segment: two rows (W = 8) / one row (W = 16) down
  constexpr int SWZ = 1;
  extern __shared__ __attribute__((aligned(16))) char smd16[];
  char* const sA = smd16;
  char* const sB = smd16 + 2 * ABYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, seg = wave & 3;
  const int g4 = lane >> 4, c16 = lane & 15;
  const int tile = blockIdx.x, cb = blockIdx.y;
  const int b0 = tile * SPT;
  const int nmain = a.C0 / KC;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  unsigned soff[NPH];
  bool sval[NPH];
#pragma unroll
  for (int j = 0; j < NPH; ++j) {
    const int pq = wave + NW * j;
    const int pc = pq < NPC ? pq : pq - NPC;
    const int rec = pc * 16 + (lane >> 2);
    const int s = rec / PREC, rr = rec - s * PREC;
    const int hy = rr / WR, hx = rr - hy * WR;
    const int y = hy - 1, x = hx - 1;
    const int logical = ((lane & 3) ^ (hx >> SWZ)) & 3;
    sval[j] = rec < HALO && y >= 0 && y < H && x >= 0 && x < W && b0 + s < a.B;
    soff[j] = sval[j] ? (unsigned)((((b0 + s) * H + y) * W + x) * nmain) * 64u + (unsigned)logical * 16u : 0u;
  }
  const char* const pin = reinterpret_cast<const char*>(a.pin0);
  const char* const zeros = reinterpret_cast<const char*>(a.zeros) + (lane & 3) * 16;
  const unsigned sA_lds = (unsigned)(size_t)sA, sB_lds = (unsigned)(size_t)sB;
  auto dma = [&](const char* gsrc, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  const bool nb128 = (a.Cout & 127) == 0;
  const int TAPS = nb128 ? 2 * TAPB : TAPB;
  const int wblk = nb128 ? cb >> 1 : cb, whalf = nb128 ? (cb & 1) * TAPB : 0;
  const char* const wpk = reinterpret_cast<const char*>(a.wpkh) + (size_t)wblk * nmain * 9 * TAPS + whalf;
  constexpr int NPW = 5;
  auto stage = [&](int c) {  // every DMA piece of chunk c: halo into halo buffer c & 1, weights into weight buffer c & 1
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int pq = wave_s + NW * j;
      const int pc = pq < 9 * PPT ? pq : pq - 9 * PPT;
      dma(wpk + (size_t)c * 9 * TAPS + (pc / PPT) * TAPS + (pc % PPT) * 1024 + lane * 16, sB_lds + (unsigned)((c & 1) * CHB + pc * 1024));
    }
#pragma unroll
    for (int j = 0; j < NPH; ++j) {
      const int pq = wave_s + NW * j;
      const int pc = pq < NPC ? pq : pq - NPC;
      dma(sval[j] ? pin + (size_t)soff[j] + (size_t)c * 64 : zeros, sA_lds + (unsigned)((c & 1) * ABYTES + pc * 1024));
    }
  };

  // ---- fragment offsets.  A: pixel c16 of 16-pixel block 0 of this wave's segment at tap column kx, logical slot g4
  // ([h0 h1 l0 l1]: the record's own order); block sm: + sm SM_OFF.  B: channel 16 sn + c16 of this wave's 32, logical
  // slot g4 ^ 2 ([l0 l1 h0 h1]).
  int aofs[3];
  {
    const int r = c16 / W, x = c16 % W;
    const int arec = (SPT == 4 ? seg * PREC : seg * RPS * WR) + r * WR + x;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aofs[kx] = (arec + kx) * HRW + ((g4 ^ (((x + kx) >> SWZ) & 3)) & 3) * 16;
  }
  int bofs[2];
#pragma unroll
  for (int sn = 0; sn < 2; ++sn) {
    const int rec = grp * 32 + sn * 16 + c16;
    bofs[sn] = rec * HRW + (((g4 ^ 2) ^ (rec >> 2)) & 3) * 16;
  }

  stage(0);

  // ---- accumulators: lane holds channels ch0 + 16 sn, pixels 16 sm + 4 g4 + r of its segment
  const float qmain = a.hq[0];
  const int sample = SPT == 4 ? b0 + seg : b0;
  const int part = SPT == 4 ? 0 : seg;
  const int ch0 = cb * CB + grp * 32 + c16;
  const size_t pix0 = (size_t)sample * (H * W) + (SPT == 4 ? 0 : seg * 64);
  f32x4 acc[4][2];
#pragma unroll
  for (int sn = 0; sn < 2; ++sn) {
    const int ch = ch0 + 16 * sn;
    float v = a.bias[ch];
    if (a.temb) v += a.temb[((size_t)(a.temb_per_row ? (sample < a.B ? sample : 0) : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + ch];
    const float add0 = v * qmain;
    if (a.res_mode == 1) {
      const size_t pixr = sample < a.B ? pix0 : 0;
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sm][sn][r] = a.res0[(pixr + 16 * sm + 4 * g4 + r) * a.Cout + ch];
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sm][sn][r] = fmaf(acc[sm][sn][r], qmain, add0);
    } else {
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sm][sn][r] = add0;
    }
  }

  struct Tap {  // one tap's fragments: 4 pixel blocks, 2 channel blocks
    f16x8 a[4], b[2];
  };
  auto ldt = [&](Tap& f, const char* sAc, const char* sBc, int t) {
    const char* pa = sAc + (t / 3) * WR * HRW + aofs[t % 3];
    const char* pb = sBc + t * TAPB;
#pragma unroll
    for (int sm = 0; sm < 4; ++sm) f.a[sm] = *reinterpret_cast<const f16x8*>(pa + sm * SM_OFF);
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) f.b[sn] = *reinterpret_cast<const f16x8*>(pb + bofs[sn]);
  };
  auto mm = [&](const Tap& fa, const Tap& fb) {  // acc[sm][sn] += fa.a[sm] . fb.b[sn]  (K = 32)
#pragma unroll
    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
      for (int sn = 0; sn < 2; ++sn) acc[sm][sn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa.a[sm], fb.b[sn], acc[sm][sn], 0, 0, 0);
  };
  // a pair of taps: a_h w_l + a_l w_h of each, then a_h w_h of both in one instruction per block
  auto pair = [&](Tap& t0, Tap& t1) {
    mm(t0, t0);
    mm(t1, t1);
#pragma unroll
    for (int sm = 0; sm < 4; ++sm) hx_swap32(t0.a[sm], t1.a[sm]);  // t0.a = [a_h(t0) | a_h(t1)]
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) hx_swap32(t0.b[sn], t1.b[sn]);  // t1.b = [w_h(t0) | w_h(t1)]
    mm(t0, t1);
  };

#pragma unroll 1
  for (int c = 0; c < nmain; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < nmain) stage(c + 1);
    const char* sAc = sA + (c & 1) * ABYTES;
    const char* sBc = sB + (c & 1) * CHB;
    Tap p0, p1, q0, q1;
    ldt(p0, sAc, sBc, 0);
    ldt(p1, sAc, sBc, 1);
    ldt(q0, sAc, sBc, 2);
    ldt(q1, sAc, sBc, 3);
    __builtin_amdgcn_sched_barrier(0);
    pair(p0, p1);
    __builtin_amdgcn_sched_barrier(0);
    ldt(p0, sAc, sBc, 4);
    ldt(p1, sAc, sBc, 5);
    __builtin_amdgcn_sched_barrier(0);
    pair(q0, q1);
    __builtin_amdgcn_sched_barrier(0);
    ldt(q0, sAc, sBc, 6);
    ldt(q1, sAc, sBc, 7);
    __builtin_amdgcn_sched_barrier(0);
    pair(p0, p1);
    __builtin_amdgcn_sched_barrier(0);
    ldt(p0, sAc, sBc, 8);
    __builtin_amdgcn_sched_barrier(0);
    pair(q0, q1);
    __builtin_amdgcn_sched_barrier(0);
    // the ninth tap: [a_h | a_l] . [w_l | w_h], then [a_h | a_l] . [w_h | 0]
    mm(p0, p0);
    {
      const f16x8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#pragma unroll
      for (int sn = 0; sn < 2; ++sn) {
        p1.b[sn] = z;
        hx_swap32(p0.b[sn], p1.b[sn]);  // p1.b = [w_h | 0]
      }
    }
    mm(p0, p1);
  }

  // ---- epilogue in the 16x16 layout
  {
    const float qinv = a.hq[1];
#pragma unroll
    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
      for (int sn = 0; sn < 2; ++sn) acc[sm][sn] = acc[sm][sn] * qinv;
  }
  if (sample >= a.B) return;  // (wave-uniform; no barrier follows)
  if (a.small_check && a.range_flag) {
    float m = 0.f;
#pragma unroll
    for (int sm = 0; sm < 4; ++sm)
#pragma unroll
      for (int sn = 0; sn < 2; ++sn)
#pragma unroll
        for (int r = 0; r < 4; r += 2) m = hx_absmax3(acc[sm][sn][r], acc[sm][sn][r + 1], m);
    hx_small_flag(a.range_flag, m);
  }
#pragma unroll
  for (int sm = 0; sm < 4; ++sm)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* op = a.out + (pix0 + 16 * sm + 4 * g4 + r) * a.Cout + ch0;
      op[0] = acc[sm][0][r];
      op[16] = acc[sm][1][r];
    }
  if (a.stats_out) {
#pragma unroll
    for (int sn = 0; sn < 2; ++sn) {
      float s = 0.f;
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[sm][sn][r];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s / 64.f;
      float m2 = 0.f;
#pragma unroll
      for (int sm = 0; sm < 4; ++sm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[sm][sn][r] - mean;
          m2 += d * d;
        }
      m2 += __shfl_xor(m2, 16);
      m2 += __shfl_xor(m2, 32);
      if (g4 == 0) store_stats(a, a.stats_out + (((size_t)sample * a.g.nparts + part) * a.Cout + ch0 + 16 * sn) * 2, mean, m2);
    }
  }
}

