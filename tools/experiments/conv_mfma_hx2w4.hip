// conv_mfma_hx2w4.hip -- the four-wave cut of the Winograd kernel (round 4; measured and rejected, kept for its numbers).
// Built by tools/kbench only:  hipcc ... -DRGFM_KB_HX2W4 conv_bench.hip;  conv_bench S Cin Cout 0 res B hx2w  with RGFM_HX2W_CUT=2.
// kbench, B = 512, statistics prologue (profiles/r04_kbench/hx2w4_vs_hx2w.txt): 16x16 128 -> 128 150.6 us against 123.6 (eight-wave
// cut) and 116.7 (direct); 256 -> 128 247.4 / 201.5 / 203.9; 32x32 192 -> 64 414.8 / 331.3 / 351.6.  hipcc does interleave the
// MFMAs with the transform's vector-ALU work (1 MFMA : 7 VALU), 198 VGPRs + 256 AGPRs, no scratch -- but ONE wave per SIMD has
// nobody to hide its LDS round trips, its exp / rcp staging phase and the barriers behind.
// (included behind conv_mfma_hx2w.hip: uses its HX2W_RREC, packed weight image and host helpers)
namespace rgfm {

// ONE wave per SIMD with the 512-register budget that buys: the accumulators of FOUR positions (4 x 2 x 2 blocks of 32x32 =
// 256 registers, in the accumulation half of the file) beside 256 ordinary registers, so the input transform of chunk c + 1
// can sit in the SAME instruction stream as the MFMAs of chunk c (V double-buffered) -- the overlap that two waves per SIMD
// in lock-step phases do not give.  Same arithmetic, images, tile and epilogue as the eight-wave cut; bit-identical to it.
template <int WL2>
__global__ __launch_bounds__(256, 1) void conv_mfma_hx2w4_kernel(const ConvArgs a, const int num_tiles) {
  constexpr int W = 1 << WL2, TH = 256 / W, TX = W / 2, WR = W + 2, HR = TH + 2, HALO = HR * WR;
  constexpr int PW = WR / 2, PH = HR / 2, PSZ = PW * PH;
  constexpr int RBYTES = ((4 * PSZ * HX2W_RREC + 1023) / 1024) * 1024;
  constexpr int VBYTES = 16 * 64 * HRW;
  constexpr int MAXIT = (HALO * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smw[];
  // K loop: [R][V 0][V 1][table]; epilogue: [E 128 KB][S 16 KB]
  char* const sV = smw + RBYTES;
  float* const sTab = reinterpret_cast<float*>(smw + RBYTES + 2 * VBYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int l31 = lane & 31, hp = lane >> 5;
  const TileGeom g = a.g;
  const int H = g.H;
  const int tile = (int)blockIdx.x;
  const int b0 = tile / g.tps, row0 = (tile - b0 * g.tps) * TH;
  const int cb = (int)blockIdx.y;
  const int cin = a.C0 + a.C1, nch = cin / KC;
  (void)num_tiles;

  // ---- staging items (halo pixel, 4 channels)
  const int q4 = tid & 3;
  int poff[MAXIT], rdst[MAXIT];
  unsigned okmask = 0u;
#pragma unroll
  for (int j = 0; j < MAXIT; ++j) {
    const int it = tid + 256 * j;
    poff[j] = 0, rdst[j] = 4 * PSZ * HX2W_RREC + q4 * 16;
    if (it < HALO * 4) {
      const int hpx = it >> 2;
      const int hy = hpx / WR, hx = hpx - hy * WR;
      const int y = row0 + hy - 1, x = hx - 1;
      const int rec = ((hy & 1) * 2 + (hx & 1)) * PSZ + (hy >> 1) * PW + (hx >> 1);
      rdst[j] = rec * HX2W_RREC + q4 * 16;
      if (y >= 0 && y < H && x >= 0 && x < W) {
        okmask |= 1u << j;
        poff[j] = (b0 * H + y) * W + x;
      }
    }
  }
  static_assert(RBYTES >= 4 * PSZ * HX2W_RREC + 64, "room for the trash record");
  f32x4 ra[MAXIT];
  float hmax = 0.f;
  auto issue_a = [&](int c) {
    const int ch0 = c * KC;
    const bool first = ch0 < a.C0;
    const float* src = first ? a.in0 + ch0 : a.in1 + (ch0 - a.C0);
    const unsigned cs = (unsigned)(first ? a.C0 : a.C1);
#pragma unroll
    for (int j = 0; j < MAXIT; ++j)
      ra[j] = *(const hx_gf32x4*)(src + (size_t)(__umul24((unsigned)poff[j], cs) + (unsigned)(q4 * 4)));
  };
  issue_a(0);

  // ---- scale / shift table (as the eight-wave cut; up to four waves)
  if (a.gn_stats0) {
    const int gn_cpg = cin >> 3;
    const int gn_wsh = gn_cpg <= 8 ? 0 : (gn_cpg <= 16 ? 1 : 2);  // log2 of the waves that take part (<= 4 here: up to two channels per lane)
    if (wave < (1 << gn_wsh)) {
      const int gn_lpg = 8 << gn_wsh;
      const int gn_gi = wave * (8 >> gn_wsh) + (lane >> (3 + gn_wsh)), gn_sub = lane & (gn_lpg - 1);
      const int gn_kmax = (gn_cpg + gn_lpg - 1) / gn_lpg;
      float gam[4], bet[4];
      double n = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 1
      for (int k = 0; k < gn_kmax; ++k) {
        const int c = gn_gi * gn_cpg + gn_sub + gn_lpg * k;
        const bool have = gn_sub + gn_lpg * k < gn_cpg;
        const bool first = !have || c < a.C0;
        const float* st = first ? a.gn_stats0 : a.gn_stats1;
        const int cs = first ? a.C0 : a.C1, cc = have ? (first ? c : c - a.C0) : 0;
        const int npt = first ? a.gn_nparts0 : a.gn_g.nparts;
        float2 gv[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) gv[p] = *reinterpret_cast<const float2*>(st + (((size_t)b0 * npt + (p < npt ? p : 0)) * cs + cc) * 2);
        const float g_ = a.gn_gamma[have ? c : 0], b_ = a.gn_beta[have ? c : 0];
        if (k == 0) gam[0] = g_, bet[0] = b_;
        else if (k == 1) gam[1] = g_, bet[1] = b_;
        else if (k == 2) gam[2] = g_, bet[2] = b_;
        else gam[3] = g_, bet[3] = b_;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const double np = (have && p < npt) ? (double)geom_part_count(a.gn_g, p % a.gn_g.nparts) : 0.0;
          const double mp = (double)gv[p].x;
          n += np;
          s1 += np * mp;
          s2 += np > 0.0 ? (double)gv[p].y + np * mp * mp : 0.0;
        }
      }
      for (int o = 1; o < gn_lpg; o <<= 1) n += __shfl_xor(n, o), s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
      const double mean = n > 0.0 ? s1 / n : 0.0;
      const double var = n > 0.0 ? s2 / n - mean * mean : 0.0;
      const float gm = (float)mean;
      const float rstd = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (gn_sub + gn_lpg * k < gn_cpg) {
          const float sc = rstd * gam[k];
          float2 o;
          o.x = HX_SA * sc, o.y = HX_SA * (bet[k] - gm * sc);
          *reinterpret_cast<float2*>(sTab + (gn_gi * gn_cpg + gn_sub + gn_lpg * k) * 2) = o;
        }
    }
  } else {
    for (int c = tid; c < cin; c += 256) {
      const float2 e = *reinterpret_cast<const float2*>(a.ab + ((size_t)b0 * cin + c) * 2);
      float2 o;
      o.x = HX_SA * e.x, o.y = HX_SA * e.y;
      *reinterpret_cast<float2*>(sTab + 2 * c) = o;
    }
  }

  auto commit_a = [&](int c) {  // -> R
    const char* ep = reinterpret_cast<const char*>(sTab) + (c * KC + 4 * q4) * 8;
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(ep), e1 = *reinterpret_cast<const f32x4*>(ep + 16);
#pragma unroll
    for (int j = 0; j < MAXIT; ++j) {
      f32x4 v = ra[j];
      v.x = silu_scaled(fmaf(e0.x, v.x, e0.y));
      v.y = silu_scaled(fmaf(e0.z, v.y, e0.w));
      v.z = silu_scaled(fmaf(e1.x, v.z, e1.y));
      v.w = silu_scaled(fmaf(e1.z, v.w, e1.w));
      const float keep = ((okmask >> j) & 1u) ? 1.f : 0.f;
      v = v * keep;
      hmax = hx_absmax3(v.x, v.y, hmax);
      hmax = hx_absmax3(v.z, v.w, hmax);
      *reinterpret_cast<f32x4*>(smw + rdst[j]) = v;
    }
  };

  // ---- transform item: (tile tt, channel quad tq = wave), the whole 4x4
  const int tt = lane, tq = wave;
  const int tty = tt / TX, ttx = tt - tty * TX;
  const int rbase = (tty * PW + ttx) * HX2W_RREC + tq * 16;
  const int vkey = (tt >> 2) & 3;
  const int vdst_h = tt * HRW + (((tq >> 1) ^ vkey) & 3) * 16 + (tq & 1) * 8;
  const int vdst_l = tt * HRW + (((2 + (tq >> 1)) ^ vkey) & 3) * 16 + (tq & 1) * 8;
  f32x4 tcol[4][4];  // B^T d of the chunk being transformed: [row][column]
  auto transform_cols = [&]() {
    auto rd = [&](int i, int j) -> f32x4 {
      const int off = (((i & 1) * 2 + (j & 1)) * PSZ + (i >> 1) * PW + (j >> 1)) * HX2W_RREC;
      return *reinterpret_cast<const f32x4*>(smw + rbase + off);
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 d0 = rd(0, j), d1 = rd(1, j), d2 = rd(2, j), d3 = rd(3, j);
      tcol[0][j] = d0 - d2, tcol[1][j] = d1 + d2, tcol[2][j] = d2 - d1, tcol[3][j] = d1 - d3;
    }
  };
  auto transform_row = [&](int r, char* vb) {  // positions 4 r .. 4 r + 3 -> V buffer vb
    const f32x4* t = tcol[r];
    const f32x4 v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      unsigned h0, l0, h1, l1;
      hsplit2(v[cc].x, v[cc].y, h0, l0);
      hsplit2(v[cc].z, v[cc].w, h1, l1);
      const hx_u32x2 ph = {h0, h1}, pl = {l0, l1};
      *reinterpret_cast<hx_u32x2*>(vb + vdst_h + (r * 4 + cc) * (64 * HRW)) = ph;
      *reinterpret_cast<hx_u32x2*>(vb + vdst_l + (r * 4 + cc) * (64 * HRW)) = pl;
    }
  };

  // ---- fragments
  int aofs[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) aofs[mt][pl] = (32 * mt + l31) * HRW + (((2 * pl + hp) ^ ((l31 >> 2) & 3)) & 3) * 16;
  const char* const wbase = reinterpret_cast<const char*>(a.wpkw) + (size_t)cb * nch * (16 * 4096) + (size_t)(4 * wave) * 4096 + lane * 16;
  f16x8 bfr[4][2][2];  // [position][nt][plane]
  auto issue_b = [&](int c) {
    const char* p = wbase + (size_t)c * (16 * 4096);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          bfr[pi][nt][pl] = __builtin_bit_cast(f16x8, *(const hx_gf32x4*)(p + pi * 4096 + nt * 2048 + pl * 1024));
  };
  f32x16 acc[4][2][2];
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[pi][mt][nt][r] = 0.f;
  auto multiply_pos = [&](int pi, const char* vb) {
    const char* vp = vb + (4 * wave + pi) * (64 * HRW);
    f16x8 af[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[mt][pl] = *reinterpret_cast<const f16x8*>(vp + aofs[mt][pl]);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};  // a_l w_h, a_h w_l, a_h w_h
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[pi][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][PA[q]], bfr[pi][nt][PB[q]], acc[pi][mt][nt], 0, 0, 0);
  };

  // ---- the K loop.  P: the MFMAs of chunk c (V[c & 1]) with the transform of chunk c + 1 (R -> V[(c + 1) & 1]) between them
  // | barrier |  Q: chunk c + 2's halo -> GroupNorm + SiLU -> R, chunk c + 3's requested, chunk c + 1's weight fragments
  // requested | barrier.
  __syncthreads();  // the table
  commit_a(0);
  issue_a(nch > 1 ? 1 : 0);
  __syncthreads();
  transform_cols();
#pragma unroll
  for (int r = 0; r < 4; ++r) transform_row(r, sV);
  __syncthreads();
  if (nch > 1) commit_a(1);
  issue_a(nch > 2 ? 2 : nch - 1);
  issue_b(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nch; ++c) {
    const char* vcur = sV + (c & 1) * VBYTES;
    char* vnext = sV + ((c + 1) & 1) * VBYTES;
    // (behind the last chunk the transform runs once more on R's stale image into the buffer nobody reads: one code path)
    transform_cols();
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
      multiply_pos(pi, vcur);
      transform_row(pi, vnext);
    }
    __syncthreads();
    if (c + 2 < nch) commit_a(c + 2);
    issue_a(c + 3 < nch ? c + 3 : nch - 1);
    issue_b(c + 1 < nch ? c + 1 : c);
    __syncthreads();
  }
  if (!(4.f * hmax < HX_BIG)) atomicOr(a.range_flag, 1u);

  // ---------------------------------------------------------------- epilogue (as the eight-wave cut; two items per thread)
  const float qinv = a.hqw[1];
  float* const sE = reinterpret_cast<float*>(smw);                      // [16][64][32]
  float* const sS = reinterpret_cast<float*>(smw + 128 * 1024);         // [64][32][2]
  const bool sample_ok = b0 < a.B;
  const int ecq = tid & 7;
  auto pass = [&](auto nt_tag) {
    constexpr int nt = decltype(nt_tag)::value;
    __syncthreads();
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float* e = sE + ((size_t)(4 * wave + pi) * 64 + 32 * mt) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) e[((r & 3) + 8 * (r >> 2) + 4 * hp) * 32] = acc[pi][mt][nt][r];
      }
    __syncthreads();
    const int c = cb * 64 + nt * 32 + 4 * ecq;
    f32x4 add = *reinterpret_cast<const f32x4*>(a.bias + c);
    if (a.temb && sample_ok)
      add += *reinterpret_cast<const f32x4*>(a.temb + ((size_t)(a.temb_per_row ? b0 : 0) + (a.step_ptr ? (size_t)*a.step_ptr : 0)) * a.temb_stride + c);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int et = (tid >> 3) + 32 * k;
      const int ety = et / TX, etx = et - ety * TX;
      const size_t pix = ((size_t)b0 * H + row0 + 2 * ety) * W + 2 * etx;
      f32x4 mm[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) mm[p] = *reinterpret_cast<const f32x4*>(sE + ((size_t)(p * 64 + et) * 32 + 4 * ecq));
      f32x4 s0[4], s1[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        s0[cc] = mm[cc] + mm[4 + cc] + mm[8 + cc];
        s1[cc] = mm[4 + cc] - mm[8 + cc] - mm[12 + cc];
      }
      f32x4 y[4] = {s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[i] = y[i] * qinv + add;
        const size_t o = (pix + (i >> 1) * W + (i & 1)) * a.Cout + c;
        if (a.res_mode == 1 && sample_ok) y[i] += *reinterpret_cast<const f32x4*>(a.res0 + o);
        if (sample_ok) *reinterpret_cast<f32x4*>(a.out + o) = y[i];
      }
      if (a.small_check && a.range_flag && sample_ok) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) m = hx_absmax3(y[i].x, y[i].y, hx_absmax3(y[i].z, y[i].w, m));
        hx_small_flag(a.range_flag, m);
      }
      const f32x4 mean = ((y[0] + y[1]) + (y[2] + y[3])) * 0.25f;
      f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) m2 += (y[i] - mean) * (y[i] - mean);
      const f32x4 lo = {mean.x, m2.x, mean.y, m2.y}, hi = {mean.z, m2.z, mean.w, m2.w};
      float* sp = sS + ((size_t)et * 32 + 4 * ecq) * 2;
      *reinterpret_cast<f32x4*>(sp) = lo;
      *reinterpret_cast<f32x4*>(sp + 4) = hi;
    }
    __syncthreads();
    if (tid < 128 && a.stats_out && sample_ok) {
      const int part = tid >> 5, co = tid & 31;
      double n = 0.0, mean = 0.0, m2 = 0.0;
      for (int k = 0; k < 16; ++k) {
        const float2 v = *reinterpret_cast<const float2*>(sS + ((size_t)(16 * part + k) * 32 + co) * 2);
        const double d = (double)v.x - mean, nn = n + 4.0;
        mean += d * (4.0 / nn);
        m2 += (double)v.y + d * d * (n * 4.0 / nn);
        n = nn;
      }
      const int gpart = (tile - b0 * g.tps) * 4 + part;
      float2 o;
      o.x = (float)mean, o.y = (float)m2;
      *reinterpret_cast<float2*>(a.stats_out + (((size_t)b0 * g.nparts + gpart) * a.Cout + cb * 64 + nt * 32 + co) * 2) = o;
    }
  };
  pass(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 1>{});
}


static size_t hx2w4_lds_bytes(const ConvArgs& a) {
  const int W = a.g.W, TH = 256 / W;
  const int psz = ((W + 2) / 2) * ((TH + 2) / 2);
  const size_t rbytes = (((size_t)4 * psz * HX2W_RREC + 1023) / 1024) * 1024;
  const size_t kloop = rbytes + (size_t)2 * 16 * 64 * HRW + (size_t)(a.C0 + a.C1) * 8, epi = (size_t)128 * 1024 + 64 * 32 * 8;
  return kloop > epi ? kloop : epi;
}
int conv_hx2w4_init() {
  int rc = 0;
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2w4_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_hx2w4_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return rc;
}
void launch_conv_hx2w4(const ConvArgs& a, hipStream_t s) {
  const int tiles = geom_num_tiles(a.g, a.B);
  const dim3 grid(tiles, a.Cout / 64);
  const size_t lds4 = hx2w4_lds_bytes(a);
  if (a.g.W == 16) hipLaunchKernelGGL((conv_mfma_hx2w4_kernel<4>), grid, dim3(256), lds4, s, a, tiles);
  else hipLaunchKernelGGL((conv_mfma_hx2w4_kernel<5>), grid, dim3(256), lds4, s, a, tiles);
}

}  // namespace rgfm
