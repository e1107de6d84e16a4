// conv_common.h — pieces shared by the fused-convolution schedules (conv_mfma.hip, gemm_k1.hip):
// vector types, activation functions, the bf16 hi/lo split and the tile epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;  // also the 128-bit container of an fp16 fragment (bit-cast at the MFMA)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int NPL = 4;  // planes per K chunk (32 channels = one MFMA K)

// D[16][16] += A[16][32] * B[32][16] on 16-bit fragments of plane format FMT (enum rsa_plane_fmt: 0 = bf16, 1 = fp16); same fragment
// layout, same cycles
template <int FMT>
__device__ __forceinline__ f32x4 mfma16(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  if constexpr (FMT == RSA_PF_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Activation CLASS of an epilogue instantiation.  The epilogue is compiled once per class and selected by ONE wave-uniform
// switch per tile: with a per-value runtime switch every one of the 64 value sites of the unrolled epilogue carried the inlined
// libm code of Mish/GELU/SiLU (24k ISA lines per kernel), and each tile streamed ~200 KB of instructions through the 64 KB
// instruction cache even when only LeakyReLU ran (profiles/r01_l_ab_epilogue.txt).
enum { AC_LINEAR = 0 /* none, LeakyReLU, PReLU */, AC_MISH = 1, AC_SILU = 2, AC_GELU = 3, AC_GATE = 4 };

// GELU(x) = x/2 * (1 + erf(x / sqrt 2)) with erf as the rational x * P(x^2) / Q(x^2) on [-4, 4] (the f32 approximation used by the
// tensor libraries; |error| of erf <= 4.5e-7, of GELU <= 2.5e-7 * max(1, |x|)): 17 instructions and no branch, against the two-branch
// erff of the device library (a hidden map costs 64 GELUs per lane and 64-token tile, beside 96 MFMAs).
__device__ __forceinline__ float gelu_fast(float x) {
  const float t = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -4.f, 4.f);
  const float t2 = t * t;
  float pn = -2.72614225801306e-10f;
  pn = fmaf(pn, t2, 2.77068142495902e-08f);
  pn = fmaf(pn, t2, -2.10102402082508e-06f);
  pn = fmaf(pn, t2, -5.69250639462346e-05f);
  pn = fmaf(pn, t2, -7.34990630326855e-04f);
  pn = fmaf(pn, t2, -2.95459980854025e-03f);
  pn = fmaf(pn, t2, -1.60960333262415e-02f);
  float qd = -1.45660718464996e-05f;
  qd = fmaf(qd, t2, -2.13374055278905e-04f);
  qd = fmaf(qd, t2, -1.68282697438203e-03f);
  qd = fmaf(qd, t2, -7.37332916720468e-03f);
  qd = fmaf(qd, t2, -1.42647390514189e-02f);
  const float e = pn * t * __builtin_amdgcn_rcpf(qd);
  const float hx = 0.5f * x;
  return fmaf(hx, e, hx);
}

// e^x and 1/x as one hardware instruction each (v_exp_f32 on x * log2 e, v_rcp_f32: 1 ulp).  The device library's expf and the IEEE
// division are ~15 and ~10 instructions; a 48-channel SPAN layer applies SiLU to 48 values per lane and tile beside 540 MFMAs and
// ran 40 % longer with them than with LeakyReLU (profiles/r02m_*).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

template <int AC>
__device__ __forceinline__ float act_apply(float v, int act, float prm) {
  if (AC == AC_LINEAR) {  // prm = LeakyReLU slope or this channel's PReLU slope
    return (act == RSA_ACT_NONE || v >= 0.f) ? v : v * prm;
  } else if (AC == AC_MISH) {
    // torch: x * tanh(softplus(x)) with softplus threshold 20.  tanh(ln(1+n)) = (n^2+2n)/(n^2+2n+2) for n = e^x, so one exp and
    // one reciprocal replace log1p + tanh (exact algebra)
    // (softplus' threshold: from v = 20 on n(n + 2) = 2.4e17 and the ratio below is 1.0f, so clamping the exponent gives torch's `x` branch
    //  without a compare and a select per value; it also keeps e^v finite)
    const float n = exp_fast(fminf(v, 20.f));
    const float t = n * (n + 2.f);
    return v * (t * __builtin_amdgcn_rcpf(t + 2.f));
  } else if (AC == AC_SILU) {
    return v * __builtin_amdgcn_rcpf(1.f + exp_fast(-v));  // exp -> inf for very negative v: rcp(inf) = 0
  } else if (AC == AC_GELU) {
    return gelu_fast(v);
  }
  return v;
}

__device__ __forceinline__ int act_class(int act) {
  switch (act) {
    case RSA_ACT_MISH:
      return AC_MISH;
    case RSA_ACT_SILU:
      return AC_SILU;
    case RSA_ACT_GELU:
      return AC_GELU;
    case RSA_ACT_SPAB_GATE:
      return AC_GATE;
    default:
      return AC_LINEAR;
  }
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// Workgroup barrier WITHOUT the vmcnt(0) drain that __syncthreads() carries: only this wave's LDS traffic is waited for, so
// weight prefetches and epilogue stores stay in flight across it.  A wave whose LDS-DMA must have landed calls dma_wait() first.
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// (a, b) -> packed 16-bit pair (RNE) of plane format FMT and the pair's rounding residuals, also packed
template <int FMT = 0>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  if constexpr (FMT == RSA_PF_F16) {
    // The values are made opaque first: with a = x * y still visible, the compiler fuses the conversion of the residual into
    // v_fma_mix*_f16 (x * y - hi rounded once from the exact product) while hi itself is v_cvt_pk_f16_f32 of the f32 product -- the two
    // roundings of x * y can differ by an fp16 ulp, and hi + lo is then off by that ulp instead of being a 22-bit value (seen on the
    // Mish / SiLU / gate epilogues: 2.4e-4 on single elements).
    asm("" : "+v"(a), "+v"(b));
    const f16x2 h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const f16x2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    lo = __builtin_bit_cast(uint32_t, l);
  } else {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    const bf16x2 l = {(__bf16)ra, (__bf16)rb};
    lo = __builtin_bit_cast(uint32_t, l);
  }
}
// runtime format (wave-uniform): the generic epilogue
__device__ __forceinline__ void split2_rt(bool f16, float a, float b, uint32_t& hi, uint32_t& lo) {
  if (f16)
    split2<RSA_PF_F16>(a, b, hi, lo);
  else
    split2<RSA_PF_BF16>(a, b, hi, lo);
}
// four channels of a unit half: hi (+ lo) -> f32
template <int FMT>
__device__ __forceinline__ f32x4 widen4(uint2 h, uint2 l) {
  if constexpr (FMT == RSA_PF_F16) {
    const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
    const f16x2 l0 = __builtin_bit_cast(f16x2, l.x), l1 = __builtin_bit_cast(f16x2, l.y);
    return (f32x4){(float)h0[0] + (float)l0[0], (float)h0[1] + (float)l0[1], (float)h1[0] + (float)l1[0], (float)h1[1] + (float)l1[1]};
  } else {
    return (f32x4){__builtin_bit_cast(float, h.x << 16) + __builtin_bit_cast(float, l.x << 16),
                   __builtin_bit_cast(float, h.x & 0xffff0000u) + __builtin_bit_cast(float, l.x & 0xffff0000u),
                   __builtin_bit_cast(float, h.y << 16) + __builtin_bit_cast(float, l.y << 16),
                   __builtin_bit_cast(float, h.y & 0xffff0000u) + __builtin_bit_cast(float, l.y & 0xffff0000u)};
  }
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// lo halves as 8-bit codes (rsa_conv_params.lo8_flags): four channels of a unit half = one dword.  The code of v = hi + lo is the distance
// from f32(hi) to v counted in steps of 32 f32 ulps along the f32 bit patterns (sign-magnitude: both have the sign of v), a signed byte:
//   d = sat16(bits(v) - bits(f32(hi)))     s = low byte of min(sat16(d + 16) >> 5, 127)     v' = as_float(bits(f32(hi)) + (s << 5))
// |lo| is at most half an fp16 ulp = 2^12 f32 ulps, so s covers it and hi + code keep 19 mantissa bits.  The `min` matters for the tie
// (+half an ulp: one step short).  Where fp16 is coarser than 2^13 f32 ulps (hi = 0 or subnormal) d saturates and the code is one of
// 0 .. 127 (hi = 0: d >= 0) or the low byte of a negative number; any of them decodes to within 2^-24 of v, never across zero.
// hi = +-inf: v' not a number or ~3.4e38, the next layer's hi is not finite either -- the fp16 range guard reads hi planes.
template <int BYTE>
__device__ __forceinline__ float lo8_decode(float hf, uint32_t l8) {  // v_bfe_i32 + v_lshl_add_u32
  int s = BYTE == 3 ? (int)l8 >> 24 : __builtin_amdgcn_sbfe(l8, 8 * BYTE, 8);
  asm("" : "+v"(s));  // keeps the shift out of the extraction (else: shift, shift, and, add)
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, hf) + (uint32_t)(s << 5));
}
__device__ __forceinline__ f32x4 widen4_lo8(uint2 h, uint32_t l8) {  // fp16 hi + code -> f32
  const f16x2 h0 = __builtin_bit_cast(f16x2, h.x), h1 = __builtin_bit_cast(f16x2, h.y);
  return (f32x4){lo8_decode<0>((float)h0[0], l8), lo8_decode<1>((float)h0[1], l8), lo8_decode<2>((float)h1[0], l8), lo8_decode<3>((float)h1[1], l8)};
}
typedef __attribute__((ext_vector_type(2))) short s16x2;
// two values -> their codes in the low bytes of two int16 (saturating 32 -> 16 bit pack, then packed 16-bit arithmetic)
__device__ __forceinline__ s16x2 lo8_encode2(float v0, float v1, uint32_t hi01) {
  const f16x2 h = __builtin_bit_cast(f16x2, hi01);
  const int d0 = (int)(__builtin_bit_cast(uint32_t, v0) - __builtin_bit_cast(uint32_t, (float)h[0]));
  const int d1 = (int)(__builtin_bit_cast(uint32_t, v1) - __builtin_bit_cast(uint32_t, (float)h[1]));
  const s16x2 d = __builtin_bit_cast(s16x2, __builtin_amdgcn_cvt_pk_i16(d0, d1)), r = {16, 16}, m = {127, 127};
  return __builtin_elementwise_min(__builtin_elementwise_add_sat(d, r) >> 5, m);
}
// (v0 .. v3) -> fp16 hi pairs and the four codes as one dword
__device__ __forceinline__ void split4_lo8(float v0, float v1, float v2, float v3, uint32_t& hi01, uint32_t& hi23, uint32_t& lo8) {
  asm("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));  // opaque: see split2
  const f16x2 ha = {(_Float16)v0, (_Float16)v1}, hb = {(_Float16)v2, (_Float16)v3};
  hi01 = __builtin_bit_cast(uint32_t, ha);
  hi23 = __builtin_bit_cast(uint32_t, hb);
  asm("" : "+v"(hi01), "+v"(hi23));  // one v_cvt_pk_f16_f32 each, widened again below
  const s16x2 sa = lo8_encode2(v0, v1, hi01), sb = lo8_encode2(v2, v3, hi23);
  lo8 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, sb), __builtin_bit_cast(uint32_t, sa), 0x06040200u);  // the low bytes
}

// Two D fragments of the pixel-tile pair (2k, 2k+1) of one 16-channel tile -> the 16-byte plane units of plane 2*ct + (lg >> 1):
// the even lane group ends up with the full unit of token tile 2k, the odd one with that of 2k+1 (v_permlane16_swap, as the
// convolution epilogue).  Returns the token tile this lane stores.
template <int FMT = 0>
__device__ __forceinline__ void pair_units(const f32x4 a, const f32x4 b, uint4& uh, uint4& ul) {
  uint32_t h[2][2], l[2][2];
  split2<FMT>(a[0], a[1], h[0][0], l[0][0]);
  split2<FMT>(a[2], a[3], h[0][1], l[0][1]);
  split2<FMT>(b[0], b[1], h[1][0], l[1][0]);
  split2<FMT>(b[2], b[3], h[1][1], l[1][1]);
  const u32x2 h0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
  const u32x2 h1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
  const u32x2 l0 = __builtin_amdgcn_permlane16_swap(l[0][0], l[1][0], false, false);
  const u32x2 l1 = __builtin_amdgcn_permlane16_swap(l[0][1], l[1][1], false, false);
  uh = make_uint4(h0.x, h1.x, h0.y, h1.y);
  ul = make_uint4(l0.x, l1.x, l0.y, l1.y);
}

// Geometry of one instantiation.  Output tile = 16 rows x 32 pixels.  A COMPUTE wave owns NPT pixel-tiles (RPW rows x 2 halves of
// 16 pixels) x CTW cout-tiles; one tap costs it 2*NPT LDS fragment reads + 2*CTW weight fragment loads for 3*NPT*CTW MFMAs.
//   NCT == 3 : 8 compute waves = 8 row-groups of 2 rows, every wave owns ALL 3 cout tiles of 4 pixel-tiles (no padded cout slot:
//              the 2-group layout below would run 4 slots for 3 tiles -- SPAN-family layers with 48 channels);
//   NCT == 2, 4 : 8 compute waves = 2 cout groups (NCT/2 tiles each) x 4 row-groups of 4 rows (8 pixel-tiles);
//   NCT == 1 : 4 compute waves = 4 row-groups.
// One extra LOADER wave per workgroup streams the halo tiles into the double-buffered LDS image by LDS-DMA.
template <int KS, int NCT>
struct GeoLW {
  static constexpr int WCT = (NCT == 2 || NCT == 4) ? 2 : 1;  // compute-wave groups along cout
  static constexpr int WPX = (NCT == 3) ? 8 : 4;              // compute-wave groups along rows
  static constexpr int NCW = WCT * WPX;                        // compute waves
  static constexpr int CTW = (NCT + WCT - 1) / WCT;            // cout tiles per compute wave (1, 2 or 3)
  static constexpr int NTHR = (NCW + 1) * 64;                  // + loader wave
  static constexpr int TH = 16;                                // output rows of a tile
  static constexpr int RPW = TH / WPX;                         // rows per compute wave (4 or 2)
  static constexpr int NPT = 2 * RPW;                          // pixel tiles per compute wave (8 or 4)
  static constexpr int TW = 32;
  static constexpr int HALO = KS / 2;
  static constexpr int IH = TH + 2 * HALO;
  static constexpr int IW = TW + 2 * HALO;
  static constexpr int PS = ((IH * IW + 15) / 16) * 16;  // plane stride in units, == 0 mod 16
};


// Epilogue of one finished tile.  Every address is  (uniform 64-bit base) + (32-bit per-lane byte offset):
//   lane (li, lg) of wave (wct, wpx) owns, for pixel-tile pt and cout-tile c, the 4 consecutive channels
//   c0 = 16*(slab*NCT + CTW*wct + c) + 4*lg .. +3  of pixel (y0 + RPW*wpx + (pt>>1), x0 + 16*(pt&1) + li),  RPW = NPT/2.
// OUTK = 0: split planes and/or f32 residual map (with activation / residual epilogues)
// OUTK = 1: final plain NCHW tensor (optional activation, depth-to-space and affine), any dtype
// EM = epilogue SHAPE.  The generic body (EM 0) tests a dozen descriptor fields per fragment and keeps all of them live in scalar
// registers (the descriptor is ~80 SGPRs: 130 of the 770 instructions of one generic iteration were SGPR spill reloads).  The two
// shapes that make up an RRDBNet frame are compiled again with those tests folded:
//   EM 1: split-plane output only (hi and lo), no residual, no f32 map, no PReLU      -- the growth convolutions (276 of 351 launches)
//   EM 2: EM 1 + residual 1 as split planes (hi + lo) -- conv5 of a residual dense block;   EM 3: EM 2 + residual 2 (the RRDB's)
// PF = plane format of the outputs and residuals of EM 1 / 2 (the generic body reads p.out_fmt / p.res_fmt).  With PF = fp16, EM 1
// writes hi ONLY (the one-product consumers never read lo); EM 2 keeps hi + lo (the residual stream: 22 bits).
// XL 1 (EM 2 / 3 only; conv_ring.h's XRES kernels): residual 1 is the first 64 channels of the layer's own INPUT (conv5 of a residual dense
// block: x5 * 0.2 + x), whose hi halves are still in the ring slots `xslots` (4 bits per 16-channel half chunk) of the LDS array `x_lds` --
// they are read from there (slot geometry: units per slot / per plane / per halo row) instead of a second time from memory; lo halves come
// from res1_lo as before.
// L8: the lo halves of the plane residuals / outputs as 8-bit codes (rsa_conv_params.lo8_flags): -1 = as the descriptor's flags say (the generic
// body), 0 = none, 1 = every lo operand of the launch (the direct instantiation of conv5 inside an RRDBNet trunk: no runtime tests, no spills),
// 2 = the residuals only (the trunk's last block hands fp16 lo halves to the three-product layers behind it)
template <int NCT, int CTW, int NPT, int OUTK, int AC, int EM = 0, int PF = 0, int XL = 0, int L8 = -1>
__device__ __forceinline__ void epilogue_impl(const rsa_conv_params& p, const f32x4 (&acc)[NPT][CTW], int n, int y0, int x0, int slab, int wct,
                                              int wpx, int li, int lg, const uint4* x_lds = nullptr, uint32_t xslots = 0u, int x_slot = 0,
                                              int x_ps = 0, int x_iw = 0) {
  constexpr int RPW = NPT / 2;
  constexpr bool G = EM == 0;
  // EM 4 (conv_ring.h XRES 3: the re-parameterised layers of the SPAN family): ONE activation class AC (Mish / SiLU / SPAB gate / linear),
  // plane output in format PF (hi, and lo where the descriptor has it), the gate's shortcut as plane residual 1 (hi, and lo where it has
  // it); no f32 map, no second residual, no PReLU, whole cout tiles
  constexpr bool S = EM == 4;
  const bool R1F = G && p.res1 != nullptr, R2F = G && p.res2 != nullptr;                          // residuals as f32 maps
  const bool R1P = G ? p.res1_hi != nullptr : (S ? AC == AC_GATE : EM >= 2), R2P = G ? p.res2_hi != nullptr : EM == 3;  // as planes
  const bool R1L = (G || S) ? p.res1_lo != nullptr : true, R2L = G ? p.res2_lo != nullptr : true;       // ... with lo planes
  const bool OF32 = G && p.out_f32 != nullptr;
  const bool OHI = G ? p.out_hi != nullptr : true, OLO = (G || S) ? p.out_lo != nullptr : !(EM == 1 && PF == RSA_PF_F16);
  static_assert(EM >= 0 && EM <= 4, "epilogue shape");
  const bool OF16 = G ? p.out_fmt == RSA_PF_F16 : PF == RSA_PF_F16;  // plane format of the outputs / of the plane residuals
  const bool RF16 = G ? p.res_fmt == RSA_PF_F16 : PF == RSA_PF_F16;
  // lo halves as 8-bit codes (fp16 planes only; wave-uniform)
  const bool R1L8 = L8 < 0 ? (G && (p.lo8_flags & RSA_LO8_RES1) != 0) : L8 >= 1, R2L8 = L8 < 0 ? (G && (p.lo8_flags & RSA_LO8_RES2) != 0) : L8 >= 1,
             OL8 = L8 < 0 ? (G && (p.lo8_flags & RSA_LO8_OUT) != 0) : L8 == 1;  // (L8 2: the residuals only)
  const bool PRELU = G && p.act == RSA_ACT_PRELU;
  const float lin_slope = p.act == RSA_ACT_NONE ? 1.f : p.act_param;  // EM 1 / 2: act(v) = max(v, v * slope)
#ifdef RSA_ABL_NOEPI
  if (p.H > 0) {  // timing-only build: no epilogue, but every accumulator stays live (no dead-code elimination of the MFMAs)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) asm volatile("" ::"v"(acc[pt][ct]));
    return;
  }
#endif
  // li / lg are fixed for the whole kernel, so the compiler hoists every 64-bit per-lane address component derived from them out
  // of the persistent tile loop, finds no registers for them across the MFMA loop and spills them: ~40 scratch reloads per tile in
  // the epilogue, each waiting (vmcnt, in order) for the stores before it.  Laundering the two lane ids makes the derived values
  // tile-local: a handful of VALU instructions per fragment instead.
  asm volatile("" : "+v"(li), "+v"(lg));
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t pix0 = (int64_t)y0 * p.W + x0;  // uniform
  const int p4 = (p.cout + 3) >> 2;
  const int cout8 = (p.cout + 7) & ~7;
  const int ctile0 = slab * NCT + wct * CTW;

  // per-lane pixel offset / validity of pixel-tile pt, recomputed at every use from ONE lane register and uniform terms: holding
  // them in arrays (and the 64-bit addresses the compiler derived from them) cost the 2-cout-tile epilogue ~40 scratch reloads per
  // tile, each of which had to wait (vmcnt, in order) for the stores issued before it
  const uint32_t lbase = (uint32_t)(wpx * RPW * p.W + li);
  const bool xv0 = x0 + li < p.W, xv1 = x0 + 16 + li < p.W;
  auto lpix_of = [&](int pt) -> uint32_t { return lbase + (uint32_t)((pt >> 1) * p.W + (pt & 1) * 16); };
  auto pvalid_of = [&](int pt) -> bool { return (y0 + wpx * RPW + (pt >> 1) < p.H) && ((pt & 1) ? xv1 : xv0); };

  // vmcnt retires IN ORDER and counts stores: a load issued after a store cannot be waited for before that store's write is
  // acknowledged.  So every load of the epilogue is issued ahead of the stores it would otherwise queue behind: bias / slope
  // vectors of all cout tiles here, each cout tile's residual fragments before that tile's first store.
  f32x4 biasv[CTW], slopev[CTW];
#pragma unroll
  for (int ct = 0; ct < CTW; ++ct) {
    const int c0 = (ctile0 + ct) * 16 + lg * 4;
    const bool live = (wct * CTW + ct < NCT) && c0 < cout8;
    biasv[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    slopev[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && live) biasv[ct] = ((const f32x4*)p.bias)[c0 >> 2];  // bias is padded to a multiple of 16
    if (PRELU && live) slopev[ct] = ((const f32x4*)p.act_vec)[c0 >> 2];  // per-channel PReLU slopes, same padding
  }

  // Residual fragments are fetched ONE STEP AHEAD (a step = one pixel-tile pair of one cout tile): the loads of step s+1 are issued
  // before the stores of step s, so they never queue behind a store of this epilogue (vmcnt retires in order and counts stores).
  // A residual is either an f32 map (res1 / res2) or split planes (res1_hi / res1_lo ...: value = hi + lo).  Plane residuals arrive
  // as two 8-byte halves of a unit (this lane's 4 channels) and are widened to f32 here, so the arithmetic below sees one form.
  const bool has_r1 = R1F || R1P;
  const bool has_r2 = R2F || R2P;
  auto widen = [&](uint2 h, uint2 l) -> f32x4 { return RF16 ? widen4<RSA_PF_F16>(h, l) : widen4<RSA_PF_BF16>(h, l); };
  f32x4 nr1[2], nr2[2];
  auto fetch_res = [&](int ct, int pp) {
    const int cbase = (ctile0 + ct) * 16;
    const int c0 = cbase + lg * 4;
    const bool cok = (wct * CTW + ct < NCT) && c0 < cout8 && c0 < (p4 << 2);
    const char* r1b = (const char*)p.res1 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const char* r2b = (const char*)p.res2 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    // plane residuals: unit (plane cbase/8 + lg/2, pixel), half lg & 1
    const int64_t runit0 = (int64_t)n * p.res_batch_stride + (int64_t)(cbase >> 3) * p.res_plane_stride + pix0;
    const int64_t runit8 = (int64_t)n * p.lo8_batch_stride + (int64_t)(cbase >> 3) * p.res_plane_stride + pix0;  // the same unit of an lo8 buffer
    const uint32_t rlane = (uint32_t)(lg >> 1) * (uint32_t)p.res_plane_stride;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pt = pp * 2 + e;
      const bool okl = pvalid_of(pt) && cok;
      const uint32_t foff = ((uint32_t)lg * (uint32_t)HW + lpix_of(pt)) * 16u;
      const uint32_t poff = (rlane + lpix_of(pt)) * 16u + (uint32_t)(lg & 1) * 8u;
      nr1[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
      nr2[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (R1F && okl) nr1[e] = *(const f32x4*)(r1b + foff);
      if (R2F && okl) nr2[e] = *(const f32x4*)(r2b + foff);
      if (R1P && okl) {
        const uint2 h = *(const uint2*)((const char*)p.res1_hi + runit0 * 16 + poff);
        if (R1L8) {
          nr1[e] = widen4_lo8(h, *(const uint32_t*)((const char*)p.res1_lo + runit8 * 8 + (poff >> 1)));
        } else {
          const uint2 l = R1L ? *(const uint2*)((const char*)p.res1_lo + runit0 * 16 + poff) : make_uint2(0u, 0u);
          nr1[e] = widen(h, l);
        }
      }
      if (R2P && okl) {
        const uint2 h = *(const uint2*)((const char*)p.res2_hi + runit0 * 16 + poff);
        if (R2L8) {
          nr2[e] = widen4_lo8(h, *(const uint32_t*)((const char*)p.res2_lo + runit8 * 8 + (poff >> 1)));
        } else {
          const uint2 l = R2L ? *(const uint2*)((const char*)p.res2_lo + runit0 * 16 + poff) : make_uint2(0u, 0u);
          nr2[e] = widen(h, l);
        }
      }
    }
  };
  // EM 2 (plane residuals of a residual dense block's conv5): the residual halves are fetched PD steps ahead in raw form.  One step
  // ahead left every one of the 8 steps of a tile waiting for a full memory latency (the one-product multiply of a tile is as long as
  // those eight waits); PD steps in flight cost (PD + 1) * 8 (16 with a second residual) registers, free after the K loop.
#ifndef RSA_EPI_PD
#define RSA_EPI_PD 3
#endif
#ifndef RSA_EPI_C5
#define RSA_EPI_C5 0  // 1: conv5's arithmetic without the per-lane selects and the (identity) activation, 2: the same on value pairs (v_pk_fma_f32).
                      // Both measured SLOWER than the select form on the frame (+0.6 % / +4.6 %: profiles/r04_t_conv5_epilogue_forms_ab.txt)
#endif
#ifndef RSA_EPI_PDX
#define RSA_EPI_PDX 1  // XL: the hi halves come from LDS at their use, only lo halves (and the second residual) are fetched ahead; one step
                       // ahead measured best (deeper: the raw values spill; profiles/r03_i_conv5_epilogue_prefetch.txt)
#endif
  constexpr int PD = XL ? RSA_EPI_PDX : (EM == 2 ? RSA_EPI_PD : (EM == 3 ? (RSA_EPI_PD + 1) / 2 : 0));
  constexpr int NSTEPS_E = CTW * RPW;
  uint2 rb1h[PD + 1][2], rb1l[PD + 1][2], rb2h[EM == 3 ? PD + 1 : 1][2], rb2l[EM == 3 ? PD + 1 : 1][2];
  auto fetch_raw = [&](int s) {
    const int ct = s / RPW, pp = s % RPW, slot = s % (PD + 1);
    const int cbase = (ctile0 + ct) * 16;
    const int c0 = cbase + lg * 4;
    const bool cok = (wct * CTW + ct < NCT) && c0 < cout8;
    const int64_t runit0 = (int64_t)n * p.res_batch_stride + (int64_t)(cbase >> 3) * p.res_plane_stride + pix0;
    const int64_t runit8 = (int64_t)n * p.lo8_batch_stride + (int64_t)(cbase >> 3) * p.res_plane_stride + pix0;
    const uint32_t rlane = (uint32_t)(lg >> 1) * (uint32_t)p.res_plane_stride;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pt = pp * 2 + e;
      const bool okl = pvalid_of(pt) && cok;
      const uint32_t poff = (rlane + lpix_of(pt)) * 16u + (uint32_t)(lg & 1) * 8u;
      rb1h[slot][e] = rb1l[slot][e] = make_uint2(0u, 0u);
      if (EM == 3) rb2h[slot][e] = rb2l[slot][e] = make_uint2(0u, 0u);
      if (okl) {
        if (!XL) rb1h[slot][e] = *(const uint2*)((const char*)p.res1_hi + runit0 * 16 + poff);  // (XL: read from the ring at its use, below)
        if (R1L8)  // one dword of codes (kept in .x; widened by widen4_lo8)
          rb1l[slot][e].x = *(const uint32_t*)((const char*)p.res1_lo + runit8 * 8 + (poff >> 1));
        else
          rb1l[slot][e] = *(const uint2*)((const char*)p.res1_lo + runit0 * 16 + poff);
        if (EM == 3) {
          rb2h[slot][e] = *(const uint2*)((const char*)p.res2_hi + runit0 * 16 + poff);
          if (R2L8)
            rb2l[slot][e].x = *(const uint32_t*)((const char*)p.res2_lo + runit8 * 8 + (poff >> 1));
          else
            rb2l[slot][e] = *(const uint2*)((const char*)p.res2_lo + runit0 * 16 + poff);
        }
      }
    }
  };
  if (OUTK == 0 && (EM == 0 || (S && AC == AC_GATE))) fetch_res(0, 0);
  if (OUTK == 0 && (EM == 2 || EM == 3)) {
#pragma unroll
    for (int s = 0; s < PD && s < NSTEPS_E; ++s) fetch_raw(s);
  }

#pragma unroll
  for (int ct = 0; ct < CTW; ++ct) {
    if (wct * CTW + ct >= NCT) break;
    const int cbase = (ctile0 + ct) * 16;  // uniform
    if (cbase >= cout8) break;
    const int c0 = cbase + lg * 4;
    const f32x4 bias = biasv[ct];
    const f32x4 slope = slopev[ct];
    const bool cvalid = c0 < cout8;
    const bool has_f32grp = c0 < (p4 << 2);
    // uniform bases for this cout tile
    char* f32b = (char*)p.out_f32 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const int64_t ounit0 = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0;
    char* ohb = (char*)p.out_hi + ounit0 * 16;
    char* olb = (char*)p.out_lo + ounit0 * 16;
    const uint32_t f32lane = (uint32_t)lg * (uint32_t)HW;                       // + lpix, in float4 units
    const uint32_t pllane = (uint32_t)(lg >> 1) * (uint32_t)p.out_plane_stride;  // + lpix, in 16-byte units
    // pixel tiles are handled in PAIRS (2k, 2k+1) = the two 16-pixel halves of one row.  After the per-fragment math the two
    // lanes that hold the two halves of a 16-byte unit (lg, lg^1 = lanes l, l^16) exchange one half each, so that lane lg-even
    // stores the FULL unit of pixel-tile 2k and lane lg-odd the full unit of pixel-tile 2k+1: 16-byte stores, 512 contiguous
    // bytes per plane and instruction instead of two half-filled 256-byte runs (the 8-byte form was store-issue bound:
    // profiles/r01_l_ab_epilogue.txt)
#pragma unroll
    for (int pp = 0; pp < RPW; ++pp) {
      float v[2][4];
      bool ok[2];
      f32x4 cr1[2], cr2[2];
      if (OUTK == 0) {
        if (EM == 2 || EM == 3) {
          const int s = ct * RPW + pp;
          if (s + PD < NSTEPS_E) fetch_raw(s + PD);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (XL) {  // the hi halves of x: ring slot of the half chunk (plane >> 1), plane & 1 inside it, halo coordinates (row + 1, column + 1)
              const int pt = pp * 2 + e;
              const int plane = (cbase >> 3) + (lg >> 1);
              const uint32_t u = ((xslots >> (4 * (plane >> 1))) & 15u) * (uint32_t)x_slot + (uint32_t)((plane & 1) * x_ps) +
                                 (uint32_t)((wpx * RPW + (pt >> 1) + 1) * x_iw + (pt & 1) * 16 + li + 1);
              rb1h[s % (PD + 1)][e] = *(const uint2*)((const char*)x_lds + u * 16u + (uint32_t)(lg & 1) * 8u);
            }
            cr1[e] = R1L8 ? widen4_lo8(rb1h[s % (PD + 1)][e], rb1l[s % (PD + 1)][e].x) : widen(rb1h[s % (PD + 1)][e], rb1l[s % (PD + 1)][e]);
            cr2[e] = EM != 3 ? (f32x4){0.f, 0.f, 0.f, 0.f}
                             : (R2L8 ? widen4_lo8(rb2h[EM == 3 ? s % (PD + 1) : 0][e], rb2l[EM == 3 ? s % (PD + 1) : 0][e].x)
                                     : widen(rb2h[EM == 3 ? s % (PD + 1) : 0][e], rb2l[EM == 3 ? s % (PD + 1) : 0][e]));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            cr1[e] = nr1[e];
            cr2[e] = nr2[e];
          }
        }
        if (EM != 0 && !(S && AC == AC_GATE)) {
        } else if (pp + 1 < RPW)
          fetch_res(ct, pp + 1);
        else if (ct + 1 < CTW)
          fetch_res(ct + 1, 0);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int pt = pp * 2 + e;
        ok[e] = pvalid_of(pt) && cvalid;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[e][r] = acc[pt][ct][r] + bias[r];
        if (OUTK == 0 && RSA_EPI_C5 && (EM == 2 || EM == 3)) {
          // conv5 of a residual dense block: (acc + bias [LeakyReLU]) * alpha + r1 [* beta + r2].  Lanes outside the map hold zeros for the
          // residuals and store nothing: no per-lane selects; no activation (conv5 has none): a wave-uniform skip of its two instructions per value.
#if RSA_EPI_C5 == 2  // value pairs (v_pk_fma_f32): fewer instructions, but the register pairs cost moves and spills (A/B: profiles/r04_t_*)
          f32x2 a = {v[e][0], v[e][1]}, b = {v[e][2], v[e][3]};
          if (p.act != RSA_ACT_NONE) {
            a = __builtin_elementwise_max(a, a * lin_slope);
            b = __builtin_elementwise_max(b, b * lin_slope);
          }
          const f32x2 al = {p.alpha, p.alpha};
          a = __builtin_elementwise_fma(a, al, (f32x2){cr1[e][0], cr1[e][1]});
          b = __builtin_elementwise_fma(b, al, (f32x2){cr1[e][2], cr1[e][3]});
          if (EM == 3) {
            const f32x2 be = {p.beta, p.beta};
            a = __builtin_elementwise_fma(a, be, (f32x2){cr2[e][0], cr2[e][1]});
            b = __builtin_elementwise_fma(b, be, (f32x2){cr2[e][2], cr2[e][3]});
          }
          v[e][0] = a[0], v[e][1] = a[1], v[e][2] = b[0], v[e][3] = b[1];
#else
          if (p.act != RSA_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[e][r] = fmaxf(v[e][r], v[e][r] * lin_slope);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.alpha + cr1[e][r];
          if (EM == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.beta + cr2[e][r];
          }
#endif
        } else if (OUTK == 0) {
          const uint32_t foff = (f32lane + lpix_of(pt)) * 16u;
          if (AC == AC_GATE) {
            const f32x4 rr = cr1[e];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float sg = __builtin_amdgcn_rcpf(1.f + exp_fast(-v[e][r]));
              v[e][r] = (v[e][r] + rr[r]) * (sg - 0.5f);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (G || (S && AC != AC_LINEAR))
                v[e][r] = act_apply<AC>(v[e][r], p.act, PRELU ? slope[r] : p.act_param);
              else
                v[e][r] = fmaxf(v[e][r], v[e][r] * lin_slope);  // none / LeakyReLU with a slope in [0, 1]: two instructions per value
            }
            if (has_r1 && has_f32grp && ok[e]) {
              const f32x4 rr = cr1[e];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.alpha + rr[r];
            }
          }
          if (has_r2 && has_f32grp && ok[e]) {
            const f32x4 rr = cr2[e];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.beta + rr[r];
          }
          if (G && (p.cout & 15)) {  // channels beyond Cout -> zeros.  A wave-uniform test first: layers whose Cout fills its tiles (32, 48, 64 ...:
                                     // every hot layer) skip the compare + select per value -- as expensive as a tenth of a Mish
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + r >= p.cout) v[e][r] = 0.f;
          }
          if (OF32 && has_f32grp && ok[e]) *(f32x4*)(f32b + foff) = (f32x4){v[e][0], v[e][1], v[e][2], v[e][3]};
        }
      }
      if (OUTK == 0) {
        if (OHI) {  // wave-uniform: every lane takes part in the exchange
          uint32_t h[2][2], l[2][2], q8[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (OLO && OL8) {  // hi halves and the 8-bit codes of the lo halves from one pass over the values
              split4_lo8(v[e][0], v[e][1], v[e][2], v[e][3], h[e][0], h[e][1], q8[e]);
              l[e][0] = l[e][1] = 0u;
            } else {
              split2_rt(OF16, v[e][0], v[e][1], h[e][0], l[e][0]);
              split2_rt(OF16, v[e][2], v[e][3], h[e][1], l[e][1]);
            }
          }
          const int odd = lg & 1;  // odd lanes keep pixel-tile 2k+1 and give away their half of 2k; even lanes the reverse
          // v_permlane16_swap_b32 a, b swaps the odd 16-lane rows of a with the even rows of b: with a = this lane's half of pixel-tile
          // 2k and b = its half of 2k+1, an even lane ends up with (own half of 2k, partner's half of 2k) and an odd lane with
          // (partner's half of 2k+1, own half of 2k+1) -- the full 16-byte units, one instruction per register pair, no selects
#ifndef RSA_EPI_SHFL
          const u32x2 h0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
          const u32x2 h1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
          const u32x2 l0 = __builtin_amdgcn_permlane16_swap(l[0][0], l[1][0], false, false);
          const u32x2 l1 = __builtin_amdgcn_permlane16_swap(l[0][1], l[1][1], false, false);
          const int pt = pp * 2 + odd;
          // unit = channels 8*(plane) .. +7: the even lane owns channels 0-3, the odd lane channels 4-7
          const uint4 uh = make_uint4(h0.x, h1.x, h0.y, h1.y);
          const uint4 ul = make_uint4(l0.x, l1.x, l0.y, l1.y);
#else  // round-1 form (ds_bpermute + selects), kept for A/B builds
          uint32_t sh[2], sl[2], rh[2], rl[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            sh[d] = odd ? h[0][d] : h[1][d];
            sl[d] = odd ? l[0][d] : l[1][d];
            rh[d] = (uint32_t)__shfl_xor((int)sh[d], 16);
            rl[d] = (uint32_t)__shfl_xor((int)sl[d], 16);
          }
          const int pt = pp * 2 + odd;
          const uint4 uh = odd ? make_uint4(rh[0], rh[1], h[1][0], h[1][1]) : make_uint4(h[0][0], h[0][1], rh[0], rh[1]);
          const uint4 ul = odd ? make_uint4(rl[0], rl[1], l[1][0], l[1][1]) : make_uint4(l[0][0], l[0][1], rl[0], rl[1]);
#endif
          // the partner lane has the same pixel column li and the same plane; both halves are valid together (same cvalid)
          if (pvalid_of(pt) && cvalid) {
            const uint32_t uoff = (pllane + lpix_of(pt)) * 16u;
            *(uint4*)(ohb + uoff) = uh;
            if (OLO && !OL8) *(uint4*)(olb + uoff) = ul;
          }
          if (OLO && OL8) {  // wave-uniform: the residuals as 8-bit codes, 8 bytes per unit (one dword per lane before the exchange)
            const u32x2 l8 = __builtin_amdgcn_permlane16_swap(q8[0], q8[1], false, false);
            if (pvalid_of(pt) && cvalid) {
              const int64_t ounit8 = (int64_t)n * p.lo8_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0;
              *(uint2*)((char*)p.out_lo + ounit8 * 8 + (pllane + lpix_of(pt)) * 8u) = make_uint2(l8.x, l8.y);
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int pt = pp * 2 + e;
          if (!ok[e]) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[e][r] = act_apply<AC>(v[e][r], p.act, p.act == RSA_ACT_PRELU ? slope[r] : p.act_param);
          const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
          const int oc_total = p.cout / (ps * ps);
          const int64_t oW = (int64_t)p.W * ps;
          const int64_t oHW = (int64_t)p.H * ps * oW;
          const int y = y0 + wpx * RPW + (pt >> 1), x = x0 + (pt & 1) * 16 + li;
          // Depth-to-space by 4 or 2 into a plain 16- / 32-bit tensor (the pixel-shuffle heads): this lane's four channels
          // c0 .. c0+3 = 16*tile + 4*lg + r are ONE output channel's pixels (row 4y + lg, columns 4x .. 4x+3) for a factor of 4, and two
          // row pairs of output channel 4*tile + lg for a factor of 2 -- one or two vector stores, 128 contiguous bytes per 16 lanes,
          // instead of four element stores each (the 2-byte form made the x4 heads store-issue-bound: 0.65 ms for 201 MB).
          if ((ps == 4 || ps == 2) && p.out_base == nullptr && p.out_dtype != RSA_U8 && cbase + 16 <= p.cout) {
            const int oc = ps == 4 ? (c0 >> 4) : (c0 >> 2);
            const float sh = p.out_shift != nullptr ? p.out_shift[oc] : 0.f;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = v[e][r] * p.out_scale + sh;
            const int64_t plane = ((int64_t)n * oc_total + oc) * oHW;
            if (ps == 4) {
              const int64_t idx = plane + ((int64_t)y * 4 + lg) * oW + (int64_t)x * 4;
              if (p.out_dtype == RSA_F32) {
                *(f32x4*)((float*)p.out_nchw + idx) = (f32x4){o[0], o[1], o[2], o[3]};
              } else if (p.out_dtype == RSA_F16) {
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                *(f16x4*)((_Float16*)p.out_nchw + idx) = (f16x4){(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
              } else {
                *(bf16x4*)((__bf16*)p.out_nchw + idx) = (bf16x4){(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
              }
            } else {
#pragma unroll
              for (int ii = 0; ii < 2; ++ii) {
                const int64_t idx = plane + ((int64_t)y * 2 + ii) * oW + (int64_t)x * 2;
                if (p.out_dtype == RSA_F32) {
                  typedef __attribute__((ext_vector_type(2))) float f32x2;
                  *(f32x2*)((float*)p.out_nchw + idx) = (f32x2){o[2 * ii], o[2 * ii + 1]};
                } else if (p.out_dtype == RSA_F16) {
                  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                  *(f16x2*)((_Float16*)p.out_nchw + idx) = (f16x2){(_Float16)o[2 * ii], (_Float16)o[2 * ii + 1]};
                } else {
                  *(bf16x2*)((__bf16*)p.out_nchw + idx) = (bf16x2){(__bf16)o[2 * ii], (__bf16)o[2 * ii + 1]};
                }
              }
            }
            continue;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = c0 + r;
            if (c >= p.cout) continue;
            // depth-to-space coordinates; without pixel shuffle (wave-uniform test) no integer division is emitted: the three divisions
            // per value made the 64 -> 3 last convolution of RRDBNet epilogue-bound (4.7 ms at 16 % MFMA busy)
            int oc = c, ii = 0, jj = 0;
            if (ps != 1) {
              oc = c / (ps * ps);
              const int rem = c - oc * ps * ps;
              ii = rem / ps;
              jj = rem - ii * ps;
            }
            float o = v[e][r] * p.out_scale;
            if (p.out_shift != nullptr) o += p.out_shift[oc];
            if (p.out_base != nullptr) {  // nearest-upsampled base image: the low-resolution pixel this output pixel sits in
              int64_t bi = (((int64_t)n * oc_total + oc) * p.H + y) * p.W + x;
              if (p.out_base_div > 0) {
                int by = (y * ps + ii) / p.out_base_div, bx = (x * ps + jj) / p.out_base_div;
                by = by < p.out_base_h ? by : p.out_base_h - 1;
                bx = bx < p.out_base_w ? bx : p.out_base_w - 1;
                bi = (((int64_t)n * oc_total + oc) * p.out_base_h + by) * p.out_base_w + bx;
              }
              if (p.out_dtype == RSA_F32)
                o += ((const float*)p.out_base)[bi];
              else if (p.out_dtype == RSA_F16)
                o += (float)((const _Float16*)p.out_base)[bi];
              else
                o += (float)((const __bf16*)p.out_base)[bi];
            }
            const int64_t idx = ((int64_t)n * oc_total + oc) * oHW + ((int64_t)y * ps + ii) * oW + ((int64_t)x * ps + jj);
            if (p.out_dtype == RSA_U8) {  // 8-bit image, channel-interleaved: clamp, scale, round half to even (v_rndne)
              const int64_t pidx = (((int64_t)n * p.H * ps + ((int64_t)y * ps + ii)) * oW + ((int64_t)x * ps + jj)) * oc_total + oc;
              ((uint8_t*)p.out_nchw)[pidx] = (uint8_t)rintf(fminf(fmaxf(o, 0.f), 1.f) * 255.f);
            } else if (p.out_dtype == RSA_F32)
              ((float*)p.out_nchw)[idx] = o;
            else if (p.out_dtype == RSA_F16)
              ((_Float16*)p.out_nchw)[idx] = (_Float16)o;
            else
              ((__bf16*)p.out_nchw)[idx] = (__bf16)o;
          }
        }
      }
      // keep the fragment epilogues from being interleaved by the scheduler: interleaving them costs more registers than the
      // 168-VGPR budget of the 9-wave schedule has and turns the epilogue into scratch traffic
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NCT, int CTW, int NPT, int OUTK>
__device__ __forceinline__ void epilogue(const rsa_conv_params& p, const f32x4 (&acc)[NPT][CTW], int n, int y0, int x0, int slab, int wct, int wpx,
                                         int li, int lg) {
  switch (act_class(p.act)) {  // wave-uniform: one compact code path is fetched per tile
    case AC_MISH:
      return epilogue_impl<NCT, CTW, NPT, OUTK, AC_MISH>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
    case AC_SILU:
      return epilogue_impl<NCT, CTW, NPT, OUTK, AC_SILU>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
    case AC_GELU:
      return epilogue_impl<NCT, CTW, NPT, OUTK, AC_GELU>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
    case AC_GATE:
      if (OUTK == 0) return epilogue_impl<NCT, CTW, NPT, OUTK, AC_GATE>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
      return;
    default:
#ifndef RSA_NO_EM
      if (OUTK == 0 && p.out_hi != nullptr && p.out_f32 == nullptr && p.res1 == nullptr && p.res2 == nullptr && p.act != RSA_ACT_PRELU && p.lo8_flags == 0 &&
          (p.cout & 15) == 0 && (p.act == RSA_ACT_NONE || (p.act_param >= 0.f && p.act_param <= 1.f))) {
        // the two shapes of an RRDBNet frame, with the descriptor tests folded (see EM above); wave-uniform choice
        const bool planes_res = p.res1_hi != nullptr && p.res1_lo != nullptr && (p.res2_hi == nullptr || p.res2_lo != nullptr);
        const bool two = p.res2_hi != nullptr;
        if (p.out_fmt == RSA_PF_BF16 && p.out_lo != nullptr) {
          if (p.res1_hi == nullptr && p.res2_hi == nullptr) return epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 1>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
          if (planes_res && p.res_fmt == RSA_PF_BF16)
            return two ? epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 3>(p, acc, n, y0, x0, slab, wct, wpx, li, lg)
                       : epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 2>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
        } else if (p.out_fmt == RSA_PF_F16) {
          if (p.out_lo == nullptr && p.res1_hi == nullptr && p.res2_hi == nullptr)
            return epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 1, RSA_PF_F16>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
          if (p.out_lo != nullptr && planes_res && p.res_fmt == RSA_PF_F16)
            return two ? epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 3, RSA_PF_F16>(p, acc, n, y0, x0, slab, wct, wpx, li, lg)
                       : epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR, 2, RSA_PF_F16>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
        }
      }
#endif
      return epilogue_impl<NCT, CTW, NPT, OUTK, AC_LINEAR>(p, acc, n, y0, x0, slab, wct, wpx, li, lg);
  }
}

}  // namespace rsa
