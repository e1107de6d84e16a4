// conv_common.h — pieces shared by the two schedules of the fused convolution (conv_mfma.hip, conv_rs.hip):
// vector types, activation functions, the bf16 hi/lo split and the tile epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int NPL = 4;  // planes per K chunk (32 channels = one MFMA K)

__device__ __forceinline__ float act_apply(float v, int act, float prm) {
  switch (act) {
    case RSA_ACT_LRELU:
    case RSA_ACT_PRELU:  // prm = this channel's slope
      return v >= 0.f ? v : v * prm;
    case RSA_ACT_MISH: {
      // torch: x * tanh(softplus(x)), softplus threshold 20
      float sp = v > 20.f ? v : log1pf(expf(v));
      return v * tanhf(sp);
    }
    case RSA_ACT_SILU:
      return v / (1.f + expf(-v));
    case RSA_ACT_GELU:
      return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    default:
      return v;
  }
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// (a, b) -> packed bf16 pair (RNE) and the pair's rounding residuals, also packed
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(uint32_t, h);
  const float ra = a - __builtin_bit_cast(float, hi << 16);
  const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
  const bf16x2 l = {(__bf16)ra, (__bf16)rb};
  lo = __builtin_bit_cast(uint32_t, l);
}

// Geometry of one instantiation.  Output tile = 16 rows x 32 pixels.  A COMPUTE wave owns 8 pixel-tiles (4 rows x 2 halves
// of 16 pixels) x CTW cout-tiles; one tap costs it 16 LDS fragment reads + 2*CTW weight fragment loads for 24*CTW MFMAs.
// NCT >= 2: 8 compute waves = 2 cout groups (CTW = ceil(NCT/2) tiles each) x 4 row-groups;  NCT == 1: 4 compute waves.
// One extra LOADER wave per workgroup streams the halo tiles into the double-buffered LDS image by LDS-DMA.
template <int KS, int NCT>
struct GeoLW {
  static constexpr int WCT = (NCT >= 2) ? 2 : 1;       // compute-wave groups along cout
  static constexpr int WPX = 4;                        // compute-wave groups along rows (4 rows each)
  static constexpr int NCW = WCT * WPX;                // compute waves
  static constexpr int CTW = (NCT + WCT - 1) / WCT;    // cout tiles per compute wave (1 or 2)
  static constexpr int NTHR = (NCW + 1) * 64;          // + loader wave
  static constexpr int TH = 4 * WPX;                   // 16 output rows
  static constexpr int TW = 32;
  static constexpr int HALO = KS / 2;
  static constexpr int IH = TH + 2 * HALO;
  static constexpr int IW = TW + 2 * HALO;
  static constexpr int PS = ((IH * IW + 15) / 16) * 16;  // plane stride in units, == 0 mod 16
};


// Epilogue of one finished tile.  Every address is  (uniform 64-bit base) + (32-bit per-lane byte offset):
//   lane (li, lg) of wave (wct, wpx) owns, for pixel-tile pt and cout-tile c, the 4 consecutive channels
//   c0 = 16*(slab*NCT + CTW*wct + c) + 4*lg .. +3  of pixel (y0 + 4*wpx + (pt>>1), x0 + 16*(pt&1) + li).
// OUTK = 0: split planes and/or f32 residual map (with activation / residual epilogues)
// OUTK = 1: final plain NCHW tensor (optional activation, depth-to-space and affine), any dtype
template <int NCT, int CTW, int OUTK>
__device__ __forceinline__ void epilogue(const rsa_conv_params& p, const f32x4 (&acc)[8][CTW], int n, int y0, int x0, int slab, int wct, int wpx,
                                         int li, int lg) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t pix0 = (int64_t)y0 * p.W + x0;  // uniform
  const int p4 = (p.cout + 3) >> 2;
  const int cout8 = (p.cout + 7) & ~7;
  const int ctile0 = slab * NCT + wct * CTW;

  uint32_t lpix[8];
  bool pvalid[8];
#pragma unroll
  for (int pt = 0; pt < 8; ++pt) {
    const int ry = wpx * 4 + (pt >> 1), rx = (pt & 1) * 16 + li;
    lpix[pt] = (uint32_t)(ry * p.W + rx);
    pvalid[pt] = (y0 + ry < p.H) && (x0 + rx < p.W);
  }

#pragma unroll
  for (int ct = 0; ct < CTW; ++ct) {
    if (wct * CTW + ct >= NCT) break;
    const int cbase = (ctile0 + ct) * 16;  // uniform
    if (cbase >= cout8) break;
    const int c0 = cbase + lg * 4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bias = ((const f32x4*)p.bias)[c0 >> 2];  // bias is padded to a multiple of 16
    f32x4 slope = {0.f, 0.f, 0.f, 0.f};
    if (p.act == RSA_ACT_PRELU) slope = ((const f32x4*)p.act_vec)[c0 >> 2];  // per-channel PReLU slopes, same padding
    const bool cvalid = c0 < cout8;
    const bool has_f32grp = c0 < (p4 << 2);
    // uniform bases for this cout tile
    const char* r1b = (const char*)p.res1 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const char* r2b = (const char*)p.res2 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    char* f32b = (char*)p.out_f32 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const int64_t ounit0 = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0;
    char* ohb = (char*)p.out_hi + ounit0 * 16;
    char* olb = (char*)p.out_lo + ounit0 * 16;
    const uint32_t f32lane = (uint32_t)lg * (uint32_t)HW;                       // + lpix, in float4 units
    const uint32_t pllane = (uint32_t)(lg >> 1) * (uint32_t)p.out_plane_stride;  // + lpix, in 16-byte units
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      if (!pvalid[pt] || !cvalid) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[pt][ct][r] + bias[r];
      if (OUTK == 0) {
        const uint32_t foff = (f32lane + lpix[pt]) * 16u;
        if (p.act == RSA_ACT_SPAB_GATE) {
          f32x4 rr = {0.f, 0.f, 0.f, 0.f};
          if (has_f32grp) rr = *(const f32x4*)(r1b + foff);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = 1.f / (1.f + expf(-v[r]));
            v[r] = (v[r] + rr[r]) * (sg - 0.5f);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act, p.act == RSA_ACT_PRELU ? slope[r] : p.act_param);
          if (p.res1 != nullptr && has_f32grp) {
            const f32x4 rr = *(const f32x4*)(r1b + foff);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + rr[r];
          }
        }
        if (p.res2 != nullptr && has_f32grp) {
          const f32x4 rr = *(const f32x4*)(r2b + foff);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * p.beta + rr[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + r >= p.cout) v[r] = 0.f;
        if (p.out_hi != nullptr) {
          uint32_t h0, l0, h1, l1;
          split2(v[0], v[1], h0, l0);
          split2(v[2], v[3], h1, l1);
          const uint32_t uoff = (pllane + lpix[pt]) * 16u + (uint32_t)(lg & 1) * 8u;
          *(uint2*)(ohb + uoff) = make_uint2(h0, h1);
          if (p.out_lo != nullptr) *(uint2*)(olb + uoff) = make_uint2(l0, l1);
        }
        if (p.out_f32 != nullptr && has_f32grp) *(f32x4*)(f32b + foff) = (f32x4){v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act, p.act == RSA_ACT_PRELU ? slope[r] : p.act_param);
        const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
        const int oc_total = p.cout / (ps * ps);
        const int64_t oW = (int64_t)p.W * ps;
        const int64_t oHW = (int64_t)p.H * ps * oW;
        const int y = y0 + wpx * 4 + (pt >> 1), x = x0 + (pt & 1) * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + r;
          if (c >= p.cout) continue;
          const int oc = c / (ps * ps);
          const int rem = c - oc * ps * ps;
          const int ii = rem / ps;
          const int jj = rem - ii * ps;
          float o = v[r] * p.out_scale;
          if (p.out_shift != nullptr) o += p.out_shift[oc];
          if (p.out_base != nullptr) {  // nearest-upsampled base image: the low-resolution pixel this output pixel sits in
            const int64_t bi = (((int64_t)n * oc_total + oc) * p.H + y) * p.W + x;
            if (p.out_dtype == RSA_F32)
              o += ((const float*)p.out_base)[bi];
            else if (p.out_dtype == RSA_F16)
              o += (float)((const _Float16*)p.out_base)[bi];
            else
              o += (float)((const __bf16*)p.out_base)[bi];
          }
          const int64_t idx = ((int64_t)n * oc_total + oc) * oHW + ((int64_t)y * ps + ii) * oW + ((int64_t)x * ps + jj);
          if (p.out_dtype == RSA_F32)
            ((float*)p.out_nchw)[idx] = o;
          else if (p.out_dtype == RSA_F16)
            ((_Float16*)p.out_nchw)[idx] = (_Float16)o;
          else
            ((__bf16*)p.out_nchw)[idx] = (__bf16)o;
        }
      }
      // keep the 8*CTW fragment epilogues from being interleaved by the scheduler: interleaving them costs more registers
      // than the 168-VGPR budget of the 9-wave schedule has and turns the epilogue into scratch traffic
      // (in-process A/B, profiles/r01_h_ab_epilogue_serial.txt: +2..4 % on bf16x3 layers)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}


}  // namespace rsa
