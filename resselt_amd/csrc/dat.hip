// dat.hip — the non-convolution kernels of the DAT path (reference resselt/archs/dat/arch.py):
//   rsa_rect_attention            rectangular (shifted) window attention core     Spatial_Attention.forward :224-267 and the
//                                 pad / roll / partition / mask / reverse steps of Adaptive_Spatial_Attention.forward :446-492
//   rsa_channel_attention_weights softmax(normalize(q) normalize(k)^T * temperature) over all tokens   :577-585
//   rsa_dwconv3x3                 depthwise 3x3 (+ folded BatchNorm, GELU, on-the-fly LayerNorm, gate multiply)  :52-59, 321-325
//   rsa_plane_stats               per-pixel LayerNorm statistics                   :49, 58
//   rsa_channel_gate              AdaptiveAvgPool -> 1x1 -> BN -> GELU -> 1x1 -> sigmoid      :326-332
//   rsa_aim_combine               spatial_interaction MLP + the two gated sums     :333-338, 494-508, 595-607
//
// Tokens are pixels; Linear layers are k1 launches of the convolution kernels.  Everything here is HBM-bound VALU work except
// the window attention, which follows swin.hip's scheme (S^T = K Q^T on v_mfma_f32_32x32x16_bf16, in-lane softmax, the S^T
// accumulators reused as the B operand of O^T = V^T P^T) extended to 8 key tiles and workgroup-shared K / V images in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// one 16-byte unit (8 channels of a pixel) of planes in format fmt (enum rsa_plane_fmt, wave-uniform) -> f32: hi (+ lo where the buffer has it)
__device__ __forceinline__ void unit_f32(const bf16x8* hi, const bf16x8* lo, int64_t u, float (&v)[8], int fmt = RSA_PF_BF16) {
  const bf16x8 h = hi[u];
  if (fmt == RSA_PF_F16) {
    const f16x8_t hf = __builtin_bit_cast(f16x8_t, h);
    if (lo != nullptr) {
      const f16x8_t lf = __builtin_bit_cast(f16x8_t, lo[u]);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)hf[j] + (float)lf[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (float)hf[j];
    }
    return;
  }
  if (lo != nullptr) {
    const bf16x8 l = lo[u];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
}

// a raw 16-byte unit already in registers -> f32 (the loads issued earlier, several at a time)
__device__ __forceinline__ void raw_f32(const bf16x8 h, const bf16x8 l, bool has_lo, float (&v)[8], int fmt) {
  if (fmt == RSA_PF_F16) {
    const f16x8_t hf = __builtin_bit_cast(f16x8_t, h), lf = __builtin_bit_cast(f16x8_t, l);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = has_lo ? (float)hf[j] + (float)lf[j] : (float)hf[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = has_lo ? (float)h[j] + (float)l[j] : (float)h[j];
  }
}

__device__ __forceinline__ void store_unit(bf16x8* hi, bf16x8* lo, int64_t u, const float (&v)[8], int fmt = RSA_PF_BF16) {
  if (fmt == RSA_PF_F16) {
    f16x8_t h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float vj = v[j];
      asm("" : "+v"(vj));  // opaque: see conv_common.h, split2
      const _Float16 hb = (_Float16)vj;
      h[j] = hb;
      l[j] = (_Float16)(vj - (float)hb);
    }
    hi[u] = __builtin_bit_cast(bf16x8, h);
    if (lo != nullptr) lo[u] = __builtin_bit_cast(bf16x8, l);
    return;
  }
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 hb = (__bf16)v[j];
    h[j] = hb;
    l[j] = (__bf16)(v[j] - (float)hb);
  }
  hi[u] = h;
  if (lo != nullptr) lo[u] = l;
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ------------------------------------------------------------------------------------------------ rectangular window attention
__device__ __forceinline__ bf16x8 pack_bf16(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
  return r;
}

// One-product fp16 form (round 3, FMT = RSA_PF_F16): q / k / v planes and the output planes hold fp16 values, the contractions run on
// v_mfma_f32_32x32x16_f16.  The 16-byte units, the LDS images and the transpose reads are format-agnostic (16-bit elements); what
// changes is the matrix instruction, how the probabilities are packed and how the output is rounded.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <int FMT>
__device__ __forceinline__ f32x16 mfma32(const bf16x8 a, const bf16x8 b, const f32x16 c) {
  if constexpr (FMT == RSA_PF_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int FMT>
__device__ __forceinline__ bf16x8 pack16(const float (&v)[8]) {
  if constexpr (FMT == RSA_PF_F16) {
    f16x8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    return __builtin_bit_cast(bf16x8, h);
  } else {
    return pack_bf16(v);
  }
}
// four output values -> the 8-byte half of a plane unit (hi, and the rounding residual for a lo plane)
template <int FMT>
__device__ __forceinline__ void round4(const float (&v)[4], bf16x4& h, bf16x4& lo4) {
  if constexpr (FMT == RSA_PF_F16) {
    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
    f16x4_t hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float ve = v[e];
      asm("" : "+v"(ve));  // opaque: keeps the conversion of the residual from being fused with the arithmetic that made v (conv_common.h, split2)
      hh[e] = (_Float16)ve;
      ll[e] = (_Float16)(ve - (float)hh[e]);
    }
    h = __builtin_bit_cast(bf16x4, hh);
    lo4 = __builtin_bit_cast(bf16x4, ll);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hb = (__bf16)v[e];
      h[e] = hb;
      lo4[e] = (__bf16)(v[e] - (float)hb);
    }
  }
}

// One workgroup (4 waves) = one (window, head).  T = tiles of 32 tokens the LDS images hold.
//   self mode  (kwin == 0): keys = the window's own tokens (<= 32*T), staged once;
//   cross mode (kwin > 0) : keys = the (kwin_h x kwin_w) window that starts kpad pixels up-left of the query window (HAT's
//                           overlapping cross-attention: unfold with zero padding, archs/hat/arch.py:403-470), staged in chunks of
//                           32*T keys; the flash-style running max / sum / output of a wave's (up to two) query tiles live in
//                           registers across the chunks.
// K and V of a chunk are staged once in LDS; wave w owns query tiles w and w + 4.
template <int PROD, int T, int FMT = 0>
__global__ __launch_bounds__(256, 2) void rect_attention_kernel(const rsa_rect_attn_params p) {
  constexpr int NT = 32 * T;
  constexpr int KROW = 40;  // bf16 per K row = 80 bytes: ds_read_b128 of 16 consecutive rows touches every bank once
  constexpr int NHL = PROD == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 s_k[NHL][NT * KROW];
  __shared__ __attribute__((aligned(16))) __bf16 s_v[NHL][NT * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntok = p.win_h * p.win_w;
  const bool cross = p.kwin_h > 0;
  const int nkey = cross ? p.kwin_h * p.kwin_w : ntok;
  const int QT = (ntok + 31) >> 5;          // query tiles (<= 8)
  const int KT = cross ? (nkey + 31) >> 5 : T;  // key tiles in total (self mode: the instantiated T, padded keys carry -1e30 bias)
  const int QTB = cross ? QT : T;               // query tiles of the bias_frag layout
  const int nwx = p.Wp / p.win_w, nwy = p.Hp / p.win_h;
  const int64_t item = blockIdx.x;
  const int head = (int)(item % p.heads);
  const int64_t win = item / p.heads;
  const int wx = (int)(win % nwx);
  const int wy = (int)((win / nwx) % nwy);
  const int n = (int)(win / ((int64_t)nwx * nwy));

  // source pixel of QUERY token t: roll(-shift) on the PADDED grid, then partition; tokens that land on padding are zero
  auto token_pix = [&](int t, bool& valid) -> int64_t {
    const int ty = t / p.win_w, tx = t - ty * p.win_w;
    int sy = wy * p.win_h + ty + p.shift_h;
    int sx = wx * p.win_w + tx + p.shift_w;
    if (sy >= p.Hp) sy -= p.Hp;
    if (sx >= p.Wp) sx -= p.Wp;
    valid = t < ntok && sy < p.H && sx < p.W;
    return valid ? (int64_t)sy * p.W + sx : 0;
  };
  // source pixel of KEY token t: the query token's in self mode; a pixel of the enlarged window (zero outside the map) in cross mode
  auto key_pix = [&](int t, bool& valid) -> int64_t {
    if (!cross) return token_pix(t, valid);
    const int ky = t / p.kwin_w, kx = t - ky * p.kwin_w;
    const int sy = wy * p.win_h - p.kpad_h + ky, sx = wx * p.win_w - p.kpad_w + kx;
    valid = t < nkey && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
    return valid ? (int64_t)sy * p.W + sx : 0;
  };

  const bf16x8* qkv_hi = (const bf16x8*)p.qkv_hi + (int64_t)n * p.qkv_batch_stride;
  const bf16x8* qkv_lo = (PROD == 3) ? (const bf16x8*)p.qkv_lo + (int64_t)n * p.qkv_batch_stride : nullptr;
  const int64_t ps = p.qkv_plane_stride;
  const int slot = p.head0 + head;
  const int64_t q_plane0 = (int64_t)(0 * p.heads_total + slot) * 4;
  const int64_t k_plane0 = (int64_t)(1 * p.heads_total + slot) * 4;
  const int64_t v_plane0 = (int64_t)(2 * p.heads_total + slot) * 4;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int g16 = lane >> 4;
  const int li16 = lane & 15;
  const bool masked = p.shift_h > 0 && (wy == nwy - 1 || wx == nwx - 1);
  auto region = [&](int t) -> int {  // calculate_mask's region id of a window token on the shifted, padded grid
    const int tt = t < ntok ? t : 0;
    const int ty = tt / p.win_w, tx = tt - ty * p.win_w;
    const int gy = wy * p.win_h + ty, gx = wx * p.win_w + tx;
    const int ry = gy < p.Hp - p.win_h ? 0 : (gy < p.Hp - p.shift_h ? 1 : 2);
    const int rx = gx < p.Wp - p.win_w ? 0 : (gx < p.Wp - p.shift_w ? 1 : 2);
    return ry * 3 + rx;
  };
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

  // ---- per-wave state of its (up to two) query tiles ----
  bool qvalid[2];
  int64_t qpix[2];
  bf16x8 qh[2][2], ql[2][2];
  float m[2], l[2];
  f32x16 ot[2];
  int rq[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int qt = wave + 4 * qi;
    qpix[qi] = token_pix(32 * qt + lr, qvalid[qi]);
    if (qt >= QT) qvalid[qi] = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // Q fragments (B operand): query 32qt + lr, channels 16s + 8lh .. +7 = plane 2s + lh
      qh[qi][s] = zero8;
      ql[qi][s] = zero8;
      if (qvalid[qi]) {
        qh[qi][s] = qkv_hi[(q_plane0 + 2 * s + lh) * ps + qpix[qi]];
        if (PROD == 3) ql[qi][s] = qkv_lo[(q_plane0 + 2 * s + lh) * ps + qpix[qi]];
      }
    }
    rq[qi] = masked ? region(32 * qt + lr) : 0;
    m[qi] = -3.0e38f;
    l[qi] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[qi][r] = 0.f;
  }

  for (int kt0 = 0; kt0 < KT; kt0 += T) {
    // ---- stage K and V of key tiles [kt0, kt0 + T): thread = key token ----
    if (kt0 > 0) __syncthreads();  // everybody is done with the previous chunk
    if (tid < NT) {
      bool valid;
      const int64_t pix = key_pix(32 * kt0 + tid, valid);
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        bf16x8 kh = zero8, kl = zero8, vh = zero8, vl = zero8;
        if (valid) {
          kh = qkv_hi[(k_plane0 + pl) * ps + pix];
          vh = qkv_hi[(v_plane0 + pl) * ps + pix];
          if (PROD == 3) {
            kl = qkv_lo[(k_plane0 + pl) * ps + pix];
            vl = qkv_lo[(v_plane0 + pl) * ps + pix];
          }
        }
        *(bf16x8*)&s_k[0][tid * KROW + pl * 8] = kh;
        *(bf16x8*)&s_v[0][tid * 32 + pl * 8] = vh;
        if (PROD == 3) {
          *(bf16x8*)&s_k[NHL - 1][tid * KROW + pl * 8] = kl;
          *(bf16x8*)&s_v[NHL - 1][tid * 32 + pl * 8] = vl;
        }
      }
    }
    __syncthreads();
    const int ktn = (KT - kt0 < T) ? KT - kt0 : T;  // key tiles in this chunk

#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const int qt = wave + 4 * qi;
      if (qt >= QT) break;
      // ---- key tiles, flash style: S^T tile -> + bias / mask -> running max and sum -> P tile -> O^T += V^T P^T ----
      // accumulator element r of lane (lr, lh): key = 32kt + (r&3) + 8(r>>2) + 4lh, query = 32qt + lr (one query column per lane)
      for (int kt = 0; kt < ktn; ++kt) {
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = (32 * kt + lr) * KROW + (2 * s + lh) * 8;
          const bf16x8 kh = *(const bf16x8*)&s_k[0][off];
          if (PROD == 3) {
            const bf16x8 kl = *(const bf16x8*)&s_k[NHL - 1][off];
            a = mfma32<FMT>(kl, qh[qi][s], a);
            a = mfma32<FMT>(kh, ql[qi][s], a);
          }
          a = mfma32<FMT>(kh, qh[qi][s], a);
        }
        // + position bias (pre-gathered per (query tile, key tile), -1e30 on padded keys) + shift mask
        const f32x4* bf = (const f32x4*)(p.bias_frag + ((((int64_t)head * QTB + qt) * KT + kt0 + kt) * 64 + lane) * 16);
        float tm = -3.0e38f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = bf[g];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = g * 4 + e;
            float v = a[r] + b[e];
            if (masked) {
              const int key = 32 * (kt0 + kt) + (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (region(key) != rq[qi]) v += -100.f;
            }
            a[r] = v;
            tm = fmaxf(tm, v);
          }
        }
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        const float mn = fmaxf(m[qi], tm);
        const float alpha = expf(m[qi] - mn);  // 0 on the first tile
        m[qi] = mn;
        l[qi] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ot[qi][r] *= alpha;
          const float e = expf(a[r] - mn);
          a[r] = e;
          l[qi] += e;
        }
        // O^T[channel][query] += V^T P^T: A = V^T through transpose reads, B = the P tile straight from the accumulators
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 vh, vl;
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int row = 32 * kt + 16 * s + 8 * g2 + 4 * lh + (li16 >> 2);
            const int col = 16 * (g16 & 1) + 4 * (li16 & 3);
            const bf16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[0][row * 32 + col]);
#pragma unroll
            for (int e = 0; e < 4; ++e) vh[g2 * 4 + e] = th[e];
            if (PROD == 3) {
              const bf16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[NHL - 1][row * 32 + col]);
#pragma unroll
              for (int e = 0; e < 4; ++e) vl[g2 * 4 + e] = tl[e];
            }
          }
          float e8[8], r8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e8[j] = a[8 * s + j];
          const bf16x8 ph = pack16<FMT>(e8);
          if (PROD == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] = e8[j] - (float)ph[j];
            const bf16x8 pl = pack_bf16(r8);
            ot[qi] = mfma32<FMT>(vl, ph, ot[qi]);
            ot[qi] = mfma32<FMT>(vh, pl, ot[qi]);
          }
          ot[qi] = mfma32<FMT>(vh, ph, ot[qi]);
        }
      }
    }
  }

  // ---- normalise and store: lane owns query 32qt + lr, channels 8g + 4lh .. +3 ----
  char* out_hi = (char*)p.out_hi + (int64_t)n * p.out_batch_stride * 16;
  char* out_lo = (p.out_lo != nullptr) ? (char*)p.out_lo + (int64_t)n * p.out_batch_stride * 16 : nullptr;
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const float lsum = l[qi] + __shfl_xor(l[qi], 32);  // the two halves of a query column share m, so their partial sums just add
    if (!qvalid[qi]) continue;
    const float inv_l = 1.f / lsum;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 h, lo4;
      float v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[e] = ot[qi][g * 4 + e] * inv_l;
      round4<FMT>(v4, h, lo4);
      const int64_t off = (((int64_t)slot * 4 + g) * p.out_plane_stride + qpix[qi]) * 16 + lh * 8;
      *(bf16x4*)(out_hi + off) = h;
      if (out_lo != nullptr) *(bf16x4*)(out_lo + off) = lo4;
    }
  }
}

// ------------------------------------------------------------------------------------------------ wide-head window attention
// Self-attention of a (shifted) window whose heads are wider than 32 channels: head_dim <= 32*DC, DC = 2..4 (DRCT's dense groups run
// 2-6 heads of 46..122 channels over 16x16 windows: reference archs/drct/arch.py:102-198, 204-329).  A head slot is DC "chunks" of 4
// planes.  Same mathematics and fragment orders as rect_attention_kernel; what changes is the staging: the K / V images of a whole
// 256-token window no longer fit LDS at 128 channels, so keys are staged 64 at a time (two key tiles, all chunks) and the running
// max / sum / output of a wave's query tile carry over the stages, flash style.  8 waves, wave w owns query tile w.
template <int PROD, int DC, int FMT = 0>
__global__ __launch_bounds__(512, 1) void rect_attention_wide_kernel(const rsa_rect_attn_params p, int TB) {
  constexpr int TS = 2;          // key tiles per stage
  constexpr int NK = 32 * TS;    // keys per stage
  constexpr int KROW = 40;       // bf16 per K row of one chunk (80 bytes: conflict-free A-fragment reads)
  constexpr int NHL = PROD == 3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 s_k[NHL][DC][NK * KROW];
  __shared__ __attribute__((aligned(16))) __bf16 s_v[NHL][DC][NK * 32];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntok = p.win_h * p.win_w;
  const int QT = (ntok + 31) >> 5;
  const int KT = QT;
  const int nwx = p.Wp / p.win_w, nwy = p.Hp / p.win_h;
  const int64_t item = blockIdx.x;
  const int head = (int)(item % p.heads);
  const int64_t win = item / p.heads;
  const int wx = (int)(win % nwx);
  const int wy = (int)((win / nwx) % nwy);
  const int n = (int)(win / ((int64_t)nwx * nwy));

  auto token_pix = [&](int t, bool& valid) -> int64_t {  // roll(-shift) on the padded grid, then partition; padding tokens are zero
    const int ty = t / p.win_w, tx = t - ty * p.win_w;
    int sy = wy * p.win_h + ty + p.shift_h;
    int sx = wx * p.win_w + tx + p.shift_w;
    if (sy >= p.Hp) sy -= p.Hp;
    if (sx >= p.Wp) sx -= p.Wp;
    valid = t < ntok && sy < p.H && sx < p.W;
    return valid ? (int64_t)sy * p.W + sx : 0;
  };
  const bf16x8* qkv_hi = (const bf16x8*)p.qkv_hi + (int64_t)n * p.qkv_batch_stride;
  const bf16x8* qkv_lo = (PROD == 3) ? (const bf16x8*)p.qkv_lo + (int64_t)n * p.qkv_batch_stride : nullptr;
  const int64_t ps = p.qkv_plane_stride;
  const int slot = p.head0 + head;
  const int64_t q_plane0 = (int64_t)(0 * p.heads_total + slot) * 4 * DC;
  const int64_t k_plane0 = (int64_t)(1 * p.heads_total + slot) * 4 * DC;
  const int64_t v_plane0 = (int64_t)(2 * p.heads_total + slot) * 4 * DC;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  const int lr = lane & 31;
  const int lh = lane >> 5;
  const int g16 = lane >> 4;
  const int li16 = lane & 15;
  const bool masked = p.shift_h > 0 && (wy == nwy - 1 || wx == nwx - 1);
  auto region = [&](int t) -> int {
    const int tt = t < ntok ? t : 0;
    const int ty = tt / p.win_w, tx = tt - ty * p.win_w;
    const int gy = wy * p.win_h + ty, gx = wx * p.win_w + tx;
    const int ry = gy < p.Hp - p.win_h ? 0 : (gy < p.Hp - p.shift_h ? 1 : 2);
    const int rx = gx < p.Wp - p.win_w ? 0 : (gx < p.Wp - p.shift_w ? 1 : 2);
    return ry * 3 + rx;
  };
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

  // ---- this wave's query tile: Q fragments of every chunk stay in registers for the whole window ----
  const int qt = wave;
  const bool qlive = qt < QT;
  bool qvalid;
  const int64_t qpix = token_pix(32 * qt + lr, qvalid);
  if (!qlive) qvalid = false;
  bf16x8 qh[DC][2], ql[DC][2];
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qh[c][s] = zero8;
      ql[c][s] = zero8;
      if (qvalid) {
        qh[c][s] = qkv_hi[(q_plane0 + 4 * c + 2 * s + lh) * ps + qpix];
        if (PROD == 3) ql[c][s] = qkv_lo[(q_plane0 + 4 * c + 2 * s + lh) * ps + qpix];
      }
    }
  const int rq = masked ? region(32 * qt + lr) : 0;
  float m = -3.0e38f, l = 0.f;
  f32x16 ot[DC];
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[c][r] = 0.f;

  for (int kt0 = 0; kt0 < KT; kt0 += TS) {
    if (kt0 > 0) __syncthreads();
    // ---- stage keys [32*kt0, 32*kt0 + 64): thread = (key, plane group); 4*DC planes of K and of V each ----
    {
      const int key = tid & (NK - 1);
      bool valid;
      const int64_t pix = token_pix(32 * kt0 + key, valid);
      for (int pl = tid / NK; pl < 4 * DC; pl += 512 / NK) {
        const int c = pl >> 2, j = pl & 3;
        bf16x8 kh = zero8, kl = zero8, vh = zero8, vl = zero8;
        if (valid) {
          kh = qkv_hi[(k_plane0 + pl) * ps + pix];
          vh = qkv_hi[(v_plane0 + pl) * ps + pix];
          if (PROD == 3) {
            kl = qkv_lo[(k_plane0 + pl) * ps + pix];
            vl = qkv_lo[(v_plane0 + pl) * ps + pix];
          }
        }
        *(bf16x8*)&s_k[0][c][key * KROW + j * 8] = kh;
        *(bf16x8*)&s_v[0][c][key * 32 + j * 8] = vh;
        if (PROD == 3) {
          *(bf16x8*)&s_k[NHL - 1][c][key * KROW + j * 8] = kl;
          *(bf16x8*)&s_v[NHL - 1][c][key * 32 + j * 8] = vl;
        }
      }
    }
    __syncthreads();
    if (!qlive) continue;  // (wave-uniform; the barriers above are reached by every wave)
    const int ktn = (KT - kt0 < TS) ? KT - kt0 : TS;
    for (int kt = 0; kt < ktn; ++kt) {
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = (32 * kt + lr) * KROW + (2 * s + lh) * 8;
          const bf16x8 kh = *(const bf16x8*)&s_k[0][c][off];
          if (PROD == 3) {
            const bf16x8 kl = *(const bf16x8*)&s_k[NHL - 1][c][off];
            a = mfma32<FMT>(kl, qh[c][s], a);
            a = mfma32<FMT>(kh, ql[c][s], a);
          }
          a = mfma32<FMT>(kh, qh[c][s], a);
        }
      const f32x4* bf = (const f32x4*)(p.bias_frag + ((((int64_t)head * TB + qt) * TB + kt0 + kt) * 64 + lane) * 16);
      float tm = -3.0e38f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = bf[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g * 4 + e;
          float v = a[r] + b[e];
          if (masked) {
            const int key = 32 * (kt0 + kt) + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (region(key) != rq) v += -100.f;
          }
          a[r] = v;
          tm = fmaxf(tm, v);
        }
      }
      tm = fmaxf(tm, __shfl_xor(tm, 32));
      const float mn = fmaxf(m, tm);
      const float alpha = expf(m - mn);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int c = 0; c < DC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[c][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(a[r] - mn);
        a[r] = e;
        l += e;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float e8[8], r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e8[j] = a[8 * s + j];
        const bf16x8 ph = pack16<FMT>(e8);
        bf16x8 pl8 = zero8;
        if (PROD == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) r8[j] = e8[j] - (float)ph[j];
          pl8 = pack_bf16(r8);
        }
#pragma unroll
        for (int c = 0; c < DC; ++c) {
          bf16x8 vh, vl;
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int row = 32 * kt + 16 * s + 8 * g2 + 4 * lh + (li16 >> 2);
            const int col = 16 * (g16 & 1) + 4 * (li16 & 3);
            const bf16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[0][c][row * 32 + col]);
#pragma unroll
            for (int e = 0; e < 4; ++e) vh[g2 * 4 + e] = th[e];
            if (PROD == 3) {
              const bf16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[NHL - 1][c][row * 32 + col]);
#pragma unroll
              for (int e = 0; e < 4; ++e) vl[g2 * 4 + e] = tl[e];
            }
          }
          if (PROD == 3) {
            ot[c] = mfma32<FMT>(vl, ph, ot[c]);
            ot[c] = mfma32<FMT>(vh, pl8, ot[c]);
          }
          ot[c] = mfma32<FMT>(vh, ph, ot[c]);
        }
      }
    }
  }

  // ---- normalise and store: lane owns query 32qt + lr, channels (chunk c) 8g + 4lh .. +3 ----
  const float lsum = l + __shfl_xor(l, 32);
  if (!qvalid) return;
  char* out_hi = (char*)p.out_hi + (int64_t)n * p.out_batch_stride * 16;
  char* out_lo = (p.out_lo != nullptr) ? (char*)p.out_lo + (int64_t)n * p.out_batch_stride * 16 : nullptr;
  const float inv_l = 1.f / lsum;
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 h, lo4;
      float v4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[e] = ot[c][g * 4 + e] * inv_l;
      round4<FMT>(v4, h, lo4);
      const int64_t off = ((((int64_t)slot * DC + c) * 4 + g) * p.out_plane_stride + qpix) * 16 + lh * 8;
      *(bf16x4*)(out_hi + off) = h;
      if (out_lo != nullptr) *(bf16x4*)(out_lo + off) = lo4;
    }
}

// ------------------------------------------------------------------------------------------------ channel attention weights
constexpr int CA_TOK = 512;                  // tokens per partial Gram matrix (round 4: 2048 -> 512; at 512^2 x 6 heads the grid was 768 single-wave
                                             // workgroups, less than one wave per SIMD, each with eight waited-for loads per 64 tokens: 240 us)
constexpr int CA_REC = 32 * 32 + 64;         // floats per partial: G[32][32], |q|^2[32], |k|^2[32]

// grid (chunks, heads, batch), one wave: partial G = sum_tokens q^T k over CA_TOK tokens, f32 FMA on hi+lo reconstructions.
__global__ __launch_bounds__(64) void channel_gram_kernel(const rsa_channel_attn_params p, int chunks) {
  __shared__ float s_q[64][33];
  __shared__ float s_k[64][36];  // rows 16-byte aligned for the float4 broadcasts
  const int lane = threadIdx.x;
  const int chunk = blockIdx.x, head = blockIdx.y, n = blockIdx.z;
  const int64_t HW = (int64_t)p.H * p.W;
  const bf16x8* q_hi = (const bf16x8*)p.q_hi + (int64_t)n * p.batch_stride + (int64_t)head * 4 * p.plane_stride;
  const bf16x8* k_hi = (const bf16x8*)p.k_hi + (int64_t)n * p.batch_stride + (int64_t)head * 4 * p.plane_stride;
  const bf16x8* q_lo = p.q_lo ? (const bf16x8*)p.q_lo + (int64_t)n * p.batch_stride + (int64_t)head * 4 * p.plane_stride : nullptr;
  const bf16x8* k_lo = p.k_lo ? (const bf16x8*)p.k_lo + (int64_t)n * p.batch_stride + (int64_t)head * 4 * p.plane_stride : nullptr;
  const int i = lane & 31, half = lane >> 5;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  float nq = 0.f, nk = 0.f;
  const int64_t t0 = (int64_t)chunk * CA_TOK;
  for (int it = 0; it < CA_TOK / 64; ++it) {
    const int64_t tok = t0 + it * 64 + lane;
    __syncthreads();
    // all eight units of the token first (tokens past the end re-read the last one and are zeroed), then the conversions
    const int64_t tokc = tok < HW ? tok : HW - 1;
    const float keep = tok < HW ? 1.f : 0.f;
    bf16x8 rqh[4], rkh[4], rql[4], rkl[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      rqh[pl] = q_hi[pl * p.plane_stride + tokc];
      rkh[pl] = k_hi[pl * p.plane_stride + tokc];
      rql[pl] = q_lo != nullptr ? q_lo[pl * p.plane_stride + tokc] : rqh[pl];
      rkl[pl] = k_lo != nullptr ? k_lo[pl * p.plane_stride + tokc] : rkh[pl];
    }
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      float qv[8], kv[8];
      raw_f32(rqh[pl], rql[pl], q_lo != nullptr, qv, p.fmt);
      raw_f32(rkh[pl], rkl[pl], k_lo != nullptr, kv, p.fmt);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s_q[lane][pl * 8 + j] = qv[j] * keep;
        s_k[lane][pl * 8 + j] = kv[j] * keep;
      }
    }
    __syncthreads();
    for (int t = 0; t < 64; ++t) {
      const float qi = s_q[t][i];
      const float ki = s_k[t][i];
      nq += qi * qi;
      nk += ki * ki;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 kk = *(const f32x4*)&s_k[t][16 * half + 4 * j4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * j4 + e] += qi * kk[e];
      }
    }
  }
  float* rec = p.workspace + (((int64_t)n * p.heads + head) * chunks + chunk) * CA_REC;
#pragma unroll
  for (int j = 0; j < 16; ++j) rec[i * 32 + 16 * half + j] = acc[j];
  if (half == 0) {
    rec[1024 + i] = nq;
    rec[1056 + i] = nk;
  }
}

// grid (heads, batch), 1024 threads = (i, j): fixed-order sum of the partials, normalise, temperature, softmax over j, pack.
__global__ __launch_bounds__(1024) void channel_attn_finish_kernel(const rsa_channel_attn_params p, int chunks) {
  const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
  const int head = blockIdx.x, n = blockIdx.y;
  const float* rec = p.workspace + ((int64_t)n * p.heads + head) * chunks * CA_REC;
  float g = 0.f, nq = 0.f, nk = 0.f;
#pragma unroll 16
  for (int c = 0; c < chunks; ++c) {  // fixed order: the result does not depend on the schedule
    g += rec[(int64_t)c * CA_REC + i * 32 + j];
    nq += rec[(int64_t)c * CA_REC + 1024 + i];
    nk += rec[(int64_t)c * CA_REC + 1056 + j];
  }
  const int hd = p.head_dim;
  const bool live = i < hd && j < hd;
  // F.normalize: x / max(||x||, 1e-12) on both operands, then * temperature
  float v = g / (fmaxf(sqrtf(nq), 1e-12f) * fmaxf(sqrtf(nk), 1e-12f)) * p.temperature[head];
  if (!live) v = -3.0e38f;
  float m = v;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));  // rows are 32-lane halves of a wave
  float e = live ? expf(v - m) : 0.f;
  float l = e;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) l += __shfl_xor(l, o);
  const float a = (live && l > 0.f) ? e / l : 0.f;
  // packed k1 weights [q = head][tap 0][ct][hl][lane][8]: cout = 32 head + i -> ct = 2 head + (i >> 4); cin = 32 head + j
  const int nct = 2 * p.heads;
  const int nhl = p.products == 3 ? 2 : 1;
  const int64_t blob = (int64_t)p.heads * nct * nhl * 64 * 8;
  __bf16* w = (__bf16*)p.w_packed + (int64_t)n * blob;
  const int ct = 2 * head + (i >> 4);
  const int ln = (j >> 3) * 16 + (i & 15);
  const int64_t base = (((int64_t)head * nct + ct) * nhl) * 64 * 8 + ln * 8 + (j & 7);
  if (p.fmt == RSA_PF_F16) {  // the blob is an opaque 16-bit container: fp16 fragments for the one-product fp16 form
    const _Float16 hf = (_Float16)a;
    ((_Float16*)w)[base] = hf;
    if (nhl == 2) ((_Float16*)w)[base + 64 * 8] = (_Float16)(a - (float)hf);
  } else {
    const __bf16 hb = (__bf16)a;
    w[base] = hb;
    if (nhl == 2) w[base + 64 * 8] = (__bf16)(a - (float)hb);
  }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3
// thread = (pixel, plane of 8 channels); grid (tiles of DW_TW x DW_TH pixels, planes, batch).  Round 4: a workgroup is a 32 x 8 pixel TILE, not
// 256 consecutive pixels of a row: the three rows a row-shaped workgroup reads were fetched by three workgroups on three XCDs (consecutive
// workgroups go round-robin over the XCDs), i.e. three times from beyond the L2s; a tile reads a 34 x 10 halo region once (1.33 x).
constexpr int DW_TW = 32, DW_TH = 8;
template <bool NORM, bool GELU, int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(const rsa_dwconv_params p) {
  constexpr int R = KS / 2, KK = KS * KS;
  const int64_t HW = (int64_t)p.H * p.W;
  const int tiles_x = (p.W + DW_TW - 1) / DW_TW;
  const int ty = (int)blockIdx.x / tiles_x, tx = (int)blockIdx.x - ty * tiles_x;
  const int x = tx * DW_TW + (int)(threadIdx.x & (DW_TW - 1)), y = ty * DW_TH + (int)(threadIdx.x / DW_TW);
  const int pl = blockIdx.y, n = blockIdx.z;
  if (x >= p.W || y >= p.H) return;
  const int64_t pix = (int64_t)y * p.W + x;
  const bf16x8* in_hi = (const bf16x8*)p.in_hi + (int64_t)n * p.in_batch_stride + (int64_t)pl * p.in_plane_stride;
  const bf16x8* in_lo = p.in_lo ? (const bf16x8*)p.in_lo + (int64_t)n * p.in_batch_stride + (int64_t)pl * p.in_plane_stride : nullptr;
  const float* wt = p.weight + pl * 8 * KK;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = p.bias[pl * 8 + j];
  float gam[8], bet[8];
  if (NORM) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gam[j] = p.gamma[pl * 8 + j];
      bet[j] = p.beta[pl * 8 + j];
    }
  }
  // One ROW of taps at a time, every load of the row issued before its first use: out-of-map taps read a clamped (valid) address and are
  // multiplied by 0 (zero padding applies to the normalised input).  The first form tested the bounds around each tap: a divergent block
  // with its own wait per tap -- nine dependent round trips (253 us for 23 planes at 512^2; profiles/r04_v_dat_stream_kernels.txt).
  const bool has_lo = in_lo != nullptr;  // uniform
#pragma unroll
  for (int dy = -R; dy <= R; ++dy) {
    bf16x8 rh[KS], rl[KS];
    float mean[KS], rstd[KS], inside[KS];
    const int yy = min(max(y + dy, 0), p.H - 1);
    const bool yok = (unsigned)(y + dy) < (unsigned)p.H;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      const int xx = min(max(x + t - R, 0), p.W - 1);
      const int64_t q = (int64_t)yy * p.W + xx;
      rh[t] = in_hi[q];
      if (has_lo) rl[t] = in_lo[q];
      if (NORM) {
        const float2 st = *(const float2*)&p.stats[((int64_t)n * HW + q) * 2];
        mean[t] = st.x, rstd[t] = st.y;
      }
      inside[t] = (yok && (unsigned)(x + t - R) < (unsigned)p.W) ? 1.f : 0.f;
    }
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      float v[8];
      if (p.fmt == RSA_PF_F16) {
        const f16x8_t hf = __builtin_bit_cast(f16x8_t, rh[t]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)hf[j];
        if (has_lo) {
          const f16x8_t lf = __builtin_bit_cast(f16x8_t, rl[t]);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)lf[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)rh[t][j];
        if (has_lo) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += (float)rl[t][j];
        }
      }
      if (NORM) {
        const float a = rstd[t] * inside[t], b = -mean[t] * a;  // (v - mean) * rstd * gamma + beta, zero outside the map
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] * a + b) * gam[j] + bet[j] * inside[t];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= inside[t];
      }
      const int tap = (dy + R) * KS + t;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wt[j * KK + tap] * v[j];
    }
  }
  if (GELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = gelu_erf(acc[j]);
  }
  if (p.mul_hi != nullptr) {
    const bf16x8* m_hi = (const bf16x8*)p.mul_hi + (int64_t)n * p.mul_batch_stride + (int64_t)pl * p.mul_plane_stride;
    const bf16x8* m_lo = p.mul_lo ? (const bf16x8*)p.mul_lo + (int64_t)n * p.mul_batch_stride + (int64_t)pl * p.mul_plane_stride : nullptr;
    float mv[8];
    unit_f32(m_hi, m_lo, pix, mv, p.fmt);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= mv[j];
  }
  bf16x8* o_hi = (bf16x8*)p.out_hi + (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride;
  bf16x8* o_lo = p.out_lo ? (bf16x8*)p.out_lo + (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride : nullptr;
  store_unit(o_hi, o_lo, pix, acc, p.fmt);
}

// ------------------------------------------------------------------------------------------------ per-pixel LN statistics
__global__ __launch_bounds__(256) void plane_stats_kernel(const bf16x8* in_hi, const bf16x8* in_lo, int64_t plane_stride, int64_t batch_stride,
                                                          int64_t HW, int C, float eps, float* stats, int fmt) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  const bf16x8* hi = in_hi + (int64_t)n * batch_stride;
  const bf16x8* lo = in_lo ? in_lo + (int64_t)n * batch_stride : nullptr;
  const int planes = (C + 7) / 8;
  // ONE pass (round 4): nothing is read twice -- the second pass of the two-pass form came from beyond the L2 again (12 MB of resident working
  // set per XCD).  Each plane's 8 channels give an exact two-pass (mean, M2) in registers; the planes are combined with the pairwise update
  // of Chan et al. (mean += d * nb / n, M2 += M2b + d^2 * na * nb / n), which has no cancellation whatever the offset of the data.
  // Everything is done on x - s0, s0 = the mean of the first plane (an exact subtraction for data near s0), so the rounding of the running
  // mean scales with the spread of the data, not with its offset.
  float mean = 0.f, m2 = 0.f, s0 = 0.f;  // mean: of x - s0
  int cnt = 0;
  for (int pl = 0; pl < planes; ++pl) {
    float v[8];
    unit_f32(hi, lo, pl * plane_stride + pix, v, fmt);
    const int nb = min(8, C - pl * 8);  // uniform
    if (pl == 0) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += j < nb ? v[j] : 0.f;
      s0 = s / (float)nb;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] -= s0;
    float sb = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sb += j < nb ? v[j] : 0.f;
    const float mb = sb / (float)nb;
    float m2b = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[j] - mb;
      m2b += j < nb ? d * d : 0.f;
    }
    const int nn = cnt + nb;
    const float d = mb - mean;
    mean += d * ((float)nb / (float)nn);
    m2 += m2b + d * d * ((float)cnt * (float)nb / (float)nn);
    cnt = nn;
  }
  const float var = m2 / (float)C;
  mean += s0;
  stats[((int64_t)n * HW + pix) * 2] = mean;
  stats[((int64_t)n * HW + pix) * 2 + 1] = 1.f / sqrtf(var + eps);
}

// ------------------------------------------------------------------------------------------------ channel gate
constexpr int CG_PIX = 4096;  // pixels per partial sum

// grid (chunks, planes, batch), 256 threads: partial[b][plane][chunk][8]
__global__ __launch_bounds__(256) void channel_sum_kernel(const rsa_channel_gate_params p, int chunks) {
  __shared__ float s_red[4][8];
  const int chunk = blockIdx.x, pl = blockIdx.y, n = blockIdx.z;
  const int64_t HW = (int64_t)p.H * p.W;
  const bf16x8* hi = (const bf16x8*)p.in_hi + (int64_t)n * p.in_batch_stride + (int64_t)pl * p.in_plane_stride;
  const bf16x8* lo = p.in_lo ? (const bf16x8*)p.in_lo + (int64_t)n * p.in_batch_stride + (int64_t)pl * p.in_plane_stride : nullptr;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int it = 0; it < CG_PIX / 256; ++it) {
    const int64_t pix = (int64_t)chunk * CG_PIX + it * 256 + threadIdx.x;
    if (pix < HW) {
      float v[8];
      unit_f32(hi, lo, pix, v, p.fmt);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc[j] += __shfl_xor(acc[j], o);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_red[wave][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int j = threadIdx.x;
    p.workspace[(((int64_t)n * p.planes + pl) * chunks + chunk) * 8 + j] = (s_red[0][j] + s_red[1][j]) + (s_red[2][j] + s_red[3][j]);
  }
}

// grid (batch), 256 threads
__global__ __launch_bounds__(256) void channel_gate_kernel(const rsa_channel_gate_params p, int chunks) {
  __shared__ float s_mean[512];
  __shared__ float s_hid[128];
  const int n = blockIdx.x;
  const int C = p.planes * 8;
  const float inv = 1.f / (float)((int64_t)p.H * p.W);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* part = p.workspace + (((int64_t)n * p.planes + (c >> 3)) * chunks) * 8 + (c & 7);
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * 8];
    s_mean[c] = s * inv;
  }
  __syncthreads();
  if (threadIdx.x < p.hidden) {
    const int k = threadIdx.x;
    float s = p.b1[k];
    for (int c = 0; c < C; ++c) s += p.w1[(int64_t)k * C + c] * s_mean[c];
    s_hid[k] = p.relu ? fmaxf(s, 0.f) : gelu_erf(s);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = p.b2[c];
    for (int k = 0; k < p.hidden; ++k) s += p.w2[(int64_t)c * p.hidden + k] * s_hid[k];
    p.gate[(int64_t)n * C + c] = p.relu == 2 ? fminf(fmaxf(s * (1.f / 6.f) + 0.5f, 0.f), 1.f) : sigmoidf(s);
  }
}

// ------------------------------------------------------------------------------------------------ AIM combine
// thread = pixel; grid (ceil(HW/256), batch)
__global__ __launch_bounds__(256) void aim_kernel(const rsa_aim_params p) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  const int C = p.planes * 8;
  const bf16x8* a_hi = (const bf16x8*)p.att_hi + (int64_t)n * p.att_batch_stride;
  const bf16x8* a_lo = p.att_lo ? (const bf16x8*)p.att_lo + (int64_t)n * p.att_batch_stride : nullptr;
  const bf16x8* c_hi = (const bf16x8*)p.conv_hi + (int64_t)n * p.conv_batch_stride;
  const bf16x8* c_lo = p.conv_lo ? (const bf16x8*)p.conv_lo + (int64_t)n * p.conv_batch_stride : nullptr;
  const bf16x8* s_hi = p.mode == 0 ? a_hi : c_hi;
  const bf16x8* s_lo = p.mode == 0 ? a_lo : c_lo;
  const int64_t s_ps = p.mode == 0 ? p.att_plane_stride : p.conv_plane_stride;
  float h[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) h[k] = 0.f;
  for (int pl = 0; pl < p.planes; ++pl) {
    float v[8];
    unit_f32(s_hi, s_lo, pl * s_ps + pix, v, p.fmt);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < p.hidden) {
        const float* w = p.w1 + (int64_t)k * C + pl * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[k] += w[j] * v[j];
      }
    }
  }
  float s = p.b2;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < p.hidden) s += p.w2[k] * gelu_erf(h[k] + p.b1[k]);
  const float sg = sigmoidf(s);
  const float* gate = p.gate + (int64_t)n * C;
  bf16x8* o_hi = (bf16x8*)p.out_hi + (int64_t)n * p.out_batch_stride;
  bf16x8* o_lo = p.out_lo ? (bf16x8*)p.out_lo + (int64_t)n * p.out_batch_stride : nullptr;
  for (int pl = 0; pl < p.planes; ++pl) {
    float a[8], c[8], o[8];
    unit_f32(a_hi, a_lo, pl * p.att_plane_stride + pix, a, p.fmt);
    unit_f32(c_hi, c_lo, pl * p.conv_plane_stride + pix, c, p.fmt);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = gate[pl * 8 + j];
      o[j] = p.mode == 0 ? a[j] * g + sg * c[j] : a[j] * sg + c[j] * g;
    }
    store_unit(o_hi, o_lo, pl * p.out_plane_stride + pix, o, p.fmt);
  }
}

// ------------------------------------------------------------------------------------------------ gated add
// out[b][c][p] = base[b][c][p] + x[b][c][p] * gate[b][c] * scale on f32 maps (HAT: shortcut + CAB(x) * 0.01, arch.py:345, with the
// channel attention of the CAB as `gate`); thread = (pixel, plane of 8 channels); grid (ceil(HW/256), planes, batch)
__global__ __launch_bounds__(256) void gated_add_kernel(const bf16x8* x_hi, const bf16x8* x_lo, int64_t plane_stride, int64_t batch_stride, int64_t HW,
                                                        int C, const float* gate, float scale, const f32x4* base, f32x4* out) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int pl = blockIdx.y, n = blockIdx.z;
  if (pix >= HW) return;
  const int p4 = (C + 3) >> 2;
  float v[8];
  unit_f32(x_hi + (int64_t)n * batch_stride + (int64_t)pl * plane_stride, x_lo ? x_lo + (int64_t)n * batch_stride + (int64_t)pl * plane_stride : nullptr,
           pix, v);
  const float* g = gate + (int64_t)n * (((C + 7) >> 3) << 3) + pl * 8;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int grp = pl * 2 + half;
    if (grp >= p4) continue;
    const int64_t i = ((int64_t)n * p4 + grp) * HW + pix;
    const f32x4 b = base[i];
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = b[r] + v[half * 4 + r] * g[half * 4 + r] * scale;
    out[i] = o;
  }
}

static bool misaligned(const void* a) { return ((uintptr_t)a & 15) != 0; }

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_rect_attention(const rsa_rect_attn_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "rect_attention: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->heads < 1 || p->head0 < 0 || p->head0 + p->heads > p->heads_total)
    return set_error(RSA_E_ARG, "rect_attention: bad geometry");
  if (p->win_h < 1 || p->win_w < 1 || p->win_h * p->win_w > 256) return set_error(RSA_E_UNSUPPORTED, "rect_attention: window must hold 1..256 tokens");
  if (p->Hp < p->H || p->Wp < p->W || p->Hp % p->win_h || p->Wp % p->win_w)
    return set_error(RSA_E_ARG, "rect_attention: padded grid must cover the map and be a multiple of the window");
  if (p->shift_h < 0 || p->shift_h >= p->win_h || p->shift_w < 0 || p->shift_w >= p->win_w || ((p->shift_h == 0) != (p->shift_w == 0)))
    return set_error(RSA_E_ARG, "rect_attention: shifts must both be 0 or both be in (0, window)");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "rect_attention: products must be 1 or 3");
  if ((p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) || p->reserved0 != 0 || (p->fmt == RSA_PF_F16 && p->products != 1))
    return set_error(RSA_E_ARG, "rect_attention: fmt must be an rsa_plane_fmt (fp16 planes: the one-product form only)");
  if (!p->qkv_hi || !p->bias_frag || !p->out_hi || (p->products == 3 && !p->qkv_lo)) return set_error(RSA_E_ARG, "rect_attention: null pointer");
  if (misaligned(p->qkv_hi) || misaligned(p->qkv_lo) || misaligned(p->bias_frag) || misaligned(p->out_hi) || misaligned(p->out_lo))
    return set_error(RSA_E_ALIGN, "rect_attention: pointers must be 16-byte aligned");
  const bool cross = p->kwin_h > 0 || p->kwin_w > 0;
  const int DCh = p->head_chunks < 1 ? 1 : p->head_chunks;
  if (DCh > 4) return set_error(RSA_E_UNSUPPORTED, "rect_attention: head_chunks must be 1..4 (head_dim <= 128)");
  if (DCh > 1 && cross) return set_error(RSA_E_UNSUPPORTED, "rect_attention: wide heads (head_chunks > 1) are self-attention only");
  if (cross) {
    if (p->kwin_h < 1 || p->kwin_w < 1 || p->kpad_h < 0 || p->kpad_w < 0) return set_error(RSA_E_ARG, "rect_attention: bad key window");
    if (p->shift_h != 0 || p->Hp != p->H || p->Wp != p->W) return set_error(RSA_E_UNSUPPORTED, "rect_attention: the cross-window mode takes no shift and no padding");
  }
  const int ntok = cross ? p->kwin_h * p->kwin_w : p->win_h * p->win_w;
  const int tiles = (ntok + 31) / 32;
  const int T = tiles <= 1 ? 1 : tiles <= 2 ? 2 : tiles <= 4 ? 4 : 8;
  const int64_t blocks = (int64_t)p->batch * (p->Hp / p->win_h) * (p->Wp / p->win_w) * p->heads;
  if (blocks > 0x7fffffff) return set_error(RSA_E_UNSUPPORTED, "rect_attention: too many windows");
  hipStream_t s = (hipStream_t)stream;
  if (DCh > 1) {
    const dim3 gridw((unsigned)blocks), blockw(512);
#define RSA_RAW(PROD, DC) hipLaunchKernelGGL((rect_attention_wide_kernel<PROD, DC>), gridw, blockw, 0, s, *p, T)
    if (p->products == 3) {
      if (DCh == 2) RSA_RAW(3, 2); else if (DCh == 3) RSA_RAW(3, 3); else RSA_RAW(3, 4);
    } else if (p->fmt == RSA_PF_F16) {
#define RSA_RAWH(DC) hipLaunchKernelGGL((rect_attention_wide_kernel<1, DC, RSA_PF_F16>), gridw, blockw, 0, s, *p, T)
      if (DCh == 2) RSA_RAWH(2); else if (DCh == 3) RSA_RAWH(3); else RSA_RAWH(4);
#undef RSA_RAWH
    } else {
      if (DCh == 2) RSA_RAW(1, 2); else if (DCh == 3) RSA_RAW(1, 3); else RSA_RAW(1, 4);
    }
#undef RSA_RAW
    const hipError_t rcw = hipGetLastError();
    return rcw ? set_error(rcw, "rect_attention: launch failed") : RSA_OK;
  }
  const dim3 grid((unsigned)blocks), block(256);
#define RSA_RA(PROD, TT) hipLaunchKernelGGL((rect_attention_kernel<PROD, TT>), grid, block, 0, s, *p)
  if (p->products == 3) {
    if (T == 1) RSA_RA(3, 1); else if (T == 2) RSA_RA(3, 2); else if (T == 4) RSA_RA(3, 4); else RSA_RA(3, 8);
  } else if (p->fmt == RSA_PF_F16) {
#define RSA_RAH(TT) hipLaunchKernelGGL((rect_attention_kernel<1, TT, RSA_PF_F16>), grid, block, 0, s, *p)
    if (T == 1) RSA_RAH(1); else if (T == 2) RSA_RAH(2); else if (T == 4) RSA_RAH(4); else RSA_RAH(8);
#undef RSA_RAH
  } else {
    if (T == 1) RSA_RA(1, 1); else if (T == 2) RSA_RA(1, 2); else if (T == 4) RSA_RA(1, 4); else RSA_RA(1, 8);
  }
#undef RSA_RA
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "rect_attention: launch failed") : RSA_OK;
}

extern "C" int64_t rsa_channel_attn_workspace_bytes(int32_t batch, int32_t H, int32_t W, int32_t heads) {
  if (batch < 1 || H < 1 || W < 1 || heads < 1) return RSA_E_ARG;
  const int64_t chunks = ((int64_t)H * W + CA_TOK - 1) / CA_TOK;
  return (int64_t)batch * heads * chunks * CA_REC * 4;
}

extern "C" int rsa_channel_attention_weights(const rsa_channel_attn_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "channel_attention_weights: null params");
  if (p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "channel_attention_weights: fmt must be an rsa_plane_fmt");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->heads < 1 || p->head_dim < 1 || p->head_dim > 32)
    return set_error(RSA_E_ARG, "channel_attention_weights: bad geometry (head_dim must be 1..32)");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "channel_attention_weights: products must be 1 or 3");
  if (!p->q_hi || !p->k_hi || !p->temperature || !p->workspace || !p->w_packed) return set_error(RSA_E_ARG, "channel_attention_weights: null pointer");
  if (misaligned(p->q_hi) || misaligned(p->q_lo) || misaligned(p->k_hi) || misaligned(p->k_lo) || misaligned(p->workspace) || misaligned(p->w_packed))
    return set_error(RSA_E_ALIGN, "channel_attention_weights: pointers must be 16-byte aligned");
  const int64_t chunks = ((int64_t)p->H * p->W + CA_TOK - 1) / CA_TOK;
  if (chunks > 65535 || p->heads > 65535 || p->batch > 65535) return set_error(RSA_E_UNSUPPORTED, "channel_attention_weights: map too large");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(channel_gram_kernel, dim3((unsigned)chunks, (unsigned)p->heads, (unsigned)p->batch), dim3(64), 0, s, *p, (int)chunks);
  hipLaunchKernelGGL(channel_attn_finish_kernel, dim3((unsigned)p->heads, (unsigned)p->batch), dim3(1024), 0, s, *p, (int)chunks);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "channel_attention_weights: launch failed") : RSA_OK;
}

extern "C" int rsa_dwconv3x3(const rsa_dwconv_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "dwconv3x3: null params");
  if (p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "dwconv3x3: fmt must be an rsa_plane_fmt");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->planes < 1 || p->planes > 65535 || p->batch > 65535) return set_error(RSA_E_ARG, "dwconv3x3: bad geometry");
  if (p->act != RSA_ACT_NONE && p->act != RSA_ACT_GELU) return set_error(RSA_E_UNSUPPORTED, "dwconv3x3: act must be none or gelu");
  if (!p->in_hi || !p->weight || !p->bias || !p->out_hi) return set_error(RSA_E_ARG, "dwconv3x3: null pointer");
  if (p->stats && (!p->gamma || !p->beta)) return set_error(RSA_E_ARG, "dwconv3x3: stats need gamma and beta");
  if (misaligned(p->in_hi) || misaligned(p->in_lo) || misaligned(p->mul_hi) || misaligned(p->mul_lo) || misaligned(p->out_hi) || misaligned(p->out_lo))
    return set_error(RSA_E_ALIGN, "dwconv3x3: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)p->H * p->W;
  const int64_t tiles = (int64_t)((p->W + DW_TW - 1) / DW_TW) * ((p->H + DW_TH - 1) / DW_TH);
  if (tiles > 0x7fffffff) return set_error(RSA_E_UNSUPPORTED, "dwconv3x3: map too large");
  const dim3 grid((unsigned)tiles, (unsigned)p->planes, (unsigned)p->batch), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool gelu = p->act == RSA_ACT_GELU;
  if (p->stats) {
    if (gelu) hipLaunchKernelGGL((dwconv_kernel<true, true, 3>), grid, block, 0, s, *p);
    else hipLaunchKernelGGL((dwconv_kernel<true, false, 3>), grid, block, 0, s, *p);
  } else {
    if (gelu) hipLaunchKernelGGL((dwconv_kernel<false, true, 3>), grid, block, 0, s, *p);
    else hipLaunchKernelGGL((dwconv_kernel<false, false, 3>), grid, block, 0, s, *p);
  }
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "dwconv3x3: launch failed") : RSA_OK;
}

extern "C" int rsa_dwconv5x5(const rsa_dwconv_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "dwconv5x5: null params");
  if (p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "dwconv5x5: fmt must be an rsa_plane_fmt");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->planes < 1 || p->planes > 65535 || p->batch > 65535) return set_error(RSA_E_ARG, "dwconv5x5: bad geometry");
  if (p->act != RSA_ACT_NONE || p->stats != nullptr) return set_error(RSA_E_UNSUPPORTED, "dwconv5x5: no activation / normalisation variant is compiled");
  if (!p->in_hi || !p->weight || !p->bias || !p->out_hi) return set_error(RSA_E_ARG, "dwconv5x5: null pointer");
  if (misaligned(p->in_hi) || misaligned(p->in_lo) || misaligned(p->mul_hi) || misaligned(p->mul_lo) || misaligned(p->out_hi) || misaligned(p->out_lo))
    return set_error(RSA_E_ALIGN, "dwconv5x5: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)p->H * p->W;
  const int64_t tiles = (int64_t)((p->W + DW_TW - 1) / DW_TW) * ((p->H + DW_TH - 1) / DW_TH);
  if (tiles > 0x7fffffff) return set_error(RSA_E_UNSUPPORTED, "dwconv5x5: map too large");
  hipLaunchKernelGGL((dwconv_kernel<false, false, 5>), dim3((unsigned)tiles, (unsigned)p->planes, (unsigned)p->batch), dim3(256), 0, (hipStream_t)stream, *p);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "dwconv5x5: launch failed") : RSA_OK;
}

extern "C" int rsa_plane_stats_fmt(const void* in_hi, const void* in_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W,
                                   int32_t C, float eps, int32_t fmt, float* stats, void* stream) {
  if (!in_hi || !stats || batch < 1 || batch > 65535 || H < 1 || W < 1 || C < 1) return set_error(RSA_E_ARG, "plane_stats: bad argument");
  if (fmt != RSA_PF_BF16 && fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "plane_stats: fmt must be an rsa_plane_fmt");
  if (misaligned(in_hi) || misaligned(in_lo)) return set_error(RSA_E_ALIGN, "plane_stats: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)in_hi,
                     (const bf16x8*)in_lo, plane_stride, batch_stride, HW, C, eps, stats, fmt);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "plane_stats: launch failed") : RSA_OK;
}
extern "C" int rsa_plane_stats(const void* in_hi, const void* in_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W,
                               int32_t C, float eps, float* stats, void* stream) {
  return rsa_plane_stats_fmt(in_hi, in_lo, plane_stride, batch_stride, batch, H, W, C, eps, RSA_PF_BF16, stats, stream);
}

extern "C" int64_t rsa_channel_gate_workspace_bytes(int32_t batch, int32_t H, int32_t W, int32_t planes) {
  if (batch < 1 || H < 1 || W < 1 || planes < 1) return RSA_E_ARG;
  const int64_t chunks = ((int64_t)H * W + CG_PIX - 1) / CG_PIX;
  return (int64_t)batch * planes * chunks * 8 * 4;
}

extern "C" int rsa_channel_gate(const rsa_channel_gate_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "channel_gate: null params");
  if (p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "channel_gate: fmt must be an rsa_plane_fmt");
  if (p->batch < 1 || p->batch > 65535 || p->H < 1 || p->W < 1 || p->planes < 1 || p->planes > 64 || p->hidden < 1 || p->hidden > 128)
    return set_error(RSA_E_ARG, "channel_gate: bad geometry (planes <= 64, hidden <= 128)");
  if (!p->in_hi || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->workspace || !p->gate) return set_error(RSA_E_ARG, "channel_gate: null pointer");
  if (misaligned(p->in_hi) || misaligned(p->in_lo)) return set_error(RSA_E_ALIGN, "channel_gate: maps must be 16-byte aligned");
  const int64_t chunks = ((int64_t)p->H * p->W + CG_PIX - 1) / CG_PIX;
  if (chunks > 0x7fffffff) return set_error(RSA_E_UNSUPPORTED, "channel_gate: map too large");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)chunks, (unsigned)p->planes, (unsigned)p->batch), dim3(256), 0, s, *p, (int)chunks);
  hipLaunchKernelGGL(channel_gate_kernel, dim3((unsigned)p->batch), dim3(256), 0, s, *p, (int)chunks);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "channel_gate: launch failed") : RSA_OK;
}

extern "C" int rsa_aim_combine(const rsa_aim_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "aim_combine: null params");
  if (p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) return set_error(RSA_E_ARG, "aim_combine: fmt must be an rsa_plane_fmt");
  if (p->batch < 1 || p->batch > 65535 || p->H < 1 || p->W < 1 || p->planes < 1 || p->hidden < 1 || p->hidden > 16 || (p->mode != 0 && p->mode != 1))
    return set_error(RSA_E_ARG, "aim_combine: bad geometry (hidden <= 16, mode 0 or 1)");
  if (!p->att_hi || !p->conv_hi || !p->gate || !p->w1 || !p->b1 || !p->w2 || !p->out_hi) return set_error(RSA_E_ARG, "aim_combine: null pointer");
  if (misaligned(p->att_hi) || misaligned(p->att_lo) || misaligned(p->conv_hi) || misaligned(p->conv_lo) || misaligned(p->out_hi) || misaligned(p->out_lo))
    return set_error(RSA_E_ALIGN, "aim_combine: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)p->H * p->W;
  hipLaunchKernelGGL(aim_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)p->batch), dim3(256), 0, (hipStream_t)stream, *p);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "aim_combine: launch failed") : RSA_OK;
}

extern "C" int rsa_gated_add(const void* x_hi, const void* x_lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t H, int32_t W,
                             int32_t C, const float* gate, float scale, const float* base_f32, float* out_f32, void* stream) {
  if (!x_hi || !gate || !base_f32 || !out_f32 || batch < 1 || batch > 65535 || H < 1 || W < 1 || C < 1 || C > 8 * 65535)
    return set_error(RSA_E_ARG, "gated_add: bad argument");
  if (misaligned(x_hi) || misaligned(x_lo) || misaligned(base_f32) || misaligned(out_f32)) return set_error(RSA_E_ALIGN, "gated_add: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(gated_add_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)((C + 7) / 8), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8*)x_hi, (const bf16x8*)x_lo, plane_stride, batch_stride, HW, C, gate, scale, (const f32x4*)base_f32, (f32x4*)out_f32);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "gated_add: launch failed") : RSA_OK;
}
