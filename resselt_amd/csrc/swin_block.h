// swin_block.h — device helpers shared by the fused Swin block kernels (swin_block.hip: the two halves; swin_block_full.hip: a
// whole block): LayerNorm into an LDS plane image, the token-stationary Linear-layer multiply, D-fragment -> operand conversions.
#pragma once
#include "conv_common.h"

namespace rsa {


constexpr int SB_TOK = 64;  // tokens per workgroup
constexpr float LOG2E = 1.44269504088896340736f;

// Timing experiments only (tools/variant.sh NAME -DRSA_SB_ABL=mask swin_block): 1 no residual-stream loads in the LayerNorm, 2 no
// GELU / exp, 4 no MFMAs in the Linear layers, 8 no operand fetches there, 16 no residual loads / stores in the epilogues, 32 every weight fetch reads K chunk 0 (no L2 streaming).  Results
// are wrong in every one of them.
#ifndef RSA_SB_ABL
#define RSA_SB_ABL 0
#endif
// (64: the residual stream addressed token-major, [pixel][group], instead of [group][pixel])
#define MAPIDX(g, pix) ((RSA_SB_ABL & 64) ? (int64_t)(pix) * p4 + (g) : (int64_t)(g) * HW + (pix))

// LayerNorm of the workgroup's tokens into the LDS plane image, in two steps so that the loads of the NEXT tile can be in flight during
// the last multiply of the current one.  Lane (j, tok8) = token 8*row + tok8, planes j, j+8, j+16, j+24; wave w owns row w (rows
// w + nwaves, ... of a workgroup with fewer than 8 waves are loaded where they are used).
struct LnRow {
  f32x4 v[4][2];
};

__device__ __forceinline__ void ln_load(LnRow& r, const f32x4* x_img, int64_t HW, int p4, int64_t pix, int lane) {
  const int j = lane >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = (j + 8 * i) * 2 + h;
      r.v[i][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (pix >= 0 && g < p4 && !(RSA_SB_ABL & 1)) r.v[i][h] = x_img[MAPIDX(g, pix)];  // C % 4 == 0: a group is whole or absent
      if (RSA_SB_ABL & 1) r.v[i][h] = (f32x4){0.1f * (float)g, 0.3f, -0.2f * (float)lane, 1.f};
    }
}

// statistics by lane shuffles (the 8 lanes of a token sit 8 apart), result as split planes; planes [ceil(C/8), planes_pad) and
// tokens without a pixel (pix < 0) get zeros
template <int PROD, int FMT = 0>
__device__ __forceinline__ void ln_store(const LnRow& r, uint4* lds, int lo0, int planes_pad, int C, const float* gamma, const float* beta, float eps,
                                         int t, int64_t pix, int lane) {
  const int j = lane >> 3;
  const int p4 = (C + 3) >> 2;
  const float inv_c = 1.f / (float)C;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) sum += (r.v[i][h][0] + r.v[i][h][1]) + (r.v[i][h][2] + r.v[i][h][3]);
  sum += __shfl_xor(sum, 8);
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float mean = sum * inv_c;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = (j + 8 * i) * 2 + h;
      if (g < p4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = r.v[i][h][e] - mean;
          var += d * d;
        }
      }
    }
  var += __shfl_xor(var, 8);
  var += __shfl_xor(var, 16);
  var += __shfl_xor(var, 32);
  const float rstd = rsqrtf(var * inv_c + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = j + 8 * i;
    if (pl >= planes_pad) continue;
    uint32_t h[4], l[4];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int g = pl * 2 + hh;
      f32x4 y = {0.f, 0.f, 0.f, 0.f};
      if (pix >= 0 && g < p4) {
        const f32x4 ga = ((const f32x4*)gamma)[g], be = ((const f32x4*)beta)[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (r.v[i][hh][e] - mean) * rstd * ga[e] + be[e];
      }
      split2<FMT>(y[0], y[1], h[2 * hh], l[2 * hh]);
      split2<FMT>(y[2], y[3], h[2 * hh + 1], l[2 * hh + 1]);
    }
    lds[pl * SB_TOK + t] = make_uint4(h[0], h[1], h[2], h[3]);
    if (PROD == 3) lds[lo0 + pl * SB_TOK + t] = make_uint4(l[0], l[1], l[2], l[3]);
  }
}

// The weight fragments of K chunk 0 of a multiply, fetched by the caller long before the multiply starts (ahead of a barrier, of the
// LayerNorm arithmetic, of the softmax): the first L2 round trip of every Linear layer would otherwise be exposed in all waves at once.
template <int PROD, int CTW>
struct W0 {
  bf16x8 w[CTW][PROD == 3 ? 2 : 1];
};

template <int PROD, int CTW>
__device__ __forceinline__ void w0_load(W0<PROD, CTW>& f, const __amdgpu_buffer_rsrc_t rw, const uint32_t (&woff)[CTW]) {
  constexpr int NHL = PROD == 3 ? 2 : 1;
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int hl = 0; hl < NHL; ++hl) f.w[c][hl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)hl * 1024u, 0));
}

// acc[c][pt] += W[cout tile c][K] . X[K][token tile pt]  over nk chunks of 32 channels.  X: LDS plane image (hi at 0, lo at lo0);
// W: packed blob (layout 0, ksize 1: [chunk][cout tile][hi|lo][lane][8]) behind the buffer resource rw, woff[c] = byte offset of
// this lane's fragment of cout tile c inside a chunk, or 0xFFFFFFFF (a tile beyond the layer: the range check of the buffer load
// looks at the vector offset alone and returns zeros).  SWAP: tokens on the MFMA rows (D[token][channel]) instead of the columns.
// XDB false: the token fragments are single-buffered (32 registers less; their LDS latency is then exposed once per chunk).
// WAH = K chunks the weight fragments are requested ahead of their multiply.  1: two named buffers (the three-product kernels: a chunk is
// 24 MFMAs = 384 cycles, about an L2 round trip).  2 (round 3, the one-product kernels: a chunk is 8 MFMAs = 128 cycles, a quarter of one): three
// buffers, six chunks per loop iteration so that buffer indices stay compile-time constants (weights rotate mod 3, token fragments mod 2).
template <int PROD, int CTW, int NPT, bool SWAP, bool XDB = true, int FMT = 0, int WAH = 1>
__device__ __forceinline__ void gemm_tile(f32x4 (&acc)[CTW][NPT], const uint4* lds, int lo0, int nk, const __amdgpu_buffer_rsrc_t rw,
                                          const uint32_t (&woff)[CTW], uint32_t wstep, const W0<PROD, CTW>& w0, int li, int lg) {
  constexpr int NHL = PROD == 3 ? 2 : 1;
  const int bu = lg * SB_TOK + li;
  bf16x8 w[WAH + 1][CTW][NHL], bh[XDB ? 2 : 1][NPT], bl[XDB ? 2 : 1][NPT];  // [buffer]: chunk kc lives in buffer kc % (WAH + 1) / kc & 1
  auto fetch_w = [&](int kc, int b) {
    if (RSA_SB_ABL & 8) return;
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int hl = 0; hl < NHL; ++hl)
        w[b][c][hl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], ((RSA_SB_ABL & 32) ? 0u : (uint32_t)kc * wstep) + (uint32_t)hl * 1024u, 0));
  };
  auto fetch_x = [&](int kc, int b_) {
    const int b = XDB ? b_ : 0;
    if (RSA_SB_ABL & 8) return;
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      const int u = kc * 4 * SB_TOK + bu + 16 * pt;
      bh[b][pt] = __builtin_bit_cast(bf16x8, lds[u]);
      if (PROD == 3) bl[b][pt] = __builtin_bit_cast(bf16x8, lds[lo0 + u]);
    }
  };
  auto multiply = [&](int b) {
    const int xb = XDB ? b : 0;
    if (RSA_SB_ABL & 4) {  // keep the fetched operands alive
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) asm volatile("" ::"v"(bh[xb][pt]), "v"(bl[xb][pt]));
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) asm volatile("" ::"v"(w[b][c][hl]));
      return;
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        // products in increasing magnitude: w_lo*x_hi, w_hi*x_lo, w_hi*x_hi
        if (PROD == 3) {
          acc[c][pt] = SWAP ? mfma16<FMT>(bh[xb][pt], w[b][c][NHL - 1], acc[c][pt]) : mfma16<FMT>(w[b][c][NHL - 1], bh[xb][pt], acc[c][pt]);
          acc[c][pt] = SWAP ? mfma16<FMT>(bl[xb][pt], w[b][c][0], acc[c][pt]) : mfma16<FMT>(w[b][c][0], bl[xb][pt], acc[c][pt]);
        }
        acc[c][pt] = SWAP ? mfma16<FMT>(bh[xb][pt], w[b][c][0], acc[c][pt]) : mfma16<FMT>(w[b][c][0], bh[xb][pt], acc[c][pt]);
      }
  };
  auto multiply3 = [&](int wb, int xb_) {  // WAH 2: weights in buffer wb (mod 3), token fragments in buffer xb_ (mod 2)
    const int xb = XDB ? xb_ : 0;
    if (RSA_SB_ABL & 4) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) asm volatile("" ::"v"(bh[xb][pt]));
#pragma unroll
      for (int c = 0; c < CTW; ++c) asm volatile("" ::"v"(w[wb][c][0]));
      return;
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int c = 0; c < CTW; ++c) acc[c][pt] = SWAP ? mfma16<FMT>(bh[xb][pt], w[wb][c][0], acc[c][pt]) : mfma16<FMT>(w[wb][c][0], bh[xb][pt], acc[c][pt]);
  };
  static_assert(WAH == 1 || PROD == 1, "the deeper weight pipeline exists for the one-product kernels");
  if (RSA_SB_ABL & 8) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) bh[XDB ? b : 0][pt] = bl[XDB ? b : 0][pt] = __builtin_bit_cast(bf16x8, make_uint4(0x3f803e12u + li, 0xbe993f01u, 0x3dcc3e4cu + lg, 0x3f003f11u));
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) w[b][c][hl] = bh[b][0];
    }
  }
  // Both operands one chunk ahead of the multiply, two chunks per iteration so that the buffers are named, not copied.  No fetch
  // sits under a branch (after a conditional fetch the compiler no longer knows how many loads are outstanding and waits for ALL of
  // them, the chunk just requested included, before the first MFMA), and scheduling barriers keep the machine scheduler from
  // sinking the fetches to their uses.  An odd chunk count leaves its last chunk in buffer 0 for the tail multiply; an even one
  // fetches its last chunk a second time instead of nothing.
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int hl = 0; hl < NHL; ++hl) w[0][c][hl] = w0.w[c][hl];
  fetch_x(0, 0);
  if constexpr (WAH == 2) {
    // Every fetch is unconditional (an index past the end re-requests the last chunk: at most two wasted requests per multiply), so the
    // compiler's count of outstanding loads stays exact; the loop is left by a uniform branch behind the last multiply.
    fetch_w(nk > 1 ? 1 : 0, 1);
#pragma unroll 1
    for (int k0 = 0;; k0 += 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int k = k0 + i;  // the chunk multiplied in this step (k < nk here)
        fetch_w(k + 2 < nk ? k + 2 : nk - 1, (i + 2) % 3);
        if (XDB) fetch_x(k + 1 < nk ? k + 1 : nk - 1, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        multiply3(i % 3, i & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!XDB) fetch_x(k + 1 < nk ? k + 1 : nk - 1, 0);
        if (k + 1 >= nk) return;
      }
    }
  }
  int kc = 0;
#pragma unroll 1
  for (; kc + 1 < nk; kc += 2) {
    fetch_w(kc + 1, 1);
    if (XDB) fetch_x(kc + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(0);
    __builtin_amdgcn_sched_barrier(0);
    if (!XDB) fetch_x(kc + 1, 0);
    const int k2 = kc + 2 < nk ? kc + 2 : kc + 1;
    fetch_w(k2, 0);
    if (XDB) fetch_x(k2, 0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(1);
    __builtin_amdgcn_sched_barrier(0);
    if (!XDB) fetch_x(k2, 0);
  }
  if (kc < nk) multiply(0);
}

// eight f32 values -> the hi fragment (plane format FMT) and (PROD 3) the residual fragment
template <int FMT = 0>
__device__ __forceinline__ void frag_of(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
  split2<FMT>(a[0], a[1], h[0], l[0]);
  split2<FMT>(a[2], a[3], h[1], l[1]);
  split2<FMT>(b[0], b[1], h[2], l[2]);
  split2<FMT>(b[2], b[3], h[3], l[3]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

template <int PROD, int FMT = 0>
__device__ __forceinline__ f32x4 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 c) {
  if (PROD == 3) {
    c = mfma16<FMT>(al, bh, c);
    c = mfma16<FMT>(ah, bl, c);
  }
  return mfma16<FMT>(ah, bh, c);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const void* w, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (uint32_t)bytes, 0x00020000);
}

}  // namespace rsa
