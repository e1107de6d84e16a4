// swin_block_full.hip — rsa_swin_block: a whole SwinTransformerBlock.forward (reference archs/swinir/arch.py:295-335) in ONE launch:
//
//   x1  = x  + proj(window_attention(qkv(norm1(x))))      (WindowAttention.forward :133-173, roll / partition / mask as index math)
//   out = x1 + fc2(GELU(fc1(norm2(x1))))                  (Mlp.forward :34-40)
//
// One workgroup (8 waves) = one (shifted) window.  The two halves are those of swin_block.hip; what the fusion adds is that x1 never
// leaves the chip: the wave that owns cout tiles 2w, 2w+1 of proj keeps its 32 values of x1 in registers, norm2's statistics are
// closed over the 8 waves through 4 KB of LDS, norm2's output goes straight into the LDS plane image that fc1 reads, and the same
// registers are the shortcut of fc2's epilogue (fc2 has proj's cout-tile ownership).  Per block the residual stream is read once and
// written once: 1.9 KB per token against 5.8 KB for the two half launches (x1 written, read by norm2, read again as the shortcut).
#include "swin_block.h"

// K chunks of weight prefetch in the one-product kernel (swin_block.h::gemm_tile).  2 was measured on SwinIR-L 1024^2: 142.5 -> 162.4 ms -- the third
// weight buffer does not fit the 128 registers of the two-windows-per-CU form (83 spilled instead of 43), so the default stays 1.
#ifndef RSA_SB_WAH1
#define RSA_SB_WAH1 1
#endif

namespace rsa {

// PROD 1 (one product, bf16 or fp16 per FMT): no lo images, so the array is 64 KB and TWO windows share a CU -- one window's LayerNorm /
// softmax / GELU / barrier phases then sit beside the other's multiplies (launch bounds: 4 waves per SIMD = 128 registers).
template <int PROD, int FMT>
__global__ __launch_bounds__(512, PROD == 1 ? 4 : 2) void swin_block_kernel(const rsa_swin_block_params p) {
  constexpr int HPL = 64;  // planes of the LDS image (512 hidden channels); the token / attention-output images use planes 0..31
  constexpr int LO0 = HPL * SB_TOK;
  // While the token / attention-output images are in use (hi planes 0..31, lo planes 64..95) the other half of the array is free:
  // the raw f32 rows norm1 loaded are parked there ([channel group][token], groups 0..31 in hi planes 32..63, groups 32..59 in lo
  // planes 96..123) and come back as proj's shortcut -- the residual stream is fetched from memory exactly once per block (15 us
  // after norm1 read them the lines have left the L2: measured 1 GB of re-fetch per launch).  Widths above 240 channels have no
  // room for that and read the shortcut from memory.  norm2's partial sums: lo planes 124..127.
  // One product: only groups 0..31 are parked (hi planes 32..63), the rest of the shortcut is read from memory again; norm2's partial sums
  // reuse the first planes of the parked rows (they are consumed before proj starts).
  constexpr int STASH0 = 32 * SB_TOK, STASH1 = 96 * SB_TOK;
  constexpr int RED0 = (PROD == 3 ? 124 : 32) * SB_TOK;
  constexpr int NHL = PROD == 3 ? 2 : 1;
  constexpr int WAH_ = PROD == 1 ? RSA_SB_WAH1 : 1;  // K chunks of weight prefetch (swin_block.h::gemm_tile)
  __shared__ uint4 s_h[(PROD == 3 ? 2 : 1) * HPL * SB_TOK];  // 128 KB / 64 KB

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int heads = p.heads;
  const int li = lane & 15, lg = lane >> 4;
  const int w = p.window, ntok = w * w;
  const int nwx = p.W / w, nwy = p.H / w;
  const int win = (int)blockIdx.x;
  const int wx = win % nwx, wy = (win / nwx) % nwy, n = win / (nwx * nwy);
  const int64_t HW = (int64_t)p.H * p.W;
  const int p4 = (p.C + 3) >> 2;
  const int planes = (p.C + 7) >> 3;
  const int nk = (planes + 3) >> 2;
  const int hplanes = (p.hidden + 7) >> 3;
  const int nk2 = (hplanes + 3) >> 2;
  const int ct_qkv = 3 * heads * 2;
  const int ct_c = (p.C + 15) >> 4, ct_h = (p.hidden + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rq = weight_rsrc(p.wqkv, (int64_t)nk * ct_qkv * NHL * 1024);
  const __amdgpu_buffer_rsrc_t rp = weight_rsrc(p.wproj, (int64_t)heads * ct_c * NHL * 1024);
  const __amdgpu_buffer_rsrc_t r1 = weight_rsrc(p.w1, (int64_t)nk * ct_h * NHL * 1024);
  const __amdgpu_buffer_rsrc_t r2 = weight_rsrc(p.w2, (int64_t)nk2 * ct_c * NHL * 1024);
  const uint32_t qstep = (uint32_t)ct_qkv * NHL * 1024u;
  const f32x4* bqkv4 = (const f32x4*)p.bqkv;
  const uint32_t wdiv = 65536u / (uint32_t)w + 1u;  // t / w == (t * wdiv) >> 16 for t < 64, w <= 8
  const bool has_head = wave < heads;
  const int head = has_head ? wave : 0;

  auto token_pix = [&](int t) -> int64_t {  // see swin_attn_block_kernel
    if (t >= ntok) return -1;
    const int ty = (int)(((uint32_t)t * wdiv) >> 16), tx = t - ty * w;
    int py = wy * w + ty + p.shift, px = wx * w + tx + p.shift;
    if (py >= p.H) py -= p.H;
    if (px >= p.W) px -= p.W;
    return (int64_t)py * p.W + px;
  };
  const f32x4* x_img = (const f32x4*)p.x + (int64_t)n * p4 * HW;

  // ---- norm1 -> LDS image ----
  uint32_t woff_qk[4], woff_2[2];  // woff_2: cout tiles 2*wave, 2*wave + 1 of proj and of fc2 (same tile count)
#pragma unroll
  for (int c = 0; c < 4; ++c) woff_qk[c] = has_head ? (uint32_t)(((((c >> 1) * heads + head) * 2 + (c & 1)) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
#pragma unroll
  for (int c = 0; c < 2; ++c) woff_2[c] = (2 * wave + c < ct_c) ? (uint32_t)(((2 * wave + c) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
  W0<PROD, 4> w0qk;
  w0_load<PROD, 4>(w0qk, rq, woff_qk);
  const int stash_groups = PROD == 3 ? (p4 <= 60 ? p4 : 0) : min(p4, 32);  // channel groups of the shortcut parked in LDS
  {
    const int t = wave * 8 + (lane & 7);
    const int64_t pix = token_pix(t);
    LnRow row;
    ln_load(row, x_img, HW, p4, pix, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = ((lane >> 3) + 8 * i) * 2 + h;
        if (g < stash_groups) s_h[(g < 32 ? STASH0 + g * SB_TOK : STASH1 + (g - 32) * SB_TOK) + t] = __builtin_bit_cast(uint4, row.v[i][h]);
      }
    ln_store<PROD, FMT>(row, s_h, LO0, 4 * nk, p.C, p.gamma1, p.beta1, p.eps, t, pix, lane);
  }
  __syncthreads();

  // shift mask: img_mask region ids (arch.py:268-293) of this lane's query column and key rows on the SHIFTED grid, 4 bits each
  const bool masked = p.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
  uint32_t qreg = 0, kreg[4] = {0, 0, 0, 0};
  if (masked) {
    auto region = [&](int t) -> uint32_t {
      const int tt = t < ntok ? t : 0;
      const int ty = (int)(((uint32_t)tt * wdiv) >> 16), tx = tt - ty * w;
      const int gy = wy * w + ty, gx = wx * w + tx;
      const int ry = gy < p.H - w ? 0 : (gy < p.H - p.shift ? 1 : 2);
      const int rx = gx < p.W - w ? 0 : (gx < p.W - p.shift ? 1 : 2);
      return (uint32_t)(ry * 3 + rx);
    };
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      qreg |= region(16 * t4 + li) << (4 * t4);
#pragma unroll
      for (int r = 0; r < 4; ++r) kreg[t4] |= region(16 * t4 + 4 * lg + r) << (4 * r);
    }
  }

  // ---- attention half: wave = head (waves beyond the head count only take part in the barriers) ----
  W0<PROD, 2> w0p;
  uint4 ouh[4], oul[4];
  if (has_head) {
    uint32_t woff_v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) woff_v[c] = (uint32_t)((((2 * heads + head) * 2 + c) * NHL * 64 + lane) * 16);
    bf16x8 qh[4], ql[4], kh[4], kl[4];
    W0<PROD, 2> w0v;
    if constexpr (PROD == 3) {
      f32x4 a[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      gemm_tile<PROD, 4, 4, false, true, FMT, WAH_>(a, s_h, LO0, nk, rq, woff_qk, qstep, w0qk, li, lg);
      w0_load<PROD, 2>(w0v, rq, woff_v);
      f32x4 b[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = bqkv4[(((c >> 1) * heads + head) * 2 + (c & 1)) * 4 + lg];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        frag_of<FMT>(a[0][tt] + b[0], a[1][tt] + b[1], qh[tt], ql[tt]);
        frag_of<FMT>(a[2][tt] + b[2], a[3][tt] + b[3], kh[tt], kl[tt]);
      }
    } else {
      // one product, 128 registers (two windows per CU): q and k in two passes of two cout tiles, each with half the accumulators
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 a[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint32_t wo[2] = {woff_qk[2 * half], woff_qk[2 * half + 1]};
        W0<PROD, 2> w0h;
#pragma unroll
        for (int c = 0; c < 2; ++c) w0h.w[c][0] = w0qk.w[2 * half + c][0];
        gemm_tile<PROD, 2, 4, false, true, FMT, WAH_>(a, s_h, LO0, nk, rq, wo, qstep, w0h, li, lg);
        if (half == 1) w0_load<PROD, 2>(w0v, rq, woff_v);
        f32x4 b[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) b[c] = bqkv4[((half * heads + head) * 2 + c) * 4 + lg];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          if (half == 0)
            frag_of<FMT>(a[0][tt] + b[0], a[1][tt] + b[1], qh[tt], ql[tt]);
          else
            frag_of<FMT>(a[0][tt] + b[0], a[1][tt] + b[1], kh[tt], kl[tt]);
        }
      }
    }
    bf16x8 vh[2][2], vl[2][2];
    {
      f32x4 a[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (one product: q and k fragments are live beside this multiply; its token fragments are single-buffered to stay within 128 registers)
      gemm_tile<PROD, 2, 4, true, PROD == 3, FMT, WAH_>(a, s_h, LO0, nk, rq, woff_v, qstep, w0v, li, lg);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float bv = p.bqkv[((2 * heads + head) * 2 + dt) * 16 + li];
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) frag_of<FMT>(a[dt][2 * kp] + bv, a[dt][2 * kp + 1] + bv, vh[dt][kp], vl[dt][kp]);
      }
    }
    f32x4 o[2][2];  // [channel tile][query tile of the current pair]: packed into plane units every second query tile
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 s[4];
      const uint32_t rq_ = (qreg >> (4 * qt)) & 15u;
      float m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = mfma3<PROD, FMT>(kh[kt], kl[kt], qh[qt], ql[qt], (f32x4){0.f, 0.f, 0.f, 0.f});
        const f32x4 bf = ((const f32x4*)p.bias_frag16)[(((int64_t)head * 4 + kt) * 4 + qt) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(s[kt][r], LOG2E, bf[r]);
          if (masked && ((kreg[kt] >> (4 * r)) & 15u) != rq_) v += -100.f * LOG2E;
          s[kt][r] = v;
          m = fmaxf(m, v);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);
          s[kt][r] = e;
          l += e;
        }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv_l = 1.f / l;
      bf16x8 ph[2], pl[2];
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) frag_of<FMT>(s[2 * kp], s[2 * kp + 1], ph[kp], pl[kp]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) acc = mfma3<PROD, FMT>(vh[dt][kp], vl[dt][kp], ph[kp], pl[kp], acc);
        o[dt][qt & 1] = acc * inv_l;
      }
      if (qt & 1) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) pair_units<FMT>(o[dt][0], o[dt][1], ouh[dt * 2 + (qt >> 1)], oul[dt * 2 + (qt >> 1)]);
      }
    }
  }
  w0_load<PROD, 2>(w0p, rp, woff_2);
  __syncthreads();  // every wave has read the norm1 image for the last time
  if (has_head) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u = (head * 4 + dt * 2 + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
        s_h[u] = ouh[dt * 2 + k];
        if (PROD == 3) s_h[LO0 + u] = oul[dt * 2 + k];
      }
  }
  // the shortcut of proj's epilogue: from the parked rows, or (wide models) requested from memory ahead of the barrier and the
  // multiply; from the epilogue on the same registers hold x1: [cout tile 2*wave + c][token tile]
  f32x4 x1[2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int g = (2 * wave + c) * 4 + lg;
      const int t = 16 * pt + li;
      x1[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (g < stash_groups) {
        if (t < ntok) x1[c][pt] = __builtin_bit_cast(f32x4, s_h[(g < 32 ? STASH0 + g * SB_TOK : STASH1 + (g - 32) * SB_TOK) + t]);
      } else {
        const int64_t px = token_pix(t);
        if (px >= 0 && g < p4) x1[c][pt] = x_img[(int64_t)g * HW + px];
      }
    }
  __syncthreads();

  // ---- proj + bias + shortcut -> x1 (registers) ----
  // hidden cout tiles of this wave: two in the lower half of the hidden image (tiles 2w, 2w+1: planes below 32, where the norm2 image lives
  // while fc1 reads it) and two in the upper half (16 + 2w, 16 + 2w + 1: free planes, stored as soon as they are computed)
  auto hid_tile = [&](int c) -> int { return (c < 2 ? 0 : 16) + 2 * wave + (c & 1); };
  uint32_t woff_1[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) woff_1[c] = (hid_tile(c) < ct_h) ? (uint32_t)((hid_tile(c) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
  W0<PROD, 4> w01;
  {
    f32x4 a[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tile<PROD, 2, 4, false, true, FMT, WAH_>(a, s_h, LO0, heads, rp, woff_2, (uint32_t)ct_c * NHL * 1024u, w0p, li, lg);
    w0_load<PROD, 4>(w01, r1, woff_1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int g = (2 * wave + c) * 4 + lg;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (g < p4) b = ((const f32x4*)p.bproj)[g];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const bool ok = g < p4 && 16 * pt + li < ntok;
        x1[c][pt] = ok ? a[c][pt] + b + x1[c][pt] : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  // ---- norm2 over x1: this wave holds 32 of a token's channels; mean and centred variance are closed over the 8 waves through LDS
  //      ([token][wave] floats), the normalised values go into the plane image as 16-byte units (lane-pair exchange) ----
  float* red = (float*)&s_h[RED0];  // [2][64 tokens][8 waves]
  const float inv_c = 1.f / (float)p.C;
  float mean[4], rstd[4];
  {
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) s += (x1[c][pt][0] + x1[c][pt][1]) + (x1[c][pt][2] + x1[c][pt][3]);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (lg == 0) red[(16 * pt + li) * 8 + wave] = s;
    }
    __syncthreads();  // (also: every wave is past its proj multiply, the attention image may be overwritten from here on)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const f32x4 a = *(const f32x4*)&red[(16 * pt + li) * 8], b = *(const f32x4*)&red[(16 * pt + li) * 8 + 4];
      mean[pt] = ((a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3])) * inv_c;
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if ((2 * wave + c) * 4 + lg >= p4) continue;  // channels beyond C hold zeros, not samples
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = x1[c][pt][r] - mean[pt];
          v += d * d;
        }
      }
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lg == 0) red[512 + (16 * pt + li) * 8 + wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const f32x4 a = *(const f32x4*)&red[512 + (16 * pt + li) * 8], b = *(const f32x4*)&red[512 + (16 * pt + li) * 8 + 4];
      rstd[pt] = rsqrtf(((a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3])) * inv_c + p.eps);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = 2 * wave + c;
      if (ct >= 2 * nk) continue;  // planes of the K padding are written (as zeros) too
      const int g = ct * 4 + lg;
      f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = {0.f, 0.f, 0.f, 0.f};
      if (g < p4) {
        ga = ((const f32x4*)p.gamma2)[g];
        be = ((const f32x4*)p.beta2)[g];
      }
      f32x4 y[4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const bool ok = g < p4 && 16 * pt + li < ntok;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[pt][r] = ok ? (x1[c][pt][r] - mean[pt]) * rstd[pt] * ga[r] + be[r] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        uint4 uh, ul;
        pair_units<FMT>(y[2 * k], y[2 * k + 1], uh, ul);
        const int u = (2 * ct + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
        s_h[u] = uh;
        if (PROD == 3) s_h[LO0 + u] = ul;
      }
    }
  }
  __syncthreads();

  // ---- fc1 + GELU in two passes of two cout tiles (x1 stays in registers beside them: four tiles at once do not fit).  The lower-half
  //      tiles wait as plane units until every wave is done with the norm2 image; the upper-half tiles go straight into their planes ----
  uint4 hu[2][2], hl[2][2];  // [lower-half cout tile][token tile pair]
  W0<PROD, 2> w02;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    f32x4 a1[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a1[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint32_t wo[2] = {woff_1[2 * ps], woff_1[2 * ps + 1]};
    W0<PROD, 2> w0h;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int hl_ = 0; hl_ < NHL; ++hl_) w0h.w[c][hl_] = w01.w[2 * ps + c][hl_];
    gemm_tile<PROD, 2, 4, false, PROD == 3, FMT, WAH_>(a1, s_h, LO0, nk, r1, wo, (uint32_t)ct_h * NHL * 1024u, w0h, li, lg);
    if (ps == 1) w0_load<PROD, 2>(w02, r2, woff_2);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = hid_tile(2 * ps + c);
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (ct < ct_h) b = ((const f32x4*)p.b1)[ct * 4 + lg];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int r = 0; r < 4; ++r) a1[c][pt][r] = gelu_fast(a1[c][pt][r] + b[r]);  // tiles beyond the layer: GELU(0) = 0
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (ps == 0) {
          pair_units<FMT>(a1[c][2 * k], a1[c][2 * k + 1], hu[c][k], hl[c][k]);
        } else {
          uint4 uh, ul;
          pair_units<FMT>(a1[c][2 * k], a1[c][2 * k + 1], uh, ul);
          if (ct < 2 * nk2) {  // planes 32 and up: nobody reads them before the barrier below (the K padding gets its zeros too)
            const int u = (2 * ct + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
            s_h[u] = uh;
            if (PROD == 3) s_h[LO0 + u] = ul;
          }
        }
      }
    }
  }
  __syncthreads();  // every wave has read the norm2 image (and the partial sums) for the last time
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int ct = 2 * wave + c;
    if (ct >= 2 * nk2) continue;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int u = (2 * ct + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
      s_h[u] = hu[c][k];
      if (PROD == 3) s_h[LO0 + u] = hl[c][k];
    }
  }
  __syncthreads();

  // ---- fc2 + bias + x1 -> residual stream (and the optional split-plane copy) ----
  {
    f32x4 a[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tile<PROD, 2, 4, false, true, FMT, WAH_>(a, s_h, LO0, nk2, r2, woff_2, (uint32_t)ct_c * NHL * 1024u, w02, li, lg);
    f32x4* o_img = (f32x4*)p.out + (int64_t)n * p4 * HW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ct = 2 * wave + c;
      const int g = ct * 4 + lg;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (g < p4) b = ((const f32x4*)p.b2)[g];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int64_t pix = token_pix(16 * pt + li);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix >= 0 && g < p4) {
          v = a[c][pt] + b + x1[c][pt];
          o_img[(int64_t)g * HW + pix] = v;
        }
        a[c][pt] = v;
      }
      if (p.out_hi != nullptr) {  // wave-uniform: every lane takes part in the exchange
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          uint4 uh, ul;
          pair_units<FMT>(a[c][2 * k], a[c][2 * k + 1], uh, ul);
          const int pl = 2 * ct + (lg >> 1);
          const int64_t pix = token_pix(16 * (2 * k + (lg & 1)) + li);
          if (pix >= 0 && pl < planes) {
            const int64_t u = (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride + pix;
            ((uint4*)p.out_hi)[u] = uh;
            if (p.out_lo != nullptr) ((uint4*)p.out_lo)[u] = ul;
          }
        }
      }
    }
  }
}

static bool aligned16f(const void* a) { return ((uintptr_t)a & 15) == 0; }

}  // namespace rsa

extern "C" int rsa_swin_block(const rsa_swin_block_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "swin_block: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->C < 4 || p->heads < 1 || p->hidden < 1) return set_error(RSA_E_ARG, "swin_block: bad geometry");
  if (p->window < 1 || p->window > 8) return set_error(RSA_E_UNSUPPORTED, "swin_block: window must be 1..8 (<= 64 tokens)");
  if (p->H % p->window || p->W % p->window) return set_error(RSA_E_ARG, "swin_block: H and W must be multiples of the window");
  if (p->shift < 0 || p->shift >= p->window) return set_error(RSA_E_ARG, "swin_block: shift must be in [0, window)");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "swin_block: products must be 1 or 3");
  if (p->C > 256 || (p->C & 3) || p->hidden > 512) return set_error(RSA_E_UNSUPPORTED, "swin_block: C must be a multiple of 4, at most 256; hidden at most 512");
  if (p->heads > 8 || p->C % p->heads || p->C / p->heads > 32) return set_error(RSA_E_UNSUPPORTED, "swin_block: at most 8 heads of at most 32 channels");
  const void* ptrs[] = {p->x, p->gamma1, p->beta1, p->wqkv, p->bqkv, p->bias_frag16, p->wproj, p->bproj, p->gamma2, p->beta2, p->w1, p->b1, p->w2, p->b2, p->out};
  for (const void* q : ptrs) {
    if (q == nullptr) return set_error(RSA_E_ARG, "swin_block: null pointer");
    if (!aligned16f(q)) return set_error(RSA_E_ALIGN, "swin_block: pointers must be 16-byte aligned");
  }
  if (!aligned16f(p->out_hi) || !aligned16f(p->out_lo)) return set_error(RSA_E_ALIGN, "swin_block: pointers must be 16-byte aligned");
  const int64_t windows = (int64_t)p->batch * (p->H / p->window) * (p->W / p->window);
  if (windows > 0x3fffffff) return set_error(RSA_E_UNSUPPORTED, "swin_block: too many windows");
  if ((p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) || p->reserved0 != 0) return set_error(RSA_E_ARG, "swin_block: fmt must be an rsa_plane_fmt");
  if (p->fmt == RSA_PF_F16 && p->products != 1) return set_error(RSA_E_UNSUPPORTED, "swin_block: fp16 operands are compiled for products == 1");
  if (p->products == 3)
    hipLaunchKernelGGL((swin_block_kernel<3, RSA_PF_BF16>), dim3((unsigned)windows), dim3(512), 0, (hipStream_t)stream, *p);
  else if (p->fmt == RSA_PF_F16)
    hipLaunchKernelGGL((swin_block_kernel<1, RSA_PF_F16>), dim3((unsigned)windows), dim3(512), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL((swin_block_kernel<1, RSA_PF_BF16>), dim3((unsigned)windows), dim3(512), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "swin_block: launch failed") : RSA_OK;
}
