// swin_block.hip — the two halves of a Swin transformer block, one launch each:
//
//   rsa_swin_attn_block   out = x + proj(window_attention(qkv(norm1(x))))   (reference archs/swinir/arch.py:295-330, :133-173)
//   rsa_swin_mlp_block    out = x + fc2(GELU(fc1(norm2(x))))                (reference archs/swinir/arch.py:331-335, :34-40)
//
// The layer-by-layer path (rsa_layernorm, gemm_k1, rsa_window_attention) moves 21 KB per token and block through HBM in the split
// layouts; here a workgroup keeps its 64 tokens on chip from the f32 residual stream in to the f32 residual stream out (1.9 KB per
// token and half block).  The workgroup is token-stationary, the weights stream from L2 (every workgroup of a launch reads the same
// 0.4–0.9 MB, which the 4 MiB L2 of each XCD holds), each wave owning its own output channels so that no weight fragment is fetched
// twice by a workgroup:
//
//   LayerNorm        lane (j, tok8) = token 8*row + tok8, planes j, j+8, ...: statistics by lane shuffles, result as split planes
//                    [plane][64 tokens][8 ch] (hi image, lo image) in LDS -- the B-fragment image of v_mfma_f32_16x16x32_bf16
//   Linear layers    gemm_tile: a wave multiplies CTW cout tiles x 4 token tiles over the whole K; B fragments by ds_read_b128
//                    (conflict-free: plane stride 1 KiB), A fragments by buffer_load from the packed blob one K chunk ahead
//   attention        wave = head.  q and k come out of the qkv multiply as D fragments [channel][token]; v is multiplied with the
//                    operands swapped, D [token][channel].  A D fragment pair is a valid MFMA operand when both sides of the
//                    contraction use the same slot order (slot j of lane group lg <-> index 16*(j>>2) + 4*lg + (j&3)), so
//                    S^T = K Q^T, the softmax and O^T = V^T P^T run entirely in the wave's registers: no LDS, no transposes.
//   hand-over        attention output / hidden map back to LDS as split planes (lane-pair exchange, 16-byte units) over the
//                    LayerNorm image, workgroup barrier, next multiply.
#include "swin_block.h"

namespace rsa {

// ------------------------------------------------------------------------------------------------ attention half
// One workgroup = one (shifted) window.  ceil(heads / 2) waves, each running two heads one after the other: a workgroup of four waves
// (one per SIMD) with 64 KB of LDS, so that TWO windows are resident on a CU and the phases of one that issue no MFMA (LayerNorm loads
// and arithmetic, softmax, the epilogue's loads and stores, barriers -- 44 % of a window's time when it had the CU to itself) run
// beside the multiplies of the other.
template <int PROD>
__global__ __launch_bounds__(256, 2) void swin_attn_block_kernel(const rsa_swin_attn_block_params p) {
  constexpr int XPL = 32;  // planes of the LDS image (256 channels)
  constexpr int LO0 = XPL * SB_TOK;
  constexpr int NHL = PROD == 3 ? 2 : 1;
  constexpr int HPW = 2;  // heads per wave
  __shared__ uint4 s_x[2 * XPL * SB_TOK];  // 64 KB: LayerNorm image, later the attention output image

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int heads = p.heads;
  const int nw = (heads + HPW - 1) / HPW;  // waves of the workgroup
  const int li = lane & 15, lg = lane >> 4;
  const int w = p.window, ntok = w * w;
  const int nwx = p.W / w, nwy = p.H / w;
  const int win = (int)blockIdx.x;
  const int wx = win % nwx, wy = (win / nwx) % nwy, n = win / (nwx * nwy);
  const int64_t HW = (int64_t)p.H * p.W;
  const int p4 = (p.C + 3) >> 2;
  const int planes = (p.C + 7) >> 3;
  const int nk = (planes + 3) >> 2;
  const int ct_qkv = 3 * heads * 2;
  const int ct_c = (p.C + 15) >> 4;
  const __amdgpu_buffer_rsrc_t rq = weight_rsrc(p.wqkv, (int64_t)nk * ct_qkv * NHL * 1024);
  const __amdgpu_buffer_rsrc_t rp = weight_rsrc(p.wproj, (int64_t)heads * ct_c * NHL * 1024);
  const uint32_t qstep = (uint32_t)ct_qkv * NHL * 1024u;
  const f32x4* bqkv4 = (const f32x4*)p.bqkv;
  const uint32_t wdiv = 65536u / (uint32_t)w + 1u;  // t / w == (t * wdiv) >> 16 for t < 64, w <= 8

  // pixel of window token t after the cyclic shift (torch.roll(-s), partition; the result is rolled back, so a token's output
  // lands on the pixel it was read from)
  auto token_pix = [&](int t) -> int64_t {
    if (t >= ntok) return -1;
    const int ty = (int)(((uint32_t)t * wdiv) >> 16), tx = t - ty * w;
    int py = wy * w + ty + p.shift, px = wx * w + tx + p.shift;
    if (py >= p.H) py -= p.H;
    if (px >= p.W) px -= p.W;
    return (int64_t)py * p.W + px;
  };
  const f32x4* x_img = (const f32x4*)p.x + (int64_t)n * p4 * HW;

  // ---- LayerNorm -> LDS image (the first K chunk of the first head's q / k weights is on its way meanwhile) ----
  uint32_t woff_qk[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) woff_qk[c] = (uint32_t)(((((c >> 1) * heads + wave) * 2 + (c & 1)) * NHL * 64 + lane) * 16);
  W0<PROD, 4> w0qk;
  w0_load<PROD, 4>(w0qk, rq, woff_qk);
  for (int r = wave; r < 8; r += nw) {
    const int t = r * 8 + (lane & 7);
    const int64_t pix = token_pix(t);
    LnRow row;
    ln_load(row, x_img, HW, p4, pix, lane);
    ln_store<PROD>(row, s_x, LO0, 4 * nk, p.C, p.gamma, p.beta, p.eps, t, pix, lane);
  }
  __syncthreads();

  // shift mask: img_mask region id (arch.py:268-293) of this lane's query column and key rows on the SHIFTED grid, 4 bits each
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

  uint4 ouh[HPW][4], oul[HPW][4];  // attention output of this wave's heads as plane units, until every wave is done with the image
#pragma unroll
  for (int hi = 0; hi < HPW; ++hi) {
    const int head = wave + hi * nw;
    if (head >= heads) break;  // wave-uniform
    uint32_t woff_v[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) woff_v[c] = (uint32_t)((((2 * heads + head) * 2 + c) * NHL * 64 + lane) * 16);
    if (hi > 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) woff_qk[c] = (uint32_t)(((((c >> 1) * heads + head) * 2 + (c & 1)) * NHL * 64 + lane) * 16);
      w0_load<PROD, 4>(w0qk, rq, woff_qk);
    }
    // ---- q and k of this head: D[channel 16*dt + 4*lg + r][token 16*tt + li] ----
    bf16x8 qh[4], ql[4], kh[4], kl[4];
    W0<PROD, 2> w0v;
    {
      f32x4 a[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (the second head runs with the first one's output held in 32 registers: single-buffered token fragments there)
      if (hi == 0)
        gemm_tile<PROD, 4, 4, false, true>(a, s_x, LO0, nk, rq, woff_qk, qstep, w0qk, li, lg);
      else
        gemm_tile<PROD, 4, 4, false, false>(a, s_x, LO0, nk, rq, woff_qk, qstep, w0qk, li, lg);
      w0_load<PROD, 2>(w0v, rq, woff_v);
      f32x4 b[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = bqkv4[(((c >> 1) * heads + head) * 2 + (c & 1)) * 4 + lg];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        frag_of(a[0][tt] + b[0], a[1][tt] + b[1], qh[tt], ql[tt]);
        frag_of(a[2][tt] + b[2], a[3][tt] + b[3], kh[tt], kl[tt]);
      }
    }
    // ---- v of this head with the operands swapped: D[token 16*tt + 4*lg + r][channel 16*dt + li] ----
    bf16x8 vh[2][2], vl[2][2];  // [dt][key tile pair]
    {
      f32x4 a[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (hi == 0)
        gemm_tile<PROD, 2, 4, true, true>(a, s_x, LO0, nk, rq, woff_v, qstep, w0v, li, lg);
      else
        gemm_tile<PROD, 2, 4, true, false>(a, s_x, LO0, nk, rq, woff_v, qstep, w0v, li, lg);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float bv = p.bqkv[((2 * heads + head) * 2 + dt) * 16 + li];
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) frag_of(a[dt][2 * kp] + bv, a[dt][2 * kp + 1] + bv, vh[dt][kp], vl[dt][kp]);
      }
    }
    // ---- attention of this head, in registers ----
    f32x4 o[2][4];  // O^T: D[channel 16*dt + 4*lg + r][query 16*qt + li]
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 s[4];  // S^T: D[key 16*kt + 4*lg + r][query 16*qt + li]
      const uint32_t rq_ = (qreg >> (4 * qt)) & 15u;
      float m = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = mfma3<PROD>(kh[kt], kl[kt], qh[qt], ql[qt], (f32x4){0.f, 0.f, 0.f, 0.f});
        const f32x4 bf = ((const f32x4*)p.bias_frag16)[(((int64_t)head * 4 + kt) * 4 + qt) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(s[kt][r], LOG2E, bf[r]);  // base-2 logits: the bias fragments are stored multiplied by log2(e)
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
          const float e = (RSA_SB_ABL & 2) ? s[kt][r] - m : __builtin_amdgcn_exp2f(s[kt][r] - m);  // v_exp_f32: arguments <= 0, underflow to 0 is the wanted result
          s[kt][r] = e;
          l += e;
        }
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
      const float inv_l = 1.f / l;
      bf16x8 ph[2], pl[2];
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) frag_of(s[2 * kp], s[2 * kp + 1], ph[kp], pl[kp]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) acc = mfma3<PROD>(vh[dt][kp], vl[dt][kp], ph[kp], pl[kp], acc);
        o[dt][qt] = acc * inv_l;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int k = 0; k < 2; ++k) pair_units(o[dt][2 * k], o[dt][2 * k + 1], ouh[hi][dt * 2 + k], oul[hi][dt * 2 + k]);
  }
  uint32_t woff_p[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) woff_p[c] = (4 * wave + c < ct_c) ? (uint32_t)(((4 * wave + c) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
  W0<PROD, 4> w0p;
  w0_load<PROD, 4>(w0p, rp, woff_p);
  __syncthreads();  // every wave has read the LayerNorm image for the last time: the attention output may overwrite it
  // ---- attention output -> LDS planes [head*4 + dt*2 + (lg >> 1)][token] ----
#pragma unroll
  for (int hi = 0; hi < HPW; ++hi) {
    const int head = wave + hi * nw;
    if (head >= heads) break;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u = (head * 4 + dt * 2 + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
        s_x[u] = ouh[hi][dt * 2 + k];
        if (PROD == 3) s_x[LO0 + u] = oul[hi][dt * 2 + k];
      }
  }
  __syncthreads();

  // ---- proj + bias + shortcut -> residual stream: wave owns cout tiles 4*wave .. +3 ----
  {
    f32x4 a[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tile<PROD, 4, 4, false>(a, s_x, LO0, heads, rp, woff_p, (uint32_t)ct_c * NHL * 1024u, w0p, li, lg);
    f32x4* o_img = (f32x4*)p.out + (int64_t)n * p4 * HW;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int64_t pix = token_pix(16 * pt + li);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int g = (4 * wave + c) * 4 + lg;
        if (RSA_SB_ABL & 16) {
          asm volatile("" ::"v"(a[c][pt]));
          continue;
        }
        if (pix < 0 || g >= p4) continue;
        const f32x4 b = ((const f32x4*)p.bproj)[g];
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (!(RSA_SB_ABL & 256)) r = x_img[MAPIDX(g, pix)];
        if (RSA_SB_ABL & 128)
          asm volatile("" ::"v"(a[c][pt] + b + r));
        else
          o_img[MAPIDX(g, pix)] = a[c][pt] + b + r;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ MLP half
// One workgroup (8 waves) = 64 consecutive tokens of one image.  (The hidden image takes 120 KB of LDS at 480 channels: one
// workgroup per CU.)
template <int PROD, int FMT = 0>
__global__ __launch_bounds__(512) void swin_mlp_block_kernel(const rsa_swin_mlp_block_params p) {
  constexpr int HPL = 64;  // planes of the LDS image (512 hidden channels)
  constexpr int LO0 = HPL * SB_TOK;
  constexpr int NHL = PROD == 3 ? 2 : 1;
  __shared__ uint4 s_h[2 * HPL * SB_TOK];  // 128 KB: LayerNorm image (planes 0..31), then the hidden map

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t HW = (int64_t)p.H * p.W;
  const int tiles_img = (int)((HW + SB_TOK - 1) / SB_TOK);
  const int n = (int)blockIdx.x / tiles_img;
  const int64_t t0 = (int64_t)((int)blockIdx.x % tiles_img) * SB_TOK;
  const int p4 = (p.C + 3) >> 2;
  const int planes = (p.C + 7) >> 3;
  const int nk1 = (planes + 3) >> 2;
  const int hplanes = (p.hidden + 7) >> 3;
  const int nk2 = (hplanes + 3) >> 2;
  const int ct_h = (p.hidden + 15) >> 4, ct_c = (p.C + 15) >> 4;
  const __amdgpu_buffer_rsrc_t r1 = weight_rsrc(p.w1, (int64_t)nk1 * ct_h * NHL * 1024);
  const __amdgpu_buffer_rsrc_t r2 = weight_rsrc(p.w2, (int64_t)nk2 * ct_c * NHL * 1024);
  uint32_t woff1[4], woff2[2];
#pragma unroll
  for (int c = 0; c < 4; ++c) woff1[c] = (4 * wave + c < ct_h) ? (uint32_t)(((4 * wave + c) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
#pragma unroll
  for (int c = 0; c < 2; ++c) woff2[c] = (2 * wave + c < ct_c) ? (uint32_t)(((2 * wave + c) * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;

  auto token_pix = [&](int t) -> int64_t { return t0 + t < HW ? t0 + t : -1; };
  const f32x4* x_img = (const f32x4*)p.x + (int64_t)n * p4 * HW;

  // ---- LayerNorm -> LDS image (the first K chunk of the fc1 weights is on its way meanwhile).  The shortcut values of the epilogue
  //      are requested right behind the LayerNorm row: the same lines, still in L2, and their latency is over long before the
  //      epilogue (asked for there, every wave waited for them at the end of the tile with nothing left to overlap). ----
  W0<PROD, 4> w01;
  w0_load<PROD, 4>(w01, r1, woff1);
  f32x4 res[2][4];
  {
    const int t = wave * 8 + (lane & 7);
    LnRow row;
    ln_load(row, x_img, HW, p4, token_pix(t), lane);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int g = (2 * wave + c) * 4 + lg;
        const int64_t pix = token_pix(16 * pt + li);
        res[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pix >= 0 && g < p4 && !(RSA_SB_ABL & (16 | 256))) res[c][pt] = x_img[MAPIDX(g, pix)];
      }
    ln_store<PROD, FMT>(row, s_h, LO0, 4 * nk1, p.C, p.gamma, p.beta, p.eps, t, token_pix(t), lane);
  }
  __syncthreads();

  // ---- fc1 + GELU: wave owns hidden cout tiles 4*wave .. +3 ----
  f32x4 a1[4][4];
  W0<PROD, 2> w02;
  {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a1[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tile<PROD, 4, 4, false, true, FMT>(a1, s_h, LO0, nk1, r1, woff1, (uint32_t)ct_h * NHL * 1024u, w01, li, lg);
    w0_load<PROD, 2>(w02, r2, woff2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ct = 4 * wave + c;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (ct < ct_h) b = ((const f32x4*)p.b1)[ct * 4 + lg];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a1[c][pt][r] = (RSA_SB_ABL & 2) ? a1[c][pt][r] + b[r] : gelu_fast(a1[c][pt][r] + b[r]);  // tiles beyond the layer: GELU(0) = 0
    }
  }
  __syncthreads();  // every wave has read the LayerNorm image for the last time
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int ct = 4 * wave + c;
    if (ct >= 2 * nk2) continue;  // planes of the K padding are written (as zeros) too
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      uint4 uh, ul;
      pair_units<FMT>(a1[c][2 * k], a1[c][2 * k + 1], uh, ul);
      const int u = (2 * ct + (lg >> 1)) * SB_TOK + 16 * (2 * k + (lg & 1)) + li;
      s_h[u] = uh;
      if (PROD == 3) s_h[LO0 + u] = ul;
    }
  }
  __syncthreads();

  // ---- fc2 + bias + shortcut: wave owns cout tiles 2*wave, 2*wave + 1 ----
  {
    f32x4 a[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) a[c][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gemm_tile<PROD, 2, 4, false, true, FMT>(a, s_h, LO0, nk2, r2, woff2, (uint32_t)ct_c * NHL * 1024u, w02, li, lg);
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
        if (RSA_SB_ABL & 16) {
          asm volatile("" ::"v"(a[c][pt]));
        } else if (pix >= 0 && g < p4) {
          v = a[c][pt] + b + res[c][pt];
          if (RSA_SB_ABL & 128)
            asm volatile("" ::"v"(v));
          else
            o_img[MAPIDX(g, pix)] = v;
        }
        a[c][pt] = v;
      }
      if (p.out_hi != nullptr) {  // wave-uniform: every lane takes part in the exchange
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          uint4 uh, ul;
          pair_units(a[c][2 * k], a[c][2 * k + 1], uh, ul);
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

static bool aligned16(const void* a) { return ((uintptr_t)a & 15) == 0; }

}  // namespace rsa

extern "C" int rsa_swin_attn_block(const rsa_swin_attn_block_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "swin_attn_block: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->C < 4 || p->heads < 1) return set_error(RSA_E_ARG, "swin_attn_block: bad geometry");
  if (p->window < 1 || p->window > 8) return set_error(RSA_E_UNSUPPORTED, "swin_attn_block: window must be 1..8 (<= 64 tokens)");
  if (p->H % p->window || p->W % p->window) return set_error(RSA_E_ARG, "swin_attn_block: H and W must be multiples of the window");
  if (p->shift < 0 || p->shift >= p->window) return set_error(RSA_E_ARG, "swin_attn_block: shift must be in [0, window)");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "swin_attn_block: products must be 1 or 3");
  if (p->C > 256 || (p->C & 3)) return set_error(RSA_E_UNSUPPORTED, "swin_attn_block: C must be a multiple of 4, at most 256");
  if (p->heads > 8 || p->C % p->heads || p->C / p->heads > 32) return set_error(RSA_E_UNSUPPORTED, "swin_attn_block: at most 8 heads of at most 32 channels");
  if (!p->x || !p->gamma || !p->beta || !p->wqkv || !p->bqkv || !p->bias_frag16 || !p->wproj || !p->bproj || !p->out)
    return set_error(RSA_E_ARG, "swin_attn_block: null pointer");
  if (!aligned16(p->x) || !aligned16(p->gamma) || !aligned16(p->beta) || !aligned16(p->wqkv) || !aligned16(p->bqkv) || !aligned16(p->bias_frag16) ||
      !aligned16(p->wproj) || !aligned16(p->bproj) || !aligned16(p->out))
    return set_error(RSA_E_ALIGN, "swin_attn_block: pointers must be 16-byte aligned");
  const int64_t windows = (int64_t)p->batch * (p->H / p->window) * (p->W / p->window);
  if (windows > 0x3fffffff) return set_error(RSA_E_UNSUPPORTED, "swin_attn_block: too many windows");
  if (p->products == 3)
    hipLaunchKernelGGL(swin_attn_block_kernel<3>, dim3((unsigned)windows), dim3(64 * ((p->heads + 1) / 2)), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL(swin_attn_block_kernel<1>, dim3((unsigned)windows), dim3(64 * ((p->heads + 1) / 2)), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "swin_attn_block: launch failed") : RSA_OK;
}

extern "C" int rsa_swin_mlp_block(const rsa_swin_mlp_block_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "swin_mlp_block: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->C < 4 || p->hidden < 1) return set_error(RSA_E_ARG, "swin_mlp_block: bad geometry");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "swin_mlp_block: products must be 1 or 3");
  if (p->C > 256 || (p->C & 3) || p->hidden > 512) return set_error(RSA_E_UNSUPPORTED, "swin_mlp_block: C must be a multiple of 4, at most 256; hidden at most 512");
  if (!p->x || !p->gamma || !p->beta || !p->w1 || !p->b1 || !p->w2 || !p->b2 || !p->out) return set_error(RSA_E_ARG, "swin_mlp_block: null pointer");
  if (!aligned16(p->x) || !aligned16(p->gamma) || !aligned16(p->beta) || !aligned16(p->w1) || !aligned16(p->b1) || !aligned16(p->w2) || !aligned16(p->b2) ||
      !aligned16(p->out) || !aligned16(p->out_hi) || !aligned16(p->out_lo))
    return set_error(RSA_E_ALIGN, "swin_mlp_block: pointers must be 16-byte aligned");
  const int64_t tiles = (int64_t)p->batch * (((int64_t)p->H * p->W + SB_TOK - 1) / SB_TOK);
  if (tiles > 0x3fffffff) return set_error(RSA_E_UNSUPPORTED, "swin_mlp_block: too many tokens");
  if ((p->fmt != RSA_PF_BF16 && p->fmt != RSA_PF_F16) || p->reserved0 != 0 || (p->fmt == RSA_PF_F16 && p->products != 1))
    return set_error(RSA_E_ARG, "swin_mlp_block: fmt must be an rsa_plane_fmt (fp16: the one-product form only)");
  if (p->products == 3)
    hipLaunchKernelGGL(swin_mlp_block_kernel<3>, dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)stream, *p);
  else if (p->fmt == RSA_PF_F16)
    hipLaunchKernelGGL((swin_mlp_block_kernel<1, RSA_PF_F16>), dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL(swin_mlp_block_kernel<1>, dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "swin_mlp_block: launch failed") : RSA_OK;
}
