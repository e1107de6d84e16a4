// swin.hip — the two non-convolution kernels of the SwinIR path:
//   rsa_layernorm         nn.LayerNorm over channels of a token map        (reference archs/swinir/arch.py:306,333,959)
//   rsa_window_attention  (shifted) window multi-head self-attention core  (reference archs/swinir/arch.py:133-173 between
//                         the qkv and proj Linear layers, plus torch.roll / window_partition / window_reverse /
//                         calculate_mask of SwinTransformerBlock.forward :295-335, which become index arithmetic)
//
// Tokens are pixels: the residual stream is the f32 NCHW4c map, Linear layers are k1 convolutions of conv_mfma.hip.
// The qkv projection is packed so that every head owns 32 channels (head_dim zero-padded to 32 = 4 planes = one MFMA K):
//   planes [ (which*heads + head)*4 , +4 )  with which = 0 q (pre-scaled), 1 k, 2 v.
//
// One wave = one (window, head).  With the key index on MFMA rows:
//   S^T = K Q^T     v_mfma_f32_32x32x16_bf16, A = K fragment, B = Q fragment: both are plain 16-byte unit loads
//   softmax         every lane owns ONE query column: row max / row sum are in-lane loops + one exchange with lane^32
//   O^T = V^T P^T   the accumulator tile of S^T is already the B operand (rows = keys = the summed index); V^T comes
//                   from a row-major LDS image of V through ds_read_b64_tr_b16 (hardware transpose read)
//   store           O^T's accumulator layout gives each lane 4 consecutive channels of its token: 8-byte plane writes
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------------------------------------ LayerNorm
// One workgroup (4 waves) = 64 consecutive tokens; lane = token, wave w owns channel planes w, w+4, w+8, ...  Every load
// and store instruction of a wave is then 64 consecutive tokens of one plane = 1 KiB contiguous in HBM.  A wave keeps its
// planes in registers (up to LN_MAXP planes = 8*4*LN_MAXP = 256 channels), so the f32 map is read exactly once; mean and the
// centred variance are each closed by a 4-way reduction through LDS.
constexpr int LN_MAXP = 8;
#ifndef LN_WAVES
#define LN_WAVES 4  // waves per SIMD the register allocation must allow (left alone the compiler took 233 VGPRs: two workgroups per CU)
#endif
constexpr int LN_LDS_C = 1024;  // gamma / beta staged in LDS up to this many channels

template <bool IN_REGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(LN_WAVES, 8))) void layernorm_kernel(const rsa_layernorm_params p) {
  __shared__ float s_red[2][4][64];
  // gamma / beta of every channel in LDS, zeros beyond C: the per-channel global loads of the output phase (a lane-dependent channel index, so
  // vector loads, each waited for: 48 round trips per wave) were 2/3 of the kernel's time (profiles/r04_u_layernorm_probe.txt)
  __shared__ __attribute__((aligned(16))) float s_gb[2][LN_LDS_C];
  const bool gb_lds = IN_REGS || p.C <= LN_LDS_C;  // (IN_REGS: C <= 32 * LN_MAXP)
  if (gb_lds) {
    for (int c = threadIdx.x; c < ((p.C + 7) & ~7); c += 256) {
      s_gb[0][c] = c < p.C ? p.gamma[c] : 0.f;
      s_gb[1][c] = c < p.C ? p.beta[c] : 0.f;
    }
    __syncthreads();
  }
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t total = (int64_t)p.batch * HW;
  const int p4 = (p.C + 3) >> 2;
  const int planes = (p.C + 7) >> 3;
  const float inv_c = 1.f / (float)p.C;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t nblk = (total + 63) >> 6;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t idx = blk * 64 + lane;
    const bool live = idx < total;
    const int64_t pix = live ? idx % HW : 0;
    const int n = live ? (int)(idx / HW) : 0;
    const f32x4* x = (const f32x4*)p.x_f32 + (int64_t)n * p4 * HW + pix;
    f32x4 v[LN_MAXP][2];
    float sum = 0.f;
    if (IN_REGS) {
#pragma unroll
      for (int k = 0; k < LN_MAXP; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int g = (wave + 4 * k) * 2 + h;
          v[k][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (g < p4) v[k][h] = x[(int64_t)g * HW];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (g * 4 + r < p.C) sum += v[k][h][r];
        }
    } else {
      for (int pl = wave; pl < planes; pl += 4)
        for (int h = 0; h < 2; ++h) {
          const int g = pl * 2 + h;
          if (g >= p4) continue;
          const f32x4 t = x[(int64_t)g * HW];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (g * 4 + r < p.C) sum += t[r];
        }
    }
    s_red[0][wave][lane] = sum;
    __syncthreads();
    const float mean = (s_red[0][0][lane] + s_red[0][1][lane] + s_red[0][2][lane] + s_red[0][3][lane]) * inv_c;
    float var = 0.f;
    if (IN_REGS) {
#pragma unroll
      for (int k = 0; k < LN_MAXP; ++k)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int g = (wave + 4 * k) * 2 + h;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (g * 4 + r < p.C) {
              const float d = v[k][h][r] - mean;
              var += d * d;
            }
        }
    } else {
      for (int pl = wave; pl < planes; pl += 4)
        for (int h = 0; h < 2; ++h) {
          const int g = pl * 2 + h;
          if (g >= p4) continue;
          const f32x4 t = x[(int64_t)g * HW];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (g * 4 + r < p.C) {
              const float d = t[r] - mean;
              var += d * d;
            }
        }
    }
    s_red[1][wave][lane] = var;
    __syncthreads();  // also orders the next iteration's s_red[0] writes after this iteration's reads
    const float rstd = rsqrtf((s_red[1][0][lane] + s_red[1][1][lane] + s_red[1][2][lane] + s_red[1][3][lane]) * inv_c + p.eps);
    auto emit = [&](int pl, const f32x4 a, const f32x4 b) {
      float y[8];
      if (gb_lds) {
        const f32x4 g0 = *(const f32x4*)&s_gb[0][pl * 8], g1 = *(const f32x4*)&s_gb[0][pl * 8 + 4];
        const f32x4 b0 = *(const f32x4*)&s_gb[1][pl * 8], b1 = *(const f32x4*)&s_gb[1][pl * 8 + 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = (a[j] - mean) * rstd * g0[j] + b0[j];  // channels beyond C: gamma = beta = 0
          y[4 + j] = (b[j] - mean) * rstd * g1[j] + b1[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = pl * 8 + j;
          const float t = j < 4 ? a[j] : b[j - 4];
          y[j] = (c < p.C) ? (t - mean) * rstd * p.gamma[c] + p.beta[c] : 0.f;
        }
      }
      if (p.out_f32 != nullptr) {
        if (pl * 2 < p4) ((f32x4*)p.out_f32)[((int64_t)n * p4 + pl * 2) * HW + pix] = (f32x4){y[0], y[1], y[2], y[3]};
        if (pl * 2 + 1 < p4) ((f32x4*)p.out_f32)[((int64_t)n * p4 + pl * 2 + 1) * HW + pix] = (f32x4){y[4], y[5], y[6], y[7]};
      }
      if (p.out_hi != nullptr) {
        typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
        u16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (p.out_fmt == RSA_PF_F16) {
            float yj = y[j];
            asm("" : "+v"(yj));  // opaque: see conv_common.h, split2
            const _Float16 hb = (_Float16)yj;
            h[j] = __builtin_bit_cast(uint16_t, hb);
            l[j] = __builtin_bit_cast(uint16_t, (_Float16)(yj - (float)hb));
          } else {
            const __bf16 hb = (__bf16)y[j];
            h[j] = __builtin_bit_cast(uint16_t, hb);
            l[j] = __builtin_bit_cast(uint16_t, (__bf16)(y[j] - (float)hb));
          }
        }
        const int64_t unit = (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride + pix;
        ((u16x8*)p.out_hi)[unit] = h;
        if (p.out_lo != nullptr) ((u16x8*)p.out_lo)[unit] = l;
      }
    };
    if (live) {
      if (IN_REGS) {
#pragma unroll
        for (int k = 0; k < LN_MAXP; ++k) {
          const int pl = wave + 4 * k;
          if (pl < planes) emit(pl, v[k][0], v[k][1]);
          __builtin_amdgcn_sched_barrier(0);  // one plane at a time: hoisting every plane's gamma / beta reads cost 120 registers
        }
      } else {
        for (int pl = wave; pl < planes; pl += 4) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
          if (pl * 2 < p4) a = x[(int64_t)(pl * 2) * HW];
          if (pl * 2 + 1 < p4) b = x[(int64_t)(pl * 2 + 1) * HW];
          emit(pl, a, b);
        }
      }
    }
  }
}

// two values -> packed hi pair and packed lo pair of a plane format (as conv_common.h::split2; the fp16 form is opaque to the contraction pass)
template <int FMT>
__device__ __forceinline__ void ln_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  if constexpr (FMT == RSA_PF_F16) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    asm("" : "+v"(a), "+v"(b));
    const f16x2 h = {(_Float16)a, (_Float16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const f16x2 l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    lo = __builtin_bit_cast(uint32_t, l);
  } else {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    hi = __builtin_bit_cast(uint32_t, h);
    const bf16x2 l = {(__bf16)(a - __builtin_bit_cast(float, hi << 16)), (__bf16)(b - __builtin_bit_cast(float, hi & 0xffff0000u))};
    lo = __builtin_bit_cast(uint32_t, l);
  }
}

// C <= 32 * MAXP (MAXP 8: every LayerNorm of the SwinIR / HAT / DAT bodies; 12: DRCT's dense blocks, up to 308 channels), round 4: the same work split with everything wave-uniform kept in scalar
// registers -- the wave id through readfirstlane, 64-token blocks that never straddle two images (so the image index and every plane base
// are uniform and a lane's address is base + 16 * pixel), whole-plane validity as scalar branches -- and gamma / beta read from LDS.
// 76 VGPRs (the first form: 233, two workgroups per CU, and 48 waited-for gamma / beta loads per wave): 121 -> 45 us for 180 channels at 512^2.
template <int MAXP>
__global__ __launch_bounds__(256) void layernorm_regs_kernel(const rsa_layernorm_params p) {
  __shared__ float s_red[2][4][64];
  __shared__ __attribute__((aligned(16))) float s_gb[2][32 * MAXP];
  bool staged = false;
  const uint32_t HW = (uint32_t)p.H * (uint32_t)p.W;
  const int p4 = (p.C + 3) >> 2;
  const int planes = (p.C + 7) >> 3;
  const float inv_c = 1.f / (float)p.C;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t bpi = (HW + 63u) >> 6;  // blocks per image
  const uint32_t nblk = bpi * (uint32_t)p.batch;
  for (uint32_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const uint32_t n = blk / bpi;
    const uint32_t pix = (blk - n * bpi) * 64u + (uint32_t)lane;
    const bool live = pix < HW;
    const uint32_t loff = (live ? pix : 0u) * 16u;  // byte offset of this lane's token inside a group plane of the f32 map / a 16-byte-unit plane
    const char* xb = (const char*)p.x_f32 + (size_t)n * p4 * HW * 16;
    f32x4 v[MAXP][2];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = (wave + 4 * k) * 2 + h;  // uniform
        v[k][h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g < p4) {
          v[k][h] = *(const f32x4*)(xb + (size_t)g * HW * 16 + loff);
          if (g * 4 + 3 >= p.C) {  // the last group of a C that is not a multiple of 4
#pragma unroll
            for (int r = 1; r < 4; ++r)
              if (g * 4 + r >= p.C) v[k][h][r] = 0.f;
          }
        }
      }
    if (!staged) {  // behind the loads of the first block, in front of the first barrier: its latency hides under theirs
#pragma unroll
      for (int c = threadIdx.x; c < 32 * MAXP; c += 256) {
        const float gv = c < p.C ? p.gamma[c] : 0.f, bv = c < p.C ? p.beta[c] : 0.f;  // zeros beyond C: padded channels come out as 0
        s_gb[0][c] = gv;
        s_gb[1][c] = bv;
      }
      staged = true;
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) sum += (v[k][h][0] + v[k][h][1]) + (v[k][h][2] + v[k][h][3]);
    s_red[0][wave][lane] = sum;
    __syncthreads();
    const float mean = (s_red[0][0][lane] + s_red[0][1][lane] + s_red[0][2][lane] + s_red[0][3][lane]) * inv_c;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int g = (wave + 4 * k) * 2 + h;
        if (g < p4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = v[k][h][r] - mean;
            var += (g * 4 + r < p.C) ? d * d : 0.f;  // uniform test
          }
        }
      }
    s_red[1][wave][lane] = var;
    __syncthreads();  // also orders the next iteration's s_red[0] writes after this iteration's reads
    const float rstd = rsqrtf((s_red[1][0][lane] + s_red[1][1][lane] + s_red[1][2][lane] + s_red[1][3][lane]) * inv_c + p.eps);
    const float nm = -mean * rstd;
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int pl = wave + 4 * k;  // uniform
      if (pl >= planes) break;
      const f32x4 g0 = *(const f32x4*)&s_gb[0][pl * 8], g1 = *(const f32x4*)&s_gb[0][pl * 8 + 4];
      const f32x4 b0 = *(const f32x4*)&s_gb[1][pl * 8], b1 = *(const f32x4*)&s_gb[1][pl * 8 + 4];
      float y[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (v[k][0][j] * rstd + nm) * g0[j] + b0[j];
        y[4 + j] = (v[k][1][j] * rstd + nm) * g1[j] + b1[j];
      }
      if (!live) continue;
      if (p.out_f32 != nullptr) {
        char* ob = (char*)p.out_f32 + ((size_t)n * p4 + (size_t)pl * 2) * HW * 16 + loff;
        if (pl * 2 < p4) *(f32x4*)ob = (f32x4){y[0], y[1], y[2], y[3]};
        if (pl * 2 + 1 < p4) *(f32x4*)(ob + (size_t)HW * 16) = (f32x4){y[4], y[5], y[6], y[7]};
      }
      if (p.out_hi != nullptr) {
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p.out_fmt == RSA_PF_F16)
            ln_split2<RSA_PF_F16>(y[2 * j], y[2 * j + 1], h[j], l[j]);
          else
            ln_split2<RSA_PF_BF16>(y[2 * j], y[2 * j + 1], h[j], l[j]);
        }
        const size_t unit = ((size_t)n * p.out_batch_stride + (size_t)pl * p.out_plane_stride) * 16 + loff;
        *(uint4*)((char*)p.out_hi + unit) = make_uint4(h[0], h[1], h[2], h[3]);
        if (p.out_lo != nullptr) *(uint4*)((char*)p.out_lo + unit) = make_uint4(l[0], l[1], l[2], l[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ window attention
__device__ __forceinline__ bf16x8 pack_hi(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
  return r;
}

template <int PROD>
__global__ __launch_bounds__(256, 2) void window_attention_kernel(const rsa_window_attn_params p) {
  // wave-private V images: [hi|lo][64 keys][32 channels] bf16, row-major (64-byte rows)
  __shared__ __attribute__((aligned(16))) __bf16 s_v[4][2][64 * 32];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = p.window;
  const int ntok = w * w;
  const int nwx = p.W / w, nwy = p.H / w;
  const int64_t nwin = (int64_t)p.batch * nwy * nwx;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  if (item >= nwin * p.heads) return;  // whole wave exits: every later instruction runs with EXEC all ones
  const int head = (int)(item % p.heads);
  const int64_t win = item / p.heads;
  const int wx = (int)(win % nwx);
  const int wy = (int)((win / nwx) % nwy);
  const int n = (int)(win / ((int64_t)nwx * nwy));

  const int lr = lane & 31;  // token column / channel row owned inside a 32-wide tile
  const int lh = lane >> 5;  // which half of the k-group / row-group

  // pixel of window token t after the cyclic shift (torch.roll(-s) then partition; the result is rolled back, so the
  // output of token t lands on the same pixel it was read from)
  auto token_pix = [&](int t) -> int64_t {
    const int tt = t < ntok ? t : 0;
    const int ty = tt / w, tx = tt - ty * w;
    int py = wy * w + ty + p.shift;
    int px = wx * w + tx + p.shift;
    if (py >= p.H) py -= p.H;
    if (px >= p.W) px -= p.W;
    return (int64_t)py * p.W + px;
  };

  const bf16x8* qkv_hi = (const bf16x8*)p.qkv_hi + (int64_t)n * p.qkv_batch_stride;
  const bf16x8* qkv_lo = (PROD == 3) ? (const bf16x8*)p.qkv_lo + (int64_t)n * p.qkv_batch_stride : nullptr;
  const int64_t ps = p.qkv_plane_stride;
  const int64_t q_plane0 = (int64_t)(0 * p.heads + head) * 4;
  const int64_t k_plane0 = (int64_t)(1 * p.heads + head) * 4;
  const int64_t v_plane0 = (int64_t)(2 * p.heads + head) * 4;

  // ---- fragments of K (A operand) and Q (B operand): token = 32*tile + lr, channels 16*s + 8*lh .. +7 = plane 2s+lh ----
  int64_t pix_t[2];
  pix_t[0] = token_pix(lr);
  pix_t[1] = token_pix(32 + lr);
  bf16x8 kh[2][2], qh[2][2], kl[2][2], ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int64_t uk = (k_plane0 + 2 * s + lh) * ps + pix_t[t];
      const int64_t uq = (q_plane0 + 2 * s + lh) * ps + pix_t[t];
      kh[t][s] = qkv_hi[uk];
      qh[t][s] = qkv_hi[uq];
      if (PROD == 3) {
        kl[t][s] = qkv_lo[uk];
        ql[t][s] = qkv_lo[uq];
      }
    }
  // ---- V image into LDS: lane = key token, 4 planes of 8 channels ----
  {
    const int64_t pv = token_pix(lane);
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const int64_t u = (v_plane0 + pl) * ps + pv;
      *(bf16x8*)&s_v[wave][0][lane * 32 + pl * 8] = qkv_hi[u];
      if (PROD == 3) *(bf16x8*)&s_v[wave][1][lane * 32 + pl * 8] = qkv_lo[u];
    }
  }

  // ---- S^T[key][query] tiles ----
  f32x16 st[2][2];  // [key tile][query tile]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (PROD == 3) {
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl[kt][s], qh[qt][s], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[kt][s], ql[qt][s], a, 0, 0, 0);
        }
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[kt][s], qh[qt][s], a, 0, 0, 0);
      }
      st[kt][qt] = a;
    }

  // ---- + relative position bias (pre-gathered in fragment order; padded keys carry -1e30) + shift mask, softmax ----
  // accumulator element r of lane (lr, lh): key = 32*kt + (r&3) + 8*(r>>2) + 4*lh, query = 32*qt + lr
  const bool masked = p.shift > 0 && (wy == nwy - 1 || wx == nwx - 1);
  auto region = [&](int t) -> int {  // img_mask region id of a window token on the SHIFTED grid (arch.py:268-293)
    const int tt = t < ntok ? t : 0;
    const int ty = tt / w, tx = tt - ty * w;
    const int gy = wy * w + ty, gx = wx * w + tx;
    const int ry = gy < p.H - w ? 0 : (gy < p.H - p.shift ? 1 : 2);
    const int rx = gx < p.W - w ? 0 : (gx < p.W - p.shift ? 1 : 2);
    return ry * 3 + rx;
  };
  float inv_l[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int rq = masked ? region(32 * qt + lr) : 0;
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x4* bf = (const f32x4*)(p.bias_frag + ((((int64_t)head * 2 + qt) * 2 + kt) * 64 + lane) * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = bf[g];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g * 4 + e;
          float v = st[kt][qt][r] + b[e];
          if (masked) {
            const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (region(key) != rq) v += -100.f;
          }
          st[kt][qt][r] = v;
          m = fmaxf(m, v);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(st[kt][qt][r] - m);
        st[kt][qt][r] = e;
        l += e;
      }
    l += __shfl_xor(l, 32);
    inv_l[qt] = 1.f / l;
  }

  // ---- O^T[channel][query] = sum_keys V^T P^T : A = V^T via transpose reads, B = P^T straight from the accumulators ----
  f32x16 ot[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[qt][r] = 0.f;
  const int g16 = lane >> 4;  // 16-lane group: channel half = g16&1, k-half = g16>>1 (== lh)
  const int li16 = lane & 15;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // V^T fragment: element j <-> key 32kt + 16s + 8(j>>2) + 4lh + (j&3), channel lr (the k-order of the accumulator-as-operand)
      bf16x8 vh, vl;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int row = 32 * kt + 16 * s + 8 * g2 + 4 * lh + (li16 >> 2);
        const int col = 16 * (g16 & 1) + 4 * (li16 & 3);
        const bf16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[wave][0][row * 32 + col]);
#pragma unroll
        for (int e = 0; e < 4; ++e) vh[g2 * 4 + e] = th[e];
        if (PROD == 3) {
          const bf16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)&s_v[wave][1][row * 32 + col]);
#pragma unroll
          for (int e = 0; e < 4; ++e) vl[g2 * 4 + e] = tl[e];
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float e8[8], r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e8[j] = st[kt][qt][8 * s + j];
        const bf16x8 ph = pack_hi(e8);
        if (PROD == 3) {
#pragma unroll
          for (int j = 0; j < 8; ++j) r8[j] = e8[j] - (float)ph[j];
          const bf16x8 pl = pack_hi(r8);
          ot[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, ot[qt], 0, 0, 0);
          ot[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, ot[qt], 0, 0, 0);
        }
        ot[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, ot[qt], 0, 0, 0);
      }
    }

  // ---- normalise and store: lane owns token 32qt+lr, channels 8g + 4lh .. +3 (g = 0..3) ----
  char* out_hi = (char*)p.out_hi + (int64_t)n * p.out_batch_stride * 16;
  char* out_lo = (p.out_lo != nullptr) ? (char*)p.out_lo + (int64_t)n * p.out_batch_stride * 16 : nullptr;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int tok = 32 * qt + lr;
    if (tok >= ntok) continue;
    const int64_t pix = pix_t[qt];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = ot[qt][g * 4 + e] * inv_l[qt];
        const __bf16 hb = (__bf16)v;
        h[e] = hb;
        l[e] = (__bf16)(v - (float)hb);
      }
      const int64_t off = (((int64_t)head * 4 + g) * p.out_plane_stride + pix) * 16 + lh * 8;
      *(bf16x4*)(out_hi + off) = h;
      if (out_lo != nullptr) *(bf16x4*)(out_lo + off) = l;
    }
  }
}

}  // namespace rsa

extern "C" int rsa_layernorm(const rsa_layernorm_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "layernorm: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->C < 1) return set_error(RSA_E_ARG, "layernorm: bad geometry");
  if (!p->x_f32 || !p->gamma || !p->beta || (!p->out_hi && !p->out_f32)) return set_error(RSA_E_ARG, "layernorm: null pointer");
  if ((p->out_fmt != RSA_PF_BF16 && p->out_fmt != RSA_PF_F16) || p->reserved0 != 0) return set_error(RSA_E_ARG, "layernorm: out_fmt must be an rsa_plane_fmt");
  if (((uintptr_t)p->x_f32 | (uintptr_t)p->out_hi | (uintptr_t)p->out_lo | (uintptr_t)p->out_f32) & 15)
    return set_error(RSA_E_ALIGN, "layernorm: maps must be 16-byte aligned");
  const int64_t total = (int64_t)p->batch * p->H * p->W;
  int64_t g = (total + 63) / 64;  // one 256-thread workgroup per 64 tokens
  if (g > 256 * 32) g = 256 * 32;
  if (p->C <= 32 * 12 && (int64_t)p->H * p->W < (1ll << 27)) {
    g = (int64_t)p->batch * (((int64_t)p->H * p->W + 63) / 64);  // blocks of 64 tokens of ONE image
    if (g > 256 * 32) g = 256 * 32;
    if (p->C <= 32 * LN_MAXP)
      hipLaunchKernelGGL(layernorm_regs_kernel<LN_MAXP>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *p);
    else
      hipLaunchKernelGGL(layernorm_regs_kernel<12>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *p);
  } else if (p->C <= 32 * LN_MAXP)
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "layernorm: launch failed") : RSA_OK;
}

extern "C" int rsa_window_attention(const rsa_window_attn_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "window_attention: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->heads < 1) return set_error(RSA_E_ARG, "window_attention: bad geometry");
  if (p->window < 1 || p->window > 8) return set_error(RSA_E_UNSUPPORTED, "window_attention: window must be 1..8 (<= 64 tokens)");
  if (p->H % p->window || p->W % p->window) return set_error(RSA_E_ARG, "window_attention: H and W must be multiples of the window");
  if (p->shift < 0 || p->shift >= p->window) return set_error(RSA_E_ARG, "window_attention: shift must be in [0, window)");
  if (p->products != 1 && p->products != 3) return set_error(RSA_E_UNSUPPORTED, "window_attention: products must be 1 or 3");
  if (!p->qkv_hi || !p->bias_frag || !p->out_hi || (p->products == 3 && !p->qkv_lo)) return set_error(RSA_E_ARG, "window_attention: null pointer");
  if (((uintptr_t)p->qkv_hi | (uintptr_t)p->qkv_lo | (uintptr_t)p->out_hi | (uintptr_t)p->out_lo | (uintptr_t)p->bias_frag) & 15)
    return set_error(RSA_E_ALIGN, "window_attention: pointers must be 16-byte aligned");
  const int64_t items = (int64_t)p->batch * (p->H / p->window) * (p->W / p->window) * p->heads;
  const int64_t blocks = (items + 3) / 4;
  if (blocks > 0x7fffffff) return set_error(RSA_E_UNSUPPORTED, "window_attention: too many windows");
  if (p->products == 3)
    hipLaunchKernelGGL(window_attention_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p);
  else
    hipLaunchKernelGGL(window_attention_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "window_attention: launch failed") : RSA_OK;
}
