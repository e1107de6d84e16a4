// pack.hip — rsa_pack_weights: OIHW f32 convolution weights -> the MFMA A-fragment blob the convolution kernels stream
// (split planes of format fmt, bf16 or fp16: hi = RNE(w), lo = RNE(w - hi)).  Replaces, on the device and behind the C-ABI, what nn.Module.load_state_dict
// does for the reference's nn.Conv2d parameters (reference registry.py:113): the engine's kernels never see OIHW.
//
// One thread per (K step, cout tile, lane) writes that lane's 8-element fragment of hi (and lo).
//   layout RSA_WL_TAPS  (conv_kernel.h, gemm_k1.hip): blob[q][t][ct][hl][lane][j], K step = (chunk q of 32 channels, tap t):
//        cout = 16*ct + (lane & 15), cin = 32*q + 8*(lane >> 4) + j
//   layout RSA_WL_PAIRS (conv_ring.h, 3x3 only): blob[q][s][ct][hl][lane][j], s = 0..8; lane group lg = lane >> 4 reads plane
//        (lg & 1) of a 16-channel half (A = planes 4q, 4q+1; B = 4q+2, 4q+3) at one tap of the step's pair, selected by h = lg >> 1:
//        s = 0,1,2: half A, tap (dy = s, dx = h)      s = 3: half A, tap (dy = h, dx = 2)      s = 4: tap (2,2) of half A (h = 0) / B (h = 1)
//        s = 5,6,7: half B, tap (dy = s-5, dx = h)    s = 8: half B, tap (dy = h, dx = 2)
//   layout RSA_WL_HALFPAIRS (conv_ring.h half mode, an odd number of half chunks): blob[half][s][ct][hl][lane][j], s = 0..4, plane
//        2*half + (lg & 1):  s = 0,1,2: tap (dy = s, dx = h)   s = 3: tap (dy = h, dx = 2)   s = 4: tap (2,2) for h = 0, ZERO for h = 1
//   layout RSA_WL_UPPHASE (conv_ring_up.h: nearest x2 upsampling + 3x3 as four 2x2 convolutions on the source map, 64 -> 64 channels):
//        blob[phase][half][s][ct][hl][lane][j], phase = 2*py + px (output pixel parity), half = 16-channel half chunk 0..3, s = source row
//        0..1, plane 2*half + (lg & 1), source column h = lg >> 1.  The weight of source pixel (s, h) is the SUM (in f32) of the 3x3 taps
//        that read it on the upsampled image: rows {0} / {1,2} for py = 0, {0,1} / {2} for py = 1, columns likewise with px.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 pk_bf16x8;

// one element -> (hi, lo) halves of plane format fmt, as 16-bit patterns
__device__ __forceinline__ void split_elem(float v, int fmt, uint16_t& hi, uint16_t& lo) {
  if (fmt == RSA_PF_F16) {
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)h));
  } else {
    const __bf16 h = (__bf16)v;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
  }
}
typedef __attribute__((ext_vector_type(8))) uint16_t pk_u16x8;

__global__ void pack_weights_kernel(const float* __restrict__ w, int cout, int cin, int cin_planes, int ksize, int products, int layout, int fmt, void* out) {
  const int T = layout == RSA_WL_HALFPAIRS ? 5 : ksize * ksize;                      // K steps per unit
  const int ct_total = (cout + 15) >> 4;
  const int nq = layout == RSA_WL_HALFPAIRS ? (cin_planes >> 1) : (cin_planes + 3) >> 2;  // units: half chunks / chunks
  const int nhl = products == 3 ? 2 : 1;
  const int64_t total = (int64_t)nq * T * ct_total * 64;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    int64_t r = idx >> 6;
    const int ct = (int)(r % ct_total);
    r /= ct_total;
    const int s = (int)(r % T);
    const int q = (int)(r / T);
    const int lg = lane >> 4;
    const int co = 16 * ct + (lane & 15);
    int plane, ky, kx;
    bool zero = false;
    if (layout == RSA_WL_HALFPAIRS) {
      const int h = lg >> 1;
      plane = 2 * q + (lg & 1);
      if (s < 3) {
        ky = s;
        kx = h;
      } else if (s == 3) {
        ky = h;
        kx = 2;
      } else {
        ky = 2;
        kx = 2;
        zero = h != 0;
      }
    } else if (layout == RSA_WL_PAIRS) {
      const int h = lg >> 1;
      const int half = s < 4 ? 0 : (s == 4 ? h : 1);
      plane = 4 * q + 2 * half + (lg & 1);
      const int ss = s < 4 ? s : (s == 4 ? 4 : s - 5);
      if (ss < 3) {
        ky = ss;
        kx = h;
      } else if (ss == 3) {
        ky = h;
        kx = 2;
      } else {
        ky = 2;
        kx = 2;
      }
    } else {
      plane = 4 * q + lg;
      ky = s / ksize;
      kx = s - ky * ksize;
    }
    pk_u16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = 8 * plane + j;
      float v = 0.f;
      if (!zero && co < cout && ci < cin && plane < cin_planes) v = w[(((int64_t)co * cin + ci) * ksize + ky) * ksize + kx];
      uint16_t h, l;
      split_elem(v, fmt, h, l);
      hi[j] = h;
      lo[j] = l;
    }
    const int64_t frag = (((int64_t)q * T + s) * ct_total + ct) * nhl;  // 1 KiB fragments
    ((pk_u16x8*)out)[frag * 64 + lane] = hi;
    if (nhl == 2) ((pk_u16x8*)out)[(frag + 1) * 64 + lane] = lo;
  }
}

__global__ void pack_weights_upphase_kernel(const float* __restrict__ w, int cout, int cin, void* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (K step t 0..31, cout tile 0..3, lane)
  if (idx >= 32 * 4 * 64) return;
  const int lane = idx & 63, ct = (idx >> 6) & 3, t = idx >> 8;
  const int phase = t >> 3, half = (t >> 1) & 3, s = t & 1;
  const int py = phase >> 1, px = phase & 1;
  const int lg = lane >> 4, h = lg >> 1;
  const int co = 16 * ct + (lane & 15);
  const int plane = 2 * half + (lg & 1);
  // taps (of the 3x3 kernel on the upsampled image) that read source row s / source column h of this phase
  const int ky0 = py == 0 ? (s == 0 ? 0 : 1) : (s == 0 ? 0 : 2), ky1 = py == 0 ? (s == 0 ? 0 : 2) : (s == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (h == 0 ? 0 : 1) : (h == 0 ? 0 : 2), kx1 = px == 0 ? (h == 0 ? 0 : 2) : (h == 0 ? 1 : 2);
  pk_bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 8 * plane + j;
    float v = 0.f;
    if (co < cout && ci < cin)
      for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) v += w[(((int64_t)co * cin + ci) * 3 + ky) * 3 + kx];
    const __bf16 hb = (__bf16)v;
    hi[j] = hb;
    lo[j] = (__bf16)(v - (float)hb);
  }
  const int64_t frag = ((int64_t)t * 4 + ct) * 2;
  ((pk_bf16x8*)out)[frag * 64 + lane] = hi;
  ((pk_bf16x8*)out)[(frag + 1) * 64 + lane] = lo;
}

int64_t packed_weight_bytes(int cout, int cin_planes, int ksize, int products, int layout) {
  if (cout < 1 || cin_planes < 1 || (ksize != 1 && ksize != 3) || (products != 1 && products != 3)) return RSA_E_ARG;
  if (layout == RSA_WL_UPPHASE) return (cout == 64 && cin_planes == 8 && ksize == 3 && products == 3) ? (int64_t)32 * 4 * 2 * 64 * 16 : (int64_t)RSA_E_UNSUPPORTED;
  if (layout < RSA_WL_TAPS || layout > RSA_WL_UPPHASE) return RSA_E_ARG;
  const int64_t ct = (cout + 15) / 16;
  const int64_t chunks = (cin_planes + 3) / 4;
  const int64_t nhl = products == 3 ? 2 : 1;
  return chunks * ksize * ksize * ct * nhl * 64 * 16;
}

}  // namespace rsa

extern "C" int64_t rsa_packed_weight_bytes_layout(int32_t cout, int32_t cin_planes, int32_t ksize, int32_t products, int32_t layout) {
  return rsa::packed_weight_bytes(cout, cin_planes, ksize, products, layout);
}

extern "C" int rsa_pack_weights(const float* w_oihw, int32_t cout, int32_t cin, int32_t cin_planes, int32_t ksize, int32_t products, int32_t layout,
                                int32_t fmt, void* out, void* stream) {
  if (fmt != RSA_PF_BF16 && fmt != RSA_PF_F16) return rsa::set_error(RSA_E_ARG, "pack_weights: fmt must be an rsa_plane_fmt");
  if (w_oihw == nullptr || out == nullptr || cout < 1 || cin < 1 || cin_planes < 1) return rsa::set_error(RSA_E_ARG, "pack_weights: bad argument");
  if ((ksize != 1 && ksize != 3) || (products != 1 && products != 3)) return rsa::set_error(RSA_E_ARG, "pack_weights: ksize must be 1 or 3, products 1 or 3");
  if (cin > 8 * cin_planes) return rsa::set_error(RSA_E_ARG, "pack_weights: cin does not fit in cin_planes");
  if (layout < rsa::RSA_WL_TAPS || layout > rsa::RSA_WL_UPPHASE) return rsa::set_error(RSA_E_ARG, "pack_weights: unknown layout");
  if (layout == rsa::RSA_WL_UPPHASE) {
    if (ksize != 3 || products != 3 || cin_planes != 8 || cout != 64 || fmt != RSA_PF_BF16)
      return rsa::set_error(RSA_E_UNSUPPORTED, "pack_weights: the upsampling-phase layout is for 3x3, 3-product, bf16, 64 -> 64 channel layers");
    if ((uintptr_t)out & 15) return rsa::set_error(RSA_E_ALIGN, "pack_weights: out must be 16-byte aligned");
    hipLaunchKernelGGL(rsa::pack_weights_upphase_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, out);
    const int rcu = (int)hipGetLastError();
    return rcu ? rsa::set_error(rcu, "pack_weights: launch failed") : RSA_OK;
  }
  if (layout == rsa::RSA_WL_PAIRS && (ksize != 3 || (cin_planes & 3))) return rsa::set_error(RSA_E_UNSUPPORTED, "pack_weights: the tap-pair layout needs a 3x3 layer with whole 32-channel chunks");
  if (layout == rsa::RSA_WL_HALFPAIRS && (ksize != 3 || (cin_planes & 1))) return rsa::set_error(RSA_E_UNSUPPORTED, "pack_weights: the half-chunk tap-pair layout needs a 3x3 layer with an even number of input planes");
  if ((uintptr_t)out & 15) return rsa::set_error(RSA_E_ALIGN, "pack_weights: out must be 16-byte aligned");
  const int nhl = products == 3 ? 2 : 1;
  const int64_t total = (layout == rsa::RSA_WL_HALFPAIRS ? (int64_t)(cin_planes / 2) * 5 : (int64_t)((cin_planes + 3) / 4) * ksize * ksize) * ((cout + 15) / 16) * 64;
  int64_t grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (layout == rsa::RSA_WL_HALFPAIRS) {  // 5 K steps per half chunk < 9 per chunk: the rest of the (fixed-size) blob is defined as zero
    const int64_t used = total * nhl * 16, size = (int64_t)((cin_planes + 3) / 4) * 9 * ((cout + 15) / 16) * nhl * 64 * 16;
    if (size > used) {
      const int rc0 = (int)hipMemsetAsync((char*)out + used, 0, (size_t)(size - used), (hipStream_t)stream);
      if (rc0) return rsa::set_error(rc0, "pack_weights: memset failed");
    }
  }
  hipLaunchKernelGGL(rsa::pack_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, cin_planes, ksize, products,
                     layout, fmt, out);
  const int rc = (int)hipGetLastError();
  return rc ? rsa::set_error(rc, "pack_weights: launch failed") : RSA_OK;
}
