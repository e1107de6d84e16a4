// capi.hip — extern "C" surface of libresselt_amd.so (see include/resselt_amd.h) and the
// layout-conversion kernels either side of the convolution path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

static thread_local char g_err[256] = "";

int set_error(int code, const char* msg) {
  if (code > 0)
    snprintf(g_err, sizeof(g_err), "%s (hipError %d: %s)", msg, code, hipGetErrorString((hipError_t)code));
  else
    snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <typename T>
__device__ __forceinline__ float ld_as_float(const void* p, int64_t i) {
  return (float)((const T*)p)[i];
}

// one thread per (n, plane, y, x) unit: gathers 8 channels of one pixel from the plain NCHW tensor
// (H, W) = size of the planes; (srcH, srcW) <= (H, W) = size of x: rows/columns past the source are filled by REFLECTION
// (F.pad(..., 'reflect') to the right/bottom, resselt/utilities/padding.py:24-29 as used by SwinIR.check_image_size)
// unshuffle = r > 1: torch.pixel_unshuffle(x, r) fused into the read: plane channel c*r*r + i*r + j at (y, x) comes from source
// channel c at (y*r + i, x*r + j) (RRDBNet x2plus / x1 front end, archs/esrgan/arch.py:130-137); C is then the SOURCE channel count.
typedef __attribute__((ext_vector_type(8))) uint16_t u16x8;
__device__ __forceinline__ void split_elem(float v, int fmt, uint16_t& hi, uint16_t& lo) {  // (hi, lo) halves of plane format fmt
  if (fmt == RSA_PF_F16) {
    asm("" : "+v"(v));  // opaque: see conv_common.h, split2
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (_Float16)(v - (float)h));
  } else {
    const __bf16 h = (__bf16)v;
    hi = __builtin_bit_cast(uint16_t, h);
    lo = __builtin_bit_cast(uint16_t, (__bf16)(v - (float)h));
  }
}

__global__ void nchw_to_planes_kernel(const void* x, int dtype, int batch, int C, int H, int W, int srcH, int srcW, int unshuffle,
                                      const float* mean, float scale, void* out_hi, void* out_lo, int64_t plane_stride,
                                      int64_t batch_stride, int fmt) {
  const int r2 = unshuffle * unshuffle;
  const int Cp = C * r2;  // channels of the planes
  const int planes = (Cp + 7) >> 3;
  const int64_t HW = (int64_t)H * W;
  const int64_t sHW = (int64_t)srcH * srcW;
  const int64_t total = (int64_t)batch * planes * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx % HW;
    const int64_t t = idx / HW;
    const int pl = (int)(t % planes);
    const int n = (int)(t / planes);
    const int py = (int)(pix / W), px = (int)(pix % W);
    u16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cp = pl * 8 + j;
      float v = 0.f;
      if (cp < Cp) {
        const int c = cp / r2;
        const int rem = cp - c * r2;
        int sy = py * unshuffle + rem / unshuffle, sx = px * unshuffle + rem % unshuffle;
        if (sy >= srcH) sy = 2 * (srcH - 1) - sy;
        if (sx >= srcW) sx = 2 * (srcW - 1) - sx;
        const int64_t src = ((int64_t)n * C + c) * sHW + (int64_t)sy * srcW + sx;
        if (dtype == RSA_U8)  // 8-bit image, channel-interleaved [N][srcH][srcW][C]
          v = (float)((const uint8_t*)x)[(((int64_t)n * srcH + sy) * srcW + sx) * C + c] / 255.f;
        else if (dtype == RSA_F32)
          v = ld_as_float<float>(x, src);
        else if (dtype == RSA_F16)
          v = ld_as_float<_Float16>(x, src);
        else
          v = ld_as_float<__bf16>(x, src);
        if (mean != nullptr) v -= mean[c];  // per SOURCE channel
        v *= scale;
      }
      uint16_t hb, lb;
      split_elem(v, fmt, hb, lb);
      h[j] = hb;
      l[j] = lb;
    }
    const int64_t unit = (int64_t)n * batch_stride + (int64_t)pl * plane_stride + pix;
    ((u16x8*)out_hi)[unit] = h;
    if (out_lo != nullptr) ((u16x8*)out_lo)[unit] = l;
  }
}

// uint8 HWC image <-> float NCHW tensor: one thread per pixel (all channels), so image bytes move as C-byte runs and every
// tensor plane is read / written with consecutive lanes on consecutive pixels
__global__ void image_u8_to_nchw_kernel(const uint8_t* img, int batch, int H, int W, int C, void* out, int dtype) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)batch * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / HW, pix = idx - n * HW;
    for (int c = 0; c < C; ++c) {
      const float v = (float)img[idx * C + c] / 255.f;  // a true division: bit-identical to torch's img.float() / 255
      const int64_t o = (n * C + c) * HW + pix;
      if (dtype == RSA_F32)
        ((float*)out)[o] = v;
      else if (dtype == RSA_F16)
        ((_Float16*)out)[o] = (_Float16)v;
      else
        ((__bf16*)out)[o] = (__bf16)v;
    }
  }
}

__global__ void nchw_to_image_u8_kernel(const void* x, int dtype, int batch, int C, int H, int W, uint8_t* img) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)batch * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = idx / HW, pix = idx - n * HW;
    for (int c = 0; c < C; ++c) {
      const int64_t i = (n * C + c) * HW + pix;
      float v = dtype == RSA_F32 ? ((const float*)x)[i] : dtype == RSA_F16 ? (float)((const _Float16*)x)[i] : (float)((const __bf16*)x)[i];
      v = fminf(fmaxf(v, 0.f), 1.f);  // NaN -> 0, like clamp followed by an integer cast of 0
      img[idx * C + c] = (uint8_t)rintf(v * 255.f);
    }
  }
}


__global__ void planes_to_nchw_kernel(const void* hi, const void* lo, int64_t plane_stride, int64_t batch_stride, int batch, int C, int H,
                                      int W, int fmt, float* out) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)batch * C * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = idx % HW;
    const int64_t t = idx / HW;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const int64_t e = ((int64_t)n * batch_stride + (int64_t)(c >> 3) * plane_stride + pix) * 8 + (c & 7);
    float v = fmt == RSA_PF_F16 ? (float)((const _Float16*)hi)[e] : (float)((const __bf16*)hi)[e];
    if (lo != nullptr) v += fmt == RSA_PF_F16 ? (float)((const _Float16*)lo)[e] : (float)((const __bf16*)lo)[e];
    out[idx] = v;
  }
}

// rsa_check_finite: one pass over a plain array; a block that sees a NaN or an infinity adds 1 to the host-visible range word
template <typename T>
__global__ void check_finite_kernel(const T* x, int64_t n, unsigned int* word) {
  // 16 bytes per lane and iteration; the exponent field of every element is tested on the raw bits (all ones = infinity or NaN)
  constexpr int EPV = 16 / sizeof(T);
  constexpr uint32_t EXP = sizeof(T) == 4 ? 0x7f800000u : (std::is_same<T, _Float16>::value ? 0x7c00u : 0x7f80u);
  const int64_t nv = n / EPV;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = ((const uint4*)x)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (sizeof(T) == 4) {
        bad |= (w[k] & EXP) == EXP;
      } else {
        bad |= (w[k] & EXP) == EXP;
        bad |= ((w[k] >> 16) & EXP) == EXP;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - nv * EPV)) {  // tail elements
    const T t = x[nv * EPV + threadIdx.x];
    uint32_t b = 0;
    if (sizeof(T) == 4)
      b = __builtin_bit_cast(uint32_t, *(const float*)&t);
    else
      b = *(const uint16_t*)&t;
    bad |= (b & EXP) == EXP;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0 && word != nullptr) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static unsigned grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace rsa

extern "C" {

int rsa_version(void) { return RSA_VERSION; }

const char* rsa_last_error_string(void) { return rsa::g_err; }

int rsa_conv2d(const rsa_conv_params* p, void* stream) {
  if (p == nullptr) return rsa::set_error(RSA_E_ARG, "rsa_conv2d: null params");
  return rsa::conv_launch(*p, (hipStream_t)stream);
}

static int list_error(int rc, int i) {
  char buf[200];
  snprintf(buf, sizeof(buf), "rsa_conv2d_list: entry %d: %.150s", i, rsa::g_err);
  strncpy(rsa::g_err, buf, sizeof(rsa::g_err) - 1);  // keep the hip error text that conv_launch recorded
  return rc;
}

int rsa_conv2d_list(const rsa_conv_params* list, int32_t n, void* stream) {
  if (list == nullptr || n < 0) return rsa::set_error(RSA_E_ARG, "rsa_conv2d_list: null list");
  const bool fuse = rsa::conv_pair_enabled();
  for (int32_t i = 0; i < n; ++i) {
    // cross-layer fusion: two consecutive growth convolutions of a residual dense block run as ONE launch (conv_ring_pair.h)
    if (fuse && i + 1 < n && rsa::conv_pair_eligible(list[i], list[i + 1])) {
      int rc = rsa::conv_validate(list[i]);
      if (rc != RSA_OK) return list_error(rc, i);
      rc = rsa::conv_validate(list[i + 1]);
      if (rc != RSA_OK) return list_error(rc, i + 1);
      rc = rsa::conv_launch_pair(list[i], list[i + 1], (hipStream_t)stream);
      if (rc != 0) return list_error(rsa::set_error(rc, "conv: pair kernel launch failed"), i);
      ++i;
      continue;
    }
    const int rc = rsa::conv_launch(list[i], (hipStream_t)stream);
    if (rc != RSA_OK) return list_error(rc, i);
  }
  return RSA_OK;
}

int rsa_conv2d_pair(const rsa_conv_params* a, const rsa_conv_params* b, void* stream) {
  if (a == nullptr || b == nullptr) return rsa::set_error(RSA_E_ARG, "rsa_conv2d_pair: null params");
  if (!rsa::conv_pair_eligible(*a, *b)) return rsa::set_error(RSA_E_UNSUPPORTED, "rsa_conv2d_pair: the two descriptors are not a fusable pair (ask rsa_conv_pair_fusable)");
  int rc = rsa::conv_validate(*a);
  if (rc != RSA_OK) return rc;
  rc = rsa::conv_validate(*b);
  if (rc != RSA_OK) return rc;
  rc = rsa::conv_launch_pair(*a, *b, (hipStream_t)stream);
  return rc ? rsa::set_error(rc, "conv: pair kernel launch failed") : RSA_OK;
}

int rsa_conv_pair_fusable(const rsa_conv_params* a, const rsa_conv_params* b) {
  return a != nullptr && b != nullptr && rsa::conv_pair_enabled() && rsa::conv_pair_eligible(*a, *b) ? 1 : 0;
}

int rsa_debug_set_pair(int32_t mode) {
  rsa::conv_pair_override(mode);
  return RSA_OK;
}

int rsa_check_finite(const void* data, int32_t dtype, int64_t count, void* stream) {
  if (data == nullptr || count < 1) return rsa::set_error(RSA_E_ARG, "check_finite: bad argument");
  if ((uintptr_t)data & 15) return rsa::set_error(RSA_E_ALIGN, "check_finite: data must be 16-byte aligned");
  unsigned int* word = rsa::range_word();
  if (word == nullptr) return rsa::set_error(RSA_E_INTERNAL, "check_finite: no host-visible status word (pinned allocation failed)");
  const int64_t vecs = count / (dtype == RSA_F32 ? 4 : 8);
  const unsigned grid = rsa::grid_for(vecs > 0 ? vecs : 1, 256 * 4);
  if (dtype == RSA_F32)
    hipLaunchKernelGGL(rsa::check_finite_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)data, count, word);
  else if (dtype == RSA_F16)
    hipLaunchKernelGGL(rsa::check_finite_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)data, count, word);
  else if (dtype == RSA_BF16)
    hipLaunchKernelGGL(rsa::check_finite_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)data, count, word);
  else
    return rsa::set_error(RSA_E_ARG, "check_finite: dtype must be RSA_F32, RSA_F16 or RSA_BF16");
  const int rc = (int)hipGetLastError();
  return rc ? rsa::set_error(rc, "check_finite: launch failed") : RSA_OK;
}

int rsa_conv_cout_tiles(int32_t cout) { return rsa::conv_nct(cout); }

int rsa_conv_weight_layout(const rsa_conv_params* p) {
  if (p == nullptr) return rsa::set_error(RSA_E_ARG, "rsa_conv_weight_layout: null params");
  return rsa::conv_weight_layout(*p);
}

const char* rsa_conv_kernel_name(const rsa_conv_params* p) { return p == nullptr ? "" : rsa::conv_kernel_name(*p); }

int rsa_debug_ring_aborts(void) { return (int)rsa::conv_ring_aborts(); }

int rsa_check_status(void) { return rsa::conv_check_status(); }

int rsa_debug_set_ring_spin_limit(int32_t polls) {
  rsa::conv_set_ring_spin_limit(polls);
  return RSA_OK;
}

int rsa_debug_set_ring(int32_t mode) {
  rsa::conv_ring_override(mode);
  return RSA_OK;
}

int64_t rsa_packed_weight_bytes(int32_t cout, int32_t cin_planes, int32_t ksize, int32_t products) {
  if (cout < 1 || cin_planes < 1 || (ksize != 1 && ksize != 3) || (products != 1 && products != 3)) return RSA_E_ARG;
  const int64_t ct = (cout + 15) / 16;
  const int64_t chunks = (cin_planes + 3) / 4;
  const int64_t nhl = products == 3 ? 2 : 1;
  return chunks * ksize * ksize * ct * nhl * 64 * 16;
}

int rsa_nchw_to_planes(const void* x, int32_t dtype, int32_t batch, int32_t C, int32_t H, int32_t W, int32_t src_h, int32_t src_w,
                       int32_t unshuffle, const float* mean, float scale, void* out_hi, void* out_lo, int64_t out_plane_stride,
                       int64_t out_batch_stride, int32_t out_fmt, void* stream) {
  if (out_fmt != RSA_PF_BF16 && out_fmt != RSA_PF_F16) return rsa::set_error(RSA_E_ARG, "nchw_to_planes: out_fmt must be an rsa_plane_fmt");
  if (unshuffle < 1) return rsa::set_error(RSA_E_ARG, "nchw_to_planes: unshuffle must be >= 1");
  if (x == nullptr || out_hi == nullptr || batch < 1 || C < 1 || H < 1 || W < 1) return rsa::set_error(RSA_E_ARG, "nchw_to_planes: bad argument");
  {
    const int64_t fh = (int64_t)H * unshuffle, fw = (int64_t)W * unshuffle;  // full-resolution size the planes cover
    if (src_h < 1 || src_w < 1 || src_h > fh || src_w > fw || fh - src_h >= src_h || fw - src_w >= src_w)
      return rsa::set_error(RSA_E_ARG, "nchw_to_planes: source size must satisfy src <= plane size * unshuffle < 2*src (reflect padding)");
  }
  if (dtype < RSA_F32 || dtype > RSA_U8) return rsa::set_error(RSA_E_ARG, "nchw_to_planes: bad dtype");
  if (((uintptr_t)out_hi | (uintptr_t)out_lo) & 15) return rsa::set_error(RSA_E_ALIGN, "nchw_to_planes: outputs must be 16-byte aligned");
  const int64_t total = (int64_t)batch * ((C * unshuffle * unshuffle + 7) / 8) * H * W;
  hipLaunchKernelGGL(rsa::nchw_to_planes_kernel, dim3(rsa::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dtype, batch, C, H, W,
                     src_h, src_w, unshuffle, mean, scale, out_hi, out_lo, out_plane_stride, out_batch_stride, out_fmt);
  const int rc = (int)hipGetLastError();
  return rc ? rsa::set_error(rc, "nchw_to_planes: launch failed") : RSA_OK;
}

int rsa_planes_to_nchw(const void* hi, const void* lo, int64_t plane_stride, int64_t batch_stride, int32_t batch, int32_t C, int32_t H, int32_t W,
                       int32_t fmt, float* out, void* stream) {
  if (fmt != RSA_PF_BF16 && fmt != RSA_PF_F16) return rsa::set_error(RSA_E_ARG, "planes_to_nchw: fmt must be an rsa_plane_fmt");
  if (hi == nullptr || out == nullptr || batch < 1 || C < 1 || H < 1 || W < 1) return rsa::set_error(RSA_E_ARG, "planes_to_nchw: bad argument");
  const int64_t total = (int64_t)batch * C * H * W;
  hipLaunchKernelGGL(rsa::planes_to_nchw_kernel, dim3(rsa::grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, hi, lo, plane_stride,
                     batch_stride, batch, C, H, W, fmt, out);
  const int rc = (int)hipGetLastError();
  return rc ? rsa::set_error(rc, "planes_to_nchw: launch failed") : RSA_OK;
}

}  // extern "C"

int rsa_image_u8_to_nchw(const uint8_t* img, int32_t batch, int32_t H, int32_t W, int32_t C, void* out, int32_t dtype, void* stream) {
  if (img == nullptr || out == nullptr || batch < 1 || H < 1 || W < 1 || C < 1) return rsa::set_error(RSA_E_ARG, "image_u8_to_nchw: bad argument");
  if (dtype < RSA_F32 || dtype > RSA_BF16) return rsa::set_error(RSA_E_ARG, "image_u8_to_nchw: bad dtype");
  hipLaunchKernelGGL(rsa::image_u8_to_nchw_kernel, dim3(rsa::grid_for((int64_t)batch * H * W, 256)), dim3(256), 0, (hipStream_t)stream, img, batch, H,
                     W, C, out, dtype);
  const hipError_t rc = hipGetLastError();
  return rc ? rsa::set_error(rc, "image_u8_to_nchw: launch failed") : RSA_OK;
}

int rsa_nchw_to_image_u8(const void* x, int32_t dtype, int32_t batch, int32_t C, int32_t H, int32_t W, uint8_t* img, void* stream) {
  if (x == nullptr || img == nullptr || batch < 1 || H < 1 || W < 1 || C < 1) return rsa::set_error(RSA_E_ARG, "nchw_to_image_u8: bad argument");
  if (dtype < RSA_F32 || dtype > RSA_BF16) return rsa::set_error(RSA_E_ARG, "nchw_to_image_u8: bad dtype");
  hipLaunchKernelGGL(rsa::nchw_to_image_u8_kernel, dim3(rsa::grid_for((int64_t)batch * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, dtype, batch,
                     C, H, W, img);
  const hipError_t rc = hipGetLastError();
  return rc ? rsa::set_error(rc, "nchw_to_image_u8: launch failed") : RSA_OK;
}
