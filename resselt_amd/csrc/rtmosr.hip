// rtmosr.hip — the non-convolution kernels of the RTMoSR path (reference resselt/archs/rtmosr/arch.py):
//   rsa_rmsnorm            RMSNorm over channels, f32 residual map -> split planes                  RMSNorm.forward :32-37
//   rsa_unshuffle_pool     PixelUnshuffle(2) (f32 map) and MaxPool2d(2) (planes) of one plane range   ParPixelUnshuffle :292-299
//   rsa_gated_shuffle_mul  mish(g) * cat(i, PixelShuffle(2)(c * gate)) on plane ranges                GatedCNNBlock.forward :331-336
// The depthwise 5x5 of OmniShift (:209-289, re-parameterised to one kernel by the host) is rsa_dwconv5x5 in dat.hip, the SE layer
// (:7-22) is rsa_channel_gate with ReLU / Hardsigmoid.  All of these are HBM-bound streaming kernels; the RepConv layers between them
// are single launches of the fused convolution.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void rt_unit_f32(const bf16x8* hi, const bf16x8* lo, int64_t u, float (&v)[8]) {
  const bf16x8 h = hi[u];
  if (lo != nullptr) {
    const bf16x8 l = lo[u];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
}

__device__ __forceinline__ void rt_store_unit(bf16x8* hi, bf16x8* lo, int64_t u, const float (&v)[8]) {
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

// thread = pixel; grid (ceil(HW/256), batch).  x / (||x||_2 * C^-1/2 + eps) * scale + offset
__global__ __launch_bounds__(256) void rmsnorm_kernel(const f32x4* x, int64_t HW, int C, float eps, const float* scale, const float* offset, bf16x8* out_hi,
                                                      bf16x8* out_lo, int64_t plane_stride, int64_t batch_stride) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  const int p4 = (C + 3) >> 2;
  const f32x4* xb = x + (int64_t)n * p4 * HW;
  float ss = 0.f;
  for (int g = 0; g < p4; ++g) {
    const f32x4 v = xb[(int64_t)g * HW + pix];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g * 4 + r < C) ss += v[r] * v[r];
  }
  const float inv = 1.f / (sqrtf(ss) * rsqrtf((float)C) + eps);
  const int planes = (C + 7) >> 3;
  for (int pl = 0; pl < planes; ++pl) {
    float o[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int g = pl * 2 + half;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (g < p4) v = xb[(int64_t)g * HW + pix];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = g * 4 + r;
        o[half * 4 + r] = c < C ? scale[c] * (v[r] * inv) + offset[c] : 0.f;
      }
    }
    rt_store_unit(out_hi + (int64_t)n * batch_stride + (int64_t)pl * plane_stride, out_lo ? out_lo + (int64_t)n * batch_stride + (int64_t)pl * plane_stride : nullptr,
                  pix, o);
  }
}

// thread = (half-resolution pixel, input plane); grid (ceil(hw/256), planes, batch).  Input planes are H x W, outputs H/2 x W/2:
//   pooled plane pl (8 channels): max over the 2x2 block;  unshuffled f32 map: channel 4c + 2i + j <- c at (2y+i, 2x+j), i.e. f32 group c
__global__ __launch_bounds__(256) void unshuffle_pool_kernel(const bf16x8* in_hi, const bf16x8* in_lo, int64_t in_plane_stride, int64_t in_batch_stride, int H,
                                                             int W, int planes, f32x4* pu, bf16x8* pool_hi, bf16x8* pool_lo, int64_t pool_plane_stride,
                                                             int64_t pool_batch_stride) {
  const int h2 = H >> 1, w2 = W >> 1;
  const int64_t hw = (int64_t)h2 * w2;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int pl = blockIdx.y, n = blockIdx.z;
  if (pix >= hw) return;
  const int y = (int)(pix / w2), x = (int)(pix - (int64_t)y * w2);
  const bf16x8* hi = in_hi + (int64_t)n * in_batch_stride + (int64_t)pl * in_plane_stride;
  const bf16x8* lo = in_lo ? in_lo + (int64_t)n * in_batch_stride + (int64_t)pl * in_plane_stride : nullptr;
  float v[4][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) rt_unit_f32(hi, lo, (int64_t)(2 * y + i) * W + (2 * x + j), v[i * 2 + j]);
  float mx[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    mx[c] = fmaxf(fmaxf(v[0][c], v[1][c]), fmaxf(v[2][c], v[3][c]));
    pu[((int64_t)n * planes * 8 + pl * 8 + c) * hw + pix] = (f32x4){v[0][c], v[1][c], v[2][c], v[3][c]};
  }
  rt_store_unit(pool_hi + (int64_t)n * pool_batch_stride + (int64_t)pl * pool_plane_stride,
                pool_lo ? pool_lo + (int64_t)n * pool_batch_stride + (int64_t)pl * pool_plane_stride : nullptr, pix, mx);
}

__device__ __forceinline__ float mish_f(float v) {
  if (v > 20.f) return v;
  const float e = expf(v);
  const float t = e * (e + 2.f);
  return v * (t / (t + 2.f));
}

// thread = (pixel, output plane); grid (ceil(HW/256), planes_out, batch).  out = mish(g) * cat(i, shuffle(c * gate)):
//   g, i : plane ranges of the fc1 output at H x W;  c : planes at H/2 x W/2 holding 4*dim channels (channel 4k + 2(y&1) + (x&1) of pixel
//   (y/2, x/2) is channel k of pixel (y, x));  gate: per-channel SE gate of c (may be NULL)
__global__ __launch_bounds__(256) void gated_shuffle_mul_kernel(const rsa_gated_shuffle_params p) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int pl = blockIdx.y, n = blockIdx.z;
  if (pix >= HW) return;
  const bf16x8* f_hi = (const bf16x8*)p.f_hi + (int64_t)n * p.f_batch_stride;
  const bf16x8* f_lo = p.f_lo ? (const bf16x8*)p.f_lo + (int64_t)n * p.f_batch_stride : nullptr;
  float g[8], m[8];
  rt_unit_f32(f_hi, f_lo, (int64_t)pl * p.f_plane_stride + pix, g);
  if (pl < p.i_planes) {
    rt_unit_f32(f_hi, f_lo, (int64_t)(p.g_planes + pl) * p.f_plane_stride + pix, m);
  } else {
    const int y = (int)(pix / p.W), x = (int)(pix - (int64_t)y * p.W);
    const int pos = 2 * (y & 1) + (x & 1);
    const int64_t hp = (int64_t)(y >> 1) * (p.W >> 1) + (x >> 1);
    const bf16x8* c_hi = (const bf16x8*)p.c_hi + (int64_t)n * p.c_batch_stride;
    const bf16x8* c_lo = p.c_lo ? (const bf16x8*)p.c_lo + (int64_t)n * p.c_batch_stride : nullptr;
    const int k0 = (pl - p.i_planes) * 8;  // first channel (after the shuffle) of this output plane
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // source channels 4*(k0 + kk) + pos, kk = 0..7  ->  plane (k0 + kk) / 2, element 4*((k0 + kk) & 1) + pos
      float u[8];
      rt_unit_f32(c_hi, c_lo, (int64_t)(k0 / 2 + q) * p.c_plane_stride + hp, u);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kk = 2 * q + e;
        const int ch = 4 * (k0 + kk) + pos;
        float gv = 1.f;
        if (p.gate != nullptr) gv = p.gate[(int64_t)n * p.gate_stride + ch];
        m[kk] = (pos == 0 ? u[4 * e] : pos == 1 ? u[4 * e + 1] : pos == 2 ? u[4 * e + 2] : u[4 * e + 3]) * gv;
      }
    }
  }
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = mish_f(g[j]) * m[j];
  rt_store_unit((bf16x8*)p.out_hi + (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride,
                p.out_lo ? (bf16x8*)p.out_lo + (int64_t)n * p.out_batch_stride + (int64_t)pl * p.out_plane_stride : nullptr, pix, o);
}

static bool rt_misaligned(const void* a) { return ((uintptr_t)a & 15) != 0; }

}  // namespace rsa

using namespace rsa;

extern "C" int rsa_rmsnorm(const float* x_f32, int32_t batch, int32_t H, int32_t W, int32_t C, float eps, const float* scale, const float* offset,
                           void* out_hi, void* out_lo, int64_t out_plane_stride, int64_t out_batch_stride, void* stream) {
  if (!x_f32 || !scale || !offset || !out_hi || batch < 1 || batch > 65535 || H < 1 || W < 1 || C < 1) return set_error(RSA_E_ARG, "rmsnorm: bad argument");
  if (rt_misaligned(x_f32) || rt_misaligned(out_hi) || rt_misaligned(out_lo)) return set_error(RSA_E_ALIGN, "rmsnorm: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x_f32, HW, C, eps,
                     scale, offset, (bf16x8*)out_hi, (bf16x8*)out_lo, out_plane_stride, out_batch_stride);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "rmsnorm: launch failed") : RSA_OK;
}

extern "C" int rsa_unshuffle_pool(const void* in_hi, const void* in_lo, int64_t in_plane_stride, int64_t in_batch_stride, int32_t batch, int32_t H, int32_t W,
                                  int32_t planes, float* unshuffled_f32, void* pool_hi, void* pool_lo, int64_t pool_plane_stride,
                                  int64_t pool_batch_stride, void* stream) {
  if (!in_hi || !unshuffled_f32 || !pool_hi || batch < 1 || batch > 65535 || H < 2 || W < 2 || (H & 1) || (W & 1) || planes < 1 || planes > 65535)
    return set_error(RSA_E_ARG, "unshuffle_pool: bad argument (H and W must be even)");
  if (rt_misaligned(in_hi) || rt_misaligned(in_lo) || rt_misaligned(unshuffled_f32) || rt_misaligned(pool_hi) || rt_misaligned(pool_lo))
    return set_error(RSA_E_ALIGN, "unshuffle_pool: maps must be 16-byte aligned");
  const int64_t hw = (int64_t)(H / 2) * (W / 2);
  hipLaunchKernelGGL(unshuffle_pool_kernel, dim3((unsigned)((hw + 255) / 256), (unsigned)planes, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8*)in_hi, (const bf16x8*)in_lo, in_plane_stride, in_batch_stride, H, W, planes, (f32x4*)unshuffled_f32, (bf16x8*)pool_hi,
                     (bf16x8*)pool_lo, pool_plane_stride, pool_batch_stride);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "unshuffle_pool: launch failed") : RSA_OK;
}

extern "C" int rsa_gated_shuffle_mul(const rsa_gated_shuffle_params* p, void* stream) {
  if (p == nullptr) return set_error(RSA_E_ARG, "gated_shuffle_mul: null params");
  if (p->batch < 1 || p->batch > 65535 || p->H < 2 || p->W < 2 || (p->H & 1) || (p->W & 1) || p->g_planes < 1 || p->i_planes < 0 || p->i_planes > p->g_planes)
    return set_error(RSA_E_ARG, "gated_shuffle_mul: bad geometry");
  if (!p->f_hi || !p->c_hi || !p->out_hi) return set_error(RSA_E_ARG, "gated_shuffle_mul: null pointer");
  if (rt_misaligned(p->f_hi) || rt_misaligned(p->f_lo) || rt_misaligned(p->c_hi) || rt_misaligned(p->c_lo) || rt_misaligned(p->out_hi) || rt_misaligned(p->out_lo))
    return set_error(RSA_E_ALIGN, "gated_shuffle_mul: maps must be 16-byte aligned");
  const int64_t HW = (int64_t)p->H * p->W;
  hipLaunchKernelGGL(gated_shuffle_mul_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)p->g_planes, (unsigned)p->batch), dim3(256), 0,
                     (hipStream_t)stream, *p);
  const hipError_t rc = hipGetLastError();
  return rc ? set_error(rc, "gated_shuffle_mul: launch failed") : RSA_OK;
}
