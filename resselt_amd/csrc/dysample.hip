// dysample.hip — DySample upsampler head (reference resselt/utilities/dysample.py:47-83) as ONE gather kernel.
//
// The reference materialises: offset/scope 1x1 convs -> sigmoid/mul/add -> meshgrid coords -> normalise ->
// pixel_shuffle -> permute -> grid_sample(bilinear, border, align_corners=False) over 4 channel groups ->
// 1x1 end_conv.  With align_corners=False the normalise/denormalise pair cancels exactly:
//     ix = w + off_x ,  iy = h + off_y        (pixel units of the LR map, then clamped to the border)
// so one thread per OUTPUT pixel reads its 2*groups offsets, gathers 4 neighbours x C channels from the f32
// feature map (NCHW4c, 16-byte loads) and applies the 1x1 end convolution in registers.
// The two 1x1 convs (offset with bias, scope without) are run by the conv engine as one k1 convolution whose
// f32 NCHW4c output holds offset channels [0, oc) and scope channels [oc, 2*oc), oc = 2*groups*scale^2.
// Pre-projected mode (end_w == NULL): the 1x1 end convolution is applied per channel group BEFORE the sampling, at low resolution
// (bilinear sampling is linear), so the gather touches 4 floats per group instead of C/groups channels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "resselt_amd.h"

namespace rsa {

typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int DYS_MAX_OUT = 8;

__global__ __launch_bounds__(256) void dysample_kernel(const rsa_dysample_params p) {
  const int s = p.scale, G = p.groups;
  const int oW = p.W * s, oH = p.H * s;
  const int64_t total = (int64_t)p.batch * oH * oW;
  const int64_t HW = (int64_t)p.H * p.W;
  const int oc = 2 * G * s * s;
  const int cpg4 = p.C / G / 4;  // float4 groups per channel group
  const int cp4 = p.C / 4;
  const int op4 = (2 * oc) / 4;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % oW);
    const int64_t t = idx / oW;
    const int oy = (int)(t % oH);
    const int n = (int)(t / oH);
    const int w = ox / s, j = ox - w * s;
    const int h = oy / s, i = oy - h * s;
    const int64_t pix = (int64_t)h * p.W + w;
    float out[DYS_MAX_OUT];
#pragma unroll
    for (int o = 0; o < DYS_MAX_OUT; ++o) out[o] = (p.end_b != nullptr && o < p.out_ch) ? p.end_b[o] : 0.f;
    const float* osc = p.offscope + (int64_t)n * op4 * HW * 4;
    const float* xb = p.x_f32 + (int64_t)n * cp4 * HW * 4;
    for (int g = 0; g < G; ++g) {
      float off[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ch = (c * G + g) * s * s + i * s + j;  // pixel_shuffle source channel (dysample.py:66-72)
        const float o = osc[((int64_t)(ch >> 2) * HW + pix) * 4 + (ch & 3)];
        const int chs = ch + oc;
        const float sc = osc[((int64_t)(chs >> 2) * HW + pix) * 4 + (chs & 3)];
        off[c] = o * (1.f / (1.f + expf(-sc))) * 0.5f + p.init_pos[ch];
      }
      // grid_sample, bilinear, padding_mode='border', align_corners=False
      float ix = (float)w + off[0];
      float iy = (float)h + off[1];
      ix = fminf(fmaxf(ix, 0.f), (float)(p.W - 1));
      iy = fminf(fmaxf(iy, 0.f), (float)(p.H - 1));
      const float fx = floorf(ix), fy = floorf(iy);
      const float ax = ix - fx, ay = iy - fy;
      const int x0 = (int)fx, y0 = (int)fy;
      const int x1 = min(x0 + 1, p.W - 1), y1 = min(y0 + 1, p.H - 1);
      const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
      const int64_t p00 = (int64_t)y0 * p.W + x0, p01 = (int64_t)y0 * p.W + x1, p10 = (int64_t)y1 * p.W + x0, p11 = (int64_t)y1 * p.W + x1;
      if (p.end_w == nullptr) {
        // pre-projected input: bilinear sampling is linear, so end_conv(sample_g(x)) = sample_g(end_conv restricted to group g);
        // x_f32 then holds z[g][0..3] = sum_{c in g} W_end[o][c] x_c (one f32x4 per group), a 1x1 convolution run at LOW resolution
        const f32x4* plane = (const f32x4*)(xb + (int64_t)g * HW * 4);
        const f32x4 v00 = plane[p00], v01 = plane[p01], v10 = plane[p10], v11 = plane[p11];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < p.out_ch) out[r] += v00[r] * w00 + v01[r] * w01 + v10[r] * w10 + v11[r] * w11;
        continue;
      }
      for (int q = 0; q < cpg4; ++q) {
        const f32x4* plane = (const f32x4*)(xb + (int64_t)(g * cpg4 + q) * HW * 4);
        const f32x4 v00 = plane[p00], v01 = plane[p01], v10 = plane[p10], v11 = plane[p11];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // same association order as ATen's CPU kernel: nw, ne, sw, se accumulated in that order
          const float v = v00[r] * w00 + v01[r] * w01 + v10[r] * w10 + v11[r] * w11;
          const int c = (g * cpg4 + q) * 4 + r;
#pragma unroll
          for (int o = 0; o < DYS_MAX_OUT; ++o)
            if (o < p.out_ch) out[o] += p.end_w[o * p.C + c] * v;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < DYS_MAX_OUT; ++o) {
      if (o >= p.out_ch) break;
      const int64_t di = (((int64_t)n * p.out_ch + o) * oH + oy) * oW + ox;
      if (p.out_dtype == RSA_F32)
        ((float*)p.out_nchw)[di] = out[o];
      else if (p.out_dtype == RSA_F16)
        ((_Float16*)p.out_nchw)[di] = (_Float16)out[o];
      else
        ((__bf16*)p.out_nchw)[di] = (__bf16)out[o];
    }
  }
}

}  // namespace rsa

extern "C" int rsa_dysample(const rsa_dysample_params* p, void* stream) {
  using namespace rsa;
  if (p == nullptr) return set_error(RSA_E_ARG, "dysample: null params");
  if (p->batch < 1 || p->H < 1 || p->W < 1 || p->scale < 1 || p->groups < 1 || p->C < 1) return set_error(RSA_E_ARG, "dysample: bad geometry");
  if (p->C % (4 * p->groups) != 0) return set_error(RSA_E_UNSUPPORTED, "dysample: C must be a multiple of 4*groups");
  if ((2 * p->groups * p->scale * p->scale) % 2 != 0) return set_error(RSA_E_UNSUPPORTED, "dysample: bad offset channel count");
  if (p->out_ch < 1 || p->out_ch > DYS_MAX_OUT) return set_error(RSA_E_UNSUPPORTED, "dysample: out_ch must be 1..8");
  if (!p->x_f32 || !p->offscope || !p->init_pos || !p->out_nchw) return set_error(RSA_E_ARG, "dysample: null pointer");
  if (!p->end_w && (p->C != 4 * p->groups || p->out_ch > 4)) return set_error(RSA_E_ARG, "dysample: pre-projected input needs C == 4*groups and out_ch <= 4");
  if (p->out_dtype < RSA_F32 || p->out_dtype > RSA_BF16) return set_error(RSA_E_ARG, "dysample: bad out_dtype");
  if (((uintptr_t)p->x_f32 | (uintptr_t)p->offscope) & 15) return set_error(RSA_E_ALIGN, "dysample: maps must be 16-byte aligned");
  const int64_t total = (int64_t)p->batch * p->H * p->scale * p->W * p->scale;
  int64_t g = (total + 255) / 256;
  if (g > 256 * 64) g = 256 * 64;
  hipLaunchKernelGGL(dysample_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *p);
  const int rc = (int)hipGetLastError();
  return rc ? set_error(rc, "dysample: launch failed") : RSA_OK;
}
