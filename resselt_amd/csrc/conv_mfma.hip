// conv_mfma.hip — fused k1/k3 convolution as an implicit GEMM on gfx950 matrix cores.
//
// Replaces the reference's per-layer ATen sequence cat -> [nearest x2] -> Conv2d -> act ->
// residual(s) -> [PixelShuffle] (utilities/block.py:148-200,340-344,454-465,510-537 of the
// reference); see include/resselt_amd.h for the contract and DESIGN.md §3 for the layout.
//
// Mapping (one workgroup = 256 threads = 4 waves, two workgroups per CU):
//   output tile   : TH x TW = 8 x 32 pixels, NCT cout-tiles of 16 channels
//   wave w        : rows 2w, 2w+1 of the tile = 4 pixel-tiles of 16 consecutive pixels
//   MFMA          : v_mfma_f32_16x16x32_bf16,  D[cout 16][pixel 16] += A[cout][k 32] * B[k][pixel]
//                   A = weights (lane l: cout l&15, k-group l>>4), pre-packed in fragment order
//                   B = activations: lane l reads ONE 16-byte unit = 8 channels of pixel (l&15) in
//                       plane (4q + (l>>4)) of the LDS halo tile, shifted by the tap (dy,dx)
//   K loop        : chunks q of 4 planes (32 channels) x taps t; the halo tile of a chunk is staged
//                   once in LDS and reused by all 9 taps; weights of a tap go through a 2-deep LDS ring
//   LDS halo tile : [hi|lo][plane 0..3][IH][IW] units, plane stride PS = 0 (mod 16 units) so that every
//                   ds_read_b128 lane group (8 lanes of plane p + 8 lanes of plane p+1, pixel offsets
//                   covering 0..15 once) hits 16 distinct 16-byte slots for ANY tap offset
//   global->LDS   : register-staged (issue-early / write-late): chunk q+1 is fetched into VGPRs while
//                   the 9 taps of chunk q run on the matrix cores
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "resselt_amd.h"
#include "common.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TH = 8;
constexpr int TW = 32;
constexpr int NPL = 4;  // planes per K chunk (32 channels = one MFMA K)
constexpr int NTHREADS = 256;

__device__ __forceinline__ float act_apply(float v, int act, float prm) {
  switch (act) {
    case RSA_ACT_LRELU:
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

__device__ __forceinline__ void split_store(void* hi_base, void* lo_base, int64_t unit, int sub, const float v[4]) {
  // 4 consecutive channels of one pixel -> 8 bytes in the hi plane, 8 bytes in the lo plane
  bf16x4 h, l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    __bf16 hb = (__bf16)v[r];
    h[r] = hb;
    l[r] = (__bf16)(v[r] - (float)hb);
  }
  char* ph = (char*)hi_base + unit * 16 + sub * 8;
  *(bf16x4*)ph = h;
  if (lo_base) {
    char* pl = (char*)lo_base + unit * 16 + sub * 8;
    *(bf16x4*)pl = l;
  }
}

template <int KS, int NCT, int PROD, int UP>
__global__ __launch_bounds__(NTHREADS, 2) void conv_kernel(const rsa_conv_params p) {
  constexpr int HALO = KS / 2;
  constexpr int IH = TH + 2 * HALO;
  constexpr int IW = TW + 2 * HALO;
  constexpr int PS = ((IH * IW + 15) / 16) * 16;  // plane stride in units, == 0 mod 16
  constexpr int ACT_UNITS = NPL * PS;
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int FILL_IT = (ACT_UNITS + NTHREADS - 1) / NTHREADS;
  constexpr int W_UNITS = NCT * NHL * 64;
  constexpr int W_IT = (W_UNITS + NTHREADS - 1) / NTHREADS;
  constexpr int T = KS * KS;

  __shared__ uint4 s_act[NHL * ACT_UNITS];
  __shared__ uint4 s_w[2 * W_UNITS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 15;
  const int lg = lane >> 4;

  const int tiles_x = (p.W + TW - 1) / TW;
  const int tile = blockIdx.x;
  const int ty = tile / tiles_x;
  const int tx = tile - ty * tiles_x;
  const int y0 = ty * TH;
  const int x0 = tx * TW;
  const int slab = blockIdx.y;
  const int n = blockIdx.z;

  const int inH = UP ? (p.H >> 1) : p.H;
  const int inW = UP ? (p.W >> 1) : p.W;
  (void)inH;

  // ---- per-thread halo-fill map (identical for every chunk) ----
  uint32_t foff[FILL_IT];
  uint32_t fmeta = 0;  // per iteration: bit (4*it+2) = in-image, bits (4*it..4*it+1) = plane in chunk
#pragma unroll
  for (int it = 0; it < FILL_IT; ++it) {
    const int u = it * NTHREADS + tid;
    const int pl = u / PS;
    const int r = u - pl * PS;
    const int py = r / IW;
    const int px = r - py * IW;
    int iy = y0 - HALO + py;
    int ix = x0 - HALO + px;
    const bool ok = (u < ACT_UNITS) && (r < IH * IW) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    if (UP) {
      iy >>= 1;
      ix >>= 1;
    }
    foff[it] = ok ? (uint32_t)(pl * (uint32_t)p.in_plane_stride + (uint32_t)iy * (uint32_t)inW + (uint32_t)ix) : 0u;
    fmeta |= ((uint32_t)(pl & 3) | (ok ? 4u : 0u)) << (4 * it);
  }

  const int nchunks = (p.cin_planes + NPL - 1) / NPL;
  const int nsteps = nchunks * T;
  const int ct_total = (p.cout + 15) >> 4;

  const uint4* g_hi = (const uint4*)p.in_hi + (int64_t)n * p.in_batch_stride;
  const uint4* g_lo = (PROD == 3) ? ((const uint4*)p.in_lo + (int64_t)n * p.in_batch_stride) : nullptr;
  const uint4* g_w = (const uint4*)p.w_packed;

  uint4 st_hi[FILL_IT];
  uint4 st_lo[(PROD == 3) ? FILL_IT : 1];
  uint4 st_w[W_IT];

  auto load_act = [&](int q) {
    const int planes_left = p.cin_planes - q * NPL;  // >= 1
    const uint4* bh = g_hi + (int64_t)q * NPL * p.in_plane_stride;
    const uint4* bl = (PROD == 3) ? (g_lo + (int64_t)q * NPL * p.in_plane_stride) : nullptr;
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const uint32_t m = fmeta >> (4 * it);
      const bool ok = (m & 4u) && ((int)(m & 3u) < planes_left);
      const uint32_t off = ok ? foff[it] : 0u;
      uint4 vh = bh[off];
      if (!ok) vh = make_uint4(0, 0, 0, 0);
      st_hi[it] = vh;
      if (PROD == 3) {
        uint4 vl = bl[off];
        if (!ok) vl = make_uint4(0, 0, 0, 0);
        st_lo[it] = vl;
      }
    }
  };
  auto store_act = [&]() {
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const int u = it * NTHREADS + tid;
      if (u < ACT_UNITS) {
        s_act[u] = st_hi[it];
        if (PROD == 3) s_act[ACT_UNITS + u] = st_lo[it];
      }
    }
  };
  // weights of step s for this slab: units [(s*ct_total + slab*NCT) * NHL*64, +W_UNITS), tiles past ct_total are zero
  auto load_w = [&](int s) {
    const int64_t base = ((int64_t)s * ct_total + (int64_t)slab * NCT) * (NHL * 64);
    const int valid_units = (ct_total - slab * NCT) * (NHL * 64);  // may exceed W_UNITS
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = it * NTHREADS + tid;
      const bool ok = (u < W_UNITS) && (u < valid_units);
      uint4 v = g_w[base + (ok ? u : 0)];
      if (!ok) v = make_uint4(0, 0, 0, 0);
      st_w[it] = v;
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = it * NTHREADS + tid;
      if (u < W_UNITS) s_w[buf * W_UNITS + u] = st_w[it];
    }
  };

  f32x4 acc[4][NCT];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // B-fragment unit of (pixel-tile pt, tap 0,0) for this lane
  int bunit[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) bunit[pt] = lg * PS + (wave * 2 + (pt >> 1)) * IW + (pt & 1) * 16 + li;

  load_act(0);
  load_w(0);
  int cur = 1;
  for (int q = 0; q < nchunks; ++q) {
    __syncthreads();  // every wave is done reading the previous chunk's tile and weights
    store_act();
    cur ^= 1;
    store_w(cur);
    __syncthreads();
    if (q + 1 < nchunks) load_act(q + 1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int s = q * T + t;
      if (s + 1 < nsteps) load_w(s + 1);
      const int dy = t / KS;
      const int dx = t - dy * KS;
      const bf16x8* wf = (const bf16x8*)&s_w[cur * W_UNITS];
      bf16x8 wa[NCT][NHL];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) wa[ct][hl] = wf[(ct * NHL + hl) * 64 + lane];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const int u = bunit[pt] + dy * IW + dx;
        const bf16x8 bh = *(const bf16x8*)&s_act[u];
        if (PROD == 3) {
          const bf16x8 bl = *(const bf16x8*)&s_act[ACT_UNITS + u];
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][NHL - 1], bh, acc[pt][ct], 0, 0, 0);
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][0], bl, acc[pt][ct], 0, 0, 0);
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][0], bh, acc[pt][ct], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][0], bh, acc[pt][ct], 0, 0, 0);
        }
      }
      if (t + 1 < T) {
        store_w(cur ^ 1);
        __syncthreads();
        cur ^= 1;
      }
    }
  }

  // ---------------- epilogue ----------------
  const int64_t HW = (int64_t)p.H * p.W;
  const int p4 = (p.cout + 3) >> 2;
  const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int y = y0 + wave * 2 + (pt >> 1);
    const int x = x0 + (pt & 1) * 16 + li;
    if (y >= p.H || x >= p.W) continue;
    const int64_t pix = (int64_t)y * p.W + x;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int c0 = (slab * NCT + ct) * 16 + lg * 4;
      if (c0 >= ((p.cout + 7) & ~7)) continue;  // no plane / no f32 group holds these channels
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = c0 + r;
        float b = (p.bias != nullptr && c < p.cout) ? p.bias[c] : 0.f;
        v[r] = acc[pt][ct][r] + b;
      }
      const bool has_f32grp = c0 < (p4 << 2);
      const int64_t f32idx = (((int64_t)n * p4 + (c0 >> 2)) * HW + pix);
      if (p.act == RSA_ACT_SPAB_GATE) {
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        if (has_f32grp) rr = ((const f32x4*)p.res1)[f32idx];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = 1.f / (1.f + expf(-v[r]));
          v[r] = (v[r] + rr[r]) * (sg - 0.5f);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act, p.act_param);
        if (p.res1 != nullptr && has_f32grp) {
          const f32x4 rr = ((const f32x4*)p.res1)[f32idx];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + rr[r];
        }
      }
      if (p.res2 != nullptr && has_f32grp) {
        const f32x4 rr = ((const f32x4*)p.res2)[f32idx];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = v[r] * p.beta + rr[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (c0 + r >= p.cout) v[r] = 0.f;

      if (p.out_hi != nullptr) {
        const int64_t unit = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (c0 >> 3)) * p.out_plane_stride + pix;
        split_store(p.out_hi, p.out_lo, unit, (c0 >> 2) & 1, v);
      }
      if (p.out_f32 != nullptr && has_f32grp) {
        ((f32x4*)p.out_f32)[f32idx] = (f32x4){v[0], v[1], v[2], v[3]};
      }
      if (p.out_nchw != nullptr) {
        const int oc_total = p.cout / (ps * ps);
        const int64_t oH = (int64_t)p.H * ps, oW = (int64_t)p.W * ps;
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
          const int64_t idx = (((int64_t)n * oc_total + oc) * oH + ((int64_t)y * ps + ii)) * oW + ((int64_t)x * ps + jj);
          if (p.out_dtype == RSA_F32)
            ((float*)p.out_nchw)[idx] = o;
          else if (p.out_dtype == RSA_F16)
            ((_Float16*)p.out_nchw)[idx] = (_Float16)o;
          else
            ((__bf16*)p.out_nchw)[idx] = (__bf16)o;
        }
      }
    }
  }
}

template <int KS, int NCT, int PROD, int UP>
static int launch_one(const rsa_conv_params& p, hipStream_t stream) {
  const int tiles_x = (p.W + TW - 1) / TW;
  const int tiles_y = (p.H + TH - 1) / TH;
  const int ct_total = (p.cout + 15) / 16;
  const int slabs = (ct_total + NCT - 1) / NCT;
  dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)slabs, (unsigned)p.batch);
  hipLaunchKernelGGL((conv_kernel<KS, NCT, PROD, UP>), grid, dim3(NTHREADS), 0, stream, p);
  return (int)hipGetLastError();
}

template <int KS, int PROD, int UP>
static int launch_nct(const rsa_conv_params& p, int nct, hipStream_t stream) {
  switch (nct) {
    case 1:
      return launch_one<KS, 1, PROD, UP>(p, stream);
    case 2:
      return launch_one<KS, 2, PROD, UP>(p, stream);
    case 3:
      return launch_one<KS, 3, PROD, UP>(p, stream);
    default:
      return launch_one<KS, 4, PROD, UP>(p, stream);
  }
}

int conv_nct(int cout) {
  const int ct = (cout + 15) / 16;
  if (ct <= 4) return ct;
  // more than one slab: pick the tile count (4 or 3) that wastes the fewest padded cout-tiles
  const int w4 = ((ct + 3) / 4) * 4 - ct;
  const int w3 = ((ct + 2) / 3) * 3 - ct;
  return (w3 < w4) ? 3 : 4;
}

int conv_launch(const rsa_conv_params& p, hipStream_t stream) {
  if (p.batch < 1 || p.H < 1 || p.W < 1 || p.cin_planes < 1 || p.cout < 1) return set_error(RSA_E_ARG, "conv: bad geometry");
  if (p.ksize != 1 && p.ksize != 3) return set_error(RSA_E_UNSUPPORTED, "conv: ksize must be 1 or 3");
  if (p.products != 1 && p.products != 3) return set_error(RSA_E_UNSUPPORTED, "conv: products must be 1 or 3");
  if (p.upsample2x && (p.ksize != 3 || (p.H & 1) || (p.W & 1))) return set_error(RSA_E_UNSUPPORTED, "conv: upsample2x needs k3 and even H, W");
  if (p.in_hi == nullptr || p.w_packed == nullptr) return set_error(RSA_E_ARG, "conv: null input/weights");
  if (p.products == 3 && p.in_lo == nullptr) return set_error(RSA_E_ARG, "conv: products=3 needs in_lo");
  if (p.act == RSA_ACT_SPAB_GATE && p.res1 == nullptr) return set_error(RSA_E_ARG, "conv: SPAB gate needs res1");
  if (p.act < 0 || p.act > RSA_ACT_SPAB_GATE) return set_error(RSA_E_ARG, "conv: bad act");
  if (((uintptr_t)p.in_hi | (uintptr_t)p.in_lo | (uintptr_t)p.w_packed | (uintptr_t)p.out_hi | (uintptr_t)p.out_lo | (uintptr_t)p.out_f32 |
       (uintptr_t)p.res1 | (uintptr_t)p.res2) & 15)
    return set_error(RSA_E_ALIGN, "conv: pointers must be 16-byte aligned");
  if (p.in_plane_stride * 4 >= (int64_t)1 << 32) return set_error(RSA_E_UNSUPPORTED, "conv: input plane too large for 32-bit unit offsets; band the image");
  if (p.out_nchw != nullptr) {
    const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
    if (p.cout % (ps * ps) != 0) return set_error(RSA_E_ARG, "conv: cout not divisible by pixel_shuffle^2");
    if (p.out_dtype < RSA_F32 || p.out_dtype > RSA_BF16) return set_error(RSA_E_ARG, "conv: bad out_dtype");
  }
  const int nct = conv_nct(p.cout);
  int rc;
  if (p.ksize == 3) {
    if (p.upsample2x)
      rc = (p.products == 3) ? launch_nct<3, 3, 1>(p, nct, stream) : launch_nct<3, 1, 1>(p, nct, stream);
    else
      rc = (p.products == 3) ? launch_nct<3, 3, 0>(p, nct, stream) : launch_nct<3, 1, 0>(p, nct, stream);
  } else {
    rc = (p.products == 3) ? launch_nct<1, 3, 0>(p, nct, stream) : launch_nct<1, 1, 0>(p, nct, stream);
  }
  if (rc != 0) return set_error(rc, "conv: kernel launch failed");
  return RSA_OK;
}

}  // namespace rsa
