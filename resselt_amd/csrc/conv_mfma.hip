// conv_mfma.hip — fused k1/k3 convolution as an implicit GEMM on gfx950 matrix cores.
//
// Replaces the reference's per-layer ATen sequence cat -> [nearest x2] -> Conv2d -> act ->
// residual(s) -> [PixelShuffle] (utilities/block.py:148-200,340-344,454-465,510-537 of the
// reference); see include/resselt_amd.h for the contract and DESIGN.md §3 for the layout.
//
// Mapping (one workgroup = 256 threads = 4 waves, two workgroups per CU, persistent over tiles):
//   output tile   : 32 pixels wide; 8 rows (cout slab of 3-4 tiles of 16) or 16 rows (1-2 tiles)
//   wave          : 8 pixel-tiles (4 rows x 2 x 16 pixels) x 2 cout-tiles of 16 channels = 16 accumulator tiles
//   MFMA          : v_mfma_f32_16x16x32_bf16,  D[cout 16][pixel 16] += A[cout][k 32] * B[k][pixel]
//                   A = weights (lane l: cout l&15, k-group l>>4), pre-packed in fragment order and streamed by each
//                       wave straight from L2 into VGPRs one tap ahead (waves own disjoint cout tiles or rows, so
//                       nothing is shared through LDS and no barrier guards the weights)
//                   B = activations: lane l reads ONE 16-byte unit = 8 channels of pixel (l&15) in
//                       plane (4q + (l>>4)) of the LDS halo tile, shifted by the tap (dy,dx)
//   K loop        : chunks q of 4 planes (32 channels) x taps t; the halo tile of a chunk is staged
//                   once in LDS (two barriers per chunk) and reused by all 9 taps
//   LDS halo tile : [hi|lo][plane 0..3][IH][IW] units, plane stride PS = 0 (mod 16 units) so that every
//                   ds_read_b128 lane group (8 lanes of plane p + 8 lanes of plane p+1, pixel offsets
//                   covering 0..15 once) hits 16 distinct 16-byte slots for ANY tap offset (measured:
//                   SQ_LDS_BANK_CONFLICT = 0)
//   global->LDS   : register-staged (issue-early / write-late) raw buffer loads: item (tile, q)+1 is fetched into
//                   VGPRs while the 9 taps of item (tile, q) run on the matrix cores; zero padding and missing
//                   planes come from the buffer range check (no branches, loads stay in flight)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_common.h"

namespace rsa {

__device__ uint4 g_zero_unit[4];  // source of zero-padding units for the loader's LDS-DMA (never written)

template <int KS, int NCT, int PROD, int UP, int OUTK>
__global__ __launch_bounds__((GeoLW<KS, NCT>::NTHR), (GeoLW<KS, NCT>::NCW == 8 ? 3 : 2)) void conv_kernel(const rsa_conv_params p) {
  using G = GeoLW<KS, NCT>;
  constexpr int TH = G::TH, TW = G::TW, HALO = G::HALO, IH = G::IH, IW = G::IW, PS = G::PS;
  constexpr int WPX = G::WPX, CTW = G::CTW, NCW = G::NCW;
  constexpr int ACT_UNITS = NPL * PS;
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int DMA_IT = (ACT_UNITS + 63) / 64;  // LDS-DMA instructions (1 KiB each) per precision per chunk
  constexpr int T = KS * KS;

  // double-buffered halo tile: [buffer][hi|lo][plane 0..3][IH][IW] 16-byte units
  __shared__ uint4 s_act[2][NHL * ACT_UNITS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_x = (p.W + TW - 1) / TW;
  const int tiles_y = (p.H + TH - 1) / TH;
  const int tiles_img = tiles_x * tiles_y;
  const int num_tiles = tiles_img * p.batch;
  const int nslabs = (((p.cout + 15) >> 4) + NCT - 1) / NCT;  // cout slabs of one tile run back to back (halo tile re-read from L2)
  const int inW = UP ? (p.W >> 1) : p.W;
  const int nchunks = (p.cin_planes + NPL - 1) / NPL;
  const int nsteps = nchunks * T;
  const int ct_total = (p.cout + 15) >> 4;

  if ((int)blockIdx.x >= num_tiles) return;  // whole workgroup

  if (wave == NCW) {
    // =========================== LOADER WAVE ===========================
    // Streams item (tile, q) into buffer (k & 1) while the compute waves multiply item k-1 out of the other buffer.
    // It owns its own vmcnt stream, so a halo tile stays in flight for a whole chunk of MFMA work (the compute
    // waves' per-tap weight waits cannot drain it).  LDS destination of one DMA instruction = base + lane*16, which
    // is exactly 64 consecutive units of the tile image; the per-lane SOURCE address does the halo gather, the
    // nearest-x2 read and the zero padding (invalid lanes read a zero unit).
    const uint32_t plane_units = (uint32_t)p.in_plane_stride;
    int k = 0;
    for (int tile = blockIdx.x; tile < num_tiles; tile += (int)gridDim.x) {
      const int n = tile / tiles_img;
      const int tr = tile - n * tiles_img;
      const int ty = tr / tiles_x;
      const int tx = tr - ty * tiles_x;
      const int y0 = ty * TH - HALO, x0 = tx * TW - HALO;
      const uint4* img_hi = (const uint4*)p.in_hi + (int64_t)n * p.in_batch_stride;
      const uint4* img_lo = (PROD == 3) ? (const uint4*)p.in_lo + (int64_t)n * p.in_batch_stride : nullptr;
      // per-lane source map of this tile, identical for all its chunks: unit offset from the chunk's first plane,
      // 0xFFFFFFFF = zero padding; bits 30-31 of a valid entry would overflow only for > 2^30-unit planes (rejected on the host)
      uint32_t doff[DMA_IT];
#pragma unroll
      for (int it = 0; it < DMA_IT; ++it) {
        const int u = it * 64 + lane;
        const int pl = u / PS;
        const int r = u - pl * PS;
        const int py = r / IW;
        const int px = r - py * IW;
        int iy = y0 + py, ix = x0 + px;
        const bool ok = (r < IH * IW) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        if (UP) {
          iy >>= 1;
          ix >>= 1;
        }
        doff[it] = ok ? (uint32_t)pl * plane_units + (uint32_t)iy * (uint32_t)inW + (uint32_t)ix : 0xFFFFFFFFu;
      }
      for (int sq = 0; sq < nslabs * nchunks; ++sq, ++k) {
        const int q = sq % nchunks;
#ifdef RSA_ABL_NOBAR
        continue;
#endif
#ifdef RSA_ABL_NODMA
        if (k > 1) { wg_barrier(); continue; }
#endif
        const int planes_left = p.cin_planes - q * NPL;
        const uint4* ch = img_hi + (int64_t)q * NPL * p.in_plane_stride;
        const uint4* cl = (PROD == 3) ? img_lo + (int64_t)q * NPL * p.in_plane_stride : nullptr;
        uint4* dst = &s_act[k & 1][0];
#pragma unroll
        for (int it = 0; it < DMA_IT; ++it) {
          const int pl = (it * 64 + lane) / PS;
          const bool ok = doff[it] != 0xFFFFFFFFu && pl < planes_left;
          const uint4* sh = ok ? ch + doff[it] : (const uint4*)&g_zero_unit[0];
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sh,
                                           (__attribute__((address_space(3))) void*)(dst + it * 64), 16, 0, 0);
          if (PROD == 3) {
            const uint4* sl = ok ? cl + doff[it] : (const uint4*)&g_zero_unit[0];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sl,
                                             (__attribute__((address_space(3))) void*)(dst + ACT_UNITS + it * 64), 16, 0, 0);
          }
        }
        dma_wait();    // this wave's DMA has landed ...
        wg_barrier();  // ... and meets the compute waves: buffer (k & 1) is ready
      }
    }
    return;
  }

  // =========================== COMPUTE WAVES ===========================
  const int wct = wave / WPX;        // which cout group
  const int wpx = wave - wct * WPX;  // which group of 4 rows
  const int li = lane & 15;
  const int lg = lane >> 4;

  // ---- weights: every wave streams ITS OWN A fragments (cout tiles 2*wct, 2*wct+1 of this slab) straight from the
  //      L2-resident packed blob into VGPRs, one tap ahead.  No LDS, no barrier: waves never share weight registers. ----
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w_packed, 0, (uint32_t)((int64_t)nsteps * ct_total * NHL * 64 * 16), 0x00020000);
  uint32_t woff[CTW];  // byte offset of this lane's fragment of (step 0, cout tile c, hi); 0xFFFFFFFF when the tile does not exist
  auto set_slab = [&](int slab) {
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int ctg = slab * NCT + wct * CTW + c;
      woff[c] = (wct * CTW + c < NCT && ctg < ct_total) ? (uint32_t)((ctg * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
    }
  };
  set_slab(0);
  const uint32_t wstep = (uint32_t)ct_total * NHL * 64 * 16;  // bytes per step
  bf16x8 wc[CTW][NHL];  // fragments of the tap being multiplied
  bf16x8 wn[CTW][NHL];  // fragments of the next tap, in flight
  auto load_w = [&](int s) {
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int hl = 0; hl < NHL; ++hl) {
        // a missing cout tile keeps voffset 0xFFFFFFFF -> out of range -> zeros from the range check
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)s * wstep + (uint32_t)hl * 1024u, 0);
        wn[c][hl] = __builtin_bit_cast(bf16x8, v);
      }
  };

  f32x4 acc[8][CTW];
#pragma unroll
  for (int pt = 0; pt < 8; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // B-fragment unit of (pixel-tile 0, tap 0,0) for this lane
  const int bunit0 = lg * PS + (wpx * 4) * IW + li;

  load_w(0);
  int k = 0;
  for (int tile = blockIdx.x; tile < num_tiles; tile += (int)gridDim.x) {
   for (int slab = 0; slab < nslabs; ++slab) {
    for (int q = 0; q < nchunks; ++q, ++k) {
#ifndef RSA_ABL_NOBAR
      wg_barrier();  // buffer (k & 1) has landed; everyone finished reading the other buffer one item ago (weight prefetches and
                     // the previous tile's stores stay in flight: no vmcnt drain here)
#endif
      const uint4* sa = &s_act[k & 1][0];
#ifdef RSA_ABL_NOMFMA
      if (k >= 0) continue;  // timing-only build: loader throughput with idle compute waves
#endif
      // software pipeline over the (tap, pixel-tile group) steps of the chunk: the B fragments of step i+LDS_DEPTH are
      // read from LDS while step i multiplies.  A step covers GP pixel tiles with GP*CTW == 2 accumulator tiles, and its
      // MFMAs are issued product-major, so two dependent MFMAs on one accumulator are never back to back.
      constexpr int GP = 2 / CTW;          // pixel tiles per step (1 when the wave owns 2 cout tiles, else 2)
      constexpr int SPT = 8 / GP;          // steps per tap
      constexpr int NSTEP = T * SPT;
      constexpr int LDS_DEPTH = 2;
      bf16x8 rh[LDS_DEPTH + 1][GP], rl[LDS_DEPTH + 1][GP];
      auto frag_unit = [&](int i, int g) -> int {
        const int t = i / SPT, pt = (i - t * SPT) * GP + g;
        const int dy = t / KS, dx = t - (t / KS) * KS;
        return bunit0 + ((pt >> 1) + dy) * IW + (pt & 1) * 16 + dx;
      };
#pragma unroll
      for (int i = 0; i < LDS_DEPTH && i < NSTEP; ++i)
#pragma unroll
        for (int g = 0; g < GP; ++g) {
          rh[i][g] = *(const bf16x8*)&sa[frag_unit(i, g)];
          if (PROD == 3) rl[i][g] = *(const bf16x8*)&sa[ACT_UNITS + frag_unit(i, g)];
        }
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        const int t = i / SPT, sp = i - t * SPT;
        if (sp == 0) {
          const int s = q * T + t;
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int hl = 0; hl < NHL; ++hl) wc[c][hl] = wn[c][hl];
#ifndef RSA_ABL_NOW
          if (s + 1 < nsteps) {
            load_w(s + 1);  // next tap's weights
          } else {
            set_slab(slab + 1 < nslabs ? slab + 1 : 0);  // last step: prefetch step 0 of the next slab / next tile
            load_w(0);
            set_slab(slab);
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef RSA_ABL_NOLDS
        if (false) {
#else
        if (i + LDS_DEPTH < NSTEP) {
#endif
#pragma unroll
          for (int g = 0; g < GP; ++g) {
            const int u = frag_unit(i + LDS_DEPTH, g);
            rh[(i + LDS_DEPTH) % (LDS_DEPTH + 1)][g] = *(const bf16x8*)&sa[u];
            if (PROD == 3) rl[(i + LDS_DEPTH) % (LDS_DEPTH + 1)][g] = *(const bf16x8*)&sa[ACT_UNITS + u];
          }
        }
        constexpr int NPR = (PROD == 3) ? 3 : 1;
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
          for (int g = 0; g < GP; ++g)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
              const int pt = sp * GP + g;
              // products in increasing magnitude: w_lo*a_hi, w_hi*a_lo, w_hi*a_hi
              const bf16x8 wf = (PROD == 3 && pr == 0) ? wc[ct][NHL - 1] : wc[ct][0];
              const bf16x8 bf = (PROD == 3 && pr == 1) ? rl[i % (LDS_DEPTH + 1)][g] : rh[i % (LDS_DEPTH + 1)][g];
              acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bf, acc[pt][ct], 0, 0, 0);
            }
        // issue order inside the step: the prefetch reads first, then this step's MFMAs
        if (i + LDS_DEPTH < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NHL * GP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NPR * GP * CTW, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- tile finished: epilogue (the loader is already streaming the next tile) ----
    {
      const int n = tile / tiles_img;
      const int tr = tile - n * tiles_img;
      const int ty = tr / tiles_x;
      const int tx = tr - ty * tiles_x;
      epilogue<NCT, CTW, OUTK>(p, acc, n, ty * TH, tx * TW, slab, wct, wpx, li, lg);
    }
#pragma unroll
    for (int pt = 0; pt < 8; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    set_slab(slab + 1 < nslabs ? slab + 1 : 0);
   }
  }
}

template <int KS, int NCT, int PROD, int UP, int OUTK>
static int launch_one(const rsa_conv_params& p, hipStream_t stream) {
  using G = GeoLW<KS, NCT>;
  const int tiles_x = (p.W + G::TW - 1) / G::TW;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int ct_total = (p.cout + 15) / 16;
  const int slabs = (ct_total + NCT - 1) / NCT;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x7fffffff) return RSA_E_UNSUPPORTED;
  // persistent workgroups: as many as the chip keeps resident (queried once per instantiation), strided over the tiles
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_kernel<KS, NCT, PROD, UP, OUTK>, G::NTHR, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
  }
  (void)slabs;  // cout slabs are looped inside the kernel
  int gx = resident;
  if (gx > num_tiles) gx = (int)num_tiles;
  dim3 grid((unsigned)gx, 1, 1);
  hipLaunchKernelGGL((conv_kernel<KS, NCT, PROD, UP, OUTK>), grid, dim3(G::NTHR), 0, stream, p);
  return (int)hipGetLastError();
}

int gemm_k1_launch(const rsa_conv_params& p, hipStream_t stream);  // gemm_k1.hip; -100 = not applicable

template <int KS, int PROD, int UP, int OUTK>
static int launch_nct2(const rsa_conv_params& p, int nct, hipStream_t stream) {
  switch (nct) {
    case 1:
      return launch_one<KS, 1, PROD, UP, OUTK>(p, stream);
    case 2:
      return launch_one<KS, 2, PROD, UP, OUTK>(p, stream);
    case 3:
      return launch_one<KS, 3, PROD, UP, OUTK>(p, stream);
    default:
      return launch_one<KS, 4, PROD, UP, OUTK>(p, stream);
  }
}

template <int KS, int PROD, int UP>
static int launch_nct(const rsa_conv_params& p, int nct, hipStream_t stream) {
  return p.out_nchw != nullptr ? launch_nct2<KS, PROD, UP, 1>(p, nct, stream) : launch_nct2<KS, PROD, UP, 0>(p, nct, stream);
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
  if (p.act < 0 || p.act > RSA_ACT_PRELU) return set_error(RSA_E_ARG, "conv: bad act");
  if (p.act == RSA_ACT_PRELU && (p.act_vec == nullptr || ((uintptr_t)p.act_vec & 15))) return set_error(RSA_E_ARG, "conv: PReLU needs 16-byte aligned act_vec");
  if (p.out_base != nullptr && p.out_nchw == nullptr) return set_error(RSA_E_ARG, "conv: out_base only applies to the out_nchw store");
  if (((uintptr_t)p.in_hi | (uintptr_t)p.in_lo | (uintptr_t)p.w_packed | (uintptr_t)p.out_hi | (uintptr_t)p.out_lo | (uintptr_t)p.out_f32 |
       (uintptr_t)p.res1 | (uintptr_t)p.res2) & 15)
    return set_error(RSA_E_ALIGN, "conv: pointers must be 16-byte aligned");
  if (p.in_plane_stride * 64 >= (int64_t)1 << 32) return set_error(RSA_E_UNSUPPORTED, "conv: input plane too large for a 4-plane buffer descriptor (>= 64 Mpx); band the image");
  if (p.out_plane_stride * 32 >= (int64_t)1 << 32 || (int64_t)p.H * p.W * 64 >= (int64_t)1 << 32)
    return set_error(RSA_E_UNSUPPORTED, "conv: output plane too large for 32-bit lane offsets; band the image");
  if ((uintptr_t)p.bias & 15) return set_error(RSA_E_ALIGN, "conv: bias must be 16-byte aligned (and padded to a multiple of 16 floats)");
  if (p.out_nchw != nullptr) {
    if (p.out_hi != nullptr || p.out_f32 != nullptr || p.res1 != nullptr || p.res2 != nullptr || p.act == RSA_ACT_SPAB_GATE)
      return set_error(RSA_E_UNSUPPORTED, "conv: out_nchw is a final store: no plane/f32 outputs or residuals with it");
    const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
    if (p.cout % (ps * ps) != 0) return set_error(RSA_E_ARG, "conv: cout not divisible by pixel_shuffle^2");
    if (p.out_dtype < RSA_F32 || p.out_dtype > RSA_BF16) return set_error(RSA_E_ARG, "conv: bad out_dtype");
  }
  if (p.ksize == 1) {  // wide k1 layers (nn.Linear over tokens): weight-stationary GEMM schedule, gemm_k1.hip
    const int g = gemm_k1_launch(p, stream);
    if (g != -100) return g == 0 ? RSA_OK : set_error(g, "conv: gemm_k1 launch failed");
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
