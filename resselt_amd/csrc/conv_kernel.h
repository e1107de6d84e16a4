// conv_kernel.h — the fused k1/k3 convolution kernel template (implicit GEMM on gfx950 matrix cores) and its launcher.
//
// Replaces the reference's per-layer ATen sequence cat -> [nearest x2] -> Conv2d -> act -> residual(s) -> [PixelShuffle]
// (utilities/block.py:148-200,340-344,454-465,510-537 of the reference); see include/resselt_amd.h for the contract and
// DESIGN.md §3/§4.1 for the layouts and the schedule.  Included by the conv_inst_*.hip translation units (one per
// (kernel size, products, upsample) family so that they compile in parallel) and dispatched from conv_mfma.hip.
//
// Mapping (one workgroup = 8 compute waves + 1 loader wave, one workgroup per CU, persistent over tiles):
//   output tile   : 16 rows x 32 pixels; compute wave = RPW rows x 2 halves of 16 pixels = NPT pixel-tiles x CTW cout-tiles of 16
//                   (8 x 1-2 for 1, 2 or 4 cout tiles per slab; 4 x 3 for 3: see GeoLW in conv_common.h)
//   MFMA          : v_mfma_f32_16x16x32_bf16,  D[cout 16][pixel 16] += A[cout][k 32] * B[k][pixel]
//                   A = weights (lane l: cout l&15, k-group l>>4), pre-packed in fragment order and streamed by each wave
//                       straight from L2 into VGPRs one tap ahead (waves own disjoint cout tiles or rows, so nothing is
//                       shared through LDS and no barrier guards the weights)
//                   B = activations: lane l reads ONE 16-byte unit = 8 channels of pixel (l&15) in plane (4q + (l>>4)) of
//                       the LDS halo tile, shifted by the tap (dy,dx)
//   K loop        : chunks q of 4 planes (32 channels) x taps t; the halo tile of a chunk is staged once in LDS and reused
//                   by all 9 taps; one raw s_barrier per chunk
//   LDS halo tile : double-buffered [hi|lo][plane 0..3][IH][IW] units, plane stride PS = 0 (mod 16 units) so that every
//                   ds_read_b128 lane group hits 16 distinct 16-byte slots for ANY tap offset (measured SQ_LDS_BANK_CONFLICT = 0)
//   global->LDS   : the loader wave's LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction); the per-lane source address
//                   does the halo gather, the nearest-x2 read and the zero padding (a zero page)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "conv_common.h"

#ifndef RSA_XCD_STRIPS
#define RSA_XCD_STRIPS 1
#endif

namespace rsa {

static __device__ uint4 g_zero_unit[4];  // source of zero-padding units for the loader's LDS-DMA (never written)

// FMT = enum rsa_plane_fmt of the input planes and weights (the matrix instruction); fp16 is instantiated for one product only
template <int KS, int NCT, int PROD, int UP, int OUTK, int FMT = 0>
__global__ __launch_bounds__((GeoLW<KS, NCT>::NTHR), (GeoLW<KS, NCT>::NCW == 8 ? 3 : 2)) void conv_kernel(const rsa_conv_params p) {
  using G = GeoLW<KS, NCT>;
  constexpr int TH = G::TH, TW = G::TW, HALO = G::HALO, IH = G::IH, IW = G::IW, PS = G::PS;
  constexpr int WPX = G::WPX, CTW = G::CTW, NCW = G::NCW, NPT = G::NPT, RPW = G::RPW;
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

  // Persistent tile order.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b -> XCD b % 8), and every XCD has its own
  // L2.  Horizontally adjacent tiles share 2 of the 6 128-byte lines of every halo row, so round r gives XCD x the CONSECUTIVE
  // tiles [r*NWG + x*NWG/8, r*NWG + (x+1)*NWG/8): neighbours then fill from the same L2 and the shared lines cross the fabric once.
  const int NWG = (int)gridDim.x;
#if RSA_XCD_STRIPS
  const int tile0 = (NWG % 8 == 0) ? ((int)blockIdx.x % 8) * (NWG / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
#else
  const int tile0 = (int)blockIdx.x;
#endif
  if (tile0 >= num_tiles) return;  // whole workgroup

  if (wave == NCW) {
    // =========================== LOADER WAVE ===========================
    // Streams item (tile, q) into buffer (k & 1) while the compute waves multiply item k-1 out of the other buffer.
    // It owns its own vmcnt stream, so a halo tile stays in flight for a whole chunk of MFMA work (the compute
    // waves' per-tap weight waits cannot drain it).  LDS destination of one DMA instruction = base + lane*16, which
    // is exactly 64 consecutive units of the tile image; the per-lane SOURCE address does the halo gather, the
    // nearest-x2 read and the zero padding (invalid lanes read a zero unit).
    const uint32_t plane_units = (uint32_t)p.in_plane_stride;
    int k = 0;
    for (int tile = tile0; tile < num_tiles; tile += NWG) {
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

  // ---- weights: every wave streams ITS OWN A fragments (cout tiles CTW*wct .. CTW*wct + CTW-1 of this slab) straight from the
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

  f32x4 acc[NPT][CTW];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // B-fragment unit of (pixel-tile 0, tap 0,0) for this lane
  const int bunit0 = lg * PS + (wpx * RPW) * IW + li;

  load_w(0);
  int k = 0;
  for (int tile = tile0; tile < num_tiles; tile += NWG) {
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
      constexpr int GP = (CTW >= 2) ? 1 : 2;  // pixel tiles per step (1 when the wave owns 2-3 cout tiles, else 2)
      constexpr int SPT = NPT / GP;          // steps per tap
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
              acc[pt][ct] = mfma16<FMT>(wf, bf, acc[pt][ct]);
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
      epilogue<NCT, CTW, NPT, OUTK>(p, acc, n, ty * TH, tx * TW, slab, wct, wpx, li, lg);
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    set_slab(slab + 1 < nslabs ? slab + 1 : 0);
   }
  }
}

template <int KS, int NCT, int PROD, int UP, int OUTK, int FMT = 0>
static int launch_one(const rsa_conv_params& p, hipStream_t stream) {
  using G = GeoLW<KS, NCT>;
  const int tiles_x = (p.W + G::TW - 1) / G::TW;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int ct_total = (p.cout + 15) / 16;
  const int slabs = (ct_total + NCT - 1) / NCT;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x7fffffff) return RSA_E_UNSUPPORTED;
  // persistent workgroups: as many as the chip keeps resident (queried once per instantiation), strided over the tiles
  static std::atomic<int> resident_cache{0};  // concurrent first calls compute the same value: idempotent, never torn
  int resident = resident_cache.load(std::memory_order_relaxed);
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_kernel<KS, NCT, PROD, UP, OUTK, FMT>, G::NTHR, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
    resident_cache.store(resident, std::memory_order_relaxed);
  }
  (void)slabs;  // cout slabs are looped inside the kernel
  int gx = resident;
  if (gx > num_tiles) gx = (int)num_tiles;
  dim3 grid((unsigned)gx, 1, 1);
  hipLaunchKernelGGL((conv_kernel<KS, NCT, PROD, UP, OUTK, FMT>), grid, dim3(G::NTHR), 0, stream, p);
  return (int)hipGetLastError();
}


template <int KS, int PROD, int UP, int OUTK, int FMT = 0>
static int launch_nct2(const rsa_conv_params& p, int nct, hipStream_t stream) {
  switch (nct) {
    case 1:
      return launch_one<KS, 1, PROD, UP, OUTK, FMT>(p, stream);
    case 2:
      return launch_one<KS, 2, PROD, UP, OUTK, FMT>(p, stream);
    case 3:
      return launch_one<KS, 3, PROD, UP, OUTK, FMT>(p, stream);
    default:
      return launch_one<KS, 4, PROD, UP, OUTK, FMT>(p, stream);
  }
}

template <int KS, int PROD, int UP, int FMT = 0>
static int launch_nct(const rsa_conv_params& p, int nct, hipStream_t stream) {
  return p.out_nchw != nullptr ? launch_nct2<KS, PROD, UP, 1, FMT>(p, nct, stream) : launch_nct2<KS, PROD, UP, 0, FMT>(p, nct, stream);
}

}  // namespace rsa
