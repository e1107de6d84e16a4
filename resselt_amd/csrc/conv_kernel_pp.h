// conv_kernel_pp.h — alternating ("ping-pong") schedule of the fused 3x3 convolution for layers with at most two cout tiles
// (Cout <= 32: the four growth convolutions of every residual dense block, utilities/block.py:454-465 of the reference).
//
// Why: with all eight compute waves in lock step on ONE tile (conv_kernel.h) every wave reaches the epilogue at the same moment; the
// matrix pipes idle while the stores are issued, and the first weight loads of the next tile queue (vmcnt retires in order) behind
// those stores.  Here the eight compute waves form two groups of four (one wave per SIMD each: waves 0-3 and 4-7).  Group g owns
// every second tile of the workgroup and LDS buffer g, and the groups take turns, one workgroup barrier per phase:
//
//   phase h:  group (h & 1) multiplies one 32-channel chunk of its tile out of buffer h & 1
//             (16 accumulator tiles per wave: 4 rows x 2 halves x 2 cout tiles);
//             the other group stores the tile it finished one phase ago (if any), then waits;
//             the loader wave streams the other group's next chunk into buffer (h+1) & 1.
//
// so a group's epilogue, and the drain of its stores, lie under the other group's MFMA phase.  The loader protocol is the one of
// conv_kernel.h (item h into buffer h & 1 while item h-1 is multiplied); only the item order differs (the two tiles interleaved).
// Measured on the 1080p RRDBNet-23 frame: 222 -> 208 ms.  The same alternation for the four-cout-tile layers (the groups taking the
// two cout halves of one tile in turn) was built and is 2 % slower than conv_kernel: those layers are MFMA-bound and lose more from
// having one instead of two waves per SIMD in the multiply than they gain from the hidden epilogue (rejected list).
#pragma once
#include <stdlib.h>

#include "conv_kernel.h"

namespace rsa {

template <int KS, int PROD, int UP, int OUTK, int FMT = 0>
__global__ __launch_bounds__(576, 3) void conv_kernel_pp(const rsa_conv_params p) {
  using G = GeoLW<KS, 4>;  // per-wave shape: CTW = 2 cout tiles, NPT = 8 pixel tiles, 4 row groups
  constexpr int TH = G::TH, TW = G::TW, HALO = G::HALO, IH = G::IH, IW = G::IW, PS = G::PS;
  constexpr int CTW = 2, NPT = G::NPT, RPW = G::RPW;
  constexpr int ACT_UNITS = NPL * PS;
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int DMA_IT = (ACT_UNITS + 63) / 64;
  constexpr int T = KS * KS;

  __shared__ uint4 s_act[2][NHL * ACT_UNITS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_x = (p.W + TW - 1) / TW;
  const int tiles_y = (p.H + TH - 1) / TH;
  const int tiles_img = tiles_x * tiles_y;
  const int num_tiles = tiles_img * p.batch;
  const int inW = UP ? (p.W >> 1) : p.W;
  const int nchunks = (p.cin_planes + NPL - 1) / NPL;
  const int nsteps = nchunks * T;
  const int ct_total = (p.cout + 15) >> 4;  // 1 or 2 (checked by the launcher)

  const int NWG = (int)gridDim.x;
  const int tile0 = (NWG % 8 == 0) ? ((int)blockIdx.x % 8) * (NWG / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;  // XCD strips, as conv_kernel
  if (tile0 >= num_tiles) return;
  const int ntw = (num_tiles - tile0 + NWG - 1) / NWG;  // tiles of this workgroup: tile0 + j*NWG, j = 0 .. ntw-1
  // phases: two per (tile pair, chunk) + one trailing phase in which group 1 stores its last tile
  const int nphases = 2 * ((ntw + 1) >> 1) * nchunks + 1;

  if (wave == 8) {
    // =========================== LOADER WAVE ===========================
    // Two tiles can be live at once, so the per-tile source map is not kept in registers: each lane keeps the tile-independent
    // part (plane, row, column of its unit in every DMA instruction, packed) and applies the tile origin, the bounds, the zero
    // padding (a zero page) and the nearest-x2 read per item.
    const uint32_t plane_units = (uint32_t)p.in_plane_stride;
    uint32_t smap[DMA_IT];
#pragma unroll
    for (int it = 0; it < DMA_IT; ++it) {
      const int u = it * 64 + lane;
      const int pl = u / PS;
      const int r = u - pl * PS;
      const int py = r / IW;
      const int px = r - py * IW;
      smap[it] = (r < IH * IW) ? (uint32_t)(pl << 16 | py << 8 | px) : 0xFFFFFFFFu;
    }
    auto fill = [&](int j, int q, int buf) {  // chunk q of the workgroup's tile j -> LDS buffer buf
      const int planes_left = p.cin_planes - q * NPL;
      const int tile = tile0 + j * NWG;
      const int n = tile / tiles_img;
      const int tr = tile - n * tiles_img;
      const int ty = tr / tiles_x;
      const int tx = tr - ty * tiles_x;
      const int y0 = ty * TH - HALO, x0 = tx * TW - HALO;
      const uint4* ch = (const uint4*)p.in_hi + (int64_t)n * p.in_batch_stride + (int64_t)q * NPL * p.in_plane_stride;
      const uint4* cl = (PROD == 3) ? (const uint4*)p.in_lo + (int64_t)n * p.in_batch_stride + (int64_t)q * NPL * p.in_plane_stride : nullptr;
      uint4* dst = &s_act[buf][0];
#pragma unroll
      for (int it = 0; it < DMA_IT; ++it) {
        const uint32_t m = smap[it];
        const int pl = (int)(m >> 16);
        int iy = y0 + (int)((m >> 8) & 255u), ix = x0 + (int)(m & 255u);
        const bool ok = m != 0xFFFFFFFFu && (uint32_t)iy < (uint32_t)p.H && (uint32_t)ix < (uint32_t)p.W && pl < planes_left;
        if (UP) {
          iy >>= 1;
          ix >>= 1;
        }
        const uint32_t off = (uint32_t)pl * plane_units + (uint32_t)iy * (uint32_t)inW + (uint32_t)ix;
        const uint4* sh = ok ? ch + off : (const uint4*)&g_zero_unit[0];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sh,
                                         (__attribute__((address_space(3))) void*)(dst + it * 64), 16, 0, 0);
        if (PROD == 3) {
          const uint4* sl = ok ? cl + off : (const uint4*)&g_zero_unit[0];
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sl,
                                           (__attribute__((address_space(3))) void*)(dst + ACT_UNITS + it * 64), 16, 0, 0);
        }
      }
    };
    auto issue = [&](int h) {  // phase h's item: chunk (h/2) % nchunks of tile 2*(h/2/nchunks) + (h & 1), into buffer h & 1
      const int idx = h >> 1;
      const int jj = idx / nchunks;
      const int q = idx - jj * nchunks;
      const int j = 2 * jj + (h & 1);
      if (j < ntw) fill(j, q, h & 1);
    };
    issue(0);
    for (int h = 0; h < nphases; ++h) {
      dma_wait();    // this wave's DMA has landed ...
      wg_barrier();  // ... phase h starts, and the barrier has released the other buffer
      issue(h + 1);
    }
    return;
  }

  // =========================== COMPUTE WAVES ===========================
  const int g = wave >> 2;   // group: waves 0-3 / 4-7, i.e. one wave of each group per SIMD
  const int wpx = wave & 3;  // group of 4 rows
  const int li = lane & 15;
  const int lg = lane >> 4;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w_packed, 0, (uint32_t)((int64_t)nsteps * ct_total * NHL * 64 * 16), 0x00020000);
  uint32_t woff[CTW];  // byte offset of this lane's fragment of (step 0, cout tile, hi); 0xFFFFFFFF (-> zeros) when the tile does not exist
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    woff[c] = (c < ct_total) ? (uint32_t)((c * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
  }
  const uint32_t wstep = (uint32_t)ct_total * NHL * 64 * 16;
  bf16x8 wc[CTW][NHL];  // fragments of the tap being multiplied
  bf16x8 wn[CTW][NHL];  // fragments of the next tap, in flight
  auto load_w = [&](int s) {
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int hl = 0; hl < NHL; ++hl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)s * wstep + (uint32_t)hl * 1024u, 0);
        wn[c][hl] = __builtin_bit_cast(bf16x8, v);
      }
  };

  f32x4 acc[NPT][CTW];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int bunit0 = lg * PS + (wpx * RPW) * IW + li;

  load_w(0);
  int pend_tile = -1;        // tile whose accumulators wait for their epilogue
  int j = g;                 // this group's next tile index within the workgroup
  int q = 0;                 // its next chunk
  const uint4* sa = &s_act[g][0];
  for (int h = 0; h < nphases; ++h) {
    wg_barrier();
    if ((h & 1) == g) {
      if (j < ntw) {
        // ---- multiply chunk q of tile j (software pipeline as in conv_kernel.h: one pixel tile x 2 cout tiles per step) ----
        constexpr int SPT = NPT;
        constexpr int NSTEP = T * SPT;
        constexpr int LDS_DEPTH = 2;
        bf16x8 rh[LDS_DEPTH + 1], rl[LDS_DEPTH + 1];
        auto frag_unit = [&](int i) -> int {
          const int t = i / SPT, pt = i - t * SPT;
          const int dy = t / KS, dx = t - (t / KS) * KS;
          return bunit0 + ((pt >> 1) + dy) * IW + (pt & 1) * 16 + dx;
        };
#pragma unroll
        for (int i = 0; i < LDS_DEPTH; ++i) {
          rh[i] = *(const bf16x8*)&sa[frag_unit(i)];
          if (PROD == 3) rl[i] = *(const bf16x8*)&sa[ACT_UNITS + frag_unit(i)];
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
            load_w(s + 1 < nsteps ? s + 1 : 0);  // next tap; after the last step: step 0 of this group's next tile
            __builtin_amdgcn_sched_barrier(0);
          }
          if (i + LDS_DEPTH < NSTEP) {
            const int u = frag_unit(i + LDS_DEPTH);
            rh[(i + LDS_DEPTH) % (LDS_DEPTH + 1)] = *(const bf16x8*)&sa[u];
            if (PROD == 3) rl[(i + LDS_DEPTH) % (LDS_DEPTH + 1)] = *(const bf16x8*)&sa[ACT_UNITS + u];
          }
          constexpr int NPR = (PROD == 3) ? 3 : 1;
#pragma unroll
          for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
              // products in increasing magnitude: w_lo*a_hi, w_hi*a_lo, w_hi*a_hi
              const bf16x8 wf = (PROD == 3 && pr == 0) ? wc[ct][NHL - 1] : wc[ct][0];
              const bf16x8 bf = (PROD == 3 && pr == 1) ? rl[i % (LDS_DEPTH + 1)] : rh[i % (LDS_DEPTH + 1)];
              acc[sp][ct] = mfma16<FMT>(wf, bf, acc[sp][ct]);
            }
          if (i + LDS_DEPTH < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NHL, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NPR * CTW, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (++q == nchunks) {
          q = 0;
          pend_tile = tile0 + j * NWG;
          j += 2;
        }
      }
    } else if (pend_tile >= 0) {
      // ---- the other group multiplies: store the tile finished one phase ago ----
      const int n = pend_tile / tiles_img;
      const int tr = pend_tile - n * tiles_img;
      const int ty = tr / tiles_x;
      const int tx = tr - ty * tiles_x;
      epilogue<2, CTW, NPT, OUTK>(p, acc, n, ty * TH, tx * TW, 0, 0, wpx, li, lg);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      pend_tile = -1;
    }
  }
}

template <int KS, int PROD, int UP, int OUTK, int FMT = 0>
static int launch_pp(const rsa_conv_params& p, hipStream_t stream) {
  using G = GeoLW<KS, 4>;
  const int tiles_x = (p.W + G::TW - 1) / G::TW;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x3fffffff) return RSA_E_UNSUPPORTED;
  static std::atomic<int> resident_cache{0};  // concurrent first calls compute the same value: idempotent, never torn
  int resident = resident_cache.load(std::memory_order_relaxed);
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_kernel_pp<KS, PROD, UP, OUTK, FMT>, 576, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
    resident_cache.store(resident, std::memory_order_relaxed);
  }
  int gx = resident;
  if (gx > num_tiles) gx = (int)num_tiles;
  hipLaunchKernelGGL((conv_kernel_pp<KS, PROD, UP, OUTK, FMT>), dim3((unsigned)gx, 1, 1), dim3(576), 0, stream, p);
  return (int)hipGetLastError();
}

// Dispatch of the 3x3 families: split-bf16 (three-product) single-slab layers with at most two cout tiles take the alternating
// schedule, everything else conv_kernel.  Single-product layers stay on conv_kernel: their multiply phase is a third as long, the
// phases are then paced by the (serialised) fills, and the alternation measured 5 % slower (125.6 vs 119.3 ms per frame in plain
// bf16 mode).  RSA_CONV_PP=0 in the environment switches it off (A/B runs).
template <int KS, int PROD, int UP, int FMT = 0>
static int launch_nct_pp(const rsa_conv_params& p, int nct, hipStream_t stream) {
  static const bool use_pp = [] {
    const char* e = getenv("RSA_CONV_PP");
    return e == nullptr || e[0] != '0';
  }();
  if constexpr (KS == 3 && PROD == 3) {
    if (nct == 2 && p.cout <= 32 && use_pp) {
      return p.out_nchw != nullptr ? launch_pp<KS, PROD, UP, 1, FMT>(p, stream) : launch_pp<KS, PROD, UP, 0, FMT>(p, stream);
    }
  }
  return launch_nct<KS, PROD, UP, FMT>(p, nct, stream);
}

}  // namespace rsa
