// conv_ring_up.h — nearest x2 upsampling + 3x3 convolution (reference utilities/block.py:510-537 `upconv_block`: nn.Upsample(nearest)
// -> Conv2d 3x3 -> LeakyReLU; archs/esrgan/arch.py:104-118, archs/swinir/arch.py:1000-1006) as FOUR 2x2 convolutions on the
// low-resolution map, one per output phase (dy, dx) = (Y & 1, X & 1).
//
// On the upsampled image the nine taps of output pixel (2y + dy, 2x + dx) read only 2 x 2 distinct source pixels: rows y - 1 + dy and
// y + dy, columns x - 1 + dx and x + dx.  Summing the weights of the taps that share a source pixel (host side, in f32, once:
// weight layout 3) leaves 4 taps per phase instead of 9: 2.25 x fewer MFMAs than conv_ring<1, UP = 1>, which stages the UPSAMPLED halo
// tile (every source unit fetched four times) and multiplies all nine taps.
//
// Schedule: the ring of conv_ring.h on the LOW-resolution map.  A 64-channel input is exactly the four ring slots, so a tile (16 x 32
// source pixels -> 32 x 64 outputs) is filled once and read by all four phases: per phase 4 half chunks x 2 K steps (a K step = 16
// channels x the two column taps of one source row); in MFMAs, 32 full K steps per tile where the upsampled schedule needs 72 for the
// same outputs.  Slots are released during the last phase; the loader refills them for the next tile under its tail.
// Eight compute waves = 2 cout groups x 4 row groups (conv_ring's SHAPE 1) + one loader wave.
#pragma once
#include "conv_ring.h"

namespace rsa {

__global__ __launch_bounds__(9 * 64, 3) void conv_ring_up2(const rsa_conv_params p, const RingAux aux) {
  using R = RingGeo;
  constexpr int TH = R::TH, TW = R::TW, IH = R::IH, IW = R::IW, PS = R::PS, HALF = R::HALF, SLOT = R::SLOT, NSLOT = R::NSLOT;
  constexpr int NCONS = 8, NHALF = 4;
  __shared__ uint4 s_ring[NSLOT * SLOT + 4];
  uint32_t* const flags = (uint32_t*)&s_ring[NSLOT * SLOT];  // [0..3] FULL, [4..7] FREE, [8] abort
  uint32_t* const f_full = flags;
  uint32_t* const f_free = flags + 4;
  uint32_t* const f_abort = flags + 8;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hs = p.H >> 1, Ws = p.W >> 1;  // the source map (the descriptor's H, W are the output's)
  const int tiles_x = (Ws + TW - 1) / TW;
  const int tiles_y = (Hs + TH - 1) / TH;
  const int num_tiles = tiles_x * tiles_y * p.batch;
  const int NWG = (int)gridDim.x;
  const int tile0 = (NWG % 8 == 0) ? ((int)blockIdx.x % 8) * (NWG / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;  // XCD strips
  if (tile0 >= num_tiles) return;
  const int ntw = (num_tiles - tile0 + NWG - 1) / NWG;

  if (tid < 16) flags[tid] = 0;
  __syncthreads();  // the only workgroup barrier of the kernel

  if (wave >= 8) {
    // =========================== LOADER WAVE: conv_ring's, on the source map ===========================
    __builtin_amdgcn_s_setprio(3);
    uint32_t lc[R::DMA_IT];
    uint32_t smap[R::DMA_IT];
#pragma unroll
    for (int it = 0; it < R::DMA_IT; ++it) {
      const int u = it * 64 + lane;
      const int pl = (u / PS) & 1;
      const int r = u % PS;
      int py = r / IW, px = r - (r / IW) * IW;
      if (r >= IH * IW) py = 0, px = 0;
      smap[it] = (uint32_t)(pl << 16 | py << 8 | px);
      lc[it] = (uint32_t)(((int64_t)pl * p.in_plane_stride + (int64_t)py * Ws + px) * 16);
    }
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)s_ring;
    int k = 0;
    int pend_slot = -1;
    uint32_t pend_val = 0;
    for (int j = 0; j < ntw; ++j) {
      int n, ty, tx;
      ring_tile_coords(tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
      const int y0 = ty * TH - 1, x0 = tx * TW - 1;  // halo origin on the source map
      const bool interior = y0 >= 0 && x0 >= 0 && y0 + IH <= Hs && x0 + IW <= Ws;
      const int64_t tile_unit = (int64_t)n * p.in_batch_stride + (int64_t)y0 * Ws + x0;
      for (int h = 0; h < NHALF; ++h, ++k) {
        const int slot = k & 3;
        const uint32_t use = (uint32_t)(k >> 2);
        if (__builtin_amdgcn_readfirstlane(lds_ld(&f_free[slot])) < NCONS * use) {
          if (pend_slot >= 0) {  // publish what has been issued before blocking: the consumers may need it to get here
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&f_full[pend_slot], pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pend_slot = -1;
          }
          ring_wait(&f_free[slot], NCONS * use, f_abort, aux);
        }
        const int64_t half_unit = tile_unit + (int64_t)(2 * h) * p.in_plane_stride;
        gcptr bh = uniform_ptr((gcptr)p.in_hi + half_unit * 16);
        gcptr bl = uniform_ptr((gcptr)p.in_lo + half_unit * 16);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * (SLOT * 16));
        if (interior) {
#pragma unroll
          for (int it = 0; it < R::DMA_IT; ++it) {
            if (it == R::DMA_IT - 1 && lane >= 32) continue;
            dma16_s(dst + it * 1024, lc[it], bh);
            dma16_s(dst + HALF * 16 + it * 1024, lc[it], bl);
          }
        } else {
#pragma unroll
          for (int it = 0; it < R::DMA_IT; ++it) {
            if (it == R::DMA_IT - 1 && lane >= 32) continue;
            const uint32_t m = smap[it];
            const int pl = (int)(m >> 16);
            const int iy = y0 + (int)((m >> 8) & 255u), ix = x0 + (int)(m & 255u);
            const bool ok = (uint32_t)iy < (uint32_t)Hs && (uint32_t)ix < (uint32_t)Ws;
            const int64_t off = ((int64_t)n * p.in_batch_stride + (int64_t)(2 * h + pl) * p.in_plane_stride + (int64_t)iy * Ws + ix) * 16;
            gcptr sh = ok ? (gcptr)p.in_hi + off : (gcptr)&g_zero_unit[0];
            gcptr sl = ok ? (gcptr)p.in_lo + off : (gcptr)&g_zero_unit[0];
            dma16_v(dst + it * 1024, sh);
            dma16_v(dst + HALF * 16 + it * 1024, sl);
          }
        }
        if (pend_slot >= 0) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * R::DMA_IT) : "memory");
          __hip_atomic_store(&f_full[pend_slot], pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        pend_slot = slot;
        pend_val = use + 1;
      }
    }
    if (pend_slot >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&f_full[pend_slot], pend_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // =========================== COMPUTE WAVES ===========================
  const int wct = wave >> 2;  // cout group: tiles 2*wct, 2*wct + 1
  const int wpx = wave & 3;   // rows 4*wpx .. +3 of the source tile
  const int li = lane & 15;
  const int lg = lane >> 4;
  const int hsel = lg >> 1;  // which of the two column taps of a K step

  // weights: [phase 4][half chunk 4][source row 2][cout tile 4][hi|lo] 1 KiB fragments, streamed one K step ahead
  constexpr int NSTEP = 32;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_packed, 0, (uint32_t)(NSTEP * 4 * 2 * 64 * 16), 0x00020000);
  uint32_t woff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) woff[c] = (uint32_t)(((wct * 2 + c) * 2 * 64 + lane) * 16);
  constexpr uint32_t wstep = 4 * 2 * 64 * 16;
  const int lane_u = (lg & 1) * PS + (wpx * 4) * IW + li + hsel;  // plane lg & 1, this wave's first row, column li + column tap
  const float slope = p.act == RSA_ACT_NONE ? 1.f : p.act_param;
  uint32_t kc = 0;
  for (int j = 0; j < ntw; ++j) {
    int n, ty, tx;
    ring_tile_coords(tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
    // the whole tile (four half chunks = the four slots) is needed by every phase: wait for it here, where no accumulator is live
#pragma unroll
    for (int h = 0; h < NHALF; ++h) ring_wait(&f_full[(kc + h) & 3], ((kc + h) >> 2) + 1, f_abort, aux);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");

    // One linear loop of 64 K steps: step i = (dy, row half rh, dx, half chunk h, source row s), most significant first.  The two column
    // phases of the same output rows run back to back on HALF of the wave's rows (4 pixel tiles): the first one's results wait as plane
    // units (32 registers) and both are stored together, two adjacent 16-byte units per lane = 512 contiguous bytes per 16 lanes.  Stored
    // one phase at a time (16 bytes every 32) the kernel was store-bound: 1.17 ms per 8.3 M outputs, 0.66 ms with the stores removed.
    const int y0 = ty * TH, x0 = tx * TW;
    f32x4 acc[4][2];
    uint4 u0h[2][2], u0l[2][2];  // [cout tile][row of the half]: the dx = 0 phase
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[e][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Operands one K step ahead, two K steps (the two source rows of a half chunk) per iteration so that the buffers are named.  No
    // fetch sits under a branch (a conditional fetch makes the compiler wait for every outstanding load before the next MFMA): the
    // last iteration prefetches step 0 of THIS tile again, and the next tile fetches its own after its wait.
    auto step_t = [](int i) -> int { return (((i >> 5) * 2 + ((i >> 3) & 1)) * 4 + ((i >> 1) & 3)) * 2 + (i & 1); };  // weight K step of loop step i
    auto unit_of = [&](int i) -> int {  // LDS unit of pixel tile 0 of loop step i
      const int s = i & 1, h = (i >> 1) & 3, dx = (i >> 3) & 1, rh = (i >> 4) & 1, dy = i >> 5;
      return (int)((kc + h) & 3) * SLOT + lane_u + (dy + 2 * rh + s) * IW + dx;
    };
    bf16x8 w2[2][2][2], bh[2][4], bl[4];  // [buffer = source row s]; the lo fragments are read at the head of their own step (registers)
    auto fetch = [&](int i, int b) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
          w2[b][c][hl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)step_t(i) * wstep + (uint32_t)hl * 1024u, 0));
      const int ub = unit_of(i);
#pragma unroll
      for (int e = 0; e < 4; ++e) bh[b][e] = __builtin_bit_cast(bf16x8, s_ring[ub + (e >> 1) * IW + (e & 1) * 16]);
    };
    auto multiply = [&](int i, int b) {  // the products with the hi fragments first: the lo fragments land meanwhile
      const int ub = unit_of(i);
#pragma unroll
      for (int e = 0; e < 4; ++e) bl[e] = __builtin_bit_cast(bf16x8, s_ring[HALF + ub + (e >> 1) * IW + (e & 1) * 16]);
#ifdef RSA_UP2_NOMFMA
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" ::"v"(bh[b][e]), "v"(bl[e]), "v"(w2[b][0][0]), "v"(w2[b][0][1]), "v"(w2[b][1][0]), "v"(w2[b][1][1]));
      return;
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[b][c][1], bh[b][e], acc[e][c], 0, 0, 0);
          acc[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[b][c][0], bh[b][e], acc[e][c], 0, 0, 0);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[e][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[b][c][0], bl[e], acc[e][c], 0, 0, 0);
    };
    fetch(0, 0);
#pragma unroll 1
    for (int i = 1; i < 64; i += 2) {  // i = the odd step of the pair (s = 1)
      const int h = (i >> 1) & 3, dx = (i >> 3) & 1, rh = (i >> 4) & 1, dy = i >> 5;
      const int slot = (int)((kc + h) & 3);
      fetch(i, 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(i - 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      fetch((i + 1) & 63, 0);
      __builtin_amdgcn_sched_barrier(0);
      multiply(i, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 56) {  // the last (dy, rh, dx) has read this half chunk for the last time (the prefetch of the last iteration reads a
                      // slot that may already be refilling: its data is never used)
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&f_free[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if ((i & 7) != 7) continue;
      // ---- eight K steps = one phase of two rows done: bias, LeakyReLU, plane units; after the second column phase: store ----
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ct = wct * 2 + c;
        const f32x4 bias = p.bias != nullptr ? ((const f32x4*)p.bias)[ct * 4 + lg] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const int64_t ounit0 = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + 2 * ct + (lg >> 1)) * p.out_plane_stride;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          f32x4 v[2];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = acc[2 * rr + e][c][r] + bias[r];
              v[e][r] = fmaxf(t, t * slope);
              acc[2 * rr + e][c][r] = 0.f;
            }
          uint4 uh, ul;
          pair_units(v[0], v[1], uh, ul);  // even lane groups: the unit of the left 16 pixels, odd ones: of the right 16
          if (dx == 0) {
            u0h[c][rr] = uh;
            u0l[c][rr] = ul;
            continue;
          }
          const int ys = y0 + wpx * 4 + 2 * rh + rr, xs = x0 + 16 * (lg & 1) + li;
#ifdef RSA_UP2_NOSTORE
          asm volatile("" ::"v"(uh.x), "v"(uh.y), "v"(uh.z), "v"(uh.w), "v"(ul.x), "v"(ul.y), "v"(ul.z), "v"(ul.w));
          asm volatile("" ::"v"(u0h[c][rr].x), "v"(u0h[c][rr].y), "v"(u0h[c][rr].z), "v"(u0h[c][rr].w), "v"(u0l[c][rr].x), "v"(u0l[c][rr].y), "v"(u0l[c][rr].z), "v"(u0l[c][rr].w));
          if (ys < 0) {
#else
          if (ys < Hs && xs < Ws) {
#endif
            const int64_t u = ounit0 + (int64_t)(2 * ys + dy) * p.W + 2 * xs;
            ((uint4*)p.out_hi)[u] = u0h[c][rr];
            ((uint4*)p.out_hi)[u + 1] = uh;
            ((uint4*)p.out_lo)[u] = u0l[c][rr];
            ((uint4*)p.out_lo)[u + 1] = ul;
          }
        }
      }
    }
    kc += NHALF;
  }
}

static int launch_ring_up2(const rsa_conv_params& p, hipStream_t stream) {
  using R = RingGeo;
  const int tiles_x = ((p.W >> 1) + R::TW - 1) / R::TW;
  const int tiles_y = ((p.H >> 1) + R::TH - 1) / R::TH;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x3fffffff) return RSA_E_UNSUPPORTED;
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  int gx = cus;  // one persistent workgroup per CU (the ring takes the whole LDS)
  if (gx > num_tiles) gx = (int)num_tiles;
  hipLaunchKernelGGL(conv_ring_up2, dim3((unsigned)gx, 1, 1), dim3(9 * 64), 0, stream, p, ring_aux());
  return (int)hipGetLastError();
}

}  // namespace rsa
