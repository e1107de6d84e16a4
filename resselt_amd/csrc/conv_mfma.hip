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

#include "resselt_amd.h"
#include "common.h"

namespace rsa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

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

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// (a, b) -> packed bf16 pair (RNE) and the pair's rounding residuals, also packed
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};
  hi = __builtin_bit_cast(uint32_t, h);
  const float ra = a - __builtin_bit_cast(float, hi << 16);
  const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
  const bf16x2 l = {(__bf16)ra, (__bf16)rb};
  lo = __builtin_bit_cast(uint32_t, l);
}

// Geometry of one instantiation.  A workgroup is always 4 waves; a wave always owns 8 pixel-tiles (4 rows x 2 halves
// of 16 pixels) x CTW cout-tiles, so that one tap costs it 16 LDS fragment reads + 4 weight fragment loads for up to
// 48 MFMAs.  NCT >= 3: tile 8x32, waves = 2 cout-pairs x 2 row-groups.  NCT <= 2: tile 16x32, waves = 4 row-groups.
template <int KS, int NCT>
struct Geo {
  static constexpr int WCT = (NCT >= 3) ? 2 : 1;       // waves along cout
  static constexpr int WPX = 4 / WCT;                  // waves along rows
  static constexpr int CTW = (NCT >= 2) ? 2 : 1;       // cout tiles per wave
  static constexpr int TH = 4 * WPX;                   // 8 or 16 output rows
  static constexpr int TW = 32;
  static constexpr int HALO = KS / 2;
  static constexpr int IH = TH + 2 * HALO;
  static constexpr int IW = TW + 2 * HALO;
  static constexpr int PS = ((IH * IW + 15) / 16) * 16;  // plane stride in units, == 0 mod 16
};

// Epilogue of one finished tile.  Every address is  (uniform 64-bit base) + (32-bit per-lane byte offset):
//   lane (li, lg) of wave (wct, wpx) owns, for pixel-tile pt and cout-tile c, the 4 consecutive channels
//   c0 = 16*(slab*NCT + 2*wct + c) + 4*lg .. +3  of pixel (y0 + 4*wpx + (pt>>1), x0 + 16*(pt&1) + li).
// OUTK = 0: split planes and/or f32 residual map (with activation / residual epilogues)
// OUTK = 1: final plain NCHW tensor (optional activation, depth-to-space and affine), any dtype
template <int NCT, int CTW, int OUTK>
__device__ __forceinline__ void epilogue(const rsa_conv_params& p, const f32x4 (&acc)[8][CTW], int n, int y0, int x0, int slab, int wct, int wpx,
                                         int li, int lg) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t pix0 = (int64_t)y0 * p.W + x0;  // uniform
  const int p4 = (p.cout + 3) >> 2;
  const int cout8 = (p.cout + 7) & ~7;
  const int ctile0 = slab * NCT + wct * 2;

  uint32_t lpix[8];
  bool pvalid[8];
#pragma unroll
  for (int pt = 0; pt < 8; ++pt) {
    const int ry = wpx * 4 + (pt >> 1), rx = (pt & 1) * 16 + li;
    lpix[pt] = (uint32_t)(ry * p.W + rx);
    pvalid[pt] = (y0 + ry < p.H) && (x0 + rx < p.W);
  }

#pragma unroll
  for (int ct = 0; ct < CTW; ++ct) {
    if (wct * 2 + ct >= NCT) break;
    const int cbase = (ctile0 + ct) * 16;  // uniform
    if (cbase >= cout8) break;
    const int c0 = cbase + lg * 4;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bias = ((const f32x4*)p.bias)[c0 >> 2];  // bias is padded to a multiple of 16
    const bool cvalid = c0 < cout8;
    const bool has_f32grp = c0 < (p4 << 2);
    // uniform bases for this cout tile
    const char* r1b = (const char*)p.res1 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const char* r2b = (const char*)p.res2 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    char* f32b = (char*)p.out_f32 + (((int64_t)n * p4 + (cbase >> 2)) * HW + pix0) * 16;
    const int64_t ounit0 = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0;
    char* ohb = (char*)p.out_hi + ounit0 * 16;
    char* olb = (char*)p.out_lo + ounit0 * 16;
    const uint32_t f32lane = (uint32_t)lg * (uint32_t)HW;                       // + lpix, in float4 units
    const uint32_t pllane = (uint32_t)(lg >> 1) * (uint32_t)p.out_plane_stride;  // + lpix, in 16-byte units
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      if (!pvalid[pt] || !cvalid) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[pt][ct][r] + bias[r];
      if (OUTK == 0) {
        const uint32_t foff = (f32lane + lpix[pt]) * 16u;
        if (p.act == RSA_ACT_SPAB_GATE) {
          f32x4 rr = {0.f, 0.f, 0.f, 0.f};
          if (has_f32grp) rr = *(const f32x4*)(r1b + foff);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sg = 1.f / (1.f + expf(-v[r]));
            v[r] = (v[r] + rr[r]) * (sg - 0.5f);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act, p.act_param);
          if (p.res1 != nullptr && has_f32grp) {
            const f32x4 rr = *(const f32x4*)(r1b + foff);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + rr[r];
          }
        }
        if (p.res2 != nullptr && has_f32grp) {
          const f32x4 rr = *(const f32x4*)(r2b + foff);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] * p.beta + rr[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (c0 + r >= p.cout) v[r] = 0.f;
        if (p.out_hi != nullptr) {
          uint32_t h0, l0, h1, l1;
          split2(v[0], v[1], h0, l0);
          split2(v[2], v[3], h1, l1);
          const uint32_t uoff = (pllane + lpix[pt]) * 16u + (uint32_t)(lg & 1) * 8u;
          *(uint2*)(ohb + uoff) = make_uint2(h0, h1);
          if (p.out_lo != nullptr) *(uint2*)(olb + uoff) = make_uint2(l0, l1);
        }
        if (p.out_f32 != nullptr && has_f32grp) *(f32x4*)(f32b + foff) = (f32x4){v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], p.act, p.act_param);
        const int ps = p.pixel_shuffle > 1 ? p.pixel_shuffle : 1;
        const int oc_total = p.cout / (ps * ps);
        const int64_t oW = (int64_t)p.W * ps;
        const int64_t oHW = (int64_t)p.H * ps * oW;
        const int y = y0 + wpx * 4 + (pt >> 1), x = x0 + (pt & 1) * 16 + li;
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
          const int64_t idx = ((int64_t)n * oc_total + oc) * oHW + ((int64_t)y * ps + ii) * oW + ((int64_t)x * ps + jj);
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

template <int KS, int NCT, int PROD, int UP, int OUTK>
__global__ __launch_bounds__(NTHREADS, 2) void conv_kernel(const rsa_conv_params p) {
  using G = Geo<KS, NCT>;
  constexpr int TH = G::TH, TW = G::TW, HALO = G::HALO, IH = G::IH, IW = G::IW, PS = G::PS;
  constexpr int WPX = G::WPX, CTW = G::CTW;
  constexpr int ACT_UNITS = NPL * PS;
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int FILL_IT = (ACT_UNITS + NTHREADS - 1) / NTHREADS;
  constexpr int T = KS * KS;

  __shared__ uint4 s_act[NHL * ACT_UNITS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wct = wave / WPX;  // which cout pair
  const int wpx = wave - wct * WPX;  // which group of 4 rows
  const int li = lane & 15;
  const int lg = lane >> 4;

  const int tiles_x = (p.W + TW - 1) / TW;
  const int tiles_y = (p.H + TH - 1) / TH;
  const int tiles_img = tiles_x * tiles_y;
  const int num_tiles = tiles_img * p.batch;
  const int slab = blockIdx.y;

  const int inW = UP ? (p.W >> 1) : p.W;

  const int nchunks = (p.cin_planes + NPL - 1) / NPL;
  const int nsteps = nchunks * T;
  const int ct_total = (p.cout + 15) >> 4;

  // ---- per-thread halo-fill map of the tile being FETCHED (recomputed when the prefetch moves to a new tile).
  //      foff = BYTE offset from the chunk's first plane, or 0xFFFFFFFF for zero padding: the fetch is a raw buffer
  //      load whose descriptor covers exactly the chunk's valid planes, so padding pixels AND missing planes come
  //      back as zeros from the hardware range check -- no branch, no select, loads stay in flight (counted vmcnt). ----
  uint32_t foff[FILL_IT];
  // tile-independent part of the map, packed so that it costs ONE register per fill iteration:
  //   bits 0-1 plane in chunk, bits 2-7 row in halo tile, bits 8-13 column, bit 16 = slot is part of the halo tile
  uint32_t fpk[FILL_IT];
#pragma unroll
  for (int it = 0; it < FILL_IT; ++it) {
    const int u = it * NTHREADS + tid;
    const int pl = u / PS;
    const int r = u - pl * PS;
    const int py = r / IW;
    const int px = r - py * IW;
    fpk[it] = (uint32_t)pl | ((uint32_t)py << 2) | ((uint32_t)px << 8) | ((u < ACT_UNITS && r < IH * IW) ? 0x10000u : 0u);
    asm volatile("" : "+v"(fpk[it]));  // keep it packed: do not let the compiler hoist the unpacked fields as loop invariants
  }
  const char* f_hi = nullptr;  // image base of the tile being fetched
  const char* f_lo = nullptr;
  const uint32_t plane_bytes = (uint32_t)p.in_plane_stride * 16u;
  auto set_fill_tile = [&](int tile) {
    const int n = tile / tiles_img;
    const int tr = tile - n * tiles_img;
    const int ty = tr / tiles_x;
    const int tx = tr - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    f_hi = (const char*)p.in_hi + (int64_t)n * p.in_batch_stride * 16;
    if (PROD == 3) f_lo = (const char*)p.in_lo + (int64_t)n * p.in_batch_stride * 16;
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const uint32_t k = fpk[it];
      int iy = y0 - HALO + (int)((k >> 2) & 63u);
      int ix = x0 - HALO + (int)((k >> 8) & 63u);
      const bool ok = (k & 0x10000u) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      if (UP) {
        iy >>= 1;
        ix >>= 1;
      }
      foff[it] = ok ? ((k & 3u) * plane_bytes + ((uint32_t)iy * (uint32_t)inW + (uint32_t)ix) * 16u) : 0xFFFFFFFFu;
    }
  };

  uint4 st_hi[FILL_IT];
  uint4 st_lo[(PROD == 3) ? FILL_IT : 1];

  // `enable == false` issues the same loads against an empty descriptor (all zeros, no memory traffic): the fetch stays
  // straight-line code, so the compiler can keep it in flight behind a COUNTED vmcnt instead of draining at a join
  auto load_act = [&](int q, bool enable) {
    const int planes_left = min(p.cin_planes - q * NPL, NPL);  // >= 1
    const uint32_t nbytes = enable ? (uint32_t)planes_left * plane_bytes : 0u;
    const int64_t chunk_off = (int64_t)q * NPL * p.in_plane_stride * 16;
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void*)(f_hi + chunk_off), 0, nbytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < FILL_IT; ++it) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rh, foff[it], 0, 0);
      st_hi[it] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    if (PROD == 3) {
      const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(f_lo + chunk_off), 0, nbytes, 0x00020000);
#pragma unroll
      for (int it = 0; it < FILL_IT; ++it) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rl, foff[it], 0, 0);
        st_lo[it] = make_uint4(v[0], v[1], v[2], v[3]);
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

  // ---- weights: every wave streams ITS OWN A fragments (cout tiles 2*wct, 2*wct+1 of this slab) straight from the
  //      L2-resident packed blob into VGPRs, one tap ahead.  No LDS, no barrier: waves never share weight registers. ----
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w_packed, 0, (uint32_t)((int64_t)nsteps * ct_total * NHL * 64 * 16), 0x00020000);
  uint32_t woff[CTW];  // byte offset of this lane's fragment of (step 0, cout tile c, hi); 0xFFFFFFFF when the tile does not exist
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    const int ctg = slab * NCT + wct * 2 + c;
    woff[c] = (wct * 2 + c < NCT && ctg < ct_total) ? (uint32_t)((ctg * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
  }
  const uint32_t wstep = (uint32_t)ct_total * NHL * 64 * 16;  // bytes per step
  bf16x8 wc[CTW][NHL];  // fragments of the tap being multiplied
  bf16x8 wn[CTW][NHL];  // fragments of the next tap, in flight
  auto load_w = [&](int s) {
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int hl = 0; hl < NHL; ++hl) {
        // a missing cout tile keeps offset 0xFFFFFFFF (s*wstep is far below the wrap) -> zeros from the range check
        const uint32_t off = woff[c];
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, off, (uint32_t)s * wstep + (uint32_t)hl * 1024u, 0);
        wn[c][hl] = __builtin_bit_cast(bf16x8, v);
      }
  };

  f32x4 acc[8][CTW];
#pragma unroll
  for (int pt = 0; pt < 8; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // B-fragment unit of (pixel-tile pt, tap 0,0) for this lane
  const int bunit0 = lg * PS + (wpx * 4) * IW + li;

  // ---- persistent loop over this workgroup's tiles; the (tile, chunk) stream is prefetched one item ahead,
  //      so only the very first tile of a workgroup exposes its global-load latency ----
  int tile = blockIdx.x;
  if (tile >= num_tiles) return;
  load_w(0);  // weights first: vmcnt retires in order
  set_fill_tile(tile);
  load_act(0, true);
  int q = 0;
  while (true) {
    __syncthreads();  // every wave is done reading the previous item's halo tile
    store_act();
    __syncthreads();
    const bool last_chunk = (q == nchunks - 1);
    const int ntile = tile + (int)gridDim.x;
    const bool more = !last_chunk || (ntile < num_tiles);
    // software pipeline over the 8 pixel tiles: fragments of (t, pt+1) are read from LDS while (t, pt) multiplies
    bf16x8 bh = *(const bf16x8*)&s_act[bunit0];
    bf16x8 bl;
    if (PROD == 3) bl = *(const bf16x8*)&s_act[ACT_UNITS + bunit0];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int s = q * T + t;
#pragma unroll
      for (int c = 0; c < CTW; ++c)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) wc[c][hl] = wn[c][hl];
      load_w(s + 1 < nsteps ? s + 1 : 0);  // next tap's weights (wraps to step 0 of the next tile)
      if (t == 0) {
        if (last_chunk) set_fill_tile(ntile);
        load_act(last_chunk ? 0 : q + 1, more);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
        // next fragment: (t, pt+1), or (t+1, 0) across the tap boundary
        const int nt = (pt == 7) ? t + 1 : t;
        const int npt = (pt == 7) ? 0 : pt + 1;
        bf16x8 nbh = bh, nbl = bh;
        if (nt < T) {
          const int ndy = nt / KS, ndx = nt - (nt / KS) * KS;
          const int u = bunit0 + ((npt >> 1) + ndy) * IW + (npt & 1) * 16 + ndx;
          nbh = *(const bf16x8*)&s_act[u];
          if (PROD == 3) nbl = *(const bf16x8*)&s_act[ACT_UNITS + u];
        }
        if (PROD == 3) {
#pragma unroll
          for (int ct = 0; ct < CTW; ++ct) {
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ct][NHL - 1], bh, acc[pt][ct], 0, 0, 0);
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ct][0], bl, acc[pt][ct], 0, 0, 0);
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ct][0], bh, acc[pt][ct], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int ct = 0; ct < CTW; ++ct)
            acc[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ct][0], bh, acc[pt][ct], 0, 0, 0);
        }
        bh = nbh;
        if (PROD == 3) bl = nbl;
        // issue order inside the step: the next fragment reads first, then this step's MFMAs (the reads then have
        // the whole MFMA group to land; the waits become counted lgkmcnt(NHL))
        __builtin_amdgcn_sched_group_barrier(0x100, NHL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (PROD == 3 ? 3 : 1) * CTW, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!last_chunk) {
      ++q;
      continue;
    }
    // ---- tile finished: epilogue for `tile`, then move on (next tile's first chunk is already in flight) ----
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
    tile = ntile;
    q = 0;
    if (tile >= num_tiles) break;
  }
}

template <int KS, int NCT, int PROD, int UP, int OUTK>
static int launch_one(const rsa_conv_params& p, hipStream_t stream) {
  using G = Geo<KS, NCT>;
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_kernel<KS, NCT, PROD, UP, OUTK>, NTHREADS, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
  }
  int gx = resident / slabs;
  if (gx < 1) gx = 1;
  if (gx > num_tiles) gx = (int)num_tiles;
  dim3 grid((unsigned)gx, (unsigned)slabs, 1);
  hipLaunchKernelGGL((conv_kernel<KS, NCT, PROD, UP, OUTK>), grid, dim3(NTHREADS), 0, stream, p);
  return (int)hipGetLastError();
}

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
  if (p.act < 0 || p.act > RSA_ACT_SPAB_GATE) return set_error(RSA_E_ARG, "conv: bad act");
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
