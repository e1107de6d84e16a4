// conv_rs.hip — register-staged schedule of the fused convolution (see conv_mfma.hip for the loader-wave schedule and
// conv_common.h for the shared epilogue).  Two independent 4-wave workgroups per CU; each prefetches the next halo tile
// into VGPRs (raw buffer loads, zero padding from the range check) while it multiplies the current one out of a
// single-buffered LDS image, weights streamed per wave from L2.  Selected for split-bf16 (3 products) layers with at most
// two cout tiles, where two independent workgroups per CU hide synchronisation better than one 9-wave workgroup
// (profiles/r01_b_conv_microbench.txt).
#include "conv_common.h"

namespace rsa {

constexpr int NPL_RS = 4;

// Geometry of one instantiation.  A workgroup is always 4 waves; a wave always owns 8 pixel-tiles (4 rows x 2 halves
// of 16 pixels) x CTW cout-tiles, so that one tap costs it 16 LDS fragment reads + 4 weight fragment loads for up to
// 48 MFMAs.  NCT >= 3: tile 8x32, waves = 2 cout-pairs x 2 row-groups.  NCT <= 2: tile 16x32, waves = 4 row-groups.
template <int KS, int NCT>
struct GeoRS {
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

template <int KS, int NCT, int PROD, int UP, int OUTK>
__global__ __launch_bounds__(256, 2) void conv_kernel_rs(const rsa_conv_params p) {
  using G = GeoRS<KS, NCT>;
  constexpr int TH = G::TH, TW = G::TW, HALO = G::HALO, IH = G::IH, IW = G::IW, PS = G::PS;
  constexpr int WPX = G::WPX, CTW = G::CTW;
  constexpr int ACT_UNITS = NPL_RS * PS;
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int FILL_IT = (ACT_UNITS + 256 - 1) / 256;
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

  const int nchunks = (p.cin_planes + NPL_RS - 1) / NPL_RS;
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
    const int u = it * 256 + tid;
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
    const int planes_left = min(p.cin_planes - q * NPL_RS, NPL_RS);  // >= 1
    const uint32_t nbytes = enable ? (uint32_t)planes_left * plane_bytes : 0u;
    const int64_t chunk_off = (int64_t)q * NPL_RS * p.in_plane_stride * 16;
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
      const int u = it * 256 + tid;
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
int launch_rs(const rsa_conv_params& p, hipStream_t stream) {
  using G = GeoRS<KS, NCT>;
  const int tiles_x = (p.W + G::TW - 1) / G::TW;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int ct_total = (p.cout + 15) / 16;
  const int slabs = (ct_total + NCT - 1) / NCT;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x7fffffff) return RSA_E_UNSUPPORTED;
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_kernel_rs<KS, NCT, PROD, UP, OUTK>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
  }
  int gx = resident / slabs;
  if (gx < 1) gx = 1;
  if (gx > num_tiles) gx = (int)num_tiles;
  dim3 grid((unsigned)gx, (unsigned)slabs, 1);
  hipLaunchKernelGGL((conv_kernel_rs<KS, NCT, PROD, UP, OUTK>), grid, dim3(256), 0, stream, p);
  return (int)hipGetLastError();
}

// the instantiations the dispatcher in conv_mfma.hip uses: 3 products, 1-2 cout tiles, plane/f32 epilogue
template int launch_rs<3, 1, 3, 0, 0>(const rsa_conv_params&, hipStream_t);
template int launch_rs<3, 2, 3, 0, 0>(const rsa_conv_params&, hipStream_t);
template int launch_rs<3, 1, 3, 1, 0>(const rsa_conv_params&, hipStream_t);
template int launch_rs<3, 2, 3, 1, 0>(const rsa_conv_params&, hipStream_t);

}  // namespace rsa
