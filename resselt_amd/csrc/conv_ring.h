// conv_ring.h — ring schedule of the fused 3x3 convolution (three products on split planes, or ONE product on hi planes) for layers whose input is a whole number of
// 16-channel half chunks and that have 17..64 output channels: every convolution of a residual dense block (utilities/block.py:454-465
// of the reference: 64/96/128/160 -> 32 and 192 -> 64), the trunk / upsampling / HR convolutions of RRDBNet (archs/esrgan/arch.py:82-118),
// the 48 -> 48 re-parameterised convolutions of SPAN / SPANPlus / SpanPP (archs/spanplus/arch.py:94-130; 1.5 chunks: the trailing half
// chunk costs 5 K steps instead of 9) and any other 64-channel 3x3 layer (SRVGGNetCompact's body, archs/compact/arch.py:28-52).
//
// What it changes against conv_kernel.h / conv_kernel_pp.h (one workgroup barrier per 32-channel chunk, one halo fill in flight,
// a fill has exactly one multiply phase to land):
//
//   * the halo tile is staged in HALF chunks of 16 channels (2 planes, hi + lo = 39 KB) through a ring of FOUR LDS slots, so a
//     fill is in flight while the previous one is multiplied AND while the one before that is still being read;
//   * no workgroup barrier in the steady state.  A slot has two LDS words: FULL (fills completed, written by the loader wave after
//     a counted vmcnt) and FREE (releases, added to by each consumer wave after its last read).  Waves therefore drift: a wave that
//     stores its tile (epilogue) does not hold up the wave it shares a SIMD with, and the loader keeps streaming across tile
//     boundaries.  Every spin is bounded and ends the kernel through an abort word (g_ring_aborts counts them; tests assert 0);
//   * one MFMA K step (32) = 16 channels x 2 taps (lane group lg: plane = lg & 1, tap = pair[lg >> 1]); nine K steps per 32
//     channels as before: four tap pairs on half A, one step pairing tap (2,2) of half A with tap (2,2) of half B, four pairs on B.
//     Bank-conflict free for the same reason as conv_kernel.h (the two lane groups of a ds_read_b128 slot group differ by a plane);
//   * the loader's source addresses are (scalar tile base) + (per-lane constant): zero vector instructions per LDS-DMA on interior
//     tiles (global_load_lds_dwordx4 v_off, s[base]); tiles that touch the image border take a per-lane path with the zero page;
//   * tiles are ordered in bands of four tile rows, column-major inside a band, so that the 32 consecutive tiles an XCD works on at
//     any time form a 4 x 8 block: the halo rows shared with the tile above / below are L2 hits instead of a second HBM fetch.
//
// One-product mode (PROD 1, round 3: the fp16 layers of the 'auto' precision policy).  A slot holds hi planes only (19.5 KB), so the ring has
// EIGHT slots (four per stream in SHAPE 2); the loader keeps two fills in flight behind the one it publishes; a K step is a third as long, so
// the weight fragments are requested two K steps ahead and the LDS fragments four pixel-tile steps ahead.  FMT selects the matrix instruction
// (v_mfma_f32_16x16x32_bf16 / _f16); it is orthogonal to PROD (the trunk convolution of RRDBNet runs three fp16 products).
//
// Three shapes:  SHAPE 2 (Cout <= 32): waves 0-3 / 4-7 are two independent streams, each with its own tiles, two ring slots
// and loader wave (8 / 9); a wave owns 4 rows x 32 pixels x 2 cout tiles, so each activation fragment read from LDS feeds both
// cout tiles.  SHAPE 1 (Cout 49..64): eight waves = 2 cout groups x 4 row groups on one tile stream over all four slots.
// SHAPE 3 (Cout 33..48): eight waves = 8 groups of 2 rows, each owning all 3 cout tiles, one stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_common.h"
#include "conv_kernel.h"

namespace rsa {

#ifdef RSA_RING_DEBUG
// experiment builds only (tools/variant.sh): runtime ablation mask, set through rsa_debug_ring_flags (not in the product library)
//   1 = loader publishes without issuing DMA   2 = consumers skip the multiply (waits and releases stay)   4 = no weight loads
//   8 = no epilogue   16 = no LDS fragment reads
static __device__ unsigned int g_ring_dbg;
#define RING_DBG(bit) (dbg & (bit))
#else
#define RING_DBG(bit) false
#endif

static __device__ unsigned int g_ring_aborts;  // spins that ran into their bound (a protocol bug): never non-zero in a correct build
// (one counter per translation unit that instantiates the schedule; conv_ring_aborts() adds them up)

static unsigned int ring_aborts_this_unit() {  // reads AND clears (rsa_debug_ring_aborts: "since the last call")
  unsigned int v = 0;
  const unsigned int zero = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ring_aborts), sizeof(v)) != hipSuccess) return 0x10000000u;
  if (v != 0 && hipMemcpyToSymbol(HIP_SYMBOL(g_ring_aborts), &zero, sizeof(zero)) != hipSuccess) return 0x10000000u;
  return v;
}

typedef const __attribute__((address_space(1))) char* gcptr;

// tuning knobs of the one-product instantiations (variant builds override them: tools/variant.sh)
#ifndef RSA_RING_INFL1
#define RSA_RING_INFL1 2  // fills left in flight behind the one being published
#endif
#ifndef RSA_RING_WD1
#define RSA_RING_WD1 3  // K steps of weight prefetch (RRDBNet frame: 2 steps 94.05 ms, 3 steps 93.2 ms, 4 steps 92.85 ms: profiles/r03_q_*)
#endif
#ifndef RSA_RING_WDX
#define RSA_RING_WDX 4  // ... of the two RRDBNet kernels that call their epilogue directly (XRES 1 / 2): they have the registers for a fourth step
#endif
#ifndef RSA_RING_DEPTH1
#define RSA_RING_DEPTH1 4  // pixel-tile steps of LDS fragment prefetch (3, 4, 5: +-0.1 %, profiles/r03_q_*)
#endif

// WL 1 (XRES 3): the layer's whole weight blob lives in LDS behind the ring (the 48 -> 48 layers of the SPAN family: 45 KB), so the ring has
// four slots instead of eight and the loader keeps one fill in flight behind the one it publishes
template <int PROD, int WL = 0>
struct RingGeoP {
  static constexpr int TH = 16, TW = 32, IH = 18, IW = 34;
  static constexpr int PS = 624;             // plane stride in units (IH*IW = 612 rounded up to 0 mod 16)
  static constexpr int HALF = 2 * PS;        // units of one precision of a slot (2 planes)
  static constexpr int NHL = PROD == 3 ? 2 : 1;
  static constexpr int SLOT = NHL * HALF;    // units per slot: [hi p0][hi p1]([lo p0][lo p1])
  static constexpr int NSLOT = (PROD == 3 || WL) ? 4 : 8;
  static constexpr int DMA_IT = (HALF + 63) / 64;  // 20 LDS-DMA instructions per precision; the last one covers 32 units
  static constexpr int DPF = NHL * DMA_IT;   // LDS-DMA instructions per fill
  static constexpr int INFL = (PROD == 3 || WL) ? 1 : RSA_RING_INFL1;  // fills left in flight behind the one being published (INFL * DPF <= 63: vmcnt)
  static constexpr int WL_FRAGS = WL ? 45 : 0;  // 1 KiB weight fragments resident in LDS (15 K steps x 3 cout tiles)
  static constexpr int BAND = 4;             // tile rows per band of the tile order
  static constexpr int FLAG_UNITS = 5;       // uint4s behind the ring: FULL[8], FREE[8], abort
};
using RingGeo = RingGeoP<3>;

// tile index (band order) -> image, tile row, tile column
__device__ __forceinline__ void ring_tile_coords(int t, int tiles_x, int tiles_y, int& n, int& ty, int& tx) {
  const int tiles_img = tiles_x * tiles_y;
  n = t / tiles_img;
  const int r = t - n * tiles_img;
  const int band = r / (RingGeo::BAND * tiles_x);
  const int rr = r - band * RingGeo::BAND * tiles_x;
  const int rows = min(RingGeo::BAND, tiles_y - band * RingGeo::BAND);
  tx = rr / rows;
  ty = band * RingGeo::BAND + (rr - tx * rows);
}

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Wait until *flag >= need.  The spin is bounded: when it runs out (a protocol bug), or a sibling wave's has, the abort word is
// set and the wait returns anyway -- every later wait of the workgroup then returns after at most 64 polls, so the kernel always
// drains (with wrong results, and g_ring_aborts != 0 to say so) instead of hanging the GPU.
__device__ __forceinline__ void ring_wait_slow(const uint32_t* flag, uint32_t need, uint32_t* abort_word, const RingAux& aux) {
  const int limit = __builtin_amdgcn_readfirstlane(aux.spin_limit);
  for (int spin = 0; spin < limit; ++spin) {
    __builtin_amdgcn_s_sleep(1);
    if (__builtin_amdgcn_readfirstlane(lds_ld(flag)) >= need) return;
    if ((spin & 63) == 63 && __builtin_amdgcn_readfirstlane(lds_ld(abort_word)) != 0) return;
  }
  __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&g_ring_aborts, 1u);
    // the host-visible word (pinned, mapped): rsa_check_status() turns it into RSA_E_INTERNAL without synchronising anything
    if (aux.fail_word != nullptr) __hip_atomic_fetch_add(aux.fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__device__ __forceinline__ void ring_wait(const uint32_t* flag, uint32_t need, uint32_t* abort_word, const RingAux& aux) {
  if (__builtin_expect(__builtin_amdgcn_readfirstlane(lds_ld(flag)) < need, 0)) ring_wait_slow(flag, need, abort_word, aux);
}

__device__ __forceinline__ gcptr uniform_ptr(gcptr q) {  // tell the compiler a pointer is wave-uniform (an SGPR pair)
  const uint64_t v = (uint64_t)q;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (gcptr)(((uint64_t)hi << 32) | lo);
}

// LDS-DMA, 16 bytes per lane: LDS[m0 + lane*16] = *(saddr + voff)   /   = *vaddr
__device__ __forceinline__ void dma16_s(uint32_t lds_addr, uint32_t voff, gcptr sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma16_v(uint32_t lds_addr, gcptr vaddr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(vaddr) : "memory");
}

// SHAPE 2: two streams, 2 cout tiles (Cout 17..32).  SHAPE 1: one stream, 4 cout tiles as 2 cout groups x 4 row groups (Cout 49..64).
// SHAPE 3: one stream, 3 cout tiles, 8 row groups of 2 rows each owning all three (Cout 33..48: the 48-channel SPAN family).
// HM 0: the unit of work is a 32-channel chunk (two ring slots, nine K steps).  HM 1 ("half mode", layers with an odd number of half
// chunks: 48 input channels = 3): the unit is one half chunk, five K steps -- four tap pairs and a step in which only lane groups
// 0-1 carry tap (2,2) (the other two multiply zero weights); 15 K steps for 48 channels where whole chunks would need 18.  A second
// kernel rather than a second loop body: two unrolled bodies in one kernel spill hundreds of registers.
// FMT: enum rsa_plane_fmt of the input planes and weights.  PROD: 3 = hi*hi + lo*hi + hi*lo on split planes, 1 = hi*hi on hi planes.
// XRES 1 (SHAPE 1, one fp16 product, whole chunks: conv5 of a residual dense block, whose residual `x5 * 0.2 + x` is the first 64 channels of
// its own input): the K loop walks the 32-channel chunks from the LAST to the first, so x's four half chunks are the last fills of a tile;
// their slots are not handed back until the epilogue has read the residual's hi halves from them (conv_common.h, XL) -- 128 B per pixel that
// are not fetched from memory a second time.  The loader runs ahead into the other four slots meanwhile.
// XRES 3 (SHAPE 3, half mode, one fp16 product, at most three half chunks: the re-parameterised 48 -> 48 layers of SPAN / SPANPlus / SpanPP,
// archs/spanplus/arch.py:94-130): the weight blob (45 KB) is copied into LDS once per workgroup and every K step reads its fragments from
// there -- a wave of this shape owns three cout tiles of only four pixel tiles, so its weight stream from L2 was 256 B per matrix
// instruction, 2.3 x that of the RRDBNet shapes -- and the epilogue is instantiated for ONE activation class XAC (Mish / SiLU / SPAB gate /
// linear) with its shape folded (conv_common.h, EM 4) instead of going through the generic dispatch (141 spilled scalar registers, 64 B of scratch).
template <int SHAPE, int UP, int OUTK, int HM = 0, int FMT = 0, int PROD = 3, int XRES = 0, int XAC = 0>
__global__ __launch_bounds__((8 + (SHAPE == 2 ? 2 : 1)) * 64, 3) void conv_ring(const rsa_conv_params p, const RingAux aux) {
  static_assert((XRES != 1 && XRES != 4 && XRES != 5) || (SHAPE == 1 && PROD == 1 && HM == 0 && UP == 0 && OUTK == 0), "XRES 1 / 4 / 5: conv5 of a residual dense block only");
  static_assert(XRES != 2 || (PROD == 1 && FMT == RSA_PF_F16 && OUTK == 0), "XRES 2: one fp16 product, hi-only plane output");
  static_assert(XRES != 3 || (SHAPE == 3 && PROD == 1 && FMT == RSA_PF_F16 && HM == 1 && UP == 0 && OUTK == 0), "XRES 3: the 48-channel SPAN-family layers");
  static_assert(XRES != 6 || (SHAPE == 2 && OUTK == 1 && UP == 0), "XRES 6: final stores with at most 16 output channels");
  constexpr bool WL = XRES == 3;
  constexpr bool XR = XRES == 1 || XRES == 4 || XRES == 5;  // XRES 4 = XRES 1 with every lo operand (residuals, output) as 8-bit codes (rsa_conv_params.lo8_flags);
                                                            // XRES 5: the residuals' lo halves as codes, the output's as fp16 (the last block of a trunk)
                                              // (XRES 2: the growth convolutions of a dense block -- hi-only fp16 plane output, LeakyReLU / none, nothing else:
                                  //  the kernel calls that one epilogue shape directly; without the generic dispatch and its dozen descriptor
                                  //  tests it keeps 40 fewer lane registers and 140 fewer scalar registers in scratch)
  using R = RingGeoP<PROD, WL>;
  constexpr int STREAMS = SHAPE == 2 ? 2 : 1;
  constexpr int TH = R::TH, TW = R::TW, IH = R::IH, IW = R::IW, PS = R::PS, HALF = R::HALF, SLOT = R::SLOT, NSLOT = R::NSLOT;
  constexpr int NHL = R::NHL, INFL = R::INFL;
  constexpr int WBASE = NSLOT * SLOT + R::FLAG_UNITS;  // WL: first unit of the resident weight blob
  constexpr int NCT = SHAPE == 2 ? 2 : (SHAPE == 1 ? 4 : 3);  // cout tiles of the layer handled by one workgroup
  constexpr int CTW = SHAPE == 3 ? 3 : 2;       // cout tiles per wave
  constexpr int NPT = SHAPE == 3 ? 4 : 8;       // pixel tiles per wave (RPW rows x 2 halves)
  // XRES 6 (round 4): a final store with at most 16 output channels (the 64 -> 3 last convolution) multiplies ONE cout tile; the two-tile form
  // issued as many MFMAs again on zero weights and was bound by them (0.92 TFLOP issued per 8.3 Mpx band in three products = its 835 us)
  constexpr int CTU = XRES == 6 ? 1 : CTW;      // cout tiles a wave multiplies
  constexpr int RPW = NPT / 2;
  constexpr int NCONS = (STREAMS == 2) ? 4 : 8; // consumer waves per slot
  constexpr int SPS = NSLOT / STREAMS;          // slots per stream (a power of two)

  // ring + flags in ONE shared array (a second __shared__ object can make hipcc drain vmcnt before LDS reads)
  __shared__ uint4 s_ring[NSLOT * SLOT + R::FLAG_UNITS + R::WL_FRAGS * 64];
  uint32_t* const flags = (uint32_t*)&s_ring[NSLOT * SLOT];  // [0..7] FULL, [8..15] FREE, [16] abort
  uint32_t* const f_full = flags;
  uint32_t* const f_free = flags + 8;
  uint32_t* const f_abort = flags + 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_x = (p.W + TW - 1) / TW;
  const int tiles_y = (p.H + TH - 1) / TH;
  const int num_tiles = tiles_x * tiles_y * p.batch;
  const int nhalf = p.cin_planes >> 1;    // half chunks of 16 channels (an even number of planes: checked by the launcher)
  const int nq = HM ? nhalf : (nhalf >> 1);  // units of work per tile: half chunks (HM 1) or whole 32-channel chunks
  const int ct_total = (p.cout + 15) >> 4;

  const int NWG = (int)gridDim.x;
  const int tile0 = (NWG % 8 == 0) ? ((int)blockIdx.x % 8) * (NWG / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;  // XCD strips, as conv_kernel
  if (tile0 >= num_tiles) return;  // whole workgroup
  const int ntw = (num_tiles - tile0 + NWG - 1) / NWG;  // tiles of this workgroup: tile0 + j*NWG

  if (tid < 4 * R::FLAG_UNITS) flags[tid] = 0;
  if (WL) {  // the layer's weight blob -> LDS, once per workgroup (the launcher has checked that it fits)
    const int wunits = nq * 5 * ct_total * 64;
    for (int i = tid; i < wunits; i += (int)blockDim.x) s_ring[WBASE + i] = ((const uint4*)p.w_packed)[i];
  }
  __syncthreads();  // the only workgroup barrier of the kernel
#ifdef RSA_RING_DEBUG
  const unsigned dbg = __builtin_amdgcn_readfirstlane(g_ring_dbg);
#endif

  if (wave >= 8) {
    // =========================== LOADER WAVE (one per stream) ===========================
    __builtin_amdgcn_s_setprio(3);
    const int g = wave - 8;
    const int inW = UP ? (p.W >> 1) : p.W;
    // per-lane constants.  lc: byte offset of the unit each DMA instruction of this lane delivers, from the source unit of the
    // tile's halo origin (halo row 0, column 0) in the half chunk's first plane -- never negative, so interior tiles address it as
    // (scalar base) + (32-bit lane offset).  smap: the same unit as packed (plane, halo row, halo column) for the border path.
    uint32_t lc[R::DMA_IT];
    uint32_t smap[R::DMA_IT];
#pragma unroll
    for (int it = 0; it < R::DMA_IT; ++it) {
      const int u = it * 64 + lane;
      const int pl = (u / PS) & 1;
      const int r = u % PS;
      int py = r / IW, px = r - (r / IW) * IW;
      if (r >= IH * IW) py = 0, px = 0;  // padding units of the plane stride: fetch the tile's first unit (never read by the multiply)
      smap[it] = (uint32_t)(pl << 16 | py << 8 | px);
      // UP: halo row py of a tile whose halo starts at output row y0 = 16*ty - 1 reads source row (y0 + py) >> 1 = (8*ty - 1) + ((py + 1) >> 1)
      const int sy = UP ? ((py + 1) >> 1) : py;
      const int sx = UP ? ((px + 1) >> 1) : px;
      lc[it] = (uint32_t)(((int64_t)pl * p.in_plane_stride + (int64_t)sy * inW + sx) * 16);  // < 2^32: the launcher bounds the plane size
    }
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)s_ring;
    int k = 0;                      // half-chunk counter of this stream
    // fills issued but not yet published, oldest first: pend[0] is published once everything but the INFL youngest fills has landed
    int pend_slot[INFL];
    uint32_t pend_val[INFL];
#pragma unroll
    for (int i = 0; i < INFL; ++i) pend_slot[i] = -1, pend_val[i] = 0;
    auto publish_all = [&]() {  // before blocking, and at the end: the consumers may need these fills to get to the release the loader waits for
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < INFL; ++i) {
        if (pend_slot[i] >= 0) __hip_atomic_store(&f_full[pend_slot[i]], pend_val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pend_slot[i] = -1;
      }
    };
    for (int j = (STREAMS == 2 ? g : 0); j < ntw; j += STREAMS) {
      int n, ty, tx;
      ring_tile_coords(p.tile_order ? num_tiles - 1 - (tile0 + j * NWG) : tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
      const int y0 = ty * TH - 1, x0 = tx * TW - 1;  // halo origin in OUTPUT coordinates
      const bool interior = y0 >= 0 && x0 >= 0 && y0 + IH <= p.H && x0 + IW <= p.W;
      // source unit of the halo origin in plane 0 of this image (used by interior tiles only, where it is inside the map)
      const int64_t tile_unit = (int64_t)n * p.in_batch_stride + (int64_t)(UP ? ty * (TH / 2) - 1 : y0) * inW + (UP ? tx * (TW / 2) - 1 : x0);
      for (int h = 0; h < nhalf; ++h, ++k) {
        const int slot = (STREAMS == 2) ? SPS * g + (k & (SPS - 1)) : (k & (NSLOT - 1));
        const uint32_t use = (uint32_t)(k / SPS);  // how many times this slot has been filled before
        // the slot must have been released by all its consumers `use` times
        if (__builtin_amdgcn_readfirstlane(lds_ld(&f_free[slot])) < NCONS * use) {
          bool any = false;
#pragma unroll
          for (int i = 0; i < INFL; ++i) any = any || pend_slot[i] >= 0;
          if (any) publish_all();
          ring_wait(&f_free[slot], NCONS * use, f_abort, aux);
        }
        const int hh = XR ? 2 * (nq - 1 - (h >> 1)) + (h & 1) : h;  // XRES: chunks from the last to the first (halves A, B of a chunk stay in order)
        const int64_t half_unit = tile_unit + (int64_t)(2 * hh) * p.in_plane_stride;  // first plane of this half chunk
        gcptr bh = uniform_ptr((gcptr)p.in_hi + half_unit * 16);
        gcptr bl = uniform_ptr((gcptr)p.in_lo + half_unit * 16);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * (SLOT * 16));
        if (RING_DBG(1)) {
        } else if (interior) {
#pragma unroll
          for (int it = 0; it < R::DMA_IT; ++it) {
            if (it == R::DMA_IT - 1 && lane >= 32) continue;  // the last instruction of a precision covers 32 units (EXEC-masked)
            dma16_s(dst + it * 1024, lc[it], bh);
            if (PROD == 3) dma16_s(dst + HALF * 16 + it * 1024, lc[it], bl);
          }
        } else {
#pragma unroll
          for (int it = 0; it < R::DMA_IT; ++it) {
            if (it == R::DMA_IT - 1 && lane >= 32) continue;
            const uint32_t m = smap[it];
            const int pl = (int)(m >> 16);
            int iy = y0 + (int)((m >> 8) & 255u), ix = x0 + (int)(m & 255u);
            const bool ok = (uint32_t)iy < (uint32_t)p.H && (uint32_t)ix < (uint32_t)p.W;
            if (UP) {
              iy >>= 1;
              ix >>= 1;
            }
            const int64_t off = ((int64_t)n * p.in_batch_stride + (int64_t)(2 * hh + pl) * p.in_plane_stride + (int64_t)iy * inW + ix) * 16;
            gcptr sh = ok ? (gcptr)p.in_hi + off : (gcptr)&g_zero_unit[0];
            dma16_v(dst + it * 1024, sh);
            if (PROD == 3) {
              gcptr sl = ok ? (gcptr)p.in_lo + off : (gcptr)&g_zero_unit[0];
              dma16_v(dst + HALF * 16 + it * 1024, sl);
            }
          }
        }
        if (pend_slot[0] >= 0) {
          // everything but the INFL fills issued last (this one included) has landed: publish the oldest pending fill
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL * R::DPF) : "memory");
          __hip_atomic_store(&f_full[pend_slot[0]], pend_val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int i = 0; i + 1 < INFL; ++i) pend_slot[i] = pend_slot[i + 1], pend_val[i] = pend_val[i + 1];
        pend_slot[INFL - 1] = slot;
        pend_val[INFL - 1] = use + 1;
      }
    }
    publish_all();
    return;
  }

  // =========================== COMPUTE WAVES ===========================
  const int g = (STREAMS == 2) ? (wave >> 2) : 0;    // stream
  const int wct = (SHAPE == 1) ? (wave >> 2) : 0;    // cout group (SHAPE 1)
  const int wpx = (SHAPE == 3) ? wave : (wave & 3);  // group of RPW rows
  const int li = lane & 15;
  const int lg = lane >> 4;
  const int hsel = lg >> 1;  // which tap of a pair / which half in the pairing step

  // weights: per (chunk, K step 0..8, cout tile, hi|lo) one 1 KiB A fragment, streamed from L2 WD K steps ahead
  constexpr int KSU = HM ? 5 : 9;  // K steps per unit
  constexpr int WD = (PROD == 3 || WL) ? 1 : (XRES ? RSA_RING_WDX : RSA_RING_WD1);  // K steps of weight prefetch (a one-product K step is 16 MFMAs = 256 cycles: less than an L2 hit under load)
  const int nks = nq * KSU;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_packed, 0, (uint32_t)((int64_t)nks * ct_total * NHL * 64 * 16), 0x00020000);
  uint32_t woff[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    const int ctg = wct * CTW + c;
    woff[c] = (ctg < ct_total) ? (uint32_t)((ctg * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;  // out of range -> zeros
  }
  const uint32_t wstep = (uint32_t)ct_total * NHL * 64 * 16;
  bf16x8 wq[WD + 1][CTW][NHL];  // wq[0]: the K step being multiplied; wq[1..WD]: the next ones, in flight
  auto load_w = [&](int s) {    // -> wq[WD]
#pragma unroll
    for (int c = 0; c < CTU; ++c)
#pragma unroll
      for (int hl = 0; hl < NHL; ++hl) {
        if (RING_DBG(4)) continue;
        if (WL) {  // resident blob: fragment (K step s, cout tile c) = 64 units
          wq[WD][c][hl] = *(const bf16x8*)&s_ring[WBASE + (s * ct_total + wct * CTW + c) * 64 + lane];
          continue;
        }
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)s * wstep + (uint32_t)hl * 1024u, 0);
        wq[WD][c][hl] = __builtin_bit_cast(bf16x8, v);
      }
  };
  auto shift_w = [&]() {
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int c = 0; c < CTU; ++c)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) wq[d][c][hl] = wq[d + 1][c][hl];
  };

  f32x4 acc[NPT][CTW];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment units (relative to the ring): lane (li, lg) reads plane lg & 1; pair steps read tap +hsel
  //   uA1: half A, taps (dy, 0 | 1)      + dy*IW by immediate        uA2: half A, taps (0 | 1, 2)
  //   uB1, uB2: the same for half B                                   uS : tap (2, 2) of half A (hsel 0) / half B (hsel 1)
  const int lane_u = (lg & 1) * PS + (wpx * RPW) * IW + li;
  int uA1, uA2, uB1, uB2, uS;  // set per chunk from the slots it lives in

  // De-phasing of the two waves that share a SIMD (stream 1 half a tile late / a lower priority for waves 0-3) measured +-0 on the
  // three-product kernels (profiles/r02_c): off unless a variant build asks for it.
#ifndef RSA_RING_STAGGER
#define RSA_RING_STAGGER 0
#endif
#ifndef RSA_RING_STAGGER_SLEEP
#define RSA_RING_STAGGER_SLEEP 100
#endif
#ifndef RSA_RING_PRIO_HI
#define RSA_RING_PRIO_HI 0  // s_setprio of waves 4-7 (the younger half of the workgroup: the arbitration loser on every SIMD at equal priority)
#endif
  if (RSA_RING_PRIO_HI && wave >= 4) __builtin_amdgcn_s_setprio(RSA_RING_PRIO_HI);
  if (RSA_RING_STAGGER) {
    // the second half of the workgroup (waves 4-7: the SIMD partners of waves 0-3) starts late, so that one partner's epilogue (vector
    // instructions, stores) falls into the other's multiply instead of both reaching the end of a tile together
    if (STREAMS == 2 ? g == 1 : wave >= 4)
      for (int d = 0; d < nq * RSA_RING_STAGGER; ++d) __builtin_amdgcn_s_sleep(RSA_RING_STAGGER_SLEEP);  // 64 cycles per unit
  }

  // blob K step of the step `koff` steps into unit c of the processing order (koff < 2 * KSU; past the last unit: the next tile's first ones)
  auto wmap = [&](int c, int koff) -> int {
    int cc = c + koff / KSU;
    if (cc >= nq) cc -= nq;
    return (XR ? nq - 1 - cc : cc) * KSU + koff % KSU;
  };
  // the first WD K steps' weights (steps wrap around: every tile of the layer reads the same blob)
#pragma unroll
  for (int d = 0; d < WD; ++d) {
    load_w(d < nks ? wmap(0, d) : wmap(0, 0));
    if (d + 1 < WD) {  // rotate so that step d sits in wq[d + 1] once all WD are issued
#pragma unroll
      for (int e = 1; e < WD; ++e)
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
          for (int hl = 0; hl < NHL; ++hl) wq[e][c][hl] = wq[e + 1][c][hl];
    }
  }
  uint32_t kc = 0;  // half chunks this stream has consumed: half chunk k lives in slot slot_of(k), which it is the use_of(k)-th to use
  auto slot_of = [&](uint32_t k) -> int { return (STREAMS == 2) ? SPS * g + (int)(k & (SPS - 1)) : (int)(k & (NSLOT - 1)); };
  auto use_of = [&](uint32_t k) -> uint32_t { return k / SPS; };
  uint32_t xslots = 0;  // XRES: ring slots of x's four half chunks (4 bits each)
  for (int j = (STREAMS == 2 ? g : 0); j < ntw; j += STREAMS) {
    for (int c = 0; c < nq; ++c) {
      const int sA = slot_of(kc), sB = HM ? sA : slot_of(kc + 1);
      const uint32_t needA = use_of(kc) + 1, needB = use_of(kc + 1) + 1;
      if (XR) {  // x = blob chunks 1 (half chunks 2, 3: unit nq - 2) and 0 (half chunks 0, 1: the last unit)
        if (c == nq - 2) xslots = (uint32_t)sA << 8 | (uint32_t)sB << 12;
        if (c == nq - 1) xslots |= (uint32_t)sA | (uint32_t)sB << 4;
      }
      {
        const int bA = sA * SLOT + lane_u, bB = sB * SLOT + lane_u;
        uA1 = bA + hsel;
        uA2 = bA + 2 + hsel * IW;
        uB1 = bB + hsel;
        uB2 = bB + 2 + hsel * IW;
        // the pairing step reads tap (2,2) of half A (hsel 0) and of half B (hsel 1); in half mode the hsel-1 lanes read half A
        // again and multiply zero weights (finite data: 0 * finite = 0)
        uS = (hsel ? bB : bA) + 2 * IW + 2;
      }
      ring_wait(&f_full[sA], needA, f_abort, aux);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");

      constexpr int NSTEP = KSU * NPT;  // K steps x pixel tiles of one unit
      constexpr int DEPTH = PROD == 3 ? 2 : RSA_RING_DEPTH1;  // pixel-tile steps of LDS prefetch (a one-product step is CTW MFMAs = 32-48 cycles)
      bf16x8 rh[DEPTH + 1], rl[PROD == 3 ? DEPTH + 1 : 1];
      auto frag = [&](int i) -> int {  // unit of pixel-tile step i (compile-time i)
        const int ks = i / NPT, pt = i % NPT;
        const int ptoff = (pt >> 1) * IW + (pt & 1) * 16;
        switch (ks) {
          case 0: return uA1 + ptoff;
          case 1: return uA1 + IW + ptoff;
          case 2: return uA1 + 2 * IW + ptoff;
          case 3: return uA2 + ptoff;
          case 4: return uS + ptoff;
          case 5: return uB1 + ptoff;
          case 6: return uB1 + IW + ptoff;
          case 7: return uB1 + 2 * IW + ptoff;
          default: return uB2 + ptoff;
        }
      };
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        rh[i] = *(const bf16x8*)&s_ring[frag(i)];
        if (PROD == 3) rl[i] = *(const bf16x8*)&s_ring[HALF + frag(i)];
      }
      // one pixel-tile step: prefetch the fragments of step i + DEPTH, multiply step i (i is a compile-time constant after unrolling)
      auto step = [&](int i) {
        const int ks = i / NPT, sp = i % NPT;
        if (sp == 0) {
          shift_w();
          load_w(wmap(c, ks + WD));  // the K step WD ahead; past the last one: the first steps of the next tile
          if (ks == 5) {
            // every read of half A has been consumed by an MFMA (the last ones in the step before): hand the slot back to the loader
            // (XRES: the slots of x's chunks -- the last two units -- stay until the epilogue has read the residual from them)
            asm volatile("" ::: "memory");
            if (lane == 0 && !(XR && c >= nq - 2)) __hip_atomic_fetch_add(&f_free[sA], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (i + DEPTH < NSTEP && !RING_DBG(16)) {
          const int u = frag(i + DEPTH);
          rh[(i + DEPTH) % (DEPTH + 1)] = *(const bf16x8*)&s_ring[u];
          if (PROD == 3) rl[(i + DEPTH) % (DEPTH + 1)] = *(const bf16x8*)&s_ring[HALF + u];
        }
        if (!RING_DBG(2))
#pragma unroll
        for (int pr = 0; pr < PROD; ++pr)
#pragma unroll
          for (int ct = 0; ct < CTU; ++ct) {
            // products in increasing magnitude: w_lo*a_hi, w_hi*a_lo, w_hi*a_hi
            const bf16x8 wf = (PROD == 3 && pr == 0) ? wq[0][ct][NHL - 1] : wq[0][ct][0];
            const bf16x8 bf = (PROD == 3 && pr == 1) ? rl[PROD == 3 ? i % (DEPTH + 1) : 0] : rh[i % (DEPTH + 1)];
            acc[sp][ct] = mfma16<FMT>(wf, bf, acc[sp][ct]);
          }
        if (i + DEPTH < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, NHL, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, PROD * CTU, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      if (HM) {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) step(i);
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&f_free[sA], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        kc += 1;
      } else {
        constexpr int FIRST_B = 4 * NPT - DEPTH;  // the step whose prefetch is the first read of the pairing step (half B)
#pragma unroll
        for (int i = 0; i < FIRST_B; ++i) step(i);
        ring_wait(&f_full[sB], needB, f_abort, aux);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = FIRST_B; i < NSTEP; ++i) step(i);
        asm volatile("" ::: "memory");
        if (lane == 0 && !(XR && c >= nq - 2)) __hip_atomic_fetch_add(&f_free[sB], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        kc += 2;
      }
    }
    // ---- tile finished: epilogue (the loader is already streaming the next tile; the wave sharing this SIMD keeps multiplying) ----
    {
      int n, ty, tx;
      ring_tile_coords(p.tile_order ? num_tiles - 1 - (tile0 + j * NWG) : tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
      if (XRES == 2) {
        if (!RING_DBG(8)) epilogue_impl<NCT, CTW, NPT, 0, AC_LINEAR, 1, RSA_PF_F16>(p, acc, n, ty * TH, tx * TW, 0, wct, wpx, li, lg);
      } else if (XRES == 3) {
        if (!RING_DBG(8)) epilogue_impl<NCT, CTW, NPT, 0, XAC, 4, RSA_PF_F16>(p, acc, n, ty * TH, tx * TW, 0, wct, wpx, li, lg);
      } else if (XR) {
        if (!RING_DBG(8)) {
          if (p.res2_hi != nullptr)
            epilogue_impl<NCT, CTW, NPT, 0, AC_LINEAR, 3, RSA_PF_F16, 1, XRES == 4 ? 1 : (XRES == 5 ? 2 : 0)>(p, acc, n, ty * TH, tx * TW, 0, wct, wpx, li, lg, s_ring, xslots, SLOT, PS, IW);
          else
            epilogue_impl<NCT, CTW, NPT, 0, AC_LINEAR, 2, RSA_PF_F16, 1, XRES == 4 ? 1 : (XRES == 5 ? 2 : 0)>(p, acc, n, ty * TH, tx * TW, 0, wct, wpx, li, lg, s_ring, xslots, SLOT, PS, IW);
        }
        // the residual's LDS reads have been consumed (their values are in the stores above): hand x's four slots back
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) __hip_atomic_fetch_add(&f_free[(xslots >> (4 * q)) & 15u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else if (!RING_DBG(8)) {
        epilogue<NCT, CTW, NPT, OUTK>(p, acc, n, ty * TH, tx * TW, 0, wct, wpx, li, lg);
      }
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

template <int SHAPE, int UP, int OUTK, int HM = 0, int FMT = 0, int PROD = 3, int XRES = 0, int XAC = 0>
static int launch_ring(const rsa_conv_params& p, hipStream_t stream) {
  constexpr int STREAMS = SHAPE == 2 ? 2 : 1;
  using R = RingGeoP<PROD>;
  const int tiles_x = (p.W + R::TW - 1) / R::TW;
  const int tiles_y = (p.H + R::TH - 1) / R::TH;
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
  hipLaunchKernelGGL((conv_ring<SHAPE, UP, OUTK, HM, FMT, PROD, XRES, XAC>), dim3((unsigned)gx, 1, 1), dim3((8 + STREAMS) * 64), 0, stream, p, ring_aux());
  return (int)hipGetLastError();
}

}  // namespace rsa
