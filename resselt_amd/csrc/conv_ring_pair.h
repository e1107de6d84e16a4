// conv_ring_pair.h — TWO consecutive growth convolutions of a residual dense block in ONE launch (cross-layer fusion).
//
// Reference: ResidualDenseBlock_5C.forward, utilities/block.py:454-465 of rewaifu/resselt:
//     x1 = lrelu(conv1(x));  x2 = lrelu(conv2(cat(x, x1)))      and      x3 = lrelu(conv3(cat(x, x1, x2)));  x4 = lrelu(conv4(cat(x .. x3)))
// Layer by layer (conv_ring.h) the pair (A, B) reads its common input planes twice and A's output once more: 128 + 192 B per pixel
// for (conv1, conv2), 256 + 320 B for (conv3, conv4) in the one-product fp16 layout.  Here the input is streamed through the LDS ring
// ONCE and both layers consume every ring slot:
//
//   * output tile of B: 16 rows x 30 pixels; A is computed on the 18 x 32 region B's taps reach (its own halo: the ring slot holds
//     20 x 34 pixels of 16 channels).  30, not 32: A's region has to be whole 16-pixel MFMA tiles, so B's rightmost two columns of
//     each 32-pixel row of tiles are computed and dropped (recompute: A x 1.20, B x 1.07; 1.12 x the pair's multiply-accumulates);
//   * waves 0-3 multiply A (9 pixel tiles x 2 cout tiles each: 4 rows x 2 halves + one of the four tiles of rows 16, 17), waves 4-7
//     multiply B's partial sum over the SAME slots (8 pixel tiles x 2 cout tiles: conv_ring.h's two-cout-tile wave shape) -- a SIMD
//     hosts one A wave and one B wave;
//   * at the end of a tile the A waves apply bias + LeakyReLU, round to fp16 and write x_A twice: to memory (the 16 x 30 interior: the
//     later layers of the block read it) and to a 36 KB LDS image of the whole 18 x 32 region with ZEROS outside the picture (B's zero
//     padding); the B waves then run B's last 32 input channels (nine K steps) out of that image and store their tile;
//   * every accumulator sees the K steps of its layer in the order conv_ring.h runs them, so the fused pair is bit-identical to the
//     two launches it replaces (tests/test_conv_pair_gpu.py).
//
// Hand-offs are LDS words as in conv_ring.h (FULL / FREE per ring slot with all eight compute waves as consumers; XFULL / XFREE for the
// x_A image), every spin bounded, a timed-out one reported through the same failure word.  Five ring slots of 21.5 KB + the image = 144 KB.
//
// Bytes per pixel of a residual dense block (one fp16 product): conv1..conv4 layer by layer 1152 B -> fused 640 B (+ halo re-reads).
#pragma once
#include <type_traits>

#include "conv_ring.h"

namespace rsa {

// what the launcher derives from the two descriptors (conv_pair_eligible has checked that they form a pair)
struct PairParams {
  int32_t batch, H, W;
  int32_t nqa;  // 32-channel chunks of A's input (B reads one more: A's output)
  const void* in_hi;
  int64_t in_plane_stride, in_batch_stride;  // 16-byte units
  const void* wa;  // packed weights, layout RSA_WL_PAIRS, fp16, one product: A [nqa][9][2][64][8], B [nqa + 1][9][2][64][8]
  const void* wb;
  const float* bias_a;  // f32[32] (or NULL)
  const float* bias_b;
  float slope_a, slope_b;  // LeakyReLU slope in [0, 1]; 1 = no activation
  void* outa_hi;
  int64_t outa_unit0;  // first output plane, in units from outa_hi
  int64_t outa_plane_stride, outa_batch_stride;
  void* outb_hi;
  int64_t outb_unit0;
  int64_t outb_plane_stride, outb_batch_stride;
  int32_t tile_order;
};

#ifndef RSA_PAIR_INFL
#define RSA_PAIR_INFL 1  // fills left in flight behind the one being published
#endif
#ifndef RSA_PAIR_WD
#define RSA_PAIR_WD 3  // K steps of weight prefetch
#endif
#ifndef RSA_PAIR_DEPTH
#define RSA_PAIR_DEPTH 3  // pixel-tile steps of LDS fragment prefetch (3 / 4 / 6: 224 / 235 / 297 us on the conv1+conv2 pair at 1080p, profiles/r04_b_*)
#endif
#ifndef RSA_PAIR_PRIO_B
#define RSA_PAIR_PRIO_B 1  // s_setprio of the layer-B waves (4-7: the younger half of the workgroup, and the longer chain per tile)
#endif
#ifndef RSA_PAIR_PRIO_A
#define RSA_PAIR_PRIO_A 0
#endif

#ifdef RSA_PAIR_STAMPS
// diagnostic build only (tools/variant.sh pair_stamps "-DRSA_PAIR_STAMPS"): per-wave cycle totals of the kernel's phases, written to memory no
// other code reads.  [workgroup][wave 0..8][slot]: 0 total, 1 waiting for ring fills (FULL) / for FREE slots (loader), 2 waiting for the
// x_A image hand-offs (XFULL / XFREE), 3 epilogue, 4 x_A unit of layer B, 5 tiles
static __device__ unsigned long long g_pair_stamps[256 * 9 * 8];
#define PSTAMP() __builtin_amdgcn_s_memtime()
#define PST_ADD(slot, t0) st[slot] += PSTAMP() - (t0)
#else
#define PSTAMP() 0ull
#define PST_ADD(slot, t0) (void)(t0)
#endif

struct PairGeo {
  static constexpr int TH = 16, TWO = 30;    // B's output tile
  static constexpr int AH = 18, AW = 32;     // A's region = the x_A image (rows -1 .. 16, columns -1 .. 30 of the tile)
  static constexpr int IH = 20, IW = 34;     // halo tile of the input in a ring slot (rows -2 .. 17, columns -2 .. 31)
  static constexpr int PS = 688;             // plane stride of a slot in units (IH * IW = 680 rounded up to 0 mod 16)
  static constexpr int SLOT = 2 * PS;        // a slot = one 16-channel half chunk, hi planes only
  static constexpr int NSLOT = 5;
  static constexpr int DMA_IT = (SLOT + 63) / 64;  // 22 LDS-DMA instructions per fill; the last one covers 32 units
  static constexpr int INFL = RSA_PAIR_INFL;
  static constexpr int PSB = AH * AW;        // 576 units per plane of the x_A image (0 mod 16)
  static constexpr int XB0 = NSLOT * SLOT;   // first unit of the image
  static constexpr int XB_UNITS = 4 * PSB + 16;  // + slack: the dropped columns of B read up to 2 units past a plane's last row
  static constexpr int FLAG_UNITS = 4;       // FULL[5], FREE[5], abort, XFULL, XFREE
  static constexpr int LDS_UNITS = XB0 + XB_UNITS + FLAG_UNITS;
  static_assert(SLOT % 64 == 32, "the last LDS-DMA instruction of a fill covers 32 units");
  static_assert(INFL * DMA_IT <= 63, "vmcnt");
  static_assert(LDS_UNITS * 16 <= 160 * 1024, "LDS");
};

// (a, b) -> packed fp16 pair (RNE; a value beyond +-65504 becomes an infinity, as in every fp16 epilogue of the library: it then travels
// with the residual stream to the end of the network, where rsa_check_finite finds it -- see include/resselt_amd.h)
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
  const f16x2 h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, h);
}

// two D fragments (the same 16 output channels of two pixel tiles) after bias + LeakyReLU -> this lane's 16-byte fp16 unit:
// even lane groups end up with the unit of tile `a`, odd ones with that of tile `b` (plane 2*ct + (lg >> 1) of the layer's output)
__device__ __forceinline__ uint4 pair_unit_f16(const f32x4 a, const f32x4 b, const f32x4 bias, float slope) {
  float va[4], vb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    va[r] = a[r] + bias[r];
    vb[r] = b[r] + bias[r];
    va[r] = fmaxf(va[r], va[r] * slope);
    vb[r] = fmaxf(vb[r], vb[r] * slope);
  }
  const uint32_t a0 = pack_f16(va[0], va[1]), a1 = pack_f16(va[2], va[3]);
  const uint32_t b0 = pack_f16(vb[0], vb[1]), b1 = pack_f16(vb[2], vb[3]);
  const u32x2 h0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
  const u32x2 h1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
  return make_uint4(h0.x, h1.x, h0.y, h1.y);
}

// The compute waves of conv_ring_pair.  ROLE 0 = layer A (waves 0-3: nine pixel tiles x two cout tiles on the 18 x 32 region), ROLE 1 = layer B
// (waves 4-7: eight pixel tiles x two cout tiles on the 16 x 32 tile whose first 30 columns are stored).
template <int ROLE>
__device__ __forceinline__ void pair_consumer(const PairParams& p, const RingAux& aux, uint4* s_ring, uint32_t* flags, int wave, int lane, int tile0, int ntw,
                                              int NWG, int tiles_x, int tiles_y, int num_tiles) {
  using G = PairGeo;
  constexpr int IW = G::IW, PS = G::PS, SLOT = G::SLOT, NSLOT = G::NSLOT, AW = G::AW, PSB = G::PSB, XB0 = G::XB0;
  constexpr int CTW = 2, KSU = 9;
  constexpr int WD = RSA_PAIR_WD, DEPTH = RSA_PAIR_DEPTH;
  constexpr bool roleB = ROLE == 1;
  constexpr int NPT = roleB ? 8 : 9;
  uint32_t* const f_full = flags;
  uint32_t* const f_free = flags + 5;
  uint32_t* const f_abort = flags + 10;
  uint32_t* const f_xfull = flags + 11;
  uint32_t* const f_xfree = flags + 12;
  const int nqa = p.nqa;
#ifdef RSA_RING_DEBUG
  const unsigned dbg = __builtin_amdgcn_readfirstlane(g_ring_dbg);
#endif
  if (roleB && RSA_PAIR_PRIO_B) __builtin_amdgcn_s_setprio(RSA_PAIR_PRIO_B);
  if (!roleB && RSA_PAIR_PRIO_A) __builtin_amdgcn_s_setprio(RSA_PAIR_PRIO_A);
  const int r4 = wave & 3;  // group of four rows
  const int li = lane & 15;
  const int lg = lane >> 4;
  const int hsel = lg >> 1;

  // weights: [unit][K step 0..8][cout tile 0..1] 1 KiB A fragments, streamed from L2 WD K steps ahead (A: nqa units; B: nqa + 1, the
  // last one = A's output channels)
  const int nunits = roleB ? nqa + 1 : nqa;
  const int nks = nunits * KSU;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(roleB ? p.wb : p.wa), 0, (uint32_t)(nks * CTW * 64 * 16), 0x00020000);
  uint32_t woff[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) woff[c] = (uint32_t)((c * 64 + lane) * 16);
  constexpr uint32_t wstep = CTW * 64 * 16;
  bf16x8 wq[WD + 1][CTW];
  auto load_w = [&](int s) {  // -> wq[WD]
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      if (RING_DBG(4)) continue;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, woff[c], (uint32_t)s * wstep, 0);
      wq[WD][c] = __builtin_bit_cast(bf16x8, v);
    }
  };
  auto shift_w = [&]() {
#pragma unroll
    for (int d = 0; d < WD; ++d)
#pragma unroll
      for (int c = 0; c < CTW; ++c) wq[d][c] = wq[d + 1][c];
  };
  auto wmap = [&](int c, int koff) -> int {  // blob K step `koff` steps into unit c (past the last unit: the next tile's first ones)
    int cc = c + koff / KSU;
    if (cc >= nunits) cc -= nunits;
    return cc * KSU + koff % KSU;
  };
#pragma unroll
  for (int d = 0; d < WD; ++d) {
    load_w(d < nks ? wmap(0, d) : wmap(0, 0));
    if (d + 1 < WD) {
#pragma unroll
      for (int e = 1; e < WD; ++e)
#pragma unroll
        for (int c = 0; c < CTW; ++c) wq[e][c] = wq[e + 1][c];
    }
  }

  f32x4 acc[NPT][CTW];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment unit inside a ring slot: lane (li, lg) reads plane lg & 1; pixel tile pt < 8 = (row 4*r4 + (pt >> 1), half pt & 1)
  //   A: region pixel (ra, ca), tap (dy, dx) -> halo (ra + dy, ca + dx)                 B: tile pixel (rb, cb) -> halo (rb + dy + 1, cb + dx + 1)
  const int lane_u = (lg & 1) * PS + (4 * r4 + (roleB ? 1 : 0)) * IW + li + (roleB ? 1 : 0);
  // A's ninth pixel tile: (row 16 + (r4 >> 1), half r4 & 1), as a wave-uniform distance from pixel tile 0
  const int extra_u = __builtin_amdgcn_readfirstlane((16 + (r4 >> 1) - 4 * r4) * IW + (r4 & 1) * 16);
  // ... inside the x_A image (B's last unit): image pixel (rb + dy, cb + dx), row stride AW
  const int lane_ub = XB0 + (lg & 1) * PSB + (4 * r4) * AW + li;

  int slot = 0;
  uint32_t use = 0;
  uint32_t tcount = 0;  // tiles this workgroup has finished
#ifdef RSA_PAIR_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = PSTAMP();
#endif

  // One unit of nine K steps over the two 16-channel halves at units bA / bB (RS = their row stride).  RING: the halves are ring slots
  // sA / sB -- wait for their fills, hand them back after their last read.
  auto run_unit = [&](auto rs_tag, auto ring_tag, int c, int bA, int bB, int sA, int sB, uint32_t needA, uint32_t needB) {
    constexpr int RS = decltype(rs_tag)::value;
    constexpr bool RING = decltype(ring_tag)::value;
    constexpr int NSTEP = KSU * NPT;
    const int uA1 = bA + hsel, uA2 = bA + 2 + hsel * RS, uB1 = bB + hsel, uB2 = bB + 2 + hsel * RS;
    const int uS = (hsel ? bB : bA) + 2 * RS + 2;
    {
      const unsigned long long tw = PSTAMP();
      if (RING) ring_wait(&f_full[sA], needA, f_abort, aux);
      if (RING) PST_ADD(1, tw);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    bf16x8 rh[DEPTH + 1];
    auto frag = [&](int i) -> int {  // unit of pixel-tile step i (compile-time i)
      const int ks = i / NPT, pt = i % NPT;
      const int ptoff = pt < 8 ? (pt >> 1) * RS + (pt & 1) * 16 : extra_u;
      switch (ks) {
        case 0: return uA1 + ptoff;
        case 1: return uA1 + RS + ptoff;
        case 2: return uA1 + 2 * RS + ptoff;
        case 3: return uA2 + ptoff;
        case 4: return uS + ptoff;
        case 5: return uB1 + ptoff;
        case 6: return uB1 + RS + ptoff;
        case 7: return uB1 + 2 * RS + ptoff;
        default: return uB2 + ptoff;
      }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) rh[i] = *(const bf16x8*)&s_ring[frag(i)];
    auto step = [&](int i) {
      const int ks = i / NPT, sp = i % NPT;
      if (sp == 0) {
        shift_w();
        load_w(wmap(c, ks + WD));
        if (RING && ks == 5) {
          // every read of half A has been consumed by an MFMA: hand the slot back to the loader
          asm volatile("" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(&f_free[sA], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i + DEPTH < NSTEP && !RING_DBG(16)) rh[(i + DEPTH) % (DEPTH + 1)] = *(const bf16x8*)&s_ring[frag(i + DEPTH)];
      if (!RING_DBG(2))
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) acc[sp][ct] = mfma16<RSA_PF_F16>(wq[0][ct], rh[i % (DEPTH + 1)], acc[sp][ct]);
      if (i + DEPTH < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, CTW, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int FIRST_B = 4 * NPT - DEPTH;  // the step whose prefetch is the first read of the pairing step (half B)
#pragma unroll
    for (int i = 0; i < FIRST_B; ++i) step(i);
    {
      const unsigned long long tw = PSTAMP();
      if (RING) ring_wait(&f_full[sB], needB, f_abort, aux);
      if (RING) PST_ADD(1, tw);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = FIRST_B; i < NSTEP; ++i) step(i);
    asm volatile("" ::: "memory");
    if (RING && lane == 0) __hip_atomic_fetch_add(&f_free[sB], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  using std::integral_constant;

  for (int j = 0; j < ntw; ++j) {
    for (int c = 0; c < nqa; ++c) {
      const int sA = slot;
      const uint32_t needA = use + 1;
      if (++slot == NSLOT) slot = 0, ++use;
      const int sB = slot;
      const uint32_t needB = use + 1;
      if (++slot == NSLOT) slot = 0, ++use;
      run_unit(integral_constant<int, IW>{}, integral_constant<bool, true>{}, c, sA * SLOT + lane_u, sB * SLOT + lane_u, sA, sB, needA, needB);
    }
    int n, ty, tx;
    ring_tile_coords(p.tile_order ? num_tiles - 1 - (tile0 + j * NWG) : tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
    const int y0 = ty * G::TH, x0 = tx * G::TWO;  // B's tile origin
    int lli = li, llg = lg;
    asm volatile("" : "+v"(lli), "+v"(llg));  // tile-local address arithmetic (see conv_common.h: hoisted 64-bit lane addresses spill)
    const int plane_l = llg >> 1;             // + 2 * ct: the output plane this lane's unit belongs to
    const int col = 16 * (llg & 1) + lli;     // column of this lane's unit inside a 32-pixel row of tiles
    if (!roleB) {
      // ---- A: bias + LeakyReLU, fp16; the 18 x 32 region -> the LDS image (zeros outside the picture), its 16 x 30 interior -> memory ----
      const unsigned long long tx0 = PSTAMP();
      ring_wait(f_xfree, 4u * tcount, f_abort, aux);  // the B waves have finished with the previous tile's image
      PST_ADD(2, tx0);
      const unsigned long long te0 = PSTAMP();
      char* const ob = (char*)p.outa_hi + (p.outa_unit0 + (int64_t)n * p.outa_batch_stride + (int64_t)(y0 - 1) * p.W + (x0 - 1)) * 16;
      const int x = x0 - 1 + col;
      const bool xin = (uint32_t)x < (uint32_t)p.W;
      const bool xst = xin && col >= 1 && col <= G::TWO;
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        if (RING_DBG(8)) break;
        const f32x4 bias = p.bias_a != nullptr ? ((const f32x4*)p.bias_a)[ct * 4 + llg] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint32_t pl_off = (uint32_t)(2 * ct + plane_l);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          uint4 u = pair_unit_f16(acc[2 * pp][ct], acc[2 * pp + 1][ct], bias, p.slope_a);
          const int ra = 4 * r4 + pp;
          const int y = y0 - 1 + ra;
          const bool yin = (uint32_t)y < (uint32_t)p.H;
          if (!(yin && xin)) u = make_uint4(0u, 0u, 0u, 0u);
          s_ring[XB0 + pl_off * PSB + ra * AW + col] = u;
          if (yin && xst && ra >= 1) *(uint4*)(ob + (pl_off * (uint32_t)p.outa_plane_stride + (uint32_t)(ra * p.W + col)) * 16u) = u;
          __builtin_amdgcn_sched_barrier(0);
        }
        {  // the ninth pixel tile: row 16 + (r4 >> 1), half r4 & 1; both lane groups of a pair hold its unit, the even one stores it
          uint4 u = pair_unit_f16(acc[NPT - 1][ct], acc[NPT - 1][ct], bias, p.slope_a);
          const int ra = 16 + (r4 >> 1);
          const int ecol = 16 * (r4 & 1) + lli;
          const int y = y0 - 1 + ra, xe = x0 - 1 + ecol;
          const bool yin = (uint32_t)y < (uint32_t)p.H, xein = (uint32_t)xe < (uint32_t)p.W;
          if (!(yin && xein)) u = make_uint4(0u, 0u, 0u, 0u);
          if ((llg & 1) == 0) {
            s_ring[XB0 + pl_off * PSB + ra * AW + ecol] = u;
            if (yin && xein && ra <= 16 && ecol >= 1 && ecol <= G::TWO)
              *(uint4*)(ob + (pl_off * (uint32_t)p.outa_plane_stride + (uint32_t)(ra * p.W + ecol)) * 16u) = u;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image writes of this wave have landed
      if (lane == 0) __hip_atomic_fetch_add(f_xfull, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      PST_ADD(3, te0);
    } else {
      // ---- B: the last 32 input channels = A's output, out of the LDS image; then bias + LeakyReLU and the 16 x 30 tile -> memory ----
      const unsigned long long tx0 = PSTAMP();
      ring_wait(f_xfull, 4u * (tcount + 1u), f_abort, aux);
      PST_ADD(2, tx0);
      const unsigned long long tu0 = PSTAMP();
      if (!RING_DBG(32)) run_unit(integral_constant<int, AW>{}, integral_constant<bool, false>{}, nqa, lane_ub, lane_ub + 2 * PSB, 0, 0, 0u, 0u);
      if (lane == 0) __hip_atomic_fetch_add(f_xfree, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      PST_ADD(4, tu0);
      const unsigned long long te0 = PSTAMP();
      char* const ob = (char*)p.outb_hi + (p.outb_unit0 + (int64_t)n * p.outb_batch_stride + (int64_t)y0 * p.W + x0) * 16;
      const bool xst = x0 + col < p.W && col < G::TWO;
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        if (RING_DBG(8)) break;
        const f32x4 bias = p.bias_b != nullptr ? ((const f32x4*)p.bias_b)[ct * 4 + llg] : (f32x4){0.f, 0.f, 0.f, 0.f};
        const uint32_t pl_off = (uint32_t)(2 * ct + plane_l);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          const uint4 u = pair_unit_f16(acc[2 * pp][ct], acc[2 * pp + 1][ct], bias, p.slope_b);
          const int rb = 4 * r4 + pp;
          if (y0 + rb < p.H && xst) *(uint4*)(ob + (pl_off * (uint32_t)p.outb_plane_stride + (uint32_t)(rb * p.W + col)) * 16u) = u;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      PST_ADD(3, te0);
    }
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) acc[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ++tcount;
  }
#ifdef RSA_PAIR_STAMPS
  st[0] = PSTAMP() - t_begin;
  st[5] = tcount;
  if (lane == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) g_pair_stamps[((int)blockIdx.x * 9 + wave) * 8 + i] = st[i];
#endif
}


__global__ __launch_bounds__(9 * 64, 3) void conv_ring_pair(const PairParams p, const RingAux aux) {
  using G = PairGeo;
  constexpr int IW = G::IW, PS = G::PS, SLOT = G::SLOT, NSLOT = G::NSLOT, AW = G::AW, PSB = G::PSB, XB0 = G::XB0, INFL = G::INFL;
  constexpr int NCONS = 8;  // consumer waves of a ring slot
  constexpr int CTW = 2, KSU = 9;
  constexpr int WD = RSA_PAIR_WD, DEPTH = RSA_PAIR_DEPTH;

  __shared__ uint4 s_ring[G::LDS_UNITS];
  uint32_t* const flags = (uint32_t*)&s_ring[XB0 + G::XB_UNITS];
  uint32_t* const f_full = flags;       // [0..4]
  uint32_t* const f_free = flags + 5;   // [5..9]
  uint32_t* const f_abort = flags + 10;
  uint32_t* const f_xfull = flags + 11;  // A waves that have written their part of the x_A image (4 per tile)
  uint32_t* const f_xfree = flags + 12;  // B waves that have finished reading it (4 per tile)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_x = (p.W + G::TWO - 1) / G::TWO;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int num_tiles = tiles_x * tiles_y * p.batch;
  const int nqa = p.nqa;
  const int nhalf = 2 * nqa;

  const int NWG = (int)gridDim.x;
  const int tile0 = (NWG % 8 == 0) ? ((int)blockIdx.x % 8) * (NWG / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;  // XCD strips, as conv_ring
  if (tile0 >= num_tiles) return;  // whole workgroup
  const int ntw = (num_tiles - tile0 + NWG - 1) / NWG;

  if (tid < 4 * G::FLAG_UNITS) flags[tid] = 0;
  __syncthreads();  // the only workgroup barrier of the kernel
#ifdef RSA_RING_DEBUG
  const unsigned dbg = __builtin_amdgcn_readfirstlane(g_ring_dbg);
#endif

  if (wave == 8) {
    // =========================== LOADER WAVE ===========================
    __builtin_amdgcn_s_setprio(3);
    uint32_t lc[G::DMA_IT];    // byte offset of the unit each LDS-DMA instruction of this lane delivers, from the tile's halo origin in the half chunk's first plane
    uint32_t smap[G::DMA_IT];  // the same unit as packed (plane, halo row, halo column): border tiles
#pragma unroll
    for (int it = 0; it < G::DMA_IT; ++it) {
      const int u = it * 64 + lane;
      const int pl = (u / PS) & 1;
      const int r = u % PS;
      int py = r / IW, px = r - (r / IW) * IW;
      if (r >= G::IH * IW) py = 0, px = 0;  // padding units of the plane stride (never read by the multiply)
      smap[it] = (uint32_t)(pl << 16 | py << 8 | px);
      lc[it] = (uint32_t)(((int64_t)pl * p.in_plane_stride + (int64_t)py * p.W + px) * 16);
    }
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)s_ring;
    int slot = 0;
    uint32_t use = 0;  // how many times `slot` has been filled before
    int pend_slot[INFL];
    uint32_t pend_val[INFL];
#pragma unroll
    for (int i = 0; i < INFL; ++i) pend_slot[i] = -1, pend_val[i] = 0;
    auto publish_all = [&]() {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < INFL; ++i) {
        if (pend_slot[i] >= 0) __hip_atomic_store(&f_full[pend_slot[i]], pend_val[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pend_slot[i] = -1;
      }
    };
#ifdef RSA_PAIR_STAMPS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = PSTAMP();
#endif
    for (int j = 0; j < ntw; ++j) {
      int n, ty, tx;
      ring_tile_coords(p.tile_order ? num_tiles - 1 - (tile0 + j * NWG) : tile0 + j * NWG, tiles_x, tiles_y, n, ty, tx);
      const int y0 = ty * G::TH - 2, x0 = tx * G::TWO - 2;  // halo origin
      const bool interior = y0 >= 0 && x0 >= 0 && y0 + G::IH <= p.H && x0 + IW <= p.W;
      const int64_t tile_unit = (int64_t)n * p.in_batch_stride + (int64_t)y0 * p.W + x0;
      for (int h = 0; h < nhalf; ++h) {
        if (__builtin_amdgcn_readfirstlane(lds_ld(&f_free[slot])) < NCONS * use) {
          bool any = false;
#pragma unroll
          for (int i = 0; i < INFL; ++i) any = any || pend_slot[i] >= 0;
          const unsigned long long tw = PSTAMP();
          if (any) publish_all();  // the consumers may need these fills to reach the release this wave is about to wait for
          ring_wait(&f_free[slot], NCONS * use, f_abort, aux);
          PST_ADD(1, tw);
        }
        const int64_t half_unit = tile_unit + (int64_t)(2 * h) * p.in_plane_stride;
        gcptr bh = uniform_ptr((gcptr)p.in_hi + half_unit * 16);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)slot * (SLOT * 16));
        if (RING_DBG(1)) {
        } else if (interior) {
#pragma unroll
          for (int it = 0; it < G::DMA_IT; ++it) {
            if (it == G::DMA_IT - 1 && lane >= 32) continue;
            dma16_s(dst + it * 1024, lc[it], bh);
          }
        } else {
#pragma unroll
          for (int it = 0; it < G::DMA_IT; ++it) {
            if (it == G::DMA_IT - 1 && lane >= 32) continue;
            const uint32_t m = smap[it];
            const int pl = (int)(m >> 16);
            const int iy = y0 + (int)((m >> 8) & 255u), ix = x0 + (int)(m & 255u);
            const bool ok = (uint32_t)iy < (uint32_t)p.H && (uint32_t)ix < (uint32_t)p.W;
            const int64_t off = ((int64_t)n * p.in_batch_stride + (int64_t)(2 * h + pl) * p.in_plane_stride + (int64_t)iy * p.W + ix) * 16;
            gcptr sh = ok ? (gcptr)p.in_hi + off : (gcptr)&g_zero_unit[0];
            dma16_v(dst + it * 1024, sh);
          }
        }
        if (pend_slot[0] >= 0) {
          const unsigned long long tw = PSTAMP();
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL * G::DMA_IT) : "memory");
          PST_ADD(2, tw);  // loader: waiting for an older fill to land
          __hip_atomic_store(&f_full[pend_slot[0]], pend_val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int i = 0; i + 1 < INFL; ++i) pend_slot[i] = pend_slot[i + 1], pend_val[i] = pend_val[i + 1];
        pend_slot[INFL - 1] = slot;
        pend_val[INFL - 1] = use + 1;
        if (++slot == NSLOT) slot = 0, ++use;
      }
    }
    publish_all();
#ifdef RSA_PAIR_STAMPS
    st[0] = PSTAMP() - t_begin;
    if (lane == 0 && blockIdx.x < 256)
      for (int i = 0; i < 8; ++i) g_pair_stamps[((int)blockIdx.x * 9 + 8) * 8 + i] = st[i];
#endif
    return;
  }

  // =========================== COMPUTE WAVES ===========================
  // (two separate instantiations: with both roles in one loop nest the compiler keeps both accumulator sets live and spills them)
  if (wave >= 4)
    pair_consumer<1>(p, aux, s_ring, flags, wave, lane, tile0, ntw, NWG, tiles_x, tiles_y, num_tiles);
  else
    pair_consumer<0>(p, aux, s_ring, flags, wave, lane, tile0, ntw, NWG, tiles_x, tiles_y, num_tiles);
}

}  // namespace rsa
