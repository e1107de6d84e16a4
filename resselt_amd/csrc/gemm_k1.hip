// gemm_k1.hip — weight-stationary schedule for 1x1 convolutions with many output channels (nn.Linear over tokens:
// SwinIR qkv / proj / fc1 / fc2, resselt/archs/swinir/arch.py:34-40,141,168).
//
// A k1 layer has no tap reuse: the halo-tile schedules of conv_mfma.hip refill LDS for every 48-96 MFMAs and end up bound by
// the LDS-DMA landing rate (profiles/r01_*: ~140 TFLOP/s issued).  Here the roles are swapped:
//   * every wave keeps the A fragments (weights) of ITS cout tiles for the WHOLE reduction in registers
//     (CTW cout tiles x NQ K-chunks x hi/lo = 128 VGPRs), loaded once per pass over the pixels;
//   * pixels stream through LDS in tiles of 64 consecutive tokens x all input planes, double-buffered; all 8 waves issue the
//     LDS-DMA of tile i+1 (each DMA instruction = one (plane, 64-pixel) row = 1 KiB contiguous in HBM), then multiply tile i;
//     no other vector-memory instruction is issued in between, so the in-order vmcnt never drains the prefetch early;
//   * one workgroup barrier per tile; 8 waves x CTW cout tiles = up to 256 output channels per pass (wider layers take
//     ceil(cout / (128*CTW)) passes over the token map).
#include <atomic>

#include "conv_common.h"

namespace rsa {

__device__ uint4 g_zero_unit_gk[4];  // source of zero units for lanes past the end of the token map (never written)

constexpr int GK_WAVES = 8;
#ifndef RSA_GK_ABL
#define RSA_GK_ABL 0  // experiment builds (tools/variant.sh): 1 = tiles after the first two are not fetched, 2 = no plane / map stores, 4 = no MFMAs
#endif
#ifndef GK_MINW
#define GK_MINW(PROD, NQ) (((PROD) == 1 && (NQ) == 6) ? 4 : 2)  // waves per SIMD the registers must allow (NQ 6: see gemm_k1_launch).  The 64 KB one-product form takes 157 VGPRs = ONE workgroup per CU; forcing 128
                             // (4: two workgroups per CU, 26 spills) is 6-9 % slower on DAT / HAT / DRCT (profiles/r04_zz_gemm_k1_occupancy_ab.txt)
#endif

// TP = pixels per tile (64, or 32 when the whole-K image of 64 pixels would not fit twice in LDS)
// FMT: plane format of the input planes and weights (selects the matrix instruction); the outputs follow p.out_fmt.  Round 3: one fp16
// product on hi planes (PROD 1, FMT RSA_PF_F16) -- the Linear layers of DRCT / HAT / DAT under their 'mixed' precision policies.
// EPI (round 4): 0 = the generic epilogue (activation classes, f32 residual maps, f32 / plane outputs, PReLU, the SPAB gate: a dozen descriptor
// tests per fragment pair); 1 / 2 = the direct form of the layers that only write fp16 hi planes -- qkv (no activation) and fc1 (GELU) of
// the transformer bodies: bias, [GELU], fp16 convert, half exchange, one 16-byte store per lane and fragment pair; 3 = the residual layers.
template <int PROD, int CTW, int NQ, int GK_TP, int FMT = 0, int EPI = 0>
__global__ __launch_bounds__(GK_WAVES * 64, GK_MINW(PROD, NQ)) void gemm_k1_kernel(const rsa_conv_params p) {
  constexpr int NHL = (PROD == 3) ? 2 : 1;
  constexpr int NPT = GK_TP / 16;
  // LDS image of one tile: [hi|lo][K-chunk q][plane in chunk 4][64 pixels] units; two buffers
  constexpr int BUF_UNITS = NHL * NQ * 4 * GK_TP;
  __shared__ uint4 s_x[2][BUF_UNITS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15;
  const int lg = lane >> 4;

  const int64_t HW = (int64_t)p.H * p.W;
  const int tiles_img = (int)((HW + GK_TP - 1) / GK_TP);
  const int num_tiles = tiles_img * p.batch;
  const int nq = (p.cin_planes + 3) >> 2;  // <= NQ (checked on the host)
  const int ct_total = (p.cout + 15) >> 4;
  const int npass = (ct_total + GK_WAVES * CTW - 1) / (GK_WAVES * CTW);
  if ((int)blockIdx.x >= num_tiles) return;

  // planes past cin_planes inside the last chunk are never written by the DMA: zero them once in both buffers
  // (their weights are zero, but 0 * stale-NaN would poison the accumulators)
  for (int u = tid; u < 2 * BUF_UNITS; u += GK_WAVES * 64) (&s_x[0][0])[u] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w_packed, 0, (uint32_t)((int64_t)nq * ct_total * NHL * 64 * 16), 0x00020000);
  const uint32_t wstep = (uint32_t)ct_total * NHL * 64 * 16;  // bytes per K-chunk (k1: one tap per chunk)

  // cooperative LDS-DMA of one tile: instruction j fills units [64j, 64j+64) of the image = row(s) (hl, plane) of GK_TP
  // pixels; instructions are dealt round-robin to the 8 waves; rows of planes past cin_planes keep their zeros
  constexpr int ROWS = NHL * NQ * 4;
  constexpr int NINST = ROWS * GK_TP / 64;
  auto issue_tile = [&](int tile, int buf) {
    const int n = tile / tiles_img;
    const int64_t pix0 = (int64_t)(tile - n * tiles_img) * GK_TP;
    const uint4* bh = (const uint4*)p.in_hi + (int64_t)n * p.in_batch_stride + pix0;
    const uint4* bl = (PROD == 3) ? (const uint4*)p.in_lo + (int64_t)n * p.in_batch_stride + pix0 : nullptr;
    for (int j = wave; j < NINST; j += GK_WAVES) {
      const int u = j * 64 + lane;
      const int row = u / GK_TP, px = u - row * GK_TP;
      const int hl = row / (NQ * 4);
      const int pl = row - hl * (NQ * 4);
      // wave-uniform skip of instructions that only cover missing planes (GK_TP divides 64: an instruction spans 64/GK_TP rows)
      const int row_first = (j * 64) / GK_TP;
      if (row_first - (row_first / (NQ * 4)) * (NQ * 4) >= p.cin_planes) continue;
      const bool ok = pl < p.cin_planes && pix0 + px < HW;
      const uint4* src = ok ? ((PROD == 3 && hl) ? bl : bh) + (int64_t)pl * p.in_plane_stride + px : (const uint4*)&g_zero_unit_gk[0];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&s_x[buf][j * 64], 16, 0, 0);
    }
  };

  const int p4 = (p.cout + 3) >> 2;
  const int cout8 = (p.cout + 7) & ~7;

  for (int pass = 0; pass < npass; ++pass) {
    // ---- this wave's weights for the whole reduction ----
    bf16x8 wr[CTW][NQ][NHL];
    int ctg[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      ctg[c] = (pass * GK_WAVES + wave) * CTW + c;
      const uint32_t base = ctg[c] < ct_total ? (uint32_t)((ctg[c] * NHL * 64 + lane) * 16) : 0xFFFFFFFFu;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int hl = 0; hl < NHL; ++hl) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, (q < nq) ? base : 0xFFFFFFFFu, (uint32_t)q * wstep + (uint32_t)hl * 1024u, 0);
          wr[c][q][hl] = __builtin_bit_cast(bf16x8, v);
        }
    }

    // bias / PReLU slopes of this wave's cout tiles: loaded once per pass, not per tile -- a load inside the tile epilogue would have
    // to wait (vmcnt retires in order) for the tile DMA issued just before it
    f32x4 biasv[CTW], slopev[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int c0 = ctg[c] * 16 + lg * 4;
      const bool live = ctg[c] < ct_total && c0 < cout8;
      biasv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      slopev[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr && live) biasv[c] = ((const f32x4*)p.bias)[c0 >> 2];
      if (p.act == RSA_ACT_PRELU && live) slopev[c] = ((const f32x4*)p.act_vec)[c0 >> 2];
    }
    // Epilogues without loads (no residual, no gate): the DMA of tile t+2 is issued right after the barrier that frees tile t's
    // buffer, BEFORE tile t's epilogue, so it flies under the epilogue and the next tile's MFMAs (two tiles in flight).  Epilogues
    // that load residuals keep the DMA behind them: their loads would otherwise queue behind it.
    // With at most 2 accumulator fragments per wave (CTW * NPT <= 2) the residual fragments are cheap to hold: they are loaded right
    // after the barrier, AHEAD of the DMA, and every layer takes the early path.
    constexpr bool RES_PREFETCH = CTW * NPT <= 2;
    const bool early = RES_PREFETCH || (p.res1 == nullptr && p.res2 == nullptr && p.act != RSA_ACT_SPAB_GATE);
    int tile = blockIdx.x;
    int buf = 0;
    issue_tile(tile, buf);
    if (early && tile + (int)gridDim.x < num_tiles) issue_tile(tile + (int)gridDim.x, 1);
    __syncthreads();  // vmcnt(0) for the DMA + everybody's rows landed
    for (; tile < num_tiles; tile += (int)gridDim.x, buf ^= 1) {
      const int ntile = tile + (int)gridDim.x;
      if (!(RSA_GK_ABL & 1) && !early && ntile < num_tiles) issue_tile(ntile, buf ^ 1);

      f32x4 acc[NPT][CTW];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[pt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const uint4* sx = &s_x[buf][0];
      const int bunit = lg * GK_TP + li;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q < nq) {
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
            const bf16x8 bh = *(const bf16x8*)&sx[(q * 4) * GK_TP + bunit + pt * 16];
            if (PROD == 3) {
              const bf16x8 bl = *(const bf16x8*)&sx[((NQ + q) * 4) * GK_TP + bunit + pt * 16];
#pragma unroll
              for (int c = 0; c < CTW; ++c) acc[pt][c] = mfma16<FMT>(wr[c][q][1], bh, acc[pt][c]);
#pragma unroll
              for (int c = 0; c < CTW; ++c) acc[pt][c] = mfma16<FMT>(wr[c][q][0], bl, acc[pt][c]);
            }
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
              if (RSA_GK_ABL & 4) {
                acc[pt][c][0] += (float)bh[0] * (float)wr[c][q][0][0];  // keeps the operands live without the matrix pipe
              } else {
                acc[pt][c] = mfma16<FMT>(wr[c][q][0], bh, acc[pt][c]);
              }
            }
          }
        }
      }

      __syncthreads();  // next tile's rows have landed (vmcnt(0) also retires the PREVIOUS tile's stores, long done); all waves are done with this buffer
      f32x4 pre1[RES_PREFETCH ? CTW * NPT : 1], pre2[RES_PREFETCH ? CTW * NPT : 1];
      if (RES_PREFETCH && EPI == 0) {
        const int n = tile / tiles_img;
        const int64_t pix0 = (int64_t)(tile - n * tiles_img) * GK_TP;
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
            const int c0 = ctg[c] * 16 + lg * 4;
            const int64_t pix = pix0 + pt * 16 + li;
            const bool ok = ctg[c] < ct_total && c0 < (p4 << 2) && pix < HW;
            const int64_t idx = ((int64_t)n * p4 + (c0 >> 2)) * HW + pix;
            pre1[c * NPT + pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pre2[c * NPT + pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.res1 != nullptr && ok) pre1[c * NPT + pt] = ((const f32x4*)p.res1)[idx];
            if (p.res2 != nullptr && ok) pre2[c * NPT + pt] = ((const f32x4*)p.res2)[idx];
          }
      }
      if (!(RSA_GK_ABL & 1) && early && ntile + (int)gridDim.x < num_tiles) issue_tile(ntile + (int)gridDim.x, buf);
      // ---- epilogue of this tile AFTER the barrier, so that its stores are in flight under the next tile's DMA and MFMAs instead
      //      of being drained by this barrier's vmcnt(0): lane owns channels c0..c0+3 of pixel pix0 + 16*pt + li ----
      if constexpr (EPI == 3) {
        // the residual layers of the transformer bodies (proj, fc2, DRCT's adjust layers): no activation or LeakyReLU, an optional f32
        // residual map (* alpha), outputs as an f32 map and / or fp16 hi planes; the descriptor's tests are wave-uniform and made once per tile
        const int n = tile / tiles_img;
        const int64_t pix0 = (int64_t)(tile - n * tiles_img) * GK_TP;
        const bool tail_tile = pix0 + GK_TP > HW;
        const bool has_res = p.res1 != nullptr, has_f32 = p.out_f32 != nullptr, has_pl = p.out_hi != nullptr;
        const float slope = p.act == RSA_ACT_NONE ? 1.f : p.act_param;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cbase = ctg[c] * 16;  // uniform
          if (ctg[c] >= ct_total || cbase >= cout8) continue;
          const int c0 = cbase + lg * 4;
          const f32x4 bias = biasv[c];
          const bool partial = cbase + 16 > p.cout;
          const bool grp_ok = c0 < (p4 << 2);  // this lane's channel group exists in the f32 maps
          const int64_t frow = ((int64_t)n * p4 + (c0 >> 2)) * HW + pix0;
          char* ob = (char*)p.out_hi + ((int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0) * 16;
          const uint32_t lane_off = ((uint32_t)(lg >> 1) * (uint32_t)p.out_plane_stride + (uint32_t)li) * 16u;
          f32x4 rr[NPT];
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
            rr[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_res && grp_ok && (!tail_tile || pix0 + pt * 16 + li < HW)) rr[pt] = ((const f32x4*)p.res1)[frow + pt * 16 + li];
          }
#pragma unroll
          for (int pp = 0; pp < NPT / 2; ++pp) {
            uint32_t h[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int pt = pp * 2 + e;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[pt][c][r] + bias[r];
              if (slope != 1.f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], v[r] * slope);
              }
              if (has_res) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + rr[pt][r];
              }
              if (partial) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (c0 + r >= p.cout) v[r] = 0.f;
              }
              if (has_f32 && grp_ok && (!tail_tile || pix0 + pt * 16 + li < HW) && !(RSA_GK_ABL & 2))
                ((f32x4*)p.out_f32)[frow + pt * 16 + li] = (f32x4){v[0], v[1], v[2], v[3]};
              uint32_t lo_unused;
              split2<RSA_PF_F16>(v[0], v[1], h[e][0], lo_unused);
              split2<RSA_PF_F16>(v[2], v[3], h[e][1], lo_unused);
            }
            if (has_pl) {  // uniform
              typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
              const u32x2 h0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
              const u32x2 h1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
              const int pt = pp * 2 + (lg & 1);
              if (c0 < cout8 && (!tail_tile || pix0 + pt * 16 + li < HW) && !(RSA_GK_ABL & 2))
                *(uint4*)(ob + lane_off + (uint32_t)(pt * 16) * 16u) = make_uint4(h0.x, h1.x, h0.y, h1.y);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (EPI != 0) {
        const int n = tile / tiles_img;
        const int64_t pix0 = (int64_t)(tile - n * tiles_img) * GK_TP;
        const bool tail_tile = pix0 + GK_TP > HW;  // uniform: only the last tile of an image tests its pixels
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cbase = ctg[c] * 16;  // uniform
          if (ctg[c] >= ct_total || cbase >= cout8) continue;
          const int c0 = cbase + lg * 4;
          const f32x4 bias = biasv[c];
          const bool partial = cbase + 16 > p.cout;  // uniform: the last cout tile zeroes the channels beyond Cout
          char* ob = (char*)p.out_hi + ((int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (cbase >> 3)) * p.out_plane_stride + pix0) * 16;
          const uint32_t lane_off = ((uint32_t)(lg >> 1) * (uint32_t)p.out_plane_stride + (uint32_t)li) * 16u;
#pragma unroll
          for (int pp = 0; pp < NPT / 2; ++pp) {
            uint32_t h[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = acc[pp * 2 + e][c][r] + bias[r];
                if (EPI == 2) v[r] = act_apply<AC_GELU>(v[r], RSA_ACT_GELU, 0.f);
              }
              if (partial) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (c0 + r >= p.cout) v[r] = 0.f;
              }
              uint32_t lo_unused;
              split2<RSA_PF_F16>(v[0], v[1], h[e][0], lo_unused);
              split2<RSA_PF_F16>(v[2], v[3], h[e][1], lo_unused);
            }
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const u32x2 h0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
            const u32x2 h1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
            const int pt = pp * 2 + (lg & 1);  // the pixel tile whose whole unit this lane stores
            if (c0 < cout8 && (!tail_tile || pix0 + pt * 16 + li < HW) && (!(RSA_GK_ABL & 2) || h0.x == 0x12345678u))
              *(uint4*)(ob + lane_off + (uint32_t)(pt * 16) * 16u) = make_uint4(h0.x, h1.x, h0.y, h1.y);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        const int n = tile / tiles_img;
        const int64_t pix0 = (int64_t)(tile - n * tiles_img) * GK_TP;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int cbase = ctg[c] * 16;
          if (ctg[c] >= ct_total || cbase >= cout8) continue;
          const int c0 = cbase + lg * 4;
          if (c0 >= cout8) continue;
          const f32x4 bias = biasv[c];
          const f32x4 slope = slopev[c];
          const bool has_f32grp = c0 < (p4 << 2);
          const int64_t f32row = ((int64_t)n * p4 + (c0 >> 2)) * HW;
          const int64_t outrow = (int64_t)n * p.out_batch_stride + (int64_t)(p.out_plane_off + (c0 >> 3)) * p.out_plane_stride;
          static_assert(NPT % 2 == 0, "pixel tiles are stored in pairs");
#pragma unroll
          for (int pp = 0; pp < NPT / 2; ++pp) {
            // the two pixel tiles (2pp, 2pp+1) of this cout tile: after the per-value math the lanes that hold the two halves of a
            // 16-byte plane unit (lg, lg^1 = lanes l, l^16) swap one half each (v_permlane16_swap_b32), so that the even lane stores the
            // whole unit of pixel tile 2pp and the odd lane the whole unit of 2pp+1: 16-byte stores instead of two 8-byte halves
            // (the 8-byte form was store-issue bound on the write-heavy layers: qkv writes 3 GB per 1 M tokens)
            float v[2][4];
            int64_t pixe[2];
            bool live[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int pt = pp * 2 + e;
              const int64_t pix = pix0 + pt * 16 + li;
              pixe[e] = pix;
              live[e] = pix < HW;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[e][r] = acc[pt][c][r] + bias[r];
              if (p.act == RSA_ACT_SPAB_GATE) {
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (RES_PREFETCH) rr = pre1[c * NPT + pt];
                else if (has_f32grp && live[e]) rr = ((const f32x4*)p.res1)[f32row + pix];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[e][r] = (v[e][r] + rr[r]) * (1.f / (1.f + expf(-v[e][r])) - 0.5f);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[e][r] = (p.act == RSA_ACT_GELU) ? act_apply<AC_GELU>(v[e][r], p.act, 0.f) : act_apply<AC_LINEAR>(v[e][r], p.act, p.act == RSA_ACT_PRELU ? slope[r] : p.act_param);
                if (p.res1 != nullptr && has_f32grp) {
                  f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                  if (RES_PREFETCH) rr = pre1[c * NPT + pt];
                  else if (live[e]) rr = ((const f32x4*)p.res1)[f32row + pix];
#pragma unroll
                  for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.alpha + rr[r];
                }
              }
              if (p.res2 != nullptr && has_f32grp) {
                f32x4 rr = {0.f, 0.f, 0.f, 0.f};
                if (RES_PREFETCH) rr = pre2[c * NPT + pt];
                else if (live[e]) rr = ((const f32x4*)p.res2)[f32row + pix];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[e][r] = v[e][r] * p.beta + rr[r];
              }
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (c0 + r >= p.cout) v[e][r] = 0.f;
              if (p.out_f32 != nullptr && has_f32grp && live[e]) ((f32x4*)p.out_f32)[f32row + pix] = (f32x4){v[e][0], v[e][1], v[e][2], v[e][3]};
            }
            if (p.out_hi != nullptr) {  // wave-uniform: every lane takes part in the exchange
              uint32_t h[2][2], l[2][2];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                split2_rt(p.out_fmt == RSA_PF_F16, v[e][0], v[e][1], h[e][0], l[e][0]);
                split2_rt(p.out_fmt == RSA_PF_F16, v[e][2], v[e][3], h[e][1], l[e][1]);
              }
              typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
              const u32x2 h0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
              const u32x2 h1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
              const int odd = lg & 1;
              const int64_t pixs = odd ? pixe[1] : pixe[0];  // the pixel whose whole unit this lane stores (same column li in both tiles)
              const int64_t off = (outrow + pixs) * 16;
              if (pixs < HW && (!(RSA_GK_ABL & 2) || h0.x == 0x12345678u)) {
                *(uint4*)((char*)p.out_hi + off) = make_uint4(h0.x, h1.x, h0.y, h1.y);
              }
              if (p.out_lo != nullptr) {
                const u32x2 l0 = __builtin_amdgcn_permlane16_swap(l[0][0], l[1][0], false, false);
                const u32x2 l1 = __builtin_amdgcn_permlane16_swap(l[0][1], l[1][1], false, false);
                if (pixs < HW) *(uint4*)((char*)p.out_lo + off) = make_uint4(l0.x, l1.x, l0.y, l1.y);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
}

template <int PROD, int CTW, int NQ, int GK_TP, int FMT = 0, int EPI = 0>
static int launch_gemm(const rsa_conv_params& p, hipStream_t stream) {
  const int64_t HW = (int64_t)p.H * p.W;
  const int64_t num_tiles = ((HW + GK_TP - 1) / GK_TP) * p.batch;
  if (num_tiles > 0x7fffffff) return RSA_E_UNSUPPORTED;
  static std::atomic<int> resident_cache{0};  // concurrent first calls compute the same value: idempotent, never torn
  int resident = resident_cache.load(std::memory_order_relaxed);
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_k1_kernel<PROD, CTW, NQ, GK_TP, FMT, EPI>, GK_WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return (int)hipGetLastError();
    resident = per_cu * prop.multiProcessorCount;
    resident_cache.store(resident, std::memory_order_relaxed);
  }
  int gx = resident;
  if (gx > num_tiles) gx = (int)num_tiles;
  hipLaunchKernelGGL((gemm_k1_kernel<PROD, CTW, NQ, GK_TP, FMT, EPI>), dim3((unsigned)gx), dim3(GK_WAVES * 64), 0, stream, p);
  return (int)hipGetLastError();
}

static bool gemm_k1_direct_enabled() {  // RSA_GK_DIRECT=0 in the environment: the generic epilogue everywhere (A/B runs)
  static const bool on = [] {
    const char* e = getenv("RSA_GK_DIRECT");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}

// Returns -100 when the layer is not a fit for this schedule (caller falls back to the halo-tile kernels).
int gemm_k1_launch(const rsa_conv_params& p, hipStream_t stream) {
  if (p.ksize != 1 || p.out_nchw != nullptr || p.upsample2x) return -100;
  if (p.in_fmt == RSA_PF_F16 && p.products != 1) return -100;  // fp16 planes: the one-product form only (three fp16 products take conv_kernel<1, ...>)
  if (p.res1_hi != nullptr || p.res2_hi != nullptr) return -100;  // plane residuals: only the halo-tile kernels' epilogue reads them
  if (p.act == RSA_ACT_MISH || p.act == RSA_ACT_SILU) return -100;  // only the linear class, GELU and the SPAB gate are compiled into this schedule
  if (p.cout < 96) return -100;  // too few cout tiles to occupy 8 waves
  const int nq = (p.cin_planes + 3) / 4;
  if (p.products == 3) {
    if (nq <= 8) return launch_gemm<3, 2, 8, 64>(p, stream);
    if (nq <= 16) return launch_gemm<3, 1, 16, 32>(p, stream);
  } else {
    // one product: a K of at most 256 channels takes the 8-chunk image (2 x 32 KB of LDS: two workgroups per CU, which the register budget of
    // the kernel already allows; the 16-chunk image fills the LDS with one)
    // K <= 192 on fp16 planes (the 180-channel bodies of DAT / HAT / DRCT): six chunks of weights are 48 registers instead of 64, which brings
    // the kernel under 128 VGPRs without spills = TWO workgroups per CU (round 4)
    if (p.in_fmt == RSA_PF_F16 && nq <= 16) {
      // the direct epilogue: fp16 hi planes out and nothing else (qkv: no activation -> EPI 1; fc1: GELU -> EPI 2)
      const bool planes_only = gemm_k1_direct_enabled() && p.out_hi != nullptr && p.out_lo == nullptr && p.out_f32 == nullptr && p.out_fmt == RSA_PF_F16 &&
                               p.res1 == nullptr && p.res2 == nullptr && p.out_plane_stride < (1ll << 27);
      // EPI 3: no activation / LeakyReLU, an optional f32 residual, f32 map and / or fp16 hi planes out (proj, fc2, DRCT's adjust layers)
      const bool resid = gemm_k1_direct_enabled() && p.res2 == nullptr && p.out_lo == nullptr && (p.out_hi == nullptr || p.out_fmt == RSA_PF_F16) &&
                         (p.out_hi != nullptr || p.out_f32 != nullptr) && p.out_plane_stride < (1ll << 27) &&
                         (p.act == RSA_ACT_NONE || (p.act == RSA_ACT_LRELU && p.act_param >= 0.f && p.act_param <= 1.f));
      const int epi = planes_only && p.act == RSA_ACT_NONE ? 1 : (planes_only && p.act == RSA_ACT_GELU ? 2 : (resid ? 3 : 0));
      if (nq <= 6) return epi == 1 ? launch_gemm<1, 2, 6, 64, RSA_PF_F16, 1>(p, stream) : epi == 2 ? launch_gemm<1, 2, 6, 64, RSA_PF_F16, 2>(p, stream) : epi == 3 ? launch_gemm<1, 2, 6, 64, RSA_PF_F16, 3>(p, stream) : launch_gemm<1, 2, 6, 64, RSA_PF_F16>(p, stream);
      if (nq <= 8) return epi == 1 ? launch_gemm<1, 2, 8, 64, RSA_PF_F16, 1>(p, stream) : epi == 2 ? launch_gemm<1, 2, 8, 64, RSA_PF_F16, 2>(p, stream) : epi == 3 ? launch_gemm<1, 2, 8, 64, RSA_PF_F16, 3>(p, stream) : launch_gemm<1, 2, 8, 64, RSA_PF_F16>(p, stream);
      return epi == 1 ? launch_gemm<1, 2, 16, 64, RSA_PF_F16, 1>(p, stream) : epi == 2 ? launch_gemm<1, 2, 16, 64, RSA_PF_F16, 2>(p, stream) : epi == 3 ? launch_gemm<1, 2, 16, 64, RSA_PF_F16, 3>(p, stream) : launch_gemm<1, 2, 16, 64, RSA_PF_F16>(p, stream);
    }
    if (nq <= 8) return launch_gemm<1, 2, 8, 64>(p, stream);
    if (nq <= 16) return launch_gemm<1, 2, 16, 64>(p, stream);
  }
  return -100;
}

}  // namespace rsa
