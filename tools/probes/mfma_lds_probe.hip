// mfma_lds_probe.hip — the conv kernels' inner step in isolation: 6 MFMAs (2 cout tiles x 3 products) per pair of ds_read_b128
// (hi and lo fragment of one pixel tile), fragments prefetched DEPTH steps ahead, weights constant in registers, LDS filled once
// with pseudo-random bf16.  Compares against the same MFMA stream without the LDS reads (READS = 0).  WEIGHTS = 1 adds the kernels'
// weight stream: 4 x 16 bytes per lane per tap from an L2-resident blob, requested one tap ahead (current/next register pair).
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int READS, int DEPTH, int WEIGHTS>
__global__ void probe(float* out, int iters, const uint4* wblob) {
  __shared__ uint4 lds[2 * 2496];  // one halo-tile buffer, hi | lo, as in conv_kernel
  unsigned h = threadIdx.x * 2654435761u + 12345u;
  for (int i = threadIdx.x; i < 2 * 2496; i += blockDim.x) {
    uint4 v;
    unsigned* w = (unsigned*)&v;
    for (int k = 0; k < 4; ++k) {
      h = h * 1664525u + 1013904223u;
      w[k] = (h & 0x007f007fu) | 0x3f803f80u | (h & 0x80008000u);  // two bf16 in +-[1, 2)
    }
    lds[i] = v;
  }
  __syncthreads();
  bf16x8 wh[2], wl[2];
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      wh[c][i] = (__bf16)((float)(int)(h >> 8 & 0xffff) / 65536.f - 0.5f);
      wl[c][i] = (__bf16)((float)(int)(h >> 12 & 0xffff) / 65536.f * 0.004f);
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = (lane >> 4) * 624 + (wave & 3) * 4 * 34 + (lane & 15);
  f32x4 acc[8][2];
  for (int i = 0; i < 8; ++i)
    for (int c = 0; c < 2; ++c) acc[i][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NSTEP = 72;  // 9 taps x 8 pixel tiles
  const uint4* wp = wblob + (threadIdx.x & 63);
  uint4 wnx[4];
  if (WEIGHTS)
    for (int k = 0; k < 4; ++k) wnx[k] = wp[k * 64];
  for (int it = 0; it < iters; ++it) {
    bf16x8 rh[DEPTH + 1], rl[DEPTH + 1];
    auto unit = [&](int i) { const int t = i / 8, pt = i % 8; return base + ((pt >> 1) + t / 3) * 34 + (pt & 1) * 16 + t % 3; };
    if (READS) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        rh[i] = *(const bf16x8*)&lds[unit(i)];
        rl[i] = *(const bf16x8*)&lds[2496 + unit(i)];
      }
    } else {
#pragma unroll
      for (int i = 0; i <= DEPTH; ++i) {
        rh[i] = *(const bf16x8*)&lds[unit(i)];
        rl[i] = *(const bf16x8*)&lds[2496 + unit(i)];
      }
    }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      if (READS && i + DEPTH < NSTEP) {
        rh[(i + DEPTH) % (DEPTH + 1)] = *(const bf16x8*)&lds[unit(i + DEPTH)];
        rl[(i + DEPTH) % (DEPTH + 1)] = *(const bf16x8*)&lds[2496 + unit(i + DEPTH)];
      }
      const int sp = i % 8;
      if (WEIGHTS && sp == 0) {
        wh[0] = __builtin_bit_cast(bf16x8, wnx[0]);
        wl[0] = __builtin_bit_cast(bf16x8, wnx[1]);
        wh[1] = __builtin_bit_cast(bf16x8, wnx[2]);
        wl[1] = __builtin_bit_cast(bf16x8, wnx[3]);
        const int s = ((it * 9 + i / 8 + 1) % 45) * 256;  // 45 steps x 4 KB, as a 160-channel layer
#pragma unroll
        for (int k = 0; k < 4; ++k) wnx[k] = wp[s + k * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 wf = pr == 0 ? wl[c] : wh[c];
          const bf16x8 bf = pr == 1 ? rl[i % (DEPTH + 1)] : rh[i % (DEPTH + 1)];
          acc[sp][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, bf, acc[sp][c], 0, 0, 0);
        }
      if (READS && i + DEPTH < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((it & 63) == 63)  // keep the accumulators finite
      for (int i = 0; i < 8; ++i)
        for (int c = 0; c < 2; ++c) acc[i][c] *= 1e-3f;
  }
  f32x4 s = acc[0][0];
  for (int i = 0; i < 8; ++i)
    for (int c = 0; c < 2; ++c) s += acc[i][c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int READS, int DEPTH, int WEIGHTS>
static void run(int waves_per_simd, float* out, const uint4* wblob) {
  const int iters = 2000, threads = 256 * waves_per_simd, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<READS, DEPTH, WEIGHTS><<<blocks, threads>>>(out, 10, wblob);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<READS, DEPTH, WEIGHTS><<<blocks, threads>>>(out, iters, wblob);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * 16 * 32 * 432.0 * iters * (threads / 64) * blocks;
  printf("waves/SIMD %d, LDS reads %s, depth %d, weight stream %s: %8.3f ms  %7.1f TFLOP/s issued\n", waves_per_simd, READS ? "yes" : "no ", DEPTH, WEIGHTS ? "yes" : "no ", ms, flop / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  uint4* wblob;
  hipMalloc(&wblob, 46 * 256 * sizeof(uint4) + 4096);
  hipMemset(wblob, 0x3c, 46 * 256 * sizeof(uint4) + 4096);  // bf16 0x3c3c = 0.0115
  for (int w = 1; w <= 2; ++w) {
    run<0, 2, 0>(w, out, wblob);
    run<1, 2, 0>(w, out, wblob);
    run<1, 4, 0>(w, out, wblob);
    run<1, 2, 1>(w, out, wblob);
  }
  return 0;
}
