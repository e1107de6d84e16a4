// mfma_issue_probe.hip — how much of the matrix pipe does ONE wave per SIMD reach, against two and four?
// Bare v_mfma_f32_16x16x32_bf16 stream, 16 independent accumulator tiles per wave, operands held in registers (no memory traffic).
// RANDOM = 0: one constant operand pair (no data toggling: the clock stays up); RANDOM = 1: eight pseudo-random operand pairs in
// rotation (values in [-2, 2) with random mantissas), the data-dependent power case a real layer is in.
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_issue_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int DIST, int RANDOM>  // DIST = number of accumulators cycled through (dependent MFMAs are DIST apart)
__global__ void probe(float* out, int iters) {
  bf16x8 a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u;
      const float va = RANDOM ? (float)(int)(h >> 8 & 0xffff) / 16384.f - 2.f : (float)(threadIdx.x % 7 + i);
      h = h * 1664525u + 1013904223u;
      const float vb = RANDOM ? (float)(int)(h >> 8 & 0xffff) / 16384.f - 2.f : (float)(threadIdx.x % 5 - i);
      a[k][i] = (__bf16)va;
      b[k][i] = (__bf16)vb;
    }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i % DIST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM ? (i + r) % 8 : 0], b[RANDOM ? (i * 3 + r) % 8 : 0], acc[i % DIST], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int DIST, int RANDOM>
static void run(int waves_per_simd, float* out) {
  const int iters = 20000, threads = 256 * waves_per_simd, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<DIST, RANDOM><<<blocks, threads>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<DIST, RANDOM><<<blocks, threads>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * 16 * 32 * 64.0 * iters * (threads / 64) * blocks;
  printf("%s operands, waves/SIMD %d, dependent MFMAs %2d apart: %7.3f ms  %7.1f TFLOP/s\n", RANDOM ? "random  " : "constant", waves_per_simd, DIST, ms, flop / ms / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  for (int w = 1; w <= 4; w *= 2) {
    run<2, 0>(w, out);
    run<16, 0>(w, out);
    run<2, 1>(w, out);
    run<16, 1>(w, out);
  }
  return 0;
}
