// Probe (not part of the product): why does a per-pixel reduction over 23 fp16 planes (96 MB at 512x512) take 74 us?
// Variants of the access pattern, timed with HIP events.   hipcc --offload-arch=gfx950 -O3 plane_stats_probe.hip -o plane_stats_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V0: thread = pixel, planes in chunks of U (the product's form)
template <int U>
__global__ __launch_bounds__(256) void v0(const h8* in, int64_t ps, int64_t HW, int planes, float* out) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  float sum = 0.f;
  for (int p0 = 0; p0 < planes; p0 += U) {
    h8 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = in[(int64_t)min(p0 + u, planes - 1) * ps + pix];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p0 + u < planes)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)r[u][j];
  }
  out[pix] = sum;
}
// V1: workgroup = 64 pixels, wave w takes planes w, w+4, ... (4x the waves), LDS reduction
__global__ __launch_bounds__(256) void v1(const h8* in, int64_t ps, int64_t HW, int planes, float* out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t pix = (int64_t)blockIdx.x * 64 + lane;
  float sum = 0.f;
  if (pix < HW) {
    h8 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = in[(int64_t)min(wave + 4 * k, planes - 1) * ps + pix];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (wave + 4 * k < planes)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += (float)r[k][j];
  }
  red[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && pix < HW) out[pix] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
// V2: plain streaming read of the same bytes (no per-pixel structure): grid-stride, 16 B per lane
__global__ __launch_bounds__(256) void v2(const h8* in, int64_t total, float* out) {
  float sum = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const h8 r = in[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += (float)r[j];
  }
  if (sum == 12345.f) out[0] = sum;
}
int main() {
  const int H = 512, W = 512, planes = 23;
  const int64_t HW = (int64_t)H * W, ps = HW;
  h8* in; float* out;
  CK(hipMalloc(&in, planes * ps * 16)); CK(hipMalloc(&out, HW * 4));
  CK(hipMemset(in, 0, planes * ps * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.1f us  %.2f TB/s\n", name, ms / 20 * 1e3, planes * ps * 16 / (ms / 20 * 1e-3) / 1e12);
  };
  timeit("v0 thread=pixel, U=1", [&] { hipLaunchKernelGGL(v0<1>, dim3((HW + 255) / 256), dim3(256), 0, 0, in, ps, HW, planes, out); });
  timeit("v0 thread=pixel, U=4", [&] { hipLaunchKernelGGL(v0<4>, dim3((HW + 255) / 256), dim3(256), 0, 0, in, ps, HW, planes, out); });
  timeit("v0 thread=pixel, U=8", [&] { hipLaunchKernelGGL(v0<8>, dim3((HW + 255) / 256), dim3(256), 0, 0, in, ps, HW, planes, out); });
  timeit("v0 thread=pixel, U=23", [&] { hipLaunchKernelGGL(v0<23>, dim3((HW + 255) / 256), dim3(256), 0, 0, in, ps, HW, planes, out); });
  timeit("v1 64 px per WG, planes over 4 waves", [&] { hipLaunchKernelGGL(v1, dim3((HW + 63) / 64), dim3(256), 0, 0, in, ps, HW, planes, out); });
  timeit("v2 linear stream, grid 2048", [&] { hipLaunchKernelGGL(v2, dim3(2048), dim3(256), 0, 0, in, planes * ps, out); });
  timeit("v2 linear stream, grid 8192", [&] { hipLaunchKernelGGL(v2, dim3(8192), dim3(256), 0, 0, in, planes * ps, out); });
  CK(hipDeviceSynchronize());
  return 0;
}
