// trapsts_probe.hip — does gfx950 record a float OVERFLOW of v_cvt_pk_f16_f32 (and of other instructions) in the sticky exception bits of
// TRAPSTS without traps enabled?  (round 4: a free fp16-overflow detector for the epilogues that write fp16 planes)
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/trapsts_probe.hip -o tools/probes/trapsts_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned trapsts() {
  unsigned v;
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(v)::"memory");
  return v;
}
__device__ __forceinline__ void clear_excp() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory"); }

__global__ void probe(const float* in, unsigned* out, float* sink) {
  const int lane = threadIdx.x;
  unsigned r[8];
  r[0] = trapsts();
  clear_excp();
  r[1] = trapsts();
  // 1: an in-range conversion
  float a = in[0] + (float)lane;  // 100 + lane
  f16x2 h = {(_Float16)a, (_Float16)(a * 2.f)};
  asm volatile("" ::"v"(h));
  r[2] = trapsts();
  // 2: an out-of-range conversion in ONE lane
  float b = lane == 5 ? in[1] : 1.0f;  // 1e6
  f16x2 g = {(_Float16)b, (_Float16)1.0f};
  sink[lane] = (float)g[0];
  r[3] = trapsts();
  clear_excp();
  // 3: exp2 overflow
  float e = __builtin_amdgcn_exp2f(lane == 7 ? in[2] : 1.0f);  // 2^200
  sink[64 + lane] = e;
  r[4] = trapsts();
  clear_excp();
  // 4: MFMA whose f32 accumulator overflows
  f16x8 av, bv;
  for (int i = 0; i < 8; ++i) av[i] = (_Float16)in[3], bv[i] = (_Float16)in[3];  // 60000 * 60000 * 32 > f32 max? no: 1.15e11 -- fine in f32; use acc seed
  f32x4 c = {in[4], in[4], in[4], in[4]};  // 3e38
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
  sink[128 + lane] = c[0];
  r[5] = trapsts();
  clear_excp();
  // 5: f32 add overflow
  float s = in[4] + in[4];
  sink[192 + lane] = s;
  r[6] = trapsts();
  clear_excp();
  r[7] = trapsts();
  if (lane == 0)
    for (int i = 0; i < 8; ++i) out[i] = r[i];
}

int main() {
  float hin[5] = {100.f, 1e6f, 200.f, 60000.f, 3e38f};
  float *in, *sink;
  unsigned* out;
  hipMalloc(&in, sizeof(hin));
  hipMalloc(&sink, 256 * 4);
  hipMalloc(&out, 8 * 4);
  hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, out, sink);
  unsigned r[8];
  float hs[256];
  if (hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) return printf("launch failed\n"), 1;
  hipMemcpy(hs, sink, sizeof(hs), hipMemcpyDeviceToHost);
  const char* what[8] = {"at entry", "after clear", "in-range cvt_f16", "OVERFLOWING cvt_f16 (lane 5)", "exp2 overflow (lane 7)", "MFMA f32 accumulator overflow", "v_add_f32 overflow", "after clear"};
  for (int i = 0; i < 8; ++i) printf("TRAPSTS %-32s = 0x%08x  EXCP[8:0] = 0x%03x (overflow bit 3 = %u, inexact bit 5 = %u)\n", what[i], r[i], r[i] & 0x1ff, (r[i] >> 3) & 1, (r[i] >> 5) & 1);
  printf("cvt result lane 5 = %g, exp2 lane 7 = %g, mfma = %g, add = %g\n", hs[5], hs[64 + 7], hs[128], hs[192]);
  return 0;
}
