// conv_inst_ringpair.hip — conv_ring_pair (two growth convolutions of a residual dense block in one launch) and its launcher.
#include <stdlib.h>

#include <atomic>

#include "conv_ring_pair.h"

namespace rsa {

static std::atomic<int> g_pair_override{-1};  // rsa_debug_set_pair: -1 = follow the environment, 0 / 1 = forced (in-process A/B runs)
void conv_pair_override(int v) { g_pair_override.store(v < 0 ? -1 : (v ? 1 : 0)); }
bool conv_pair_enabled() {
  static const bool on = [] {
    const char* e = getenv("RSA_CONV_PAIR");
    return e == nullptr || e[0] != '0';
  }();
  const int o = g_pair_override.load(std::memory_order_relaxed);
  return (o < 0 ? on : o != 0) && conv_ring_enabled();
}

int conv_launch_pair(const rsa_conv_params& a, const rsa_conv_params& b, hipStream_t stream) {
  using G = PairGeo;
  PairParams p;
  p.batch = a.batch;
  p.H = a.H;
  p.W = a.W;
  p.nqa = a.cin_planes >> 2;
  p.in_hi = a.in_hi;
  p.in_plane_stride = a.in_plane_stride;
  p.in_batch_stride = a.in_batch_stride;
  p.wa = a.w_packed;
  p.wb = b.w_packed;
  p.bias_a = a.bias;
  p.bias_b = b.bias;
  p.slope_a = a.act == RSA_ACT_NONE ? 1.f : a.act_param;
  p.slope_b = b.act == RSA_ACT_NONE ? 1.f : b.act_param;
  p.outa_hi = a.out_hi;
  p.outa_unit0 = (int64_t)a.out_plane_off * a.out_plane_stride;
  p.outa_plane_stride = a.out_plane_stride;
  p.outa_batch_stride = a.out_batch_stride;
  p.outb_hi = b.out_hi;
  p.outb_unit0 = (int64_t)b.out_plane_off * b.out_plane_stride;
  p.outb_plane_stride = b.out_plane_stride;
  p.outb_batch_stride = b.out_batch_stride;
  p.tile_order = a.tile_order;
  const int tiles_x = (p.W + G::TWO - 1) / G::TWO;
  const int tiles_y = (p.H + G::TH - 1) / G::TH;
  const int64_t num_tiles = (int64_t)tiles_x * tiles_y * p.batch;
  if (num_tiles > 0x3fffffff) return RSA_E_UNSUPPORTED;
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
  }();
  int gx = cus;  // one persistent workgroup per CU (the ring and the x_A image take the whole LDS)
  if (gx > num_tiles) gx = (int)num_tiles;
  hipLaunchKernelGGL(conv_ring_pair, dim3((unsigned)gx, 1, 1), dim3(9 * 64), 0, stream, p, ring_aux());
  return (int)hipGetLastError();
}

unsigned int conv_ring_pair_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_PAIR_STAMPS
}  // namespace rsa
extern "C" int rsa_debug_pair_stamps(unsigned long long* out, int n) {  // diagnostic build: copies the stamp table (synchronises) and clears it
  static unsigned long long zero[256 * 9 * 8];
  if (n > 256 * 9 * 8) n = 256 * 9 * 8;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rsa::g_pair_stamps), sizeof(unsigned long long) * n) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(rsa::g_pair_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
namespace rsa {
#endif
#ifdef RSA_RING_DEBUG
int conv_ringpair_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif

}  // namespace rsa
