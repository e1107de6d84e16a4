// common.h — shared host-side helpers of the C-ABI library (error string, dispatch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "resselt_amd.h"

namespace rsa {

// Records a message for rsa_last_error_string() (thread-local) and returns `code`.
int set_error(int code, const char* msg);

// conv_mfma.hip
int conv_launch(const rsa_conv_params& p, hipStream_t stream);
int conv_nct(int cout);
const char* conv_kernel_name(const rsa_conv_params& p);

// Packed-weight layouts (rsa_pack_weights): 0 = tap-major chunks (conv_kernel.h, gemm_k1.hip), 1 = tap-pair order (conv_ring.h)
enum { RSA_WL_TAPS = 0, RSA_WL_PAIRS = 1, RSA_WL_HALFPAIRS = 2 };  // 2: the ring schedule's half mode (odd number of half chunks)

// A descriptor takes the ring schedule (conv_ring.h) iff: 3x3, three products, split-plane / f32-map outputs (final NCHW stores: the
// three-tile shape only) and either whole
// 32-channel input chunks with two, three or four cout tiles (three: no fused upsampling), or an odd number of 16-channel half chunks
// with three cout tiles (the 48 -> 48 layers of the SPAN family: half mode).  RSA_CONV_RING=0 in the environment switches the schedule off (A/B runs).
bool conv_ring_enabled();
void conv_ring_override(int v);
inline bool conv_ring_eligible(const rsa_conv_params& p) {
  const int ct = (p.cout + 15) / 16;
  if (p.ksize != 3 || p.products != 3 || p.cin_planes < 2 || (p.cin_planes & 1)) return false;
  if (p.out_nchw != nullptr) return ct == 3 && !p.upsample2x;  // final stores: the three-tile shape only (the pixel-shuffle heads of SPAN / Compact)
  if ((p.cin_planes & 3) == 0) return ct == 2 || ct == 4 || (ct == 3 && !p.upsample2x);
  return ct == 3 && !p.upsample2x;
}
inline int conv_ring_layout(const rsa_conv_params& p) { return (p.cin_planes & 3) == 0 ? RSA_WL_PAIRS : RSA_WL_HALFPAIRS; }
inline int conv_weight_layout(const rsa_conv_params& p) { return (conv_ring_enabled() && conv_ring_eligible(p)) ? conv_ring_layout(p) : RSA_WL_TAPS; }

// conv_inst_ring*.hip
int conv_launch_ring(const rsa_conv_params& p, hipStream_t stream);
unsigned int conv_ring_aborts();

}  // namespace rsa
