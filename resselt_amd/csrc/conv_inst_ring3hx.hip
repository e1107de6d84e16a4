// conv_inst_ring3hx.hip — conv_ring<SHAPE = 3> in three fp16 products (the full-amplitude layers of the SPAN family under the 'auto' policy:
// the pixel-shuffle head).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring3_f16x3(const rsa_conv_params& p, hipStream_t stream) {
  if (p.out_nchw != nullptr) return (p.cin_planes & 3) ? launch_ring<3, 0, 1, 1, RSA_PF_F16, 3>(p, stream) : launch_ring<3, 0, 1, 0, RSA_PF_F16, 3>(p, stream);
  return (p.cin_planes & 3) ? launch_ring<3, 0, 0, 1, RSA_PF_F16, 3>(p, stream) : launch_ring<3, 0, 0, 0, RSA_PF_F16, 3>(p, stream);
}
unsigned int conv_ring3hx_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring3hx_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
