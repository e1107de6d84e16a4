// conv_inst_ring1h.hip — conv_ring<SHAPE = 1> on fp16 planes: one product (conv5 of a residual dense block under the 'auto' precision
// policy) and three products (the trunk convolution, whose input is the fp16 residual stream).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring1_f16(const rsa_conv_params& p, hipStream_t stream) {
  if (p.products == 1 && conv_ring_xres_enabled() && conv_ring_xres_eligible(p)) {
    if (p.lo8_flags == 0) return launch_ring<1, 0, 0, 0, RSA_PF_F16, 1, 1>(p, stream);
    return (p.lo8_flags & RSA_LO8_OUT) ? launch_ring<1, 0, 0, 0, RSA_PF_F16, 1, 4>(p, stream) : launch_ring<1, 0, 0, 0, RSA_PF_F16, 1, 5>(p, stream);
  }
  return p.products == 1 ? launch_ring<1, 0, 0, 0, RSA_PF_F16, 1>(p, stream) : launch_ring<1, 0, 0, 0, RSA_PF_F16, 3>(p, stream);
}
unsigned int conv_ring1h_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring1h_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
