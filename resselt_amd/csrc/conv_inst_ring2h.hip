// conv_inst_ring2h.hip — conv_ring<SHAPE = 2> in one fp16 product (the growth convolutions of a residual dense block under the 'auto'
// precision policy: 276 of the 363 launches of a 1080p RRDBNet frame).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring2_f16(const rsa_conv_params& p, hipStream_t stream) {
  if (conv_ring_xres_enabled() && conv_ring_em1_eligible(p)) return launch_ring<2, 0, 0, 0, RSA_PF_F16, 1, 2>(p, stream);  // the growth convolutions
  return p.products == 1 ? launch_ring<2, 0, 0, 0, RSA_PF_F16, 1>(p, stream) : launch_ring<2, 0, 0, 0, RSA_PF_F16, 3>(p, stream);
}
unsigned int conv_ring2h_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring2h_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
