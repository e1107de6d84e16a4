// conv_inst_ring3h.hip — conv_ring<SHAPE = 3> in one fp16 product (the 48-channel layers of the SPAN family).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring3_f16x3(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring3hx.hip
int conv_launch_ring3_span(const rsa_conv_params& p, hipStream_t stream);   // conv_inst_ring3hs.hip
int conv_launch_ring3_f16(const rsa_conv_params& p, hipStream_t stream) {
  if (p.products == 3) return conv_launch_ring3_f16x3(p, stream);
  if (conv_ring_xres_enabled() && conv_ring_span_eligible(p)) return conv_launch_ring3_span(p, stream);  // conv_inst_ring3hs.hip: weights in LDS, direct epilogue
  if (p.out_nchw != nullptr) return (p.cin_planes & 3) ? launch_ring<3, 0, 1, 1, RSA_PF_F16, 1>(p, stream) : launch_ring<3, 0, 1, 0, RSA_PF_F16, 1>(p, stream);
  return (p.cin_planes & 3) ? launch_ring<3, 0, 0, 1, RSA_PF_F16, 1>(p, stream) : launch_ring<3, 0, 0, 0, RSA_PF_F16, 1>(p, stream);
}
unsigned int conv_ring3h_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring3h_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
