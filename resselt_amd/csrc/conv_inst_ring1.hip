// conv_inst_ring1.hip — conv_ring<SHAPE = 1> (49..64 output channels) and the dispatch of the ring schedule.
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring2(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring2.hip
int conv_launch_ring3(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring3.hip
int conv_launch_ring1_f16(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring1h.hip
int conv_launch_ring2_f16(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring2h.hip
int conv_launch_ring3_f16(const rsa_conv_params& p, hipStream_t stream);  // conv_inst_ring3h.hip

unsigned int conv_ring2_aborts();     // conv_inst_ring2.hip
unsigned int conv_ring3_aborts();     // conv_inst_ring3.hip
unsigned int conv_ring_up2_aborts();  // conv_inst_ringup.hip
unsigned int conv_ring1h_aborts();
unsigned int conv_ring2h_aborts();
unsigned int conv_ring3h_aborts();
unsigned int conv_ring3hx_aborts();
unsigned int conv_ring3hs_aborts();

int conv_launch_ring(const rsa_conv_params& p, hipStream_t stream) {
  if (p.w_layout == RSA_WL_UPPHASE) return conv_launch_ring_up2(p, stream);
  const int ct = (p.cout + 15) >> 4;
  if (p.in_fmt == RSA_PF_F16) return ct == 2 ? conv_launch_ring2_f16(p, stream) : ct == 3 ? conv_launch_ring3_f16(p, stream) : conv_launch_ring1_f16(p, stream);
  if (ct <= 2) return conv_launch_ring2(p, stream);
  if (ct == 3) return conv_launch_ring3(p, stream);
  return p.upsample2x ? launch_ring<1, 1, 0>(p, stream) : launch_ring<1, 0, 0>(p, stream);
}

unsigned int conv_ring_aborts() {
  return ring_aborts_this_unit() + conv_ring2_aborts() + conv_ring3_aborts() + conv_ring_up2_aborts() + conv_ring1h_aborts() + conv_ring2h_aborts() + conv_ring3h_aborts() +
         conv_ring3hx_aborts() + conv_ring3hs_aborts() + conv_ring_pair_aborts();
}
}  // namespace rsa

#ifdef RSA_RING_DEBUG
namespace rsa {
int conv_ring2_set_dbg(unsigned v);
int conv_ring3_set_dbg(unsigned v);
int conv_ring1h_set_dbg(unsigned v);
int conv_ring2h_set_dbg(unsigned v);
int conv_ring3h_set_dbg(unsigned v);
int conv_ring3hx_set_dbg(unsigned v);
int conv_ringpair_set_dbg(unsigned v);
int conv_ring3hs_set_dbg(unsigned v);
}  // namespace rsa
extern "C" int rsa_debug_ring_flags(unsigned v) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(rsa::g_ring_dbg), &v, sizeof(v)) != hipSuccess) return -1;
  return rsa::conv_ring2_set_dbg(v) | rsa::conv_ring3_set_dbg(v) | rsa::conv_ring1h_set_dbg(v) | rsa::conv_ring2h_set_dbg(v) | rsa::conv_ring3h_set_dbg(v) | rsa::conv_ring3hx_set_dbg(v) | rsa::conv_ringpair_set_dbg(v) | rsa::conv_ring3hs_set_dbg(v);
}
#endif
