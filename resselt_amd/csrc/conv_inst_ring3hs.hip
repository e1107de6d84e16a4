// conv_inst_ring3hs.hip — conv_ring<SHAPE = 3, half mode, XRES = 3>: the re-parameterised 48 -> 48 layers of the SPAN family (reference
// archs/spanplus/arch.py:94-130, archs/span/arch.py:152-180) in one fp16 product with the weight blob resident in LDS and one epilogue
// instantiation per activation class.
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring3_span(const rsa_conv_params& p, hipStream_t stream) {
  switch (p.act) {
    case RSA_ACT_MISH:
      return launch_ring<3, 0, 0, 1, RSA_PF_F16, 1, 3, AC_MISH>(p, stream);
    case RSA_ACT_SILU:
      return launch_ring<3, 0, 0, 1, RSA_PF_F16, 1, 3, AC_SILU>(p, stream);
    case RSA_ACT_SPAB_GATE:
      return launch_ring<3, 0, 0, 1, RSA_PF_F16, 1, 3, AC_GATE>(p, stream);
    default:
      return launch_ring<3, 0, 0, 1, RSA_PF_F16, 1, 3, AC_LINEAR>(p, stream);
  }
}
unsigned int conv_ring3hs_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring3hs_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
