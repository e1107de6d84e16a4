// conv_inst_ring2.hip — conv_ring<SHAPE = 2> (two streams) (17..32 output channels: the growth convolutions of a residual dense block).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring2(const rsa_conv_params& p, hipStream_t stream) {
  if (p.out_nchw != nullptr)  // final store of a layer with at most 32 output channels (64 -> 3: one cout tile, XRES 6)
    return (p.cout <= 16 && conv_ring_xres_enabled()) ? launch_ring<2, 0, 1, 0, 0, 3, 6>(p, stream) : launch_ring<2, 0, 1>(p, stream);  // (RSA_RING_XRES=0: A/B runs)
  return p.upsample2x ? launch_ring<2, 1, 0>(p, stream) : launch_ring<2, 0, 0>(p, stream);
}
unsigned int conv_ring2_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring2_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
