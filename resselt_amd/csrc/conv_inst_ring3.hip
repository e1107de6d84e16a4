// conv_inst_ring3.hip — conv_ring<SHAPE = 3> (33..48 output channels: the 48-channel layers of the SPAN family).
#include "conv_ring.h"

namespace rsa {
int conv_launch_ring3(const rsa_conv_params& p, hipStream_t stream) {
  if (p.out_nchw != nullptr)  // final store (NCHW any dtype / 8-bit image, depth-to-space, affine): the x4 pixel-shuffle heads
    return (p.cin_planes & 3) ? launch_ring<3, 0, 1, 1>(p, stream) : launch_ring<3, 0, 1, 0>(p, stream);
  return (p.cin_planes & 3) ? launch_ring<3, 0, 0, 1>(p, stream) : launch_ring<3, 0, 0, 0>(p, stream);  // half mode for 1.5, 2.5 ... chunks
}
unsigned int conv_ring3_aborts() { return ring_aborts_this_unit(); }
#ifdef RSA_RING_DEBUG
int conv_ring3_set_dbg(unsigned v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_dbg), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
#endif
}  // namespace rsa
