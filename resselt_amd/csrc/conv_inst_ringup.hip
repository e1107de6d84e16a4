// conv_inst_ringup.hip — conv_ring_up2: nearest x2 upsampling + 3x3 as four 2x2 phase convolutions (conv_ring_up.h).
#include "conv_ring_up.h"

namespace rsa {
int conv_launch_ring_up2(const rsa_conv_params& p, hipStream_t stream) { return launch_ring_up2(p, stream); }
unsigned int conv_ring_up2_aborts() { return ring_aborts_this_unit(); }
}  // namespace rsa
