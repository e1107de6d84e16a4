// conv_inst_k3p1u1h.hip — instantiations of conv_kernel<KS=3, NCT, PROD=1, UP=1, OUTK, FMT=fp16> (own translation unit: parallel compile).
#include "conv_kernel.h"

namespace rsa {
int conv_launch_k3p1u1_f16(const rsa_conv_params& p, int nct, hipStream_t stream) { return launch_nct<3, 1, 1, RSA_PF_F16>(p, nct, stream); }
}  // namespace rsa
