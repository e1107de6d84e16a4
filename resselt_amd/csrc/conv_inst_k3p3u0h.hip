// conv_inst_k3p3u0h.hip — instantiations of conv_kernel<KS=3, NCT, PROD=3, UP=0, OUTK, FMT=fp16> (own translation unit: parallel compile).
#include "conv_kernel_pp.h"

namespace rsa {
int conv_launch_k3p3u0_f16(const rsa_conv_params& p, int nct, hipStream_t stream) { return launch_nct_pp<3, 3, 0, RSA_PF_F16>(p, nct, stream); }
}  // namespace rsa
