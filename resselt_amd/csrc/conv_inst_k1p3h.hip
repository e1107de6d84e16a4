// conv_inst_k1p3h.hip — instantiations of conv_kernel<KS=1, NCT, PROD=3, UP=0, OUTK, FMT=fp16> (own translation unit: parallel compile).
#include "conv_kernel.h"

namespace rsa {
int conv_launch_k1p3_f16(const rsa_conv_params& p, int nct, hipStream_t stream) { return launch_nct<1, 3, 0, RSA_PF_F16>(p, nct, stream); }
}  // namespace rsa
