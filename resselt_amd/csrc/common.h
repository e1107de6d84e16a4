// common.h — shared host-side helpers of the C-ABI library (error string, dispatch prototypes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "resselt_amd.h"

namespace rsa {

// Records a message for rsa_last_error_string() (thread-local) and returns `code`.
int set_error(int code, const char* msg);

// conv_mfma.hip
int conv_launch(const rsa_conv_params& p, hipStream_t stream);
int conv_nct(int cout);

}  // namespace rsa
