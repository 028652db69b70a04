// Halo-tile convolution (halo_conv.hip): tried first by rehr_gather_gemm_f32.
#pragma once
#include "common.h"
// REHR_OK = launched; REHR_ENOSUP = not applicable (use the generic kernel); other = error.
int halo_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream);
