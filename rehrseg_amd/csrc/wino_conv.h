// Winograd F(2x2,3x3) x depth-direct convolution (wino_conv.hip): tried first by
// rehr_gather_gemm_f32 when the descriptor carries scratch for the transformed weights.
#pragma once
#include "common.h"
int64_t wino_workspace_bytes(const rehr_gather_gemm_desc& d);
int wino_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream);
