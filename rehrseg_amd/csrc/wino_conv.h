// Winograd F(2x2,3x3) x depth-direct convolution (wino_conv.hip): tried first by
// rehr_gather_gemm_f32 when the descriptor carries scratch for the transformed weights.
#pragma once
#include "common.h"
int64_t wino_workspace_bytes(const rehr_gather_gemm_desc& d);
int wino_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream);
// split-K parts (depth-tap ranges of one layer, one slab each) in one launch of the big-tile kernel
int wino_conv_split_try(const rehr_gather_gemm_desc* ds, int count, hipStream_t stream);
// F(2x2,2x2) variant for 2-tap phases / stride-2 4-tap gathers (wino22_conv.hip)
int64_t wino22_workspace_bytes(const rehr_gather_gemm_desc& d);
int wino22_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream);
// the output phases of one transposed convolution in one grid of the flattened-tile F(2x2,2x2) kernel
int wino22_flat_multi_try(const rehr_gather_gemm_desc* ds, int count, hipStream_t stream);
// kernel == stride transposed convolution: all stride phases of a 128-voxel input tile in one block (tconv_ks.hip)
int tconv_ks_try(const rehr_gather_gemm_desc* ds, int count, bool bf16, hipStream_t stream);
// fragment-ordered weight transform shared by the big-tile, 32-channel-tile and flattened-tile kernels
int wino_weights_frag_launch(const rehr_gather_gemm_desc& d, int kchunks, hipStream_t stream);
// small planes (12 x 12, 24 x 24, ...): flattened tile numbering, the big-tile kernel's schedule, row-range staging
// (wino_flat8_conv.hip); tried first
int64_t wino_flat8_workspace_bytes(const rehr_gather_gemm_desc& d);
int wino_flat8_conv_try(const rehr_gather_gemm_desc& d, hipStream_t stream);
