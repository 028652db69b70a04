// Types shared by the two weight-gradient kernels (wgrad.hip: slab kernel + reduction,
// wgrad_brick.hip: LDS halo-brick kernel).
#pragma once
#include "common.h"

struct Magic {  // q = n / d for 0 <= n < 2^31
  uint32_t m;
  uint32_t sh;  // 255: d == 1
};

struct WGParams {
  rehr_wgrad_desc d;
  int a_tiles, c_tiles, T;
  int64_t kv_total;   // N * Ld*Lh*Lw
  int64_t kv_per_split;
  int splits;
  int Capad, Cgpad;   // tile-padded channel counts of the slab
  float* slab_bias;   // [splits][Ca] or null
  Magic mg_vox, mg_hw, mg_w;
  uint32_t g_bytes;
};

struct BrickPlanOut {
  int HD, HH, HW, mind, minh, minw, nb_d, nb_h, nb_w;
  int64_t nbricks;
  int bricks_per_split;
  size_t smem;
};

// brick plan of the mixed-precision weight gradient (wgrad_brick_bf16.hip)
struct BrickBf16 {
  int mind, minh, minw;      // halo origin relative to the lattice brick origin
  int nb_d, nb_h, nb_w, tiles_per_img, ntiles, tiles_per_block;
};
bool wgrad_brick_bf16_plan(const rehr_wgrad_desc& d, WGParams& w, BrickBf16& out);
int wgrad_brick_bf16_launch(const WGParams& w, const BrickBf16& o, hipStream_t stream);

bool wgrad_brick_plan(const rehr_wgrad_desc& d, WGParams& w, BrickPlanOut& out);
int wgrad_brick_launch(const WGParams& w, const BrickPlanOut& o, hipStream_t stream);

// Winograd weight gradient (wino_wgrad.hip): tried first for unit-stride 3x3 (H, W) taps
int64_t wino_wgrad_workspace_bytes(const rehr_wgrad_desc& d);  // 0 = not applicable
int wino_wgrad_try(const rehr_wgrad_desc& d, hipStream_t stream);
// F(2x2,2x2) weight gradient of stride-2 4-tap transposed convolutions (wino22_wgrad.hip)
int64_t wino22_wgrad_workspace_bytes(const rehr_wgrad_desc& d);  // 0 = not applicable
int wino22_wgrad_try(const rehr_wgrad_desc& d, hipStream_t stream);
