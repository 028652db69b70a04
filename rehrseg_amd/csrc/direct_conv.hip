// Direct (non-MFMA) convolutions for thin channel counts.  These layers have
// almost no arithmetic per byte (Cin in {1,2} or Cout <= 4), so they are written
// for the HBM roofline: NDHWC, 16-byte stores, weights staged once per block in
// LDS and read with wave-uniform (broadcast) addresses.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int DC_THREADS = 256;

// ------------------------------------------------------------------ small Cin, forward
// wave -> 16-channel group (and voxel sub-tile when Cout < 64); lane -> voxel.
__global__ __launch_bounds__(DC_THREADS) void small_cin_fwd_kernel(const rehr_direct_conv_desc d,
                                                                   int groups, int tiles_per_img) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [Cin*T][Cout]
  const int T = d.KD * d.KH * d.KW;
  const int tid = threadIdx.x;
  for (int i = tid; i < d.Cout * d.Cin * T; i += DC_THREADS) {
    const int co = i / (d.Cin * T), rem = i - co * (d.Cin * T);
    wl[rem * d.Cout + co] = d.w[i];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int g = wave % groups, vt = wave / groups;
  const int vpb = 64 * (4 / groups);
  const int n = blockIdx.y;
  const int64_t ovox = (int64_t)d.Do * d.Ho * d.Wo;
  const int how = d.Ho * d.Wo;
  const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
  float* yn = d.y + (int64_t)n * ovox * d.ldy;

  float bv[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bv[j] = d.bias ? d.bias[g * 16 + j] : 0.f;
  float s1[16], s2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  for (int tile = blockIdx.x; tile < tiles_per_img; tile += gridDim.x) {
    const int64_t v = (int64_t)tile * vpb + vt * 64 + lane;
    const bool valid = v < ovox;
    int od = 0, oh = 0, ow = 0;
    if (valid) {
      od = (int)(v / how);
      const int rem = (int)(v - (int64_t)od * how);
      oh = rem / d.Wo;
      ow = rem - oh * d.Wo;
    }
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bv[j];
    const int id0 = od * d.sd - d.pd, ih0 = oh * d.sh - d.ph, iw0 = ow * d.sw - d.pw;
    for (int kd = 0; kd < d.KD; ++kd) {
      const int id = id0 + kd;
      for (int kh = 0; kh < d.KH; ++kh) {
        const int ih = ih0 + kh;
        const bool rowok = valid && (unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi;
        const float* xr = xn + ((int64_t)id * d.Hi + ih) * d.Wi * d.ldx;
        for (int kw = 0; kw < d.KW; ++kw) {
          const int iw = iw0 + kw;
          const bool ok = rowok && (unsigned)iw < (unsigned)d.Wi;
          const int t = (kd * d.KH + kh) * d.KW + kw;
          for (int ci = 0; ci < d.Cin; ++ci) {
            const float xv = ok ? xr[(int64_t)iw * d.ldx + ci] : 0.f;
            const f32x4* wp = reinterpret_cast<const f32x4*>(wl + (ci * T + t) * d.Cout + g * 16);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const f32x4 wv = wp[k4];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[k4 * 4 + e] += xv * wv[e];
            }
          }
        }
      }
    }
    if (valid) {
      float* yo = yn + v * d.ldy + g * 16;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = apply_act(acc[k4 * 4 + e], d.act, d.slope);
          o[e] = r;
          s1[k4 * 4 + e] += r;
          s2[k4 * 4 + e] += r * r;
        }
        *reinterpret_cast<f32x4*>(yo + k4 * 4) = o;
      }
    }
  }
  if (d.stats_mode != 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float a = wave_sum(s1[j]);
      const float b = wave_sum(s2[j]);
      if (lane == 0) {
        double* st = d.stats + ((int64_t)n * d.Cout + g * 16 + j) * 2;
        atomicAdd(st, (double)a);
        if (d.stats_mode == 2) atomicAdd(st + 1, (double)b);
      }
    }
  }
}

// ------------------------------------------------------------------ small Cin, weight gradient
// lanes <-> output channels (Cout <= 64), waves split the taps, blocks split the
// voxels; the x sample of a (voxel, tap) is wave-uniform.  Partials go to
// slab[block][Cin*T][Cout] (+ bias row) and are reduced deterministically.
template <int TPW>  // taps per wave, compile-time bound for the register accumulators
__global__ __launch_bounds__(DC_THREADS) void small_cin_wgrad_kernel(const rehr_direct_conv_desc d,
                                                                     float* __restrict__ slab,
                                                                     int64_t vox_per_block) {
  const int T = d.KD * d.KH * d.KW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpw = (T + 3) / 4;
  const int t_begin = wave * tpw;
  const int64_t ovox = (int64_t)d.Do * d.Ho * d.Wo;
  const int64_t total = (int64_t)d.N * ovox;
  const int64_t v_begin = (int64_t)blockIdx.x * vox_per_block;
  int64_t v_end = v_begin + vox_per_block;
  if (v_end > total) v_end = total;
  const int how = d.Ho * d.Wo;
  const bool lane_ok = lane < d.Cout;

  float acc[2][TPW];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[ci][j] = 0.f;
  float bsum = 0.f;

  for (int64_t v = v_begin; v < v_end; ++v) {
    const int n = (int)(v / ovox);
    const int64_t r0 = v - (int64_t)n * ovox;
    const int od = (int)(r0 / how);
    const int rem = (int)(r0 - (int64_t)od * how);
    const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
    const float dyv = lane_ok ? d.y[v * d.ldy + lane] : 0.f;
    bsum += dyv;
    const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
    const int id0 = od * d.sd - d.pd, ih0 = oh * d.sh - d.ph, iw0 = ow * d.sw - d.pw;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int t = t_begin + j;
      if (j < tpw && t < T) {
        const int kw = t % d.KW, kh = (t / d.KW) % d.KH, kd = t / (d.KW * d.KH);
        const int id = id0 + kd, ih = ih0 + kh, iw = iw0 + kw;
        if ((unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi && (unsigned)iw < (unsigned)d.Wi) {
          const float* xp = xn + (((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx;
          acc[0][j] += dyv * xp[0];
          if (d.Cin > 1) acc[1][j] += dyv * xp[1];
        }
      }
    }
  }
  float* sb = slab + (int64_t)blockIdx.x * ((int64_t)d.Cin * T + 1) * 64;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = t_begin + j;
    if (j < tpw && t < T) {
      sb[(int64_t)(0 * T + t) * 64 + lane] = acc[0][j];
      if (d.Cin > 1) sb[(int64_t)(1 * T + t) * 64 + lane] = acc[1][j];
    }
  }
  if (wave == 0) sb[(int64_t)d.Cin * T * 64 + lane] = bsum;
}

__global__ void small_cin_wgrad_reduce_kernel(const float* __restrict__ slab, int nblocks, int Cin,
                                              int T, int Cout, float* __restrict__ dw,
                                              float* __restrict__ dbias) {
  const int rows = Cin * T + 1;
  const int total = rows * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i & 63, r = i >> 6;
    if (co >= Cout) continue;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += slab[((int64_t)b * rows + r) * 64 + co];
    if (r == Cin * T) {
      if (dbias) dbias[co] = s;
    } else {
      const int ci = r / T, t = r - ci * T;
      dw[((int64_t)co * Cin + ci) * T + t] = s;
    }
  }
}

bool small_cin_ok(const rehr_direct_conv_desc& d) {
  if (!d.x || !d.w || !d.y) return false;
  if (d.Cin < 1 || d.Cin > 2) return false;
  if (d.Cout != 16 && d.Cout != 32 && d.Cout != 64) return false;
  if (d.ldy % 4 || (((uintptr_t)d.y) & 15)) return false;
  if (d.N < 1 || d.N > 65535) return false;
  if (d.KD < 1 || d.KH < 1 || d.KW < 1 || d.sd < 1 || d.sh < 1 || d.sw < 1) return false;
  // output extent must follow from the input extent
  if ((d.Di + 2 * d.pd - d.KD) / d.sd + 1 != d.Do) return false;
  if ((d.Hi + 2 * d.ph - d.KH) / d.sh + 1 != d.Ho) return false;
  if ((d.Wi + 2 * d.pw - d.KW) / d.sw + 1 != d.Wo) return false;
  return true;
}

int wgrad_blocks(const rehr_direct_conv_desc& d, int64_t* vpb) {
  const int64_t total = (int64_t)d.N * d.Do * d.Ho * d.Wo;
  int64_t blocks = 1024;
  int64_t per = (total + blocks - 1) / blocks;
  if (per < 64) per = 64;
  blocks = (total + per - 1) / per;
  *vpb = per;
  return (int)blocks;
}

}  // namespace

int thin_cin_fwd_try(const rehr_direct_conv_desc& d, hipStream_t stream, bool y_bf16);   // thin_cin_conv.hip: fp32 matrix cores
int64_t thin_cin_wgrad_workspace_bytes(const rehr_direct_conv_desc& d);
int thin_cin_wgrad_try(const rehr_direct_conv_desc& d, float* dw, float* dbias, float* workspace, int64_t workspace_bytes,
                       hipStream_t stream, bool dy_bf16);

extern "C" int rehr_conv_small_cin_fwd_f32(const rehr_direct_conv_desc* dp, void* stream) {
  if (!dp || !small_cin_ok(*dp)) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (d.stats_mode != 0 && !d.stats) return REHR_EINVAL;
  {
    const int rc = thin_cin_fwd_try(d, (hipStream_t)stream, false);   // C_out 32 / 64, kW <= 8, stride_w <= 2
    if (rc != REHR_ENOSUP) return rc;
  }
  const int T = d.KD * d.KH * d.KW;
  const size_t smem = (size_t)d.Cin * T * d.Cout * sizeof(float);
  if (smem > 150 * 1024) return REHR_ENOSUP;
  if (smem > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_cin_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        return REHR_EHIP;
      attr_set = true;
    }
  }
  const int groups = d.Cout / 16;
  const int vpb = 64 * (4 / groups);
  const int64_t ovox = (int64_t)d.Do * d.Ho * d.Wo;
  const int tiles = (int)((ovox + vpb - 1) / vpb);
  int bx = tiles;
  // with a statistics epilogue every wave ends in 16-32 double atomics on the same N*Cout addresses: 4096 blocks
  // serialised 8192 atomics per address (1.4 ms for a 0.1 ms conv); 512 persistent blocks keep the chip full
  const int maxb = d.stats_mode ? 512 : 4096;
  const int cap = maxb / d.N > 0 ? maxb / d.N : 1;
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL(small_cin_fwd_kernel, dim3(bx, d.N), dim3(DC_THREADS), smem, (hipStream_t)stream, d,
                     groups, tiles);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv_small_cin_wgrad_f32(const rehr_direct_conv_desc* dp, float* dw, float* dbias,
                                             float* workspace, int64_t workspace_bytes, void* stream) {
  if (!dp || !small_cin_ok(*dp) || !dw || !workspace) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  {
    const int rc = thin_cin_wgrad_try(d, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream, false);   // matrix cores
    if (rc != REHR_ENOSUP) return rc;
  }
  const int T = d.KD * d.KH * d.KW;
  int64_t vpb;
  const int blocks = wgrad_blocks(d, &vpb);
  const int64_t need = (int64_t)blocks * ((int64_t)d.Cin * T + 1) * 64 * sizeof(float);
  if (workspace_bytes < need) return REHR_EINVAL;
  const int tpw = (T + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (tpw <= 8)
    hipLaunchKernelGGL(small_cin_wgrad_kernel<8>, dim3(blocks), dim3(DC_THREADS), 0, st, d, workspace, vpb);
  else if (tpw <= 40)
    hipLaunchKernelGGL(small_cin_wgrad_kernel<40>, dim3(blocks), dim3(DC_THREADS), 0, st, d, workspace, vpb);
  else
    return REHR_ENOSUP;
  hipLaunchKernelGGL(small_cin_wgrad_reduce_kernel, dim3(((d.Cin * T + 1) * 64 + 255) / 256), dim3(256), 0,
                     st, workspace, blocks, d.Cin, T, d.Cout, dw, dbias);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// Mixed precision (BASELINE configs[4]): the thin-input layers compute in fp32 (their input is the fp32 image) but the
// layer behind takes bf16 activations and hands back bf16 gradients -- y / dY of the descriptor point at bf16 elements
// (ldy in elements), so no separate cast pass runs over the largest activation of the network.  Matrix-core shapes only.
extern "C" int rehr_conv_small_cin_fwd_ybf16(const rehr_direct_conv_desc* dp, void* stream) {
  if (!dp || !dp->x || !dp->w || !dp->y) return REHR_EINVAL;
  if (dp->stats_mode != 0 && !dp->stats) return REHR_EINVAL;
  return thin_cin_fwd_try(*dp, (hipStream_t)stream, true);
}
extern "C" int rehr_conv_small_cin_wgrad_dybf16(const rehr_direct_conv_desc* dp, float* dw, float* dbias,
                                                float* workspace, int64_t workspace_bytes, void* stream) {
  if (!dp || !dp->x || !dp->w || !dp->y || !dw || !workspace) return REHR_EINVAL;
  return thin_cin_wgrad_try(*dp, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream, true);
}

extern "C" int rehr_conv_small_cin_wgrad_on_mfma(const rehr_direct_conv_desc* dp) {
  return dp && small_cin_ok(*dp) && thin_cin_wgrad_workspace_bytes(*dp) > 0;
}

extern "C" int64_t rehr_conv_small_cin_wgrad_workspace_bytes(const rehr_direct_conv_desc* dp) {
  if (!dp) return REHR_EINVAL;
  {
    const int64_t b = small_cin_ok(*dp) ? thin_cin_wgrad_workspace_bytes(*dp) : 0;
    if (b > 0) return b;
  }
  int64_t vpb;
  const int blocks = wgrad_blocks(*dp, &vpb);
  const int T = dp->KD * dp->KH * dp->KW;
  return (int64_t)blocks * ((int64_t)dp->Cin * T + 1) * 64 * sizeof(float);
}

// ------------------------------------------------------------------ im2col for thin inputs
// out[voxel][k], k = ci*T + tap (the weight's own (Cin,kD,kH,kW) order), zero for
// k >= Cin*T and for taps in the padding.  With it the weight gradient of a thin-input
// conv is a plain 1x1x1 weight gradient on the matrix cores (rehr_wgrad_f32).
namespace {
__global__ void im2col_kernel(const rehr_direct_conv_desc d, float* __restrict__ out, int Kpad) {
  extern __shared__ int lut[];  // [Kpad]: ci<<24 | kd<<16 | kh<<8 | kw, -1 for padding columns
  const int T = d.KD * d.KH * d.KW, K = d.Cin * T;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    int v = -1;
    if (k < K) {
      const int ci = k / T, t = k - ci * T;
      const int kw = t % d.KW, kh = (t / d.KW) % d.KH, kd = t / (d.KW * d.KH);
      v = (ci << 24) | (kd << 16) | (kh << 8) | kw;
    }
    lut[k] = v;
  }
  __syncthreads();
  const uint32_t G = (uint32_t)Kpad / 4;
  const int64_t ovox = (int64_t)d.Do * d.Ho * d.Wo;
  const int64_t total = (int64_t)d.N * ovox * G;
  const int how = d.Ho * d.Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / G;
    const int g = (int)(i - v * G);
    const int n = (int)(v / ovox);
    const int64_t r0 = v - (int64_t)n * ovox;
    const int od = (int)(r0 / how);
    const int rem = (int)(r0 - (int64_t)od * how);
    const int oh = rem / d.Wo, ow = rem - oh * d.Wo;
    const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int code = lut[g * 4 + e];
      float val = 0.f;
      if (code >= 0) {
        const int ci = code >> 24, kd = (code >> 16) & 255, kh = (code >> 8) & 255, kw = code & 255;
        const int id = od * d.sd - d.pd + kd, ih = oh * d.sh - d.ph + kh, iw = ow * d.sw - d.pw + kw;
        if ((unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi && (unsigned)iw < (unsigned)d.Wi)
          val = xn[(((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx + ci];
      }
      o[e] = val;
    }
    *reinterpret_cast<f32x4*>(out + v * Kpad + g * 4) = o;
  }
}
}  // namespace

extern "C" int rehr_im2col_f32(const rehr_direct_conv_desc* dp, float* out, int32_t Kpad, void* stream) {
  if (!dp || !out) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  if (!d.x || d.Cin < 1 || d.Cin > 127 || d.N < 1) return REHR_EINVAL;
  if (d.KD < 1 || d.KH < 1 || d.KW < 1 || d.KD > 255 || d.KH > 255 || d.KW > 255) return REHR_EINVAL;
  if (Kpad % 4 || Kpad < d.Cin * d.KD * d.KH * d.KW || (((uintptr_t)out) & 15)) return REHR_EINVAL;
  if ((d.Di + 2 * d.pd - d.KD) / d.sd + 1 != d.Do || (d.Hi + 2 * d.ph - d.KH) / d.sh + 1 != d.Ho ||
      (d.Wi + 2 * d.pw - d.KW) / d.sw + 1 != d.Wo)
    return REHR_EINVAL;
  const size_t smem = (size_t)Kpad * sizeof(int);
  if (smem > 48 * 1024) return REHR_ENOSUP;
  const int64_t total = (int64_t)d.N * d.Do * d.Ho * d.Wo * (Kpad / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, d, out, Kpad);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

// ------------------------------------------------------------------ thin-output convolutions
// Cout <= 4 (segmentation logits: 1x1x1 seg layer, sr_head's 5x5x5 16->2), stride 1.
// HBM/VALU bound.  A thread owns 4 consecutive output voxels along W; for every
// (kd,kh) it loads the row segment it needs once into registers and slides the KW
// taps over it, with the weights of that (kd,kh) broadcast from LDS.
namespace {

constexpr int SC_VOX = 8;     // voxels per thread along W
constexpr int SC_MAXKW = 7;
constexpr int SC_BIAS_BLOCKS = 1024;

// forward: y[o][co] = b[co] + sum x[o - p + k][ci] * w[co][ci][k]
template <int CO>
__global__ __launch_bounds__(256) void small_cout_fwd_kernel(const rehr_direct_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [T][CO][Cin]
  const int T = d.KD * d.KH * d.KW;
  for (int i = threadIdx.x; i < CO * d.Cin * T; i += 256) {
    const int co = i / (d.Cin * T), rem = i - co * d.Cin * T;
    const int ci = rem / T, t = rem - ci * T;
    wl[(t * CO + co) * d.Cin + ci] = (co < d.Cout) ? d.w[((int64_t)co * d.Cin + ci) * T + t] : 0.f;
  }
  __syncthreads();
  const int wg = (d.Wo + SC_VOX - 1) / SC_VOX;
  const int64_t groups = (int64_t)d.N * d.Do * d.Ho * wg;
  for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
    const int gw = (int)(gidx % wg);
    int64_t r = gidx / wg;
    const int oh = (int)(r % d.Ho); r /= d.Ho;
    const int od = (int)(r % d.Do);
    const int n = (int)(r / d.Do);
    const int ow0 = gw * SC_VOX;
    float acc[SC_VOX][CO];
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[v][c] = (d.bias && c < d.Cout) ? d.bias[c] : 0.f;
    const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
    for (int kd = 0; kd < d.KD; ++kd) {
      const int id = od - d.pd + kd;
      if ((unsigned)id >= (unsigned)d.Di) continue;
      for (int kh = 0; kh < d.KH; ++kh) {
        const int ih = oh - d.ph + kh;
        if ((unsigned)ih >= (unsigned)d.Hi) continue;
        const float* xr = xn + ((int64_t)id * d.Hi + ih) * d.Wi * d.ldx;
        const float* wr = wl + (kd * d.KH + kh) * d.KW * CO * d.Cin;
        for (int c4 = 0; c4 < d.Cin; c4 += 4) {
          f32x4 seg[SC_VOX + SC_MAXKW - 1];
#pragma unroll
          for (int j = 0; j < SC_VOX + SC_MAXKW - 1; ++j) {
            const int iw = ow0 - d.pw + j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < SC_VOX + d.KW - 1 && (unsigned)iw < (unsigned)d.Wi)
              v = *reinterpret_cast<const f32x4*>(xr + (int64_t)iw * d.ldx + c4);
            seg[j] = v;
          }
#pragma unroll
          for (int kw = 0; kw < SC_MAXKW; ++kw) {
            if (kw < d.KW) {
#pragma unroll
              for (int c = 0; c < CO; ++c) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + (kw * CO + c) * d.Cin + c4);
#pragma unroll
                for (int v = 0; v < SC_VOX; ++v) {
                  const f32x4 xv = seg[v + kw];
                  acc[v][c] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
                }
              }
            }
          }
        }
      }
    }
    float* yo = d.y + ((((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo + ow0) * d.ldy;
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
      if (ow0 + v < d.Wo)
#pragma unroll
        for (int c = 0; c < CO; ++c)
          if (c < d.Cout) yo[(int64_t)v * d.ldy + c] = apply_act(acc[v][c], d.act, d.slope);
  }
}

// input gradient: dx[i][ci] = sum_{k,co} dy[i + p - k][co] * w[co][ci][k]   (stride 1)
template <int CO>
__global__ __launch_bounds__(256) void small_cout_dgrad_kernel(const rehr_direct_conv_desc d,
                                                               float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [T][CO][Cin]
  const int T = d.KD * d.KH * d.KW;
  for (int i = threadIdx.x; i < CO * d.Cin * T; i += 256) {
    const int co = i / (d.Cin * T), rem = i - co * d.Cin * T;
    const int ci = rem / T, t = rem - ci * T;
    wl[(t * CO + co) * d.Cin + ci] = (co < d.Cout) ? d.w[((int64_t)co * d.Cin + ci) * T + t] : 0.f;
  }
  __syncthreads();
  const int wg = (d.Wi + SC_VOX - 1) / SC_VOX;
  const int c16n = d.Cin / 16;
  // (SC_VOX is even, so a thread's first column iw0 + pw - (KW-1) is even iff pw - (KW-1) is)
  const bool pairs16 = d.ldy == 2 && d.Cout == 2 && !(d.Wo & 1) && !((d.pw - (d.KW - 1)) & 1) &&
                       !((uintptr_t)d.y & 15);
  const int64_t groups = (int64_t)d.N * d.Di * d.Hi * wg * c16n;
  for (int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * 256) {
    const int gw = (int)(gidx % wg);
    int64_t r = gidx / wg;
    const int cb = (int)(r % c16n) * 16; r /= c16n;
    const int ih = (int)(r % d.Hi); r /= d.Hi;
    const int id = (int)(r % d.Di);
    const int n = (int)(r / d.Di);
    const int iw0 = gw * SC_VOX;
    f32x4 acc[SC_VOX][4];
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[v][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* yn = d.y + (int64_t)n * d.Do * d.Ho * d.Wo * d.ldy;
    for (int kd = 0; kd < d.KD; ++kd) {
      const int od = id + d.pd - kd;
      if ((unsigned)od >= (unsigned)d.Do) continue;
      for (int kh = 0; kh < d.KH; ++kh) {
        const int oh = ih + d.ph - kh;
        if ((unsigned)oh >= (unsigned)d.Ho) continue;
        const float* yr = yn + ((int64_t)od * d.Ho + oh) * d.Wo * d.ldy;
        const float* wr = wl + (kd * d.KH + kh) * d.KW * CO * d.Cin;
        // dy row segment: positions iw0 + p - (KW-1) .. iw0 + p + SC_VOX - 1
        float seg[SC_VOX + SC_MAXKW - 1][CO];
        const int ow_first = iw0 + d.pw - (d.KW - 1);
        if (CO == 2 && pairs16) {
          // two-channel dY rows are contiguous: the thread's segment in 16-byte loads (two voxels each; the
          // segment starts on an even column and Wo is even, so a pair is inside or outside as a whole),
          // clamped address + select instead of a branch
#pragma unroll
          for (int j = 0; j < SC_VOX + SC_MAXKW - 1; j += 2) {
            const int ow = ow_first + j;
            const bool ok = (j < SC_VOX + d.KW - 1) & ((unsigned)ow < (unsigned)d.Wo);
            const f32x4 ld = *reinterpret_cast<const f32x4*>(yr + (int64_t)(ok ? ow : 0) * 2);
            seg[j][0] = ok ? ld[0] : 0.f;
            seg[j][1] = ok ? ld[1] : 0.f;
            if (j + 1 < SC_VOX + SC_MAXKW - 1) {
              seg[j + 1][0] = ok ? ld[2] : 0.f;
              seg[j + 1][1] = ok ? ld[3] : 0.f;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < SC_VOX + SC_MAXKW - 1; ++j) {
            const int ow = ow_first + j;
            const bool ok = j < SC_VOX + d.KW - 1 && (unsigned)ow < (unsigned)d.Wo;
#pragma unroll
            for (int c = 0; c < CO; ++c) seg[j][c] = (ok && c < d.Cout) ? yr[(int64_t)ow * d.ldy + c] : 0.f;
          }
        }
#pragma unroll
        for (int kw = 0; kw < SC_MAXKW; ++kw) {
          if (kw < d.KW) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + (kw * CO + c) * d.Cin + cb + q * 4);
#pragma unroll
                for (int v = 0; v < SC_VOX; ++v) acc[v][q] += wv * seg[v + (d.KW - 1) - kw][c];
              }
            }
          }
        }
      }
    }
    float* xo = dx + ((((int64_t)n * d.Di + id) * d.Hi + ih) * d.Wi + iw0) * d.ldx + cb;
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
      if (iw0 + v < d.Wi)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(xo + (int64_t)v * d.ldx + q * 4) = acc[v][q];
  }
}

// weight gradient: dw[co][ci][k] = sum dy[o][co] * x[o - p + k][ci].  A block owns one
// (kd,kh) and a 4-channel slice of ci; a thread owns output positions along W with lanes
// on consecutive w (coalesced) and walks a strip of (n,d,h) rows, keeping KW*4*CO sums.
template <int CO>
__global__ __launch_bounds__(256) void small_cout_wgrad_kernel(const rehr_direct_conv_desc d,
                                                               float* __restrict__ slab, int rows_per_block) {
  __shared__ float red[256];
  const int c4n = d.Cin / 4;
  int b = blockIdx.x;
  const int c4 = (b % c4n) * 4; b /= c4n;
  const int kh = b % d.KH; b /= d.KH;
  const int kd = b % d.KD; b /= d.KD;
  const int strip = b;  // strip of rows
  const int64_t nrows = (int64_t)d.N * d.Do * d.Ho;
  const int64_t row0 = (int64_t)strip * rows_per_block;
  float acc[SC_MAXKW][4][CO];
#pragma unroll
  for (int kw = 0; kw < SC_MAXKW; ++kw)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[kw][e][c] = 0.f;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int64_t row = row0 + rr;
    if (row >= nrows) break;
    const int oh = (int)(row % d.Ho);
    const int64_t r2 = row / d.Ho;
    const int od = (int)(r2 % d.Do);
    const int n = (int)(r2 / d.Do);
    const int id = od - d.pd + kd, ih = oh - d.ph + kh;
    if ((unsigned)id >= (unsigned)d.Di || (unsigned)ih >= (unsigned)d.Hi) continue;
    const float* yr = d.y + (((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo * d.ldy;
    const float* xr = d.x + (((int64_t)n * d.Di + id) * d.Hi + ih) * d.Wi * d.ldx + c4;
    for (int ow = threadIdx.x; ow < d.Wo; ow += 256) {
      float dyv[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) dyv[c] = (c < d.Cout) ? yr[(int64_t)ow * d.ldy + c] : 0.f;
#pragma unroll
      for (int kw = 0; kw < SC_MAXKW; ++kw) {
        const int iw = ow - d.pw + kw;
        if (kw < d.KW && (unsigned)iw < (unsigned)d.Wi) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + (int64_t)iw * d.ldx);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CO; ++c) acc[kw][e][c] += xv[e] * dyv[c];
        }
      }
    }
  }
  // block reduction of KW*4*CO sums -> slab[strip][co][ci][tap]
  const int T = d.KD * d.KH * d.KW;
  float* sb = slab + (int64_t)strip * d.Cout * d.Cin * T;
#pragma unroll
  for (int kw = 0; kw < SC_MAXKW; ++kw) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        if (kw < d.KW && c < d.Cout) {  // block-uniform
          float v = wave_sum(acc[kw][e][c]);
          __syncthreads();
          if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
          __syncthreads();
          if (threadIdx.x == 0)
            sb[((int64_t)c * d.Cin + c4 + e) * T + (kd * d.KH + kh) * d.KW + kw] = red[0] + red[1] + red[2] + red[3];
        }
      }
    }
  }
}


// ---- coalesced variants for Cin in {16, 32, 64}: CG = Cin/4 adjacent lanes share one voxel,
// each owning a 4-channel quad, so a wave-load is whole 64..256-byte voxel records instead
// of 64 scattered 16-byte pieces.  Partial sums are combined across the CG lanes at the end.
template <int CO, int CG>
__global__ __launch_bounds__(256) void small_cout_fwd_cg_kernel(const rehr_direct_conv_desc d) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [T][CO][Cin]; CO == 2: [T][Cin][2] (co pairs adjacent)
  const int T = d.KD * d.KH * d.KW;
  for (int i = threadIdx.x; i < CO * d.Cin * T; i += 256) {
    const int co = i / (d.Cin * T), rem = i - co * d.Cin * T;
    const int ci = rem / T, t = rem - ci * T;
    const float v = (co < d.Cout) ? d.w[((int64_t)co * d.Cin + ci) * T + t] : 0.f;
    if constexpr (CO == 2) wl[(t * d.Cin + ci) * 2 + co] = v;
    else wl[(t * CO + co) * d.Cin + ci] = v;
  }
  __syncthreads();
  const int c4 = (threadIdx.x % CG) * 4;
  const int wg = (d.Wo + SC_VOX - 1) / SC_VOX;
  const int64_t groups = (int64_t)d.N * d.Do * d.Ho * wg;
  constexpr int GPB = 256 / CG;  // voxel groups per block
  for (int64_t g0 = (int64_t)blockIdx.x * GPB; g0 < groups; g0 += (int64_t)gridDim.x * GPB) {
    const int64_t gidx = g0 + threadIdx.x / CG;
    const bool live = gidx < groups;
    const int64_t gi = live ? gidx : 0;
    const int gw = (int)(gi % wg);
    int64_t r = gi / wg;
    const int oh = (int)(r % d.Ho); r /= d.Ho;
    const int od = (int)(r % d.Do);
    const int n = (int)(r / d.Do);
    const int ow0 = gw * SC_VOX;
    float acc[SC_VOX][CO];
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[v][c] = 0.f;
    const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx + c4;
    for (int kd = 0; kd < d.KD; ++kd) {
      const int id = od - d.pd + kd;
      for (int kh = 0; kh < d.KH; ++kh) {
        const int ih = oh - d.ph + kh;
        const bool rowok = live && (unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi;
        const float* xr = xn + ((int64_t)id * d.Hi + ih) * d.Wi * d.ldx;
        const float* wr = wl + (kd * d.KH + kh) * d.KW * CO * d.Cin + (CO == 2 ? 2 * c4 : c4);
        f32x4 seg[SC_VOX + SC_MAXKW - 1];
#pragma unroll
        for (int j = 0; j < SC_VOX + SC_MAXKW - 1; ++j) {
          const int iw = ow0 - d.pw + j;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (rowok && j < SC_VOX + d.KW - 1 && (unsigned)iw < (unsigned)d.Wi)
            v = *reinterpret_cast<const f32x4*>(xr + (int64_t)iw * d.ldx);
          seg[j] = v;
        }
#pragma unroll
        for (int kw = 0; kw < SC_MAXKW; ++kw) {
          if (kw < d.KW) {
            if constexpr (CO == 2) {
              // both output channels of a (voxel, input channel) in one packed FMA: {x, x} * {w0, w1}
              typedef float f32x2_ __attribute__((ext_vector_type(2)));
              const f32x4 wa = *reinterpret_cast<const f32x4*>(wr + kw * 2 * d.Cin);      // ci c4, c4+1
              const f32x4 wb = *reinterpret_cast<const f32x4*>(wr + kw * 2 * d.Cin + 4);  // ci c4+2, c4+3
              const f32x2_ wp[4] = {{wa[0], wa[1]}, {wa[2], wa[3]}, {wb[0], wb[1]}, {wb[2], wb[3]}};
#pragma unroll
              for (int v = 0; v < SC_VOX; ++v) {
                const f32x4 xv = seg[v + kw];
                f32x2_ a = {acc[v][0], acc[v][1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) a = __builtin_elementwise_fma(f32x2_{xv[e], xv[e]}, wp[e], a);
                acc[v][0] = a[0];
                acc[v][1] = a[1];
              }
            } else {
#pragma unroll
              for (int c = 0; c < CO; ++c) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + (kw * CO + c) * d.Cin);
#pragma unroll
                for (int v = 0; v < SC_VOX; ++v) {
                  const f32x4 xv = seg[v + kw];
                  acc[v][c] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
                }
              }
            }
          }
        }
      }
    }
    // combine the CG channel quads of a voxel group (adjacent lanes)
#pragma unroll
    for (int v = 0; v < SC_VOX; ++v)
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        float a = acc[v][c];
#pragma unroll
        for (int o = CG / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        acc[v][c] = a;
      }
    if (live && (threadIdx.x % CG) == 0) {
      float* yo = d.y + ((((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo + ow0) * d.ldy;
#pragma unroll
      for (int v = 0; v < SC_VOX; ++v)
        if (ow0 + v < d.Wo)
#pragma unroll
          for (int c = 0; c < CO; ++c)
            if (c < d.Cout)
              yo[(int64_t)v * d.ldy + c] = apply_act(acc[v][c] + (d.bias ? d.bias[c] : 0.f), d.act, d.slope);
    }
  }
}

// weight gradient, same lane mapping: a block owns (kd, kh) and a strip of output rows;
// lane = (w position, channel quad); sums go through LDS atomics once per block.
template <int CO, int CG>
__global__ __launch_bounds__(256) void small_cout_wgrad_cg_kernel(const rehr_direct_conv_desc d,
                                                                  float* __restrict__ slab, int rows_per_block) {
  __shared__ float red[SC_MAXKW * 64 * 4];  // [kw][ci (<=64)][co (<=4)]
  for (int i = threadIdx.x; i < SC_MAXKW * 64 * 4; i += 256) red[i] = 0.f;
  __syncthreads();
  int b = blockIdx.x;
  const int kh = b % d.KH; b /= d.KH;
  const int kd = b % d.KD; b /= d.KD;
  const int strip = b;
  const int cq = threadIdx.x % CG, wl_ = threadIdx.x / CG;
  constexpr int WPB = 256 / CG;  // w positions per pass
  const int64_t nrows = (int64_t)d.N * d.Do * d.Ho;
  const int64_t row0 = (int64_t)strip * rows_per_block;
  float acc[SC_MAXKW][4][CO];
#pragma unroll
  for (int kw = 0; kw < SC_MAXKW; ++kw)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[kw][e][c] = 0.f;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int64_t row = row0 + rr;
    if (row >= nrows) break;
    const int oh = (int)(row % d.Ho);
    const int64_t r2 = row / d.Ho;
    const int od = (int)(r2 % d.Do);
    const int n = (int)(r2 / d.Do);
    const int id = od - d.pd + kd, ih = oh - d.ph + kh;
    if ((unsigned)id >= (unsigned)d.Di || (unsigned)ih >= (unsigned)d.Hi) continue;
    const float* yr = d.y + (((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo * d.ldy;
    const float* xr = d.x + (((int64_t)n * d.Di + id) * d.Hi + ih) * d.Wi * d.ldx + cq * 4;
    for (int ow = wl_; ow < d.Wo; ow += WPB) {
      float dyv[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) dyv[c] = (c < d.Cout) ? yr[(int64_t)ow * d.ldy + c] : 0.f;
#pragma unroll
      for (int kw = 0; kw < SC_MAXKW; ++kw) {
        const int iw = ow - d.pw + kw;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (kw < d.KW && (unsigned)iw < (unsigned)d.Wi) xv = *reinterpret_cast<const f32x4*>(xr + (int64_t)iw * d.ldx);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[kw][e][c] += xv[e] * dyv[c];
      }
    }
  }
#pragma unroll
  for (int kw = 0; kw < SC_MAXKW; ++kw)
    if (kw < d.KW)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CO; ++c)
          if (c < d.Cout) atomicAdd(&red[(kw * 64 + cq * 4 + e) * 4 + c], acc[kw][e][c]);
  __syncthreads();
  const int T = d.KD * d.KH * d.KW;
  float* sb = slab + (int64_t)strip * d.Cout * d.Cin * T;
  for (int i = threadIdx.x; i < d.KW * d.Cin * d.Cout; i += 256) {
    const int c = i % d.Cout, ci = (i / d.Cout) % d.Cin, kw = i / (d.Cout * d.Cin);
    sb[((int64_t)c * d.Cin + ci) * T + (kd * d.KH + kh) * d.KW + kw] = red[(kw * 64 + ci) * 4 + c];
  }
}


// ---- many-tap thin-output weight gradient through an LDS halo brick (sr_head's 5x5x5 16->2):
// a block stages a 2x8x8 brick of dY and the x halo around it in LDS ONCE and every
// (tap, channel quad) pair -- one or two per thread -- sweeps the brick from LDS, so x is read
// from L2/HBM once per brick instead of once per tap row (25x less traffic).
constexpr int HB_D = 2, HB_H = 8, HB_W = 8, HB_VOX = HB_D * HB_H * HB_W;
template <int CO>
__global__ __launch_bounds__(256) void small_cout_wgrad_halo_kernel(const rehr_direct_conv_desc d,
                                                                    float* __restrict__ slab, int bricks_per_block,
                                                                    int nb_d, int nb_h, int nb_w) {
  extern __shared__ __attribute__((aligned(16))) float hl[];
  const int HD = HB_D + d.KD - 1, HH = HB_H + d.KH - 1, HW = HB_W + d.KW - 1;
  const int hvox = HD * HH * HW;
  float* xs = hl;                      // [hvox][Cin]
  float* ys = hl + hvox * d.Cin;       // [HB_VOX][CO]
  const int T = d.KD * d.KH * d.KW, c4n = d.Cin / 4, npairs = T * c4n;
  // up to two (tap, quad) pairs per thread
  int tapbase[2], quad[2];
  bool has[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int pr = threadIdx.x + 256 * k;
    has[k] = pr < npairs;
    const int t = has[k] ? pr / c4n : 0;
    quad[k] = has[k] ? pr - t * c4n : 0;
    const int kw = t % d.KW, kh = (t / d.KW) % d.KH, kd = t / (d.KW * d.KH);
    tapbase[k] = (kd * HH + kh) * HW + kw;
  }
  f32x4 acc[2][CO];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[k][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t nbricks = (int64_t)d.N * nb_d * nb_h * nb_w;
  const int64_t b0 = (int64_t)blockIdx.x * bricks_per_block;
  for (int bi = 0; bi < bricks_per_block; ++bi) {
    const int64_t br = b0 + bi;
    if (br >= nbricks) break;
    const int bw = (int)(br % nb_w);
    int64_t r = br / nb_w;
    const int bh = (int)(r % nb_h); r /= nb_h;
    const int bd = (int)(r % nb_d);
    const int n = (int)(r / nb_d);
    const int od0 = bd * HB_D, oh0 = bh * HB_H, ow0 = bw * HB_W;
    __syncthreads();  // previous brick fully consumed
    // stage x halo (16-byte pieces, coalesced along channels then w)
    const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
    for (int i = threadIdx.x; i < hvox * c4n; i += 256) {
      const int q = i % c4n, hv = i / c4n;
      const int hw_ = hv % HW, hh_ = (hv / HW) % HH, hd_ = hv / (HW * HH);
      const int id = od0 - d.pd + hd_, ih = oh0 - d.ph + hh_, iw = ow0 - d.pw + hw_;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)id < (unsigned)d.Di && (unsigned)ih < (unsigned)d.Hi && (unsigned)iw < (unsigned)d.Wi)
        v = *reinterpret_cast<const f32x4*>(xn + (((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx + q * 4);
      *reinterpret_cast<f32x4*>(xs + hv * d.Cin + q * 4) = v;
    }
    for (int i = threadIdx.x; i < HB_VOX * CO; i += 256) {
      const int c = i % CO, v = i / CO;
      const int vw = v % HB_W, vh = (v / HB_W) % HB_H, vd = v / (HB_W * HB_H);
      const int od = od0 + vd, oh = oh0 + vh, ow = ow0 + vw;
      float val = 0.f;
      if (c < d.Cout && od < d.Do && oh < d.Ho && ow < d.Wo)
        val = d.y[((((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo + ow) * d.ldy + c];
      ys[i] = val;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (has[k]) {
        const float* xb = xs + tapbase[k] * d.Cin + quad[k] * 4;
#pragma unroll 4
        for (int v = 0; v < HB_VOX; ++v) {
          const int vrow = ((v >> 6) * HH + ((v >> 3) & 7)) * HW + (v & 7);
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + vrow * d.Cin);
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[k][c] += xv * ys[v * CO + c];
        }
      }
    }
  }
  float* sb = slab + (int64_t)blockIdx.x * d.Cout * d.Cin * T;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (has[k]) {
      const int pr = threadIdx.x + 256 * k;
      const int t = pr / c4n;
#pragma unroll
      for (int c = 0; c < CO; ++c)
        if (c < d.Cout)
#pragma unroll
          for (int e = 0; e < 4; ++e) sb[((int64_t)c * d.Cin + quad[k] * 4 + e) * T + t] = acc[k][c][e];
    }
  }
}

// Brick staging for the two kernels below: the 16-byte pieces of the x halo of a brick are fetched into registers
// (all loads in flight at once -- a load -> LDS store loop pays one memory round trip per piece) and stored to
// LDS one iteration later, so the fetch of the NEXT brick overlaps the sweep of the current one.
constexpr int BRK_MAXP = 14;  // pieces per thread: 6 x 12 rows x 12 columns x 4 quads / 256 threads = 13.5
// (extents are template constants: the piece -> (row, column, quad) maps are divisions by 4 and 12, not by runtime values)
template <int HH_, int HWC_, int C4N_> struct BrickGeom { static constexpr int HH = HH_, HWc = HWC_, c4n = C4N_; int RS, npieces; };
template <typename G>
__device__ __forceinline__ void brick_fetch(const rehr_direct_conv_desc& d, const G& g, int64_t br,
                                            int64_t nbricks, int nb_d, int nb_h, int nb_w, f32x4 (&tmp)[BRK_MAXP]) {
  const bool live = br < nbricks;
  const int64_t b = live ? br : 0;
  const int bw = (int)(b % nb_w);
  int64_t r = b / nb_w;
  const int bh = (int)(r % nb_h); r /= nb_h;
  const int bd = (int)(r % nb_d);
  const int n = (int)(r / nb_d);
  const int od0 = bd * HB_D, oh0 = bh * HB_H, ow0 = bw * HB_W;
  const float* xn = d.x + (int64_t)n * d.Di * d.Hi * d.Wi * d.ldx;
#pragma unroll
  for (int k = 0; k < BRK_MAXP; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int q = i % g.c4n, hv = i / g.c4n;
    const int hw_ = hv % g.HWc, row = hv / g.HWc;
    const int hh_ = row % g.HH, hd_ = row / g.HH;
    const int id = od0 - d.pd + hd_, ih = oh0 - d.ph + hh_, iw = ow0 - d.pw + hw_;
    const bool ok = live & (i < g.npieces) & ((unsigned)id < (unsigned)d.Di) & ((unsigned)ih < (unsigned)d.Hi) &
                    ((unsigned)iw < (unsigned)d.Wi);
    const int64_t off = ok ? (((int64_t)id * d.Hi + ih) * d.Wi + iw) * d.ldx + q * 4 : 0;
    const f32x4 v = *reinterpret_cast<const f32x4*>(xn + off);  // clamped address + select: no branch
    tmp[k] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
template <typename G>
__device__ __forceinline__ void brick_store(const G& g, int Cin, float* xs, const f32x4 (&tmp)[BRK_MAXP]) {
#pragma unroll
  for (int k = 0; k < BRK_MAXP; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < g.npieces) {
      const int q = i % g.c4n, hv = i / g.c4n;
      const int hw_ = hv % g.HWc, row = hv / g.HWc;
      *reinterpret_cast<f32x4*>(xs + row * g.RS + hw_ * Cin + q * 4) = tmp[k];
    }
  }
}

// ---- the same brick, register-blocked over the row taps (2 output channels, KW = 5, <= 128 (kd, kh, quad)
// triples: sr_head's 5x5x5 16->2).  A thread owns ALL KW taps of one (kd, kh, channel quad) for one of the two
// brick slices: per brick row it reads the 12 x voxels and the 8 dY pairs once and forms 5 x 8 x 4 packed
// FMAs ({x, x} * {dy0, dy1}) -- 3.3x fewer LDS reads per FMA than one (tap, quad) pair per thread, which was LDS
// bound at a quarter of the vector rate.  LDS rows are padded by 16 floats so that the lanes of a wave
// ((kd, kh) neighbours x 4 quads) hit distinct banks.
template <int KW>
__global__ __launch_bounds__(256) void small_cout2_wgrad_rows_kernel(const rehr_direct_conv_desc d,
                                                                     float* __restrict__ slab, int bricks_per_block,
                                                                     int nb_d, int nb_h, int nb_w) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float hl[];
  const int HD = HB_D + d.KD - 1, HH = HB_H + d.KH - 1;
  constexpr int HW = HB_W + KW - 1;
  const int RS = HW * d.Cin + 16;         // floats per (hd, hh) row of the x halo
  const int nrows = HD * HH;
  float* xs = hl;                         // [nrows][RS]
  float* ys = hl + nrows * RS;            // [HB_VOX][2]
  const int c4n = d.Cin / 4, ntrip = d.KD * d.KH * c4n;
  const int trip = threadIdx.x >> 1, half = threadIdx.x & 1;   // neighbouring lanes: the two brick slices
  const bool has = trip < ntrip;
  const int tq = has ? trip : 0;
  const int quad = tq % c4n, kh = (tq / c4n) % d.KH, kd = tq / (c4n * d.KH);
  f32x2_ acc[KW][4];
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[k][e] = f32x2_{0.f, 0.f};
  const int64_t nbricks = (int64_t)d.N * nb_d * nb_h * nb_w;
  const int64_t b0 = (int64_t)blockIdx.x * bricks_per_block;
  const BrickGeom<HB_H + 4, HB_W + KW - 1, 4> geo = {RS, nrows * HW * c4n};  // KH = 5, Cin = 16 (dispatch)
  f32x4 tmp[BRK_MAXP];
  brick_fetch(d, geo, b0, nbricks, nb_d, nb_h, nb_w, tmp);
  for (int bi = 0; bi < bricks_per_block; ++bi) {
    const int64_t br = b0 + bi;
    if (br >= nbricks) break;
    const int bw = (int)(br % nb_w);
    int64_t r = br / nb_w;
    const int bh = (int)(r % nb_h); r /= nb_h;
    const int bd = (int)(r % nb_d);
    const int n = (int)(r / nb_d);
    const int od0 = bd * HB_D, oh0 = bh * HB_H, ow0 = bw * HB_W;
    __syncthreads();  // previous brick fully consumed
    brick_store(geo, d.Cin, xs, tmp);
    brick_fetch(d, geo, (bi + 1 < bricks_per_block) ? br + 1 : nbricks, nbricks, nb_d, nb_h, nb_w, tmp);
    for (int i = threadIdx.x; i < HB_VOX * 2; i += 256) {
      const int c = i & 1, v = i >> 1;
      const int vw = v % HB_W, vh = (v / HB_W) % HB_H, vd = v / (HB_W * HB_H);
      const int od = od0 + vd, oh = oh0 + vh, ow = ow0 + vw;
      float val = 0.f;
      if (c < d.Cout && od < d.Do && oh < d.Ho && ow < d.Wo)
        val = d.y[((((int64_t)n * d.Do + od) * d.Ho + oh) * d.Wo + ow) * d.ldy + c];
      ys[i] = val;
    }
    __syncthreads();
    if (has) {
      const float* xb = xs + ((kd + half) * HH + kh) * RS + quad * 4;
      const float* yb = ys + half * HB_H * HB_W * 2;
#pragma unroll 2
      for (int vh = 0; vh < HB_H; ++vh) {
        f32x4 xv[HW];
#pragma unroll
        for (int j = 0; j < HW; ++j) xv[j] = *reinterpret_cast<const f32x4*>(xb + vh * RS + j * d.Cin);
        f32x4 dq[HB_W / 2];  // dY pairs of the 8 voxels of this row
#pragma unroll
        for (int j = 0; j < HB_W / 2; ++j) dq[j] = *reinterpret_cast<const f32x4*>(yb + vh * HB_W * 2 + j * 4);
#pragma unroll
        for (int v = 0; v < HB_W; ++v) {
          const f32x2_ dy2 = {dq[v >> 1][(v & 1) * 2], dq[v >> 1][(v & 1) * 2 + 1]};
#pragma unroll
          for (int k = 0; k < KW; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[k][e] = __builtin_elementwise_fma(f32x2_{xv[v + k][e], xv[v + k][e]}, dy2, acc[k][e]);
        }
      }
    }
  }
  // the two slices of a triple are neighbouring lanes: one shuffle, the even lane stores
  const int T = d.KD * d.KH * KW;
  float* sb = slab + (int64_t)blockIdx.x * d.Cout * d.Cin * T;
#pragma unroll
  for (int k = 0; k < KW; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = acc[k][e][0] + __shfl_xor(acc[k][e][0], 1, 64);
      const float a1 = acc[k][e][1] + __shfl_xor(acc[k][e][1], 1, 64);
      if (has && half == 0) {
        const int t = (kd * d.KH + kh) * KW + k;
        sb[((int64_t)0 * d.Cin + quad * 4 + e) * T + t] = a0;
        if (d.Cout > 1) sb[((int64_t)1 * d.Cin + quad * 4 + e) * T + t] = a1;
      }
    }
}

// partial[b][c] = sum over the block's voxels of dy[.][c]
__global__ __launch_bounds__(256) void thin_bias_partial_kernel(const float* __restrict__ dy, int ld, int C,
                                                                int64_t rows, float* __restrict__ partial) {
  __shared__ float red[4][4];
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256)
    for (int c = 0; c < C; ++c) a[c] += dy[r * ld + c];
  for (int c = 0; c < 4; ++c) {
    const float v = wave_sum(a[c]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < C)
    partial[(int64_t)blockIdx.x * 4 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dw[i] = sum over strips of slab[k][i], in a fixed order: a block owns 16 consecutive weights, its 16 thread
// groups walk interleaved strips (a thread per weight looping over thousands of strips serially took 1 ms)
__global__ __launch_bounds__(256) void small_cout_wgrad_reduce_kernel(const float* __restrict__ slab, int strips,
                                                                      int64_t nw, float* __restrict__ dw,
                                                                      const float* __restrict__ bpart, int bblocks,
                                                                      int C, float* __restrict__ dbias) {
  __shared__ float part[16][16];
  const int wl = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + wl;
  float s = 0.f;
  if (i < nw)
    for (int k = sg; k < strips; k += 16) s += slab[(int64_t)k * nw + i];
  part[sg][wl] = s;
  __syncthreads();
  if (sg == 0 && i < nw) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][wl];
    dw[i] = t;
  }
  if (dbias != nullptr && blockIdx.x == 0 && threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < bblocks; ++k) t += bpart[(int64_t)k * 4 + threadIdx.x];
    dbias[threadIdx.x] = t;
  }
}

bool small_cout_ok(const rehr_direct_conv_desc& d) {
  if (!d.x || !d.w || !d.y) return false;
  if (d.Cout < 1 || d.Cout > 4 || d.Cin < 16 || d.Cin % 16) return false;
  if (d.sd != 1 || d.sh != 1 || d.sw != 1) return false;
  if (d.KD < 1 || d.KH < 1 || d.KW < 1 || d.KW > SC_MAXKW) return false;
  if (d.ldx % 4 || (((uintptr_t)d.x) & 15)) return false;
  if (d.N < 1) return false;
  if (d.Di + 2 * d.pd - d.KD + 1 != d.Do || d.Hi + 2 * d.ph - d.KH + 1 != d.Ho || d.Wi + 2 * d.pw - d.KW + 1 != d.Wo)
    return false;
  return true;
}
bool sc_cg(const rehr_direct_conv_desc& d) { return d.Cin == 16 || d.Cin == 32 || d.Cin == 64; }
bool sc_halo(const rehr_direct_conv_desc& d) {
  const int np = d.KD * d.KH * d.KW * (d.Cin / 4);
  const int64_t hv = (int64_t)(HB_D + d.KD - 1) * (HB_H + d.KH - 1) * (HB_W + d.KW - 1);
  return np > 128 && np <= 512 && (hv * d.Cin + HB_VOX * 4) * 4 <= 64 * 1024;
}
int sc_halo_blocks(const rehr_direct_conv_desc& d, int* bpb, int* nbd, int* nbh, int* nbw) {
  *nbd = (d.Do + HB_D - 1) / HB_D; *nbh = (d.Ho + HB_H - 1) / HB_H; *nbw = (d.Wo + HB_W - 1) / HB_W;
  const int64_t nbricks = (int64_t)d.N * *nbd * *nbh * *nbw;
  int64_t blocks = nbricks < 2048 ? nbricks : 2048;
  int64_t per = (nbricks + blocks - 1) / blocks;
  blocks = (nbricks + per - 1) / per;
  *bpb = (int)per;
  return (int)blocks;
}
int sc_strips(const rehr_direct_conv_desc& d, int* rpb) {
  if (sc_halo(d)) { int a, b, c; return sc_halo_blocks(d, rpb, &a, &b, &c); }
  const int64_t nrows = (int64_t)d.N * d.Do * d.Ho;
  const int64_t per_strip_blocks = (int64_t)d.KD * d.KH * (sc_cg(d) ? 1 : d.Cin / 4);
  int64_t strips = (4096 + per_strip_blocks - 1) / per_strip_blocks;
  if (strips > nrows) strips = nrows;
  if (strips < 1) strips = 1;
  int64_t r = (nrows + strips - 1) / strips;
  strips = (nrows + r - 1) / r;
  *rpb = (int)r;
  return (int)strips;
}
}  // namespace

extern "C" int rehr_conv_small_cout_fwd_f32(const rehr_direct_conv_desc* dp, void* stream) {
  if (!dp || !small_cout_ok(*dp)) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  const int T = d.KD * d.KH * d.KW;
  const int CO = d.Cout <= 2 ? 2 : 4;
  const size_t smem = (size_t)T * CO * d.Cin * sizeof(float);
  if (smem > 64 * 1024) return REHR_ENOSUP;
  const int64_t groups = (int64_t)d.N * d.Do * d.Ho * ((d.Wo + SC_VOX - 1) / SC_VOX);
  int64_t blocks = (groups + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (sc_cg(d)) {
    const int cg = d.Cin / 4;
    int64_t b2 = (groups + (256 / cg) - 1) / (256 / cg);
    if (b2 > 16384) b2 = 16384;
    const dim3 g((unsigned)b2), t(256);
#define SC_FWD(CO_, CG_) hipLaunchKernelGGL((small_cout_fwd_cg_kernel<CO_, CG_>), g, t, smem, st, d)
    if (CO == 2) { if (cg == 4) SC_FWD(2, 4); else if (cg == 8) SC_FWD(2, 8); else SC_FWD(2, 16); }
    else         { if (cg == 4) SC_FWD(4, 4); else if (cg == 8) SC_FWD(4, 8); else SC_FWD(4, 16); }
#undef SC_FWD
  } else if (CO == 2)
    hipLaunchKernelGGL(small_cout_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), smem, st, d);
  else
    hipLaunchKernelGGL(small_cout_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), smem, st, d);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_conv_small_cout_dgrad_f32(const rehr_direct_conv_desc* dp, float* dx, void* stream) {
  if (!dp || !small_cout_ok(*dp) || !dx || (((uintptr_t)dx) & 15)) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  const int T = d.KD * d.KH * d.KW;
  const int CO = d.Cout <= 2 ? 2 : 4;
  const size_t smem = (size_t)T * CO * d.Cin * sizeof(float);
  if (smem > 64 * 1024) return REHR_ENOSUP;
  const int64_t groups = (int64_t)d.N * d.Di * d.Hi * ((d.Wi + SC_VOX - 1) / SC_VOX) * (d.Cin / 16);
  int64_t blocks = (groups + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (CO == 2)
    hipLaunchKernelGGL(small_cout_dgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, d, dx);
  else
    hipLaunchKernelGGL(small_cout_dgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, d, dx);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int64_t rehr_conv_small_cout_wgrad_workspace_bytes(const rehr_direct_conv_desc* dp) {
  if (!dp || !small_cout_ok(*dp)) return REHR_EINVAL;
  int rpb;
  const int strips = sc_strips(*dp, &rpb);
  return ((int64_t)strips * dp->Cout * dp->Cin * dp->KD * dp->KH * dp->KW + SC_BIAS_BLOCKS * 4) * sizeof(float);
}

extern "C" int rehr_conv_small_cout_wgrad_f32(const rehr_direct_conv_desc* dp, float* dw, float* dbias,
                                              float* workspace, int64_t workspace_bytes, void* stream) {
  if (!dp || !small_cout_ok(*dp) || !dw || !workspace) return REHR_EINVAL;
  const rehr_direct_conv_desc& d = *dp;
  int rpb;
  const int strips = sc_strips(d, &rpb);
  const int64_t nw = (int64_t)d.Cout * d.Cin * d.KD * d.KH * d.KW;
  if (workspace_bytes < ((int64_t)strips * nw + SC_BIAS_BLOCKS * 4) * (int64_t)sizeof(float)) return REHR_EINVAL;
  float* bpart = workspace + (int64_t)strips * nw;
  const int CO = d.Cout <= 2 ? 2 : 4;
  const int64_t blocks = (int64_t)strips * d.KD * d.KH * (d.Cin / 4);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, (size_t)strips * nw * sizeof(float), st) != hipSuccess) return REHR_EHIP;
  if (sc_halo(d)) {
    int bpb, nbd, nbh, nbw;
    const int hblocks = sc_halo_blocks(d, &bpb, &nbd, &nbh, &nbw);
    const int64_t hv = (int64_t)(HB_D + d.KD - 1) * (HB_H + d.KH - 1) * (HB_W + d.KW - 1);
    const size_t hsmem = (size_t)(hv * d.Cin + HB_VOX * CO) * sizeof(float);
    const int64_t rows_floats = (int64_t)(HB_D + d.KD - 1) * (HB_H + d.KH - 1) * ((HB_W + 4) * d.Cin + 16) + HB_VOX * 2;
    if (CO == 2 && d.KW == 5 && d.KH == 5 && d.KD == 5 && d.Cin == 16 && rows_floats * 4 <= 64 * 1024 &&
        (HB_D + d.KD - 1) * (HB_H + d.KH - 1) * (HB_W + 4) * (d.Cin / 4) <= BRK_MAXP * 256) {
      hipLaunchKernelGGL(small_cout2_wgrad_rows_kernel<5>, dim3(hblocks), dim3(256), (size_t)rows_floats * 4, st, d,
                         workspace, bpb, nbd, nbh, nbw);
    } else if (CO == 2)
      hipLaunchKernelGGL(small_cout_wgrad_halo_kernel<2>, dim3(hblocks), dim3(256), hsmem, st, d, workspace, bpb, nbd,
                         nbh, nbw);
    else
      hipLaunchKernelGGL(small_cout_wgrad_halo_kernel<4>, dim3(hblocks), dim3(256), hsmem, st, d, workspace, bpb, nbd,
                         nbh, nbw);
  } else if (sc_cg(d)) {
    const int cg = d.Cin / 4;
    const dim3 g((unsigned)((int64_t)strips * d.KD * d.KH)), t(256);
#define SC_WG(CO_, CG_) hipLaunchKernelGGL((small_cout_wgrad_cg_kernel<CO_, CG_>), g, t, 0, st, d, workspace, rpb)
    if (CO == 2) { if (cg == 4) SC_WG(2, 4); else if (cg == 8) SC_WG(2, 8); else SC_WG(2, 16); }
    else         { if (cg == 4) SC_WG(4, 4); else if (cg == 8) SC_WG(4, 8); else SC_WG(4, 16); }
#undef SC_WG
  } else if (CO == 2)
    hipLaunchKernelGGL(small_cout_wgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, d, workspace, rpb);
  else
    hipLaunchKernelGGL(small_cout_wgrad_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, d, workspace, rpb);
  if (dbias != nullptr)
    hipLaunchKernelGGL(thin_bias_partial_kernel, dim3(SC_BIAS_BLOCKS), dim3(256), 0, st, d.y, d.ldy, d.Cout,
                       (int64_t)d.N * d.Do * d.Ho * d.Wo, bpart);
  hipLaunchKernelGGL(small_cout_wgrad_reduce_kernel, dim3((unsigned)((nw + 15) / 16)), dim3(256), 0, st, workspace,
                     strips, nw, dw, bpart, SC_BIAS_BLOCKS, d.Cout, dbias);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
