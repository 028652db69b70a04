// UASR head of FLAVR's UNet_3D_3D with use_uncertainty (reference models/FLAVR/FLAVR_arch.py:203-246):
// the K per-voxel candidate (image, segmentation) pairs are blended with softmax weights and the same
// weights give the aleatoric uncertainty through a 1x1x1 convolution to one channel + sigmoid:
//
//   s      = softmax_i(ue[.., i])                                   i < K
//   out0   = sum_i s_i * (tanh(om[.., 2i]) + 1) / 2                 image
//   out1   = sum_i s_i * om[.., 2i+1]                               segmentation logit
//   unc    = sigmoid(bu + sum_i s_i * wu_i)
//
// The reference (and the torch composition this replaces) walks i in a Python loop: ~330 elementwise launches
// per step on the reference's own training shape, the slice gradients each through a zero fill + add.
//
// Layout.  om and ue are the NDHWC outputs of the two 1x1 convolutions on the fused slice, (N, D*2K, 1, H, W) and
// (N, D*K, 1, H, W): per voxel (n, hw) the channels are innermost, output slice d owning channels [d*2K, (d+1)*2K)
// resp. [d*K, (d+1)*K) -- the reference's split(dim=1) + stack(dim=2) is this indexing.  One thread per (n, hw, d)
// reads 2K + K consecutive floats (16-byte loads, a wave reads 64 x 192 B contiguous for K = 16) and writes three:
// out (N, 2, D, H, W) and unc (N, 1, D, H, W), plain NCDHW.
// Backward recomputes s from ue, reads the same 3K floats + 3 gradients and writes 3K; the K + 1 parameter
// gradients of the 1x1x1 convolution are block partials in double (fixed order), summed by the host.
// HBM-bound: 12K + 12 B per voxel-slice forward, 24K + 24 B backward.
#include "common.h"

namespace {

constexpr int UH_THREADS = 256;
constexpr int UH_MAX_BLOCKS = 2048;

template <int K>
struct UHVox {
  float a[K], b[K], s[K];  // image logit, segmentation logit, softmax weight
};

template <int K>
__device__ __forceinline__ void uh_load(const float* __restrict__ om, const float* __restrict__ ue, int64_t t,
                                        UHVox<K>& v) {
  const f32x4* po = reinterpret_cast<const f32x4*>(om + t * (2 * K));
  const f32x4* pu = reinterpret_cast<const f32x4*>(ue + t * K);
#pragma unroll
  for (int q = 0; q < K / 2; ++q) {
    const f32x4 x = po[q];
    v.a[2 * q] = x[0], v.b[2 * q] = x[1], v.a[2 * q + 1] = x[2], v.b[2 * q + 1] = x[3];
  }
#pragma unroll
  for (int q = 0; q < K / 4; ++q) {
    const f32x4 x = pu[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) v.s[4 * q + j] = x[j];
  }
  float m = v.s[0];
#pragma unroll
  for (int i = 1; i < K; ++i) m = fmaxf(m, v.s[i]);
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    v.s[i] = expf(v.s[i] - m);
    tot += v.s[i];
  }
  const float inv = 1.f / tot;
#pragma unroll
  for (int i = 0; i < K; ++i) v.s[i] *= inv;
}

template <int K>
__global__ __launch_bounds__(UH_THREADS) void uasr_mix_fwd_kernel(const float* __restrict__ om,
                                                                  const float* __restrict__ ue,
                                                                  const float* __restrict__ wu,
                                                                  const float* __restrict__ bu,
                                                                  float* __restrict__ out, float* __restrict__ unc,
                                                                  int64_t total, int D, int64_t HW) {
  float w[K];
#pragma unroll
  for (int i = 0; i < K; ++i) w[i] = wu[i];
  const float b0 = bu[0];
  for (int64_t t = (int64_t)blockIdx.x * UH_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * UH_THREADS) {
    UHVox<K> v;
    uh_load<K>(om, ue, t, v);
    float o0 = 0.f, o1 = 0.f, z = b0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      o0 += v.s[i] * (0.5f * (tanhf(v.a[i]) + 1.f));
      o1 += v.s[i] * v.b[i];
      z += v.s[i] * w[i];
    }
    const int d = (int)(t % D);
    const int64_t r = t / D, hw = r % HW, n = r / HW;
    const int64_t o = (n * 2 * D + d) * HW + hw;
    out[o] = o0;
    out[o + (int64_t)D * HW] = o1;
    unc[(n * D + d) * HW + hw] = 1.f / (1.f + expf(-z));
  }
}

template <int K>
__global__ __launch_bounds__(UH_THREADS) void uasr_mix_bwd_kernel(const float* __restrict__ om,
                                                                  const float* __restrict__ ue,
                                                                  const float* __restrict__ wu,
                                                                  const float* __restrict__ bu,
                                                                  const float* __restrict__ gout,
                                                                  const float* __restrict__ gunc,
                                                                  float* __restrict__ dom, float* __restrict__ due,
                                                                  double* __restrict__ partial, int64_t total, int D,
                                                                  int64_t HW) {
  float w[K], aw[K];
#pragma unroll
  for (int i = 0; i < K; ++i) w[i] = wu[i], aw[i] = 0.f;
  const float b0 = bu[0];
  float ab = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * UH_THREADS + threadIdx.x; t < total; t += (int64_t)gridDim.x * UH_THREADS) {
    UHVox<K> v;
    uh_load<K>(om, ue, t, v);
    const int d = (int)(t % D);
    const int64_t r = t / D, hw = r % HW, n = r / HW;
    const int64_t o = (n * 2 * D + d) * HW + hw;
    const float g0 = gout[o], g1 = gout[o + (int64_t)D * HW];
    float img[K], o0 = 0.f, o1 = 0.f, zs = 0.f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float th = tanhf(v.a[i]);
      img[i] = 0.5f * (th + 1.f);
      o0 += v.s[i] * img[i];
      o1 += v.s[i] * v.b[i];
      zs += v.s[i] * w[i];
      v.a[i] = 0.5f * (1.f - th * th);  // d img / d logit
    }
    const float u = 1.f / (1.f + expf(-(zs + b0)));
    const float gz = gunc[(n * D + d) * HW + hw] * u * (1.f - u);
    const float dot = g0 * o0 + g1 * o1 + gz * zs;  // sum_j s_j * dL/ds_j
    f32x4* pd = reinterpret_cast<f32x4*>(dom + t * (2 * K));
    f32x4* pe = reinterpret_cast<f32x4*>(due + t * K);
    float de[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float ds = g0 * img[i] + g1 * v.b[i] + gz * w[i];
      de[i] = v.s[i] * (ds - dot);
      aw[i] += gz * v.s[i];
    }
    ab += gz;
#pragma unroll
    for (int q = 0; q < K / 2; ++q) {
      f32x4 x;
      x[0] = g0 * v.s[2 * q] * v.a[2 * q], x[1] = g1 * v.s[2 * q];
      x[2] = g0 * v.s[2 * q + 1] * v.a[2 * q + 1], x[3] = g1 * v.s[2 * q + 1];
      pd[q] = x;
    }
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
      f32x4 x;
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = de[4 * q + j];
      pe[q] = x;
    }
  }
  // block partial of (dwu[0..K), dbu): waves in fixed order
  __shared__ double red[UH_THREADS / 64][K + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const double s = wave_sum_d((double)aw[i]);
    if (lane == 0) red[wave][i] = s;
  }
  {
    const double s = wave_sum_d((double)ab);
    if (lane == 0) red[wave][K] = s;
  }
  __syncthreads();
  if (threadIdx.x <= K) {
    double s = 0.0;
#pragma unroll
    for (int wv = 0; wv < UH_THREADS / 64; ++wv) s += red[wv][threadIdx.x];
    partial[(int64_t)blockIdx.x * (K + 1) + threadIdx.x] = s;
  }
}

inline int uh_blocks(int64_t total) {
  int64_t b = (total + UH_THREADS - 1) / UH_THREADS;
  if (b > UH_MAX_BLOCKS) b = UH_MAX_BLOCKS;
  return (int)(b < 1 ? 1 : b);
}

inline bool uh_ok(const void* a, const void* b, int32_t N, int32_t K, int32_t D, int64_t HW) {
  return a && b && N > 0 && D > 0 && HW > 0 && (K == 4 || K == 8 || K == 16 || K == 32) &&
         (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

}  // namespace

extern "C" int32_t rehr_uasr_mix_blocks(int32_t N, int32_t D, int64_t HW) {
  if (N <= 0 || D <= 0 || HW <= 0) return 0;
  return uh_blocks((int64_t)N * D * HW);
}

extern "C" int rehr_uasr_mix_fwd_f32(const float* om, const float* ue, const float* wu, const float* bu, float* out,
                                     float* unc, int32_t N, int32_t K, int32_t D, int64_t HW, void* stream) {
  if (!uh_ok(om, ue, N, K, D, HW) || !wu || !bu || !out || !unc) return REHR_EINVAL;
  const int64_t total = (int64_t)N * D * HW;
  hipStream_t ST = (hipStream_t)stream;
  const dim3 grid(uh_blocks(total)), block(UH_THREADS);
  switch (K) {
    case 4: hipLaunchKernelGGL(uasr_mix_fwd_kernel<4>, grid, block, 0, ST, om, ue, wu, bu, out, unc, total, D, HW); break;
    case 8: hipLaunchKernelGGL(uasr_mix_fwd_kernel<8>, grid, block, 0, ST, om, ue, wu, bu, out, unc, total, D, HW); break;
    case 16: hipLaunchKernelGGL(uasr_mix_fwd_kernel<16>, grid, block, 0, ST, om, ue, wu, bu, out, unc, total, D, HW); break;
    default: hipLaunchKernelGGL(uasr_mix_fwd_kernel<32>, grid, block, 0, ST, om, ue, wu, bu, out, unc, total, D, HW); break;
  }
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_uasr_mix_bwd_f32(const float* om, const float* ue, const float* wu, const float* bu,
                                     const float* gout, const float* gunc, float* dom, float* due, double* partial,
                                     int32_t N, int32_t K, int32_t D, int64_t HW, void* stream) {
  if (!uh_ok(om, ue, N, K, D, HW) || !uh_ok(dom, due, N, K, D, HW) || !wu || !bu || !gout || !gunc || !partial)
    return REHR_EINVAL;
  const int64_t total = (int64_t)N * D * HW;
  hipStream_t ST = (hipStream_t)stream;
  const dim3 grid(uh_blocks(total)), block(UH_THREADS);
#define UH_BWD(KK)                                                                                                  \
  hipLaunchKernelGGL(uasr_mix_bwd_kernel<KK>, grid, block, 0, ST, om, ue, wu, bu, gout, gunc, dom, due, partial, total, \
                     D, HW)
  switch (K) {
    case 4: UH_BWD(4); break;
    case 8: UH_BWD(8); break;
    case 16: UH_BWD(16); break;
    default: UH_BWD(32); break;
  }
#undef UH_BWD
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
