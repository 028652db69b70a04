// Fused segmentation loss: softmax + cross-entropy (optionally uncertainty-weighted) + soft Dice in
// ONE pass over the logits, and one pass for the gradient.
//
// Reference: utils/seg_utils.py:289-351 (RobustCrossEntropyLoss, DC_and_weighted_CE_loss) on top of
// nnunetv2 MemoryEfficientSoftDiceLoss (restated in rehrseg_amd/utils/seg_utils.py::SoftDiceLoss):
//
//   ce[b,v]   = -log softmax(logits[b,:,v])[target[b,v]]
//   CE term   = mean over (a, b, v) of ce[b,v] * u[a,v]      (the (B,D,H,W)*(B,1,D,H,W) broadcast of the
//               reference; without uncertainty: mean over (b, v) of ce)
//   Dice term = -mean over (b, c in classes) of (2 I + s) / clip(G + P + s, 1e-8),
//               I = sum_v p*y, P = sum_v p, G = sum_v y   per (sample, class)
//
// The HR logits (2 x 2 x 512 x 128 x 128) are the largest tensors of a stage-2 step; the torch
// composition reads them ~10 times.  Here: forward 8+4(+4) B/voxel-sample, backward the same + 8 B write.
// Statistics are accumulated per block and added with double atomics: stats[N][C][3] = {I, P, G}, then
// stats[N*C*3] = sum_v (sum_b ce) * (sum_a u).
#include "common.h"

namespace {

constexpr int SL_MAXN = 4;

template <int C>
__device__ __forceinline__ void softmax_ce(const float* __restrict__ lp, int t, float (&p)[C], float& ce) {
  float l[C];
#pragma unroll
  for (int c = 0; c < C; ++c) l[c] = lp[c];
  float m = l[0];
#pragma unroll
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float s = 0.f, lt = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    p[c] = __expf(l[c] - m);
    s += p[c];
    lt = (c == t) ? l[c] : lt;
  }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
  ce = __logf(s) - (lt - m);
}

template <int C>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ logits, int ld,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ unc, int N, int64_t S,
                                                           double* __restrict__ stats) {
  float aI[SL_MAXN][C], aP[SL_MAXN][C], aG[SL_MAXN][C];
#pragma unroll
  for (int b = 0; b < SL_MAXN; ++b)
#pragma unroll
    for (int c = 0; c < C; ++c) aI[b][c] = aP[b][c] = aG[b][c] = 0.f;
  float ace = 0.f;
  // samples are taken SL_MAXN at a time (grid.y): the cross-sample term sum_v (sum_b ce_b)(sum_a u_a) splits over
  // chunks of b, every chunk pairing its ce with the uncertainty sum over ALL samples
  const int b0 = blockIdx.y * SL_MAXN;
  const int nb = (N - b0 < SL_MAXN) ? N - b0 : SL_MAXN;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < S; v += (int64_t)gridDim.x * blockDim.x) {
    float sce = 0.f, su = 0.f;
    if (unc != nullptr)
      for (int a = 0; a < N; ++a) su += unc[(int64_t)a * S + v];
#pragma unroll
    for (int b = 0; b < SL_MAXN; ++b) {
      if (b < nb) {
        const int64_t i = (int64_t)(b0 + b) * S + v;
        const int t = (int)target[i];
        float p[C], ce;
        softmax_ce<C>(logits + i * ld, t, p, ce);
        sce += ce;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float y = (c == t) ? 1.f : 0.f;
          aI[b][c] += p[c] * y;
          aP[b][c] += p[c];
          aG[b][c] += y;
        }
      }
    }
    ace += (unc != nullptr) ? sce * su : sce;
  }
  // block reduction, then one double atomic per statistic and block
  __shared__ float red[4][SL_MAXN * C * 3 + 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int b = 0; b < SL_MAXN; ++b)
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float i_ = wave_sum(aI[b][c]), p_ = wave_sum(aP[b][c]), g_ = wave_sum(aG[b][c]);
      if (lane == 0) {
        red[w][(b * C + c) * 3 + 0] = i_;
        red[w][(b * C + c) * 3 + 1] = p_;
        red[w][(b * C + c) * 3 + 2] = g_;
      }
    }
  const float ce_ = wave_sum(ace);
  if (lane == 0) red[w][SL_MAXN * C * 3] = ce_;
  __syncthreads();
  const int nstat = nb * C * 3;
  for (int k = threadIdx.x; k <= nstat; k += blockDim.x) {
    const int src = (k == nstat) ? SL_MAXN * C * 3 : k;
    const float s = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    atomicAdd(stats + ((k == nstat) ? N * C * 3 : b0 * C * 3 + k), (double)s);
  }
}

template <int C>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits, int ld,
                                                           const float* __restrict__ target,
                                                           const float* __restrict__ unc, int N, int64_t S,
                                                           const double* __restrict__ stats, float w_ce, float w_dice,
                                                           float smooth, int do_bg,
                                                           const float* __restrict__ grad_out,
                                                           float* __restrict__ dlogits, int ldd) {
  // per (sample, class) Dice constants: d(-dc)/dp = -(2 y den - num) / den^2 = y*A + B
  __shared__ float cA[SL_MAXN * C], cB[SL_MAXN * C];
  const int b0 = blockIdx.y * SL_MAXN;
  const int nb = (N - b0 < SL_MAXN) ? N - b0 : SL_MAXN;
  if (threadIdx.x < nb * C) {
    const int c = threadIdx.x % C;
    const double* sp = stats + (size_t)(b0 * C + threadIdx.x) * 3;
    const double I = sp[0], P = sp[1], G = sp[2];
    const double num = 2.0 * I + smooth;
    const double raw = G + P + smooth;
    const double den = raw < 1e-8 ? 1e-8 : raw;
    const int ncls = do_bg ? C : C - 1;
    const double k = (double)w_dice / ((double)N * ncls);
    const bool on = do_bg || c > 0;
    // clipped denominator: its derivative vanishes, only the numerator's 2*y term remains
    cA[threadIdx.x] = on ? (float)(-k * 2.0 / den) : 0.f;
    cB[threadIdx.x] = (on && raw >= 1e-8) ? (float)(k * num / (den * den)) : 0.f;
  }
  __syncthreads();
  const float go = grad_out[0];
  const float kce = w_ce / ((unc != nullptr) ? (float)((double)N * N * S) : (float)((double)N * S));
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < S; v += (int64_t)gridDim.x * blockDim.x) {
    float su = 1.f;
    if (unc != nullptr) {
      su = 0.f;
      for (int a = 0; a < N; ++a) su += unc[(int64_t)a * S + v];
    }
    const float cw = kce * su;
#pragma unroll
    for (int b = 0; b < SL_MAXN; ++b) {
      if (b < nb) {
        const int64_t i = (int64_t)(b0 + b) * S + v;
        const int t = (int)target[i];
        float p[C], ce;
        softmax_ce<C>(logits + i * ld, t, p, ce);
        float g[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float y = (c == t) ? 1.f : 0.f;
          g[c] = y * cA[b * C + c] + cB[b * C + c];
          dot += g[c] * p[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float y = (c == t) ? 1.f : 0.f;
          dlogits[i * ldd + c] = go * (cw * (p[c] - y) + p[c] * (g[c] - dot));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// BCE-with-logits + Dice of the SR stage's segmentation channel (utils/seg_utils.py:786-885: BCEDiceLoss =
// alpha * BCEWithLogitsLoss + beta * (1 - mean_c 2 sum(p t) / clamp(sum p^2 + sum t^2, 1e-6)), sums per channel over
// batch and space, p = sigmoid(x)) in one pass each way.  x, t dense [N][C][S]; stats[c][4] = {sum bce, sum p t,
// sum p^2, sum t^2} (double, accumulated with one atomic per statistic and block).
__global__ __launch_bounds__(256) void bce_dice_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                          int C, int64_t S, double* __restrict__ stats) {
  const int c = blockIdx.y, n = blockIdx.z;
  const float* xp = x + ((int64_t)n * C + c) * S;
  const float* tp = t + ((int64_t)n * C + c) * S;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = xp[i], y = tp[i];
    const float e = __expf(-fabsf(v));
    const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    a0 += fmaxf(v, 0.f) - v * y + __logf(1.f + e);
    a1 += p * y;
    a2 += p * p;
    a3 += y * y;
  }
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  if (lane == 0) { red[w][0] = a0; red[w][1] = a1; red[w][2] = a2; red[w][3] = a3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    atomicAdd(stats + c * 4 + k, (double)(red[0][k] + red[1][k] + red[2][k] + red[3][k]));
  }
}

__global__ __launch_bounds__(256) void bce_dice_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                          int C, int64_t S, const double* __restrict__ stats, float kbce,
                                                          float kdice, const float* __restrict__ grad_out,
                                                          float* __restrict__ dx) {
  const int c = blockIdx.y, n = blockIdx.z;
  const int64_t base = ((int64_t)n * C + c) * S;
  // d(-dice_c)/dp = -(2 y den - inter * 2 p * 2) / den^2  (den unclamped), times kdice = beta / C
  const double inter = stats[c * 4 + 1], raw = stats[c * 4 + 2] + stats[c * 4 + 3];
  const double den = raw < 1e-6 ? 1e-6 : raw;
  const float A = (float)(-2.0 / den) * kdice;                               // coefficient of y
  const float B = raw >= 1e-6 ? (float)(4.0 * inter / (den * den)) * kdice : 0.f;   // coefficient of p
  const float go = grad_out[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[base + i], y = t[base + i];
    const float e = __expf(-fabsf(v));
    const float p = v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
    dx[base + i] = go * (kbce * (p - y) + (A * y + B * p) * p * (1.f - p));
  }
}

int grid_for(int64_t S) {
  int64_t b = (S + 255) / 256;
  return (int)(b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" int rehr_seg_loss_fwd_f32(const float* logits, int32_t ld, const float* target, const float* unc,
                                     int32_t N, int32_t C, int64_t S, double* stats, void* stream) {
  if (!logits || !target || !stats || N < 1 || S < 1 || ld < C) return REHR_EINVAL;
  if (C < 2 || C > 4 || N > 65535 * SL_MAXN) return REHR_ENOSUP;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, sizeof(double) * ((size_t)N * C * 3 + 1), st) != hipSuccess) return REHR_EHIP;
  const dim3 g(grid_for(S), (N + SL_MAXN - 1) / SL_MAXN), t(256);
  if (C == 2) hipLaunchKernelGGL(seg_loss_fwd_kernel<2>, g, t, 0, st, logits, ld, target, unc, N, S, stats);
  else if (C == 3) hipLaunchKernelGGL(seg_loss_fwd_kernel<3>, g, t, 0, st, logits, ld, target, unc, N, S, stats);
  else hipLaunchKernelGGL(seg_loss_fwd_kernel<4>, g, t, 0, st, logits, ld, target, unc, N, S, stats);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_seg_loss_bwd_f32(const float* logits, int32_t ld, const float* target, const float* unc,
                                     int32_t N, int32_t C, int64_t S, const double* stats, float w_ce,
                                     float w_dice, float smooth, int32_t do_bg, const float* grad_out,
                                     float* dlogits, int32_t ldd, void* stream) {
  if (!logits || !target || !stats || !grad_out || !dlogits || N < 1 || S < 1 || ld < C || ldd < C)
    return REHR_EINVAL;
  if (C < 2 || C > 4 || N > 65535 * SL_MAXN) return REHR_ENOSUP;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid_for(S), (N + SL_MAXN - 1) / SL_MAXN), t(256);
#define SL_BWD(C_)                                                                                            \
  hipLaunchKernelGGL(seg_loss_bwd_kernel<C_>, g, t, 0, st, logits, ld, target, unc, N, S, stats, w_ce, w_dice, \
                     smooth, do_bg, grad_out, dlogits, ldd)
  if (C == 2) SL_BWD(2);
  else if (C == 3) SL_BWD(3);
  else SL_BWD(4);
#undef SL_BWD
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_bce_dice_fwd_f32(const float* x, const float* t, int32_t N, int32_t C, int64_t S, double* stats,
                                     void* stream) {
  if (!x || !t || !stats || N < 1 || N > 65535 || C < 1 || C > 65535 || S < 1) return REHR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(stats, 0, sizeof(double) * (size_t)C * 4, st) != hipSuccess) return REHR_EHIP;
  int bx = grid_for(S);
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(bce_dice_fwd_kernel, dim3(bx, C, N), dim3(256), 0, st, x, t, C, S, stats);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_bce_dice_bwd_f32(const float* x, const float* t, int32_t N, int32_t C, int64_t S, const double* stats,
                                     float alpha, float beta, const float* grad_out, float* dx, void* stream) {
  if (!x || !t || !stats || !grad_out || !dx || N < 1 || N > 65535 || C < 1 || C > 65535 || S < 1) return REHR_EINVAL;
  int bx = grid_for(S);
  if (bx > 256) bx = 256;
  const float kbce = alpha / (float)((double)N * C * S), kdice = beta / (float)C;
  hipLaunchKernelGGL(bce_dice_bwd_kernel, dim3(bx, C, N), dim3(256), 0, (hipStream_t)stream, x, t, C, S, stats, kbce, kdice,
                     grad_out, dx);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
