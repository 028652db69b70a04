// Streaming per-(sample, channel) affine kernels of the hot path -- the HBM-bound half of the fused blocks:
//
//   InstanceNorm3d(affine) + LeakyReLU apply / gradient   nnU-Net ConvDropoutNormReLU (dynamic_network_architectures
//                                                         ==0.3.1, built at train_all.py:474-493)
//   SEGating scale (+ residual) + ReLU / LeakyReLU, gradient   models/FLAVR/resnet_3D.py:112-116,144-149,
//                                                              models/FLAVR/FLAVR_arch.py:188-200
//
// Layout: NDHWC fp32, a row = one voxel's C channels.  A block owns a contiguous run of rows of ONE sample
// (grid.y = sample), a thread owns ONE channel quad for the whole run: the per-(sample, channel) constants
// (mean, rstd, gamma, beta, gate, ...) are folded into 2-3 registers per channel before the loop, the loop
// itself is 16-byte loads / stores with no index arithmetic beyond a pointer bump -- UNROLL independent rows in
// flight per thread -- and 2-8 VALU operations per element.  (The first version of these kernels spent its
// time in 64-bit div/mod per element and ran at ~3 TB/s)
// Column sums are kept in fp32 over a short run of rows and folded into fp64 per thread, reduced over the block
// in LDS, and finished with one double atomic per (quantity, channel) and block: they cancel heavily.
#include "common.h"

namespace {

constexpr int SW_THREADS = 256;
constexpr int SW_UNROLL = 4;

struct Span {
  int64_t r0, r1;   // rows [r0, r1) of the whole tensor (sample offset included)
  int c;            // first channel of this thread's quad
  int rl, rpp;      // row lane, rows per pass
  bool active;
};

__device__ __forceinline__ Span make_span(int64_t S, int C, int64_t rows_per_block) {
  Span s;
  const int c4n = C >> 2;
  s.rpp = SW_THREADS / c4n;
  const int tid = threadIdx.x;
  s.active = tid < s.rpp * c4n;
  s.c = (tid % c4n) * 4;
  s.rl = tid / c4n;
  const int64_t b0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t b1 = b0 + rows_per_block;
  if (b1 > S) b1 = S;
  const int64_t base = (int64_t)blockIdx.y * S;
  s.r0 = base + b0;
  s.r1 = base + b1;
  return s;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// act(v) for act in {none, relu, lrelu} with one multiply + select (slope_eff = 0 for relu, 1 for none)
__device__ __forceinline__ float act_sel(float v, float slope_eff) { return v > 0.f ? v : v * slope_eff; }
inline float slope_eff_of(int act, float slope) {
  return act == REHR_ACT_RELU ? 0.f : (act == REHR_ACT_LRELU ? slope : 1.f);
}

// block-level reduction of NQ x 4 per-thread double partials -> one double atomic per (quantity, channel)
template <int NQ>
__device__ __forceinline__ void block_column_atomics(const Span& s, int C, double (&acc)[NQ][4], double* out,
                                                     int out_stride) {
  __shared__ double red[SW_THREADS * NQ * 4];
  const int c4n = C >> 2, tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(q * 4 + e) * SW_THREADS + tid] = s.active ? acc[q][e] : 0.0;
  __syncthreads();
  for (int o = tid; o < NQ * 4 * c4n; o += SW_THREADS) {
    const int cq = o % c4n, qe = o / c4n;
    double t = 0.0;
    for (int k = 0; k < s.rpp; ++k) t += red[qe * SW_THREADS + k * c4n + cq];
    atomicAdd(out + (int64_t)(cq * 4 + (qe & 3)) * out_stride + (qe >> 2), t);
  }
}

// ---------------------------------------------------------------- InstanceNorm + activation, forward
// y = act((x - mean) * rstd * gamma + beta) = act(x * sc + sh); mean / rstd from the conv epilogue's {sum, sum^2}.
__global__ __launch_bounds__(SW_THREADS) void instnorm_act_fwd_kernel(
    const float* __restrict__ x, int ldx, const double* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, int ldy, float* __restrict__ mr, int64_t S, int C,
    int64_t rows_per_block, double invS, float eps, float se) {
  const Span s = make_span(S, C, rows_per_block);
  if (!s.active) return;
  const int n = blockIdx.y;
  f32x4 sc, sh;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double* st = stats + ((int64_t)n * C + s.c + e) * 2;
    const double m = st[0] * invS;
    double var = st[1] * invS - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && s.rl == 0) {   // saved for the backward pass
      mr[((int64_t)n * C + s.c + e) * 2] = mean;
      mr[((int64_t)n * C + s.c + e) * 2 + 1] = rstd;
    }
    sc[e] = rstd * gamma[s.c + e];
    sh[e] = beta[s.c + e] - mean * sc[e];
  }
  const int64_t step = s.rpp;
  int64_t r = s.r0 + s.rl;
  for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
    f32x4 v[SW_UNROLL];
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) v[u] = ld4(x + (r + u * step) * ldx + s.c);
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = act_sel(fmaf(v[u][e], sc[e], sh[e]), se);
      st4(y + (r + u * step) * ldy + s.c, v[u]);
    }
  }
  for (; r < s.r1; r += step) {
    f32x4 v = ld4(x + r * ldx + s.c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_sel(fmaf(v[e], sc[e], sh[e]), se);
    st4(y + r * ldy + s.c, v);
  }
}

// ---------------------------------------------------------------- InstanceNorm + activation, backward
// pass 1: red[n][c] = { sum dz, sum dz * xhat },  dz = dy * act'(xhat * gamma + beta)
__global__ __launch_bounds__(SW_THREADS) void instnorm_bwd_reduce_kernel(
    const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, const float* __restrict__ mr,
    const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ red, int64_t S, int C,
    int64_t rows_per_block, float ga) {   // ga = act'(negative side): slope, 0 (relu) or 1 (none)
  const Span s = make_span(S, C, rows_per_block);
  const int n = blockIdx.y;
  double acc[2][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[0][e] = acc[1][e] = 0.0;
  if (s.active) {
    f32x4 rs, ms, g, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* m = mr + ((int64_t)n * C + s.c + e) * 2;
      rs[e] = m[1];
      ms[e] = m[0] * m[1];
      g[e] = gamma[s.c + e];
      b[e] = beta[s.c + e];
    }
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    auto one = [&](const f32x4& dyv, const f32x4& xv, f32x4& a0, f32x4& a1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = fmaf(xv[e], rs[e], -ms[e]);
        const float dz = fmaf(xh, g[e], b[e]) > 0.f ? dyv[e] : dyv[e] * ga;
        a0[e] += dz;
        a1[e] = fmaf(dz, xh, a1[e]);
      }
    };
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
      f32x4 dv[SW_UNROLL], xv[SW_UNROLL];
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        dv[u] = ld4(dy + (r + u * step) * lddy + s.c);
        xv[u] = ld4(x + (r + u * step) * ldx + s.c);
      }
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};   // fp32 over 4 rows, then into fp64
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) one(dv[u], xv[u], a0, a1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][e] += (double)a0[e];
        acc[1][e] += (double)a1[e];
      }
    }
    for (; r < s.r1; r += step) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      one(ld4(dy + r * lddy + s.c), ld4(x + r * ldx + s.c), a0, a1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][e] += (double)a0[e];
        acc[1][e] += (double)a1[e];
      }
    }
  }
  block_column_atomics<2>(s, C, acc, red + (int64_t)n * C * 2, 2);
}

// dgamma[c] = sum_n red[n][c][1]; dbeta[c] = sum_n red[n][c][0]
__global__ void instnorm_bwd_params_kernel(const double* __restrict__ red, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int N, int C) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
    double a = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) {
      b += red[((int64_t)n * C + c) * 2];
      a += red[((int64_t)n * C + c) * 2 + 1];
    }
    dgamma[c] = (float)a;
    dbeta[c] = (float)b;
  }
}

// pass 2: dx = rstd * gamma * (dz - m1 - xhat * m2),  m1 = sum dz / S, m2 = sum dz*xhat / S
__global__ __launch_bounds__(SW_THREADS) void instnorm_bwd_apply_kernel(
    const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, const float* __restrict__ mr,
    const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ red,
    float* __restrict__ dx, int lddx, int64_t S, int C, int64_t rows_per_block, double invS, float ga) {
  const Span s = make_span(S, C, rows_per_block);
  if (!s.active) return;
  const int n = blockIdx.y;
  f32x4 rs, ms, g, b, k0, m1, m2;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* m = mr + ((int64_t)n * C + s.c + e) * 2;
    const double* rd = red + ((int64_t)n * C + s.c + e) * 2;
    rs[e] = m[1];
    ms[e] = m[0] * m[1];
    g[e] = gamma[s.c + e];
    b[e] = beta[s.c + e];
    k0[e] = m[1] * g[e];
    m1[e] = (float)(rd[0] * invS);
    m2[e] = (float)(rd[1] * invS);
  }
  auto one = [&](const f32x4& dyv, const f32x4& xv) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = fmaf(xv[e], rs[e], -ms[e]);
      const float dz = fmaf(xh, g[e], b[e]) > 0.f ? dyv[e] : dyv[e] * ga;
      o[e] = k0[e] * (dz - m1[e] - xh * m2[e]);
    }
    return o;
  };
  const int64_t step = s.rpp;
  int64_t r = s.r0 + s.rl;
  for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
    f32x4 dv[SW_UNROLL], xv[SW_UNROLL];
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) {
      dv[u] = ld4(dy + (r + u * step) * lddy + s.c);
      xv[u] = ld4(x + (r + u * step) * ldx + s.c);
    }
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) st4(dx + (r + u * step) * lddx + s.c, one(dv[u], xv[u]));
  }
  for (; r < s.r1; r += step) st4(dx + r * lddx + s.c, one(ld4(dy + r * lddy + s.c), ld4(x + r * ldx + s.c)));
}

// ---------------------------------------------------------------- y = act(x * gate + res)
__global__ __launch_bounds__(SW_THREADS) void scale_res_act_fwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ gate, const float* __restrict__ res, int ldr,
    float* __restrict__ y, int ldy, int64_t S, int C, int64_t rows_per_block, float se) {
  const Span s = make_span(S, C, rows_per_block);
  if (!s.active) return;
  const f32x4 gv = ld4(gate + (int64_t)blockIdx.y * C + s.c);
  const int64_t step = s.rpp;
  int64_t r = s.r0 + s.rl;
  if (res != nullptr) {
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
      f32x4 v[SW_UNROLL], q[SW_UNROLL];
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        v[u] = ld4(x + (r + u * step) * ldx + s.c);
        q[u] = ld4(res + (r + u * step) * ldr + s.c);
      }
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = act_sel(fmaf(v[u][e], gv[e], q[u][e]), se);
        st4(y + (r + u * step) * ldy + s.c, v[u]);
      }
    }
    for (; r < s.r1; r += step) {
      f32x4 v = ld4(x + r * ldx + s.c);
      const f32x4 q = ld4(res + r * ldr + s.c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_sel(fmaf(v[e], gv[e], q[e]), se);
      st4(y + r * ldy + s.c, v);
    }
  } else {
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
      f32x4 v[SW_UNROLL];
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) v[u] = ld4(x + (r + u * step) * ldx + s.c);
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = act_sel(v[u][e] * gv[e], se);
        st4(y + (r + u * step) * ldy + s.c, v[u]);
      }
    }
    for (; r < s.r1; r += step) {
      f32x4 v = ld4(x + r * ldx + s.c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_sel(v[e] * gv[e], se);
      st4(y + r * ldy + s.c, v);
    }
  }
}

// dz = dy * act'(y); dres = dz; dx = dz * gate; dgate_acc[n][c] += sum dz * x
__global__ __launch_bounds__(SW_THREADS) void scale_res_act_bwd_kernel(
    const float* __restrict__ dy, int lddy, const float* __restrict__ y, int ldy, const float* __restrict__ x,
    int ldx, const float* __restrict__ gate, float* __restrict__ dx, int lddx, float* __restrict__ dres, int lddr,
    double* __restrict__ dgate_acc, int64_t S, int C, int64_t rows_per_block, float ga) {
  const Span s = make_span(S, C, rows_per_block);
  const int n = blockIdx.y;
  double acc[1][4] = {{0.0, 0.0, 0.0, 0.0}};
  if (s.active) {
    const f32x4 gv = ld4(gate + (int64_t)n * C + s.c);
    const int64_t step = s.rpp;
    int64_t r = s.r0 + s.rl;
    for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
      f32x4 dv[SW_UNROLL], yv[SW_UNROLL], xv[SW_UNROLL];
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        dv[u] = ld4(dy + (r + u * step) * lddy + s.c);
        yv[u] = ld4(y + (r + u * step) * ldy + s.c);
        xv[u] = ld4(x + (r + u * step) * ldx + s.c);
      }
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < SW_UNROLL; ++u) {
        f32x4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dz[e] = yv[u][e] > 0.f ? dv[u][e] : dv[u][e] * ga;
          a[e] = fmaf(dz[e], xv[u][e], a[e]);
        }
        if (dres != nullptr) st4(dres + (r + u * step) * lddr + s.c, dz);
        st4(dx + (r + u * step) * lddx + s.c, dz * gv);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[0][e] += (double)a[e];
    }
    for (; r < s.r1; r += step) {
      const f32x4 dv = ld4(dy + r * lddy + s.c), yv = ld4(y + r * ldy + s.c), xv = ld4(x + r * ldx + s.c);
      f32x4 dz;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dz[e] = yv[e] > 0.f ? dv[e] : dv[e] * ga;
        acc[0][e] += (double)(dz[e] * xv[e]);
      }
      if (dres != nullptr) st4(dres + r * lddr + s.c, dz);
      st4(dx + r * lddx + s.c, dz * gv);
    }
  }
  block_column_atomics<1>(s, C, acc, dgate_acc + (int64_t)n * C, 1);
}

// x += k[n][c] in place (the mean-pool branch of the SEGating gradient, known only after the block sums)
__global__ __launch_bounds__(SW_THREADS) void add_channel_const_kernel(float* __restrict__ x, int ldx,
                                                                      const float* __restrict__ k, int64_t S, int C,
                                                                      int64_t rows_per_block) {
  const Span s = make_span(S, C, rows_per_block);
  if (!s.active) return;
  const f32x4 kv = ld4(k + (int64_t)blockIdx.y * C + s.c);
  const int64_t step = s.rpp;
  int64_t r = s.r0 + s.rl;
  for (; r + (SW_UNROLL - 1) * step < s.r1; r += SW_UNROLL * step) {
    f32x4 v[SW_UNROLL];
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) v[u] = ld4(x + (r + u * step) * ldx + s.c);
#pragma unroll
    for (int u = 0; u < SW_UNROLL; ++u) st4(x + (r + u * step) * ldx + s.c, v[u] + kv);
  }
  for (; r < s.r1; r += step) st4(x + r * ldx + s.c, ld4(x + r * ldx + s.c) + kv);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// rows per block: ~2048 blocks over the whole tensor (8 per CU), at least 8 unrolled passes per block
inline int64_t rows_per_block_for(int64_t S, int C, int N) {
  const int rpp = SW_THREADS / (C / 4);
  int64_t target = 2048 / (N > 0 ? N : 1);
  if (target < 1) target = 1;
  int64_t rpb = (S + target - 1) / target;
  const int64_t min_rows = (int64_t)rpp * SW_UNROLL * 8;
  if (rpb < min_rows) rpb = min_rows;
  const int64_t q = (int64_t)rpp * SW_UNROLL;   // whole unrolled passes: only the sample's last block has a tail
  return (rpb + q - 1) / q * q;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int rehr_scale_res_act_fwd_f32(const float* x, int32_t ldx, const float* gate, const float* res,
                                          int32_t ldr, float* y, int32_t ldy, int32_t N, int64_t S, int32_t C,
                                          int32_t act, float slope, void* stream) {
  if (!x || !gate || !y || N < 1 || N > 65535 || S < 1 || C < 4 || C % 4 || C > 1024 || ldx % 4 || ldy % 4 ||
      (res && ldr % 4))
    return REHR_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(gate) || (res && !aligned16(res))) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N);
  hipLaunchKernelGGL(scale_res_act_fwd_kernel, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, x,
                     ldx, gate, res, ldr, y, ldy, S, C, rpb, slope_eff_of(act, slope));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_scale_res_act_bwd_f32(const float* dy, int32_t lddy, const float* y, int32_t ldy, const float* x,
                                          int32_t ldx, const float* gate, float* dx, int32_t lddx, float* dres,
                                          int32_t lddr, double* dgate_acc, int32_t N, int64_t S, int32_t C,
                                          int32_t act, float slope, void* stream) {
  if (!dy || !y || !x || !gate || !dx || !dgate_acc) return REHR_EINVAL;
  if (N < 1 || N > 65535 || S < 1 || C < 4 || C % 4 || C > 1024) return REHR_EINVAL;
  if (lddy % 4 || ldy % 4 || ldx % 4 || lddx % 4 || (dres && lddr % 4)) return REHR_EINVAL;
  if (!aligned16(dy) || !aligned16(y) || !aligned16(x) || !aligned16(dx) || !aligned16(gate) ||
      (dres && !aligned16(dres)))
    return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N);
  hipLaunchKernelGGL(scale_res_act_bwd_kernel, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, dy,
                     lddy, y, ldy, x, ldx, gate, dx, lddx, dres, lddr, dgate_acc, S, C, rpb,
                     slope_eff_of(act, slope));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_add_channel_const_f32(float* x, int32_t ldx, const float* k, int32_t N, int64_t S, int32_t C,
                                          void* stream) {
  if (!x || !k || N < 1 || N > 65535 || S < 1 || C < 4 || C % 4 || C > 1024 || ldx % 4 || !aligned16(x) ||
      !aligned16(k))
    return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N);
  hipLaunchKernelGGL(add_channel_const_kernel, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, x,
                     ldx, k, S, C, rpb);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_instnorm_act_fwd_f32(const float* x, int32_t ldx, const double* stats, const float* gamma,
                                         const float* beta, float* y, int32_t ldy, float* mean_rstd, int32_t N,
                                         int64_t S, int32_t C, float eps, int32_t act, float slope, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || !mean_rstd) return REHR_EINVAL;
  if (N < 1 || N > 65535 || S < 1 || C < 4 || C % 4 || C > 1024 || ldx % 4 || ldy % 4 || !aligned16(x) ||
      !aligned16(y))
    return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N);
  hipLaunchKernelGGL(instnorm_act_fwd_kernel, dim3((unsigned)((S + rpb - 1) / rpb), N), dim3(SW_THREADS), 0, ST, x,
                     ldx, stats, gamma, beta, y, ldy, mean_rstd, S, C, rpb, 1.0 / (double)S, eps,
                     slope_eff_of(act, slope));
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}

extern "C" int rehr_instnorm_act_bwd_f32(const float* dy, int32_t lddy, const float* x, int32_t ldx,
                                         const float* mean_rstd, const float* gamma, const float* beta, float* dx,
                                         int32_t lddx, float* dgamma, float* dbeta, double* red, int32_t N, int64_t S,
                                         int32_t C, int32_t act, float slope, void* stream) {
  if (!dy || !x || !mean_rstd || !gamma || !beta || !dx || !dgamma || !dbeta || !red) return REHR_EINVAL;
  if (N < 1 || N > 65535 || S < 1 || C < 4 || C % 4 || C > 1024) return REHR_EINVAL;
  if (lddy % 4 || ldx % 4 || lddx % 4 || !aligned16(dy) || !aligned16(x) || !aligned16(dx)) return REHR_EINVAL;
  const int64_t rpb = rows_per_block_for(S, C, N);
  const dim3 grid((unsigned)((S + rpb - 1) / rpb), N);
  const float ga = slope_eff_of(act, slope);
  hipLaunchKernelGGL(instnorm_bwd_reduce_kernel, grid, dim3(SW_THREADS), 0, ST, dy, lddy, x, ldx, mean_rstd, gamma,
                     beta, red, S, C, rpb, ga);
  hipLaunchKernelGGL(instnorm_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, red, dgamma, dbeta, N, C);
  hipLaunchKernelGGL(instnorm_bwd_apply_kernel, grid, dim3(SW_THREADS), 0, ST, dy, lddy, x, ldx, mean_rstd, gamma,
                     beta, red, dx, lddx, S, C, rpb, 1.0 / (double)S, ga);
  REHR_LAUNCH_CHECK();
  return REHR_OK;
}
